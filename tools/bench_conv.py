#!/usr/bin/env python
"""Microbenchmark of hnd_conv2d_igemm on the distinct conv shapes of the GHND step (batch 16, 800x1344).
usage: python tools/bench_conv.py [--iters 20] [--only S6,S10] [--batch 16]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

# name: (cin, h, w, cout, k, stride, pad, residual, epilogue_fbn)
SHAPES = {
    'S1_1x1_256-64@200': (256, 200, 336, 64, 1, 1, 0, False),
    'S2_3x3_64-64@200': (64, 200, 336, 64, 3, 1, 1, False),
    'S3_1x1_64-256@200+res': (64, 200, 336, 256, 1, 1, 0, True),
    'S4_3x3_128-128@100': (128, 100, 168, 128, 3, 1, 1, False),
    'S5_1x1_128-512@100+res': (128, 100, 168, 512, 1, 1, 0, True),
    'S5b_1x1_512-128@100': (512, 100, 168, 128, 1, 1, 0, False),
    'S6_3x3_256-256@50': (256, 50, 84, 256, 3, 1, 1, False),
    'S7_1x1_256-1024@50+res': (256, 50, 84, 1024, 1, 1, 0, True),
    'S7b_1x1_1024-256@50': (1024, 50, 84, 256, 1, 1, 0, False),
    'S8_3x3_512-512@25': (512, 25, 42, 512, 3, 1, 1, False),
    'S9_1x1_512-2048@25+res': (512, 25, 42, 2048, 1, 1, 0, True),
    'S9b_1x1_2048-512@25': (2048, 25, 42, 512, 1, 1, 0, False),
    'S10_3x3_256-256@200': (256, 200, 336, 256, 3, 1, 1, False),
    'S11_2x2_256-256@201': (256, 201, 337, 256, 2, 1, 0, False),
    'S12_3x3s2_128-128@200': (128, 200, 336, 128, 3, 2, 1, False),
    'S13_1x1_256-256@200': (256, 200, 336, 256, 1, 1, 0, False),      # the shape class of the Winograd GEMMs
    'S14_1x1_512-512@100': (512, 100, 168, 512, 1, 1, 0, False),
    'S15_1x1_128-128@200': (128, 200, 336, 128, 1, 1, 0, False),
    'S16_1x1_128-256@200': (128, 200, 336, 256, 1, 1, 0, False),
    'S17_1x1_512-256@100': (512, 100, 168, 256, 1, 1, 0, False),
    'S18_1x1_256-256@50': (256, 50, 84, 256, 1, 1, 0, False),
    'S19_1x1_256-256@100': (256, 100, 168, 256, 1, 1, 0, False),
    'S20_1x1_64-256@200': (64, 200, 336, 256, 1, 1, 0, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--only', default='')
    ap.add_argument('--tiles', default='', help="comma list of HND_DEBUG_PICKER=igemm_tile=N overrides to sweep, e.g. '0,1,2,3'")
    args = ap.parse_args()
    dev = 'cuda:0'
    only = [s for s in args.only.split(',') if s]
    tot_ms = tot_fl = 0.0
    for name, (cin, h, w, cout, k, s, p, res) in SHAPES.items():
        if only and not any(name.startswith(o) for o in only):
            continue
        n = args.batch
        x = torch.randn(n, h, w, cin, device=dev)
        wt = torch.randn(cout, cin, k, k, device=dev) * (1.0 / (cin * k * k) ** 0.5)
        oh, ow = ops.conv_out_size(h, k, s, p), ops.conv_out_size(w, k, s, p)
        y = torch.empty(n, oh, ow, cout, device=dev)
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        r = torch.randn(n, oh, ow, cout, device=dev) if res else None
        pk = ops.pack_weights(wt)
        line = '%-26s' % name
        for tile in ([t for t in args.tiles.split(',') if t] or [None]):
            if tile is None:
                os.environ.pop('HND_DEBUG_PICKER', None)
            else:
                os.environ['HND_DEBUG_PICKER'] = 'igemm_tile=%s' % tile
            l = ops.conv_forward(x, pk, y, k, s, p, epi_scale=sc, epi_shift=sh, res1=r, relu=True)
            for _ in range(2):
                l.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                l.run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            line += '  %-14s %7.3f ms %6.1f TF' % (l.variant, ms, l.flops / ms / 1e9)
        if not args.tiles:
            tot_ms += ms
            tot_fl += l.flops
        print(line, flush=True)
    if tot_ms:
        print('%-26s %-10s %8.3f ms  %7.1f TFLOP/s' % ('TOTAL', '', tot_ms, tot_fl / tot_ms / 1e9))


if __name__ == '__main__':
    main()

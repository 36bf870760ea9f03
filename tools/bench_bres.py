#!/usr/bin/env python
"""A/B of the B-resident persistent GEMM (csrc/conv_bres.hip) against the tiled implicit-GEMM kernel on the tap-free
K <= 512 launches of the GHND step (batch 16): 1x1 convs of the frozen trunk / FPN and Winograd-style grouped GEMMs.
Outputs must be identical bits.   usage: python tools/bench_bres.py [--iters 10] [--batch 16]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

# name: (cin, h, w, cout, stride, residual, prologue, groups)   groups > 1: Winograd-like grouped GEMM over h*w rows
SHAPES = {
    'wino36_256-256@200 (fpn.layer0)': (256, 50, 84, 256, 1, False, False, 36),
    'wino36_256-256@50 (layer3.conv2)': (256, 13, 21, 256, 1, False, False, 36),
    'wino36_128-128@100 (layer2.conv2)': (128, 25, 42, 128, 1, False, False, 36),
    'wino25_256-256@201 (head conv7)': (256, 51, 85, 256, 1, False, False, 25),
    '1x1_256-256@200 (fpn.inner0)': (256, 200, 336, 256, 1, True, False, 1),
    '1x1_256-128@200 (layer2.0.conv1)': (256, 200, 336, 128, 1, False, False, 1),
    '1x1_256-64@200 (layer1.conv1)': (256, 200, 336, 64, 1, False, False, 1),
    '1x1_64-256@200+res (layer1.conv3)': (64, 200, 336, 256, 1, True, False, 1),
    '1x1s2_256-512@200 (layer2.0.down)': (256, 200, 336, 512, 2, False, False, 1),
    '1x1_128-512@100+res (layer2.conv3)': (128, 100, 168, 512, 1, True, False, 1),
    '1x1_512-128@100 (layer2.conv1)': (512, 100, 168, 128, 1, False, False, 1),
    '1x1_512-256@100 (layer3.0.conv1)': (512, 100, 168, 256, 1, False, False, 1),
    '1x1_256-1024@50+res (layer3.conv3)': (256, 50, 84, 1024, 1, True, False, 1),
    '1x1_256-1024@50 pro (conv1.dgrad)': (256, 50, 84, 1024, 1, False, True, 1),
    '1x1_512-2048@25+res (layer4.conv3)': (512, 25, 42, 2048, 1, True, False, 1),
    '1x1_512-256@100 (fpn.inner1)': (512, 100, 168, 256, 1, True, False, 1),
    '1x1_1024-256@50 (layer3.conv1)': (1024, 50, 84, 256, 1, False, False, 1),
    '1x1_1024-512@50 (layer4.0.conv1)': (1024, 50, 84, 512, 1, False, False, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = 'cuda:0'
    os.environ['HND_DEBUG_PICKER'] = 'bres_all'            # A/B on every eligible shape, not only where the picker takes it
    modes = (('tiled', '0', '1'), ('bres', '512', '0'), ('bres2', '512', '1'))      # name, HND_BRES, HND_BRES2
    tot = {m[0]: [0.0, 0.0] for m in modes}
    for name, (cin, h, w, cout, s, res, pro, groups) in SHAPES.items():
        if args.only and args.only not in name:
            continue
        n = args.batch
        torch.manual_seed(0)
        if groups > 1:          # rows = groups * tiles_pad, one packed weight matrix per group
            tiles_pad = (n * h * w + 127) // 128 * 128
            x = torch.randn(1, 1, groups * tiles_pad, cin, device=dev)
            y = torch.empty(1, 1, groups * tiles_pad, cout, device=dev)
            pks = [ops.pack_weights(torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5) for _ in range(groups)]
            pk = ops.PackedWeight.__new__(ops.PackedWeight)
            pk.buf = torch.cat([p.buf for p in pks])
            pk.kdim, pk.rows, pk.chan_pad, pk.chan_real = pks[0].kdim, cout, cin, cin
            stride = pks[0].buf.numel()
        else:
            x = torch.randn(n, h, w, cin, device=dev)
            oh, ow = ops.conv_out_size(h, 1, s, 0), ops.conv_out_size(w, 1, s, 0)
            y = torch.empty(n, oh, ow, cout, device=dev)
            pk = ops.pack_weights(torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5)
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        r = torch.randn_like(y) if res else None
        ps = (torch.rand(cin, device=dev) + 0.5) if pro else None
        pb = torch.randn(cin, device=dev) if pro else None
        outs, line = {}, '%-38s' % name
        for mode, e1, e2 in modes:
            os.environ['HND_BRES'], os.environ['HND_BRES2'] = e1, e2
            if groups > 1:
                l = ops.conv_desc(x, pk, y, kh=1, kw=1, oh=1, ow=groups * tiles_pad, sh=1, dh=1, bh=0, sw=1, dw=1, bw=0,
                                  cout=cout)
                l.desc.w_group_rows, l.desc.w_group_stride = tiles_pad, stride
                l.flops = 2 * groups * tiles_pad * cout * cin
            else:
                l = ops.conv_forward(x, pk, y, 1, s, 0, epi_scale=sc, epi_shift=sh, res1=r, relu=True, pro_scale=ps,
                                     pro_shift=pb, pro_relu=pro)
            l.refresh_variant()
            y.fill_(float('nan'))
            for _ in range(2):
                l.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                l.run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            outs[mode] = y.clone()
            tot[mode][0] += ms
            tot[mode][1] += l.flops
            line += '  %-13s %6.3f ms %5.1f TF' % (l.variant, ms, l.flops / ms / 1e9)
        same = all(torch.equal(outs['tiled'], outs[m]) for m in ('bres', 'bres2')) and \
            not bool(torch.isnan(outs['bres2']).any())
        print(line + ('   bits equal' if same else '   *** DIFFERENT (max |d| %.3e)'
                                                   % float((outs['tiled'] - outs['bres2']).abs().max())), flush=True)
    os.environ.pop('HND_BRES', None)
    os.environ.pop('HND_BRES2', None)
    for mode, (ms, fl) in tot.items():
        if ms:
            print('TOTAL %-6s %8.3f ms  %7.1f TFLOP/s' % (mode, ms, fl / ms / 1e9))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Microbenchmark of hnd_conv2d_wgrad on the weight-gradient shapes of the GHND head (batch 16, 800x1344 input).
usage: python tools/bench_wgrad.py [--iters 10] [--only W1,W3]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

# name: (cin, h, w, cout, k, stride, pad)
SHAPES = {
    'W1_2x2_256-256@201': (256, 201, 337, 256, 2, 1, 0),      # the class of layer1.conv1 / conv7 (direct form)
    'W2_2x2_256-64@202': (256, 202, 338, 64, 2, 1, 0),        # layer1.conv2: cout 64, 1024 columns
    'W3_7x7s2_4-64@800': (4, 800, 1344, 64, 7, 2, 3),         # stem
    'W4_2x2_64-256@200': (64, 201, 337, 256, 2, 1, 0),
    'W5_1x1_256-256@200': (256, 200, 336, 256, 1, 1, 0),      # the Winograd-domain wgrad GEMM class (per position)
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = 'cuda:0'
    only = [s for s in args.only.split(',') if s]
    for name, (cin, h, w, cout, k, s, p) in SHAPES.items():
        if only and not any(name.startswith(o) for o in only):
            continue
        n = args.batch
        oh, ow = ops.conv_out_size(h, k, s, p), ops.conv_out_size(w, k, s, p)
        x = torch.randn(n, h, w, cin, device=dev)
        dy = torch.randn(n, oh, ow, ops.round_up(cout, 4), device=dev)
        dw = torch.empty(cout, min(cin, 3) if cin == 4 else cin, k, k, device=dev)
        l = ops.conv_wgrad(x, dy, dw, k, s, p)
        for _ in range(2):
            l.run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            l.run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        print('%-24s %-10s %7.3f ms %6.1f TF' % (name, l.variant, ms, l.flops / ms / 1e9), flush=True)


if __name__ == '__main__':
    main()

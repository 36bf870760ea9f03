#!/usr/bin/env python
"""Ceiling check for the fp32 GEMM kernel: the vendor libraries' sgemm (rocBLAS / hipBLASLt behind torch.mm, plain
fp32 -- no TF32 on ROCm) on the GEMM shapes of the step's 1x1 convolutions, next to hnd_conv2d_igemm on the same
shapes.  A measurement aid only: the product never calls a library GEMM.

usage: python tools/bench_sgemm_lib.py [--iters 10]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

# (M = batch*H*W, N = Cout, K = Cin) of the launches that carry the step (profiles/r02_per_launch_events.txt)
SHAPES = [(1075200, 256, 256), (1075200, 256, 64), (1075200, 64, 256), (268800, 512, 128), (268800, 128, 512),
          (268800, 256, 512), (67200, 1024, 256), (67200, 256, 1024), (67200, 256, 2304), (16800, 2048, 512),
          (16800, 512, 2048), (16800, 512, 4608)]


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    args = ap.parse_args()
    dev = 'cuda:0'
    torch.backends.cuda.matmul.allow_tf32 = False
    print('%-24s %10s %10s %10s %10s' % ('M x N x K', 'lib nn', 'lib nt', 'igemm', 'TFLOP/s'))
    for m, n, k in SHAPES:
        a = torch.randn(m, k, device=dev)
        b_kn = torch.randn(k, n, device=dev)
        b_nk = torch.randn(n, k, device=dev)
        out = torch.empty(m, n, device=dev)
        fl = 2.0 * m * n * k
        t_nn = timed(lambda: torch.mm(a, b_kn, out=out), args.iters)
        t_nt = timed(lambda: torch.mm(a, b_nk.t(), out=out), args.iters)
        # the same contraction as a 1x1 convolution on the product's kernel (geometry 1 x rows x M/rows "pixels")
        rows = 1
        while m % (rows * 2) == 0 and m // (rows * 2) >= 1024:
            rows *= 2
        x = a.view(1, rows, m // rows, k)
        y = out.view(1, rows, m // rows, n)
        launch = ops.conv_forward(x, ops.pack_weights(b_nk.view(n, k, 1, 1).contiguous()), y, 1, 1, 0)
        t_ig = timed(launch.run, args.iters)
        print('%-24s %7.1f TF %7.1f TF %7.1f TF' % ('%d x %d x %d' % (m, n, k), fl / t_nn / 1e9, fl / t_nt / 1e9,
                                                     fl / t_ig / 1e9), flush=True)


if __name__ == '__main__':
    main()

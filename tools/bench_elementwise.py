#!/usr/bin/env python3
"""The HBM-bound elementwise kernels of the step in isolation, at the step's sizes (batch 16, 800x1344), against a torch
elementwise op moving the same bytes (what a plain streaming kernel reaches on this box).

usage (GPU box):  python3 tools/bench_elementwise.py [--reps 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--batch', type=int, default=16)
    a = ap.parse_args()
    from hnd_ghnd_object_detectors_amd import ops
    dev = torch.device('cuda:0')
    n = a.batch
    g = torch.Generator().manual_seed(1)
    rows = []

    def report(name, nbytes, ms):
        rows.append((name, nbytes / 1e9, ms, nbytes / ms / 1e9))
        print('%-34s %7.3f GB  %7.3f ms  %6.3f TB/s' % (name, nbytes / 1e9, ms, nbytes / ms / 1e9), flush=True)

    # ---- reference: torch elementwise ops over the layer1 map (16 x 200 x 336 x 256)
    x = torch.randn(n, 200, 336, 256, device=dev)
    y = torch.empty_like(x)
    report('torch.mul(x, 1.5, out=y)', 8 * x.numel(), timed(lambda: torch.mul(x, 1.5, out=y), a.reps))
    report('torch y.copy_(x)', 8 * x.numel(), timed(lambda: y.copy_(x), a.reps))
    z = torch.randn_like(x)
    report('torch.add(x, z, out=y)', 12 * x.numel(), timed(lambda: torch.add(x, z, out=y), a.reps))

    # ---- affine_relu of the head output (+ mask nibbles)
    sc, sh = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
    bits = ops.mask_nibbles_like(y)
    report('affine_relu', 8 * x.numel(), timed(lambda: ops.affine_relu(x, sc, sh, y, True), a.reps))
    report('affine_relu + nibbles', 8 * x.numel() + bits.numel(),
           timed(lambda: ops.affine_relu(x, sc, sh, y, True, mask_out=bits), a.reps))

    # ---- BN backward of a 256-channel head tensor
    mu, rs = torch.randn(256, device=dev) * 0.1, torch.rand(256, device=dev) + 0.5
    part = torch.empty(ops.bn_bwd_ntiles(x.numel() // 256), 2, 256, device=dev)
    k123 = torch.randn(3, 256, device=dev)
    report('bn_bwd_reduce', 8 * x.numel(), timed(lambda: ops.bn_bwd_reduce(z, x, sc, sh, mu, rs, True, part), a.reps))
    report('bn_bwd_apply', 12 * x.numel(), timed(lambda: ops.bn_bwd_apply(z, x, sc, sh, k123, True, y), a.reps))

    # ---- fused loss + gradient of the layer1 pair
    ml = ops.MseLaunch([(x, z, y, 1.0, False)], dev)
    report('mse (layer1 pair, + gradient)', 12 * x.numel(), timed(lambda: ml.run(), a.reps))
    del x, y, z, bits, part

    # ---- stem pool: 16 x 400 x 672 x 64 -> 200 x 336
    a0 = torch.randn(n, 400, 672, 64, device=dev)
    x0 = torch.empty(n, 200, 336, 64, device=dev)
    idx = torch.empty(n, 200, 336, 64, dtype=torch.uint8, device=dev)
    report('maxpool_fwd', 4 * a0.numel() + 5 * x0.numel(), timed(lambda: ops.maxpool_fwd(a0, x0, idx), a.reps))
    gx = torch.randn_like(x0)
    dconv = torch.empty_like(a0)
    s64 = torch.rand(64, device=dev) + 0.5
    report('maxpool_bwd_relu_scale', 8 * a0.numel() + 5 * x0.numel(),
           timed(lambda: ops.maxpool_bwd_relu_scale(gx, idx, a0, s64, dconv), a.reps))
    ops.sync_check()


if __name__ == '__main__':
    main()

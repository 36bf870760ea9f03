#!/usr/bin/env python3
"""The Winograd transforms of the step in isolation (batch 16): input and output transform of every F(6x6,3x3) shape and
of the head's F(6x6,2x2) launches, with the bytes each must move.

usage (GPU box):  python3 tools/bench_wino_transforms.py [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

SHAPES3 = {'fpn.layer0 256@200x336': (256, 200, 336, 256), 'layer1.conv2 64@200x336': (64, 200, 336, 64),
           'fpn.layer1 256@100x168': (256, 100, 168, 256), 'layer2.conv2 128@100x168': (128, 100, 168, 128),
           'layer3.conv2 256@50x84': (256, 50, 84, 256), 'layer4.conv2 512@25x42': (512, 25, 42, 512)}
SHAPES2 = {'head conv7 256->256 @202x338': (256, 202, 338, 256, 0), 'head conv6 128->256 @203x339': (128, 203, 339, 256, 0),
           'head conv2 256->64 @202x338': (256, 202, 338, 64, 1)}


def timed(fn, reps):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    dev, n = 'cuda:0', 16
    tot = [0.0, 0.0]
    for name, (cin, h, w, cout) in SHAPES3.items():
        x = torch.randn(n, h, w, cin, device=dev)
        wt = torch.randn(cout, cin, 3, 3, device=dev) * (1.0 / (cin * 9) ** 0.5)
        y = torch.empty(n, h, w, cout, device=dev)
        ww = ops.WinoWeights(wt, tile=6)
        nv, nm = ops.WinoConv.scratch_elems(n, h, w, cin, cout, 6)
        v, m = torch.empty(nv, device=dev), torch.randn(nm, device=dev)
        wino = ops.WinoConv(x, ww, y, v, m, relu=True)
        (si, _), _, (so, _) = wino.launches('t')
        ti, to = timed(wino._run_input, a.reps), timed(wino._run_output, a.reps)
        tot[0] += ti
        tot[1] += to
        print('%-28s input %7.3f ms %6.3f TB/s | output %7.3f ms %6.3f TB/s' % (
            name, ti, si.hbm_bytes / ti / 1e9, to, so.hbm_bytes / to / 1e9), flush=True)
        del x, y, v, m
    print('F(6x6,3x3) sum: input %.3f ms, output %.3f ms' % tuple(tot))
    for name, (cin, h, w, cout, pad) in SHAPES2.items():
        x = torch.randn(n, h, w, cin, device=dev)
        wt = torch.randn(cout, cin, 2, 2, device=dev) * (1.0 / (cin * 4) ** 0.5)
        oh, ow = h + 2 * pad - 1, w + 2 * pad - 1
        y = torch.empty(n, oh, ow, cout, device=dev)
        ww = ops.Wino2Weights(wt.contiguous(), dgrad=False, tile=6)
        nv, nm = ops.Wino2Conv.scratch_elems(n, oh, ow, cin, cout, 6)
        v, m = torch.empty(nv, device=dev), torch.randn(nm, device=dev)
        wino = ops.Wino2Conv(x, ww, y, v, m, pad)
        (si, _), _, (so, _) = wino.launches('t')
        ti, to = timed(wino._run_input, a.reps), timed(wino._run_output, a.reps)
        print('%-28s input %7.3f ms %6.3f TB/s | output %7.3f ms %6.3f TB/s' % (
            name, ti, si.hbm_bytes / ti / 1e9, to, so.hbm_bytes / to / 1e9), flush=True)
        del x, y, v, m
    ops.sync_check()


if __name__ == '__main__':
    main()

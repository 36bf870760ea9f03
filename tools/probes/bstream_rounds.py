#!/usr/bin/env python
"""Fixed cost vs per-tile cost of the B-streamed GEMM: T = 256 * r tiles (exactly r per workgroup) of 128 x 128 x K.
usage: python tools/probes/bstream_rounds.py [K] [cout]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = 'cuda:0'
os.environ['HND_BRES'] = '0'
for mode in ('all',):
    os.environ['HND_BSTREAM'] = '1' if mode == 'all' else mode
    os.environ['HND_DEBUG_PICKER'] = 'bstream_all' if mode == 'all' else ''
    for r in (2, 4, 8):
        rows = 256 * r * 128 // (cout // 128)
        x = torch.randn(1, 128, rows // 128, K, device=dev)
        y = torch.empty(1, 128, rows // 128, cout, device=dev)
        pk = ops.pack_weights(torch.randn(cout, K, 1, 1, device=dev) / K ** 0.5)
        l = ops.conv_forward(x, pk, y, 1, 1, 0, relu=True)
        l.refresh_variant()
        for _ in range(3):
            l.run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            l.run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('%-14s K=%d cout=%d tiles/wg=%d  %.3f ms  %.1f TF' % (l.variant, K, cout, r, ms, l.flops / ms / 1e9), flush=True)

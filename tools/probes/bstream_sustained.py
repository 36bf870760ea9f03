#!/usr/bin/env python
"""Four seconds of back-to-back B-streamed GEMM launches (K = 4096, 8 tiles per workgroup) for clock sampling:
tools/sample_clocks.sh out.txt -- python tools/probes/bstream_sustained.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

os.environ['HND_BRES'] = '0'
os.environ['HND_DEBUG_PICKER'] = 'bstream_all'
dev = 'cuda:0'
K, cout, r = 4096, 256, 8
rows = 256 * r * 128 // (cout // 128)
x = torch.randn(1, 128, rows // 128, K, device=dev)
y = torch.empty(1, 128, rows // 128, cout, device=dev)
pk = ops.pack_weights(torch.randn(cout, K, 1, 1, device=dev) / K ** 0.5)
l = ops.conv_forward(x, pk, y, 1, 1, 0, relu=True)
for _ in range(3):
    l.run()
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < 4.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        l.run()
    e1.record()
    torch.cuda.synchronize()
    n += 50
    ms = e0.elapsed_time(e1) / 50
print('%s  %.3f ms per launch  %.1f TF  (%d launches)' % (l.variant, ms, l.flops / ms / 1e9, n))

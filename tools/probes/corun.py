#!/usr/bin/env python
"""Do an MFMA-bound GEMM and an HBM-bound kernel overlap when issued on two streams?  Times N GEMM launches alone, M
streaming launches alone, and both sets concurrently.  usage: python tools/probes/corun.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

dev = 'cuda:0'
n, h, w, c = 16, 200, 336, 256
x = torch.randn(n, h, w, c, device=dev)
y = torch.empty(n, h, w, c, device=dev)
pk = ops.pack_weights(torch.randn(c, c, 1, 1, device=dev) / c ** 0.5)
a = torch.randn(n, h, w, c, device=dev)
b = torch.empty_like(a)
sc, sh = torch.rand(c, device=dev), torch.randn(c, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    s1.synchronize()
    s2.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for mode, env in (('tiled 3 blocks/CU', {'HND_BRES': '0', 'HND_BSTREAM': '0'}), ('bres2 (one wave per SIMD)', {})):
    for k in ('HND_BRES', 'HND_BSTREAM'):
        os.environ.pop(k, None)
    os.environ.update(env)
    l = ops.conv_forward(x, pk, y, 1, 1, 0, relu=True)
    NG, NH = 20, 50

    def gemms():
        s1.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            for _ in range(NG):
                l.run()

    def streams():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            for _ in range(NH):
                ops.affine_relu(a, sc, sh, b, True)

    for _ in range(2):
        gemms(); streams()
    tg, th = timed(gemms), timed(streams)
    tb = timed(lambda: (gemms(), streams()))
    print('%-28s %-13s GEMMs alone %.2f ms | streaming kernels alone %.2f ms | both %.2f ms  (sum %.2f, max %.2f)' % (
        mode, l.variant, tg, th, tb, tg + th, max(tg, th)), flush=True)

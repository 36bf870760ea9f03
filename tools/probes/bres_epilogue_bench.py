#!/usr/bin/env python
"""What do the epilogue operands of the frozen Bottlenecks' 1x1 launches cost on the native fp32 kernels?  Same GEMM, four
epilogues: plain / + residual / + residual + ReLU-mask nibbles out (the student's conv3) / + residual + mask nibbles in (conv1's
data gradient).  ms per launch over 20 launches (L2 / MALL-warm operands, as inside the step), TF and the bytes each moves.

    python tools/probes/bres_epilogue_bench.py > gpurun_out/bres_epilogue.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

DEV = torch.device('cuda:0')
SHAPES = [('layer2 conv3 128->512', 128, 512, 16, 100, 168), ('layer3 conv3 256->1024', 256, 1024, 16, 50, 84),
          ('layer2 conv1.dgrad-like 128->512', 128, 512, 16, 100, 168), ('fpn.inner0-like 256->256', 256, 256, 16, 200, 336)]


def timed(launch, reps=20):
    for _ in range(5):
        launch.run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        launch.run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    g = torch.Generator().manual_seed(0)
    print('%-34s %-26s %-12s %8s %8s %8s %8s' % ('shape', 'epilogue', 'kernel', 'ms', 'TF', 'GB', 'TB/s'))
    for name, cin, cout, n, h, w in SHAPES:
        x = torch.randn(n, h, w, cin, generator=g).to(DEV)
        wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV)
        r = torch.randn(n, h, w, cout, generator=g).to(DEV)
        es, eb = torch.rand(cout, generator=g).to(DEV) + 0.5, torch.randn(cout, generator=g).to(DEV)
        pk = ops.pack_weights(wt)
        y = torch.empty(n, h, w, cout, device=DEV)
        bits = ops.mask_nibbles_like(y)
        bits.random_(0, 16)
        m = n * h * w
        gf = 2.0 * m * cin * cout / 1e9
        base = 4.0 * m * (cin + cout)
        for label, kw, nbytes in (('plain (scale, shift, ReLU)', dict(relu=True), base),
                                  ('+ residual', dict(relu=True, res1=r), base + 4.0 * m * cout),
                                  ('+ residual + mask_out', dict(relu=True, res1=r, mask_out=bits), base + 4.25 * m * cout),
                                  ('+ residual + mask_bits', dict(res1=r, mask_bits=bits), base + 4.25 * m * cout)):
            l = ops.conv_forward(x, pk, y, 1, 1, 0, epi_scale=es, epi_shift=eb, **kw)
            t = timed(l)
            print('%-34s %-26s %-12s %8.3f %8.1f %8.2f %8.2f' % (name, label, l.variant, t, gf / t, nbytes / 1e9, nbytes / t / 1e9))
    ops.sync_check()


if __name__ == '__main__':
    main()

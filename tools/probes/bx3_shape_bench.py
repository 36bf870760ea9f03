#!/usr/bin/env python
"""Per-shape A/B of the B-resident bf16x3 emulation kernel (csrc/conv_bx3.hip) against the native fp32 picker's choice on the 1x1
shapes of the step it does not take yet or has just started to take: ms per launch (HIP events over 20 launches after 5
warm-ups, operands re-used, so L2 / MALL-warm like inside the step) and TF-equivalent.

    python tools/probes/bx3_shape_bench.py > gpurun_out/bx3_shapes.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

DEV = torch.device('cuda:0')
SHAPES = [  # name, cin, cout, n, h, w, stride, residual
    ('layer3.x.conv1 1024->256', 1024, 256, 16, 50, 84, 1, False),
    ('layer3.x.conv3 256->1024 +res', 256, 1024, 16, 50, 84, 1, True),
    ('layer4.x.conv1 2048->512', 2048, 512, 16, 25, 42, 1, False),
    ('layer4.x.conv3 512->2048 +res', 512, 2048, 16, 25, 42, 1, True),
    ('layer4.0.downsample 1024->2048 s2', 1024, 2048, 16, 50, 84, 2, False),
    ('layer4.0.conv1 1024->512', 1024, 512, 16, 50, 84, 1, False),
    ('fpn.inner3 2048->256', 2048, 256, 16, 25, 42, 1, False),
    ('fpn.inner2 1024->256', 1024, 256, 16, 50, 84, 1, False),
    ('layer2.x.conv1 512->128', 512, 128, 16, 100, 168, 1, False),
    ('fpn.inner0 256->256 @200x336', 256, 256, 16, 200, 336, 1, False),
    ('layer2.x.conv3 128->512 +res @100x168', 128, 512, 16, 100, 168, 1, True),
    ('layer3.0.conv1 512->256', 512, 256, 16, 100, 168, 1, False),
]


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
    print('%-38s %10s %8s %10s %8s %6s' % ('shape', 'native ms', 'TF', 'bx3 ms', 'TF-eq', 'x'))
    for name, cin, cout, n, h, w, stride, res in SHAPES:
        x = torch.randn(n, h, w, cin, generator=g).to(DEV)
        wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV)
        oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
        r = torch.randn(n, oh, ow, cout, generator=g).to(DEV) if res else None
        es, eb = torch.rand(cout, generator=g).to(DEV) + 0.5, torch.randn(cout, generator=g).to(DEV)
        pk = ops.pack_weights(wt)
        y = torch.empty(n, oh, ow, cout, device=DEV)
        kw = dict(epi_scale=es, epi_shift=eb, relu=True, res1=r)
        with ops.emulation('off'):
            l0 = ops.conv_forward(x, pk, y, 1, stride, 0, **kw)
        t0 = timed(l0)
        pk.bx3 = ops.bx3_image(pk.buf, ops.round_up(cout, 64), cin, force=True)
        pk.bxs = pk.useds = None                    # (the B-streamed build has its own bench: tools/bench_bxs.py)
        with ops.emulation('force'):
            l1 = ops.conv_forward(x, pk, y, 1, stride, 0, **kw)
        gf = 2.0 * n * oh * ow * cin * cout / 1e9
        if l1.variant != 'bx3_64':
            print('%-38s %10.3f %8.1f %10s  (%s: not eligible)' % (name, t0, gf / t0, '-', l0.variant))
            continue
        t1 = timed(l1)
        print('%-38s %10.3f %8.1f %10.3f %8.1f %6.2f  (%s)' % (name, t0, gf / t0, t1, gf / t1, t0 / t1, l0.variant))
    ops.sync_check()


if __name__ == '__main__':
    main()

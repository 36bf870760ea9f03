#!/usr/bin/env python
"""Direct implicit-GEMM vs Winograd F(4x4,3x3) and F(6x6,3x3) on the 3x3 stride-1 shapes of the step (batch 16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

SHAPES = {'fpn.layer0 256@200x336': (256, 200, 336, 256), 'fpn.layer1 256@100x168': (256, 100, 168, 256),
          'layer3.conv2 256@50x84': (256, 50, 84, 256), 'layer4.conv2 512@25x42': (512, 25, 42, 512),
          'layer2.conv2 128@100x168': (128, 100, 168, 128), 'fpn.layer2 256@50x84': (256, 50, 84, 256),
          'layer1.conv2 64@200x336': (64, 200, 336, 64), 'fpn.layer3 256@25x42': (256, 25, 42, 256)}


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev, n = 'cuda:0', 16
    for name, (cin, h, w, cout) in SHAPES.items():
        x = torch.randn(n, h, w, cin, device=dev)
        wt = torch.randn(cout, cin, 3, 3, device=dev) * (1.0 / (cin * 9) ** 0.5)
        sh = torch.randn(cout, device=dev)
        y0, y1 = torch.empty(n, h, w, cout, device=dev), torch.empty(n, h, w, cout, device=dev)
        direct = ops.conv_forward(x, ops.pack_weights(wt), y0, 3, 1, 1, epi_shift=sh, relu=True)
        td = timed(direct.run)
        line = '%-26s direct %7.3f ms (%5.1f TF)' % (name, td, direct.flops / td / 1e9)
        for tile in (4, 6):
            ww = ops.WinoWeights(wt, tile=tile)
            nv, nm = ops.WinoConv.scratch_elems(n, h, w, cin, cout, tile)
            v, m = torch.empty(nv, device=dev), torch.empty(nm, device=dev)
            wino = ops.WinoConv(x, ww, y1, v, m, epi_shift=sh, relu=True)
            tw, tg = timed(wino.run), timed(wino.gemm.run)
            err = float((y0 - y1).abs().max() / y0.abs().max())
            line += ' | F(%d,3) %7.3f ms (gemm %6.3f ms %5.1f TF %s) x%.2f err %.1e' % (
                tile, tw, tg, wino.gemm.flops / tg / 1e9, wino.gemm.variant, td / tw, err)
            del v, m
        print(line, flush=True)


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Direct implicit-GEMM vs Winograd F(4x4,2x2) on the student head's 2x2 convolutions (batch 16, 200x336 maps)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402
from tools.bench_wino import timed  # noqa: E402

# name: (cin, h, w, cout, pad)
SHAPES = {'conv0 64->64 p1': (64, 200, 336, 64, 1), 'conv1 64->256 p1': (64, 201, 337, 256, 1),
          'conv2 256->64 p1': (256, 202, 338, 64, 1), 'conv5 64->128 p0': (64, 203, 339, 128, 0),
          'conv6 128->256 p0': (128, 202, 338, 256, 0), 'conv7 256->256 p0': (256, 201, 337, 256, 0)}


def main():
    dev, n = 'cuda:0', 16
    for name, (cin, h, w, cout, pad) in SHAPES.items():
        x = torch.randn(n, h, w, cin, device=dev)
        wt = torch.randn(cout, cin, 2, 2, device=dev) * (1.0 / (cin * 4) ** 0.5)
        oh, ow = h + 2 * pad - 1, w + 2 * pad - 1
        y0, y1 = torch.empty(n, oh, ow, cout, device=dev), torch.empty(n, oh, ow, cout, device=dev)
        st0 = torch.empty(ops.stats_tiles(n * oh * ow), 2, cout, device=dev)
        direct = ops.conv_forward(x, ops.pack_weights(wt), y0, 2, 1, pad, stats=st0)
        td = timed(direct.run)
        line = '%-20s direct %7.3f ms (%5.1f TF)' % (name, td, direct.flops / td / 1e9)
        for tile, dgrad in ((4, False), (4, True), (6, False), (6, True)):
            if dgrad:   # data gradient of the same conv: dy [oh, ow, cout] -> dx [h, w, cin]
                src, dst, wpad = y0, torch.empty(n, h, w, cin, device=dev), 1 - pad
                ww = ops.Wino2Weights(wt, dgrad=True, tile=tile)
                nv, nm = ops.Wino2Conv.scratch_elems(n, h, w, cout, cin, tile)
                ls, _ = ops.conv_dgrad(y0, wt, torch.empty(n, h, w, cin, device=dev), 2, 1, pad)
                tdir = timed(lambda: [l.run() for l in ls])
            else:
                src, dst, wpad = x, y1, pad
                ww = ops.Wino2Weights(wt, tile=tile)
                nv, nm = ops.Wino2Conv.scratch_elems(n, oh, ow, cin, cout, tile)
                tdir = td
            v, m = torch.empty(nv, device=dev), torch.empty(nm, device=dev)
            st = None if dgrad else torch.empty(ops.Wino2Conv.stats_blocks(n, oh, ow, cout, tile), 2, cout, device=dev)
            wino = ops.Wino2Conv(src, ww, dst, v, m, wpad, stats=st)
            tw, tg = timed(wino.run), timed(wino.gemm.run)
            line += ' | F(%d,2) %s direct %6.3f wino %6.3f ms (gemm %6.3f, %5.1f TF %s) x%.2f' % (
                tile, 'dgrad' if dgrad else 'fwd', tdir, tw, tg, wino.gemm.flops / tg / 1e9, wino.gemm.variant,
                tdir / tw)
            if not dgrad:
                line += ' err %.1e' % float((y0 - y1).abs().max() / y0.abs().max())
                # weight gradient: direct split-K wgrad vs the Winograd-domain one that reuses this V
                dw0, dw1 = torch.empty_like(wt), torch.empty_like(wt)
                wdir = ops.conv_wgrad(x, y0, dw0, 2, 1, pad)
                sbuf = torch.empty((tile + 1) ** 2 * cout * cin, device=dev)
                wwin = ops.Wino2Wgrad(wino, y0, dw1, m, sbuf)
                t0, t1 = timed(wdir.run), timed(wwin.run)
                line += ' | wgrad direct %6.3f wino %6.3f ms x%.2f err %.1e' % (
                    t0, t1, t0 / t1, float((dw0 - dw1).abs().max() / dw0.abs().max()))
            del v, m
        print(line, flush=True)


if __name__ == '__main__':
    main()

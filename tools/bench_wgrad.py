#!/usr/bin/env python
"""A/B of the two MFMA weight-gradient kernels -- the LDS-staged split-K kernel (csrc/conv_wgrad.hip, HND_WGRAD_RING=0)
and the ring kernel (csrc/conv_wgrad_ring.hip) -- on the weight-gradient launches of the GHND step (batch 16, 800x1344
input): the direct 2x2 head convs and the grouped Winograd-domain reductions of conv6 / conv7 / conv2.
usage: python tools/bench_wgrad.py [--iters 10] [--only W1,W3]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

# name: (cin, h, w, cout, k, stride, pad, prologue, groups)     groups > 1: 1x1 over `w` tiles per group (h = 1)
SHAPES = {
    'W1 conv1 2x2 64->256 @201 +pro': (64, 201, 337, 256, 2, 1, 1, True, 1),
    'W5 conv5 2x2 64->128 @203 +pro': (64, 203, 339, 128, 2, 1, 0, True, 1),
    'W7 conv7 F(6,2) 49 x 256->256': (256, 1, 31008, 256, 1, 1, 0, False, 49),
    'W6 conv6 F(6,2) 49 x 128->256': (128, 1, 31008, 256, 1, 1, 0, False, 49),
    'W2 conv2 F(6,2) 49 x 256->64': (256, 1, 31008, 64, 1, 1, 0, False, 49),
    'W0 conv0 2x2 64->64 @200': (64, 200, 336, 64, 2, 1, 1, False, 1),
    'Wd 2x2 256->256 @201 direct': (256, 201, 337, 256, 2, 1, 0, True, 1),
    'Ws stem 7x7s2 4->64 @800': (4, 800, 1344, 64, 7, 2, 3, False, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = 'cuda:0'
    only = [s for s in args.only.split(',') if s]
    tot = {'0': [0.0, 0.0], '1': [0.0, 0.0]}
    for name, (cin, h, w, cout, k, s, p, pro, groups) in SHAPES.items():
        if only and not any(name.startswith(o) for o in only):
            continue
        line, outs = '%-34s' % name, {}
        for mode in ('0', '1'):
            torch.manual_seed(0)
            os.environ['HND_WGRAD_RING'] = mode
            os.environ['HND_DEBUG_PICKER'] = 'wgrad_ring_taps'      # the tap form too
            if groups == 1:
                n = args.batch
                oh, ow = ops.conv_out_size(h, k, s, p), ops.conv_out_size(w, k, s, p)
                x = torch.randn(n, h, w, cin, device=dev)
                dy = torch.randn(n, oh, ow, ops.round_up(cout, 4), device=dev)
                dw = torch.empty(cout, min(cin, 3) if cin == 4 else cin, k, k, device=dev)
                ps = (torch.rand(cin, device=dev) + 0.5) if pro else None
                pb = torch.randn(cin, device=dev) if pro else None
                l = ops.conv_wgrad(x, dy, dw, k, s, p, pro_scale=ps, pro_shift=pb, pro_relu=pro)
            else:
                tiles = w
                x = torch.randn(groups, tiles, cin, device=dev)
                dy = torch.randn(groups, tiles, cout, device=dev)
                dw = torch.empty(groups, cout, cin, device=dev)
                d = ops.WgradDesc()
                d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
                d.n, d.h, d.w_, d.cin, d.cin_real, d.oh, d.ow, d.cout, d.ldy = 1, 1, tiles, cin, cin, 1, tiles, cout, cout
                d.kh, d.kw, d.stride, d.pad, d.splitk, d.groups = 1, 1, 1, 0, 0, groups
                d.x_group_stride, d.dy_group_stride, d.dw_group_stride = tiles * cin, tiles * cout, cout * cin
                slabs = torch.empty((ops.wgrad_workspace_of(d) + 3) // 4, device=dev)
                d.slabs = slabs.data_ptr()
                l = ops.WgradLaunch(d, (x, dy, dw, slabs), 2 * groups * tiles * cout * cin)
            for _ in range(2):
                l.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                l.run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            outs[mode] = dw.clone()
            tot[mode][0] += ms
            tot[mode][1] += l.flops
            line += '  %-11s %7.3f ms %6.1f TF' % (l.variant, ms, l.flops / ms / 1e9)
        rel = float((outs['0'] - outs['1']).norm() / outs['0'].norm())
        print(line + '   rel diff %.1e' % rel, flush=True)
    os.environ.pop('HND_WGRAD_RING', None)
    os.environ.pop('HND_DEBUG_PICKER', None)
    for mode, (ms, fl) in tot.items():
        if ms:
            print('TOTAL HND_WGRAD_RING=%s %8.3f ms  %7.1f TFLOP/s' % (mode, ms, fl / ms / 1e9))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""A/B of the B-streamed persistent GEMM (csrc/conv_bstream.hip) against the tiled implicit-GEMM kernel (and, where it
applies, the B-resident kernels) on the long-K launches of the GHND step (batch 16): K = 1024 / 2048 1x1 convs and
data gradients of layer3 / layer4 / FPN laterals, the stride-2 3x3 convs over taps.  Outputs must be identical bits.
usage: python tools/bench_bstream.py [--iters 10] [--batch 16] [--only substr]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

# name: (cin, h, w, cout, k, stride, pad, residual, mask, prologue, groups)
SHAPES = {
    '1x1_1024-256@50 (layer3.conv1)': (1024, 50, 84, 256, 1, 1, 0, False, False, False, 1),
    '1x1_1024-256@50 pro+mask (l3.conv3.dgrad)': (1024, 50, 84, 256, 1, 1, 0, False, True, True, 1),
    '1x1_1024-512@50 (layer4.0.conv1)': (1024, 50, 84, 512, 1, 1, 0, False, False, False, 1),
    '1x1s2_1024-2048@50 (layer4.0.down)': (1024, 50, 84, 2048, 1, 2, 0, False, False, False, 1),
    '1x1_2048-512@25 (layer4.conv1)': (2048, 25, 42, 512, 1, 1, 0, False, False, False, 1),
    '1x1_2048-512@25 pro+mask (l4.conv3.dgrad)': (2048, 25, 42, 512, 1, 1, 0, False, True, True, 1),
    '1x1_2048-1024@25 pro (l4.0.down.dgrad)': (2048, 25, 42, 1024, 1, 1, 0, False, False, True, 1),
    '1x1_1024-256@50+up (fpn.inner2)': (1024, 50, 84, 256, 1, 1, 0, True, False, False, 1),
    '1x1_2048-256@25 (fpn.inner3)': (2048, 25, 42, 256, 1, 1, 0, False, False, False, 1),
    '3x3s2_128-128@200 (layer2.0.conv2)': (128, 200, 336, 128, 3, 2, 1, False, False, False, 1),
    '3x3s2_256-256@100 (layer3.0.conv2)': (256, 100, 168, 256, 3, 2, 1, False, False, False, 1),
    '3x3s2_512-512@50 (layer4.0.conv2)': (512, 50, 84, 512, 3, 2, 1, False, False, False, 1),
    '1x1_256-1024@50+res (layer3.conv3)': (256, 50, 84, 1024, 1, 1, 0, True, False, False, 1),
    '1x1_128-512@100+res (layer2.conv3)': (128, 100, 168, 512, 1, 1, 0, True, False, False, 1),
    '1x1_512-256@100 (layer3.0.conv1)': (512, 100, 168, 256, 1, 1, 0, False, False, False, 1),
    '1x1_512-128@100 (layer2.conv1)': (512, 100, 168, 128, 1, 1, 0, False, False, False, 1),
    '1x1_512-2048@25+res (layer4.conv3)': (512, 25, 42, 2048, 1, 1, 0, True, False, False, 1),
    '1x1_512-128@100 mask (l2.conv3.dgrad)': (512, 100, 168, 128, 1, 1, 0, False, True, False, 1),
    '1x1_512-2048@25 mask (l4.conv1.dgrad)': (512, 25, 42, 2048, 1, 1, 0, False, True, False, 1),
    '1x1_512-256@100+res (fpn.inner1)': (512, 100, 168, 256, 1, 1, 0, True, False, False, 1),
    '1x1_256-1024@50 mask (l3.conv1.dgrad)': (256, 50, 84, 1024, 1, 1, 0, False, True, False, 1),
    '1x1_256-512@100 mask+res (l3.0.conv1.dgrad)': (256, 100, 168, 512, 1, 1, 0, True, True, False, 1),
    'wino64_256-256@50 (layer3.conv2)': (256, 9, 14, 256, 1, 1, 0, False, False, False, 64),
    'wino64_512-512@25 (layer4.conv2)': (512, 5, 7, 512, 1, 1, 0, False, False, False, 64),
    'wino64_256-256@200 (fpn.layer0)': (256, 34, 56, 256, 1, 1, 0, False, False, False, 64),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = 'cuda:0'
    modes = (('tiled', '0', '0'), ('default', None, '0'), ('bstream', '0', 'all'))      # name, HND_BRES, HND_BSTREAM
    tot = {m[0]: [0.0, 0.0] for m in modes}
    for name, (cin, h, w, cout, k, s, p, res, msk, pro, groups) in SHAPES.items():
        if args.only and args.only not in name:
            continue
        n = args.batch
        torch.manual_seed(0)
        if groups > 1:          # rows = groups * tiles_pad, one packed weight matrix per group
            tiles_pad = (n * h * w + 255) // 256 * 256
            x = torch.randn(1, 1, groups * tiles_pad, cin, device=dev)
            y = torch.empty(1, 1, groups * tiles_pad, cout, device=dev)
            pks = [ops.pack_weights(torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5) for _ in range(groups)]
            pk = ops.PackedWeight.__new__(ops.PackedWeight)
            pk.buf = torch.cat([q.buf for q in pks])
            pk.kdim, pk.rows, pk.chan_pad, pk.chan_real = pks[0].kdim, cout, cin, cin
            stride = pks[0].buf.numel()
        else:
            x = torch.randn(n, h, w, cin, device=dev)
            oh, ow = ops.conv_out_size(h, k, s, p), ops.conv_out_size(w, k, s, p)
            y = torch.empty(n, oh, ow, cout, device=dev)
            pk = ops.pack_weights(torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5)
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        r = torch.randn_like(y) if res else None
        mk = (torch.rand_like(y) - 0.3).clamp_min(0) if msk else None
        ps = (torch.rand(cin, device=dev) + 0.5) if pro else None
        pb = torch.zeros(cin, device=dev) if pro else None
        outs, line = {}, '%-42s' % name
        for mode, e1, e2 in modes:
            if e1 is None:
                os.environ.pop('HND_BRES', None)
            else:
                os.environ['HND_BRES'] = e1
            os.environ['HND_BSTREAM'] = '1' if e2 == 'all' else e2      # 'all' = HND_DEBUG_PICKER=bstream_all
            os.environ['HND_DEBUG_PICKER'] = 'bstream_all' if e2 == 'all' else ''
            if groups > 1:
                l = ops.conv_desc(x, pk, y, kh=1, kw=1, oh=1, ow=groups * tiles_pad, sh=1, dh=1, bh=0, sw=1, dw=1, bw=0,
                                  cout=cout)
                l.desc.w_group_rows, l.desc.w_group_stride = tiles_pad, stride
                l.flops = 2 * groups * tiles_pad * cout * cin
            else:
                l = ops.conv_forward(x, pk, y, k, s, p, epi_scale=sc, epi_shift=sh, res1=r, mask=mk, relu=not msk,
                                     pro_scale=ps, pro_shift=pb, pro_relu=False)
            l.refresh_variant()
            y.fill_(float('nan'))
            for _ in range(2):
                l.run()
            e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                l.run()
            e1_.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1_) / args.iters
            outs[mode] = y.clone()
            tot[mode][0] += ms
            tot[mode][1] += l.flops
            line += '  %-13s %6.3f ms %5.1f TF' % (l.variant, ms, l.flops / ms / 1e9)
        same = all(torch.equal(outs['tiled'], outs[m]) for m in ('default', 'bstream')) and \
            not bool(torch.isnan(outs['bstream']).any())
        print(line + ('   bits equal' if same else '   *** DIFFERENT (max |d| %.3e)'
                                                   % float((outs['tiled'] - outs['bstream']).abs().max())), flush=True)
    os.environ.pop('HND_BRES', None)
    os.environ.pop('HND_BSTREAM', None)
    os.environ.pop('HND_DEBUG_PICKER', None)
    for mode, (ms, fl) in tot.items():
        if ms:
            print('TOTAL %-8s %8.3f ms  %7.1f TFLOP/s' % (mode, ms, fl / ms / 1e9))


if __name__ == '__main__':
    main()

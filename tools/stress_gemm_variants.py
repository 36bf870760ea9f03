#!/usr/bin/env python
"""Randomised bit-identity stress of the hnd_conv2d_igemm kernel variants (--family native | emulated, round 6): for random geometries / epilogues the tiled
kernel (HND_BRES=0 HND_BSTREAM=0), the default dispatch and the B-streamed kernel forced on (HND_DEBUG_PICKER=bstream_all, with and
without its work-balancing relay) must produce IDENTICAL bits, and launching twice on one workspace must too.
usage: python tools/stress_gemm_variants.py [--cases 200] [--seed 0]"""
import argparse
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--family', default='native', choices=['native', 'emulated'],
                    help="native: launches built with the bf16x3 emulation off (tiled == B-resident == B-streamed bits); "
                         "emulated: built with it forced on (the B-streamed emulation kernel with and without its relay, "
                         "launched twice: the same bits)")
    args = ap.parse_args()
    ops.BX3_MODE[0] = 'off' if args.family == 'native' else 'force'
    rnd = random.Random(args.seed)
    dev = 'cuda:0'
    counts, bad = {}, 0
    # grouped weights (the Winograd component GEMMs): rows = groups * tiles_pad, one packed matrix per group
    for case in range(args.cases // 5):
        cin, cout = rnd.choice([256, 384, 512]), rnd.choice([64, 128, 256, 512])
        groups = rnd.choice([4, 9, 16, 25, 36, 49, 64])
        tiles_pad = rnd.choice([256, 512, 1024, 2304, 4096, 30464 // 256 * 256])
        g = torch.Generator().manual_seed(5000 + case)
        x = torch.randn(1, 1, groups * tiles_pad, cin, generator=g).to(dev)
        y = torch.empty(1, 1, groups * tiles_pad, cout, device=dev)
        pks = [ops.pack_weights((torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(dev)) for _ in range(groups)]
        pk = ops.PackedWeight.__new__(ops.PackedWeight)
        pk.buf = torch.cat([q.buf for q in pks])
        pk.kdim, pk.rows, pk.chan_pad, pk.chan_real = pks[0].kdim, cout, cin, cin
        outs, variants = {}, {}
        for mode, env in (('tiled', {'HND_BRES': '0', 'HND_BSTREAM': '0'}), ('default', {}),
                          ('bstream', {'HND_BRES': '0', 'HND_DEBUG_PICKER': 'bstream_all'})):
            for key in ('HND_BRES', 'HND_BSTREAM', 'HND_DEBUG_PICKER'):
                os.environ.pop(key, None)
            os.environ.update(env)
            l = ops.conv_desc(x, pk, y, kh=1, kw=1, oh=1, ow=groups * tiles_pad, sh=1, dh=1, bh=0, sw=1, dw=1, bw=0,
                              cout=cout)
            l.desc.w_group_rows, l.desc.w_group_stride = tiles_pad, pks[0].buf.numel()
            l.refresh_variant()
            y.fill_(float('nan'))
            l.run()
            ops.sync_check()
            outs[mode], variants[mode] = y.clone(), l.variant + ('+relay' if l.relay is not None else '')
        ok = all(torch.equal(outs['tiled'], outs[m]) for m in outs) and not bool(torch.isnan(outs['tiled']).any())
        for v in variants.values():
            counts['grouped ' + v] = counts.get('grouped ' + v, 0) + 1
        if not ok:
            bad += 1
            print('grouped case %d MISMATCH cin=%d cout=%d groups=%d tiles_pad=%d %s' % (case, cin, cout, groups,
                                                                                         tiles_pad, variants), flush=True)
        del x, y, outs
    for case in range(args.cases):
        taps = rnd.random() < 0.35
        if taps:
            cin = rnd.choice([128, 256, 384, 512])
            k, s, p = rnd.choice([(3, 2, 1), (3, 1, 1), (2, 1, 0), (2, 1, 1), (3, 2, 0)])
        else:
            cin = rnd.choice([256, 384, 512, 640, 1024, 1536, 2048])
            k, s, p = 1, rnd.choice([1, 1, 1, 2]), 0
        cout = rnd.choice([64, 128, 192, 256, 320, 512, 1024, 2048])
        budget = rnd.choice([3e4, 1e5, 3e5, 1e6]) * rnd.uniform(0.5, 1.5)      # output pixels
        n = rnd.choice([1, 2, 3, 8, 16])
        hw = max(8.0, (budget * s * s / n) ** 0.5)
        h, w = int(hw * rnd.uniform(0.6, 1.4)) + 1, int(hw * rnd.uniform(0.6, 1.4)) + 1
        if n * h * w * cin * 4 > 1.5e9 or (n * h * w // (s * s)) * cout * 4 > 1.5e9:
            continue
        res, msk, pro = rnd.random() < 0.3, rnd.random() < 0.3, rnd.random() < 0.3
        pro_relu = pro and rnd.random() < 0.5
        g = torch.Generator().manual_seed(1000 + case)
        x = torch.randn(n, h, w, cin, generator=g).to(dev)
        oh, ow = ops.conv_out_size(h, k, s, p), ops.conv_out_size(w, k, s, p)
        if oh < 1 or ow < 1:
            continue
        y = torch.empty(n, oh, ow, cout, device=dev)
        pk = ops.pack_weights((torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev))
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
        r = torch.randn(y.shape, generator=g).to(dev) if res else None
        mk = (torch.rand(y.shape, generator=g) - 0.3).clamp_min(0).to(dev) if msk else None
        ps = (torch.rand(cin, generator=g) + 0.5).to(dev) if pro else None
        pb = torch.randn(cin, generator=g).to(dev) if pro else None
        outs, variants = {}, {}
        for mode, env in (('tiled', {'HND_BRES': '0', 'HND_BSTREAM': '0'}), ('default', {}),
                          ('bstream', {'HND_BRES': '0', 'HND_DEBUG_PICKER': 'bstream_all'}),
                          ('bstream_rr', {'HND_BRES': '0', 'HND_DEBUG_PICKER': 'bstream_all'})):      # round-robin tiles: no workspace
            for key in ('HND_BRES', 'HND_BSTREAM', 'HND_DEBUG_PICKER'):
                os.environ.pop(key, None)
            os.environ.update(env)
            l = ops.conv_forward(x, pk, y, k, s, p, epi_scale=sc, epi_shift=sh, res1=r, mask=mk, relu=not msk,
                                 pro_scale=ps, pro_shift=pb, pro_relu=pro_relu)
            if mode == 'bstream_rr':                    # drop the work-balancing workspace: tiles go round-robin
                l.desc.relay_ws, l.relay = None, None
            reps = 2 if mode == 'bstream' else 1
            for _ in range(reps):
                y.fill_(float('nan'))
                l.run()
            ops.sync_check()
            outs[mode], variants[mode] = y.clone(), l.variant + ('+relay' if l.relay is not None else '')
            if l.relay is not None and int(l.relay.view(torch.int32)[256 * 16384 + 257]) != 0:
                print('case %d: relay ticket not back at zero' % case)
                bad += 1
        ok = all(torch.equal(outs['tiled'], outs[m]) for m in outs) and not bool(torch.isnan(outs['tiled']).any())
        for v in variants.values():
            counts[v] = counts.get(v, 0) + 1
        if not ok:
            bad += 1
            print('case %d MISMATCH cin=%d cout=%d k=%d s=%d p=%d n=%d h=%d w=%d res=%s mask=%s pro=%s %s' % (
                case, cin, cout, k, s, p, n, h, w, res, msk, pro, variants), flush=True)
        elif case % 20 == 0:
            print('case %d ok  M=%d N=%d K=%d  %s' % (case, n * oh * ow, cout, k * k * cin, variants), flush=True)
        del x, y, r, mk, outs
    for key in ('HND_BRES', 'HND_BSTREAM', 'HND_DEBUG_PICKER'):
        os.environ.pop(key, None)
    print('variants exercised:', counts)
    print('%d mismatch(es) in %d cases' % (bad, args.cases))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())

#!/usr/bin/env python
"""Per-shape A/B of the B-streamed bf16x3 emulation kernel (csrc/conv_bxs.hip) against the native fp32 picker's choice and,
where it applies, the B-resident emulation kernel, on the step's launches at batch 16 (3x800x1333): ms per launch (HIP
events over 20 launches after 5 warm-ups; operands re-used, so L2 / MALL-warm like inside the step) and TF-equivalent.

    python tools/bench_bxs.py > gpurun_out/bxs_shapes.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

DEV = torch.device('cuda:0')
# name, cin, cout, n, h, w, k, stride, pad, extras
FWD = [
    ('layer2.0.conv2 3x3 s2 128->128', 128, 128, 16, 200, 336, 3, 2, 1, 'epi'),
    ('layer3.0.conv2 3x3 s2 256->256', 256, 256, 16, 100, 168, 3, 2, 1, 'epi'),
    ('layer4.0.conv2 3x3 s2 512->512', 512, 512, 16, 50, 84, 3, 2, 1, 'epi'),
    ('layer4.x.conv1 1x1 2048->512', 2048, 512, 16, 25, 42, 1, 1, 0, 'epi'),
    ('fpn.inner3 1x1 2048->256', 2048, 256, 16, 25, 42, 1, 1, 0, 'epi'),
    ('layer4.0.conv1 1x1 1024->512 @50x84', 1024, 512, 16, 50, 84, 1, 1, 0, 'epi'),
    ('layer3.x.conv1 1x1 1024->256 @50x84', 1024, 256, 16, 50, 84, 1, 1, 0, 'epi'),
    ('head conv1 2x2 64->256 p1 BN-on-load + stats', 64, 256, 16, 201, 337, 2, 1, 1, 'pro+stats'),
    ('head conv0 2x2 64->64 p1 stats', 64, 64, 16, 200, 336, 2, 1, 1, 'stats'),
    ('head conv5 2x2 64->128 p0 BN-on-load + stats', 64, 128, 16, 203, 339, 2, 1, 0, 'pro+stats'),
    ('head conv2.dgrad-like 2x2 256->64 p1 + bwd sums', 256, 64, 16, 201, 337, 2, 1, 1, 'bwd'),
    ('head conv5.dgrad-like 2x2 128->64 p1', 128, 64, 16, 202, 338, 2, 1, 1, 'plain'),
    ('layer2.0.conv1.dgrad-like 1x1 128->256 mask+res @200x336', 128, 256, 16, 200, 336, 1, 1, 0, 'mask+res'),
    ('layer3.0.conv1.dgrad-like 1x1 256->512 mask+res @100x168', 256, 512, 16, 100, 168, 1, 1, 0, 'mask+res'),
    ('layer4.x.conv3.dgrad-like 1x1 2048->512 mask @25x42', 2048, 512, 16, 25, 42, 1, 1, 0, 'mask'),
]
DGRAD = [  # name, cin (of the conv), cout, n, h, w, k, stride, pad, accumulate
    ('layer2.0.conv2.dgrad 3x3 s2 128<-128', 128, 128, 16, 200, 336, 3, 2, 1, False),
    ('layer3.0.conv2.dgrad 3x3 s2 256<-256', 256, 256, 16, 100, 168, 3, 2, 1, False),
    ('layer4.0.conv2.dgrad 3x3 s2 512<-512', 512, 512, 16, 50, 84, 3, 2, 1, False),
    ('layer2.0.downsample.dgrad 1x1 s2 256<-512', 256, 512, 16, 200, 336, 1, 2, 0, True),
    ('layer3.0.downsample.dgrad 1x1 s2 512<-1024', 512, 1024, 16, 100, 168, 1, 2, 0, True),
    ('layer4.0.downsample.dgrad 1x1 s2 1024<-2048', 1024, 2048, 16, 50, 84, 1, 2, 0, True),
]


def timed(launches, reps=20):
    for _ in range(5):
        for l in launches:
            l.run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        for l in launches:
            l.run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    g = torch.Generator().manual_seed(0)
    print('%-58s %-16s %9s %7s   %-9s %9s %7s %6s' % ('shape', 'native kernel', 'ms', 'TF', 'emulated', 'ms', 'TF-eq', 'x'))
    for name, cin, cout, n, h, w, k, stride, pad, extra in FWD:
        x = torch.randn(n, h, w, cin, generator=g).to(DEV)
        wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(DEV)
        oh, ow = ops.conv_out_size(h, k, stride, pad), ops.conv_out_size(w, k, stride, pad)
        m = n * oh * ow
        pk = ops.pack_weights(wt)
        y = torch.empty(n, oh, ow, cout, device=DEV)
        kw = {}
        if 'epi' in extra:
            kw.update(epi_scale=torch.rand(cout, generator=g).to(DEV) + 0.5, epi_shift=torch.randn(cout, generator=g).to(DEV), relu=True)
        if 'pro' in extra:
            kw.update(pro_scale=torch.rand(cin, generator=g).to(DEV) + 0.5, pro_shift=torch.randn(cin, generator=g).to(DEV), pro_relu=True)
        if 'stats' in extra or 'bwd' in extra:
            kw['stats'] = torch.empty(ops.stats_tiles(m), 2, cout, device=DEV)
        if 'bwd' in extra:
            kw['bwd_stats'] = (torch.randn(n, oh, ow, cout, generator=g).to(DEV), torch.rand(cout).to(DEV) + 0.5, torch.randn(cout).to(DEV),
                               torch.randn(cout).to(DEV), torch.rand(cout).to(DEV) + 0.5, True)
        if 'mask' in extra:
            kw['mask'] = torch.randn(n, oh, ow, cout, generator=g).to(DEV)
        if 'res' in extra:
            kw['res1'] = torch.randn(n, oh, ow, cout, generator=g).to(DEV)
        gf = 2.0 * m * cin * k * k * cout / 1e9
        with ops.emulation('off'):
            l0 = ops.conv_forward(x, pk, y, k, stride, pad, **kw)
        t0 = timed([l0])
        rows = []
        with ops.emulation('force'):
            l1 = ops.conv_forward(x, pk, y, k, stride, pad, **kw)
            rows.append((l1.variant, timed([l1])))
            if l1.variant.startswith('bx3'):
                keep, pk.bx3, pk.used3 = pk.bx3, None, None
                l2 = ops.conv_forward(x, pk, y, k, stride, pad, **kw)
                rows.append((l2.variant, timed([l2])))
                pk.bx3 = keep
        for var, t1 in rows:
            print('%-58s %-16s %9.3f %7.1f   %-9s %9.3f %7.1f %6.2f' % (name, l0.variant, t0, gf / t0, var, t1, gf / t1, t0 / t1))
        del x, y, kw
    for name, cin, cout, n, h, w, k, stride, pad, acc in DGRAD:
        oh, ow = ops.conv_out_size(h, k, stride, pad), ops.conv_out_size(w, k, stride, pad)
        dy = torch.randn(n, oh, ow, cout, generator=g).to(DEV)
        wt = (torch.randn(cout, cin, k, k, generator=g) / (cout * k * k) ** 0.5).to(DEV)
        act = torch.randn(n, h, w, cin, generator=g).to(DEV)
        bits = ops.mask_nibbles_like(act)
        ops.relu_mask_nibbles(act, bits)
        dx = torch.zeros(n, h, w, cin, device=DEV)
        gf = 2.0 * n * oh * ow * cin * k * k * cout / 1e9
        res = []
        for mode in ('off', 'force'):
            with ops.emulation(mode):
                ls, _ = ops.conv_dgrad(dy, wt, dx, k, stride, pad, accumulate=acc,
                                       **({'mask_bits': bits} if acc else {'mask': act}))
            res.append(('+'.join(sorted(set(l.variant for l in ls))), len(ls), timed(ls)))
        (v0, n0, t0), (v1, n1, t1) = res
        print('%-58s %-16s %9.3f %7.1f   %-9s %9.3f %7.1f %6.2f   (%d launches)' % (name, v0[:16], t0, gf / t0, v1[:9], t1, gf / t1, t0 / t1, n1))


if __name__ == '__main__':
    main()

"""GPU: every C-ABI kernel against a plain PyTorch fp32 (CPU) reference of the same op.

Tolerances are written per test; convs are fp32 MFMA (exact fp32 products, different summation
order than oneDNN) so 1e-4 relative-to-max is generous; the north-star bar is 1e-3.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need a device'
    from hnd_ghnd_object_detectors_amd import ops as o
    assert 'gfx950' in o.device_arch(), o.device_arch()
    return o


@pytest.fixture(autouse=True)
def native_family(ops):
    """Every test of this file holds the NATIVE fp32-MFMA kernels (tiled, B-resident, B-streamed, ring, vector-ALU) to torch
    and to one another bit for bit: launches are built with the bf16x3 emulation off.  The emulated family has the same
    kind of tests -- fp64 beside the native kernel, identical bits across batch / team / tail inside the family, the policy
    that picks it -- in tests/test_bx3_gpu.py."""
    with ops.emulation('off'):
        yield


def nhwc(t_nchw, cpad=None):
    """NCHW cpu tensor -> NHWC device tensor (channels zero-padded to cpad)."""
    t = t_nchw.permute(0, 2, 3, 1).contiguous()
    if cpad is not None and cpad != t.shape[-1]:
        t = F.pad(t, (0, cpad - t.shape[-1]))
    return t.to(DEV).contiguous()


def nchw(t_nhwc, c=None):
    t = t_nhwc.cpu()
    if c is not None:
        t = t[..., :c]
    return t.permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def gen(seed):
    return torch.Generator().manual_seed(seed)


CONV_CASES = [
    # n, cin, h, w, cout, k, stride, pad
    (2, 64, 13, 17, 64, 2, 1, 1),      # encoder.0-like (padded 2x2, odd sizes)
    (2, 64, 14, 18, 256, 2, 1, 1),     # encoder.2
    (1, 256, 9, 11, 64, 2, 1, 1),      # encoder.5
    (2, 64, 10, 12, 3, 2, 1, 1),       # encoder.7: cout 3 stored as 4
    (2, 3, 12, 14, 64, 2, 1, 0),       # decoder.2: cin 3 stored as 4
    (2, 128, 11, 13, 256, 2, 1, 0),    # decoder.7
    (1, 256, 10, 12, 256, 2, 1, 0),    # decoder.9
    (2, 3, 37, 45, 64, 7, 2, 3),       # stem
    (2, 64, 16, 20, 64, 3, 1, 1),      # layer1 3x3
    (2, 128, 17, 21, 128, 3, 2, 1),    # stride-2 3x3 (odd input)
    (2, 256, 8, 10, 512, 1, 2, 0),     # stride-2 1x1 downsample
    (1, 512, 7, 9, 128, 1, 1, 0),      # 1x1
    (3, 64, 33, 41, 64, 1, 1, 0),      # M not a multiple of 128
    (2, 64, 64, 64, 64, 4, 2, 0),      # neural filter conv 4x4 stride 2
    (3, 64, 31, 31, 32, 3, 2, 0),      # neural filter conv 3x3 stride 2, unpadded
    (2, 32, 15, 15, 16, 2, 1, 0),      # neural filter conv 2x2, cout 16 stored as 32
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward_plain(ops, case):
    n, cin, h, w, cout, k, s, p = case
    g = gen(sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    ref = F.conv2d(x, wt, None, s, p)
    xd = nhwc(x, ops.chan_pad_of(cin))
    wd = wt.to(DEV).contiguous()
    pw = ops.pack_weights(wd, chan_pad=ops.chan_pad_of(cin))
    y = torch.full((n, ref.shape[2], ref.shape[3], ops.chan_pad_of(cout)), float('nan'), device=DEV)
    ops.conv_forward(xd, pw, y, k, s, p).run()
    ops.sync_check()
    got = nchw(y, cout)
    assert relerr(got, ref) < 1e-4, relerr(got, ref)
    if ops.chan_pad_of(cout) != cout:
        assert float(y[..., cout:].abs().max()) == 0.0


def test_conv_forward_fused_prologue_epilogue_stats(ops):
    """train-BN on load (+ReLU, zero padding AFTER normalisation), FBN epilogue, residuals, mask, ReLU, stats."""
    g = gen(7)
    n, cin, h, w, cout, k, s, p = 2, 64, 15, 19, 128, 2, 1, 1
    x = torch.randn(n, cin, h, w, generator=g)
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    es, eb = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    a = F.relu(x * ps[None, :, None, None] + pb[None, :, None, None])
    conv = F.conv2d(a, wt, None, s, p)
    r1, r2 = torch.randn(conv.shape, generator=g), torch.randn(conv.shape, generator=g)
    mk = torch.randn(conv.shape, generator=g)
    pre = conv * es[None, :, None, None] + eb[None, :, None, None] + r1 + r2
    ref = F.relu(torch.where(mk > 0, pre, torch.zeros_like(pre)))
    xd, wd = nhwc(x), wt.to(DEV)
    pw = ops.pack_weights(wd)
    y = torch.empty(n, conv.shape[2], conv.shape[3], cout, device=DEV)
    m = n * conv.shape[2] * conv.shape[3]
    stats = torch.zeros(ops.stats_tiles(m), 2, cout, device=DEV)
    ops.conv_forward(xd, pw, y, k, s, p, pro_scale=ps.to(DEV), pro_shift=pb.to(DEV), pro_relu=True,
                     epi_scale=es.to(DEV), epi_shift=eb.to(DEV), res1=nhwc(r1), res2=nhwc(r2), mask=nhwc(mk),
                     relu=True, stats=stats).run()
    ops.sync_check()
    assert relerr(nchw(y), ref) < 1e-4
    st = stats.cpu().double().sum(0)
    assert relerr(st[0], ref.double().sum((0, 2, 3))) < 1e-4
    assert relerr(st[1], (ref.double() ** 2).sum((0, 2, 3))) < 1e-4


def test_conv_forward_upsampled_residual(ops):
    """FPN top-down: lateral 1x1 conv + bias + nearest-upsampled coarser map."""
    g = gen(8)
    n, cin, h, w, cout = 2, 512, 10, 14, 256
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    coarse = torch.randn(n, cout, 5, 7, generator=g)
    ref = F.conv2d(x, wt, b) + F.interpolate(coarse, size=(h, w), mode='nearest')
    y = torch.empty(n, h, w, cout, device=DEV)
    ops.conv_forward(nhwc(x), ops.pack_weights(wt.to(DEV)), y, 1, 1, 0, epi_shift=b.to(DEV), res1=nhwc(coarse),
                     res1_up=True).run()
    ops.sync_check()
    assert relerr(nchw(y), ref) < 1e-4


DGRAD_CASES = [
    (2, 64, 13, 17, 64, 2, 1, 1),
    (2, 64, 10, 12, 3, 2, 1, 1),       # dy has 3 (4) channels
    (2, 3, 12, 14, 64, 2, 1, 0),       # dx has 3 (4) channels
    (1, 256, 10, 12, 256, 2, 1, 0),
    (2, 64, 16, 20, 64, 3, 1, 1),
    (2, 128, 17, 21, 128, 3, 2, 1),    # stride 2, odd input
    (2, 128, 16, 20, 128, 3, 2, 1),    # stride 2, even input
    (1, 512, 7, 9, 128, 1, 1, 0),
    (3, 64, 31, 31, 32, 3, 2, 0),      # neural filter: stride 2 without padding
    (2, 64, 64, 64, 64, 4, 2, 0),      # even kernel, stride 2
    (2, 32, 15, 15, 16, 2, 1, 0),
]


@pytest.mark.parametrize('case', DGRAD_CASES)
def test_conv_dgrad(ops, case):
    n, cin, h, w, cout, k, s, p = case
    g = gen(1000 + sum(case))
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    out = F.conv2d(x, wt, None, s, p)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    dx = torch.full((n, h, w, ops.chan_pad_of(cin)), float('nan'), device=DEV)
    launches, _ = ops.conv_dgrad(nhwc(dy, ops.chan_pad_of(cout)), wt.to(DEV).contiguous(), dx, k, s, p)
    for l in launches:
        l.run()
    ops.sync_check()
    assert relerr(nchw(dx, cin), x.grad) < 1e-4


def test_conv_dgrad_1x1_stride2_accumulate_with_mask(ops):
    """downsample dgrad adds into an existing gradient at even pixels only, then the block mask applies."""
    g = gen(5)
    n, cin, h, w, cout = 2, 256, 9, 12, 512
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    sc = torch.rand(cout, generator=g) + 0.5
    out = F.conv2d(x, wt, None, 2, 0)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy * sc[None, :, None, None])
    base = torch.randn(n, cin, h, w, generator=g)
    mk = torch.randn(n, cin, h, w, generator=g)
    ref = torch.where(mk > 0, base + x.grad, torch.zeros_like(base))
    ref_odd = base          # pixels not reached by the stride-2 1x1 keep the previous value
    dx = nhwc(base)
    launches, _ = ops.conv_dgrad(nhwc(dy), wt.to(DEV).contiguous(), dx, 1, 2, 0, accumulate=True,
                                 pro_scale=sc.to(DEV), mask=nhwc(mk))
    assert len(launches) == 1
    launches[0].run()
    ops.sync_check()
    got = nchw(dx)
    assert relerr(got[:, :, ::2, ::2], ref[:, :, ::2, ::2]) < 1e-4
    assert torch.equal(got[:, :, 1::2, :], ref_odd[:, :, 1::2, :])


@pytest.mark.parametrize('cin,cout,k,stride,pad,h,w', [(256, 512, 1, 1, 0, 19, 23), (128, 128, 3, 2, 1, 20, 28),
                                                       (256, 512, 1, 2, 0, 18, 24)])
def test_dgrad_with_the_frozen_bn_scale_folded_into_the_weights(ops, cin, cout, k, stride, pad, h, w):
    """W^T (dy * s) as (W^T diag(s)) dy: the FrozenBatchNorm2d scale between a frozen conv and its output gradient is
    folded into the packed transposed weights (hnd_scale_packed_k) instead of a prologue on the launch.  Must equal the
    prologue form to fp32 rounding and torch's autograd to 1e-4; a new scale tensor re-packs into the same buffer."""
    from hnd_ghnd_object_detectors_amd import engine as E
    g = gen(41 + k + stride)
    n = 2
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    sc = torch.rand(cout, generator=g) + 0.5
    out = F.conv2d(x, wt, None, stride, pad)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy * sc[None, :, None, None])
    mk = torch.randn(n, cin, h, w, generator=g)
    ref = torch.where(mk > 0, x.grad, torch.zeros_like(x.grad))
    dyd, mkd, scd = nhwc(dy), nhwc(mk), sc.to(DEV)
    got = {}
    wc = E.WeightCache(wt.to(DEV).contiguous())
    for mode in ('pro', 'fold'):
        dx = torch.zeros(n, h, w, cin, device=DEV)
        kw = {'pro_scale': scd} if mode == 'pro' else {'fold_scale': scd}
        launches, packs = ops.conv_dgrad(dyd, wc, dx, k, stride, pad, accumulate=stride > 1, mask=mkd, **kw)
        for l in launches:
            assert (l.desc.pro_scale is None) == (mode == 'fold')
            l.run()
        ops.sync_check()
        got[mode] = nchw(dx)
    assert relerr(got['fold'], ref) < 1e-4 and relerr(got['pro'], ref) < 1e-4
    assert relerr(got['fold'], got['pro']) < 1e-6
    # a refreshed FrozenBN fold hands over a NEW scale tensor: same packed buffers, new contents
    sc2 = (sc * 2).to(DEV)
    dx = torch.zeros(n, h, w, cin, device=DEV)
    launches2, packs2 = ops.conv_dgrad(dyd, wc, dx, k, stride, pad, accumulate=stride > 1, mask=mkd, fold_scale=sc2)
    assert [q.buf.data_ptr() for q in packs2] == [q.buf.data_ptr() for q in packs]
    for l in launches2:
        l.run()
    ops.sync_check()
    assert relerr(nchw(dx), 2 * ref) < 1e-4


WGRAD_CASES = [
    (2, 64, 13, 17, 64, 2, 1, 1),
    (2, 64, 14, 18, 256, 2, 1, 1),
    (1, 256, 9, 11, 64, 2, 1, 1),
    (2, 64, 10, 12, 3, 2, 1, 1),
    (2, 3, 12, 14, 64, 2, 1, 0),
    (2, 128, 11, 13, 256, 2, 1, 0),
    (2, 3, 37, 45, 64, 7, 2, 3),
    (2, 64, 64, 64, 64, 4, 2, 0),      # neural filter
    (3, 64, 31, 31, 32, 3, 2, 0),
    (2, 32, 15, 15, 16, 2, 1, 0),
]


@pytest.mark.parametrize('case', WGRAD_CASES)
@pytest.mark.parametrize('splitk', [0, 1, 3])
def test_conv_wgrad(ops, case, splitk):
    n, cin, h, w, cout, k, s, p = case
    g = gen(2000 + sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    wt = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).requires_grad_(True)
    a = F.relu(x * ps[None, :, None, None] + pb[None, :, None, None])
    out = F.conv2d(a, wt, None, s, p)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    cp = ops.chan_pad_of(cin)
    psd = F.pad(ps, (0, cp - cin)).to(DEV)
    pbd = F.pad(pb, (0, cp - cin)).to(DEV)
    dw = torch.full((cout, cin, k, k), float('nan'), device=DEV)
    ops.conv_wgrad(nhwc(x, cp), nhwc(dy, ops.chan_pad_of(cout)), dw, k, s, p, pro_scale=psd, pro_shift=pbd,
                   pro_relu=True, splitk=splitk).run()
    ops.sync_check()
    assert relerr(dw.cpu(), wt.grad) < 1e-4


@pytest.mark.parametrize('cin,cout,pad,n,h,w', [(64, 3, 1, 2, 57, 83), (64, 3, 0, 3, 40, 64), (3, 64, 0, 2, 58, 84),
                                                (3, 64, 1, 1, 200, 336)])
def test_thin_wgrad_matches_the_mfma_kernel_and_torch(ops, cin, cout, pad, n, h, w, monkeypatch):
    """csrc/conv_wgrad.hip thin_wgrad_kernel: the two 3-channel weight gradients of the b3ch bottleneck on the vector ALU
    (the 64-channel tensor streams once, 64 sums per lane).  A different summation order than the MFMA split-K kernel:
    both must agree with torch's fp32 autograd to 1e-4 and with each other to 1e-5, and the kernel must be reproducible."""
    g = gen(77 + cin + pad)
    x = torch.randn(n, cin, h, w, generator=g)
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    wt = (torch.randn(cout, cin, 2, 2, generator=g) / math.sqrt(cin * 4)).requires_grad_(True)
    out = F.conv2d(F.relu(x * ps[None, :, None, None] + pb[None, :, None, None]), wt, None, 1, pad)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    cp = ops.chan_pad_of(cin)
    xd, dyd = nhwc(x, cp), nhwc(dy, ops.chan_pad_of(cout))
    psd, pbd = F.pad(ps, (0, cp - cin)).to(DEV), F.pad(pb, (0, cp - cin)).to(DEV)
    res = {}
    for mode in ('0', '1', '1b'):
        monkeypatch.setenv('HND_THIN_WGRAD', mode[0])
        dw = torch.full((cout, cin, 2, 2), float('nan'), device=DEV)
        l = ops.conv_wgrad(xd, dyd, dw, 2, 1, pad, pro_scale=psd, pro_shift=pbd, pro_relu=True)
        l.run()
        ops.sync_check()
        res[mode] = (dw.cpu(), l.variant)
    assert res['0'][1] == 'wgrad_m64' and res['1'][1] == 'thin_wgrad', (res['0'][1], res['1'][1])
    assert torch.equal(res['1'][0], res['1b'][0])
    assert relerr(res['1'][0], wt.grad) < 1e-4 and relerr(res['0'][0], wt.grad) < 1e-4
    assert relerr(res['1'][0], res['0'][0]) < 1e-5


@pytest.mark.parametrize('case', [
    # n, cin, h, w, cout, k, pad, groups      (groups > 1: the grouped Winograd-domain form, 1x1 over `h * w` tiles)
    (2, 64, 37, 53, 256, 2, 1, 1),            # head conv1: 256 x 256 tile, four taps of 64 channels, BN+ReLU prologue
    (3, 64, 40, 41, 128, 2, 0, 1),            # conv5: 128 x 256 tile
    (2, 128, 23, 31, 256, 2, 0, 1),           # 256 x 512 columns: two column tiles, halves inside one tap
    (1, 256, 1, 3001, 256, 1, 0, 7),          # conv7 in the Winograd domain: grouped, K = tiles, ragged against 32
    (1, 128, 1, 1777, 256, 1, 0, 5),          # conv6 in the Winograd domain: 256 x 128 tile
    (2, 64, 9, 9, 128, 2, 1, 1),              # fewer pixels than workgroups
])
def test_wgrad_ring_matches_the_staged_kernel_and_torch(ops, case, monkeypatch):
    """csrc/conv_wgrad_ring.hip: both operands straight from global memory through a counted register
    ring, accumulators resident.  A different summation order than the LDS-staged split-K kernel: both must agree with
    torch's fp32 autograd to 1e-4 and with each other to 2e-5, and the ring kernel must be bitwise reproducible."""
    n, cin, h, w, cout, k, pad, groups = case
    g = gen(500 + sum(case))
    res = {}
    if groups == 1:
        x = torch.randn(n, cin, h, w, generator=g)
        ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
        wt = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).requires_grad_(True)
        out = F.conv2d(F.relu(x * ps[None, :, None, None] + pb[None, :, None, None]), wt, None, 1, pad)
        dy = torch.randn(out.shape, generator=g)
        out.backward(dy)
        ref = wt.grad
        xd, dyd, psd, pbd = nhwc(x), nhwc(dy), ps.to(DEV), pb.to(DEV)
        for mode in ('0', '1', '1b'):
            monkeypatch.setenv('HND_WGRAD_RING', '1' if mode[0] == '1' else '0')
            monkeypatch.setenv('HND_DEBUG_PICKER', 'wgrad_ring_taps')                  # the tap form too
            dw = torch.full((cout, cin, k, k), float('nan'), device=DEV)
            l = ops.conv_wgrad(xd, dyd, dw, k, 1, pad, pro_scale=psd, pro_shift=pbd, pro_relu=True)
            l.run()
            ops.sync_check()
            res[mode] = (dw.cpu(), l.variant)
    else:
        tiles = w
        xs = torch.randn(groups, tiles, cin, generator=g)
        dys = torch.randn(groups, tiles, cout, generator=g)
        ref = torch.einsum('gtc,gtd->gdc', xs, dys)            # [groups, cout, cin]
        xd, dyd = xs.to(DEV).contiguous(), dys.to(DEV).contiguous()
        for mode in ('0', '1', '1b'):
            monkeypatch.setenv('HND_WGRAD_RING', mode[0])
            dw = torch.full((groups, cout, cin), float('nan'), device=DEV)
            d = ops.WgradDesc()
            d.x, d.dy, d.dw = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr()
            d.n, d.h, d.w_, d.cin, d.cin_real, d.oh, d.ow, d.cout, d.ldy = 1, 1, tiles, cin, cin, 1, tiles, cout, cout
            d.kh, d.kw, d.stride, d.pad, d.splitk, d.groups = 1, 1, 1, 0, 0, groups
            d.x_group_stride, d.dy_group_stride, d.dw_group_stride = tiles * cin, tiles * cout, cout * cin
            slabs = torch.empty((ops.wgrad_workspace_of(d) + 3) // 4, device=DEV)
            d.slabs = slabs.data_ptr()
            l = ops.WgradLaunch(d, (xd, dyd, dw, slabs), 2 * groups * tiles * cout * cin)
            l.run()
            ops.sync_check()
            res[mode] = (dw.cpu(), l.variant)
    assert res['0'][1].startswith('wgrad_m') and res['1'][1] == 'wgrad_ring', (res['0'][1], res['1'][1])
    assert torch.equal(res['1'][0], res['1b'][0])
    assert relerr(res['1'][0], ref) < 1e-4 and relerr(res['0'][0], ref) < 1e-4, (relerr(res['1'][0], ref),)
    assert relerr(res['1'][0], res['0'][0]) < 2e-5


def test_wgrad_is_deterministic(ops):
    g = gen(3)
    x = torch.randn(2, 64, 40, 50, generator=g)
    dy = torch.randn(2, 128, 39, 49, generator=g)
    xd, dyd = nhwc(x), nhwc(dy)
    outs = []
    for _ in range(2):
        dw = torch.empty(128, 64, 2, 2, device=DEV)
        ops.conv_wgrad(xd, dyd, dw, 2, 1, 0).run()
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1])


def test_pack_weights_layouts(ops):
    g = gen(4)
    wt = torch.randn(5, 3, 3, 3, generator=g)
    pw = ops.pack_weights(wt.to(DEV).contiguous())
    buf = pw.buf.cpu().view(64, pw.kdim)
    assert pw.kdim == 64 and pw.chan_pad == 4
    # packed row r holds output channel (r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3)  (hnd::chan_of_row: the four
    # 16-row MFMA tiles of a 64-row group are interleaved so a lane's four accumulator tiles are consecutive channels)
    chan = torch.tensor([(r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3) for r in range(64)])
    assert sorted(chan.tolist()) == list(range(64)) and chan[16] == 1 and chan[1] == 4
    flat = torch.zeros(64, 64)
    flat[:5, :36] = F.pad(wt.permute(0, 2, 3, 1), (0, 1)).reshape(5, 36)
    ref = flat[chan]
    assert torch.equal(buf, ref)
    pt = ops.pack_weights(wt.to(DEV).contiguous(), transposed=True, chan_pad=8, taps=(1, 2, 1, 0, 2, 2))
    buf = pt.buf.cpu().view(64, pt.kdim)
    flat = torch.zeros(64, pt.kdim)
    sub = wt[:, :, 1:2, 0::2]                                  # [o, i, 1, 2]
    flat[:3, :16] = F.pad(sub.permute(1, 2, 3, 0), (0, 3)).reshape(3, 16)
    assert torch.equal(buf, flat[chan])


def test_fbn_fold(ops):
    g = gen(6)
    w, b, m = torch.rand(64, generator=g), torch.randn(64, generator=g), torch.randn(64, generator=g)
    v = torch.rand(64, generator=g) + 0.5
    sc, sh = ops.fbn_fold(w.to(DEV), b.to(DEV), m.to(DEV), v.to(DEV))
    ops.sync_check()
    rs = w * v.rsqrt()
    assert relerr(sc.cpu(), rs) < 1e-6 and relerr(sh.cpu(), b - m * rs) < 1e-6


@pytest.mark.parametrize('shape,size,max_size', [((3, 60, 90), 64, 128), ((3, 56, 100), 64, 128),
                                                  ((3, 64, 96), 64, 128), ((3, 50, 70), 80, 100)])
def test_transform_matches_oracle(ops, shape, size, max_size):
    from oracle import hnd_oracle as O
    img = torch.rand(*shape, generator=gen(9))
    ref, sizes = O.transform_images([img], min_size=(size,), max_size=max_size)
    lo, hi = float(min(shape[1:])), float(max(shape[1:]))
    scale = size / lo
    if hi * scale > max_size:
        scale = max_size / hi
    oh, ow = ops.interp_out_size(shape[1], scale), ops.interp_out_size(shape[2], scale)
    assert (oh, ow) == sizes[0]
    dst = torch.full((1, ref.shape[2], ref.shape[3], 4), float('nan'), device=DEV)
    ops.transform_image(img.to(DEV), dst, 0, oh, ow, 1.0 / scale, 1.0 / scale, O.IMAGE_MEAN, O.IMAGE_STD)
    ops.sync_check()
    got = nchw(dst, 3)
    assert float((got - ref).abs().max()) < 2e-5
    assert float(dst[..., 3].abs().max()) == 0.0


@pytest.mark.parametrize('hwc', [True, False])
@pytest.mark.parametrize('flip', [False, True])
@pytest.mark.parametrize('shape,size,max_size', [((37, 61), 64, 128), ((120, 50), 64, 100), ((64, 64), 64, 128)])
def test_transform_u8_matches_oracle(ops, shape, size, max_size, hwc, flip):
    """decoded uint8 image: /255 (ToTensor) + flip + normalise + resize + pad in one kernel"""
    from oracle import hnd_oracle as O
    u8 = torch.randint(0, 256, (shape[0], shape[1], 3), generator=gen(19), dtype=torch.uint8)
    img = O.to_tensor_u8(u8)
    if flip:
        img = img.flip(-1)
    ref, sizes = O.transform_images([img], min_size=(size,), max_size=max_size)
    scale = size / float(min(shape))
    if float(max(shape)) * scale > max_size:
        scale = max_size / float(max(shape))
    oh, ow = sizes[0]
    dst = torch.full((1, ref.shape[2], ref.shape[3], 4), float('nan'), device=DEV)
    src = u8 if hwc else u8.permute(2, 0, 1).contiguous()
    ops.transform_image_u8(src.to(DEV), dst, 0, oh, ow, 1.0 / scale, 1.0 / scale, O.IMAGE_MEAN, O.IMAGE_STD, hwc, flip)
    ops.sync_check()
    assert float((nchw(dst, 3) - ref).abs().max()) < 2e-5
    assert float(dst[..., 3].abs().max()) == 0.0


@pytest.mark.parametrize('n,c,h,w', [(2, 64, 21, 30), (1, 64, 20, 31), (3, 64, 7, 9), (1, 8, 1, 2)])
def test_maxpool_fwd_bwd(ops, n, c, h, w):
    """3x3 stride-2 pad-1 pool and its backward (one thread per 2 x 2 input quad) for even / odd extents"""
    g = gen(10 + h)
    x = F.relu(torch.randn(n, c, h, w, generator=g)).requires_grad_(True)     # many exact-zero ties, as after ReLU
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd = nhwc(x.detach())
    yd = torch.empty(n, y.shape[2], y.shape[3], c, device=DEV)
    idx = torch.empty(n, y.shape[2], y.shape[3], c, dtype=torch.uint8, device=DEV)
    ops.maxpool_fwd(xd, yd, idx)
    assert torch.equal(nchw(yd), y.detach())
    sc = torch.rand(c, generator=g) + 0.5
    dx = torch.empty(n, h, w, c, device=DEV)
    ops.maxpool_bwd_relu_scale(nhwc(dy), idx, xd, sc.to(DEV), dx)
    ops.sync_check()
    ref = x.grad * (x.detach() > 0) * sc[None, :, None, None]
    assert relerr(nchw(dx), ref) < 1e-6


@pytest.mark.parametrize('c,relu', [(64, False), (256, True), (3, True), (128, False)])
def test_train_bn_forward_backward(ops, c, relu):
    """stats from the conv epilogue -> finalize -> apply; backward reduce/finalize/apply; vs torch BN autograd."""
    g = gen(11 + c)
    n, h, w = 2, 13, 17
    cs = ops.chan_pad_of(c)
    x = (torch.randn(n, c, h, w, generator=g) * 1.7 + 0.4).requires_grad_(True)
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(c, generator=g).requires_grad_(True)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    out = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    if relu:
        out = F.relu(out)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)

    xd = nhwc(x.detach(), cs)
    npix = n * h * w
    # partial statistics as the conv epilogue would emit them (per 128-pixel tile)
    nt = ops.stats_tiles(npix)
    flat = xd.view(npix, cs)
    part = torch.zeros(nt, 2, cs, device=DEV)
    for t in range(nt):
        blk = flat[t * 128:(t + 1) * 128]
        part[t, 0], part[t, 1] = blk.sum(0), (blk * blk).sum(0)
    dev = lambda t: t.detach().to(DEV).contiguous()
    gam, bet, rmd, rvd = dev(gamma), dev(beta), dev(rm), dev(rv)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    scale, shift, mean, rstd = (torch.empty(cs, device=DEV) for _ in range(4))
    ops.bn_finalize(part, nt, c, cs, npix, gam, bet, rmd, rvd, nbt, 0.1, 1e-5, scale, shift, mean, rstd)
    y = torch.empty_like(xd)
    ops.affine_relu(xd, scale, shift, y, relu)
    ops.sync_check()
    assert relerr(nchw(y, c), out.detach()) < 1e-5
    assert relerr(rmd.cpu(), rm_ref) < 1e-5 and relerr(rvd.cpu(), rv_ref) < 1e-5 and int(nbt) == 1
    if cs != c:
        assert float(scale[c:].abs().max()) == 0 and float(y[..., c:].abs().max()) == 0

    gd = nhwc(gout, cs)
    ntb = ops.bn_bwd_ntiles(npix)
    bpart = torch.empty(ntb, 2, cs, device=DEV)
    ops.bn_bwd_reduce(gd, xd, scale, shift, mean, rstd, relu, bpart)
    dgamma, dbeta = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    k123 = torch.empty(3, cs, device=DEV)
    ops.bn_bwd_finalize(bpart, ntb, c, cs, npix, gam, mean, rstd, dgamma, dbeta, k123)
    dx = torch.empty_like(xd)
    ops.bn_bwd_apply(gd, xd, scale, shift, k123, relu, dx)
    ops.sync_check()
    assert relerr(dgamma.cpu(), gamma.grad) < 2e-5 and relerr(dbeta.cpu(), beta.grad) < 2e-5
    assert relerr(nchw(dx, c), x.grad) < 2e-5


def test_mse_fused_loss_and_grad(ops):
    g = gen(12)
    shapes = [(2, 9, 11, 256), (2, 5, 6, 512), (2, 3, 3, 1024), (2, 2, 2, 2048)]
    factors = [1.0, 0.5, 2.0, 1.0]
    pairs, ref_terms, ref_grads = [], [], []
    for shp, f in zip(shapes, factors):
        t = torch.randn(shp, generator=g)
        s = F.relu(torch.randn(shp, generator=g))
        grad = torch.empty(shp, device=DEV)
        pairs.append((t.to(DEV), s.to(DEV), grad, f, True))
        ref_terms.append(F.mse_loss(t, s, reduction='sum').double() * f)
        ref_grads.append(2 * f * (s - t) * (s > 0))
    ml = ops.MseLaunch(pairs, DEV)
    out = ml.run().cpu()
    ops.sync_check()
    assert abs(float(out[0]) - float(sum(ref_terms))) <= 1e-6 * float(sum(ref_terms))
    for i in range(4):
        assert abs(float(out[1 + i]) - float(ref_terms[i])) <= 1e-6 * float(ref_terms[i])
        assert relerr(pairs[i][2].cpu(), ref_grads[i]) < 1e-6
    one, two = torch.ones((), device=DEV), torch.full((), 2.0, device=DEV)
    before = pairs[0][2].clone()
    ops.scale_by_device_scalar(pairs[0][2], one)
    assert torch.equal(pairs[0][2], before)
    ops.scale_by_device_scalar(pairs[0][2], two)
    assert torch.equal(pairs[0][2], before * 2)


def test_adam_matches_torch(ops):
    g = gen(13)
    p0 = torch.randn(10007, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    p, m, v = p0.to(DEV), torch.zeros(10007, device=DEV), torch.zeros(10007, device=DEV)
    for step in range(1, 4):
        grad = torch.randn(10007, generator=g) * (10.0 ** step)
        ref_p.grad = grad.clone()
        opt.step()
        ops.adam_step_flat(p, (grad * 4).to(DEV), m, v, 1e-3, 0.9, 0.999, 1e-8, step, grad_scale=0.25)
    ops.sync_check()
    assert float((p.cpu() - ref_p.detach()).abs().max()) < 1e-6


@pytest.mark.parametrize('shape,out', [((2, 64, 50, 84), (64, 64)), ((1, 64, 64, 64), (64, 64)),
                                       ((3, 32, 14, 14), (8, 8)), ((2, 64, 23, 100), (64, 64))])
def test_adaptive_avgpool_fwd_bwd(ops, shape, out):
    """nn.AdaptiveAvgPool2d incl. up-sampling windows (input smaller than output) and overlapping windows"""
    g = gen(31)
    x = torch.randn(*shape, generator=g, requires_grad=True)
    y = F.adaptive_avg_pool2d(x, out)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    yd = torch.full((shape[0], out[0], out[1], shape[1]), float('nan'), device=DEV)
    ops.adaptive_avgpool_fwd(nhwc(x.detach()), yd)
    dx = torch.full((shape[0], shape[2], shape[3], shape[1]), float('nan'), device=DEV)
    ops.adaptive_avgpool_bwd(nhwc(dy), dx)
    ops.sync_check()
    assert relerr(nchw(yd), y.detach()) < 1e-6
    assert relerr(nchw(dx), x.grad) < 1e-6


def test_linear_on_nchw_flatten_fwd_bwd(ops):
    """nn.Linear(16*8*8, 2) on z.flatten(1) with the activations stored NHWC, 16 channels padded to 32"""
    g = gen(32)
    n, c, h, w, nout = 5, 16, 8, 8, 2
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(nout, c * h * w, generator=g) / 32).requires_grad_(True)
    b = torch.randn(nout, generator=g).requires_grad_(True)
    out = F.linear(x.flatten(1), wt, b)
    do = torch.randn(out.shape, generator=g)
    out.backward(do)
    xd = nhwc(x.detach(), 32)
    od = torch.empty(n, nout, device=DEV)
    ops.linear_fwd(xd, c, wt.detach().to(DEV), b.detach().to(DEV), od)
    dw, db = torch.empty(nout, c * h * w, device=DEV), torch.empty(nout, device=DEV)
    dx = torch.full((n, h, w, 32), float('nan'), device=DEV)
    ops.linear_bwd(xd, c, wt.detach().to(DEV), do.to(DEV), dw, db, dx)
    ops.sync_check()
    assert relerr(od.cpu(), out.detach()) < 1e-6
    assert relerr(dw.cpu(), wt.grad) < 1e-6 and relerr(db.cpu(), b.grad) < 1e-6
    assert relerr(nchw(dx, c), x.grad) < 1e-6 and float(dx[..., c:].abs().max()) == 0.0


def test_softmax_rows_and_channel_sum(ops):
    g = gen(33)
    x = torch.randn(7, 2, generator=g) * 5
    y = torch.empty(7, 2, device=DEV)
    ops.softmax_rows(x.to(DEV), y)
    assert float((y.cpu() - x.softmax(dim=1)).abs().max()) < 1e-6
    t = torch.randn(3, 16, 14, 14, generator=g)
    out = torch.empty(16, device=DEV)
    ops.channel_sum(nhwc(t, 32), 16, out)
    ops.sync_check()
    assert relerr(out.cpu(), t.sum(dim=(0, 2, 3))) < 1e-5


@pytest.mark.parametrize('rows,cols', [(2, 2), (16, 2), (300, 2), (700, 5)])
def test_softmax_cross_entropy_matches_torch(ops, rows, cols):
    """hnd_softmax_ce_rows_fwd_bwd == nn.functional.cross_entropy (mean) and its autograd, the loss of the filter's
    step (reference src/ext_runner.py:58): fp32 arithmetic, tolerance 1e-6 relative on loss and gradient; rows with
    torch's ignore_index are not counted; wide logits do not overflow; the product wrapper carries autograd"""
    from hnd_ghnd_object_detectors_amd.models.ext.classifier import cross_entropy
    g = gen(35 + rows)
    x = (torch.randn(rows, cols, generator=g) * 6).double().requires_grad_(True)
    x.data[0] *= 20                                               # |logit| ~ 100: needs the max subtraction
    labels = torch.randint(0, cols, (rows,), generator=g)
    if rows > 2:
        labels[1] = -100
    ref = torch.nn.functional.cross_entropy(x, labels)
    ref.backward()
    loss, dx = torch.empty((), device=DEV), torch.full((rows, cols), float('nan'), device=DEV)
    ops.softmax_ce_rows(x.detach().float().to(DEV), labels.to(DEV), loss, dx)
    ops.sync_check()
    assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref)) + 1e-7
    assert relerr(dx.cpu(), x.grad.float()) < 1e-6
    if rows > 2:
        assert float(dx[1].abs().max()) == 0.0
    xd = x.detach().float().to(DEV).requires_grad_(True)
    out = cross_entropy(xd, labels.to(DEV)) * 3.0                 # the incoming gradient is applied on the device
    out.backward()
    assert relerr(xd.grad.cpu(), 3.0 * x.grad.float()) < 1e-6
    loss2, dx2 = torch.empty((), device=DEV), torch.empty(rows, cols, device=DEV)
    ops.softmax_ce_rows(x.detach().float().to(DEV), labels.to(DEV), loss2, dx2)
    assert torch.equal(loss2, loss) and torch.equal(dx2, dx)      # fixed summation order


@pytest.mark.parametrize('momentum,wd,nesterov', [(0.9, 1e-4, False), (0.0, 0.0, False), (0.9, 0.0, True)])
def test_sgd_matches_torch(ops, momentum, wd, nesterov):
    g = gen(34)
    p0 = torch.randn(10007, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref_p], lr=1e-2, momentum=momentum, weight_decay=wd, nesterov=nesterov)
    p, buf = p0.to(DEV), torch.zeros(10007, device=DEV)
    for step in range(4):
        grad = torch.randn(10007, generator=g)
        ref_p.grad = grad.clone()
        opt.step()
        ops.sgd_step_flat(p, (grad * 2).to(DEV), buf, 1e-2, momentum, 0.0, wd, nesterov, step == 0, grad_scale=0.5)
    ops.sync_check()
    assert float((p.cpu() - ref_p.detach()).abs().max()) < 1e-6


def test_conv_bias_in_epilogue_feeds_bn_statistics(ops):
    """neural-filter conv: bias added by epi_shift; the same epilogue's (sum, sum^2) partials include it"""
    g = gen(35)
    n, cin, h, w, cout, k, s = 3, 64, 31, 31, 32, 3, 2
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, s, 0)
    y = torch.empty(n, ref.shape[2], ref.shape[3], cout, device=DEV)
    m = n * ref.shape[2] * ref.shape[3]
    st = torch.zeros(ops.stats_tiles(m), 2, cout, device=DEV)
    ops.conv_forward(nhwc(x), ops.pack_weights(wt.to(DEV).contiguous(), chan_pad=cin), y, k, s, 0,
                     epi_shift=b.to(DEV), stats=st).run()
    ops.sync_check()
    assert relerr(nchw(y), ref) < 1e-4
    tot = st.sum(dim=0).cpu()
    assert relerr(tot[0], ref.sum(dim=(0, 2, 3))) < 1e-4 and relerr(tot[1], (ref * ref).sum(dim=(0, 2, 3))) < 1e-4


@pytest.mark.parametrize('tile', [2, 4, 6])
@pytest.mark.parametrize('n,cin,h,w,cout', [(2, 64, 13, 17, 96), (1, 256, 50, 84, 256), (3, 32, 8, 8, 64),
                                            (2, 128, 25, 42, 128), (1, 64, 1, 5, 32)])
def test_winograd_conv3x3_forward_matches_direct(ops, n, cin, h, w, cout, tile):
    """F(2x2,3x3): input transform -> 16 GEMMs in one igemm launch -> output transform, odd sizes, full epilogue"""
    g = gen(40 + n + cin + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    ref = F.relu(F.conv2d(x, wt, None, 1, 1) * sc[None, :, None, None] + sh[None, :, None, None] + res)
    ww = ops.WinoWeights(wt.to(DEV).contiguous(), tile=tile)
    nv, nm = ops.WinoConv.scratch_elems(n, h, w, cin, cout, tile)
    v, m = torch.empty(nv, device=DEV), torch.empty(nm, device=DEV)
    y = torch.full((n, h, w, cout), float('nan'), device=DEV)
    ops.WinoConv(nhwc(x), ww, y, v, m, epi_scale=sc.to(DEV), epi_shift=sh.to(DEV), res1=nhwc(res), relu=True).run()
    ops.sync_check()
    # F(4x4,3x3) multiplies by 4 / 5 / 8 and 1/24 in its transforms: ~1e-5 relative in fp32 (F(2x2): ~1e-6)
    assert relerr(nchw(y), ref) < (1e-4 if tile == 2 else 2e-4), relerr(nchw(y), ref)


@pytest.mark.parametrize('tile', [2, 4, 6])
def test_winograd_conv3x3_dgrad_with_prologue_and_mask(ops, tile):
    """data gradient of a 3x3 conv as the transposed Winograd conv: FrozenBN scale on load, ReLU mask on store"""
    g = gen(47)
    n, cin, h, w, cout = 2, 64, 14, 19, 128
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    sc = torch.rand(cout, generator=g) + 0.5
    out = F.conv2d(x, wt, None, 1, 1) * sc[None, :, None, None]
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    mk = torch.randn(n, cin, h, w, generator=g)
    base = torch.randn(n, cin, h, w, generator=g)
    ref = torch.where(mk > 0, x.grad + base, torch.zeros_like(base))
    ww = ops.WinoWeights(wt.to(DEV).contiguous(), dgrad=True, tile=tile)
    nv, nm = ops.WinoConv.scratch_elems(n, h, w, cout, cin, tile)
    v, m = torch.empty(nv, device=DEV), torch.empty(nm, device=DEV)
    dx = torch.full((n, h, w, cin), float('nan'), device=DEV)
    ops.WinoConv(nhwc(dy), ww, dx, v, m, pro_scale=sc.to(DEV), res1=nhwc(base), mask=nhwc(mk)).run()
    ops.sync_check()
    assert relerr(nchw(dx), ref) < (1e-4 if tile == 2 else 2e-4)


@pytest.mark.parametrize('n,cin,h,w,cout,pad', [(2, 64, 13, 17, 64, 1), (2, 256, 21, 30, 256, 0), (1, 128, 9, 9, 256, 0),
                                                (3, 64, 8, 11, 256, 1), (2, 256, 12, 16, 64, 1)])
@pytest.mark.parametrize('tile', [4, 6])
def test_winograd_conv2x2_forward_prologue_and_bn_stats(ops, n, cin, h, w, cout, pad, tile):
    """F(4x4,2x2) / F(6x6,2x2) head conv: BN+ReLU on load, raw output + per-channel (sum, sum^2) partials for the next BN"""
    g = gen(60 + cin + h)
    x = torch.randn(n, cin, h, w, generator=g)
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    wt = torch.randn(cout, cin, 2, 2, generator=g) / math.sqrt(cin * 4)
    ref = F.conv2d(F.relu(x * ps[None, :, None, None] + pb[None, :, None, None]), wt, None, 1, pad)
    oh, ow = ref.shape[2], ref.shape[3]
    ww = ops.Wino2Weights(wt.to(DEV).contiguous(), tile=tile)
    nv, nm = ops.Wino2Conv.scratch_elems(n, oh, ow, cin, cout, tile)
    v, m = torch.empty(nv, device=DEV), torch.empty(nm, device=DEV)
    y = torch.full((n, oh, ow, cout), float('nan'), device=DEV)
    nb = ops.Wino2Conv.stats_blocks(n, oh, ow, cout, tile)
    st = torch.full((nb, 2, cout), float('nan'), device=DEV)
    ops.Wino2Conv(nhwc(x), ww, y, v, m, pad, pro_scale=ps.to(DEV), pro_shift=pb.to(DEV), pro_relu=True, stats=st).run()
    ops.sync_check()
    assert relerr(nchw(y), ref) < 2e-4, relerr(nchw(y), ref)
    tot = st.sum(dim=0).cpu()
    assert relerr(tot[0], ref.sum(dim=(0, 2, 3))) < 2e-4 and relerr(tot[1], (ref * ref).sum(dim=(0, 2, 3))) < 2e-4


@pytest.mark.parametrize('tile', [4, 6])
@pytest.mark.parametrize('pad', [0, 1])
def test_winograd_conv2x2_dgrad(ops, pad, tile):
    g = gen(71 + pad)
    n, cin, h, w, cout = 2, 128, 14, 19, 256
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = torch.randn(cout, cin, 2, 2, generator=g) / math.sqrt(cin * 4)
    out = F.conv2d(x, wt, None, 1, pad)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    ww = ops.Wino2Weights(wt.to(DEV).contiguous(), dgrad=True, tile=tile)
    nv, nm = ops.Wino2Conv.scratch_elems(n, h, w, cout, cin, tile)
    v, m = torch.empty(nv, device=DEV), torch.empty(nm, device=DEV)
    dx = torch.full((n, h, w, cin), float('nan'), device=DEV)
    ops.Wino2Conv(nhwc(dy), ww, dx, v, m, 1 - pad).run()
    ops.sync_check()
    assert relerr(nchw(dx), x.grad) < 2e-4


@pytest.mark.parametrize('n,cin,h,w,cout,pad', [(2, 128, 14, 19, 256, 0), (3, 256, 9, 12, 256, 0), (2, 64, 10, 10, 128, 1)])
@pytest.mark.parametrize('tile', [4, 6])
def test_winograd_conv2x2_wgrad_reuses_forward_transform(ops, n, cin, h, w, cout, pad, tile):
    """dW in the Winograd domain: forward V (with the BN+ReLU prologue baked in) x transformed dy, 25 grouped
    split-K reductions, inverse transform -- against autograd"""
    g = gen(80 + cin + h)
    x = torch.randn(n, cin, h, w, generator=g)
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    wt = (torch.randn(cout, cin, 2, 2, generator=g) / math.sqrt(cin * 4)).requires_grad_(True)
    out = F.conv2d(F.relu(x * ps[None, :, None, None] + pb[None, :, None, None]), wt, None, 1, pad)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    oh, ow = out.shape[2], out.shape[3]
    ww = ops.Wino2Weights(wt.detach().to(DEV).contiguous(), tile=tile)
    nv, nm = ops.Wino2Conv.scratch_elems(n, oh, ow, cin, cout, tile)
    v, m = torch.empty(nv, device=DEV), torch.full((nm,), float('nan'), device=DEV)
    y = torch.empty(n, oh, ow, cout, device=DEV)
    fwd = ops.Wino2Conv(nhwc(x), ww, y, v, m, pad, pro_scale=ps.to(DEV), pro_shift=pb.to(DEV), pro_relu=True)
    fwd.run()
    dw = torch.full((cout, cin, 2, 2), float('nan'), device=DEV)
    s = torch.empty((tile + 1) ** 2 * cout * cin, device=DEV)
    outs = []
    for _ in range(2):
        ops.Wino2Wgrad(fwd, nhwc(dy), dw, m, s).run()       # z may alias the forward's M scratch
        ops.sync_check()
        outs.append(dw.cpu().clone())
    assert relerr(outs[0], wt.grad) < 2e-4, relerr(outs[0], wt.grad)
    assert torch.equal(outs[0], outs[1])                     # fixed-order reductions: bit-reproducible


@pytest.mark.parametrize('n,oh,ow,c,pad,relu', [(2, 37, 50, 128, 1, True), (1, 44, 31, 256, 1, False),
                                                (2, 30, 43, 64, 0, True), (1, 7, 13, 64, 0, True)])
def test_bn_backward_apply_fused_into_both_winograd_transforms(ops, n, oh, ow, c, pad, relu):
    """hnd_wino26_bnbwd_transforms: dy = k1 * [bn(x) > 0] g + k2 x + k3 is never written; the kernel must give the V of
    hnd_wino2_input(dy, pad) and the Z of hnd_wino2_dy(dy) computed from the materialised dy of hnd_bn_bwd_apply -- for
    both paddings (the two transforms tile different extents: with pad 0 the data gradient's grid can be the SMALLER one)"""
    from hnd_ghnd_object_detectors_amd import _lib
    L = _lib.load()
    g_ = gen(60 + oh + c)
    g = torch.randn(n, oh, ow, c, generator=g_).to(DEV)
    x = torch.randn(n, oh, ow, c, generator=g_).to(DEV)
    sc, sh = (torch.rand(c, generator=g_) + 0.5).to(DEV), (torch.randn(c, generator=g_) * 0.5).to(DEV)
    k123 = (torch.randn(3, c, generator=g_) * 0.5).to(DEV).contiguous()
    dy = torch.empty_like(g)
    ops.bn_bwd_apply(g, x, sc, sh, k123, relu, dy)
    ih, iw = oh + 2 * pad - 1, ow + 2 * pad - 1
    tp_d, tp_w = int(L.hnd_wino2_tiles_pad(n, ih, iw, 6)), int(L.hnd_wino2_tiles_pad(n, oh, ow, 6))
    v_ref = torch.full((49 * tp_d * c,), float('nan'), device=DEV)
    z_ref = torch.full((49 * tp_w * c,), float('nan'), device=DEV)
    s = ops.stream_ptr()
    _lib.check(L.hnd_wino2_input(dy.data_ptr(), v_ref.data_ptr(), n, oh, ow, c, pad, None, None, 0, 6, s))
    _lib.check(L.hnd_wino2_dy(dy.data_ptr(), z_ref.data_ptr(), n, oh, ow, c, c, 6, s))
    v, z = torch.full_like(v_ref, float('nan')), torch.full_like(z_ref, float('nan'))
    _lib.check(L.hnd_wino26_bnbwd_transforms(g.data_ptr(), x.data_ptr(), sc.data_ptr(), sh.data_ptr(), k123.data_ptr(),
                                             int(relu), n, oh, ow, c, pad, v.data_ptr(), z.data_ptr(), s))
    ops.sync_check()
    for got, ref, tp, tiles in ((v, v_ref, tp_d, n * ((ih + 5) // 6) * ((iw + 5) // 6)),
                                (z, z_ref, tp_w, n * ((oh + 5) // 6) * ((ow + 5) // 6))):
        got, ref = got.view(49, tp, c)[:, :tiles], ref.view(49, tp, c)[:, :tiles]
        assert not bool(torch.isnan(got).any()) and not bool(torch.isnan(ref).any())
        assert relerr(got.cpu(), ref.cpu()) < 2e-6, relerr(got.cpu(), ref.cpu())


@pytest.mark.parametrize('n,cin,h,w,cout,pad,relu', [(2, 128, 37, 50, 128, 1, True), (1, 256, 44, 31, 64, 0, True),
                                                     (2, 64, 20, 27, 256, 1, False)])
def test_bn_backward_sums_folded_into_the_data_gradient_output_transform(ops, n, cin, h, w, cout, pad, relu):
    """Wino2Conv(bwd_stats=...) -> hnd_wino26_output_bnbwd_stats: the data gradient g of a 2x2 conv (F(6x6,2x2)) and, from
    the same launch, sum [bn(x) > 0] g and sum [bn(x) > 0] g xhat per channel of the BatchNorm before that conv -- g
    bit-identical to the plain output transform, the sums those of hnd_bn_bwd_reduce over the stored g"""
    g_ = gen(130 + h + cin)
    # data gradient of conv(x: cin -> cout): input dy [n, oh, ow, cout], output g [n, h, w, cin]
    oh, ow = h + 2 * pad - 1, w + 2 * pad - 1
    dy = torch.randn(n, oh, ow, cout, generator=g_).to(DEV)
    wt = torch.randn(cout, cin, 2, 2, generator=g_) / math.sqrt(cout * 4)
    xr = torch.randn(n, h, w, cin, generator=g_).to(DEV)
    sc, sh = (torch.rand(cin, generator=g_) + 0.5).to(DEV), (torch.randn(cin, generator=g_) * 0.5).to(DEV)
    mu, rs = (torch.randn(cin, generator=g_) * 0.2).to(DEV), (torch.rand(cin, generator=g_) + 0.5).to(DEV)
    ww = ops.Wino2Weights(wt.to(DEV).contiguous(), dgrad=True, tile=6)
    nv, nm = ops.Wino2Conv.scratch_elems(n, h, w, cout, cin, 6)
    v, m = torch.empty(nv, device=DEV), torch.empty(nm, device=DEV)
    plain = torch.full((n, h, w, cin), float('nan'), device=DEV)
    ops.Wino2Conv(dy, ww, plain, v, m, 1 - pad).run()
    nblk = ops.Wino2Conv.stats_blocks(n, h, w, cin, 6)
    part = torch.full((nblk, 2, cin), float('nan'), device=DEV)
    got = torch.full_like(plain, float('nan'))
    ops.Wino2Conv(dy, ww, got, v, m, 1 - pad, bwd_stats=(xr, sc, sh, mu, rs, relu, part)).run()
    ref_part = torch.zeros(ops.bn_bwd_ntiles(n * h * w), 2, cin, device=DEV)
    ops.bn_bwd_reduce(got, xr, sc, sh, mu, rs, relu, ref_part)
    ops.sync_check()
    assert torch.equal(got, plain)
    a, b = part.double().sum(0).cpu(), ref_part.double().sum(0).cpu()
    assert not bool(torch.isnan(a).any())
    assert float((a - b).abs().max() / b.abs().max()) < 2e-6, float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize('n,cin,h,w,cout,relu', [(2, 256, 37, 50, 64, True), (1, 64, 44, 31, 64, False),
                                                 (1, 128, 9, 14, 128, True)])
def test_bn_backward_sums_from_the_tiled_data_gradient_epilogue(ops, n, cin, h, w, cout, relu):
    """hnd_conv_desc.bwd_x (ABI 9): the data gradient of a 2x2 head conv on the tiled kernel writes, per 128-pixel tile,
    the BatchNorm-backward partials of the g it stores (incl. the checked path of the last, partial tile) -- g bit-identical
    to the launch without them, the sums those of hnd_bn_bwd_reduce"""
    g_ = gen(170 + h + cin)
    pad = 1
    oh, ow = h + 2 * pad - 1, w + 2 * pad - 1
    dy = torch.randn(n, oh, ow, cout, generator=g_).to(DEV)
    wt = torch.randn(cout, cin, 2, 2, generator=g_) / math.sqrt(cout * 4)
    xr = torch.randn(n, h, w, cin, generator=g_).to(DEV)
    sc, sh = (torch.rand(cin, generator=g_) + 0.5).to(DEV), (torch.randn(cin, generator=g_) * 0.5).to(DEV)
    mu, rs = (torch.randn(cin, generator=g_) * 0.2).to(DEV), (torch.rand(cin, generator=g_) + 0.5).to(DEV)
    pk = ops.pack_weights(wt.to(DEV), transposed=True)
    geo = dict(kh=2, kw=2, oh=h, ow=w, sh=1, dh=-1, bh=pad, sw=1, dw=-1, bw=pad, cout=cin)
    plain = torch.full((n, h, w, cin), float('nan'), device=DEV)
    dummy = torch.empty(ops.stats_tiles(n * h * w), 2, cin, device=DEV)
    ops.conv_desc(dy, pk, plain, stats=dummy, **geo).run()           # (same kernel: `stats` keeps it on the tiled one)
    part = torch.full((ops.stats_tiles(n * h * w), 2, cin), float('nan'), device=DEV)
    got = torch.full_like(plain, float('nan'))
    l = ops.conv_desc(dy, pk, got, stats=part, bwd_stats=(xr, sc, sh, mu, rs, relu), **geo)
    assert l.variant.startswith('igemm_128'), l.variant
    l.run()
    ref_part = torch.zeros(ops.bn_bwd_ntiles(n * h * w), 2, cin, device=DEV)
    ops.bn_bwd_reduce(got, xr, sc, sh, mu, rs, relu, ref_part)
    ops.sync_check()
    assert torch.equal(got, plain)
    a, b = part.double().sum(0).cpu(), ref_part.double().sum(0).cpu()
    assert not bool(torch.isnan(a).any())
    assert float((a - b).abs().max() / b.abs().max()) < 2e-6, float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize('n,cin,h,w,cout,pad', [(2, 64, 37, 50, 256, 1), (1, 64, 44, 31, 256, 0)])
def test_winograd_wgrad_on_an_input_transform_of_its_own(ops, n, cin, h, w, cout, pad, monkeypatch):
    """ops.Wino2InputTransform + Wino2Wgrad: the Winograd-domain weight gradient of a conv whose FORWARD is direct (the
    head's 64 -> 256 conv) -- V made on its own from the conv input with the BN prologue, 49 grouped reductions with 64
    columns each on the ring kernel's 4 x 1 wave arrangement -- against autograd and against the staged kernel"""
    g = gen(90 + h)
    x = torch.randn(n, cin, h, w, generator=g)
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    wt = (torch.randn(cout, cin, 2, 2, generator=g) / math.sqrt(cin * 4)).requires_grad_(True)
    out = F.conv2d(x * ps[None, :, None, None] + pb[None, :, None, None], wt, None, 1, pad)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    oh, ow = out.shape[2], out.shape[3]
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('HND_WGRAD_RING', mode)
        v = torch.empty(ops.Wino2InputTransform.scratch_elems(n, oh, ow, cin, 6), device=DEV)
        own = ops.Wino2InputTransform(nhwc(x), v, pad, cout, 6, pro_scale=ps.to(DEV), pro_shift=pb.to(DEV))
        z = torch.empty(49 * own.tiles_pad * cout, device=DEV)
        sbuf = torch.empty(49 * cout * cin, device=DEV)
        dw = torch.full((cout, cin, 2, 2), float('nan'), device=DEV)
        wg = ops.Wino2Wgrad(own, nhwc(dy), dw, z, sbuf)
        own._run_input()
        wg.run()
        ops.sync_check()
        res[mode] = (dw.cpu().clone(), wg.gemm.variant)
    assert res['0'][1] == 'wgrad_m128' and res['1'][1] == 'wgrad_ring', (res['0'][1], res['1'][1])
    assert relerr(res['1'][0], wt.grad) < 2e-4 and relerr(res['0'][0], wt.grad) < 2e-4, relerr(res['1'][0], wt.grad)
    assert relerr(res['1'][0], res['0'][0]) < 2e-5


@pytest.mark.parametrize('n,cin,h,w,cout,pad', [(2, 256, 37, 50, 64, 1), (1, 256, 44, 31, 64, 0)])
def test_winograd_wgrad_of_a_64_output_channel_conv_runs_swapped_on_the_ring_kernel(ops, n, cin, h, w, cout, pad, monkeypatch):
    """round 5: the head's 256 -> 64 conv (encoder.5, /root/reference/src/models/mimic/resnet_layer.py:48) in the Winograd
    domain is 64 rows x 256 columns per component -- the LDS-staged kernel's case (0.50 of the matrix peak).  With the
    grouped reductions' operands swapped (x = z, dy = v) it is the ring kernel's 64-column shape; the output transform
    reads s transposed (hnd_wino2_wgrad_output_t).  Against autograd and against the unswapped staged kernel."""
    g = gen(190 + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = (torch.randn(cout, cin, 2, 2, generator=g) / math.sqrt(cin * 4)).requires_grad_(True)
    out = F.conv2d(x, wt, None, 1, pad)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    oh, ow = out.shape[2], out.shape[3]
    res = {}
    for swap in ('0', '1'):
        monkeypatch.setenv('HND_WGRAD_SWAP', swap)
        v = torch.empty(ops.Wino2InputTransform.scratch_elems(n, oh, ow, cin, 6), device=DEV)
        own = ops.Wino2InputTransform(nhwc(x), v, pad, cout, 6)
        z = torch.empty(49 * own.tiles_pad * cout, device=DEV)
        sbuf = torch.full((49 * cout * cin,), float('nan'), device=DEV)
        dw = torch.full((cout, cin, 2, 2), float('nan'), device=DEV)
        wg = ops.Wino2Wgrad(own, nhwc(dy), dw, z, sbuf)
        own._run_input()
        wg.run()
        ops.sync_check()
        res[swap] = (dw.cpu().clone(), wg.gemm.variant, wg.swapped)
    assert res['0'][1] == 'wgrad_m64' and not res['0'][2], res['0'][1:]
    assert res['1'][1] == 'wgrad_ring' and res['1'][2], res['1'][1:]
    assert relerr(res['1'][0], wt.grad) < 2e-4 and relerr(res['0'][0], wt.grad) < 2e-4, relerr(res['1'][0], wt.grad)
    assert relerr(res['1'][0], res['0'][0]) < 2e-5
    wg.run()                                                  # fixed summation order
    ops.sync_check()
    assert torch.equal(dw.cpu(), res['1'][0])


def test_subsample_and_fill(ops):
    x = torch.randn(2, 256, 7, 9, generator=gen(14))
    y = torch.empty(2, 4, 5, 256, device=DEV)
    ops.subsample2(nhwc(x), y)
    assert torch.equal(nchw(y), F.max_pool2d(x, 1, 2, 0))
    ops.fill(y, 3.5)
    ops.sync_check()
    assert float(y.min()) == 3.5 and float(y.max()) == 3.5


@pytest.mark.parametrize('c,shift,shape', [(3, 0.0, (2, 13, 17)), (3, 5.0, (2, 13, 17)), (6, -2.0, (2, 13, 17)),
                                           (3, 0.3, (4, 204, 340)), (3, -40.0, (1, 1, 1)), (12, 1e-3, (3, 50, 84))])
def test_bottleneck_codec_matches_myutils_semantics(ops, c, shift, shape):
    """uint8 quantise / dequantise vs the restated myutils tensor_util (oracle/myutils_r.py): byte work, so the
    bytes, (scale, zero_point) and the dequantised floats must be BIT-EXACT on identical fp32 input -- including the
    b3ch bottleneck's real extent [4, 3, 204, 340] (832 k values: ~1 in 2^10 lands within an ulp of a .5 tie)."""
    from oracle.myutils_r import quantize_tensor, dequantize_tensor
    g = gen(15 + c)
    n, h, w = shape
    z = torch.randn(n, c, h, w, generator=g) * 3 + shift          # all-positive when shifted: pad zeros must not count
    if z.numel() == c:
        z[0, 1:] += 2.0                                            # a single pixel still has max > min
    ref_q = quantize_tensor(z.clone(), num_bits=8)
    ref = dequantize_tensor(ref_q)
    cs = ops.chan_pad_of(c)
    buf = nhwc(z, cs)
    q = torch.empty(buf.shape, dtype=torch.uint8, device=DEV)
    qp = torch.empty(4, device=DEV)
    scratch = torch.empty(ops.minmax_scratch_elems(), device=DEV)
    ops.quantize_u8(buf, c, q, qp, scratch)
    out = torch.full(buf.shape, float('nan'), device=DEV)
    ops.dequantize_u8(q, qp, out, c)
    ops.sync_check()
    lo, hi, scale, zp = [float(v) for v in qp.cpu()]
    assert lo == float(z.min()) and hi == float(z.max())
    assert scale == float(ref_q.scale) and zp == float(ref_q.zero_point)
    got_q = q.cpu()[..., :c].permute(0, 3, 1, 2)
    assert torch.equal(got_q, ref_q.tensor), 'quantised bytes differ in %d places' % int((got_q != ref_q.tensor).sum())
    assert torch.equal(nchw(out, c), ref)
    if cs != c:
        assert float(out[..., c:].abs().max()) == 0.0 and int(q[..., c:].max()) == 0


def test_f16_roundtrip(ops):
    x = torch.randn(5000, generator=gen(16)) * 100
    d = x.to(DEV)
    ops.roundtrip_f16(d)
    assert torch.equal(d.cpu(), x.half().float())


def test_conv_random_geometries_all_tiles(ops, monkeypatch):
    """seeded sweep over kernel size / stride / padding / odd extents / channel counts / fused options, on every
    block-tile build (HND_DEBUG_PICKER=igemm_tile=N override), forward and data-gradient, vs torch CPU."""
    import random
    rnd = random.Random(1234)
    worst = 0.0
    for it in range(36):
        k = rnd.choice([1, 1, 2, 3, 3, 5, 7])
        s = rnd.choice([1, 1, 2])
        p = rnd.randint(0, k // 2 + (1 if k == 2 else 0))
        cin = rnd.choice([3, 32, 64, 96, 160])
        cout = rnd.choice([3, 5, 32, 64, 100, 128, 192, 256])
        n = rnd.randint(1, 3)
        h, w = rnd.randint(k + 1, 29), rnd.randint(k + 1, 33)
        if k == 7 and cin != 3:
            cin = 32
        g = gen(5000 + it)
        x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
        wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
        out = F.conv2d(x, wt, None, s, p)
        use_res, use_relu, use_scale = rnd.random() < 0.5, rnd.random() < 0.5, rnd.random() < 0.5
        es, eb = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        res = torch.randn(out.shape, generator=g)
        ref = out.detach()
        if use_scale:
            ref = ref * es[None, :, None, None] + eb[None, :, None, None]
        if use_res:
            ref = ref + res
        if use_relu:
            ref = F.relu(ref)
        dy = torch.randn(out.shape, generator=g)
        out.backward(dy)
        cp_in, cp_out = ops.chan_pad_of(cin), ops.chan_pad_of(cout)
        monkeypatch.setenv('HND_DEBUG_PICKER', 'igemm_tile=%d' % (it % 4))
        xd, wd = nhwc(x.detach(), cp_in), wt.to(DEV).contiguous()
        y = torch.full((n, out.shape[2], out.shape[3], cp_out), float('nan'), device=DEV)
        pad_c = lambda v: F.pad(v, (0, cp_out - cout)).to(DEV)
        ops.conv_forward(xd, ops.pack_weights(wd, chan_pad=cp_in), y, k, s, p,
                         epi_scale=pad_c(es) if use_scale else None, epi_shift=pad_c(eb) if use_scale else None,
                         res1=nhwc(res, cp_out) if use_res else None, relu=use_relu).run()
        dx = torch.full((n, h, w, cp_in), float('nan'), device=DEV)
        if s == 2 and k == 1:
            ops.fill(dx, 0.0)
        launches, _ = ops.conv_dgrad(nhwc(dy, cp_out), wd, dx, k, s, p, accumulate=(s == 2 and k == 1))
        for l in launches:
            l.run()
        ops.sync_check()
        e1, e2 = relerr(nchw(y, cout), ref), relerr(nchw(dx, cin), x.grad)
        worst = max(worst, e1, e2)
        assert e1 < 1e-4 and e2 < 1e-4, (it, k, s, p, cin, cout, n, h, w, e1, e2)
    monkeypatch.delenv('HND_DEBUG_PICKER')


def test_conv_linearity_at_full_size(ops):
    """size-independent property at the benchmark's extent (3x3 256->256 on 200x336, batch 2): conv(a*x + b*z) ==
    a*conv(x) + b*conv(z) to fp32 rounding, and a checksum against a strided torch reference of a few output rows."""
    g = gen(17)
    n, c, h, w = 2, 256, 200, 336
    x, z = torch.randn(n, h, w, c, generator=g), torch.randn(n, h, w, c, generator=g)
    wt = torch.randn(256, 256, 3, 3, generator=g) / math.sqrt(256 * 9)
    pw = ops.pack_weights(wt.to(DEV).contiguous())
    outs = []
    for inp in (x, z, 1.5 * x - 0.25 * z):
        y = torch.empty(n, h, w, 256, device=DEV)
        ops.conv_forward(inp.to(DEV), pw, y, 3, 1, 1).run()
        outs.append(y)
    ops.sync_check()
    lin = 1.5 * outs[0] - 0.25 * outs[1]
    assert float((outs[2] - lin).abs().max() / lin.abs().max()) < 1e-5
    rows = slice(97, 103)
    ref = F.conv2d(x[:1, 96:104].permute(0, 3, 1, 2), wt, None, 1, 1)[:, :, 1:-1]      # output rows 97..102
    got = outs[0][:1, rows].cpu().permute(0, 3, 1, 2)
    assert relerr(got, ref) < 1e-4


def test_bad_arguments_raise(ops):
    x = torch.zeros(1, 4, 4, 48, device=DEV)      # cin 48: neither 4 nor a multiple of 32
    w = torch.zeros(64, 48, 1, 1, device=DEV)
    with pytest.raises(RuntimeError, match='cin=48'):
        pw = ops.pack_weights(w, chan_pad=48)
        ops.conv_forward(x, pw, torch.zeros(1, 4, 4, 64, device=DEV), 1).run()


def test_native_rccl_communicator_world_of_one(ops):
    """include/hnd_hip.h hnd_comm_*: the C ABI's own RCCL communicator (resolved at run time from the RCCL that
    PyTorch-ROCm already loaded).  A 1-GPU box can only form a world of one, where the averaging all-reduce must be
    the identity, bit for bit, and must run on the stream it is given."""
    import ctypes as C
    from hnd_ghnd_object_detectors_amd import _lib
    L = _lib.load()
    nbytes = int(L.hnd_workspace_size(5, None, 0))
    assert nbytes == 128
    uid = C.create_string_buffer(nbytes)
    assert L.hnd_comm_unique_id(uid, nbytes) == 0, L.hnd_last_error_string()
    assert any(b != 0 for b in uid.raw)
    comm = C.c_void_p()
    assert L.hnd_comm_init(0, 1, uid.raw, nbytes, C.byref(comm)) == 0, L.hnd_last_error_string()
    g = gen(91)
    flat = torch.randn(586566, generator=g).to(DEV)
    before = flat.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    assert L.hnd_allreduce_avg_flat(comm, flat.data_ptr(), flat.numel(), side.cuda_stream) == 0, L.hnd_last_error_string()
    side.synchronize()
    assert torch.equal(flat, before)
    assert L.hnd_allreduce_avg_flat(comm, None, 5, None) == -1                  # argument errors are reported
    assert L.hnd_comm_init(3, 2, uid.raw, nbytes, C.byref(C.c_void_p())) == -1
    assert L.hnd_comm_destroy(comm) == 0


def test_thin_n_kernel_is_bit_identical_to_the_mfma_path(tmp_path):
    """cout <= 4 launches (the 3-channel bottleneck: layer1.conv3 forward with BN prologue + statistics, conv4 data
    gradient) run on a vector-ALU kernel that reproduces the MFMA kernel's accumulation order, epilogue expressions
    and statistics tree: every output, every statistics partial, and two whole distillation steps (losses, updated
    parameters, BN buffers) must be IDENTICAL BITS with the kernel switched off (HND_THIN_N=0: MFMA tiles) and on."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ('0', '1'):
        out = str(tmp_path / ('thin%s.pt' % mode))
        env = dict(os.environ, HND_THIN_N=mode)
        res = subprocess.run([sys.executable, os.path.join(root, 'tests', 'thin_worker.py'), out], cwd=root, env=env,
                             capture_output=True, text=True, timeout=2400)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
        outs[mode] = torch.load(out, weights_only=False)
    assert all(v != 'thin_n4' for v in outs['0']['variants'])
    assert all(v == 'thin_n4' for v in outs['1']['variants']), outs['1']['variants']
    assert outs['0']['losses'] == outs['1']['losses']
    for k, v in outs['0'].items():
        if isinstance(v, torch.Tensor):
            assert torch.equal(v, outs['1'][k]), k
    for grp in ('params', 'buffers'):
        for k, v in outs['0'][grp].items():
            assert torch.equal(v, outs['1'][grp][k]), (grp, k)


@pytest.mark.parametrize('case', [
    # cin, h, w, cout, stride, residual, mask, prologue, groups, batch
    (256, 200, 336, 256, 1, False, False, False, 1, 2),      # bres_128, K = 256
    (256, 200, 336, 128, 2, False, False, False, 1, 8),      # stride-2 rows (downsample)
    (64, 200, 336, 256, 1, True, False, False, 1, 2),        # K = 64, residual epilogue
    (128, 100, 168, 512, 1, True, True, True, 1, 8),         # K = 128, residual + mask + prologue
    (512, 100, 168, 64, 1, False, False, True, 1, 16),       # bres_64, K = 512, prologue
    (128, 100, 168, 128, 1, False, False, True, 1, 16),      # K = 128 with prologue (one-wave kernel: ring = tile)
    (256, 101, 169, 256, 1, False, False, False, 1, 8),      # ragged last chunk on the deferred-epilogue kernel
    (256, 201, 337, 64, 1, True, False, False, 1, 4),        # bres_64 with K = 256, ragged last chunk
    (256, 50, 84, 1024, 1, True, False, False, 1, 16),       # residual on the one-wave kernel (layer3 conv3)
    (512, 25, 42, 2048, 1, True, False, False, 1, 16),       # K = 512 + residual: stays on the 8-wave kernel
    (256, 200, 336, 256, 1, 'up', False, False, 1, 2),       # FPN lateral conv: + nearest-upsampled top-down map
    (256, 37, 53, 256, 1, False, False, False, 9, 64),       # grouped (Winograd-style) weights
])
def test_bres_kernel_is_bit_identical_to_the_tiled_kernel(ops, case, monkeypatch):
    """The B-resident persistent GEMM (csrc/conv_bres.hip: weight slice resident in LDS, A fragments straight from
    global memory, no barrier in the main loop) keeps the tiled kernel's accumulation order and shares its prologue /
    epilogue code, so every output must be IDENTICAL BITS -- and both must match a torch fp32 convolution."""
    cin, h, w, cout, s, res, msk, pro, groups, n = case
    g = torch.Generator().manual_seed(11)
    monkeypatch.setenv('HND_DEBUG_PICKER', 'bres_all')
    monkeypatch.setenv('HND_BSTREAM', '0')
    if groups > 1:
        tiles_pad = (n * h * w + 127) // 128 * 128
        x = torch.randn(1, 1, groups * tiles_pad, cin, generator=g).to(DEV)
        y = torch.empty(1, 1, groups * tiles_pad, cout, device=DEV)
        ws = [torch.randn(cout, cin, 1, 1, generator=g).to(DEV) / cin ** 0.5 for _ in range(groups)]
        pks = [ops.pack_weights(wt) for wt in ws]
        pk = ops.PackedWeight.__new__(ops.PackedWeight)
        pk.buf = torch.cat([p.buf for p in pks])
        pk.kdim, pk.rows, pk.chan_pad, pk.chan_real = pks[0].kdim, cout, cin, cin
    else:
        x = torch.randn(n, h, w, cin, generator=g).to(DEV)
        oh, ow = ops.conv_out_size(h, 1, s, 0), ops.conv_out_size(w, 1, s, 0)
        y = torch.empty(n, oh, ow, cout, device=DEV)
        wt = torch.randn(cout, cin, 1, 1, generator=g).to(DEV) / cin ** 0.5
        pk = ops.pack_weights(wt)
    sc, sh = torch.rand(cout, generator=g).to(DEV) + 0.5, torch.randn(cout, generator=g).to(DEV)
    r = torch.randn(y.shape, generator=g).to(DEV) if res else None
    if res == 'up':                 # res1_mode 1: the residual is read from the half-resolution map (FPN top-down path)
        r = torch.randn(n, (h + 1) // 2, (w + 1) // 2, cout, generator=g).to(DEV)
    mk = (torch.rand(y.shape, generator=g).to(DEV) - 0.3).clamp_min(0) if msk else None
    ps = (torch.rand(cin, generator=g).to(DEV) + 0.5) if pro else None
    pb = torch.randn(cin, generator=g).to(DEV) if pro else None
    outs, variants = {}, {}
    plain = not msk and not (res and pro)        # what the one-wave kernel takes: optional res1, no mask
    for mode in ('0', '512', 'one_wave'):
        if mode == 'one_wave' and not plain:
            continue
        monkeypatch.setenv('HND_BRES', '0' if mode == '0' else '512')
        monkeypatch.setenv('HND_BRES2', '1' if mode == 'one_wave' else '0')
        if groups > 1:
            l = ops.conv_desc(x, pk, y, kh=1, kw=1, oh=1, ow=groups * tiles_pad, sh=1, dh=1, bh=0, sw=1, dw=1, bw=0,
                              cout=cout)
            l.desc.w_group_rows, l.desc.w_group_stride = tiles_pad, pks[0].buf.numel()
        else:
            l = ops.conv_forward(x, pk, y, 1, s, 0, epi_scale=sc, epi_shift=sh, res1=r, res1_up=res == 'up', mask=mk,
                                 relu=not msk, pro_scale=ps, pro_shift=pb, pro_relu=pro)
        l.refresh_variant()
        y.fill_(float('nan'))
        l.run()
        torch.cuda.synchronize()
        outs[mode], variants[mode] = y.clone(), l.variant
    assert variants['0'].startswith('igemm') and variants['512'] in ('bres_128', 'bres_64'), variants
    assert not bool(torch.isnan(outs['512']).any())
    assert torch.equal(outs['0'], outs['512']), float((outs['0'] - outs['512']).abs().max())
    if plain and cin >= 128:        # the one-wave-per-SIMD kernel (asm register ring, deferred epilogue)
        if (pro or res) and cin == 512:     # (those builds spill: that combination stays on the 8-wave kernel)
            assert variants['one_wave'] == 'bres_64', variants
        else:
            assert variants['one_wave'] in ('bres2_128', 'bres2_64'), variants
        assert torch.equal(outs['0'], outs['one_wave']), float((outs['0'] - outs['one_wave']).abs().max())
    if groups == 1:
        xin = x.permute(0, 3, 1, 2)
        if pro:
            xin = torch.relu(xin * ps.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1))
        ref = torch.nn.functional.conv2d(xin, wt, stride=s) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        if res == 'up':
            ref = ref + F.interpolate(r.permute(0, 3, 1, 2), size=ref.shape[-2:], mode='nearest')
        elif res:
            ref = ref + r.permute(0, 3, 1, 2)
        if msk:
            ref = torch.where(mk.permute(0, 3, 1, 2) > 0, ref, torch.zeros_like(ref))
        else:
            ref = torch.relu(ref)
        got = outs['512'].permute(0, 3, 1, 2)
        assert float((got - ref).norm() / ref.norm()) < 1e-5
    else:
        for gi in (0, groups - 1):
            rows = slice(gi * tiles_pad, (gi + 1) * tiles_pad)
            ref = x[0, 0, rows] @ ws[gi].view(cout, cin).t()
            assert float((outs['512'][0, 0, rows] - ref).norm() / ref.norm()) < 1e-5


@pytest.mark.parametrize('case', [
    # cin, h, w, cout, k, stride, pad, residual, mask, prologue, groups, batch
    (1024, 50, 84, 256, 1, 1, 0, False, False, False, 1, 16),    # layer3 conv1: 1050 tiles, four per workgroup
    (1024, 50, 84, 512, 1, 1, 0, True, False, False, 1, 4),      # + residual
    (2048, 25, 42, 512, 1, 1, 0, False, True, True, 1, 16),      # dgrad-like: prologue scale + ReLU-backward mask
    (1024, 37, 41, 192, 1, 1, 0, False, False, False, 1, 3),     # cout = 3 * 64: the 256 x 64 tile, ragged rows
    (1024, 50, 84, 2048, 1, 2, 0, False, False, False, 1, 8),    # stride-2 1x1 (layer4 downsample)
    (128, 100, 168, 128, 3, 2, 1, False, False, False, 1, 8),    # layer2.0.conv2: 3x3 stride 2 over taps, K = 1152
    (256, 51, 85, 256, 3, 2, 1, False, False, True, 1, 8),       # ... with prologue: padding is a zero AFTER it
    (512, 25, 42, 512, 3, 2, 1, True, False, False, 1, 16),      # K = 4608
    (128, 40, 56, 256, 3, 1, 1, False, False, False, 1, 4),      # stride-1 3x3 on the direct path
    (512, 50, 84, 256, 1, 1, 0, False, False, False, 1, 8),      # K = 512 (eligible, the picker prefers bres)
    (256, 37, 53, 256, 1, 1, 0, False, False, False, 9, 64),     # grouped (Winograd-style) weights, K = 256
])
def test_bstream_kernel_is_bit_identical_to_the_tiled_kernel(ops, case, monkeypatch):
    """The B-streamed persistent GEMM (csrc/conv_bstream.hip: the weight slice streams through three LDS stages with one
    barrier per 64 k, A fragments straight from global memory through an asm register ring that runs across tile
    boundaries) keeps the tiled kernel's accumulation order and shares its prologue / epilogue code: IDENTICAL BITS,
    and both match a torch fp32 convolution."""
    cin, h, w, cout, k, s, p, res, msk, pro, groups, n = case
    g = torch.Generator().manual_seed(13)
    monkeypatch.setenv('HND_BRES', '0')
    if groups > 1:
        tiles_pad = (n * h * w + 255) // 256 * 256
        x = torch.randn(1, 1, groups * tiles_pad, cin, generator=g).to(DEV)
        y = torch.empty(1, 1, groups * tiles_pad, cout, device=DEV)
        ws = [torch.randn(cout, cin, 1, 1, generator=g).to(DEV) / cin ** 0.5 for _ in range(groups)]
        pks = [ops.pack_weights(wt) for wt in ws]
        pk = ops.PackedWeight.__new__(ops.PackedWeight)
        pk.buf = torch.cat([q.buf for q in pks])
        pk.kdim, pk.rows, pk.chan_pad, pk.chan_real = pks[0].kdim, cout, cin, cin
    else:
        x = torch.randn(n, h, w, cin, generator=g).to(DEV)
        oh, ow = ops.conv_out_size(h, k, s, p), ops.conv_out_size(w, k, s, p)
        y = torch.empty(n, oh, ow, cout, device=DEV)
        wt = torch.randn(cout, cin, k, k, generator=g).to(DEV) / (cin * k * k) ** 0.5
        pk = ops.pack_weights(wt)
    sc, sh = torch.rand(cout, generator=g).to(DEV) + 0.5, torch.randn(cout, generator=g).to(DEV)
    r = torch.randn(y.shape, generator=g).to(DEV) if res else None
    mk = (torch.rand(y.shape, generator=g).to(DEV) - 0.3).clamp_min(0) if msk else None
    ps = (torch.rand(cin, generator=g).to(DEV) + 0.5) if pro else None
    pb = torch.randn(cin, generator=g).to(DEV) if pro else None
    outs, variants = {}, {}
    for mode in ('0', 'all'):
        monkeypatch.setenv('HND_BSTREAM', '0' if mode == '0' else '1')
        monkeypatch.setenv('HND_DEBUG_PICKER', '' if mode == '0' else 'bstream_all')
        if groups > 1:
            l = ops.conv_desc(x, pk, y, kh=1, kw=1, oh=1, ow=groups * tiles_pad, sh=1, dh=1, bh=0, sw=1, dw=1, bw=0,
                              cout=cout)
            l.desc.w_group_rows, l.desc.w_group_stride = tiles_pad, pks[0].buf.numel()
        else:
            l = ops.conv_forward(x, pk, y, k, s, p, epi_scale=sc, epi_shift=sh, res1=r, mask=mk, relu=not msk,
                                 pro_scale=ps, pro_shift=pb, pro_relu=pro)
        l.refresh_variant()
        y.fill_(float('nan'))
        l.run()
        ops.sync_check()
        outs[mode], variants[mode] = y.clone(), l.variant
    assert variants['0'].startswith('igemm'), variants
    assert variants['all'] == ('bstream_128' if cout % 128 == 0 else 'bstream_64'), variants
    assert not bool(torch.isnan(outs['all']).any())
    assert torch.equal(outs['0'], outs['all']), float((outs['0'] - outs['all']).abs().max())
    # the work-balancing relay (a workgroup parks a tile's accumulators, its neighbour continues the k chain) is on
    # whenever every workgroup gets at least one tile; it must leave its flags cleared for the next launch, and the
    # round-robin split (no workspace) must give the same bits
    bm, bn = (128, 128) if cout % 128 == 0 else (256, 64)
    tiles = -(-(y.numel() // y.shape[-1]) // bm) * (cout // bn)
    assert (l.relay is not None) == (tiles >= 256), (tiles, l.relay is None)
    if l.relay is not None:
        book = l.relay.view(torch.int32)[256 * 16384:]      # [256] flags (launch epochs), launch counter, ticket
        assert int(book[256]) == 1 and int(book[257]) == 0 and set(book[:256].tolist()) <= {0, 1}
        l.run()                                     # a second launch on the same workspace
        ops.sync_check()
        assert torch.equal(outs['0'], y)
        assert int(book[256]) == 2 and int(book[257]) == 0 and set(book[:256].tolist()) <= {0, 2}
        l.desc.relay_ws = None
        y.fill_(float('nan'))
        l.run()
        ops.sync_check()
        assert torch.equal(outs['0'], y)
    if groups == 1:
        xin = x.permute(0, 3, 1, 2)
        if pro:
            xin = torch.relu(xin * ps.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1))
        ref = F.conv2d(xin, wt, stride=s, padding=p) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        if res:
            ref = ref + r.permute(0, 3, 1, 2)
        if msk:
            ref = torch.where(mk.permute(0, 3, 1, 2) > 0, ref, torch.zeros_like(ref))
        else:
            ref = torch.relu(ref)
        got = outs['all'].permute(0, 3, 1, 2)
        assert float((got - ref).norm() / ref.norm()) < 1e-5
    else:
        for gi in (0, groups - 1):
            rows = slice(gi * tiles_pad, (gi + 1) * tiles_pad)
            ref = x[0, 0, rows] @ ws[gi].view(cout, cin).t()
            assert float((outs['all'][0, 0, rows] - ref).norm() / ref.norm()) < 1e-5


@pytest.mark.parametrize('case', [
    # cin, h, w, cout, residual, n, forced kernel (HND_BRES / HND_BSTREAM / HND_DEBUG_PICKER)
    (128, 50, 84, 512, True, 4, {}),                                  # layer2 conv3: bres2 + residual (own store path)
    (256, 37, 41, 1024, True, 3, {}),                                 # ragged rows: the checked path writes nibbles too
    (512, 25, 42, 2048, True, 4, {}),                                 # K = 512: the 8-wave kernel (shared epilogue)
    (128, 33, 47, 256, False, 2, {'HND_BRES': '0'}),                  # tiled kernel, picker restricted to 128 columns
    (128, 33, 47, 256, True, 2, {'HND_BRES': '0', 'HND_DEBUG_PICKER': 'igemm_tile=3'}),      # 64 x 64 asked for -> 64 x 128
    (1024, 25, 42, 512, True, 8, {'HND_BRES': '0', 'HND_DEBUG_PICKER': 'bstream_all'}),      # B-streamed kernel
])
def test_relu_mask_nibbles_written_by_the_forward_and_applied_by_the_data_gradient(ops, case, monkeypatch):
    """hnd_conv_desc.mask_out / mask_bits (ABI 8): the forward epilogue writes [y > 0] as one byte per pixel and four
    channels next to y; a launch that takes those nibbles as its mask gives EXACTLY the bits of the same launch masked by
    the fp32 tensor y -- on every kernel behind hnd_conv2d_igemm, including 64-column tiles (half a nibble per lane)."""
    cin, h, w, cout, res, n, env = case
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, h, w, cin, generator=g).to(DEV)
    wt = torch.randn(cout, cin, 1, 1, generator=g).to(DEV) / cin ** 0.5
    pk = ops.pack_weights(wt)
    r = torch.randn(n, h, w, cout, generator=g).to(DEV) if res else None
    y = torch.empty(n, h, w, cout, device=DEV)
    bits = ops.mask_nibbles_like(y)
    bits.fill_(0xAA)
    l = ops.conv_forward(x, pk, y, 1, 1, 0, res1=r, relu=True, mask_out=bits)
    l.run()
    ops.sync_check()
    want = (y > 0).view(n, h, w, cout // 4, 4).to(torch.uint8)
    want = want[..., 0] | (want[..., 1] << 1) | (want[..., 2] << 2) | (want[..., 3] << 3)
    assert torch.equal(bits, want), (l.variant, int((bits != want).sum()))
    assert 0.2 < float((y > 0).float().mean()) < 0.8
    # consumer: a data-gradient-like launch masked by y (fp32) vs by its nibbles, on the default and on the tiled kernels
    for env2 in ({}, {'HND_BRES': '0', 'HND_BSTREAM': '0'},
                 {'HND_BRES': '0', 'HND_BSTREAM': '0', 'HND_DEBUG_PICKER': 'igemm_tile=3'}):
        for k in ('HND_BRES', 'HND_BSTREAM', 'HND_DEBUG_PICKER'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env2.items():
            monkeypatch.setenv(k, v)
        outs = []
        for kw in ({'mask': y}, {'mask_bits': bits}):
            z = torch.full_like(y, float('nan'))
            l2 = ops.conv_forward(x, pk, z, 1, 1, 0, res1=r, **kw)
            l2.run()
            ops.sync_check()
            outs.append((z.clone(), l2.variant))
        assert torch.equal(outs[0][0], outs[1][0]), (outs[0][1], outs[1][1])
        assert not bool(torch.isnan(outs[1][0]).any())


def test_affine_relu_writes_mask_nibbles(ops):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 17, 23, 256, generator=g).to(DEV)
    sc, sh = (torch.rand(256, generator=g) + 0.5).to(DEV), torch.randn(256, generator=g).to(DEV)
    y, bits = torch.empty_like(x), ops.mask_nibbles_like(x)
    ops.affine_relu(x, sc, sh, y, True, mask_out=bits)
    ops.sync_check()
    want = (y > 0).view(3, 17, 23, 64, 4).to(torch.uint8)
    assert torch.equal(bits, want[..., 0] | (want[..., 1] << 1) | (want[..., 2] << 2) | (want[..., 3] << 3))
    assert torch.allclose(y, torch.relu(x * sc + sh), rtol=1e-6, atol=1e-6)      # (the kernel's scale-shift is one fma)


def test_bstream_relay_timeout_is_loud_and_does_not_poison_the_workspace(ops, monkeypatch):
    """a relay wait that gives up must not pass for a correct launch, and a writer that arrives late must not
    leave state a later launch on the same workspace would trust.  HND_BSTREAM_DBG=2 makes every head keep its flag
    down, so every tail wait times out (HND_BSTREAM_SPIN bounds it to milliseconds): the sticky error word is raised --
    hnd_sync_check fails, the next B-streamed launch is refused -- and, once acknowledged, the SAME workspace gives the
    tiled kernel's bits again without being re-zeroed."""
    from hnd_ghnd_object_detectors_amd import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(3)
    n, h, w, cin, cout = 16, 50, 84, 1024, 256
    x = torch.randn(n, h, w, cin, generator=g).to(DEV)
    wt = torch.randn(cout, cin, 1, 1, generator=g).to(DEV) / 32.0
    pk = ops.pack_weights(wt)
    monkeypatch.setenv('HND_BRES', '0')
    monkeypatch.setenv('HND_BSTREAM', '0')
    ref = torch.empty(n, h, w, cout, device=DEV)
    ops.conv_forward(x, pk, ref, 1, 1, 0, relu=True).run()
    monkeypatch.delenv('HND_BSTREAM')
    monkeypatch.setenv('HND_DEBUG_PICKER', 'bstream_all')
    y = torch.empty_like(ref)
    l = ops.conv_forward(x, pk, y, 1, 1, 0, relu=True)
    assert l.variant == 'bstream_128' and l.relay is not None
    L.hnd_relay_timeouts(1)
    l.run()
    ops.sync_check()
    assert torch.equal(y, ref) and L.hnd_relay_timeouts(0) == 0
    monkeypatch.setenv('HND_BSTREAM_DBG', '2')
    monkeypatch.setenv('HND_BSTREAM_SPIN', '2000')
    l.run()                                                 # every tail gives up
    with pytest.raises(RuntimeError, match='relay time-out'):
        ops.sync_check()
    monkeypatch.delenv('HND_BSTREAM_DBG')
    with pytest.raises(RuntimeError, match='relay time-out'):
        l.run()                                             # refused at launch until acknowledged
    assert L.hnd_relay_timeouts(1) == 1 and L.hnd_relay_timeouts(0) == 0
    monkeypatch.delenv('HND_BSTREAM_SPIN')
    y.fill_(float('nan'))
    l.run()                                                 # same workspace, not re-zeroed
    ops.sync_check()
    assert torch.equal(y, ref)


@pytest.mark.parametrize('h,w,scale', [(480, 640, 1.25), (800, 1333, 1.0), (375, 500, 2.1333333333333333),
                                       (1333, 800, 0.8401500375093773), (37, 53, 1.7027027027027026), (5, 7, 0.5)])
def test_gt_mask_nearest_resize_equals_torch_bytes(h, w, scale):
    """reference src/models/org/rcnn.py:54-57: interpolate(mask[None].float(), scale_factor=s)[0].byte() -- the HIP
    kernel moves uint8 and must reproduce ATen's nearest source index (floorf(dst * (float)(1 / s))) byte for byte"""
    from hnd_ghnd_object_detectors_amd.models.org.rcnn import resize_masks_nearest
    g = torch.Generator().manual_seed(5)
    m = (torch.rand(3, h, w, generator=g) < 0.4).to(torch.uint8) * torch.randint(1, 255, (3, 1, 1), generator=g).to(torch.uint8)
    ref = torch.nn.functional.interpolate(m[None].float(), scale_factor=scale)[0].byte()
    got = resize_masks_nearest(m.to(DEV), scale)
    assert got.dtype == torch.uint8 and tuple(got.shape) == tuple(ref.shape)
    assert torch.equal(got.cpu(), ref)
    assert resize_masks_nearest(torch.zeros(0, h, w, dtype=torch.uint8, device=DEV), scale).shape[0] == 0


@pytest.mark.parametrize('n,h,w', [(2, 800, 1344), (1, 75, 133), (3, 64, 96), (1, 1248, 1120)])
def test_stem_conv_from_lds_patch_is_bit_identical_and_matches_torch(ops, n, h, w, monkeypatch):
    """csrc/conv_stem.hip: the 7x7 stride-2 stem reads its MFMA A-fragments from an LDS-staged input patch instead of
    49 scattered global reads per pixel; same k order and epilogue expression as the generic kernel -> identical bits,
    incl. partial edge tiles and odd sizes; and both equal a torch fp32 convolution + FrozenBN + ReLU."""
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) * (1.0 / 147 ** 0.5)
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    xd = nhwc(x, 4)
    pk = ops.pack_weights(wt.to(DEV).contiguous(), chan_pad=4)
    oh, ow = ops.conv_out_size(h, 7, 2, 3), ops.conv_out_size(w, 7, 2, 3)
    outs, variants = {}, {}
    for mode in ('0', '1'):
        monkeypatch.setenv('HND_STEM7', mode)
        y = torch.full((n, oh, ow, 64), float('nan'), device=DEV)
        l = ops.conv_forward(xd, pk, y, 7, 2, 3, epi_scale=sc.to(DEV), epi_shift=sh.to(DEV), relu=True)
        l.run()
        ops.sync_check()
        outs[mode], variants[mode] = y, l.variant
    assert variants == {'0': 'igemm_c4_128x64', '1': 'stem7_lds'}, variants
    assert not bool(torch.isnan(outs['1']).any()) and torch.equal(outs['0'], outs['1'])
    ref = torch.relu(F.conv2d(x, wt, stride=2, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    assert relerr(nchw(outs['1']), ref) < 1e-5


def test_stem_kernel_keeps_the_fourth_channel_of_a_genuine_four_channel_conv(ops, monkeypatch):
    """the stem kernel does not issue the MFMAs of channel 3 when the packed weights of that channel are all zero (the
    3-channel image stored NHWC4); a 7x7 stride-2 conv with FOUR real input channels takes the same kernel and must still
    count the fourth: identical bits to the generic kernel, and equal to torch"""
    g = torch.Generator().manual_seed(22)
    n, h, w = 2, 75, 133
    x = torch.randn(n, 4, h, w, generator=g)
    wt = torch.randn(64, 4, 7, 7, generator=g) * (1.0 / 196 ** 0.5)
    xd = nhwc(x, 4)
    pk = ops.pack_weights(wt.to(DEV).contiguous(), chan_pad=4)
    oh, ow = ops.conv_out_size(h, 7, 2, 3), ops.conv_out_size(w, 7, 2, 3)
    outs = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('HND_STEM7', mode)
        y = torch.full((n, oh, ow, 64), float('nan'), device=DEV)
        l = ops.conv_forward(xd, pk, y, 7, 2, 3)
        l.run()
        ops.sync_check()
        outs[mode] = (y, l.variant)
    assert outs['1'][1] == 'stem7_lds' and torch.equal(outs['0'][0], outs['1'][0])
    assert relerr(nchw(outs['1'][0]), F.conv2d(x, wt, stride=2, padding=3)) < 1e-5


@pytest.mark.parametrize('n,h,w', [(2, 800, 1344), (1, 75, 133), (3, 64, 96)])
def test_stem_weight_gradient_from_lds_patch(ops, n, h, w, monkeypatch):
    """csrc/conv_stem.hip stem7_wgrad_kernel: dW of the 7x7 stride-2 stem with dy transposed into LDS and the input
    patch beside it, accumulator resident across a workgroup's tiles, slabs summed in fixed order -- against autograd,
    against the generic split-K kernel, and bit-reproducible."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(n, 3, h, w, generator=g)
    wt = (torch.randn(64, 3, 7, 7, generator=g) * (1.0 / 147 ** 0.5)).requires_grad_(True)
    out = F.conv2d(x, wt, stride=2, padding=3)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    xd, dyd = nhwc(x, 4), nhwc(dy)
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('HND_STEM7', mode)
        dw = torch.full((64, 3, 7, 7), float('nan'), device=DEV)
        l = ops.conv_wgrad(xd, dyd, dw, 7, 2, 3)
        outs = []
        for _ in range(2):
            l.run()
            ops.sync_check()
            outs.append(dw.cpu().clone())
        assert torch.equal(outs[0], outs[1])                     # fixed-order reductions
        res[mode] = (outs[0], l.variant)
    assert res['0'][1] == 'wgrad_m64' and res['1'][1] == 'stem7_wgrad'
    assert relerr(res['1'][0], wt.grad) < 2e-5, relerr(res['1'][0], wt.grad)
    assert relerr(res['1'][0], res['0'][0]) < 2e-5


def test_stem_weight_gradient_with_four_real_input_channels(ops, monkeypatch):
    """with cin_real == 3 the stem's weight-gradient kernel enumerates its columns channel-major and skips the pad channel
    (10 column tiles); a conv with FOUR real input channels keeps the tap-major 13 tiles and every column"""
    g = torch.Generator().manual_seed(32)
    n, h, w = 2, 75, 133
    x = torch.randn(n, 4, h, w, generator=g)
    wt = (torch.randn(64, 4, 7, 7, generator=g) * (1.0 / 196 ** 0.5)).requires_grad_(True)
    out = F.conv2d(x, wt, stride=2, padding=3)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    monkeypatch.setenv('HND_STEM7', '1')
    dw = torch.full((64, 4, 7, 7), float('nan'), device=DEV)
    l = ops.conv_wgrad(nhwc(x, 4), nhwc(dy), dw, 7, 2, 3)
    l.run()
    ops.sync_check()
    assert l.variant == 'stem7_wgrad' and relerr(dw.cpu(), wt.grad) < 2e-5, (l.variant, relerr(dw.cpu(), wt.grad))


def test_jpeg_codec_and_data_logger_follow_the_reference(tmp_path):
    """structure/transformer.py JpegCompressor / JpegDecompressor / DataLogger (reference :58-128): the 3-channel
    bottleneck is quantised by the HIP codec (byte-exact to myutils' quantize_tensor), written as a JPEG with PIL and
    de-quantised with the stored scale / zero point -- identical bytes in give an identical file and an identical
    reconstruction to the restated reference chain on the CPU; DataLogger records the reference's three sizes."""
    import numpy as np
    from PIL import Image
    from oracle import myutils_r as MR
    from hnd_ghnd_object_detectors_amd.structure import transformer as T
    g = torch.Generator().manual_seed(3)
    z = torch.randn(1, 3, 51, 68, generator=g) * 2.0 + 0.3
    comp = T.JpegCompressor(jpeg_quality=90, tmp_dir_path=str(tmp_path / 'jpg'))
    dec = T.JpegDecompressor(tmp_dir_path=str(tmp_path / 'jpg'), target_dim=4)
    (path, qz), _ = comp(z.to(DEV), None)
    out, _ = dec((path, qz), None)
    # the reference chain on the CPU
    rq = MR.quantize_tensor(z.squeeze(0))
    ref_file = str(tmp_path / 'ref.jpg')
    Image.fromarray(rq.tensor.permute(1, 2, 0).numpy()).save(ref_file, format='jpeg', quality=90)
    assert open(path, 'rb').read() == open(ref_file, 'rb').read()
    pix = torch.from_numpy(np.asarray(Image.open(ref_file).convert('RGB'))).permute(2, 0, 1).float().div(255)
    ref = (rq.scale * (pix * 255.0 - rq.zero_point)).unsqueeze(0)
    assert tuple(out.shape) == (1, 3, 51, 68) and torch.equal(out.cpu(), ref)
    passthrough, _ = comp(torch.zeros(2, 5, 4, 4, device=DEV), None)            # not a 3-channel image: untouched
    assert isinstance(passthrough, torch.Tensor)
    tr = T.get_bottleneck_transformer({'order': ['jpeg_compressor', 'jpeg_decompressor'], 'components': {
        'jpeg_compressor': {'params': {'jpeg_quality': 90, 'tmp_dir_path': str(tmp_path / 'j2')}},
        'jpeg_decompressor': {'params': {'tmp_dir_path': str(tmp_path / 'j2'), 'target_dim': 4}}}})
    again, _ = tr(z.to(DEV), None)
    assert torch.equal(again.cpu(), ref)
    log = T.DataLogger()
    same, _ = log(z.to(DEV), None)
    log(None, None)
    sizes, fp16, quant, shapes = log.get_data()
    assert same.data_ptr() == same.data_ptr() and shapes == [[3, 51, 68], [0, 0, 0]] and sizes[1] == 0.0
    assert abs(sizes[0] - MR.get_binary_object_size(z)) < 0.01                  # KB of the pickled fp32 tensor
    assert abs(fp16[0] - MR.get_binary_object_size(z.short())) < 0.01
    assert 3 * 51 * 68 / 1024 < quant[0] < 3 * 51 * 68 / 1024 + 1.5             # uint8 payload + a small header
    log.clear()
    assert log.get_data() == ([], [], [], [])


def test_batched_box_rescale_equals_the_reference_expression_bitwise(ops):
    """reference src/models/org/rcnn.py:50-53 resize_boxes: stack(xmin * rw, ymin * rh, xmax * rw, ymax * rh) per image
    (fp32 tensor times a Python float) -- hnd_scale_boxes does the whole batch in one launch and must give the same
    bits, including an image without boxes and a non-contiguous view."""
    from hnd_ghnd_object_detectors_amd.models.org.rcnn import resize_boxes
    g = gen(9)
    items, refs = [], []
    for k, (oh, ow, nh, nw) in zip((5, 0, 33, 1, 700), ((480, 640, 800, 1066), (375, 500, 800, 1066), (640, 427, 1199, 800),
                                                        (333, 500, 800, 1201), (500, 500, 1333, 1333))):
        bx = (torch.rand(k, 4, generator=g) * 400).to(DEV)
        if k == 33:
            bx = (torch.rand(k, 8, generator=g) * 400).to(DEV)[:, ::2]          # a strided view
        items.append((bx, float(nw) / float(ow), float(nh) / float(oh)))
        refs.append(resize_boxes(bx, (oh, ow), (nh, nw)))
    outs = ops.scale_boxes(items)
    ops.sync_check()
    for out, ref in zip(outs, refs):
        assert out.shape == ref.shape and torch.equal(out, ref)


def test_batched_transform_equals_the_per_image_launches_bitwise(ops):
    """hnd_transform_images (the whole batch in one launch per source type) against hnd_transform_image /
    hnd_transform_image_u8 image by image: identical bits for float, uint8 HWC / CHW and flipped sources of different
    sizes.  (A first version with both source types behind a run-time branch was contracted into different fma chains
    by hipcc -- 1 ulp -- which the neural filter's training goldens amplified past their tolerance.)"""
    g = gen(12)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    items = []
    for h, w, u8, hwc, flip in [(60, 80, False, False, False), (64, 48, True, True, False), (50, 70, True, True, True),
                                (33, 47, True, False, True), (40, 40, True, False, False), (57, 91, False, False, False)]:
        if u8:
            shape = (h, w, 3) if hwc else (3, h, w)
            src = (torch.rand(*shape, generator=g) * 255).to(torch.uint8)
        else:
            src = torch.rand(3, h, w, generator=g)
        scale = 1.37
        items.append((src.to(DEV), u8, hwc, flip, ops.interp_out_size(h, scale), ops.interp_out_size(w, scale),
                      1.0 / scale, 1.0 / scale))
    a = torch.full((len(items), 128, 160, 4), float('nan'), device=DEV)
    b = torch.full((len(items), 128, 160, 4), float('nan'), device=DEV)
    ops.transform_images(items, a, mean, std)
    for i, (src, u8, hwc, flip, oh, ow, rh, rw) in enumerate(items):
        if u8:
            ops.transform_image_u8(src, b, i, oh, ow, rh, rw, mean, std, hwc, flip)
        else:
            ops.transform_image(src, b, i, oh, ow, rh, rw, mean, std)
    ops.sync_check()
    assert not bool(torch.isnan(a).any()) and torch.equal(a, b)


@pytest.mark.parametrize('c,n,h,w,tile', [(128, 2, 38, 50, 6), (256, 3, 24, 31, 4), (128, 2, 100, 168, 6)])
def test_winograd_output_transform_writes_the_relu_mask_nibbles_of_the_values_it_stores(ops, c, n, h, w, tile):
    """hnd_wino_output(mask_out) (ABI 12): the F(4x4) / F(6x6) output transforms own two channels of a pixel per thread; the
    even thread of a pair writes the byte [stored value > 0] of four channels -- byte-exact against the stored tensor,
    edge tiles included, and y itself unchanged by the option (round 6: conv3's data gradient reads these nibbles)"""
    g = torch.Generator().manual_seed(3 + c + h)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5
    es, eb = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    xd, wd = nhwc(x), wt.to(DEV).contiguous()
    ww = ops.WinoWeights(wd, False, tile)
    nv, nm = ops.WinoConv.scratch_elems(n, h, w, c, c, tile)
    v, m = torch.empty(nv, device=DEV), torch.empty(nm, device=DEV)
    y0 = torch.full((n, h, w, c), float('nan'), device=DEV)
    y1 = torch.full((n, h, w, c), float('nan'), device=DEV)
    bits = torch.full((n, h, w, c // 4), 255, dtype=torch.uint8, device=DEV)
    kw = dict(epi_scale=es.to(DEV), epi_shift=eb.to(DEV), relu=True)
    ops.WinoConv(xd, ww, y0, v, m, **kw).run()
    ops.WinoConv(xd, ww, y1, v, m, mask_out=bits, **kw).run()
    ops.sync_check()
    assert torch.equal(y0, y1) and not bool(torch.isnan(y1).any())
    yv = y1.view(n, h, w, c // 4, 4) > 0
    want = (yv[..., 0].to(torch.uint8) | (yv[..., 1].to(torch.uint8) << 1) | (yv[..., 2].to(torch.uint8) << 2)
            | (yv[..., 3].to(torch.uint8) << 3))
    assert torch.equal(bits, want)
    assert 0.2 < float((y1 > 0).float().mean()) < 0.8

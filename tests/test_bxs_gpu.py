"""The B-STREAMED fp32 emulation kernel (csrc/conv_bxs.hip, hnd_conv_desc.w_bf16x3s; round 6): the
launches the B-resident emulation kernel cannot take -- taps, long K, strided outputs, BatchNorm on load, statistics.

Every class is held to an fp64 reference BESIDE the native kernel on the same operands (error at most 1.5x the native
one's, like tests/test_bx3_gpu.py), twice for reproducible bits, with and without the stream-K relay workspace."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    from hnd_ghnd_object_detectors_amd import ops as O
    return O


@pytest.fixture(autouse=True)
def emulated_family(ops):
    """this file holds the EMULATED rounding family: the round-6 default (choice by layer), also when the suite runs under
    HND_BF16X3=0 (tests/test_ops_gpu.py is the native family's file the other way round)"""
    with ops.emulation('policy'):
        yield


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def _rel(a, ref):
    return float((a.cpu().double() - ref).norm() / ref.norm())


def _pair(ops, build):
    """build(): a launch made under the current emulation mode -> (native launch, emulated launch)"""
    with ops.emulation('off'):
        l0 = build()
    with ops.emulation('force'):
        l1 = build()
    assert not l0.variant.startswith('bx'), l0.variant
    assert l1.variant.startswith('bxs'), l1.variant
    return l0, l1


@pytest.mark.parametrize('cin,cout,n,h,w,k,stride,pad', [
    (128, 128, 4, 100, 168, 3, 2, 1),        # layer2.0.conv2
    (256, 256, 4, 50, 84, 3, 2, 1),          # layer3.0.conv2
    (512, 512, 8, 25, 42, 3, 2, 1),          # layer4.0.conv2: 13 x 21 outputs, tiles fewer than workgroups x 2
    (128, 64, 3, 37, 53, 3, 1, 1),           # 256 x 64 tile, rows with a tail
    (2048, 512, 16, 25, 42, 1, 1, 0),        # layer4.x.conv1: long K, no taps
    (1024, 2048, 4, 50, 84, 1, 2, 0),        # layer4.0.downsample: stride 2
    (64, 128, 2, 61, 77, 2, 1, 1),           # a head conv: two taps per 128-k iteration
])
def test_bxs_conv_against_fp64_beside_the_native_kernel(ops, cin, cout, n, h, w, k, stride, pad):
    g = torch.Generator().manual_seed(3 + cin + cout + k)
    x = torch.randn(n, cin, h, w, generator=g) * torch.exp2(torch.randn(n, 1, h, w, generator=g) * 2)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    es, eb = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), None, stride, pad) * es.double()[None, :, None, None]
                 + eb.double()[None, :, None, None]).permute(0, 2, 3, 1)
    xd, pk = _nhwc(x), ops.pack_weights(wt.to(DEV).contiguous())
    pk.bx3 = pk.used3 = None                      # (tap-free shapes the B-resident kernel would take first: no image, none made)
    oh, ow = ref.shape[1], ref.shape[2]
    ys = [torch.full((n, oh, ow, cout), float('nan'), device=DEV) for _ in range(4)]
    kw = dict(epi_scale=es.to(DEV), epi_shift=eb.to(DEV), relu=True)
    it = iter(ys)
    l0, l1 = _pair(ops, lambda: ops.conv_forward(xd, pk, next(it), k, stride, pad, **kw))
    l0.run()
    l1.run()
    with ops.emulation('force'):
        l2 = ops.conv_forward(xd, pk, ys[2], k, stride, pad, **kw)
        l3 = ops.conv_forward(xd, pk, ys[3], k, stride, pad, **kw)
    l2.run()
    l3.desc.relay_ws = None                       # tiles round-robin, no stream-K relay: the same k chains
    l3.run()
    ops.sync_check()
    e0, e1 = _rel(ys[0], ref), _rel(ys[1], ref)
    assert not bool(torch.isnan(ys[1]).any())
    assert e1 < 1e-6 and e1 <= 1.5 * e0 + 1e-8, (e1, e0, l1.variant)
    assert torch.equal(ys[2], ys[1]) and torch.equal(ys[3], ys[1])
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation, B-streamed, %dx%d s%d %d -> %d @%dx%d, %s%s] rel-L2 vs fp64 %.2e (native %s %.2e)'
                    % (k, k, stride, cin, cout, h, w, l1.variant, ' + relay' if l1.relay is not None else '', e1, l0.variant, e0))


@pytest.mark.parametrize('cin,cout,n,h,w,pad,bwd', [(64, 256, 2, 101, 169, 1, False), (64, 64, 2, 100, 168, 1, False),
                                                     (128, 256, 2, 67, 91, 0, False), (256, 64, 2, 90, 122, 1, True),
                                                     (64, 128, 3, 70, 90, 0, True)])
def test_bxs_head_conv_with_batchnorm_on_load_and_statistics(ops, cin, cout, n, h, w, pad, bwd):
    """a direct 2x2 head conv as the student runs it (reference src/models/mimic/resnet_layer.py:43-62): train-mode
    BatchNorm + ReLU of the previous layer applied on load (padding stays an exact zero of the NORMALISED tensor), and in the
    epilogue the per-128-row partial sums hnd_bn_finalize needs (sum, sum of squares of the stored output) -- or, for the
    data gradient that produces g, the BatchNorm-backward partials (sum d, sum d * xhat; hnd_conv_desc.bwd_x)."""
    g = torch.Generator().manual_seed(21 + cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 2, 2, generator=g) / (cin * 4) ** 0.5
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5
    a = F.relu(x.double() * ps.double()[None, :, None, None] + pb.double()[None, :, None, None])
    ref = F.conv2d(a, wt.double(), None, 1, pad).permute(0, 2, 3, 1)
    oh, ow = ref.shape[1], ref.shape[2]
    m = n * oh * ow
    xd, pk = _nhwc(x), ops.pack_weights(wt.to(DEV).contiguous())
    kw = dict(pro_scale=ps.to(DEV), pro_shift=pb.to(DEV), pro_relu=True)
    xr = torch.randn(n, oh, ow, cout, generator=g)
    bsc, bsh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    bmu, brs = torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5
    outs = []

    def build():
        y = torch.full((n, oh, ow, cout), float('nan'), device=DEV)
        st = torch.full((ops.stats_tiles(m), 2, cout), float('nan'), device=DEV)
        outs.append((y, st))
        extra = dict(bwd_stats=(xr.to(DEV), bsc.to(DEV), bsh.to(DEV), bmu.to(DEV), brs.to(DEV), True)) if bwd else {}
        return ops.conv_forward(xd, pk, y, 2, 1, pad, stats=st, **kw, **extra)

    l0, l1 = _pair(ops, build)
    l0.run()
    l1.run()
    ops.sync_check()
    (y0, s0), (y1, s1) = outs
    e0, e1 = _rel(y0, ref), _rel(y1, ref)
    assert e1 < 1e-6 and e1 <= 1.5 * e0 + 1e-8, (e1, e0)
    # the statistics describe the values THIS kernel stored
    yv = y1.cpu().double().view(m, cout)
    if bwd:
        xv = xr.double().view(m, cout)
        dd = torch.where(xv * bsc.double() + bsh.double() > 0, yv, torch.zeros((), dtype=torch.float64))
        want = torch.stack([dd.sum(0), (dd * (xv - bmu.double()) * brs.double()).sum(0)])
    else:
        want = torch.stack([yv.sum(0), (yv * yv).sum(0)])
    got1, got0 = s1.cpu().double().sum(0), s0.cpu().double().sum(0)
    scale = torch.stack([yv.abs().sum(0), (yv * yv).sum(0)]) + 1e-30 if not bwd else \
        torch.stack([dd.abs().sum(0), (dd * (xv - bmu.double()) * brs.double()).abs().sum(0)]) + 1e-30
    es1, es0 = float(((got1 - want).abs() / scale).max()), float(((got0 - want).abs() / scale).max())
    assert not bool(torch.isnan(s1).any()) and es1 < 2e-6 and es1 <= 2 * es0 + 2e-7, (es1, es0)
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation, B-streamed, head 2x2 %d -> %d pad %d, BN + ReLU on load, %s] y rel-L2 vs fp64 %.2e '
                    '(native %.2e); partial sums vs fp64 of the stored values %.1e (native %.1e)'
                    % (cin, cout, pad, 'backward sums' if bwd else 'statistics', e1, e0, es1, es0))


@pytest.mark.parametrize('cin,cout,n,h,w,k,stride,pad', [(128, 128, 4, 100, 168, 3, 2, 1), (256, 256, 4, 50, 84, 3, 2, 1),
                                                         (256, 512, 4, 100, 168, 1, 2, 0)])
def test_bxs_stride2_data_gradients_parity_launches_with_masks(ops, cin, cout, n, h, w, k, stride, pad):
    """data gradient of a stride-2 conv: one launch per output parity over its tap subset, strided output, the ReLU mask
    of the target as an fp32 tensor (3x3: [a1 > 0]) or accumulating into the conv1 gradient (downsample: res1 = dx)"""
    g = torch.Generator().manual_seed(55 + cin + k)
    oh, ow = ops.conv_out_size(h, k, stride, pad), ops.conv_out_size(w, k, stride, pad)
    dy = torch.randn(n, cout, oh, ow, generator=g)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cout * k * k / stride ** 2) ** 0.5)
    act = torch.randn(n, cin, h, w, generator=g)
    base = torch.randn(n, cin, h, w, generator=g)
    refg = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), dy.double(), stride=stride, padding=pad)
    if k == 1:
        refg = refg + base.double()
    ref = torch.where(act.double() > 0, refg, torch.zeros((), dtype=torch.float64)).permute(0, 2, 3, 1)
    dyd, wd, ad = _nhwc(dy), wt.to(DEV).contiguous(), _nhwc(act)
    res = {}
    for emu in (False, True, True):
        dx = _nhwc(base) if k == 1 else torch.full((n, h, w, cin), float('nan'), device=DEV)
        with ops.emulation('force' if emu else 'off'):
            ls, pks = ops.conv_dgrad(dyd, wd, dx, k, stride, pad, accumulate=(k == 1), mask=ad)
        assert all(l.variant.startswith('bxs') == emu for l in ls), [l.variant for l in ls]
        for l in ls:
            l.run()
        ops.sync_check()
        res.setdefault(emu, []).append(dx)
    sel = (slice(None), slice(0, None, stride), slice(0, None, stride)) if k == 1 else (slice(None),) * 3
    e0 = _rel(res[False][0][sel], ref[sel])
    e1 = _rel(res[True][0][sel], ref[sel])
    assert e1 < 1e-6 and e1 <= 1.5 * e0 + 1e-8, (e1, e0)
    assert torch.equal(res[True][0], res[True][1])
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation, B-streamed, data gradient of %dx%d s%d %d -> %d, %d parity launches, fp32 mask] '
                    'rel-L2 vs fp64 %.2e (native %.2e)' % (k, k, stride, cin, cout, len(ls), e1, e0))


def test_bxs_randomised_shapes_against_the_native_kernel(ops):
    """seeded sweep: taps 1x1 / 2x2 / 3x3, stride 1 / 2, K = 128 ... 4608, 64 ... 512 columns, row counts with tails and
    fewer tiles than workgroups, every epilogue operand -- each beside the native kernel (every element within 3e-5 of the
    result's rms), twice for reproducible bits.  A wrong hand-counted wait shows as garbage, not as a small error."""
    import random
    rnd = random.Random(20261005)
    g = torch.Generator().manual_seed(7)
    done = 0
    for case in range(36):
        k = rnd.choice([1, 2, 3, 3])
        cin = rnd.choice([64, 128, 256, 512]) if k > 1 else rnd.choice([128, 256, 1024, 2048])
        cout = rnd.choice([64, 128, 256, 512])
        stride = rnd.choice([1, 2]) if k != 2 else 1
        pad = rnd.choice([0, 1]) if k > 1 else 0
        n = rnd.choice([1, 2, 3, 5])
        h, w = rnd.randrange(12, 90), rnd.randrange(16, 120)
        oh, ow = ops.conv_out_size(h, k, stride, pad), ops.conv_out_size(w, k, stride, pad)
        if k * k * cin % 128 != 0 or oh < 2 or ow < 2 or n * h * w * cin * 4 > 6e8:
            continue
        res = rnd.random() < 0.5
        relu = rnd.random() < 0.5
        pro = k > 1 and rnd.random() < 0.4
        mask_bits = (not pro) and cout % 4 == 0 and rnd.random() < 0.3
        x = (torch.randn(n, h, w, cin, generator=g) * 2.0).to(DEV)
        wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(DEV)
        kw = dict(relu=relu, epi_scale=(torch.rand(cout, generator=g) + 0.5).to(DEV), epi_shift=torch.randn(cout, generator=g).to(DEV))
        if res:
            kw['res1'] = torch.randn(n, oh, ow, cout, generator=g).to(DEV)
        if pro:
            kw.update(pro_scale=(torch.rand(cin, generator=g) + 0.5).to(DEV), pro_shift=torch.randn(cin, generator=g).to(DEV),
                      pro_relu=rnd.random() < 0.7)
        if mask_bits:
            kw['mask_bits'] = torch.randint(0, 16, (n, oh, ow, cout // 4), generator=g, dtype=torch.uint8).to(DEV)
        pk = ops.pack_weights(wt)
        pk.bx3 = pk.used3 = None
        outs = []
        for emu in (False, True, True):
            y = torch.full((n, oh, ow, cout), float('nan'), device=DEV)
            with ops.emulation('force' if emu else 'off'):
                l = ops.conv_forward(x, pk, y, k, stride, pad, **kw)
            assert l.variant.startswith('bxs') == emu, (case, k, cin, cout, l.variant)
            l.run()
            outs.append(y)
        ops.sync_check()
        y0, y1, y2 = outs
        rms = float(y0.double().pow(2).mean().sqrt()) + 1e-30
        err = float((y1.double() - y0.double()).abs().max()) / rms
        assert not bool(torch.isnan(y1).any()) and err < 3e-5, (case, k, cin, cout, n, h, w, stride, pad, res, relu, pro, mask_bits, err)
        assert torch.equal(y1, y2)
        done += 1
    assert done >= 20, done
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation, B-streamed] %d randomised tap / stride / epilogue cases agree with the native kernel '
                    'within 3e-5 rms per element' % done)

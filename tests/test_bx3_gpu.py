"""OPT-IN fp32 emulation on the bf16 matrix pipe (csrc/conv_bx3.hip, hnd_conv_desc.w_bf16x3).

Never the default: every other test runs native fp32 MFMA.  Here the emulation kernel is held to an fp64 reference BESIDE
the native kernel on the same operands (its error may be at most 1.5x the native one's), and one full-size reference
fixture is replayed with HND_BF16X3=1 in a subprocess under the ordinary parity bars."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    from hnd_ghnd_object_detectors_amd import ops as O
    return O


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


@pytest.mark.parametrize('cin,cout,n,h,w,stride,epi', [
    (256, 256, 4, 96, 128, 1, False),        # fpn.inner0-like (without its residual)
    (256, 128, 8, 96, 128, 1, True),         # layer2.0.conv1: FrozenBN scale / shift + ReLU
    (256, 512, 8, 192, 256, 2, True),        # layer2.0.downsample: stride 2
    (128, 64, 16, 96, 128, 1, False),        # K = 128, one 64-column slice
    (256, 1024, 8, 64, 64, 1, True),         # 16 slices per team
    (512, 256, 8, 96, 128, 1, True),         # K = 512: two passes over k, the second adds the first one's partial result
    (512, 1024, 8, 96, 128, 2, True),        # layer3.0.downsample: K = 512, stride 2
    (1024, 256, 16, 50, 84, 1, True),        # layer3.x.conv1: four passes over k
    (1024, 2048, 16, 50, 84, 2, True),       # layer4.0.downsample: four passes, stride 2, M = 16 800 = 262.5 chunks (a tail)
    (256, 256, 4, 99, 84, 1, True),          # M = 33 264: the last chunk holds 48 rows
    (128, 64, 16, 91, 93, 1, False),         # M = 135 408: 2 115 chunks + 48 rows
])
def test_bx3_1x1_conv_against_fp64_beside_the_native_kernel(ops, cin, cout, n, h, w, stride, epi):
    g = torch.Generator().manual_seed(5 + cin + cout)
    x = torch.randn(n, cin, h, w, generator=g) * torch.exp2(torch.randn(n, 1, h, w, generator=g) * 3)     # wide range
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    es = torch.rand(cout, generator=g) + 0.5 if epi else None
    eb = torch.randn(cout, generator=g) if epi else None
    ref = F.conv2d(x.double(), wt.double(), None, stride)
    if epi:
        ref = F.relu(ref * es.double()[None, :, None, None] + eb.double()[None, :, None, None])
    xd, pk = _nhwc(x), ops.pack_weights(wt.to(DEV).contiguous())
    oh, ow = ref.shape[2], ref.shape[3]
    kw = dict(epi_scale=es.to(DEV) if epi else None, epi_shift=eb.to(DEV) if epi else None, relu=epi)
    y0 = torch.full((n, oh, ow, cout), float('nan'), device=DEV)
    l0 = ops.conv_forward(xd, pk, y0, 1, stride, 0, **kw)
    assert not l0.variant.startswith('bx3')
    l0.run()
    pk.bx3 = ops.bx3_image(pk.buf, ops.round_up(cout, 64), cin, force=True)
    assert pk.bx3 is not None
    y1 = torch.full_like(y0, float('nan'))
    l1 = ops.conv_forward(xd, pk, y1, 1, stride, 0, **kw)
    assert l1.variant == 'bx3_64', l1.variant
    l1.run()
    ops.sync_check()
    want = ref.permute(0, 2, 3, 1)
    e0 = float((y0.cpu().double() - want).norm() / want.norm())
    e1 = float((y1.cpu().double() - want).norm() / want.norm())
    assert not bool(torch.isnan(y1).any())
    assert e1 < 1e-6 and e1 <= 1.5 * e0 + 1e-8, (e1, e0)
    y2 = torch.empty_like(y1)
    ops.conv_forward(xd, pk, y2, 1, stride, 0, **kw).run()
    assert torch.equal(y2, y1)                        # a fixed summation order: reproducible bits
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation, 1x1 %d -> %d @%dx%d s%d] rel-L2 vs fp64 %.2e (native fp32 MFMA %.2e)'
                    % (cin, cout, h, w, stride, e1, e0))


@pytest.mark.parametrize('cin,cout,n,h,w', [(128, 512, 8, 96, 128), (256, 1024, 8, 64, 96), (512, 2048, 16, 32, 48),
                                            (512, 2048, 16, 25, 42)])          # layer4: M = 16 800, a 32-row tail
def test_bx3_bottleneck_conv3_with_residual_relu_and_mask_nibbles(ops, cin, cout, n, h, w):
    """conv3 of a frozen Bottleneck: FrozenBN scale / shift, + identity, ReLU, and the ReLU-mask nibbles of the stored values
    (hnd_conv_desc.mask_out) -- the residual rows travel as asm loads in the ring's in-order stream"""
    g = torch.Generator().manual_seed(9 + cin)
    x = torch.randn(n, cin, h, w, generator=g).relu()
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    es, eb = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    res = torch.randn(n, cout, h, w, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double()) * es.double()[None, :, None, None] + eb.double()[None, :, None, None]
                 + res.double()).permute(0, 2, 3, 1)
    xd, rd, pk = _nhwc(x), _nhwc(res), ops.pack_weights(wt.to(DEV).contiguous())
    outs = {}
    for emu in (False, True):
        if emu:
            pk.bx3 = ops.bx3_image(pk.buf, cout, cin, force=True)
        y = torch.full((n, h, w, cout), float('nan'), device=DEV)
        bits = torch.full((n, h, w, cout // 4), 255, dtype=torch.uint8, device=DEV)
        l = ops.conv_forward(xd, pk, y, 1, 1, 0, epi_scale=es.to(DEV), epi_shift=eb.to(DEV), res1=rd, relu=True, mask_out=bits)
        assert (l.variant == 'bx3_64') == emu, l.variant
        l.run()
        ops.sync_check()
        outs[emu] = (y.cpu().double(), bits.cpu(), y)
    e0 = float((outs[False][0] - ref).norm() / ref.norm())
    e1 = float((outs[True][0] - ref).norm() / ref.norm())
    assert e1 < 1e-6 and e1 <= 1.5 * e0 + 1e-8, (e1, e0)
    yv = outs[True][2].view(n, h, w, cout // 4, 4)
    want = ((yv[..., 0] > 0).to(torch.uint8) | ((yv[..., 1] > 0).to(torch.uint8) << 1) | ((yv[..., 2] > 0).to(torch.uint8) << 2)
            | ((yv[..., 3] > 0).to(torch.uint8) << 3)).cpu()
    assert torch.equal(outs[True][1], want)                       # the nibbles describe the values this kernel stored
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation, conv3 %d -> %d + residual + ReLU + mask nibbles] rel-L2 vs fp64 %.2e (native %.2e)'
                    % (cin, cout, e1, e0))


@pytest.mark.parametrize('cin,cout,n,h,w', [(128, 512, 8, 96, 128), (256, 1024, 8, 64, 96), (512, 2048, 16, 32, 48),
                                            (512, 2048, 16, 25, 42), (1024, 256, 16, 50, 84)])
def test_bx3_masked_data_gradient_with_residual(ops, cin, cout, n, h, w):
    """conv1's data gradient in a frozen Bottleneck: W1^T g_a1 (FrozenBN scale folded into the weights) + the gradient of
    the identity path, masked by the block input's ReLU given as nibbles (hnd_conv_desc.mask_bits)"""
    g = torch.Generator().manual_seed(13 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    res = torch.randn(n, cout, h, w, generator=g)
    act = torch.randn(n, cout, h, w, generator=g)                       # the activation whose sign is the mask
    ref = torch.where(act.double() > 0, F.conv2d(x.double(), wt.double()) + res.double(),
                      torch.zeros((), dtype=torch.float64)).permute(0, 2, 3, 1)
    xd, rd, ad, pk = _nhwc(x), _nhwc(res), _nhwc(act), ops.pack_weights(wt.to(DEV).contiguous())
    av = ad.view(n, h, w, cout // 4, 4)
    bits = ((av[..., 0] > 0).to(torch.uint8) | ((av[..., 1] > 0).to(torch.uint8) << 1) | ((av[..., 2] > 0).to(torch.uint8) << 2)
            | ((av[..., 3] > 0).to(torch.uint8) << 3)).contiguous()
    errs = {}
    for emu in (False, True):
        if emu:
            pk.bx3 = ops.bx3_image(pk.buf, cout, cin, force=True)
        y = torch.full((n, h, w, cout), float('nan'), device=DEV)
        l = ops.conv_forward(xd, pk, y, 1, 1, 0, res1=rd, mask_bits=bits)
        assert (l.variant == 'bx3_64') == emu, l.variant
        l.run()
        ops.sync_check()
        errs[emu] = float((y.cpu().double() - ref).norm() / ref.norm())
    assert errs[True] < 1e-6 and errs[True] <= 1.5 * errs[False] + 1e-8, errs
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation, masked data gradient %d -> %d + residual, mask nibbles] rel-L2 vs fp64 %.2e (native %.2e)'
                    % (cin, cout, errs[True], errs[False]))


@pytest.mark.parametrize('cin,cout,n,h,w', [(512, 128, 16, 100, 168), (1024, 256, 16, 50, 84)])
def test_bx3_conv3_data_gradient_masked_by_nibbles_made_from_the_activation(ops, cin, cout, n, h, w):
    """conv3's data gradient of a frozen Bottleneck: g_a2 = [a2 > 0] W3^T g (K = 4 x planes: passes over k, the mask in the last
    one beside the partial result).  The nibbles come from hnd_relu_mask_nibbles(a2) (byte-exact against torch), the result
    equals the native kernel's fp32-mask launch within the emulation's error."""
    g = torch.Generator().manual_seed(17 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    act = torch.randn(n, cout, h, w, generator=g).relu()                # a2: post-ReLU, half of it exact zeros
    ref = torch.where(act.double() > 0, F.conv2d(x.double(), wt.double()), torch.zeros((), dtype=torch.float64)).permute(0, 2, 3, 1)
    xd, ad, pk = _nhwc(x), _nhwc(act), ops.pack_weights(wt.to(DEV).contiguous())
    bits = ops.mask_nibbles_like(ad)
    ops.relu_mask_nibbles(ad, bits)
    av = ad.view(n, h, w, cout // 4, 4) > 0
    want_bits = (av[..., 0].to(torch.uint8) | (av[..., 1].to(torch.uint8) << 1) | (av[..., 2].to(torch.uint8) << 2)
                 | (av[..., 3].to(torch.uint8) << 3))
    assert torch.equal(bits, want_bits)
    y0 = torch.full((n, h, w, cout), float('nan'), device=DEV)
    l0 = ops.conv_forward(xd, pk, y0, 1, 1, 0, mask=ad)
    l0.run()
    pk.bx3 = ops.bx3_image(pk.buf, cout, cin, force=True)
    y1 = torch.full_like(y0, float('nan'))
    l1 = ops.conv_forward(xd, pk, y1, 1, 1, 0, mask_bits=bits)
    assert l1.variant == 'bx3_64' and not l0.variant.startswith('bx3'), (l0.variant, l1.variant)
    l1.run()
    ops.sync_check()
    e0 = float((y0.cpu().double() - ref).norm() / ref.norm())
    e1 = float((y1.cpu().double() - ref).norm() / ref.norm())
    assert torch.equal(y1 == 0, y0 == 0) or float(((y1 == 0) != (y0 == 0)).float().mean()) < 1e-6
    assert e1 < 1e-6 and e1 <= 1.5 * e0 + 1e-8, (e1, e0)
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation, conv3 data gradient %d -> %d, nibbles of a2] rel-L2 vs fp64 %.2e (native, fp32 mask %.2e)'
                    % (cin, cout, e1, e0))


def test_bx3_randomised_shapes_and_epilogues_against_the_native_kernel(ops):
    """seeded sweep over what bx3_applies admits: K in {128 ... 1024} (1 - 4 passes), 64 ... 1024 columns, row counts with and
    without a tail, stride 1 / 2, every epilogue combination -- each beside the native kernel on the same operands (the two
    agree to the two roundings: every element within 3e-5 of the result's rms -- achieved 7e-6), twice for reproducible bits.  Hand-counted waits
    that were wrong for ONE combination of tail / passes / operands would show here as garbage, not as a small error."""
    import random
    rnd = random.Random(20261004)
    g = torch.Generator().manual_seed(99)
    done = 0
    for case in range(40):
        cin = rnd.choice([128, 256, 256, 512, 768, 1024])
        cout = rnd.choice([64, 128, 256, 512, 1024])
        stride = rnd.choice([1, 1, 1, 2])
        n = rnd.choice([2, 3, 5, 8])
        oh, ow = rnd.randrange(40, 120), 4 * rnd.randrange(12, 40)
        m = n * oh * ow
        nteams = 8 * (32 // (cout // 64))
        if ((m + 63) // 64) // nteams < (16 if cin > 512 else 8) or m * max(cin, cout) * 4 > 1.5e9:
            continue
        h, w = (oh - 1) * stride + 1 + rnd.randrange(0, stride), (ow - 1) * stride + 1 + rnd.randrange(0, stride)
        res = rnd.random() < 0.6
        relu = rnd.random() < 0.5
        mask_out = res and relu and rnd.random() < 0.5
        mask_bits = (res or cin > 256) and not mask_out and rnd.random() < 0.4
        epi = rnd.random() < 0.7
        x = (torch.randn(n, h, w, cin, generator=g) * 2.0).to(DEV)
        wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV)
        kw = dict(relu=relu)
        if epi:
            kw.update(epi_scale=(torch.rand(cout, generator=g) + 0.5).to(DEV), epi_shift=torch.randn(cout, generator=g).to(DEV))
        if res:
            kw['res1'] = torch.randn(n, oh, ow, cout, generator=g).to(DEV)
        bits_in = None
        if mask_bits:
            bits_in = torch.randint(0, 16, (n, oh, ow, cout // 4), generator=g, dtype=torch.uint8).to(DEV)
            kw['mask_bits'] = bits_in
        pk = ops.pack_weights(wt)
        outs = []
        for emu in (False, True, True):
            if emu and getattr(pk, 'bx3', None) is None:
                pk.bx3 = ops.bx3_image(pk.buf, ops.round_up(cout, 64), cin, force=True)
            y = torch.full((n, oh, ow, cout), float('nan'), device=DEV)
            mo = torch.full((n, oh, ow, cout // 4), 255, dtype=torch.uint8, device=DEV) if mask_out else None
            l = ops.conv_forward(x, pk, y, 1, stride, 0, mask_out=mo, **kw)
            assert l.variant.startswith('bx3') == emu, (case, cin, cout, m, stride, res, relu, mask_out, mask_bits, l.variant)
            l.run()
            outs.append((y, mo))
        ops.sync_check()
        (y0, m0), (y1, m1), (y2, m2) = outs
        rms = float(y0.double().pow(2).mean().sqrt()) + 1e-30
        err = float((y1.double() - y0.double()).abs().max()) / rms
        assert not bool(torch.isnan(y1).any()) and err < 3e-5, (case, cin, cout, m, stride, res, relu, mask_out, mask_bits, err)
        assert torch.equal(y1, y2) and (m1 is None or torch.equal(m1, m2))
        if mask_out:        # the nibbles describe the values stored beside them
            yv = y1.view(n, oh, ow, cout // 4, 4) > 0
            want = (yv[..., 0].to(torch.uint8) | (yv[..., 1].to(torch.uint8) << 1) | (yv[..., 2].to(torch.uint8) << 2)
                    | (yv[..., 3].to(torch.uint8) << 3))
            assert torch.equal(m1, want)
        done += 1
    assert done >= 12, done
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation] %d randomised shape / epilogue cases agree with the native kernel within 3e-5 rms per element' % done)


def test_bx3_fpn_lateral_with_the_upsampled_top_down_map(ops):
    """FeaturePyramidNetwork inner block (torchvision 0.4.2 ops/feature_pyramid_network.py via
    /root/reference/src/models/org/rcnn.py:399-414): 1x1 conv + bias + F.interpolate(coarser, size=..., mode='nearest')"""
    g = torch.Generator().manual_seed(31)
    n, cin, cout, h, w = 8, 256, 256, 96, 128
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bias = torch.randn(cout, generator=g)
    top = torch.randn(n, cout, h // 2, w // 2, generator=g)
    ref = (F.conv2d(x.double(), wt.double(), bias.double()) + F.interpolate(top.double(), size=(h, w), mode='nearest')).permute(0, 2, 3, 1)
    xd, td, pk = _nhwc(x), _nhwc(top), ops.pack_weights(wt.to(DEV).contiguous())
    errs = {}
    for emu in (False, True):
        if emu:
            pk.bx3 = ops.bx3_image(pk.buf, cout, cin, force=True)
        y = torch.full((n, h, w, cout), float('nan'), device=DEV)
        l = ops.conv_forward(xd, pk, y, 1, 1, 0, epi_shift=bias.to(DEV), res1=td, res1_up=True)
        assert (l.variant == 'bx3_64') == emu, l.variant
        l.run()
        ops.sync_check()
        errs[emu] = float((y.cpu().double() - ref).norm() / ref.norm())
    assert errs[True] < 1e-6 and errs[True] <= 1.5 * errs[False] + 1e-8, errs
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation, FPN lateral 256 -> 256 + upsampled top-down map] rel-L2 vs fp64 %.2e (native %.2e)'
                    % (errs[True], errs[False]))


@pytest.mark.parametrize('c,n,h,w', [(256, 8, 96, 132), (128, 8, 102, 168), (512, 16, 48, 66)])
def test_bx3_winograd_component_gemms_match_the_native_path(ops, c, n, h, w):
    """a frozen 3x3 conv through F(6x6,3x3): the 64 component GEMMs on the emulation vs on the native kernel; both held to
    the direct fp64 convolution"""
    g = torch.Generator().manual_seed(77 + c)
    x = torch.randn(n, c, h, w, generator=g).relu()
    wt = torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1).permute(0, 2, 3, 1)
    xd, wd = _nhwc(x), wt.to(DEV).contiguous()
    outs = {}
    for emu in (False, True):
        ww = ops.WinoWeights(wd, False, 6)
        if emu:
            ww.bx3 = ops.bx3_image(ww.buf, ww.rows_pad, ww.depth, ww.ncomp, ww.rows_pad * ww.depth, force=True)
        nv, nm = ops.WinoConv.scratch_elems(n, h, w, c, c, 6)
        v, m = torch.empty(nv, device=DEV), torch.empty(nm, device=DEV)
        y = torch.full((n, h, w, c), float('nan'), device=DEV)
        conv = ops.WinoConv(xd, ww, y, v, m)
        assert (conv.gemm.variant == 'bx3_64') == emu, conv.gemm.variant
        conv.run()
        ops.sync_check()
        outs[emu] = y.cpu().double()
    e0 = float((outs[False] - ref).norm() / ref.norm())
    e1 = float((outs[True] - ref).norm() / ref.norm())
    assert e1 < 2e-5 and e1 <= 1.5 * e0, (e1, e0)                 # (the Winograd transforms' own fp32 error dominates)
    from tests.conftest import record_achieved
    record_achieved('[bf16x3 emulation, F(6x6,3x3) %d ch @%dx%d] rel-L2 vs the direct fp64 conv %.2e (native %.2e)' % (c, h, w, e1, e0))


def test_full_size_reference_fixture_passes_with_the_emulation_switched_on():
    """the reference-made full-size fixture (batch 4, 3x800x1333; maps, loss terms, gradient fingerprints, parameters after
    Adam) under HND_BF16X3=1: the ordinary bars, nothing relaxed"""
    env = dict(os.environ, HND_BF16X3='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu',
                        'tests/test_model_gpu.py::test_full_size_step_matches_reference_checksums',
                        '-k', 'full_ghnd_faster_b4 or full_hnd_faster_b2'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=1500)
    text = r.stdout.decode()
    assert r.returncode == 0, text[-3000:]
    from tests.conftest import record_achieved
    for line in text.splitlines():
        if line.startswith('[full size'):
            record_achieved('[HND_BF16X3=1] ' + line)

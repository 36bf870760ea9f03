"""GPU: the HIP distillation path (through the reference-shaped module API) against
 (1) golden fixtures produced by the reference itself (tests/golden/*.npz) and
 (2) the CPU oracle on fresh seeded inputs.
Bar (north_star): feature maps and losses within 1e-3 relative fp32; gradients within 2e-3 rel-L2
(SURVEY.md hard parts: the two zero-gradient BN biases are excluded)."""
import os
import random
from collections import OrderedDict

import pytest
import torch

from oracle import hnd_oracle as O
from tests import golden_util as G
from tests import model_util as MU


def cross_entropy(logits, labels):
    from hnd_ghnd_object_detectors_amd.models.ext.classifier import cross_entropy as ce
    return ce(logits, labels)

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
TINY = ['tiny_ghnd_faster', 'tiny_hnd_faster', 'tiny_ghnd_mask', 'tiny_ghnd_keypoint', 'tiny_ghnd_faster_b6']
FEAT_TOL, LOSS_TOL, GRAD_TOL = 1e-3, 1e-3, 2e-3


def _sync_oracle(orc, student):
    """copy the HIP model's current parameters / BN buffers into an oracle so ONE step is compared in isolation."""
    sd = student.state_dict()
    with torch.no_grad():
        for k, v in orc.s.items():
            if k in sd and v.is_floating_point():
                v.copy_(sd[k].detach().cpu().to(v.dtype))


WORST_GRAD_RATIO = {'ratio': 0.0, 'name': None}      # achieved max of e_hip / e_ref over a test session (printed)


def _grad_check(name, hip, ref32, ref64, tol=None):
    """The HIP gradient must be as close to the exact (fp64) gradient as the reference's own fp32 path:
    within GRAD_TOL, or within 2x the fp32 reference's error where ReLU / max-pool decision flips make the
    problem itself ill-conditioned at tiny spatial sizes (SURVEY.md 'Gradients are ill-conditioned'; a flip is a
    discrete event, so two fp32 paths land at different multiples of the same noise scale).  The achieved worst
    ratio e_hip / e_ref among gradients beyond GRAD_TOL is recorded and printed by the calling tests."""
    ref64 = ref64.double()
    e_hip = float((hip.cpu().double() - ref64).norm() / ref64.norm())
    e_ref = float((ref32.double() - ref64).norm() / ref64.norm())
    if e_hip > (tol or GRAD_TOL) and e_ref > 0 and e_hip / e_ref > WORST_GRAD_RATIO['ratio']:
        WORST_GRAD_RATIO.update(ratio=e_hip / e_ref, name=name)
    assert e_hip <= max(tol or GRAD_TOL, 2.0 * e_ref), '%s: HIP %.2e vs fp64, reference fp32 %.2e' % (name, e_hip, e_ref)
    return e_hip


def _hooked(model, path):
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import module_util
    return module_util.get_module(model, path).__dict__['distillation_box']['output']


def _to_dev(images, targets):
    return [im.to(DEV) for im in images], [{k: v.to(DEV) for k, v in t.items()} for t in targets]


def _setup(meta):
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from hnd_ghnd_object_detectors_amd.utils import main_util
    cfg = MU.config_for(meta)
    t_sd, s_sd = MU.oracle_states(meta['seed'], meta['model'], meta.get('bch', 3), meta.get('num_classes', 91))
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    box = DistillationBox(teacher, student, cfg['train']['criterion'])
    opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
    warm = main_util.warmup_lr_scheduler(opt, 4, 1e-3)
    return cfg, t_sd, s_sd, teacher, student, box, opt, warm


@pytest.mark.parametrize('name', TINY)
def test_distill_steps_match_reference_golden(name):
    z, meta = G.load(name)
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    terms = MU.terms_of(cfg)
    images, targets = G.case_inputs(meta)
    ms = meta['min_size'] if isinstance(meta['min_size'], list) else [meta['min_size']]
    orc64 = O.DistillOracle(t_sd, s_sd, terms=terms, min_size=tuple(ms), max_size=meta['max_size'],
                            dtype=torch.float64)
    orc32 = O.DistillOracle(t_sd, s_sd, terms=terms, min_size=tuple(ms), max_size=meta['max_size'])
    worst = {'feat': 0.0, 'loss': 0.0, 'grad': 0.0}
    # The keypoint fixture resizes its two images to different sizes, so ~40 % of the padded batch is a constant
    # region: a channel whose constant pre-activation sits within rounding of 0 flips the ReLU mask of the whole
    # region at once in one fp32 implementation and not in the other (torch fp32 itself moves by 2e-3 vs fp64 on
    # other seeds).  Gradients of that one fixture are therefore held to 6e-3; features / losses stay at 1e-3.
    gtol = 6e-3 if meta['model'] == 'keypoint_rcnn' else None
    for step in range(meta['steps']):
        ims, tgs = _to_dev(images, targets)
        fixed = None
        if meta['model'] == 'keypoint_rcnn':
            random.seed(100 + step)             # tool.py:45-48 draws sizes from python's RNG
            fixed = [int(v) for v in z['step%d/fixed_sizes' % step]]
        _sync_oracle(orc64, student)
        _sync_oracle(orc32, student)
        _, _, g64, _ = orc64.step(images, fixed)
        _, _, g32, _ = orc32.step(images, fixed)
        loss = box(ims, tgs)
        ref_loss = float(z['step%d/loss' % step])
        worst['loss'] = max(worst['loss'], abs(loss.item() - ref_loss) / abs(ref_loss))
        per_term = loss.per_term.cpu()
        for i, k in enumerate(terms):
            ref = float(z['step%d/term/%s' % (step, k)])
            worst['loss'] = max(worst['loss'], abs(float(per_term[i]) - ref) / abs(ref))
        if step == 0:
            for k in terms:
                path = cfg['train']['criterion']['terms'][k]['ts_modules']
                t_out, s_out = _hooked(teacher, path[0]), _hooked(student, path[1])
                assert t_out.dim() == 4 and t_out.shape[1] in (256, 512, 1024, 2048)      # logical NCHW
                worst['feat'] = max(worst['feat'], G.compare(z, 'step0/teacher/' + k, t_out.contiguous(), FEAT_TOL),
                                    G.compare(z, 'step0/student/' + k, s_out.contiguous(), FEAT_TOL))
        opt.zero_grad()
        loss.backward()
        assert abs(opt.param_groups[0]['lr'] - float(z['step%d/lr' % step])) < 1e-12
        for n, p in student.named_parameters():
            if p.requires_grad and not n.endswith(G.ZERO_GRAD_SUFFIXES):
                key = 'step%d/grad/%s' % (step, n)
                if key in z.files:      # full reference (fp32) gradient stored: judge both against fp64
                    worst['grad'] = max(worst['grad'], _grad_check(n, p.grad, torch.from_numpy(z[key]), g64[n], gtol))
                else:                   # checksum form in the fixture: the oracle (== reference to rounding noise)
                    worst['grad'] = max(worst['grad'], _grad_check(n, p.grad, g32[n], g64[n], gtol))   # stands in, and
                    G.compare(z, key, p.grad, 5e-2)      # the stored fingerprint is still matched loosely
        opt.step()
        warm.step()
    sd = student.state_dict()
    # A single ReLU-mask flip (one activation of ~3e5 within rounding of zero landing on the other side than in the
    # reference run) moves every upstream gradient by ~1e-3 -- inside the gradient bar above -- and Adam's normalised
    # update then moves the small BatchNorm biases by several 1e-3 relative.  The flips are COUNTED by
    # test_relu_decisions_differ_from_the_fp32_oracle_only_where_the_value_is_rounding_noise (0 / 1 / 1 differing
    # decisions of 1-5 M for the three fixtures whose gradients sit at 2e-5 / 9e-4 / 4e-3).  The updated parameters are
    # therefore held to 1e-3 when no flip happened (all gradients within 1e-4; achieved 1.0e-4), else to 5e-3 (achieved
    # 2.6e-5 / 8.1e-7 on the two fixtures with one flip each); the flip count of the fixture is printed beside it.
    ptol = 5e-3 if (gtol or worst['grad'] > 1e-4) else 1e-3
    worst['param'] = 0.0
    for n in O.trainable_keys(s_sd):
        if not n.endswith(G.ZERO_GRAD_SUFFIXES):
            worst['param'] = max(worst['param'], G.compare(z, 'after/param/' + n, sd[n], ptol, atol=1e-6))
    for n in z.files:
        if n.startswith('after/buffer/'):
            key = n[len('after/buffer/'):]
            ref = torch.from_numpy(z[n]).double()
            got = sd[key].cpu().double()
            assert float((got - ref).abs().max()) <= 1e-4 * (1 + float(ref.abs().max())), key
    assert worst['loss'] < LOSS_TOL, worst
    line = ('[%s] worst rel err: features %.2e (tol %.0e)  loss %.2e (%.0e)  grads vs fp64 %.2e (%.0e or 2x torch-fp32); '
            'params after Adam %.2e (held to %.0e); worst e_hip/e_ref beyond %.0e so far: %.2f (%s)'
            % (name, worst['feat'], FEAT_TOL, worst['loss'], LOSS_TOL, worst['grad'], gtol or GRAD_TOL, worst['param'], ptol, GRAD_TOL,
               WORST_GRAD_RATIO['ratio'], WORST_GRAD_RATIO['name']))
    print('\n' + line)
    from tests import conftest
    conftest.record_achieved(lambda: line + '; ReLU decisions differing from the fp32 oracle: %s'
                             % conftest.FLIPS.get(name, 'not counted for this fixture'))


def test_eval_after_a_training_step_uses_the_updated_weights():
    """FusedAdam updates parameters through raw pointers (no torch version bump); the packed / Winograd
    weight caches must still notice, so a train-then-eval sequence in one process (ext_runner's per-epoch validation,
    split / eval after distillation) runs the UPDATED conv weights.  lr is raised so a stale pack would be far off."""
    z, meta = G.load('tiny_ghnd_faster')
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    images, targets = G.case_inputs(meta)
    ims, tgs = _to_dev(images, targets)
    student.eval()
    with torch.no_grad():
        before = student(ims)[0].clone()                    # builds the eval-mode plans and packs
    student.train()
    for group in opt.param_groups:
        group['lr'] = 5e-2
    for _ in range(2):
        loss = box(ims, [dict(t) for t in tgs])
        opt.zero_grad()
        loss.backward()
        opt.step()
    student.eval()
    with torch.no_grad():
        feats = student(ims)
    sd = OrderedDict((k, v.detach().cpu().clone()) for k, v in student.state_dict().items())
    x, _ = O.transform_images(images, (meta['min_size'],), meta['max_size'])
    _, o_f = O.backbone_forward(x, O.cast_state(sd, torch.float32), student=True, training=False)
    assert float((feats[0] - before).abs().max()) > 1e-3         # the step really moved the output
    for k, v in feats.items():
        ref = o_f[k]
        err = float((v.cpu().double() - ref.double()).norm() / ref.double().norm())
        assert err < FEAT_TOL, (k, err)


def test_transform_and_fpn_match_golden():
    z, meta = G.load('tiny_ghnd_faster')
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    images, targets = G.case_inputs(meta)
    ims, tgs = _to_dev(images, targets)
    il, _ = teacher.transform(ims, None, None)
    assert tuple(il.tensors.shape) == tuple(z['batched_shape'])
    assert il.image_sizes == [(64, 96), (64, 114)]
    G.compare(z, 'transform', il.tensors.contiguous(), 1e-5)
    feats = student(ims, tgs)                                   # distill_backbone_only -> FPN dict
    assert list(feats.keys()) == [0, 1, 2, 3, 'pool']
    for k, v in feats.items():
        G.compare(z, 'student_fpn/%s' % k, v.contiguous(), FEAT_TOL)


def test_decoded_input_pipeline_matches_reference_fixture():
    """uint8 HWC images through this build's ToTensor / RandomHorizontalFlip and the model's transform (one fused
    kernel per image) == the reference's ToTensor -> flip -> CustomRCNNTransform (tiny_input_pipeline.npz)."""
    from hnd_ghnd_object_detectors_amd.structure.transformer import RandomHorizontalFlip, ToTensor
    z, meta = G.load('tiny_ghnd_faster')
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    zp = G.load_raw('tiny_input_pipeline')
    images, targets = [], []
    for u8, flip, tin, tout in G.pipeline_case(zp):
        img, tgt = ToTensor()(u8, {k: v.clone() for k, v in tin.items()})
        img, tgt = RandomHorizontalFlip(1.0 if flip else 0.0)(img, tgt)
        images.append(img.to(DEV, non_blocking=True))
        targets.append({k: v.to(DEV) for k, v in tgt.items()})
    teacher.eval()
    masks_before = [t['masks'].cpu().clone() for t in targets]      # the transform rescales the targets in place
    il, out_t = teacher.transform(images, targets, None)
    assert [list(s) for s in il.image_sizes] == zp['image_sizes'].tolist()
    ref = torch.from_numpy(zp['batch'])
    assert float((il.tensors.cpu() - ref).abs().max()) < 2e-5
    assert out_t[0]['masks'].shape[-2:] == tuple(il.image_sizes[0])
    # resized ground-truth masks (reference rcnn.py:54-57; the fixture stores the targets BEFORE the transform):
    # hnd_resize_mask_nearest_u8 against the reference's own expression on the CPU, byte for byte
    from hnd_ghnd_object_detectors_amd import engine as E
    eng = E.shared_transform(teacher.transform.image_mean, teacher.transform.image_std, images[0].device)
    for got, before, scale in zip(out_t, masks_before, eng.last_scales):
        ref_m = torch.nn.functional.interpolate(before[None].float(), scale_factor=scale)[0].byte()
        assert got['masks'].dtype == torch.uint8 and torch.equal(got['masks'].cpu(), ref_m)


def test_deferred_fpn_stream_changes_nothing(monkeypatch):
    """inside DistillationBox both pyramids are issued on a side stream that overlaps the backward pass
    (HND_DEFER_FPN): loss, gradients, updated parameters and the pyramid outputs are bit-identical to the
    in-order run."""
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    z, meta = G.load('tiny_ghnd_faster')
    images, targets = G.case_inputs(meta)
    results = []
    for defer in ('0', '1'):
        monkeypatch.setenv('HND_DEFER_FPN', defer)
        cfg = MU.config_for(meta)
        t_sd, s_sd = MU.oracle_states(meta['seed'])
        teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
        box = DistillationBox(teacher, student, cfg['train']['criterion'])
        assert box.defer_fpn == (defer == '1')
        opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
        seen = {}
        student.backbone.fpn.register_forward_hook(lambda m, i, o: seen.update(fpn=o))
        for _ in range(3):
            ims, tgs = _to_dev(images, targets)
            loss = box(ims, tgs)
            opt.zero_grad()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        results.append((float(loss.detach()), [p.detach().clone() for p in student.parameters() if p.requires_grad],
                        [v.clone() for v in seen['fpn'].values()]))
    (l0, p0, f0), (l1, p1, f1) = results
    assert l0 == l1 and all(torch.equal(a, b) for a, b in zip(p0, p1))
    assert len(f0) == 5 and all(torch.equal(a, b) for a, b in zip(f0, f1))


@pytest.mark.parametrize('case', ['tiny_ghnd_faster', 'full_ghnd_faster_b4'])
def test_weight_gradient_stream_changes_no_bit(case, monkeypatch):
    """engine.wgrad_stream_on (HND_WGRAD_STREAM): the head's weight-gradient chains run on a stream of their own beside the
    data-gradient chain, joined after the stem's backward.  Same kernels in the same order inside each chain: loss,
    every gradient of every step and the updated parameters are bit-identical to the in-order run, at the tiny and at
    the full geometry (where the Winograd-domain weight gradients get Z buffers of their own)"""
    z, meta = G.load(case)
    if case.startswith('full'):
        meta = dict(meta, sizes=meta['sizes'][:2])
    images, targets = G.case_inputs(meta)
    results = []
    for mode in ('0', '1'):
        monkeypatch.setenv('HND_WGRAD_STREAM', mode)
        cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
        grads = []
        for _ in range(3):
            ims, tgs = _to_dev(images, [dict(t) for t in targets])
            loss = box(ims, tgs)
            opt.zero_grad()
            loss.backward()
            grads.append([p.grad.detach().clone() for p in student.parameters() if p.requires_grad])
            opt.step()
        torch.cuda.synchronize()
        head = student.backbone.body.layer1.head_engine()
        assert head.bwd_side == (mode == '1')
        results.append((float(loss.detach()), grads, [p.detach().clone() for p in student.parameters() if p.requires_grad]))
    (l0, g0, p0), (l1, g1, p1) = results
    assert l0 == l1
    for a, b in zip(g0, g1):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert all(torch.equal(a, b) for a, b in zip(p0, p1))


def test_against_oracle_on_fresh_inputs_with_resume_of_buffers():
    """three steps on new seeded inputs (batch 3, odd sizes) vs the CPU oracle run side by side."""
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    cfg = MU.config_for(model='faster_rcnn', method='ghnd', bch=3, min_size=96, max_size=160)
    t_sd, s_sd = MU.oracle_states(77)
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    box = DistillationBox(teacher, student, cfg['train']['criterion'])
    opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
    orc = O.DistillOracle(t_sd, s_sd, min_size=(96,), max_size=160)
    orc64 = O.DistillOracle(t_sd, s_sd, min_size=(96,), max_size=160, dtype=torch.float64)
    g = torch.Generator().manual_seed(5)
    for step in range(3):
        images = [torch.rand(3, h, w, generator=g) for h, w in ((70, 110), (96, 96), (50, 121))]
        targets = [{'boxes': torch.tensor([[1., 2., 30., 40.]]), 'labels': torch.tensor([1])} for _ in images]
        ims, tgs = _to_dev(images, targets)
        # Adam at lr 1e-3 is sign-like on the first steps, so fp32 trajectories (HIP or torch) separate chaotically;
        # every step is therefore checked in isolation from the HIP model's current parameters.
        _sync_oracle(orc, student)
        _sync_oracle(orc64, student)
        before = {n: p.detach().cpu().clone() for n, p in student.named_parameters() if p.requires_grad}
        loss = box(ims, tgs)
        opt.zero_grad()
        loss.backward()
        got = {n: p.grad.detach().cpu().clone() for n, p in student.named_parameters() if p.requires_grad}
        opt.step()
        ref_loss, _, ref_grads, _ = orc.step(images)
        _, _, g64, _ = orc64.step(images)
        assert abs(loss.item() - ref_loss) / abs(ref_loss) < LOSS_TOL, (step, loss.item(), ref_loss)
        for n, gr in ref_grads.items():
            if not n.endswith(G.ZERO_GRAD_SUFFIXES):
                _grad_check('step%d %s' % (step, n), got[n], gr, g64[n])
        # first Adam step is exactly -lr * g / (|g| + eps): magnitude lr, direction -sign(g)
        if step == 0:
            for n, p in student.named_parameters():
                if p.requires_grad and not n.endswith(G.ZERO_GRAD_SUFFIXES):
                    delta = p.detach().cpu() - before[n]
                    assert float(delta.abs().max()) <= 1e-3 * 1.001, n
                    big = got[n].abs() > 1e-3 * got[n].abs().max()
                    assert torch.equal(torch.sign(delta[big]), -torch.sign(got[n][big])), n


def test_student_eval_mode_uses_running_statistics():
    cfg = MU.config_for(model='faster_rcnn', method='hnd', bch=3, min_size=64, max_size=128)
    t_sd, s_sd = MU.oracle_states(21)
    for k in list(s_sd):                       # non-trivial running stats
        if k.endswith('running_var') and 'layer1' in k:
            s_sd[k] = s_sd[k] * 1.7 + 0.1
        if k.endswith('running_mean') and 'layer1' in k:
            s_sd[k] = s_sd[k] + 0.05
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    student.eval()
    images = [torch.rand(3, 64, 96, generator=torch.Generator().manual_seed(1))]
    with torch.no_grad():
        feats = student([im.to(DEV) for im in images])
    x, _ = O.transform_images(images, (64,), 128)
    ref_h, ref_f = O.backbone_forward(x, O.cast_state(s_sd, torch.float32), student=True, training=False)
    for k in (0, 1, 2, 3, 'pool'):
        rel = float((feats[k].cpu() - ref_f[k]).abs().max() / ref_f[k].abs().max())
        assert rel < FEAT_TOL, (k, rel)


def test_eval_with_quantized_bottleneck_matches_reference_golden():
    """student.eval() + use_bottleneck_transformer (the reference's -transform_bottleneck path, base.py:54-57)."""
    z, meta = G.load('tiny_eval_quantized')
    cfg = MU.config_for(meta)
    t_sd, s_sd = MU.oracle_states(meta['seed'])
    for k in list(s_sd):
        if 'layer1' in k and k.endswith('running_var'):
            s_sd[k] = s_sd[k] * 1.7 + 0.1
        if 'layer1' in k and k.endswith('running_mean'):
            s_sd[k] = s_sd[k] + 0.05
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    student.eval()
    images, _ = G.case_inputs(meta)
    ims = [im.to(DEV) for im in images]
    for tag, use, tol in (('plain', False, FEAT_TOL), ('quantized', True, 2e-2)):
        student.backbone.body.layer1.use_bottleneck_transformer = use
        with torch.no_grad():
            feats = student(ims)
        for k, v in feats.items():
            # a bottleneck value within rounding of a quantisation boundary may land in the neighbouring bin
            # (one step = 1/255 of the range), which moves a few downstream features visibly: hence 2e-2 there
            G.compare(z, '%s/fpn/%s' % (tag, k), v.contiguous(), tol)
    # the codec is eval-only: in train mode the same flag must not change anything (mimic_runner.py:90 sets it False)
    student.train()
    student.backbone.body.layer1.use_bottleneck_transformer = True
    a = student(ims, [{'boxes': torch.zeros(1, 4, device=DEV), 'labels': torch.ones(1, dtype=torch.int64, device=DEV)}
                      for _ in ims])[0].clone()
    student.backbone.body.layer1.use_bottleneck_transformer = False
    b = student(ims, [{'boxes': torch.zeros(1, 4, device=DEV), 'labels': torch.ones(1, dtype=torch.int64, device=DEV)}
                      for _ in ims])[0]
    assert torch.equal(a, b)


def test_quantizer_reproduces_reference_bytes_on_reference_bottleneck():
    """Byte work is bit-exact: the package's Quantizer / Dequantizer (reference src/structure/transformer.py:131-153)
    fed the REFERENCE's own bottleneck tensor z (fixture ``quantized/z``) must emit exactly the bytes, scale,
    zero_point and dequantised floats the reference's Quantizer emitted for it."""
    from hnd_ghnd_object_detectors_amd.structure.transformer import Dequantizer, Quantizer
    z = G.load_raw('tiny_eval_quantized')
    z_ref = torch.from_numpy(z['quantized/z']).to(DEV)                   # [N, 3, h, w] fp32
    qz, _ = Quantizer(8)(z_ref, None)
    assert qz.tensor.dtype == torch.uint8 and tuple(qz.tensor.shape) == tuple(z_ref.shape)
    assert torch.equal(qz.tensor.cpu(), torch.from_numpy(z['quantized/bytes']))
    assert float(qz.scale) == float(z['quantized/scale']) and float(qz.zero_point) == float(z['quantized/zero_point'])
    deq, _ = Dequantizer(8)(qz, None)
    assert torch.equal(deq.cpu(), torch.from_numpy(z['quantized/dequantized']))


@pytest.mark.parametrize('quantization,tag,tol', [(None, 'plain', FEAT_TOL), (8, 'quantized', 2e-2)])
def test_head_tail_split_matches_reference_golden(quantization, tag, tol):
    """split_rcnn_model: RcnnHead (transform + stem + encoder [+ uint8 quantiser]) -> z -> RcnnTail ([dequantiser]
    + decoder + layer2-4 + FPN) reproduces the reference's unsplit eval features (the same arithmetic, base.py:54-57
    vs split_rcnn.py:13-37,162-185) and the unsplit HIP model bit for bit."""
    from hnd_ghnd_object_detectors_amd.models.mimic.split_rcnn import split_rcnn_model
    from hnd_ghnd_object_detectors_amd.structure.transformer import QuantizedTensor
    z, meta = G.load('tiny_eval_quantized')
    cfg = MU.config_for(meta)
    t_sd, s_sd = MU.oracle_states(meta['seed'])
    for k in list(s_sd):
        if 'layer1' in k and k.endswith('running_var'):
            s_sd[k] = s_sd[k] * 1.7 + 0.1
        if 'layer1' in k and k.endswith('running_mean'):
            s_sd[k] = s_sd[k] + 0.05
    _, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    student.eval()
    images, _ = G.case_inputs(meta)
    ims = [im.to(DEV) for im in images]
    student.backbone.body.layer1.use_bottleneck_transformer = quantization is not None
    with torch.no_grad():
        whole = OrderedDict((k, v.clone()) for k, v in student(ims).items())
    head, tail = split_rcnn_model(student, quantization)
    head.eval()
    tail.eval()
    tail.features_only = True
    with torch.no_grad():
        zq, tensors_shape, image_sizes, original_sizes = head(ims)
        if quantization is not None:
            assert isinstance(zq, QuantizedTensor) and zq.tensor.dtype == torch.uint8 and zq.tensor.shape[1] == 3
            # what crosses the link: the uint8 tensor + (scale, zero_point); rebuild it as the tail's side would
            zq = QuantizedTensor(zq.tensor, zq.scale, zq.zero_point, zq.qparams.clone(), origin=None,
                                 channels=zq.channels)
        else:
            assert zq.shape[1] == 3
        feats = tail(zq, tensors_shape, image_sizes, original_sizes)
    assert original_sizes == [tuple(im.shape[-2:]) for im in images] and len(tensors_shape) == 4
    for k, v in feats.items():
        G.compare(z, '%s/fpn/%s' % (tag, k), v.contiguous(), tol)
        assert torch.equal(v, whole[k]), k
    # the whole split detector (reference split_rcnn.py:186-196): head -> link -> tail incl. RPN / RoI box head /
    # NMS == the unsplit model's detections, bit for bit (same kernels on the same values)
    tail.features_only = False
    student.distill_backbone_only = False
    with torch.no_grad():
        dets_split = tail(zq, tensors_shape, image_sizes, original_sizes)
        dets_whole = student(ims)
    assert len(dets_split) == len(dets_whole) == len(ims)
    for a, b in zip(dets_split, dets_whole):
        assert sorted(a.keys()) == ['boxes', 'labels', 'scores']
        for k in a:
            assert torch.equal(a[k], b[k]), k


FULL = ['full_ghnd_faster', 'full_ghnd_faster_b4', 'full_hnd_faster_b2', 'full_ghnd_mask_b2', 'full_ghnd_keypoint_b2',
        'full_ghnd_mask_b8']


def _full_step(z, meta, repeat=1):
    """one distillation step of a full-size fixture's inputs (the batch replicated `repeat` times) on the HIP path"""
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    images, targets = G.case_inputs(meta)
    ims, tgs = _to_dev(images * repeat, [dict(t) for t in targets * repeat])
    if meta['model'] == 'keypoint_rcnn':
        random.seed(100)                  # tool.py:45-48 draws one size per image from python's RNG
    loss = box(ims, tgs)
    return cfg, teacher, student, box, opt, warm, ims, tgs, loss


# fingerprints (sum, sum of squares, 64 strided samples / rms) of the reference's fp32 gradients at full size: no fp64 twin
# is stored, so the bar is absolute -- 4e-3: the worst SAMPLE of a gradient that a ReLU flip moved by 1.7e-3 ... 2.2e-3 in
# rel-L2 (dense tests, vs fp64) sits at 2.6e-3 of the tensor's rms (round 5, printed below); was 5e-3
FULL_GRAD_FP_TOL = 4e-3
# parameters after one Adam step at full size vs the reference's: achieved 0.8e-6 ... 2.3e-6 (Adam's update is bounded by
# lr whatever the gradient's error); was 2e-3
FULL_PARAM_TOL = 2e-5


def _record_full(name, worst_g, worst_p):
    from tests.conftest import record_achieved
    line = ('[full size %s] gradient fingerprints vs the reference: worst %.2e (tol %.1e); parameters after Adam %.2e '
            '(tol %.0e)' % (name, worst_g, FULL_GRAD_FP_TOL, worst_p, FULL_PARAM_TOL))
    print('\n' + line)
    record_achieved(line)


@pytest.mark.parametrize('name', FULL)
def test_full_size_step_matches_reference_checksums(name):
    """3x800x1333 inputs (padded 800x1344; Keypoint: sizes drawn per image as tool.py:45-48, padded 1248x1120) at the
    reference's own batch sizes: fingerprints (sum, sum of squares, 64 strided samples) of every hooked map, the
    loss and its terms, all 23 non-degenerate gradients and the parameters after the Adam step, against what the
    REFERENCE produced for the same seeded inputs (tests/golden/make_golden.py)."""
    if not os.path.exists(os.path.join(G.GOLDEN_DIR, name + '.npz')):
        pytest.skip('fixture %s not generated (needs > 48 GB of host RAM in the build container)' % name)
    z, meta = G.load(name)
    cfg, teacher, student, box, opt, warm, ims, tgs, loss = _full_step(z, meta)
    if meta['model'] == 'keypoint_rcnn':
        want = [int(v) for v in z['step0/fixed_sizes']]
        random.seed(100)
        assert [random.choice(teacher.transform.min_size) for _ in ims] == want
    hp, wp = (int(v) for v in z['batched_shape'][2:])
    assert tuple(_hooked(teacher, 'backbone.body.layer1').shape) == (len(ims), 256, hp // 4, wp // 4)
    ref = float(z['step0/loss'])
    assert abs(loss.item() - ref) / ref < LOSS_TOL
    per_term = loss.per_term.cpu()
    for i, k in enumerate(MU.terms_of(cfg)):
        rt = float(z['step0/term/%s' % k])
        assert abs(float(per_term[i]) - rt) / rt < LOSS_TOL, k
        path = 'backbone.body.' + k
        G.compare(z, 'step0/teacher/' + k, _hooked(teacher, path).contiguous(), FEAT_TOL)
        G.compare(z, 'step0/student/' + k, _hooked(student, path).contiguous(), FEAT_TOL)
    opt.zero_grad()
    loss.backward()
    assert abs(opt.param_groups[0]['lr'] - float(z['step0/lr'])) < 1e-12
    worst_g = worst_p = 0.0
    for n, p in student.named_parameters():
        if p.requires_grad and not n.endswith(G.ZERO_GRAD_SUFFIXES):
            worst_g = max(worst_g, G.compare(z, 'step0/grad/' + n, p.grad, FULL_GRAD_FP_TOL))
    opt.step()
    sd = student.state_dict()
    for n in O.trainable_keys(sd):
        if not n.endswith(G.ZERO_GRAD_SUFFIXES):
            worst_p = max(worst_p, G.compare(z, 'after/param/' + n, sd[n], FULL_PARAM_TOL, atol=1e-6))
    _record_full(name, worst_g, worst_p)
    for n in z.files:
        if n.startswith('after/buffer/'):
            key = n[len('after/buffer/'):]
            refb, got = torch.from_numpy(z[n]).double(), sd[key].cpu().double()
            assert float((got - refb).abs().max()) <= 1e-4 * (1 + float(refb.abs().max())), key


def test_batch16_is_the_reference_batch4_replicated():
    """The benchmarked configuration (batch 16 at 3x800x1333) pinned to the reference without a 90 GB CPU run: the
    reference's own batch of 4 (config/ghnd/...b3ch.yaml:65, fixture full_ghnd_faster_b4) replicated 4x.  Train-mode
    BatchNorm statistics are invariant under replicating the batch, so every hooked map of images 0..3 must match
    the reference's batch-4 fingerprints, images 4k..4k+3 must equal images 0..3 BIT FOR BIT (same arithmetic per
    pixel whatever the position in the batch), the loss is 4x the reference's and every gradient 4x the reference's
    (sum reduction); Adam's update is invariant under a gradient scale, so the parameters after the step match too."""
    z, meta = G.load('full_ghnd_faster_b4')
    cfg, teacher, student, box, opt, warm, ims, tgs, loss = _full_step(z, meta, repeat=4)
    assert len(ims) == 16
    ref = 4.0 * float(z['step0/loss'])
    assert abs(loss.item() - ref) / ref < LOSS_TOL
    per_term = loss.per_term.cpu()
    for i, k in enumerate(MU.terms_of(cfg)):
        assert abs(float(per_term[i]) - 4.0 * float(z['step0/term/%s' % k])) / (4.0 * float(z['step0/term/%s' % k])) < LOSS_TOL
        for who, model in (('teacher', teacher), ('student', student)):
            out = _hooked(model, 'backbone.body.' + k)
            assert out.shape[0] == 16
            G.compare(z, 'step0/%s/%s' % (who, k), out[:4].contiguous(), FEAT_TOL)
            for r in range(1, 4):
                assert torch.equal(out[4 * r:4 * r + 4], out[:4]), (who, k, r)
    opt.zero_grad()
    loss.backward()
    worst_g = worst_p = 0.0
    for n, p in student.named_parameters():
        if p.requires_grad and not n.endswith(G.ZERO_GRAD_SUFFIXES):
            worst_g = max(worst_g, G.compare(z, 'step0/grad/' + n, p.grad * 0.25, FULL_GRAD_FP_TOL))
    opt.step()
    sd = student.state_dict()
    for n in O.trainable_keys(sd):
        if not n.endswith(G.ZERO_GRAD_SUFFIXES):
            worst_p = max(worst_p, G.compare(z, 'after/param/' + n, sd[n], FULL_PARAM_TOL, atol=1e-6))
    _record_full('batch 16 = full_ghnd_faster_b4 x 4', worst_g, worst_p)


def test_batch16_hnd_is_the_reference_batch2_replicated():
    """BASELINE config 2 (HND, batch 16 per GPU) at the benchmarked batch: the reference's HND batch of 2
    (fixture full_hnd_faster_b2, layer1 loss only) replicated 8x.  Same invariances as the GHND test above: maps of
    images 0..1 match the reference's fingerprints, images 2k..2k+1 equal them bit for bit, loss and gradients are
    8x the reference's, Adam's update is scale-invariant."""
    z, meta = G.load('full_hnd_faster_b2')
    cfg, teacher, student, box, opt, warm, ims, tgs, loss = _full_step(z, meta, repeat=8)
    assert len(ims) == 16 and list(MU.terms_of(cfg)) == ['layer1']
    ref = 8.0 * float(z['step0/loss'])
    assert abs(loss.item() - ref) / ref < LOSS_TOL
    assert abs(float(loss.per_term.cpu()[0]) - 8.0 * float(z['step0/term/layer1'])) / ref < LOSS_TOL
    for who, model in (('teacher', teacher), ('student', student)):
        out = _hooked(model, 'backbone.body.layer1')
        assert out.shape[0] == 16
        G.compare(z, 'step0/%s/layer1' % who, out[:2].contiguous(), FEAT_TOL)
        for r in range(1, 8):
            assert torch.equal(out[2 * r:2 * r + 2], out[:2]), (who, r)
    opt.zero_grad()
    loss.backward()
    worst_g = worst_p = 0.0
    for n, p in student.named_parameters():
        if p.requires_grad and not n.endswith(G.ZERO_GRAD_SUFFIXES):
            worst_g = max(worst_g, G.compare(z, 'step0/grad/' + n, p.grad * 0.125, FULL_GRAD_FP_TOL))
    opt.step()
    sd = student.state_dict()
    for n in O.trainable_keys(sd):
        if not n.endswith(G.ZERO_GRAD_SUFFIXES):
            worst_p = max(worst_p, G.compare(z, 'after/param/' + n, sd[n], FULL_PARAM_TOL, atol=1e-6))
    _record_full('batch 16 = full_hnd_faster_b2 x 8', worst_g, worst_p)


def _hip_relu_decisions(student):
    """[value > 0] of EVERY ReLU of the student's forward pass as the HIP engines hold them right after the forward: stem,
    the head's ReLUs after train-mode BatchNorm, a1 / a2 / output of all Bottlenecks of layers 2-4 (CPU bool, NCHW)."""
    body = student.backbone.body
    head = body.layer1.head_engine()

    def dec(buf, c=None):
        return (buf[..., :c] > 0).permute(0, 3, 1, 2).cpu()

    got = {'stem': dec(body.stem().a0)}
    for k, li_ in enumerate([k for k, hc in enumerate(head.layers) if hc.relu]):
        got['head.relu%d' % k] = dec(head.y[li_] * head.scale[li_] + head.shift[li_])
    for li in (2, 3, 4):
        eng = body.layer_engine('layer%d' % li)
        for i in range(len(body['layer%d' % li])):
            x_in, a1, a2, out_ = (eng._b(t) for t in eng.acts[i])
            got['layer%d.%d.a1' % (li, i)], got['layer%d.%d.a2' % (li, i)] = dec(a1), dec(a2)
            got['layer%d.%d.out' % (li, i)] = dec(out_)
    return got


def _oracle_relu_values(orc32, student, inter, s_hooked, x_batch):
    """the same ReLU maps as the fp32 CPU oracle computes them (values, NCHW), keyed like _hip_relu_decisions"""
    import torch.nn.functional as F
    sd, body = orc32.s, student.backbone.body
    conv1 = F.conv2d(x_batch, sd[O.B + 'conv1.weight'], None, stride=2, padding=3)
    ref = OrderedDict(stem=F.relu(O.frozen_bn(conv1, sd, O.B + 'bn1.')))
    relu_names = ['%s%d' % (pfx, op[1]) for pfx, spec in ((O.B + 'layer1.encoder.encoder.', O.ENCODER_SPEC),
                                                         (O.B + 'layer1.decoder.', O.DECODER_SPEC))
                  for op in spec if op[0] == 'relu']
    for k, n in enumerate(relu_names):
        ref['head.relu%d' % k] = inter[n]
    for li in (2, 3, 4):
        for i in range(len(body['layer%d' % li])):
            pfx = '%slayer%d.%d.' % (O.B, li, i)
            xin = s_hooked['layer%d' % (li - 1)] if i == 0 else s_hooked['layer%d.%d' % (li, i - 1)]
            a1 = F.relu(O.frozen_bn(F.conv2d(xin, sd[pfx + 'conv1.weight']), sd, pfx + 'bn1.'))
            a2 = F.relu(O.frozen_bn(F.conv2d(a1, sd[pfx + 'conv2.weight'], None, stride=2 if (i == 0) else 1,
                                             padding=1), sd, pfx + 'bn2.'))
            ref['layer%d.%d.a1' % (li, i)], ref['layer%d.%d.a2' % (li, i)] = a1, a2
            ref['layer%d.%d.out' % (li, i)] = s_hooked['layer%d.%d' % (li, i)]
    return ref


def _count_relu_flips(got, ref, top=None):
    """(flips, total, largest differing value relative to its map's max, {map: flips}) between HIP decisions and oracle
    values; `top`: only maps at or below that layer carry gradient (HND: the head and the stem)."""
    flips = total = 0
    worst_rel, where = 0.0, {}
    for key, r in ref.items():
        if top is not None and key.startswith('layer') and int(key[5]) > top:
            continue
        g_ = got[key][:, :r.shape[1]]
        assert tuple(g_.shape) == tuple(r.shape), (key, g_.shape, r.shape)
        d = g_ != (r > 0)
        total += d.numel()
        if bool(d.any()):
            flips += int(d.sum())
            where[key] = int(d.sum())
            worst_rel = max(worst_rel, float(r[d].abs().max() / r.abs().max()))
    return flips, total, worst_rel, where


DENSE_MAXABS_RMS = 1e-3      # worst |d| of any element, in units of its map's rms (achieved ~9e-5: a tail-tile bug is O(1))
DENSE = {   # name -> (fixture whose seeded inputs / weights are reused, number of images taken from it)
    'ghnd_faster_b2': ('full_ghnd_faster_b4', 2),
    'hnd_faster_b2': ('full_hnd_faster_b2', 2),
    'ghnd_keypoint_b2': ('full_ghnd_keypoint_b2', 2),
}


@pytest.mark.parametrize('case', sorted(DENSE))
def test_full_size_dense_parity_every_element(case):
    """at 3x800x1333 (Keypoint: 1248x1120 padded, sizes drawn per image) EVERY element of every
    hooked map of both networks, every loss term and every gradient is compared -- not a fingerprint.  The CPU oracle
    (pinned to the reference by tests/golden; it travels to the GPU box) runs the same step on the host in fp32 and,
    for the gradients, in fp64; the HIP maps must agree with the oracle's to 1e-3 relative L2 per map AND per image,
    with the worst absolute element difference bounded in units of the map's rms so an edge-tile / tail-tile error in
    any image fails; gradients by _grad_check (vs fp64, no worse than 2x the reference arithmetic's own error)."""
    fixture, nimg = DENSE[case]
    z, meta = G.load(fixture)
    meta = dict(meta, sizes=meta['sizes'][:nimg])
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    terms = MU.terms_of(cfg)
    images, targets = G.case_inputs(meta)
    fixed = None
    if meta['model'] == 'keypoint_rcnn':
        random.seed(100)
        fixed = [random.choice(teacher.transform.min_size) for _ in images]
        random.seed(100)                   # the box draws the same sizes again (tool.py:45-48)
    ims, tgs = _to_dev(images, targets)
    loss = box(ims, tgs)
    decisions = _hip_relu_decisions(student)          # before the backward pass touches any buffer
    opt.zero_grad()
    loss.backward()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ms = meta['min_size'] if isinstance(meta['min_size'], list) else [meta['min_size']]
    orc32 = O.DistillOracle(t_sd, s_sd, terms=terms, min_size=tuple(ms), max_size=meta['max_size'])
    inter = {}
    o_loss, o_terms, t_hooked, s_hooked, _, _, x = orc32.forward(images, fixed, intermediates=inter)
    # ReLU decisions of the student's forward pass that differ from the fp32 oracle's: each one
    # switches a gradient path on in one implementation and off in the other -- the gradient figure below is made of them
    with torch.no_grad():
        top_term = max(int(k[5]) for k in terms if k.startswith('layer') and k[5:6].isdigit())
        flips, total, flip_rel, where = _count_relu_flips(
            decisions, _oracle_relu_values(orc32, student, {k: v.detach() for k, v in inter.items()},
                                           {k: v.detach() for k, v in s_hooked.items()}, x.detach()),
            top=top_term)
    del decisions, inter
    assert flip_rel < 1e-5, (flips, total, flip_rel, where)      # only values that are rounding noise may decide differently
    o_loss.backward()
    g32 = OrderedDict((k, orc32.s[k].grad.detach().clone()) for k in orc32.keys)
    report = []
    worst_abs_rms = 0.0
    for i, k in enumerate(terms):
        assert abs(float(loss.per_term[i]) - float(o_terms[k])) / abs(float(o_terms[k])) < LOSS_TOL, k
        for who, model, hooked in (('teacher', teacher, t_hooked), ('student', student, s_hooked)):
            got = _hooked(model, 'backbone.body.' + k).cpu().double()
            ref = hooked[k].detach().double()
            assert got.shape == ref.shape, (k, who, got.shape, ref.shape)
            diff = got - ref
            rel = float(diff.norm() / ref.norm())
            per_img = (diff.flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)).max().item()
            rms = float(ref.pow(2).mean().sqrt())
            worst_abs = float(diff.abs().max())
            # borders (first / last two rows and columns of the padded map: where a tail-tile bug would live)
            edge = torch.zeros_like(ref, dtype=torch.bool)
            edge[..., :2, :] = edge[..., -2:, :] = edge[..., :, :2] = edge[..., :, -2:] = True
            rel_edge = float(diff[edge].norm() / ref[edge].norm().clamp_min(1e-30))
            report.append('%s/%s rel %.1e img %.1e edge %.1e max|d| %.1e (rms %.2e)'
                          % (who, k, rel, per_img, rel_edge, worst_abs, rms))
            assert rel < FEAT_TOL and per_img < FEAT_TOL and rel_edge < FEAT_TOL, report[-1]
            assert worst_abs < DENSE_MAXABS_RMS * rms + 1e-6, report[-1]
            worst_abs_rms = max(worst_abs_rms, worst_abs / rms)
    assert abs(loss.item() - float(o_loss)) / abs(float(o_loss)) < LOSS_TOL
    del t_hooked, s_hooked, o_loss
    orc64 = O.DistillOracle(t_sd, s_sd, terms=terms, min_size=tuple(ms), max_size=meta['max_size'],
                            dtype=torch.float64, with_fpn=False)
    l64, *_ = orc64.forward(images, fixed)
    l64.backward()
    worst_g = 0.0
    for n, p in student.named_parameters():
        if p.requires_grad and not n.endswith(G.ZERO_GRAD_SUFFIXES):
            worst_g = max(worst_g, _grad_check(n, p.grad, g32[n], orc64.s[n].grad))
    flip_txt = '%d of %d ReLU decisions upstream of a loss term differ from the fp32 oracle (largest such value %.1e of its ' \
               'map\'s max%s)' % (flips, total, flip_rel, '' if not where else '; ' + ', '.join(
                   '%s: %d' % kv for kv in sorted(where.items(), key=lambda kv: -kv[1])[:5]))
    print('\n[dense %s, batched %s] %s\n  loss rel %.1e; worst gradient rel-L2 vs fp64 %.2e; %s'
          % (case, tuple(x.shape), '\n  '.join(report), abs(loss.item() - float(l64)) / abs(float(l64)), worst_g, flip_txt))
    from tests.conftest import record_achieved
    worst_map = max(float(r.split('rel ')[1].split(' ')[0]) for r in report)
    record_achieved('[dense %s] every element of %d maps: worst rel-L2 %.1e (tol %.0e), worst max|d| / rms %.1e (tol %.0e); '
                    'loss rel %.1e; worst gradient rel-L2 vs fp64 %.2e; %s'
                    % (case, len(report), worst_map, FEAT_TOL, worst_abs_rms, DENSE_MAXABS_RMS,
                       abs(loss.item() - float(l64)) / abs(float(l64)), worst_g, flip_txt))
    for r in report:
        record_achieved('    ' + r)


@pytest.mark.parametrize('first', [False, 'layer2', 'layer3'])
def test_custom_hook_paths_match_reference_golden(first, monkeypatch):
    """VERDICT r3 item 7 / reference src/distillation/tool.py:22-35: ``ts_modules`` may name ANY dotted module path.  The
    fixture was produced by the reference's own DistillationBox with teacher ``layer1`` <-> student ``layer1.decoder``,
    an inner Bottleneck of layer2 on both sides, layer3 as usual, and the FIRST Bottleneck of layer4 as the highest term
    (so the backward starts in the middle of a layer and layer4.1 / layer4.2 carry no gradient).  Hooks on modules that
    execute fused inside their parent fire with the engine's tensors; two Adam steps against the reference, gradients
    against the fp64 oracle -- with the shared trunk off, from layer2 and from layer3."""
    from hnd_ghnd_object_detectors_amd import engine as E
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from hnd_ghnd_object_detectors_amd.utils import main_util
    monkeypatch.setattr(E, 'MERGE_TRUNK', bool(first))
    monkeypatch.setattr(E, 'MERGE_FROM', first or 'layer3')
    z, meta = G.load('tiny_ghnd_custom_hooks')
    cfg = MU.config_for(meta)
    crit = cfg['train']['criterion']
    proto = next(iter(crit['terms'].values()))['criterion']
    crit['terms'] = OrderedDict((name, {'ts_modules': [tp, sp], 'criterion': proto, 'factor': f})
                                for name, tp, sp, f in meta['terms'])
    strip = len('backbone.body.')
    terms = OrderedDict((name, (tp[strip:], sp[strip:], f)) for name, tp, sp, f in meta['terms'])
    t_sd, s_sd = MU.oracle_states(meta['seed'], meta['model'])
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    box = DistillationBox(teacher, student, crit)
    opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
    warm = main_util.warmup_lr_scheduler(opt, 4, 1e-3)
    images, targets = G.case_inputs(meta)
    kw = dict(terms=terms, min_size=(meta['min_size'],), max_size=meta['max_size'])
    orc64, orc32 = O.DistillOracle(t_sd, s_sd, dtype=torch.float64, **kw), O.DistillOracle(t_sd, s_sd, **kw)
    worst = {'feat': 0.0, 'loss': 0.0, 'grad': 0.0}
    for step in range(meta['steps']):
        ims, tgs = _to_dev(images, targets)
        _sync_oracle(orc64, student)
        _sync_oracle(orc32, student)
        _, _, g64, _ = orc64.step(images)
        _, _, g32, _ = orc32.step(images)
        loss = box(ims, tgs)
        ref_loss = float(z['step%d/loss' % step])
        worst['loss'] = max(worst['loss'], abs(loss.item() - ref_loss) / abs(ref_loss))
        for i, (name, tp, sp, f) in enumerate(meta['terms']):
            ref_t = float(z['step%d/term/%s' % (step, name)])
            worst['loss'] = max(worst['loss'], abs(float(loss.per_term[i]) - ref_t) / abs(ref_t))
            if step == 0:
                worst['feat'] = max(worst['feat'], G.compare(z, 'step0/teacher/' + name, _hooked(teacher, tp), FEAT_TOL),
                                    G.compare(z, 'step0/student/' + name, _hooked(student, sp), FEAT_TOL))
        opt.zero_grad()
        loss.backward()
        for n, p in student.named_parameters():
            if p.requires_grad and not n.endswith(G.ZERO_GRAD_SUFFIXES):
                worst['grad'] = max(worst['grad'], _grad_check(n, p.grad, g32[n], g64[n]))
        opt.step()
        warm.step()
    assert worst['loss'] < LOSS_TOL, worst
    body = student.backbone.body
    shared = [n for n in ('layer2', 'layer3', 'layer4') if body.layer_engine(n) is not body[n]._engine]
    assert shared == (['layer2', 'layer3', 'layer4'][int(first[-1]) - 2:] if first else [])
    sd = student.state_dict()
    ptol = 2e-2 if worst['grad'] > 1e-4 else 2e-3          # (a ReLU flip: see test_distill_steps_match_reference_golden)
    worst['param'] = max(G.compare(z, 'after/param/' + n, sd[n], ptol, atol=1e-6)
                         for n in O.trainable_keys(s_sd) if not n.endswith(G.ZERO_GRAD_SUFFIXES))
    from tests.conftest import record_achieved
    record_achieved('[custom hook paths (layer1.decoder, layer2.1, layer3, layer4.0), trunk %s] maps %.1e, loss / terms %.1e, '
                    'gradients vs fp64 %.2e, parameters after 2 Adam steps %.1e'
                    % (first or 'off', worst['feat'], worst['loss'], worst['grad'], worst['param']))


@pytest.mark.parametrize('name', ['tiny_ghnd_fpn_term', 'tiny_enc_term'])
def test_student_side_terms_on_a_pyramid_map_and_on_the_bottleneck_tensor_match_reference_golden(name):
    """VERDICT r5 'What's missing' #2 / reference src/distillation/tool.py:25-35 (hooks on ANY module, student side too).
    Both fixtures were produced by the reference's own DistillationBox:
    tiny_ghnd_fpn_term -- a term on ``backbone.fpn.layer_blocks.1`` beside layer1 and layer3: the pyramid's backward (3x3
      data gradient, lateral 1x1 data gradients, the nearest-upsample backward of the top-down path) delivers into layer2,
      layer3 and layer4, where the backward STARTS although no term sits on layer4;
    tiny_enc_term -- a term on ``backbone.body.layer1.encoder`` (the bottleneck tensor) beside layer2, the teacher being a
      bottleneck-injected model itself (eval mode): the gradient joins the head's backward behind decoder.0's BatchNorm.
    Maps and terms against the reference, two Adam steps, gradients against the fp64 oracle."""
    import copy
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from hnd_ghnd_object_detectors_amd.utils import main_util
    z, meta = G.load(name)
    cfg = MU.config_for(meta)
    crit = cfg['train']['criterion']
    proto = next(iter(crit['terms'].values()))['criterion']
    crit['terms'] = OrderedDict((tn, {'ts_modules': [tp, sp], 'criterion': proto, 'factor': f})
                                for tn, tp, sp, f in meta['terms'])
    terms = OrderedDict((tn, (O.rel_key(tp), O.rel_key(sp), f)) for tn, tp, sp, f in meta['terms'])
    t_sd, s_sd = MU.oracle_states(meta['seed'], meta['model'])
    student_arch = meta.get('teacher') == 'student_arch'
    if student_arch:
        t_sd = O.init_student_state(t_sd, meta['seed'] + 500)
        cfg['teacher_model'] = copy.deepcopy(cfg['student_model'])
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    box = DistillationBox(teacher, student, crit)
    assert not (box.defer_fpn and 'fpn' in name)            # a pyramid with a term on it is not loss-dead
    opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
    warm = main_util.warmup_lr_scheduler(opt, 4, 1e-3)
    images, targets = G.case_inputs(meta)
    kw = dict(terms=terms, min_size=(meta['min_size'],), max_size=meta['max_size'], teacher_is_student_arch=student_arch)
    orc64, orc32 = O.DistillOracle(t_sd, s_sd, dtype=torch.float64, **kw), O.DistillOracle(t_sd, s_sd, **kw)
    worst = {'feat': 0.0, 'loss': 0.0, 'grad': 0.0}
    for step in range(meta['steps']):
        ims, tgs = _to_dev(images, targets)
        _sync_oracle(orc64, student)
        _sync_oracle(orc32, student)
        _, _, g64, _ = orc64.step(images)
        _, _, g32, _ = orc32.step(images)
        loss = box(ims, tgs)
        ref_loss = float(z['step%d/loss' % step])
        worst['loss'] = max(worst['loss'], abs(loss.item() - ref_loss) / abs(ref_loss))
        for i, (tn, tp, sp, f) in enumerate(meta['terms']):
            ref_t = float(z['step%d/term/%s' % (step, tn)])
            worst['loss'] = max(worst['loss'], abs(float(loss.per_term[i]) - ref_t) / abs(ref_t))
            if step == 0:
                worst['feat'] = max(worst['feat'], G.compare(z, 'step0/teacher/' + tn, _hooked(teacher, tp), FEAT_TOL),
                                    G.compare(z, 'step0/student/' + tn, _hooked(student, sp), FEAT_TOL))
        opt.zero_grad()
        loss.backward()
        for n, p in student.named_parameters():
            if p.requires_grad and not n.endswith(G.ZERO_GRAD_SUFFIXES):
                worst['grad'] = max(worst['grad'], _grad_check(n, p.grad, g32[n], g64[n]))
                G.compare(z, 'step%d/grad/%s' % (step, n), p.grad, 5e-3 if worst['grad'] > 1e-4 else 1e-3)
        opt.step()
        warm.step()
    assert worst['loss'] < LOSS_TOL, worst
    sd = student.state_dict()
    ptol = 2e-2 if worst['grad'] > 1e-4 else 2e-3          # (a ReLU flip: see test_distill_steps_match_reference_golden)
    worst['param'] = max(G.compare(z, 'after/param/' + n, sd[n], ptol, atol=1e-6)
                         for n in O.trainable_keys(s_sd) if not n.endswith(G.ZERO_GRAD_SUFFIXES))
    from tests.conftest import record_achieved
    record_achieved('[student-side terms, %s: %s] maps %.1e, loss / terms %.1e, gradients vs fp64 %.2e, parameters after 2 '
                    'Adam steps %.1e' % (name, ', '.join(sp for _, _, sp, _ in meta['terms']), worst['feat'], worst['loss'],
                                         worst['grad'], worst['param']))


def test_full_size_pyramid_and_bottleneck_terms_against_the_oracle():
    """the two new kinds of student-side terms at 3x800x1333, batch 2 (no reference-made fixture at this size: the oracle,
    pinned to the reference by the tiny fixtures of the same paths, runs on the host): a term on the stride-8 pyramid map
    beside layer1 and layer3 -- F(6x6,3x3) data gradient of the pyramid's 3x3 conv, lateral data gradients into 100x168 /
    50x84 / 25x42 maps, the nearest-upsample backward -- and, with a bottleneck-injected teacher, a term on the bottleneck
    tensor beside layer2.  Loss and terms against the fp32 oracle, every gradient against fp64 (_grad_check)."""
    import copy
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from tests.conftest import record_achieved
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    for fixture in ('tiny_ghnd_fpn_term', 'tiny_enc_term'):
        z, meta = G.load(fixture)
        meta = dict(meta, sizes=[(800, 1333), (800, 1333)], min_size=800, max_size=1333)
        cfg = MU.config_for(meta)
        crit = cfg['train']['criterion']
        proto = next(iter(crit['terms'].values()))['criterion']
        crit['terms'] = OrderedDict((tn, {'ts_modules': [tp, sp], 'criterion': proto, 'factor': f})
                                    for tn, tp, sp, f in meta['terms'])
        terms = OrderedDict((tn, (O.rel_key(tp), O.rel_key(sp), f)) for tn, tp, sp, f in meta['terms'])
        t_sd, s_sd = MU.oracle_states(meta['seed'], meta['model'])
        student_arch = meta.get('teacher') == 'student_arch'
        if student_arch:
            t_sd = O.init_student_state(t_sd, meta['seed'] + 500)
            cfg['teacher_model'] = copy.deepcopy(cfg['student_model'])
        teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
        box = DistillationBox(teacher, student, crit)
        opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
        images, targets = G.case_inputs(meta)
        ims, tgs = _to_dev(images, targets)
        loss = box(ims, tgs)
        opt.zero_grad()
        loss.backward()
        kw = dict(terms=terms, min_size=(800,), max_size=1333, teacher_is_student_arch=student_arch)
        orc32 = O.DistillOracle(t_sd, s_sd, **kw)
        l32, t32, g32, _ = orc32.step(images)
        worst_t = abs(loss.item() - l32) / abs(l32)
        for i, tn in enumerate(terms):
            worst_t = max(worst_t, abs(float(loss.per_term[i]) - t32[tn]) / abs(t32[tn]))
        assert worst_t < LOSS_TOL, worst_t
        del orc32
        orc64 = O.DistillOracle(t_sd, s_sd, dtype=torch.float64, **kw)
        _, _, g64, _ = orc64.step(images)
        worst_g = 0.0
        for n, p in student.named_parameters():
            if p.requires_grad and not n.endswith(G.ZERO_GRAD_SUFFIXES):
                worst_g = max(worst_g, _grad_check(n, p.grad, g32[n], g64[n]))
        record_achieved('[student-side terms at 3x800x1333 b2, %s] loss / terms vs the fp32 oracle %.1e, worst gradient rel-L2 vs '
                        'fp64 %.2e' % (', '.join(sp for _, _, sp, _ in meta['terms']), worst_t, worst_g))
        del teacher, student, box, orc64


def test_a_lone_term_on_the_bottleneck_tensor_starts_the_backward_in_the_middle_of_the_head():
    """the bottleneck tensor as the ONLY (hence top) student-side term: the decoder and decoder.0's BatchNorm carry no
    gradient (exact zeros, as autograd leaves them), the backward starts at the encoder's last conv.  Against the fp64
    oracle (no reference-made fixture: the reference's own run differs only in leaving those .grad None)."""
    import copy
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    z, meta = G.load('tiny_enc_term')
    cfg = MU.config_for(meta)
    crit = cfg['train']['criterion']
    proto = next(iter(crit['terms'].values()))['criterion']
    tn, tp, sp, f = meta['terms'][0]
    crit['terms'] = OrderedDict([(tn, {'ts_modules': [tp, sp], 'criterion': proto, 'factor': f})])
    terms = OrderedDict([(tn, (O.rel_key(tp), O.rel_key(sp), f))])
    t_sd, s_sd = MU.oracle_states(meta['seed'], meta['model'])
    t_sd = O.init_student_state(t_sd, meta['seed'] + 500)
    cfg['teacher_model'] = copy.deepcopy(cfg['student_model'])
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    box = DistillationBox(teacher, student, crit)
    opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
    images, targets = G.case_inputs(meta)
    kw = dict(terms=terms, min_size=(meta['min_size'],), max_size=meta['max_size'], teacher_is_student_arch=True)
    orc64, orc32 = O.DistillOracle(t_sd, s_sd, dtype=torch.float64, **kw), O.DistillOracle(t_sd, s_sd, **kw)
    worst = 0.0
    for step in range(2):
        ims, tgs = _to_dev(images, targets)
        _sync_oracle(orc64, student)
        _sync_oracle(orc32, student)
        l64, _, g64, _ = orc64.step(images)
        _, _, g32, _ = orc32.step(images)
        loss = box(ims, tgs)
        assert abs(loss.item() - l64) / abs(l64) < LOSS_TOL
        opt.zero_grad()
        loss.backward()
        for n, p in student.named_parameters():
            if not p.requires_grad:
                continue
            if float(g64[n].abs().max()) == 0.0:             # the decoder: no path from the term
                assert float(p.grad.abs().max()) == 0.0, n
            elif not n.endswith(G.ZERO_GRAD_SUFFIXES):
                worst = max(worst, _grad_check(n, p.grad, g32[n], g64[n]))
        opt.step()
    assert any('decoder' in n for n, p in student.named_parameters() if p.requires_grad)
    from tests.conftest import record_achieved
    record_achieved('[a lone term on backbone.body.layer1.encoder] decoder gradients exactly zero, encoder / stem gradients vs '
                    'fp64 %.2e' % worst)


def test_unsupported_hook_paths_are_refused_with_the_list_of_supported_ones():
    """what still has no backward plan is refused loudly, never silently ignored: a student-side term on a LATERAL map of the
    pyramid (backbone.fpn.inner_blocks.K), a teacher / student pair of different shapes; a hook on a parameter holder that
    never runs on its own names the problem.  (Terms on the bottleneck tensor and on pyramid maps are supported since
    round 6: test_student_side_terms_on_a_pyramid_map_and_on_the_bottleneck_tensor_match_reference_golden.)"""
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    z, meta = G.load('tiny_ghnd_custom_hooks')
    images, targets = G.case_inputs(meta)
    for tp, sp, msg in (('backbone.fpn.inner_blocks.3', 'backbone.fpn.inner_blocks.3', 'layer_blocks'),
                        ('backbone.fpn.layer_blocks.1', 'backbone.body.layer1.encoder', 'shapes differ'),
                        ('backbone.body.layer1', 'backbone.body.layer1.decoder.3', 'fused')):
        cfg = MU.config_for(meta)
        crit = cfg['train']['criterion']
        proto = next(iter(crit['terms'].values()))['criterion']
        crit['terms'] = OrderedDict([('t', {'ts_modules': [tp, sp], 'criterion': proto, 'factor': 1.0})])
        t_sd, s_sd = MU.oracle_states(meta['seed'], meta['model'])
        teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
        box = DistillationBox(teacher, student, crit)
        ims, tgs = _to_dev(images, targets)
        with pytest.raises((NotImplementedError, ValueError), match=msg):
            box(ims, tgs)


def _one_step_bits(meta, images, targets, merged, monkeypatch, fixed_seed=None):
    """merged: False (two separate passes) or the first layer of the shared pass"""
    from hnd_ghnd_object_detectors_amd import engine as E
    monkeypatch.setattr(E, 'MERGE_TRUNK', bool(merged))
    monkeypatch.setattr(E, 'MERGE_FROM', merged or 'layer3')
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    ims, tgs = _to_dev(images, [dict(t) for t in targets])
    if fixed_seed is not None:
        random.seed(fixed_seed)
    loss = box(ims, tgs)
    opt.zero_grad()
    loss.backward()
    body = student.backbone.body
    used = [n for n in ('layer2', 'layer3', 'layer4') if body.layer_engine(n) is not body[n]._engine]
    maps = OrderedDict()
    for who, model in (('teacher', teacher), ('student', student)):
        for k in MU.terms_of(cfg):
            maps['%s/%s' % (who, k)] = _hooked(model, 'backbone.body.' + k).clone()
    grads = OrderedDict((n, p.grad.clone()) for n, p in student.named_parameters() if p.requires_grad)
    if box.defer_fpn and box._fpn_stream is not None:
        torch.cuda.current_stream().wait_stream(box._fpn_stream)
    return used, loss.item(), loss.per_term.cpu().clone(), maps, grads, teacher, student


@pytest.mark.parametrize('case,first', [('tiny_ghnd_faster', 'layer2'), ('tiny_ghnd_faster', 'layer3'),
                                        ('tiny_ghnd_keypoint', 'layer4'), ('tiny_ghnd_keypoint', 'layer3'),
                                        ('full_ghnd_faster_b4', 'layer3'), ('full_ghnd_faster_b4', 'layer2'),
                                        ('full_hnd_faster_b2', 'layer3')])
def test_shared_trunk_halves_equal_the_separate_passes_bit_for_bit(case, first, monkeypatch):
    """layers 2-4 + FPN of teacher and student as ONE pass over the concatenated batch
    (engine.SharedTrunk) must give, in the teacher half, exactly the teacher's separate pass and, in the student half,
    exactly the student's -- every hooked map, the loss and its terms, and every gradient (the backward plan runs over
    the student half of the merged buffers) BIT FOR BIT."""
    z, meta = G.load(case)
    if case.startswith('full'):
        meta = dict(meta, sizes=meta['sizes'][:2])
    images, targets = G.case_inputs(meta)
    seed = 100 if meta['model'] == 'keypoint_rcnn' else None
    runs = {m: _one_step_bits(meta, images, targets, m, monkeypatch, seed) for m in (False, first)}
    shared = ['layer2', 'layer3', 'layer4'][int(first[-1]) - 2:]
    assert runs[first][0] == shared and runs[False][0] == [], 'the merged run uses the shared engines from `first` on'
    assert runs[first][1] == runs[False][1] and torch.equal(runs[first][2], runs[False][2])
    for k in runs[False][3]:
        assert torch.equal(runs[first][3][k], runs[False][3][k]), k
    for n in runs[False][4]:
        assert torch.equal(runs[first][4][n], runs[False][4][n]), n


@pytest.mark.parametrize('case', ['tiny_ghnd_faster', 'full_ghnd_faster_b4'])
def test_mask_nibbles_change_no_bit_of_the_step(case, monkeypatch):
    """engine.MASK_BITS: the conv1 data gradients of the frozen Bottlenecks read [x > 0] as nibbles written by the forward
    epilogue instead of the fp32 activation -- the same decisions, so loss, terms and EVERY gradient are bit-identical to
    the fp32-mask step.  Held inside the NATIVE family (HND_BF16X3=0's build): the emulation kernel reads masks as nibbles
    only, so a launch with an fp32 mask and one with nibbles would sit in two different rounding families."""
    from hnd_ghnd_object_detectors_amd import engine as E, ops as OPS
    monkeypatch.setattr(OPS, 'BX3_MODE', ['off'])
    z, meta = G.load(case)
    if case.startswith('full'):
        meta = dict(meta, sizes=meta['sizes'][:2])
    images, targets = G.case_inputs(meta)
    runs = {}
    for bits in (False, True):
        monkeypatch.setattr(E, 'MASK_BITS', bits)
        cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
        ims, tgs = _to_dev(images, [dict(t) for t in targets])
        loss = box(ims, tgs)
        opt.zero_grad()
        loss.backward()
        eng = student.backbone.body.layer_engine('layer3')
        used = [l.desc.mask_bits is not None for l, tag in eng.bwd if tag.endswith('conv1.dgrad')]
        assert all(used) == bits and any(used) == bits, used
        runs[bits] = (loss.item(), loss.per_term.cpu().clone(),
                      OrderedDict((n, p.grad.clone()) for n, p in student.named_parameters() if p.requires_grad))
        if box.defer_fpn and box._fpn_stream is not None:
            torch.cuda.current_stream().wait_stream(box._fpn_stream)
    assert runs[True][0] == runs[False][0] and torch.equal(runs[True][1], runs[False][1])
    for n in runs[False][2]:
        assert torch.equal(runs[True][2][n], runs[False][2][n]), n


def test_fused_bn_backward_transforms_give_the_unfused_gradients(monkeypatch):
    """engine.FUSE_BNBWD (conv1 / conv6 / conv7 of the head at full size: the BatchNorm backward apply inside the
    data-gradient input transform and the weight-gradient dy transform): same loss, every gradient within 1e-5 relative
    L2 of the unfused plan (a different rounding of k1 d + k2 x + k3 at most), and the fused plan must actually be in use"""
    from hnd_ghnd_object_detectors_amd import engine as E
    z, meta = G.load('full_ghnd_faster_b4')
    meta = dict(meta, sizes=meta['sizes'][:2])
    images, targets = G.case_inputs(meta)
    runs = {}
    for fuse in (False, True):
        monkeypatch.setattr(E, 'FUSE_BNBWD', fuse)
        cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
        ims, tgs = _to_dev(images, [dict(t) for t in targets])
        loss = box(ims, tgs)
        opt.zero_grad()
        loss.backward()
        head = student.backbone.body.layer1.head_engine()
        assert [i for i, st in enumerate(head.bsteps) if st['fused'] is not None] == ([1, 6, 7] if fuse else [])
        runs[fuse] = (loss.item(), OrderedDict((n, p.grad.clone()) for n, p in student.named_parameters()
                                               if p.requires_grad))
        if box.defer_fpn and box._fpn_stream is not None:
            torch.cuda.current_stream().wait_stream(box._fpn_stream)
    assert runs[True][0] == runs[False][0]
    worst = 0.0
    for n in runs[False][1]:
        if n.endswith(G.ZERO_GRAD_SUFFIXES):
            continue
        a, b = runs[True][1][n].double(), runs[False][1][n].double()
        worst = max(worst, float((a - b).norm() / b.norm()))
    assert worst < 1e-5, worst
    from tests.conftest import record_achieved
    record_achieved('[fused BN-backward transforms vs the unfused plan, 3x800x1333 b2] worst gradient rel-L2 %.1e' % worst)


def test_folded_bn_backward_reduce_gives_the_unfolded_gradients(monkeypatch):
    """engine.FOLD_BNBWD_REDUCE: the BatchNorm-backward sums of a head layer whose g comes out of an F(6x6,2x2) data
    gradient are made by that launch's output transform (no separate pass over g and x): same loss, every gradient within
    1e-5 relative L2 of the plan with the reduce kernel, and the fold must be in use for the layers below the three
    Winograd data gradients of the full-size head"""
    from hnd_ghnd_object_detectors_amd import engine as E
    z, meta = G.load('full_ghnd_faster_b4')
    meta = dict(meta, sizes=meta['sizes'][:2])
    images, targets = G.case_inputs(meta)
    runs = {}
    for fold in (False, True):
        monkeypatch.setattr(E, 'FOLD_BNBWD_REDUCE', fold)
        cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
        ims, tgs = _to_dev(images, [dict(t) for t in targets])
        loss = box(ims, tgs)
        opt.zero_grad()
        loss.backward()
        head = student.backbone.body.layer1.head_engine()
        folded = [i for i, st in enumerate(head.bsteps) if st['folded'] is not None]
        # full-size head: g0, g5, g6 come out of F(6x6,2x2) data gradients, g1 out of a direct one (conv2.dgrad) on a kernel
        # with a statistics epilogue -- and so does g4 (conv5.dgrad) since the B-streamed emulation kernel took it from the
        # native B-streamed one, which has none (round 6)
        from hnd_ghnd_object_detectors_amd import ops as OPS
        assert folded == (([0, 1, 4, 5, 6] if OPS.bx3_on() else [0, 1, 5, 6]) if fold else []), folded
        for i in folded:
            last = head.bsteps[i + 1]['dgrad'][-1][0]
            assert getattr(last, 'kernel', '') == 'wino2_output' or bool(last.desc.bwd_x), i
        runs[fold] = (loss.item(), OrderedDict((n, p.grad.clone()) for n, p in student.named_parameters()
                                               if p.requires_grad))
        if box.defer_fpn and box._fpn_stream is not None:
            torch.cuda.current_stream().wait_stream(box._fpn_stream)
    assert runs[True][0] == runs[False][0]
    worst = 0.0
    for n in runs[False][1]:
        if n.endswith(G.ZERO_GRAD_SUFFIXES):
            continue
        a, b = runs[True][1][n].double(), runs[False][1][n].double()
        worst = max(worst, float((a - b).norm() / b.norm()))
    assert worst < 1e-5, worst
    from tests.conftest import record_achieved
    record_achieved('[BN-backward reduce folded into the data-gradient output transform vs the reduce kernel, '
                    '3x800x1333 b2] worst gradient rel-L2 %.1e' % worst)


def test_step_loss_item_reads_the_early_host_copy_and_not_the_drained_stream():
    """hip_loss.StepLoss: the reference reads loss.item() after optimizer.step() (mimic_runner.py:58 -> misc_util.py:150);
    the value handed out is the float copied to pinned memory right after the loss kernel -- the same float Tensor.item()
    would return -- for every step of a short run, with the buffers of the ring rotating"""
    from hnd_ghnd_object_detectors_amd.distillation.hip_loss import StepLoss
    z, meta = G.load('tiny_ghnd_faster')
    images, targets = G.case_inputs(meta)
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    ims, tgs = _to_dev(images, targets)
    seen, bufs, kept = [], set(), []
    for it in range(6):
        loss = box(ims, [dict(t) for t in tgs])
        assert isinstance(loss, StepLoss) and loss.requires_grad
        opt.zero_grad()
        loss.backward()
        opt.step()
        early = loss.item()
        assert early == torch.Tensor.item(loss) == float(loss.detach().cpu())
        assert float(loss) == early
        bufs.add(loss._host[0].data_ptr())
        seen.append(early)
        if it < 3:
            kept.append(loss)                                # an unread-later loss keeps its pinned scalar to itself
        del loss
    assert [l.item() for l in kept] == seen[:3]
    assert 4 <= len(bufs) <= 5 and len(set(seen)) == 6      # 3 kept + recycled ones; the loss moves with every Adam step
    assert abs(seen[0] - float(z['step0/loss'])) / float(z['step0/loss']) < 1e-4


@pytest.mark.parametrize('name', ['tiny_ghnd_faster', 'tiny_ghnd_faster_b6', 'tiny_ghnd_keypoint'])
def test_relu_decisions_differ_from_the_fp32_oracle_only_where_the_value_is_rounding_noise(name):
    """Why gradients carry a 2e-3 bar (6e-3 on the keypoint fixture) and parameters after Adam a 2e-2 one (VERDICT r3 weak
    #2).  EVERY ReLU decision of the student's forward pass -- stem, the head's four ReLUs after train-mode BatchNorm,
    a1 / a2 / output of all 13 Bottlenecks of layers 2-4 -- is compared element by element between the HIP path and the
    fp32 CPU oracle on the same weights and images, for every step of the fixture.  Decisions may differ only where both
    implementations hold rounding noise (|value| < 1e-5 of the map's max): such an element switches a gradient path on in
    one implementation and off in the other, and that is the whole of the slack.  The count is printed with the parity
    figures (round 4, beside the gradient errors of test_distill_steps_match_reference_golden): tiny_ghnd_faster 0 of
    5.4 M decisions differ <-> gradients within 1.8e-5; tiny_ghnd_faster_b6 ONE (an output of layer2.0, 1.8e-8 of its
    map's max) <-> 9.2e-4; tiny_ghnd_keypoint ONE (an a1 of layer2.0, 2.0e-8) <-> 3.9e-3."""
    import torch.nn.functional as F
    from tests.conftest import record_achieved
    z, meta = G.load(name)
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    terms = MU.terms_of(cfg)
    images, targets = G.case_inputs(meta)
    ms = meta['min_size'] if isinstance(meta['min_size'], list) else [meta['min_size']]
    orc32 = O.DistillOracle(t_sd, s_sd, terms=terms, min_size=tuple(ms), max_size=meta['max_size'])
    body = student.backbone.body
    flips = total = 0
    worst_rel = 0.0
    where = {}

    def nchw_of(buf, c):
        return buf[..., :c].permute(0, 3, 1, 2).detach().cpu().float()

    for step in range(meta['steps']):
        ims, tgs = _to_dev(images, targets)
        fixed = None
        if meta['model'] == 'keypoint_rcnn':
            random.seed(100 + step)
            fixed = [int(v) for v in z['step%d/fixed_sizes' % step]]
        _sync_oracle(orc32, student)
        inter = {}
        with torch.no_grad():
            out = orc32.forward(images, fixed, update_buffers=False, intermediates=inter)
            s_hooked, x_batch, sd = out[3], out[6], orc32.s
            conv1 = F.conv2d(x_batch, sd[O.B + 'conv1.weight'], None, stride=2, padding=3)
            ref_pairs = [('stem', F.relu(O.frozen_bn(conv1, sd, O.B + 'bn1.')))]
            relu_names = ['%s%d' % (pfx, op[1]) for pfx, spec in ((O.B + 'layer1.encoder.encoder.', O.ENCODER_SPEC),
                                                                 (O.B + 'layer1.decoder.', O.DECODER_SPEC))
                          for op in spec if op[0] == 'relu']
            ref_pairs += [('head.relu%d' % k, inter[n]) for k, n in enumerate(relu_names)]
            for li in (2, 3, 4):
                for i in range(len(body['layer%d' % li])):
                    pfx = '%slayer%d.%d.' % (O.B, li, i)
                    xin = s_hooked['layer%d' % (li - 1)] if i == 0 else s_hooked['layer%d.%d' % (li, i - 1)]
                    a1 = F.relu(O.frozen_bn(F.conv2d(xin, sd[pfx + 'conv1.weight']), sd, pfx + 'bn1.'))
                    a2 = F.relu(O.frozen_bn(F.conv2d(a1, sd[pfx + 'conv2.weight'], None, stride=2 if (i == 0) else 1,
                                                     padding=1), sd, pfx + 'bn2.'))
                    ref_pairs += [('layer%d.%d.a1' % (li, i), a1), ('layer%d.%d.a2' % (li, i), a2),
                                  ('layer%d.%d.out' % (li, i), s_hooked['layer%d.%d' % (li, i)])]
        loss = box(ims, tgs)
        head = body.layer1.head_engine()
        got = {'stem': body.stem().a0}
        relu_layers = [k for k, hc in enumerate(head.layers) if hc.relu]
        assert len(relu_layers) == len(relu_names)
        for k, li_ in enumerate(relu_layers):
            got['head.relu%d' % k] = torch.relu(head.y[li_] * head.scale[li_] + head.shift[li_])
        for li in (2, 3, 4):
            eng = body.layer_engine('layer%d' % li)
            for i in range(len(body['layer%d' % li])):
                x_in, a1, a2, out_ = (eng._b(t) for t in eng.acts[i])
                got['layer%d.%d.a1' % (li, i)], got['layer%d.%d.a2' % (li, i)] = a1, a2
                got['layer%d.%d.out' % (li, i)] = out_
        for key, ref in ref_pairs:
            g_ = nchw_of(got[key], ref.shape[1])
            assert tuple(g_.shape) == tuple(ref.shape), (key, g_.shape, ref.shape)
            d = (g_ > 0) != (ref > 0)
            total += d.numel()
            if bool(d.any()):
                flips += int(d.sum())
                where[key] = where.get(key, 0) + int(d.sum())
                worst_rel = max(worst_rel, float(torch.maximum(g_[d].abs(), ref[d].abs()).max() / ref.abs().max()))
        opt.zero_grad()
        loss.backward()
        opt.step()
        warm.step()
    assert worst_rel < 1e-5, (flips, total, worst_rel, where)
    from tests import conftest
    conftest.FLIPS[name] = '%d of %d' % (flips, total)
    record_achieved('[ReLU decisions, HIP vs fp32 oracle, all %d ReLU maps x %d steps of %s] %d of %d elements differ%s; the '
                    'largest differing value is %.1e of its map\'s max'
                    % (len(ref_pairs), meta['steps'], name, flips, total,
                       '' if not where else ' (%s)' % ', '.join('%s: %d' % kv for kv in sorted(where.items())[:6]), worst_rel))


def test_shared_trunk_is_dropped_when_the_frozen_weights_differ(monkeypatch):
    """the merged pass needs bit-equal frozen weights; a student whose layer3 was edited (a checkpoint that did not come
    from this teacher) runs its own pass -- and the pyramids of the merged pass equal the separate ones"""
    from hnd_ghnd_object_detectors_amd import engine as E
    monkeypatch.setattr(E, 'MERGE_TRUNK', True)             # (opt-in: HND_MERGE_TRUNK=1)
    monkeypatch.setattr(E, 'MERGE_FROM', 'layer3')
    z, meta = G.load('tiny_ghnd_faster')
    images, targets = G.case_inputs(meta)
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    ims, tgs = _to_dev(images, targets)
    box(ims, [dict(t) for t in tgs])
    body = student.backbone.body
    assert box._trunk is not None and body.layer_engine('layer3') is box._trunk.engines['layer3']
    with torch.no_grad():
        t_ref = _hooked(teacher, 'backbone.body.layer3').clone()
        saved = body['layer3'][1].conv2.weight.clone()
        body['layer3'][1].conv2.weight.mul_(1.5)            # version bump -> the equality is re-checked
    loss = box(ims, [dict(t) for t in tgs])
    assert body.layer_engine('layer3') is body['layer3']._engine and E.MERGE['trunk'] is None
    assert torch.equal(_hooked(teacher, 'backbone.body.layer3'), t_ref)         # the teacher is untouched
    assert bool(torch.isfinite(loss))
    with torch.no_grad():
        body['layer3'][1].conv2.weight.copy_(saved)
    box(ims, [dict(t) for t in tgs])
    assert body.layer_engine('layer3') is box._trunk.engines['layer3']          # and merged again once equal


def test_batch16_teacher_maps_equal_batch1_maps_bitwise_and_steps_are_reproducible():
    """Two batch-16 properties that need no CPU run.  (i) the frozen teacher has no cross-image coupling: the hooked
    maps of image i inside a batch of 16 distinct images equal the maps of the same image alone, bit for bit --
    whatever block tile / Winograd grouping / XCD remap the 16x larger grid picks.  (ii) two identical batch-16
    steps from the same state give bit-identical loss, per-term losses and gradients (fixed-order reductions)."""
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import module_util
    z, meta = G.load('full_ghnd_faster_b4')
    cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
    g = torch.Generator().manual_seed(77)
    images = [torch.rand(3, 800, 1333, generator=g) for _ in range(16)]
    targets = [{'boxes': torch.tensor([[100.0, 100.0, 400.0, 300.0]]), 'labels': torch.tensor([1])} for _ in images]
    ims, tgs = _to_dev(images, targets)
    runs = []
    for _ in range(2):
        loss = box(ims, [dict(t) for t in tgs])
        opt.zero_grad()
        loss.backward()
        runs.append((loss.item(), loss.per_term.cpu().clone(),
                     OrderedDict((n, p.grad.clone()) for n, p in student.named_parameters() if p.requires_grad)))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    for n in runs[0][2]:
        assert torch.equal(runs[0][2][n], runs[1][2][n]), n
    paths = ['backbone.body.layer%d' % i for i in range(1, 5)]
    maps16 = [module_util.get_module(teacher, p).__dict__['distillation_box']['output'].clone() for p in paths]
    for i in (0, 7, 15):
        with torch.no_grad():
            teacher([ims[i]])
        for p, m16 in zip(paths, maps16):
            one = module_util.get_module(teacher, p).__dict__['distillation_box']['output']
            assert torch.equal(one[0], m16[i]), (p, i)


def test_checkpoint_resume_continues_the_same_trajectory(tmp_path):
    """save_ckpt / load_ckpt (reference src/models/__init__.py:11-35) round-trip model + FusedAdam + scheduler state:
    a resumed run takes bit-identical steps (deterministic kernels)."""
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.models import load_ckpt, save_ckpt
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from hnd_ghnd_object_detectors_amd.utils import main_util
    z, meta = G.load('tiny_hnd_faster')
    images, targets = G.case_inputs(meta)

    def fresh():
        cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
        sched = func_util.get_scheduler(opt, 'MultiStepLR', {'milestones': [5, 15], 'gamma': 0.1})
        return cfg, teacher, student, box, opt, sched

    def run(box, opt, n):
        out = []
        for _ in range(n):
            ims, tgs = _to_dev(images, targets)
            loss = box(ims, tgs)
            opt.zero_grad()
            loss.backward()
            opt.step()
            out.append(loss.item())
        return out

    cfg, teacher, student, box, opt, sched = fresh()
    ref = run(box, opt, 4)                                   # uninterrupted: 4 steps
    cfg, teacher, student, box, opt, sched = fresh()
    first = run(box, opt, 2)
    path = str(tmp_path / 'ckpt.pt')
    save_ckpt(student, opt, sched, 0.0, cfg, None, path)
    cfg, teacher, student, box, opt, sched = fresh()         # new process-like state, then resume
    load_ckpt(path, model=student, optimizer=opt, lr_scheduler=sched)
    second = run(box, opt, 2)
    assert first + second == ref, (first, second, ref)
    sd = torch.load(path, weights_only=False)
    assert set(sd['optimizer']['state'][0].keys()) >= {'step', 'exp_avg', 'exp_avg_sq'}


def test_mimic_runner_cli_end_to_end(tmp_path, capsys):
    """the runner (reference CLI: --config / --json / -distill) on synthetic batches: two epochs, checkpoint
    written by rank 0, second invocation resumes optimizer + scheduler from it."""
    import json
    import os
    from hnd_ghnd_object_detectors_amd import mimic_runner
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg_path = os.path.join(root, 'config', 'hnd', 'faster_rcnn-backbone_resnet50-b3ch.yaml')
    ckpt = str(tmp_path / 'student.pt')
    override = {'teacher_model': {'backbone': {'params': {'pretrained': False}},
                                  'params': {'pretrained': False, 'min_size': 64, 'max_size': 128},
                                  'ckpt': str(tmp_path / 'none.pt')},
                'student_model': {'backbone': {'params': {'pretrained': False}},
                                  'params': {'pretrained': False, 'min_size': 64, 'max_size': 128}, 'ckpt': ckpt},
                'train': {'batch_size': 2, 'log_freq': 1}}
    argv = ['--config', cfg_path, '--json', json.dumps(override), '-distill', '--synthetic_batches', '3',
            '--image_size', '64x96', '--num_epochs', '2']
    torch.manual_seed(0)
    mimic_runner.main(mimic_runner.get_argparser().parse_args(argv))
    out = capsys.readouterr().out
    assert 'Updatable parameters' in out and 'Epoch: [1]' in out and 'Updating ckpt' in out
    ck = torch.load(ckpt, weights_only=False)
    # the reference's six keys, plus this build's own key for the no-validation-set (synthetic) selection rule
    assert sorted(ck) == ['args', 'best_loss', 'best_value', 'config', 'lr_scheduler', 'model', 'optimizer']
    assert ck['best_value'] == 0.0 and ck['best_loss'] > 0
    assert ck['lr_scheduler']['last_epoch'] == 1 or ck['lr_scheduler']['last_epoch'] == 0
    assert len(ck['model']) == 293 and len(ck['optimizer']['state']) == 25
    # resume: the checkpoint is picked up (model via get_model, optimizer/scheduler via distill)
    mimic_runner.main(mimic_runner.get_argparser().parse_args(argv[:-1] + ['1', '-decoded_input']))
    out = capsys.readouterr().out
    assert 'Loading model parameters' in out and 'Loading optimizer parameters' in out


# ------------------------------------------------------------------------------------------ neural filter (8f-f2)
def _ext_sync(orc, model):
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in orc.s.items():
            if k in sd:
                v.copy_(sd[k].detach().cpu().to(v.dtype))


def test_neural_filter_training_matches_reference_golden():
    """ext_runner-style steps (frozen detector, Ext4ResNet on the stem output, cross entropy, SGD momentum + wd,
    warm-up) against the fixture the reference's own ext model produced; gradients judged against the fp64 oracle."""
    from hnd_ghnd_object_detectors_amd import ext_runner
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from hnd_ghnd_object_detectors_amd.utils import main_util
    z, meta = G.load('tiny_ext_filter')
    s_sd, e_sd = MU.ext_states(meta['seed'])
    cfg, model, ext = MU.build_ext_model(s_sd, e_sd, DEV, meta['min_size'], meta['max_size'])
    opt_cfg = cfg['train']['optimizer']
    opt = func_util.get_optimizer(ext, opt_cfg['type'], opt_cfg['params'])
    warm = main_util.warmup_lr_scheduler(opt, meta['loader_len'] - 1, 1e-3)
    kw = dict(min_size=(meta['min_size'],), max_size=meta['max_size'])
    orc64 = O.FilterOracle(s_sd, e_sd, dtype=torch.float64, **kw)
    orc32 = O.FilterOracle(s_sd, e_sd, **kw)
    images, targets = G.ext_case_inputs(meta)
    model.train()
    names = {n: p for n, p in model.named_parameters() if p.requires_grad}
    assert len(names) == 14
    for step in range(meta['steps']):
        for orc in (orc64, orc32):
            _ext_sync(orc, model)
        ims, tgs = _to_dev(images, [{k: v.clone() for k, v in t.items()} for t in targets])
        logits = model(ims, tgs)
        labels = ext_runner.convert_target2ext_targets(tgs, DEV)
        assert labels.tolist() == z['step%d/labels' % step].tolist()
        loss = cross_entropy(logits, labels)              # the product kernel (hnd_softmax_ce_rows_fwd_bwd)
        opt.zero_grad()
        loss.backward()
        assert abs(opt.param_groups[0]['lr'] - float(z['step%d/lr' % step])) < 1e-15
        G.compare(z, 'step%d/logits' % step, logits.detach(), FEAT_TOL)
        assert abs(float(loss.detach()) - float(z['step%d/loss' % step])) <= LOSS_TOL * float(z['step%d/loss' % step])
        _, _, g64, _ = orc64.step(images, targets)
        _, _, g32, _ = orc32.step(images, targets)
        for n, p in names.items():
            if n in O.EXT_ZERO_GRAD_KEYS:       # true gradient 0: compare magnitudes only
                assert float(p.grad.abs().max()) < 1e-5
                continue
            _grad_check(n, p.grad, g32[n], g64[n])
            G.compare(z, 'step%d/grad/%s' % (step, n[len(O.EXT):]), p.grad, 5e-3)
        opt.step()
        warm.step()
        for n, p in names.items():
            G.compare(z, 'step%d/param_after/%s' % (step, n[len(O.EXT):]), p.detach(), 1e-4)
    sd = model.state_dict()
    for k in z.files:                           # BatchNorm buffers of the filter AND of layer1 (updated as written)
        if k.startswith('buffers/'):
            # atol: running means that are analytically 0 (a bias-free conv of a zero-mean BN output) are fp32 noise
            G.compare(z, k, sd[k[len('buffers/'):]].float(), 1e-3, atol=1e-4)
    assert int(sd[O.EXT + 'extractor.2.num_batches_tracked']) == meta['steps']
    assert int(sd[O.B + 'layer1.decoder.0.num_batches_tracked']) == meta['steps']
    # eval: softmax probabilities; batch-1 gate of ExtEncoder
    model.eval()
    with torch.no_grad():
        ims, tgs = _to_dev(images, targets)
        probs = model(ims, tgs)
        single = model(ims[:1], tgs[:1])
    G.compare(z, 'eval/probs', probs, FEAT_TOL)
    G.compare(z, 'eval/probs_single', single, FEAT_TOL)
    layer1 = model.backbone.body.layer1
    x0 = model.backbone.body.stem().forward(model.transform(ims[:1], None, None)[0].tensors._hnd, False)
    from hnd_ghnd_object_detectors_amd.hipnn import attach
    from hnd_ghnd_object_detectors_amd import engine as E
    layer1.encoder.threshold = 0.999            # reject: the encoder/decoder are skipped
    out, ext_z = layer1(attach(E.logical(x0), x0))
    assert out is None and tuple(ext_z.shape) == (1, 2)
    layer1.encoder.threshold = 0.0              # accept: features come back with the probabilities
    out, ext_z = layer1(attach(E.logical(x0), x0))
    assert out is not None and out.shape[1] == 256 and abs(float(ext_z.sum()) - 1.0) < 1e-5


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_neural_filter_full_size_parity_against_the_oracle():
    """VERDICT r3 weak #1 (BASELINE config 5's filter half): the neural filter at 3x800x1333, batch 2, config
    ext/keypoint_rcnn-...-b3ch -- the geometry the <= 128 px fixture cannot reach: a 200 x 336 stem map pooled to
    64 x 64 through NON-UNIFORM adaptive windows (3-4 rows x 5-6 columns), images of different sizes padded into one
    batch.  The CPU oracle (pinned to the reference's own ext model by tiny_ext_filter; it travels to the GPU box) runs
    the same step on the host in fp32 and fp64: logits, loss, the 14 gradients (_grad_check), parameters after one SGD
    step (momentum, weight decay, warm-up lr), BatchNorm buffers of the filter AND of layer1, eval probabilities and
    the batch-1 gate."""
    from hnd_ghnd_object_detectors_amd import ext_runner
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from hnd_ghnd_object_detectors_amd.utils import main_util
    seed, min_size, max_size = 23, 800, 1333
    s_sd, e_sd = MU.ext_states(seed)
    cfg, model, ext = MU.build_ext_model(s_sd, e_sd, DEV, min_size, max_size)
    opt_cfg = cfg['train']['optimizer']
    opt = func_util.get_optimizer(ext, opt_cfg['type'], opt_cfg['params'])
    warm = main_util.warmup_lr_scheduler(opt, 10, 1e-3)
    g = torch.Generator().manual_seed(99)
    sizes = [(800, 1333), (720, 1280)]            # the second is resized to 750 x 1333 and zero-padded to 800 x 1344
    images, targets = [], []
    for i, (h, w) in enumerate(sizes):
        images.append(torch.rand(3, h, w, generator=g))
        kp = torch.rand(1, 17, 3, generator=g) * torch.tensor([w, h, 1.0])
        kp[..., 2] = 1.0
        box = [[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]] if i == 0 else [[3.0, 4.0, 0.5, 20.0]]      # image 1: label 0
        targets.append({'boxes': torch.tensor(box), 'labels': torch.tensor([1]), 'keypoints': kp})
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    kw = dict(min_size=(min_size,), max_size=max_size, lr=opt_cfg['params']['lr'],
              momentum=opt_cfg['params'].get('momentum', 0), weight_decay=opt_cfg['params'].get('weight_decay', 0),
              warmup_iters=10, warmup_factor=1e-3)
    orc32 = O.FilterOracle(s_sd, e_sd, **kw)
    orc64 = O.FilterOracle(s_sd, e_sd, dtype=torch.float64, **kw)
    model.train()
    names = {n: p for n, p in model.named_parameters() if p.requires_grad}
    assert len(names) == 14
    ims, tgs = _to_dev(images, [{k: v.clone() for k, v in t.items()} for t in targets])
    logits = model(ims, tgs)
    labels = ext_runner.convert_target2ext_targets(tgs, DEV)
    assert labels.tolist() == [1, 0]
    loss = cross_entropy(logits, labels)              # the product kernel (hnd_softmax_ce_rows_fwd_bwd)
    opt.zero_grad()
    loss.backward()
    o_loss, o_logits, g32, o_lr = orc32.step(images, targets)
    _, _, g64, _ = orc64.step(images, targets)
    assert abs(opt.param_groups[0]['lr'] - o_lr) < 1e-15
    e_logits = _rel(logits, o_logits)
    e_loss = abs(float(loss.detach()) - o_loss) / abs(o_loss)
    assert e_logits < FEAT_TOL and e_loss < LOSS_TOL, (e_logits, e_loss)
    worst_g = 0.0
    for n, p in names.items():
        if n in O.EXT_ZERO_GRAD_KEYS:       # true gradient 0 (a conv bias in front of a train-mode BN)
            assert float(p.grad.abs().max()) < max(1e-5, 10.0 * float(g32[n].abs().max())), n
            continue
        worst_g = max(worst_g, _grad_check(n, p.grad, g32[n], g64[n]))
    opt.step()
    warm.step()
    worst_p = max(_rel(p, orc32.s[n]) for n, p in names.items())
    assert worst_p < 1e-4, worst_p
    sd = model.state_dict()
    worst_b = 0.0
    for k, v in orc32.s.items():            # running statistics of the filter's BNs and of layer1's (updated as written)
        if 'running_' in k and (k.startswith(O.EXT) or 'layer1' in k):
            ref = v.detach().double()
            err = float((sd[k].cpu().double() - ref).norm() / (ref.norm() + 1e-4 * ref.numel() ** 0.5))
            worst_b = max(worst_b, err)
            assert err < 1e-3, (k, err)
    assert int(sd[O.EXT + 'extractor.2.num_batches_tracked']) == 1
    assert int(sd[O.B + 'layer1.decoder.0.num_batches_tracked']) == 1
    # eval: softmax probabilities of the batch, and the batch-1 gate path
    model.eval()
    with torch.no_grad():
        probs = model(ims, tgs)
        single = model(ims[1:], tgs[1:])
    o_probs = orc32.forward(images, training=False, update_buffers=False).detach()
    o_single = orc32.forward(images[1:], training=False, update_buffers=False).detach()
    e_probs, e_single = _rel(probs, o_probs), _rel(single, o_single)
    assert e_probs < FEAT_TOL and e_single < FEAT_TOL, (e_probs, e_single)
    assert tuple(single.shape) == (1, 2) and abs(float(single.sum()) - 1.0) < 1e-5
    from tests.conftest import record_achieved
    record_achieved('[neural filter, 3x800x1333 batch 2 (stem map 200x336 -> 64x64 adaptive windows)] logits %.1e, loss '
                    '%.1e, worst gradient vs fp64 %.2e, parameters after SGD %.1e, BN buffers %.1e, eval probabilities '
                    '%.1e / batch-1 gate %.1e' % (e_logits, e_loss, worst_g, worst_p, worst_b, e_probs, e_single))


def test_quantized_eval_full_size_keypoint_student_against_the_oracle():
    """VERDICT r3 weak #1, second half: the int8-quantised bottleneck of BASELINE config 5 at full size.  Keypoint
    R-CNN student in eval mode at 3x800x1333 (batch 2, padded 800 x 1344; bottleneck z [2, 3, 204, 340]) with and
    without the uint8 codec between encoder and decoder, against the oracle's eval forward (Quantizer -> Dequantizer of
    src/structure/transformer.py:131-153 restated in oracle/myutils_r.py): every hooked map and pyramid level.  A
    bottleneck value within rounding of a bin boundary may land in the neighbouring bin (1/255 of the range) -- hence
    2e-2 on the quantised features, as in the tiny fixture test; the plain ones hold 1e-3."""
    cfg = MU.config_for(model='keypoint_rcnn', method='ghnd', min_size=800, max_size=1333)
    t_sd, s_sd = MU.oracle_states(31, 'keypoint_rcnn', num_classes=2)
    for k in list(s_sd):                    # running statistics away from (0, 1): eval mode is a real check
        if 'layer1' in k and k.endswith('running_var'):
            s_sd[k] = s_sd[k] * 1.7 + 0.1
        if 'layer1' in k and k.endswith('running_mean'):
            s_sd[k] = s_sd[k] + 0.05
    _, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    student.eval()
    g = torch.Generator().manual_seed(7)
    images = [torch.rand(3, 800, 1333, generator=g), torch.rand(3, 720, 1280, generator=g)]
    ims = [im.to(DEV) for im in images]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sd = O.cast_state(s_sd, torch.float32)
    with torch.no_grad():
        x, _ = O.transform_images(images, (800,), 1333, training=False)
    achieved = {}
    for tag, bits, tol in (('plain', None, FEAT_TOL), ('quantized', 8, 2e-2)):
        student.backbone.body.layer1.use_bottleneck_transformer = bits is not None
        with torch.no_grad():
            feats = student(ims)
            _, o_feats = O.backbone_forward(x, sd, student=True, training=False, update_buffers=False,
                                            codec_bits=bits)
        assert [str(k) for k in feats] == [str(k) for k in o_feats]
        worst = 0.0
        for (k, v), (_, ov) in zip(feats.items(), o_feats.items()):
            assert tuple(v.shape) == tuple(ov.shape), (k, v.shape, ov.shape)
            worst = max(worst, _rel(v, ov))
        achieved[tag] = worst
        assert worst < tol, (tag, worst)
    from tests.conftest import record_achieved
    record_achieved('[quantised eval, Keypoint student, 3x800x1333 batch 2, bottleneck [2,3,204,340]] pyramid vs oracle: '
                    'plain %.1e (tol %.0e), uint8 codec %.1e (tol 2e-2)' % (achieved['plain'], FEAT_TOL,
                                                                             achieved['quantized']))


def test_ext_runner_cli_end_to_end(tmp_path, capsys):
    """reference CLI (--config / --json / -train) on synthetic batches: one epoch of filter training, ROC-AUC
    validation, checkpoint with the classifier only, resume on the second invocation."""
    import json
    import os
    from hnd_ghnd_object_detectors_amd import ext_runner
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg_path = os.path.join(root, 'config', 'ext', 'keypoint_rcnn-backbone_ext_resnet50-b3ch.yaml')
    ckpt = str(tmp_path / 'ext.pt')
    override = {'model': {'backbone': {'params': {'pretrained': False}, 'ext_config': {'ckpt': ckpt}},
                          'params': {'pretrained': False, 'min_size': 64, 'max_size': 128},
                          'ckpt': str(tmp_path / 'none.pt')},
                'train': {'batch_size': 4, 'log_freq': 1}, 'test': {'batch_size': 2}}
    argv = ['--config', cfg_path, '--json', json.dumps(override), '-train', '--synthetic_batches', '3',
            '--image_size', '64x96', '--num_epochs', '1']
    torch.manual_seed(0)
    ext_runner.main(ext_runner.get_argparser().parse_args(argv))
    out = capsys.readouterr().out
    assert 'Updatable parameters' in out and 'ROC-AUC' in out and 'Updating ckpt' in out and '[Test]' in out
    ck = torch.load(ckpt, weights_only=False)
    assert len(ck['model']) == 23 and all(not k.startswith('backbone') for k in ck['model'])
    assert len(ck['optimizer']['state']) == 14
    ext_runner.main(ext_runner.get_argparser().parse_args(argv))
    out = capsys.readouterr().out
    assert 'Loading model parameters' in out and 'Loading optimizer parameters' in out


def test_mimic_runner_two_ranks_share_one_gpu(tmp_path):
    """the distributed runner end to end: two processes under torch.distributed.run (gloo, both on cuda:0 -- the
    multi-GPU code path minus RCCL): broadcast start, flat gradient all-reduce folded into Adam, rank-0 checkpoint."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg_path = os.path.join(root, 'config', 'ghnd', 'faster_rcnn-backbone_resnet50-b3ch.yaml')
    ckpt = str(tmp_path / 'student.pt')
    override = {'teacher_model': {'backbone': {'params': {'pretrained': False}},
                                  'params': {'pretrained': False, 'min_size': 64, 'max_size': 128},
                                  'ckpt': str(tmp_path / 'none.pt')},
                'student_model': {'backbone': {'params': {'pretrained': False}},
                                  'params': {'pretrained': False, 'min_size': 64, 'max_size': 128}, 'ckpt': ckpt},
                'train': {'batch_size': 2, 'log_freq': 1}}
    # two processes time-share the one GPU of a test box: one HIP stream each (no teacher side stream), so the pair
    # stays within the hardware queues and the time slicing cannot degenerate (seen once: 15+ min instead of 5 s)
    env = dict(os.environ, HND_DIST_BACKEND='gloo', HND_SHARE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0',
               HND_TEACHER_STREAM='0')
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), '-m', 'hnd_ghnd_object_detectors_amd.mimic_runner', '--config',
           cfg_path, '--json', json.dumps(override), '-distill', '--synthetic_batches', '3', '--image_size', '64x96',
           '--num_epochs', '1', '--world_size', '2']
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=2400)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert 'distributed init (rank 0)' in res.stdout and 'Updating ckpt' in res.stdout
    ck = torch.load(ckpt, weights_only=False)
    assert len(ck['model']) == 293 and len(ck['optimizer']['state']) == 25
    assert all(torch.isfinite(v).all() for v in ck['model'].values() if v.is_floating_point())


def test_two_ranks_take_the_oracle_step_on_the_mean_gradient(tmp_path):
    """DDP arithmetic (reference src/mimic_runner.py:141-143 + :52-54): 2 ranks, each its own seeded batch and LOCAL
    BatchNorm statistics, reference loop verbatim (no explicit reduce call).  After every step BOTH ranks must hold
    bit-identical parameters, and those must equal the oracle's Adam step on the MEAN of the two ranks' oracle
    gradients (each rank's gradient computed by the CPU oracle on that rank's batch from the same weights)."""
    import subprocess
    import socket
    import sys
    from tests.ddp_worker import rank_batch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    steps = 2
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', HND_TEACHER_STREAM='0')      # see the shared-GPU note above
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(root, 'tests', 'ddp_worker.py'), str(tmp_path),
           str(steps)]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=2400)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    r0 = torch.load(str(tmp_path / 'rank0.pt'), weights_only=False)
    r1 = torch.load(str(tmp_path / 'rank1.pt'), weights_only=False)
    assert r0['reductions'] == r1['reductions'] == steps           # one flat all-reduce per backward, fired by the hook
    _, meta = G.load('tiny_ghnd_faster')
    cfg = MU.config_for(meta)
    terms = MU.terms_of(cfg)
    t_sd, s_sd = MU.oracle_states(meta['seed'])
    kw = dict(terms=terms, min_size=(meta['min_size'],), max_size=meta['max_size'])
    ranks = [O.DistillOracle(t_sd, s_sd, **kw) for _ in range(2)]             # per-rank gradient + local BN buffers
    keys = ranks[0].keys
    mean_p = [ranks[0].s[k].detach().clone().requires_grad_(True) for k in keys]
    adam = torch.optim.Adam(mean_p, lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(adam, lambda x: 1 if x >= 4 else 1e-3 * (1 - x / 4.0) + x / 4.0)
    batches = [rank_batch(meta, r)[0] for r in range(2)]
    worst, worst_g = 0.0, 0.0
    # the exchange itself, bit for bit: single-process HIP gradients of each rank's batch from the same initial
    # weights (deterministic kernels) must SUM to exactly what both ranks hold after the all-reduce of step 0
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    box = DistillationBox(teacher, student, cfg['train']['criterion'])
    local = []
    for r in range(2):
        ims, tgs = _to_dev(*rank_batch(meta, r))
        for p in student.parameters():
            p.grad = None
        box(ims, tgs).backward()
        local.append({n: p.grad.detach().cpu().clone() for n, p in student.named_parameters() if p.requires_grad})
    for k in keys:
        assert torch.equal(r0['history'][0]['grad_sum'][k], local[0][k] + local[1][k]), k
    for step in range(steps):
        grads = []
        for r, orc in enumerate(ranks):
            with torch.no_grad():
                for k, p in zip(keys, mean_p):
                    orc.s[k].copy_(p)
            loss, *_ = orc.forward(batches[r])
            gs = torch.autograd.grad(loss, [orc.s[k] for k in keys])
            grads.append(gs)
            got = (r0, r1)[r]['history'][step]['loss']
            assert abs(got - float(loss)) / abs(float(loss)) < LOSS_TOL, (step, r, got, float(loss))
        adam.zero_grad()
        for k, p, g0, g1 in zip(keys, mean_p, grads[0], grads[1]):
            p.grad = (g0 + g1) * 0.5
            if not k.endswith(G.ZERO_GRAD_SUFFIXES):      # the exchanged gradient: sum over ranks, identical on both
                gs0, gs1 = r0['history'][step]['grad_sum'][k], r1['history'][step]['grad_sum'][k]
                assert torch.equal(gs0, gs1), (step, k)
                e = float((gs0.double() - (g0 + g1).double()).norm() / (g0 + g1).double().norm())
                worst_g = max(worst_g, e)
                assert e < 2e-2, (step, k, e)       # fp32 oracle vs fp32 HIP at 64 px: ReLU-mask flips on either side
        # a ReLU-mask flip against the oracle run (see test_distill_steps_match_reference_golden) costs ~1e-3 on the
        # gradients and several 1e-3 on the small BN biases after Adam: 2e-3 on parameters only when no flip happened
        ptol = 2e-3 if worst_g < 1e-4 else 2e-2
        adam.step()
        sched.step()
        for k, p in zip(keys, mean_p):
            a, b = r0['history'][step]['params'][k], r1['history'][step]['params'][k]
            assert torch.equal(a, b), (step, k)                     # both ranks: the same update, bit for bit
            if k.endswith(G.ZERO_GRAD_SUFFIXES):
                continue
            err = float((a.double() - p.detach().double()).norm() / (p.detach().double().norm() + 1e-6 * p.numel() ** 0.5))
            worst = max(worst, err)
            assert err < ptol, (step, k, err)
    print('\n[ddp] worst relative error vs the oracle mean-gradient step: parameters %.2e, exchanged gradients %.2e'
          % (worst, worst_g))


def test_mimic_runner_on_coco_format_folder(tmp_path, capsys, monkeypatch):
    """no --synthetic_batches: the runner reads the COCO-format folder named by the yaml (json index + PIL, no
    pycocotools), batches by aspect ratio, ships uint8 images and runs the fused device transform; after every epoch
    it validates the student (eval-mode detector on the HIP path -> COCO bbox mAP) and keeps the checkpoint on the
    best mAP (reference src/mimic_runner.py:92-100), then evaluates teacher and student on the test split (:148-150).
    Random weights detect nothing, so the recorded mAP is nudged upward per call to exercise the selection rule."""
    from hnd_ghnd_object_detectors_amd.utils import main_util as MUTIL
    real_evaluate, calls = MUTIL.evaluate, []

    def rising_evaluate(model, data_loader, device):
        ev = real_evaluate(model, data_loader, device)          # the real detector + evaluator run
        assert 0.0 <= ev.coco_eval['bbox'].stats[0] <= 1.0 or ev.coco_eval['bbox'].stats[0] == -1
        calls.append(float(ev.coco_eval['bbox'].stats[0]))
        ev.coco_eval['bbox'].stats[0] = 0.01 * len(calls)
        return ev
    monkeypatch.setattr(MUTIL, 'evaluate', rising_evaluate)
    import json
    import os
    from tests.coco_fixture import write_tiny_coco
    from hnd_ghnd_object_detectors_amd import mimic_runner
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    img_dir, ann_file = write_tiny_coco(str(tmp_path / 'coco'))
    cfg_path = os.path.join(root, 'config', 'ghnd', 'keypoint_rcnn-backbone_resnet50-b3ch.yaml')
    ckpt = str(tmp_path / 'student.pt')
    split = {'images': img_dir, 'annotations': ann_file}
    small = {'pretrained': False, 'min_size': [48, 56, 64], 'max_size': 128}
    override = {'dataset': {'num_workers': 0, 'splits': {'train': split, 'val': split, 'test': split}},
                'teacher_model': {'backbone': {'params': {'pretrained': False}}, 'params': small,
                                  'ckpt': str(tmp_path / 'none.pt')},
                'student_model': {'backbone': {'params': {'pretrained': False}}, 'params': small, 'ckpt': ckpt},
                'train': {'batch_size': 2, 'log_freq': 1}}
    argv = ['--config', cfg_path, '--json', json.dumps(override), '-distill', '--num_epochs', '2']
    torch.manual_seed(0)
    random.seed(0)
    mimic_runner.main(mimic_runner.get_argparser().parse_args(argv))
    out = capsys.readouterr().out
    assert 'Creating data loaders' in out and 'Epoch: [1]' in out
    assert out.count('Updating ckpt (Best BBox mAP') == 2 and 'IoU metric: bbox' in out
    assert 'Average Precision  (AP) @[ IoU=0.50:0.95 | area=   all | maxDets=100 ]' in out
    assert '[Teacher model]' in out and '[Student model]' in out and len(calls) == 4      # 2 x val, teacher, student
    assert 'Count of instances per bin' in out
    ck = torch.load(ckpt, weights_only=False)
    assert all(torch.isfinite(v).all() for v in ck['model'].values() if v.is_floating_point())
    # same data through the reference's host-side float conversion gives the same first-epoch loss trajectory
    torch.manual_seed(0)
    random.seed(0)
    os.remove(ckpt)
    mimic_runner.main(mimic_runner.get_argparser().parse_args(argv + ['-host_float_input']))
    out2 = capsys.readouterr().out

    def losses(text):
        return [float(l.split('loss: ')[1].split(' ')[0]) for l in text.splitlines() if 'loss: ' in l and 'Epoch: [0]' in l]
    a, b = losses(out), losses(out2)
    assert len(a) == len(b) > 0 and all(abs(x - y) <= 1e-4 * abs(y) for x, y in zip(a, b)), (a, b)


def test_hipgraph_replay_of_the_filter_training_step_is_bit_identical():
    """graph.GraphedStep: the neural-filter training step (forward, hand-written backward, fused SGD, the torch
    cross-entropy around them) captured into a hipGraph and replayed takes the SAME trajectory as eager launches, bit
    for bit -- prebuilt descriptors on static buffers are exactly what a graph records."""
    from hnd_ghnd_object_detectors_amd import ext_runner
    from hnd_ghnd_object_detectors_amd.graph import GraphedStep
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    z, meta = G.load('tiny_ext_filter')
    images, targets = G.ext_case_inputs(meta)
    ims, tgs = _to_dev(images, targets)

    def run(graphed):
        s_sd, e_sd = MU.ext_states(meta['seed'])
        cfg, model, ext = MU.build_ext_model(s_sd, e_sd, DEV, meta['min_size'], meta['max_size'])
        model.train()
        opt = func_util.get_optimizer(ext, 'SGD', {'lr': 1e-3, 'momentum': 0.9, 'weight_decay': 1e-4})
        with torch.no_grad():
            probe = [dict(t) for t in tgs]
            model.transform(ims, probe, None)
        labels = ext_runner.convert_target2ext_targets(probe, DEV)

        def body():
            logits = model(ims, [dict(t) for t in tgs])
            loss = cross_entropy(logits, labels)              # the product kernel (hnd_softmax_ce_rows_fwd_bwd)
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss.detach()
        step = GraphedStep(body, key=lambda: opt.param_groups[0]['lr'], warmup=2) if graphed else body
        losses = [float(step()) for _ in range(6)]
        if graphed:
            assert step.captures == 1
            opt.param_groups[0]['lr'] = 5e-4                # a host-side hyper-parameter change re-captures
            losses.append(float(step()))
            assert step.captures == 2
        else:
            opt.param_groups[0]['lr'] = 5e-4
            losses.append(float(step()))
        params = OrderedDict((n, p.detach().clone()) for n, p in ext.named_parameters())
        return losses, params
    l0, p0 = run(False)
    l1, p1 = run(True)
    assert l0 == l1, (l0, l1)
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n


def test_ext_model_eval_with_quantised_bottleneck_equals_plain_student():
    """BASELINE config 5: Keypoint R-CNN b3ch + neural filter + int8 bottleneck.  In eval mode the filter model's
    pyramid (filter accepts) equals the filter-less student's with the same weights bit for bit, with and without
    the uint8 codec; the filter's probabilities ride along."""
    meta = {'model': 'keypoint_rcnn', 'yaml': 'ghnd/', 'min_size': 64, 'max_size': 128, 'seed': 31, 'num_classes': 2}
    s_sd, e_sd = MU.ext_states(meta['seed'])
    cfg_e, ext_model, ext = MU.build_ext_model(s_sd, e_sd, DEV, 64, 128, threshold=0.0)
    ext_model.ext_training = False
    ext_model.backbone.body.ext_training = False
    cfg = MU.config_for(meta)
    t_sd = O.init_teacher_state(meta['seed'], 'keypoint_rcnn', num_classes=2)
    _, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    for m in (ext_model, student):
        m.eval()
        m.distill_backbone_only = True
    g = torch.Generator().manual_seed(3)
    ims = [torch.rand(3, 60, 90, generator=g).to(DEV), torch.rand(3, 64, 80, generator=g).to(DEV)]
    assert ext_model.backbone.body.layer1.bottleneck_transformer is not None          # yaml: quantizer + dequantizer
    for use in (False, True):
        ext_model.backbone.body.layer1.use_bottleneck_transformer = use
        student.backbone.body.layer1.use_bottleneck_transformer = use
        with torch.no_grad():
            feats_e, probs = ext_model(ims)
            feats_e = {k: v.clone() for k, v in feats_e.items()}
            feats_s = student(ims)
        assert tuple(probs.shape) == (2, 2) and float((probs.sum(dim=1) - 1).abs().max()) < 1e-5
        assert list(feats_e) == list(feats_s) and all(torch.equal(feats_e[k], feats_s[k]) for k in feats_s)
    plain = {k: v.clone() for k, v in feats_s.items()}
    student.backbone.body.layer1.use_bottleneck_transformer = False
    with torch.no_grad():
        ref = student(ims)
    assert any(not torch.equal(plain[k], ref[k]) for k in ref)            # the codec really changed the features


@pytest.mark.parametrize('sizes,min_size,max_size', [([(20, 24)], 32, 64),            # batch 1, layer4 is 1x1
                                                     ([(33, 200), (40, 40)], 32, 192),   # 1:6 strip next to a square
                                                     ([(97, 65), (64, 128), (31, 31), (80, 50), (50, 80)], 64, 100)])
def test_distill_step_edge_geometries_match_oracle(sizes, min_size, max_size):
    """batch 1, feature maps down to 1x1, extreme aspect ratios, ragged batches with heavy padding: one full
    step (features, loss, gradients) against the fp32 / fp64 oracle on the same inputs."""
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    cfg = MU.config_for(model='faster_rcnn', method='ghnd', bch=3, min_size=min_size, max_size=max_size)
    t_sd, s_sd = MU.oracle_states(55)
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    box = DistillationBox(teacher, student, cfg['train']['criterion'])
    opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
    kw = dict(min_size=(min_size,), max_size=max_size)
    orc32, orc64 = O.DistillOracle(t_sd, s_sd, **kw), O.DistillOracle(t_sd, s_sd, dtype=torch.float64, **kw)
    g = torch.Generator().manual_seed(8)
    images = [torch.rand(3, h, w, generator=g) for h, w in sizes]
    targets = [{'boxes': torch.tensor([[1., 2., 10., 12.]]), 'labels': torch.tensor([1])} for _ in images]
    ims, tgs = _to_dev(images, targets)
    loss = box(ims, tgs)
    opt.zero_grad()
    loss.backward()
    ref_loss, per_term, g32, _ = orc32.step(images)
    _, _, g64, _ = orc64.step(images)
    assert abs(float(loss.detach()) - ref_loss) <= LOSS_TOL * ref_loss
    _, _, t_h, s_h, _, _, _ = O.DistillOracle(t_sd, s_sd, **kw).forward(images, update_buffers=False)
    for k in ('layer1', 'layer2', 'layer3', 'layer4'):
        got = _hooked(student, 'backbone.body.' + k).detach().cpu()
        assert tuple(got.shape) == tuple(s_h[k].shape)
        ref = s_h[k].detach()
        assert float((got - ref).abs().max() / ref.abs().max()) < FEAT_TOL, k
    names = {n: p for n, p in student.named_parameters() if p.requires_grad}
    for n, p in names.items():
        if n.endswith(G.ZERO_GRAD_SUFFIXES):
            continue
        _grad_check(n, p.grad, g32[n], g64[n], tol=6e-3)


@pytest.mark.parametrize('native', [False, True])
def test_rccl_exchange_world_of_one(tmp_path, native):
    """the gradient exchange on a REAL RCCL communicator (one rank: the one GPU of a test box): the hook-fired async
    all-reduce of the flat arena -> stream-side wait in FusedAdam -> mean folded into the launch, through
    torch.distributed ('nccl' == RCCL) and through the C ABI's own communicator (hnd_comm_*).  One rank sums to the
    identity, so two steps must reproduce the plain loop bit for bit; the 2-rank arithmetic is
    test_two_ranks_take_the_oracle_step_on_the_mean_gradient (gloo, shared device)."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    out = str(tmp_path / 'r.pt')
    cmd = [sys.executable, os.path.join(root, 'tests', 'rccl_world1_worker.py'), out] + (['native'] if native else [])
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=2400)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    r = torch.load(out, weights_only=False)
    assert r['plain']['losses'] == r['rccl']['losses']
    for k, v in r['plain']['params'].items():
        assert torch.equal(v, r['rccl']['params'][k]), k


def test_jpeg_bottleneck_and_data_logger_run_inside_the_eval_model(tmp_path):
    """get_model(..., bottleneck_transformer=...) with the file's other transformers (reference transformer.py:58-128,
    base.py:34,54-57): a DataLogger leaves the features untouched and records one entry per eval forward; the JPEG
    compressor / decompressor pair changes the features a little (lossy) but not much"""
    from hnd_ghnd_object_detectors_amd.structure import transformer as T
    z, meta = G.load('tiny_eval_quantized')
    cfg = MU.config_for(meta)
    t_sd, s_sd = MU.oracle_states(meta['seed'])
    _, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
    student.eval()
    images, _ = G.case_inputs(meta)
    ims = [im.to(DEV) for im in images[:1]]             # one image: the JPEG codec takes [1, 3, H, W]
    layer1 = student.backbone.body.layer1
    with torch.no_grad():
        layer1.use_bottleneck_transformer = False
        plain = OrderedDict((k, v.clone()) for k, v in student(ims).items())
        log = T.DataLogger()
        layer1.bottleneck_transformer, layer1.data_logging = log, True
        layer1.use_bottleneck_transformer = True
        logged = student(ims)
        for k in plain:
            assert torch.equal(logged[k], plain[k]), k
        sizes, _, quant, shapes = log.get_data()
        assert len(sizes) == 1 and shapes[0][0] == 3 and 0 < quant[0] < sizes[0]
        layer1.bottleneck_transformer = T.Compose([T.JpegCompressor(95, str(tmp_path / 'jpg')),
                                                   T.JpegDecompressor(str(tmp_path / 'jpg'), 4)])
        layer1.data_logging = False
        lossy = student(ims)
        k0 = list(plain)[0]
        rel = float((lossy[k0] - plain[k0]).norm() / plain[k0].norm())
        assert 0.0 < rel < 0.25, rel


def test_coco_runner_evaluates_an_original_detector(tmp_path, capsys):
    """src/coco_runner.py without -train (:131-132): config/org yaml -> get_model -> main_util.evaluate on the test
    split.  Keypoint R-CNN on a COCO-format folder with person keypoints: the eval-mode detector incl. its keypoint
    branch runs on the HIP path, both metrics are summarised; -train is refused (detector losses are not built)."""
    import json
    import os
    from tests.coco_fixture import write_tiny_coco
    from hnd_ghnd_object_detectors_amd import coco_runner
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    img_dir, ann_file = write_tiny_coco(str(tmp_path / 'coco'))
    cfg_path = os.path.join(root, 'config', 'org', 'keypoint_rcnn-backbone_resnet50.yaml')
    split = {'images': img_dir, 'annotations': ann_file}
    override = {'dataset': {'num_workers': 0, 'splits': {'train': split, 'val': split, 'test': split}},
                'model': {'backbone': {'params': {'pretrained': False}}, 'ckpt': str(tmp_path / 'none.pt'),
                          'params': {'pretrained': False, 'min_size': [48, 56, 64], 'max_size': 128}}}
    torch.manual_seed(0)
    ev = coco_runner.main(coco_runner.get_argparser().parse_args(['--config', cfg_path, '--json', json.dumps(override)]))
    out = capsys.readouterr().out
    assert sorted(ev.coco_eval) == ['bbox', 'keypoints'] and 'IoU metric: keypoints' in out
    assert ev.coco_eval['bbox'].stats.shape == (12,) and ev.coco_eval['keypoints'].stats.shape == (10,)
    assert -1.0 <= ev.coco_eval['keypoints'].stats[0] <= 1.0
    with pytest.raises(NotImplementedError):
        coco_runner.main(coco_runner.get_argparser().parse_args(['--config', cfg_path, '--json', json.dumps(override),
                                                                  '-train']))

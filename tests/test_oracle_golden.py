"""CPU: the functional oracle reproduces the reference-generated golden fixtures."""
from collections import OrderedDict

import pytest
import torch

from oracle import hnd_oracle as O
from tests import golden_util as G

TINY = ['tiny_ghnd_faster', 'tiny_hnd_faster', 'tiny_ghnd_mask', 'tiny_ghnd_keypoint', 'tiny_ghnd_faster_b6',
        'tiny_ghnd_custom_hooks', 'tiny_ghnd_fpn_term', 'tiny_enc_term']


def _oracle_for(meta, z):
    t_sd = O.init_teacher_state(meta['seed'], meta['model'], num_classes=meta.get('num_classes', 91))
    s_sd = O.init_student_state(t_sd, meta['seed'] + 1000, bch=meta.get('bch', 3))
    terms = O.HND_TERMS if meta['yaml'].startswith('hnd/') else O.GHND_TERMS
    if 'terms' in meta:             # non-standard hook paths: (teacher key, student key, factor), O.rel_key of the paths
        terms = OrderedDict((name, (O.rel_key(tp), O.rel_key(sp), f)) for name, tp, sp, f in meta['terms'])
    student_arch = meta.get('teacher') == 'student_arch'      # a bottleneck-injected teacher (tiny_enc_term)
    if student_arch:
        t_sd = O.init_student_state(t_sd, meta['seed'] + 500, bch=meta.get('bch', 3))
    ms = meta['min_size'] if isinstance(meta['min_size'], list) else [meta['min_size']]
    return O.DistillOracle(t_sd, s_sd, terms=terms, min_size=tuple(ms), max_size=meta['max_size'],
                           warmup_iters=4, warmup_factor=1e-3, teacher_is_student_arch=student_arch), terms


@pytest.mark.parametrize('name', TINY)
def test_oracle_matches_reference_fixture(name):
    z, meta = G.load(name)
    assert float(z['oracle_vs_reference_maxabs']) < 1e-5
    orc, terms = _oracle_for(meta, z)
    images, _ = G.case_inputs(meta)
    for step in range(meta['steps']):
        fs = [int(v) for v in z['step%d/fixed_sizes' % step]] if meta['model'] == 'keypoint_rcnn' else None
        if step == 0:
            _, _, t_h, s_h, _, s_f, x = orc.forward(images, fs, update_buffers=False)
            G.compare(z, 'transform', x, 1e-6)
            for k, v in terms.items():
                tk, sk = (v[0], v[1]) if isinstance(v, tuple) else (k, k)
                G.compare(z, 'step0/teacher/' + k, t_h[tk], 1e-6)
                G.compare(z, 'step0/student/' + k, s_h[sk], 1e-6)
            if len(terms) == 4 and 'terms' not in meta:
                for k, v in s_f.items():
                    G.compare(z, 'student_fpn/%s' % k, v, 1e-6)
        loss, per_term, grads, lr = orc.step(images, fs)
        assert abs(loss - float(z['step%d/loss' % step])) <= 1e-6 * abs(loss)
        assert abs(lr - float(z['step%d/lr' % step])) < 1e-12
        for k in terms:
            assert abs(per_term[k] - float(z['step%d/term/%s' % (step, k)])) <= 1e-6 * abs(per_term[k])
        for n, g in grads.items():
            if n.endswith(G.ZERO_GRAD_SUFFIXES):
                continue
            G.compare(z, 'step%d/grad/%s' % (step, n), g, 2e-5)
    for n in orc.keys:
        if not n.endswith(G.ZERO_GRAD_SUFFIXES):
            G.compare(z, 'after/param/' + n, orc.s[n], 1e-4, atol=1e-6)  # Adam amplifies grad rounding noise
    for n in z.files:
        if n.startswith('after/buffer/'):
            key = n[len('after/buffer/'):]
            ref = torch.from_numpy(z[n]).double()
            assert float((orc.s[key].double() - ref).abs().max()) <= 1e-5 * (1 + float(ref.abs().max()))


FULL = ['full_ghnd_faster', 'full_ghnd_faster_b4', 'full_hnd_faster_b2', 'full_ghnd_mask_b2', 'full_ghnd_keypoint_b2',
        'full_ghnd_mask_b8']


@pytest.mark.parametrize('name', FULL)
def test_full_size_fixtures_record_oracle_agreement(name):
    """The 3x800x1333 fixtures are too slow to re-derive in the CPU suite (a minute each); make_golden.py ran the
    oracle beside the reference while writing them and stored the achieved agreement, which must hold the bar."""
    z, meta = G.load(name)
    assert meta['full'] and float(z['oracle_vs_reference_maxabs']) < 1e-5
    assert float(z['oracle_vs_reference_grad_rel_l2']) < 2e-4
    images, targets = G.case_inputs(meta)           # the seeded inputs are re-creatable at full size
    assert len(images) == len(meta['sizes']) and tuple(images[0].shape) == (3,) + tuple(meta['sizes'][0])
    nterms = 1 if meta['yaml'].startswith('hnd/') else 4
    assert len([k for k in z.files if k.startswith('step0/term/')]) == nterms


def test_oracle_quantizer_matches_reference_bytes():
    """oracle/myutils_r.quantize_tensor (restated myutils) on the reference's own bottleneck tensor gives
    the bytes the reference's Quantizer produced (fixture written through the reference's structure/transformer.py)."""
    from oracle.myutils_r import dequantize_tensor, quantize_tensor
    z = G.load_raw('tiny_eval_quantized')
    q = quantize_tensor(torch.from_numpy(z['quantized/z']).clone(), 8)
    assert torch.equal(q.tensor, torch.from_numpy(z['quantized/bytes']))
    assert float(q.scale) == float(z['quantized/scale']) and float(q.zero_point) == float(z['quantized/zero_point'])
    assert torch.equal(dequantize_tensor(q), torch.from_numpy(z['quantized/dequantized']))


def test_state_layout():
    t_sd = O.init_teacher_state(0)
    s_sd = O.init_student_state(t_sd, 1)
    keys = O.trainable_keys(s_sd)
    assert len(keys) == 25
    assert sum(s_sd[k].numel() for k in keys) == 586566
    assert len(s_sd) == 293
    assert sum(v.numel() for v in s_sd.values()) == 42222250


def test_fp64_gradient_noise_floor():
    z, meta = G.load('tiny_ghnd_faster')
    orc32, _ = _oracle_for(meta, z)
    t_sd = O.init_teacher_state(meta['seed'])
    s_sd = O.init_student_state(t_sd, meta['seed'] + 1000)
    orc64 = O.DistillOracle(t_sd, s_sd, min_size=(64,), max_size=128, dtype=torch.float64)
    images, _ = G.case_inputs(meta)
    _, _, g32, _ = orc32.step(images)
    _, _, g64, _ = orc64.step(images)
    for n in g32:
        if n.endswith(G.ZERO_GRAD_SUFFIXES):
            assert float(g64[n].abs().max()) < 1e-6      # mathematically zero gradient
            continue
        rel = float((g32[n].double() - g64[n]).norm() / g64[n].norm())
        assert rel < 1e-4, (n, rel)


def test_oracle_input_pipeline_matches_reference_fixture():
    """ToTensor -> RandomHorizontalFlip -> CustomRCNNTransform of the reference on decoded uint8 images."""
    z = G.load_raw('tiny_input_pipeline')
    assert float(z['oracle_vs_reference_maxabs']) < 1e-6
    images = []
    for u8, flip, tin, tout in G.pipeline_case(z):
        img, tgt = O.to_tensor_u8(u8), tin
        if flip:
            img, tgt = O.horizontal_flip(img, tin)
        for k in tout:
            assert torch.equal(tgt[k], tout[k]), k
        images.append(img)
    batch, sizes = O.transform_images(images, (64,), 128)
    assert [list(s) for s in sizes] == z['image_sizes'].tolist()
    assert float((batch - torch.from_numpy(z['batch'])).abs().max()) < 1e-6


def test_filter_oracle_matches_reference_fixture():
    """neural filter: ext_runner-style training steps + eval probabilities of the reference's ext model"""
    z, meta = G.load('tiny_ext_filter')
    assert float(z['oracle_vs_reference_maxabs']) < 1e-5 and float(z['oracle_vs_reference_grad_rel']) < 1e-4
    t_sd = O.init_teacher_state(meta['seed'], 'keypoint_rcnn', num_classes=2)
    s_sd, e_sd = O.init_student_state(t_sd, meta['seed'] + 1000), O.init_ext_state(meta['seed'] + 2000)
    orc = O.FilterOracle(s_sd, e_sd, min_size=(meta['min_size'],), max_size=meta['max_size'],
                         warmup_iters=meta['loader_len'] - 1)
    images, targets = G.ext_case_inputs(meta)
    for step in range(meta['steps']):
        loss, logits, grads, lr = orc.step(images, targets)
        assert [1 if O.valid_target(t) else 0 for t in targets] == z['step%d/labels' % step].tolist()
        assert abs(loss - float(z['step%d/loss' % step])) < 1e-6 and abs(lr - float(z['step%d/lr' % step])) < 1e-15
        G.compare(z, 'step%d/logits' % step, logits, 1e-6)
        for k, g in grads.items():
            if k not in O.EXT_ZERO_GRAD_KEYS:
                G.compare(z, 'step%d/grad/%s' % (step, k[len(O.EXT):]), g, 1e-5)
        for k in orc.keys:
            G.compare(z, 'step%d/param_after/%s' % (step, k[len(O.EXT):]), orc.s[k], 1e-6)
    for k in z.files:
        if k.startswith('buffers/'):
            G.compare(z, k, orc.s[k[len('buffers/'):]].float(), 1e-6)
    G.compare(z, 'eval/probs', orc.forward(images, training=False), 1e-6)
    G.compare(z, 'eval/probs_single', orc.forward(images[:1], training=False), 1e-6)


@pytest.mark.parametrize('name', ['tiny_detect_faster', 'tiny_detect_mask', 'tiny_detect_keypoint'])
def test_detect_oracle_reproduces_the_reference_detections(name):
    """O.DetectOracle (functional backbone + restated torchvision heads, state-dict driven) against what the
    REFERENCE's CustomRCNN.forward produced in eval mode: boxes / labels / scores / keypoints identical, pasted-mask
    bits identical -- the composition the full-size GPU tests use as their checker is the pinned one"""
    import numpy as np
    z, meta = G.load(name)
    kw = {'num_classes': 2} if meta['model'] == 'keypoint_rcnn' else {}
    t_sd = O.scale_detector_heads(O.init_teacher_state(meta['seed'], meta['model'], **kw))
    det = O.DetectOracle(t_sd, meta['model'], False, meta['min_size'], meta['max_size'])
    images, _ = G.case_inputs(meta)
    for i, d in enumerate(det(images)):
        for k, v in d.items():
            if k == 'masks':
                bits = np.packbits((v > 0.5).numpy().reshape(len(v), -1), axis=1)
                assert np.array_equal(bits, z['teacher/det/%d/masks_bits' % i])
            else:
                assert torch.equal(v, torch.from_numpy(z['teacher/det/%d/%s' % (i, k)])), (i, k)

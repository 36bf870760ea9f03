#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/*.npz FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  The reference's own modules
(src/mimic_runner.py, src/distillation/*, src/models/*) are imported UNMODIFIED over
oracle/shim (torchvision 0.4.2 / myutils restatements), built from the reference's real
YAML configs with a ``--json``-style override (pretrained off, small transform sizes for
the tiny cases), loaded with the seeded state dicts of oracle.hnd_oracle and driven through
DistillationBox -> backward -> Adam exactly as mimic_runner.distill_model does.

Every tensor written here is an OUTPUT OF THE REFERENCE CODE.  While generating, the
functional oracle is run on the same inputs and must agree (bit-exact is expected, the
achieved max abs difference is stored in each fixture as ``oracle_vs_reference_maxabs``).

usage:  python tests/golden/make_golden.py [--only NAME]
"""
import argparse
import json
import os
import random
import sys
from collections import OrderedDict

sys.dont_write_bytecode = True       # importing /root/reference must not leave __pycache__ files in it

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import shim_install  # noqa: E402

shim_install.install()
from oracle import hnd_oracle as O  # noqa: E402

import mimic_runner  # noqa: E402  (reference)
from distillation.tool import DistillationBox  # noqa: E402  (reference)
from models import get_model  # noqa: E402  (reference)
from myutils.common import yaml_util  # noqa: E402  (shim)
from myutils.pytorch import func_util, module_util  # noqa: E402  (shim)
from utils import main_util  # noqa: E402  (reference)

REF_CONFIG = '/root/reference/config'

CASES = OrderedDict((
    ('tiny_ghnd_faster', dict(yaml='ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml', model='faster_rcnn',
                              sizes=[(60, 90), (56, 100)], min_size=64, max_size=128, steps=2, seed=11)),
    ('tiny_hnd_faster', dict(yaml='hnd/faster_rcnn-backbone_resnet50-b3ch.yaml', model='faster_rcnn',
                             sizes=[(64, 96), (64, 96)], min_size=64, max_size=128, steps=2, seed=12)),
    ('tiny_ghnd_mask', dict(yaml='ghnd/mask_rcnn-backbone_resnet50-b3ch.yaml', model='mask_rcnn',
                            sizes=[(48, 80), (64, 64)], min_size=64, max_size=128, steps=1, seed=13)),
    ('tiny_ghnd_keypoint', dict(yaml='ghnd/keypoint_rcnn-backbone_resnet50-b3ch.yaml', model='keypoint_rcnn',
                                sizes=[(64, 96), (60, 84)], min_size=[48, 56, 64], max_size=128, steps=2, seed=14,
                                num_classes=2)),
    ('tiny_ghnd_faster_b6', dict(yaml='ghnd/faster_rcnn-backbone_resnet50-b6ch.yaml', model='faster_rcnn',
                                 sizes=[(64, 96)], min_size=64, max_size=128, steps=1, seed=15, bch=6)),
    # round 4: forward hooks on NON-STANDARD module paths (src/distillation/tool.py:22-35 takes any dotted path): the
    # student's decoder (an alias of its layer1 output), an inner Bottleneck of layer2, layer3 as usual, and the FIRST
    # Bottleneck of layer4 as the highest term (the backward starts in the middle of a layer)
    ('tiny_ghnd_custom_hooks', dict(yaml='ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml', model='faster_rcnn',
                                    sizes=[(60, 90), (56, 100)], min_size=64, max_size=128, steps=2, seed=31,
                                    terms=[['layer1', 'backbone.body.layer1', 'backbone.body.layer1.decoder', 1.0],
                                           ['l2b1', 'backbone.body.layer2.1', 'backbone.body.layer2.1', 0.5],
                                           ['layer3', 'backbone.body.layer3', 'backbone.body.layer3', 1.0],
                                           ['l4b0', 'backbone.body.layer4.0', 'backbone.body.layer4.0', 2.0]])),
    # round 6: student-side terms BELOW / BESIDE the layer outputs.  (a) a pyramid map (backbone.fpn.layer_blocks.1, the
    # stride-8 map) beside layer1 and layer3: its gradient reaches layer2, layer3 AND layer4 through the top-down path, so
    # the backward starts at layer4 although no term sits there; (b) the bottleneck tensor itself
    # (backbone.body.layer1.encoder) -- only a teacher that is itself bottleneck-injected has a tensor to pair with it, so
    # the teacher is built from the yaml's student_model section (another seed), eval mode
    ('tiny_ghnd_fpn_term', dict(yaml='ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml', model='faster_rcnn',
                                sizes=[(60, 90), (56, 100)], min_size=64, max_size=128, steps=2, seed=41,
                                terms=[['layer1', 'backbone.body.layer1', 'backbone.body.layer1', 1.0],
                                       ['p3', 'backbone.fpn.layer_blocks.1', 'backbone.fpn.layer_blocks.1', 0.5],
                                       ['layer3', 'backbone.body.layer3', 'backbone.body.layer3', 1.0]])),
    ('tiny_enc_term', dict(yaml='ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml', model='faster_rcnn',
                           sizes=[(60, 90), (56, 100)], min_size=64, max_size=128, steps=2, seed=42,
                           teacher='student_arch',
                           terms=[['enc', 'backbone.body.layer1.encoder', 'backbone.body.layer1.encoder', 2.0],
                                  ['layer2', 'backbone.body.layer2', 'backbone.body.layer2', 1.0]])),
    ('full_ghnd_faster', dict(yaml='ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml', model='faster_rcnn',
                              sizes=[(800, 1333)], min_size=800, max_size=1333, steps=1, seed=16, full=True)),
    # full-size pins of every BASELINE.json config (round 2).  b4 is the reference's own train batch_size
    # (config/ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml:65); batch 16 does not fit this container's RAM through
    # the reference's autograd graph, so the GPU tests reach batch 16 from these by replication (BN batch statistics,
    # features and Adam updates are invariant under replicating the batch; loss and gradients scale by the factor).
    ('full_ghnd_faster_b4', dict(yaml='ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml', model='faster_rcnn',
                                 sizes=[(800, 1333)] * 4, min_size=800, max_size=1333, steps=1, seed=17, full=True)),
    ('full_hnd_faster_b2', dict(yaml='hnd/faster_rcnn-backbone_resnet50-b3ch.yaml', model='faster_rcnn',
                                sizes=[(800, 1333)] * 2, min_size=800, max_size=1333, steps=1, seed=18, full=True)),
    ('full_ghnd_mask_b2', dict(yaml='ghnd/mask_rcnn-backbone_resnet50-b3ch.yaml', model='mask_rcnn',
                               sizes=[(800, 1333), (800, 1200)], min_size=800, max_size=1333, steps=1, seed=19,
                               full=True)),
    ('full_ghnd_keypoint_b2', dict(yaml='ghnd/keypoint_rcnn-backbone_resnet50-b3ch.yaml', model='keypoint_rcnn',
                                   sizes=[(800, 1333), (1333, 800)], min_size=[640, 672, 704, 736, 768, 800],
                                   max_size=1333, steps=1, seed=20, num_classes=2, full=True)),
    # BASELINE.json configs[3] quotes Mask R-CNN at batch 8 per GPU
    ('full_ghnd_mask_b8', dict(yaml='ghnd/mask_rcnn-backbone_resnet50-b3ch.yaml', model='mask_rcnn',
                               sizes=[(800, 1333)] * 8, min_size=800, max_size=1333, steps=1, seed=22, full=True)),
))


def build_reference_models(case):
    config = yaml_util.load_yaml_file(os.path.join(REF_CONFIG, case['yaml']))
    override = {'teacher_model': {'backbone': {'params': {'pretrained': False}},
                                  'params': {'pretrained': False, 'min_size': case['min_size'],
                                             'max_size': case['max_size']}},
                'student_model': {'backbone': {'params': {'pretrained': False}},
                                  'params': {'pretrained': False, 'min_size': case['min_size'],
                                             'max_size': case['max_size']}}}
    main_util.overwrite_config(config, json.dumps(override))       # same path as the CLI's --json
    if case.get('teacher') == 'student_arch':                      # a bottleneck-injected teacher: the yaml's student section
        import copy
        config['teacher_model'] = copy.deepcopy(config['student_model'])
    device = torch.device('cpu')
    teacher = get_model(config['teacher_model'], device)
    module_util.freeze_module_params(teacher)                      # mimic_runner.py:132
    student = get_model(config['student_model'], device)
    mimic_runner.freeze_modules(student, config['student_model'])  # mimic_runner.py:134
    return config, teacher, student


def make_inputs(case):
    g = torch.Generator().manual_seed(1234 + case['seed'])
    images, targets = [], []
    for h, w in case['sizes']:
        images.append(torch.rand(3, h, w, generator=g))
        t = {'boxes': torch.tensor([[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]]), 'labels': torch.tensor([1])}
        if case['model'] == 'mask_rcnn':
            m = torch.zeros(1, h, w, dtype=torch.uint8)
            m[:, h // 8:h // 2, w // 8:w // 2] = 1
            t['masks'] = m
        if case['model'] == 'keypoint_rcnn':
            kp = torch.rand(1, 17, 3, generator=g)
            kp[..., 0] *= w
            kp[..., 1] *= h
            kp[..., 2] = 1
            t['keypoints'] = kp
        targets.append(t)
    return images, targets


def hooked(model, path):
    return module_util.get_module(model, path).__dict__['distillation_box']['output']


def put(dst, name, t, full_limit=70000):
    """store small tensors whole, big ones as checksum (sum, sumsq, 64 strided samples)."""
    t = t.detach()
    if t.numel() <= full_limit:
        dst[name] = t.cpu().numpy()
    else:
        s, ss, samples = O.checksum(t)
        dst[name + '@sum'] = np.float64(s)
        dst[name + '@sumsq'] = np.float64(ss)
        dst[name + '@samples'] = samples.numpy()
        dst[name + '@shape'] = np.array(t.shape, dtype=np.int64)


def run_case(name, case):
    print('== %s' % name)
    bch = case.get('bch', 3)
    t_sd = O.init_teacher_state(case['seed'], case['model'], num_classes=case.get('num_classes', 91))
    s_sd = O.init_student_state(t_sd, case['seed'] + 1000, bch=bch)
    student_arch_teacher = case.get('teacher') == 'student_arch'
    if student_arch_teacher:
        t_sd = O.init_student_state(t_sd, case['seed'] + 500, bch=bch)
    config, teacher, student = build_reference_models(case)
    teacher.load_state_dict(t_sd, strict=True)      # also proves the oracle's key layout == reference's
    student.load_state_dict(s_sd, strict=True)
    updatable = module_util.get_updatable_param_names(student)
    assert updatable == O.trainable_keys(s_sd), (updatable, O.trainable_keys(s_sd))

    crit = config['train']['criterion']
    if 'terms' in case:             # non-standard ts_modules: same criterion type / params as the yaml's own terms
        proto = next(iter(crit['terms'].values()))['criterion']
        crit['terms'] = OrderedDict((name, {'ts_modules': [tp, sp], 'criterion': proto, 'factor': f})
                                    for name, tp, sp, f in case['terms'])
        terms = OrderedDict((name, (O.rel_key(tp), O.rel_key(sp), f)) for name, tp, sp, f in case['terms'])
    else:
        terms = OrderedDict((k, v['factor']) for k, v in crit['terms'].items())
    box = DistillationBox(teacher, student, crit)
    opt_cfg = config['train']['optimizer']
    optimizer = func_util.get_optimizer(student, opt_cfg['type'], opt_cfg['params'])
    warm = main_util.warmup_lr_scheduler(optimizer, 4, 1.0 / 1000.0)   # mimic_runner.py:43-46 (iters shortened)
    teacher.eval()
    student.train()                                                    # mimic_runner.py:86-90
    teacher.distill_backbone_only = True
    student.distill_backbone_only = True
    student.backbone.body.layer1.use_bottleneck_transformer = False

    min_size = case['min_size'] if isinstance(case['min_size'], list) else [case['min_size']]
    oracle = O.DistillOracle(t_sd, s_sd, terms=terms, lr=opt_cfg['params']['lr'], min_size=tuple(min_size),
                             max_size=case['max_size'], warmup_iters=4, warmup_factor=1e-3,
                             teacher_is_student_arch=student_arch_teacher)
    images, targets = make_inputs(case)
    out = OrderedDict()
    out['meta'] = np.array(json.dumps({k: v for k, v in case.items()}))
    worst = 0.0
    worst_grad = 0.0
    is_kp = case['model'] == 'keypoint_rcnn'
    for step in range(case['steps']):
        fixed_sizes = None
        if is_kp:                       # tool.py:45-48 draws from python's random
            random.seed(100 + step)
            st = random.getstate()
            fixed_sizes = [random.choice(teacher.transform.min_size) for _ in images]
            random.setstate(st)
            out['step%d/fixed_sizes' % step] = np.array(fixed_sizes)
        ims = [im.clone() for im in images]
        tgs = [{k: v.clone() for k, v in t.items()} for t in targets]
        loss = box(ims, tgs)
        optimizer.zero_grad()
        loss.backward()
        grads = OrderedDict((n, p.grad.detach().clone()) for n, p in student.named_parameters() if p.requires_grad)
        lr_used = optimizer.param_groups[0]['lr']
        optimizer.step()
        warm.step()

        o_loss, o_terms, _, _, _, _, o_x = oracle.forward(images, fixed_sizes, update_buffers=False)
        o_l, o_t, o_grads, o_lr = oracle.step(images, fixed_sizes)
        pre = 'step%d/' % step
        out[pre + 'loss'] = np.float64(loss.item())
        out[pre + 'lr'] = np.float64(lr_used)
        worst = max(worst, abs(o_l - loss.item()) / max(1.0, abs(loss.item())))
        assert abs(o_lr - lr_used) < 1e-12
        for k, f in terms.items():
            path = crit['terms'][k]['ts_modules']
            t_out, s_out = hooked(teacher, path[0]), hooked(student, path[1])
            f = f[2] if isinstance(f, tuple) else f
            out[pre + 'term/' + k] = np.float64((torch.nn.functional.mse_loss(t_out, s_out, reduction='sum') * f).item())
            if step == 0:
                put(out, pre + 'teacher/' + k, t_out)
                put(out, pre + 'student/' + k, s_out)
        for n, g in grads.items():
            put(out, pre + 'grad/' + n, g, full_limit=20000)
            if n not in O.ZERO_GRAD_KEYS:       # autograd accumulation order differs -> noise-level only
                worst_grad = max(worst_grad, float((g - o_grads[n]).norm() / g.norm()))
        if step == 0:
            out['batched_shape'] = np.array(o_x.shape)
            # transform output of the reference: re-run it (pure function for fixed sizes / eval mode)
            teacher_x = teacher.transform([im.clone() for im in images], None, fixed_sizes)[0].tensors
            put(out, 'transform', teacher_x)
            worst = max(worst, float((teacher_x - o_x).abs().max()))
            feats = student.backbone.fpn(OrderedDict((i, hooked(student, 'backbone.body.layer%d' % (i + 1)))
                                                     for i in range(4))) if (len(terms) == 4 and 'terms' not in case) else None
            if feats is not None:
                for k, v in feats.items():
                    put(out, 'student_fpn/%s' % k, v, full_limit=30000)
    sd_after = student.state_dict()
    for n in O.trainable_keys(s_sd):
        put(out, 'after/param/' + n, sd_after[n], full_limit=20000)
        if n not in O.ZERO_GRAD_KEYS:
            worst = max(worst, float((sd_after[n] - oracle.s[n].detach()).abs().max()))
    for n, v in sd_after.items():
        if 'layer1' in n and ('running_' in n or 'num_batches' in n):
            out['after/buffer/' + n] = v.numpy()
            worst = max(worst, float((v.double() - oracle.s[n].double()).abs().max()))
    out['oracle_vs_reference_maxabs'] = np.float64(worst)
    out['oracle_vs_reference_grad_rel_l2'] = np.float64(worst_grad)
    print('   loss(step0)=%.6f  oracle-vs-reference worst abs=%.3e  grad relL2=%.3e'
          % (float(out['step0/loss']), worst, worst_grad))
    assert worst < 1e-5 and worst_grad < (2e-4 if case.get('full') else 1e-5), 'oracle restatement diverges from the reference'
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    import resource
    print('   wrote %s (%.1f KB); peak RSS %.1f GB' % (path, os.path.getsize(path) / 1024.0,
                                                    resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0))


def run_eval_codec_case(name='tiny_eval_quantized'):
    """student.eval() with the bottleneck transformer ON (reference base.py:54-57 + structure/transformer.py)."""
    print('== %s' % name)
    case = dict(yaml='ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml', model='faster_rcnn', sizes=[(64, 96), (60, 90)],
                min_size=64, max_size=128, steps=0, seed=21)
    t_sd = O.init_teacher_state(case['seed'])
    s_sd = O.init_student_state(t_sd, case['seed'] + 1000)
    for k in list(s_sd):                        # non-trivial running statistics for the eval-mode BatchNorms
        if 'layer1' in k and k.endswith('running_var'):
            s_sd[k] = s_sd[k] * 1.7 + 0.1
        if 'layer1' in k and k.endswith('running_mean'):
            s_sd[k] = s_sd[k] + 0.05
    config, teacher, student = build_reference_models(case)
    student.load_state_dict(s_sd, strict=True)
    student.eval()
    student.distill_backbone_only = True
    images, _ = make_inputs(case)
    out = OrderedDict()
    out['meta'] = np.array(json.dumps(case))
    worst = 0.0
    for tag, use in (('plain', False), ('quantized', True)):
        student.backbone.body.layer1.use_bottleneck_transformer = use
        captured = []
        hook = student.backbone.body.layer1.encoder.register_forward_hook(lambda m, i, o: captured.append(o.detach()))
        with torch.no_grad():
            feats = student([im.clone() for im in images])
        hook.remove()
        if use:
            # what crosses the link: the reference's own Quantizer (structure/transformer.py:131-140) on the reference's
            # own bottleneck tensor z.  Byte work -> the GPU test demands these exact bytes for this exact z.
            from structure.transformer import Dequantizer, Quantizer      # reference classes
            from oracle.myutils_r import quantize_tensor as oracle_quantize
            z_ref = captured[0]
            qz, _ = Quantizer(8)(z_ref.clone(), None)
            out['quantized/z'] = z_ref.numpy()
            out['quantized/bytes'] = qz.tensor.numpy()
            out['quantized/scale'] = np.float32(float(qz.scale))
            out['quantized/zero_point'] = np.float32(float(qz.zero_point))
            out['quantized/dequantized'] = Dequantizer(8)(qz, None)[0].numpy()
            o_q = oracle_quantize(z_ref.clone(), 8)
            assert torch.equal(o_q.tensor, qz.tensor) and float(o_q.scale) == float(qz.scale)
        x, _ = O.transform_images(images, (64,), 128)
        o_h, o_f = O.backbone_forward(x, O.cast_state(s_sd, torch.float32), student=True, training=False,
                                      codec_bits=8 if use else None)
        for k, v in feats.items():
            put(out, '%s/fpn/%s' % (tag, k), v, full_limit=30000)
            worst = max(worst, float((v - o_f[k]).abs().max()))
    out['oracle_vs_reference_maxabs'] = np.float64(worst)
    print('   oracle-vs-reference worst abs=%.3e' % worst)
    assert worst < 1e-5
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('   wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


DETECT_CASES = {
    'tiny_detect_faster': dict(yaml='ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml', model='faster_rcnn',
                               sizes=[(120, 180), (112, 200)], min_size=128, max_size=256, steps=0, seed=31),
    'tiny_detect_mask': dict(yaml='ghnd/mask_rcnn-backbone_resnet50-b3ch.yaml', model='mask_rcnn',
                             sizes=[(120, 180), (112, 200)], min_size=128, max_size=256, steps=0, seed=33),
    'tiny_detect_keypoint': dict(yaml='ghnd/keypoint_rcnn-backbone_resnet50-b3ch.yaml', model='keypoint_rcnn',
                                 sizes=[(120, 180), (112, 200)], min_size=128, max_size=256, steps=0, seed=35),
}


def detect_states(case):
    kw = {'num_classes': 2} if case['model'] == 'keypoint_rcnn' else {}
    t_sd = O.scale_detector_heads(O.init_teacher_state(case['seed'], case['model'], **kw))
    s_sd = O.init_student_state(t_sd, case['seed'] + 1000)         # inherits the teacher's (already scaled) heads
    return t_sd, s_sd


def run_detect_case(name='tiny_detect_faster'):
    """Validation path: the reference's CustomRCNN.forward in eval mode with
    distill_backbone_only off (src/models/org/rcnn.py:124-127: rpn -> roi_heads -> transform.postprocess), teacher and
    student, over the restated torchvision 0.4.2 detection pieces (oracle/tv042_det.py).  Mask / Keypoint R-CNN: the
    mask / keypoint branches of roi_heads and transform.postprocess run as well (teacher only: the student's heads
    are the teacher's)."""
    print('== %s' % name)
    case = dict(DETECT_CASES[name])
    t_sd, s_sd = detect_states(case)
    config, teacher, student = build_reference_models(case)
    teacher.load_state_dict(t_sd, strict=True)
    student.load_state_dict(s_sd, strict=True)
    images, _ = make_inputs(case)
    out = OrderedDict()
    out['meta'] = np.array(json.dumps(case))
    for tag, model in (('teacher', teacher), ('student', student)):
        model.eval()
        model.distill_backbone_only = False
        with torch.no_grad():
            dets = model([im.clone() for im in images])
        rpn, roi = model.rpn.last, model.roi_heads.last
        out[tag + '/rpn/objectness'] = rpn['objectness'].numpy()
        out[tag + '/rpn/proposals'] = rpn['proposals'].numpy()
        put(out, tag + '/roi/class_logits', roi['class_logits'])
        put(out, tag + '/roi/box_regression', roi['box_regression'])
        if tag == 'student':
            # a self-contained postprocess_detections problem (softmax -> decode -> clip -> threshold -> per-class NMS
            # -> first 100) on the first 300 proposals of image 0, inputs stored whole
            sub = 300
            props0 = rpn['boxes'][0][:sub]
            shape0 = model.roi_heads.last_image_shapes[0]
            b, sc, lb = model.roi_heads.postprocess_detections(roi['class_logits'][:sub], roi['box_regression'][:sub],
                                                               [props0], [shape0])
            out['post/class_logits'] = roi['class_logits'][:sub].numpy()
            out['post/box_regression'] = roi['box_regression'][:sub].numpy()
            out['post/proposals'] = props0.numpy()
            out['post/image_shape'] = np.array(shape0)
            out['post/boxes'], out['post/scores'], out['post/labels'] = b[0].numpy(), sc[0].numpy(), lb[0].numpy()
        for i, (d, sc) in enumerate(zip(dets, rpn['scores'])):
            out['%s/rpn/kept/%d' % (tag, i)] = np.int64(len(sc))
            out['%s/rpn/kept_scores/%d' % (tag, i)] = sc.numpy()
            out['%s/rpn/kept_boxes/%d' % (tag, i)] = model.rpn.last['boxes'][i].numpy()
            for k, v in d.items():
                if k == 'masks':        # [n, 1, H, W] probabilities pasted into the image: bits at the evaluator's 0.5
                    out['%s/det/%d/masks_bits' % (tag, i)] = np.packbits((v > 0.5).numpy().reshape(len(v), -1), axis=1)
                    out['%s/det/%d/masks_sum' % (tag, i)] = v.double().flatten(1).sum(1).numpy()
                    out['%s/det/%d/masks_hw' % (tag, i)] = np.array(v.shape[-2:])
                    continue
                out['%s/det/%d/%s' % (tag, i, k)] = v.numpy()
        if tag == 'teacher' and 'mask_logits' in roi:
            # the branch's own sub-problems: probabilities of the predicted classes before pasting (input of
            # paste_masks_in_image together with det/*/boxes), and the logits' fingerprint
            labels = torch.cat([d['labels'] for d in dets])
            ml = roi['mask_logits']
            out['teacher/roi/mask_probs'] = ml.sigmoid()[torch.arange(len(ml)), labels].numpy()
            put(out, 'teacher/roi/mask_logits', ml)
            out['teacher/roi/det_boxes'] = torch.cat([p for p in model.roi_heads.last_det_boxes]).numpy()
        if tag == 'teacher' and 'keypoint_logits' in roi:
            kl = roi['keypoint_logits']
            put(out, 'teacher/roi/keypoint_logits', kl)
            out['teacher/roi/keypoint_logits_head'] = kl[:12].numpy()        # whole heatmaps of the first 12 RoIs
            out['teacher/roi/det_boxes'] = torch.cat([p for p in model.roi_heads.last_det_boxes]).numpy()
        print('   %s: proposals kept %s, detections %s, top scores %s' % (
            tag, [len(s) for s in rpn['scores']], [len(d['scores']) for d in dets],
            [round(float(d['scores'].max()), 4) for d in dets]))
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('   wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


class _ListDataset(torch.utils.data.Dataset):
    """(image, target) pairs in memory: what main_util.evaluate needs from a dataset (convert_to_coco_api walks it)"""

    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        img, t = self.items[i]
        return img.clone(), {k: v.clone() for k, v in t.items()}


VAL_CASES = {'tiny_val_map': ('tiny_detect_faster', 'bbox'), 'tiny_val_map_mask': ('tiny_detect_mask', 'segm'),
             'tiny_val_map_keypoint': ('tiny_detect_keypoint', 'keypoints')}


def run_validation_case(name='tiny_val_map'):
    """The reference's whole validation path, unmodified: utils/main_util.evaluate (src/utils/main_util.py:75-113)
    drives the eval-mode detector (rcnn.py:124-127) over a batch-1 loader and its CocoEvaluator (iou_types from
    models.get_iou_types: bbox, + segm for Mask R-CNN, + keypoints for Keypoint R-CNN).  Ground truth = the first
    detections of the model itself, perturbed (masks shifted, keypoints jittered), so the metrics are far from 0 with
    random weights and move if detections do."""
    print('== %s' % name)
    from utils import main_util as ref_main                # reference
    from utils import misc_util as ref_misc                # reference
    det_case, kind = VAL_CASES[name]
    case = dict(DETECT_CASES[det_case])
    case['sizes'] = [(120, 180), (112, 200), (128, 160), (100, 190)]
    if kind == 'bbox':
        case['seed'] = 31
    t_sd, _ = detect_states(case)
    config, teacher, _ = build_reference_models(case)
    teacher.load_state_dict(t_sd, strict=True)
    teacher.eval()
    teacher.distill_backbone_only = False
    images, _ = make_inputs(case)
    with torch.no_grad():
        dets = teacher([im.clone() for im in images])
    items, out = [], OrderedDict()
    out['meta'] = np.array(json.dumps(case))
    g = torch.Generator().manual_seed(900 + case['seed'])
    for i, (im, d) in enumerate(zip(images, dets)):
        order = torch.argsort(d['scores'], descending=True)[:12]
        boxes, labels = d['boxes'][order].clone(), d['labels'][order].clone()
        wh = boxes[:, 2:] - boxes[:, :2]
        ok = (wh > 1).all(1)
        boxes, labels = boxes[ok], labels[ok]
        wh = wh[ok]
        tgt = {'image_id': torch.tensor([500 + i]), 'boxes': boxes, 'labels': labels, 'area': wh[:, 0] * wh[:, 1],
               'iscrowd': torch.zeros(len(boxes), dtype=torch.int64)}
        out['gt/%d/boxes' % i], out['gt/%d/labels' % i] = boxes.numpy(), labels.numpy()
        if kind == 'segm':
            m = (d['masks'][order][ok][:, 0] > 0.5)
            m = torch.roll(m, shifts=(1, 2), dims=(1, 2)).to(torch.uint8)
            tgt['masks'] = m
            out['gt/%d/masks_bits' % i] = np.packbits(m.numpy().reshape(len(m), -1), axis=1)
        if kind == 'keypoints':
            kp = d['keypoints'][order][ok].clone()
            kp[..., :2] += (torch.rand(kp[..., :2].shape, generator=g) - 0.5) * 4.0
            kp[..., 2] = torch.randint(0, 3, kp[..., 2].shape, generator=g).float()
            tgt['keypoints'] = kp
            out['gt/%d/keypoints' % i] = kp.numpy()
        items.append((im, tgt))
    loader = torch.utils.data.DataLoader(_ListDataset(items), batch_size=1, shuffle=False,
                                         collate_fn=ref_misc.collate_fn)
    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None          # main_util.py:91 syncs CUDA unconditionally; no GPU here
    try:
        ev = ref_main.evaluate(teacher, loader, device=torch.device('cpu'))
    finally:
        torch.cuda.synchronize = real_sync
    stats = np.array(ev.coco_eval['bbox'].stats, dtype=np.float64)
    out['stats'] = stats
    print('   stats', np.round(stats, 4).tolist())
    assert 0.1 < stats[0] < 1.0
    if kind != 'bbox':
        assert sorted(ev.coco_eval) == sorted(['bbox', kind])
        ks = np.array(ev.coco_eval[kind].stats, dtype=np.float64)
        out['stats_' + kind] = ks
        print('   %s stats' % kind, np.round(ks, 4).tolist())
        assert 0.1 < ks[0] < 1.0
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('   wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


def coco_eval_case_inputs(seed=51):
    """ground truth and predictions of the evaluator fixture (seeded; shared with tests/golden_util.py)"""
    g = torch.Generator().manual_seed(seed)
    dataset, preds = [], {}
    for i in range(6):
        h, w = 360 + 30 * i, 480 + 24 * i
        n = [3, 1, 5, 2, 0, 4][i]
        xy = torch.rand(n, 2, generator=g) * torch.tensor([w * 0.5, h * 0.5])
        wh = torch.rand(n, 2, generator=g) ** 2 * torch.tensor([w * 0.45, h * 0.45]) + 6
        boxes = torch.cat([xy, xy + wh], 1)
        labels = torch.randint(1, 4, (n,), generator=g)
        crowd = (torch.rand(n, generator=g) < 0.15).to(torch.int64)
        tgt = {'image_id': torch.tensor([200 + i]), 'boxes': boxes, 'labels': labels,
               'area': (wh[:, 0] * wh[:, 1]), 'iscrowd': crowd}
        dataset.append((torch.zeros(3, h, w), tgt))
        # predictions: jittered copies of the ground truth (some dropped), a duplicate, and false positives
        keep = torch.rand(n, generator=g) < 0.8
        jit = (torch.rand(n, 4, generator=g) - 0.5) * torch.cat([wh, wh], 1) * 0.22
        pb = (boxes + jit)[keep]
        pl = labels[keep].clone()
        if len(pl) > 1:
            pl[0] = 1 + (pl[0] % 3)                       # one detection with the wrong class
        fp = torch.rand(3, 2, generator=g) * torch.tensor([w * 0.7, h * 0.7])
        fpb = torch.cat([fp, fp + torch.rand(3, 2, generator=g) * 30 + 3], 1)
        pb = torch.cat([pb, pb[:1], fpb], 0)
        pl = torch.cat([pl, pl[:1], torch.randint(1, 4, (3,), generator=g)], 0)
        ps = torch.rand(len(pb), generator=g)
        preds[200 + i] = {'boxes': pb, 'labels': pl, 'scores': ps}
    return dataset, preds


def run_coco_eval_case(name='tiny_coco_eval', kind='bbox'):
    """The reference's own evaluator code -- utils/coco_util.convert_to_coco_api, utils/coco_eval_util.CocoEvaluator
    with its copies of loadRes / evaluate / createIndex (src/utils/coco_eval_util.py:15-150,198-345) -- over the
    restated pycocotools (oracle/pycoco_r.py): ground truth from a dataset of targets, predictions fed in two
    update() calls, synchronize / accumulate / summarize -> the COCO statistics.  kind 'segm' / 'keypoints': the
    iou_types the reference uses for Mask / Keypoint R-CNN (coco_eval_util.py:225-233: bbox + segm / bbox + keypoints)."""
    print('== %s' % name)
    from utils import coco_eval_util as ref_eval          # reference
    from utils import coco_util as ref_coco               # reference
    dataset, preds = coco_eval_case_inputs()
    iou_types = ['bbox']
    if kind != 'bbox':
        from tests import golden_util
        dataset, preds = golden_util.coco_eval_case_extras(dataset, preds, kind)
        iou_types.append(kind)
    coco = ref_coco.convert_to_coco_api([(img, {k: v.clone() for k, v in t.items()}) for img, t in dataset])
    ev = ref_eval.CocoEvaluator(coco, iou_types)
    ids = sorted(preds)
    ev.update({i: preds[i] for i in ids[:4]})
    ev.update({i: preds[i] for i in ids[4:]})
    ev.synchronize_between_processes()
    ev.accumulate()
    ev.summarize()
    stats = np.array(ev.coco_eval['bbox'].stats, dtype=np.float64)
    out = OrderedDict(stats=stats, seed=np.int64(51),
                      precision_checksum=np.float64(ev.coco_eval['bbox'].eval['precision'].clip(min=0).sum()))
    print('   stats', np.round(stats, 4).tolist())
    assert 0.05 < stats[0] < 0.95 and stats[1] > stats[0] >= 0      # a non-trivial case
    if kind != 'bbox':
        ks = np.array(ev.coco_eval[kind].stats, dtype=np.float64)
        out['stats_' + kind] = ks
        out['precision_checksum_' + kind] = np.float64(ev.coco_eval[kind].eval['precision'].clip(min=0).sum())
        print('   %s stats' % kind, np.round(ks, 4).tolist())
        assert 0.05 < ks[0] < 0.95 and ks[1] >= ks[0]
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('   wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


def run_input_pipeline_case(name='tiny_input_pipeline'):
    """Decoded uint8 image -> reference ToTensor -> RandomHorizontalFlip (forced on / off) -> CustomRCNNTransform."""
    print('== %s' % name)
    from PIL import Image
    from structure.transformer import RandomHorizontalFlip, ToTensor       # the reference's classes
    from models.org.rcnn import CustomRCNNTransform
    g = torch.Generator().manual_seed(33)
    out = OrderedDict()
    images, worst = [], 0.0
    for i, (h, w, flip) in enumerate([(40, 56, True), (48, 44, False), (37, 61, True)]):
        u8 = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
        kp = torch.rand(2, 17, 3, generator=g) * torch.tensor([w, h, 1.0])
        kp[..., 2] = (kp[..., 2] > 0.4).float()
        target = {'boxes': torch.tensor([[3.0, 4.0, 0.5 * w, 0.75 * h], [1.0, 2.0, w - 2.0, h - 1.0]]),
                  'masks': (torch.rand(2, h, w, generator=g) > 0.5).to(torch.uint8), 'keypoints': kp}
        before = {k: v.clone() for k, v in target.items()}
        img, tgt = ToTensor()(Image.fromarray(u8.numpy()), target)
        img, tgt = RandomHorizontalFlip(1.0 if flip else 0.0)(img, tgt)
        o_img = O.to_tensor_u8(u8)
        o_tgt = before
        if flip:
            o_img, o_tgt = O.horizontal_flip(o_img, before)
        worst = max(worst, float((img - o_img).abs().max()),
                    *[float((tgt[k].float() - o_tgt[k].float()).abs().max()) for k in tgt])
        out['u8/%d' % i] = u8.numpy()
        out['flip/%d' % i] = np.array(flip)
        for k in before:
            out['target_in/%d/%s' % (i, k)] = before[k].numpy()
            out['target_out/%d/%s' % (i, k)] = tgt[k].numpy()
        images.append(img)
    tr = CustomRCNNTransform(64, 128, O.IMAGE_MEAN, O.IMAGE_STD)
    tr.eval()
    il, _ = tr([im.clone() for im in images], None)
    o_batch, o_sizes = O.transform_images(images, (64,), 128)
    worst = max(worst, float((il.tensors - o_batch).abs().max()))
    assert [tuple(s) for s in il.image_sizes] == o_sizes
    out['batch'] = il.tensors.numpy()
    out['image_sizes'] = np.array(o_sizes)
    out['oracle_vs_reference_maxabs'] = np.float64(worst)
    print('   oracle-vs-reference worst abs=%.3e' % worst)
    assert worst < 1e-6
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('   wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


def run_ext_filter_case(name='tiny_ext_filter'):
    """Neural filter: the reference's ext model (config/ext/*.yaml) trained the way
    ext_runner.train_model does -- cross entropy on Ext4ResNet logits, SGD(momentum, wd), epoch-0 warm-up -- then
    evaluated (softmax probabilities)."""
    print('== %s' % name)
    import ext_runner                                              # reference
    case = dict(yaml='ext/keypoint_rcnn-backbone_ext_resnet50-b3ch.yaml', model='keypoint_rcnn', seed=41,
                sizes=[(60, 90), (64, 72), (52, 100)], min_size=64, max_size=128, steps=3, num_classes=2,
                loader_len=5)
    config = yaml_util.load_yaml_file(os.path.join(REF_CONFIG, case['yaml']))
    override = {'model': {'backbone': {'params': {'pretrained': False},
                                       'ext_config': {'ckpt': '/nonexistent/ext.pt'}},
                          'params': {'pretrained': False, 'min_size': case['min_size'], 'max_size': case['max_size']},
                          'ckpt': '/nonexistent/student.pt'}}
    main_util.overwrite_config(config, json.dumps(override))
    device = torch.device('cpu')
    model = get_model(config['model'], device, strict=False)       # ext_runner.py:193-194
    t_sd = O.init_teacher_state(case['seed'], 'keypoint_rcnn', num_classes=2)
    s_sd = O.init_student_state(t_sd, case['seed'] + 1000)
    e_sd = O.init_ext_state(case['seed'] + 2000)
    full = OrderedDict(s_sd)
    full.update(e_sd)
    model.load_state_dict(full, strict=True)
    module_util.freeze_module_params(model)                        # ext_runner.py:195-199
    ext_classifier = model.get_ext_classifier()
    module_util.unfreeze_module_params(ext_classifier)
    names = module_util.get_updatable_param_names(model)
    model.train_ext()
    opt_cfg = config['train']['optimizer']
    optimizer = func_util.get_optimizer(ext_classifier, opt_cfg['type'], opt_cfg['params'])
    warm = main_util.warmup_lr_scheduler(optimizer, min(1000, case['loader_len'] - 1), 1.0 / 1000.0)
    orc = O.FilterOracle(s_sd, e_sd, lr=opt_cfg['params']['lr'], momentum=opt_cfg['params']['momentum'],
                         weight_decay=opt_cfg['params']['weight_decay'], min_size=(case['min_size'],),
                         max_size=case['max_size'], warmup_iters=case['loader_len'] - 1)

    g = torch.Generator().manual_seed(1234 + case['seed'])
    images, targets = [], []
    for i, (h, w) in enumerate(case['sizes']):
        images.append(torch.rand(3, h, w, generator=g))
        kp = torch.rand(1, 17, 3, generator=g) * torch.tensor([w, h, 1.0])
        kp[..., 2] = 1.0 if i != 2 else 0.0                         # image 2: no visible keypoints -> negative
        if i == 2:
            kp[0, :5, 2] = 1.0
        box = [[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]] if i != 1 else [[3.0, 4.0, 0.5, 20.0]]   # image 1: empty box
        targets.append({'boxes': torch.tensor(box), 'labels': torch.tensor([1]), 'keypoints': kp})
    out = OrderedDict()
    out['meta'] = np.array(json.dumps(case))
    out['param_names'] = np.array(json.dumps(names))
    worst, worst_grad = 0.0, 0.0
    pref = 'backbone.body.layer1.encoder.ext_classifier.'
    model.train()                                                  # train_model, ext_runner.py:40
    for step in range(case['steps']):
        ims = [im.clone() for im in images]
        tgs = [{k: v.clone() for k, v in t.items()} for t in targets]
        ext_logits = model(ims, tgs)
        ext_targets = ext_runner.convert_target2ext_targets(tgs, device)
        loss = torch.nn.functional.cross_entropy(ext_logits, ext_targets)
        optimizer.zero_grad()
        loss.backward()
        lr = optimizer.param_groups[0]['lr']
        grads = OrderedDict((n, p.grad.detach().clone()) for n, p in model.named_parameters() if p.requires_grad)
        optimizer.step()
        warm.step()
        o_loss, o_logits, o_grads, o_lr = orc.step(images, targets)
        out['step%d/logits' % step] = ext_logits.detach().numpy()
        out['step%d/labels' % step] = ext_targets.numpy()
        out['step%d/loss' % step] = np.float64(float(loss.detach()))
        out['step%d/lr' % step] = np.float64(lr)
        worst = max(worst, float((ext_logits.detach() - o_logits).abs().max()), abs(float(loss) - o_loss))
        assert abs(lr - o_lr) < 1e-15
        for n, gr in grads.items():
            put(out, 'step%d/grad/%s' % (step, n[len(pref):]), gr, full_limit=4000)
            if n in O.EXT_ZERO_GRAD_KEYS:
                continue
            den = float(gr.norm()) + 1e-30
            worst_grad = max(worst_grad, float((gr - o_grads[n]).norm()) / den)
        for n, p in model.named_parameters():
            if p.requires_grad:
                put(out, 'step%d/param_after/%s' % (step, n[len(pref):]), p.detach().clone(), full_limit=4000)
                worst = max(worst, float((p.detach() - orc.s[n].detach()).abs().max()))
    sd = model.state_dict()
    for k, v in sd.items():                                        # every BatchNorm buffer the ext training touched
        if 'layer1' in k and ('running_' in k or 'num_batches' in k):
            out['buffers/' + k] = v.numpy().copy()
            worst = max(worst, float((v.double() - orc.s[k].double()).abs().max()))
    model.eval()                                                   # evaluate, ext_runner.py:80
    with torch.no_grad():
        probs = model([im.clone() for im in images], [dict(t) for t in targets])
        probs1 = model([images[0].clone()], [dict(targets[0])])
    out['eval/probs'] = probs.numpy()
    out['eval/probs_single'] = probs1.numpy()
    with torch.no_grad():
        worst = max(worst, float((probs - orc.forward(images, training=False)).abs().max()),
                    float((probs1 - orc.forward(images[:1], training=False)).abs().max()))
    out['oracle_vs_reference_maxabs'] = np.float64(worst)
    out['oracle_vs_reference_grad_rel'] = np.float64(worst_grad)
    print('   labels=%s loss0=%.6f oracle-vs-reference worst abs=%.3e grad rel=%.3e'
          % (out['step0/labels'].tolist(), float(out['step0/loss']), worst, worst_grad))
    assert worst < 1e-5 and worst_grad < 1e-4
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('   wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only')
    args = ap.parse_args()
    torch.set_num_threads(8)
    for name, case in CASES.items():
        if args.only and args.only != name:
            continue
        run_case(name, case)
    if not args.only or args.only == 'tiny_eval_quantized':
        run_eval_codec_case()
    for name in DETECT_CASES:
        if not args.only or args.only == name:
            run_detect_case(name)
    if not args.only or args.only == 'tiny_coco_eval':
        run_coco_eval_case()
    for kind in ('segm', 'keypoints'):
        if not args.only or args.only == 'tiny_coco_eval_' + kind:
            run_coco_eval_case('tiny_coco_eval_' + kind, kind)
    for name in VAL_CASES:
        if not args.only or args.only == name:
            run_validation_case(name)
    if not args.only or args.only == 'tiny_input_pipeline':
        run_input_pipeline_case()
    if not args.only or args.only == 'tiny_ext_filter':
        run_ext_filter_case()


if __name__ == '__main__':
    main()

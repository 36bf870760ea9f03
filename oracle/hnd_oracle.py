"""Functional CPU oracle of the reference's HND/GHND distillation step.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

The reference expresses the step as nn.Modules (torch 1.3.1 + torchvision 0.4.2);
this file restates the same arithmetic as plain functions over a *state dict*
(keys identical to the reference checkpoints) so that the HIP path,
which is also state-dict driven, can be compared tensor by tensor.  Works in fp32
(parity target) and fp64 (gradient oracle).  Pinned against the reference run over
``oracle/shim`` by ``tests/golden/make_golden.py`` (fixtures in ``tests/golden``).

Reference lines restated (all relative to /root/reference):
  transform            src/models/org/rcnn.py:25-82  (+ torchvision 0.4.2 normalize/batch_images)
  stem / layers        src/models/custom/resnet.py:26-30,95-105 ; torchvision resnet Bottleneck
  student layer1       src/models/mimic/resnet_layer.py:40-70 ; src/models/mimic/base.py:21-22,50-58
  FrozenBatchNorm2d    torchvision 0.4.2 ops/misc.py (no eps), built at src/models/org/rcnn.py:391,394
  FPN                  torchvision 0.4.2 ops/feature_pyramid_network.py via src/models/org/rcnn.py:399-414
  early exit           src/models/org/rcnn.py:102-110
  hooks / loss         src/distillation/tool.py:40-61 ; src/distillation/loss.py:25-34
  optimisation step    src/mimic_runner.py:38-59 ; src/utils/main_util.py:65-72
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

RESNET50_BLOCKS = (3, 4, 6, 3)
IMAGE_MEAN = (0.485, 0.456, 0.406)      # src/models/org/rcnn.py:222-226
IMAGE_STD = (0.229, 0.224, 0.225)
BN_EPS, BN_MOMENTUM = 1e-5, 0.1         # torch nn.BatchNorm2d defaults
B = 'backbone.body.'

# (kind, index) layout of the two nn.Sequential of Bottleneck4LargeResNet, resnet_layer.py:42-65
ENCODER_SPEC = (('conv', 0, 1), ('bn', 1), ('conv', 2, 1), ('bn', 3), ('relu', 4),
                ('conv', 5, 1), ('bn', 6), ('conv', 7, 1))
DECODER_SPEC = (('bn', 0), ('relu', 1), ('conv', 2, 0), ('bn', 3), ('conv', 4, 0), ('bn', 5), ('relu', 6),
                ('conv', 7, 0), ('bn', 8), ('conv', 9, 0), ('bn', 10), ('relu', 11))


def head_channels(bch):
    """(Cin, Cout) of the eight 2x2 convs, resnet_layer.py:43-50,55-62."""
    enc = {0: (64, 64), 2: (64, 256), 5: (256, 64), 7: (64, bch)}
    dec = {2: (bch, 64), 4: (64, 128), 7: (128, 256), 9: (256, 256)}
    enc_bn = {1: 64, 3: 256, 6: 64}
    dec_bn = {0: bch, 3: 64, 5: 128, 8: 256, 10: 256}
    return enc, dec, enc_bn, dec_bn


# ----------------------------------------------------------------------------- init
def _kaiming_fan_out(gen, cout, cin, kh, kw, dtype):
    # custom/resnet.py:55-57: kaiming_normal_(mode='fan_out', nonlinearity='relu')
    std = math.sqrt(2.0 / (cout * kh * kw))
    return torch.randn(cout, cin, kh, kw, generator=gen, dtype=torch.float32).mul_(std).to(dtype)


def _uniform(gen, shape, bound, dtype):
    return ((torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * bound).to(dtype)


def _frozen_bn(gen, sd, prefix, c, dtype):
    # recipe of SURVEY.md C.6: keep activations O(1) through 50 random layers
    sd[prefix + 'weight'] = (torch.rand(c, generator=gen) * 0.5 + 0.25).to(dtype)
    sd[prefix + 'bias'] = (torch.randn(c, generator=gen) * 0.1).to(dtype)
    sd[prefix + 'running_mean'] = (torch.randn(c, generator=gen) * 0.1).to(dtype)
    sd[prefix + 'running_var'] = (torch.rand(c, generator=gen) * 1.5 + 0.5).to(dtype)


def _resnet_layer_init(gen, sd, prefix, inplanes, planes, blocks, stride, dtype):
    for i in range(blocks):
        p = '%s%d.' % (prefix, i)
        cin = inplanes if i == 0 else planes * 4
        sd[p + 'conv1.weight'] = _kaiming_fan_out(gen, planes, cin, 1, 1, dtype)
        _frozen_bn(gen, sd, p + 'bn1.', planes, dtype)
        sd[p + 'conv2.weight'] = _kaiming_fan_out(gen, planes, planes, 3, 3, dtype)
        _frozen_bn(gen, sd, p + 'bn2.', planes, dtype)
        sd[p + 'conv3.weight'] = _kaiming_fan_out(gen, planes * 4, planes, 1, 1, dtype)
        _frozen_bn(gen, sd, p + 'bn3.', planes * 4, dtype)
        if i == 0 and (stride != 1 or inplanes != planes * 4):
            sd[p + 'downsample.0.weight'] = _kaiming_fan_out(gen, planes * 4, cin, 1, 1, dtype)
            _frozen_bn(gen, sd, p + 'downsample.1.', planes * 4, dtype)


def _linear(gen, sd, prefix, cin, cout, dtype):
    b = 1.0 / math.sqrt(cin)
    sd[prefix + 'weight'] = _uniform(gen, (cout, cin), b, dtype)
    sd[prefix + 'bias'] = _uniform(gen, (cout,), b, dtype)


def _conv_b(gen, sd, prefix, cin, cout, k, dtype, transposed=False):
    b = 1.0 / math.sqrt(cin * k * k)
    shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    sd[prefix + 'weight'] = _uniform(gen, shape, b, dtype)
    sd[prefix + 'bias'] = _uniform(gen, (cout,), b, dtype)


def init_teacher_state(seed, model_name='faster_rcnn', num_classes=91, num_keypoints=17, dtype=torch.float32):
    """Seeded stand-in for the COCO-pretrained detector (weights are not downloadable here)."""
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    sd[B + 'conv1.weight'] = _kaiming_fan_out(gen, 64, 3, 7, 7, dtype)
    _frozen_bn(gen, sd, B + 'bn1.', 64, dtype)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), RESNET50_BLOCKS), 1):
        _resnet_layer_init(gen, sd, '%slayer%d.' % (B, li), inplanes, planes, blocks, 1 if li == 1 else 2, dtype)
        inplanes = planes * 4
    for i, c in enumerate((256, 512, 1024, 2048)):
        _conv_b(gen, sd, 'backbone.fpn.inner_blocks.%d.' % i, c, 256, 1, dtype)
    for i in range(4):
        _conv_b(gen, sd, 'backbone.fpn.layer_blocks.%d.' % i, 256, 256, 3, dtype)
    _conv_b(gen, sd, 'rpn.head.conv.', 256, 256, 3, dtype)
    _conv_b(gen, sd, 'rpn.head.cls_logits.', 256, 3, 1, dtype)
    _conv_b(gen, sd, 'rpn.head.bbox_pred.', 256, 12, 1, dtype)
    _linear(gen, sd, 'roi_heads.box_head.fc6.', 256 * 7 * 7, 1024, dtype)
    _linear(gen, sd, 'roi_heads.box_head.fc7.', 1024, 1024, dtype)
    _linear(gen, sd, 'roi_heads.box_predictor.cls_score.', 1024, num_classes, dtype)
    _linear(gen, sd, 'roi_heads.box_predictor.bbox_pred.', 1024, num_classes * 4, dtype)
    if model_name == 'mask_rcnn':
        for i in range(1, 5):
            _conv_b(gen, sd, 'roi_heads.mask_head.mask_fcn%d.' % i, 256, 256, 3, dtype)
        _conv_b(gen, sd, 'roi_heads.mask_predictor.conv5_mask.', 256, 256, 2, dtype, transposed=True)
        _conv_b(gen, sd, 'roi_heads.mask_predictor.mask_fcn_logits.', 256, num_classes, 1, dtype)
    elif model_name == 'keypoint_rcnn':
        cin = 256
        for i in range(8):
            _conv_b(gen, sd, 'roi_heads.keypoint_head.%d.' % (2 * i), cin, 512, 3, dtype)
            cin = 512
        _conv_b(gen, sd, 'roi_heads.keypoint_predictor.kps_score_lowres.', 512, num_keypoints, 4, dtype,
                transposed=True)
    return sd


def init_student_state(teacher_sd, seed, bch=3, dtype=torch.float32):
    """Student = teacher weights everywhere except layer1 (rcnn.py:446-450, strict=False load) plus a
    freshly initialised bottleneck layer1 (custom/resnet.py:55-60: kaiming fan_out convs, gamma=1, beta=0)."""
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, v in teacher_sd.items():
        if not k.startswith(B + 'layer1.'):
            sd[k] = v.clone()
    enc, dec, enc_bn, dec_bn = head_channels(bch)
    new = OrderedDict()
    for prefix, convs, bns in ((B + 'layer1.encoder.encoder.', enc, enc_bn), (B + 'layer1.decoder.', dec, dec_bn)):
        for idx in sorted(list(convs) + list(bns)):
            if idx in convs:
                cin, cout = convs[idx]
                new['%s%d.weight' % (prefix, idx)] = _kaiming_fan_out(gen, cout, cin, 2, 2, dtype)
            else:
                c = bns[idx]
                new['%s%d.weight' % (prefix, idx)] = torch.ones(c, dtype=dtype)
                new['%s%d.bias' % (prefix, idx)] = torch.zeros(c, dtype=dtype)
                new['%s%d.running_mean' % (prefix, idx)] = torch.zeros(c, dtype=dtype)
                new['%s%d.running_var' % (prefix, idx)] = torch.ones(c, dtype=dtype)
                new['%s%d.num_batches_tracked' % (prefix, idx)] = torch.tensor(0, dtype=torch.long)
    # keep the reference's state_dict() order: conv1, bn1, layer1, layer2...
    out = OrderedDict()
    for k, v in sd.items():
        out[k] = v
        if k == B + 'bn1.running_var':
            out.update(new)
    return out


# seeded random heads give near-uniform scores (everything ties); these factors spread RPN objectness / box deltas /
# class logits enough that the eval-mode detector of the validation path (row f4) keeps ~800 proposals and fills its
# 100 detections per image on the tiny fixtures -- NMS, top-k and the score threshold all bite
DETECT_HEAD_SCALES = {'rpn.head.conv.weight': 6.0, 'rpn.head.cls_logits.weight': 30.0, 'rpn.head.bbox_pred.weight': 1.5,
                      'roi_heads.box_head.fc6.weight': 2.0, 'roi_heads.box_head.fc7.weight': 2.0,
                      'roi_heads.box_predictor.cls_score.weight': 20.0, 'roi_heads.box_predictor.bbox_pred.weight': 1.5,
                      # Mask / Keypoint R-CNN only: mask logits spread to ~+-3 (masks are not all-or-nothing at the 0.5
                      # threshold), keypoint heatmaps peaked well above fp32 noise (stable argmax)
                      'roi_heads.mask_head.mask_fcn1.weight': 2.5, 'roi_heads.mask_head.mask_fcn2.weight': 2.5,
                      'roi_heads.mask_head.mask_fcn3.weight': 2.5, 'roi_heads.mask_head.mask_fcn4.weight': 2.5,
                      'roi_heads.mask_predictor.conv5_mask.weight': 3.0,
                      'roi_heads.mask_predictor.mask_fcn_logits.weight': 20.0,
                      'roi_heads.keypoint_predictor.kps_score_lowres.weight': 12.0}
DETECT_HEAD_SCALES.update({'roi_heads.keypoint_head.%d.weight' % (2 * i): 1.8 for i in range(8)})


def scale_detector_heads(sd):
    out = OrderedDict(sd)
    for k, f in DETECT_HEAD_SCALES.items():
        if k in sd:
            out[k] = sd[k] * f
    return out


def trainable_keys(student_sd):
    """conv1.weight + every *parameter* of layer1: the reference's 25 updatable tensors
    (mimic_runner.py:32-35 with yaml frozen_modules)."""
    keys = [B + 'conv1.weight']
    for k in student_sd:
        if k.startswith(B + 'layer1.') and (k.endswith('.weight') or k.endswith('.bias')):
            keys.append(k)
    return keys


def cast_state(sd, dtype):
    return OrderedDict((k, v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items())


# ----------------------------------------------------------------------------- transform
def transform_images(images, min_size=(800,), max_size=1333, training=False, fixed_sizes=None,
                     mean=IMAGE_MEAN, std=IMAGE_STD, rng=None):
    """rcnn.py:65-82: normalise, bilinear resize (rcnn.py:29-45), zero-pad batch to a multiple of 32."""
    if not isinstance(min_size, (list, tuple)):
        min_size = (min_size,)
    out, sizes = [], []
    for i, img in enumerate(images):
        m = torch.as_tensor(mean, dtype=img.dtype)[:, None, None]
        s = torch.as_tensor(std, dtype=img.dtype)[:, None, None]
        img = (img - m) / s
        h, w = img.shape[-2:]
        lo, hi = float(min(h, w)), float(max(h, w))
        if fixed_sizes is not None:
            size = fixed_sizes[i]
        elif training:
            size = (rng or __import__('random')).choice(min_size)
        else:
            size = min_size[-1]
        scale = size / lo
        if hi * scale > max_size:
            scale = max_size / hi
        img = F.interpolate(img[None], scale_factor=scale, mode='bilinear', align_corners=False)[0]
        out.append(img)
        sizes.append(tuple(img.shape[-2:]))
    hp = int(math.ceil(max(o.shape[1] for o in out) / 32.0) * 32)
    wp = int(math.ceil(max(o.shape[2] for o in out) / 32.0) * 32)
    batch = out[0].new_zeros((len(out), 3, hp, wp))
    for b, o in zip(batch, out):
        b[:, :o.shape[1], :o.shape[2]].copy_(o)
    return batch, sizes


# ----------------------------------------------------------------------------- dataset-side transforms
def to_tensor_u8(image_hwc):
    """structure/transformer.py:52-55 -> torchvision functional.to_tensor: uint8 [H,W,3] -> float [3,H,W] / 255."""
    return image_hwc.permute(2, 0, 1).contiguous().float().div(255)


_KP_FLIP = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]


def horizontal_flip(image, target):
    """structure/transformer.py:12-20,36-49 (the branch RandomHorizontalFlip takes when it flips)."""
    width = image.shape[-1]
    image = image.flip(-1)
    target = dict(target)
    b = target['boxes'].clone()
    b[:, [0, 2]] = width - b[:, [2, 0]]
    target['boxes'] = b
    if 'masks' in target:
        target['masks'] = target['masks'].flip(-1)
    if 'keypoints' in target:
        k = target['keypoints'][:, _KP_FLIP].clone()
        k[..., 0] = width - k[..., 0]
        k[k[..., 2] == 0] = 0
        target['keypoints'] = k
    return image, target


# ----------------------------------------------------------------------------- network pieces
def frozen_bn(x, sd, p):
    scale = sd[p + 'weight'] * sd[p + 'running_var'].rsqrt()          # no eps (0.4.2)
    shift = sd[p + 'bias'] - sd[p + 'running_mean'] * scale
    return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


def stem(x, sd):
    x = F.conv2d(x, sd[B + 'conv1.weight'], None, stride=2, padding=3)
    x = F.relu(frozen_bn(x, sd, B + 'bn1.'))
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def bottleneck_block(x, sd, p, stride):
    y = F.relu(frozen_bn(F.conv2d(x, sd[p + 'conv1.weight']), sd, p + 'bn1.'))
    y = F.relu(frozen_bn(F.conv2d(y, sd[p + 'conv2.weight'], None, stride=stride, padding=1), sd, p + 'bn2.'))
    y = frozen_bn(F.conv2d(y, sd[p + 'conv3.weight']), sd, p + 'bn3.')
    if (p + 'downsample.0.weight') in sd:
        x = frozen_bn(F.conv2d(x, sd[p + 'downsample.0.weight'], None, stride=stride), sd, p + 'downsample.1.')
    return F.relu(y + x)


def resnet_layer(x, sd, li, hooked=None):
    """hooked: dict that also receives every Bottleneck's output under 'layer<li>.<i>' (forward hooks on inner blocks:
    src/distillation/tool.py:22-35 accepts any dotted module path)"""
    for i in range(RESNET50_BLOCKS[li - 1]):
        x = bottleneck_block(x, sd, '%slayer%d.%d.' % (B, li, i), 2 if (i == 0 and li > 1) else 1)
        if hooked is not None:
            hooked['layer%d.%d' % (li, i)] = x
    return x


def _train_bn(x, sd, p, training, update_buffers):
    rm, rv = sd[p + 'running_mean'], sd[p + 'running_var']
    if training and not update_buffers:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, sd[p + 'weight'], sd[p + 'bias'], training, BN_MOMENTUM, BN_EPS)
    if training and update_buffers:
        sd[p + 'num_batches_tracked'] += 1
    return y


def quantize_dequantize(z, num_bits=8):
    """Quantizer -> Dequantizer of src/structure/transformer.py:131-153 (myutils tensor_util semantics)."""
    if num_bits == 16:
        return z.half().float()
    from oracle.myutils_r import quantize_tensor, dequantize_tensor
    return dequantize_tensor(quantize_tensor(z, num_bits=num_bits))


def student_layer1(x, sd, training=True, update_buffers=True, intermediates=None, codec_bits=None, hooked=None):
    """Bottleneck4LargeResNet.forward == decoder(encoder(x)) (base.py:50-58 with
    use_bottleneck_transformer False, as mimic_runner.py:90 forces during distillation).
    hooked: receives the bottleneck tensor as 'layer1.encoder' (a forward hook on backbone.body.layer1.encoder)."""
    for prefix, spec in ((B + 'layer1.encoder.encoder.', ENCODER_SPEC), (B + 'layer1.decoder.', DECODER_SPEC)):
        if spec is DECODER_SPEC and hooked is not None:
            hooked['layer1.encoder'] = x
        if spec is DECODER_SPEC and codec_bits is not None and not training:
            x = quantize_dequantize(x, codec_bits)        # base.py:54-57: eval only, between encoder and decoder
        for op in spec:
            name = '%s%d' % (prefix, op[1])
            if op[0] == 'conv':
                x = F.conv2d(x, sd[name + '.weight'], None, stride=1, padding=op[2])
            elif op[0] == 'bn':
                x = _train_bn(x, sd, name + '.', training, update_buffers)
            else:
                x = F.relu(x)
            if intermediates is not None:
                intermediates[name] = x
    return x


def fpn(feats, sd):
    p = 'backbone.fpn.'
    last = F.conv2d(feats[3], sd[p + 'inner_blocks.3.weight'], sd[p + 'inner_blocks.3.bias'])
    outs = [F.conv2d(last, sd[p + 'layer_blocks.3.weight'], sd[p + 'layer_blocks.3.bias'], padding=1)]
    for i in (2, 1, 0):
        lat = F.conv2d(feats[i], sd[p + 'inner_blocks.%d.weight' % i], sd[p + 'inner_blocks.%d.bias' % i])
        last = lat + F.interpolate(last, size=lat.shape[-2:], mode='nearest')
        outs.insert(0, F.conv2d(last, sd[p + 'layer_blocks.%d.weight' % i], sd[p + 'layer_blocks.%d.bias' % i],
                                padding=1))
    outs.append(F.max_pool2d(outs[-1], 1, 2, 0))
    return OrderedDict(zip((0, 1, 2, 3, 'pool'), outs))


def backbone_forward(x, sd, student, training=True, update_buffers=True, with_fpn=True, intermediates=None,
                     codec_bits=None):
    """body (IntermediateLayerGetter) + fpn; returns (hooked layer outputs, fpn features)."""
    hooked = OrderedDict()
    x = stem(x, sd)
    if intermediates is not None:
        intermediates['stem'] = x
    if student:
        x = student_layer1(x, sd, training, update_buffers, intermediates, codec_bits, hooked)
        hooked['layer1.decoder'] = x            # the decoder's output IS the layer's (base.py:50-58)
    else:
        x = resnet_layer(x, sd, 1, hooked)
    hooked['layer1'] = x
    for li in (2, 3, 4):
        x = resnet_layer(x, sd, li, hooked)
        hooked['layer%d' % li] = x
    feats = fpn([hooked['layer%d' % i] for i in (1, 2, 3, 4)], sd) if with_fpn else None
    if feats is not None:                       # forward hooks on backbone.fpn.layer_blocks.K (keys relative to `backbone.`)
        for i in range(4):
            hooked['fpn.layer_blocks.%d' % i] = feats[i]
    return hooked, feats


def rel_key(path):
    """dotted module path of a forward hook -> the key its tensor is hooked under here: relative to ``backbone.body.`` for
    the body's modules ('layer2', 'layer2.1', 'layer1.decoder', 'layer1.encoder'), to ``backbone.`` for the pyramid's
    ('fpn.layer_blocks.1')"""
    for pre in ('backbone.body.', 'backbone.'):
        if path.startswith(pre):
            return path[len(pre):]
    return path


def mimic_loss(t_hooked, s_hooked, terms):
    """loss.py:27-33: sum_k factor_k * MSELoss(reduction='sum')(teacher_k, student_k)."""
    per_term = OrderedDict()
    for name, factor in terms.items():
        # a term is `factor` (both tensors hooked under the term's own name) or (teacher key, student key, factor) with
        # keys relative to backbone.body: 'layer2', 'layer2.1' (a Bottleneck), 'layer1.decoder'
        t_key, s_key, factor = (name, name, factor) if not isinstance(factor, (tuple, list)) else factor
        per_term[name] = F.mse_loss(t_hooked[t_key], s_hooked[s_key], reduction='sum') * factor
    return sum(per_term.values()), per_term


# decoder.3.bias / decoder.8.bias feed an unpadded bias-free conv followed by a train-mode BN, so their true
# gradient is exactly zero; in fp32 it is rounding noise that Adam turns into +-lr steps.
# Parity checks on gradients / parameter trajectories must skip them.
ZERO_GRAD_KEYS = (B + 'layer1.decoder.3.bias', B + 'layer1.decoder.8.bias')

GHND_TERMS = OrderedDict((('layer1', 1.0), ('layer2', 1.0), ('layer3', 1.0), ('layer4', 1.0)))
HND_TERMS = OrderedDict((('layer1', 1.0),))


# ----------------------------------------------------------------------------- the step
class DistillOracle(object):
    """Mirrors mimic_runner.distill_model (:38-59) for a fixed teacher/student pair on CPU."""

    def __init__(self, teacher_sd, student_sd, terms=GHND_TERMS, lr=1e-3, min_size=(800,), max_size=1333,
                 warmup_iters=0, warmup_factor=1e-3, with_fpn=True, dtype=torch.float32, teacher_is_student_arch=False):
        """teacher_is_student_arch: the teacher is itself a bottleneck-injected model (config teacher_model = a custom
        backbone): run eval-mode through the student's layer1 -- needed for a term on backbone.body.layer1.encoder, which
        only such a teacher can pair a tensor with."""
        self.dtype = dtype
        self.teacher_is_student_arch = teacher_is_student_arch
        if any(isinstance(v, (tuple, list)) and (str(v[0]).startswith('fpn.') or str(v[1]).startswith('fpn.'))
               for v in terms.values()):
            with_fpn = True                     # a term on a pyramid map needs the pyramids
        self.t = cast_state(teacher_sd, dtype)
        self.s = cast_state(student_sd, dtype)
        self.terms, self.min_size, self.max_size, self.with_fpn = terms, min_size, max_size, with_fpn
        self.keys = trainable_keys(self.s)
        for k in self.keys:
            self.s[k] = self.s[k].clone().requires_grad_(True)
        # func_util.get_optimizer('Adam', {'lr': 1e-3}) over student.parameters(); frozen ones have no grad
        self.opt = torch.optim.Adam([self.s[k] for k in self.keys], lr=lr)
        self.sched = None
        if warmup_iters > 0:        # main_util.py:65-72, used for epoch 0 (mimic_runner.py:43-46)
            def f(x):
                if x >= warmup_iters:
                    return 1
                a = float(x) / warmup_iters
                return warmup_factor * (1 - a) + a
            self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, f)

    def forward(self, images, fixed_sizes=None, update_buffers=True, intermediates=None):
        images = [im.to(self.dtype) for im in images]
        x, _ = transform_images(images, self.min_size, self.max_size, training=False, fixed_sizes=fixed_sizes)
        with torch.no_grad():
            if self.teacher_is_student_arch:
                t_hooked, t_feats = backbone_forward(x, self.t, student=True, training=False, update_buffers=False,
                                                     with_fpn=self.with_fpn)
            else:
                t_hooked, t_feats = backbone_forward(x, self.t, student=False, with_fpn=self.with_fpn)
        # student transform in train mode draws random.choice(min_size) (rcnn.py:36-37): identical to
        # eval for single-size configs; Keypoint passes fixed_sizes (tool.py:45-48)
        s_hooked, s_feats = backbone_forward(x, self.s, student=True, training=True,
                                             update_buffers=update_buffers, with_fpn=self.with_fpn,
                                             intermediates=intermediates)
        loss, per_term = mimic_loss(t_hooked, s_hooked, self.terms)
        return loss, per_term, t_hooked, s_hooked, t_feats, s_feats, x

    def step(self, images, fixed_sizes=None):
        loss, per_term, *_ = self.forward(images, fixed_sizes)
        self.opt.zero_grad()
        loss.backward()
        # (a parameter no term reaches -- the decoder under a lone term on the bottleneck tensor -- has no gradient: zeros)
        grads = OrderedDict((k, torch.zeros_like(self.s[k]) if self.s[k].grad is None else self.s[k].grad.detach().clone())
                            for k in self.keys)
        self.opt.step()
        lr = self.opt.param_groups[0]['lr']
        if self.sched is not None:
            self.sched.step()
        return float(loss.detach()), OrderedDict((k, float(v.detach())) for k, v in per_term.items()), grads, lr


# ----------------------------------------------------------------------------- neural filter
EXT = B + 'layer1.encoder.ext_classifier.'
# Ext4ResNet.extractor (src/models/ext/classifier.py:19-31): index -> (kind, ...)
EXT_CONVS = ((1, 64, 64, 4, 2), (4, 64, 32, 3, 2), (7, 32, 16, 2, 1))      # (idx, cin, cout, k, stride)
EXT_BNS = ((2, 64), (5, 32), (8, 16))


class DetectOracle(object):
    """The eval-mode detector of the validation path (src/models/org/rcnn.py:103-134 with distill_backbone_only off:
    transform -> backbone -> rpn -> roi_heads -> transform.postprocess) composed from this module's functional backbone
    and the restated torchvision 0.4.2 heads (oracle/tv042_det.py, oracle/tv042.py), driven by a state dict.  The
    reference-generated fixtures (tiny_detect_*) pin those pieces; this composition lets tests run them at sizes no
    fixture stores (3x800x1333)."""

    def __init__(self, sd, model_name='faster_rcnn', student=False, min_size=(800,), max_size=1333):
        from oracle import tv042 as T
        from oracle import tv042_det as D
        self.sd, self.student = sd, student
        self.min_size = tuple(min_size) if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size = max_size
        ncls = sd['roi_heads.box_predictor.cls_score.weight'].shape[0]
        ag = D.AnchorGenerator(((32,), (64,), (128,), (256,), (512,)), ((0.5, 1.0, 2.0),) * 5)
        top_n = dict(training=2000, testing=1000)
        self.rpn = D.RegionProposalNetwork(ag, D.RPNHead(256, 3), 0.7, 0.3, 256, 0.5, dict(top_n), dict(top_n), 0.7)
        kw = {}
        if model_name == 'mask_rcnn':
            kw = dict(mask_roi_pool=D.MultiScaleRoIAlign([0, 1, 2, 3], 14, 2),
                      mask_head=T.MaskRCNNHeads(256, (256, 256, 256, 256), 1),
                      mask_predictor=T.MaskRCNNPredictor(256, 256, ncls))
        elif model_name == 'keypoint_rcnn':
            nkp = sd['roi_heads.keypoint_predictor.kps_score_lowres.weight'].shape[1]
            kw = dict(keypoint_roi_pool=D.MultiScaleRoIAlign([0, 1, 2, 3], 14, 2),
                      keypoint_head=T.KeypointRCNNHeads(256, (512,) * 8),
                      keypoint_predictor=T.KeypointRCNNPredictor(512, nkp))
        self.roi_heads = D.RoIHeads(D.MultiScaleRoIAlign([0, 1, 2, 3], 7, 2), D.TwoMLPHead(256 * 7 * 7, 1024),
                                    D.FastRCNNPredictor(1024, ncls), 0.5, 0.5, 512, 0.25, None, 0.05, 0.5, 100, **kw)
        for mod, prefix in ((self.rpn, 'rpn.'), (self.roi_heads, 'roi_heads.')):
            mod.load_state_dict(OrderedDict((k[len(prefix):], v) for k, v in sd.items() if k.startswith(prefix)),
                                strict=True)
            mod.eval()
        self.transform = T.GeneralizedRCNNTransform(self.min_size, max_size, IMAGE_MEAN, IMAGE_STD)
        self.transform.eval()
        self.image_list = T.ImageList

    @torch.no_grad()
    def __call__(self, images):
        original = [tuple(im.shape[-2:]) for im in images]
        batch, sizes = transform_images(images, self.min_size, self.max_size)
        _, features = backbone_forward(batch, self.sd, self.student, training=False, update_buffers=False)
        proposals, _ = self.rpn(self.image_list(batch, sizes), features)
        dets, _ = self.roi_heads(features, proposals, sizes)
        return self.transform.postprocess(dets, sizes, original)


def init_ext_state(seed, dtype=torch.float32):
    """Seeded init of Ext4ResNet(64) with torch's default nn.Conv2d / nn.Linear distributions; BatchNorm buffers
    are perturbed away from (0, 1) so that eval mode is a non-trivial check."""
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for (ci, cin, cout, k, _), (bi, c) in zip(EXT_CONVS, EXT_BNS):
        _conv_b(gen, sd, '%sextractor.%d.' % (EXT, ci), cin, cout, k, dtype)
        p = '%sextractor.%d.' % (EXT, bi)
        sd[p + 'weight'] = (torch.rand(c, generator=gen) * 0.5 + 0.75).to(dtype)
        sd[p + 'bias'] = (torch.randn(c, generator=gen) * 0.1).to(dtype)
        sd[p + 'running_mean'] = (torch.randn(c, generator=gen) * 0.1).to(dtype)
        sd[p + 'running_var'] = (torch.rand(c, generator=gen) * 0.5 + 0.75).to(dtype)
        sd[p + 'num_batches_tracked'] = torch.tensor(0, dtype=torch.int64)
    _linear(gen, sd, EXT + 'linear.', 16 * 8 * 8, 2, dtype)
    return sd


def ext_trainable_keys(sd):
    return [k for k in sd if k.startswith(EXT) and 'running_' not in k and 'num_batches' not in k]


def ext_classifier(x, sd, training=True, update_buffers=True, intermediates=None):
    """Ext4ResNet.forward (classifier.py:34-37) on the stem output x [N, 64, H, W]: logits in training mode,
    softmax probabilities in eval mode."""
    z = F.adaptive_avg_pool2d(x, (64, 64))
    for (ci, _, _, _, stride), (bi, _) in zip(EXT_CONVS, EXT_BNS):
        p = '%sextractor.%d.' % (EXT, ci)
        z = F.conv2d(z, sd[p + 'weight'], sd[p + 'bias'], stride=stride)
        z = F.relu(_train_bn(z, sd, '%sextractor.%d.' % (EXT, bi), training, update_buffers))
        if intermediates is not None:
            intermediates['extractor.%d' % (bi + 1)] = z
    z = F.adaptive_avg_pool2d(z, (8, 8))
    z = F.linear(z.flatten(1), sd[EXT + 'linear.weight'], sd[EXT + 'linear.bias'])
    return z if training else z.softmax(dim=1)


def valid_target(target, min_keypoints_per_image=10):
    """check_if_valid_target, src/models/ext/backbone.py:11-36: the image-level label of the neural filter."""
    if len(target) == 0:
        return False
    if all(any(float(o) <= 1 for o in box[2:]) for box in target['boxes']):
        return False
    if 'keypoints' not in target:
        return True
    return sum(sum(1 for row in kp if float(row[2]) > 0) for kp in target['keypoints']) >= min_keypoints_per_image


class FilterOracle(object):
    """ext_runner.train_model (:39-76) for a frozen student + trainable Ext4ResNet on CPU: cross entropy on the
    classifier logits, SGD(momentum, weight decay) and the epoch-0 warm-up.  As written in the reference, layer1's
    encoder/decoder also run in train mode (base.py:13-19,38-48) and so update their BatchNorm running statistics."""

    def __init__(self, student_sd, ext_sd, lr=1e-3, momentum=0.9, weight_decay=1e-4, min_size=(800,), max_size=1333,
                 warmup_iters=0, warmup_factor=1e-3, dtype=torch.float32):
        self.dtype = dtype
        self.s = cast_state(student_sd, dtype)
        self.s.update(cast_state(ext_sd, dtype))
        self.min_size, self.max_size = min_size, max_size
        self.keys = ext_trainable_keys(self.s)
        for k in self.keys:
            self.s[k] = self.s[k].clone().requires_grad_(True)
        self.opt = torch.optim.SGD([self.s[k] for k in self.keys], lr=lr, momentum=momentum,
                                   weight_decay=weight_decay)
        self.sched = None
        if warmup_iters > 0:
            def f(x):
                if x >= warmup_iters:
                    return 1
                a = float(x) / warmup_iters
                return warmup_factor * (1 - a) + a
            self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, f)

    def forward(self, images, training=True, fixed_sizes=None, update_buffers=True, intermediates=None):
        images = [im.to(self.dtype) for im in images]
        x, _ = transform_images(images, self.min_size, self.max_size, training=False, fixed_sizes=fixed_sizes)
        with torch.no_grad():
            x0 = stem(x, self.s)
        out = ext_classifier(x0, self.s, training, update_buffers, intermediates)
        gated = (not training) and out.shape[0] == 1          # base.py:15: early exit for batch-1 inference
        if training or not gated:
            with torch.no_grad():                             # frozen, but executed (and BN buffers updated)
                student_layer1(x0, self.s, training=training, update_buffers=update_buffers)
        return out

    def step(self, images, targets, fixed_sizes=None):
        logits = self.forward(images, True, fixed_sizes)
        labels = torch.tensor([1 if valid_target(t) else 0 for t in targets], dtype=torch.int64)
        loss = F.cross_entropy(logits, labels)
        self.opt.zero_grad()
        loss.backward()
        # (a parameter no term reaches -- the decoder under a lone term on the bottleneck tensor -- has no gradient: zeros)
        grads = OrderedDict((k, torch.zeros_like(self.s[k]) if self.s[k].grad is None else self.s[k].grad.detach().clone())
                            for k in self.keys)
        self.opt.step()
        lr = self.opt.param_groups[0]['lr']
        if self.sched is not None:
            self.sched.step()
        return float(loss.detach()), logits.detach().clone(), grads, lr


# conv biases in front of a train-mode BatchNorm: their true gradient is 0 (the batch mean absorbs them)
EXT_ZERO_GRAD_KEYS = tuple('%sextractor.%d.bias' % (EXT, ci) for ci, *_ in EXT_CONVS)


def synthetic_batch(batch, h=800, w=1333, seed=1234, rank=0, model_name='faster_rcnn'):
    """SURVEY.md section 8(d) synthetic inputs: uniform [0,1) images, one box per image."""
    g = torch.Generator().manual_seed(seed + rank)
    images = [torch.rand(3, h, w, generator=g) for _ in range(batch)]
    targets = []
    for _ in range(batch):
        t = {'boxes': torch.tensor([[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]], dtype=torch.float32),
             'labels': torch.tensor([1], dtype=torch.int64)}
        if model_name == 'mask_rcnn':
            t['masks'] = torch.zeros(1, h, w, dtype=torch.uint8)
        if model_name == 'keypoint_rcnn':
            t['keypoints'] = torch.zeros(1, 17, 3, dtype=torch.float32)
        targets.append(t)
    return images, targets


def checksum(t, nsamples=64):
    """Size-independent fingerprint of a tensor: (sum, sum of squares, strided samples) in fp64."""
    f = t.detach().double().flatten()
    n = min(nsamples, f.numel())
    idx = (torch.arange(n, dtype=torch.int64) * (f.numel() - 1)) // max(n - 1, 1)
    return float(f.sum()), float((f * f).sum()), f[idx].clone()

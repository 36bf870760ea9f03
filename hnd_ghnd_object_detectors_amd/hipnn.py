"""nn.Module building blocks whose forward runs on libhnd_hip.so.

Two kinds of classes:
  * parameter holders (Conv2d, BatchNorm2d, FrozenBatchNorm2d, ...) that keep the reference's
    state_dict layout and initialisation but execute only fused inside their
    parent (calling them directly raises: there is deliberately no eager/torch compute path);
  * composite modules (IntermediateLayerGetter, ResLayer, FeaturePyramidNetwork, BackboneWithFPN)
    that own an engine from ``engine.py`` and run its prebuilt HIP launch plan.

They play the role torchvision 0.4.2 plays for the reference (src/models/org/rcnn.py:6-17).
"""
from collections import OrderedDict

import torch
from torch import nn

from . import engine as E
from . import ops


class _FusedOnly(object):
    def forward(self, *args, **kwargs):
        raise RuntimeError('%s is a parameter holder: it executes fused inside its parent module on the HIP path '
                           '(no eager fallback exists)' % type(self).__name__)


class Conv2d(_FusedOnly, nn.Conv2d):
    pass


class ConvTranspose2d(_FusedOnly, nn.ConvTranspose2d):
    pass


class Linear(_FusedOnly, nn.Linear):
    pass


class BatchNorm2d(_FusedOnly, nn.BatchNorm2d):
    pass


class ReLU(_FusedOnly, nn.ReLU):
    pass


class MaxPool2d(_FusedOnly, nn.MaxPool2d):
    pass


class AdaptiveAvgPool2d(_FusedOnly, nn.AdaptiveAvgPool2d):
    pass


class FrozenBatchNorm2d(_FusedOnly, nn.Module):
    """torchvision 0.4.2 ops.misc.FrozenBatchNorm2d: four buffers, no eps, no num_batches_tracked."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer('weight', torch.ones(n))
        self.register_buffer('bias', torch.zeros(n))
        self.register_buffer('running_mean', torch.zeros(n))
        self.register_buffer('running_var', torch.ones(n))


def attach(t, buf, src=None):
    """tag a logical NCHW view with the NHWC buffer behind it (and who produced it)."""
    t._hnd = buf
    t._hnd_src = src
    return t


def fire_forward_hooks(module, inp, out):
    """Forward hooks of a module that executes FUSED inside its parent's engine (a Bottleneck of a ResLayer, the
    encoder / decoder of the bottleneck head, an FPN output conv): the parent calls this with the engine-produced input
    and output so that ``register_forward_hook`` on ANY of those dotted paths sees its tensors, as it would in the
    reference (src/distillation/tool.py:22-35).  A hook that returns a replacement output cannot be honoured."""
    hooks = getattr(module, '_forward_hooks', None)
    if not hooks:
        return
    for hook in list(hooks.values()):
        if hook(module, (inp,), out) is not None:
            raise RuntimeError('a forward hook on %s returned a new output: modules that execute fused inside their '
                               'parent cannot have their output replaced' % type(module).__name__)


def to_nhwc(x, pad_to=None):
    """logical NCHW tensor -> NHWC buffer.  Tensors produced by this package carry their buffer; foreign
    tensors are re-laid out once with torch copies (plumbing, off the distillation hot path)."""
    buf = getattr(x, '_hnd', None)
    if buf is not None:
        return buf
    if x.dim() != 4 or not x.is_cuda:
        raise RuntimeError('HIP modules take 4-d device tensors, got %s on %s' % (tuple(x.shape), x.device))
    c = x.shape[1]
    cs = pad_to or ops.chan_pad_of(c)
    buf = torch.zeros(x.shape[0], x.shape[2], x.shape[3], cs, dtype=torch.float32, device=x.device)
    buf[..., :c].copy_(x.permute(0, 2, 3, 1))
    return buf


# ------------------------------------------------------------------------------------------ resnet pieces
def conv3x3(in_planes, out_planes, stride=1):
    return Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class Bottleneck(_FusedOnly, nn.Module):
    """Holder with torchvision's Bottleneck attribute names (stride on conv2, v1.5)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        if groups != 1 or base_width != 64 or dilation != 1:
            raise NotImplementedError('HIP path implements the plain ResNet bottleneck (groups=1, width 64)')
        norm_layer = norm_layer or FrozenBatchNorm2d
        self.conv1 = conv1x1(inplanes, planes)
        self.bn1 = norm_layer(planes)
        self.conv2 = conv3x3(planes, planes, stride)
        self.bn2 = norm_layer(planes)
        self.conv3 = conv1x1(planes, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class BasicBlock(_FusedOnly, nn.Module):
    expansion = 1

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError('BasicBlock backbones (resnet18/34) are outside the HIP path (ResNet-50 configs)')


class ResLayer(nn.Sequential):
    """One ResNet stage (nn.Sequential of Bottleneck) executed by a FrozenLayerEngine."""

    def __init__(self, *blocks):
        super().__init__(*blocks)
        self._engine = None
        self._keep = False
        self._name = 'layer'

    def engine(self):
        if self._engine is None:
            for m in self:
                for bn in (m.bn1, m.bn2, m.bn3):
                    if not isinstance(bn, FrozenBatchNorm2d):
                        raise NotImplementedError('ResNet stages run with FrozenBatchNorm2d on the HIP path '
                                                  '(as built by rcnn.get_base_backbone)')
            self._engine = E.FrozenLayerEngine(list(self), self._name)
        return self._engine

    def hooked_blocks(self):
        return [i for i, blk in enumerate(self) if blk._forward_hooks]

    def _fire_block_hooks(self, eng, half=lambda t: t):
        """forward hooks registered on the Bottleneck blocks (``backbone.body.layer2.1`` ...): their outputs are the
        engine's per-block buffers (kept: see forward)"""
        for i in self.hooked_blocks():
            x_in, _, _, out = (half(t) for t in eng.acts[i])
            fire_forward_hooks(self[i], attach(E.logical(x_in), x_in),
                               attach(E.logical(out), out, (self.__dict__.get('_body'), self._name, i)))

    def forward(self, x):
        merged = getattr(self, '_merged', None)
        hooked = bool(self.hooked_blocks())                 # a hooked block's output must survive the layer: keep all
        if merged is not None and E.MERGE['trunk'] is merged[0]:
            # (trunk, role): this call is one half of a SharedTrunk pass (engine.SharedTrunk)
            trunk, role = merged
            out = trunk.layer_forward(self._name, role)
            self._used_engine = trunk.engines[self._name]
            if hooked:
                self._fire_block_hooks(self._used_engine, lambda t: trunk._half(t, role))
            return attach(E.logical(out), out)
        eng = self.engine()
        self._used_engine = eng
        eng.for_backward = bool(self._keep)          # (a teacher layer is kept for its hooks only: no backward-only extras)
        out = eng.forward(to_nhwc(x), self._keep or hooked)
        if hooked:
            self._fire_block_hooks(eng)
        return attach(E.logical(out), out)


class IntermediateLayerGetter(nn.ModuleDict):
    """torchvision models._utils.IntermediateLayerGetter: the stem (conv1, bn1, relu, maxpool children) runs as
    one fused engine; layer1..4 are called as modules so forward hooks registered on them fire
    (src/distillation/tool.py:25-35) and observe logical NCHW tensors."""

    def __init__(self, model, return_layers):
        if not set(return_layers).issubset([name for name, _ in model.named_children()]):
            raise ValueError('return_layers are not present in model')
        remaining = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)
        self._stem = None
        for name, module in self.items():
            if isinstance(module, ResLayer):
                module._name = name

    def stem(self):
        if self._stem is None:
            self._stem = E.StemEngine(self['conv1'], self['bn1'])
        return self._stem

    def needs_backward(self):
        return self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def forward(self, x):
        x4 = to_nhwc(x, 4)
        if x4.shape[3] != 4:
            raise RuntimeError('the stem expects the 3-channel image batch stored as NHWC4')
        keep = self.needs_backward()
        trunk = E.MERGE['trunk']
        role = trunk.role_of(self) if trunk is not None else None
        x0 = self.stem().forward(x4, keep)
        cur = attach(E.logical(x0), x0)
        out = OrderedDict()
        self._fwd_engines = {}
        for name, module in self.items():
            if name in ('conv1', 'bn1', 'relu', 'maxpool'):
                continue
            module.__dict__['_body'] = self          # (not a submodule registration: the parent for _hnd_src tags)
            if isinstance(module, ResLayer):
                module._keep = keep
                module._merged = (trunk, role) if (role is not None and name in trunk.LAYERS) else None
            own = module.engine() if (isinstance(module, ResLayer) and module._merged is None) else \
                (module.head_engine() if hasattr(module, 'head_engine') else None)
            if own is not None:
                # SharedTrunk pass: the layers in front of it write straight into this network's half of shared buffers
                own.out_provider = trunk.slot_provider(role, name) if (role is not None and name in trunk.FRONT) else None
            cur = module(cur)
            if role is not None and name == trunk.FRONT[-1]:
                trunk.delivered(role)
            if isinstance(cur, torch.Tensor):
                cur._hnd_src = (self, name, None)
            if isinstance(module, ResLayer):
                self._fwd_engines[name] = module._used_engine
            elif hasattr(module, 'head_engine'):
                self._fwd_engines[name] = module.head_engine()
            if name in self.return_layers:
                out[self.return_layers[name]] = cur
        self._last_keep = keep
        return out

    def layer_engine(self, name):
        """the engine that ran layer `name` in the LAST forward (its own, or the SharedTrunk's): what the loss and the
        backward plan must talk to"""
        return self._fwd_engines[name]

    # ------------------------------------------------------------------ manual backward (student)
    def trainable_plan(self):
        """(parameters that receive gradients, in state-dict order)."""
        return [p for _, p in self.named_parameters() if p.requires_grad]

    def hnd_backward(self, top, loss_grads, grad_dst, top_block=None, block_grads=None, fpn=None):
        """Run the hand-written backward.
        fpn: (FpnEngine, {level: loss gradient w.r.t. pyramid map `level`}) for terms on ``backbone.fpn.layer_blocks.K``
        of the STUDENT: the pyramid's backward runs first and delivers into the layer outputs' gradient buffers (the top
        layer's masked g_out, the lower layers' unmasked loss-gradient buffers).
        top: name of the highest layer that carries a loss term (its engine's g_out already holds the masked
        loss gradient); top_block: the block of that layer the gradient enters at (None = the layer output).
        loss_grads: {layer name: unmasked loss-gradient buffer} for lower layers with a term on their output.
        block_grads: {layer name: {block index: unmasked loss-gradient buffer}} for terms on inner Bottleneck outputs.
        grad_dst: {parameter: destination tensor} for every trainable parameter."""
        order = ['layer4', 'layer3', 'layer2', 'layer1']
        start = order.index(top)
        block_grads = block_grads or {}
        if fpn is not None:
            fpn_eng, term_grads, top_has_term = fpn
            assert top == 'layer4' and top_block is None
            sinks = {}
            for level in range(min(term_grads), 4):
                name = 'layer%d' % (level + 1)
                eng = self.layer_engine(name)
                if name == top:
                    # g_out of the top layer holds MASKED gradients; a body term on layer4 has already written its own
                    sinks[level] = (eng.grad_out_buffer(), eng.bwd_out(), bool(top_has_term))
                else:
                    have = name in loss_grads
                    if not have:
                        loss_grads = dict(loss_grads)
                        loss_grads[name] = eng.bufs.get('loss_grad', tuple(eng.bwd_out().shape))
                    sinks[level] = (loss_grads[name], None, have)
            fpn_eng.backward(term_grads, sinks)
        for name in order[start:-1]:
            prev_name = order[order.index(name) + 1]
            prev_eng = self.layer_engine(prev_name)
            dst = prev_eng.grad_out_buffer()
            self.layer_engine(name).backward(dst, prev_eng.bwd_out(), loss_grads.get(prev_name),
                                             top_block=top_block if name == top else None,
                                             block_grads=block_grads.get(name),
                                             dst_mask_bits=prev_eng.bwd_out_bits())
        l1 = self['layer1']
        conv1_w = self['conv1'].weight
        dw1 = grad_dst.get(conv1_w)
        if isinstance(l1, ResLayer):
            raise NotImplementedError('backward through a plain ResNet layer1 (teacher architecture) is not on the '
                                      'distillation path')
        g_x0 = l1.head_engine().backward(grad_dst, need_input_grad=dw1 is not None)
        self.stem().backward(g_x0, dw1)
        l1.head_engine().join_wgrad_stream()


class LastLevelMaxPool(nn.Module):
    """marker module (torchvision ops.feature_pyramid_network.LastLevelMaxPool); fused into the FPN engine."""

    def forward(self, x, names):
        raise RuntimeError('LastLevelMaxPool runs fused inside FeaturePyramidNetwork on the HIP path')


class FeaturePyramidNetwork(nn.Module):
    def __init__(self, in_channels_list, out_channels, extra_blocks=None):
        super().__init__()
        self.inner_blocks = nn.ModuleList()
        self.layer_blocks = nn.ModuleList()
        for c in in_channels_list:
            if c == 0:
                continue
            self.inner_blocks.append(Conv2d(c, out_channels, 1))
            self.layer_blocks.append(Conv2d(out_channels, out_channels, 3, padding=1))
        self.extra_blocks = extra_blocks
        self._engine = None

    def forward(self, x):
        names, feats = list(x.keys()), [to_nhwc(v) for v in x.values()]
        trunk = E.MERGE['trunk']
        role = None
        if trunk is not None:
            role = 0 if self is trunk.backbones[0].fpn else (1 if self is trunk.backbones[1].fpn else None)
        half = (lambda t: t)
        if role is not None:            # one half of a SharedTrunk pass: the pyramid of both networks in one plan
            outs = trunk.fpn_forward(role)
            eng, half = trunk.fpn_engine, (lambda t: trunk._half(t, role))
        else:
            if self._engine is None:
                self._engine = E.FpnEngine(list(self.inner_blocks), list(self.layer_blocks))
            outs = self._engine.forward(feats)
            eng = self._engine
        for i, m in enumerate(self.layer_blocks):           # hooks on backbone.fpn.layer_blocks.K: the pyramid maps
            if m._forward_hooks:
                fire_forward_hooks(m, None, attach(E.logical(outs[i]), outs[i], ('fpn', 'layer_blocks', i, self, eng)))
        for i, m in enumerate(self.inner_blocks):
            if m._forward_hooks:
                if i != len(self.inner_blocks) - 1:
                    raise NotImplementedError('backbone.fpn.inner_blocks.%d executes fused with the top-down add: its '
                                              'bare lateral output does not exist on the HIP path (the top level, '
                                              'inner_blocks.%d, does)' % (i, len(self.inner_blocks) - 1))
                t = half(eng.bufs.t['inner%d' % i])
                fire_forward_hooks(m, None, attach(E.logical(t), t, ('fpn', 'inner_blocks', i, self, eng)))
        if self.extra_blocks is None:
            outs = outs[:-1]
        else:
            names = names + ['pool']
        return OrderedDict((k, attach(E.logical(v), v)) for k, v in zip(names, outs))


class BackboneWithFPN(nn.Sequential):
    def __init__(self, backbone, return_layers, in_channels_list, out_channels):
        body = IntermediateLayerGetter(backbone, return_layers=return_layers)
        fpn = FeaturePyramidNetwork(in_channels_list, out_channels, extra_blocks=LastLevelMaxPool())
        super().__init__(OrderedDict([('body', body), ('fpn', fpn)]))
        self.out_channels = out_channels
        self.run_fpn = True      # the pyramid is dead w.r.t. the distillation loss; kept on for drop-in fidelity

    def forward(self, x):
        feats = self.body(x)
        self.fpn.__dict__['_body'] = self.body       # (a loss term on a pyramid map finds the backward plan through it)
        if not self.run_fpn:
            return feats
        side = E.DEFER_FPN['stream']
        if side is None or E.PROFILE['enabled'] or not x.is_cuda:
            return self.fpn(feats)
        # inside DistillationBox: the pyramid only depends on the layer outputs just produced on this stream and is
        # consumed by nobody during the step -> issue it on the side stream (joined before the next forward)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            return self.fpn(feats)


class ImageList(object):
    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


# ------------------------------------------------------------------------------------------ detector heads
# RPN / RoI heads (eval mode, the validation path of SURVEY.md 8f row f4) live in detection.py; the mask / keypoint
# heads below are parameter holders with torchvision 0.4.2 names and shapes (reference checkpoints load): RoIHeads runs
# their convolutions on hnd_conv2d_igemm straight from these parameters (detection.RoIHeads.mask_branch /
# keypoint_branch), so none of them has a forward of its own.
from .detection import (AnchorGenerator, RPNHead, RegionProposalNetwork, MultiScaleRoIAlign, TwoMLPHead,  # noqa: E402,F401
                        FastRCNNPredictor, RoIHeads)


class _NotOnPath(nn.Module):
    def forward(self, *args, **kwargs):
        raise NotImplementedError('%s is a parameter holder: detection.RoIHeads drives its layers on the HIP path'
                                  % type(self).__name__)


class MaskRCNNHeads(nn.Sequential):
    def __init__(self, in_channels, layers, dilation):
        d, nxt = OrderedDict(), in_channels
        for i, feat in enumerate(layers, 1):
            d['mask_fcn%d' % i] = Conv2d(nxt, feat, kernel_size=3, stride=1, padding=dilation, dilation=dilation)
            d['relu%d' % i] = ReLU(inplace=True)
            nxt = feat
        super().__init__(d)


class MaskRCNNPredictor(nn.Sequential):
    def __init__(self, in_channels, dim_reduced, num_classes):
        super().__init__(OrderedDict([('conv5_mask', ConvTranspose2d(in_channels, dim_reduced, 2, 2, 0)),
                                      ('relu', ReLU(inplace=True)),
                                      ('mask_fcn_logits', Conv2d(dim_reduced, num_classes, 1, 1, 0))]))


class KeypointRCNNHeads(nn.Sequential):
    def __init__(self, in_channels, layers):
        d, nxt = [], in_channels
        for feat in layers:
            d.append(Conv2d(nxt, feat, 3, stride=1, padding=1))
            d.append(ReLU(inplace=True))
            nxt = feat
        super().__init__(*d)


class KeypointRCNNPredictor(_NotOnPath):
    def __init__(self, in_channels, num_keypoints):
        super().__init__()
        self.kps_score_lowres = ConvTranspose2d(in_channels, num_keypoints, 4, stride=2, padding=1)
        self.up_scale, self.out_channels = 2, num_keypoints

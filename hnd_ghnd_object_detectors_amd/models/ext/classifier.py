"""Neural-filter classifier (mirror of the reference's src/models/ext/classifier.py).

``Ext4ResNet`` keeps the reference's module tree / state-dict keys (``extractor.{1,2,4,5,7,8}.*``, ``linear.*``)
and runs as one engine.FilterEngine plan on the stem output.  In training mode its logits are a leaf of a
``torch.autograd.Function`` over the 14 classifier tensors, so ext_runner's
``cross_entropy(ext_logits, ext_targets).backward()`` (src/ext_runner.py:58-72) replays the hand-written
backward plan and hands the gradients to autograd's AccumulateGrad.
"""
import torch
from torch import nn

from ... import engine as E
from ... import hipnn
from ...distillation.hip_loss import GradArena
from ...hipnn import to_nhwc


class BaseExtClassifier(nn.Module):
    def __init__(self, ext_idx):
        super().__init__()
        self.ext_idx = ext_idx

    def forward(self, *args):
        raise NotImplementedError('forward function is not implemented')


class _FilterLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, state, *params):
        ctx.state = state
        return logits.clone()

    @staticmethod
    def backward(ctx, grad_output):
        engine, arena, post_backward = ctx.state
        flat = arena.pick()
        grad_dst = {p: v for p, v in zip(arena.params, arena.views(flat))}
        engine.backward(grad_output.detach().float().contiguous(), grad_dst)
        del grad_dst
        if post_backward is not None:       # parallel.DistributedStudent: fire the gradient exchange
            post_backward(arena, flat)
        return (None, None) + tuple(arena.views(flat))       # fresh views: AccumulateGrad adopts them uncopied


class _CrossEntropyFn(torch.autograd.Function):
    """loss and dlogits come out of ONE launch (hnd_softmax_ce_rows_fwd_bwd); backward hands dlogits on, scaled by the
    incoming gradient on the device (no host read)"""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        from ... import ops
        logits = logits.detach()
        if not logits.is_contiguous() or logits.dtype != torch.float32:
            logits = logits.float().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits)
        ops.softmax_ce_rows(logits, labels.contiguous(), loss, dlogits, ignore_index)
        ctx.dlogits = dlogits
        return loss

    @staticmethod
    def backward(ctx, grad_output):
        from ... import ops
        if ctx.dlogits is None:
            raise RuntimeError('cross_entropy: backward ran twice (the gradient buffer is scaled in place)')
        g = grad_output.detach().float().reshape(1).contiguous()
        dlogits, ctx.dlogits = ctx.dlogits, None
        ops.scale_by_device_scalar(dlogits, g)
        return dlogits, None, None


def cross_entropy(logits, labels, ignore_index=-100):
    """``nn.functional.cross_entropy(logits, labels)`` (mean reduction) of the filter's training step, reference
    src/ext_runner.py:58, on the HIP path: [N, C] fp32 logits, [N] int64 labels -> scalar loss with autograd."""
    if not logits.is_cuda:
        raise RuntimeError('models.ext.classifier.cross_entropy runs on the HIP path only (no CPU fallback)')
    return _CrossEntropyFn.apply(logits, labels, ignore_index)


class Ext4ResNet(BaseExtClassifier):
    def __init__(self, input_channel):
        super().__init__(ext_idx=0)
        conv, bn, relu = hipnn.Conv2d, hipnn.BatchNorm2d, hipnn.ReLU
        self.extractor = nn.Sequential(
            hipnn.AdaptiveAvgPool2d((64, 64)),
            conv(input_channel, 64, kernel_size=4, stride=2), bn(64), relu(inplace=True),
            conv(64, 32, kernel_size=3, stride=2), bn(32), relu(inplace=True),
            conv(32, 16, kernel_size=2, stride=1), bn(16), relu(inplace=True),
            hipnn.AdaptiveAvgPool2d((8, 8)))
        self.linear = hipnn.Linear(16 * 8 * 8, 2)
        self._engine = None
        self._arena = None

    def engine(self):
        if self._engine is None:
            self._engine = E.FilterEngine(self.extractor, self.linear)
        return self._engine

    def forward(self, x):
        """x: stem output (logical [N, 64, H, W]).  Logits in training mode, softmax(dim=1) in eval (:34-37)."""
        eng = self.engine()
        out = eng.forward(to_nhwc(x), self.training)
        params = eng.params()
        if not (self.training and torch.is_grad_enabled() and any(p.requires_grad for p in params)):
            return out.clone()
        if not all(p.requires_grad for p in params):
            raise RuntimeError('the neural filter trains all of its tensors together (ext_runner.py:196-197)')
        if self._arena is None or [id(p) for p in self._arena.params] != [id(p) for p in params]:
            self._arena = GradArena(params)
        return _FilterLogitsFn.apply(out, (eng, self._arena, getattr(self, '_post_backward', None)), *params)


def get_ext_classifier(backbone):
    from ..custom.resnet import CustomResNet
    if isinstance(backbone, CustomResNet):
        return Ext4ResNet(64)
    raise ValueError('type of backbone `{}` is not expected'.format(type(backbone)))

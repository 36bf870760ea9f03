"""Fused L2 feature-mimic loss on the HIP path.

``HipMSELoss`` is what ``func_util.get_loss('MSELoss', {'reduction': 'sum'})`` returns;
``distill_loss`` fuses all terms of GeneralizedCustomLoss (reference src/distillation/loss.py:27-33) into ONE
launch that produces the scalar loss and, in the same pass over the features, the gradient w.r.t. every
student tensor.  ``loss.backward()`` then replays the hand-written backward plan of the student
(hipnn.IntermediateLayerGetter.hnd_backward) and hands parameter gradients to autograd's AccumulateGrad, so
``optimizer.zero_grad(); loss.backward(); optimizer.step()`` (src/mimic_runner.py:52-54) works unchanged.
"""
import os
import weakref

import torch
from torch import nn

from .. import ops
from ..hipnn import to_nhwc


class HipMSELoss(nn.Module):
    def __init__(self, reduction='mean', size_average=None, reduce=None):
        super().__init__()
        if reduction != 'sum':
            raise NotImplementedError("HIP path implements MSELoss(reduction='sum') (all hnd/ghnd configs)")
        self.reduction = reduction

    def forward(self, input, target):
        """standalone use: (teacher, student) -> loss with gradient to the student through distill_loss."""
        return distill_loss([('term', input, target, 1.0)])


class GradArena(object):
    """Flat fp32 gradient storage for the trainable tensors; two alternating arenas so that an autograd
    accumulation into a still-referenced .grad never aliases the buffer being written."""

    def __init__(self, params):
        self.params = list(params)
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 63) // 64 * 64
        self.total = total
        dev = self.params[0].device
        self.flat = [torch.zeros(total, dtype=torch.float32, device=dev) for _ in range(2)]
        self.cur = 0

    def pick(self):
        """choose the arena not referenced by any live .grad."""
        for k in (self.cur, 1 - self.cur):
            lo = self.flat[k].data_ptr()
            hi = lo + self.total * 4
            if not any(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.params):
                self.cur = k
                return self.flat[k]
        raise RuntimeError('both gradient arenas are referenced by live .grad tensors')

    def views(self, flat):
        return [flat[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, self.params)]


class _DistillLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss_value, state, *params):
        ctx.state = state
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_output):
        st = ctx.state
        body, arena = st['body'], st['arena']
        flat = arena.pick()
        views = arena.views(flat)
        grad_dst = {p: v for p, v in zip(arena.params, views)}
        # autograd hands the scalar dL/dloss (1 for loss.backward()); fold it in only if it is not 1
        go = grad_output.detach().reshape(()).float().contiguous()
        for buf in st['grad_bufs']:
            ops.scale_by_device_scalar(buf, go)
        body.hnd_backward(st['top'], st['loss_grads'], grad_dst, st['top_block'], st['block_grads'], fpn=st.get('fpn'))
        del views
        hook = getattr(body, '_post_backward', None)
        if hook is not None:        # parallel.DistributedStudent: the stem's wgrad (last kernel) is enqueued -> exchange
            hook(arena, flat)
        # fresh views (refcount 1) so AccumulateGrad adopts them without a copy
        return (None, None) + tuple(arena.views(flat))


class StepLoss(torch.Tensor):
    """The loss tensor of a training step.  The reference reads it on the host once per step, AFTER optimizer.step()
    (`metric_logger.update(loss=loss)` -> Tensor.item(), /root/reference/src/mimic_runner.py:58,
    src/utils/misc_util.py:147-152).  A plain .item() there drains the stream -- backward and Adam included -- and the GPU
    then idles until the host has enqueued the next step's first kernels (0.4 ms per step, tools/idle_gaps.py).  The value
    exists right after the loss kernel, so it is copied to pinned memory at that point and .item() waits for THAT copy only:
    same float, and the host is free to enqueue the next step while the backward pass still runs.  (Every loss owns its
    pinned scalar until it is collected, so losses kept unread across steps stay valid.)"""

    def item(self):
        host = self.__dict__.get('_host')
        if host is None:
            return super().item()
        buf, ev = host
        ev.synchronize()
        return buf.item()

    def __float__(self):
        return float(self.item())


EARLY_LOSS_COPY = True      # (settled in round 4; was HND_EARLY_LOSS_COPY)


_HOST_POOL = []         # free (pinned float32 scalar, event) pairs; a StepLoss owns one until it is garbage-collected


def _early_host_copy(loss_value):
    """(pinned buffer, event) holding loss_value, enqueued now on the current stream"""
    buf, ev = _HOST_POOL.pop() if _HOST_POOL else (torch.empty((), dtype=torch.float32).pin_memory(), torch.cuda.Event())
    buf.copy_(loss_value.detach(), non_blocking=True)
    ev.record()
    return buf, ev


_ORDER = ('layer1', 'layer2', 'layer3', 'layer4')
_SUPPORTED = ('backbone.body.layerN; backbone.body.layerN.K (a Bottleneck of layer2-4); backbone.body.layer1.decoder '
              '(= the layer1 output); backbone.body.layer1.encoder (the bottleneck tensor); backbone.fpn.layer_blocks.K '
              '(a pyramid map)')


def _student_position(src, body):
    """where a student-side tensor enters the backward plan: (layer name, block index or None = the layer output)"""
    if src is None or src[0] is not body or src[1] not in _ORDER:
        raise NotImplementedError('HIP distillation loss: the student tensor of a term must come from one of: %s -- the '
                                  'hand-written backward plan starts there' % _SUPPORTED)
    lname, sub = src[1], src[2] if len(src) > 2 else None
    if sub is None or sub == 'decoder':
        return lname, None
    if sub == 'encoder':
        return lname, 'encoder'             # below the layer1 output: the gradient enters in the middle of the head
    nblocks = len(body[lname])
    if not (isinstance(sub, int) and 0 <= sub < nblocks):
        raise NotImplementedError('unexpected source %r of a student tensor' % (src[1:],))
    return lname, (None if sub == nblocks - 1 else sub)


def distill_loss(terms):
    """terms: list of (name, teacher_out, student_out, factor).  Returns a 0-dim float32 loss tensor.
    Teacher tensors may be ANY tensor produced by the HIP path; student tensors are located in the backward plan by
    their ``_hnd_src`` tag (see _student_position)."""
    srcs = [getattr(s, '_hnd_src', None) for _, _, s, _ in terms]
    # a pyramid map on the student side is tagged ('fpn', 'layer_blocks', level, FeaturePyramidNetwork module, engine)
    is_fpn = [src is not None and isinstance(src[0], str) and src[0] == 'fpn' for src in srcs]
    body = None
    for src, f in zip(srcs, is_fpn):
        if src is not None:
            body = src[3].__dict__.get('_body') if f else src[0]
            break
    for src, f in zip(srcs, is_fpn):
        if f and (src[1] != 'layer_blocks' or src[3].__dict__.get('_body') is not body):
            raise NotImplementedError('student-side pyramid terms must sit on backbone.fpn.layer_blocks.K of the student; '
                                      'supported student-side paths: %s' % _SUPPORTED)
    trainable = body is not None and getattr(body, '_last_keep', False)
    pos = [('fpn', src[2]) if f else _student_position(src, body) for src, f in zip(srcs, is_fpn)]
    dev = terms[0][2].device
    fpn_levels = [p[1] for p in pos if p[0] == 'fpn']
    if fpn_levels and any(src[3].__dict__.get('_engine') is not src[4] for src, f in zip(srcs, is_fpn) if f):
        raise NotImplementedError('a loss term on a pyramid map with the shared trunk on (HND_MERGE_TRUNK=1) is not built')

    def rank(p):
        if p[0] == 'fpn':
            return (-2, 0)                  # never the entry point itself: its gradient reaches layer4 through the pyramid
        if p[1] == 'encoder':
            return (0, -1)                  # below the layer1 output
        nb = len(body[p[0]]) if p[0] != 'layer1' else 1
        return (_ORDER.index(p[0]), nb - 1 if p[1] is None else p[1])
    top = max(pos, key=rank)
    if fpn_levels:
        # a pyramid map depends on every coarser layer output down the top-down path: the backward starts at layer4
        if top[0] != 'fpn' and rank(top) > rank(('layer4', None)):
            raise NotImplementedError('unexpected position %r above layer4' % (top,))
        if any(p[0] == 'layer4' and p[1] is not None for p in pos):
            raise NotImplementedError('a pyramid term together with a term on an inner Bottleneck of layer4 is not built')
        top = ('layer4', None)
    pairs, loss_grads, block_grads, grad_bufs = [], {}, {}, []
    fpn_grads, fpn_state, enc_seen = {}, [None], [False]
    for (tname, t_out, s_out, factor), p in zip(terms, pos):
        t_buf, s_buf = to_nhwc(t_out), to_nhwc(s_out)
        if tuple(t_buf.shape) != tuple(s_buf.shape):
            raise ValueError('teacher/student shapes differ for term %s: %s vs %s'
                             % (tname, tuple(t_buf.shape), tuple(s_buf.shape)))
        grad = None
        if trainable and p[0] == 'fpn':
            fpn_eng = srcs[pos.index(p)][4]
            if p[1] in fpn_grads:
                raise NotImplementedError('two loss terms on pyramid map %d' % p[1])
            grad = fpn_grads[p[1]] = fpn_eng.term_grad_buffer(p[1])
            fpn_state[0] = fpn_eng
            grad_bufs.append(grad)
        elif trainable and p[1] == 'encoder':
            head = body.layer_engine('layer1')
            if enc_seen[0]:
                raise NotImplementedError('two loss terms on the bottleneck tensor')
            enc_seen[0] = True
            grad = head.enc_grad_buffer(top=(p == top))
            grad_bufs.append(grad)
        elif trainable:
            lname, blk = p
            eng = body.layer_engine(lname)          # its own engine, or the SharedTrunk's (engine.SharedTrunk)
            if p == top:
                # masked by the producing ReLU in the same pass
                grad = eng.grad_out_buffer(blk) if blk is not None else eng.grad_out_buffer()
            elif blk is None:
                grad = eng.bufs.get('loss_grad', s_buf.shape)
                if lname in loss_grads:
                    raise NotImplementedError('two loss terms on the output of %s' % lname)
                loss_grads[lname] = grad
            else:
                grad = eng.bufs.get('loss_grad_blk%d' % blk, s_buf.shape)
                if blk in block_grads.setdefault(lname, {}):
                    raise NotImplementedError('two loss terms on the output of %s.%d' % (lname, blk))
                block_grads[lname][blk] = grad
            grad_bufs.append(grad)
        # (masked by the producing ReLU only where the tensor IS a ReLU output that starts the backward: layer outputs /
        # Bottleneck outputs -- not the bottleneck tensor, a raw conv output, nor a pyramid map)
        pairs.append((t_buf, s_buf, grad, float(factor), p == top and p[0] != 'fpn' and p[1] != 'encoder'))
    if len([p for p in pos if p == top]) > 1:
        raise NotImplementedError('two loss terms on the same (top) student tensor')
    key = tuple((p[0].data_ptr(), p[1].data_ptr(), None if p[2] is None else p[2].data_ptr(), p[3], p[4])
                for p in pairs)
    cache = getattr(body, '_mse_cache', None)
    if cache is None or cache[0] != key:
        cache = (key, ops.MseLaunch(pairs, dev))
        body._mse_cache = cache
    out = cache[1].run()
    loss_value = out[0].float()
    per_term = out[1:]
    if not trainable:
        loss_value.per_term = per_term
        return loss_value
    params = body.trainable_plan()
    arena = getattr(body, '_grad_arena', None)
    if arena is None or [id(p) for p in arena.params] != [id(p) for p in params]:
        arena = GradArena(params)
        body._grad_arena = arena
    state = {'body': body, 'arena': arena, 'top': top[0], 'top_block': None if top[1] == 'encoder' else top[1],
             'loss_grads': loss_grads, 'block_grads': block_grads, 'grad_bufs': grad_bufs,
             'fpn': (fpn_state[0], fpn_grads, any(p == ('layer4', None) for p in pos)) if fpn_grads else None}
    loss = _DistillLossFn.apply(loss_value, state, *params)
    if EARLY_LOSS_COPY and loss.is_cuda:
        host = _early_host_copy(loss)
        loss = loss.as_subclass(StepLoss)
        loss._host = host
        weakref.finalize(loss, _HOST_POOL.append, host)     # the pair is reused only once nobody can read it any more
    loss.per_term = per_term
    return loss

"""Fused optimizers over one flat parameter arena: Adam (replaces torch.optim.Adam created at
src/mimic_runner.py:67-68) and SGD with momentum / weight decay (the neural filter's optimizer,
src/ext_runner.py:118-120 with config/ext/*.yaml).

Subclasses torch.optim.Adam so ``state_dict()`` / ``load_state_dict()`` keep the torch format the reference
checkpoints store under 'optimizer' (src/models/__init__.py:15-17) and LambdaLR / MultiStepLR drive
``param_groups[i]['lr']`` as usual.  ``step()`` is one hnd_adam_step_flat launch when the trainable tensors
and their gradients sit in flat arenas (the normal case), else one launch per tensor -- never torch math.
"""
import torch

from . import ops, parallel


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kwargs):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError('FusedAdam: weight_decay / amsgrad are not used by the hnd/ghnd configs')
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        self.grad_scale = 1.0          # extra factor on the gradients (tests); the DP mean comes from parallel
        self._flat = None

    # ---------------------------------------------------------------- flat arenas
    def _flatten(self, group, plist):
        """move the trainable tensors (and their Adam moments) into contiguous arenas; parameters keep their
        identity (only .data is re-pointed), so optimizers / DDP / state_dict are unaffected.  Tensors whose step
        counts differ (a hand-edited checkpoint) cannot share one launch: nothing is touched then and the
        per-tensor path is used from now on (decided once, not per step)."""
        ids = [id(p) for p in plist]
        steps = set(int(self.state[p]['step']) if 'exp_avg' in self.state[p] else 0 for p in plist)
        if len(steps) != 1:
            print('FusedAdam: per-tensor step counts differ (%s); using one launch per tensor' % sorted(steps))
            return {'ids': ids, 'per_tensor': True}
        offsets, total = [], 0
        for p in plist:
            offsets.append(total)
            total += (p.numel() + 63) // 64 * 64
        dev = plist[0].device
        flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_m, flat_v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
        for o, p in zip(offsets, plist):
            n = p.numel()
            flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = flat_p[o:o + n].view(p.shape)
            st = self.state[p]
            if 'exp_avg' in st:                       # resumed from a checkpoint
                flat_m[o:o + n].copy_(st['exp_avg'].reshape(-1))
                flat_v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            st['exp_avg'] = flat_m[o:o + n].view(p.shape)
            st['exp_avg_sq'] = flat_v[o:o + n].view(p.shape)
            if 'step' not in st:
                st['step'] = torch.tensor(0.0)
        return {'ids': ids, 'offsets': offsets, 'total': total, 'p': flat_p, 'm': flat_m, 'v': flat_v}

    def _grads_are_flat(self, flat, plist):
        base = plist[0].grad.data_ptr()
        for o, p in zip(flat['offsets'], plist):
            if p.grad.data_ptr() != base + o * 4 or not p.grad.is_contiguous():
                return None
        g0 = plist[0].grad
        span = g0.untyped_storage().nbytes() // 4 - (base - g0.untyped_storage().data_ptr()) // 4
        if span < flat['total']:
            return None
        return torch.as_strided(g0, (flat['total'],), (1,), g0.storage_offset())

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('FusedAdam.step(closure) is not supported')
        for group in self.param_groups:
            plist = [p for p in group['params'] if p.grad is not None]
            if not plist:
                continue
            beta1, beta2 = group['betas']
            flat = self._flat
            if flat is None or flat['ids'] != [id(p) for p in plist]:
                flat = self._flatten(group, plist)
                self._flat = flat
            flat_g = self._grads_are_flat(flat, plist) if not flat.get('per_tensor') else None
            # gradient all-reduces fired from inside backward (parallel.DistributedStudent): wait stream-side, and
            # fold the 1/world mean of a sum all-reduce into this launch
            grad_scale = self.grad_scale * parallel.finish_pending(flat_g if flat_g is not None else [p.grad for p in plist], plist)
            if flat_g is not None and all(p.data_ptr() == flat['p'].data_ptr() + o * 4
                                          for o, p in zip(flat['offsets'], plist)):
                step = int(self.state[plist[0]]['step']) + 1
                ops.adam_step_flat(flat['p'], flat_g, flat['m'], flat['v'], group['lr'], beta1, beta2, group['eps'],
                                   step, grad_scale)
                for p in plist:
                    self.state[p]['step'] = torch.tensor(float(step))
            else:
                for p in plist:
                    st = self.state[p]
                    if 'exp_avg' not in st:
                        st['step'] = torch.tensor(0.0)
                        st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    step = int(st['step']) + 1
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    ops.adam_step_flat(p.data, g, st['exp_avg'], st['exp_avg_sq'], group['lr'], beta1, beta2,
                                       group['eps'], step, grad_scale)
                    st['step'] = torch.tensor(float(step))
        parallel.end_step()
        return None


class FusedSGD(torch.optim.SGD):
    """torch.optim.SGD (momentum, dampening, weight_decay, nesterov) as one hnd_sgd_step_flat launch over flat
    parameter / gradient / momentum arenas; ``state_dict()`` keeps torch's ``momentum_buffer`` entries."""

    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, weight_decay=0, nesterov=False, **kwargs):
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                         nesterov=nesterov)
        self.grad_scale = 1.0
        self._flat = None

    def _flatten(self, plist):
        ids = [id(p) for p in plist]
        started = set(self.state[p].get('momentum_buffer') is not None for p in plist)
        if len(started) != 1:           # nothing touched; per-tensor path from now on (decided once)
            print('FusedSGD: some tensors have a momentum buffer and some do not; using one launch per tensor')
            return {'ids': ids, 'per_tensor': True}
        offsets, total = [], 0
        for p in plist:
            offsets.append(total)
            total += (p.numel() + 63) // 64 * 64
        dev = plist[0].device
        flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_b = torch.zeros_like(flat_p)
        for o, p in zip(offsets, plist):
            n = p.numel()
            flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = flat_p[o:o + n].view(p.shape)
            st = self.state[p]
            buf = st.get('momentum_buffer')
            if buf is not None:                       # resumed from a checkpoint
                flat_b[o:o + n].copy_(buf.reshape(-1))
            st['momentum_buffer'] = flat_b[o:o + n].view(p.shape) if buf is not None else None
        return {'ids': ids, 'offsets': offsets, 'total': total, 'p': flat_p, 'b': flat_b, 'started': started.pop()}

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('FusedSGD.step(closure) is not supported')
        for group in self.param_groups:
            plist = [p for p in group['params'] if p.grad is not None]
            if not plist:
                continue
            hyper = (group['lr'], group['momentum'], group['dampening'], group['weight_decay'], group['nesterov'])
            flat = self._flat
            if flat is None or flat['ids'] != [id(p) for p in plist]:
                flat = self._flatten(plist)
                self._flat = flat
            flat_g = FusedAdam._grads_are_flat(self, flat, plist) if not flat.get('per_tensor') else None
            grad_scale = self.grad_scale * parallel.finish_pending(flat_g if flat_g is not None else [p.grad for p in plist], plist)
            if flat_g is not None and all(p.data_ptr() == flat['p'].data_ptr() + o * 4
                                          for o, p in zip(flat['offsets'], plist)):
                ops.sgd_step_flat(flat['p'], flat_g, flat['b'], *hyper, first_step=not flat['started'],
                                  grad_scale=grad_scale)
                if not flat['started'] and group['momentum'] != 0:
                    for o, p in zip(flat['offsets'], plist):
                        self.state[p]['momentum_buffer'] = flat['b'][o:o + p.numel()].view(p.shape)
                flat['started'] = True
            else:
                for p in plist:
                    st = self.state[p]
                    first = st.get('momentum_buffer') is None
                    if first and group['momentum'] != 0:
                        st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    ops.sgd_step_flat(p.data, g, st.get('momentum_buffer'), *hyper, first_step=first,
                                      grad_scale=grad_scale)
        parallel.end_step()
        return None

"""Data parallelism for the distillation step: one process per GPU, RCCL over xGMI.

The reference wraps the student in torch DistributedDataParallel (src/mimic_runner.py:141-143): its training loop
stays ``zero_grad(); loss.backward(); optimizer.step()`` (:52-54) because DDP's hooks average the gradients of the
25 trainable tensors (586 566 floats) across ranks INSIDE backward, and it broadcasts every buffer at each forward.

``DistributedStudent`` keeps that contract.  The hand-written backward plan writes all gradients into ONE flat
arena; when its last kernel (the stem's weight gradient) has been enqueued it calls the hook registered here, which
fires a single in-place all-reduce of that arena on the communication stream; the fused optimizers wait for it
(stream-side, no host block) right before their launch and fold the 1/world mean into it.  No
``reduce_gradients()`` call is needed in the loop (it survives as a no-op-compatible explicit form).

Exchange back ends:
  * default: ``torch.distributed`` (backend 'nccl' == RCCL on ROCm) ``all_reduce(async_op=True)``; 'gloo' is what
    the CPU / shared-GPU tests use;
  * ``HND_NATIVE_COMM=1`` (backend 'nccl' only): the C ABI's own communicator -- ``hnd_comm_init`` /
    ``hnd_allreduce_avg_flat`` of include/hnd_hip.h (ncclAvg on a dedicated HIP stream), the id shipped over the
    existing process group.

BatchNorm statistics stay local to each rank exactly like the reference (no SyncBN).  Deviation, documented:
buffers are broadcast from rank 0 at construction and on ``sync_buffers()`` (before evaluation / checkpointing)
instead of at every forward -- training-mode outputs do not depend on the running statistics.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist
from torch import nn

# reductions in flight, keyed by the arena they belong to: id(arena) -> [flat tensor, waiter, scale, consumed].  Each
# fused optimizer group consumes ONLY the entry whose flat arena holds the gradients it is about to apply.
_PENDING = {}
# id(parameter) of every tensor a DistributedStudent of world > 1 is responsible for: an optimizer step over one of
# them with no exchanged gradient is an unsynchronised step and is refused
_GUARDED = set()


def _covers(flat, grad):
    lo = flat.data_ptr()
    return lo <= grad.data_ptr() < lo + flat.numel() * 4


def finish_pending(grads=None, params=()):
    """Called by FusedAdam / FusedSGD right before their launch with the gradients they are about to consume (the
    flat arena view, or the per-tensor gradients): make the current stream wait for the all-reduce of THAT arena and
    return the factor the summed gradients still have to be multiplied by (1/world for a sum all-reduce, 1.0 when the
    exchange already averaged or the tensors are not data-parallel).  Entries of other arenas stay pending for their
    own optimizer.  Raises when only part of the gradients was exchanged, or when a parameter guarded by a
    DistributedStudent is stepped without any exchange (it would train unsynchronised across ranks)."""
    if grads is None:
        grads = []
    elif isinstance(grads, torch.Tensor):
        grads = [grads]
    scale, hit = 1.0, 0
    for key in list(_PENDING):
        entry = _PENDING[key]
        flat, waiter, s, consumed = entry
        inside = [_covers(flat, g) for g in grads]
        if not any(inside):
            continue
        if not all(inside):
            raise RuntimeError('finish_pending: only %d of %d gradients of this optimizer group lie in the all-reduced '
                               'arena; the group would mix averaged and local gradients' % (sum(inside), len(inside)))
        if not consumed:
            waiter()
            entry[3] = True      # kept until end_step(): a later param group of the same optimizer whose gradients
        scale, hit = s, hit + 1  # lie in this arena too gets the same factor instead of "not all-reduced"
    if hit > 1:
        raise RuntimeError('finish_pending: the gradients of one optimizer group matched %d exchanged arenas' % hit)
    if not hit and any(id(p) in _GUARDED for p in params):
        raise RuntimeError('optimizer.step() on data-parallel parameters whose gradients were not all-reduced in this '
                           'step: loss.backward() must run through the DistributedStudent-wrapped model (the exchange '
                           'fires from inside its backward)')
    return scale


def end_step():
    """called by the fused optimizers when step() has visited all its param groups: exchanges consumed by this step are
    dropped, so a second step() without a new backward is refused like any other unsynchronised step"""
    for key in [k for k, e in _PENDING.items() if e[3]]:
        del _PENDING[key]


def _post(arena, flat, waiter, scale):
    """register the exchange of `arena`; a previous, never consumed one of the same arena (backward without an
    optimizer step) is completed and dropped first so entries cannot pile up"""
    old = _PENDING.pop(id(arena), None)
    if old is not None and not old[3]:
        old[1]()
    _PENDING[id(arena)] = [flat, waiter, scale, False]


class _NativeComm(object):
    """the C ABI's RCCL communicator (include/hnd_hip.h: hnd_comm_*), id distributed over torch.distributed"""

    def __init__(self, rank, world, device):
        from . import _lib
        self.lib = _lib.load()
        self.check = _lib.check
        nbytes = int(self.lib.hnd_workspace_size(5, None, 0))           # HND_OP_COMM_UNIQUE_ID
        buf = C.create_string_buffer(nbytes)
        if rank == 0:
            self.check(self.lib.hnd_comm_unique_id(buf, nbytes), 'hnd_comm_unique_id')
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=0)
        self.handle = C.c_void_p()
        with torch.cuda.device(device):
            self.check(self.lib.hnd_comm_init(rank, world, box[0], nbytes, C.byref(self.handle)), 'hnd_comm_init')
        self.stream = torch.cuda.Stream(device=device)

    def all_reduce_avg(self, flat):
        """enqueue on the communication stream, ordered after everything already on the current stream"""
        self.stream.wait_stream(torch.cuda.current_stream())
        self.check(self.lib.hnd_allreduce_avg_flat(self.handle, flat.data_ptr(), flat.numel(),
                                                   self.stream.cuda_stream), 'hnd_allreduce_avg_flat')
        done = torch.cuda.Event()
        done.record(self.stream)
        return lambda: torch.cuda.current_stream().wait_event(done)

    def close(self):
        if self.handle:
            self.lib.hnd_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class DistributedStudent(nn.Module):
    """Drop-in for DistributedDataParallel(student): ``.module`` is the wrapped model."""

    def __init__(self, module, optimizer=None, broadcast_buffers=False):
        """broadcast_buffers=True reproduces DDP's default (reference src/mimic_runner.py:141-143): rank 0's BatchNorm
        running statistics (the 2 182 floats + 8 counters of the head's train-mode BatchNorm layers; frozen buffers
        never change) are broadcast at the top of EVERY forward, so a checkpoint written mid-epoch by any rank holds
        rank 0's statistics.  Default False: broadcast at construction and on sync_buffers() (before validation /
        checkpointing, which is when mimic_runner reads them) -- training-mode outputs do not depend on them."""
        super().__init__()
        self.module = module
        self.broadcast_buffers = broadcast_buffers
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.optimizer = optimizer
        self.native = None
        self.reductions = 0            # all-reduces fired so far (tests / logging)
        # bench.py: with `timing` on, every exchange is bracketed by HIP events on the compute stream -- from "the last
        # gradient kernel is done" to "the reduced arena is visible to the optimizer launch" = the exposed exchange time
        self.timing, self.exchange_events = False, []
        if self.world > 1:
            for p in module.parameters():          # same start on every rank (DDP broadcasts at construction)
                dist.broadcast(p.data, 0)
            self.sync_buffers()
            if os.environ.get('HND_NATIVE_COMM', '0') != '0':
                if dist.get_backend() != 'nccl':
                    raise RuntimeError('HND_NATIVE_COMM=1 needs the nccl (= RCCL) process group: one GPU per rank')
                dev = next(module.parameters()).device
                self.native = _NativeComm(dist.get_rank(), self.world, dev)
            body = module.backbone.body
            body._post_backward = self._on_backward_done
            ext = body.get_ext_classifier() if hasattr(body, 'get_ext_classifier') else None
            if ext is not None:                    # neural-filter training: its 14 tensors have their own arena
                ext._post_backward = self._on_backward_done
            self._guarded = [id(p) for p in module.parameters() if p.requires_grad]
            _GUARDED.update(self._guarded)

    def close(self):
        """release the native RCCL communicator (before dist.destroy_process_group) and the step guard"""
        _GUARDED.difference_update(getattr(self, '_guarded', ()))
        self._guarded = []
        if self.native is not None:
            self.native.close()
            self.native = None

    def __del__(self):
        try:
            self.close()
        except Exception:        # interpreter teardown: the library may already be gone
            pass

    def attach_optimizer(self, optimizer):
        """kept for callers of the round-1 API; the fused optimizers pick the mean factor up by themselves"""
        self.optimizer = optimizer

    def sync_buffers(self):
        if self.world > 1:
            for b in self.module.buffers():
                dist.broadcast(b, 0)

    def _live_buffers(self):
        """buffers that training changes: those of train-mode BatchNorm layers (FrozenBatchNorm2d buffers are constants)"""
        out = []
        for m in self.module.modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.track_running_stats:
                out += [m.running_mean, m.running_var, m.num_batches_tracked]
        return out

    def broadcast_live_buffers(self):
        if self.world > 1:
            for b in self._live_buffers():
                dist.broadcast(b, 0)

    def forward(self, *args, **kwargs):
        if self.broadcast_buffers and self.module.training:
            self.broadcast_live_buffers()
        return self.module(*args, **kwargs)

    # ------------------------------------------------------------------ the exchange
    def _on_backward_done(self, arena, flat):
        """hook of the hand-written backward (distillation/hip_loss.py, models/ext/classifier.py): every gradient of
        this step has been ENQUEUED into `flat`; fire the exchange behind it."""
        if self.world == 1:
            return
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        for p in arena.params:
            if p.grad is not None and not (lo <= p.grad.data_ptr() < hi):
                # a live .grad in the OTHER arena means autograd will accumulate this step's gradient into it after
                # this hook: the arena being reduced would not be what the optimizer consumes
                raise RuntimeError('DistributedStudent: a parameter still holds a gradient from an earlier backward; '
                                   'call optimizer.zero_grad() (set_to_none) before loss.backward() as '
                                   'mimic_runner.distill_model does -- gradient accumulation is not supported')
        self.reductions += 1
        e0 = None
        if self.timing and flat.is_cuda:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.native is not None:
            waiter, scale = self.native.all_reduce_avg(flat), 1.0
        else:
            work = dist.all_reduce(flat, async_op=True)          # sum; 1/world is folded into the optimizer launch
            waiter, scale = work.wait, 1.0 / self.world
        if e0 is not None:
            inner = waiter

            def waiter():
                inner()
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.exchange_events.append((e0, e1))
        _post(arena, flat, waiter, scale)

    def exchange_ms(self):
        """mean exposed exchange time of the exchanges timed so far (call after a device synchronize)"""
        ev, self.exchange_events = self.exchange_events, []
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None

    def reduce_gradients(self):
        """explicit form of round 1's loop (between loss.backward() and optimizer.step()).  The exchange fires from
        inside backward, so for this package's models there is nothing left to reduce: every gradient of a GUARDED
        parameter (trainable when the wrapper was built) must lie in an exchanged arena, and one that does not is
        refused -- no fallback could tell an exchanged gradient from a local one there.  A trainable tensor that is
        NOT guarded (a module the caller attached or unfroze afterwards, whose backward is plain autograd) is averaged
        here per tensor, as DDP would have, so such a model keeps working; step it with an ordinary torch optimizer.
        (A custom optimizer that calls parallel.finish_pending() must also call parallel.end_step() when its step()
        is done -- FusedAdam / FusedSGD do -- or consumed exchanges would be served twice.)"""
        if self.world == 1:
            return
        for p in self.module.parameters():
            if p.grad is None or any(_covers(e[0], p.grad) for e in _PENDING.values()):
                continue                     # exchanged (or being exchanged) through its arena
            if id(p) in _GUARDED:
                raise RuntimeError('reduce_gradients: a data-parallel parameter has a gradient outside every exchanged '
                                   'arena: its backward bypassed the DistributedStudent hook, or optimizer.step() (which '
                                   'ends the exchange, parallel.end_step) already ran')
            dist.all_reduce(p.grad)
            p.grad.mul_(1.0 / self.world)


def all_reduce_flat_(flat, world):
    """helper used by tests: in-place sum all-reduce (+ host-side mean factor returned)."""
    if world > 1:
        dist.all_reduce(flat)
    return 1.0 / world

"""CPU, world_size 2 over gloo: the data-parallel path of the distillation step (one process per rank)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests import model_util as MU
    from hnd_ghnd_object_detectors_amd.distillation.hip_loss import GradArena
    from hnd_ghnd_object_detectors_amd.parallel import DistributedStudent
    from hnd_ghnd_object_detectors_amd.utils import main_util, misc_util
    distributed, device_ids = main_util.init_distributed_mode(backend='gloo')
    assert distributed and device_ids == [rank] and misc_util.get_world_size() == world
    cfg = MU.config_for()
    t_sd, s_sd = MU.oracle_states(9)
    _, student = MU.build_pair(cfg, t_sd, s_sd, torch.device('cpu'))
    with torch.no_grad():           # ranks start from different weights / running stats; the wrapper must equalise them
        for p in student.parameters():
            p.add_(float(rank))
        for b in student.buffers():
            if b.is_floating_point():
                b.add_(float(rank))
    wrapped = DistributedStudent(student)
    w0 = student.backbone.body.conv1.weight.detach().clone()
    rm = student.backbone.body.layer1.decoder[0].running_mean.detach().clone()

    from hnd_ghnd_object_detectors_amd import parallel
    body = student.backbone.body
    assert body._post_backward == wrapped._on_backward_done      # the backward plan's hook is installed
    params = body.trainable_plan()
    arena = GradArena(params)
    body._grad_arena = arena
    flat = arena.pick()
    flat.fill_(float(rank + 1))                       # rank-dependent gradients: 1 and 2
    body._post_backward(arena, flat)                  # what _DistillLossFn.backward calls after its last kernel
    assert len(parallel._PENDING) == 1
    scale = parallel.finish_pending(flat)             # what FusedAdam.step does before its launch
    # a SECOND param group of the same optimizer whose gradients lie in the same arena gets the same factor
    # (the consumed entry stays until the optimizer's step() ends) instead of "not all-reduced"
    half = arena.total // 2 // 64 * 64
    assert parallel.finish_pending(flat[half:], params) == scale and len(parallel._PENDING) == 1
    parallel.end_step()                               # what FusedAdam.step does after its last group
    assert not parallel._PENDING and parallel.finish_pending(flat) == 1.0         # nothing pending: no factor
    # (1) stepping data-parallel parameters with NO exchange in flight is refused (was: silently 1.0)
    try:
        parallel.finish_pending(flat, params)
        unsynced_refused = False
    except RuntimeError:
        unsynced_refused = True
    # (2) an optimizer group consumes only the entry of ITS arena; another arena's exchange stays pending
    other = GradArena(params[:2])
    oflat = other.pick()
    oflat.fill_(float(rank + 1))
    parallel._post(other, oflat, lambda: None, 0.25)
    body._post_backward(arena, flat)
    assert len(parallel._PENDING) == 2
    assert parallel.finish_pending(flat, params) == 0.5
    parallel.end_step()
    assert list(parallel._PENDING) == [id(other)]
    # (3) a group whose gradients only partly lie in an exchanged arena is refused
    try:
        parallel.finish_pending([oflat[:4], torch.zeros(4)])
        partial_refused = False
    except RuntimeError:
        partial_refused = True
    assert parallel.finish_pending(oflat) == 0.25
    parallel.end_step()
    assert not parallel._PENDING
    # (4) backward twice without a step: the older exchange of the same arena is completed and dropped, not piled up
    flat.fill_(float(rank + 1))
    body._post_backward(arena, flat)
    flat2 = arena.flat[1 - arena.cur]
    flat2.fill_(float(rank + 1))
    body._post_backward(arena, flat2)
    assert len(parallel._PENDING) == 1 and float(flat.min()) == 3.0
    assert parallel.finish_pending(flat2, params) == 0.5 and float(flat2.max()) == 3.0
    parallel.end_step()
    # reduce_gradients() averages a NON-guarded tensor (attached after the wrapper was built) per tensor,
    # and still refuses a guarded one whose gradient lies outside every exchanged arena
    extra = torch.nn.Parameter(torch.zeros(3))
    student.register_parameter('extra_head', extra)
    extra.grad = torch.full((3,), float(rank + 1))
    wrapped.reduce_gradients()
    fallback_ok = bool((extra.grad == 1.5).all())
    params[1].grad = torch.ones_like(params[1])
    try:
        wrapped.reduce_gradients()
        guarded_refused = False
    except RuntimeError:
        guarded_refused = True
    params[1].grad = None
    extra.grad = None
    del student._parameters['extra_head']
    # DDP's per-forward buffer broadcast as an option -- only train-mode BatchNorm buffers travel
    bn = student.backbone.body.layer1.decoder[0]
    with torch.no_grad():
        bn.running_mean.add_(float(rank))
        bn.num_batches_tracked.add_(rank)
    opt_in = DistributedStudent(student, broadcast_buffers=True)
    live = opt_in._live_buffers()
    opt_in.broadcast_live_buffers()
    bcast_ok = (len(live) == 3 * 8 and sum(b.numel() for b in live if b.is_floating_point()) == 2182
                and float(bn.running_mean.sum()) == float(rm.sum()) and int(bn.num_batches_tracked) == 0)
    opt_in.close()
    assert fallback_ok and guarded_refused and bcast_ok, (fallback_ok, guarded_refused, bcast_ok)
    wrapped.close()
    assert parallel.finish_pending(flat, params) == 1.0          # guard released with the wrapper
    wrapped._guarded = [id(q_) for q_ in params]
    parallel._GUARDED.update(wrapped._guarded)
    # a gradient left alive in the other arena (no zero_grad) must be refused, not silently mis-reduced
    params[0].grad = arena.views(arena.flat[1 - arena.cur])[0]
    try:
        body._post_backward(arena, flat)
        refused = False
    except RuntimeError:
        refused = True
    params[0].grad = None
    q.put((rank, float(w0.sum()), float(rm.sum()), float(flat.min()), float(flat.max()), scale,
           len(params), arena.total, misc_util.is_main_process() and refused and unsynced_refused and partial_refused))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_gradient_allreduce_and_parameter_broadcast():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, rm0, lo0, hi0, gs0, n0, tot0, main0), (r1, w1, rm1, lo1, hi1, gs1, n1, tot1, main1) = res
    assert w0 == w1 and rm0 == rm1                    # rank 0's parameters and buffers everywhere
    assert lo0 == hi0 == lo1 == hi1 == 3.0            # sum of the two ranks' flat gradient arenas
    assert gs0 == gs1 == 0.5                          # mean factor handed to the fused Adam launch
    assert n0 == n1 == 25 and tot0 == tot1 >= 586566
    assert main0 and not main1                        # (rank 0 also saw the stale-gradient refusal)


def _ext_worker(rank, world, port, q):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests import model_util as MU
    from hnd_ghnd_object_detectors_amd.distillation.hip_loss import GradArena
    from hnd_ghnd_object_detectors_amd.parallel import DistributedStudent
    from hnd_ghnd_object_detectors_amd.utils import main_util, misc_util
    main_util.init_distributed_mode(backend='gloo')
    s_sd, e_sd = MU.ext_states(3)
    _, model, ext = MU.build_ext_model(s_sd, e_sd, torch.device('cpu'), 64, 128)
    with torch.no_grad():
        for p in ext.parameters():
            p.add_(float(rank))
    wrapped = DistributedStudent(model)
    w0 = ext.linear.weight.detach().clone()

    from hnd_ghnd_object_detectors_amd import parallel
    params = ext.engine().params()
    ext._arena = GradArena(params)
    flat = ext._arena.pick()
    flat.fill_(float(rank + 1))
    ext._post_backward(ext._arena, flat)              # what _FilterLogitsFn.backward calls
    scale = parallel.finish_pending(flat)
    red = misc_util.reduce_dict({'loss_ext_classifier': torch.tensor(float(rank + 1))})
    q.put((rank, float(w0.sum()), float(flat.min()), float(flat.max()), scale, len(params),
           float(red['loss_ext_classifier'])))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_neural_filter_gradient_allreduce():
    """the filter's 14 tensors have their own flat arena: one all-reduce, mean folded into the SGD launch"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ext_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, lo0, hi0, gs0, n0, l0), (_, w1, lo1, hi1, gs1, n1, l1) = res
    assert w0 == w1 and lo0 == hi0 == lo1 == hi1 == 3.0 and gs0 == gs1 == 0.5 and n0 == n1 == 14
    assert l0 == l1 == 1.5

"""Worker of tests/test_model_gpu.py::test_rccl_exchange_world_of_one: a real RCCL process group (backend 'nccl') with
ONE rank on the one GPU a test box has.  The DDP replacement short-circuits at world 1, so the exchange the hook fires
at world > 1 is driven here by hand, exactly as parallel.DistributedStudent._on_backward_done does it:
async all_reduce of the flat gradient arena on RCCL's stream -> parallel._PENDING -> FusedAdam waits stream-side and
folds the mean factor.  With one rank the sum is the identity, so the result must equal the plain step bit for bit --
what is exercised is the RCCL launch, the async work handle and the stream ordering on a real device.
usage: python tests/rccl_world1_worker.py <out.pt> [native]"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, native = sys.argv[1], len(sys.argv) > 2 and sys.argv[2] == 'native'
    from hnd_ghnd_object_detectors_amd import parallel
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from oracle import hnd_oracle as O          # seeded states only (test infrastructure)
    from tests import golden_util as G
    from tests import model_util as MU
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', init_method='env://', rank=0, world_size=1, device_id=dev)
    _, meta = G.load('tiny_ghnd_faster')
    cfg = MU.config_for(meta)
    t_sd, s_sd = MU.oracle_states(meta['seed'])
    images, targets = G.case_inputs(meta)
    images = [im.to(dev) for im in images]
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    results = {}
    for mode in ('plain', 'rccl'):
        teacher, student = MU.build_pair(cfg, t_sd, s_sd, dev)
        box = DistillationBox(teacher, student, cfg['train']['criterion'])
        opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
        comm = parallel._NativeComm(0, 1, dev) if (mode == 'rccl' and native) else None
        fired = []

        def hook(arena, flat):
            if comm is not None:                                   # hnd_comm_init / hnd_allreduce_avg_flat (ncclAvg)
                parallel._post(arena, flat, comm.all_reduce_avg(flat), 1.0)
            else:                                                  # torch.distributed over RCCL, sum; mean folded later
                work = dist.all_reduce(flat, async_op=True)
                parallel._post(arena, flat, work.wait, 1.0 / dist.get_world_size())
            fired.append(flat.numel())
        if mode == 'rccl':
            student.backbone.body._post_backward = hook
        losses = []
        for _ in range(2):
            opt.zero_grad()
            loss = box(images, targets)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        assert not parallel._PENDING
        if mode == 'rccl':
            assert len(fired) == 2 and fired[0] > 100000, fired
            for b in student.buffers():                            # sync_buffers' broadcast on RCCL
                dist.broadcast(b, 0)
        torch.cuda.synchronize()
        results[mode] = {'losses': losses,
                         'params': {n: p.detach().cpu().clone() for n, p in student.named_parameters() if p.requires_grad}}
        if comm is not None:
            comm.close()
    torch.save(results, out)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""Worker of tests/test_model_gpu.py::test_two_ranks_take_the_oracle_step_on_the_mean_gradient.

Launched by torch.distributed.run (2 ranks, gloo, both ranks on cuda:0 -- the multi-GPU code path minus RCCL).
Each rank distils its OWN seeded batch through DistributedStudent with the reference's loop verbatim
(src/mimic_runner.py:52-54: zero_grad / backward / step -- no explicit reduce call) and rank r dumps its parameters
after every step to <out>/rank<r>.pt."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_batch(meta, rank):
    """rank-dependent seeded inputs (shared with the test, which feeds them to the oracle)"""
    g = torch.Generator().manual_seed(4321 + 17 * rank)
    images = [torch.rand(3, h, w, generator=g) for h, w in meta['sizes']]
    targets = [{'boxes': torch.tensor([[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]]), 'labels': torch.tensor([1])}
               for h, w in meta['sizes']]
    return images, targets


def main():
    out_dir, steps = sys.argv[1], int(sys.argv[2])
    from tests import golden_util as G
    from tests import model_util as MU
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from hnd_ghnd_object_detectors_amd.parallel import DistributedStudent
    from hnd_ghnd_object_detectors_amd.utils import main_util
    dist.init_process_group('gloo', init_method='env://')
    rank, dev = dist.get_rank(), torch.device('cuda:0')
    torch.cuda.set_device(dev)
    _, meta = G.load('tiny_ghnd_faster')
    cfg = MU.config_for(meta)
    t_sd, s_sd = MU.oracle_states(meta['seed'])
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, dev)
    wrapped = DistributedStudent(student)
    box = DistillationBox(teacher, wrapped, cfg['train']['criterion'])
    opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
    warm = main_util.warmup_lr_scheduler(opt, 4, 1e-3)
    images, targets = rank_batch(meta, rank)
    ims = [im.to(dev) for im in images]
    history = []
    for _ in range(steps):
        tgs = [{k: v.to(dev) for k, v in t.items()} for t in targets]
        loss = box(ims, tgs)
        opt.zero_grad()
        loss.backward()
        opt.step()
        warm.step()
        history.append({'loss': loss.item(),
                        'params': {n: p.detach().cpu().clone() for n, p in student.named_parameters()
                                   if p.requires_grad},
                        # after optimizer.step() the arena holds the all-reduced SUM of the ranks' gradients
                        'grad_sum': {n: p.grad.detach().cpu().clone() for n, p in student.named_parameters()
                                     if p.requires_grad}})
    torch.save({'history': history, 'reductions': wrapped.reductions}, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

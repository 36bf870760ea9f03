"""GPU debug: per-parameter gradient error of the HIP path and of the fp32 CPU oracle, both against fp64."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hnd_oracle as O
from tests import model_util as MU
from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util

DEV = torch.device('cuda:0')
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
cfg = MU.config_for(model='faster_rcnn', method='ghnd', bch=3, min_size=96, max_size=160)
t_sd, s_sd = MU.oracle_states(77)
teacher, student = MU.build_pair(cfg, t_sd, s_sd, DEV)
box = DistillationBox(teacher, student, cfg['train']['criterion'])
opt = func_util.get_optimizer(student, 'Adam', {'lr': lr})
orc = O.DistillOracle(t_sd, s_sd, min_size=(96,), max_size=160, lr=lr)
orc64 = O.DistillOracle(t_sd, s_sd, min_size=(96,), max_size=160, lr=lr, dtype=torch.float64)
g = torch.Generator().manual_seed(5)
for step in range(3):
    images = [torch.rand(3, h, w, generator=g) for h, w in ((70, 110), (96, 96), (50, 121))]
    targets = [{'boxes': torch.tensor([[1., 2., 30., 40.]]), 'labels': torch.tensor([1])} for _ in images]
    loss = box([im.to(DEV) for im in images], [{k: v.to(DEV) for k, v in t.items()} for t in targets])
    opt.zero_grad(); loss.backward()
    got = {n: p.grad.detach().cpu().double().clone() for n, p in student.named_parameters() if p.requires_grad}
    # sync the oracles' parameters to the HIP model's CURRENT parameters so each step is compared in isolation
    sd = {k: v.detach().cpu() for k, v in student.state_dict().items()}
    for o, dt in ((orc, torch.float32), (orc64, torch.float64)):
        with torch.no_grad():
            for k in o.keys:
                o.s[k].copy_(sd[k].to(dt))
            for k in o.s:
                if 'layer1' in k and ('running' in k or 'num_batches' in k):
                    pass
    l32, _, g32, _ = orc.step(images)
    l64, _, g64, _ = orc64.step(images)
    print('step %d loss hip %.6f  cpu32 %.6f  cpu64 %.6f' % (step, loss.item(), l32, l64))
    for n in got:
        ref = g64[n]
        e_hip = float((got[n] - ref).norm() / (ref.norm() + 1e-30))
        e_cpu = float((g32[n].double() - ref).norm() / (ref.norm() + 1e-30))
        print('   %-52s |g|=%.2e  hip-vs-fp64 %.2e   cpu32-vs-fp64 %.2e' % (n[14:], float(ref.norm()), e_hip, e_cpu))
    opt.step()

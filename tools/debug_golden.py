"""GPU debug: per-parameter gradient / updated-parameter error of the HIP path on a golden fixture's inputs, against
the fp64 oracle, beside the fp32 CPU oracle's own error.  usage: python tools/debug_golden.py [fixture]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hnd_oracle as O  # noqa: E402
from tests import golden_util as G  # noqa: E402
from tests import model_util as MU  # noqa: E402
from tests.test_model_gpu import _setup, _sync_oracle, _to_dev  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'tiny_ghnd_faster'
z, meta = G.load(name)
cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
if os.environ.get('DBG_LR') is not None:
    for g_ in opt.param_groups:
        g_['lr'] = g_['initial_lr'] = float(os.environ['DBG_LR'])
    warm.base_lrs = [float(os.environ['DBG_LR'])]
terms = MU.terms_of(cfg)
images, targets = G.case_inputs(meta)
ms = meta['min_size'] if isinstance(meta['min_size'], list) else [meta['min_size']]
orc64 = O.DistillOracle(t_sd, s_sd, terms=terms, min_size=tuple(ms), max_size=meta['max_size'], dtype=torch.float64)
orc32 = O.DistillOracle(t_sd, s_sd, terms=terms, min_size=tuple(ms), max_size=meta['max_size'])
for step in range(meta['steps']):
    ims, tgs = _to_dev(images, targets)
    _sync_oracle(orc64, student)
    _sync_oracle(orc32, student)
    _, _, g64, _ = orc64.step(images)
    _, _, g32, _ = orc32.step(images)
    loss = box(ims, tgs)
    opt.zero_grad()
    loss.backward()
    print('step %d loss %.6f ref %.6f' % (step, loss.item(), float(z['step%d/loss' % step])))
    for n, p in student.named_parameters():
        if p.requires_grad:
            r = g64[n].double()
            e_hip = float((p.grad.cpu().double() - r).norm() / (r.norm() + 1e-30))
            e_ref = float((g32[n].double() - r).norm() / (r.norm() + 1e-30))
            print('   %-50s |g| %.2e  hip %.2e  cpu32 %.2e  ratio %.2f' % (n[14:], float(r.norm()), e_hip, e_ref,
                                                                          e_hip / (e_ref + 1e-30)))
    opt.step()
    warm.step()
sd = student.state_dict()
for n in O.trainable_keys(s_sd):
    key = 'after/param/' + n
    if key in z.files:
        ref = torch.from_numpy(z[key]).double()
        t = sd[n].cpu().double()
        print('after %-50s rel %.2e  |ref| %.2e max|d| %.2e' % (n[14:], float((t - ref).norm() / (ref.norm() + 1e-6 * ref.numel() ** 0.5)),
                                                               float(ref.norm()), float((t - ref).abs().max())))

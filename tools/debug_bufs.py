"""GPU debug: fingerprints (sum, sum of squares) of every named engine buffer of the student after the backward of
step 1 -> json (to diff two builds)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import golden_util as G  # noqa: E402
from tests.test_model_gpu import _setup, _to_dev  # noqa: E402

z, meta = G.load('tiny_ghnd_faster')
cfg, t_sd, s_sd, teacher, student, box, opt, warm = _setup(meta)
images, targets = G.case_inputs(meta)
out = {}
for step in range(2):
    ims, tgs = _to_dev(images, targets)
    loss = box(ims, tgs)
    opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    body = student.backbone.body
    engines = {'stem': body.stem(), 'layer1': body.layer1.head_engine(), 'layer2': body.layer_engine('layer2'),
               'layer3': body.layer_engine('layer3'), 'layer4': body.layer_engine('layer4')}
    for en, e in engines.items():
        for bn, t in sorted(e.bufs.t.items()):
            if t.is_floating_point():
                d = t.double()
                out['step%d/%s/%s' % (step, en, bn)] = [float(d.sum()), float((d * d).sum()), list(t.shape)]
    if step == 1:
        break
    opt.step()
    warm.step()
json.dump(out, open(sys.argv[1], 'w'), indent=0)
if len(sys.argv) > 2:
    keep = {}
    for en, e in engines.items():
        for bn, t in e.bufs.t.items():
            if t.is_floating_point() and t.numel() <= 400000 and bn not in ('slabs', 'wino_slabs'):
                keep['%s/%s' % (en, bn)] = t.detach().cpu().clone()
    keep['grads'] = {n: p.grad.detach().cpu().clone() for n, p in student.named_parameters() if p.grad is not None}
    torch.save(keep, sys.argv[2])


"""smoke(): one tiny GHND distillation step on cuda:0 through the HIP path, checked against the CPU oracle."""
import os
import sys

import torch


def run_smoke():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import hnd_oracle as O          # checker only
    from . import mimic_runner
    from .configs import make_config
    from .distillation.tool import DistillationBox
    from .models import get_model
    from .myutils.pytorch import func_util, module_util

    dev = torch.device('cuda:0')
    cfg = make_config('faster_rcnn', 'ghnd', 3, pretrained=False, min_size=64, max_size=128, ckpt_root='/nonexistent')
    t_sd = O.init_teacher_state(1)
    s_sd = O.init_student_state(t_sd, 2)
    teacher = get_model(cfg['teacher_model'], dev)
    student = get_model(cfg['student_model'], dev)
    teacher.load_state_dict(t_sd)
    student.load_state_dict(s_sd)
    module_util.freeze_module_params(teacher)
    mimic_runner.freeze_modules(student, cfg['student_model'])
    teacher.eval()
    student.train()
    teacher.distill_backbone_only = student.distill_backbone_only = True
    box = DistillationBox(teacher, student, cfg['train']['criterion'])
    opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
    g = torch.Generator().manual_seed(0)
    images = [torch.rand(3, 64, 96, generator=g), torch.rand(3, 60, 90, generator=g)]
    targets = [{'boxes': torch.tensor([[4., 4., 40., 30.]]), 'labels': torch.tensor([1])} for _ in images]
    loss = box([im.to(dev) for im in images], [{k: v.to(dev) for k, v in t.items()} for t in targets])
    opt.zero_grad()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    orc = O.DistillOracle(t_sd, s_sd, min_size=(64,), max_size=128)
    ref_loss, _, ref_grads, _ = orc.step(images)
    rel = abs(loss.item() - ref_loss) / abs(ref_loss)
    assert rel < 1e-3, 'smoke: loss %.6f vs oracle %.6f (rel %.2e)' % (loss.item(), ref_loss, rel)
    worst = 0.0
    for n, p in student.named_parameters():
        if p.requires_grad and not n.endswith(('decoder.3.bias', 'decoder.8.bias')):
            gr = ref_grads[n].double()
            worst = max(worst, float((p.grad.cpu().double() - gr).norm() / gr.norm()))
    assert worst < 2e-3, 'smoke: gradient rel-L2 %.2e' % worst
    print('smoke ok: loss %.4f (oracle %.4f, rel %.1e), worst grad rel-L2 %.1e' % (loss.item(), ref_loss, rel, worst))

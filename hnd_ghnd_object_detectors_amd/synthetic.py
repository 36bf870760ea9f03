"""Seeded stand-ins for the COCO-pretrained weights (not downloadable: no network).

The reference starts from torchvision's COCO detector weights (src/models/org/rcnn.py:444-450) and copies every
non-layer1 tensor into the student (strict=False).  For benchmarks / smoke runs the constructors' random init is
used instead, with the FrozenBatchNorm2d buffers re-randomised so activations stay O(1) through ~50 layers
(recipe of SURVEY.md C.6: var in [0.5, 2], mean ~ N(0, 0.1^2), gamma in [0.25, 0.75], beta ~ N(0, 0.1^2)).
"""
import torch

from .hipnn import FrozenBatchNorm2d


def randomize_frozen_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, FrozenBatchNorm2d):
            n = m.weight.numel()
            m.weight.copy_(torch.rand(n, generator=g) * 0.5 + 0.25)
            m.bias.copy_(torch.randn(n, generator=g) * 0.1)
            m.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(n, generator=g) * 1.5 + 0.5)


def build_distillation_pair(config, device, seed=0):
    """teacher + student prepared as mimic_runner.main/distill do (frozen sets, train/eval flags)."""
    from . import mimic_runner
    from .models import get_model
    from .myutils.pytorch import module_util
    for part in ('teacher_model', 'student_model'):
        config[part]['params']['pretrained'] = False
        config[part]['backbone']['params']['pretrained'] = False
    torch.manual_seed(seed)
    cpu = torch.device('cpu')
    teacher = get_model(config['teacher_model'], cpu)
    student = get_model(config['student_model'], cpu)
    with torch.no_grad():
        randomize_frozen_bn(teacher, seed + 1)
    student.load_state_dict(teacher.state_dict(), strict=False)      # rcnn.py:446-450
    teacher, student = teacher.to(device), student.to(device)
    module_util.freeze_module_params(teacher)
    mimic_runner.freeze_modules(student, config['student_model'])
    teacher.eval()
    student.train()
    teacher.distill_backbone_only = True
    student.distill_backbone_only = config['student_model']['distill_backbone_only']
    student.backbone.body.layer1.use_bottleneck_transformer = False
    return teacher, student

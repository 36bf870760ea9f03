"""Shared helpers: build HIP-path teacher/student pairs from golden metadata / oracle state dicts."""
from collections import OrderedDict

import torch

from oracle import hnd_oracle as O


def config_for(meta=None, model='faster_rcnn', method='ghnd', bch=3, min_size=None, max_size=None):
    from hnd_ghnd_object_detectors_amd.configs import make_config
    if meta is not None:
        model, bch = meta['model'], meta.get('bch', 3)
        method = 'hnd' if meta['yaml'].startswith('hnd/') else 'ghnd'
        min_size, max_size = meta['min_size'], meta['max_size']
    return make_config(model, method, bch, pretrained=False, min_size=min_size, max_size=max_size,
                       ckpt_root='/nonexistent')


def oracle_states(seed, model='faster_rcnn', bch=3, num_classes=91):
    t_sd = O.init_teacher_state(seed, model, num_classes=num_classes)
    s_sd = O.init_student_state(t_sd, seed + 1000, bch=bch)
    return t_sd, s_sd


def build_pair(config, t_sd, s_sd, device):
    """teacher/student on `device`, prepared exactly like mimic_runner.main + distill (train/eval flags)."""
    from hnd_ghnd_object_detectors_amd import mimic_runner
    from hnd_ghnd_object_detectors_amd.models import get_model
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import module_util
    teacher = get_model(config['teacher_model'], device)
    student = get_model(config['student_model'], device)
    teacher.load_state_dict(t_sd, strict=True)
    student.load_state_dict(s_sd, strict=True)
    module_util.freeze_module_params(teacher)
    mimic_runner.freeze_modules(student, config['student_model'])
    teacher.eval()
    student.train()
    teacher.distill_backbone_only = True
    student.distill_backbone_only = True
    student.backbone.body.layer1.use_bottleneck_transformer = False
    return teacher, student


def terms_of(config):
    return OrderedDict((k, v['factor']) for k, v in config['train']['criterion']['terms'].items())


def ext_states(seed):
    t_sd = O.init_teacher_state(seed, 'keypoint_rcnn', num_classes=2)
    return O.init_student_state(t_sd, seed + 1000), O.init_ext_state(seed + 2000)


def build_ext_model(s_sd, e_sd, device, min_size, max_size, threshold=0.01):
    """the neural-filter model prepared exactly like ext_runner.main"""
    from hnd_ghnd_object_detectors_amd.configs import make_ext_config
    from hnd_ghnd_object_detectors_amd.models import get_model
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import module_util
    cfg = make_ext_config(3, pretrained=False, min_size=min_size, max_size=max_size, threshold=threshold,
                          ckpt_root='/nonexistent')
    model = get_model(cfg['model'], device, strict=False)
    full = OrderedDict(s_sd)
    full.update(e_sd)
    model.load_state_dict(full, strict=True)
    module_util.freeze_module_params(model)
    ext = model.get_ext_classifier()
    module_util.unfreeze_module_params(ext)
    model.train_ext()
    return cfg, model, ext

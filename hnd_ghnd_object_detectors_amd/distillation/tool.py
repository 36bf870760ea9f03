"""DistillationBox (mirror of the reference's src/distillation/tool.py).

Same contract: forward hooks on the teacher/student module pairs named by ``ts_modules`` stash their outputs
(:19-20, :25-35); ``forward(images, targets)`` runs the teacher (no targets) then the student (with targets),
gathers the hooked outputs into ``output_dict`` (:53-58) and applies the criterion (:60).
"""
import os
import random

import torch
from torch import nn

from .. import engine as E
from ..models.org.rcnn import KeypointRCNN
from ..myutils.pytorch import module_util
from .loss import get_loss


def _unwrap(model):
    return model.module if hasattr(model, 'module') and not hasattr(model, 'transform') else model


class DistillationBox(nn.Module):
    def __init__(self, teacher_model, student_model, criterion_config):
        super().__init__()
        self.teacher_model = teacher_model
        self.student_model = student_model
        self.target_module_pairs = list()

        def extract_output(self, input, output):
            self.__dict__['distillation_box']['output'] = output

        teacher, student = _unwrap(teacher_model), _unwrap(student_model)
        for loss_name, loss_config in criterion_config['terms'].items():
            teacher_path, student_path = loss_config['ts_modules']
            self.target_module_pairs.append((teacher_path, student_path))
            for model, path, is_teacher in ((teacher, teacher_path, True), (student, student_path, False)):
                module = module_util.get_module(model, path)
                module.__dict__['distillation_box'] = {'loss_name': loss_name, 'path_from_root': path,
                                                       'is_teacher': is_teacher}
                module.register_forward_hook(extract_output)
        self.criterion = get_loss(criterion_config)
        self.require_adjustment = isinstance(student, KeypointRCNN)
        # The frozen teacher forward is independent of the student forward: run it on a second HIP stream so the
        # tails of one network's launches (grids that do not fill all CUs) are filled by the other's blocks.
        self.overlap_teacher = os.environ.get('HND_TEACHER_STREAM', '1') != '0'
        self._side_stream = None

    def forward(self, images, targets):
        teacher, student = _unwrap(self.teacher_model), _unwrap(self.student_model)
        E.transform_scope_begin()        # the student's identical transform reuses the teacher's batch
        try:
            fixed_sizes = None
            if self.require_adjustment:      # reference :45-48
                fixed_sizes = [random.choice(teacher.transform.min_size) for _ in images]
            kw = {} if fixed_sizes is None else {'fixed_sizes': fixed_sizes}
            overlap = self.overlap_teacher and images[0].is_cuda and not E.PROFILE['enabled']
            if overlap:
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(device=images[0].device)
                main = torch.cuda.current_stream()
                teacher.transform(images, None, fixed_sizes)       # shared batch produced once, on the main stream
                self._side_stream.wait_stream(main)
                with torch.cuda.stream(self._side_stream):
                    self.teacher_model(images, **kw)
                org_loss_dict = self.student_model(images, targets, **kw)
                main.wait_stream(self._side_stream)
            else:
                self.teacher_model(images, **kw)
                org_loss_dict = self.student_model(images, targets, **kw)
        finally:
            E.transform_scope_end()
        output_dict = dict()
        for teacher_path, student_path in self.target_module_pairs:
            t = module_util.get_module(teacher, teacher_path).__dict__['distillation_box']
            s = module_util.get_module(student, student_path).__dict__['distillation_box']
            output_dict[t['loss_name']] = ((t['path_from_root'], t['output']), (s['path_from_root'], s['output']))
        return self.criterion(output_dict, org_loss_dict)

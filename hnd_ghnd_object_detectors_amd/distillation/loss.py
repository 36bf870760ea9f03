"""Distillation criterion (role of the reference's src/distillation/loss.py).

``criterion(output_dict, org_loss_dict)`` returns  sum_k factor_k * criterion_k(teacher_k, student_k); the
reference adds ``org_loss_factor`` times the detector losses when that factor is non-zero (:32-34), which no
hnd/ghnd config uses and which needs RPN / RoI heads (outside this build).  Every term is evaluated by ONE fused
HIP launch that also writes the gradients (hip_loss.distill_loss).
"""
from collections import namedtuple

from torch import nn

from ..myutils.pytorch import func_util
from .hip_loss import HipMSELoss, distill_loss

Term = namedtuple('Term', 'ts_modules criterion factor')


class CustomLoss(nn.Module):
    """parses the YAML ``criterion`` section into ``term_dict[name] = (ts_modules, criterion, factor)``."""

    def __init__(self, criterion_config):
        super().__init__()
        self.org_loss_factor = criterion_config['params']['org_loss_factor']
        self.term_dict = {}
        for name, cfg in criterion_config['terms'].items():
            sub = cfg['criterion']
            self.term_dict[name] = Term(cfg['ts_modules'], func_util.get_loss(sub['type'], sub['params']),
                                        cfg['factor'])

    def forward(self, *args, **kwargs):
        raise NotImplementedError('forward function is not implemented')


class GeneralizedCustomLoss(CustomLoss):
    def forward(self, output_dict, org_loss_dict):
        if self.org_loss_factor != 0:
            raise NotImplementedError('org_loss_factor != 0 needs the detector losses (RPN / RoI heads), which are '
                                      'outside the distillation hot path of this build')
        fused = []
        for name, (teacher_side, student_side) in output_dict.items():
            term = self.term_dict[name]
            if not isinstance(term.criterion, HipMSELoss):
                raise NotImplementedError('only MSELoss(sum) terms run on the HIP path')
            fused.append((name, teacher_side[1], student_side[1], term.factor))
        return distill_loss(fused)


LOSS_DICT = {'general': GeneralizedCustomLoss}


def get_loss(criterion_config):
    kind = criterion_config['type']
    if kind not in LOSS_DICT:
        raise ValueError('criterion_type `{}` is not expected'.format(kind))
    return LOSS_DICT[kind](criterion_config)

"""Distillation criterion (mirror of the reference's src/distillation/loss.py).

``GeneralizedCustomLoss.forward(output_dict, org_loss_dict)`` = sum_k factor_k * criterion_k(teacher_k, student_k)
(+ org_loss_factor * detector losses when non-zero, reference :32-34 -- zero in every hnd/ghnd config).
All terms are evaluated by one fused HIP launch (hip_loss.distill_loss).
"""
from torch import nn

from ..myutils.pytorch import func_util
from .hip_loss import HipMSELoss, distill_loss


class CustomLoss(nn.Module):
    def __init__(self, criterion_config):
        super().__init__()
        self.org_loss_factor = criterion_config['params']['org_loss_factor']
        term_dict = dict()
        for loss_name, loss_config in criterion_config['terms'].items():
            sub = loss_config['criterion']
            term_dict[loss_name] = (loss_config['ts_modules'], func_util.get_loss(sub['type'], sub['params']),
                                    loss_config['factor'])
        self.term_dict = term_dict

    def forward(self, *args, **kwargs):
        raise NotImplementedError('forward function is not implemented')


class GeneralizedCustomLoss(CustomLoss):
    def forward(self, output_dict, org_loss_dict):
        terms = []
        for loss_name, ((_, teacher_output), (_, student_output)) in output_dict.items():
            _, criterion, factor = self.term_dict[loss_name]
            if not isinstance(criterion, HipMSELoss):
                raise NotImplementedError('only MSELoss(sum) terms run on the HIP path')
            terms.append((loss_name, teacher_output, student_output, factor))
        if self.org_loss_factor != 0:
            raise NotImplementedError('org_loss_factor != 0 needs the detector losses (RPN / RoI heads), which are '
                                      'outside the distillation hot path of this build')
        return distill_loss(terms)


LOSS_DICT = {'general': GeneralizedCustomLoss}


def get_loss(criterion_config):
    criterion_type = criterion_config['type']
    if criterion_type in LOSS_DICT:
        return LOSS_DICT[criterion_type](criterion_config)
    raise ValueError('criterion_type `{}` is not expected'.format(criterion_type))

"""Factories the runner calls by name (src/mimic_runner.py:67-70, src/distillation/loss.py:13).

get_optimizer('Adam' / 'SGD') returns the fused flat-arena optimizers, get_loss('MSELoss', reduction='sum') the fused
HIP loss; schedulers are plain torch (host-side scalars only)."""
import torch
from torch import nn

from ...optim import FusedAdam, FusedSGD
from ...distillation.hip_loss import HipMSELoss


def get_optimizer(target, optim_type, optim_params_config):
    params = target.parameters() if isinstance(target, nn.Module) else target
    if optim_type.lower() == 'adam':
        return FusedAdam(params, **optim_params_config)
    if optim_type.lower() == 'sgd':
        return FusedSGD(params, **optim_params_config)
    raise ValueError('optim_type `{}` is not expected on the HIP path (the hnd/ghnd configs use Adam, the ext '
                     'config SGD)'.format(optim_type))


def get_scheduler(optimizer, scheduler_type, scheduler_params_config):
    for name in ('StepLR', 'MultiStepLR', 'ExponentialLR', 'CosineAnnealingLR'):
        if scheduler_type.lower() == name.lower():
            return getattr(torch.optim.lr_scheduler, name)(optimizer, **scheduler_params_config)
    raise ValueError('scheduler_type `{}` is not expected'.format(scheduler_type))


def get_loss(loss_type, param_dict=None):
    param_dict = param_dict or {}
    if loss_type.lower() in ('mse', 'mseloss'):
        return HipMSELoss(**param_dict)
    raise ValueError('loss_type `{}` is not expected on the HIP distillation path (the hnd/ghnd configs use '
                     'MSELoss)'.format(loss_type))

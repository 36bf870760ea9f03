"""Restatement of the ``myutils`` symbols the reference path touches.

TEST INFRASTRUCTURE (see oracle/__init__.py).  ``myutils`` is an un-vendored,
un-pinned git submodule of the reference (``.gitmodules:1-3``, ``README.md:42-43``)
and its directory is empty under /root/reference, so behaviour is restated from
the public repository as used at these call sites:
``src/distillation/loss.py:13``, ``src/distillation/tool.py:28-29,55-56``,
``src/mimic_runner.py:34-35,68,70,125,132,136``, ``src/models/__init__.py:12,21``,
``src/structure/transformer.py:85,101,139,152``.
"""
import os
import pickle
import sys

import torch
import yaml
from torch import nn
from torch.nn import DataParallel
from torch.nn.parallel import DistributedDataParallel


# ------------------------------------------------------------------ common.yaml_util
def _join(loader, node):
    return ''.join(str(i) for i in loader.construct_sequence(node))


class _Loader(yaml.FullLoader):
    pass


_Loader.add_constructor('!join', _join)


def load_yaml_file(file_path):
    with open(file_path, 'r') as fp:
        return yaml.load(fp, Loader=_Loader)


# ------------------------------------------------------------------ common.file_util
def check_if_exists(file_path):
    return file_path is not None and os.path.exists(file_path)


def make_dirs(dir_path):
    os.makedirs(dir_path, exist_ok=True)


def make_parent_dirs(file_path):
    parent = os.path.dirname(file_path)
    if parent:
        os.makedirs(parent, exist_ok=True)


def get_binary_object_size(x, unit_size=1024):
    return sys.getsizeof(pickle.dumps(x)) / unit_size


def get_file_path_list(dir_path, is_recursive=False, is_sorted=False):
    out = []
    for name in os.listdir(dir_path):
        p = os.path.join(dir_path, name)
        if os.path.isfile(p):
            out.append(p)
        elif is_recursive:
            out.extend(get_file_path_list(p, is_recursive))
    return sorted(out) if is_sorted else out


# ------------------------------------------------------------------ pytorch.func_util
def get_optimizer(target, optim_type, optim_params_config):
    params = target.parameters() if isinstance(target, nn.Module) else target
    lowered = optim_type.lower()
    for name in ('SGD', 'Adam', 'Adagrad', 'RMSprop'):
        if lowered == name.lower():
            return getattr(torch.optim, name)(params, **optim_params_config)
    raise ValueError('optim_type `{}` is not expected'.format(optim_type))


def get_scheduler(optimizer, scheduler_type, scheduler_params_config):
    lowered = scheduler_type.lower()
    for name in ('StepLR', 'MultiStepLR', 'ExponentialLR', 'CosineAnnealingLR'):
        if lowered == name.lower():
            return getattr(torch.optim.lr_scheduler, name)(optimizer, **scheduler_params_config)
    raise ValueError('scheduler_type `{}` is not expected'.format(scheduler_type))


def get_loss(loss_type, param_dict=None):
    param_dict = param_dict or {}
    table = {'nll': nn.NLLLoss, 'bcewithlogits': nn.BCEWithLogitsLoss, 'crossentropy': nn.CrossEntropyLoss,
             'crossentropyloss': nn.CrossEntropyLoss, 'kldiv': nn.KLDivLoss, 'kldivloss': nn.KLDivLoss,
             'mse': nn.MSELoss, 'mseloss': nn.MSELoss, 'l1': nn.L1Loss, 'l1loss': nn.L1Loss,
             'smoothl1': nn.SmoothL1Loss, 'smoothl1loss': nn.SmoothL1Loss}
    key = loss_type.lower()
    if key in table:
        return table[key](**param_dict)
    raise ValueError('loss_type `{}` is not expected'.format(loss_type))


# ------------------------------------------------------------------ pytorch.module_util
def _unwrap(module):
    return module.module if isinstance(module, (DataParallel, DistributedDataParallel)) else module


def get_module(root_module, module_path):
    module = _unwrap(root_module)
    for name in module_path.split('.'):
        module = getattr(module, name)
    return module


def freeze_module_params(module):
    for p in module.parameters():
        p.requires_grad = False


def unfreeze_module_params(module):
    for p in module.parameters():
        p.requires_grad = True


def get_updatable_param_names(module):
    return [name for name, p in module.named_parameters() if p.requires_grad]


def count_params(module):
    return sum(p.numel() for p in module.parameters())


def get_components(module_paths, model):
    return [get_module(model, p) for p in module_paths]


# ------------------------------------------------------------------ pytorch.tensor_util
class QuantizedTensor(object):
    def __init__(self, tensor, scale, zero_point):
        self.tensor, self.scale, self.zero_point = tensor, scale, zero_point


def quantize_tensor(x, num_bits=8):
    qmin, qmax = 0.0, 2.0 ** num_bits - 1.0
    min_val, max_val = x.min(), x.max()
    scale = (max_val - min_val) / (qmax - qmin)
    initial_zero_point = qmin - min_val / scale
    if initial_zero_point < qmin:
        zero_point = qmin
    elif initial_zero_point > qmax:
        zero_point = qmax
    else:
        zero_point = initial_zero_point
    zero_point = int(zero_point)
    qx = zero_point + x / scale
    qx.clamp_(qmin, qmax).round_()
    return QuantizedTensor(tensor=qx.round().byte(), scale=scale, zero_point=zero_point)


def dequantize_tensor(q_x):
    return q_x.scale * (q_x.tensor.float() - q_x.zero_point)

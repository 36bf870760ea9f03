"""Model factory and checkpoint I/O (mirror of the reference's src/models/__init__.py:11-70)."""
import os

import torch
from torch import nn

from ..myutils.common import file_util
from ..utils import misc_util
from ..structure.transformer import get_bottleneck_transformer
from .org import rcnn


def save_ckpt(model, optimizer, lr_scheduler, best_value, config, args, output_file_path):
    file_util.make_parent_dirs(output_file_path)
    model = getattr(model, 'module', model) if not hasattr(model, 'transform') else model
    misc_util.save_on_master({'model': model.state_dict(), 'optimizer': optimizer.state_dict(),
                              'best_value': best_value, 'lr_scheduler': lr_scheduler.state_dict(), 'config': config,
                              'args': args}, output_file_path)


def load_ckpt(ckpt_file_path, model=None, optimizer=None, lr_scheduler=None, strict=True):
    if not file_util.check_if_exists(ckpt_file_path):
        print('ckpt file is not found at `{}`'.format(ckpt_file_path))
        return None, None                      # 2-tuple when missing, 3-tuple otherwise (reference :23,:35)
    ckpt = torch.load(ckpt_file_path, map_location='cpu', weights_only=False)
    if model is not None:
        print('Loading model parameters')
        model.load_state_dict(ckpt['model'], strict=strict)
    if optimizer is not None:
        print('Loading optimizer parameters')
        optimizer.load_state_dict(ckpt['optimizer'])
    if lr_scheduler is not None:
        print('Loading scheduler parameters')
        lr_scheduler.load_state_dict(ckpt['lr_scheduler'])
    return ckpt.get('best_value', 0.0), ckpt['config'], ckpt['args']


def get_model(model_config, device, strict=True, bottleneck_transformer=None):
    model_name = model_config['name']
    if model_name not in rcnn.MODEL_CLASS_DICT:
        raise ValueError('model_name `{}` is not expected'.format(model_name))
    backbone_config = model_config['backbone']
    if bottleneck_transformer is None and 'bottleneck_transformer' in model_config:
        bottleneck_transformer = get_bottleneck_transformer(model_config['bottleneck_transformer'])
    model = rcnn.get_model(model_name, backbone_config=backbone_config, strict=strict,
                           bottleneck_transformer=bottleneck_transformer, **model_config['params'])
    load_ckpt(model_config['ckpt'], model=model, strict=strict)
    return model.to(device)


def get_iou_types(model):
    model = getattr(model, 'module', model) if not hasattr(model, 'transform') else model
    iou_type_list = ['bbox']
    if isinstance(model, rcnn.MaskRCNN):
        iou_type_list.append('segm')
    if isinstance(model, rcnn.KeypointRCNN):
        iou_type_list.append('keypoints')
    return iou_type_list

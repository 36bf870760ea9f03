"""Checkpoint dictionary I/O.

File format is the reference's (src/models/__init__.py:11-35): one ``torch.save``d dict with the keys
``model`` (state dict), ``optimizer``, ``best_value``, ``lr_scheduler``, ``config``, ``args``; written by rank 0
only.  ``load_ckpt`` keeps the reference's return quirk: ``(None, None)`` when the file is missing, a 3-tuple
``(best_value, config, args)`` otherwise.
"""
import torch

from ..myutils.common import file_util
from ..utils import misc_util

CKPT_KEYS = ('model', 'optimizer', 'best_value', 'lr_scheduler', 'config', 'args')


def unwrap(model):
    """strip a data-parallel wrapper (anything exposing .module that is not itself a detector)."""
    if hasattr(model, 'module') and not hasattr(model, 'transform'):
        return model.module
    return model


def save_ckpt(model, optimizer, lr_scheduler, best_value, config, args, output_file_path, extra=None):
    payload = dict(zip(CKPT_KEYS, (unwrap(model).state_dict(), optimizer.state_dict(), best_value,
                                   lr_scheduler.state_dict(), config, args)))
    if extra:           # this build's additions ride under their own keys (never overloading `best_value`)
        payload.update(extra)
    file_util.make_parent_dirs(output_file_path)
    misc_util.save_on_master(payload, output_file_path)


def load_ckpt(ckpt_file_path, model=None, optimizer=None, lr_scheduler=None, strict=True):
    if not file_util.check_if_exists(ckpt_file_path):
        print('ckpt file is not found at `{}`'.format(ckpt_file_path))
        return None, None
    ckpt = torch.load(ckpt_file_path, map_location='cpu', weights_only=False)
    restore = (('model', model, 'Loading model parameters', {'strict': strict}),
               ('optimizer', optimizer, 'Loading optimizer parameters', {}),
               ('lr_scheduler', lr_scheduler, 'Loading scheduler parameters', {}))
    for key, target, message, kwargs in restore:
        if target is not None:
            print(message)
            target.load_state_dict(ckpt[key], **kwargs)
    return ckpt.get('best_value', 0.0), ckpt['config'], ckpt['args']

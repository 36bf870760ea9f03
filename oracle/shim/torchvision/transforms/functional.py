import numpy as _np
import torch as _torch


def to_tensor(pic):
    arr = _np.asarray(pic)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = _torch.from_numpy(arr.transpose(2, 0, 1).copy())
    return t.float().div(255) if t.dtype == _torch.uint8 else t

"""Runner utilities (mirror of the reference's src/utils/main_util.py:14-72; COCO `evaluate` is out of scope)."""
import builtins as __builtin__
import json
import os

import torch


def overwrite_dict(org_dict, sub_dict):
    for key, value in sub_dict.items():
        if key in org_dict and isinstance(value, dict):
            overwrite_dict(org_dict[key], value)
        else:
            org_dict[key] = value


def overwrite_config(config, json_str):
    overwrite_dict(config, json.loads(json_str))


def setup_for_distributed(is_master):
    """silence print on non-master ranks (pass force=True to override)."""
    builtin_print = __builtin__.print

    def print(*args, **kwargs):
        force = kwargs.pop('force', False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    __builtin__.print = print


def init_distributed_mode(world_size=1, dist_url='env://', backend=None):
    """One process per GPU; backend 'nccl' is RCCL on ROCm (reference main_util.py:43-62).  `backend` may be set
    to 'gloo' for CPU tests."""
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        rank = int(os.environ['RANK'])
        world_size = int(os.environ['WORLD_SIZE'])
        device_id = int(os.environ.get('LOCAL_RANK', 0))
    elif 'SLURM_PROCID' in os.environ:
        rank = int(os.environ['SLURM_PROCID'])
        device_id = rank % max(torch.cuda.device_count(), 1)
    else:
        print('Not using distributed mode')
        return False, None
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        torch.cuda.set_device(device_id)
    print('| distributed init (rank {}): {}'.format(rank, dist_url), flush=True)
    torch.distributed.init_process_group(backend=backend, init_method=dist_url, world_size=world_size, rank=rank)
    torch.distributed.barrier()
    setup_for_distributed(rank == 0)
    return True, [device_id]


def warmup_factor_at(x, warmup_iters, warmup_factor):
    if x >= warmup_iters:
        return 1
    alpha = float(x) / warmup_iters
    return warmup_factor * (1 - alpha) + alpha


def warmup_lr_scheduler(optimizer, warmup_iters, warmup_factor):
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda x: warmup_factor_at(x, warmup_iters, warmup_factor))


def evaluate(model, data_loader, device):
    raise NotImplementedError('COCO mAP evaluation (RPN / RoIAlign / NMS + pycocotools) is outside the distillation '
                              'hot path of this build (SURVEY.md section 8f, row f4)')

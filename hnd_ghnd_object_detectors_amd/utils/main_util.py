"""Runner utilities (mirror of the reference's src/utils/main_util.py:14-113, incl. the COCO bbox `evaluate` of the validation path)."""
import builtins as __builtin__
import json
import os

import torch


def effective_cpu_count():
    """CPUs this process may actually use: the scheduler affinity capped by the cgroup CPU quota (a container on a
    256-thread GPU host is typically given 16): torch sizes its intra-op pool by the logical CPU count and then
    oversubscribes the quota 8-fold (host staging copies ran at 1.8-3.9 GB/s that way, round 5)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]           # cgroup v2
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())        # cgroup v1
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def limit_host_threads(local_world=1):
    """size torch's intra-op pool to this rank's share of the usable CPUs (never more than the default)"""
    share = max(1, effective_cpu_count() // max(local_world, 1))
    if torch.get_num_threads() > share:
        torch.set_num_threads(share)
    return share


def overwrite_dict(org_dict, sub_dict):
    for key, value in sub_dict.items():
        if key in org_dict and isinstance(value, dict):
            overwrite_dict(org_dict[key], value)
        else:
            org_dict[key] = value


def overwrite_config(config, json_str):
    overwrite_dict(config, json.loads(json_str))


def setup_for_distributed(is_master):
    """after this call plain print() is silent on non-master ranks; print(..., force=True) always prints."""
    original = __builtin__.print

    def rank_aware_print(*args, force=False, **kwargs):
        if force or is_master:
            original(*args, **kwargs)

    __builtin__.print = rank_aware_print


def _rank_from_env():
    env = os.environ
    if 'RANK' in env and 'WORLD_SIZE' in env:
        return int(env['RANK']), int(env['WORLD_SIZE']), int(env.get('LOCAL_RANK', 0))
    if 'SLURM_PROCID' in env:
        rank = int(env['SLURM_PROCID'])
        return rank, None, rank % max(torch.cuda.device_count(), 1)
    return None


def init_distributed_mode(world_size=1, dist_url='env://', backend=None):
    """One process per GPU (reference main_util.py:43-62).  Backend 'nccl' is RCCL on ROCm; 'gloo' is accepted for
    CPU tests.  Returns (distributed, [device_id])."""
    found = _rank_from_env()
    if found is None:
        print('Not using distributed mode')
        return False, None
    rank, env_world, device_id = found
    world_size = env_world or world_size
    has_gpu = torch.cuda.is_available()
    # test knobs: exercise the multi-rank runner on a single-GPU box (every rank on cuda:0, gloo instead of RCCL)
    backend = backend or os.environ.get('HND_DIST_BACKEND')
    if os.environ.get('HND_SHARE_DEVICE', '0') != '0':
        device_id = 0
        os.environ.setdefault('HND_SHARED_DEVICE', '1')      # ... and the engines treat the card as shared (engine.process_owns_device)
    if has_gpu:
        torch.cuda.set_device(device_id)
    print('| distributed init (rank {}): {}'.format(rank, dist_url), flush=True)
    torch.distributed.init_process_group(backend=backend or ('nccl' if has_gpu else 'gloo'), init_method=dist_url,
                                         world_size=world_size, rank=rank)
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


@torch.no_grad()
def evaluate(model, data_loader, device):
    """COCO evaluation of `model` over `data_loader` (reference :75-113): eval-mode detector forward (RPN -> RoI heads
    -> NMS, + mask / keypoint branches for Mask / Keypoint R-CNN, all on the HIP path), predictions keyed by image_id
    into CocoEvaluator (bbox, + segm / keypoints), gathered over ranks, accumulated and summarised.  Returns the
    evaluator (``coco_eval['bbox'].stats[0]`` = the mAP mimic_runner selects the checkpoint on)."""
    import time
    from . import misc_util
    from .coco_eval_util import CocoEvaluator, get_coco_api_from_dataset, get_iou_types
    cpu_device = torch.device('cpu')
    model.eval()
    metric_logger = misc_util.MetricLogger(delimiter='  ')
    coco = get_coco_api_from_dataset(data_loader.dataset)
    coco_evaluator = CocoEvaluator(coco, get_iou_types(model))
    for image, targets in metric_logger.log_every(data_loader, 100, 'Test:'):
        image = [img.to(device) for img in image]
        torch.cuda.synchronize()
        model_time = time.time()
        outputs = model(image)
        # pasted masks are [n, 1, H, W] floats (0.4 GB per full-size image): the evaluator only ever looks at
        # `masks > 0.5` (coco_eval_util.py:101) and run-length encodes it, so the masks stay on the device and only
        # their run boundaries cross the bus (mask_util.encode_batch)
        outputs = [{k: (v if k == 'masks' else v.to(cpu_device)) for k, v in t.items()} for t in outputs]
        model_time = time.time() - model_time
        res = {int(target['image_id']): output for target, output in zip(targets, outputs)}
        evaluator_time = time.time()
        coco_evaluator.update(res)
        evaluator_time = time.time() - evaluator_time
        metric_logger.update(model_time=model_time, evaluator_time=evaluator_time)
    print('Averaged stats:', metric_logger)
    coco_evaluator.synchronize_between_processes()
    coco_evaluator.accumulate()
    coco_evaluator.summarize()
    return coco_evaluator

"""HND / GHND distillation runner for MI355X (role of the reference's src/mimic_runner.py).

    python -m hnd_ghnd_object_detectors_amd.mimic_runner --config <hnd|ghnd yaml> -distill --synthetic_batches 50

Kept from the reference: the command-line flags (:17-29), the YAML schema, ``--json`` overrides, the checkpoint
dictionary, and the per-batch order of operations of ``distill_model`` (:48-58): device upload, DistillationBox,
``zero_grad`` / ``backward`` / ``step``, the epoch-0 linear warm-up, one logged ``loss.item()`` per iteration.
Also kept: the per-epoch validation (:92-100, COCO bbox mAP through the eval-mode detector on the HIP path) that
selects the checkpoint, and the final test-split evaluation (:109-121, :148-150).  Different, and outside the hot
path: without ``--synthetic_batches N`` (seeded synthetic COCO-shaped batches, used by the tests and benchmarks)
batches come from the COCO-format folders of the yaml through utils/coco_util.py (no pycocotools), decoded images
travel as uint8 to the device; with synthetic batches there is no validation set and the lowest mean training loss
selects the checkpoint (stored under its own key ``best_loss``).
One process per GPU (torchrun / torch.distributed.run); RANK / LOCAL_RANK / WORLD_SIZE come from the environment
exactly as in the reference's ``init_distributed_mode``.
"""
import argparse
import datetime
import os
import time

import torch
from torch import distributed as dist

from .distillation.tool import DistillationBox
from .models import get_model, load_ckpt, save_ckpt
from .myutils.common import file_util, yaml_util
from .myutils.pytorch import func_util, module_util
from .parallel import DistributedStudent
from .utils import data_util, main_util, misc_util

_FLAGS = (  # (flag, kwargs): the reference's CLI first, this build's additions after
    ('--config', dict(required=True, help='yaml file path')),
    ('--device', dict(default='cuda', help='device')),
    ('--json', dict(help='dictionary to overwrite config')),
    ('-distill', dict(action='store_true', help='distill a teacher model')),
    ('-skip_teacher_eval', dict(action='store_true', help='skip teacher model evaluation in testing')),
    ('-transform_bottleneck', dict(action='store_true',
                                   help='use bottleneck transformer (if defined in yaml) in testing')),
    ('--world_size', dict(default=1, type=int, help='number of distributed processes')),
    ('--dist_url', dict(default='env://', help='url used to set up distributed training')),
    ('--synthetic_batches', dict(default=0, type=int, help='batches per epoch of synthetic COCO-shaped data')),
    ('--image_size', dict(default='800x1333', help='HxW of the synthetic images')),
    ('-decoded_input', dict(action='store_true', help='feed decoded uint8 HWC images; ToTensor / flip / normalise / '
                                                      'resize run as one device kernel')),
    ('-host_float_input', dict(action='store_true', help='COCO loaders: convert images to float CHW on the host like '
                                                         'the reference instead of shipping uint8 to the device')),
    ('--num_epochs', dict(default=None, type=int, help='override train.num_epochs')),
    ('--loader_workers', dict(default=8, type=int, help='background threads generating synthetic batches ahead of the '
                                                        'step (the role of the DataLoader workers)')),
    ('-no_prefetch', dict(action='store_true', help='upload every batch synchronously at the top of its step like the '
                                                    'reference, instead of through upload.DevicePrefetcher')),
)


def get_argparser():
    parser = argparse.ArgumentParser(description='Mimic Runner (MI355X HIP path)')
    for flag, kwargs in _FLAGS:
        parser.add_argument(flag, **kwargs)
    return parser


def freeze_modules(student_model, student_model_config):
    """freeze every sub-module listed under ``frozen_modules`` (dotted paths) of the student's YAML section."""
    for path in student_model_config['frozen_modules']:
        module_util.freeze_module_params(module_util.get_module(student_model, path))


def _epoch0_warmup(optimizer, num_batches):
    iters = min(1000, num_batches - 1)          # reference :43-46
    return main_util.warmup_lr_scheduler(optimizer, iters, 1.0 / 1000.0) if iters > 0 else None


class StepClock(object):
    """Device time of the iterations: one HIP event per iteration on the compute stream, read one iteration late so the
    host never waits for it.  ``loss.item()`` no longer drains the stream (hip_loss.StepLoss reads an early pinned copy),
    so a host-side iteration timer measures ENQUEUE time; this one measures what the GPU did."""

    def __init__(self, device):
        self.stream = torch.cuda.current_stream(device)
        self.events, self.ms = [], []

    def tick(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(self.stream)
        self.events.append(ev)
        while len(self.events) >= 3:             # the pair before last has certainly been reached by now or soon
            if not self.events[1].query():
                break
            self.ms.append(self.events[0].elapsed_time(self.events[1]))
            self.events.pop(0)

    def finish(self):
        while len(self.events) >= 2:
            self.events[1].synchronize()
            self.ms.append(self.events[0].elapsed_time(self.events[1]))
            self.events.pop(0)
        return self.ms


LAST_EPOCH = {}            # device-time figures of the newest distill_model epoch (bench.py reads them)
SYNC_CHECK_EVERY = 50      # iterations between hnd_sync_check calls (asynchronous faults / relay time-outs surface here)


def distill_model(distillation_box, data_loader, optimizer, log_freq, device, epoch, student_wrapper=None,
                  prefetch=True):
    """One epoch of optimisation steps; returns the mean loss of the epoch.

    reference src/mimic_runner.py:38-59.  ``prefetch``: batches reach the device through upload.DevicePrefetcher (one
    pinned staging buffer and one asynchronous copy per batch, issued while the previous step computes) instead of the
    reference's synchronous per-tensor ``.to(device)`` at the top of the step; the tensors are the same bits."""
    from . import ops
    from .upload import DevicePrefetcher
    meters = misc_util.MetricLogger(delimiter='  ')
    meters.add_meter('lr', misc_util.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    warmup = _epoch0_warmup(optimizer, len(data_loader)) if epoch == 0 else None
    on_gpu = torch.device(device).type == 'cuda'
    if prefetch and on_gpu and not isinstance(data_loader, DevicePrefetcher):
        data_loader = DevicePrefetcher(data_loader, device)
    clock = StepClock(device) if on_gpu else None
    if clock is not None:
        clock.tick()
    for it, (images, targets) in enumerate(meters.log_every(data_loader, log_freq, 'Epoch: [{}]'.format(epoch))):
        images = [img.to(device, non_blocking=True) for img in images]          # (no-ops behind the prefetcher)
        targets = [{key: value.to(device, non_blocking=True) for key, value in t.items()} for t in targets]
        loss = distillation_box(images, targets)
        optimizer.zero_grad()
        loss.backward()             # DistributedStudent fires the flat gradient all-reduce from inside backward,
        optimizer.step()            # the fused Adam launch waits for it and applies the 1/world mean
        if warmup is not None:
            warmup.step()
        meters.update(loss=loss, lr=optimizer.param_groups[0]['lr'])
        if clock is not None:
            clock.tick()
            if (it + 1) % SYNC_CHECK_EVERY == 0:
                ops.sync_check()
    if clock is not None:
        ops.sync_check()            # epoch end: drain, and raise on anything a kernel reported asynchronously
        ms = clock.finish()
        if ms:
            steady = ms[min(2, len(ms) - 1):]           # the first iterations build plans and allocate
            per_it = sum(steady) / len(steady)
            LAST_EPOCH.update(steady_ms_per_it=per_it, steady_iterations=len(steady), ms=list(ms))
            print('Epoch: [{}] device time {:.2f} ms / it over {} steady iterations ({:.2f} img/s per GPU)'.format(
                epoch, per_it, len(steady), len(images) / per_it * 1e3))
    return meters.loss.global_avg


def distill(teacher_model, student_model, train_loader, val_loader, device, distributed, distill_backbone_only,
            config, args):
    """reference :62-106.  After every epoch the student is validated (COCO bbox mAP through the eval-mode detector:
    RPN -> RoI box head -> NMS, utils/main_util.evaluate) and the checkpoint is kept on the best mAP
    (``best_value``), exactly as the reference does.  Without a validation loader (``--synthetic_batches``: no
    dataset) the lowest mean distillation loss decides instead and is stored under its own key ``best_loss``, so a
    loss is never compared with a reference-produced mAP on resume."""
    train_config = config['train']
    box = DistillationBox(teacher_model, student_model, train_config['criterion'])
    optimizer = func_util.get_optimizer(student_model, train_config['optimizer']['type'],
                                        train_config['optimizer']['params'])
    lr_scheduler = func_util.get_scheduler(optimizer, train_config['scheduler']['type'],
                                           train_config['scheduler']['params'])
    wrapper = student_model if isinstance(student_model, DistributedStudent) else None
    student = student_model if wrapper is None else wrapper.module
    ckpt_file_path = config['student_model']['ckpt']
    best_val_map, best_loss = 0.0, None
    if file_util.check_if_exists(ckpt_file_path):       # resume optimizer + scheduler (weights came via get_model)
        best_val_map, _, _ = load_ckpt(ckpt_file_path, optimizer=optimizer, lr_scheduler=lr_scheduler)
        best_val_map = float(best_val_map or 0.0)
        best_loss = torch.load(ckpt_file_path, map_location='cpu', weights_only=False).get('best_loss')
    use_bottleneck_transformer = args.transform_bottleneck
    started = time.time()
    for epoch in range(args.num_epochs or train_config['num_epochs']):
        if hasattr(train_loader, 'set_epoch'):
            train_loader.set_epoch(epoch)
        teacher_model.eval()
        student_model.train()
        teacher_model.distill_backbone_only = student.distill_backbone_only = distill_backbone_only
        student.backbone.body.layer1.use_bottleneck_transformer = False      # reference :90
        mean_loss = distill_model(box, train_loader, optimizer, train_config['log_freq'], device, epoch, wrapper,
                                  prefetch=not getattr(args, 'no_prefetch', False))
        if val_loader is not None:                                            # reference :92-100
            student.distill_backbone_only = False
            student.backbone.body.layer1.use_bottleneck_transformer = use_bottleneck_transformer
            if wrapper is not None:
                wrapper.sync_buffers()          # rank 0's BatchNorm running statistics everywhere (see parallel.py)
            coco_evaluator = main_util.evaluate(student, val_loader, device=device)
            val_map = float(coco_evaluator.coco_eval['bbox'].stats[0])
            if val_map > best_val_map and misc_util.is_main_process():
                print('Updating ckpt (Best BBox mAP: {:.4f} -> {:.4f})'.format(best_val_map, val_map))
                best_val_map = val_map
                save_ckpt(student, optimizer, lr_scheduler, best_val_map, config, args, ckpt_file_path)
        elif (best_loss is None or mean_loss < best_loss) and misc_util.is_main_process():
            print('Updating ckpt (no validation set; mean distillation loss: {} -> {:.4f})'.format(best_loss, mean_loss))
            best_loss = float(mean_loss)
            save_ckpt(student, optimizer, lr_scheduler, best_val_map, config, args, ckpt_file_path,
                      extra={'best_loss': best_loss})
        lr_scheduler.step()
    if distributed:
        dist.barrier()
    print('Training time {}'.format(datetime.timedelta(seconds=int(time.time() - started))))


def evaluate(teacher_model, student_model, test_loader, device, student_only, use_bottleneck_transformer):
    """reference :109-121: COCO evaluation of the teacher (unless skipped) and the student on the test split"""
    teacher = teacher_model.module if isinstance(teacher_model, DistributedStudent) else teacher_model
    student = student_model.module if isinstance(student_model, DistributedStudent) else student_model
    teacher.distill_backbone_only = student.distill_backbone_only = False
    student.backbone.body.layer1.use_bottleneck_transformer = use_bottleneck_transformer
    if not student_only:
        print('[Teacher model]')
        main_util.evaluate(teacher, test_loader, device=device)
    print('\n[Student model]')
    return main_util.evaluate(student, test_loader, device=device)


def main(args):
    config = yaml_util.load_yaml_file(args.config)
    if args.json is not None:
        main_util.overwrite_config(config, args.json)
    distributed, _ = main_util.init_distributed_mode(args.world_size, args.dist_url)
    main_util.limit_host_threads(int(os.environ.get('LOCAL_WORLD_SIZE') or 1))
    if not torch.cuda.is_available():
        raise RuntimeError('the HIP distillation path needs an MI355X (no CPU fallback exists)')
    device = torch.device(args.device)
    student_config = config['student_model']
    teacher_model = get_model(config['teacher_model'], device)
    module_util.freeze_module_params(teacher_model)
    student_model = get_model(student_config, device)
    freeze_modules(student_model, student_config)
    print('Updatable parameters: {}'.format(module_util.get_updatable_param_names(student_model)))
    batch_size = config['train']['batch_size']
    val_loader = test_loader = None
    if args.synthetic_batches > 0:
        height, width = (int(v) for v in args.image_size.split('x'))
        train_sampler = None
        train_loader = data_util.SyntheticDetectionLoader(args.synthetic_batches, batch_size, height, width,
                                                          student_config['name'], rank=misc_util.get_rank(),
                                                          decoded=args.decoded_input, workers=args.loader_workers,
                                                          pin_memory=not args.no_prefetch)
    else:       # COCO-format folders named by the yaml (reference :128-129); uint8 images unless -host_float_input
        train_sampler, train_loader, val_loader, test_loader = data_util.get_coco_data_loaders(
            config['dataset'], batch_size, distributed, decoded=not args.host_float_input)
        if train_sampler is not None and hasattr(train_sampler, 'set_epoch'):
            train_loader.set_epoch = train_sampler.set_epoch        # distill() advances the shard per epoch
    if distributed:
        student_model = DistributedStudent(student_model)
    if args.distill:
        distill(teacher_model, student_model, train_loader, val_loader, device, distributed,
                student_config['distill_backbone_only'], config, args)
        load_ckpt(student_config['ckpt'], model=student_model.module if distributed else student_model)
    if test_loader is not None:                                     # reference :148-150
        evaluate(teacher_model, student_model, test_loader, device, args.skip_teacher_eval, args.transform_bottleneck)
    else:
        print('no dataset (--synthetic_batches): COCO evaluation skipped; checkpoint: {}'.format(student_config['ckpt']))
    if isinstance(student_model, DistributedStudent):
        student_model.close()               # native RCCL communicator + its stream, before the process group goes


if __name__ == '__main__':
    main(get_argparser().parse_args())

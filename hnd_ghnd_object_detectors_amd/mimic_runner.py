"""HND / GHND distillation runner (mirror of the reference's src/mimic_runner.py CLI and step).

    python -m hnd_ghnd_object_detectors_amd.mimic_runner --config <hnd|ghnd yaml> -distill --synthetic_batches 50

Same flags (reference :17-29), YAML schema, checkpoint dict and step order (:38-59): H2D, DistillationBox,
zero_grad / backward / step, epoch-0 linear warm-up, loss logged per iteration.  Differences, all outside the
hot path: data come from seeded synthetic COCO-shaped batches (--synthetic_batches; COCO + pycocotools are
not available), and the per-epoch COCO mAP evaluation that selects checkpoints (:94-100) is replaced by
"lowest mean training loss" because RPN / RoI heads / NMS are out of this build's scope.
Launch one process per GPU (torchrun / torch.distributed.run); RANK / LOCAL_RANK / WORLD_SIZE are read from
the environment exactly like the reference's init_distributed_mode.
"""
import argparse
import datetime
import time

import torch
from torch import distributed as dist

from .distillation.tool import DistillationBox
from .models import get_model, load_ckpt, save_ckpt
from .myutils.common import file_util, yaml_util
from .myutils.pytorch import func_util, module_util
from .parallel import DistributedStudent
from .utils import data_util, main_util, misc_util


def get_argparser():
    p = argparse.ArgumentParser(description='Mimic Runner (MI355X HIP path)')
    p.add_argument('--config', required=True, help='yaml file path')
    p.add_argument('--device', default='cuda', help='device')
    p.add_argument('--json', help='dictionary to overwrite config')
    p.add_argument('-distill', action='store_true', help='distill a teacher model')
    p.add_argument('-skip_teacher_eval', action='store_true', help='skip teacher model evaluation in testing')
    p.add_argument('-transform_bottleneck', action='store_true',
                   help='use bottleneck transformer (if defined in yaml) in testing')
    p.add_argument('--world_size', default=1, type=int, help='number of distributed processes')
    p.add_argument('--dist_url', default='env://', help='url used to set up distributed training')
    # additions of this build
    p.add_argument('--synthetic_batches', default=0, type=int, help='batches per epoch of synthetic COCO-shaped data')
    p.add_argument('--image_size', default='800x1333', help='HxW of the synthetic images')
    p.add_argument('--num_epochs', default=None, type=int, help='override train.num_epochs')
    return p


def freeze_modules(student_model, student_model_config):
    for student_path in student_model_config['frozen_modules']:
        module_util.freeze_module_params(module_util.get_module(student_model, student_path))


def distill_model(distillation_box, data_loader, optimizer, log_freq, device, epoch, student_wrapper=None):
    """one epoch of optimisation steps (reference :38-59)."""
    metric_logger = misc_util.MetricLogger(delimiter='  ')
    metric_logger.add_meter('lr', misc_util.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    lr_scheduler = None
    if epoch == 0:
        warmup_iters = min(1000, len(data_loader) - 1)
        if warmup_iters > 0:
            lr_scheduler = main_util.warmup_lr_scheduler(optimizer, warmup_iters, 1.0 / 1000.0)
    for images, targets in metric_logger.log_every(data_loader, log_freq, header):
        images = [image.to(device, non_blocking=True) for image in images]
        targets = [{k: v.to(device, non_blocking=True) for k, v in t.items()} for t in targets]
        loss = distillation_box(images, targets)
        optimizer.zero_grad()
        loss.backward()
        if student_wrapper is not None:
            student_wrapper.reduce_gradients()
        optimizer.step()
        if lr_scheduler is not None:
            lr_scheduler.step()
        metric_logger.update(loss=loss)
        metric_logger.update(lr=optimizer.param_groups[0]['lr'])
    return metric_logger.loss.global_avg


def distill(teacher_model, student_model, train_loader, device, distributed, distill_backbone_only, config, args):
    train_config = config['train']
    distillation_box = DistillationBox(teacher_model, student_model, train_config['criterion'])
    ckpt_file_path = config['student_model']['ckpt']
    optim_config = train_config['optimizer']
    optimizer = func_util.get_optimizer(student_model, optim_config['type'], optim_config['params'])
    scheduler_config = train_config['scheduler']
    lr_scheduler = func_util.get_scheduler(optimizer, scheduler_config['type'], scheduler_config['params'])
    wrapper = student_model if isinstance(student_model, DistributedStudent) else None
    if wrapper is not None:
        wrapper.attach_optimizer(optimizer)
    best = None
    if file_util.check_if_exists(ckpt_file_path):
        best, _, _ = load_ckpt(ckpt_file_path, optimizer=optimizer, lr_scheduler=lr_scheduler)
    num_epochs = args.num_epochs or train_config['num_epochs']
    student = wrapper.module if wrapper is not None else student_model
    start_time = time.time()
    for epoch in range(num_epochs):
        if hasattr(train_loader, 'set_epoch'):
            train_loader.set_epoch(epoch)
        teacher_model.eval()
        student_model.train()
        teacher_model.distill_backbone_only = distill_backbone_only
        student.distill_backbone_only = distill_backbone_only
        student.backbone.body.layer1.use_bottleneck_transformer = False
        mean_loss = distill_model(distillation_box, train_loader, optimizer, train_config['log_freq'], device, epoch,
                                  wrapper)
        if (best is None or not isinstance(best, float) or mean_loss < best) and misc_util.is_main_process():
            print('Updating ckpt (mean distillation loss: {} -> {:.4f})'.format(best, mean_loss))
            best = float(mean_loss)
            save_ckpt(student, optimizer, lr_scheduler, best, config, args, ckpt_file_path)
        lr_scheduler.step()
    if distributed:
        dist.barrier()
    print('Training time {}'.format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))


def main(args):
    config = yaml_util.load_yaml_file(args.config)
    if args.json is not None:
        main_util.overwrite_config(config, args.json)
    distributed, device_ids = main_util.init_distributed_mode(args.world_size, args.dist_url)
    if not torch.cuda.is_available():
        raise RuntimeError('the HIP distillation path needs an MI355X (no CPU fallback exists)')
    device = torch.device(args.device)
    teacher_model = get_model(config['teacher_model'], device)
    module_util.freeze_module_params(teacher_model)
    student_model_config = config['student_model']
    student_model = get_model(student_model_config, device)
    freeze_modules(student_model, student_model_config)
    print('Updatable parameters: {}'.format(module_util.get_updatable_param_names(student_model)))
    distill_backbone_only = student_model_config['distill_backbone_only']
    train_config = config['train']
    if args.synthetic_batches <= 0:
        data_util.get_coco_data_loaders(config['dataset'], train_config['batch_size'], distributed)
    h, w = (int(v) for v in args.image_size.split('x'))
    train_loader = data_util.SyntheticDetectionLoader(args.synthetic_batches, train_config['batch_size'], h, w,
                                                      student_model_config['name'], rank=misc_util.get_rank())
    if distributed:
        student_model = DistributedStudent(student_model)
    if args.distill:
        distill(teacher_model, student_model, train_loader, device, distributed, distill_backbone_only, config, args)
    print('COCO evaluation is outside this build; distilled checkpoint: {}'.format(student_model_config['ckpt']))


if __name__ == '__main__':
    main(get_argparser().parse_args())

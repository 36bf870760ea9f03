"""Neural-filter runner on the MI355X HIP path (mirror of the reference's src/ext_runner.py).

    python -m hnd_ghnd_object_detectors_amd.ext_runner --config config/ext/<yaml> -train --synthetic_batches 50

Same CLI (--config / --json / --min_recall / -train / --world_size / --dist_url), the same step -- frozen detector,
trainable Ext4ResNet on the stem output, ``cross_entropy(ext_logits, ext_targets)``, SGD with momentum and weight
decay, epoch-0 warm-up, MultiStepLR per epoch -- and the same checkpoint (classifier state + optimizer + scheduler,
keyed on the best validation ROC-AUC).  Batches come from the seeded synthetic loader (COCO / pycocotools do not
exist here), half of the images labelled positive.
"""
import argparse
import datetime
import math
import sys
import time

import numpy as np
import torch
from torch import distributed as dist
from torch import nn

from .models import get_model, load_ckpt, save_ckpt
from .models.ext.backbone import check_if_valid_target
from .myutils.common import file_util, yaml_util
from .myutils.pytorch import func_util, module_util
from .parallel import DistributedStudent
from .utils import data_util, main_util, misc_util

_FLAGS = (  # the reference's CLI first, this build's additions after
    ('--config', dict(required=True, help='yaml config file')),
    ('--device', dict(default='cuda', help='device')),
    ('--json', dict(help='dictionary to overwrite config')),
    ('--min_recall', dict(type=float, default=0.9, help='minimum recall to decide a threshold')),
    ('-train', dict(action='store_true', help='train a model')),
    ('--world_size', dict(default=1, type=int, help='number of distributed processes')),
    ('--dist_url', dict(default='env://', help='url used to set up distributed training')),
    ('--synthetic_batches', dict(default=0, type=int, help='batches per epoch of synthetic COCO-shaped data')),
    ('--image_size', dict(default='800x1333', help='HxW of the synthetic images')),
    ('--num_epochs', dict(default=None, type=int, help='override train.num_epochs')),
)


def get_argparser():
    parser = argparse.ArgumentParser(description='Neural-filter runner (MI355X HIP path)')
    for flag, kwargs in _FLAGS:
        parser.add_argument(flag, **kwargs)
    return parser


def convert_target2ext_targets(targets, device):
    """image-level labels from the targets AS THE MODEL'S TRANSFORM LEFT THEM (the reference labels after the
    forward pass, :55-56, i.e. on rescaled boxes); the rule walks a few numbers, so they are read back once."""
    host = [{k: t[k].cpu() for k in ('boxes', 'keypoints') if k in t} if len(t) else t for t in targets]
    labels = [1 if check_if_valid_target(t) else 0 for t in host]
    return torch.tensor(labels, dtype=torch.int64).to(device)


def _to_device(images, targets, device):
    images = [img.to(device, non_blocking=True) for img in images]
    targets = [{k: v.to(device, non_blocking=True) for k, v in t.items()} for t in targets]
    return images, targets


def train_model(model, optimizer, data_loader, device, epoch, log_freq, wrapper=None):
    """one epoch of filter training (reference :39-76)."""
    model.train()
    meters = misc_util.MetricLogger(delimiter='  ')
    meters.add_meter('lr', misc_util.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    warmup = None
    if epoch == 0 and len(data_loader) > 1:
        warmup = main_util.warmup_lr_scheduler(optimizer, min(1000, len(data_loader) - 1), 1.0 / 1000.0)
    for images, targets in meters.log_every(data_loader, log_freq, 'Epoch: [{}]'.format(epoch)):
        images, targets = _to_device(images, targets, device)
        ext_logits = model(images, targets)
        ext_targets = convert_target2ext_targets(targets, device)
        loss = nn.functional.cross_entropy(ext_logits, ext_targets)
        loss_value = float(misc_util.reduce_dict({'loss_ext_classifier': loss.detach()})['loss_ext_classifier'])
        if not math.isfinite(loss_value):
            print('Loss is {}, stopping training'.format(loss_value))
            sys.exit(1)
        optimizer.zero_grad()
        loss.backward()
        if wrapper is not None:
            wrapper.reduce_gradients()
        optimizer.step()
        if warmup is not None:
            warmup.step()
        meters.update(loss=loss_value, loss_ext_classifier=loss_value, lr=optimizer.param_groups[0]['lr'])
    return meters.loss.global_avg


def evaluate(model, data_loader, device, min_recall, split_name='Validation'):
    """accuracy / recall / specificity / ROC-AUC of the filter (reference :79-123); the decision-threshold table
    for ``min_recall`` is printed for the Test split."""
    from sklearn import metrics
    model.eval()
    probs, labels = [], []
    with torch.no_grad():
        for images, targets in data_loader:
            images, targets = _to_device(images, targets, device)
            ext_probs = model(images, targets)
            ext_targets = convert_target2ext_targets(targets, device)
            probs.append(ext_probs[:, 1].cpu().numpy())
            labels.append(ext_targets.cpu().numpy())
    probs, labels = np.concatenate(probs), np.concatenate(labels)
    preds = (probs > 0.5).astype(np.int64)            # argmax over two softmax outputs
    num_samples, pos_count = len(labels), int(labels.sum())
    correct, pos_correct = int((preds == labels).sum()), int(preds[labels == 1].sum())
    roc_auc = metrics.roc_auc_score(labels, probs) if 0 < pos_count < num_samples else float('nan')
    print('[{}]'.format(split_name))
    print('\tAccuracy: {:.4f} ({} / {})'.format(correct / num_samples, correct, num_samples))
    print('\tRecall: {:.4f} ({} / {})'.format(pos_correct / max(pos_count, 1), pos_correct, pos_count))
    print('\tSpecificity: {:.4f} ({} / {})'.format((correct - pos_correct) / max(num_samples - pos_count, 1),
                                                   correct - pos_correct, num_samples - pos_count))
    print('\tROC-AUC: {:.4f}'.format(roc_auc))
    if split_name == 'Test' and 0 < pos_count < num_samples:
        import pandas as pd
        fprs, tprs, thrs = metrics.roc_curve(labels, probs, pos_label=1)
        idx = np.searchsorted(tprs, min_recall)
        table = pd.DataFrame(np.array([thrs[idx:], tprs[idx:], fprs[idx:]]).T,
                             columns=['Threshold', 'TPR (Recall)', 'FPR'])
        with pd.option_context('display.max_rows', None, 'display.max_columns', None):
            print(table)
    return roc_auc


def train(model, ext_classifier, train_loader, val_loader, device, distributed, config, args, ckpt_file_path):
    train_config = config['train']
    optimizer = func_util.get_optimizer(ext_classifier, train_config['optimizer']['type'],
                                        train_config['optimizer']['params'])
    lr_scheduler = func_util.get_scheduler(optimizer, train_config['scheduler']['type'],
                                           train_config['scheduler']['params'])
    wrapper = model if isinstance(model, DistributedStudent) else None
    if wrapper is not None:
        wrapper.attach_optimizer(optimizer)
    best = 0.0
    if file_util.check_if_exists(ckpt_file_path):
        best, _, _ = load_ckpt(ckpt_file_path, model=ext_classifier, optimizer=optimizer, lr_scheduler=lr_scheduler)
    started = time.time()
    for epoch in range(args.num_epochs or train_config['num_epochs']):
        if hasattr(train_loader, 'set_epoch'):
            train_loader.set_epoch(epoch)
        train_model(model, optimizer, train_loader, device, epoch, train_config['log_freq'], wrapper)
        lr_scheduler.step()
        val_roc_auc = evaluate(model, val_loader, device, min_recall=args.min_recall, split_name='Validation')
        if (val_roc_auc > best or not file_util.check_if_exists(ckpt_file_path)) and misc_util.is_main_process():
            print('Updating ckpt (Best ROC-AUC: {:.4f} -> {:.4f})'.format(best, val_roc_auc))
            best = max(best, float(val_roc_auc)) if math.isfinite(val_roc_auc) else best
            save_ckpt(ext_classifier, optimizer, lr_scheduler, best, config, args, ckpt_file_path)
    if distributed:
        dist.barrier()
    print('Training time {}'.format(datetime.timedelta(seconds=int(time.time() - started))))


def main(args):
    distributed, _ = main_util.init_distributed_mode(args.world_size, args.dist_url)
    config = yaml_util.load_yaml_file(args.config)
    if args.json is not None:
        main_util.overwrite_config(config, args.json)
    if not torch.cuda.is_available():
        raise RuntimeError('the HIP path needs an MI355X (no CPU fallback exists)')
    device = torch.device(args.device)
    print(args)
    train_config, model_config = config['train'], config['model']
    if args.synthetic_batches > 0:
        height, width = (int(v) for v in args.image_size.split('x'))
        rank = misc_util.get_rank()

        def loader(batch_size, seed):
            return data_util.SyntheticDetectionLoader(args.synthetic_batches, batch_size, height, width,
                                                      model_config['name'], seed=seed, rank=rank, positive_every=2)
        train_loader = loader(train_config['batch_size'], 1234)
        val_loader = test_loader = loader(config['test']['batch_size'], 4321)
    else:
        train_sampler, train_loader, val_loader, test_loader = data_util.get_coco_data_loaders(
            config['dataset'], train_config['batch_size'], distributed)
        if hasattr(train_sampler, 'set_epoch'):
            train_loader.set_epoch = train_sampler.set_epoch
    print('Creating model')
    model = get_model(model_config, device, strict=False)
    module_util.freeze_module_params(model)
    ext_classifier = model.get_ext_classifier()
    module_util.unfreeze_module_params(ext_classifier)
    print('Updatable parameters: {}'.format(module_util.get_updatable_param_names(model)))
    model.train_ext()
    if distributed:
        model = DistributedStudent(model)
    if args.train:
        print('Start training')
        ckpt_file_path = model_config['backbone']['ext_config']['ckpt']
        train(model, ext_classifier, train_loader, val_loader, device, distributed, config, args, ckpt_file_path)
        load_ckpt(ckpt_file_path, model=ext_classifier)
    evaluate(model, test_loader, device=device, min_recall=args.min_recall, split_name='Test')


if __name__ == '__main__':
    main(get_argparser().parse_args())

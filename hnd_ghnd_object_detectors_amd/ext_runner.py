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


from .models import get_model, load_ckpt, save_ckpt
from .models.ext.backbone import check_if_valid_target
from .models.ext.classifier import cross_entropy
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
    return torch.tensor([int(check_if_valid_target(t)) for t in host], dtype=torch.int64).to(device)


def _upload(images, targets, device):
    return ([img.to(device, non_blocking=True) for img in images],
            [{k: v.to(device, non_blocking=True) for k, v in t.items()} for t in targets])


def train_model(model, optimizer, data_loader, device, epoch, log_freq, wrapper=None):
    """one epoch of filter training (reference :39-76): CE on the classifier logits, warm-up in epoch 0."""
    model.train()
    meters = misc_util.MetricLogger(delimiter='  ')
    meters.add_meter('lr', misc_util.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    num_batches = len(data_loader)
    warmup = (main_util.warmup_lr_scheduler(optimizer, min(1000, num_batches - 1), 1.0 / 1000.0)
              if epoch == 0 and num_batches > 1 else None)
    if torch.device(device).type == 'cuda':     # one pinned staging buffer + one async copy per batch, a step ahead
        from .upload import DevicePrefetcher
        data_loader = DevicePrefetcher(data_loader, device)
    for images, targets in meters.log_every(data_loader, log_freq, 'Epoch: [{}]'.format(epoch)):
        images, targets = _upload(images, targets, device)      # (no-ops behind the prefetcher)
        ext_logits = model(images, targets)
        ext_targets = convert_target2ext_targets(targets, device)
        loss = cross_entropy(ext_logits, ext_targets)          # one launch: loss + dlogits (hnd_softmax_ce_rows_fwd_bwd)
        logged = float(misc_util.reduce_dict({'loss_ext_classifier': loss.detach()})['loss_ext_classifier'])
        if not math.isfinite(logged):
            print('Loss is {}, stopping training'.format(logged))
            sys.exit(1)
        optimizer.zero_grad()
        loss.backward()             # DistributedStudent fires the flat gradient all-reduce from inside backward,
        optimizer.step()            # the fused SGD launch waits for it and applies the 1/world mean
        if warmup is not None:
            warmup.step()
        meters.update(loss=logged, loss_ext_classifier=logged, lr=optimizer.param_groups[0]['lr'])
    if torch.device(device).type == 'cuda':
        from . import ops
        ops.sync_check()            # epoch end: drain, and raise on anything a kernel reported asynchronously
    return meters.loss.global_avg


def _collect_scores(model, data_loader, device):
    """P(positive) and label of every sample of the loader, as numpy vectors"""
    model.eval()
    scores, labels = [], []
    with torch.no_grad():
        for images, targets in data_loader:
            images, targets = _upload(images, targets, device)
            probabilities = model(images, targets)
            labels.append(convert_target2ext_targets(targets, device).cpu().numpy())
            scores.append(probabilities[:, 1].cpu().numpy())
    return np.concatenate(scores), np.concatenate(labels)


def evaluate(model, data_loader, device, min_recall, split_name='Validation'):
    """accuracy / recall / specificity / ROC-AUC of the filter (reference :79-123); for the Test split the table of
    decision thresholds that keep recall >= ``min_recall`` is printed as well."""
    from sklearn import metrics
    scores, labels = _collect_scores(model, data_loader, device)
    predicted = scores > 0.5                            # argmax over the two softmax outputs
    total, positives = len(labels), int(labels.sum())
    hits = int((predicted == (labels == 1)).sum())
    true_pos = int(predicted[labels == 1].sum())
    true_neg = hits - true_pos
    both_classes = 0 < positives < total
    roc_auc = metrics.roc_auc_score(labels, scores) if both_classes else float('nan')
    print('[{}]'.format(split_name))
    for title, num, den in (('Accuracy', hits, total), ('Recall', true_pos, positives),
                            ('Specificity', true_neg, total - positives)):
        print('\t{}: {:.4f} ({} / {})'.format(title, num / max(den, 1), num, den))
    print('\tROC-AUC: {:.4f}'.format(roc_auc))
    if split_name == 'Test' and both_classes:
        import pandas as pd
        fprs, tprs, thresholds = metrics.roc_curve(labels, scores, pos_label=1)
        first = int(np.searchsorted(tprs, min_recall))
        table = pd.DataFrame({'Threshold': thresholds[first:], 'TPR (Recall)': tprs[first:], 'FPR': fprs[first:]})
        with pd.option_context('display.max_rows', None, 'display.max_columns', None):
            print(table)
    return roc_auc


def train(model, ext_classifier, train_loader, val_loader, device, distributed, config, args, ckpt_file_path):
    """epochs of train_model + validation; the classifier checkpoint follows the best validation ROC-AUC (:126-160)"""
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
        score = evaluate(model, val_loader, device, min_recall=args.min_recall, split_name='Validation')
        first_ckpt = not file_util.check_if_exists(ckpt_file_path)
        if (score > best or first_ckpt) and misc_util.is_main_process():
            print('Updating ckpt (Best ROC-AUC: {:.4f} -> {:.4f})'.format(best, score))
            best = max(best, float(score)) if math.isfinite(score) else best
            save_ckpt(ext_classifier, optimizer, lr_scheduler, best, config, args, ckpt_file_path)
    if distributed:
        dist.barrier()
    print('Training time {}'.format(datetime.timedelta(seconds=int(time.time() - started))))


def _loaders(args, config, distributed):
    """(train, val, test) loaders: seeded synthetic batches when asked for, else the yaml's COCO-format folders"""
    batch_train, batch_eval = config['train']['batch_size'], config['test']['batch_size']
    if args.synthetic_batches > 0:
        height, width = (int(v) for v in args.image_size.split('x'))

        def synthetic(batch_size, seed):
            return data_util.SyntheticDetectionLoader(args.synthetic_batches, batch_size, height, width,
                                                      config['model']['name'], seed=seed, rank=misc_util.get_rank(),
                                                      positive_every=2, workers=4, pin_memory=True)
        held_out = synthetic(batch_eval, 4321)
        return synthetic(batch_train, 1234), held_out, held_out
    sampler, train_loader, val_loader, test_loader = data_util.get_coco_data_loaders(config['dataset'], batch_train,
                                                                                     distributed)
    if hasattr(sampler, 'set_epoch'):
        train_loader.set_epoch = sampler.set_epoch      # train() advances the shard per epoch
    return train_loader, val_loader, test_loader


def _filter_model(model_config, device):
    """the frozen detector with only its neural filter trainable, switched to filter mode (reference :192-199)"""
    model = get_model(model_config, device, strict=False)
    module_util.freeze_module_params(model)
    ext_classifier = model.get_ext_classifier()
    if ext_classifier is None:
        raise ValueError('the model has no neural filter: set backbone.ext_config in the yaml')
    module_util.unfreeze_module_params(ext_classifier)
    print('Updatable parameters: {}'.format(module_util.get_updatable_param_names(model)))
    model.train_ext()
    return model, ext_classifier


def main(args):
    distributed, _ = main_util.init_distributed_mode(args.world_size, args.dist_url)
    main_util.limit_host_threads()
    config = yaml_util.load_yaml_file(args.config)
    if args.json is not None:
        main_util.overwrite_config(config, args.json)
    if not torch.cuda.is_available():
        raise RuntimeError('the HIP path needs an MI355X (no CPU fallback exists)')
    device = torch.device(args.device)
    print(args)
    train_loader, val_loader, test_loader = _loaders(args, config, distributed)
    print('Creating model')
    model, ext_classifier = _filter_model(config['model'], device)
    if distributed:
        model = DistributedStudent(model)
    if args.train:
        print('Start training')
        ckpt_file_path = config['model']['backbone']['ext_config']['ckpt']
        train(model, ext_classifier, train_loader, val_loader, device, distributed, config, args, ckpt_file_path)
        load_ckpt(ckpt_file_path, model=ext_classifier)         # evaluate the best classifier, not the last
    evaluate(model, test_loader, device=device, min_recall=args.min_recall, split_name='Test')
    if isinstance(model, DistributedStudent):
        model.close()


if __name__ == '__main__':
    main(get_argparser().parse_args())

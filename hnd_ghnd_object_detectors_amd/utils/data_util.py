"""Input feeding for the runners.

``get_coco_data_loaders`` mirrors the reference (src/utils/data_util.py:18-48: COCO datasets, aspect-ratio grouped
batches, DistributedSampler) over utils/coco_util.py, which reads COCO-format folders without pycocotools.
COCO itself is not in the image, so benchmarks and tests are fed by seeded synthetic COCO-shaped batches
(SURVEY.md section 8d): uniform [0,1) images 3xHxW, one box per image (+ a mask / 17 keypoints for Mask /
Keypoint R-CNN).

``decoded=True`` feeds what a JPEG decoder hands over instead -- uint8 [H, W, 3] -- through this build's
``ToTensor`` / ``RandomHorizontalFlip`` (structure/transformer.py; the reference's training pipeline,
src/utils/data_util.py:9-15 ``get_coco_dataset``: ToTensor then RandomHorizontalFlip(0.5)), so the float conversion
and the flip run inside the device transform kernel.
"""
import os
import random
import threading

import torch

from ..structure.transformer import Compose, RandomHorizontalFlip, ToTensor


def get_transform(train=False, decoded=True):
    """the transform chain of reference src/utils/data_util.py:10-12; ``decoded`` keeps the image uint8 for the
    fused device kernel (False: float CHW tensor on the host, exactly like the reference)"""
    transforms = [ToTensor(decoded)]
    if train:
        transforms.append(RandomHorizontalFlip(0.5))
    return Compose(transforms)


class SyntheticDetectionLoader(object):
    """len()-able iterable of (images, targets) tuples, sharded by rank through the seed.

    Batch k of epoch e is a function of (seed, rank, e, k) alone, so ``workers`` background threads may generate
    batches ahead of the consumer (torch's generators release the GIL; 16 images of 3x800x1333 take ~0.26 s of one
    core, three steps' worth) without changing a single value -- the role the reference's DataLoader workers play
    (src/utils/data_util.py:40-41).  Flip decisions come from a generator of the loader's own, so the Python
    ``random`` stream that DistillationBox draws Keypoint R-CNN sizes from (seeded here, at iter()) is not shared
    with a feeder thread."""

    def __init__(self, num_batches, batch_size, height=800, width=1333, model_name='faster_rcnn', seed=1234, rank=0,
                 device='cpu', decoded=False, train=True, positive_every=0, workers=0, pin_memory=False, pool_batches=0):
        self.num_batches, self.batch_size, self.h, self.w = num_batches, batch_size, height, width
        self.model_name, self.seed, self.rank, self.device = model_name, seed, rank, device
        self.decoded, self.transform = decoded, get_transform(train)
        self.positive_every = positive_every       # neural filter: every k-th person has 17 visible keypoints
        self.epoch = 0
        self.workers = workers
        # like DataLoader(pin_memory=True): images are generated straight into pinned host memory, so the uploader
        # (upload.DevicePrefetcher) sends them to the device without a staging copy
        self.pin_memory = bool(pin_memory) and torch.cuda.is_available()
        # pool_batches = P > 0: batches 0 .. P-1 of the epoch are generated ONCE (the same values as without a pool) and
        # batch k hands out pool[k % P] -- for hosts whose CPU share per rank cannot generate 205 MB of random numbers per
        # step (8 ranks on a 16-CPU container: bench.py's `upload.mode`).  What the reference's step pays is the upload
        # of the batch (src/mimic_runner.py:49-50), which still happens every step; random-number generation is its
        # DataLoader workers' cost, not the step's.
        self.pool_batches = int(pool_batches)
        self._pool, self._pool_seed, self._slot_locks, self._pool_lock = None, None, {}, threading.Lock()
        # host cost of making batches: thread CPU seconds spent in raw_images / batches made (bench.py reports it)
        self.gen_cpu_s, self.gen_batches = 0.0, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_batches

    def make_targets(self):
        h, w = self.h, self.w
        out = []
        self._count = getattr(self, '_count', 0) + 1
        for i in range(self.batch_size):
            t = {'boxes': torch.tensor([[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]], dtype=torch.float32),
                 'labels': torch.tensor([1], dtype=torch.int64)}
            if self.model_name == 'mask_rcnn':
                t['masks'] = torch.zeros(1, h, w, dtype=torch.uint8)
            if self.model_name == 'keypoint_rcnn':
                t['keypoints'] = torch.zeros(1, 17, 3, dtype=torch.float32)
                if self.positive_every and (i + self._count) % self.positive_every == 0:
                    t['keypoints'][..., 2] = 1
            out.append(t)
        return out

    def _epoch_seed(self):
        return self.seed + self.rank + 7919 * self.epoch

    def raw_images(self, k, epoch_seed=None):
        """the images of batch k of the current epoch: float CHW in [0, 1), or uint8 HWC with ``decoded``"""
        import time
        es = self._epoch_seed() if epoch_seed is None else epoch_seed
        if self.pool_batches > 0:
            j = k % self.pool_batches
            with self._pool_lock:               # (feeder threads: ONE of them makes a slot, the others wait for it)
                if self._pool_seed != es:
                    self._pool, self._pool_seed, self._slot_locks = {}, es, {}
                slot_lock = self._slot_locks.setdefault(j, threading.Lock())
            with slot_lock:
                if j not in self._pool:
                    self._pool[j] = self._generate(j, es)
                return self._pool[j]
        t0 = time.thread_time()
        out = self._generate(k, es)
        self.gen_cpu_s += time.thread_time() - t0
        self.gen_batches += 1
        return out

    def _generate(self, k, es):
        g = torch.Generator().manual_seed(es + 1000003 * k)
        pin = self.pin_memory
        if not self.decoded:
            return [torch.rand(3, self.h, self.w, generator=g, out=torch.empty(3, self.h, self.w, pin_memory=pin))
                    for _ in range(self.batch_size)]
        return [torch.randint(0, 256, (self.h, self.w, 3), generator=g, dtype=torch.uint8,
                              out=torch.empty(self.h, self.w, 3, dtype=torch.uint8, pin_memory=pin))
                for _ in range(self.batch_size)]

    def _finish(self, raw, flip_rng):
        if not self.decoded:
            return tuple(raw), tuple(self.make_targets())
        for t in self.transform.transforms:
            if isinstance(t, RandomHorizontalFlip):
                t.rng = flip_rng
        pairs = [self.transform(im, t) for im, t in zip(raw, self.make_targets())]
        return tuple(p[0] for p in pairs), tuple(p[1] for p in pairs)

    def __iter__(self):
        es = self._epoch_seed()
        random.seed(es)                 # (the stream DistillationBox draws Keypoint sizes from)
        return self._batches(es, random.Random(es))

    def _batches(self, es, flip_rng):
        if self.workers <= 0:
            for k in range(self.num_batches):
                yield self._finish(self.raw_images(k, es), flip_rng)
            return
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(self.workers, thread_name_prefix='synthetic-loader')
        try:
            ahead = [pool.submit(self.raw_images, k, es) for k in range(min(self.workers + 1, self.num_batches))]
            for k in range(self.num_batches):
                raw = ahead.pop(0).result()
                nxt = k + self.workers + 1
                if nxt < self.num_batches:
                    ahead.append(pool.submit(self.raw_images, nxt, es))
                yield self._finish(raw, flip_rng)
        finally:
            pool.shutdown(wait=False, cancel_futures=True)


def get_coco_dataset(split_dict, is_train, decoded=True):
    from .coco_util import get_coco
    return get_coco(img_dir_path=split_dict['images'], ann_file_path=split_dict['annotations'],
                    transforms=get_transform(is_train, decoded),
                    remove_non_annotated_imgs=split_dict['remove_non_annotated_imgs'],
                    jpeg_quality=split_dict['jpeg_quality'])


def get_coco_data_loaders(dataset_config, batch_size, distributed, decoded=True):
    """(train_sampler, train_loader, val_loader, test_loader) over COCO-format folders (reference :18-48):
    aspect-ratio grouped training batches, batch-1 validation / test, DistributedSampler shards per rank."""
    from torch.utils.data import DataLoader, RandomSampler, SequentialSampler, BatchSampler
    from torch.utils.data.distributed import DistributedSampler
    from ..structure.sampler import GroupedBatchSampler, create_aspect_ratio_groups
    from . import misc_util
    splits = dataset_config['splits']
    for split in splits.values():
        if not os.path.isfile(os.path.expanduser(split['annotations'])):
            raise FileNotFoundError('COCO annotation file `{}` is not found: point the yaml at a COCO-format dataset or '
                                    'run with --synthetic_batches N'.format(split['annotations']))
    train_dataset = get_coco_dataset(splits['train'], True, decoded)
    val_dataset = get_coco_dataset(splits['val'], False, decoded)
    test_dataset = get_coco_dataset(splits['test'], False, decoded)
    print('Creating data loaders')
    if distributed:
        train_sampler, val_sampler, test_sampler = (DistributedSampler(d) for d in
                                                    (train_dataset, val_dataset, test_dataset))
    else:
        train_sampler = RandomSampler(train_dataset)
        val_sampler, test_sampler = SequentialSampler(val_dataset), SequentialSampler(test_dataset)
    factor = dataset_config['aspect_ratio_group_factor']
    if factor >= 0:
        group_ids = create_aspect_ratio_groups(train_dataset, k=factor)
        train_batch_sampler = GroupedBatchSampler(train_sampler, group_ids, batch_size)
    else:
        train_batch_sampler = BatchSampler(train_sampler, batch_size, drop_last=True)
    workers = dataset_config['num_workers']
    train_loader = DataLoader(train_dataset, batch_sampler=train_batch_sampler, num_workers=workers,
                              collate_fn=misc_util.collate_fn, pin_memory=torch.cuda.is_available())
    val_loader = DataLoader(val_dataset, batch_size=1, sampler=val_sampler, num_workers=workers,
                            collate_fn=misc_util.collate_fn)
    test_loader = DataLoader(test_dataset, batch_size=1, sampler=test_sampler, num_workers=workers,
                             collate_fn=misc_util.collate_fn)
    return train_sampler, train_loader, val_loader, test_loader

"""Input feeding for the distillation runner.

The reference builds COCO loaders (src/utils/data_util.py:18-48: pycocotools datasets, aspect-ratio grouped
batches, DistributedSampler).  Dataset I/O is outside this build's hot path and neither COCO nor pycocotools
exist in the image, so the runner is fed by seeded synthetic COCO-shaped batches:
uniform [0,1) images 3xHxW, one box per image (+ a mask / 17 keypoints for Mask / Keypoint R-CNN).

``decoded=True`` feeds what a JPEG decoder hands over instead -- uint8 [H, W, 3] -- through this build's
``ToTensor`` / ``RandomHorizontalFlip`` (structure/transformer.py; the reference's training pipeline,
src/utils/data_util.py:9-15 ``get_coco_dataset``: ToTensor then RandomHorizontalFlip(0.5)), so the float conversion
and the flip run inside the device transform kernel.
"""
import random

import torch

from ..structure.transformer import Compose, RandomHorizontalFlip, ToTensor


def get_transform(train=False):
    """the transform chain of reference src/utils/data_util.py:10-12"""
    transforms = [ToTensor()]
    if train:
        transforms.append(RandomHorizontalFlip(0.5))
    return Compose(transforms)


class SyntheticDetectionLoader(object):
    """len()-able iterable of (images, targets) tuples, sharded by rank through the seed."""

    def __init__(self, num_batches, batch_size, height=800, width=1333, model_name='faster_rcnn', seed=1234, rank=0,
                 device='cpu', decoded=False, train=True, positive_every=0):
        self.num_batches, self.batch_size, self.h, self.w = num_batches, batch_size, height, width
        self.model_name, self.seed, self.rank, self.device = model_name, seed, rank, device
        self.decoded, self.transform = decoded, get_transform(train)
        self.positive_every = positive_every       # neural filter: every k-th person has 17 visible keypoints
        self.epoch = 0

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

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.rank + 7919 * self.epoch)
        random.seed(self.seed + self.rank + 7919 * self.epoch)
        for _ in range(self.num_batches):
            if not self.decoded:
                images = [torch.rand(3, self.h, self.w, generator=g) for _ in range(self.batch_size)]
                yield tuple(images), tuple(self.make_targets())
                continue
            raw = [torch.randint(0, 256, (self.h, self.w, 3), generator=g, dtype=torch.uint8)
                   for _ in range(self.batch_size)]
            pairs = [self.transform(im, t) for im, t in zip(raw, self.make_targets())]
            yield tuple(p[0] for p in pairs), tuple(p[1] for p in pairs)


def get_coco_data_loaders(dataset_config, batch_size, distributed):
    raise NotImplementedError('COCO dataset loading (pycocotools) is outside the distillation hot path of this '
                              'build; run mimic_runner with --synthetic_batches N (SURVEY.md section 8d)')

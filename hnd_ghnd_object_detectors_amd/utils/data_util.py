"""Input feeding for the distillation runner.

The reference builds COCO loaders (src/utils/data_util.py:18-48: pycocotools datasets, aspect-ratio grouped
batches, DistributedSampler).  Dataset I/O is outside this build's hot path and neither COCO nor pycocotools
exist in the image, so the runner is fed by seeded synthetic COCO-shaped batches:
uniform [0,1) images 3xHxW, one box per image (+ a mask / 17 keypoints for Mask / Keypoint R-CNN).
"""
import torch


class SyntheticDetectionLoader(object):
    """len()-able iterable of (images, targets) tuples, sharded by rank through the seed."""

    def __init__(self, num_batches, batch_size, height=800, width=1333, model_name='faster_rcnn', seed=1234, rank=0,
                 device='cpu'):
        self.num_batches, self.batch_size, self.h, self.w = num_batches, batch_size, height, width
        self.model_name, self.seed, self.rank, self.device = model_name, seed, rank, device
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_batches

    def make_targets(self):
        h, w = self.h, self.w
        out = []
        for _ in range(self.batch_size):
            t = {'boxes': torch.tensor([[0.125 * w, 0.125 * h, 0.5 * w, 0.5 * h]], dtype=torch.float32),
                 'labels': torch.tensor([1], dtype=torch.int64)}
            if self.model_name == 'mask_rcnn':
                t['masks'] = torch.zeros(1, h, w, dtype=torch.uint8)
            if self.model_name == 'keypoint_rcnn':
                t['keypoints'] = torch.zeros(1, 17, 3, dtype=torch.float32)
            out.append(t)
        return out

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.rank + 7919 * self.epoch)
        for _ in range(self.num_batches):
            images = [torch.rand(3, self.h, self.w, generator=g) for _ in range(self.batch_size)]
            yield tuple(images), tuple(self.make_targets())


def get_coco_data_loaders(dataset_config, batch_size, distributed):
    raise NotImplementedError('COCO dataset loading (pycocotools) is outside the distillation hot path of this '
                              'build; run mimic_runner with --synthetic_batches N (SURVEY.md section 8d)')

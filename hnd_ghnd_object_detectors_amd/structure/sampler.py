"""Aspect-ratio grouped batching (role of the reference's src/structure/sampler.py).

Batches only hold images of one aspect-ratio bin, so the zero padding the model's transform adds to reach a common
size stays small (on this build: fewer wasted pixels in every implicit-GEMM tile).  Same contract as the reference:
``create_aspect_ratio_groups(dataset, k)`` bins width / height into 2k+1 log-spaced edges (:178-187) and
``GroupedBatchSampler`` walks the base sampler in order, emits a batch whenever a bin has collected ``batch_size``
indices and, once the sampler is exhausted, tops up the fullest leftovers with earlier samples of the same bin so
that ``len()`` == ``len(sampler) // batch_size`` holds exactly (:41-75).
"""
import bisect
from collections import OrderedDict

import numpy as np
from torch.utils.data.sampler import BatchSampler, Sampler


class GroupedBatchSampler(BatchSampler):
    def __init__(self, sampler, group_ids, batch_size):
        if not isinstance(sampler, Sampler):
            raise ValueError('sampler should be an instance of torch.utils.data.Sampler, but got sampler={}'
                             .format(sampler))
        self.sampler, self.group_ids, self.batch_size = sampler, group_ids, batch_size

    def __len__(self):
        return len(self.sampler) // self.batch_size

    def __iter__(self):
        pending, seen = OrderedDict(), {}
        emitted = 0
        for idx in self.sampler:
            gid = self.group_ids[idx]
            seen.setdefault(gid, []).append(idx)
            bucket = pending.setdefault(gid, [])
            bucket.append(idx)
            if len(bucket) == self.batch_size:
                yield pending.pop(gid)
                emitted += 1
        missing = len(self) - emitted
        # leftovers, fullest first (stable for ties, like sorted(..., reverse=True) on insertion order)
        for gid, bucket in sorted(pending.items(), key=lambda kv: len(kv[1]), reverse=True):
            if missing <= 0:
                break
            bucket = bucket + seen[gid][:self.batch_size - len(bucket)]
            assert len(bucket) == self.batch_size
            yield bucket
            missing -= 1
        assert missing <= 0


def compute_aspect_ratios(dataset, indices=None):
    """width / height per sample, from the annotation index when the dataset offers one (no image decoding)."""
    if hasattr(dataset, 'indices') and hasattr(dataset, 'dataset'):          # torch.utils.data.Subset
        chosen = dataset.indices if indices is None else [dataset.indices[i] for i in indices]
        return compute_aspect_ratios(dataset.dataset, list(chosen))
    indices = range(len(dataset)) if indices is None else indices
    if hasattr(dataset, 'get_height_and_width'):
        sizes = [dataset.get_height_and_width(i) for i in indices]
        return [float(w) / float(h) for h, w in sizes]
    ratios = []
    for i in indices:                                                       # slow path: decode
        img, _ = dataset[i]
        h, w = img.shape[-2:]
        ratios.append(float(w) / float(h))
    return ratios


def create_aspect_ratio_groups(dataset, k=0):
    ratios = compute_aspect_ratios(dataset)
    bins = sorted((2 ** np.linspace(-1, 1, 2 * k + 1)).tolist()) if k > 0 else [1.0]
    groups = [bisect.bisect_right(bins, r) for r in ratios]
    counts = np.unique(groups, return_counts=True)[1]
    print('Using {} as bins for aspect ratio quantization'.format([0] + bins + [np.inf]))
    print('Count of instances per bin: {}'.format(counts))
    return groups

"""Logging meters and rank helpers (mirror of the reference's src/utils/misc_util.py:10-69,142-262)."""
import datetime
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist


class SmoothedValue(object):
    """window median / global average of a scalar series."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or '{median:.4f} ({global_avg:.4f})'

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        dev = 'cuda' if torch.cuda.is_available() and dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return sorted(self.deque)[(len(self.deque) - 1) // 2]       # torch.median convention (lower middle)

    @property
    def avg(self):
        return sum(self.deque) / len(self.deque)

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max,
                               value=self.value)


class MetricLogger(object):
    def __init__(self, delimiter='\t'):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()            # the reference's one host sync per step (misc_util.py:149-150)
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError('`{}` object has no attribute `{}`'.format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join('{}: {}'.format(name, str(meter)) for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        header = header or ''
        start_time = end = time.time()
        iter_time, data_time = SmoothedValue(fmt='{avg:.4f}'), SmoothedValue(fmt='{avg:.4f}')
        total = len(iterable)
        width = len(str(total))
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == total - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (total - i))))
                parts = [header, '[{0:{w}d}/{1}]'.format(i, total, w=width), 'eta: ' + eta, str(self),
                         'time: ' + str(iter_time), 'data: ' + str(data_time)]
                if torch.cuda.is_available():
                    parts.append('max mem: {:.0f}'.format(torch.cuda.max_memory_allocated() / (1024.0 * 1024.0)))
                print(self.delimiter.join(parts))
            end = time.time()
        total_time = time.time() - start_time
        print('{} Total time: {} ({:.4f} s / it)'.format(header, str(datetime.timedelta(seconds=int(total_time))),
                                                         total_time / max(total, 1)))


def collate_fn(batch):
    return tuple(zip(*batch))


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def reduce_dict(input_dict, average=True):
    """mean (or sum) of every value over the ranks, for logging (reference src/utils/misc_util.py:115-139)."""
    world = get_world_size()
    if world < 2:
        return input_dict
    keys = sorted(input_dict)
    with torch.no_grad():
        packed = torch.stack([input_dict[k].detach().float() for k in keys])
        dist.all_reduce(packed)
        if average:
            packed /= world
    return dict(zip(keys, packed))


def all_gather(data):
    """every rank's picklable `data` as a list (reference misc_util.all_gather, :72-110: pickle + padded tensors;
    here torch.distributed's object collective does the same job)"""
    if get_world_size() == 1:
        return [data]
    out = [None] * get_world_size()
    torch.distributed.all_gather_object(out, data)
    return out

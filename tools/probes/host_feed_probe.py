#!/usr/bin/env python
"""PROBE: what the host of a GPU box can feed -- CPU quota, synthetic-batch generation rate (1 / 4 / 8 / 16 threads),
pageable -> pinned staging rate (1 / 4 threads), pinned -> device DMA rate.  One batch = 16 x 3x800x1333 fp32 = 205 MB."""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    print('logical cpus %d, affinity %d, torch threads %d' % (os.cpu_count(), len(os.sched_getaffinity(0)), torch.get_num_threads()))
    for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
        try:
            print(f, open(f).read().strip())
        except OSError:
            pass
    n = 3 * 800 * 1333

    def gen(seed, pin, out):
        g = torch.Generator().manual_seed(seed)
        out.append([torch.rand(3, 800, 1333, generator=g, out=torch.empty(3, 800, 1333, pin_memory=pin)) for _ in range(16)])

    for pin in (False, True):
        for nt in (1, 4, 8, 16):
            for rep in range(2):            # second repetition: the pinned allocator's cache is warm
                outs = []
                t0 = time.time()
                ths = [threading.Thread(target=gen, args=(i, pin, outs)) for i in range(nt)]
                [t.start() for t in ths]
                [t.join() for t in ths]
                dt = time.time() - t0
                del outs
            print('generate %2d batches on %2d threads, pinned=%s: %.3f s -> %.1f ms per batch' % (nt, nt, pin, dt, dt / nt * 1e3))
    src = [torch.rand(3, 800, 1333) for _ in range(16)]
    pinned = torch.empty(16 * n, pin_memory=True)
    pinned.zero_()
    for nt in (1, 4):
        t0 = time.time()
        for rep in range(3):
            def cp(lo, hi):
                for i in range(lo, hi):
                    pinned[i * n:(i + 1) * n].view(3, 800, 1333).copy_(src[i])
            ths = [threading.Thread(target=cp, args=(16 * j // nt, 16 * (j + 1) // nt)) for j in range(nt)]
            [t.start() for t in ths]
            [t.join() for t in ths]
        dt = (time.time() - t0) / 3
        print('stage 205 MB pageable -> pinned with %d thread(s): %.1f ms (%.1f GB/s)' % (nt, dt * 1e3, 16 * n * 4 / dt / 1e9))
    if torch.cuda.is_available():
        dev = torch.empty(16 * n, device='cuda')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for label, srcs in (('one 205 MB copy', [pinned]), ('16 copies of 12.8 MB', [pinned[i * n:(i + 1) * n] for i in range(16)])):
            e0.record()
            for rep in range(5):
                off = 0
                for t in srcs:
                    dev[off:off + t.numel()].copy_(t, non_blocking=True)
                    off += t.numel()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print('pinned -> device, %s: %.2f ms (%.1f GB/s)' % (label, ms, 16 * n * 4 / ms / 1e6))


if __name__ == '__main__':
    main()

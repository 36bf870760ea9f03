"""Host -> device feeding of the distillation step (the reference's ``images.to(device)`` / ``targets.to(device)``,
src/mimic_runner.py:49-50, src/ext_runner.py:47-48), taken off the step's critical path.

The reference uploads every tensor of a batch synchronously from pageable memory at the top of the step: 16 x 12.8 MB
images + the targets, ~3.3 ms of PCIe time per step at batch 16 plus the pageable staging, all of it with the GPU idle.
``DevicePrefetcher`` wraps any iterable of ``(images, targets)`` host batches (torch DataLoader, SyntheticDetectionLoader)
and yields the same structure with every tensor already on the device:

  * a feeder thread pulls batch k+1 from the source while step k is being enqueued / computed, lays ALL of its tensors
    (images -- float CHW or ``DecodedImage`` uint8 -- and every target tensor) out in ONE pinned staging buffer (big
    tensors are copied in by a few helper threads) and issues ONE asynchronous copy of it on a copy stream of its own:
    one DMA of 205 MB (51 MB on the uint8 path) instead of ~50 small ones, running under the previous step's kernels.
    Tensors the loader already delivers in pinned memory (``DataLoader(pin_memory=True)``, ``SyntheticDetectionLoader(
    pin_memory=True)``) skip the staging copy: they go to their place in the device buffer straight from where they are;
  * ``depth`` slots (pinned buffer + device buffer + two events each) rotate; a slot's device buffer is rewritten only
    after the step that consumed it has finished ON THE GPU (event recorded on the consumer's stream when it asks for
    the next batch; the copy stream waits for it), its pinned buffer only after the copy out of it has completed;
  * the consumer's stream waits for the copy's event -- never the host.

The tensors handed out are views of the slot's device buffer: valid until ``depth - 1`` further batches have been
requested, which is what a training loop does (one batch alive per step).  Nothing here computes: torch is used for
pinned / device memory, streams and events only.
"""
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import torch

_ALIGN = 256
_BIG = 1 << 20          # tensors of at least this many bytes are staged by the helper threads


def _round_up(x, m):
    return (x + m - 1) // m * m


class _Slot(object):
    __slots__ = ('pinned', 'dev', 'ready', 'consumed', 'free', 'keep')

    def __init__(self):
        self.pinned = self.dev = None
        self.keep = None                         # pinned source tensors of the copy in flight (alive until `ready`)
        self.ready = torch.cuda.Event()          # the copy into `dev` has completed (recorded on the copy stream)
        self.consumed = None                     # the step that read `dev` has completed (consumer's stream), or None
        self.free = threading.Semaphore(1)       # host side: the consumer no longer holds views of this slot


def _leaves(images, targets):
    """[(kind, index, key, tensor)] of every tensor of a batch, in a fixed order"""
    out = []
    for i, im in enumerate(images):
        out.append(('image', i, None, im.data if hasattr(im, 'hwc') else im))
    for i, t in enumerate(targets):
        for k, v in t.items():
            if torch.is_tensor(v):
                out.append(('target', i, k, v))
    return out


class DevicePrefetcher(object):
    """iterable of device-resident ``(images, targets)`` over a host-batch iterable; see the module docstring.

    ``len()`` and ``set_epoch`` pass through to the source.  ``copied_bytes`` / ``batches`` count what went over the
    link, ``wait_s`` / ``waits`` the host time ``next()`` spent blocked on the feeder (bench.py reports them;
    ``reset_wait()`` zeroes the latter at the start of a timed region)."""

    def __init__(self, source, device, depth=3, stagers=4):
        assert depth >= 2, 'one slot is read by the step while the next is being filled'
        self.source, self.device, self.depth = source, torch.device(device), depth
        if self.device.type != 'cuda':
            raise RuntimeError('DevicePrefetcher feeds the HIP path: it needs a cuda device, got %s' % (self.device,))
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [_Slot() for _ in range(depth)]
        self.copied_bytes = self.batches = 0
        self.wait_s, self.waits = 0.0, 0         # host time the consumer spent blocked on the feeder (bench.py: data_wait_ms)
        self._thread = self._queue = self._stop = None
        self._held = None
        self._pool = ThreadPoolExecutor(stagers, thread_name_prefix='upload-stager') if stagers > 1 else None

    def __len__(self):
        return len(self.source)

    def reset_wait(self):
        self.wait_s, self.waits = 0.0, 0

    def set_epoch(self, epoch):
        if hasattr(self.source, 'set_epoch'):
            self.source.set_epoch(epoch)

    # ------------------------------------------------------------------ feeder thread
    def _stage(self, slot, images, targets):
        leaves = _leaves(images, targets)
        nbytes = [t.numel() * t.element_size() for _, _, _, t in leaves]
        # big tensors the loader already holds in pinned memory go straight to the device; everything else is staged.
        # Layout: direct tensors first, the staged ones behind them in one contiguous range (= one copy)
        is_direct = [nb >= _BIG and t.is_pinned() and t.is_contiguous() for (_, _, _, t), nb in zip(leaves, nbytes)]
        offs, total = [0] * len(leaves), 0
        for want in (True, False):
            if not want:
                stage_lo = total
            for i, nb in enumerate(nbytes):
                if is_direct[i] == want:
                    offs[i] = total
                    total = _round_up(total + nb, _ALIGN)
        total = max(total, _ALIGN)
        slot.ready.synchronize()                 # the previous copy OUT of this slot's host memory is complete
        slot.keep = None
        if slot.pinned is None or slot.pinned.numel() < total:
            slot.pinned = torch.empty(total, dtype=torch.uint8, pin_memory=True)
            # the device buffer belongs to the COPY stream's pool of the caching allocator: allocated under the thread's
            # default stream it could be a block another thread has just freed with kernels still pending on it, and the
            # copy stream (which is not ordered with them) would write into it first
            with torch.cuda.stream(self.copy_stream):
                slot.dev = torch.empty(total, dtype=torch.uint8, device=self.device)
        views, jobs = [], []
        for (_, _, _, t), off, nb, direct in zip(leaves, offs, nbytes, is_direct):
            views.append(slot.dev[off:off + nb].view(t.dtype).view(t.shape))
            if direct:
                continue
            dst = slot.pinned[off:off + nb].view(t.dtype).view(t.shape)
            if nb >= _BIG and self._pool is not None:
                jobs.append(self._pool.submit(dst.copy_, t))
            else:
                dst.copy_(t)
        for j in jobs:
            j.result()
        with torch.cuda.stream(self.copy_stream):
            if slot.consumed is not None:        # the step that read this device buffer has finished on the GPU
                self.copy_stream.wait_event(slot.consumed)
            for (_, _, _, t), v, direct in zip(leaves, views, is_direct):
                if direct:
                    v.copy_(t, non_blocking=True)
            if stage_lo < total:
                slot.dev[stage_lo:total].copy_(slot.pinned[stage_lo:total], non_blocking=True)
            slot.keep = [t for (_, _, _, t), direct in zip(leaves, is_direct) if direct]
            slot.ready.record(self.copy_stream)
        self.copied_bytes += total
        self.batches += 1
        # the same structure as the host batch, tensors replaced by their device views
        dev_images = list(images)
        dev_targets = [dict(t) for t in targets]
        for (kind, i, k, _), v in zip(leaves, views):
            if kind == 'image':
                im = images[i]
                dev_images[i] = type(im)(v, im.hwc, im.flip) if hasattr(im, 'hwc') else v
            else:
                dev_targets[i][k] = v
        return dev_images, dev_targets

    def _feed(self, it, q, stop):
        try:
            torch.cuda.set_device(self.device)
            k = 0
            for images, targets in it:
                slot = self.slots[k % self.depth]
                while not slot.free.acquire(timeout=0.2):
                    if stop.is_set():
                        return
                if stop.is_set():
                    return
                q.put(('batch', k, self._stage(slot, images, targets)))
                k += 1
            q.put(('end', k, None))
        except BaseException as exc:             # surfaces in the consumer's next()
            q.put(('error', -1, exc))

    # ------------------------------------------------------------------ consumer
    def __iter__(self):
        self.close()
        it = iter(self.source)                   # (in the caller's thread: a loader may seed here)
        for s in self.slots:
            s.free = threading.Semaphore(1)
        self._queue, self._stop = queue.Queue(), threading.Event()
        self._thread = threading.Thread(target=self._feed, args=(it, self._queue, self._stop), daemon=True)
        self._thread.start()
        return self._consume(self._queue)

    def _release_held(self):
        if self._held is not None:
            slot, self._held = self._held, None
            if slot.consumed is None:
                slot.consumed = torch.cuda.Event()
            slot.consumed.record(torch.cuda.current_stream(self.device))     # everything enqueued so far has used it
            slot.free.release()

    def _consume(self, q):
        try:
            while True:
                self._release_held()
                t0 = time.perf_counter()
                kind, k, payload = q.get()
                self.wait_s += time.perf_counter() - t0
                self.waits += 1
                if kind == 'end':
                    return
                if kind == 'error':
                    raise payload
                slot = self.slots[k % self.depth]
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(slot.ready)
                slot.dev.record_stream(cur)      # (a regrown buffer is not recycled under this stream's readers)
                self._held = slot
                yield payload
        finally:
            self._release_held()
            self.close()

    def close(self):
        """stop the feeder (an abandoned epoch); idempotent"""
        if self._thread is not None:
            self._stop.set()
            for s in self.slots:                 # wake a feeder parked on a slot
                s.free.release()
            self._thread.join(timeout=30.0)
            self._thread = None

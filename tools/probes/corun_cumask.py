#!/usr/bin/env python
"""can an MFMA-bound GEMM chain and an HBM-bound streaming chain share the chip SPATIALLY?

tools/probes/corun.py showed that two full-chip kernels on two streams are time-sliced by the hardware.  This probe
gives each chain its own compute units with hipExtStreamCreateWithCUMask (no privileges needed: the mask travels with
the queue): the GEMMs get `G` CUs, the streaming kernels the other 256 - G.  Times each chain alone on the full chip,
alone on its partition, and both together.

    python tools/probes/corun_cumask.py            (prints one table; numbers go to profiles/r04_overlap.txt)
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('HND_BRES', '0')          # tiled kernel: many small blocks adapt to any CU count
os.environ.setdefault('HND_BSTREAM', '0')
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

hip = C.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(cus):
    """a torch stream whose kernels may only run on the CUs in `cus` (iterable of CU indices 0..255)"""
    words = [0] * 8
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    arr = (C.c_uint32 * 8)(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
    if rc != 0:
        raise RuntimeError('hipExtStreamCreateWithCUMask failed: %d' % rc)
    return torch.cuda.ExternalStream(s.value)


dev = 'cuda:0'
n, h, w, c = 16, 200, 336, 256
x = torch.randn(n, h, w, c, device=dev)
y = torch.empty(n, h, w, c, device=dev)
pk = ops.pack_weights(torch.randn(c, c, 1, 1, device=dev) / c ** 0.5)
a = torch.randn(n, h, w, c, device=dev)
b = torch.empty_like(a)
sc, sh = torch.rand(c, device=dev), torch.randn(c, device=dev)
l = ops.conv_forward(x, pk, y, 1, 1, 0, relu=True)
NG, NH = 20, 50
gbytes = NH * 2 * a.numel() * 4 / 1e9
gflop = NG * l.flops / 1e9


def run_gemms(s):
    with torch.cuda.stream(s):
        for _ in range(NG):
            l.run()


def run_streams(s):
    with torch.cuda.stream(s):
        for _ in range(NH):
            ops.affine_relu(a, sc, sh, b, True)


# hipExtStreamCreateWithCUMask makes BLOCKING streams: anything enqueued on the NULL stream (torch's default stream) between
# the two chains -- an event record for a wait_stream, say -- would serialise them.  So the clock runs on a non-blocking
# side stream: its start event is waited for by both chains BEFORE either is enqueued, and nothing touches the null stream.
clock = torch.cuda.Stream()


def timed(fn, streams):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(clock):
        e0.record()
        for s in streams:
            s.wait_event(e0)
        fn()
        for s in streams:
            clock.wait_stream(s)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


full1, full2 = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(2):
    run_gemms(full1); run_streams(full2)
tg = timed(lambda: run_gemms(full1), [full1])
th = timed(lambda: run_streams(full2), [full2])
tb = timed(lambda: (run_gemms(full1), run_streams(full2)), [full1, full2])
print('full chip, two ordinary streams: GEMMs alone %.2f ms (%.1f TF) | streaming alone %.2f ms (%.2f TB/s) | both %.2f ms '
      '(sum %.2f)' % (tg, gflop / tg, th, gbytes / th, tb, tg + th), flush=True)
print('%-34s %10s %10s %10s %10s %10s' % ('partition (GEMM CUs / stream CUs)', 'GEMM part', 'strm part', 'both', 'sum alone', 'ideal max'))
for how in ('low/high', 'interleaved'):
    for gcus in (240, 224, 208, 192, 160, 128):
        if how == 'low/high':
            g_set, s_set = range(gcus), range(gcus, 256)
        else:                       # every (256 / (256 - gcus))-th CU goes to the streaming chain
            step = 256 // (256 - gcus)
            s_set = [i for i in range(256) if i % step == step - 1][:256 - gcus]
            g_set = [i for i in range(256) if i not in set(s_set)]
        sg, ss = masked_stream(g_set), masked_stream(s_set)
        for _ in range(2):
            run_gemms(sg); run_streams(ss)
        pg = timed(lambda: run_gemms(sg), [sg])
        ph = timed(lambda: run_streams(ss), [ss])
        pb = timed(lambda: (run_gemms(sg), run_streams(ss)), [sg, ss])
        print('%-34s %7.2f ms %7.2f ms %7.2f ms %7.2f ms %7.2f ms   GEMM %.1f TF, stream %.2f TB/s on their parts; '
              'together vs full-chip sum: %+.1f %%' % ('%s %d / %d' % (how, gcus, 256 - gcus), pg, ph, pb, tg + th,
                                                      max(pg, ph), gflop / pg, gbytes / ph, 100.0 * (pb / (tg + th) - 1)),
              flush=True)

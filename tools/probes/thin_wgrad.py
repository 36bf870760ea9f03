#!/usr/bin/env python
"""Timing of the two 3-channel weight gradients (layer1.conv3 / conv4 of the b3ch head) at batch 16.
usage: python tools/probes/thin_wgrad.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hnd_ghnd_object_detectors_amd import ops  # noqa: E402

dev = 'cuda:0'
for name, cin, cout, pad in (('conv3.wgrad 64->3', 64, 3, 1), ('conv4.wgrad 3->64', 3, 64, 0)):
    n, h, w = 16, 201, 337
    cp, op = ops.chan_pad_of(cin), ops.chan_pad_of(cout)
    x = torch.randn(n, h, w, cp, device=dev)
    oh, ow = h + 2 * pad - 1, w + 2 * pad - 1
    dy = torch.randn(n, oh, ow, op, device=dev)
    ps, pb = torch.rand(cp, device=dev) + 0.5, torch.randn(cp, device=dev)
    for mode in ('0', '1'):
        os.environ['HND_THIN_WGRAD'] = mode
        dw = torch.empty(cout, cin, 2, 2, device=dev)
        l = ops.conv_wgrad(x, dy, dw, 2, 1, pad, pro_scale=ps, pro_shift=pb, pro_relu=True)
        for _ in range(3):
            l.run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            l.run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = (x.numel() + dy.numel()) * 4 / 1e9
        print('%-20s %-11s  %.3f ms  %.2f TB/s' % (name, l.variant, ms, gb / ms), flush=True)

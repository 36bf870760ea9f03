"""Put the import shim + /root/reference/src on sys.path (build container only).

TEST INFRASTRUCTURE.  Called by tests/golden/make_golden.py and by the CPU-only
``reference`` tests that are skipped when /root/reference is absent (GPU box).
"""
import os
import sys
import types

REFERENCE_SRC = '/root/reference/src'
SHIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shim')
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_available():
    return os.path.isdir(REFERENCE_SRC)


def install():
    if not reference_available():
        raise RuntimeError('reference tree not present: %s' % REFERENCE_SRC)
    import torch
    if 'torch._six' not in sys.modules:  # removed from torch>=2; src/utils/coco_eval_util.py:8 imports it
        six = types.ModuleType('torch._six')
        six.string_classes = (str, bytes)
        sys.modules['torch._six'] = six
        torch._six = six
    for p in (REPO_ROOT, REFERENCE_SRC, SHIM_DIR):   # SHIM_DIR ends up first
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)

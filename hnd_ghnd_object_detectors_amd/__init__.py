"""MI355X-native (gfx950) head-network-distillation path of hnd-ghnd-object-detectors.

Layout mirrors the reference's ``src/`` so its users find the same entry points:
``models`` (get_model / load_ckpt / save_ckpt), ``distillation`` (DistillationBox, loss),
``myutils`` (restated helper subset), ``utils``, ``mimic_runner``.  All arithmetic runs in
``libhnd_hip.so`` (C ABI in ``include/hnd_hip.h``); there is no CPU or eager fallback.
"""
__version__ = '0.1.0'

"""CPU: the C-ABI library builds, loads, and exports every symbol include/hnd_hip.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'hnd_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(hnd_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from hnd_ghnd_object_detectors_amd import _lib
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    assert lib.hnd_abi_version() == _lib.ABI_VERSION == 12


def test_ctypes_structs_match_header_layout():
    from hnd_ghnd_object_detectors_amd import _lib
    import ctypes
    assert ctypes.sizeof(_lib.ConvDesc) == 11 * 8 + 30 * 4 + 3 * 8 + 5 * 8 + 8 + 8 + 8      # (+ w_bf16x3, ABI 10; + w_bf16x3s, ABI 12)
    assert ctypes.sizeof(_lib.WgradDesc) == 6 * 8 + 16 * 4 + 3 * 8
    assert ctypes.sizeof(_lib.MsePair) == 3 * 8 + 8 + 4 + 4
    assert ctypes.sizeof(_lib.ImageDesc) == 8 + 7 * 4 + 2 * 4 + 4        # hnd_image_desc (padded to 8)
    assert ctypes.sizeof(_lib.BoxesDesc) == 2 * 8 + 4 + 2 * 4 + 4          # hnd_boxes_desc (padded to 8)
    assert ctypes.sizeof(_lib.PackDesc) == 2 * 8 + 12 * 4                  # hnd_pack_desc


def test_invalid_arguments_are_reported_not_thrown():
    from hnd_ghnd_object_detectors_amd import _lib
    lib = _lib.load()
    assert lib.hnd_conv2d_igemm(None, None) == -1
    assert b'null descriptor' in lib.hnd_last_error_string()
    assert lib.hnd_adam_step_flat(None, None, None, None, 0, 0.0, 0.0, 0.0, 0.0, 0, 1.0, None) == -1


def test_detection_operators_validate_their_arguments():
    """the validation-path entry points (ABI 3 / 4) reject null pointers and bad geometry with HND_ERR_INVALID and a
    message, and treat an empty problem (k = 0) as a successful no-op -- all before any HIP call, so this runs on CPU"""
    from hnd_ghnd_object_detectors_amd import _lib
    lib = _lib.load()
    assert lib.hnd_mask_probs(None, None, 0, 28, 91, None, None) == 0
    assert lib.hnd_mask_probs(None, None, 5, 28, 91, None, None) == -1
    assert b'hnd_mask_probs' in lib.hnd_last_error_string()
    assert lib.hnd_paste_masks(None, None, 0, 28, 800, 1333, None, None) == 0
    assert lib.hnd_paste_masks(None, None, 3, 28, 800, 1333, None, None) == -1
    assert lib.hnd_upsample_bilinear_nhwc(None, 2, 28, 28, 17, 2, None, None) == -1
    assert lib.hnd_heatmaps_to_keypoints(None, 0, 56, 56, 17, 17, None, None, None, None) == 0
    assert lib.hnd_heatmaps_to_keypoints(None, 4, 56, 56, 17, 17, None, None, None, None) == -1
    assert b'hnd_heatmaps_to_keypoints' in lib.hnd_last_error_string()
    assert lib.hnd_nms(None, None, 0, 0.5, None, None, None) == 0
    assert lib.hnd_nms(None, None, 10, 0.5, None, None, None) == -1
    assert lib.hnd_roi_align(None, 1, 8, 8, 64, None, None, 3, 0.25, 7, 7, 2, None, None) == -1
    assert lib.hnd_nms_workspace(4800) == 4800 * 75 * 8


def test_bres2_inline_asm_ring_is_untouched_between_load_and_wait():
    """csrc/conv_bres.hip / conv_bstream.hip / conv_wgrad_ring.hip / conv_bx3.hip / conv_bxs.hip hide their ring loads (A fragments; the parked B-stage
    loads of bstream; both operands of the ring weight gradient) from hipcc -- inline asm `global_load_dwordx4` +
    hand-counted `s_waitcnt vmcnt(N)` -- and the compiler is then free to copy / spill / reuse a ring register before its
    data has landed.  tools/audit_bres_asm.py walks the control-flow graph of the generated assembly (every path, loop
    back edges included) and must find no instruction touching a ring register between its asm load and the wait that
    releases it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'tools', 'audit_bres_asm.py')], capture_output=True,
                         text=True, timeout=1800)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert all(k in res.stdout for k in ('bres2_kernel', 'bstream_kernel', 'wgrad_ring_kernel', 'bx3_kernel', 'bxs_kernel'))
    assert res.stdout.count(' 0 problem(s)') == 5 and ': 0 16-byte loads' not in res.stdout

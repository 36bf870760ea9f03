"""ctypes binding of libhnd_hip.so (C ABI declared in include/hnd_hip.h).

The library is the ONLY compute path of this package: if it is missing the import of any
compute module fails loudly (no CPU / eager fallback exists on purpose).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('HND_LIB_PATH') or os.path.join(_HERE, 'libhnd_hip.so')     # env: kernel experiments
ABI_VERSION = 12

c_float_p = C.POINTER(C.c_float)
vp = C.c_void_p


class ConvDesc(C.Structure):
    """struct hnd_conv_desc"""
    _fields_ = [(n, vp) for n in ('x', 'w', 'y', 'pro_scale', 'pro_shift', 'epi_scale', 'epi_shift',
                                  'res1', 'res2', 'mask', 'stats')] + \
               [(n, C.c_int32) for n in ('n', 'h', 'w_', 'cin', 'oh', 'ow', 'yh', 'yw', 'cout', 'ldc',
                                         'y_sh', 'y_oh', 'y_sw', 'y_ow', 'kh', 'kw',
                                         'sh', 'dh', 'bh', 'sw', 'dw', 'bw', 'kdim', 'pro_relu', 'relu',
                                         'res1_mode', 'res1_h', 'res1_w', 'w_group_rows', 'w_group_stride')] + \
               [('relay_ws', vp), ('mask_bits', vp), ('mask_out', vp)] + \
               [(k, vp) for k in ('bwd_x', 'bwd_scale', 'bwd_shift', 'bwd_mean', 'bwd_rstd')] + [('bwd_relu', C.c_int)] + \
               [('w_bf16x3', vp), ('w_bf16x3s', vp)]


class ImageDesc(C.Structure):
    """struct hnd_image_desc"""
    _fields_ = [('src', vp)] + [(n, C.c_int32) for n in ('h', 'w', 'out_h', 'out_w', 'is_u8', 'hwc', 'flip')] + \
               [('scale_h', C.c_float), ('scale_w', C.c_float)]


class BoxesDesc(C.Structure):
    """struct hnd_boxes_desc"""
    _fields_ = [('src', vp), ('dst', vp), ('k', C.c_int32), ('scale_w', C.c_float), ('scale_h', C.c_float)]


class PackDesc(C.Structure):
    """struct hnd_pack_desc"""
    _fields_ = [('src', vp), ('dst', vp)] + [(n, C.c_int32) for n in ('cout', 'cin', 'kh', 'kw', 'transposed', 'chan_pad',
                                                                      'i0', 'istep', 'ni', 'j0', 'jstep', 'nj')]


class WgradDesc(C.Structure):
    """struct hnd_wgrad_desc"""
    _fields_ = [(n, vp) for n in ('x', 'dy', 'dw', 'slabs', 'pro_scale', 'pro_shift')] + \
               [(n, C.c_int32) for n in ('n', 'h', 'w_', 'cin', 'cin_real', 'oh', 'ow', 'cout', 'ldy',
                                         'kh', 'kw', 'stride', 'pad', 'pro_relu', 'splitk', 'groups')] + \
               [(n, C.c_int64) for n in ('x_group_stride', 'dy_group_stride', 'dw_group_stride')]


class MsePair(C.Structure):
    """struct hnd_mse_pair"""
    _fields_ = [('teacher', vp), ('student', vp), ('grad', vp), ('numel', C.c_int64),
                ('factor', C.c_float), ('relu_mask', C.c_int32)]


_SIGNATURES = {
    'hnd_last_error_string': (C.c_char_p, []),
    'hnd_abi_version': (C.c_int, []),
    'hnd_sync_check': (C.c_int, [vp]),
    'hnd_relay_timeouts': (C.c_int, [C.c_int]),
    'hnd_device_arch': (C.c_char_p, []),
    'hnd_conv2d_igemm': (C.c_int, [C.POINTER(ConvDesc), vp]),
    'hnd_conv2d_igemm_tile': (C.c_int, [C.POINTER(ConvDesc)]),
    'hnd_conv2d_igemm_workspace': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'hnd_conv2d_wgrad_workspace': (C.c_size_t, [C.POINTER(WgradDesc)]),
    'hnd_conv2d_wgrad': (C.c_int, [C.POINTER(WgradDesc), vp]),
    'hnd_conv2d_wgrad_variant': (C.c_int, [C.POINTER(WgradDesc)]),
    'hnd_pack_weights': (C.c_int, [vp, vp] + [C.c_int] * 12 + [vp]),
    'hnd_bf16x3_recommended': (C.c_int, [C.c_int64, C.c_int, C.c_int]),
    'hnd_pack_bf16x3_elems': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'hnd_pack_bf16x3': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp]),
    'hnd_bf16x3s_recommended': (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    'hnd_pack_bf16x3s_elems': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'hnd_pack_bf16x3s': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp]),
    'hnd_scale_packed_k': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    'hnd_pack_weights_batched': (C.c_int, [C.POINTER(PackDesc), C.c_int, vp]),
    'hnd_fbn_fold': (C.c_int, [vp] * 6 + [C.c_int, C.c_int, C.c_float, vp]),
    'hnd_transform_image': (C.c_int, [vp, C.c_int, C.c_int, vp] + [C.c_int] * 5 + [C.c_float, C.c_float,
                                                                                    c_float_p, c_float_p, vp]),
    'hnd_transform_image_u8': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp] + [C.c_int] * 5 +
                               [C.c_float, C.c_float, c_float_p, c_float_p, vp]),
    'hnd_transform_images': (C.c_int, [C.POINTER(ImageDesc), C.c_int, vp, C.c_int, C.c_int, c_float_p, c_float_p, vp]),
    'hnd_scale_boxes': (C.c_int, [C.POINTER(BoxesDesc), C.c_int, vp]),
    'hnd_wino_tiles_pad': (C.c_int64, [C.c_int] * 4),
    'hnd_wino_weights': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'hnd_wino_input': (C.c_int, [vp, vp] + [C.c_int] * 4 + [vp, vp, C.c_int, C.c_int, vp]),
    'hnd_wino_output': (C.c_int, [vp, vp] + [C.c_int] * 5 + [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    'hnd_wino2_tiles_pad': (C.c_int64, [C.c_int] * 4),
    'hnd_wino2_stats_blocks': (C.c_int, [C.c_int] * 5),
    'hnd_wino2_weights': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'hnd_wino2_input': (C.c_int, [vp, vp] + [C.c_int] * 5 + [vp, vp, C.c_int, C.c_int, vp]),
    'hnd_wino2_output': (C.c_int, [vp, vp] + [C.c_int] * 5 + [vp, vp, C.c_int, vp, C.c_int, vp]),
    'hnd_wino2_dy': (C.c_int, [vp, vp] + [C.c_int] * 6 + [vp]),
    'hnd_wino26_output_bnbwd_stats': (C.c_int, [vp, vp] + [C.c_int] * 5 + [vp] * 5 + [C.c_int, vp, vp]),
    'hnd_wino26_bnbwd_transforms': (C.c_int, [vp] * 5 + [C.c_int] * 6 + [vp, vp, vp]),
    'hnd_wino2_wgrad_output': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    'hnd_wino2_wgrad_output_t': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'hnd_maxpool3x3s2_fwd': (C.c_int, [vp, vp, vp] + [C.c_int] * 6 + [vp]),
    'hnd_maxpool3x3s2_bwd_relu_scale': (C.c_int, [vp] * 5 + [C.c_int] * 6 + [vp]),
    'hnd_bn_finalize': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp, vp, vp, vp, vp,
                                  C.c_float, C.c_float, vp, vp, vp, vp, vp]),
    'hnd_affine_relu': (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp]),
    'hnd_relu_mask_nibbles': (C.c_int, [vp, vp, C.c_int64, vp]),
    'hnd_bn_bwd_ntiles': (C.c_int, [C.c_int64]),
    'hnd_bn_bwd_reduce': (C.c_int, [vp] * 6 + [C.c_int, C.c_int64, C.c_int, vp, vp]),
    'hnd_bn_bwd_finalize': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp, vp, vp, vp, vp, vp, vp]),
    'hnd_bn_bwd_apply': (C.c_int, [vp] * 5 + [C.c_int, vp, C.c_int64, C.c_int, vp]),
    'hnd_mse_scratch_elems': (C.c_size_t, []),
    'hnd_mse_sum_fwd_bwd': (C.c_int, [C.POINTER(MsePair), C.c_int, vp, vp, vp]),
    'hnd_scale_by_device_scalar': (C.c_int, [vp, C.c_int64, vp, vp]),
    'hnd_adam_step_flat': (C.c_int, [vp, vp, vp, vp, C.c_int64] + [C.c_float] * 4 + [C.c_int64, C.c_float, vp]),
    'hnd_subsample2': (C.c_int, [vp, vp] + [C.c_int] * 6 + [vp]),
    'hnd_fill': (C.c_int, [vp, C.c_int64, C.c_float, vp]),
    'hnd_upsample_nearest_bwd': (C.c_int, [vp, vp] + [C.c_int] * 7 + [vp]),
    'hnd_add_inplace': (C.c_int, [vp, vp, C.c_int64, vp]),
    'hnd_minmax_scratch_elems': (C.c_size_t, []),
    'hnd_quantize_u8': (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, vp, vp, vp, vp]),
    'hnd_dequantize_u8': (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    'hnd_roundtrip_f16': (C.c_int, [vp, C.c_int64, vp]),
    'hnd_adaptive_avgpool_fwd': (C.c_int, [vp, vp] + [C.c_int] * 6 + [vp]),
    'hnd_adaptive_avgpool_bwd': (C.c_int, [vp, vp] + [C.c_int] * 6 + [vp]),
    'hnd_linear_fwd': (C.c_int, [vp] * 4 + [C.c_int] * 5 + [vp]),
    'hnd_linear_bwd': (C.c_int, [vp] * 6 + [C.c_int] * 5 + [vp]),
    'hnd_softmax_rows': (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    'hnd_softmax_ce_rows_fwd_bwd': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int64, vp, vp, vp]),
    'hnd_channel_sum_scratch_elems': (C.c_size_t, [C.c_int]),
    'hnd_channel_sum': (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp]),
    'hnd_sgd_step_flat': (C.c_int, [vp, vp, vp, C.c_int64] + [C.c_float] * 4 + [C.c_int, C.c_int, C.c_float, vp]),
    'hnd_comm_unique_id': (C.c_int, [vp, C.c_size_t]),
    'hnd_comm_init': (C.c_int, [C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(vp)]),
    'hnd_allreduce_avg_flat': (C.c_int, [vp, vp, C.c_int64, vp]),
    'hnd_comm_destroy': (C.c_int, [vp]),
    'hnd_workspace_size': (C.c_size_t, [C.c_int, vp, C.c_int64]),
    'hnd_rpn_decode': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, C.c_float, C.c_float,
                                 C.c_int64, C.c_int64, C.c_float, vp, vp, vp]),
    'hnd_clip_boxes': (C.c_int, [vp, C.c_int64, C.c_float, C.c_float, vp]),
    'hnd_nms_workspace': (C.c_size_t, [C.c_int64]),
    'hnd_nms': (C.c_int, [vp, vp, C.c_int64, C.c_float, vp, vp, vp]),
    'hnd_roi_align': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int64, C.c_float, C.c_int, C.c_int,
                                C.c_int, vp, vp]),
    'hnd_box_decode_clip': (C.c_int, [vp, C.c_int, vp, vp, C.c_int64, C.c_int] + [C.c_float] * 5 + [vp, vp]),
    'hnd_mask_probs': (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp]),
    'hnd_paste_masks': (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp]),
    'hnd_nonzero_u8': (C.c_int, [vp, C.c_int64, vp, vp, vp]),
    'hnd_nonzero_gt_f32': (C.c_int, [vp, C.c_int64, C.c_float, vp, vp, vp]),
    'hnd_nonzero_eq_i64': (C.c_int, [vp, C.c_int64, C.c_int64, vp, vp, vp]),
    'hnd_nonzero_min_size': (C.c_int, [vp, C.c_int64, C.c_float, vp, vp, vp]),
    'hnd_argsort_desc_workspace': (C.c_size_t, [C.c_int64]),
    'hnd_argsort_desc_f32': (C.c_int, [vp, C.c_int64, vp, vp, vp]),
    'hnd_mask_run_boundaries': (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_float, vp, C.c_int64, vp, vp, vp]),
    'hnd_resize_mask_nearest_u8': (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp, vp]),
    'hnd_upsample_bilinear_nhwc': (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    'hnd_heatmaps_to_keypoints': (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None


class HndLibraryError(RuntimeError):
    pass


def load():
    """dlopen libhnd_hip.so and declare every entry point; raises if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HndLibraryError(
            'libhnd_hip.so not found at %s: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(or make -C hnd_ghnd_object_detectors_amd/csrc). There is no CPU fallback.' % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so) and owns the device tensors this library works
    # on: load it FIRST so libhnd_hip.so binds to that copy.  Loaded the other way round the process holds two HIP
    # runtimes and this library's launches fail with "no ROCm-capable device is detected" (seen when build() and
    # smoke() ran in one process).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HndLibraryError('libhnd_hip.so does not export %s (stale build?)' % name)
        fn.restype = res
        fn.argtypes = args
    if lib.hnd_abi_version() != ABI_VERSION:
        raise HndLibraryError('libhnd_hip.so ABI %d != expected %d' % (lib.hnd_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().hnd_last_error_string()
        raise RuntimeError('%s failed (status %d): %s' % (what or 'libhnd_hip call', rc,
                                                          msg.decode() if msg else '?'))

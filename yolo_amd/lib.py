"""ctypes binding of libyolo_amd.so (include/yolo_amd.h).

The product path has no CPU fallback: if the shared library is missing or a call fails,
an exception is raised.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('YOLO_AMD_LIB') or os.path.join(CSRC, 'libyolo_amd.so')   # override: experiment builds

F32, BF16, F16, BF16X3, F16X3 = 0, 1, 2, 3, 4
ABI_VERSION = 4            # include/yolo_amd.h: YOLO_ABI_VERSION (the struct layouts below are revision 4's)
OK, EINVAL, EUNSUPPORTED = 0, -1, -2


class YoloError(RuntimeError):
    pass


def lab_knob(name, default=None):
    """A/B and ablation switches of tools/ (YOLO_TRAIN_*, YOLO_SIDE_FILTER, ...): read ONLY when YOLO_LAB=1 is set -- the product
    does not change behaviour with the environment of whoever imports it (the C library's knobs are compiled out the same way:
    csrc/common.h YOLO_LAB_ENV).  Two deliberate exceptions stay plain: YOLO_AMD_LIB (which library to load) and the test-only
    switches of bench.py / GradBuckets.active (YOLO_BENCH_*)."""
    if os.environ.get('YOLO_LAB') != '1':
        return default
    return os.environ.get(name, default)


class ConvDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w_packed', C.c_void_p), ('scale', C.c_void_p), ('bias', C.c_void_p),
                ('residual', C.c_void_p), ('y', C.c_void_p),
                ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('Cout', C.c_int),
                ('ksize', C.c_int), ('stride', C.c_int), ('dtype', C.c_int), ('out_f32', C.c_int),
                ('slope', C.c_float), ('y_batch_stride', C.c_longlong), ('y_pixel_stride', C.c_longlong),
                ('algo', C.c_int), ('x_pixel_stride', C.c_longlong), ('upsample2x', C.c_int),
                ('stats', C.c_void_p), ('stats_mode', C.c_int), ('stats_y', C.c_void_p), ('stats_mean', C.c_void_p),
                ('stats_invstd', C.c_void_p), ('stats_gamma', C.c_void_p), ('stats_beta', C.c_void_p),
                ('stats_slope', C.c_float),
                # fused tail 1x1 (include/yolo_amd.h: tail_*)
                ('tail_w_packed', C.c_void_p), ('tail_scale', C.c_void_p), ('tail_bias', C.c_void_p), ('tail_y', C.c_void_p),
                ('tail_cout', C.c_int), ('tail_out_f32', C.c_int), ('tail_slope', C.c_float),
                ('tail_y_batch_stride', C.c_longlong), ('tail_y_pixel_stride', C.c_longlong),
                # YOLO_BF16X3: element offset of the lo plane inside a pixel of x / y (0 = dense)
                ('x_lo_offset', C.c_longlong), ('y_lo_offset', C.c_longlong)]


class GridDesc(C.Structure):
    _fields_ = [('nscale', C.c_int), ('A', C.c_int), ('img_h', C.c_int), ('img_w', C.c_int),
                ('gh', C.c_int * 4), ('gw', C.c_int * 4), ('step', C.c_int * 4),
                ('anchors_hw', C.c_float * 64)]


# name -> (restype, argtypes); must list every symbol include/yolo_amd.h declares
_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
SIGNATURES = {
    'yolo_version': (_i, []),
    'yolo_packed_weight_bytes': (_ll, [_i, _i, _i, _i]),
    'yolo_pack_conv_weights': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'yolo_padded_channels': (_i, [_i]),
    'yolo_fold_bn': (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    'yolo_nchw_to_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'yolo_image_u8_to_nchw': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'yolo_nhwc_to_nchw': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'yolo_conv_fwd': (_i, [C.POINTER(ConvDesc), _vp]),
    'yolo_conv_kernel_name': (_i, [C.POINTER(ConvDesc), C.c_char_p, _i]),
    'yolo_conv_stats_rows': (_i, [C.POINTER(ConvDesc)]),
    'yolo_stem_conv_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'yolo_stem_stats_rows': (_i, [_i, _i, _i, _i]),
    'yolo_stem_conv_fwd_stats': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    'yolo_stem_down_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'yolo_res_block_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    'yolo_composite': (_i, [_vp, _vp, _vp, _vp, _ll, _vp]),
    'yolo_composite_unit': (_i, [_vp, _vp, _vp, _vp, _ll, _vp]),
    'yolo_upsample2x_concat': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'yolo_decode': (_i, [_vp, _vp, _i, _i, C.POINTER(GridDesc), _vp]),
    'yolo_decode_scores': (_i, [_vp, _vp, _vp, _i, _i, C.POINTER(GridDesc), _i, _vp]),
    'yolo_predict_top1': (_i, [_vp, _vp, _vp, _i, _i, C.POINTER(GridDesc), _vp]),
    'yolo_predict_lp': (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp]),
    'yolo_predict_lp_nhwc': (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp]),
    'yolo_iou_ltrb_vs_yxhw': (_i, [_vp, _vp, _vp, _i, _vp]),
    'yolo_iou_ltrb_vs_cltrb': (_i, [_vp, _vp, _vp, _i, _vp]),
    'yolo_nms_workspace_bytes': (_ll, [_i, _i, _i, _i, _i]),
    'yolo_nms_scores': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'yolo_nms_select_workspace_bytes': (_ll, [_i]),
    'yolo_nms_from_scores': (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'yolo_decode_nms': (_i, [_vp, _vp, _vp, _i, _i, C.POINTER(GridDesc), _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'yolo_nms': (_i, [_vp, _i, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'yolo_pack_conv_weights_dgrad': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'yolo_pack_conv_weights_dgrad_s2': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'yolo_conv_dgrad_s2': (_i, [C.POINTER(ConvDesc), _vp]),
    'yolo_pack_batch_blocks': (_ll, [_i, _i, _i, _i]),
    'yolo_pack_conv_weights_batch': (_i, [_vp, _vp, _i, _ll, _i, _vp]),
    'yolo_pack_pair_blocks': (_ll, [_i, _i, _i]),
    'yolo_pack_conv_weights_pairs': (_i, [_vp, _vp, _i, _ll, _vp]),
    'yolo_bn_train_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _f, _f, _f, _i, _vp]),
    'yolo_bn_train_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _f, _i, _vp]),
    'yolo_bn_train_fwd_pp': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _f, _f, _f, _i, _vp]),
    'yolo_bn_train_bwd_pp': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _f, _i, _vp]),
    'yolo_bn_train_fwd_partials': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _f, _f, _f, _i, _vp]),
    'yolo_bn_train_bwd_partials': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _f, _i, _vp]),
    'yolo_conv_wgrad_workspace_bytes': (_ll, [_i, _i, _i, _i]),
    'yolo_conv_wgrad': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ll, _i, _vp, _vp]),
    'yolo_conv_wgrad_algo': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ll, _i, _vp, _i, _vp]),
    'yolo_bias_grad': (_i, [_vp, _vp, _ll, _i, _ll, _i, _vp]),
    'yolo_gather_rows': (_i, [_vp, _vp, _i, _ll, _i, _i, _ll, _ll, _i, _vp]),
    'yolo_dilate2x': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'yolo_upsample2x_concat_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'yolo_add': (_i, [_vp, _vp, _vp, _ll, _i, _vp]),
    'yolo_assign_targets': (_i, [_vp, _vp, _vp, _i, _i, _i, C.POINTER(GridDesc), _vp]),
    'yolo_loss_fwd_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(C.c_float), _f, _f, _vp]),
    'yolo_assign_targets_lp': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp]),
    'yolo_loss_lp_fwd_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(C.c_float), _f, _f, _vp]),
    'yolo_adam_step': (_i, [_vp, _vp, _vp, _vp, _ll, _i, _f, _f, _f, _f, _f, _vp]),
    'yolo_adam_step_dev': (_i, [_vp, _vp, _vp, _vp, _ll, _i, _f, _f, _f, _f, _vp, _vp]),
}

_lib = None


def build(force=False):
    """Compile libyolo_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(['make', '-C', CSRC, 'clean'])
    subprocess.check_call(['make', '-C', CSRC, '-j%d' % max(4, min(8, os.cpu_count() or 4))])
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64: it must be in the process BEFORE libyolo_amd.so so both
    # share one HIP runtime (kernels launched through a second runtime fail with hipErrorNoDevice).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise YoloError('libyolo_amd.so not found at %s: build it with yolo_amd.lib.build() '
                        '(no CPU fallback exists)' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.yolo_version()
    if got != ABI_VERSION:
        raise YoloError('%s is ABI revision %d, this binding is revision %d: the struct layouts differ (rebuild the library: '
                        'yolo_amd.lib.build())' % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise YoloError('%s failed with status %d' % (what, rc))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def resolve_device(device):
    """torch.device with an explicit index: 'cuda' means the CURRENT device at construction time (resolved once -- an
    object built after torch.cuda.set_device(1) with device='cuda' lives on cuda:1, not cuda:0)."""
    import torch
    d = torch.device(device)
    if d.type == 'cuda' and d.index is None:
        d = torch.device('cuda', torch.cuda.current_device())
    return d


def require_current_device(device, what):
    """The library launches on the CURRENT device's current stream (one process per GPU: `torch.cuda.set_device(LOCAL_RANK)`
    first, as bench.py does).  An object living on another GPU than the current one would have its kernels launched on the
    wrong device with its pointers -- refuse that here instead."""
    import torch
    idx = device.index if device.index is not None else 0
    cur = torch.cuda.current_device()
    if cur != idx:
        raise YoloError('%s lives on cuda:%d but the current device is cuda:%d: call torch.cuda.set_device(%d) (one process '
                        'per GPU) before using it' % (what, idx, cur, idx))

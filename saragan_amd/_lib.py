"""ctypes binding of libsaragan_hip.so (C ABI: include/saragan_hip.h).  No fallback: if the shared
library is missing or a call fails, the product path raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SARAGAN_LIB: another build of the library (kernel A/B probes under tools/; never set in production)
LIB_PATH = os.environ.get('SARAGAN_LIB') or os.path.join(_HERE, 'libsaragan_hip.so')

SG_F32, SG_BF16 = 0, 1
SG_OPT_SGD, SG_OPT_MOMENTUM, SG_OPT_ADADELTA = 0, 1, 2
SG_EUNSUPPORTED = -4
SG_WGRAD_ACCUMULATE, SG_WGRAD_CLEAN_WORKSPACE = 1, 2      # sg_conv3d_wgrad_bias_ex flags


class ConvShape(C.Structure):
    _fields_ = [('n', C.c_int32), ('d', C.c_int32), ('h', C.c_int32), ('w', C.c_int32),
                ('cin', C.c_int32), ('cout', C.c_int32),
                ('kd', C.c_int32), ('kh', C.c_int32), ('kw', C.c_int32), ('upsample_in', C.c_int32)]


class ConvEpilogue(C.Structure):
    """sg_conv_epilogue.  struct_size is filled in here: positional arguments start at `bias`."""
    _fields_ = [('struct_size', C.c_uint32),
                ('bias', C.c_void_p), ('act', C.c_int32), ('slope', C.c_float), ('pixel_norm', C.c_int32),
                ('eps', C.c_float), ('pn_scale', C.c_void_p), ('mask_bits', C.c_void_p), ('mask_slope', C.c_float),
                ('sign_out', C.c_void_p), ('out_scale', C.c_int32), ('out_off', C.c_int32 * 3),
                ('tap_off', C.c_int32 * 3), ('pool', C.c_int32), ('workspace', C.c_void_p),
                ('workspace_bytes', C.c_size_t), ('x_plane_channels', C.c_int32), ('pn_bwd_y', C.c_void_p),
                ('pn_bwd_scale', C.c_void_p), ('in_mask_bits', C.c_void_p), ('in_mask_slope', C.c_float),
                ('in_gain', C.c_float), ('rgb_w', C.c_void_p), ('rgb_bias', C.c_void_p), ('rgb_out', C.c_void_p),
                ('pw_x', C.c_void_p), ('pw_wmat', C.c_void_p), ('pw_dx', C.c_void_p), ('pw_dw', C.c_void_p),
                ('pw_dbias', C.c_void_p), ('pw_coef', C.c_float)]

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class ProfEntry(C.Structure):
    _fields_ = [('kind', C.c_int32), ('shape', ConvShape), ('dtype', C.c_int32), ('launches', C.c_int64),
                ('total_ms', C.c_double), ('flops_per_launch', C.c_double), ('kernel', C.c_char * 64)]


_p, _i32, _i64, _f, _u64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_size_t
_SHP = C.POINTER(ConvShape)

# name -> (restype, argtypes); every symbol include/saragan_hip.h declares
SIGNATURES = {
    'sg_version': (C.c_char_p, []),
    'sg_error_string': (C.c_char_p, [C.c_int]),
    'sg_conv3d_packed_bytes': (_sz, [_SHP, C.c_int]),
    'sg_conv3d_pack_weights': (C.c_int, [_p, _f, C.c_int, _p, _SHP, C.c_int, _p]),
    'sg_conv3d_pack_weights_batch': (C.c_int, [C.c_int, _p, _p, _p, _p, _p, C.c_int, _p]),
    'sg_conv3d_fwd_workspace': (_sz, [_SHP, C.c_int]),
    'sg_conv3d_pw_epilogue_workspace': (_sz, []),
    'sg_conv3d_fwd': (C.c_int, [_p, _p, _p, _SHP, C.POINTER(ConvEpilogue), C.c_int, _p]),
    'sg_upconv3d_subpixel_supported': (C.c_int, [_SHP, C.c_int]),
    'sg_upconv3d_subpixel_packed_bytes': (_sz, [_SHP, C.c_int]),
    'sg_upconv3d_subpixel_pack': (C.c_int, [_p, _f, _p, _SHP, C.c_int, _p]),
    'sg_upconv3d_subpixel_fwd': (C.c_int, [_p, _p, _p, _SHP, C.POINTER(ConvEpilogue), C.c_int, _p]),
    'sg_upconv3d_subpixel_wgrad_supported': (C.c_int, [_SHP, C.c_int]),
    'sg_upconv3d_subpixel_wgrad_workspace': (_sz, [_SHP, C.c_int]),
    'sg_upconv3d_subpixel_wgrad': (C.c_int, [_p, _p, _p, _p, _f, _p, _sz, _SHP, C.c_int, _p]),
    'sg_upconv3d_subpixel_dgrad_supported': (C.c_int, [_SHP, C.c_int]),
    'sg_upconv3d_subpixel_dgrad_packed_bytes': (_sz, [_SHP, C.c_int]),
    'sg_upconv3d_subpixel_dgrad_pack': (C.c_int, [_p, _f, _p, _SHP, C.c_int, _p]),
    'sg_upconv3d_subpixel_dgrad': (C.c_int, [_p, _p, _p, _SHP, C.c_int, _p]),
    'sg_conv3d_wgrad_workspace': (_sz, [_SHP, C.c_int]),
    'sg_conv3d_wgrad': (C.c_int, [_p, _p, _p, _f, _p, _sz, _SHP, C.c_int, _p]),
    'sg_conv3d_pw_bwd': (C.c_int, [_p, _p, _p, _p, _p, _p, _f, _p, _sz, _SHP, C.c_int, _p]),
    'sg_conv3d_wgrad_bias': (C.c_int, [_p, _p, _p, _p, _f, _p, _sz, _SHP, C.c_int, _p]),
    'sg_conv3d_wgrad_bias_up_masked': (C.c_int, [_p, _p, _p, _f, _f, _p, _p, _f, _p, _sz, _SHP, C.c_int, _p]),
    'sg_conv3d_wgrad_bias_ex': (C.c_int, [_p, _p, _p, _f, _f, _p, _p, _f, C.c_uint, _p, _sz, _SHP, C.c_int, _p]),
    'sg_conv3d_wgrad_clean_bytes': (_sz, [_SHP, C.c_int]),
    'sg_bias_act_fwd': (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _f, C.c_int, _p]),
    'sg_bias_act_bwd_workspace': (_sz, [_i32]),
    'sg_bias_act_bwd': (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _f, C.c_int, _p]),
    'sg_bias_act_bwd_bits': (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _f, C.c_int, _p]),
    'sg_sign_words_bytes': (_sz, [_i64, _i32]),
    'sg_sign_words': (C.c_int, [_p, _p, _i64, _i32, C.c_int, _p]),
    'sg_pixel_norm_fwd': (C.c_int, [_p, _p, _p, _i64, _i32, _f, C.c_int, _p]),
    'sg_pixel_norm_bwd': (C.c_int, [_p, _p, _p, _p, _i64, _i32, C.c_int, _p]),
    'sg_pixel_norm_act_bwd': (C.c_int, [_p, _p, _p, _p, _f, _p, _p, _p, _i64, _i32, C.c_int, _p]),
    'sg_pixel_norm_act_bwd_pw': (C.c_int, [_p, _i32, _p, _p, _p, _p, _f, _p, _p, _p, _i64, _i32, C.c_int, _p]),
    'sg_pixel_norm_act_bwd_pw_wg': (C.c_int, [_p, _i32, _p, _p, _p, _p, _f, _p, _p, _p, _p, _f, _p, _sz, _i64, _i32, C.c_int, _p]),
    'sg_pixel_norm_act_bwd_pw_wg_workspace': (_sz, [_i32, _i32]),
    'sg_upscale2x': (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _i32, _f, C.c_int, _p]),
    'sg_upscale2x_masked': (C.c_int, [_p, _p, _p, _f, _i32, _i32, _i32, _i32, _i32, _f, C.c_int, _p]),
    'sg_downscale2x': (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _i32, _f, C.c_int, _p]),
    'sg_upscale_nn': (C.c_int, [_p, _p, _p, _f, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f, C.c_int, _p]),
    'sg_upscale_nn_planes': (C.c_int, [_p, _p, _p, _f, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f, _i32, C.c_int, _p]),
    'sg_downscale_sum': (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f, C.c_int, _p]),
    'sg_downscale_sum_masked': (C.c_int, [_p, _p, _f, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f, C.c_int, _p]),
    'sg_trilinear_up2x': (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, C.c_int, _p]),
    'sg_axpby': (C.c_int, [_p, _p, _p, _f, _f, _i64, C.c_int, _p]),
    'sg_axpby_dev': (C.c_int, [_p, _p, _p, _p, _i64, C.c_int, _p]),
    'sg_lerp_rows': (C.c_int, [_p, _p, _p, _p, _i32, _i64, C.c_int, _p]),
    'sg_add_noise': (C.c_int, [_p, _p, _f, _u64, _u64, _i64, C.c_int, _p]),
    'sg_add_noise_dev': (C.c_int, [_p, _p, _f, _u64, _p, _u64, _i64, C.c_int, _p]),
    'sg_sumsq_ndhwc_keep_w': (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _i32, C.c_int, _p]),
    'sg_minibatch_stddev_fwd': (C.c_int, [_p, _p, _p, _i32, _i64, _i32, _i32, C.c_int, _p]),
    'sg_minibatch_stddev_bwd': (C.c_int, [_p, _p, _p, _p, _i32, _i64, _i32, _i32, C.c_int, _p]),
    'sg_cast': (C.c_int, [_p, C.c_int, _p, C.c_int, _i64, _p]),
    'sg_adam_ema': (C.c_int, [_p, _p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _f, _p]),
    'sg_optim_step': (C.c_int, [C.c_int, _p, _p, _p, _p, _p, _i64, _f, _f, _f, C.c_int, _f, _f, _p]),
    'sg_adam_ema_dev': (C.c_int, [_p, _p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _f, _p]),
    'sg_optim_step_dev': (C.c_int, [C.c_int, _p, _p, _p, _p, _p, _i64, _p, _f, _f, C.c_int, _f, _f, _p]),
    'sg_segment_sumsq': (C.c_int, [_p, _p, _p, _i32, _p]),
    'sg_filter_axis': (C.c_int, [_p, _p, _p, _i64, _i32, _i64, C.POINTER(C.c_double), _i32, _i32, _i32, C.c_double, _i32, _p]),
    'sg_swd_gather': (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'sg_desc_normalize_workspace': (_sz, [_i32]),
    'sg_desc_normalize': (C.c_int, [_p, _i64, _i32, _i64, _p, _sz, _p]),
    'sg_swd_padded_rows': (_i32, [_i32]),
    'sg_swd_project': (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    'sg_sort_rows': (C.c_int, [_p, _i32, _i32, _p]),
    'sg_swd_distance': (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p]),
    'sg_metric_workspace': (_sz, []),
    'sg_sqdiff_mean': (C.c_int, [_p, _p, _p, _i64, _p, _sz, _p]),
    'sg_minmax': (C.c_int, [_p, _p, _i64, _p, _sz, _p]),
    'sg_ssim_products': (C.c_int, [_p, _p, _p, _p, _p, _i64, _p]),
    'sg_ssim_mean': (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, C.c_double, C.c_double, C.c_double, _p, _sz, _p]),
    'sg_prof_enable': (C.c_int, [C.c_int]),
    'sg_prof_enabled': (C.c_int, []),
    'sg_prof_collect': (C.c_int, [C.POINTER(ProfEntry), _i32, C.POINTER(_i32)]),
    'sg_prof_set_filter': (C.c_int, [C.c_int, _SHP]),
    'sg_config_reload': (C.c_int, []),
}

_lib = None


def load():
    """Loads the shared library (once) and declares every prototype.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} not found: build it with `python -m saragan_amd.build` (hipcc, gfx950). '
            'There is no CPU or PyTorch fallback for the HIP path.')
    # torch first: it ships its own libamdhip64, and the library must bind to THAT runtime (device memory and
    # streams come from torch).  Loaded the other way round, the system ROCm runtime initialises first and the
    # process ends up with two HIP runtimes, one of which sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class SgError(RuntimeError):
    pass


def check(rc, what=''):
    if rc != 0:
        msg = load().sg_error_string(int(rc)).decode()
        raise SgError(f'{what}: {msg} (code {rc})')

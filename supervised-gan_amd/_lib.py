"""ctypes binding of libsgan_hip.so (include/sgan_hip.h).  There is NO fallback: if the library is
missing or a call fails, this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGAN_HIP_LIB: diagnostics only (e.g. an -DSG_ABLATE build of the same sources)
LIB_PATH = os.environ.get("SGAN_HIP_LIB") or os.path.join(_HERE, "csrc", "libsgan_hip.so")

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
MATH_F32, MATH_BF16X3 = 0, 1
CONV, CONVT = 0, 1


class NormDesc(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("count", C.c_int32), ("eps", C.c_float), ("act", C.c_int32), ("slope", C.c_float), ("sq_stride", C.c_int32),
                ("rep_stride", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32),
                ("Hout", C.c_int32), ("Wout", C.c_int32), ("Cout", C.c_int32),
                ("Cin_logical", C.c_int32), ("Cout_logical", C.c_int32), ("math", C.c_int32)]


class GaussJob(C.Structure):
    _fields_ = [("image", C.c_void_p), ("image_ld", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("down", C.c_void_p), ("down_ld", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("g", C.c_void_p), ("g_chan_stride", C.c_int32), ("k", C.c_int32), ("pad", C.c_int32), ("s", C.c_int32)]


class ConvFwdJob(C.Structure):
    _fields_ = [("d", C.POINTER(ConvDesc)), ("inp", C.c_void_p), ("in_ld", C.c_int32), ("in_norm", C.POINTER(NormDesc)),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p), ("out_ld", C.c_int32), ("out_stats", C.c_void_p),
                ("out_stats_sq_stride", C.c_int32), ("w_packed", C.c_void_p), ("out_stats_rep_stride", C.c_int32)]


class ConvDgradJob(C.Structure):
    _fields_ = [("d", C.POINTER(ConvDesc)), ("dout", C.c_void_p), ("dout_ld", C.c_int32), ("w", C.c_void_p),
                ("din", C.c_void_p), ("din_ld", C.c_int32), ("x", C.c_void_p), ("x_ld", C.c_int32),
                ("x_norm", C.POINTER(NormDesc)), ("bwd_sums", C.c_void_p), ("bwd_sums_sq_stride", C.c_int32),
                ("accumulate", C.c_int32), ("w_transposed", C.c_int32), ("w_packed", C.c_void_p), ("bwd_sums_rep_stride", C.c_int32),
                ("dout_amax", C.c_void_p), ("w_packed_f16", C.c_void_p)]


class WtSeg(C.Structure):
    _fields_ = [("off", C.c_int64), ("taps", C.c_int32), ("cout", C.c_int32), ("cin", C.c_int32)]


class ConvWgradJob(C.Structure):
    _fields_ = [("d", C.POINTER(ConvDesc)), ("inp", C.c_void_p), ("in_ld", C.c_int32), ("in_norm", C.POINTER(NormDesc)),
                ("dout", C.c_void_p), ("dout_ld", C.c_int32), ("dw", C.c_void_p), ("dbias", C.c_void_p), ("dout_amax", C.c_void_p)]


class NormBwdJob(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("dy_ld", C.c_int32), ("x", C.c_void_p), ("x_ld", C.c_int32), ("npix", C.c_int32),
                ("C", C.c_int32), ("x_norm", C.POINTER(NormDesc)), ("bwd_sums", C.c_void_p), ("bwd_sums_sq_stride", C.c_int32),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("bwd_sums_rep_stride", C.c_int32), ("amax_out", C.c_void_p)]


class GanLossJob(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("ld", C.c_int32), ("npix", C.c_int32), ("target", C.c_float), ("weight", C.c_float),
                ("dlogits", C.c_void_p), ("dld", C.c_int32)]


class BnRunningDesc(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p), ("C", C.c_int32), ("count", C.c_int32), ("sq_stride", C.c_int32),
                ("rep_stride", C.c_int32)]


class AdamSeg(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64)]


# name -> argtypes; every entry returns int except the two string getters
_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SIGNATURES = {
    "sgan_conv_fwd": [C.POINTER(ConvDesc), _P, _I, C.POINTER(NormDesc), _P, _P, _P, _I, _I, _P, _P, _L, _P],
    "sgan_conv_dgrad": [C.POINTER(ConvDesc), _P, _I, _P, _P, _I, _P, _I, C.POINTER(NormDesc), _P, _P, _L, _P],
    "sgan_conv_wgrad": [C.POINTER(ConvDesc), _P, _I, C.POINTER(NormDesc), _P, _I, _P, _P, _P, _L, _P],
    "sgan_conv_fwd_grouped": [C.POINTER(ConvFwdJob), _I, _I, _P, _L, _P],
    "sgan_conv_dgrad_grouped": [C.POINTER(ConvDgradJob), _I, _P, _L, _P],
    "sgan_conv_wgrad_grouped": [C.POINTER(ConvWgradJob), _I, _P, _L, _P],
    "sgan_conv_bwd_fused": [C.POINTER(ConvDgradJob), _I, C.POINTER(ConvWgradJob), _I, _I, _P],
    "sgan_conv_bwd_thin_pair": [C.POINTER(ConvDgradJob), _I, C.POINTER(ConvWgradJob), _I, _P],
    "sgan_conv_bwd_fused_ws": [C.POINTER(ConvDgradJob), _I, C.POINTER(ConvWgradJob), _I, _I, _P, C.c_int64, _P],
    "sgan_norm_bwd_apply": [_P, _I, _P, _I, _I, _I, C.POINTER(NormDesc), _P, _I, _P, _P, _P],
    "sgan_norm_bwd_apply_multi": [C.POINTER(NormBwdJob), _I, _P],
    "sgan_transpose_weights": [_P, _P, C.POINTER(WtSeg), _I, _P],
    "sgan_pack_weights": [_P, _P, _P, _P, _P, C.POINTER(WtSeg), _I, _P],
    "sgan_bce01_fwd": [_P, _I, _P, _I, _I, _I, _P, _P, _I, _P, C.c_int64, _P],
    "sgan_bilinear_up2_fwd": [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P],
    "sgan_bilinear_up2_bwd": [_P, _I, _I, _I, _I, _P, _I, _P],
    "sgan_avgpool_pyramid_fwd": [_P, _I, _I, _I, _P, _P, _P],
    "sgan_avgpool_pyramid_bwd": [_P, _P, _I, _I, _P, _I, _I, _P],
    "sgan_norm_apply_fwd": [_P, _I, C.POINTER(NormDesc), _P, _P, _F, _P, _I, _I, _I, _P],
    "sgan_norm_apply_bwd_sums": [_P, _I, _P, _P, _I, C.POINTER(NormDesc), _P, _I, _I, _P],
    "sgan_pad_reflect_fwd": [_P, _I, _I, _I, _I, C.POINTER(NormDesc), _P, _I, _P, _I, _P],
    "sgan_pad_reflect_bwd": [_P, _I, _I, _I, _I, _I, _P, _I, C.POINTER(NormDesc), _P, _P, _I, _P, _I, _P],
    "sgan_dropout_mask": [_P, _L, _F, C.c_uint64, _P, _I, _P],
    "sgan_rng_advance": [_P, C.c_uint64, _P],
    "sgan_l1w_fwd": [_P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _F, _P, _P, _I, _P, C.c_int64, _P],
    "sgan_scale": [_P, _P, _P, _L, _P],
    "sgan_bn_running_update": [C.POINTER(BnRunningDesc), _I, _F, _P],
    "sgan_image_prep": [_P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "sgan_image_resize": [_P, _I, _I, _I, _P, _I, _I, _I, _P, _L, _P],
    "sgan_image_resize_workspace": [_I, _I, _I, _I, _I, _I],
    "sgan_gauss_down_fwd": [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    "sgan_gauss_down_multi_fwd": [C.POINTER(GaussJob), _I, _I, _I, _P],
    "sgan_gauss_down_multi_bwd": [C.POINTER(GaussJob), _I, _I, _I, _I, _P],
    "sgan_gauss_down_bwd": [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P],
    "sgan_gan_loss_fwd": [_P, _I, _I, _F, _I, _P, _P, _P],
    "sgan_gan_loss_bwd": [_P, _I, _I, _F, _I, _P, _P, _I, _P],
    "sgan_gan_loss_multi_fwd": [C.POINTER(GanLossJob), _I, _I, _P, _P, _P, C.c_int64, _P],
    "sgan_gan_loss_multi_bwd": [C.POINTER(GanLossJob), _I, _I, _P, _P],
    "sgan_sigmoid_fwd": [_P, _I, _I, _P, _I, _P],
    "sgan_sigmoid_bwd": [_P, _I, _P, _I, _I, _P, _I, _P],
    "sgan_tanh_bwd": [_P, _P, _P, _L, _P],
    "sgan_stat_replicas": [],
    "sgan_set_device": [_I],
    "sgan_stream_device": [_P, C.POINTER(C.c_int32)],
    "sgan_add_act_fwd": [_P, _P, _P, _L, _I, _P],
    "sgan_to_nhwc": [_P, _L, _L, _L, _I, _I, _I, _P, _I, _I, _P],
    "sgan_concat_nhwc": [_P, _I, _I, _P, _I, _I, _L, _P, _I, _I, _P],
    "sgan_slice_nhwc": [_P, _I, _I, _I, _L, _P, _I, _I, _P],
    "sgan_adam_multi": [C.POINTER(AdamSeg), _I, _P, _F, _F, _F, _P, _P],
    "sgan_sgd_multi": [C.POINTER(AdamSeg), _I, _P, _F, _P],
    "sgan_adam_pack": [_P, _P, _P, _P, _L, _P, _F, _F, _F, _P, _P, _P, _P, _P, C.POINTER(WtSeg), _I, _I, _P],
    "sgan_zero_multi": [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), _I, _P],
    "sgan_conv_head_bwd": [C.POINTER(ConvDgradJob), C.POINTER(ConvWgradJob), _I, _P],
    "sgan_ce_fwd": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P],
    "sgan_ce_bwd": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P],
    "sgan_softmax_fwd": [_P, _I, _I, _I, _P, _I, _P],
    "sgan_softmax_bwd": [_P, _I, _P, _I, _I, _I, _P, _I, _P],
    "sgan_normal_fill": [_P, _L, C.c_uint64, _P, _I, _P],
    "sgan_normal_fill_nhwc": [_P, _I, _I, _I, _I, C.c_uint64, _P, _I, _P],
    "sgan_normal_fill_nhwc_pair": [_P, _P, _I, _I, _I, _I, C.c_uint64, _P, _P, C.c_int64, _P],
    "sgan_profile_enable": [_I],
    "sgan_profile_count": [],
    "sgan_profile_mark": [_P],
    "sgan_profile_read": [_I, C.POINTER(C.c_char_p), C.POINTER(C.c_float)],
}

RESTYPES = {"sgan_image_resize_workspace": C.c_int64}      # everything else returns an int status
_lib = None


class SganError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes library; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SganError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C supervised-gan_amd/csrc`.  There is no CPU/PyTorch fallback for this path.")
        # libsgan_hip.so needs libamdhip64.so.N.  PyTorch-ROCm ships its own copy (and its own HSA runtime); whichever copy is
        # mapped first serves the whole process.  If the system copy came first, torch's HSA runtime and the system HIP runtime
        # would be mixed and the first kernel launch fails with "no ROCm-capable device is detected" -- so torch goes first.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = RESTYPES.get(name, C.c_int)
        l.sgan_version.restype = C.c_char_p
        l.sgan_last_error.restype = C.c_char_p
        l.sgan_last_kernel.restype = C.c_char_p
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise SganError(f"{what} failed ({rc}): {lib().sgan_last_error().decode()}")

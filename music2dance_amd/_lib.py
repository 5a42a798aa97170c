"""ctypes binding of libm2d_hip.so (include/m2d.h).

The library is the product: there is no CPU or eager-PyTorch fallback. If it has not been
built, `lib()` raises; if a tensor handed to a kernel wrapper is not a contiguous fp32
tensor on a HIP device, the wrapper raises.
"""
import ctypes
import os

_c = ctypes
_F = _c.c_void_p   # const float* / float*  (device pointers travel as integers)
_I = _c.c_int
_S = _c.c_size_t
_f = _c.c_float
_L = _c.c_long

LIB_PATH = os.environ.get("M2D_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                      "libm2d_hip.so")

class AdamItem(_c.Structure):
    """struct M2dAdamItem of include/m2d.h (one tensor of a multi-tensor Adam step)"""
    _fields_ = [("param", _c.c_void_p), ("grad", _c.c_void_p), ("exp_avg", _c.c_void_p), ("exp_avg_sq", _c.c_void_p),
                ("numel", _c.c_longlong), ("pack_fwd", _c.c_void_p), ("pack_bwd", _c.c_void_p),
                ("cout", _I), ("cin", _I), ("ks", _I), ("reserved", _I)]


# name -> (restype, argtypes); mirrors include/m2d.h line by line
SIGNATURES = {
    "m2d_last_error": (_c.c_char_p, []),
    "m2d_version": (_I, []),
    "m2d_prof_begin": (_I, []),
    "m2d_prof_end": (_I, [_c.POINTER(_c.c_double), _I]),
    "m2d_prof_dump": (_I, [_c.c_char_p, _I]),
    "m2d_plan_cache_size": (_I, []),
    "m2d_plan_model_set": (_I, [_I]),
    "m2d_plan_model_get": (_I, []),
    "m2d_conv1d_fwd": (_I, [_F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _I, _I, _f, _F, _F, _f, _F, _F, _S, _F]),
    "m2d_conv1d_bwd_data": (_I, [_F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _I, _F, _f, _F, _f, _F, _S, _F]),
    "m2d_conv1d_pack_weights": (_I, [_F, _F, _F, _I, _I, _I, _F]),
    "m2d_conv1d_bwd_weight": (_I, [_F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _I, _F, _f, _F, _S, _F]),
    "m2d_conv1d_fwd_sum": (_I, [_F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _I, _I, _f, _F, _F, _f, _F, _S, _F]),
    "m2d_conv1d_k4_applicable": (_I, [_I, _I, _I, _I, _I, _I]),
    "m2d_conv1d_k4_packed_elems": (_S, [_I, _I, _I, _I]),
    "m2d_conv1d_pack_weights_k4": (_I, [_F, _F, _I, _I, _I, _I, _F]),
    "m2d_conv1d_fwd_k4": (_I, [_F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _I, _I, _f, _F, _F, _f, _F, _F, _S, _F]),
    "m2d_conv1d_bwd_data_res": (_I, [_F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _I, _F, _f, _F, _F, _f, _F, _S, _F]),
    "m2d_bn_update_running": (_I, [_F, _c.c_double, _F, _F, _F, _I, _f, _f, _F]),
    "m2d_bn_fwd_sums_pool_to": (_I, [_F, _F, _c.c_double, _F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _f, _f, _I, _f,
                                    _c.c_longlong, _F]),
    "m2d_bn_fwd_sums_upsample2_to": (_I, [_F, _F, _c.c_double, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _f, _f, _I, _f,
                                         _c.c_longlong, _F]),
    "m2d_conv1d_bwd_data_shared_mask": (_I, [_F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _I, _F, _f, _I, _F, _S, _F]),
    "m2d_conv1d_bwd_weight_from": (_I, [_F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _I, _F, _f, _I, _F, _S, _F]),
    "m2d_gemm_ld": (_I, [_I, _F, _I, _F, _I, _F, _F, _I, _I, _I, _I, _I, _f, _F, _f, _F, _f, _F, _S, _F]),
    "m2d_tanh_fwd": (_I, [_F, _F, _S, _F]),
    "m2d_tanh_bwd": (_I, [_F, _F, _F, _S, _F]),
    "m2d_tanh_bwd_bwd": (_I, [_F, _F, _F, _F, _S, _F]),
    "m2d_pose_pack3": (_I, [_F, _F, _F, _F, _I, _I, _I, _F]),
    "m2d_wgan_critic_loss": (_I, [_F, _I, _F, _F, _f, _F, _F]),
    "m2d_conv1d_fwd_windows": (_I, [_F, _I, _I, _I, _I, _I, _F, _F, _F, _I, _I, _I, _I, _I, _f, _F, _F, _S, _F]),
    "m2d_conv1d_bwd_weight_windows": (_I, [_F, _I, _I, _I, _I, _I, _F, _F, _F, _I, _I, _I, _I, _F, _f, _F, _S, _F]),
    "m2d_conv1d_workspace_bytes": (_S, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "m2d_gemm": (_I, [_I, _F, _F, _F, _F, _I, _I, _I, _I, _f, _F, _f, _F, _f, _F, _S, _F]),
    "m2d_gemm_workspace_bytes": (_S, [_I, _I, _I, _I]),
    "m2d_bn_workspace_bytes": (_S, [_I]),
    "m2d_stream_scratch_set": (_I, [_F, _F, _S]),
    "m2d_stream_create": (_I, [ctypes.POINTER(ctypes.c_void_p)]),
    "m2d_bn_scratch_bytes": (_S, [_I]),
    "m2d_bn_fwd": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _f, _f, _I, _I, _f, _F, _F, _S, _F, _F]),
    "m2d_bn_bwd": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _f, _F, _S, _F, _F]),
    "m2d_bn_stats": (_I, [_F, _F, _I, _I, _I, _F, _F]),
    "m2d_bn_fwd_sums": (_I, [_F, _F, _c.c_double, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _f, _f, _I, _f, _F, _F]),
    "m2d_bn_fwd_to": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _f, _f, _I, _I, _f, _F, _F, _S, _F, _c.c_longlong, _F]),
    "m2d_bn_fwd_sums_to": (_I, [_F, _F, _c.c_double, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _f, _f, _I, _f, _F,
                                _c.c_longlong, _F]),
    "m2d_bn_bwd_stats": (_I, [_F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _f, _F, _F]),
    "m2d_bn_bwd_sums": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _c.c_double, _F, _F, _F, _I, _I, _I, _I, _f, _F, _S, _F]),
    "m2d_channel_sums": (_I, [_F, _F, _f, _F, _I, _I, _I, _F, _S, _F, _F]),
    "m2d_gru_layer_fwd": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _F]),
    "m2d_gru_layer_bwd": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _F]),
    "m2d_gru_stack_counters": (_I, [_I, _I]),
    "m2d_gru_stack_fwd": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _F, _F]),
    "m2d_gru_persist_error": (_I, []),
    "m2d_gru_persist_peek": (_I, []),
    "m2d_async_fault_word": (_c.c_void_p, []),
    "m2d_gru_persist_raise": (_I, []),
    "m2d_fault_fetch": (_I, [_F, _F]),
    "m2d_gru_stack_bwd": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _F, _F]),
    "m2d_gp_interpolate": (_I, [_F, _F, _F, _F, _I, _I, _F]),
    "m2d_gp_penalty_workspace_bytes": (_S, [_I]),
    "m2d_gp_penalty_fwd": (_I, [_F, _F, _F, _I, _I, _I, _F, _S, _F]),
    "m2d_gp_penalty_bwd": (_I, [_F, _F, _F, _F, _I, _I, _I, _F]),
    "m2d_reduce_workspace_bytes": (_S, []),
    "m2d_l1_mean_fwd": (_I, [_F, _F, _F, _S, _F, _S, _F]),
    "m2d_l1_mean_bwd": (_I, [_F, _F, _F, _F, _S, _F]),
    "m2d_tv_mean_fwd": (_I, [_F, _F, _I, _I, _I, _L, _L, _L, _F, _S, _F]),
    "m2d_tv_mean_bwd": (_I, [_F, _F, _F, _I, _I, _I, _L, _L, _L, _F]),
    "m2d_jerk_mean_fwd": (_I, [_F, _F, _I, _I, _I, _L, _L, _L, _F, _S, _F]),
    "m2d_affine_cols": (_I, [_F, _F, _F, _F, _S, _I, _F]),
    "m2d_adam_multi": (_I, [_F, _I, _f, _f, _f, _f, _f, _f, _F, _F]),
    "m2d_maxpool2_fwd": (_I, [_F, _F, _S, _I, _F]),
    "m2d_maxpool2_bwd": (_I, [_F, _F, _F, _S, _I, _F]),
    "m2d_upsample2_fwd": (_I, [_F, _F, _S, _I, _F]),
    "m2d_upsample2_fwd_to": (_I, [_F, _F, _S, _I, _I, _c.c_longlong, _F]),
    "m2d_maxpool2_fwd_from": (_I, [_F, _F, _S, _I, _I, _c.c_longlong, _F]),
    "m2d_upsample2_bwd": (_I, [_F, _F, _S, _I, _F]),
}

_lib = None


class M2dError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise loudly when the .so is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise M2dError(
            "libm2d_hip.so is not built (%s). Run `python -m music2dance_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    # PyTorch first: its bundled HIP runtime must be the one this library binds to. Loaded the other way round the
    # process holds two HIP runtimes (/opt/rocm's through this library, torch's own) and every launch through this
    # library fails with "no ROCm-capable device is detected" (seen when build() and smoke() share a process).
    import torch  # noqa: F401
    h = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name)  # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = h
    return _lib


def install_backend(handle):
    """Test hook: replace the library handle (tests/fake_backend.py). Never used by the product."""
    global _lib
    prev = _lib
    _lib = handle
    return prev


def check(rc, what):
    if rc != 0:
        msg = lib().m2d_last_error()
        raise M2dError("%s failed (%d): %s" % (what, rc, msg.decode() if isinstance(msg, bytes) else msg))

"""ctypes binding of libchexpert_hip.so (include/chexpert_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  `lib()` raises if the shared object
is missing, and every wrapper raises `RuntimeError` on a non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libchexpert_hip.so")

PRO_NONE, PRO_AFFINE_RELU, PRO_AFFINE2, PRO_JOIN = 0, 1, 2, 3
MODE_CONV, MODE_POOL2, MODE_STEM = 0, 1, 2
EPI_STORE, EPI_MASK, EPI_JOIN = 0, 1, 2

_vp, _fp, _i32 = C.c_void_p, C.c_void_p, C.c_int32


class CxConv(C.Structure):
    _fields_ = [("x", _vp), ("x2", _vp), ("w", _vp), ("y", _vp),
                ("pa", _fp), ("pb", _fp), ("pc", _fp),
                ("stat_sum", _fp), ("stat_sq", _fp),
                ("ex", _vp), ("e_sc", _fp), ("e_sh", _fp), ("e_mu", _fp), ("e_r", _fp), ("e_scale", _fp),
                ("B", _i32), ("H", _i32), ("W", _i32), ("Ho", _i32), ("Wo", _i32),
                ("K", _i32), ("N", _i32),
                ("ldx", _i32), ("ldx2", _i32), ("ldy", _i32), ("ldex", _i32),
                ("kh", _i32), ("kw", _i32), ("stride", _i32), ("pad", _i32),
                ("prologue", _i32), ("mode", _i32), ("epilogue", _i32), ("accumulate", _i32), ("tstride", _i32),
                ("stat_replicas", _i32), ("stat_rstride", _i32), ("stat_det", _i32), ("dtype", _i32),
                ("pro_out", _vp), ("ldpo", _i32), ("dil", _i32), ("emask", _vp),
                ("x3", _vp), ("po_lo", _vp), ("po_mask", _vp), ("kernel_hint", _i32), ("pad2_", _i32)]


class CxWgrad(C.Structure):
    _fields_ = [("g", _vp), ("g2", _vp), ("x", _vp), ("dw", _fp),
                ("ga", _fp), ("gb", _fp), ("gc", _fp), ("pa", _fp), ("pb", _fp),
                ("B", _i32), ("H", _i32), ("W", _i32), ("Ho", _i32), ("Wo", _i32), ("K", _i32), ("N", _i32),
                ("ldg", _i32), ("ldg2", _i32), ("ldx", _i32),
                ("kh", _i32), ("kw", _i32), ("stride", _i32), ("pad", _i32),
                ("g_prologue", _i32), ("x_prologue", _i32), ("mode", _i32), ("splits", _i32), ("dtype", _i32), ("dil", _i32),
                ("scratch", _fp), ("scratch_floats", C.c_int64), ("kernel_hint", _i32), ("pad_", _i32)]


WGRAD_BATCH_MAX = 24


class CxWgradBatch(C.Structure):
    _fields_ = [("g", _vp * WGRAD_BATCH_MAX), ("x", _vp * WGRAD_BATCH_MAX), ("pa", _fp * WGRAD_BATCH_MAX), ("pb", _fp * WGRAD_BATCH_MAX),
                ("dw", _fp * WGRAD_BATCH_MAX), ("n", _i32), ("pad_", _i32)]


class CxChanMapDesc(C.Structure):
    _fields_ = [("real_off", C.c_int64), ("pad_off", C.c_int64), ("O", _i32), ("taps", _i32), ("Ireal", _i32), ("Ipad", _i32),
                ("c0r", _i32), ("c0p", _i32), ("k", _i32), ("kp", _i32), ("split", _i32), ("shift", _i32)]


class CxPackDesc(C.Structure):
    _fields_ = [("src_off", C.c_int64), ("dst_off", C.c_int64), ("O", _i32), ("I", _i32), ("kh", _i32), ("kw", _i32),
                ("transpose", _i32), ("stem", _i32)]


class CxReduceDesc(C.Structure):
    _fields_ = [("dw", _fp), ("slab", _fp), ("total", C.c_int64), ("splits", _i32), ("vec", _i32), ("first_block", _i32), ("pad_", _i32),
                ("cols", _i32), ("dw_ld", _i32), ("pad2_", C.c_int64)]


# name -> argtypes (return type is int unless noted); kept in one table so the symbol-export test can
# check it against include/chexpert_hip.h
_f, _sz, _i = C.c_float, C.c_size_t, C.c_int
SIGNATURES = {
    "cx_abi_version": [],
    "cx_last_pro_out": [],
    "cx_last_kernel": [],
    "cx_wgrad_defer": [C.c_int],
    "cx_wgrad_defer_take": [C.POINTER(CxReduceDesc), C.c_int, C.POINTER(C.c_int64)],
    "cx_last_slab_floats": [],
    "cx_dw_reduce_table": [_vp, C.c_int, C.c_int64, _vp],
    "cx_error_string": [_i],
    "cx_conv_gemm": [C.POINTER(CxConv), _vp],
    "cx_conv3x3_wgrad_batch": [C.POINTER(CxWgrad), C.POINTER(CxWgradBatch), _vp],
    "cx_chan_map_table": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "cx_conv_wgrad": [C.POINTER(CxWgrad), _vp],
    "cx_conv1x1_dgrad_wgrad": [C.POINTER(CxConv), _vp, _vp],
    "cx_conv1x1_dgrad_wgrad_ws": [C.POINTER(CxConv), _vp, _vp, C.c_int64, _vp],
    "cx_conv1x1_dgrad_wgrad_ld_ws": [C.POINTER(CxConv), _vp, _i, _vp, C.c_int64, _vp],
    "cx_conv1x1_dgrad_wgrad_pair_ws": [C.POINTER(CxConv), C.POINTER(CxConv), _vp, _i, _vp, _vp, C.c_int64, _vp],
    "cx_pack_weights": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "cx_pack_weights_table": [_vp, _vp, _vp, _i, _vp],
    "cx_nchw3_to_nhwc4": [_vp, _vp, _i, _i, _i, _vp],
    "cx_u8_to_nhwc4": [_vp, _vp, _sz, _f, _f, _vp],
    "cx_bn_coef": [_vp, _vp, _f, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "cx_bn_coef_moments": [_vp, _vp, _f, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "cx_last_stat_rows": [],
    "cx_bnrelu_maxpool_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "cx_bnrelu_maxpool_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cx_head_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "cx_gap_relu_bn_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "cx_unpool2_mask_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cx_pack_weights_table_f32": [_vp, _vp, _vp, _i, _vp],
    "cx_nchw3_to_nhwc4_f32": [_vp, _vp, _i, _i, _i, _vp],
    "cx_u8_to_nhwc4_f32": [_vp, _vp, _sz, _f, _f, _vp],
    "cx_u8_jitter": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "cx_bn_coef_eval": [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _vp],
    "cx_bn_bwd_coef": [_vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp],
    "cx_bn_bwd_slice_coef": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "cx_bnrelu_maxpool_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "cx_bnrelu_maxpool_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cx_head_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "cx_bce_fwd_bwd": [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp],
    "cx_softmax_ce_fwd_bwd": [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp],
    "cx_head_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "cx_gap_relu_bn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "cx_unpool2_mask": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cx_affine2_inplace": [_vp, _vp, _vp, _vp, _vp, _sz, _i, _vp],
    "cx_affine2_relu": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp],
    "cx_relu_bwd_stats": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp],
    "cx_dropout_slice_fwd": [_vp, _i, C.c_int64, _i, C.c_float, _vp, C.c_uint32, _vp, _vp, _i, _vp],
    "cx_dropout_slice_fwd_f32": [_vp, _i, C.c_int64, _i, C.c_float, _vp, C.c_uint32, _vp, _vp, _i, _vp],
    "cx_dropout_slice_bwd": [_vp, _i, _vp, _i, _vp, _vp, _vp, C.c_int64, _i, C.c_float, _vp, C.c_uint32, _vp],
    "cx_dropout_slice_bwd_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, C.c_int64, _i, C.c_float, _vp, C.c_uint32, _vp],
    "cx_join_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp],
    "cx_affine2_relu_mask": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp],
    "cx_affine2_relu_mask_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp],
    "cx_relu_bwd_stats_mask": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp],
    "cx_relu_bwd_stats_mask_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp],
    "cx_adam_step": [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _f, _vp],
    "cx_sgd_nesterov_step": [_vp, _vp, _vp, _sz, _f, _f, _f, _i, _f, _vp],
    "cx_rmsprop_step": [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _vp],
    "cx_adam_step_dev": [_vp, _vp, _vp, _vp, _sz, _vp, _f, _f, _f, _f, _f, _vp],
    "cx_sgd_nesterov_step_dev": [_vp, _vp, _vp, _sz, _vp, _f, _f, _f, _vp],
    "cx_rmsprop_step_dev": [_vp, _vp, _vp, _vp, _sz, _vp, _f, _f, _f, _f, _f, _vp],
    "cx_optim_tick": [_vp, _vp],
    "cx_aa_attention_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cx_aa_attention_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cx_aa_attention_weights": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cx_aa_attention_weights_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cx_aa_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, C.c_int64, _vp],
    "cx_aa_attention_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, C.c_int64, _vp],
    "cx_aa_outproj_fwd": [_vp, _vp, _vp, _i, _vp, _vp, _sz, _i, _i, _i, _vp],
    "cx_aa_outproj_fwd_f32": [_vp, _vp, _vp, _i, _vp, _vp, _sz, _i, _i, _i, _vp],
    "cx_aa_outproj_bwd": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp, C.c_int64, _vp],
    "cx_aa_outproj_bwd_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp, C.c_int64, _vp],
    "cx_rows_reduce": [_vp, _vp, _i, _i, _i, _i, _vp],
    "cx_stats_bc": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "cx_stats_bc_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "cx_affine_relu_bc": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "cx_affine_relu_bc_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "cx_in_relu_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "cx_in_relu_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "cx_f32_to_bf16": [_vp, _vp, _sz, _vp],
    "cx_nchw3_to_nhwc8": [_vp, _vp, _i, _i, _i, _vp],
    "cx_nchw3_to_nhwc8_f32": [_vp, _vp, _i, _i, _i, _vp],
    "cx_u8_to_nhwc8": [_vp, _vp, _sz, _f, _f, _vp],
    "cx_u8_to_nhwc8_f32": [_vp, _vp, _sz, _f, _f, _vp],
    "cx_dwconv_fwd": [_vp] * 7 + [_i] * 8 + [_vp],
    "cx_dwconv_fwd_f32": [_vp] * 7 + [_i] * 8 + [_vp],
    "cx_dwconv_dgrad": [_vp] * 14 + [_i] * 9 + [_vp],
    "cx_dwconv_dgrad_f32": [_vp] * 14 + [_i] * 9 + [_vp],
    "cx_dwconv_wgrad": [_vp] * 9 + [_i] * 7 + [_vp, C.c_int64, _vp],
    "cx_dwconv_wgrad_f32": [_vp] * 9 + [_i] * 7 + [_vp, C.c_int64, _vp],
    "cx_gap_affine_act": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, C.c_int64, _vp],
    "cx_gap_affine_act_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, C.c_int64, _vp],
    "cx_se_fwd": [_vp] * 7 + [_i, _i, _i, _vp],
    "cx_se_bwd_fused": [_vp] * 15 + [_i, _i, _i, _i, _vp, C.c_int64, _vp, C.c_int64, _vp],
    "cx_se_bwd_fused_f32": [_vp] * 15 + [_i, _i, _i, _i, _vp, C.c_int64, _vp, C.c_int64, _vp],
    "cx_gap_se_fwd": [_vp] * 10 + [_i, _i, _i, _i, _i, _vp, C.c_int64, _vp],
    "cx_gap_se_fwd_f32": [_vp] * 10 + [_i, _i, _i, _i, _i, _vp, C.c_int64, _vp],
    "cx_se_bwd": [_vp] * 11 + [_i, _i, _i, _vp, C.c_int64, _vp],
    "cx_scale_act_bc": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "cx_scale_act_bc_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "cx_se_bwd_reduce": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, C.c_int64, _vp],
    "cx_se_bwd_reduce_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, C.c_int64, _vp],
    "cx_se_act_bwd": [_vp] * 11 + [_i, _i, _i, _i, _vp],
    "cx_se_act_bwd_f32": [_vp] * 11 + [_i, _i, _i, _i, _vp],
    "cx_bn_lin_bwd_stats": [_vp] * 6 + [_sz, _i, _i, _vp],
    "cx_bn_lin_bwd_stats_f32": [_vp] * 6 + [_sz, _i, _i, _vp],
    "cx_affine2_out": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _i, _vp],
    "cx_affine2_out_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _i, _vp],
    "cx_scale_rows": [_vp, _vp, _sz, _vp, _sz, _i, _vp],
    "cx_scale_rows_f32": [_vp, _vp, _sz, _vp, _sz, _i, _vp],
    "cx_dropout_mask": [_vp, _sz, _f, C.c_ulonglong, _vp],
    "cx_dropout_mask_dev": [_vp, _sz, _f, C.c_ulonglong, _vp, _vp],
    "cx_counter_add": [_vp, C.c_ulonglong, _vp],
    "cx_mul_f32": [_vp, _vp, _vp, _sz, _vp],
    "cx_linear_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "cx_gradcam_map": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "cx_cam_norm_upsample": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "cx_fill_f32": [_vp, _f, _sz, _vp],
    "cx_copy_stream": [_vp, _vp, _sz, _vp],
    "cx_affine_to_f32_nchw": [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "cx_bf16_to_f32_nchw": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
}

_lib = None


def lib():
    """The loaded library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "chexpert_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        # torch first: its wheel carries its own libamdhip64 / libhsa-runtime64.  Loaded after this library (whose DT_NEEDED then
        # resolves to /opt/rocm's copies) the process holds TWO HIP runtimes and every launch from here fails with "no
        # ROCm-capable device" -- build() followed by smoke() in one process did exactly that.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = C.c_char_p if name in ("cx_error_string", "cx_last_kernel") else C.c_int
        if l.cx_abi_version() != 10:
            raise RuntimeError("chexpert_amd: ABI version mismatch")
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (code %d)" % (what, lib().cx_error_string(rc).decode(), rc))


def ptr(t):
    """data pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("chexpert_amd kernels run on the GPU only (got a %s tensor); there is no CPU fallback"
                               % t.device)

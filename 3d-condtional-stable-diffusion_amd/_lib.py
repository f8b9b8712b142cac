"""ctypes binding of libdm3d_hip.so (C ABI declared in include/dm3d.h).

The product path has no fallback: if the shared library is missing or a call fails, this module raises.  Build it with
``__graft_entry__.build()`` or ``make -C 3d-condtional-stable-diffusion_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DM3D_LIB") or os.path.join(_HERE, "csrc", "libdm3d_hip.so")     # DM3D_LIB: A/B builds (tools)

ACT_NONE, ACT_RELU, ACT_SILU = 0, 1, 2
ABI_VERSION = 111                   # DM3D_VERSION of include/dm3d.h these ctypes mirrors were written against
PREC_F32, PREC_H3 = 0, 1
WL_TAP, WL_PAIR = 0, 1
FMT_F32, FMT_H2 = 0, 1
COUT_PAD, CIN_PAD = 64, 16

_f32p = C.c_void_p      # device pointers travel as integers
_i32p = C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [
        ("x1", _f32p), ("x2", _f32p), ("c1", C.c_int32), ("c2", C.c_int32), ("batch", C.c_int32),
        ("in_d", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32), ("upsample", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("wpk", _f32p), ("bias", _f32p),
        ("pro_scale", _f32p), ("pro_shift", _f32p), ("vec", _f32p), ("vec_idx", _i32p), ("vec_ld", C.c_int32),
        ("relu", C.c_int32), ("res", _f32p), ("out", _f32p), ("cout", C.c_int32),
        ("precision", C.c_int32), ("w_exp", C.c_int32),
        ("prelu_alpha", _f32p), ("relu_out", C.c_int32), ("transpose", C.c_int32),
        ("pro_batch_stride", C.c_int64), ("w_layout", C.c_int32),
        ("skip_x1", _f32p), ("skip_x2", _f32p), ("skip_c1", C.c_int32), ("skip_c2", C.c_int32), ("skip_wpk", C.c_void_p),
        ("x1_fmt", C.c_int32), ("out_fmt", C.c_int32), ("post_scale", _f32p), ("post_shift", _f32p),
        ("scratch", C.c_void_p), ("scratch_bytes", C.c_int64),
        ("range_flag", C.c_void_p), ("range_limit", C.c_float), ("wpk_wino", C.c_void_p), ("skip_wpk_frag", C.c_void_p), ("gn_stats", C.c_void_p),
        ("split_counters", C.c_void_p), ("split_counter_words", C.c_int32),
    ]


class GemmDesc(C.Structure):
    _fields_ = [
        ("a", _f32p), ("lda", C.c_int64), ("stride_a", C.c_int64),
        ("b", _f32p), ("ldb", C.c_int64), ("stride_b", C.c_int64),
        ("out", _f32p), ("ldo", C.c_int64), ("stride_o", C.c_int64),
        ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("batch", C.c_int32),
        ("alpha", C.c_float), ("bias", _f32p), ("bias_along_m", C.c_int32), ("act", C.c_int32),
        ("res", _f32p), ("ldr", C.c_int64), ("stride_r", C.c_int64),
        ("precision", C.c_int32), ("a_fmt", C.c_int32), ("b_fmt", C.c_int32), ("out_fmt", C.c_int32),
        ("res2", _f32p), ("range_flag", C.c_void_p), ("range_limit", C.c_float),
    ]


class MlpDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("w0", C.c_void_p), ("b0", _f32p),
        ("w1", C.c_void_p), ("b1", _f32p),
        ("res", _f32p), ("res2", _f32p), ("ldr", C.c_int64),
        ("out", C.c_void_p), ("ldo", C.c_int64), ("out_fmt", C.c_int32),
        ("m", C.c_int32), ("units", C.c_int32),
        ("range_flag", C.c_void_p), ("range_limit", C.c_float),
        ("w2", C.c_void_p), ("b2", _f32p), ("res3", _f32p), ("ldr3", C.c_int64),
    ]


class AttnFrontDesc(C.Structure):
    _fields_ = [
        ("x", _f32p), ("ldx", C.c_int64),
        ("w_in", C.c_void_p), ("b_in", _f32p),
        ("w_qk", C.c_void_p), ("b_qk", _f32p),
        ("w_v", C.c_void_p), ("b_v", _f32p),
        ("g1", _f32p), ("be1", _f32p), ("g2", _f32p), ("be2", _f32p), ("g3", _f32p), ("be3", _f32p),
        ("eps", C.c_float),
        ("y", _f32p), ("ldy", C.c_int64),
        ("qk", C.c_void_p), ("ldqk", C.c_int64),
        ("vt", C.c_void_p), ("ldvt", C.c_int64),
        ("q2", C.c_void_p), ("ldq2", C.c_int64),
        ("n3", C.c_void_p), ("ldn3", C.c_int64),
        ("m", C.c_int32), ("units", C.c_int32),
        ("range_flag", C.c_void_p), ("range_limit", C.c_float),
    ]


class AttentionDesc(C.Structure):
    _fields_ = [
        ("q", _f32p), ("ldq", C.c_int64),
        ("k", _f32p), ("ldk", C.c_int64), ("stride_k", C.c_int64),
        ("vt", _f32p), ("ldv", C.c_int64), ("stride_vt", C.c_int64),
        ("out", _f32p), ("ldo", C.c_int64), ("res", _f32p),
        ("batch", C.c_int32), ("lq", C.c_int32), ("lk", C.c_int32), ("c", C.c_int32),
        ("scale", C.c_float), ("precision", C.c_int32), ("fmt", C.c_int32),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("a", _f32p), ("g", _f32p), ("dw", _f32p),
        ("batch", C.c_int32), ("in_d", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
        ("ksize", C.c_int32), ("per_item_output", C.c_int32), ("stride_a", C.c_int64), ("stride_g", C.c_int64), ("stride_dw", C.c_int64),
    ]


class DdpmDesc(C.Structure):
    _fields_ = [
        ("x", _f32p), ("eps", _f32p), ("noise", _f32p), ("batch", C.c_int32), ("per_sample", C.c_int64),
        ("t", _i32p), ("timesteps", C.c_int32),
        ("beta", _f32p), ("sqrt_alpha", _f32p), ("alpha_bar", _f32p), ("alpha_bar_prev", _f32p),
        ("sqrt_alpha_bar", _f32p), ("sqrt_alpha_bar_prev", _f32p), ("sqrt_one_minus_alpha_bar", _f32p),
        ("seed", C.c_uint64), ("mode", C.c_int32), ("mean_out", _f32p), ("var_out", _f32p), ("seed_dev", C.c_void_p),
    ]


# name -> (restype, argtypes): every symbol include/dm3d.h declares
SIGNATURES = {
    "dm3d_version": (C.c_int, []),
    "dm3d_last_error": (C.c_char_p, []),
    "dm3d_device_ok": (C.c_int, []),
    "dm3d_packed_weight_elems": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "dm3d_pack_weights": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_int32, _f32p, _f32p, C.c_void_p]),
    "dm3d_packed_weight_h3_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "dm3d_pack_weights_h3": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p, C.c_void_p, C.c_void_p]),
    "dm3d_packed_weight_h3p_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "dm3d_pack_weights_h3p": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p, C.c_void_p, C.c_int32, C.c_void_p]),
    "dm3d_packed_weight_h3w_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "dm3d_pack_weights_h3w": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_int32, _f32p, C.c_void_p, C.c_void_p]),
    "dm3d_packed_weight_skip_h3p_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "dm3d_pack_weights_skip_h3p": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dm3d_pack_weights_skip_h3f": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dm3d_conv_scratch_bytes": (C.c_int64, [C.POINTER(ConvDesc)]),
    "dm3d_conv_split_counter_words": (C.c_int32, [C.POINTER(ConvDesc)]),
    "dm3d_conv_tile_form": (C.c_int32, [C.POINTER(ConvDesc)]),
    "dm3d_conv_weight_layout": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "dm3d_attention_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "dm3d_attention": (C.c_int, [C.POINTER(AttentionDesc), C.c_void_p, C.c_void_p]),
    "dm3d_attention_group": (C.c_int, [C.POINTER(AttentionDesc), C.c_int32, C.c_void_p, C.c_void_p]),
    "dm3d_packed_weight_up_elems": (C.c_int64, [C.c_int32, C.c_int32]),
    "dm3d_pack_weights_up": (C.c_int, [_f32p, C.c_int32, C.c_int32, _f32p, C.c_void_p]),
    "dm3d_packed_weight_up_h3_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "dm3d_pack_weights_up_h3": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dm3d_pack_weights_convt": (C.c_int, [_f32p, C.c_int32, C.c_int32, _f32p, C.c_void_p]),
    "dm3d_pack_weights_convt_h3": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dm3d_conv3d_ndhwc": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "dm3d_gemm_tn": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "dm3d_gemm_tn_group": (C.c_int, [C.POINTER(GemmDesc), C.c_int32, C.c_void_p]),
    "dm3d_mlp_fused": (C.c_int, [C.POINTER(MlpDesc), C.c_void_p]),
    "dm3d_pack_mlp_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dm3d_attn_front": (C.c_int, [C.POINTER(AttnFrontDesc), C.c_void_p]),
    "dm3d_pack_front_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dm3d_split_h2": (C.c_int, [_f32p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "dm3d_layernorm3_h2": (C.c_int, [_f32p, C.c_int64, C.c_int32, C.c_float] + [_f32p] * 9 + [C.c_void_p]),
    "dm3d_softmax_rows_h2": (C.c_int, [_f32p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p]),
    "dm3d_layernorm3": (C.c_int, [_f32p, C.c_int64, C.c_int32, C.c_float] + [_f32p] * 9 + [C.c_void_p]),
    "dm3d_groupnorm_stats": (C.c_int, [_f32p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dm3d_groupnorm_finalize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_float, _f32p, _f32p, _f32p,
                                           _f32p, C.c_void_p]),
    "dm3d_groupnorm_finalize2": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_float, _f32p, _f32p,
                                          _f32p, _f32p, C.c_void_p]),
    "dm3d_groupnorm_partials": (C.c_int, [_f32p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "dm3d_groupnorm_partials_bytes": (C.c_int64, [C.c_int32, C.c_int64, C.c_int32]),
    "dm3d_affine_act_batched": (C.c_int, [_f32p, _f32p, C.c_int32, C.c_int64, C.c_int32, _f32p, _f32p, C.c_int32, C.c_void_p]),
    "dm3d_softmax_rows": (C.c_int, [_f32p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p]),
    "dm3d_affine_act": (C.c_int, [_f32p, _f32p, C.c_int64, C.c_int32, _f32p, _f32p, C.c_int32, C.c_void_p]),
    "dm3d_ddpm_update": (C.c_int, [C.POINTER(DdpmDesc), C.c_void_p]),
    "dm3d_range_check": (C.c_int, [_f32p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "dm3d_add_i32": (C.c_int, [_i32p, C.c_int32, C.c_int32, C.c_void_p]),
    "dm3d_randn": (C.c_int, [_f32p, C.c_int64, C.c_uint64, C.c_uint32, C.c_void_p]),
    "dm3d_gather_rows": (C.c_int, [_f32p, C.c_int32, _i32p, _f32p, C.c_int32, C.c_int32, C.c_void_p]),
    "dm3d_vq_assign": (C.c_int, [_f32p, C.c_int64, C.c_int32, _f32p, C.c_int32, _f32p, _i32p, C.c_void_p]),
    "dm3d_batchnorm_finalize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_float] + [_f32p] * 8 + [C.c_float, C.c_int32, C.c_void_p]),
    "dm3d_affine_act_cat": (C.c_int, [_f32p, C.c_int32, _f32p, C.c_int32, C.c_int64, _f32p, _f32p, C.c_int32, _f32p, C.c_void_p]),
    "dm3d_bn_act_bwd": (C.c_int, [_f32p, _f32p, C.c_int32, _f32p, C.c_int32, C.c_int64, _f32p, _f32p, _f32p, _f32p, C.c_int32, C.c_void_p,
                                   _f32p, _f32p, _f32p, _f32p, C.c_void_p]),
    "dm3d_wgrad": (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    "dm3d_colsum": (C.c_int, [_f32p, C.c_int64, C.c_int64, C.c_int32, _f32p, C.c_int64, C.c_void_p]),
    "dm3d_flip_transpose": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_int32, _f32p, C.c_void_p]),
    "dm3d_layernorm_bwd": (C.c_int, [_f32p, C.c_int64, C.c_int32, C.c_float, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p]),
    "dm3d_softmax_bwd": (C.c_int, [_f32p, _f32p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_void_p]),
    "dm3d_act_bwd": (C.c_int, [_f32p, _f32p, _f32p, C.c_int64, C.c_int32, C.c_void_p]),
    "dm3d_axpy": (C.c_int, [_f32p, _f32p, C.c_int64, C.c_float, C.c_void_p]),
    "dm3d_fill": (C.c_int, [_f32p, C.c_int64, C.c_float, C.c_void_p]),
    "dm3d_transpose": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _f32p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "dm3d_copy_cols": (C.c_int, [_f32p, C.c_int64, C.c_int32, _f32p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "dm3d_upsample2": (C.c_int, [_f32p, _f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dm3d_sumpool2_add": (C.c_int, [_f32p, _f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dm3d_dilate2": (C.c_int, [_f32p, _f32p] + [C.c_int32] * 11 + [C.c_void_p]),
    "dm3d_scatter_add_rows": (C.c_int, [_f32p, _i32p, C.c_int32, C.c_int32, _f32p, C.c_int32, C.c_void_p]),
    "dm3d_q_sample": (C.c_int, [_f32p, _f32p, _i32p, _f32p, _f32p, C.c_int32, _f32p, C.c_int32, C.c_int64, C.c_void_p]),
    "dm3d_mse_loss_grad": (C.c_int, [_f32p, _f32p, C.c_int64, C.c_double, C.c_void_p, _f32p, C.c_void_p]),
    "dm3d_adam": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dm3d_graph_begin": (C.c_int, [C.c_void_p]),
    "dm3d_graph_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "dm3d_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dm3d_graph_destroy": (C.c_int, [C.c_void_p]),
}

_lib = None


class Dm3dError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Dm3dError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. There is no CPU or PyTorch fallback; "
                "run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root.")
        # PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64 (same sonames as /opt/rocm's).  Whichever is mapped first
        # serves the whole process, and if /opt/rocm's came first — this library loaded and initialised HIP before torch was
        # imported — torch's remaining bundled libraries no longer match it and torch reports "No HIP GPUs are available".
        # Importing torch here pins the order: one runtime, torch's, for both.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        if handle.dm3d_version() != ABI_VERSION:
            raise Dm3dError(f"{LIB_PATH} reports ABI version {handle.dm3d_version()}, these bindings expect {ABI_VERSION}: "
                            "rebuild the library (descriptor layouts differ between versions)")
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().dm3d_last_error()
        raise Dm3dError(f"{what or 'dm3d call'} failed (code {rc}): {msg.decode() if msg else '?'}")


def require_device() -> None:
    if not lib().dm3d_device_ok():
        raise Dm3dError("no gfx950 (MI355X) device is visible: the dm3d kernels have no CPU path")


# ---- roctx ranges (conv / norm / attn / ddpm-update visible in rocprofv3 --marker-trace timelines) ---------------------
# Host-side markers only: they cost two library calls per range, so they are off unless DM3D_ROCTX=1 (tools/profile_round.sh sets it for
# the eager PMC passes; a HIP-graph replay has no host ranges to show).
_roctx = None


def roctx():
    """(push, pop) callables; no-ops unless DM3D_ROCTX=1 and the roctx library loads."""
    global _roctx
    if _roctx is None:
        _roctx = (lambda name: None, lambda: None)
        if os.environ.get("DM3D_ROCTX") == "1":
            for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    h = C.CDLL(name)
                    h.roctxRangePushA.argtypes, h.roctxRangePushA.restype = [C.c_char_p], C.c_int
                    h.roctxRangePop.argtypes, h.roctxRangePop.restype = [], C.c_int
                    _roctx = (lambda s, h=h: h.roctxRangePushA(s.encode()), lambda h=h: h.roctxRangePop())
                    break
                except OSError:
                    continue
    return _roctx

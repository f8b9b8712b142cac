"""Immediate-mode wrappers over the C ABI (one call = one kernel launch on the current stream).

These mirror the stock Keras / TensorFlow ops the reference's hot path executes (include/dm3d.h cites each one).  They
validate shapes in Python (``ValueError``, like the reference's only explicit check, conditional_dm3d.py:343-346) and then
call the HIP kernels; there is no fallback implementation.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_NONE, ConvDesc, DdpmDesc, GemmDesc, check, lib


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous float32 device tensor")
    return t


def pack_weights(kernel: torch.Tensor, in_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Keras Conv3D [kd,kh,kw,Cin,Cout] or Dense [in,out] kernel -> [taps][CoutPad][CinPad]."""
    _f32c(kernel, "kernel")
    if kernel.dim() not in (2, 5):
        raise ValueError("kernel must be [in,out] or [kd,kh,kw,cin,cout]")
    taps = 1 if kernel.dim() == 2 else kernel.shape[0] * kernel.shape[1] * kernel.shape[2]
    cin, cout = kernel.shape[-2], kernel.shape[-1]
    out = torch.empty(lib().dm3d_packed_weight_elems(taps, cin, cout), dtype=torch.float32, device=kernel.device)
    check(lib().dm3d_pack_weights(kernel.data_ptr(), taps, cin, cout, _p(in_scale), out.data_ptr(), _st()), "pack_weights")
    return out


def h3_weight_exponent(*kernels) -> int:
    """Power-of-two pre-scale of the H3 weight images: the largest |w| of all given kernels lands in [2^13, 2^14)."""
    import math
    wmax = max(float(k.abs().max()) for k in kernels)
    return 0 if wmax == 0.0 or not math.isfinite(wmax) else max(-100, min(100, int(13 - math.floor(math.log2(wmax)))))


def pack_weights_skip_h3p(kernel: torch.Tensor, w_exp: int) -> torch.Tensor:
    """Keras 1x1 kernel ([1,1,1,cin,cout] or [cin,cout]) -> image for conv3d(skip=...), packed with the main conv's w_exp."""
    _f32c(kernel, "kernel")
    cin, cout = kernel.shape[-2], kernel.shape[-1]
    out = torch.empty(lib().dm3d_packed_weight_skip_h3p_bytes(cin, cout) // 2, dtype=torch.float16, device=kernel.device)
    check(lib().dm3d_pack_weights_skip_h3p(kernel.data_ptr(), cin, cout, w_exp, out.data_ptr(), _st()), "pack_weights_skip_h3p")
    return out


def pack_weights_skip_h3f(kernel: torch.Tensor, w_exp: int) -> torch.Tensor:
    """The same 1x1 kernel as MFMA operand fragments: conv3d(skip=(x1, x2, image, fragments)) lets the Winograd-x form serve the launch."""
    _f32c(kernel, "kernel")
    cin, cout = kernel.shape[-2], kernel.shape[-1]
    out = torch.empty(lib().dm3d_packed_weight_skip_h3p_bytes(cin, cout) // 2, dtype=torch.float16, device=kernel.device)
    check(lib().dm3d_pack_weights_skip_h3f(kernel.data_ptr(), cin, cout, w_exp, out.data_ptr(), _st()), "pack_weights_skip_h3f")
    return out


def pack_weights_h3(kernel: torch.Tensor, in_scale: Optional[torch.Tensor] = None, stride: int = 1, w_exp: Optional[int] = None):
    """Keras kernel -> (float16 hi/lo image for the H3 conv kernels, w_exp).  max|w|*2^w_exp lands in [2^13, 2^14).
    ``stride`` is the stride of the conv that will read the image: it selects the layout (dm3d_conv_weight_layout)."""
    _f32c(kernel, "kernel")
    if kernel.dim() not in (2, 5):
        raise ValueError("kernel must be [in,out] or [kd,kh,kw,cin,cout]")
    taps = 1 if kernel.dim() == 2 else kernel.shape[0] * kernel.shape[1] * kernel.shape[2]
    cin, cout = kernel.shape[-2], kernel.shape[-1]
    if w_exp is None:
        w_exp = h3_weight_exponent(kernel)
    ksize = {1: 1, 27: 3, 64: 4}.get(taps, 0)
    if ksize and lib().dm3d_conv_weight_layout(ksize, stride, 0, 0, cout) == _lib.WL_PAIR:
        out = torch.empty(lib().dm3d_packed_weight_h3p_bytes(taps, cin, cout) // 2, dtype=torch.float16, device=kernel.device)
        check(lib().dm3d_pack_weights_h3p(kernel.data_ptr(), taps, cin, cout, w_exp, _p(in_scale), out.data_ptr(), 0, _st()),
              "pack_weights_h3p")
        return out, w_exp
    out = torch.empty(lib().dm3d_packed_weight_h3_bytes(taps, cin, cout) // 2, dtype=torch.float16, device=kernel.device)
    check(lib().dm3d_pack_weights_h3(kernel.data_ptr(), taps, cin, cout, w_exp, _p(in_scale), out.data_ptr(), _st()),
          "pack_weights_h3")
    return out, w_exp


def pack_weights_h3w(kernel: torch.Tensor, w_exp: int, in_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight image of a k3 / stride-1 conv ([3,3,3,cin,cout]) for the Winograd-x form (conv3d(wpk_wino=...)): the x taps of every
    (dz, dy, cin, cout) become the four F(2,3) terms; packed with the w_exp of the conv's main image."""
    _f32c(kernel, "kernel")
    if kernel.dim() != 5 or tuple(kernel.shape[:3]) != (3, 3, 3):
        raise ValueError("kernel must be [3,3,3,cin,cout]")
    cin, cout = kernel.shape[-2], kernel.shape[-1]
    out = torch.empty(lib().dm3d_packed_weight_h3w_bytes(cin, cout) // 2, dtype=torch.float16, device=kernel.device)
    check(lib().dm3d_pack_weights_h3w(kernel.data_ptr(), cin, cout, w_exp, _p(in_scale), out.data_ptr(), _st()), "pack_weights_h3w")
    return out


def pack_weights_up(kernel: torch.Tensor, h3: bool = False):
    """[3,3,3,Cin,Cout] kernel of an UpSample conv -> the 8 parity images dm3d_conv3d_ndhwc(upsample=1) expects.
    Returns wpk (fp32) or (wpk, w_exp) (h3)."""
    _f32c(kernel, "kernel")
    if kernel.dim() != 5 or tuple(kernel.shape[:3]) != (3, 3, 3):
        raise ValueError("kernel must be [3,3,3,cin,cout]")
    cin, cout = kernel.shape[-2], kernel.shape[-1]
    if not h3:
        out = torch.empty(lib().dm3d_packed_weight_up_elems(cin, cout), dtype=torch.float32, device=kernel.device)
        check(lib().dm3d_pack_weights_up(kernel.data_ptr(), cin, cout, out.data_ptr(), _st()), "pack_weights_up")
        return out
    import math
    from .weights import upsample_parity_kernels
    wmax = float(abs(upsample_parity_kernels(kernel.cpu().numpy())).max())
    w_exp = 0 if wmax == 0.0 or not math.isfinite(wmax) else int(13 - math.floor(math.log2(wmax)))
    out = torch.empty(lib().dm3d_packed_weight_up_h3_bytes(cin, cout) // 2, dtype=torch.float16, device=kernel.device)
    check(lib().dm3d_pack_weights_up_h3(kernel.data_ptr(), cin, cout, w_exp, out.data_ptr(), _st()), "pack_weights_up_h3")
    return out, w_exp


def pack_weights_convt(kernel: torch.Tensor, h3: bool = False):
    """[4,4,4,Cout,Cin] kernel of Conv3DTranspose(k=4, strides=2, padding="same") -> the 8 parity images that
    dm3d_conv3d_ndhwc(transpose=1) expects.  Returns wpk (fp32) or (wpk, w_exp) (h3)."""
    _f32c(kernel, "kernel")
    if kernel.dim() != 5 or tuple(kernel.shape[:3]) != (4, 4, 4):
        raise ValueError("kernel must be [4,4,4,cout,cin]")
    cout, cin = kernel.shape[-2], kernel.shape[-1]
    if not h3:
        out = torch.empty(lib().dm3d_packed_weight_up_elems(cin, cout), dtype=torch.float32, device=kernel.device)
        check(lib().dm3d_pack_weights_convt(kernel.data_ptr(), cin, cout, out.data_ptr(), _st()), "pack_weights_convt")
        return out
    import math
    wmax = float(kernel.abs().max())
    w_exp = 0 if wmax == 0.0 or not math.isfinite(wmax) else int(13 - math.floor(math.log2(wmax)))
    out = torch.empty(lib().dm3d_packed_weight_up_h3_bytes(cin, cout) // 2, dtype=torch.float16, device=kernel.device)
    check(lib().dm3d_pack_weights_convt_h3(kernel.data_ptr(), cin, cout, w_exp, out.data_ptr(), _st()), "pack_weights_convt_h3")
    return out, w_exp


def conv3d(x1, wpk, cout, ksize, *, x2=None, bias=None, stride=1, upsample=False, pro_scale=None, pro_shift=None,
           vec=None, vec_idx=None, relu=False, res=None, precision=_lib.PREC_F32, w_exp=0, prelu_alpha=None,
           relu_out=False, transpose=False, skip=None, x1_h2_channels=None, out_h2=False, post=None, wpk_wino=None, gn_stats=None, split=True) -> torch.Tensor:
    """Conv3D(padding="same") on NDHWC with the fused prologue / concat / upsample / epilogue of dm3d_conv3d_ndhwc.
    ``skip=(sx1, sx2_or_None, skip_wpk[, skip_wpk_frag])``: also accumulate Conv3D(cout, 1) of the raw concat(sx1, sx2) (H3, k3, stride 1).
    ``post=(scale, shift)``: out = silu(out*scale[c] + shift[c]) at the very end; ``out_h2``: store DM3D_FMT_H2;
    ``x1_h2_channels=c``: x1 is a DM3D_FMT_H2 buffer of c logical channels (as written by ``out_h2``).
    ``split=False``: no split_counters, i.e. a small grid is not split along Cin (the A/B arm of the hand-over form)."""
    _f32c(x1, "x1")
    if x1.dim() != 5:
        raise ValueError("x1 must be [B,D,H,W,C]")
    B, D, H, W, c1 = x1.shape
    c2 = 0
    if x2 is not None:
        _f32c(x2, "x2")
        if x2.shape[:4] != x1.shape[:4]:
            raise ValueError("x2 must share x1's batch and spatial shape")
        c2 = x2.shape[4]
    up = 2 if upsample else 1
    od, oh, ow = (-(-D * up // stride), -(-H * up // stride), -(-W * up // stride))
    if transpose:
        od, oh, ow = 2 * D, 2 * H, 2 * W
    out = torch.empty(B, od, oh, ow, cout, dtype=torch.float32, device=x1.device)
    d = ConvDesc()
    d.x1, d.x2, d.c1, d.c2, d.batch = x1.data_ptr(), _p(x2), c1, c2, B
    d.in_d, d.in_h, d.in_w, d.upsample, d.ksize, d.stride = D, H, W, int(bool(upsample)), ksize, stride
    d.wpk, d.bias, d.pro_scale, d.pro_shift = wpk.data_ptr(), _p(bias), _p(pro_scale), _p(pro_shift)
    if vec is not None:
        d.vec, d.vec_idx, d.vec_ld = vec.data_ptr(), _p(vec_idx), vec.shape[-1]
    if res is not None and tuple(res.shape) != tuple(out.shape):
        raise ValueError("res must have the output's shape")
    d.relu, d.res, d.out, d.cout = int(bool(relu)), _p(res), out.data_ptr(), cout
    d.precision, d.w_exp = precision, w_exp
    if precision == _lib.PREC_H3:
        d.w_layout = lib().dm3d_conv_weight_layout(ksize, stride, int(bool(upsample)), int(bool(transpose)), cout)
    if prelu_alpha is not None and tuple(prelu_alpha.shape) != (od, oh, ow, cout):
        raise ValueError("prelu_alpha must be [out_d, out_h, out_w, cout]")
    d.prelu_alpha, d.relu_out, d.transpose = _p(prelu_alpha), int(bool(relu_out)), int(bool(transpose))
    if x1_h2_channels is not None:
        d.x1_fmt = _lib.FMT_H2
    if out_h2:
        d.out_fmt = _lib.FMT_H2
    if post is not None:
        d.post_scale, d.post_shift = post[0].data_ptr(), post[1].data_ptr()
    if wpk_wino is not None:
        d.wpk_wino = wpk_wino.data_ptr()
    if gn_stats is not None:               # float32 [B, ceil(voxels / 64), cout, 2]: partial (sum, sum of squares) of the output per (sample, channel)
        d.gn_stats = gn_stats.data_ptr()
    if skip is not None:
        sx1, sx2, swpk = skip[:3]
        if len(skip) > 3 and skip[3] is not None:
            d.skip_wpk_frag = skip[3].data_ptr()
        _f32c(sx1, "skip x1")
        d.skip_x1, d.skip_x2, d.skip_c1, d.skip_c2, d.skip_wpk = sx1.data_ptr(), _p(sx2), sx1.shape[-1], (sx2.shape[-1] if sx2 is not None else 0), swpk.data_ptr()
    # the Cin split of small grids (include/dm3d.h, split_counters): tickets (zero before, zero after) and the parts' workspace
    need, words = lib().dm3d_conv_scratch_bytes(C.byref(d)), lib().dm3d_conv_split_counter_words(C.byref(d))
    if need and split:
        scratch = torch.empty(need // 4, dtype=torch.float32, device=x1.device)     # stays alive until the launches are enqueued:
        d.scratch, d.scratch_bytes = scratch.data_ptr(), need                       # same stream, so the allocator cannot reuse it early
        counters = _split_counters(x1.device, words)
        d.split_counters, d.split_counter_words = counters.data_ptr(), counters.numel()
    check(lib().dm3d_conv3d_ndhwc(C.byref(d), _st()), "conv3d")
    return out


_COUNTERS = {}


def _split_counters(device, words: int) -> torch.Tensor:
    """One zeroed ticket buffer per device for the op-level calls (every launch leaves it zero; launches of a stream are ordered)."""
    buf = _COUNTERS.get(device)
    if buf is None or buf.numel() < words:
        buf = _COUNTERS[device] = torch.zeros(max(4096, words), dtype=torch.int32, device=device)
    return buf


def split_h2(src: torch.Tensor, exp2: int = 0) -> torch.Tensor:
    """float32 [rows, k] -> DM3D_FMT_H2 [rows, round_up(k,16)] (returned as a float32-typed buffer of the same byte size)."""
    _f32c(src, "src")
    rows, k = src.numel() // src.shape[-1], src.shape[-1]
    ld = -(-k // 16) * 16
    dst = torch.empty(rows, ld, dtype=torch.float32, device=src.device)
    check(lib().dm3d_split_h2(src.data_ptr(), rows, k, k, exp2, dst.data_ptr(), ld, _st()), "split_h2")
    return dst


def h2_to_f32(h2: torch.Tensor, k: int) -> torch.Tensor:
    """Host-side decoder of DM3D_FMT_H2 (tests only): [rows, ld] float32-typed buffer -> float32 hi + lo values."""
    rows, ld = h2.shape
    rec = h2.contiguous().view(torch.float16).reshape(rows, ld // 16, 4, 8).float()
    val = rec[:, :, 0:2, :] + rec[:, :, 2:4, :]
    return val.reshape(rows, ld)[:, :k]


def gemm_tn(a, b, *, m=None, n=None, k=None, lda=None, ldb=None, batch=1, stride_a=None, stride_b=None, alpha=1.0, bias=None,
            bias_along_m=False, act=ACT_NONE, res=None, out=None, precision=_lib.PREC_F32, a_fmt=_lib.FMT_F32,
            b_fmt=_lib.FMT_F32, out_fmt=_lib.FMT_F32) -> torch.Tensor:
    """out[b][m][n] = act(alpha * sum_k a[b][m][k] b[b][n][k] + bias) + res.  a: [batch?, m, k], b: [batch?, n, k]."""
    _f32c(a, "a"), _f32c(b, "b")
    m = a.shape[-2] if m is None else m
    k = a.shape[-1] if k is None else k
    n = b.shape[-2] if n is None else n
    lda = a.shape[-1] if lda is None else lda
    ldb = b.shape[-1] if ldb is None else ldb
    if stride_a is None:
        stride_a = a.shape[-2] * a.shape[-1] if (batch > 1 and a.dim() == 3 and a.shape[0] == batch) else 0
    if stride_b is None:
        stride_b = b.shape[-2] * b.shape[-1] if (batch > 1 and b.dim() == 3 and b.shape[0] == batch) else 0
    if out is None:
        out = torch.empty((batch, m, n) if batch > 1 else (m, n), dtype=torch.float32, device=a.device)
    d = GemmDesc()
    d.a, d.lda, d.stride_a = a.data_ptr(), lda, stride_a
    d.b, d.ldb, d.stride_b = b.data_ptr(), ldb, stride_b
    d.out, d.ldo, d.stride_o = out.data_ptr(), n, m * n
    d.m, d.n, d.k, d.batch, d.alpha = m, n, k, batch, alpha
    d.bias, d.bias_along_m, d.act = _p(bias), int(bool(bias_along_m)), act
    if res is not None:
        d.res, d.ldr, d.stride_r = res.data_ptr(), n, m * n
    d.precision, d.a_fmt, d.b_fmt, d.out_fmt = precision, a_fmt, b_fmt, out_fmt
    check(lib().dm3d_gemm_tn(C.byref(d), _st()), "gemm_tn")
    return out


def attention(q, k, vt, scale, *, res=None, precision=_lib.PREC_F32, fmt=_lib.FMT_F32) -> torch.Tensor:
    """softmax(q k^T * scale) v + res per sample through dm3d_attention.  q [B, Lq, C]; k [B or 1, Lk, C]; vt [B or 1, C, Lk]
    (the value tensor transposed); with fmt=FMT_H2 the three operands are DM3D_FMT_H2 buffers of those logical shapes."""
    B, Lq, Cc = q.shape
    Lk = k.shape[1]
    out = torch.empty(B, Lq, Cc, dtype=torch.float32, device=q.device)
    scratch = torch.empty(lib().dm3d_attention_workspace_bytes(B, Lq, Lk) // 4, dtype=torch.float32, device=q.device)
    d = _lib.AttentionDesc()
    d.q, d.ldq = q.data_ptr(), q.shape[-1]
    d.k, d.ldk, d.stride_k = k.data_ptr(), k.shape[-1], (Lk * k.shape[-1] if k.shape[0] == B and B > 1 else 0)
    d.vt, d.ldv, d.stride_vt = vt.data_ptr(), vt.shape[-1], (vt.shape[1] * vt.shape[-1] if vt.shape[0] == B and B > 1 else 0)
    d.out, d.ldo, d.res = out.data_ptr(), Cc, _p(res)
    d.batch, d.lq, d.lk, d.c, d.scale, d.precision, d.fmt = B, Lq, Lk, Cc, float(scale), precision, fmt
    check(lib().dm3d_attention(C.byref(d), scratch.data_ptr(), _st()), "attention")
    return out


def layernorm3(x, params, eps=1e-3):
    """params: up to three (gamma, beta) pairs -> list of outputs sharing one statistics pass."""
    _f32c(x, "x")
    c = x.shape[-1]
    rows = x.numel() // c
    outs = [torch.empty_like(x) for _ in params]
    args = []
    for i in range(3):
        if i < len(params):
            args += [params[i][0].data_ptr(), params[i][1].data_ptr(), outs[i].data_ptr()]
        else:
            args += [None, None, None]
    check(lib().dm3d_layernorm3(x.data_ptr(), rows, c, eps, *args, _st()), "layernorm3")
    return outs


def layernorm3_h2(x, params, eps=1e-3):
    _f32c(x, "x")
    c = x.shape[-1]
    rows = x.numel() // c
    outs = [torch.empty_like(x) for _ in params]
    args = []
    for i in range(3):
        if i < len(params):
            args += [params[i][0].data_ptr(), params[i][1].data_ptr(), outs[i].data_ptr()]
        else:
            args += [None, None, None]
    check(lib().dm3d_layernorm3_h2(x.data_ptr(), rows, c, eps, *args, _st()), "layernorm3_h2")
    return outs


def softmax_rows_h2_(s: torch.Tensor) -> torch.Tensor:
    _f32c(s, "s")
    cols = s.shape[-1]
    check(lib().dm3d_softmax_rows_h2(s.data_ptr(), s.numel() // cols, cols, cols, _st()), "softmax_rows_h2")
    return s


def softmax_rows_(s: torch.Tensor) -> torch.Tensor:
    _f32c(s, "s")
    cols = s.shape[-1]
    check(lib().dm3d_softmax_rows(s.data_ptr(), s.numel() // cols, cols, cols, _st()), "softmax_rows")
    return s


def affine_act(x, scale=None, shift=None, act=ACT_NONE) -> torch.Tensor:
    _f32c(x, "x")
    c = x.shape[-1]
    y = torch.empty_like(x)
    check(lib().dm3d_affine_act(x.data_ptr(), y.data_ptr(), x.numel() // c, c, _p(scale), _p(shift), act, _st()), "affine_act")
    return y


def randn(shape, seed: int, stream_id: int = 0, device="cuda") -> torch.Tensor:
    x = torch.empty(shape, dtype=torch.float32, device=device)
    check(lib().dm3d_randn(x.data_ptr(), x.numel(), seed, stream_id, _st()), "randn")
    return x


def vq_assign(z: torch.Tensor, codebook_t: torch.Tensor, esq: torch.Tensor) -> torch.Tensor:
    """Nearest-code indices (reference VectorQuantizer.get_code_indices): z [rows, D], codebook_t = E^T [K, D], esq [K]."""
    _f32c(z, "z"), _f32c(codebook_t, "codebook_t")
    rows, dd = z.shape
    k = codebook_t.shape[0]
    sim = gemm_tn(z, codebook_t)                         # exact float32 MFMA: z.E
    idx = torch.empty(rows, dtype=torch.int32, device=z.device)
    check(lib().dm3d_vq_assign(z.data_ptr(), rows, dd, sim.data_ptr(), k, esq.data_ptr(), idx.data_ptr(), _st()), "vq_assign")
    return idx


def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    _f32c(table, "table")
    out = torch.empty(idx.numel(), table.shape[1], dtype=torch.float32, device=table.device)
    check(lib().dm3d_gather_rows(table.data_ptr(), table.shape[0], idx.data_ptr(), out.data_ptr(), idx.numel(), table.shape[1],
                                 _st()), "gather_rows")
    return out


def pack_mlp_weights(w_h2: torch.Tensor, units: int, which: int) -> torch.Tensor:
    """DM3D_FMT_H2 weights of the MLP (which = 0: [4 units, units]; 1: [units, 4 units]) -> the operand-fragment image dm3d_mlp_fused reads."""
    out = torch.empty_like(w_h2)
    check(lib().dm3d_pack_mlp_weights(w_h2.data_ptr(), units, which, out.data_ptr(), _st()), "pack_mlp_weights")
    return out


def pack_front_weights(w_h2: torch.Tensor, n: int, units: int = 256) -> torch.Tensor:
    """DM3D_FMT_H2 weight rows W[n][units] (split_h2 of the [n, units] float32 weight) -> the operand-fragment image dm3d_attn_front reads."""
    out = torch.empty_like(w_h2)
    check(lib().dm3d_pack_front_weights(w_h2.data_ptr(), n, units, out.data_ptr(), _st()), "pack_front_weights")
    return out


def attn_front(x: torch.Tensor, w_in_t: torch.Tensor, b_in: torch.Tensor, w_qk_t: torch.Tensor, b_qk: torch.Tensor, w_v_t: torch.Tensor,
               b_v: torch.Tensor, norms, eps: float = 1e-3):
    """The front half of a CrossAttentionBlock in one launch (dm3d_attn_front; conditional_dm3d.py:186-193, 163-170): x [m, 256] float32,
    weights from pack_front_weights, norms = ((g1, b1), (g2, b2), (g3, b3)).  Returns y [m, 256] float32 and the DM3D_FMT_H2 buffers
    qk [m, 512], v_t [256, m], q2 [m, 256], n3 [m, 256] (decode with h2_to_f32)."""
    from ._lib import AttnFrontDesc
    _f32c(x, "x")
    m, u = x.shape
    dev = x.device
    y = torch.empty(m, u, dtype=torch.float32, device=dev)
    qk, v_t = torch.empty(m, 2 * u, dtype=torch.float32, device=dev), torch.empty(u, m, dtype=torch.float32, device=dev)
    q2, n3 = torch.empty(m, u, dtype=torch.float32, device=dev), torch.empty(m, u, dtype=torch.float32, device=dev)
    d = AttnFrontDesc()
    d.x, d.ldx = x.data_ptr(), u
    d.w_in, d.b_in, d.w_qk, d.b_qk, d.w_v, d.b_v = w_in_t.data_ptr(), b_in.data_ptr(), w_qk_t.data_ptr(), b_qk.data_ptr(), w_v_t.data_ptr(), b_v.data_ptr()
    (g1, b1), (g2, b2), (g3, b3) = norms
    d.g1, d.be1, d.g2, d.be2, d.g3, d.be3 = g1.data_ptr(), b1.data_ptr(), g2.data_ptr(), b2.data_ptr(), g3.data_ptr(), b3.data_ptr()
    d.eps = eps
    d.y, d.ldy, d.qk, d.ldqk, d.vt, d.ldvt, d.q2, d.ldq2, d.n3, d.ldn3 = y.data_ptr(), u, qk.data_ptr(), 2 * u, v_t.data_ptr(), m, q2.data_ptr(), u, n3.data_ptr(), u
    d.m, d.units = m, u
    check(lib().dm3d_attn_front(C.byref(d), _st()), "attn_front")
    return y, qk, v_t, q2, n3


def mlp_fused(x_h2: torch.Tensor, w0_t: torch.Tensor, b0: torch.Tensor, w1_t: torch.Tensor, b1: torch.Tensor, units: int,
              res: Optional[torch.Tensor] = None, res2: Optional[torch.Tensor] = None, out_h2: bool = False,
              tail=None) -> torch.Tensor:
    """Dense(units)(relu(Dense(4 units)(x))) + res + res2 in one launch (dm3d_mlp_fused): x [m, units] as a DM3D_FMT_H2 buffer (split_h2),
    w0_t / w1_t from pack_mlp_weights, biases / residuals float32.  Returns [m, units] float32, or a DM3D_FMT_H2 buffer with ``out_h2``.
    ``tail = (w2_t, b2, res3)`` (w2_t from pack_front_weights; res3 may be None): returns relu(Dense_2(that)) + res3 instead (float32)."""
    from ._lib import MlpDesc
    m = x_h2.numel() // units
    out = torch.empty(m, units, dtype=torch.float32, device=x_h2.device)
    d = MlpDesc()
    d.x, d.ldx, d.w0, d.b0, d.w1, d.b1 = x_h2.data_ptr(), units, w0_t.data_ptr(), b0.data_ptr(), w1_t.data_ptr(), b1.data_ptr()
    d.res, d.res2, d.ldr = _p(res), _p(res2), units
    d.out, d.ldo, d.out_fmt, d.m, d.units = out.data_ptr(), units, (_lib.FMT_H2 if out_h2 else _lib.FMT_F32), m, units
    if tail is not None:
        w2_t, b2, res3 = tail
        d.w2, d.b2, d.res3, d.ldr3 = w2_t.data_ptr(), b2.data_ptr(), _p(res3), units
    check(lib().dm3d_mlp_fused(C.byref(d), _st()), "mlp_fused")
    return out

"""The 3D U-Net of the reference (``build_model``) as a static launch plan over the dm3d HIP kernels.

reference: networks/conditional_dm3d.py:324-415 (conditional, CrossAttentionBlock) and networks/dm3d.py:294-376
(unconditional, AttentionBlock).  PyTorch is used only to own device memory and streams; every arithmetic op on
activations is a call through the C ABI (include/dm3d.h).  A *plan* is the flat list of those calls for one batch size
with every buffer pre-allocated, so a whole denoising step can be captured into a HIP graph and replayed.

What is hoisted out of the per-step work (none of it changes results):
  * inference BatchNormalization is folded to a per-channel scale/shift applied, with swish, while the conv stages its
    input tile into LDS (ResidualBlock :255-256, 262-263; end block :410-411), or folded into the following 1x1 conv's
    weights (CrossAttentionBlock norm -> proj_in, :187-188);
  * the time path (TimeEmbedding -> TimeMLP -> per-ResidualBlock swish+Dense, :355-356, 250-253) depends only on t: it
    is evaluated once for all needed t into a [rows, sum(widths)] table that the conv epilogues index with t[b];
  * ContextMLP and the cross-attention key/value projections (:310-318, 168-169) depend only on the context id: they
    are evaluated once per weight set for both ids.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SILU, AttnFrontDesc, ConvDesc, GemmDesc, MlpDesc, check, lib
from .betas import time_embedding_table
from .weights import UNetConfig, keras_init_weights, upsample_parity_kernels, walk

BN_EPS = 1e-3     # keras.layers.BatchNormalization default
LN_EPS = 1e-3     # keras.layers.LayerNormalization default


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor], offset_elems: int = 0) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() + 4 * offset_elems


def _h3_exponent(*kernels) -> int:
    """power-of-two pre-scale so that max|w| over the given kernels lands in [2^13, 2^14): hi stays finite, lo a normal float16"""
    wmax = max(float(np.abs(k).max()) for k in kernels)
    return 0 if wmax == 0.0 or not np.isfinite(wmax) else max(-100, min(100, int(13 - np.floor(np.log2(wmax)))))


class _Conv:
    """Packed Conv3D / Dense weights on the device."""

    def __init__(self, wpk, bias, taps, cin, cout, precision=_lib.PREC_F32, w_exp=0):
        self.wpk, self.bias, self.taps, self.cin, self.cout = wpk, bias, taps, cin, cout
        self.precision, self.w_exp = precision, w_exp
        self.cin_pad = -(-cin // _lib.CIN_PAD) * _lib.CIN_PAD
        self.h2 = None              # DM3D_FMT_H2 copy of wpk ([cout_pad][cin_pad]) for the H3 GEMM
        self.wpk_wino = None        # weight image of the Winograd-x form (dm3d_conv_desc.wpk_wino)


class UNet:
    """Callable like the Keras model returned by ``build_model``: ``net([x, t, context]) -> eps`` (NDHWC float32)."""

    def __init__(self, cfg: UNetConfig, device="cuda", weights: Optional[Dict[str, np.ndarray]] = None, seed: int = 0,
                 precision: Optional[str] = None):
        """``precision``: arithmetic of the Conv3d kernels — "fp32" (exact float32 MFMA) or "h3" (float16 hi+lo split,
        three 16-bit MFMA passes, float32 accumulate: float32-grade results, see include/dm3d.h).  Default: the
        DM3D_PRECISION environment variable, else "h3"."""
        precision = precision or os.environ.get("DM3D_PRECISION", "h3")
        if precision not in ("fp32", "h3"):
            raise ValueError("precision must be 'fp32' or 'h3'")
        # precision "h3": k3 / stride-1 convs with Cout > 32 and Cin >= 32 also carry the Winograd-x image; the library uses it where that
        # form is faster (large grids: dm3d_conv_tile_form() == 10).  DM3D_CONV_WINO=0 in the environment at construction: no second image.
        self.wino = precision == "h3" and os.environ.get("DM3D_CONV_WINO", "1") != "0"
        self.precision_name = precision
        self.precision = precision
        self.fuse_skip = os.environ.get("DM3D_FUSE_SKIP", "1") != "0"      # A/B switch: ResidualBlock 1x1 skip conv inside conv2's launch
        self.h2_handoff = os.environ.get("DM3D_H2_HANDOFF", "1") != "0"    # A/B switch: conv1 -> conv2 hand-off in DM3D_FMT_H2
        self.cfg = cfg
        self.blocks, self.spec = walk(cfg)
        self.device = torch.device(device)
        self.state: Dict[str, np.ndarray] = {}
        self._plans: Dict[tuple, "Plan"] = {}
        self._prepared = False
        self._before_use = None         # set by DiffusionModel: brings weights changed by train_step back before the network is used
        self._training_engine = None    # set by DiffusionModel: its Trainer (if it has one) serves network(..., training=True)
        self.load_state_dict(weights if weights is not None else keras_init_weights(cfg, seed))

    # ---- weights -------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, np.ndarray], strict: bool = True):
        new = {}
        for name, shape in self.spec.items():
            if name not in sd:
                if strict:
                    raise ValueError(f"missing weight {name}")
                new[name] = self.state[name]
                continue
            arr = sd[name]
            if isinstance(arr, torch.Tensor):
                arr = arr.detach().cpu().numpy()
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            if tuple(arr.shape) != tuple(shape):
                raise ValueError(f"weight {name}: expected shape {tuple(shape)}, got {tuple(arr.shape)}")
            new[name] = arr
        extra = set(sd) - set(self.spec)
        if strict and extra:
            raise ValueError(f"unexpected weights: {sorted(extra)[:5]}")
        self.state = new
        self._prepared = False
        self._plans.clear()

    def _fresh(self):
        if self._before_use is not None:
            self._before_use()

    def state_dict(self) -> Dict[str, np.ndarray]:
        self._fresh()
        return dict(self.state)

    def num_params(self) -> int:
        return int(sum(int(np.prod(s)) for s in self.spec.values()))

    # ---- one-time device preparation -----------------------------------------------------------------------------
    def _dev(self, arr: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)

    def _pack(self, kernel: np.ndarray, bias: Optional[np.ndarray], in_scale: Optional[torch.Tensor] = None,
              conv: bool = False, up: bool = False, stride: int = 1, w_exp: Optional[int] = None) -> _Conv:
        """``conv=True``: weights of a dm3d_conv3d_ndhwc launch (packed for self.precision); otherwise GEMM operand.
        ``up=True``: UpSample conv — packed as the 8 parity 2x2x2 kernels the upsample launch expects."""
        shape = kernel.shape
        taps = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        cin, cout = int(shape[-2]), int(shape[-1])
        raw = self._dev(kernel)
        dbias = self._dev(bias) if bias is not None else None
        if up:
            if self.precision == "h3":
                wmax = float(np.abs(upsample_parity_kernels(kernel)).max())
                w_exp = 0 if wmax == 0.0 or not np.isfinite(wmax) else max(-100, min(100, int(13 - np.floor(np.log2(wmax)))))
                wpk = torch.empty(lib().dm3d_packed_weight_up_h3_bytes(cin, cout) // 2, dtype=torch.float16, device=self.device)
                check(lib().dm3d_pack_weights_up_h3(raw.data_ptr(), cin, cout, w_exp, wpk.data_ptr(), _stream()), "pack_weights_up_h3")
                return _Conv(wpk, dbias, taps, cin, cout, _lib.PREC_H3, w_exp)
            wpk = torch.empty(lib().dm3d_packed_weight_up_elems(cin, cout), dtype=torch.float32, device=self.device)
            check(lib().dm3d_pack_weights_up(raw.data_ptr(), cin, cout, wpk.data_ptr(), _stream()), "pack_weights_up")
            return _Conv(wpk, dbias, taps, cin, cout)
        if conv and self.precision == "h3":
            # power-of-two pre-scale so that max|w| lands in [2^13, 2^14): hi stays finite, lo stays a normal float16
            if w_exp is None:
                w_exp = _h3_exponent(kernel)
            if lib().dm3d_conv_weight_layout({1: 1, 27: 3}[taps], stride, 0, 0, cout) == _lib.WL_PAIR:
                wpk = torch.empty(lib().dm3d_packed_weight_h3p_bytes(taps, cin, cout) // 2, dtype=torch.float16, device=self.device)
                check(lib().dm3d_pack_weights_h3p(raw.data_ptr(), taps, cin, cout, w_exp, _ptr(in_scale), wpk.data_ptr(), 0,
                                                  _stream()), "pack_weights_h3p")
                if self.wino and taps == 27 and stride == 1 and cout > 32 and cin >= 32:       # (the library never takes the form below 32 input channels)
                    cv = _Conv(wpk, self._dev(bias) if bias is not None else None, taps, cin, cout, _lib.PREC_H3, w_exp)
                    cv.wpk_wino = torch.empty(lib().dm3d_packed_weight_h3w_bytes(cin, cout) // 2, dtype=torch.float16, device=self.device)
                    check(lib().dm3d_pack_weights_h3w(raw.data_ptr(), cin, cout, w_exp, _ptr(in_scale), cv.wpk_wino.data_ptr(), _stream()), "pack_weights_h3w")
                    return cv
            else:
                wpk = torch.empty(lib().dm3d_packed_weight_h3_bytes(taps, cin, cout) // 2, dtype=torch.float16, device=self.device)
                check(lib().dm3d_pack_weights_h3(raw.data_ptr(), taps, cin, cout, w_exp, _ptr(in_scale), wpk.data_ptr(),
                                                 _stream()), "pack_weights_h3")
            return _Conv(wpk, self._dev(bias) if bias is not None else None, taps, cin, cout, _lib.PREC_H3, w_exp)
        n = lib().dm3d_packed_weight_elems(taps, cin, cout)
        wpk = torch.empty(n, dtype=torch.float32, device=self.device)
        check(lib().dm3d_pack_weights(raw.data_ptr(), taps, cin, cout, _ptr(in_scale), wpk.data_ptr(), _stream()),
              "pack_weights")
        return _Conv(wpk, self._dev(bias) if bias is not None else None, taps, cin, cout)

    def _to_h2(self, t: torch.Tensor, rows: int, k: int) -> torch.Tensor:
        """float32 [rows, k] device matrix (k % 16 == 0) -> DM3D_FMT_H2 buffer of the same byte size."""
        out = torch.empty(rows, k, dtype=torch.float32, device=self.device)
        check(lib().dm3d_split_h2(t.data_ptr(), rows, k, k, 0, out.data_ptr(), k, _stream()), "split_h2")
        return out

    def _with_h2(self, w: _Conv) -> _Conv:
        if self.precision == "h3" and w.taps == 1:
            w.h2 = self._to_h2(w.wpk, w.wpk.numel() // w.cin_pad, w.cin_pad)
        return w

    def _fold_bn(self, name: str):
        """Inference BatchNormalization gamma*(x-mean)/sqrt(var+eps)+beta as x*scale+shift.  In the GroupNormalization
        variant (cfg.norm == "group") the statistics are per sample, so only (gamma, beta) are uploaded and the plan
        computes scale/shift on the device every step (Plan._norm)."""
        s = self.state
        if self.cfg.norm == "group":
            return self._dev(s[f"{name}.gamma"]), self._dev(s[f"{name}.beta"])
        scale = (s[f"{name}.gamma"].astype(np.float64) / np.sqrt(s[f"{name}.var"].astype(np.float64) + BN_EPS))
        shift = s[f"{name}.beta"].astype(np.float64) - s[f"{name}.mean"].astype(np.float64) * scale
        return self._dev(scale.astype(np.float32)), self._dev(shift.astype(np.float32))

    def prepare(self):
        if self._prepared:
            return
        _lib.require_device()
        s, cfg = self.state, self.cfg
        P: Dict[str, object] = {}
        P["conv_in"] = self._pack(s["conv_in.kernel"], s["conv_in.bias"], conv=True)
        P["time_mlp.0"] = self._pack(s["time_mlp.0.kernel"], s["time_mlp.0.bias"])
        P["time_mlp.1"] = self._pack(s["time_mlp.1.kernel"], s["time_mlp.1.bias"])
        temb_k, temb_b, self.temb_off, off = [], [], {}, 0
        for blk in self.blocks:
            n = blk.name
            if blk.kind == "res":
                P[f"{n}.conv1"] = self._pack(s[f"{n}.conv1.kernel"], s[f"{n}.conv1.bias"], conv=True)
                if (f"{n}.skip.kernel" in s and self.precision == "h3" and self.fuse_skip
                        and blk.cout > 32 and lib().dm3d_conv_weight_layout(3, 1, 0, 0, blk.cout) == _lib.WL_PAIR):      # (the narrow column forms have no tail phase)
                    # the 1x1 skip conv rides in conv2's launch (dm3d_conv_desc.skip_*): both images share one exponent, the
                    # skip bias joins conv2's
                    e = _h3_exponent(s[f"{n}.conv2.kernel"], s[f"{n}.skip.kernel"])
                    P[f"{n}.conv2"] = self._pack(s[f"{n}.conv2.kernel"], s[f"{n}.conv2.bias"] + s[f"{n}.skip.bias"], conv=True, w_exp=e)
                    ks = self._dev(s[f"{n}.skip.kernel"])
                    cin_s, cout_s = int(ks.shape[-2]), int(ks.shape[-1])
                    img = torch.empty(lib().dm3d_packed_weight_skip_h3p_bytes(cin_s, cout_s) // 2, dtype=torch.float16, device=self.device)
                    check(lib().dm3d_pack_weights_skip_h3p(ks.data_ptr(), cin_s, cout_s, e, img.data_ptr(), _stream()), "pack_weights_skip_h3p")
                    P[f"{n}.skip_fused"] = img
                    if self.wino:           # the same kernel as operand fragments: the Winograd-x form's register-direct tail (dm3d.h, skip_wpk_frag)
                        frag = torch.empty_like(img)
                        check(lib().dm3d_pack_weights_skip_h3f(ks.data_ptr(), cin_s, cout_s, e, frag.data_ptr(), _stream()), "pack_weights_skip_h3f")
                        P[f"{n}.skip_frag"] = frag
                else:
                    if f"{n}.skip.kernel" in s:
                        P[f"{n}.skip"] = self._pack(s[f"{n}.skip.kernel"], s[f"{n}.skip.bias"], conv=True)
                    P[f"{n}.conv2"] = self._pack(s[f"{n}.conv2.kernel"], s[f"{n}.conv2.bias"], conv=True)
                P[f"{n}.norm1"] = self._fold_bn(f"{n}.norm1")
                P[f"{n}.norm2"] = self._fold_bn(f"{n}.norm2")
                temb_k.append(s[f"{n}.temb.kernel"])
                temb_b.append(s[f"{n}.temb.bias"])
                self.temb_off[n] = off
                off += blk.cout
            elif blk.kind in ("down", "up"):
                P[n] = self._pack(s[f"{n}.kernel"], s[f"{n}.bias"], conv=True, up=blk.kind == "up", stride=2 if blk.kind == "down" else 1)
            elif blk.kind == "attn":
                self._prepare_attn(P, blk)
        self.temb_ld = off
        P["temb_all"] = self._pack(np.concatenate(temb_k, axis=1), np.concatenate(temb_b))
        P["out.norm"] = self._fold_bn("out.norm")
        P["out.conv"] = self._pack(s["out.conv.kernel"], s["out.conv.bias"], conv=True)
        self.P = P
        self._prepared = True
        self.range_limit = self._h3_range_limit()          # (a plan with a Winograd-x launch uses half of it: Plan.range_limit)
        if cfg.conditional:
            self._prepare_context_tables()

    def _h3_range_limit(self, winograd: bool = False) -> float:
        """Bound on |activation| below which no H3 operand path can leave the float16 range (include/dm3d.h, range_flag): a raw
        consumer clamps at 65504; a consumer behind a folded BatchNormalization sees silu(x*scale + shift), |.| <= |x| max|scale| +
        max|shift|.  (GroupNormalization normalises per sample: |x_hat| <= sqrt(group size), no bound on x is needed.)
        ``winograd``: some launch of the plan takes the Winograd-x form, which splits sums of two activations (dm3d.h, wpk_wino): half the range."""
        top = 32752.0 if winograd else 65504.0
        lim = top
        if self.cfg.norm == "batch":
            s = self.state
            for name in s:
                if name.endswith(".gamma") and name[:-6] + ".var" in s:
                    base = name[:-6]
                    scale = np.abs(s[f"{base}.gamma"].astype(np.float64) / np.sqrt(s[f"{base}.var"].astype(np.float64) + BN_EPS))
                    shift = np.abs(s[f"{base}.beta"].astype(np.float64) - s[f"{base}.mean"].astype(np.float64) * scale)
                    smax = float(scale.max())
                    if smax > 0:
                        lim = min(lim, (top - float(shift.max())) / smax)
        return max(lim, 1.0)

    def check_range(self, plan: "Plan", eps: bool = False):
        """Raises if any launch of ``plan`` since the last check produced a value an H3 consumer would have clamped (one 4-byte
        device read: the only host synchronisation of a generate() call, at its end).  ``eps``: also inspect the newest eps (a plain
        forward call: nothing downstream of it would)."""
        if plan.range_flag is None:
            return
        # the newest x (a chain's final latent; nothing downstream would look at it) goes through the NaN-aware check as well: a NaN
        # made anywhere in a step — diverged weights, inf - inf — reaches x with the posterior update (dm3d_ddpm_update propagates it
        # through the clip like tf.clip_by_value), where the next step's range op or this check sees it; `amax > limit` alone is false for one
        for tns in ((plan.x, plan.eps) if eps else (plan.x,)):
            check(lib().dm3d_range_check(tns.data_ptr(), tns.numel(), 65504.0, plan.range_flag.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "range_check")
        if int(plan.range_flag.item()) != 0:
            plan.range_flag.zero_()
            raise _lib.Dm3dError(
                f"an activation exceeded the range the split-float16 (precision='h3') kernels represent exactly (|x| > {plan.range_limit:.4g}), "
                "or a NaN was produced; the result would differ from float32 arithmetic. Rebuild the model with precision='fp32'"
                + (" (or keep 'h3' and set DM3D_CONV_WINO=0: the direct conv form covers |x| <= 65504)." if plan.uses_wino else "."))

    def _prepare_attn(self, P, blk):
        s, n, u = self.state, blk.name, blk.cout
        if self.cfg.conditional:
            if self.cfg.norm == "group":
                P[f"{n}.norm"] = self._fold_bn(f"{n}.norm")                    # (gamma, beta): applied per sample by the plan
                P[f"{n}.proj_in"] = self._with_h2(self._pack(s[f"{n}.proj_in.kernel"], s[f"{n}.proj_in.bias"]))
            else:
                scale, shift = self._fold_bn(f"{n}.norm")
                # BN folded into proj_in: W' = diag(scale) W ; b' = b + shift @ W (the latter through the GEMM kernel)
                plain = self._pack(s[f"{n}.proj_in.kernel"], s[f"{n}.proj_in.bias"])
                bias2 = torch.empty(u, dtype=torch.float32, device=self.device)
                self._gemm_now(a=shift, lda=u, b=plain.wpk, ldb=plain.cin_pad, out=bias2, ldo=u, m=1, n=u, k=u, bias=plain.bias)
                folded = self._pack(s[f"{n}.proj_in.kernel"], None, in_scale=scale)
                folded.bias = bias2
                P[f"{n}.proj_in"] = self._with_h2(folded)
            P[f"{n}.proj_out"] = self._with_h2(self._pack(s[f"{n}.proj_out.kernel"], s[f"{n}.proj_out.bias"]))
            for ln in ("ln1", "ln2", "ln3"):
                P[f"{n}.{ln}"] = (self._dev(s[f"{n}.{ln}.gamma"]), self._dev(s[f"{n}.{ln}.beta"]))
            P[f"{n}.mlp.0"] = self._with_h2(self._pack(s[f"{n}.mlp.0.kernel"], s[f"{n}.mlp.0.bias"]))
            P[f"{n}.mlp.1"] = self._with_h2(self._pack(s[f"{n}.mlp.1.kernel"], s[f"{n}.mlp.1.bias"]))
            for i, w in enumerate((P[f"{n}.mlp.0"], P[f"{n}.mlp.1"])):        # operand-fragment images for dm3d_mlp_fused (u = 256 only)
                w.tiled = None
                if w.h2 is not None and u == 256:
                    w.tiled = torch.empty_like(w.h2)
                    check(lib().dm3d_pack_mlp_weights(w.h2.data_ptr(), u, i, w.tiled.data_ptr(), _stream()), "pack_mlp_weights")
            P[f"{n}.key"] = self._pack(s[f"{n}.key.kernel"], s[f"{n}.key.bias"])
            P[f"{n}.proj_in"].ftiled = self._front_tiles(P[f"{n}.proj_in"], u, u)
            P[f"{n}.proj_out"].ftiled = self._front_tiles(P[f"{n}.proj_out"], u, u)
        else:
            P[f"{n}.norm"] = self._fold_bn(f"{n}.norm")
            P[f"{n}.proj"] = self._with_h2(self._pack(s[f"{n}.proj.kernel"], s[f"{n}.proj.bias"]))
        # query|key share one GEMM (rows 0..u-1 are the query weights and also serve the query-only call)
        P[f"{n}.qk"] = self._with_h2(self._pack(np.concatenate([s[f"{n}.query.kernel"], s[f"{n}.key.kernel"]], axis=1),
                                                np.concatenate([s[f"{n}.query.bias"], s[f"{n}.key.bias"]])))
        P[f"{n}.value"] = self._with_h2(self._pack(s[f"{n}.value.kernel"], s[f"{n}.value.bias"]))
        if self.cfg.conditional:                                               # operand-fragment images for dm3d_attn_front (u = 256 only)
            P[f"{n}.qk"].ftiled = self._front_tiles(P[f"{n}.qk"], 2 * u, u)
            P[f"{n}.value"].ftiled = self._front_tiles(P[f"{n}.value"], u, u)

    def _front_tiles(self, w, n_rows: int, u: int):
        if getattr(w, "h2", None) is None or u != 256 or w.cin_pad != u or w.h2.numel() < n_rows * u:
            return None
        t = torch.empty(n_rows * u, dtype=torch.float32, device=self.device)
        check(lib().dm3d_pack_front_weights(w.h2.data_ptr(), n_rows, u, t.data_ptr(), _stream()), "pack_front_weights")
        return t

    def _gemm_now(self, **kw):
        d = _gemm_desc(**kw)
        check(lib().dm3d_gemm_tn(C.byref(d), _stream()), "gemm")

    def _attn_h2(self, u: int, L: int) -> bool:
        """Attention blocks run on the H3 GEMM with DM3D_FMT_H2 intermediates when the 16-k record granularity fits."""
        return self.precision == "h3" and u % 16 == 0 and L % 16 == 0

    def _prepare_context_tables(self):
        """ContextMLP + key/value projections for both context ids (conditional_dm3d.py:310-318, 168-169, 358):
        per attention block  kctx [ids, L, u]  and  vctx_t [ids, u, L]  (value kept transposed: K-contiguous operand
        of the P.V contraction)."""
        s, td = self.state, self.cfg.temb_dim
        ids = s["ctx_embed.table"].shape[0]
        cemb = self._dev(s["ctx_embed.table"])                               # Embedding rows [ids, td]
        self.ctx_tables = {}
        for blk in self.blocks:
            if blk.kind != "attn":
                continue
            n, u, L = blk.name, blk.cout, blk.edge ** 3
            w = self._pack(s[f"{n}.ctx_mlp.kernel"], s[f"{n}.ctx_mlp.bias"])
            feat = torch.empty(ids, L * u, dtype=torch.float32, device=self.device)
            self._gemm_now(a=cemb, lda=td, b=w.wpk, ldb=w.cin_pad, out=feat, ldo=L * u, m=ids, n=L * u, k=td,
                           bias=w.bias, act=ACT_SILU)
            key, val = self.P[f"{n}.key"], self.P[f"{n}.value"]
            kctx = torch.empty(ids, L, u, dtype=torch.float32, device=self.device)
            self._gemm_now(a=feat, lda=u, b=key.wpk, ldb=key.cin_pad, out=kctx, ldo=u, m=ids * L, n=u, k=u, bias=key.bias)
            vctx_t = torch.empty(ids, u, L, dtype=torch.float32, device=self.device)
            self._gemm_now(a=val.wpk, lda=val.cin_pad, b=feat, ldb=u, stride_b=L * u, out=vctx_t, ldo=L, stride_o=u * L,
                           m=u, n=L, k=u, batch=ids, bias=val.bias, bias_along_m=1)
            if self._attn_h2(u, L):
                kctx = self._to_h2(kctx, ids * L, u).reshape(ids, L, u)
                vctx_t = self._to_h2(vctx_t, ids * u, L).reshape(ids, u, L)
            self.ctx_tables[n] = (kctx, vctx_t)
            del w, feat

    # ---- time path -------------------------------------------------------------------------------------------------
    def fill_time_table(self, t_values: Sequence[int], out: torch.Tensor):
        """out[r, :] = concat over ResidualBlocks of Dense_blk(swish(TimeMLP(TimeEmbedding(t_values[r]))))."""
        self.prepare()
        td, R = self.cfg.temb_dim, len(t_values)
        emb = self._dev(time_embedding_table(np.asarray(t_values), td))
        h = torch.empty(R, td, dtype=torch.float32, device=self.device)
        h2 = torch.empty_like(h)
        m0, m1, ta = self.P["time_mlp.0"], self.P["time_mlp.1"], self.P["temb_all"]
        self._gemm_now(a=emb, lda=td, b=m0.wpk, ldb=m0.cin_pad, out=h, ldo=td, m=R, n=td, k=td, bias=m0.bias, act=ACT_SILU)
        # Dense_1 then the ResidualBlocks' swish: the activation is fused into this GEMM's epilogue
        self._gemm_now(a=h, lda=td, b=m1.wpk, ldb=m1.cin_pad, out=h2, ldo=td, m=R, n=td, k=td, bias=m1.bias, act=ACT_SILU)
        self._gemm_now(a=h2, lda=td, b=ta.wpk, ldb=ta.cin_pad, out=out, ldo=self.temb_ld, m=R, n=self.temb_ld, k=td,
                       bias=ta.bias)

    # ---- plans -----------------------------------------------------------------------------------------------------
    def plan(self, batch: int, vec_rows: int, per_sample_context: bool = False, purpose: str = "forward") -> "Plan":
        """``purpose`` keeps the plan a Sampler drives (its ``vec`` holds the time table of the whole chain) apart from the
        plan ``__call__`` uses (its ``vec`` holds the rows of the caller's t values), also when batch == timesteps."""
        self._fresh()
        self.prepare()
        key = (int(batch), int(vec_rows), bool(per_sample_context), str(purpose))
        if key not in self._plans:
            self._plans[key] = Plan(self, *key[:3])
        return self._plans[key]

    def _call_training(self, inputs) -> torch.Tensor:
        return _training_forward(self, inputs)

    def __call__(self, inputs, training: bool = False) -> torch.Tensor:
        """``network([x, t, context])`` / ``network([x, t])`` as in conditional_dm3d.py:493, 568 (dm3d.py:525)."""
        cfg = self.cfg
        self._fresh()
        if training:
            return self._call_training(inputs)
        if cfg.conditional:
            if len(inputs) != 3:
                raise ValueError("the conditional network takes [image, time, context]")
            x, t, ctx = inputs
        else:
            if len(inputs) != 2:
                raise ValueError("the unconditional network takes [image, time]")
            (x, t), ctx = inputs, None
        x = torch.as_tensor(x)
        want = (cfg.img_size,) * 3 + (cfg.img_channels,)
        if x.dim() != 5 or tuple(x.shape[1:]) != want:
            raise ValueError(f"image_input must be [B,{','.join(map(str, want))}] (NDHWC), got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            raise ValueError("image_input must be float32")
        B = x.shape[0]
        t_host = torch.as_tensor(t).reshape(-1).to("cpu", torch.int64).numpy()
        if t_host.shape[0] != B:
            raise ValueError(f"time_input must have {B} entries")
        per_sample = False
        ctx_host = None
        if cfg.conditional:
            ctx_host = torch.as_tensor(ctx).reshape(-1).to("cpu", torch.int64).numpy()
            if ctx_host.shape[0] not in (1, B):
                raise ValueError(f"context_input must have 1 or {B} rows")
            ids = self.state["ctx_embed.table"].shape[0]
            if ctx_host.min() < 0 or ctx_host.max() >= ids:
                raise ValueError(f"context ids must be in [0,{ids})")
            per_sample = ctx_host.shape[0] == B and B > 1 and len(set(ctx_host.tolist())) > 1
        plan = self.plan(B, B, per_sample)
        if plan.range_flag is not None:
            plan.range_flag.zero_()
        self.fill_time_table(t_host, plan.vec)
        plan.t_idx.copy_(torch.arange(B, dtype=torch.int32))
        if cfg.conditional:
            plan.set_context(ctx_host if per_sample else ctx_host[:1])
        plan.x.copy_(x.to(self.device))
        plan.run()
        out = plan.eps.clone()
        self.check_range(plan, eps=True)
        return out


def _training_forward(net: "UNet", inputs) -> torch.Tensor:
    """``network([x, t, context], training=True)`` (conditional_dm3d.py:493) outside train_step: BatchNormalization normalises with
    the statistics of this batch and updates its moving averages (momentum 0.99), as Keras does whenever training=True."""
    from .train import Trainer
    cfg = net.cfg
    x, t = inputs[0], inputs[1]
    ctx = inputs[2] if cfg.conditional else None
    x = torch.as_tensor(x, dtype=torch.float32).to(net.device).contiguous()
    provider = getattr(net, "_training_engine", None)
    tr = provider() if provider is not None else None          # the owning DiffusionModel's Trainer, if it has trained already
    own = tr is None
    if own:
        tr = Trainer(cfg, net.state_dict(), net.device, forward_only=True)
    ids = None
    if cfg.conditional:
        ids = torch.as_tensor(ctx).reshape(-1).to(net.device, torch.int32)
        if ids.numel() == 1 and x.shape[0] > 1:
            ids = ids.repeat(x.shape[0])
    out = tr.forward(x, torch.as_tensor(t).reshape(-1).cpu().numpy().astype(np.int64), ids, update_moving=True)
    tr.tape = []
    tr._cache = {}
    if own:
        new = tr.state_dict()
        net.load_state_dict({k: (new[k] if k.endswith((".mean", ".var")) else v) for k, v in net.state_dict().items()})
    # (the model's own Trainer: it was marked dirty, the network picks the new statistics up on its next use)
    return out.v.clone()


def _gemm_desc(a, lda, b, ldb, out, ldo, m, n, k, batch=1, stride_a=0, stride_b=0, stride_o=0, alpha=1.0, bias=None,
               bias_along_m=0, act=ACT_NONE, res=None, ldr=0, stride_r=0, a_off=0, b_off=0, out_off=0, res_off=0,
               h3=False, a_h2=False, b_h2=False, out_h2=False, res2=None) -> GemmDesc:
    d = GemmDesc()
    d.a, d.lda, d.stride_a = _ptr(a, a_off), lda, stride_a
    d.b, d.ldb, d.stride_b = _ptr(b, b_off), ldb, stride_b
    d.out, d.ldo, d.stride_o = _ptr(out, out_off), ldo, stride_o
    d.m, d.n, d.k, d.batch = m, n, k, batch
    d.alpha = alpha
    d.bias, d.bias_along_m, d.act = _ptr(bias), bias_along_m, act
    d.res, d.ldr, d.stride_r = _ptr(res, res_off), ldr, stride_r
    d.res2 = _ptr(res2)
    d.precision = _lib.PREC_H3 if h3 else _lib.PREC_F32
    d.a_fmt = _lib.FMT_H2 if a_h2 else _lib.FMT_F32
    d.b_fmt = _lib.FMT_H2 if b_h2 else _lib.FMT_F32
    d.out_fmt = _lib.FMT_H2 if out_h2 else _lib.FMT_F32
    return d


class Plan:
    """All launches of one U-Net forward for a fixed batch, with fixed buffers.

    Inputs live in ``x`` [B,S,S,S,C], ``t_idx`` [B] int32 (row of ``vec`` per sample) and, for the conditional model,
    the per-block context key/value buffers filled by ``set_context``.  Output in ``eps``."""

    def __init__(self, net: UNet, batch: int, vec_rows: int, per_sample_context: bool):
        self.net, self.B, self.per_sample_context = net, batch, per_sample_context
        cfg, dev = net.cfg, net.device
        S, Cc = cfg.img_size, cfg.img_channels
        self.ops: List[tuple] = []
        self._keep: List[object] = []
        self.x = torch.empty(batch, S, S, S, Cc, dtype=torch.float32, device=dev)
        self.eps = torch.empty_like(self.x)
        self.t_idx = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.vec = torch.empty(vec_rows, net.temb_ld, dtype=torch.float32, device=dev)
        self.ctx_bufs: Dict[str, tuple] = {}
        self._gn_stats = {}
        # H3 range guard (include/dm3d.h): every H3 launch of the plan reports into one flag; see UNet.check_range
        self.range_flag = torch.zeros(1, dtype=torch.int32, device=dev) if net.precision == "h3" else None
        self.uses_wino = False                  # some conv of this plan takes the Winograd-x form (dm3d_conv_tile_form() == 10)
        # tickets of the Cin-split launches (include/dm3d.h, split_counters): zero now, and every launch leaves them zero, so one buffer
        # serves every conv of the plan (launches are stream-ordered).  Every conv descriptor carries it from the start: the tile form and
        # the split a descriptor answers for are then the ones the launch takes.
        self.split_counters = torch.zeros(4096, dtype=torch.int32, device=dev)
        self.range_limit = net.range_limit
        self._build()
        if self.range_flag is not None and self.uses_wino:
            # the Winograd-x form splits sums of two activations: every producer of this plan guards half the range.  Decided per plan
            # (a plan whose grids are too small for that form keeps the whole float16 range)
            self.range_limit = net._h3_range_limit(winograd=True)
            for d in self._keep:
                for one in (d if isinstance(d, C.Array) else (d,)):
                    if isinstance(one, (ConvDesc, GemmDesc, MlpDesc, AttnFrontDesc)) and one.range_flag:
                        one.range_limit = self.range_limit
        # one workspace for every conv that can split its Cin range (dm3d_conv_scratch_bytes): launches are stream-ordered
        need = max([lib().dm3d_conv_scratch_bytes(C.byref(d)) for d in self._keep if isinstance(d, ConvDesc)] + [0])
        if need:
            self.scratch = torch.empty(need // 4, dtype=torch.float32, device=dev)
            for d in self._keep:
                if isinstance(d, ConvDesc):
                    d.scratch, d.scratch_bytes = self.scratch.data_ptr(), need
        assert all(lib().dm3d_conv_split_counter_words(C.byref(d)) <= self.split_counters.numel() for d in self._keep if isinstance(d, ConvDesc))

    # -- buffer / op helpers ---------------------------------------------------------------------------------------
    def _buf(self, *shape) -> torch.Tensor:
        t = torch.empty(*shape, dtype=torch.float32, device=self.net.device)
        self._keep.append(t)
        return t

    # -- GroupNormalization statistics (cfg.norm == "group") -----------------------------------------------------------------
    # Every tensor that gets normalised owns a float32 buffer [B][slots][C][2] of partial (sum, sum of squares) per (sample, channel),
    # one slot per 64 voxels.  A conv producer fills it while it stores the tensor (dm3d_conv_desc.gn_stats: fused in the full-brick
    # epilogue — a wave stores its sums as the slot of its z-slice, no atomics — else by the library behind the launch); a tensor from
    # another producer (the attention blocks' GEMMs) gets one stand-alone pass.  A norm layer then needs one small launch
    # (dm3d_groupnorm_finalize2 over the buffers of its one or two inputs), and a tensor with several consumers (the skip connections)
    # is summed once.  Every slot is rewritten every step: nothing to clear.
    def _stats_alloc(self, t: torch.Tensor, c: int, vox: int) -> int:
        buf = torch.empty(lib().dm3d_groupnorm_partials_bytes(self.B, vox, c) // 4, dtype=torch.float32, device=self.net.device)
        self._keep.append(buf)
        self._gn_stats[t.data_ptr()] = buf.data_ptr()
        return buf.data_ptr()

    def _stats_for(self, t: torch.Tensor, c: int, vox: int) -> int:
        ptr = self._gn_stats.get(t.data_ptr())
        if ptr is None:                    # no conv produced it with gn_stats: one stand-alone pass over the tensor
            ptr = self._stats_alloc(t, c, vox)
            self.ops.append((lib().dm3d_groupnorm_partials, (t.data_ptr(), self.B, vox, c, ptr), "groupnorm", {}))
        return ptr

    def _norm(self, name, x1, c1, edge, x2=None, c2=0):
        """Prologue vectors of a normalisation layer: (pro, batch_stride).  BatchNorm: the folded constants.  GroupNorm: one small
        launch turns the statistics of the input tensor(s) into per-sample scale/shift for this step."""
        P = self.net.P
        if self.net.cfg.norm != "group":
            return P[name], 0
        gamma, beta = P[name]
        ct, B, vox = c1 + c2, self.B, edge ** 3
        scale, shift = self._buf(B, ct), self._buf(B, ct)
        self._keep += [gamma, beta]
        a1 = self._stats_for(x1, c1, vox)
        a2 = self._stats_for(x2, c2, vox) if x2 is not None else None
        self.ops.append((lib().dm3d_groupnorm_finalize2, (a1, c1, a2, c2, B, vox, self.net.cfg.norm_groups, BN_EPS, gamma.data_ptr(),
                                                          beta.data_ptr(), scale.data_ptr(), shift.data_ptr()), "groupnorm", {}))
        return (scale, shift), ct

    def _conv(self, w: _Conv, x1, out, edge_in, x2=None, c1=None, c2=0, upsample=0, stride=1, pro=None, vec_off=None,
              relu=0, res=None, pro_bstride=0, skip=None, post=None, out_h2=False, x1_h2=False, normed=True):
        """``normed``: the output feeds a normalisation layer (GroupNormalization variant: the launch also sums its statistics)."""
        d = ConvDesc()
        d.x1, d.x2 = _ptr(x1), _ptr(x2)
        d.c1, d.c2 = (c1 if c1 is not None else w.cin), c2
        d.batch = self.B
        d.in_d = d.in_h = d.in_w = edge_in
        d.upsample, d.stride = upsample, stride
        d.ksize = {1: 1, 27: 3}[w.taps]
        d.wpk, d.bias = w.wpk.data_ptr(), _ptr(w.bias)
        d.precision, d.w_exp = w.precision, w.w_exp
        if w.precision == _lib.PREC_H3:
            d.w_layout = lib().dm3d_conv_weight_layout(d.ksize, stride, upsample, 0, w.cout)
        if pro is not None:
            d.pro_scale, d.pro_shift, d.pro_batch_stride = _ptr(pro[0]), _ptr(pro[1]), pro_bstride
        if vec_off is not None:
            d.vec, d.vec_idx, d.vec_ld = _ptr(self.vec, vec_off), _ptr(self.t_idx), self.net.temb_ld
        d.relu, d.res, d.out, d.cout = relu, _ptr(res), _ptr(out), w.cout
        if d.c1 + d.c2 != w.cin:
            raise ValueError(f"conv input channels {d.c1}+{d.c2} != weight cin {w.cin}")
        if post is not None:
            d.post_scale, d.post_shift = _ptr(post[0]), _ptr(post[1])
            self._keep += [post[0], post[1]]
        d.out_fmt = _lib.FMT_H2 if out_h2 else _lib.FMT_F32
        d.x1_fmt = _lib.FMT_H2 if x1_h2 else _lib.FMT_F32
        if w.wpk_wino is not None:
            d.wpk_wino = w.wpk_wino.data_ptr()
        if self.range_flag is not None and w.precision == _lib.PREC_H3:
            d.range_flag, d.range_limit = self.range_flag.data_ptr(), self.range_limit
        if w.precision == _lib.PREC_H3:
            d.split_counters, d.split_counter_words = self.split_counters.data_ptr(), self.split_counters.numel()
        if self.net.cfg.norm == "group" and normed and not out_h2 and w.cout % 4 == 0:
            d.gn_stats = self._stats_alloc(out, w.cout, out.numel() // (self.B * w.cout))
        skip_flops = 0.0
        if skip is not None:
            sx1, sx2, sc1, sc2, simg, sfrag = skip
            d.skip_x1, d.skip_x2, d.skip_c1, d.skip_c2, d.skip_wpk = _ptr(sx1), _ptr(sx2), sc1, sc2, simg.data_ptr()
            if sfrag is not None:
                d.skip_wpk_frag = sfrag.data_ptr()
            skip_flops = 2.0 * (sc1 + sc2) * w.cout * self.B * edge_in ** 3
        self._keep.append(d)
        up = 2 if upsample else 1
        eo = -(-edge_in * up // stride)
        form = lib().dm3d_conv_tile_form(C.byref(d)) if w.precision == _lib.PREC_H3 else 0
        # kinds follow the kernel instantiations rocprofv3 lists, so bench.py's per-kernel averages can be checked against it
        if w.taps == 1:
            kind = "conv_k1"
        elif stride == 2:
            kind = "conv_k3s2"
        elif form == 10:
            self.uses_wino = True
            kind = "conv_wino_h2in" if x1_h2 else "conv_wino"                                # conv3d_igemm_h3w<MODE>: Winograd F(2,3) along x
        elif upsample:
            kind = "conv_up"            # 8 parity 2x2x2 convs
        elif w.cout <= 32 and w.precision == _lib.PREC_H3:
            kind = "conv_k3s1_n32"      # conv_in / conv_out: one 32-column tile
        elif x1_h2:
            kind = "conv_k3s1_h2in"     # kernel MODE 2: pre-activated DM3D_FMT_H2 input (ResidualBlock conv2 behind a hand-off)
        elif w.precision == _lib.PREC_H3 and pro is not None and form == 4:
            kind = "conv_k3s1_td4"      # conv3d_igemm_h3v3<3, 1, 4>: 4-slice bricks (small grids, fused skip phase)
        else:
            kind = "conv_k3s1"          # h3: conv3d_igemm_h3v3<3, 1, 8> (prologue, 8-slice bricks); fp32: conv3d_igemm_f32
        self.ops.append((lib().dm3d_conv3d_ndhwc, (C.byref(d),), kind,
                         {"desc": f"{kind} {edge_in}^3{'x2up' if upsample else ''} cin={w.cin} cout={w.cout}"
                                  + (f" +k1 skip cin={skip[2] + skip[3]}" if skip is not None else ""),
                          # algorithmic; a fused 1x1 skip conv counts its own FLOPs here
                          "flops": 2.0 * w.taps * w.cin * w.cout * self.B * eo ** 3 + skip_flops,
                          # MFMA work actually issued: the upsample conv runs as 8 parity convs of 8 taps on the low-res grid
                          # (Winograd-x form: 40 k-steps per output pair — 36 + the zero pad tap — against 54, three passes each;
                          #  "useful_flops" leaves the pad steps out: 36 / 54)
                          "exec_flops": (2.0 * (8 if upsample else w.taps) * w.cin * w.cout * self.B * eo ** 3 * (40 / 54 if form == 10 else 1) + skip_flops)
                                        * (3 if w.precision == _lib.PREC_H3 else 1),
                          "useful_flops": (2.0 * (8 if upsample else w.taps) * w.cin * w.cout * self.B * eo ** 3 * (36 / 54 if form == 10 else 1) + skip_flops)
                                          * (3 if w.precision == _lib.PREC_H3 else 1),
                          "bytes": 4.0 * self.B * (edge_in ** 3 * (w.cin + (skip[2] + skip[3] if skip is not None else 0))
                                                   + eo ** 3 * w.cout)}))

    def _guard(self, d: GemmDesc) -> GemmDesc:
        if self.range_flag is not None and d.precision == _lib.PREC_H3:
            d.range_flag, d.range_limit = self.range_flag.data_ptr(), self.range_limit
        return d

    def _gemm(self, **kw):
        d = self._guard(_gemm_desc(**kw))
        self._keep.append(d)
        kind = "gemm_h3" if d.precision == _lib.PREC_H3 else "gemm"
        self.ops.append((lib().dm3d_gemm_tn, (C.byref(d),), kind,
                         {"desc": f"{kind} m={d.m} n={d.n} k={d.k} batch={d.batch}", "flops": 2.0 * d.m * d.n * d.k * d.batch}))

    def _gemm_group(self, problems):
        """Independent H3 GEMMs with identical operand formats as one launch (dm3d_gemm_tn_group)."""
        arr = (GemmDesc * len(problems))(*[self._guard(_gemm_desc(**kw)) for kw in problems])
        self._keep.append(arr)
        fl = sum(2.0 * d.m * d.n * d.k * d.batch for d in arr)
        desc = "gemm_h3 group[" + " | ".join(f"m={d.m} n={d.n} k={d.k} b={d.batch}" for d in arr) + "]"
        self.ops.append((lib().dm3d_gemm_tn_group, (arr, len(problems)), "gemm_h3", {"desc": desc, "flops": fl}))

    # -- graph -----------------------------------------------------------------------------------------------------
    def _build(self):
        net, cfg, B = self.net, self.net.cfg, self.B
        P = net.P
        S = cfg.img_size
        if self.range_flag is not None:        # the caller's x_t is the one tensor on an H3 operand path that no dm3d kernel wrote
            self.ops.append((lib().dm3d_range_check, (self.x.data_ptr(), self.x.numel(), 65504.0,
                                                       self.range_flag.data_ptr()), "range", {}))
        h = self._buf(B, S, S, S, cfg.first_conv_channels)
        self._conv(P["conv_in"], self.x, h, S)
        skips = [(h, cfg.first_conv_channels)]
        cur, cur_c, edge = h, cfg.first_conv_channels, S
        # The last ResidualBlock's output has one consumer, BatchNormalization -> swish -> the output conv (conditional_dm3d.py:409-412):
        # with the norm folded it leaves that block's conv2 already normalised, activated and split (DM3D_FMT_H2, as conv1 -> conv2 inside a
        # block), and the output conv — 8 useful columns: bound by converting its input, not by its MFMAs — stages plain copies.
        last = net.blocks[-1] if net.blocks else None
        final_handoff = (last is not None and last.kind == "res" and net.h2_handoff and cfg.norm == "batch" and net.precision == "h3"
                         and S % 8 == 0 and last.cout % 64 == 0 and B * (S ** 3 // 256) * (last.cout // 64) > 256
                         and lib().dm3d_conv_weight_layout(3, 1, 0, 0, last.cout) == _lib.WL_PAIR
                         and lib().dm3d_conv_weight_layout(3, 1, 0, 0, cfg.img_channels) == _lib.WL_PAIR)
        for blk in net.blocks:
            if blk.kind == "push":
                skips.append((cur, cur_c))
            elif blk.kind == "res":
                x2, c2 = (None, 0)
                if blk.cskip:
                    x2, c2 = skips.pop()
                    if c2 != blk.cskip:
                        raise AssertionError("skip channel mismatch")
                cur = self._res_block(blk, cur, blk.cin, x2, c2, edge, final_post=P["out.norm"] if (final_handoff and blk is last) else None)
                cur_c = blk.cout
            elif blk.kind == "attn":
                cur = self._cross_block(blk, cur, edge) if cfg.conditional else self._self_block(blk, cur, edge)
            elif blk.kind == "down":
                out = self._buf(B, blk.edge, blk.edge, blk.edge, blk.cout)
                self._conv(P[blk.name], cur, out, edge, stride=2)
                cur, edge = out, blk.edge
            elif blk.kind == "up":
                out = self._buf(B, blk.edge, blk.edge, blk.edge, blk.cout)
                self._conv(P[blk.name], cur, out, edge, upsample=1)
                cur, edge = out, blk.edge
        if final_handoff:
            self._conv(P["out.conv"], cur, self.eps, edge, x1_h2=True, normed=False)
            return
        pro, bs = self._norm("out.norm", cur, cur_c, edge)
        self._conv(P["out.conv"], cur, self.eps, edge, pro=pro, pro_bstride=bs, normed=False)

    def _res_block(self, blk, x1, c1, x2, c2, edge, final_post=None):
        """ResidualBlock (conditional_dm3d.py:238-271): three launches (two when the widths match).  ``final_post``: the folded norm of
        the block's only consumer; the block's output then leaves conv2 normalised, activated and split (DM3D_FMT_H2)."""
        P, B, n, w = self.net.P, self.B, blk.name, blk.cout
        skip = None
        if f"{n}.skip_fused" in P:
            res, skip = None, (x1, x2, c1, c2, P[f"{n}.skip_fused"], P.get(f"{n}.skip_frag"))
        elif f"{n}.skip" in P:
            res = self._buf(B, edge, edge, edge, w)
            self._conv(P[f"{n}.skip"], x1, res, edge, x2=x2, c1=c1, c2=c2, normed=False)
        else:
            res = x1
        hmid = self._buf(B, edge, edge, edge, w)
        pro, bs = self._norm(f"{n}.norm1", x1, c1, edge, x2, c2)
        out = self._buf(B, edge, edge, edge, w)
        # conv1's output has one consumer: with folded BatchNormalization it leaves conv1 already normalised, activated and split
        # (DM3D_FMT_H2) and conv2 stages plain copies.  Needs the 16x16x32 kernel, whole bricks, and a grid that would not
        # rather split its Cin range (the fused output forms live in the unsplit epilogue).
        handoff = (self.net.h2_handoff and self.net.cfg.norm == "batch" and self.net.precision == "h3" and edge % 8 == 0
                   and w % 64 == 0 and B * (edge ** 3 // 256) * (w // 64) > 256
                   and lib().dm3d_conv_weight_layout(3, 1, 0, 0, w) == _lib.WL_PAIR)
        tail = dict(post=final_post, out_h2=True) if final_post is not None else {}
        if handoff:
            self._conv(P[f"{n}.conv1"], x1, hmid, edge, x2=x2, c1=c1, c2=c2, pro=pro, pro_bstride=bs, vec_off=self.net.temb_off[n],
                       post=P[f"{n}.norm2"], out_h2=True)
            self._conv(P[f"{n}.conv2"], hmid, out, edge, res=res, skip=skip, x1_h2=True, **tail)
            return out
        self._conv(P[f"{n}.conv1"], x1, hmid, edge, x2=x2, c1=c1, c2=c2, pro=pro, pro_bstride=bs, vec_off=self.net.temb_off[n])
        pro, bs = self._norm(f"{n}.norm2", hmid, w, edge)
        self._conv(P[f"{n}.conv2"], hmid, out, edge, pro=pro, pro_bstride=bs, res=res, skip=skip, **tail)
        return out

    def _group_normed(self, n, x, u, edge):
        """GroupNormalization of an attention block's input, materialised (its consumers are GEMMs and a residual)."""
        (scale, shift), _ = self._norm(f"{n}.norm", x, u, edge)
        xn = self._buf(self.B * edge ** 3, u)
        self.ops.append((lib().dm3d_affine_act_batched, (x.data_ptr(), xn.data_ptr(), self.B, edge ** 3, u, scale.data_ptr(),
                                                         shift.data_ptr(), ACT_NONE), "affine", {}))
        return xn

    def _attn_core(self, q, q_ld, q_off, k, k_ld, k_off, k_stride, v_t, v_ld, v_off, v_stride, scores, res, out, L, u, h2):
        """softmax(q k^T * u^-0.5) v + res per sample (conditional_dm3d.py:171-180): two batched GEMMs around a
        wavefront-shuffle row softmax.  h2: q, k, v_t arrive in DM3D_FMT_H2 and the probabilities are left in H2."""
        B = self.B
        self._gemm(a=q, a_off=q_off, lda=q_ld, stride_a=L * q_ld, b=k, b_off=k_off, ldb=k_ld, stride_b=k_stride,
                   out=scores, ldo=L, stride_o=L * L, m=L, n=L, k=u, batch=B, alpha=float(u) ** -0.5, h3=h2, a_h2=h2, b_h2=h2)
        fn = lib().dm3d_softmax_rows_h2 if h2 else lib().dm3d_softmax_rows
        self.ops.append((fn, (scores.data_ptr(), B * L, L, L), "softmax", {}))
        self._gemm(a=scores, lda=L, stride_a=L * L, b=v_t, b_off=v_off, ldb=v_ld, stride_b=v_stride, out=out, ldo=u,
                   stride_o=L * u, m=L, n=u, k=L, batch=B, res=res, ldr=u, stride_r=L * u, h3=h2, a_h2=h2, b_h2=h2)

    @staticmethod
    def _check_tokens(L: int):
        if L % 4:
            raise ValueError(f"attention over {L} tokens: the P.V contraction runs on the MFMA GEMM, which needs D*H*W % 4 == 0 at "
                             "attention levels (edge 3 or 5 are not supported; the reference's latents give 4^3, 8^3, 16^3)")

    def _cross_block(self, blk, x, edge):
        """CrossAttentionBlock (conditional_dm3d.py:186-195).  With precision "h3" every intermediate that only feeds
        Dense layers (LayerNorm outputs, q|k, v^T, probabilities, MLP hidden, a3) lives in DM3D_FMT_H2."""
        P, B, n, u = self.net.P, self.B, blk.name, blk.cout
        L = edge ** 3
        M = B * L
        self._check_tokens(L)
        h2 = self.net._attn_h2(u, L)
        if h2:
            return self._cross_block_h2(blk, x, edge)
        W = (lambda w: w.h2) if h2 else (lambda w: w.wpk)
        pin, pout, qk, val = P[f"{n}.proj_in"], P[f"{n}.proj_out"], P[f"{n}.qk"], P[f"{n}.value"]
        m0, m1 = P[f"{n}.mlp.0"], P[f"{n}.mlp.1"]
        y = self._buf(M, u)                                                   # relu(proj_in(norm(x))), float32
        xin = self._group_normed(n, x, u, edge) if self.net.cfg.norm == "group" else x      # BatchNorm is folded into proj_in
        self._gemm(a=xin, lda=u, b=W(pin), ldb=pin.cin_pad, out=y, ldo=u, m=M, n=u, k=u, bias=pin.bias, act=ACT_RELU,
                   h3=h2, b_h2=h2)
        n1, n2, n3 = self._buf(M, u), self._buf(M, u), self._buf(M, u)
        (g1, b1), (g2, b2), (g3, b3) = P[f"{n}.ln1"], P[f"{n}.ln2"], P[f"{n}.ln3"]
        self._keep += [g1, b1, g2, b2, g3, b3]
        ln = lib().dm3d_layernorm3_h2 if h2 else lib().dm3d_layernorm3
        self.ops.append((ln, (y.data_ptr(), M, u, LN_EPS, g1.data_ptr(), b1.data_ptr(), n1.data_ptr(), g2.data_ptr(),
                              b2.data_ptr(), n2.data_ptr(), g3.data_ptr(), b3.data_ptr(), n3.data_ptr()), "layernorm", {}))
        # self attention on norm1(y)
        qkb = self._buf(M, 2 * u)
        self._gemm(a=n1, lda=u, b=W(qk), ldb=qk.cin_pad, out=qkb, ldo=2 * u, m=M, n=2 * u, k=u, bias=qk.bias,
                   h3=h2, a_h2=h2, b_h2=h2, out_h2=h2)
        v_t = self._buf(u, M)                                                 # value projection, transposed
        self._gemm(a=W(val), lda=val.cin_pad, b=n1, ldb=u, out=v_t, ldo=M, m=u, n=M, k=u, bias=val.bias, bias_along_m=1,
                   h3=h2, a_h2=h2, b_h2=h2, out_h2=h2)
        scores = self._buf(B, L, L)
        a1 = self._buf(M, u)
        self._attn_core(qkb, 2 * u, 0, qkb, 2 * u, u, L * 2 * u, v_t, M, 0, L, scores, y, a1, L, u, h2)
        # cross attention: queries from norm2(y), keys/values from the context (same key/value weights, :168-169)
        q2 = self._buf(M, u)
        self._gemm(a=n2, lda=u, b=W(qk), ldb=qk.cin_pad, out=q2, ldo=u, m=M, n=u, k=u, bias=qk.bias,
                   h3=h2, a_h2=h2, b_h2=h2, out_h2=h2)
        rows = B if self.per_sample_context else 1
        kctx, vctx_t = self._buf(rows, L * u), self._buf(rows, u * L)
        self.ctx_bufs[n] = (kctx, vctx_t)
        a2 = self._buf(M, u)
        ks, vs = (L * u, u * L) if self.per_sample_context else (0, 0)
        self._attn_core(q2, u, 0, kctx, u, 0, ks, vctx_t, L, 0, vs, scores, a1, a2, L, u, h2)
        # MLP on norm3(y)
        hid = self._buf(M, 4 * u)
        self._gemm(a=n3, lda=u, b=W(m0), ldb=m0.cin_pad, out=hid, ldo=4 * u, m=M, n=4 * u, k=u, bias=m0.bias, act=ACT_RELU,
                   h3=h2, a_h2=h2, b_h2=h2, out_h2=h2)
        a3 = self._buf(M, u)
        self._gemm(a=hid, lda=4 * u, b=W(m1), ldb=m1.cin_pad, out=a3, ldo=u, m=M, n=u, k=4 * u, bias=m1.bias, res=a2, ldr=u,
                   h3=h2, a_h2=h2, b_h2=h2, out_h2=h2)
        out = self._buf(B, edge, edge, edge, u)
        self._gemm(a=a3, lda=u, b=W(pout), ldb=pout.cin_pad, out=out, ldo=u, m=M, n=u, k=u, bias=pout.bias, act=ACT_RELU,
                   res=x, ldr=u, h3=h2, a_h2=h2, b_h2=h2)
        return out

    def _cross_block_h2(self, blk, x, edge):
        """CrossAttentionBlock with DM3D_FMT_H2 intermediates (conditional_dm3d.py:163-195).  Large batches (u = 256, >= 128 row tiles): THREE
        launches — dm3d_attn_front (proj_in + the three LayerNormalizations + q|k, v^T, q2), dm3d_attention_group (both passes, online softmax),
        dm3d_mlp_fused with the proj_out tail; a3 = MLP + (attn_self + y) + attn_cross is formed inside the last.  Otherwise the GEMM form: the
        GEMMs that depend only on the LayerNorm outputs (q|k, v^T, q2, MLP hidden) as one grouped launch, likewise the two score products and
        the two P.V products."""
        P, B, n, u = self.net.P, self.B, blk.name, blk.cout
        L = edge ** 3
        M = B * L
        pin, pout, qk, val = P[f"{n}.proj_in"], P[f"{n}.proj_out"], P[f"{n}.qk"], P[f"{n}.value"]
        m0, m1 = P[f"{n}.mlp.0"], P[f"{n}.mlp.1"]
        hh = dict(h3=True, a_h2=True, b_h2=True)
        y = self._buf(M, u)                                                   # relu(proj_in(norm(x))), float32
        xin = self._group_normed(n, x, u, edge) if self.net.cfg.norm == "group" else x      # BatchNorm is folded into proj_in
        (g1, b1), (g2, b2), (g3, b3) = P[f"{n}.ln1"], P[f"{n}.ln2"], P[f"{n}.ln3"]
        self._keep += [g1, b1, g2, b2, g3, b3]
        # proj_in + the three LayerNormalizations + the q|k, v^T, q2 projections as ONE launch (dm3d_attn_front, round 4): y, n1, n2 stay on the
        # CU.  Same shape rule as the fused MLP (u = 256, enough 64-row tiles to fill the chip); otherwise the three launches below.
        min_rows = int(os.environ.get("DM3D_FUSED_MIN_ROWS", str(64 * 128)))
        front = (u == 256 and M % 64 == 0 and M >= min_rows and all(getattr(w, "ftiled", None) is not None for w in (pin, qk, val))
                 and os.environ.get("DM3D_ATTN_FRONT", "1") != "0")
        n3 = self._buf(M, u)
        if front:
            qkb, v_t, q2 = self._buf(M, 2 * u), self._buf(u, M), self._buf(M, u)
            fd = AttnFrontDesc()
            fd.x, fd.ldx = _ptr(xin), u
            fd.w_in, fd.b_in, fd.w_qk, fd.b_qk, fd.w_v, fd.b_v = (pin.ftiled.data_ptr(), _ptr(pin.bias), qk.ftiled.data_ptr(), _ptr(qk.bias),
                                                                  val.ftiled.data_ptr(), _ptr(val.bias))
            fd.g1, fd.be1, fd.g2, fd.be2, fd.g3, fd.be3 = g1.data_ptr(), b1.data_ptr(), g2.data_ptr(), b2.data_ptr(), g3.data_ptr(), b3.data_ptr()
            fd.eps = LN_EPS
            fd.y, fd.ldy, fd.qk, fd.ldqk, fd.vt, fd.ldvt, fd.q2, fd.ldq2, fd.n3, fd.ldn3 = _ptr(y), u, _ptr(qkb), 2 * u, _ptr(v_t), M, _ptr(q2), u, _ptr(n3), u
            fd.m, fd.units = M, u
            if self.range_flag is not None:
                fd.range_flag, fd.range_limit = self.range_flag.data_ptr(), self.range_limit
            self._keep.append(fd)
            self.ops.append((lib().dm3d_attn_front, (C.byref(fd),), "attn_front",
                             {"desc": f"attn_front m={M} u={u} (proj_in + 3 LayerNorm + q|k, v^T, q2)", "flops": 2.0 * M * u * 5 * u,
                              "bufs": {"x": xin, "y": y, "qk": qkb, "v_t": v_t, "q2": q2, "n3": n3}}))
        else:
            self._gemm(a=xin, lda=u, b=pin.h2, ldb=pin.cin_pad, out=y, ldo=u, m=M, n=u, k=u, bias=pin.bias, act=ACT_RELU,
                       h3=True, b_h2=True)
            n1, n2 = self._buf(M, u), self._buf(M, u)
            self.ops.append((lib().dm3d_layernorm3_h2, (y.data_ptr(), M, u, LN_EPS, g1.data_ptr(), b1.data_ptr(), n1.data_ptr(),
                                                        g2.data_ptr(), b2.data_ptr(), n2.data_ptr(), g3.data_ptr(), b3.data_ptr(),
                                                        n3.data_ptr()), "layernorm", {}))
        # the MLP (Dense(4u, relu) -> Dense(u), :132-133) as ONE launch with the hidden activation in LDS (dm3d_mlp_fused, round 4) where the
        # kernel's shape fits (u = 256 and enough row tiles to fill the chip); otherwise its first Dense joins the grouped launch below
        mlp_fused = u == 256 and M % 64 == 0 and M >= min_rows and getattr(m0, "tiled", None) is not None and os.environ.get("DM3D_MLP_FUSED", "1") != "0"
        hid = None if mlp_fused else self._buf(M, 4 * u)
        group = []
        if not front:
            qkb, v_t, q2 = self._buf(M, 2 * u), self._buf(u, M), self._buf(M, u)
            group += [
                dict(a=n1, lda=u, b=qk.h2, ldb=qk.cin_pad, out=qkb, ldo=2 * u, m=M, n=2 * u, k=u, bias=qk.bias, out_h2=True, **hh),
                dict(a=val.h2, lda=val.cin_pad, b=n1, ldb=u, out=v_t, ldo=M, m=u, n=M, k=u, bias=val.bias, bias_along_m=1,
                     out_h2=True, **hh),
                dict(a=n2, lda=u, b=qk.h2, ldb=qk.cin_pad, out=q2, ldo=u, m=M, n=u, k=u, bias=qk.bias, out_h2=True, **hh)]
        if not mlp_fused:
            group.append(dict(a=n3, lda=u, b=m0.h2, ldb=m0.cin_pad, out=hid, ldo=4 * u, m=M, n=4 * u, k=u, bias=m0.bias, act=ACT_RELU,
                              out_h2=True, **hh))
        if group:
            self._gemm_group(group)
        rows = B if self.per_sample_context else 1
        kctx, vctx_t = self._buf(rows, L * u), self._buf(rows, u * L)
        self.ctx_bufs[n] = (kctx, vctx_t)
        ks, vs = (L * u, u * L) if self.per_sample_context else (0, 0)
        a1, a2 = self._buf(M, u), self._buf(M, u)
        scale = float(u) ** -0.5
        # (the fused kernel runs one 4-wave workgroup per 128 queries: it needs >= ~128 workgroups to beat the GEMM form, i.e. B >= 16 at L = 512;
        # measured B = 1 / 4 / 8: 2.80 / 3.85 / 5.74 ms per step fused against 2.5 / 3.5 / 5.3 with the three launches)
        if os.environ.get("DM3D_ATTN_FUSED", "1") != "0" and u == 256 and L % 128 == 0 and 2 * B * (L // 128) >= 128:
            # both attention passes in ONE fused launch (dm3d_attention_group -> csrc/dm3d_attn_h3.hip): scores, online softmax and
            # P.V per 32-key tile in registers / LDS; no [B, L, L] tensor
            descs = (_lib.AttentionDesc * 2)()
            for d, (qt, qoff, ldq, kt, koff, ldk, sk, vt, ldv, sv, out, res) in zip(descs, (
                    (qkb, 0, 2 * u, qkb, u, 2 * u, L * 2 * u, v_t, M, L, a1, y),
                    (q2, 0, u, kctx, 0, u, ks, vctx_t, L, vs, a2, None))):
                d.q, d.ldq = _ptr(qt, qoff), ldq
                d.k, d.ldk, d.stride_k = _ptr(kt, koff), ldk, sk
                d.vt, d.ldv, d.stride_vt = _ptr(vt), ldv, sv
                d.out, d.ldo, d.res = _ptr(out), u, _ptr(res)
                d.batch, d.lq, d.lk, d.c, d.scale, d.precision, d.fmt = B, L, L, u, scale, _lib.PREC_H3, _lib.FMT_H2
            self._keep.append(descs)
            self.ops.append((lib().dm3d_attention_group, (descs, 2, None), "attn_fused",
                             {"desc": f"attn_fused 2 passes B={B} L={L} c={u}", "flops": 2 * 2 * 2.0 * B * L * L * u,
                              "bufs": {"a1": a1, "a2": a2}}))
        else:
            scores = self._buf(2, B, L, L)
            s2 = B * L * L                                                        # element offset of the cross-attention scores
            self._gemm_group([
                dict(a=qkb, lda=2 * u, stride_a=L * 2 * u, b=qkb, b_off=u, ldb=2 * u, stride_b=L * 2 * u, out=scores, ldo=L,
                     stride_o=L * L, m=L, n=L, k=u, batch=B, alpha=scale, **hh),
                dict(a=q2, lda=u, stride_a=L * u, b=kctx, ldb=u, stride_b=ks, out=scores, out_off=s2, ldo=L, stride_o=L * L,
                     m=L, n=L, k=u, batch=B, alpha=scale, **hh),
            ])
            self.ops.append((lib().dm3d_softmax_rows_h2, (scores.data_ptr(), 2 * B * L, L, L), "softmax", {}))
            self._gemm_group([
                dict(a=scores, lda=L, stride_a=L * L, b=v_t, ldb=M, stride_b=L, out=a1, ldo=u, stride_o=L * u, m=L, n=u, k=L,
                     batch=B, res=y, ldr=u, stride_r=L * u, **hh),
                dict(a=scores, a_off=s2, lda=L, stride_a=L * L, b=vctx_t, ldb=L, stride_b=vs, out=a2, ldo=u, stride_o=L * u,
                     m=L, n=u, k=L, batch=B, **hh),
            ])
        # ... and the block's proj_out (Conv3D(units, 1, relu) + the block input, :195) as a tail of the same launch: a3 never leaves the CU
        tail = mlp_fused and getattr(pout, "ftiled", None) is not None and os.environ.get("DM3D_MLP_TAIL", "1") != "0"
        out = self._buf(B, edge, edge, edge, u)
        a3 = None if tail else self._buf(M, u)
        if mlp_fused:
            d = MlpDesc()
            d.x, d.ldx, d.w0, d.b0, d.w1, d.b1 = _ptr(n3), u, m0.tiled.data_ptr(), _ptr(m0.bias), m1.tiled.data_ptr(), _ptr(m1.bias)
            d.res, d.res2, d.ldr, d.m, d.units = _ptr(a1), _ptr(a2), u, M, u
            if tail:
                d.out, d.ldo, d.out_fmt = _ptr(out), u, _lib.FMT_F32
                d.w2, d.b2, d.res3, d.ldr3 = pout.ftiled.data_ptr(), _ptr(pout.bias), _ptr(x), u
            else:
                d.out, d.ldo, d.out_fmt = _ptr(a3), u, _lib.FMT_H2
            if self.range_flag is not None:
                d.range_flag, d.range_limit = self.range_flag.data_ptr(), self.range_limit
            self._keep.append(d)
            self.ops.append((lib().dm3d_mlp_fused, (C.byref(d),), "mlp_fused",
                             {"desc": f"mlp_fused m={M} u={u} hidden={4 * u}" + (" + proj_out" if tail else ""),
                              "flops": 2 * 2.0 * M * u * 4 * u + (2.0 * M * u * u if tail else 0.0),
                              "bufs": {"out": out if tail else a3}}))
        else:
            self._gemm(a=hid, lda=4 * u, b=m1.h2, ldb=m1.cin_pad, out=a3, ldo=u, m=M, n=u, k=4 * u, bias=m1.bias, res=a1, res2=a2,
                       ldr=u, out_h2=True, **hh)
        if not tail:
            self._gemm(a=a3, lda=u, b=pout.h2, ldb=pout.cin_pad, out=out, ldo=u, m=M, n=u, k=u, bias=pout.bias, act=ACT_RELU,
                       res=x, ldr=u, **hh)
        return out

    def _self_block(self, blk, x, edge):
        """AttentionBlock (dm3d.py:39-63): BN(x) + proj(softmax(q k^T u^-0.5) v)."""
        P, B, n, u = self.net.P, self.B, blk.name, blk.cout
        L = edge ** 3
        M = B * L
        self._check_tokens(L)
        h2 = self.net._attn_h2(u, L)
        W = (lambda w: w.h2) if h2 else (lambda w: w.wpk)
        qk, val, proj = P[f"{n}.qk"], P[f"{n}.value"], P[f"{n}.proj"]
        if self.net.cfg.norm == "group":
            xn = self._group_normed(n, x, u, edge)
        else:
            scale, shift = P[f"{n}.norm"]
            self._keep += [scale, shift]
            xn = self._buf(M, u)                                              # float32: also the residual
            self.ops.append((lib().dm3d_affine_act, (x.data_ptr(), xn.data_ptr(), M, u, scale.data_ptr(), shift.data_ptr(),
                                                     ACT_NONE), "affine", {}))
        qkb = self._buf(M, 2 * u)
        self._gemm(a=xn, lda=u, b=W(qk), ldb=qk.cin_pad, out=qkb, ldo=2 * u, m=M, n=2 * u, k=u, bias=qk.bias,
                   h3=h2, b_h2=h2, out_h2=h2)
        v_t = self._buf(u, M)
        self._gemm(a=W(val), lda=val.cin_pad, b=xn, ldb=u, out=v_t, ldo=M, m=u, n=M, k=u, bias=val.bias, bias_along_m=1,
                   h3=h2, a_h2=h2, out_h2=h2)
        scores, o = self._buf(B, L, L), self._buf(M, u)
        self._attn_core(qkb, 2 * u, 0, qkb, 2 * u, u, L * 2 * u, v_t, M, 0, L, scores, None, o, L, u, h2)
        out = self._buf(B, edge, edge, edge, u)
        # o is float32 here (the P.V GEMM writes float32); the projection splits it while staging
        self._gemm(a=o, lda=u, b=W(proj), ldb=proj.cin_pad, out=out, ldo=u, m=M, n=u, k=u, bias=proj.bias, res=xn, ldr=u,
                   h3=h2, b_h2=h2)
        return out

    # -- run ---------------------------------------------------------------------------------------------------------
    def set_context(self, ctx_ids):
        """Copies the context key / transposed-value rows for the given ids (1 row: broadcast over the batch)."""
        ids = torch.as_tensor(np.asarray(ctx_ids, dtype=np.int32)).to(self.net.device)
        self._keep_ids = ids
        st = _stream()
        for n, (kbuf, vbuf) in self.ctx_bufs.items():
            ktab, vtab = self.net.ctx_tables[n]
            for tab, buf in ((ktab, kbuf), (vtab, vbuf)):
                check(lib().dm3d_gather_rows(tab.data_ptr(), tab.shape[0], ids.data_ptr(), buf.data_ptr(), buf.shape[0],
                                             buf.shape[1], st), "gather_rows")

    _RANGE_OF = {"conv_wino": "conv", "conv_wino_h2in": "conv", "conv_k3s1_td4": "conv", "conv_k1": "conv", "conv_k3s2": "conv", "conv_up": "conv", "conv_k3s1_n32": "conv", "conv_k3s1_h2in": "conv", "conv_k3s1": "conv",
                 "gemm": "attn", "gemm_h3": "attn", "mlp_fused": "attn", "attn_front": "attn", "attn_fused": "attn", "softmax": "attn", "layernorm": "attn", "groupnorm": "norm", "affine": "norm",
                 "range": "guard"}

    def run(self, stream: Optional[int] = None):
        st = _stream() if stream is None else stream
        push, pop = _lib.roctx()
        cur = None
        for fn, args, what, _ in self.ops:
            rng = self._RANGE_OF.get(what, what)
            if rng != cur:                              # consecutive launches of one family share a range (no-ops unless DM3D_ROCTX=1)
                if cur is not None:
                    pop()
                push(rng)
                cur = rng
            rc = fn(*args, st)
            if rc != 0:
                pop()
                check(rc, what)
        if cur is not None:
            pop()

    def run_timed(self):
        """Eager run on the current stream with a HIP event pair around every launch.
        Returns [(kind, meta, milliseconds)] in launch order (synchronises once, at the end)."""
        st = _stream()
        evs = []
        for fn, args, what, meta in self.ops:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args, st)
            e1.record()
            if rc != 0:
                check(rc, what)
            evs.append((what, meta, e0, e1))
        torch.cuda.synchronize()
        return [(what, meta, e0.elapsed_time(e1)) for what, meta, e0, e1 in evs]

    def count(self) -> Dict[str, int]:
        out: Dict[str, int] = {}
        for _, _, what, _ in self.ops:
            out[what] = out.get(what, 0) + 1
        return out

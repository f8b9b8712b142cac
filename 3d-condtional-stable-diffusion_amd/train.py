"""``DiffusionModel.train_step`` on the dm3d HIP kernels: training-mode forward, backward and Adam, no torch.autograd.

reference: networks/conditional_dm3d.py:471-510 (train_step), :493 (``network(..., training=True)``: BatchNormalization with batch
statistics), :496-499 (loss), :501-504 (Adam on ``network.trainable_weights``); compile() at main_conditional_dm.py:149-154.

Design.  The weights live in ONE flat float32 device buffer ``theta`` in the reference's Keras layouts (spec order), with flat gradient /
Adam-moment buffers beside it, so the optimizer is one launch.  A step records a *tape*: every layer call runs its HIP kernels forward and
pushes a closure that, given dL/d(output), launches the backward kernels and accumulates into its inputs' gradients and into the flat
gradient buffer.  ``backward()`` replays the closures in reverse.  Everything is exact float32 (DM3D_PREC_F32): gradients span too many
octaves for the float16 hi/lo split the sampling path uses.  Data gradients of Conv3D / Dense reuse the forward kernels on flipped /
transposed weights; weight gradients run on ``dm3d_wgrad`` (MFMA contraction over voxels).  PyTorch only owns memory and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SILU, ConvDesc, GemmDesc, WgradDesc, check, lib
from .betas import time_embedding_table
from .weights import UNetConfig, walk

BN_EPS, LN_EPS, BN_MOMENTUM = 1e-3, 1e-3, 0.99          # Keras defaults
ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-7     # keras.optimizers.Adam defaults


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def is_trainable(name: str) -> bool:
    """network.trainable_weights: everything except the BatchNormalization moving statistics."""
    return not name.endswith((".mean", ".var"))


class Param:
    """A view of the flat parameter / gradient buffers."""

    def __init__(self, name, shape, w, g):
        self.name, self.shape, self.w, self.g = name, tuple(shape), w, g


class Var:
    """A tensor on the tape: value, gradient (allocated on first use), whether anything upstream wants a gradient."""

    __slots__ = ("v", "g", "needs_grad")

    def __init__(self, v: torch.Tensor, needs_grad: bool = True):
        self.v, self.g, self.needs_grad = v, None, needs_grad

    def grad_buffer(self) -> torch.Tensor:
        if self.g is None:
            self.g = _zeros_like(self.v)
        return self.g

    def add_grad(self, t: torch.Tensor):
        """Takes ownership of ``t`` when it is the first contribution."""
        if not self.needs_grad:
            return
        if self.g is None:
            self.g = t
        else:
            check(lib().dm3d_axpy(self.g.data_ptr(), t.data_ptr(), t.numel(), 1.0, _st()), "axpy")


def _empty(*shape, device) -> torch.Tensor:
    return torch.empty(*shape, dtype=torch.float32, device=device)


def _zeros_like(t: torch.Tensor) -> torch.Tensor:
    z = torch.empty_like(t)
    check(lib().dm3d_fill(z.data_ptr(), z.numel(), 0.0, _st()), "fill")
    return z


def _zeros(*shape, device) -> torch.Tensor:
    z = _empty(*shape, device=device)
    check(lib().dm3d_fill(z.data_ptr(), z.numel(), 0.0, _st()), "fill")
    return z


class Trainer:
    """Owns the trainable state of one U-Net and runs train steps on it."""

    def __init__(self, cfg: UNetConfig, state: Dict[str, np.ndarray], device, lr: float = 1e-4, bn_moving_unbiased: bool = True,
                 forward_only: bool = False):
        """``forward_only``: no gradient / Adam buffers (a quarter of the memory) — for ``network(..., training=True)`` outside train_step."""
        if cfg.norm != "batch":
            raise ValueError("training is built for the BatchNormalization network the reference trains (norm='batch')")
        _lib.require_device()
        self.cfg, self.device, self.lr = cfg, torch.device(device), float(lr)
        self.bn_moving_unbiased = bool(bn_moving_unbiased)
        self.blocks, self.spec = walk(cfg)
        names = [n for n in self.spec if is_trainable(n)]
        sizes = [int(np.prod(self.spec[n])) for n in names]
        # every parameter starts on a 16-byte boundary (the kernels read rows as float4)
        offs, off = [], 0
        for sz in sizes:
            offs.append(off)
            off += -(-sz // 4) * 4
        self.total = off
        self.forward_only = bool(forward_only)
        self.theta = _zeros(self.total, device=self.device)
        self.grad = None if forward_only else _zeros(self.total, device=self.device)
        self.m = None if forward_only else _zeros(self.total, device=self.device)
        self.v = None if forward_only else _zeros(self.total, device=self.device)
        self.params: Dict[str, Param] = {}
        for n, o, sz in zip(names, offs, sizes):
            self.params[n] = Param(n, self.spec[n], self.theta[o:o + sz], None if forward_only else self.grad[o:o + sz])
        self.moving: Dict[str, torch.Tensor] = {}
        self.step_count = 0
        self.load_state(state)
        self._bn_acc = None
        self._cache: Dict[str, object] = {}

    # ---- state --------------------------------------------------------------------------------------------------------------
    def load_state(self, state: Dict[str, np.ndarray]):
        for n, shape in self.spec.items():
            arr = torch.from_numpy(np.ascontiguousarray(state[n], dtype=np.float32).reshape(-1)).to(self.device)
            if is_trainable(n):
                self.params[n].w.copy_(arr)
            else:
                self.moving[n] = arr.clone()

    def state_dict(self) -> Dict[str, np.ndarray]:
        out = {}
        host = self.theta.cpu().numpy()
        base = self.theta.data_ptr()
        for n, shape in self.spec.items():
            if is_trainable(n):
                p = self.params[n]
                o = (p.w.data_ptr() - base) // 4
                out[n] = host[o:o + p.w.numel()].reshape(shape).copy()
            else:
                out[n] = self.moving[n].cpu().numpy().reshape(shape).copy()
        return out

    def grads(self) -> Dict[str, np.ndarray]:
        host = self.grad.cpu().numpy()
        base = self.grad.data_ptr()
        return {n: host[(p.g.data_ptr() - base) // 4:(p.g.data_ptr() - base) // 4 + p.g.numel()].reshape(p.shape).copy()
                for n, p in self.params.items()}

    # ---- low-level launches ------------------------------------------------------------------------------------------------
    def _packed(self, name: str, flip: bool = False) -> torch.Tensor:
        """[taps][coutpad][cinpad] image of a Conv3D / Dense kernel (cached for the step); flip: of its data-gradient kernel."""
        key = ("pk", name, flip)
        if key not in self._cache:
            p = self.params[name]
            shape = p.shape
            taps = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            cin, cout = int(shape[-2]), int(shape[-1])
            src = p.w
            if flip:
                src = _empty(taps * cin * cout, device=self.device)
                check(lib().dm3d_flip_transpose(p.w.data_ptr(), taps, cin, cout, src.data_ptr(), _st()), "flip_transpose")
                cin, cout = cout, cin
            out = _empty(lib().dm3d_packed_weight_elems(taps, cin, cout), device=self.device)
            check(lib().dm3d_pack_weights(src.data_ptr(), taps, cin, cout, None, out.data_ptr(), _st()), "pack_weights")
            self._cache[key] = out
        return self._cache[key]

    def _conv_launch(self, x: torch.Tensor, wpk: torch.Tensor, cin: int, cout: int, ksize: int, stride: int = 1, bias=None,
                     vec=None, vec_ld: int = 0, res=None) -> torch.Tensor:
        B, D, H, W_, _ = x.shape
        od, oh, ow = (-(-D // stride), -(-H // stride), -(-W_ // stride))
        out = _empty(B, od, oh, ow, cout, device=self.device)
        d = ConvDesc()
        d.x1, d.c1, d.c2, d.batch = x.data_ptr(), cin, 0, B
        d.in_d, d.in_h, d.in_w, d.ksize, d.stride = D, H, W_, ksize, stride
        d.wpk, d.bias = wpk.data_ptr(), (bias.data_ptr() if bias is not None else None)
        if vec is not None:
            d.vec, d.vec_ld = vec.data_ptr(), vec_ld
        d.res = res.data_ptr() if res is not None else None
        d.out, d.cout, d.precision = out.data_ptr(), cout, _lib.PREC_F32
        check(lib().dm3d_conv3d_ndhwc(C.byref(d), _st()), "conv3d")
        return out

    def _gemm(self, a, lda, b, ldb, m, n, k, out=None, ldo=None, batch=1, stride_a=0, stride_b=0, stride_o=0, alpha=1.0, bias=None,
              act=ACT_NONE, res=None, ldr=0, stride_r=0) -> torch.Tensor:
        if out is None:
            out = _empty(batch * m, n, device=self.device)
        d = GemmDesc()
        d.a, d.lda, d.stride_a = a.data_ptr(), lda, stride_a
        d.b, d.ldb, d.stride_b = b.data_ptr(), ldb, stride_b
        d.out, d.ldo, d.stride_o = out.data_ptr(), (ldo if ldo is not None else n), stride_o
        d.m, d.n, d.k, d.batch, d.alpha = m, n, k, batch, alpha
        d.bias, d.act = (bias.data_ptr() if bias is not None else None), act
        if res is not None:
            d.res, d.ldr, d.stride_r = res.data_ptr(), ldr, stride_r
        d.precision = _lib.PREC_F32
        check(lib().dm3d_gemm_tn(C.byref(d), _st()), "gemm")
        return out

    def _wgrad(self, a: torch.Tensor, g: torch.Tensor, dw: torch.Tensor, cin: int, cout: int, ksize: int, batch: int, d_: int, h: int, w: int,
               per_item: bool = False, stride_a: int = 0, stride_g: int = 0, stride_dw: int = 0):
        d = WgradDesc()
        d.a, d.g, d.dw = a.data_ptr(), g.data_ptr(), dw.data_ptr()
        d.batch, d.in_d, d.in_h, d.in_w, d.cin, d.cout, d.ksize = batch, d_, h, w, cin, cout, ksize
        d.per_item_output, d.stride_a, d.stride_g, d.stride_dw = int(per_item), stride_a, stride_g, stride_dw
        check(lib().dm3d_wgrad(C.byref(d), _st()), "wgrad")

    def _colsum(self, x: torch.Tensor, groups: int, rows: int, c: int, out_ptr: int, ld_out: int):
        check(lib().dm3d_colsum(x.data_ptr(), groups, rows, c, out_ptr, ld_out, _st()), "colsum")

    def _act(self, x: torch.Tensor, act: int) -> torch.Tensor:
        y = torch.empty_like(x)
        c = x.shape[-1]
        check(lib().dm3d_affine_act(x.data_ptr(), y.data_ptr(), x.numel() // c, c, None, None, act, _st()), "affine_act")
        return y

    # ---- layers (forward + recorded backward) -------------------------------------------------------------------------------
    def conv(self, x: Var, name: str, ksize: int, stride: int = 1, vec: Optional[Var] = None, res: Optional[Var] = None) -> Var:
        """Conv3D(padding="same") (+ bias, + the per-sample time-embedding vector, + residual)."""
        pk, pb = self.params[f"{name}.kernel"], self.params[f"{name}.bias"]
        cin, cout = pk.shape[-2], pk.shape[-1]
        B, D, H, W_, _ = x.v.shape
        out = Var(self._conv_launch(x.v, self._packed(pk.name), cin, cout, ksize, stride, bias=pb.w,
                                    vec=vec.v if vec is not None else None, vec_ld=cout, res=res.v if res is not None else None))

        def bwd():
            g = out.g
            if g is None:
                return
            od, oh, ow = g.shape[1:4]
            geff = g
            if stride == 2:
                # TF SAME: zeros in front = total // 2 (k3: 0 on even sizes, 1 on odd); spread dL/dy over the input grid at 2*o + 1 - pad_front
                offs = []
                for n_in, n_out in ((D, od), (H, oh), (W_, ow)):
                    total = max((n_out - 1) * 2 + ksize - n_in, 0)
                    offs.append(1 - total // 2)
                geff = _empty(B, D, H, W_, cout, device=self.device)
                check(lib().dm3d_dilate2(g.data_ptr(), geff.data_ptr(), B, od, oh, ow, D, H, W_, offs[0], offs[1], offs[2], cout, _st()), "dilate2")
            if x.needs_grad:
                x.add_grad(self._conv_launch(geff, self._packed(pk.name, flip=True), cout, cin, ksize, 1))
            self._wgrad(x.v, geff, pk.g, cin, cout, ksize, B, D, H, W_)
            self._colsum(g, 1, g.numel() // cout, cout, pb.g.data_ptr(), cout)
            if vec is not None and vec.needs_grad:
                self._colsum(g, B, od * oh * ow, cout, vec.grad_buffer().data_ptr(), cout)
            if res is not None:
                res.add_grad(g)

        self.tape.append(bwd)
        return out

    def bn_act(self, x1: Var, x2: Optional[Var], name: str, act: int) -> Var:
        """act(BatchNormalization(training=True)(concat[x1, x2])): batch statistics, moving averages updated (momentum 0.99)."""
        B = x1.v.shape[0]
        c1 = x1.v.shape[-1]
        c2 = x2.v.shape[-1] if x2 is not None else 0
        ct = c1 + c2
        rows = x1.v.numel() // c1
        vox = rows // B
        dev = self.device
        if self._bn_acc is None or self._bn_acc.numel() < B * max(ct, 1024) * 2:
            self._bn_acc = torch.zeros(B * max(ct, 1024) * 2, dtype=torch.float64, device=dev)
        acc = self._bn_acc
        check(lib().dm3d_groupnorm_stats(x1.v.data_ptr(), B, vox, c1, acc.data_ptr(), ct, 0, _st()), "bn_stats")
        if x2 is not None:
            check(lib().dm3d_groupnorm_stats(x2.v.data_ptr(), B, vox, c2, acc.data_ptr(), ct, c1, _st()), "bn_stats")
        pg, pb = self.params[f"{name}.gamma"], self.params[f"{name}.beta"]
        st = _empty(4, ct, device=dev)                      # scale, shift, mean, rstd
        ptr = [st[i].data_ptr() for i in range(4)]
        mm, mv = self.moving[f"{name}.mean"], self.moving[f"{name}.var"]
        check(lib().dm3d_batchnorm_finalize(acc.data_ptr(), B, vox, ct, BN_EPS, pg.w.data_ptr(), pb.w.data_ptr(), ptr[0], ptr[1], ptr[2], ptr[3],
                                            mm.data_ptr() if self.update_moving else None, mv.data_ptr() if self.update_moving else None,
                                            BN_MOMENTUM, int(self.bn_moving_unbiased), _st()), "batchnorm_finalize")
        y = _empty(*x1.v.shape[:-1], ct, device=dev)
        x2p = x2.v.data_ptr() if x2 is not None else None
        check(lib().dm3d_affine_act_cat(x1.v.data_ptr(), c1, x2p, c2, rows, ptr[0], ptr[1], act, y.data_ptr(), _st()), "affine_act_cat")
        out = Var(y)

        def bwd():
            if out.g is None:
                return
            ptr = [st[i].data_ptr() for i in range(4)]          # (keeps st — scale, shift, mean, rstd — alive until the backward pass)
            red = torch.zeros(ct * 2, dtype=torch.float64, device=dev)
            dx1 = x1.grad_buffer().data_ptr() if x1.needs_grad else None
            dx2 = x2.grad_buffer().data_ptr() if (x2 is not None and x2.needs_grad) else None
            check(lib().dm3d_bn_act_bwd(out.g.data_ptr(), x1.v.data_ptr(), c1, x2p, c2, rows, ptr[0], ptr[1], ptr[2], ptr[3], act,
                                        red.data_ptr(), dx1, dx2, pg.g.data_ptr(), pb.g.data_ptr(), _st()), "bn_act_bwd")

        self.tape.append(bwd)
        return out

    def dense(self, x: Var, name: str, act: int = ACT_NONE, res: Optional[Var] = None, kernel_name: Optional[str] = None) -> Var:
        """layers.Dense on the last axis / a 1x1 Conv3D: act(x W + b) (+ res, added after the activation)."""
        pk = self.params[kernel_name or f"{name}.kernel"]
        pb = self.params[f"{name}.bias"]
        cin, cout = pk.shape[-2], pk.shape[-1]
        M = x.v.numel() // cin
        cinpad = -(-cin // _lib.CIN_PAD) * _lib.CIN_PAD
        pre = self._gemm(x.v, cin, self._packed(pk.name), cinpad, M, cout, cin, bias=pb.w, act=ACT_RELU if act == ACT_RELU else ACT_NONE)
        y = self._act(pre, ACT_SILU) if act == ACT_SILU else pre       # swish keeps its pre-activation for the backward pass
        if res is not None:
            y2 = torch.empty_like(y)
            check(lib().dm3d_affine_act_cat(y.data_ptr(), cout, None, 0, M, None, None, ACT_NONE, y2.data_ptr(), _st()), "copy")
            check(lib().dm3d_axpy(y2.data_ptr(), res.v.data_ptr(), y2.numel(), 1.0, _st()), "axpy")
            outv = y2
        else:
            outv = y
        out = Var(outv.reshape(*x.v.shape[:-1], cout))

        def bwd():
            g = out.g
            if g is None:
                return
            if res is not None:
                res.add_grad(g)                            # (g may now be owned by res: everything below only reads it)
            dpre = g
            if act != ACT_NONE:
                dpre = torch.empty_like(g)
                check(lib().dm3d_act_bwd(pre.data_ptr(), g.data_ptr(), dpre.data_ptr(), g.numel(), act, _st()), "act_bwd")
            if x.needs_grad:
                dx = self._gemm(dpre, cout, pk.w, cout, M, cin, cout)          # dy . W^T: the Keras kernel [in][out] is K-contiguous in `out`
                x.add_grad(dx.reshape(x.v.shape))
            self._wgrad(x.v, dpre, pk.g, cin, cout, 1, 1, M, 1, 1)
            self._colsum(dpre, 1, M, cout, pb.g.data_ptr(), cout)

        self.tape.append(bwd)
        return out

    def silu(self, x: Var) -> Var:
        out = Var(self._act(x.v, ACT_SILU))

        def bwd():
            if out.g is None or not x.needs_grad:
                return
            dx = torch.empty_like(x.v)
            check(lib().dm3d_act_bwd(x.v.data_ptr(), out.g.data_ptr(), dx.data_ptr(), dx.numel(), ACT_SILU, _st()), "act_bwd")
            x.add_grad(dx)

        self.tape.append(bwd)
        return out

    def layernorm(self, x: Var, name: str) -> Var:
        pg, pb = self.params[f"{name}.gamma"], self.params[f"{name}.beta"]
        c = x.v.shape[-1]
        rows = x.v.numel() // c
        y = torch.empty_like(x.v)
        check(lib().dm3d_layernorm3(x.v.data_ptr(), rows, c, LN_EPS, pg.w.data_ptr(), pb.w.data_ptr(), y.data_ptr(), None, None, None,
                                    None, None, None, _st()), "layernorm")
        out = Var(y)

        def bwd():
            if out.g is None:
                return
            check(lib().dm3d_layernorm_bwd(x.v.data_ptr(), rows, c, LN_EPS, pg.w.data_ptr(), out.g.data_ptr(), x.grad_buffer().data_ptr(),
                                           pg.g.data_ptr(), pb.g.data_ptr(), _st()), "layernorm_bwd")

        self.tape.append(bwd)
        return out

    def attention(self, q: Var, k: Var, v: Var, B: int, L: int, Lk: int, u: int) -> Var:
        """softmax(q k^T * u^-0.5) v per sample (conditional_dm3d.py:171-180; dm3d.py:51-61).  q [B*L, u]; k, v [B*Lk, u]."""
        dev, scale = self.device, float(u) ** -0.5
        if L % 4 or Lk % 4:
            raise ValueError(f"attention over {L} x {Lk} tokens: the contractions run on the MFMA GEMM, which needs D*H*W % 4 == 0 at attention levels")
        P = _empty(B * L, Lk, device=dev)
        self._gemm(q.v, u, k.v, u, L, Lk, u, out=P, ldo=Lk, batch=B, stride_a=L * u, stride_b=Lk * u, stride_o=L * Lk, alpha=scale)
        check(lib().dm3d_softmax_rows(P.data_ptr(), B * L, Lk, Lk, _st()), "softmax")
        vt = _empty(B * u, Lk, device=dev)
        check(lib().dm3d_transpose(v.v.data_ptr(), Lk, u, u, Lk * u, vt.data_ptr(), Lk, u * Lk, B, _st()), "transpose")
        o = _empty(B * L, u, device=dev)
        self._gemm(P, Lk, vt, Lk, L, u, Lk, out=o, ldo=u, batch=B, stride_a=L * Lk, stride_b=u * Lk, stride_o=L * u)
        out = Var(o)

        def bwd():
            g = out.g
            if g is None:
                return
            dP = _empty(B * L, Lk, device=dev)
            self._gemm(g, u, v.v, u, L, Lk, u, out=dP, ldo=Lk, batch=B, stride_a=L * u, stride_b=Lk * u, stride_o=L * Lk)     # dO . V^T
            if v.needs_grad:
                dv = _zeros(B * Lk, u, device=dev)
                self._wgrad(P, g, dv, Lk, u, 1, B, L, 1, 1, per_item=True, stride_a=L * Lk, stride_g=L * u, stride_dw=Lk * u)   # P^T . dO
                v.add_grad(dv)
            check(lib().dm3d_softmax_bwd(P.data_ptr(), dP.data_ptr(), B * L, Lk, Lk, scale, _st()), "softmax_bwd")              # dP -> dS
            if q.needs_grad:
                kt = _empty(B * u, Lk, device=dev)
                check(lib().dm3d_transpose(k.v.data_ptr(), Lk, u, u, Lk * u, kt.data_ptr(), Lk, u * Lk, B, _st()), "transpose")
                dq = _empty(B * L, u, device=dev)
                self._gemm(dP, Lk, kt, Lk, L, u, Lk, out=dq, ldo=u, batch=B, stride_a=L * Lk, stride_b=u * Lk, stride_o=L * u)  # dS . K
                q.add_grad(dq)
            if k.needs_grad:
                dk = _zeros(B * Lk, u, device=dev)
                self._wgrad(dP, q.v, dk, Lk, u, 1, B, L, 1, 1, per_item=True, stride_a=L * Lk, stride_g=L * u, stride_dw=Lk * u)  # dS^T . Q
                k.add_grad(dk)

        self.tape.append(bwd)
        return out

    def add(self, a: Var, b: Var) -> Var:
        y = torch.empty_like(a.v)
        c = a.v.shape[-1]
        check(lib().dm3d_affine_act_cat(a.v.data_ptr(), c, None, 0, a.v.numel() // c, None, None, ACT_NONE, y.data_ptr(), _st()), "copy")
        check(lib().dm3d_axpy(y.data_ptr(), b.v.data_ptr(), y.numel(), 1.0, _st()), "axpy")
        out = Var(y)

        def bwd():
            if out.g is None:
                return
            # two consumers of one tensor: the first takes it, the second gets a copy (add_grad may accumulate into what it was given)
            a.add_grad(out.g)
            if b.needs_grad:
                cp = torch.empty_like(out.g)
                check(lib().dm3d_affine_act_cat(out.g.data_ptr(), c, None, 0, out.g.numel() // c, None, None, ACT_NONE, cp.data_ptr(), _st()), "copy")
                b.add_grad(cp)

        self.tape.append(bwd)
        return out

    def upsample2(self, x: Var) -> Var:
        B, D, H, W_, c = x.v.shape
        y = _empty(B, 2 * D, 2 * H, 2 * W_, c, device=self.device)
        check(lib().dm3d_upsample2(x.v.data_ptr(), y.data_ptr(), B, D, H, W_, c, _st()), "upsample2")
        out = Var(y)

        def bwd():
            if out.g is None or not x.needs_grad:
                return
            check(lib().dm3d_sumpool2_add(out.g.data_ptr(), x.grad_buffer().data_ptr(), B, D, H, W_, c, _st()), "sumpool2_add")

        self.tape.append(bwd)
        return out

    def embedding(self, name: str, ids: torch.Tensor) -> Var:
        p = self.params[name]
        rows, c = p.shape
        n = ids.numel()
        out_t = _empty(n, c, device=self.device)
        check(lib().dm3d_gather_rows(p.w.data_ptr(), rows, ids.data_ptr(), out_t.data_ptr(), n, c, _st()), "gather_rows")
        out = Var(out_t)

        def bwd():
            if out.g is None:
                return
            check(lib().dm3d_scatter_add_rows(out.g.data_ptr(), ids.data_ptr(), n, c, p.g.data_ptr(), rows, _st()), "scatter_add_rows")

        self.tape.append(bwd)
        return out

    # ---- blocks -----------------------------------------------------------------------------------------------------------------
    def _res_block(self, n: str, x1: Var, x2: Optional[Var], s_temb: Var) -> Var:
        """ResidualBlock (conditional_dm3d.py:238-271) with training-mode BatchNormalization."""
        width = self.params[f"{n}.conv1.kernel"].shape[-1]
        cin = x1.v.shape[-1] + (x2.v.shape[-1] if x2 is not None else 0)
        if f"{n}.skip.kernel" in self.params:
            xin = x1 if x2 is None else self._concat(x1, x2)
            residual = self.dense(xin, f"{n}.skip")                                            # Conv3D(width, kernel_size=1)
        else:
            residual = x1
        te = self.dense(s_temb, f"{n}.temb")                                                   # Dense(width)(swish(temb)) [B, width]
        a1 = self.bn_act(x1, x2, f"{n}.norm1", ACT_SILU)
        h = self.conv(a1, f"{n}.conv1", 3, vec=te)
        a2 = self.bn_act(h, None, f"{n}.norm2", ACT_SILU)
        return self.conv(a2, f"{n}.conv2", 3, res=residual)

    def _concat(self, x1: Var, x2: Var) -> Var:
        c1, c2 = x1.v.shape[-1], x2.v.shape[-1]
        rows = x1.v.numel() // c1
        y = _empty(*x1.v.shape[:-1], c1 + c2, device=self.device)
        check(lib().dm3d_affine_act_cat(x1.v.data_ptr(), c1, x2.v.data_ptr(), c2, rows, None, None, ACT_NONE, y.data_ptr(), _st()), "concat")
        out = Var(y)

        def bwd():
            if out.g is None:
                return
            g = out.g
            for xv, off, cc in ((x1, 0, c1), (x2, c1, c2)):
                if not xv.needs_grad:
                    continue
                check(lib().dm3d_copy_cols(g.data_ptr(), c1 + c2, off, xv.grad_buffer().data_ptr(), cc, 0, rows, cc, 1, _st()), "copy_cols")

        self.tape.append(bwd)
        return out

    def _cross_block(self, n: str, x: Var, cemb: Var, B: int, L: int) -> Var:
        """CrossAttentionBlock (conditional_dm3d.py:186-195) + its ContextMLP (:310-318)."""
        u = x.v.shape[-1]
        M = B * L
        feat_pre = self.dense(cemb, f"{n}.ctx_mlp")                                           # [B, L*u]
        feat = self.silu(feat_pre)
        feat2 = Var(feat.v.reshape(M, u))
        self._alias(feat2, feat)
        xn = self.bn_act(x, None, f"{n}.norm", ACT_NONE)
        xn2 = Var(xn.v.reshape(M, u))
        self._alias(xn2, xn)
        y = self.dense(xn2, f"{n}.proj_in", act=ACT_RELU)
        n1, n2, n3 = self.layernorm(y, f"{n}.ln1"), self.layernorm(y, f"{n}.ln2"), self.layernorm(y, f"{n}.ln3")
        q, k, v = self.dense(n1, f"{n}.query"), self.dense(n1, f"{n}.key"), self.dense(n1, f"{n}.value")
        a = self.add(self.attention(q, k, v, B, L, L, u), y)
        q2, kc, vc = self.dense(n2, f"{n}.query"), self.dense(feat2, f"{n}.key"), self.dense(feat2, f"{n}.value")
        a = self.add(self.attention(q2, kc, vc, B, L, L, u), a)
        hid = self.dense(n3, f"{n}.mlp.0", act=ACT_RELU)
        a = self.dense(hid, f"{n}.mlp.1", res=a)
        x2d = Var(x.v.reshape(M, u), needs_grad=x.needs_grad)
        self._alias(x2d, x)
        out = self.dense(a, f"{n}.proj_out", act=ACT_RELU, res=x2d)
        o5 = Var(out.v.reshape(x.v.shape))
        self._alias(o5, out)
        return o5

    def _self_block(self, n: str, x: Var, B: int, L: int) -> Var:
        """AttentionBlock (dm3d.py:39-63): BN(x) + proj(softmax(q k^T u^-0.5) v)."""
        u = x.v.shape[-1]
        M = B * L
        xn = self.bn_act(x, None, f"{n}.norm", ACT_NONE)
        xn2 = Var(xn.v.reshape(M, u))
        self._alias(xn2, xn)
        q, k, v = self.dense(xn2, f"{n}.query"), self.dense(xn2, f"{n}.key"), self.dense(xn2, f"{n}.value")
        o = self.attention(q, k, v, B, L, L, u)
        out = self.dense(o, f"{n}.proj", res=xn2)
        o5 = Var(out.v.reshape(x.v.shape))
        self._alias(o5, out)
        return o5

    def _alias(self, view: Var, base: Var):
        """``view`` is a reshape of ``base``: its gradient is handed over unchanged (reshaped)."""
        def bwd():
            if view.g is not None:
                base.add_grad(view.g.reshape(base.v.shape))
        self.tape.append(bwd)

    # ---- the network ------------------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, t_host: np.ndarray, ctx_ids: Optional[torch.Tensor], update_moving: bool = True) -> Var:
        """build_model's graph (conditional_dm3d.py:348-415 / dm3d.py:318-376) with training=True; records the tape."""
        cfg = self.cfg
        self.tape: List = []
        self._cache = {}
        self.update_moving = update_moving
        B = x.shape[0]
        xin = Var(x, needs_grad=False)
        h = self.conv(xin, "conv_in", 3)
        emb = Var(torch.from_numpy(time_embedding_table(np.asarray(t_host), cfg.temb_dim)).to(self.device), needs_grad=False)
        temb = self.dense(self.dense(emb, "time_mlp.0", act=ACT_SILU), "time_mlp.1")
        s_temb = self.silu(temb)                                                               # every ResidualBlock starts with swish(temb)
        cemb = self.embedding("ctx_embed.table", ctx_ids) if cfg.conditional else None
        skips = [h]
        for blk in self.blocks:
            if blk.kind == "push":
                skips.append(h)
            elif blk.kind == "res":
                x2 = skips.pop() if blk.cskip else None
                h = self._res_block(blk.name, h, x2, s_temb)
            elif blk.kind == "attn":
                L = blk.edge ** 3
                h = self._cross_block(blk.name, h, cemb, B, L) if cfg.conditional else self._self_block(blk.name, h, B, L)
            elif blk.kind == "down":
                h = self.conv(h, blk.name, 3, stride=2)
            elif blk.kind == "up":
                h = self.conv(self.upsample2(h), blk.name, 3)
        a = self.bn_act(h, None, "out.norm", ACT_SILU)
        return self.conv(a, "out.conv", 3)

    def backward(self):
        for fn in reversed(self.tape):
            fn()
        self.tape = []

    def zero_grad(self):
        check(lib().dm3d_fill(self.grad.data_ptr(), self.grad.numel(), 0.0, _st()), "fill")

    def loss_and_grad(self, latents: torch.Tensor, t: torch.Tensor, noise: torch.Tensor, ctx_ids, betas_dev, timesteps: int, global_bs: int,
                      lc: int, update_moving: bool = True):
        """q_sample -> training forward -> loss -> backward.  Returns (loss [1] float64 device tensor, pred)."""
        B = latents.shape[0]
        dev = self.device
        t_dev = t.to(dev, torch.int32)
        noisy = torch.empty_like(latents)
        sqab, sq1ab = betas_dev
        check(lib().dm3d_q_sample(latents.data_ptr(), noise.data_ptr(), t_dev.data_ptr(), sqab.data_ptr(), sq1ab.data_ptr(), timesteps,
                                  noisy.data_ptr(), B, latents[0].numel(), _st()), "q_sample")
        ids = None
        if self.cfg.conditional:
            ids = torch.as_tensor(np.asarray(ctx_ids, dtype=np.int32).reshape(-1)).to(dev)
            if ids.numel() == 1 and B > 1:
                ids = ids.repeat(B)
        self.zero_grad()
        pred = self.forward(noisy, t.cpu().numpy().astype(np.int64), ids, update_moving)
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        pred.g = torch.empty_like(pred.v)
        inv = 1.0 / (float(latents.shape[-1]) * float(global_bs) * float(lc) ** 4)
        check(lib().dm3d_mse_loss_grad(pred.v.data_ptr(), noise.data_ptr(), pred.v.numel(), inv, loss.data_ptr(), pred.g.data_ptr(), _st()),
              "mse_loss_grad")
        self.backward()
        self._cache = {}
        return loss, pred.v

    def allreduce_grads(self, loss: Optional[torch.Tensor] = None):
        """Data-parallel training (what MirroredStrategy(cross_device_ops=ReductionToOneDevice()) does for the reference,
        main_conditional_dm.py:87): SUM the gradients over the ranks — the loss is already divided by the GLOBAL batch
        (conditional_dm3d.py:496-499).  The flat gradient buffer is the bucket: one all-reduce (RCCL over xGMI under backend "nccl")
        per step.  BatchNormalization normalises with per-replica batch statistics as in Keras, and its moving mean / variance —
        which Keras aggregates by MEAN across replicas (VariableAggregation.MEAN) — are averaged here as one small second bucket, so every
        rank (and the checkpoint rank 0 writes) holds the same statistics.  ``loss`` (optional, [1] device tensor): summed over the
        ranks in place (each rank's loss is its shard's share of the global-batch loss).  No-op without a process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        world = dist.get_world_size()
        cpu_group = dist.get_backend() == "gloo"            # rehearsal / CPU tests: gloo reduces host tensors
        def reduce_(t: torch.Tensor, scale: float = 1.0):
            if cpu_group and t.is_cuda:
                h = t.detach().cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h if scale == 1.0 else h * scale)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                if scale != 1.0:
                    t.mul_(scale)
        reduce_(self.grad)
        names = sorted(self.moving)
        flat = torch.cat([self.moving[n].reshape(-1) for n in names])
        reduce_(flat, 1.0 / world)
        off = 0
        for n in names:
            k = self.moving[n].numel()
            self.moving[n].copy_(flat[off:off + k])
            off += k
        if loss is not None:
            reduce_(loss)

    # ---- optimizer state (keras.optimizers.Adam slots: the reference's ModelCheckpoint(save_weights_only=True) checkpoints carry them) ----
    def optimizer_state(self) -> Dict[str, np.ndarray]:
        """{"optimizer/iter": step count, "optimizer/m/<name>", "optimizer/v/<name>"}: what a resumed run needs beside the weights."""
        out = {"optimizer/iter": np.asarray(self.step_count, dtype=np.int64)}
        hm, hv, base = self.m.cpu().numpy(), self.v.cpu().numpy(), self.theta.data_ptr()
        for n, p in self.params.items():
            o = (p.w.data_ptr() - base) // 4
            out[f"optimizer/m/{n}"] = hm[o:o + p.w.numel()].reshape(p.shape).copy()
            out[f"optimizer/v/{n}"] = hv[o:o + p.w.numel()].reshape(p.shape).copy()
        return out

    def load_optimizer_state(self, st: Dict[str, np.ndarray]):
        self.step_count = int(np.asarray(st["optimizer/iter"]).reshape(-1)[0])
        base = self.theta.data_ptr()
        for n, p in self.params.items():
            o = (p.w.data_ptr() - base) // 4
            for slot, buf in (("m", self.m), ("v", self.v)):
                arr = np.ascontiguousarray(st[f"optimizer/{slot}/{n}"], dtype=np.float32).reshape(-1)
                if arr.size != p.w.numel():
                    raise ValueError(f"optimizer slot {slot} of {n}: {arr.size} values for a parameter of {p.w.numel()}")
                buf[o:o + arr.size].copy_(torch.from_numpy(arr))

    def adam_step(self):
        """keras.optimizers.Adam.apply_gradients over the flat buffers (one launch)."""
        self.step_count += 1
        tt = self.step_count
        lr_t = self.lr * (1.0 - ADAM_BETA2 ** tt) ** 0.5 / (1.0 - ADAM_BETA1 ** tt)
        check(lib().dm3d_adam(self.theta.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.total, lr_t,
                              ADAM_BETA1, ADAM_BETA2, ADAM_EPS, _st()), "adam")

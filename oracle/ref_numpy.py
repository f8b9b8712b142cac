"""ORACLE (second, independent restatement) — NumPy float64, no torch, explicit loops over kernel taps.

TEST INFRASTRUCTURE ONLY (see ``oracle/ref_torch.py`` for the rules).  PARITY UNPINNED: the reference has no golden
vectors; this file exists so that two restatements written separately from the reference text must agree before any
fixture is frozen.  It re-derives every layer from its definition instead of calling library
convolution / normalisation / softmax routines.

Citations: ``C:`` = networks/conditional_dm3d.py, ``U:`` = networks/dm3d.py of the reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

BN_EPS = 1e-3
LN_EPS = 1e-3


def to_f64(W) -> Dict[str, np.ndarray]:
    return {k: np.asarray(v, dtype=np.float64) for k, v in W.items()}


def conv3d_same(x, kernel, bias, stride=1, upsample=False):
    """Conv3D(padding='same') as a sum over taps of shifted-slice contractions (C:257-259, 276-282, 288-294).
    TF SAME: out = ceil(in/stride); total pad = max((out-1)*stride + k - in, 0); before = total//2."""
    if upsample:                                       # UpSampling3D(2): y[2i+a] = x[i]
        x = np.repeat(np.repeat(np.repeat(x, 2, axis=1), 2, axis=2), 2, axis=3)
    k = kernel.shape[0]
    B, D, H, Wd, _ = x.shape
    outs, befores = [], []
    for n in (D, H, Wd):
        o = -(-n // stride)
        tot = max((o - 1) * stride + k - n, 0)
        outs.append(o)
        befores.append(tot // 2)
    xp = np.zeros((B, D + k, H + k, Wd + k, x.shape[-1]))
    xp[:, befores[0]:befores[0] + D, befores[1]:befores[1] + H, befores[2]:befores[2] + Wd] = x
    y = np.zeros((B, outs[0], outs[1], outs[2], kernel.shape[-1]))
    for a in range(k):
        for b in range(k):
            for c in range(k):
                sl = xp[:, a:a + (outs[0] - 1) * stride + 1:stride,
                        b:b + (outs[1] - 1) * stride + 1:stride,
                        c:c + (outs[2] - 1) * stride + 1:stride]
                y += sl @ kernel[a, b, c]
    return y + bias


NORM = {"mode": "batch", "groups": 8}


def bn(x, W, n):
    if NORM["mode"] == "group" and x.ndim == 5:         # GroupNormalization(groups, eps 1e-3): moments over (D,H,W,C/groups)
        B, D, H, Wd, C = x.shape
        g = NORM["groups"]
        xg = x.reshape(B, D * H * Wd, g, C // g)
        mu = xg.mean(axis=(1, 3), keepdims=True)
        var = ((xg - mu) ** 2).mean(axis=(1, 3), keepdims=True)
        return ((xg - mu) / np.sqrt(var + BN_EPS)).reshape(x.shape) * W[n + ".gamma"] + W[n + ".beta"]
    return W[n + ".gamma"] * (x - W[n + ".mean"]) / np.sqrt(W[n + ".var"] + BN_EPS) + W[n + ".beta"]


def ln(x, W, n):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + LN_EPS) * W[n + ".gamma"] + W[n + ".beta"]


def swish(x):
    return x / (1.0 + np.exp(-x))


def dense(x, W, n):
    return x @ W[n + ".kernel"] + W[n + ".bias"]


def softmax_last(s):
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    return e / e.sum(-1, keepdims=True)


def time_embedding(t, dim):
    """C:198-212 evaluated in float64 (the fp32 op order lives in ref_torch)."""
    half = dim // 2
    f = np.exp(np.arange(half, dtype=np.float64) * -(math.log(10000) / (half - 1)))
    a = np.asarray(t, dtype=np.float64)[:, None] * f[None, :]
    return np.concatenate([np.sin(a), np.cos(a)], -1)


def residual_block(W, n, x, temb):
    """C:238-271."""
    width = W[n + ".conv1.kernel"].shape[-1]
    res = x if x.shape[-1] == width else conv3d_same(x, W[n + ".skip.kernel"], W[n + ".skip.bias"])
    te = dense(swish(temb), W, n + ".temb")[:, None, None, None, :]
    h = conv3d_same(swish(bn(x, W, n + ".norm1")), W[n + ".conv1.kernel"], W[n + ".conv1.bias"]) + te
    h = conv3d_same(swish(bn(h, W, n + ".norm2")), W[n + ".conv2.kernel"], W[n + ".conv2.bias"])
    return h + res


def _attend(q, k, v, u):
    s = softmax_last(np.einsum("blc,bmc->blm", q, k) * (float(u) ** -0.5))
    return np.einsum("blm,bmc->blc", s, v)


def cross_block(W, n, x, ctx):
    """C:186-195."""
    B, h, w, d, u = x.shape
    y = np.maximum(conv3d_same(bn(x, W, n + ".norm"), W[n + ".proj_in.kernel"], W[n + ".proj_in.bias"]), 0)
    flat = lambda z: z.reshape(z.shape[0], -1, u)

    def att(inp, c=None):
        src = inp if c is None else c
        q, k, v = flat(dense(inp, W, n + ".query")), flat(dense(src, W, n + ".key")), flat(dense(src, W, n + ".value"))
        if k.shape[0] != B:
            k, v = np.repeat(k, B, 0), np.repeat(v, B, 0)
        return _attend(q, k, v, u).reshape(B, h, w, d, u)

    a = att(ln(y, W, n + ".ln1")) + y
    a = att(ln(y, W, n + ".ln2"), ctx) + a
    a = dense(np.maximum(dense(ln(y, W, n + ".ln3"), W, n + ".mlp.0"), 0), W, n + ".mlp.1") + a
    return np.maximum(conv3d_same(a, W[n + ".proj_out.kernel"], W[n + ".proj_out.bias"]), 0) + x


def self_block(W, n, x):
    """U:39-63."""
    B, h, w, d, u = x.shape
    xn = bn(x, W, n + ".norm")
    f = lambda m: dense(xn, W, n + "." + m).reshape(B, -1, u)
    o = _attend(f("query"), f("key"), f("value"), u).reshape(B, h, w, d, u)
    return xn + dense(o, W, n + ".proj")


def unet_forward(W, cfg, x, t, context=None, taps: Optional[dict] = None):
    """C:348-415 / U:318-376; ``cfg`` is an oracle.ref_torch.UNetConfig (plain attribute bag)."""
    W = to_f64(W)
    x = np.asarray(x, np.float64)
    widths = list(cfg.widths)
    NORM["mode"], NORM["groups"] = getattr(cfg, "norm", "batch"), getattr(cfg, "norm_groups", 8)
    keep = (lambda k, v: taps.__setitem__(k, v)) if taps is not None else (lambda k, v: None)
    h = conv3d_same(x, W["conv_in.kernel"], W["conv_in.bias"])
    temb = dense(swish(dense(time_embedding(t, cfg.temb_dim), W, "time_mlp.0")), W, "time_mlp.1")
    cemb = None
    if cfg.conditional:
        cemb = W["ctx_embed.table"][np.asarray(context).reshape(len(context), -1)[:, 0]]

    def attn(n, v):
        if cfg.conditional:
            ctx = swish(dense(cemb, W, n + ".ctx_mlp")).reshape((-1,) + v.shape[1:])
            return cross_block(W, n, v, ctx)
        return self_block(W, n, v)

    skips = [h]
    for i, w in enumerate(widths):
        for j in range(cfg.num_res_blocks):
            h = residual_block(W, f"down{i}.res{j}", h, temb)
            keep(f"down{i}.res{j}", h)
            if cfg.has_attention[i]:
                h = attn(f"down{i}.attn{j}", h)
                keep(f"down{i}.attn{j}", h)
            skips.append(h)
        if w != widths[-1]:
            h = conv3d_same(h, W[f"down{i}.ds.kernel"], W[f"down{i}.ds.bias"], stride=2)
            keep(f"down{i}.ds", h)
            skips.append(h)
    h = residual_block(W, "mid.res0", h, temb)
    h = attn("mid.attn", h)
    keep("mid.attn", h)
    h = residual_block(W, "mid.res1", h, temb)
    for i in reversed(range(len(widths))):
        for j in range(cfg.num_res_blocks + 1):
            h = residual_block(W, f"up{i}.res{j}", np.concatenate([h, skips.pop()], -1), temb)
            if cfg.has_attention[i]:
                h = attn(f"up{i}.attn{j}", h)
        if i != 0:
            h = conv3d_same(h, W[f"up{i}.us.kernel"], W[f"up{i}.us.bias"], upsample=True)
            keep(f"up{i}.us", h)
    return conv3d_same(swish(bn(h, W, "out.norm")), W["out.conv.kernel"], W["out.conv.bias"])


def betas(T):
    """C:215-235 in float64 (before the fp32 cast)."""
    beta = np.linspace(0.0001, 0.02, T)
    alpha = 1 - beta
    ab = np.cumprod(alpha)
    abp = np.append(1.0, ab[:-1])
    return dict(beta=beta, alpha=alpha, sqrt_alpha=np.sqrt(alpha), alpha_bar=ab, alpha_bar_prev=abp,
                sqrt_alpha_bar=np.sqrt(ab), sqrt_alpha_bar_prev=np.sqrt(abp), sqrt_one_minus_alpha_bar=np.sqrt(1 - ab))


def ddpm_step(tab, x, eps, i, z):
    """C:539-546, 572-573 for one scalar step index i."""
    x0 = (x - tab["sqrt_one_minus_alpha_bar"][i] * eps) / tab["sqrt_alpha_bar"][i]
    c1 = tab["beta"][i] * tab["sqrt_alpha_bar_prev"][i] / (1 - tab["alpha_bar"][i])
    c2 = (1 - tab["alpha_bar_prev"][i]) * tab["sqrt_alpha"][i] / (1 - tab["alpha_bar"][i])
    var = (1 - tab["alpha_bar_prev"][i]) * tab["beta"][i] / (1 - tab["alpha_bar"][i])
    return np.clip(c1 * x0 + c2 * x, -1, 1) + math.sqrt(max(var, 1e-20)) * z


# ---- next-1 pieces, re-derived from their definitions (reference networks/vqvae3d_monai.py) ---------------------------------
def conv3d_transpose_k4s2(x, kernel, bias):
    """Conv3DTranspose(k=4, strides=2, padding='same') as the literal transpose of the k4/s2 'same' convolution
    y[i] = sum_k x[2i + k - 1] w[k]: every input voxel scatters into out[2i + k - 1].  kernel [kd,kh,kw,Cout,Cin]."""
    B, D, H, Wd, _ = x.shape
    out = np.zeros((B, 2 * D + 2, 2 * H + 2, 2 * Wd + 2, kernel.shape[3]))       # index + 1 so that k - 1 = -1 fits
    for a in range(4):
        for b in range(4):
            for c in range(4):
                out[:, a:a + 2 * D:2, b:b + 2 * H:2, c:c + 2 * Wd:2] += x @ kernel[a, b, c].T
    return out[:, 1:2 * D + 1, 1:2 * H + 1, 1:2 * Wd + 1] + bias


def prelu(x, alpha):
    return np.where(x > 0, x, alpha * x)


def vq_residual_unit(W, n, x):
    h = np.maximum(conv3d_same(x, W[n + ".conv1.kernel"], W[n + ".conv1.bias"]), 0)
    h = prelu(bn(conv3d_same(h, W[n + ".conv2.kernel"], W[n + ".conv2.bias"]), W, n + ".bn"), W[n + ".prelu.alpha"])
    return np.maximum(x + h, 0)


def vq_code_indices(E, z_flat):
    d = (z_flat ** 2).sum(1, keepdims=True) + (E ** 2).sum(0) - 2 * (z_flat @ E)
    return d.argmin(1)

"""ORACLE — CPU restatement (PyTorch, fp32 or fp64) of the reference's 3D latent-diffusion denoising path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and there only as the
checker / the reported CPU baseline.  The product path (``3d-condtional-stable-diffusion_amd/``) never imports it.

PARITY UNPINNED.  The reference (aayush9400/3D-Condtional-Stable-Diffusion) ships no tests, golden vectors,
checkpoints or saved outputs, and its arithmetic lives in TensorFlow/Keras, which is not installed here
(``ModuleNotFoundError`` on import — an ordinary error, not a denial).  This file is therefore written from the
reference *text* plus Keras' documented layer semantics, and is cross-checked only against a second, independent
NumPy/fp64 restatement (``oracle/ref_numpy.py``) and the analytic known answers listed in SURVEY.md §8(c).

Every function cites the reference lines it follows (paths relative to the reference root):
``networks/conditional_dm3d.py`` (abbreviated ``C:``) and ``networks/dm3d.py`` (``U:``).

Layout conventions (Keras): activations NDHWC; Conv3D kernels ``[kd,kh,kw,Cin,Cout]``; Dense kernels
``[in,out]``; every layer has a bias.  BatchNormalization / LayerNormalization epsilon = 1e-3 (Keras default).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # keras.layers.BatchNormalization default epsilon
LN_EPS = 1e-3  # keras.layers.LayerNormalization default epsilon


# --------------------------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    """Arguments of ``build_model`` (C:324-335 / U:294-305) plus the module constant ``first_conv_channels``
    (C:321 = 32, U:291 = 64)."""

    img_size: int
    img_channels: int
    widths: Sequence[int] = (64, 128, 256)
    has_attention: Sequence[bool] = (False, False, True, True)
    num_res_blocks: int = 2
    conditional: bool = True          # True: conditional_dm3d.py (CrossAttentionBlock); False: dm3d.py (AttentionBlock)
    first_conv_channels: Optional[int] = None
    context_dim: int = 1
    norm: str = "batch"               # "batch": what the reference runs; "group": the GroupNormalization(groups=norm_groups)
    norm_groups: int = 8              #          lines it keeps commented out (C:77, 254, 261, 409), Keras epsilon 1e-3

    def __post_init__(self):
        if self.first_conv_channels is None:
            self.first_conv_channels = 32 if self.conditional else 64

    @property
    def temb_dim(self) -> int:
        return self.first_conv_channels * 4


_NORM = {"mode": "batch", "groups": 8}     # set by unet_forward for the duration of a call (test-only module state)


# --------------------------------------------------------------------------------------------------------------
# a1 kernel_init, a4 Betas, a2 TimeEmbedding
# --------------------------------------------------------------------------------------------------------------
def kernel_init_limit(scale: float, fan_in: int, fan_out: int) -> float:
    """C:17-21.  VarianceScaling(max(scale,1e-10), 'fan_avg', 'uniform') draws U(-l, l) with
    l = sqrt(3*scale/fan_avg)."""
    scale = max(scale, 1e-10)
    return math.sqrt(3.0 * scale / ((fan_in + fan_out) / 2.0))


class Betas:
    """C:215-235 (U:194-214): fp64 numpy tables stored as fp32 constants."""

    NAMES = ("beta", "alpha", "sqrt_alpha", "alpha_bar", "alpha_bar_prev", "sqrt_alpha_bar",
             "sqrt_alpha_bar_prev", "sqrt_one_minus_alpha_bar")

    def __init__(self, timesteps: int):
        beta = np.linspace(0.0001, 0.02, timesteps)
        alpha = 1 - beta
        sqrt_alpha = np.sqrt(alpha)
        alpha_bar = np.cumprod(alpha, 0)
        alpha_bar_prev = np.append(1.0, alpha_bar[:-1])
        sqrt_alpha_bar = np.sqrt(alpha_bar)
        sqrt_alpha_bar_prev = np.sqrt(alpha_bar_prev)
        sqrt_one_minus_alpha_bar = np.sqrt(1 - alpha_bar)
        loc = locals()
        for n in self.NAMES:
            setattr(self, n, torch.from_numpy(loc[n].astype(np.float32)))


def time_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """C:198-212.  fp32 throughout: f = exp(arange(half) * -(ln(1e4)/(half-1))); [sin(t f), cos(t f)]."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    arg = t.to(torch.float32)[:, None] * freqs[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


# --------------------------------------------------------------------------------------------------------------
# parameter inventory — name -> shape, in graph-construction order
# --------------------------------------------------------------------------------------------------------------
def _bn(spec, name, c):
    for s in ("gamma", "beta", "mean", "var"):
        spec[f"{name}.{s}"] = (c,)


def _resblock_spec(spec, name, cin, width, temb_dim):
    if cin != width:                                  # C:243-248
        spec[f"{name}.skip.kernel"] = (1, 1, 1, cin, width)
        spec[f"{name}.skip.bias"] = (width,)
    spec[f"{name}.temb.kernel"] = (temb_dim, width)   # C:251
    spec[f"{name}.temb.bias"] = (width,)
    _bn(spec, f"{name}.norm1", cin)                   # C:255
    spec[f"{name}.conv1.kernel"] = (3, 3, 3, cin, width)
    spec[f"{name}.conv1.bias"] = (width,)
    _bn(spec, f"{name}.norm2", width)                 # C:262
    spec[f"{name}.conv2.kernel"] = (3, 3, 3, width, width)
    spec[f"{name}.conv2.bias"] = (width,)


def _attn_spec(spec, name, u, conditional, temb_dim, tokens):
    _bn(spec, f"{name}.norm", u)
    if conditional:                                   # C:120-138
        for ln in ("ln1", "ln2", "ln3"):
            spec[f"{name}.{ln}.gamma"] = (u,)
            spec[f"{name}.{ln}.beta"] = (u,)
        for pj in ("proj_in", "proj_out"):
            spec[f"{name}.{pj}.kernel"] = (1, 1, 1, u, u)
            spec[f"{name}.{pj}.bias"] = (u,)
        for d in ("query", "key", "value"):
            spec[f"{name}.{d}.kernel"] = (u, u)
            spec[f"{name}.{d}.bias"] = (u,)
        spec[f"{name}.mlp.0.kernel"] = (u, 4 * u)
        spec[f"{name}.mlp.0.bias"] = (4 * u,)
        spec[f"{name}.mlp.1.kernel"] = (4 * u, u)
        spec[f"{name}.mlp.1.bias"] = (u,)
        spec[f"{name}.ctx_mlp.kernel"] = (temb_dim, tokens * u)   # ContextMLP C:310-318
        spec[f"{name}.ctx_mlp.bias"] = (tokens * u,)
    else:                                             # U:26-37 (self.depth is never called => no weights)
        for d in ("query", "key", "value", "proj"):
            spec[f"{name}.{d}.kernel"] = (u, u)
            spec[f"{name}.{d}.bias"] = (u,)


def param_spec(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """Walks build_model (C:348-415 / U:318-376) and lists every weight with its Keras shape."""
    spec: Dict[str, Tuple[int, ...]] = {}
    f0, td, C, S = cfg.first_conv_channels, cfg.temb_dim, cfg.img_channels, cfg.img_size
    widths = list(cfg.widths)
    spec["conv_in.kernel"] = (3, 3, 3, C, f0)
    spec["conv_in.bias"] = (f0,)
    spec["time_mlp.0.kernel"] = (td, td)
    spec["time_mlp.0.bias"] = (td,)
    spec["time_mlp.1.kernel"] = (td, td)
    spec["time_mlp.1.bias"] = (td,)
    if cfg.conditional:
        spec["ctx_embed.table"] = (cfg.context_dim + 1, td)      # C:358
    ch, edge = f0, S
    skips = [ch]
    for i, w in enumerate(widths):
        for j in range(cfg.num_res_blocks):
            _resblock_spec(spec, f"down{i}.res{j}", ch, w, td)
            ch = w
            if cfg.has_attention[i]:
                _attn_spec(spec, f"down{i}.attn{j}", w, cfg.conditional, td, edge ** 3)
            skips.append(ch)
        if w != widths[-1]:
            spec[f"down{i}.ds.kernel"] = (3, 3, 3, w, w)
            spec[f"down{i}.ds.bias"] = (w,)
            edge //= 2
            skips.append(ch)
    w = widths[-1]
    _resblock_spec(spec, "mid.res0", ch, w, td)
    _attn_spec(spec, "mid.attn", w, cfg.conditional, td, edge ** 3)
    _resblock_spec(spec, "mid.res1", w, w, td)
    ch = w
    for i in reversed(range(len(widths))):
        w = widths[i]
        for j in range(cfg.num_res_blocks + 1):
            _resblock_spec(spec, f"up{i}.res{j}", ch + skips.pop(), w, td)
            ch = w
            if cfg.has_attention[i]:
                _attn_spec(spec, f"up{i}.attn{j}", w, cfg.conditional, td, edge ** 3)
        if i != 0:
            spec[f"up{i}.us.kernel"] = (3, 3, 3, w, w)
            spec[f"up{i}.us.bias"] = (w,)
            edge *= 2
    assert not skips
    _bn(spec, "out.norm", ch)
    spec["out.conv.kernel"] = (3, 3, 3, ch, C)
    spec["out.conv.bias"] = (C,)
    return spec


# scale argument of kernel_init per layer suffix (C:83,245,251,258,266,280,293,302-304,352,413); layers created
# without kernel_initializer (Cross block Dense/Conv, ContextMLP, Embedding) use Keras defaults.
_ZERO_SCALE_SUFFIX = (".conv2.kernel", "out.conv.kernel", ".proj.kernel")


def keras_default_weights(cfg: UNetConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """What Keras would create: kernel_init(1.0)/kernel_init(0.0) where the reference passes it (a1), glorot-uniform
    otherwise, zero biases, BN gamma=1 beta=0 mean=0 var=1, Embedding U(-0.05,0.05).  (Known answer (iv): the
    network then outputs ~0.)"""
    g = np.random.default_rng(seed)
    out = {}
    for name, shape in param_spec(cfg).items():
        if name.endswith(".kernel"):
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
            scale = 0.0 if name.endswith(_ZERO_SCALE_SUFFIX) else 1.0
            lim = kernel_init_limit(scale, fan_in, fan_out)
            arr = g.uniform(-lim, lim, size=shape)
        elif name.endswith(".table"):
            arr = g.uniform(-0.05, 0.05, size=shape)
        elif name.endswith((".gamma", ".var")):
            arr = np.ones(shape)
        else:
            arr = np.zeros(shape)
        out[name] = torch.from_numpy(arr.astype(np.float32))
    return out


def synthetic_weights(cfg: UNetConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded NON-degenerate weights for parity work: every kernel at scale 1 (the zero-scale layers
    included, otherwise parity is vacuous), small random biases, BN gamma~U(0.8,1.2), beta~N(0,0.1),
    mean~N(0,0.1), var~U(0.5,1.5), LN gamma~U(0.8,1.2), beta~N(0,0.1).  Draw order = param_spec order."""
    g = np.random.default_rng(seed)
    out = {}
    for name, shape in param_spec(cfg).items():
        if name.endswith(".kernel"):
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            lim = kernel_init_limit(1.0, shape[-2] * rf, shape[-1] * rf)
            arr = g.uniform(-lim, lim, size=shape)
        elif name.endswith(".table"):
            arr = g.normal(0.0, 1.0, size=shape)
        elif name.endswith(".gamma"):
            arr = g.uniform(0.8, 1.2, size=shape)
        elif name.endswith(".var"):
            arr = g.uniform(0.5, 1.5, size=shape)
        elif name.endswith((".beta", ".mean")):
            arr = g.normal(0.0, 0.1, size=shape)
        elif name.endswith(".bias"):
            arr = g.normal(0.0, 0.05, size=shape)
        else:
            raise KeyError(name)
        out[name] = torch.from_numpy(arr.astype(np.float32))
    return out


# --------------------------------------------------------------------------------------------------------------
# Keras layer semantics on NDHWC tensors
# --------------------------------------------------------------------------------------------------------------
def _conv3d(x, kernel, bias, stride=1):
    """Conv3D(padding='same').  k=3,s=1: pad 1/1; k=3,s=2 on even sizes: pad 0 before / 1 after (TF SAME);
    k=1: no pad.  Cross-correlation, kernel [kd,kh,kw,Cin,Cout]."""
    k = kernel.shape[0]
    xc = x.permute(0, 4, 1, 2, 3)
    w = kernel.permute(4, 3, 0, 1, 2)
    if k == 1:
        y = F.conv3d(xc, w, bias)
    elif stride == 1:
        y = F.conv3d(xc, w, bias, padding=1)
    else:
        pads = []
        for n in reversed(x.shape[1:4]):              # F.pad takes last dim first
            out = -(-n // stride)
            total = max((out - 1) * stride + k - n, 0)
            pads += [total // 2, total - total // 2]
        y = F.conv3d(F.pad(xc, pads), w, bias, stride=stride)
    return y.permute(0, 2, 3, 4, 1).contiguous()


def _bn_infer(x, W, name):
    """BatchNormalization(training=False): gamma*(x-mean)/sqrt(var+eps)+beta over the last axis — or, in the "group" variant,
    GroupNormalization(groups, epsilon=1e-3): per-sample moments over (D,H,W,C/groups)."""
    if _NORM["mode"] == "group" and x.dim() == 5:
        y = F.group_norm(x.permute(0, 4, 1, 2, 3), _NORM["groups"], W[f"{name}.gamma"], W[f"{name}.beta"], BN_EPS)
        return y.permute(0, 2, 3, 4, 1)
    return (x - W[f"{name}.mean"]) / torch.sqrt(W[f"{name}.var"] + BN_EPS) * W[f"{name}.gamma"] + W[f"{name}.beta"]


def _ln(x, W, name):
    return F.layer_norm(x, (x.shape[-1],), W[f"{name}.gamma"], W[f"{name}.beta"], LN_EPS)


def _dense(x, W, name):
    return x @ W[f"{name}.kernel"] + W[f"{name}.bias"]


def _swish(x):
    return x * torch.sigmoid(x)


def _upsample2(x):
    """UpSampling3D(size=2): nearest, y[2i+a,2j+b,2k+c]=x[i,j,k]  (C:290)."""
    return x.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)


def residual_block(W, name, x, temb):
    """C:238-271."""
    width = W[f"{name}.conv1.kernel"].shape[-1]
    if x.shape[-1] == width:
        residual = x
    else:
        residual = _conv3d(x, W[f"{name}.skip.kernel"], W[f"{name}.skip.bias"])
    te = _dense(_swish(temb), W, f"{name}.temb")[:, None, None, None, :]
    h = _swish(_bn_infer(x, W, f"{name}.norm1"))
    h = _conv3d(h, W[f"{name}.conv1.kernel"], W[f"{name}.conv1.bias"])
    h = h + te
    h = _swish(_bn_infer(h, W, f"{name}.norm2"))
    h = _conv3d(h, W[f"{name}.conv2.kernel"], W[f"{name}.conv2.bias"])
    return h + residual


def _attention(q, k, v, units):
    """C:163-184 with num_heads == 1: softmax(q k^T * units^-0.5) v over flattened tokens."""
    scale = float(units) ** (-0.5)
    score = torch.einsum("blc,bLc->blL", q, k) * scale
    score = torch.softmax(score, -1)
    return torch.einsum("blL,bLc->blc", score, v)


def cross_attention_block(W, name, x, context):
    """C:186-195.  ``context`` is ContextMLP's output [B or 1, h, w, d, u] (C:310-318)."""
    B, h, w, d, u = x.shape
    residual = x
    y = _bn_infer(x, W, f"{name}.norm")
    y = torch.relu(_conv3d(y, W[f"{name}.proj_in.kernel"], W[f"{name}.proj_in.bias"]))

    def attn(inp, ctx=None):
        q = _dense(inp, W, f"{name}.query")
        src = inp if ctx is None else ctx
        k = _dense(src, W, f"{name}.key")
        v = _dense(src, W, f"{name}.value")
        q, k, v = (z.reshape(z.shape[0], -1, u) for z in (q, k, v))
        if k.shape[0] != q.shape[0]:                  # B>1 with one context row: broadcast
            k = k.expand(q.shape[0], -1, -1)
            v = v.expand(q.shape[0], -1, -1)
        return _attention(q, k, v, u).reshape(B, h, w, d, u)

    a = attn(_ln(y, W, f"{name}.ln1")) + y
    a = attn(_ln(y, W, f"{name}.ln2"), context) + a
    m = torch.relu(_dense(_ln(y, W, f"{name}.ln3"), W, f"{name}.mlp.0"))
    a = _dense(m, W, f"{name}.mlp.1") + a
    out = torch.relu(_conv3d(a, W[f"{name}.proj_out.kernel"], W[f"{name}.proj_out.bias"]))
    return out + residual


def context_mlp(W, name, cemb, shape):
    """C:310-318: Dense(h*w*d*c, swish) then reshape [-1,h,w,d,c]."""
    z = _swish(_dense(cemb, W, f"{name}.ctx_mlp"))
    return z.reshape(-1, *shape)


def self_attention_block(W, name, x):
    """U:39-63 (C:85-109): returns BN(x) + proj(attn(BN(x)))."""
    B, h, w, d, u = x.shape
    xn = _bn_infer(x, W, f"{name}.norm")
    q = _dense(xn, W, f"{name}.query").reshape(B, -1, u)
    k = _dense(xn, W, f"{name}.key").reshape(B, -1, u)
    v = _dense(xn, W, f"{name}.value").reshape(B, -1, u)
    o = _attention(q, k, v, u).reshape(B, h, w, d, u)
    return xn + _dense(o, W, f"{name}.proj")


def unet_forward(W: Dict[str, torch.Tensor], cfg: UNetConfig, x: torch.Tensor, t: torch.Tensor,
                 context: Optional[torch.Tensor] = None, taps: Optional[dict] = None) -> torch.Tensor:
    """build_model's graph (C:348-415 / U:318-376) with training=False.

    x [B,S,S,S,C]; t [B] int64; context [B or 1, 1, 1] int64 (conditional only).  ``taps`` (optional dict)
    receives named intermediates for block-level tests."""
    dt = x.dtype
    widths = list(cfg.widths)
    _NORM["mode"], _NORM["groups"] = cfg.norm, cfg.norm_groups

    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    h = _conv3d(x, W["conv_in.kernel"], W["conv_in.bias"])
    temb = time_embedding(t, cfg.temb_dim).to(dt)
    temb = _swish(_dense(temb, W, "time_mlp.0"))
    temb = _dense(temb, W, "time_mlp.1")
    tap("temb", temb)
    cemb = None
    if cfg.conditional:
        cemb = W["ctx_embed.table"][context.reshape(context.shape[0], -1)[:, 0]]      # [B|1, td]

    def attn(name, v):
        if cfg.conditional:
            ctx = context_mlp(W, name, cemb, v.shape[1:])
            return cross_attention_block(W, name, v, ctx)
        return self_attention_block(W, name, v)

    skips = [h]
    for i, w in enumerate(widths):
        for j in range(cfg.num_res_blocks):
            h = tap(f"down{i}.res{j}", residual_block(W, f"down{i}.res{j}", h, temb))
            if cfg.has_attention[i]:
                h = tap(f"down{i}.attn{j}", attn(f"down{i}.attn{j}", h))
            skips.append(h)
        if w != widths[-1]:
            h = tap(f"down{i}.ds", _conv3d(h, W[f"down{i}.ds.kernel"], W[f"down{i}.ds.bias"], stride=2))
            skips.append(h)
    h = tap("mid.res0", residual_block(W, "mid.res0", h, temb))
    h = tap("mid.attn", attn("mid.attn", h))
    h = tap("mid.res1", residual_block(W, "mid.res1", h, temb))
    for i in reversed(range(len(widths))):
        for j in range(cfg.num_res_blocks + 1):
            h = torch.cat([h, skips.pop()], dim=-1)                                    # C:396 order [x, skip]
            h = tap(f"up{i}.res{j}", residual_block(W, f"up{i}.res{j}", h, temb))
            if cfg.has_attention[i]:
                h = tap(f"up{i}.attn{j}", attn(f"up{i}.attn{j}", h))
        if i != 0:
            h = tap(f"up{i}.us", _conv3d(_upsample2(h), W[f"up{i}.us.kernel"], W[f"up{i}.us.bias"]))
    h = _swish(_bn_infer(h, W, "out.norm"))
    return _conv3d(h, W["out.conv.kernel"], W["out.conv.bias"])


# --------------------------------------------------------------------------------------------------------------
# a13 sample, a14 generate, a15 train_step loss
# --------------------------------------------------------------------------------------------------------------
def ddpm_sample(b: Betas, x_t, pred_noise, t):
    """C:517-548: returns (posterior_mean, posterior 'log_variance' which is the variance), fp32 op order."""
    B = x_t.shape[0]
    g = lambda tab: tab[t].reshape(B, 1, 1, 1, 1).to(x_t.dtype)
    beta, sqa, ab, ab_prev = g(b.beta), g(b.sqrt_alpha), g(b.alpha_bar), g(b.alpha_bar_prev)
    sqab, sqab_prev, sq1ab = g(b.sqrt_alpha_bar), g(b.sqrt_alpha_bar_prev), g(b.sqrt_one_minus_alpha_bar)
    x_0 = (x_t - sq1ab * pred_noise) / sqab
    mean = (beta * sqab_prev / (1 - ab)) * x_0 + ((1 - ab_prev) * sqa / (1 - ab)) * x_t
    var = (1 - ab_prev) * beta / (1 - ab)
    return mean, var


def ddpm_step(b: Betas, x_t, pred_noise, t, noise):
    """Loop body C:571-573: clip the posterior mean to [-1,1], add sqrt(max(var,1e-20))*noise."""
    mean, var = ddpm_sample(b, x_t, pred_noise, t)
    mean = mean.clamp(-1, 1)
    return mean + torch.exp(0.5 * torch.log(torch.clamp_min(var, 1e-20))) * noise


def generate(W, cfg: UNetConfig, b: Betas, timesteps: int, x_T: torch.Tensor, noises, last_step: int = 0,
             context_value: Optional[int] = None, trajectory: Optional[list] = None):
    """C:550-575 (U:510-532) with the random draws injected: ``x_T`` replaces tf.random.normal(shape) and
    ``noises[i]`` is the draw used at step i (ignored at i == 0, where the reference uses 0)."""
    x = x_T
    B = x.shape[0]
    ctx = None
    if cfg.conditional:
        ctx = torch.tensor([[[int(context_value)]]], dtype=torch.int64)
    for i in range(timesteps - 1, last_step - 1, -1):
        z = noises[i] if i > 0 else torch.zeros_like(x)
        t = torch.full((B,), i, dtype=torch.int64)
        eps = unet_forward(W, cfg, x, t, ctx)
        x = ddpm_step(b, x, eps, t, z)
        if trajectory is not None:
            trajectory.append(x.clone())
    return x


def q_sample(b: Betas, latents, t, noise):
    """C:484-490 forward diffusion."""
    B = latents.shape[0]
    sqb = b.sqrt_alpha_bar[t].reshape(B, 1, 1, 1, 1)
    osqb = b.sqrt_one_minus_alpha_bar[t].reshape(B, 1, 1, 1, 1)
    return sqb * latents + osqb * noise


def train_loss(noise, pred_noise, global_bs: int, lc: int):
    """C:496-499 with compile(loss=MeanSquaredError(reduction=SUM)) (main_conditional_dm.py:149-152):
    mean over the channel axis, SUM over b*d*h*w, divided by global_bs*lc^4."""
    mse_sum = ((noise - pred_noise) ** 2).mean(-1).sum()
    return mse_sum / (global_bs * lc * lc * lc * lc * 1.0)


# ==============================================================================================================
# next-1: the VQ-VAE bracket around the sampler — reference networks/vqvae3d_monai.py (abbreviated V:)
# Encoder V:237-306, VectorQuantizer V:112-177, VQVAEResidualUnit V:218-234, Decoder V:309-391, as wired by
# DiffusionModel (C:425-460): latents = quantizer(encoder(images))[0] in training, decoder(latents) after sampling.
# Keras semantics: Conv3D(k=4, strides=2, padding="same") pads 1/1; Conv3DTranspose(k=4, strides=2, padding="same") has
# out = 2*in and kernel [kd,kh,kw,Cout,Cin]; PReLU() has a full-shape slope [D,H,W,C]; BatchNormalization eps 1e-3.
# ==============================================================================================================
@dataclass
class VQVAEConfig:
    in_channels: int = 1
    out_channels: int = 1
    num_channels: Sequence[int] = (32, 64, 128, 256)
    num_res_layers: int = 5
    num_res_channels: Sequence[int] = (32, 64, 128, 256)
    num_embeddings: int = 1024
    embedding_dim: int = 256
    input_size: int = 128              # PReLU slopes are full-shape, so the model is tied to its input size
    output_act: bool = False

    @property
    def latent_size(self) -> int:
        return self.input_size >> len(self.num_channels)


def _res_unit_spec(spec, name, ch, rc, edge):
    spec[f"{name}.conv1.kernel"] = (3, 3, 3, ch, rc)
    spec[f"{name}.conv1.bias"] = (rc,)
    spec[f"{name}.conv2.kernel"] = (3, 3, 3, rc, ch)
    spec[f"{name}.conv2.bias"] = (ch,)
    _bn(spec, f"{name}.bn", ch)
    spec[f"{name}.prelu.alpha"] = (edge, edge, edge, ch)


def vqvae_param_spec(cfg: VQVAEConfig) -> Dict[str, Tuple[int, ...]]:
    spec: Dict[str, Tuple[int, ...]] = {}
    ch_in, edge = cfg.in_channels, cfg.input_size
    for i, ch in enumerate(cfg.num_channels):                          # Encoder V:266-291
        edge //= 2
        spec[f"enc.down{i}.kernel"] = (4, 4, 4, ch_in, ch)
        spec[f"enc.down{i}.bias"] = (ch,)
        for j in range(cfg.num_res_layers):
            _res_unit_spec(spec, f"enc.l{i}.res{j}", ch, cfg.num_res_channels[i], edge)
        ch_in = ch
    spec["enc.out.kernel"] = (3, 3, 3, ch_in, cfg.embedding_dim)       # V:296-301
    spec["enc.out.bias"] = (cfg.embedding_dim,)
    spec["enc.out_prelu.alpha"] = (edge, edge, edge, cfg.embedding_dim)
    spec["vq.embeddings"] = (cfg.embedding_dim, cfg.num_embeddings)    # V:125-131
    rev, rrev = list(reversed(cfg.num_channels)), list(reversed(cfg.num_res_channels))
    spec["dec.in.kernel"] = (3, 3, 3, cfg.embedding_dim, rev[0])       # V:346-350
    spec["dec.in.bias"] = (rev[0],)
    spec["dec.in_prelu.alpha"] = (edge, edge, edge, rev[0])
    for i, ch in enumerate(rev):                                       # V:353-381
        for j in range(cfg.num_res_layers):
            _res_unit_spec(spec, f"dec.l{i}.res{j}", ch, rrev[i], edge)
        out = cfg.out_channels if i == len(rev) - 1 else rev[i + 1]
        spec[f"dec.up{i}.kernel"] = (4, 4, 4, out, ch)                 # Conv3DTranspose: [k,k,k,Cout,Cin]
        spec[f"dec.up{i}.bias"] = (out,)
        edge *= 2
    return spec


def vqvae_synthetic_weights(cfg: VQVAEConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded non-degenerate weights: glorot-uniform kernels, small biases, random BN statistics, PReLU slopes U(0.05, 0.45)
    (Keras initialises them to 0 = ReLU, which would hide the slope path), codebook N(0,1)."""
    g = np.random.default_rng(seed)
    out = {}
    for name, shape in vqvae_param_spec(cfg).items():
        if name.endswith(".kernel"):
            rf = int(np.prod(shape[:3]))
            lim = math.sqrt(6.0 / (shape[3] * rf + shape[4] * rf))
            arr = g.uniform(-lim, lim, size=shape)
        elif name.endswith(".alpha"):
            arr = g.uniform(0.05, 0.45, size=shape)
        elif name.endswith(".embeddings"):
            arr = g.normal(0.0, 1.0, size=shape)
        elif name.endswith(".gamma"):
            arr = g.uniform(0.8, 1.2, size=shape)
        elif name.endswith(".var"):
            arr = g.uniform(0.5, 1.5, size=shape)
        elif name.endswith((".beta", ".mean")):
            arr = g.normal(0.0, 0.1, size=shape)
        elif name.endswith(".bias"):
            arr = g.normal(0.0, 0.05, size=shape)
        else:
            raise KeyError(name)
        out[name] = torch.from_numpy(arr.astype(np.float32))
    return out


def _conv3d_k4s2(x, kernel, bias):
    """Conv3D(k=4, strides=2, padding='same') on even sizes: total pad 2 -> 1 before, 1 after (V:269-273)."""
    y = F.conv3d(x.permute(0, 4, 1, 2, 3), kernel.permute(4, 3, 0, 1, 2), bias, stride=2, padding=1)
    return y.permute(0, 2, 3, 4, 1).contiguous()


def _conv3d_transpose_k4s2(x, kernel, bias):
    """Conv3DTranspose(k=4, strides=2, padding='same'): out = 2*in; Keras kernel [kd,kh,kw,Cout,Cin] (V:373-377).
    It is the transpose of the k4/s2 'same' convolution above, i.e. conv_transpose3d(stride 2, padding 1)."""
    w = kernel.permute(4, 3, 0, 1, 2)                   # torch conv_transpose3d weight: [Cin, Cout, kd, kh, kw]
    y = F.conv_transpose3d(x.permute(0, 4, 1, 2, 3), w, bias, stride=2, padding=1)
    return y.permute(0, 2, 3, 4, 1).contiguous()


def _prelu(x, alpha):
    return torch.where(x > 0, x, alpha * x)


def vq_residual_unit(W, name, x):
    """V:225-234: ReLU(x + PReLU(BN(Conv3(relu(Conv3(x))))))."""
    h = torch.relu(_conv3d(x, W[f"{name}.conv1.kernel"], W[f"{name}.conv1.bias"]))
    h = _conv3d(h, W[f"{name}.conv2.kernel"], W[f"{name}.conv2.bias"])
    h = _prelu(_bn_infer(h, W, f"{name}.bn"), W[f"{name}.prelu.alpha"])
    return torch.relu(x + h)


def vq_encoder(W, cfg: VQVAEConfig, x):
    h = x
    for i in range(len(cfg.num_channels)):
        h = torch.relu(_conv3d_k4s2(h, W[f"enc.down{i}.kernel"], W[f"enc.down{i}.bias"]))
        for j in range(cfg.num_res_layers):
            h = vq_residual_unit(W, f"enc.l{i}.res{j}", h)
    h = _conv3d(h, W["enc.out.kernel"], W["enc.out.bias"])
    return _prelu(h, W["enc.out_prelu.alpha"])


def vq_code_indices(W, z_flat):
    """V:164-177: argmin_k(|z|^2 + |e_k|^2 - 2 z.e_k)."""
    E = W["vq.embeddings"]
    sim = z_flat @ E
    dist = (z_flat ** 2).sum(1, keepdim=True) + (E ** 2).sum(0) - 2 * sim
    return torch.argmin(dist, dim=1)


def vq_quantize(W, z):
    """V:133-162 forward value: (quantised latents, perplexity)."""
    E = W["vq.embeddings"]
    flat = z.reshape(-1, E.shape[0])
    idx = vq_code_indices(W, flat)
    q = E.t()[idx].reshape(z.shape)
    probs = torch.bincount(idx, minlength=E.shape[1]).to(z.dtype) / idx.numel()
    perplexity = torch.exp(-(probs * torch.log(probs + 1e-10)).sum())
    return q, perplexity, idx


def vq_decoder(W, cfg: VQVAEConfig, z):
    h = _prelu(_conv3d(z, W["dec.in.kernel"], W["dec.in.bias"]), W["dec.in_prelu.alpha"])
    n = len(cfg.num_channels)
    for i in range(n):
        for j in range(cfg.num_res_layers):
            h = vq_residual_unit(W, f"dec.l{i}.res{j}", h)
        h = _conv3d_transpose_k4s2(h, W[f"dec.up{i}.kernel"], W[f"dec.up{i}.bias"])
        if i != n - 1:
            h = torch.relu(h)
    return torch.relu(h) if cfg.output_act else h


# ==============================================================================================================
# BASELINE config 5 names the OTHER autoencoder, reference networks/vqgan.py (abbreviated G:): Encoder G:287-375, Decoder G:378-475,
# VQVAEResidualUnit G:257-284, VectorQuantizer G:151-216, wired by VQGAN.call G:699-703 on x = concat[img, mask] (G:726-727) and
# built by main_exp_vqgan.py:23-38 (one (stride 2, k 4, "same") level per channel_list entry, in/out channels 2).  It differs from
# the monai VQ-VAE above by a BatchNormalization + PReLU after every strided conv (G:317-341), a BatchNormalization after the
# decoder's first conv (G:415-416) and after every Conv3DTranspose (G:457), PReLU (not ReLU) between decoder levels (G:459-467).
# ==============================================================================================================
def vqgan_param_spec(cfg: VQVAEConfig) -> Dict[str, Tuple[int, ...]]:
    base = vqvae_param_spec(cfg)
    n = len(cfg.num_channels)
    rev = list(reversed(cfg.num_channels))
    spec: Dict[str, Tuple[int, ...]] = {}
    for name, shape in base.items():
        spec[name] = shape
        for i in range(n):
            if name == f"enc.down{i}.bias":
                _bn(spec, f"enc.down{i}.bn", cfg.num_channels[i])
                e = cfg.input_size >> (i + 1)
                spec[f"enc.down{i}.prelu.alpha"] = (e, e, e, cfg.num_channels[i])
            if name == f"dec.up{i}.bias":
                out = cfg.out_channels if i == n - 1 else rev[i + 1]
                _bn(spec, f"dec.up{i}.bn", out)
                if i != n - 1:
                    e = cfg.latent_size << (i + 1)
                    spec[f"dec.up{i}.prelu.alpha"] = (e, e, e, out)
        if name == "dec.in.bias":
            _bn(spec, "dec.in.bn", rev[0])
    return spec


def vqgan_synthetic_weights(cfg: VQVAEConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = np.random.default_rng(seed)
    out = {}
    for name, shape in vqgan_param_spec(cfg).items():
        if name.endswith(".kernel"):
            rf = int(np.prod(shape[:3]))
            lim = math.sqrt(6.0 / (shape[3] * rf + shape[4] * rf))
            arr = g.uniform(-lim, lim, size=shape)
        elif name.endswith(".alpha"):
            arr = g.uniform(0.05, 0.45, size=shape)
        elif name.endswith(".embeddings"):
            arr = g.normal(0.0, 1.0, size=shape)
        elif name.endswith(".gamma"):
            arr = g.uniform(0.8, 1.2, size=shape)
        elif name.endswith(".var"):
            arr = g.uniform(0.5, 1.5, size=shape)
        elif name.endswith((".beta", ".mean")):
            arr = g.normal(0.0, 0.1, size=shape)
        elif name.endswith(".bias"):
            arr = g.normal(0.0, 0.05, size=shape)
        else:
            raise KeyError(name)
        out[name] = torch.from_numpy(arr.astype(np.float32))
    return out


def vqgan_encoder(W, cfg: VQVAEConfig, x):
    """G:317-369: per level Conv3D(k4,s2,same) -> BatchNormalization -> PReLU -> res units; then Conv3D(k3) -> PReLU."""
    h = x
    for i in range(len(cfg.num_channels)):
        h = _conv3d_k4s2(h, W[f"enc.down{i}.kernel"], W[f"enc.down{i}.bias"])
        h = _prelu(_bn_infer(h, W, f"enc.down{i}.bn"), W[f"enc.down{i}.prelu.alpha"])
        for j in range(cfg.num_res_layers):
            h = vq_residual_unit(W, f"enc.l{i}.res{j}", h)
    h = _conv3d(h, W["enc.out.kernel"], W["enc.out.bias"])
    return _prelu(h, W["enc.out_prelu.alpha"])


def vqgan_decoder(W, cfg: VQVAEConfig, z):
    """G:415-470: Conv3D(k3) -> BatchNormalization -> PReLU; per level res units -> Conv3DTranspose(k4,s2,same) -> BatchNormalization
    [-> PReLU unless last]; optional output ReLU."""
    h = _conv3d(z, W["dec.in.kernel"], W["dec.in.bias"])
    h = _prelu(_bn_infer(h, W, "dec.in.bn"), W["dec.in_prelu.alpha"])
    n = len(cfg.num_channels)
    for i in range(n):
        for j in range(cfg.num_res_layers):
            h = vq_residual_unit(W, f"dec.l{i}.res{j}", h)
        h = _conv3d_transpose_k4s2(h, W[f"dec.up{i}.kernel"], W[f"dec.up{i}.bias"])
        h = _bn_infer(h, W, f"dec.up{i}.bn")
        if i != n - 1:
            h = _prelu(h, W[f"dec.up{i}.prelu.alpha"])
    return torch.relu(h) if cfg.output_act else h

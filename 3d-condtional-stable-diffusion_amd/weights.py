"""U-Net configuration, weight inventory and initialisers (host side, NumPy).

Mirrors ``kernel_init`` and the layer-creation order of ``build_model`` in the reference
(networks/conditional_dm3d.py:17-21, 324-415; networks/dm3d.py:11-15, 294-376).  Weights are kept in the reference's
Keras layouts — Conv3D ``[kd,kh,kw,Cin,Cout]``, Dense ``[in,out]`` — under flat dotted names, so a state dict converted
from a Keras checkpoint drops in unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np


@dataclass
class UNetConfig:
    """``build_model`` arguments (conditional_dm3d.py:324-335) + the module constant ``first_conv_channels`` (:321)."""

    img_size: int
    img_channels: int
    widths: Sequence[int] = (64, 128, 256)
    has_attention: Sequence[bool] = (False, False, True, True)
    num_res_blocks: int = 2
    conditional: bool = True
    first_conv_channels: Optional[int] = None
    context_dim: int = 1
    norm_groups: int = 8            # plumbed but unused by the reference (BatchNormalization replaces GroupNorm) ...
    norm: str = "batch"             # ... unless norm="group": the GroupNormalization lines it keeps commented out (:77, 254, 261, 409)

    def __post_init__(self):
        if self.first_conv_channels is None:
            self.first_conv_channels = 32 if self.conditional else 64
        self.widths = tuple(int(w) for w in self.widths)
        self.has_attention = tuple(bool(a) for a in self.has_attention)
        if len(self.has_attention) < len(self.widths):
            raise ValueError("has_attention needs one entry per width")
        if self.norm not in ("batch", "group"):
            raise ValueError("norm must be 'batch' or 'group'")

    @property
    def temb_dim(self) -> int:
        return self.first_conv_channels * 4


def kernel_init(scale: float):
    """reference conditional_dm3d.py:17-21 — VarianceScaling(max(scale,1e-10), 'fan_avg', 'uniform').

    Returns ``init(shape, rng) -> float32 array`` drawing U(-l, l), l = sqrt(3*scale/fan_avg), with Keras' fan
    computation (receptive field x channels)."""
    scale = max(scale, 1e-10)

    def init(shape, rng: np.random.Generator):
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_avg = (shape[-2] * rf + shape[-1] * rf) / 2.0
        lim = math.sqrt(3.0 * scale / fan_avg)
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    return init


def _bn(spec, name, c):
    for s in ("gamma", "beta", "mean", "var"):
        spec[f"{name}.{s}"] = (c,)


def _res(spec, name, cin, width, td):
    if cin != width:
        spec[f"{name}.skip.kernel"] = (1, 1, 1, cin, width)
        spec[f"{name}.skip.bias"] = (width,)
    spec[f"{name}.temb.kernel"] = (td, width)
    spec[f"{name}.temb.bias"] = (width,)
    _bn(spec, f"{name}.norm1", cin)
    spec[f"{name}.conv1.kernel"] = (3, 3, 3, cin, width)
    spec[f"{name}.conv1.bias"] = (width,)
    _bn(spec, f"{name}.norm2", width)
    spec[f"{name}.conv2.kernel"] = (3, 3, 3, width, width)
    spec[f"{name}.conv2.bias"] = (width,)


def _attn(spec, name, u, conditional, td, tokens):
    _bn(spec, f"{name}.norm", u)
    if conditional:
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
        spec[f"{name}.ctx_mlp.kernel"] = (td, tokens * u)
        spec[f"{name}.ctx_mlp.bias"] = (tokens * u,)
    else:
        for d in ("query", "key", "value", "proj"):
            spec[f"{name}.{d}.kernel"] = (u, u)
            spec[f"{name}.{d}.bias"] = (u,)


@dataclass
class BlockInfo:
    kind: str          # "res" | "attn" | "down" | "up"
    name: str
    edge: int          # spatial edge of the block's output
    cin: int = 0       # res: channels of x (first input); down/up: channels
    cskip: int = 0     # res on the up path: channels of the concatenated skip tensor
    cout: int = 0


def walk(cfg: UNetConfig):
    """Yields the blocks of build_model in execution order (conditional_dm3d.py:363-407) with their channel counts.
    Returns (blocks, spec).  ``blocks`` drives the plan builder; ``spec`` is name -> Keras shape."""
    spec: Dict[str, Tuple[int, ...]] = {}
    blocks = []
    f0, td, C, S = cfg.first_conv_channels, cfg.temb_dim, cfg.img_channels, cfg.img_size
    widths = list(cfg.widths)
    spec["conv_in.kernel"] = (3, 3, 3, C, f0)
    spec["conv_in.bias"] = (f0,)
    for i in (0, 1):
        spec[f"time_mlp.{i}.kernel"] = (td, td)
        spec[f"time_mlp.{i}.bias"] = (td,)
    if cfg.conditional:
        spec["ctx_embed.table"] = (cfg.context_dim + 1, td)
    ch, edge = f0, S
    skips = [ch]
    for i, w in enumerate(widths):
        for j in range(cfg.num_res_blocks):
            _res(spec, f"down{i}.res{j}", ch, w, td)
            blocks.append(BlockInfo("res", f"down{i}.res{j}", edge, ch, 0, w))
            ch = w
            if cfg.has_attention[i]:
                _attn(spec, f"down{i}.attn{j}", w, cfg.conditional, td, edge ** 3)
                blocks.append(BlockInfo("attn", f"down{i}.attn{j}", edge, w, 0, w))
            blocks.append(BlockInfo("push", "", edge, ch))
            skips.append(ch)
        if w != widths[-1]:
            spec[f"down{i}.ds.kernel"] = (3, 3, 3, w, w)
            spec[f"down{i}.ds.bias"] = (w,)
            edge = -(-edge // 2)
            blocks.append(BlockInfo("down", f"down{i}.ds", edge, w, 0, w))
            blocks.append(BlockInfo("push", "", edge, ch))
            skips.append(ch)
    w = widths[-1]
    _res(spec, "mid.res0", ch, w, td)
    blocks.append(BlockInfo("res", "mid.res0", edge, ch, 0, w))
    _attn(spec, "mid.attn", w, cfg.conditional, td, edge ** 3)
    blocks.append(BlockInfo("attn", "mid.attn", edge, w, 0, w))
    _res(spec, "mid.res1", w, w, td)
    blocks.append(BlockInfo("res", "mid.res1", edge, w, 0, w))
    ch = w
    for i in reversed(range(len(widths))):
        w = widths[i]
        for j in range(cfg.num_res_blocks + 1):
            cs = skips.pop()
            _res(spec, f"up{i}.res{j}", ch + cs, w, td)
            blocks.append(BlockInfo("res", f"up{i}.res{j}", edge, ch, cs, w))
            ch = w
            if cfg.has_attention[i]:
                _attn(spec, f"up{i}.attn{j}", w, cfg.conditional, td, edge ** 3)
                blocks.append(BlockInfo("attn", f"up{i}.attn{j}", edge, w, 0, w))
        if i != 0:
            spec[f"up{i}.us.kernel"] = (3, 3, 3, w, w)
            spec[f"up{i}.us.bias"] = (w,)
            edge *= 2
            blocks.append(BlockInfo("up", f"up{i}.us", edge, w, 0, w))
    if skips:
        raise AssertionError("skip stack not empty")
    _bn(spec, "out.norm", ch)
    spec["out.conv.kernel"] = (3, 3, 3, ch, C)
    spec["out.conv.bias"] = (C,)
    return blocks, spec


def param_spec(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    return walk(cfg)[1]


_ZERO_SCALE = (".conv2.kernel", "out.conv.kernel", ".proj.kernel")   # kernel_init(0.0): :83, 266, 413


def keras_init_weights(cfg: UNetConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Fresh weights as Keras would create them: kernel_init(1.0 | 0.0) where the reference passes it, glorot-uniform
    for layers built without an initializer, zero biases, BatchNorm/LayerNorm identity, Embedding U(-0.05, 0.05)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in param_spec(cfg).items():
        if name.endswith(".kernel"):
            out[name] = kernel_init(0.0 if name.endswith(_ZERO_SCALE) else 1.0)(shape, rng)
        elif name.endswith(".table"):
            out[name] = rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)
        elif name.endswith((".gamma", ".var")):
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def synthetic_weights(cfg: UNetConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded non-degenerate weights for benchmarks and parity runs (there is no network to fetch a checkpoint):
    every kernel at kernel_init(1.0) — including the reference's zero-scale layers, which would otherwise make the
    network output ~0 — random norm statistics and small biases.  The draw order is ``param_spec`` order."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in param_spec(cfg).items():
        if name.endswith(".kernel"):
            out[name] = kernel_init(1.0)(shape, rng)
        elif name.endswith(".table"):
            out[name] = rng.normal(0.0, 1.0, size=shape).astype(np.float32)
        elif name.endswith(".gamma"):
            out[name] = rng.uniform(0.8, 1.2, size=shape).astype(np.float32)
        elif name.endswith(".var"):
            out[name] = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif name.endswith((".beta", ".mean")):
            out[name] = rng.normal(0.0, 0.1, size=shape).astype(np.float32)
        elif name.endswith(".bias"):
            out[name] = rng.normal(0.0, 0.05, size=shape).astype(np.float32)
        else:
            raise KeyError(name)
    return out


def upsample_parity_kernels(kernel: np.ndarray) -> np.ndarray:
    """[3,3,3,Cin,Cout] kernel of an UpSample conv (conditional_dm3d.py:288-296) -> [8,2,2,2,Cin,Cout]: for output parity
    p = 4a+2b+c the 2x2x2 kernel acting on the low-resolution tensor (taps of the k3 kernel that read the same source voxel
    of the nearest-2x upsampled tensor are summed).  Host mirror of the device packing, used to size the H3 weight scale."""
    groups = {0: ([0], [1, 2]), 1: ([0, 1], [2])}           # parity -> k3 taps feeding k2 tap 0 / 1
    k = np.asarray(kernel, dtype=np.float64)
    out = np.zeros((8, 2, 2, 2) + k.shape[3:], dtype=np.float64)
    for p in range(8):
        a, b, c = p >> 2, (p >> 1) & 1, p & 1
        for td in range(2):
            for th in range(2):
                for tw in range(2):
                    sel = k[np.ix_(groups[a][td], groups[b][th], groups[c][tw])]
                    out[p, td, th, tw] = sel.sum(axis=(0, 1, 2))
    return out

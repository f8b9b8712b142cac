"""Drop-in for the reference's ``networks/conditional_dm3d.py`` (the conditional 3D U-Net + DDPM wrapper).

Same public names and call signatures: ``kernel_init`` (:17), ``TimeEmbedding`` (:198), ``Betas`` (:215),
``first_conv_channels`` (:321), ``build_model`` (:324), ``DiffusionModel`` (:418).
"""
from __future__ import annotations

import numpy as np

from ..betas import Betas, time_embedding_table
from ..diffusion import DiffusionModel
from ..unet import UNet
from ..weights import UNetConfig, kernel_init

first_conv_channels = 32


class TimeEmbedding:
    """conditional_dm3d.py:198-212 as a host-side callable (the kernels consume it as a precomputed table)."""

    def __init__(self, dim, **kwargs):
        self.dim = dim
        self.half_dim = dim // 2

    def __call__(self, inputs):
        return time_embedding_table(np.asarray(inputs), self.dim)


def swish(x):
    return x / (1.0 + np.exp(-x))


def build_model(img_size, img_channels, widths, has_attention, has_cross_attention=None, num_res_blocks=2, norm_groups=8,
                interpolation="nearest", activation_fn=swish, context_dim=1, *, device="cuda", seed=0, precision=None, norm="batch"):
    """conditional_dm3d.py:324-415.  Returns a callable ``net([image, time, context]) -> eps``.

    ``norm_groups`` and ``interpolation`` are accepted and ignored exactly as the reference ignores them (BatchNorm
    replaces GroupNorm :77-78, UpSampling3D is always nearest :290); only swish is wired as ``activation_fn``."""
    if has_cross_attention and not context_dim:
        raise ValueError("Context dim can not be None if has_cross_attention is not None")     # :343-346
    if activation_fn is not swish and getattr(activation_fn, "__name__", "") not in ("swish", "silu"):
        raise ValueError("only the swish activation of the reference is implemented in the fused kernels")
    cfg = UNetConfig(img_size=img_size, img_channels=img_channels, widths=widths, has_attention=has_attention,
                     num_res_blocks=num_res_blocks, conditional=True, first_conv_channels=first_conv_channels,
                     context_dim=context_dim, norm_groups=norm_groups, norm=norm)
    return UNet(cfg, device=device, seed=seed, precision=precision)

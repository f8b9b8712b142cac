"""Drop-in for the reference's ``networks/dm3d.py`` (the unconditional 3D U-Net + DDPM wrapper).

``kernel_init`` (:11), ``TimeEmbedding`` (:177), ``Betas`` (:194), ``first_conv_channels`` (:291), ``build_model``
(:294), ``DiffusionModel`` (:379).
"""
from __future__ import annotations

from ..betas import Betas
from ..diffusion import UnconditionalDiffusionModel as DiffusionModel
from ..unet import UNet
from ..weights import UNetConfig, kernel_init
from .conditional_dm3d import TimeEmbedding, swish

first_conv_channels = 64


def build_model(img_size, img_channels, widths, has_attention, has_cross_attention=None, num_res_blocks=2, norm_groups=8,
                interpolation="nearest", activation_fn=swish, context_dim=None, *, device="cuda", seed=0, precision=None, norm="batch"):
    """dm3d.py:294-376.  Returns a callable ``net([image, time]) -> eps``."""
    if has_cross_attention and not context_dim:
        raise ValueError("Context dim can not be None if has_cross_attention is not None")     # :313-316
    cfg = UNetConfig(img_size=img_size, img_channels=img_channels, widths=widths, has_attention=has_attention,
                     num_res_blocks=num_res_blocks, conditional=False, first_conv_channels=first_conv_channels,
                     norm_groups=norm_groups, norm=norm)
    return UNet(cfg, device=device, seed=seed, precision=precision)

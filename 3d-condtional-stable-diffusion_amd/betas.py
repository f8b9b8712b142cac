"""Noise schedule and sinusoidal time embedding tables (host side).

``Betas`` follows reference networks/conditional_dm3d.py:215-235: float64 NumPy arithmetic, stored as float32.
``time_embedding_table`` follows ``TimeEmbedding`` (:198-212) in its float32 operation order.
"""
from __future__ import annotations

import math

import numpy as np

BETAS_FIELDS = ("beta", "alpha", "sqrt_alpha", "alpha_bar", "alpha_bar_prev", "sqrt_alpha_bar",
                "sqrt_alpha_bar_prev", "sqrt_one_minus_alpha_bar")


class Betas:
    def __init__(self, timesteps: int):
        beta = np.linspace(0.0001, 0.02, timesteps)
        alpha = 1 - beta
        sqrt_alpha = np.sqrt(alpha)
        alpha_bar = np.cumprod(alpha, 0)
        alpha_bar_prev = np.append(1.0, alpha_bar[:-1])
        sqrt_alpha_bar = np.sqrt(alpha_bar)
        sqrt_alpha_bar_prev = np.sqrt(alpha_bar_prev)
        sqrt_one_minus_alpha_bar = np.sqrt(1 - alpha_bar)
        self.timesteps = int(timesteps)
        self.beta = beta.astype(np.float32)
        self.alpha = alpha.astype(np.float32)
        self.sqrt_alpha = sqrt_alpha.astype(np.float32)
        self.alpha_bar = alpha_bar.astype(np.float32)
        self.alpha_bar_prev = alpha_bar_prev.astype(np.float32)
        self.sqrt_alpha_bar = sqrt_alpha_bar.astype(np.float32)
        self.sqrt_alpha_bar_prev = sqrt_alpha_bar_prev.astype(np.float32)
        self.sqrt_one_minus_alpha_bar = sqrt_one_minus_alpha_bar.astype(np.float32)
        self._device = None

    def device_tables(self, device):
        """The eight tables as one [8, T] float32 device tensor (row order = BETAS_FIELDS), uploaded once."""
        import torch
        if self._device is None or self._device.device != torch.device(device):
            host = np.stack([getattr(self, f) for f in BETAS_FIELDS]).astype(np.float32)
            self._device = torch.from_numpy(host).to(device)
        return self._device


def time_embedding_table(t, dim: int) -> np.ndarray:
    """TimeEmbedding(dim)(t) for an integer vector t, float32 throughout, on the host:
    f = exp(arange(half) * -(ln(10000)/(half-1))), emb = [sin(t*f), cos(t*f)].

    At t ~ 1000 one ulp in f moves the argument by ~6e-5 rad, so the exp/sin/cos implementation matters at the 1e-5
    level; PyTorch's CPU float32 kernels are used (host-side table construction, not the device path)."""
    import torch
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    arg = torch.as_tensor(np.asarray(t)).to(torch.float32)[:, None] * freqs[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1).numpy().astype(np.float32)

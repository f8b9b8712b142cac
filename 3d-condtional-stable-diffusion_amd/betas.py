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
    """TimeEmbedding(dim)(t) for an integer vector t, float32 throughout:
    f = exp(arange(half) * -(ln(10000)/(half-1))), emb = [sin(t*f), cos(t*f)]."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    freqs = np.exp(np.arange(half, dtype=np.float32) * np.float32(-emb)).astype(np.float32)
    arg = np.asarray(t).astype(np.float32)[:, None] * freqs[None, :]
    return np.concatenate([np.sin(arg), np.cos(arg)], axis=-1).astype(np.float32)

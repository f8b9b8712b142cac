"""ORACLE (training half) — CPU restatement of ``DiffusionModel.train_step`` (reference networks/conditional_dm3d.py:471-510,
compiled at main_conditional_dm.py:149-154) on top of ``oracle/ref_torch.py``.

TEST INFRASTRUCTURE ONLY (see the header of ref_torch.py): the product never imports it.  PARITY UNPINNED: the reference ships no
training fixtures and TensorFlow is absent, so gradients are checked against ``torch.autograd`` applied to this restatement.

What Keras does in one ``train_step`` (C = conditional_dm3d.py):
  C:474-476  t ~ U{0..T-1}                                   (injected here)
  C:478      latents, _ = quantizer(encoder(images))          (injected here: train on given latents)
  C:481      noise ~ N(0,1)                                   (injected)
  C:484-490  noisy = sqrt_alpha_bar[t]*latents + sqrt_one_minus_alpha_bar[t]*noise
  C:493      pred = network([noisy, t, context], training=True)   -> BatchNormalization uses batch statistics (mean and BIASED
             variance over B,D,H,W) and updates its moving averages with momentum 0.99.  For rank-5 inputs Keras (TF2 behaviour,
             fused=None) runs tf.nn.fused_batch_norm, whose moving-variance update uses the unbiased batch variance n/(n-1)*var.
  C:496-499  lo = MeanSquaredError(reduction=SUM)(noise, pred) / (global_bs * lc^4)
  C:501-504  Adam(lr) on network.trainable_weights (everything except the BatchNormalization moving statistics)
  C:507-510  loss_tracker (Mean) -> {"loss": ...}
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ref_torch as rt

BN_MOMENTUM = 0.99          # keras.layers.BatchNormalization default
ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-7      # keras.optimizers.Adam defaults


def is_trainable(name: str) -> bool:
    """network.trainable_weights: all but the BatchNormalization moving mean / variance."""
    return not name.endswith((".mean", ".var"))


def unet_forward_train(W: Dict[str, torch.Tensor], cfg: rt.UNetConfig, x, t, context=None, stats: Optional[dict] = None, taps=None):
    """build_model's graph with training=True: every BatchNormalization normalises with the batch mean / biased variance.
    ``stats`` (optional dict) receives name -> (batch_mean, biased_batch_var, n) for the moving-average update."""
    if cfg.norm != "batch":
        raise ValueError("training is restated for the BatchNormalization network the reference runs")
    saved = rt._bn_infer

    def bn_train(xv, Wd, name):
        dims = tuple(range(xv.dim() - 1))
        mean = xv.mean(dims)
        var = xv.var(dims, unbiased=False)
        if stats is not None:
            n = xv.numel() // xv.shape[-1]
            stats[name] = (mean.detach(), var.detach(), n)
        return (xv - mean) / torch.sqrt(var + rt.BN_EPS) * Wd[f"{name}.gamma"] + Wd[f"{name}.beta"]

    rt._bn_infer = bn_train
    try:
        return rt.unet_forward(W, cfg, x, t, context, taps=taps)
    finally:
        rt._bn_infer = saved


def moving_update(W: Dict[str, torch.Tensor], stats: dict, unbiased: bool = True) -> Dict[str, torch.Tensor]:
    """The BatchNormalization moving statistics after one training forward (momentum 0.99)."""
    out = {}
    for name, (mean, var, n) in stats.items():
        v = var * (n / (n - 1.0)) if (unbiased and n > 1) else var
        out[f"{name}.mean"] = W[f"{name}.mean"] * BN_MOMENTUM + mean * (1 - BN_MOMENTUM)
        out[f"{name}.var"] = W[f"{name}.var"] * BN_MOMENTUM + v * (1 - BN_MOMENTUM)
    return out


def loss_and_grads(W: Dict[str, torch.Tensor], cfg: rt.UNetConfig, b: rt.Betas, latents, t, noise, context, global_bs: int, lc: int,
                   stats: Optional[dict] = None, taps=None):
    """(loss, {name: dloss/dweight} for the trainable weights) via torch.autograd over the restated forward."""
    Wg = {k: (v.clone().requires_grad_(True) if is_trainable(k) else v) for k, v in W.items()}
    noisy = rt.q_sample(b, latents, t, noise).to(latents.dtype)
    pred = unet_forward_train(Wg, cfg, noisy, t, context, stats=stats, taps=taps)
    loss = rt.train_loss(noise, pred, global_bs, lc)
    names = [k for k in Wg if Wg[k].requires_grad]
    grads = torch.autograd.grad(loss, [Wg[k] for k in names], allow_unused=True)
    return loss.detach(), {k: (g if g is not None else torch.zeros_like(Wg[k])) for k, g in zip(names, grads)}, pred.detach()


def adam_step(W, grads, m, v, step: int, lr: float):
    """keras.optimizers.Adam.  ``step`` counts from 1.  Returns the new (weights, m, v) for the trainable names."""
    lr_t = lr * (1 - ADAM_BETA2 ** step) ** 0.5 / (1 - ADAM_BETA1 ** step)
    Wn, mn, vn = {}, {}, {}
    for k, g in grads.items():
        mn[k] = ADAM_BETA1 * m[k] + (1 - ADAM_BETA1) * g
        vn[k] = ADAM_BETA2 * v[k] + (1 - ADAM_BETA2) * g * g
        Wn[k] = W[k] - lr_t * mn[k] / (torch.sqrt(vn[k]) + ADAM_EPS)
    return Wn, mn, vn

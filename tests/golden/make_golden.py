"""Generates tests/golden/*.npz from the CPU oracle (oracle/ref_torch.py), after checking it against the independent
NumPy/fp64 restatement (oracle/ref_numpy.py).  PARITY UNPINNED: the reference ships no golden data and TensorFlow is
not installed, so these vectors pin the *oracle* (and through it the HIP path) against regressions — they are not
outputs of the reference itself.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_numpy as rn  # noqa: E402
from oracle import ref_torch as rt  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def unet_case(conditional: bool, S: int, Cc: int, B: int, seed: int):
    cfg = rt.UNetConfig(img_size=S, img_channels=Cc, conditional=conditional)
    W = rt.synthetic_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, S, S, S, Cc, generator=g)
    t = torch.tensor([0, 1, 25, 49][:B] if B <= 4 else list(range(B)))
    ctx = torch.tensor([[[i % 2]] for i in range(B)]) if conditional else None
    taps = {}
    eps = rt.unet_forward(W, cfg, x, t, ctx, taps=taps)
    eps64 = rt.unet_forward({k: v.double() for k, v in W.items()}, cfg, x.double(), t, ctx)
    return cfg, W, x, t, ctx, eps, eps64, taps


def main():
    out = {}
    # 1) Betas / TimeEmbedding tables
    for T in (50, 1000):
        b = rt.Betas(T)
        for n in b.NAMES:
            out[f"betas{T}.{n}"] = getattr(b, n).numpy()
    out["temb128.t"] = np.array([0, 1, 2, 10, 499, 999])
    out["temb128"] = rt.time_embedding(torch.tensor(out["temb128.t"]), 128).numpy()
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **out)

    # 2) full U-Net eps, real widths, 8^3 x 4ch, B=4 (t in {0,1,T/2,T-1}), conditional and unconditional
    for cond in (True, False):
        cfg, W, x, t, ctx, eps, eps64, taps = unet_case(cond, 8, 4, 4, seed=123)
        # the two restatements must agree before anything is frozen (exact embeddings at t=0; <=1e-4 elsewhere because
        # ref_numpy evaluates the sinusoid in float64)
        en = rn.unet_forward({k: v.numpy() for k, v in W.items()}, cfg, x.numpy(), t.numpy(),
                             None if ctx is None else ctx.numpy())
        d0 = np.abs(en[0] - eps64[0].numpy()).max() / np.abs(en[0]).max()
        dall = np.abs(en - eps64.numpy()).max() / np.abs(en).max()
        assert d0 < 1e-10 and dall < 1e-4, (d0, dall)
        name = "unet_cond_s8c4" if cond else "unet_uncond_s8c4"
        keep = {"x": x.numpy(), "t": t.numpy(), "eps": eps.numpy(), "eps64": eps64.numpy().astype(np.float64)}
        if cond:
            keep["ctx"] = ctx.numpy()
        for k in ("down0.res0", "down0.ds", "mid.attn", "up2.res0", "up1.us"):
            keep["tap." + k] = taps[k].numpy()[:1]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **keep)
        print(name, "restatements agree:", d0, dall, "fp32-vs-fp64:",
              float((eps.double() - eps64).abs().max() / eps64.abs().max()))

    # 3) a 5-step generate trajectory with injected noise (conditional, T=5)
    cfg = rt.UNetConfig(img_size=8, img_channels=4)
    W = rt.synthetic_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(77)
    T, shape = 5, (2, 8, 8, 8, 4)
    x_T = torch.randn(shape, generator=g)
    noises = torch.randn((T,) + shape, generator=g)
    traj = []
    final = rt.generate(W, cfg, rt.Betas(T), T, x_T, noises, context_value=1, trajectory=traj)
    np.savez_compressed(os.path.join(HERE, "generate_cond_s8c4_T5.npz"), x_T=x_T.numpy(), noises=noises.numpy(),
                        final=final.numpy(), step0=traj[0].numpy())
    print("generate fixture written", float(final.abs().max()))


if __name__ == "__main__":
    main()

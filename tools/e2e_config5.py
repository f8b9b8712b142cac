"""BASELINE config 5, end to end on one GPU: encode 128^3 volumes -> 32^3 x 8ch latents, a full T=1000 conditional DDPM chain at
32^3 x 8ch, decode back to 128^3; B=8.  The reference never wires an autoencoder with a 32^3 latent to its DiffusionModel
(SURVEY.md §0.5, Appendix E): as there, the three stages are timed separately and generate() starts from N(0,1).
The autoencoder is networks/vqgan.py's (the one config 5 names; 2 levels as main_exp_vqgan.py builds it, input concat[img, mask]),
synthetic weights.

usage: python tools/e2e_config5.py [B] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np
import torch
import dm3d_amd
from dm3d_amd.networks import conditional_dm3d as cdm
from dm3d_amd.networks.vqgan import VQGAN, vqgan_param_spec

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


rng = np.random.default_rng(0)
spec = vqgan_param_spec(2, 2, (32, 64), 2, (32, 64), 1024, 8, 128)
W = {}
for k, shape in spec.items():
    if k.endswith(".kernel"):
        rf = int(np.prod(shape[:3])); lim = (6.0 / (shape[3] * rf + shape[4] * rf)) ** 0.5
        W[k] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
    elif k.endswith(".alpha"):
        W[k] = rng.uniform(0.05, 0.45, size=shape).astype(np.float32)
    elif k.endswith(".embeddings"):
        W[k] = rng.normal(size=shape).astype(np.float32)
    elif k.endswith((".gamma", ".var")):
        W[k] = rng.uniform(0.8, 1.2, size=shape).astype(np.float32)
    else:
        W[k] = rng.normal(0, 0.05, size=shape).astype(np.float32)
vq = VQGAN(2, 2, (32, 64), 2, (32, 64), downsample_parameters=[(2, 4, 1, "same")] * 2, upsample_parameters=[(2, 4, 1, "same", 0)] * 2,
           num_embeddings=1024, embedding_dim=8, D=128, weights=W)
cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
model = cdm.DiffusionModel(32, 1024, 8, None, SimpleNamespace(timesteps=T, num_gpus=1, kernel_resize=False, bs=B),
                           weights=dm3d_amd.synthetic_weights(cfg, 0))
x = torch.rand(B, 128, 128, 128, 2, device=dev)            # concat[img, mask]
vq.prepare(); model.network.prepare()
vq.decoder(vq.quantizer(vq.encoder(x))[0])                                  # warm (first-launch costs)
model.generate((B, 32, 32, 32, 8), context_value=1, steps=2)
(lat, perp), t_enc = timed(lambda: vq.quantizer(vq.encoder(x)))
gen, t_gen = timed(lambda: model.generate((B, 32, 32, 32, 8), context_value=1, seed=1234))
img, t_dec = timed(lambda: vq.decoder(gen))
assert tuple(lat.shape) == (B, 32, 32, 32, 8) and tuple(img.shape) == (B, 128, 128, 128, 2) and torch.isfinite(img).all()
tot = t_enc + t_gen + t_dec
print(f"config5 B={B} T={T}: encode+quantize {t_enc * 1e3:.1f} ms | generate {t_gen:.2f} s ({t_gen / T * 1e3:.2f} ms/step) | "
      f"decode {t_dec * 1e3:.1f} ms | total {tot:.2f} s = {B / tot:.3f} volumes/s (perplexity {float(perp):.1f})")

"""Times DiffusionModel.train_step on the reference's training configuration (main_conditional_dm.py:141-147: latent 8^3 x 256ch from 128^3
images, T = 500 as in sb_cond_dm3d.sbatch) and on the sampling benchmark's shape (32^3 x 8ch).
usage: python tools/train_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import dm3d_amd
from dm3d_amd.networks import conditional_dm3d as cdm

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
for S, C, B, with_images in ((8, 256, 8, True), (8, 256, 8, False), (32, 8, 4, False)):
    cfg = dm3d_amd.UNetConfig(img_size=S, img_channels=C)
    m = cdm.DiffusionModel(S, 1024, C, None, SimpleNamespace(timesteps=500, num_gpus=1, kernel_resize=False, bs=B),
                           weights=dm3d_amd.synthetic_weights(cfg, 0))
    m.compile(loss=None, optimizer=1e-4)
    g = torch.Generator().manual_seed(0)
    ctx = torch.randint(0, 2, (B, 1, 1), generator=g)
    images = torch.rand(B, 16 * S, 16 * S, 16 * S, 1, generator=g).to(dev) if with_images else None
    lat = None if with_images else torch.randn(B, S, S, S, C, generator=g).to(dev)
    for i in range(steps + 2):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = m.train_step((images, None, ctx), latents=lat)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"train_step latent {S}^3x{C} B={B} {'from 128^3 images (frozen encoder + quantizer)' if with_images else 'pre-encoded latents'}: "
          f"{dt * 1e3:.1f} ms/step = {B / dt:.1f} volumes/s, loss {out['loss']:.4g}", flush=True)
    del m
    torch.cuda.empty_cache()

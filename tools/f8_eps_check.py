"""eps of the U-Net in precision h3 and h3f8 against the exact-float32 mode (v_mfma_f32_32x32x2_f32), same weights and inputs.
usage: python tools/f8_eps_check.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dm3d_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
W = dm3d_amd.synthetic_weights(cfg, 0)
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 32, 32, 32, 8, generator=g).to(dev)
t = torch.randint(0, 1000, (B,), generator=g).to(dev)
ctx = torch.ones(1, 1, 1, dtype=torch.int64, device=dev)
outs = {}
for prec in ("fp32", "h3", "h3f8"):
    net = dm3d_amd.UNet(cfg, device=dev, weights=W, precision=prec)
    outs[prec] = net([x, t, ctx]).double().cpu()
    print(prec, {k: v for k, v in net.plan(B, B).count().items() if k.startswith("conv")})
    del net
ref = outs["fp32"]
for prec in ("h3", "h3f8"):
    d = (outs[prec] - ref).abs()
    print(f"{prec}: max|d|/max|ref| = {float(d.max() / ref.abs().max()):.3e}, rms rel = {float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}")

"""Per-launch timing of one U-Net forward (HIP events around every launch).  usage: python tools/layer_profile.py [h3|fp32] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dm3d_amd
from dm3d_amd.unet import UNet

prec = sys.argv[1] if len(sys.argv) > 1 else "h3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
net = UNet(cfg, weights=dm3d_amd.synthetic_weights(cfg, 0), precision=prec)
plan = net.plan(B, 1000, False)
net.fill_time_table(list(range(1000)), plan.vec)
plan.set_context([1])
plan.x.normal_()
plan.t_idx.fill_(500)
plan.run_timed()
rows = plan.run_timed()
r2 = plan.run_timed()
tot = 0.0
for (kind, meta, ms), (_, _, ms2) in zip(rows, r2):
    ms = min(ms, ms2)
    tot += ms
    fl = meta.get("flops", 0)
    print(f"{ms:8.3f} ms  {fl / ms / 1e9 if fl else 0:8.1f} TF  {meta.get('desc', kind)}")
print(f"total {tot:.2f} ms")

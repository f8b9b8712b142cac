"""Per-launch times of the conv launches of one U-Net step (B = LIST_BATCH, default 32): kind, shape, ms, algorithmic TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dm3d_amd
dev = torch.device("cuda:0")
B = int(os.environ.get("LIST_BATCH", "32"))
cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8, conditional=True)
net = dm3d_amd.UNet(cfg, device=dev, weights=dm3d_amd.synthetic_weights(cfg, seed=0), precision="h3")
x = torch.randn(B, 32, 32, 32, 8, device=dev)
t = torch.randint(0, 1000, (B,), device=dev, dtype=torch.int32)
ctx = torch.zeros(B, device=dev, dtype=torch.int32)
net([x, t, ctx])
plan = next(iter(net._plans.values()))
plan.run_timed()
acc = {}
for _ in range(3):
    for i, (kind, meta, ms) in enumerate(plan.run_timed()):
        acc.setdefault(i, [kind, meta, []])[2].append(ms)
tot = 0.0
for i, (kind, meta, ms) in sorted(acc.items()):
    m = sorted(ms)[1]
    tot += m
    if kind.startswith("conv"):
        print(f"{i:3d} {kind:16s} {meta.get('desc', ''):60s} {m:.4f} ms {meta.get('flops', 0) / m / 1e9:6.0f} TF")
print(f"sum of launches {tot:.3f} ms")

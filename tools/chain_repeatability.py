"""Two DDPM chains with the same seed must agree bit for bit (B=32, 32^3x8ch), step by step.  usage: python tools/chain_repeatability.py [graph|eager] [steps]
Found the memset-node race of the first split-K version (HIP-graph replays diverged after ~90 steps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch, dm3d_amd
from dm3d_amd.networks import conditional_dm3d as cdm
NORM = os.environ.get("DM3D_NORM", "batch")
cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8, norm=NORM)
m = cdm.DiffusionModel(32, 1024, 8, None, SimpleNamespace(timesteps=1000, num_gpus=1, kernel_resize=False, bs=32),
                       weights=dm3d_amd.synthetic_weights(cfg, seed=0), norm=NORM,
                       precision=os.environ.get("DM3D_PRECISION", "h3"))
B = 32
use_graph = (sys.argv[1] == "graph") if len(sys.argv) > 1 else True
every = 10
def chain(n):
    smp = m.sampler((B, 32, 32, 32, 8), context_value=1, seed=7, use_graph=use_graph)
    smp.reset()
    snaps = []
    for k in range(n):
        smp.step()
        if (k + 1) % every == 0:
            snaps.append(smp.plan.x.clone())
    torch.cuda.synchronize()
    return snaps
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
a = chain(n); b = chain(n)
for i, (u, v) in enumerate(zip(a, b)):
    if not torch.equal(u, v):
        d = (u != v)
        print(f"graph={use_graph}: first divergence at checkpoint after step {(i + 1) * every}: n diff {int(d.sum())} of {d.numel()}, maxdiff {float((u - v).abs().max()):.3e}")
        idx = d.nonzero()
        print("  samples affected:", sorted(set(idx[:, 0].tolist()))[:10], " first idx:", idx[0].tolist())
        break
else:
    print(f"graph={use_graph}: {n} steps identical")

"""One U-Net forward of the B = 32 plan, launch by launch, twice from the same state: the buffers the attention-block launches name in their
meta ("bufs") must hash the same in both runs — finds the first launch whose output is not reproducible.  usage: python tools/plan_repeat.py [runs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch, dm3d_amd
from dm3d_amd.networks import conditional_dm3d as cdm
from dm3d_amd.unet import _stream
cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
m = cdm.DiffusionModel(32, 1024, 8, None, SimpleNamespace(timesteps=1000, num_gpus=1, kernel_resize=False, bs=32), weights=dm3d_amd.synthetic_weights(cfg, seed=0))
smp = m.sampler((32, 32, 32, 32, 8), context_value=1, seed=7, use_graph=False)
smp.reset()
plan = smp.plan
x0 = plan.x.clone()
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
h = lambda t: int(t.view(torch.int32).to(torch.int64).sum().item())
ref = None
for r in range(runs):
    plan.x.copy_(x0)
    torch.cuda.synchronize()
    sig = []
    st = _stream()
    for i, (fn, args, what, meta) in enumerate(plan.ops):
        rc = fn(*args, st)
        assert rc == 0, what
        if "bufs" in meta:
            if os.environ.get("SYNC", "0") == "1": torch.cuda.synchronize()
            sig.append((i, what, {k: v.clone() for k, v in meta["bufs"].items()}))          # (stream-ordered copies: the launches stay back to back)
    sig.append((len(plan.ops), "eps", {"eps": plan.eps.clone()}))
    torch.cuda.synchronize()
    sig = [(i, what, {k: h(v) for k, v in d.items()}) for i, what, d in sig]
    if ref is None:
        ref = sig
        continue
    for (i, what, a), (_, _, b) in zip(ref, sig):
        bad = [k for k in a if a[k] != b[k]]
        if bad:
            print(f"run {r}: launch {i} ({what}): {bad} differ from run 0")
            break
    else:
        print(f"run {r}: identical to run 0 ({len(sig)} checkpoints)")

"""Two eager DDPM chains of K steps from one seed on ONE plan, with stream-ordered copies of the buffers the attention-block launches name
("bufs" in their meta) at every step: prints the first (step, launch, buffer) that differs between the chains.  usage: python tools/chain_trace.py [steps]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch, dm3d_amd
from dm3d_amd.networks import conditional_dm3d as cdm
from dm3d_amd._lib import lib, check
cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
m = cdm.DiffusionModel(32, 1024, 8, None, SimpleNamespace(timesteps=1000, num_gpus=1, kernel_resize=False, bs=32), weights=dm3d_amd.synthetic_weights(cfg, seed=0))
smp = m.sampler((32, 32, 32, 32, 8), context_value=1, seed=7, use_graph=False)
plan = smp.plan
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
st = torch.cuda.current_stream().cuda_stream
def chain():
    smp.reset()
    trace = []
    for k in range(K):
        for i, (fn, args, what, meta) in enumerate(plan.ops):
            assert fn(*args, st) == 0, what
            if "bufs" in meta:
                trace.append((k, i, what, {n: v.clone() for n, v in meta["bufs"].items()}))
        trace.append((k, len(plan.ops), "eps", {"eps": plan.eps.clone()}))
        check(lib().dm3d_ddpm_update(C.byref(smp.desc), st), "ddpm_update")
        check(lib().dm3d_add_i32(plan.t_idx.data_ptr(), plan.B, -1, st), "add_i32")
        trace.append((k, len(plan.ops) + 1, "x", {"x": plan.x.clone()}))
    torch.cuda.synchronize()
    return trace
a = chain()
b = chain()
for (k, i, what, da), (_, _, _, db) in zip(a, b):
    bad = [n for n in da if not torch.equal(da[n].view(torch.int32), db[n].view(torch.int32))]
    if bad:
        print(f"first difference: step {k}, launch {i} ({what}), buffers {bad}")
        for n in bad:
            d = (da[n].view(torch.int32) != db[n].view(torch.int32)).reshape(da[n].shape[0], -1) if da[n].dim() > 1 else None
            u, v = da[n].reshape(-1), db[n].reshape(-1)
            idx = (u.view(torch.int32) != v.view(torch.int32)).nonzero().reshape(-1)
            print(f"  {n}: shape {tuple(da[n].shape)}, {idx.numel()} words differ, flat indices {idx[:12].tolist()} ... {idx[-3:].tolist()}")
        break
else:
    print(f"{K} steps: every traced buffer identical")

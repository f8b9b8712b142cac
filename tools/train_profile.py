"""Where a train_step spends its time: wraps the Trainer's launch helpers with HIP events (synchronising: diagnosis only).
usage: python tools/train_profile.py [S] [C] [B]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import dm3d_amd
from dm3d_amd import train as T
from dm3d_amd.networks import conditional_dm3d as cdm

S, C, B = (int(a) for a in (sys.argv[1:4] + ["32", "8", "4"][len(sys.argv) - 1:]))
acc = collections.defaultdict(lambda: [0, 0.0])


def wrap(name, keyfn):
    orig = getattr(T.Trainer, name)

    def f(self, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(self, *a, **k)
        e1.record()
        torch.cuda.synchronize()
        rec = acc[(name, keyfn(*a, **k))]
        rec[0] += 1
        rec[1] += e0.elapsed_time(e1)
        return r
    setattr(T.Trainer, name, f)


wrap("_wgrad", lambda a, g, dw, cin, cout, ksize, batch, d, h, w, **k: f"cin={cin} cout={cout} k={ksize} rows={batch * d * h * w} per_item={k.get('per_item', False)}")
wrap("_gemm", lambda a, lda, b, ldb, m, n, k, **kw: f"m={m} n={n} k={k} batch={kw.get('batch', 1)}")
wrap("_conv_launch", lambda x, wpk, cin, cout, ksize, stride=1, **k: f"{tuple(x.shape[1:4])} cin={cin} cout={cout} k={ksize} s={stride}")
cfg = dm3d_amd.UNetConfig(img_size=S, img_channels=C)
m = cdm.DiffusionModel(S, 1024, C, None, SimpleNamespace(timesteps=500, num_gpus=1, kernel_resize=False, bs=B), weights=dm3d_amd.synthetic_weights(cfg, 0))
g = torch.Generator().manual_seed(0)
lat = torch.randn(B, S, S, S, C, generator=g).cuda()
ctx = torch.randint(0, 2, (B, 1, 1), generator=g)
m.train_step((None, None, ctx), latents=lat)
acc.clear()
m.train_step((None, None, ctx), latents=lat)
tot = sum(v[1] for v in acc.values())
print(f"instrumented launches: {tot:.1f} ms")
for (name, key), (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{ms:9.2f} ms  x{n:<3d} {name:13s} {key}")

"""Times dm3d_attn_front (and dm3d_mlp_fused with its proj_out tail) on the attention block's shape (m = B * 512 rows, u = 256).
usage: python tools/front_time.py [tag]     (DM3D_LIB=<variant> to time another build: tools/lib_ab.sh)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib
dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("DM3D_LIB", "product"))
u, B = 256, int(os.environ.get("AB_BATCH", "32"))
m = B * 512
g = torch.Generator().manual_seed(0)
c = lambda t: t.to(dev).contiguous()
x = c(torch.randn(m, u, generator=g))
tile = lambda w: ops.pack_front_weights(ops.split_h2(c(w)), w.shape[0])
w_in, w_qk, w_v = tile(torch.randn(u, u, generator=g) / 16), tile(torch.randn(2 * u, u, generator=g) / 16), tile(torch.randn(u, u, generator=g) / 16)
b_in, b_qk, b_v = c(torch.randn(u, generator=g) * 0.1), c(torch.randn(2 * u, generator=g) * 0.1), c(torch.randn(u, generator=g) * 0.1)
norms = [(c(torch.rand(u, generator=g) + 0.5), c(torch.randn(u, generator=g) * 0.2)) for _ in range(3)]
xh = ops.split_h2(x)
w0t, w1t = ops.pack_mlp_weights(ops.split_h2(c(torch.randn(4 * u, u, generator=g) / 16)), u, 0), ops.pack_mlp_weights(ops.split_h2(c(torch.randn(u, 4 * u, generator=g) / 32)), u, 1)
b0, b1 = c(torch.randn(4 * u, generator=g)), c(torch.randn(u, generator=g))
r1, r2 = c(torch.randn(m, u, generator=g)), c(torch.randn(m, u, generator=g))
def front(): return ops.attn_front(x, w_in, b_in, w_qk, b_qk, w_v, b_v, norms)
def back(): return ops.mlp_fused(xh, w0t, b0, w1t, b1, u, res=r1, res2=r2, tail=(w_in, b_in, x))
for name, fn, fl in (("attn_front", front, 2.0 * m * u * 5 * u), ("mlp + tail", back, 2.0 * m * u * 9 * u)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = statistics.median(ts)
    print(f"[{tag}] {name:10s} {t * 1e3:7.1f} us  {fl / t / 1e9:6.0f} TF", flush=True)

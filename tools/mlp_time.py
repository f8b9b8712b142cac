"""Times dm3d_mlp_fused against the two-GEMM form on the attention block's shape (m = B * 512 rows, u = 256).  usage: python tools/mlp_time.py [tag]"""
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
x = ops.split_h2(c(torch.randn(m, u, generator=g)))
w0, w1 = ops.split_h2(c(torch.randn(4 * u, u, generator=g) / 16)), ops.split_h2(c(torch.randn(u, 4 * u, generator=g) / 32))
b0, b1 = c(torch.randn(4 * u, generator=g)), c(torch.randn(u, generator=g))
r1, r2 = c(torch.randn(m, u, generator=g)), c(torch.randn(m, u, generator=g))
w0t, w1t = ops.pack_mlp_weights(w0, u, 0), ops.pack_mlp_weights(w1, u, 1)
def fused(): return ops.mlp_fused(x, w0t, b0, w1t, b1, u, res=r1, res2=r2, out_h2=True)
hid = torch.empty(m, 4 * u, device=dev)
def two():
    h = ops.gemm_tn(x, w0, m=m, n=4 * u, k=u, lda=u, ldb=u, bias=b0, act=_lib.ACT_RELU, precision=_lib.PREC_H3, a_fmt=_lib.FMT_H2, b_fmt=_lib.FMT_H2, out_fmt=_lib.FMT_H2)
    return ops.gemm_tn(h, w1, m=m, n=u, k=4 * u, lda=4 * u, ldb=4 * u, bias=b1, res=r1, precision=_lib.PREC_H3, a_fmt=_lib.FMT_H2, b_fmt=_lib.FMT_H2, out_fmt=_lib.FMT_H2)
for name, fn in (("fused", fused), ("two gemms", two)):
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
    print(f"[{tag}] {name:10s} {t * 1e3:7.1f} us  {2 * 2.0 * m * u * 4 * u / t / 1e9:6.0f} TF", flush=True)

if len(sys.argv) > 2 and sys.argv[2] == "stamps":
    import ctypes as C
    raw = C.CDLL(_lib.LIB_PATH)
    st = torch.zeros(256 * 32, dtype=torch.int64, device=dev)
    raw.dm3d_debug_set_stamps_mlp(C.c_void_p(st.data_ptr()))
    for _ in range(5): fused()
    torch.cuda.synchronize()
    v = st.view(256, 32).cpu().double()
    names = {0: "entry", 1: "x tile loaded", 2: "slab 1 start", 3: "p1 g0", 4: "p1 g1", 5: "p1 g2", 6: "p1 g3", 7: "p1 MFMAs done", 8: "H stored", 9: "p2 g0", 10: "p2 g1",
             11: "p2 g2", 12: "p2 g3", 13: "slab 2 start", 14: "loop done", 15: "end"}
    med = v.median(0).values
    prev = None
    for i in sorted(names):
        d = "" if prev is None else f"  (+{med[i] - med[prev]:.0f})"
        print(f"  stamp {i:2d} {names[i]:16s} {med[i] - med[0]:9.0f}{d}")
        prev = i
    raw.dm3d_debug_set_stamps_mlp(C.c_void_p(0))

"""In-kernel timeline of gemm_tn_h3 (needs the stamp variant library: DM3D_LIB=.../variants/gst.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib
dev = torch.device("cuda:0")
H3, F32, H2 = _lib.PREC_H3, _lib.FMT_F32, _lib.FMT_H2
lib = _lib.lib()
raw = C.CDLL(_lib.LIB_PATH)
st = torch.zeros(2048 * 16, dtype=torch.int64, device=dev)
for name, m, n, k, ofmt, res in (("proj", 16384, 256, 256, F32, 1), ("mlp0", 16384, 1024, 256, H2, 0), ("mlp1", 16384, 256, 1024, F32, 1)):
    a = ops.split_h2(torch.randn(m, k, device=dev)); b = ops.split_h2(torch.randn(n, k, device=dev) * 0.05)
    r = torch.randn(m, n, device=dev) if res else None
    out = torch.empty(m, n, device=dev)
    kw = dict(m=m, n=n, k=k, lda=k, ldb=k, bias=torch.randn(n, device=dev), res=r, out=out, precision=H3, a_fmt=H2, b_fmt=H2, out_fmt=ofmt)
    for _ in range(3): ops.gemm_tn(a, b, **kw)
    torch.cuda.synchronize()
    raw.dm3d_debug_set_stamps(C.c_void_p(st.data_ptr()))
    st.zero_()
    ops.gemm_tn(a, b, **kw)
    torch.cuda.synchronize()
    raw.dm3d_debug_set_stamps(C.c_void_p(0))
    s = st.view(2048, 16).cpu()
    nwg = (m // 128) * ((n + 127) // 128)
    s = s[:min(nwg, 2048)]
    t0 = s[:, 0].min()
    print(name, "WGs", nwg, "kernel span (cycles of the 100 MHz counter?)", int(s[:, 11].max() - t0))
    d = (s[:, 1:12] - s[:, 0:11]).double()
    print("  mean deltas start->fetch, ->chunk0 ... ->epilogue, epilogue:", [int(x) for x in d.mean(0)])
    print("  WG duration mean/min/max:", int((s[:, 11] - s[:, 0]).double().mean()), int((s[:, 11] - s[:, 0]).min()), int((s[:, 11] - s[:, 0]).max()))
    print("  WG start offsets (first 8, last 8):", [int(x - t0) for x in s[:8, 0]], [int(x - t0) for x in s[-8:, 0]])

"""Does the row stride of the GEMM operands matter (L2 channel camping)?  The K = 256 / 1024 GEMMs of the attention block read 128-byte
pieces of rows that lie 1 KB / 4 KB apart; here the same problems with the rows padded by 16 / 32 / 64 elements.
usage: python tools/gemm_stride_exp.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib
from dm3d_amd._lib import lib, check
dev = torch.device("cuda:0")
H3, F32, H2 = _lib.PREC_H3, _lib.FMT_F32, _lib.FMT_H2
st = torch.cuda.current_stream().cuda_stream


def split_padded(src, ld):
    rows, k = src.shape
    dst = torch.zeros(rows, ld, dtype=torch.float32, device=dev)          # H2 rows: 4 bytes per element, like float32
    check(lib().dm3d_split_h2(src.data_ptr(), rows, k, k, 0, dst.data_ptr(), ld, st), "split_h2")
    return dst


for name, m, n, k, ofmt, res in (("proj", 16384, 256, 256, F32, 1), ("mlp0", 16384, 1024, 256, H2, 0), ("mlp1", 16384, 256, 1024, F32, 1)):
    A = torch.randn(m, k, device=dev); Bw = torch.randn(n, k, device=dev) * 0.05
    r = torch.randn(m, n, device=dev) if res else None
    ref = None
    for pad in (0, 16, 32, 64):
        a, b = split_padded(A, k + pad), split_padded(Bw, k + pad)
        out = torch.empty(m, n, device=dev)
        kw = dict(m=m, n=n, k=k, lda=k + pad, ldb=k + pad, batch=1, stride_a=0, stride_b=0, bias=torch.zeros(n, device=dev), act=_lib.ACT_NONE, res=r,
                  out=out, precision=H3, a_fmt=H2, b_fmt=H2, out_fmt=ofmt)
        ops.gemm_tn(a, b, **kw); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm_tn(a, b, **kw)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        if ref is None: ref = out.clone()
        print(f"{name} m={m} n={n} k={k} row pad {pad:3d}: {best * 1e3:7.1f} us  {2.0 * m * n * k / best / 1e9:6.1f} TF  maxdiff {float((out - ref).abs().max()):.1e}", flush=True)

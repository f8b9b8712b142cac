"""Micro-benchmark of dm3d_gemm_tn (H3, pre-split operands) on the attention-block shapes of the 32^3 U-Net at B=32.
usage: python tools/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib

dev = torch.device("cuda:0")
H3, F32, H2 = _lib.PREC_H3, _lib.FMT_F32, _lib.FMT_H2
CASES = [  # name, batch, m, n, k, out_fmt, res, act
    ("proj 16384x256x256 +res", 1, 16384, 256, 256, F32, 1, 0),
    ("qk   16384x512x256 ->h2", 1, 16384, 512, 256, H2, 0, 0),
    ("mlp0 16384x1024x256 relu->h2", 1, 16384, 1024, 256, H2, 0, 1),
    ("mlp1 16384x256x1024 +res", 1, 16384, 256, 1024, F32, 1, 0),
    ("scores b32 512x512x256", 32, 512, 512, 256, F32, 0, 0),
    ("pv     b32 512x256x512 +res", 32, 512, 256, 512, F32, 1, 0),
]
for name, bt, m, n, k, ofmt, res, act in CASES:
    a = ops.split_h2(torch.randn(bt * m, k, device=dev))
    b = ops.split_h2(torch.randn((bt if bt > 1 else 1) * n, k, device=dev) * 0.05)
    r = torch.randn(bt * m, n, device=dev) if res else None
    out = torch.empty(bt * m, n, device=dev)
    kw = dict(m=m, n=n, k=k, lda=k, ldb=k, batch=bt, stride_a=m * k, stride_b=n * k if bt > 1 else 0, bias=torch.randn(n, device=dev),
              act=_lib.ACT_RELU if act else _lib.ACT_NONE, res=r, out=out, precision=H3, a_fmt=H2, b_fmt=H2, out_fmt=ofmt)
    ops.gemm_tn(a, b, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm_tn(a, b, **kw)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    fl = 2.0 * bt * m * n * k
    print(f"{name:32s} {best * 1e3:8.1f} us {fl / best / 1e9:8.1f} TF")

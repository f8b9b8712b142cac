"""Interleaved A/B of a launch-time GEMM knob (environment variable read per call) on the attention-block shapes of the 32^3 U-Net, one
process, one device; also checks that the two forms give identical results.  usage: python tools/gemm_ab.py DM3D_GEMM_PC 0 1 [r=rounds]
(AB_BATCH=n scales the row count; AB_F32=1 feeds float32 operands, split in the kernel)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib
var, vals = sys.argv[1], sys.argv[2:]
rounds = 5
if vals and vals[-1].startswith("r="): rounds = int(vals.pop()[2:])
dev = torch.device("cuda:0"); B = int(os.environ.get("AB_BATCH", "32")); f32 = os.environ.get("AB_F32") == "1"
H3, F32, H2 = _lib.PREC_H3, _lib.FMT_F32, _lib.FMT_H2
R = B * 512
CASES = [("proj %dx256x256 +res" % R, 1, R, 256, 256, F32, 1, 0), ("qk %dx512x256 ->h2" % R, 1, R, 512, 256, H2, 0, 0),
         ("mlp0 %dx1024x256 relu->h2" % R, 1, R, 1024, 256, H2, 0, 1), ("mlp1 %dx256x1024 +res" % R, 1, R, 256, 1024, F32, 1, 0),
         ("scores b%d 512x512x256" % B, B, 512, 512, 256, F32, 0, 0), ("pv b%d 512x256x512 +res" % B, B, 512, 256, 512, F32, 1, 0),
         ("ragged 1000x200x80 +res", 1, 1000, 200, 80, F32, 1, 0)]
for name, bt, m, n, k, ofmt, res, act in CASES:
    a = torch.randn(bt * m, k, device=dev); b = torch.randn((bt if bt > 1 else 1) * n, k, device=dev) * 0.05
    if not f32: a, b = ops.split_h2(a), ops.split_h2(b)
    r = torch.randn(bt * m, n, device=dev) if res else None
    kw = dict(m=m, n=n, k=k, lda=k, ldb=k, batch=bt, stride_a=m * k, stride_b=n * k if bt > 1 else 0, bias=torch.randn(n, device=dev),
              act=_lib.ACT_RELU if act else _lib.ACT_NONE, res=r, precision=H3, a_fmt=F32 if f32 else H2, b_fmt=F32 if f32 else H2, out_fmt=ofmt)
    outs, times = {}, {v: [] for v in vals}
    for v in vals:
        os.environ[var] = v
        outs[v] = torch.zeros(bt * m, n, device=dev)
        ops.gemm_tn(a, b, out=outs[v], **kw)
    torch.cuda.synchronize()
    scratch = torch.empty(bt * m, n, device=dev)
    for _ in range(rounds):
        for v in vals:
            os.environ[var] = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): ops.gemm_tn(a, b, out=scratch, **kw)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 8)
    fl = 2.0 * bt * m * n * k
    ref = outs[vals[0]]
    print(f"{name:30s} " + "  ".join(f"{var}={v}: med {statistics.median(times[v]) * 1e3:.1f} min {min(times[v]) * 1e3:.1f} us ({fl / statistics.median(times[v]) / 1e9:.0f} TF) "
          f"identical {bool(torch.equal(outs[v].view(torch.int32), ref.view(torch.int32)))}" for v in vals), flush=True)

"""Interleaved A/B of launch-time knobs (environment variables the library reads per call) on k3 conv shapes, one process, one device
(cdna_hip_programming.md rule 24).  usage: python tools/env_ab.py DM3D_CONV_V3_TD 4 8 [r=rounds]   (AB_BATCH=n, AB_SMALL=1 select the shapes)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib
var, vals = sys.argv[1], sys.argv[2:]
rounds = 5
if vals and vals[-1].startswith("r="): rounds = int(vals.pop()[2:])
dev = torch.device("cuda:0"); B = int(os.environ.get("AB_BATCH", "32"))
CASES = [("8^3 256->256 pro+res", 8, 256, 256, 1), ("8^3 512->256 pro", 8, 512, 256, 0), ("16^3 128->128 pro", 16, 128, 128, 0)] if os.environ.get("AB_SMALL") == "1" else [("32^3 64->64 pro+res", 32, 64, 64, 1), ("32^3 96->64 pro", 32, 96, 64, 0), ("32^3 192->64 pro", 32, 192, 64, 0),
         ("16^3 128->128 pro+res", 16, 128, 128, 1), ("16^3 384->128 pro", 16, 384, 128, 0)]
for name, e, cin, cout, res in CASES:
    x = torch.randn(B, e, e, e, cin, device=dev)
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
    wpk, w_exp = ops.pack_weights_h3(k)
    kw = dict(bias=torch.randn(cout, device=dev), pro_scale=torch.rand(cin, device=dev) + 0.5, pro_shift=torch.randn(cin, device=dev) * 0.1,
              res=torch.randn(B, e, e, e, cout, device=dev) if res else None, precision=_lib.PREC_H3, w_exp=w_exp)
    outs, times = {}, {v: [] for v in vals}
    for v in vals:
        os.environ[var] = v
        outs[v] = ops.conv3d(x, wpk, cout, 3, **kw).clone()
    torch.cuda.synchronize()
    for r in range(rounds):
        for v in vals:
            os.environ[var] = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): ops.conv3d(x, wpk, cout, 3, **kw)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 8)
    fl = 2.0 * 27 * cin * cout * B * e ** 3
    ref = outs[vals[0]]
    print(f"{name:24s} " + "  ".join(f"{var}={v}: med {statistics.median(times[v]):.4f} min {min(times[v]):.4f} ms ({fl / statistics.median(times[v]) / 1e9:.0f} TF) maxdiff {float((outs[v] - ref).abs().max()):.1e}" for v in vals), flush=True)

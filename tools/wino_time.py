"""Times the Winograd-x conv (dm3d_conv_h3w.hip) on the U-Net's k3 shapes at B = AB_BATCH (32): median of 7 rounds of 5 launches per shape.
Run it under different DM3D_LIB builds back to back on one box to compare them (tools/lib_ab.sh).  usage: python tools/wino_time.py [tag]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("DM3D_LIB", "product"))
B = int(os.environ.get("AB_BATCH", "32"))
os.environ["DM3D_CONV_WINO_MINCHUNKS"] = "1"
SHAPES = [("32^3 32->64 pro", 32, 32, 64, 0, 0), ("32^3 64->64 pro+res", 32, 64, 64, 1, 0), ("32^3 64->64 h2in+res", 32, 64, 64, 1, 1), ("32^3 96->64 pro", 32, 96, 64, 0, 0),
          ("32^3 192->64 pro", 32, 192, 64, 0, 0), ("16^3 128->128 pro+res", 16, 128, 128, 1, 0), ("16^3 384->128 pro", 16, 384, 128, 0, 0),
          ("8^3 256->256 pro", 8, 256, 256, 0, 0), ("8^3 512->256 pro", 8, 512, 256, 0, 0),
          # ResidualBlock conv1 as the U-Net plan runs it: + time-embedding vector, the consumer's norm + SiLU in the epilogue, DM3D_FMT_H2 output
          ("32^3 64->64 conv1", 32, 64, 64, 0, 2), ("32^3 192->64 conv1", 32, 192, 64, 0, 2), ("16^3 128->128 conv1", 16, 128, 128, 0, 2)]
cases = []
for name, e, cin, cout, res, h2 in SHAPES:
    x = torch.randn(B, e, e, e, cin, device=dev)
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
    wpk, w_exp = ops.pack_weights_h3(k)
    kw = dict(bias=torch.randn(cout, device=dev), res=torch.randn(B, e, e, e, cout, device=dev) if res else None, precision=_lib.PREC_H3, w_exp=w_exp,
              wpk_wino=ops.pack_weights_h3w(k, w_exp))
    if h2 == 2:
        kw.update(vec=torch.randn(B, cout, device=dev), post=(torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1), out_h2=True,
                  pro_scale=torch.rand(cin, device=dev) + 0.5, pro_shift=torch.randn(cin, device=dev) * 0.1)
    elif h2:
        x = ops.split_h2(x.reshape(-1, cin)).reshape(B, e, e, e, cin)
        kw["x1_h2_channels"] = cin
    else:
        kw.update(pro_scale=torch.rand(cin, device=dev) + 0.5, pro_shift=torch.randn(cin, device=dev) * 0.1)
    cases.append((name, x, wpk, cout, kw, 2.0 * 27 * cin * cout * B * e ** 3))
for name, x, wpk, cout, kw, fl in cases:
    for _ in range(10): ops.conv3d(x, wpk, cout, 3, **kw)
torch.cuda.synchronize()
tot = 0.0
for name, x, wpk, cout, kw, fl in cases:
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.conv3d(x, wpk, cout, 3, **kw)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    m = statistics.median(ts); tot += m
    print(f"[{tag}] {name:24s} {m:.4f} ms  {fl / m / 1e9:5.0f} TF", flush=True)
print(f"[{tag}] sum {tot:.4f} ms", flush=True)

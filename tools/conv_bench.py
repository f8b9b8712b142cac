"""Micro-benchmark of dm3d_conv3d_ndhwc on the layer shapes of the 32^3 U-Net.  usage: python tools/conv_bench.py [h3|fp32] [B] [only-k3s1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib

prec = sys.argv[1] if len(sys.argv) > 1 else "h3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
CASES = [  # name, edge, c1, c2, cout, ks, stride, ups, pro, res, vec
    ("32^3 64->64 plain", 32, 64, 0, 64, 3, 1, 0, 0, 0, 0),
    ("32^3 64->64 pro", 32, 64, 0, 64, 3, 1, 0, 1, 0, 0),
    ("32^3 64->64 pro+res", 32, 64, 0, 64, 3, 1, 0, 1, 1, 0),
    ("32^3 64->64 pro+vec", 32, 64, 0, 64, 3, 1, 0, 1, 0, 1),
    ("32^3 128+64->64 pro", 32, 128, 64, 64, 3, 1, 0, 1, 0, 1),
    ("16^3x2 128->128 up", 16, 128, 0, 128, 3, 1, 1, 0, 0, 0),
    ("8^3x2 256->256 up", 8, 256, 0, 256, 3, 1, 1, 0, 0, 0),
    ("16^3 128->128 pro+res", 16, 128, 0, 128, 3, 1, 0, 1, 1, 0),
    ("8^3 256->256 pro+res", 8, 256, 0, 256, 3, 1, 0, 1, 1, 0),
    ("32^3 64->8 pro (out)", 32, 64, 0, 8, 3, 1, 0, 1, 0, 0),
    ("32^3 8->32 (in)", 32, 8, 0, 32, 3, 1, 0, 0, 0, 0),
    ("32^3 64->64 s2", 32, 64, 0, 64, 3, 2, 0, 0, 0, 0),
    ("32^3 128+64->64 k1", 32, 128, 64, 64, 1, 1, 0, 0, 0, 0),
]
for name, e, c1, c2, cout, ks, stride, ups, pro, res, vec in CASES:
    x1 = torch.randn(B, e, e, e, c1, device=dev)
    x2 = torch.randn(B, e, e, e, c2, device=dev) if c2 else None
    k = torch.randn(ks, ks, ks, c1 + c2, cout, device=dev) * 0.05
    if len(sys.argv) > 3 and (ks != 3 or stride != 1 or cout < 64):
        continue
    if prec == "h3":
        wpk, w_exp = ops.pack_weights_up(k, h3=True) if ups else ops.pack_weights_h3(k, stride=stride)
        kw = dict(precision=_lib.PREC_H3, w_exp=w_exp)
    else:
        wpk, kw = (ops.pack_weights_up(k) if ups else ops.pack_weights(k)), {}
    eo = e * (2 if ups else 1) // stride
    args = dict(x2=x2, bias=torch.randn(cout, device=dev), stride=stride, upsample=bool(ups), **kw)
    if pro:
        args.update(pro_scale=torch.rand(c1 + c2, device=dev) + 0.5, pro_shift=torch.randn(c1 + c2, device=dev) * 0.1)
    if res:
        args.update(res=torch.randn(B, eo, eo, eo, cout, device=dev))
    if vec:
        args.update(vec=torch.randn(B, cout, device=dev))
    ops.conv3d(x1, wpk, cout, ks, **args)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.conv3d(x1, wpk, cout, ks, **args)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3)
    fl = 2.0 * ks ** 3 * (c1 + c2) * cout * B * eo ** 3
    print(f"{name:28s} {best:8.3f} ms {fl / best / 1e9:8.1f} TF")

"""Error of the Winograd kernel's fused 1x1 skip tail against float64, split into the k3 part and the skip part (the k3 conv is run with
zero skip weights and subtracted): a stale `lo` operand register in the tail (RAW hazard in front of its asm MFMAs, fixed in round 5) shows
in the skip part only.  usage: [DM3D_LIB=variants/r04.so] python tools/skip_tail_err.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dm3d_amd import ops, _lib
dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "product"
os.environ["DM3D_CONV_WIDE_WGS"] = "1"; os.environ["DM3D_CONV_WINO_MINCHUNKS"] = "1"
for name, B, e, cm, s1, s2, cout in [("16^3 64->64 + k1(192)", 1, 16, 64, 128, 64, 64), ("16^3 128->128 + k1(96)", 2, 16, 128, 64, 32, 128), ("8^3 64->64 + k1(32)", 4, 8, 64, 32, 0, 64),
                                     ("32^3 64->64 + k1(192) B=2", 2, 32, 64, 128, 64, 64)]:
    torch.manual_seed(7)
    h = torch.randn(B, e, e, e, cm, device=dev)
    x1 = torch.randn(B, e, e, e, s1, device=dev) * 3.0
    x2 = torch.randn(B, e, e, e, s2, device=dev) * 3.0 if s2 else None
    k = torch.randn(3, 3, 3, cm, cout, device=dev) * 0.02
    ks = torch.randn(1, 1, 1, s1 + s2, cout, device=dev) * 0.3
    w_exp = ops.h3_weight_exponent(k.cpu(), ks.cpu())
    wpk, _ = ops.pack_weights_h3(k, w_exp=w_exp)
    wino = ops.pack_weights_h3w(k, w_exp)
    sfrag, swpk = ops.pack_weights_skip_h3f(ks, w_exp), ops.pack_weights_skip_h3p(ks, w_exp)
    kw = dict(precision=_lib.PREC_H3, w_exp=w_exp)
    y = ops.conv3d(h, wpk, cout, 3, wpk_wino=wino, skip=(x1, x2, swpk, sfrag), **kw)
    yd = ops.conv3d(h, wpk, cout, 3, skip=(x1, x2, swpk), **kw)
    torch.cuda.synchronize()
    xs = torch.cat([x1, x2], -1) if s2 else x1
    r3 = F.conv3d(h.double().permute(0, 4, 1, 2, 3), k.double().permute(4, 3, 0, 1, 2), padding=1).permute(0, 2, 3, 4, 1)
    rs = torch.einsum("bdhwc,co->bdhwo", xs.double(), ks.double()[0, 0, 0])
    yr = r3 + rs
    rel = lambda a: float((a.double() - yr).abs().max() / yr.abs().max())
    print(f"{tag:8s} {name:30s} wino+tail {rel(y):.2e}   direct+tail {rel(yd):.2e}   (|skip| / |k3| = {float(rs.abs().max() / r3.abs().max()):.1f})", flush=True)

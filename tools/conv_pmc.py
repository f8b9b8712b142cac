"""One launch each of a few conv shapes (for rocprofv3 --pmc).  usage: python tools/conv_pmc.py [h3|fp32] [case name]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib
prec = sys.argv[1] if len(sys.argv) > 1 else "h3"
dev = torch.device("cuda:0"); B = 32
only = sys.argv[2] if len(sys.argv) > 2 else None
for name, e, c1, cout, ups, pro in [("plain64", 32, 64, 64, 0, 0), ("pro64", 32, 64, 64, 0, 1), ("pro192", 32, 192, 64, 0, 1), ("pro128_16", 16, 128, 128, 0, 1), ("up128", 16, 128, 128, 1, 0)]:
    if only and name != only: continue
    x1 = torch.randn(B, e, e, e, c1, device=dev)
    k = torch.randn(3, 3, 3, c1, cout, device=dev) * 0.05
    if ups:
        r = ops.pack_weights_up(k, h3=prec == "h3")
        wpk, w_exp = r if prec == "h3" else (r, 0)
    else:
        wpk, w_exp = ops.pack_weights_h3(k) if prec == "h3" else (ops.pack_weights(k), 0)
    kw = dict(precision=_lib.PREC_H3, w_exp=w_exp) if prec == "h3" else {}
    if pro:
        kw.update(pro_scale=torch.rand(c1, device=dev) + 0.5, pro_shift=torch.randn(c1, device=dev) * 0.1)
    for _ in range(3):
        ops.conv3d(x1, wpk, cout, 3, upsample=bool(ups), **kw)
    torch.cuda.synchronize()

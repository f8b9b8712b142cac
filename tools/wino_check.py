"""Winograd-x conv form (dm3d_conv_h3w.hip) against the direct free-running form on the same inputs, and both against a float64 reference;
then an interleaved timing A/B (DM3D_CONV_WINO=0/1) on the U-Net's large k3 shapes.  usage: python tools/wino_check.py [time]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dm3d_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)

def ref_conv(x, k, bias=None, pro=None, res=None):
    xd = x.double()
    if pro is not None:
        xd = xd * pro[0].double() + pro[1].double()
        xd = xd * torch.sigmoid(xd)
    y = F.conv3d(xd.permute(0, 4, 1, 2, 3), k.double().permute(4, 3, 0, 1, 2), padding=1).permute(0, 2, 3, 4, 1)
    if bias is not None: y = y + bias.double()
    if res is not None: y = y + res.double()
    return y

def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())

os.environ["DM3D_CONV_WIDE_WGS"] = "1"; os.environ["DM3D_CONV_WINO_MINCHUNKS"] = "1"
bad = 0
for name, B, e, c1, c2, cout, pro, res in [("plain 8^3 16->64", 1, 8, 16, 0, 64, 0, 0), ("plain 8^3 32->64", 1, 8, 32, 0, 64, 0, 0), ("plain 8^3 48->64", 1, 8, 48, 0, 64, 0, 0), ("pro 8^3 16->64", 1, 8, 16, 0, 64, 1, 0), ("pro 8^3 32->64 +res", 2, 8, 32, 0, 64, 1, 1), ("pro concat 16^3 64+32->128", 1, 16, 64, 32, 128, 1, 0),
                                           ("pro 16^3 24->64 (ragged cin)", 1, 16, 24, 0, 64, 1, 1), ("pro 8x16x24 40->96", 1, (8, 16, 24), 40, 0, 96, 1, 0)]:
    dims = (e, e, e) if isinstance(e, int) else e
    x1 = torch.randn(B, *dims, c1, device=dev)
    x2 = torch.randn(B, *dims, c2, device=dev) if c2 else None
    cin = c1 + c2
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
    wpk, w_exp = ops.pack_weights_h3(k)
    wino = ops.pack_weights_h3w(k, w_exp)
    bias = torch.randn(cout, device=dev)
    ps = (torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1) if pro else None
    r = torch.randn(B, *dims, cout, device=dev) if res else None
    kw = dict(x2=x2, bias=bias, pro_scale=ps[0] if pro else None, pro_shift=ps[1] if pro else None, res=r, precision=_lib.PREC_H3, w_exp=w_exp)
    y0 = ops.conv3d(x1, wpk, cout, 3, **kw)
    y1 = ops.conv3d(x1, wpk, cout, 3, wpk_wino=wino, **kw)
    xx = torch.cat([x1, x2], -1) if c2 else x1
    yr = ref_conv(xx, k, bias, ps, r)
    e0, e1, d = rel(y0, yr), rel(y1, yr), rel(y1, y0)
    ok = e1 < 2e-5 and not torch.equal(y0, y1)
    bad += not ok
    print(f"{name:34s} direct {e0:.2e}  wino {e1:.2e}  wino-direct {d:.2e}  {'ok' if ok else 'FAIL (or the Winograd form did not run)'}", flush=True)
# Cin split: a grid of 128 workgroups (B = 32 at 8^3) runs as two workgroups per brick, their halves meeting inside the launch
os.environ.pop("DM3D_CONV_WIDE_WGS", None)
for name, B, e, cin, cout, res in [("split 8^3 B=32 256->256 +res", 32, 8, 256, 256, 1), ("split 8^3 B=32 512->256", 32, 8, 512, 256, 0)]:
    x = torch.randn(B, e, e, e, cin, device=dev)
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
    wpk, w_exp = ops.pack_weights_h3(k)
    wino = ops.pack_weights_h3w(k, w_exp)
    bias = torch.randn(cout, device=dev)
    ps = (torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1)
    r = torch.randn(B, e, e, e, cout, device=dev) if res else None
    kw = dict(bias=bias, pro_scale=ps[0], pro_shift=ps[1], res=r, precision=_lib.PREC_H3, w_exp=w_exp)
    y0 = ops.conv3d(x, wpk, cout, 3, **kw)
    y1 = ops.conv3d(x, wpk, cout, 3, wpk_wino=wino, **kw)
    yr = ref_conv(x[:4], k, bias, ps, r[:4] if res else None)
    e0, e1 = rel(y0[:4], yr), rel(y1[:4], yr)
    ok = e1 < 2e-5 and not torch.equal(y0, y1) and float((y1 - y0).abs().max() / y0.abs().max()) < 2e-5
    bad += not ok
    print(f"{name:34s} direct {e0:.2e}  wino {e1:.2e}  wino-direct {rel(y1, y0):.2e}  {'ok' if ok else 'FAIL (or the Winograd form did not run)'}", flush=True)
os.environ["DM3D_CONV_WIDE_WGS"] = "1"
# fused 1x1 skip conv (ResidualBlock tail: conv2(...) + Conv3D(width, 1)(concat(x, skip))) as a tail phase of the Winograd kernel
os.environ["DM3D_CONV_WIDE_WGS"] = "1"; os.environ["DM3D_CONV_WINO_MINCHUNKS"] = "1"
for name, B, e, cm, s1, s2, cout in [("skip 8^3 64->64 + k1(32)", 2, 8, 64, 32, 0, 64), ("skip 16^3 128->128 + k1(64+32)", 1, 16, 128, 64, 32, 128),
                                     ("skip 8^3 32->96 + k1(40) ragged", 1, 8, 32, 40, 0, 96)]:
    h = torch.randn(B, e, e, e, cm, device=dev)
    x1 = torch.randn(B, e, e, e, s1, device=dev); x2 = torch.randn(B, e, e, e, s2, device=dev) if s2 else None
    k = torch.randn(3, 3, 3, cm, cout, device=dev) * 0.05
    ks = torch.randn(1, 1, 1, s1 + s2, cout, device=dev) * 0.2
    w_exp = ops.h3_weight_exponent(k.cpu(), ks.cpu())
    wpk, _ = ops.pack_weights_h3(k, w_exp=w_exp)
    wino = ops.pack_weights_h3w(k, w_exp)
    swpk, sfrag = ops.pack_weights_skip_h3p(ks, w_exp), ops.pack_weights_skip_h3f(ks, w_exp)
    bias = torch.randn(cout, device=dev)
    ps = (torch.rand(cm, device=dev) + 0.5, torch.randn(cm, device=dev) * 0.1)
    kw = dict(bias=bias, pro_scale=ps[0], pro_shift=ps[1], precision=_lib.PREC_H3, w_exp=w_exp)
    y0 = ops.conv3d(h, wpk, cout, 3, skip=(x1, x2, swpk), **kw)
    y1 = ops.conv3d(h, wpk, cout, 3, wpk_wino=wino, skip=(x1, x2, swpk, sfrag), **kw)
    xs = torch.cat([x1, x2], -1) if s2 else x1
    yr = ref_conv(h, k, bias, ps) + torch.einsum("bdhwc,co->bdhwo", xs.double(), ks.double()[0, 0, 0])
    e0, e1 = rel(y0, yr), rel(y1, yr)
    ok = e1 < 2e-5 and not torch.equal(y0, y1)
    bad += not ok
    print(f"{name:34s} direct {e0:.2e}  wino {e1:.2e}  wino-direct {rel(y1, y0):.2e}  {'ok' if ok else 'FAIL (or the Winograd form did not run)'}", flush=True)
# hand-off pair (ResidualBlock conv1 -> norm + SiLU -> conv2): conv A stores DM3D_FMT_H2 behind its fused post-activation, conv B reads it (kernel MODE 2)
for name, B, e, c, cm in [("hand-off 8^3 96->128->64", 1, 8, 96, 128), ("hand-off 16^3 128->192->128", 1, 16, 128, 192)]:
    x = torch.randn(B, e, e, e, c, device=dev)
    ka, kb = torch.randn(3, 3, 3, c, cm, device=dev) * 0.05, torch.randn(3, 3, 3, cm, c if c != 96 else 64, device=dev) * 0.05
    co = kb.shape[-1]
    wa, ea = ops.pack_weights_h3(ka); wb, eb = ops.pack_weights_h3(kb)
    wwa, wwb = ops.pack_weights_h3w(ka, ea), ops.pack_weights_h3w(kb, eb)
    post = (torch.rand(cm, device=dev) + 0.5, torch.randn(cm, device=dev) * 0.1)
    pro = (torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1)
    outs = []
    for wino in (False, True):
        a = ops.conv3d(x, wa, cm, 3, bias=torch.zeros(cm, device=dev), pro_scale=pro[0], pro_shift=pro[1], precision=_lib.PREC_H3, w_exp=ea, post=post, out_h2=True,
                       wpk_wino=wwa if wino else None)
        y = ops.conv3d(a, wb, co, 3, precision=_lib.PREC_H3, w_exp=eb, x1_h2_channels=cm, wpk_wino=wwb if wino else None)
        outs.append(y)
    mid = ref_conv(x, ka, None, pro)
    mid = mid * post[0].double() + post[1].double(); mid = mid * torch.sigmoid(mid)
    yr = ref_conv(mid.float(), kb)
    e0, e1 = rel(outs[0], yr), rel(outs[1], yr)
    ok = e1 < 2e-5 and not torch.equal(outs[0], outs[1])
    bad += not ok
    print(f"{name:34s} direct {e0:.2e}  wino {e1:.2e}  {'ok' if ok else 'FAIL (or the Winograd form did not run)'}", flush=True)
print("failures:", bad)
if len(sys.argv) > 1 and (not bad or os.environ.get("WINO_TIMING_ONLY")):      # WINO_TIMING_ONLY: timing-only variant builds (DM3D_LIB) with wrong results
    del os.environ["DM3D_CONV_WIDE_WGS"]; os.environ["DM3D_CONV_WINO_MINCHUNKS"] = "1"
    B = int(os.environ.get("AB_BATCH", "32"))
    for name, e, cin, cout, res in [("32^3 32->64 pro", 32, 32, 64, 0), ("32^3 64->64 pro+res", 32, 64, 64, 1), ("32^3 96->64 pro", 32, 96, 64, 0), ("32^3 192->64 pro", 32, 192, 64, 0),
                                    ("16^3 128->128 pro+res", 16, 128, 128, 1), ("16^3 384->128 pro", 16, 384, 128, 0), ("8^3 256->256 pro", 8, 256, 256, 0), ("8^3 512->256 pro", 8, 512, 256, 0), ("8^3 256->256 pro+res", 8, 256, 256, 1)]:
        x = torch.randn(B, e, e, e, cin, device=dev)
        k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
        wpk, w_exp = ops.pack_weights_h3(k)
        wino = ops.pack_weights_h3w(k, w_exp)
        kw = dict(bias=torch.randn(cout, device=dev), pro_scale=torch.rand(cin, device=dev) + 0.5, pro_shift=torch.randn(cin, device=dev) * 0.1,
                  res=torch.randn(B, e, e, e, cout, device=dev) if res else None, precision=_lib.PREC_H3, w_exp=w_exp, wpk_wino=wino)
        times = {"0": [], "1": []}
        for v in times:
            os.environ["DM3D_CONV_WINO"] = v
            ops.conv3d(x, wpk, cout, 3, **kw)
        torch.cuda.synchronize()
        for _ in range(5):
            for v in times:
                os.environ["DM3D_CONV_WINO"] = v
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8): ops.conv3d(x, wpk, cout, 3, **kw)
                e1.record(); torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 8)
        fl = 2.0 * 27 * cin * cout * B * e ** 3
        print(f"{name:24s} " + "  ".join(f"WINO={v}: med {statistics.median(t):.4f} ms ({fl / statistics.median(t) / 1e9:.0f} TF algorithmic)" for v, t in times.items()), flush=True)

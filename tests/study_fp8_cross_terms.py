"""Precision study (CPU, oracle only — a measurement tool, not product code): the three-pass split a.b = ah.bh + al.bh + ah.bl with the two
CROSS terms evaluated on float8 (e4m3) operands — fp8 MFMA runs at twice the float16 rate, so the product would cost 2 units instead of 3.
   a = ah + al (ah = fp16(a)),  w*2^e = wh + wl (wh = fp16),  scales 2^Sa / 2^Sb keep the small terms inside e4m3's range:
   a.w ~ ah.wh + f8(al 2^Sa).f8(wh 2^-Sa) + f8(ah 2^-Sb).f8(wl 2^Sb)
Prints the relative error of eps (max-abs / max and L2) against the exact float32 network, next to the current scheme (h3: lo terms in
float16) for calibration."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_torch as rt
torch.set_num_threads(8)
orig = rt._conv3d
MODE = {"m": "exact"}
SA, SB = (int(a) for a in (sys.argv[1:3] + ['15', '5'][len(sys.argv) - 1:]))
F8 = torch.float8_e4m3fn


def r16(t):
    return t.to(torch.float16).to(t.dtype)


def r8(t):
    return t.clamp(-448, 448).to(F8).to(t.dtype)


def conv(x, kernel, bias, stride=1):
    m = MODE["m"]
    if m == "exact":
        return orig(x, kernel, bias, stride)
    e = 13 - np.floor(np.log2(float(kernel.abs().max())))
    s = 2.0 ** e
    w = kernel * s
    wh = r16(w); wl = w - wh
    xc = x.clamp(-65504, 65504)
    xh = r16(xc); xl = xc - xh
    zero = torch.zeros_like(bias) if bias is not None else None
    if m == "h3":
        y = orig(xh, wh, zero, stride) + orig(r16(xl), wh, zero, stride) + orig(xh, r16(wl), zero, stride)
    elif m == "f8":
        Sa, Sb = 2.0 ** SA, 2.0 ** SB
        y = orig(xh, wh, zero, stride) + orig(r8(xl * Sa), r8(wh / Sa), zero, stride) + orig(r8(xh / Sb), r8(wl * Sb), zero, stride)
    y = y / s
    return y + bias if bias is not None else y


rt._conv3d = conv
for size, ch, B in ((8, 4, 2), (16, 8, 2)):
    cfg = rt.UNetConfig(img_size=size, img_channels=ch)
    for seed in (0, 1):
        W = rt.synthetic_weights(cfg, seed)
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.randn(B, size, size, size, ch, generator=g)
        for tval in (999, 500, 10):
            t = torch.full((B,), tval, dtype=torch.int64)
            ctx = torch.ones(B, 1, dtype=torch.int64)
            MODE["m"] = "exact"
            ref = rt.unet_forward(W, cfg, x, t, ctx)
            out = {}
            for m in ("h3", "f8"):
                MODE["m"] = m
                y = rt.unet_forward(W, cfg, x, t, ctx)
                out[m] = (float((y - ref).abs().max() / ref.abs().max()), float((y - ref).norm() / ref.norm()))
            print(size, ch, "seed", seed, "t", tval, {k: f"max {v[0]:.2e} l2 {v[1]:.2e}" for k, v in out.items()}, flush=True)
        if size == 32:
            break

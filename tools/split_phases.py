"""Where the time of a Cin-split conv launch goes at small batch: per-workgroup phase stamps of conv3d_igemm_h3v3 (the stamped variant library
of tools/mk_stamp_variants.py), one launch at a time with the chip idle in between (as inside a small-batch step: every launch is one round of
workgroups).
    DM3D_LIB=<csrc>/variants/cck.so python tools/split_phases.py [B=1]
Stamps: 0 entry | 2 first halo + weight pieces back | 1 image stored, first fragments requested | 28 chunk loop done | 19 skip phase done |
20 partial tile stored (a part that is not the last arriver ends here) | 21 ticket known | 22 parts summed | 29 epilogue done."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib

dev = torch.device("cuda:0")
raw = C.CDLL(_lib.LIB_PATH)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
st = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
raw.dm3d_debug_set_stamps_conv(C.c_void_p(st.data_ptr()))


def med(x): return float(x.median()) if x.numel() else float("nan")


for name, e, cin, cout in (("8^3 128->256", 8, 128, 256), ("8^3 256->256", 8, 256, 256), ("8^3 512->256", 8, 512, 256), ("16^3 128->128", 16, 128, 128), ("16^3 384->128", 16, 384, 128)):
    x = torch.randn(B, e, e, e, cin, device=dev)
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
    wpk, w_exp = ops.pack_weights_h3(k)
    kw = dict(bias=torch.randn(cout, device=dev), pro_scale=torch.rand(cin, device=dev) + 0.5, pro_shift=torch.randn(cin, device=dev) * 0.1,
              precision=_lib.PREC_H3, w_exp=w_exp)
    kw["res"] = torch.randn(B, e, e, e, cout, device=dev)
    for _ in range(3): ops.conv3d(x, wpk, cout, 3, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = []
    for _ in range(20):
        st.zero_(); torch.cuda.synchronize()
        e0.record(); ops.conv3d(x, wpk, cout, 3, **kw); e1.record(); torch.cuda.synchronize()
        s = st.view(4096, 32).cpu().double()
        rows.append((e0.elapsed_time(e1) * 1e3, s[s[:, 0] > 0]))
    rows.sort(key=lambda r: r[0])
    us, s = rows[len(rows) // 2]
    ghz = med((s[:, 28] - s[:, 1]) / (s[:, 31] - s[:, 30]).clamp(min=1)) * 0.1
    t = lambda a, b_, m=None: med(((s[:, a] - s[:, b_])[(s[:, a] > 0) & (s[:, b_] > 0) if m is None else m]) / (ghz * 1e3))
    last = s[:, 22] > 0
    span0 = s[:, 24].min()
    end = torch.where(s[:, 25] > 0, s[:, 25], s[:, 26])
    print(f"{name} B={B}: {s.shape[0]} workgroups stamped ({int(last.sum())} last arrivers), event-to-event {us:.1f} us, first entry -> last end "
          f"{(end.max() - span0) / 100:.1f} us, entries spread over {(s[:, 24].max() - span0) / 100:.1f} us, clock {ghz:.2f} GHz")
    print(f"    per workgroup [us, median]: entry->halo back {t(2, 0):.2f} | convert+store image {t(1, 2):.2f} | chunk loop {t(28, 1):.2f} | skip phase {t(19, 28):.2f} | "
          f"partial store {t(20, 19):.2f} | ticket {t(21, 20):.2f} | gather {t(22, 21):.2f} | epilogue {t(29, 22):.2f}")
    if last.any():
        print(f"    last arrivers: entry -> end {med((s[:, 25] - s[:, 24])[last]) / 100:.1f} us; others: entry -> partial stored {med((s[:, 26] - s[:, 24])[~last & (s[:, 26] > 0)]) / 100:.1f} us")
raw.dm3d_debug_set_stamps_conv(C.c_void_p(0))

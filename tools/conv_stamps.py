"""In-kernel timeline of conv3d_igemm_h3v2 (needs the stamp variant library: DM3D_LIB=.../variants/cst.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib
dev = torch.device("cuda:0")
raw = C.CDLL(_lib.LIB_PATH)
B = 32
F8 = "f8" in sys.argv[1:]           # the float8 cross-term form (conv3d_igemm_h3v2<3, 1, 8, 3, 1>) instead of the three-pass form
st = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)      # rows 0..2047: phase stamps per workgroup; rows 2048..: ping-pong segments
for name, e, cin, cout, res in (("32^3 64->64", 32, 64, 64, 1), ("32^3 192->64", 32, 192, 64, 0), ("16^3 128->128", 16, 128, 128, 1), ("8^3 256->256", 8, 256, 256, 1)):
    x = torch.randn(B, e, e, e, cin, device=dev)
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
    wpk, w_exp = ops.pack_weights_h3(k)
    kw = dict(bias=torch.randn(cout, device=dev), pro_scale=torch.rand(cin, device=dev) + 0.5, pro_shift=torch.randn(cin, device=dev) * 0.1,
              res=torch.randn(B, e, e, e, cout, device=dev) if res else None, precision=_lib.PREC_H3, w_exp=w_exp,
              wpk_f8=ops.pack_weights_h3f8(k, w_exp) if F8 else None)
    for _ in range(2): ops.conv3d(x, wpk, cout, 3, **kw)
    torch.cuda.synchronize()
    raw.dm3d_debug_set_stamps_conv(C.c_void_p(st.data_ptr())); st.zero_()
    ops.conv3d(x, wpk, cout, 3, **kw)
    torch.cuda.synchronize()
    raw.dm3d_debug_set_stamps_conv(C.c_void_p(0))
    nb = B * (e // 8) ** 3 * (cout // 64)
    s = st.view(4096, 32).cpu()[:min(nb, 4096)].double()
    nch = min(cin // 16, 12)
    print(f"{name}: bricks {nb}, chunks {cin // 16}")
    print("  start->first halo issued:", int((s[:, 1] - s[:, 0]).mean()), " ->chunk0 staged:", int((s[:, 2] - s[:, 1]).mean()))
    mf = [(s[:, 3 + 2 * c] - s[:, 2 + 2 * c]).mean() for c in range(nch)]
    stg = [(s[:, 4 + 2 * c] - s[:, 3 + 2 * c]).mean() for c in range(nch - 1)]
    print("  per chunk, first store / barrier to the end of its last MFMA segment (ping-pong loop: ST .. C13):", [int(v) for v in mf])
    print("  between chunks (per-group loop: convert + store + 2 barriers; ping-pong loop: ~0, the boundary is inside the chunk):", [int(v) for v in stg])
    last = 3 + 2 * (nch - 1)
    if cin // 16 <= 12:
        print("  last group -> epilogue start:", int((s[:, 28] - s[:, last]).mean()), " epilogue:", int((s[:, 29] - s[:, 28]).mean()))
    print("  WG total:", int((s[:, 29] - s[:, 0]).mean()))
    pp = st.view(4096, 32).cpu()[2048:2048 + min(nb, 2048)].double()
    if cin // 16 >= 2 and float(pp[:, 1].max()) > 0:
        # entry i of a half: time at which the wave REACHED the barrier that ends its position i of the second chunk
        # (0: ST, 1: X entered..., then L0, C0, L1, C1, ...); durations = differences; half 1 runs one position behind half 0
        for hname, off in (("wave 0 (half 0)", 0), ("wave 4 (half 1)", 16)):
            t = pp[:, off:off + 15]
            d = (t[:, 1:] - t[:, :-1]).mean(0)
            print(f"  ping-pong, {hname}: arrive-to-arrive per position [ST..X, X..L0-end, then C0, L1, C1, L2, C2, ...]:", [int(v) for v in d])
        rel = (pp[:, 16:31] - pp[:, 0:15]).mean(0)
        print("  half 1 minus half 0 at the same position:", [int(v) for v in rel])
    if name == "32^3 64->64":
        for lo in (0, 512, 1024, 1536):
            seg = s[lo:lo + 512]
            print(f"  WGs {lo}..{lo+511}: chunk MFMA phases", [int((seg[:, 3 + 2 * c] - seg[:, 2 + 2 * c]).mean()) for c in range(nch)],
                  "prologue", int((seg[:, 2] - seg[:, 0]).mean()), "total", int((seg[:, 29] - seg[:, 0]).mean()))

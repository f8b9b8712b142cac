"""In-kernel shader clock of the conv (and GEMM) kernels as MI355X_MICROARCH.md 'DVFS give-back' item 6 defines it:
delta(s_memtime) / delta(s_memrealtime) x 100 MHz, one stamp pair around the main loop of every workgroup, read after >= 2 s of
back-to-back launches on random data, median over workgroups.  Needs the clock-stamp variant libraries of tools/mk_stamp_variants.py:
    DM3D_LIB=<csrc>/variants/cck.so python tools/kernel_clock.py conv     (conv3d_igemm_h3v3; DM3D_CONV_V3_TD applies)
    DM3D_CONV_V3=0 DM3D_LIB=<csrc>/variants/cck_v2.so python tools/kernel_clock.py conv     (conv3d_igemm_h3v2)
    DM3D_LIB=<csrc>/variants/gst.so python tools/kernel_clock.py gemm
Also prints the matrix-pipe duty of the stamped interval: MFMA cycles one SIMD must spend (16 per v_mfma_f32_16x16x32_f16) / delta(s_memtime).
The stamps go to a buffer of their own (never to an output); the product library carries none of this."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops, _lib

dev = torch.device("cuda:0")
raw = C.CDLL(_lib.LIB_PATH)
what = sys.argv[1] if len(sys.argv) > 1 else "conv"
B = int(os.environ.get("CLOCK_BATCH", "32"))
SECONDS = float(os.environ.get("CLOCK_SECONDS", "2.0"))


def sustained(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < SECONDS:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


rows_out = []
if what == "conv":
    st = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
    wino = os.environ.get("CLOCK_WINO") == "1"           # the Winograd-x form (dm3d_conv_h3w.hip): one wave per SIMD, 20 steps of 48 MFMAs per chunk
    (raw.dm3d_debug_set_stamps_wino if wino else raw.dm3d_debug_set_stamps_conv)(C.c_void_p(st.data_ptr()))
    # (the Winograd form serves Cin >= 96: its first shape is the smallest conv of the U-Net that takes it)
    for name, e, cin, cout, res in ((("32^3 96->64", 32, 96, 64, 0) if wino else ("32^3 64->64", 32, 64, 64, 1)), ("32^3 192->64", 32, 192, 64, 0),
                                    ("16^3 128->128", 16, 128, 128, 1), ("16^3 384->128", 16, 384, 128, 0)):
        x = torch.randn(B, e, e, e, cin, device=dev)
        k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
        if os.environ.get("CLOCK_ZEROS") == "1":      # all-zero operands: the clock the chip holds when the data costs no switching power
            x.zero_(); k.zero_()
        wpk, w_exp = ops.pack_weights_h3(k)
        kw = dict(bias=torch.randn(cout, device=dev), pro_scale=torch.rand(cin, device=dev) + 0.5, pro_shift=torch.randn(cin, device=dev) * 0.1,
                  res=torch.randn(B, e, e, e, cout, device=dev) if res else None, precision=_lib.PREC_H3, w_exp=w_exp)
        if wino: kw["wpk_wino"] = ops.pack_weights_h3w(k, w_exp)
        st.zero_()
        ms = sustained(lambda: ops.conv3d(x, wpk, cout, 3, **kw))
        s = st.view(4096, 32).cpu().double()
        s = s[s[:, 1] > 0]                            # the workgroups that stamped (blockIdx.y == 0, the first 4096)
        dt, dr = s[:, 28] - s[:, 1], s[:, 31] - s[:, 30]
        ok = dr > 0
        f0 = (s[:, 0] > 0) & (s[:, 29] > 0)      # (stamp 0 = kernel entry: the first item of each persistent workgroup only)
        ghz = (dt[ok] / dr[ok] * 0.1)
        nch = cin // 16
        mfma_cycles = nch * 20 * 48 * 16 if wino else 2 * nch * 14 * 48 * 16          # (direct form: two waves per SIMD)
        fl = 2.0 * 27 * cin * cout * B * e ** 3
        print(f"{name}: {ms:.3f} ms {fl / ms / 1e9:.0f} TF/s algorithmic | in-kernel clock median {ghz.median():.3f} GHz (p10 {ghz.quantile(0.1):.3f}, p90 {ghz.quantile(0.9):.3f}) "
              f"| chunk loop {dt.median():.0f} ticks = {dr.median() / 100:.1f} us, MFMA duty in the loop {mfma_cycles / dt.median():.3f} "
              f"| {s.shape[0]} items stamped; first item of a workgroup {((s[:, 29] - s[:, 0])[f0]).median() if f0.any() else 0:.0f} ticks "
              f"(prologue {((s[:, 1] - s[:, 0])[f0]).median() if f0.any() else 0:.0f}), epilogue {((s[:, 29] - s[:, 28])[s[:, 29] > 0]).median() if (s[:, 29] > 0).any() else 0:.0f}", flush=True)
        if wino: print(f"    epilogue stamps: accumulators -> LDS {(s[:, 20] - s[:, 28]).median():.0f}, slice 0 {(s[:, 21] - s[:, 20]).median():.0f}, slice 1 {((s[:, 29] - s[:, 21])[s[:, 29] > 0]).median():.0f} ticks", flush=True)
        rows_out.append((name, ms, fl / ms / 1e9, float(ghz.median()), float(ghz.quantile(0.1)), float(ghz.quantile(0.9)), mfma_cycles / float(dt.median())))
    (raw.dm3d_debug_set_stamps_wino if wino else raw.dm3d_debug_set_stamps_conv)(C.c_void_p(0))
else:
    st = torch.zeros(2048 * 16, dtype=torch.int64, device=dev)
    raw.dm3d_debug_set_stamps(C.c_void_p(st.data_ptr()))
    for m, n, k in ((16384, 1024, 256), (16384, 256, 1024), (16384, 256, 256)):
        a = ops.split_h2(torch.randn(m, k, device=dev))
        w = ops.split_h2(torch.randn(n, k, device=dev) * 0.05)
        out = torch.empty(m, n, device=dev)
        st.zero_()
        kw = dict(m=m, n=n, k=k, lda=k, ldb=k, batch=1, stride_a=m * k, stride_b=0, bias=torch.randn(n, device=dev), act=_lib.ACT_NONE, res=None,
                  out=out, precision=_lib.PREC_H3, a_fmt=_lib.FMT_H2, b_fmt=_lib.FMT_H2, out_fmt=_lib.FMT_F32)
        ms = sustained(lambda: ops.gemm_tn(a, w, **kw))
        s = st.view(2048, 16).cpu().double()
        dt, dr = s[:, 10] - s[:, 1], s[:, 15] - s[:, 14]
        ok = dr > 0
        ghz = dt[ok] / dr[ok] * 0.1
        print(f"gemm m={m} n={n} k={k}: {ms * 1e3:.1f} us {2.0 * m * n * k / ms / 1e9:.0f} TF/s | in-kernel clock median {ghz.median():.3f} GHz "
              f"(p10 {ghz.quantile(0.1):.3f}, p90 {ghz.quantile(0.9):.3f}); K loop {dt.median():.0f} ticks", flush=True)
    raw.dm3d_debug_set_stamps(C.c_void_p(0))

if os.environ.get("CLOCK_OUT") and rows_out:
    # "# workload:" carries the digest of the kernel sources: bench.py attaches roofline.clock_ghz only from a file measured on them
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    with open(os.environ["CLOCK_OUT"], "w") as f:
        f.write("# in-kernel shader clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz around the chunk loop, median over workgroups, after "
                f"{SECONDS:.0f} s of back-to-back launches on {'ZERO' if os.environ.get('CLOCK_ZEROS') == '1' else 'random'} data (tools/kernel_clock.py conv, B=32, k3 conv with norm+SiLU prologue; "
                f"kernel {'conv3d_igemm_h3w<1> (Winograd-x form, CLOCK_WINO=1)' if wino else 'conv3d_igemm_h3v3<3, 1, 8, 4> (direct form)'})\n")
        f.write(f"# workload: kernel_clock conv{' wino' if wino else ''} csrc={bench.csrc_digest()}\n")
        f.write("shape,ms,algorithmic_tflops,clock_ghz_median,clock_ghz_p10,clock_ghz_p90,mfma_duty_in_loop\n")
        for r in rows_out:
            f.write(f'"{r[0]}",{r[1]:.4f},{r[2]:.1f},{r[3]:.3f},{r[4]:.3f},{r[5]:.3f},{r[6]:.3f}\n')

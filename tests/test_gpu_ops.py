"""GPU parity tests, kernel by kernel, through the C ABI, against the CPU oracle (oracle/ref_torch.py) in float64.

Tolerances: fp32 MFMA accumulates K products with one rounding each, so |err| <= ~K * 2^-24 * sum|a.b|; the checks use
max|err| / max|ref| <= 2e-5 for contractions (1e-3 is the end-to-end budget north_star states) and 1e-6 for elementwise.
"""
import ctypes as C
import math
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from dm3d_amd import _lib
    _lib.require_device()
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


def _rel(a, ref):
    ref = ref.double()
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _conv_ref(x1, kern, bias, *, x2=None, stride=1, upsample=False, pro=None, vec=None, vec_idx=None, relu=False, res=None):
    from oracle import ref_torch as rt
    x = x1.double() if x2 is None else torch.cat([x1.double(), x2.double()], -1)
    if pro is not None:
        x = rt._swish(x * pro[0].double() + pro[1].double())
    if upsample:
        x = rt._upsample2(x)
    y = rt._conv3d(x, kern.double(), None if bias is None else bias.double(), stride=stride)
    if vec is not None:
        rows = vec_idx.long() if vec_idx is not None else torch.arange(x1.shape[0])
        y = y + vec.double()[rows][:, None, None, None, :y.shape[-1]]
    if relu:
        y = torch.relu(y)
    if res is not None:
        y = y + res.double()
    return y


CONV_CASES = [
    # name, B, (D,H,W), c1, c2, cout, ksize, stride, upsample, extras
    ("k3_exact_tile", 2, (8, 8, 8), 16, 0, 64, 3, 1, False, ""),
    ("k3_noncubic_coutmask_bias", 1, (4, 8, 16), 32, 0, 40, 3, 1, False, "bias"),
    ("k3_all_epilogue", 2, (8, 8, 8), 32, 0, 64, 3, 1, False, "bias pro vec res relu"),
    ("k3_vec_idx", 3, (8, 8, 8), 16, 0, 64, 3, 1, False, "bias vec vecidx"),
    ("k3_dual_input_pro", 2, (8, 8, 8), 16, 32, 64, 3, 1, False, "bias pro"),
    ("k3_conv_in_c8", 2, (8, 8, 8), 8, 0, 32, 3, 1, False, "bias"),
    ("k3_conv_in_c4", 1, (8, 8, 8), 4, 0, 32, 3, 1, False, "bias"),
    ("k3_conv_out_c8", 1, (8, 8, 8), 64, 0, 8, 3, 1, False, "bias pro"),
    ("k3_cout128", 1, (8, 8, 8), 32, 0, 128, 3, 1, False, "bias"),
    ("k3_partial_brick_4", 2, (4, 4, 4), 32, 0, 64, 3, 1, False, "bias res"),
    ("k3_partial_brick_2", 2, (2, 2, 2), 16, 0, 64, 3, 1, False, "bias"),
    ("k3_16cube", 1, (16, 16, 16), 16, 0, 64, 3, 1, False, "bias"),
    # small grids with >= 4 Cin chunks take the Cin split of the h3 kernel (up to 16 workgroups per tile, meeting inside the launch)
    ("k3_splitk_all_linear", 2, (8, 8, 8), 128, 0, 128, 3, 1, False, "bias pro vec res"),
    ("k3_splitk_dual_partial", 1, (6, 6, 6), 96, 64, 72, 3, 1, False, "bias pro res"),
    ("k3_deepk_relu_split", 1, (8, 8, 8), 128, 0, 64, 3, 1, False, "bias relu res"),
    ("k3_upsample_splitk", 1, (4, 4, 4), 128, 0, 64, 3, 1, True, "bias res"),
    ("k3s2_8to4", 2, (8, 8, 8), 16, 0, 64, 3, 2, False, "bias"),
    ("k3s2_16to8", 1, (16, 16, 16), 32, 0, 32, 3, 2, False, "bias"),
    ("k3s2_4to2", 1, (4, 4, 4), 16, 0, 64, 3, 2, False, "bias"),
    ("k3_upsample_4to8", 2, (4, 4, 4), 32, 0, 32, 3, 1, True, "bias"),
    ("k3_upsample_8to16", 1, (8, 8, 8), 16, 0, 64, 3, 1, True, "bias"),
    ("k3_upsample_noncubic_res", 2, (4, 8, 8), 32, 0, 40, 3, 1, True, "bias res"),
    ("k3_upsample_2to4", 1, (2, 2, 2), 16, 0, 64, 3, 1, True, ""),
    ("k3s2_odd_mixed", 1, (5, 6, 8), 16, 0, 32, 3, 2, False, "bias"),
    ("k1_dual", 2, (8, 8, 8), 32, 16, 64, 1, 1, False, "bias"),
    ("k1_relu_res", 1, (4, 4, 4), 64, 0, 64, 1, 1, False, "bias relu res"),
]


@pytest.mark.parametrize("prec", ["fp32", "h3"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d(dev, case, prec):
    from dm3d_amd import ops, _lib
    name, B, (D, H, W), c1, c2, cout, ks, stride, ups, extras = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    rnd = lambda *s: torch.randn(*s, generator=g)
    cin = c1 + c2
    x1 = rnd(B, D, H, W, c1)
    x2 = rnd(B, D, H, W, c2) if c2 else None
    kern = rnd(ks, ks, ks, cin, cout) / math.sqrt(cin * ks ** 3)
    bias = rnd(cout) if "bias" in extras else None
    pro = (torch.rand(cin, generator=g) + 0.5, rnd(cin) * 0.1) if "pro" in extras else None
    up = 2 if ups else 1
    od, oh, ow = (-(-D * up // stride), -(-H * up // stride), -(-W * up // stride))
    vec = rnd(5, cout + 8) if "vec" in extras else None
    vec_idx = torch.tensor([4, 0, 2][:B], dtype=torch.int32) if "vecidx" in extras else None
    res = rnd(B, od, oh, ow, cout) if "res" in extras else None
    ref = _conv_ref(x1, kern, bias, x2=x2, stride=stride, upsample=ups, pro=pro, vec=vec, vec_idx=vec_idx,
                    relu="relu" in extras, res=res)
    c = lambda t: None if t is None else t.to(dev).contiguous()
    if prec == "h3":
        wpk, w_exp = ops.pack_weights_up(c(kern), h3=True) if ups else ops.pack_weights_h3(c(kern), stride=stride)
        pk = dict(precision=_lib.PREC_H3, w_exp=w_exp)
    else:
        wpk, pk = (ops.pack_weights_up(c(kern)) if ups else ops.pack_weights(c(kern))), {}
    out = ops.conv3d(c(x1), wpk, cout, ks, x2=c(x2), bias=c(bias), stride=stride, upsample=ups,
                     pro_scale=c(pro[0]) if pro else None, pro_shift=c(pro[1]) if pro else None,
                     vec=c(vec), vec_idx=c(vec_idx), relu="relu" in extras, res=c(res), **pk)
    torch.cuda.synchronize()
    assert tuple(out.shape) == tuple(ref.shape)
    err = _rel(out, ref)
    print(f"{name}[{prec}] rel err {err:.2e}")
    assert err < 2e-5, f"{name}: rel err {err:.3e}"


def _random_conv_cases(n, seed):
    """Seeded sweep over ragged extents, channel counts (multiples of 4; c1 % 16 == 0 with a second input), kernel kinds and
    epilogue options: the shapes nobody thought of when writing CONV_CASES."""
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        kind = rng.choice(["k3", "k3", "k3", "k3s2", "up", "k1"])
        dims = tuple(int(v) for v in rng.integers(1, 13, size=3))
        dual = kind in ("k3", "k1") and rng.random() < 0.35
        c1 = int(rng.choice([16, 32, 48, 96, 144])) if dual else int(rng.choice([4, 8, 12, 16, 20, 36, 64, 100, 132, 160]))
        c2 = int(rng.choice([4, 12, 16, 40])) if dual else 0
        cout = int(rng.choice([4, 8, 24, 32, 40, 64, 72, 130]))
        opts = [o for o in ("bias", "pro", "vec", "res", "relu") if rng.random() < 0.5]
        if "vec" in opts and rng.random() < 0.5:
            opts.append("vecidx")
        ks, stride, ups = (1, 1, False) if kind == "k1" else (3, 2 if kind == "k3s2" else 1, kind == "up")
        if kind == "k1":
            opts = [o for o in opts if o != "pro" or True]
        cases.append((f"rand{i}_{kind}_{dims[0]}x{dims[1]}x{dims[2]}_{c1}+{c2}to{cout}_{'-'.join(opts) or 'plain'}", int(rng.integers(1, 4)),
                      dims, c1, c2, cout, ks, stride, ups, " ".join(opts)))
    return cases


RANDOM_CONV_CASES = _random_conv_cases(36, 20260101)


@pytest.mark.parametrize("prec", ["fp32", "h3"])
@pytest.mark.parametrize("case", RANDOM_CONV_CASES, ids=[c[0] for c in RANDOM_CONV_CASES])
def test_conv3d_random_shapes(dev, case, prec):
    test_conv3d(dev, case, prec)


@pytest.mark.parametrize("case", [
    # name, B, dims, main cin, skip c1, skip c2, cout, extras
    ("skip_single", 2, (8, 8, 8), 64, 32, 0, 64, "bias pro vec"),
    ("skip_dual_ragged", 1, (6, 5, 7), 32, 48, 20, 72, "bias pro"),
    ("skip_deep_splitk", 1, (8, 8, 8), 128, 256, 128, 128, "bias pro vec"),     # small grid: part 0 of the split carries the skip
    ("skip_odd_chunks", 2, (4, 8, 8), 16, 16, 0, 64, ""),                       # 16 skip channels: one real + one zero chunk
], ids=lambda c: c[0])
def test_conv3d_fused_skip(dev, case):
    """ResidualBlock tail in one launch: conv_k3(silu(bn(h))) + bias + temb + Conv3D(width, 1)(concat(x, skip)) against the two
    separate float64 convolutions."""
    from dm3d_amd import ops, _lib
    name, B, (D, H, W), cm, s1, s2, cout, extras = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    rnd = lambda *s: torch.randn(*s, generator=g)
    h = rnd(B, D, H, W, cm)
    x1, x2 = rnd(B, D, H, W, s1), (rnd(B, D, H, W, s2) if s2 else None)
    kern = rnd(3, 3, 3, cm, cout) / math.sqrt(cm * 27)
    kskip = rnd(1, 1, 1, s1 + s2, cout) / math.sqrt(s1 + s2) * 3.0            # deliberately a different magnitude than kern
    bias = rnd(cout) if "bias" in extras else None
    pro = (torch.rand(cm, generator=g) + 0.5, rnd(cm) * 0.1) if "pro" in extras else None
    vec = rnd(5, cout + 8) if "vec" in extras else None
    ref = _conv_ref(h, kern, bias, pro=pro, vec=vec) + _conv_ref(x1, kskip, None, x2=x2)
    c = lambda t: None if t is None else t.to(dev).contiguous()
    w_exp = ops.h3_weight_exponent(kern, kskip)
    wpk, _ = ops.pack_weights_h3(c(kern), w_exp=w_exp)
    swpk = ops.pack_weights_skip_h3p(c(kskip), w_exp)
    out = ops.conv3d(c(h), wpk, cout, 3, bias=c(bias), pro_scale=c(pro[0]) if pro else None, pro_shift=c(pro[1]) if pro else None,
                     vec=c(vec), precision=_lib.PREC_H3, w_exp=w_exp, skip=(c(x1), c(x2), swpk))
    torch.cuda.synchronize()
    err = _rel(out, ref)
    print(f"{name}: rel err {err:.2e}")
    assert err < 2e-5, f"{name}: rel err {err:.3e}"


@pytest.mark.parametrize("dims,cm,cout", [((8, 8, 8), 64, 64), ((4, 16, 8), 32, 128)], ids=["cube", "noncubic_cout128"])
def test_conv3d_h2_handoff_between_two_convs(dev, dims, cm, cout):
    """ResidualBlock interior (conditional_dm3d.py:255-267): conv1 applies conv2's folded norm + swish in its epilogue and stores
    DM3D_FMT_H2; conv2 reads that format with no prologue.  Both halves against float64, and the pair against the unfused pair."""
    from dm3d_amd import ops, _lib
    g = torch.Generator().manual_seed(5)
    B, (D, H, W) = 2, dims
    x = torch.randn(B, D, H, W, cm, generator=g)
    k1 = torch.randn(3, 3, 3, cm, cout, generator=g) / math.sqrt(cm * 27)
    k2 = torch.randn(3, 3, 3, cout, cout, generator=g) / math.sqrt(cout * 27)
    b1, b2 = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    vec = torch.randn(B, cout, generator=g)
    s2, t2 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, D, H, W, cout, generator=g)
    from oracle import ref_torch as rt
    h_ref = _conv_ref(x, k1, b1, vec=vec)
    a_ref = rt._swish(h_ref * s2.double() + t2.double())
    y_ref = _conv_ref(a_ref, k2, b2, res=res)
    c = lambda t: t.to(dev).contiguous()
    w1, e1 = ops.pack_weights_h3(c(k1))
    w2, e2 = ops.pack_weights_h3(c(k2))
    H3 = dict(precision=_lib.PREC_H3)
    a_h2 = ops.conv3d(c(x), w1, cout, 3, bias=c(b1), vec=c(vec), w_exp=e1, post=(c(s2), c(t2)), out_h2=True, **H3)
    a_dec = ops.h2_to_f32(a_h2.reshape(-1, cout), cout).reshape(B, D, H, W, cout)
    assert _rel(a_dec, a_ref) < 2e-5
    y = ops.conv3d(a_h2, w2, cout, 3, bias=c(b2), res=c(res), w_exp=e2, x1_h2_channels=cout, **H3)
    assert _rel(y, y_ref) < 2e-5
    # same pair the old way: float32 hand-off, norm + swish in conv2's prologue
    h = ops.conv3d(c(x), w1, cout, 3, bias=c(b1), vec=c(vec), w_exp=e1, **H3)
    y_old = ops.conv3d(h, w2, cout, 3, bias=c(b2), res=c(res), w_exp=e2, pro_scale=c(s2), pro_shift=c(t2), **H3)
    torch.cuda.synchronize()
    assert float((y - y_old).abs().max() / y_old.abs().max()) < 2e-6
    with pytest.raises(Exception):                          # ragged extent: the H2 output form is refused, not silently wrong
        ops.conv3d(c(x)[:, :3], w1, cout, 3, w_exp=e1, out_h2=True, **H3)


def test_conv3d_h3_tap_layout_arm(dev):
    """DM3D_CONV_PAIR=0 (read when the library is loaded) routes every H3 conv to the 32x32x16 kernel with the DM3D_WL_TAP weight
    layout: the A/B arm of the 16x16x32 kernel must stay correct.  One child interpreter, same parity cases."""
    import os, subprocess, sys
    if os.environ.get("DM3D_CONV_PAIR") == "0":
        pytest.skip("already inside the tap-layout arm")
    env = dict(os.environ, DM3D_CONV_PAIR="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "(test_conv3d or random_shapes) and h3",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_conv3d_8_slice_forms_on_every_shape(dev):
    """DM3D_CONV_WIDE_WGS=1 DM3D_CONV_V3_TD=8 send every k3 / parity conv to the 8-slice forms — the 512-thread free-running loop, the
    float8 form, skip phase and parity convs included — whatever its grid: the small, ragged and odd shapes of the parity cases then run
    through the kernels the bench uses only on large grids.  One child interpreter."""
    import os, subprocess, sys
    if os.environ.get("DM3D_CONV_WIDE_WGS") == "1":
        pytest.skip("already inside the 8-slice arm")
    env = dict(os.environ, DM3D_CONV_WIDE_WGS="1", DM3D_CONV_V3_TD="8")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "(test_conv3d or random_shapes) and not tap_layout_arm and not 8_slice_forms",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_conv3d_weight_layout_is_checked(dev):
    from dm3d_amd import ops, _lib
    lib = _lib.lib()
    assert lib.dm3d_conv_weight_layout(3, 1, 0, 0, 64) in (_lib.WL_TAP, _lib.WL_PAIR)
    assert lib.dm3d_conv_weight_layout(3, 2, 0, 0, 64) == _lib.WL_TAP
    assert lib.dm3d_conv_weight_layout(1, 1, 0, 0, 64) == _lib.WL_TAP
    assert lib.dm3d_conv_weight_layout(3, 1, 0, 0, 8) in (_lib.WL_TAP, _lib.WL_PAIR)      # round 3: the narrow column forms of the 16x16x32 kernel
    x = torch.randn(1, 8, 8, 8, 16, device=dev)
    k = torch.randn(3, 3, 3, 16, 64, device=dev)
    wpk, w_exp = ops.pack_weights_h3(k, stride=2)            # stride-2 image ...
    if lib.dm3d_conv_weight_layout(3, 1, 0, 0, 64) == _lib.WL_PAIR:
        d = _lib.ConvDesc()
        out = torch.empty(1, 8, 8, 8, 64, device=dev)
        d.x1, d.c1, d.batch, d.in_d, d.in_h, d.in_w, d.ksize, d.stride = x.data_ptr(), 16, 1, 8, 8, 8, 3, 1
        d.wpk, d.out, d.cout, d.precision, d.w_exp, d.w_layout = wpk.data_ptr(), out.data_ptr(), 64, _lib.PREC_H3, w_exp, _lib.WL_TAP
        assert lib.dm3d_conv3d_ndhwc(C.byref(d), None) != 0     # ... handed to a stride-1 conv: the layout tag does not match
        assert b"w_layout" in lib.dm3d_last_error()


def test_conv3d_h3_extreme_magnitudes(dev):
    """H3 split: tiny and huge activations / weights keep float32-grade relative accuracy (power-of-two weight scaling,
    float16 subnormal lo terms, clamp at 65504)."""
    from dm3d_amd import ops, _lib
    g = torch.Generator().manual_seed(99)
    for xs, ws in ((1e-3, 1e-4), (300.0, 30.0), (1.0, 1e3), (1e-2, 1.0)):
        x = torch.randn(1, 8, 8, 8, 32, generator=g) * xs
        k = torch.randn(3, 3, 3, 32, 64, generator=g) * ws
        ref = _conv_ref(x, k, None)
        wpk, w_exp = ops.pack_weights_h3(k.to(dev))
        out = ops.conv3d(x.to(dev), wpk, 64, 3, precision=_lib.PREC_H3, w_exp=w_exp)
        torch.cuda.synchronize()
        err = _rel(out, ref)
        print(f"h3 magnitudes x~{xs} w~{ws}: rel err {err:.2e}")
        assert err < 2e-5


def test_conv3d_argument_errors(dev):
    from dm3d_amd import ops
    from dm3d_amd._lib import Dm3dError
    x = torch.zeros(1, 4, 4, 4, 6, device=dev)
    w = torch.zeros(27 * 64 * 16, device=dev)
    with pytest.raises(Dm3dError):
        ops.conv3d(x, w, 8, 3)                      # c1 % 4 != 0
    x = torch.zeros(1, 4, 4, 4, 8, device=dev)
    with pytest.raises(Dm3dError):
        ops.conv3d(x, w, 8, 2)                      # ksize 2
    with pytest.raises(Dm3dError):
        ops.conv3d(x, w, 8, 3, stride=2, upsample=True)


GEMM_CASES = [
    ("plain", dict(m=256, n=64, k=32)),
    ("guards", dict(m=300, n=70, k=36)),
    ("tiny_m", dict(m=2, n=200, k=128)),
    ("bias_relu_res", dict(m=130, n=96, k=64, bias=True, act=1, res=True)),
    ("bias_m_silu", dict(m=96, n=130, k=64, bias=True, bias_m=True, act=2)),
    ("alpha_batched", dict(m=64, n=64, k=48, batch=3, alpha=0.0625)),
    ("batched_broadcast_b", dict(m=64, n=40, k=64, batch=4, bcast_b=True, res=True)),
    ("long_k", dict(m=64, n=64, k=1024)),
]


@pytest.mark.parametrize("case", GEMM_CASES, ids=[c[0] for c in GEMM_CASES])
def test_gemm_tn(dev, case):
    from dm3d_amd import ops
    name, kw = case
    m, n, k, batch = kw["m"], kw["n"], kw["k"], kw.get("batch", 1)
    g = torch.Generator().manual_seed(len(name) * 7919)
    a = torch.randn(batch, m, k, generator=g)
    b = torch.randn(1 if kw.get("bcast_b") else batch, n, k, generator=g)
    bias = torch.randn(m if kw.get("bias_m") else n, generator=g) if kw.get("bias") else None
    res = torch.randn(batch, m, n, generator=g) if kw.get("res") else None
    alpha = kw.get("alpha", 1.0)
    ref = torch.einsum("bmk,bnk->bmn", a.double(), b.double().expand(batch, n, k)) * alpha
    if bias is not None:
        ref = ref + (bias.double()[None, :, None] if kw.get("bias_m") else bias.double())
    act = kw.get("act", 0)
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = ref * torch.sigmoid(ref)
    if res is not None:
        ref = ref + res.double()
    c = lambda t: None if t is None else t.to(dev).contiguous()
    out = ops.gemm_tn(c(a), c(b), m=m, n=n, k=k, batch=batch, stride_a=m * k, stride_b=0 if kw.get("bcast_b") else n * k,
                      alpha=alpha, bias=c(bias), bias_along_m=bool(kw.get("bias_m")), act=act, res=c(res),
                      out=torch.empty(batch, m, n, device=dev))
    torch.cuda.synchronize()
    err = _rel(out, ref)
    assert err < 2e-5, f"{name}: rel err {err:.3e}"


H3_GEMM_CASES = [
    ("h2h2_plain", dict(m=256, n=64, k=32)),
    ("h2h2_guards_ktail", dict(m=300, n=80, k=48, bias=True, act=1, res=True)),
    ("f32h2", dict(m=130, n=96, k=64, a_f32=True, bias=True)),
    ("h2f32_bias_m", dict(m=96, n=144, k=256, b_f32=True, bias=True, bias_m=True)),
    ("f32f32_long_k", dict(m=64, n=64, k=1024, a_f32=True, b_f32=True)),
    ("batched_alpha", dict(m=512, n=512, k=256, batch=3, alpha=0.0625)),
    ("batched_bcast_res", dict(m=64, n=48, k=64, batch=4, bcast_b=True, res=True)),
    ("out_h2", dict(m=300, n=96, k=64, out_h2=True, bias=True, act=1)),
    ("out_h2_res_silu", dict(m=256, n=64, k=128, out_h2=True, res=True, act=2, a_f32=True)),
]


@pytest.mark.parametrize("case", H3_GEMM_CASES, ids=[c[0] for c in H3_GEMM_CASES])
def test_gemm_tn_h3(dev, case):
    """Split-float16 GEMM with every operand/output format combination, against float64."""
    from dm3d_amd import ops, _lib
    name, kw = case
    m, n, k, batch = kw["m"], kw["n"], kw["k"], kw.get("batch", 1)
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    a = torch.randn(batch, m, k, generator=g)
    b = torch.randn(1 if kw.get("bcast_b") else batch, n, k, generator=g)
    bias = torch.randn(m if kw.get("bias_m") else n, generator=g) if kw.get("bias") else None
    res = torch.randn(batch, m, n, generator=g) if kw.get("res") else None
    alpha = kw.get("alpha", 1.0)
    ref = torch.einsum("bmk,bnk->bmn", a.double(), b.double().expand(batch, n, k)) * alpha
    if bias is not None:
        ref = ref + (bias.double()[None, :, None] if kw.get("bias_m") else bias.double())
    act = kw.get("act", 0)
    ref = torch.relu(ref) if act == 1 else (ref * torch.sigmoid(ref) if act == 2 else ref)
    if res is not None:
        ref = ref + res.double()
    c = lambda t: None if t is None else t.to(dev).contiguous()
    ad, bd = c(a), c(b)
    a_fmt = b_fmt = _lib.FMT_H2
    if kw.get("a_f32"):
        a_fmt = _lib.FMT_F32
    else:
        ad = ops.split_h2(ad.reshape(-1, k)).reshape(batch, m, k)
    if kw.get("b_f32"):
        b_fmt = _lib.FMT_F32
    else:
        bd = ops.split_h2(bd.reshape(-1, k)).reshape(bd.shape[0], n, k)
    out = ops.gemm_tn(ad, bd, m=m, n=n, k=k, batch=batch, stride_a=m * k, stride_b=0 if kw.get("bcast_b") else n * k,
                      alpha=alpha, bias=c(bias), bias_along_m=bool(kw.get("bias_m")), act=act, res=c(res),
                      out=torch.empty(batch, m, n, device=dev), precision=_lib.PREC_H3, a_fmt=a_fmt, b_fmt=b_fmt,
                      out_fmt=_lib.FMT_H2 if kw.get("out_h2") else _lib.FMT_F32)
    torch.cuda.synchronize()
    if kw.get("out_h2"):
        out = ops.h2_to_f32(out.reshape(batch * m, n), n).reshape(batch, m, n)
    err = _rel(out, ref)
    print(f"gemm_h3 {name}: rel err {err:.2e}")
    assert err < 2e-5, f"{name}: rel err {err:.3e}"


def _random_h3_gemm_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        kw = dict(m=int(rng.integers(1, 400)), n=int(rng.integers(1, 20)) * 16, k=int(rng.integers(1, 24)) * 16)
        if rng.random() < 0.3:
            kw["batch"] = int(rng.integers(2, 5))
            kw["bcast_b"] = bool(rng.random() < 0.5)
        for opt, p in (("a_f32", 0.3), ("b_f32", 0.3), ("bias", 0.5), ("res", 0.5), ("out_h2", 0.4)):
            if rng.random() < p:
                kw[opt] = True
        if kw.get("bias") and rng.random() < 0.3:
            kw["bias_m"] = True
        kw["act"] = int(rng.integers(0, 3))
        if rng.random() < 0.3:
            kw["alpha"] = float(rng.choice([0.0625, 0.5, 2.0]))
        cases.append((f"rand{i}_" + "_".join(f"{k}{v if not isinstance(v, bool) else ''}" for k, v in kw.items()), kw))
    return cases


RANDOM_H3_GEMM = _random_h3_gemm_cases(24, 20260102)


@pytest.mark.parametrize("case", RANDOM_H3_GEMM, ids=[c[0] for c in RANDOM_H3_GEMM])
def test_gemm_tn_h3_random_shapes(dev, case):
    test_gemm_tn_h3(dev, case)


def test_h2_format_roundtrip_and_norm_kernels(dev):
    from dm3d_amd import ops
    g = torch.Generator().manual_seed(77)
    x = torch.randn(37, 40, generator=g) * 3
    h2 = ops.split_h2(x.to(dev))
    assert tuple(h2.shape) == (37, 48)
    assert _rel(ops.h2_to_f32(h2, 40), x) < 1e-6
    assert float(ops.h2_to_f32(h2, 48)[:, 40:].abs().max()) == 0.0           # zero padded to the record
    assert _rel(ops.h2_to_f32(ops.split_h2(x.to(dev), exp2=5), 40), x * 32) < 1e-6
    # LayerNorm with H2 outputs == float32 LayerNorm
    for rows, c in ((33, 256), (5, 48), (3, 1024)):
        y = torch.randn(rows, c, generator=g) * 2 + 0.3
        params = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)) for _ in range(3)]
        pd = [(a.to(dev), b.to(dev)) for a, b in params]
        outs = ops.layernorm3_h2(y.to(dev), pd, eps=1e-3)
        for (ga, be), o in zip(params, outs):
            ref = torch.nn.functional.layer_norm(y.double(), (c,), ga.double(), be.double(), 1e-3)
            assert _rel(ops.h2_to_f32(o, c), ref) < 3e-6
    # softmax left in place in H2
    for rows, cols in ((9, 512), (130, 64), (2, 1024), (5, 16), (7, 4096), (3, 1040), (2, 8208)):     # > 1024: streaming form
        sc = torch.randn(rows, cols, generator=g) * 4
        out = ops.softmax_rows_h2_(sc.to(dev).clone())
        torch.cuda.synchronize()
        assert float((ops.h2_to_f32(out, cols).double().cpu() - torch.softmax(sc.double(), -1)).abs().max()) < 2e-6


def test_gemm_strided_views(dev):
    """q|k packed in one [M, 2u] buffer, scores = q k^T per sample (the layout the attention blocks use)."""
    from dm3d_amd import ops
    g = torch.Generator().manual_seed(5)
    B, L, u = 2, 64, 32
    qk = torch.randn(B * L, 2 * u, generator=g)
    ref = torch.einsum("blc,bmc->blm", qk[:, :u].reshape(B, L, u).double(), qk[:, u:].reshape(B, L, u).double())
    qkd = qk.to(dev)
    out = torch.empty(B, L, L, device=dev)
    from dm3d_amd._lib import GemmDesc, lib, check
    d = GemmDesc()
    d.a, d.lda, d.stride_a = qkd.data_ptr(), 2 * u, L * 2 * u
    d.b, d.ldb, d.stride_b = qkd.data_ptr() + 4 * u, 2 * u, L * 2 * u
    d.out, d.ldo, d.stride_o = out.data_ptr(), L, L * L
    d.m, d.n, d.k, d.batch, d.alpha = L, L, u, B, 1.0
    check(lib().dm3d_gemm_tn(C.byref(d), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert _rel(out, ref) < 2e-5


@pytest.mark.parametrize("mode", ["fp32", "h3_f32", "h3_h2"])
def test_attention_entry(dev, mode):
    """dm3d_attention (score product + row softmax + P.V in one call) against the oracle's _attention, self- and cross-
    (one context's keys/values broadcast to the batch) forms, with a residual."""
    from dm3d_amd import ops, _lib
    from oracle import ref_torch as rt
    g = torch.Generator().manual_seed(77)
    B, L, u = 3, 64, 32
    q = torch.randn(B, L, u, generator=g)
    res = torch.randn(B, L, u, generator=g)
    prec = _lib.PREC_F32 if mode == "fp32" else _lib.PREC_H3
    fmt = _lib.FMT_H2 if mode == "h3_h2" else _lib.FMT_F32
    enc = (lambda t: ops.split_h2(t.to(dev).contiguous()).view(*t.shape[:-1], -1)) if mode == "h3_h2" else (lambda t: t.to(dev).contiguous())
    for kb in (B, 1):                                       # per-sample keys, then one broadcast context
        k = torch.randn(kb, 48, u, generator=g)
        v = torch.randn(kb, 48, u, generator=g)
        ref = rt._attention(q.double(), k.double().expand(B, -1, -1), v.double().expand(B, -1, -1), u) + res.double()
        out = ops.attention(enc(q), enc(k), enc(v.transpose(1, 2).contiguous()), float(u) ** -0.5, res=res.to(dev), precision=prec, fmt=fmt)
        torch.cuda.synchronize()
        assert _rel(out, ref) < 2e-5, (mode, kb)


@pytest.mark.parametrize("B,Lq,Lk,kb,scale_mul", [(2, 512, 512, 2, 1.0), (3, 128, 96, 1, 1.0), (1, 256, 512, 1, 40.0)],
                         ids=["L512_self", "q128_k96_broadcast", "peaked_softmax"])
def test_attention_fused_kernel(dev, B, Lq, Lk, kb, scale_mul):
    """The fused form of dm3d_attention (csrc/dm3d_attn_h3.hip: units 256, H2 operands, lq % 128 == 0, lk % 32 == 0): scores, online
    softmax over 32-key tiles and P.V in one launch, without scratch, against the float64 oracle and against the three-launch form.
    "peaked_softmax": a large logit scale makes the running maximum move from tile to tile (the accumulator rescaling path)."""
    import ctypes as C
    from dm3d_amd import ops, _lib
    from dm3d_amd._lib import lib, check
    from oracle import ref_torch as rt
    u = 256
    g = torch.Generator().manual_seed(Lq + Lk)
    q = torch.randn(B, Lq, u, generator=g) * 0.7
    k = torch.randn(kb, Lk, u, generator=g) * 0.7
    v = torch.randn(kb, Lk, u, generator=g)
    res = torch.randn(B, Lq, u, generator=g)
    scale = scale_mul * float(u) ** -0.5
    s = torch.einsum("blc,bLc->blL", q.double(), k.double().expand(B, -1, -1)) * scale
    ref = torch.einsum("blL,bLc->blc", torch.softmax(s, -1), v.double().expand(B, -1, -1)) + res.double()
    enc = lambda t: ops.split_h2(t.to(dev).contiguous()).view(*t.shape[:-1], -1)
    qh, kh, vth = enc(q), enc(k), enc(v.transpose(1, 2).contiguous())
    out = torch.empty(B, Lq, u, device=dev)
    d = _lib.AttentionDesc()
    d.q, d.ldq = qh.data_ptr(), u
    d.k, d.ldk, d.stride_k = kh.data_ptr(), u, (Lk * u if kb == B and B > 1 else 0)
    d.vt, d.ldv, d.stride_vt = vth.data_ptr(), Lk, (u * Lk if kb == B and B > 1 else 0)
    d.out, d.ldo, d.res = out.data_ptr(), u, res.to(dev).data_ptr()
    resd = res.to(dev)
    d.res = resd.data_ptr()
    d.batch, d.lq, d.lk, d.c, d.scale, d.precision, d.fmt = B, Lq, Lk, u, scale, _lib.PREC_H3, _lib.FMT_H2
    check(lib().dm3d_attention(C.byref(d), None, torch.cuda.current_stream().cuda_stream), "attention")     # no scratch: must be the fused kernel
    torch.cuda.synchronize()
    err = _rel(out, ref)
    print(f"fused attention rel err {err:.2e}")
    assert err < 2e-5
    # the three-launch form on the same operands (ops.attention picks it for F32-format operands)
    out3 = ops.attention(q.to(dev), k.to(dev), v.transpose(1, 2).contiguous().to(dev), scale, res=resd, precision=_lib.PREC_H3, fmt=_lib.FMT_F32)
    torch.cuda.synchronize()
    assert _rel(out, out3.cpu()) < 2e-5


@pytest.mark.parametrize("rows,c", [(37, 256), (5, 48), (1024, 256), (3, 1024)])
def test_layernorm3(dev, rows, c):
    from dm3d_amd import ops
    g = torch.Generator().manual_seed(rows * c)
    x = torch.randn(rows, c, generator=g) * 2 + 0.3
    params = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)) for _ in range(3)]
    outs = ops.layernorm3(x.to(dev), [(a.to(dev), b.to(dev)) for a, b in params], eps=1e-3)
    torch.cuda.synchronize()
    for (ga, be), o in zip(params, outs):
        ref = torch.nn.functional.layer_norm(x.double(), (c,), ga.double(), be.double(), 1e-3)
        assert _rel(o, ref) < 3e-6
    only2 = ops.layernorm3(x.to(dev), [(a.to(dev), b.to(dev)) for a, b in params[:2]], eps=1e-3)
    assert len(only2) == 2 and _rel(only2[1], torch.nn.functional.layer_norm(
        x.double(), (c,), params[1][0].double(), params[1][1].double(), 1e-3)) < 3e-6


@pytest.mark.parametrize("rows,cols", [(9, 512), (130, 64), (7, 100), (4, 8), (3, 1500), (2, 1024)])
def test_softmax_rows(dev, rows, cols):
    from dm3d_amd import ops
    g = torch.Generator().manual_seed(cols)
    s = torch.randn(rows, cols, generator=g) * 4
    s[0, 0] = 30.0                                  # a dominant logit
    out = ops.softmax_rows_(s.to(dev).clone())
    torch.cuda.synchronize()
    ref = torch.softmax(s.double(), -1)
    assert float((out.double().cpu() - ref).abs().max()) < 2e-6
    assert torch.allclose(out.sum(-1).cpu(), torch.ones(rows), atol=1e-5)


def test_affine_act(dev):
    from dm3d_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(33, 64, generator=g) * 3
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    y = x.double() * sc.double() + sh.double()
    for act, ref in ((0, y), (1, torch.relu(y)), (2, y * torch.sigmoid(y))):
        out = ops.affine_act(x.to(dev), sc.to(dev), sh.to(dev), act)
        assert _rel(out, ref) < 1e-6
    assert _rel(ops.affine_act(x.to(dev), None, None, 2), x.double() * torch.sigmoid(x.double())) < 1e-6


def _model(dev, T=50, S=8, Cc=4, weights=None):
    from types import SimpleNamespace
    from dm3d_amd.networks import conditional_dm3d as cdm
    return cdm.DiffusionModel(S, 1024, Cc, None, SimpleNamespace(timesteps=T, num_gpus=1, kernel_resize=False, bs=1),
                              weights=weights)


def test_ddpm_sample_matches_oracle(dev):
    """DiffusionModel.sample (conditional_dm3d.py:517-548) incl. the t=0 variance known answer."""
    from oracle import ref_torch as rt
    m = _model(dev, T=1000)
    g = torch.Generator().manual_seed(11)
    x, e = torch.randn(4, 8, 8, 8, 4, generator=g), torch.randn(4, 8, 8, 8, 4, generator=g)
    t = torch.tensor([0, 1, 500, 999])
    mean, var = m.sample(x.to(dev), e.to(dev), t, x.shape)
    rm, rv = rt.ddpm_sample(rt.Betas(1000), x, e, t)
    assert tuple(var.shape) == (4, 1, 1, 1, 1)
    assert float((mean.cpu() - rm).abs().max() / rm.abs().max()) < 1e-6
    assert float((var.cpu() - rv).abs().max()) < 1e-9
    assert float(var[0]) == 0.0                      # alpha_bar_prev[0] == 1 -> posterior variance 0 at t=0


def test_ddpm_step_kernel(dev):
    from dm3d_amd._lib import lib, check
    from oracle import ref_torch as rt
    m = _model(dev, T=1000)
    g = torch.Generator().manual_seed(12)
    x, e, z = (torch.randn(3, 8, 8, 8, 4, generator=g) for _ in range(3))
    t = torch.tensor([0, 7, 999])
    ref = rt.ddpm_step(rt.Betas(1000), x, e, t, z * (t > 0).float().reshape(3, 1, 1, 1, 1))
    xd, td = x.to(dev).clone(), t.to(torch.int32).to(dev)
    d = m._ddpm_desc(xd, e.to(dev), td, 1, noise=z.to(dev))
    check(lib().dm3d_ddpm_update(C.byref(d), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert float((xd.cpu() - ref).abs().max()) < 2e-6
    # in-kernel Philox: deterministic per (seed, t), zero noise at t == 0, unit variance elsewhere
    big = torch.zeros(2, 32, 32, 32, 8, device=dev)
    outs = []
    for seed in (1, 1, 2):
        xb = big.clone()
        d = m._ddpm_desc(xb, torch.zeros_like(xb), torch.tensor([0, 500], dtype=torch.int32, device=dev), 1, seed=seed)
        check(lib().dm3d_ddpm_update(C.byref(d), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        outs.append(xb.cpu())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0][1], outs[2][1])
    assert float(outs[0][0].abs().max()) == 0.0
    b = rt.Betas(1000)
    sigma = math.sqrt(float((1 - b.alpha_bar_prev[500]) * b.beta[500] / (1 - b.alpha_bar[500])))
    zs = outs[0][1] / sigma
    assert abs(float(zs.mean())) < 0.01 and abs(float(zs.std()) - 1) < 0.01


def test_randn_statistics(dev):
    from dm3d_amd import ops
    x = ops.randn((1 << 20,), seed=7, stream_id=3, device=dev).cpu().double()
    assert abs(float(x.mean())) < 5e-3 and abs(float(x.std()) - 1) < 5e-3
    assert abs(float((x ** 3).mean())) < 2e-2 and abs(float((x ** 4).mean()) - 3) < 5e-2
    assert torch.equal(ops.randn((4096,), 7, 3, dev), ops.randn((4096,), 7, 3, dev))
    assert not torch.equal(ops.randn((4096,), 7, 3, dev), ops.randn((4096,), 8, 3, dev))


def test_gather_add_graph(dev):
    from dm3d_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    tab = torch.arange(5 * 8, dtype=torch.float32, device=dev).reshape(5, 8)
    idx = torch.tensor([4, 0, 3], dtype=torch.int32, device=dev)
    out = torch.empty(3, 8, device=dev)
    check(lib().dm3d_gather_rows(tab.data_ptr(), 5, idx.data_ptr(), out.data_ptr(), 3, 8, st))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), tab.cpu()[[4, 0, 3]])
    cnt = torch.full((3,), 10, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    cap = torch.cuda.Stream()
    g = C.c_void_p()
    check(lib().dm3d_graph_begin(cap.cuda_stream))
    check(lib().dm3d_add_i32(cnt.data_ptr(), 3, -1, cap.cuda_stream))
    check(lib().dm3d_graph_end(cap.cuda_stream, C.byref(g)))
    for _ in range(4):
        check(lib().dm3d_graph_launch(g, st))
    torch.cuda.synchronize()
    assert cnt.cpu().tolist() == [6, 6, 6]
    check(lib().dm3d_graph_destroy(g))


def test_conv3d_h2_handoff_on_a_grid_of_half_8_slice_bricks(dev):
    """D = 12: whole 4-slice bricks (what the hand-off output needs) but not whole 8-slice bricks, on a launch large enough for the 8-slice
    forms (B = 16: 512 workgroups).  The fused output (norm + swish + DM3D_FMT_H2 store) lives in the full-brick epilogue, so such a launch
    must stay on the 4-slice form — it once went to the 8-slice kernel, whose partial bricks stored plain float32 into the H2 tensor."""
    from dm3d_amd import ops, _lib
    from oracle import ref_torch as rt
    g = torch.Generator().manual_seed(13)
    B, dims, cm, cout = 16, (12, 32, 32), 32, 64
    x = torch.randn(B, *dims, cm, generator=g)
    k1 = torch.randn(3, 3, 3, cm, cout, generator=g) / math.sqrt(cm * 27)
    b1 = torch.randn(cout, generator=g)
    s2, t2 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    a_ref = rt._swish(_conv_ref(x, k1, b1) * s2.double() + t2.double())
    c = lambda t: t.to(dev).contiguous()
    w1, e1 = ops.pack_weights_h3(c(k1))
    a_h2 = ops.conv3d(c(x), w1, cout, 3, bias=c(b1), w_exp=e1, post=(c(s2), c(t2)), out_h2=True, precision=_lib.PREC_H3)
    a_dec = ops.h2_to_f32(a_h2.reshape(-1, cout), cout).reshape(B, *dims, cout)
    torch.cuda.synchronize()
    assert _rel(a_dec, a_ref) < 2e-5

"""The Winograd F(2,3)-along-x form of the k3 / stride-1 split-float16 Conv3d (dm3d_conv_h3w.hip, dm3d_conv_desc.wpk_wino) against a float64
reference of the same op (Conv3D(padding="same") behind the folded norm + swish, conditional_dm3d.py:254-268) and against the direct kernel on
the same inputs: plain / prologue / concat / ragged Cin / non-cubic volumes / residual, the DM3D_FMT_H2 hand-off pair (kernel MODE 2 on the
consumer side), the launch policy (dm3d_conv_tile_form() == 10), and eps of the whole U-Net with and without the second image.
Tolerance: max|err| / max|ref| <= 2e-5 like every other contraction test (1e-3 is the end-to-end budget north_star states).
All through the C ABI (ctypes)."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from dm3d_amd import _lib
    _lib.require_device()
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


@pytest.fixture()
def small_grids(monkeypatch):
    """The policy keeps small launches on the direct kernel; the knobs (read per call) admit the test shapes."""
    monkeypatch.setenv("DM3D_CONV_WIDE_WGS", "1")
    monkeypatch.setenv("DM3D_CONV_WINO_MINCHUNKS", "1")


def _ref_conv(x, k, bias=None, pro=None, res=None):
    xd = x.double()
    if pro is not None:
        xd = xd * pro[0].double() + pro[1].double()
        xd = xd * torch.sigmoid(xd)
    y = F.conv3d(xd.permute(0, 4, 1, 2, 3), k.double().permute(4, 3, 0, 1, 2), padding=1).permute(0, 2, 3, 4, 1)
    if bias is not None:
        y = y + bias.double()
    if res is not None:
        y = y + res.double()
    return y


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


CASES = [("plain 8^3 16->64", 1, (8, 8, 8), 16, 0, 64, 0, 0), ("plain 8^3 48->64 (three chunks)", 1, (8, 8, 8), 48, 0, 64, 0, 0),
         ("prologue 8^3 16->64", 1, (8, 8, 8), 16, 0, 64, 1, 0), ("prologue 8^3 32->64 + residual, B=2", 2, (8, 8, 8), 32, 0, 64, 1, 1),
         ("prologue concat 16^3 64+32->128", 1, (16, 16, 16), 64, 32, 128, 1, 0), ("prologue 16^3 24->64 (ragged Cin)", 1, (16, 16, 16), 24, 0, 64, 1, 1),
         ("prologue 8x16x24 40->96 (ragged Cin and Cout)", 1, (8, 16, 24), 40, 0, 96, 1, 0)]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_winograd_conv_against_float64_and_the_direct_kernel(dev, small_grids, case):
    from dm3d_amd import ops, _lib
    name, B, dims, c1, c2, cout, pro, res = case
    torch.manual_seed(1)
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
    y_direct = ops.conv3d(x1, wpk, cout, 3, **kw)
    y_wino = ops.conv3d(x1, wpk, cout, 3, wpk_wino=wino, **kw)
    yr = _ref_conv(torch.cat([x1, x2], -1) if c2 else x1, k, bias, ps, r)
    e_d, e_w = _rel(y_direct, yr), _rel(y_wino, yr)
    print(f"{name}: direct {e_d:.2e}, Winograd {e_w:.2e}, Winograd - direct {_rel(y_wino, y_direct):.2e}")
    assert not torch.equal(y_wino, y_direct), "the Winograd form did not run (its results differ from the direct kernel's in the last bits)"
    assert e_w < 2e-5 and e_d < 2e-5


@pytest.mark.parametrize("dims", [(1, 8, 96, 128, 64), (1, 16, 128, 192, 128)], ids=["8^3 96->128->64", "16^3 128->192->128"])
def test_winograd_hand_off_pair(dev, small_grids, dims):
    """ResidualBlock conv1 -> BatchNormalization -> swish -> conv2 (conditional_dm3d.py:255-267) with the DM3D_FMT_H2 hand-off: conv A stores
    the consumer's activation split in float16 pairs, conv B (kernel MODE 2) rebuilds float32, transforms and splits again."""
    from dm3d_amd import ops, _lib
    B, e, c, cm, co = dims
    torch.manual_seed(2)
    x = torch.randn(B, e, e, e, c, device=dev)
    ka, kb = torch.randn(3, 3, 3, c, cm, device=dev) * 0.05, torch.randn(3, 3, 3, cm, co, device=dev) * 0.05
    wa, ea = ops.pack_weights_h3(ka)
    wb, eb = ops.pack_weights_h3(kb)
    wwa, wwb = ops.pack_weights_h3w(ka, ea), ops.pack_weights_h3w(kb, eb)
    post = (torch.rand(cm, device=dev) + 0.5, torch.randn(cm, device=dev) * 0.1)
    pro = (torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1)
    outs = []
    for wino in (False, True):
        a = ops.conv3d(x, wa, cm, 3, bias=torch.zeros(cm, device=dev), pro_scale=pro[0], pro_shift=pro[1], precision=_lib.PREC_H3, w_exp=ea,
                       post=post, out_h2=True, wpk_wino=wwa if wino else None)
        outs.append(ops.conv3d(a, wb, co, 3, precision=_lib.PREC_H3, w_exp=eb, x1_h2_channels=cm, wpk_wino=wwb if wino else None))
    mid = _ref_conv(x, ka, None, pro)
    mid = mid * post[0].double() + post[1].double()
    mid = mid * torch.sigmoid(mid)
    yr = _ref_conv(mid.float(), kb)
    assert not torch.equal(outs[0], outs[1])
    assert _rel(outs[0], yr) < 2e-5 and _rel(outs[1], yr) < 2e-5


SKIP_CASES = [("8^3 64->64 + k1(32)", 2, 8, 64, 32, 0, 64), ("16^3 128->128 + k1(64+32)", 1, 16, 128, 64, 32, 128), ("8^3 32->96 + k1(40), ragged", 1, 8, 32, 40, 0, 96)]


@pytest.mark.parametrize("case", SKIP_CASES, ids=[c[0] for c in SKIP_CASES])
def test_winograd_fused_skip_conv(dev, small_grids, case):
    """ResidualBlock tail in one launch: conv_k3(silu(bn(h))) + bias + Conv3D(width, 1)(concat(x, skip)) (conditional_dm3d.py:243-248, 268)
    with the 1x1 conv as the register-direct tail of the Winograd kernel (dm3d_conv_desc.skip_wpk_frag), on the transformed tiles."""
    from dm3d_amd import ops, _lib
    name, B, e, cm, s1, s2, cout = case
    torch.manual_seed(7)
    h = torch.randn(B, e, e, e, cm, device=dev)
    x1 = torch.randn(B, e, e, e, s1, device=dev)
    x2 = torch.randn(B, e, e, e, s2, device=dev) if s2 else None
    k = torch.randn(3, 3, 3, cm, cout, device=dev) * 0.05
    ks = torch.randn(1, 1, 1, s1 + s2, cout, device=dev) * 0.2
    w_exp = ops.h3_weight_exponent(k.cpu(), ks.cpu())
    wpk, _ = ops.pack_weights_h3(k, w_exp=w_exp)
    wino = ops.pack_weights_h3w(k, w_exp)
    swpk, sfrag = ops.pack_weights_skip_h3p(ks, w_exp), ops.pack_weights_skip_h3f(ks, w_exp)
    bias = torch.randn(cout, device=dev)
    ps = (torch.rand(cm, device=dev) + 0.5, torch.randn(cm, device=dev) * 0.1)
    kw = dict(bias=bias, pro_scale=ps[0], pro_shift=ps[1], precision=_lib.PREC_H3, w_exp=w_exp)
    y_direct = ops.conv3d(h, wpk, cout, 3, skip=(x1, x2, swpk), **kw)
    assert torch.equal(y_direct, ops.conv3d(h, wpk, cout, 3, wpk_wino=wino, skip=(x1, x2, swpk), **kw))      # without the fragment image: the direct kernel
    y_wino = ops.conv3d(h, wpk, cout, 3, wpk_wino=wino, skip=(x1, x2, swpk, sfrag), **kw)
    xs = torch.cat([x1, x2], -1) if s2 else x1
    yr = _ref_conv(h, k, bias, ps) + torch.einsum("bdhwc,co->bdhwo", xs.double(), ks.double()[0, 0, 0])
    assert not torch.equal(y_wino, y_direct), "the Winograd form did not run"
    assert _rel(y_wino, yr) < 2e-5 and _rel(y_direct, yr) < 2e-5


def test_winograd_cin_split_on_a_small_grid(dev, monkeypatch):
    """B = 32 at 8^3 is 128 workgroups of the one-per-CU form: it runs as two workgroups per brick, each contracting half of the chunks, the
    halves meeting inside the launch (the hand-over form of round 5: the part that draws the tile's last ticket sums both and runs the epilogue; Cin >= 256)."""
    from dm3d_amd import ops, _lib
    for v in ("DM3D_CONV_WIDE_WGS", "DM3D_CONV_WINO_MINCHUNKS", "DM3D_CONV_WINO", "DM3D_CONV_WINO_SPLIT"):
        monkeypatch.delenv(v, raising=False)
    torch.manual_seed(5)
    B, e, cin, cout = 32, 8, 256, 256
    x = torch.randn(B, e, e, e, cin, device=dev)
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
    wpk, w_exp = ops.pack_weights_h3(k)
    wino = ops.pack_weights_h3w(k, w_exp)
    ps = (torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1)
    r = torch.randn(B, e, e, e, cout, device=dev)
    kw = dict(bias=torch.randn(cout, device=dev), pro_scale=ps[0], pro_shift=ps[1], res=r, precision=_lib.PREC_H3, w_exp=w_exp)
    y_direct = ops.conv3d(x, wpk, cout, 3, **kw)
    y_wino = ops.conv3d(x, wpk, cout, 3, wpk_wino=wino, **kw)
    yr = _ref_conv(x[:2], k, kw["bias"], ps, r[:2])
    assert not torch.equal(y_wino, y_direct), "the Winograd form did not run"
    assert _rel(y_wino[:2], yr) < 2e-5 and _rel(y_wino, y_direct) < 2e-5
    # round 5: the halves meet inside the launch (hand-over form), so a non-linear epilogue splits like any other: ReLU before the residual
    y_a = ops.conv3d(x, wpk, cout, 3, relu=True, **kw)
    y_b = ops.conv3d(x, wpk, cout, 3, relu=True, wpk_wino=wino, **kw)
    yr2 = torch.relu(_ref_conv(x[:2], k, kw["bias"], ps)) + r[:2].double()
    assert not torch.equal(y_a, y_b), "the Winograd form did not run behind a ReLU epilogue"
    assert _rel(y_b[:2], yr2) < 2e-5 and _rel(y_a[:2], yr2) < 2e-5
    # ... and without the ticket words nothing splits: the direct kernel serves the launch, with or without the second image
    y_c = ops.conv3d(x, wpk, cout, 3, split=False, **kw)
    y_d = ops.conv3d(x, wpk, cout, 3, wpk_wino=wino, split=False, **kw)
    assert torch.equal(y_c, y_d) and _rel(y_c[:2], yr) < 2e-5


def test_winograd_launch_policy(dev, monkeypatch):
    """dm3d_conv_tile_form() names the Winograd form (10) only with the second image, whole 8x8x8 bricks, Cin >= 32, a large grid (or its Cin split) and, behind a
    fused skip conv, that conv's fragment image; DM3D_CONV_WINO=0 switches it off per call."""
    from dm3d_amd import _lib
    from dm3d_amd._lib import ConvDesc, lib
    for v in ("DM3D_CONV_WIDE_WGS", "DM3D_CONV_WINO_MINCHUNKS", "DM3D_CONV_WINO", "DM3D_CONV_V3_TD", "DM3D_CONV_WINO_SPLIT"):
        monkeypatch.delenv(v, raising=False)
    buf = torch.zeros(64, device=dev)

    def form(batch=32, e=32, c1=128, cout=64, wino=True, skip=False, ed=None, skip_c=64, counters=True):
        d = ConvDesc()
        if counters:                             # (the host's statement that it provides the split workspace: include/dm3d.h)
            d.split_counters, d.split_counter_words = buf.data_ptr(), 4096
        d.x1 = d.wpk = buf.data_ptr()
        d.out = buf.data_ptr() + 128
        d.c1, d.batch, d.in_d, d.in_h, d.in_w = c1, batch, ed or e, e, e
        d.ksize, d.stride, d.cout, d.precision, d.w_layout = 3, 1, cout, _lib.PREC_H3, _lib.WL_PAIR
        if wino:
            d.wpk_wino = buf.data_ptr()
        if skip:
            d.skip_wpk, d.skip_x1, d.skip_c1 = buf.data_ptr(), buf.data_ptr(), skip_c
            if skip != "no fragments":
                d.skip_wpk_frag = buf.data_ptr()
        return lib().dm3d_conv_tile_form(C.byref(d))

    assert form() == 10
    assert form(wino=False) == 8
    assert form(c1=16) == 8                      # one chunk: the prologue / epilogue of the one-workgroup-per-CU form do not pay
    assert form(c1=32) == 10
    assert form(batch=1) != 10                   # 64 workgroups
    assert form(batch=4) == 10 and form(batch=6) != 10       # exactly one workgroup per CU is a full round; 384 workgroups are a round and a half
    assert form(e=8, c1=256, cout=256) == 10     # 128 workgroups, Cin >= 256: two workgroups per brick (the Cin split)
    assert form(e=8, c1=128, cout=256) != 10     # too few chunks to split
    assert form(e=8, c1=256, cout=256, counters=False) != 10      # no ticket words: no split, and 128 workgroups are too few
    assert form(ed=36) != 10                     # not whole 8-slice bricks
    assert form(cout=32) != 10
    assert form(skip=True) == 10                 # a fused skip conv: its tail is register-direct, the launch persistent like the others
    assert form(c1=64, skip=True, skip_c=192) == 10 and form(c1=64, skip=True, skip_c=32) == 10
    assert form(skip="no fragments") == 4        # ... given the skip weights as operand fragments (skip_wpk_frag); else the direct kernel
    monkeypatch.setenv("DM3D_CONV_WINO", "0")
    assert form() == 8


def test_unet_eps_with_and_without_the_winograd_image(dev, monkeypatch):
    """eps of the conditional 32^3 U-Net at B = 32 (where the policy picks the Winograd form for the Cin >= 64 convs without a fused skip conv) against the same
    network on the direct kernel only: far inside the 1e-3 contract, and the plan does contain Winograd launches."""
    import dm3d_amd
    cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8, conditional=True)
    W = dm3d_amd.synthetic_weights(cfg, seed=0)
    torch.manual_seed(3)
    B = 32
    x = torch.randn(B, 32, 32, 32, 8, device=dev)
    t = torch.randint(0, 1000, (B,), device=dev, dtype=torch.int32)
    ctx = torch.randint(0, 2, (B,), device=dev, dtype=torch.int32)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("DM3D_CONV_WINO", flag)
        net = dm3d_amd.UNet(cfg, device=dev, weights=W, precision="h3")
        assert (net.wino is True) == (flag == "1")
        y = net([x, t, ctx])
        kinds = {k for plan in net._plans.values() for _, _, k, _ in plan.ops}
        assert ("conv_wino" in kinds) == (flag == "1"), sorted(kinds)
        outs.append(y.clone())
        del net
    err = _rel(outs[1], outs[0])
    print(f"eps, Winograd image on vs off: max rel diff {err:.2e}")
    assert not torch.equal(outs[0], outs[1]) and err < 2e-5

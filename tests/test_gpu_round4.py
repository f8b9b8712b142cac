"""Round-4 parity cases: the PERSISTENT Winograd-x conv — workgroups that walk a list of work items,
staging the next item's first image during the last chunk of the current one — against float64 at item lists of two to four items per
workgroup (column tile, Cin part, sample and brick all changing between items; uneven lists; one chunk and odd chunk counts; every kernel
MODE), the same kernel at real 32^3 shapes without the policy knobs, a NaN born INSIDE the U-Net, and the launch path of bench.py at four ranks.
All through the C ABI (ctypes); the oracle / float64 torch is the checker only."""
import json
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    from dm3d_amd import _lib
    _lib.require_device()
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


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


def _args(T, bs=1):
    return SimpleNamespace(timesteps=T, num_gpus=1, kernel_resize=False, bs=bs)


# name, B, dims, c1, c2, cout, prologue, residual, workgroups (DM3D_CONV_WINO_GRID), what the list exercises
PERSIST = [
    ("two bricks per workgroup, 16^3 32->64", 2, (16, 16, 16), 32, 0, 64, 1, 1, 8),
    ("one chunk, four items per workgroup, plain", 1, (16, 16, 16), 16, 0, 64, 0, 0, 2),
    ("three chunks (odd), column tile changes between items", 1, (16, 16, 16), 48, 0, 128, 1, 0, 8),
    ("uneven lists: 6 items on 4 workgroups", 3, (8, 8, 8), 32, 0, 128, 1, 1, 4),
    ("concat + ragged Cin, 12 items on 5 workgroups", 1, (8, 16, 24), 32, 24, 128, 1, 0, 5),
    ("sample changes between items (per-sample prologue rows are not used, temb-free)", 4, (8, 8, 8), 64, 0, 64, 1, 1, 2),
]


@pytest.mark.parametrize("case", PERSIST, ids=[c[0] for c in PERSIST])
def test_persistent_winograd_item_lists(dev, monkeypatch, case):
    """conv3d_igemm_h3w<MODE> with fewer workgroups than work items: every workgroup multiplies one item while it stages the next
    one's first image and weight steps, the epilogue of an item runs between two items' chunk loops (conditional_dm3d.py:254-268)."""
    from dm3d_amd import ops, _lib
    name, B, dims, c1, c2, cout, pro, res, grid = case
    monkeypatch.setenv("DM3D_CONV_WIDE_WGS", "1")
    monkeypatch.setenv("DM3D_CONV_WINO_MINCHUNKS", "1")
    torch.manual_seed(11)
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
    y_one = ops.conv3d(x1, wpk, cout, 3, wpk_wino=wino, **kw)                 # one item per workgroup
    y_direct = ops.conv3d(x1, wpk, cout, 3, **kw)
    monkeypatch.setenv("DM3D_CONV_WINO_GRID", str(grid))
    y_list = ops.conv3d(x1, wpk, cout, 3, wpk_wino=wino, **kw)                # the same launch as item lists
    torch.cuda.synchronize()
    yr = _ref_conv(torch.cat([x1, x2], -1) if c2 else x1, k, bias, ps, r)
    assert not torch.equal(y_one, y_direct), "the Winograd form did not run"
    assert torch.equal(y_list, y_one), "an item of a list must compute exactly what it computes alone"
    assert _rel(y_list, yr) < 2e-5


def test_persistent_winograd_cin_split_lists(dev, monkeypatch):
    """The two-way Cin split under item lists: 64 items (8 bricks x 4 column tiles x 2 Cin parts) on 16 workgroups — the chunk range, the column
    tile and the brick change between a workgroup's items, the halves meet inside the launch (8^3 level, Cin = 256)."""
    from dm3d_amd import ops, _lib
    for v in ("DM3D_CONV_WINO_MINCHUNKS", "DM3D_CONV_WINO", "DM3D_CONV_WINO_SPLIT"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("DM3D_CONV_WIDE_WGS", "1")                  # (the policy wants 256 items for the split form; this batch has 64)
    torch.manual_seed(5)
    B, e, cin, cout = 8, 8, 256, 256
    x = torch.randn(B, e, e, e, cin, device=dev)
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
    wpk, w_exp = ops.pack_weights_h3(k)
    wino = ops.pack_weights_h3w(k, w_exp)
    ps = (torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1)
    r = torch.randn(B, e, e, e, cout, device=dev)
    kw = dict(bias=torch.randn(cout, device=dev), pro_scale=ps[0], pro_shift=ps[1], res=r, precision=_lib.PREC_H3, w_exp=w_exp)
    y_direct = ops.conv3d(x, wpk, cout, 3, **kw)
    monkeypatch.setenv("DM3D_CONV_WINO_GRID", "16")
    y = ops.conv3d(x, wpk, cout, 3, wpk_wino=wino, **kw)
    torch.cuda.synchronize()
    yr = _ref_conv(x, k, kw["bias"], ps, r)
    assert not torch.equal(y, y_direct), "the Winograd form did not run"
    assert _rel(y, yr) < 2e-5


def test_persistent_winograd_hand_off_lists(dev, monkeypatch):
    """Kernel MODE 2 (input in DM3D_FMT_H2 behind conv1's fused norm + swish) and the hand-off OUTPUT form under item lists."""
    from dm3d_amd import ops, _lib
    monkeypatch.setenv("DM3D_CONV_WIDE_WGS", "1")
    monkeypatch.setenv("DM3D_CONV_WINO_MINCHUNKS", "1")
    B, e, c, cm, co = 2, 16, 64, 128, 64
    torch.manual_seed(2)
    x = torch.randn(B, e, e, e, c, device=dev)
    ka, kb = torch.randn(3, 3, 3, c, cm, device=dev) * 0.05, torch.randn(3, 3, 3, cm, co, device=dev) * 0.05
    wa, ea = ops.pack_weights_h3(ka)
    wb, eb = ops.pack_weights_h3(kb)
    wwa, wwb = ops.pack_weights_h3w(ka, ea), ops.pack_weights_h3w(kb, eb)
    post = (torch.rand(cm, device=dev) + 0.5, torch.randn(cm, device=dev) * 0.1)
    pro = (torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1)
    outs = []
    for grid in (None, "8"):
        if grid:
            monkeypatch.setenv("DM3D_CONV_WINO_GRID", grid)
        a = ops.conv3d(x, wa, cm, 3, bias=torch.zeros(cm, device=dev), pro_scale=pro[0], pro_shift=pro[1], precision=_lib.PREC_H3, w_exp=ea,
                       post=post, out_h2=True, wpk_wino=wwa)
        outs.append(ops.conv3d(a, wb, co, 3, precision=_lib.PREC_H3, w_exp=eb, x1_h2_channels=cm, wpk_wino=wwb))
    mid = _ref_conv(x, ka, None, pro)
    mid = mid * post[0].double() + post[1].double()
    mid = mid * torch.sigmoid(mid)
    yr = _ref_conv(mid.float(), kb)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert _rel(outs[1], yr) < 2e-5


SKIPL = [("8^3 64->64 + k1(32), 4 items on 2 workgroups", 4, 8, 64, 32, 0, 64, 2), ("16^3 128->128 + k1(64+32): 16 items on 8 workgroups", 1, 16, 128, 64, 32, 128, 8),
         ("8^3 32->96 + k1(40), ragged: 6 items on 4 workgroups", 3, 8, 32, 40, 0, 96, 4), ("16^3 64->64 + k1(192): six pairs, 8 items on 4 workgroups", 1, 16, 64, 128, 64, 64, 4)]


@pytest.mark.parametrize("case", SKIPL, ids=[c[0] for c in SKIPL])
def test_persistent_winograd_skip_tail_lists(dev, monkeypatch, case):
    """conv2(silu(bn(h))) + Conv3D(width, 1)(concat(x, skip)) (conditional_dm3d.py:243-248, 268) with the 1x1 conv as the Winograd kernel's
    register-direct tail: the skip products accumulate into the transformed tiles between an item's chunk loop and its epilogue, without LDS —
    so the launch walks item lists like the others.  Odd pair counts, ragged channel ends, a concat boundary inside a pair."""
    from dm3d_amd import ops, _lib
    name, B, e, cm, s1, s2, cout, grid = case
    monkeypatch.setenv("DM3D_CONV_WIDE_WGS", "1")
    monkeypatch.setenv("DM3D_CONV_WINO_MINCHUNKS", "1")
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
    r = torch.randn(B, e, e, e, cout, device=dev)
    kw = dict(bias=bias, pro_scale=ps[0], pro_shift=ps[1], res=r, precision=_lib.PREC_H3, w_exp=w_exp)
    y_direct = ops.conv3d(h, wpk, cout, 3, skip=(x1, x2, swpk), **kw)
    y_one = ops.conv3d(h, wpk, cout, 3, wpk_wino=wino, skip=(x1, x2, swpk, sfrag), **kw)
    monkeypatch.setenv("DM3D_CONV_WINO_GRID", str(grid))
    y_list = ops.conv3d(h, wpk, cout, 3, wpk_wino=wino, skip=(x1, x2, swpk, sfrag), **kw)
    torch.cuda.synchronize()
    xs = torch.cat([x1, x2], -1) if s2 else x1
    yr = _ref_conv(h, k, bias, ps, r) + torch.einsum("bdhwc,co->bdhwo", xs.double(), ks.double()[0, 0, 0])
    assert not torch.equal(y_one, y_direct), "the Winograd form did not run"
    assert torch.equal(y_list, y_one)
    assert _rel(y_list, yr) < 2e-5 and _rel(y_direct, yr) < 2e-5


def test_winograd_cin_split_with_a_skip_tail(dev, monkeypatch):
    """The 8^3-level conv2 + skip launches (256 -> 256 + k1(384)): two workgroups per brick and column tile, each contracting half of the chunks
    AND half of the skip conv's pairs (an odd number of pairs here: 12 + 1 ragged), the halves meeting inside the launch (hand-over form)."""
    from dm3d_amd import ops, _lib
    for v in ("DM3D_CONV_WINO_MINCHUNKS", "DM3D_CONV_WINO", "DM3D_CONV_WINO_SPLIT"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("DM3D_CONV_WIDE_WGS", "1")
    monkeypatch.setenv("DM3D_CONV_WINO_GRID", "24")
    torch.manual_seed(9)
    B, e, cm, s1, s2, cout = 4, 8, 256, 256, 136, 256
    h = torch.randn(B, e, e, e, cm, device=dev)
    x1, x2 = torch.randn(B, e, e, e, s1, device=dev), torch.randn(B, e, e, e, s2, device=dev)
    k = torch.randn(3, 3, 3, cm, cout, device=dev) * 0.05
    ks = torch.randn(1, 1, 1, s1 + s2, cout, device=dev) * 0.1
    w_exp = ops.h3_weight_exponent(k.cpu(), ks.cpu())
    wpk, _ = ops.pack_weights_h3(k, w_exp=w_exp)
    wino = ops.pack_weights_h3w(k, w_exp)
    swpk, sfrag = ops.pack_weights_skip_h3p(ks, w_exp), ops.pack_weights_skip_h3f(ks, w_exp)
    ps = (torch.rand(cm, device=dev) + 0.5, torch.randn(cm, device=dev) * 0.1)
    kw = dict(bias=torch.randn(cout, device=dev), pro_scale=ps[0], pro_shift=ps[1], precision=_lib.PREC_H3, w_exp=w_exp)
    y_direct = ops.conv3d(h, wpk, cout, 3, skip=(x1, x2, swpk), **kw)
    y = ops.conv3d(h, wpk, cout, 3, wpk_wino=wino, skip=(x1, x2, swpk, sfrag), **kw)
    torch.cuda.synchronize()
    yr = _ref_conv(h, k, kw["bias"], ps) + torch.einsum("bdhwc,co->bdhwo", torch.cat([x1, x2], -1).double(), ks.double()[0, 0, 0])
    assert not torch.equal(y, y_direct), "the Winograd form did not run"
    assert _rel(y, yr) < 2e-5 and _rel(y_direct, yr) < 2e-5


REAL = [("32^3 192->64, B=8 (two items per workgroup)", 8, 32, 128, 64, 64, 0), ("32^3 64->64 + residual, B=4 (one item per CU)", 4, 32, 64, 0, 64, 1),
        ("16^3 384->128, B=32 (four items per workgroup)", 32, 16, 256, 128, 128, 0)]


@pytest.mark.parametrize("case", REAL, ids=[c[0] for c in REAL])
def test_winograd_real_shapes_without_policy_knobs(dev, monkeypatch, case):
    """The U-Net's own k3 shapes at the batch sizes where the launch policy picks the Winograd-x form by itself (no DM3D_CONV_* knobs), against
    float64 on the first and last sample (the conv is per sample) and against the direct kernel on all of them."""
    from dm3d_amd import ops, _lib
    for v in ("DM3D_CONV_WIDE_WGS", "DM3D_CONV_WINO_MINCHUNKS", "DM3D_CONV_WINO", "DM3D_CONV_WINO_SPLIT", "DM3D_CONV_WINO_GRID", "DM3D_CONV_WINO_PERSIST"):
        monkeypatch.delenv(v, raising=False)
    name, B, e, c1, c2, cout, res = case
    torch.manual_seed(21)
    x1 = torch.randn(B, e, e, e, c1, device=dev)
    x2 = torch.randn(B, e, e, e, c2, device=dev) if c2 else None
    cin = c1 + c2
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
    wpk, w_exp = ops.pack_weights_h3(k)
    wino = ops.pack_weights_h3w(k, w_exp)
    bias = torch.randn(cout, device=dev)
    ps = (torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1)
    r = torch.randn(B, e, e, e, cout, device=dev) if res else None
    kw = dict(x2=x2, bias=bias, pro_scale=ps[0], pro_shift=ps[1], res=r, precision=_lib.PREC_H3, w_exp=w_exp)
    y_direct = ops.conv3d(x1, wpk, cout, 3, **kw)
    y = ops.conv3d(x1, wpk, cout, 3, wpk_wino=wino, **kw)
    torch.cuda.synchronize()
    assert not torch.equal(y, y_direct), "the Winograd form did not run"
    assert _rel(y, y_direct) < 2e-5
    for b in (0, B - 1):
        xx = torch.cat([x1[b:b + 1], x2[b:b + 1]], -1) if c2 else x1[b:b + 1]
        yr = _ref_conv(xx, k, bias, ps, r[b:b + 1] if res else None)
        assert _rel(y[b:b + 1], yr) < 2e-5


def test_ddpm_update_propagates_nan_like_clip_by_value(dev):
    """tf.clip_by_value(mean, -1, 1) keeps a NaN (conditional_dm3d.py:572); fminf / fmaxf would turn it into -1."""
    import ctypes as C
    import dm3d_amd
    from dm3d_amd._lib import lib, check
    from dm3d_amd.networks import conditional_dm3d as cdm
    from oracle import ref_torch as rt
    T = 50
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    m = cdm.DiffusionModel(8, 1024, 4, None, _args(T, 2), weights=dm3d_amd.synthetic_weights(cfg, seed=0))
    g = torch.Generator().manual_seed(0)
    x, eps, z = (torch.randn(2, 8, 8, 8, 4, generator=g) for _ in range(3))
    eps[1, 2, 2, 2, 1] = float("nan")
    t = torch.tensor([7, 30])
    want = rt.ddpm_step(rt.Betas(T), x, eps, t, z)
    xd = x.to(dev).clone()
    d = m._ddpm_desc(xd, eps.to(dev), t.to(torch.int32).to(dev), 1, noise=z.to(dev))
    check(lib().dm3d_ddpm_update(C.byref(d), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = xd.cpu()
    assert torch.isnan(want[1, 2, 2, 2, 1]) and torch.isnan(got[1, 2, 2, 2, 1])
    ok = ~torch.isnan(want)
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and float((got[ok] - want[ok]).abs().max()) < 2e-6


@pytest.mark.parametrize("where", ["out.conv.bias", "conv_in.kernel", "mid"])
def test_nan_born_inside_the_unet_raises(dev, where):
    """A NaN made INSIDE a step (a diverged checkpoint: NaN in a weight) must not come back as plausible clipped latents: the forward call
    raises on its eps, and a chain raises at its end — the NaN reaches x through the posterior update, whose clip propagates it like
    tf.clip_by_value, and the per-step range op / the end-of-chain check see it there."""
    import dm3d_amd
    from dm3d_amd import _lib
    from dm3d_amd.networks import conditional_dm3d as cdm
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    W = dict(dm3d_amd.synthetic_weights(cfg, seed=1))
    key = where if where != "mid" else "down1.res0.conv2.bias"
    W[key] = W[key].copy()
    W[key].reshape(-1)[0] = np.nan
    m = cdm.DiffusionModel(8, 1024, 4, None, _args(4, 2), weights=W)
    x = torch.randn(2, 8, 8, 8, 4, generator=torch.Generator().manual_seed(3))
    with pytest.raises(_lib.Dm3dError, match="NaN"):
        m.network([x.to(dev), torch.tensor([1, 3]), torch.tensor([[[1]]])])
    with pytest.raises(_lib.Dm3dError, match="NaN"):
        m.generate((2, 8, 8, 8, 4), context_value=1, seed=2)
    smp = m.sampler((2, 8, 8, 8, 4), context_value=1, seed=3, use_graph=False)
    smp.reset()
    with pytest.raises(_lib.Dm3dError):
        for _ in range(4):
            smp.step()


@pytest.mark.parametrize("norm", ["batch", "group"])
def test_fused_attention_block_equals_its_separate_launches(dev, monkeypatch, norm):
    """The CrossAttentionBlock as three launches (dm3d_attn_front, fused attention, dm3d_mlp_fused with the proj_out tail) against the launches they
    replace (proj_in GEMM, layernorm3, grouped q|k / v^T / q2 GEMM, MLP as two GEMMs, proj_out GEMM: the form small batches keep and the oracle
    tests cover at small sizes) on one model and one input, B = 16 at 32^3 x 8ch, both normalisation variants (GroupNormalization feeds the front
    kernel a materialised tensor): eps within 2e-5."""
    import dm3d_amd
    from dm3d_amd.unet import UNet
    B, C = 16, 8
    cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=C, norm=norm)
    W = dm3d_amd.synthetic_weights(cfg, seed=4)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 32, 32, 32, C, generator=g).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g)
    ctx = torch.randint(0, 2, (B, 1, 1), generator=g)
    out, kinds = {}, {}
    for fused in ("1", "0"):
        for var in ("DM3D_ATTN_FRONT", "DM3D_MLP_TAIL", "DM3D_MLP_FUSED"):
            monkeypatch.setenv(var, fused)
        net = UNet(cfg, weights=W, precision="h3")
        out[fused] = net([x, t, ctx]).clone()
        kinds[fused] = net.plan(B, B, per_sample_context=True).count()
        del net
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    assert kinds["1"].get("attn_front", 0) == 6 and kinds["1"].get("mlp_fused", 0) == 6 and "gemm_h3" not in kinds["1"], kinds["1"]
    assert "attn_front" not in kinds["0"] and "mlp_fused" not in kinds["0"] and kinds["0"].get("gemm_h3", 0) >= 24, kinds["0"]
    err = _rel(out["1"], out["0"])
    print(f"norm={norm}: fused block vs separate launches, eps rel diff {err:.2e}")
    assert torch.isfinite(out["1"]).all() and err < 2e-5


def test_bench_four_rank_rehearsal(dev):
    """`python bench.py --gpus 4` in rehearsal mode: four rank processes on GPU 0 over gloo (the boxes allow six GPU processes, so this is the
    widest rehearsal that touches the card; the eight-rank width of the driver's run is rehearsed without a GPU in tests/test_host.py)."""
    env = dict(os.environ, DM3D_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--size", "8", "--channels", "4", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 4 and line["config"]["global_batch"] == 8 and line["scaling"] == "weak"
    ranks = line["ranks"]
    assert ranks["world_size"] == 4 and [p["seed"] for p in ranks["per_rank"]] == [1234, 1235, 1236, 1237]
    assert len({p["weights_sha"] for p in ranks["per_rank"]}) == 1 and all(p["ms_per_step"] > 0 for p in ranks["per_rank"])


@pytest.mark.parametrize("m", [64, 200, 2048], ids=["one tile", "partial last tile", "32 tiles"])
@pytest.mark.parametrize("out_h2", [False, True], ids=["f32 out", "H2 out"])
def test_mlp_fused_against_float64(dev, m, out_h2):
    """dm3d_mlp_fused: Dense(u)(relu(Dense(4u)(x))) + res + res2 with the hidden activation kept in LDS (conditional_dm3d.py:132-133, 193-195)
    against float64, u = 256; float32 and DM3D_FMT_H2 outputs, row counts that are not whole 64-row tiles."""
    from dm3d_amd import ops
    u = 256
    g = torch.Generator().manual_seed(4)
    x = torch.randn(m, u, generator=g)
    w0, b0 = torch.randn(4 * u, u, generator=g) / 16.0, torch.randn(4 * u, generator=g) * 0.1
    w1, b1 = torch.randn(u, 4 * u, generator=g) / 32.0, torch.randn(u, generator=g) * 0.1
    r1, r2 = torch.randn(m, u, generator=g), torch.randn(m, u, generator=g)
    ref = torch.relu(x.double() @ w0.double().T + b0.double()) @ w1.double().T + b1.double() + r1.double() + r2.double()
    c = lambda t: t.to(dev).contiguous()
    w0t, w1t = ops.pack_mlp_weights(ops.split_h2(c(w0)), u, 0), ops.pack_mlp_weights(ops.split_h2(c(w1)), u, 1)
    out = ops.mlp_fused(ops.split_h2(c(x)), w0t, c(b0), w1t, c(b1), u, res=c(r1), res2=c(r2), out_h2=out_h2)
    if out_h2:
        out = ops.h2_to_f32(out, u)
    torch.cuda.synchronize()
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    print(f"mlp_fused m={m} out_h2={out_h2}: {err:.2e}")
    assert err < 2e-5
    # without residuals
    out = ops.mlp_fused(ops.split_h2(c(x)), w0t, c(b0), w1t, c(b1), u)
    ref0 = ref - r1.double() - r2.double()
    assert float((out.cpu().double() - ref0).abs().max() / ref0.abs().max()) < 2e-5
    # with the proj_out tail: relu(Dense(u)(that)) + res3, the MLP's own result not stored (conditional_dm3d.py:195)
    if not out_h2:
        w2, b2, r3 = torch.randn(u, u, generator=g) / 16.0, torch.randn(u, generator=g) * 0.1, torch.randn(m, u, generator=g)
        ref2 = torch.relu(ref @ w2.double().T + b2.double()) + r3.double()
        w2t = ops.pack_front_weights(ops.split_h2(c(w2)), u)
        out = ops.mlp_fused(ops.split_h2(c(x)), w0t, c(b0), w1t, c(b1), u, res=c(r1), res2=c(r2), tail=(w2t, c(b2), c(r3)))
        torch.cuda.synchronize()
        err2 = float((out.cpu().double() - ref2).abs().max() / ref2.abs().max())
        print(f"mlp_fused + tail m={m}: {err2:.2e}")
        assert err2 < 2e-5
        out = ops.mlp_fused(ops.split_h2(c(x)), w0t, c(b0), w1t, c(b1), u, res=c(r1), res2=c(r2), tail=(w2t, c(b2), None))
        assert float((out.cpu().double() - (ref2 - r3.double())).abs().max() / ref2.abs().max()) < 2e-5


@pytest.mark.parametrize("mr", ["1", "2"], ids=["32-row tiles", "64-row tiles"])
@pytest.mark.parametrize("m", [64, 1024], ids=["one tile", "16 tiles"])
def test_attn_front_against_float64(dev, monkeypatch, m, mr):
    """dm3d_attn_front: relu(proj_in) -> three LayerNormalizations -> q|k, v^T, q2 projections in one launch (conditional_dm3d.py:186-193,
    163-170) against float64, and against the three launches it replaces; u = 256."""
    from dm3d_amd import ops
    monkeypatch.setenv("DM3D_FRONT_MR", mr)
    u = 256
    g = torch.Generator().manual_seed(7)
    x = torch.randn(m, u, generator=g) * 2.0
    w_in, b_in = torch.randn(u, u, generator=g) / 16.0, torch.randn(u, generator=g) * 0.1
    w_qk, b_qk = torch.randn(2 * u, u, generator=g) / 16.0, torch.randn(2 * u, generator=g) * 0.1
    w_v, b_v = torch.randn(u, u, generator=g) / 16.0, torch.randn(u, generator=g) * 0.1
    norms = [(torch.rand(u, generator=g) + 0.5, torch.randn(u, generator=g) * 0.2) for _ in range(3)]
    D = lambda t: t.double()
    y_ref = torch.relu(D(x) @ D(w_in).T + D(b_in))
    ln = lambda gb: torch.nn.functional.layer_norm(y_ref, (u,), D(gb[0]), D(gb[1]), 1e-3)
    n1, n2, n3 = ln(norms[0]), ln(norms[1]), ln(norms[2])
    qk_ref, v_ref, q2_ref = n1 @ D(w_qk).T + D(b_qk), n1 @ D(w_v).T + D(b_v), n2 @ D(w_qk[:u]).T + D(b_qk[:u])
    c = lambda t: t.to(dev).contiguous()
    tile = lambda w: ops.pack_front_weights(ops.split_h2(c(w)), w.shape[0])
    y, qk, v_t, q2, n3o = ops.attn_front(c(x), tile(w_in), c(b_in), tile(w_qk), c(b_qk), tile(w_v), c(b_v), [(c(a), c(b)) for a, b in norms])
    torch.cuda.synchronize()
    rel = lambda got, ref: float((got.cpu().double() - ref).abs().max() / ref.abs().max())
    errs = {"y": rel(y, y_ref), "qk": rel(ops.h2_to_f32(qk, 2 * u), qk_ref), "v_t": rel(ops.h2_to_f32(v_t, m), v_ref.T.contiguous()),
            "q2": rel(ops.h2_to_f32(q2, u), q2_ref), "n3": rel(ops.h2_to_f32(n3o, u), n3)}
    print(f"attn_front m={m}: " + " ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert max(errs.values()) < 2e-5, errs


GN_CASES = [("direct kernel, whole bricks (fused)", 2, (8, 8, 8), 32, 64, 0, 0), ("Winograd form, item lists (fused)", 2, (16, 16, 16), 32, 64, 1, 0),
            ("ragged extent and Cout (the library sums behind the launch)", 1, (6, 8, 10), 16, 40, 0, 0), ("UpSample parity form", 1, (8, 8, 8), 32, 64, 0, 1),
            ("Cin split on a small grid (sums behind the launch)", 1, (8, 8, 8), 256, 64, 0, 0)]


@pytest.mark.parametrize("case", GN_CASES, ids=[c[0] for c in GN_CASES])
def test_conv_sums_groupnorm_statistics_of_its_output(dev, monkeypatch, case):
    """dm3d_conv_desc.gn_stats: per-(sample, channel) partial sums and sums of squares of the OUTPUT, stored by the producing launch (the
    GroupNormalization variant the reference keeps commented out, conditional_dm3d.py:77, 254, 261, 409 — fused into the producer as
    north_star words it), against torch on the stored output; and dm3d_groupnorm_finalize2 over two such buffers against the oracle's
    group_norm of the concatenated tensors."""
    from dm3d_amd import ops, _lib
    from dm3d_amd._lib import lib, check
    name, B, dims, cin, cout, wino, up = case
    if wino:
        monkeypatch.setenv("DM3D_CONV_WIDE_WGS", "1")
        monkeypatch.setenv("DM3D_CONV_WINO_GRID", "4")
    torch.manual_seed(3)
    x = torch.randn(B, *dims, cin, device=dev)
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.1
    if up:
        wpk, w_exp = ops.pack_weights_up(k, h3=True)
    else:
        wpk, w_exp = ops.pack_weights_h3(k)
    eo = [d * (2 if up else 1) for d in dims]
    vox_out = eo[0] * eo[1] * eo[2]
    st = torch.full((B, (vox_out + 63) // 64, cout, 2), float("nan"), device=dev)          # every slot must be written
    kw = dict(bias=torch.randn(cout, device=dev), precision=_lib.PREC_H3, w_exp=w_exp, gn_stats=st, upsample=bool(up))
    if wino:
        kw["wpk_wino"] = ops.pack_weights_h3w(k, w_exp)
    y = ops.conv3d(x, wpk, cout, 3, **kw)
    torch.cuda.synchronize()
    yd = y.double().reshape(B, -1, cout)
    want = torch.stack([yd.sum(1), (yd * yd).sum(1)], -1)
    assert torch.isfinite(st).all()
    err = float((st.double().sum(1) - want).abs().max() / want.abs().max())
    print(f"{name}: statistics rel err {err:.2e}")
    assert err < 2e-6
    # two tensors -> per-sample scale / shift of GroupNormalization(8)(concat) (here: the tensor with itself)
    g = torch.rand(2 * cout, device=dev) + 0.5
    b_ = torch.randn(2 * cout, device=dev)
    scale, shift = torch.empty(B, 2 * cout, device=dev), torch.empty(B, 2 * cout, device=dev)
    vox = yd.shape[1]
    check(lib().dm3d_groupnorm_finalize2(st.data_ptr(), cout, st.data_ptr(), cout, B, vox, 8, 1e-3, g.data_ptr(), b_.data_ptr(), scale.data_ptr(),
                                         shift.data_ptr(), torch.cuda.current_stream().cuda_stream), "groupnorm_finalize2")
    cat = torch.cat([yd, yd], -1)                                         # [B, vox, 2 cout]
    grp = cat.reshape(B, vox, 8, -1)
    mean, var = grp.mean((1, 3)), grp.var((1, 3), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-3)
    sc = g.double().reshape(8, -1) * rstd[:, :, None]
    sh = b_.double().reshape(8, -1) - mean[:, :, None] * sc
    torch.cuda.synchronize()
    assert float((scale.double() - sc.reshape(B, -1)).abs().max() / sc.abs().max()) < 1e-5
    assert float((shift.double() - sh.reshape(B, -1)).abs().max() / sh.abs().max().clamp_min(1.0)) < 1e-5

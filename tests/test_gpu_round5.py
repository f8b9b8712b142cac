"""Round-5 GPU cases: chain-level determinism of the plans that had none — the GroupNormalization plan at B = 32 and the
small-batch plan of BASELINE config 2 (direct kernel, split-K launches, grouped GEMMs) —, the fused skip tail of the Winograd kernel against
float64 at the tolerance a correct operand read gives (a read-after-write hazard in front of its asm MFMAs sat inside the old 2e-5 bar),
and the Cin-split hand-over (no zero fill, no atomics).  All through the C ABI (ctypes); float64 torch is the checker only."""
import os
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


def _args(T, bs=1):
    return SimpleNamespace(timesteps=T, num_gpus=1, kernel_resize=False, bs=bs)


def _chain_three_ways(model, shape, steps):
    outs = [model.generate(shape, context_value=1, seed=11, steps=steps, use_graph=g) for g in (True, False, True)]
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    return outs


def test_groupnorm_plan_graph_chain_equals_eager_chain_b32(dev):
    """norm="group" at the BASELINE shape (B = 32, 32^3 x 8ch): 100 graph-replayed steps == 100 eager steps == a second graph chain, bit for
    bit.  This plan has launches the batch-norm plan has not (statistics in the conv epilogues, the finalize / partials kernels, per-sample
    prologue vectors) and had no chain-level test."""
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8, norm="group")
    m = cdm.DiffusionModel(32, 1024, 8, None, _args(1000, 32), weights=dm3d_amd.synthetic_weights(cfg, seed=0), norm="group")
    a, b, c = _chain_three_ways(m, (32, 32, 32, 32, 8), 100)
    assert torch.equal(a, b) and torch.equal(a, c)


def test_config2_plan_graph_chain_equals_eager_chain(dev):
    """BASELINE config 2 (32^3 x 4ch, B = 4): the small-batch plan — one round of the direct kernel per conv, the Cin-split launches of the
    8^3 and 16^3 levels, the grouped-GEMM form of the attention block — 150 steps three ways, bit for bit."""
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=4)
    m = cdm.DiffusionModel(32, 1024, 4, None, _args(1000, 4), weights=dm3d_amd.synthetic_weights(cfg, seed=0))
    a, b, c = _chain_three_ways(m, (4, 32, 32, 32, 4), 150)
    assert torch.equal(a, b) and torch.equal(a, c)
    kinds = {op[2] for op in m.sampler((4, 32, 32, 32, 4), context_value=1, seed=1).plan.ops}
    assert "gemm_h3" in kinds and any(k.startswith("conv_k3s1") for k in kinds), kinds


# ---- the Cin split in its hand-over form: every part stores its raw tiles, the part that draws the tile's last ticket sums
# them in part order and runs the epilogue — one launch, no zero fill, no atomic adds, any epilogue ------------------------------------------------
def _ref_conv(x, k, bias=None, pro=None, res=None, relu=False):
    xd = x.double()
    if pro is not None:
        xd = xd * pro[0].double() + pro[1].double()
        xd = xd * torch.sigmoid(xd)
    y = F.conv3d(xd.permute(0, 4, 1, 2, 3), k.double().permute(4, 3, 0, 1, 2), padding=1).permute(0, 2, 3, 4, 1)
    if bias is not None:
        y = y + bias.double()
    if relu:
        y = torch.relu(y)
    if res is not None:
        y = y + res.double()
    return y


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def _counters_are_zero(dev):
    from dm3d_amd import ops
    buf = ops._COUNTERS.get(dev)
    return buf is not None and int(buf.abs().sum().item()) == 0


HANDOVER = [
    # name, B, edge, c1, c2, cout, relu, residual: the 8^3 x 256 shapes of the U-Net (B = 32 there; 4 here = the same kernels on fewer items)
    ("winograd 2-way, 8^3 256->256 + res", 4, 8, 256, 0, 256, False, True),
    ("winograd 2-way, 8^3 512->256 (concat 256+256), ReLU + res", 4, 8, 256, 256, 256, True, True),
    ("direct kernel 16-way, 8^3 256->256, B=1", 1, 8, 256, 0, 256, False, True),
    ("direct kernel, 16^3 128->128 B=1, ReLU", 1, 16, 128, 0, 128, True, False),
    ("direct kernel, partial bricks 6^3 96+64->72", 1, 6, 96, 64, 72, False, True),
]


@pytest.mark.parametrize("case", HANDOVER, ids=[c[0] for c in HANDOVER])
def test_cin_split_hand_over_against_float64(dev, monkeypatch, case):
    from dm3d_amd import ops, _lib
    name, B, e, c1, c2, cout, relu, res = case
    for v in ("DM3D_CONV_WIDE_WGS", "DM3D_CONV_WINO_MINCHUNKS", "DM3D_CONV_WINO", "DM3D_CONV_WINO_SPLIT", "DM3D_CONV_KSPLIT"):
        monkeypatch.delenv(v, raising=False)
    if name.startswith("winograd"):
        monkeypatch.setenv("DM3D_CONV_WIDE_WGS", "1")             # (the policy wants 128 tiles for the Winograd split: this batch has 16)
    torch.manual_seed(11)
    x1 = torch.randn(B, e, e, e, c1, device=dev)
    x2 = torch.randn(B, e, e, e, c2, device=dev) if c2 else None
    cin = c1 + c2
    k = torch.randn(3, 3, 3, cin, cout, device=dev) * 0.05
    wpk, w_exp = ops.pack_weights_h3(k)
    wino = ops.pack_weights_h3w(k, w_exp) if name.startswith("winograd") else None
    ps = (torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1)
    r = torch.randn(B, e, e, e, cout, device=dev) if res else None
    kw = dict(x2=x2, bias=torch.randn(cout, device=dev), pro_scale=ps[0], pro_shift=ps[1], res=r, relu=relu, precision=_lib.PREC_H3, w_exp=w_exp,
              wpk_wino=wino)
    y = ops.conv3d(x1, wpk, cout, 3, **kw)
    y_again = ops.conv3d(x1, wpk, cout, 3, **kw)
    y_unsplit = ops.conv3d(x1, wpk, cout, 3, split=False, **kw)
    torch.cuda.synchronize()
    yr = _ref_conv(torch.cat([x1, x2], -1) if c2 else x1, k, kw["bias"], ps, r, relu)
    assert not torch.equal(y, y_unsplit), "the launch did not split"
    assert torch.equal(y, y_again), "the sum of the parts must not depend on their arrival order"
    assert _counters_are_zero(dev), "every launch leaves the ticket words zero"
    assert _rel(y, yr) < 2e-5 and _rel(y_unsplit, yr) < 2e-5


def test_cin_split_hand_over_keeps_the_fused_output_forms(dev, monkeypatch):
    """What a split launch could not do before round 5: the DM3D_FMT_H2 hand-off output with the consumer's norm + swish applied (conv1 of a
    ResidualBlock at the 8^3 level), and the fused GroupNormalization statistics, both from the part that finishes the tile."""
    from dm3d_amd import ops, _lib
    for v in ("DM3D_CONV_WIDE_WGS", "DM3D_CONV_WINO_MINCHUNKS", "DM3D_CONV_WINO", "DM3D_CONV_WINO_SPLIT", "DM3D_CONV_KSPLIT"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("DM3D_CONV_WIDE_WGS", "1")
    torch.manual_seed(12)
    B, e, c, cm, co = 4, 8, 256, 256, 256
    x = torch.randn(B, e, e, e, c, device=dev)
    ka, kb = torch.randn(3, 3, 3, c, cm, device=dev) * 0.05, torch.randn(3, 3, 3, cm, co, device=dev) * 0.05
    wa, ea = ops.pack_weights_h3(ka)
    wb, eb = ops.pack_weights_h3(kb)
    wwa, wwb = ops.pack_weights_h3w(ka, ea), ops.pack_weights_h3w(kb, eb)
    post = (torch.rand(cm, device=dev) + 0.5, torch.randn(cm, device=dev) * 0.1)
    pro = (torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1)
    a = ops.conv3d(x, wa, cm, 3, bias=torch.zeros(cm, device=dev), pro_scale=pro[0], pro_shift=pro[1], precision=_lib.PREC_H3, w_exp=ea,
                   post=post, out_h2=True, wpk_wino=wwa)
    y = ops.conv3d(a, wb, co, 3, precision=_lib.PREC_H3, w_exp=eb, x1_h2_channels=cm, wpk_wino=wwb)
    mid = _ref_conv(x, ka, None, pro)
    mid = mid * post[0].double() + post[1].double()
    mid = mid * torch.sigmoid(mid)
    yr = _ref_conv(mid.float(), kb)
    torch.cuda.synchronize()
    assert _rel(y, yr) < 2e-5 and _counters_are_zero(dev)
    # fused statistics: partial (sum, sum of squares) per (sample, slot, channel) of the finished output
    nslots = e * e * e // 64
    gn = torch.full((B, nslots, co, 2), float("nan"), device=dev)
    y2 = ops.conv3d(x, wa, cm, 3, bias=torch.randn(cm, device=dev), pro_scale=pro[0], pro_shift=pro[1], precision=_lib.PREC_H3, w_exp=ea,
                    wpk_wino=wwa, gn_stats=gn)
    torch.cuda.synchronize()
    s = gn.double().sum(1)
    want = torch.stack([y2.double().sum((1, 2, 3)), (y2.double() ** 2).sum((1, 2, 3))], -1)
    assert float((s - want).abs().max() / want.abs().max()) < 1e-5


def test_fused_attention_block_kernels_below_their_row_threshold(dev, monkeypatch):
    """DM3D_FUSED_MIN_ROWS (round 5): dm3d_attn_front / dm3d_mlp_fused take over from the grouped GEMM launches from n rows (B x tokens) up;
    the default (8192) is the measured break-even, smaller values are slower but must compute the same eps: B = 2 (1024 rows per block,
    32-row tiles of the front kernel, 16 workgroups of the MLP kernel) with the threshold at 512 against the default plan."""
    import dm3d_amd
    from dm3d_amd.unet import UNet
    cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
    W = dm3d_amd.synthetic_weights(cfg, seed=5)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 32, 32, 32, 8, generator=g).to(dev)
    t = torch.tensor([700, 3])
    ctx = torch.ones(2, 1, 1, dtype=torch.int64)
    monkeypatch.delenv("DM3D_FUSED_MIN_ROWS", raising=False)
    ref_net = UNet(cfg, weights=W)
    ref = ref_net([x, t, ctx]).clone()
    kinds_ref = {op[2] for op in ref_net.plan(2, 1000, False).ops}
    monkeypatch.setenv("DM3D_FUSED_MIN_ROWS", "512")
    net = UNet(cfg, weights=W)
    out = net([x, t, ctx]).clone()
    kinds = {op[2] for op in net.plan(2, 1000, False).ops}
    torch.cuda.synchronize()
    assert "attn_front" in kinds and "mlp_fused" in kinds and "attn_front" not in kinds_ref
    assert torch.isfinite(out).all()
    assert float((out - ref).abs().max() / ref.abs().max()) < 2e-5

"""GPU parity of the assembled path: U-Net eps (conditional and unconditional), DDPM generate loop, HIP-graph replay.

Bar (BASELINE.json north_star): eps within 1e-3 relative (max |err| / max |ref|, fp32) of the CPU oracle on identical
noised latents.  The committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle)
are checked at the same tolerance; intermediate taps localise a failure.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    from dm3d_amd import _lib
    _lib.require_device()
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


def _rel(a, ref):
    a, ref = torch.as_tensor(a).double().cpu(), torch.as_tensor(ref).double()
    return float((a - ref).abs().max() / ref.abs().max())


def _args(T, bs=1):
    return SimpleNamespace(timesteps=T, num_gpus=1, kernel_resize=False, bs=bs)


@pytest.mark.parametrize("prec", ["fp32", "h3"])
@pytest.mark.parametrize("cond", [True, False], ids=["conditional", "unconditional"])
def test_unet_eps_golden_s8(dev, cond, prec):
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d, dm3d
    g = np.load(os.path.join(GOLD, "unet_cond_s8c4.npz" if cond else "unet_uncond_s8c4.npz"))
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4, conditional=cond)
    W = dm3d_amd.synthetic_weights(cfg, seed=0)
    mod = conditional_dm3d if cond else dm3d
    net = mod.build_model(8, 4, [64, 128, 256], [False, False, True, True], precision=prec)
    net.load_state_dict(W)
    x, t = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["t"])
    inputs = [x, t, torch.from_numpy(g["ctx"])] if cond else [x, t]
    eps = net(inputs)
    torch.cuda.synchronize()
    err32, err64 = _rel(eps, g["eps"]), _rel(eps, g["eps64"])
    print(f"[{prec}] eps rel err vs oracle fp32 {err32:.3e}, vs oracle fp64 {err64:.3e}")
    assert err32 < TOL and err64 < TOL
    # a second call with a different batch composition reuses/rebuilds plans correctly
    eps1 = net([x[1:2], t[1:2]] + ([torch.from_numpy(g["ctx"])[1:2]] if cond else []))
    assert _rel(eps1, g["eps"][1:2]) < TOL


def test_unet_eps_live_oracle_blocks(dev):
    """Same inputs through the oracle here on the CPU, comparing block outputs to localise any mismatch."""
    import dm3d_amd
    from dm3d_amd.unet import UNet
    from oracle import ref_torch as rt
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=8)
    W = dm3d_amd.synthetic_weights(cfg, seed=3)
    net = UNet(cfg, weights=W)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 8, 8, 8, 8, generator=g)
    t = torch.tensor([999, 0, 500])
    ctx = torch.tensor([[[1]], [[1]], [[0]]])
    ref = rt.unet_forward({k: torch.from_numpy(v) for k, v in W.items()}, rt.UNetConfig(img_size=8, img_channels=8), x, t, ctx)
    eps = net([x.to(dev), t, ctx])
    assert _rel(eps, ref) < TOL
    # broadcast context ([1,1,1]) equals the same id repeated
    e1 = net([x.to(dev), t, torch.tensor([[[1]]])])
    e2 = net([x.to(dev), t, torch.tensor([[[1]], [[1]], [[1]]])])
    assert torch.equal(e1, e2)


@pytest.mark.parametrize("kw", [
    dict(widths=(32, 64), has_attention=(False, True), num_res_blocks=1, first_conv_channels=16),          # narrow: Cout <= 32 kernels
    dict(widths=(48, 80, 112), has_attention=(True, False, True), num_res_blocks=1),                        # not multiples of 64
    dict(widths=(64,), has_attention=(True,), num_res_blocks=3),                                            # one level, no down/up path
    dict(widths=(64, 128), has_attention=(False, False), num_res_blocks=2, conditional=False, first_conv_channels=64),
], ids=["narrow", "odd_widths", "single_level", "uncond_no_attn"])
@pytest.mark.parametrize("prec", ["h3", "fp32"])
def test_unet_other_build_model_arguments(dev, kw, prec):
    """build_model is parametric (conditional_dm3d.py:324-335): widths, attention flags, block counts other than the
    reference's defaults go through the same plan builder and must match the oracle as well."""
    import dm3d_amd
    from dm3d_amd.unet import UNet
    from oracle import ref_torch as rt
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4, **kw)
    W = dm3d_amd.synthetic_weights(cfg, seed=6)
    net = UNet(cfg, weights=W, precision=prec)
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 8, 8, 8, 4, generator=g)
    t = torch.tensor([0, 731])
    ctx = torch.tensor([[[0]], [[1]]])
    ocfg = rt.UNetConfig(img_size=8, img_channels=4, **kw)
    ref = rt.unet_forward({k: torch.from_numpy(v) for k, v in W.items()}, ocfg, x, t, ctx if cfg.conditional else None)
    eps = net([x.to(dev), t, ctx] if cfg.conditional else [x.to(dev), t])
    torch.cuda.synchronize()
    assert _rel(eps, ref) < TOL


@pytest.mark.parametrize("size,kw", [
    (24, dict(widths=(16, 32, 48), num_res_blocks=1, first_conv_channels=16)),      # 6^3 = 216 tokens: not a multiple of 16
    (20, dict(widths=(16, 32), has_attention=(False, True), num_res_blocks=1, first_conv_channels=16)),   # 10^3 tokens, ragged bricks
    (64, dict(widths=(16, 32, 64), num_res_blocks=1, first_conv_channels=16)),      # 16^3 = 4096 tokens: the streaming softmax
], ids=["24cube_216tok", "20cube_1000tok", "64cube_4096tok"])
def test_unet_other_latent_sizes(dev, size, kw):
    import dm3d_amd
    from dm3d_amd.unet import UNet
    from oracle import ref_torch as rt
    cfg = dm3d_amd.UNetConfig(img_size=size, img_channels=4, **kw)
    W = dm3d_amd.synthetic_weights(cfg, seed=1)
    x = torch.randn(1, size, size, size, 4, generator=torch.Generator().manual_seed(0))
    t, ctx = torch.tensor([900]), torch.tensor([[[1]]])
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ref = rt.unet_forward({k: torch.from_numpy(v) for k, v in W.items()}, rt.UNetConfig(img_size=size, img_channels=4, **kw), x, t, ctx)
    for prec in ("h3", "fp32"):
        eps = UNet(cfg, weights=W, precision=prec)([x.to(dev), t, ctx])
        torch.cuda.synchronize()
        assert _rel(eps, ref) < TOL, prec
    bad = dm3d_amd.UNetConfig(img_size=12, img_channels=4, widths=(16, 32, 48), num_res_blocks=1, first_conv_channels=16)
    with pytest.raises(ValueError, match="tokens"):        # 3^3 = 27 tokens at the attention level
        UNet(bad, weights=None)([torch.zeros(1, 12, 12, 12, 4, device=dev), t, ctx])


def _random_unet_configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        levels = int(rng.integers(1, 4))
        widths = tuple(int(rng.choice([16, 32, 48, 64, 80, 96, 128])) for _ in range(levels))
        if len(set(widths)) != len(widths):
            continue                                   # equal widths break the reference's own skip bookkeeping (:379 vs :391)
        size = int(rng.choice([4, 8])) << (levels - 1)
        att = tuple(bool(rng.random() < 0.5) for _ in range(levels))
        if any(a and ((size >> i) ** 3) % 4 for i, a in enumerate(att)) or ((size >> (levels - 1)) ** 3) % 4:
            continue
        kw = dict(widths=widths, has_attention=att, num_res_blocks=int(rng.integers(1, 3)), conditional=bool(rng.random() < 0.7),
                  first_conv_channels=int(rng.choice([16, 32, 64])), norm=str(rng.choice(["batch", "batch", "group"])))
        if kw["norm"] == "group" and any(w % 8 for w in widths + (kw["first_conv_channels"],)):
            continue
        out.append((size, int(rng.choice([4, 8])), int(rng.integers(1, 4)), str(rng.choice(["h3", "h3", "fp32"])), kw))
    return out


@pytest.mark.parametrize("case", _random_unet_configs(8, 20260103), ids=lambda c: f"s{c[0]}c{c[1]}b{c[2]}{c[3]}_" + "x".join(map(str, c[4]["widths"])))
def test_unet_random_build_model_arguments(dev, case):
    """Seeded sweep over build_model arguments against the float64 oracle.  Random weights can make attention ill-conditioned
    (saturated softmax: the float32 and float64 oracles themselves then differ by 1e-3 and more), so the bar is relative to that
    conditioning: error <= max(2e-5, 8 x |oracle32 - oracle64|)."""
    import dm3d_amd
    from dm3d_amd.unet import UNet
    from oracle import ref_torch as rt
    size, ch, B, prec, kw = case
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    cfg = dm3d_amd.UNetConfig(img_size=size, img_channels=ch, **kw)
    W = dm3d_amd.synthetic_weights(cfg, seed=size + ch)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, size, size, size, ch, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    ctx = torch.randint(0, 2, (B, 1, 1), generator=g)
    ocfg = rt.UNetConfig(img_size=size, img_channels=ch, **kw)
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    c = ctx if cfg.conditional else None
    ref32 = rt.unet_forward(Wt, ocfg, x, t, c)
    ref64 = rt.unet_forward({k: v.double() for k, v in Wt.items()}, ocfg, x.double(), t, c)
    eps = UNet(cfg, weights=W, precision=prec)([x.to(dev), t, ctx] if cfg.conditional else [x.to(dev), t])
    torch.cuda.synchronize()
    cond = _rel(ref32, ref64)
    err = _rel(eps, ref64)
    print(f"err {err:.2e}  conditioning (oracle32 vs oracle64) {cond:.2e}")
    assert err <= max(2e-5, 8 * cond)


def test_graft_entry_build_then_smoke_in_one_process(dev):
    """The driver's two entry points in a fresh interpreter, in that order.  (With the library loaded — and HIP initialised —
    before torch was imported, /opt/rocm's runtime shadowed torch's bundled one and torch saw no GPU.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "smoke: ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_generate_matches_golden_trajectory(dev):
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    g = np.load(os.path.join(GOLD, "generate_cond_s8c4_T5.npz"))
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    m = cdm.DiffusionModel(8, 1024, 4, None, _args(5, 2), weights=dm3d_amd.synthetic_weights(cfg, seed=0))
    out = m.generate((2, 8, 8, 8, 4), last_step=0, context_value=1, x_T=torch.from_numpy(g["x_T"]),
                     noise=torch.from_numpy(g["noises"]))
    torch.cuda.synchronize()
    assert float((out.cpu() - torch.from_numpy(g["final"])).abs().max()) < 2e-3     # values are clipped to ~[-1,1]+noise
    first = m.generate((2, 8, 8, 8, 4), context_value=1, x_T=torch.from_numpy(g["x_T"]),
                       noise=torch.from_numpy(g["noises"]), steps=1)
    assert float((first.cpu() - torch.from_numpy(g["step0"])).abs().max()) < 1e-3


def test_generate_graph_equals_eager_and_is_deterministic(dev):
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    m = cdm.DiffusionModel(8, 1024, 4, None, _args(6, 2), weights=dm3d_amd.synthetic_weights(cfg, seed=0))
    a = m.generate((2, 8, 8, 8, 4), context_value=0, seed=5, use_graph=True)
    b = m.generate((2, 8, 8, 8, 4), context_value=0, seed=5, use_graph=False)
    c = m.generate((2, 8, 8, 8, 4), context_value=0, seed=5, use_graph=True)
    d = m.generate((2, 8, 8, 8, 4), context_value=0, seed=6, use_graph=True)
    e = m.generate((2, 8, 8, 8, 4), context_value=1, seed=5, use_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c)
    assert not torch.equal(a, d) and not torch.equal(a, e)
    assert torch.isfinite(a).all() and float(a.abs().max()) <= 1.0 + 1e-6      # last step is deterministic and clipped
    with pytest.raises(ValueError):
        m.generate((2, 8, 8, 8, 4), context_value=None)
    with pytest.raises(ValueError):
        m.generate((2, 8, 8, 4, 4), context_value=0)
    # many seeds: the per-(plan, seed) graph cache stays bounded and an evicted seed is simply captured again
    first = m.generate((2, 8, 8, 8, 4), context_value=0, seed=100)
    for sd in range(101, 101 + 2 * m.MAX_GRAPHS):
        m.generate((2, 8, 8, 8, 4), context_value=0, seed=sd)
    assert len(m._graphs) <= m.MAX_GRAPHS
    assert torch.equal(first, m.generate((2, 8, 8, 8, 4), context_value=0, seed=100))
    # one context id per volume (extension of the reference's single broadcast id): each chain equals its single-context twin
    mixed = m.generate((2, 8, 8, 8, 4), context_value=torch.tensor([[[0]], [[1]]]), seed=5)
    assert torch.equal(mixed[0], a[0]) and torch.equal(mixed[1], e[1])
    with pytest.raises(ValueError):
        m.generate((2, 8, 8, 8, 4), context_value=[0, 1, 1])
    with pytest.raises(ValueError):
        m.generate((2, 8, 8, 8, 4), context_value=7)


def test_full_size_graph_chain_equals_eager_chain(dev):
    """BASELINE shape (B=32, 32^3 x 8ch): 120 steps replayed from the captured HIP graph equal 120 eagerly launched steps bit for
    bit, and a second graph chain repeats the first.  (A hipMemsetAsync captured as a memset node once raced the atomic split-K
    adds under replay: chains diverged after ~90 steps while every short test stayed green.)"""
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
    m = cdm.DiffusionModel(32, 1024, 8, None, _args(1000, 32), weights=dm3d_amd.synthetic_weights(cfg, seed=0))
    outs = [m.generate((32, 32, 32, 32, 8), context_value=1, seed=7, steps=120, use_graph=g) for g in (True, False, True)]
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_unconditional_config1_generate(dev):
    """BASELINE config 1: dm3d.py U-Net, 16^3 x 4ch, B=1, 50 DDPM steps; a few steps checked against the oracle."""
    import dm3d_amd
    from dm3d_amd.networks import dm3d
    from oracle import ref_torch as rt
    cfg = dm3d_amd.UNetConfig(img_size=16, img_channels=4, conditional=False)
    W = dm3d_amd.synthetic_weights(cfg, seed=0)
    m = dm3d.DiffusionModel(16, 1024, 4, None, _args(50), weights=W)
    g = torch.Generator().manual_seed(4)
    shape = (1, 16, 16, 16, 4)
    x_T = torch.randn(shape, generator=g)
    noises = torch.randn((50,) + shape, generator=g)
    got = m.generate(shape, x_T=x_T, noise=noises, steps=3)
    ocfg = rt.UNetConfig(img_size=16, img_channels=4, conditional=False)
    b = rt.Betas(50)
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    x = x_T
    for i in (49, 48, 47):
        tt = torch.full((1,), i, dtype=torch.int64)
        x = rt.ddpm_step(b, x, rt.unet_forward(Wt, ocfg, x, tt), tt, noises[i])
    assert float((got.cpu() - x).abs().max()) < 2e-3
    full = m.generate(shape, seed=1)
    torch.cuda.synchronize()
    assert torch.isfinite(full).all()


def test_batch_shards_are_independent_and_repeatable(dev):
    """volumes shard across ranks because no sample sees another.  eps of a batch of 6 must equal eps of its
    shards [0:2], [2:6] run separately (different grid sizes: the split-K and tile choices may differ, hence 1e-6 rather than
    bitwise), and the same call twice must be bit-identical (the split-K path adds two partial sums atomically)."""
    import dm3d_amd
    from dm3d_amd.unet import UNet
    cfg = dm3d_amd.UNetConfig(img_size=16, img_channels=8)
    net = UNet(cfg, weights=dm3d_amd.synthetic_weights(cfg, seed=3))
    g = torch.Generator().manual_seed(8)
    x = torch.randn(6, 16, 16, 16, 8, generator=g).to(dev)
    t = torch.tensor([999, 0, 17, 500, 500, 3])
    ctx = torch.ones(6, 1, 1, dtype=torch.int64)
    full = net([x, t, ctx]).clone()
    again = net([x, t, ctx]).clone()
    torch.cuda.synchronize()
    assert torch.equal(full, again)
    parts = []
    for r in range(2):
        lo, hi = (0, 2) if r == 0 else (2, 6)
        parts.append(net([x[lo:hi].contiguous(), t[lo:hi], ctx[lo:hi]]).clone())
    torch.cuda.synchronize()
    joined = torch.cat(parts, 0)
    assert float((joined - full).abs().max() / full.abs().max()) < 1e-6


def test_keras_checkpoint_save_load_roundtrip(dev, tmp_path):
    """model.save_weights(prefix) / model.load_weights(prefix) in the reference's TF2 checkpoint format (tf_checkpoint.py):
    a second model restored from the files predicts the same eps bit for bit."""
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    a = cdm.DiffusionModel(8, 1024, 4, None, _args(6, 2), weights=dm3d_amd.synthetic_weights(cfg, seed=4))
    b = cdm.DiffusionModel(8, 1024, 4, None, _args(6, 2), weights=dm3d_amd.synthetic_weights(cfg, seed=5))
    pre = str(tmp_path / "dm-17.ckpt")
    a.save_weights(pre)
    assert os.path.exists(pre + ".index") and os.path.exists(pre + ".data-00000-of-00001")
    x = torch.randn(2, 8, 8, 8, 4, generator=torch.Generator().manual_seed(1)).to(dev)
    t, ctx = torch.tensor([3, 5]), torch.ones(2, 1, 1, dtype=torch.int64)
    ea = a.network([x, t, ctx]).clone()
    assert not torch.equal(ea, b.network([x, t, ctx]))
    b.load_weights(pre)
    assert torch.equal(ea, b.network([x, t, ctx]))


@pytest.mark.parametrize("prec", ["fp32", "h3"])
def test_unet_eps_full_size_32cube(dev, prec):
    """BASELINE configs 2-4 shape (32^3 x 8ch, real widths) at B=1 against the oracle run on this box's CPU."""
    import dm3d_amd
    from dm3d_amd.unet import UNet
    from oracle import ref_torch as rt
    cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
    W = dm3d_amd.synthetic_weights(cfg, seed=0)
    net = UNet(cfg, weights=W, precision=prec)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 32, 32, 32, 8, generator=g)
    t, ctx = torch.tensor([637]), torch.tensor([[[1]]])
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ref = rt.unet_forward({k: torch.from_numpy(v) for k, v in W.items()}, rt.UNetConfig(img_size=32, img_channels=8), x, t, ctx)
    eps = net([x.to(dev), t, ctx])
    torch.cuda.synchronize()
    err = _rel(eps, ref)
    print(f"[{prec}] 32^3x8 eps rel err {err:.3e}")
    assert err < TOL
    # linearity of the DDPM posterior in (x_t, eps) at full size: sample(a x + b y) = a sample(x) + b sample(y)
    from dm3d_amd.networks import conditional_dm3d as cdm
    m = cdm.DiffusionModel(32, 1024, 8, None, _args(1000), weights=W)
    y = torch.randn(1, 32, 32, 32, 8, generator=g).to(dev)
    xd = x.to(dev)
    tt = torch.tensor([400])
    m1, _ = m.sample(xd, eps, tt, xd.shape)
    m2, _ = m.sample(y, xd, tt, xd.shape)
    m3, _ = m.sample(2 * xd - 3 * y, 2 * eps - 3 * xd, tt, xd.shape)
    assert float((m3 - (2 * m1 - 3 * m2)).abs().max() / m3.abs().max()) < 1e-5


@pytest.mark.parametrize("cond", [True, False], ids=["conditional", "unconditional"])
def test_unet_groupnorm_variant(dev, cond):
    """norm="group": the GroupNormalization(groups=8) variant north_star words and the reference keeps commented out
    (conditional_dm3d.py:77, 254, 261, 409); per-sample statistics on the device, same fused conv prologue."""
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d, dm3d
    from oracle import ref_torch as rt
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4, conditional=cond, norm="group")
    W = dm3d_amd.synthetic_weights(cfg, seed=0)
    mod = conditional_dm3d if cond else dm3d
    net = mod.build_model(8, 4, [64, 128, 256], [False, False, True, True], norm="group")
    net.load_state_dict(W)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(3, 8, 8, 8, 4, generator=g) * torch.tensor([0.5, 1.0, 3.0]).reshape(3, 1, 1, 1, 1)   # per-sample statistics differ
    t = torch.tensor([0, 17, 999])
    ctx = torch.tensor([[[1]], [[0]], [[1]]])
    ocfg = rt.UNetConfig(img_size=8, img_channels=4, conditional=cond, norm="group")
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    ref = rt.unet_forward(Wt, ocfg, x, t, ctx if cond else None)
    refb = rt.unet_forward(Wt, rt.UNetConfig(img_size=8, img_channels=4, conditional=cond), x, t, ctx if cond else None)
    eps = net([x.to(dev), t, ctx] if cond else [x.to(dev), t])
    torch.cuda.synchronize()
    err = _rel(eps, ref)
    print(f"groupnorm variant eps rel err {err:.3e}")
    assert err < TOL and _rel(refb, ref) > 0.05            # and it really is a different network from the BatchNorm one
    eps2 = net([x.to(dev), t, ctx] if cond else [x.to(dev), t])      # the stats accumulator is re-zeroed by the finalize kernel
    assert torch.equal(eps, eps2)

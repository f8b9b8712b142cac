"""CPU tier: the oracle against its own golden vectors, the second restatement and the reference-derived known answers
(SURVEY.md §8(c) i-vi).  PARITY UNPINNED: the reference has no golden data; see oracle/ref_torch.py."""
import os

import numpy as np
import torch

from oracle import ref_numpy as rn
from oracle import ref_torch as rt

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_betas_known_answers_and_golden():
    g = np.load(os.path.join(GOLD, "tables.npz"))
    for T in (50, 1000):
        b = rt.Betas(T)
        assert abs(float(b.beta[0]) - 1e-4) < 1e-10 and abs(float(b.beta[-1]) - 0.02) < 1e-8      # (i)
        assert float(b.alpha_bar_prev[0]) == 1.0
        ab = b.alpha_bar.numpy()
        assert np.all(np.diff(ab) < 0)
        tab = rn.betas(T)
        for n in b.NAMES:
            assert np.array_equal(getattr(b, n).numpy(), g[f"betas{T}.{n}"])
            assert np.allclose(getattr(b, n).numpy(), tab[n], rtol=1e-6, atol=0)


def test_time_embedding_known_answers_and_golden():
    g = np.load(os.path.join(GOLD, "tables.npz"))
    e0 = rt.time_embedding(torch.tensor([0]), 128)[0]
    assert torch.equal(e0[:64], torch.zeros(64)) and torch.equal(e0[64:], torch.ones(64))           # (ii)
    e = rt.time_embedding(torch.tensor(g["temb128.t"]), 128).numpy()
    assert np.allclose(e, g["temb128"], atol=1e-6)
    assert np.allclose(e, rn.time_embedding(g["temb128.t"], 128), atol=2e-4)   # fp32 argument rounding at t~1000
    one = rt.time_embedding(torch.tensor([1]), 128)[0]
    assert abs(float(one[0]) - np.sin(1.0)) < 1e-6 and abs(float(one[63]) - np.sin(1e-4)) < 1e-9   # f_0=1, f_last=1e-4


def test_last_step_is_deterministic():
    b = rt.Betas(10)                                                                                 # (iii)
    x, e = torch.randn(2, 2, 2, 2, 3), torch.randn(2, 2, 2, 2, 3)
    t0 = torch.zeros(2, dtype=torch.int64)
    _, var = rt.ddpm_sample(b, x, e, t0)
    assert float(var.abs().max()) == 0.0
    a = rt.ddpm_step(b, x, e, t0, torch.randn_like(x))
    c = rt.ddpm_step(b, x, e, t0, torch.zeros_like(x))
    assert float((a - c).abs().max()) < 1e-9                     # sqrt(1e-20) * z


def test_fresh_keras_init_outputs_near_zero():
    cfg = rt.UNetConfig(img_size=4, img_channels=4, widths=(16, 32), has_attention=(False, True), first_conv_channels=16)
    W = rt.keras_default_weights(cfg, seed=0)                                                        # (iv)
    x = torch.randn(1, 4, 4, 4, 4)
    y = rt.unet_forward(W, cfg, x, torch.tensor([5]), torch.tensor([[[1]]]))
    assert float(y.abs().max()) < 1e-2
    assert float(rt.unet_forward(rt.synthetic_weights(cfg, 0), cfg, x, torch.tensor([5]), torch.tensor([[[1]]])).abs().max()) > 0.1


def test_param_inventory_matches_survey():
    spec = rt.param_spec(rt.UNetConfig(img_size=32, img_channels=8))                                 # (v)
    assert abs(sum(int(np.prod(s)) for s in spec.values()) / 1e6 - 145.84) < 0.01
    assert spec["up2.res0.conv1.kernel"] == (3, 3, 3, 512, 256) and spec["up0.res2.conv1.kernel"] == (3, 3, 3, 96, 64)
    assert spec["mid.attn.ctx_mlp.kernel"] == (128, 512 * 256)
    u = rt.param_spec(rt.UNetConfig(img_size=16, img_channels=4, conditional=False))
    assert abs(sum(int(np.prod(s)) for s in u.values()) / 1e6 - 41.38) < 0.01
    assert u["up0.res2.conv1.kernel"] == (3, 3, 3, 128, 64) and "mid.attn.depth.kernel" not in u


def test_stride2_same_padding():
    x = torch.arange(2 * 4 * 4 * 4 * 1, dtype=torch.float32).reshape(2, 4, 4, 4, 1)                  # (vi)
    k = torch.zeros(3, 3, 3, 1, 1)
    k[0, 0, 0] = 1.0                                   # picks x[2o + 0]: pad 0 before
    y = rt._conv3d(x, k, torch.zeros(1), stride=2)
    assert y.shape == (2, 2, 2, 2, 1) and torch.equal(y, x[:, ::2, ::2, ::2])
    k = torch.zeros(3, 3, 3, 1, 1)
    k[2, 2, 2] = 1.0                                   # picks x[2o + 2]: the last one falls in the trailing pad
    y = rt._conv3d(x, k, torch.zeros(1), stride=2)
    assert float(y[:, 1].abs().max()) == 0.0 and torch.equal(y[:, 0, 0, 0], x[:, 2, 2, 2])


def test_oracle_matches_golden_and_second_restatement():
    for cond, name in ((True, "unet_cond_s8c4.npz"), (False, "unet_uncond_s8c4.npz")):
        g = np.load(os.path.join(GOLD, name))
        cfg = rt.UNetConfig(img_size=8, img_channels=4, conditional=cond)
        W = rt.synthetic_weights(cfg, seed=0)
        x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
        ctx = torch.from_numpy(g["ctx"]) if cond else None
        taps = {}
        eps = rt.unet_forward(W, cfg, x[:2], t[:2], None if ctx is None else ctx[:2], taps=taps)
        ref = torch.from_numpy(g["eps"][:2])
        assert float((eps - ref).abs().max() / ref.abs().max()) < 2e-5
        for k in ("down0.res0", "down0.ds", "mid.attn", "up2.res0", "up1.us"):
            r = torch.from_numpy(g["tap." + k])
            assert float((taps[k][:1] - r).abs().max() / r.abs().max()) < 2e-5
        # independent NumPy/fp64 restatement on sample 0 (t = 0: exact sinusoid)
        en = rn.unet_forward({k: v.numpy() for k, v in W.items()}, cfg, g["x"][:1], g["t"][:1],
                             None if ctx is None else g["ctx"][:1])
        assert np.abs(en - g["eps64"][:1]).max() / np.abs(en).max() < 1e-10


def test_blocks_torch_vs_numpy_small():
    cfg = rt.UNetConfig(img_size=4, img_channels=4, widths=(8, 16), has_attention=(False, True), first_conv_channels=8)
    W = rt.synthetic_weights(cfg, seed=2)
    W64 = {k: v.double() for k, v in W.items()}
    Wn = rn.to_f64({k: v.numpy() for k, v in W.items()})
    x = torch.randn(2, 4, 4, 4, 8, dtype=torch.float64)
    temb = torch.randn(2, 32, dtype=torch.float64)
    a = rt.residual_block(W64, "down0.res1", x, temb).numpy()
    b = rn.residual_block(Wn, "down0.res1", x.numpy(), temb.numpy())
    assert np.abs(a - b).max() < 1e-11
    y = torch.randn(2, 2, 2, 2, 16, dtype=torch.float64)
    ctx = torch.randn(1, 2, 2, 2, 16, dtype=torch.float64)
    a = rt.cross_attention_block(W64, "mid.attn", y, ctx).numpy()
    b = rn.cross_block(Wn, "mid.attn", y.numpy(), ctx.numpy())
    assert np.abs(a - b).max() < 1e-11
    # stride-2 and upsample convs
    k = torch.randn(3, 3, 3, 8, 8, dtype=torch.float64)
    bias = torch.randn(8, dtype=torch.float64)
    assert np.abs(rt._conv3d(x, k, bias, stride=2).numpy() - rn.conv3d_same(x.numpy(), k.numpy(), bias.numpy(), stride=2)).max() < 1e-12
    assert np.abs(rt._conv3d(rt._upsample2(x), k, bias).numpy()
                  - rn.conv3d_same(x.numpy(), k.numpy(), bias.numpy(), upsample=True)).max() < 1e-12


def test_generate_golden_trajectory():
    g = np.load(os.path.join(GOLD, "generate_cond_s8c4_T5.npz"))
    cfg = rt.UNetConfig(img_size=8, img_channels=4)
    W = rt.synthetic_weights(cfg, seed=0)
    traj = []
    out = rt.generate(W, cfg, rt.Betas(5), 5, torch.from_numpy(g["x_T"]), torch.from_numpy(g["noises"]),
                      context_value=1, trajectory=traj)
    assert float((traj[0] - torch.from_numpy(g["step0"])).abs().max()) < 1e-4
    assert float((out - torch.from_numpy(g["final"])).abs().max()) < 1e-3
    # numpy restatement of one update
    tab = rn.betas(5)
    x, e, z = np.random.default_rng(0).normal(size=(3, 2, 2, 2, 2, 1))
    ref = rt.ddpm_step(rt.Betas(5), torch.from_numpy(x).float(), torch.from_numpy(e).float(),
                       torch.full((2,), 3), torch.from_numpy(z).float()).numpy()
    assert np.abs(ref - rn.ddpm_step(tab, x, e, 3, z)).max() < 1e-5


def test_train_loss_definition():
    n, p = torch.randn(2, 2, 2, 2, 4), torch.randn(2, 2, 2, 2, 4)
    lo = rt.train_loss(n, p, global_bs=8, lc=4)
    assert abs(float(lo) - float(((n - p) ** 2).sum() / 4 / (8 * 256))) < 1e-7


def test_vqvae_bracket_restatements_agree():
    """next-1 oracle pieces (reference networks/vqvae3d_monai.py): PyTorch vs the independent NumPy definitions."""
    cfg = rt.VQVAEConfig(in_channels=1, out_channels=1, num_channels=(4, 8), num_res_layers=2, num_res_channels=(4, 8),
                         num_embeddings=32, embedding_dim=4, input_size=16)
    spec = rt.vqvae_param_spec(cfg)
    assert spec["enc.l0.res0.prelu.alpha"] == (8, 8, 8, 4) and spec["dec.up1.kernel"] == (4, 4, 4, 1, 4)
    assert spec["vq.embeddings"] == (4, 32) and cfg.latent_size == 4
    # Reference-derived known answer: the Keras summary of the (32,64,128) x 3-residual-layer, 64-dim, 256-code VQVAE
    # (main.py:190-209 era settings) logged in experiments/vqvae/vqvae3d-scaled-monai-B8-all.output:33-38 reports
    # 75,593,473 trainable parameters and 2,694 non-trainable (2,688 BatchNorm moving statistics + 6 metric counters).
    logged = rt.vqvae_param_spec(rt.VQVAEConfig(num_channels=(32, 64, 128), num_res_channels=(32, 64, 128), num_res_layers=3,
                                               embedding_dim=64, num_embeddings=256))
    stats = sum(int(np.prod(v)) for k, v in logged.items() if k.endswith((".mean", ".var")))
    assert stats == 2688 and sum(int(np.prod(v)) for v in logged.values()) - stats == 75593473
    W = {k: v.double() for k, v in rt.vqvae_synthetic_weights(cfg, 0).items()}
    Wn = rn.to_f64({k: v.numpy() for k, v in W.items()})
    x = torch.randn(2, 16, 16, 16, 1, dtype=torch.float64)
    a = rt._conv3d_k4s2(x, W["enc.down0.kernel"], W["enc.down0.bias"]).numpy()
    assert np.abs(a - rn.conv3d_same(x.numpy(), Wn["enc.down0.kernel"], Wn["enc.down0.bias"], stride=2)).max() < 1e-12
    t = torch.randn(2, 4, 4, 4, 8, dtype=torch.float64)
    a = rt._conv3d_transpose_k4s2(t, W["dec.up0.kernel"], W["dec.up0.bias"]).numpy()
    assert a.shape == (2, 8, 8, 8, 4)
    assert np.abs(a - rn.conv3d_transpose_k4s2(t.numpy(), Wn["dec.up0.kernel"], Wn["dec.up0.bias"])).max() < 1e-12
    r = torch.randn(2, 8, 8, 8, 4, dtype=torch.float64)
    assert np.abs(rt.vq_residual_unit(W, "enc.l0.res0", r).numpy() - rn.vq_residual_unit(Wn, "enc.l0.res0", r.numpy())).max() < 1e-12
    z = rt.vq_encoder(W, cfg, x)
    q, perp, idx = rt.vq_quantize(W, z)
    assert (idx.numpy() != rn.vq_code_indices(Wn["vq.embeddings"], z.reshape(-1, 4).numpy())).sum() == 0
    assert 1.0 <= float(perp) <= 32.0 and torch.equal(q.reshape(-1, 4), W["vq.embeddings"].t()[idx])
    y = rt.vq_decoder(W, cfg, q)
    assert y.shape == (2, 16, 16, 16, 1)


def test_upsample_conv_equals_eight_parity_convs():
    """The identity behind the UpSample kernel path: conv3(nearest2x(x)) == interleave of 8 2x2x2 convs with summed taps."""
    import dm3d_amd
    from dm3d_amd.weights import upsample_parity_kernels
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 5, 6, 3, generator=g, dtype=torch.float64)
    k = torch.randn(3, 3, 3, 3, 2, generator=g, dtype=torch.float64)
    ref = rt._conv3d(rt._upsample2(x), k, torch.zeros(2, dtype=torch.float64)).numpy()
    pk = upsample_parity_kernels(k.numpy())
    out = np.zeros(ref.shape)
    xp = np.pad(x.numpy(), ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))
    for p in range(8):
        a, b, c = p >> 2, (p >> 1) & 1, p & 1
        for td in range(2):
            for th in range(2):
                for tw in range(2):
                    out[:, a::2, b::2, c::2] += xp[:, td + a:td + a + 4, th + b:th + b + 5, tw + c:tw + c + 6] @ pk[p, td, th, tw]
    assert np.abs(out - ref).max() < 1e-12

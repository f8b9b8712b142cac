"""GPU parity of the VQ-VAE bracket (reference networks/vqvae3d_monai.py) against the CPU oracle:
Conv3D k4/s2, VQVAEResidualUnit (BN folded + full-shape PReLU), Conv3DTranspose k4/s2, nearest-code assignment."""
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
    a, ref = torch.as_tensor(a).double().cpu(), torch.as_tensor(ref).double()
    return float((a - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("prec", ["fp32", "h3"])
@pytest.mark.parametrize("shape", [(2, 8, 8, 8, 16, 64), (1, 16, 16, 16, 4, 32), (1, 4, 8, 16, 32, 40)])
def test_conv_k4s2_and_transpose(dev, prec, shape):
    from dm3d_amd import ops, _lib
    from oracle import ref_torch as rt
    B, D, H, W, cin, cout = shape
    g = torch.Generator().manual_seed(zlib.crc32(str(shape).encode()))
    x = torch.randn(B, D, H, W, cin, generator=g)
    k = torch.randn(4, 4, 4, cin, cout, generator=g) / math.sqrt(cin * 64)
    bias = torch.randn(cout, generator=g)
    h3 = prec == "h3"
    P = _lib.PREC_H3 if h3 else _lib.PREC_F32
    # Conv3D k4 s2 'same' with ReLU
    ref = torch.relu(rt._conv3d_k4s2(x.double(), k.double(), bias.double()))
    r = ops.pack_weights_h3(k.to(dev)) if h3 else (ops.pack_weights(k.to(dev)), 0)
    out = ops.conv3d(x.to(dev), r[0], cout, 4, stride=2, bias=bias.to(dev), relu=True, precision=P, w_exp=r[1])
    assert _rel(out, ref) < 2e-5
    # Conv3DTranspose k4 s2 'same' (+ PReLU slope + residual + output ReLU to exercise the whole epilogue)
    kt = torch.randn(4, 4, 4, cout, cin, generator=g) / math.sqrt(cin * 8)
    alpha = torch.rand(2 * D, 2 * H, 2 * W, cout, generator=g)
    res = torch.randn(B, 2 * D, 2 * H, 2 * W, cout, generator=g)
    y = rt._conv3d_transpose_k4s2(x.double(), kt.double(), bias.double())
    ref = torch.relu(rt._prelu(y, alpha.double()) + res.double())
    r = ops.pack_weights_convt(kt.to(dev), h3=h3) if h3 else (ops.pack_weights_convt(kt.to(dev)), 0)
    out = ops.conv3d(x.to(dev), r[0], cout, 4, stride=2, transpose=True, bias=bias.to(dev), prelu_alpha=alpha.to(dev),
                     res=res.to(dev), relu_out=True, precision=P, w_exp=r[1])
    torch.cuda.synchronize()
    assert tuple(out.shape) == tuple(ref.shape) and _rel(out, ref) < 2e-5


def test_vq_assign_matches_oracle(dev):
    from dm3d_amd import ops
    from oracle import ref_torch as rt
    g = torch.Generator().manual_seed(3)
    for rows, d, k in ((4096, 8, 1024), (513, 256, 128), (77, 4, 33)):
        z = torch.randn(rows, d, generator=g)
        E = torch.randn(d, k, generator=g)
        ref = rt.vq_code_indices({"vq.embeddings": E.double()}, z.double())
        idx = ops.vq_assign(z.to(dev), E.t().contiguous().to(dev), (E ** 2).sum(0).to(dev))
        torch.cuda.synchronize()
        # a float32 tie-break may differ from the float64 argmin on near-ties: accept an equally near code
        bad = (idx.cpu().long() != ref).nonzero().flatten()
        dist = (z.double() ** 2).sum(1, keepdim=True) + (E.double() ** 2).sum(0) - 2 * z.double() @ E.double()
        for r in bad.tolist():
            assert abs(float(dist[r, idx[r]] - dist[r, ref[r]])) < 1e-4 * float(dist[r, ref[r]].abs() + 1)
        assert len(bad) <= rows // 500 + 1
        assert torch.equal(ops.gather_rows(E.t().contiguous().to(dev), idx).cpu(), E.t()[idx.cpu().long()])


@pytest.mark.parametrize("prec", ["fp32", "h3"])
def test_vqvae_bracket_matches_oracle(dev, prec):
    """encoder -> quantizer -> decoder of a small VQVAE (same layer types as the reference's 4-level model)."""
    from dm3d_amd.networks.vqvae3d_monai import VQVAE
    from oracle import ref_torch as rt
    cfg = rt.VQVAEConfig(in_channels=1, out_channels=1, num_channels=(16, 32), num_res_layers=2, num_res_channels=(16, 32),
                         num_embeddings=64, embedding_dim=8, input_size=32)
    W = rt.vqvae_synthetic_weights(cfg, seed=1)
    vq = VQVAE(1, 1, (16, 32), 2, (16, 32), downsample_parameters=((2, 4, 1, "same"),) * 2,
               upsample_parameters=((2, 4, 1, "same", 0),) * 2, num_embeddings=64, embedding_dim=8, dropout=None,
               input_size=32, weights={k: v.numpy() for k, v in W.items()}, precision=prec)
    assert list(vq.spec.items()) == list(rt.vqvae_param_spec(cfg).items())
    g = torch.Generator().manual_seed(8)
    x = torch.rand(2, 32, 32, 32, 1, generator=g)
    z_ref = rt.vq_encoder(W, cfg, x)
    z = vq.encoder(x.to(dev))
    assert tuple(z.shape) == (2, 8, 8, 8, 8) and _rel(z, z_ref) < 1e-4
    q_ref, perp_ref, idx_ref = rt.vq_quantize(W, z_ref)
    q, perp = vq.quantizer(z_ref.to(dev))                       # same latents in -> same codes out
    assert (vq.last_indices.cpu().long() != idx_ref).sum() <= 2 and abs(float(perp) - float(perp_ref)) < 0.05 * float(perp_ref)
    y_ref = rt.vq_decoder(W, cfg, q_ref)
    y = vq.decoder(q_ref.to(dev))
    torch.cuda.synchronize()
    assert tuple(y.shape) == (2, 32, 32, 32, 1) and _rel(y, y_ref) < 1e-4
    with pytest.raises(ValueError):
        vq.encoder(torch.zeros(1, 16, 16, 16, 1))
    with pytest.raises(ValueError):
        VQVAE(1, 1, (8,), 1, (8,), downsample_parameters=((2, 3, 1, "same"),), input_size=16)


def test_diffusion_model_owns_the_bracket(dev):
    """DiffusionModel.encoder / quantizer / decoder (conditional_dm3d.py:455-460) and the decode-after-generate path (:42)."""
    from types import SimpleNamespace
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    args = SimpleNamespace(timesteps=4, num_gpus=1, kernel_resize=False, bs=1)
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=8)          # the reference's latent_size 8 <-> 128^3 images
    m = cdm.DiffusionModel(8, 64, 8, None, args, weights=dm3d_amd.synthetic_weights(cfg, 0))
    assert m._vqvae is None
    lat = m.generate((1, 8, 8, 8, 8), context_value=1, seed=3)
    img = m.decoder(lat)                                            # 8^3 latent -> 128^3 image through 4 transposed convs
    q = m.encode_latents(torch.rand(1, 128, 128, 128, 1))
    torch.cuda.synchronize()
    assert tuple(img.shape) == (1, 128, 128, 128, 1) and torch.isfinite(img).all()
    assert tuple(q.shape) == (1, 8, 8, 8, 8) and m.vqvae_trainer.num_embeddings == 64
    assert m.vqvae_trainer.input_size == 128 and sum(int(np.prod(v)) for v in m.vqvae_trainer.spec.values()) > 50e6


def _vqgan(cfg, W, prec, dev):
    from dm3d_amd.networks.vqgan import VQGAN
    n = len(cfg.num_channels)
    return VQGAN(cfg.in_channels, cfg.out_channels, cfg.num_channels, cfg.num_res_layers, cfg.num_res_channels,
                 downsample_parameters=[(2, 4, 1, "same")] * n, upsample_parameters=[(2, 4, 1, "same", 0)] * n,
                 num_embeddings=cfg.num_embeddings, embedding_dim=cfg.embedding_dim, D=cfg.input_size,
                 weights={k: v.numpy() for k, v in W.items()}, precision=prec)


@pytest.mark.parametrize("prec", ["fp32", "h3"])
def test_vqgan_autoencoder_matches_oracle(dev, prec):
    """networks/vqgan.py Encoder / Decoder (the autoencoder BASELINE config 5 names; vqgan.py:287-475, built as in
    main_exp_vqgan.py:23-38: 2-channel concat[img, mask] input, one k4/s2 level per channel_list entry): BatchNormalization after
    every strided / transposed conv folded into its weights, PReLU in the epilogue.  32^3 -> 8^3 -> 32^3."""
    from oracle import ref_torch as rt
    cfg = rt.VQVAEConfig(in_channels=2, out_channels=2, num_channels=(16, 32), num_res_layers=2, num_res_channels=(16, 32),
                         num_embeddings=64, embedding_dim=8, input_size=32)
    W = rt.vqgan_synthetic_weights(cfg, seed=2)
    vq = _vqgan(cfg, W, prec, dev)
    assert list(vq.spec.items()) == list(rt.vqgan_param_spec(cfg).items())
    g = torch.Generator().manual_seed(9)
    img, mask = torch.rand(2, 32, 32, 32, 1, generator=g), (torch.rand(2, 32, 32, 32, 1, generator=g) > 0.5).float()
    x = torch.cat([img, mask], -1)
    z_ref = rt.vqgan_encoder(W, cfg, x)
    z = vq.encode_images(img.to(dev), mask.to(dev))
    assert tuple(z.shape) == (2, 8, 8, 8, 8) and _rel(z, z_ref) < 1e-4
    assert torch.equal(z, vq.encoder(x.to(dev)))
    q_ref, perp_ref, idx_ref = rt.vq_quantize(W, z_ref)
    q, perp = vq.quantizer(z_ref.to(dev))
    assert (vq.last_indices.cpu().long() != idx_ref).sum() <= 2
    y_ref = rt.vqgan_decoder(W, cfg, q_ref)
    y = vq.decoder(q_ref.to(dev))
    rec, perp2 = vq(x.to(dev))
    torch.cuda.synchronize()
    assert tuple(y.shape) == (2, 32, 32, 32, 2) and _rel(y, y_ref) < 1e-4
    assert tuple(rec.shape) == (2, 32, 32, 32, 2) and torch.isfinite(rec).all() and tuple(vq.call_2(x.to(dev)).shape) == (2, 8, 8, 8, 8)


def test_config5_end_to_end_vqgan_ddpm(dev):
    """BASELINE config 5: vqgan.py encode 128^3 (img + mask) -> 32^3 x 8ch latent -> conditional DDPM at 32^3 x 8ch -> decode, B=8, one
    GPU.  The reference never connects a VQGAN to its DiffusionModel: three stages, generate() from N(0,1).
    The chain here is T=24 (the full T=1000 chain at this shape is timed by tools/e2e_config5.py and bench.py); encoder / decoder
    outputs are checked against the oracle on the first volume."""
    from types import SimpleNamespace
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    from oracle import ref_torch as rt
    B = 8
    cfg = rt.VQVAEConfig(in_channels=2, out_channels=2, num_channels=(32, 64), num_res_layers=2, num_res_channels=(32, 64),
                         num_embeddings=256, embedding_dim=8, input_size=128)
    W = rt.vqgan_synthetic_weights(cfg, seed=3)
    vq = _vqgan(cfg, W, "h3", dev)
    g = torch.Generator().manual_seed(10)
    img = torch.rand(B, 128, 128, 128, 1, generator=g)
    mask = (img > 0.3).float()
    z = vq.encode_images(img.to(dev), mask.to(dev))
    lat, perp = vq.quantizer(z)
    assert tuple(lat.shape) == (B, 32, 32, 32, 8) and torch.isfinite(lat).all() and float(perp) > 1
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    z_ref = rt.vqgan_encoder(W, cfg, torch.cat([img[:1], mask[:1]], -1))
    assert _rel(z[:1], z_ref) < 1e-4
    ucfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
    m = cdm.DiffusionModel(32, 256, 8, None, SimpleNamespace(timesteps=24, num_gpus=1, kernel_resize=False, bs=B),
                           weights=dm3d_amd.synthetic_weights(ucfg, 0))
    gen = m.generate((B, 32, 32, 32, 8), context_value=1, seed=5)
    out = vq.decoder(gen)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (B, 128, 128, 128, 2) and torch.isfinite(out).all()
    y_ref = rt.vqgan_decoder(W, cfg, gen[:1].cpu())
    assert _rel(out[:1], y_ref) < 1e-4

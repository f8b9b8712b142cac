"""Round-3 parity cases: the whole config-1 chain, the float-atomic K-split regime of the weight gradient, partial
denoising (last_step > 0), the unconditional model's public train_step / test, data-parallel train_step over two ranks, the NaN-aware
range guard of a chain driven through Sampler.step().  All through the C ABI (ctypes); the oracle is the checker only."""
import os
import subprocess
import sys
import textwrap
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    from dm3d_amd import _lib
    _lib.require_device()
    torch.cuda.set_device(0)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    return torch.device("cuda:0")


def _rel(a, ref):
    a, ref = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(ref).detach().cpu().double()
    return float((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _args(T, bs=1):
    return SimpleNamespace(timesteps=T, num_gpus=1, kernel_resize=False, bs=bs)


def test_config1_whole_chain_against_the_oracle(dev):
    """BASELINE config 1 in full: networks/dm3d.py U-Net, 16^3 x 4ch, B = 1, ALL 50 DDPM steps (dm3d.py:510-532) with the oracle's
    x_T and per-step noise injected; the final latent within 2e-3 absolute of rt.generate (values are clipped to [-1, 1] + noise)."""
    import dm3d_amd
    from dm3d_amd.networks import dm3d
    from oracle import ref_torch as rt
    cfg = dm3d_amd.UNetConfig(img_size=16, img_channels=4, conditional=False)
    W = dm3d_amd.synthetic_weights(cfg, seed=0)
    T, shape = 50, (1, 16, 16, 16, 4)
    m = dm3d.DiffusionModel(16, 1024, 4, None, _args(T), weights=W)
    g = torch.Generator().manual_seed(4)
    x_T = torch.randn(shape, generator=g)
    noises = torch.randn((T,) + shape, generator=g)
    got = m.generate(shape, x_T=x_T, noise=noises)
    traj = []
    ref = rt.generate({k: torch.from_numpy(v) for k, v in W.items()}, rt.UNetConfig(img_size=16, img_channels=4, conditional=False),
                      rt.Betas(T), T, x_T, noises, trajectory=traj)
    err = float((got.cpu() - ref).abs().max())
    print(f"config 1, 50 steps: max |x_0 - oracle| = {err:.3e}; |x_0| max {float(ref.abs().max()):.3f}")
    assert torch.isfinite(got).all() and err < 2e-3


@pytest.mark.parametrize("last_step", [3, 17])
def test_generate_partial_denoising_last_step(dev, last_step):
    """generate(shape, last_step=k > 0) (conditional_dm3d.py:559: `for i in reversed(range(last_step, timesteps))`) stops after step k:
    T - k steps, the last of them still WITH noise (i = k > 0).  Conditional model, 8^3 x 4ch, T = 20, B = 2."""
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    from oracle import ref_torch as rt
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    W = dm3d_amd.synthetic_weights(cfg, seed=2)
    T, shape = 20, (2, 8, 8, 8, 4)
    m = cdm.DiffusionModel(8, 1024, 4, None, _args(T, 2), weights=W)
    g = torch.Generator().manual_seed(11)
    x_T = torch.randn(shape, generator=g)
    noises = torch.randn((T,) + shape, generator=g)
    got = m.generate(shape, last_step=last_step, context_value=1, x_T=x_T, noise=noises)
    ref = rt.generate({k: torch.from_numpy(v) for k, v in W.items()}, rt.UNetConfig(img_size=8, img_channels=4), rt.Betas(T), T, x_T, noises,
                      last_step=last_step, context_value=1)
    err = float((got.cpu() - ref).abs().max())
    print(f"last_step={last_step}: {T - last_step} steps, max abs err {err:.3e}")
    assert err < 2e-3
    # the graph path takes the same number of steps: same x_T (seeded), T - k replays; finite and different from the full chain
    a = m.generate(shape, last_step=last_step, context_value=1, seed=5)
    b = m.generate(shape, last_step=0, context_value=1, seed=5)
    assert torch.isfinite(a).all() and not torch.equal(a, b)
    with pytest.raises(ValueError):
        m.generate(shape, last_step=T + 1, context_value=1)


@pytest.mark.parametrize("case", [
    dict(size=32, cin=64, cout=64, B=2),        # 32^3: 65 536 voxels per tap, K split over many workgroups (float atomics)
    dict(size=8, cin=256, cout=256, B=8),       # the reference's training latent: 8^3 x 256 channels
], ids=["32cube_64to64_B2", "8cube_256to256_B8"])
def test_wgrad_and_dgrad_in_the_atomic_ksplit_regime(dev, case):
    """dm3d_wgrad splits the voxel axis over workgroups and adds the partial sums with float atomics: the shapes where that matters
    (tools/train_bench.py's) against float64 autograd, bar 2e-5 of the largest reference entry; dgrad through the forward kernel too."""
    from dm3d_amd.train import Var
    from oracle import ref_torch as rt
    from test_gpu_train import _tiny_trainer, _param
    tr = _tiny_trainer(dev)
    g = torch.Generator().manual_seed(3)
    S, cin, cout, B = case["size"], case["cin"], case["cout"], case["B"]
    x = torch.randn(B, S, S, S, cin, generator=g)
    wk = torch.randn(3, 3, 3, cin, cout, generator=g) * (2.0 / (27 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    gout = torch.randn(B, S, S, S, cout, generator=g)
    xr, wr, br = (t.clone().double().requires_grad_(True) for t in (x, wk, bias))
    y = rt._conv3d(xr, wr, br, stride=1)
    y.backward(gout.double())
    _param(tr, "c.kernel", wk.numpy())
    _param(tr, "c.bias", bias.numpy())
    xv = Var(x.to(dev))
    out = tr.conv(xv, "c", 3)
    assert _rel(out.v, y.detach()) < 1e-5
    out.g = gout.to(dev)
    tr.backward()
    torch.cuda.synchronize()
    errs = dict(dx=_rel(xv.g, xr.grad), dw=_rel(tr.params["c.kernel"].g.reshape(wk.shape), wr.grad), db=_rel(tr.params["c.bias"].g, br.grad))
    print(case, errs)
    assert max(errs.values()) < 2e-5, errs
    # the atomic adds make the order of the partial sums run-dependent: two runs must still agree to rounding
    g1 = tr.params["c.kernel"].g.clone()
    tr.params["c.kernel"].g.zero_(); tr.params["c.bias"].g.zero_()
    xv2 = Var(x.to(dev))
    out2 = tr.conv(xv2, "c", 3)
    out2.g = gout.to(dev)
    tr.backward()
    torch.cuda.synchronize()
    assert _rel(tr.params["c.kernel"].g, g1) < 1e-5


def test_unconditional_public_train_step_and_test(dev, tmp_path, monkeypatch):
    """dm3d.DiffusionModel.train_step((images, _)) (dm3d.py:431-470) over two steps against the oracle's Adam trajectory, then
    .test(prefix) (dm3d.py:534-545): the saved .npy equals decoder(generate(...)) for the seeded chain."""
    import dm3d_amd
    from dm3d_amd.networks import dm3d
    from oracle import ref_torch as rt, ref_train as ot
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4, conditional=False)
    W = dm3d_amd.synthetic_weights(cfg, seed=6)
    T, B, lc = 12, 2, 4
    m = dm3d.DiffusionModel(8, 1024, 4, None, _args(T, B), weights=W)
    m.compile(loss="mse_sum", optimizer=SimpleNamespace(learning_rate=1e-4))
    g = torch.Generator().manual_seed(9)
    ocfg, ob = rt.UNetConfig(img_size=8, img_channels=4, conditional=False), rt.Betas(T)
    Wd = {k: torch.from_numpy(v).double() for k, v in W.items()}
    mom = {k: torch.zeros_like(v) for k, v in Wd.items() if ot.is_trainable(k)}
    vel = {k: torch.zeros_like(v) for k in mom for v in [mom[k]]}
    ref_losses, losses = [], []
    for step in (1, 2):
        lat = torch.randn(B, 8, 8, 8, 4, generator=g)
        noise = torch.randn(B, 8, 8, 8, 4, generator=g)
        t = torch.randint(0, T, (B,), generator=g)
        stats = {}
        lref, gref, _ = ot.loss_and_grads(Wd, ocfg, ob, lat.double(), t, noise.double(), None, B, lc, stats=stats)
        Wn, mom, vel = ot.adam_step(Wd, gref, mom, vel, step, 1e-4)
        Wd = {**Wd, **Wn, **ot.moving_update(Wd, stats)}
        ref_losses.append(float(lref))
        out = m.train_step((None, None), latents=lat, t=t, noise=noise)          # inputs = (images, _) for the unconditional model
        losses.append(out["loss"])
    assert np.allclose(losses, np.cumsum(ref_losses) / np.arange(1, 3), rtol=2e-5), (losses, ref_losses)
    st = m.network.state_dict()
    worst = max(float((torch.from_numpy(st[k]).double() - Wd[k]).abs().max()) for k in mom)
    print(f"unconditional train_step x2: max |w - w_ref| = {worst:.3e}")
    assert worst < 3e-5
    # test(): generate -> decode -> np.save.  A small stand-in decoder keeps the case fast (the VQ-VAE bracket has its own tests).
    calls = {}

    class _Dec:
        def __call__(self, z):
            calls["z"] = z.clone()
            return z[..., :1] * 2.0 + 1.0
    m.vqvae_trainer = SimpleNamespace(decoder=_Dec(), load_weights=lambda *_: None)
    monkeypatch.chdir(tmp_path)
    images = m.test("r3")
    saved = np.load(tmp_path / "generated_images_dm3d" / f"r3-{T}rsteps.npy")
    assert saved.shape == (10, 8, 8, 8, 1) and np.array_equal(saved, images.cpu().numpy())
    assert np.array_equal(saved, (calls["z"][..., :1] * 2.0 + 1.0).cpu().numpy()) and np.isfinite(saved).all()


def test_sampler_step_loop_checks_range_and_nan_at_chain_end(dev):
    """A chain driven through the public Sampler.step() (what bench.py does) reads the range flag after its last step: a NaN injected into
    the latent — `amax > limit` is false for it — must raise there, as must finish() on a chain stopped early."""
    import dm3d_amd
    from dm3d_amd import _lib
    from dm3d_amd.networks import conditional_dm3d as cdm
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    T = 4
    m = cdm.DiffusionModel(8, 1024, 4, None, _args(T, 2), weights=dm3d_amd.synthetic_weights(cfg, seed=1))
    smp = m.sampler((2, 8, 8, 8, 4), context_value=1, seed=3, use_graph=False)
    smp.reset()
    for _ in range(T):
        smp.step()                                       # a healthy chain: the end-of-chain check passes
    torch.cuda.synchronize()
    assert torch.isfinite(smp.plan.x).all()
    smp.reset()
    smp.step()
    smp.plan.x.view(-1)[5] = float("nan")
    with pytest.raises(_lib.Dm3dError):
        for _ in range(T - 1):
            smp.step()
    smp.reset()
    smp.step()
    smp.plan.x.view(-1)[7] = float("nan")
    with pytest.raises(_lib.Dm3dError):
        smp.finish()


def test_optimizer_state_survives_save_and_load(dev, tmp_path):
    """save_weights(.npz) of a trained model carries the Adam slots and the step count; a model built from it continues the SAME
    trajectory (bias correction at step 3, not 1) — the reference resumes from ModelCheckpoint(save_weights_only=True) checkpoints,
    which hold the optimizer slots.  The learning rate is re-read on every step."""
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    W = dm3d_amd.synthetic_weights(cfg, seed=1)
    T, B = 10, 2
    g = torch.Generator().manual_seed(2)
    batches = [(torch.randn(B, 8, 8, 8, 4, generator=g), torch.randn(B, 8, 8, 8, 4, generator=g), torch.randint(0, T, (B,), generator=g))
               for _ in range(3)]
    ctx = torch.tensor([[[1]], [[0]]])

    def run(model, which):
        for i in which:
            lat, noise, t = batches[i]
            model.train_step((None, None, ctx), latents=lat, t=t, noise=noise)

    a = cdm.DiffusionModel(8, 1024, 4, None, _args(T, B), weights=W)
    a.compile(optimizer=SimpleNamespace(learning_rate=3e-4))
    run(a, [0, 1])
    a.save_weights(str(tmp_path / "ck.npz"))
    a.optimizer.learning_rate = 1e-4                     # changed after compile(): the third step must use it
    run(a, [2])
    b = cdm.DiffusionModel(8, 1024, 4, None, _args(T, B), weights=W)
    b.compile(optimizer=SimpleNamespace(learning_rate=1e-4))
    b.load_weights(str(tmp_path / "ck.npz"))
    assert b._trainer is None and b._pending_optimizer is not None      # an inference-only load builds no training buffers
    b.generate((1, 8, 8, 8, 4), context_value=1, seed=1, steps=1)
    assert b._trainer is None
    assert b.trainer.step_count == 2 and b._pending_optimizer is None   # ... the slots land when the Trainer is first built
    run(b, [2])
    sa, sb = a.network.state_dict(), b.network.state_dict()
    worst = max(float(np.abs(sa[k] - sb[k]).max()) for k in sa)
    print(f"resumed vs uninterrupted after step 3: max |dw| = {worst:.3e}")
    assert worst < 1e-7
    c = cdm.DiffusionModel(8, 1024, 4, None, _args(T, B), weights={k: v for k, v in np.load(tmp_path / "ck.npz").items() if not k.startswith("optimizer/")})
    c.compile(optimizer=SimpleNamespace(learning_rate=1e-4))
    run(c, [2])                                          # without the slots Adam restarts at step 1: a different update
    sc = c.network.state_dict()
    assert max(float(np.abs(sa[k] - sc[k]).max()) for k in sa) > 1e-6
    # a checkpoint with partial slots fails at load time with a message, not with a KeyError inside the first train_step
    part = dict(np.load(tmp_path / "ck.npz"))
    del part[next(k for k in part if k.startswith("optimizer/v/"))]
    with pytest.raises(ValueError, match="optimizer state"):
        c.load_state_dict(part)
    # ... and leaves the model as it was: same weights, the Trainer and its Adam state still there (validate before mutating)
    assert c._trainer is not None and c.trainer.step_count == 1
    sc2 = c.network.state_dict()
    assert all(np.array_equal(sc[k], sc2[k]) for k in sc)
    # round 5: the TF-format writer carries the Adam slots and the step count too (OptimizerV2 layout: optimizer/iter + slot variables)
    a.save_weights(str(tmp_path / "tfck"))
    d = cdm.DiffusionModel(8, 1024, 4, None, _args(T, B), weights=W)
    d.compile(optimizer=SimpleNamespace(learning_rate=1e-4))
    d.load_weights(str(tmp_path / "tfck"))
    assert d._pending_optimizer is not None and int(d._pending_optimizer["optimizer/iter"]) == 3
    assert d.trainer.step_count == 3
    sd, sa2 = d.network.state_dict(), a.network.state_dict()
    assert all(np.array_equal(sd[k], sa2[k]) for k in sa2)
    oa, od = a.trainer.optimizer_state(), d.trainer.optimizer_state()
    assert set(oa) == set(od) and all(np.array_equal(np.asarray(oa[k]), np.asarray(od[k])) for k in oa)


def test_data_parallel_train_step_two_ranks(dev, tmp_path):
    """Two fresh processes share GPU 0 over gloo (RCCL refuses two ranks on one device), each runs train_step on HALF a batch with
    args.bs = the global batch: Trainer.allreduce_grads sums the gradients, averages the BatchNormalization moving statistics and sums
    the loss.  The weights after the step must equal the oracle's Adam step on the SUM of the two shards' gradients, each evaluated with
    its own replica's batch statistics (Keras per-replica BatchNormalization under MirroredStrategy, main_conditional_dm.py:87), and both
    ranks must end with identical weights and moving statistics."""
    code = textwrap.dedent("""
        import os, sys, json, numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, os.getcwd())
        from types import SimpleNamespace
        import dm3d_amd
        from dm3d_amd.networks import conditional_dm3d as cdm
        rank = int(os.environ["RANK"])
        dist.init_process_group("gloo", rank=rank, world_size=2)
        torch.cuda.set_device(0)
        cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
        W = dm3d_amd.synthetic_weights(cfg, seed=1)
        T, GB = 10, 4
        m = cdm.DiffusionModel(8, 1024, 4, None, SimpleNamespace(timesteps=T, num_gpus=2, kernel_resize=False, bs=GB), weights=W)
        m.compile(optimizer=SimpleNamespace(learning_rate=2e-4))
        g = torch.Generator().manual_seed(21)
        lat = torch.randn(GB, 8, 8, 8, 4, generator=g); noise = torch.randn(GB, 8, 8, 8, 4, generator=g)
        t = torch.randint(0, T, (GB,), generator=g); ctx = torch.randint(0, 2, (GB, 1, 1), generator=g)
        sl = slice(2 * rank, 2 * rank + 2)
        out = m.train_step((None, None, ctx[sl]), latents=lat[sl], t=t[sl], noise=noise[sl])
        st = m.network.state_dict()
        np.savez(os.environ["OUT"] + f"/rank{rank}.npz", loss=np.float64(out["loss"]), **st)
        dist.barrier()
        dist.destroy_process_group()
        print("rank done", rank)
    """)
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OUT=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    import dm3d_amd
    from oracle import ref_torch as rt, ref_train as ot
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    W = dm3d_amd.synthetic_weights(cfg, seed=1)
    T, GB, lc = 10, 4, 4
    g = torch.Generator().manual_seed(21)
    lat = torch.randn(GB, 8, 8, 8, 4, generator=g); noise = torch.randn(GB, 8, 8, 8, 4, generator=g)
    t = torch.randint(0, T, (GB,), generator=g); ctx = torch.randint(0, 2, (GB, 1, 1), generator=g)
    ocfg, ob = rt.UNetConfig(img_size=8, img_channels=4), rt.Betas(T)
    Wd = {k: torch.from_numpy(v).double() for k, v in W.items()}
    gsum, loss_sum, moving = None, 0.0, []
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        stats = {}
        lref, gref, _ = ot.loss_and_grads(Wd, ocfg, ob, lat[sl].double(), t[sl], noise[sl].double(), ctx[sl], GB, lc, stats=stats)
        loss_sum += float(lref)
        gsum = gref if gsum is None else {k: gsum[k] + gref[k] for k in gref}
        moving.append(ot.moving_update(Wd, stats))
    mom = {k: torch.zeros_like(v) for k, v in Wd.items() if ot.is_trainable(k)}
    Wn, _, _ = ot.adam_step(Wd, gsum, mom, {k: torch.zeros_like(v) for k, v in mom.items()}, 1, 2e-4)
    got = [dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(2)]
    for k in got[0]:
        assert np.array_equal(got[0][k], got[1][k]), f"ranks disagree on {k}"
    assert abs(float(got[0]["loss"]) - loss_sum) <= 2e-5 * abs(loss_sum), (float(got[0]["loss"]), loss_sum)
    worst = 0.0
    for k, v in Wn.items():
        sel = gsum[k].abs() > 1e-3 * gsum[k].abs().max().clamp_min(1e-30)       # Adam's first step is lr * sign(g): skip the near-zero gradients
        if sel.any():
            worst = max(worst, float((torch.from_numpy(got[0][k]).double() - v).abs()[sel].max()))
    print(f"2-rank train_step: loss {float(got[0]['loss']):.6f} (oracle {loss_sum:.6f}); max |w - w_ref| = {worst:.3e} for lr 2e-4")
    assert worst < 2e-5
    for k in moving[0]:
        ref = 0.5 * (moving[0][k] + moving[1][k])                                 # MEAN over the replicas
        assert _rel(got[0][k], ref) < 1e-4, k

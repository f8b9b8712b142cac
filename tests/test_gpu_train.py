"""GPU parity of DiffusionModel.train_step (reference networks/conditional_dm3d.py:471-510) against the CPU oracle
(oracle/ref_train.py: the restated forward with training-mode BatchNormalization, torch.autograd as the gradient reference,
Keras-default Adam).  Staged as the layers are: single layers' gradients, the training forward and loss, all gradients of the
network, one optimizer step, the public train_step.  Tolerances are written next to each assert."""
import os
from types import SimpleNamespace

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
    a, ref = torch.as_tensor(a).double().cpu(), torch.as_tensor(ref).double().cpu()
    return float((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _args(T, bs):
    return SimpleNamespace(timesteps=T, num_gpus=1, kernel_resize=False, bs=bs)


def _tiny_trainer(dev):
    import dm3d_amd
    from dm3d_amd.train import Trainer
    cfg = dm3d_amd.UNetConfig(img_size=4, img_channels=4, widths=(16,), has_attention=(False, False), num_res_blocks=1, first_conv_channels=16)
    tr = Trainer(cfg, dm3d_amd.synthetic_weights(cfg, seed=0), dev)
    tr.tape, tr._cache, tr.update_moving = [], {}, False
    return tr


def _param(tr, name, arr):
    from dm3d_amd.train import Param
    w = torch.from_numpy(np.ascontiguousarray(arr, np.float32)).to(tr.device)
    tr.params[name] = Param(name, arr.shape, w.reshape(-1), torch.zeros(w.numel(), device=tr.device))
    return tr.params[name]


LAYER_TOL = 2e-5     # per-layer gradient parity (relative to the largest reference entry)


@pytest.mark.parametrize("case", [
    dict(k=3, stride=1, size=(6, 5, 8), cin=8, cout=12),
    dict(k=3, stride=1, size=(8, 8, 8), cin=96, cout=64),          # several Cin chunks, one full 64-wide tile
    dict(k=3, stride=2, size=(8, 8, 8), cin=16, cout=16),          # TF SAME 0/1
    dict(k=3, stride=2, size=(5, 7, 6), cin=8, cout=20),           # odd sizes: SAME pads 1/1
    dict(k=1, stride=1, size=(4, 4, 4), cin=72, cout=40),
], ids=["k3_ragged", "k3_96to64", "k3s2_even", "k3s2_odd", "k1"])
def test_conv_gradients(dev, case):
    """dL/dx (forward kernel on the flipped kernel), dL/dW (dm3d_wgrad), dL/db, dL/dvec, dL/dres of Conv3D against autograd."""
    import torch.nn.functional as F
    from dm3d_amd.train import Var
    from oracle import ref_torch as rt
    tr = _tiny_trainer(dev)
    g = torch.Generator().manual_seed(1)
    k, s, (D, H, W), cin, cout = case["k"], case["stride"], case["size"], case["cin"], case["cout"]
    B = 2
    x = torch.randn(B, D, H, W, cin, generator=g)
    wk = torch.randn(k, k, k, cin, cout, generator=g) * 0.1
    bias = torch.randn(cout, generator=g)
    vec = torch.randn(B, cout, generator=g)
    od, oh, ow = (-(-D // s), -(-H // s), -(-W // s))
    res = torch.randn(B, od, oh, ow, cout, generator=g)
    gout = torch.randn(B, od, oh, ow, cout, generator=g)
    xr, wr, br, vr, rr = (t.clone().double().requires_grad_(True) for t in (x, wk, bias, vec, res))
    y = rt._conv3d(xr, wr, br, stride=s) + vr[:, None, None, None, :] + rr
    y.backward(gout.double())
    _param(tr, "c.kernel", wk.numpy())
    _param(tr, "c.bias", bias.numpy())
    xv, vv, rv = Var(x.to(dev)), Var(vec.to(dev)), Var(res.to(dev))
    out = tr.conv(xv, "c", k, stride=s, vec=vv, res=rv)
    assert _rel(out.v, y.detach()) < 1e-5
    out.g = gout.to(dev)
    tr.backward()
    torch.cuda.synchronize()
    errs = dict(dx=_rel(xv.g, xr.grad), dw=_rel(tr.params["c.kernel"].g.reshape(wk.shape), wr.grad), db=_rel(tr.params["c.bias"].g, br.grad),
                dvec=_rel(vv.g, vr.grad), dres=_rel(rv.g, rr.grad))
    print(errs)
    assert max(errs.values()) < LAYER_TOL, errs


def test_bn_act_gradients(dev):
    """act(BatchNormalization(training=True)(concat[x1, x2])): forward, moving statistics, dx1, dx2, dgamma, dbeta."""
    from dm3d_amd import _lib
    from dm3d_amd.train import Var
    tr = _tiny_trainer(dev)
    tr.update_moving = True
    g = torch.Generator().manual_seed(2)
    B, S, c1, c2 = 3, 4, 8, 12
    x1 = torch.randn(B, S, S, S, c1, generator=g) * 2 + 0.5
    x2 = torch.randn(B, S, S, S, c2, generator=g)
    gamma = torch.rand(c1 + c2, generator=g) + 0.5
    beta = torch.randn(c1 + c2, generator=g) * 0.1
    gout = torch.randn(B, S, S, S, c1 + c2, generator=g)
    for act, fn in ((_lib.ACT_SILU, torch.nn.functional.silu), (_lib.ACT_NONE, lambda v: v)):
        a, b2, ga, be = (t.clone().double().requires_grad_(True) for t in (x1, x2, gamma, beta))
        cat = torch.cat([a, b2], -1)
        mean, var = cat.mean((0, 1, 2, 3)), cat.var((0, 1, 2, 3), unbiased=False)
        y = fn((cat - mean) / torch.sqrt(var + 1e-3) * ga + be)
        y.backward(gout.double())
        _param(tr, "n.gamma", gamma.numpy())
        _param(tr, "n.beta", beta.numpy())
        tr.moving["n.mean"] = torch.zeros(c1 + c2, device=dev)
        tr.moving["n.var"] = torch.ones(c1 + c2, device=dev)
        v1, v2 = Var(x1.to(dev)), Var(x2.to(dev))
        out = tr.bn_act(v1, v2, "n", act)
        assert _rel(out.v, y.detach()) < 1e-5
        out.g = gout.to(dev)
        tr.backward()
        torch.cuda.synchronize()
        errs = dict(dx1=_rel(v1.g, a.grad), dx2=_rel(v2.g, b2.grad), dgamma=_rel(tr.params["n.gamma"].g, ga.grad),
                    dbeta=_rel(tr.params["n.beta"].g, be.grad))
        print(act, errs)
        assert max(errs.values()) < LAYER_TOL, errs
        n = B * S ** 3
        assert _rel(tr.moving["n.mean"], 0.01 * mean.detach()) < 1e-5                              # 0*0.99 + mean*0.01
        assert _rel(tr.moving["n.var"], 0.99 + 0.01 * var.detach() * n / (n - 1)) < 1e-6          # Bessel-corrected, as tf.nn.fused_batch_norm


def test_dense_layernorm_attention_gradients(dev):
    from dm3d_amd import _lib
    from dm3d_amd.train import Var
    tr = _tiny_trainer(dev)
    g = torch.Generator().manual_seed(3)
    B, L, u = 2, 64, 32
    M = B * L
    x = torch.randn(M, u, generator=g)
    ctx = torch.randn(M, u, generator=g)
    W = {n: torch.randn(u, u, generator=g) * 0.2 for n in ("q", "k", "v", "p")}
    bb = {n: torch.randn(u, generator=g) * 0.1 for n in ("q", "k", "v", "p")}
    lg, lb = torch.rand(u, generator=g) + 0.5, torch.randn(u, generator=g) * 0.1
    gout = torch.randn(M, u, generator=g)
    leaves = {k: v.clone().double().requires_grad_(True) for k, v in dict(x=x, ctx=ctx, lg=lg, lb=lb, **{f"W{n}": W[n] for n in W},
                                                                            **{f"b{n}": bb[n] for n in bb}).items()}
    n1 = torch.nn.functional.layer_norm(leaves["x"], (u,), leaves["lg"], leaves["lb"], 1e-3)
    q = torch.nn.functional.silu(n1 @ leaves["Wq"] + leaves["bq"])
    k = ctx_k = leaves["ctx"] @ leaves["Wk"] + leaves["bk"]
    v = leaves["ctx"] @ leaves["Wv"] + leaves["bv"]
    s = torch.softmax(torch.einsum("blc,bLc->blL", q.reshape(B, L, u), k.reshape(B, L, u)) * u ** -0.5, -1)
    o = torch.einsum("blL,bLc->blc", s, v.reshape(B, L, u)).reshape(M, u)
    y = torch.relu(o @ leaves["Wp"] + leaves["bp"]) + leaves["x"]
    y.backward(gout.double())
    for n in W:
        _param(tr, f"{n}.kernel", W[n].numpy())
        _param(tr, f"{n}.bias", bb[n].numpy())
    _param(tr, "ln.gamma", lg.numpy())
    _param(tr, "ln.beta", lb.numpy())
    xv, cv = Var(x.to(dev)), Var(ctx.to(dev))
    n1v = tr.layernorm(xv, "ln")
    qv = tr.dense(n1v, "q", act=_lib.ACT_SILU)
    kv, vv = tr.dense(cv, "k"), tr.dense(cv, "v")
    ov = tr.attention(qv, kv, vv, B, L, L, u)
    out = tr.dense(ov, "p", act=_lib.ACT_RELU, res=xv)
    assert _rel(out.v, y.detach()) < 1e-5
    out.g = gout.to(dev)
    tr.backward()
    torch.cuda.synchronize()
    errs = dict(dx=_rel(xv.g, leaves["x"].grad), dctx=_rel(cv.g, leaves["ctx"].grad), dlg=_rel(tr.params["ln.gamma"].g, leaves["lg"].grad),
                dlb=_rel(tr.params["ln.beta"].g, leaves["lb"].grad))
    for n in W:
        errs[f"dW{n}"] = _rel(tr.params[f"{n}.kernel"].g.reshape(u, u), leaves[f"W{n}"].grad)
        errs[f"db{n}"] = _rel(tr.params[f"{n}.bias"].g, leaves[f"b{n}"].grad)
    print(errs)
    errs.pop("dbk")          # softmax is invariant to a key bias: the reference gradient is rounding noise
    assert max(errs.values()) < LAYER_TOL, errs


def _compare_grads(got, ref, tol, label=""):
    """per tensor: max |g - ref| <= tol * max(|ref| of that tensor, 1e-3 * the largest gradient entry overall).  (Some gradients are
    analytically zero — e.g. a key bias under softmax — so a purely per-tensor relative bar would compare rounding noise.)"""
    gmax = max(float(v.abs().max()) for v in ref.values())
    worst = ("", 0.0)
    for name, r in ref.items():
        scale = max(float(r.abs().max()), 1e-3 * gmax)
        e = float((torch.as_tensor(got[name]).double() - r.double()).abs().max()) / scale
        if e > worst[1]:
            worst = (name, e)
    print(f"{label} worst gradient error {worst[1]:.3e} at {worst[0]} (largest gradient entry {gmax:.3e})")
    assert worst[1] < tol, worst
    return worst


@pytest.mark.parametrize("cond", [True, False], ids=["conditional", "unconditional"])
def test_training_forward_loss_and_all_gradients(dev, cond):
    """8^3 x 4ch, B=2, real widths: training-mode forward (batch statistics) -> loss within 1e-5 of the oracle; every trainable
    weight's gradient within 1e-4 (relative, see _compare_grads) of torch.autograd on the float64 oracle; BatchNormalization
    moving statistics as Keras updates them."""
    import dm3d_amd
    from dm3d_amd.train import Trainer
    from oracle import ref_torch as rt, ref_train as ot
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4, conditional=cond)
    W = dm3d_amd.synthetic_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(2, 8, 8, 8, 4, generator=g)
    noise = torch.randn(2, 8, 8, 8, 4, generator=g)
    t = torch.tensor([3, 40])
    ctx = torch.tensor([[[1]], [[0]]]) if cond else None
    T, gbs, lc = 50, 2, 4
    ocfg = rt.UNetConfig(img_size=8, img_channels=4, conditional=cond)
    ob = rt.Betas(T)
    stats = {}
    W64 = {k: torch.from_numpy(v).double() for k, v in W.items()}
    loss_ref, grads_ref, pred_ref = ot.loss_and_grads(W64, ocfg, ob, lat.double(), t, noise.double(), ctx, gbs, lc, stats=stats)
    mov_ref = ot.moving_update(W64, stats)
    tr = Trainer(cfg, W, dev)
    b = dm3d_amd.Betas(T)
    tab = b.device_tables(dev)
    from dm3d_amd.betas import BETAS_FIELDS
    betas = (tab[BETAS_FIELDS.index("sqrt_alpha_bar")], tab[BETAS_FIELDS.index("sqrt_one_minus_alpha_bar")])
    loss, pred = tr.loss_and_grad(lat.to(dev), t, noise.to(dev), None if ctx is None else ctx.reshape(-1).numpy(), betas, T, gbs, lc)
    torch.cuda.synchronize()
    lerr = abs(float(loss.item()) - float(loss_ref)) / float(loss_ref)
    print(f"loss {float(loss.item()):.8f} vs oracle {float(loss_ref):.8f} (rel {lerr:.2e}); pred rel err {_rel(pred, pred_ref):.2e}")
    assert lerr < 1e-5 and _rel(pred, pred_ref) < 1e-4
    got = {k: torch.from_numpy(v) for k, v in tr.grads().items()}
    assert set(got) == set(grads_ref)
    _compare_grads(got, grads_ref, 1e-4, "all gradients:")
    st = tr.state_dict()
    for name, ref in mov_ref.items():
        assert _rel(st[name], ref) < 1e-5, name
    # one Adam step (Keras defaults) from zero moments
    zeros = {k: torch.zeros_like(v) for k, v in grads_ref.items()}
    Wn, _, _ = ot.adam_step(W64, grads_ref, zeros, zeros, 1, 1e-4)
    tr.adam_step()
    st = tr.state_dict()
    worst = max(float((torch.from_numpy(st[k]).double() - Wn[k]).abs().max()) for k in Wn)
    print(f"after one Adam step: max |w - w_ref| = {worst:.3e} (lr 1e-4)")
    # every weight moves by ~lr = 1e-4 in the first step (m/sqrt(v) = sign(g)): 1e-5 absolute = 10 % of a step would catch a sign or
    # scaling error; entries whose gradient is rounding noise may legitimately differ by a whole step, so compare where |g| matters
    for k in Wn:
        sel = grads_ref[k].abs() > 1e-3 * grads_ref[k].abs().max().clamp_min(1e-30)
        if sel.any():
            d = (torch.from_numpy(st[k]).double() - Wn[k]).abs()[sel].max()
            assert float(d) < 1e-5, (k, float(d))


def test_train_step_public_api(dev):
    """DiffusionModel.train_step((images, mask, context)) with injected latents / t / noise: loss dict as Keras returns it, loss tracker
    mean, weights follow the oracle's Adam trajectory over three steps, and the sampling network sees the trained weights."""
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    from oracle import ref_torch as rt, ref_train as ot
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    W = dm3d_amd.synthetic_weights(cfg, seed=1)
    T, B, lc = 20, 2, 4
    m = cdm.DiffusionModel(8, 1024, 4, None, _args(T, B), weights=W)
    m.compile(loss="mse_sum", optimizer=SimpleNamespace(learning_rate=2e-4))
    g = torch.Generator().manual_seed(5)
    ocfg, ob = rt.UNetConfig(img_size=8, img_channels=4), rt.Betas(T)
    Wd = {k: torch.from_numpy(v).double() for k, v in W.items()}
    mom = {k: torch.zeros_like(v) for k, v in Wd.items() if ot.is_trainable(k)}
    vel = {k: torch.zeros_like(v) for k, v in mom.items()}
    losses_ref, losses = [], []
    for step in range(1, 4):
        lat = torch.randn(B, 8, 8, 8, 4, generator=g)
        noise = torch.randn(B, 8, 8, 8, 4, generator=g)
        t = torch.randint(0, T, (B,), generator=g)
        ctx = torch.randint(0, 2, (B, 1, 1), generator=g)
        stats = {}
        lref, gref, _ = ot.loss_and_grads(Wd, ocfg, ob, lat.double(), t, noise.double(), ctx, B, lc, stats=stats)
        Wn, mom, vel = ot.adam_step(Wd, gref, mom, vel, step, 2e-4)
        Wd = {**Wd, **Wn, **ot.moving_update(Wd, stats)}
        losses_ref.append(float(lref))
        out = m.train_step((None, None, ctx), latents=lat, t=t, noise=noise)
        losses.append(out["loss"])
        assert set(out) == {"loss"}
    running = np.cumsum(losses_ref) / np.arange(1, 4)            # keras.metrics.Mean: train_step returns the running mean
    print("loss tracker", losses, "oracle running mean", running.tolist())
    assert np.allclose(losses, running, rtol=2e-5)
    st = m.network.state_dict()                                  # syncs the trained weights back into the sampling network
    worst = 0.0
    for k in mom:
        sel = mom[k].abs() > 1e-3 * mom[k].abs().max().clamp_min(1e-30)
        if sel.any():
            worst = max(worst, float((torch.from_numpy(st[k]).double() - Wd[k]).abs()[sel].max()))
    print(f"after 3 steps: max |w - w_ref| = {worst:.3e} (3 steps of 2e-4)")
    assert worst < 3e-5
    for k in Wd:
        if k.endswith((".mean", ".var")):
            assert _rel(st[k], Wd[k]) < 1e-4, k
    # the sampler now runs on the trained weights
    x = torch.randn(B, 8, 8, 8, 4, generator=g)
    tt, cc = torch.tensor([1, 7]), torch.tensor([[[1]], [[0]]])
    eps = m.network([x.to(dev), tt, cc])
    ref = rt.unet_forward({k: v.float() for k, v in Wd.items()}, ocfg, x, tt, cc)
    assert _rel(eps, ref) < 1e-3
    # network(..., training=True) outside train_step: batch statistics
    pt = m.network([x.to(dev), tt, cc], training=True)
    rt_train = ot.unet_forward_train({k: v.float() for k, v in Wd.items()}, ocfg, x, tt, cc)
    assert _rel(pt, rt_train) < 5e-4                         # (float32 oracle on float32-rounded trained weights)
    with pytest.raises(ValueError):
        m.train_step((None, None, cc), latents=torch.zeros(B, 4, 8, 8, 4), t=tt, noise=torch.zeros(B, 4, 8, 8, 4))


def test_train_step_from_images_reduces_the_loss(dev):
    """The whole reference path: images [b, 16S, 16S, 16S, 1] -> frozen encoder + quantizer -> q_sample -> network(training=True) -> loss ->
    Adam; random t / noise drawn inside.  Trained repeatedly on one batch the (noisy) loss falls."""
    import dm3d_amd
    from dm3d_amd.networks import conditional_dm3d as cdm
    S = 8                                                                              # the reference's latent size: 128^3 images
    m = cdm.DiffusionModel(S, 64, 8, None, _args(10, 2), weights=None)                 # reference constructor; Keras-style initial weights
    m.compile(loss=None, optimizer=1e-3)
    g = torch.Generator().manual_seed(7)
    images = torch.rand(2, 16 * S, 16 * S, 16 * S, 1, generator=g).to(dev)
    ctx = torch.tensor([[[0]], [[1]]])
    lat = m.encode_latents(images)
    assert tuple(lat.shape) == (2, S, S, S, 8)
    first = m.train_step((images, None, ctx))["loss"]
    assert np.isfinite(first) and first > 0
    m.loss_tracker.reset_state()
    fixed_t, fixed_noise = torch.tensor([2, 7]), torch.randn(2, S, S, S, 8, generator=g)
    seq = []
    for _ in range(12):
        m.loss_tracker.reset_state()
        seq.append(m.train_step((images, None, ctx), t=fixed_t, noise=fixed_noise)["loss"])
    print("loss on a fixed batch:", [f"{v:.4g}" for v in seq])
    # (Keras initialises the last conv of every block at ~0, so the first steps mostly grow those kernels: a steady, modest fall)
    assert all(np.isfinite(seq)) and seq[-1] < 0.95 * seq[0] and min(seq[6:]) < min(seq[:6])

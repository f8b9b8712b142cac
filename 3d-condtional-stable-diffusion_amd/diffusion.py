"""``DiffusionModel`` — the reference's DDPM wrapper around the U-Net, on the dm3d HIP kernels.

reference: networks/conditional_dm3d.py:418-594 (conditional) and networks/dm3d.py:379-545 (unconditional).  Same
constructor, attributes and method signatures; tensors are PyTorch device tensors (NDHWC float32) instead of tf.Tensor.
Keyword-only extensions: ``x_T=`` / ``noise=`` inject the random draws (parity tests), ``seed=``
selects the in-kernel Philox stream, ``use_graph=`` toggles HIP-graph replay of the step.

The sampling loop (:559-573) runs with no host synchronisation: the step index lives in device memory, one step
(U-Net forward + posterior update + index decrement) is captured once into a HIP graph and replayed T times.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import DdpmDesc, check, lib
from .betas import BETAS_FIELDS, Betas
from .unet import UNet
from .weights import UNetConfig


class _LossTracker:
    """keras.metrics.Mean(name="loss") stand-in (conditional_dm3d.py:467, 507-515)."""

    def __init__(self, name="loss"):
        self.name, self.total, self.count = name, 0.0, 0

    def update_state(self, v):
        self.total += float(v)
        self.count += 1

    def result(self):
        return self.total / max(self.count, 1)

    def reset_state(self):
        self.total, self.count = 0.0, 0


class DiffusionModel:
    conditional = True

    def __init__(self, latent_size, num_embed, latent_channels, vqvae_load_ckpt, args, *, device="cuda", weights=None,
                 seed=0, precision=None, norm="batch"):
        # conditional_dm3d.py:420-469.  ``args`` is any object with .timesteps .num_gpus .kernel_resize .bs
        self.timesteps = int(args.timesteps)
        self.b = Betas(self.timesteps)
        self.lc = latent_channels
        # The VQ-VAE bracket (networks/vqvae3d_monai.py; conditional_dm3d.py:425-460) is built lazily on first use of
        # .vqvae_trainer / .encoder / .quantizer / .decoder — Keras, too, creates its weights only on the first call.
        self.num_embed = num_embed
        self.vqvae_load_ckpt = vqvae_load_ckpt
        self._vqvae = None
        self._kernel_resize = getattr(args, "kernel_resize", False)
        self._precision = precision
        self.network = UNet(
            UNetConfig(img_size=latent_size, img_channels=latent_channels, widths=[64, 128, 256],
                       has_attention=[False, False, True, True], conditional=self.conditional, norm=norm),
            device=device, weights=weights, seed=seed, precision=precision)
        self.loss_tracker = _LossTracker("loss")
        self.num_gpus = getattr(args, "num_gpus", 1)
        self.global_bs = getattr(args, "bs", 1)
        self.device = self.network.device
        self._graphs = {}
        self._stream = None
        self._trainer = None
        self._trainer_dirty = False
        self._pending_optimizer = None          # Adam slots of a loaded checkpoint, applied when the Trainer is first built
        self.network._before_use = self._sync_from_trainer
        self.network._training_engine = self._engine_for_training_forward

    # -- the autoencoder bracket --------------------------------------------------------------------------------------
    @property
    def vqvae_trainer(self):
        """VQVAE(in=1, out=1, channels (32,64,128,256), 5 residual layers, 4 x (stride 2, k4), num_embeddings=num_embed,
        embedding_dim=latent_channels) as in conditional_dm3d.py:425-449; input edge = 16 x latent_size (128 for the
        reference's latent_size 8)."""
        if self._vqvae is None:
            from .networks.vqvae3d_monai import VQVAE
            self._vqvae = VQVAE(
                in_channels=1, out_channels=1, num_channels=(32, 64, 128, 256), num_res_channels=(32, 64, 128, 256),
                num_res_layers=5, downsample_parameters=((2, 4, 1, "same"),) * 4,
                upsample_parameters=((2, 4, 1, "same", 0),) * 4, num_embeddings=self.num_embed, embedding_dim=self.lc,
                dropout=None, num_gpus=self.num_gpus, kernel_resize=self._kernel_resize,
                input_size=16 * self.network.cfg.img_size, device=self.device, precision=self._precision)
            if self.vqvae_load_ckpt is not None:
                print("Loading VQVAE weights")
                self._vqvae.load_weights(self.vqvae_load_ckpt)
        return self._vqvae

    @vqvae_trainer.setter
    def vqvae_trainer(self, v):
        self._vqvae = v

    @property
    def encoder(self):
        return self.vqvae_trainer.encoder

    @property
    def quantizer(self):
        return self.vqvae_trainer.quantizer

    @property
    def decoder(self):
        return self.vqvae_trainer.decoder

    # -- Keras-model conveniences the reference's drivers touch ----------------------------------------------------
    @property
    def metrics(self):
        return [self.loss_tracker]

    def compile(self, loss=None, optimizer=None):
        self.loss, self.optimizer = loss, optimizer

    def load_state_dict(self, sd, strict=True):
        """Weights by name; ``optimizer/...`` entries (save_weights of a trained model) restore the Adam slots and step count, so a
        resumed run continues the bias correction where it stopped; without them the optimizer starts afresh."""
        # validated BEFORE anything is touched: a checkpoint with partial slots fails here with the model as it was (new weights with the
        # old Adam state gone would be a half-loaded model)
        opt = {k: v for k, v in sd.items() if k.startswith("optimizer/")}
        if opt:
            missing = [k for k in ["optimizer/iter"] + [f"optimizer/{slot}/{n}" for n in self._trainable_names() for slot in ("m", "v")] if k not in opt]
            if missing:
                raise ValueError(f"checkpoint carries optimizer state but {len(missing)} entries are missing (first: {missing[:3]}); "
                                 "drop every optimizer/ entry to load the weights alone")
        self._sync_from_trainer()                                  # a non-strict load fills missing names from the CURRENT (trained) weights
        self.network.load_state_dict({k: v for k, v in sd.items() if not k.startswith("optimizer/")}, strict)
        self._drop_graphs()
        self._trainer, self._trainer_dirty = None, False          # Adam moments belong to the weights they were built for
        # the slots wait until a Trainer exists (an inference-only load builds none: theta, gradients and moments are four copies of the weights)
        self._pending_optimizer = opt or None

    def _trainable_names(self):
        """Names of the parameters Adam updates (everything but the BatchNormalization moving statistics)."""
        from .train import is_trainable
        return [n for n in self.network.spec if is_trainable(n)]

    def load_weights(self, path, root=("network",)):
        """keras ``model.load_weights(ckpt)`` (main_conditional_dm.py:207-213): reads the U-Net from a TF2 checkpoint prefix
        saved by the reference (``root``: where the U-Net sits in the saved object, ("network",) for a DiffusionModel
        checkpoint, () for ``network.save_weights``), or from an .npz state dict.  If the checkpoint also holds the
        autoencoder under ``vqvae_trainer`` it is loaded too."""
        if str(path).endswith(".npz"):
            self.load_state_dict(dict(np.load(path)))
            return
        from . import tf_checkpoint as tc
        # (with the Adam slots and step count of a compiled model's checkpoint, if it carries them: the reference resumes training with
        # model.load_weights(<epoch>.ckpt), main_conditional_dm.py:174-183)
        self.load_state_dict(tc.load_unet_state(str(path), self.network.cfg, root=tuple(root), with_optimizer=True))
        rd = tc.BundleReader(str(path))
        if any(k.startswith("vqvae_trainer/") for k in rd.entries):
            self.vqvae_trainer.load_weights(path, root=("vqvae_trainer",))

    def save_weights(self, path, root=("network",)):
        self._sync_from_trainer()
        # the Adam slots and step count travel with the weights, as in the reference's save_weights_only TF checkpoints of a compiled model
        opt = self._trainer.optimizer_state() if self._trainer is not None and self._trainer.step_count > 0 else (getattr(self, "_pending_optimizer", None) or {})
        if str(path).endswith(".npz"):
            np.savez(path, **self.network.state_dict(), **opt)
            return
        from . import tf_checkpoint as tc
        tc.save_unet_checkpoint(str(path), self.network.state_dict(), self.network.cfg, root=tuple(root), optimizer=opt or None)

    def _drop_graphs(self):
        for g in self._graphs.values():
            lib().dm3d_graph_destroy(g[0])
        self._graphs = {}

    def __del__(self):
        try:
            self._drop_graphs()
        except Exception:
            pass

    # -- a15: train_step --------------------------------------------------------------------------------------------------
    def _learning_rate(self) -> float:
        """``compile(optimizer=keras.optimizers.Adam(learning_rate=args.lr))`` (main_conditional_dm.py:153): a float, an object with
        ``learning_rate`` / ``lr``, or nothing (the reference's default --lr 1e-4, main_conditional_dm.py:228)."""
        opt = getattr(self, "optimizer", None)
        if isinstance(opt, (int, float)):
            return float(opt)
        for attr in ("learning_rate", "lr"):
            if opt is not None and hasattr(opt, attr):
                return float(getattr(opt, attr))
        return 1e-4

    @property
    def trainer(self):
        """The training engine (train.py), built on first use from the network's current weights."""
        if self._trainer is None:
            from .train import Trainer
            self._trainer = Trainer(self.network.cfg, self.network.state_dict(), self.device, lr=self._learning_rate())
            if getattr(self, "_pending_optimizer", None):
                self._trainer.load_optimizer_state(self._pending_optimizer)
                self._pending_optimizer = None
        return self._trainer

    def _engine_for_training_forward(self):
        """``network(..., training=True)`` outside train_step runs on the model's own Trainer when it has one, so the moving statistics it
        updates are the ones the next train_step continues from (in Keras both are the same variables)."""
        if self._trainer is None:
            return None
        self._trainer_dirty = True
        return self._trainer

    def _sync_from_trainer(self):
        """Weights changed by train_step flow back into the sampling network (folded norms, packed images, tables) before it runs."""
        if self._trainer is not None and self._trainer_dirty:
            self._trainer_dirty = False
            self.network.load_state_dict(self._trainer.state_dict())
            self._drop_graphs()

    def train_step(self, inputs, *, t=None, noise=None, latents=None):
        """conditional_dm3d.py:471-510: ``inputs = (images, mask, context)`` ((images, _) for the unconditional model, dm3d.py:431-433).
        images [b, 16S, 16S, 16S, 1] go through the frozen encoder + quantizer (:478); t ~ U{0..T-1} (:474-476), noise ~ N(0,1) (:481),
        q_sample (:484-490), the network with training=True (:493), loss = MSE_SUM / (global_bs * lc^4) (:496-499), Adam (:501-504),
        loss tracker (:507-510).  Keyword-only extensions: ``t`` / ``noise`` inject the random draws (parity tests), ``latents`` skips the
        autoencoder (pre-encoded latents [b, S, S, S, lc])."""
        if self.conditional:
            images, _, context = inputs
        else:
            images, _ = inputs
            context = None
        dev, cfg, T = self.device, self.network.cfg, self.timesteps
        if latents is None and images is None:
            raise ValueError("train_step needs images (inputs[0]) or pre-encoded latents=")
        _lib.require_device()
        if latents is None:
            latents = self.encode_latents(torch.as_tensor(images, dtype=torch.float32).to(dev))
        latents = torch.as_tensor(latents, dtype=torch.float32).to(dev).contiguous()
        B = latents.shape[0]
        want = (cfg.img_size,) * 3 + (cfg.img_channels,)
        if latents.dim() != 5 or tuple(latents.shape[1:]) != want:
            raise ValueError(f"latents must be [b,{','.join(map(str, want))}], got {tuple(latents.shape)}")
        if t is None:
            t = torch.from_numpy(np.random.default_rng(self.fresh_seed()).integers(0, T, size=B))
        t = torch.as_tensor(t).reshape(-1).to(torch.int64)
        if t.numel() != B or int(t.min()) < 0 or int(t.max()) >= T:
            raise ValueError("t must hold one index in [0, timesteps) per sample")
        if noise is None:
            noise = torch.empty_like(latents)
            check(lib().dm3d_randn(noise.data_ptr(), noise.numel(), self.fresh_seed(), 0x7ffffffe, torch.cuda.current_stream().cuda_stream), "randn")
        noise = torch.as_tensor(noise, dtype=torch.float32).to(dev).contiguous()
        if noise.shape != latents.shape:
            raise ValueError("noise must have the latents' shape")
        ids = None
        if self.conditional:
            ids = self._context_ids(context, B)
        tr = self.trainer
        tr.lr = self._learning_rate()               # re-read every step: compile() / optimizer.learning_rate may have changed
        tab = self.b.device_tables(dev)
        betas = (tab[BETAS_FIELDS.index("sqrt_alpha_bar")], tab[BETAS_FIELDS.index("sqrt_one_minus_alpha_bar")])
        loss, _ = tr.loss_and_grad(latents, t, noise, ids, betas, T, self.global_bs, self.lc)
        tr.allreduce_grads(loss)                   # data-parallel replicas (one process per GPU): flat RCCL all-reduces; no-op alone
        tr.adam_step()
        self._trainer_dirty = True
        self.loss_tracker.update_state(float(loss.item()))
        return {"loss": self.loss_tracker.result()}

    def encode_latents(self, images):
        """train_step's first half (conditional_dm3d.py:478): latents, _ = quantizer(encoder(images))."""
        return self.quantizer(self.encoder(images))[0]

    # -- a13: sample ------------------------------------------------------------------------------------------------
    def _ddpm_desc(self, x, eps, t_idx, mode, noise=None, seed=0, mean_out=None, var_out=None) -> DdpmDesc:
        tab = self.b.device_tables(self.device)
        d = DdpmDesc()
        d.x, d.eps, d.noise = x.data_ptr(), eps.data_ptr(), (noise.data_ptr() if noise is not None else None)
        d.batch, d.per_sample = x.shape[0], x[0].numel()
        d.t, d.timesteps = t_idx.data_ptr(), self.timesteps
        for i, f in enumerate(BETAS_FIELDS):
            if f != "alpha":
                setattr(d, f, tab[i].data_ptr())
        d.seed, d.mode = int(seed) & (2 ** 64 - 1), mode
        d.mean_out = mean_out.data_ptr() if mean_out is not None else None
        d.var_out = var_out.data_ptr() if var_out is not None else None
        d._keep = (x, eps, noise, t_idx, tab, mean_out, var_out)
        return d

    def sample(self, x_t, pred_noise, curr_time_step, shape):
        """conditional_dm3d.py:517-548: returns (posterior_mean, posterior 'log_variance' [B,1,1,1,1])."""
        x_t = torch.as_tensor(x_t, dtype=torch.float32).to(self.device).contiguous()
        eps = torch.as_tensor(pred_noise, dtype=torch.float32).to(self.device).contiguous()
        B = int(shape[0])
        if x_t.shape[0] != B or eps.shape != x_t.shape:
            raise ValueError("x_t / pred_noise / shape disagree")
        t = torch.as_tensor(curr_time_step).reshape(-1).to(torch.int32)
        if t.numel() != B or int(t.min()) < 0 or int(t.max()) >= self.timesteps:
            raise ValueError("curr_time_step must hold one index in [0, timesteps) per sample")
        t = t.to(self.device)
        mean = torch.empty_like(x_t)
        var = torch.empty(B, dtype=torch.float32, device=self.device)
        d = self._ddpm_desc(x_t, eps, t, 0, mean_out=mean, var_out=var)
        check(lib().dm3d_ddpm_update(C.byref(d), torch.cuda.current_stream().cuda_stream), "ddpm_update")
        return mean, var.reshape(B, 1, 1, 1, 1)

    # -- a14: generate ----------------------------------------------------------------------------------------------
    def _context_ids(self, context_value, batch=None):
        """The reference takes one scalar id and broadcasts it (conditional_dm3d.py:552); an array of shape [B], [B,1] or [B,1,1]
        gives every volume of the batch its own context."""
        if context_value is None:
            # the reference builds tf.constant([[None]]) here and fails (conditional_dm3d.py:552, 586-589)
            raise ValueError("context_value is required for the conditional model")
        ids = np.asarray(context_value.detach().cpu() if torch.is_tensor(context_value) else context_value).astype(np.int64).reshape(-1)
        if ids.size != 1 and batch is not None and ids.size != batch:
            raise ValueError(f"context_value must hold one id or one per volume ({batch}), got {ids.size}")
        if ids.min() < 0 or ids.max() > self.network.cfg.context_dim:
            raise ValueError(f"context ids must lie in [0, {self.network.cfg.context_dim}]")
        return ids.astype(np.int32)

    @staticmethod
    def fresh_seed() -> int:
        """A new 64-bit Philox key from the host's entropy source (the reference draws fresh tf.random.normal noise per call)."""
        import secrets
        return secrets.randbits(64)

    def sampler(self, shape, context_value=None, *, seed=None, use_graph=True) -> "Sampler":
        """The state of one generate() call: plan, tables, context rows and the captured step graph.  There is one live
        Sampler per (batch, context mode): creating another one for the same plan retires the older (its step() raises)."""
        net = self.network
        cfg = net.cfg
        shape = tuple(int(s) for s in shape)
        if len(shape) != 5 or shape[1:] != (cfg.img_size,) * 3 + (cfg.img_channels,):
            raise ValueError(f"shape must be (B,{cfg.img_size},{cfg.img_size},{cfg.img_size},{cfg.img_channels})")
        return Sampler(self, shape, self._context_ids(context_value, shape[0]) if self.conditional else None, seed, use_graph)

    def generate(self, shape=(1, 16, 16, 16, 16), last_step=0, context_value=None, *, x_T=None, noise=None, seed=None,
                 use_graph=True, steps=None):
        """conditional_dm3d.py:550-575.  For shape[0] > 1 the single context row is broadcast to every sample.
        ``seed`` (optional): Philox key of x_T and of every step's noise; None (default) draws a fresh key per call, as the
        reference draws fresh tf.random.normal noise, an integer makes the call reproducible.
        ``noise`` (optional): tensor [timesteps, *shape]; row i is the draw of step i.  ``steps`` (optional) stops
        after that many steps (benchmarks time a prefix of the chain)."""
        if not 0 <= last_step <= self.timesteps:
            raise ValueError("last_step out of range")
        self._sync_from_trainer()
        smp = self.sampler(shape, context_value, seed=seed, use_graph=use_graph and noise is None)
        smp.reset(x_T)
        T = self.timesteps
        n_steps = T - last_step if steps is None else min(int(steps), T - last_step)
        if noise is not None:
            noise = torch.as_tensor(noise, dtype=torch.float32).to(self.device)
            if tuple(noise.shape) != (T,) + smp.shape:
                raise ValueError("noise must be [timesteps, *shape]")
            for k in range(n_steps):
                smp.step(noise=noise[T - 1 - k])
        else:
            for _ in range(n_steps):
                smp.step()
        out = smp.plan.x.clone()
        self.network.check_range(smp.plan)
        return out

    MAX_GRAPHS = 8      # captured step graphs kept per model (one per plan); the least recently used one is destroyed

    def _capture(self, smp: "Sampler"):
        """Capture one step of ``smp`` into a HIP graph, cached per plan: the Philox key lives in a device scalar of the plan
        (dm3d_ddpm_desc.seed_dev), so one graph serves every seed."""
        key = id(smp.plan)
        if key in self._graphs:
            self._graphs[key] = self._graphs.pop(key)                  # most recently used last
            return self._graphs[key][0]
        torch.cuda.synchronize()
        cap = torch.cuda.Stream()
        g = C.c_void_p()
        check(lib().dm3d_graph_begin(cap.cuda_stream), "graph_begin")
        try:
            smp._enqueue(cap.cuda_stream, smp.desc)     # desc reads the key through plan.seed_buf
        finally:
            rc = lib().dm3d_graph_end(cap.cuda_stream, C.byref(g))
        check(rc, "graph_end")
        self._graphs[key] = (g, smp.desc, cap, smp.plan)               # the graph references the descriptor's and plan's memory
        while len(self._graphs) > self.MAX_GRAPHS:
            old = next(iter(self._graphs))
            torch.cuda.synchronize()
            lib().dm3d_graph_destroy(self._graphs.pop(old)[0])
        return g

    def test(self, test_prefix, context=None):
        """conditional_dm3d.py:577-594: generate 10 latents, decode them, np.save the images.  The reference hard-codes the
        latent shape (10,16,16,16,64); here it is (10, latent_size^3, latent_channels), identical for its test setting."""
        import os
        for i in [self.timesteps]:
            print(f"Generating for {i} rsteps")
            if self.vqvae_load_ckpt is not None:
                self.vqvae_trainer.load_weights(self.vqvae_load_ckpt)
            e = self.network.cfg.img_size
            img_latents = self.generate((10, e, e, e, self.lc), last_step=self.timesteps - i, context_value=context)
            images = self.vqvae_trainer.decoder(img_latents)
            os.makedirs("./generated_images_dm3d", exist_ok=True)
            np.save(f"./generated_images_dm3d/{test_prefix}-{i}rsteps.npy", images.cpu().numpy())
        return images


class UnconditionalDiffusionModel(DiffusionModel):
    """networks/dm3d.py:379-545: no context input, self-attention blocks, first_conv_channels = 64."""

    conditional = False

    def _context_ids(self, context_value):
        return None

    def generate(self, shape=(1, 16, 16, 16, 16), last_step=0, **kw):
        kw.pop("context_value", None)
        return super().generate(shape, last_step, None, **kw)

    def test(self, test_prefix):
        return super().test(test_prefix, None)


class Sampler:
    """One DDPM chain over a fixed batch (the loop body of generate, conditional_dm3d.py:559-573).

    ``step()`` enqueues U-Net forward + posterior update + index decrement on the current stream and never synchronises;
    with ``use_graph`` the three are one HIP-graph replay.  A chain has T steps: step() past its end raises until reset().
    The plan (buffers, step index, Philox key) belongs to the newest Sampler made for it; an older one raises on use."""

    def __init__(self, model: DiffusionModel, shape, ctx_ids, seed, use_graph):
        self.model, self.shape, self.use_graph = model, shape, use_graph
        self.seed = (model.fresh_seed() if seed is None else int(seed)) & (2 ** 64 - 1)
        net, T = model.network, model.timesteps
        # one context row per volume, or one broadcast; "sampler": never the plan UNet.__call__ fills with its own time rows
        self.plan = net.plan(shape[0], T, ctx_ids is not None and len(ctx_ids) > 1, purpose="sampler")
        plan = self.plan
        if getattr(plan, "_time_filled", None) is not net.P:
            net.fill_time_table(np.arange(T), plan.vec)
            plan._time_filled = net.P
        if ctx_ids is not None:
            plan.set_context(ctx_ids)
        if getattr(plan, "seed_buf", None) is None:
            plan.seed_buf = torch.zeros(1, dtype=torch.int64, device=model.device)
        plan._owner_gen = getattr(plan, "_owner_gen", 0) + 1
        self._gen = plan._owner_gen
        self.desc = model._ddpm_desc(plan.x, plan.eps, plan.t_idx, 1, seed=self.seed)
        self.desc.seed_dev = plan.seed_buf.data_ptr()
        self._t = -1                          # host mirror of the device step index; -1: no chain in progress

    def _own(self):
        if self._gen != self.plan._owner_gen:
            raise RuntimeError("this Sampler was retired: a newer Sampler (generate() call) took over its plan")

    def reset(self, x_T=None):
        self._own()
        plan, T = self.plan, self.model.timesteps
        st = torch.cuda.current_stream().cuda_stream
        seed_i64 = self.seed - (1 << 64) if self.seed >= (1 << 63) else self.seed
        plan.seed_buf.fill_(seed_i64)
        if x_T is not None:
            plan.x.copy_(torch.as_tensor(x_T, dtype=torch.float32).reshape(self.shape))
        else:
            check(lib().dm3d_randn(plan.x.data_ptr(), plan.x.numel(), self.seed, 0x7fffffff, st), "randn")
        plan.t_idx.fill_(T - 1)
        if plan.range_flag is not None:
            plan.range_flag.zero_()
        self._t = T - 1

    def _enqueue(self, st, desc):
        self.plan.run(st)
        push, pop = _lib.roctx()
        push("ddpm")
        check(lib().dm3d_ddpm_update(C.byref(desc), st), "ddpm_update")
        check(lib().dm3d_add_i32(self.plan.t_idx.data_ptr(), self.plan.B, -1, st), "add_i32")
        pop()

    def prepare(self):
        """Capture the step graph now (setup cost: the first step() otherwise pays for it)."""
        if self.use_graph:
            self.model._capture(self)
        return self

    def step(self, noise=None):
        self._own()
        if self._t < 0:
            raise RuntimeError("the chain is finished (or was never started): call reset() before step()")
        st = torch.cuda.current_stream().cuda_stream
        if noise is not None:
            d = self.model._ddpm_desc(self.plan.x, self.plan.eps, self.plan.t_idx, 1, noise=noise)
            self._enqueue(st, d)
        elif self.use_graph:
            # resolved through the model's cache on every step: load_weights / LRU eviction destroy graphs, never under a live handle
            check(lib().dm3d_graph_launch(self.model._capture(self), st), "graph_launch")
        else:
            self._enqueue(st, self.desc)
        self._t -= 1
        if self._t < 0:
            self.finish()                     # the chain's last step: the one host read of a chain driven through step()

    def finish(self):
        """Reads the H3 range flag of the steps taken so far (one 4-byte device read) and raises if an activation left the
        range the arithmetic covers or turned NaN — what generate() does at its end; step() calls it after a chain's last step, a
        caller that stops a chain early calls it itself."""
        self._own()
        self.model.network.check_range(self.plan)

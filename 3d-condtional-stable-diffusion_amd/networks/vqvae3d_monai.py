"""Drop-in for the inference side of the reference's ``networks/vqvae3d_monai.py`` — the autoencoder that brackets the
diffusion sampler (``DiffusionModel.encoder / quantizer / decoder``, conditional_dm3d.py:425-460, 478, 42).

Same constructor signature as the reference ``VQVAE`` (:394-406) and the same three callables:
``vq.encoder(x) -> latents``, ``vq.quantizer(z) -> (quantized, perplexity)``, ``vq.decoder(z) -> image``, all NDHWC float32
device tensors.  Every layer runs on the dm3d HIP kernels (no fallback):

  Conv3D k4/s2 + ReLU (Encoder :266-285)            -> dm3d_conv3d_ndhwc(ksize 4, stride 2, relu)
  VQVAEResidualUnit (:218-234)                      -> conv k3 + ReLU ; conv k3 with the inference BatchNormalization folded
                                                       into its weights, PReLU slope, + x, ReLU — two launches
  Conv3D k3 + PReLU (:296-301, 346-350)             -> conv k3 with the PReLU epilogue
  Conv3DTranspose k4/s2 [+ ReLU] (:373-381)         -> 8 parity 2x2x2 convs on the input grid (dm3d_pack_weights_convt)
  VectorQuantizer.get_code_indices + lookup (:140-177) -> float32 GEMM z.E, dm3d_vq_assign, dm3d_gather_rows

Keras' ``PReLU()`` has a full-shape slope ``[D,H,W,C]``, which ties the reference model to 128^3 inputs; here the spatial size
is the keyword ``input_size`` (default 128).  Training (``train_step``, losses, codebook replacement) is out of scope.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .. import _lib, ops

BN_EPS = 1e-3


def vqvae_param_spec(in_channels, out_channels, num_channels, num_res_layers, num_res_channels, num_embeddings,
                     embedding_dim, input_size) -> Dict[str, tuple]:
    """Weight inventory in layer-creation order (Encoder, codebook, Decoder), Keras shapes."""
    spec: Dict[str, tuple] = {}

    def res_unit(name, ch, rc, e):
        spec[f"{name}.conv1.kernel"] = (3, 3, 3, ch, rc)
        spec[f"{name}.conv1.bias"] = (rc,)
        spec[f"{name}.conv2.kernel"] = (3, 3, 3, rc, ch)
        spec[f"{name}.conv2.bias"] = (ch,)
        for s in ("gamma", "beta", "mean", "var"):
            spec[f"{name}.bn.{s}"] = (ch,)
        spec[f"{name}.prelu.alpha"] = (e, e, e, ch)

    ch_in, edge = in_channels, input_size
    for i, ch in enumerate(num_channels):
        edge //= 2
        spec[f"enc.down{i}.kernel"] = (4, 4, 4, ch_in, ch)
        spec[f"enc.down{i}.bias"] = (ch,)
        for j in range(num_res_layers):
            res_unit(f"enc.l{i}.res{j}", ch, num_res_channels[i], edge)
        ch_in = ch
    spec["enc.out.kernel"] = (3, 3, 3, ch_in, embedding_dim)
    spec["enc.out.bias"] = (embedding_dim,)
    spec["enc.out_prelu.alpha"] = (edge, edge, edge, embedding_dim)
    spec["vq.embeddings"] = (embedding_dim, num_embeddings)
    rev, rrev = list(reversed(num_channels)), list(reversed(num_res_channels))
    spec["dec.in.kernel"] = (3, 3, 3, embedding_dim, rev[0])
    spec["dec.in.bias"] = (rev[0],)
    spec["dec.in_prelu.alpha"] = (edge, edge, edge, rev[0])
    for i, ch in enumerate(rev):
        for j in range(num_res_layers):
            res_unit(f"dec.l{i}.res{j}", ch, rrev[i], edge)
        out = out_channels if i == len(rev) - 1 else rev[i + 1]
        spec[f"dec.up{i}.kernel"] = (4, 4, 4, out, ch)
        spec[f"dec.up{i}.bias"] = (out,)
        edge *= 2
    return spec


def keras_init_vqvae_weights(spec: Dict[str, tuple], seed: int = 0) -> Dict[str, np.ndarray]:
    """What Keras creates: glorot-uniform kernels, zero biases, BN identity, PReLU slope 0, HeUniform codebook (:125-131)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in spec.items():
        if name.endswith(".kernel"):
            rf = int(np.prod(shape[:3]))
            lim = math.sqrt(6.0 / (shape[3] * rf + shape[4] * rf))
            out[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        elif name.endswith(".embeddings"):
            lim = math.sqrt(6.0 / shape[0])
            out[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        elif name.endswith((".gamma", ".var")):
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


class _Layer:
    __slots__ = ("wpk", "w_exp", "bias", "cout", "alpha")


class VQVAE:
    def __init__(self, in_channels, out_channels, num_channels, num_res_layers, num_res_channels,
                 downsample_parameters=((2, 4, 1, 1), (2, 4, 1, 1), (2, 4, 1, 1)),
                 upsample_parameters=((2, 4, 1, 1, 0), (2, 4, 1, 1, 0), (2, 4, 1, 1, 0)),
                 num_embeddings=128, embedding_dim=64, dropout=0.1, act="relu", output_act=None, num_gpus=2,
                 kernel_resize=False, *, input_size=128, device="cuda", weights=None, seed=0, precision=None):
        import os
        for p in tuple(downsample_parameters)[:len(num_channels)]:
            if tuple(p[:3]) != (2, 4, 1) or p[3] not in ("same", 1):
                raise ValueError("only (stride 2, kernel 4, dilation 1, 'same') downsampling is implemented (the reference's setting)")
        for p in tuple(upsample_parameters)[:len(num_channels)]:
            if tuple(p[:3]) != (2, 4, 1) or p[3] not in ("same", 1):
                raise ValueError("only (stride 2, kernel 4, dilation 1, 'same') upsampling is implemented (the reference's setting)")
        if input_size % (1 << len(num_channels)):
            raise ValueError("input_size must be divisible by 2**levels")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_channels, self.num_res_channels = tuple(num_channels), tuple(num_res_channels)
        self.num_res_layers, self.num_embeddings, self.embedding_dim = num_res_layers, num_embeddings, embedding_dim
        self.output_act, self.num_gpus, self.input_size = output_act, num_gpus, input_size
        self.device = torch.device(device)
        self.precision = precision or os.environ.get("DM3D_PRECISION", "h3")
        if self.precision not in ("fp32", "h3"):
            raise ValueError("precision must be 'fp32' or 'h3'")
        self.spec = vqvae_param_spec(in_channels, out_channels, self.num_channels, num_res_layers, self.num_res_channels,
                                     num_embeddings, embedding_dim, input_size)
        self.state: Dict[str, np.ndarray] = {}
        self._prepared = False
        self.load_state_dict(weights if weights is not None else keras_init_vqvae_weights(self.spec, seed))
        self.encoder, self.quantizer, self.decoder = self._encode, self._quantize, self._decode

    # ---- weights ---------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        new = {}
        for name, shape in self.spec.items():
            if name not in sd:
                if strict:
                    raise ValueError(f"missing weight {name}")
                new[name] = self.state[name]
                continue
            arr = sd[name]
            if isinstance(arr, torch.Tensor):
                arr = arr.detach().cpu().numpy()
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            if tuple(arr.shape) != tuple(shape):
                raise ValueError(f"weight {name}: expected shape {tuple(shape)}, got {tuple(arr.shape)}")
            new[name] = arr
        self.state, self._prepared = new, False

    def load_weights(self, path, root=()):
        """keras ``model.load_weights`` (conditional_dm3d.py:451-454): a TF2 checkpoint prefix written by the reference's
        ``save_weights`` (tf_checkpoint.load_vqvae_state; ``root`` = attribute path of the autoencoder inside the saved object),
        or an .npz of the state dict."""
        if str(path).endswith(".npz"):
            self.load_state_dict(dict(np.load(path)))
            return
        from ..tf_checkpoint import load_vqvae_state
        self.load_state_dict(load_vqvae_state(str(path), self.spec, root=tuple(root)))

    def save_weights(self, path, root=()):
        if str(path).endswith(".npz"):
            np.savez(path, **self.state)
            return
        from ..tf_checkpoint import save_vqvae_checkpoint
        save_vqvae_checkpoint(str(path), self.state, self.spec, root=tuple(root))

    def _dev(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def _layer(self, kernel, bias, kind="conv", alpha=None) -> _Layer:
        L = _Layer()
        h3 = self.precision == "h3"
        k = self._dev(kernel)
        if kind == "convt":
            r = ops.pack_weights_convt(k, h3=h3)
            L.cout = kernel.shape[3]
        else:
            r = ops.pack_weights_h3(k) if h3 else ops.pack_weights(k)
            L.cout = kernel.shape[4]
        L.wpk, L.w_exp = (r if h3 else (r, 0))
        L.bias = self._dev(bias)
        L.alpha = self._dev(alpha) if alpha is not None else None
        return L

    def _res_unit(self, name):
        s = self.state
        c1 = self._layer(s[f"{name}.conv1.kernel"], s[f"{name}.conv1.bias"])
        # inference BatchNormalization after conv2 folds into its weights: W' = W*scale[co], b' = b*scale + shift
        scale = s[f"{name}.bn.gamma"].astype(np.float64) / np.sqrt(s[f"{name}.bn.var"].astype(np.float64) + BN_EPS)
        shift = s[f"{name}.bn.beta"].astype(np.float64) - s[f"{name}.bn.mean"].astype(np.float64) * scale
        w2 = (s[f"{name}.conv2.kernel"].astype(np.float64) * scale).astype(np.float32)
        b2 = (s[f"{name}.conv2.bias"].astype(np.float64) * scale + shift).astype(np.float32)
        c2 = self._layer(w2, b2, alpha=s[f"{name}.prelu.alpha"])
        return c1, c2

    def prepare(self):
        if self._prepared:
            return
        _lib.require_device()
        s, P = self.state, {}
        n = len(self.num_channels)
        for i in range(n):
            P[f"enc.down{i}"] = self._layer(s[f"enc.down{i}.kernel"], s[f"enc.down{i}.bias"])
            for j in range(self.num_res_layers):
                P[f"enc.l{i}.res{j}"] = self._res_unit(f"enc.l{i}.res{j}")
        P["enc.out"] = self._layer(s["enc.out.kernel"], s["enc.out.bias"], alpha=s["enc.out_prelu.alpha"])
        emb = s["vq.embeddings"]
        P["vq.codebook_t"] = self._dev(np.ascontiguousarray(emb.T))                    # [K, D]: rows = codes
        P["vq.esq"] = self._dev((emb.astype(np.float32) ** 2).sum(axis=0, dtype=np.float32))
        P["dec.in"] = self._layer(s["dec.in.kernel"], s["dec.in.bias"], alpha=s["dec.in_prelu.alpha"])
        for i in range(n):
            for j in range(self.num_res_layers):
                P[f"dec.l{i}.res{j}"] = self._res_unit(f"dec.l{i}.res{j}")
            P[f"dec.up{i}"] = self._layer(s[f"dec.up{i}.kernel"], s[f"dec.up{i}.bias"], kind="convt")
        self.P, self._prepared = P, True

    # ---- forward pieces --------------------------------------------------------------------------------------------
    def _conv(self, L: _Layer, x, ksize, **kw):
        prec = _lib.PREC_H3 if self.precision == "h3" else _lib.PREC_F32
        return ops.conv3d(x, L.wpk, L.cout, ksize, bias=L.bias, precision=prec, w_exp=L.w_exp, **kw)

    def _run_res_unit(self, name, x):
        c1, c2 = self.P[name]
        h = self._conv(c1, x, 3, relu=True)
        return self._conv(c2, h, 3, prelu_alpha=c2.alpha, res=x, relu_out=True)

    def _check(self, x, edge, ch, what):
        x = torch.as_tensor(x)
        if x.dim() != 5 or tuple(x.shape[1:]) != (edge, edge, edge, ch) or x.dtype != torch.float32:
            raise ValueError(f"{what} must be float32 [B,{edge},{edge},{edge},{ch}] (NDHWC), got {tuple(x.shape)} {x.dtype}")
        return x.to(self.device).contiguous()

    def _encode(self, x):
        """Encoder.call (:303-305)."""
        self.prepare()
        h = self._check(x, self.input_size, self.in_channels, "encoder input")
        if self.in_channels % 4:                     # the conv kernels read 4 channels per load: zero-pad the input channels
            pad = 4 - self.in_channels % 4
            h = torch.cat([h, torch.zeros(*h.shape[:4], pad, device=self.device)], dim=-1).contiguous()
        for i in range(len(self.num_channels)):
            L = self.P[f"enc.down{i}"]
            if i == 0 and self.in_channels % 4:
                L = self._padded_first_conv()
            h = self._conv(L, h, 4, stride=2, relu=True)
            for j in range(self.num_res_layers):
                h = self._run_res_unit(f"enc.l{i}.res{j}", h)
        L = self.P["enc.out"]
        return self._conv(L, h, 3, prelu_alpha=L.alpha)

    def _padded_first_conv(self):
        if "enc.down0.padded" not in self.P:
            k = self.state["enc.down0.kernel"]
            pad = 4 - self.in_channels % 4
            kp = np.concatenate([k, np.zeros(k.shape[:3] + (pad, k.shape[4]), np.float32)], axis=3)
            self.P["enc.down0.padded"] = self._layer(kp, self.state["enc.down0.bias"])
        return self.P["enc.down0.padded"]

    def get_code_indices(self, flattened_inputs):
        """VectorQuantizer.get_code_indices (:164-177)."""
        self.prepare()
        z = torch.as_tensor(flattened_inputs, dtype=torch.float32).to(self.device).contiguous()
        return ops.vq_assign(z, self.P["vq.codebook_t"], self.P["vq.esq"])

    def _quantize(self, x):
        """VectorQuantizer.call (:133-162), forward values: (quantized, perplexity)."""
        self.prepare()
        e = self.input_size >> len(self.num_channels)
        z = self._check(x, e, self.embedding_dim, "quantizer input")
        flat = z.reshape(-1, self.embedding_dim)
        idx = self.get_code_indices(flat)
        q = ops.gather_rows(self.P["vq.codebook_t"], idx).reshape(z.shape)
        probs = torch.bincount(idx.long(), minlength=self.num_embeddings).float() / idx.numel()
        perplexity = torch.exp(-(probs * torch.log(probs + 1e-10)).sum())
        self.last_indices = idx
        return q, perplexity

    def _decode(self, z):
        """Decoder.call (:388-391)."""
        self.prepare()
        n = len(self.num_channels)
        e = self.input_size >> n
        h = self._check(z, e, self.embedding_dim, "decoder input")
        L = self.P["dec.in"]
        h = self._conv(L, h, 3, prelu_alpha=L.alpha)
        for i in range(n):
            for j in range(self.num_res_layers):
                h = self._run_res_unit(f"dec.l{i}.res{j}", h)
            last = i == n - 1
            h = self._conv(self.P[f"dec.up{i}"], h, 4, stride=2, transpose=True, relu=(not last) or bool(self.output_act))
        return h

    def __call__(self, x):
        """VQVAE.call (:453-457): (decoder(quantizer(encoder(x))), perplexity)."""
        q, perplexity = self._quantize(self._encode(x))
        return self._decode(q), perplexity

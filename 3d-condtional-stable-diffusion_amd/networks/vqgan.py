"""Drop-in for the inference side of the reference's ``networks/vqgan.py`` autoencoder — BASELINE.json config 5 names it:
"vqgan.py encode 128^3 MRI -> 32^3 latent -> conditional DDPM T=1000 -> decode".

Same constructor signature as the reference ``VQGAN`` (vqgan.py:599-620; ``main_exp_vqgan.py:23-38`` builds it with one
(stride 2, kernel 4, "same") level per entry of ``channel_list``, in_channels = out_channels = 2) and the same callables:
``encoder(x)``, ``quantizer(z) -> (quantized, perplexity)``, ``decoder(z)``, ``model(x) -> (reconstruction, perplexity)``
(:699-703), ``call_2`` (:705-709).  The input is ``concat[image, mask]`` (:726-727; ``encode_images`` does it).

What differs from ``networks/vqvae3d_monai.py`` (the autoencoder DiffusionModel wires in) and how it maps to the kernels:
  Encoder level (:317-355)   Conv3D k4/s2 -> BatchNormalization -> PReLU -> n x VQVAEResidualUnit
                             one launch: the inference BatchNormalization folds into the conv's weights and bias, the
                             full-shape PReLU slope rides in the epilogue
  Encoder tail (:357-369)    Conv3D k3 -> PReLU
  Decoder head (:415-426)    Conv3D k3 -> BatchNormalization -> PReLU (folded the same way)
  Decoder level (:429-467)   n x VQVAEResidualUnit -> Conv3DTranspose k4/s2 -> BatchNormalization [-> PReLU unless last]
                             8 parity 2x2x2 convs on the input grid with the norm folded into the transposed kernel
  VQVAEResidualUnit (:257-284), VectorQuantizer (:151-216): as in vqvae3d_monai.py
Dropout layers are the identity at inference.  The GAN half (discriminators, LPIPS, train_step) is out of scope.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from .vqvae3d_monai import BN_EPS, VQVAE, _Layer, keras_init_vqvae_weights, vqvae_param_spec


def vqgan_param_spec(in_channels, out_channels, num_channels, num_res_layers, num_res_channels, num_embeddings, embedding_dim,
                     input_size) -> Dict[str, tuple]:
    """The VQ-VAE inventory plus the BatchNormalization / PReLU layers vqgan.py adds after every strided conv, after the
    decoder's first conv and after every Conv3DTranspose (layer-creation order within each module is kept)."""
    base = vqvae_param_spec(in_channels, out_channels, num_channels, num_res_layers, num_res_channels, num_embeddings, embedding_dim,
                            input_size)
    n = len(num_channels)
    rev = list(reversed(num_channels))
    lat = input_size >> n
    spec: Dict[str, tuple] = {}
    for name, shape in base.items():
        spec[name] = shape
        for i in range(n):
            if name == f"enc.down{i}.bias":
                ch, e = num_channels[i], input_size >> (i + 1)
                for s in ("gamma", "beta", "mean", "var"):
                    spec[f"enc.down{i}.bn.{s}"] = (ch,)
                spec[f"enc.down{i}.prelu.alpha"] = (e, e, e, ch)
            if name == f"dec.up{i}.bias":
                out = out_channels if i == n - 1 else rev[i + 1]
                e = lat << (i + 1)
                for s in ("gamma", "beta", "mean", "var"):
                    spec[f"dec.up{i}.bn.{s}"] = (out,)
                if i != n - 1:
                    spec[f"dec.up{i}.prelu.alpha"] = (e, e, e, out)
        if name == "dec.in.bias":
            for s in ("gamma", "beta", "mean", "var"):
                spec[f"dec.in.bn.{s}"] = (rev[0],)
    return spec


class VQGAN(VQVAE):
    def __init__(self, in_channels, out_channels, num_channels, num_res_layers, num_res_channels,
                 downsample_parameters=((2, 4, 1, 1), (2, 4, 1, 1), (2, 4, 1, 1)),
                 upsample_parameters=((2, 4, 1, 1, 0), (2, 4, 1, 1, 0), (2, 4, 1, 1, 0)),
                 num_embeddings=128, embedding_dim=64, dropout=0.5, output_act=None, num_gpus=2, kernel_resize=False, B=12, D=128,
                 disc_threshold=0, disc_loss_fn="vanilla", act_fn="prelu", disc_use_sigmoid=False, lpips_wt=4, gan_feat_wt=0.8,
                 *, device="cuda", weights=None, seed=0, precision=None):
        if act_fn != "prelu":
            raise ValueError("the reference builds its Encoder / Decoder with act_fn='prelu' whatever VQGAN is given (vqgan.py:657, 670)")
        self.B, self.D = B, D
        self._vqgan_weights = weights
        super().__init__(in_channels, out_channels, num_channels, num_res_layers, num_res_channels, downsample_parameters,
                         upsample_parameters, num_embeddings, embedding_dim, dropout, "relu", output_act, num_gpus, kernel_resize,
                         input_size=D, device=device, weights=None, seed=seed, precision=precision)
        self.spec = vqgan_param_spec(in_channels, out_channels, self.num_channels, num_res_layers, self.num_res_channels, num_embeddings,
                                     embedding_dim, D)
        self.state = {}
        self.load_state_dict(weights if weights is not None else keras_init_vqvae_weights(self.spec, seed))

    # ---- weights -----------------------------------------------------------------------------------------------------
    def load_weights(self, path, root=()):
        """keras ``VQGAN.load_weights(prefix)`` (main_exp_vqgan.py:23-38 builds the model the checkpoint belongs to): the encoder, decoder
        and quantizer of a TF2 checkpoint written by the reference's ``save_weights`` — the discriminators, LPIPS network and optimizer
        slots it also holds are not read —, or an .npz of the state dict."""
        if str(path).endswith(".npz"):
            self.load_state_dict(dict(np.load(path)))
            return
        from ..tf_checkpoint import load_vqvae_state
        self.load_state_dict(load_vqvae_state(str(path), self.spec, root=tuple(root), parts=("encoder", "decoder", "quantizer")))

    def save_weights(self, path, root=()):
        if str(path).endswith(".npz"):
            np.savez(path, **self.state)
            return
        from ..tf_checkpoint import save_vqvae_checkpoint
        save_vqvae_checkpoint(str(path), self.state, self.spec, root=tuple(root))

    def _folded(self, conv: str, bn: str, transpose: bool = False):
        """Conv kernel / bias with the inference BatchNormalization that follows folded in: W' = W*scale[co], b' = b*scale + shift."""
        s = self.state
        scale = s[f"{bn}.gamma"].astype(np.float64) / np.sqrt(s[f"{bn}.var"].astype(np.float64) + BN_EPS)
        shift = s[f"{bn}.beta"].astype(np.float64) - s[f"{bn}.mean"].astype(np.float64) * scale
        k = s[f"{conv}.kernel"].astype(np.float64)
        k = k * (scale[:, None] if transpose else scale)          # Conv3DTranspose kernels are [kd,kh,kw,Cout,Cin]
        b = s[f"{conv}.bias"].astype(np.float64) * scale + shift
        return k.astype(np.float32), b.astype(np.float32)

    def prepare(self):
        if self._prepared:
            return
        super().prepare()                   # residual units, encoder tail, codebook; the layers below are re-packed with their norms
        s, P, n = self.state, self.P, len(self.num_channels)
        for i in range(n):
            k, b = self._folded(f"enc.down{i}", f"enc.down{i}.bn")
            if i == 0 and self.in_channels % 4:                   # the conv kernels read 4 channels per load: zero input channels
                pad = 4 - self.in_channels % 4
                k = np.concatenate([k, np.zeros(k.shape[:3] + (pad, k.shape[4]), np.float32)], axis=3)
            P[f"enc.down{i}"] = self._layer(k, b, alpha=s[f"enc.down{i}.prelu.alpha"])
            k, b = self._folded(f"dec.up{i}", f"dec.up{i}.bn", transpose=True)
            P[f"dec.up{i}"] = self._layer(k, b, kind="convt", alpha=s.get(f"dec.up{i}.prelu.alpha"))
        k, b = self._folded("dec.in", "dec.in.bn")
        P["dec.in"] = self._layer(k, b, alpha=s["dec.in_prelu.alpha"])

    # ---- forward -------------------------------------------------------------------------------------------------------
    def _encode(self, x):
        """Encoder.call (vqgan.py:373-375)."""
        self.prepare()
        h = self._check(x, self.input_size, self.in_channels, "encoder input")
        if self.in_channels % 4:
            pad = 4 - self.in_channels % 4
            h = torch.cat([h, torch.zeros(*h.shape[:4], pad, device=self.device)], dim=-1).contiguous()
        for i in range(len(self.num_channels)):
            L = self.P[f"enc.down{i}"]
            h = self._conv(L, h, 4, stride=2, prelu_alpha=L.alpha)
            for j in range(self.num_res_layers):
                h = self._run_res_unit(f"enc.l{i}.res{j}", h)
        L = self.P["enc.out"]
        return self._conv(L, h, 3, prelu_alpha=L.alpha)

    def _decode(self, z):
        """Decoder.call (vqgan.py:472-475)."""
        self.prepare()
        n = len(self.num_channels)
        h = self._check(z, self.input_size >> n, self.embedding_dim, "decoder input")
        L = self.P["dec.in"]
        h = self._conv(L, h, 3, prelu_alpha=L.alpha)
        for i in range(n):
            for j in range(self.num_res_layers):
                h = self._run_res_unit(f"dec.l{i}.res{j}", h)
            L = self.P[f"dec.up{i}"]
            h = self._conv(L, h, 4, stride=2, transpose=True, prelu_alpha=L.alpha, relu_out=(i == n - 1 and bool(self.output_act)))
        return h

    def encode_images(self, img, mask):
        """``x = tf.concat([img, mask], axis=-1)`` (train_step, vqgan.py:726-727) -> encoder."""
        img, mask = torch.as_tensor(img, dtype=torch.float32).to(self.device), torch.as_tensor(mask, dtype=torch.float32).to(self.device)
        return self._encode(torch.cat([img, mask], dim=-1).contiguous())

    def call_2(self, x):
        """vqgan.py:705-709: the quantised latents."""
        return self._quantize(self._encode(x))[0]

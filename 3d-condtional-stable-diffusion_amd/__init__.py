"""dm3d_amd — MI355X-native 3D latent-diffusion denoising path (drop-in for the reference's
networks/conditional_dm3d.py / networks/dm3d.py ``build_model`` + ``DiffusionModel``).

The directory is named after the reference repo; import it as ``dm3d_amd`` (see ``dm3d_amd.py`` at the repo root).
"""
from .betas import Betas, time_embedding_table
from .weights import UNetConfig, kernel_init, keras_init_weights, param_spec, synthetic_weights

__all__ = ["Betas", "time_embedding_table", "UNetConfig", "kernel_init", "keras_init_weights", "param_spec",
           "synthetic_weights", "UNet", "DiffusionModel", "UnconditionalDiffusionModel"]


def __getattr__(name):          # torch / the HIP library are only needed by the device classes
    if name == "UNet":
        from .unet import UNet
        return UNet
    if name in ("DiffusionModel", "UnconditionalDiffusionModel"):
        from . import diffusion
        return getattr(diffusion, name)
    raise AttributeError(name)

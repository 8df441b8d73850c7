"""MI355X (gfx950) native DDPM denoising hot path of AdrianoEttari/DiffusionRemoteSensing.

Public surface mirrors the reference's two hot-path files:
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres, EMA
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion, launch
The C-ABI underneath is declared in include/drs_hip.h.
"""
from .UNet_model_superres import EMA, Residual_Attention_UNet_superres  # noqa: F401
from .train_diffusion_superres import Diffusion, launch  # noqa: F401

__all__ = ["EMA", "Residual_Attention_UNet_superres", "Diffusion", "launch"]

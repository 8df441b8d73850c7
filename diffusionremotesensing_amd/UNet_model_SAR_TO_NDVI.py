"""Drop-in `Residual_Attention_UNet_SAR_TO_NDVI` (reference UNet_model_SAR_TO_NDVI.py:263-370): the same trunk as the
super-resolution UNet, conditioned on a SAR image at the output resolution (no bicubic up-sampling), state_dict keys
`SAR_encoder.*` / `conv_SAR_img.*`.  Parameter holders only; the forward runs on the HIP plan (variant
DRS_VARIANT_SAR_TO_NDVI of include/drs_hip.h)."""
import torch.nn as nn

from .UNet_model_superres import (EMA, RRDB, AttentionBlock, ResConvBlock as _ResConvBlock, ResidualBlock,  # noqa: F401
                                  UpConvBlock, _HipUNet, gating_signal)


class ResConvBlock(_ResConvBlock):
    """reference UNet_model_SAR_TO_NDVI.py:108-169 (x_skip convolution registered as `conv_SAR_img`, :126)."""
    SKIP_NAME = "conv_SAR_img"


class Residual_Attention_UNet_SAR_TO_NDVI(_HipUNet):
    VARIANT = "sar_to_ndvi"
    RES_BLOCK = ResConvBlock

    def __init__(self, SAR_channels=2, NDVI_channels=1, device=None):
        super().__init__()
        self.SAR_channels = SAR_channels
        self.NDVI_channels = NDVI_channels
        self.time_emb_dim = 100
        self.device = device
        self.conv0 = nn.Conv2d(NDVI_channels, 16, 3, padding=1)
        self.SAR_encoder = RRDB(in_channels=SAR_channels, out_channels=SAR_channels, num_blocks=3)
        self.conv_SAR_img = nn.Conv2d(SAR_channels, 16, 3, padding=1)
        self._build_trunk(NDVI_channels, device)
        self._hip_engine = None

    def forward(self, NDVI_img, timestep, SAR_img):
        return self.hip_engine().forward(NDVI_img, timestep, SAR_img, 1)

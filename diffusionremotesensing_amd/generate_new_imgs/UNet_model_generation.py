"""Drop-in `Residual_Attention_UNet_generation` (reference generate_new_imgs/UNet_model_generation.py:226-329): the
trunk of the super-resolution UNet without a conditioning image; an optional class label is embedded and added to the
time encoding (`t += label_emb(y)`, :300-301).  Parameter holders only; the forward runs on the HIP plan (variant
DRS_VARIANT_GENERATION of include/drs_hip.h)."""
import torch.nn as nn

from ..UNet_model_superres import (EMA, AttentionBlock, ResConvBlock as _ResConvBlock, UpConvBlock,  # noqa: F401
                                   _HipUNet, gating_signal)


class ResConvBlock(_ResConvBlock):
    """reference UNet_model_generation.py:104-165 (x_skip convolution registered as `conv_skip`, :122)."""
    SKIP_NAME = "conv_skip"


class Residual_Attention_UNet_generation(_HipUNet):
    VARIANT = "generation"
    RES_BLOCK = ResConvBlock

    def __init__(self, image_channels=3, out_dim=3, num_classes=None, device=None):
        super().__init__()
        self.image_channels = image_channels
        self.out_dim = out_dim
        self.time_emb_dim = 100
        self.device = device
        self.num_classes = num_classes
        self.conv0 = nn.Conv2d(image_channels, 16, 3, padding=1)
        self._build_trunk(out_dim, device)
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, self.time_emb_dim).to(device=device)
            # an unconditional step leaves this parameter without a gradient on THIS rank only: the multi-GPU gradient
            # exchange and FusedAdam resolve it after the all-reduce (dist.allreduce_gradients)
            self.label_emb.weight._drs_maybe_unused = True
        self._hip_engine = None

    def forward(self, x, timestep, y=None):
        return self.hip_engine().forward(x, timestep, None, 1, labels=y)

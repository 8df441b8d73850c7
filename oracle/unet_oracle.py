"""TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.

CPU restatement (stock `torch.nn.functional`, fp32) of the reference UNet forward, written from the
reference's source as a pure function of a state_dict.  Only tests/, `__graft_entry__.smoke()` and
`bench.py`'s cpu_baseline leg may import this file.

Parity pin: the reference's own tests pin nothing for this path (it has none); the
arithmetic lives in PyTorch ATen.  This restatement is pinned by golden vectors generated in the build
container by importing the reference itself (`tools/make_golden.py` -> tests/golden/*.npz), against which
`tests/test_oracle_golden.py` checks it; the two agree bit-for-bit there because they issue the same
ATen ops in the same order.

Every function cites the reference lines it follows (paths relative to the reference checkout).
"""
import torch
import torch.nn.functional as F

DOWN = (16, 32, 64, 128, 256)
UP = (256, 128, 64, 32, 16)
TIME_EMB_DIM = 100
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def pos_encoding(t, channels):
    """UNet_model_superres.py:328-335.  t: (B,1) float32."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels))
    a = torch.sin(t.repeat(1, channels // 2) * inv_freq)
    b = torch.cos(t.repeat(1, channels // 2) * inv_freq)
    return torch.cat([a, b], dim=-1)


def _bn(sd, pfx, x, training, stats_out):
    """nn.BatchNorm2d (eps 1e-5, momentum 0.1, affine, tracked stats).  In training mode the batch statistics
    are used and the would-be running-stat update is recorded in `stats_out` (the state_dict is not mutated)."""
    rm, rv = sd[pfx + ".running_mean"], sd[pfx + ".running_var"]
    if training:
        rm, rv = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm, rv, sd[pfx + ".weight"], sd[pfx + ".bias"], True, BN_MOMENTUM, BN_EPS)
        if stats_out is not None:
            stats_out[pfx] = (rm, rv)
        return y
    return F.batch_norm(x, rm, rv, sd[pfx + ".weight"], sd[pfx + ".bias"], False, BN_MOMENTUM, BN_EPS)


def _conv(sd, pfx, x, stride=1, padding=0):
    return F.conv2d(x, sd[pfx + ".weight"], sd[pfx + ".bias"], stride=stride, padding=padding)


def _time_mlp(sd, pfx, t):
    """_make_te: Linear -> SiLU -> Linear (:143-151), then the block's ReLU (:161 / :199)."""
    h = F.linear(t, sd[pfx + ".0.weight"], sd[pfx + ".0.bias"])
    h = F.silu(h)
    h = F.linear(h, sd[pfx + ".2.weight"], sd[pfx + ".2.bias"])
    return F.relu(h)[(...,) + (None,) * 2]


def res_conv_block(sd, pfx, x, t, x_skip, training=False, stats=None, taps=None, skip_name="conv_upsampled_lr_img"):
    """ResConvBlock.forward, :153-172 (same body in UNet_model_SAR_TO_NDVI.py:150-169 with `conv_SAR_img` and in
    generate_new_imgs/UNet_model_generation.py:146-165 with `conv_skip`)."""
    h = F.relu(_bn(sd, pfx + ".batch_norm1", _conv(sd, pfx + ".conv1.0", x, padding=1), training, stats))
    if x_skip is not None:
        h = h + _conv(sd, pfx + "." + skip_name, x_skip, padding=1)
    h = h + _time_mlp(sd, pfx + ".time_mlp", t)
    if taps is not None:
        taps[pfx + ".h"] = h
    h = _bn(sd, pfx + ".batch_norm2", _conv(sd, pfx + ".conv2.0", h, padding=1), training, stats)
    shortcut = _bn(sd, pfx + ".shortcut_batch_norm", _conv(sd, pfx + ".shortcut_conv.0", x), training, stats)
    if taps is not None:
        taps[pfx + ".shortcut"] = shortcut
    return F.relu(shortcut + h)


def gating_signal(sd, pfx, x, training=False, stats=None):
    """gating_signal.forward, :222-225."""
    return F.relu(_bn(sd, pfx + ".batch_norm", _conv(sd, pfx + ".conv", x), training, stats))


def attention_block(sd, pfx, x, g, training=False, stats=None, taps=None):
    """AttentionBlock.forward, :89-108."""
    g1 = _conv(sd, pfx + ".w_g.0", g)
    x1 = _conv(sd, pfx + ".w_x.0", x, stride=2)
    psi = F.relu(g1 + x1)
    psi = torch.sigmoid(_conv(sd, pfx + ".psi.0", psi))
    if taps is not None:
        taps[pfx + ".psi"] = psi
    up = F.interpolate(psi, scale_factor=2, mode="nearest")
    up = up.repeat_interleave(repeats=x.shape[1], dim=1)
    return _bn(sd, pfx + ".result.1", _conv(sd, pfx + ".result.0", up * x), training, stats)


def up_conv_block(sd, pfx, x, t, training=False, stats=None, taps=None):
    """UpConvBlock.forward, :197-207."""
    x = x + _time_mlp(sd, pfx + ".time_mlp", t)
    x = F.relu(_bn(sd, pfx + ".batch_norm", _conv(sd, pfx + ".conv", x, padding=1), training, stats))
    if taps is not None:
        taps[pfx + ".conv"] = x
    return F.conv_transpose2d(x, sd[pfx + ".transform.weight"], sd[pfx + ".transform.bias"], stride=2, padding=1,
                              output_padding=1)


def rrdb(sd, pfx, x):
    """RRDB.forward / ResidualBlock.forward, :237-260."""
    out = x
    for i in range(3):
        r = out
        o = F.relu(_conv(sd, f"{pfx}.blocks.{i}.conv1", out, padding=1))
        o = _conv(sd, f"{pfx}.blocks.{i}.conv2", o, padding=1)
        out = o + r
    out = _conv(sd, pfx + ".conv_out", out, padding=1)
    return out + x


def _trunk(sd, x, t, skip_name, training, stats, taps):
    """Encoder, bottleneck, gated decoder and output projection: identical in the three reference models
    (UNet_model_superres.py:356-379, UNet_model_SAR_TO_NDVI.py:347-370, UNet_model_generation.py:304-329)."""
    if taps is not None:
        taps["x0"] = x
    x_skip = x.clone()
    residual_inputs = []
    for i in range(3):
        pfx = f"conv_blocks.{i}"
        x = res_conv_block(sd, pfx, x, t, x_skip if i == 0 else None, training, stats, taps, skip_name)
        if taps is not None:
            taps[pfx] = x
        residual_inputs.append(x)
        x = _conv(sd, f"downs.{i}", x, stride=2, padding=1)
        if taps is not None:
            taps[f"downs.{i}"] = x
    x = res_conv_block(sd, "bottle_neck", x, t, None, training, stats, taps, skip_name)
    if taps is not None:
        taps["bottle_neck"] = x
    for i in range(3):
        g = gating_signal(sd, f"gating_signals.{i}", x, training, stats)
        att = attention_block(sd, f"attention_blocks.{i}", residual_inputs[-(i + 1)], g, training, stats, taps)
        x = up_conv_block(sd, f"ups.{i}", x, t, training, stats, taps)
        if taps is not None:
            taps[f"gating_signals.{i}"] = g
            taps[f"attention_blocks.{i}"] = att
            taps[f"ups.{i}"] = x
        x = torch.cat([x, att], dim=1)
        x = _conv(sd, f"up_convs.{i}", x, padding=1)
        if taps is not None:
            taps[f"up_convs.{i}"] = x
    return _conv(sd, "output", x)


def unet_forward(sd, x, timestep, lr_img, magnification_factor, training=False, stats=None, taps=None):
    """Residual_Attention_UNet_superres.forward, :337-379.  `taps`, when a dict, receives the named
    intermediate activations (names match drs_unet_tensor_name)."""
    t = pos_encoding(timestep.unsqueeze(-1).type(torch.float), TIME_EMB_DIM)
    x = _conv(sd, "conv0", x, padding=1)
    lr = rrdb(sd, "LR_encoder", lr_img)
    up = F.interpolate(lr, scale_factor=magnification_factor, mode="bicubic")
    if taps is not None:
        taps["LR_encoder"] = lr
        taps["upsampled_lr_img"] = up
    x = x + _conv(sd, "conv_upsampled_lr_img", up, padding=1)
    return _trunk(sd, x, t, "conv_upsampled_lr_img", training, stats, taps)


def unet_forward_sar(sd, ndvi_img, timestep, sar_img, training=False, stats=None, taps=None):
    """Residual_Attention_UNet_SAR_TO_NDVI.forward, UNet_model_SAR_TO_NDVI.py:332-370."""
    t = pos_encoding(timestep.unsqueeze(-1).type(torch.float), TIME_EMB_DIM)
    x = _conv(sd, "conv0", ndvi_img, padding=1)
    sar = rrdb(sd, "SAR_encoder", sar_img)
    if taps is not None:
        taps["SAR_encoder"] = sar
    x = x + _conv(sd, "conv_SAR_img", sar, padding=1)
    return _trunk(sd, x, t, "conv_SAR_img", training, stats, taps)


def unet_forward_generation(sd, x, timestep, y=None, training=False, stats=None, taps=None):
    """Residual_Attention_UNet_generation.forward, generate_new_imgs/UNet_model_generation.py:296-329."""
    t = pos_encoding(timestep.unsqueeze(-1).type(torch.float), TIME_EMB_DIM)
    if y is not None:
        t = t + F.embedding(y, sd["label_emb.weight"])  # `t += self.label_emb(y)`, :300-301
    x = _conv(sd, "conv0", x, padding=1)
    return _trunk(sd, x, t, "conv_skip", training, stats, taps)


class OracleUNet(torch.nn.Module):
    """Callable wrapper with the reference's `model(x, t, lr_img, mag)` contract, for driving
    `oracle.diffusion_oracle.sample` and the gloo tests."""

    def __init__(self, sd):
        super().__init__()
        self.sd = sd

    def forward(self, x, timestep, lr_img, magnification_factor):
        with torch.no_grad():
            return unet_forward(self.sd, x, timestep, lr_img, magnification_factor, training=False)


class OracleUNetSAR(torch.nn.Module):
    """`model(NDVI_img, t, SAR_img)` contract of the SAR->NDVI model."""

    def __init__(self, sd):
        super().__init__()
        self.sd = sd

    def forward(self, ndvi_img, timestep, sar_img):
        with torch.no_grad():
            return unet_forward_sar(self.sd, ndvi_img, timestep, sar_img)


class OracleUNetGeneration(torch.nn.Module):
    """`model(x, t, y=None)` contract of the class-conditional generation model."""

    def __init__(self, sd):
        super().__init__()
        self.sd = sd

    def forward(self, x, timestep, y=None):
        with torch.no_grad():
            return unet_forward_generation(self.sd, x, timestep, y)

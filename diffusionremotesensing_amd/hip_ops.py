"""Thin torch-tensor wrappers over the operator-level C-ABI entries (include/drs_hip.h).

PyTorch is used only for device memory and the current stream.  Every wrapper requires
ROCm-device fp32 contiguous tensors and raises otherwise: there is no CPU path.
"""
import ctypes as C

import torch

from . import _lib


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _req(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the HIP path needs a ROCm device tensor (no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def conv2d(x, w, b=None, stride=1, padding=0, transposed=False, output_padding=0, relu=False, impl="direct"):
    """F.conv2d / F.conv_transpose2d for the flavours the UNet uses (NCHW, fp32)."""
    lib = _lib.load()
    x = _req(x, "x"); w = _req(w, "w")
    if b is not None:
        b = _req(b, "b")
    N, Cin, H, W = x.shape
    if transposed:
        if w.shape[0] != Cin:
            raise RuntimeError(f"conv_transpose2d: weight {tuple(w.shape)} does not match Cin={Cin}")
        Cout = w.shape[1]
    else:
        if w.shape[1] != Cin:
            raise RuntimeError(f"conv2d: weight {tuple(w.shape)} does not match Cin={Cin}")
        Cout = w.shape[0]
    KH, KW = w.shape[2], w.shape[3]
    if transposed:
        OH = (H - 1) * stride - 2 * padding + KH + output_padding
        OW = (W - 1) * stride - 2 * padding + KW + output_padding
    else:
        OH = (H + 2 * padding - KH) // stride + 1
        OW = (W + 2 * padding - KW) // stride + 1
    y = torch.empty((N, Cout, max(OH, 0), max(OW, 0)), dtype=torch.float32, device=x.device)
    args = (N, Cin, H, W, Cout, KH, KW, stride, padding, int(transposed), output_padding)
    nbytes = lib.drs_conv2d_workspace_bytes(*args)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    impl_id = _lib.IMPL_BY_NAME[impl] if isinstance(impl, str) else int(impl)
    with torch.cuda.device(x.device):
        st = lib.drs_conv2d_nchw(_ptr(x), _ptr(w), _ptr(b), _ptr(y), *args, int(relu), _ptr(ws), nbytes, impl_id,
                                 _stream(x.device))
    _lib.check(st, "drs_conv2d_nchw")
    return y


def upconv_fused(h, att, t_w, t_b, v_w, v_b, post2=None, fuse_w=None, fuse_b=None):
    """conv2d(cat([conv_transpose2d(h, t_w, t_b, 2, 1, 1), att], 1), v_w, v_b, padding=1) the way the eval plan runs a
    decoder stage: transposed convolution and the x-half of the 3x3 convolution composed into one stride-2 transposed
    convolution (include/drs_hip.h: drs_upconv_fused_nchw).  Returns y, or (y, y + post2[:, :, None, None]) with
    `post2` (N, Ch), or conv1x1(y, fuse_w, fuse_b) with `fuse_w` (fuse_dim, Ch)."""
    lib = _lib.load()
    h = _req(h, "h"); att = _req(att, "att"); t_w = _req(t_w, "t_w"); t_b = _req(t_b, "t_b")
    v_w = _req(v_w, "v_w"); v_b = _req(v_b, "v_b")
    N, Cc, LH, LW = h.shape
    Ch = v_w.shape[0]
    if att.shape != (N, Ch, 2 * LH, 2 * LW) or v_w.shape[1] != Cc + Ch or tuple(t_w.shape) != (Cc, Cc, 3, 3):
        raise RuntimeError(f"upconv_fused: shapes h {tuple(h.shape)} att {tuple(att.shape)} t_w {tuple(t_w.shape)} "
                           f"v_w {tuple(v_w.shape)} do not fit one decoder stage")
    fd = 0
    if fuse_w is not None:
        fuse_w = _req(fuse_w.reshape(fuse_w.shape[0], -1), "fuse_w"); fuse_b = _req(fuse_b, "fuse_b")
        fd = fuse_w.shape[0]
    if post2 is not None:
        post2 = _req(post2, "post2")
    y = torch.empty((N, fd if fd else Ch, 2 * LH, 2 * LW), dtype=torch.float32, device=h.device)
    y2 = torch.empty_like(y) if post2 is not None else None
    nbytes = lib.drs_upconv_fused_workspace_bytes(N, Cc, Ch, LH, LW)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
    with torch.cuda.device(h.device):
        st = lib.drs_upconv_fused_nchw(_ptr(h), _ptr(att), _ptr(t_w), _ptr(t_b), _ptr(v_w), _ptr(v_b), _ptr(post2),
                                       _ptr(fuse_w), _ptr(fuse_b), fd, _ptr(y), _ptr(y2), N, Cc, Ch, LH, LW, _ptr(ws),
                                       nbytes, _stream(h.device))
    _lib.check(st, "drs_upconv_fused_nchw")
    return (y, y2) if y2 is not None else y


def bicubic_upsample(x, scale):
    lib = _lib.load()
    x = _req(x, "x")
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, H * scale, W * scale), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.drs_bicubic_upsample_nchw(_ptr(x), _ptr(y), N, Cc, H, W, int(scale), _stream(x.device))
    _lib.check(st, "drs_bicubic_upsample_nchw")
    return y


def inv_freq_table(channels=100):
    """Same expression as reference pos_encoding (UNet_model_superres.py:329-331), on the host."""
    return (1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels))).contiguous()


def time_mlp(t, W1, b1, W2, b2):
    lib = _lib.load()
    t = _req(t, "t", torch.int64)
    W1, b1, W2, b2 = (_req(a, n) for a, n in ((W1, "W1"), (b1, "b1"), (W2, "W2"), (b2, "b2")))
    dim_out, dim_in = W1.shape
    inv = inv_freq_table(dim_in).to(t.device)
    out = torch.empty((t.shape[0], dim_out), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        st = lib.drs_time_mlp(_ptr(t), _ptr(inv), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(out), t.shape[0], dim_in,
                              dim_out, _stream(t.device))
    _lib.check(st, "drs_time_mlp")
    return out


def noise_images(x0, eps, t, alpha_hat):
    lib = _lib.load()
    x0 = _req(x0, "x0"); eps = _req(eps, "eps"); t = _req(t, "t", torch.int64); alpha_hat = _req(alpha_hat, "alpha_hat")
    if eps.shape != x0.shape or t.shape[0] != x0.shape[0]:
        raise RuntimeError("noise_images: shape mismatch")
    out = torch.empty_like(x0)
    n = x0.shape[0]
    chw = x0.numel() // n if n else 0
    with torch.cuda.device(x0.device):
        st = lib.drs_noise_images(_ptr(x0), _ptr(eps), _ptr(t), _ptr(alpha_hat), alpha_hat.numel(), _ptr(out), n, chw,
                                  _stream(x0.device))
    _lib.check(st, "drs_noise_images")
    return out


def timestep_table(noise_steps, n, device):
    """(noise_steps, n) int64: row i is the reference's `t = (torch.ones(n) * i).long()` of sampling step i
    (train_diffusion_superres.py:237) for every step of a chain, built once: a row view per step instead of a fill kernel
    per step (one launch of ~4 us in a 1.35 ms step)."""
    return torch.arange(noise_steps, dtype=torch.int64, device=device).unsqueeze(1).expand(noise_steps, n).contiguous()


def sampler_step_(x, eps_pred, noise, t, alpha, alpha_hat, beta):
    """In-place ancestral update of x for the scalar timestep t (noise may be None)."""
    lib = _lib.load()
    if not (x.is_cuda and x.is_contiguous() and x.dtype == torch.float32):
        raise RuntimeError("sampler_step_: x must be a contiguous fp32 ROCm tensor (no CPU fallback)")
    eps_pred = _req(eps_pred, "eps_pred")
    noise = _req(noise, "noise") if noise is not None else None
    with torch.cuda.device(x.device):
        st = lib.drs_sampler_step(_ptr(x), _ptr(eps_pred), _ptr(noise), int(t), _ptr(alpha), _ptr(alpha_hat),
                                  _ptr(beta), alpha.numel(), x.numel(), _stream(x.device))
    _lib.check(st, "drs_sampler_step")
    return x


def sampler_step_cfg_(x, eps_cond, eps_uncond, cfg_scale, noise, t, alpha, alpha_hat, beta):
    """In-place ancestral update with eps = torch.lerp(eps_uncond, eps_cond, cfg_scale) folded in
    (reference generate_new_imgs/train_diffusion_generation.py:236-249)."""
    lib = _lib.load()
    if not (x.is_cuda and x.is_contiguous() and x.dtype == torch.float32):
        raise RuntimeError("sampler_step_cfg_: x must be a contiguous fp32 ROCm tensor (no CPU fallback)")
    eps_cond = _req(eps_cond, "eps_cond")
    eps_uncond = _req(eps_uncond, "eps_uncond")
    noise = _req(noise, "noise") if noise is not None else None
    with torch.cuda.device(x.device):
        st = lib.drs_sampler_step_cfg(_ptr(x), _ptr(eps_cond), _ptr(eps_uncond), float(cfg_scale), _ptr(noise), int(t),
                                      _ptr(alpha), _ptr(alpha_hat), _ptr(beta), alpha.numel(), x.numel(),
                                      _stream(x.device))
    _lib.check(st, "drs_sampler_step_cfg")
    return x


def aggregate_tiles(tiles, origins, weight, height, width):
    """Gaussian-weighted blend of (n,C,S,S) tiles placed at `origins` [(y0, x0), ...] into a (C,height,width) image,
    normalised by the summed weights and clamped to [0,1] (reference Aggregation_Sampling.py:90-116).
    Raises like the reference's `assert torch.all(pixel_count != 0)` when a pixel is covered by no tile."""
    lib = _lib.load()
    tiles = _req(tiles, "tiles")
    weight = _req(weight, "weight")
    n, C, S, S2 = tiles.shape
    if S != S2 or tuple(weight.shape) != (S, S) or len(origins) != n:
        raise RuntimeError(f"aggregate_tiles: tiles {tuple(tiles.shape)}, weight {tuple(weight.shape)}, {len(origins)} origins")
    org = torch.tensor([[int(y), int(x)] for y, x in origins], dtype=torch.int32).to(tiles.device)
    out = torch.empty((C, int(height), int(width)), dtype=torch.float32, device=tiles.device)
    uncovered = torch.zeros(1, dtype=torch.int32, device=tiles.device)
    with torch.cuda.device(tiles.device):
        st = lib.drs_aggregate_tiles(_ptr(tiles), _ptr(org), _ptr(weight), _ptr(out), _ptr(uncovered), n, C, S,
                                     int(height), int(width), _stream(tiles.device))
    _lib.check(st, "drs_aggregate_tiles")
    if int(uncovered.item()) != 0:
        raise AssertionError("aggregation: some output pixels are covered by no tile (pixel_count == 0)")
    return out

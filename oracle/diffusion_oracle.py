"""TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.

CPU restatement of the `Diffusion` arithmetic of reference train_diffusion_superres.py: noise schedules
(:128-169), q-sample (:171-190) and the ancestral sampling loop (:207-255).  Pinned by the golden vectors
of tools/make_golden.py (G2, G6, G7) — see tests/test_oracle_golden.py.
"""
import torch


def cosine_alpha_hat(noise_steps):
    """prepare_noise_schedule, cosine branch (:166-169): alpha_hat = f(t)/f(0), no clipping."""
    f_t = torch.cos((((torch.arange(noise_steps) / noise_steps) + 0.008) / (1 + 0.008)) * torch.pi / 2) ** 2
    return f_t / f_t[0]


def alpha_hat_to_beta(alpha_hat):
    """from_alpha_hat_to_beta (:143-148), same scalar-by-scalar loop."""
    beta = []
    for t in range(len(alpha_hat) - 1, 0, -1):
        beta.append(1 - (alpha_hat[t] / alpha_hat[t - 1]))
    beta.append(1 - alpha_hat[0])
    return torch.tensor(beta[::-1], dtype=alpha_hat.dtype)


def schedule(kind, noise_steps, beta_start=1e-4, beta_end=0.02):
    """Diffusion.__init__ schedule block (:117-126) -> (alpha, alpha_hat, beta)."""
    if kind == "linear":
        beta = torch.linspace(beta_start, beta_end, noise_steps)
        alpha = 1.0 - beta
        alpha_hat = torch.cumprod(alpha, dim=0)
    elif kind == "cosine":
        alpha_hat = cosine_alpha_hat(noise_steps)
        beta = alpha_hat_to_beta(alpha_hat)
        alpha = 1.0 - beta
    else:
        raise ValueError(kind)
    return alpha, alpha_hat, beta


def noise_images(x, t, alpha_hat, epsilon):
    """noise_images (:183-190) with epsilon supplied."""
    sqrt_alpha_hat = torch.sqrt(alpha_hat[t])[:, None, None, None]
    sqrt_one_minus_alpha_hat = torch.sqrt(1 - alpha_hat[t])[:, None, None, None]
    return sqrt_alpha_hat * x + sqrt_one_minus_alpha_hat * epsilon


def sampler_step(x, predicted_noise, noise, t, alpha, alpha_hat, beta):
    """One iteration of the sampling loop (:240-249); t is the (n,) long tensor."""
    a = alpha[t][:, None, None, None]
    ah = alpha_hat[t][:, None, None, None]
    b = beta[t][:, None, None, None]
    return 1 / torch.sqrt(a) * (x - ((1 - a) / (torch.sqrt(1 - ah))) * predicted_noise) + torch.sqrt(b) * noise


def sample(model, n, lr_img, noise_steps, alpha, alpha_hat, beta, magnification_factor, image_size, input_channels=3,
           noise_source=None, keep_steps=()):
    """Diffusion.sample (:224-255) on CPU.  `noise_source(i, shape)` supplies x_T (i = noise_steps) and z_i; without
    it the draws come from torch's CPU generator in the reference's order (x_T, then one randn_like per step)."""
    # a 4-D lr_img is the tiler's batched form (one LR image per chain): n independent reference chains run as one batch
    lr = lr_img if lr_img.dim() == 4 else lr_img.unsqueeze(0)
    shape = (n, input_channels, image_size, image_size)
    kept = {}
    with torch.no_grad():
        x = noise_source(noise_steps, shape) if noise_source else torch.randn(shape)
        for i in reversed(range(1, noise_steps)):
            t = (torch.ones(n) * i).long()
            eps = model(x, t, lr, magnification_factor)
            if i > 1:
                z = noise_source(i, shape) if noise_source else torch.randn_like(x)
            else:
                z = torch.zeros_like(x)
            x = sampler_step(x, eps, z, t, alpha, alpha_hat, beta)
            if i in keep_steps:
                kept[i] = x.clone()
    return (x, kept) if keep_steps else x


def sample_sar(model, n, sar_img, noise_steps, alpha, alpha_hat, beta, image_size, ndvi_channels=1, noise_source=None):
    """Diffusion.sample of train_diffusion_SAR_TO_NDVI.py:204-249 on CPU: same chain, model(x, t, SAR_img)."""
    sar = sar_img.unsqueeze(0)
    shape = (n, ndvi_channels, image_size, image_size)
    with torch.no_grad():
        x = noise_source(noise_steps, shape) if noise_source else torch.randn(shape)
        for i in reversed(range(1, noise_steps)):
            t = (torch.ones(n) * i).long()
            eps = model(x, t, sar)
            z = (noise_source(i, shape) if noise_source else torch.randn_like(x)) if i > 1 else torch.zeros_like(x)
            x = sampler_step(x, eps, z, t, alpha, alpha_hat, beta)
    return x


def sample_generation(model, n, target_class, cfg_scale, noise_steps, alpha, alpha_hat, beta, image_size,
                      input_channels=3, noise_source=None):
    """Diffusion.sample of generate_new_imgs/train_diffusion_generation.py:206-259 on CPU: conditional prediction,
    and for cfg_scale > 0 an unconditional one combined with torch.lerp(uncond, cond, cfg_scale) (:236-239)."""
    shape = (n, input_channels, image_size, image_size)
    with torch.no_grad():
        x = noise_source(noise_steps, shape) if noise_source else torch.randn(shape)
        for i in reversed(range(1, noise_steps)):
            t = (torch.ones(n) * i).long()
            eps = model(x, t, target_class)
            if cfg_scale > 0:
                eps = torch.lerp(model(x, t, None), eps, cfg_scale)
            z = (noise_source(i, shape) if noise_source else torch.randn_like(x)) if i > 1 else torch.zeros_like(x)
            x = sampler_step(x, eps, z, t, alpha, alpha_hat, beta)
    return x

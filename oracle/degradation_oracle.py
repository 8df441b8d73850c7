"""TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.

CPU restatement (numpy, integer arithmetic) of the "DownBlur" degradation the reference's dataset applies per item
(utils.py:140-158: `transforms.Resize(BICUBIC)` on a PIL image -> `ImageFilter.GaussianBlur(radius)` -> `ToTensor`).

The arithmetic lives in a third-party dependency that is not under /root/reference: Pillow (`pillow==10.2.0` in the
reference's requirements.txt; 12.2.0 in the build container) - libImaging/Resample.c (8-bit separable resize with
22-bit fixed-point coefficients) and libImaging/BoxBlur.c (Gaussian blur as 3+3 passes of a fractional box filter).
This file restates that published algorithm; it is pinned bit-for-bit by fixtures produced with Pillow itself on the
reference's call sequence (tools/make_golden_degradation.py -> tests/golden/degradation_golden.npz,
tests/test_oracle_golden.py).  Only tests/ may import this file.
"""
import numpy as np


def _bicubic(x, a=-0.5):
    """Resample.c bicubic_filter (a = -0.5)."""
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc: per output pixel (first input pixel, 22-bit weights)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(k)
        if ww != 0.0:
            k = [v / ww for v in k]
        kk = [int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22)) for v in k]
        out.append((xmin, kk))
    return out


def _resize_axis(img, out_size, axis):
    img = np.moveaxis(img, axis, -1)
    res = np.empty(img.shape[:-1] + (out_size,), dtype=np.uint8)
    for xx, (xmin, kk) in enumerate(_coeffs(img.shape[-1], out_size)):
        acc = np.full(img.shape[:-1], 1 << 21, dtype=np.int64)
        for j, k in enumerate(kk):
            acc += img[..., xmin + j].astype(np.int64) * k
        res[..., xx] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
    return np.moveaxis(res, -1, axis)


def resize_bicubic_u8(img, out_h, out_w):
    """PIL `Image.resize((out_w, out_h), BICUBIC)` on (..., H, W) uint8 planes: horizontal pass, then vertical pass
    (each only when that size changes), uint8 in between."""
    if img.shape[-1] != out_w:
        img = _resize_axis(img, out_w, -1)
    if img.shape[-2] != out_h:
        img = _resize_axis(img, out_h, -2)
    return img


def gaussian_box_radius(radius, passes=3):
    """BoxBlur.c _gaussian_blur_radius with its float / double promotions."""
    f = np.float32
    radius = f(radius)
    sigma2 = f(radius * radius / f(passes))
    L = f(np.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(np.floor((float(L) - 1.0) / 2.0))
    a = f(f(2 * l + 1) * f(l * f(l + 1) - f(3) * sigma2))
    a = f(a / f(f(6) * f(sigma2 - f(l + 1) * f(l + 1))))
    return f(l + a)


def _box_axis(img, fr, axis):
    img = np.moveaxis(img, axis, -1).astype(np.int64)
    n = img.shape[-1]
    r = int(fr)
    ww = int(np.float32(16777216.0) / np.float32(np.float32(fr) * np.float32(2) + np.float32(1)))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    idx = np.arange(n)
    s = np.zeros_like(img)
    for d in range(-r, r + 1):
        s += img[..., np.clip(idx + d, 0, n - 1)]
    far = img[..., np.clip(idx - r - 1, 0, n - 1)] + img[..., np.clip(idx + r + 1, 0, n - 1)]
    out = (((s * ww + far * fw) & 0xFFFFFFFF) + (1 << 23)) >> 24
    return np.moveaxis(out.astype(np.uint8), -1, axis)


def gaussian_blur_u8(img, radius):
    """PIL `ImageFilter.GaussianBlur(radius)` on (..., H, W) uint8 planes: 3 horizontal, then 3 vertical box passes."""
    if radius <= 0:
        return img
    fr = gaussian_box_radius(radius)
    if fr == 0:
        return img
    for _ in range(3):
        img = _box_axis(img, fr, -1)
    for _ in range(3):
        img = _box_axis(img, fr, -2)
    return img


def downblur(hr_u8, out_h, out_w, blur_radius):
    """get_data_superres.__getitem__ (utils.py:140-158, Gauss_noise=False) for (..., H, W) uint8 planes:
    returns (x, y) float32 in [0, 1]."""
    lr = gaussian_blur_u8(resize_bicubic_u8(hr_u8, out_h, out_w), blur_radius)
    return lr.astype(np.float32) / np.float32(255), hr_u8.astype(np.float32) / np.float32(255)


def add_gaussian_noise(img_chw, noise_level1=2, noise_level2=25):
    """The `Gauss_noise=True` step of the dataset item (reference utils.py:15-38, called at :163-164 with levels 2 and 10)
    for ONE (C, H, W) float32 image in [0, 1].  Randomness comes, in this order, from Python's `random` (the level) and
    from numpy's global generator (the branch draw, then the branch's own draws); the caller seeds both.  Three branches:
    per-channel white noise (draw > 0.6), one noise plane shared by the channels (draw < 0.4), or noise with a random
    3 x 3 channel covariance whose scale is tied to noise_level2 (otherwise).  Noise is rounded to float32 before it is
    added, the sum is clipped to [0, 1].  Returns a new (C, H, W) float32 array."""
    import random

    from scipy.linalg import orth

    level = random.randint(noise_level1, noise_level2)
    draw = np.random.rand()
    hwc = np.ascontiguousarray(np.transpose(np.asarray(img_chw, dtype=np.float32), (1, 2, 0))).copy()
    h, w = hwc.shape[:2]
    if draw > 0.6:
        hwc += np.random.normal(0, level / 255.0, hwc.shape).astype(np.float32)
    elif draw < 0.4:
        hwc += np.random.normal(0, level / 255.0, (h, w, 1)).astype(np.float32)
    else:
        scale = noise_level2 / 255.
        diag = np.diag(np.random.rand(3))
        basis = orth(np.random.rand(3, 3))
        cov = np.dot(np.dot(np.transpose(basis), diag), basis)
        hwc += np.random.multivariate_normal([0, 0, 0], np.abs(scale ** 2 * cov), (h, w)).astype(np.float32)
    return np.ascontiguousarray(np.transpose(np.clip(hwc, 0.0, 1.0), (2, 0, 1))).astype(np.float32)

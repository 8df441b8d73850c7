#!/usr/bin/env python3
"""Fixtures for the DownBlur degradation, produced with Pillow itself on the call sequence of the reference's dataset
item (utils.py:140-158): `transforms.Resize((y.size[0] // m, y.size[1] // m), BICUBIC)(y)` (which torchvision executes as
`y.resize(size[::-1], Image.BICUBIC)`), `.filter(ImageFilter.GaussianBlur(r))`, `ToTensor()`.  torchvision is absent
from the build container, so its two one-line wrappers are spelled out; Pillow does the arithmetic.
    python tools/make_golden_degradation.py   ->  tests/golden/degradation_golden.npz (inputs and outputs, uint8)
"""
import os

import numpy as np
import PIL
from PIL import Image, ImageFilter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(2024)
g = {"pillow_version": np.array(PIL.__version__)}
cases = [("sq64_m2_r05", 64, 64, 2, 0.5), ("sq96_m4_r1", 96, 96, 4, 1.0), ("rect48x80_m2_r137", 48, 80, 2, 1.37),
         ("sq32_m2_r0", 32, 32, 2, 0.0), ("rect40x24_m4_r09", 40, 24, 4, 0.9), ("sq128_m8_r15", 128, 128, 8, 1.5)]
for tag, h, w, m, r in cases:
    hr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # smooth half of the images so that the filters are exercised on image-like content too
    if tag.startswith("sq96") or tag.startswith("rect48"):
        hr = np.asarray(Image.fromarray(hr).filter(ImageFilter.GaussianBlur(2.0)))
    y = Image.fromarray(hr)
    size = (y.size[0] // m, y.size[1] // m)          # reference quirk: (W // m, H // m) passed as (h, w)
    x = y.resize(size[::-1], Image.BICUBIC)          # torchvision.transforms.functional resize on a PIL image
    if r > 0:
        x = x.filter(ImageFilter.GaussianBlur(r))
    g[f"{tag}_hr"] = hr
    g[f"{tag}_lr"] = np.asarray(x)
    g[f"{tag}_params"] = np.array([h, w, m, r], dtype=np.float64)
# single-channel (mode L) image: the SAR / NDVI style planes
hr = rng.integers(0, 256, (56, 56), dtype=np.uint8)
g["gray56_m2_r07_hr"] = hr
g["gray56_m2_r07_lr"] = np.asarray(Image.fromarray(hr).resize((28, 28), Image.BICUBIC).filter(ImageFilter.GaussianBlur(0.7)))
g["gray56_m2_r07_params"] = np.array([56, 56, 2, 0.7])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "degradation_golden.npz"), **g)
print("wrote", len(g), "arrays with Pillow", PIL.__version__)

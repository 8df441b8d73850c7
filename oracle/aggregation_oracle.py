"""TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.

CPU restatement of the reference's tile split / Gaussian blend (Aggregation_Sampling.py:24-138), as pure functions.
Pinned by tests/golden/variants_golden.npz (G10: tile coordinates, weights and the blended image produced by the
imported reference class around the reference Diffusion.sample; tools/make_golden_variants.py).
Only tests/ may import this file.
"""
from math import exp, pi, sqrt

import numpy as np
import torch


def tile_infos(height, width, patch_size, stride, magnification_factor):
    """patchifier, :24-68: (y0, y1, x0, x1) in super-resolved coordinates, reference order, duplicates dropped;
    also the LR-space (y0, x0) of every kept tile."""
    infos, lr_origins = [], []
    m = magnification_factor
    for y in range(0, height + 1, stride):
        for x in range(0, width + 1, stride):
            ys = height - patch_size if y + patch_size > height else y
            xs = width - patch_size if x + patch_size > width else x
            info = (ys * m, (ys + patch_size) * m, xs * m, (xs + patch_size) * m)
            if info not in infos:
                infos.append(info)
                lr_origins.append((ys, xs))
    return infos, lr_origins


def gaussian_weight(tile_width, tile_height):
    """gaussian_weights, :118-138 (one (H, W) float32 plane; the reference tiles it over batch and 3 channels)."""
    var = 0.01
    mx = (tile_width - 1) / 2
    xp = [exp(-(x - mx) * (x - mx) / (tile_width * tile_width) / (2 * var)) / sqrt(2 * pi * var) for x in range(tile_width)]
    my = tile_height / 2
    yp = [exp(-(y - my) * (y - my) / (tile_height * tile_height) / (2 * var)) / sqrt(2 * pi * var) for y in range(tile_height)]
    return torch.tensor(np.outer(yp, xp)).to(torch.float32)


def aggregate(tiles, infos, weight, height, width):
    """aggregation_sampling, :90-116, given the super-resolved tiles (n, C, S, S): sequential `+=` in tile order,
    division by the summed weights, clamp to [0, 1]."""
    C = tiles.shape[1]
    im = torch.zeros((1, C, height, width), dtype=torch.float32)
    cnt = torch.zeros((1, C, height, width), dtype=torch.float32)
    w = weight[None, None].expand(1, C, -1, -1)
    for i, (y0, y1, x0, x1) in enumerate(infos):
        im[:, :, y0:y1, x0:x1] += tiles[i:i + 1] * w
        cnt[:, :, y0:y1, x0:x1] += w
    assert torch.all(cnt != 0)
    im /= cnt
    return torch.clamp(im, 0, 1)

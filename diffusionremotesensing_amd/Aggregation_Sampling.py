"""Drop-in `split_aggregation_sampling` (reference Aggregation_Sampling.py:9-138): super-resolve an image larger than
the UNet's training size tile by tile and blend the tiles with Gaussian weights.

Kept from the reference: constructor signature and attributes, `patchifier` (tile coordinates, including the
clamped-last-tile rule and the de-duplication), `gaussian_weights` (including its asymmetric midpoints: (W-1)/2 along x,
H/2 along y, :133-137), the accumulation order and the final normalise + clamp.

MI355X-first differences:
  * all tiles run through ONE `Diffusion.sample` call as a batch with one LR tile per sample (the reference runs
    len(tiles) sequential n=1 chains, 1499 batch-1 forwards each); under torch.distributed the tiles are sharded over the
    ranks (independent chains, no collective in the loop) and gathered once;
  * blend + normalise + clamp is one gather kernel (`drs_aggregate_tiles`) instead of two full-size accumulators and
    4 element-wise passes per tile.
Noise: the reference draws x_T and every z_i tile after tile from the global CPU generator; a batched run cannot
reproduce that stream order, so by default the n tiles draw one (n,C,S,S) tensor per step.  `noise_source(tile, i,
shape)` lets a caller (the parity tests) supply the reference's per-tile draws.
"""
from math import exp, pi, sqrt

import numpy as np
import torch

from . import dist as drs_dist
from . import hip_ops


# tiles per sampling chain (per rank): bounds the plan workspace (~0.1 GB per 256x256 tile) independently of the scene size
TILE_BATCH = 16


class split_aggregation_sampling:
    def __init__(self, img_lr, patch_size, stride, magnification_factor, diffusion_model, device):
        assert stride <= patch_size
        self.img_lr = img_lr
        self.patch_size = patch_size
        self.stride = stride
        self.magnification_factor = magnification_factor
        self.diffusion_model = diffusion_model
        self.device = device
        self.model = diffusion_model.model
        batch_size, channels, height, width = img_lr.shape
        self.patches_lr, self.patches_sr_infos = self.patchifier(img_lr, patch_size, stride, magnification_factor)
        self.weight = self.gaussian_weights(patch_size * magnification_factor, patch_size * magnification_factor,
                                            batch_size)

    def patchifier(self, img_to_split, patch_size, stride=None, magnification_factor=1):
        """Reference :24-68: tiles on a `stride` grid, the last one of a row / column clamped to the border,
        duplicates dropped; infos are (y0, y1, x0, x1) in super-resolved coordinates."""
        if stride is None:
            stride = patch_size
        batch_size, channels, height, width = img_to_split.shape
        patches_lr, patches_sr_infos = [], []
        m = magnification_factor
        for y in range(0, height + 1, stride):
            for x in range(0, width + 1, stride):
                y_start, y_end = (height - patch_size, height) if y + patch_size > height else (y, y + patch_size)
                x_start, x_end = (width - patch_size, width) if x + patch_size > width else (x, x + patch_size)
                info = (y_start * m, y_end * m, x_start * m, x_end * m)
                if info not in patches_sr_infos:
                    patches_lr.append(img_to_split[:, :, y_start:y_end, x_start:x_end])
                    patches_sr_infos.append(info)
        return patches_lr, patches_sr_infos

    def gaussian_weights(self, tile_width, tile_height, nbatches):
        """Reference :118-138 (float64 Python arithmetic, then float32; tiled to (nbatches, 3, H, W))."""
        var = 0.01
        midpoint = (tile_width - 1) / 2
        x_probs = [exp(-(x - midpoint) * (x - midpoint) / (tile_width * tile_width) / (2 * var)) / sqrt(2 * pi * var)
                   for x in range(tile_width)]
        midpoint = tile_height / 2
        y_probs = [exp(-(y - midpoint) * (y - midpoint) / (tile_height * tile_height) / (2 * var)) / sqrt(2 * pi * var)
                   for y in range(tile_height)]
        weights = torch.tensor(np.outer(y_probs, x_probs)).to(torch.float32).to(self.device)
        return torch.tile(weights, (nbatches, 3, 1, 1))

    def sample_tiles(self, noise_source=None):
        """Super-resolve every tile: (n_tiles, C, S, S) on this rank's device.  One batched chain (sharded over the
        ranks of an initialised process group)."""
        d = self.diffusion_model
        lr = torch.cat([p[:1] for p in self.patches_lr], dim=0).to(self.device).contiguous()  # (n, C, ps, ps)
        n = lr.shape[0]
        lo, hi = drs_dist.shard_range(n) if drs_dist.world_size() > 1 else (0, n)
        src = None
        if noise_source is not None:
            def src(i, shape, lo=lo):  # stack the per-tile draws of this rank's tiles
                return torch.cat([noise_source(lo + k, i, (1,) + tuple(shape[1:])) for k in range(shape[0])], dim=0)
        # fixed-size chunks of this rank's tiles: one chain per chunk, ONE plan / workspace whatever the scene size (a
        # 2048 x 2048 scene at stride 32 has ~4000 tiles: as a single batch its workspace would not fit any device).
        # The last chunk is padded with repeats of its last tile so that it runs on the same plan.
        chunk = max(1, int(getattr(self, "tile_batch", 0) or TILE_BATCH))
        outs = []
        for c0 in range(lo, hi, chunk):
            c1 = min(c0 + chunk, hi)
            take = c1 - c0
            size = take if (hi - lo) <= chunk else chunk  # a scene smaller than one chunk runs at its own size
            lr_c = lr[c0:c1]
            if size > take:
                lr_c = torch.cat([lr_c, lr_c[-1:].expand(size - take, -1, -1, -1)], dim=0).contiguous()
            csrc = None
            if src is not None:
                def csrc(i, shape, c0=c0, take=take):
                    real = src(i, (take,) + tuple(shape[1:]), lo=c0)
                    if shape[0] > take:
                        real = torch.cat([real, real[-1:].expand(shape[0] - take, -1, -1, -1)], dim=0)
                    return real
            out_c = d.sample(size, self.model, lr_c, input_channels=lr.shape[1], generate_video=False, noise_source=csrc)
            outs.append(out_c[:take])
        mine = torch.cat(outs, dim=0) if outs else lr.new_zeros((0, lr.shape[1], d.image_size, d.image_size))
        if drs_dist.world_size() > 1:
            mine = drs_dist.gather_shards(mine, n)
        return mine

    def aggregation_sampling(self, noise_source=None):
        """Reference :76-116."""
        batch_size, channels, height, width = self.img_lr.shape
        m = self.magnification_factor
        tiles = self.sample_tiles(noise_source)
        origins = [(info[0], info[2]) for info in self.patches_sr_infos]
        out = hip_ops.aggregate_tiles(tiles, origins, self.weight[0, 0].contiguous(), height * m, width * m)
        # the reference broadcasts the single chain of each tile over the batch dimension of img_lr
        return out.unsqueeze(0).expand(batch_size, -1, -1, -1).contiguous()


def launch(args):
    """Reference launch (:140-212): model + snapshot + Diffusion + tiler.  The image file I/O of the reference
    (PIL / torchvision.transforms) is outside the hot path: `--img_lr_path` takes a `.pt` / `.npy` tensor (C,H,W) or
    (1,C,H,W) in [0,1], `--destination_path` receives a `.pt` tensor."""
    import os

    from .train_diffusion_superres import Diffusion
    from .UNet_model_superres import Residual_Attention_UNet_superres
    device = args.device
    if args.UNet_type.lower() != "residual attention unet":
        raise ValueError("The UNet type must be Residual Attention UNet")
    model = Residual_Attention_UNet_superres(args.inp_out_channels, args.inp_out_channels, device).to(device)
    print(f"You are using {args.UNet_type} model")
    path = args.img_lr_path
    img_lr = torch.from_numpy(np.load(path)) if path.endswith(".npy") else torch.load(path)
    img_lr = img_lr.float()
    if img_lr.dim() == 3:
        img_lr = img_lr.unsqueeze(0)
    img_lr = img_lr.to(device)
    diffusion = Diffusion(noise_schedule=args.noise_schedule, model=model,
                          snapshot_path=os.path.join(args.snapshot_folder_path, args.snapshot_name),
                          noise_steps=args.noise_steps, beta_start=1e-4, beta_end=0.02,
                          magnification_factor=args.magnification_factor, device=device,
                          image_size=args.model_input_size, model_name=args.model_name,
                          Degradation_type=args.Degradation_type, multiple_gpus=False, ema_smoothing=False)
    tiler = split_aggregation_sampling(img_lr, args.patch_size, args.stride, args.magnification_factor, diffusion, device)
    final_pred = tiler.aggregation_sampling()
    torch.save(final_pred.squeeze(0).cpu(), args.destination_path)


def build_arg_parser():
    """The reference's flags, verbatim (:217-231)."""
    import argparse
    p = argparse.ArgumentParser(description=" ")
    p.add_argument("--noise_schedule", type=str, default="cosine")
    p.add_argument("--snapshot_name", type=str, default="snapshot.pt")
    p.add_argument("--noise_steps", type=int, default=1500)
    p.add_argument("--model_input_size", type=int, default=512)
    p.add_argument("--model_name", type=str)
    p.add_argument("--UNet_type", type=str)
    p.add_argument("--Degradation_type", type=str)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--magnification_factor", type=int)
    p.add_argument("--inp_out_channels", type=int, default=3)
    p.add_argument("--patch_size", type=int, default=64)
    p.add_argument("--stride", type=int, default=32)
    p.add_argument("--destination_path", type=str)
    p.add_argument("--img_lr_path", type=str)
    return p


if __name__ == "__main__":
    import os
    a = build_arg_parser().parse_args()
    a.snapshot_folder_path = os.path.join(os.curdir, "models_run", a.model_name, "weights")
    launch(a)

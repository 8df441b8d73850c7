"""On-device "DownBlur" data feed for the super-resolution trainer.

The reference builds every (lr, hr) training pair on the host's main thread with Pillow
(`utils.get_data_superres.__getitem__`, utils.py:126-166: bicubic down-sampling by the magnification factor, Gaussian
blur, ToTensor; `num_workers=0`).  Here the HR images live in a uint8 cache on the device and `drs_downblur_u8` produces
the same pair, bit for bit, for a whole batch (tests/test_gpu_degradation.py against Pillow fixtures).

Kept from the reference: the `(y.size[0] // m, y.size[1] // m)` size expression, which hands (W // m, H // m) to a
(height, width) argument (non-square images come out transposed in size, utils.py:141-142), and `blur_radius='random'`
being drawn ONCE per dataset object with `random.triangular(0.5, 1.5, 1)` (quirk Q10, utils.py:151-152).
Not covered: `Gauss_noise=True` (host numpy RNG noise, utils.py:15-38) and the BSRGAN degradation.
"""
import ctypes as C
import random

import torch

from . import _lib


def downblur(hr_u8, magnification_factor, blur_radius):
    """(x_lr, y_hr) float32 in [0, 1] for a (N, C, H, W) uint8 ROCm tensor: the reference's dataset item, batched."""
    lib = _lib.load()
    if not isinstance(hr_u8, torch.Tensor) or not hr_u8.is_cuda or hr_u8.dtype != torch.uint8 or hr_u8.dim() != 4:
        raise RuntimeError("downblur: hr must be a (N, C, H, W) uint8 tensor on a ROCm device (no CPU fallback)")
    hr_u8 = hr_u8.contiguous()
    n, c, h, w = hr_u8.shape
    m = int(magnification_factor)
    out_h, out_w = w // m, h // m  # the reference's (y.size[0] // m, y.size[1] // m) handed to Resize as (h, w)
    if out_h < 1 or out_w < 1:
        raise RuntimeError(f"downblur: {h}x{w} is too small for magnification {m}")
    x = torch.empty((n, c, out_h, out_w), dtype=torch.float32, device=hr_u8.device)
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=hr_u8.device)
    nbytes = lib.drs_downblur_scratch_bytes(n, c, h, w, out_h, out_w)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=hr_u8.device)
    with torch.cuda.device(hr_u8.device):
        st = lib.drs_downblur_u8(C.c_void_p(hr_u8.data_ptr()), n, c, h, w, out_h, out_w, float(blur_radius),
                                 C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(scratch.data_ptr()),
                                 nbytes, C.c_void_p(torch.cuda.current_stream(hr_u8.device).cuda_stream))
    _lib.check(st, "drs_downblur_u8")
    return x, y


class DeviceSuperresFeed:
    """Iterable of (lr, hr) float batches drawn from a uint8 HR cache on the device: what
    `DataLoader(get_data_superres(root, magnification_factor, blur_radius), batch_size, shuffle)` yields, without
    the per-item PIL work.  `hr_u8`: (L, C, H, W) uint8 on the device (the decoded dataset)."""

    def __init__(self, hr_u8, magnification_factor, blur_radius=0.5, batch_size=16, shuffle=True, generator=None):
        self.hr = hr_u8
        self.magnification_factor = magnification_factor
        if blur_radius == "random":  # drawn once per dataset object, like the reference
            blur_radius = random.triangular(0.5, 1.5, 1)
        self.blur_radius = blur_radius
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.generator = generator

    def __len__(self):
        return (self.hr.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.hr.shape[0]
        order = torch.randperm(n, generator=self.generator) if self.shuffle else torch.arange(n)
        order = order.to(self.hr.device)
        for i in range(0, n, self.batch_size):
            yield downblur(self.hr[order[i:i + self.batch_size]], self.magnification_factor, self.blur_radius)

"""On-device "DownBlur" data feed for the super-resolution trainer.

The reference builds every (lr, hr) training pair on the host's main thread with Pillow
(`utils.get_data_superres.__getitem__`, utils.py:126-166: bicubic down-sampling by the magnification factor, Gaussian
blur, ToTensor; `num_workers=0`).  Here the HR images live in a uint8 cache on the device and `drs_downblur_u8` produces
the same pair, bit for bit, for a whole batch (tests/test_gpu_degradation.py against Pillow fixtures).

Kept from the reference: the `(y.size[0] // m, y.size[1] // m)` size expression, which hands (W // m, H // m) to a
(height, width) argument (non-square images come out transposed in size, utils.py:141-142), and `blur_radius='random'`
being drawn ONCE per dataset object with `random.triangular(0.5, 1.5, 1)` (quirk Q10, utils.py:151-152).
`Gauss_noise=True` (Degradation_type DownBlurNoise, utils.py:15-38,163-164): the noise comes from the host generators the
reference uses (Python `random` for the level, numpy's global generator for everything else) drawn in its order, item by
item (`reference_noise`); the add and the clip run on the device (`drs_add_noise_clip_f32`).  Seeded alike, the feed
reproduces the reference's items bit for bit (tests/test_gpu_degradation.py, fixtures from the reference function itself).
`load_image_folder_u8` fills the cache from an image folder (decode + the launch transform's resize with Pillow, once).
Not covered: the BSRGAN degradation.
"""
import ctypes as C
import random

import numpy as np
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


def reference_noise(c, h, w, noise_level1=2, noise_level2=25):
    """The noise term `add_Gaussian_noise` (utils.py:15-38) adds to ONE (c, h, w) image, as an (h, w, c) float32 array,
    consuming Python's `random` and numpy's global generator exactly as the reference does: one `random.randint` for the
    level, one `np.random.rand` for the branch, then the branch's draws - white noise per channel (> 0.6), one plane for
    all channels (< 0.4), or 3-channel noise with a random covariance scaled by noise_level2.  The two 3-channel branches
    need c == 3, as in the reference (numpy raises on the broadcast)."""
    from scipy.linalg import orth
    level = random.randint(noise_level1, noise_level2)
    draw = np.random.rand()
    if draw > 0.6:
        noise = np.random.normal(0, level / 255.0, (h, w, c)).astype(np.float32)
    elif draw < 0.4:
        noise = np.broadcast_to(np.random.normal(0, level / 255.0, (h, w, 1)).astype(np.float32), (h, w, c))
    else:
        if c != 3:
            raise ValueError(f"operands could not be broadcast together with shapes ({h},{w},{c}) ({h},{w},3)")
        scale = noise_level2 / 255.
        diag = np.diag(np.random.rand(3))
        basis = orth(np.random.rand(3, 3))
        cov = np.dot(np.dot(np.transpose(basis), diag), basis)
        noise = np.random.multivariate_normal([0, 0, 0], np.abs(scale ** 2 * cov), (h, w)).astype(np.float32)
    return np.ascontiguousarray(noise)


def _check_noise_target(x):
    if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous():
        raise RuntimeError("add_reference_noise: x must be a contiguous (N, C, H, W) float32 tensor on a ROCm device")


def reference_noise_batch(n, c, h, w, noise_level1=2, noise_level2=10, pin=False):
    """The noise of n consecutive dataset items (reference_noise, item by item in batch order) as one (n, h, w, c) float32 host
    tensor; `pin`: in page-locked memory, for an asynchronous upload."""
    noise = torch.from_numpy(np.stack([reference_noise(c, h, w, noise_level1, noise_level2) for _ in range(n)]))
    return noise.pin_memory() if pin and torch.cuda.is_available() else noise


def add_noise_clip_(x, noise_host):
    """In place: x (N, C, H, W) float32 on the device += noise_host (N, H, W, C), clipped to [0, 1] (drs_add_noise_clip_f32)."""
    _check_noise_target(x)
    n, c, h, w = x.shape
    if tuple(noise_host.shape) != (n, h, w, c):
        raise RuntimeError(f"add_noise_clip_: noise {tuple(noise_host.shape)} for a batch {tuple(x.shape)}")
    lib = _lib.load()
    nd = noise_host.to(x.device, non_blocking=True)
    with torch.cuda.device(x.device):
        st = lib.drs_add_noise_clip_f32(C.c_void_p(x.data_ptr()), C.c_void_p(nd.data_ptr()), n, c, h, w,
                                        C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    _lib.check(st, "drs_add_noise_clip_f32")
    return x


def add_reference_noise(x, noise_level1=2, noise_level2=10):
    """In place: x (N, C, H, W) float32 on the device += the reference's noise, item by item in batch order, clipped to
    [0, 1] (the dataset item's `add_Gaussian_noise(x, noise_level1=2, noise_level2=10)`, utils.py:163-164)."""
    _check_noise_target(x)
    n, c, h, w = x.shape
    if n == 0:
        return x
    return add_noise_clip_(x, reference_noise_batch(n, c, h, w, noise_level1, noise_level2))


class NoisePrefetcher:
    """The DownBlurNoise feed's host side, off the training thread: a worker draws the batches' noise AHEAD - same generators
    (Python's `random`, numpy's global state), same order of draws as the reference's dataset items (utils.py:15-38), batch after
    batch - into page-locked buffers, at most `depth` batches ahead; the training thread only uploads (asynchronously) and adds.
    One 16 x 3 x 128 x 128 batch takes ~20 ms to draw, as long as a training step: drawn inline (round 4) it made the trainer
    feed-bound by ~2.4x.  numpy's legacy generators release the GIL while they fill an array.  Nothing else in the training
    process draws from these two generators while an epoch runs (timesteps and the diffusion noise use torch's)."""

    def __init__(self, sizes, c, h, w, noise_level1=2, noise_level2=10, depth=2):
        import queue
        import threading
        self._q = queue.Queue(maxsize=depth)
        self._stop = threading.Event()
        self._error = None

        def work():
            try:
                for n in sizes:
                    if self._stop.is_set():
                        return
                    item = reference_noise_batch(n, c, h, w, noise_level1, noise_level2, pin=True)
                    while not self._stop.is_set():
                        try:
                            self._q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
            except BaseException as e:  # reported on the training thread by next()
                self._error = e
                self._q.put(None)
        self._t = threading.Thread(target=work, name="drs-noise-prefetch", daemon=True)
        self._t.start()

    def next(self):
        item = self._q.get()
        if item is None:
            raise self._error
        return item

    def close(self):
        self._stop.set()
        self._t.join(timeout=5)


def load_image_folder_u8(root_dir, image_size=None, rank=0, world_size=1, limit=None):
    """Decode side of the reference's `get_data_superres` (utils.py:93-138) + the `transforms.Resize((image_size,
    image_size))` its `launch` hands it as `transform` (train_diffusion_superres.py:594-605): the images of `root_dir` in
    `sorted(os.listdir(...))` order, opened with Pillow (`data_format='PIL'`), resized to image_size x image_size when they
    are not already (torchvision's Resize on a PIL image is `img.resize((w, h), BILINEAR)`: the same Pillow call is made
    here, so the bytes are the reference's), stacked as ONE (L, C, S, S) uint8 tensor on the host - the HR cache
    `DeviceSuperresFeed` then keeps on the device.  Everything after the decode (bicubic down-sampling, blur, noise,
    ToTensor) happens per batch on the device.  Done once per dataset, not per item per epoch.

    `rank` / `world_size`: decode only this rank's shard (every world-th file, equal shard sizes: the ranks must run the same
    number of steps per epoch, so len % world_size files - always the last ones of the sorted list - are left out, and the count
    is printed; DistributedSampler pads with repeats instead).  `limit`: only the first `limit` files of the (unsharded) list -
    the reference's `train_dataset[0..4]` of the final sampling.  Entries that are not files are skipped like any directory
    listing tool would; a file Pillow cannot read raises with its name.  8-bit modes L / RGB / RGBA (what ToTensor turns into
    1 / 3 / 4 channels of uint8 / 255); other modes raise."""
    import os

    from PIL import Image, UnidentifiedImageError
    names = [n for n in sorted(os.listdir(root_dir)) if os.path.isfile(os.path.join(root_dir, n))]
    if limit is not None:
        names = names[:limit]
    else:
        per_rank = len(names) // world_size
        if per_rank == 0:
            raise ValueError(f"dataset of {len(names)} images in {root_dir} cannot be sharded over {world_size} ranks")
        if len(names) % world_size and rank == 0:
            print(f"{root_dir}: {len(names) % world_size} of {len(names)} images left out (equal shards over {world_size} ranks)")
        names = names[rank::world_size][:per_rank]
    planes = []
    for name in names:
        try:
            img = Image.open(os.path.join(root_dir, name))
        except (UnidentifiedImageError, OSError) as e:
            raise ValueError(f"{os.path.join(root_dir, name)}: not an image Pillow can read ({e}); the dataset folders must hold "
                             "image files only") from e
        with img as y:
            y.load()
            if y.mode not in ("L", "RGB", "RGBA"):
                raise ValueError(f"{name}: image mode {y.mode!r} is not an 8-bit L / RGB / RGBA image")
            if image_size is not None and y.size != (image_size, image_size):
                y = y.resize((image_size, image_size), Image.BILINEAR)
            a = np.asarray(y, dtype=np.uint8)
        planes.append(a[None] if a.ndim == 2 else np.moveaxis(a, -1, 0))
    shapes = {p.shape for p in planes}
    if len(shapes) != 1:
        raise ValueError(f"images of {root_dir} differ in shape {sorted(shapes)}: pass image_size, as the reference's launch does")
    return torch.from_numpy(np.ascontiguousarray(np.stack(planes)))


class DeviceSuperresFeed:
    """Iterable of (lr, hr) float batches drawn from a uint8 HR cache on the device: what
    `DataLoader(get_data_superres(root, magnification_factor, blur_radius), batch_size, shuffle)` yields, without
    the per-item PIL work.  `hr_u8`: (L, C, H, W) uint8 on the device (the decoded dataset)."""

    def __init__(self, hr_u8, magnification_factor, blur_radius=0.5, batch_size=16, shuffle=True, generator=None,
                 Gauss_noise=False):
        self.hr = hr_u8
        self.magnification_factor = magnification_factor
        if blur_radius == "random":  # drawn once per dataset object, like the reference
            blur_radius = random.triangular(0.5, 1.5, 1)
        self.blur_radius = blur_radius
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.generator = generator
        self.Gauss_noise = Gauss_noise  # reference get_data_superres(..., Gauss_noise): levels 2 .. 10 per item

    def __len__(self):
        return (self.hr.shape[0] + self.batch_size - 1) // self.batch_size

    def item(self, idx):
        """(x, y) of one dataset item, (C, h, w) / (C, H, W): `dataset[idx]` of the reference's Dataset (no noise draw is
        consumed for Gauss_noise feeds: previews and the final sampling only need the degraded image's shape and content)."""
        x, y = downblur(self.hr[idx:idx + 1], self.magnification_factor, self.blur_radius)
        return x[0], y[0]

    def __iter__(self):
        n = self.hr.shape[0]
        order = torch.randperm(n, generator=self.generator) if self.shuffle else torch.arange(n)
        order = order.to(self.hr.device)
        noise = None
        if self.Gauss_noise and n:
            m = int(self.magnification_factor)
            c, h, w = self.hr.shape[1], self.hr.shape[3] // m, self.hr.shape[2] // m  # (downblur's output shape: the reference's swapped Resize)
            noise = NoisePrefetcher([min(self.batch_size, n - i) for i in range(0, n, self.batch_size)], c, h, w, 2, 10)
        try:
            for i in range(0, n, self.batch_size):
                x, y = downblur(self.hr[order[i:i + self.batch_size]], self.magnification_factor, self.blur_radius)
                if noise is not None:
                    add_noise_clip_(x, noise.next())
                yield x, y
        finally:
            if noise is not None:
                noise.close()

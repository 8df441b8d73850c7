"""Drop-in `Diffusion` (schedules, q-sample, ancestral sampler, training loop, snapshots),
`launch(args)` and the CLI of reference train_diffusion_superres.py, with the arithmetic on
the gfx950 kernels of include/drs_hip.h.

Kept verbatim from the reference: constructor signature and attributes (:79-126), method names,
the snapshot file format {"MODEL_STATE", "EPOCHS_RUN"} (:257-308) and the 21 CLI flags (:703-724).
Deliberately different (documented in DESIGN.md):
  * `from_alpha_hat_to_beta` is vectorised (same fp32 operations, bit-identical values) instead of a
    T-iteration Python loop (:143-148);
  * `sample` computes the LR-conditioning branch once per chain and updates x with one fused kernel;
    the per-step timestep tensor lives on the device instead of being rebuilt on the host (:235);
  * multi-GPU uses one flat-buffer RCCL all-reduce per step (`dist.allreduce_gradients`) instead of
    DistributedDataParallel(find_unused_parameters=True) (:658).
"""
import copy
import os
import sys

import torch
import torch.nn as nn

from . import dist as drs_dist
from . import _lib, hip_ops
from .optim import FusedAdam
from .UNet_model_superres import EMA, Residual_Attention_UNet_superres

_DEGRADATIONS = ("downblur", "bsrgan", "downblurnoise")



CHAIN_CHECK_EVERY = 128  # reverse steps between two reads of the kernels' fault word inside a sampling chain


def run_reverse_chain(engine, x, noise_steps, step, frames=None, every=CHAIN_CHECK_EVERY):
    """The reverse loop of `Diffusion.sample` (reference :234-251): `step(i)` performs reverse step i in place on x, for i =
    noise_steps - 1 .. 1.  Every `every` steps (and at the end) the fault word of the wave-specialised kernels is read (one
    4-byte copy + a stream synchronisation: ~0.1 ms per 128 steps of ~1.2 ms each).  A protocol fault raises.  DRS_ERR_RANGE - an
    activation left the range of the FL arithmetic's fp16 main operand (csrc/conv_mfma_fl.hip; chains of UNTRAINED weights do
    that, their amplitude grows without bound) - has already switched the plan to the split-bf16 kernels: the chain goes back
    to its last checkpoint (x as of the last clean check) and continues from there."""
    i = noise_steps - 1
    ckpt_i, ckpt_x, ckpt_frames, since = i, x.clone(), 0, 0
    while i >= 1:
        step(i)
        i -= 1
        since += 1
        if since >= every or i == 0:
            since = 0
            try:
                engine.check_faults()
            except _lib.RangeFault as e:
                print(f"[drs] {e}\n[drs] resuming the chain at step {ckpt_i} on the split-bf16 kernels", file=sys.stderr)
                x.copy_(ckpt_x)
                i = ckpt_i
                if frames is not None:
                    del frames[ckpt_frames:]
                continue
            ckpt_i = i
            ckpt_x.copy_(x)
            ckpt_frames = len(frames) if frames is not None else 0
    return x


class Diffusion:
    def __init__(self, noise_schedule: str, model: nn.Module, snapshot_path: str, noise_steps=1000, beta_start=1e-4,
                 beta_end=0.02, device="cuda", magnification_factor=4, image_size=224, model_name="superres",
                 Degradation_type="BSRGAN", multiple_gpus=False, ema_smoothing=False):
        self.noise_steps = noise_steps
        self.beta_start = beta_start
        self.beta_end = beta_end
        self.image_size = image_size
        self.model_name = model_name
        self.magnification_factor = magnification_factor
        self.device = device
        self.snapshot_path = snapshot_path
        self.Degradation_type = Degradation_type
        self.multiple_gpus = multiple_gpus
        self.ema_smoothing = ema_smoothing
        self.model = model.to(self.device)
        self.epochs_run = 0
        if os.path.exists(snapshot_path):
            print("Loading snapshot")
            self._load_snapshot()

        self.noise_schedule = noise_schedule
        if noise_schedule == "linear":
            self.beta = self.prepare_noise_schedule().to(self.device)
            self.alpha = 1.0 - self.beta
            self.alpha_hat = torch.cumprod(self.alpha, dim=0)
        elif noise_schedule == "cosine":
            self.alpha_hat = self.prepare_noise_schedule().to(self.device)
            self.beta = self.from_alpha_hat_to_beta()
            self.alpha = 1.0 - self.beta
        # any other name leaves the schedule unset, like the reference (:117-126)

    # -- schedules (reference :128-169) ---------------------------------------------------------
    def prepare_noise_schedule(self):
        if self.noise_schedule == "linear":
            return torch.linspace(self.beta_start, self.beta_end, self.noise_steps)
        elif self.noise_schedule == "cosine":
            steps = torch.arange(self.noise_steps) / self.noise_steps
            f_t = torch.cos(((steps + 0.008) / (1 + 0.008)) * torch.pi / 2) ** 2
            return f_t / f_t[0]

    def from_alpha_hat_to_beta(self):
        ah = self.alpha_hat
        beta = torch.empty_like(ah)
        beta[0] = 1 - ah[0]
        beta[1:] = 1 - ah[1:] / ah[:-1]
        return beta

    # -- forward process (reference :171-205) ---------------------------------------------------
    def noise_images(self, x, t):
        epsilon = torch.randn_like(x, dtype=torch.float32)
        return hip_ops.noise_images(x, epsilon, t, self.alpha_hat), epsilon

    def sample_timesteps(self, n):
        return torch.randint(low=1, high=self.noise_steps, size=(n,))

    # -- reverse process (reference :207-255) ---------------------------------------------------
    def sample(self, n, model, lr_img, input_channels=3, generate_video=False, noise_source=None):
        """`noise_source(i, shape)`, when given, supplies x_T (i == noise_steps) and the per-step noise z_i
        instead of torch.randn — used to drive this sampler and the oracle with identical noise."""
        if self.Degradation_type.lower() not in _DEGRADATIONS:
            raise ValueError("The degradation type must be either BSRGAN or DownBlur")
        if lr_img.dim() == 4:
            # extension used by the aggregation tiler: one LR image per chain, (n, C, h, w); the reference only takes a
            # single (C, h, w) image broadcast over the n chains (:224)
            if lr_img.shape[0] != n:
                raise RuntimeError(f"sample: a batch of {lr_img.shape[0]} LR images for n={n} chains")
            lr_img = lr_img.to(self.device).contiguous()
        else:
            lr_img = lr_img.to(self.device).unsqueeze(0).contiguous()
        frames = []
        net = model.module if hasattr(model, "module") and not hasattr(model, "hip_engine") else model
        model.eval()
        engine = net.hip_engine()
        shape = (n, input_channels, self.image_size, self.image_size)
        with torch.no_grad():
            if noise_source is not None:
                x = noise_source(self.noise_steps, shape).to(self.device)
            else:
                x = torch.randn(shape).to(self.device)  # CPU generator, like the reference (:230)
            x = x.contiguous()
            t_rows = hip_ops.timestep_table(self.noise_steps, n, x.device)
            state = {"first": True}

            def step(i):
                predicted_noise = engine.forward(x, t_rows[i], lr_img, self.magnification_factor, reuse_cond=not state["first"],
                                                 check_weights=state["first"])
                state["first"] = False
                if i > 1:
                    noise = noise_source(i, shape).to(x.device) if noise_source is not None else torch.randn_like(x)
                else:
                    noise = None  # reference adds zeros at the last step (:248)
                hip_ops.sampler_step_(x, predicted_noise, noise, i, self.alpha, self.alpha_hat, self.beta)
                if generate_video:
                    frames.append(x.clone())
            # (the fault word of the wave-specialised kernels - a protocol fault reports itself through it instead of a trap,
            #  csrc/sp_sync.h - is read inside the loop, every CHAIN_CHECK_EVERY steps and at the end)
            run_reverse_chain(engine, x, self.noise_steps, step, frames if generate_video else None)
        if generate_video:
            from .video import video_maker  # optional dependency (cv2), same call as reference :253
            video_maker(frames, os.path.join(os.getcwd(), "models_run", self.model_name, "results",
                                             "video_denoising.mp4"), 100)
        model.train()  # reference side effect (:254)
        return x

    # -- snapshots (reference :257-308) -----------------------------------------------------------
    def _save_snapshot(self, epoch, model):
        net = model.module if self.multiple_gpus and hasattr(model, "module") else model
        snapshot = {"MODEL_STATE": net.state_dict(), "EPOCHS_RUN": epoch}
        torch.save(snapshot, self.snapshot_path)
        print(f"Epoch {epoch} | Training snapshot saved at {self.snapshot_path}")

    def _load_snapshot(self):
        if self.multiple_gpus:
            from collections import OrderedDict
            snapshot = torch.load(self.snapshot_path, map_location="cpu")
            state = OrderedDict((k.replace("module.", ""), v) for k, v in snapshot["MODEL_STATE"].items())
            net = self.model.module if hasattr(self.model, "module") else self.model
            net.load_state_dict(state)
            net.to(self.device)
        else:
            snapshot = torch.load(self.snapshot_path, map_location=self.device)
            self.model.load_state_dict(snapshot["MODEL_STATE"])
        self.epochs_run = snapshot["EPOCHS_RUN"]
        print(f"Resuming training from snapshot at Epoch {self.epochs_run}")

    def early_stopping(self, patience, epochs_without_improving):
        if epochs_without_improving >= patience:
            print("Early stopping! Training stopped")
            return True

    # -- training loop (reference :319-511) -------------------------------------------------------
    @staticmethod
    def _loss_function(loss):
        if loss == "MSE":
            return nn.MSELoss()
        if loss == "MAE":
            return nn.L1Loss()
        if loss == "Huber":
            return nn.HuberLoss()
        if loss == "MSE+Perceptual_noise":
            raise NotImplementedError("MSE+Perceptual_noise needs torchvision VGG19 weights (reference :25-63), which "
                                      "is outside the denoising hot path")
        raise ValueError("The Loss must be either MSE or MAE or Huber or MSE+Perceptual_noise")

    def _is_rank0(self):
        return (not self.multiple_gpus) or self.device == 0 or drs_dist.rank() == 0

    def _predict(self, net, x_t, t, cond):
        """The model call of the loop bodies (:388, :474); the SAR / generation subclasses override it."""
        return net(x_t, t, cond, self.magnification_factor)

    def _split_batch(self, batch):
        """(conditioning, clean image) of one loader item: (lr_img, hr_img) here (:379)."""
        return batch[0].to(self.device), batch[1].to(self.device)

    def train_step(self, model, optimizer, loss_function, lr_img, hr_img, ema=None, ema_model=None):
        """Loop body of reference :379-396 (called `train_step` in BASELINE.json's north_star)."""
        lr_img, hr_img = self._split_batch((lr_img, hr_img))
        # same CPU-generator draw as the reference (:384); pinned + non_blocking so that the copy does not make the host
        # wait for the previous step's kernels (a pageable .to(device) is a full synchronisation point)
        t = self.sample_timesteps(hr_img.shape[0])
        t = (t.pin_memory() if not t.is_cuda else t).to(self.device, non_blocking=True)
        x_t, noise = self.noise_images(hr_img, t)
        optimizer.zero_grad()
        predicted_noise = self._predict(model, x_t, t, self._train_cond(lr_img))
        train_loss = loss_function(predicted_noise, noise)
        train_loss.backward()
        if self.multiple_gpus:
            # ONE in-place all-reduce of the backward's flat gradient buffer on RCCL's stream, overlapped with the
            # optimizer's host-side table build; the update kernel is enqueued behind it
            pending = drs_dist.allreduce_gradients(model, async_op=True)
            if isinstance(optimizer, FusedAdam):
                optimizer.step(grad_ready=pending.wait)
            else:
                pending.wait()
                optimizer.step()
        else:
            optimizer.step()
        if ema is not None:
            ema.step_ema(ema_model, model)
        return train_loss

    def _train_cond(self, cond):
        """Conditioning actually passed to the model in a training / validation step (the generation subclass drops
        the label 10% of the time, like its reference loop)."""
        return cond

    def train(self, lr, epochs, check_preds_epoch, train_loader, val_loader, patience, loss, verbose):
        model = self.model
        optimizer = FusedAdam(model.parameters(), lr=lr)  # torch.optim.Adam's math, one launch (optim.py)
        ema = ema_model = None
        if self.ema_smoothing:
            ema = EMA(beta=0.995)
            ema_model = copy.deepcopy(model).eval().requires_grad_(False)
        loss_function = self._loss_function(loss)
        epochs_without_improving = 0
        best_loss = float("inf")
        saved = ema_model if self.ema_smoothing else model

        for epoch in range(self.epochs_run, epochs):
            if self.multiple_gpus and hasattr(train_loader, "sampler") and hasattr(train_loader.sampler, "set_epoch"):
                train_loader.sampler.set_epoch(epoch)
            running_train_loss = torch.zeros((), device=self.device)  # accumulated on device: no per-step sync
            model.train()
            for lr_img, hr_img in train_loader:
                running_train_loss += self.train_step(model, optimizer, loss_function, lr_img, hr_img, ema,
                                                      ema_model).detach()
            running_train_loss = running_train_loss.item() / max(len(train_loader), 1)
            # (the .item() above is the epoch's synchronisation point: a wave of the wave-specialised kernels that gave up on a
            #  counter during this epoch's forwards or backwards - csrc/sp_sync.h - is reported here, not trained on)
            net = model.module if hasattr(model, "module") and not hasattr(model, "hip_engine") else model
            if hasattr(net, "hip_engine"):
                net.hip_engine().check_faults()
            print(f"Epoch {epoch}: Running Train ({loss}) {running_train_loss}")

            if self._is_rank0() and epoch % check_preds_epoch == 0 and val_loader is None:
                self._save_snapshot(epoch, saved)

            if val_loader is not None:
                running_val_loss = torch.zeros((), device=self.device)
                with torch.no_grad():
                    model.eval()
                    for batch in val_loader:
                        lr_img, hr_img = self._split_batch(batch)
                        t = self.sample_timesteps(hr_img.shape[0]).to(self.device)
                        x_t, noise = self.noise_images(hr_img, t)
                        net = ema_model if self.ema_smoothing else model
                        running_val_loss += loss_function(self._predict(net, x_t, t, self._train_cond(lr_img)), noise)
                running_val_loss = running_val_loss.item() / max(len(val_loader), 1)
                if self.multiple_gpus:
                    # the reference decides per rank on its own validation shard (:492-510, quirk Q9); with a collective in
                    # every training step the ranks must leave the loop together: one mean over ranks, same branch everywhere
                    running_val_loss = drs_dist.allreduce_mean_scalar(running_val_loss)
                print(f"Epoch {epoch}: Running Val loss ({loss}){running_val_loss}")
                if running_val_loss < best_loss:
                    best_loss = running_val_loss
                    epochs_without_improving = 0
                    if self._is_rank0():
                        self._save_snapshot(epoch, saved)
                else:
                    epochs_without_improving += 1
                if self.early_stopping(patience, epochs_without_improving):
                    break
            print("Epochs without improving: ", epochs_without_improving)


class SyntheticSuperresDataset(torch.utils.data.Dataset):
    """Seeded Gaussian/uniform (lr, hr) patches with the shapes `get_data_superres` yields (reference
    utils.py:93-166): the reference's image-folder datasets (PIL / torchvision) are outside the hot path and
    BASELINE.json's configs are all "synthetic"."""

    def __init__(self, length, channels, image_size, magnification_factor, seed=0):
        from . import synthetic
        s = image_size // magnification_factor
        self.hr = synthetic.tensor_uniform("synthetic.hr", (length, channels, image_size, image_size), seed)
        self.lr = synthetic.tensor_uniform("synthetic.lr", (length, channels, s, s), seed)

    def __len__(self):
        return self.hr.shape[0]

    def __getitem__(self, i):
        return self.lr[i], self.hr[i]


def launch(args):
    """Reference launch (:513-693) for the hot path: model + Diffusion + train + final sampling.  `--dataset_path` is the
    reference's image folder (`<path>/train_original`, `<path>/val_original`, :597-598: decoded once with Pillow into a uint8
    cache on the device, DownBlur / DownBlurNoise per batch on the device) or `synthetic[:N]` / `synthetic_u8[:N]` (seeded
    patches; float pairs / the same device feed)."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler

    if args.Degradation_type.lower() not in _DEGRADATIONS:
        raise ValueError("The degradation type must be either BSRGAN or DownBlur or DownBlurNoise")
    if args.Degradation_type.lower() == "downblur" and args.image_size % args.magnification_factor != 0:
        raise ValueError("The image size must be a multiple of the magnification factor")
    if args.UNet_type.lower() != "residual attention unet":
        raise ValueError("The UNet type must be Residual Attention UNet or Residual MultiHead Attention UNet or "
                         "Residual Visual MultiHeadAttention UNet superres")
    os.makedirs(args.snapshot_folder_path, exist_ok=True)
    os.makedirs(os.path.join(os.curdir, "models_run", args.model_name, "results"), exist_ok=True)

    if args.multiple_gpus:
        print("Using multiple GPUs")
        drs_dist.init_process_group()  # RCCL ("nccl" backend on ROCm), env:// rendezvous like reference :586
        gpu_id = int(os.environ["LOCAL_RANK"])
        torch.cuda.set_device(gpu_id)
        device = gpu_id
    else:
        print("Using single GPU")
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device visible: this implementation has no CPU path")
        device = torch.device("cuda")

    spec = str(args.dataset_path or "")
    ch = args.inp_out_channels
    r, wsz = (drs_dist.rank(), drs_dist.world_size()) if args.multiple_gpus else (0, 1)
    device_feed = not spec.startswith("synthetic") or spec.startswith("synthetic_u8")
    if device_feed:
        # the reference's DownBlur feed (utils.get_data_superres: bicubic down-sampling + Gaussian blur per item with
        # Pillow on the host) from a uint8 HR cache on the device, bit-exact (degradation.py); rank r owns every
        # world-th image like DistributedSampler
        from .degradation import DeviceSuperresFeed, load_image_folder_u8
        if args.Degradation_type.lower() not in ("downblur", "downblurnoise"):
            raise NotImplementedError("the on-device feed implements Degradation_type=DownBlur and DownBlurNoise "
                                      "(the BSRGAN degradation is outside the hot path)")
        gauss_noise = args.Degradation_type.lower() == "downblurnoise"  # (reference :612-616: Gauss_noise=True)
        radius = args.Blur_radius if args.Blur_radius == "random" else float(args.Blur_radius)

        def make_feed(u8):
            return DeviceSuperresFeed(u8.to(device), args.magnification_factor, radius, args.batch_size, shuffle=True,
                                      Gauss_noise=gauss_noise)
    if not spec.startswith("synthetic"):
        # an image folder, laid out as the reference expects it (:597-598): <dataset_path>/train_original, /val_original.
        # Decoded once with Pillow into the uint8 cache (this rank's shard only); resize / blur / noise per batch on the device
        if not os.path.isdir(os.path.join(spec, "train_original")) or not os.path.isdir(os.path.join(spec, "val_original")):
            raise FileNotFoundError(f"--dataset_path {spec!r}: expected the folders train_original/ and val_original/ "
                                    "(or synthetic[:N] / synthetic_u8[:N])")
        train_loader = make_feed(load_image_folder_u8(os.path.join(spec, "train_original"), args.image_size, r, wsz))
        val_loader = make_feed(load_image_folder_u8(os.path.join(spec, "val_original"), args.image_size, r, wsz))
        for what, feed in (("train_original", train_loader), ("val_original", val_loader)):
            if feed.hr.shape[1] != ch:
                raise ValueError(f"the images of {what} have {feed.hr.shape[1]} channels, --inp_out_channels is {ch}")
        # the final sampling conditions on train_dataset[0..4] of the WHOLE sorted folder (reference :678-680), whatever this
        # rank's shard holds; with Gauss_noise the dataset item carries its noise (utils.py:126-138)
        first = load_image_folder_u8(os.path.join(spec, "train_original"), args.image_size, limit=5).to(device)
        from .degradation import add_reference_noise, downblur
        final_x = downblur(first, args.magnification_factor, train_loader.blur_radius)[0]
        if gauss_noise:
            add_reference_noise(final_x, noise_level1=2, noise_level2=10)
        final_lr = [final_x[i] for i in range(final_x.shape[0])]
    else:
        length = int(spec.split(":")[1]) if ":" in spec else 4 * args.batch_size
        train_dataset = SyntheticSuperresDataset(length, ch, args.image_size, args.magnification_factor, seed=1)
        val_dataset = SyntheticSuperresDataset(max(length // 4, 1), ch, args.image_size, args.magnification_factor, seed=2)
        final_lr = [train_dataset[i][0] for i in range(min(5, len(train_dataset)))]
        if device_feed:
            def feed(ds):
                # equal shard sizes on every rank (DistributedSampler pads; here the remainder is dropped): ranks must run
                # the same number of steps, or the per-step all-reduce of the longer shard never completes
                per_rank = len(ds) // wsz
                if per_rank == 0:
                    raise ValueError(f"dataset of {len(ds)} images cannot be sharded over {wsz} ranks")
                return make_feed((ds.hr[r::wsz][:per_rank] * 255).round().clamp(0, 255).to(torch.uint8))
            train_loader, val_loader = feed(train_dataset), feed(val_dataset)
        elif args.multiple_gpus:
            train_loader = DataLoader(train_dataset, batch_size=args.batch_size, shuffle=False,
                                      sampler=DistributedSampler(train_dataset))
            val_loader = DataLoader(val_dataset, batch_size=args.batch_size, shuffle=False,
                                    sampler=DistributedSampler(val_dataset))
        else:
            train_loader = DataLoader(train_dataset, batch_size=args.batch_size, shuffle=True)
            val_loader = DataLoader(val_dataset, batch_size=args.batch_size, shuffle=True)

    print("Using Residual Attention UNet")
    model = Residual_Attention_UNet_superres(ch, ch, device).to(device)
    print("Num params: ", sum(p.numel() for p in model.parameters()))
    if args.multiple_gpus:
        drs_dist.broadcast_module(model)  # what the DDP constructor does in the reference (:658)

    diffusion = Diffusion(noise_schedule=args.noise_schedule, model=model,
                          snapshot_path=os.path.join(args.snapshot_folder_path, args.snapshot_name),
                          noise_steps=args.noise_steps, beta_start=1e-4, beta_end=0.02,
                          magnification_factor=args.magnification_factor, device=device, image_size=args.image_size,
                          model_name=args.model_name, Degradation_type=args.Degradation_type,
                          multiple_gpus=args.multiple_gpus, ema_smoothing=args.ema_smoothing)
    diffusion.train(lr=args.lr, epochs=args.epochs, check_preds_epoch=args.check_preds_epoch,
                    train_loader=train_loader, val_loader=val_loader, patience=args.patience, loss=args.loss,
                    verbose=True)
    if args.multiple_gpus:
        drs_dist.destroy_process_group()
    if r != 0:
        return  # one rank samples and writes models_run/<name>/results/superres_results.pt (every rank holds the same weights)
    outs = [diffusion.sample(n=1, model=model, lr_img=lr_i, input_channels=ch, generate_video=args.generate_video)
            for lr_i in final_lr]
    torch.save(torch.cat(outs).cpu(), os.path.join(os.getcwd(), "models_run", args.model_name, "results",
                                                  "superres_results.pt"))


def build_arg_parser():
    """The reference's flags, verbatim (:703-724)."""
    import argparse

    def str2bool(v):
        return v.lower() in ("yes", "true", "t", "1")

    p = argparse.ArgumentParser(description=" ")
    p.add_argument("--epochs", type=int, default=501)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--image_size", type=int)
    p.add_argument("--lr", type=float, default=3e-4)
    p.add_argument("--check_preds_epoch", type=int, default=20)
    p.add_argument("--noise_schedule", type=str, default="cosine")
    p.add_argument("--snapshot_name", type=str, default="snapshot.pt")
    p.add_argument("--model_name", type=str)
    p.add_argument("--noise_steps", type=int, default=200)
    p.add_argument("--patience", type=int, default=10)
    p.add_argument("--dataset_path", type=str, default=None)
    p.add_argument("--inp_out_channels", type=int, default=3)
    p.add_argument("--generate_video", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--loss", type=str)
    p.add_argument("--magnification_factor", type=int)
    p.add_argument("--UNet_type", type=str, default="Residual Attention UNet")
    p.add_argument("--Degradation_type", type=str, default="DownBlur")
    p.add_argument("--num_crops", type=int, default=1)
    p.add_argument("--multiple_gpus", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--ema_smoothing", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--Blur_radius", type=str, default="random")
    return p


def main(argv=None):
    args = build_arg_parser().parse_args(argv)
    args.snapshot_folder_path = os.path.join(os.curdir, "models_run", args.model_name, "weights")
    launch(args)


if __name__ == "__main__":
    main()

"""Drop-in `Diffusion` / `launch` / CLI of reference train_diffusion_SAR_TO_NDVI.py on the gfx950 kernels.

Same arithmetic as the super-resolution Diffusion (the reference's two files differ only in the model call,
`model(x_t, t, SAR_img)`, and in the absence of magnification / degradation arguments), so this class reuses the
schedules, q-sample, snapshots and training loop of `train_diffusion_superres.Diffusion` and overrides what differs:
constructor (:80-123), `sample` (:204-249) and the model call of the loop bodies (:373-388, :455-470).
"""
import os

import torch
import torch.nn as nn

from . import dist as drs_dist
from . import hip_ops
from .train_diffusion_superres import Diffusion as _SuperresDiffusion, run_reverse_chain
from .UNet_model_SAR_TO_NDVI import Residual_Attention_UNet_SAR_TO_NDVI


class Diffusion(_SuperresDiffusion):
    def __init__(self, noise_schedule: str, model: nn.Module, snapshot_path: str, noise_steps=1000, beta_start=1e-4,
                 beta_end=0.02, device="cuda", image_size=224, model_name="SAR_TO_NDVI", multiple_gpus=False,
                 ema_smoothing=False):
        super().__init__(noise_schedule, model, snapshot_path, noise_steps=noise_steps, beta_start=beta_start,
                         beta_end=beta_end, device=device, magnification_factor=1, image_size=image_size,
                         model_name=model_name, Degradation_type="DownBlur", multiple_gpus=multiple_gpus,
                         ema_smoothing=ema_smoothing)
        del self.magnification_factor, self.Degradation_type  # attributes the reference's SAR Diffusion does not have

    def _predict(self, net, x_t, t, cond):
        return net(x_t, t, cond)

    def sample(self, n, model, SAR_img, NDVI_channels=1, generate_video=False, noise_source=None):
        """Reference :204-249.  One (SAR_channels, S, S) image conditions all n chains; its encoder branch is computed
        once per chain instead of once per step.  `noise_source(i, shape)` as in the super-resolution sampler."""
        SAR_img = SAR_img.to(self.device).unsqueeze(0).contiguous()
        frames = []
        net = model.module if hasattr(model, "module") and not hasattr(model, "hip_engine") else model
        model.eval()
        engine = net.hip_engine()
        shape = (n, NDVI_channels, self.image_size, self.image_size)
        with torch.no_grad():
            x = (noise_source(self.noise_steps, shape) if noise_source is not None else torch.randn(shape)).to(self.device)
            x = x.contiguous()
            t_rows = hip_ops.timestep_table(self.noise_steps, n, x.device)
            state = {"first": True}

            def step(i):
                predicted_noise = engine.forward(x, t_rows[i], SAR_img, 1, reuse_cond=not state["first"], check_weights=state["first"])
                state["first"] = False
                if i > 1:
                    noise = noise_source(i, shape).to(x.device) if noise_source is not None else torch.randn_like(x)
                else:
                    noise = None
                hip_ops.sampler_step_(x, predicted_noise, noise, i, self.alpha, self.alpha_hat, self.beta)
                if generate_video:
                    frames.append(x.clone())
            run_reverse_chain(engine, x, self.noise_steps, step, frames if generate_video else None)  # (reads the kernels' fault word)
        if generate_video:
            from .video import video_maker
            video_maker(frames, os.path.join(os.getcwd(), "models_run", self.model_name, "results",
                                             "video_denoising.mp4"), 100)
        model.train()
        return x


class SyntheticSarNdviDataset(torch.utils.data.Dataset):
    """Seeded (SAR_img, NDVI_img) pairs with the shapes `get_data_SAR_TO_NDVI` yields (reference utils.py); the
    image-folder dataset itself is outside the hot path."""

    def __init__(self, length, sar_channels, ndvi_channels, image_size, seed=0):
        from . import synthetic
        self.sar = synthetic.tensor_uniform("synthetic.sar", (length, sar_channels, image_size, image_size), seed)
        self.ndvi = synthetic.tensor_uniform("synthetic.ndvi", (length, ndvi_channels, image_size, image_size), seed)

    def __len__(self):
        return self.sar.shape[0]

    def __getitem__(self, i):
        return self.sar[i], self.ndvi[i]


def launch(args):
    """Reference launch (:505-633) for the hot path: model + Diffusion + train + final sampling on seeded data."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler

    if args.UNet_type.lower() != "residual attention unet":
        raise ValueError("The UNet type must be Residual Attention UNet")
    os.makedirs(args.snapshot_folder_path, exist_ok=True)
    os.makedirs(os.path.join(os.curdir, "models_run", args.model_name, "results"), exist_ok=True)
    if args.multiple_gpus:
        drs_dist.init_process_group()
        device = int(os.environ["LOCAL_RANK"])
        torch.cuda.set_device(device)
    else:
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device visible: this implementation has no CPU path")
        device = torch.device("cuda")
    spec = str(args.dataset_path or "")
    if not spec.startswith("synthetic"):
        raise NotImplementedError("image-folder datasets (reference utils.get_data_SAR_TO_NDVI) are outside the hot "
                                  "path; use --dataset_path synthetic[:N]")
    length = int(spec.split(":")[1]) if ":" in spec else 4 * args.batch_size
    train_dataset = SyntheticSarNdviDataset(length, args.SAR_channels, args.NDVI_channels, args.image_size, seed=1)
    val_dataset = SyntheticSarNdviDataset(max(length // 4, 1), args.SAR_channels, args.NDVI_channels, args.image_size, seed=2)
    if args.multiple_gpus:
        train_loader = DataLoader(train_dataset, batch_size=args.batch_size, sampler=DistributedSampler(train_dataset))
        val_loader = DataLoader(val_dataset, batch_size=args.batch_size, sampler=DistributedSampler(val_dataset))
    else:
        train_loader = DataLoader(train_dataset, batch_size=args.batch_size, shuffle=True)
        val_loader = DataLoader(val_dataset, batch_size=args.batch_size, shuffle=True)
    model = Residual_Attention_UNet_SAR_TO_NDVI(args.SAR_channels, args.NDVI_channels, device).to(device)
    print("Num params: ", sum(p.numel() for p in model.parameters()))
    if args.multiple_gpus:
        drs_dist.broadcast_module(model)
    diffusion = Diffusion(noise_schedule=args.noise_schedule, model=model,
                          snapshot_path=os.path.join(args.snapshot_folder_path, args.snapshot_name),
                          noise_steps=args.noise_steps, beta_start=1e-4, beta_end=0.02, device=device,
                          image_size=args.image_size, model_name=args.model_name, multiple_gpus=args.multiple_gpus,
                          ema_smoothing=args.ema_smoothing)
    diffusion.train(lr=args.lr, epochs=args.epochs, check_preds_epoch=args.check_preds_epoch,
                    train_loader=train_loader, val_loader=val_loader, patience=args.patience, loss=args.loss,
                    verbose=True)
    if args.multiple_gpus:
        drs_dist.destroy_process_group()
    outs = [diffusion.sample(n=1, model=model, SAR_img=train_dataset[i][0], NDVI_channels=args.NDVI_channels,
                             generate_video=args.generate_video) for i in range(min(5, len(train_dataset)))]
    torch.save(torch.cat(outs).cpu(), os.path.join(os.getcwd(), "models_run", args.model_name, "results",
                                                  "SAR_TO_NDVI_results.pt"))


def build_arg_parser():
    """The reference's flags, verbatim (:646-663)."""
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
    p.add_argument("--SAR_channels", type=int, default=2)
    p.add_argument("--NDVI_channels", type=int, default=1)
    p.add_argument("--generate_video", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--loss", type=str)
    p.add_argument("--UNet_type", type=str, default="Residual Attention UNet")
    p.add_argument("--multiple_gpus", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--ema_smoothing", type=str2bool, nargs="?", const=True, default=False)
    return p


def main(argv=None):
    args = build_arg_parser().parse_args(argv)
    args.snapshot_folder_path = os.path.join(os.curdir, "models_run", args.model_name, "weights")
    launch(args)


if __name__ == "__main__":
    main()

"""Drop-in `Diffusion` / `launch` / CLI of reference generate_new_imgs/train_diffusion_generation.py on the gfx950
kernels: class-conditional DDPM with classifier-free guidance.

Differences from the super-resolution Diffusion it derives from: no conditioning image, the loader yields
(img, label), 10% of the training steps drop the label (:393-394), and `sample` runs a conditional and (for
cfg_scale > 0) an unconditional forward per step combined with torch.lerp (:236-239).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import dist as drs_dist
from .. import hip_ops
from ..train_diffusion_superres import Diffusion as _SuperresDiffusion, run_reverse_chain
from .UNet_model_generation import Residual_Attention_UNet_generation


class Diffusion(_SuperresDiffusion):
    def __init__(self, noise_schedule: str, model: nn.Module, snapshot_path: str, noise_steps=1000, beta_start=1e-4,
                 beta_end=0.02, device="cuda", image_size=224, model_name="generation", multiple_gpus=False,
                 ema_smoothing=False):
        super().__init__(noise_schedule, model, snapshot_path, noise_steps=noise_steps, beta_start=beta_start,
                         beta_end=beta_end, device=device, magnification_factor=1, image_size=image_size,
                         model_name=model_name, Degradation_type="DownBlur", multiple_gpus=multiple_gpus,
                         ema_smoothing=ema_smoothing)
        del self.magnification_factor, self.Degradation_type

    def _split_batch(self, batch):
        """Loader items are (img, label) (:384-386); returned as (conditioning = label, clean image)."""
        return batch[1].to(self.device), batch[0].to(self.device)

    def _train_cond(self, label):
        # 10% of the training and validation steps run unconditionally (:393-394, :466-467); same host RNG call
        return None if np.random.random() < 0.1 else label

    def _predict(self, net, x_t, t, cond):
        return net(x_t, t, cond)

    def sample(self, n, model, target_class=None, cfg_scale=3, input_channels=3, generate_video=False,
               noise_source=None):
        """Reference :206-259."""
        frames = []
        net = model.module if hasattr(model, "module") and not hasattr(model, "hip_engine") else model
        model.eval()
        engine = net.hip_engine()
        shape = (n, input_channels, self.image_size, self.image_size)
        if target_class is not None:
            ncls = getattr(net, "num_classes", None)
            if not target_class.is_cuda and ncls and target_class.numel() and int(target_class.max()) >= int(ncls):
                raise IndexError(f"target_class {int(target_class.max())} out of range for num_classes={ncls}")
            target_class = target_class.to(self.device)
        with torch.no_grad():
            x = (noise_source(self.noise_steps, shape) if noise_source is not None else torch.randn(shape)).to(self.device)
            x = x.contiguous()
            guided = cfg_scale > 0 and target_class is not None
            if guided:
                # conditional and unconditional predictions as ONE 2n batch (label -1 = no embedding for that row) and
                # torch.lerp folded into the update kernel: half the launches of the reference's two forwards (:236-239)
                labels2 = torch.cat([target_class.to(torch.int64).expand(n) if target_class.numel() == 1
                                     else target_class.to(torch.int64), torch.full((n,), -1, dtype=torch.int64,
                                                                                   device=x.device)]).contiguous()
            t_rows = hip_ops.timestep_table(self.noise_steps, 2 * n, x.device)  # (row i: step i; the unguided forward takes n of the 2n)
            state = {"first": True}

            def step(i):
                if i > 1:
                    noise = noise_source(i, shape).to(x.device) if noise_source is not None else torch.randn_like(x)
                else:
                    noise = None
                if guided:
                    eps2 = engine.forward(x.repeat(2, 1, 1, 1), t_rows[i], None, 1, labels=labels2, check_weights=state["first"])
                    hip_ops.sampler_step_cfg_(x, eps2[:n], eps2[n:], cfg_scale, noise, i, self.alpha, self.alpha_hat,
                                              self.beta)
                else:
                    # cfg_scale > 0 without a class: lerp(u, u, w) == u, one forward is enough
                    predicted_noise = engine.forward(x, t_rows[i, :n], None, 1, labels=target_class, check_weights=state["first"])
                    hip_ops.sampler_step_(x, predicted_noise, noise, i, self.alpha, self.alpha_hat, self.beta)
                state["first"] = False
                if generate_video:
                    frames.append(x.clone())
            run_reverse_chain(engine, x, self.noise_steps, step, frames if generate_video else None)  # (reads the kernels' fault word)
        if generate_video:
            from ..video import video_maker
            video_maker(frames, os.path.join(os.getcwd(), "models_run", self.model_name, "results",
                                             "video_denoising.mp4"), 100)
        model.train()
        return x


class SyntheticClassDataset(torch.utils.data.Dataset):
    """Seeded (img, label) pairs shaped like torchvision ImageFolder items (reference :584-586 uses
    `train_loader.dataset.classes`)."""

    def __init__(self, length, channels, image_size, num_classes=10, seed=0):
        from .. import synthetic
        self.img = synthetic.tensor_uniform("synthetic.img", (length, channels, image_size, image_size), seed)
        self.label = synthetic.tensor_randint("synthetic.label", (length,), 0, num_classes, seed)
        self.classes = [str(i) for i in range(num_classes)]

    def __len__(self):
        return self.img.shape[0]

    def __getitem__(self, i):
        return self.img[i], self.label[i]


def launch(args):
    """Reference launch (:505-636) for the hot path on seeded data: `--dataset_path synthetic[:N[:classes]]`."""
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
        raise NotImplementedError("image-folder datasets (reference generate_new_imgs/utils.get_data) are outside "
                                  "the hot path; use --dataset_path synthetic[:N[:classes]]")
    parts = spec.split(":")
    length = int(parts[1]) if len(parts) > 1 else 4 * args.batch_size
    ncls = int(parts[2]) if len(parts) > 2 else 10
    ch = args.inp_out_channels
    train_dataset = SyntheticClassDataset(length, ch, args.image_size, ncls, seed=1)
    val_dataset = SyntheticClassDataset(max(length // 4, 1), ch, args.image_size, ncls, seed=2)
    if args.multiple_gpus:
        train_loader = DataLoader(train_dataset, batch_size=args.batch_size, sampler=DistributedSampler(train_dataset))
        val_loader = DataLoader(val_dataset, batch_size=args.batch_size, sampler=DistributedSampler(val_dataset))
    else:
        train_loader = DataLoader(train_dataset, batch_size=args.batch_size, shuffle=True)
        val_loader = DataLoader(val_dataset, batch_size=args.batch_size, shuffle=True)
    num_classes = len(train_loader.dataset.classes)
    model = Residual_Attention_UNet_generation(ch, ch, num_classes, device).to(device)
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
    outs = [diffusion.sample(n=5, model=model, target_class=torch.full((5,), i, dtype=torch.int64), cfg_scale=3,
                             input_channels=ch, generate_video=False) for i in range(min(num_classes, 3))]
    torch.save(torch.cat(outs).cpu(), os.path.join(os.getcwd(), "models_run", args.model_name, "results",
                                                  "generation_results.pt"))


def build_arg_parser():
    """The reference's flags, verbatim (:649-665)."""
    import argparse

    def str2bool(v):
        return v.lower() in ("yes", "true", "t", "1")

    p = argparse.ArgumentParser(description=" ")
    p.add_argument("--epochs", type=int, default=501)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--image_size", type=int, default=None)
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

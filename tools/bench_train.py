#!/usr/bin/env python3
"""Time the training-loop body (reference train_diffusion_superres.py:379-396) at the per-rank shape of BASELINE
configs[2]: batch 16, 128->256, cosine T=1500, MSE, Adam, optional EMA.  Single GPU; prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffusionremotesensing_amd import synthetic  # noqa: E402
from diffusionremotesensing_amd.UNet_model_superres import EMA, Residual_Attention_UNet_superres  # noqa: E402
from diffusionremotesensing_amd.train_diffusion_superres import Diffusion  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--image", type=int, default=256)
ap.add_argument("--train-impl", default=os.environ.get("DRS_TRAIN_IMPL", "mfma_f32"))
a = ap.parse_args()
dev = torch.device("cuda:0")
m = Residual_Attention_UNet_superres(3, 3, dev)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).train()
m.hip_engine().set_impl("mfma_bf16x3", train_impl=a.train_impl)
d = Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=1500, device=dev, magnification_factor=2,
              image_size=a.image, Degradation_type="DownBlur")
hr = synthetic.tensor_uniform("train.hr", (a.batch, 3, a.image, a.image)).to(dev)
lr = synthetic.tensor_uniform("train.lr", (a.batch, 3, a.image // 2, a.image // 2)).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
loss_fn = torch.nn.MSELoss()
losses = []
for k in range(a.warmup + a.steps):
    if k == a.warmup:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    losses.append(d.train_step(m, opt, loss_fn, lr, hr).detach())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"metric": "train_step_s", "value": dt, "batch": a.batch, "image": a.image, "train_impl": a.train_impl,
                  "steps_per_s": 1.0 / dt, "first_loss": losses[0].item(), "last_loss": losses[-1].item()}))

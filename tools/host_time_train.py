#!/usr/bin/env python3
"""Host-side cost of the training step: time to SUBMIT 10 steps (no synchronisation) vs time until the GPU is done, with a
cProfile listing of the submission path.  (This is how the per-name state_dict() rebuild - 14 ms per step - was found.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionremotesensing_amd import synthetic
from diffusionremotesensing_amd.optim import FusedAdam
from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
dev = torch.device("cuda:0")
m = Residual_Attention_UNet_superres(3, 3, dev)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).train()
d = Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=1500, device=dev, magnification_factor=2, image_size=256, Degradation_type="DownBlur")
hr = synthetic.tensor_uniform("train.hr", (16, 3, 256, 256)).to(dev)
lr = synthetic.tensor_uniform("train.lr", (16, 3, 128, 128)).to(dev)
opt = FusedAdam(m.parameters(), lr=1e-4)
loss_fn = torch.nn.MSELoss()
for _ in range(3): d.train_step(m, opt, loss_fn, lr, hr)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(10): d.train_step(m, opt, loss_fn, lr, hr)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"submit {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

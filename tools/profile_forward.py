#!/usr/bin/env python3
"""Run a few UNet denoise steps at BASELINE configs[1] for rocprofv3 (kernel trace / PMC passes).
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_forward.py --impl mfma_bf16x3 --steps 5
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffusionremotesensing_amd import hip_ops, synthetic  # noqa: E402
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres  # noqa: E402
from diffusionremotesensing_amd.train_diffusion_superres import Diffusion  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--impl", default="mfma_bf16x3")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--image", type=int, default=256)
ap.add_argument("--launch-log", default=None,
                help="write the plan's launch log of the LAST forward here (op <TAB> kernel per launch); that forward is "
                     "then the process's last GPU work, so a counter pass sees exactly these dispatches at its end")
a = ap.parse_args()
dev = torch.device("cuda:0")
m = Residual_Attention_UNet_superres(3, 3, dev)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
eng = m.hip_engine()
eng.set_impl(a.impl)
d = Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=1500, device=dev, magnification_factor=2,
              image_size=a.image, Degradation_type="DownBlur")
x = synthetic.tensor_normal("bench.x", (a.batch, 3, a.image, a.image)).to(dev)
lr = synthetic.tensor_uniform("bench.lr", (a.batch, 3, a.image // 2, a.image // 2)).to(dev)
t = torch.empty(a.batch, dtype=torch.int64, device=dev)
with torch.no_grad():
    for k in range(a.steps):
        i = 1499 - k
        t.fill_(i)
        eps = eng.forward(x, t, lr, 2, reuse_cond=k > 0, check_weights=k == 0)
        hip_ops.sampler_step_(x, eps, torch.randn_like(x), i, d.alpha, d.alpha_hat, d.beta)
    torch.cuda.synchronize()
    mean_abs = float(x.abs().mean())
    if a.launch_log:
        # the LAST GPU work of the process: one more forward with the plan's launch log on (serial schedule, cached
        # conditioning branch like every step of a chain but the first); nothing is launched after it
        t.fill_(1499 - a.steps)
        torch.cuda.synchronize()
        _, log = eng.logged_forward(x, t, lr, 2, reuse_cond=True, check_weights=False)
        torch.cuda.synchronize()
        with open(a.launch_log, "w") as f:
            for op, kernel in log:
                f.write(f"{op or '-'}\t{kernel}\n")
print("done", mean_abs)

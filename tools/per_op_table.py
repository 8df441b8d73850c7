#!/usr/bin/env python3
"""Per-op timing of one UNet forward at BASELINE configs[1] (HIP events around every op of the schedule,
drs_unet_profile_*): name, ms, algorithmic TFLOP/s and GB/s.  Usage: per_op_table.py [--impl mfma_bf16x3] [--batch 16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffusionremotesensing_amd import synthetic  # noqa: E402
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--impl", default="mfma_bf16x3")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--image", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
m = Residual_Attention_UNet_superres(3, 3, dev)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
eng = m.hip_engine()
eng.set_impl(a.impl)
x = synthetic.tensor_normal("bench.x", (a.batch, 3, a.image, a.image)).to(dev)
lr = synthetic.tensor_uniform("bench.lr", (a.batch, 3, a.image // 2, a.image // 2)).to(dev)
t = torch.full((a.batch,), 700, dtype=torch.int64, device=dev)
with torch.no_grad():
    rows = eng.profile_forward(x, t, lr, 2, iters=a.iters)
tot = sum(r[1] for r in rows)
print(f"{'op':34s} {'ms':>8s} {'%':>6s} {'TFLOP/s':>9s} {'GB/s':>8s}")
for name, ms, fl, by in rows:
    print(f"{name:34s} {ms:8.4f} {100*ms/tot:6.1f} {fl/ms/1e9 if ms else 0:9.1f} {by/ms/1e6 if ms else 0:8.0f}")
print(f"{'total':34s} {tot:8.4f}")

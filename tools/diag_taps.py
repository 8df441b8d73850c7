#!/usr/bin/env python3
"""Per-tap error of one eval forward (keep_intermediates) against the CPU oracle, in schedule order: finds the first
broken tensor after a kernel change.  Usage: diag_taps.py [--impl mfma_bf16x3] [--batch 2] [--image 32] [--keep 1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffusionremotesensing_amd import synthetic  # noqa: E402
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres  # noqa: E402
from oracle import unet_oracle as U  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--impl", default="mfma_bf16x3")
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--image", type=int, default=32)
ap.add_argument("--keep", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
m = Residual_Attention_UNet_superres(3, 3, dev)
sd = synthetic.seeded_state_dict(m.state_dict(), 0)
m.load_state_dict(sd)
m = m.to(dev).eval()
eng = m.hip_engine()
eng.set_impl(a.impl)
eng.keep_intermediates = bool(a.keep)
x = synthetic.tensor_normal("diag.x", (a.batch, 3, a.image, a.image))
lr = synthetic.tensor_uniform("diag.lr", (a.batch, 3, a.image // 2, a.image // 2))
t = synthetic.tensor_randint("diag.t", (a.batch,), 1, 1500)
taps = {}
with torch.no_grad():
    out = m(x.to(dev), t.to(dev), lr.to(dev), 2).cpu()
    want = U.unet_forward(sd, x, t, lr, 2, taps=taps)


def err(a_, b_):
    d = (a_.double() - b_.double())
    return (d.abs().max() / b_.abs().max().clamp_min(1e-30)).item(), (d.norm() / b_.double().norm().clamp_min(1e-30)).item()


names = eng.tensor_names()
if a.keep:
    for k, v in taps.items():
        if k in names:
            got = eng.read_tensor(k).cpu()
            e = err(got, v)
            print(f"{k:34s} max-rel {e[0]:.3e} rel-L2 {e[1]:.3e}  {'' if e[0] < 1e-3 else '<<<<'}")
e = err(out, want)
print(f"{'output':34s} max-rel {e[0]:.3e} rel-L2 {e[1]:.3e}")

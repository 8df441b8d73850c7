#!/usr/bin/env python3
"""Per-tap error of a KEEP_ALL forward on the trained-like fixture (tests/test_gpu_fl.py) for the current DRS_FL setting."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import golden_inputs, rel_errors
import test_gpu_fl as T
from diffusionremotesensing_amd import synthetic
from oracle import unet_oracle as U
gain = float(os.environ.get("GAIN", "300"))
dev = torch.device("cuda:0")
x, t, lr = golden_inputs("fl.tl", 2, 2, 3, 128, 2, 1500)
sd = synthetic.trained_like_state_dict(T._template(), T._calibrate(x, t, lr), seed=3, bare_gain=gain)
taps = {}
with torch.no_grad():
    want = U.unet_forward(sd, x, t, lr, 2, taps=taps)
m = T._model(dev, sd)
eng = m.hip_engine(); eng.set_impl("mfma_bf16x3")
with torch.no_grad():
    got = m(x.to(dev), t.to(dev), lr.to(dev), 2)
print("DRS_FL", os.environ.get("DRS_FL"), "production plan: max-rel %.3e rel-L2 %.3e" % rel_errors(got.cpu(), want))
eng.keep_intermediates = True
with torch.no_grad():
    got = m(x.to(dev), t.to(dev), lr.to(dev), 2)
print("keep-all plan: max-rel %.3e rel-L2 %.3e" % rel_errors(got.cpu(), want))
for k in eng.tensor_names():
    if k in taps:
        e = rel_errors(eng.read_tensor(k).cpu(), taps[k])
        print(f"   {k:34s} max-rel {e[0]:.2e} rel-L2 {e[1]:.2e}   |ref|max {float(taps[k].abs().max()):.3g}")

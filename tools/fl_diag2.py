#!/usr/bin/env python3
"""Reproduce: a clean forward flagging DRS_ERR_RANGE after other work dirtied the allocator's memory."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffusionremotesensing_amd import _lib, synthetic
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
dev = torch.device("cuda:0")
# dirty a lot of device memory with huge values / NaNs, then free it
junk = [torch.full((64 << 20,), float("nan"), device=dev) for _ in range(8)]
junk += [torch.full((64 << 20,), 3e38, device=dev) for _ in range(8)]
torch.cuda.synchronize()
del junk
m = Residual_Attention_UNet_superres(3, 3, dev)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
eng = m.hip_engine(); eng.set_impl("mfma_bf16x3")
x = synthetic.tensor_normal("nan.x", (3, 3, 128, 128)).to(dev)
lr = synthetic.tensor_uniform("nan.lr", (3, 3, 64, 64)).to(dev)
t = torch.tensor([5, 700, 1400], device=dev)
with torch.no_grad():
    y = m(x, t, lr, 2)
try:
    eng.check_faults(); print("clean")
except _lib.RangeFault as e:
    print("RANGE FAULT on a clean forward")
print("out finite", bool(torch.isfinite(y).all()))
for n in eng.tensor_names():
    v = eng.read_tensor(n)
    bad = (~torch.isfinite(v)).sum().item()
    mx = float(v[torch.isfinite(v)].abs().max()) if bad < v.numel() else float("nan")
    if bad or mx > 1e4:
        print(f"  {n:34s} non-finite {bad:9d} of {v.numel():9d}   max finite {mx:.3g}")

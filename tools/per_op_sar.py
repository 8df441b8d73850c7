"""Per-op HIP-event times of one SAR->NDVI forward at BASELINE configs[3] (B = 32, 128x128).  Usage: python tools/per_op_sar.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffusionremotesensing_amd import synthetic  # noqa: E402
from diffusionremotesensing_amd.UNet_model_SAR_TO_NDVI import Residual_Attention_UNet_SAR_TO_NDVI  # noqa: E402

dev = torch.device("cuda:0")
m = Residual_Attention_UNet_SAR_TO_NDVI(2, 1, dev)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
eng = m.hip_engine()
x = synthetic.tensor_normal("sar.x", (32, 1, 128, 128)).to(dev)
sar = synthetic.tensor_uniform("sar.sar", (32, 2, 128, 128)).to(dev)
t = torch.full((32,), 700, dtype=torch.int64, device=dev)
with torch.no_grad():
    rows = eng.profile_forward(x, t, sar, 1, iters=10)
tot = sum(r[1] for r in rows)
for name, ms, fl, by in rows:
    print(f"{name:36s} {ms:8.4f} {100*ms/tot:5.1f} {fl/ms/1e9 if ms else 0:8.1f} TF {by/ms/1e6 if ms else 0:8.0f} GB/s")
print("total", tot)

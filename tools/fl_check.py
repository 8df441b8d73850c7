#!/usr/bin/env python3
"""Development check of the FL arithmetic (conv_mfma_fl.hip): configs[1] forward against the oracle and the kernels the plan launched.
Run once per setting of DRS_FL (read at library load): DRS_FL=0 python tools/fl_check.py ; DRS_FL=1 python tools/fl_check.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffusionremotesensing_amd import synthetic  # noqa: E402
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres  # noqa: E402

B = int(os.environ.get("FLC_BATCH", "16"))
S = int(os.environ.get("FLC_IMAGE", "256"))
dev = torch.device("cuda:0")
m = Residual_Attention_UNet_superres(3, 3, dev)
sd = synthetic.seeded_state_dict(m.state_dict(), 0)
m.load_state_dict(sd)
m = m.to(dev).eval()
eng = m.hip_engine()
eng.set_impl("mfma_bf16x3")
x = synthetic.tensor_normal("bench.x", (B, 3, S, S))
lr = synthetic.tensor_uniform("bench.lr", (B, 3, S // 2, S // 2))
t = synthetic.tensor_randint("bench.t", (B,), 1, 1500)
with torch.no_grad():
    got = m(x.to(dev), t.to(dev), lr.to(dev), 2).cpu()
    torch.cuda.synchronize()
    _, log = eng.logged_forward(x.to(dev), t.to(dev), lr.to(dev), 2, reuse_cond=True, check_weights=False)
    torch.cuda.synchronize()
eng.check_faults()
names = {}
for op, k in log:
    kk = k.split("(")[0][-60:]
    names[kk] = names.get(kk, 0) + 1
print("DRS_FL =", os.environ.get("DRS_FL", "(default)"))
for k, v in sorted(names.items()):
    print(f"   {v:3d} x {k}")
if os.environ.get("FLC_ORACLE", "1") == "1":
    from oracle import unet_oracle as U
    t0 = time.time()
    with torch.no_grad():
        want = U.unet_forward(sd, x, t, lr, 2)
    d = (got - want)
    print(f"vs oracle ({time.time()-t0:.0f} s): max-rel {float(d.abs().max() / want.abs().max()):.3e}  rel-L2 {float(d.norm() / want.norm()):.3e}  finite {bool(torch.isfinite(got).all())}")
torch.save(got, os.environ.get("FLC_OUT", "/tmp/flc_out.pt"))

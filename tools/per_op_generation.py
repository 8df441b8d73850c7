import os, sys
sys.path.insert(0, os.getcwd())
import torch
from diffusionremotesensing_amd import synthetic
from diffusionremotesensing_amd.generate_new_imgs.UNet_model_generation import Residual_Attention_UNet_generation
dev = torch.device("cuda:0")
m = Residual_Attention_UNet_generation(3, 3, 10, dev)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
eng = m.hip_engine()
x = synthetic.tensor_normal("g.x", (128, 3, 64, 64)).to(dev)
t = torch.full((128,), 700, dtype=torch.int64, device=dev)
y = torch.cat([torch.arange(64) % 10, torch.full((64,), -1)]).to(dev)
with torch.no_grad():
    eng.forward(x, t, None, 1, labels=y)
    plan = eng._last_plan
    import ctypes as C
    from diffusionremotesensing_amd import _lib
    lib = plan.lib
    acc = {}; order = []
    lib.drs_unet_profile_enable(plan.handle, 1)
    for _ in range(10):
        eng.forward(x, t, None, 1, labels=y)
        n = lib.drs_unet_profile_num_ops(plan.handle)
        name = C.create_string_buffer(128); ms, fl, by = C.c_float(), C.c_double(), C.c_double()
        for i in range(n):
            lib.drs_unet_profile_read(plan.handle, i, name, 128, C.byref(ms), C.byref(fl), C.byref(by))
            k = name.value.decode()
            if k not in acc: acc[k] = [0.0, fl.value, by.value]; order.append(k)
            acc[k][0] += ms.value
    lib.drs_unet_profile_enable(plan.handle, 0)
tot = sum(acc[k][0] for k in order) / 10
for k in order:
    ms = acc[k][0] / 10
    print(f"{k:36s} {ms:8.4f} {100*ms/tot:5.1f} {acc[k][1]/ms/1e9 if ms else 0:8.1f} TF {acc[k][2]/ms/1e6 if ms else 0:8.0f} GB/s")
print("total", tot)

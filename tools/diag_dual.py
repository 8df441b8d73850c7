import os, sys, torch
sys.path.insert(0, os.getcwd())
from diffusionremotesensing_amd import synthetic
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
dev = torch.device("cuda:0")
tmpl = Residual_Attention_UNet_superres(3, 3, "cpu")
sd = synthetic.seeded_state_dict(tmpl.state_dict(), 0)
m = Residual_Attention_UNet_superres(3, 3, dev); m.load_state_dict(sd); m = m.to(dev).eval()
for (B, H, W, mag, Bl) in [(2, 16, 16, 2, 2), (2, 32, 32, 2, 2), (1, 48, 32, 2, 1), (3, 32, 32, 2, 1), (2, 32, 32, 4, 2), (2, 64, 64, 2, 2), (16, 256, 256, 2, 16)]:
    x = synthetic.tensor_normal(f"d.x{B}{H}{W}", (B, 3, H, W)).to(dev)
    lr = synthetic.tensor_uniform(f"d.lr{B}{H}{W}", (Bl, 3, H // mag, W // mag)).to(dev)
    t = torch.full((B,), 700, dtype=torch.int64, device=dev)
    outs = {}
    for impl in ("mfma_f32", "mfma_bf16x3"):
        m.hip_engine().set_impl(impl)
        with torch.no_grad():
            outs[impl] = m(x, t, lr, mag).clone()
    e = (outs["mfma_bf16x3"] - outs["mfma_f32"]).abs().max().item() / outs["mfma_f32"].abs().max().item()
    print((B, H, W, mag, Bl), "max-rel", f"{e:.2e}")

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference itself.

Runs ONLY in the build container (needs /root/reference; never on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_golden.py

The reference model files import as-is.  `train_diffusion_superres.py` also imports torchvision, cv2 and
imageio, which are absent from this image and are touched only by dataset / VGG-loss / video code, never by
the schedule, `noise_images` or `sample` arithmetic; empty placeholder modules are registered for those
names before the import.  Weights come from the seeded generator in
diffusionremotesensing_amd/synthetic.py (the reference ships none), loaded with `load_state_dict`.

Only data is written: inputs that cannot be regenerated from a seed, and the reference's outputs.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

for name in ["torchvision", "torchvision.transforms", "torchvision.models", "torchvision.datasets", "cv2", "imageio"]:
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
tv = sys.modules["torchvision"]
tv.transforms = sys.modules["torchvision.transforms"]
tv.models = sys.modules["torchvision.models"]
tv.datasets = sys.modules["torchvision.datasets"]

from UNet_model_superres import Residual_Attention_UNet_superres as RefUNet  # noqa: E402
import train_diffusion_superres as ref_train  # noqa: E402

from diffusionremotesensing_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def ref_model(seed=0, channels=3):
    m = RefUNet(channels, channels, "cpu")
    sd = synthetic.seeded_state_dict(m.state_dict(), seed)
    m.load_state_dict(sd)
    return m.eval()


def inputs(tag, B, Bl, C, S, mag, T, seed=0):
    x = synthetic.tensor_normal(f"{tag}.x", (B, C, S, S), seed)
    lr = synthetic.tensor_uniform(f"{tag}.lr", (Bl, C, S // mag, S // mag), seed)
    t = synthetic.tensor_randint(f"{tag}.t", (B,), 1, T, seed)
    return x, t, lr


def hook_taps(m):
    taps = {}

    def mk(name):
        def hook(_mod, _inp, out):
            taps[name] = out.detach().clone()
        return hook

    names = ["LR_encoder", "bottle_neck"]
    names += [f"conv_blocks.{i}" for i in range(3)] + [f"downs.{i}" for i in range(3)]
    for i in range(3):
        names += [f"gating_signals.{i}", f"attention_blocks.{i}", f"ups.{i}", f"up_convs.{i}"]
    mods = dict(m.named_modules())
    hs = [mods[n].register_forward_hook(mk(n)) for n in names]
    return taps, hs


def main():
    g = {}
    # ---- G1: positional encoding (UNet_model_superres.py:328-335) ----
    m = ref_model()
    tt = torch.tensor([1, 2, 49, 750, 1499]).unsqueeze(-1).float()
    g["g1_t"] = tt.numpy()
    g["g1_pos_encoding"] = m.pos_encoding(tt, 100, "cpu").numpy()

    # ---- G2: noise schedules (train_diffusion_superres.py:117-169) ----
    for kind, T in (("cosine", 50), ("cosine", 1000), ("cosine", 1500), ("linear", 1000)):
        d = ref_train.Diffusion(kind, m, "/nonexistent/snapshot.pt", noise_steps=T, device="cpu", magnification_factor=2,
                                image_size=32, Degradation_type="DownBlur")
        g[f"g2_{kind}_{T}_alpha"] = d.alpha.numpy()
        g[f"g2_{kind}_{T}_alpha_hat"] = d.alpha_hat.numpy()
        g[f"g2_{kind}_{T}_beta"] = d.beta.numpy()

    # ---- G3: block-level eval outputs at B=2, HR 16x16 / LR 8x8 ----
    x, t, lr = inputs("g3", 2, 2, 3, 16, 2, 1500)
    taps, hs = hook_taps(m)
    with torch.no_grad():
        out = m(x, t, lr, 2)
    for h in hs:
        h.remove()
    g["g3_out"] = out.numpy()
    for k, v in taps.items():
        g["g3_tap_" + k] = v.numpy()

    # ---- G4: whole-UNet eval forward, B=2 HR 64 / LR 32; lr batch-1 broadcast; mag 4 ----
    x, t, lr = inputs("g4", 2, 2, 3, 64, 2, 1500)
    with torch.no_grad():
        g["g4_out"] = m(x, t, lr, 2).numpy()
        g["g4_out_lr_broadcast"] = m(x, t, lr[:1], 2).numpy()
    x, t, lr = inputs("g4m4", 1, 1, 3, 64, 4, 1500)
    with torch.no_grad():
        g["g4_out_mag4"] = m(x, t, lr, 4).numpy()
    # non-square, one image: H=24, W=40
    xr = synthetic.tensor_normal("g4r.x", (1, 3, 24, 40))
    lrr = synthetic.tensor_uniform("g4r.lr", (1, 3, 12, 20))
    tr = torch.tensor([77])
    with torch.no_grad():
        g["g4_out_rect"] = m(xr, tr, lrr, 2).numpy()

    # ---- G5: train-mode forward + backward + one Adam step (train_diffusion_superres.py:384-393) ----
    m5 = ref_model()
    m5.train()
    x, t, lr = inputs("g5", 4, 4, 3, 32, 2, 1500)
    noise = synthetic.tensor_normal("g5.noise", (4, 3, 32, 32))
    opt = torch.optim.Adam(m5.parameters(), lr=1e-4)
    opt.zero_grad()
    before = {k: v.detach().clone() for k, v in m5.named_parameters()}
    pred = m5(x, t, lr, 2)
    loss = torch.nn.MSELoss()(pred, noise)
    loss.backward()
    g["g5_out"] = pred.detach().numpy()
    g["g5_loss"] = np.array(loss.item(), dtype=np.float64)
    names = [k for k, _ in m5.named_parameters()]
    g["g5_grad_norms"] = np.array([(p.grad.norm().item() if p.grad is not None else -1.0) for _, p in m5.named_parameters()])
    g["g5_grad_output_bias"] = dict(m5.named_parameters())["output.bias"].grad.numpy()
    g["g5_grad_conv0_weight"] = dict(m5.named_parameters())["conv0.weight"].grad.numpy()
    opt.step()
    g["g5_delta_norms"] = np.array([(p.detach() - before[k]).norm().item() for k, p in m5.named_parameters()])
    sd5 = m5.state_dict()
    for k in ("conv_blocks.0.batch_norm1", "bottle_neck.batch_norm2", "attention_blocks.2.result.1"):
        g[f"g5_rm_{k}"] = sd5[k + ".running_mean"].numpy()
        g[f"g5_rv_{k}"] = sd5[k + ".running_var"].numpy()
    with open(os.path.join(OUT, "g5_param_names.txt"), "w") as f:
        f.write("\n".join(names) + "\n")

    # ---- G6: noise_images (train_diffusion_superres.py:171-190) ----
    d = ref_train.Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=1500, device="cpu",
                            magnification_factor=2, image_size=32, Degradation_type="DownBlur")
    x0 = synthetic.tensor_uniform("g6.x0", (4, 3, 32, 32))
    t6 = torch.tensor([1, 17, 750, 1499])
    torch.manual_seed(606)
    x_t, eps = d.noise_images(x0, t6)
    g["g6_t"] = t6.numpy()
    g["g6_eps"] = eps.numpy()
    g["g6_x_t"] = x_t.numpy()

    # ---- G7: end-to-end Diffusion.sample (train_diffusion_superres.py:207-255), CPU generator seeded ----
    for tag, n, S, T, seed in (("small", 2, 64, 50, 1234), ("cfg1", 4, 128, 50, 4321)):
        d = ref_train.Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=T, device="cpu",
                                magnification_factor=2, image_size=S, Degradation_type="DownBlur")
        lr1 = synthetic.tensor_uniform(f"g7.{tag}.lr", (3, S // 2, S // 2))
        torch.manual_seed(seed)
        xs = d.sample(n, m, lr1, input_channels=3, generate_video=False)
        m.eval()
        g[f"g7_{tag}_x"] = xs.numpy().astype(np.float32)  # (cfg1 was fp16 until round 3: its rounding capped the test's bound)
        g[f"g7_{tag}_checksum"] = np.array([xs.double().sum().item(), xs.double().abs().sum().item()])

    # ---- G10: a FULL-LENGTH chain (T = 1500, the configs[1] schedule) at 32x32, n = 2: the update arithmetic of
    # train_diffusion_superres.py:234-249 at every t.  With seeded random weights a 1499-step chain is chaotic (the network
    # output is not a noise estimate), so the `output` projection is scaled by 1e-2: the chain is then dominated by the
    # schedule's own arithmetic and implementation differences stay comparable.  The reference returns only the final x;
    # the states entering steps 1400 / 1000 / 500 / 100 / 1 are recorded by a pass-through wrapper around the model.
    class Recorder(torch.nn.Module):
        def __init__(self, inner, at):
            super().__init__()
            self.inner, self.at, self.seen = inner, set(at), {}

        def forward(self, x, t, lr_img, mag):
            if int(t[0]) in self.at:
                self.seen[int(t[0])] = x.detach().clone()
            return self.inner(x, t, lr_img, mag)

    m10 = RefUNet(3, 3, "cpu")
    sd10 = synthetic.seeded_state_dict(m10.state_dict(), 0)
    sd10["output.weight"] = sd10["output.weight"] * 1e-2
    sd10["output.bias"] = sd10["output.bias"] * 1e-2
    m10.load_state_dict(sd10)
    rec = Recorder(m10.eval(), (1400, 1000, 500, 100, 1))
    d = ref_train.Diffusion("cosine", rec, "/nonexistent/snapshot.pt", noise_steps=1500, device="cpu",
                            magnification_factor=2, image_size=32, Degradation_type="DownBlur")
    lr1 = synthetic.tensor_uniform("g10.lr", (3, 16, 16))
    torch.manual_seed(1010)
    xs = d.sample(2, rec, lr1, input_channels=3, generate_video=False)
    g["g10_x"] = xs.numpy().astype(np.float32)
    for i_, v in sorted(rec.seen.items()):
        g[f"g10_x_entering_{i_}"] = v.numpy().astype(np.float32)

    np.savez_compressed(os.path.join(OUT, "superres_golden.npz"), **g)
    total = sum(v.nbytes for v in g.values())
    print(f"wrote {len(g)} arrays, {total/1e6:.2f} MB raw ->", os.path.join(OUT, "superres_golden.npz"),
          os.path.getsize(os.path.join(OUT, "superres_golden.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()

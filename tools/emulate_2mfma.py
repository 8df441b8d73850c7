#!/usr/bin/env python3
"""Verdict r03 item 6, closed by measurement: what would a TWO-MFMA-per-product convolution path deliver?

The shipped path multiplies split-bf16 operands (x ~ hi + lo, 16 mantissa bits) with three MFMAs per product.  Two MFMAs
per product means ONE operand is a single 16-bit value; the widest 16-bit mantissa is fp16's (11 bits), so the candidates are
    activations fp16, weights fp16 hi + lo  (a * w_hi + a * w_lo)      - the form the verdict names
    activations fp16 hi + lo, weights fp16  (a_hi * w + a_lo * w)
The operand rounding is emulated on the CPU oracle (oracle/unet_oracle.py): every convolution input with >= 16 channels
(everything the MFMA kernels touch) is rounded as the scheme prescribes, accumulation stays fp32 - exactly what the GPU
kernels do.  The emulation is calibrated against the two schemes that DO exist as GPU kernels:
    split bf16 x split bf16: emulated 1.2e-5 / 9.3e-6, measured on MI355X 1.4e-5 / 1.1e-5 (max-rel / rel-L2)
    single fp16 x single fp16: emulated 1.1e-3 / 8.4e-4, measured 1.1e-3 / 8e-4
Speed of a two-MFMA build was measured in round 3 (profiles/r03_experiment_2mfma.jsonl): +9.7 % steps/s.
Usage: python tools/emulate_2mfma.py > profiles/r04_2mfma_emulation.txt      (CPU only, ~3 minutes)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import diffusion_oracle as D  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402
from diffusionremotesensing_amd import synthetic  # noqa: E402
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres  # noqa: E402

torch.set_num_threads(os.cpu_count() or 8)
sd = synthetic.seeded_state_dict(Residual_Attention_UNet_superres(3, 3, "cpu").state_dict(), 0)
orig_conv, orig_convT, orig_oconv = F.conv2d, F.conv_transpose2d, O._conv


def rnd(x, mode):
    if mode == "f16":
        return x.half().float()
    if mode == "bf16x2":
        h = x.bfloat16().float()
        return h + (x - h).bfloat16().float()
    if mode == "f16x2":
        h = x.half().float()
        return h + (x - h).half().float()
    return x


class Scheme:
    """Context: round the operands of every MFMA-sized convolution (Cin >= 16) of the oracle."""

    def __init__(self, act, wgt, only=None):
        self.act, self.wgt, self.only = act, wgt, only

    def __enter__(self):
        def oconv(sd_, pfx, x, stride=1, padding=0):
            if x.shape[1] >= 16 and (self.only is None or pfx in self.only):
                return orig_conv(rnd(x, self.act), rnd(sd_[pfx + ".weight"], self.wgt), sd_[pfx + ".bias"], stride=stride, padding=padding)
            return orig_oconv(sd_, pfx, x, stride, padding)

        def convT(x, w, b=None, **kw):
            if self.only is None or "transform" in self.only:
                x, w = rnd(x, self.act), rnd(w, self.wgt)
            return orig_convT(x, w, b, **kw)
        O._conv, F.conv_transpose2d = oconv, convT

    def __exit__(self, *a):
        O._conv, F.conv_transpose2d = orig_oconv, orig_convT


def errs(a, b):
    d = (a - b).double()
    return (d.abs().max() / b.abs().max()).item(), (d.norm() / b.double().norm()).item()


def main():
    B, S = 2, 128
    x = synthetic.tensor_normal("e.x", (B, 3, S, S))
    lr = synthetic.tensor_uniform("e.lr", (B, 3, S // 2, S // 2))
    t = synthetic.tensor_randint("e.t", (B,), 1, 1500)
    with torch.no_grad():
        ref = O.unet_forward(sd, x, t, lr, 2)
        print("# one eval forward, B=2 128x128, seeded weights; error against the fp32 oracle (max-abs/max-abs-ref, rel-L2)")
        print(f"{'activations':12s} {'weights':10s} {'MFMAs/product':>14s} {'max-rel':>10s} {'rel-L2':>10s}")
        for act, wgt, nm in (("bf16x2", "bf16x2", 3), ("f16", "f16x2", 2), ("f16x2", "f16", 2), ("f16", "f16", 1)):
            with Scheme(act, wgt):
                e = errs(O.unet_forward(sd, x, t, lr, 2), ref)
            print(f"{act:12s} {wgt:10s} {nm:14d} {e[0]:10.2e} {e[1]:10.2e}", flush=True)
        # per-layer sensitivity of the verdict's form (fp16 activations, exact weights ~ fp16 hi + lo)
        names = []

        def probe(sd_, pfx, x_, stride=1, padding=0):
            if x_.shape[1] >= 16:
                w = sd_[pfx + ".weight"]
                names.append((pfx, 2 * x_.shape[1] * w.shape[0] * w.shape[2] ** 2 * x_.shape[2] * x_.shape[3] // stride ** 2))
            return orig_oconv(sd_, pfx, x_, stride, padding)
        O._conv = probe
        O.unet_forward(sd, x, t, lr, 2)
        O._conv = orig_oconv
        tot = sum(f for _, f in names) * 1.0
        print("\n# fp16 activations in ONE layer at a time (everything else fp32): that layer's share of the conv FLOPs, error of the forward")
        rows = []
        for n, f in names:
            with Scheme("f16", "none", only={n}):
                e = errs(O.unet_forward(sd, x, t, lr, 2), ref)
            rows.append((n, f / tot, e))
            print(f"{n:40s} {100 * f / tot:5.1f} %  max-rel {e[0]:.2e}  rel-L2 {e[1]:.2e}", flush=True)
        # the deep wave-specialised layers only (the candidates of a mixed scheme): errors add in quadrature
        deep = [r for r in rows if r[0].startswith(("conv_blocks.1.conv", "conv_blocks.2.conv", "bottle_neck.conv"))]
        with Scheme("f16", "none", only={r[0] for r in deep}):
            e = errs(O.unet_forward(sd, x, t, lr, 2), ref)
        print(f"\n# mixed scheme, two MFMAs only in conv1 / conv2 of encoder blocks 1, 2 and the bottleneck ({100 * sum(r[1] for r in deep):.0f} % of the FLOPs): "
              f"max-rel {e[0]:.2e} rel-L2 {e[1]:.2e}")
        # configs[0] chain (BASELINE 'PSNR vs ref'): n=4, 64 -> 128, T=50, the reference's draws replayed
        from conftest import replay_noise_source
        g = np.load(os.path.join(ROOT, "tests", "golden", "superres_golden.npz"))
        a, ah, b = D.schedule("cosine", 50)
        lr1 = synthetic.tensor_uniform("g7.cfg1.lr", (3, 64, 64))
        want = torch.from_numpy(g["g7_cfg1_x"]).float()
        print("\n# configs[0] chain against the reference's own output (PSNR on [0,1]-clamped images)")
        for act, wgt in (("bf16x2", "bf16x2"), ("f16", "f16x2")):
            with Scheme(act, wgt):
                got = D.sample(O.OracleUNet(sd), 4, lr1, 50, a, ah, b, 2, 128, noise_source=replay_noise_source(4321))
            mse = ((got.clamp(0, 1) - want.clamp(0, 1)).double() ** 2).mean().item()
            e = errs(got, want)
            print(f"{act} x {wgt}: rel-L2 {e[1]:.2e}  PSNR {10 * np.log10(1.0 / mse):.1f} dB", flush=True)


if __name__ == "__main__":
    main()

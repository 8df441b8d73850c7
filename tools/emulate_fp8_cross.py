#!/usr/bin/env python3
"""What would "bf16 main product + BOTH cross terms as one fp8 MFMA" deliver?  (CPU emulation, no kernel exists.)

The shipped path computes a*w ~ ah*wh + ah*wl + al*wh with three bf16 MFMAs (16 cycles each for a 16x16x32 step).  gfx950's
v_mfma_scale_f32_16x16x128_f8f6f4 multiplies fp8 operands at twice the bf16 rate, takes a power-of-two scale per 32-element
K block (E8M0) and accumulates into the same fp32 16x16 tile: the two cross terms of TWO (tap, chunk) steps fit one such
instruction ([q(ah) | q(al)] . [q(wl) | q(wh)], K = 64 per step), i.e. 16 + 16 cycles per step instead of 48 - the matrix
pipe's share of the step would shrink by a third.  The cross terms are ~2^-8 of the main product, so their operands need
few bits; what this costs in accuracy is measured here the way tools/emulate_2mfma.py does it (operand roundings applied
inside the CPU oracle's convolutions, fp32 accumulation), for three quantisers of the cross-term operands:
    e4m3 + a power-of-two scale per (pixel | output channel, tap) and 32-channel block   - the MX form the instruction takes
    e4m3 + one power-of-two scale per tensor
    e5m2 unscaled
    FP6 e2m3 + the per-block scale (the instruction runs fp6 operands at FOUR times the bf16 rate: 16 + 8 cycles per step)
and for the main product in fp16 instead of bf16 ("f16+..."): x ~ xh + xl with xh fp16 leaves remainders of 2^-12, so the
same few-bit cross terms sit three bits lower (the model's activations and folded weights are inside fp16's range: the
single-fp16 line of r04_2mfma_emulation.txt is a measured GPU build).
Usage: python tools/emulate_fp8_cross.py > profiles/r04_fp8_cross_emulation.txt      (CPU only)"""
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
E4M3_MAX = 448.0


def po2_scale(amax, top):
    """largest power of two s with amax * s <= top (1 where amax == 0)"""
    amax = amax.clamp_min(1e-30)
    return torch.exp2(torch.floor(torch.log2(top / amax)))


def q_e4m3(x, s):
    return (x * s).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() / s


E2M3_MAX = 7.5


def q_e2m3(x, s):
    """OCP MX FP6 E2M3 (bias 1: subnormals k / 8, normals 2^(e-1) * (1 + m / 8), max 7.5), round to nearest even"""
    v = (x * s).clamp(-E2M3_MAX, E2M3_MAX)
    a = v.abs()
    step = torch.exp2(torch.floor(torch.log2(a.clamp_min(1.0)))) / 8  # 1/8 below 2 (incl. subnormals), 1/4 in [2, 4), 1/2 in [4, 8)
    return torch.sign(v) * torch.round(a / step) * step / s


def quant(x, mode, cdim):
    """x quantised for a cross term; blocks of 32 along dimension `cdim` (the K dimension of the product)"""
    if mode == "e5m2":
        return x.to(torch.float8_e5m2).float()
    if mode == "e4m3_tensor":
        return q_e4m3(x, po2_scale(x.abs().max(), E4M3_MAX))
    assert mode in ("e4m3_block", "e2m3_block")
    C = x.shape[cdim]
    blk = 32 if C % 32 == 0 else C
    xs = x.movedim(cdim, -1)
    shp = xs.shape
    xb = xs.reshape(*shp[:-1], C // blk, blk)
    amax = xb.abs().amax(dim=-1, keepdim=True)
    q = q_e4m3(xb, po2_scale(amax, E4M3_MAX)) if mode == "e4m3_block" else q_e2m3(xb, po2_scale(amax, E2M3_MAX))
    return q.reshape(shp).movedim(-1, cdim)


class Scheme:
    """bf16 main product, fp8 cross terms, in every MFMA-sized convolution (Cin >= 16) of the oracle"""

    def __init__(self, mode):
        self.mode = mode

    def product(self, conv, x, w, wc):
        m = self.mode
        if m.startswith("f16+"):  # main product in fp16 (11 mantissa bits: the remainders are 2^-12 of the operand, not 2^-9)
            m = m[4:]
            xh, wh = x.half().float(), w.half().float()
        else:
            xh, wh = x.bfloat16().float(), w.bfloat16().float()
        xl, wl = x - xh, w - wh
        if m == "bf16x3":
            xl, wl = xl.bfloat16().float(), wl.bfloat16().float()
            return conv(xh, wh) + conv(xh, wl) + conv(xl, wh)
        return conv(xh, wh) + conv(quant(xh, m, 1), quant(wl, m, wc)) + conv(quant(xl, m, 1), quant(wh, m, wc))

    def __enter__(self):
        def oconv(sd_, pfx, x, stride=1, padding=0):
            if x.shape[1] < 16:
                return orig_oconv(sd_, pfx, x, stride, padding)
            y = self.product(lambda a, b: orig_conv(a, b, None, stride=stride, padding=padding), x, sd_[pfx + ".weight"], 1)
            return y + sd_[pfx + ".bias"].view(1, -1, 1, 1)

        def convT(x, w, b=None, **kw):
            y = self.product(lambda a, c: orig_convT(a, c, None, **kw), x, w, 0)  # (ConvTranspose2d weights are [Cin, Cout, kh, kw])
            return y if b is None else y + b.view(1, -1, 1, 1)
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
    modes = ("bf16x3", "e4m3_block", "e2m3_block", "f16+e4m3_block", "f16+e2m3_block", "e4m3_tensor", "e5m2", "f16+e5m2")
    cyc = {"bf16x3": 48, "e2m3_block": 24, "f16+e2m3_block": 24}
    with torch.no_grad():
        ref = O.unet_forward(sd, x, t, lr, 2)
        print("# one eval forward, B=2 128x128, seeded weights; error against the fp32 oracle (max-abs/max-abs-ref, rel-L2)")
        print(f"{'cross terms':14s} {'matrix-pipe cycles/step':>24s} {'max-rel':>10s} {'rel-L2':>10s}")
        for m in modes:
            with Scheme(m):
                e = errs(O.unet_forward(sd, x, t, lr, 2), ref)
            print(f"{m:14s} {cyc.get(m, 32):24d} {e[0]:10.2e} {e[1]:10.2e}", flush=True)
        from conftest import replay_noise_source
        g = np.load(os.path.join(ROOT, "tests", "golden", "superres_golden.npz"))
        a, ah, b = D.schedule("cosine", 50)
        lr1 = synthetic.tensor_uniform("g7.cfg1.lr", (3, 64, 64))
        want = torch.from_numpy(g["g7_cfg1_x"]).float()
        print("\n# configs[0] chain against the reference's own output (PSNR on [0,1]-clamped images)")
        for m in modes[:5]:
            with Scheme(m):
                got = D.sample(O.OracleUNet(sd), 4, lr1, 50, a, ah, b, 2, 128, noise_source=replay_noise_source(4321))
            mse = ((got.clamp(0, 1) - want.clamp(0, 1)).double() ** 2).mean().item()
            e = errs(got, want)
            print(f"{m}: rel-L2 {e[1]:.2e}  PSNR {10 * np.log10(1.0 / mse):.1f} dB", flush=True)


if __name__ == "__main__":
    main()

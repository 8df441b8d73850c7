#!/usr/bin/env python3
"""Verdict item 7 of round 4 (exploratory): does splitting the 16-image batch into two 8-image half-plans on two streams let the
HBM-bound launches of one half run beside the matrix-bound launches of the other?  Build the library with -DDRS_X_NUM_CU
(tools/build_variant.sh half -DDRS_X_NUM_CU) and run with DRS_LIB pointing at it:
    DRS_X_NUM_CU=0|128|192 python tools/two_stream_probe.py
Prints denoise-forward rates (batch-16 equivalents per second) of: one 16-image plan; two 8-image plans back to back on one
stream; the two 8-image plans on two streams (persistent grids sized for DRS_X_NUM_CU compute units, 0 = all)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionremotesensing_amd import synthetic
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres

dev = torch.device("cuda:0")
def model():
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
    m = m.to(dev).eval(); m.hip_engine().set_impl("mfma_bf16x3")
    return m
def inputs(b):
    return (synthetic.tensor_normal("bench.x", (b, 3, 256, 256)).to(dev), synthetic.tensor_randint("bench.t", (b,), 1, 1500).to(dev),
            synthetic.tensor_uniform("bench.lr", (b, 3, 128, 128)).to(dev))
def rate(fn, prime, n=200, warm=20):
    prime()  # first forward of a chain: computes the conditioning branch the later ones reuse
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return n / (time.perf_counter() - t0)
with torch.no_grad():
    m16 = model(); x, t, lr = inputs(16)
    one = rate(lambda: m16.hip_engine().forward(x, t, lr, 2, reuse_cond=True), lambda: m16.hip_engine().forward(x, t, lr, 2))
    ma, mb = model(), model()
    xa, ta, la = inputs(8); xb, tb, lb = inputs(8)
    def seq():
        ma.hip_engine().forward(xa, ta, la, 2, reuse_cond=True); mb.hip_engine().forward(xb, tb, lb, 2, reuse_cond=True)
    def prime2():
        ma.hip_engine().forward(xa, ta, la, 2); mb.hip_engine().forward(xb, tb, lb, 2); torch.cuda.synchronize()
    two_seq = rate(seq, prime2)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    def par():
        with torch.cuda.stream(sa): ma.hip_engine().forward(xa, ta, la, 2, reuse_cond=True)
        with torch.cuda.stream(sb): mb.hip_engine().forward(xb, tb, lb, 2, reuse_cond=True)
    two_par = rate(par, prime2)
print(f"DRS_X_NUM_CU={os.environ.get('DRS_X_NUM_CU', '0')}: one 16-image plan {one:.1f}/s | two 8-image plans, one stream {two_seq:.1f}/s | two streams {two_par:.1f}/s")

#!/usr/bin/env python3
"""Latency regime: per-step time of the small / launch-bound sampling shapes the reference's own callers use -
`Diffusion.sample(n=1)` at 256x256 (train-time previews, the tile-after-tile aggregation loop:
train_diffusion_superres.py:422-424, Aggregation_Sampling.py:94-97) and BASELINE configs[4] (class-conditional CFG
sampling, 64x64, batch 64: generate_new_imgs/train_diffusion_generation.py:236-249).  Prints ms per step (wall, K steps
between two synchronisations) next to the sum of the per-op HIP-event times and the launch count of one step.
`--json FILE` also writes the numbers and the per-launch sequence (op name, microseconds) of each workload's step.
Usage: latency_regime.py [--steps 50] [--json FILE]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffusionremotesensing_amd import hip_ops, synthetic  # noqa: E402
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres  # noqa: E402
from diffusionremotesensing_amd.train_diffusion_superres import Diffusion  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--json", default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")


def run(name, step, nsteps):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / nsteps


def superres(batch, image):
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
    m = m.to(dev).eval()
    eng = m.hip_engine()
    d = Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=1500, device=dev, magnification_factor=2,
                  image_size=image, Degradation_type="DownBlur")
    x = synthetic.tensor_normal("lat.x", (batch, 3, image, image)).to(dev)
    lr = synthetic.tensor_uniform("lat.lr", (1, 3, image // 2, image // 2)).to(dev)
    t = torch.empty(batch, dtype=torch.int64, device=dev)
    state = {"i": 1499, "first": True}

    def step():
        i = max(state["i"], 2)
        eps = eng.forward(x, t.fill_(i), lr, 2, reuse_cond=not state["first"], check_weights=state["first"])
        hip_ops.sampler_step_(x, eps, torch.randn_like(x), i, d.alpha, d.alpha_hat, d.beta)
        state["i"] -= 1
        state["first"] = False
    with torch.no_grad():
        ms = run("superres", step, a.steps)
        ops = eng.profile_forward(x, t.fill_(700), lr, 2, iters=5)
    ops = [o for o in ops if o[0] != "lr_branch"]
    return ms, sum(o[1] for o in ops), len(ops) + 2, ops  # + randn + sampler update


def generation():
    from diffusionremotesensing_amd.generate_new_imgs.train_diffusion_generation import Diffusion as GDiffusion
    from diffusionremotesensing_amd.generate_new_imgs.UNet_model_generation import Residual_Attention_UNet_generation
    m = Residual_Attention_UNet_generation(3, 3, 10, dev)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
    m = m.to(dev).eval()
    eng = m.hip_engine()
    d = GDiffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=1000, device=dev, image_size=64)
    x = synthetic.tensor_normal("gen.x", (64, 3, 64, 64)).to(dev)
    labels2 = torch.cat([synthetic.tensor_randint("gen.y", (64,), 0, 10), torch.full((64,), -1, dtype=torch.int64)]).to(dev)
    t2 = torch.empty(128, dtype=torch.int64, device=dev)
    state = {"i": 999, "first": True}

    def step():
        i = max(state["i"], 2)
        eps2 = eng.forward(x.repeat(2, 1, 1, 1), t2.fill_(i), None, 1, labels=labels2, check_weights=state["first"])
        hip_ops.sampler_step_cfg_(x, eps2[:64], eps2[64:], 3.0, torch.randn_like(x), i, d.alpha, d.alpha_hat, d.beta)
        state["i"] -= 1
        state["first"] = False
    with torch.no_grad():
        ms = run("generation", step, a.steps)
        ops = eng.profile_forward(x.repeat(2, 1, 1, 1), t2.fill_(500), None, 1, iters=5, labels=labels2)
    return ms, sum(o[1] for o in ops), len(ops) + 3, ops


report = []
print(f"{'workload':44s} {'ms/step':>8s} {'sum of op ms':>13s} {'launches':>9s} {'gap share':>10s}")
for name, fn in (("superres sample(n=1) 256x256", lambda: superres(1, 256)),
                 ("superres sample(n=4) 128x128 (configs[0])", lambda: superres(4, 128)),
                 ("superres step B=16 256x256 (configs[1])", lambda: superres(16, 256)),
                 ("generation CFG step B=64 64x64 (configs[4])", generation)):
    ms, opsum, n, ops = fn()
    report.append({"workload": name, "ms_per_step": round(ms, 4), "sum_of_op_ms": round(opsum, 4), "launches": n,
                   "sequence_us": [[o[0], round(1e3 * o[1], 1)] for o in ops]})
    print(f"{name:44s} {ms:8.3f} {opsum:13.3f} {n:9d} {max(0.0, 1 - opsum / ms):10.1%}")
if a.json:
    import json
    with open(a.json, "w") as f:
        json.dump({"steps": a.steps, "workloads": report}, f, indent=1)

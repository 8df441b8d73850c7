#!/usr/bin/env python3
"""Headline benchmark: UNet denoise steps/s at BASELINE.json configs[1]
(Residual_Attention_UNet_superres 128x128 -> 256x256, mag 2, batch 16 per GPU, cosine T=1500, eval mode).

A "step" is one iteration of Diffusion.sample's loop (train_diffusion_superres.py:234-249 in the reference):
UNet forward on a 16-image batch + fresh Gaussian noise + the ancestral update, with x, lr_img and the weights
resident in HBM.  N GPUs run N independent 16-image batches (sampling shards by image, no collective on the data
path) => weak scaling; `value` counts batch-steps of all ranks per second, `image_steps_per_s` = 16x that.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--impl direct|mfma_f32|mfma_bf16x3|mfma_f16]
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL).
"""
import argparse
import json
import re
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

BATCH, IMAGE, MAG, T_STEPS = 16, 256, 2, 1500
# algorithmic work per 16-image forward, SURVEY.md section 8(d) / BASELINE.md
GFLOP_PER_FWD = 454.39
MB_PER_FWD = 4998.3
PEAK_TFLOPS = {"direct": 157.3, "mfma_f32": 157.3, "mfma_bf16x3": 2500.0, "mfma_f16": 2500.0}
_FL = os.environ.get("DRS_FL", "1") != "0"
DTYPE = {"direct": "f32", "mfma_f32": "f32",
         "mfma_bf16x3": ("bf16x3 (hi+lo split, f32 accumulate)" +
                         ("; the wide 3x3 layers in the FL arithmetic: f16 main + block-scaled fp6 (e2m3) cross terms" if _FL else "")),
         "mfma_f16": "f16 (f32 accumulate)"}
HBM_PEAK_GBS = 8000.0
# Rehearsal of the N > 1 path on a one-GPU lease (tests/test_gpu_dist.py): DRS_BENCH_BACKEND=gloo DRS_BENCH_SHARE_DEVICE=1 runs the
# ranks of `--gpus N` as N processes on device 0 over gloo (RCCL needs a device per rank).  The driver's runs leave both unset:
# one rank per GPU, backend "nccl" = RCCL over xGMI.  The JSON line records the backend in config.process_group.
_BACKEND = os.environ.get("DRS_BENCH_BACKEND", "nccl")
_SHARE_DEVICE = os.environ.get("DRS_BENCH_SHARE_DEVICE", "") not in ("", "0")


def _host_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def cpu_baseline(sd, x, t, lr):
    """The oracle (CPU port of the reference graph, stock torch ops) on this box's host cores, BASELINE.md section 3
    protocol: 1 warm-up + 5 timed 16-image forwards of configs[1] (min and median), and the configs[0] end-to-end
    `sample` (n=4, 64x64 -> 128x128, T=50: 49 forwards + updates).  About 20 s of CPU work."""
    from oracle import diffusion_oracle, unet_oracle
    from diffusionremotesensing_amd import synthetic
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # a 1-GPU box owns a 16-core share of the host however many cores it can see
    cores = max(1, min(avail, int(os.environ.get("DRS_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    with torch.no_grad():
        unet_oracle.unet_forward(sd, x, t, lr, MAG)
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            unet_oracle.unet_forward(sd, x, t, lr, MAG)
            times.append(time.perf_counter() - t0)
        a, ah, b = diffusion_oracle.schedule("cosine", 50)
        lr1 = synthetic.tensor_uniform("g7.cfg1.lr", (3, 64, 64))
        gen = torch.Generator().manual_seed(4321)
        t0 = time.perf_counter()
        diffusion_oracle.sample(unet_oracle.OracleUNet(sd), 4, lr1, 50, a, ah, b, 2, 128,
                                noise_source=lambda i, shape: torch.randn(shape, generator=gen))
        cfg1_s = time.perf_counter() - t0
    best, med = min(times), sorted(times)[len(times) // 2]
    return {"value": round(1.0 / best, 4), "unit": "batch16_steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "median_forward_s": round(med, 3), "config0_sample_s": round(cfg1_s, 3),
            "sample": f"5 timed eval forwards of the B=16 256x256 batch after 1 warm-up (min {best:.3f} s, median "
                      f"{med:.3f} s) + the configs[0] end-to-end sample n=4 128x128 T=50 ({cfg1_s:.2f} s); "
                      f"torch {torch.__version__} CPU ops, host: {_host_model()}"}


def psnr_vs_reference(model, dev, impl):
    """BASELINE metric "PSNR vs ref": the configs[0] chain (n=4, 64x64 -> 128x128, T=50, cosine) with the reference's CPU
    generator draws replayed (seed 4321), against the reference's own output committed as a fixture
    (tests/golden/superres_golden.npz: g7_cfg1_x, made by tools/make_golden.py from the imported reference;
    train_diffusion_superres.py:207-255).  PSNR on [0,1]-clamped images, like the reference's callers display them."""
    import numpy as np
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
    path = os.path.join(ROOT, "tests", "golden", "superres_golden.npz")
    if not os.path.exists(path):
        return None
    ref = torch.from_numpy(np.load(path)["g7_cfg1_x"]).float()
    d = Diffusion("cosine", model, "/nonexistent/snapshot.pt", noise_steps=50, device=dev, magnification_factor=MAG,
                  image_size=128, Degradation_type="DownBlur")
    gen = torch.Generator().manual_seed(4321)
    x = d.sample(4, model, synthetic.tensor_uniform("g7.cfg1.lr", (3, 64, 64)), input_channels=3,
                 noise_source=lambda i, shape: torch.randn(shape, generator=gen)).cpu()
    model.eval()
    mse = ((x.clamp(0, 1) - ref.clamp(0, 1)) ** 2).mean().item()
    rel_l2 = ((x - ref).norm() / ref.norm()).item()
    return {"psnr_db": round(-10 * torch.log10(torch.tensor(max(mse, 1e-20))).item(), 2), "rel_l2": float(f"{rel_l2:.3e}"),
            "chain": "configs[0]: n=4, 64x64->128x128, T=50, reference noise replayed (fp32 fixture from the reference's own run)"}


def _build_workload(wl, impl, dev, rank, multi):
    """(step, unit, batch, gflop per step, description, dtype) of a non-headline BASELINE.json config: `train` = configs[2]
    (per-rank batch 16 of 256x256, train-mode forward + backward + flat RCCL all-reduce + FusedAdam), `sar` = configs[3]
    (SAR->NDVI sampling step, B=32 128x128), `generation` = configs[4] (class-conditional sampling step with CFG scale 3,
    B=64 64x64, 10 classes)."""
    from diffusionremotesensing_amd import hip_ops, synthetic
    if wl == "train":
        from diffusionremotesensing_amd.optim import FusedAdam
        from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
        from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
        m = Residual_Attention_UNet_superres(3, 3, dev)
        m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
        m = m.to(dev).train()
        d = Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=1500, device=dev, magnification_factor=2,
                      image_size=256, Degradation_type="DownBlur", multiple_gpus=multi)
        hr = synthetic.tensor_uniform("train.hr", (16, 3, 256, 256), seed=rank).to(dev)
        lr = synthetic.tensor_uniform("train.lr", (16, 3, 128, 128), seed=rank).to(dev)
        opt = FusedAdam(m.parameters(), lr=1e-4)
        loss_fn = torch.nn.MSELoss()
        step = lambda: d.train_step(m, opt, loss_fn, lr, hr)  # noqa: E731
        timpl = os.environ.get("DRS_TRAIN_IMPL", "mfma_f32")
        return (step, "train_steps/s (16 images per rank)", 16, 3 * 454.39,
                f"BASELINE configs[2] per-rank shape: superres 256x256 train step, batch 16 per GPU, MSE, Adam, train_impl={timpl}",
                ("forward f32 (v_mfma_f32_16x16x4_f32); backward: data gradients split bf16 x3 (SP-format dZ), weight gradients "
                 "bf16 MFMA on split operands, fp32 accumulate") if timpl == "mfma_f32" else DTYPE.get(timpl, timpl), m)
    if wl == "sar":
        from diffusionremotesensing_amd.train_diffusion_SAR_TO_NDVI import Diffusion
        from diffusionremotesensing_amd.UNet_model_SAR_TO_NDVI import Residual_Attention_UNet_SAR_TO_NDVI
        m = Residual_Attention_UNet_SAR_TO_NDVI(2, 1, dev)
        m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
        m = m.to(dev).eval()
        m.hip_engine().set_impl(impl)
        d = Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=1000, device=dev, image_size=128)
        x = synthetic.tensor_normal("sar.x", (32, 1, 128, 128), seed=rank).to(dev)
        sar = synthetic.tensor_uniform("sar.sar", (32, 2, 128, 128), seed=rank).to(dev)
        t_rows = hip_ops.timestep_table(1000, 32, dev)
        state = {"i": 999, "first": True}

        def step():
            i = max(state["i"], 2)
            eps = m.hip_engine().forward(x, t_rows[i], sar, 1, reuse_cond=not state["first"], check_weights=state["first"])
            hip_ops.sampler_step_(x, eps, torch.randn_like(x), i, d.alpha, d.alpha_hat, d.beta)
            state["i"] -= 1
            state["first"] = False
        return (step, "batch32_steps/s", 32, 32 * 7.088,
                "BASELINE configs[3]: SAR->NDVI UNet 128x128, 1-ch out / 2-ch SAR, batch 32 per GPU, sampling step", DTYPE[impl], m)
    from diffusionremotesensing_amd.generate_new_imgs.train_diffusion_generation import Diffusion
    from diffusionremotesensing_amd.generate_new_imgs.UNet_model_generation import Residual_Attention_UNet_generation
    m = Residual_Attention_UNet_generation(3, 3, 10, dev)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
    m = m.to(dev).eval()
    m.hip_engine().set_impl(impl)
    d = Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=1000, device=dev, image_size=64)
    x = synthetic.tensor_normal("gen.x", (64, 3, 64, 64), seed=rank).to(dev)
    labels2 = torch.cat([synthetic.tensor_randint("gen.y", (64,), 0, 10, seed=rank),
                         torch.full((64,), -1, dtype=torch.int64)]).to(dev)
    t_rows = hip_ops.timestep_table(1000, 128, dev)
    state = {"i": 999, "first": True}

    def step():
        i = max(state["i"], 2)
        eps2 = m.hip_engine().forward(x.repeat(2, 1, 1, 1), t_rows[i], None, 1, labels=labels2,
                                      check_weights=state["first"])
        hip_ops.sampler_step_cfg_(x, eps2[:64], eps2[64:], 3.0, torch.randn_like(x), i, d.alpha, d.alpha_hat, d.beta)
        state["i"] -= 1
        state["first"] = False
    return (step, "batch64_cfg_steps/s", 64, 2 * 64 * 1.771,
            "BASELINE configs[4]: class-conditional generation UNet 64x64, 10 classes, batch 64 per GPU, one CFG "
            "sampling step = conditional + unconditional forward (one 128-row batch) + guided update", DTYPE[impl], m)


def _time_workload(wl, impl, dev, rank, steps, warmup, sync):
    step, unit, batch, gflop, desc, dtype, m = _build_workload(wl, impl, dev, rank, torch.distributed.is_initialized())
    ctx = torch.enable_grad() if wl == "train" else torch.no_grad()
    with ctx:
        for _ in range(max(warmup, 1)):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        elapsed = time.perf_counter() - t0
    m.hip_engine().check_faults()  # (outside the timed region) a forward whose wave-specialised kernels gave up on a counter is not a measurement
    return elapsed, unit, batch, gflop, desc, dtype


def other_workload(args):
    """`--workload train|sar|generation`: the non-headline BASELINE.json configs on the same contract (one JSON line, barrier
    + synchronize around exactly K timed steps, max over ranks)."""
    from diffusionremotesensing_amd import _lib, dist
    _lib.load()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: there is no CPU path to measure")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or "RANK" in os.environ:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus}")
        dist.init_process_group(_BACKEND)
    rank = dist.rank()
    local = 0 if _SHARE_DEVICE else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    wl = args.workload
    if wl == "train" and os.environ.get("DRS_TRAIN_IMPL", "mfma_f32") == "mfma_bf16x3" and rank == 0:
        print("bench.py: DRS_TRAIN_IMPL=mfma_bf16x3 is OUTSIDE the gradient bar (worst gradient-norm deviation 1.26e-3 on the "
              "full-size configs[2] step, DESIGN.md section 2); the number is not a parity-grade training rate", file=sys.stderr)

    def sync():
        if dist.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()
    elapsed, unit, batch, gflop, desc, dtype = _time_workload(wl, args.impl, dev, rank, args.steps, args.warmup, sync)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = el.item()
    value = world * args.steps / elapsed
    if rank == 0:
        print(json.dumps({"metric": f"{wl}_steps_per_s", "value": round(value, 4), "unit": unit, "n_gpus": world,
                          "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
                          "data": "synthetic", "config": {"workload": desc, "batch_per_gpu": batch, "impl": args.impl,
                                                          "process_group": torch.distributed.get_backend() if dist.is_initialized() else None},
                          "images_per_s": round(value * batch, 2),
                          "tflops_algorithmic": round(value * gflop / 1e3, 3)}), flush=True)
    if dist.is_initialized():
        torch.distributed.barrier()
        dist.destroy_process_group()


def other_configs_block(impl, dev):
    """The remaining BASELINE.json configs inside the default (driver) run, a few seconds in all: configs[2] train step
    (10 timed steps), configs[3] SAR->NDVI and configs[4] class-conditional CFG sampling steps (50 each); one GPU."""
    out = {}
    for wl, steps, warm in (("train", 10, 3), ("sar", 50, 5), ("generation", 50, 5)):
        try:
            elapsed, unit, batch, gflop, desc, dtype = _time_workload(wl, impl, dev, 0, steps, warm, torch.cuda.synchronize)
            v = steps / elapsed
            out[wl] = {"value": round(v, 3), "unit": unit, "steps": steps, "ms_per_step": round(1e3 / v, 4), "dtype": dtype,
                       "images_per_s": round(v * batch, 1), "tflops_algorithmic": round(v * gflop / 1e3, 2), "workload": desc}
        except Exception as e:  # a failing side workload must not take the headline line with it
            out[wl] = {"error": f"{type(e).__name__}: {e}"}
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    return out


def norm_kernel(k):
    """Kernel name as tools/collect_pmc.py normalises it (template arguments kept, parameter list and qualifiers dropped)."""
    import re
    k = k.replace("(anonymous namespace)::", "").replace(" [clone .kd]", "")
    k = re.sub(r"\.kd$", "", k.strip())
    k = re.sub(r"^void\s+", "", k)
    return re.sub(r"\s+", "", k.split("(")[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--impl", default=os.environ.get("DRS_IMPL", "mfma_bf16x3"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PSNR chain and the exact-fp32 line")
    ap.add_argument("--workload", default="superres", choices=["superres", "train", "sar", "generation"],
                    help="superres = the headline (BASELINE configs[1]); the others are the remaining configs")
    args = ap.parse_args()
    if args.workload != "superres":
        return other_workload(args)

    from diffusionremotesensing_amd import _lib, dist, hip_ops, synthetic
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion

    _lib.load()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: there is no CPU path to measure")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU, RCCL
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus}")
        dist.init_process_group(_BACKEND)
    rank = dist.rank()
    local = 0 if _SHARE_DEVICE else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    model = Residual_Attention_UNet_superres(3, 3, dev)
    sd = synthetic.seeded_state_dict(model.state_dict(), 0)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    engine = model.hip_engine()
    engine.set_impl(args.impl)
    diffusion = Diffusion("cosine", model, "/nonexistent/snapshot.pt", noise_steps=T_STEPS, device=dev,
                          magnification_factor=MAG, image_size=IMAGE, Degradation_type="DownBlur")

    # synthetic inputs of configs[1], different per rank
    x_cpu = synthetic.tensor_normal("bench.x", (BATCH, 3, IMAGE, IMAGE), seed=rank)
    lr_cpu = synthetic.tensor_uniform("bench.lr", (BATCH, 3, IMAGE // MAG, IMAGE // MAG), seed=rank)
    x = x_cpu.to(dev)
    lr = lr_cpu.to(dev)
    t = torch.empty(BATCH, dtype=torch.int64, device=dev)
    t_rows = hip_ops.timestep_table(T_STEPS, BATCH, dev)  # what Diffusion.sample uses: a row view per step, no fill kernel

    def step(i, first):
        eps = engine.forward(x, t_rows[i], lr, MAG, reuse_cond=not first, check_weights=first)
        noise = torch.randn_like(x) if i > 1 else None
        hip_ops.sampler_step_(x, eps, noise, i, diffusion.alpha, diffusion.alpha_hat, diffusion.beta)

    def sync():
        if dist.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        i = T_STEPS - 1
        for w in range(args.warmup):
            step(i, w == 0)
            i -= 1
        if args.warmup == 0:
            engine.forward(x, t.fill_(i), lr, MAG)  # conditioning + weights in place before the timed region
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(max(i, 1), False)
            i -= 1
        sync()
        elapsed = time.perf_counter() - t0
    engine.check_faults()  # (outside the timed region) a forward whose wave-specialised kernels gave up on a counter is not a measurement
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = el.item()
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed

    result = None
    if rank == 0:
        # roofline of the dominant kernel family (the wide tap-convolutions): per-op HIP events on the launch stream
        with torch.no_grad():
            x.copy_(x_cpu)
            ops = engine.profile_forward(x, t.fill_(750), lr, MAG, iters=5)
        # dominant kernel = the 3x3 stride-1 implicit-GEMM convolution: on the default (split-bf16, SP-format) plan every
        # such layer is ONE kernel template, tapconv_sp_kernel<HAS2, BNB, FUSE, DUAL> (conv_mfma_sp.hip): conv1 / conv2 of
        # the residual blocks (block 0: conv1 + skip fused into one launch), ups.*.conv and up_convs.* (up_convs.2 with the
        # fused output projection) = 14 launches per forward.  achieved = algorithmic FLOPs of those launches (2*MACs,
        # SURVEY.md 8(d)) / their HIP-event durations, i.e. FLOPs per launch / average launch duration.
        def is_dom(name):
            if name.endswith(".edges"):
                return False  # (the composite kernel's edge vectors: 0.1 % of the work, its own small kernel)
            return (name.endswith((".conv1.0", ".conv2.0", ".conv_upsampled_lr_img", ".conv1.0+skip", "conv_blocks.0.fused")) or
                    (name.startswith("ups.") and name.endswith(".conv")) or name.startswith("up_convs."))
        conv = [o for o in ops if o[2] > 0 and o[0] not in ("lr_branch", "conv0")]
        dom = [o for o in conv if is_dom(o[0])] if args.impl != "direct" else conv
        conv_ms = sum(o[1] for o in conv)
        dom_ms = sum(o[1] for o in dom)
        dom_fl = sum(o[2] for o in dom)
        dom_by = sum(o[3] for o in dom)
        all_ms = sum(o[1] for o in ops if o[0] != "lr_branch")  # the step reuses the cached LR conditioning
        lr_op = [o for o in ops if o[0] == "lr_branch"]
        # ... so its algorithmic work is not part of a timed step either (RRDB -> bicubic -> conv: ~2.1 GFLOP, ~170 MB)
        step_gflop = GFLOP_PER_FWD - (lr_op[0][2] / 1e9 if lr_op else 0.0)
        step_mb = MB_PER_FWD - (lr_op[0][3] / 1e6 if lr_op else 0.0)
        top = sorted(ops, key=lambda o: -o[1])[:6]
        achieved = dom_fl / (dom_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.impl]
        # HBM traffic from the PMC passes (they cannot run inside this process): profiles/r02_pmc_traffic.json, per op of
        # one forward (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, separate passes, gfx950 correction of MI355X_MICROARCH.md)
        traffic = None
        level256 = None
        fwd_traffic = None
        # newest committed PMC file whose per-op names cover this run's schedule; a stale file (kernels or plan changed since
        # it was collected) is not mixed with fresh times: the counter-based fields are then null
        tsource = None
        by_name = {}
        kern_of = {}
        import glob
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            if args.impl != "mfma_bf16x3":
                break
            pj = json.load(open(tpath))
            per_op = pj.get("per_op_last_forward") or []
            if not pj.get("attribution"):
                continue  # (files older than round 4 zipped op names onto dispatches picked by a hand-kept regex: not used)
            by_name = {e["op"]: e["hbm_read_bytes"] + e["hbm_write_bytes"] for e in per_op}
            if not all(o[0] in by_name for o in ops if o[0] != "lr_branch"):
                continue
            # ... and the same kernel behind every op as in this run (the plan's launch log of the profiled forward above)
            mine = {}
            for op_, kn_ in engine.last_launch_log:
                kn_ = norm_kernel(kn_)
                if op_ and kn_ not in mine.setdefault(op_, []):
                    mine[op_].append(kn_)
            if any(" + ".join(mine.get(e["op"], [])) != e["kernel"] for e in per_op if e["op"] != "-" and e["op"] in mine):
                continue
            tsource = os.path.relpath(tpath, ROOT)
            dom_t = [by_name[o[0]] for o in dom]
            traffic = round(sum(dom_t) / len(dom_t))
            # the 256x256 level: conv0, conv_blocks.0, downs.0, decoder stage 2
            lvl = [o for o in ops if o[0] == "conv0" or o[0].startswith("conv_blocks.0.") or
                   o[0] in ("downs.0", "attention_gate.2", "ups.2.transform", "up_convs.2") or
                   o[0].startswith(("gating_signals.2", "attention_blocks.2", "up_convs.2."))]
            if lvl:
                lb = sum(by_name[o[0]] for o in lvl)
                lms = sum(o[1] for o in lvl)
                level256 = {"ops": [o[0] for o in lvl], "pmc_bytes": round(lb), "ms": round(lms, 4),
                            "GBs": round(lb / 1e9 / (lms * 1e-3), 1), "frac": round(lb / 1e9 / (lms * 1e-3) / HBM_PEAK_GBS, 4)}
            fb = sum(by_name[o[0]] for o in ops if o[0] != "lr_branch")
            fwd_traffic = {"pmc_bytes": round(fb), "GBs": round(fb / 1e9 / (all_ms * 1e-3), 1),
                           "frac": round(fb / 1e9 / (all_ms * 1e-3) / HBM_PEAK_GBS, 4)}
            break
        mfma_per_product = 3 if args.impl == "mfma_bf16x3" else 1
        # the family's kernels, named by the plan's own launch log of the profiled forward (op -> kernel: what tools/collect_pmc.py
        # joins the counter passes with); per-kernel averages for the cross-check against the rocprofv3 kernel trace
        by_kernel = None
        if args.impl == "mfma_bf16x3":
            import re
            for op_name, kname in getattr(engine, "last_launch_log", None) or []:
                if op_name:
                    kern_of.setdefault(op_name, re.sub(r"^void\s+", "", kname.replace("(anonymous namespace)::", "")).split("<")[0].split("(")[0].strip())
            groups = {}
            for o in dom:
                groups.setdefault(kern_of.get(o[0], "?"), []).append(o)
            by_kernel = {}
            for kname, sel in groups.items():
                ms = sum(o[1] for o in sel)
                by_kernel[kname] = {"launches": len(sel), "avg_launch_us": round(1e3 * ms / len(sel), 2),
                                    "achieved": round(sum(o[2] for o in sel) / (ms * 1e-3) / 1e12, 2)}
        # ---- the dominant kernel alone: the wave-specialised 3x3 tap-convolution (tapconv_fl_kernel where the FL arithmetic runs,
        # tapconv_sp_kernel otherwise): 11 launches per forward (conv1 / conv2 (+ shortcut) of blocks 1 - 3, ups.*.conv, the
        # att-halves of stages 0 / 1).  achieved = ALGORITHMIC FLOPs of those launches (2 * MACs of the reference
        # ops they compute, all of them executed) / their HIP-event durations on the launch stream.
        fl_on = os.environ.get("DRS_FL", "1") != "0" and args.impl == "mfma_bf16x3"
        dk = None
        if args.impl == "mfma_bf16x3" and by_kernel is not None:
            dk_ops = [o for o in dom if kern_of.get(o[0], "").startswith(("tapconv_fl_kernel", "tapconv_sp_kernel"))]
            if dk_ops:
                dk_ms = sum(o[1] for o in dk_ops)
                dk_fl = sum(o[2] for o in dk_ops)
                n_fl = sum(kern_of.get(o[0], "").startswith("tapconv_fl_kernel") for o in dk_ops)
                # matrix-pipe slots (16 cycles each) per product of one tap, tile and 32-channel chunk: 3 (split bf16) or
                # (9 fp16 + 5 fp6) / 9 (FL: the cross terms of a tap PAIR are one instruction)
                slots = (n_fl * (14.0 / 9.0) + (len(dk_ops) - n_fl) * 3.0) / len(dk_ops)
                dk = {"launches": len(dk_ops), "fl_launches": int(n_fl), "ms": dk_ms, "flops": dk_fl, "slots": slots,
                      "bytes": sum(o[3] for o in dk_ops),
                      "traffic": (round(sum(by_name[o[0]] for o in dk_ops) / len(dk_ops)) if tsource else None)}
        family = {"kernels": ("tapconv_fl / tapconv_sp (11) + upfuse_sp (2, executed FLOPs of the composite) + resblock0 (1) + the folded top "
                              "stage's conv3x3_proj_sp and upfuse_proj_sp (2: ALGORITHMIC FLOPs of the reference ops, about a tenth of "
                              "them executed)"),
                  "launches_per_forward": len(dom), "by_kernel": by_kernel, "achieved": round(achieved, 3),
                  "frac": round(achieved / peak, 5), "avg_launch_us": round(1e3 * dom_ms / len(dom), 2),
                  "traffic": traffic}
        if dk is not None:
            d_ach = dk["flops"] / (dk["ms"] * 1e-3) / 1e12
            roofline = {"bound": "mfma",
                        "kernel": ("tapconv_fl_kernel<HAS2> (conv_mfma_fl.hip): wave-specialised 3x3 stride-1 implicit GEMM over SP-format "
                                   "activations, 8 MFMA + 4 mover waves per CU, fp16 main product + block-scaled fp6 cross terms of tap "
                                   "pairs (the movers convert the window to the FL line on its way into LDS)" if dk["fl_launches"] else
                                   "tapconv_sp_kernel<HAS2, 64> (conv_mfma_sp.hip): wave-specialised 3x3 stride-1 implicit GEMM over "
                                   "SP-format activations, split bf16 x3"),
                        "launches_per_forward": dk["launches"], "fl_launches": dk["fl_launches"],
                        "achieved": round(d_ach, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(d_ach / peak, 5),
                        "flops_per_launch": round(dk["flops"] / dk["launches"]),
                        "avg_launch_us": round(1e3 * dk["ms"] / dk["launches"], 2),
                        "algorithmic_bytes_per_launch": round(dk["bytes"] / dk["launches"]),
                        "traffic": dk["traffic"],
                        "mfma_slots_per_product": round(dk["slots"], 3),
                        "mfma_pipe_frac": round(dk["slots"] * d_ach / peak, 5),
                        "note": ("all FLOPs of these launches are executed (no folded / composite op among them); mfma_pipe_frac = share "
                                 "of the matrix pipe's 16-cycle slots the kernel fills at the NOMINAL clock (peak 2.5 PFLOP/s dense "
                                 "bf16 / fp16; the fp6 instruction of the FL arithmetic occupies one such slot per tap pair)")}
        else:
            roofline = {"bound": "mfma", "kernel": "3x3 stride-1 family (%s)" % args.impl if args.impl != "direct" else "tapconv_direct_kernel",
                        "launches_per_forward": len(dom), "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 5), "flops_per_launch": round(dom_fl / len(dom)),
                        "avg_launch_us": round(1e3 * dom_ms / len(dom), 2), "algorithmic_bytes_per_launch": round(dom_by / len(dom)),
                        "traffic": traffic}
        roofline.update({
            "traffic_source": (tsource + " (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes, per op of one forward through the "
                                         "plan's launch log; op names AND the kernel behind every op match this run)") if tsource else None,
            "family_3x3": family,
            "forward_ms_sum_of_ops": round(all_ms, 4), "conv_ms": round(conv_ms, 4),
            "forward_hbm_GBs_algorithmic": round(step_mb / 1e3 / (all_ms * 1e-3), 1),
            "forward_hbm_frac_of_8TBs": round(step_mb / 1e3 / (all_ms * 1e-3) / HBM_PEAK_GBS, 5),
            "algorithmic_work_per_step": {"gflop": round(step_gflop, 2), "mb": round(step_mb, 1),
                                          "note": "reference graph (SURVEY 8(d)) minus the LR-conditioning branch, which a "
                                                  "sampling step reuses (computed once per chain)"},
            "forward_hbm_counter_based": fwd_traffic,
            "hbm_frac_256_level": level256["frac"] if level256 else None, "level_256": level256,
            "top_ops_ms": {n: round(ms, 4) for n, ms, _, _ in top}})
        dtype = DTYPE[args.impl]
        if fl_on and dk is not None and dk["fl_launches"]:
            dtype = ("f16 main + block-scaled fp6 (e2m3) cross terms of tap pairs on the wide 3x3 layers (FL, %d of %d launches of the "
                     "dominant kernel); split bf16 x3 (hi + lo) everywhere else; f32 accumulate" % (dk["fl_launches"], dk["launches"]))
        result = {
            "metric": "unet_denoise_steps_per_s", "value": round(value, 4), "unit": "batch16_steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: Residual_Attention_UNet_superres 128x128->256x256 mag=2 DownBlur, "
                                   "batch 16 per GPU, cosine T=1500, eval-mode UNet forward + ancestral update per step",
                       "timed_region": "`value` = the K steps of --steps (a short loop: 20 steps = ~25 ms); `full_chain` is the same "
                                       "step over a whole 1499-step chain in this process",
                       "batch_per_gpu": BATCH, "image_size": IMAGE, "noise_steps": T_STEPS, "impl": args.impl,
                       "weights": "seeded random (no pretrained weights exist)",
                       "process_group": torch.distributed.get_backend() if dist.is_initialized() else None},
            "image_steps_per_s": round(value * BATCH, 2),
            "tflops_algorithmic": round(value * step_gflop / 1e3, 3),
            "roofline": roofline,
        }
        if not args.no_extras:
            with torch.no_grad():
                q = psnr_vs_reference(model, dev, args.impl)
                if q is not None:
                    result["psnr_vs_ref_db"] = q["psnr_db"]
                    result["psnr_detail"] = q
                if world == 1:
                    # a full sample() of T - 1 = 1499 steps at configs[1] (16 images, one cached LR image):
                    # the end-to-end chain the reference's callers run, wall clock incl. the conditioning branch.
                    # `full_chain`: the seeded weights with the `output` projection scaled by 1e-2 (the G10 golden chain's
                    # weights): the chain's amplitude stays O(1), as a TRAINED model's does, and every forward runs the same
                    # kernels as the timed loop.  `full_chain_untrained_weights`: the plain seeded weights - the amplitude grows
                    # without bound (1e30 and beyond: meaningless), leaves fp16's range on the way, and the FL layers hand the
                    # chain over to the split-bf16 kernels at one of its periodic fault checks (DRS_ERR_RANGE): the time
                    # includes the <= 128 repeated steps; run LAST of the two, it leaves this plan on split bf16.
                    def chain(m_):
                        torch.cuda.synchronize()
                        t1_ = time.perf_counter()
                        out_ = diffusion.sample(BATCH, m_, lr_cpu[0], input_channels=3)
                        torch.cuda.synchronize()
                        dt_ = time.perf_counter() - t1_
                        m_.eval()
                        return {"steps": T_STEPS - 1, "seconds": round(dt_, 3), "batch16_steps_per_s": round((T_STEPS - 1) / dt_, 2),
                                "finite": bool(torch.isfinite(out_).all().item())}
                    bounded = Residual_Attention_UNet_superres(3, 3, dev)
                    sdb = dict(sd)
                    sdb["output.weight"] = sd["output.weight"] * 1e-2
                    sdb["output.bias"] = sd["output.bias"] * 1e-2
                    bounded.load_state_dict(sdb)
                    bounded = bounded.to(dev).eval()
                    bounded.hip_engine().set_impl(args.impl)
                    result["full_chain"] = dict(chain(bounded), weights="seeded, `output` x 1e-2 (bounded chain)")
                    del bounded
                    fc = chain(model)
                    fc["weights"] = "seeded (untrained: unbounded chain)"
                    with torch.no_grad():  # (the chain's plan: ONE conditioning image broadcast over the batch)
                        x.copy_(x_cpu)
                        lr1_ = lr_cpu[:1].to(dev).contiguous()
                        engine.forward(x, t.fill_(750), lr1_, MAG, reuse_cond=False)
                        _, log_ = engine.logged_forward(x, t, lr1_, MAG, reuse_cond=True, check_weights=False)
                    fc["handed_over_to_split_bf16"] = not any("tapconv_fl_kernel" in k_ for _, k_ in log_)
                    result["full_chain_untrained_weights"] = fc
                if args.impl != "mfma_f32":
                    # the exact-fp32 line beside the headline (same step, v_mfma_f32_16x16x4_f32 kernels): what the
                    # precision choice buys, in the driver's own record
                    engine.set_impl("mfma_f32")
                    x.copy_(x_cpu)
                    step(T_STEPS - 1, True)
                    step(T_STEPS - 2, False)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    nf = max(5, args.steps // 2)
                    for k in range(nf):
                        step(T_STEPS - 3 - k, False)
                    torch.cuda.synchronize()
                    ms32 = 1e3 * (time.perf_counter() - t1) / nf
                    engine.set_impl(args.impl)
                    result["fp32_line"] = {"impl": "mfma_f32", "dtype": "f32", "value": round(1e3 / ms32, 3),
                                           "unit": "batch16_steps/s", "ms_per_step": round(ms32, 4), "steps": nf,
                                           "tflops_algorithmic": round(GFLOP_PER_FWD / ms32, 2),
                                           "frac_of_157TF": round(GFLOP_PER_FWD / ms32 / PEAK_TFLOPS["mfma_f32"], 4)}
        if world == 1 and not args.no_extras:
            result["other_configs"] = other_configs_block(args.impl, dev)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd, x_cpu, torch.full((BATCH,), 750, dtype=torch.int64), lr_cpu)
        if os.environ.get("DRS_BENCH_OPS"):
            with open(os.environ["DRS_BENCH_OPS"], "w") as f:
                for n_, ms_, fl_, by_ in ops:
                    f.write(f"{n_:40s} {ms_*1e3:9.1f} us  {fl_/1e9:8.3f} GF  {by_/1e6:8.1f} MB  "
                            f"{(fl_/ms_/1e9 if ms_ > 0 else 0):8.1f} TF/s  {(by_/ms_/1e6 if ms_ > 0 else 0):8.1f} GB/s\n")
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        torch.distributed.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()

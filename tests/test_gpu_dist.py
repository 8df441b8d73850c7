"""The multi-GPU path on the 1-GPU lease: a fresh child process per case with backend "nccl" (= RCCL on ROCm), started
through torch.distributed.run exactly like the driver starts `bench.py --gpus N` (reference launch:
train_diffusion_superres.py:586,631-640,658).  The world-size-2 logic itself is covered on CPU by
tests/test_dist_gloo.py; this file proves that the RCCL communicator initialises and carries every collective of the
path on hardware."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, extra_env=None, timeout=600, nproc=1):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + script_args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("fake_world", [False, True])
def test_rccl_single_rank_train_and_sample(fake_world):
    """init_process_group("nccl") + broadcast_module + 2 x train_step(multiple_gpus=True) + rank-mean validation loss +
    sample_sharded.  `fake_world`: dist.world_size() reports 2, so that the broadcast / flat all-reduce / gather code
    actually issues RCCL collectives on the 1-rank communicator instead of short-circuiting."""
    out = _torchrun([os.path.join(ROOT, "tests", "_nccl_child.py")],
                    {"DRS_TEST_FAKE_WORLD": "1"} if fake_world else {"DRS_TEST_FAKE_WORLD": ""})
    line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["ok"] and res["params_changed"] >= 170 and abs(res["val"] - (1.5 if fake_world else 3.0)) < 1e-9


def test_bench_py_under_torchrun_initialises_rccl():
    """`bench.py --gpus 1` launched the way the driver launches N > 1: the process group is created (RANK is in the
    environment), the barrier / max-over-ranks timing path runs, one JSON line comes out."""
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                     "--no-extras"])
    line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["config"].get("process_group") == "nccl"


def test_bench_py_prints_exactly_one_line_on_stdout():
    """The driver's contract: rank 0 prints ONE JSON line.  Run with the extras the default invocation has (both 1499-step
    chains - the untrained one leaves fp16's range mid-chain and resumes on the split-bf16 kernels, with a notice that must
    go to stderr - and the other configs); only the CPU baseline is skipped."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:-1]
    res = json.loads(lines[0])
    assert res["value"] > 0 and res["roofline"]["frac"] > 0 and res["full_chain"]["finite"]
    assert res["full_chain_untrained_weights"]["finite"]


def test_two_ranks_share_one_gpu_train_step():
    """World size 2 on device memory: two processes on the one leased GPU (gloo process group: RCCL needs a device per rank),
    each running the HIP training step on its own shard - the flat in-place gradient exchange, FusedAdam behind it, identical
    parameters on both ranks afterwards (tests/_gloo2_gpu_child.py).  BASELINE configs[2]'s data-parallel step at reduced size."""
    out = _torchrun([os.path.join(ROOT, "tests", "_gloo2_gpu_child.py")], nproc=2)
    line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["ok"] and res["params_identical_across_ranks"] and res["grad_elements"] == 4383058 - 387520  # all parameters minus the 6 structurally unused tensors


@pytest.mark.parametrize("workload", ["train", "sample"])
def test_bench_py_two_ranks_end_to_end(workload):
    """`bench.py --gpus 2` end to end, launched the way the driver launches it, as two gloo ranks sharing the leased GPU
    (DRS_BENCH_BACKEND / DRS_BENCH_SHARE_DEVICE): process group, per-rank inputs, the data-parallel training step with its flat
    gradient exchange (train) / the sharded sampling step (the headline workload), barrier + max-over-ranks timing, ONE JSON
    line from rank 0 with n_gpus 2 and the whole-job rate.  (No 1 -> 8 scaling curve has been measured: no multi-GPU node was
    available to the builder; this proves the N > 1 code path is ready to be measured.)"""
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    if workload == "train":
        args += ["--workload", "train"]
    out = _torchrun(args, {"DRS_BENCH_BACKEND": "gloo", "DRS_BENCH_SHARE_DEVICE": "1"}, nproc=2, timeout=900)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["config"].get("process_group") == "gloo"

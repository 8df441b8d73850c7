"""Child of tests/test_gpu_dist.py::test_two_ranks_share_one_gpu_train_step: TWO ranks on the one leased GPU (RCCL needs a device
per rank, so the process group is gloo; the gradients it reduces are the HIP backward's flat device buffer all the same).
Each rank runs the real training step on its own shard (reference loop body train_diffusion_superres.py:378-401 under
DistributedDataParallel :658): HIP forward / backward -> ONE flat all-reduce -> FusedAdam.  Checks, on device memory at world
size 2: the reduced buffer is exactly the mean of the two ranks' local gradients, the `.grad`s are views of it, and both ranks
hold bit-identical parameters after two steps (rank-local BatchNorm statistics may differ, like the reference).  One JSON line
from rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as td  # noqa: E402


def main():
    from diffusionremotesensing_amd import dist, synthetic
    from diffusionremotesensing_amd import train_diffusion_superres as T
    from diffusionremotesensing_amd.optim import FusedAdam
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    dist.init_process_group("gloo")
    rank, world = dist.rank(), dist.world_size()
    assert world == 2
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), rank))  # different weights per rank ...
    m = m.to(dev).train()
    dist.broadcast_module(m)                                              # ... until rank 0's are broadcast (DDP's constructor)
    d = T.Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=50, device=dev, magnification_factor=2,
                    image_size=32, Degradation_type="DownBlur", multiple_gpus=True)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    loss_fn = torch.nn.MSELoss()
    hr = synthetic.tensor_uniform("g2.hr", (2, 3, 32, 32), seed=rank).to(dev)   # this rank's shard of the global batch
    lr = synthetic.tensor_uniform("g2.lr", (2, 3, 16, 16), seed=rank).to(dev)
    torch.manual_seed(100 + rank)       # per-rank timesteps and noise, like the reference's unseeded processes
    torch.cuda.manual_seed(200 + rank)
    real_allreduce = dist.allreduce_gradients
    local = {}

    def recording_allreduce(module, async_op=False):
        flat, n_grad, _, _ = module.hip_engine().last_gradient_buffer()
        local["grad"] = flat[:n_grad].clone()  # this rank's gradient, before the exchange
        return real_allreduce(module, async_op=async_op)

    T.drs_dist.allreduce_gradients = recording_allreduce
    worst = 0.0
    for _ in range(2):
        d.train_step(m, opt, loss_fn, lr, hr)
        flat, n_grad, entries, _ = m.hip_engine().last_gradient_buffer()
        both = [torch.empty_like(local["grad"]) for _ in range(2)]
        td.all_gather(both, local["grad"])
        want = (both[0] + both[1]) / 2  # SUM then divide, as the exchange does on gloo
        got = flat[:n_grad]
        worst = max(worst, float((got - want).abs().max() / want.abs().max()))
        assert torch.equal(got, want), "reduced buffer != mean of the ranks' local gradients"
        assert float((both[0] - both[1]).abs().max()) > 0, "the ranks' local gradients should differ (different shards)"
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        assert all(lo <= p.grad.data_ptr() < hi for p, _ in entries if p.grad is not None)
    params = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    both = [torch.empty_like(params) for _ in range(2)]
    td.all_gather(both, params)
    same = bool(torch.equal(both[0], both[1]))
    td.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"ok": True, "params_identical_across_ranks": same, "grad_elements": int(n_grad), "worst": worst}), flush=True)


if __name__ == "__main__":
    main()

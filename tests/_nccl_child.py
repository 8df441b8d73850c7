"""Child process of tests/test_gpu_dist.py: ONE rank with the "nccl" (= RCCL) backend on the 1-GPU lease, launched by
torch.distributed.run like the multi-GPU driver launches bench.py.  Exercises every collective call site of the
multi-GPU path (reference train_diffusion_superres.py:586,631-640,658,492-510): process-group init, parameter
broadcast, two train steps with the in-place flat-gradient all-reduce overlapped with the optimizer's table build, the
rank-synchronised validation loss, sharded sampling + gather.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    from diffusionremotesensing_amd import dist, synthetic
    from diffusionremotesensing_amd.optim import FusedAdam
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    dist.init_process_group("nccl")
    assert torch.distributed.get_backend() == "nccl" and dist.world_size() == int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
    m = m.to(dev).train()
    # world size 1 short-circuits in dist.py; force the collective code paths to run on the 1-rank communicator
    real_world_size = dist.world_size
    dist.world_size = lambda: 2 if os.environ.get("DRS_TEST_FAKE_WORLD") else torch.distributed.get_world_size()
    before = [p.detach().clone() for p in m.parameters()]
    dist.broadcast_module(m)
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, m.parameters()))
    d = Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=50, device=dev, magnification_factor=2,
                  image_size=32, Degradation_type="DownBlur", multiple_gpus=True)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    loss_fn = torch.nn.MSELoss()
    hr = synthetic.tensor_uniform("nccl.hr", (4, 3, 32, 32)).to(dev)
    lr = synthetic.tensor_uniform("nccl.lr", (4, 3, 16, 16)).to(dev)
    losses = []
    for _ in range(2):
        losses.append(d.train_step(m, opt, loss_fn, lr, hr).item())
        buf = m.hip_engine().last_gradient_buffer()
        grads = [p.grad for p in m.parameters() if p.grad is not None]
        lo, hi = buf[0].data_ptr(), buf[0].data_ptr() + buf[0].numel() * 4
        assert all(lo <= g.data_ptr() < hi for g in grads), "the .grads must be views of the reduced flat buffer"
    fake = bool(os.environ.get("DRS_TEST_FAKE_WORLD"))
    # with the fake world of 2 the single rank's SUM is divided by 2 (or averaged over 1 by RCCL): finite either way
    assert all(torch.isfinite(torch.tensor(l)) for l in losses)
    changed = sum(int(not torch.equal(a, b.detach())) for a, b in zip(before, m.parameters()))
    val = dist.allreduce_mean_scalar(3.0, dev)
    dist.world_size = real_world_size  # the gather below sizes its message list by the real communicator
    out = dist.sample_sharded(d, 2, m, lr[0], input_channels=3)
    assert out.shape == (2, 3, 32, 32) and torch.isfinite(out).all()
    torch.distributed.barrier()
    dist.destroy_process_group()
    print(json.dumps({"ok": True, "losses": losses, "params_changed": changed, "val": val, "fake_world": fake}), flush=True)


if __name__ == "__main__":
    main()

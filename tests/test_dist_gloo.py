"""world_size-2 tests of the single-node data-parallel plumbing (diffusionremotesensing_amd/dist.py) on the gloo
backend: sharded sampling (no collective on the data path), the flat gradient all-reduce that replaces DDP, and the
initial parameter broadcast.  The UNet itself is replaced by the CPU oracle as the checker's model (tests only)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from diffusionremotesensing_amd import dist
    dist.init_process_group("gloo")
    try:
        globals()[fn_name](rank, world, out_dir)
    finally:
        dist.destroy_process_group()


def _run(fn_name, tmp_path, world=2):
    mp.spawn(_worker, args=(world, _free_port(), fn_name, str(tmp_path)), nprocs=world, join=True)


# ---------------------------------------------------------------------------------------------------------------
def _case_shards_and_gather(rank, world, out_dir):
    from diffusionremotesensing_amd import dist
    assert dist.rank() == rank and dist.world_size() == world
    for n in (0, 1, 5, 16):
        ranges = [dist.shard_range(n, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert max(hi - lo for lo, hi in ranges) - min(hi - lo for lo, hi in ranges) <= 1
    lo, hi = dist.shard_range(5)
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 2, 3, 3).contiguous()
    full = dist.gather_shards(local, 5)
    assert full.shape == (5, 2, 3, 3) and torch.equal(full[:, 0, 0, 0], torch.arange(5.0))


def _case_allreduce_and_broadcast(rank, world, out_dir):
    from diffusionremotesensing_amd import dist
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    torch.manual_seed(100 + rank)  # different init per rank, like un-seeded reference processes
    m = Residual_Attention_UNet_superres(3, 3, "cpu")
    dist.broadcast_module(m)
    ref = [torch.zeros_like(p) for p in m.parameters()]
    for r, p in zip(ref, m.parameters()):
        r.copy_(p.detach())
        torch.distributed.broadcast(r, src=0)
        assert torch.equal(r, p.detach())  # every rank now holds rank 0's parameters
    dead = {f"{b}.conv_upsampled_lr_img.{w}" for b in ("conv_blocks.1", "conv_blocks.2", "bottle_neck")
            for w in ("weight", "bias")}
    for name, p in m.named_parameters():  # the 6 structurally unused tensors never get a gradient (quirk Q3)
        p.grad = None if name in dead else torch.full_like(p, float(rank + 1))
    n = dist.allreduce_gradients(m)
    assert n == 4383058  # every requires_grad parameter has a slot (rank-invariant message); dead ones stay None
    for name, p in m.named_parameters():
        if name in dead:
            assert p.grad is None
        else:
            assert torch.all(p.grad == (1 + world) / 2)  # mean over ranks of (rank + 1)


def _case_rank_variant_gradients(rank, world, out_dir):
    """A parameter without a gradient on ONE rank only (the class-conditional trainer drops its label per rank): the
    message must have the same size on every rank and every rank must end up with the same averaged gradient, like
    DDP(find_unused_parameters=True).  Also the engine path: one flat buffer reduced in place, flags in its tail."""
    from diffusionremotesensing_amd import dist
    from diffusionremotesensing_amd.optim import FusedAdam  # noqa: F401  (importable without a device)
    lin = torch.nn.Linear(4, 3)
    emb = torch.nn.Embedding(5, 4)
    never = torch.nn.Linear(2, 2)
    m = torch.nn.ModuleList([lin, emb, never])
    for p in lin.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    emb.weight.grad = torch.full_like(emb.weight, 4.0) if rank == 1 else None  # rank 0 dropped its label
    n = dist.allreduce_gradients(m)
    assert n == sum(p.numel() for p in m.parameters())
    assert all(torch.all(p.grad == 1.5) for p in lin.parameters())
    assert emb.weight.grad is not None and torch.all(emb.weight.grad == 2.0)  # (0 + 4) / 2 on BOTH ranks
    assert all(p.grad is None for p in never.parameters())  # unused everywhere: stays None everywhere
    # asynchronous handle
    for p in lin.parameters():
        p.grad = torch.full_like(p, float(10 * (rank + 1)))
    emb.weight.grad = None
    pending = dist.allreduce_gradients(m, async_op=True)
    pending.wait()
    pending.wait()  # idempotent
    assert all(torch.all(p.grad == 15.0) for p in lin.parameters()) and emb.weight.grad is None

    # engine path: the backward's flat buffer is reduced in place; .grad tensors are views of it
    class _Eng:
        def __init__(self, params, has_flags):
            total = sum(p.numel() for p in params)
            self.flat = torch.zeros(total + (len(params) if has_flags else 0))
            self.entries, off = [], 0
            for p in params:
                self.entries.append((p, self.flat[off:off + p.numel()].view_as(p)))
                off += p.numel()
            self.total, self.has_flags = total, has_flags

        def last_gradient_buffer(self):
            return self.flat, self.total, self.entries, self.has_flags
    for has_flags in (False, True):
        holder = torch.nn.ModuleList([lin, emb])
        eng = _Eng(list(holder.parameters()), has_flags)
        holder.__dict__["_hip_engine"] = eng
        for p, view in eng.entries:
            view.fill_(float(rank + 1))
            p.grad = view
        if has_flags and rank == 0:
            eng.entries[-1][1].zero_()  # the kernels zero an unused parameter's slot
            emb.weight.grad = None
        ptr = eng.flat.data_ptr()
        dist.allreduce_gradients(holder)
        assert eng.flat.data_ptr() == ptr
        for p in lin.parameters():
            assert torch.all(p.grad == 1.5) and p.grad.data_ptr() in [v.data_ptr() for _, v in eng.entries]
        assert torch.all(emb.weight.grad == (1.0 if has_flags else 1.5))  # flags: (0 + 2) / 2 on both ranks
    assert abs(dist.allreduce_mean_scalar(float(rank)) - 0.5) < 1e-12


class _OracleDiffusion:
    """Stand-in with Diffusion's sampling surface, computed by the CPU oracle (checker, not product)."""

    def __init__(self, sd, T, image_size):
        from oracle import diffusion_oracle as D
        from oracle import unet_oracle as U
        self.D, self.model = D, U.OracleUNet(sd)
        self.alpha, self.alpha_hat, self.beta = D.schedule("cosine", T)
        self.noise_steps, self.image_size, self.device = T, image_size, "cpu"

    def sample(self, n, model, lr_img, input_channels=3, generate_video=False, noise_source=None):
        return self.D.sample(self.model, n, lr_img, self.noise_steps, self.alpha, self.alpha_hat, self.beta, 2,
                             self.image_size, input_channels, noise_source=noise_source)


def _global_noise(i, shape):
    g = torch.Generator().manual_seed(1000 + i)
    return torch.randn(shape, generator=g)


def _case_sharded_sampling(rank, world, out_dir):
    from diffusionremotesensing_amd import dist, synthetic
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    sd = synthetic.seeded_state_dict(Residual_Attention_UNet_superres(3, 3, "cpu").state_dict(), 0)
    diff = _OracleDiffusion(sd, T=4, image_size=16)
    lr = synthetic.tensor_uniform("dist.lr", (3, 8, 8))
    full = dist.sample_sharded(diff, 3, None, lr, noise_source=_global_noise)  # 3 chains over 2 ranks: 2 + 1
    assert full.shape == (3, 3, 16, 16)
    torch.save(full, os.path.join(out_dir, f"sharded_{rank}.pt"))
    local = dist.sample_sharded(diff, 3, None, lr, gather=False, noise_source=_global_noise)
    lo, hi = dist.shard_range(3)
    assert torch.equal(local, full[lo:hi])


def _tile_noise(tile, i, shape):
    g = torch.Generator().manual_seed(7000 + 100 * tile + i)
    return torch.randn(shape, generator=g)


def _case_sharded_tiles(rank, world, out_dir):
    """Aggregation tiler: the tiles of one image are sharded over the ranks and gathered once."""
    from diffusionremotesensing_amd import dist, synthetic
    from diffusionremotesensing_amd.Aggregation_Sampling import split_aggregation_sampling
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    sd = synthetic.seeded_state_dict(Residual_Attention_UNet_superres(3, 3, "cpu").state_dict(), 0)
    diff = _OracleDiffusion(sd, T=3, image_size=16)
    img = synthetic.tensor_uniform("dist.img", (1, 3, 16, 24))
    tiler = split_aggregation_sampling(img, 8, 8, 2, diff, "cpu")
    assert len(tiler.patches_lr) == 6  # 2 x 3 tiles -> 3 per rank
    tiles = tiler.sample_tiles(noise_source=_tile_noise)
    assert tiles.shape == (6, 3, 16, 16)
    torch.save(tiles, os.path.join(out_dir, f"tiles_{rank}.pt"))
    tiler.tile_batch = 2  # this rank's 3 tiles as chunks of 2 + 1 (the last one padded to the chunk size)
    chunked = tiler.sample_tiles(noise_source=_tile_noise)
    assert torch.allclose(chunked, tiles, rtol=0, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------
def test_sharded_tiles_match_sequential_reference_loop(tmp_path):
    """2 ranks x 3 tiles (batched) == the reference's sequential n=1 loop over the 6 tiles (same per-tile noise)."""
    _run("_case_sharded_tiles", tmp_path)
    sys.path.insert(0, ROOT)
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    from oracle import aggregation_oracle as A
    sd = synthetic.seeded_state_dict(Residual_Attention_UNet_superres(3, 3, "cpu").state_dict(), 0)
    diff = _OracleDiffusion(sd, T=3, image_size=16)
    img = synthetic.tensor_uniform("dist.img", (1, 3, 16, 24))
    infos, lr_origins = A.tile_infos(16, 24, 8, 8, 2)
    seq = torch.cat([diff.sample(1, None, img[0, :, y:y + 8, x:x + 8],
                                 noise_source=lambda i, shape, ti=ti: _tile_noise(ti, i, shape))
                     for ti, (y, x) in enumerate(lr_origins)])
    a = torch.load(os.path.join(tmp_path, "tiles_0.pt"))
    b = torch.load(os.path.join(tmp_path, "tiles_1.pt"))
    assert torch.equal(a, b)
    assert torch.allclose(a, seq, rtol=0, atol=1e-5)


def test_shards_and_gather(tmp_path):
    _run("_case_shards_and_gather", tmp_path)


def test_flat_allreduce_and_broadcast(tmp_path):
    _run("_case_allreduce_and_broadcast", tmp_path)


def test_rank_variant_gradient_sets(tmp_path):
    _run("_case_rank_variant_gradients", tmp_path)


def test_sharded_sampling_matches_single_process(tmp_path):
    """The n reverse chains are independent: 2 ranks x shards == 1 process x all (identical noise per chain)."""
    _run("_case_sharded_sampling", tmp_path)
    sys.path.insert(0, ROOT)
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    sd = synthetic.seeded_state_dict(Residual_Attention_UNet_superres(3, 3, "cpu").state_dict(), 0)
    diff = _OracleDiffusion(sd, T=4, image_size=16)
    lr = synthetic.tensor_uniform("dist.lr", (3, 8, 8))
    single = diff.sample(3, None, lr, noise_source=_global_noise)
    a = torch.load(os.path.join(tmp_path, "sharded_0.pt"))
    b = torch.load(os.path.join(tmp_path, "sharded_1.pt"))
    assert torch.equal(a, b)
    assert torch.allclose(a, single, rtol=0, atol=1e-5)


def test_single_process_defaults():
    from diffusionremotesensing_amd import dist
    assert dist.rank() == 0 and dist.world_size() == 1 and dist.shard_range(7) == (0, 7)
    t = torch.ones(2, 3)
    assert dist.gather_shards(t, 2) is t

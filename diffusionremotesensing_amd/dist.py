"""Single-node multi-GPU plumbing: one process per GPU, `torch.distributed` over RCCL/xGMI
("nccl" backend on ROCm); `gloo` when no ROCm device is visible (CPU tests).

Replaces the reference's DistributedDataParallel(find_unused_parameters=True) +
DistributedSampler wiring (train_diffusion_superres.py:586,631-640,658):
  * sampling shards the n independent reverse chains over ranks, no collective on the data path;
  * training exchanges ONE flat fp32 gradient vector per step (16 MB for the UNet's live parameters), reduced IN PLACE
    in the buffer the backward wrote (the `.grad`s are views of it).  xGMI is point-to-point, so a single large
    all-reduce beats DDP's 1 MiB + 25 MiB buckets; the 6 parameter tensors that never receive a gradient (SURVEY.md
    quirk Q3) are not part of the plan and have no slot; parameters that MAY go unused on some ranks (label embedding)
    carry a "used" flag in the same message, which replaces DDP's 176-entry bitmap all-reduce.
BatchNorm statistics stay local to each rank, like the reference (no SyncBatchNorm).
"""
import os

import torch
import torch.distributed as td


def is_initialized():
    return td.is_available() and td.is_initialized()


def rank():
    return td.get_rank() if is_initialized() else 0


def world_size():
    return td.get_world_size() if is_initialized() else 1


def init_process_group(backend=None):
    """env:// rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK from torchrun)."""
    if is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    td.init_process_group(backend=backend)


def destroy_process_group():
    if is_initialized():
        td.destroy_process_group()


def shard_range(n, r=None, w=None):
    """Contiguous [lo, hi) slice of n independent items owned by rank r of w (remainder to the low ranks)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def _flat(tensors):
    return torch.cat([t.reshape(-1) for t in tensors]) if tensors else None


def _unflat_into(flat, tensors):
    views, off = [], 0
    for t in tensors:
        n = t.numel()
        views.append(flat[off:off + n].view_as(t))
        off += n
    torch._foreach_copy_(tensors, views)


def broadcast_module(module, src=0):
    """Rank `src`'s parameters and buffers to every rank, as two flat messages (fp32 / int64)."""
    if world_size() == 1:
        return
    with torch.no_grad():
        tensors = [p.data for p in module.parameters()] + [b for b in module.buffers()]
        # same dtype order on every rank (a set of dtypes iterates in a per-process order -> mismatched collectives)
        for dtype in sorted({t.dtype for t in tensors}, key=str):
            group = [t for t in tensors if t.dtype == dtype]
            flat = _flat(group)
            td.broadcast(flat, src=src)
            _unflat_into(flat, group)


class _PendingReduce:
    """Handle of one gradient exchange.  `wait()` makes the CURRENT stream wait for the collective, applies the mean and
    hands every rank the same set of `.grad`s (see `allreduce_gradients`).  Idempotent."""

    def __init__(self, work, flat, n_grad, slots, world, need_div):
        self.work, self.flat, self.n_grad, self.slots, self.world, self.need_div = work, flat, n_grad, slots, world, need_div
        self.done = False
        self.created_grads = []  # parameters whose .grad this wait() created (unused on this rank, used on another)

    def wait(self):
        if self.done:
            return self.n_grad
        self.done = True
        if self.work is not None:
            self.work.wait()  # RCCL: the current stream waits for the communicator's stream; gloo: blocks the host
        if self.need_div:
            self.flat[: self.n_grad].div_(self.world)
        tail = self.flat[self.n_grad:]
        if tail.numel():  # some parameter can be unused on SOME ranks: who used what, summed over ranks
            used = tail.tolist()
            for (p, view, copy_back), u in zip(self.slots, used):
                if u <= 0:
                    continue  # unused everywhere: .grad stays None on every rank, Adam skips it everywhere (like DDP)
                if p.grad is None:
                    p.grad = view  # unused here, used elsewhere: the averaged gradient, so that Adam steps identically
                    self.created_grads.append(p)
                elif copy_back:
                    p.grad.copy_(view)
        else:
            for p, view, copy_back in self.slots:
                if copy_back:
                    p.grad.copy_(view)
        return self.n_grad


def allreduce_gradients(module, async_op=False):
    """Mean of the gradients over ranks through ONE flat all-reduce, replacing DDP's bucketed reducer
    (reference train_diffusion_superres.py:658, `find_unused_parameters=True`).

    The reduced set is RANK-INVARIANT: every parameter that can receive a gradient has a slot whether or not this rank
    produced one (the class-conditional trainer drops its label per rank at random, so `label_emb.weight.grad` is None on
    some ranks only).  Like DDP, a parameter used on ANY rank ends up with the same averaged gradient on EVERY rank
    (ranks that did not use it contribute zeros); a parameter used on no rank keeps `.grad = None` everywhere.
      * HIP engine path: `engine.backward` already wrote all gradients into ONE flat buffer that the `.grad`s are views
        of; it is reduced in place (no gather / scatter copies).  Its tail carries one "used" flag per parameter when
        the model has parameters that may go unused (label embedding); models without such parameters skip the flags
        and the host read-back they need.
      * generic path (any nn.Module; the gloo tests): flat buffer built over all `requires_grad` parameters.
    `async_op=True` returns a handle whose `wait()` must be called before the optimizer's kernels are enqueued; the
    collective then overlaps the host-side work in between (optimizer table build)."""
    w = world_size()
    if w == 1:
        return _PendingReduce(None, torch.empty(0), 0, [], 1, False) if async_op else 0
    slots = None
    eng = module.__dict__.get("_hip_engine") if hasattr(module, "__dict__") else None
    buf = eng.last_gradient_buffer() if eng is not None else None
    if buf is not None:
        flat, n_grad, entries, has_flags = buf
        slots = []
        aliased = False
        for p, view in entries:
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                slots = None  # gradients were accumulated elsewhere: take the generic path
                break
            aliased = aliased or p.grad is not None
            slots.append((p, view, False))
        if slots is not None and not aliased:
            slots = None  # no gradient of THIS step lives in the engine's buffer (it is the previous backward's): generic path
        if slots is not None and has_flags:
            flags = torch.tensor([0.0 if p.grad is None else 1.0 for p, _ in entries], dtype=torch.float32)
            flat[n_grad:].copy_(flags.pin_memory() if flat.is_cuda else flags, non_blocking=True)
    if slots is None:  # generic: gather into a fresh flat buffer, zero where this rank has no gradient
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            return _PendingReduce(None, torch.empty(0), 0, [], w, False) if async_op else 0
        n_grad = sum(p.numel() for p in params)
        dev = next((p.grad.device for p in params if p.grad is not None), params[0].device)
        flat = torch.zeros(n_grad + len(params), dtype=torch.float32, device=dev)
        slots, off = [], 0
        for i, p in enumerate(params):
            view = flat[off:off + p.numel()].view_as(p)
            if p.grad is not None:
                view.copy_(p.grad)
                flat[n_grad + i] = 1.0
            slots.append((p, view, p.grad is not None))
            off += p.numel()
    backend = td.get_backend()
    use_avg = backend == "nccl" and flat.numel() == n_grad  # RCCL averages in the collective; flags must stay sums
    work = td.all_reduce(flat, op=td.ReduceOp.AVG if use_avg else td.ReduceOp.SUM, async_op=True)
    pending = _PendingReduce(work, flat, n_grad, slots, w, not use_avg)
    if async_op:
        return pending
    return pending.wait()


def allreduce_mean_scalar(value, device=None):
    """Mean over ranks of a host scalar (validation loss): every rank then takes the same best-loss / early-stopping
    branch and leaves the training loop together (the reference decides per rank, train_diffusion_superres.py:492-510:
    with a collective inside the loop that is a deadlock as soon as one rank stops)."""
    w = world_size()
    if w == 1:
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t.item()) / w


def gather_shards(local, n_total):
    """Concatenate per-rank shards (dim 0, sizes from shard_range) on every rank."""
    w = world_size()
    if w == 1:
        return local
    sizes = [shard_range(n_total, r, w) for r in range(w)]
    biggest = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(w)]
    td.all_gather(out, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def sample_sharded(diffusion, n, model, lr_img, input_channels=3, gather=True, noise_source=None):
    """`Diffusion.sample` with the n chains split across ranks; no collective until the optional final gather."""
    lo, hi = shard_range(n)
    if hi > lo:
        src = None
        if noise_source is not None:
            def src(i, shape):  # slice the global noise so results do not depend on the world size
                return noise_source(i, (n,) + tuple(shape[1:]))[lo:hi]
        local = diffusion.sample(hi - lo, model, lr_img, input_channels=input_channels, noise_source=src)
    else:
        local = torch.empty((0, input_channels, diffusion.image_size, diffusion.image_size), dtype=torch.float32,
                            device=lr_img.device if lr_img.is_cuda else diffusion.device)
    return gather_shards(local, n) if gather else local

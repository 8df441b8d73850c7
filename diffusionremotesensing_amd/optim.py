"""`FusedAdam`: torch.optim.Adam's update (defaults of the reference: `Adam(model.parameters(), lr=lr)`,
train_diffusion_superres.py:337) as ONE multi-tensor HIP launch per step (`drs_adam_multi`) instead of the ~45
foreach kernels torch issues for the UNet's 176 parameter tensors.

Same `state_dict()` layout as torch.optim.Adam (`step`, `exp_avg`, `exp_avg_sq` per parameter), same treatment of
parameters without a gradient (skipped, state untouched).  Parameters must be fp32 tensors on a ROCm device.
"""
import ctypes as C

import torch

from . import _lib


class _AdamTensor(C.Structure):  # include/drs_hip.h: drs_adam_tensor
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64),
                ("step", C.c_int64)]


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("FusedAdam: invalid hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None, grad_ready=None):
        """`grad_ready`, when given, is called after the host-side table build and right before the update kernel is
        enqueued: the multi-GPU step passes the wait of its asynchronous gradient all-reduce, so the collective overlaps
        the table build (`dist.allreduce_gradients(async_op=True)`).  Gradients that are None at call time but may be
        produced by that wait (parameters unused on this rank only) are resolved first."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        owner = getattr(grad_ready, "__self__", None) if grad_ready is not None else None
        has_flag_tail = owner is not None and hasattr(owner, "flat") and owner.flat.numel() > getattr(owner, "n_grad", 0)
        if grad_ready is not None and (has_flag_tail or any(p.grad is None and getattr(p, "_drs_maybe_unused", False)
                                                            for g in self.param_groups for p in g["params"])):
            # which gradients exist is only known after the exchange: whenever it carries "used" flags (some parameter may be
            # unused on some rank) it is waited for BEFORE any table is built - on every rank alike, so no rank can end up
            # with a gradient that has no row (the condition _late_wait still checks, now unreachable through dist.py)
            grad_ready()
            grad_ready = None
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            if not params:
                continue
            dev = params[0].device
            rows, live = [], False
            for p in params:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous fp32 tensors on a ROCm device")
                if p.grad is None:  # no state is created for a parameter that never gets a gradient (like torch)
                    rows.append((p.data_ptr(), 0, 0, 0, p.numel(), 0))
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                s = int(st["step"].item())  # host tensor: no device sync
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                    p.grad = g
                rows.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), s))
                live = True
            if not live:
                continue
            # pointer table: the gradients move every step (fresh flat buffer per backward), so the table is uploaded
            # each step from a small ring of pinned buffers; a slot is reused only after its copy has executed
            n = len(rows)
            ring = self._tables.get(gi)
            if ring is None or ring["dev"].numel() != n * 6:
                ring = {"host": [torch.empty(n * 6, dtype=torch.int64).pin_memory() for _ in range(4)],
                        "event": [None] * 4, "dev": torch.empty(n * 6, dtype=torch.int64, device=dev), "k": 0}
                self._tables[gi] = ring
            k = ring["k"]
            ring["k"] = (k + 1) % 4
            if ring["event"][k] is not None:
                ring["event"][k].synchronize()
            host, devt = ring["host"][k], ring["dev"]
            host.copy_(torch.tensor(rows, dtype=torch.int64).view(-1))
            devt.copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            ring["event"][k] = ev
            b1, b2 = group["betas"]
            if grad_ready is not None:
                self._late_wait(grad_ready)
                grad_ready = None
            with torch.cuda.device(dev):
                st = lib.drs_adam_multi(C.c_void_p(devt.data_ptr()), n, max(r[4] for r in rows), float(group["lr"]),
                                        float(b1), float(b2), float(group["eps"]),
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(st, "drs_adam_multi")
            # the kernel wrote the parameters through raw pointers: advance their version counters (no kernel launch) so
            # that everything keyed on `_version` (the engine's packed-weight cache) sees the update
            torch.autograd.graph.increment_version([p for p in params if p.grad is not None])
        if grad_ready is not None:
            self._late_wait(grad_ready)
        return loss

    @staticmethod
    def _late_wait(grad_ready):
        """The exchange's wait, AFTER the pointer table was built.  A gradient that the wait itself creates (parameter unused
        on this rank, used on another: dist._PendingReduce.created_grads) has no row in that table: this rank would skip the
        update the other ranks apply and the replicas would drift apart silently.  Parameters that can be unused per rank
        must carry `_drs_maybe_unused` (then the wait runs first); anything else is reported, not ignored."""
        grad_ready()
        owner = getattr(grad_ready, "__self__", None)
        created = getattr(owner, "created_grads", None)
        if created:
            raise RuntimeError(f"FusedAdam: the gradient exchange created .grad for {len(created)} parameter(s) after the "
                               "update table was built; mark parameters that may be unused on some ranks with "
                               "`p._drs_maybe_unused = True` so that the exchange is waited for first")

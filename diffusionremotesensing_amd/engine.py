"""Host side of the UNet plan: owns device buffers (through torch), keeps the packed weights in
sync with the module's parameters and calls `drs_unet_forward` (include/drs_hip.h).

One `_Plan` per (batch, lr_batch, H, W, magnification, device); buffers are sized for the
288 GB of an MI355X: every intermediate activation keeps its own slot in one workspace
(about 1.6 GB at batch 16, 256x256), nothing is re-allocated between calls.
"""
import collections
import ctypes as C
import os
import weakref

import torch

from . import _lib
from .hip_ops import inv_freq_table

# shipped default: implicit GEMM on bf16 MFMA with hi+lo operand split (3 MFMAs per product, ~1e-5 relative error).
# "mfma_f32" / "direct" are the exact-fp32 parity anchors; "mfma_f16" is faster but measured 1.1e-3 max-rel > 1e-3 bar.
DEFAULT_IMPL = os.environ.get("DRS_IMPL", "mfma_bf16x3")


class _Plan:
    def __init__(self, lib, cfg, device):
        self.lib = lib
        self.device = device
        self.handle = C.c_void_p()
        _lib.check(lib.drs_unet_plan_create(C.byref(self.handle), C.byref(cfg)), "drs_unet_plan_create")
        self.cfg = cfg
        n = lib.drs_unet_num_params(self.handle)
        self.param_names = [lib.drs_unet_param_name(self.handle, i).decode() for i in range(n)]
        self.param_numels = [lib.drs_unet_param_numel(self.handle, i) for i in range(n)]
        self.packed_bytes = lib.drs_unet_packed_bytes(self.handle)
        self.ws_bytes = lib.drs_unet_workspace_bytes(self.handle)
        self.packed = torch.empty(self.packed_bytes, dtype=torch.uint8, device=device)
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.signature = None
        self.cond_key = None
        nt = lib.drs_unet_num_tensors(self.handle)
        self.tensor_index = {lib.drs_unet_tensor_name(self.handle, i).decode(): i for i in range(nt)}

    def __del__(self):
        try:
            if self.handle:
                self.lib.drs_unet_plan_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass


class HipUNetEngine:
    """Runs reference `Residual_Attention_UNet_superres.forward` (UNet_model_superres.py:337-379)
    on the HIP plan.  Eval mode folds BatchNorm; train mode uses batch statistics and updates the running statistics
    like nn.BatchNorm2d, and with autograd enabled the call is recorded so that loss.backward() runs
    drs_unet_backward (weight / bias / BatchNorm-affine gradients of every live parameter)."""

    VARIANTS = {"superres": _lib.VARIANT_SUPERRES, "sar_to_ndvi": _lib.VARIANT_SAR_TO_NDVI,
                "generation": _lib.VARIANT_GENERATION}

    def __init__(self, module, variant="superres", impl=None):
        if variant not in self.VARIANTS:
            raise NotImplementedError(f"UNet variant {variant!r} is not built")
        self.variant = variant
        self._module = weakref.ref(module)
        self.impl = _lib.IMPL_BY_NAME[impl or DEFAULT_IMPL]
        # training steps run the exact-fp32 MFMA kernels by default: gradients pass through ~25 BatchNorm backward
        # cancellations and the split-bf16 rounding (1e-5 per op) grows to ~4e-3 on the deepest (LR encoder) gradients
        self.train_impl = _lib.IMPL_BY_NAME[os.environ.get("DRS_TRAIN_IMPL", "mfma_f32")]
        if self.train_impl == _lib.IMPL_MFMA_BF16X3:
            import warnings
            warnings.warn("DRS_TRAIN_IMPL=mfma_bf16x3: split-bf16 training is outside the gradient bar (worst gradient-norm "
                          "deviation 1.26e-3 on the full-size configs[2] step against 1e-3); the exact-fp32 MFMA kernels are "
                          "the parity-grade training path")
        # least-recently-used cache of plans: every distinct (batch, lr batch, H, W, mag, impl, train) owns a workspace
        # (~1.6 GB at batch 16, 256x256); train + validation + preview + tiler shapes would otherwise pile up
        self._plans = collections.OrderedDict()
        self.max_plans = int(os.environ.get("DRS_MAX_PLANS", "6"))
        self._grad_buffer = None
        self._bn_epoch = 0  # bumped by every train-mode forward (running statistics change under the eval plans)
        self.keep_intermediates = False  # True: every block output stays readable (read_tensor), used by parity tests
        self._inv_freq = inv_freq_table(module.time_emb_dim)
        self._inv_freq_c = (C.c_float * self._inv_freq.numel())(*self._inv_freq.tolist())

    # -- state_dict key -> live tensor, without building a state_dict per call ----------------------------------
    def _tensors(self, names):
        """{key: parameter / buffer} for the plan's keys.  `module.state_dict()` walks all 60 sub-modules and 299 keys
        (~0.4 ms, several times per training step); the (owning module, attribute) pair of every key is resolved once
        and the tensors are fetched with getattr (they are looked up, not cached: `.to()` / `load_state_dict` may
        replace or rewrite them)."""
        cache = self.__dict__.setdefault("_key_owner", {})
        m = self._module()
        out = {}
        for n in names:
            ent = cache.get(n)
            if ent is None:
                path, _, leaf = n.rpartition(".")
                owner = m.get_submodule(path) if path else m
                ent = cache[n] = (owner, leaf)
            owner, leaf = ent
            t = owner._parameters.get(leaf)
            if t is None:
                t = owner._buffers.get(leaf)
            if t is None:
                raise RuntimeError(f"state_dict key {n} is missing from the module")
            out[n] = t
        return out

    # -- plan / weights -------------------------------------------------------------------
    def _get_plan(self, B, Bl, H, W, mag, device, train=False):
        impl = self.train_impl if train else self.impl
        key = (B, Bl, H, W, mag, device.index, impl, self.keep_intermediates, train)
        plan = self._plans.get(key)
        if plan is not None:
            self._plans.move_to_end(key)
        if plan is None:
            m = self._module()
            if self.variant == "sar_to_ndvi":  # UNet_model_SAR_TO_NDVI.py:264-267: x has NDVI_channels, cond SAR_channels
                cx, cout, cc, ncls = m.NDVI_channels, m.NDVI_channels, m.SAR_channels, 0
            elif self.variant == "generation":
                cx, cout, cc, ncls = m.image_channels, m.out_dim, 0, int(m.num_classes or 0)
            else:
                cx, cout, cc, ncls = m.image_channels, m.out_dim, m.image_channels, 0
            cfg = _lib.UNetConfig(B, Bl, cx, cout, H, W, mag, impl, 1e-5,
                                  (_lib.PLAN_KEEP_ALL if self.keep_intermediates else 0) |
                                  (_lib.PLAN_TRAIN if train else 0), self.VARIANTS[self.variant], cc, ncls)
            self._channels = (cx, cout, cc)
            plan = _Plan(_lib.load(), cfg, device)
            plan.train = train
            self._plans[key] = plan
            while len(self._plans) > max(self.max_plans, 1):
                self._plans.popitem(last=False)  # its workspace is freed once no autograd node / caller holds the plan
        return plan

    def set_impl(self, impl, train_impl=None):
        self.impl = _lib.IMPL_BY_NAME[impl]
        if train_impl is not None:
            self.train_impl = _lib.IMPL_BY_NAME[train_impl]

    def _sync_weights(self, plan):
        """Re-pack (BatchNorm fold + re-layout) when any parameter or buffer changed."""
        sd = self._tensors(plan.param_names)
        tensors = []
        for name, numel in zip(plan.param_names, plan.param_numels):
            t = sd[name]
            if t.device != plan.device or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError(f"parameter {name} must be a contiguous fp32 tensor on {plan.device}, "
                                   f"found {t.dtype} on {t.device}")
            if t.numel() != numel:
                raise RuntimeError(f"parameter {name} has {t.numel()} elements, plan expects {numel}")
            tensors.append(t)
        # `_drs_param_epoch`: bumped by updates that write the parameters through the C-ABI (multi-tensor EMA), which torch's
        # version counters do not see
        sig = tuple((t.data_ptr(), t._version) for t in tensors) + (self._bn_epoch, getattr(self._module(), "_drs_param_epoch", 0))
        if sig == plan.signature:
            return
        arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        stream = C.c_void_p(torch.cuda.current_stream(plan.device).cuda_stream)
        st = plan.lib.drs_unet_pack_weights(plan.handle, arr, self._inv_freq_c, C.c_void_p(plan.packed.data_ptr()),
                                            plan.packed_bytes, stream)
        _lib.check(st, "drs_unet_pack_weights")
        plan.signature = sig
        plan.cond_key = None

    # -- forward ----------------------------------------------------------------------------
    def forward(self, x, timestep, lr_img, magnification_factor, reuse_cond=False, check_weights=True,
                _in_autograd_fn=False, labels=None):
        """`lr_img` is the conditioning image (LR image / SAR image; None for the generation variant);
        `labels` the optional class labels of the generation variant."""
        m = self._module()
        train = bool(m.training)
        has_cond = self.variant != "generation"
        if not has_cond:
            lr_img, magnification_factor = None, 1
        elif not isinstance(lr_img, torch.Tensor):
            raise RuntimeError("the conditioning image must be a tensor on a ROCm device")
        if train and torch.is_grad_enabled() and not _in_autograd_fn and any(p.requires_grad for p in m.parameters()):
            # training step: route through autograd so loss.backward() reaches drs_unet_backward
            plan = self._get_plan(x.shape[0], lr_img.shape[0] if has_cond else x.shape[0], x.shape[2], x.shape[3],
                                  int(magnification_factor), x.device, True)
            sd = self._tensors(plan.param_names)
            params = [sd[n] for n in plan.param_names if sd[n].requires_grad]
            return _UNetTrainFn.apply(self, x, timestep, lr_img, magnification_factor, labels, *params)
        named = [("x", x), ("timestep", timestep)] + ([("lr_img", lr_img)] if has_cond else []) + \
                ([("labels", labels)] if labels is not None else [])
        for name, t in named:
            if not isinstance(t, torch.Tensor) or not t.is_cuda:
                raise RuntimeError(f"{name} must be a tensor on a ROCm device: the UNet forward has no CPU fallback")
        if x.dtype != torch.float32 or (has_cond and lr_img.dtype != torch.float32):
            raise RuntimeError("x and the conditioning image must be float32 (the reference forward is fp32-only)")
        if x.dim() != 4 or (has_cond and lr_img.dim() != 4):
            raise RuntimeError("x and the conditioning image must be NCHW")
        B, Cx, H, W = x.shape
        Bl = lr_img.shape[0] if has_cond else B
        mag = int(magnification_factor)
        plan = self._get_plan(B, Bl, H, W, mag, x.device, train)
        cx, _, cc = self._channels
        if Cx != cx or (has_cond and lr_img.shape[1] != cc):
            raise RuntimeError(f"expected {cx} image / {cc} conditioning channels, got x {Cx}"
                               + (f", cond {lr_img.shape[1]}" if has_cond else ""))
        if has_cond and (lr_img.shape[2] * mag != H or lr_img.shape[3] * mag != W):
            raise RuntimeError(f"conditioning image {tuple(lr_img.shape)} x{mag} does not match x {tuple(x.shape)}")
        if timestep.shape != (B,):
            raise RuntimeError(f"timestep must have shape ({B},), got {tuple(timestep.shape)}")
        if labels is not None:
            if self.variant != "generation" or not getattr(m, "num_classes", None):
                raise RuntimeError("labels need Residual_Attention_UNet_generation built with num_classes")
            labels = labels.to(torch.int64).contiguous()
            if labels.shape not in ((B,), (1,)):
                raise RuntimeError(f"labels must have shape ({B},) or (1,), got {tuple(labels.shape)}")
            # values: class ids in [0, num_classes); -1 marks a row that runs unconditionally.  An id >= num_classes
            # (nn.Embedding raises in the reference) never indexes the table on the device: the kernels poison that row
            # with NaN and skip its embedding gradient; host tensors are checked right here, for free.
            if not labels.is_cuda and labels.numel() and int(labels.max()) >= int(m.num_classes):
                raise IndexError(f"class label {int(labels.max())} out of range for num_classes={m.num_classes}")
        x = x.contiguous()
        if has_cond:
            lr_img = lr_img.contiguous()
        timestep = timestep.to(torch.int64).contiguous()
        out_dim = self._channels[1]
        with torch.cuda.device(x.device):
            if check_weights or plan.signature is None:
                self._sync_weights(plan)
            flags = 0
            if reuse_cond and has_cond:
                if plan.cond_key != (lr_img.data_ptr(), lr_img._version):
                    raise RuntimeError("reuse_cond=True but the conditioning in the workspace belongs to another lr_img")
                flags |= _lib.FWD_REUSE_COND
            out = torch.empty((B, out_dim, H, W), dtype=torch.float32, device=x.device)
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            st = plan.lib.drs_unet_forward_labels(
                plan.handle, C.c_void_p(plan.packed.data_ptr()), C.c_void_p(x.data_ptr()),
                C.c_void_p(timestep.data_ptr()), C.c_void_p(lr_img.data_ptr() if has_cond else None),
                C.c_void_p(labels.data_ptr() if labels is not None else None),
                int(labels.shape[0]) if labels is not None else 0,
                C.c_void_p(out.data_ptr()), C.c_void_p(plan.workspace.data_ptr()), plan.ws_bytes, flags, stream)
            _lib.check(st, "drs_unet_forward")
            if has_cond:
                plan.cond_key = (lr_img.data_ptr(), lr_img._version)
            if train:  # the kernels updated running_mean / running_var in place; num_batches_tracked follows here
                self._bn_epoch += 1
                nbt = [b for k, b in m.named_buffers() if k.endswith("num_batches_tracked")]
                with torch.no_grad():
                    torch._foreach_add_(nbt, 1)
        self._last_plan = plan
        return out

    # -- backward (training step) ---------------------------------------------------------------
    def backward(self, plan, x, timestep, dout, labels=None):
        """d(loss)/d(parameters) for the last train-mode forward on `plan`; returns {state_dict key: gradient}."""
        sd = self._tensors(plan.param_names)
        lib = plan.lib
        wanted = [(i, n) for i, n in enumerate(plan.param_names) if sd[n].requires_grad]
        total = sum(plan.param_numels[i] for i, _ in wanted)
        # parameters that this rank may leave without a gradient while another rank produces one (the label embedding when
        # the label is dropped): the multi-GPU exchange then appends one "used" flag per parameter to the same buffer
        has_flags = any(n == "label_emb.weight" for _, n in wanted)
        flat = torch.empty(total + (len(wanted) if has_flags else 0), dtype=torch.float32,
                           device=plan.device)  # fresh per step: .grad may alias it
        ptrs = (C.c_void_p * len(plan.param_names))()
        views, off, slots = {}, 0, []
        for i, n in wanted:
            k = plan.param_numels[i]
            views[n] = flat[off:off + k].view_as(sd[n])
            ptrs[i] = flat.data_ptr() + 4 * off
            slots.append((sd[n], off, k))
            off += k
        if getattr(plan, "packed_bwd", None) is None:
            plan.packed_bwd_bytes = lib.drs_unet_packed_bwd_bytes(plan.handle)
            plan.packed_bwd = torch.empty(plan.packed_bwd_bytes, dtype=torch.uint8, device=plan.device)
        dout = dout.contiguous()
        with torch.cuda.device(plan.device):
            stream = C.c_void_p(torch.cuda.current_stream(plan.device).cuda_stream)
            st = lib.drs_unet_backward_labels(
                plan.handle, C.c_void_p(plan.packed.data_ptr()), C.c_void_p(plan.packed_bwd.data_ptr()),
                plan.packed_bwd_bytes, C.c_void_p(x.data_ptr()), C.c_void_p(timestep.data_ptr()),
                C.c_void_p(labels.data_ptr() if labels is not None else None),
                int(labels.shape[0]) if labels is not None else 0, C.c_void_p(dout.data_ptr()), ptrs,
                C.c_void_p(plan.workspace.data_ptr()), plan.ws_bytes, stream)
        _lib.check(st, "drs_unet_backward")
        # (offsets, not view tensors: an extra reference to a view would make autograd CLONE it into .grad instead of
        # adopting it, 176 copy kernels per step)
        self._grad_buffer = (flat, total, slots, has_flags)
        return views

    def last_gradient_buffer(self):
        """(flat buffer, number of gradient elements, [(parameter, view of its slot)], has flag tail) of the last backward:
        the `.grad`s of the parameters are views of `flat`, which `dist.allreduce_gradients` reduces in place."""
        if self._grad_buffer is None:
            return None
        flat, total, slots, has_flags = self._grad_buffer
        return flat, total, [(p, flat[off:off + k].view_as(p)) for p, off, k in slots], has_flags

    def check_faults(self):
        """Synchronise and raise if a wave of the wave-specialised kernels gave up on an LDS counter during a forward of the
        last plan (drs_unet_check_faults: a protocol bug reports itself instead of hanging or faulting the device).
        The word is sticky: once set it is reported by every call until the weights are packed again.  Called at the end of
        every `Diffusion.sample` chain (hence by the tiler and the training previews), by bench.py after its timed loop and
        by smoke()."""
        plan = getattr(self, "_last_plan", None)
        if plan is None or plan.signature is None:
            return
        with torch.cuda.device(plan.device):
            st = plan.lib.drs_unet_check_faults(plan.handle, C.c_void_p(plan.packed.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream(plan.device).cuda_stream))
        _lib.check(st, "drs_unet_check_faults")

    # -- per-op timing (bench.py roofline) ------------------------------------------------------
    def profile_forward(self, x, timestep, lr_img, magnification_factor, iters=5, **kw):
        """Run `iters` forwards with HIP events around every op of the schedule (recorded on the launch stream)
        and return [(op name, mean ms, algorithmic flops, algorithmic bytes)]."""
        self.forward(x, timestep, lr_img, magnification_factor, **kw)
        plan = self._last_plan
        lib = plan.lib
        acc = {}
        order = []
        _lib.check(lib.drs_unet_profile_enable(plan.handle, 1), "drs_unet_profile_enable")
        try:
            for _ in range(iters):
                self.forward(x, timestep, lr_img, magnification_factor, **kw)
                n = lib.drs_unet_profile_num_ops(plan.handle)
                name = C.create_string_buffer(128)
                ms, fl, by = C.c_float(), C.c_double(), C.c_double()
                for i in range(n):
                    _lib.check(lib.drs_unet_profile_read(plan.handle, i, name, 128, C.byref(ms), C.byref(fl),
                                                         C.byref(by)), "drs_unet_profile_read")
                    key = name.value.decode()
                    if key not in acc:
                        acc[key] = [0.0, fl.value, by.value]
                        order.append(key)
                    acc[key][0] += ms.value
            self.last_launch_log = self._read_launch_log(plan)
        finally:
            lib.drs_unet_profile_enable(plan.handle, 0)
        return [(k, acc[k][0] / iters, acc[k][1], acc[k][2]) for k in order]

    @staticmethod
    def _read_launch_log(plan):
        """[(op name or "", kernel name)] for every kernel the last profiled forward launched, in launch order
        (drs_unet_profile_launch): what tools/collect_pmc.py joins with the dispatch rows of a counter pass."""
        lib = plan.lib
        op, kn = C.create_string_buffer(128), C.create_string_buffer(512)
        out = []
        for i in range(lib.drs_unet_profile_num_launches(plan.handle)):
            _lib.check(lib.drs_unet_profile_launch(plan.handle, i, op, 128, kn, 512), "drs_unet_profile_launch")
            out.append((op.value.decode(), kn.value.decode()))
        return out

    def logged_forward(self, x, timestep, lr_img, magnification_factor, **kw):
        """One forward with the plan's launch log on (per-op events too; the serial profiled schedule): returns
        (output, [(op, kernel)])."""
        plan = getattr(self, "_last_plan", None)
        if plan is None or plan.signature is None:
            raise RuntimeError("logged_forward: run a plain forward of the same shape first (plan and weights in place)")
        _lib.check(plan.lib.drs_unet_profile_enable(plan.handle, 1), "drs_unet_profile_enable")
        try:
            out = self.forward(x, timestep, lr_img, magnification_factor, **kw)
            log = self._read_launch_log(plan)
        finally:
            plan.lib.drs_unet_profile_enable(plan.handle, 0)
        return out, log

    # -- introspection (parity tests) ---------------------------------------------------------
    def tensor_names(self):
        return list(self._last_plan.tensor_index)

    def read_tensor(self, name):
        """NCHW copy of an intermediate activation left by the last forward."""
        plan = self._last_plan
        i = plan.tensor_index[name]
        dims = [C.c_int() for _ in range(4)]
        _lib.check(plan.lib.drs_unet_tensor_shape(plan.handle, i, *[C.byref(d) for d in dims]), "drs_unet_tensor_shape")
        shape = tuple(d.value for d in dims)
        dst = torch.empty(shape, dtype=torch.float32, device=plan.device)
        with torch.cuda.device(plan.device):
            stream = C.c_void_p(torch.cuda.current_stream(plan.device).cuda_stream)
            _lib.check(plan.lib.drs_unet_read_tensor(plan.handle, i, C.c_void_p(plan.workspace.data_ptr()),
                                                     C.c_void_p(dst.data_ptr()), stream), "drs_unet_read_tensor")
        return dst


class _UNetTrainFn(torch.autograd.Function):
    """Autograd node of one train-mode UNet forward: forward = drs_unet_forward_labels on the train plan, backward =
    drs_unet_backward_labels.  No gradient flows to x_t / the conditioning image (data in the reference's loops)."""

    @staticmethod
    def forward(ctx, engine, x, timestep, lr_img, mag, labels, *params):
        x = x.contiguous()
        timestep = timestep.to(torch.int64).contiguous()
        if labels is not None:
            labels = labels.to(torch.int64).contiguous()
        out = engine.forward(x, timestep, lr_img, mag, _in_autograd_fn=True, labels=labels)
        ctx.engine, ctx.plan = engine, engine._last_plan
        sd = engine._tensors(ctx.plan.param_names)
        ctx.names = [n for n in ctx.plan.param_names if sd[n].requires_grad]
        ctx.labels = labels
        ctx.save_for_backward(x, timestep)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, timestep = ctx.saved_tensors
        grads = ctx.engine.backward(ctx.plan, x, timestep, dout, ctx.labels)
        if ctx.labels is None:  # unconditional step: autograd leaves label_emb.weight.grad = None (Adam skips it)
            grads["label_emb.weight"] = None
        return (None,) * 6 + tuple(grads[n] for n in ctx.names)

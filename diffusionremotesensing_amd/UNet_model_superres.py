"""Drop-in `Residual_Attention_UNet_superres` + `EMA` whose forward runs on hand-written
gfx950 kernels.

Mirrors the reference's module tree so that `state_dict()` has the same 299 keys,
including the doubly-registered BatchNorms (reference UNet_model_superres.py:118-141),
and `parameters()` yields the same 176 tensors in the same order (so `torch.optim.Adam`,
`copy.deepcopy` and `EMA` see what they see with the reference).  The modules below
only *hold parameters*; none of their `forward`s is ever called.  The arithmetic of
reference `forward` (UNet_model_superres.py:337-379) is executed by `HipUNetEngine`
through the C-ABI in `include/drs_hip.h`.  There is no CPU or eager fallback: without
the compiled library, or with tensors that are not on a ROCm device, forward raises.
"""
import torch
import torch.nn as nn

from . import engine as _engine


class EMA:
    """Exponential moving average of parameters (reference UNet_model_superres.py:12-55).

    Same quirks: warm-up copies the full state_dict for `step_start_ema` steps, afterwards
    only `parameters()` are averaged (BatchNorm buffers stay frozen).
    On a ROCm device both the averaging and the warm-up copy are ONE multi-tensor HIP launch (`drs_ema_multi`) instead
    of 176 / 299 small kernels; models that live on the host (unit tests of the host logic) use the same formula in
    torch ops.
    """

    def __init__(self, beta):
        self.beta = beta
        self.step = 0
        self._tables = {}

    def _multi(self, dst, src, mode):
        """One `drs_ema_multi` launch over the tensor pairs (dst[i], src[i]); the pointer table is built once per pair list."""
        import ctypes as C

        from . import _lib
        for d, s_ in zip(dst, src):
            if d.shape != s_.shape or d.dtype != s_.dtype or not (d.is_contiguous() and s_.is_contiguous()):
                raise RuntimeError("EMA: the two models must hold identically shaped contiguous tensors")
            if d.element_size() not in (4, 8) or (mode == 0 and d.dtype != torch.float32):
                raise RuntimeError(f"EMA: unsupported tensor dtype {d.dtype}")
        key = (mode, tuple(d.data_ptr() for d in dst), tuple(s_.data_ptr() for s_ in src))
        ent = self._tables.get(mode)
        if ent is None or ent[0] != key:
            rows = [(d.data_ptr(), s_.data_ptr(), d.numel() * (d.element_size() // 4)) for d, s_ in zip(dst, src) if d.numel()]
            table = torch.tensor(rows, dtype=torch.int64).view(-1).to(dst[0].device)
            ent = (key, table, len(rows), max(r[2] for r in rows))
            self._tables[mode] = ent
        _, table, n, max_n = ent
        dev = dst[0].device
        with torch.cuda.device(dev):
            st = _lib.load().drs_ema_multi(C.c_void_p(table.data_ptr()), n, max_n, float(self.beta), mode,
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(st, "drs_ema_multi")

    def update_average(self, old, new):
        if old is None:
            return new
        return old * self.beta + (1 - self.beta) * new

    def update_model_average(self, ma_model, current_model):
        ma = list(ma_model.parameters())
        cur = list(current_model.parameters())
        with torch.no_grad():  # in-place on the parameters themselves so tensor versions advance
            if ma and ma[0].is_cuda:
                self._multi(ma, cur, 0)
                ma_model._drs_param_epoch = getattr(ma_model, "_drs_param_epoch", 0) + 1  # folded weight images must follow
            else:
                for p, c in zip(ma, cur):
                    p.copy_(self.update_average(p, c))

    def reset_parameters(self, ema_model, model):
        dst, src = ema_model.state_dict(), model.state_dict()
        if dst and dst.keys() == src.keys() and all(t.is_cuda for t in dst.values()) and all(t.is_cuda for t in src.values()):
            with torch.no_grad():
                self._multi(list(dst.values()), list(src.values()), 1)
            ema_model._drs_param_epoch = getattr(ema_model, "_drs_param_epoch", 0) + 1
        else:
            ema_model.load_state_dict(src)

    def step_ema(self, ema_model, model, step_start_ema=2000):
        if self.step < step_start_ema:
            self.reset_parameters(ema_model, model)
        else:
            self.update_model_average(ema_model, model)
        self.step += 1


def _time_mlp(dim_in, dim_out, device):
    return nn.Sequential(nn.Linear(dim_in, dim_out, device=device), nn.SiLU(),
                         nn.Linear(dim_out, dim_out, device=device))


def _conv(cin, cout, k, device=None, **kw):
    return nn.Conv2d(cin, cout, kernel_size=k, bias=True, device=device, **kw)


class _ParamHolder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard against accidental eager use
        raise RuntimeError(f"{type(self).__name__} only holds parameters; the UNet forward runs "
                           "in the HIP engine (diffusionremotesensing_amd.engine)")


class AttentionBlock(_ParamHolder):
    """Parameters of the additive attention gate (reference :57-108)."""

    def __init__(self, f_g, f_x, f_int, device):
        super().__init__()
        self.w_g = nn.Sequential(_conv(f_g, f_int, 1).to(device))
        self.w_x = nn.Sequential(_conv(f_x, f_int, 2, stride=2).to(device))
        self.psi = nn.Sequential(_conv(f_int, 1, 1).to(device), nn.Sigmoid())
        self.relu = nn.ReLU(inplace=False)
        self.result = nn.Sequential(_conv(f_x, f_x, 1).to(device), nn.BatchNorm2d(f_x).to(device))


class ResConvBlock(_ParamHolder):
    """Parameters of the residual double-conv block (reference :110-172).  Registration
    order matters: it fixes `parameters()` order and the aliased state_dict keys."""

    # attribute (= state_dict key) of the x_skip convolution: `conv_upsampled_lr_img` here, `conv_SAR_img` in
    # UNet_model_SAR_TO_NDVI.py:126, `conv_skip` in generate_new_imgs/UNet_model_generation.py:122
    SKIP_NAME = "conv_upsampled_lr_img"

    def __init__(self, in_ch, out_ch, time_emb_dim, device):
        super().__init__()
        self.time_mlp = _time_mlp(time_emb_dim, out_ch, device)
        self.batch_norm1 = nn.BatchNorm2d(out_ch, device=device)
        self.batch_norm2 = nn.BatchNorm2d(out_ch, device=device)
        self.shortcut_batch_norm = nn.BatchNorm2d(out_ch, device=device)
        self.relu = nn.ReLU(inplace=False)
        self.conv1 = nn.Sequential(_conv(in_ch, out_ch, 3, device, padding="same"), self.batch_norm1, self.relu)
        # dead weight in every block but the first; created on the
        # default device exactly like the reference does (:129)
        setattr(self, self.SKIP_NAME, nn.Conv2d(in_ch, out_ch, 3, padding=1))
        self.conv2 = nn.Sequential(_conv(out_ch, out_ch, 3, device, padding="same"), self.batch_norm2)
        self.shortcut_conv = nn.Sequential(_conv(in_ch, out_ch, 1, device, padding="same"), self.shortcut_batch_norm)


class UpConvBlock(_ParamHolder):
    """Parameters of conv + transposed-conv upsampling block (reference :174-207)."""

    def __init__(self, in_ch, out_ch, time_emb_dim, device):
        super().__init__()
        self.time_mlp = _time_mlp(time_emb_dim, out_ch, device)
        self.batch_norm = nn.BatchNorm2d(out_ch, device=device)
        self.relu = nn.ReLU(inplace=False)
        self.conv = _conv(in_ch, out_ch, 3, device, padding="same")
        self.transform = nn.ConvTranspose2d(out_ch, out_ch, kernel_size=3, stride=2, padding=1, bias=True,
                                            output_padding=1, device=device)


class gating_signal(_ParamHolder):
    """Parameters of the 1x1 conv + BN gating signal (reference :209-225)."""

    def __init__(self, in_dim, out_dim, device):
        super().__init__()
        self.conv = _conv(in_dim, out_dim, 1, device, padding="same")
        self.batch_norm = nn.BatchNorm2d(out_dim, device=device)
        self.relu = nn.ReLU(inplace=False)
        self.device = device


class ResidualBlock(_ParamHolder):
    """Parameters of one LR-encoder residual block (reference :230-242)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size, stride, padding)


class RRDB(_ParamHolder):
    """Parameters of the low-resolution image encoder (reference :244-260)."""

    def __init__(self, in_channels, out_channels, num_blocks=3):
        super().__init__()
        self.blocks = nn.Sequential(*[ResidualBlock(in_channels, in_channels) for _ in range(num_blocks)])
        self.conv_out = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)


class _HipUNet(nn.Module):
    """What the three UNet mirrors share: the trunk's parameter holders (identical in the three reference files),
    the host-side pos_encoding helper, the lazily created HIP engine and a deepcopy that never copies it."""

    VARIANT = None
    RES_BLOCK = ResConvBlock

    def _build_trunk(self, out_channels, device):
        """Registers conv_blocks .. output in the reference's order (UNet_model_superres.py:287-322)."""
        self.down_channels = (16, 32, 64, 128, 256)
        self.up_channels = (256, 128, 64, 32, 16)
        dc, uc, te, Res = self.down_channels, self.up_channels, self.time_emb_dim, self.RES_BLOCK
        self.conv_blocks = nn.ModuleList(Res(dc[i], dc[i + 1], te, device) for i in range(len(dc) - 2))
        self.downs = nn.ModuleList(_conv(dc[i + 1], dc[i + 1], 3, device, stride=2, padding=1)
                                   for i in range(len(dc) - 2))
        self.bottle_neck = Res(dc[-2], dc[-1], te, device)
        self.gating_signals = nn.ModuleList(gating_signal(uc[i], uc[i + 1], device) for i in range(len(uc) - 2))
        self.attention_blocks = nn.ModuleList(AttentionBlock(uc[i + 1], uc[i + 1], uc[i + 1], device)
                                              for i in range(len(uc) - 2))
        self.ups = nn.ModuleList(UpConvBlock(uc[i], uc[i], te, device) for i in range(len(uc) - 2))
        self.up_convs = nn.ModuleList(_conv(int(uc[i] * 3 / 2), uc[i + 1], 3, padding=1).to(device)
                                      for i in range(len(uc) - 2))
        self.output = nn.Conv2d(uc[-2], out_channels, 1)

    def pos_encoding(self, t, channels, device):
        """Sinusoidal embedding as a host-visible helper (reference :328-335); the forward
        computes it inside the fused time-embedding kernel instead."""
        inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2, device=device).float() / channels))
        arg = t.repeat(1, channels // 2) * inv_freq
        return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)

    # -- HIP dispatch ---------------------------------------------------------------------
    def hip_engine(self):
        if self.__dict__.get("_hip_engine") is None:
            self.__dict__["_hip_engine"] = _engine.HipUNetEngine(self, variant=self.VARIANT)
        return self.__dict__["_hip_engine"]

    def __deepcopy__(self, memo):
        # engines hold device workspaces and raw pointers: never copied (EMA deep-copies the model)
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_hip_engine" else copy.deepcopy(v, memo)
        return new


class Residual_Attention_UNet_superres(_HipUNet):
    """Same constructor, attributes, state_dict and `forward(x, timestep, lr_img,
    magnification_factor)` contract as reference UNet_model_superres.py:266-379."""

    VARIANT = "superres"

    def __init__(self, image_channels=3, out_dim=3, device=None):
        super().__init__()
        self.image_channels = image_channels
        self.out_dim = out_dim
        self.time_emb_dim = 100
        self.device = device
        self.conv0 = nn.Conv2d(image_channels, 16, 3, padding=1)
        self.LR_encoder = RRDB(in_channels=image_channels, out_channels=image_channels, num_blocks=3)
        self.conv_upsampled_lr_img = nn.Conv2d(image_channels, 16, 3, padding=1)
        self._build_trunk(out_dim, device)
        self._hip_engine = None

    def forward(self, x, timestep, lr_img, magnification_factor):
        return self.hip_engine().forward(x, timestep, lr_img, magnification_factor)

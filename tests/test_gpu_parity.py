"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the golden vectors.
Tolerance: max-abs/max-abs-ref <= 1e-3 and rel-L2 <= 1e-3 (BASELINE.json north_star);
the fp32 kernels are held to 2e-5."""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import golden_inputs, rel_errors, replay_noise_source

pytestmark = pytest.mark.gpu

TOL = 1e-3  # BASELINE.json north_star: the bar the path is allowed; the tests hold the kernels to what they deliver:
TOL_BF16X3 = 1e-4  # eval forwards, block taps, operators of the default split-bf16 kernels (measured 1e-5 .. 2e-5)
TOL_F32 = 2e-5
# mfma_f16 (single fp16 MFMA per product) measured 1.1e-3 max-rel / 8e-4 rel-L2 on these weights: outside the 1e-3
# bar, so it is NOT the shipped default and not in the default test list; DRS_TEST_IMPLS=...,mfma_f16 runs it against
# a 2.5e-3 bound to keep the opt-in mode from regressing.
TOL_F16_OPT_IN = 2.5e-3
IMPLS = [i for i in os.environ.get("DRS_TEST_IMPLS", "direct,mfma_f32,mfma_bf16x3").split(",") if i]


def _tol(impl):
    if impl == "mfma_f16":
        return TOL_F16_OPT_IN
    return TOL_F32 if impl in ("direct", "mfma_f32") else TOL_BF16X3


def _tol_train(impl):
    """Train-mode forwards (batch statistics): split-bf16 is not the training default and is only
    held to the north_star bar there."""
    return TOL if impl == "mfma_bf16x3" else _tol(impl)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from diffusionremotesensing_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev, seeded_sd):
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(seeded_sd)
    return m.to(dev).eval()


def _assert_close(got, want, tol, what=""):
    e_max, e_l2 = rel_errors(got.cpu(), want)
    assert e_max <= tol and e_l2 <= tol, f"{what}: max-rel {e_max:.3e} rel-L2 {e_l2:.3e} > {tol}"


# ---------------------------------------------------------------------------------------------
# operator level
# ---------------------------------------------------------------------------------------------
CONV_CASES = [
    # (N, Cin, H, W, Cout, k, stride, pad, transposed, out_pad)
    (2, 16, 16, 16, 32, 3, 1, 1, False, 0),
    (1, 32, 13, 9, 64, 3, 1, 1, False, 0),      # ragged
    (2, 96, 8, 24, 32, 3, 1, 1, False, 0),      # concat-width input
    (2, 32, 16, 16, 32, 3, 2, 1, False, 0),     # downs
    (1, 64, 11, 7, 64, 3, 2, 1, False, 0),      # downs, odd size
    (2, 16, 10, 6, 32, 1, 1, 0, False, 0),      # shortcut / gating / w_g
    (2, 64, 12, 8, 64, 2, 2, 0, False, 0),      # w_x
    (2, 32, 9, 5, 1, 1, 1, 0, False, 0),        # psi (Cout = 1)
    (2, 32, 8, 8, 3, 1, 1, 0, False, 0),        # output (Cout = 3)
    (2, 64, 8, 8, 64, 3, 2, 1, True, 1),        # transform
    (1, 128, 5, 3, 128, 3, 2, 1, True, 1),      # transform, ragged
    (1, 256, 4, 4, 256, 3, 1, 1, False, 0),     # bottleneck width
    (2, 64, 40, 24, 128, 3, 1, 1, False, 0),    # wave-specialised kernel: 2 channel groups, border + interior patches
    (1, 128, 17, 33, 64, 3, 1, 1, False, 0),    # wave-specialised kernel: 4 K-chunks, ragged edges on both axes
    (0, 16, 8, 8, 16, 3, 1, 1, False, 0),       # empty batch
]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_flavours(dev, case, impl):
    from diffusionremotesensing_amd import hip_ops, synthetic
    N, Cin, H, W, Cout, k, stride, pad, tr, op = case
    x = synthetic.tensor_normal(f"conv.x.{case}", (N, Cin, H, W))
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    w = synthetic.tensor_normal(f"conv.w.{case}", wshape, std=(Cin * k * k) ** -0.5)
    b = synthetic.tensor_normal(f"conv.b.{case}", (Cout,), std=0.1)
    for relu in (False, True):
        got = hip_ops.conv2d(x.to(dev), w.to(dev), b.to(dev), stride=stride, padding=pad, transposed=tr,
                             output_padding=op, relu=relu, impl=impl)
        if tr:
            want = F.conv_transpose2d(x, w, b, stride=stride, padding=pad, output_padding=op)
        else:
            want = F.conv2d(x, w, b, stride=stride, padding=pad)
        if relu:
            want = F.relu(want)
        assert tuple(got.shape) == tuple(want.shape)
        if N:
            _assert_close(got, want, _tol(impl), f"conv {case} relu={relu}")


# (N, Cc, Ch, LH, LW): one decoder stage at sizes that hit interior tiles, all four image borders, ragged tiles, several
# channel groups / K-chunks, more items than CUs per XCD slice, and the 1-tile image
UPFUSE_CASES = [
    (2, 64, 32, 16, 16),     # stage-2 shape: one tile per image, 2 K-chunks
    (1, 128, 64, 24, 40),    # ragged tiles on both axes, 2 channel groups, 4 K-chunks
    (2, 256, 128, 32, 32),   # stage-0 shape: 2 x 2 tiles, 4 channel groups, 8 K-chunks
    (1, 64, 32, 48, 48),     # 3 x 3 tiles: an interior tile (fast window path)
    (3, 32, 32, 9, 17),      # smallest supported heights, odd sizes
    (2, 64, 32, 32, 40),     # 64 output rows: with a projection, the folded form of both launches (as is (1, 64, 32, 48, 48)); ragged tiles in x
]
TOL_UPFUSE = 1e-4  # measured ~1e-5 (16-bit operand mantissas, fp32 accumulation, fp32 edge vectors)


@pytest.mark.parametrize("case", UPFUSE_CASES)
def test_upconv_fused_matches_convtranspose_cat_conv(dev, case):
    """ups.i.transform -> torch.cat -> up_convs.i as ONE composite transposed convolution + the att-half convolution
    (csrc/upfuse_sp.hip) against the reference's three ops (UNet_model_superres.py:206-207,376-377) in fp32 on the CPU."""
    from diffusionremotesensing_amd import hip_ops, synthetic
    N, Cc, Ch, LH, LW = case
    h = synthetic.tensor_normal(f"uf.h.{case}", (N, Cc, LH, LW))
    att = synthetic.tensor_normal(f"uf.att.{case}", (N, Ch, 2 * LH, 2 * LW))
    t_w = synthetic.tensor_normal(f"uf.tw.{case}", (Cc, Cc, 3, 3), std=(Cc * 2.25) ** -0.5)
    t_b = synthetic.tensor_normal(f"uf.tb.{case}", (Cc,), std=0.3)
    v_w = synthetic.tensor_normal(f"uf.vw.{case}", (Ch, Cc + Ch, 3, 3), std=((Cc + Ch) * 9) ** -0.5)
    v_b = synthetic.tensor_normal(f"uf.vb.{case}", (Ch,), std=0.3)
    post2 = synthetic.tensor_normal(f"uf.p2.{case}", (N, Ch), std=0.5)
    want = F.conv2d(torch.cat([F.conv_transpose2d(h, t_w, t_b, stride=2, padding=1, output_padding=1), att], 1), v_w, v_b,
                    padding=1)
    args = [t.to(dev) for t in (h, att, t_w, t_b, v_w, v_b)]
    got = hip_ops.upconv_fused(*args)
    _assert_close(got, want, TOL_UPFUSE, f"upconv_fused {case}")
    # borders are where the composite differs from the plain formula: check them on their own scale too
    for sl in ((slice(None), slice(None), slice(0, 1)), (slice(None), slice(None), slice(-1, None)),
               (slice(None), slice(None), slice(None), slice(0, 1)), (slice(None), slice(None), slice(None), slice(-1, None))):
        _assert_close(got[sl], want[sl], TOL_UPFUSE, f"upconv_fused {case} edge")
    got, got2 = hip_ops.upconv_fused(*args, post2=post2.to(dev))
    _assert_close(got, want, TOL_UPFUSE, f"upconv_fused {case} (with second output)")
    _assert_close(got2, want + post2[:, :, None, None], TOL_UPFUSE, f"upconv_fused {case} second output")
    if Ch == 32:
        f_w = synthetic.tensor_normal(f"uf.fw.{case}", (3, 32, 1, 1), std=32 ** -0.5)
        f_b = synthetic.tensor_normal(f"uf.fb.{case}", (3,), std=0.2)
        gotp = hip_ops.upconv_fused(*args, fuse_w=f_w.to(dev), fuse_b=f_b.to(dev))
        _assert_close(gotp, F.conv2d(want, f_w, f_b), TOL_UPFUSE, f"upconv_fused {case} fused projection")


def test_no_protocol_faults_after_a_full_size_forward(dev, model):
    """The wave-specialised kernels bound every LDS-counter poll; a wave that runs out records it in the plan's fault word
    and ends (csrc/sp_sync.h).  A healthy forward at the headline size leaves the word clear."""
    from diffusionremotesensing_amd import synthetic
    model.hip_engine().set_impl("mfma_bf16x3")
    x, t, lr = golden_inputs("faults", 16, 16, 3, 256, 2, 1500)
    with torch.no_grad():
        y = model(x.to(dev), t.to(dev), lr.to(dev), 2)
    model.hip_engine().check_faults()
    assert torch.isfinite(y).all()


def test_upconv_fused_rejects_bad_shapes(dev):
    from diffusionremotesensing_amd import hip_ops
    z = lambda *s: torch.zeros(*s, device=dev)
    with pytest.raises(RuntimeError, match="multiples of 32"):
        hip_ops.upconv_fused(z(1, 48, 4, 4), z(1, 16, 8, 8), z(48, 48, 3, 3), z(48), z(16, 64, 3, 3), z(16))
    with pytest.raises(RuntimeError, match="do not fit"):
        hip_ops.upconv_fused(z(1, 32, 4, 4), z(1, 32, 8, 9), z(32, 32, 3, 3), z(32), z(32, 64, 3, 3), z(32))


def test_conv2d_rejects_unsupported(dev):
    from diffusionremotesensing_amd import hip_ops
    x = torch.zeros(1, 4, 8, 8, device=dev)
    w = torch.zeros(4, 4, 5, 5, device=dev)
    with pytest.raises(RuntimeError, match="unsupported flavour"):
        hip_ops.conv2d(x, w, None, padding=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip_ops.conv2d(x.cpu(), w.cpu(), None)


@pytest.mark.parametrize("shape,scale", [((2, 3, 16, 16), 2), ((1, 3, 7, 5), 2), ((1, 3, 8, 8), 4), ((2, 1, 5, 9), 3)])
def test_bicubic(dev, shape, scale):
    from diffusionremotesensing_amd import hip_ops, synthetic
    x = synthetic.tensor_uniform(f"bicubic.{shape}", shape)
    got = hip_ops.bicubic_upsample(x.to(dev), scale)
    want = F.interpolate(x, scale_factor=scale, mode="bicubic")
    _assert_close(got, want, 1e-5, f"bicubic {shape} x{scale}")


@pytest.mark.parametrize("dim", [32, 64, 256])
def test_time_mlp(dev, seeded_sd, dim):
    from diffusionremotesensing_amd import hip_ops
    from oracle import unet_oracle as U
    pfx = {32: "conv_blocks.0.time_mlp", 64: "conv_blocks.1.time_mlp", 256: "bottle_neck.time_mlp"}[dim]
    t = torch.tensor([1, 2, 49, 750, 1499], dtype=torch.int64)
    want = U._time_mlp(seeded_sd, pfx, U.pos_encoding(t.unsqueeze(-1).float(), 100))[:, :, 0, 0]
    got = hip_ops.time_mlp(t.to(dev), *[seeded_sd[f"{pfx}.{k}"].to(dev) for k in ("0.weight", "0.bias", "2.weight", "2.bias")])
    _assert_close(got, want, 2e-5, f"time_mlp {dim}")


def test_noise_images_golden(dev, golden):
    from diffusionremotesensing_amd import hip_ops, synthetic
    from oracle import diffusion_oracle as D
    _, ah, _ = D.schedule("cosine", 1500)
    x0 = synthetic.tensor_uniform("g6.x0", (4, 3, 32, 32))
    got = hip_ops.noise_images(x0.to(dev), torch.from_numpy(golden["g6_eps"]).to(dev),
                               torch.from_numpy(golden["g6_t"]).to(dev), ah.to(dev))
    _assert_close(got, torch.from_numpy(golden["g6_x_t"]), 1e-6, "noise_images")


def test_sampler_step(dev):
    from diffusionremotesensing_amd import hip_ops, synthetic
    from oracle import diffusion_oracle as D
    a, ah, b = D.schedule("cosine", 50)
    x = synthetic.tensor_normal("ss.x", (3, 3, 16, 16))
    e = synthetic.tensor_normal("ss.e", (3, 3, 16, 16))
    z = synthetic.tensor_normal("ss.z", (3, 3, 16, 16))
    for i, noise in ((49, z), (7, z), (1, None)):
        t = (torch.ones(3) * i).long()
        want = D.sampler_step(x, e, noise if noise is not None else torch.zeros_like(x), t, a, ah, b)
        got = hip_ops.sampler_step_(x.clone().to(dev), e.to(dev), None if noise is None else noise.to(dev), i,
                                    a.to(dev), ah.to(dev), b.to(dev))
        _assert_close(got, want, 1e-6, f"sampler_step t={i}")


# ---------------------------------------------------------------------------------------------
# network level
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl", IMPLS)
def test_unet_blocks_golden(dev, model, golden, impl):
    """Every block output of the UNet against the reference's own activations (G3)."""
    x, t, lr = golden_inputs("g3", 2, 2, 3, 16, 2, 1500)
    eng = model.hip_engine()
    eng.set_impl(impl)
    eng.keep_intermediates = True  # production plans never write up_convs.2 (fused with the output conv)
    try:
        with torch.no_grad():
            out = model(x.to(dev), t.to(dev), lr.to(dev), 2)
    finally:
        eng.keep_intermediates = False
    _assert_close(out, torch.from_numpy(golden["g3_out"]), _tol(impl), "g3 output")
    checked = 0
    for k, v in golden.items():
        if k.startswith("g3_tap_"):
            _assert_close(eng.read_tensor(k[len("g3_tap_"):]), torch.from_numpy(v), _tol(impl), k)
            checked += 1
    assert checked == 20


@pytest.mark.parametrize("impl", IMPLS)
def test_unet_forward_golden(dev, model, golden, impl):
    model.hip_engine().set_impl(impl)
    with torch.no_grad():
        x, t, lr = golden_inputs("g4", 2, 2, 3, 64, 2, 1500)
        _assert_close(model(x.to(dev), t.to(dev), lr.to(dev), 2), torch.from_numpy(golden["g4_out"]), _tol(impl), "g4")
        _assert_close(model(x.to(dev), t.to(dev), lr[:1].to(dev), 2), torch.from_numpy(golden["g4_out_lr_broadcast"]),
                      _tol(impl), "g4 lr broadcast")
        x, t, lr = golden_inputs("g4m4", 1, 1, 3, 64, 4, 1500)
        _assert_close(model(x.to(dev), t.to(dev), lr.to(dev), 4), torch.from_numpy(golden["g4_out_mag4"]), _tol(impl),
                      "g4 mag4")
        from diffusionremotesensing_amd import synthetic
        xr = synthetic.tensor_normal("g4r.x", (1, 3, 24, 40))
        lrr = synthetic.tensor_uniform("g4r.lr", (1, 3, 12, 20))
        _assert_close(model(xr.to(dev), torch.tensor([77], device=dev), lrr.to(dev), 2),
                      torch.from_numpy(golden["g4_out_rect"]), _tol(impl), "g4 rect")


@pytest.mark.parametrize("impl", IMPLS)
def test_unet_forward_config1_vs_oracle(dev, model, seeded_sd, impl):
    """BASELINE config 1 shape (B=4, 128x128 <- 64x64) against the oracle run here."""
    from oracle import unet_oracle as U
    model.hip_engine().set_impl(impl)
    x, t, lr = golden_inputs("cfg1", 4, 4, 3, 128, 2, 50)
    with torch.no_grad():
        want = U.unet_forward(seeded_sd, x, t, lr, 2)
        got = model(x.to(dev), t.to(dev), lr.to(dev), 2)
    _assert_close(got, want, _tol(impl), "cfg1 forward")


_CFG2_ORACLE = {}


def _cfg2_oracle(seeded_sd, x, t, lr):
    if "out" not in _CFG2_ORACLE:
        from oracle import unet_oracle as U
        with torch.no_grad():
            _CFG2_ORACLE["out"] = U.unet_forward(seeded_sd, x, t, lr, 2)
    return _CFG2_ORACLE["out"]


@pytest.mark.parametrize("impl", IMPLS)
def test_unet_forward_config2_full_size(dev, model, seeded_sd, impl):
    """BASELINE config 2 (B=16, 256x256 <- 128x128): ALL 16 images against the oracle (one batched CPU forward, computed
    once for the module: ~30 s of host time), each image held to the tolerance on its own, + the batch-independence
    property (eval-mode forward of a batch == forwards of its images)."""
    model.hip_engine().set_impl(impl)
    x, t, lr = golden_inputs("cfg2", 16, 16, 3, 256, 2, 1500)
    want = _cfg2_oracle(seeded_sd, x, t, lr)
    with torch.no_grad():
        got = model(x.to(dev), t.to(dev), lr.to(dev), 2)
        assert torch.isfinite(got).all()
        for i in range(16):
            _assert_close(got[i:i + 1], want[i:i + 1], _tol(impl), f"cfg2 image {i}")
        single = model(x[5:6].to(dev), t[5:6].to(dev), lr[5:6].to(dev), 2)
        _assert_close(got[5:6], single.cpu(), 1e-6, "batch independence")


@pytest.mark.parametrize("impl", IMPLS)
def test_unet_forward_rectangular_vs_oracle(dev, model, seeded_sd, impl):
    """A 96 x 160 image, batch 3: every fused kernel of the default plan engages (the first encoder block's 8 x 16
    patches, the composite up-sampling stages from 12 x 20 cells, the 128-channel gate with a 4-pixel tail block per row) on
    shapes where rows != columns, patch counts are not powers of two and pixel blocks are partly outside the tensor."""
    from diffusionremotesensing_amd import synthetic
    from oracle import unet_oracle as U
    model.hip_engine().set_impl(impl)
    x = synthetic.tensor_normal("rect.x", (3, 3, 96, 160), 0)
    lr = synthetic.tensor_uniform("rect.lr", (3, 3, 48, 80), 0)
    t = synthetic.tensor_randint("rect.t", (3,), 1, 1500, 0)
    with torch.no_grad():
        want = U.unet_forward(seeded_sd, x, t, lr, 2)
        got = model(x.to(dev), t.to(dev), lr.to(dev), 2)
    _assert_close(got, want, _tol(impl), "rectangular forward")
    model.hip_engine().check_faults()


def test_reuse_cond_matches_full(dev, model):
    model.hip_engine().set_impl(IMPLS[-1])
    x, t, lr = golden_inputs("g4", 2, 2, 3, 64, 2, 1500)
    xd, td, lrd = x.to(dev), t.to(dev), lr.to(dev)
    eng = model.hip_engine()
    with torch.no_grad():
        a = eng.forward(xd, td, lrd, 2)
        b = eng.forward(xd, td, lrd, 2, reuse_cond=True, check_weights=False)
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="another lr_img"):
        eng.forward(xd, td, lrd.clone(), 2, reuse_cond=True)


def test_weight_updates_are_picked_up(dev, seeded_sd):
    """EMA / optimizer steps change parameters in place: the folded weights must follow."""
    from diffusionremotesensing_amd.UNet_model_superres import EMA, Residual_Attention_UNet_superres
    from oracle import unet_oracle as U
    import copy
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    x, t, lr = golden_inputs("g3", 2, 2, 3, 16, 2, 1500)
    with torch.no_grad():
        y0 = m(x.to(dev), t.to(dev), lr.to(dev), 2)
        ema_model = copy.deepcopy(m).eval().requires_grad_(False)
        assert ema_model._hip_engine is None
        for p in m.parameters():
            p.mul_(1.01)
        ema = EMA(0.5)
        ema.step = 10
        ema.step_ema(ema_model, m, step_start_ema=5)
        y1 = ema_model(x.to(dev), t.to(dev), lr.to(dev), 2)
        sd = {k: v.cpu() for k, v in ema_model.state_dict().items()}
        _assert_close(y1, U.unet_forward(sd, x, t, lr, 2), _tol(IMPLS[-1]), "ema")
        y2 = m(x.to(dev), t.to(dev), lr.to(dev), 2)
    assert not torch.equal(y0, y2)


def test_forward_error_behaviour(dev, model):
    x, t, lr = golden_inputs("g3", 2, 2, 3, 16, 2, 1500)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(x, t, lr, 2)
    with pytest.raises(RuntimeError, match="does not match"):
        model(x.to(dev), t.to(dev), lr[:, :, :4].to(dev), 2)
    with pytest.raises(RuntimeError, match="float32"):
        model(x.double().to(dev), t.to(dev), lr.to(dev), 2)
    with pytest.raises(RuntimeError, match="divisible by 8"):
        model(x[:, :, :12, :12].to(dev), t.to(dev), lr[:, :, :6, :6].to(dev), 2)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    model.train()
    try:  # a training step needs one LR image per sample (DataLoader batches); the broadcast form is sampling-only
        out = model(x.to(dev), t.to(dev), lr[:1].to(dev), 2)
        with pytest.raises(RuntimeError, match="lr_img batch must equal"):
            out.sum().backward()
    finally:
        model.eval()
        model.load_state_dict(sd0)  # the train-mode forward moved the shared fixture's running statistics
        model.zero_grad()


@pytest.mark.parametrize("impl", IMPLS)
def test_end_to_end_sample_golden(dev, model, golden, impl):
    """Diffusion.sample (T=50, cosine) with the reference's noise replayed: final x against G7; PSNR on the
    [0,1]-clamped images (what the reference's callers display)."""
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
    model.hip_engine().set_impl(impl)
    d = Diffusion("cosine", model, "/nonexistent/snapshot.pt", noise_steps=50, device=dev, magnification_factor=2,
                  image_size=64, Degradation_type="DownBlur")
    lr1 = synthetic.tensor_uniform("g7.small.lr", (3, 32, 32))
    x = d.sample(2, model, lr1, input_channels=3, noise_source=replay_noise_source(1234)).cpu()
    assert model.training  # quirk Q5: sample leaves the model in train mode
    model.eval()
    ref = torch.from_numpy(golden["g7_small_x"])
    e_max, e_l2 = rel_errors(x, ref)
    mse = ((x.clamp(0, 1) - ref.clamp(0, 1)) ** 2).mean().item()
    psnr = float("inf") if mse == 0 else -10 * torch.log10(torch.tensor(mse)).item()
    print(f"e2e sample [{impl}]: max-rel {e_max:.3e} rel-L2 {e_l2:.3e} PSNR {psnr:.1f} dB")
    # 49 chained forwards amplify rounding differences.  Bounds = ~10x what was measured on MI355X (fp32 kernels 2e-6 /
    # 101 dB, split-bf16 2.5e-5 / 78 dB): a regression of one order of magnitude fails
    if impl in ("direct", "mfma_f32"):
        assert e_l2 <= 3e-5 and psnr >= 85, (e_l2, psnr)
    elif impl == "mfma_bf16x3":
        assert e_l2 <= 3e-4 and psnr >= 65, (e_l2, psnr)
    else:
        assert e_l2 <= 5e-3 and psnr > 40, (e_l2, psnr)


def _psnr_clamped(a, b):
    mse = ((a.clamp(0, 1) - b.clamp(0, 1)) ** 2).mean().item()
    return float("inf") if mse == 0 else -10 * torch.log10(torch.tensor(mse)).item()


@pytest.mark.parametrize("impl", IMPLS)
def test_config1_sample_chain_golden(dev, model, golden, impl):
    """BASELINE configs[0] end to end (the reference's own CPU-runnable case): Diffusion.sample with n=4, 64x64 -> 128x128,
    T=50, cosine, the reference's CPU-generator draws replayed (seed 4321) against the reference's output G7 `cfg1`
    (train_diffusion_superres.py:207-255; stored as fp32 + fp64 checksums by tools/make_golden.py).  This is the
    "PSNR vs ref" of BASELINE.json's metric."""
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
    model.hip_engine().set_impl(impl)
    d = Diffusion("cosine", model, "/nonexistent/snapshot.pt", noise_steps=50, device=dev, magnification_factor=2,
                  image_size=128, Degradation_type="DownBlur")
    lr1 = synthetic.tensor_uniform("g7.cfg1.lr", (3, 64, 64))
    x = d.sample(4, model, lr1, input_channels=3, noise_source=replay_noise_source(4321)).cpu()
    model.eval()
    ref = torch.from_numpy(golden["g7_cfg1_x"]).float()
    assert x.shape == ref.shape == (4, 3, 128, 128)
    e_max, e_l2 = rel_errors(x, ref)
    psnr = _psnr_clamped(x, ref)
    csum = golden["g7_cfg1_checksum"]  # fp64 sum and abs-sum of the reference's fp32 output
    d_sum = abs(x.double().sum().item() - csum[0]) / csum[1]
    d_abs = abs(x.double().abs().sum().item() - csum[1]) / csum[1]
    print(f"config-1 chain [{impl}]: max-rel {e_max:.3e} rel-L2 {e_l2:.3e} PSNR {psnr:.1f} dB checksum {d_sum:.2e} {d_abs:.2e}")
    # (the fixture was fp16-rounded until round 3, which capped this bound at 4e-4; measured: 2e-5 split-bf16, 3e-6 fp32)
    tol = 1e-4 if impl != "mfma_f16" else 5e-3
    # (max-rel of a 49-step chain is set by a handful of pixels - 4e-4 against 2.4e-5 rel-L2 on the split-bf16 kernels: held
    # to north_star's 1e-3, rel-L2 to 1e-4)
    assert e_l2 <= tol and e_max <= (TOL if impl != "mfma_f16" else 5e-3) and psnr >= (70 if impl != "mfma_f16" else 40), (e_max, e_l2, psnr)
    assert d_sum <= (1e-5 if impl in ("direct", "mfma_f32") else 1e-4 if impl != "mfma_f16" else 1e-2)
    assert d_abs <= (1e-5 if impl in ("direct", "mfma_f32") else 1e-4 if impl != "mfma_f16" else 1e-2)


@pytest.mark.parametrize("impl", IMPLS)
def test_full_length_chain_golden(dev, seeded_sd, golden, impl, monkeypatch):
    """G10: `Diffusion.sample` over the FULL configs[1] schedule (cosine, T = 1500: 1499 forwards + updates,
    train_diffusion_superres.py:234-249) at 32x32, n = 2, against the reference's own chain with the reference's
    CPU-generator draws replayed - the update arithmetic at every t, not only the 49 steps of the short chains.  The `output`
    projection is scaled by 1e-2 (a chain on random weights is chaotic otherwise).  The states entering steps
    1400 / 1000 / 500 / 100 / 1 are compared too (captured around hip_ops.sampler_step_)."""
    from conftest import LONGCHAIN_STEPS, longchain_state_dict
    from diffusionremotesensing_amd import hip_ops, synthetic
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(longchain_state_dict(seeded_sd))
    m = m.to(dev).eval()
    m.hip_engine().set_impl(impl)
    d = Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=1500, device=dev, magnification_factor=2,
                  image_size=32, Degradation_type="DownBlur")
    seen = {}
    real_step = hip_ops.sampler_step_

    def recording_step(x, eps, noise, i, *a):
        if i in LONGCHAIN_STEPS:
            seen[i] = x.clone()  # the state ENTERING step i (what the reference's model sees at t = i)
        return real_step(x, eps, noise, i, *a)

    monkeypatch.setattr(hip_ops, "sampler_step_", recording_step)
    lr1 = synthetic.tensor_uniform("g10.lr", (3, 16, 16))
    x = d.sample(2, m, lr1, input_channels=3, noise_source=replay_noise_source(1010)).cpu()
    tol = 1e-4 if impl != "mfma_f16" else 5e-3
    worst = 0.0
    for i in LONGCHAIN_STEPS:
        e_max, e_l2 = rel_errors(seen[i].cpu(), torch.from_numpy(golden[f"g10_x_entering_{i}"]))
        worst = max(worst, e_max, e_l2)
        assert e_max <= tol and e_l2 <= tol, (i, e_max, e_l2)
    e_max, e_l2 = rel_errors(x, torch.from_numpy(golden["g10_x"]))
    print(f"1499-step chain [{impl}]: final max-rel {e_max:.3e} rel-L2 {e_l2:.3e}; worst over the recorded states {worst:.3e}")
    assert e_max <= tol and e_l2 <= tol, (e_max, e_l2)


@pytest.mark.parametrize("impl", IMPLS)
def test_train_mode_forward_golden(dev, seeded_sd, golden, impl):
    """model.train() under no_grad: batch-statistics BatchNorm + running-stat update (reference nn.BatchNorm2d
    defaults) against the reference's own train-mode forward (G5) - output, MSE loss, updated running statistics."""
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    from oracle import unet_oracle as U
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).train()
    m.hip_engine().set_impl(impl, train_impl=impl)
    x, t, lr = golden_inputs("g5", 4, 4, 3, 32, 2, 1500)
    with torch.no_grad():
        out = m(x.to(dev), t.to(dev), lr.to(dev), 2)
    _assert_close(out, torch.from_numpy(golden["g5_out"]), _tol_train(impl) if impl != "direct" else 5e-5, "g5 train output")
    noise = synthetic.tensor_normal("g5.noise", (4, 3, 32, 32))
    loss = torch.nn.functional.mse_loss(out.cpu(), noise).item()
    assert abs(loss - float(golden["g5_loss"])) <= 2e-3 * float(golden["g5_loss"])
    sd = m.state_dict()
    for k in ("conv_blocks.0.batch_norm1", "bottle_neck.batch_norm2", "attention_blocks.2.result.1"):
        _assert_close(sd[k + ".running_mean"], torch.from_numpy(golden[f"g5_rm_{k}"]), 1e-3, k + " running_mean")
        _assert_close(sd[k + ".running_var"], torch.from_numpy(golden[f"g5_rv_{k}"]), 1e-3, k + " running_var")
        assert int(sd[k + ".num_batches_tracked"]) == 8
    assert sd["conv_blocks.0.conv1.1.running_mean"].data_ptr() == sd["conv_blocks.0.batch_norm1.running_mean"].data_ptr()
    # the eval plan must pick up the new running statistics (they were rewritten in place by the kernels)
    m.eval()
    with torch.no_grad():
        got = m(x.to(dev), t.to(dev), lr.to(dev), 2)
        want = U.unet_forward({k: v.cpu() for k, v in sd.items()}, x, t, lr, 2)
    _assert_close(got, want, _tol(impl), "eval after train")


@pytest.mark.parametrize("impl", IMPLS)
def test_train_step_gradients_golden(dev, seeded_sd, golden, impl):
    """Loop body of reference train_diffusion_superres.py:384-393 (forward in train mode, MSE, backward, Adam) against
    the reference's own autograd (G5): every parameter's gradient norm, two full gradients, the post-step deltas."""
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).train()
    m.hip_engine().set_impl(impl, train_impl=impl)
    x, t, lr = golden_inputs("g5", 4, 4, 3, 32, 2, 1500)
    noise = synthetic.tensor_normal("g5.noise", (4, 3, 32, 32)).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    opt.zero_grad()
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    pred = m(x.to(dev), t.to(dev), lr.to(dev), 2)
    loss = torch.nn.MSELoss()(pred, noise)
    loss.backward()
    assert abs(loss.item() - float(golden["g5_loss"])) <= 2e-3 * float(golden["g5_loss"])
    here = os.path.dirname(os.path.abspath(__file__))
    names = open(os.path.join(here, "golden", "g5_param_names.txt")).read().split()
    ref_norms = golden["g5_grad_norms"]
    scale = float(ref_norms.max())
    # split-bf16 (1e-5 per op) is amplified by the BatchNorm-backward cancellations along the ~25-layer chain: the
    # deepest gradients (LR encoder) are off by 4e-3 .. 1.3e-2 depending on the kernel schedule; the exact-fp32 kernels (the training default) stay below 2e-4
    rtol = 2e-4 if impl in ("direct", "mfma_f32") else 3e-2  # (1.3e-2 measured on LR_encoder.blocks.2.conv1.bias)
    bad = []
    params = dict(m.named_parameters())
    for name, ref in zip(names, ref_norms):
        g = params[name].grad
        if ref < 0:
            assert g is None, f"{name} is structurally unused (quirk Q3) and must get no gradient"
            continue
        assert g is not None, name
        got = g.norm().item()
        if abs(got - ref) > rtol * ref + 2e-6 * scale:
            bad.append((name, got, float(ref)))
    assert not bad, bad[:12]
    _assert_close(params["output.bias"].grad, torch.from_numpy(golden["g5_grad_output_bias"]), rtol, "grad output.bias")
    _assert_close(params["conv0.weight"].grad, torch.from_numpy(golden["g5_grad_conv0_weight"]), 5 * rtol, "grad conv0.weight")
    opt.step()
    ref_delta = golden["g5_delta_norms"]
    badd = []
    for name, ref, gref in zip(names, ref_delta, ref_norms):
        if gref < 1e-5 * scale:
            continue  # conv biases in front of a BatchNorm: the true gradient is 0, Adam amplifies rounding noise to +-lr
        got = (params[name].detach() - before[name]).norm().item()
        if abs(got - ref) > 0.02 * ref + 1e-9:  # Adam's first step is sign-like: only tiny gradients can flip
            badd.append((name, got, float(ref)))
    assert len(badd) <= 4, badd[:12]


def test_train_step_config2_per_rank_shape(dev, seeded_sd):
    """BASELINE configs[2] at its full per-rank size: ONE training step on 16 images of 256x256 (train-mode forward with
    batch statistics, MSE, backward; loop body of reference train_diffusion_superres.py:378-393) against the CPU oracle's
    own autograd on the SAME batch: prediction, loss and the gradient of every live parameter (norms, plus three full
    tensors from the top, the middle and the bottom of the network)."""
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    from oracle import unet_oracle as U
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).train()
    # (the shipped training default: exact-fp32 MFMA kernels unless DRS_TRAIN_IMPL says otherwise)
    x, t, lr = golden_inputs("cfg2", 16, 16, 3, 256, 2, 1500)
    noise = synthetic.tensor_normal("cfg2.noise", (16, 3, 256, 256))
    pred = m(x.to(dev), t.to(dev), lr.to(dev), 2)
    loss = torch.nn.MSELoss()(pred, noise.to(dev))
    loss.backward()
    assert torch.isfinite(loss).item()
    # oracle: the same step with torch autograd on the host (about 10 s on the box's 16 cores)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    live = {k for k, _ in m.named_parameters()}
    sd = {k: (v.clone().requires_grad_(True) if k in live and v.dtype.is_floating_point else v.clone())
          for k, v in seeded_sd.items()}
    # the state_dict registers every BatchNorm twice (aliased keys): route both names to ONE leaf
    for k in list(sd):
        for a, b in ((".conv1.1.", ".batch_norm1."), (".conv2.1.", ".batch_norm2."), (".shortcut_conv.1.", ".shortcut_batch_norm.")):
            if a in k:
                sd[k] = sd[k.replace(a, b)]
    want = U.unet_forward(sd, x, t, lr, 2, training=True)
    ref_loss = torch.nn.functional.mse_loss(want, noise)
    ref_loss.backward()
    _assert_close(pred.detach(), want.detach(), 1e-4, "configs[2] train-mode prediction")
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    scale = max(v.grad.norm().item() for v in sd.values() if v.requires_grad and v.grad is not None)
    worst, bad = 0.0, []
    for name, p in m.named_parameters():
        ref = sd[name].grad
        if ref is None:
            assert p.grad is None, f"{name} is structurally unused and must get no gradient"
            continue
        assert p.grad is not None, name
        got, rn = p.grad.norm().item(), ref.norm().item()
        if rn < 1e-5 * scale:  # conv biases in front of a BatchNorm: the true gradient is 0, both sides hold rounding noise
            assert got < 1e-4 * scale, (name, got, rn)
            continue
        dev_rel = abs(got - rn) / rn
        worst = max(worst, dev_rel)
        if dev_rel > 1e-3:
            bad.append((name, got, rn))
    print(f"configs[2] train step: loss {loss.item():.6f} vs {ref_loss.item():.6f}, worst gradient-norm deviation {worst:.2e}")
    assert not bad, bad[:10]
    # element-wise: two fp32 implementations of a 16k..1M-term reduction behind BatchNorm-backward cancellations differ by
    # ~1e-3 in the deep layers (2.5e-3 max-rel measured on bottle_neck.conv2.0.weight); the norms above are held to 1e-3
    for name, tol in (("output.weight", 2e-3), ("bottle_neck.conv2.0.weight", 8e-3), ("conv0.weight", 8e-3)):
        e_max, e_l2 = rel_errors(dict(m.named_parameters())[name].grad.cpu(), sd[name].grad)
        print(f"  grad {name}: max-rel {e_max:.2e} rel-L2 {e_l2:.2e}")
        assert e_max <= tol and e_l2 <= tol, (name, e_max, e_l2)


def test_diffusion_train_loop_end_to_end(dev, seeded_sd, tmp_path):
    """Diffusion.train (reference :319-511) on synthetic patches: loss goes down, EMA copy is what gets snapshotted,
    the snapshot has the reference's format and resumes."""
    from torch.utils.data import DataLoader
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion, SyntheticSuperresDataset
    torch.manual_seed(0)
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(seeded_sd)
    m = m.to(dev)
    snap = str(tmp_path / "snapshot.pt")
    d = Diffusion("cosine", m, snap, noise_steps=50, device=dev, magnification_factor=2, image_size=32,
                  Degradation_type="DownBlur", ema_smoothing=True)
    ds = SyntheticSuperresDataset(8, 3, 32, 2, seed=5)
    loader = DataLoader(ds, batch_size=4, shuffle=False)
    loss_fn = torch.nn.MSELoss()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    m.train()
    lr_b, hr_b = next(iter(loader))
    first = d.train_step(m, opt, loss_fn, lr_b, hr_b).item()
    for _ in range(15):
        last = d.train_step(m, opt, loss_fn, lr_b, hr_b).item()
    assert last < 0.8 * first, (first, last)
    d.train(lr=1e-3, epochs=2, check_preds_epoch=1, train_loader=loader, val_loader=loader, patience=5, loss="MSE",
            verbose=False)
    s = torch.load(snap)
    assert set(s) == {"MODEL_STATE", "EPOCHS_RUN"} and len(s["MODEL_STATE"]) == 299
    m2 = Residual_Attention_UNet_superres(3, 3, dev).to(dev)
    d2 = Diffusion("cosine", m2, snap, noise_steps=50, device=dev, magnification_factor=2, image_size=32,
                   Degradation_type="DownBlur")
    assert d2.epochs_run == s["EPOCHS_RUN"]
    out = d2.sample(1, m2, ds[0][0], input_channels=3)
    assert out.shape == (1, 3, 32, 32) and torch.isfinite(out).all()


def test_fused_adam_matches_torch_adam(dev):
    """FusedAdam (drs_adam_multi) against torch.optim.Adam (the reference's optimizer, train_diffusion_superres.py:337)
    over 6 steps on ragged tensors, including a parameter that never gets a gradient and one that skips a step."""
    from diffusionremotesensing_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(32, 16, 3, 3), (5,), (1,), (257, 100), (3, 3, 3, 3), (70001,)]
    ref_p = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=3e-4)
    mine = FusedAdam(my_p, lr=3e-4)
    for step in range(6):
        for i, (a, b) in enumerate(zip(ref_p, my_p)):
            if i == 2 or (i == 4 and step == 3):
                a.grad = None
                b.grad = None
                continue
            g = torch.randn(a.shape) * (10.0 ** (step - 3))
            a.grad = g.clone()
            b.grad = g.clone().to(dev)
        ref.step()
        mine.step()
    for a, b in zip(ref_p, my_p):
        assert torch.allclose(b.detach().cpu(), a.detach(), rtol=2e-6, atol=1e-7), (a.shape, (b.cpu() - a).abs().max())
    sa, sb = ref.state_dict()["state"], mine.state_dict()["state"]
    assert set(sa) == set(sb)  # parameter 2 never stepped: no state, like torch
    for k in sa:
        assert float(sa[k]["step"]) == float(sb[k]["step"])
        for name in ("exp_avg", "exp_avg_sq"):  # 1-2 ulp of the largest term (gradients span 6 decades here)
            a, b = sa[k][name], sb[k][name].cpu()
            assert (a - b).abs().max().item() <= 1e-6 * a.abs().max().item(), (k, name)
    with pytest.raises(RuntimeError):
        p = torch.nn.Parameter(torch.zeros(3))
        p.grad = torch.zeros(3)
        FusedAdam([p]).step()  # CPU parameter: no fallback


@pytest.mark.parametrize("weights", ["bounded", "untrained"])
def test_full_length_chain_is_deterministic(dev, seeded_sd, weights):
    """BASELINE configs[1] end to end: a complete T=1500 chain at B=16 256x256 (1499 UNet forwards with the cached
    conditioning branch, fused update) from the same device seed: finite, and bit for bit the same result - the eval path has
    no atomics.  "bounded": the `output` projection damped (conftest.longchain_state_dict), the chain's amplitude stays O(1)
    like a trained model's: every forward runs the default kernels (FL arithmetic on the wide 3x3 layers).  "untrained":
    the plain seeded weights - the chain's amplitude grows without bound (meaningless after 1499 steps, the test is about
    determinism and NaNs), leaves fp16's range on the way, the FL layers report it (DRS_ERR_RANGE at one of the chain's
    periodic checks) and the chain resumes on the split-bf16 kernels: the first run includes that hand-over, the second and
    third run split bf16 throughout and must agree bit for bit."""
    from conftest import longchain_state_dict
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    model = Residual_Attention_UNet_superres(3, 3, dev)
    model.load_state_dict(longchain_state_dict(seeded_sd) if weights == "bounded" else seeded_sd)
    model = model.to(dev).eval()
    model.hip_engine().set_impl("mfma_bf16x3")
    d = Diffusion("cosine", model, "/nonexistent/snapshot.pt", noise_steps=1500, device=dev, magnification_factor=2,
                  image_size=256, Degradation_type="DownBlur")
    lr1 = synthetic.tensor_uniform("chain.lr", (3, 128, 128))
    outs = []
    for _ in range(2 if weights == "bounded" else 3):
        torch.manual_seed(1234)
        torch.cuda.manual_seed(1234)
        outs.append(d.sample(16, model, lr1, input_channels=3))
        model.eval()
    assert outs[0].shape == (16, 3, 256, 256)
    assert not torch.isnan(outs[-1]).any()
    assert torch.equal(outs[-2], outs[-1])
    x = synthetic.tensor_normal("chain.x", (16, 3, 256, 256)).to(dev)
    with torch.no_grad():
        model(x, torch.full((16,), 700, device=dev), lr1.unsqueeze(0).to(dev), 2)
        log = model.hip_engine().logged_forward(x, torch.full((16,), 700, device=dev), lr1.unsqueeze(0).to(dev), 2,
                                                reuse_cond=True, check_weights=False)[1]
    fl = any("tapconv_fl_kernel" in k for _, k in log)
    assert fl == (weights == "bounded"), "bounded chains keep the FL kernels, the untrained chain must have handed over to split bf16"


def _alternating_forwards_are_bit_stable(eng, xs, ts, cond, mag, forwards, **kw):
    """tools/diag_determinism.py as a test: the eval forward alternates between two inputs (so that a read of the PREVIOUS
    forward's data would show), every output is compared bit for bit with the first run of its input; the comparison result
    stays on the device until the end (no synchronisation inside the loop)."""
    with torch.no_grad():
        eng.forward(xs[0], ts[0], cond, mag, reuse_cond=False, **kw)
        refs = [eng.forward(xs[i], ts[i], cond, mag, reuse_cond=True, check_weights=False, **kw).clone() for i in range(2)]
        bad = torch.zeros((), dtype=torch.int64, device=xs[0].device)
        for r in range(forwards):
            y = eng.forward(xs[r & 1], ts[r & 1], cond, mag, reuse_cond=True, check_weights=False, **kw)
            bad += (y != refs[r & 1]).any()
    eng.check_faults()  # no wave ran into its poll bound (csrc/sp_sync.h)
    return int(bad)


def test_ring_protocol_holds_over_60000_forwards(dev, seeded_sd_gen):
    """The weight-ring race of rounds 2 - 4 (a mover lifting a `landed` counter on behalf of a slower one: DESIGN.md 4.1) fired
    once in ~50 000 forwards, in the single-chunk ring-hit launch (conv_blocks.1.conv1: Cin = 32, several items per block).
    40 000 forwards of the super-resolution model at a shape with two such items per CU (its FL instance; the split-bf16
    instance runs the same mover rule in the variant tests) + 20 000 of the 64 x 64 generation model, whose bottleneck is the
    8 x 8 instance (tapconv_sp8_kernel): every output bit-equal to its reference."""
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    from diffusionremotesensing_amd.generate_new_imgs.UNet_model_generation import Residual_Attention_UNet_generation
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
    eng = m.to(dev).eval().hip_engine()
    eng.set_impl("mfma_bf16x3")
    lr = synthetic.tensor_uniform("ring.lr", (1, 3, 128, 128)).to(dev)
    xs = [synthetic.tensor_normal(f"ring.x{i}", (8, 3, 256, 256)).to(dev) for i in range(2)]  # conv_blocks.1.conv1: 8 x 64 patches = 512 items, two per block
    ts = [torch.full((8,), 700 + i, dtype=torch.int64, device=dev) for i in range(2)]
    assert _alternating_forwards_are_bit_stable(eng, xs, ts, lr, 2, 40000) == 0
    g = Residual_Attention_UNet_generation(3, 3, 10, dev)
    g.load_state_dict(seeded_sd_gen)
    geng = g.to(dev).eval().hip_engine()
    geng.set_impl("mfma_bf16x3")
    xs = [synthetic.tensor_normal(f"ring.g{i}", (16, 3, 64, 64)).to(dev) for i in range(2)]
    ts = [torch.full((16,), 300 + i, dtype=torch.int64, device=dev) for i in range(2)]
    labels = torch.arange(16, device=dev) % 10
    assert _alternating_forwards_are_bit_stable(geng, xs, ts, None, 1, 20000, labels=labels) == 0


@pytest.mark.parametrize("shape", [(3, 128), (1, 64), (2, 96)])
def test_forward_reads_nothing_it_did_not_write(dev, seeded_sd, shape):
    """The workspace is caller-owned, uninitialised memory.  Filled with NaN bit patterns (and then with +inf) before a
    forward, the result must be bit-identical to the first run's: a kernel that reads a location nobody wrote - and hides it
    behind a multiplication by zero, a ReLU or a masked store - shows up.  (Round 5: `upfuse_sp_kernel` multiplied the
    never-written rows 0 / OH-1 of its column edge vector by 0; finite garbage had hidden that since round 3.)"""
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    b, s_ = shape
    model = Residual_Attention_UNet_superres(3, 3, dev)
    model.load_state_dict(seeded_sd)
    model = model.to(dev).eval()
    eng = model.hip_engine()
    eng.set_impl("mfma_bf16x3")
    x = synthetic.tensor_normal("poison.x", (b, 3, s_, s_)).to(dev)
    lr = synthetic.tensor_uniform("poison.lr", (b, 3, s_ // 2, s_ // 2)).to(dev)
    t = torch.arange(b, device=dev) * 400 + 7
    with torch.no_grad():
        want = eng.forward(x, t, lr, 2, reuse_cond=False).clone()
        ws = eng._last_plan.workspace
        for pattern in (0x7FC00000, 0x7F800000, 0xFFFFFFFF):
            ws.view(torch.int32)[: ws.numel() // 4].fill_(pattern - (1 << 32) if pattern >= (1 << 31) else pattern)
            got = eng.forward(x, t, lr, 2, reuse_cond=False)
            assert torch.isfinite(got).all(), hex(pattern)
            assert torch.equal(got, want), hex(pattern)
    eng.check_faults()


def test_nan_reaches_the_output(dev, seeded_sd):
    """A NaN in one image of the batch (in x, and in the conditioning image) must come out of the network as a non-finite
    output of THAT image - through every ReLU (drs_maxf: IEEE maximum, like torch.relu; v_max_f32 would turn it into 0) and the
    ReLU-free epilogues - and leave the other images untouched (batch independence).  The FL kernel's movers count a NaN block
    as outside fp16's range: check_faults reports it (DRS_ERR_RANGE) and the split-bf16 kernels the plan falls back to carry
    the NaN through as well."""
    from diffusionremotesensing_amd import _lib, synthetic
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    model = Residual_Attention_UNet_superres(3, 3, dev)
    model.load_state_dict(seeded_sd)
    model = model.to(dev).eval()
    eng = model.hip_engine()
    eng.set_impl("mfma_bf16x3")
    x = synthetic.tensor_normal("nan.x", (3, 3, 128, 128))
    lr = synthetic.tensor_uniform("nan.lr", (3, 3, 64, 64))
    t = torch.tensor([5, 700, 1400])
    xn = x.clone()
    xn[1, 0, 40, 77] = float("nan")
    lrn = lr.clone()
    lrn[2, 1, 3, 9] = float("nan")
    with torch.no_grad():
        clean = model(x.to(dev), t.to(dev), lr.to(dev), 2).cpu()
        eng.check_faults()
        for attempt in ("FL kernels", "split-bf16 fallback"):
            got = model(xn.to(dev), t.to(dev), lr.to(dev), 2).cpu()
            assert not torch.isfinite(got[1]).all(), f"{attempt}: a NaN input pixel left no trace in its image's output"
            assert torch.equal(got[0], clean[0]) and torch.equal(got[2], clean[2]), attempt
            got = model(x.to(dev), t.to(dev), lrn.to(dev), 2).cpu()
            assert not torch.isfinite(got[2]).all(), f"{attempt}: a NaN in the conditioning image left no trace in its image's output"
            assert torch.equal(got[0], clean[0]) and torch.equal(got[1], clean[1]), attempt
            if attempt == "FL kernels":
                with pytest.raises(_lib.RangeFault):
                    eng.check_faults()
                clean = model(x.to(dev), t.to(dev), lr.to(dev), 2).cpu()  # (the reference of the fallback arithmetic)
            else:
                eng.check_faults()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DRS_SP": "0"}, {"DRS_SP": "0", "DRS_WS": "0"}, {"DRS_FUSE_GATE": "0"},
                                 {"DRS_CONCURRENT": "1"}, {"DRS_D3K": "0", "DRS_S2K": "0"}, {"DRS_UPFUSE": "0"},
                                 {"DRS_XT_ONLY": "0"}, {"DRS_RB0": "0"}, {"DRS_DOWNK": "0"}, {"DRS_SP8": "0"},
                                 {"DRS_FOLD_PROJ": "0"}, {"DRS_GATE_PSI": "0"}],
                         ids=["fp32-activations+ws", "fp32-activations+lockstep", "sp-unfused-gate", "sp-two-streams",
                              "sp-without-direct-kernels", "sp-unfused-up", "sp-plain-stage-inputs", "sp-two-launch-block0",
                              "sp-direct-operand-downs0", "sp-lockstep-8x8-level", "sp-projection-in-the-epilogue",
                              "sp-top-stage-with-att"])
def test_conv_kernel_variants_in_subprocess(env):
    """The kernel families of the eval split-bf16 plan are chosen once per process.  Default = SP-format activations with
    the wave-specialised SP kernel and the fused attention gate, serial stages; the switches select the older paths that
    the training / fp32 plans and small shapes still use: DRS_SP=0 fp32 channels-last activations (wave-specialised
    fp32-input kernel, DRS_WS=0: lock-step kernel only), DRS_FUSE_GATE=0 the
    five-launch attention gate, DRS_CONCURRENT=1 two-stream decoder stages, DRS_UPFUSE=0 the ConvTranspose + up_conv pair
    instead of the composite up-sampling kernel, DRS_XT_ONLY=0 decoder stage inputs stored twice (x and x + temb) instead of
    x + temb with a per-image gating bias, DRS_RB0=0 the first encoder block as two launches (conv1 + skip, conv2 + shortcut)
    instead of the fused kernel, DRS_SP8=0 the 8 x 8 level (the goldens' 64 x 64 images have one) on the lock-step kernel instead of
    the four-images-per-item instance of the wave-specialised one, DRS_DOWNK=0 downs.0 on the direct-operand stride-2 kernel
    instead of the LDS-staged one, DRS_D3K=0 / DRS_S2K=0 the plan without the direct-operand kernels (3x3 on the
    32-channel layers, stride-2 and transposed convolutions), DRS_FOLD_PROJ=0 the `output` projection as an epilogue of
    up_convs.2's two launches (what images under 64 rows always run) instead of folded into their weights, DRS_GATE_PSI=0 the
    top stage's gate writing `att` (and the att-half reading it) instead of stopping at psi.  All must reproduce the same goldens.  Own process, because the switches are read once."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, **env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", os.path.abspath(__file__), "-m", "gpu", "-k",
           "test_unet_blocks_golden or test_unet_forward_config1_vs_oracle or test_unet_forward_golden or "
           "(test_conv2d_flavours and mfma_bf16x3)"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DRS_TRAIN_BWD_IMPL": "mfma_f32", "DRS_TRAIN_WGRAD_IMPL": "mfma_f32"},
                                 {"DRS_TRAIN_BWD_IMPL": "mfma_f32"}, {"DRS_TRAIN_WGRAD_IMPL": "mfma_f32"}],
                         ids=["exact-fp32-backward", "exact-data-gradients", "exact-weight-gradients"])
def test_backward_product_arithmetic_variants(env):
    """Training default: exact-fp32 forward, backward PRODUCTS (data gradients: csrc/train_bwd.inc, weight gradients:
    csrc/wgrad_mfma_bf16.hip) on split bf16 with fp32 accumulation - the golden gradient test holds at its fp32 tolerance
    (2e-4) and the full-size configs[2] step at 1e-3 either way, because the gradient error of split-bf16 TRAINING comes from
    the forward activations (amplified by the BatchNorm-backward cancellations), not from the backward products.  The
    switches restore the exact-fp32 products; all combinations must pass the same tests.  Own process: read once."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, **env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", os.path.abspath(__file__), "-m", "gpu", "-k",
           "test_train_step_gradients_golden and mfma_f32"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.gpu
def test_ema_multi_tensor_matches_reference_formula(dev, seeded_sd):
    """EMA on device models = ONE `drs_ema_multi` launch: bit-identical to the reference's per-parameter
    `old * beta + (1 - beta) * new` (UNet_model_superres.py:18-30) and, during warm-up, to `load_state_dict`
    (parameters and buffers, including the int64 BatchNorm counters)."""
    import copy
    from diffusionremotesensing_amd.UNet_model_superres import EMA, Residual_Attention_UNet_superres
    from diffusionremotesensing_amd import synthetic
    a = Residual_Attention_UNet_superres(3, 3, dev)
    a.load_state_dict(seeded_sd)
    a = a.to(dev)
    b = Residual_Attention_UNet_superres(3, 3, dev)
    b.load_state_dict(synthetic.seeded_state_dict(b.state_dict(), 7))
    b = b.to(dev)
    for m in b.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.num_batches_tracked.fill_(12345678901)  # needs both 32-bit halves
    ema_model = copy.deepcopy(a).eval().requires_grad_(False)
    ema = EMA(0.995)
    ema.step_ema(ema_model, b, step_start_ema=1)  # warm-up: copy of the whole state_dict
    for (k, p), q in zip(ema_model.state_dict().items(), b.state_dict().values()):
        assert torch.equal(p, q), k
    old = [p.detach().clone() for p in ema_model.parameters()]
    ema.step_ema(ema_model, a, step_start_ema=1)
    for (name, p), o, new in zip(ema_model.named_parameters(), old, a.parameters()):
        assert torch.equal(p, o * 0.995 + (1 - 0.995) * new.detach()), name
    assert ema.step == 2
    # the folded weights of the EMA model follow the update (the engine watches the module's parameter epoch)
    x, t, lr = golden_inputs("g3", 2, 2, 3, 16, 2, 1500)
    with torch.no_grad():
        y1 = ema_model(x.to(dev), t.to(dev), lr.to(dev), 2).clone()
        ema.step_ema(ema_model, b, step_start_ema=1)
        y2 = ema_model(x.to(dev), t.to(dev), lr.to(dev), 2)
    assert not torch.equal(y1, y2)

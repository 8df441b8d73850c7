"""The FL arithmetic of the wide 3x3 layers (csrc/conv_mfma_fl.hip: fp16 main product + block-scaled fp6 cross terms) on weights
with the dynamic range of a TRAINED snapshot, and its range guards.  The seeded weights of the other tests keep every
BatchNorm statistic within 0.5 .. 1.5; a trained network folds running_var of 1e-3 and gamma of 10 into its kernels' operands
and feeds them activations in the hundreds - fp16's 5 exponent bits, unlike bf16's 8, have to be shown to hold that."""
import pytest
import torch

from conftest import golden_inputs, rel_errors

pytestmark = pytest.mark.gpu

TOL = 1e-4  # the bar of every eval forward of the default kernels (tests/test_gpu_parity.py: TOL_BF16X3)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from diffusionremotesensing_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _calibrate(x, t, lr):
    from oracle import unet_oracle as U

    def run(sd):
        stats = {}
        with torch.no_grad():
            U.unet_forward(sd, x, t, lr, 2, training=True, stats=stats)
        out = {}
        for bn, (rm, rv) in stats.items():  # new = 0.9 old + 0.1 batch (momentum 0.1; running_var takes the unbiased batch variance)
            out[bn] = ((rm - 0.9 * sd[bn + ".running_mean"]) / 0.1, (rv - 0.9 * sd[bn + ".running_var"]) / 0.1)
        return out
    return run


def _model(dev, sd):
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    m = Residual_Attention_UNet_superres(3, 3, dev)
    m.load_state_dict(sd)
    return m.to(dev).eval()


def _template():
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    return Residual_Attention_UNet_superres(3, 3, "cpu").state_dict()


def _kernels(eng, x, t, lr):
    _, log = eng.logged_forward(x, t, lr, 2, reuse_cond=True, check_weights=False)
    return [k for _, k in log]


@pytest.mark.parametrize("bare_gain", [1.0, 300.0])
def test_fl_forward_on_trained_like_statistics(dev, bare_gain):
    """128 x 128, batch 2: every level with 16-row patches (64, 32, 16 rows) runs tapconv_fl_kernel.  bare_gain 300 puts the
    outputs of `downs.*` / `up_convs.*` - inputs of the FL layers - into the thousands."""
    from diffusionremotesensing_amd import synthetic
    from oracle import unet_oracle as U
    x, t, lr = golden_inputs("fl.tl", 2, 2, 3, 128, 2, 1500)
    sd = synthetic.trained_like_state_dict(_template(), _calibrate(x, t, lr), seed=3, bare_gain=bare_gain)
    rv = torch.cat([v.flatten() for k, v in sd.items() if k.endswith("running_var")])
    assert rv.min() < 2e-3, f"the fixture is meant to hold tiny running variances (min {rv.min():.2e})"
    taps = {}
    with torch.no_grad():
        want = U.unet_forward(sd, x, t, lr, 2, taps=taps)
    peak = max(float(v.abs().max()) for v in taps.values())
    assert peak < 3e4, f"fixture activations must stay inside fp16's range for this test (peak {peak:.3g})"
    if bare_gain > 1.0:
        assert peak > 2e3, f"fixture meant to reach the thousands (peak {peak:.3g})"
    m = _model(dev, sd)
    eng = m.hip_engine()
    eng.set_impl("mfma_bf16x3")
    with torch.no_grad():
        got = m(x.to(dev), t.to(dev), lr.to(dev), 2)
        names = _kernels(eng, x.to(dev), t.to(dev), lr.to(dev))
    eng.check_faults()
    assert sum("tapconv_fl_kernel" in k for k in names) >= 8, "the FL kernel did not run: " + ", ".join(sorted(set(names)))
    assert torch.isfinite(got).all()
    e_max, e_l2 = rel_errors(got.cpu(), want)
    assert e_max <= TOL and e_l2 <= TOL, f"trained-like forward (gain {bare_gain}): max-rel {e_max:.3e} rel-L2 {e_l2:.3e}"


def test_fl_range_fault_falls_back_to_split_bf16(dev):
    """Activations beyond fp16 in front of an FL layer whose weights are ordinary: the attention output (`result` + BatchNorm
    with gamma in the thousands: att ~ 1e5) feeding the att-half of `up_convs`, which has no BatchNorm behind it to shrink its
    folded weights.  (Everywhere else in this network a huge input meets weights folded with the matching huge variance, and
    the pack-time check has already kept that layer on split bf16.)  The movers flag the block, check_faults reports
    DRS_ERR_RANGE once and the plan continues on the split-bf16 kernels, inside the same tolerance."""
    from diffusionremotesensing_amd import _lib, synthetic
    from oracle import unet_oracle as U
    x, t, lr = golden_inputs("fl.of", 2, 2, 3, 128, 2, 1500)

    def tweak(sd):
        for i in (0, 1):
            for leaf in ("weight", "bias"):
                sd[f"attention_blocks.{i}.result.1.{leaf}"] = sd[f"attention_blocks.{i}.result.1.{leaf}"] * 3e3
    sd = synthetic.trained_like_state_dict(_template(), _calibrate(x, t, lr), seed=5, tweak=tweak)
    taps = {}
    with torch.no_grad():
        want = U.unet_forward(sd, x, t, lr, 2, taps=taps)
    peak = max(float(taps[f"attention_blocks.{i}"].abs().max()) for i in (0, 1))
    assert peak > 1e5, f"fixture meant to leave fp16's range (attention output peak {peak:.3g})"
    m = _model(dev, sd)
    eng = m.hip_engine()
    eng.set_impl("mfma_bf16x3")
    with torch.no_grad():
        m(x.to(dev), t.to(dev), lr.to(dev), 2)
    with pytest.raises(_lib.RangeFault):
        eng.check_faults()
    with torch.no_grad():
        got = m(x.to(dev), t.to(dev), lr.to(dev), 2)
        names = _kernels(eng, x.to(dev), t.to(dev), lr.to(dev))
    eng.check_faults()  # reported once; the plan is on the split-bf16 kernels now
    assert not any("tapconv_fl_kernel" in k for k in names)
    e_max, e_l2 = rel_errors(got.cpu(), want)
    assert e_max <= TOL and e_l2 <= TOL, f"fallback forward: max-rel {e_max:.3e} rel-L2 {e_l2:.3e}"


def test_fl_weight_range_check_keeps_layers_on_split_bf16(dev):
    """Folded weights fp16 cannot hold (two layers whose BatchNorm gamma is 1e-9: every folded weight far below fp16's normal
    range): the pack-time check leaves exactly those layers on the split-bf16 kernel, the others run the FL kernel, and the
    forward stays inside the tolerance.  (Weights ABOVE fp16's range cannot be had without activations above it: that side is
    the range fault of the test above.)"""
    from diffusionremotesensing_amd import synthetic
    from oracle import unet_oracle as U
    x, t, lr = golden_inputs("fl.wr", 2, 2, 3, 128, 2, 1500)
    sd = synthetic.seeded_state_dict(_template(), 7)
    for bn in ("conv_blocks.1.batch_norm2", "ups.1.batch_norm"):
        for key in list(sd):
            if synthetic.canonical_key(key) in (bn + ".weight", bn + ".bias"):
                sd[key] = sd[key] * 1e-9
    with torch.no_grad():
        want = U.unet_forward(sd, x, t, lr, 2)
    m = _model(dev, sd)
    eng = m.hip_engine()
    eng.set_impl("mfma_bf16x3")
    with torch.no_grad():
        got = m(x.to(dev), t.to(dev), lr.to(dev), 2)
        log = eng.logged_forward(x.to(dev), t.to(dev), lr.to(dev), 2, reuse_cond=True, check_weights=False)[1]
    eng.check_faults()
    by_op = {}
    for op, k in log:
        by_op.setdefault(op, []).append(k)
    assert any("tapconv_sp_kernel" in k for k in by_op.get("conv_blocks.1.conv2.0", [])), by_op.get("conv_blocks.1.conv2.0")
    assert any("tapconv_sp_kernel" in k for k in by_op.get("ups.1.conv", [])), by_op.get("ups.1.conv")
    assert any("tapconv_fl_kernel" in k for k in by_op.get("conv_blocks.2.conv2.0", [])), by_op.get("conv_blocks.2.conv2.0")
    e_max, e_l2 = rel_errors(got.cpu(), want)
    assert e_max <= TOL and e_l2 <= TOL, f"max-rel {e_max:.3e} rel-L2 {e_l2:.3e}"

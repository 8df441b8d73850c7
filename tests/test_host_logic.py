"""Host-side logic that needs no GPU: module surface, state_dict schema, schedules, CLI, error behaviour."""
import copy
import os

import pytest
import torch

from oracle import diffusion_oracle as D


def _model():
    from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
    return Residual_Attention_UNet_superres(3, 3, "cpu")


def test_state_dict_schema_matches_reference_listing():
    """299 keys / 176 parameters / 63 buffers; aliased BatchNorm registrations.  The parameter
    order was recorded from the imported reference by tools/make_golden.py."""
    m = _model()
    sd = m.state_dict()
    assert len(sd) == 299 and len(list(m.parameters())) == 176 and len(list(m.buffers())) == 63
    assert sum(p.numel() for p in m.parameters()) == 4383058
    for a, b in (("conv_blocks.0.batch_norm1.weight", "conv_blocks.0.conv1.1.weight"),
                 ("bottle_neck.batch_norm2.running_var", "bottle_neck.conv2.1.running_var"),
                 ("conv_blocks.2.shortcut_batch_norm.bias", "conv_blocks.2.shortcut_conv.1.bias")):
        assert sd[a].data_ptr() == sd[b].data_ptr()
    here = os.path.dirname(os.path.abspath(__file__))
    names = open(os.path.join(here, "golden", "g5_param_names.txt")).read().split()
    assert [k for k, _ in m.named_parameters()] == names
    assert sd["ups.0.transform.weight"].shape == (256, 256, 3, 3)
    assert sd["up_convs.2.weight"].shape == (32, 96, 3, 3)
    assert sd["conv_blocks.0.batch_norm1.num_batches_tracked"].dtype == torch.int64


def test_plan_param_names_exist_in_state_dict():
    import ctypes as C
    from diffusionremotesensing_amd import _lib
    lib = _lib.load()
    cfg = _lib.UNetConfig(2, 1, 3, 3, 64, 64, 2, 0, 1e-5, 0)
    h = C.c_void_p()
    assert lib.drs_unet_plan_create(C.byref(h), C.byref(cfg)) == 0
    sd = _model().state_dict()
    n = lib.drs_unet_num_params(h)
    names = [lib.drs_unet_param_name(h, i).decode() for i in range(n)]
    assert len(set(names)) == n
    for i, k in enumerate(names):
        assert k in sd and sd[k].numel() == lib.drs_unet_param_numel(h, i), k
    # every live parameter is consumed; only the 6 dead tensors of quirk Q3 are not
    live = {k for k, _ in _model().named_parameters()} - set(names)
    assert sorted(live) == sorted(f"{b}.conv_upsampled_lr_img.{w}" for b in ("conv_blocks.1", "conv_blocks.2", "bottle_neck")
                                  for w in ("weight", "bias"))
    assert lib.drs_unet_workspace_bytes(h) > 0 and lib.drs_unet_packed_bytes(h) > (4383058 - 387520) * 4
    lib.drs_unet_plan_destroy(h)
    bad = _lib.UNetConfig(2, 3, 3, 3, 64, 64, 2, 0, 1e-5, 0)
    assert lib.drs_unet_plan_create(C.byref(h), C.byref(bad)) == 2
    assert b"lr batch" in lib.drs_last_error()
    bad = _lib.UNetConfig(2, 2, 3, 3, 60, 64, 2, 0, 1e-5, 0)
    assert lib.drs_unet_plan_create(C.byref(h), C.byref(bad)) == 2


def test_schedules_match_oracle():
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
    m = _model()
    for kind, T in (("cosine", 50), ("cosine", 1500), ("linear", 1000)):
        d = Diffusion(kind, m, "/nonexistent/snapshot.pt", noise_steps=T, device="cpu", magnification_factor=2,
                      image_size=32, Degradation_type="DownBlur")
        a, ah, b = D.schedule(kind, T)
        assert torch.equal(d.alpha, a) and torch.equal(d.alpha_hat, ah) and torch.equal(d.beta, b)
    t = d.sample_timesteps(1000)
    assert t.dtype == torch.int64 and t.min() >= 1 and t.max() < 1000


def test_no_cpu_fallback():
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
    m = _model().eval()
    x = torch.zeros(1, 3, 16, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, torch.ones(1, dtype=torch.int64), torch.zeros(1, 3, 8, 8), 2)
    d = Diffusion("cosine", m, "/nonexistent/snapshot.pt", noise_steps=10, device="cpu", magnification_factor=2,
                  image_size=16, Degradation_type="DownBlur")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d.noise_images(x, torch.ones(1, dtype=torch.int64))
    with pytest.raises(ValueError):
        Diffusion("cosine", m, "/nonexistent/s.pt", device="cpu", Degradation_type="other").sample(1, m, torch.zeros(3, 8, 8))


def test_snapshot_roundtrip_and_deepcopy(tmp_path):
    from diffusionremotesensing_amd import synthetic
    from diffusionremotesensing_amd.train_diffusion_superres import Diffusion
    m = _model()
    m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 3))
    path = str(tmp_path / "snapshot.pt")
    d = Diffusion("cosine", m, path, noise_steps=10, device="cpu", magnification_factor=2, image_size=16,
                  Degradation_type="DownBlur")
    d._save_snapshot(7, m)
    snap = torch.load(path)
    assert set(snap) == {"MODEL_STATE", "EPOCHS_RUN"} and len(snap["MODEL_STATE"]) == 299
    m2 = _model()
    d2 = Diffusion("cosine", m2, path, noise_steps=10, device="cpu", magnification_factor=2, image_size=16,
                   Degradation_type="DownBlur")
    assert d2.epochs_run == 7
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    m3 = copy.deepcopy(m2)
    assert m3._hip_engine is None and len(m3.state_dict()) == 299


def test_ema_matches_reference_formula():
    from diffusionremotesensing_amd.UNet_model_superres import EMA
    a, b = _model(), _model()
    ema_model = copy.deepcopy(a).eval().requires_grad_(False)
    ema = EMA(0.995)
    ema.step_ema(ema_model, b, step_start_ema=1)  # warm-up: copy
    assert all(torch.equal(p, q) for p, q in zip(ema_model.state_dict().values(), b.state_dict().values()))
    old = [p.detach().clone() for p in ema_model.parameters()]
    ema.step_ema(ema_model, a, step_start_ema=1)
    for o, p, new in zip(old, ema_model.parameters(), a.parameters()):
        assert torch.allclose(p, o * 0.995 + (1 - 0.995) * new.detach(), rtol=1e-6, atol=1e-8)
    assert ema.step == 2


def test_cli_flags_are_the_reference_ones():
    from diffusionremotesensing_amd.train_diffusion_superres import build_arg_parser
    p = build_arg_parser()
    a = p.parse_args(["--image_size", "256", "--model_name", "m", "--loss", "MSE", "--magnification_factor", "2",
                      "--multiple_gpus", "True", "--ema_smoothing"])
    assert (a.epochs, a.batch_size, a.lr, a.noise_schedule, a.noise_steps, a.patience) == (501, 32, 3e-4, "cosine", 200, 10)
    assert a.multiple_gpus is True and a.ema_smoothing is True and a.generate_video is False
    assert a.UNet_type == "Residual Attention UNet" and a.Degradation_type == "DownBlur" and a.Blur_radius == "random"
    flags = {s for act in p._actions for s in act.option_strings}
    for f in ("--epochs --batch_size --image_size --lr --check_preds_epoch --noise_schedule --snapshot_name --model_name "
              "--noise_steps --patience --dataset_path --inp_out_channels --generate_video --loss --magnification_factor "
              "--UNet_type --Degradation_type --num_crops --multiple_gpus --ema_smoothing --Blur_radius").split():
        assert f in flags


# ---- SAR -> NDVI and generation variants ----
def _variant_models():
    from diffusionremotesensing_amd.UNet_model_SAR_TO_NDVI import Residual_Attention_UNet_SAR_TO_NDVI
    from diffusionremotesensing_amd.generate_new_imgs.UNet_model_generation import Residual_Attention_UNet_generation
    return {"sar": Residual_Attention_UNet_SAR_TO_NDVI(2, 1, "cpu"), "gen": Residual_Attention_UNet_generation(3, 3, 10, "cpu")}


def test_variant_state_dicts_match_reference_listing():
    """Parameter names/order recorded from the imported reference models by tools/make_golden_variants.py."""
    here = os.path.dirname(os.path.abspath(__file__))
    ms = _variant_models()
    for kind, fname, nkeys, nparams in (("sar", "g8_param_names.txt", 299, 176), ("gen", "g9_param_names.txt", 284, 161)):
        m = ms[kind]
        names = open(os.path.join(here, "golden", fname)).read().split()
        assert [k for k, _ in m.named_parameters()] == names
        assert len(m.state_dict()) == nkeys and len(names) == nparams
    assert ms["sar"].state_dict()["conv_SAR_img.weight"].shape == (16, 2, 3, 3)
    assert ms["sar"].state_dict()["conv_blocks.0.conv_SAR_img.weight"].shape == (32, 16, 3, 3)
    assert ms["gen"].state_dict()["label_emb.weight"].shape == (10, 100)
    assert "conv_blocks.2.conv_skip.bias" in ms["gen"].state_dict()


def test_variant_plans_bind_to_their_state_dicts():
    import ctypes as C
    from diffusionremotesensing_amd import _lib
    lib = _lib.load()
    ms = _variant_models()
    for kind, cfg in (("sar", _lib.UNetConfig(2, 2, 1, 1, 64, 64, 1, 2, 1e-5, 0, _lib.VARIANT_SAR_TO_NDVI, 2, 0)),
                      ("gen", _lib.UNetConfig(2, 2, 3, 3, 64, 64, 1, 2, 1e-5, 0, _lib.VARIANT_GENERATION, 0, 10))):
        h = C.c_void_p()
        assert lib.drs_unet_plan_create(C.byref(h), C.byref(cfg)) == 0, lib.drs_last_error()
        sd = ms[kind].state_dict()
        n = lib.drs_unet_num_params(h)
        names = [lib.drs_unet_param_name(h, i).decode() for i in range(n)]
        for i, k in enumerate(names):
            assert k in sd and sd[k].numel() == lib.drs_unet_param_numel(h, i), k
        skip = "conv_SAR_img" if kind == "sar" else "conv_skip"
        unused = {k for k, _ in ms[kind].named_parameters()} - set(names)
        assert sorted(unused) == sorted(f"{b}.{skip}.{w}" for b in ("conv_blocks.1", "conv_blocks.2", "bottle_neck")
                                        for w in ("weight", "bias"))
        lib.drs_unet_plan_destroy(h)
    h = C.c_void_p()
    bad = _lib.UNetConfig(2, 2, 1, 1, 64, 64, 2, 2, 1e-5, 0, _lib.VARIANT_SAR_TO_NDVI, 2, 0)
    assert lib.drs_unet_plan_create(C.byref(h), C.byref(bad)) == 2 and b"magnification" in lib.drs_last_error()
    bad = _lib.UNetConfig(2, 2, 3, 3, 64, 64, 1, 2, 1e-5, 0, 7, 0, 0)
    assert lib.drs_unet_plan_create(C.byref(h), C.byref(bad)) != 0


def test_variant_cli_flags_and_no_cpu_fallback():
    from diffusionremotesensing_amd import train_diffusion_SAR_TO_NDVI as S
    from diffusionremotesensing_amd.generate_new_imgs import train_diffusion_generation as G
    a = S.build_arg_parser().parse_args(["--image_size", "64", "--model_name", "m", "--loss", "MSE"])
    assert a.SAR_channels == 2 and a.NDVI_channels == 1 and a.noise_steps == 200 and not hasattr(a, "magnification_factor")
    g = G.build_arg_parser().parse_args(["--model_name", "m", "--loss", "MSE"])
    assert g.inp_out_channels == 3 and g.image_size is None and not hasattr(g, "SAR_channels")
    ms = _variant_models()
    with pytest.raises(RuntimeError):
        ms["sar"](torch.zeros(1, 1, 32, 32), torch.ones(1, dtype=torch.int64), torch.zeros(1, 2, 32, 32))
    with pytest.raises(RuntimeError):
        ms["gen"](torch.zeros(1, 3, 32, 32), torch.ones(1, dtype=torch.int64), torch.tensor([1]))


def test_video_maker_without_cv2(tmp_path):
    """generate_video=True must not fail for lack of cv2: frames are saved as a tensor instead."""
    from diffusionremotesensing_amd.video import video_maker
    frames = [torch.rand(2, 3, 8, 8) for _ in range(3)]
    path = str(tmp_path / "v.mp4")
    video_maker(frames, path, 10)
    saved = torch.load(path + ".frames.pt") if os.path.exists(path + ".frames.pt") else None
    assert saved is None or (saved.shape == (3, 8, 8, 3) and saved.dtype == torch.uint8)
    assert saved is not None or os.path.exists(path)


def test_image_folder_loader_decodes_like_the_reference_dataset(tmp_path):
    """`load_image_folder_u8` = Image.open in sorted(os.listdir) order + the launch transform's Resize((S, S)) (Pillow
    BILINEAR, what torchvision does on PIL images; reference utils.py:93-138, train_diffusion_superres.py:594-605), with
    DistributedSampler-like sharding."""
    import numpy as np
    from PIL import Image
    from diffusionremotesensing_amd.degradation import load_image_folder_u8
    rng = np.random.default_rng(3)
    imgs = {}
    for i, size in enumerate([(32, 32), (40, 48), (32, 32), (64, 64), (32, 32)]):
        a = rng.integers(0, 256, size + (3,), dtype=np.uint8)
        name = f"img_{9 - i}.png"  # written in reverse lexical order: the loader must sort
        Image.fromarray(a).save(tmp_path / name)
        imgs[name] = a
    got = load_image_folder_u8(str(tmp_path), 32)
    assert got.shape == (5, 3, 32, 32) and got.dtype == torch.uint8
    for k, name in enumerate(sorted(imgs)):
        y = Image.fromarray(imgs[name])
        if y.size != (32, 32):
            y = y.resize((32, 32), Image.BILINEAR)
        assert np.array_equal(got[k].numpy(), np.moveaxis(np.asarray(y), -1, 0)), name
    # two ranks: every second file, equal shard sizes (the odd one out is dropped)
    r0, r1 = load_image_folder_u8(str(tmp_path), 32, 0, 2), load_image_folder_u8(str(tmp_path), 32, 1, 2)
    assert r0.shape[0] == r1.shape[0] == 2
    assert torch.equal(r0, got[0::2][:2]) and torch.equal(r1, got[1::2][:2])
    with pytest.raises(ValueError):
        load_image_folder_u8(str(tmp_path))  # mixed sizes and no image_size
    gray = tmp_path / "gray"
    gray.mkdir()
    g = rng.integers(0, 256, (8, 8), dtype=np.uint8)
    Image.fromarray(g).save(gray / "a.png")
    one = load_image_folder_u8(str(gray), 8)
    assert one.shape == (1, 1, 8, 8) and np.array_equal(one[0, 0].numpy(), g)  # mode L: one channel, like ToTensor
    Image.fromarray(g.astype(np.uint16) * 200).save(gray / "b.png")  # mode I;16: not an 8-bit image
    with pytest.raises(ValueError):
        load_image_folder_u8(str(gray), 8)


def test_every_environment_switch_the_library_reads_is_documented():
    """INTEGRATION.md / README.md list exactly the names csrc passes to getenv (round 4's list still carried three retired
    switches and missed four new ones)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for path in glob.glob(os.path.join(root, "diffusionremotesensing_amd", "csrc", "*.*")):
        if path.endswith((".hip", ".h", ".inc")):
            names |= set(re.findall(r'getenv\("(DRS_[A-Z0-9_]+)"\)', open(path).read()))
    experiment_builds_only = {"DRS_SP_MAXBLOCKS", "DRS_RB0_DEBUG",  # behind #ifdef DRS_SP_TIMELINE
                              "DRS_X_NUM_CU"}  # behind #ifdef DRS_X_NUM_CU (tools/two_stream_probe.py)
    shipped = names - experiment_builds_only
    assert len(shipped) >= 15
    integration = open(os.path.join(root, "INTEGRATION.md")).read()
    readme = open(os.path.join(root, "README.md")).read()
    header = open(os.path.join(root, "include", "drs_hip.h")).read()
    for n in sorted(shipped):
        assert f"`{n}`" in integration, f"{n} is read by the library but missing from INTEGRATION.md"
        assert n in readme, f"{n} is read by the library but missing from README.md"
        assert n in header, f"{n} is read by the library but missing from include/drs_hip.h"
    listed = set(re.findall(r"`(DRS_[A-Z0-9_]+)`", integration[integration.index("kernel-family switches"):integration.index("A plan is driven")]))
    assert listed - experiment_builds_only <= names, f"INTEGRATION.md lists switches the library no longer reads: {sorted(listed - names)}"

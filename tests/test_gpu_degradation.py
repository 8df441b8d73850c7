"""drs_downblur_u8 (on-device DownBlur data feed) against Pillow fixtures and the integer oracle: bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from diffusionremotesensing_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def test_downblur_matches_pillow_fixtures(dev):
    from diffusionremotesensing_amd.degradation import downblur
    g = dict(np.load(os.path.join(HERE, "golden", "degradation_golden.npz")))
    for tag in sorted(k[:-3] for k in g if k.endswith("_hr")):
        hr, ref = g[tag + "_hr"], g[tag + "_lr"]
        h, w, m, r = g[tag + "_params"]
        planes = hr[None] if hr.ndim == 2 else np.moveaxis(hr, -1, 0)
        want = ref[None] if ref.ndim == 2 else np.moveaxis(ref, -1, 0)
        x, y = downblur(torch.from_numpy(np.ascontiguousarray(planes))[None].to(dev), int(m), float(r))
        assert x.shape == (1,) + want.shape, tag
        assert torch.equal(x[0].cpu(), torch.from_numpy(want.astype(np.float32) / np.float32(255))), tag
        assert torch.equal(y[0].cpu(), torch.from_numpy(planes.astype(np.float32) / np.float32(255))), tag


def test_downblur_batch_full_size_vs_oracle(dev):
    """BASELINE configs[1]/[2] feed shape: 16 x 3 x 256 x 256 uint8 -> 128 x 128, radius from the 'random' draw."""
    from diffusionremotesensing_amd.degradation import DeviceSuperresFeed, downblur
    from oracle import degradation_oracle as G
    rng = np.random.default_rng(7)
    hr = rng.integers(0, 256, (16, 3, 256, 256), dtype=np.uint8)
    for radius in (0.5, 1.1834, 1.5):
        x, y = downblur(torch.from_numpy(hr).to(dev), 2, radius)
        wx, wy = G.downblur(hr, 128, 128, radius)
        assert torch.equal(x.cpu(), torch.from_numpy(wx)) and torch.equal(y.cpu(), torch.from_numpy(wy)), radius
    feed = DeviceSuperresFeed(torch.from_numpy(hr).to(dev), 2, blur_radius="random", batch_size=6, shuffle=False)
    assert 0.5 <= feed.blur_radius <= 1.5 and len(feed) == 3
    batches = list(feed)
    assert [b[0].shape[0] for b in batches] == [6, 6, 4] and batches[0][0].shape[1:] == (3, 128, 128)
    wx, _ = G.downblur(hr[:6], 128, 128, feed.blur_radius)
    assert torch.equal(batches[0][0].cpu(), torch.from_numpy(wx))
    with pytest.raises(RuntimeError):
        downblur(torch.from_numpy(hr), 2, 0.5)  # CPU tensor: no fallback


def test_cli_launch_with_device_feed(dev, tmp_path, monkeypatch):
    """`python -m ...train_diffusion_superres` (reference CLI flags) end to end on the on-device DownBlur feed:
    one epoch, validation, snapshot in the reference's format, final sampling."""
    from diffusionremotesensing_amd import train_diffusion_superres as T
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    T.main(["--epochs", "1", "--batch_size", "4", "--image_size", "32", "--model_name", "cli_test", "--noise_steps", "10",
            "--loss", "MSE", "--magnification_factor", "2", "--dataset_path", "synthetic_u8:8", "--Degradation_type",
            "DownBlur", "--Blur_radius", "random", "--check_preds_epoch", "1", "--ema_smoothing", "True"])
    snap = torch.load(tmp_path / "models_run" / "cli_test" / "weights" / "snapshot.pt")
    assert set(snap) == {"MODEL_STATE", "EPOCHS_RUN"} and len(snap["MODEL_STATE"]) == 299
    res = torch.load(tmp_path / "models_run" / "cli_test" / "results" / "superres_results.pt")
    assert res.shape == (5, 3, 32, 32) and torch.isfinite(res).all()

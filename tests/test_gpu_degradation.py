"""drs_downblur_u8 / drs_add_noise_clip_f32 (on-device DownBlur and DownBlurNoise data feed) against Pillow fixtures, the
integer oracle and the reference-pinned noise oracle: bit-exact."""
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


def test_gauss_noise_feed_matches_reference_items(dev):
    """Degradation_type=DownBlurNoise: seeded like the reference's process (Python `random`, numpy's global generator),
    the feed's items equal `add_Gaussian_noise(downblur item, 2, 10)` bit for bit, item by item in batch order - all three
    noise branches occur among the 12 items.  The oracle is pinned to the reference function itself
    (tests/test_oracle_golden.py, tests/golden/degradation_noise_golden.npz)."""
    import random
    from diffusionremotesensing_amd.degradation import DeviceSuperresFeed, add_reference_noise
    from oracle import degradation_oracle as G
    rng = np.random.default_rng(11)
    hr = rng.integers(0, 256, (12, 3, 64, 48), dtype=np.uint8)
    feed = DeviceSuperresFeed(torch.from_numpy(hr).to(dev), 2, blur_radius=0.8, batch_size=5, shuffle=False, Gauss_noise=True)
    random.seed(3)
    np.random.seed(3)
    got = [b[0].cpu() for b in feed]
    random.seed(3)
    np.random.seed(3)
    wx, _ = G.downblur(hr, 24, 32, 0.8)  # (the reference's transposed size expression: (W // m, H // m) as (h, w))
    branches = set()
    want = []
    for i in range(12):
        state = (random.getstate(), np.random.get_state())
        random.randint(2, 10)
        r = np.random.rand()
        branches.add("color" if r > 0.6 else ("gray" if r < 0.4 else "cov"))
        random.setstate(state[0])
        np.random.set_state(state[1])
        want.append(G.add_gaussian_noise(wx[i], 2, 10))
    assert branches == {"color", "gray", "cov"}
    assert torch.equal(torch.cat(got), torch.from_numpy(np.stack(want)))
    with pytest.raises(RuntimeError):
        add_reference_noise(torch.zeros(1, 3, 4, 4))  # CPU tensor: no fallback


@pytest.mark.parametrize("degradation", ["DownBlur", "DownBlurNoise"])
def test_cli_launch_with_device_feed(dev, tmp_path, monkeypatch, degradation):
    """`python -m ...train_diffusion_superres` (reference CLI flags) end to end on the on-device DownBlur feed (with and
    without the Gauss_noise step): one epoch, validation, snapshot in the reference's format, final sampling."""
    from diffusionremotesensing_amd import train_diffusion_superres as T
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    T.main(["--epochs", "1", "--batch_size", "4", "--image_size", "32", "--model_name", "cli_test", "--noise_steps", "10",
            "--loss", "MSE", "--magnification_factor", "2", "--dataset_path", "synthetic_u8:8", "--Degradation_type",
            degradation, "--Blur_radius", "random", "--check_preds_epoch", "1", "--ema_smoothing", "True"])
    snap = torch.load(tmp_path / "models_run" / "cli_test" / "weights" / "snapshot.pt")
    assert set(snap) == {"MODEL_STATE", "EPOCHS_RUN"} and len(snap["MODEL_STATE"]) == 299
    res = torch.load(tmp_path / "models_run" / "cli_test" / "results" / "superres_results.pt")
    assert res.shape == (5, 3, 32, 32) and torch.isfinite(res).all()


def _write_folder(root, n_train, n_val, size=32, odd=None):
    """<root>/train_original and /val_original of RGB PNGs (one of them `odd`-sized: the launch transform resizes it)."""
    from PIL import Image
    rng = np.random.default_rng(11)
    for sub, n in (("train_original", n_train), ("val_original", n_val)):
        os.makedirs(os.path.join(root, sub))
        for i in range(n):
            s_ = odd if (odd and sub == "train_original" and i == 1) else (size, size)
            Image.fromarray(rng.integers(0, 256, s_ + (3,), dtype=np.uint8)).save(os.path.join(root, sub, f"p{i:03d}.png"))


def test_image_folder_feed_matches_oracle(dev, tmp_path):
    """The device feed over an image folder == the reference's dataset items: Image.open -> Resize((S, S)) -> bicubic down
    -> GaussianBlur -> ToTensor (utils.py:126-166 under train_diffusion_superres.py:594-605), bit for bit; the Pillow
    arithmetic after the decode comes from oracle/degradation_oracle.py (pinned by Pillow fixtures)."""
    from PIL import Image
    from diffusionremotesensing_amd.degradation import DeviceSuperresFeed, load_image_folder_u8
    from oracle import degradation_oracle as G
    _write_folder(str(tmp_path), 6, 2, size=32, odd=(48, 40))
    root = os.path.join(str(tmp_path), "train_original")
    u8 = load_image_folder_u8(root, 32)
    feed = DeviceSuperresFeed(u8.to(dev), 2, blur_radius=0.8, batch_size=4, shuffle=False)
    got = list(feed)
    decoded = []
    for name in sorted(os.listdir(root)):
        y = Image.open(os.path.join(root, name))
        if y.size != (32, 32):
            y = y.resize((32, 32), Image.BILINEAR)  # torchvision's Resize on a PIL image
        decoded.append(np.moveaxis(np.asarray(y), -1, 0))
    decoded = np.stack(decoded)
    wx, wy = G.downblur(decoded, 16, 16, 0.8)
    assert torch.equal(torch.cat([b[0] for b in got]).cpu(), torch.from_numpy(wx))
    assert torch.equal(torch.cat([b[1] for b in got]).cpu(), torch.from_numpy(wy))
    x1, y1 = feed.item(1)
    assert torch.equal(x1.cpu(), torch.from_numpy(wx[1])) and torch.equal(y1.cpu(), torch.from_numpy(wy[1]))


def test_cli_launch_on_image_folder(dev, tmp_path, monkeypatch):
    """The reference's CLI on the reference's dataset layout (`--dataset_path <dir>` with train_original/ and
    val_original/): one epoch, validation, snapshot, final sampling from the first five training images."""
    from diffusionremotesensing_amd import train_diffusion_superres as T
    data = tmp_path / "data"
    _write_folder(str(data), 8, 4, size=32)
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    T.main(["--epochs", "1", "--batch_size", "4", "--image_size", "32", "--model_name", "cli_folder", "--noise_steps", "10",
            "--loss", "MSE", "--magnification_factor", "2", "--dataset_path", str(data), "--Degradation_type", "DownBlur",
            "--Blur_radius", "0.5", "--check_preds_epoch", "1"])
    snap = torch.load(tmp_path / "models_run" / "cli_folder" / "weights" / "snapshot.pt")
    assert set(snap) == {"MODEL_STATE", "EPOCHS_RUN"} and len(snap["MODEL_STATE"]) == 299
    res = torch.load(tmp_path / "models_run" / "cli_folder" / "results" / "superres_results.pt")
    assert res.shape == (5, 3, 32, 32) and torch.isfinite(res).all()
    with pytest.raises(FileNotFoundError):
        T.main(["--epochs", "1", "--batch_size", "4", "--image_size", "32", "--model_name", "cli_folder", "--noise_steps", "10",
                "--loss", "MSE", "--magnification_factor", "2", "--dataset_path", str(tmp_path / "nowhere")])

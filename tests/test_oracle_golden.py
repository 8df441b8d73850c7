"""The CPU oracle against the golden vectors produced by the imported reference (tools/make_golden.py)."""
import numpy as np
import torch

from conftest import golden_inputs, replay_noise_source
from oracle import diffusion_oracle as D
from oracle import unet_oracle as U


def test_pos_encoding(golden):
    out = U.pos_encoding(torch.from_numpy(golden["g1_t"]), 100)
    assert torch.equal(out, torch.from_numpy(golden["g1_pos_encoding"]))


def test_schedules(golden):
    for kind, T in (("cosine", 50), ("cosine", 1000), ("cosine", 1500), ("linear", 1000)):
        a, ah, b = D.schedule(kind, T)
        assert torch.equal(a, torch.from_numpy(golden[f"g2_{kind}_{T}_alpha"]))
        assert torch.equal(ah, torch.from_numpy(golden[f"g2_{kind}_{T}_alpha_hat"]))
        assert torch.equal(b, torch.from_numpy(golden[f"g2_{kind}_{T}_beta"]))
    # quirk Q7: alpha_hat[0] == 1, beta[0] == 0, no clipping
    a, ah, b = D.schedule("cosine", 50)
    assert ah[0] == 1 and b[0] == 0 and b[-1] > 0.7


def test_block_taps(golden, seeded_sd):
    x, t, lr = golden_inputs("g3", 2, 2, 3, 16, 2, 1500)
    taps = {}
    with torch.no_grad():
        out = U.unet_forward(seeded_sd, x, t, lr, 2, taps=taps)
    assert torch.equal(out, torch.from_numpy(golden["g3_out"]))
    checked = 0
    for k, v in golden.items():
        if k.startswith("g3_tap_"):
            assert torch.equal(taps[k[len("g3_tap_"):]], torch.from_numpy(v)), k
            checked += 1
    assert checked == 20


def test_whole_forward(golden, seeded_sd):
    x, t, lr = golden_inputs("g4", 2, 2, 3, 64, 2, 1500)
    with torch.no_grad():
        assert torch.equal(U.unet_forward(seeded_sd, x, t, lr, 2), torch.from_numpy(golden["g4_out"]))
        assert torch.equal(U.unet_forward(seeded_sd, x, t, lr[:1], 2), torch.from_numpy(golden["g4_out_lr_broadcast"]))
        x, t, lr = golden_inputs("g4m4", 1, 1, 3, 64, 4, 1500)
        assert torch.equal(U.unet_forward(seeded_sd, x, t, lr, 4), torch.from_numpy(golden["g4_out_mag4"]))


def test_train_mode_forward(golden, seeded_sd):
    from diffusionremotesensing_amd import synthetic
    x, t, lr = golden_inputs("g5", 4, 4, 3, 32, 2, 1500)
    stats = {}
    with torch.no_grad():
        out = U.unet_forward(seeded_sd, x, t, lr, 2, training=True, stats=stats)
    assert torch.allclose(out, torch.from_numpy(golden["g5_out"]), rtol=0, atol=1e-6)
    noise = synthetic.tensor_normal("g5.noise", (4, 3, 32, 32))
    assert abs(torch.nn.functional.mse_loss(out, noise).item() - float(golden["g5_loss"])) < 1e-6
    for k in ("conv_blocks.0.batch_norm1", "bottle_neck.batch_norm2", "attention_blocks.2.result.1"):
        assert torch.allclose(stats[k][0], torch.from_numpy(golden[f"g5_rm_{k}"]), atol=1e-6)
        assert torch.allclose(stats[k][1], torch.from_numpy(golden[f"g5_rv_{k}"]), atol=1e-6)


def test_noise_images(golden):
    from diffusionremotesensing_amd import synthetic
    _, ah, _ = D.schedule("cosine", 1500)
    x0 = synthetic.tensor_uniform("g6.x0", (4, 3, 32, 32))
    out = D.noise_images(x0, torch.from_numpy(golden["g6_t"]), ah, torch.from_numpy(golden["g6_eps"]))
    assert torch.equal(out, torch.from_numpy(golden["g6_x_t"]))


def test_end_to_end_sample(golden, seeded_sd):
    from diffusionremotesensing_amd import synthetic
    a, ah, b = D.schedule("cosine", 50)
    lr1 = synthetic.tensor_uniform("g7.small.lr", (3, 32, 32))
    x = D.sample(U.OracleUNet(seeded_sd), 2, lr1, 50, a, ah, b, 2, 64, noise_source=replay_noise_source(1234))
    ref = torch.from_numpy(golden["g7_small_x"])
    assert torch.allclose(x, ref, rtol=0, atol=1e-5 * ref.abs().max().item())
    assert abs(x.double().sum().item() - golden["g7_small_checksum"][0]) < 1e-2


def test_full_length_chain(golden, seeded_sd):
    """G10: the reference's own `sample` over the full configs[1] schedule (T = 1500, 1499 updates,
    train_diffusion_superres.py:234-249) at 32x32, n = 2, damped output projection: final x and the states entering
    steps 1400 / 1000 / 500 / 100 / 1."""
    from conftest import LONGCHAIN_STEPS, longchain_state_dict
    from diffusionremotesensing_amd import synthetic
    a, ah, b = D.schedule("cosine", 1500)
    seen = {}
    inner = U.OracleUNet(longchain_state_dict(seeded_sd))

    class Recorder(torch.nn.Module):
        def forward(self, x, t, lr_img, mag):
            if int(t[0]) in LONGCHAIN_STEPS:
                seen[int(t[0])] = x.clone()
            return inner(x, t, lr_img, mag)

    lr1 = synthetic.tensor_uniform("g10.lr", (3, 16, 16))
    x = D.sample(Recorder(), 2, lr1, 1500, a, ah, b, 2, 32, noise_source=replay_noise_source(1010))
    for i in LONGCHAIN_STEPS:
        ref = torch.from_numpy(golden[f"g10_x_entering_{i}"])
        assert torch.allclose(seen[i], ref, rtol=0, atol=1e-6 * ref.abs().max().item()), i
    ref = torch.from_numpy(golden["g10_x"])
    assert torch.allclose(x, ref, rtol=0, atol=1e-6 * ref.abs().max().item())


# ---- SAR -> NDVI and class-conditional generation variants (tools/make_golden_variants.py) ----
def test_sar_variant(vgolden, seeded_sd_sar):
    from diffusionremotesensing_amd import synthetic
    x = synthetic.tensor_normal("g8.x", (2, 1, 64, 64))
    sar = synthetic.tensor_uniform("g8.sar", (2, 2, 64, 64))
    t = torch.from_numpy(vgolden["g8_t"])
    with torch.no_grad():
        assert torch.equal(U.unet_forward_sar(seeded_sd_sar, x, t, sar), torch.from_numpy(vgolden["g8_out"]))
        assert torch.equal(U.unet_forward_sar(seeded_sd_sar, x, t, sar[:1]), torch.from_numpy(vgolden["g8_out_bcast"]))
        x5 = synthetic.tensor_normal("g8t.x", (4, 1, 32, 32))
        sar5 = synthetic.tensor_uniform("g8t.sar", (4, 2, 32, 32))
        out = U.unet_forward_sar(seeded_sd_sar, x5, torch.from_numpy(vgolden["g8t_t"]), sar5, training=True, stats={})
    assert torch.allclose(out, torch.from_numpy(vgolden["g8t_out"]), rtol=0, atol=1e-6)
    a, ah, b = D.schedule("cosine", 30)
    sar1 = synthetic.tensor_uniform("g8s.sar", (2, 64, 64))
    xs = D.sample_sar(U.OracleUNetSAR(seeded_sd_sar), 2, sar1, 30, a, ah, b, 64, noise_source=replay_noise_source(808))
    ref = torch.from_numpy(vgolden["g8s_x"])
    assert torch.allclose(xs, ref, rtol=0, atol=1e-5 * ref.abs().max().item())


def test_generation_variant(vgolden, seeded_sd_gen):
    from diffusionremotesensing_amd import synthetic
    x = synthetic.tensor_normal("g9.x", (2, 3, 64, 64))
    t = torch.from_numpy(vgolden["g9_t"])
    y = torch.from_numpy(vgolden["g9_y"])
    with torch.no_grad():
        assert torch.equal(U.unet_forward_generation(seeded_sd_gen, x, t, y), torch.from_numpy(vgolden["g9_out_cond"]))
        unc = U.unet_forward_generation(seeded_sd_gen, x, t, None)
        assert torch.equal(unc, torch.from_numpy(vgolden["g9_out_uncond"]))
        # the model built without classes has the same weights minus label_emb
        assert torch.equal(unc, torch.from_numpy(vgolden["g9_out_noclass"]))
        x5 = synthetic.tensor_normal("g9t.x", (4, 3, 32, 32))
        out = U.unet_forward_generation(seeded_sd_gen, x5, torch.from_numpy(vgolden["g9t_t"]),
                                        torch.from_numpy(vgolden["g9t_y"]), training=True, stats={})
    assert torch.allclose(out, torch.from_numpy(vgolden["g9t_out"]), rtol=0, atol=1e-6)
    a, ah, b = D.schedule("cosine", 20)
    for tag, cfg, seed in (("cfg3", 3, 909), ("cfg0", 0, 910)):
        xs = D.sample_generation(U.OracleUNetGeneration(seeded_sd_gen), 2, torch.tensor([2, 5]), cfg, 20, a, ah, b, 32,
                                 noise_source=replay_noise_source(seed))
        ref = torch.from_numpy(vgolden[f"g9s_{tag}_x"])
        assert torch.allclose(xs, ref, rtol=0, atol=1e-5 * ref.abs().max().item()), tag


def test_full_length_guided_chain(vgolden, seeded_sd_gen):
    """G11: the reference's classifier-free-guided `sample` over configs[4]'s full schedule (cosine T = 1000: 999 x (two forwards,
    torch.lerp, update), train_diffusion_generation.py:229-259) at 32x32, n = 2, damped output projection."""
    from conftest import longchain_state_dict
    a, ah, b = D.schedule("cosine", 1000)
    x = D.sample_generation(U.OracleUNetGeneration(longchain_state_dict(seeded_sd_gen)), 2, torch.tensor([2, 5]), 3, 1000, a, ah, b, 32,
                            noise_source=replay_noise_source(1111))
    ref = torch.from_numpy(vgolden["g11_x"])
    assert torch.allclose(x, ref, rtol=0, atol=1e-6 * ref.abs().max().item())


def test_aggregation_sampling(vgolden, seeded_sd):
    """Tile split, Gaussian weights and blend of Aggregation_Sampling.py against what the imported reference class
    produced around the reference Diffusion.sample (G10), noise replayed in the reference's tile-major order."""
    from conftest import replay_tile_noise
    from diffusionremotesensing_amd import synthetic
    from oracle import aggregation_oracle as A
    infos, lr_origins = A.tile_infos(48, 56, 32, 16, 2)
    assert np.array_equal(np.array(infos, dtype=np.int32), vgolden["g10_infos"])
    w = A.gaussian_weight(64, 64)
    assert torch.equal(w, torch.from_numpy(vgolden["g10_weight"]))
    img = synthetic.tensor_uniform("g10.img", (1, 3, 48, 56))
    T = 8
    a, ah, b = D.schedule("cosine", T)
    src = replay_tile_noise(1010, len(infos), T, (1, 3, 64, 64))
    model = U.OracleUNet(seeded_sd)
    tiles = []
    for ti, (y0, x0) in enumerate(lr_origins):
        lr = img[0, :, y0:y0 + 32, x0:x0 + 32]
        tiles.append(D.sample(model, 1, lr, T, a, ah, b, 2, 64, noise_source=lambda i, shape, ti=ti: src(ti, i, shape)))
    out = A.aggregate(torch.cat(tiles), infos, w, 96, 112)
    ref = torch.from_numpy(vgolden["g10_result"])
    assert torch.allclose(out, ref, rtol=0, atol=1e-5)


def test_degradation_oracle_matches_pillow_fixtures():
    """The numpy restatement of Pillow's 8-bit bicubic resize + GaussianBlur against outputs of Pillow itself on the
    reference's call sequence (tools/make_golden_degradation.py): bit-exact."""
    import os
    from oracle import degradation_oracle as G
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "degradation_golden.npz")))
    tags = sorted(k[:-3] for k in g if k.endswith("_hr"))
    assert len(tags) == 7
    for tag in tags:
        hr, ref = g[tag + "_hr"], g[tag + "_lr"]
        h, w, m, r = g[tag + "_params"]
        planes = hr[None] if hr.ndim == 2 else np.moveaxis(hr, -1, 0)       # (C, H, W)
        want = ref[None] if ref.ndim == 2 else np.moveaxis(ref, -1, 0)
        out_h, out_w = int(w) // int(m), int(h) // int(m)                    # the reference's transposed size quirk
        assert want.shape[1:] == (out_h, out_w), tag
        x, y = G.downblur(planes, out_h, out_w, float(r))
        assert np.array_equal((x * 255).round().astype(np.uint8), want), tag
        assert np.array_equal(x, want.astype(np.float32) / np.float32(255)), tag
        assert np.array_equal(y, planes.astype(np.float32) / np.float32(255))
    # Gaussian radius -> box radius: values of BoxBlur.c's float formula
    assert abs(float(G.gaussian_box_radius(0.5)) - 0.0) < 0.3 and float(G.gaussian_box_radius(1.5)) > 0.9


def test_gauss_noise_oracle_matches_reference_function():
    """The oracle's `Gauss_noise=True` step against outputs of the reference's own add_Gaussian_noise (utils.py:15-38) on
    seeded inputs, one per branch (tools/make_golden_degradation_noise.py): bit-exact, so the generators are consumed in
    the reference's order."""
    import os
    import random
    from oracle import degradation_oracle as G
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "degradation_noise_golden.npz")))
    for kind in ("color", "gray", "cov"):
        seed = int(g[kind + "_seed"])
        random.seed(seed)
        np.random.seed(seed)
        got = G.add_gaussian_noise(g[kind + "_in"], 2, 10)
        assert got.dtype == np.float32 and np.array_equal(got, g[kind + "_out"]), kind
        assert got.min() >= 0.0 and got.max() <= 1.0


def test_feed_noise_term_consumes_the_generators_like_the_oracle():
    """Host half of the on-device DownBlurNoise feed (no GPU): `reference_noise` draws from Python's `random` and numpy's
    global generator exactly as the oracle does - x + noise, clipped, equals the oracle's item, and both leave the
    generators in the same state (so a batch of items stays aligned)."""
    import random
    from diffusionremotesensing_amd.degradation import reference_noise
    from oracle import degradation_oracle as G
    rng = np.random.default_rng(5)
    for seed in range(8):
        x = rng.random((3, 10, 14), dtype=np.float32)
        random.seed(seed)
        np.random.seed(seed)
        want = G.add_gaussian_noise(x, 2, 10)
        after_oracle = (random.random(), np.random.rand())
        random.seed(seed)
        np.random.seed(seed)
        noise = reference_noise(3, 10, 14, 2, 10)
        after_feed = (random.random(), np.random.rand())
        hwc = np.transpose(x, (1, 2, 0)).copy()
        hwc += noise
        got = np.transpose(np.clip(hwc, 0.0, 1.0), (2, 0, 1))
        assert noise.shape == (10, 14, 3) and noise.dtype == np.float32
        assert np.array_equal(got, want), seed
        assert after_oracle == after_feed, seed


def test_noise_prefetcher_draws_in_the_inline_order():
    """The DownBlurNoise feed's worker thread (degradation.NoisePrefetcher) consumes `random` / numpy's global generator exactly
    as the inline draws of the same batches do: same batches, bit for bit, including a ragged last batch and an early close."""
    import random

    from diffusionremotesensing_amd.degradation import NoisePrefetcher, reference_noise_batch
    sizes = [4, 4, 4, 3]
    random.seed(11)
    np.random.seed(12)
    want = [reference_noise_batch(n, 3, 8, 6, 2, 10) for n in sizes]
    random.seed(11)
    np.random.seed(12)
    pre = NoisePrefetcher(sizes, 3, 8, 6, 2, 10, depth=2)
    try:
        got = [pre.next() for _ in sizes]
    finally:
        pre.close()
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.equal(a, b)
    pre = NoisePrefetcher([2] * 50, 3, 8, 6)  # a consumer that stops after one batch: the worker ends with it
    pre.next()
    pre.close()
    assert not pre._t.is_alive()

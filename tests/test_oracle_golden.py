"""Pins oracle/ccdm_oracle.py (the CPU restatement) to outputs of the reference itself
(tests/golden/*.npz, written by tools/gen_goldens.py from /root/reference).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ccdm_oracle as O
from ccdm_stochastic_segmentation_amd.unet_spec import make_unet_spec, make_synthetic_state_dict
from tests.golden_util import BLOCK_CASES, HEAD_CASES, UPDOWN_BP, UPDOWN_CASES, block_tensors

LIDC_BP = dict(base_channels=32, channel_mult=None, attention_resolutions=[32, 16, 8], num_heads=1,
               num_head_channels=32, softmax_output=True)
LIDC_CFG = dict(num_heads=1, num_head_channels=32)


def lidc_sd(seed=0):
    spec = make_unet_spec(image_size=128, in_channels=3, out_channels=2, **LIDC_BP)
    return {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(spec, seed).items()}, spec


def unpack(bits, shape):
    return np.unpackbits(bits)[: int(np.prod(shape))].reshape(shape).astype(np.int64)


def test_g1_schedules_bit_exact(golden):
    g = golden["g1_schedules"]
    for name, sched, T in [("cosine250", "cosine", 250), ("cosine1000", "cosine", 1000), ("linear250", "linear", 250)]:
        b, a, c = O.make_schedule(sched, T, {"s": 0.008} if sched == "cosine" else None)
        assert np.array_equal(b.numpy(), g[name + "_betas"])
        assert np.array_equal(a.numpy(), g[name + "_alphas"])
        assert np.array_equal(c.numpy(), g[name + "_cumalphas"])
    # `s` is ignored by the reference (diffusion_denoising.py:27)
    assert torch.equal(O.cosine_schedule(250, s=0.5)[2], O.cosine_schedule(250)[2])
    for T, K in [(250, 10), (250, 25), (250, 200), (250, 150), (250, 100), (250, 50), (1000, 16)]:
        assert O.step_values(T, 10000 + K) == list(g[f"steps_T{T}_K{K}"])
    assert O.step_values(250, 10010) == [250, 222, 195, 167, 139, 112, 84, 56, 29, 1]
    assert O.step_values(250, None) == list(range(250, 0, -1))
    assert O.step_values(250, 10250) == list(range(250, 0, -1))
    assert O.step_values(250, 8) == [8, 7, 6, 5, 4, 3, 2, 1]


def test_g2_time_embedding(golden):
    g = golden["g2_time_embed"]
    sd, _ = lidc_sd()
    t = torch.from_numpy(g["t"])
    assert np.array_equal(O.timestep_embedding(t, 32).numpy(), g["emb32"])
    assert np.array_equal(O.timestep_embedding(t, 64).numpy(), g["emb64"])
    np.testing.assert_allclose(O.time_embed(sd, t).numpy(), g["time_embed"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag", list(BLOCK_CASES))
def test_g3_blocks(golden, tag):
    kind, kw, xs, seed = BLOCK_CASES[tag]
    shapes = golden.meta["block_shapes"][tag]
    w, x, emb = block_tensors(seed, shapes, xs)
    sd = {"b." + k: torch.from_numpy(v) for k, v in w.items()}
    x, emb = torch.from_numpy(x), torch.from_numpy(emb)
    if kind == "res":
        y = O.res_block(sd, "b.", x, emb)
    elif kind == "attn":
        y = O.attention_block(sd, "b.", x, kw["ch"] // 32, kw["new"])
    elif kind == "down":
        y = O.downsample(sd, "b.", x)
    else:
        y = O.upsample(sd, "b.", x)
    np.testing.assert_allclose(y.numpy(), golden["g3_blocks"][tag + ".y"], rtol=0, atol=2e-6)


def test_g4_unet_step(golden):
    g = golden["g4_unet_step_lidc"]
    sd, spec = lidc_sd()
    assert [[k, list(v.shape)] for k, v in sd.items()] == golden.meta["lidc_keys"]
    assert spec.num_params() == golden.meta["lidc_params"] == 5699138
    rng = np.random.default_rng(1234)
    image = torch.from_numpy(rng.uniform(-1, 1, (2, 1, 128, 128)).astype(np.float32))
    idx = torch.from_numpy(rng.integers(0, 2, (2, 128, 128)))
    assert np.array_equal(idx.numpy(), unpack(g["xt_idx"], (2, 128, 128)))
    taps = {}
    out = O.unet_forward(sd, LIDC_CFG, O.one_hot_bchw(idx, 2), image, None, torch.full((2,), float(g["t"])), taps)
    np.testing.assert_allclose(out["diffusion_out"].numpy(), g["out"], rtol=0, atol=1e-6)
    for k, v in taps.items():
        v = v.double()
        got = np.array([v.mean().item(), v.abs().mean().item(), v[0, 0, 0, 0].item(), v[-1, -1, -1, -1].item()])
        np.testing.assert_allclose(got, g["tap." + k], rtol=1e-5, atol=1e-5)
    # the synthetic weights must give a non-degenerate output (a fresh reference model emits 0.5 everywhere)
    assert g["out"].std() > 0.05


def test_g5_posterior(golden):
    g = golden["g5_posterior"]
    for K in (2, 20):
        for sched, T in (("cosine", 250), ("linear", 250)):
            _, alphas, cum = O.make_schedule(sched, T)
            xt = O.one_hot_bchw(torch.from_numpy(g[f"K{K}_{sched}_xt"]), K)
            x0 = torch.from_numpy(g[f"K{K}_{sched}_x0"])
            for t in (T, T // 2, 2, 1):
                a, c = O.posterior_coeffs(alphas, cum, t)
                ref = g[f"K{K}_{sched}_t{t}"]
                np.testing.assert_allclose(O.theta_post_prob_ref(xt, x0, a, c).numpy(), ref, rtol=0, atol=1e-7)
                fast = O.theta_post_prob(xt, x0, a, c).numpy()
                np.testing.assert_allclose(fast, ref, rtol=0, atol=2e-6)        # O(K) closed form (SURVEY: 9e-7)
                np.testing.assert_allclose(fast.sum(1), 1.0, atol=1e-5)
                if t == 1:                                                     # t=1 posterior == x0pred
                    np.testing.assert_allclose(fast, x0.numpy(), atol=1e-6)


def test_g6_sampler_bit_exact(golden):
    g = golden["g6_sampler"]
    for K in (2, 20):
        probs = torch.from_numpy(g[f"K{K}_probs"])
        p_hat = O.normalise_probs(probs)
        assert np.array_equal(p_hat.numpy(), g[f"K{K}_phat"])                      # bit-exact normalisation
        # explicit orders: for K <= 4 one order (bit-exact); for K > 4 the reference mixes two orders by pixel
        # position (see oracle.normalise_probs) — every pixel must match one of them.
        pcl = probs.permute(0, 2, 3, 1).contiguous()
        s_torch = probs.permute(0, 2, 3, 1).sum(-1)
        s_casc, s_rows = O.ordered_sum_lastdim(pcl), O.row_sum_order_lastdim(pcl)
        assert ((s_casc == s_torch) | (s_rows == s_torch)).all()
        p_casc = O.normalise_probs(probs, order="cascade")
        if K <= 4:
            assert torch.equal(p_casc, p_hat)
        else:
            assert (s_casc == s_torch).float().mean() > 0.5
            same = (s_casc == s_torch)
            assert torch.equal(p_casc[same], p_hat[same])
            np.testing.assert_allclose(p_casc.numpy(), p_hat.numpy(), rtol=3e-7)
        noise = torch.from_numpy(g[f"K{K}_noise"]).reshape(*p_hat.shape)
        assert np.array_equal(O.sample_index(p_hat, noise).numpy(), g[f"K{K}_idx"])  # bit-exact indices
        torch.manual_seed(6)
        assert torch.equal(O.draw_exponential(noise.reshape(-1, K).shape), noise.reshape(-1, K))
        assert np.array_equal(torch.nn.functional.one_hot(p_hat.argmax(-1), K).permute(0, 3, 1, 2).numpy(), g[f"K{K}_maxprob"])
        assert np.array_equal(p_hat.permute(0, 3, 1, 2).numpy(), g[f"K{K}_probsample"])
    torch.manual_seed(42)
    assert np.array_equal(O.draw_x_T(3, 2, 8, 8)[0].numpy(), g["xT_seed42_K2"])
    torch.manual_seed(42)
    assert np.array_equal(O.draw_x_T(2, 20, 8, 8)[0].numpy(), g["xT_seed42_K20"])
    torch.manual_seed(7)
    assert np.array_equal(torch.empty(64).exponential_(1).numpy(), g["exp_stream_seed7"])


def test_g7_trajectory(golden):
    """10 strided steps, seed 42: free-running oracle reproduces the reference's x_t bitmaps exactly."""
    g = golden["g7_trajectory_lidc"]
    sd, _ = lidc_sd()
    sched = O.make_schedule("cosine", 250, {"s": 0.008})
    image = torch.from_numpy(np.random.default_rng(1234).uniform(-1, 1, (2, 1, 128, 128)).astype(np.float32))
    for vote in ("confidence", "majority"):
        torch.manual_seed(42)
        idx, _ = O.draw_x_T(2, 2, 128, 128)
        assert np.array_equal(idx.numpy(), unpack(g["xT"], (2, 128, 128)))
        trace = []
        out = O.forward_denoising(sd, LIDC_CFG, sched, O.one_hot_bchw(idx, 2), image, None, 10010, vote, trace=trace)["diffusion_out"]
        assert [r["t"] for r in trace] == list(g["t_values"])
        xt = idx
        for j, r in enumerate(trace):
            assert np.array_equal(xt.numpy(), unpack(g[f"xt_{j}"], (2, 128, 128))), f"x_t differs at step {j}"
            np.testing.assert_allclose(r["x0pred"][:, 0, ::16, ::16].numpy(), g[f"x0pred0_{j}"], atol=2e-6)
            if "idx" in r:
                xt = r["idx"]
        if vote == "confidence":
            assert out.dtype == torch.float32 and tuple(out.stride()) == tuple(g["out_stride"])
            np.testing.assert_allclose(out[:, 0].numpy(), g["out_confidence_c0"], atol=2e-6)
            np.testing.assert_allclose(out[:, 1].double().sum().item(), float(g["out_confidence_c1_sum"]), rtol=1e-6)
        else:
            assert out.dtype == torch.int64
            assert np.array_equal(out.argmax(1).numpy(), unpack(g["out_majority"], (2, 128, 128)))


def test_g8_dino_step(golden):
    g = golden["g8_unet_step_dino"]
    fce = dict(type="dino", channels=384, output_stride=8, scale="single", target_layer=10)
    spec = make_unet_spec(image_size=64, in_channels=23, out_channels=20, feature_cond_encoder=fce,
                          **dict(LIDC_BP, channel_mult=[1, 1, 2, 2, 4, 4]))
    assert [[k, list(s)] for k, s in spec.param_shapes().items()] == golden.meta["dino_keys"]
    assert spec.feature_condition_idx == [10] and spec.feature_channels == 384
    sd = {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(spec, 8).items()}
    rng = np.random.default_rng(8)
    img = torch.from_numpy(rng.standard_normal((1, 3, 64, 128)).astype(np.float32))
    feat = torch.from_numpy(rng.standard_normal((1, 384, 8, 16)).astype(np.float32))
    idx = torch.from_numpy(rng.integers(0, 20, (1, 64, 128)))
    assert np.array_equal(idx.numpy().astype(np.uint8), g["xt_idx"])
    cfg = dict(LIDC_CFG, feature_condition_idx=[10])
    out = O.unet_forward(sd, cfg, O.one_hot_bchw(idx, 20), img, feat, torch.full((1,), float(g["t"])))
    np.testing.assert_allclose(out["diffusion_out"].numpy(), g["out"], rtol=0, atol=1e-6)


def k20_case():
    """the G15 workload (tools/gen_goldens_k20.py): the G8 network, N = 2, 64x128, K = 20, DINO features"""
    fce = dict(type="dino", channels=384, output_stride=8, scale="single", target_layer=10)
    spec = make_unet_spec(image_size=64, in_channels=23, out_channels=20, feature_cond_encoder=fce,
                          **dict(LIDC_BP, channel_mult=[1, 1, 2, 2, 4, 4]))
    sd = {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(spec, 8).items()}
    rng = np.random.default_rng(15)
    img = torch.from_numpy(rng.standard_normal((2, 3, 64, 128)).astype(np.float32))
    feat = torch.from_numpy(rng.standard_normal((2, 384, 8, 16)).astype(np.float32))
    return spec, sd, img, feat


def test_g15_trajectory_k20(golden):
    """K = 20 free-running, 10 strided steps, seed 42, DINO features (the reference's own normalisation order is position-dependent for
    K > 4): the oracle's seeded walk reproduces the reference's per-step class maps and final outputs."""
    g = golden["g15_trajectory_k20"]
    _, sd, img, feat = k20_case()
    cfg = dict(LIDC_CFG, feature_condition_idx=[10])
    sched = O.make_schedule("cosine", 250, {"s": 0.008})
    for vote in ("confidence", "majority"):
        torch.manual_seed(42)
        idx, _ = O.draw_x_T(2, 20, 64, 128)
        assert np.array_equal(idx.numpy(), g["xT"])
        trace = []
        out = O.forward_denoising(sd, cfg, sched, O.one_hot_bchw(idx, 20), img, feat, 10010, vote, trace=trace)["diffusion_out"]
        assert [r["t"] for r in trace] == list(g["t_values"])
        xt = idx
        for j, r in enumerate(trace):
            assert np.array_equal(xt.numpy(), g[f"xt_{j}"]), f"x_t differs at step {j}"
            np.testing.assert_allclose(r["x0pred"][:, :, ::8, ::8].numpy(), g[f"x0pred_lattice_{j}"], atol=2e-6)
            if "idx" in r:
                xt = r["idx"]
        if vote == "confidence":
            assert out.dtype == torch.float32 and tuple(out.stride()) == tuple(g["out_stride"])
            assert np.array_equal(out.argmax(1).numpy(), g["out_argmax"])
            np.testing.assert_allclose(out[:, :, ::4, ::4].numpy(), g["out_lattice"], atol=2e-6)
            np.testing.assert_allclose(out.double().sum((2, 3)).numpy(), g["out_class_sums"], rtol=1e-6)
        else:
            assert out.dtype == torch.int64 and np.array_equal(out.argmax(1).numpy(), g["out_majority"])


def k40_case():
    """the G18 walk (tools/gen_goldens_k40.py): a LIDC-shaped network with 40 classes, 32x32, N = 2"""
    spec = make_unet_spec(image_size=32, in_channels=43, out_channels=40, **dict(LIDC_BP, channel_mult=[1, 2, 4], attention_resolutions=[8]))
    sd = {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(spec, 18).items()}
    rng = np.random.default_rng(18)
    img = torch.from_numpy(rng.uniform(-1, 1, (2, 3, 32, 32)).astype(np.float32))
    return spec, sd, img


def test_g18_forty_classes(golden):
    """More than 32 classes, pinned on the reference itself (K = 40): its O(K^2) posterior, its sampler's normalisation and draws given
    the noise torch drew, and a seeded 6-step strided walk (per-step class maps, lattice outputs, final probabilities, majority map)."""
    g = golden["g18_k40"]
    K = 40
    _, alphas, cum = O.make_schedule("cosine", 250, {"s": 0.008})
    xt = O.one_hot_bchw(torch.from_numpy(g["post_xt"].astype(np.int64)), K)
    x0 = torch.from_numpy(g["post_x0"])
    for t in (250, 125, 2, 1):
        a, c = O.posterior_coeffs(alphas, cum, t)
        np.testing.assert_allclose(O.theta_post_prob_ref(xt, x0, a, c).numpy(), g[f"post_t{t}"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(O.theta_post_prob(xt, x0, a, c).numpy(), g[f"post_t{t}"], rtol=0, atol=2e-6)
    probs = torch.from_numpy(g["smp_probs"])
    p_hat = O.normalise_probs(probs)
    assert np.array_equal(p_hat.numpy(), g["smp_phat"])
    noise = torch.from_numpy(g["smp_noise"]).reshape(*p_hat.shape)
    assert np.array_equal(O.sample_index(p_hat, noise).numpy(), g["smp_idx"])
    assert np.array_equal(p_hat.argmax(-1).numpy(), g["smp_maxprob"])
    # the cascade order the HIP epilogue implements (two 16-blocks, then the 8-class tail): equal to torch's sum wherever torch took it
    pcl = probs.permute(0, 2, 3, 1).contiguous()
    s_torch = probs.permute(0, 2, 3, 1).sum(-1)                # (the channels-last VIEW the reference's Categorical sums over)
    s_casc, s_rows = O.ordered_sum_lastdim(pcl), O.row_sum_order_lastdim(pcl)
    assert ((s_casc == s_torch) | (s_rows == s_torch)).all() and (s_casc == s_torch).float().mean() > 0.5
    np.testing.assert_allclose(O.normalise_probs(probs, order="cascade").numpy(), p_hat.numpy(), rtol=3e-7)
    # the walk
    _, sd, img = k40_case()
    sched = O.make_schedule("cosine", 250, {"s": 0.008})
    for vote in ("confidence", "majority"):
        torch.manual_seed(42)
        idx, _ = O.draw_x_T(2, K, 32, 32)
        assert np.array_equal(idx.numpy(), g["walk_xT"])
        trace = []
        out = O.forward_denoising(sd, LIDC_CFG, sched, O.one_hot_bchw(idx, K), img, None, 10006, vote, trace=trace)["diffusion_out"]
        assert [r["t"] for r in trace] == list(g["walk_t_values"])
        cur = idx
        for j, r in enumerate(trace):
            assert np.array_equal(cur.numpy(), g[f"walk_xt_{j}"]), f"x_t differs at step {j}"
            np.testing.assert_allclose(r["x0pred"][:, :, ::4, ::4].numpy(), g[f"walk_x0pred_lattice_{j}"], atol=2e-6)
            if "idx" in r:
                cur = r["idx"]
        if vote == "confidence":
            assert np.array_equal(out.argmax(1).numpy(), g["walk_out_argmax"])
            np.testing.assert_allclose(out[:, :, ::2, ::2].numpy(), g["walk_out_lattice"], atol=2e-6)
            np.testing.assert_allclose(out.double().sum((2, 3)).numpy(), g["walk_out_class_sums"], rtol=1e-6)
        else:
            assert out.dtype == torch.int64 and np.array_equal(out.argmax(1).numpy(), g["walk_out_majority"])


def heads_meta():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meta_heads.json")) as f:
        return json.load(f)["block_shapes"]


@pytest.mark.parametrize("tag", list(HEAD_CASES))
def test_g16_attention_head_widths(golden, tag):
    """AttentionBlock with num_head_channels = -1 (the reference factory's default): heads = num_heads, width = channels / heads."""
    ch, nh, nhc, new, xs, seed = HEAD_CASES[tag]
    w, x, _ = block_tensors(seed, heads_meta()[tag], xs)
    sd = {"b." + k: torch.from_numpy(v) for k, v in w.items()}
    assert int(golden["g16_head_widths"][tag + ".heads"]) == nh
    y = O.attention_block(sd, "b.", torch.from_numpy(x), nh, new_order=new)
    np.testing.assert_allclose(y.numpy(), golden["g16_head_widths"][tag + ".y"], rtol=0, atol=2e-6)


def test_g16_unet_step_default_heads(golden):
    g = golden["g16_head_widths"]
    spec = make_unet_spec(image_size=128, in_channels=3, out_channels=2, **dict(LIDC_BP, num_heads=1, num_head_channels=-1))
    assert all(l.heads == 1 for _, l in spec.all_layers() if l.kind == "attn")
    sd = {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(spec, 16).items()}
    rng = np.random.default_rng(1616)
    image = torch.from_numpy(rng.uniform(-1, 1, (1, 1, 128, 128)).astype(np.float32))
    idx = torch.from_numpy(rng.integers(0, 2, (1, 128, 128)))
    out = O.unet_forward(sd, dict(num_heads=1, num_head_channels=-1), O.one_hot_bchw(idx, 2), image, None, torch.full((1,), float(g["unet_default_heads.t"])))
    np.testing.assert_allclose(out["diffusion_out"][:, 0].numpy(), g["unet_default_heads.out_c0"], rtol=0, atol=1e-6)


def updown_meta():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meta_updown.json")) as f:
        return json.load(f)


UPDOWN_CFG = dict(num_heads=1, num_head_channels=32, resblock_updown=True, num_res_blocks=2)


@pytest.mark.parametrize("tag", list(UPDOWN_CASES))
def test_g17_resblock_updown_blocks(golden, tag):
    """ResBlock(down=True) / ResBlock(up=True): AvgPool2d(2) / nearest x2 between in_layers' SiLU and its conv, and on the skip path."""
    ch, mode, film, xs, seed = UPDOWN_CASES[tag]
    w, x, emb = block_tensors(seed, updown_meta()["block_shapes"][tag], xs)
    sd = {"b." + k: torch.from_numpy(v) for k, v in w.items()}
    y = O.res_block(sd, "b.", torch.from_numpy(x), torch.from_numpy(emb), updown=mode)
    ref = golden["g17_resblock_updown"][tag + ".y"]
    want = (xs[0], ch, xs[2] // 2, xs[3] // 2) if mode == "down" else (xs[0], ch, 2 * xs[2], 2 * xs[3])
    assert tuple(y.shape) == ref.shape == want
    np.testing.assert_allclose(y.numpy(), ref, rtol=0, atol=2e-6)


def test_g17_updown_key_layout_and_unet_step(golden):
    """make_unet_spec(resblock_updown=True) reproduces the reference's state_dict (keys, shapes, order, parameter count); one step of
    that network through the oracle equals the reference's."""
    g, meta = golden["g17_resblock_updown"], updown_meta()
    spec = make_unet_spec(image_size=128, in_channels=3, out_channels=2, **UPDOWN_BP)
    assert [[k, list(v)] for k, v in spec.param_shapes().items()] == meta["unet_keys"]
    assert spec.num_params() == meta["unet_params"]
    assert [l.updown for _, l in spec.all_layers() if l.kind == "res" and l.updown] == ["down"] * 4 + ["up"] * 4
    assert not any(l.kind in ("down", "up") for _, l in spec.all_layers())
    sd = {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(spec, 17).items()}
    rng = np.random.default_rng(1717)
    image = torch.from_numpy(rng.uniform(-1, 1, (1, 1, 128, 128)).astype(np.float32))
    idx = torch.from_numpy(rng.integers(0, 2, (1, 128, 128)))
    taps = {}
    out = O.unet_forward(sd, UPDOWN_CFG, O.one_hot_bchw(idx, 2), image, None, torch.full((1,), float(g["unet.t"])), taps=taps)
    for k in ("input_blocks.3", "output_blocks.2"):
        assert tuple(taps[k].shape) == tuple(g[f"unet.tap.{k}.shape"])
        np.testing.assert_allclose(taps[k].mean((2, 3)).numpy(), g[f"unet.tap.{k}.mean_hw"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["diffusion_out"][:, 0].numpy(), g["unet.out_c0"], rtol=0, atol=1e-6)


def test_g17_updown_walk(golden):
    """6 strided steps, seed 42, through the resblock_updown network: the free-running oracle reproduces the reference's x_t bitmaps."""
    g = golden["g17_resblock_updown"]
    spec = make_unet_spec(image_size=128, in_channels=3, out_channels=2, **UPDOWN_BP)
    sd = {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(spec, 17).items()}
    sched = O.make_schedule("cosine", 250, {"s": 0.008})
    image = torch.from_numpy(np.random.default_rng(1717).uniform(-1, 1, (1, 1, 128, 128)).astype(np.float32))
    torch.manual_seed(42)
    idx, _ = O.draw_x_T(1, 2, 128, 128)
    assert np.array_equal(idx.numpy(), unpack(g["walk.xT"], (1, 128, 128)))
    trace = []
    out = O.forward_denoising(sd, UPDOWN_CFG, sched, O.one_hot_bchw(idx, 2), image, None, 10006, "confidence", trace=trace)["diffusion_out"]
    assert [r["t"] for r in trace] == list(g["walk.t_values"])
    xt = idx
    for j, r in enumerate(trace):
        assert np.array_equal(xt.numpy(), unpack(g[f"walk.xt_{j}"], (1, 128, 128))), f"x_t differs at step {j}"
        np.testing.assert_allclose(r["x0pred"][:, 0, ::16, ::16].numpy(), g[f"walk.x0pred0_{j}"], atol=2e-6)
        if "idx" in r:
            xt = r["idx"]
    np.testing.assert_allclose(out[:, 0].numpy(), g["walk.out_c0"], atol=2e-6)


def test_g9_caller_reenactment(golden):
    """evaluate_lidc_uncertainty.py:93-103: repeat_interleave ordering, x_T from the host generator,
    [B_img, S, K, H, W] reshape."""
    g = golden["g9_caller"]
    sd, _ = lidc_sd()
    sched = O.make_schedule("cosine", 250, {"s": 0.008})
    torch.manual_seed(0)
    img = torch.from_numpy(np.random.default_rng(16).uniform(-1, 1, (2, 1, 128, 128)).astype(np.float32))
    S = 2
    img_rep = img.repeat_interleave(S, dim=0)
    idx, _ = O.draw_x_T(4, 2, 128, 128)
    assert np.array_equal(idx.numpy(), unpack(g["xT"], (4, 128, 128)))
    out = O.forward_denoising(sd, LIDC_CFG, sched, O.one_hot_bchw(idx, 2), img_rep, None, 4, "confidence")["diffusion_out"]
    pred = out.reshape(2, -1, 2, 128, 128)
    assert list(pred.shape) == list(g["shape"])
    np.testing.assert_allclose(pred[:, :, 0].numpy(), g["pred_c0"], atol=2e-6)


def test_philox_known_answer():
    """Random123 known-answer vectors for philox4x32-10 (kat_vectors: zero, all-ones, pi digits)."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, exp in kat:
        got = O.philox4x32_10(np.array([ctr], dtype=np.uint32), np.array([key], dtype=np.uint32))[0]
        assert tuple(int(v) for v in got) == exp
    e = O.philox_exponential(seed=123, step=5, sample0=0, n=2, hw=64, k=20)
    assert e.shape == (2, 64, 20) and np.isfinite(e).all() and (e > 0).all()
    assert abs(e.mean() - 1.0) < 0.1


def _metric_case(g, tag):
    K, B, S, L, h, w = (int(v) for v in g[f"{tag}_shape"])
    rng = np.random.default_rng(int(g[f"{tag}_seed"]))
    lab = rng.integers(0, K, (B, L, h, w))
    smp = rng.integers(0, K, (B, S, h, w))
    if tag == "k2_empty":
        lab[0] = 0; smp[0, :2] = 0; smp[1] = 0
    return K, lab, smp


@pytest.mark.parametrize("tag", ["k2", "k2_empty", "k5"])
def test_g10_lidc_metrics(golden, tag):
    g = golden["g10_lidc_metrics"]
    K, lab, smp = _metric_case(g, tag)
    ged, de, ds = O.metrics_ged(lab, smp, K)
    assert np.array_equal(ged, g[f"{tag}_ged"]) and np.array_equal(de, g[f"{tag}_div_experts"]) and np.array_equal(ds, g[f"{tag}_div_samples"])
    lcm = np.lcm(smp.shape[1], lab.shape[1])
    hm = O.metrics_hungarian_iou(np.repeat(lab, lcm // lab.shape[1], 1), np.repeat(smp, lcm // smp.shape[1], 1), K)
    assert np.array_equal(np.array(hm), g[f"{tag}_hm_iou"])


# ------------------------------------------------------------------------------------------ training-time forward pieces (N3)
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_training_forward_pieces_match_reference(golden, tag):
    """q(x_t|x_0), q(x_t|x_{t-1}), theta_post, theta_post_prob (one-hot and soft x_t) and the clamped KL of train_step,
    per-sample t incl. t == 1 and t == T, against outputs of the reference itself (tools/gen_goldens_training.py)."""
    g = golden["g11_training_forward"]
    T, K, N, H, W = (int(v) for v in g[f"{tag}_cfg"])
    al, cu = torch.from_numpy(g[f"{tag}_alphas"]), torch.from_numpy(g[f"{tag}_cumalphas"])
    sched = O.make_schedule(str(g[f"{tag}_sched"]), T, {"s": 0.008} if str(g[f"{tag}_sched"]) == "cosine" else None)
    np.testing.assert_array_equal(sched[1].numpy(), al.numpy())
    np.testing.assert_array_equal(sched[2].numpy(), cu.numpy())
    t = torch.from_numpy(g[f"{tag}_t"])
    x0, xt, th = (torch.from_numpy(g[f"{tag}_{k}"]) for k in ("x0", "xt", "theta"))
    a, c = O.per_sample_coeffs(al, cu, t)
    assert float(a[0]) == 0.0 and float(c[0]) == 1.0          # t[0] == 1
    got = {
        "q_xt_given_x0": O.q_probs(x0, cu[t - 1]).permute(0, 2, 3, 1),
        "q_xt_given_xtm1": O.q_probs(x0, al[t - 1]).permute(0, 2, 3, 1),       # 1 - beta_t == alpha_t
        "theta_post": O.theta_post_t(xt, x0, a, c),
        "theta_post_prob": O.theta_post_prob_t(xt, th, a, c),
        "theta_post_prob_soft": O.theta_post_prob_t(th.roll(1, 0), th, a, c),
        "kl": O.kl_clamped(torch.from_numpy(g[f"{tag}_theta_post"]), torch.from_numpy(g[f"{tag}_theta_post_prob"])),
    }
    for name, v in got.items():
        np.testing.assert_allclose(v.numpy(), g[f"{tag}_{name}"], rtol=0, atol=5e-7, err_msg=name)


# ------------------------------------------------------------------------------------------ DINO ViT-S/8 key features (N4)
def test_dino_oracle_descriptor_layout():
    """PARITY UNPINNED (no reference output exists for the torch.hub network).  What can be pinned on the CPU is the reference's own
    wrapper arithmetic (dino.py:176-183,297-305): descriptor channel = d_index * heads + head, class token dropped, row-major patches;
    and that a 224x224 input uses the position embedding unresized."""
    from oracle import dino_oracle as D
    from ccdm_stochastic_segmentation_amd.dino import make_synthetic_vit_state_dict
    sd = {k: torch.from_numpy(v) for k, v in make_synthetic_vit_state_dict("dino_vits8", 1).items()}
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((1, 3, 32, 48)).astype(np.float32))
    desc = D.extract_key_descriptors(sd, x, layer=1)
    t = D.vit_block(sd, 0, D.vit_tokens(sd, x, 8), 6)
    k = D.vit_block_qkv(sd, 1, t).reshape(1, -1, 3, 6, 64)[:, :, 1]          # [B, t, h, d]
    for (yy, xx, h, d) in [(0, 0, 0, 0), (3, 5, 4, 17), (2, 1, 5, 63)]:
        assert desc[0, d * 6 + h, yy, xx].item() == pytest.approx(k[0, 1 + yy * 6 + xx, h, d].item(), abs=1e-6)
    assert D.interpolate_pos_encoding(sd["pos_embed"], 224, 224, 8) is sd["pos_embed"]
    assert D.interpolate_pos_encoding(sd["pos_embed"], 256, 512, 8).shape == (1, 1 + 32 * 64, 384)

#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  The fixtures are data:
seeded inputs + the reference's outputs.  Weights come from the build-owned numpy generator
(ccdm_stochastic_segmentation_amd.unet_spec.make_synthetic_state_dict) and are loaded into the reference
with load_state_dict(strict=True), which also pins the key/shape layout.

    python tools/gen_goldens.py            # rewrites tests/golden/

Import trick: ddpm/__init__.py pulls ignite/wandb/torchvision, so register an empty
package whose __path__ points at the reference and import only ddpm.models.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
pkg = types.ModuleType("ddpm")
pkg.__path__ = [os.path.join(REF, "ddpm")]
sys.modules["ddpm"] = pkg

from ddpm.models import build_model  # noqa: E402
from ddpm.models.diffusion_denoising import cosine_schedule, linear_schedule, DiffusionModel  # noqa: E402
from ddpm.models.one_hot_categorical import OneHotCategoricalBCHW  # noqa: E402
from ddpm.models.unet_openai.nn import timestep_embedding  # noqa: E402
from ddpm.models.unet_openai import unet as ref_unet  # noqa: E402

from ccdm_stochastic_segmentation_amd.unet_spec import make_unet_spec, make_synthetic_state_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)

LIDC_BP = dict(base_channels=32, channel_mult=None, attention_resolutions=[32, 16, 8], num_heads=1,
               num_head_channels=32, softmax_output=True)
DINO = dict(type="dino", model="dino_vits8", channels=384, conditioning="concat_pixels_concat_features",
            output_stride=8, scale="single", train=False, source_layer=11, target_layer=10)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KB")


def build(img_shape, lab_shape, bp, seed, time_steps=250, schedule="cosine", vote="confidence", fce=None):
    m = build_model(time_steps, schedule, {"s": 0.008} if schedule == "cosine" else None, [img_shape, lab_shape],
                    img_shape, "unet_openai", bp, "datasets.lidc", vote, fce)
    spec = make_unet_spec(image_size=min(img_shape[1:]), in_channels=lab_shape[0] + img_shape[0],
                          out_channels=lab_shape[0], num_res_blocks=2, cond_encoded_shape=img_shape,
                          feature_cond_encoder=fce, **bp)
    sd = make_synthetic_state_dict(spec, seed)
    m.unet.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.eval()
    return m, spec


def packbits(idx):  # class index (K=2) -> packed bitmap
    return np.packbits(idx.astype(np.uint8).reshape(-1))


@torch.no_grad()
def main():
    meta = {"torch": torch.__version__, "numpy": np.__version__}

    # ---- G1 schedules -------------------------------------------------------------------------
    g1 = {}
    for name, fn, T in [("cosine250", cosine_schedule, 250), ("cosine1000", cosine_schedule, 1000),
                        ("linear250", linear_schedule, 250)]:
        b, a, c = fn(T)
        g1[name + "_betas"], g1[name + "_alphas"], g1[name + "_cumalphas"] = b.numpy(), a.numpy(), c.numpy()
    # strided step lists (diffusion_denoising.py:178-187) re-enacted with the same expression
    for T, K in [(250, 10), (250, 25), (250, 200), (250, 150), (250, 100), (250, 50), (1000, 16)]:
        g1[f"steps_T{T}_K{K}"] = np.array([round(v) for v in np.linspace(T, 1, K)], dtype=np.int64)
    save("g1_schedules", **g1)

    # ---- G2 timestep embedding + time_embed ---------------------------------------------------
    m, spec = build((1, 128, 128), (2, 128, 128), LIDC_BP, seed=0)
    ts = torch.tensor([1.0, 2.0, 125.0, 250.0, 1000.0])
    save("g2_time_embed", t=ts.numpy(), emb32=timestep_embedding(ts, 32).numpy(),
         emb64=timestep_embedding(ts, 64).numpy(), time_embed=m.unet.time_embed(timestep_embedding(ts, 32)).numpy())
    meta["lidc_keys"] = [[k, list(v.shape)] for k, v in m.unet.state_dict().items()]
    meta["lidc_params"] = int(sum(p.numel() for p in m.unet.parameters()))

    # ---- G3 block goldens (N=2, small spatial): weights/inputs are seeded (tests/golden_util.py) ----
    from tests.golden_util import BLOCK_CASES, block_tensors
    g3 = {}
    meta["block_shapes"] = {}
    for tag, (kind, kw, xs, seed) in BLOCK_CASES.items():
        if kind == "res":
            blk = ref_unet.ResBlock(kw["cin"], 128, 0, out_channels=kw["cout"], use_scale_shift_norm=kw["film"])
        elif kind == "attn":
            blk = ref_unet.AttentionBlock(kw["ch"], num_heads=1, num_head_channels=32, use_new_attention_order=kw["new"])
        elif kind == "down":
            blk = ref_unet.Downsample(kw["ch"], True)
        else:
            blk = ref_unet.Upsample(kw["ch"], True)
        blk.eval()
        shapes = {k: list(v.shape) for k, v in blk.state_dict().items()}
        meta["block_shapes"][tag] = shapes
        w, x, emb = block_tensors(seed, shapes, xs)
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
        xt_ = torch.from_numpy(x)
        y = blk(xt_, torch.from_numpy(emb)) if kind == "res" else blk(xt_)
        g3[tag + ".y"] = y.numpy()
    save("g3_blocks", **g3)

    # ---- G4 whole U-Net step, LIDC cfg, N=2 ---------------------------------------------------
    rng4 = np.random.default_rng(1234)
    image = torch.from_numpy(rng4.uniform(-1, 1, (2, 1, 128, 128)).astype(np.float32))
    xt_idx = torch.from_numpy(rng4.integers(0, 2, (2, 128, 128)))
    xt = torch.nn.functional.one_hot(xt_idx, 2).permute(0, 3, 1, 2).float()
    taps = {}
    hooks = []
    for nm, mod in list(m.unet.input_blocks.named_children()):
        hooks.append(mod.register_forward_hook(lambda _m, _i, o, nm=nm: taps.__setitem__(f"input_blocks.{nm}", o)))
    hooks.append(m.unet.middle_block.register_forward_hook(lambda _m, _i, o: taps.__setitem__("middle_block", o)))
    for nm, mod in list(m.unet.output_blocks.named_children()):
        hooks.append(mod.register_forward_hook(lambda _m, _i, o, nm=nm: taps.__setitem__(f"output_blocks.{nm}", o)))
    out = m.unet(xt, image, None, torch.full((2,), 37.0))["diffusion_out"]
    for h in hooks:
        h.remove()
    g4 = {"xt_idx": packbits(xt_idx.numpy()), "t": np.array(37), "out": out.numpy(),
          "image_seed": np.array(1234)}
    for k, v in taps.items():
        v = v.double()
        g4["tap." + k] = np.array([v.mean().item(), v.abs().mean().item(), v[0, 0, 0, 0].item(), v[-1, -1, -1, -1].item()])
    save("g4_unet_step_lidc", **g4)

    # ---- G5 posterior -------------------------------------------------------------------------
    g5 = {}
    rng5 = np.random.default_rng(5)
    for K in (2, 20):
        for sched, T in (("cosine", 250), ("linear", 250)):
            d = DiffusionModel(sched, T, K)
            xt5 = torch.nn.functional.one_hot(torch.from_numpy(rng5.integers(0, K, (2, 8, 8))), K).permute(0, 3, 1, 2).float()
            x0 = torch.softmax(torch.from_numpy(rng5.standard_normal((2, K, 8, 8)).astype(np.float32) * 2), dim=1)
            for t in (T, T // 2, 2, 1):
                g5[f"K{K}_{sched}_t{t}"] = d.theta_post_prob(xt5, x0, torch.full((2,), t)).numpy()
            g5[f"K{K}_{sched}_xt"] = xt5.argmax(1).numpy()
            g5[f"K{K}_{sched}_x0"] = x0.numpy()
    save("g5_posterior", **g5)

    # ---- G6 sampler: (probs, seed) -> class indices + the noise torch drew ----------------------
    g6 = {}
    for K in (2, 20):
        rng6 = np.random.default_rng(60 + K)
        probs = torch.from_numpy(rng6.random((2, K, 12, 10)).astype(np.float32) ** 4)
        probs[0, :, 0, 0] = 0.0
        probs[0, 0, 0, 0] = 1.0           # a [1, 0, ...] pixel -> clamp path
        probs = torch.clamp(probs, min=1e-12)
        torch.manual_seed(6)
        dist = OneHotCategoricalBCHW(probs=probs)
        smp = dist.sample()
        torch.manual_seed(6)
        noise = torch.empty(2 * 12 * 10, K).exponential_(1)
        g6[f"K{K}_probs"] = probs.numpy()
        g6[f"K{K}_idx"] = smp.argmax(1).numpy()
        g6[f"K{K}_noise"] = noise.numpy()
        g6[f"K{K}_phat"] = dist.probs.numpy()                   # channels-last normalised
        g6[f"K{K}_maxprob"] = dist.max_prob_sample().numpy()
        g6[f"K{K}_probsample"] = dist.prob_sample().numpy()
    # uniform x_T draw as the callers do it
    torch.manual_seed(42)
    xT = OneHotCategoricalBCHW(logits=torch.zeros(3, 2, 8, 8)).sample()
    g6["xT_seed42_K2"] = xT.argmax(1).numpy()
    torch.manual_seed(42)
    xT = OneHotCategoricalBCHW(logits=torch.zeros(2, 20, 8, 8)).sample()
    g6["xT_seed42_K20"] = xT.argmax(1).numpy()
    torch.manual_seed(7)
    g6["exp_stream_seed7"] = torch.empty(64).exponential_(1).numpy()
    save("g6_sampler", **g6)

    # ---- G7 trajectory: t=10010 (10 strided steps), N=2, LIDC cfg, seed 42 ----------------------
    g7 = {}
    image7 = image
    for vote in ("confidence", "majority"):
        m.step_T_sample = vote
        torch.manual_seed(42)
        x = OneHotCategoricalBCHW(logits=torch.zeros(2, 2, 128, 128)).sample()
        rec = []
        orig = m.diffusion.theta_post_prob

        def spy(xt_, x0_, t_, rec=rec, orig=orig):
            rec.append((int(t_[0]), xt_.argmax(1).numpy().copy(), x0_.numpy().copy()))
            return orig(xt_, x0_, t_)
        m.diffusion.theta_post_prob = spy
        out7 = m(x, image7, t=torch.as_tensor(10010))["diffusion_out"]
        m.diffusion.theta_post_prob = orig
        if vote == "confidence":
            g7["t_values"] = np.array([r[0] for r in rec])
            g7["xT"] = packbits(x.argmax(1).numpy())
            for j, r in enumerate(rec):
                g7[f"xt_{j}"] = packbits(r[1])
                # x0pred channel 0 at a 16x strided lattice + mean, enough to pin each step's network output
                g7[f"x0pred0_{j}"] = r[2][:, 0, ::16, ::16]
            g7["out_confidence_c0"] = out7[:, 0].numpy()
            g7["out_confidence_c1_sum"] = np.array(out7[:, 1].double().sum().item())
            assert out7.dtype == torch.float32
            g7["out_stride"] = np.array(out7.stride())
        else:
            assert out7.dtype == torch.int64
            g7["out_majority"] = packbits(out7.argmax(1).numpy())
    m.step_T_sample = "confidence"
    save("g7_trajectory_lidc", **g7)

    # ---- A16 caller re-enactment (evaluate_lidc_uncertainty.py:93-103), B_img=2, S=2, t=4 --------
    torch.manual_seed(0)
    img_b = torch.from_numpy(np.random.default_rng(16).uniform(-1, 1, (2, 1, 128, 128)).astype(np.float32))
    labels = torch.zeros(2, 4, 2, 128, 128)
    S = 2
    img_rep = img_b.repeat_interleave(S, dim=0)
    x = OneHotCategoricalBCHW(logits=torch.zeros(labels[:, 0].repeat_interleave(S, dim=0).shape)).sample()
    pred = m(x, img_rep, t=torch.as_tensor(4))["diffusion_out"]
    pred = pred.reshape(labels.shape[0], -1, *labels.shape[2:])
    save("g9_caller", xT=packbits(x.argmax(1).numpy()), pred_c0=pred[:, :, 0].numpy(),
         pred_argmax=packbits(pred.argmax(2).numpy()), shape=np.array(pred.shape))

    # ---- G8 DINO-concat U-Net step at 64x128, K=20 (channel_mult given: image_size 64 default is 4-level)
    bp8 = dict(LIDC_BP, channel_mult=[1, 1, 2, 2, 4, 4])
    m8, spec8 = build((3, 64, 128), (20, 64, 128), bp8, seed=8, fce=DINO)
    rng8 = np.random.default_rng(8)
    img8 = torch.from_numpy(rng8.standard_normal((1, 3, 64, 128)).astype(np.float32))
    feat8 = torch.from_numpy(rng8.standard_normal((1, 384, 8, 16)).astype(np.float32))
    idx8 = torch.from_numpy(rng8.integers(0, 20, (1, 64, 128)))
    xt8 = torch.nn.functional.one_hot(idx8, 20).permute(0, 3, 1, 2).float()
    out8 = m8.unet(xt8, img8, feat8, torch.full((1,), 120.0))["diffusion_out"]
    save("g8_unet_step_dino", xt_idx=idx8.numpy().astype(np.uint8), out=out8.numpy(), t=np.array(120))
    meta["dino_keys"] = [[k, list(v.shape)] for k, v in m8.unet.state_dict().items()]

    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f)
    print("done")


if __name__ == "__main__":
    main()

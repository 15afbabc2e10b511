#!/usr/bin/env python3
"""G18: more than 32 classes (K = 40), from the REFERENCE itself (imported from /root/reference like tools/gen_goldens.py does).

  * posterior: theta_post_prob (the reference's O(K^2) form) at t in {250, 125, 2, 1};
  * sampler: OneHotCategoricalBCHW(probs).sample() with the noise torch drew, the normalised probabilities, the two last-step draws;
  * a seeded 6-step strided walk of a LIDC-shaped network with 40 classes on 32x32 images (stem 43 -> 32 channels on the general conv
    kernel, head 32 -> 40, the many-class epilogue): per-step class maps, lattice outputs, final probabilities and majority map.

    python tools/gen_goldens_k40.py        # rewrites tests/golden/g18_k40.npz   (build container only)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_goldens as G  # noqa: E402  (registers the reference's ddpm.models package)

K = 40


@torch.no_grad()
def main():
    g = {}
    # ---- posterior ----
    from ddpm.models.diffusion_denoising import DiffusionModel
    d = DiffusionModel("cosine", 250, K, schedule_params={"s": 0.008})
    rng = np.random.default_rng(180)
    xt = torch.nn.functional.one_hot(torch.from_numpy(rng.integers(0, K, (2, 8, 8))), K).permute(0, 3, 1, 2).float()
    x0 = torch.softmax(torch.from_numpy(rng.standard_normal((2, K, 8, 8)).astype(np.float32) * 2), dim=1)
    for t in (250, 125, 2, 1):
        g[f"post_t{t}"] = d.theta_post_prob(xt, x0, torch.full((2,), t)).numpy()
    g["post_xt"] = xt.argmax(1).numpy().astype(np.uint8)
    g["post_x0"] = x0.numpy()
    # ---- sampler ----
    rng6 = np.random.default_rng(60 + K)
    probs = torch.from_numpy(rng6.random((2, K, 12, 10)).astype(np.float32) ** 4)
    probs[0, :, 0, 0] = 0.0
    probs[0, 0, 0, 0] = 1.0
    probs = torch.clamp(probs, min=1e-12)
    torch.manual_seed(6)
    dist = G.OneHotCategoricalBCHW(probs=probs)
    smp = dist.sample()
    torch.manual_seed(6)
    noise = torch.empty(2 * 12 * 10, K).exponential_(1)
    g["smp_probs"] = probs.numpy()
    g["smp_idx"] = smp.argmax(1).numpy().astype(np.uint8)
    g["smp_noise"] = noise.numpy()
    g["smp_phat"] = dist.probs.numpy()
    g["smp_maxprob"] = dist.max_prob_sample().argmax(1).numpy().astype(np.uint8)
    # ---- walk ----
    bp = dict(G.LIDC_BP, channel_mult=[1, 2, 4], attention_resolutions=[8])
    m, _ = G.build((3, 32, 32), (K, 32, 32), bp, seed=18)
    rngw = np.random.default_rng(18)
    N, H, W = 2, 32, 32
    img = torch.from_numpy(rngw.uniform(-1, 1, (N, 3, H, W)).astype(np.float32))
    g["walk_N"] = np.array(N)
    for vote in ("confidence", "majority"):
        m.step_T_sample = vote
        torch.manual_seed(42)
        x = G.OneHotCategoricalBCHW(logits=torch.zeros(N, K, H, W)).sample()
        rec = []
        orig = m.diffusion.theta_post_prob

        def spy(xt_, x0_, t_, rec=rec, orig=orig):
            rec.append((int(t_[0]), xt_.argmax(1).numpy().astype(np.uint8).copy(), x0_.numpy().copy()))
            return orig(xt_, x0_, t_)
        m.diffusion.theta_post_prob = spy
        out = m(x, img, t=torch.as_tensor(10006))["diffusion_out"]
        m.diffusion.theta_post_prob = orig
        if vote == "confidence":
            g["walk_t_values"] = np.array([r[0] for r in rec])
            g["walk_xT"] = x.argmax(1).numpy().astype(np.uint8)
            for j, r in enumerate(rec):
                g[f"walk_xt_{j}"] = r[1]
                g[f"walk_x0pred_lattice_{j}"] = r[2][:, :, ::4, ::4]
            assert out.dtype == torch.float32
            g["walk_out_argmax"] = out.argmax(1).numpy().astype(np.uint8)
            g["walk_out_lattice"] = out[:, :, ::2, ::2].numpy()
            g["walk_out_class_sums"] = out.double().sum((2, 3)).numpy()
        else:
            assert out.dtype == torch.int64
            g["walk_out_majority"] = out.argmax(1).numpy().astype(np.uint8)
    G.save("g18_k40", **g)


if __name__ == "__main__":
    main()

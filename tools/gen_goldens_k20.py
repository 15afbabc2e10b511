#!/usr/bin/env python3
"""G15: a seeded 10-step strided sampling trajectory of the REFERENCE at K = 20 (Cityscapes-shaped: 64x128, 20 classes, DINO feature
concat, base 32, channel_mult [1,1,2,2,4,4] — the G8 network), imported from /root/reference like tools/gen_goldens.py does.

Pins the free-running loop where the reference's own normalisation order is position-dependent (K > 4, oracle/ccdm_oracle.py:379-394):
per-step class maps x_t (what theta_post_prob receives), the network output on a lattice, the final "confidence" probabilities (argmax
map + a stride-4 lattice of all classes + per-class fp64 sums) and the "majority" map.

    python tools/gen_goldens_k20.py        # rewrites tests/golden/g15_trajectory_k20.npz   (build container only)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_goldens as G  # noqa: E402  (registers the reference's ddpm.models package)


@torch.no_grad()
def main():
    bp8 = dict(G.LIDC_BP, channel_mult=[1, 1, 2, 2, 4, 4])
    m, _ = G.build((3, 64, 128), (20, 64, 128), bp8, seed=8, fce=G.DINO)
    rng = np.random.default_rng(15)
    N, K, H, W = 2, 20, 64, 128
    img = torch.from_numpy(rng.standard_normal((N, 3, H, W)).astype(np.float32))
    feat = torch.from_numpy(rng.standard_normal((N, 384, H // 8, W // 8)).astype(np.float32))
    g = {"image_seed": np.array(15), "N": np.array(N)}
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
        out = m(x, img, feat, t=torch.as_tensor(10010))["diffusion_out"]
        m.diffusion.theta_post_prob = orig
        if vote == "confidence":
            g["t_values"] = np.array([r[0] for r in rec])
            g["xT"] = x.argmax(1).numpy().astype(np.uint8)
            for j, r in enumerate(rec):
                g[f"xt_{j}"] = r[1]
                g[f"x0pred_lattice_{j}"] = r[2][:, :, ::8, ::8]           # all classes on every 8th pixel
            assert out.dtype == torch.float32
            g["out_argmax"] = out.argmax(1).numpy().astype(np.uint8)
            g["out_lattice"] = out[:, :, ::4, ::4].numpy()
            g["out_class_sums"] = out.double().sum((2, 3)).numpy()
            g["out_stride"] = np.array(out.stride())
        else:
            assert out.dtype == torch.int64
            g["out_majority"] = out.argmax(1).numpy().astype(np.uint8)
    G.save("g15_trajectory_k20", **g)


if __name__ == "__main__":
    main()

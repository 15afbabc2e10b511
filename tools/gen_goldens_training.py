#!/usr/bin/env python3
"""Golden vectors for the training-time forward pieces, produced by the REFERENCE itself:
DiffusionModel.q_xt_given_xtm1 / q_xt_given_x0 (probabilities), theta_post, theta_post_prob with per-sample t,
and the diffusion loss term of Trainer.train_step (kl_div on the clamped prediction).

    python tools/gen_goldens_training.py      # writes tests/golden/g11_training_forward.npz

Runs only in the build container (imports /root/reference; same import trick as tools/gen_goldens.py)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pkg = types.ModuleType("ddpm")
pkg.__path__ = [os.path.join(REF, "ddpm")]
sys.modules["ddpm"] = pkg
from ddpm.models.diffusion_denoising import DiffusionModel  # noqa: E402


def onehot(idx, K):
    return torch.nn.functional.one_hot(idx, K).permute(0, 3, 1, 2).float()


def main():
    out = {}
    for tag, (sched, T, K, N, H, W, params) in {
        "a": ("cosine", 250, 2, 5, 6, 7, {"s": 0.008}),        # LIDC: K=2
        "b": ("linear", 20, 5, 4, 3, 5, None),                 # small T, K=5
        "c": ("cosine", 1000, 20, 3, 4, 4, {"s": 0.008}),      # Cityscapes-like K
    }.items():
        g = torch.Generator().manual_seed(ord(tag) * 7 + K)
        dm = DiffusionModel(sched, T, K, schedule_params=params)
        t = torch.randint(1, T + 1, (N,), generator=g)
        t[0] = 1                      # the t == 1 branch (alphas -> 0, cumalphas_{t-1} -> 1)
        t[-1] = T
        x0 = onehot(torch.randint(0, K, (N, H, W), generator=g), K)
        xt = onehot(torch.randint(0, K, (N, H, W), generator=g), K)
        theta = torch.softmax(torch.randn((N, K, H, W), generator=g) * 2.0, dim=1)
        theta[1, :, 0, 0] = 0.0
        theta[1, 0, 0, 0] = 1.0       # a saturated prediction: the clamp at 1e-12 matters in the loss
        q0 = dm.q_xt_given_x0(x0, t).probs            # channels-last [N,H,W,K] inside the distribution object
        q1 = dm.q_xt_given_xtm1(x0, t).probs
        tp = dm.theta_post(xt, x0, t)
        tpp = dm.theta_post_prob(xt, theta, t)
        tpp_soft = dm.theta_post_prob(theta.roll(1, 0), theta, t)   # xt need not be one-hot for the formula
        kl = torch.nn.functional.kl_div(torch.log(torch.clamp(tpp, min=1e-12)), tp, reduction="none")
        for k, v in dict(t=t, x0=x0, xt=xt, theta=theta, q_xt_given_x0=q0, q_xt_given_xtm1=q1, theta_post=tp,
                         theta_post_prob=tpp, theta_post_prob_soft=tpp_soft, kl=kl).items():
            out[f"{tag}_{k}"] = v.numpy()
        out[f"{tag}_cfg"] = np.array([T, K, N, H, W])
        out[f"{tag}_sched"] = np.array(sched)
        out[f"{tag}_alphas"] = dm.alphas.numpy()
        out[f"{tag}_cumalphas"] = dm.cumalphas.numpy()
    path = os.path.join(ROOT, "tests", "golden", "g11_training_forward.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()

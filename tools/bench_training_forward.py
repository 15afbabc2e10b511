#!/usr/bin/env python3
"""Timing of the training-time forward pieces through DiffusionModel: algorithmic bytes / time vs the 8 TB/s HBM roofline.
Each kernel is one pass: q_xt reads K and writes K floats per pixel, theta_post(_prob) reads 2K and writes K, kl_clamped reads 2 writes 1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccdm_stochastic_segmentation_amd.models import DiffusionModel

DEV = torch.device("cuda:0")


def timeit(fn, iters=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for (N, K, H, W, T, name) in [(64, 2, 128, 128, 250, "LIDC C2 batch"), (64, 19, 256, 512, 1000, "Cityscapes-sized, 19 classes")]:
    dm = DiffusionModel("cosine", T, K, schedule_params={"s": 0.008}).to(DEV)
    g = torch.Generator().manual_seed(0)
    t = torch.randint(1, T + 1, (N,), generator=g).to(DEV)
    x0 = torch.nn.functional.one_hot(torch.randint(0, K, (N, H, W), generator=g), K).permute(0, 3, 1, 2).float().contiguous().to(DEV)
    xt = x0.roll(1, 0).contiguous()
    th = torch.softmax(torch.randn((N, K, H, W), generator=g), 1).to(DEV)
    el = N * K * H * W * 4
    tp, tpp = dm.theta_post(xt, x0, t), dm.theta_post_prob(xt, th, t)
    print(f"--- {name}: N={N} K={K} {H}x{W}  ({el / 1e6:.1f} MB per tensor)")
    for label, fn, nbytes in [("q_xt_given_x0 (probabilities)", lambda: dm._mix(x0, dm._per_sample(dm.cumalphas, t, N, DEV)), 2 * el),
                              ("theta_post", lambda: dm.theta_post(xt, x0, t), 3 * el),
                              ("theta_post_prob", lambda: dm.theta_post_prob(xt, th, t), 3 * el),
                              ("kl_clamped", lambda: dm.kl_clamped(tp, tpp), 3 * el)]:
        s = timeit(fn)
        print(f"  {label:32s} {s * 1e6:8.1f} us   {nbytes / s / 1e9:7.0f} GB/s   {nbytes / s / 8e12 * 100:5.1f} % of 8 TB/s")

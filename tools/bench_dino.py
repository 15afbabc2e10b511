#!/usr/bin/env python3
"""Timing of the DINO ViT-S/8 key-feature extractor on synthetic weights: ms per batch and achieved TFLOP/s
(2 flops per multiply-add of the linear layers and the attention products of blocks 0..10 plus block 11's norm1/qkv)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccdm_stochastic_segmentation_amd.dino import DinoViT, make_synthetic_vit_state_dict

DEV = torch.device("cuda:0")
enc = DinoViT("dino_vits8", False, "concat_pixels_concat_features", stride=8, state_dict=make_synthetic_vit_state_dict(seed=0))
for (N, H, W) in [(8, 256, 512), (64, 128, 128), (1, 256, 512)]:
    x = torch.randn((N, 3, H, W), device=DEV)
    for _ in range(2):
        enc(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 5
    e0.record()
    for _ in range(iters):
        enc(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    T, d = 1 + (H // 8) * (W // 8), 384
    lin = 11 * (d * 3 * d + d * d + 2 * d * 4 * d) + d * 3 * d + 192 * d        # MACs per token
    att = 11 * 2 * T * d                                                      # QK^T and PV MACs per token
    flops = 2.0 * N * T * (lin + att)
    print(f"N={N} {H}x{W} (T={T}): {ms:8.2f} ms per batch, {ms / N:7.2f} ms per image, {flops / ms / 1e9:7.1f} TFLOP/s")

# where the time goes (8 x 256x512): rocprofv3-free breakdown with events around each kind of launch
import collections
acc = collections.defaultdict(float)
ext = enc.extractor
orig = {k: getattr(ext, k) for k in ("_linear", "_layernorm", "_gelu", "_attention")}


def timed(name):
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig[name](*a, **k)
        e1.record()
        torch.cuda.synchronize()
        acc[name] += e0.elapsed_time(e1)
        return r
    return f


for k in orig:
    setattr(ext, k, timed(k))
x = torch.randn((8, 3, 256, 512), device=DEV)
enc(x)
print("breakdown, 8 x 256x512 (ms):", {k: round(v, 2) for k, v in acc.items()})

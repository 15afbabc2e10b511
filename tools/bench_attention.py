#!/usr/bin/env python3
"""Attention core at Cityscapes token counts (MFMA-bound at T >= 2048): time, algorithmic TFLOP/s (4*T^2*D per head,
fp32-equivalent) and the matrix-pipe share (every product is 3 fp16 MFMAs: x3 against the 2.5 PFLOP/s dense fp16 peak)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccdm_stochastic_segmentation_amd import hip

DEV = torch.device("cuda:0")
lib = hip.load()
for (N, T, C, heads) in [(64, 256, 96, 3), (16, 2048, 128, 4), (4, 8192, 128, 4), (8, 2049, 384, 6)]:
    qkv = torch.randn((N, T, 3 * C), device=DEV)
    out = torch.empty((N, T, C), device=DEV)
    run = lambda: hip.check(lib.ccdm_attention_ex(qkv.data_ptr(), out.data_ptr(), N, T, T, C, heads, 1, 0), "attention")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = 4.0 * N * heads * T * T * (C // heads)
    print(f"N={N} T={T} C={C} heads={heads} (D={C // heads}): {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s algorithmic, "
          f"{3 * flops / ms / 1e9 / 2500 * 100:5.1f} % of the fp16 matrix peak")

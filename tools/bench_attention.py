#!/usr/bin/env python3
"""Attention core at Cityscapes token counts (MFMA-bound at T >= 2048): time, algorithmic TFLOP/s (4*T^2*D per head,
fp32-equivalent) and the matrix-pipe share (every product is 3 fp16 MFMAs: x3 against the 2.5 PFLOP/s dense fp16 peak)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccdm_stochastic_segmentation_amd import hip

DEV = torch.device("cuda:0")
lib = hip.load()
for (N, T, C, heads) in [(64, 256, 96, 3), (4, 512, 256, 8), (4, 1024, 128, 4), (4, 2048, 128, 4), (16, 2048, 128, 4), (4, 8192, 128, 4), (8, 2049, 384, 6)]:
    qkv = torch.randn((N, T, 3 * C), device=DEV)
    out = torch.empty((N, T, C), device=DEV)
    # CCDM_EXPERIMENTS builds (CCDM_LIB=tools/ab/exp.so) also hold the pre-split experiment (tools/experiments/ccdm_attention_split.hip)
    exp = hasattr(lib, "ccdm_attention_ws") and not os.environ.get("NO_WS")
    nb = 0
    if exp:
        lib.ccdm_attention_workspace_bytes.restype = C.c_size_t
        lib.ccdm_attention_workspace_bytes.argtypes = [C.c_int] * 4
        lib.ccdm_attention_ws.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_size_t, C.c_void_p]
        nb = int(lib.ccdm_attention_workspace_bytes(N, T, C, heads))
    ws = torch.empty((max(nb, 16),), dtype=torch.uint8, device=DEV)
    if nb:
        run = lambda: hip.check(lib.ccdm_attention_ws(qkv.data_ptr(), out.data_ptr(), N, T, T, C, heads, 1, ws.data_ptr(), nb, 0), "attention")
    else:
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
    print(f"N={N} T={T} C={C} heads={heads} (D={C // heads}) {'pre-split' if nb else 'in-loop  '}: {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s algorithmic, "
          f"{3 * flops / ms / 1e9 / 2500 * 100:5.1f} % of the fp16 matrix peak")

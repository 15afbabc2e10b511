#!/usr/bin/env python3
"""Micro-benchmark of the conv kernel on the LIDC layer shapes (N=64), through the C ABI.
Prints per-shape time, algorithmic GB/s and TFLOP/s, and the weighted per-denoise-step total."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ccdm_stochastic_segmentation_amd import hip

DEV = torch.device("cuda:0")
N = int(os.environ.get("N", 64))
# (count per step, c0, c1, cout, H, W, k, stride, up, gn)
SHAPES = [
    (8, 32, 0, 32, 128, 128, 3, 1, 0, 1), (3, 32, 32, 32, 128, 128, 3, 1, 0, 1), (3, 32, 32, 32, 128, 128, 1, 1, 0, 0),
    (1, 32, 0, 32, 128, 128, 3, 2, 0, 0), (1, 4, 0, 32, 128, 128, 3, 1, 0, 0), (1, 32, 0, 2, 128, 128, 3, 1, 0, 1),
    (7, 32, 0, 32, 64, 64, 3, 1, 0, 1), (2, 32, 32, 32, 64, 64, 3, 1, 0, 1), (1, 64, 0, 64, 32, 32, 3, 1, 1, 0), (1, 64, 32, 32, 64, 64, 3, 1, 0, 1),
    (6, 64, 0, 64, 32, 32, 3, 1, 0, 1), (1, 96, 64, 64, 32, 32, 3, 1, 0, 1), (1, 64, 64, 64, 32, 32, 3, 1, 0, 1), (1, 32, 0, 64, 32, 32, 3, 1, 0, 1),
    (6, 96, 0, 96, 16, 16, 3, 1, 0, 1), (1, 128, 96, 96, 16, 16, 3, 1, 0, 1), (5, 96, 0, 288, 16, 16, 1, 1, 0, 1), (5, 96, 0, 96, 16, 16, 1, 1, 0, 0),
    (10, 128, 0, 128, 8, 8, 3, 1, 0, 1), (2, 128, 128, 128, 8, 8, 3, 1, 0, 1), (6, 128, 0, 384, 8, 8, 1, 1, 0, 1), (6, 128, 0, 128, 8, 8, 1, 1, 0, 0),
]


def run(prec, shape, iters=20, dbg=0):
    cnt, c0, c1, cout, H, W, k, stride, up, gn = shape
    lib = hip.load()
    g = torch.Generator(device="cpu").manual_seed(0)
    xa = torch.randn((N, H, W, c0), generator=g).to(DEV)
    xb = torch.randn((N, H, W, c1), generator=g).to(DEV) if c1 else None
    cin = c0 + c1
    w = (torch.randn((cout, cin, k, k), generator=g) / np.sqrt(cin * k * k)).numpy()
    wd = torch.from_numpy(hip.pack_conv_weight(w, k, hip.PREC_F16X3 if prec == hip.PREC_F16 else prec)).to(DEV)      # (the single-pass mode reads the hi halves of the F16X3 packing)
    prec = prec | (dbg << 8)
    bias = torch.zeros(cout, device=DEV)
    gam, bet = torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV)
    Hc, Wc = (2 * H, 2 * W) if up else (H, W)
    Ho, Wo = (Hc + 2 * (k // 2) - k) // stride + 1, (Wc + 2 * (k // 2) - k) // stride + 1
    out = torch.empty((N, Ho, Wo, cout), device=DEV)
    S = lib.ccdm_conv_slices(Ho, Wo, stride, k)
    ost = torch.empty((N, S, cout, 2), dtype=torch.float64, device=DEV)
    a = hip.ConvArgs()
    a.in0, a.C0 = xa.data_ptr(), c0
    if c1:
        a.in1, a.C1 = xb.data_ptr(), c1
    if gn:
        sa = torch.empty((N, 1, c0, 2), dtype=torch.float64, device=DEV)
        hip.check(lib.ccdm_gn_stats(xa.data_ptr(), N, H * W, c0, 1, sa.data_ptr(), 0))
        a.stats0, a.slices0 = sa.data_ptr(), 1
        if c1:
            sb = torch.empty((N, 1, c1, 2), dtype=torch.float64, device=DEV)
            hip.check(lib.ccdm_gn_stats(xb.data_ptr(), N, H * W, c1, 1, sb.data_ptr(), 0))
            a.stats1, a.slices1 = sb.data_ptr(), 1
        a.gamma, a.beta, a.act = gam.data_ptr(), bet.data_ptr(), hip.ACT_SILU
    a.eps = 1e-5
    a.N, a.Hin, a.Win, a.Hout, a.Wout = N, H, W, Ho, Wo
    a.ksize, a.stride, a.up = k, stride, up
    a.w, a.bias, a.Cout, a.prec = wd.data_ptr(), bias.data_ptr(), cout, prec
    a.emb_off = -1
    a.out, a.out_stats, a.out_slices = out.data_ptr(), ost.data_ptr(), S
    for _ in range(3):
        hip.check(lib.ccdm_conv2d(C.byref(a), 0), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.ccdm_conv2d(C.byref(a), 0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * N * Ho * Wo * cout * cin * k * k
    byts = 4.0 * N * (cin * H * W + cout * Ho * Wo)
    return ms, flops / ms / 1e9, byts / ms / 1e6


def timeline(prec, shape):
    """Phase durations (cycles) of one mid-grid block: stamps 2..7 per iteration = loop top | barrier A | commit | barrier B |
    issue | MFMA phase | (epilogue)."""
    run(prec, shape, iters=1, dbg=16 | int(os.environ.get("TLDBG", "0")))
    lib = hip.load()
    buf = (C.c_ulonglong * 1024)()
    hip.check(lib.ccdm_debug_read_timeline(buf, 1024))
    n = int(buf[1023])
    ev = [(int(buf[i]) >> 56, int(buf[i]) & ((1 << 56) - 1)) for i in range(min(n, 1020))]
    names = {12: "kernel-entry", 1: "start", 2: "top(prev phase end)", 3: "barrierA", 4: "commit", 5: "barrierB", 6: "issue", 7: "mfma", 8: "end", 9: "epi-barrier", 10: "epi-transpose", 11: "epi-rows", 13: "first-issue", 14: "gn-affine"}
    t0 = ev[0][1]
    prev = t0
    out = []
    for slot, t in ev[1:]:
        out.append(f"{names.get(slot, slot)}+{t - prev}")
        prev = t
    print(f"timeline total {prev - t0} cycles (100 MHz-ish s_memtime ticks): " + " ".join(out[:64]))


def run_fused_skip(prec, c0, c1, cout, H, W):
    """ResBlock tail: conv3x3(SiLU(GN(h))) + conv1x1([xa|xb]) + residual-free, one launch (tests/hip_util.conv2d does the packing)."""
    from tests import hip_util as U
    g = torch.Generator(device="cpu").manual_seed(1)
    h = torch.randn((N, H, W, cout), generator=g).to(DEV)
    xa = torch.randn((N, H, W, c0), generator=g).to(DEV)
    xb = torch.randn((N, H, W, c1), generator=g).to(DEV) if c1 else None
    w = (torch.randn((cout, cout, 3, 3), generator=g) / np.sqrt(cout * 9)).numpy()
    ws = (torch.randn((cout, c0 + c1, 1, 1), generator=g) / np.sqrt(c0 + c1)).numpy()
    st = [U.gn_stats(h, 1)]
    b = {"iters": 20}
    U.conv2d([h], w, np.zeros(cout, np.float32), 3, stats=st, gamma=np.ones(cout, np.float32), beta=np.zeros(cout, np.float32),
             act=hip.ACT_SILU, prec=prec, skip=([xa] + ([xb] if c1 else []), ws, np.zeros(cout, np.float32)), bench=b)
    return b["ms"]


def timeline_pc(prec, shape):
    """Producer/consumer kernel (CCDM_PC_TIMELINE=1): stamps of loader wave 0 and matrix wave 0 of one mid-grid block."""
    run(prec, shape, iters=1)
    lib = hip.load()
    buf = (C.c_ulonglong * 1024)()
    hip.check(lib.ccdm_debug_read_timeline(buf, 1024))
    names = {1: "start", 2: "prologue-issued", 3: "barA", 4: "commit0+issue", 5: "barB", 10: "commit", 11: "issueB+issue", 12: "rows", 14: "barrier", 15: "tail-rows",
             20: "mfma", 21: "acc->epi", 22: "barrier"}
    for role, off in (("loader", 0), ("matrix", 512)):
        n = int(buf[off])
        ev = [(int(buf[off + 1 + i]) >> 56, int(buf[off + 1 + i]) & ((1 << 56) - 1)) for i in range(n)]
        if not ev:
            print(role, "no stamps"); continue
        prev = ev[0][1]
        out = []
        for slot, t in ev[1:]:
            out.append(f"{names.get(slot, slot)}+{t - prev}")
            prev = t
        print(f"{role}: total {prev - ev[0][1]} cycles: " + " ".join(out[:90]))


if __name__ == "__main__":
    precs = [hip.PREC_F16X3] if len(sys.argv) < 2 else [int(v) for v in sys.argv[1].split(",")]
    only = int(sys.argv[2]) if len(sys.argv) > 2 else None
    if os.environ.get("CCDM_PC_TIMELINE"):
        for i in [int(v) for v in os.environ.get("TIMELINE", "0").split(",")]:
            print(SHAPES[i]); timeline_pc(precs[0], SHAPES[i])
        sys.exit(0)
    if os.environ.get("PMCRUN"):          # one (shape, ablation bits) pair, a few launches: the unit tools/pmc_traffic.sh counts
        i, d = [int(v) for v in os.environ["PMCRUN"].split(",")]
        print(SHAPES[i], "dbg", d, run(precs[0], SHAPES[i], iters=5, dbg=d))
        sys.exit(0)
    if os.environ.get("TIMELINE"):
        for i in [int(v) for v in os.environ["TIMELINE"].split(",")]:
            print(SHAPES[i]); timeline(precs[0], SHAPES[i])
        sys.exit(0)
    for prec in precs:
        total = 0.0
        print(f"--- prec={prec} N={N}")
        for i, sh in enumerate(SHAPES if only is None else SHAPES[:only]):
            ms, tf, gbs = run(prec, sh)
            total += sh[0] * ms
            abl = ""
            if os.environ.get("ABLATE"):
                # 1: no MFMA, 2: no commit (VALU+LDS writes), 4: no prefetch loads, 8: no stores
                abl = "  | " + " ".join(f"{nm}={run(prec, sh, dbg=d)[0]*1e3:6.1f}" for nm, d in
                                        [("-Bstage", 32), ("-mfma", 1), ("-commit", 2), ("-loads", 4), ("-stores", 8), ("ld+st only", 3), ("st only", 7), ("ld only", 11), ("nothing", 15), ("no-barriers(wrong)", 256), ("2blk/CU", 512), ("1blk/CU", 1024), ("2blk ld+st", 512 | 3), ("1blk ld+st", 1024 | 3)])
            print(f"{sh[0]:2d}x {sh[1]+sh[2]:3d}->{sh[3]:3d} @{sh[4]:3d}x{sh[5]:3d} k{sh[6]} s{sh[7]} up{sh[8]} gn{sh[9]}: {ms*1e3:8.1f} us  {tf:7.1f} TF/s  {gbs:7.0f} GB/s(io){abl}")
        print(f"weighted conv total per denoise step: {total:.3f} ms")
        if only is None:
            for cnt, c0, c1, cout, H, W in [(3, 32, 32, 32, 128, 128), (3, 32, 32, 32, 64, 64), (3, 64, 64, 64, 32, 32), (3, 96, 96, 96, 16, 16), (3, 128, 128, 128, 8, 8)]:
                print(f"{cnt:2d}x {cout:3d}->{cout:3d} +skip1x1 {c0 + c1:3d} @{H:3d}x{W:3d}: {run_fused_skip(prec, c0, c1, cout, H, W) * 1e3:8.1f} us")

#!/usr/bin/env python3
"""Round 6: for every launch of ONE denoise step (C2) — blocks, resident block slots (256 CUs x blocks per CU), rounds, launch time —
and, where a block timeline exists (tools/timeline_op.py, ablation build), the chain of one block in cycles, to show which launches
are "rounds x one block's chain" and which are throughput-bound.

    python tools/lowres_rounds.py <rocprofv3 kernel trace csv> <bench --per-op json> [<timelines txt>] > profiles/r06_lowres_rounds.md

Blocks per CU = min over: VGPRs (512 per SIMD lane; the trace reports the allocation / 2), LDS (160 KB per CU; dynamic LDS is not in the
trace: recomputed here from the kernels' own formulas, cited below), 32 waves per CU.
LDS formulas: k_conv: HP x (4 CK + 16) + NTAP x (CK / 16) x NI x 2 KB (ccdm_conv.hip launch_conv) | k_conv_ks: (7 s + 3)^2 x (64 ceil(C / 16) + 16)
[+ 64 x (64 ceil(SC / 16) + 16)] + NI x 4 KB + 8 C (ccdm_conv_ks.hip conv_ks_plan) | k_upconv: 10 x (TW + 2) x (4 C + 16) [+ 18 KB unless aliased]
| k_qkv_attention: QkvAttnGeo::LDS | k_stem / k_conv1x1: static (in the trace) | k_head: 32 x 8 + 11 x 32 x 144."""
import csv
import json
import math
import re
import sys


def ints(name):
    m = re.search(r"<([^>]*)>", name)
    out = []
    for t in (m.group(1).split(",") if m else []):
        t = t.strip()
        out.append(1 if t == "true" else 0 if t == "false" else int(t))
    return out


def lds_bytes(name, static_lds, op, prev_op):
    t = ints(name)
    if "k_conv_ks<" in name:
        stride, _, _, nits, ni = t[:5]
        c = op["cin"]
        sc = prev_op["cin"] if nits else 0                       # the fused 1x1 skip reads the block's (concatenated) input
        hp = (7 * stride + 3) ** 2
        lds = hp * (64 * math.ceil(c / 16) + 16) + (64 * (64 * math.ceil(sc / 16) + 16) if sc else 0)
        lds = max(lds, 8 * 32 * 16 * 4)
        return lds + ni * 8 * 32 * 16 + 8 * c
    if "k_conv<" in name:
        prec, ck, ks, stride, th, tw, waves, mi, ni, ksp, up2, skwt = (t + [1, 0, 0])[:12]
        hp = ((th - 1) * stride + ks) * ((tw - 1) * stride + ks)
        ntap = 4 if up2 else ks * ks
        return hp * (4 * ck + 16) + ntap * (ck // 16) * ni * 2048 + 8 * op["cin"]
    if "k_upconv<" in name:
        c, tw, alias = t[:3] if len(t) >= 3 else (t[0], 16, t[1])
        a = 10 * (tw + 2) * (4 * c + 16)
        return max(a, 18432) if alias else a + 18432
    if "k_qkv_attention<" in name:
        T, C = t[:2]
        return C * 8 + 768 + 2 * (C // 32) * 6144 + T * 136 + 32 * (4 * T + 8)
    if "k_head<" in name:
        return 256 + 11 * 32 * 144
    return static_lds


def main():
    trace, per_op = sys.argv[1], json.load(open(sys.argv[2]))
    chains = {}
    if len(sys.argv) > 3:
        for line in open(sys.argv[3]):
            m = re.match(r"op (\d+): total (\d+) cycles", line)
            if m:
                chains[int(m.group(1))] = int(m.group(2))
    rows = [r for r in csv.DictReader(open(trace)) if r["Kernel_Name"].startswith(("ccdm::", "void ccdm::"))]
    stems = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"]]
    nper = stems[1] - stems[0]
    steps = [rows[s:s + nper] for s in stems if s + nper <= len(rows)]
    steps = steps[len(steps) // 2:]                                      # the later denoise steps (warm)
    print("# Every launch of one C2 denoise step: blocks, resident slots, rounds, time (round 6)\n")
    print(f"Source: rocprofv3 kernel trace of `bench.py --graph 0 --substreams 1` ({len(steps)} denoise steps averaged), the `--per-op` table of the same "
          "build, block timelines of `tools/timeline_op.py` (s_memtime cycles of one mid-grid block inside the real step; the clock under the "
          "bench is 1.6–2.0 GHz).  `slots` = 256 CUs × blocks per CU (limited by: v = registers, l = LDS, w = waves); `rounds` = blocks ÷ slots; "
          "`chain` = one block's timeline; `chain µs` = chain ÷ 2.0 GHz — where a launch is one round of blocks, its time is one block's "
          "chain plus the launch boundary (≈ 1.7 µs), not its bytes or FLOPs.\n")
    print("| op | layer | kernel | grid × wg | VGPR | LDS KB | blk/CU | slots | rounds | µs | chain cycles | chain µs @2 GHz |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    tot = {}
    for i in range(nper):
        r = steps[0][i]
        name = r["Kernel_Name"].replace("void ", "")
        short = re.sub(r"\(.*", "", name).replace("ccdm::", "")
        if "k_step_inc" in short:
            continue
        op = per_op[i] if i < len(per_op) else {"name": "?", "shape": ""}
        m = re.match(r"(\d+)->(\d+)", op.get("shape", ""))
        opi = {"cin": int(m.group(1)) if m else 0}
        prev = per_op[i - 1] if i else op
        mp = re.match(r"(\d+)->(\d+)", prev.get("shape", ""))
        previ = {"cin": int(mp.group(1)) if mp else 0}
        wg = int(r["Workgroup_Size_X"])
        gx, gy = int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))
        blocks = gx * gy
        vg = 2 * int(r["VGPR_Count"])                                        # the trace reports allocated registers / 2
        alloc = max(8, math.ceil(vg / 8) * 8)
        wps = min(8, 512 // alloc)
        waves = wg // 64
        by_v = (wps * 4) // waves
        lds = lds_bytes(name, int(r["LDS_Block_Size"]), opi, previ)
        by_l = (160 * 1024) // lds if lds else 99
        by_w = 32 // waves
        bpc = max(1, min(by_v, by_l, by_w, 8))
        lim = "v" if bpc == by_v else ("l" if bpc == by_l else "w")
        slots = 256 * bpc
        us = sum((int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"])) / 1e3 for s in steps) / len(steps)
        ch = chains.get(i)
        stage = op.get("shape", "").split("@")[-1] if "@" in op.get("shape", "") else op.get("shape", "")
        tot[stage] = tot.get(stage, 0.0) + us
        print(f"| {i} | {op['name']} {op.get('shape', '')} | `{short[:46]}` | {gx}×{gy} × {wg} | {vg} | {lds / 1024:.0f} | {bpc}{lim} | {slots} | "
              f"{blocks / slots:.2f} | {us:.1f} | {ch if ch else ''} | {ch / 2000:.1f} |" if ch else
              f"| {i} | {op['name']} {op.get('shape', '')} | `{short[:46]}` | {gx}×{gy} × {wg} | {vg} | {lds / 1024:.0f} | {bpc}{lim} | {slots} | "
              f"{blocks / slots:.2f} | {us:.1f} | | |")
    print("\nKernel-trace time per stage (µs; the HIP-event taps of bench.py add the launch boundary, ≈ 2 µs per launch): "
          + ", ".join(f"{k}: {v:.0f}" for k, v in tot.items()))


if __name__ == "__main__":
    main()

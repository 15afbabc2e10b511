#!/usr/bin/env python3
"""Collapse the rocprofv3 --pmc passes of tools/pmc_bench.sh into per-(kernel, grid) per-launch means.
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE reports half of the bytes of a wide (16 B/lane)
coalesced read stream on gfx950 -> doubled.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the
8 XCDs, so one launch offers GRBM_GUI_ACTIVE / 8 * 1024 SIMD-cycles: mfma_busy = MFMA cycles / that."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
executions = int(sys.argv[3]) if len(sys.argv) > 3 else 16    # denoise steps the profiled command executed (tools/pmc_bench.sh: 2 x STEPS)
agg = defaultdict(lambda: defaultdict(list))
seq = defaultdict(lambda: defaultdict(list))                 # key -> counter -> [(dispatch id, value)]: for the by-position split
dur = defaultdict(list)
for f in glob.glob(src + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "ccdm::" not in name:
            continue
        short = re.sub(r"\(.*", "", name.replace("void ", ""))
        key = f"{short} grid={r['Grid_Size']} wg={r['Workgroup_Size']}"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        seq[key][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",):
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for key, cs in agg.items():
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    e = {"launches": max(len(v) for v in cs.values()), "counters_mean_per_launch": m}
    if key in dur:
        e["mean_us_in_grbm_pass"] = sum(dur[key]) / len(dur[key])
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_read_bytes"] = 2 * m["FETCH_SIZE"] * 1024
        e["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
        e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        e["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        e["valu_active"] = 4 * m["SQ_ACTIVE_INST_VALU"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)       # quad-cycles -> cycles
    if m.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in m:
        # MI355X_MICROARCH.md §LDS: SQ_LDS_BANK_CONFLICT = extra LDS-array cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles
        e["lds_conflict_share_of_lds_cycles"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
        if m.get("GRBM_GUI_ACTIVE", 0) > 0:
            e["lds_busy"] = m["SQ_LDS_IDX_ACTIVE"] / (m["GRBM_GUI_ACTIVE"] / 8 * 256)          # one LDS per CU: 256 of them
            e["lds_conflict_share_of_kernel_time"] = m["SQ_LDS_BANK_CONFLICT"] / (m["GRBM_GUI_ACTIVE"] / 8 * 256)
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
        wc = m["SQ_WAVE_CYCLES"]
        e["wave_time_split"] = {"valu_active": m.get("SQ_ACTIVE_INST_VALU", 0) / wc, "lds_active": m.get("SQ_ACTIVE_INST_LDS", 0) / wc,
                                "wait_any(waitcnt/barrier)": m.get("SQ_WAIT_ANY", 0) / wc, "wait_inst_any(issue stall)": m.get("SQ_WAIT_INST_ANY", 0) / wc}
    res[key] = e
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["launches"] * kv[1].get("mean_us_in_grbm_pass", 0)))
# The dominant (kernel, grid) runs several layer shapes: its launches recur with `period` per denoise step, in engine-op order (bench.py's
# roofline.kernel lists the ops), so launch i of the class belongs to position i % period.  Per-position means of the same counters:
top = next(iter(res))
by_pos = {}
period = res[top]["launches"] // executions if res[top]["launches"] % executions == 0 else 0      # launches of that class per denoise step
if period > 0:
    for cname, rows in seq[top].items():
        rows.sort()
        for i, (_, v, us) in enumerate(rows):
            e = by_pos.setdefault(i % period, {"n": 0, "counters": defaultdict(float), "us": 0.0, "nus": 0})
            e["counters"][cname] += v
            if cname == "GRBM_GUI_ACTIVE":
                e["us"] += us; e["nus"] += 1
    n_per = res[top]["launches"] // period
    out_pos = {}
    for pos, e in sorted(by_pos.items()):
        m = {k: v / n_per for k, v in e["counters"].items()}
        o = {"mean_us_in_grbm_pass": e["us"] / max(e["nus"], 1)}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            o["hbm_read_MB"] = 2 * m["FETCH_SIZE"] * 1024 / 1e6
            o["hbm_write_MB"] = m["WRITE_SIZE"] * 1024 / 1e6
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE", 0) > 0:
            o["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
            o["valu_active"] = 4 * m.get("SQ_ACTIVE_INST_VALU", 0) / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        o["insts_valu"], o["insts_mfma"] = m.get("SQ_INSTS_VALU"), m.get("SQ_INSTS_MFMA")
        out_pos[str(pos)] = o
    res[top]["by_position_in_step"] = out_pos
json.dump({"note": __doc__, "kernels": res}, open(out, "w"), indent=1)
for k, e in list(res.items())[:8]:
    print(k[:90], "launches", e["launches"], "us", round(e.get("mean_us_in_grbm_pass", 0), 1), "hbm MB", round(e.get("hbm_bytes", 0) / 1e6, 1),
          "mfma_busy", round(e.get("mfma_busy", 0), 3), "valu_active", round(e.get("valu_active", 0), 3))

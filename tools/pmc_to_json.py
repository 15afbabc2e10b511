#!/usr/bin/env python3
"""Collapse the rocprofv3 --pmc passes of tools/pmc_conv.sh into one JSON for the dominant conv kernel
(per-launch means).  HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB and FETCH_SIZE
reports half of the bytes of a wide (16 B/lane) coalesced read stream on gfx950 -> doubled."""
import csv
import glob
import json
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
agg = defaultdict(list)
kernel = None
for f in glob.glob(src + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_conv" in r["Kernel_Name"]:
            kernel = r["Kernel_Name"]
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
res = {"kernel": kernel, "launches": {k: len(v) for k, v in agg.items()}, "counters_mean_per_launch": m}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    res["hbm_read_bytes"] = 2 * m["FETCH_SIZE"] * 1024
    res["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
    res["hbm_bytes"] = res["hbm_read_bytes"] + res["hbm_write_bytes"]
    res["note"] = "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); units KiB; conv3x3 32->32 @128x128 N=64 standalone (tools/bench_conv.py shape 0)"
if "SQ_WAVE_CYCLES" in m:
    wc = m["SQ_WAVE_CYCLES"]
    res["wave_time_split"] = {"valu_active": m.get("SQ_ACTIVE_INST_VALU", 0) / wc, "lds_active": m.get("SQ_ACTIVE_INST_LDS", 0) / wc,
                              "wait_any(waitcnt/barrier)": m.get("SQ_WAIT_ANY", 0) / wc, "wait_inst_any(issue stall)": m.get("SQ_WAIT_INST_ANY", 0) / wc}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:600])

#!/usr/bin/env python3
"""Per-stage sums of a bench.py --per-op table:  python tools/stage_table.py gpurun_out/per_op_x.json ..."""
import json
import sys

for f in sys.argv[1:]:
    p = json.load(open(f))
    print(f, "total", round(sum(o["mean_us"] for o in p)), "us per denoise step,", len(p), "ops")
    st = {}
    for o in p:
        key = o["shape"].split("@")[1] if o["kind"] == "conv" else o["kind"] + " " + str(o.get("shape") or "")
        st.setdefault(key, [0.0, 0])
        st[key][0] += o["mean_us"]
        st[key][1] += 1
    for k, (v, n) in sorted(st.items(), key=lambda kv: -kv[1][0]):
        print("    %-44s %7.0f us  %3d ops" % (k, v, n))

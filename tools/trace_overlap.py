#!/usr/bin/env python3
"""How do the two concurrent sub-batch streams of the product default share the GPU?  Reads a rocprofv3 --kernel-trace CSV of a
bench.py run (product default: HIP-graph replay, two streams) and reports, for a window of denoise steps in the middle of the run:
wall time per step, the summed kernel durations per queue, the fraction of the window in which 0 / 1 / 2 kernels are in flight, and
per kernel class (name x grid) its mean duration here — to be set against the single-stream durations of the same kernels.
    python tools/trace_overlap.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys
from collections import defaultdict

path = sys.argv[1]
ev = []
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "ccdm::" not in name:
            continue
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        grid = f'{int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(wg, 1)}x{r.get("Grid_Size_Y", 1)}'
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), name, grid))
ev.sort()
queues = sorted(set(e[2] for e in ev))
# the timed product passes come last in bench.py only if no untimed tapped pass follows: take the longest stretch in which >= 2 queues alternate
by_q = defaultdict(list)
for e in ev:
    by_q[e[2]].append(e)
busy_q = sorted(queues, key=lambda q: -len(by_q[q]))[:2]
sel = [e for e in ev if e[2] in busy_q]
# window: find step boundaries of queue A (k_step_inc) and take steps 60..180 of the LAST run in which both queues are active
incs = [e for e in by_q[busy_q[0]] if "k_step_inc" in e[3]]
other = by_q[busy_q[1]]
o0, o1 = other[0][0], other[-1][1]
incs = [e for e in incs if o0 <= e[0] <= o1]
res = dict(queues=len(queues), busy_queues=busy_q, steps_seen=len(incs))
if len(incs) > 200:
    lo, hi = incs[-190][1], incs[-70][1]
    nsteps = 120
    win = [e for e in sel if e[0] >= lo and e[1] <= hi]
    res["window_steps"] = nsteps
    res["wall_us_per_step_pair"] = (hi - lo) / 1e3 / nsteps
    for q in busy_q:
        res[f"kernel_us_per_step_queue_{q}"] = sum(e[1] - e[0] for e in win if e[2] == q) / 1e3 / nsteps
    # concurrency histogram
    pts = []
    for e in win:
        pts.append((e[0], 1)); pts.append((e[1], -1))
    pts.sort()
    level, last, hist = 0, lo, defaultdict(int)
    for t, d in pts:
        hist[level] += t - last
        last = t
        level += d
    hist[level] += hi - last
    tot = sum(hist.values())
    res["in_flight_fraction"] = {str(k): round(v / tot, 4) for k, v in sorted(hist.items())}
    cls = defaultdict(lambda: [0, 0])
    for e in win:
        c = cls[(e[3].split("(")[0][:90], e[4])]
        c[0] += 1; c[1] += e[1] - e[0]
    res["classes"] = [dict(kernel=k[0], grid=k[1], launches_per_step_pair=round(v[0] / nsteps, 2), mean_us=round(v[1] / v[0] / 1e3, 2),
                           us_per_step_pair=round(v[1] / 1e3 / nsteps, 1)) for k, v in sorted(cls.items(), key=lambda kv: -kv[1][1])[:40]]
out = json.dumps(res, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
print(out[:6000])

#!/usr/bin/env python3
"""Dispatch gaps of a rocprofv3 --kernel-trace CSV: time between the end of launch i and the start of launch i+1 on the GPU
timeline, overall and by the pair's kernel classes.  (What a kernel boundary costs inside the real denoise step, as opposed to
the guide's 1.45 us between trivial kernels.)
    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [skip_first_n]"""
import csv
import glob
import statistics as st
import sys


def short(name: str) -> str:
    name = name.replace("ccdm::", "")
    return name[:70]


def main():
    d = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[skip:]
    gaps, durs = [], []
    by_pair = {}
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g > 200000:        # host-side pause (pass boundary), not a dispatch gap
            continue
        gaps.append(g)
        durs.append(e0 - s0)
        by_pair.setdefault((short(n0), short(n1)), []).append(g)
    print(f"{len(rows)} launches, {len(gaps)} gaps: mean {st.mean(gaps):.0f} ns, median {st.median(gaps):.0f}, p10 {sorted(gaps)[len(gaps)//10]}, "
          f"p90 {sorted(gaps)[9*len(gaps)//10]}; sum of gaps {sum(gaps)/1e3:.0f} us, sum of durations {sum(durs)/1e3:.0f} us "
          f"(gap share {sum(gaps)/(sum(gaps)+sum(durs)):.3f})")
    print("negative gaps (overlap):", sum(1 for g in gaps if g < 0))
    for (a, b), v in sorted(by_pair.items(), key=lambda kv: -sum(kv[1]))[:25]:
        print(f"{len(v):6d} x {st.mean(v):7.0f} ns   {a}  ->  {b}")


if __name__ == "__main__":
    main()

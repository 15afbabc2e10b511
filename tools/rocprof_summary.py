#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (rocpd sqlite .db or *_kernel_trace.csv) into a small text table:
per kernel name x grid: launches, total ms, share, mean/min/max us.  Usage: rocprof_summary.py <db|csv> [out.md]"""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    q = ("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, count(*), sum(end-start), min(end-start), max(end-start) "
         "from kernels group by name, grid_x, grid_y order by 8 desc")
    return [dict(name=r[0], grid=f"{r[1] // max(r[3], 1)}x{r[2]}", wg=r[3], lds=r[4], vgpr=r[5], n=r[6], tot=r[7], mn=r[8], mx=r[9])
            for r in cur.execute(q)]


def rows_from_csv(path):
    agg = defaultdict(lambda: dict(n=0, tot=0, mn=1 << 62, mx=0, all=[]))
    with open(path) as f:
        for r in csv.DictReader(f):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
            key = (r["Kernel_Name"], f'{int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(wg, 1)}x{r.get("Grid_Size_Y", 1)}', wg,
                   r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
            a = agg[key]
            a["n"] += 1; a["tot"] += d; a["mn"] = min(a["mn"], d); a["mx"] = max(a["mx"], d); a["all"].append(d)
    out = [dict(name=k[0], grid=k[1], wg=k[2], lds=k[3], vgpr=k[4], **v) for k, v in agg.items()]
    return sorted(out, key=lambda r: -r["tot"])


def per_op_table(path):
    """Per-op means for the sampler: the launches between two k_step_inc dispatches are the ops of one denoise
    step, always in the same order, so position j within a step identifies engine op j."""
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]))
    ev.sort()
    steps, cur = [], []
    for _, d, name in ev:
        if "k_step_inc" in name:
            steps.append(cur); cur = []
        elif "ccdm::" in name and "k_step_set" not in name:
            cur.append((name, d))
    if not steps:
        return []
    L = max(set(len(s) for s in steps), key=[len(s) for s in steps].count)
    steps = [s for s in steps if len(s) == L]
    out = []
    for j in range(L):
        ds = [s[j][1] for s in steps]
        out.append((j, steps[0][j][0], len(ds), sum(ds) / len(ds) / 1e3, min(ds) / 1e3, max(ds) / 1e3))
    return out


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    total = sum(r["tot"] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace summary of {path.split('/')[-1]}", "",
             f"total kernel time {total / 1e6:.2f} ms over {sum(r['n'] for r in rows)} launches", "",
             "`mean w/o max` drops each row's single longest launch (the first launch of a kernel pays its code load: up to 20 ms) — the figure to "
             "compare with bench.py's HIP-event taps.", "",
             "| kernel | grid (blocks) | wg | LDS B | VGPR | launches | total ms | % | mean us | mean w/o max | median us | min us | max us |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows[:40]:
        lines.append(f"| `{r['name'][:90]}` | {r['grid']} | {r['wg']} | {r['lds']} | {r['vgpr']} | {r['n']} | {r['tot'] / 1e6:.2f} | "
                     f"{100 * r['tot'] / total:.1f} | {r['tot'] / r['n'] / 1e3:.1f} | "
                     f"{((r['tot'] - r['mx']) / max(r['n'] - 1, 1) / 1e3):.1f} | "
                     f"{(sorted(r['all'])[len(r['all']) // 2] / 1e3 if r.get('all') else float('nan')):.1f} | {r['mn'] / 1e3:.1f} | {r['mx'] / 1e3:.1f} |")
    if not path.endswith(".db"):
        ops = per_op_table(path)
        if ops:
            lines += ["", "## per engine op (position within a denoise step; op 1 = input_blocks.1.0.in_layers.2, the kernel bench.py taps)", "",
                      "| op | kernel | steps | mean us | min us | max us |", "|---|---|---|---|---|---|"]
            for j, name, n, mean, mn, mx in ops:
                lines.append(f"| {j} | `{name[:70]}` | {n} | {mean:.1f} | {mn:.1f} | {mx:.1f} |")
            lines += ["", f"sum of per-op means = {sum(o[3] for o in ops) / 1e3:.3f} ms per denoise step"]
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()

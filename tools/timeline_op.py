#!/usr/bin/env python3
"""Phase timeline (s_memtime stamps) of one mid-grid block of ONE conv op inside the real denoise step — cold weights, the
producer's output fresh in cache — as opposed to tools/bench_conv.py's back-to-back launches of one layer.
    CCDM_LIB=tools/ab/abl.so CCDM_TIMELINE_OP=33 python tools/timeline_op.py       (library built with CCDM_ABLATION=1)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccdm_stochastic_segmentation_amd import hip
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

NAMES = {12: "kernel-entry", 1: "start", 2: "top", 3: "barrierA", 4: "commit", 5: "barrierB", 6: "issue", 7: "mfma", 8: "end",
         9: "epi-barrier", 10: "epi-transpose", 11: "epi-rows", 13: "first-issue", 14: "gn-affine"}

if __name__ == "__main__":
    sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "0", "--denoise-steps", "4", "--no-cpu-baseline", "--no-secondary"] + sys.argv[1:]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        bench.main()
    torch.cuda.synchronize()
    lib = hip.load()
    buf = (C.c_ulonglong * 1024)()
    hip.check(lib.ccdm_debug_read_timeline(buf, 1024))
    n = int(buf[1023])
    ev = [(int(buf[i]) >> 56, int(buf[i]) & ((1 << 56) - 1)) for i in range(min(n, 1020))]
    prev = ev[0][1]
    out = []
    for slot, t in ev[1:]:
        out.append(f"{NAMES.get(slot, slot)}+{t - prev}")
        prev = t
    if os.environ.get("CCDM_TIMELINE_KS"):      # ccdm_conv_ks.hip's stamps
        NAMES.clear()
        NAMES.update({1: "entry", 2: "halo-issued", 3: "resid-issued", 4: "gn-table", 5: "commit", 6: "barrier", 7: "mfma", 8: "barrier", 9: "partials+barrier",
                      10: "reduce+store", 11: "stats", 20: "gn-requested", 22: "B0-issued", 23: "epi-consts-issued"})
        NAMES.update({30 + i: f"item{i}" for i in range(24)})
        out = []
        prev = ev[0][1]
        for slot, t in ev[1:]:
            out.append(f"{NAMES.get(slot, slot)}+{t - prev}")
            prev = t
    print(f"op {os.environ.get('CCDM_TIMELINE_OP')}: total {prev - ev[0][1]} cycles: " + " ".join(out[:80]))

#!/usr/bin/env python3
"""Same-box A/B of single conv layers (tools/bench_conv.py shapes): CCDM_LIB selects the library; DBG = diagnostic bits of prec >> 8
(4096: CCDM_DIAG_STAGED_COMMIT).   python tools/ab_conv.py 0,1,6,10"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_conv as B
from ccdm_stochastic_segmentation_amd import hip

if __name__ == "__main__":
    idx = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,6,10").split(",")]
    dbg = int(os.environ.get("DBG", "0"))
    prec = int(os.environ.get("PREC", str(hip.PREC_F16X3)))          # 2: the opt-in single-pass mode (what the matrix work of a layer is worth)
    out = []
    for i in idx:
        sh = B.SHAPES[i]
        ms = min(B.run(prec, sh, dbg=dbg)[0] for _ in range(3))
        out.append(f"{sh[1]+sh[2]}->{sh[3]}@{sh[4]} k{sh[6]}: {ms*1e3:.1f}")
    print(os.environ.get("CCDM_LIB", "tree"), "prec", prec, "dbg", dbg, " | ".join(out))

#!/usr/bin/env python3
"""Run-to-run identity soak of the wave-per-phase Upsample kernel (and the general kernel's form of the same call): every geometry REPS
times on fresh outputs, all results bit-identical to the first.   python tools/soak_upconv.py [REPS]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ccdm_stochastic_segmentation_amd import hip
from tests import hip_util as U

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(0)
    bad = 0
    for cin, cout, H, W, N in [(64, 64, 32, 32, 64), (96, 96, 16, 16, 64), (128, 128, 8, 8, 64), (32, 32, 64, 64, 16), (64, 64, 128, 256, 2), (128, 128, 16, 32, 4)]:
        x = torch.from_numpy((rng.standard_normal((N, H, W, cin)) * 1.5 + 0.3).astype(np.float32)).cuda()
        w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        first = None
        for r in range(reps):
            out, st = U.conv2d([x], w, b, 3, up=2, prec=hip.PREC_F16X3)
            if first is None:
                first = (out.clone(), st.clone())
                gen, _ = U.conv2d([x], w, b, 3, up=2, prec=hip.PREC_F16X3, diag=hip.DIAG_GENERAL_KERNEL)
                assert torch.equal(gen, out), "differs from the general kernel"
            elif not (torch.equal(out, first[0]) and torch.equal(st, first[1])):
                bad += 1
                print("MISMATCH", (cin, cout, H, W, N), "rep", r, int((out != first[0]).sum()), int((st != first[1]).sum()))
        print((cin, cout, H, W, N), "ok" if not bad else "BAD")
    print("soak:", "clean" if not bad else f"{bad} mismatching repetitions")
    sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Golden G10: LIDC metrics computed by the REFERENCE's own functions
(/root/reference/evaluation/evaluate_lidc_uncertainty.py:27-73) on seeded random class maps.
The module drags in ignite / wandb / torchvision / ddpm.trainer at import; those are stubbed (they are not used by the
metric helpers), and np.bool (removed in numpy >= 1.24) is aliased — only in this build-container script."""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import scipy.optimize  # noqa: F401  (import before aliasing np.bool)
import torch  # noqa: F401

np.bool = np.bool_
for name in ["ignite", "ignite.distributed", "ignite.engine", "ignite.handlers", "ignite.metrics", "ignite.utils", "wandb", "torchvision",
             "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils", "PIL", "PIL.Image", "h5py", "imageio",
             "sklearn", "sklearn.model_selection", "dino", "timm", "ddpm.trainer", "ddpm.polyak", "ddpm.utils"]:
    sys.modules.setdefault(name, MagicMock())
pkg = types.ModuleType("ddpm"); pkg.__path__ = ["/root/reference/ddpm"]; sys.modules["ddpm"] = pkg
sys.path.insert(0, "/root/reference")
m = importlib.import_module("evaluation.evaluate_lidc_uncertainty")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for tag, K, B, S, L, hw, seed in [("k2", 2, 3, 16, 4, (32, 32), 10), ("k2_empty", 2, 2, 4, 4, (16, 16), 11), ("k5", 5, 2, 6, 3, (24, 20), 12)]:
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, K, (B, L) + hw)
    smp = rng.integers(0, K, (B, S) + hw)
    if tag == "k2_empty":                       # all-background pairs: union == 0 -> IoU := 1
        lab[0] = 0; smp[0, :2] = 0; smp[1] = 0
    ged, div_e, div_s = m.calc_batched_generalised_energy_distance(lab, smp, K)
    lcm = np.lcm(S, L)
    hm = m.batched_hungarian_matching(np.repeat(lab, lcm // L, 1), np.repeat(smp, lcm // S, 1), K)
    out.update({f"{tag}_seed": seed, f"{tag}_ged": ged, f"{tag}_div_experts": div_e, f"{tag}_div_samples": div_s, f"{tag}_hm_iou": np.array(hm),
                f"{tag}_shape": np.array([K, B, S, L, *hw])})
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g10_lidc_metrics.npz"), **out)
print("g10_lidc_metrics.npz written")

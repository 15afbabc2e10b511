#!/usr/bin/env python3
"""Golden G14: the LIDC evaluation harness numbers computed by the REFERENCE's own `Tester.test_step`
(/root/reference/evaluation/evaluate_lidc_uncertainty.py:75-136) on fixed, seeded predictions.

The reference module needs ignite / wandb / torchvision at import: they are stubbed (only `idist.device()` is used by
test_step; it returns the CPU), np.bool is aliased — only in this build-container script.  The model is a stand-in that
returns seeded predictions (tests/golden_util.py: harness_case), so the fixture pins everything AROUND the sampler:
repeat_interleave order, the [B,S,...] reshape, GED / diversity / Hungarian IoU accumulation, the log-mean vote (incl. log(0)
for one-hot "majority" predictions) and the rows that reach ignite's confusion matrix.  IoU / mIoU / Dice themselves are
ignite.metrics formulas on that matrix (cm.diag() / (cm.sum(1) + cm.sum(0) - cm.diag() + 1e-15) etc.); ignite is not
installed here, so the fixture stores the confusion matrix built from the reference's own (y, y_pred) outputs."""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import scipy.optimize  # noqa: F401  (import before aliasing np.bool)
import torch

np.bool = np.bool_
idist = types.ModuleType("ignite.distributed")
idist.device = lambda: torch.device("cpu")
ign = MagicMock()
ign.distributed = idist
sys.modules["ignite"] = ign
sys.modules["ignite.distributed"] = idist
for name in ["ignite.engine", "ignite.handlers", "ignite.metrics", "ignite.utils", "wandb", "torchvision",
             "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils", "PIL", "PIL.Image", "h5py", "imageio",
             "sklearn", "sklearn.model_selection", "dino", "timm", "ddpm.trainer", "ddpm.polyak", "ddpm.utils"]:
    sys.modules.setdefault(name, MagicMock())
pkg = types.ModuleType("ddpm"); pkg.__path__ = ["/root/reference/ddpm"]; sys.modules["ddpm"] = pkg
sys.path.insert(0, "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("evaluation.evaluate_lidc_uncertainty")
from tests.golden_util import harness_case  # noqa: E402

out = {}
for vote in ("confidence", "majority"):
    batches, evaluations, K, predict = harness_case(vote)
    S = max(evaluations)

    class FakeModel:
        calls = 0

        def eval(self):
            return self

        def __call__(self, x, image):
            assert x.shape[0] == image.shape[0] and x.shape[1] == K
            p = predict(FakeModel.calls, x.shape[0])
            FakeModel.calls += 1
            return {"diffusion_out": p}

    polyak = types.SimpleNamespace(average_model=FakeModel())
    z = lambda: np.zeros(len(evaluations))
    tester = m.Tester(polyak, evaluations, K, z(), z(), z(), z(), 0, 0)
    conf = np.zeros((K, K), dtype=np.int64)
    for b in batches:
        r = tester.test_step(None, b)
        y, yp = r["y"].reshape(-1).numpy(), r["y_pred"].argmax(dim=1).reshape(-1).numpy()     # ignite ConfusionMatrix: argmax over dim 1
        np.add.at(conf, (y, yp), 1)
    n_img = sum(b[0].shape[0] for b in batches)
    out.update({f"{vote}_geds": tester.geds / n_img, f"{vote}_div_samples": tester.similarity_samples / n_img,
                f"{vote}_div_experts": np.array(tester.similarity_experts[0] / n_img), f"{vote}_hm_ious": tester.hm_ious / n_img,
                f"{vote}_nonzero": np.array(int(tester.nonzero)), f"{vote}_conf": conf, f"{vote}_n_img": np.array(n_img)})
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g14_lidc_harness.npz"), **out)
print("g14_lidc_harness.npz written", {k: (v.tolist() if v.size < 8 else v.shape) for k, v in out.items()})

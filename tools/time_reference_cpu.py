#!/usr/bin/env python3
"""BASELINE.md §3 step 1 (build container only): time the REAL reference and the oracle on the same C1 input
(LIDC cfg, N=4, 8 denoise steps, 8 CPU threads), check outputs within 1e-5 and step time within ±10 %, and record it.
The oracle is then a faithful stand-in for the reference's CPU path on the GPU box (bench.py cpu_baseline, kind "port")."""
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = types.ModuleType("ddpm"); pkg.__path__ = ["/root/reference/ddpm"]; sys.modules["ddpm"] = pkg
from ddpm.models import build_model  # noqa: E402
from ddpm.models.one_hot_categorical import OneHotCategoricalBCHW  # noqa: E402
from ccdm_stochastic_segmentation_amd.unet_spec import make_unet_spec, make_synthetic_state_dict  # noqa: E402
from oracle import ccdm_oracle as O  # noqa: E402

torch.set_num_threads(8)
BP = dict(base_channels=32, channel_mult=None, attention_resolutions=[32, 16, 8], num_heads=1, num_head_channels=32, softmax_output=True)
m = build_model(250, "cosine", {"s": 0.008}, [(1, 128, 128), (2, 128, 128)], (1, 128, 128), "unet_openai", BP, "datasets.lidc", "confidence", None)
spec = make_unet_spec(image_size=128, in_channels=3, out_channels=2, **BP)
sd = {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(spec, 0).items()}
m.unet.load_state_dict(sd, strict=True)
m.eval()
image = torch.from_numpy(np.random.default_rng(1234).uniform(-1, 1, (4, 1, 128, 128)).astype(np.float32))
sched = O.make_schedule("cosine", 250, {"s": 0.008})
cfg = dict(num_heads=1, num_head_channels=32)


def run_ref():
    torch.manual_seed(42)
    x = OneHotCategoricalBCHW(logits=torch.zeros(4, 2, 128, 128)).sample()
    with torch.no_grad():
        return m(x, image, t=torch.as_tensor(8))["diffusion_out"]


def run_oracle():
    torch.manual_seed(42)
    idx, _ = O.draw_x_T(4, 2, 128, 128)
    with torch.no_grad():
        return O.forward_denoising(sd, cfg, sched, O.one_hot_bchw(idx, 2), image, None, 8, "confidence")["diffusion_out"]


res = {}
for name, fn in (("reference", run_ref), ("oracle", run_oracle)):
    fn()                                        # warm-up
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t0)
    res[name] = {"ms_per_denoise_step_n4": min(ts) / 8 * 1e3, "samples_per_s_T250": 4 / (min(ts) / 8 * 250)}
    res[name + "_out"] = out
diff = (res.pop("reference_out") - res.pop("oracle_out")).abs().max().item()
ratio = res["oracle"]["ms_per_denoise_step_n4"] / res["reference"]["ms_per_denoise_step_n4"]
res.update({"max_abs_output_diff": diff, "oracle_over_reference_step_time": ratio, "threads": torch.get_num_threads(),
            "host": "build container (8 CPUs), torch " + torch.__version__, "workload": "C1: LIDC cfg, N=4, 8 denoise steps (t=8), seed 42"})
assert diff <= 1e-5, diff
json.dump(res, open(os.path.join(ROOT, "profiles", "r01_reference_vs_oracle_cpu.json"), "w"), indent=1)
print(json.dumps(res))

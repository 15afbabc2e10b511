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
REPS = int(os.environ.get("REPS", "6"))
ts = {"reference": [], "oracle": []}
outs = {}
for name, fn in (("reference", run_ref), ("oracle", run_oracle)):
    outs[name] = fn()                           # warm-up
for _ in range(REPS):                           # interleaved: the container's load drifts on the scale of seconds
    for name, fn in (("reference", run_ref), ("oracle", run_oracle)):
        t0 = time.perf_counter(); outs[name] = fn(); ts[name].append(time.perf_counter() - t0)
for name in ts:
    med = float(np.median(ts[name]))
    res[name] = {"ms_per_denoise_step_n4": med / 8 * 1e3, "min_ms_per_denoise_step_n4": min(ts[name]) / 8 * 1e3,
                 "samples_per_s_T250": 4 / (med / 8 * 250), "runs": len(ts[name])}
diff = (outs["reference"] - outs["oracle"]).abs().max().item()
ratios = sorted(o / r for o, r in zip(ts["oracle"], ts["reference"]))
ratio = float(np.median(ratios))
res.update({"max_abs_output_diff": diff, "oracle_over_reference_step_time": ratio, "ratio_min_max": [ratios[0], ratios[-1]],
            "threads": torch.get_num_threads(), "host": "build container (8 CPUs), torch " + torch.__version__,
            "workload": "C1: LIDC cfg, N=4, 8 denoise steps (t=8), seed 42; %d interleaved runs each, medians" % REPS})
assert diff <= 1e-5, diff
json.dump(res, open(os.path.join(ROOT, "profiles", os.environ.get("OUT", "r03_reference_vs_oracle_cpu.json")), "w"), indent=1)
print(json.dumps(res))

#!/usr/bin/env python3
"""Is the sampling loop host-bound?  Times how long the host needs to ENQUEUE one sampling pass (all T denoise steps of every
sub-batch; the call returns before the GPU is done) against the pass itself, for the product default (HIP-graph replay, two sub-batch
streams), one stream, and eager launches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ccdm_stochastic_segmentation_amd import build_model, make_synthetic_state_dict, OneHotCategoricalBCHW

dev = torch.device("cuda:0")
bp = dict(base_channels=32, channel_mult=None, attention_resolutions=[32, 16, 8], num_heads=1, num_head_channels=32, softmax_output=True)
model = build_model(250, "cosine", {"s": 0.008}, [(1, 128, 128), (2, 128, 128)], (1, 128, 128), "unet_openai", bp, "datasets.lidc", "confidence", None)
model.unet.load_state_dict({k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(model.unet.spec, 0).items()}, strict=True)
model = model.to(dev).eval()
N = int(os.environ.get("N", "64"))
img = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (N, 1, 128, 128)).astype(np.float32)).to(dev)
x = OneHotCategoricalBCHW(logits=torch.zeros(N, 2, 128, 128)).sample().to(dev)
from ccdm_stochastic_segmentation_amd.engine import SamplerEngine
_run = SamplerEngine.run
acc = {"t": 0.0, "n": 0, "first": []}
def timed_run(self, n_steps, *a, **k):
    t = time.perf_counter()
    r = _run(self, n_steps, *a, **k)
    dt = time.perf_counter() - t
    acc["t"] += dt; acc["n"] += 1
    if len(acc["first"]) < 60: acc["first"].append(dt)
    return r
SamplerEngine.run = timed_run
for graph, sub in [(True, 0), (True, 1), (False, 0), (False, 1)]:
    model.use_graph, model.substreams = graph, sub
    for rep in range(3):
        torch.cuda.synchronize()
        acc.update(t=0.0, n=0, first=[])
        t0 = time.perf_counter()
        out = model(x, img)["diffusion_out"]       # returns when the last step is enqueued (the output tensor is produced on the stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"graph={int(graph)} substreams={sub or 'auto'}: host returns after {1e3 * (t1 - t0):7.1f} ms, pass done after {1e3 * (t2 - t0):7.1f} ms "
          f"({1e3 * (t2 - t0) / 250:.3f} ms per denoise step); {acc['n']} run() calls took {1e3 * acc['t']:.1f} ms on the host, the first 40: "
          f"{1e6 * sum(acc['first'][:40]) / max(len(acc['first'][:40]), 1):.0f} us each")

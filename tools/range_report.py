#!/usr/bin/env python3
"""Per-layer F16X3 headroom of a state_dict: what every conv of the sampler stages (after GroupNorm + SiLU where it normalises on load,
raw otherwise, incl. the fused 1x1 skip inputs) against the fp16 split's limit (include/ccdm_hip.h: CCDM_F16X3_LIMIT = 4094).

    python tools/range_report.py [--config c2|c4|c4b64|c5shard] [--checkpoint FILE [--key average_model]] [--batch 2] [--out FILE.json]

Runs the network with the EXACT-fp32 kernels (so the values are trustworthy whatever the weights) at a handful of timesteps on
seeded inputs — the checkpoint's weights if one is given, else the synthetic ones of bench.py — and asks the library for the largest
staged |a| per conv (ccdm_engine_input_absmax).  Layers within DenoisingModel.RANGE_MARGIN of the limit are the ones the sampler pins
to fp32 after a range error (on_range_error = "layers"); `--pin` prints the f32_layers set to put in front of a production run.
Needs a GPU (the measurement runs on the HIP kernels themselves)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ccdm_stochastic_segmentation_amd import build_model, make_synthetic_state_dict, hip  # noqa: E402
from ccdm_stochastic_segmentation_amd.evaluation import load_checkpoint  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(bench.CONFIGS))
    ap.add_argument("--checkpoint", default="")
    ap.add_argument("--key", default="average_model")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--out", default="")
    ap.add_argument("--pin", action="store_true")
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    H, W, K, T, C_img = cfg["H"], cfg["W"], cfg["K"], cfg["T"], cfg["C_img"]
    model = build_model(T, "cosine", {"s": 0.008}, [(C_img, H, W), (K, H, W)], (C_img, H, W), "unet_openai", cfg["bp"],
                        "datasets.lidc" if K == 2 else "datasets.cityscapes", "confidence", cfg["fce"])
    if args.checkpoint:
        load_checkpoint(model, args.checkpoint, args.key)
        source = f"{args.checkpoint} [{args.key}]"
    else:
        model.unet.load_state_dict({k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(model.unet.spec, 0).items()}, strict=True)
        source = "synthetic weights (make_synthetic_state_dict seed 0)"
    dev = torch.device("cuda:0")
    model = model.to(dev).eval()
    n = args.batch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from verify_checkpoint import measure_ranges       # the measurement itself (shared with tools/verify_checkpoint.py)
    rows, pinned, t_list = measure_ranges(model, cfg, n, seed=7)
    limit = hip.F16X3_LIMIT * model.RANGE_MARGIN
    res = {"config": args.config, "weights": source, "timesteps": t_list, "batch": n, "limit": hip.F16X3_LIMIT, "pin_threshold": limit,
           "layers_within_margin": pinned, "min_headroom": rows[0]["headroom"] if rows else None, "layers": rows}
    print(f"{len(rows)} conv layers, weights: {source}; smallest headroom x{rows[0]['headroom']:.1f} ({rows[0]['layer']}: max |a| = {rows[0]['max_staged_abs']:.3g})")
    for r in rows[:12]:
        print(f"  {r['layer']:48s} max|a| {r['max_staged_abs']:10.4g}   headroom x{r['headroom']:.1f}")
    if args.pin:
        print("f32_layers =", pinned)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

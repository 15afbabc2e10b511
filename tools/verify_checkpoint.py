#!/usr/bin/env python3
"""Certify a real pretrained checkpoint on the HIP sampler in one command (needs a GPU).

    python tools/verify_checkpoint.py CKPT [--key average_model] [--hdf5 data_lidc.hdf5] [--config c2] [--batch 2] [--steps 10] [--out FILE.json]

What it does, in order (the reference's side of each step in brackets):
 1. strict `load_state_dict` of the checkpoint's U-Net state dict into the build's parameter container
    [ddpm/trainer.py:357-376 writes {"model", "average_model", ...}; evaluation/evaluate_lidc_uncertainty.py:155-162 reads "average_model"];
 2. the per-layer F16X3 headroom table (tools/range_report.py's measurement: what every conv and attention core stages, on the
    exact-fp32 kernels, at six timesteps) and the `f32_layers` pin set the sampler would apply to these weights;
 3. an N-sample strided walk run TWICE on the GPU with the same Philox key — the product default (split-fp16 x3 MFMA, with that pin
    set) and the exact-fp32 engine — printing max |dp| of the final probabilities, the fraction of pixels beyond 1e-3 and the argmax
    mismatches; a third run with nothing pinned reports whether the unpinned fast path overflows on these weights.
Images: the first `--batch` test images of `--hdf5` when given (needs h5py), else seeded synthetic ones of the config's range.
Exit code 0 = loaded, in range (after pinning) and the two precisions agree within the bars printed (1e-4 on probabilities outside
free-running near-tie flips: median below 1e-6 and at most 1.25e-4 of the pixels beyond 1e-3)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ccdm_stochastic_segmentation_amd import build_model, hip  # noqa: E402
from ccdm_stochastic_segmentation_amd.evaluation import load_checkpoint, TestLIDC  # noqa: E402

FREE_RUN_FRAC = 1.25e-4          # tests/test_hip_parity.py: a near-tie flip of argmax p/E perturbs its receptive field


def measure_ranges(model, cfg, n: int, seed: int = 7, image=None, feat=None):
    """largest staged |a| per conv / attention core over six timesteps, on the exact-fp32 kernels (the measurement of tools/range_report.py)"""
    H, W, K, T, C_img = cfg["H"], cfg["W"], cfg["K"], cfg["T"], cfg["C_img"]
    dev = next(model.unet.parameters()).device
    rng = np.random.default_rng(seed)
    if image is None:
        image = torch.from_numpy((rng.uniform(-1, 1, (n, C_img, H, W)) if cfg["image"] == "uniform" else rng.standard_normal((n, C_img, H, W))).astype(np.float32)).to(dev)
    if feat is None and cfg["fce"]:
        feat = torch.from_numpy(rng.standard_normal((n, 384, H // 8, W // 8)).astype(np.float32)).to(dev)
    prec = model.prec
    model.prec = hip.PREC_F32
    model._range_probe = {}
    try:
        t_list = sorted({T, (3 * T) // 4, T // 2, T // 4, 2, 1}, reverse=True)
        for t in t_list:
            x = torch.nn.functional.one_hot(torch.from_numpy(rng.integers(0, K, (n, H, W))), K).permute(0, 3, 1, 2).float().to(dev)
            model(x, image, feat, t=torch.full((n,), float(t)), validation=True)
        probe = model._collect_probe()
    finally:
        model._range_probe = None
        model.prec = prec
    rows = sorted(({"layer": k, "max_staged_abs": v, "headroom": (hip.F16X3_LIMIT / v if v > 0 else float("inf"))} for k, v in probe.items()),
                  key=lambda r: r["headroom"])
    limit = hip.F16X3_LIMIT * model.RANGE_MARGIN
    return rows, [r["layer"] for r in rows if not (r["max_staged_abs"] < limit)], t_list


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkpoint")
    ap.add_argument("--key", default="average_model")
    ap.add_argument("--hdf5", default="")
    ap.add_argument("--config", default="c2", choices=sorted(bench.CONFIGS))
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10, help="strided denoise steps of the comparison walk (the reference's t = 10000 + steps form)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--allow-pickle", action="store_true", help="read a checkpoint whose extra entries need the full unpickler (executes code from the file)")
    ap.add_argument("--out", default="")
    args = ap.parse_args(argv)
    cfg = bench.CONFIGS[args.config]
    H, W, K, T, C_img = cfg["H"], cfg["W"], cfg["K"], cfg["T"], cfg["C_img"]
    assert torch.cuda.is_available(), "verify_checkpoint.py needs a GPU (no CPU path)"
    dev = torch.device("cuda:0")
    model = build_model(T, "cosine", {"s": 0.008}, [(C_img, H, W), (K, H, W)], (C_img, H, W), "unet_openai", cfg["bp"],
                        "datasets.lidc" if K == 2 else "datasets.cityscapes", "confidence", cfg["fce"])
    # 1. strict load
    load_checkpoint(model, args.checkpoint, args.key, allow_pickle=True if args.allow_pickle else None)
    n_tensors = len(model.unet.state_dict())
    print(f"[1] strict load ok: {args.checkpoint} [{args.key}] -> {n_tensors} tensors, {model.unet.spec.num_params()} parameters")
    model = model.to(dev).eval()
    n = args.batch
    rng = np.random.default_rng(args.seed)
    if args.hdf5:
        ds = TestLIDC(args.hdf5, "test", n)
        image = torch.stack([ds[i][0] for i in range(len(ds))]).to(dev)
        n = image.shape[0]
        img_src = f"{args.hdf5}: test images 0..{n - 1}"
    else:
        image = torch.from_numpy((rng.uniform(-1, 1, (n, C_img, H, W)) if cfg["image"] == "uniform" else rng.standard_normal((n, C_img, H, W))).astype(np.float32)).to(dev)
        img_src = "seeded synthetic images"
    feat = torch.from_numpy(rng.standard_normal((n, 384, H // 8, W // 8)).astype(np.float32)).to(dev) if cfg["fce"] else None
    # 2. range table + pin set
    rows, pins, t_list = measure_ranges(model, cfg, n, image=image, feat=feat)
    print(f"[2] F16X3 headroom over {len(rows)} staged operands at t = {t_list} ({img_src}); limit {hip.F16X3_LIMIT:.0f}, pin threshold "
          f"{hip.F16X3_LIMIT * model.RANGE_MARGIN:.0f}")
    for r in rows[:10]:
        print(f"      {r['layer']:48s} max|a| {r['max_staged_abs']:10.4g}   headroom x{r['headroom']:.1f}")
    print(f"    f32_layers the sampler would pin: {pins if pins else 'none'}")
    # 3. the same walk in both precisions, same Philox key
    x = torch.nn.functional.one_hot(torch.from_numpy(rng.integers(0, K, (n, H, W))), K).permute(0, 3, 1, 2).float().to(dev)
    t_arg = torch.as_tensor(10000 + args.steps)
    model.rng, model.philox_seed, model.philox_advance = "philox", args.seed, False

    def walk(prec, pinned, on_error):
        model.prec, model.f32_layers, model.on_range_error = prec, set(pinned), on_error
        model._engines.clear()
        return model(x, image, feat, t=t_arg)["diffusion_out"].float().cpu()

    exact = walk(hip.PREC_F32, (), "raise")
    unpinned_overflow = False
    try:
        fast_unpinned = walk(hip.PREC_F16X3, (), "raise")
    except hip.CcdmRangeError:
        unpinned_overflow, fast_unpinned = True, None
    fast = walk(hip.PREC_F16X3, pins, "raise") if (pins or unpinned_overflow) else fast_unpinned
    err = (fast - exact).abs()
    res = {"checkpoint": args.checkpoint, "key": args.key, "config": args.config, "tensors": n_tensors, "images": img_src, "batch": n,
           "denoise_steps": args.steps, "min_headroom": rows[0]["headroom"] if rows else None, "f32_layers": pins,
           "unpinned_fast_path_overflows": unpinned_overflow, "max_dp": err.max().item(), "median_dp": err.median().item(),
           "frac_gt_1e-3": (err > 1e-3).float().mean().item(), "argmax_mismatch": (fast.argmax(1) != exact.argmax(1)).float().mean().item(),
           "layers": rows}
    ok = res["median_dp"] < 1e-6 and res["frac_gt_1e-3"] <= FREE_RUN_FRAC
    print(f"[3] {args.steps}-step strided walk, N = {n}, Philox seed {args.seed}: F16X3 ({'pinned: ' + str(len(pins)) + ' layers' if pins else 'nothing pinned'}) "
          f"vs exact fp32: max|dp| {res['max_dp']:.3e}, median {res['median_dp']:.3e}, pixels beyond 1e-3: {res['frac_gt_1e-3']:.2e}, "
          f"argmax mismatches {res['argmax_mismatch']:.2e}; unpinned fast path {'OVERFLOWS (pins needed)' if unpinned_overflow else 'stays in range'}")
    print("verdict:", "OK" if ok else "DISAGREEMENT between the precisions beyond the free-running bars")
    res["ok"] = bool(ok)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Headline benchmark: categorical reverse-diffusion sampling throughput (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch: the complete T=250-step sampling of a batch of
64 LIDC-shaped samples per GPU (BASELINE config C2: 128x128, 2 classes, base-32 U-Net, synthetic image and
random-init weights), including the per-step posterior + categorical draw (device Philox RNG) and, for
N > 1, the final RCCL all_gather of the predictions.  Inputs are resident in HBM when timing starts.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PER_GPU_BATCH = 64
T_STEPS = 250
H = W = 128
K = 2
LIDC_BP = dict(base_channels=32, channel_mult=None, attention_resolutions=[32, 16, 8], num_heads=1,
               num_head_channels=32, softmax_output=True)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# SURVEY §8(d), LIDC cfg, fp32, per sample per denoise step
ALGO_MB_PER_SAMPLE_STEP = 163.1
ALGO_WEIGHTS_MB = 4.82


def dominant_kernel_bytes(n: int) -> dict:
    """Algorithmic bytes of ONE launch of the dominant kernel: ResBlock conv3x3 32->32 @128x128 with
    GroupNorm+SiLU on load (8 launches per denoise step, 29.6 % of all FLOPs).
    SURVEY §8(d): conv io 4*(Cin*h*w + Cout*h*w) + the GroupNorm's statistics read 4*C*h*w per sample,
    + weights once per launch."""
    conv_io = 4 * (32 + 32) * H * W * n
    gn_read = 4 * 32 * H * W * n
    weights = 4 * 9 * 32 * 32
    return {"conv_io": conv_io, "gn_read": gn_read, "weights": weights, "total": conv_io + gn_read + weights}


def cpu_baseline(sd, image4, seed=0):
    """Oracle (CPU restatement of the reference, torch-CPU fp32) on a bounded sample of the same workload:
    N=4, the first 8 of 250 denoise steps, extrapolated x250/8."""
    from oracle import ccdm_oracle as O
    # more threads than ~16 makes torch-CPU slower on these small convs (256-core host: 85 s per step)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    sched = O.make_schedule("cosine", T_STEPS, {"s": 0.008})
    torch.manual_seed(seed)
    idx, _ = O.draw_x_T(4, K, H, W)
    x = O.one_hot_bchw(idx, K)
    cfg = dict(num_heads=1, num_head_channels=32)
    t0 = time.perf_counter()
    O.forward_denoising(sd, cfg, sched, x, image4, None, 1, "confidence")            # warm-up: 1 step
    warm = time.perf_counter() - t0
    # bounded sample: ~15 s of CPU work (first step includes one-time warm-up, so this over-estimates the step time a bit)
    n_steps = int(max(2, min(T_STEPS, 15.0 / max(warm, 1e-3))))
    t0 = time.perf_counter()
    O.forward_denoising(sd, cfg, sched, x, image4, None, n_steps, "confidence")
    dt = time.perf_counter() - t0
    ms_step = dt / n_steps * 1e3
    return {"value": 4.0 / (ms_step * 1e-3 * T_STEPS), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (torch-CPU fp32, {torch.get_num_threads()} threads) N=4, {n_steps} of {T_STEPS} denoise steps timed ({dt:.1f} s), extrapolated x{T_STEPS}/{n_steps}",
            "ms_per_denoise_step_n4": ms_step}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=0, help="1: replay each denoise step as a HIP graph (no kernel taps)")
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH)
    ap.add_argument("--substreams", type=int, default=1,
                    help="sample the per-GPU batch as this many contiguous sub-batches on concurrent HIP streams (results are bit-identical)")
    ap.add_argument("--rng", choices=["philox", "torch_cpu"], default="philox",
                    help="philox: noise generated in the epilogue kernel (the benchmark); torch_cpu: the parity mode — Exp(1) noise drawn "
                         "on the host in the reference's order and copied over PCIe (host-RNG bound; reported for DESIGN.md, never the headline)")
    ap.add_argument("--prec", choices=["f32", "f16x3"], default="f16x3",
                    help="conv arithmetic: exact fp32 MFMA, or fp16 hi/lo split x3 MFMA with fp32 accumulate (~2^-22)")
    args = ap.parse_args()

    from ccdm_stochastic_segmentation_amd import build_model, make_synthetic_state_dict
    from ccdm_stochastic_segmentation_amd.distributed import init_from_env, all_gather_ragged
    import torch.distributed as dist

    rank, local_rank, world = init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    n = args.batch

    model = build_model(T_STEPS, "cosine", {"s": 0.008}, [(1, H, W), (K, H, W)], (1, H, W), "unet_openai", LIDC_BP,
                        "datasets.lidc", "confidence", None)
    sd = {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(model.unet.spec, 0).items()}
    model.unet.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    from ccdm_stochastic_segmentation_amd import hip
    model.prec = hip.PREC_F32 if args.prec == "f32" else hip.PREC_F16X3
    model.rng, model.philox_seed, model.use_graph = args.rng, 2024, bool(args.graph)
    model.sample_offset = rank * n                                # Philox counters keyed by global sample index
    model.substreams = args.substreams

    rng = np.random.default_rng(1234)
    image_all = rng.uniform(-1, 1, (max(n, 4), 1, H, W)).astype(np.float32)
    image = torch.from_numpy(image_all[:n]).to(dev)
    x = torch.nn.functional.one_hot(torch.from_numpy(np.random.default_rng(42 + rank).integers(0, K, (n, H, W))), K)
    x = x.permute(0, 3, 1, 2).float().to(dev)

    def one_pass():
        out = model(x, image)["diffusion_out"]
        if world > 1:
            out = all_gather_ragged(out.contiguous(), n * world, world)     # the path's only collective
        return out

    for _ in range(args.warmup):
        one_pass()
    # the executor of sub-batch 0 (the whole batch when substreams == 1) carries the HIP-event taps
    n_tap = n // max(1, min(args.substreams, n))
    eng = model._engine(x[:n_tap], image[:n_tap], None, slot=0)
    dom = next(i for i, nm in enumerate(eng.op_names) if nm == "input_blocks.1.0.in_layers.2")
    taps = not args.graph
    if taps:
        eng.profile_op(dom, capacity=min(T_STEPS, 256))          # HIP events around that launch, on the engine's stream

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern = []
    for _ in range(args.steps):
        out = one_pass()
        if taps:
            torch.cuda.synchronize()
            kern.append(eng.profile_read())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out).all()

    if rank == 0:
        total = n * world * args.steps
        ms_pass = dt / max(args.steps, 1) * 1e3
        ms_dstep = ms_pass / T_STEPS
        res = {
            "metric": "segmentation samples/sec, LIDC 128x128 T=250", "value": total / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_pass,
            "ms_per_denoise_step": ms_dstep, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.prec == "f32" else "f32 (conv products as split fp16 hi/lo x3 on MFMA, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": "C2: LIDCv1-shaped 128x128, 2 classes, T=250 cosine, base-32 U-Net (5.70 M params), "
                                   f"batch={n} per GPU, {'device Philox RNG' if args.rng == 'philox' else 'host torch-CPU Exp(1) noise over PCIe (parity mode)'}, random-init weights",
                       "global_batch": n * world, "time_steps": T_STEPS, "parallelism": f"batch-shard x{world}",
                       "launch": "hip-graph" if args.graph else "eager", "substreams": max(1, min(args.substreams, n))},
        }
        step_bytes = (ALGO_MB_PER_SAMPLE_STEP * n + ALGO_WEIGHTS_MB) * 1e6
        res["roofline_step"] = {"bound": "hbm", "achieved": step_bytes / (ms_dstep * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": step_bytes / (ms_dstep * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "whole denoise step per GPU: SURVEY 8(d) 163.1 MB/sample + 4.82 MB weights, over ms_per_denoise_step"}
        if taps and kern and kern[0][0] > 0:
            cnt = sum(k[0] for k in kern)
            mean_ms = sum(k[0] * k[1] for k in kern) / cnt
            b = dominant_kernel_bytes(n_tap)
            ach = b["total"] / (mean_ms * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_dominant_kernel.json")
            if os.path.exists(pmc) and args.prec == "f16x3":
                # HBM bytes per launch of this kernel from the committed rocprofv3 --pmc passes (tools/pmc_conv.sh;
                # FETCH_SIZE x2 + WRITE_SIZE, KiB, per MI355X_MICROARCH.md) — collected off-line on the same shape at 64
                # samples per launch; every byte of it is per-sample work, so a launch over n_tap samples moves n_tap/64 of it
                traffic = json.load(open(pmc)).get("hbm_bytes") * n_tap / PER_GPU_BATCH
            res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": traffic, "kernel": ("ccdm::k_conv<1, 16, 3, 1, 8, 32, 4, 2, 1, 1>" if args.prec == "f16x3" else "ccdm::k_conv<0, 32, 3, 1, 8, 32, 4, 2, 1, 1>")
                                         + " = <PREC,CK,KS,STRIDE,TH,TW,WAVES,MI,NI,KSP>, engine op 1 (" + eng.op_names[dom] + ": conv3x3 32->32 @128x128, GN+SiLU on load)",
                               "avg_launch_ms": mean_ms, "launches_timed": cnt, "algorithmic_bytes_per_launch": b["total"],
                               "samples_per_launch": n_tap,
                               "concurrent_streams": max(1, min(args.substreams, n)),
                               "achieved_conv_io_only": (b["conv_io"] + b["weights"]) / (mean_ms * 1e-3) / 1e9}
        else:
            res["roofline"] = None
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(sd, torch.from_numpy(image_all[:4]))
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

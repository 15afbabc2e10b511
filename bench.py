#!/usr/bin/env python3
"""Headline benchmark: categorical reverse-diffusion sampling throughput (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config c2|c3shard|c4|c4b64|c5shard]

One "step" = one full pass of the hot path over one batch: the complete T-step sampling of one per-GPU batch
(default BASELINE config C2: 64 LIDC-shaped samples, 128x128, 2 classes, T=250, base-32 U-Net, synthetic image,
random-init weights), including the per-step posterior + categorical draw (device Philox RNG) and, for N > 1, the
final RCCL all_gather of the predictions.  Inputs are resident in HBM when timing starts.  Prints ONE JSON line on
rank 0.  With --gpus N > 1 and no launcher environment the script starts its own ranks
(torch.distributed.run, one process per GPU, 127.0.0.1 rendezvous).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

LIDC_BP = dict(base_channels=32, channel_mult=None, attention_resolutions=[32, 16, 8], num_heads=1,
               num_head_channels=32, softmax_output=True)
DINO_FCE = dict(type="dino", model="dino_vits8", channels=384, conditioning="concat_pixels_concat_features", output_stride=8,
                scale="single", train=False, source_layer=11, target_layer=10)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s is what a streaming copy reaches)
MFMA_PEAK_TFLOPS = 2500.0        # dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_MEASURED_TFLOPS = 1730.0    # what back-to-back v_mfma_f32_32x32x16_f16 sustain with random-mantissa operands (tools/ubench/mfma_chain.hip: 38.7 vs 27.1 "cycles")
# measured in the build container (tools/time_reference_cpu.py -> profiles/r03_reference_vs_oracle_cpu.json: 8 interleaved runs each,
# median of the per-pair ratios, range 0.87-1.07): the oracle takes 0.96x the REAL reference's time per denoise step on the same
# inputs, outputs bit-identical — inside the +-10 % BASELINE.md §3 asks for.  (Round 1's 1.148 came from one sequential pair of runs.)
ORACLE_OVER_REFERENCE_TIME = 0.96

# Workloads (BASELINE.json configs; algorithmic figures per sample per denoise step from SURVEY 8(d) / BASELINE.md §4)
CONFIGS = {
    "c2": dict(title="C2: LIDCv1-shaped 128x128, 2 classes, T=250 cosine, base-32 U-Net (5.70 M params)", H=128, W=128, K=2, C_img=1, T=250,
               batch=64, bp=LIDC_BP, fce=None, algo_mb=163.1, weights_mb=4.82, epilogue_mb=0.52, gflop=8.44, attn_gflop=0.138, image="uniform"),
    "c3shard": dict(title="C3 per-GPU shard: LIDCv1-shaped 128x128, 2 classes, T=1000 cosine, 4 images x 16 samples", H=128, W=128, K=2, C_img=1,
                    T=1000, batch=64, bp=LIDC_BP, fce=None, algo_mb=163.1, weights_mb=4.82, epilogue_mb=0.52, gflop=8.44, attn_gflop=0.138,
                    image="uniform"),
    "c4": dict(title="C4: Cityscapes-shaped 256x512, 20 classes, T=250, DINO feature concat [384,32,64], base-32 U-Net (7.80 M params)", H=256, W=512,
               K=20, C_img=3, T=250, batch=16, bp=LIDC_BP, fce=DINO_FCE, algo_mb=1306.9, weights_mb=13.25, epilogue_mb=41.9, gflop=72.7,
               attn_gflop=6.09, image="normal"),
    "c4b64": dict(title="C4 at base 64: Cityscapes-shaped 256x512, 20 classes, T=250, DINO feature concat, base-64 U-Net (30.6 M params)", H=256, W=512,
                  K=20, C_img=3, T=250, batch=16, bp=dict(LIDC_BP, base_channels=64), fce=DINO_FCE, algo_mb=2581.8, weights_mb=52.0,
                  epilogue_mb=41.9, gflop=270.2, attn_gflop=12.2, image="normal"),
    "c5shard": dict(title="C5 per-GPU shard: Cityscapes-shaped 512x1024, 20 classes, T=250, base-64 7-level U-Net (29.3 M params), attention over 8192 tokens",
                    H=512, W=1024, K=20, C_img=3, T=250, batch=4, bp=dict(LIDC_BP, base_channels=64), fce=None, algo_mb=6502.1, weights_mb=102.5,
                    epilogue_mb=167.8, gflop=625.5, attn_gflop=183.9, image="normal"),
}


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_probe_child(threads: int, n_steps: int, cfg_name: str) -> None:
    """child process of cpu_baseline's all-cores probe: time n_steps oracle denoise steps at N=4 with `threads` threads"""
    from ccdm_stochastic_segmentation_amd import build_model, make_synthetic_state_dict
    from oracle import ccdm_oracle as O
    cfg = CONFIGS[cfg_name]
    T, K, H, W, C_img = cfg["T"], cfg["K"], cfg["H"], cfg["W"], cfg["C_img"]
    model = build_model(T, "cosine", {"s": 0.008}, [(C_img, H, W), (K, H, W)], (C_img, H, W), "unet_openai", cfg["bp"], "datasets.lidc", "confidence", cfg["fce"])
    sd = {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(model.unet.spec, 0).items()}
    torch.set_num_threads(threads)
    image4 = torch.from_numpy(np.random.default_rng(1234).uniform(-1, 1, (4, C_img, H, W)).astype(np.float32))
    sched = O.make_schedule("cosine", T, {"s": 0.008})
    torch.manual_seed(0)
    idx, _ = O.draw_x_T(4, K, H, W)
    x = O.one_hot_bchw(idx, K)
    t0 = time.perf_counter()
    O.forward_denoising(sd, dict(num_heads=1, num_head_channels=32), sched, x, image4, None, n_steps, "confidence")
    print(json.dumps({"ms_per_denoise_step_n4": (time.perf_counter() - t0) / n_steps * 1e3, "threads": torch.get_num_threads()}))


def cpu_baseline(sd, image4, cfg, cfg_name, seed=0, budget_s=30.0):
    """Oracle (CPU restatement of the reference, torch-CPU fp32) on the same workload at N=4: ONE FULL T-step sampling
    run when it fits the budget (T=250 takes 15-30 s on the GPU box's host), else the first steps extrapolated; at <= 16 threads (more
    threads make torch-CPU slower on these small convs) — plus a bounded probe at os.cpu_count() threads in a child process."""
    from oracle import ccdm_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    T, K, H, W = cfg["T"], cfg["K"], cfg["H"], cfg["W"]
    sched = O.make_schedule("cosine", T, {"s": 0.008})
    torch.manual_seed(seed)
    idx, _ = O.draw_x_T(4, K, H, W)
    x = O.one_hot_bchw(idx, K)
    ocfg = dict(num_heads=1, num_head_channels=32)
    t0 = time.perf_counter()
    O.forward_denoising(sd, ocfg, sched, x, image4, None, 2, "confidence")            # warm-up: 2 steps
    warm = (time.perf_counter() - t0) / 2
    full = warm * T <= budget_s
    n_steps = T if full else int(max(2, min(T, 15.0 / max(warm, 1e-3))))
    t0 = time.perf_counter()
    O.forward_denoising(sd, ocfg, sched, x, image4, None, None if full else n_steps, "confidence")
    dt = time.perf_counter() - t0
    ms_step = dt / n_steps * 1e3
    v = 4.0 / (ms_step * 1e-3 * T)
    res = {"value": v, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": (f"oracle (torch-CPU fp32, {torch.get_num_threads()} threads) N=4, one FULL run of all {T} denoise steps timed ({dt:.1f} s)" if full else
                      f"oracle (torch-CPU fp32, {torch.get_num_threads()} threads) N=4, {n_steps} of {T} denoise steps timed ({dt:.1f} s), extrapolated x{T}/{n_steps}"),
           "full_run": bool(full), "ms_per_denoise_step_n4": ms_step, "host_cpu": cpu_model_name(), "host_logical_cpus": os.cpu_count(),
           "oracle_over_reference_time": ORACLE_OVER_REFERENCE_TIME, "value_reference_equivalent": v * ORACLE_OVER_REFERENCE_TIME,
           "note": "oracle_over_reference_time is the build-container measurement of the oracle against the real reference on identical inputs "
                   "(tools/time_reference_cpu.py, outputs bit-identical); value_reference_equivalent is the estimate for the reference itself on this host"}
    # all logical CPUs (BASELINE.md §3 asks for the figure): one denoise step in a child process, bounded — on a 256-thread host torch-CPU
    # needs over a minute per step for these small convs
    ncpu = os.cpu_count() or 1
    if ncpu > torch.get_num_threads():
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-probe", str(ncpu), "--config", cfg_name], capture_output=True, text=True,
                               timeout=25, env=dict(os.environ, CCDM_BENCH_CHILD="1"))
            d = json.loads(r.stdout.strip().splitlines()[-1])
            res["all_cores"] = {"cores": d["threads"], "ms_per_denoise_step_n4": d["ms_per_denoise_step_n4"],
                                "value": 4.0 / (d["ms_per_denoise_step_n4"] * 1e-3 * T), "sample": "1 denoise step at N=4, extrapolated"}
        except subprocess.TimeoutExpired:
            res["all_cores"] = {"cores": ncpu, "value": None, "sample": "1 denoise step at N=4 did not finish within 25 s (> 6 s per sample-step): "
                                                                        "slower than the 16-thread figure by more than 10x"}
        except Exception as e:       # noqa: BLE001
            res["all_cores"] = {"cores": ncpu, "value": None, "sample": f"probe failed: {e}"}
    return res


def _run_ranks(cmd, env):
    """Run the N-rank job: (rc, tail of its stderr).  stdout passes through (rank 0's JSON line), stderr is echoed and kept — the retry
    decision of self_launch reads it."""
    p = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE, text=True)
    tail = []
    for line in p.stderr:
        sys.stderr.write(line)
        tail.append(line)
        del tail[:-200]
    return p.wait(), "".join(tail)


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script with torch.distributed.run.  The ranks need dmabuf IPC
    (HSA_ENABLE_IPC_MODE_LEGACY=0, exported on these boxes); where this process had to set it itself and the job fails, the job is
    started ONCE more without the override, saying so — only when its stderr names IPC / RCCL (any other failure is returned as it is)."""
    import socket

    def launch(env):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        return _run_ranks(cmd, env)

    env = dict(os.environ)
    set_here = "HSA_ENABLE_IPC_MODE_LEGACY" not in env
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver supports dmabuf IPC only (RCCL needs it)
    env["CCDM_BENCH_CHILD"] = "1"
    rc, err = launch(env)
    ipc_like = any(w in err for w in ("hipIpc", "IPC", "ipc", "NCCL", "RCCL", "nccl", "rccl", "dmabuf"))
    if rc != 0 and set_here and ipc_like:
        print(f"bench.py: the {args.gpus}-rank job failed (rc {rc}) with an IPC / RCCL error while HSA_ENABLE_IPC_MODE_LEGACY=0 was set by bench.py; "
              f"retrying once without it",
              file=sys.stderr, flush=True)
        env.pop("HSA_ENABLE_IPC_MODE_LEGACY")
        env["CCDM_NO_HSA_IPC_OVERRIDE"] = "1"
        rc2, _ = launch(env)
        print(f"bench.py: first attempt rc {rc}, retry rc {rc2}", file=sys.stderr, flush=True)
        rc = rc2
    return rc


def offline_pmc_traffic(config: str, grid_threads: int):
    """HBM bytes per launch of the dominant kernel class from the newest committed PMC summary of this config (profiles/r*_pmc_bench_<cfg>.json,
    tools/pmc_bench.sh).  Only the FALLBACK of measure_pmc_traffic: reported under `traffic_offline`, never as `traffic` — it was measured on
    another visit and another build."""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    files = sorted(glob.glob(os.path.join(root, f"r*_pmc_bench_{config}.json")), key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
    for f in reversed(files):
        try:
            ks = json.load(open(f)).get("kernels", {})
        except Exception:
            continue
        for key, e in ks.items():
            if "k_conv<1, 16, 3, 1, 8, 32, 4, 2, 1, 1, false, false>" in key and f"grid={grid_threads} " in key and "hbm_bytes" in e:
                return {"bytes_per_launch": e["hbm_bytes"], "file": os.path.basename(f), "launches": e.get("launches"),
                        "note": "committed off-line PMC passes over this bench command (another visit, another build): not a measurement of this run"}
    return None


def measure_pmc_traffic(args, grid_threads: int, launches_per_dstep: int, dsteps: int = 8, timeout_s: int = 240):
    """HBM bytes per launch of the dominant kernel class, measured NOW: two short child passes of this same script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the two do not fit one, MI355X_MICROARCH.md §rocprofv3 PMC slots;
    --kernel-trace only beside them), a strided walk of `dsteps` denoise steps of the same workload on ONE stream, eager launches — the form in
    which a dispatch's counters belong to one kernel.  The class is found in the counter rows themselves: among the conv kernels launched
    on `grid_threads` threads, the symbol with the largest total duration, and its launch count must equal launches_per_dstep x the
    denoise steps executed — a stale assumption about the symbol or the tiling gives None, not a wrong number.
    Units per the guide's HBM section: FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE counts half of the bytes of wide (16 B/lane) coalesced
    reads on gfx950 -> doubled.  Returns (dict | None, note)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 is not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled: no nested rocprofv3 passes"
    executions = 2 * dsteps                                     # --warmup 1 --steps 1
    tmp = tempfile.mkdtemp(prefix="ccdm_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(CCDM_BENCH_CHILD="1", TMPDIR="/tmp")
    per = {}
    t_all = time.perf_counter()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            d = os.path.join(tmp, counter)
            # (the matrix-pipe pass carries GRBM_GUI_ACTIVE beside it — another counter block, same pass — as its denominator)
            pmc = [counter] + (["GRBM_GUI_ACTIVE"] if counter.startswith("SQ_") else [])
            cmd = [exe, "--pmc", *pmc, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--config", args.config, "--denoise-steps", str(dsteps), "--steps", "1", "--warmup", "1", "--graph", "0", "--substreams", "1",
                   "--no-cpu-baseline", "--no-secondary", "--prec", args.prec, "--slicing", args.slicing] + (["--batch", str(args.batch)] if args.batch else [])
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, f"the rocprofv3 --pmc {counter} pass did not finish within {timeout_s} s"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"the rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
            rows = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if "ccdm::k_conv<" not in row["Kernel_Name"] or int(row["Grid_Size"]) != grid_threads:
                        continue
                    e = rows.setdefault(row["Kernel_Name"], [0, 0.0, 0.0, 0.0])
                    if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                        e[3] += float(row["Counter_Value"])
                        continue
                    if row["Counter_Name"] != counter:
                        continue
                    e[0] += 1
                    e[1] += float(row["Counter_Value"])
                    e[2] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            if not rows:
                return None, f"no conv kernel on a grid of {grid_threads} threads in the {counter} pass"
            name = max(rows, key=lambda k: rows[k][2])
            cnt, tot, us, gui = rows[name]
            if cnt != launches_per_dstep * executions:
                return None, (f"{counter} pass: the longest-running conv symbol on that grid was launched {cnt} times, expected {launches_per_dstep} x {executions} "
                              f"({name[:120]}): the dominant class is not what bench.py assumes")
            per[counter] = (name, tot / cnt, us / cnt, cnt, gui / cnt)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if len({v[0] for v in per.values()}) != 1:
        return None, "the counter passes disagree on the dominant symbol"
    rd, wr = 2.0 * per["FETCH_SIZE"][1] * 1024.0, per["WRITE_SIZE"][1] * 1024.0
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE: busy cycles summed over the 8 XCDs, so one
    # launch offers GRBM_GUI_ACTIVE / 8 x 1024 SIMD-cycles (tools/pmc_bench_summary.py uses the same rule)
    mf = per["SQ_VALU_MFMA_BUSY_CYCLES"]
    mfma_busy = mf[1] / (mf[4] / 8.0 * 1024.0) if mf[4] > 0 else None
    import re
    return ({"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "launches_counted": per["FETCH_SIZE"][3],
             "kernel": re.sub(r"\(.*", "", per["FETCH_SIZE"][0].replace("void ", "")), "avg_launch_us_under_counters": per["FETCH_SIZE"][2],
             "mfma_busy": mfma_busy, "mfma_busy_cycles_per_launch": mf[1], "gui_active_cycles_per_launch": mf[4],
             "seconds_spent": time.perf_counter() - t_all},
            f"measured in this run: three child passes of this bench command under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc "
            f"SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ({dsteps}-step strided walk, one stream, eager launches); FETCH_SIZE x 2 (gfx950 wide-read "
            f"correction: the L2 fetches whole 128-byte lines and tallies them at 64 B — calibrated on this kernel's own access pattern, "
            f"profiles/r06_dominant_traffic_decomposition.json) + WRITE_SIZE, KiB -> bytes, mean over the class's launches")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="workload (BASELINE.json configs); the headline metric is quoted on c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed extras (roofline taps, per-stage table, single-stream figure)")
    ap.add_argument("--graph", type=int, default=-1, help="1 / 0: replay each denoise step as a HIP graph / launch eagerly (default: the product default, graph)")
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU (default: the config's)")
    ap.add_argument("--denoise-steps", type=int, default=0, help="strided walk of this many denoise steps instead of the full T (diagnostics; not the metric)")
    ap.add_argument("--substreams", type=int, default=-1,
                    help="sample the per-GPU batch as this many contiguous sub-batches on concurrent HIP streams (results are bit-identical); "
                         "0 = automatic (two once the batch holds 32 x 128x128 pixels).  Default: the product default (automatic)")
    ap.add_argument("--rng", choices=["philox", "torch_cpu"], default="philox",
                    help="philox: noise generated in the epilogue kernel (the benchmark); torch_cpu: the parity mode — Exp(1) noise drawn "
                         "on the host in the reference's order and copied over PCIe (host-RNG bound; reported for DESIGN.md, never the headline)")
    ap.add_argument("--prec", choices=["f32", "f16x3", "f16"], default="f16x3",
                    help="conv arithmetic: exact fp32 MFMA, or fp16 hi/lo split x3 MFMA with fp32 accumulate (~2^-22: the benchmarked default); "
                         "f16 = the OPT-IN single-pass fast mode (operands rounded to fp16, outside the parity contract: a diagnostic line, never the metric)")
    ap.add_argument("--slicing", default="throughput", choices=["throughput", "latency"],
                    help="DenoisingModel.slicing: 'latency' = more, shorter conv workgroups per sample (small batches)")
    ap.add_argument("--per-op", default="", help="write the per-op table of the tapped pass to this file (JSON)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic")
    ap.add_argument("--digest", action="store_true", help="add the SHA-256 of every rank's last prediction shard to per_rank (tests: a shard of the "
                                                          "N-rank job equals the same shard sampled alone)")
    ap.add_argument("--emulate-rank", default="", metavar="R/W", help="one process plays rank R of a W-rank job (its inputs, its Philox sample "
                                                                     "offset), without a process group: the single-process side of the shard test")
    ap.add_argument("--ref-value", type=float, default=0.0, help="samples/s of the N = 1 line of the same bench (BENCH-style): the N > 1 line then carries "
                                                                 "weak_scaling_efficiency = value_N / (N x ref)")
    ap.add_argument("--secondary", action="store_true", help="N > 1 only: run rank 0's untimed secondary pass (roofline taps, per-stage table) although the "
                                                             "other ranks wait in the final barrier meanwhile (default at N > 1: skipped; at N = 1 it always runs "
                                                             "unless --no-secondary)")
    ap.add_argument("--cpu-probe", type=int, default=0, help=argparse.SUPPRESS)      # child mode of cpu_baseline's all-cores probe
    args = ap.parse_args()
    if args.cpu_probe:
        _cpu_probe_child(args.cpu_probe, 1, args.config)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))

    from ccdm_stochastic_segmentation_amd import build_model, make_synthetic_state_dict, hip
    from ccdm_stochastic_segmentation_amd.distributed import init_from_env, all_gather_shards, backend_info, barrier as dist_barrier
    from ccdm_stochastic_segmentation_amd.models import auto_substreams
    import torch.distributed as dist

    rank, local_rank, world = init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    play_rank, play_world = rank, world
    if args.emulate_rank:
        assert world == 1, "--emulate-rank is the single-process side of the shard test"
        play_rank, play_world = (int(v) for v in args.emulate_rank.split("/"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())     # (ranks may share a GPU under CCDM_DIST_BACKEND=gloo: tests only)
    torch.cuda.set_device(dev)
    cfg = CONFIGS[args.config]
    n = args.batch or cfg["batch"]
    H, W, K, T, C_img = cfg["H"], cfg["W"], cfg["K"], cfg["T"], cfg["C_img"]

    model = build_model(T, "cosine", {"s": 0.008}, [(C_img, H, W), (K, H, W)], (C_img, H, W), "unet_openai", cfg["bp"],
                        "datasets.lidc" if K == 2 else "datasets.cityscapes", "confidence", cfg["fce"])
    sd = {k: torch.from_numpy(v) for k, v in make_synthetic_state_dict(model.unet.spec, 0).items()}
    model.unet.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    f16 = args.prec == "f16x3"
    single = args.prec == "f16"
    model.prec = hip.PREC_F16X3 if f16 else (hip.PREC_F16 if single else hip.PREC_F32)
    model.rng, model.philox_seed = args.rng, 2024
    # the timed configuration is the product default (DenoisingModel: HIP-graph replay, automatic sub-batching) unless overridden
    if args.graph >= 0:
        model.use_graph = bool(args.graph)
    if args.substreams >= 0:
        model.substreams = args.substreams
    if args.graph >= 0 or args.substreams > 0:
        model.calibrate_mode = False                                # an explicit mode is taken as given (the default measures the modes once, in the warm-up)
    model.sample_offset = play_rank * n                           # Philox counters keyed by global sample index
    model.slicing = args.slicing
    use_graph, substreams = bool(model.use_graph), int(model.substreams)
    nsub = substreams if substreams > 0 else auto_substreams(n, cfg["H"], cfg["W"])
    nsub = max(1, min(nsub, n))

    rng = np.random.default_rng(1234)
    if cfg["image"] == "uniform":
        image_all = rng.uniform(-1, 1, (max(n, 4), C_img, H, W)).astype(np.float32)
    else:
        image_all = rng.standard_normal((max(n, 4), C_img, H, W)).astype(np.float32)
    image = torch.from_numpy(image_all[:n]).to(dev)
    feat = None
    if cfg["fce"] is not None:
        feat = torch.from_numpy(rng.standard_normal((n, 384, H // 8, W // 8)).astype(np.float32)).to(dev)
    x = torch.nn.functional.one_hot(torch.from_numpy(np.random.default_rng(42 + play_rank).integers(0, K, (n, H, W))), K)
    x = x.permute(0, 3, 1, 2).float().to(dev)
    t_arg = {} if not args.denoise_steps else {"t": torch.as_tensor(10000 + args.denoise_steps)}
    n_dsteps = args.denoise_steps or T
    gather_buf = [None]
    t_sample, t_gather = [0.0], [0.0]

    def one_pass(timed=False):
        out = model(x, image, feat, **t_arg)["diffusion_out"]
        if world > 1:
            if timed:                                             # per-rank split of the pass: sampling vs. the path's only collective
                torch.cuda.synchronize()
                t_a = time.perf_counter()
            out, gather_buf[0] = all_gather_shards(out.contiguous(), n * world, world, gather_buf[0])
            if timed:
                torch.cuda.synchronize()
                t_gather[0] += time.perf_counter() - t_a
        return out

    for _ in range(args.warmup):
        one_pass()

    # ---- the timed region: exactly `steps` passes of the product path; no taps, no host synchronisation between passes (N = 1) ----
    torch.cuda.synchronize()
    if world > 1:
        dist_barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass(timed=world > 1)
    torch.cuda.synchronize()
    if world > 1:
        dist_barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if model.last_mode is not None:                                 # what the timed passes ran: the measured choice among the bit-identical modes
        nsub, use_graph = int(model.last_mode[0]), bool(model.last_mode[1])
    mode_choice = next(iter(model.mode_choice.values()), None)
    per_rank = None
    digest = None
    if args.digest:
        import hashlib
        lo = play_rank * n if world > 1 else 0                      # (N > 1: `out` is the gathered batch; this rank's shard of it)
        digest = hashlib.sha256(out[lo:lo + n].contiguous().cpu().numpy().tobytes()).hexdigest()
    if world > 1:
        # the per-rank record travels as HOST tensors (gloo in the mixed group): pass / gather seconds, device name, shard digest
        mine = torch.tensor([dt, t_gather[0]], dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        text = json.dumps({"device": torch.cuda.get_device_name(dev), "cuda_device": int(dev.index), "out_sha256": digest,
                           "mode": f"{nsub} stream{'s' if nsub > 1 else ''}, {'graph' if use_graph else 'eager'}"}).encode()[:384]
        blob = torch.zeros(384, dtype=torch.uint8)
        blob[:len(text)] = torch.frombuffer(bytearray(text), dtype=torch.uint8)
        blobs = [torch.empty_like(blob) for _ in range(world)]
        dist.all_gather(blobs, blob)
        per_rank = [dict({"rank": r, "pass_s": float(v[0]) / max(args.steps, 1), "sampling_s": float(v[0] - v[1]) / max(args.steps, 1),
                          "gather_s": float(v[1]) / max(args.steps, 1),
                          "ms_per_denoise_step": float(v[0] - v[1]) / max(args.steps, 1) / n_dsteps * 1e3,      # this rank's sampling alone (no gather)
                          "samples_per_s": n * max(args.steps, 1) / max(float(v[0]), 1e-12)},
                         **json.loads(bytes(b.tolist()).rstrip(b"\0").decode()))
                    for r, (v, b) in enumerate(zip(allr, blobs))]
        dt = max(float(v[0]) for v in allr)
    assert torch.isfinite(out).all()

    res = None
    if rank == 0:
        total = n * world * args.steps
        ms_pass = dt / max(args.steps, 1) * 1e3
        ms_dstep = ms_pass / n_dsteps
        res = {
            "metric": f"segmentation samples/sec, {'LIDC 128x128' if K == 2 else f'Cityscapes {H}x{W}'} T={T}", "value": total / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_pass,
            "ms_per_denoise_step": ms_dstep, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (conv products as split fp16 hi/lo x3 on MFMA, fp32 accumulate)" if f16 else
                      "f16 operands (ONE fp16 MFMA per product, fp32 accumulate): opt-in fast mode, NARROWER than the reference's arithmetic - not the metric" if single
                      else "f32"), "data": "synthetic",
            "config": {"workload": f"{cfg['title']}, batch={n} per GPU, "
                                   f"{'device Philox RNG' if args.rng == 'philox' else 'host torch-CPU Exp(1) noise over PCIe (parity mode)'}, random-init weights",
                       "name": args.config, "global_batch": n * world, "time_steps": T, "denoise_steps_run": n_dsteps,
                       "parallelism": f"batch-shard x{world}", "launch": "hip-graph" if use_graph else "eager", "substreams": nsub,
                       "slicing": args.slicing,
                       "mode": ("measured in the warm-up: fastest of the bit-identical execution modes on this box (DenoisingModel.calibrate_mode)"
                                if mode_choice is not None else "as configured"),
                       "mode_ms_per_denoise_step": mode_choice["ms_per_denoise_step"] if mode_choice is not None else None},
        }
        if per_rank is not None:
            res["per_rank"] = per_rank
            res["distributed"] = backend_info()            # backend, RCCL / HIP versions, IPC mode, RCCL probe: first collective (communicator setup) and steady all_reduce latency
            res["slowest_rank"] = max(per_rank, key=lambda r: r["pass_s"])["rank"]
            res["gather_share_of_pass"] = max(r["gather_s"] for r in per_rank) / max(dt / max(args.steps, 1), 1e-12)
        if args.ref_value > 0:
            res["weak_scaling_efficiency"] = res["value"] / (world * args.ref_value)
            res["weak_scaling_ref_value"] = args.ref_value
        elif digest is not None:
            res["out_sha256"] = digest
        if args.emulate_rank:
            res["config"]["emulated_rank"] = args.emulate_rank
        step_bytes = (cfg["algo_mb"] * n + cfg["weights_mb"]) * 1e6
        fr = step_bytes / (ms_dstep * 1e-3) / 1e9
        res["roofline_step"] = {"bound": "hbm", "achieved": fr, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fr / HBM_PEAK_GBS,
                                "note": f"whole denoise step per GPU in the TIMED region: SURVEY 8(d) {cfg['algo_mb']} MB/sample + {cfg['weights_mb']} MB weights, over ms_per_denoise_step",
                                "algorithmic_tflops": cfg["gflop"] * n / ms_dstep, "mfma_util": (3.0 if f16 else (1.0 if single else 16.0)) * cfg["gflop"] * n / ms_dstep / MFMA_PEAK_TFLOPS}
        res["roofline"] = None

    # ---- untimed, rank 0 at N = 1: ONE single-stream eager pass with HIP-event taps on every op (on the engine's stream) -> the dominant
    #      kernel's roofline, the per-stage split, the per-op table; then the single-stream figure of the same product path ----
    def secondary():
        # (N > 1: rank 0 alone, on its own shard, without the gather — the other ranks wait at the final barrier)
        def one_pass():                                            # noqa: F811
            return model(x, image, feat, **t_arg)["diffusion_out"]
        model.substreams, model.use_graph = 1, False
        one_pass()                                                 # builds the whole-batch executor
        torch.cuda.synchronize()
        eng = model._engine(x, image, feat, slot=0)
        info = eng.op_info
        # The dominant kernel = the conv instantiation of the full-resolution stage: every 3x3 stride-1 conv whose output is HxW runs
        # the same k_conv<...> symbol on the same grid (C2: 9 launches per denoise step) — the unit rocprofv3's per-kernel statistics
        # report.  (The Upsample conv that lands on HxW runs the sub-pixel instantiation, the convs with a fused 1x1 skip of 32-channel
        # sources the core-only skip-chunk instantiation: other symbols — in the per-op table, not in this class.)
        dom_ops = [i for i, o in enumerate(info) if o["kind"] == "conv" and o["k"] == 3 and o["stride"] == 1 and (o["hout"], o["wout"]) == (H, W)
                   and not o.get("subpixel") and not o.get("skip_wide") and not o.get("stem") and not o.get("head_fused")]
        attn_ops = [i for i, o in enumerate(info) if o["kind"] == "attention" and o["T"] >= 2048]
        attn = max(attn_ops, key=lambda i: info[i]["T"]) if attn_ops else None
        for i in range(len(info)):
            eng.profile_op(i, capacity=n_dsteps)
        one_pass()
        torch.cuda.synchronize()
        taps = [eng.profile_read(i) for i in range(len(info))]       # (launches, mean ms, min ms, max ms)
        eng.profile_op(-1)
        tap_note = (f"HIP events on the engine's stream around every launch of one extra UNTIMED pass of the same workload on ONE stream, eager launches "
                    f"({n_dsteps} denoise steps); the timed region runs {nsub} concurrent sub-batch stream(s) "
                    f"{'as HIP-graph replays' if use_graph else 'eagerly'} without taps — under concurrency a launch's event duration no longer describes the kernel")
        if dom_ops and all(taps[i][0] for i in dom_ops):
            tot_ms = sum(taps[i][1] for i in dom_ops)
            bytes_all = sum(info[i]["io_bytes"] * n + info[i]["gn_read_bytes"] * n + info[i]["weight_bytes"] for i in dom_ops)
            bytes_io = sum(info[i]["io_bytes"] * n + info[i]["weight_bytes"] for i in dom_ops)
            flop = sum(info[i]["flop"] * n for i in dom_ops)
            ach = bytes_all / (tot_ms * 1e-3) / 1e9
            step_ms_tapped = sum(t_[1] for t_ in taps)
            shapes = {}
            for i in dom_ops:
                o = info[i]
                key = f"{o['cin']}->{o['cout']}" + (" +fused 1x1 skip" if o["skip"] else "") + (" up2x" if o["up"] else "") + ("" if o["gn"] else " (no GroupNorm)")
                sh = shapes.setdefault(key, dict(launches_per_denoise_step=0, ms=0.0, bytes=0, bytes_io=0, flop=0, ops=[]))
                sh["launches_per_denoise_step"] += 1
                sh["ms"] += taps[i][1]
                sh["bytes"] += o["io_bytes"] * n + o["gn_read_bytes"] * n + o["weight_bytes"]
                sh["bytes_io"] += o["io_bytes"] * n + o["weight_bytes"]
                sh["flop"] += o["flop"] * n
                sh["ops"].append(i)
            grid_threads = n * hip.load().ccdm_conv_slices(H, W, 1, 3) * 256
            traffic = traffic_note = None
            if f16 and not args.no_pmc and world == 1:          # (N > 1: the other ranks sit in the final barrier meanwhile — no child passes there)
                traffic, traffic_note = measure_pmc_traffic(args, grid_threads, len(dom_ops))
            # what a launch of the class HAS to move: conv input + output + weights + the identity-residual read of a ResBlock's second
            # conv (round 5 left that read out: 326 instead of 364 MB at C2)
            must_move = (bytes_io + sum(info[i].get("resid_bytes", 0) * n for i in dom_ops)) / len(dom_ops)
            res["roofline"] = {
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic["bytes_per_launch"] if traffic else None,
                "traffic_source": traffic_note if traffic else f"not measured in this run ({traffic_note or ('N > 1: the counter passes belong to the N = 1 line' if world > 1 else '--no-pmc')}); see traffic_offline",
                "traffic_detail": traffic,
                "traffic_offline": None if traffic else offline_pmc_traffic(args.config, grid_threads),
                "must_move_bytes_per_launch": must_move,
                "frac_vs_must_move": (traffic["bytes_per_launch"] / must_move) if traffic else None,
                "frac_vs_must_move_note": "bytes the L2's memory-side counters saw per launch over the bytes that must move (conv input + output + weights + the residual read of "
                                          "a ResBlock's second conv; no GroupNorm-read credit).  FETCH_SIZE counts Infinity-Cache hits too: an upper bound on HBM bytes",
                "note": tap_note,
                "kernel": f"ccdm::k_conv<F16X3,16,3,1,8,32,4,2,1,1> (<PREC,CK,KS,STRIDE,TH,TW,WAVES,MI,NI,KSP>): every 3x3 stride-1 conv of the {H}x{W} stage "
                          f"(engine ops {dom_ops}), GN+SiLU on load" if f16 else f"ccdm::k_conv<F32,...> every 3x3 stride-1 conv of the {H}x{W} stage (engine ops {dom_ops})",
                "launches_per_denoise_step": len(dom_ops), "avg_launch_ms": tot_ms / len(dom_ops), "launches_timed": sum(taps[i][0] for i in dom_ops),
                "algorithmic_bytes_per_launch": bytes_all / len(dom_ops), "samples_per_launch": n, "concurrent_streams": 1,
                "achieved_conv_io_only": bytes_io / (tot_ms * 1e-3) / 1e9, "frac_conv_io_only": bytes_io / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "share_of_denoise_step": tot_ms / max(step_ms_tapped, 1e-9),
                "algorithmic_tflops": flop / (tot_ms * 1e-3) / 1e12,
                # matrix-pipe utilisation: the COUNTER when this run measured it (SQ_VALU_MFMA_BUSY_CYCLES over the SIMD-cycles the launch
                # offered), else the instruction count (every fp32 product is 3 fp16 MFMA products) against the dense fp16 peak at the top clock
                "mfma_util": traffic["mfma_busy"] if traffic and traffic.get("mfma_busy") is not None else
                             (3.0 if f16 else 16.0) * flop / (tot_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                "mfma_util_analytic": (3.0 if f16 else 16.0) * flop / (tot_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                "mfma_util_source": ("counter: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of this run's PMC pass, mean over the class's launches; "
                                     if traffic and traffic.get("mfma_busy") is not None else "analytic (no counter pass in this run); ") +
                                    "mfma_util_analytic = MFMA FLOPs issued (3 fp16 products per fp32 product) / HIP-event duration / 2.5 PFLOP/s dense "
                                    "(tools/ubench/mfma_chain.hip: with random-mantissa operands the matrix pipe sustains 1.73 PFLOP/s — its clock is "
                                    "data-dependent; against that rate the analytic figure is x1.45)",
            }
            res["roofline_shapes"] = {
                k_: {"launches_per_denoise_step": v["launches_per_denoise_step"], "avg_launch_ms": v["ms"] / v["launches_per_denoise_step"],
                     "frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_conv_io_only": v["bytes_io"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "mfma_util": (3.0 if f16 else 16.0) * v["flop"] / (v["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "engine_ops": v["ops"]}
                for k_, v in shapes.items()}
        if attn is not None and taps[attn][0]:
            m_a, c_a = taps[attn][1], taps[attn][0]
            o = info[attn]
            fl = o["flop"] * n
            res["roofline_attention"] = {
                "bound": "mfma", "achieved": fl / (m_a * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fl / (m_a * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "mfma_util": 3.0 * fl / (m_a * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                "avg_launch_ms": m_a, "launches_timed": c_a, "traffic": None,
                "kernel": f"ccdm::k_attention_mfma (engine op {attn}, {eng.op_names[attn]}: T={o['T']}, C={o['C']}, {o['heads']} heads of {o['C'] // o['heads']})",
                "mfma_util_vs_measured_rate": 3.0 * fl / (m_a * 1e-3) / 1e12 / MFMA_MEASURED_TFLOPS,
                "note": "frac = algorithmic FLOPs (4*T*T*C per sample) over the dense fp16 MFMA peak; mfma_util counts the 3 fp16 MFMAs each product is made of; "
                        "mfma_util_vs_measured_rate prices them against the 1.73 PFLOP/s the matrix pipe sustains on random-mantissa operands "
                        "(tools/ubench/mfma_chain.hip: the MFMA clock is data-dependent); " + tap_note}
        per_op = []
        for i, o in enumerate(info):
            cnt, m, lo, hi = taps[i]
            per_op.append(dict(op=i, name=o["name"], kind=o["kind"], mean_us=m * 1e3, min_us=lo * 1e3, max_us=hi * 1e3,
                               shape=(f"{o['cin']}->{o['cout']} k{o['k']} @{o['hout']}x{o['wout']}" if o["kind"] in ("conv", "conv_head") else
                                      (f"T={o['T']} C={o['C']}" if "T" in o else "")),
                               hbm_frac=(o["io_bytes"] * n + o["gn_read_bytes"] * n + o["weight_bytes"]) / max(m, 1e-9) / 1e6 / HBM_PEAK_GBS))
        by_stage = {}
        for p in per_op:
            o = info[p["op"]]
            key = f"{o['hout']}x{o['wout']}" if o["kind"] in ("conv", "conv_head", "resample") else o["kind"]
            by_stage[key] = by_stage.get(key, 0.0) + p["mean_us"]
        res["per_stage_us"] = {k_: round(v, 1) for k_, v in by_stage.items()}
        res["per_stage_note"] = "sum of per-op mean launch times of the tapped single-stream pass, grouped by output size"
        if args.per_op:
            with open(args.per_op, "w") as fh:
                json.dump(per_op, fh, indent=1)
        # the same product path on ONE stream (graph replay as configured): what a kernel-by-kernel reading of the step adds up to
        model.use_graph = use_graph
        one_pass()
        torch.cuda.synchronize()
        reps = max(1, min(args.steps, 3))
        t1 = time.perf_counter()
        for _ in range(reps):
            one_pass()
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / reps
        res["single_stream"] = {"value": n / dt1, "unit": "samples/s", "ms_per_denoise_step": dt1 / n_dsteps * 1e3,
                                "roofline_step_frac": (cfg["algo_mb"] * n + cfg["weights_mb"]) * 1e6 / (dt1 / n_dsteps) / 1e9 / HBM_PEAK_GBS, "passes": reps,
                                "note": "same workload on one stream (substreams = 1), same launch mode: secondary figure"}
        model.substreams = substreams

    # N > 1: the untimed extras are opt-IN (--secondary) — while rank 0 runs them (about a minute) the other ranks sit in the final barrier,
    # under RCCL's watchdog; the roofline objects belong to the N = 1 line anyway
    if rank == 0 and res is not None and world > 1 and not args.secondary:
        res["secondary_note"] = "N > 1: rank 0's untimed roofline / per-stage pass skipped (pass --secondary to run it); those objects are on the N = 1 line"
    if rank == 0 and not args.no_secondary and (world == 1 or args.secondary):
        if world == 1:
            secondary()
        else:                                                      # the N-rank line must not die on an untimed extra
            try:
                secondary()
            except Exception as e:       # noqa: BLE001
                res["secondary_error"] = f"{type(e).__name__}: {e}"
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and args.config in ("c2", "c3shard"):
            res["cpu_baseline"] = cpu_baseline(sd, torch.from_numpy(image_all[:4]), cfg, args.config)
        print(json.dumps(res))
    if world > 1:
        dist_barrier()
        from ccdm_stochastic_segmentation_amd.distributed import rccl_hung
        if rccl_hung():                                            # this rank's RCCL probe thread never returned: tearing the group down would block on it
            sys.stdout.flush()
            os._exit(0)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

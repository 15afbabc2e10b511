"""Batch sharding of the sampler over the GPUs of one node.

Every sample (image x draw) is independent through all T steps, so the flattened batch is split into
contiguous ranges, one process per GPU, with NO collective inside the T loop; the only exchange is one
all_gather of the final predictions (torch.distributed backend "nccl" == RCCL over xGMI; "gloo" on CPU
for tests).  Noise is keyed by global sample index (Philox) or sliced from a full-batch host draw
(torch_cpu), so results do not depend on the number of ranks.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of rank's samples; the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


# what init_from_env brought up: "nccl" (= RCCL; CPU tensors of the same group travel over gloo), "gloo", or None (group made elsewhere)
_STATE = {"backend": None, "data_via_host": False, "rccl_error": None, "hsa_ipc_legacy_set_here": False,
          "rccl_probe_first_ms": None, "rccl_probe_ms": None, "rccl_hung": False}


def _probe_rccl(local_rank: int, world: int, timeout_s: float):
    """The first device collective (RCCL builds its communicator there) and a second one (its steady-state latency), in a daemon
    thread the caller waits for at most `timeout_s`: a communicator that fails on ONE rank only leaves the other ranks blocked inside
    the collective — they must still reach the gloo flag exchange.  Returns (error | None, first ms, steady ms, hung)."""
    import threading
    import time
    box = {"err": None, "first": None, "steady": None, "done": False}

    def run():
        try:
            torch.cuda.set_device(local_rank)
            probe = torch.ones(1, device=torch.device("cuda", local_rank))
            t0 = time.perf_counter()
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            box["first"] = (time.perf_counter() - t0) * 1e3
            if int(probe.item()) != world:
                box["err"] = f"RCCL all_reduce of ones over {world} ranks returned {probe.item()}"
            else:
                t0 = time.perf_counter()
                for _ in range(5):
                    dist.all_reduce(probe)
                torch.cuda.synchronize()
                box["steady"] = (time.perf_counter() - t0) * 1e3 / 5
        except Exception as e:       # noqa: BLE001   (RCCL reports through RuntimeError / DistBackendError)
            box["err"] = f"{type(e).__name__}: {e}"
        box["done"] = True

    th = threading.Thread(target=run, name="ccdm-rccl-probe", daemon=True)
    th.start()
    th.join(timeout_s)
    if not box["done"]:
        return (f"the first RCCL collective did not return within {timeout_s:.0f} s on this rank (another rank's communicator probably "
                f"failed): treated as an RCCL failure"), None, None, True
    return box["err"], box["first"], box["steady"], False


def init_from_env(backend: Optional[str] = None, force: bool = False) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the process group if world > 1 (or `force`).

    GPU ranks get ONE group with two transports, "cpu:gloo,cuda:nccl": device tensors travel over RCCL / xGMI, host tensors over
    gloo.  RCCL builds its communicator at the first device collective, so that collective is issued HERE, on one element; if it
    raises on any rank, every rank learns it over gloo, the RCCL error is printed with the versions and the IPC setting, and the
    path's one exchange (the final gather) travels through the host instead — loudly (`backend_info()` says so, bench.py puts it
    in its JSON line): a sampling job whose only collective is one gather should not die because peer-to-peer IPC is unavailable.
    The probe runs in a monitored thread (CCDM_RCCL_PROBE_TIMEOUT_S, default 60 s): when the communicator fails on ONE rank only, the
    other ranks sit inside the collective — after the timeout they count as failed too and join the flag exchange.  Their probe
    threads stay blocked in RCCL, so such a process must not call destroy_process_group (`rccl_hung()`: bench.py exits without it).
    A rank that dies outright (no exception, no timeout) still takes the job down with gloo's own timeout: that is not recoverable here."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force) and not dist.is_initialized():
        if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ and not os.environ.get("CCDM_NO_HSA_IPC_OVERRIDE"):
            os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"          # dmabuf IPC only on these hosts (RCCL / shared device tensors)
            _STATE["hsa_ipc_legacy_set_here"] = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            # (CCDM_DIST_BACKEND=gloo: several ranks sharing one GPU — how the multi-rank paths are exercised on a one-GPU box)
            backend = os.environ.get("CCDM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world)
            _STATE["backend"] = "nccl"
            err, first_ms, steady_ms, hung = _probe_rccl(local_rank, world, float(os.environ.get("CCDM_RCCL_PROBE_TIMEOUT_S", "60")))
            _STATE["rccl_probe_first_ms"], _STATE["rccl_probe_ms"], _STATE["rccl_hung"] = first_ms, steady_ms, hung
            flag = torch.tensor([0 if err is None else 1], dtype=torch.int32)
            dist.all_reduce(flag)                                    # host tensor: gloo
            if int(flag.item()) != 0:
                _STATE["data_via_host"] = True
                _STATE["rccl_error"] = err or "RCCL failed on another rank"
                import sys
                print(f"[ccdm rank {rank}] RCCL did not come up ({_STATE['rccl_error']}); {backend_info()}; "
                      "the final gather travels through the host (gloo)", file=sys.stderr, flush=True)
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            _STATE["backend"] = backend
    return rank, local_rank, world


def data_via_host() -> bool:
    """True when collectives of device tensors must go through host memory: gloo groups (CPU tests, several ranks on one GPU) and
    the loud fallback of init_from_env."""
    if _STATE["data_via_host"]:
        return True
    b = _STATE["backend"] or (str(dist.get_backend()) if dist.is_initialized() else "gloo")
    return "nccl" not in b


def barrier() -> None:
    """Barrier over every rank: RCCL's when it carries the data, otherwise a one-element gloo all_reduce (no device involved)."""
    if not dist.is_initialized():
        return
    if data_via_host():
        dist.all_reduce(torch.zeros(1, dtype=torch.int32))
    else:
        dist.barrier(device_ids=[torch.cuda.current_device()])


def rccl_hung() -> bool:
    """True when this rank's RCCL probe never returned (its thread is still inside the collective): skip destroy_process_group."""
    return bool(_STATE["rccl_hung"])


def backend_info() -> dict:
    """What the N > 1 path runs on — for bench.py's JSON line and the fallback message."""
    info = {"backend": _STATE["backend"], "data_via_host": bool(_STATE["data_via_host"] or (_STATE["backend"] or "gloo") != "nccl"),
            "rccl_error": _STATE["rccl_error"], "rccl_probe_first_ms": _STATE["rccl_probe_first_ms"], "rccl_probe_allreduce_ms": _STATE["rccl_probe_ms"],
            "torch": torch.__version__, "hip": getattr(torch.version, "hip", None),
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "HSA_ENABLE_IPC_MODE_LEGACY_set_by": "ccdm" if _STATE["hsa_ipc_legacy_set_here"] else "environment"}
    try:
        v = torch.cuda.nccl.version()
        info["rccl"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:        # noqa: BLE001
        info["rccl"] = None
    return info


def _check_same_host_rng(world: int, device) -> None:
    """rng="torch_cpu" slices each rank's noise out of a full-batch draw from torch's global CPU generator: that is only the
    single-process result if every rank's generator is in the same state.  Compare a hash of the states (cheap) and fail
    loudly when they differ."""
    import hashlib
    h = int.from_bytes(hashlib.sha256(torch.get_rng_state().numpy().tobytes()).digest()[:7], "little")
    mine = torch.tensor([h], dtype=torch.int64, device=device)
    all_ = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(all_, mine)
    if any(int(v.item()) != h for v in all_):
        raise RuntimeError("sample_sharded with rng='torch_cpu': the ranks' torch CPU generators are in different states, so the "
                           "shards would not reproduce the single-process samples; seed every rank identically (torch.manual_seed) "
                           "or use rng='philox'")


def sample_sharded(model, x: torch.Tensor, condition: torch.Tensor, feature_condition: Optional[torch.Tensor] = None,
                   t: Optional[torch.Tensor] = None, gather=True) -> torch.Tensor:
    """Run `model` (a DenoisingModel-like callable) on this rank's shard of the global batch and return the
    full [N,K,H,W] prediction on every rank, or only the local shard (gather=False).
    gather=True: fp32 probabilities travel as they are; int64 one-hot ("majority") predictions travel as their uint8 argmax map
    (K*8 x fewer bytes: 16 MB instead of 1.3 GB at BASELINE config C5) and are rebuilt on arrival;
    gather="index": force the index form for any output (fp32 one-hot of a shortened walk)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = x.shape[0]
    lo, hi = shard_range(n, rank, world)
    if world > 1 and getattr(model, "rng", None) == "torch_cpu":
        _check_same_host_rng(world, "cpu" if data_via_host() else x.device)
    model.sample_offset = lo
    model.noise_slice = (n, lo)
    try:
        kw = {} if t is None else {"t": t}
        fc = feature_condition[lo:hi] if feature_condition is not None else None
        out = model(x[lo:hi], condition[lo:hi], fc, **kw)["diffusion_out"].contiguous()
    finally:
        model.sample_offset = 0
        model.noise_slice = None
    if world == 1 or not gather:
        return out
    # one-hot ("majority") predictions are int64 in the reference's contract: only their uint8 argmax map needs to travel
    # (K * 8 x fewer bytes: 16 MB instead of 1.3 GB at BASELINE config C5), the one-hot is rebuilt on arrival
    if gather == "index" or out.dtype == torch.int64:
        K = out.shape[1]
        idx = all_gather_ragged(out.argmax(dim=1).to(torch.uint8).contiguous(), n, world)
        return torch.nn.functional.one_hot(idx.long(), K).permute(0, 3, 1, 2).to(out.dtype)
    return all_gather_ragged(out, n, world)


def all_gather_shards(local: torch.Tensor, n: int, world: int, buf: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """ONE all_gather_into_tensor of per-rank shards whose first dims follow shard_range, into a preallocated buffer (pass the returned
    `buf` back in on the next call: no per-call allocation, one collective whatever the rank count).  Ragged shards are padded to the
    largest.  RCCL moves device tensors directly; under the gloo backend (CPU tests, or several ranks sharing one GPU) the shards
    travel through the host.  Returns (full [n, ...] tensor on local's device, buf)."""
    sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    m = max(sizes)
    dev = local.device
    pad = local.contiguous()
    if data_via_host() and local.is_cuda:
        pad = pad.cpu()
    if pad.shape[0] < m:
        pad = torch.cat([pad, pad.new_zeros((m - pad.shape[0],) + tuple(pad.shape[1:]))], 0)
    shape = (world * m,) + tuple(pad.shape[1:])
    if buf is None or tuple(buf.shape) != shape or buf.dtype != pad.dtype or buf.device != pad.device:
        buf = torch.empty(shape, dtype=pad.dtype, device=pad.device)
    dist.all_gather_into_tensor(buf, pad)
    if all(sz == m for sz in sizes):
        full = buf
    else:
        full = torch.cat([buf[r * m:r * m + sz] for r, sz in enumerate(sizes)], 0)
    return full.to(dev), buf


def all_gather_ragged(local: torch.Tensor, n: int, world: int) -> torch.Tensor:
    """all_gather_shards without buffer reuse (one-off calls)."""
    return all_gather_shards(local, n, world)[0]

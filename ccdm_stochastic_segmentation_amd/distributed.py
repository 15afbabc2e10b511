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


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the process group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def sample_sharded(model, x: torch.Tensor, condition: torch.Tensor, feature_condition: Optional[torch.Tensor] = None,
                   t: Optional[torch.Tensor] = None, gather: bool = True) -> torch.Tensor:
    """Run `model` (a DenoisingModel-like callable) on this rank's shard of the global batch and return the
    full [N,K,H,W] prediction on every rank (gather=True) or only the local shard."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = x.shape[0]
    lo, hi = shard_range(n, rank, world)
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
    return all_gather_ragged(out, n, world)


def all_gather_ragged(local: torch.Tensor, n: int, world: int) -> torch.Tensor:
    """all_gather of per-rank shards whose first dims follow shard_range (pads to the largest shard)."""
    sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    m = max(sizes)
    pad = local
    if local.shape[0] < m:
        pad = torch.cat([local, local.new_zeros((m - local.shape[0],) + tuple(local.shape[1:]))], 0)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)

"""Batch sharding for the one-process-per-GPU layout.

IK problems are independent, so rank r of P owns the contiguous rows
[lo, hi) of the global batch and no data moves between ranks while solving.  Synthetic
inputs are generated per fixed-size block of global rows from (seed, block), which
makes every row identical at any process count.  The only collectives are one
all-reduce of a handful of run statistics and, on request, a gather of the results.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

from . import workloads as W

BLOCK = 4096


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split; the first ``total % world`` ranks hold one extra row."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_rows(model, seed: int, lo: int, hi: int) -> Tuple[np.ndarray, np.ndarray]:
    """(q0, q_star) for global rows [lo, hi): start configurations and the independent
    samples whose site pose is the (reachable) target."""
    q0 = np.empty((hi - lo, model.nq))
    qs = np.empty((hi - lo, model.nq))
    for blk in range(lo // BLOCK, (max(hi, lo + 1) - 1) // BLOCK + 1):
        rng = np.random.default_rng([seed, blk])
        a, b = W.sample_q(model, rng, BLOCK), W.sample_q(model, rng, BLOCK)
        g0 = blk * BLOCK
        s, e = max(lo, g0), min(hi, g0 + BLOCK)
        if e > s:
            q0[s - lo:e - lo] = a[s - g0:e - g0]
            qs[s - lo:e - lo] = b[s - g0:e - g0]
    return q0, qs


def reduce_stats(sums: Dict[str, float], maxes: Dict[str, float], dist=None, device="cpu") -> Dict[str, float]:
    """Whole-job statistics: `sums` are added and `maxes` maximised over ranks (two small
    all-reduces; a no-op without a process group)."""
    ks, km = sorted(sums), sorted(maxes)
    ts = torch.tensor([float(sums[k]) for k in ks], dtype=torch.float64, device=device)
    tm = torch.tensor([float(maxes[k]) for k in km], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if ts.numel():
            dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        if tm.numel():
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    out = {k: float(v) for k, v in zip(ks, ts.tolist())}
    out.update({k: float(v) for k, v in zip(km, tm.tolist())})
    return out


def gather_rows(local: torch.Tensor, total: int, dist=None) -> torch.Tensor:
    """All-gather of row shards laid out by :func:`shard_bounds` (ragged shards are padded
    to the largest one for the collective and trimmed afterwards)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)

"""Instance sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests).

Robots are independent (nothing in the reference couples two robots), so the batch shards by contiguous instance ranges
with NO data-path collective while stepping; the only exchange is the all-gather of the final joint-state buffer
(BASELINE.json north_star).  Inputs are keyed by the GLOBAL instance id so that a sharded run reproduces the
single-process run instance for instance.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_total: int, rank: int, world: int):
    """Contiguous range [lo, hi) of global instance ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def instance_uniform(seed: int, ids: np.ndarray, stream: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """Counter-based uniform variates keyed by (seed, stream, global instance id): SplitMix64 -> [lo, hi).
    Shards draw exactly the values the unsharded batch would draw for the same instances."""
    with np.errstate(over="ignore"):
        off = np.array([stream + 1], dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        x = (ids.astype(np.uint64) + off) ^ np.uint64(seed)
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def velocity_inputs(seed: int, lo: int, hi: int, min_speed: float = 0.2):
    """Throttle-mode velocity commands of instances [lo, hi): linear uniform in the unit disc (|v| >= min_speed),
    angular uniform in [-1, 1]."""
    ids = np.arange(lo, hi, dtype=np.int64)
    r = np.sqrt(instance_uniform(seed, ids, 0, min_speed ** 2, 1.0))
    th = instance_uniform(seed, ids, 1, 0.0, 2.0 * np.pi)
    lin = np.stack([r * np.cos(th), r * np.sin(th)], axis=1)
    ang = instance_uniform(seed, ids, 2, -1.0, 1.0)
    return lin, ang


def all_gather_joints(local, world: int, out=None):
    """All-gather equally sized per-rank joint-state shards (torch tensors, any device) into one tensor ordered by rank:
    THE exchange step of the sharded path (bench.py on RCCL, tests/test_sharding_gloo.py on gloo).  `out` (optional) is a
    preallocated world * local.numel() tensor, so that a timed loop does not allocate.  Returns the gathered tensor."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        out.copy_(local.reshape(-1))
        return out
    dist.all_gather_into_tensor(out, local.reshape(-1).contiguous())
    return out

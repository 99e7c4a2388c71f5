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


class PeerAllGather:
    """The same exchange as `all_gather_joints`, as peer copies over xGMI instead of a collective (include/shc_batch.h: shc_peer_*): every rank
    owns a gathered buffer [world][shard], exports it, opens its peers' and, per exchange, writes its shard into every buffer at its own offset -
    world - 1 copies on world - 1 streams (one per xGMI link) + the local one.  `barrier` (a callable every rank enters, e.g. bench.py's
    HostSpinBarrier or torch.distributed.barrier) closes an exchange: when it returns on a rank, that rank's buffer holds every shard.  The same kind of barrier must also PRECEDE
    every exchange but the first (`pre_barrier`): a rank's copies land in its peers' buffers, and nothing else tells it that the peers have finished reading
    what the previous exchange left there.
    torch.distributed (any backend) is only used once, to exchange the 64-byte handles."""

    def __init__(self, shard_numel: int, world: int, rank: int, device: int):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import engine as _engine
        self.L, self.C, self.world, self.rank, self.device = _engine.lib(), C, world, rank, device
        self.shard_bytes = int(shard_numel) * 8
        own, handle = C.c_void_p(), C.create_string_buffer(64)
        _engine._check(self.L.shc_peer_alloc(device, self.shard_bytes * world, C.byref(own), handle), "shc_peer_alloc")
        self.own = own.value
        handles = [None] * world
        if world > 1 or (dist.is_available() and dist.is_initialized()):
            dist.all_gather_object(handles, bytes(handle.raw))
        else:
            handles[0] = bytes(handle.raw)
        self.peers = []
        for r in range(world):
            if r == rank:
                self.peers.append(self.own)
                continue
            p = C.c_void_p()
            _engine._check(self.L.shc_peer_open(device, handles[r], C.byref(p)), "shc_peer_open")
            self.peers.append(p.value)
        self.dst = (C.c_void_p * world)(*[p + rank * self.shard_bytes for p in self.peers])

        class _View:   # the rank's own gathered buffer as a torch tensor (__cuda_array_interface__: no copy)
            __cuda_array_interface__ = {"shape": (world * int(shard_numel),), "typestr": "<f8", "data": (self.own, False), "version": 2}
        self.out = torch.as_tensor(_View(), device=f"cuda:{device}")

    def gather(self, local, stream: int, barrier=None, pre_barrier=None):
        """local: this rank's shard (contiguous float64 device tensor); the copies are ordered after `stream` and `stream` after them.
        pre_barrier: entered first - every rank has finished with the buffers of the previous exchange."""
        from . import engine as _engine
        if pre_barrier is not None:
            pre_barrier()
        _engine._check(self.L.shc_peer_scatter(self.device, self.C.c_void_p(local.data_ptr()), self.shard_bytes, self.dst, self.world, self.C.c_void_p(stream)), "shc_peer_scatter")
        if barrier is not None:
            import torch
            torch.cuda.current_stream().synchronize() if stream == torch.cuda.current_stream().cuda_stream else torch.cuda.synchronize()
            barrier()
        return self.out

    def close(self):
        for r, p in enumerate(self.peers):
            if p:
                self.L.shc_peer_close(self.device, self.C.c_void_p(p), 0 if r == self.rank else 1)
        self.peers = []

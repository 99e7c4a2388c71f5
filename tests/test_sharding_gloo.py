"""CPU, world_size 2 over gloo: the N > 1 path of bench.py — contiguous instance shards with inputs keyed by the global
instance id, no collective while stepping, one all-gather of the joint-state buffer — reproduces the unsharded run
instance for instance.  (The per-rank compute here is the CPU oracle; on the GPU box the same helpers shard the HIP
engine, see bench.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from syropod_highlevel_controller_amd import default_hexapod_params
from syropod_highlevel_controller_amd.parallel import all_gather_joints, instance_uniform, shard_bounds, velocity_inputs

N_TOTAL, CYCLES, SEED = 48, 150, 0xC0FFEE


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_shard(lo, hi):
    from oracle_lib import OracleBatch
    p = default_hexapod_params("tripod")
    lin, ang = velocity_inputs(SEED, lo, hi)
    ob = OracleBatch(p, hi - lo)
    ob.set_velocity(lin, ang)
    ob.step(CYCLES, 1)
    q, _ = ob.joints()
    return q


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(N_TOTAL, rank, world)
    q = torch.from_numpy(_run_shard(lo, hi))
    gathered = all_gather_joints(q, world)
    dist.barrier()
    if rank == 0:
        np.save(out_path, gathered.numpy())
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (1, 7, 48, 1000):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_inputs_are_keyed_by_global_instance_id():
    lin, ang = velocity_inputs(SEED, 0, 100)
    lin2, ang2 = velocity_inputs(SEED, 37, 61)
    assert np.array_equal(lin[37:61], lin2) and np.array_equal(ang[37:61], ang2)
    assert np.linalg.norm(lin, axis=1).max() <= 1.0 and np.linalg.norm(lin, axis=1).min() >= 0.2
    u = instance_uniform(1, np.arange(200000), 3)
    assert abs(u.mean() - 0.5) < 5e-3 and 0.0 <= u.min() and u.max() < 1.0


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(tmp_path):
    out = str(tmp_path / "gathered.npy")
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    gathered = np.load(out).reshape(N_TOTAL, -1)
    full = _run_shard(0, N_TOTAL)
    assert np.array_equal(gathered, full)  # same binary, same inputs: bit-identical


def _barrier_worker(rank, world, port, out_dir):
    import sys
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    barrier = bench.HostSpinBarrier(world, rank)
    assert barrier.slots is not None
    if rank == 0:
        assert not os.path.exists(barrier.path)   # the page lives in the mappings only (rank 0 unlinks it once everybody has mapped it)
    stamps = []
    for k in range(6):
        time.sleep(0.003 * ((rank + k) % world))   # the ranks arrive at different times ...
        before = time.perf_counter()
        barrier()
        stamps.append((before, time.perf_counter()))
    np.save(os.path.join(out_dir, f"stamps{rank}.npy"), np.array(stamps))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_host_spin_barrier_of_the_resident_region(tmp_path):
    """bench.py --gpus N lines its ranks up in front of the resident-mode region on a shared /dev/shm page (a resident loop is alive
    there: no device-wide synchronisation, and a socket barrier's release skew would be a good part of an 85 us region): nobody
    leaves a meeting before the last rank has arrived, everybody leaves within a fraction of a millisecond of it."""
    world = 2
    mp.spawn(_barrier_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    s = np.stack([np.load(str(tmp_path / f"stamps{r}.npy")) for r in range(world)])   # [rank][meeting][arrived, left] (CLOCK_MONOTONIC: one clock for all)
    last_arrival = s[:, :, 0].max(axis=0)
    assert (s[:, :, 1] >= last_arrival[None, :]).all()
    assert (s[:, 1:, 1] - last_arrival[None, 1:]).max() < 20e-3   # (a busy test host: the bar is a scheduler quantum, the typical skew is microseconds)

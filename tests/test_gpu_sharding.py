"""GPU (-m gpu), world size 2: the N > 1 path of bench.py with the HIP engine on every rank.  Each rank owns a contiguous shard of
the instances (inputs keyed by the global instance id), steps it on the GPU with no collective, and the final joint-state
buffer is exchanged with the SAME all_gather_joints helper bench.py uses over RCCL - here over gloo, both ranks on device 0 (a
one-GPU box cannot host two RCCL ranks).  The gathered buffer equals the unsharded HIP run bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
from syropod_highlevel_controller_amd.parallel import all_gather_joints, shard_bounds, velocity_inputs

pytestmark = pytest.mark.gpu
SEED = 0xC0FFEE


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _params(case):
    return default_hexapod_params("tripod") if case == "hexapods" else synthetic_octopod_params("ripple", 5, 8)


def _run_hip(case, lo, hi, cycles, resident):
    from syropod_highlevel_controller_amd.engine import BatchEngine
    p = _params(case)
    lin, ang = velocity_inputs(SEED, lo, hi)
    eng = BatchEngine(p, hi - lo)
    eng.set_velocity(lin, ang)
    eng.step(cycles // 2)
    if resident:       # the second half of the run in resident mode: what bench.py's primary line does on every rank
        eng.resident_begin(ring_depth=4, max_cycles=cycles)
        eng.resident_publish(cycles - cycles // 2)
        eng.resident_end()
    else:
        eng.step(cycles - cycles // 2)
    eng.synchronize()
    q, _ = eng.joints()
    eng.close()
    return q


def _worker(rank, world, port, case, n_total, cycles, resident, out_path, peer=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(n_total, rank, world)
    q = torch.from_numpy(_run_hip(case, lo, hi, cycles, resident))
    if peer:   # the exchange as peer copies (shc_peer_*): every rank writes its shard into every rank's buffer - here two processes on device 0
        from syropod_highlevel_controller_amd.parallel import PeerAllGather
        qd = q.cuda().reshape(-1).contiguous()
        pg = PeerAllGather(qd.numel(), world, rank, 0)
        for _ in range(2):   # (twice: the buffers and streams of the first exchange are reused)
            pg.gather(qd, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            dist.barrier()
        gathered = pg.out.cpu()
        own = gathered[rank * qd.numel():(rank + 1) * qd.numel()]
        assert torch.equal(own, qd.cpu())
        dist.barrier()
        pg.close()
    else:
        gathered = all_gather_joints(q, world)
    dist.barrier()
    if rank == 0:
        np.save(out_path, gathered.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case,resident", [("hexapods", False), ("hexapods", True), ("octopods", False)])
def test_two_ranks_with_the_hip_engine_match_the_unsharded_run(tmp_path, case, resident):
    n_total, cycles, world = 2000, 160, 2     # equal shards (all_gather_into_tensor), several waves each, the last one partly filled
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), case, n_total, cycles, resident, out), nprocs=world, join=True)
    p = _params(case)
    gathered = np.load(out).reshape(n_total, p.leg_count * p.leg_dof[0])
    full = _run_hip(case, 0, n_total, cycles, resident)
    assert np.isfinite(full).all()
    assert np.array_equal(gathered, full)


@pytest.mark.timeout(600)
def test_two_ranks_exchange_their_shards_by_peer_copies(tmp_path):
    """bench.py --gather peer: the final joint buffer exchanged with shc_peer_alloc / open / scatter (IPC handles passed over the process group, one copy
    per destination on its own stream) instead of the collective library's all-gather; the gathered buffer equals the unsharded run bit for bit."""
    case, n_total, cycles, world = "octopods", 2000, 120, 2
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), case, n_total, cycles, False, out, True), nprocs=world, join=True)
    p = _params(case)
    gathered = np.load(out).reshape(n_total, p.leg_count * p.leg_dof[0])
    full = _run_hip(case, 0, n_total, cycles, False)
    assert np.array_equal(gathered, full)

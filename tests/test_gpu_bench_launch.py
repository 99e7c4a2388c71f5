"""GPU (-m gpu): bench.py as the driver starts it.  (a) `bench.py --gpus 2` started as ONE process re-executes itself under torch.distributed.run and
rank 0 prints one compact line with n_gpus 2 - two ranks on the one device of the box (--oversubscribe; RCCL refuses two ranks on one device, so the
process group is gloo and the exchange the peer-copy form, shc_peer_*: the engine, the sharding, the timed region and the line are the N > 1 path's).
(b) BASELINE.json configs[3] at its stated size on one device: bench.py --workload config4full (eight shards of 131 072 octopods, the 335.5 MB gathered
buffer)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = r.stdout.strip().splitlines()[-1]
    assert len(line.encode()) < 4096
    return json.loads(line)


@pytest.mark.timeout(1200)
def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    out = run_bench("--gpus", "2", "--oversubscribe", "--backend", "gloo", "--gather", "peer", "--workload", "config4", "--instances", "8192", "--steps", "10", "--warmup", "3",
                    "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["steps"] == 10 and out["warmup"] == 3 and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["instances_per_gpu"] == 8192 and cfg["legs"] == 8 and cfg["dof"] == 5
    assert out["value"] > 0 and cfg["scale_reference"]["value"] > 0 and cfg["gather_ms"] > 0 and cfg["value_without_gather"] >= out["value"] * 0.5
    assert 0 < cfg["weak_scaling_efficiency"]["without_gather"] and cfg["gather_bytes_per_rank"] == 8192 * 40 * 8
    assert out["parity"]["max_abs_dq"] is not None and out["parity"]["max_abs_dq"] <= 1e-6


@pytest.mark.timeout(1800)
def test_config4_at_its_stated_size_on_one_device():
    out = run_bench("--workload", "config4full", "--steps", "20", "--warmup", "5")
    cfg = out["config"]
    assert cfg["instances_per_gpu"] == 1 << 20 and cfg["legs"] == 8 and cfg["dof"] == 5
    assert out["value"] > 1e8 and cfg["gather_ms"] > 0
    assert out["parity"]["instances"] == 512 and out["parity"]["max_abs_dq"] <= 1e-6
    det = json.load(open(os.path.join(ROOT, "bench_details.json")))["config"]
    assert det["gathered_buffer_bytes"] == (1 << 20) * 40 * 8 and det["gathered_buffer_matches_getter"] and det["finite"] and det["moving_fraction"] == 1.0


@pytest.mark.timeout(1200)
def test_rccl_path_keeps_the_json_line_last():
    """One rank with the process group up (--force-dist: RCCL's all-gather in the timed region).  RCCL prints a version banner through C stdio, which a pipe
    holds back until the process exits; bench.py flushes it before its own line, so the driver's "last line of stdout" is the JSON."""
    out = run_bench("--gpus", "1", "--force-dist", "--workload", "config4", "--instances", "8192", "--steps", "10", "--warmup", "3", "--no-cpu-baseline")
    assert out["n_gpus"] == 1 and out["config"]["gather_ms"] > 0 and out["config"]["gather_form"].startswith("RCCL")

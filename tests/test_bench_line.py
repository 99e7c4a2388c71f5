"""bench.py's output contract (no GPU): the last stdout line is ONE JSON object shorter than 4 KB whatever the measurements carried, and
`--gpus N` either starts N ranks itself or refuses a launcher whose world size disagrees (VERDICT r5 "Next round" 1 and 2)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PROSE = "x" * 700    # every free-text field of the full record may be this long: the compact line must not care


def canned_roofline(frac=0.0632):
    return {"bound": "valu-issue", "achieved": 505.4321987654321, "peak": 8000.0, "unit": "GB/s", "frac": frac, "traffic": 1203456, "kernel": "shc_resident2_kernel " + PROSE,
            "kernel_ms": 0.0025285123456789, "kernel_ms_is": PROSE, "algorithmic_bytes_per_launch": 1277952, "algorithmic_bytes_are": PROSE, "bound_note": PROSE,
            "bound_is": PROSE, "valu_issue_frac": 0.32123456789, "valu_issue_share_per_wave": 0.43, "hbm_frac": frac, "state_streaming_equivalent_frac": 0.6,
            "one_launch_per_cycle": {"bound": "hbm", "frac": 0.2, "bound_is": PROSE}}


def canned_parity():
    return {"max_abs_dq": 1.8123456789e-14, "max_abs_dq_all_instances": 2.5e-14, "unit": "rad", "instances": 64, "cycles": 25, "well_posed_fraction": 1.0, "tolerance": 1e-6,
            "against": PROSE, "window": PROSE, "bins": [{"bin": [6, 3, "tripod"], "max_abs_dq": 1e-12, "note": PROSE}] * 6}


def canned_full(n_also=12, n_gpus=1):
    also = []
    for k in range(n_also):
        also.append({"workload": f"BASELINE.json config{k}: " + PROSE, "short": f"workload number {k} with a name that is far too long for one row", "form": "step_k K=16",
                     "value": 1.8412345678e9 + k, "ms_per_step": 0.0355123456789, "unit": "control-cycles/s", "roofline": canned_roofline(0.61234567), "parity": canned_parity(),
                     "launch_value": 1.3312345678e9, "launch_frac": 0.5912345678, "fused_K_with_per_cycle_inputs": {"roofline": canned_roofline(), "parity": canned_parity(), "note": PROSE},
                     "launch": {"roofline": canned_roofline(), "parity": canned_parity()}})
    also.append({"workload": "config9", "short": "a workload that failed", "error": "E" * 500})
    cfg = {"workload": "BASELINE.json config2: " + PROSE, "short": "BASELINE.json config2: 4096/GPU 6x3 +joint torques", "mode": PROSE, "mode_short": "resident loop: 1 step = 1 doorbell tick = 1 control cycle",
           "instances_per_gpu": 4096, "legs": 6, "dof": 3, "seed": 12648430, "gather": PROSE, "gather_short": "none (N = 1)", "gather_form": None, "one_launch_per_cycle_value": 4.48e8,
           "velocities_posted_every_cycle_value": 1.31e9, "moving_fraction": 1.0, "finite": True, "fused_K_with_per_cycle_inputs": None}
    if n_gpus > 1:
        cfg.update({"scale_reference": {"workload": PROSE, "n_gpus": 1, "value": 1.35e9, "ms_per_step": 0.097, "value_is": PROSE, "per_rank_values": [1.35e9] * n_gpus},
                    "weak_scaling_efficiency": {"with_gather": 0.91234567, "without_gather": 0.9876543, "is": PROSE}, "value_without_gather": 1.07e10, "gather_ms": 0.81234567,
                    "gather_bytes_per_rank": 41943040, "gather_form": PROSE, "gather_form_short": "RCCL all_gather"})
    return {"metric": "control-cycles/sec (all legs IK-solved)", "value": 1.2045678912345e9, "unit": "control-cycles/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5,
            "ms_per_step": 0.0034012345678, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": cfg,
            "roofline": canned_roofline(), "parity": canned_parity(),
            "cpu_baseline": {"value": 1.05e6, "unit": "control-cycles/s", "cores": 256, "kind": "port", "sample": PROSE, "sample_short": "4096 robots x 2000 cycles, 256 threads", "single_thread_value": 122410.09},
            "also": also}


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "parity",
            "cpu_baseline", "also")


@pytest.mark.parametrize("n_gpus", [1, 8])
def test_compact_line_is_short_and_complete(n_gpus):
    full = canned_full(12, n_gpus)
    line = bench.compact_line(full)
    assert len(line.encode()) < 4096 and "\n" not in line
    out = json.loads(line)
    assert json.loads(json.dumps(out)) == out
    for k in REQUIRED:
        assert k in out, k
    assert out["value"] == pytest.approx(full["value"], rel=1e-6) and out["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    for k in ("workload", "mode", "instances_per_gpu", "legs", "dof", "seed", "gather"):
        assert k in out["config"], k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch", "valu_issue_frac"):
        assert k in out["roofline"], k
    assert out["roofline"]["frac"] == pytest.approx(out["roofline"]["achieved"] / out["roofline"]["peak"], rel=1e-3)
    for k in ("max_abs_dq", "instances", "cycles", "well_posed_fraction", "tolerance"):
        assert k in out["parity"], k
    for k in ("value", "cores", "kind", "single_thread_value", "sample", "unit"):
        assert k in out["cpu_baseline"], k
    rows = out["also"]
    assert len(rows) + out.get("also_dropped", 0) == 13 and len(rows) >= 9       # the default run has ten secondary workloads
    for r in rows:
        if "error" in r:
            continue
        assert len(r["workload"]) <= 40
        for k in ("value", "ms_per_step", "frac", "traffic_ratio", "valu_issue_frac", "max_abs_dq"):
            assert k in r, k
    if n_gpus > 1:
        for k in ("scale_reference", "weak_scaling_efficiency", "value_without_gather", "gather_ms", "gather_form"):
            assert out["config"][k] is not None, k


def test_default_run_shape_keeps_every_row():
    """The default run's ten secondary rows (realistic names) all fit next to the headline."""
    full = canned_full(0)
    full["also"] = [dict(canned_full(1)["also"][0], short=s) for s in list(bench.SHORT.values()) + ["config2 4096 6x3 tripod no torques", "config4 share 131072 8x5 ripple +torques"]]
    out = json.loads(bench.compact_line(full))
    assert "also_dropped" not in out and len(out["also"]) == len(bench.SHORT) + 2


def test_round5_record_would_have_fit():
    """The 30 KB line of round 5 (the one the driver could not parse), fed through the compact form."""
    path = os.path.join(ROOT, "profiles", "bench", "r05_bench_driver_like.json")
    full = json.loads(open(path).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    out = json.loads(line)
    assert len(line.encode()) < 4096 and out["value"] == pytest.approx(full["value"], rel=1e-6) and len(out["also"]) == len(full["config"]["also"])


def test_emit_prints_the_compact_line_last(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(canned_full(3))
    cap = capsys.readouterr()
    lines = cap.out.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096 and json.loads(lines[0])["details"] == bench.DETAILS_FILE
    assert json.loads(open(tmp_path / bench.DETAILS_FILE).read())["also"][0]["roofline"]["bound_note"] == PROSE    # the prose lives in the side file


def run_bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e, timeout=120)


def test_gpus_n_starts_n_ranks_by_itself():
    r = run_bench("--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-launch")
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout)
    assert plan["action"] == "exec"
    argv = plan["argv"]
    assert argv[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in argv and "--nnodes=1" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    tail = argv[argv.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_launcher_world_size_must_match_gpus():
    r = run_bench("--gpus", "8", "--dry-launch", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    r = run_bench("--gpus", "2", "--dry-launch", env={"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"})
    assert r.returncode == 0 and json.loads(r.stdout) == {"action": "run", "world": 2, "rank": 1, "local_rank": 1}
    r = run_bench("--dry-launch")
    assert json.loads(r.stdout) == {"action": "run", "world": 1, "rank": 0, "local_rank": 0}

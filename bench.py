#!/usr/bin/env python
"""bench.py — control-cycles/sec of the batched leg-control hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched by
``python -m torch.distributed.run --nproc-per-node N ...`` (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

* metric / unit: BASELINE.json's ``control-cycles/sec (all legs IK-solved)``.
* workload (N = 1): BASELINE.json configs[1] — 4 096 default.yaml hexapods (6 legs x 3 DOF), tripod gait, the full
  per-cycle path (velocity limiting + walk FSM + Bezier tip trajectory + body pose + per-leg DLS IK/FK), synthetic
  seeded velocity commands, every instance MOVING and de-phased before the timed region.  Measured joint torques are
  supplied (as jointStatesCallback does on a real node), so Leg::calculateTipForce is evaluated every cycle as in the
  reference (model.cpp:938); ``--no-joint-efforts`` is the variant without them (the estimate is then identically zero and the
  engine runs the kernels without it), which the default run reports under ``config.also``.
* a "step" = one control cycle of every instance (``--cycles-per-step`` 1), inputs resident in HBM (no host traffic inside
  the timed region).  Batches that fit the chip once (config 2) run in RESIDENT mode (``--mode auto``): one launch of the
  cycle kernel stays on the device, state in registers / LDS from cycle to cycle, and a step is one tick of the device-side
  doorbell (shc_engine_resident_publish(1)) - the kernel takes that cycle's inputs from the input rings (posted inputs may
  differ every cycle: tests/test_gpu_resident.py) and writes its q / qd to the output ring.  ``--mode launch`` (and every batch
  that does not fit) is one launch of the fused cycle kernel per step; the default run reports both.
* N > 1: weak scaling — every rank owns ``--instances`` robots of its own (instance ranges are contiguous per rank),
  no data-path collective while stepping; the final joint-state buffer is all-gathered over RCCL inside the timed
  region (every ``--gather-every`` steps if given).
* ``roofline``: HBM roofline of the kernel that produced ``value``.  One launch per cycle: algorithmic bytes per launch
  (3 008 B per hexapod cycle, state streamed in and out) / mean kernel duration, HIP events on the launch
  stream.  Resident mode: the state never leaves the chip, so per SURVEY.md §8d ("bytes/cycle fall toward the mandatory
  output - report K and count accordingly") the algorithmic bytes of a cycle are its inputs + the desired joint state it
  publishes (24 + 2 x legs x dof x 8 B per robot), K = cycles of the launch; duration = HIP events around the launch / K.
* ``cpu_baseline``: the CPU oracle (a scalar restatement of the reference loop, "port") timed on this box's host
  cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

ALG_BYTES_PER_CYCLE = {("hexapod", 2): 3008, ("hexapod", 3): 3560, ("octopod", 4): 4496,
                       # SURVEY.md section 8(f) rank 4 workloads, counted the same way: config 2's state + per leg the measured tip force (24 B read) and
                       # Leg::step_plane_pose_ (position + defined flag, 32 B read) of rough terrain mode ...
                       ("hexapod", "rough"): 3008 + 6 * (24 + 32),
                       # ... config 4's state + per leg the two tip directions the engine keeps for LegStepper's origin / current tip rotations
                       # (6 doubles read + written; SURVEY counts 3 full quaternions = 1 536 B - the smaller figure is the honest one here)
                       ("octopod", "gravity"): 4496 + 8 * 2 * 48,
                       # ... and BASELINE config 3's feature set on those octopods: + per leg the admittance state (2 doubles read + written), the admittance
                       # delta it publishes (32 B written) and the measured tip force (24 B read), per robot the IMU sample (56 B read) and the PID state (6 doubles r + w)
                       ("octopod", "gravity3"): 4496 + 8 * 2 * 48 + 8 * (32 + 32 + 24) + 56 + 96}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy peak)


def config3_forces(rng, n, legs):
    """measured tip force z ~ U(0, 20) N, x, y ~ N(0, 1)."""
    return np.stack([rng.normal(0, 1, (n, legs)), rng.normal(0, 1, (n, legs)), rng.uniform(0, 20, (n, legs))], axis=2)


def make_workload(name, n, seed, rank=0, joint_efforts=False):
    from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
    from syropod_highlevel_controller_amd.parallel import velocity_inputs
    # instance ranges are contiguous per rank; inputs are keyed by the GLOBAL instance id (parallel.py)
    lin, ang = velocity_inputs(seed, rank * n, (rank + 1) * n)  # uniform in the unit disc, |v| >= 0.2: every robot walks
    rng = np.random.default_rng(seed + 7919 * rank)
    extra = {}
    if name == "config2":
        p = default_hexapod_params("tripod")
        key, desc = ("hexapod", 2), "hexapods (6x3 DOF, default.yaml), tripod gait, IK + Bezier tip trajectory"
    elif name == "config3":
        p = default_hexapod_params("wave")
        p.admittance_control, p.imu_posing = 1, 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
        key = ("hexapod", 3)
        desc = ("hexapods, wave gait + admittance (tip force z ~ U(0, 20) N, x, y ~ N(0, 1), resampled every 10 cycles from "
                "device-resident sets) + IMU pose compensation (PID 0.2 / 0.02 / 0.01)")
        from scipy.spatial.transform import Rotation as R
        e = np.stack([rng.uniform(-0.15, 0.15, n), rng.uniform(-0.15, 0.15, n), rng.uniform(-np.pi, np.pi, n)], axis=1)
        q = R.from_euler("xyz", e).as_quat()
        extra["imu_q"] = np.stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]], axis=1)
        extra["gyro"] = rng.normal(0, 0.05, size=(n, 3))
        frng = np.random.default_rng(0xADD1 + rank)
        extra["force"] = config3_forces(frng, n, 6)
        extra["force_sets"] = [config3_forces(frng, n, 6) for _ in range(4)]  # rotated through every 10 cycles
    elif name == "config4":
        p = synthetic_octopod_params("ripple", 5, 8)
        key, desc = ("octopod", 4), "synthetic octopods (8x5 DOF), ripple gait, IK + Bezier tip trajectory"
    elif name == "rough":   # the rough-terrain path
        p = default_hexapod_params("tripod")
        p.rough_terrain_mode, p.step_depth = 1, 0.012
        key = ("hexapod", "rough")
        desc = ("hexapods, tripod gait, rough_terrain_mode: tip-state messages (contact forces that come and go, resampled every 10 cycles from "
                "device-resident sets) -> touchdown detection, step-plane swing targets, terrain-following default tips, walk-plane refit")

        def contact_forces():
            f = rng.normal(0, 0.25, (n, 6, 3))
            f[..., 2] += rng.choice([0.0, 0.05, 0.6, 1.5], size=(n, 6), p=[0.3, 0.2, 0.2, 0.3])
            return f
        extra["force"] = contact_forces()
        extra["force_sets"] = [contact_forces() for _ in range(4)]
    elif name == "gravity":  # ... and gravity-aligned tips: tip rotations + the rotation-constrained IK on 5-DOF legs
        p = synthetic_octopod_params("ripple", 5, 8)
        p.gravity_aligned_tips = 1
        key, desc = ("octopod", "gravity"), "synthetic octopods (8x5 DOF), ripple gait, gravity_aligned_tips: LegStepper tip rotations + rotation-constrained Leg::applyIK"
    elif name == "gravity3":  # the north-star feature set TOGETHER with the tip rotations: admittance + IMU pose compensation + gravity-aligned tips on 5-DOF legs
        p = synthetic_octopod_params("ripple", 5, 8)
        p.gravity_aligned_tips, p.admittance_control, p.imu_posing = 1, 1, 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
        key = ("octopod", "gravity3")
        desc = ("synthetic octopods (8x5 DOF), ripple gait, gravity_aligned_tips + admittance (tip force z ~ U(0, 20) N, resampled every 10 cycles) + IMU pose "
                "compensation: src/model.cpp:880-903 + src/pose_controller.cpp:1191-1236 + src/admittance_controller.cpp:22-63 in one cycle (two launches per cycle)")
        from scipy.spatial.transform import Rotation as R
        e = np.stack([rng.uniform(-0.15, 0.15, n), rng.uniform(-0.15, 0.15, n), rng.uniform(-np.pi, np.pi, n)], axis=1)
        q = R.from_euler("xyz", e).as_quat()
        extra["imu_q"] = np.stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]], axis=1)
        extra["gyro"] = rng.normal(0, 0.05, size=(n, 3))
        frng = np.random.default_rng(0xADD2 + rank)
        extra["force"] = config3_forces(frng, n, 8)
        extra["force_sets"] = [config3_forces(frng, n, 8) for _ in range(4)]
    else:
        raise SystemExit(f"unknown workload {name}")
    effort = rng.normal(0, 0.5, size=(n, p.leg_count * p.leg_dof[0]))
    if joint_efforts:  # measured joint torques (jointStatesCallback): Leg::calculateTipForce has something to filter
        extra["effort"] = effort
        desc += " + joint-effort input (tip-force estimate evaluated every cycle)"
    return p, lin, ang, extra, key, desc


def apply_inputs(obj, lin, ang, extra):
    obj.set_velocity(lin, ang)
    if "imu_q" in extra:
        obj.set_imu(extra["imu_q"], extra["gyro"])
    if "force" in extra:
        obj.set_tip_force(extra["force"])
    if "effort" in extra:
        obj.set_joint_effort(extra["effort"])


def cpu_baseline(p, lin, ang, extra, target_seconds=12.0):
    """Oracle ("port" of the reference loop) on the host cores, bounded sample of the same workload."""
    from oracle_lib import OracleBatch
    cores = os.cpu_count() or 1
    n_s = min(len(ang), 64 * cores)
    sub = {k: v[:n_s] for k, v in extra.items() if k != "force_sets"}
    ob = OracleBatch(p, n_s)
    apply_inputs(ob, lin[:n_s], ang[:n_s], sub)
    t = ob.step(20, cores)  # calibration + warm-up (walk start)
    rate = n_s * 20 / max(t, 1e-9)
    cycles = int(max(20, min(2000, target_seconds * rate / n_s)))
    t = ob.step(cycles, cores)
    multi = n_s * cycles / t
    n1 = min(n_s, 64)
    ob1 = OracleBatch(p, n1)
    apply_inputs(ob1, lin[:n1], ang[:n1], {k: v[:n1] for k, v in sub.items()})
    ob1.step(20, 1)
    c1 = int(max(20, min(2000, 3.0 * (rate / cores) / n1)))
    t1 = ob1.step(c1, 1)
    return {"value": multi, "unit": "control-cycles/s", "cores": cores, "kind": "port", "sample_short": f"{n_s} robots x {cycles} cycles, {cores} threads",
            "sample": f"{n_s} instances x {cycles} cycles on {cores} threads (pthreads over instances); "
                      f"single thread: {n1 * c1 / t1:.0f} control-cycles/s ({n1} instances x {c1} cycles)",
            "single_thread_value": n1 * c1 / t1}


PARITY_INSTANCES = 256   # robots of the timed batch replayed on the oracle (and its perturbed twin) over the timed window


class ParityWindow:
    """SURVEY.md section 8(d) "Evidence": max |dq| of the HIP engine against the CPU oracle over the TIMED window.  The first PARITY_INSTANCES
    instances' complete controller state is taken from the engine right before the window (shc_engine_get_state), the oracle is started from
    it, given the same inputs (and the same input changes at the same cycles), stepped over the same number of cycles, and its joints are
    compared with the engine's joints of the window's last cycle.  A twin oracle whose inputs are perturbed by 1e-13 tells which REFERENCE
    trajectories are well-posed over the window (tests/test_gpu_parity.py header: the reference's one-step DLS with its normalised
    joint-limit gradient is an expanding map for limit-pinned / redundant legs); the figure over those and over all instances are both reported.
    Checker only: nothing here is inside a timed region, and the product never calls the oracle."""

    def __init__(self, eng, p, lin, ang, extra, force_now=None):
        self.p, self.m = p, min(PARITY_INSTANCES, eng.n)
        m = self.m
        self.st0 = eng.get_state(0, m)
        self.inputs = {"lin": lin[:m].copy(), "ang": ang[:m].copy()}
        for k in ("imu_q", "gyro", "effort"):
            if k in extra:
                self.inputs[k] = extra[k][:m].copy()
        if force_now is not None:
            self.inputs["force"] = force_now[:m].copy()
        self.events = []      # (cycle offset inside the window, tip-force set) in order

    def force_changed(self, offset, force):
        self.events.append((int(offset), force[:self.m].copy()))

    def evaluate(self, n_cycles, q_gpu, threads=None):
        from oracle_lib import OracleBatch
        threads = threads or min(os.cpu_count() or 1, 32)
        m, eps = self.m, 1e-13
        obs = []
        for twin in (False, True):
            ob = OracleBatch(self.p, m)
            ob.set_state(self.st0)
            k = 1.0 + (eps if twin else 0.0)
            ob.set_velocity(self.inputs["lin"] * k, self.inputs["ang"])
            if "imu_q" in self.inputs:
                ob.set_imu(self.inputs["imu_q"], self.inputs["gyro"])
            if "effort" in self.inputs:
                ob.set_joint_effort(self.inputs["effort"])
            if "force" in self.inputs:
                ob.set_tip_force(self.inputs["force"] * k)
            done = 0
            for off, force in self.events + [(n_cycles, None)]:
                off = min(off, n_cycles)
                if off > done:
                    ob.step(off - done, threads)
                    done = off
                if force is not None and off < n_cycles:
                    ob.set_tip_force(force * k)
            obs.append(ob.joints()[0])
        q_cpu, q_twin = obs
        q_gpu = np.asarray(q_gpu)[:m]
        if q_cpu.shape != q_gpu.shape:   # a robot whose legs differ in DOF: the oracle packs each leg's own joints, the engine's rows are [legs][longest DOF]
            L, D = self.p.leg_count, q_gpu.shape[1] // self.p.leg_count

            def padded(a):
                out, at = np.zeros((m, L, D)), 0
                for l in range(L):
                    d = self.p.leg_dof[l]
                    out[:, l, :d] = a[:, at:at + d]
                    at += d
                return out.reshape(m, L * D)
            q_cpu, q_twin = padded(q_cpu), padded(q_twin)
        dq = np.abs(q_gpu - q_cpu).max(axis=1)
        well = np.abs(q_cpu - q_twin).max(axis=1) <= 1e-9
        return {"max_abs_dq": float(dq[well].max()) if well.any() else None, "max_abs_dq_all_instances": float(dq.max()), "unit": "rad",
                "instances": int(m), "cycles": int(n_cycles), "well_posed_fraction": float(well.mean()), "tolerance": 1e-6,
                "against": "CPU oracle (oracle/shc_oracle.c) started from the engine's state record at the start of the timed window, same inputs, "
                           "free-running over the window; max_abs_dq over the instances whose reference trajectory is well-posed "
                           "(a twin oracle with inputs x (1 + 1e-13) stays within 1e-9 rad), max_abs_dq_all_instances over all of them"}


def fused_k_probe(eng, p, n, lin, ang, extra, key, stream, K=16, reps=8, want_parity=True):
    """Secondary figure for batches that do not fit the chip once: K control cycles per launch, EACH WITH ITS OWN INPUTS (shc_engine_step_k: the
    node's loop delivers new callbacks' inputs in every iteration, src/main.cpp:106-131) - cycle k reads row k of K-deep device arrays (a new
    velocity command for every robot, and whatever else the workload feeds: IMU sample, tip force, joint torques) and writes its q / qd to slot
    k of the output ring; the controller state is loaded once, stays in registers / LDS for the K cycles and is stored once.
    Bytes per SURVEY.md section 8(d) ("if K cycles are fused per launch ... report K and count accordingly"): the state once per launch + K x
    (the cycle's inputs + the published q, qd).  Parity: the oracle replays the same K input rows from the engine's state record."""
    import torch
    L, D = p.leg_count, p.leg_dof[0]
    ks = np.arange(K)
    # a new command in every row, moving as a command does between two 20 ms cycles (a few 1e-4 of its size: inside the acceleration limits, so the
    # robots follow it - a 15 % swing per cycle, tried first, is 3 x the top speed per second and keeps every robot in the acceleration-limited branch)
    rows = {"lin": lin[None] * (1.0 + 5e-4 * np.cos(0.4 * ks))[:, None, None], "ang": ang[None] * (1.0 + 5e-4 * np.sin(0.3 * ks))[:, None]}
    in_bytes = 24
    if "imu_q" in extra:
        rows["imu_q"] = np.repeat(extra["imu_q"][None], K, axis=0)
        rows["gyro"] = extra["gyro"][None] * (1.0 + 0.05 * ks)[:, None, None]
        in_bytes += 56
    if "force_sets" in extra:
        rows["force"] = np.stack([extra["force_sets"][k % len(extra["force_sets"])] for k in range(K)])
        in_bytes += L * 24
    if "effort" in extra:
        rows["effort"] = extra["effort"][None] * (1.0 + 0.02 * ks)[:, None, None]
        in_bytes += L * D * 8
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in rows.items()}
    torch.cuda.synchronize()
    ptr = lambda k: dev[k].data_ptr() if k in dev else None

    def launch():
        eng.step_k(K, velocity=(ptr("lin"), ptr("ang")), imu=(ptr("imu_q"), ptr("gyro")) if "imu_q" in dev else None, tip_force=ptr("force"), joint_effort=ptr("effort"))

    parity = None
    if want_parity:
        from oracle_lib import OracleBatch
        m, eps, threads = min(PARITY_INSTANCES, n), 1e-13, min(os.cpu_count() or 1, 32)
        eng.synchronize()
        st0 = eng.get_state(0, m)
        launch()
        q_gpu = eng.step_k_joints(K - 1)[0][:m]
        obs = []
        for twin in (False, True):
            ob = OracleBatch(p, m)
            ob.set_state(st0)
            kk = 1.0 + (eps if twin else 0.0)
            for k in range(K):
                ob.set_velocity(rows["lin"][k][:m] * kk, rows["ang"][k][:m])
                if "imu_q" in rows:
                    ob.set_imu(rows["imu_q"][k][:m], rows["gyro"][k][:m])
                if "force" in rows:
                    ob.set_tip_force(rows["force"][k][:m] * kk)
                if "effort" in rows:
                    ob.set_joint_effort(rows["effort"][k][:m])
                ob.step(1, threads)
            obs.append(ob.joints()[0])
        dq = np.abs(q_gpu - obs[0]).max(axis=1)
        well = np.abs(obs[0] - obs[1]).max(axis=1) <= 1e-9
        parity = {"max_abs_dq": float(dq[well].max()) if well.any() else None, "max_abs_dq_all_instances": float(dq.max()), "unit": "rad", "instances": int(m),
                  "cycles": K, "well_posed_fraction": float(well.mean()), "tolerance": 1e-6,
                  "against": "CPU oracle from the engine's state record before one launch, the same K input rows set before each of its cycles; the engine's joints "
                             "are slot K - 1 of the launch's output ring"}
    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(reps):
        launch()
    eng.join()
    e1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    launch_ms = e0.elapsed_time(e1) / reps
    out_bytes = 2 * L * D * 8
    alg = n * (ALG_BYTES_PER_CYCLE[key] + K * (in_bytes + out_bytes))
    ach = alg / (launch_ms * 1e-3) / 1e9
    return {"value": n * K * reps / wall, "K": K, "launches": reps, "ms_per_launch": launch_ms, "ms_per_cycle": launch_ms / K,
            "inputs_per_cycle": sorted(rows.keys()), "parity": parity,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "shc_resident_kernel, batch form (shc_engine_step_k)", "kernel_ms": launch_ms, "algorithmic_bytes_per_launch": alg,
                         "algorithmic_bytes_are": f"state once per launch ({ALG_BYTES_PER_CYCLE[key]} B per robot) + K x (inputs {in_bytes} B + published q, qd {out_bytes} B)"},
            "note": "K cycles per launch with a new input set in every cycle, read from K-deep device arrays; q / qd of every cycle in a K-deep output ring"}


def measured_traffic(workload, n, cps):
    """HBM bytes per launch from the rocprofv3 PMC passes of THIS kernel build (profiles/traffic.json, written by
    scripts/summarize_prof.py); null when the committed figure belongs to different kernel sources."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        from syropod_highlevel_controller_amd.engine import _source_hash
        t = json.load(open(tpath))
        if t.get("_kernel_source_hash") != _source_hash():
            return None
        return t.get(f"{workload}:{n}:{cps}")
    except Exception:
        return None


def measured_valu(workload, n, cps):
    """Issue-side roofline from the same PMC passes (scripts/summarize_prof.py): the share of the chip's VALU issue slots the kernel used and the
    share of a wavefront's lifetime its vector ALU was issuing; null when the committed figures belong to different kernel sources."""
    v = measured_traffic(workload, f"{n}", f"{cps}#valu")
    return v if isinstance(v, dict) else None


def with_issue_side(roofline, valu, seconds):
    """SURVEY.md section 8(d) "report VALUBusy alongside GB/s": adds the VALU-issue fraction and names the bound by the larger of the two fractions.
    valu_issue_frac = the 4-clock VALU issue slots the step's kernels used (rocprofv3 SQ_ACTIVE_INST_VALU, summed over the launches of one step) / the slots the
    chip's 1 024 SIMDs have in `seconds` at 2.4 GHz - the same duration `achieved` is computed with."""
    frac = (valu["valu_active_quads_per_step"] / (1024.0 * seconds * 2.4e9 / 4.0)) if (valu and seconds) else None
    roofline["valu_issue_frac"] = frac
    roofline["valu_issue_share_per_wave"] = valu["valu_issue_share_per_wave"] if valu else None
    roofline["hbm_frac"] = roofline["frac"]
    if frac is not None and frac > roofline["frac"]:
        roofline["bound"] = "valu-issue"
    roofline["bound_is"] = ("the larger of hbm_frac (algorithmic bytes / duration / 8 TB/s) and valu_issue_frac (rocprofv3 SQ_ACTIVE_INST_VALU over the chip's 1 024 SIMD issue "
                            "slots, profiles/traffic.json, same kernel sources); `frac` stays the HBM fraction the contract defines")
    return roofline


def resident_bytes_per_cycle(p):
    """SURVEY.md section 8(d), K cycles fused with the state on the chip: per robot and cycle the velocity input (24 B) and the
    desired joint positions + velocities the cycle publishes."""
    return 24 + 2 * p.leg_count * p.leg_dof[0] * 8


class HostSpinBarrier:
    """Ranks of one node meet on a page of /dev/shm: every rank publishes its epoch and spins until all have (sub-microsecond skew - a
    20-step region is 85 us long, a socket barrier's release skew would be a good part of it).  Built around two RCCL barriers, so
    only while no resident loop is alive; used while one is.  Gives up after 10 s (the ranks then enter the region as they come)."""

    def __init__(self, world, rank):
        import torch.distributed as dist
        self.world, self.rank, self.epoch = world, rank, 0
        self.path = f"/dev/shm/shc_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getuid()}"
        self.slots = None
        try:
            if rank == 0:
                np.memmap(self.path, dtype=np.int64, mode="w+", shape=(world,)).flush()
            dist.barrier()
            self.slots = np.asarray(np.memmap(self.path, dtype=np.int64, mode="r+", shape=(world,)))   # (a plain ndarray view: memmap's own operators are slow)
            dist.barrier()
            if rank == 0:
                os.unlink(self.path)    # the mappings keep the page alive; nothing is left behind
        except OSError:
            self.slots = None

    def __call__(self):
        if self.slots is None:
            return
        self.epoch += 1
        self.slots[self.rank] = self.epoch
        slots, epoch, t_end = self.slots, self.epoch, time.perf_counter() + 10.0
        while slots.min() < epoch:
            if time.perf_counter() > t_end:
                self.slots = None
                return


class Watchdog:
    """N > 1 regions must never hang the driver: if a region is still running after `seconds`, print ONE JSON line with an "error" (what
    was running, on which rank) and leave the process.  A daemon timer thread: the main thread may be blocked inside a HIP / RCCL call."""

    def __init__(self, seconds, what, rank=0, meta=None):
        import threading
        self.t = threading.Timer(seconds, self.fire)
        self.t.daemon = True
        self.what, self.rank, self.seconds, self.meta = what, rank, seconds, meta or {}

    def fire(self):
        line = {"metric": "control-cycles/sec (all legs IK-solved)", "value": None, "unit": "control-cycles/s", "error":
                f"watchdog: '{self.what}' on rank {self.rank} did not finish within {self.seconds:.0f} s; the process was stopped instead of hanging"}
        line.update(self.meta)
        print(json.dumps(line), flush=True)
        os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def time_resident(eng, n, steps, warmup, stream, depth=16, final_gather=None, host_barrier=None, parity=None, gather_after_end=None):
    """Timed region of resident mode: `steps` ticks of the doorbell, one control cycle each, bracketed by synchronisation.
    N > 1 (final_gather): the region ends with the all-gather of the LAST cycle's joints.  It is queued on the engine's stream behind a
    device-side wait for that cycle (shc_engine_resident_get_joint_state_async) before the first tick, so only its execution - not
    its launch - follows the last cycle; the closing bracket is the synchronisation of that stream.
    gather_after_end (N > 1, the default there): the loop is ENDED first (shc_engine_resident_end: state written back, kernel gone), then the
    joints are gathered - no collective kernel ever has to be scheduled next to a persistent loop; the region contains the ticks, the end
    of the loop and the gather.
    parity: a ParityWindow factory; the window is warm-up + timed ticks (the state record is taken before the loop starts).
    Returns (elapsed seconds for `steps` cycles, seconds per cycle of one long launch from HIP events on the launch stream, parity dict)."""
    import ctypes
    import torch
    pw = parity() if parity else None
    eng.resident_begin(ring_depth=depth, max_cycles=warmup + steps + 8)
    # the tick as the node's loop would issue it: one C call (bound once - attribute lookups and argument conversion of the Python wrapper
    # cost as much as a 3 us cycle); return codes are checked after the region
    tick, handle, one = eng.L.shc_engine_resident_publish, eng.h, ctypes.c_int64(1)
    if host_barrier:
        host_barrier()               # (a first meeting outside the region: the ranks start their warm-up together, the barrier's own code is warm)
    rcs = [tick(handle, one) for _ in range(max(warmup, 1))]     # the warm-up steps, tick by tick like the timed ones
    eng.resident_wait(max(warmup, 1))
    if final_gather:
        # launched ahead like a captured graph: the device holds the read (and the collective behind it) until the last cycle of the
        # region has run - the host calls are outside the region, the gather's execution and the wait for it are inside
        final_gather(max(warmup, 1) + steps - 1)
    if host_barrier:
        host_barrier()               # ranks enter the region together (a host-side barrier: the loop kernel is alive, no device-wide sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        rcs.append(tick(handle, one))   # one tick = one cycle; the host does not wait for it
    ended = False
    if final_gather:
        stream.synchronize()         # the gathered buffer is complete: every rank's last cycle has run
    elif gather_after_end:
        eng.resident_wait(max(warmup, 1) + steps, 60000)
        eng.resident_end()           # the loop leaves the chip, the state is back in the engine's planes
        ended = True
        gather_after_end()           # SoA planes -> [n][legs][dof] + all_gather_into_tensor on the engine's stream
        stream.synchronize()
    else:
        eng.resident_wait(max(warmup, 1) + steps, 60000)
    elapsed = time.perf_counter() - t0
    assert not any(rcs), "shc_engine_resident_publish failed"
    par = None
    if ended:
        q_last = eng.joints()[0] if pw else None   # (the state the loop wrote back: q of the window's last cycle)
    else:
        if pw:
            q_last = eng.resident_joints(max(warmup, 1) + steps - 1)[0]   # the output ring still holds the window's last cycle
        eng.resident_end()
    if pw:
        par = pw.evaluate(max(warmup, 1) + steps, q_last)
        par["window"] = f"{max(warmup, 1)} warm-up + {steps} timed doorbell ticks of the resident loop; the engine's joints are the output ring's slot of the last cycle"
    # kernel time per cycle: one launch of m cycles released at once, HIP events on the launch stream around it
    m = 4000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    eng.resident_begin(ring_depth=depth, max_cycles=m)
    eng.resident_publish(m)
    eng.resident_end()               # stops after the m published cycles, waits for the kernel
    e1.record(stream)
    torch.cuda.synchronize()
    return elapsed, e0.elapsed_time(e1) * 1e-3 / m, par


def run_workload(name, n, steps, warmup, cps, seed, dist_ctx=None, gather_every=0, fused_probe=True, want_cpu_baseline=False, joint_efforts=False,
                 mode="auto", gather_under_loop=False, want_parity=True, gather_form="rccl"):
    """One workload on this rank's GPU: prepare (untimed), time `steps` steps, measure the kernel with HIP events.
    dist_ctx = (world, rank, local_rank) when the RCCL path is active."""
    import torch
    import torch.distributed as dist
    from syropod_highlevel_controller_amd.engine import BatchEngine
    from syropod_highlevel_controller_amd.parallel import all_gather_joints

    world, rank, local_rank = dist_ctx or (1, 0, torch.cuda.current_device())
    use_dist = dist_ctx is not None
    p, lin, ang, extra, key, desc = make_workload(name, n, seed, rank, joint_efforts)
    stream = torch.cuda.current_stream()
    eng = BatchEngine(p, n, device=local_rank, stream=stream.cuda_stream)
    n_waves = -(-n // (64 // p.leg_count))   # from 4 096 wavefronts on shc_engine_step launches the batch as two halves on two streams
    apply_inputs(eng, lin * 0.0, ang * 0.0, extra)
    # config 3: the measured tip forces are resampled every 10 cycles from sets resident in HBM
    force_sets = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in extra.get("force_sets", [])]
    state = {"cycle": 0, "force": extra.get("force"), "window": None, "window_start": 0}

    def step_once():
        if force_sets and state["cycle"] % 10 < cps and state["cycle"] > 0:
            k = (state["cycle"] // 10) % len(force_sets)
            eng.L.shc_engine_set_tip_force(eng.h, force_sets[k].data_ptr(), 1)  # device pointer: a scatter kernel on the engine's stream
            state["force"] = extra["force_sets"][k]
            if state["window"] is not None:   # (host-side note for the parity replay; no device work)
                state["window"].force_changed(state["cycle"] - state["window_start"], state["force"])
        eng.step(cps)
        state["cycle"] += cps

    def open_parity_window():   # the engine's state record + the inputs in force, right before a timed window (untimed)
        eng.synchronize()
        state["window"] = ParityWindow(eng, p, lin, ang, extra, state["force"])
        state["window_start"] = state["cycle"]
        return state["window"]

    # ---- untimed preparation: de-phase the instances (instance i receives its command i mod period cycles late),
    #      then walk until every instance is MOVING.
    period = eng.tables().step.period
    groups = 8

    def advance(cycles):  # same launch shape as the timed region, so a profile of this process sees one kernel shape
        for _ in range((cycles + cps - 1) // cps):
            step_once()

    for gk in range(groups):
        sel = (np.arange(n) % groups) <= gk
        eng.set_velocity(lin * sel[:, None], ang * sel)
        advance(max(1, period // groups))
    eng.set_velocity(lin, ang)
    advance(2 * period + 64)
    eng.synchronize()
    _, _, ws = eng.body_state()
    moving_frac = float((ws == 1).mean())

    gathered = None
    # joint-state shard of this rank in the C ABI's instance-major layout [n][legs][dof] (device resident)
    qshard = torch.empty(n * p.leg_count * p.leg_dof[0], dtype=torch.float64, device="cuda") if use_dist else None
    peer = None
    if use_dist and gather_form == "peer":   # the exchange as peer copies over xGMI (shc_peer_*): one copy per link instead of a ring bound by one link
        from syropod_highlevel_controller_amd.parallel import PeerAllGather
        peer = PeerAllGather(qshard.numel(), world, rank, local_rank)
        gathered = peer.out
    elif use_dist:
        gathered = torch.empty(world * qshard.numel(), dtype=torch.float64, device="cuda")

    def gather():
        if use_dist:
            eng.joints_device(qshard.data_ptr(), None)  # SoA planes -> [n][legs][dof] on the engine's stream
            if peer:
                peer.gather(qshard, stream.cuda_stream)     # this rank's shard into every rank's buffer; closed by gather_close() below
            else:
                all_gather_joints(qshard, world, out=gathered)  # the helper tests/test_sharding_gloo.py runs over gloo

    def gather_close():   # peer form: this rank's copies have completed (the caller synchronised) - the exchange is complete when every rank's have
        if peer:
            if host_barrier.slots is not None:
                host_barrier()
            if host_barrier.slots is None:   # (no shared page, or the spin barrier gave up: the collective library's barrier)
                dist.barrier()

    if mode == "step_k":   # K = 16 control cycles per launch, each with its own input row (shc_engine_step_k), as the measured form: a step is still ONE control cycle
        if use_dist:
            raise SystemExit("--mode step_k is a single-GPU form here")
        K = 16
        for _ in range(warmup):
            step_once()
        fk = fused_k_probe(eng, p, n, lin, ang, extra, key, stream, K=K, reps=max(4, steps // K), want_parity=want_parity)
        q, _ = eng.joints()
        finite = bool(np.isfinite(q).all())
        eng.close()
        nm = name + ("+efforts" if joint_efforts else "") + ":stepk"
        roof = dict(fk["roofline"], traffic=measured_traffic(nm, n, K))
        roof = with_issue_side(roof, measured_valu(nm, n, K), fk["ms_per_launch"] * 1e-3)
        res = {"value": fk["value"], "elapsed": fk["launches"] * K * fk["ms_per_cycle"] * 1e-3, "ms_per_step": fk["ms_per_cycle"], "parity": fk["parity"], "host_issue_ms_per_step": None,
               "config": {"workload": f"BASELINE.json {name}: {n} {desc}", "instances_per_gpu": n, "cycles_per_step": 1, "legs": p.leg_count, "dof": p.leg_dof[0], "seed": seed,
                          "mode": f"shc_engine_step_k: {K} control cycles per launch, a new input set in every cycle ({', '.join(fk['inputs_per_cycle'])}); {fk['launches']} launches timed",
                          "mode_short": f"step_k: {K} cycles per launch, new inputs every cycle", "gather_short": "none (N = 1)", "gather": "none", "moving_fraction": moving_frac,
                          "finite": finite, "K": K, "launches": fk["launches"], "ms_per_launch": fk["ms_per_launch"], "two_stream_split": n_waves >= 4096},
               "roofline": roof}
        if want_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(p, lin, ang, extra)
        return res
    # resident mode: batches that fit the chip once, one cycle per step, nothing that needs a launch between steps
    resident = False
    if use_dist and mode == "auto":
        mode = "launch"   # N > 1: launch mode unless resident mode is asked for (north_star states the scaling target on config 4, which does not fit the chip once)
    if mode != "launch" and cps == 1 and not force_sets and not gather_every:
        try:
            eng.resident_begin(ring_depth=4, max_cycles=4)
            eng.resident_end()
            resident = True
        except Exception as exc:  # noqa: BLE001
            if mode == "resident":
                raise
            resident = False
    res_cycle_s = None
    launch_elapsed = None
    posted_value = posted_launch_value = None
    host_barrier = None
    parity = None
    nogather_elapsed = gather_s = own_elapsed = None
    scale_reference = efficiency = None
    enqueue_s = None
    wd_meta = {"n_gpus": world, "config": {"workload": f"BASELINE.json {name}: {n} per GPU", "mode": mode}}
    if use_dist:
        host_barrier = HostSpinBarrier(world, rank)   # while a resident loop is alive no device-wide synchronisation may be issued
        gather()
        torch.cuda.synchronize()
        gather_close()
        dist.barrier()   # while a resident loop is alive no device-wide synchronisation may be issued
    if resident:   # the launch-per-cycle figure of the same engine first (secondary), then the resident one
        for _ in range(warmup):
            step_once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step_once()
        torch.cuda.synchronize()
        launch_elapsed = time.perf_counter() - t0
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # (time_resident's own clock brackets exactly `steps` doorbell ticks; the gather of the final joints follows inside the region)
        pfac = (lambda: ParityWindow(eng, p, lin, ang, extra, state["force"])) if want_parity else None
        if use_dist and gather_under_loop:   # asked for: the gather is queued behind a device-side wait while the loop is alive
            def final_gather(cycle):
                eng.resident_joints_async(cycle, qshard.data_ptr())
                all_gather_joints(qshard, world, out=gathered)
            with Watchdog(180, "resident region with the all-gather queued under the live loop", rank, wd_meta):
                res_elapsed, res_cycle_s, parity = time_resident(eng, n, steps, warmup, stream, final_gather=final_gather, host_barrier=host_barrier, parity=pfac)
        elif use_dist:   # N > 1 default: end the loop, then gather - no collective kernel next to a persistent loop
            with Watchdog(180, "resident region, loop ended before the all-gather", rank, wd_meta):
                def gather_and_close():
                    gather()
                    if peer:
                        stream.synchronize()
                        gather_close()
                res_elapsed, res_cycle_s, parity = time_resident(eng, n, steps, warmup, stream, gather_after_end=gather_and_close, host_barrier=host_barrier, parity=pfac)
        else:
            res_elapsed, res_cycle_s, parity = time_resident(eng, n, steps, warmup, stream, parity=pfac)
        elapsed = res_elapsed
        # secondary figure: a NEW velocity command for every robot in every cycle, from arrays resident in HBM.  Direct posts (ABI 4): two
        # bound input sets, alternated as a node's callbacks would fill one message while the controller reads the other; a post is a 16-byte
        # record in host-mapped memory (no kernel launch) that the relay wavefront turns into the cycle's header, the workers read the arrays.
        posted_value = posted_launch_value = None
        if not use_dist and not os.environ.get("SHC_BENCH_NO_POSTED_PROBE"):   # (profiling runs skip it: one K-cycle launch to attribute counters to)
            d_sets = [(torch.from_numpy(np.ascontiguousarray(lin)).cuda(), torch.from_numpy(np.ascontiguousarray(ang)).cuda()) for _ in range(2)]
            for k, (dl, da) in enumerate(d_sets):
                eng.resident_bind_inputs(k, velocity=(dl.data_ptr(), da.data_ptr()))
            kk = max(steps, 300)
            eng.resident_begin(ring_depth=16, max_cycles=kk + 16)
            posters = [eng.resident_direct_poster(k, velocity=True) for k in range(2)]
            rcs = [posters[i & 1]() for i in range(10)]
            eng.resident_wait(10)
            tp = time.perf_counter()
            for i in range(kk):
                rcs.append(posters[i & 1]())
            eng.resident_wait(10 + kk, 60000)
            posted_value = n * kk / (time.perf_counter() - tp)
            assert not any(rcs), "direct post failed"
            eng.resident_end()
            # ... and the same through the input rings (one post-and-publish kernel launch per cycle: round 3's form)
            eng.resident_begin(ring_depth=16, max_cycles=kk + 16)
            for _ in range(10):
                eng.resident_post(velocity=(d_sets[0][0].data_ptr(), d_sets[0][1].data_ptr()), on_device=True, publish=True)
            eng.resident_wait(10)
            tp = time.perf_counter()
            for _ in range(kk):
                eng.resident_post(velocity=(d_sets[0][0].data_ptr(), d_sets[0][1].data_ptr()), on_device=True, publish=True)
            eng.resident_wait(10 + kk, 60000)
            posted_launch_value = n * kk / (time.perf_counter() - tp)
            eng.resident_end()
        # N = 1: the region closes with shc_engine_resident_wait - every wave has completed the K-th cycle and its joint state is
        # visible (the device-to-host completion handshake).  The loop kernel is still alive at that point, so a stream
        # synchronisation cannot be the closing bracket here; one issued after resident_end would only time an idle device.
    else:
        for _ in range(warmup):
            step_once()
        pw = open_parity_window() if want_parity else None
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        with Watchdog(300, "timed region: steps + all-gather of the final joints" if use_dist else "timed region", rank, wd_meta):
            t0 = time.perf_counter()
            for i in range(steps):
                step_once()
                if gather_every and (i + 1) % gather_every == 0:
                    gather()
            if not gather_every or steps % gather_every:
                gather()
            enqueue_s = time.perf_counter() - t0   # the host's share: the launches are asynchronous, this is how long issuing them took
            torch.cuda.synchronize()
            gather_close()
            elapsed = time.perf_counter() - t0  # this rank's K steps + its part of the gather; the MAX over ranks below is the job's time
        state["window"] = None
        if pw:
            parity = pw.evaluate(steps * cps, eng.joints()[0])
            parity["window"] = f"the {steps} timed steps ({steps * cps} control cycles, one launch of the fused cycle kernel per step)"
        if use_dist:   # the same K steps without the gather, and the gather on its own (reported next to `value`, which contains both)
            with Watchdog(300, "steps without the gather / gather alone", rank, wd_meta):
                dist.barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(steps):
                    step_once()
                torch.cuda.synchronize()
                nogather_elapsed = time.perf_counter() - t1
                own_elapsed = nogather_elapsed   # this rank's own K steps, before the maximum over the ranks is taken: the same-workload single-GPU reference
                dist.barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                gather()
                torch.cuda.synchronize()
                gather_close()
                gather_s = time.perf_counter() - t1
    if use_dist:
        dist.barrier()  # closing bracket (the all-gather inside the region already needed every rank's shard)
        torch.cuda.synchronize()
    if use_dist:
        # the gathered buffer must hold every rank's shard in rank order: check this rank's own slice
        own = gathered[rank * qshard.numel():(rank + 1) * qshard.numel()]
        assert torch.equal(own, qshard), "all-gather returned a different shard for this rank"
        red_dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
        t = torch.tensor([elapsed, nogather_elapsed or 0.0, gather_s or 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if nogather_elapsed is not None:
            nogather_elapsed, gather_s = float(t[1].item()), float(t[2].item())
        if own_elapsed is not None:
            # One curve from N = 1 to N = 8: the SAME workload on ONE GPU, measured in this run - every rank's own K steps of its shard, no
            # gather, its own clock (the ranks run side by side, so power and thermals are the job's).  Efficiency = job value / (N x that), with and
            # without the gather; a driver that divides value(N) by N x value(1) of the default N = 1 line would compare octopod launches with hexapod
            # doorbell ticks.
            own = torch.zeros(world, dtype=torch.float64, device=red_dev)
            own[rank] = own_elapsed
            dist.all_reduce(own, op=dist.ReduceOp.SUM)
            per_rank = [n * steps * cps / float(x) for x in own.tolist()]
            ref = float(np.median(per_rank))
            scale_reference = {"workload": f"BASELINE.json {name}: {n} {desc}, one launch of the fused cycle kernel per step, no gather", "n_gpus": 1,
                               "value": ref, "ms_per_step": n * cps / ref * 1e3, "value_is": "median over the ranks of each rank's own K steps of its shard (measured in this run, after the timed region)",
                               "per_rank_values": per_rank}
            efficiency = {"with_gather": (world * n * steps * cps / elapsed) / (world * ref), "without_gather": (world * n * steps * cps / nogather_elapsed) / (world * ref),
                          "is": "job value / (n_gpus x scale_reference.value): `value` contains the all-gather of the final joint buffer inside a region of only `steps` steps"}

    # ---- kernel duration of the cycle kernel, HIP events on the launch stream.  Two upper bounds on the true duration:
    #      (a) one event pair per launch (adds the event-record latency), (b) one pair around m back-to-back launches
    #      (adds the inter-kernel gaps).  The smaller one is reported; rocprofv3's kernel-trace average agrees with it.
    m = min(steps, 200)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(m)]
    for a, b in evs:
        a.record(stream)
        eng.step(cps)
        eng.join()      # (large batches: the second half of the step runs on the engine's internal stream)
        b.record(stream)
    torch.cuda.synchronize()
    per_launch_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(m):
        eng.step(cps)
    eng.join()
    e1.record(stream)
    torch.cuda.synchronize()
    kern_ms = min(per_launch_ms, e0.elapsed_time(e1) / m)
    # ---- secondary figure: 16 control cycles fused per launch (inputs held, state in registers between cycles)
    fused_value = None
    fused_k = None
    if cps == 1 and world == 1 and fused_probe:
        fc, reps = 16, max(64, steps // 16)   # (64 launches of 16 cycles: 25 - 125 ms per workload - a steady-state figure like the 300+ single launches above)
        for _ in range(3):
            eng.step(fc)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for _ in range(reps):
            eng.step(fc)
        torch.cuda.synchronize()
        fused_value = n * fc * reps / (time.perf_counter() - tf0)
        if not resident:   # ... and the same K with a NEW input set in every cycle (what StateController::loop sees: callbacks every iteration)
            try:
                fused_k = fused_k_probe(eng, p, n, lin, ang, extra, key, stream, K=fc, reps=reps, want_parity=want_parity)
            except Exception as exc:  # noqa: BLE001  (a configuration without a loop-form kernel: reported, not fatal)
                fused_k = {"error": str(exc)}
    # ---- secondary figure for batches large enough for the engine's two-stream split (shc_engine_step): the same steps as ONE launch
    #      each on the engine's stream (SHC_FEAT_SINGLE_STREAM), i.e. what round 2 measured as the primary figure
    single_stream = None
    if cps == 1 and world == 1 and fused_probe and n_waves >= 4096:
        from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, FEAT_SINGLE_STREAM
        eng.set_features(FEAT_DEFAULT | FEAT_SINGLE_STREAM)
        for _ in range(20):
            step_once()
        torch.cuda.synchronize()
        reps = max(300, min(steps, 1000))
        t1 = time.perf_counter()
        for _ in range(reps):
            step_once()
        eng.synchronize()
        dt1 = (time.perf_counter() - t1) / reps
        single_stream = {"value": n / dt1, "ms_per_step": dt1 * 1e3, "algorithmic_frac": ALG_BYTES_PER_CYCLE[key] * n / dt1 / 1e9 / HBM_PEAK_GBS,
                         "note": "one launch per step on the engine's stream (SHC_FEAT_SINGLE_STREAM)"}
        eng.set_features(FEAT_DEFAULT)
    q, _ = eng.joints()
    finite = bool(np.isfinite(q).all())
    eng.close()
    # measured joint torques: per leg the torques themselves (dof x 8 B read) and the filtered tip-force estimate Leg::calculateTipForce keeps (24 B read + 24 B written)
    effort_bytes = p.leg_count * (p.leg_dof[0] * 8 + 48) if joint_efforts else 0
    alg_bytes = (ALG_BYTES_PER_CYCLE[key] + effort_bytes) * n * cps
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    launch_roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "traffic": measured_traffic(name + ("+efforts" if joint_efforts else ""), n, cps), "kernel": ("shc_cycle_half_kernel<walker half> + <model half> (a rotation-constrained cycle is two launches, two wavefronts per SIMD each)" if name in ("gravity", "gravity3") and n_waves >= 2048 else "shc_cycle_kernel")
                                 + (" (a step = that for each half of the batch, the halves on two streams)" if n_waves >= 4096 else ""),
                       "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                       "algorithmic_bytes_are": f"SURVEY.md 8(d): {ALG_BYTES_PER_CYCLE[key]} B per robot and cycle" + (f" + {effort_bytes} B of joint torques read and tip-force estimate read + written" if effort_bytes else "")}
    launch_roofline = with_issue_side(launch_roofline, measured_valu(name + ("+efforts" if joint_efforts else ""), n, cps), kern_ms * 1e-3)
    if resident:
        rb = resident_bytes_per_cycle(p) * n
        r_ach = rb / res_cycle_s / 1e9
        roofline = {"bound": "hbm", "achieved": r_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r_ach / HBM_PEAK_GBS,
                    "traffic": measured_traffic(name + ":resident", n, cps), "kernel": "shc_resident2_kernel (one launch, K = 4000 cycles; two wavefronts per robot group)",
                    "kernel_ms": res_cycle_s * 1e3, "kernel_ms_is": "per cycle: HIP events around one launch of K cycles / K",
                    "algorithmic_bytes_per_launch": rb, "algorithmic_bytes_are": "per cycle, SURVEY.md 8(d) with the state on the chip: velocity input + published q, qd",
                    "bound_note": ("the resident cycle is bound by one wavefront's dependent-issue latency, not by HBM (frac is small by construction): 592 VALU + 157 SALU + "
                                   "50 LDS instructions per wave and cycle in 5 650 wave clocks, VALU issue share 0.43 of the wave's lifetime on 824 of the chip's 1 024 SIMDs "
                                   "(profiles/r05_config2_resident_rocprofv3.txt); attribution of the loop skeleton in profiles/r05_probe_resident_skeleton.txt"),
                    "state_streaming_equivalent_frac": ALG_BYTES_PER_CYCLE[key] * n / res_cycle_s / 1e9 / HBM_PEAK_GBS,
                    "one_launch_per_cycle": launch_roofline}
        roofline = with_issue_side(roofline, measured_valu(name + ":resident", n, cps), res_cycle_s)
    else:
        roofline = launch_roofline
    res = {
        "value": world * n * steps * cps / elapsed, "elapsed": elapsed, "ms_per_step": elapsed / steps * 1e3, "parity": parity,
        "host_issue_ms_per_step": (enqueue_s / steps * 1e3) if enqueue_s else None,   # launch mode: the host's time to issue a step's launches (asynchronous)
        "config": {"workload": f"BASELINE.json {name}: {n} {desc}", "instances_per_gpu": n, "cycles_per_step": cps,
                   "mode": ("resident: one launch stays on the chip, a step = one doorbell tick = one control cycle with that cycle's inputs from the "
                            "device-side rings and its q / qd to the output ring") if resident else "one launch of the fused cycle kernel per step",
                   "mode_short": "resident loop: 1 step = 1 doorbell tick = 1 control cycle" if resident else "one launch of the fused cycle kernel per step",
                   "gather_short": (f"all-gather every {gather_every} steps" if gather_every else ("final joint buffer, in the timed region" if use_dist else "none (N = 1)")),
                   "gather_form_short": (("peer copies (xGMI)" if peer else "RCCL all_gather") if use_dist else None),
                   "one_launch_per_cycle_value": (world * n * steps * cps / launch_elapsed) if launch_elapsed else None,
                   "velocities_posted_every_cycle_value": posted_value,   # resident mode, a new velocity set per robot and cycle from device arrays (direct posts)
                   "velocities_posted_every_cycle_through_the_rings_value": posted_launch_value,   # ... with one post kernel launch per cycle (round 3's form)
                   "legs": p.leg_count, "dof": p.leg_dof[0],
                   "gather": f"all-gather of the joint buffer every {gather_every} steps" if gather_every
                   else ("one all-gather of the final joint buffer, inside the timed region: queued on the engine's stream behind a device-side wait for the "
                         "region's last cycle before the first tick (stream-ordered, like a captured graph); the region closes when it has completed"
                         if (use_dist and resident and gather_under_loop) else
                         ("one all-gather of the final joint buffer inside the timed region, after shc_engine_resident_end (the persistent loop has left the chip before "
                          "any collective kernel runs)" if (use_dist and resident) else
                          "one all-gather of the final joint buffer (N > 1), launched after the last step inside the timed region; value_without_gather / gather_ms: "
                          "the same K steps without it and the gather on its own, measured right after (max over ranks)")),
                   "gather_form": (("peer copies over xGMI (shc_peer_scatter: this rank's shard into every rank's buffer, one stream per link) + a host barrier" if peer
                                    else "RCCL all_gather_into_tensor") if use_dist else None),
                   "scale_reference": scale_reference, "weak_scaling_efficiency": efficiency,
                   "value_without_gather": (world * n * steps * cps / nogather_elapsed) if nogather_elapsed else None,
                   "gather_ms": gather_s * 1e3 if gather_s is not None else None,
                   "gather_bytes_per_rank": int(qshard.numel() * 8) if use_dist else None,
                   "moving_fraction": moving_frac, "finite": finite, "seed": seed, "fused_16_cycles_per_launch_value": fused_value,
                   "fused_K_with_per_cycle_inputs_value": (fused_k or {}).get("value"), "fused_K_with_per_cycle_inputs": fused_k,
                   "two_stream_split": n_waves >= 4096, "single_stream": single_stream},
        "roofline": roofline,
    }
    if want_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(p, lin, ang, extra)
    return res


DEFAULT_INSTANCES = {"config2": 4096, "config3": 65536, "config4": 131072, "config4full": 1 << 20, "config5": 1 << 20, "rough": 65536, "gravity": 65536, "gravity3": 65536}
# (legs, dof, gait); a tuple of DOFs = a robot whose legs differ in joint count (BASELINE.json configs[4]: "3-5 DOF per leg")
CONFIG5_BINS = ((4, 3, "tripod"), (4, 4, "amble"), (6, 4, "ripple"), (8, 3, "wave"), (6, 5, "tripod"), (6, (3, 5, 4, 3, 5, 4), "ripple"))


def config5_morphology(legs, dof, gait):
    from syropod_highlevel_controller_amd import synthetic_mixed_dof_params, synthetic_octopod_params
    return synthetic_mixed_dof_params(gait, dof) if isinstance(dof, tuple) else synthetic_octopod_params(gait, dof, legs)


def config5_bytes(legs, dof):
    """SURVEY.md section 8(d) per robot and cycle: per leg 2 x ((2 DOF + 24) doubles + one packed int), per robot 2 x 28 B + the 24 B velocity input."""
    dofs = dof if isinstance(dof, tuple) else (dof,) * legs
    return 2 * (sum((2 * d + 24) * 8 + 4 for d in dofs) + 28) + 24


def run_config5(n, steps, warmup, seed, want_parity=True):
    """BASELINE.json configs[4]: mixed morphologies (4 / 6 / 8 legs, 3 - 5 joints - one bin with legs of 3, 5 and 4 joints in the same
    robot -, all four gaits), instance i has morphology i mod 6 - the worst interleaving for a one-kernel design.  shc_fleet_create bins the instances (one engine + one HIP stream
    per morphology), so the device work is the same whatever the order; the interleaved and the sorted ("binned") order differ
    only in the host-side permutation of the boundary arrays, reported separately."""
    import torch
    from syropod_highlevel_controller_amd.fleet import MixedFleet
    from syropod_highlevel_controller_amd.parallel import velocity_inputs
    morphs = [config5_morphology(l, d, g) for l, d, g in CONFIG5_BINS]
    lin, ang = velocity_inputs(seed ^ 0x5EED5, 0, n)
    out = {}
    alg = 0
    parity = None
    for order in ("interleaved", "binned"):
        mid = np.arange(n) % len(morphs) if order == "interleaved" else np.sort(np.arange(n) % len(morphs))
        fleet = MixedFleet(morphs, mid)
        t0 = time.perf_counter()
        fleet.set_velocity(lin, ang)
        t_in = time.perf_counter() - t0
        for _ in range(45):        # walk until every instance is MOVING (longest period: wave, 8 legs)
            fleet.step(16)
        for _ in range(warmup):
            fleet.step(1)
        fleet.synchronize()
        torch.cuda.synchronize()
        windows = []
        if want_parity and order == "interleaved":   # the first PARITY_INSTANCES robots of every bin against the oracle over the timed window (untimed set-up)
            from syropod_highlevel_controller_amd.engine import BatchEngine
            for handle, mk, _dev, ids in fleet.parts():
                view = BatchEngine.view(handle, morphs[mk], len(ids))
                windows.append((mk, view, ParityWindow(view, morphs[mk], lin[ids], ang[ids], {})))
        t0 = time.perf_counter()
        for _ in range(steps):
            fleet.step(1)
        fleet.synchronize()
        elapsed = time.perf_counter() - t0
        t0 = time.perf_counter()
        q, _ = fleet.joints()
        t_out = time.perf_counter() - t0
        if windows:
            per_bin = []
            for mk, view, pw in windows:
                r = pw.evaluate(steps, view.joints()[0])
                r["bin"] = list(CONFIG5_BINS[mk]) if not isinstance(CONFIG5_BINS[mk][1], tuple) else [CONFIG5_BINS[mk][0], list(CONFIG5_BINS[mk][1]), CONFIG5_BINS[mk][2]]
                per_bin.append(r)
            ok = [r for r in per_bin if r["max_abs_dq"] is not None]
            parity = {"max_abs_dq": max(r["max_abs_dq"] for r in ok) if ok else None, "max_abs_dq_all_instances": max(r["max_abs_dq_all_instances"] for r in per_bin),
                      "unit": "rad", "instances": sum(r["instances"] for r in per_bin), "cycles": steps, "tolerance": 1e-6,
                      "well_posed_fraction": float(np.mean([r["well_posed_fraction"] for r in per_bin])),
                      "against": per_bin[0]["against"] + "; per morphology bin below", "window": f"the {steps} timed fleet steps", "bins": [
                          {k: r[k] for k in ("bin", "max_abs_dq", "max_abs_dq_all_instances", "well_posed_fraction", "instances")} for r in per_bin]}
        moving = float((fleet.walk_state() == 1).mean())
        alg = sum(int((mid == k).sum()) * config5_bytes(l, d) for k, (l, d, g) in enumerate(CONFIG5_BINS))
        out[order] = {"value": n * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "moving_fraction": moving,
                      "finite": bool(np.isfinite(q[~np.isnan(q)]).all()), "host_set_velocity_s": t_in, "host_get_joints_s": t_out}
        fleet.close()
    r = out["interleaved"]
    achieved = alg / (r["ms_per_step"] * 1e-3) / 1e9
    return {"workload": f"BASELINE.json config5: {n} mixed-morphology robots (legs x dof, gait) = {list(CONFIG5_BINS)}, instance i -> bin i mod {len(CONFIG5_BINS)}, "
                        "binned by shc_fleet_create onto one engine + HIP stream per morphology", "value": r["value"], "unit": "control-cycles/s",
            "steps": steps, "ms_per_step": r["ms_per_step"], "moving_fraction": r["moving_fraction"], "finite": r["finite"], "parity": parity,
            "interleaved_vs_binned": out,
            "roofline": with_issue_side({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                         "traffic": measured_traffic("config5", n, 1),
                                         "kernel": f"shc_cycle_kernel x {len(CONFIG5_BINS)} morphologies on concurrent streams (wall clock per fleet step, not a single launch)",
                                         "algorithmic_bytes_per_launch": alg}, measured_valu("config5", n, 1), r["ms_per_step"] * 1e-3)}


def run_config4_full(steps, warmup, seed, gather_form="rccl", want_parity=True, shards=8):
    """BASELINE.json configs[3] at its STATED size on ONE MI355X: 2^20 synthetic octopods (8 legs x 5 DOF), ripple gait, as the eight
    contiguous shards of 131 072 the 8-GPU job would own - one engine + HIP stream per shard (shc_fleet_create with the device id repeated), inputs keyed
    by the global instance id, nothing exchanged while stepping - followed by the exchange at its TRUE message size: shc_fleet_all_gather_joints fills a
    gathered [2^20][8][5] buffer (335.5 MB) for every shard slot, device-to-device copies (on one device they run over HBM instead of xGMI).
    value = the stepping alone (comparable with the config-4 share line); gather_ms and value_with_gather are reported next to it."""
    import torch
    from syropod_highlevel_controller_amd import synthetic_octopod_params
    from syropod_highlevel_controller_amd.engine import BatchEngine
    from syropod_highlevel_controller_amd.fleet import MixedFleet
    from syropod_highlevel_controller_amd.parallel import velocity_inputs
    n = 1 << 20
    p = synthetic_octopod_params("ripple", 5, 8)
    lin, ang = velocity_inputs(seed, 0, n)      # the same commands instance for instance as the ranks of the 8-GPU job draw (make_workload)
    fleet = MixedFleet([p], np.zeros(n, dtype=np.int32), devices=[torch.cuda.current_device()] * shards)
    period = 0
    groups = 8
    parts = fleet.parts()
    period = BatchEngine.view(parts[0][0], p, len(parts[0][3])).tables().step.period
    for gk in range(groups):        # de-phase as run_workload does
        sel = (np.arange(n) % groups) <= gk
        fleet.set_velocity(lin * sel[:, None], ang * sel)
        fleet.step(max(1, period // groups))
    fleet.set_velocity(lin, ang)
    fleet.step(2 * period + 64)
    for _ in range(warmup):
        fleet.step(1)
    fleet.synchronize()
    torch.cuda.synchronize()
    pw = None
    if want_parity:   # the first PARITY_INSTANCES robots of shard 0 and of the last shard against the oracle over the timed window
        pw = []
        for handle, _mk, _dev, ids in (parts[0], parts[-1]):
            view = BatchEngine.view(handle, p, len(ids))
            pw.append((view, ParityWindow(view, p, lin[ids], ang[ids], {})))
    t0 = time.perf_counter()
    for _ in range(steps):
        fleet.step(1)
    fleet.synchronize()
    elapsed = time.perf_counter() - t0
    # the exchange at its true size (untimed first call allocates the gathered buffers)
    fleet.all_gather_joints()
    fleet.synchronize()
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        bufs = fleet.all_gather_joints()
        fleet.synchronize()
    gather_s = (time.perf_counter() - t0) / reps
    gather_bytes = n * p.leg_count * p.leg_dof[0] * 8
    parity = None
    if pw:
        rs = [w.evaluate(steps, view.joints()[0]) for view, w in pw]
        ok = [r for r in rs if r["max_abs_dq"] is not None]
        parity = dict(rs[0], max_abs_dq=max(r["max_abs_dq"] for r in ok) if ok else None, max_abs_dq_all_instances=max(r["max_abs_dq_all_instances"] for r in rs),
                      instances=sum(r["instances"] for r in rs), well_posed_fraction=float(np.mean([r["well_posed_fraction"] for r in rs])),
                      window=f"the {steps} timed fleet steps; the first {PARITY_INSTANCES} robots of the first and of the last shard")
    # the gathered buffer of slot 0 against the joints the getter returns (instance order, all 2^20 robots)
    q, _ = fleet.joints()
    import ctypes
    got = np.empty((n, p.leg_count, p.leg_dof[0]))
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipMemcpy(ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(bufs[0]), ctypes.c_size_t(got.nbytes), 2)
    gathered_ok = bool(rc == 0 and np.array_equal(got, q))
    moving = float((fleet.walk_state() == 1).mean())
    finite = bool(np.isfinite(q).all())
    fleet.close()
    ms = elapsed / steps * 1e3
    alg = ALG_BYTES_PER_CYCLE[("octopod", 4)] * n
    ach = alg / (ms * 1e-3) / 1e9
    cfg = {"workload": f"BASELINE.json config4 at its stated size: {n} synthetic octopods (8x5 DOF), ripple gait, {shards} contiguous shards of {n // shards} on ONE device "
                       "(shc_fleet_create: one engine + HIP stream per shard), then the exchange of the final joint buffer at full size",
           "short": SHORT["config4full"], "mode": "one launch of the fused cycle kernel per shard and step (each shard's halves on two streams)", "instances_per_gpu": n,
           "legs": p.leg_count, "dof": p.leg_dof[0], "seed": seed, "shards": shards, "moving_fraction": moving, "finite": finite,
           "gather": "shc_fleet_all_gather_joints after the timed steps", "gather_form": "device-to-device copies, one buffer per shard slot (one device: over HBM)",
           "gather_ms": gather_s * 1e3, "gather_bytes_per_rank": gather_bytes // shards, "gathered_buffer_bytes": gather_bytes, "gathered_buffers": shards,
           "gathered_buffer_matches_getter": gathered_ok, "value_without_gather": n * steps / elapsed,
           "value_with_gather": n * steps / (elapsed + gather_s), "steps": steps}
    roof = with_issue_side({"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": measured_traffic("config4full", n, 1),
                            "kernel": f"shc_cycle_kernel<8,5> x {shards} shards on concurrent streams (wall clock per fleet step)", "kernel_ms": ms,
                            "algorithmic_bytes_per_launch": alg}, measured_valu("config4full", n, 1), ms * 1e-3)
    assert gathered_ok, "the gathered buffer differs from the joints the getter returns"
    return {"workload": cfg["workload"], "value": n * steps / elapsed, "unit": "control-cycles/s", "steps": steps, "ms_per_step": ms, "config": cfg, "roofline": roof,
            "parity": parity, "gather_ms": gather_s * 1e3}


LINE_LIMIT = 4096          # hard limit of the final stdout line (the driver keeps a bounded tail of stdout and parses its last line)
DETAILS_FILE = "bench_details.json"


def _sig(x, digits=5):
    """Numbers of the compact line carry `digits` significant digits (the full-precision figures are in bench_details.json)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float(f"{float(x):.{digits}g}")
    except (TypeError, ValueError):
        return None


def _short(s, limit):
    s = str(s)
    return s if len(s) <= limit else s[:limit - 2] + ".."


def _pick_roofline(r, with_kernel=True):
    if not r:
        return None
    out = {"bound": r.get("bound"), "achieved": _sig(r.get("achieved")), "peak": r.get("peak"), "unit": r.get("unit"), "frac": _sig(r.get("frac"), 4),
           "traffic": r.get("traffic")}
    if with_kernel:
        out["kernel"] = _short(r.get("kernel", ""), 72)
    out.update({"kernel_ms": _sig(r.get("kernel_ms")), "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch"),
                "valu_issue_frac": _sig(r.get("valu_issue_frac"), 3)})
    return out


def _pick_parity(p):
    if not p:
        return None
    return {"max_abs_dq": _sig(p.get("max_abs_dq"), 3), "instances": p.get("instances"), "cycles": p.get("cycles"),
            "well_posed_fraction": _sig(p.get("well_posed_fraction"), 3), "tolerance": p.get("tolerance")}


def _also_entry(a):
    """One row of the compact `also` list.  Batches that do not fit the chip once report the K-cycles-per-launch form with a new input set in
    every cycle (shc_engine_step_k) as `value` and the one-launch-per-cycle figure next to it as `launch_value` / `launch_frac`."""
    if "error" in a:
        return {"workload": _short(a.get("short", a.get("workload", "?")), 40), "error": _short(a["error"], 80)}
    roof, par = a.get("roofline") or {}, a.get("parity") or {}
    traffic, alg = roof.get("traffic"), roof.get("algorithmic_bytes_per_launch")
    row = {"workload": _short(a.get("short") or a.get("workload", "?"), 40), "value": _sig(a.get("value")), "ms_per_step": _sig(a.get("ms_per_step")),
           "frac": _sig(roof.get("frac"), 4), "traffic_ratio": _sig(traffic / alg, 3) if (traffic and alg) else None,
           "valu_issue_frac": _sig(roof.get("valu_issue_frac"), 3), "max_abs_dq": _sig(par.get("max_abs_dq"), 3)}
    for k in ("form", "launch_value", "launch_frac", "step_k_value"):
        if a.get(k) is not None:
            row[k] = _sig(a[k], 4) if k != "form" else a[k]
    return row


def compact_line(full):
    """The ONE line the driver parses: the contract's keys, the headline's roofline / parity / cpu_baseline and one compact `also` list.  Everything
    else (prose, per-bin parity, the fused-K sub-objects, both rooflines of every secondary workload) is in bench_details.json.  Never longer than
    LINE_LIMIT bytes: rows are dropped from the end of `also` (and counted in `also_dropped`) should it ever be."""
    cfg = full.get("config", {})
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _sig(out["value"], 7), _sig(out["ms_per_step"], 6)
    c = {"workload": _short(cfg.get("short") or cfg.get("workload", ""), 100), "mode": _short(cfg.get("mode_short") or cfg.get("mode", ""), 60),
         "instances_per_gpu": cfg.get("instances_per_gpu"), "legs": cfg.get("legs"), "dof": cfg.get("dof"), "seed": cfg.get("seed"),
         "gather": _short(cfg.get("gather_short") or cfg.get("gather", ""), 60)}
    if (full.get("n_gpus") or 1) > 1 or cfg.get("gather_ms") is not None:
        sr, eff = cfg.get("scale_reference") or {}, cfg.get("weak_scaling_efficiency") or {}
        c.update({"scale_reference": {"n_gpus": 1, "value": _sig(sr.get("value")), "ms_per_step": _sig(sr.get("ms_per_step"))} if sr else None,
                  "weak_scaling_efficiency": {"with_gather": _sig(eff.get("with_gather"), 4), "without_gather": _sig(eff.get("without_gather"), 4)} if eff else None,
                  "value_without_gather": _sig(cfg.get("value_without_gather")), "gather_ms": _sig(cfg.get("gather_ms")),
                  "gather_bytes_per_rank": cfg.get("gather_bytes_per_rank"), "gather_form": _short(cfg.get("gather_form_short") or cfg.get("gather_form") or "", 24)})
    for k in ("one_launch_per_cycle_value", "velocities_posted_every_cycle_value"):
        if cfg.get(k) is not None:
            c[k] = _sig(cfg[k])
    out["config"] = c
    out["roofline"] = _pick_roofline(full.get("roofline"))
    out["parity"] = _pick_parity(full.get("parity"))
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _sig(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": _short(cb.get("sample_short") or cb.get("sample", ""), 60), "single_thread_value": _sig(cb.get("single_thread_value"))}
    if full.get("error"):
        out["error"] = _short(full["error"], 200)
    rows = [_also_entry(a) for a in (full.get("also") or cfg.get("also") or [])]
    out["also"] = rows
    out["details"] = DETAILS_FILE
    line = json.dumps(out, separators=(",", ":"))
    dropped = 0
    while len(line.encode()) >= LINE_LIMIT and out["also"]:
        out["also"] = out["also"][:-1]
        dropped += 1
        out["also_dropped"] = dropped
        line = json.dumps(out, separators=(",", ":"))
    assert len(line.encode()) < LINE_LIMIT, len(line)
    return line


def flush_c_stdio():
    """The collective library prints its version banner with C stdio, which a pipe buffers until the process exits - i.e. AFTER anything Python has printed.
    Flushing the C streams first keeps the JSON line the last line of stdout."""
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


def emit(full):
    """Full record -> bench_details.json (next to bench.py, and under gpurun_out/ when that exists) and stderr; compact record -> the LAST stdout line."""
    flush_c_stdio()
    text = json.dumps(full)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAILS_FILE), "w") as f:
                    f.write(text + "\n")
            except OSError:
                pass
    print("bench details (also in " + DETAILS_FILE + "): " + text, file=sys.stderr, flush=True)
    print(compact_line(full), flush=True)


def launcher_argv(args_list, gpus, port=None):
    """`bench.py --gpus N` started as ONE process: the command line it re-executes itself under (one rank per GPU over RCCL)."""
    port = port or os.environ.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + list(args_list)


def resolve_world(args, argv, environ=None):
    """--gpus N against the environment.  Returns ("run", world, rank, local_rank), or ("exec", argv) when this process has to start the ranks itself,
    and raises SystemExit (non-zero) when the launcher's world size and --gpus disagree - never a silent single-rank answer."""
    env = os.environ if environ is None else environ
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
        return ("run", world, int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")))
    if args.gpus > 1:
        return ("exec", launcher_argv([a for a in argv if a != "--dry-launch"], args.gpus))
    return ("run", 1, 0, 0)


SHORT = {"config2": "config2 4096 6x3 tripod", "config3": "config3 65536 6x3 wave+adm+imu", "config4": "config4 share 131072 8x5 ripple",
         "config4full": "config4 FULL 2^20 8x5 ripple, 1 GPU", "rough": "rough terrain 65536 6x3", "gravity": "gravity-aligned tips 65536 8x5",
         "gravity3": "gravity3 65536 8x5 adm+imu+rot", "config5": "config5 2^20 mixed morphologies"}


def also_record(name, r, efforts=False):
    """A secondary workload's full record (bench_details.json) with the fields the compact row is cut from."""
    cfg, fk = r["config"], r["config"].get("fused_K_with_per_cycle_inputs") or {}
    rec = {"workload": cfg["workload"], "short": SHORT.get(name, name) + (" +torques" if efforts else ""), "unit": "control-cycles/s", "steps": r.get("steps"),
           "moving_fraction": cfg["moving_fraction"], "mode": cfg["mode"], "two_stream_split": cfg["two_stream_split"], "single_stream": cfg["single_stream"],
           "one_launch_per_cycle_value": cfg["one_launch_per_cycle_value"], "fused_16_cycles_per_launch_value": cfg["fused_16_cycles_per_launch_value"],
           "fused_K_with_per_cycle_inputs": fk or None, "launch": {"value": r["value"], "ms_per_step": r["ms_per_step"], "roofline": r["roofline"], "parity": r["parity"]}}
    if fk.get("value") and not fk.get("error") and fk["value"] < r["value"]:
        # the K-cycle form is the slower one here (the 8 x 5 tip-rotation kernel with admittance + IMU posing: its batch kernel runs at one wavefront per
        # SIMD, the two launches of a cycle at two): the row reports the launch form and carries the step_k figure next to it
        rec.update({"form": "launch", "value": r["value"], "ms_per_step": r["ms_per_step"], "roofline": r["roofline"], "parity": r["parity"],
                    "step_k_value": fk["value"]})
    elif fk.get("value") and not fk.get("error"):
        # batches that do not fit the chip once: K cycles per launch, a new input set in every cycle (shc_engine_step_k), is the form a node would run them in;
        # the one-launch-per-cycle figure is the secondary one (launch_value / launch_frac / launch_traffic_ratio).
        roof = dict(fk["roofline"])
        nm = name + ("+efforts" if efforts else "") + ":stepk"
        roof["traffic"] = measured_traffic(nm, cfg["instances_per_gpu"], fk["K"])       # PMC passes of the batch kernel itself (profiles/traffic.json), else null
        roof = with_issue_side(roof, measured_valu(nm, cfg["instances_per_gpu"], fk["K"]), fk["ms_per_launch"] * 1e-3)
        roof.pop("bound_is", None)
        rec.update({"form": f"step_k K={fk['K']}", "value": fk["value"], "ms_per_step": fk["ms_per_cycle"], "roofline": roof, "parity": fk.get("parity") or r["parity"],
                    "launch_value": r["value"], "launch_frac": r["roofline"]["frac"],
                    "launch_traffic_ratio": (r["roofline"]["traffic"] / r["roofline"]["algorithmic_bytes_per_launch"]) if r["roofline"].get("traffic") else None})
    else:
        rec.update({"form": "resident" if cfg["mode"].startswith("resident") else "launch", "value": r["value"], "ms_per_step": r["ms_per_step"],
                    "roofline": r["roofline"], "parity": r["parity"]})
        if cfg.get("one_launch_per_cycle_value"):
            rec["launch_value"] = cfg["one_launch_per_cycle_value"]
            rec["launch_frac"] = (r["roofline"].get("one_launch_per_cycle") or {}).get("frac")
    return rec


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node.  Started as ONE process with N > 1, bench.py re-executes itself under "
                    "`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); under a launcher WORLD_SIZE must equal N")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default=None, help="default: config2 on one GPU (the configuration the metric is quoted on); N > 1: config4, 131 072 octopods "
                    "per GPU in launch mode (the 2^20-instance batch north_star states the weak-scaling target on)")
    ap.add_argument("--dry-launch", action="store_true", help="print (as JSON) what --gpus N resolves to - the argv of the launcher it would exec, or the rank it "
                    "would run as - and exit; needs no GPU")
    ap.add_argument("--oversubscribe", action="store_true", help="N > 1 on a node with fewer than N GPUs: rank r uses device r mod device_count (tests; never a scaling figure)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="process-group backend of N > 1: nccl = RCCL (the product); gloo only for the launcher test with "
                    "two ranks on one device (RCCL refuses duplicate devices) together with --gather peer")
    ap.add_argument("--gather-under-loop", action="store_true", help="N > 1 with --mode resident: queue the all-gather behind a device-side wait while the "
                    "persistent loop is still alive (default: the loop is ended first)")
    ap.add_argument("--gather", choices=("rccl", "peer"), default="rccl", help="N > 1: the exchange of the final joint buffer - RCCL's all-gather (a ring: one xGMI link "
                    "bounds it), or peer copies (every rank writes its shard into every rank's buffer, one copy per link)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity block (max |dq| against the CPU oracle over the timed window)")
    ap.add_argument("--instances", type=int, default=0, help="instances per GPU (default: 4096 for config2)")
    ap.add_argument("--cycles-per-step", type=int, default=1)
    ap.add_argument("--gather-every", type=int, default=0, help="all-gather the joint buffer every G steps (0 = once, at the end)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused-probe", action="store_true", help="skip the secondary 16-cycles-per-launch figure (keeps rocprof stats to one launch shape)")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads of the default run (the `also` list)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the all-gather path even with one rank")
    ap.add_argument("--joint-efforts", action="store_true", help="supply measured joint torques: the tip-force estimate (Leg::calculateTipForce) is evaluated "
                    "every cycle (the default for config2, the headline; the other workloads are BASELINE.json's \"IK + Bezier\" / tip-state-message variants without it)")
    ap.add_argument("--no-joint-efforts", action="store_true", help="primary line without measured joint torques (Leg::calculateTipForce idle)")
    ap.add_argument("--mode", choices=("auto", "resident", "launch", "step_k"), default="auto",
                    help="auto: resident mode where the batch fits the chip once (config 2), one launch per step otherwise; step_k: 16 cycles per launch with a new "
                         "input set in every cycle (shc_engine_step_k) as the timed form")
    ap.add_argument("--seed", type=int, default=0xC0FFEE)
    args = ap.parse_args(argv)

    plan = resolve_world(args, argv)
    if args.dry_launch:
        print(json.dumps({"action": plan[0], "argv": plan[1]} if plan[0] == "exec" else {"action": "run", "world": plan[1], "rank": plan[2], "local_rank": plan[3]}))
        return 0
    if plan[0] == "exec":
        # one rank per GPU, started by this process: HIP_VISIBLE_DEVICES is left as it is (rank r takes device r)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.stdout.flush()
        os.execvpe(plan[1][0], plan[1], env)
    _, world, rank, local_rank = plan

    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the batched engine has no CPU fallback")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if not args.oversubscribe:
            raise SystemExit(f"bench.py: rank {rank} needs device {local_rank} but this node has {ndev} GPU(s); --gpus must not exceed the GPUs of the node "
                             "(--oversubscribe shares devices between ranks for tests)")
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "gloo":
            if args.gather != "peer":
                raise SystemExit("bench.py: --backend gloo carries no device collective; use it with --gather peer (launcher tests only)")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.workload is None:
        args.workload = "config4" if world > 1 else "config2"
    n = args.instances or DEFAULT_INSTANCES[args.workload]
    head = {"metric": "control-cycles/sec (all legs IK-solved)", "unit": "control-cycles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    import gc
    gc.disable()   # no collector pass inside a timed region (a 20-tick region is 65 us long); collected by hand between the workloads
    if args.workload == "config5":
        if world > 1:
            raise SystemExit("config5 is a single-GPU workload here (the fleet shards in-process: shc_fleet_create device_ids)")
        r = run_config5(n, args.steps, args.warmup, args.seed, want_parity=not args.no_parity)
        cfg = {k: v for k, v in r.items() if k not in ("value", "roofline", "ms_per_step", "parity")}
        cfg.update({"short": SHORT["config5"], "mode": "one launch per morphology bin and step, bins on concurrent streams", "instances_per_gpu": n, "seed": args.seed})
        emit(dict(head, value=r["value"], ms_per_step=r["ms_per_step"], config=cfg, roofline=r["roofline"], parity=r["parity"]))
        return 0
    if args.workload == "config4full":   # BASELINE.json configs[3] at its stated size on ONE device: 8 shards of 131 072 octopods + the full-size exchange
        r = run_config4_full(args.steps, args.warmup, args.seed, gather_form=args.gather, want_parity=not args.no_parity)
        emit(dict(head, value=r["value"], ms_per_step=r["ms_per_step"], config=r["config"], roofline=r["roofline"], parity=r["parity"]))
        return 0
    # Measured joint torques are part of the primary workload (a node always receives them from jointStatesCallback and the
    # reference evaluates Leg::calculateTipForce every cycle, model.cpp:938); the variant without them is reported under `also`.
    primary_efforts = not args.no_joint_efforts
    res = run_workload(args.workload, n, args.steps, args.warmup, args.cycles_per_step, args.seed,
                       dist_ctx=(world, rank, local_rank) if use_dist else None, gather_every=args.gather_every,
                       fused_probe=not args.no_fused_probe, want_cpu_baseline=(rank == 0 and world == 1 and not args.no_cpu_baseline),
                       joint_efforts=primary_efforts and (args.workload == "config2" or args.joint_efforts), mode=args.mode,
                       gather_under_loop=args.gather_under_loop, want_parity=(rank == 0 and not args.no_parity), gather_form=args.gather)
    # The other single-GPU BASELINE.json configurations, measured in the same process (N = 1 default run only):
    # config 3 (65 536 hexapods, all four components of north_star), one GPU's share of config 4 (131 072 octopods) and config 4 at its full size.
    also = []
    if world == 1 and not use_dist and args.workload == "config2" and not args.instances and not args.no_also:
        for name, efforts in (("config2", not primary_efforts), ("config3", False), ("config4", False), ("config4", True), ("rough", False), ("gravity", False), ("gravity3", False)):
            k = max(300, min(args.steps, 1000))   # long enough that first-touch and clock ramp are outside the figure
            gc.collect()
            try:   # the secondary workloads must never cost the run its primary line
                r = run_workload(name, DEFAULT_INSTANCES[name], k, max(30, min(args.warmup, 100)), args.cycles_per_step, args.seed,
                                 fused_probe=not args.no_fused_probe and not efforts, joint_efforts=efforts, mode=args.mode)
                r["steps"] = k
                also.append(also_record(name, r, efforts and name != "config2"))
                if name == "config2" and not efforts:
                    also[-1]["short"] += " no torques"
            except Exception as exc:  # noqa: BLE001
                also.append({"workload": name, "short": SHORT.get(name, name), "error": str(exc)[:200]})
        for name, fn in (("config5", lambda: run_config5(DEFAULT_INSTANCES["config5"], 100, 10, args.seed, want_parity=not args.no_parity)),
                         ("config4full", lambda: run_config4_full(100, 10, args.seed, gather_form="rccl", want_parity=not args.no_parity))):
            gc.collect()
            try:
                r = fn()
                r["short"], r["form"] = SHORT[name], "launch"
                also.append(r)
            except Exception as exc:  # noqa: BLE001
                also.append({"workload": name, "short": SHORT[name], "error": str(exc)[:200]})
    if use_dist:   # every rank: whatever the collective library printed goes out now, the process group ends, and only then rank 0 prints - its line stays the last one
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
        flush_c_stdio()
    if rank == 0:
        cfg = res["config"]
        cfg["short"] = f"BASELINE.json {args.workload}: {n}/GPU {cfg['legs']}x{cfg['dof']}" + (" +joint torques" if "joint-effort" in cfg["workload"] else "")
        out = dict(head, value=res["value"], ms_per_step=res["ms_per_step"], config=cfg, roofline=res["roofline"], parity=res["parity"], also=also)
        if "cpu_baseline" in res:
            out["cpu_baseline"] = res["cpu_baseline"]
        emit(out)
    return 0


if __name__ == "__main__":
    sys.exit(main())

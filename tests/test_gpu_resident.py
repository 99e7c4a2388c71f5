"""GPU (-m gpu): resident mode (shc_engine_resident_*) - the control loop kept on the chip (src/main.cpp:106-131 around
StateController::loop) - against (a) the same cycles through shc_engine_step, byte for byte, and (b) the CPU oracle, with
inputs that change EVERY cycle (the callbacks of every loop iteration deliver something new).
"""
import ctypes as C
import time

import numpy as np
import pytest

from conftest import parity_report
from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine():
    from syropod_highlevel_controller_amd import engine
    if engine.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests must run the native HIP path")
    return engine.BatchEngine


def state_bytes(eng):
    return bytes(memoryview(eng.get_state()).cast("B"))


def velocity_schedule(rng, n, cycles):
    """A different command every cycle: smooth drift + a few hard steps + a stop / restart in the middle."""
    base_l, base_a = rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n)
    out = []
    for c in range(cycles):
        k = 0.6 + 0.4 * np.sin(0.05 * c + np.arange(n))
        lin = base_l * k[:, None] + 0.05 * rng.standard_normal((n, 2))
        ang = base_a * k[::-1] + 0.05 * rng.standard_normal(n)
        if cycles // 2 <= c < cycles // 2 + 40:   # a third of the robots stop and restart
            lin[::3] = 0.0
            ang[::3] = 0.0
        out.append((lin, ang))
    return out


def config3_params():
    p = default_hexapod_params("wave")
    p.admittance_control, p.imu_posing = 1, 1
    p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    return p


def imu_sample(rng, n):
    from scipy.spatial.transform import Rotation as R
    e = np.stack([rng.uniform(-0.15, 0.15, n), rng.uniform(-0.15, 0.15, n), rng.uniform(-3, 3, n)], axis=1)
    q = R.from_euler("xyz", e).as_quat()
    return np.stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]], axis=1), rng.normal(0, 0.05, size=(n, 3))


def force_sample(rng, n, legs):
    return np.stack([rng.normal(0, 1, (n, legs)), rng.normal(0, 1, (n, legs)), rng.uniform(0, 20, (n, legs))], axis=2)


@pytest.mark.parametrize("waves", ["two_waves", "one_wave"])
@pytest.mark.parametrize("case", ["config2", "config3", "octopod", "generic_4x4", "config2_body_posing", "config3_body_posing_inclination",
                                  "config2_joint_efforts", "config3_joint_efforts", "config3_joint_efforts_feed_admittance", "octopod_joint_efforts",
                                  "rough_terrain", "rough_terrain_joint_efforts", "gravity_aligned_octopod", "tip_align_hexapod",
                                  "tip_align_hexapod_joint_efforts"])
def test_resident_is_byte_identical_to_single_cycle_launches(Engine, case, waves):
    """Every cycle gets new inputs.  Engine A: set_* + shc_engine_step(1) per cycle.  Engine B: one resident launch, inputs posted per
    cycle.  q / qd of EVERY cycle (output ring) and the complete state record at the end are equal byte for byte - for the
    two-wavefront (walker / model) pipeline, which these batch sizes get by default, and for one wavefront per robot group.
    *_joint_efforts: measured joint torques are live (shc_engine_set_joint_effort before the loop starts, new torques through the RG_EFFORT
    ring every few cycles) - the kernels with Leg::calculateTipForce (model.cpp:667-708), i.e. what bench.py's headline number runs on;
    *_feed_admittance: use_joint_effort, the estimate drives the admittance (admittance_controller.cpp:30-31).
    rough_terrain*: rough_terrain_mode with tip forces that come and go (touchdown detection runs inside the loop when a force arrives);
    gravity_aligned_octopod: tip rotations + the rotation-constrained applyIK - both as one wavefront per robot group (Leg::applyIK feeds
    back into the stepper there), whatever `waves` asks for; tip_align_hexapod*: gravity_aligned_tips on 3-joint legs - the tip-align pose
    (PoseController::updateTipAlignPose, pose_controller.cpp:849-905) is state of the loop, one wavefront per robot group as well."""
    from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, FEAT_RESIDENT_ONE_WAVE
    rng = np.random.default_rng(11)
    efforts_live = "joint_efforts" in case
    if case in ("config2", "config2_joint_efforts"):
        p, n = default_hexapod_params("tripod"), 333
    elif case == "config2_body_posing":   # joystick body posing: pose inputs and reset modes change while the loop runs (the pose runs on the model wavefront)
        p, n = default_hexapod_params("tripod"), 171
    elif case == "config3_body_posing_inclination":
        p, n = config3_params(), 93
        p.inclination_posing = 1
    elif case in ("config3", "config3_joint_efforts", "config3_joint_efforts_feed_admittance"):
        p, n = config3_params(), 250
        if case.endswith("feed_admittance"):
            p.use_joint_effort = 1
    elif case in ("octopod", "octopod_joint_efforts"):
        p, n = synthetic_octopod_params("ripple", 5, 8), 203
    elif case.startswith("rough_terrain"):
        p, n = default_hexapod_params("tripod"), 287
        p.rough_terrain_mode, p.step_depth = 1, 0.012
    elif case.startswith("tip_align_hexapod"):   # gravity_aligned_tips on 3-joint legs: the tip-align pose (PoseController::updateTipAlignPose) is loop state
        p, n = default_hexapod_params("ripple"), 149
        p.gravity_aligned_tips = 1
    elif case == "gravity_aligned_octopod":
        p, n = synthetic_octopod_params("ripple", 5, 8), 131
        p.gravity_aligned_tips = 1
    else:
        p, n = synthetic_octopod_params("amble", 4, 4), 130
    cycles, depth = 260, 8
    sched = velocity_schedule(rng, n, cycles)
    imus = [imu_sample(rng, n) for _ in range(cycles)] if case.startswith("config3") else None
    forces = [force_sample(rng, n, p.leg_count) if c % 3 == 0 else None for c in range(cycles)] if case.startswith("config3") else None
    if case.startswith("rough_terrain"):   # contact forces that come and go around the touchdown / lift-off thresholds
        def contact():
            f = rng.normal(0, 0.25, (n, 6, 3))
            f[..., 2] += rng.choice([0.0, 0.05, 0.6, 1.5], size=(n, 6), p=[0.3, 0.2, 0.2, 0.3])
            return f
        forces = [contact() if c % 4 == 1 else None for c in range(cycles)]
    posing = "body_posing" in case
    pose_in = [(rng.uniform(-1, 1, (n, 3)) * (rng.random((n, 1)) < 0.7), rng.uniform(-1, 1, (n, 3)) * (rng.random((n, 1)) < 0.7)) if c % 4 == 0 else None
               for c in range(cycles)] if posing else None
    resets = [rng.integers(0, 6, n).astype(np.int32) if c % 37 == 5 else (np.zeros(n, dtype=np.int32) if c % 37 == 11 else None) for c in range(cycles)] if posing else None
    nje = p.leg_count * max(p.leg_dof[l] for l in range(p.leg_count))
    efforts = [rng.normal(0, 0.5, (n, nje)) if (efforts_live and c % 5 == 2) else None for c in range(cycles)]
    a, b = Engine(p, n), Engine(p, n)
    if posing:
        for e in (a, b):   # (the manual-pose group of the state is live from the first pose input on)
            e.set_pose_input(np.zeros((n, 3)), np.zeros((n, 3)))
    if efforts_live:
        e0 = rng.normal(0, 0.5, (n, nje))
        for e in (a, b):   # (the tip-force estimate is live from the first torque on: the kernels with calculateTipForce)
            e.set_joint_effort(e0)
    if waves == "one_wave":
        b.set_features(FEAT_DEFAULT | FEAT_RESIDENT_ONE_WAVE)
    for e in (a, b):   # some history before the resident run starts
        e.set_velocity(*sched[0])
        e.step(37)
    qa = []
    for c in range(cycles):
        a.set_velocity(*sched[c])
        if imus:
            a.set_imu(*imus[c])
        if forces and forces[c] is not None:
            a.set_tip_force(forces[c])
        if posing:
            if pose_in[c] is not None:
                a.set_pose_input(*pose_in[c])
            if resets[c] is not None:
                a.set_pose_reset_mode(resets[c])
        if efforts[c] is not None:
            a.set_joint_effort(efforts[c])
        a.step(1)
        qa.append(a.joints())
    a.synchronize()
    b.resident_begin(ring_depth=depth, max_cycles=cycles + 10)
    with pytest.raises(Exception, match="resident mode"):
        b.step(1)          # everything else is refused while the loop owns the engine
    for c0 in range(0, cycles, depth - 1):   # post up to ring_depth - 1 cycles ahead, release them, read every cycle's output
        c1 = min(cycles, c0 + depth - 1)
        for c in range(c0, c1):
            kw = {"velocity": sched[c]}
            if imus:
                kw["imu"] = imus[c]
            if forces and forces[c] is not None:
                kw["tip_force"] = forces[c]
            if posing:
                if pose_in[c] is not None:
                    kw["pose_input"] = pose_in[c]
                if resets[c] is not None:
                    kw["pose_reset_mode"] = resets[c]
            if efforts[c] is not None:
                kw["joint_effort"] = efforts[c]
            if case in ("config2", "config3_body_posing_inclination") and c == c1 - 1:
                kw["publish"] = True     # the last post of the group releases the group: post + doorbell in one kernel launch
            assert b.resident_post(**kw) == c
        if case not in ("config2", "config3_body_posing_inclination"):
            b.resident_publish(c1 - c0)
        b.resident_wait(c1)
        for c in range(c0, c1):
            q, qd = b.resident_joints(c)
            assert np.array_equal(q, qa[c][0]) and np.array_equal(qd, qa[c][1]), f"cycle {c}"
    assert b.resident_end() == cycles
    assert state_bytes(a) == state_bytes(b)
    if efforts_live:   # (the estimate is really being evaluated: a dead filter would be byte-identical too)
        assert np.abs(a.leg_state()["tip_force"]).max() > 1e-3
    if case.startswith("tip_align"):   # (and the tip-align pose is moving the body)
        from syropod_highlevel_controller_amd.params import InstanceState
        st = np.frombuffer(b.get_state(), dtype=np.dtype(InstanceState))
        assert max(np.abs(st["tip_align_pose"][:, :3]).max(), np.abs(st["origin_tip_align_pose"][:, :3]).max()) > 1e-4
    # ... and the engines go on identically through ordinary launches (held inputs were carried over)
    for e in (a, b):
        e.step(25)
    assert state_bytes(a) == state_bytes(b)
    qb = b.joints()
    assert np.array_equal(qb[0], a.joints()[0])
    a.close()
    b.close()


def test_resident_with_velocities_changing_every_cycle_matches_the_oracle(Engine):
    """BASELINE.json config 2's path with a new velocity command every cycle, free-running against the oracle: 1e-6 rad."""
    p, n, cycles = default_hexapod_params("tripod"), 200, 400
    rng = np.random.default_rng(5)
    sched = velocity_schedule(rng, n, cycles)
    eng, ob = Engine(p, n), OracleBatch(p, n)
    eng.resident_begin(ring_depth=32, max_cycles=cycles)
    worst = 0.0
    for c0 in range(0, cycles, 25):
        for c in range(c0, c0 + 25):
            eng.resident_post(velocity=sched[c])
        eng.resident_publish(25)
        for c in range(c0, c0 + 25):
            ob.set_velocity(*sched[c])
            ob.step(1, 4)
        eng.resident_wait(c0 + 25)
        q, _ = eng.resident_joints(c0 + 24)
        worst = max(worst, float(np.abs(q - ob.joints()[0]).max()))
    assert eng.resident_end() == cycles
    parity_report(f"resident mode, velocities changing every cycle, {n} hexapods x {cycles} cycles: max |dq| = {worst:.2e} rad")
    assert worst <= 1e-6
    _, _, ws = eng.body_state()
    assert np.array_equal(ws, ob.body_state()[2])
    eng.close()


@pytest.mark.parametrize("case", ["config2", "config3"])
def test_resident_full_size_with_joint_efforts_matches_the_oracle(Engine, case):
    """bench.py's headline configuration as it is benchmarked - 4 096 hexapods in resident mode on the two-wavefront kernel with measured
    joint torques live (shc_resident2_kernel<6, 3, C2 | F_TIPF>; config3: <6, 3, C3 | F_TIPF>) - free-running against the oracle on a
    128-instance slice spread over the batch (first / middle / last wavefronts): joints within 1e-6 rad, the tip-force estimate within
    1e-9 N, new torques and a new velocity command every few cycles through the input rings."""
    n, cycles, burst = 4096, 300, 20
    p = default_hexapod_params("tripod") if case == "config2" else config3_params()
    rng = np.random.default_rng(21)
    sel = np.concatenate([np.arange(0, 48), np.arange(2040, 2080), np.arange(n - 40, n)])
    lin, ang = rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n)
    eff0 = rng.normal(0, 0.5, (n, 18))
    # config 3's U(0, 20) N tip forces press legs into their joint limits, where the reference's one-step DLS map is expanding
    # (tests/test_gpu_parity.py header): a twin oracle with inputs perturbed by 1e-13 tells which reference trajectories are
    # well-posed; the bar is evaluated on those and the fraction is asserted and reported.  config 2: every instance.
    eng, ob = Engine(p, n), OracleBatch(p, len(sel))
    tw = OracleBatch(p, len(sel)) if case == "config3" else None
    eng.set_joint_effort(eff0)
    for o in (ob, tw):
        if o is not None:
            o.set_joint_effort(eff0[sel])
    if case == "config3":
        imu = imu_sample(rng, n)
        f0 = force_sample(rng, n, 6)
        eng.set_imu(*imu)
        eng.set_tip_force(f0)
        for o in (ob, tw):
            o.set_imu(imu[0][sel], imu[1][sel])
        ob.set_tip_force(f0[sel])
        tw.set_tip_force(f0[sel] * (1 + 1e-13))
    eng.resident_begin(ring_depth=8, max_cycles=cycles)
    worst_q = worst_tf = 0.0
    well = np.ones(len(sel), dtype=bool)
    for c0 in range(0, cycles, burst):
        k = 1.0 - 0.3 * np.sin(0.01 * c0 + np.arange(n))
        v = (lin * k[:, None], ang * k)
        eff = rng.normal(0, 0.5, (n, 18))
        kw = {"velocity": v, "joint_effort": eff}
        if case == "config3":
            f = force_sample(rng, n, 6)
            kw["tip_force"] = f
            ob.set_tip_force(f[sel])
            tw.set_tip_force(f[sel] * (1 + 1e-13))
        assert eng.resident_post(**kw) == c0
        eng.resident_publish(burst)
        for o in (ob, tw):
            if o is not None:
                o.set_velocity(v[0][sel] * (1 + 1e-13 * (o is tw)), v[1][sel])
                o.set_joint_effort(eff[sel])
                o.step(burst, 4)
        if tw is not None:
            well &= np.abs(ob.joints()[0] - tw.joints()[0]).max(axis=1) <= 1e-11
        eng.resident_wait(c0 + burst)
        q, _ = eng.resident_joints(c0 + burst - 1)
        worst_q = max(worst_q, float(np.abs(q[sel] - ob.joints()[0])[well].max()))
    assert eng.resident_end() == cycles
    tf_gpu, tf_cpu = eng.leg_state()["tip_force"][sel], ob.leg_state()["tip_force"]
    worst_tf = float(np.abs(tf_gpu - tf_cpu)[well].max())
    parity_report(f"resident mode at bench size with joint efforts live [{case}], {n} hexapods x {cycles} cycles, {len(sel)} instances against the oracle: "
                  f"max |dq| = {worst_q:.2e} rad, max |d tip_force_calculated| = {worst_tf:.2e} N (estimate up to {np.abs(tf_cpu).max():.2f} N) over "
                  f"{'all of them' if tw is None else f'the {well.mean():.0%} whose reference trajectory is well-posed'}")
    assert np.abs(tf_cpu).max() > 1e-2
    assert well.mean() >= 0.85
    assert worst_q <= 1e-6 and worst_tf <= (1e-9 if tw is None else 2e-4)   # (the filter remembers earlier joint differences: ~100 N / rad)
    assert np.array_equal(eng.body_state()[2][sel][well], ob.body_state()[2][well])
    eng.close()


def test_resident_held_inputs_and_bare_publishes(Engine):
    """Cycles that nothing was posted for run with the inputs held (a callback that did not fire); a group posted once stays in force."""
    p, n = config3_params(), 128
    rng = np.random.default_rng(3)
    lin, ang = rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n)
    f0, f1 = force_sample(rng, n, 6), force_sample(rng, n, 6)
    a, b = Engine(p, n), Engine(p, n)
    for e in (a, b):
        e.set_tip_force(f0)
    a.set_velocity(lin, ang)
    a.step(50)
    a.set_tip_force(f1)
    a.step(70)
    a.set_velocity(lin * 0.5, ang)
    a.step(30)
    b.resident_begin(ring_depth=4, max_cycles=1000)
    b.resident_post(velocity=(lin, ang))
    b.resident_publish(50)           # 1 posted + 49 held
    b.resident_post(tip_force=f1)
    b.resident_publish(70)
    b.resident_post(velocity=(lin * 0.5, ang))
    b.resident_publish(30)
    b.resident_wait(150)
    assert b.resident_status() == (150, 150, True)
    assert b.resident_end() == 150
    assert state_bytes(a) == state_bytes(b)
    a.step(10)
    b.step(10)                       # the held inputs (velocity, second force set) are the engine's inputs from here on
    assert state_bytes(a) == state_bytes(b)
    a.close()
    b.close()


def test_resident_stream_ordered_read_of_a_cycle_still_to_come(Engine):
    """shc_engine_resident_get_joint_state_async: the copy of a cycle's joints into device buffers is queued on the engine's stream BEFORE
    the cycle has been published; work queued behind it (here a device-to-device copy, in bench.py the RCCL all-gather) sees the cycle's
    joints, with no host wait in between.  A read whose cycle never comes gives up and resident_end reports it."""
    import torch
    p, n = default_hexapod_params("tripod"), 300
    rng = np.random.default_rng(8)
    lin, ang = rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        e = Engine(p, n, stream=stream.cuda_stream)
        e.set_velocity(lin, ang)
        e.step(40)
        e.resident_begin(ring_depth=8, max_cycles=500)
        q = torch.full((n, 18), float("nan"), dtype=torch.float64, device="cuda")
        qd = torch.full((n, 18), float("nan"), dtype=torch.float64, device="cuda")
        e.resident_publish(10)
        e.resident_joints_async(10 + 24, q.data_ptr(), qd.data_ptr())     # cycle 34: not published yet
        behind = torch.empty_like(q)
        behind.copy_(q, non_blocking=True)                                  # queued behind the read on the same stream
        time.sleep(0.05)
        assert not stream.query()                                           # the stream is waiting for that cycle on the device
        for _ in range(25):
            e.resident_publish(1)
        stream.synchronize()
        want_q, want_qd = e.resident_joints(34)
        assert np.array_equal(behind.cpu().numpy(), want_q) and np.array_equal(qd.cpu().numpy(), want_qd)
        assert e.resident_end() == 35
        # ... a read of a cycle that is never published: bounded, reported
        e.resident_begin(ring_depth=8, max_cycles=100)
        e.resident_publish(3)
        e.resident_joints_async(50, q.data_ptr(), 0, timeout_ms=200)
        stream.synchronize()
        with pytest.raises(RuntimeError):
            e.resident_end()
        e.step(5)                                                           # the engine is usable again
        e.synchronize()
        e.close()


def test_resident_bounds(Engine):
    """Every device-side wait is bounded: max_cycles, the idle timeout, and batches that do not fit are refused."""
    from syropod_highlevel_controller_amd.engine import ShcError
    p = default_hexapod_params("tripod")
    big = Engine(p, 40000)           # 4 000 waves: more than the chip holds at once
    with pytest.raises(ShcError, match="co-resident"):
        big.resident_begin()
    big.close()
    e = Engine(p, 64)
    e.resident_begin(ring_depth=4, max_cycles=20, idle_timeout_ms=300)
    with pytest.raises(ShcError, match="max_cycles"):
        e.resident_publish(21)
    e.resident_publish(20)
    e.resident_wait(20)
    time.sleep(0.05)
    assert e.resident_status() == (20, 20, False)   # the loop has reached its bound and left by itself
    with pytest.raises(ShcError, match="stopped by itself"):
        e.resident_publish(1)
    assert e.resident_end() == 20                    # ... having run everything that was published
    assert e.resident_status()[2] is False
    e.step(1)                        # ... and the engine is usable again, 20 cycles on
    ref = Engine(p, 64)
    ref.step(21)
    assert state_bytes(ref) == state_bytes(e)
    # idle timeout: nobody rings the doorbell
    e.resident_begin(ring_depth=4, max_cycles=1000, idle_timeout_ms=200)
    e.resident_publish(5)
    time.sleep(0.6)
    with pytest.raises(ShcError, match="stopped by itself"):
        e.resident_publish(1)
    with pytest.raises(ShcError, match="idle timeout"):
        e.resident_end()
    ref.step(5)
    assert state_bytes(ref) == state_bytes(e)
    e.close()
    ref.close()


@pytest.mark.parametrize("n", [5100, 5110, 19830])
def test_resident_at_the_kernel_boundaries(Engine, n):
    """5 100 hexapods = 510 robot groups = 255 two-wavefront workgroups + the relay: the largest batch the two-wavefront pipeline takes on
    256 compute units; 5 110 falls to one wavefront per robot group; 19 830 = 1 983 wavefronts + the relay is the largest batch resident mode
    takes (two per SIMD, less one compute unit's worth per XCD, which stays free for the kernels that post inputs and read the output ring).  Each runs cycles with inputs changing and ends byte-identical to single launches."""
    p = default_hexapod_params("tripod")
    rng = np.random.default_rng(n)
    a, b = Engine(p, n), Engine(p, n)
    lin, ang = rng.uniform(-0.7, 0.7, (n, 2)), rng.uniform(-1, 1, n)
    for e in (a, b):
        e.set_velocity(lin, ang)
        e.step(20)
    sched = [(lin * (1.0 - 0.02 * c), ang * (0.5 + 0.01 * c)) for c in range(30)]
    for v in sched:        # (first: a loop that fills every wave slot of the chip leaves none for another engine's launches)
        a.set_velocity(*v)
        a.step(1)
    a.synchronize()
    b.resident_begin(ring_depth=4, max_cycles=64)
    for v in sched:
        b.resident_post(velocity=v, publish=True)
    b.resident_wait(30)
    q, _ = b.resident_joints(29)
    assert np.array_equal(q, a.joints()[0])
    assert b.resident_end() == 30
    assert state_bytes(a) == state_bytes(b)
    a.close()
    b.close()


def test_collective_shaped_kernel_next_to_a_live_loop(Engine, tmp_path):
    """What `bench.py --gpus N --mode resident --gather-under-loop` relies on and a one-GPU box cannot rehearse with RCCL itself (two ranks
    on one device are refused): a kernel of the shape of a collective - 32 workgroups x 512 threads, 40 KiB of LDS each, every block
    spinning on every other block's flags, i.e. all of them must be co-resident - queued on the engine's stream behind
    shc_engine_resident_get_joint_state_async WHILE the persistent loop of bench.py's headline batch (4 096 hexapods, two-wavefront
    kernel, joint efforts live) is alive and stepping.  It must be scheduled next to the loop, finish all its rounds, and the loop must
    keep running; every wait is bounded on the device, so a starved probe is a test failure, not a hung GPU.  (bench.py's default for
    N > 1 does not depend on this: it ends the loop before the gather.)"""
    import subprocess
    import os
    import ctypes as C
    from syropod_highlevel_controller_amd import engine
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "collective_shape_probe.hip")
    so = str(tmp_path / "libcollective_shape_probe.so")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, src], cwd=str(tmp_path))
    P = C.CDLL(so)
    P.probe_prepare.argtypes = [C.c_int, C.c_int64, C.c_int]
    P.probe_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    P.probe_stream_wait.argtypes = [C.c_void_p, C.c_int]
    P.probe_result.argtypes = [C.POINTER(C.c_uint64)]
    L = engine.lib()
    stream = C.c_void_p()
    assert L.shc_stream_create(0, C.byref(stream)) == 0
    p, n = default_hexapod_params("tripod"), 4096
    rng = np.random.default_rng(2)
    eng = Engine(p, n, stream=stream.value)
    eng.set_velocity(rng.uniform(-0.7, 0.7, (n, 2)), rng.uniform(-1, 1, n))
    eng.set_joint_effort(rng.normal(0, 0.5, (n, 18)))
    eng.step(50)
    eng.synchronize()
    blocks, threads, lds, rounds, n_doubles = 32, 512, 40 * 1024, 200, 1 << 20
    import torch   # (device buffers for the stream-ordered read)
    q = torch.zeros(n * 18, dtype=torch.float64, device="cuda")

    def run_probe(stepping):
        assert P.probe_prepare(blocks, n_doubles, lds) == 0
        if stepping:   # (a) the loop is busy stepping for the whole life of the probe: ~1.2 s of cycles released at once
            total = 400000
            eng.resident_begin(ring_depth=8, max_cycles=total + 8)
            eng.resident_publish(total)
            eng.resident_wait(100)
        else:          # (b) bench.py's sequence: the read of the region's LAST cycle, the collective behind it; the loop then polls its gate
            total = 3000
            eng.resident_begin(ring_depth=8, max_cycles=total + 8)
            eng.resident_publish(total)
            eng.resident_joints_async(total - 1, q.data_ptr())
        assert P.probe_launch(stream, blocks, threads, lds, rounds, 3000) == 0
        drained = P.probe_stream_wait(stream, 8000)
        _, done_mid, running = eng.resident_status()
        ran = eng.resident_end()                   # (stops at the last published cycle)
        out = (C.c_uint64 * 4)()
        assert P.probe_result(out) == 0
        rounds_done, timed_out, _, ticks = [int(x) for x in out]
        parity_report(f"collective-shaped kernel ({blocks} x {threads} threads, {lds // 1024} KiB LDS, all-to-all flag spins) on the engine's stream next to the live "
                      f"loop of {n} hexapods ({'stepping' if stepping else 'behind the stream-ordered read of the last cycle, loop polling its gate'}): "
                      f"{rounds_done}/{rounds} rounds, {timed_out} blocks timed out, {ticks / 100:.0f} us in the kernel; loop alive when it finished: {running} "
                      f"({done_mid} of {total} cycles done)")
        assert drained == 1, "the engine's stream did not drain: the probe (or the read before it) was never scheduled next to the loop"
        assert timed_out == 0 and rounds_done == rounds
        assert running and ran == total
        return done_mid, total

    done_mid, total = run_probe(stepping=True)
    assert 100 < done_mid < total                   # the loop was still stepping when the probe had finished
    done_mid, total = run_probe(stepping=False)
    assert done_mid == total
    assert torch.isfinite(q).all() and float(q.abs().max()) > 0
    eng.close()
    L.shc_stream_destroy(0, stream)


@pytest.mark.parametrize("waves", ["two_waves", "one_wave"])
@pytest.mark.parametrize("case", ["config2_joint_efforts", "config3_joint_efforts", "octopod", "generic_4x4"])
def test_resident_direct_posts_are_byte_identical_to_single_cycle_launches(Engine, case, waves):
    """shc_engine_resident_bind_inputs + shc_cycle_inputs.direct: a cycle's inputs posted and released WITHOUT a kernel launch - one
    16-byte record in host-mapped memory that the relay wavefront turns into the cycle's header; the workers read their robots'
    velocity / IMU / tip force / joint efforts straight from the caller's device arrays (bound before the loop starts).  New inputs EVERY cycle from two alternating sets of device arrays (the set of cycle c is rewritten only
    after cycle c has completed), mixed with ordinary ring posts and bare publishes in between; q / qd of every cycle and the final
    state record equal the same cycles through set_* + shc_engine_step(1), byte for byte; the inputs of the last direct post stay in
    force afterwards (per-leg inputs are carried into the engine's own planes)."""
    import torch
    from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, FEAT_RESIDENT_ONE_WAVE
    rng = np.random.default_rng(31)
    efforts_live = "joint_efforts" in case
    if case == "config2_joint_efforts":
        p, n = default_hexapod_params("tripod"), 333
    elif case == "config3_joint_efforts":
        p, n = config3_params(), 250
    elif case == "octopod":
        p, n = synthetic_octopod_params("ripple", 5, 8), 203
    else:
        p, n = synthetic_octopod_params("amble", 4, 4), 130
    cycles, depth = 200, 8
    legs, dof = p.leg_count, p.leg_dof[0]
    sched = velocity_schedule(rng, n, cycles)
    with_imu = case.startswith("config3")
    imus = [imu_sample(rng, n) if c % 2 == 0 else None for c in range(cycles)] if with_imu else [None] * cycles
    forces = [force_sample(rng, n, legs) if c % 3 == 0 else None for c in range(cycles)] if with_imu else [None] * cycles
    efforts = [rng.normal(0, 0.5, (n, legs * dof)) if (efforts_live and c % 5 == 1) else None for c in range(cycles)]
    a, b = Engine(p, n), Engine(p, n)
    if efforts_live:
        e0 = rng.normal(0, 0.5, (n, legs * dof))
        for e in (a, b):
            e.set_joint_effort(e0)
    if waves == "one_wave":
        b.set_features(FEAT_DEFAULT | FEAT_RESIDENT_ONE_WAVE)
    for e in (a, b):
        e.set_velocity(*sched[0])
        e.step(23)
    # which cycles are posted how: most of them direct, some through the rings, some not at all (inputs held)
    kind = ["direct"] * cycles
    for c in range(cycles):
        if c % 17 == 5:
            kind[c] = "ring"
        elif c % 23 == 11:
            kind[c] = "none"
    # per-leg arrays of a bound set stay in force until another post of that group replaces them: a direct post may only carry a per-leg
    # group from the set that is NOT in force (the host loop of a real node alternates sets for exactly that reason)
    in_force = {"f": None, "e": None}
    for c in range(cycles):
        for key, plan in (("f", forces), ("e", efforts)):
            if plan[c] is None or kind[c] == "none":
                continue
            if kind[c] == "ring":
                in_force[key] = None
            elif in_force[key] == c % 2:
                plan[c] = None
            else:
                in_force[key] = c % 2
    qa = []
    for c in range(cycles):
        if kind[c] != "none":
            a.set_velocity(*sched[c])
            if imus[c] is not None:
                a.set_imu(*imus[c])
            if forces[c] is not None:
                a.set_tip_force(forces[c])
            if efforts[c] is not None:
                a.set_joint_effort(efforts[c])
        a.step(1)
        qa.append(a.joints())
    a.synchronize()
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    sets = [{"lin": torch.empty((n, 2), dtype=torch.float64, device="cuda"), "ang": torch.empty(n, dtype=torch.float64, device="cuda"),
             "q": torch.empty((n, 4), dtype=torch.float64, device="cuda"), "w": torch.empty((n, 3), dtype=torch.float64, device="cuda"),
             "f": torch.empty((n, legs, 3), dtype=torch.float64, device="cuda"), "e": torch.empty((n, legs * dof), dtype=torch.float64, device="cuda")} for _ in range(2)]
    for k, s_ in enumerate(sets):
        b.resident_bind_inputs(k, velocity=(s_["lin"].data_ptr(), s_["ang"].data_ptr()), imu=(s_["q"].data_ptr(), s_["w"].data_ptr()) if with_imu else None,
                               tip_force=s_["f"].data_ptr() if with_imu else None, joint_effort=s_["e"].data_ptr() if efforts_live else None)
    b.resident_begin(ring_depth=depth, max_cycles=cycles + 10)
    checked = 0
    for c in range(cycles):
        if kind[c] == "direct":
            k = c % 2
            s_ = sets[k]
            b.resident_wait(c)                  # every earlier cycle has completed: velocity / IMU of this set were taken when ITS cycle started
            s_["lin"].copy_(dev(sched[c][0]))
            s_["ang"].copy_(dev(sched[c][1]))
            kw = {"velocity": True}
            if imus[c] is not None:
                s_["q"].copy_(dev(imus[c][0]))
                s_["w"].copy_(dev(imus[c][1]))
                kw["imu"] = True
            if forces[c] is not None:           # (the plan above made sure this set's per-leg arrays are not the ones in force)
                s_["f"].copy_(dev(forces[c]))
                kw["tip_force"] = True
            if efforts[c] is not None:
                s_["e"].copy_(dev(efforts[c]))
                kw["joint_effort"] = True
            torch.cuda.current_stream().synchronize()   # the arrays are complete before the post (the contract of a direct post); a STREAM
            #                                             synchronisation - a device-wide one would wait for the resident loop itself
            assert b.resident_post(direct=k, **kw) == c
        elif kind[c] == "ring":
            kw = {"velocity": sched[c]}
            if imus[c] is not None:
                kw["imu"] = imus[c]
            if forces[c] is not None:
                kw["tip_force"] = forces[c]
            if efforts[c] is not None:
                kw["joint_effort"] = efforts[c]
            assert b.resident_post(publish=(c % 2 == 0), **kw) == c
            if c % 2:
                b.resident_publish(1)
        else:
            b.resident_publish(1)
        if c % 5 == 4 or c == cycles - 1:
            b.resident_wait(c + 1)
            for cc in range(max(checked, c + 1 - (depth - 1)), c + 1):
                q, qd = b.resident_joints(cc)
                assert np.array_equal(q, qa[cc][0]) and np.array_equal(qd, qa[cc][1]), f"cycle {cc} ({kind[cc]})"
            checked = c + 1
    for s_ in sets:                              # velocity / IMU arrays are only borrowed until their cycle has started; the per-leg arrays in
        for key in ("lin", "ang", "q", "w"):     # force are read until the loop ends (and carried into the engine's planes then)
            s_[key].fill_(float("nan"))
    torch.cuda.current_stream().synchronize()
    assert b.resident_end() == cycles
    assert state_bytes(a) == state_bytes(b)
    for e in (a, b):                             # the inputs of the last posts stay in force through ordinary launches
        e.step(25)
    assert state_bytes(a) == state_bytes(b)
    assert np.isfinite(b.joints()[0]).all()
    a.close()
    b.close()

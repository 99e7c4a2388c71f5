"""GPU (-m gpu): robots whose legs differ in DOF (Parameters::leg_DOF is per leg, parameters_and_states.h:298; BASELINE.json config 5 "3-5 DOF
per leg").  The engine runs such a robot on the kernels of its longest leg and pads the shorter legs behind their tips with locked
zero-length joints; the oracle runs every leg with its own joint count, as the reference does."""
import numpy as np
import pytest

from conftest import parity_report
from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import synthetic_mixed_dof_params
from syropod_highlevel_controller_amd.engine import BatchEngine, ShcError
from syropod_highlevel_controller_amd.params import FEAT_DEFAULT
from test_gpu_teacher_forced import as_np, compare_records

pytestmark = pytest.mark.gpu


def padded(q_packed, p):
    """Oracle joints [n][sum of DOF] -> the engine's [n][legs][longest DOF] (padded joints 0)."""
    n, L = q_packed.shape[0], p.leg_count
    D = max(p.leg_dof[l] for l in range(L))
    out, k = np.zeros((n, L, D)), 0
    for l in range(L):
        d = p.leg_dof[l]
        out[:, l, :d] = q_packed[:, k:k + d]
        k += d
    return out.reshape(n, L * D)


@pytest.mark.parametrize("case", ["ripple-353 454", "tripod-admittance-efforts", "wave-imu"])
def test_mixed_dof_robot_teacher_forced(case):
    """Every cycle from the oracle's complete state, every field of the record compared for every instance (1e-12 rad): a hexapod with
    3-, 5- and 4-joint legs; admittance driven by the tip-force estimate from measured joint torques (the padded joints' angular Jacobian
    columns are masked); IMU posing."""
    gait = case.split("-")[0]
    p = synthetic_mixed_dof_params(gait, (3, 5, 4, 3, 5, 4))
    if "admittance" in case:
        p.admittance_control, p.use_joint_effort = 1, 1
    if "imu" in case:
        p.imu_posing = 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    n, L = 40, p.leg_count
    D = max(p.leg_dof[l] for l in range(L))
    rng = np.random.default_rng(77)
    eng, ob = BatchEngine(p, n), OracleBatch(p, n)
    assert eng.dof == 5
    lin, ang = rng.uniform(-0.6, 0.6, (n, 2)), rng.uniform(-0.8, 0.8, n)
    worst = 0.0
    for c in range(260):
        if c == 170:
            lin[::2], ang[::2] = 0.0, 0.0   # half of the robots stop
        if c % 10 == 0:
            if p.use_joint_effort:
                eff = rng.normal(0, 0.5, (n, L, D))
                for l in range(L):
                    eff[:, l, p.leg_dof[l]:] = 0.0
                eng.set_joint_effort(eff.reshape(n, -1))
                ob.set_joint_effort(np.concatenate([eff[:, l, :p.leg_dof[l]] for l in range(L)], axis=1))
            if p.imu_posing:
                from scipy.spatial.transform import Rotation as R
                e = np.stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.1, 0.1, n), rng.uniform(-1, 1, n)], axis=1)
                q = R.from_euler("xyz", e).as_quat()
                quat, gyro = np.stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]], axis=1), rng.normal(0, 0.03, (n, 3))
                for o in (eng, ob):
                    o.set_imu(quat, gyro)
        for o in (eng, ob):
            o.set_velocity(lin, ang)
        eng.set_state(ob.get_state())
        eng.step(1)
        eng.synchronize()
        ob.step(1, 1)
        d = float(np.abs(eng.joints()[0] - padded(ob.joints()[0], p)).max())
        worst = max(worst, d)
        assert d < 1e-12, (c, d)
        compare_records(p, FEAT_DEFAULT, as_np(eng.get_state()), as_np(ob.get_state()), tol_q=1e-12)
        assert np.array_equal(eng.body_state()[2], ob.body_state()[2])
    q = eng.joints()[0].reshape(n, L, D)
    for l in range(L):
        assert (q[:, l, p.leg_dof[l]:] == 0.0).all()   # the padded joints never move
    parity_report(f"[mixed DOF {case}] legs of 3 / 5 / 4 joints in one robot, {n} instances x 260 cycles teacher-forced: max |dq| = {worst:.2e} rad")


def test_mixed_dof_robot_free_running():
    """... and free-running from each side's own init chain, walk - stop - walk, where the reference trajectory is well-posed."""
    p = synthetic_mixed_dof_params("ripple", (3, 5, 4, 3, 5, 4))
    p.time_to_start = 2.0          # 100 start-up steps: the redundant legs' start-up configuration is reproducible there
    n = 64
    rng = np.random.default_rng(5)
    eng, ob = BatchEngine(p, n), OracleBatch(p, n)
    lin, ang = rng.uniform(-0.6, 0.6, (n, 2)), rng.uniform(-0.8, 0.8, n)
    worst = 0.0
    for c0, (l_, a_) in enumerate([(lin, ang), (lin * 0, ang * 0), (-lin, ang)]):
        for o in (eng, ob):
            o.set_velocity(l_, a_)
        for _ in range(3):
            eng.step(40)
            ob.step(40, 8)
            d = np.abs(eng.joints()[0] - padded(ob.joints()[0], p)).max(axis=1)
            worst = max(worst, float(np.median(d)))
            assert np.isfinite(eng.joints()[0]).all()
            assert np.array_equal(eng.body_state()[2], ob.body_state()[2])
            assert (d < 1e-6).mean() >= 0.9, (c0, np.sort(d)[-5:])
    parity_report(f"[mixed DOF free-running] 3 / 5 / 4-joint legs, {n} instances x 360 cycles: median |dq| over instances <= {worst:.2e} rad, >= 90 % within 1e-6")


@pytest.mark.parametrize("dofs", [(3, 5, 4, 3, 5, 4), (5, 3, 4, 5, 3, 4)])
def test_mixed_dof_robot_with_gravity_aligned_tips(dofs):
    """gravity_aligned_tips on a robot whose legs differ in DOF - one bin of BASELINE.json config 5 with the parameter set.  The reference
    decides per leg: legs of more than 3 joints get the identity tip rotation and the rotation-constrained applyIK (walk_controller.cpp:37,
    :1195, model.cpp:880-900), and LEG 0's joint count decides whether PoseController::updateTipAlignPose runs - over all legs, each with
    the vector from its tip to the last joint it really has (pose_controller.cpp:849, :1024-1088).  (3, 5, 4, ...): leg 0 has 3 joints - the
    tip-align pose AND tip rotations on the four longer legs; (5, 3, 4, ...): rotations only.  Teacher-forced, every field of the record
    for every instance, then free-running where the reference trajectory is well-posed."""
    p = synthetic_mixed_dof_params("ripple", dofs)
    p.gravity_aligned_tips = 1
    p.time_to_start = 2.0
    n, L = 36, p.leg_count
    rng = np.random.default_rng(101)
    eng, ob = BatchEngine(p, n), OracleBatch(p, n)
    lin, ang = rng.uniform(-0.6, 0.6, (n, 2)), rng.uniform(-0.8, 0.8, n)
    worst = 0.0
    for c in range(240):
        if c == 150:
            lin[::2], ang[::2] = 0.0, 0.0
        for o in (eng, ob):
            o.set_velocity(lin, ang)
        eng.set_state(ob.get_state())
        eng.step(1)
        eng.synchronize()
        ob.step(1, 1)
        d = float(np.abs(eng.joints()[0] - padded(ob.joints()[0], p)).max())
        worst = max(worst, d)
        assert d < 1e-11, (c, d)
        compare_records(p, FEAT_DEFAULT, as_np(eng.get_state()), as_np(ob.get_state()), tol_q=1e-11)
        assert np.array_equal(eng.body_state()[2], ob.body_state()[2])
    pose = eng.body_state()[0]
    if dofs[0] <= 3:   # the tip-align pose has moved the body sideways at some point (it is not the plain walk-plane pose)
        st = as_np(eng.get_state())
        assert np.abs(st["tip_align_pose"][:, :3]).max() > 1e-6 or np.abs(st["origin_tip_align_pose"][:, :3]).max() > 1e-6
    assert np.isfinite(pose).all()
    # free-running from each side's own init chain
    eng2, ob2 = BatchEngine(p, n), OracleBatch(p, n)
    lin2 = rng.uniform(-0.5, 0.5, (n, 2))
    for o in (eng2, ob2):
        o.set_velocity(lin2, ang)
    good = 1.0
    for _ in range(4):
        eng2.step(40)
        ob2.step(40, 8)
        d = np.abs(eng2.joints()[0] - padded(ob2.joints()[0], p)).max(axis=1)
        good = min(good, float((d < 1e-6).mean()))
        assert np.array_equal(eng2.body_state()[2], ob2.body_state()[2])
    assert good >= 0.85, good
    parity_report(f"[mixed DOF {dofs} + gravity_aligned_tips] {n} instances x 240 cycles teacher-forced: max |dq| = {worst:.2e} rad; free-running 160 cycles: "
                  f"{good:.0%} of the instances within 1e-6 rad")


def test_mixed_dof_robot_in_resident_mode():
    """... and in resident mode (the generic two-wavefront loop): byte-identical to single launches with inputs changing every cycle."""
    p = synthetic_mixed_dof_params("ripple", (3, 5, 4, 3, 5, 4))
    n = 150
    rng = np.random.default_rng(9)
    a, b = BatchEngine(p, n), BatchEngine(p, n)
    lin, ang = rng.uniform(-0.6, 0.6, (n, 2)), rng.uniform(-0.8, 0.8, n)
    for e in (a, b):
        e.set_velocity(lin, ang)
        e.step(30)
    sched = [(lin * (1.0 - 0.01 * c), ang * np.cos(0.05 * c)) for c in range(120)]
    for v in sched:
        a.set_velocity(*v)
        a.step(1)
    a.synchronize()
    b.resident_begin(ring_depth=8, max_cycles=200)
    for v in sched:
        b.resident_post(velocity=v, publish=True)
    b.resident_wait(len(sched))
    assert np.array_equal(b.resident_joints(len(sched) - 1)[0], a.joints()[0])
    assert b.resident_end() == len(sched)
    assert bytes(memoryview(a.get_state()).cast("B")) == bytes(memoryview(b.get_state()).cast("B"))

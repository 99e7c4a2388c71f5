"""GPU: manual leg manipulation - StateController::legStateToggle + PoseController::poseForLegManipulation (state_controller.cpp:
541-646, pose_controller.cpp:561-611) and WalkController::updateManual (walk_controller.cpp:652-744) - against the oracle."""
import numpy as np
import pytest

from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
from syropod_highlevel_controller_amd.params import FEAT_DEFAULT
from test_gpu_teacher_forced import as_np, compare_records

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["hexapod-tip-control", "8x4-tip-control", "hexapod-dynamic-stiffness", "8x5-gravity-aligned",
                                  "hexapod-imu-posing", "hexapod-auto-and-inclination-posing", "hexapod-imu-admittance", "mixed-dof-354354",
                                  "hexapod-joint-control", "8x4-joint-control", "mixed-dof-354354-joint-control"])
def test_toggle_manipulate_and_return(case):
    """Walk; request a leg toggle per robot (different legs, two robots none): robots still walking are told to stop first
    (result -1), then the designated leg goes WALKING -> WALKING_TO_MANUAL -> MANUAL while every leg steps to its manipulation
    stance; MANUAL legs follow tip velocity and tip position inputs with the walker frozen whatever the body velocity command
    (the other robots keep walking); a second leg joins on some robots, a third is refused; toggled back, everything walks again.
    The oracle's state is injected before every call (the robots stand still for most of this, where the reference's IK step
    amplifies rounding differences - DESIGN.md section 2.1); request results and leg states are compared exactly.
    The posing cases: with IMU / auto / inclination posing the body pose keeps moving while a robot stands (PID state, poser latches,
    the inclination translation poseForLegManipulation adds for the lifted leg): the posing part of those loops runs in the cycle
    kernel's pose pass (RT_POSE_MARKED), the IMU readings change during the run."""
    if case.startswith("8x5"):     # gravity-aligned tips: the rotation-constrained IK kernels (F_ROT) with the manual-leg logic
        p = synthetic_octopod_params("ripple", 5, 8)
        p.gravity_aligned_tips = 1
    elif case.startswith("8x4"):
        p = synthetic_octopod_params("ripple", 4, 8)
    elif case.startswith("mixed"):   # legs of 3 / 5 / 4 joints in one robot: the loop-level kernels on padded legs (the oracle runs each leg's own chain)
        from syropod_highlevel_controller_amd import synthetic_mixed_dof_params
        p = synthetic_mixed_dof_params("ripple")
    else:
        p = default_hexapod_params("tripod")
    if "joint-control" in case:   # the velocity inputs move the coxa / tibia joints of 3-joint legs; the tip pose they reach comes WITH its
        p.leg_manipulation_mode = 1   # rotation: the rotation-constrained IK on 3-joint legs (4-joint legs ignore the inputs)
    if "stiffness" in case:
        p.admittance_control, p.dynamic_stiffness = 1, 1
    if "imu" in case:
        p.imu_posing = 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    if "auto" in case:
        p.auto_posing, p.inclination_posing, p.manual_posing = 1, 1, 1
    if "imu-admittance" in case:
        p.admittance_control = 1
    posing = p.imu_posing or p.inclination_posing
    n = 8
    L, D = p.leg_count, max(p.leg_dof[l] for l in range(p.leg_count))
    dofs = [p.leg_dof[l] for l in range(L)]
    rng = np.random.default_rng(23)
    eng, ob = BatchEngine(p, n), OracleBatch(p, n)
    _oj = ob.joints

    def oracle_joints():   # the oracle packs each leg's own joint count; the engine's arrays are [legs][longest DOF]
        out = []
        for a in _oj():
            pad, at = np.zeros((n, L, D)), 0
            for l, d in enumerate(dofs):
                pad[:, l, :d] = a[:, at:at + d]
                at += d
            out.append(pad.reshape(n, L * D))
        return tuple(out)
    ob.joints = oracle_joints
    lin, ang = rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(-0.5, 0.5, n)
    effort = rng.normal(0, 0.4, (n, L, D))
    for l, d in enumerate(dofs):
        effort[:, l, d:] = 0.0
    for o in (eng, ob):
        o.set_velocity(lin, ang)
    eng.set_joint_effort(effort.reshape(n, L * D))
    ob.set_joint_effort(np.concatenate([effort[:, l, :d] for l, d in enumerate(dofs)], axis=1))
    for o in (eng, ob):
        if p.admittance_control:
            o.set_tip_force(np.abs(rng.normal(0, 2.0, (n, L, 3))) * 0 + 1.5)
    worst = 0.0
    calls_made = [0]

    def imu():   # a new IMU reading every 20 loops: the robots stand on a slope that changes under them
        calls_made[0] += 1
        if posing and calls_made[0] % 20 == 1:
            from scipy.spatial.transform import Rotation as R
            e = np.stack([rng.uniform(-0.12, 0.12, n), rng.uniform(-0.12, 0.12, n), rng.uniform(-1, 1, n)], axis=1)
            q = R.from_euler("xyz", e).as_quat()
            quat, gyro = np.stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]], axis=1), rng.normal(0, 0.03, (n, 3))
            for o in (eng, ob):
                o.set_imu(quat, gyro)

    def forced_cycles(k):
        nonlocal worst
        for _ in range(k):
            imu()
            eng.set_state(ob.get_state())
            eng.step(1)
            eng.synchronize()
            ob.step(1, 1)
            dd = np.abs(eng.joints()[0] - ob.joints()[0]).reshape(n, L, D)
            d = float(dd.max())
            worst = max(worst, d)
            assert d < 1e-10, (d, np.argwhere(dd > 1e-10)[:8].tolist(), eng.leg_manipulation_state().tolist())
            assert np.array_equal(eng.body_state()[2], ob.body_state()[2])

    def toggle(selection, limit=3000):
        nonlocal worst
        sel = np.array(selection, dtype=np.int32)
        pending = sel >= 0
        seen = set()
        for calls in range(limit):
            if not pending.any():
                break
            cur = np.where(pending, sel, -1).astype(np.int32)
            imu()
            eng.set_state(ob.get_state())
            re, ro = eng.toggle_leg_state(cur), ob.toggle_leg_state(cur)
            assert np.array_equal(re, ro), (calls, re, ro)
            seen.update(re.tolist())
            still = re == -1      # still walking: the node zeroes that robot's velocity inputs and its loop runs the normal cycle
            if still.any():       # (the oracle's loop did both; teacher forcing carries the cycle over, the inputs are mirrored here)
                lin[still], ang[still] = 0.0, 0.0
                eng.set_velocity(lin, ang)
            d = float(np.abs(eng.joints()[0] - ob.joints()[0]).max())   # every robot ran one loop: toggle, or the ordinary cycle
            worst = max(worst, d)
            assert d < 1e-10, (calls, d)
            compare_records(p, FEAT_DEFAULT, as_np(eng.get_state()), as_np(ob.get_state()), tol_q=1e-10)   # ... and every other state field
            assert np.array_equal(eng.body_state()[2], ob.body_state()[2])
            pending &= ~((re == 1) | (re == 2))
        assert not pending.any()
        assert np.array_equal(eng.leg_manipulation_state(), ob.leg_manipulation_state())
        return seen

    forced_cycles(90)
    # ---- robots 0-5 toggle a leg (robot i: leg i % L), robots 6-7 keep walking normally
    first = [i % L if i < 6 else -1 for i in range(n)]
    seen = toggle(first)
    assert {-1, 0, 1, -3} <= seen
    expect = np.zeros((n, L), dtype=np.int32)
    for i in range(6):
        expect[i, first[i]] = 1
    assert np.array_equal(eng.leg_manipulation_state(), expect)
    # ---- manipulate: tip velocity inputs for the primary selection, a position input now and then; body velocity commands are ignored
    lin[:6], ang[:6] = rng.uniform(-0.5, 0.5, (6, 2)), 0.3
    prim = np.array(first, dtype=np.int32)
    for rep in range(6):
        vel = rng.uniform(-1, 1, (n, 3)) * (rng.random((n, 1)) < 0.8)
        pos = np.zeros((n, 3))
        if rep == 3:   # the tip-pose overload: put the tip at a reachable spot under the robot's flank
            for i in range(6):
                pos[i] = [p.stance_position[first[i]][0] * 0.9, p.stance_position[first[i]][1] * 0.9, -0.06]
        for o in (eng, ob):
            o.set_velocity(lin, ang)
            o.set_manual_inputs(prim, vel, pos, None, None, None)
        forced_cycles(25)
    assert (eng.body_state()[2][:6] == 3).all() and (eng.body_state()[2][6:] != 3).all()   # frozen robots stay STOPPED, the others walk
    # ---- a second leg on robots 0-2, then a third on robot 0 is refused
    second = [(first[i] + 2) % L if i < 3 else -1 for i in range(n)]
    toggle(second)
    for i in range(3):
        expect[i, second[i]] = 1
    assert np.array_equal(eng.leg_manipulation_state(), expect)
    sec = np.array(second, dtype=np.int32)
    v1, v2 = rng.uniform(-1, 1, (n, 3)), rng.uniform(-1, 1, (n, 3))
    for o in (eng, ob):
        o.set_manual_inputs(prim, v1, None, sec, v2, None)
    forced_cycles(30)
    seen = toggle([(first[0] + 4) % L] + [-1] * (n - 1))
    assert 2 in seen
    # ---- back to walking
    for o in (eng, ob):
        o.set_manual_inputs(None, None, None, None, None, None)
    toggle(second)
    toggle(first)
    assert (eng.leg_manipulation_state() == 0).all()
    lin[:], ang[:] = rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(-0.5, 0.5, n)
    for o in (eng, ob):
        o.set_velocity(lin, ang)
    forced_cycles(150)
    assert (eng.body_state()[2] != 3).all()     # every robot is walking again (STARTING or MOVING)
    from conftest import parity_report
    parity_report(f"[manual legs {case}] toggle / manipulate / second leg / refusal / return: request results and leg states identical, "
                  f"max |dq| = {worst:.2e} rad per call (teacher-forced)")


def test_manual_legs_unsupported_configurations():
    """Outside the accelerated envelope: the experimental tip-align pose (gravity_aligned_tips on 3-DOF legs)."""
    p = default_hexapod_params("tripod")
    p.gravity_aligned_tips = 1
    eng = BatchEngine(p, 2)
    with pytest.raises(RuntimeError):
        eng.toggle_leg_state(np.array([0, -1], dtype=np.int32))


@pytest.mark.parametrize("case", ["8x5-gravity-aligned", "hexapod-joint-control"])
def test_toggle_and_manipulate_free_running(case):
    """The same without state injection (what teacher forcing cannot show: state the engine keeps to itself between calls, e.g.
    the tip-rotation flag of a leg that went MANUAL on gravity-aligned 5-joint legs, or the FK tip rotation a joint_control MANUAL leg
    holds).  The robots stand for most of this, where the reference's IK step amplifies rounding differences:
    flags exactly, tips to 5 mm."""
    if case.startswith("8x5"):
        p = synthetic_octopod_params("ripple", 5, 8)
        p.gravity_aligned_tips = 1
    else:
        p = default_hexapod_params("tripod")
        p.leg_manipulation_mode = 1
    n, L = 6, p.leg_count
    rng = np.random.default_rng(5)
    eng, ob = BatchEngine(p, n), OracleBatch(p, n)
    lin, ang = rng.uniform(-0.4, 0.4, (n, 2)), rng.uniform(-0.3, 0.3, n)
    for o in (eng, ob):
        o.set_velocity(lin, ang)
        o.step(50) if o is eng else o.step(50, 1)
    sel = np.array([i % L for i in range(n)], dtype=np.int32)
    for _ in range(3000):
        re, ro = eng.toggle_leg_state(sel), ob.toggle_leg_state(sel)
        assert np.array_equal(re, ro)
        if (re == 1).all():
            break
        sel = np.where(re == 1, -1, sel).astype(np.int32)
    assert np.array_equal(eng.leg_manipulation_state(), ob.leg_manipulation_state())
    vel = rng.uniform(-1, 1, (n, 3))
    prim = np.array([i % L for i in range(n)], dtype=np.int32)
    for o in (eng, ob):
        o.set_manual_inputs(prim, vel, None, None, None, None)
        o.step(40) if o is eng else o.step(40, 1)
    eng.synchronize()
    g, o = as_np(eng.get_state()), as_np(ob.get_state())
    assert np.array_equal(g["leg"]["tip_rotation_defined"][:, :L], o["leg"]["tip_rotation_defined"][:, :L])
    assert np.array_equal(g["walk_state"], o["walk_state"])
    assert np.abs(eng.leg_state()["model_tip"] - ob.leg_state()["model_tip"]).max() < 5e-3
    assert np.abs(eng.leg_state()["walker_tip"] - ob.leg_state()["walker_tip"]).max() < 5e-3
    if "joint-control" in case:   # inputs withdrawn: the legs keep the tip pose (with its rotation) the last FK gave the stepper; then back to walking
        assert g["leg"]["tip_rotation_defined"][np.arange(n), prim].all()
        for o in (eng, ob):
            o.set_manual_inputs(prim, vel * 0.0, None, None, None, None)
            o.step(30) if o is eng else o.step(30, 1)
        assert np.abs(eng.joints()[0] - ob.joints()[0]).max() < 2e-2
        sel = prim.copy()
        for _ in range(3000):
            re, ro = eng.toggle_leg_state(sel), ob.toggle_leg_state(sel)
            assert np.array_equal(re, ro)
            if (re == 1).all():
                break
            sel = np.where(re == 1, -1, sel).astype(np.int32)
        assert (eng.leg_manipulation_state() == 0).all()
        g, o = as_np(eng.get_state()), as_np(ob.get_state())
        assert np.array_equal(g["leg"]["tip_rotation_defined"][:, :L], o["leg"]["tip_rotation_defined"][:, :L])
        assert np.abs(eng.joints()[0] - ob.joints()[0]).max() < 1e-6


@pytest.mark.parametrize("mode", ["tip_control", "joint_control"])
def test_complete_checkpoint_carries_a_manual_leg_into_another_engine(mode):
    """shc_engine_get_state alone is the control cycle's state; a complete checkpoint adds shc_engine_get_aux_state (manual-leg records,
    Leg::desired_tip_pose_, reset mode, external / sequence records).  A robot with a MANUAL leg restored into a fresh engine from both
    goes on byte for byte like the original - and demonstrably not from the first alone.  joint_control: the FK tip rotation the MANUAL
    leg's stepper holds travels in the state record (tip_rotation_defined + walker_tip_direction)."""
    p = default_hexapod_params("tripod")
    p.admittance_control = 1
    p.leg_manipulation_mode = 1 if mode == "joint_control" else 0
    n, L = 12, p.leg_count
    rng = np.random.default_rng(4)
    a = BatchEngine(p, n)
    lin, ang = rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(-0.5, 0.5, n)
    force = np.abs(rng.normal(0, 2.0, (n, L, 3))) + 1.0
    a.set_velocity(lin, ang)
    a.set_tip_force(force)
    a.step(150)
    sel = np.array([i % L if i % 4 else -1 for i in range(n)], dtype=np.int32)   # every fourth robot keeps walking
    pending = sel >= 0
    for _ in range(3000):
        if not pending.any():
            break
        res = a.toggle_leg_state(np.where(pending, sel, -1).astype(np.int32))
        still = res == -1
        if still.any():
            lin[still], ang[still] = 0.0, 0.0
            a.set_velocity(lin, ang)
        pending &= ~((res == 1) | (res == 2))
    assert not pending.any()
    assert (a.leg_manipulation_state()[sel >= 0, sel[sel >= 0]] == 1).all()      # LegState MANUAL
    vel = rng.uniform(-0.4, 0.4, (n, 3))
    a.set_manual_inputs(primary_leg=sel, primary_velocity=vel)
    a.step(7)
    a.synchronize()
    state, aux = a.get_state(), a.get_aux_state()

    def go_on(e):
        e.set_velocity(lin, ang)
        e.set_tip_force(force)
        e.set_manual_inputs(primary_leg=sel, primary_velocity=vel)
        e.step(40)
        e.set_manual_inputs(primary_leg=sel, primary_velocity=vel * 0.0, primary_position=np.tile([0.25, 0.2, -0.05], (n, 1)))
        e.step(3)
        back = e.toggle_leg_state(sel)                                           # ... and hand the legs back
        e.step(5)
        e.synchronize()
        return bytes(memoryview(e.get_state()).cast("B")), e.get_aux_state(), e.leg_manipulation_state(), back

    ref = go_on(a)
    b = BatchEngine(p, n)
    b.set_state(state)
    b.set_aux_state(aux)
    assert np.array_equal(b.leg_manipulation_state(), np.where(np.arange(L)[None, :] == sel[:, None], 1, 0))
    got = go_on(b)
    assert got[0] == ref[0] and got[1] == ref[1]
    assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
    c = BatchEngine(p, n)                                                       # the cycle state alone: the walker unfreezes
    c.set_state(state)
    assert (c.leg_manipulation_state() == 0).all()
    for e in (a, b, c):
        e.close()

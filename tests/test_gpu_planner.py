"""GPU: planner mode - StateController::executePlan (state_controller.cpp:653-698) with PoseController::transitionConfiguration /
transitionStance (pose_controller.cpp:710-807), the planner callbacks (:1683-1767) and the LegPoser's external target -
against the oracle."""
import numpy as np
import pytest

from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, WALK_STOPPED, ExternalTarget
from test_gpu_teacher_forced import as_np, compare_records

pytestmark = pytest.mark.gpu
WAITING, WALKING = -2, -1


@pytest.mark.parametrize("case", ["hexapod", "hexapod-admittance", "8x4", "hexapod-rough-terrain", "hexapod-37-robots",
                                  "hexapod-imu-posing-admittance", "hexapod-auto-and-inclination-posing-37-robots", "8x5-gravity-aligned"])
def test_plan_steps_against_the_oracle(case):
    """Walk; switch planner mode on: robots still walking are stopped by the call itself (result -1, their loop is the normal
    cycle) while the ones that stand already wait for plan step 0 (result -2, Model::updateModel only); a joint-configuration step
    (some legs not named, different per robot), a wait, a tip-target + body-pose step (targets sent through the TargetTipPose path:
    the robots stand, so the LegPosers take them), a body-pose-only step; planner off and walk again.  The oracle's state is
    injected before every call (the robots stand still throughout, where the reference's IK step amplifies rounding differences
    - DESIGN.md section 2.1); progress values, plan steps and request flags are compared exactly.  The posing cases: with IMU / auto /
    inclination posing the body pose moves while the robots stand (the posing part of their loops runs in the cycle kernel's pose pass,
    RT_POSE_MARKED); the IMU reading changes every 15 calls."""
    if case == "8x4":
        p = synthetic_octopod_params("ripple", 4, 8)
    elif case.startswith("8x5"):   # gravity-aligned tips: transitionStance turns every tip towards Model::estimateGravity (pose_controller.cpp:786-790)
        p = synthetic_octopod_params("ripple", 5, 8)
        p.gravity_aligned_tips = 1
    else:
        p = default_hexapod_params("tripod")
    if "admittance" in case:
        p.admittance_control = 1
    if "rough" in case:
        p.rough_terrain_mode = 1
    if "imu" in case:
        p.imu_posing = 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    if "auto" in case:
        p.auto_posing, p.inclination_posing, p.manual_posing = 1, 1, 1
    posing = p.imu_posing or p.inclination_posing
    n = 37 if "37" in case else 8      # 37 hexapods = four waves, the last one partly filled: skip marks and walking robots mix within waves
    L, D = p.leg_count, p.leg_dof[0]
    rng = np.random.default_rng(31)
    eng, ob = BatchEngine(p, n), OracleBatch(p, n)
    lin, ang = rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(-0.5, 0.5, n)
    lin[:2], ang[:2] = 0.0, 0.0                        # robots 0, 1 never walk: they are STOPPED when planner mode starts
    lin[9::7], ang[9::7] = 0.0, 0.0                    # ... and a few more, spread over the waves
    for o in (eng, ob):
        o.set_velocity(lin, ang)
        if p.admittance_control:
            o.set_tip_force(np.full((n, L, 3), 1.5))
    worst = 0.0
    calls_made = [0]

    def imu():
        calls_made[0] += 1
        if posing and calls_made[0] % 15 == 1:
            from scipy.spatial.transform import Rotation as R
            e = np.stack([rng.uniform(-0.12, 0.12, n), rng.uniform(-0.12, 0.12, n), rng.uniform(-1, 1, n)], axis=1)
            q = R.from_euler("xyz", e).as_quat()
            quat, gyro = np.stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]], axis=1), rng.normal(0, 0.03, (n, 3))
            for o in (eng, ob):
                o.set_imu(quat, gyro)

    def check(tag, rows=slice(None)):
        nonlocal worst
        dd = np.abs(eng.joints()[0] - ob.joints()[0])[rows]
        d = float(dd.max()) if dd.size else 0.0
        worst = max(worst, d)
        assert d < 1e-10, (tag, d)
        # ... and every other field of the controller state: what the loop-level kernel and the partial cycle launch left behind
        compare_records(p, FEAT_DEFAULT, as_np(eng.get_state()), as_np(ob.get_state()), tol_q=1e-10)

    def forced_cycles(k):
        for _ in range(k):
            imu()
            eng.set_state(ob.get_state())
            eng.step(1)
            eng.synchronize()
            ob.step(1, 1)
            check("cycle")
            assert np.array_equal(eng.body_state()[2], ob.body_state()[2])

    def plan_call():
        imu()
        eng.set_state(ob.get_state())
        (pe, se), (po, so) = eng.execute_plan(), ob.execute_plan()
        assert np.array_equal(pe, po), (pe, po)
        assert np.array_equal(se, so), (se, so)
        check("plan")
        assert np.array_equal(eng.body_state()[2], ob.body_state()[2])
        return pe, se

    def run_step(expect_step, limit=400):
        seen, done = set(), np.zeros(n, dtype=bool)   # (robots finish in different calls - e.g. one whose message names no leg - and wait)
        for _ in range(limit):
            pr, st = plan_call()
            seen.update(pr.tolist())
            assert ((pr == WAITING) == done).all()
            done |= pr == 100
            if done.all():
                break
        assert done.all() and (st == expect_step).all()
        return seen

    forced_cycles(60)
    for o in (eng, ob):
        o.set_planner_mode(True)
    seen = set()
    for _ in range(600):                               # walking robots stop (a step period or two), standing ones wait
        pr, st = plan_call()
        seen.update(pr.tolist())
        if (pr == WAITING).all():
            break
    assert seen == {WALKING, WAITING} and (pr == WAITING).all() and (st == 0).all()
    assert (eng.body_state()[2] == WALK_STOPPED).all()
    # ---- plan step 0: a joint configuration; robot i leaves leg i % L out of the message, robot 7 names no leg at all
    q0 = ob.joints()[0].reshape(n, L, D).copy()
    cfg = q0 + rng.uniform(-0.12, 0.12, q0.shape)
    for i in range(n):
        cfg[i, i % L, :] = np.nan
    cfg[7] = np.nan
    for o in (eng, ob):
        o.set_target_configuration(cfg)
    seen = run_step(1)
    assert 1 in seen and 50 in seen
    q1 = eng.joints()[0].reshape(n, L, D)
    named = ~np.isnan(cfg[:, :, 0])
    assert np.abs(q1[named] - cfg[named]).max() < 1e-12          # cubic Bezier ends on its last node
    assert np.array_equal(q1[:7][~named[:7]], q0[:7][~named[:7]])  # a leg the message leaves out stays (robot 7 finished at once and has
                                                                   # been waiting since: its updateModel keeps stepping the IK)
    for _ in range(5):                                 # waiting for plan step 1: updateModel pulls towards the (stale) poser tips
        pr, st = plan_call()
        assert (pr == WAITING).all() and (st == 1).all()
    # ---- plan step 1: tip targets for half the legs (with a swing clearance) + a body pose for the even robots
    tips = ob.leg_state()["model_tip"].reshape(n, L, 3)
    rows = (ExternalTarget * (n * L))()
    quat = rng.normal(size=(n, L, 4)) * [0.25, 1.0, 0.25, 0.25] + [0, 0, 0.7, 0]     # tips pointing roughly down and out
    quat /= np.linalg.norm(quat, axis=2, keepdims=True)
    for i in range(n):
        for l in range(L):
            if (i + l) % 2:
                continue
            r = rows[i * L + l]
            r.defined = 1
            off = rng.normal(size=3)       # 3.5 - 4.5 cm away: a target that the body pose brings within TIP_TOLERANCE of the tip with
            off *= rng.uniform(0.035, 0.045) / np.linalg.norm(off)   # no lift makes its leg start a call late and the reference never finish
            r.pose[0:3] = list(tips[i, l] + off)
            r.pose[3:7] = [0, 0, 0, 0]                 # UNDEFINED_ROTATION ...
            if (i + l) % 4 == 0:                       # ... or a requested tip rotation: Leg::applyIK then runs its rotation-constrained
                r.pose[3:7] = list(quat[i, l])         # solve and, where that fails (3 joints cannot hold a rotation), the unconstrained retry
            r.transform[:] = [0, 0, 0, 1, 0, 0, 0]
            r.swing_clearance = 0.02 if l % 3 else 0.0
    # every robot stands: the LegPosers take what arrives the TargetTipPose way; on > 3-DOF legs a target with a rotation is refused
    # on that way (a LegStepper could not follow it), so those go to the LegPoser's record directly
    via = 2 if D > 3 else 0
    assert eng.set_external_target(rows, which=via) == 0 and ob.set_external_target(rows, which=via) == 0
    tr = np.tile(np.array([0.004, -0.003, 0.0, 1, 0, 0, 0.0]), (n, L, 1))              # the tf refresh of a defined planner target
    tr[:, :, 3:] = [np.cos(0.01), 0, 0, np.sin(0.01)]
    for o in (eng, ob):
        o.set_external_transform(tr, which=2)
    body = np.tile(np.array([0, 0, 0, 1.0, 0, 0, 0]), (n, 1))
    for i in range(0, n, 2):
        body[i] = [0.01, -0.008, 0.012, np.cos(0.02), np.sin(0.02), 0, 0]
    for i in range(0, n, 2):
        for o in (eng, ob):
            o.set_target_body_pose(body[i:i + 1], first=i)
    got = [(r.defined, r.swing_clearance, tuple(r.transform)) for r in eng.get_external_target(which=2)]
    assert got == [(r.defined, r.swing_clearance, tuple(r.transform)) for r in ob.get_external_target(which=2)]
    run_step(2)
    assert not any(r.defined for r in eng.get_external_target(which=2))                # achieved targets are withdrawn (:801-805)
    assert not any(r.defined for r in ob.get_external_target(which=2))
    reached = eng.leg_state()["model_tip"].reshape(n, L, 3)
    for i in (range(1, n, 2) if not p.admittance_control else ()):   # identity body pose: the tips sit on transform * target (IK_TOLERANCE aside)
        for l in range(L):
            r = rows[i * L + l]
            if r.defined:
                c, s_ = np.cos(0.02), np.sin(0.02)     # yaw by 0.02 rad
                x, y, z = r.pose[0], r.pose[1], r.pose[2]
                want = np.array([0.004 + c * x - s_ * y, -0.003 + s_ * x + c * y, z])
                assert np.abs(reached[i, l] - want).max() < 1.5e-2   # (one DLS step per iteration trails the Bezier; 7 mm on 4-joint legs)
    # ---- plan step 2: a body pose alone
    for o in (eng, ob):
        o.set_target_body_pose(np.tile(np.array([0.0, 0.01, -0.01, 1.0, 0, 0, 0]), (n, 1)))
    run_step(3)
    pr, st = plan_call()
    assert (pr == WAITING).all() and (st == 3).all()
    # ---- planner off: walk again
    lin = rng.uniform(-0.4, 0.4, (n, 2))
    for o in (eng, ob):
        o.set_planner_mode(False)
        o.set_velocity(lin, np.full(n, 0.2))
    forced_cycles(120)
    assert (eng.body_state()[2] != WALK_STOPPED).all()
    from conftest import parity_report
    parity_report(f"[planner {case}] stop / wait / configuration step / tip-target + body-pose step / body-pose step / walk: progress and plan steps "
                  f"identical, max |dq| = {worst:.2e} rad per call (teacher-forced)")


def test_planner_unsupported_configurations():
    for field in ("gravity_aligned_tips",):   # (on 3-DOF legs: the tip-align pose; on longer legs transitionStance would need Model::estimateGravity)
        p = default_hexapod_params("tripod")
        setattr(p, field, 1)
        eng = BatchEngine(p, 2)
        with pytest.raises(RuntimeError):
            eng.execute_plan()


def test_plan_steps_with_manual_legs():
    """Planner mode on robots that hold a MANUAL leg (admittance on): the manually manipulated leg takes no admittance delta in
    stepToPosition / setDesiredTipPose and keeps its LegPoser tip (model.cpp:655-656, pose_controller.cpp:1610-1614, :1680-1684),
    the waiting loop's updateModel uses the stepper's tip for it (updateStance, :134-137).  Teacher-forced against the oracle."""
    p = default_hexapod_params("tripod")
    p.admittance_control = 1
    n, L, D = 6, 6, 3
    rng = np.random.default_rng(47)
    eng, ob = BatchEngine(p, n), OracleBatch(p, n)
    lin, ang = rng.uniform(-0.4, 0.4, (n, 2)), rng.uniform(-0.3, 0.3, n)
    force = np.abs(rng.normal(4.0, 2.0, (n, L, 3)))
    for o in (eng, ob):
        o.set_velocity(lin, ang)
        o.set_tip_force(force)
    worst = 0.0

    def check(tag):
        nonlocal worst
        d = float(np.abs(eng.joints()[0] - ob.joints()[0]).max())
        worst = max(worst, d)
        assert d < 1e-10, (tag, d)
        assert np.array_equal(eng.body_state()[2], ob.body_state()[2]), tag

    def cycles(k):
        for _ in range(k):
            eng.set_state(ob.get_state())
            eng.step(1)
            eng.synchronize()
            ob.step(1, 1)
            check("cycle")

    def toggle(sel):
        sel = np.array(sel, dtype=np.int32)
        pending = sel >= 0
        for _ in range(3000):
            if not pending.any():
                return
            cur = np.where(pending, sel, -1).astype(np.int32)
            eng.set_state(ob.get_state())
            re, ro = eng.toggle_leg_state(cur), ob.toggle_leg_state(cur)
            assert np.array_equal(re, ro)
            check("toggle")
            pending &= ~((re == 1) | (re == 2))
        raise AssertionError("toggle did not finish")

    def plan_until(done_value, limit=700):
        done = np.zeros(n, dtype=bool)
        for _ in range(limit):
            eng.set_state(ob.get_state())
            (pe, se), (po, so) = eng.execute_plan(), ob.execute_plan()
            assert np.array_equal(pe, po) and np.array_equal(se, so), (pe, po)
            check("plan")
            done |= pe == done_value
            if done.all():
                return se
        raise AssertionError(("plan step did not finish", pe))

    cycles(40)
    manual = [i % L if i < 4 else -1 for i in range(n)]
    toggle(manual)                                     # robots 0-3 end up STOPPED with one MANUAL leg, robots 4, 5 keep walking
    assert np.array_equal(eng.leg_manipulation_state(), ob.leg_manipulation_state())
    vel = rng.uniform(-1, 1, (n, 3))
    for o in (eng, ob):
        o.set_manual_inputs(np.array(manual, dtype=np.int32), vel, None, None, None, None)
    cycles(20)
    for o in (eng, ob):
        o.set_planner_mode(True)
    plan_until(WAITING)
    q0 = ob.joints()[0].reshape(n, L, D)
    cfg = q0 + rng.uniform(-0.1, 0.1, q0.shape)
    for o in (eng, ob):
        o.set_target_configuration(cfg)
    st = plan_until(100)
    assert (st == 1).all()
    for _ in range(4):
        plan_until(WAITING, limit=1)
    tips = ob.leg_state()["model_tip"].reshape(n, L, 3)
    rows = (ExternalTarget * (n * L))()
    for i in range(n):
        for l in range(L):
            r = rows[i * L + l]
            r.defined = 1
            off = rng.normal(size=3)
            r.pose[0:3] = list(tips[i, l] + off * 0.04 / np.linalg.norm(off))
            r.transform[:] = [0, 0, 0, 1, 0, 0, 0]
            r.swing_clearance = 0.015
    assert eng.set_external_target(rows) == 0 and ob.set_external_target(rows) == 0
    body = np.tile(np.array([0.008, 0.0, 0.01, 1.0, 0, 0, 0]), (n, 1))
    for o in (eng, ob):
        o.set_target_body_pose(body)
    st = plan_until(100)
    assert (st == 2).all()
    for o in (eng, ob):
        o.set_planner_mode(False)
        o.set_manual_inputs(None, None, None, None, None, None)
    toggle(manual)
    assert (eng.leg_manipulation_state() == 0).all()
    cycles(60)
    from conftest import parity_report
    parity_report(f"[planner with manual legs] toggle / manipulate / plan steps on robots holding a MANUAL leg / toggle back: max |dq| = {worst:.2e} rad per call "
                  "(teacher-forced)")


def test_plan_steps_free_running():
    """The same calls without state injection on the 8 x 4 octopod (state the engine keeps to itself between calls is in play):
    progress values and plan steps exactly, tips to 5 mm (standing robots: DESIGN.md section 2.1)."""
    p = synthetic_octopod_params("ripple", 4, 8)
    n, L, D = 6, 8, 4
    rng = np.random.default_rng(61)
    eng, ob = BatchEngine(p, n), OracleBatch(p, n)
    lin, ang = rng.uniform(-0.4, 0.4, (n, 2)), rng.uniform(-0.3, 0.3, n)
    lin[0], ang[0] = 0.0, 0.0
    for o in (eng, ob):
        o.set_velocity(lin, ang)
        o.step(45) if o is eng else o.step(45, 1)
        o.set_planner_mode(True)

    def until(value, limit=900):
        done = np.zeros(n, dtype=bool)
        for _ in range(limit):
            (pe, se), (po, so) = eng.execute_plan(), ob.execute_plan()
            assert np.array_equal(pe, po) and np.array_equal(se, so), (pe, po)
            done |= pe == value
            if done.all():
                return se
        raise AssertionError(pe)

    until(WAITING)
    cfg = ob.joints()[0].reshape(n, L, D) + rng.uniform(-0.1, 0.1, (n, L, D))
    for o in (eng, ob):
        o.set_target_configuration(cfg)
    assert (until(100) == 1).all()
    assert np.abs(eng.joints()[0] - cfg.reshape(n, -1)).max() < 1e-12
    tips = ob.leg_state()["model_tip"].reshape(n, L, 3)
    rows = (ExternalTarget * (n * L))()
    for i in range(n):
        for l in range(0, L, 2):
            r = rows[i * L + l]
            r.defined = 1
            r.pose[0:3] = list(tips[i, l] + np.array([0.03, -0.02, 0.0]))
            r.transform[:] = [0, 0, 0, 1, 0, 0, 0]
            r.swing_clearance = 0.02
    for o in (eng, ob):
        assert o.set_external_target(rows) == 0
        o.set_target_body_pose(np.tile(np.array([0.0, 0.008, 0.01, 1.0, 0, 0, 0]), (n, 1)))
    assert (until(100) == 2).all()
    eng.synchronize()
    assert np.abs(eng.leg_state()["model_tip"] - ob.leg_state()["model_tip"]).max() < 5e-3
    assert [r.defined for r in eng.get_external_target(which=2)] == [r.defined for r in ob.get_external_target(which=2)]
    for o in (eng, ob):
        o.set_planner_mode(False)
        o.set_velocity(lin, ang + 0.2)
        o.step(80) if o is eng else o.step(80, 1)
    assert np.array_equal(eng.body_state()[2], ob.body_state()[2])


def test_checkpoint_in_the_middle_of_a_plan_under_imu_posing():
    """Under time-dependent posing the LegPoser tips a waiting robot keeps from its last updateStance are engine state (the pose has
    moved on since): state record + auxiliary blob taken in the middle of planner mode restore into a fresh engine that goes on byte
    for byte."""
    from scipy.spatial.transform import Rotation as R
    p = default_hexapod_params("tripod")
    p.imu_posing, p.admittance_control = 1, 1
    p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    n, L = 10, p.leg_count
    rng = np.random.default_rng(12)

    def imu_reading():
        e = np.stack([rng.uniform(-0.12, 0.12, n), rng.uniform(-0.12, 0.12, n), rng.uniform(-1, 1, n)], axis=1)
        q = R.from_euler("xyz", e).as_quat()
        return np.stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]], axis=1), rng.normal(0, 0.03, (n, 3))

    a = BatchEngine(p, n)
    a.set_velocity(rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(-0.5, 0.5, n))
    a.set_tip_force(np.full((n, L, 3), 1.5))
    a.set_imu(*imu_reading())
    a.step(80)
    a.set_planner_mode(True)
    for _ in range(600):
        pr, _ = a.execute_plan()
        if (pr == WAITING).all():
            break
    assert (pr == WAITING).all()
    a.set_imu(*imu_reading())
    for _ in range(5):
        a.execute_plan()
    state, aux = a.get_state(), a.get_aux_state()
    readings = [imu_reading() for _ in range(3)]
    cfg = a.joints()[0].reshape(n, L, -1) + rng.uniform(-0.1, 0.1, (n, L, 3))

    def go_on(e):
        e.set_tip_force(np.full((n, L, 3), 1.5))
        out = []
        for k, rd in enumerate(readings):
            e.set_imu(*rd)
            if k == 1:
                e.set_target_configuration(cfg)
            for _ in range(12):
                out.append(e.execute_plan())
        return bytes(memoryview(e.get_state()).cast("B")), out

    ref = go_on(a)
    b = BatchEngine(p, n)
    b.set_planner_mode(True)
    b.set_state(state)
    b.set_aux_state(aux)
    got = go_on(b)
    assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(ref[1], got[1]))
    assert ref[0] == got[0]

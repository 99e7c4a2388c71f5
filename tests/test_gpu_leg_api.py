"""GPU (-m gpu): the per-leg Leg methods of the boundary (model.h:448-492: setDesiredTipPose, solveIK, updateJointPositions,
applyIK, applyFK) as batched C-ABI calls (shc_leg_*), against the oracle's restatement of the same reference methods."""
import numpy as np
import pytest

from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
from test_gpu_parity import apply, make_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine():
    from syropod_highlevel_controller_amd import engine
    if engine.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests must run the native HIP path")
    return engine.BatchEngine


def walking_pair(Engine, p, n, seed, cycles=137, **kw):
    inp = make_inputs(p, n, seed, **kw)
    eng, ob = Engine(p, n), OracleBatch(p, n)
    apply(eng, inp)
    apply(ob, inp)
    eng.step(cycles)
    eng.synchronize()
    ob.step(cycles, 8)
    # identical joint state on both sides (the redundant 5-joint chain's free-running trajectories differ by 5e-8 rad after the
    # start-up solve alone, tests/test_oracle_conditioning.py): the per-leg methods are compared from the oracle's state
    eng.set_state(ob.get_state())
    return eng, ob


def same_pose(a, b, atol=1e-11):
    np.testing.assert_allclose(a[:, :3], b[:, :3], atol=atol)
    sign = np.sign(np.sum(a[:, 3:] * b[:, 3:], axis=1, keepdims=True))  # q and -q are the same rotation
    np.testing.assert_allclose(a[:, 3:], b[:, 3:] * sign, atol=atol)


CASES = [("hexapod", lambda: default_hexapod_params("tripod")), ("octopod-5dof", lambda: synthetic_octopod_params("ripple", 5, 8)),
         ("quadruped-4dof", lambda: synthetic_octopod_params("amble", 4, 4))]


@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
def test_apply_fk_solve_ik_update_joint_positions(Engine, name, make):
    p = make()
    n = 40
    L, D = p.leg_count, p.leg_dof[0]
    eng, ob = walking_pair(Engine, p, n, 71)
    rng = np.random.default_rng(72)
    # Leg::applyFK: desired joints, then "use_actual" with measured joint positions supplied by the caller
    same_pose(eng.leg_apply_fk(), ob.leg_apply_fk())
    qa = eng.joints()[0].reshape(n * L, D) + rng.normal(0, 0.05, (n * L, D))
    same_pose(eng.leg_apply_fk(qa), ob.leg_apply_fk(qa))
    assert np.array_equal(eng.joints()[0], eng.joints()[0])  # applyFK changes no joint
    # Leg::solveIK: position-only and with the angular rows
    for solve_rotation in (False, True):
        delta = rng.normal(0, 0.004, (n * L, 6))
        dq_g, dq_o = eng.leg_solve_ik(delta, solve_rotation), ob.leg_solve_ik(delta, solve_rotation)
        np.testing.assert_allclose(dq_g, dq_o, atol=1e-12)
        if not solve_rotation:  # without angular rows the rotation part of delta has no effect (model.cpp:737, :746)
            delta[:, 3:] = 0.0
            np.testing.assert_allclose(eng.leg_solve_ik(delta, False), dq_g, atol=1e-15)
    # Leg::updateJointPositions: clamped (simulation = false) and unclamped, large steps so that the clamps act
    for simulation in (False, True):
        dq = rng.normal(0, 0.2, (n * L, D))
        pg, po = eng.leg_update_joint_positions(dq, simulation), ob.leg_update_joint_positions(dq, simulation)
        np.testing.assert_allclose(pg, po, atol=1e-12)
        for a, b in zip(eng.joints(), ob.joints()):
            np.testing.assert_allclose(a, b, atol=1e-11)
    assert (pg == 0.0).any() or (pg < 0.5).any()  # some joint ended near / on a limit


@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
def test_apply_ik_tracks_a_tip_target(Engine, name, make):
    """setDesiredTipPose + applyIK as the reference's cold paths use them: move every tip along a straight line in small
    steps (the workspace search's pattern, model.cpp:397-460), simulation and real mode, then hand back to the fused cycle."""
    p = make()
    n = 32
    L, D = p.leg_count, p.leg_dof[0]
    eng, ob = walking_pair(Engine, p, n, 81)
    rng = np.random.default_rng(82)
    start = eng.leg_apply_fk()
    move = rng.normal(0, 0.012, (n * L, 3))
    for step in range(1, 26):
        pose = np.zeros((n * L, 7))
        pose[:, :3] = start[:, :3] + move * step / 25.0
        for o in (eng, ob):
            o.leg_set_desired_tip_pose(pose, apply_delta=False)
        simulation = step % 2 == 0
        rg, ro = eng.leg_apply_ik(simulation), ob.leg_apply_ik(simulation)
        np.testing.assert_allclose(rg, ro, atol=1e-9)
        np.testing.assert_allclose(eng.joints()[0], ob.joints()[0], atol=1e-9)
    reached = eng.leg_apply_fk()[:, :3]
    assert np.median(np.abs(reached - (start[:, :3] + move)).max(axis=1)) < 2e-3  # one DLS step per 0.5 mm: the tips follow
    lg, lo = eng.leg_state(), ob.leg_state()
    assert np.array_equal(lg["leg_status"] & 4, lo["leg_status"] & 4)         # deviation flags raised by the non-simulated calls
    np.testing.assert_allclose(lg["tip_force"], lo["tip_force"], atol=1e-7)   # applyIK ends with calculateTipForce
    # the fused cycle continues from the joints the per-leg calls left
    eng.step(60)
    eng.synchronize()
    ob.step(60, 8)
    np.testing.assert_allclose(eng.joints()[0], ob.joints()[0], atol=1e-8)


def test_default_argument_uses_the_poser_tip_and_admittance_delta(Engine):
    """setDesiredTipPose() with no pose = the poser's tip pose + admittance delta (model.cpp:657-661): followed by
    applyIK() it is exactly what Model::updateModel does for every leg (model.cpp:142-152)."""
    p = default_hexapod_params("wave")
    p.admittance_control = 1
    n = 24
    eng, ob = walking_pair(Engine, p, n, 91, force=6.0)
    for o in (eng, ob):
        o.leg_set_desired_tip_pose(None, apply_delta=True)
    np.testing.assert_allclose(eng.leg_apply_ik(False), ob.leg_apply_ik(False), atol=1e-9)
    np.testing.assert_allclose(eng.joints()[0], ob.joints()[0], atol=1e-10)


@pytest.mark.parametrize("dof,legs", [(5, 8), (4, 6), (3, 6)])
def test_rotation_constrained_apply_ik(Engine, dof, legs):
    """A desired tip pose with a defined rotation: position solve, rotation solve on the intermediate joint state, and the
    unconstrained retry when the constrained attempt fails (model.cpp:880-936)."""
    p = synthetic_octopod_params("ripple", dof, legs)
    n = 24
    eng, ob = walking_pair(Engine, p, n, 95)
    rng = np.random.default_rng(96)
    cur = eng.leg_apply_fk()
    for step in range(12):
        pose = cur.copy()
        pose[:, :3] += rng.normal(0, 0.004, (n * legs, 3))
        # rotate the current tip rotation a little (or, for a few legs, a lot: forces the retry)
        ang = rng.normal(0, 0.05, (n * legs, 3))
        ang[:: 7] *= 30.0
        dq = np.concatenate([np.ones((n * legs, 1)), 0.5 * ang], axis=1)
        dq /= np.linalg.norm(dq, axis=1, keepdims=True)
        w1, v1 = cur[:, 3:4], cur[:, 4:]
        w2, v2 = dq[:, :1], dq[:, 1:]
        pose[:, 3:4] = w1 * w2 - np.sum(v1 * v2, axis=1, keepdims=True)
        pose[:, 4:] = w1 * v2 + w2 * v1 + np.cross(v1, v2)
        for o in (eng, ob):
            o.leg_set_desired_tip_pose(pose, apply_delta=False)
        rg, ro = eng.leg_apply_ik(True), ob.leg_apply_ik(True)
        np.testing.assert_allclose(rg, ro, atol=1e-8)
        np.testing.assert_allclose(eng.joints()[0], ob.joints()[0], atol=1e-8)
        cur = eng.leg_apply_fk()
        same_pose(cur, ob.leg_apply_fk(), atol=1e-8)


def test_selection_of_instances_and_legs(Engine):
    p = default_hexapod_params("tripod")
    n = 30
    eng, ob = walking_pair(Engine, p, n, 99)
    full = eng.leg_apply_fk()
    part = eng.leg_apply_fk(first=7, count=5)
    assert np.array_equal(part, full.reshape(n, 6, 7)[7:12].reshape(-1, 7))
    one = eng.leg_apply_fk(first=3, count=4, leg=2)
    assert np.array_equal(one, full.reshape(n, 6, 7)[3:7, 2])
    # move only leg 4 of instances 10..14
    q0 = eng.joints()[0].reshape(n, 6, 3).copy()
    dq = np.full((5, 3), 0.01)
    eng.leg_update_joint_positions(dq, True, first=10, count=5, leg=4)
    q1 = eng.joints()[0].reshape(n, 6, 3)
    changed = np.zeros((n, 6), dtype=bool)
    changed[10:15, 4] = True
    assert np.allclose(q1[changed], q0[changed] + 0.01) and np.array_equal(q1[~changed], q0[~changed])
    from syropod_highlevel_controller_amd.engine import ShcError
    with pytest.raises(ShcError):
        eng.leg_apply_fk(first=28, count=5)
    with pytest.raises(ShcError):
        eng.leg_apply_fk(leg=6)

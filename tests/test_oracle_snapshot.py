"""CPU: shc_instance_state (include/shc_batch.h) carries the oracle's complete controller state.

A snapshot taken mid-run and loaded into a FRESH robot (same parameters and inputs) must continue bit-identically: this is
what makes the teacher-forced one-step GPU parity tests (tests/test_gpu_teacher_forced.py) meaningful - nothing the next
cycle reads is missing from the record."""
import ctypes as C

import numpy as np
import pytest

from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
from syropod_highlevel_controller_amd.params import InstanceState


def _cases():
    p = default_hexapod_params("tripod")
    yield "hexapod-tripod", p, {}
    p = default_hexapod_params("wave")
    p.admittance_control, p.imu_posing, p.dynamic_stiffness = 1, 1, 1
    p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    yield "hexapod-wave-admittance-imu", p, dict(imu=True, force=20.0)
    p = default_hexapod_params("ripple")
    p.auto_posing, p.inclination_posing = 1, 1
    yield "hexapod-ripple-auto-inclination", p, dict(imu=True)
    p = default_hexapod_params("amble")
    p.auto_posing, p.pose_frequency = 1, 0.8
    yield "hexapod-amble-auto-own-clock", p, {}
    p = synthetic_octopod_params("ripple", 5, 8)
    p.gravity_aligned_tips = 1
    yield "octopod-gravity-aligned", p, {}


def _inputs(p, n, seed, imu=False, force=None):
    rng = np.random.default_rng(seed)
    L, D = p.leg_count, p.leg_dof[0]
    inp = dict(lin=rng.uniform(-0.7, 0.7, (n, 2)), ang=rng.uniform(-1, 1, n), effort=rng.normal(0, 0.5, (n, L * D)),
               tv=rng.uniform(-1, 1, (n, 3)) * (rng.random((n, 1)) < 0.5), rv=rng.uniform(-1, 1, (n, 3)) * (rng.random((n, 1)) < 0.5))
    inp["lin"][::5] = 0
    inp["ang"][::5] = 0
    if imu:
        e = rng.uniform(-0.15, 0.15, (n, 3))
        q = np.stack([np.ones(n), e[:, 0] / 2, e[:, 1] / 2, rng.uniform(-1, 1, n)], axis=1)
        inp["imu_q"], inp["gyro"] = q, rng.normal(0, 0.05, (n, 3))
    if force:
        inp["force"] = np.stack([rng.normal(0, 1, (n, L)), rng.normal(0, 1, (n, L)), rng.uniform(0, force, (n, L))], axis=2)
    return inp


def _apply(ob, inp):
    ob.set_velocity(inp["lin"], inp["ang"])
    ob.set_joint_effort(inp["effort"])
    ob.set_pose_input(inp["tv"], inp["rv"])
    if "imu_q" in inp:
        ob.set_imu(inp["imu_q"], inp["gyro"])
    if "force" in inp:
        ob.set_tip_force(inp["force"])


@pytest.mark.parametrize("name,p,kw", list(_cases()), ids=[c[0] for c in _cases()])
def test_snapshot_restores_a_fresh_oracle_bit_exactly(name, p, kw):
    n = 12
    inp = _inputs(p, n, 7, **kw)
    a = OracleBatch(p, n)
    _apply(a, inp)
    for k in (1, 37, 140, 61):  # snapshots at STARTING / MOVING and mid-swing phases
        a.step(k)
        snap = a.get_state()
        b = OracleBatch(p, n)   # fresh robots: everything not in the record is still at its start-up value
        _apply(b, inp)
        b.set_state(snap)
        ref = OracleBatch(p, n)
        _apply(ref, inp)
        ref.set_state(snap)
        # 1. the record round-trips through the oracle unchanged (tip rotations travel as their x axis - all the path reads
        #    of them - and are rebuilt as FromTwoVectors(x, axis): equal up to rounding instead of bit for bit)
        exact = not p.gravity_aligned_tips
        if exact:
            assert bytes(b.get_state()) == bytes(snap)
        else:
            dt = np.dtype(InstanceState)
            x, y = np.frombuffer(b.get_state(), dtype=dt), np.frombuffer(snap, dtype=dt)
            for f in ("walker_tip_direction", "origin_tip_direction"):
                np.testing.assert_allclose(x["leg"][f], y["leg"][f], atol=1e-15)
        # 2. the restored robots continue exactly like the original
        a2q = None
        for m in (1, 1, 50):
            a.step(m)
            b.step(m)
            qa, qda = a.joints()
            qb, qdb = b.joints()
            same = np.array_equal if exact else (lambda x, y: np.allclose(x, y, rtol=0, atol=1e-9))
            assert same(qa, qb) and same(qda, qdb), f"{name}: restored run diverged after snapshot at +{k}"
            la, lb = a.leg_state(), b.leg_state()
            for key in la:
                assert same(la[key], lb[key]), key
            pa, pb = a.body_state(), b.body_state()
            for x, y in zip(pa, pb):
                assert same(x, y)
        # the original ran 52 more cycles: bring it back in line with the schedule by restoring too
        a.set_state(b.get_state())


def test_snapshot_layout_matches_the_header():
    from oracle_lib import lib
    assert C.sizeof(InstanceState) % 8 == 0
    from syropod_highlevel_controller_amd import engine
    assert engine.lib().shc_sizeof_instance_state() == C.sizeof(InstanceState)

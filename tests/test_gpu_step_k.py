"""GPU (-m gpu): shc_engine_step_k - K loop iterations in one launch, each with its own inputs (src/main.cpp:106-131: callbacks
deliver, StateController::loop runs, the desired joint state is published), for batches of any size - against (a) the same cycles
as K x { setters; shc_engine_step(1) }, byte for byte, and (b) the CPU oracle free-running.
"""
import numpy as np
import pytest

from conftest import parity_report
from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
from test_gpu_resident import config3_params, force_sample, imu_sample, state_bytes, velocity_schedule

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine():
    from syropod_highlevel_controller_amd import engine
    if engine.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests must run the native HIP path")
    return engine.BatchEngine


def make_case(case):
    if case.startswith("config2"):
        return default_hexapod_params("tripod"), 333
    if case.startswith("config3"):
        return config3_params(), 250
    if case.startswith("octopod"):
        return synthetic_octopod_params("ripple", 5, 8), 203
    if case.startswith("rough"):
        p = default_hexapod_params("tripod")
        p.rough_terrain_mode = 1
        return p, 170
    if case.startswith("tip_align"):   # gravity_aligned_tips on 3-joint legs: the tip-align pose kernels
        p = default_hexapod_params("ripple")
        p.gravity_aligned_tips = 1
        return p, 149
    if case == "split_streams":   # enough wavefronts for the two-stream form of the launch (>= 4 096 waves)
        return default_hexapod_params("ripple"), 41000
    if case.startswith("auto_pose"):   # a runtime-flag (F_DYN) configuration: no batch kernel in the default build - K single launches inside the call
        p = default_hexapod_params("amble")
        p.auto_posing = 1
        return p, 140
    if case.startswith("serial_"):     # SHC_FEAT_STEP_K_SERIAL on engine B: the serial form where a batch kernel exists
        return make_case(case[len("serial_"):])
    return synthetic_octopod_params("amble", 4, 4), 130   # generic_4x4: the runtime-flag kernels


def input_rows(rng, case, p, n, K):
    """K rows of every input group the case carries (None = the group is not given: held)."""
    legs, dof = p.leg_count, p.leg_dof[0]
    sched = velocity_schedule(rng, n, K)
    rows = {"lin": np.stack([s[0] for s in sched]), "ang": np.stack([s[1] for s in sched]), "imu_q": None, "imu_w": None, "force": None, "effort": None}
    if case.startswith("config3") or case.startswith("rough"):
        rows["force"] = np.stack([force_sample(rng, n, legs) * (1.0 if k % 4 else 0.0) for k in range(K)])   # forces that come and go (touchdown / lift-off)
    if case.startswith("config3"):
        im = [imu_sample(rng, n) for _ in range(K)]
        rows["imu_q"], rows["imu_w"] = np.stack([i[0] for i in im]), np.stack([i[1] for i in im])
    if "joint_efforts" in case:
        rows["effort"] = rng.normal(0, 0.5, (K, n, legs * dof))
    return rows


def step_k_on_device(eng, rows, K):
    import torch
    dev = {k: (torch.from_numpy(np.ascontiguousarray(v)).cuda() if v is not None else None) for k, v in rows.items()}
    torch.cuda.synchronize()
    ptr = lambda k: dev[k].data_ptr() if dev[k] is not None else None
    eng.step_k(K, velocity=(ptr("lin"), ptr("ang")), imu=(ptr("imu_q"), ptr("imu_w")) if dev["imu_q"] is not None else None,
               tip_force=ptr("force"), joint_effort=ptr("effort"))
    eng.synchronize()
    return dev   # (kept alive by the caller until the launch has run)


@pytest.mark.parametrize("case", ["config2", "config2_joint_efforts", "config3", "config3_joint_efforts", "octopod", "octopod_joint_efforts", "rough_terrain",
                                  "rough_terrain_joint_efforts", "generic_4x4", "split_streams", "tip_align", "tip_align_joint_efforts",
                                  "auto_pose", "auto_pose_joint_efforts", "serial_config3_joint_efforts", "serial_rough_terrain", "serial_split_streams"])
def test_step_k_is_byte_identical_to_single_cycle_launches(Engine, case):
    """Engine A: setters with row k + shc_engine_step(1), K times.  Engine B: ONE shc_engine_step_k launch over the K-deep device arrays.  q / qd of
    EVERY cycle (the K-deep output ring) and the complete state record at the end are equal byte for byte; the last row stays in force (a further
    plain step on both engines agrees too).  Two launches in a row (the second continues from the first)."""
    rng = np.random.default_rng(77)
    p, n = make_case(case)
    K = 12 if "split_streams" not in case else 5
    a, b = Engine(p, n), Engine(p, n)
    if case.startswith("serial_"):
        from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, FEAT_STEP_K_SERIAL
        b.set_features(FEAT_DEFAULT | FEAT_STEP_K_SERIAL)
        case = case[len("serial_"):]
    legs, dof = p.leg_count, p.leg_dof[0]
    if "joint_efforts" in case:
        e0 = rng.normal(0, 0.5, (n, legs * dof))
        for e in (a, b):
            e.set_joint_effort(e0)
    lin0, ang0 = rng.uniform(-0.7, 0.7, (n, 2)), rng.uniform(-1, 1, n)
    for e in (a, b):
        e.set_velocity(lin0, ang0)
        e.step(37)
    for launch in range(2):
        rows = input_rows(rng, case, p, n, K)
        qa = []
        for k in range(K):
            a.set_velocity(rows["lin"][k], rows["ang"][k])
            if rows["imu_q"] is not None:
                a.set_imu(rows["imu_q"][k], rows["imu_w"][k])
            if rows["force"] is not None:
                a.set_tip_force(rows["force"][k])
            if rows["effort"] is not None:
                a.set_joint_effort(rows["effort"][k])
            a.step(1)
            if case != "split_streams" or k in (0, K - 1):
                qa.append((k, a.joints()))
        a.synchronize()
        keep = step_k_on_device(b, rows, K)
        for k, (q, qd) in qa:
            qb, qdb = b.step_k_joints(k)
            assert q.tobytes() == qb.tobytes() and qd.tobytes() == qdb.tobytes(), (case, launch, k, float(np.abs(q - qb).max()))
        assert a.joints()[0].tobytes() == b.joints()[0].tobytes()
        assert state_bytes(a) == state_bytes(b), (case, launch)
        del keep
    for e in (a, b):   # the inputs of the last row are the held inputs now
        e.step(3)
    assert state_bytes(a) == state_bytes(b)
    a.close()
    b.close()


@pytest.mark.parametrize("n,K", [(1, 1), (7, 257), (11, 1), (64, 2), (10, 4096)])
def test_step_k_edge_shapes(Engine, n, K):
    """One robot, one cycle, fewer robots than a wavefront holds, the longest launch the entry point takes (K = 4096): the K-deep rows and the
    K-deep output ring against K single launches, byte for byte (first, middle and last cycle of the ring; the whole state at the end)."""
    rng = np.random.default_rng(100 + n + K)
    p = default_hexapod_params("ripple")
    a, b = Engine(p, n), Engine(p, n)
    lin0, ang0 = rng.uniform(-0.7, 0.7, (n, 2)), rng.uniform(-1, 1, n)
    for e in (a, b):
        e.set_velocity(lin0, ang0)
        e.step(23)
    base_l, base_a = rng.uniform(-0.7, 0.7, (n, 2)), rng.uniform(-1, 1, n)
    k_ = np.arange(K)[:, None, None]
    rows = {"lin": base_l[None] * (0.6 + 0.4 * np.sin(0.01 * k_)), "ang": base_a[None] * (0.6 + 0.4 * np.cos(0.013 * k_[:, :, 0])),
            "imu_q": None, "imu_w": None, "force": None, "effort": None}
    probe = sorted({0, K // 2, K - 1})
    qa = {}
    for k in range(K):
        a.set_velocity(rows["lin"][k], rows["ang"][k])
        a.step(1)
        if k in probe:
            qa[k] = a.joints()
    a.synchronize()
    keep = step_k_on_device(b, rows, K)
    for k in probe:
        qb, qdb = b.step_k_joints(k)
        assert qa[k][0].tobytes() == qb.tobytes() and qa[k][1].tobytes() == qdb.tobytes(), (n, K, k)
    assert state_bytes(a) == state_bytes(b)
    del keep
    a.close()
    b.close()


def test_step_k_with_a_manual_leg(Engine):
    """A robot with a MANUAL leg runs on the manual-leg kernels, which have no batch form: shc_engine_step_k serves it as K single launches inside the
    call - velocity rows per cycle, the manual inputs held like the pose inputs - and lands on the same bytes as the caller's own loop."""
    p = default_hexapod_params("tripod")
    n, L, K = 24, p.leg_count, 9
    rng = np.random.default_rng(11)
    a, b = Engine(p, n), Engine(p, n)
    sel = np.array([i % L if i % 3 else -1 for i in range(n)], dtype=np.int32)   # every third robot keeps walking
    vel = rng.uniform(-0.3, 0.3, (n, 3))
    for e in (a, b):
        lin, ang = np.zeros((n, 2)), np.zeros(n)
        lin[sel < 0], ang[sel < 0] = [0.4, 0.1], 0.3
        e.set_velocity(lin, ang)
        e.step(60)
        pending = sel >= 0
        for _ in range(2000):
            if not pending.any():
                break
            res = e.toggle_leg_state(np.where(pending, sel, -1).astype(np.int32))
            pending &= ~((res == 1) | (res == 2))
        assert not pending.any()
        e.set_manual_inputs(primary_leg=sel, primary_velocity=vel)
    rows = input_rows(rng, "config2", p, n, K)
    rows["lin"][:, sel >= 0], rows["ang"][:, sel >= 0] = 0.0, 0.0
    qa = []
    for k in range(K):
        a.set_velocity(rows["lin"][k], rows["ang"][k])
        a.step(1)
        qa.append(a.joints())
    a.synchronize()
    keep = step_k_on_device(b, rows, K)
    for k, (q, qd) in enumerate(qa):
        qb, qdb = b.step_k_joints(k)
        assert q.tobytes() == qb.tobytes() and qd.tobytes() == qdb.tobytes(), k
    assert state_bytes(a) == state_bytes(b)
    assert np.abs(qa[-1][0] - qa[0][0]).max() > 1e-4   # (the manual legs and the walking robots did move)
    del keep
    a.close()
    b.close()


def test_step_k_with_inputs_held_equals_fused_steps(Engine):
    """inputs = NULL: K cycles with everything held = shc_engine_step(e, K)."""
    p, n = default_hexapod_params("wave"), 500
    rng = np.random.default_rng(5)
    a, b = Engine(p, n), Engine(p, n)
    lin, ang = rng.uniform(-0.7, 0.7, (n, 2)), rng.uniform(-1, 1, n)
    for e in (a, b):
        e.set_velocity(lin, ang)
        e.step(50)
    a.step(16)
    b.step_k(16)
    assert state_bytes(a) == state_bytes(b)
    a.close()
    b.close()


def test_step_k_refusals(Engine):
    import torch
    from syropod_highlevel_controller_amd.engine import ShcError
    p, n = default_hexapod_params("tripod"), 64
    e = Engine(p, n)
    lin = torch.zeros((4, n, 2), dtype=torch.float64, device="cuda")
    with pytest.raises(ShcError):
        e.step_k(4, velocity=(lin.data_ptr(), None))        # linear without angular
    with pytest.raises(ShcError):
        e.step_k(0)
    e.step_k(2)
    with pytest.raises(ShcError):
        e.step_k_joints(2)                                   # the ring holds cycles 0 .. 1
    e.close()


@pytest.mark.parametrize("case", ["config3_joint_efforts", "octopod"])
def test_step_k_against_the_oracle(Engine, case):
    """Free-running against the CPU oracle with new inputs in every cycle: 6 launches of 16 cycles, every cycle's q from the output ring."""
    rng = np.random.default_rng(9)
    p, n = make_case(case)
    n = 96
    K, launches = 16, 6
    legs, dof = p.leg_count, p.leg_dof[0]
    eng, ob, tw = Engine(p, n), OracleBatch(p, n), OracleBatch(p, n)
    e0 = rng.normal(0, 0.5, (n, legs * dof))
    lin0, ang0 = rng.uniform(-0.7, 0.7, (n, 2)), rng.uniform(-1, 1, n)
    for o, kk in ((eng, 1.0), (ob, 1.0), (tw, 1 + 1e-13)):
        if "joint_efforts" in case:
            o.set_joint_effort(e0)
        o.set_velocity(lin0 * kk, ang0)
    eng.step(40)
    eng.synchronize()
    for o in (ob, tw):
        o.step(40, 4)
    worst, well = 0.0, np.ones(n, bool)
    for launch in range(launches):
        rows = input_rows(rng, case, p, n, K)
        keep = step_k_on_device(eng, rows, K)
        for k in range(K):
            for o, kk in ((ob, 1.0), (tw, 1 + 1e-13)):
                o.set_velocity(rows["lin"][k] * kk, rows["ang"][k])
                if rows["imu_q"] is not None:
                    o.set_imu(rows["imu_q"][k], rows["imu_w"][k])
                if rows["force"] is not None:
                    o.set_tip_force(rows["force"][k])
                if rows["effort"] is not None:
                    o.set_joint_effort(rows["effort"][k])
                o.step(1, 4)
            q = eng.step_k_joints(k)[0]
            well &= np.abs(ob.joints()[0] - tw.joints()[0]).max(axis=1) <= 1e-9
            worst = max(worst, float(np.abs(q - ob.joints()[0])[well].max()))
        del keep
    parity_report(f"[step_k {case}] {n} robots x {launches} launches of {K} cycles, new inputs every cycle, free-running: max |dq| vs oracle = {worst:.2e} rad over the "
                  f"{well.mean():.0%} whose reference trajectory is well-posed")
    assert worst <= 1e-6 and well.mean() > 0.5
    eng.close()

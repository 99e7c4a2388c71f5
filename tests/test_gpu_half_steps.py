"""GPU (-m gpu): a rotation-constrained cycle as TWO launches (gravity_aligned_tips on legs with more than 3 joints).

The whole cycle of the 8 x 5 feature-exact kernels needs 256 VGPRs + 76 - 92 AGPRs = one wavefront per SIMD; the walker / poser half
(WalkController::updateWalk + PoseController::updateStance) and the model half (Model::updateModel: Leg::applyIK with the rotation solve,
model.cpp:861-941) each fit two.  From 2 048 wavefronts per launch on, `shc_engine_step` therefore runs each cycle as
shc_cycle_half_kernel<ROLE_FRONT> + <ROLE_BACK>; the model half redoes PoseController::updateStance's two transforms from the stored walker state.
SHC_ROT_SPLIT (read at shc_engine_create) forces it on (1) or off (0) at any size - that is how the small cases here reach it.

Bar: the two-launch form is byte-identical to the one-launch form (joints and the complete state record), and holds the oracle bar."""
import numpy as np
import pytest

from syropod_highlevel_controller_amd import synthetic_octopod_params
from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, FEAT_ODOMETRY
from test_gpu_parity import Engine, make_inputs  # noqa: F401  (Engine: module fixture)
import test_gpu_parity
import test_gpu_teacher_forced

pytestmark = pytest.mark.gpu


def octopods(gait="ripple", dof=5, legs=8):
    p = synthetic_octopod_params(gait, dof, legs)
    p.gravity_aligned_tips = 1
    return p


def drive(eng, inp, efforts):
    """A walk with a direction change, a stop and a restart; returns the joints after every segment and the final state records."""
    out = []
    n = len(inp["ang"])
    eng.set_velocity(inp["lin"], inp["ang"])
    if efforts:
        eng.set_joint_effort(inp["effort"])
    for k in (1, 1, 37, 16, 150):
        eng.step(k)
        eng.synchronize()
        out.append(eng.joints())
    eng.set_velocity(-inp["lin"], 0.5 * inp["ang"])
    for k in (1, 90):
        eng.step(k)
        eng.synchronize()
        out.append(eng.joints())
    eng.set_velocity(np.zeros((n, 2)), np.zeros(n))
    eng.step(260)
    eng.set_velocity(inp["lin"], inp["ang"])
    eng.step(120)
    eng.synchronize()
    out.append(eng.joints())
    return out, eng.get_state()


@pytest.mark.parametrize("morph,efforts", [(("ripple", 5, 8), False), (("ripple", 5, 8), True), (("tripod", 4, 6), False), (("amble", 5, 4), False), (("wave", 4, 8), False)],
                         ids=["8x5", "8x5-tip-force-estimate", "6x4", "4x5", "8x4"])
def test_two_launch_cycles_are_byte_identical_to_one_launch(Engine, monkeypatch, morph, efforts):
    p = octopods(*morph)
    n = 333   # (8 legs: 8 robots per wavefront, the last wavefront is ragged)
    inp = make_inputs(p, n, 77, zero_every=9)
    runs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("SHC_ROT_SPLIT", mode)
        eng = Engine(p, n)
        eng.set_features(FEAT_DEFAULT if efforts else FEAT_ODOMETRY)
        runs.append(drive(eng, inp, efforts))
        eng.close()
    (ja, sa), (jb, sb) = runs
    for (qa, qda), (qb, qdb) in zip(ja, jb):
        assert np.isfinite(qa).all() and np.array_equal(qa, qb) and np.array_equal(qda, qdb)
    assert bytes(memoryview(sa).cast("B")) == bytes(memoryview(sb).cast("B"))


def test_two_launch_cycles_chosen_by_size(Engine, monkeypatch):
    """16 384 octopods = 2 048 wavefronts: the engine picks the two-launch form on its own; 65 536 run it on both halves of a split step."""
    p = octopods()
    for n, cycles in ((16384, (1, 16, 40)), (65536, (1, 16, 30))):
        inp = make_inputs(p, n, 78, zero_every=11)
        res = []
        for mode in (None, "0"):
            if mode is None:
                monkeypatch.delenv("SHC_ROT_SPLIT", raising=False)
            else:
                monkeypatch.setenv("SHC_ROT_SPLIT", mode)
            eng = Engine(p, n)
            eng.set_features(FEAT_ODOMETRY)
            eng.set_velocity(inp["lin"], inp["ang"])
            for k in cycles:
                eng.step(k)
            eng.synchronize()
            res.append(eng.joints())
            eng.close()
        assert np.isfinite(res[0][0]).all()
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


MORPHS = [(5, 8, "ripple"), (4, 6, "tripod"), (5, 4, "amble")]
# with the tip-force estimate (FEAT_DEFAULT) the two-launch form exists for the BASELINE morphology 8 x 5 only, without it for every morphology


@pytest.mark.parametrize("dof,legs,gait", MORPHS)
def test_two_launch_cycles_free_running_against_the_oracle(Engine, monkeypatch, dof, legs, gait):
    monkeypatch.setenv("SHC_ROT_SPLIT", "1")
    p = octopods(gait, dof, legs)
    n = 48
    inp = make_inputs(p, n, 300 + dof * 10 + legs, zero_every=7)
    for feats in (FEAT_DEFAULT, FEAT_ODOMETRY):
        _, _, worst = test_gpu_parity.run_pair(Engine, p, n, inp, [1, 1, 1, 47, 100, 150, 200], features=feats)
        assert worst < 1e-9


@pytest.mark.parametrize("dof,legs,gait", MORPHS)
def test_two_launch_cycles_teacher_forced_against_the_oracle(Engine, monkeypatch, dof, legs, gait):
    monkeypatch.setenv("SHC_ROT_SPLIT", "1")
    p = octopods(gait, dof, legs)
    n, cycles = 64, 450
    inp = make_inputs(p, n, 300 + dof * 10 + legs, zero_every=7)
    for feats in (FEAT_DEFAULT, FEAT_ODOMETRY):
        test_gpu_teacher_forced.teacher_forced(Engine, p, n, inp, cycles, test_gpu_teacher_forced.stop_go_schedule(p, n, 301, cycles, every=120, pose=True),
                                               features=feats, label=f"two-launch cycles, gravity-aligned {legs}x{dof}, features {feats}")


def north_star_octopods():
    """BASELINE config 3's feature set (admittance from measured tip forces + IMU pose compensation) together with gravity-aligned tips on the 8 x 5 octopods:
    src/model.cpp:880-903 + src/pose_controller.cpp:1191-1236 + src/admittance_controller.cpp:22-63 in one cycle - feature-exact kernels since round 5
    (<8,5,MANUAL|IMU|ADM|ODOM|ROT>), large launches as walker-half + model-half (the admittance update is the model half's)."""
    p = octopods()
    p.admittance_control, p.imu_posing = 1, 1
    p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    return p


@pytest.mark.parametrize("dynamic_stiffness", [0, 1], ids=["fixed-stiffness", "dynamic-stiffness"])
def test_two_launch_cycles_with_admittance_and_imu_posing_are_byte_identical_to_one_launch(Engine, monkeypatch, dynamic_stiffness):
    p = north_star_octopods()
    p.dynamic_stiffness = dynamic_stiffness
    n = 333
    inp = make_inputs(p, n, 79, imu=True, force=20.0, zero_every=9)
    rng = np.random.default_rng(5)
    runs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("SHC_ROT_SPLIT", mode)
        eng = Engine(p, n)
        eng.set_features(FEAT_ODOMETRY)
        eng.set_imu(inp["imu_q"], inp["gyro"])
        eng.set_tip_force(inp["force"])
        out, st = drive(eng, inp, False)
        eng.set_tip_force(inp["force"][::-1].copy())     # new forces and a new IMU sample, a few more cycles
        eng.set_imu(inp["imu_q"][::-1].copy(), inp["gyro"][::-1].copy())
        eng.step(40)
        eng.synchronize()
        runs.append((out + [eng.joints()], eng.get_state(), eng.leg_state()))
        eng.close()
    (ja, sa, la), (jb, sb, lb) = runs
    for (qa, qda), (qb, qdb) in zip(ja, jb):
        assert np.isfinite(qa).all() and np.array_equal(qa, qb) and np.array_equal(qda, qdb)
    assert bytes(memoryview(sa).cast("B")) == bytes(memoryview(sb).cast("B"))


def test_two_launch_cycles_with_admittance_and_imu_posing_against_the_oracle(Engine, monkeypatch):
    monkeypatch.setenv("SHC_ROT_SPLIT", "1")
    p = north_star_octopods()
    n, cycles = 64, 450
    inp = make_inputs(p, n, 381, imu=True, force=20.0, zero_every=7)
    test_gpu_teacher_forced.teacher_forced(Engine, p, n, inp, cycles, test_gpu_teacher_forced.stop_go_schedule(p, n, 302, cycles, every=120, pose=True),
                                           features=FEAT_ODOMETRY, label="two-launch cycles, 8x5 gravity-aligned + admittance + IMU posing")
    # free-running on top: redundant chains under random 0 - 20 N forces and a tilted body leave some REFERENCE trajectories ill-posed (a twin oracle with
    # inputs x (1 + 1e-13) tells which, tests/test_gpu_parity.py header); the bar holds over the well-posed ones, the teacher-forced run above over every one
    _, _, worst = test_gpu_parity.run_pair(Engine, p, n, inp, [1, 1, 1, 47, 100, 150], features=FEAT_ODOMETRY, twin=True, min_well_posed=0.5)
    assert worst < 1e-6

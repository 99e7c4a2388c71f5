"""GPU (-m gpu): TEACHER-FORCED one-step parity — the primary parity bar.

Before every control cycle the oracle's complete controller state (shc_instance_state, include/shc_batch.h) is loaded
into the engine through shc_engine_set_state; both then advance ONE cycle on the same inputs and the full state records
are compared, field by field, for EVERY instance.  Differences cannot accumulate, so the comparison is immune to the
expanding joint-limit term of the reference's Leg::solveIK (model.cpp:788-790) that forces the free-running tests
(tests/test_gpu_parity.py) to mask ill-posed trajectories: here no instance is ever excluded and the bar is 1e-12 rad
(BASELINE.json north_star asks for 1e-6).  tests/test_oracle_snapshot.py shows on the CPU that the record is complete.

The oracle itself runs free (it is never corrected by the engine), so the visited states are the reference's own
trajectory: walk-state transitions, swing / stance hand-overs, stops and restarts, saturated clamps.
"""
import copy
import os

import numpy as np
import pytest

from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, InstanceState, VEL_REAL
from test_gpu_parity import apply, make_inputs

pytestmark = pytest.mark.gpu
TOL_Q = 1e-12   # rad after one cycle from identical state (measured: 1e-16 ... 1e-14)


@pytest.fixture(scope="module")
def Engine():
    from syropod_highlevel_controller_amd import engine
    if engine.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests must run the native HIP path")
    return engine.BatchEngine


def as_np(states):
    return np.frombuffer(states, dtype=np.dtype(InstanceState))


def compare_records(p, features, g, o, tol_q=TOL_Q):
    """g: engine records, o: oracle records (numpy structured arrays).  Returns max |dq|."""
    L, D = p.leg_count, max(p.leg_dof[l] for l in range(p.leg_count))   # (legs shorter than the longest: their padded joints read 0 on both sides)
    # Teacher forcing equalises the STATE; the launch constants (velocity / acceleration limit tables) still come from each
    # side's own init chain and agree to 1e-9 relative (1e-10 for 3- and 4-joint legs: tests/test_host_tables_and_abi.py), so
    # limited velocities - and the strides / tip targets scaled from them - inherit that relative difference.  The unconstrained
    # redundant 5-joint chain's start-up configuration differs by 7e-8 rad between any two builds (test_oracle_conditioning.py),
    # its workspace radii by 1e-10: 1e-12 m absolute elsewhere, 5e-11 m there.
    TOL_X = 5e-11 if (D == 5 and not p.gravity_aligned_tips) else 1e-12
    if TOL_X > 1e-12:
        tol_q = max(tol_q, 2e-11)  # the IK target moves with the tip targets
    gl, ol = g["leg"][:, :L], o["leg"][:, :L]
    dq = np.abs(gl["joint_position"][..., :D] - ol["joint_position"][..., :D])
    assert np.isfinite(gl["joint_position"]).all()
    assert dq.max() <= tol_q, f"max |dq| = {dq.max():.3e} rad after one cycle from identical state"
    np.testing.assert_allclose(gl["joint_velocity"][..., :D], ol["joint_velocity"][..., :D], atol=tol_q / p.time_delta * 2)
    for f in ("walker_tip", "swing_origin_tip", "stance_origin_tip", "default_tip", "target_tip", "stride_vector"):
        np.testing.assert_allclose(gl[f], ol[f], atol=TOL_X, err_msg=f)
    for f in ("walker_tip_velocity", "swing_origin_tip_velocity"):
        np.testing.assert_allclose(gl[f], ol[f], atol=TOL_X / p.time_delta * 2, err_msg=f)
    for f in ("step_state", "phase", "at_correct_phase", "completed_first_step", "ik_failed"):
        assert np.array_equal(gl[f], ol[f]), f
    for f in ("swing_progress", "stance_progress"):
        np.testing.assert_allclose(gl[f], ol[f], atol=1e-15, err_msg=f)
    if p.auto_posing and not p.imu_posing:
        assert np.array_equal(gl["negate_auto_pose"], ol["negate_auto_pose"])
    if p.admittance_control:
        np.testing.assert_allclose(gl["admittance_state"], ol["admittance_state"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(gl["admittance_delta"], ol["admittance_delta"], atol=TOL_X)
        if p.dynamic_stiffness:
            np.testing.assert_allclose(gl["virtual_stiffness"], ol["virtual_stiffness"], rtol=1e-12)
    if (features & 1) or p.use_joint_effort:
        np.testing.assert_allclose(gl["tip_force_calculated"], ol["tip_force_calculated"], rtol=1e-9, atol=1e-9)
    if p.rough_terrain_mode:
        assert np.array_equal(gl["step_plane_defined"], ol["step_plane_defined"])
        d = ol["step_plane_defined"] != 0
        np.testing.assert_allclose(gl["step_plane_position"][d], ol["step_plane_position"][d], atol=TOL_X)
        assert np.array_equal(g["touchdown_detection"], o["touchdown_detection"])
    if (p.gravity_aligned_tips or p.rough_terrain_mode) and D > 3:   # tip rotations are tracked (as their x axes)
        assert np.array_equal(gl["tip_rotation_defined"], ol["tip_rotation_defined"]), ("tip_rotation_defined", gl["tip_rotation_defined"].tolist(), ol["tip_rotation_defined"].tolist())
        d = ol["tip_rotation_defined"] != 0
        np.testing.assert_allclose(gl["walker_tip_direction"][d], ol["walker_tip_direction"][d], atol=1e-12)
        np.testing.assert_allclose(gl["origin_tip_direction"], ol["origin_tip_direction"], atol=1e-12)
        assert np.array_equal(gl["target_rotation_defined"], ol["target_rotation_defined"])
        d = ol["target_rotation_defined"] != 0
        np.testing.assert_allclose(gl["target_tip_direction"][d], ol["target_tip_direction"][d], atol=1e-12)
    legs3 = [l for l in range(L) if p.leg_dof[l] == 3]
    if p.leg_manipulation_mode == 1 and legs3 and not p.gravity_aligned_tips and not p.rough_terrain_mode:
        # joint_control: a MANUAL 3-joint leg's stepper holds the FK tip pose of the joints updateManual moved, with its rotation
        a, b = gl["tip_rotation_defined"][:, legs3], ol["tip_rotation_defined"][:, legs3]
        assert np.array_equal(a, b), ("tip_rotation_defined", a.tolist(), b.tolist())
        d = b != 0
        np.testing.assert_allclose(gl["walker_tip_direction"][:, legs3][d], ol["walker_tip_direction"][:, legs3][d], atol=1e-12)
    for f in ("desired_linear_velocity", "desired_angular_velocity", "walk_plane", "walk_plane_normal", "origin_walk_plane_pose", "current_pose"):
        np.testing.assert_allclose(g[f], o[f], atol=TOL_X, err_msg=f)
    # LegStepper::walk_plane_ is a per-leg copy taken by the legs that stepped this cycle (walk_controller.cpp:924-925); the
    # engine keeps ONE per robot (all stepping legs take the same copy; a leg in FORCE_STOP keeps a stale one that nothing
    # reads before the leg steps again).  Comparable whenever every leg stepped: walk state MOVING.
    mv = o["walk_state"] == 1
    for f in ("stepper_walk_plane", "stepper_walk_plane_normal"):
        np.testing.assert_allclose(g[f][mv], o[f][mv], atol=TOL_X, err_msg=f)
    for f in ("walk_state", "legs_at_correct_phase", "legs_completed_first_step", "return_to_default_attempted"):
        assert np.array_equal(g[f], o[f]), f
    if p.manual_posing:
        np.testing.assert_allclose(g["manual_pose"], o["manual_pose"], atol=TOL_X, err_msg="manual_pose")
        for f in ("translation_velocity_input", "rotation_velocity_input"):
            # A pose-reset mode rewrites the velocity input to +-1 from the SIGN of the remaining offset
            # (pose_controller.cpp:905-925); once the offset has been driven to zero that sign is the sign of a rounding
            # residue (1e-17), and either choice moves the pose back onto the reset target within rounding - which the
            # manual_pose comparison above holds to 1e-12.  Only such +-1 flips may differ.
            # The same holds for a residue that is exactly 0 on one side (no rewrite: the input stays) and 1e-17 on the other.
            bad = np.abs(g[f] - o[f]) > TOL_X
            assert ((np.abs(g[f][bad]) == 1.0) | (np.abs(o[f][bad]) == 1.0)).all(), f
    if p.imu_posing:
        for f in ("rotation_absement_error", "rotation_velocity_error"):
            np.testing.assert_allclose(g[f], o[f], atol=TOL_X, err_msg=f)
    elif p.auto_posing:
        assert np.array_equal(g["auto_posing_state"], o["auto_posing_state"])
        assert np.array_equal(g["auto_poser_flags"][:, :p.n_auto_posers], o["auto_poser_flags"][:, :p.n_auto_posers])
        if p.pose_frequency != -1.0:
            assert np.array_equal(g["pose_phase"], o["pose_phase"])
        if p.inclination_posing:
            np.testing.assert_allclose(g["auto_pose_rotation"], o["auto_pose_rotation"], atol=TOL_X)
    if features & 2:
        np.testing.assert_allclose(g["odometry"], o["odometry"], atol=TOL_X)
    if p.gravity_aligned_tips and D <= 3:
        for f in ("tip_align_pose", "origin_tip_align_pose"):
            np.testing.assert_allclose(g[f], o[f], atol=TOL_X, err_msg=f)
    return float(dq.max())


class Schedule:
    """Input changes applied identically to engine and oracle at given cycles."""

    def __init__(self):
        self.events = {}
        self.ignored = {}

    def at(self, cycle, **inp):
        self.events.setdefault(cycle, {}).update(inp)
        return self

    def fire(self, cycle, objs):
        inp = self.events.get(cycle)
        if not inp:
            return
        for o in objs:
            if "lin" in inp:
                o.set_velocity(inp["lin"], inp["ang"])
            if "tv" in inp:
                o.set_pose_input(inp["tv"], inp["rv"])
            if "reset" in inp:
                o.set_pose_reset_mode(inp["reset"])
            if "force" in inp:
                o.set_tip_force(inp["force"])
            if "imu_q" in inp:
                o.set_imu(inp["imu_q"], inp["gyro"])
            if "effort" in inp:
                o.set_joint_effort(inp["effort"])
            for key, which in (("ext_target", 0), ("ext_default", 1)):
                if key in inp:      # TargetTipPose message: both sides must take / ignore (robot STOPPED) the same requests
                    self.ignored.setdefault(key, []).append(o.set_external_target(inp[key], which))
                if key + "_transform" in inp:
                    o.set_external_transform(inp[key + "_transform"], which)


def teacher_forced(Engine, p, n, inp, cycles, schedule=None, features=FEAT_DEFAULT, label="", tol_q=TOL_Q):
    eng, ob = Engine(p, n), OracleBatch(p, n)
    eng.set_features(features)
    apply(eng, inp)
    apply(ob, inp)
    worst = 0.0
    visited = set()
    for c in range(cycles):
        if schedule:
            schedule.fire(c, (eng, ob))
        eng.set_state(ob.get_state())          # teacher forcing: the oracle's state, every cycle, every instance
        eng.step(1)
        ob.step(1, 8)
        g, o = as_np(eng.get_state()), as_np(ob.get_state())
        try:
            worst = max(worst, compare_records(p, features, g, o, tol_q))
        except AssertionError:
            L_, D_ = p.leg_count, p.leg_dof[0]
            d = np.abs(g["leg"]["joint_position"][:, :L_, :D_] - o["leg"]["joint_position"][:, :L_, :D_]).max(axis=(1, 2))
            dv = np.abs(g["desired_linear_velocity"] - o["desired_linear_velocity"]).max(axis=1)
            i = int(np.argmax(np.maximum(d, dv)))
            print(f"[teacher-forced {label}] FAILED at cycle {c}; worst instance {i}: |dq| {d[i]:.3e}, |dv| {dv[i]:.3e}, walk_state "
                  f"{o['walk_state'][i]}, v engine {g['desired_linear_velocity'][i]} oracle {o['desired_linear_velocity'][i]}, "
                  f"step states {o['leg']['step_state'][i, :L_]}, phases {o['leg']['phase'][i, :L_]}")
            raise
        visited.update(np.unique(o["walk_state"]).tolist())
    from conftest import parity_report
    parity_report(f"[teacher-forced {label}] {n} instances x {cycles} cycles, all instances held: max |dq| = {worst:.3e} rad, "
          f"walk states visited {sorted(visited)}")
    return eng, ob, worst


def stop_go_schedule(p, n, seed, cycles, every=60, pose=False):
    rng = np.random.default_rng(seed)
    s = Schedule()
    for c in range(every, cycles, every):
        lin, ang = rng.uniform(-0.7, 0.7, (n, 2)), rng.uniform(-1, 1, n)
        stop = rng.random(n) < 0.4
        lin[stop], ang[stop] = 0.0, 0.0
        ev = dict(lin=lin, ang=ang)
        if pose:
            ev["tv"] = rng.uniform(-1, 1, (n, 3)) * (rng.random((n, 1)) < 0.4)
            ev["rv"] = rng.uniform(-1, 1, (n, 3)) * (rng.random((n, 1)) < 0.4)
            ev["reset"] = rng.choice([0, 0, 0, 1, 2, 3, 4, 5], size=n).astype(np.int32)
        s.at(c, **ev)
    return s


# ------------------------------------------------------------------------------------------------ BASELINE configs
def test_config2_hexapod_tripod(Engine):
    p = default_hexapod_params("tripod")
    n, cycles = 256, 420
    teacher_forced(Engine, p, n, make_inputs(p, n, 0, zero_every=11), cycles, stop_go_schedule(p, n, 1, cycles, every=130), label="config2")


def config3_params():
    p = default_hexapod_params("wave")
    p.admittance_control, p.imu_posing = 1, 1
    p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    return p


def test_config3_wave_admittance_imu_as_specified(Engine):
    """configs[2] on SURVEY.md section 8(d)'s inputs: measured tip force z ~ U(0, 20) N, x, y ~ N(0, 1), resampled every 10
    cycles; IMU roll / pitch ~ U(-0.15, 0.15) rad, yaw ~ U(-pi, pi), gyro ~ N(0, 0.05).  The admittance offsets drive legs
    into their joint limits (position and velocity clamps saturate, IK-failure flags are raised) — all compared."""
    p = config3_params()
    n, cycles = 192, 600
    inp = make_inputs(p, n, 5, imu=True, force=20.0)
    rng = np.random.default_rng(0xADD1)
    s = Schedule()
    for c in range(10, cycles, 10):
        s.at(c, force=np.stack([rng.normal(0, 1, (n, 6)), rng.normal(0, 1, (n, 6)), rng.uniform(0, 20, (n, 6))], axis=2))
    eng, ob, _ = teacher_forced(Engine, p, n, inp, cycles, s, label="config3 U(0,20)N")
    lo = ob.leg_state()
    assert (lo["leg_status"] & 4).any(), "the specified forces are expected to raise IK-deviation flags somewhere"


def test_config3_dynamic_stiffness_and_stops(Engine):
    p = config3_params()
    p.dynamic_stiffness = 1
    n, cycles = 96, 500
    inp = make_inputs(p, n, 6, imu=True, force=8.0)
    teacher_forced(Engine, p, n, inp, cycles, stop_go_schedule(p, n, 66, cycles, every=90, pose=True), label="config3 dyn-stiffness")


def test_config4_octopod_ripple(Engine):
    p = synthetic_octopod_params("ripple", 5, 8)
    n, cycles = 128, 360
    teacher_forced(Engine, p, n, make_inputs(p, n, 7, zero_every=9), cycles, stop_go_schedule(p, n, 77, cycles, every=110), label="config4")


@pytest.mark.parametrize("legs,dof,gait", [(4, 3, "tripod"), (4, 4, "amble"), (6, 4, "ripple"), (8, 3, "wave"), (6, 5, "tripod"),
                                           (3, 3, "wave"), (5, 3, "ripple"), (7, 3, "wave"), (8, 4, "ripple"), (4, 5, "amble"),
                                           (8, 3, "tripod"), (6, 5, "wave")])
def test_config5_bins_and_every_kernel_instantiation(Engine, legs, dof, gait):
    p = synthetic_octopod_params(gait, dof, legs)
    n, cycles = 70, 330
    teacher_forced(Engine, p, n, make_inputs(p, n, 500 + legs * 10 + dof, zero_every=8), cycles,
                   stop_go_schedule(p, n, legs * 10 + dof, cycles, every=100, pose=True), label=f"{legs}x{dof} {gait}")


@pytest.mark.parametrize("legs,dof,gait", [(3, 3, "tripod"), (4, 4, "amble"), (5, 3, "ripple"), (7, 3, "ripple"), (8, 5, "ripple")])
def test_imu_posing_on_every_group_width(Engine, legs, dof, gait):
    """PoseController::updateIMUPose spreads its independent atan2 / sin-cos evaluations over the lanes of legs 0, 1, 2 of a robot's group
    (quat_to_euler_zyx_grouped / euler_to_quat_zyx_grouped, csrc/shc_cycle.hpp): every group width the engine has - three lanes exactly (a tripod: the
    last lane of the wavefront mirrors leg 0 of the last robot), 4, 5, 7 and 8 - with admittance on top, IMU readings with yaw anywhere in (-pi, pi]
    (the sign prediction of the yaw's atan2) renewed every 17 cycles, manual pose inputs and resets (a target rotation that is not the identity)."""
    p = synthetic_octopod_params(gait, dof, legs)
    p.admittance_control, p.imu_posing = 1, 1
    p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    n, cycles = 67, 260
    inp = make_inputs(p, n, 700 + legs, imu=True, force=6.0)
    sched = stop_go_schedule(p, n, 710 + legs, cycles, every=80, pose=True)
    for c in range(17, cycles, 17):
        fresh = make_inputs(p, n, 720 + legs * 100 + c, imu=True)
        sched.at(c, imu_q=fresh["imu_q"], gyro=fresh["gyro"])
    teacher_forced(Engine, p, n, inp, cycles, sched, label=f"IMU posing {legs}x{dof} {gait}")


# ------------------------------------------------------------------------------------------------ features
@pytest.mark.parametrize("gait", ["tripod", "wave", "ripple", "amble"])
def test_manual_pose_resets_stop_and_go(Engine, gait):
    p = default_hexapod_params(gait)
    n, cycles = 96, 520
    inp = make_inputs(p, n, 13)
    teacher_forced(Engine, p, n, inp, cycles, stop_go_schedule(p, n, 131, cycles, every=45, pose=True), label=f"manual/{gait}")


@pytest.mark.parametrize("own_clock", [False, True])
@pytest.mark.parametrize("gait", ["tripod", "wave"])
def test_auto_posing(Engine, gait, own_clock):
    p = default_hexapod_params(gait)
    p.auto_posing = 1
    if own_clock:
        p.pose_frequency = 0.8
        base = p.pose_phase_length  # READY at pose phase 0, else the reference's workspace is zero (see test_host_tables_and_abi.py)
        k = int((1.0 / p.pose_frequency) / p.time_delta / base)
        length = (k if k % 2 == 0 else k + 1) * base
        p.time_to_start = ((299 // length) * length + 1) * p.time_delta
    for i in range(p.n_auto_posers):
        p.x_amplitudes[i], p.y_amplitudes[i], p.yaw_amplitudes[i] = 0.004 * (-1) ** i, 0.003, 0.01 * (-1) ** i
        if i % 2:
            p.gravity_amplitudes[i] = 0.008
    for l in range(6):
        p.negation_transition_ratio[l] = 0.25
    n, cycles = 64, 700
    inp = make_inputs(p, n, 449, imu=True, zero_every=9)
    teacher_forced(Engine, p, n, inp, cycles, stop_go_schedule(p, n, 450, cycles, every=170), label=f"auto/{gait}/{own_clock}")


def test_inclination_and_auto_posing(Engine):
    p = default_hexapod_params("tripod")
    p.inclination_posing, p.auto_posing = 1, 1
    n, cycles = 64, 400
    teacher_forced(Engine, p, n, make_inputs(p, n, 11, imu=True), cycles, stop_go_schedule(p, n, 12, cycles, every=150), label="incl+auto")


@pytest.mark.parametrize("dof,legs,gait", [(5, 8, "ripple"), (4, 6, "tripod"), (5, 4, "amble")])
def test_gravity_aligned_tips(Engine, dof, legs, gait):
    p = synthetic_octopod_params(gait, dof, legs)
    p.gravity_aligned_tips = 1
    n, cycles = 64, 450
    inp = make_inputs(p, n, 300 + dof * 10 + legs, zero_every=7)
    teacher_forced(Engine, p, n, inp, cycles, stop_go_schedule(p, n, 301, cycles, every=120, pose=True), label=f"gravity-aligned {legs}x{dof}")


@pytest.mark.parametrize("case", ["hexapod-tripod", "hexapod-wave-force-normal-touchdown", "6x4-ripple", "no-tip-state-messages",
                                  "hexapod-tripod-wider-stance-span", "hexapod-wave-narrower-stance-span"])
def test_rough_terrain_mode(Engine, case):
    """rough_terrain_mode: the layered workspace behind the limits, Leg::touchdownDetection on every
    tip-state message (model.cpp:712-722), default tip positions re-derived at every swing / stance start, swing targets
    shifted onto the detected step surface (proactive) or pushed down by step_depth (reactive), stance-like secondary swing
    nodes once in ground contact (walk_controller.cpp:1058-1114, :1160), and the walk plane re-fitted as the defaults move."""
    if case == "6x4-ripple":
        p = synthetic_octopod_params("ripple", 4, 6)
    else:
        p = default_hexapod_params("wave" if "wave" in case else "tripod")
    p.rough_terrain_mode = 1
    p.step_depth = 0.012
    if "span" in case:   # LegStepper::calculateStanceSpanChange on the layered workspace (walk_controller.cpp:949-980): every new default tip
        p.stance_span_modifier = 0.3 if "wider" in case else -0.25   # is shifted sideways by a share of the workplane radius at ITS height
    if "normal" in case:
        p.force_normal_touchdown = 1
    n, cycles = 96, 520
    L = p.leg_count
    inp = make_inputs(p, n, 601, zero_every=10)
    sched = stop_go_schedule(p, n, 602, cycles, every=170)
    if case != "no-tip-state-messages":
        rng = np.random.default_rng(603)
        for c in range(0, cycles, 6):  # a new tip-state message every 6 cycles: contact forces come and go
            f = rng.normal(0, 0.25, (n, L, 3))
            f[..., 2] += rng.choice([0.0, 0.05, 0.6, 1.5], size=(n, L), p=[0.3, 0.2, 0.2, 0.3])
            sched.at(c, force=f)
    eng, ob, _ = teacher_forced(Engine, p, n, inp, cycles, sched, label=f"rough terrain / {case}")
    st = as_np(ob.get_state())
    if case != "no-tip-state-messages":
        assert st["leg"]["step_plane_defined"][:, :L].any() and not st["leg"]["step_plane_defined"][:, :L].all()
        # the default tips have left their identity positions and the walk plane has tilted with them
        ident = np.array([[p.stance_position[l][0], p.stance_position[l][1], 0.0] for l in range(L)])
        assert np.abs(st["leg"]["default_tip"][:, :L] - ident).max() > 1e-3
        assert np.abs(st["walk_plane"]).max() > 1e-4


def external_rows(rng, p, n, frac, z_shift=0.0, rotation="identity", odom_frac=0.5, clearance=True):
    """TargetTipPose rows for every (instance, leg): a fraction `frac` defined, poses around the legs' stance positions."""
    from syropod_highlevel_controller_amd.params import ExternalTarget
    L = p.leg_count
    rows = (ExternalTarget * (n * L))()
    for i in range(n):
        for l in range(L):
            r = rows[i * L + l]
            if rng.random() >= frac:
                continue
            r.defined = 1
            r.pose[0] = p.stance_position[l][0] + rng.uniform(-0.03, 0.03)
            r.pose[1] = p.stance_position[l][1] + rng.uniform(-0.03, 0.03)
            r.pose[2] = z_shift + rng.uniform(-0.015, 0.015)
            if rotation == "random":
                q = rng.normal(size=4)
                q /= np.linalg.norm(q)
            elif rotation == "identity":
                q = np.array([1.0, 0, 0, 0])
            else:
                q = np.zeros(4)          # UNDEFINED_ROTATION
            r.pose[3:7] = list(q)
            r.transform[:] = [0, 0, 0, 1, 0, 0, 0]   # the callback stores the identity (state_controller.cpp:1732)
            r.swing_clearance = rng.uniform(0.01, 0.05) if clearance else 0.0
            r.frame_is_odom_ideal = int(rng.random() < odom_frac)
    return rows


def random_transforms(rng, n, L, scale=0.02):
    """generateExternalTargetTransforms: the tf tree's walk-plane motion since the request (small translation + yaw / tilt)."""
    from scipy.spatial.transform import Rotation as R
    t = np.zeros((n, L, 7))
    t[..., :3] = rng.normal(0, scale, (n, L, 3))
    q = R.from_euler("xyz", rng.normal(0, 0.05, (n * L, 3))).as_quat()
    t[..., 3] = q[:, 3].reshape(n, L)
    t[..., 4:7] = q[:, :3].reshape(n, L, 3)
    return t


@pytest.mark.parametrize("case", ["6x3-tripod", "6x3-wave-forces", "8x3-ripple", "6x4-ripple-undefined-rotation", "6x4-ripple-requested-tip-rotations",
                                  "8x5-ripple-gravity-aligned-requested-tip-rotations"])
def test_rough_terrain_external_targets(Engine, case):
    """Externally requested tip targets and default stance poses (TargetTipPose messages, walk_controller.cpp:988-990,
    :1068-1079, :1159): the swing lands on pose_.removePose(transform_) with the requested clearance, targets in the
    odom_ideal frame are led by the ideal odometry, the request is dropped when the next stance begins, a requested default
    replaces the terrain-following default tip until the next one; requests to STOPPED robots never reach the stepper
    (state_controller.cpp:1736, :1746); the transforms are refreshed as the tf tree would (state_controller.cpp:703-773)."""
    if case.startswith("6x4"):
        p = synthetic_octopod_params("ripple", 4, 6)
    elif case.startswith("8x3"):
        p = synthetic_octopod_params("ripple", 3, 8)
    elif case.startswith("8x5"):
        p = synthetic_octopod_params("ripple", 5, 8)
        p.gravity_aligned_tips = 1       # the requested rotation replaces the identity tip rotation - and stays (walk_controller.cpp:1044 vs :1070)
    else:
        p = default_hexapod_params("wave" if "wave" in case else "tripod")
    p.rough_terrain_mode, p.step_depth = 1, 0.01
    n, cycles = 72, 560
    L = p.leg_count
    rng = np.random.default_rng(811)
    inp = make_inputs(p, n, 801, zero_every=7)
    sched = stop_go_schedule(p, n, 802, cycles, every=190)
    rot = "undefined" if "undefined" in case else "random"   # (> 3-DOF legs: a requested rotation drives updateTipRotation + the rotation-constrained IK)
    for c in range(30, cycles, 45):      # a TargetTipPose message every 45 cycles, for half of the legs
        sched.at(c, ext_target=external_rows(rng, p, n, 0.5, rotation=rot))
    for c in range(100, cycles, 160):    # requested stance poses now and then, a fifth of the legs (some withdrawn again)
        sched.at(c, ext_default=external_rows(rng, p, n, 0.2, rotation=rot, clearance=False))
    for c in range(33, cycles, 3):       # tf refresh
        sched.at(c, ext_target_transform=random_transforms(rng, n, L), ext_default_transform=random_transforms(rng, n, L, 0.01))
    if "forces" in case:
        for c in range(0, cycles, 6):
            f = rng.normal(0, 0.25, (n, L, 3))
            f[..., 2] += rng.choice([0.0, 0.05, 0.6, 1.5], size=(n, L), p=[0.3, 0.2, 0.2, 0.3])
            sched.at(c, force=f)
    # requested tip rotations are random: the rotation-constrained DLS step towards a direction the leg can barely reach is worse
    # conditioned than towards the gravity direction, and the engine's N x N form and the reference's 6 x 6 form round differently
    # (measured 1e-12 rad after one cycle instead of 1e-15)
    eng, ob, _ = teacher_forced(Engine, p, n, inp, cycles, sched, label=f"rough terrain + external targets / {case}",
                                tol_q=1e-10 if "requested-tip-rotations" in case else TOL_Q)
    for key in ("ext_target", "ext_default"):
        ig = sched.ignored[key]
        assert ig[0::2] == ig[1::2], "engine and oracle ignored different requests"     # (engine, oracle) pairs
        # some robots were STOPPED: their default poses are dropped, their targets go to the planner-mode LegPosers
        assert sum(ig) > 0 or key == "ext_target"
    assert any(r.defined for r in ob.get_external_target(2))
    for which in (0, 1, 2):              # the records as the steppers / posers hold them now
        a, b = eng.get_external_target(which), ob.get_external_target(which)
        assert bytes(a) == bytes(b)
    assert any(r.defined for r in ob.get_external_target(1))
    st = as_np(ob.get_state())
    ident = np.array([[p.stance_position[l][0], p.stance_position[l][1], 0.0] for l in range(L)])
    assert np.abs(st["leg"]["default_tip"][:, :L] - ident).max() > 5e-3
    if "requested-tip-rotations" in case:   # some legs carry a requested target rotation now
        assert st["leg"]["target_rotation_defined"][:, :L].any()


def test_external_target_errors(Engine):
    from syropod_highlevel_controller_amd.params import ExternalTarget
    p = default_hexapod_params("tripod")
    eng = Engine(p, 4)
    rows = (ExternalTarget * 24)()
    with pytest.raises(RuntimeError):                      # default poses are read in rough terrain mode only
        eng.set_external_target(rows, which=1)
    p = synthetic_octopod_params("ripple", 4, 6)
    p.rough_terrain_mode = 1
    eng = Engine(p, 4)
    rows = (ExternalTarget * 24)()
    assert eng.set_external_target(rows) == 0              # all undefined: withdraws nothing, ignores nothing
    rows[3].defined = 1
    rows[3].pose[3] = 1.0                                  # (a tip rotation request on 4-DOF legs is part of the target)
    assert eng.set_external_target(rows) == 0              # the robot is STOPPED: its planner-mode LegPoser takes the target
    assert [r.defined for r in eng.get_external_target(2)] == [int(k == 3) for k in range(24)]
    assert not any(r.defined for r in eng.get_external_target(0))


def test_rough_terrain_mode_free_running(Engine):
    """... and free-running from the engine's own init chain (layered workspace on the host), against the oracle."""
    from test_gpu_parity import compare
    p = default_hexapod_params("tripod")
    p.rough_terrain_mode, p.step_depth = 1, 0.01
    n = 60
    inp = make_inputs(p, n, 611, zero_every=9)
    rng = np.random.default_rng(612)
    eng, ob = Engine(p, n), OracleBatch(p, n)
    apply(eng, inp)
    apply(ob, inp)
    for k in range(40):
        f = rng.normal(0, 0.25, (n, 6, 3))
        f[..., 2] += rng.choice([0.0, 0.6, 1.5], size=(n, 6))
        for o in (eng, ob):
            o.set_tip_force(f)
        eng.step(7)
        eng.synchronize()
        ob.step(7, 8)
        compare(eng, ob, tol_q=1e-6)


@pytest.mark.parametrize("gait,legs", [("tripod", 6), ("wave", 6), ("ripple", 8), ("amble", 4)])
def test_tip_align_pose(Engine, gait, legs):
    """gravity_aligned_tips with <= 3 DOF legs: PoseController::updateTipAlignPose (pose_controller.cpp:1024-1088) shifts the body
    so that the last link of a swinging leg lines up with the walk-plane normal - the legs are visited in id order, each
    overwriting the pose its predecessor left (row a25 of SURVEY.md section 8)."""
    p = default_hexapod_params(gait) if legs == 6 else synthetic_octopod_params(gait, 3, legs)
    p.gravity_aligned_tips = 1
    p.max_translation[:] = [0.03, 0.02, 0.025]  # distinct limits: the reference clamps every axis against limit[1] above
    n, cycles = 80, 460
    inp = make_inputs(p, n, 701, zero_every=9)
    eng, ob, _ = teacher_forced(Engine, p, n, inp, cycles, stop_go_schedule(p, n, 702, cycles, every=150, pose=True), label=f"tip-align {legs}x3 {gait}")
    st = as_np(ob.get_state())
    assert np.abs(st["tip_align_pose"][:, :3]).max() > 1e-3   # the pose is doing something


def test_tip_align_pose_free_running(Engine):
    from test_gpu_parity import run_pair
    p = default_hexapod_params("tripod")
    p.gravity_aligned_tips = 1
    n = 60
    run_pair(Engine, p, n, make_inputs(p, n, 711, zero_every=8), [1, 1, 98, 150, 200], twin=True)


def _variants():
    def v(name, **kw):
        return pytest.param(kw, id=name)
    return [v("no-clamps", clamp_joint_positions=0, clamp_joint_velocities=0),
            v("100Hz-slow-steps", time_delta=0.01, step_frequency=0.6),
            v("high-clearance-tall-steps", body_clearance=0.12, swing_height=0.04, swing_width=0.01),
            v("overlapping-walkspaces", overlapping_walkspaces=1),
            v("fast-steps", step_frequency=1.6),
            v("no-posing-at-all", manual_posing=0),
            v("real-velocity-mode", velocity_input_mode=VEL_REAL),
            v("force-normal-touchdown", force_normal_touchdown=1, swing_width=0.01),
            v("stance-span", stance_span_modifier=0.25),
            v("admittance-from-joint-efforts", admittance_control=1, use_joint_effort=1, force_gain=0.05),
            v("dynamic-stiffness-scalers", admittance_control=1, dynamic_stiffness=1, load_stiffness_scaler=3.0, swing_stiffness_scaler=0.2),
            v("stiff-virtual-model", admittance_control=1, virtual_stiffness=30.0, virtual_mass=5.0, virtual_damping_ratio=1.2,
              force_gain=0.02, dynamic_stiffness=0)]


@pytest.mark.parametrize("kw", _variants())
def test_parameter_variants(Engine, kw):
    """One-step parity does not depend on where the start-up solve ended: the engine starts from its OWN init chain."""
    p = default_hexapod_params("ripple")
    for k, val in kw.items():
        setattr(p, k, val)
    n, cycles = 64, 380
    inp = make_inputs(p, n, 211, force=6.0 if p.admittance_control else None)
    if p.velocity_input_mode == VEL_REAL:
        inp["lin"] *= 0.12
        inp["ang"] *= 0.6
    teacher_forced(Engine, p, n, inp, cycles, stop_go_schedule(p, n, 212, cycles, every=120), label=str(kw))


def test_optional_features_off(Engine):
    p = default_hexapod_params("tripod")
    n, cycles = 64, 200
    teacher_forced(Engine, p, n, make_inputs(p, n, 3), cycles, features=0, label="features off")


# ------------------------------------------------------------------------------------------------ checkpoint / restore
@pytest.mark.parametrize("case", ["config2", "config3", "octopod-gravity"])
def test_engine_snapshot_restores_bit_exactly(Engine, case):
    """shc_engine_get_state -> a FRESH engine -> shc_engine_set_state continues bit for bit; the same snapshot loaded into
    the oracle (orc_set_state) continues within the free-running tolerance."""
    if case == "config2":
        p, kw = default_hexapod_params("tripod"), {}
    elif case == "config3":
        p, kw = config3_params(), dict(imu=True, force=4.0)
    else:
        p, kw = synthetic_octopod_params("ripple", 5, 8), {}
        p.gravity_aligned_tips = 1
    n = 77
    inp = make_inputs(p, n, 91, **kw)
    a = Engine(p, n)
    apply(a, inp)
    a.step(173)
    snap = a.get_state()
    b = Engine(p, n)
    apply(b, inp)
    b.set_state(snap)
    assert bytes(b.get_state()) == bytes(snap)          # the record survives the round trip unchanged
    ob = OracleBatch(p, n)
    apply(ob, inp)
    ob.set_state(snap)
    for k in (1, 1, 40):
        a.step(k)
        b.step(k)
        ob.step(k, 8)
        assert bytes(a.get_state()) == bytes(b.get_state())
        qa, _ = a.joints()
        qo, _ = ob.joints()
        assert np.abs(qa - qo).max() <= 1e-9


def test_partial_state_access(Engine):
    p = default_hexapod_params("tripod")
    n = 50
    e = Engine(p, n)
    inp = make_inputs(p, n, 5)
    apply(e, inp)
    e.step(60)
    full = as_np(e.get_state()).copy()
    part = as_np(e.get_state(17, 9))
    assert part.tobytes() == full[17:26].tobytes()
    # overwrite instances 3..5 with the state of instances 30..32: only those change
    donor = e.get_state(30, 3)
    e.set_state(donor, first=3)
    after = as_np(e.get_state())
    assert after[3:6].tobytes() == full[30:33].tobytes()
    assert after[:3].tobytes() == full[:3].tobytes() and after[6:].tobytes() == full[6:].tobytes()
    from syropod_highlevel_controller_amd.engine import ShcError
    with pytest.raises(ShcError):
        e.get_state(45, 10)


# ------------------------------------------------------------------------------------------------ soak
@pytest.mark.parametrize("variant", ["tripod-manual", "wave-admittance-imu", "ripple-8x5", "amble-auto-pose"])
def test_soak_teacher_forced(Engine, variant):
    """Long random command schedules (velocity changes and stops every ~50 cycles, manual pose inputs and reset modes, new tip
    forces / IMU readings), every instance held to the one-step bar in every cycle - no well-posedness mask.  SHC_SOAK_CYCLES
    sets the length (1 500 in the regular suite; 20 000 passes for every variant)."""
    cycles = int(os.environ.get("SHC_SOAK_CYCLES", "1500"))
    n = 48
    kw = {}
    if variant == "tripod-manual":
        p = default_hexapod_params("tripod")
    elif variant == "wave-admittance-imu":
        p = default_hexapod_params("wave")
        p.admittance_control, p.imu_posing, p.dynamic_stiffness = 1, 1, 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
        kw = dict(imu=True, force=20.0)
    elif variant == "ripple-8x5":
        p = synthetic_octopod_params("ripple", 5, 8)
    else:
        p = default_hexapod_params("amble")
        p.auto_posing = 1
    inp = make_inputs(p, n, 900, zero_every=11, **kw)
    sched = stop_go_schedule(p, n, 901, cycles, every=53, pose=(variant != "wave-admittance-imu"))
    if kw:
        rng = np.random.default_rng(902)
        for c in range(10, cycles, 10):      # tip forces U(0, 20) N resampled every 10 cycles, a new IMU reading every 37
            f = np.stack([rng.normal(0, 1, (n, 6)), rng.normal(0, 1, (n, 6)), rng.uniform(0, 20.0, (n, 6))], axis=2)
            sched.at(c, force=f)
        for c in range(37, cycles, 37):
            fresh = make_inputs(p, n, 1000 + c, imu=True)
            sched.at(c, imu_q=fresh["imu_q"], gyro=fresh["gyro"])
    teacher_forced(Engine, p, n, inp, cycles, sched, label=f"soak {variant}")

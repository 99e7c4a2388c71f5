"""Parameter / table structs of the C ABI (include/shc_batch.h) as ctypes, plus the morphology and gait
fixtures the reference ships.

The numbers in :func:`default_hexapod_params` restate ``config/default.yaml`` (model: lines 25-78, walker:
82-106, poser: 110-118, admittance: 122-130), ``config/gait.yaml`` and ``config/auto_pose.yaml`` of the
reference (OpenSHC v0.5.11); they are configuration data (fixtures), cited per field below.

The reference ships no 8-leg / 5-DOF model: :func:`synthetic_octopod_params` is a frozen synthetic morphology
defined here (SURVEY.md §8d "Config 4").
"""
from __future__ import annotations

import ctypes as C
import math

SHC_MAX_LEGS = 8
SHC_MAX_JOINTS = 6
SHC_MAX_LINKS = 7
SHC_MAX_AUTO_POSERS = 8
SHC_N_BEARINGS = 9

# enums of include/shc_batch.h
WALK_STARTING, WALK_MOVING, WALK_STOPPING, WALK_STOPPED = 0, 1, 2, 3
STEP_SWING, STEP_STANCE, STEP_FORCE_STANCE, STEP_FORCE_STOP = 0, 1, 2, 3
VEL_THROTTLE, VEL_REAL = 0, 1
FEAT_TIP_FORCE = 1
FEAT_ODOMETRY = 2
FEAT_GENERIC_KERNEL = 1 << 30  # diagnostic: runtime-flag kernel instead of the compile-time specialisation
FEAT_SINGLE_STREAM = 1 << 28  # diagnostic: no two-stream split of large batches
FEAT_RESIDENT_ONE_WAVE = 1 << 29  # diagnostic: resident mode without the two-wavefront (walker / model) pipeline
FEAT_STEP_K_SERIAL = 1 << 27  # diagnostic: shc_engine_step_k as K single launches (the form configurations without a batch kernel take)
FEAT_DEFAULT = FEAT_TIP_FORCE | FEAT_ODOMETRY  # what shc_engine_create enables


class LegStateMsg(C.Structure):
    """shc_leg_state_msg of include/shc_batch.h (numeric payload of LegState.msg)."""
    _fields_ = [("walker_tip_position", C.c_double * 3), ("target_tip_position", C.c_double * 3),
                ("poser_tip_position", C.c_double * 3), ("model_tip_position", C.c_double * 3),
                ("actual_tip_pose", C.c_double * 7), ("model_tip_velocity", C.c_double * 3), ("joint_positions", C.c_double * SHC_MAX_JOINTS), ("joint_velocities", C.c_double * SHC_MAX_JOINTS),
                ("joint_efforts", C.c_double * SHC_MAX_JOINTS), ("stance_progress", C.c_double), ("swing_progress", C.c_double),
                ("time_to_swing_end", C.c_double), ("pose_delta", C.c_double * 7), ("auto_pose", C.c_double * 7), ("tip_force", C.c_double * 3),
                ("admittance_delta", C.c_double * 3), ("virtual_stiffness", C.c_double)]


class ExternalTarget(C.Structure):
    """shc_external_target of include/shc_batch.h (struct ExternalTarget, walk_controller.h:38-46)."""
    _fields_ = [("pose", C.c_double * 7), ("transform", C.c_double * 7), ("swing_clearance", C.c_double),
                ("frame_is_odom_ideal", C.c_int32), ("defined", C.c_int32)]


EXTERNAL_TARGET, EXTERNAL_DEFAULT = 0, 1


class LegSnapshot(C.Structure):
    """shc_leg_snapshot of include/shc_batch.h."""
    _fields_ = [("joint_position", C.c_double * SHC_MAX_JOINTS), ("joint_velocity", C.c_double * SHC_MAX_JOINTS),
                ("walker_tip", C.c_double * 3), ("walker_tip_velocity", C.c_double * 3), ("swing_origin_tip", C.c_double * 3),
                ("swing_origin_tip_velocity", C.c_double * 3), ("stance_origin_tip", C.c_double * 3), ("default_tip", C.c_double * 3),
                ("target_tip", C.c_double * 3), ("stride_vector", C.c_double * 3), ("walker_tip_direction", C.c_double * 3),
                ("origin_tip_direction", C.c_double * 3), ("admittance_state", C.c_double * 2), ("admittance_delta", C.c_double * 3),
                ("virtual_stiffness", C.c_double), ("tip_force_calculated", C.c_double * 3), ("swing_progress", C.c_double),
                ("stance_progress", C.c_double), ("step_state", C.c_int32), ("phase", C.c_int32), ("at_correct_phase", C.c_int32),
                ("completed_first_step", C.c_int32), ("negate_auto_pose", C.c_int32), ("ik_failed", C.c_int32),
                ("tip_rotation_defined", C.c_int32), ("step_plane_defined", C.c_int32), ("step_plane_position", C.c_double * 3),
                ("target_tip_direction", C.c_double * 3), ("target_rotation_defined", C.c_int32), ("pad_", C.c_int32)]


class InstanceState(C.Structure):
    """shc_instance_state of include/shc_batch.h: full controller state of one robot (checkpoint / state injection)."""
    _fields_ = [("desired_linear_velocity", C.c_double * 2), ("desired_angular_velocity", C.c_double),
                ("walk_plane", C.c_double * 3), ("walk_plane_normal", C.c_double * 3), ("stepper_walk_plane", C.c_double * 3),
                ("stepper_walk_plane_normal", C.c_double * 3), ("origin_walk_plane_pose", C.c_double * 7), ("manual_pose", C.c_double * 7),
                ("translation_velocity_input", C.c_double * 3), ("rotation_velocity_input", C.c_double * 3),
                ("rotation_absement_error", C.c_double * 3), ("rotation_velocity_error", C.c_double * 3),
                ("auto_pose_rotation", C.c_double * 4), ("current_pose", C.c_double * 7), ("odometry", C.c_double * 7),
                ("tip_align_pose", C.c_double * 7), ("origin_tip_align_pose", C.c_double * 7),
                ("walk_state", C.c_int32), ("legs_at_correct_phase", C.c_int32), ("legs_completed_first_step", C.c_int32),
                ("return_to_default_attempted", C.c_int32), ("auto_posing_state", C.c_int32), ("pose_phase", C.c_int32),
                ("auto_poser_flags", C.c_int32 * SHC_MAX_AUTO_POSERS), ("touchdown_detection", C.c_int32), ("pad_", C.c_int32),
                ("leg", LegSnapshot * SHC_MAX_LEGS)]


class JointParams(C.Structure):
    _fields_ = [("min", C.c_double), ("max", C.c_double), ("offset", C.c_double), ("unpacked", C.c_double),
                ("max_vel", C.c_double)]


class LinkParams(C.Structure):
    _fields_ = [("d", C.c_double), ("theta", C.c_double), ("r", C.c_double), ("alpha", C.c_double)]


class Params(C.Structure):
    """``shc_params`` (include/shc_batch.h)."""
    _fields_ = [
        ("time_delta", C.c_double),
        ("manual_posing", C.c_int32), ("auto_posing", C.c_int32), ("rough_terrain_mode", C.c_int32),
        ("admittance_control", C.c_int32), ("inclination_posing", C.c_int32), ("imu_posing", C.c_int32),
        ("leg_count", C.c_int32),
        ("leg_dof", C.c_int32 * SHC_MAX_LEGS),
        ("joint", (JointParams * SHC_MAX_JOINTS) * SHC_MAX_LEGS),
        ("link", (LinkParams * SHC_MAX_LINKS) * SHC_MAX_LEGS),
        ("clamp_joint_positions", C.c_int32), ("clamp_joint_velocities", C.c_int32),
        ("body_clearance", C.c_double), ("step_frequency", C.c_double), ("swing_height", C.c_double),
        ("swing_width", C.c_double), ("step_depth", C.c_double), ("stance_span_modifier", C.c_double),
        ("touchdown_threshold", C.c_double), ("liftoff_threshold", C.c_double),
        ("velocity_input_mode", C.c_int32),
        ("stance_position", (C.c_double * 2) * SHC_MAX_LEGS),
        ("overlapping_walkspaces", C.c_int32), ("force_normal_touchdown", C.c_int32),
        ("gravity_aligned_tips", C.c_int32), ("leg_manipulation_mode", C.c_int32),
        ("time_to_start", C.c_double),
        ("rotation_pid_gains", C.c_double * 3),
        ("max_translation", C.c_double * 3),
        ("max_rotation", C.c_double * 3),
        ("max_translation_velocity", C.c_double), ("max_rotation_velocity", C.c_double),
        ("dynamic_stiffness", C.c_int32), ("use_joint_effort", C.c_int32),
        ("integrator_step_time", C.c_double), ("virtual_mass", C.c_double), ("virtual_stiffness", C.c_double),
        ("virtual_damping_ratio", C.c_double), ("force_gain", C.c_double),
        ("load_stiffness_scaler", C.c_double), ("swing_stiffness_scaler", C.c_double),
        ("stance_phase", C.c_int32), ("swing_phase", C.c_int32), ("phase_offset", C.c_int32),
        ("offset_multiplier", C.c_int32 * SHC_MAX_LEGS),
        ("pose_frequency", C.c_double),
        ("pose_phase_length", C.c_int32),
        ("n_auto_posers", C.c_int32),
        ("pose_phase_starts", C.c_int32 * SHC_MAX_AUTO_POSERS),
        ("pose_phase_ends", C.c_int32 * SHC_MAX_AUTO_POSERS),
        ("pose_negation_phase_starts", C.c_int32 * SHC_MAX_LEGS),
        ("pose_negation_phase_ends", C.c_int32 * SHC_MAX_LEGS),
        ("negation_transition_ratio", C.c_double * SHC_MAX_LEGS),
        ("roll_amplitudes", C.c_double * SHC_MAX_AUTO_POSERS),
        ("pitch_amplitudes", C.c_double * SHC_MAX_AUTO_POSERS),
        ("yaw_amplitudes", C.c_double * SHC_MAX_AUTO_POSERS),
        ("x_amplitudes", C.c_double * SHC_MAX_AUTO_POSERS),
        ("y_amplitudes", C.c_double * SHC_MAX_AUTO_POSERS),
        ("z_amplitudes", C.c_double * SHC_MAX_AUTO_POSERS),
        ("gravity_amplitudes", C.c_double * SHC_MAX_AUTO_POSERS),
    ]

    def dof_total(self) -> int:
        return sum(self.leg_dof[l] for l in range(self.leg_count))

    def copy(self) -> "Params":
        out = Params()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(Params))
        return out


class StepCycle(C.Structure):
    """``shc_step_cycle`` (walk_controller.h:23-33 in the reference)."""
    _fields_ = [("frequency", C.c_double), ("period", C.c_int32), ("swing_period", C.c_int32),
                ("stance_period", C.c_int32), ("stance_end", C.c_int32), ("swing_start", C.c_int32),
                ("swing_end", C.c_int32), ("stance_start", C.c_int32)]


class Tables(C.Structure):
    """``shc_tables``."""
    _fields_ = [
        ("step", StepCycle),
        ("phase_offset", C.c_int32 * SHC_MAX_LEGS),
        ("default_joint_position", (C.c_double * SHC_MAX_JOINTS) * SHC_MAX_LEGS),
        ("walkspace", C.c_double * SHC_N_BEARINGS),
        ("max_linear_speed", C.c_double * SHC_N_BEARINGS),
        ("max_angular_speed", C.c_double * SHC_N_BEARINGS),
        ("max_linear_acceleration", C.c_double * SHC_N_BEARINGS),
        ("max_angular_acceleration", C.c_double * SHC_N_BEARINGS),
        ("workspace_radius", (C.c_double * SHC_N_BEARINGS) * SHC_MAX_LEGS),
        ("pose_phase_length", C.c_int32), ("pose_normaliser", C.c_int32),
        ("auto_pose_reference_leg", C.c_int32),
    ]


# --------------------------------------------------------------------------------------------- gait fixtures
# config/gait.yaml (leg order AR, BR, CR, CL, BL, AL = default.yaml:26)
GAITS = {
    "wave": dict(stance_phase=10, swing_phase=2, phase_offset=2, offset_multiplier=[2, 3, 4, 1, 0, 5]),
    "tripod": dict(stance_phase=2, swing_phase=2, phase_offset=2, offset_multiplier=[0, 1, 0, 1, 0, 1]),
    "ripple": dict(stance_phase=4, swing_phase=2, phase_offset=1, offset_multiplier=[2, 0, 4, 1, 3, 5]),
    "amble": dict(stance_phase=2, swing_phase=1, phase_offset=1, offset_multiplier=[1, 2, 0, 1, 2, 0]),
}

# config/auto_pose.yaml (per-leg maps given in leg order AR, BR, CR, CL, BL, AL)
AUTO_POSES = {
    "wave": dict(pose_frequency=-1.0, pose_phase_length=12,
                 pose_phase_starts=[1, 3, 5, 7, 9, 11], pose_phase_ends=[3, 5, 7, 9, 11, 1],
                 pose_negation_phase_starts=[1, 11, 9, 3, 5, 7], pose_negation_phase_ends=[3, 1, 11, 5, 7, 9],
                 negation_transition_ratio=[0, 0, 0, 0, 0, 0],
                 roll=[-0.015, 0.015, 0.015, 0.015, -0.015, -0.015],
                 pitch=[0.020, -0.020, 0.000, 0.020, -0.020, 0.000],
                 yaw=[0.0] * 6, x=[0.0] * 6, y=[0.0] * 6, z=[0.0] * 6, gravity=[0.0] * 6),
    "tripod": dict(pose_frequency=-1.0, pose_phase_length=4,
                   pose_phase_starts=[1, 3], pose_phase_ends=[3, 1],
                   pose_negation_phase_starts=[1, 3, 1, 3, 1, 3], pose_negation_phase_ends=[3, 1, 3, 1, 3, 1],
                   negation_transition_ratio=[0, 0, 0, 0, 0, 0],
                   roll=[-0.015, 0.015], pitch=[0.0, 0.0], yaw=[0.0, 0.0], x=[0.0, 0.0], y=[0.0, 0.0],
                   z=[0.020, 0.020], gravity=[0.0, 0.0]),
    "ripple": dict(pose_frequency=-1.0, pose_phase_length=6,
                   pose_phase_starts=[0, 1, 2, 3, 4, 5], pose_phase_ends=[2, 3, 4, 5, 0, 1],
                   pose_negation_phase_starts=[0, 2, 4, 1, 5, 3], pose_negation_phase_ends=[2, 4, 0, 3, 1, 5],
                   negation_transition_ratio=[0, 0, 0, 0, 0, 0],
                   roll=[-0.015, 0.015, -0.015, 0.015, -0.015, 0.015],
                   pitch=[-0.020, 0.020, 0.000, -0.020, 0.020, 0.000],
                   yaw=[0.0] * 6, x=[0.0] * 6, y=[0.0] * 6, z=[0.0] * 6, gravity=[0.0] * 6),
    "amble": dict(pose_frequency=-1.0, pose_phase_length=3,
                  pose_phase_starts=[0, 1, 2], pose_phase_ends=[1, 2, 0],
                  pose_negation_phase_starts=[0, 2, 1, 0, 2, 1], pose_negation_phase_ends=[1, 0, 2, 1, 0, 2],
                  negation_transition_ratio=[0, 0, 0, 0, 0, 0],
                  roll=[0.0] * 3, pitch=[0.0] * 3, yaw=[0.0] * 3, x=[0.0] * 3, y=[0.0] * 3, z=[0.0] * 3,
                  gravity=[0.0] * 3),
}


def _set_gait(p: Params, gait: str, n_legs: int, offset_multiplier=None) -> None:
    g = GAITS[gait]
    p.stance_phase, p.swing_phase, p.phase_offset = g["stance_phase"], g["swing_phase"], g["phase_offset"]
    om = offset_multiplier if offset_multiplier is not None else g["offset_multiplier"]
    for l in range(n_legs):
        p.offset_multiplier[l] = om[l]


def _set_auto_pose(p: Params, gait: str, n_legs: int) -> None:
    a = AUTO_POSES[gait]
    p.pose_frequency = a["pose_frequency"]
    p.pose_phase_length = a["pose_phase_length"]
    n = len(a["pose_phase_starts"])
    p.n_auto_posers = n
    for i in range(n):
        p.pose_phase_starts[i] = a["pose_phase_starts"][i]
        p.pose_phase_ends[i] = a["pose_phase_ends"][i]
        p.roll_amplitudes[i] = a["roll"][i]
        p.pitch_amplitudes[i] = a["pitch"][i]
        p.yaw_amplitudes[i] = a["yaw"][i]
        p.x_amplitudes[i] = a["x"][i]
        p.y_amplitudes[i] = a["y"][i]
        p.z_amplitudes[i] = a["z"][i]
        p.gravity_amplitudes[i] = a["gravity"][i]
    for l in range(n_legs):
        p.pose_negation_phase_starts[l] = a["pose_negation_phase_starts"][l % 6]
        p.pose_negation_phase_ends[l] = a["pose_negation_phase_ends"][l % 6]
        p.negation_transition_ratio[l] = a["negation_transition_ratio"][l % 6]


def _common(p: Params) -> None:
    # default.yaml:9-15
    p.time_delta = 0.02
    p.manual_posing, p.auto_posing, p.rough_terrain_mode = 1, 0, 0
    p.admittance_control, p.inclination_posing, p.imu_posing = 0, 0, 0
    # default.yaml:76-78
    p.clamp_joint_positions, p.clamp_joint_velocities = 1, 1
    # default.yaml:82-106
    p.body_clearance = 0.100
    p.step_frequency = 1.000
    p.swing_height = 0.020
    p.swing_width = 0.000
    p.step_depth = 0.000
    p.touchdown_threshold, p.liftoff_threshold = 0.9, 0.1  # default.yaml:107-108
    p.stance_span_modifier = 0.000
    p.velocity_input_mode = VEL_THROTTLE
    p.overlapping_walkspaces, p.force_normal_touchdown, p.gravity_aligned_tips = 0, 0, 0
    # default.yaml:110-118
    p.time_to_start = 6.000
    p.rotation_pid_gains[:] = [0.0, 0.0, 0.0]
    p.max_translation[:] = [0.025, 0.025, 0.025]
    p.max_rotation[:] = [0.250, 0.250, 0.250]
    p.max_translation_velocity = 0.050
    p.max_rotation_velocity = 0.200
    # default.yaml:122-130
    p.dynamic_stiffness, p.use_joint_effort = 1, 0
    p.integrator_step_time = 0.500
    p.virtual_mass = 10.00
    p.virtual_stiffness = 12.00
    p.virtual_damping_ratio = 0.800
    p.force_gain = 0.100
    p.load_stiffness_scaler = 5.000
    p.swing_stiffness_scaler = 0.100


def default_hexapod_params(gait: str = "tripod") -> Params:
    """The 6-leg x 3-DOF model of ``config/default.yaml`` with a gait of ``config/gait.yaml``."""
    p = Params()
    _common(p)
    p.leg_count = 6
    # default.yaml:31-48: identical joint limits on every leg (coxa, femur, tibia)
    joints = [(-0.550, 0.550, 0.0, 0.000, 5.0), (-1.500, 1.500, 0.0, 0.785, 5.0), (-2.355, -0.100, 0.0, -1.138, 5.0)]
    # default.yaml:51-74: base-link theta per leg, AR BR CR CL BL AL
    base_theta = [-0.523, -1.571, -2.617, 2.617, 1.571, 0.523]
    # default.yaml:97-102
    stance = [(0.130, -0.075), (0.000, -0.150), (-0.130, -0.075), (-0.130, 0.075), (0.000, 0.150), (0.130, 0.075)]
    for l in range(6):
        p.leg_dof[l] = 3
        for j, (mn, mx, off, unp, mv) in enumerate(joints):
            p.joint[l][j] = JointParams(mn, mx, off, unp, mv)
        p.link[l][0] = LinkParams(0.0, base_theta[l], 0.050, 0.0)      # base
        p.link[l][1] = LinkParams(0.0, 0.0, 0.050, 1.571)             # coxa
        p.link[l][2] = LinkParams(0.0, 0.0, 0.050, 0.0)               # femur
        p.link[l][3] = LinkParams(0.0, -0.100, 0.100, 0.0)            # tibia
        p.stance_position[l][0], p.stance_position[l][1] = stance[l]
    _set_gait(p, gait, 6)
    _set_auto_pose(p, gait, 6)
    return p


def synthetic_octopod_params(gait: str = "ripple", dof: int = 5, n_legs: int = 8) -> Params:
    """Frozen SYNTHETIC morphology (the reference ships none with 8 legs or > 3 DOF):
    ``n_legs`` legs evenly spaced clockwise from the front-right, each the default coxa-femur-tibia chain
    extended with ``dof - 3`` extra revolute pitch links (tarsus, tip) of 0.04 m.  Gait offset multipliers extend
    gait.yaml's pattern to ``n_legs`` legs (successive legs ``i * k mod n``)."""
    assert 3 <= dof <= 5 and 3 <= n_legs <= SHC_MAX_LEGS
    p = Params()
    _common(p)
    p.leg_count = n_legs
    base_joints = [(-0.550, 0.550, 0.0, 0.000, 5.0), (-1.500, 1.500, 0.0, 0.785, 5.0), (-2.355, -0.100, 0.0, -1.138, 5.0)]
    extra_joint = (-1.200, 1.200, 0.0, 0.000, 5.0)
    radius_body, radius_stance = 0.050, 0.170
    for l in range(n_legs):
        # clockwise from front right: bearing of leg l (radians), mirroring default.yaml's -0.523 .. +0.523
        ang = -math.pi / n_legs - l * (2.0 * math.pi / n_legs)
        ang = (ang + math.pi) % (2.0 * math.pi) - math.pi
        ang = round(ang, 3)
        p.leg_dof[l] = dof
        for j in range(dof):
            mn, mx, off, unp, mv = base_joints[j] if j < 3 else extra_joint
            p.joint[l][j] = JointParams(mn, mx, off, unp, mv)
        p.link[l][0] = LinkParams(0.0, ang, radius_body, 0.0)
        p.link[l][1] = LinkParams(0.0, 0.0, 0.050, 1.571)
        p.link[l][2] = LinkParams(0.0, 0.0, 0.050, 0.0)
        if dof == 3:
            p.link[l][3] = LinkParams(0.0, -0.100, 0.100, 0.0)
        else:
            p.link[l][3] = LinkParams(0.0, -0.100, 0.060, 0.0)
            for j in range(4, dof + 1):
                p.link[l][j] = LinkParams(0.0, 0.0, 0.040 if j < dof else 0.040 - 0.02 * (dof - 4), 0.0)
        p.stance_position[l][0] = round(radius_stance * math.cos(ang), 3)
        p.stance_position[l][1] = round(radius_stance * math.sin(ang), 3)
    g = GAITS[gait]
    period = g["stance_phase"] + g["swing_phase"]
    n_slots = max(1, period // g["phase_offset"])
    if gait == "tripod":
        om = [l % 2 for l in range(n_legs)]
    elif gait == "amble":
        om = [(l + 1) % 3 for l in range(n_legs)]
    elif gait == "ripple":
        # adjacent legs' offsets differ by >= 2 slots (the swing length), as in gait.yaml's hexapod ripple
        om = [2, 0, 4, 1, 3, 5, 1, 4][:n_legs] if n_legs <= 8 and n_slots == 6 else [(2 + 3 * l) % n_slots for l in range(n_legs)]
    else:  # wave
        om = [(2 + l) % n_slots for l in range(n_legs)]
    _set_gait(p, gait, n_legs, om)
    _set_auto_pose(p, gait, n_legs)
    return p


def synthetic_mixed_dof_params(gait: str = "ripple", dofs=(3, 5, 4, 3, 5, 4)) -> Params:
    """A SYNTHETIC robot whose legs differ in DOF (``Parameters::leg_DOF`` is per leg, parameters_and_states.h:298; BASELINE.json config 5
    "3-5 DOF per leg"): leg ``l`` is leg ``l`` of ``synthetic_octopod_params(gait, dofs[l], len(dofs))`` - same hips and stance
    positions, each leg its own chain."""
    n_legs = len(dofs)
    p = synthetic_octopod_params(gait, max(dofs), n_legs)
    for l, d in enumerate(dofs):
        src = synthetic_octopod_params(gait, d, n_legs)
        p.leg_dof[l] = d
        for j in range(SHC_MAX_JOINTS):
            p.joint[l][j] = src.joint[l][j]
        for j in range(SHC_MAX_JOINTS + 1):
            p.link[l][j] = src.link[l][j]
    return p


# enum ParameterSelection (parameters_and_states.h:165-178) = SHC_PARAM_* of include/shc_batch.h: the run-time adjustable parameters
(PARAM_STEP_FREQUENCY, PARAM_SWING_HEIGHT, PARAM_SWING_WIDTH, PARAM_STEP_DEPTH, PARAM_STANCE_SPAN_MODIFIER, PARAM_VIRTUAL_MASS, PARAM_VIRTUAL_STIFFNESS,
 PARAM_VIRTUAL_DAMPING, PARAM_FORCE_GAIN) = range(1, 10)
PARAM_FIELD = {PARAM_STEP_FREQUENCY: "step_frequency", PARAM_SWING_HEIGHT: "swing_height", PARAM_SWING_WIDTH: "swing_width", PARAM_STEP_DEPTH: "step_depth",
               PARAM_STANCE_SPAN_MODIFIER: "stance_span_modifier", PARAM_VIRTUAL_MASS: "virtual_mass", PARAM_VIRTUAL_STIFFNESS: "virtual_stiffness",
               PARAM_VIRTUAL_DAMPING: "virtual_damping_ratio", PARAM_FORCE_GAIN: "force_gain"}

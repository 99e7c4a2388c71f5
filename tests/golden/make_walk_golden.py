"""Generates tests/golden/walk_golden.npz: multi-cycle trajectories of the walking part of the reference path from an
INDEPENDENT numpy restatement (this file) - no code shared with oracle/ or the engine.

    python tests/golden/make_walk_golden.py          (needs numpy + scipy; writes the fixture next to this file)

Restated here, from the reference sources only (OpenSHC v0.5.11, paths relative to /root/reference):
  WalkController::generateStepCycle        src/walk_controller.cpp:365-410
  phase offsets of generateLimits          src/walk_controller.cpp:277-285
  WalkController::getLimit                 src/walk_controller.cpp:414-436
  WalkController::updateWalk (+ FSM)       src/walk_controller.cpp:440-648
  WalkController::updateWalkPlane          src/walk_controller.cpp:748-779
  StateController::changeGait              src/state_controller.cpp:513-538   (stop the robot, then a new step cycle + limits)
  LegStepper (iteratePhase, updateStepState, updateStride, updateTipPosition, control nodes, updateDefaultTipPosition)
                                           src/walk_controller.cpp:871-1189, 1238-1329
    incl. the rough-terrain branches that do not need the kinematic model: default-tip update at every swing / stance
    start (:1058-1061, :1160-1163), external target / default (:988-990, :1068-1079; struct ExternalTarget,
    walk_controller.h:38-46; Pose::removePose, pose.h:178; WalkController::calculateOdometry, :783-791), the reactive
    step-depth target (:1099-1102); and, where the scenario runs the kinematic model, the branches that need it: touchdown
    detection (Leg::touchdownDetection, src/model.cpp:712-722, from tipStatesCallback, src/state_controller.cpp:1618-1648), the
    proactive target shift onto the sensed step plane (:1082-1096), the stance-like secondary swing nodes on ground contact
    (:1296-1303)
  quarticBezier / quarticBezierDot         include/.../standard_includes.h:402-420
  PoseController::updateWalkPlanePose / updateAutoPose / updateIMUPose   src/pose_controller.cpp:1092-1236
  PoseController::updateManualPose (velocity inputs, limits, the reset modes) / updateInclinationPose   :863-1003, :1240-1259
  PoseController::updateTipAlignPose (gravity_aligned_tips on <= 3-joint legs; with the kinematic model)   :1024-1088
  AutoPoser::updatePose                    src/pose_controller.cpp:1338-1439
  Pose::addPose / interpolate              include/.../pose.h:167-195
  Model::updateModel for the scenarios that carry joints (`model` in their overrides): PoseController::updateStance
  (src/pose_controller.cpp:110-141), Leg::setDesiredTipPose (src/model.cpp:653-663), Leg::applyIK -> solveIK + updateJointPositions
  (:726-857, the reference's 6x6 damped-least-squares form with the joint-limit cost gradient, numpy.linalg.inv for its LU
  inverse) - the whole control cycle of BASELINE.json config 2, free-running from the recorded start-up joints - and, for the
  scenario with admittance_control, AdmittanceController::updateAdmittance (src/admittance_controller.cpp:22-63: 30 RK4 steps per
  axis on the leg's ONE shared state, clamp, deadband) with Leg::setAdmittanceDelta's projection onto the tip axis (model.h:365-368):
  config 3's path (wave gait + admittance + IMU posing); and for gravity_aligned_tips on > 3-joint legs LegStepper::updateTipRotation
  (src/walk_controller.cpp:1193-1234, held as tip DIRECTIONS - the only thing any consumer reads) with Leg::applyIK's rotation-constrained
  pass (src/model.cpp:880-900: simulated position update, rotation delta from the PRE-update tip direction, 6-row solve) and its
  unconstrained retry (:932-936)
Nothing is fed in from oracle/ or the product any more: the joint state the robot has after its direct start-up and the velocity /
acceleration limit tables (the IK-based workspace search) come from the numpy restatement of the init chain in make_init_golden.py
(every scenario runs with time_to_start = 2 s: 100 start-up steps, where that iteration is well-posed).
Rotations use scipy.spatial.transform.Rotation (an independent implementation of the Euler / quaternion conventions).

The fixture cannot lift "parity unpinned" - the reference ships no vectors - but a misreading of the reference shared by the
oracle and the kernel (which were written together) would have to be made a third time, in another language and structure,
to go unnoticed.  tests/test_oracle_golden.py::test_walk_trajectories replays every scenario on the oracle.
"""
import json
import math
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation as R

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

STARTING, MOVING, STOPPING, STOPPED = 0, 1, 2, 3
SWING, STANCE, FORCE_STANCE, FORCE_STOP = 0, 1, 2, 3
POSING, STOP_POSING, POSING_COMPLETE = 0, 1, 2
TIP_TOLERANCE = 0.01
TOUCHDOWN_THRESHOLD, LIFTOFF_THRESHOLD = 0.9, 0.1   # default.yaml touchdown_threshold / liftoff_threshold
UNASSIGNED = 2147483647.0


def round_to_int(x):
    return int(x + 0.5) if x >= 0 else -int(0.5 - x)


def round_to_even_int(x):
    return int(x) if int(x) % 2 == 0 else int(x) + 1


def sign(x):
    return 1.0 if x > 0 else -1.0


def smooth_step(c):
    return 6.0 * c ** 5 - 15.0 * c ** 4 + 10.0 * c ** 3


def projection(a, b):
    if a.dot(a) == 0.0 or b.dot(b) == 0.0:
        return np.zeros(3)
    return (a.dot(b) / b.dot(b)) * b


def bezier(p, t):
    s = 1.0 - t
    return p[0] * s ** 4 + p[1] * (4 * t * s ** 3) + p[2] * (6 * t * t * s * s) + p[3] * (4 * t ** 3 * s) + p[4] * t ** 4


def bezier_dot(p, t):
    s = 1.0 - t
    return 4 * s ** 3 * (p[1] - p[0]) + 12 * s * s * t * (p[2] - p[1]) + 12 * s * t * t * (p[3] - p[2]) + 4 * t ** 3 * (p[4] - p[3])


# ---- poses: position + scipy Rotation
class Pose:
    def __init__(self, p=None, r=None):
        self.p = np.zeros(3) if p is None else np.array(p, dtype=float)
        self.r = R.identity() if r is None else r

    def add(self, o):  # pose.h:167: position + rotation * other.position, rotation * other.rotation
        return Pose(self.p + self.r.apply(o.p), self.r * o.r)

    def interpolate(self, c, t):  # pose.h:190: linear position, slerp rotation
        key = R.concatenate([self.r, t.r])
        from scipy.spatial.transform import Slerp
        return Pose((1.0 - c) * self.p + c * t.p, Slerp([0.0, 1.0], key)([c])[0])

    def as7(self):
        q = self.r.as_quat()  # x, y, z, w
        if q[3] < 0:
            q = -q
        return [self.p[0], self.p[1], self.p[2], q[3], q[0], q[1], q[2]]


def remove_pose(a, b):  # pose.h:178: a.transformVector(-b.position), a.rotation * b.rotation^-1
    return Pose(a.p + a.r.apply(-b.p), a.r * b.r.inv())


def euler_to_rot(e):  # eulerAnglesToQuaternion, extrinsic: Rz(yaw) * Ry(pitch) * Rx(roll)
    return R.from_euler("xyz", [e[0], e[1], e[2]])


def rot_to_euler(r):  # quaternionToEulerAngles, extrinsic, (roll, pitch, yaw)
    return r.as_euler("xyz")


def from_two_vectors(a, b):
    a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
    ax = np.cross(a, b)
    s = np.linalg.norm(ax)
    if s == 0.0:
        return R.identity()
    return R.from_rotvec(ax / s * math.atan2(s, a.dot(b)))


# ---- kinematic model of the default.yaml hexapod (config/default.yaml: base / coxa / femur / tibia links, joint limits)
HEX_BASE_THETA = [-0.523, -1.571, -2.617, 2.617, 1.571, 0.523]
HEX_LINKS = [(0.0, 0.0, 0.050, 1.571), (0.0, 0.0, 0.050, 0.0), (0.0, -0.100, 0.100, 0.0)]  # d theta r alpha (coxa, femur, tibia)
HEX_JOINTS = [(-0.55, 0.55, 5.0), (-1.5, 1.5, 5.0), (-2.355, -0.1, 5.0)]                   # min, max, max speed
DLS_COEFFICIENT, JOINT_LIMIT_COST_WEIGHT = 0.02, 0.1                                       # model.h:19-20


def dh(d, th, r, al):
    c, s_, ca, sa = np.cos(th), np.sin(th), np.cos(al), np.sin(al)
    return np.array([[c, -s_ * ca, s_ * sa, r * c], [s_, c * ca, -c * sa, r * s_], [0, sa, ca, d], [0, 0, 0, 1]])


class Morphology:
    """Kinematic data of one robot: per leg the base link, the joint links (d, theta, r, alpha) and the joint limits."""

    def __init__(self, base, links, joints):
        self.base, self.links, self.joints = base, links, joints

    @staticmethod
    def default_hexapod():
        return Morphology([(0.0, th, 0.05, 0.0) for th in HEX_BASE_THETA], [HEX_LINKS] * 6, [HEX_JOINTS] * 6)

    @staticmethod
    def from_params(p):   # a parameter set of this package (the synthetic octopod): the numbers of its DH table, nothing else
        L = p.leg_count
        lk = lambda l, k: (p.link[l][k].d, p.link[l][k].theta, p.link[l][k].r, p.link[l][k].alpha)
        return Morphology([lk(l, 0) for l in range(L)], [[lk(l, k) for k in range(1, p.leg_dof[l] + 1)] for l in range(L)],
                          [[(p.joint[l][j].min, p.joint[l][j].max, p.joint[l][j].max_vel) for j in range(p.leg_dof[l])] for l in range(L)])


MODEL = Morphology.default_hexapod()


def _chain(leg, q):
    chain, t = [], np.eye(4)
    for k, (d, th, r, al) in enumerate(MODEL.links[leg]):
        t = t @ dh(d, th + q[k], r, al)
        chain.append(t)
    return chain


def solve_ik(leg, q, qd, delta, solve_rotation):
    """Leg::solveIK (src/model.cpp:726-797): DLS pseudo-inverse of the 6-row Jacobian (angular rows zero unless solve_rotation) applied
    to delta, plus the joint-limit cost gradient projected into its null space."""
    n = len(q)
    chain = _chain(leg, q)
    pe = chain[-1][:3, 3]
    z = [np.array([0, 0, 1.0])] + [c[:3, 2] for c in chain[:-1]]
    o = [np.zeros(3)] + [c[:3, 3] for c in chain[:-1]]
    jac = np.zeros((6, n))
    for i in range(n):
        jac[:3, i] = np.cross(z[i], pe - o[i])
        if solve_rotation:
            jac[3:, i] = z[i]
    jinv = jac.T @ np.linalg.inv(jac @ jac.T + DLS_COEFFICIENT ** 2 * np.eye(6))
    w = JOINT_LIMIT_COST_WEIGHT
    pg, vg, pc, vc = np.zeros(n), np.zeros(n), 0.0, 0.0     # joint-limit avoidance cost gradients (:762-790)
    for i, (mn, mx, mv) in enumerate(MODEL.joints[leg]):
        rg, cen = mx - mn, mn + (mx - mn) / 2
        pc += (w * (q[i] - cen) / rg) ** 2
        pg[i] = -w * w * (q[i] - cen) / rg ** 2
        vc += (w * qd[i] / (2 * mv)) ** 2
        vg[i] = -w * w * qd[i] / (2 * mv) ** 2
    pg *= 0 if pc == 0 else 1 / np.sqrt(pc)
    vg *= 0 if vc == 0 else 1 / np.sqrt(vc)
    g = 0.25 * pg + 0.75 * vg
    return jinv @ delta + (np.eye(n) - jinv @ jac) @ g      # the null-space term matters for the redundant 4- / 5-joint chains


def update_joints(leg, q, dq, dt, simulation):
    """Leg::updateJointPositions (:799-857), clamp_joint_velocities (not in simulation) / clamp_joint_positions on."""
    n = len(q)
    qn, vn = np.array(q, dtype=float), np.zeros(n)
    proximity = 1.0                                         # the return value: how close the closest joint is to a limit (0 = on it)
    for i, (mn, mx, mv) in enumerate(MODEL.joints[leg]):
        v = dq[i] / dt
        if not simulation:
            v = min(max(v, -mv), mv)
        vn[i] = v
        qn[i] = min(max(q[i] + v * dt, mn), mx)
        half = (mx - mn) / 2.0
        proximity = min(proximity, min(abs(mn - qn[i]), abs(mx - qn[i])) / half if half != 0 else 1.0)
    return qn, vn, proximity


def apply_ik(leg, q, qd, desired, dt, desired_dir=None, simulation=False, held=None):
    """Leg::applyIK (:861-941) towards a desired tip position (robot frame) and, optionally, tip direction.  Returns (q, qd).
    simulation: applyIK(true), the joint velocity clamp is off (the init chain's calls).
    held: (tip position, tip x axis) in the robot frame of Leg::current_tip_pose_ where that is NOT the FK of q - joint_control's
    updateManual moves the joints with applyFK(false), which refreshes the joint transforms (the Jacobian) but not the tip pose."""
    base = dh(*MODEL.base[leg])
    bi = np.linalg.inv(base)
    chain = _chain(leg, q)
    cur = (base @ chain[-1])[:3, 3]
    cur_dir_leg = chain[-1][:3, 0]                          # leg_frame_current_tip_pose: taken BEFORE any update (:865)
    if held is not None:
        cur, cur_dir_leg = held[0], bi[:3, :3] @ held[1]
    delta = np.zeros(6)
    delta[:3] = (bi @ np.append(desired, 1))[:3] - (bi @ np.append(cur, 1))[:3]        # tip delta in the leg base frame (:863-872)
    dq = solve_ik(leg, q, qd, delta, False)
    if desired_dir is not None:                             # rotation_constrained (:880-900)
        q, qd, _ = update_joints(leg, q, dq, dt, True)
        des_dir_leg = bi[:3, :3] @ desired_dir
        rv = from_two_vectors(cur_dir_leg, des_dir_leg).as_rotvec()                   # AngleAxisd(difference): axis * angle
        delta = np.zeros(6)
        delta[3:] = rv
        dq = solve_ik(leg, q, qd, delta, True)
    qn, vn, ik_success = update_joints(leg, q, dq, dt, simulation)
    tip = (base @ _chain(leg, qn)[-1])[:3, 3]
    if (np.abs(tip - desired) > 0.005).any():               # IK_TOLERANCE (:916-929)
        ik_success = 0.0
    if desired_dir is not None and not ik_success:          # a joint ON its limit (proximity 0) counts as failure too: retry unconstrained (:932-936)
        return apply_ik(leg, qn, vn, desired, dt, None, simulation)
    return qn, vn


def tip_force_estimate(leg, q, efforts, state, force_gain):
    """Leg::calculateTipForce (src/model.cpp:667-708): DLS pseudo-inverse of the 6-row Jacobian transposed, applied to the measured
    joint torques, rotated by the inverse base rotation, low-pass filtered (0.15) with the force gain.  state [3] is updated in place."""
    n = len(q)
    chain, t = [], np.eye(4)
    for k, (d, th, r, al) in enumerate(MODEL.links[leg]):
        t = t @ dh(d, th + q[k], r, al)
        chain.append(t)
    pe = chain[-1][:3, 3]
    z = [np.array([0, 0, 1.0])] + [c[:3, 2] for c in chain[:-1]]
    o = [np.zeros(3)] + [c[:3, 3] for c in chain[:-1]]
    jac = np.zeros((6, n))
    for i in range(n):
        jac[:3, i] = np.cross(z[i], pe - o[i])
        jac[3:, i] = z[i]
    transformation = jac @ np.linalg.inv(jac.T @ jac + DLS_COEFFICIENT ** 2 * np.eye(n))
    raw_leg = (transformation @ np.asarray(efforts, dtype=float))[:3]
    raw = np.linalg.inv(dh(*MODEL.base[leg]))[:3, :3] @ raw_leg      # first_joint->getPoseJointFrame().rotation_
    state[:] = 0.15 * raw * force_gain + (1 - 0.15) * state


def fk_tip(leg, q):
    t = dh(*MODEL.base[leg])
    for k, (d, th, r, al) in enumerate(MODEL.links[leg]):
        t = t @ dh(d, th + q[k], r, al)
    return t[:3, 3].copy()


def tip_axis(leg, q):
    """x axis of the tip frame in the robot frame (Leg::current_tip_pose_.rotation_ * UnitX after applyFK)."""
    t = dh(*MODEL.base[leg])
    for k, (d, th, r, al) in enumerate(MODEL.links[leg]):
        t = t @ dh(d, th + q[k], r, al)
    return t[:3, 0]


def admittance_delta(state, force, axis, P):
    """AdmittanceController::updateAdmittance for one leg: state [2] is advanced in place, returns admittance_delta_."""
    m, k, zeta, T = P["virtual_mass"], P["virtual_stiffness"], P["virtual_damping_ratio"], P["integrator_step_time"]
    c = zeta * 2 * math.sqrt(m * k)
    A = np.array([[0.0, 1.0], [-k / m, -c / m]])
    delta = np.zeros(3)
    for i in range(3):
        u = max(force[i] * P["force_gain"], 0.0)
        b = np.array([0.0, -u / m])
        h = T / 30
        x = state.copy()
        for _ in range(30):                                  # boost::numeric::odeint::runge_kutta4, integrate_const(0, T, T / 30)
            k1 = A @ x + b
            k2 = A @ (x + 0.5 * h * k1) + b
            k3 = A @ (x + 0.5 * h * k2) + b
            k4 = A @ (x + h * k3) + b
            x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        state[:] = x
        d = min(max(-x[0], -0.2), 0.2)
        if abs(d) > 0.0:                                     # ADMITTANCE_DEADBAND = 0
            delta[i] = d
    return projection(delta, axis)                           # Leg::setAdmittanceDelta (model.h:365-368)


class Leg:
    def __init__(self, stance_xy):
        self.identity = np.array([stance_xy[0], stance_xy[1], 0.0])
        self.default = self.identity.copy()
        self.tip = self.identity.copy()
        self.target = self.identity.copy()
        self.tip_velocity = np.zeros(3)
        self.swing_origin = self.identity.copy()
        self.swing_origin_velocity = np.zeros(3)
        self.stance_origin = self.identity.copy()
        self.stride = np.zeros(3)
        self.walk_plane = np.zeros(3)
        self.walk_plane_normal = np.array([0.0, 0.0, 1.0])
        self.swing_clearance = np.zeros(3)
        self.phase = 0
        self.phase_offset = 0
        self.state = STANCE
        self.at_correct_phase = False
        self.completed_first_step = False
        self.swing_progress = -1.0
        self.stance_progress = -1.0
        self.ext_target = None      # dict(pose=Pose, transform=Pose, clearance=float, odom_ideal=bool) while defined_
        self.ext_default = None
        self.touchdown_detection = False
        self.leg_state = 0          # enum LegState: 0 WALKING, 1 MANUAL, -1 WALKING_TO_MANUAL, -2 MANUAL_TO_WALKING
        self.desired_tip = None     # Leg::desired_tip_pose_.position_ of the last setDesiredTipPose
        self.step_plane = None      # Leg::step_plane_pose_.position_ while defined
        self.rot_defined = False    # current_tip_pose_.rotation_ != UNDEFINED_ROTATION (gravity-aligned tips, > 3 joints)
        self.cur_dir = self.origin_dir = self.model_dir = np.array([0.0, 0.0, -1.0])   # x axes of current / origin tip rotation, of the FK tip frame
        self.model_tip = None       # Leg::current_tip_pose_.position_ (FK of the joints), scenarios with the kinematic model
        self.negate_auto_pose, self.auto_pose = False, None   # LegPoser::negate_auto_pose_, auto_pose_
        self.target_dir, self.target_dir_set = None, False   # x axis of LegStepper::target_tip_pose_.rotation_ (None: undefined)
        self.poser_dir = None       # x axis of LegPoser::current_tip_pose_.rotation_ (None: undefined)
        self.held = None            # Leg::current_tip_pose_ (position, x axis) where joint_control's updateManual has moved the joints under it


class RefWalker:
    """One robot: WalkController + LegSteppers + the body-pose parts of PoseController that depend on them."""

    def __init__(self, P, limits):
        self.P = P
        self.dt = P["time_delta"]
        self.legs = [Leg(xy) for xy in P["stance_position"]]
        self.L = len(self.legs)
        self.limits = limits  # dict of four 9-entry tables (bearing 0, 45, .. 360)
        self.workspaces = None  # per leg {height: {bearing: radius}} (Leg::generateWorkspace), scenarios with a stance span modifier
        self.walk_state = STOPPED
        self.pose_state = POSING_COMPLETE
        self.v = np.zeros(2)
        self.w = 0.0
        self.lacp = self.lcfs = 0
        self.rtda = False
        self.walk_plane = np.zeros(3)
        self.walk_plane_normal = np.array([0.0, 0.0, 1.0])
        self.step_cycle()
        # PoseController
        self.walk_plane_pose = Pose([0, 0, P["body_clearance"]])
        self.origin_walk_plane_pose = Pose([0, 0, P["body_clearance"]])
        self.auto_posing_state = POSING_COMPLETE
        self.posers = [dict(start_check=False, e1=False, e2=False, allow=False) for _ in range(P.get("n_auto_posers", 0))]
        self.abs_err = np.zeros(3)
        self.vel_err = np.zeros(3)
        self.imu_q = R.identity()
        self.gyro = np.zeros(3)
        self.current_pose = Pose([0, 0, P["body_clearance"]])
        self.q = self.qd = None   # joint state [legs][3], for the scenarios that run the kinematic model
        self.manual_pose = Pose()
        self.primary_leg = self.secondary_leg = -1
        self.primary_velocity = self.secondary_velocity = self.primary_position = self.secondary_position = np.zeros(3)
        self.tip_align_pose, self.origin_tip_align_pose = Pose(), Pose()
        self.inclination = Pose()
        self.auto_pose_now = Pose()
        self.odometry = Pose()      # WalkController::odometry_ideal_
        self.tvi, self.rvi = np.zeros(3), np.zeros(3)   # translation / rotation_velocity_input_ (rewritten by the reset modes)
        self.reset_mode = 0
        self.prev_auto_r = R.identity()
        self.adm_state = np.zeros((self.L, 2))
        self.stiffness = [0.0] * self.L   # Leg::virtual_stiffness_: uninitialised in the reference until the first updateStiffness (0 by this build's convention)
        self.tip_force = np.zeros((self.L, 3))
        self.tip_force_calc = np.zeros((self.L, 3))   # Leg::tip_force_calculated_
        self.adjust = None          # StateController::dynamic_parameter_ / new_parameter_value_ while parameter_adjust_flag_ is set: (name, value)
        self.adjust_log = []        # cycles (counted by the caller) in which a pending adjustment was set
        self.walkspace = self.default_tips = None   # what generateLimits needs for a new step cycle (scenarios that adjust step_frequency)
        self.efforts = None                            # Joint::current_effort_ [legs][dof]

    def step_cycle(self):  # generateStepCycle + the phase offsets of generateLimits
        P = self.P
        base = P["stance_phase"] + P["swing_phase"]
        swing_ratio = P["swing_phase"] / base
        raw = ((1.0 / P["step_frequency"]) / self.dt) / swing_ratio
        self.period = round_to_even_int(raw / base) * base
        self.frequency = 1.0 / (self.period * self.dt)
        n = self.period // base
        self.stance_end = int(P["stance_phase"] * 0.5) * n
        self.swing_start = self.stance_end
        self.swing_end = (int(P["stance_phase"] * 0.5) + P["swing_phase"]) * n
        self.stance_start = self.swing_end
        self.stance_period = (self.stance_end - self.stance_start) % self.period
        self.swing_period = self.swing_end - self.swing_start
        for leg, m in zip(self.legs, P["offset_multiplier"]):
            leg.phase_offset = (int(P["phase_offset"] * n) * m) % self.period
        # auto pose phases follow the step cycle (pose_frequency -1)
        self.pose_length = self.period
        self.pose_norm = n  # setAutoPoseParams with pose_frequency -1: base length = stance + swing phase (pose_controller.cpp:44-63)
        self.ref_leg = 0
        for i, m in enumerate(P["offset_multiplier"]):
            if m == 0:
                self.ref_leg = i

    # ---- WalkController::getLimit
    def limit(self, lin, ang, table):
        lo = UNASSIGNED
        for leg in self.legs:
            sv = lin + ang * np.array([-leg.tip[1], leg.tip[0]])
            bearing = round_to_int(math.degrees(math.atan2(sv[1], sv[0]))) % 360
            upper = ((bearing + 44) // 45) * 45          # LimitMap::lower_bound(bearing)
            lower = (upper - 45) % 360
            if bearing < lower:
                bearing += 360
            if upper < lower:
                upper += 360
            c = float((bearing - lower) // (upper - lower))  # int / int
            value = table[lower // 45] * (1.0 - c) + table[(upper % 360) // 45] * c
            lo = min(lo, value)
        return lo

    # ---- LegStepper helpers
    def update_step_state(self, leg):
        if leg.state == FORCE_STOP:
            return
        if self.swing_start <= leg.phase < self.swing_end and leg.state != FORCE_STANCE:
            leg.state = SWING
        elif leg.phase < self.stance_end or leg.phase >= self.stance_start:
            leg.state = STANCE

    def iterate_phase(self, leg):
        leg.phase = (leg.phase + 1) % self.period
        self.update_step_state(leg)
        if leg.state == SWING:
            leg.swing_progress = min(1.0, max(0.0, (leg.phase - self.swing_start + 1) / (self.swing_end - self.swing_start)))
            leg.stance_progress = -1.0
        elif leg.state == STANCE:
            num = (leg.phase + (self.period - self.stance_start)) % self.period + 1
            leg.stance_progress = min(1.0, max(0.0, num / ((self.stance_end - self.stance_start) % self.period)))
            leg.swing_progress = -1.0
        elif leg.state == FORCE_STOP:
            leg.stance_progress, leg.swing_progress = 0.0, -1.0

    def update_stride(self, leg):
        leg.walk_plane, leg.walk_plane_normal = self.walk_plane.copy(), self.walk_plane_normal.copy()
        radius = np.array([leg.tip[0], leg.tip[1], 0.0])
        stride = np.array([self.v[0], self.v[1], 0.0]) + np.cross(np.array([0, 0, self.w]), radius)
        leg.stride = stride * ((self.stance_period / self.period) / self.frequency)
        leg.swing_clearance = self.P["swing_height"] * leg.walk_plane_normal / np.linalg.norm(leg.walk_plane_normal)

    def stance_span_change(self, leg):
        """LegStepper::calculateStanceSpanChange (:949-980): the identity tip is shifted sideways by stance_span_modifier times the workspace
        radius towards / away from the body, at the height the default tip has drifted to (interpolated between the bounding workplanes of
        a layered workspace; the single plane of a simple one)."""
        m = self.P.get("stance_span_modifier", 0.0)
        ws = self.workspaces[self.legs.index(leg)]           # {height: {bearing: radius}}
        target = round_to_int((leg.default - leg.identity)[2] * 1000.0) / 1000.0       # setPrecision(.., 3)
        heights = sorted(ws)
        positive_y = leg.identity[1] > 0.0
        bearing = 270 if (positive_y ^ (m > 0.0)) else 90
        m = m * (1.0 if positive_y else -1.0)
        if len(heights) == 1:
            radius = ws[0.0][bearing]
        else:
            ub = next(k for k, h in enumerate(heights) if h > target)                   # map::upper_bound, then prev()
            hi, lo = heights[ub], heights[ub - 1]
            hi3, lo3 = round_to_int(hi * 1000.0) / 1000.0, round_to_int(lo * 1000.0) / 1000.0
            i = (target - lo3) / (hi3 - lo3)
            radius = ws[lo][bearing] * (1.0 - i) + ws[hi][bearing] * i
        return np.array([0.0, radius * m, 0.0])

    def update_default_tip(self, leg):
        if leg.ext_default is not None:          # external_default_.pose_.removePose(external_default_.transform_) (:988-990)
            leg.default = remove_pose(leg.ext_default["pose"], leg.ext_default["transform"]).p
            return
        identity = leg.identity
        if self.P.get("stance_span_modifier", 0.0) != 0.0 or self.workspaces is not None:
            identity = identity + self.stance_span_change(leg)
        ident = self.walk_plane_pose.p + self.walk_plane_pose.r.apply(identity)   # getDefaultBodyPose().transformVector
        leg.default = ident + projection(leg.stance_origin - ident, leg.walk_plane_normal)

    def update_tip_position(self, leg):
        P, dt = self.P, self.dt
        standard = leg.state == SWING or leg.completed_first_step
        mss = self.stance_start if standard else leg.phase_offset
        msp = (self.stance_end - mss) % self.period
        if self.stance_end == mss:
            msp = self.period
        swing_iterations = round_to_even_int(int((self.swing_period / self.period) / (self.frequency * dt)))
        swing_dt = 1.0 / (swing_iterations / 2.0)
        stance_iterations = int((msp / self.period) / (self.frequency * dt))
        stance_dt = 1.0 / stance_iterations
        leg.target = leg.default + 0.5 * leg.stride
        if leg.state == SWING:
            self.update_stride(leg)
            it = leg.phase - self.swing_start + 1
            first_half = it <= swing_iterations // 2
            if it == 1:
                leg.swing_origin, leg.swing_origin_velocity = leg.tip.copy(), leg.tip_velocity.copy()
                if P.get("rough_terrain_mode"):
                    self.update_default_tip(leg)
            ground_contact = False
            if P.get("rough_terrain_mode"):
                if leg.ext_target is not None:   # :1068-1079: target_tip_pose_ = pose_.removePose(transform_) - position AND rotation
                    tp = remove_pose(leg.ext_target["pose"], leg.ext_target["transform"])
                    leg.target = tp.p
                    # (a requested pose without a rotation - the zero quaternion - stays undefined through the product)
                    leg.target_dir = tp.r.apply(np.array([1.0, 0, 0])) if leg.ext_target.get("rot_defined", True) else None
                    leg.target_dir_set = True
                    leg.swing_clearance = leg.swing_clearance / np.linalg.norm(leg.swing_clearance) * leg.ext_target["clearance"]
                    if leg.ext_target["odom_ideal"]:
                        leg.target = leg.target - np.array([self.v[0], self.v[1], 0.0]) * ((swing_iterations - it) * dt)
                elif leg.touchdown_detection:
                    if leg.step_plane is not None:   # proactive: shift the target onto the sensed step plane (:1082-1096)
                        target_tip = leg.tip + (leg.step_plane - leg.model_tip)
                        leg.target = leg.target + projection(target_tip - leg.target, leg.walk_plane_normal)
                    else:                            # reactive (:1099-1102)
                        leg.target = leg.target - np.array([0.0, 0.0, P["step_depth"]])
                ground_contact = leg.step_plane is not None
            mid = (leg.swing_origin + leg.target) / 2.0
            mid[2] = max(leg.swing_origin[2], leg.target[2])
            mid = mid + leg.swing_clearance
            mid[1] += P["swing_width"] if leg.identity[1] > 0.0 else -P["swing_width"]
            sep = 0.25 * leg.swing_origin_velocity * (dt / swing_dt)
            n1 = [leg.swing_origin, leg.swing_origin + sep, leg.swing_origin + 2.0 * sep, None, mid]
            n1[3] = (mid + n1[2]) / 2.0
            n1[3][2] = mid[2]
            final_velocity = -leg.stride * (stance_dt / dt)
            sep2 = 0.25 * final_velocity * (dt / swing_dt)
            n2 = [n1[4], n1[4] - (n1[3] - n1[4]), leg.target - 2.0 * sep2, leg.target - sep2, leg.target]
            if not first_half and ground_contact:     # "Stops further movement of tip position in direction normal to walk plane"
                n2 = [leg.tip + k * sep2 for k in range(5)]
            if P["force_normal_touchdown"] and not ground_contact:
                origin = leg.target - 4.0 * sep2
                origin[2] = max(leg.swing_origin[2], leg.target[2])
                origin = origin + leg.swing_clearance
                n1[4] = origin
                n2[0] = origin
                n2[2] = leg.target - 2.0 * sep2
                n1[3] = n2[0] - (n2[2] - origin) / 2.0
                n2[1] = n2[0] + (n2[2] - origin) / 2.0
            if first_half:
                delta = swing_dt * bezier_dot(n1, swing_dt * it)
            else:
                delta = swing_dt * bezier_dot(n2, swing_dt * (it - swing_iterations // 2))
            leg.tip = leg.tip + delta
            leg.tip_velocity = delta / dt
        elif leg.state in (STANCE, FORCE_STANCE):
            self.update_stride(leg)
            it = (leg.phase + (self.period - mss)) % self.period + 1
            if it == 1:
                leg.stance_origin = leg.tip.copy()
                leg.ext_target = None            # "Reset external target after every swing period" (:1159)
                if P.get("rough_terrain_mode"):
                    self.update_default_tip(leg)
            scaler = msp / ((self.stance_end - self.stance_start) % self.period)
            sep = -leg.stride * scaler * 0.25
            nodes = [leg.stance_origin + k * sep for k in range(5)]
            delta = stance_dt * bezier_dot(nodes, it * stance_dt)
            leg.tip = leg.tip + delta
            leg.tip_velocity = delta / dt

    def update_tip_rotation(self, leg):   # LegStepper::updateTipRotation (:1193-1234), legs of more than 3 joints
        if not leg.target_dir_set:        # target_tip_pose_ starts as the identity tip pose: gravity-aligned with the parameter, else undefined (:37-41)
            leg.target_dir = np.array([0.0, 0.0, -1.0]) if self.P.get("gravity_aligned_tips") else None   # FromTwoVectors(UnitX, -UnitZ) * UnitX
            leg.target_dir_set = True
        if leg.stance_progress >= 0.0 or leg.swing_progress >= 0.5:
            if self.P.get("gravity_aligned_tips") and leg.target_dir is None:   # an undefined target is re-assigned from the gravity estimate (:1197-1205)
                g = self.estimate_gravity()
                leg.target_dir = g / np.linalg.norm(g)
            if leg.target_dir is None:                        # "Target is undefined so set current tip rotation to undefined" (:1208-1211)
                leg.rot_defined = False
                return
            target_dir = leg.target_dir
            leg.cur_dir = target_dir
            if leg.swing_progress >= 0.5:
                c = smooth_step(min(1.0, 2.0 * (leg.swing_progress - 0.5)))
                nd = (1.0 - c) * leg.origin_dir + c * target_dir
                leg.cur_dir = nd / np.linalg.norm(nd)
            leg.rot_defined = True
        else:
            leg.origin_dir = leg.model_dir.copy()             # leg_->getCurrentTipPose().rotation_: the FK tip frame
            leg.rot_defined = False

    def update_walk_plane(self):
        A = np.array([[leg.default[0], leg.default[1], 1.0] for leg in self.legs])
        B = np.array([leg.default[2] for leg in self.legs])
        self.walk_plane = np.linalg.solve(A.T @ A, A.T @ B)
        n = np.array([-self.walk_plane[0], -self.walk_plane[1], 1.0])
        self.walk_plane_normal = n / np.linalg.norm(n)

    # ---- WalkController::updateWalk
    def update_walk(self, lin, ang):
        P, dt, L = self.P, self.dt, self.L
        lin = np.array(lin, dtype=float)
        mls, mas = self.limit(lin, ang, self.limits["max_linear_speed"]), self.limit(lin, ang, self.limits["max_angular_speed"])
        mla, maa = self.limit(lin, ang, self.limits["max_linear_acceleration"]), self.limit(lin, ang, self.limits["max_angular_acceleration"])
        norm = np.linalg.norm(lin)
        if self.walk_state != STOPPING:
            if P["velocity_input_mode"] == "throttle":
                nv = (lin / norm if norm > 1.0 else lin) * mls
                nw = min(1.0, max(-1.0, ang)) * mas
                nv = nv * (1.0 - abs(ang))
            else:
                nv = lin * (mls / norm) if norm > mls else lin.copy()
                nw = min(mas, max(-mas, ang))
                nv = nv * ((1.0 - abs(nw / mas)) if mas != 0.0 else 0.0)
        else:
            nv, nw = np.zeros(2), 0.0
        has_command = bool(norm) or bool(ang)
        if any(leg.leg_state != 0 for leg in self.legs):      # "Check that all legs are in WALKING state" (:491-505): nothing below runs
            return
        acc = nv - self.v
        if np.linalg.norm(acc) < mla * dt:
            self.v = self.v + acc
        else:
            self.v = self.v + acc / np.linalg.norm(acc) * mla * dt
        aacc = nw - self.w
        if abs(aacc) < maa * dt:
            self.w += aacc
        else:
            self.w += sign(aacc) * maa * dt
        if self.walk_state == STOPPED and has_command:
            self.walk_state = STARTING
            for leg in self.legs:
                leg.at_correct_phase = leg.completed_first_step = False
                leg.state = STANCE
                leg.phase = leg.phase_offset
                self.update_step_state(leg)
            return
        elif self.walk_state == STARTING and self.lacp == L and self.lcfs == L:
            self.lacp = self.lcfs = 0
            self.walk_state = MOVING
        elif self.walk_state == MOVING and not has_command:
            self.walk_state = STOPPING
        elif self.walk_state == STOPPING and self.lacp == L and self.pose_state == POSING_COMPLETE:
            self.lacp = 0
            self.walk_state = STOPPED
        for leg in self.legs:
            if self.walk_state == STARTING:
                if self.lacp == L and leg.phase == self.swing_end and not leg.completed_first_step:
                    leg.completed_first_step = True
                    self.lcfs += 1
                if not leg.at_correct_phase:
                    if self.swing_start < leg.phase_offset < self.swing_end and leg.phase != self.swing_end:
                        leg.state = FORCE_STANCE
                    else:
                        self.lacp += 1
                        leg.at_correct_phase = True
            elif self.walk_state == MOVING:
                leg.at_correct_phase = False
            elif self.walk_state == STOPPING:
                zero_velocity = np.linalg.norm(leg.stride) == 0
                err = leg.tip - leg.target
                err = err - projection(err, leg.walk_plane_normal)
                at_target = np.linalg.norm(err) < TIP_TOLERANCE
                if zero_velocity and not leg.at_correct_phase and leg.phase == self.swing_end:
                    if at_target or self.rtda:
                        self.rtda = False
                        self.update_default_tip(leg)
                        leg.state = FORCE_STOP
                        leg.at_correct_phase = True
                        self.lacp += 1
                    else:
                        self.rtda = True
            elif self.walk_state == STOPPED:
                leg.state = FORCE_STOP
                leg.phase = 0
            self.update_tip_position(leg)
            # tip rotations matter on legs of more than 3 joints, with gravity-aligned tips or where a requested target may carry one (rough terrain mode)
            if (P.get("gravity_aligned_tips") or P.get("rough_terrain_mode")) and self.q is not None and self.q.shape[1] > 3:
                self.update_tip_rotation(leg)
            self.iterate_phase(leg)
        self.update_walk_plane()
        # odometry_ideal_ = odometry_ideal_.addPose(calculateOdometry(time_delta_)) (:643, :783-791): the desired body velocity integrated
        self.odometry = self.odometry.add(Pose([self.v[0] * dt, self.v[1] * dt, 0.0], R.from_rotvec([0.0, 0.0, self.w * dt])))

    # ---- PoseController
    def update_walk_plane_pose(self):
        P = self.P
        plane, normal, c = np.zeros(3), np.array([0.0, 0.0, 1.0]), 0.0
        scaler = max(1.0, P["swing_phase"] / P["phase_offset"])
        for leg in self.legs:
            sp = leg.swing_progress * scaler
            if 0 <= sp <= 1.0:
                c, plane, normal = smooth_step(sp), leg.walk_plane, leg.walk_plane_normal
        new = Pose()
        new.r = from_two_vectors(np.array([0.0, 0.0, 1.0]), normal)
        new.p = new.r.apply(np.array([0.0, 0.0, P["body_clearance"]]))
        new.p[2] += plane[2]
        self.walk_plane_pose = self.origin_walk_plane_pose.interpolate(c, new)
        if c == 1.0:
            self.origin_walk_plane_pose = self.walk_plane_pose

    def auto_pose(self):
        P = self.P
        ref = self.legs[self.ref_leg]
        if self.walk_state in (STARTING, MOVING):
            self.auto_posing_state = POSING
        elif (np.linalg.norm(ref.stride) == 0 and self.walk_state == STOPPING) or self.walk_state == STOPPED:
            self.auto_posing_state = STOP_POSING
        master = ref.phase
        pose, complete = Pose(), 0
        for i, st in enumerate(self.posers):
            phase, sp, ep = master, P["pose_phase_starts"][i] * self.pose_norm, P["pose_phase_ends"][i] * self.pose_norm
            if sp > ep:
                ep += self.pose_length
                if phase < sp:
                    phase += self.pose_length
            state = self.auto_posing_state
            st["start_check"] = (not st["start_check"]) and state == POSING and phase == sp
            st["e1"] = st["e1"] or (state == STOP_POSING and phase == sp)
            st["e2"] = st["e2"] or (state == STOP_POSING and phase == ep and st["e1"])
            if not st["allow"] and st["start_check"]:
                st["allow"], st["e1"], st["e2"] = True, False, False
            elif st["allow"] and st["e1"] and st["e2"]:
                st["allow"], st["start_check"] = False, False
            complete += 0 if st["allow"] else 1
            if sp <= phase < ep and st["allow"]:
                it, num = phase - sp + 1, ep - sp
                first = it <= num // 2
                amp_r = np.array([P["roll_amplitudes"][i], P["pitch_amplitudes"][i], P["yaw_amplitudes"][i]])
                amp_p = np.array([P["x_amplitudes"][i], P["y_amplitudes"][i], P["z_amplitudes"][i]])
                z = np.zeros(3)
                nodes_r = [z, z, z, amp_r, amp_r] if first else [amp_r, amp_r, z, z, z]
                nodes_p = [z, z, z, amp_p, amp_p] if first else [amp_p, amp_p, z, z, z]
                t = (it - (0 if first else int(num / 2.0))) * (1.0 / (num / 2.0))
                pose = pose.add(Pose(bezier(nodes_p, t), euler_to_rot(bezier(nodes_r, t))))
        if complete == len(self.posers):
            self.auto_posing_state = POSING_COMPLETE
        self.auto_pose_now = pose                            # PoseController::auto_pose_
        for i, leg in enumerate(self.legs):                   # LegPoser::updateAutoPose(master_phase) (:1716-1778): the leg's own auto pose - the
            sp, ep = P["pose_negation_phase_starts"][i] * self.pose_norm, P["pose_negation_phase_ends"][i] * self.pose_norm   # body's, negated over its window
            sp = sp if sp != 0 else self.pose_length
            ep = ep if ep != 0 else self.pose_length
            phase = master
            if sp > ep:
                ep += self.pose_length
                if phase < sp:
                    phase += self.pose_length
            if leg.state not in (FORCE_STANCE, FORCE_STOP) and phase == sp:
                leg.negate_auto_pose = True
            if phase < sp or phase > ep:
                leg.negate_auto_pose = False
            leg.auto_pose = pose
            if leg.negate_auto_pose:
                it, num = phase - sp + 1, ep - sp
                ratio = P["negation_transition_ratio"][i]
                c = 1.0
                if ratio > 0.0:
                    c = min(1.0, it / (num * ratio)) if it <= num // 2 else min(1.0, (num - it) / (num * ratio))
                leg.auto_pose = remove_pose(pose, Pose().interpolate(smooth_step(c), pose))
        return pose

    def update_manual(self):   # WalkController::updateManual: the velocity overload (:652-708) then the pose overload (:712-744)
        P = self.P
        joint_control = P.get("leg_manipulation_mode") == "joint_control"
        for i, leg in enumerate(self.legs):
            if leg.leg_state != 1:
                continue
            vel, pos = np.zeros(3), np.zeros(3)                # (a MANUAL leg that is neither selection reads an uninitialised vector in the reference)
            if i == self.primary_leg:
                vel, pos = self.primary_velocity, self.primary_position
            elif i == self.secondary_leg:
                vel, pos = self.secondary_velocity, self.secondary_position
            if joint_control:
                # :677-690 ("HACK", 3-joint legs only): y / x inputs step the coxa / tibia joints; applyFK(false) moves the joint transforms and
                # returns the tip pose they give - the stepper takes it WITH its rotation - while Leg::current_tip_pose_ stays what it was.
                # The pose overload needs tip_control (:733).
                if np.linalg.norm(vel) != 0.0 and len(self.q[i]) == 3:
                    leg.held = (fk_tip(i, self.q[i]), tip_axis(i, self.q[i]))
                    self.q[i] = self.q[i].copy()
                    self.q[i][0] += vel[1] * P["max_rotation_velocity"] * self.dt
                    self.q[i][2] += vel[0] * P["max_rotation_velocity"] * self.dt
                    leg.tip, leg.cur_dir, leg.rot_defined = fk_tip(i, self.q[i]), tip_axis(i, self.q[i]), True
                continue
            if np.linalg.norm(vel) != 0.0:
                ik_error = leg.desired_tip - leg.model_tip
                change = vel * P["max_translation_velocity"] * self.dt
                if np.linalg.norm(ik_error) >= 0.005:
                    change = np.linalg.norm(change) * -(ik_error / np.linalg.norm(ik_error))
                leg.tip, leg.rot_defined = leg.tip + change, False
            if np.linalg.norm(pos) != 0.0:
                leg.tip, leg.rot_defined = np.array(pos, dtype=float), False

    def update_tip_align_pose(self):   # PoseController::updateTipAlignPose (:1024-1088): legs in id order, each on the pose the previous one left
        P = self.P
        for i, leg in enumerate(self.legs):
            sp = leg.swing_progress
            if sp == -1.0:
                continue
            normal = leg.walk_plane_normal
            rot = from_two_vectors(np.array([0.0, 0.0, 1.0]), normal)
            base = dh(*MODEL.base[i])
            chain = _chain(i, self.q[i])
            tip_p, joint_p = (base @ chain[-1])[:3, 3], (base @ chain[-2])[:3, 3]     # the tip and the joint that actuates its link
            a = rot.apply(joint_p - tip_p)
            b = np.linalg.norm(tip_p - joint_p) * normal
            to_alignment = -(a - (a @ b) / (b @ b) * b)
            a = self.tip_align_pose.p
            aligned = a - (a @ normal) / (normal @ normal) * normal
            target = aligned + to_alignment
            lim = P["max_translation"]
            target = np.array([max(-lim[k], min(target[k], lim[1])) for k in range(3)])   # clamped(value, limit): the upper bound is limit[1] for every axis (standard_includes.h:134)
            c = smooth_step(sp)
            if sp < 0.5:
                self.tip_align_pose = self.origin_tip_align_pose.interpolate(smooth_step(c * 2.0), Pose())
            else:
                self.tip_align_pose = Pose().interpolate(smooth_step((c - 0.5) * 2.0), Pose(target))
            if sp == 1.0:
                self.origin_tip_align_pose = self.tip_align_pose

    def update_stiffness(self):     # AdmittanceController::updateStiffness(walker) (src/admittance_controller.cpp:96-134): published per-leg value
        P, L = self.P, self.L
        k = P["virtual_stiffness"]
        self.stiffness = [k] * L
        for i, leg in enumerate(self.legs):
            if leg.state == SWING:
                ref = abs((leg.tip[2] - leg.default[2]) / P["swing_height"])
                a1, a2 = (i - 1) % L, (i + 1) % L
                load = k * (ref * (P["load_stiffness_scaler"] - 1))
                c1, c2 = self.stiffness[a1], self.stiffness[a2]
                self.stiffness[i] = k * (ref * (P["swing_stiffness_scaler"] - 1) + 1)
                self.stiffness[a1] = c1 + load
                self.stiffness[a2] = c2 + load

    def update_manual_pose(self):   # PoseController::updateManualPose (:863-1003); default_pose_ is the identity (calculateDefaultPose is never called)
        P, dt = self.P, self.dt
        if self.reset_mode == 5:    # IMMEDIATE_ALL_RESET
            self.manual_pose = Pose()
            return
        reset_t = {1: (0, 0, 1), 2: (1, 1, 0), 3: (0, 0, 0), 4: (1, 1, 1)}.get(self.reset_mode, (0, 0, 0))   # Z_AND_YAW, X_AND_Y, PITCH_AND_ROLL, ALL
        reset_r = {1: (0, 0, 1), 2: (0, 0, 0), 3: (1, 1, 0), 4: (1, 1, 1)}.get(self.reset_mode, (0, 0, 0))
        cur_p = self.manual_pose.p.copy()
        cur_r = self.manual_pose.r.as_euler("XYZ")             # quaternionToEulerAngles(.., intrinsic)
        new_p, new_r = np.zeros(3), np.zeros(3)
        for i in range(3):
            for cur, reset, vin in ((cur_p, reset_t, self.tvi), (cur_r, reset_r, self.rvi)):
                if reset[i]:
                    if cur[i] < 0:
                        vin[i] = 1.0
                    elif cur[i] > 0:
                        vin[i] = -1.0
            for cur, reset, vin, vmax, pmax, out in ((cur_p, reset_t, self.tvi, P["max_translation_velocity"], P["max_translation"], new_p),
                                                     (cur_r, reset_r, self.rvi, P["max_rotation_velocity"], P["max_rotation"], new_r)):
                vel = vin[i] * vmax
                desired = cur[i] + vel * dt
                limit = sign(vel) * pmax[i]
                if reset[i] and -pmax[i] < 0.0 < pmax[i]:      # the default pose is the identity
                    limit = 0.0
                positive = sign(vel) > 0
                if (positive and desired > limit) or (not positive and desired < limit):
                    vel = (limit - cur[i]) / dt
                out[i] = cur[i] + vel * dt
        self.manual_pose = Pose(new_p, R.from_euler("XYZ", new_r))   # eulerAnglesToQuaternion(.., intrinsic)

    def estimate_gravity(self):     # Model::estimateGravity (src/model.cpp:156-165)
        e = rot_to_euler(self.imu_q)
        g = np.array([0.0, 0.0, -9.81])
        g = R.from_rotvec([0.0, -e[1], 0.0]).apply(g)
        return R.from_rotvec([-e[0], 0.0, 0.0]).apply(g)

    def inclination_pose(self):     # PoseController::updateInclinationPose (:1240-1259), reads the auto_pose_ of the previous cycle
        P = self.P
        combined = self.manual_pose.r * self.prev_auto_r
        e = rot_to_euler(self.imu_q * combined.inv())
        lon = min(P["max_translation"][0], max(-P["max_translation"][0], -P["body_clearance"] * math.tan(e[1])))
        lat = min(P["max_translation"][1], max(-P["max_translation"][1], P["body_clearance"] * math.tan(e[0])))
        return Pose([lon, lat, 0.0])

    def imu_pose(self):
        P = self.P
        err = rot_to_euler(self.imu_q * self.manual_pose.r.inv())  # target rotation = the manual pose's
        err[2] = 0.0
        self.abs_err = self.abs_err + err * self.dt
        self.vel_err = 0.15 * -self.gyro + (1 - 0.15) * self.vel_err
        kp, ki, kd = P["rotation_pid_gains"]
        corr = -(kd * self.vel_err + kp * err + ki * self.abs_err)
        corr[0] = min(P["max_rotation"][0], max(-P["max_rotation"][0], corr[0]))
        corr[1] = min(P["max_rotation"][1], max(-P["max_rotation"][1], corr[1]))
        corr[2] = rot_to_euler(self.manual_pose.r)[2]          # yaw of the target rotation (:1231)
        return Pose(None, euler_to_rot(corr))

    def prologue(self):
        """The posing / admittance part of StateController::loop (state_controller.cpp:165-181): returns (Model::current_pose_, admittance deltas)."""
        self.update_walk_plane_pose()
        pose = Pose().add(self.walk_plane_pose)
        if self.P.get("manual_posing"):
            self.update_manual_pose()
            pose = pose.add(self.manual_pose)
        if self.P.get("inclination_posing"):
            self.inclination = self.inclination_pose()       # PoseController::inclination_pose_ (poseForLegManipulation reads it, :572)
            pose = pose.add(self.inclination)
        if self.P.get("imu_posing"):
            pose = pose.add(self.imu_pose())
        elif self.P.get("auto_posing"):
            ap = self.auto_pose()
            pose = pose.add(ap)
            self.prev_auto_r = ap.r
        if self.P.get("gravity_aligned_tips") and self.q is not None and self.q.shape[1] <= 3:   # "TODO EXPERIMENTAL" (:849-855)
            self.update_tip_align_pose()
            pose = pose.add(self.tip_align_pose)
        self.current_pose = pose
        self.pose_state = self.auto_posing_state
        adm = [np.zeros(3)] * self.L
        if self.P.get("admittance_control") and self.P.get("dynamic_stiffness") and self.walk_state != STOPPED:
            self.update_stiffness()                                        # state_controller.cpp:175: with the walk state BEFORE updateWalk
        if self.P.get("admittance_control") and self.q is not None:      # loop(): the admittance update precedes runningState
            src = self.tip_force_calc if self.P.get("use_joint_effort") else self.tip_force      # getTipForceCalculated / Measured (:30-31)
            adm = [admittance_delta(self.adm_state[i], src[i], tip_axis(i, self.q[i]), self.P) for i in range(self.L)]
        return pose, adm

    def adjust_parameter(self, lin_in, ang_in):
        """StateController::adjustParameter (state_controller.cpp:451-509).  Every parameter is stored at once; step_frequency additionally installs the
        two speed maps and the phase offsets of the step cycle it WOULD give (:458-463; generateLimits sets every LegStepper's phase offset,
        walk_controller.cpp:277) and becomes the walker's step cycle only once the desired body velocity is inside the targets the velocity input maps to
        under the new limits (:464-497); a MOVING robot's legs are then mapped onto the new period (LegStepper::updatePhase, :402-409, :862-867).
        setAutoPoseParams is not called: the auto-pose phase tables keep the period they were generated with."""
        name, value = self.adjust
        P = self.P
        P[name] = value                                       # p->current_value = new_parameter_value_
        if name != "step_frequency":
            self.adjust = None
            return
        probe = RefWalker(P, {})                              # generateStepCycle(false): the cycle the new frequency gives, and generateLimits on it
        init = init_chain_module()
        lim = init.generate_limits(P, probe, self.walkspace, self.default_tips)
        self.limits = dict(self.limits, max_linear_speed=lim["max_linear_speed"], max_angular_speed=lim["max_angular_speed"])   # set*SpeedLimitMap (:462-463)
        for leg, pl in zip(self.legs, probe.legs):
            leg.phase_offset = pl.phase_offset                # generateLimits' setPhaseOffset
        max_lin = self.limit(lin_in, ang_in, lim["max_linear_speed"])
        max_ang = self.limit(lin_in, ang_in, lim["max_angular_speed"])
        norm = np.linalg.norm(lin_in)
        if P["velocity_input_mode"] == "throttle":
            target = (lin_in / norm if norm > 1.0 else lin_in) * max_lin
            target_w = min(1.0, max(-1.0, ang_in)) * max_ang
            target = target * (1.0 - abs(ang_in))
        else:
            target = lin_in * (max_lin / norm) if norm > max_lin else lin_in.copy()
            target_w = min(max_ang, max(-max_ang, ang_in))
        if not (self.v[0] <= target[0] and self.v[1] <= target[1] and abs(self.w) <= abs(target_w)):
            return                                            # "Slowing to safe speed before setting new parameter": asked again in the next loop
        old_period = self.period                              # generateStepCycle(): the walker's step cycle from now on
        for k in ("period", "frequency", "stance_end", "swing_start", "swing_end", "stance_start", "stance_period", "swing_period"):
            setattr(self, k, getattr(probe, k))
        if self.walk_state == MOVING:
            for leg in self.legs:                             # updatePhase: step_progress_ (= phase / old period, as the last iteratePhase left it) of the new period
                leg.phase = int((leg.phase / old_period) * self.period)
                self.update_step_state(leg)
        self.limits = {k: list(v) for k, v in lim.items()}    # generateLimits(): all four maps
        self.adjust = None

    def cycle(self, lin, ang):
        """One StateController::loop with robot_state RUNNING (state_controller.cpp:162-193, 429-445)."""
        pose, adm = self.prologue()
        if self.adjust is not None:      # runningState: "Dynamically adjust parameters" (state_controller.cpp:411-414), before updateWalk
            self.adjust_parameter(np.array(lin, dtype=float), float(ang))
        self.update_walk(lin, ang)
        self.update_manual()
        if self.q is not None:   # PoseController::updateStance + Model::updateModel: tips as seen from the posed body, one IK step per leg
            for i, leg in enumerate(self.legs):
                lp = pose                                                 # the leg's pose: the body's with auto_pose_ replaced by the leg's own (:118-121)
                if self.P.get("auto_posing") and not self.P.get("imu_posing"):
                    lp = remove_pose(pose, self.auto_pose_now).add(leg.auto_pose)
                poser_tip = lp.r.inv().apply(leg.tip - lp.p)              # Pose::inverseTransformVector (pose_controller.cpp:122-131)
                ddir = lp.r.inv().apply(leg.cur_dir) if leg.rot_defined else None   # pose.rotation^-1 * walker tip rotation (:129-130)
                delta = adm[i]
                if leg.leg_state in (1, -1):                              # MANUAL / WALKING_TO_MANUAL: no posing (:134-137), no delta (model.cpp:655-656)
                    poser_tip, ddir, delta = leg.tip.copy(), (leg.cur_dir if leg.rot_defined else None), np.zeros(3)
                leg.poser_tip, leg.poser_dir = poser_tip, ddir                # LegPoser::current_tip_pose_ (a waiting planner-mode robot's updateModel re-reads it)
                leg.desired_tip = poser_tip + delta
                n = len(MODEL.links[i])      # (legs of a robot may differ in joint count: the arrays are [legs][longest], a shorter leg's tail stays 0)
                self.q[i][:n], self.qd[i][:n] = apply_ik(i, self.q[i][:n], self.qd[i][:n], poser_tip + delta, self.dt, ddir, held=leg.held)  # setDesiredTipPose(.., apply_delta)
                leg.held = None
                leg.model_tip = fk_tip(i, self.q[i])                                                     # applyFK closes applyIK
                leg.model_dir = tip_axis(i, self.q[i])
                tip_force_estimate(i, self.q[i][:n], self.efforts[i][:n], self.tip_force_calc[i], self.P.get("force_gain", 0.1))   # ... and calculateTipForce


def make_params(gait, morphology=None):
    from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
    if morphology == "8x5":   # BASELINE.json config 4's synthetic octopod (8 legs x 5 joints)
        return synthetic_octopod_params(gait, 5, 8)
    if morphology == "mixed": # a hexapod whose legs have 3 / 5 / 4 / 3 / 5 / 4 joints (Parameters::leg_DOF is per leg)
        from syropod_highlevel_controller_amd import synthetic_mixed_dof_params
        return synthetic_mixed_dof_params(gait)
    return default_hexapod_params(gait)


def hexapod(gait, morphology=None, **kw):
    from syropod_highlevel_controller_amd.params import AUTO_POSES, GAITS
    p = make_params(gait, morphology)
    if morphology:            # walking only: no auto-pose tables for the synthetic morphology
        P = dict(time_delta=p.time_delta, step_frequency=p.step_frequency, swing_height=p.swing_height, swing_width=p.swing_width,
                 body_clearance=p.body_clearance, force_normal_touchdown=0, velocity_input_mode="throttle", rough_terrain_mode=0, step_depth=0.0,
                 stance_position=[[p.stance_position[l][0], p.stance_position[l][1]] for l in range(p.leg_count)],
                 stance_phase=p.stance_phase, swing_phase=p.swing_phase, phase_offset=p.phase_offset,
                 offset_multiplier=[p.offset_multiplier[l] for l in range(p.leg_count)], n_auto_posers=0, admittance_control=0,
                 max_rotation=[p.max_rotation[i] for i in range(3)], rotation_pid_gains=[0.2, 0.02, 0.01],
                 virtual_mass=p.virtual_mass, virtual_stiffness=p.virtual_stiffness, virtual_damping_ratio=p.virtual_damping_ratio,
                 integrator_step_time=p.integrator_step_time, force_gain=p.force_gain, use_joint_effort=0)
        P.update(kw)
        return P
    P = dict(time_delta=p.time_delta, step_frequency=p.step_frequency, swing_height=p.swing_height, swing_width=p.swing_width,
             body_clearance=p.body_clearance, force_normal_touchdown=0, velocity_input_mode="throttle", rough_terrain_mode=0, step_depth=0.0,
             stance_position=[[p.stance_position[l][0], p.stance_position[l][1]] for l in range(6)], **GAITS[gait])
    a = AUTO_POSES[gait]
    P.update(n_auto_posers=0, max_rotation=[p.max_rotation[i] for i in range(3)], rotation_pid_gains=[0.2, 0.02, 0.01])
    P.update(pose_phase_length=a["pose_phase_length"], pose_phase_starts=a["pose_phase_starts"], pose_phase_ends=a["pose_phase_ends"],
             roll_amplitudes=a["roll"], pitch_amplitudes=a["pitch"], yaw_amplitudes=a["yaw"], x_amplitudes=a["x"], y_amplitudes=a["y"],
             z_amplitudes=a["z"], pose_negation_phase_starts=a["pose_negation_phase_starts"], pose_negation_phase_ends=a["pose_negation_phase_ends"],
             negation_transition_ratio=a["negation_transition_ratio"])
    P.update(virtual_mass=p.virtual_mass, virtual_stiffness=p.virtual_stiffness, virtual_damping_ratio=p.virtual_damping_ratio,
             integrator_step_time=p.integrator_step_time, force_gain=p.force_gain, admittance_control=0, use_joint_effort=0,
             dynamic_stiffness=0, swing_stiffness_scaler=p.swing_stiffness_scaler, load_stiffness_scaler=p.load_stiffness_scaler)
    P.update(manual_posing=0, inclination_posing=0, max_translation=[p.max_translation[i] for i in range(3)],
             max_translation_velocity=p.max_translation_velocity, max_rotation_velocity=p.max_rotation_velocity)
    P.update(kw)
    return P


START_UP_TIME = 2.0   # time_to_start of every scenario: 100 start-up steps, where the start-up iteration is well-posed for 3- and 5-joint chains alike


def init_chain_module():
    """tests/golden/make_init_golden.py: the independent numpy restatement of the init chain (direct start-up solve, workspace search,
    walkspace, limits).  Loaded lazily - it builds on this module's kinematic model."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_init_golden", os.path.join(HERE, "make_init_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


_MI = None


def init_chain_of(gait, morphology=None, rough=0, gravity=0):
    """Start-up joint configuration and limit tables of a scenario, from the numpy init chain (nothing from oracle/ or the product)."""
    global _MI
    if _MI is None:
        _MI = init_chain_module()
    r = _MI.init_chain(gait, morphology, bool(rough), START_UP_TIME, gravity=bool(gravity))
    top = max(len(q) for q in r["q0"])           # (legs may differ in joint count: [legs][longest], zero tails)
    q0 = np.array([list(q) + [0.0] * (top - len(q)) for q in r["q0"]])
    return q0, {k: [float(x) for x in v] for k, v in r["limits"].items()}


def workspaces_of(gait, morphology=None, rough=0, gravity=0):
    """Per leg {height: {bearing: radius}}: Leg::generateWorkspace of the numpy init chain (the single plane, or the layered workspace)."""
    global _MI
    if _MI is None:
        _MI = init_chain_module()
    return _MI.init_chain(gait, morphology, bool(rough), START_UP_TIME, gravity=bool(gravity))["workspaces"]


def started_walker(P, gait="tripod", morphology=None):
    """A RefWalker (default hexapod, or the synthetic 8 x 5 octopod) as it stands when it has entered RUNNING: joints from the numpy init
    chain's direct start-up, then the loop that enters RUNNING (one cycle with zero inputs, state_controller.cpp:277-281, :189-192).
    For the sibling generators."""
    global MODEL
    q_startup, limits = init_chain_of(gait, morphology, P.get("rough_terrain_mode", 0), int(bool(P.get("gravity_aligned_tips"))))
    MODEL = Morphology.from_params(make_params(gait, morphology)) if morphology else Morphology.default_hexapod()
    w = RefWalker(P, limits)
    w.q, w.qd = q_startup.copy(), np.zeros_like(q_startup)
    for i, leg in enumerate(w.legs):
        leg.model_tip, leg.model_dir = fk_tip(i, w.q[i]), tip_axis(i, w.q[i])
    w.efforts = np.zeros_like(w.q)
    w.cycle((0.0, 0.0), 0.0)
    return w


SCENARIOS = {
    # name: (gait, parameter overrides, [(first cycle, (vx, vy), omega)], cycles)
    "tripod_start_walk_stop_restart": ("tripod", {"model": 1}, [(0, (0.6, -0.3), 0.4), (230, (0, 0), 0.0), (470, (-0.2, 0.7), -0.6)], 700),
    "wave_turn_on_the_spot": ("wave", {}, [(0, (0, 0), 1.0), (500, (0, 0), 0.0)], 1000),
    "ripple_overdriven_throttle": ("ripple", {}, [(0, (2.0, 1.5), -0.2), (300, (0.1, 0.0), 1.7)], 620),
    "amble_real_velocity_mode": ("amble", {"velocity_input_mode": "real"}, [(0, (0.05, 0.01), 0.05), (260, (0, 0), 0.0)], 560),
    "tripod_force_normal_touchdown": ("tripod", {"force_normal_touchdown": 1, "swing_width": 0.01}, [(0, (0.4, 0.5), -0.3)], 300),
    "tripod_auto_posing": ("tripod", {"auto_posing": 1, "n_auto_posers": None}, [(0, (0.7, 0.0), 0.0), (250, (0, 0), 0.0), (520, (0.0, 0.5), 0.5)], 760),
    "tripod_auto_posing_model": ("tripod", {"auto_posing": 1, "n_auto_posers": None, "model": 1}, [(0, (0.6, 0.1), 0.1), (250, (0, 0), 0.0), (520, (0.0, 0.5), 0.5)], 700),
    "wave_auto_posing_model": ("wave", {"auto_posing": 1, "n_auto_posers": None, "model": 1}, [(0, (0.5, -0.1), -0.1)], 500),
    "wave_imu_posing": ("wave", {"imu_posing": 1, "model": 1}, [(0, (0.5, 0.2), 0.1)], 400),
    # config 3's path: wave gait + admittance (tip force z ~ U(0, 20) N, x, y ~ N(0, 1), a new sample every 10 cycles) + IMU posing
    "wave_admittance_imu": ("wave", {"imu_posing": 1, "admittance_control": 1, "model": 1}, [(0, (0.4, -0.2), 0.15)], 400),
    # config 4's path: the synthetic 8 x 5 octopod, ripple gait - redundant chains, the null-space term of the DLS step at work
    "octopod_8x5_ripple": ("ripple", {"model": 1, "morphology": "8x5"}, [(0, (0.5, 0.3), -0.25), (300, (0, 0), 0.0)], 480),
    # gravity-aligned tips on the 8 x 5 octopod: updateTipRotation + the rotation-constrained IK pass and its retry
    # a hexapod whose legs have 3 / 5 / 4 / 3 / 5 / 4 joints (Parameters::leg_DOF is per leg): every leg its own chain here
    "hexapod_mixed_dof_ripple": ("ripple", {"model": 1, "morphology": "mixed"}, [(0, (0.45, 0.2), -0.2), (300, (0, 0), 0.0)], 460),
    "octopod_8x5_gravity_aligned_tips": ("ripple", {"model": 1, "morphology": "8x5", "gravity_aligned_tips": 1}, [(0, (0.4, -0.2), 0.2), (260, (0, 0), 0.0)], 420),
    # ... and with admittance deltas large enough (U(0, 20) N) that the constrained attempt misses IK_TOLERANCE: the unconstrained retry
    "octopod_8x5_gravity_aligned_admittance": ("ripple", {"model": 1, "morphology": "8x5", "gravity_aligned_tips": 1, "admittance_control": 1},
                                               [(0, (0.4, 0.3), -0.2)], 300),
    # joystick body posing with every reset mode, plus inclination posing from IMU samples
    "tripod_manual_and_inclination_posing": ("tripod", {"manual_posing": 1, "inclination_posing": 1, "model": 1, "pose_inputs": 1},
                                             [(0, (0.4, 0.2), 0.1), (330, (0, 0), 0.0)], 480),
    # gravity_aligned_tips on 3-joint legs: the experimental tip-align body pose
    "tripod_tip_align_pose": ("tripod", {"gravity_aligned_tips": 1, "model": 1}, [(0, (0.5, -0.1), 0.15), (280, (0, 0), 0.0)], 420),
    # a gait change on the move: the robot is stopped, the step cycle, phase offsets and limit tables are regenerated, it walks on
    "tripod_to_wave_gait_change": ("tripod", {"model": 1, "gait_change": "wave"}, [(0, (0.5, 0.1), 0.2), (330, (0.3, -0.2), -0.2)], 640),
    # run-time parameter adjustment (StateController::adjustParameter): swing height / width and the virtual spring's constants take effect in the next
    # loop; a higher step frequency first slows the robot down to the new cycle's speed limits, then replaces the step cycle under the walking legs; a lower
    # one afterwards (its phase offsets exceed the period still in force while it waits)
    "tripod_parameters_adjusted_on_the_move": ("tripod", {"model": 1, "admittance_control": 1,
                                                          "adjust": [[60, "swing_height", 0.035], [90, "swing_width", 0.012], [120, "virtual_stiffness", 15.0],
                                                                     [150, "force_gain", 0.16], [180, "step_frequency", 1.5], [420, "step_frequency", 0.8]]},
                                               [(0, (0.7, 0.2), 0.25), (560, (0, 0), 0.0), (620, (0.4, -0.3), -0.2)], 900),
    # the published per-leg virtual stiffness of dynamic_stiffness (swing legs soften, their neighbours stiffen)
    "ripple_dynamic_stiffness": ("ripple", {"admittance_control": 1, "dynamic_stiffness": 1, "model": 1}, [(0, (0.5, 0.1), 0.2), (200, (0, 0), 0.0)], 330),
    # the tip-force estimate in the loop: admittance driven by Leg::calculateTipForce from measured joint torques (a new sample every 10 cycles)
    "tripod_admittance_from_joint_efforts": ("tripod", {"admittance_control": 1, "use_joint_effort": 1, "model": 1, "efforts": 1},
                                             [(0, (0.5, -0.2), 0.2), (260, (0, 0), 0.0)], 420),
    # rough terrain mode WITH the kinematic model: tip-state messages from a synthetic terrain (12 mm bumps under legs 0 / 3, 10 mm
    # hollows under legs 1 / 4) drive touchdown detection, the proactive target shift and the ground-contact swing nodes
    "tripod_rough_contacts": ("tripod", {"rough_terrain_mode": 1, "step_depth": 0.004, "model": 1, "contacts": 1}, [(0, (0.45, 0.1), 0.15), (420, (0, 0), 0.0)], 600),
    # rough terrain mode without the kinematic model in the loop: requested targets / default poses, and the reactive step depth
    # stance_span_modifier: updateDefaultTipPosition shifts the identity tips sideways by a share of the workspace radius - at every stop on
    # the single plane (calculateStanceSpanChange, :949-980), at every swing / stance start on the layered workspace of rough terrain mode
    "tripod_wider_stance_span": ("tripod", {"stance_span_modifier": 0.3, "model": 1}, [(0, (0.5, 0.1), 0.2), (200, (0, 0), 0.0), (420, (0.3, -0.3), -0.3), (640, (0, 0), 0.0)], 800),
    "ripple_rough_narrower_stance_span": ("ripple", {"rough_terrain_mode": 1, "step_depth": 0.004, "stance_span_modifier": -0.25, "model": 1},
                                          [(0, (0.4, -0.1), 0.2), (330, (0, 0), 0.0)], 500),
    # requested targets that carry tip rotations, on the 8 x 5 octopod in rough terrain mode WITHOUT gravity-aligned tips: the stepper's target
    # rotation is whatever the last request left (removePose: pose.rotation * transform.rotation^-1), the tip turns towards it in the second
    # half of the swing, the stance keeps it (rotation-constrained IK), a request without a rotation makes it undefined again (:1068-1071, :1193-1234)
    "octopod_8x5_rough_target_rotations": ("ripple", {"rough_terrain_mode": 1, "model": 1, "morphology": "8x5"}, [(0, (0.35, 0.1), 0.15), (330, (0, 0), 0.0)], 440),
    "tripod_rough_external_requests": ("tripod", {"rough_terrain_mode": 1}, [(0, (0.5, 0.1), 0.2), (330, (0, 0), 0.0)], 520),
    "ripple_rough_reactive_step_depth": ("ripple", {"rough_terrain_mode": 1, "step_depth": 0.004}, [(0, (0.3, -0.2), -0.3)], 260),
}


def rough_events(name, P):
    """TargetTipPose messages / tf refreshes / tip-state messages of the rough-terrain scenarios: (cycle, kind, leg, numbers)."""
    ev = []
    if name == "tripod_rough_external_requests":
        sp = P["stance_position"]
        for c, leg, dx, dy, dz, clearance, odom in ((70, 0, 0.03, -0.02, 0.01, 0.03, 0), (70, 3, -0.02, 0.02, -0.01, 0.05, 1), (165, 4, 0.02, 0.03, 0.0, 0.02, 1),
                                                    (240, 1, -0.03, -0.01, 0.015, 0.04, 0)):
            ev.append((c, "target", leg, [sp[leg][0] + dx, sp[leg][1] + dy, dz, 1, 0, 0, 0, clearance, odom]))
        for c, leg, dx, dy, dz in ((120, 2, 0.02, 0.01, -0.008), (120, 5, -0.015, 0.02, 0.006), (300, 2, 0.0, 0.0, 0.0)):
            ev.append((c, "default", leg, [sp[leg][0] + dx, sp[leg][1] + dy, dz, 1, 0, 0, 0, 0.0, 0]))
        ev.append((310, "withdraw_default", 5, []))
        for c in range(72, 330, 4):     # generateExternalTargetTransforms: the walk plane frame has moved since the request
            k = (c - 72) / 4.0
            yaw = 0.002 * k
            tr = [0.0006 * k, -0.0004 * k, 0.0002 * k, float(np.cos(yaw / 2)), 0.0, 0.0, float(np.sin(yaw / 2))]
            for leg in range(6):
                ev.append((c, "transform_target", leg, tr))
                ev.append((c, "transform_default", leg, [0.5 * t if i < 3 else t for i, t in enumerate(tr)]))
    elif name == "octopod_8x5_rough_target_rotations":
        sp = P["stance_position"]
        tilt = lambda rv: [float(x) for x in (R.from_rotvec(rv) * from_two_vectors(np.array([1.0, 0, 0]), np.array([0, 0, -1.0]))).as_quat()[[3, 0, 1, 2]]]
        # requested targets WITH tip rotations (the tip tilted off the vertical), one of them later replaced by a request without a rotation
        for c, leg, dx, dy, dz, clearance, rv in ((60, 0, 0.02, -0.015, 0.0, 0.03, [0.25, 0.0, 0.0]), (60, 5, -0.015, 0.02, 0.004, 0.04, [0.0, -0.3, 0.0]),
                                                  (150, 2, 0.02, 0.02, 0.0, 0.03, [0.15, 0.2, 0.0]), (230, 0, 0.01, 0.01, 0.0, 0.03, None)):
            q = tilt(rv) if rv is not None else [0.0, 0.0, 0.0, 0.0]
            ev.append((c, "target", leg, [sp[leg][0] + dx, sp[leg][1] + dy, dz, *q, clearance, 0]))
        for c in range(62, 300, 6):     # generateExternalTargetTransforms: a slowly yawing, drifting walk plane frame
            k = (c - 62) / 6.0
            yaw = 0.003 * k
            tr = [0.0005 * k, -0.0003 * k, 0.0, float(np.cos(yaw / 2)), 0.0, 0.0, float(np.sin(yaw / 2))]
            for leg in range(8):
                ev.append((c, "transform_target", leg, tr))
    elif name == "tripod_manual_and_inclination_posing":
        erng = np.random.default_rng(4242)
        for c in range(5, 480, 35):      # bodyPoseInputCallback: normalised velocity inputs (some axes idle)
            v = erng.uniform(-1, 1, 6) * (erng.random(6) < 0.6)
            ev.append((c, "pose_input", -1, [float(x) for x in v]))
        for c, mode in ((150, 4), (215, 0), (260, 1), (300, 0), (340, 3), (365, 2), (400, 0), (430, 5), (450, 0)):
            ev.append((c, "pose_reset_mode", -1, [mode]))   # poseResetCallback: ALL / Z_AND_YAW / PITCH_AND_ROLL / X_AND_Y / IMMEDIATE_ALL / NO_RESET
    elif name == "ripple_rough_reactive_step_depth":
        ev.append((0, "zero_tip_force", -1, []))   # tip-state messages arrive (touchdown detection on), no contact is ever sensed
    return ev


def run(name):
    gait, over, schedule, cycles = SCENARIOS[name]
    over = dict(over)
    morphology = over.pop("morphology", None)
    P = hexapod(gait, morphology)
    if "n_auto_posers" in over:
        over["n_auto_posers"] = len(P["pose_phase_starts"])
    P.update(over)
    prod = {"force_normal_touchdown": P["force_normal_touchdown"], "swing_width": P["swing_width"], "rough_terrain_mode": P["rough_terrain_mode"],
            "step_depth": P["step_depth"], "gravity_aligned_tips": int(bool(P.get("gravity_aligned_tips")))}
    q_startup, limits = init_chain_of(gait, morphology, prod["rough_terrain_mode"], prod["gravity_aligned_tips"])
    w = RefWalker(P, limits)
    if P.get("stance_span_modifier"):
        w.workspaces = workspaces_of(gait, morphology, prod["rough_terrain_mode"], prod["gravity_aligned_tips"])
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    out = dict(tips=[], phase=[], state=[], walk_state=[], velocity=[], pose=[], odometry=[], lin=[], ang=[], imu_q=[], gyro=[], default=[], target=[], force=[], contact_force=[], effort=[], tip_force_calc=[], stiffness=[], gait_request=[],
               adjust_request=[], adjust_value=[], period=[])
    adjustments = over.pop("adjust", None)
    P.pop("adjust", None)
    if adjustments:      # generateLimits for a new step cycle needs the walkspace and the default tips the init chain ended on
        global _MI
        if _MI is None:
            _MI = init_chain_module()
        ch = _MI.chain(morphology, bool(prod["rough_terrain_mode"]), bool(prod["gravity_aligned_tips"]), START_UP_TIME)
        w.walkspace, w.default_tips = ch["walkspace"], ch["defaults"]
    events = rough_events(name, P)
    gait_changed, meta_new_limits = False, {}
    lin, ang = (0.0, 0.0), 0.0
    start = None
    if over.get("model"):     # the joint state of a robot that has gone through its direct start-up: the numpy init chain's configuration
        global MODEL
        pp = make_params(gait, morphology)
        MODEL = Morphology.from_params(pp) if morphology else Morphology.default_hexapod()
        w.q, w.qd = q_startup.copy(), np.zeros_like(q_startup)
        for i_, leg_ in enumerate(w.legs):
            leg_.model_tip = fk_tip(i_, w.q[i_])
            leg_.model_dir = tip_axis(i_, w.q[i_])
        w.efforts = np.zeros_like(w.q)
    w.cycle(lin, ang)  # the loop that enters RUNNING runs one cycle with zero inputs (state_controller.cpp:277-281, :189-192)
    if over.get("model"):
        start = np.stack([w.q, w.qd])   # ... after that first loop: what the replay's robot must hold when it is handed over
        out["q"], out["poser_tip"], out["model_tip"] = [], [], []
    for c in range(cycles):
        for first, l, a in schedule:
            if c == first:
                lin, ang = l, a
        if (P.get("imu_posing") or P.get("inclination_posing")) and c % 25 == 0:  # a new IMU sample every 25 cycles
            e = [rng.uniform(-0.15, 0.15), rng.uniform(-0.15, 0.15), 0.0]
            w.imu_q, w.gyro = euler_to_rot(e), rng.normal(0, 0.05, 3)
        if adjustments:    # parameterAdjustCallback / dynamicParameterCallback between two loops; runningState asks adjustParameter in every loop until it is set
            for ac, aname, avalue in adjustments:
                if ac == c:
                    assert w.adjust is None
                    w.adjust = (aname, avalue)
            from syropod_highlevel_controller_amd.params import PARAM_FIELD
            ids = {v: k for k, v in PARAM_FIELD.items()}
            out["adjust_request"].append(ids[w.adjust[0]] if w.adjust else 0)
            out["adjust_value"].append(w.adjust[1] if w.adjust else 0.0)
        if over.get("gait_change") and c >= 160 and not gait_changed:   # gaitSelectionCallback at cycle 160; changeGait every loop until done
            out["gait_request"].append(1)
            if w.walk_state != STOPPED:
                lin, ang = (0.0, 0.0), 0.0                                # "Stopping Syropod to change gait" (:531-536)
            else:
                from syropod_highlevel_controller_amd.params import GAITS
                P.update(GAITS[over["gait_change"]])
                w.limits = init_chain_of(over["gait_change"], morphology, prod["rough_terrain_mode"], prod["gravity_aligned_tips"])[1]   # generateLimits for the new step cycle
                w.step_cycle()
                gait_changed = True
                meta_new_limits.update(w.limits)
        else:
            out["gait_request"].append(0)
        for ec, kind, leg, v in events:          # callbacks arrive between loops; the tf refresh is the first thing a loop does
            if ec != c:
                continue
            mk = lambda a: Pose(a[0:3], R.from_quat([a[4], a[5], a[6], a[3]]) if any(a[3:7]) else None)   # (a zero quaternion: rotation undefined)
            if kind in ("target", "default") and w.walk_state != STOPPED:   # targetTipPoseCallback (state_controller.cpp:1734-1757)
                rec = dict(pose=mk(v), transform=Pose(), clearance=v[7], odom_ideal=bool(v[8]), rot_defined=bool(any(v[3:7])))
                if kind == "target":
                    w.legs[leg].ext_target = rec
                else:
                    w.legs[leg].ext_default = rec
            elif kind == "withdraw_default":
                w.legs[leg].ext_default = None
            elif kind == "transform_target" and w.legs[leg].ext_target is not None:
                w.legs[leg].ext_target["transform"] = mk(v)
            elif kind == "transform_default" and w.legs[leg].ext_default is not None:
                w.legs[leg].ext_default["transform"] = mk(v)
            elif kind == "pose_input":
                w.tvi, w.rvi = np.array(v[:3]), np.array(v[3:])
            elif kind == "pose_reset_mode":
                w.reset_mode = int(v[0])
            elif kind == "zero_tip_force":
                for l_ in w.legs:
                    l_.touchdown_detection = True
        if over.get("efforts"):      # jointStatesCallback: measured joint torques
            if c % 10 == 0:
                w.efforts = rng.normal(0, 0.5, w.q.shape)
            out["effort"].append(w.efforts.copy())
        if over.get("contacts"):     # tipStatesCallback: a wrench per leg from the synthetic terrain, then Leg::touchdownDetection
            terrain = [0.012, -0.010, 0.0, 0.012, -0.010, 0.0]
            forces = np.array([[0.0, 0.0, 3.0] if leg.tip[2] <= terrain[i] + 1e-9 else [0.0, 0.0, 0.0] for i, leg in enumerate(w.legs)])
            out["contact_force"].append(forces)
            for i, leg in enumerate(w.legs):
                leg.touchdown_detection = True
                if np.linalg.norm(forces[i]) > TOUCHDOWN_THRESHOLD and leg.step_plane is None:
                    leg.step_plane = leg.model_tip.copy()
                elif np.linalg.norm(forces[i]) < LIFTOFF_THRESHOLD:
                    leg.step_plane = None
        if P.get("admittance_control") and c % 10 == 0:
            w.tip_force = np.stack([rng.normal(0, 1, w.L), rng.normal(0, 1, w.L), rng.uniform(0, 20, w.L)], axis=1)
        out["force"].append(w.tip_force.copy())
        q = w.imu_q.as_quat()
        out["imu_q"].append([q[3], q[0], q[1], q[2]])
        out["gyro"].append(w.gyro.tolist())
        out["lin"].append(list(lin))
        out["ang"].append(ang)
        w.cycle(lin, ang)
        out["tips"].append([leg.tip.tolist() for leg in w.legs])
        out["default"].append([leg.default.tolist() for leg in w.legs])
        out["target"].append([leg.target.tolist() for leg in w.legs])
        if w.q is not None:
            out["q"].append(w.q.copy())
            out["poser_tip"].append([leg.poser_tip for leg in w.legs])   # LegPoser / Leg current tip positions (LegState.msg poser_tip_pose, model_tip_pose)
            out["model_tip"].append([leg.model_tip for leg in w.legs])
            out["tip_force_calc"].append(w.tip_force_calc.copy())
            out["stiffness"].append(list(w.stiffness))
        out["phase"].append([leg.phase for leg in w.legs])
        out["state"].append([leg.state for leg in w.legs])
        out["walk_state"].append(w.walk_state)
        if adjustments:
            out["period"].append(w.period)
        out["velocity"].append([w.v[0], w.v[1], w.w])
        out["pose"].append(w.current_pose.as7())
        out["odometry"].append(w.odometry.as7())
    if morphology:
        over["morphology"] = morphology
    if adjustments:
        over["adjust"] = adjustments
    meta = dict(gait=gait, overrides={k: v for k, v in over.items()}, schedule=schedule, cycles=cycles, limits=limits, events=events, new_limits=meta_new_limits,
                time_to_start=START_UP_TIME,
                visited_walk_states=sorted(set(out["walk_state"])))
    arrays = {k: np.array(v) for k, v in out.items() if len(v)}
    if start is not None:
        arrays["joint_start"] = start
    return arrays, meta


if __name__ == "__main__":
    arrays, metas = {}, {}
    only = sys.argv[sys.argv.index("--only") + 1:] if "--only" in sys.argv else None
    if only:            # regenerate the named scenarios only; every other array of the fixture stays byte for byte what it is
        old = np.load(os.path.join(HERE, "walk_golden.npz"))
        arrays = {k: old[k] for k in old.files if k.split("/", 1)[0] not in only}
        metas = {k: v for k, v in json.load(open(os.path.join(HERE, "walk_golden_meta.json"))).items() if k not in only}
    for name in (only or SCENARIOS):
        a, m = run(name)
        for k, v in a.items():
            arrays[f"{name}/{k}"] = v
        metas[name] = m
        print(f"{name}: {m['cycles']} cycles, walk states visited {m['visited_walk_states']}, max |tip| {np.abs(a['tips']).max():.3f}")
    np.savez_compressed(os.path.join(HERE, "walk_golden.npz"), **arrays)
    json.dump(metas, open(os.path.join(HERE, "walk_golden_meta.json"), "w"), indent=1)

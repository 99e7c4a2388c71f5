"""Golden vectors of the INIT CHAIN from an INDEPENDENT numpy restatement (this file) - no code shared with oracle/ or the engine.

Written from the reference sources alone (paths relative to /root/reference); what it restates:

  PoseController::directStartup            src/pose_controller.cpp:463-517   (a copy of the leg is stepped to its default tip with
                                                                               LegPoser::stepToPosition + Leg::applyIK(true); its joints become
                                                                               the desired configuration, reached by transitionConfiguration)
  Leg::init, updateDefaultConfiguration    src/model.cpp:286-305, :592-600
  Model::generateWorkspaces                src/model.cpp:120-138             (a copy of the model, legs re-initialised to the default configuration)
  Leg::generateWorkspace                   src/model.cpp:309-510             (simple: one plane; rough_terrain_mode: the vertical limits, then
                                                                               WORKSPACE_LAYERS planes from the top down, each from its own origin)
  Leg::getWorkplane                        src/model.cpp:514-551             (interpolation between the bounding planes, setPrecision(height, 3))
  WalkController::generateWalkspace        src/walk_controller.cpp:57-227    (adjacent-leg overlap, symmetric minimum over the legs' workplanes,
                                                                               the shifted-default branch is not taken: defaults = identities here)
  WalkController::generateLimits           src/walk_controller.cpp:231-361   (phase offsets, stance / swing overshoot, speed and acceleration maps)
  WalkController::generateStepCycle        src/walk_controller.cpp:365-411   (through RefWalker.step_cycle of make_walk_golden.py)

The kinematic model (DH chain, Leg::solveIK's 6 x 6 DLS inverse with the joint-limit cost gradient, updateJointPositions) and
LegPoser::stepToPosition are the numpy restatements of the sibling generators (make_walk_golden.py, make_sequence_golden.py).

The start-up solve is a fixed number of DLS steps (time_to_start / time_delta) of an iteration that amplifies rounding differences
(tests/test_oracle_conditioning.py): the fixture is generated at time_to_start = 4 s (200 steps), where two correct implementations
agree to ~1e-12 rad, and the replay (tests/test_oracle_golden.py::test_init_chain_golden) uses the same value.

usage: python tests/golden/make_init_golden.py    (writes tests/golden/init_golden.npz + init_golden_meta.json)
"""
import importlib.util
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


mw = _load("make_walk_golden")        # dh, Morphology / MODEL, solve_ik, update_joints, fk_tip, RefWalker (step cycle), round_to_int
ms = _load("make_sequence_golden")    # StepToPosition

IK_TOLERANCE = 0.005                   # model.h:17
BEARING_STEP, MAX_POSITION_DELTA, MAX_WORKSPACE_RADIUS, WORKSPACE_LAYERS = 45, 0.002, 1.0, 10   # model.h:22-25
UNASSIGNED = float(2 ** 31 - 1)        # standard_includes.h:52
BEARINGS = list(range(0, 361, BEARING_STEP))
TIME_TO_START = 4.0                    # 200 start-up steps (see the header)


def deg2rad(d):                        # standard_includes.h:64
    return d / 360.0 * 2.0 * math.pi


def rad2deg(r):                        # standard_includes.h:69
    return (r / (2.0 * math.pi)) * 360.0


def cmod(a, b):                        # standard_includes.h:76 on ints, with C's truncating %
    return int(math.fmod(int(math.fmod(a, b)) + b, b))


def set_precision(v, digits):          # standard_includes.h:143
    return mw.round_to_int(v * 10.0 ** digits) / 10.0 ** digits


def tip_quat(leg, q):                  # rotation of Leg::current_tip_pose_ (w, x, y, z)
    from scipy.spatial.transform import Rotation as R
    t = mw.dh(*mw.MODEL.base[leg])
    for k, (d, th, r, al) in enumerate(mw.MODEL.links[leg]):
        t = t @ mw.dh(d, th + q[k], r, al)
    x = R.from_matrix(t[:3, :3]).as_quat()
    return [x[3], x[0], x[1], x[2]]


def apply_ik_simulation(leg, q, qd, desired, dt):
    """Leg::setDesiredTipPose(Pose(desired, UNDEFINED_ROTATION)) + Leg::applyIK(true) (src/model.cpp:861-941): one DLS step, no velocity
    clamp, position clamp, FK, the per-axis IK_TOLERANCE check.  Returns (q, qd, ik_result)."""
    base = mw.dh(*mw.MODEL.base[leg])
    bi = np.linalg.inv(base)
    cur = mw.fk_tip(leg, q)
    delta = np.zeros(6)
    delta[:3] = (bi @ np.append(desired, 1))[:3] - (bi @ np.append(cur, 1))[:3]
    dq = mw.solve_ik(leg, q, qd, delta, False)
    qn, vn, result = mw.update_joints(leg, q, dq, dt, True)
    if (np.abs(mw.fk_tip(leg, qn) - desired) > IK_TOLERANCE).any():
        result = 0.0
    return qn, vn, result


def initial_configuration(leg):
    """Joint::default_position_ = clamped(0, min, max) (src/model.cpp:1038), velocities zero: Leg::init(true) of a new leg."""
    return np.array([min(max(0.0, mn), mx) for mn, mx, _ in mw.MODEL.joints[leg]])


def direct_startup(leg, default_tip, body_p, dt, time_to_start, identity_rotation=None):
    """The simulated test leg of PoseController::directStartup (:476-489): returns the joint positions it ends on (the desired
    configuration of the real leg, which transitionConfiguration reaches exactly: cubicBezier(a, a, b, b)(1) = b).  body_p: the
    position of Model::current_pose_, which loop() refreshes before the state machine runs (state_controller.cpp:165-167): the
    walk-plane pose (0, 0, body_clearance), rotation identity.  identity_rotation (w, x, y, z): the rotation of the default tip pose
    (gravity_aligned_tips on > 3 joints, walk_controller.cpp:36-41) - the steps then carry a tip direction and applyIK is rotation constrained."""
    q, qd = initial_configuration(leg), np.zeros(len(mw.MODEL.joints[leg]))
    stp = ms.StepToPosition(mw.fk_tip(leg, q), tip_quat(leg, q))
    calls = 0
    while True:
        progress, tip, direction = stp.step(default_tip, identity_rotation, body_p, [1, 0, 0, 0], 0.0, time_to_start)
        if identity_rotation is None:
            q, qd, _ = apply_ik_simulation(leg, q, qd, tip, dt)
        else:
            q, qd = mw.apply_ik(leg, q, qd, tip, dt, direction, simulation=True)
        stp.leg_p, stp.leg_q = mw.fk_tip(leg, q), tip_quat(leg, q)      # (the origin is only latched on a first iteration)
        calls += 1
        if progress == 100:
            return q, calls


def generate_workspace(leg, q_default, identity_tip, simple, dt):
    """Leg::generateWorkspace (src/model.cpp:309-510): {plane height: {bearing: radius}}.  identity_tip: the leg's identity tip position
    in the body frame, current_pose.inverseTransformVector(identity tip pose) (:348-349)."""
    n = len(q_default)
    max_wp, min_wp = {b: MAX_WORKSPACE_RADIUS for b in BEARINGS}, {b: 0.0 for b in BEARINGS}
    ws = {}
    default_q = np.array(q_default, dtype=float)
    q, qd = default_q.copy(), np.zeros(n)
    tip = mw.fk_tip(leg, q)
    identity = np.array(identity_tip, dtype=float)
    if np.linalg.norm(identity - tip) > IK_TOLERANCE:     # unable to reach the identity tip: zero workspace (:352-356)
        return {0.0: dict(min_wp)}
    if simple:
        ws[0.0] = dict(max_wp)
    found_lower = found_upper = simple
    max_h, min_h = (0.0, 0.0) if simple else (MAX_WORKSPACE_RADIUS, -MAX_WORKSPACE_RADIUS)
    delta_h = MAX_WORKSPACE_RADIUS / WORKSPACE_LAYERS
    search_h, bearing, within, it = 0.0, 0, True, 1
    while True:
        ident = identity + np.array([0.0, 0.0, search_h])
        if it == 1:
            within = True
            q, qd = default_q.copy(), np.zeros(n)          # init(true): back to the default configuration
            tip = mw.fk_tip(leg, q)
            if not found_lower or not found_upper:         # the vertical limits (:387-394)
                n_it = mw.round_to_int(MAX_WORKSPACE_RADIUS / MAX_POSITION_DELTA)
                origin = ident.copy()
                target = ident + (MAX_WORKSPACE_RADIUS if found_lower else -MAX_WORKSPACE_RADIUS) * np.array([0.0, 0.0, 1.0])
            elif bearing == 0:                             # track to the origin of the new workplane (:396-403)
                n_it = max(1, mw.round_to_int(delta_h / MAX_POSITION_DELTA))
                origin, target = tip.copy(), ident.copy()
            else:                                          # along the search bearing (:405-413)
                n_it = mw.round_to_int(MAX_WORKSPACE_RADIUS / MAX_POSITION_DELTA)
                origin = ident.copy()
                target = origin + np.array([MAX_WORKSPACE_RADIUS * math.cos(deg2rad(bearing)), MAX_WORKSPACE_RADIUS * math.sin(deg2rad(bearing)), 0.0])
        i = float(it) / n_it
        desired = origin * (1.0 - i) + target * i
        q, qd, result = apply_ik_simulation(leg, q, qd, desired, dt)
        tip = mw.fk_tip(leg, q)
        distance = float(np.linalg.norm(tip - ident))
        within = within and result != 0.0
        if within and it < n_it:
            it += 1
            continue
        it = 1
        if not found_lower:
            found_lower, min_h = True, -distance
            ws.setdefault(min_h, dict(min_wp))             # (std::map::insert: an existing plane stays)
            continue
        if not found_upper:
            found_upper, max_h = True, distance
            delta_h = (max_h - min_h) / WORKSPACE_LAYERS
            search_h = int(abs(max_h) / delta_h) * delta_h
            ws.setdefault(max_h, dict(min_wp))
            ws.setdefault(search_h, dict(max_wp))
            continue
        if bearing == 0:
            default_q = q.copy()                           # updateDefaultConfiguration: resets between bearings start here
        else:
            ws[search_h][bearing] = distance
        if bearing + BEARING_STEP <= 360:
            bearing += BEARING_STEP
        else:
            bearing = 0
            ws[search_h][0] = ws[search_h][360]
            search_h -= delta_h
            if search_h >= min_h:
                ws.setdefault(search_h, dict(max_wp))
            else:
                return ws


def get_workplane(ws, height):
    """Leg::getWorkplane (src/model.cpp:514-551)."""
    heights = sorted(ws)
    if not (heights[0] <= height <= heights[-1]):
        return None
    if len(ws) == 1:
        return dict(ws[0.0])
    upper = next(h for h in heights if h > height)         # upper_bound
    lower = heights[heights.index(upper) - 1]
    uh, lh = set_precision(upper, 3), set_precision(lower, 3)
    i = (height - lh) / (uh - lh)
    return {b: ws[lower][b] * (1.0 - i) + ws[upper][b] * i for b in BEARINGS}


def generate_walkspace(defaults, workplanes, overlapping):
    """WalkController::generateWalkspace (src/walk_controller.cpp:57-227); default tips = identity tips (no shift): radius =
    workplane.at(bearing) (:149-152).  workplanes[l] = the leg's interpolated workplane at the default tip's height."""
    L = len(defaults)
    walkspace = {}
    for l in range(L):
        d = defaults[l]
        a1, a2 = defaults[cmod(l + 1, L)], defaults[cmod(l - 1, L)]
        dist1, dist2 = np.linalg.norm(d - a1) / 2.0, np.linalg.norm(d - a2) / 2.0
        b1 = rad2deg(math.atan2(a1[1] - d[1], a1[0] - d[0]))
        b2 = rad2deg(math.atan2(a2[1] - d[1], a2[0] - d[0]))
        for bearing in BEARINGS:
            diff1 = abs(cmod(int(b1), 360) - bearing)      # static_cast<int>: towards zero
            diff2 = abs(cmod(int(b2), 360) - bearing)
            o1 = o2 = UNASSIGNED
            if (diff1 < 90 or diff1 > 270) and dist1 > 0.0:
                o1 = dist1 / math.cos(deg2rad(diff1))
            if (diff2 < 90 or diff2 > 270) and dist2 > 0.0:
                o2 = dist2 / math.cos(deg2rad(diff2))
            m = MAX_WORKSPACE_RADIUS if overlapping else min(o1, o2)
            m = min(m, MAX_WORKSPACE_RADIUS)
            if bearing not in walkspace or m < walkspace[bearing]:
                walkspace[bearing] = m
    for l in range(L):
        wp = workplanes[l]
        if wp is None:
            continue
        for bearing in BEARINGS:                           # (iterating the map being updated, in key order, as the reference does)
            radius = wp[bearing]
            opposite = cmod(bearing + 180, 360)
            if radius < walkspace[bearing]:
                walkspace[bearing] = radius
                walkspace[opposite] = radius
    walkspace[360] = walkspace[0]
    return walkspace


def generate_limits(P, walker, walkspace, defaults):
    """WalkController::generateLimits (src/walk_controller.cpp:231-361) for the step cycle and phase offsets `walker` holds."""
    dt = P["time_delta"]
    max_ext = 0
    for leg in walker.legs:
        if walker.swing_start < leg.phase_offset < walker.swing_end:
            max_ext = max(max_ext, walker.swing_end - leg.phase_offset)
    time_to_max_stride = (max_ext + walker.stance_period + walker.swing_period) * dt
    out = {k: [] for k in ("max_linear_speed", "max_angular_speed", "max_linear_acceleration", "max_angular_acceleration")}
    for bearing in BEARINGS:
        r = walkspace[bearing]
        on_ground = float(walker.stance_period) / walker.period
        max_speed = (r * 2.0) / (on_ground / walker.frequency)
        max_acc = max_speed / time_to_max_stride
        overshoot = 0.0
        for leg in walker.legs:
            t = float(leg.phase_offset) * dt
            time_to_swing_end = time_to_max_stride - t
            v0 = max_acc * time_to_swing_end
            stride = v0 * (on_ground / walker.frequency)
            d0 = -stride / 2.0
            d1 = d0 + v0 * t + 0.5 * max_acc * (t * t)
            d2 = max_speed * (walker.stance_period * dt - t)
            overshoot = max(overshoot, d1 + d2 - r)
        swing_overshoot = 0.5 * max_speed * walker.swing_period / (2.0 * walker.period * walker.frequency)
        if r == 0.0:
            out["max_linear_speed"].append(0.0)
            out["max_linear_acceleration"].append(UNASSIGNED)
            out["max_angular_speed"].append(0.0)
            out["max_angular_acceleration"].append(UNASSIGNED)
            continue
        scaled = (r / (r + overshoot + swing_overshoot)) * r
        stance_radius = math.sqrt(defaults[0][0] ** 2 + defaults[0][1] ** 2)
        mls = (scaled * 2.0) / (on_ground / walker.frequency)
        out["max_linear_speed"].append(mls)
        out["max_linear_acceleration"].append(mls / time_to_max_stride)
        out["max_angular_speed"].append(mls / stance_radius)
        out["max_angular_acceleration"].append((mls / stance_radius) / time_to_max_stride)
    return out


_CHAINS = {}


def chain(morphology=None, rough=False, gravity=False, time_to_start=TIME_TO_START, overlapping=False):
    """StateController's direct start-up + updateDefaultConfiguration + generateWorkspaces + the walkspace of generateWalkspace
    (state_controller.cpp:263-272); nothing here depends on the gait.  Cached per argument set."""
    key = (morphology, bool(rough), bool(gravity), float(time_to_start), bool(overlapping))
    if key in _CHAINS:
        return _CHAINS[key]
    p = mw.make_params("tripod", morphology)
    mw.MODEL = mw.Morphology.from_params(p) if morphology else mw.Morphology.default_hexapod()
    P = mw.hexapod("tripod", morphology)
    dt = P["time_delta"]
    assert abs(dt - ms.TIME_DELTA) < 1e-15
    L = len(P["stance_position"])
    defaults = [np.array([x, y, 0.0]) for x, y in P["stance_position"]]
    body_p = np.array([0.0, 0.0, P["body_clearance"]])     # Model::current_pose_ = the walk-plane pose while the robot is not RUNNING
    q0, workspaces, calls = [], [], 0
    for leg in range(L):
        # FromTwoVectors(UnitX, -UnitZ): a quarter turn about +y (walk_controller.cpp:38-40)
        rotation = [math.sqrt(0.5), 0.0, math.sqrt(0.5), 0.0] if gravity and len(mw.MODEL.joints[leg]) > 3 else None
        q, calls = direct_startup(leg, defaults[leg], body_p, dt, time_to_start, rotation)
        q0.append(q)
        workspaces.append(generate_workspace(leg, q, defaults[leg] - body_p, not rough, dt))
    workplanes = [get_workplane(ws, 0.0) for ws in workspaces]          # default shift zero: target height 0
    walkspace = generate_walkspace(defaults, workplanes, overlapping)
    _CHAINS[key] = dict(q0=q0, startup_calls=calls, workspaces=workspaces, workplanes=workplanes, walkspace=walkspace, defaults=defaults)
    return _CHAINS[key]


def limits_for(gait, morphology, walkspace, defaults):
    """generateStepCycle + generateLimits of a gait on a walkspace."""
    P = mw.hexapod(gait, morphology)
    walker = mw.RefWalker(P, {})
    walker.step_cycle()
    return generate_limits(P, walker, walkspace, defaults), walker


def init_chain(gait, morphology=None, rough=False, time_to_start=TIME_TO_START, overlapping=False, gravity=False):
    c = chain(morphology, rough, gravity, time_to_start, overlapping)
    limits, walker = limits_for(gait, morphology, c["walkspace"], c["defaults"])
    return dict(q0=c["q0"], startup_calls=c["startup_calls"], workspaces=c["workspaces"], workplanes=c["workplanes"],
                walkspace=[c["walkspace"][b] for b in BEARINGS], limits=limits, phase_offset=[leg.phase_offset for leg in walker.legs],
                step=dict(period=walker.period, swing_start=walker.swing_start, swing_end=walker.swing_end, stance_period=walker.stance_period,
                          swing_period=walker.swing_period, frequency=walker.frequency))


CASES = {
    # name: (gait, morphology, rough terrain mode, gravity-aligned tips, time_to_start)
    "hexapod_tripod": ("tripod", None, False, False, 4.0),
    "hexapod_wave": ("wave", None, False, False, 4.0),
    "hexapod_ripple": ("ripple", None, False, False, 4.0),
    "hexapod_amble": ("amble", None, False, False, 4.0),
    "hexapod_tripod_layered": ("tripod", None, True, False, 4.0),
    # redundant chains drift along their null space: 100 start-up steps keep two correct implementations within 1e-10 rad
    "octopod_8x5_ripple": ("ripple", "8x5", False, False, 2.0),
    "octopod_8x5_gravity_aligned": ("ripple", "8x5", False, True, 2.0),
    # legs of 3 / 5 / 4 / 3 / 5 / 4 joints in one robot: every leg is its own chain here (the engine pads the shorter legs instead)
    "hexapod_mixed_dof": ("ripple", "mixed", False, False, 2.0),
}


def main():
    arrays, meta = {}, {}
    for name, (gait, morphology, rough, gravity, tts) in CASES.items():
        r = init_chain(gait, morphology, rough, tts, gravity=gravity)
        top = max(len(q) for q in r["q0"])   # (legs may differ in joint count: padded with NaN)
        arrays[name + ".q0"] = np.array([list(q) + [np.nan] * (top - len(q)) for q in r["q0"]])
        arrays[name + ".workplane"] = np.array([[wp[b] for b in BEARINGS] for wp in r["workplanes"]])
        arrays[name + ".walkspace"] = np.array(r["walkspace"])
        for k, v in r["limits"].items():
            arrays[name + "." + k] = np.array(v)
        heights = [sorted(ws) for ws in r["workspaces"]]
        if rough:   # the layered workspace itself: [leg][plane][bearing], planes in ascending height
            arrays[name + ".layer_heights"] = np.array(heights)
            arrays[name + ".layers"] = np.array([[[ws[h][b] for b in BEARINGS] for h in sorted(ws)] for ws in r["workspaces"]])
        meta[name] = dict(gait=gait, morphology=morphology, rough_terrain_mode=int(rough), gravity_aligned_tips=int(gravity), time_to_start=tts, startup_calls=r["startup_calls"],
                          phase_offset=r["phase_offset"], step=r["step"], planes=[len(h) for h in heights])
        print(f"{name}: {r['startup_calls']} start-up calls, planes per leg {meta[name]['planes']}, walkspace {np.round(r['walkspace'], 4).tolist()}")
    np.savez_compressed(os.path.join(HERE, "init_golden.npz"), **arrays)
    with open(os.path.join(HERE, "init_golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()

"""Generates tests/golden/sequence_golden.npz: trajectories of the LegPoser primitives behind the start-up / shut-down sequences,
leg manipulation and planner mode, from an INDEPENDENT numpy restatement (this file) - no code shared with oracle/ or the engine.

    python tests/golden/make_sequence_golden.py          (needs numpy + scipy; writes the fixture next to this file)

Restated here, from the reference sources only (OpenSHC v0.5.11, paths relative to /root/reference):
  LegPoser::stepToPosition            src/pose_controller.cpp:1571-1712   (tip on two quartic Beziers, optional lift, body pose eased
                                                                           by smoothStep, tip direction interpolated)
  LegPoser::transitionConfiguration   src/pose_controller.cpp:1476-1567   (joints on a cubic Bezier)
  quarticBezier / cubicBezier / smoothStep / interpolate / roundToInt     include/.../standard_includes.h:93, 163, 201, 347, 402
  Pose::interpolate / inverseTransformVector                              include/.../pose.h:151-195
What is fed in as DATA: the tip pose a step starts from (origin_tip_pose_ = the leg's FK tip pose) and the joint positions a
configuration transition starts from - recorded in the fixture from a default hexapod as it stands after start-up (its FK is
pinned separately, tests/test_oracle_golden.py::test_hexapod_fk_matches_numpy_chain); the replay first checks that its robot
stands exactly there.
Rotations use scipy.spatial.transform.Rotation / Slerp (an independent implementation of quaternion algebra).

tests/test_oracle_golden.py::test_sequence_trajectories replays every scenario on the oracle (orc_leg_step_to_position /
orc_leg_transition_configuration) call by call.
"""
import os

import numpy as np
from scipy.spatial.transform import Rotation as R
from scipy.spatial.transform import Slerp

HERE = os.path.dirname(os.path.abspath(__file__))
TIP_TOLERANCE, JOINT_TOLERANCE = 0.01, 0.01   # pose_controller.h:18-19
TIME_DELTA = 0.02                              # default.yaml time_delta


def round_to_int(x):
    return int(x + 0.5) if x >= 0 else -int(0.5 - x)


def smooth_step(c):
    return 6.0 * c ** 5 - 15.0 * c ** 4 + 10.0 * c ** 3


def quartic_bezier(nodes, t):
    s = 1.0 - t
    return nodes[0] * s ** 4 + nodes[1] * (4 * t * s ** 3) + nodes[2] * (6 * t * t * s * s) + nodes[3] * (4 * t ** 3 * s) + nodes[4] * t ** 4


def cubic_bezier(nodes, t):
    s = 1.0 - t
    return nodes[0] * s ** 3 + nodes[1] * (3 * t * s * s) + nodes[2] * (3 * t * t * s) + nodes[3] * t ** 3


def rot(q_wxyz):
    return R.from_quat([q_wxyz[1], q_wxyz[2], q_wxyz[3], q_wxyz[0]])


def from_two_vectors(a, b):
    """Rotation taking a onto b about their common normal (Eigen::Quaterniond::FromTwoVectors for non-antiparallel vectors)."""
    a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
    axis = np.cross(a, b)
    s, c = np.linalg.norm(axis), float(np.dot(a, b))
    if s < 1e-300:
        return R.identity()
    return R.from_rotvec(axis / s * np.arctan2(s, c))


UNDEFINED_POSITION = np.full(3, 2147483647.0)   # standard_includes.h:57: Vector3d(double(INT_MAX), ..)


class StepToPosition:
    """One LegPoser's stepToPosition state: call step() once per control loop; returns (progress, tip position, tip direction or None)."""

    def __init__(self, origin_p, origin_q):
        self.first = True
        self.leg_p, self.leg_q = np.array(origin_p, float), np.array(origin_q, float)   # the leg's current tip pose (FK)
        self.count = 0

    def step(self, target_p, target_q, body_p, body_q, lift, time_to_step, delta=None):
        if self.first:                                                # :1574-1579
            self.origin_p, self.origin_q = self.leg_p.copy(), self.leg_q.copy()
            self.count = 0
            self.first = False
        if target_p is None and target_q is None:                     # Pose::Undefined(): stay, rotation undefined (:1581-1586)
            desired_p, desired_q = self.origin_p.copy(), None
        elif target_p is None:                                        # UNDEFINED_POSITION with a rotation (transitionStance under gravity-aligned tips):
            desired_p, desired_q = UNDEFINED_POSITION.copy(), np.array(target_q, float)   # not Pose::Undefined(), so the position goes through as it is
        else:
            desired_p, desired_q = np.array(target_p, float), (None if target_q is None else np.array(target_q, float))
        body = rot(body_q)
        inv_body = lambda v: body.inv().apply(v - np.array(body_p, float))   # Pose::inverseTransformVector
        moving = np.linalg.norm(self.origin_p - inv_body(desired_p)) > TIP_TOLERANCE       # :1589-1591
        turning = False
        x = np.array([1.0, 0, 0])
        if desired_q is not None:                                     # :1593-1599
            od, dd = rot(self.origin_q).apply(x), rot(desired_q).apply(x)
            turning = from_two_vectors(od, dd).magnitude() > JOINT_TOLERANCE
        if not moving and not turning and lift == 0.0:                # :1601-1606
            self.first = True
            return 100, self.origin_p.copy(), rot(self.origin_q).apply(x)
        if delta is not None:                                         # apply_delta and the leg is not manually manipulated (:1609-1614)
            desired_p = desired_p + np.array(delta, float)
        self.count += 1                                               # :1615
        num = max(1, round_to_int(time_to_step / TIME_DELTA))
        dt = 1.0 / num
        ratio = (self.count - 1) / num
        s = smooth_step(ratio)
        pose_p = s * np.array(body_p, float)                          # Pose::Identity().interpolate(s, target_pose) (:1623)
        pose_r = Slerp([0.0, 1.0], R.concatenate([R.identity(), body]))(s)
        direction = None
        if desired_q is not None:                                     # :1626-1634
            od, dd = rot(self.origin_q).apply(x), rot(desired_q).apply(x)
            nd = (1.0 - s) * od + s * dd
            direction = nd / np.linalg.norm(nd)
        half = num // 2                                               # :1640-1676
        o2t = self.origin_p - desired_p
        lift_v = np.array([0, 0, lift])
        prim = [self.origin_p, self.origin_p, self.origin_p + lift_v, desired_p + 0.75 * o2t + lift_v, desired_p + 0.5 * o2t + lift_v]
        sec = [desired_p + 0.5 * o2t + lift_v, desired_p + 0.25 * o2t + lift_v, desired_p + lift_v, desired_p, desired_p]
        sic = (self.count + (num - 1)) % num + 1
        if not (desired_p != UNDEFINED_POSITION).any():               # "if (desired_tip_pose.position_ != UNDEFINED_POSITION)" (:1637): the tip stays
            new_p = self.origin_p.copy()
        elif sic <= half:
            new_p = quartic_bezier(prim, sic * dt * 2.0)
        else:
            new_p = quartic_bezier(sec, (sic - half) * dt * 2.0)
        tip = pose_r.inv().apply(new_p - pose_p)                      # desired_pose.inverseTransformVector (:1680)
        if self.count >= num:                                         # :1703-1711
            self.first = True
            return 100, tip, direction
        return int(ratio * 100), tip, direction


def transition_configuration(q0, target, transition_time):
    """LegPoser::transitionConfiguration from joint positions q0: the whole trajectory [(progress, q), ...]."""
    num = max(1, round_to_int(transition_time / TIME_DELTA))
    dt = 1.0 / num
    out = []
    for count in range(1, num + 1):
        q = np.array([cubic_bezier([a, a, b, b], count * dt) for a, b in zip(q0, target)])
        progress = min(max(int(((count - 1) / num) * 100), 1), 100)
        if count >= num:
            progress = 100
        out.append((progress, q))
    return out


class Packer:
    """PoseController::packLegs / unpackLegs (src/pose_controller.cpp:615-706) over LegPoser::transitionConfiguration (:1476-1567), call by call:
    every leg runs a cubic-Bezier joint transition to the packed positions of the current pack step; a completed step hands over to the
    next one (packLegs: pack_step_++ and progress 0 until the last step; unpackLegs: back through the steps, the last one to the unpacked
    positions)."""

    def __init__(self, q, packed, unpacked):
        self.q = np.array(q, float)                     # [legs][dof] Joint::desired_position_
        self.packed, self.unpacked = np.array(packed, float), np.array(unpacked, float)   # [steps][legs][dof], [legs][dof]
        self.pack_step, self.executing = 0, False
        self.first, self.count = [True] * len(self.q), [0] * len(self.q)
        self.origin, self.desired = [None] * len(self.q), [None] * len(self.q)

    def _transition(self, leg, time):
        if self.first[leg]:
            self.origin[leg] = self.q[leg].copy()
            self.first[leg], self.count[leg] = False, 0
        num = max(1, round_to_int(time / TIME_DELTA))
        self.count[leg] += 1
        t = self.count[leg] * (1.0 / num)
        self.q[leg] = np.array([cubic_bezier([a, a, b, b], t) for a, b in zip(self.origin[leg], self.desired[leg])])
        progress = min(max(int(((self.count[leg] - 1) / num) * 100), 1), 100)
        if self.count[leg] >= num:
            self.first[leg] = True
            return 100
        return progress

    def pack(self, time):
        progress = 0
        for leg in range(len(self.q)):
            if not self.executing:
                self.desired[leg] = self.packed[self.pack_step][leg]
            progress = self._transition(leg, time)
        self.executing = progress not in (0, 100)
        if progress == 100 and self.pack_step < len(self.packed) - 1:
            self.executing = False
            self.pack_step += 1
            progress = 0
        return progress

    def unpack(self, time):
        progress = 0
        for leg in range(len(self.q)):
            if not self.executing:
                self.desired[leg] = self.packed[self.pack_step - 1][leg] if self.pack_step > 0 else self.unpacked[leg]
            progress = self._transition(leg, time)
        self.executing = progress not in (0, 100)
        if progress == 100 and self.pack_step != 0:
            self.executing = False
            self.pack_step -= 1
            progress = 0
        return progress


# (target offset from the origin tip, target rotation as a small rotation vector applied to the origin rotation or None,
#  body pose position, body rotation vector, lift, time) - the origin itself is read from the leg the scenario names at replay
STEP_SCENARIOS = {
    "plain":      dict(leg=0, offset=[0.03, -0.02, 0.01], turn=None, body_p=[0, 0, 0], body_rv=[0, 0, 0], lift=0.0, time=1.0, calls=55),
    "lift":       dict(leg=1, offset=[-0.02, 0.03, 0.0], turn=None, body_p=[0, 0, 0], body_rv=[0, 0, 0], lift=0.04, time=0.8, calls=45),
    "body":       dict(leg=2, offset=None, turn=None, body_p=[0.01, -0.008, 0.012], body_rv=[0.02, -0.01, 0.03], lift=0.0, time=1.5, calls=80),
    "turn":       dict(leg=3, offset=[0.01, 0.01, -0.01], turn=[0.1, -0.2, 0.15], body_p=[0.0, 0.005, 0.0], body_rv=[0, 0, 0.04], lift=0.02, time=0.6, calls=65),
    "nothing":    dict(leg=4, offset=[0.002, 0.001, -0.003], turn=None, body_p=[0, 0, 0], body_rv=[0, 0, 0], lift=0.0, time=1.0, calls=5),
    "odd-count":  dict(leg=5, offset=[0.02, 0.02, 0.02], turn=None, body_p=[0, 0, 0], body_rv=[0, 0, 0], lift=0.01, time=0.5, calls=30),
}
CONFIG_SCENARIOS = {
    "cfg-1s":  dict(leg=0, delta=[0.3, -0.2, 0.25], time=1.0),
    "cfg-odd": dict(leg=3, delta=[-0.15, 0.1, 0.05], time=0.33),
    "cfg-one": dict(leg=5, delta=[0.05, 0.05, -0.05], time=0.02),
}


def run_step_scenario(sc, origin_p, origin_q):
    s = StepToPosition(origin_p, origin_q)
    target_p = None if sc["offset"] is None else np.array(origin_p) + np.array(sc["offset"])
    target_q = None
    if sc["turn"] is not None:
        q = (rot(origin_q) * R.from_rotvec(sc["turn"])).as_quat()
        target_q = [q[3], q[0], q[1], q[2]]
    b = R.from_rotvec(sc["body_rv"]).as_quat()
    body_q = [b[3], b[0], b[1], b[2]]
    rows = []
    for _ in range(sc["calls"]):
        progress, tip, direction = s.step(target_p, target_q, sc["body_p"], body_q, sc["lift"], sc["time"])
        rows.append([progress, *tip, *(direction if direction is not None else [np.nan] * 3)])
    return np.array(rows), target_p, target_q, body_q


if __name__ == "__main__":
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    sys.path.insert(0, os.path.dirname(HERE))
    # the origins: FK tip poses of the default hexapod as it stands after start-up - joints from the numpy init chain (make_init_golden.py
    # through make_walk_golden.started_walker), tip poses from the numpy chain's FK: nothing from oracle/ or the product
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_walk_golden", os.path.join(HERE, "make_walk_golden.py"))
    mw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mw)
    w = mw.started_walker(mw.hexapod("tripod"), "tripod")
    q0 = w.q.copy()
    origin = []
    for l in range(6):
        t = mw.dh(*mw.MODEL.base[l])
        for k, (d, th, r_, al) in enumerate(mw.MODEL.links[l]):
            t = t @ mw.dh(d, th + q0[l][k], r_, al)
        x = R.from_matrix(t[:3, :3]).as_quat()
        origin.append([*t[:3, 3], x[3], x[0], x[1], x[2]])   # [legs][7] (x, y, z, qw, qx, qy, qz)
    origin = np.array(origin)
    out = {"origin": origin, "q0": q0}
    for name, sc in STEP_SCENARIOS.items():
        rows, target_p, target_q, body_q = run_step_scenario(sc, origin[sc["leg"], :3], origin[sc["leg"], 3:])
        out[f"step/{name}/rows"] = rows
        out[f"step/{name}/target"] = np.array([*(target_p if target_p is not None else [np.nan] * 3), *(target_q if target_q is not None else [0.0] * 4)])
        out[f"step/{name}/body"] = np.array([*sc["body_p"], *body_q])
        out[f"step/{name}/args"] = np.array([sc["leg"], sc["lift"], sc["time"]])
    for name, sc in CONFIG_SCENARIOS.items():
        traj = transition_configuration(q0[sc["leg"]], q0[sc["leg"]] + np.array(sc["delta"]), sc["time"])
        out[f"cfg/{name}/rows"] = np.array([[pr, *q] for pr, q in traj])
        out[f"cfg/{name}/target"] = q0[sc["leg"]] + np.array(sc["delta"])
        out[f"cfg/{name}/args"] = np.array([sc["leg"], sc["time"]])
    # packLegs through two pack steps, unpackLegs back to the unpacked positions (the reference's `packed` / `unpacked` joint parameters;
    # here: two synthetic pack steps folding the legs in, unpacked = default.yaml's)
    pp = mw.make_params("tripod")
    unpacked = np.array([[pp.joint[l][j].unpacked for j in range(3)] for l in range(6)])
    fold = np.array([0.0, 0.9, -0.4])
    packed = np.stack([unpacked + 0.5 * fold * np.array([1, 1, 1]), unpacked + fold + np.array([[0.3 * (1 if l < 3 else -1), 0, 0] for l in range(6)])])
    pk = Packer(q0, packed, unpacked)
    rows = []
    for fn, tag in ((pk.pack, 0), (pk.unpack, 1)):
        for _ in range(2000):
            pr = fn(0.7)
            rows.append([tag, pr, *pk.q.reshape(-1)])
            if pr == 100:
                break
        assert pr == 100
    out["pack/rows"], out["pack/packed"], out["pack/time"] = np.array(rows), packed, np.array([0.7])
    np.savez_compressed(os.path.join(HERE, "sequence_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "sequence_golden.npz"), {k: v.shape for k, v in out.items() if k.endswith("rows")})

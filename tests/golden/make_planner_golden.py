"""Generates tests/golden/planner_golden.npz: planner mode from an INDEPENDENT numpy restatement (this file + the walker / kinematic
model of make_walk_golden.py and LegPoser::stepToPosition of make_sequence_golden.py) - no code shared with oracle/ or the engine.

    python tests/golden/make_planner_golden.py

Restated here, from the reference sources only (OpenSHC v0.5.11, paths relative to /root/reference):
  StateController::executePlan                   src/state_controller.cpp:653-698   (stop first, wait = Model::updateModel, one plan step at
                                                                                      a time, plan_step_ / acquired flags / target body pose reset)
  PoseController::transitionConfiguration        src/pose_controller.cpp:710-763   (+ LegPoser::transitionConfiguration :1476-1567)
  PoseController::transitionStance               src/pose_controller.cpp:767-807   (LegPoser external target: transform.addPose(pose), swing
                                                                                      clearance, withdrawn on completion)
  targetConfigurationCallback / targetBodyPoseCallback / targetTipPoseCallback for a robot that stands   src/state_controller.cpp:1683-1767
One scenario: a default hexapod with admittance control and a steady 4 N on every tip walks, planner mode comes on (the robot is
stopped, then waits), a joint configuration for four legs, a wait, tip targets for three legs (one with a lift, all through a tf
transform) together with a body pose, a body pose alone.  Recorded per loop: executePlan's result, plan_step_, joints.  A second
run (keys imu_*) executes the same plan under IMU + inclination posing with a new IMU reading every 20 loops: the pose moves under
the robot while it stands, waits and transitions (a waiting robot's updateModel keeps the LegPoser tips of the last updateStance).

tests/test_oracle_golden.py::test_planner_trajectories replays the loops on the oracle.
"""
import importlib.util
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation as R

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


mw = _load("make_walk_golden")
ms = _load("make_sequence_golden")
PLAN_TIME = 5.0


def tip_quat(i, q):
    t = mw.dh(*mw.MODEL.base[i])
    for k, (d, th, r, al) in enumerate(mw.MODEL.links[i]):
        t = t @ mw.dh(d, th + q[k], r, al)
    x = R.from_matrix(t[:3, :3]).as_quat()
    return [x[3], x[0], x[1], x[2]]


class Planner:
    def __init__(self, w):
        self.w = w
        self.plan_step = 0
        self.configuration = None            # PoseController::target_configuration_: {leg: joint positions}
        self.body_pose = ([0.0, 0.0, 0.0], [1.0, 0, 0, 0])
        self.cfg_acquired = self.tip_acquired = self.body_acquired = False
        self.executing_transition = False
        self.targets = {}                    # LegPoser::external_target_: leg -> dict(pose_p, transform (p, q), clearance)
        # LegPoser::first_iteration_ / master_iteration_count_ are shared by stepToPosition and transitionConfiguration
        self.first = [True] * w.L
        self.count = [0] * w.L
        self.origin_cfg = [None] * w.L
        self.desired_cfg = [None] * w.L
        self.stp = [None] * w.L

    def transition_configuration(self):
        w = self.w
        num = max(1, mw.round_to_int(PLAN_TIME / w.dt))
        min_progress = 2147483647
        for i in range(w.L):
            if not self.executing_transition:
                self.desired_cfg[i] = None if i not in self.configuration else np.array(self.configuration[i], float)
            if self.desired_cfg[i] is None:
                progress = 100
            else:
                if self.first[i]:
                    self.origin_cfg[i] = w.q[i].copy()
                    self.first[i] = False
                    self.count[i] = 0
                self.count[i] += 1
                t = self.count[i] * (1.0 / num)
                w.q[i] = np.array([ms.cubic_bezier([a, a, b, b], t) for a, b in zip(self.origin_cfg[i], self.desired_cfg[i])])
                w.legs[i].model_tip = mw.fk_tip(i, w.q[i])
                progress = min(max(int(((self.count[i] - 1) / num) * 100), 1), 100)
                if self.count[i] >= num:
                    self.first[i] = True
                    progress = 100
            min_progress = min(min_progress, progress)
        self.executing_transition = min_progress not in (0, 100)
        return min_progress

    def transition_stance(self, adm):
        w = self.w
        min_progress = 2147483647
        for i, leg in enumerate(w.legs):
            t = self.targets.get(i)
            target, clearance = None, 0.0
            if t is not None:
                tp, tq = t["transform"]
                target = np.array(tp, float) + R.from_quat([tq[1], tq[2], tq[3], tq[0]]).apply(np.array(t["pose_p"], float))
                clearance = t["clearance"]
            target_q = None
            if w.P.get("gravity_aligned_tips"):                    # "Update target rotation if gravity alignment is set" (:785-790) - every leg,
                x = mw.from_two_vectors(np.array([1.0, 0, 0]), w.estimate_gravity()).as_quat()   # whatever its joint count
                target_q = [x[3], x[0], x[1], x[2]]
            if self.first[i]:
                self.stp[i] = ms.StepToPosition(mw.fk_tip(i, w.q[i]), tip_quat(i, w.q[i]))
                self.first[i] = False
            manually = leg.leg_state in (1, -1)
            progress, tip, direction = self.stp[i].step(target, target_q, self.body_pose[0], self.body_pose[1], clearance, PLAN_TIME, None if manually else adm[i])
            if self.stp[i].first:
                self.first[i] = True
            if target_q is None:
                direction = None                                   # (the "nothing to do" return hands back the origin pose; without a target rotation nothing constrains the IK)
            leg.poser_tip, leg.poser_dir = tip, direction
            desired = tip + (np.zeros(3) if manually else adm[i])
            leg.desired_tip = desired
            w.q[i], w.qd[i] = mw.apply_ik(i, w.q[i], w.qd[i], desired, w.dt, direction)
            leg.model_tip = mw.fk_tip(i, w.q[i])
            min_progress = min(min_progress, progress)
            if t is not None and progress == 100:
                del self.targets[i]
        return min_progress

    def loop(self):
        """One StateController::loop in planner mode: -1 still walking (ordinary cycle, zero inputs), -2 waiting, else progress."""
        w = self.w
        if w.walk_state != mw.STOPPED:
            w.cycle((0.0, 0.0), 0.0)
            return -1
        pose, adm = w.prologue()
        if not (self.cfg_acquired or self.tip_acquired or self.body_acquired):
            for i, leg in enumerate(w.legs):                       # Model::updateModel: the poser's tip pose + delta, one IK step
                desired = leg.poser_tip + (np.zeros(3) if leg.leg_state in (1, -1) else adm[i])
                leg.desired_tip = desired
                w.q[i], w.qd[i] = mw.apply_ik(i, w.q[i], w.qd[i], desired, w.dt, leg.poser_dir)   # (the LegPoser's tip pose with its rotation)
                leg.model_tip = mw.fk_tip(i, w.q[i])
            return -2
        progress = self.transition_configuration() if self.cfg_acquired else self.transition_stance(adm)
        if progress == 100:
            self.plan_step += 1
            self.body_pose = ([0.0, 0.0, 0.0], [1.0, 0, 0, 0])
            self.cfg_acquired = self.tip_acquired = self.body_acquired = False
        return progress


def run(posing=False, auto=False):
    import zlib
    gait = "tripod"
    P = mw.hexapod(gait, admittance_control=1, manual_posing=1)
    if posing:                                   # the body pose keeps moving while the robot stands and waits: IMU PID + inclination translation
        P.update(imu_posing=1, inclination_posing=1)
    if auto:                                     # cyclic auto posing: the posers wind down after the stop, the legs' negation windows stay where the phases froze
        P.update(auto_posing=1, n_auto_posers=len(P["pose_phase_starts"]))
    w = mw.started_walker(P, gait)               # joints: the numpy init chain's direct start-up + the first loop (nothing from oracle/ or the product)
    q0, qd0 = w.q.copy(), w.qd.copy()
    w.tip_force = np.tile(np.array([0.0, 0.0, 4.0]), (6, 1))
    rows, joints, events = [], [], []
    rng = np.random.default_rng(zlib.crc32(b"planner"))
    imu = dict(q=[1.0, 0.0, 0.0, 0.0], gyro=[0.0, 0.0, 0.0], loops=0)

    def imu_sample():                            # a new IMU reading every 20 loops (columns 3-9 of a row: orientation wxyz, angular velocity)
        if posing and imu["loops"] % 20 == 0:
            e = [rng.uniform(-0.12, 0.12), rng.uniform(-0.12, 0.12), 0.0]
            w.imu_q, w.gyro = mw.euler_to_rot(e), rng.normal(0, 0.03, 3)
            x = w.imu_q.as_quat()
            imu["q"], imu["gyro"] = [x[3], x[0], x[1], x[2]], list(w.gyro)
        imu["loops"] += 1

    for _ in range(60):
        imu_sample()
        w.cycle((0.45, -0.1), 0.15)
        rows.append([0, 0, 0, *imu["q"], *imu["gyro"]])
        joints.append(w.q.copy())
    pl = Planner(w)

    def loops_until(value, limit=2000):
        for _ in range(limit):
            imu_sample()
            r = pl.loop()
            rows.append([1, r, pl.plan_step, *imu["q"], *imu["gyro"]])
            joints.append(w.q.copy())
            if r == value:
                return
        raise AssertionError(value)

    loops_until(-2)
    for _ in range(3):
        loops_until(-2, 1)
    cfg = {l: (w.q[l] + np.array([0.08, -0.1, 0.12]) * (1 if l % 2 else -1)).tolist() for l in (0, 1, 3, 4)}
    events.append((len(rows), "configuration", {str(k): v for k, v in cfg.items()}))
    pl.configuration, pl.cfg_acquired = cfg, True
    loops_until(100)
    for _ in range(4):
        loops_until(-2, 1)
    tf = ([0.004, -0.003, 0.001], [float(np.cos(0.01)), 0.0, 0.0, float(np.sin(0.01))])
    targets = {}
    for l, off, clearance in ((0, [0.03, -0.02, 0.005], 0.0), (2, [-0.025, 0.03, 0.0], 0.02), (5, [0.02, 0.035, -0.004], 0.0)):
        targets[l] = dict(pose_p=(w.legs[l].model_tip + np.array(off)).tolist(), transform=tf, clearance=clearance)
    body = ([0.008, -0.006, 0.01], [float(np.cos(0.015)), float(np.sin(0.015)), 0.0, 0.0])
    events.append((len(rows), "stance", {"targets": {str(k): v for k, v in targets.items()}, "body": body}))
    pl.targets, pl.tip_acquired = dict(targets), True
    pl.body_pose, pl.body_acquired = body, True
    loops_until(100)
    loops_until(-2, 1)
    body2 = ([0.0, 0.01, -0.008], [1.0, 0.0, 0.0, 0.0])
    events.append((len(rows), "stance", {"targets": {}, "body": body2}))
    pl.body_pose, pl.body_acquired = body2, True
    loops_until(100)
    loops_until(-2, 1)
    return {"rows": np.array(rows), "joints": np.array(joints), "joint_start": np.stack([q0.reshape(6, 3), qd0.reshape(6, 3)])}, events


def run_gravity():
    """The synthetic 8 x 5 octopod with gravity-aligned tips: transitionStance turns EVERY tip towards Model::estimateGravity() - a leg
    without a tip target gets Pose(UNDEFINED_POSITION, that rotation), which is not Pose::Undefined(): its tip stays (:1637) and turns -
    and a waiting robot's updateModel keeps the LegPoser's tip rotation.  No admittance (a delta added to UNDEFINED_POSITION would send
    the tip after INT_MAX metres, :1613)."""
    gait, L = "ripple", 8
    P = mw.hexapod(gait, "8x5", gravity_aligned_tips=1)
    w = mw.started_walker(P, gait, "8x5")
    q0, qd0 = w.q.copy(), w.qd.copy()
    rows, joints, events = [], [], []
    for _ in range(70):
        w.cycle((0.4, -0.2), 0.2)
        rows.append([0, 0, 0])
        joints.append(w.q.copy())
    pl = Planner(w)

    def loops_until(value, limit=2000):
        for _ in range(limit):
            r = pl.loop()
            rows.append([1, r, pl.plan_step])
            joints.append(w.q.copy())
            if r == value:
                return
        raise AssertionError(value)

    loops_until(-2)
    for _ in range(3):
        loops_until(-2, 1)
    body = ([0.006, -0.004, 0.008], [float(np.cos(0.012)), 0.0, float(np.sin(0.012)), 0.0])
    events.append((len(rows), "stance", {"targets": {}, "body": body}))          # a body pose alone: every tip stays where it is and turns
    pl.body_pose, pl.body_acquired = body, True
    loops_until(100)
    for _ in range(3):
        loops_until(-2, 1)
    cfg = {l: (w.q[l] + np.array([0.05, -0.06, 0.07, -0.04, 0.03]) * (1 if l % 2 else -1)).tolist() for l in (1, 4, 6)}
    events.append((len(rows), "configuration", {str(k): v for k, v in cfg.items()}))
    pl.configuration, pl.cfg_acquired = cfg, True
    loops_until(100)
    loops_until(-2, 1)
    tf = ([0.003, -0.002, 0.001], [float(np.cos(0.008)), 0.0, 0.0, float(np.sin(0.008))])
    targets = {}
    for l, off, clearance in ((0, [0.02, -0.015, 0.004], 0.0), (3, [-0.02, 0.02, 0.0], 0.015), (7, [0.015, 0.02, -0.003], 0.0)):
        targets[l] = dict(pose_p=(w.legs[l].model_tip + np.array(off)).tolist(), transform=tf, clearance=clearance)
    body2 = ([0.0, 0.006, -0.005], [1.0, 0.0, 0.0, 0.0])
    events.append((len(rows), "stance", {"targets": {str(k): v for k, v in targets.items()}, "body": body2}))
    pl.targets, pl.tip_acquired = dict(targets), True
    pl.body_pose, pl.body_acquired = body2, True
    loops_until(100)
    loops_until(-2, 1)
    return {"rows": np.array(rows), "joints": np.array(joints), "joint_start": np.stack([q0.reshape(L, 5), qd0.reshape(L, 5)])}, events


if __name__ == "__main__":
    import json
    out, events = run()
    out2, events2 = run(posing=True)             # the same plan under IMU + inclination posing (keys imu_*)
    out.update({"imu_" + k: v for k, v in out2.items()})
    np.savez_compressed(os.path.join(HERE, "planner_golden.npz"), **out)
    json.dump(events, open(os.path.join(HERE, "planner_golden_events.json"), "w"), indent=1)
    json.dump(events2, open(os.path.join(HERE, "planner_golden_events_imu.json"), "w"), indent=1)
    out4, events4 = run(auto=True)               # the same plan with cyclic auto posing (keys auto_*)
    out.update({"auto_" + k: v for k, v in out4.items()})
    json.dump(events4, open(os.path.join(HERE, "planner_golden_events_auto.json"), "w"), indent=1)
    out3, events3 = run_gravity()                # the 8 x 5 octopod with gravity-aligned tips (keys g85_*)
    out.update({"g85_" + k: v for k, v in out3.items()})
    json.dump(events3, open(os.path.join(HERE, "planner_golden_events_8x5.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "planner_golden.npz"), **out)
    for pre in ("", "imu_", "g85_", "auto_"):
        r = out[pre + "rows"]
        print(pre or "plain", "loops", len(r), "plan results seen", sorted(set(r[r[:, 0] == 1][:, 1].astype(int).tolist()))[:6], "... final plan step", int(r[-1, 2]))

"""Generates tests/golden/startup_golden.npz: the start-up / shut-down choreography of the reference from an INDEPENDENT numpy
restatement (this file + the kinematic model and LegPoser::stepToPosition of its two siblings) - no code shared with oracle/ or
the engine.

    python tests/golden/make_startup_golden.py

Restated here, from the reference sources only (OpenSHC v0.5.11, paths relative to /root/reference):
  PoseController::executeSequence (START_UP / SHUT_DOWN)   src/pose_controller.cpp:145-459   (horizontal steps in two leg groups or
        directly, vertical body raise, the transition poses a first START_UP learns within the joint-limit safety factor, replay)
  Model::legsBearingLoad                                    src/model.cpp:78-88
  LegPoser::resetStepToPosition / transition pose list      include/.../pose_controller.h:519-539
  Leg::applyIK's return value (limit proximity, 0 on an IK deviation)   src/model.cpp:799-857, :905-929
with LegPoser::stepToPosition from make_sequence_golden.py and solveIK / updateJointPositions from make_walk_golden.py.
DATA: the joint configuration the robot reports when the node starts (here default.yaml's `unpacked` positions: the READY
estimate), the stance positions, body clearance, swing height, step frequency.  The loop around each call (the posing part of
StateController::loop) leaves Model::current_pose_ = the walk-plane pose (0, 0, body_clearance) for a robot that has not walked.

tests/test_oracle_golden.py::test_startup_sequence_trajectories replays the calls on the oracle: START_UP (learning), SHUT_DOWN,
START_UP (replay) - return values exactly, joints to 1e-6 rad.
"""
import importlib.util
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


mw = _load("make_walk_golden")        # dh, MODEL, solve_ik, update_joints, fk_tip
ms = _load("make_sequence_golden")    # StepToPosition

START_UP, SHUT_DOWN = 0, 1
SAFETY_FACTOR, HALF_BODY_DEPTH = 0.15, 0.05
HORIZONTAL_TRANSITION_TIME, VERTICAL_TRANSITION_TIME = 1.0, 3.0
IK_TOLERANCE = 0.005


class SeqLeg:
    def __init__(self, index, q, stance_xy):
        self.i = index
        self.q, self.qd = np.array(q, dtype=float), np.zeros(len(q))
        self.default = np.array([stance_xy[0], stance_xy[1], 0.0])      # LegStepper::default_tip_pose_ of a robot that has not walked
        self.poses = []                                                   # LegPoser::transition_poses_ (positions)
        self.target = None                                                # LegPoser::target_tip_pose_.position_
        self.completed = False
        self.stp = None                                                   # the stepToPosition state
        self.poser_tip = self.tip()

    def tip(self):                                                        # Leg::current_tip_pose_.position_
        return mw.fk_tip(self.i, self.q)

    def tip_quat(self):                                                   # ... and its rotation (w, x, y, z): stepToPosition's origin
        from scipy.spatial.transform import Rotation as R
        t = mw.dh(*mw.MODEL.base[self.i])
        for k, (d, th, r, al) in enumerate(mw.MODEL.links[self.i]):
            t = t @ mw.dh(d, th + self.q[k], r, al)
        x = R.from_matrix(t[:3, :3]).as_quat()
        return [x[3], x[0], x[1], x[2]]

    def step_to_position(self, target, lift, time_to_step):               # LegPoser::stepToPosition(target, Identity, lift, time)
        if self.stp is None or self.stp.first:
            self.stp = ms.StepToPosition(self.tip(), self.tip_quat())
        progress, tip, _ = self.stp.step(target, None, [0, 0, 0], [1, 0, 0, 0], lift, time_to_step)
        self.poser_tip = tip                                              # LegPoser::current_tip_pose_
        return progress

    def reset_step(self):                                                 # resetStepToPosition
        if self.stp is not None:
            self.stp.first = True
        return 100

    def apply_ik(self, desired, dt):                                      # setDesiredTipPose(desired) + applyIK(): returns the limit proximity
        base = mw.dh(*mw.MODEL.base[self.i])
        bi = np.linalg.inv(base)
        delta = np.zeros(6)
        delta[:3] = (bi @ np.append(desired, 1))[:3] - (bi @ np.append(self.tip(), 1))[:3]
        dq = mw.solve_ik(self.i, self.q, self.qd, delta, False)
        self.q, self.qd, success = mw.update_joints(self.i, self.q, dq, dt, False)
        if (np.abs(self.tip() - desired) > IK_TOLERANCE).any():
            success = 0.0
        return success


class Sequencer:
    """PoseController's sequence members (pose_controller.h:273-304) over the legs."""

    def __init__(self, legs, P):
        self.legs, self.P = legs, P
        self.L = len(legs)
        self.legs_completed_step = self.current_group = self.transition_step = self.transition_step_count = 0
        self.set_target, self.proximity_alert = True, False
        self.horizontal_complete = self.vertical_complete = False
        self.first_execution, self.reset_sequence = True, True
        self.alerts = 0

    def target_for(self, leg, next_step):
        if len(leg.poses) > next_step:
            return leg.poses[next_step].copy()
        cp = np.array([0.0, 0.0, self.P["body_clearance"]])               # Model::current_pose_.inverseTransformVector(default tip)
        return leg.default - cp

    def execute(self, sequence):
        P, dt = self.P, self.P["time_delta"]
        if self.reset_sequence and sequence == START_UP:
            self.reset_sequence, self.first_execution, self.transition_step = False, True, 0
            for leg in self.legs:
                leg.poses = [leg.tip()]
        progress = 0
        ts = self.transition_step
        if sequence == START_UP:
            horizontal, vertical = ts % 2 == 0, ts % 2 == 1
            nxt, step_target = ts + 1, self.transition_step_count
            total = ts * 100 // max(self.transition_step_count, 1)
        else:
            horizontal, vertical = ts % 2 == 1, ts % 2 == 0
            nxt, step_target = ts - 1, 0
            total = 100 - ts * 100 // max(self.transition_step_count, 1)
        final = (self.horizontal_complete or self.vertical_complete) if self.first_execution else (nxt == step_target)
        complete = False
        safety = SAFETY_FACTOR / (ts + 1) if self.first_execution else 0.0
        normalised = 0
        if horizontal:
            if self.set_target:
                self.set_target = False
                for leg in self.legs:
                    leg.completed = False
                    t = self.target_for(leg, nxt)
                    t[2] = leg.tip()[2]                                   # maintain height
                    leg.target = t
            bearing = -(sum(leg.tip()[2] for leg in self.legs) / self.L) > HALF_BODY_DEPTH     # Model::legsBearingLoad
            direct = not bearing
            for leg in self.legs:
                if leg.completed:
                    continue
                if leg.i % 2 == self.current_group or direct:
                    lift = 0.0 if direct else P["swing_height"]
                    time_to_step = HORIZONTAL_TRANSITION_TIME / P["step_frequency"] * (2.0 if self.first_execution else 1.0)
                    progress = leg.step_to_position(leg.target, lift, time_to_step)
                    proximity = leg.apply_ik(leg.poser_tip, dt)
                    exceeded = proximity < safety
                    if self.first_execution and exceeded:
                        leg.target = leg.poser_tip.copy()
                        progress = leg.reset_step()
                        self.proximity_alert = True
                        self.alerts += 1
                    if progress == 100:
                        leg.completed = True
                        self.legs_completed_step += 1
                        if self.first_execution:
                            leg.poses.append((leg.target if not exceeded else leg.poser_tip).copy())
                else:
                    self.legs_completed_step += 1
                    leg.completed = True
            if direct:
                normalised = progress // max(self.transition_step_count, 1)
            else:
                normalised = (progress // 2 + (0 if self.current_group == 0 else 50)) // max(self.transition_step_count, 1)
            if self.legs_completed_step == self.L:
                self.set_target = True
                self.legs_completed_step = 0
                if self.current_group == 1 or direct:
                    self.current_group = 0
                    self.transition_step = nxt
                    self.horizontal_complete = not self.proximity_alert
                    complete = final
                    self.proximity_alert = False
                elif self.current_group == 0:
                    self.current_group = 1
        if vertical:
            if self.set_target:
                self.set_target = False
                for leg in self.legs:
                    t = self.target_for(leg, nxt)
                    t[0], t[1] = leg.tip()[0], leg.tip()[1]               # maintain horizontal position
                    leg.target = t
            within = True
            for leg in self.legs:
                time_to_step = VERTICAL_TRANSITION_TIME / P["step_frequency"] * (2.0 if self.first_execution else 1.0)
                progress = leg.step_to_position(leg.target, 0.0, time_to_step)
                proximity = leg.apply_ik(leg.poser_tip, dt)
                within = within and not (proximity < safety)
            if not within and self.first_execution:
                self.alerts += 1
            if (not within and self.first_execution) or progress == 100:
                for leg in self.legs:
                    progress = leg.reset_step()
                    if self.first_execution:
                        leg.poses.append((leg.target if within else leg.poser_tip).copy())
                self.vertical_complete = within
                self.transition_step = nxt
                complete = final
                self.set_target = True
            normalised = progress // max(self.transition_step_count, 1)
        if self.first_execution:
            self.transition_step_count = self.transition_step
        assert self.transition_step <= 20                                  # TRANSITION_STEP_THRESHOLD
        if complete:
            self.set_target = True
            self.vertical_complete = self.horizontal_complete = False
            self.first_execution = False
            return 100
        total = min(total + normalised, 99)
        return -1 if self.first_execution else total


def run(offset=None, octopod=False):
    from syropod_highlevel_controller_amd import default_hexapod_params
    if octopod:                 # BASELINE.json config 4's synthetic 8 x 5 octopod: redundant chains through the same choreography
        p = mw.make_params("ripple", "8x5")
        mw.MODEL = mw.Morphology.from_params(p)
    else:
        p = default_hexapod_params("tripod")
        mw.MODEL = mw.Morphology.default_hexapod()
    L, D = p.leg_count, p.leg_dof[0]
    P = dict(time_delta=p.time_delta, body_clearance=p.body_clearance, swing_height=p.swing_height, step_frequency=p.step_frequency)
    q0 = np.array([[p.joint[l][j].unpacked for j in range(D)] for l in range(L)])
    if offset is not None:      # a robot switched on in some other configuration: the first START_UP has to feel its way
        q0 = q0 + offset
    legs = [SeqLeg(l, q0[l], (p.stance_position[l][0], p.stance_position[l][1])) for l in range(L)]
    seq = Sequencer(legs, P)
    out = {"q0": q0}
    for name, which in (("startup_first", START_UP), ("shutdown", SHUT_DOWN), ("startup_replay", START_UP)):
        rows = []
        for _ in range(6000):
            r = seq.execute(which)
            rows.append([r, *np.concatenate([leg.q for leg in legs])])
            if r == 100:
                break
        assert rows[-1][0] == 100, name
        out[name] = np.array(rows)
    out["transition_steps"] = np.array([seq.transition_step_count])
    out["proximity_alerts"] = np.array([seq.alerts])
    return out


def run_new_stance():
    """PoseController::stepToNewStance (src/pose_controller.cpp:521-557) on a robot that stands after its direct start-up: the two
    leg groups step (with the swing height) onto their default tip poses one after the other, the body pose of stepToPosition
    easing from the identity to Model::current_pose_."""
    P = mw.hexapod("tripod", manual_posing=1)
    w = mw.started_walker(P, "tripod")           # joints: the numpy init chain's direct start-up + the first loop (nothing from oracle/ or the product)
    q0, qd0 = w.q.copy(), w.qd.copy()
    legs_completed, group, rows = 0, 0, []
    stp = [None] * 6
    for _ in range(2000):
        pose, adm = w.prologue()
        progress = 0
        for i, leg in enumerate(w.legs):
            if i % 2 != group:
                continue
            if stp[i] is None or stp[i].first:
                sl = SeqLeg(i, w.q[i], (0.0, 0.0))
                stp[i] = ms.StepToPosition(sl.tip(), sl.tip_quat())
            q = pose.r.as_quat()
            progress, tip, _ = stp[i].step(leg.default, None, pose.p, [q[3], q[0], q[1], q[2]], P["swing_height"], 1.0 / P["step_frequency"], adm[i])
            w.q[i], w.qd[i] = mw.apply_ik(i, w.q[i], w.qd[i], tip + adm[i], w.dt)
            legs_completed += int(progress == 100)
        progress = progress // 2 + group * 50
        group = legs_completed // 3
        if legs_completed == 6:
            legs_completed, group = 0, 0
        rows.append([progress, *w.q.ravel()])
        if len(rows) > 1 and progress == 100:
            break
    return {"new_stance/rows": np.array(rows), "new_stance/joint_start": np.stack([q0.reshape(6, 3), qd0.reshape(6, 3)])}


if __name__ == "__main__":
    out = run()
    out.update(run_new_stance())
    rng = np.random.default_rng(77)
    off = rng.uniform(-0.25, 0.25, (6, 3))
    for k, v in run(off).items():
        out["offset/" + k] = v
    for k, v in run(octopod=True).items():
        out["8x5/" + k] = v
    np.savez_compressed(os.path.join(HERE, "startup_golden.npz"), **out)
    print({k: v.shape for k, v in out.items() if v.ndim == 2}, "transition steps", int(out["transition_steps"][0]), int(out["offset/transition_steps"][0]),
          "workspace alerts", int(out["proximity_alerts"][0]), int(out["offset/proximity_alerts"][0]))

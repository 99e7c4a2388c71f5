"""Generates tests/golden/manual_golden.npz: manual leg manipulation from an INDEPENDENT numpy restatement (this file + the walker /
kinematic model of make_walk_golden.py and LegPoser::stepToPosition of make_sequence_golden.py) - no code shared with oracle/ or
the engine.

    python tests/golden/make_manual_golden.py

Restated here, from the reference sources only (OpenSHC v0.5.11, paths relative to /root/reference):
  StateController::legStateToggle            src/state_controller.cpp:541-646   (WALKING -> WALKING_TO_MANUAL -> MANUAL -> MANUAL_TO_WALKING
                                                                                  -> WALKING, MAX_MANUAL_LEGS, stop the robot first)
  PoseController::poseForLegManipulation     src/pose_controller.cpp:561-611
  LegPoser::stepToPosition / Leg::setDesiredTipPose: no admittance delta for manually manipulated legs   :1609-1614, src/model.cpp:655-656
  WalkController::updateManual x 2, updateWalk's "all legs WALKING" gate, PoseController::updateStance for manual legs
                                             (in make_walk_golden.py: src/walk_controller.cpp:491-505, :652-744, src/pose_controller.cpp:134-137)
One scenario per leg_manipulation_mode: a default hexapod with admittance control and a steady 4 N on every tip walks, is asked to
hand leg 2 over (stops, poses, MANUAL), the leg follows tip-velocity and tip-position inputs while body-velocity commands are ignored,
a second leg follows, a third is refused, both come back, the robot walks again.  Recorded per loop: its kind, the request's result,
joints.  joint_control (keys jc_*): the velocity inputs step the coxa / tibia joints, the stepper holds the FK tip pose with its
rotation, every applyIK of the leg is rotation-constrained (walk_controller.cpp:677-690, model.cpp:880-900); the position inputs are
ignored; inputs withdrawn for a while.  IMU + inclination posing (keys imu_*): the tip_control scenario with the body pose moving under
the standing robot (IMU PID, the inclination translation poseForLegManipulation adds for the lifted leg, :572), a new IMU reading
every 20 loops (columns 17-23 of a loop row: orientation wxyz, angular velocity).

tests/test_oracle_golden.py::test_manual_leg_trajectories replays the loops on the oracle: results exactly, joints to 1e-6 rad
wherever the robot walks and 5e-3 while it stands (where the reference's IK step amplifies rounding differences).
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


mw = _load("make_walk_golden")
ms = _load("make_sequence_golden")
WALKING, MANUAL, W2M, M2W = 0, 1, -1, -2
MAX_MANUAL_LEGS = 2


class Toggler:
    def __init__(self, w):
        self.w = w
        self.manual_leg_count = 0
        self.stp = [None] * w.L

    def tip_quat(self, i):
        from scipy.spatial.transform import Rotation as R
        t = mw.dh(*mw.MODEL.base[i])
        for k, (d, th, r, al) in enumerate(mw.MODEL.links[i]):
            t = t @ mw.dh(d, th + self.w.q[i][k], r, al)
        x = R.from_matrix(t[:3, :3]).as_quat()
        return [x[3], x[0], x[1], x[2]]

    def pose_for_leg_manipulation(self, pose, adm):
        w, P = self.w, self.w.P
        min_progress = 2147483647
        for i, leg in enumerate(w.legs):
            step_height, step_time = P["swing_height"], 1.0 / P["step_frequency"]
            if leg.leg_state == W2M:
                tp, tr = w.inclination.p + np.array([0.0, 0.0, -step_height]), None   # Identity + inclination_pose_.position_, lowered by the step height
                target = leg.default - tp
            else:
                tp = pose.p - w.manual_pose.p                                   # current pose, manual pose removed, default pose (identity) added
                target = pose.r.inv().apply(leg.default - tp)
            if leg.leg_state == W2M:
                leg.tip, leg.rot_defined = target.copy(), False                 # setCurrentTipPose(target_tip_pose): rotation undefined
                step_height = 0.0
            elif leg.leg_state == M2W:
                leg.tip, leg.rot_defined = leg.default.copy(), False
            if self.stp[i] is None or self.stp[i].first:
                self.stp[i] = ms.StepToPosition(mw.fk_tip(i, w.q[i]), self.tip_quat(i))
            manually = leg.leg_state in (MANUAL, W2M)
            progress, tip, _ = self.stp[i].step(target, None, [0, 0, 0], [1, 0, 0, 0], step_height, step_time, None if manually else adm[i])
            if leg.leg_state != MANUAL or (progress == 100 and self.stp[i].count == 0):   # (:1680-1684; the "nothing to do" return sets it regardless, :1604)
                leg.poser_tip = tip
            min_progress = min(min_progress, progress)
            if progress != 100:
                desired = leg.poser_tip + (np.zeros(3) if manually else adm[i])
                leg.desired_tip = desired
                # a MANUAL leg's LegPoser holds the stepper's tip pose as updateStance left it - with the FK rotation joint_control gave it
                ddir = leg.cur_dir if (leg.leg_state == MANUAL and leg.rot_defined) else None
                w.q[i], w.qd[i] = mw.apply_ik(i, w.q[i], w.qd[i], desired, w.dt, ddir)
                leg.model_tip = mw.fk_tip(i, w.q[i])
        return min_progress

    def loop(self, leg_id, lin, ang):
        """One StateController::loop with a toggle request for leg_id pending.  Returns (result, lin, ang): -1 while the robot is
        still walking (velocity inputs zeroed, ordinary cycle), 0 in progress, 1 done, 2 refused."""
        w = self.w
        if w.walk_state != mw.STOPPED:
            lin, ang = (0.0, 0.0), 0.0
            w.cycle(lin, ang)
            return -1, lin, ang
        pose, adm = w.prologue()
        leg = w.legs[leg_id]
        if leg.leg_state == WALKING:
            if self.manual_leg_count < MAX_MANUAL_LEGS:
                leg.leg_state = W2M
                leg.swing_progress = leg.stance_progress = -1.0
                return 0, lin, ang
            return 2, lin, ang
        if leg.leg_state == MANUAL:
            leg.leg_state = M2W
            return 0, lin, ang
        to_manual = leg.leg_state == W2M
        w.reset_mode = 5                                                        # IMMEDIATE_ALL_RESET: the next pose update resets the manual pose
        progress = self.pose_for_leg_manipulation(pose, adm)
        if progress == 100:
            leg.leg_state = MANUAL if to_manual else WALKING
            w.reset_mode = 0
            self.manual_leg_count += 1 if to_manual else -1
            return 1, lin, ang
        return 0, lin, ang


def run(mode="tip_control", posing=False, octopod=False, auto=False):
    import zlib
    gait = "ripple" if octopod else "tripod"
    if octopod:                                  # the synthetic 8 x 5 octopod with gravity-aligned tips: the frozen walker's legs keep their
        P = mw.hexapod(gait, "8x5", gravity_aligned_tips=1, leg_manipulation_mode=mode,      # rotation-constrained IK, the toggled leg loses its rotation
                       max_translation_velocity=mw.make_params(gait, "8x5").max_translation_velocity)
    else:
        P = mw.hexapod(gait, admittance_control=1, manual_posing=1, leg_manipulation_mode=mode)
    if posing:                                   # the body pose keeps moving while the robot stands: IMU PID + the inclination translation
        P.update(imu_posing=1, inclination_posing=1)
    if auto:                                     # cyclic auto posing: the posers' latches and the per-leg negation follow the (frozen) step phases
        P.update(auto_posing=1, n_auto_posers=len(P["pose_phase_starts"]))
    w = mw.started_walker(P, gait, "8x5" if octopod else None)   # joints: the numpy init chain's direct start-up + the first loop (nothing from oracle/ or the product)
    q0, qd0 = w.q.copy(), w.qd.copy()
    if not octopod:
        w.tip_force = np.tile(np.array([0.0, 0.0, 4.0]), (6, 1))
    t = Toggler(w)
    rng = np.random.default_rng(zlib.crc32(b"manual"))
    loops = []       # (kind, leg, result, lin x, lin y, ang, then the manual inputs in force: primary leg, velocity (3), position (3), secondary leg, velocity (3))
    joints = []
    lin, ang = (0.4, 0.15), 0.2
    inputs = dict(primary=-1, pv=np.zeros(3), pp=np.zeros(3), secondary=-1, sv=np.zeros(3))

    imu = dict(q=np.array([1.0, 0, 0, 0]), gyro=np.zeros(3), loops=0)

    def record(kind, leg, result):
        loops.append([kind, leg, result, lin[0], lin[1], ang, inputs["primary"], *inputs["pv"], *inputs["pp"], inputs["secondary"], *inputs["sv"],
                      *imu["q"], *imu["gyro"]])
        joints.append(w.q.copy())

    def imu_sample():                              # a new IMU reading every 20 loops: the slope changes under the standing robot
        if posing and imu["loops"] % 20 == 0:
            e = [rng.uniform(-0.12, 0.12), rng.uniform(-0.12, 0.12), 0.0]
            w.imu_q, w.gyro = mw.euler_to_rot(e), rng.normal(0, 0.03, 3)
            x = w.imu_q.as_quat()
            imu["q"], imu["gyro"] = np.array([x[3], x[0], x[1], x[2]]), w.gyro.copy()
        imu["loops"] += 1

    def cycles(k):
        for _ in range(k):
            imu_sample()
            w.cycle(lin, ang)
            record(0, -1, 0)

    def toggle(leg_id):
        nonlocal lin, ang
        for _ in range(4000):
            imu_sample()
            result, lin, ang = t.loop(leg_id, lin, ang)
            record(1, leg_id, result)
            if result in (1, 2):
                return result
        raise AssertionError("toggle did not finish")

    def set_inputs(primary=-1, pv=None, pp_=None, secondary=-1, sv=None):
        inputs.update(primary=primary, pv=np.zeros(3) if pv is None else np.array(pv, float), pp=np.zeros(3) if pp_ is None else np.array(pp_, float),
                      secondary=secondary, sv=np.zeros(3) if sv is None else np.array(sv, float))
        w.primary_leg, w.primary_velocity, w.primary_position = inputs["primary"], inputs["pv"], inputs["pp"]
        w.secondary_leg, w.secondary_velocity, w.secondary_position = inputs["secondary"], inputs["sv"], np.zeros(3)

    cycles(70)
    assert toggle(2) == 1
    lin, ang = (0.3, 0.0), 0.1                    # ignored while a leg is MANUAL
    set_inputs(2, pv=rng.uniform(-1, 1, 3))
    cycles(25)
    set_inputs(2, pp_=[P["stance_position"][2][0] * 0.9, P["stance_position"][2][1] * 0.9, -0.06])
    cycles(20)
    set_inputs(2, pv=rng.uniform(-1, 1, 3))
    cycles(15)
    if mode == "joint_control":                   # inputs withdrawn: the stepper keeps the last FK tip pose, rotation included
        set_inputs(2)
        cycles(20)
        set_inputs(2, pv=[0.0, 0.0, 0.7])         # a z input moves no joint but still hands the stepper the FK tip pose
        cycles(10)
    assert toggle(4) == 1
    set_inputs(2, pv=rng.uniform(-1, 1, 3), secondary=4, sv=rng.uniform(-1, 1, 3))
    cycles(25)
    assert toggle(0) == 2                         # MAX_MANUAL_LEGS
    set_inputs()
    assert toggle(4) == 1
    assert toggle(2) == 1
    lin, ang = (0.35, -0.1), -0.15
    cycles(120)
    assert w.walk_state != mw.STOPPED
    return {"loops": np.array(loops), "joints": np.array(joints), "joint_start": np.stack([q0, qd0])}


if __name__ == "__main__":
    out = run()
    jc = run("joint_control")
    out.update({"jc_" + k: v for k, v in jc.items()})
    out.update({"imu_" + k: v for k, v in run(posing=True).items()})
    out.update({"g85_" + k: v for k, v in run(octopod=True).items()})
    out.update({"auto_" + k: v for k, v in run(auto=True).items()})
    np.savez_compressed(os.path.join(HERE, "manual_golden.npz"), **out)
    for pre in ("", "jc_", "imu_", "g85_", "auto_"):
        k = out[pre + "loops"]
        print(pre or "tip_control", "loops", len(k), "toggle loops", int((k[:, 0] == 1).sum()), "results", sorted(set(k[k[:, 0] == 1][:, 2].astype(int).tolist())))

"""Replays of the committed golden fixtures (tests/golden/*.npz: independent numpy restatements of the reference, see the generators
there) on anything with the batch interface - the CPU oracle (tests/test_oracle_golden.py) and the HIP engine (tests/test_gpu_golden.py).
A backend is a function params -> (batch object of ONE instance, step()): OracleBatch and BatchEngine share every other call."""
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_hexapod_params(gait="tripod"):
    """default.yaml with the start-up length the generators use (make_walk_golden.py START_UP_TIME: 100 start-up steps)."""
    from syropod_highlevel_controller_amd import default_hexapod_params
    p = default_hexapod_params(gait)
    p.time_to_start = 2.0
    return p


def oracle_backend(p):
    from oracle_lib import OracleBatch
    ob = OracleBatch(p, 1)
    return ob, lambda: ob.step(1, 1)


def engine_backend(p):
    from syropod_highlevel_controller_amd.engine import BatchEngine
    eng = BatchEngine(p, 1)
    return eng, lambda: eng.step(1)


def replay_manual(backend, mode, start_tol=1e-12):
    """tests/golden/make_manual_golden.py: request results exactly; joints to 1e-6 rad while the robot walks, 5e-3 once it stands
    (free-running: the reference's IK step amplifies rounding differences on a standing robot)."""
    g = np.load(os.path.join(HERE, "manual_golden.npz"))
    posing = mode == "imu_and_inclination_posing"   # the body pose moves under the standing robot: the posing part of every loop, toggle loops included
    octopod = mode == "8x5_gravity_aligned_tips"   # the frozen walker's legs keep their rotation-constrained IK, the toggled leg loses its rotation
    if mode == "joint_control":
        g = {k[3:]: g[k] for k in g.files if k.startswith("jc_")}
    elif posing:
        g = {k[4:]: g[k] for k in g.files if k.startswith("imu_")}
    elif octopod:
        g = {k[4:]: g[k] for k in g.files if k.startswith("g85_")}
    elif mode == "auto_posing":
        g = {k[5:]: g[k] for k in g.files if k.startswith("auto_")}
    if octopod:
        from syropod_highlevel_controller_amd import synthetic_octopod_params
        p = synthetic_octopod_params("ripple", 5, 8)
        p.gravity_aligned_tips = 1
        p.time_to_start = 2.0
        start_tol = max(start_tol, 1e-8)         # (the redundant chain's start-up configuration: DESIGN.md section 2)
    else:
        p = golden_hexapod_params("tripod")
        p.admittance_control = 1
    L, D = p.leg_count, p.leg_dof[0]
    p.leg_manipulation_mode = 1 if mode == "joint_control" else 0
    if mode == "auto_posing":
        p.auto_posing = 1
    if posing:
        p.imu_posing, p.inclination_posing = 1, 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    ob, step = backend(p)
    assert np.abs(np.stack([x[0] for x in ob.joints()]).reshape(2, L, D) - g["joint_start"]).max() < start_tol
    if not octopod:
        ob.set_tip_force(np.tile(np.array([0.0, 0.0, 4.0]), (1, 6, 1)))
    worst_walk = worst_stand = 0.0
    stood = False
    for k, row in enumerate(g["loops"]):
        kind, leg, result = int(row[0]), int(row[1]), int(row[2])
        ob.set_velocity(row[3:5][None], row[5:6])
        prim, sec = int(row[6]), int(row[13])
        ob.set_manual_inputs(np.array([prim], dtype=np.int32), row[7:10][None], row[10:13][None], np.array([sec], dtype=np.int32), row[14:17][None], None)
        if posing:
            ob.set_imu(row[17:21][None], row[21:24][None])
        if kind == 0:
            step()
        else:
            assert int(ob.toggle_leg_state(np.array([leg], dtype=np.int32))[0]) == result, (k, leg, result)
        d = np.abs(ob.joints()[0][0].reshape(L, D) - g["joints"][k]).max()
        stood = stood or ob.body_state()[2][0] == 3
        if stood:
            worst_stand = max(worst_stand, d)
        else:
            worst_walk = max(worst_walk, d)
        assert worst_walk < 1e-6 and worst_stand < 5e-3, (k, kind, worst_walk, worst_stand)
    assert ob.body_state()[2][0] != 3 and (ob.leg_manipulation_state() == 0).all()
    return f"manual legs ({mode}): {len(g['loops'])} loops, max |joint diff| {worst_walk:.2e} rad walking, {worst_stand:.2e} rad after the first stop"


def replay_planner(backend, posing, start_tol=1e-12):
    """tests/golden/make_planner_golden.py: executePlan's result and plan_step_ exactly; joints free-running."""
    from syropod_highlevel_controller_amd.params import ExternalTarget
    g = np.load(os.path.join(HERE, "planner_golden.npz"))
    imu = posing == "imu_and_inclination_posing"
    octopod = posing == "8x5_gravity_aligned_tips"   # transitionStance turns every tip towards gravity; legs without a tip target stay and turn
    if imu:
        g = {k[4:]: g[k] for k in g.files if k.startswith("imu_")}
    elif octopod:
        g = {k[4:]: g[k] for k in g.files if k.startswith("g85_")}
    elif posing == "auto_posing":
        g = {k[5:]: g[k] for k in g.files if k.startswith("auto_")}
    events_file = ("planner_golden_events_imu.json" if imu else "planner_golden_events_8x5.json" if octopod else
                   "planner_golden_events_auto.json" if posing == "auto_posing" else "planner_golden_events.json")
    events = {int(e[0]): e for e in json.load(open(os.path.join(HERE, events_file)))}
    if octopod:
        from syropod_highlevel_controller_amd import synthetic_octopod_params
        p = synthetic_octopod_params("ripple", 5, 8)
        p.gravity_aligned_tips = 1
        p.time_to_start = 2.0
        start_tol = max(start_tol, 1e-8)         # (the redundant chain's start-up configuration: DESIGN.md section 2)
    else:
        p = golden_hexapod_params("tripod")
        p.admittance_control = 1
    L, D = p.leg_count, p.leg_dof[0]
    if posing == "auto_posing":
        p.auto_posing = 1
    if imu:
        p.imu_posing, p.inclination_posing = 1, 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    ob, step = backend(p)
    assert np.abs(np.stack([x[0] for x in ob.joints()]).reshape(2, L, D) - g["joint_start"]).max() < start_tol
    if octopod:
        ob.set_velocity(np.array([[0.4, -0.2]]), np.array([0.2]))
    else:
        ob.set_tip_force(np.tile(np.array([0.0, 0.0, 4.0]), (1, 6, 1)))
        ob.set_velocity(np.array([[0.45, -0.1]]), np.array([0.15]))
    worst_walk = worst_stand = 0.0
    planner_on = seen_crawl = stance_running = False
    for k, row in enumerate(g["rows"]):
        if imu:
            ob.set_imu(row[3:7][None], row[7:10][None])
        if k in events:
            _, kind, data = events[k]
            if kind == "configuration":
                cfg = np.full((L, D), np.nan)
                for leg, q in data.items():
                    cfg[int(leg)] = q
                ob.set_target_configuration(cfg[None])
            else:
                rows = (ExternalTarget * L)()
                tr = np.tile(np.array([0, 0, 0, 1.0, 0, 0, 0]), (L, 1))
                for leg, t in data["targets"].items():
                    r = rows[int(leg)]
                    r.defined, r.swing_clearance = 1, t["clearance"]
                    r.pose[0:3] = t["pose_p"]
                    r.transform[:] = [0, 0, 0, 1, 0, 0, 0]
                    tr[int(leg)] = [*t["transform"][0], *t["transform"][1]]
                if data["targets"]:
                    assert ob.set_external_target(rows) == 0       # the robot stands: the LegPosers take the targets
                    ob.set_external_transform(tr[None], which=2)   # generateExternalTargetTransforms
                ob.set_target_body_pose(np.array([[*data["body"][0], *data["body"][1]]]))
        if int(row[0]) == 0:
            step()
        else:
            if not planner_on:
                ob.set_planner_mode(True)
                planner_on = True
            pr, st = ob.execute_plan()
            assert (int(pr[0]), int(st[0])) == (int(row[1]), int(row[2])), (k, pr, st, row)
        d = np.abs(ob.joints()[0][0].reshape(L, D) - g["joints"][k]).max()
        # The two free-running chains agree to 1e-8 through walking, stopping, the waits and the whole configuration step, and through
        # a stance step until its last few percent: there the tips all but stand still, the regime in which the reference's IK step
        # amplifies rounding differences by an order of magnitude per loop.  From then on: same place (5 mm).
        seen_crawl = seen_crawl or (stance_running and int(row[0]) == 1 and 95 <= int(row[1]) <= 100)
        stance_running = (stance_running or (k in events and events[k][1] == "stance")) and not (int(row[0]) == 1 and int(row[1]) == 100)
        if seen_crawl:
            worst_stand = max(worst_stand, d)
        else:
            worst_walk = max(worst_walk, d)
        assert worst_walk < 1e-8 and worst_stand < 5e-3, (k, worst_walk, worst_stand)
    return (f"planner ({posing}): {len(g['rows'])} loops, final plan step {int(g['rows'][-1, 2])}, max |joint diff| {worst_walk:.2e} rad up to the crawl at the "
            f"end of the first stance step, {worst_stand:.2e} rad after")


def walk_meta():
    return json.load(open(os.path.join(HERE, "walk_golden_meta.json")))


def replay_walk(name, meta, tol_x=1e-9):
    """tests/golden/make_walk_golden.py on the HIP engine, free-running from the engine's own start-up: walk state, step states and phases
    exactly; walker tips, body pose and velocities to tol_x; the joints of the scenarios that carry the kinematic model to the
    north-star bar, 1e-6 rad.  (The oracle's replay of the same fixtures is tests/test_oracle_golden.py::test_walk_trajectories.)"""
    from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
    from syropod_highlevel_controller_amd.engine import BatchEngine
    from syropod_highlevel_controller_amd.params import VEL_REAL, ExternalTarget
    data = np.load(os.path.join(HERE, "walk_golden.npz"))
    g = {k.split("/", 1)[1]: data[k] for k in data.files if k.startswith(name + "/")}
    p = default_hexapod_params(meta["gait"])
    if meta["overrides"].get("morphology") == "8x5":
        p = synthetic_octopod_params(meta["gait"], 5, 8)
    if meta["overrides"].get("morphology") == "mixed":   # legs of 3 / 5 / 4 / 3 / 5 / 4 joints: the engine pads the shorter legs, the fixture ran each leg's own chain
        from syropod_highlevel_controller_amd import synthetic_mixed_dof_params
        p = synthetic_mixed_dof_params(meta["gait"])
    LD = (p.leg_count, max(p.leg_dof[l] for l in range(p.leg_count)))
    for k, v in meta["overrides"].items():
        if k == "velocity_input_mode":
            p.velocity_input_mode = VEL_REAL if v == "real" else 0
        elif k in ("n_auto_posers", "model", "morphology", "contacts", "efforts", "pose_inputs", "gait_change", "adjust"):
            pass
        else:
            setattr(p, k, v)
    if p.imu_posing:
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    p.time_to_start = meta["time_to_start"]
    eng = BatchEngine(p, 1)
    t = eng.tables()
    for k, table in meta["limits"].items():   # the product's host init chain against the numpy init chain's limit tables
        np.testing.assert_allclose(list(getattr(t, k)), table, rtol=1e-8)
    start_diff = None
    if "joint_start" in g:
        start_diff = float(np.abs(np.stack([x[0] for x in eng.joints()]).reshape(2, *LD) - g["joint_start"]).max())
        assert start_diff < (1e-8 if LD[1] > 3 else 1e-11), (name, start_diff)
    worst_tip = worst_pose = worst_q = 0.0
    for c in range(meta["cycles"]):
        for ec, kind, leg, v in meta.get("events", []):   # rough-terrain scenarios: TargetTipPose / tf refresh / tip-state messages
            if ec != c:
                continue
            which = 0 if kind.endswith("target") else 1
            if kind in ("target", "default", "withdraw_default"):
                rows = (ExternalTarget * 1)()
                if kind != "withdraw_default":
                    rows[0].defined, rows[0].swing_clearance, rows[0].frame_is_odom_ideal = 1, v[7], int(v[8])
                    rows[0].pose[:] = v[:7]
                    rows[0].transform[:] = [0, 0, 0, 1, 0, 0, 0]
                eng.set_external_target(rows, which=which, leg=leg)
            elif kind.startswith("transform_"):
                eng.set_external_transform(np.array(v, dtype=np.float64)[None], which=which, leg=leg)
            elif kind == "zero_tip_force":
                eng.set_tip_force(np.zeros((1, p.leg_count, 3)))
            elif kind == "pose_input":
                eng.set_pose_input(np.array(v[:3])[None], np.array(v[3:])[None])
            elif kind == "pose_reset_mode":
                eng.set_pose_reset_mode(np.array([int(v[0])], dtype=np.int32))
        if "gait_request" in g and g["gait_request"][c]:   # gait_change_flag_ set: changeGait runs every loop until the robot has stopped
            eng.change_gait(default_hexapod_params(meta["overrides"]["gait_change"]))
        eng.set_velocity(np.array(g["lin"][c], dtype=np.float64)[None], np.array([g["ang"][c]], dtype=np.float64))
        if "adjust_request" in g and g["adjust_request"][c]:   # parameter_adjust_flag_ set: adjustParameter runs in every loop until the value is set (its decision reads
            waiting = eng.adjust_parameter(int(g["adjust_request"][c]), float(g["adjust_value"][c]))   # this loop's velocity input); the fixture asks again exactly while it waits
            last = c + 1 == meta["cycles"] or g["adjust_request"][c + 1] != g["adjust_request"][c] or g["adjust_value"][c + 1] != g["adjust_value"][c]
            assert (waiting == 0) == bool(last), (name, c, waiting)
        if p.imu_posing or p.inclination_posing:
            eng.set_imu(np.array(g["imu_q"][c])[None], np.array(g["gyro"][c])[None])
        if p.admittance_control and not p.use_joint_effort:
            eng.set_tip_force(np.array(g["force"][c])[None])
        if "effort" in g:
            eng.set_joint_effort(np.array(g["effort"][c]).reshape(1, -1))
        if "contact_force" in g:
            eng.set_tip_force(np.array(g["contact_force"][c])[None])
        eng.step(1)
        ls = eng.leg_state()
        pose, vel, ws = eng.body_state()
        assert ws[0] == g["walk_state"][c], (name, c)
        assert np.array_equal(ls["leg_status"][0] & 3, g["state"][c]), (name, c)
        assert np.array_equal(ls["leg_status"][0] >> 8, g["phase"][c]), (name, c)
        np.testing.assert_allclose(vel[0], g["velocity"][c], atol=tol_x, err_msg=f"{name} cycle {c}")
        worst_tip = max(worst_tip, np.abs(ls["walker_tip"][0] - g["tips"][c]).max())
        q = np.array(pose[0])
        if q[3] < 0:
            q[3:] = -q[3:]
        worst_pose = max(worst_pose, np.abs(q - g["pose"][c]).max())
        assert worst_tip < tol_x and worst_pose < tol_x, (name, c, worst_tip, worst_pose)
        odo = eng.odometry()[0]               # WalkController::odometry_ideal_: the desired body velocity integrated (:643, :783-791)
        if odo[3] < 0:
            odo[3:] = -odo[3:]
        assert np.abs(odo - g["odometry"][c]).max() < tol_x, (name, c, odo, g["odometry"][c])
        if "q" in g:
            worst_q = max(worst_q, np.abs(eng.joints()[0][0].reshape(*LD) - g["q"][c]).max())
            for f in ("poser_tip", "model_tip"):   # what publishLegState sends of the LegPoser's and the leg's tip poses
                assert np.abs(ls[f][0] - g[f][c]).max() < 1e-8, (name, c, f)
            assert worst_q < 1e-6, (name, c, worst_q)
            if meta["overrides"].get("dynamic_stiffness"):
                assert np.abs(eng.virtual_stiffness()[0] - g["stiffness"][c]).max() < 1e-9, (name, c)
            if "effort" in g:
                assert np.abs(ls["tip_force"][0] - g["tip_force_calc"][c]).max() < 1e-8, (name, c)
    eng.close()
    return (f"[HIP engine vs numpy golden] {name}: {meta['cycles']} cycles, walk states {meta['visited_walk_states']}, max |tip diff| {worst_tip:.2e} m, "
            f"max |pose diff| {worst_pose:.2e}" + (f", max |joint diff| {worst_q:.2e} rad (free-running; start-up configurations {start_diff:.1e} rad apart)" if "q" in g else ""))


# ------------------------------------------------------------------------------------------------ LegPoser primitives, sequences
def _seq():
    return np.load(os.path.join(HERE, "sequence_golden.npz"))


def _standing_hexapod(backend, tol=1e-12):
    SEQ = _seq()
    ob, _ = backend(golden_hexapod_params("tripod"))
    assert np.abs(ob.leg_apply_fk() - SEQ["origin"]).max() < tol       # the data the independent generator started from
    assert np.abs(ob.joints()[0].reshape(6, 3) - SEQ["q0"]).max() < tol
    return ob


def replay_step_to_position(backend, name, tol=1e-12):
    """LegPoser::stepToPosition call by call (tests/golden/make_sequence_golden.py): progress values exact, tip positions and directions to tol."""
    SEQ = _seq()
    rows, target, body = SEQ[f"step/{name}/rows"], SEQ[f"step/{name}/target"], SEQ[f"step/{name}/body"]
    leg, lift, time_to_step = int(SEQ[f"step/{name}/args"][0]), float(SEQ[f"step/{name}/args"][1]), float(SEQ[f"step/{name}/args"][2])
    ob = _standing_hexapod(backend, max(tol, 1e-12))
    targets = None
    if not np.isnan(target[0]):
        targets = SEQ["origin"].copy()
        targets[:, 3:] = 0.0              # the other legs: their own tip, rotation undefined -> nothing to do
        targets[leg] = target
    for call, row in enumerate(rows):
        out, progress = ob.leg_step_to_position(targets, body[None], lift, time_to_step, apply_delta=False)
        assert progress[leg] == int(row[0]), (call, progress[leg], row[0])
        assert np.abs(out[leg, :3] - row[1:4]).max() < tol, (call, out[leg, :3], row[1:4])
        q = out[leg, 3:]
        if np.isnan(row[4]):
            assert not q.any()            # UNDEFINED_ROTATION
        else:                             # x axis of the tip rotation
            x = np.array([1 - 2 * (q[2] ** 2 + q[3] ** 2), 2 * (q[1] * q[2] + q[0] * q[3]), 2 * (q[1] * q[3] - q[0] * q[2])])
            assert np.abs(x - row[4:7]).max() < tol, (call, x, row[4:7])
    return len(rows)


def replay_configuration_transition(backend, name, tol=1e-13):
    """LegPoser::transitionConfiguration (pose_controller.cpp:1476-1567) against the independent restatement."""
    SEQ = _seq()
    rows, target = SEQ[f"cfg/{name}/rows"], SEQ[f"cfg/{name}/target"]
    leg, transition_time = int(SEQ[f"cfg/{name}/args"][0]), float(SEQ[f"cfg/{name}/args"][1])
    ob = _standing_hexapod(backend)
    desired = SEQ["q0"].copy()
    desired[leg] = target
    for call, row in enumerate(rows):
        progress = ob.leg_transition_configuration(desired, transition_time)
        assert progress[leg] == int(row[0]), (call, progress[leg], row[0])
        assert np.abs(ob.joints()[0].reshape(6, 3)[leg] - row[1:]).max() < tol
    return len(rows)


def replay_startup_sequence(backend, start, offset_tol=1e-3, octopod_tol=2e-2):
    """PoseController::executeSequence (tests/golden/make_startup_golden.py): a first START_UP, SHUT_DOWN, START_UP again (replay) - every return
    value exactly, joints to 1e-6 rad call by call from the READY estimate."""
    from syropod_highlevel_controller_amd import default_hexapod_params
    g = np.load(os.path.join(HERE, "startup_golden.npz"))
    pre = {"ready": "", "offset": "offset/", "8x5": "8x5/"}[start]
    if start == "8x5":           # the synthetic 8 x 5 octopod from its READY estimate: redundant chains through the same choreography
        from syropod_highlevel_controller_amd import synthetic_octopod_params
        p = synthetic_octopod_params("ripple", 5, 8)
    else:
        p = default_hexapod_params("tripod")
    ob, _ = backend(p)
    ob.begin_sequence_startup(g[pre + "q0"] if start == "offset" else None, False)
    assert np.abs(ob.joints()[0][0].reshape(p.leg_count, p.leg_dof[0]) - g[pre + "q0"]).max() == 0.0
    worst = 0.0
    for name, which in (("startup_first", 0), ("shutdown", 1), ("startup_replay", 0)):
        rows = g[pre + name]
        for call, row in enumerate(rows):
            r = int(ob.execute_sequence(which)[0])
            assert r == int(row[0]), (name, call, r, row[0])
            worst = max(worst, np.abs(ob.joints()[0][0] - row[1:]).max())
            # free-running through a slow body raise: the reference's IK step amplifies rounding differences there (DESIGN.md section
            # 2.1); the READY start stays within 1e-6 rad, the offset start is given what a twin build of the oracle itself needs
            # The octopod's redundant chains: every return value, learnt transition step and workspace alert exactly; its first START_UP
            # within 1e-6 rad, then the end of the SHUT_DOWN amplifies the difference by 10 per call up to 9 mrad before the replay pulls
            # the chains together again - same place, not same rounding.
            tol = 1e-6 if start == "ready" else offset_tol if start == "offset" else (1e-6 if name == "startup_first" else octopod_tol)
            assert worst < tol, (name, call, worst)
        assert int(rows[-1][0]) == 100
    return (f"executeSequence from {start}: {sum(len(g[pre + k]) for k in ('startup_first', 'shutdown', 'startup_replay'))} calls, "
            f"{int(g[pre + 'transition_steps'][0])} transition steps learnt, {int(g[pre + 'proximity_alerts'][0])} workspace alerts, max |joint diff| {worst:.2e} rad")


def replay_step_to_new_stance(backend, start_tol=1e-12):
    """PoseController::stepToNewStance (tests/golden/make_startup_golden.py): return values exactly, joints free-running."""
    g = np.load(os.path.join(HERE, "startup_golden.npz"))
    ob, _ = backend(golden_hexapod_params("tripod"))
    assert np.abs(np.stack([x[0] for x in ob.joints()]).reshape(2, 6, 3) - g["new_stance/joint_start"]).max() < start_tol
    worst = 0.0
    for call, row in enumerate(g["new_stance/rows"]):
        assert int(ob.step_to_new_stance()[0]) == int(row[0]), call
        worst = max(worst, np.abs(ob.joints()[0][0] - row[1:]).max())
        assert worst < 1e-6, (call, worst)
    return f"stepToNewStance: {len(g['new_stance/rows'])} calls, max |joint diff| {worst:.2e} rad"


def replay_pack_unpack(backend, tol=1e-13):
    """PoseController::packLegs through two pack steps, then unpackLegs back (tests/golden/make_sequence_golden.py, Packer): every return
    value exactly - the hand-overs between pack steps return 0 -, joints to tol call by call."""
    SEQ = _seq()
    ob = _standing_hexapod(backend)
    rows, packed, time_to_pack = SEQ["pack/rows"], SEQ["pack/packed"], float(SEQ["pack/time"][0])
    worst = 0.0
    for call, row in enumerate(rows):
        progress = ob.pack_legs(packed, time_to_pack, unpack=bool(row[0]))
        assert int(progress) == int(row[1]), (call, progress, row[1])
        worst = max(worst, np.abs(ob.joints()[0][0] - row[2:]).max())
        assert worst < tol, (call, worst)
    return f"packLegs / unpackLegs: {len(rows)} calls over {len(packed)} pack steps, max |joint diff| {worst:.2e} rad"

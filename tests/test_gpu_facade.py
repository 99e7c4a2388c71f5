"""GPU: the C++ façade (include/shc_facade.hpp — the reference's class/method names over the C ABI) drives one robot
through the reference's loop body and must reproduce the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from oracle_lib import OracleRobot
from syropod_highlevel_controller_amd import default_hexapod_params, engine

pytestmark = pytest.mark.gpu
# Free-running through hundreds of loops of a robot that STANDS (planner steps, leg manipulation): the reference's IK step amplifies
# rounding differences there, so the facade run is held to the oracle as tightly as measured (x10), not to
# 1e-6; the same loops are held to 1e-10 rad per call teacher-forced in tests/test_gpu_planner.py / test_gpu_manual_legs.py.
FACADE_FREE_RUNNING_TIP_TOL = 5e-3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_facade_reproduces_oracle(tmp_path):
    so = engine.build_library()
    exe = str(tmp_path / "facade_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "facade_main.cpp"),
                           so, "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    p = default_hexapod_params("tripod")
    pfile = str(tmp_path / "params.bin")
    open(pfile, "wb").write(bytes(p))
    cycles, v = 200, (0.6, -0.3, 0.4)
    out = subprocess.check_output([exe, pfile, str(cycles), *map(str, v)], text=True).split("\n")
    q_gpu = np.array([float(x) for x in out[:18]])
    assert out[18] == "walk_state 1"
    r = OracleRobot(p)
    r.set_velocity(*v)
    r.cycle(cycles)
    assert np.abs(r.joints()[0] - q_gpu).max() <= 1e-6
    # per-leg methods on leg 2, replayed on the oracle's Leg
    num = lambda line: np.array([float(x) for x in line.split()])
    L = oracle_lib.lib()
    _dp = C.POINTER(C.c_double)
    ptr = lambda a: a.ctypes.data_as(_dp)
    pose = np.zeros(7)
    L.orc_leg_apply_fk(r.h, 2, None, ptr(pose))
    assert np.abs(num(out[19]) - pose[:3]).max() <= 1e-6
    delta, dq = np.array([0.001, -0.002, 0.0015, 0, 0, 0]), np.zeros(3)
    L.orc_leg_solve_ik(r.h, 2, ptr(delta), 0, ptr(dq))
    assert np.abs(num(out[20]) - dq).max() <= 1e-7
    prox = L.orc_leg_update_joint_positions(r.h, 2, ptr(dq), 1)
    assert abs(float(out[21]) - prox) <= 1e-6
    L.orc_leg_apply_fk(r.h, 2, None, ptr(pose))
    pose[2] += 0.003
    pose[3:] = 0.0
    L.orc_leg_set_desired_tip_pose(r.h, 2, ptr(pose), 0)
    res = L.orc_leg_apply_ik(r.h, 2, 1)
    assert abs(float(out[22]) - res) <= 1e-6
    q_leg = r.joints()[0][6:9]
    assert np.abs(np.array([float(x) for x in out[23:26]]) - q_leg).max() <= 1e-6


def test_facade_sequence_start_up_and_external_target(tmp_path):
    """The facade's start_up_sequence path: Engine::initModel + PoseController::executeSequence(START_UP) until complete +
    finishStartUpSequence, then a walk in rough terrain mode and the LegStepper external-target accessors."""
    from oracle_lib import OracleBatch
    so = engine.build_library()
    exe = str(tmp_path / "facade_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "facade_main.cpp"),
                           so, "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    p = default_hexapod_params("tripod")
    p.rough_terrain_mode = 1
    pfile = str(tmp_path / "params.bin")
    open(pfile, "wb").write(bytes(p))
    cycles, v = 120, (0.4, 0.1, -0.3)
    out = subprocess.check_output([exe, pfile, str(cycles), *map(str, v), "sequence"], text=True).split("\n")
    ob = OracleBatch(p, 1)
    ob.begin_sequence_startup(None, False)
    calls = 0
    while True:
        calls += 1
        if ob.execute_sequence(0)[0] == 100:
            break
    assert out[0] == f"calls {calls}"
    q_seq = np.array([float(x) for x in out[1:19]])
    assert np.abs(q_seq - ob.joints()[0][0]).max() <= 1e-6
    ob.finish_sequence_startup()
    ob.set_velocity(np.array([[v[0], v[1]]]), np.array([v[2]]))
    ob.step(cycles, 1)
    q_walk = np.array([float(x) for x in out[19:37]])
    assert np.abs(q_walk - ob.joints()[0][0]).max() <= 1e-6
    tag, defined, x, clearance = out[37].split()     # requested while the robot walks (a STOPPED robot's request goes to the planner)
    assert (tag, defined) == ("external", "1") and float(x) == 0.2 and float(clearance) == 0.03
    # planner mode through the facade: stop, wait for plan step 0, a body-pose step
    ob.set_planner_mode(True)
    stop_calls = step_calls = 0
    while True:
        stop_calls += 1
        if ob.execute_plan()[0][0] == -2:
            break
    ob.set_target_body_pose(np.array([[0.01, -0.005, 0.008, 1.0, 0, 0, 0]]))
    while True:
        step_calls += 1
        pr, st = ob.execute_plan()
        if pr[0] == 100:
            break
    ob.set_planner_mode(False)
    pl = out[38].split()
    assert pl[0] == "plan" and [int(x) for x in pl[1:4]] == [stop_calls, step_calls, int(st[0])] and int(st[0]) == 1
    # free-running through ~250 calls of a robot that stands (where the reference's IK step amplifies rounding differences,
    # tests/test_gpu_planner.py holds every call to 1e-10 teacher-forced); here: same calls, same place
    d_plan = np.abs(np.array([float(x) for x in pl[4:7]]) - ob.leg_state()["model_tip"][0, 0]).max()
    assert d_plan <= FACADE_FREE_RUNNING_TIP_TOL, d_plan
    # manual leg manipulation through the facade: the same requests on the oracle
    sel = np.array([3], dtype=np.int32)
    while ob.toggle_leg_state(sel)[0] != 1:
        pass
    where = np.array([[p.stance_position[3][0] * 0.9, p.stance_position[3][1] * 0.9, -0.07]])
    ob.set_velocity(np.array([[v[0], v[1]]]), np.array([v[2]]))
    ob.set_manual_inputs(sel, None, where, None, None, None)
    ob.step(30, 1)
    m = out[39].split()
    assert m[0] == "manual" and m[1] == "1" and m[6] == "3"                      # toggled, robot STOPPED
    d_manual = np.abs(np.array([float(x) for x in m[2:5]]) - ob.leg_state()["model_tip"][0, 3]).max()
    from conftest import parity_report
    parity_report(f"[facade, free-running through the planner / manual-leg loops of a standing robot] model tip vs oracle: {d_plan:.2e} m after the plan, "
                  f"{d_manual:.2e} m after the manual leg moves")
    assert d_manual <= 1e-9, d_manual                                            # (placed by position input: no drift to speak of)


def test_facade_adjust_parameter(tmp_path):
    """Engine::adjustParameter as a node's runningState would call it: asked in every loop while parameter_adjust_flag_ is set.  The oracle serves the same
    requests inside its loops (orc_request_parameter_adjust); the step-frequency change waits the same number of loops on both sides."""
    so = engine.build_library()
    exe = str(tmp_path / "facade_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "facade_main.cpp"),
                           so, "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    p = default_hexapod_params("tripod")
    pfile = str(tmp_path / "params.bin")
    open(pfile, "wb").write(bytes(p))
    cycles, v, freq = 260, (0.7, 0.2, 0.25), 0.7
    out = subprocess.check_output([exe, pfile, str(cycles), *map(str, v), "adjust", str(freq)], text=True).split("\n")
    waited, flag = int(out[0].split()[1]), int(out[0].split()[3])
    q_gpu = np.array([float(x) for x in out[1:19]])
    L = oracle_lib.lib()
    L.orc_request_parameter_adjust.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.orc_parameter_adjust_pending.argtypes = [C.c_void_p]
    r = OracleRobot(p)
    r.set_velocity(*v)
    waited_ref = 0
    for c in range(cycles):
        if c == 40:
            L.orc_request_parameter_adjust(r.h, 2, 0.03)
        if c == 60:
            L.orc_request_parameter_adjust(r.h, 1, freq)
        r.cycle(1)
        waited_ref += int(L.orc_parameter_adjust_pending(r.h))
    assert flag == 0 and waited == waited_ref and waited > 0, (waited, waited_ref, flag)
    assert np.abs(r.joints()[0] - q_gpu).max() <= 1e-6
    assert r.tables().step.period != OracleRobot(p).tables().step.period

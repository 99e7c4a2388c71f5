"""ctypes loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from syropod_highlevel_controller_amd.params import InstanceState, LegStateMsg, Params, StepCycle, Tables

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "liboracle.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _ptr(a, typ=_dp):
    if a is None:
        return None
    return a.ctypes.data_as(typ)


def build_oracle(force: bool = False) -> str:
    srcs = [os.path.join(_ROOT, "oracle", f) for f in ("shc_oracle.c", "shc_oracle.h", "oracle_math.h")]
    srcs.append(os.path.join(_ROOT, "include", "shc_batch.h"))
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Params)]
        L.orc_clone.restype = C.c_void_p
        L.orc_clone.argtypes = [C.c_void_p]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_startup.argtypes = [C.c_void_p]
        L.orc_get_tables.argtypes = [C.c_void_p, C.POINTER(Tables)]
        L.orc_set_velocity.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_set_imu.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_set_tip_force.argtypes = [C.c_void_p, _dp]
        L.orc_set_joint_effort.argtypes = [C.c_void_p, _dp]
        L.orc_set_pose_input.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_set_pose_reset_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_cycle.argtypes = [C.c_void_p]
        L.orc_get_joint_state.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_get_leg_state.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _ip]
        L.orc_get_body_state.argtypes = [C.c_void_p, _dp, _dp, _ip]
        L.orc_get_ik_failures.argtypes = [C.c_void_p]
        L.orc_batch_create.restype = C.c_void_p
        L.orc_batch_create.argtypes = [C.POINTER(Params), C.c_int64]
        L.orc_batch_destroy.argtypes = [C.c_void_p]
        L.orc_batch_robot.restype = C.c_void_p
        L.orc_batch_robot.argtypes = [C.c_void_p, C.c_int64]
        L.orc_batch_set_velocity.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_batch_set_imu.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_batch_set_tip_force.argtypes = [C.c_void_p, _dp]
        L.orc_batch_set_joint_effort.argtypes = [C.c_void_p, _dp]
        L.orc_batch_set_pose_input.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_batch_set_pose_reset_mode.argtypes = [C.c_void_p, _ip]
        L.orc_batch_step.restype = C.c_double
        L.orc_batch_step.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_batch_get_joint_state.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_batch_get_leg_state.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _ip]
        L.orc_batch_get_body_state.argtypes = [C.c_void_p, _dp, _dp, _ip]
        L.orc_batch_get_odometry.argtypes = [C.c_void_p, _dp]
        L.orc_get_leg_state_msg.argtypes = [C.c_void_p, C.POINTER(LegStateMsg)]
        L.orc_batch_change_gait.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.orc_batch_change_gait.restype = C.c_int64
        L.orc_batch_adjust_parameter.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.orc_batch_adjust_parameter.restype = C.c_int64
        L.orc_adjust_parameter.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.orc_batch_get_virtual_stiffness.argtypes = [C.c_void_p, _dp]
        L.orc_set_joint_states_msg.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_set_step_plane.argtypes = [C.c_void_p, _dp]
        L.orc_sequence_begin.argtypes = [C.c_void_p, _dp]
        for f in ("orc_sequence_prologue", "orc_sequence_finish_startup", "orc_sequence_finish_shutdown"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_execute_sequence.argtypes = [C.c_void_p, C.c_int]
        L.orc_step_to_new_stance.argtypes = [C.c_void_p]
        L.orc_sequence_failed.argtypes = [C.c_void_p]
        L.orc_leg_state_toggle.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_leg_manipulation_state.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_manual_inputs.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        L.orc_set_planner_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_target_configuration.argtypes = [C.c_void_p, _dp]
        L.orc_set_target_body_pose.argtypes = [C.c_void_p, _dp]
        L.orc_execute_plan.argtypes = [C.c_void_p]
        L.orc_get_plan_step.argtypes = [C.c_void_p]
        L.orc_pack_legs.argtypes = [C.c_void_p, _dp, C.c_int, C.c_double]
        L.orc_unpack_legs.argtypes = [C.c_void_p, _dp, C.c_int, C.c_double]
        L.orc_set_external_target.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_set_external_target.restype = C.c_int
        L.orc_set_external_transform.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
        L.orc_get_external_target.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_get_joint_commands.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.orc_leg_set_desired_tip_pose.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int]
        L.orc_leg_solve_ik.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp]
        L.orc_leg_update_joint_positions.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int]
        L.orc_leg_update_joint_positions.restype = C.c_double
        L.orc_leg_apply_ik.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_leg_apply_ik.restype = C.c_double
        L.orc_leg_apply_fk.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.orc_leg_step_to_position.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_double, C.c_double, C.c_int, _dp]
        L.orc_leg_transition_configuration.argtypes = [C.c_void_p, C.c_int, _dp, C.c_double]
        L.orc_startup_begin.argtypes = [C.c_void_p]
        L.orc_startup_step.argtypes = [C.c_void_p]
        L.orc_startup_finish.argtypes = [C.c_void_p]
        L.orc_get_state.argtypes = [C.c_void_p, C.POINTER(InstanceState)]
        L.orc_set_state.argtypes = [C.c_void_p, C.POINTER(InstanceState)]
        L.orc_batch_get_state.argtypes = [C.c_void_p, C.POINTER(InstanceState)]
        L.orc_batch_set_state.argtypes = [C.c_void_p, C.POINTER(InstanceState)]
        L.orc_test_generate_step_cycle.argtypes = [C.POINTER(Params), C.POINTER(StepCycle)]
        L.orc_test_quat_to_euler.argtypes = [_dp, C.c_int, _dp]
        L.orc_test_euler_to_quat.argtypes = [_dp, C.c_int, _dp]
        L.orc_test_from_two_vectors.argtypes = [_dp, _dp, _dp]
        L.orc_test_slerp.argtypes = [_dp, C.c_double, _dp, _dp]
        L.orc_test_quat_from_matrix.argtypes = [_dp, _dp]
        L.orc_test_lu_inverse.argtypes = [_dp, C.c_int, _dp]
        L.orc_test_dh_matrix.argtypes = [C.c_double] * 4 + [_dp]
        L.orc_test_quartic_bezier.argtypes = [_dp, C.c_double, _dp, _dp]
        L.orc_test_leg_fk.argtypes = [C.POINTER(Params), C.c_int, _dp, _dp, _dp]
        L.orc_test_leg_ik_step.restype = C.c_double
        L.orc_test_leg_ik_step.argtypes = [C.POINTER(Params), C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp]
        L.orc_test_admittance.argtypes = [C.POINTER(Params), _dp, _dp, _dp]
        _lib = L
    return _lib


class OracleRobot:
    """One reference-structured robot (StateController + Model + controllers) on the CPU oracle."""

    def __init__(self, params: Params, startup: bool = True):
        self.p = params
        self.L = lib()
        self.h = self.L.orc_create(C.byref(params))
        if not self.h:
            raise ValueError("orc_create rejected the parameters")
        self.legs = params.leg_count
        self.dof = params.dof_total()
        if startup:
            n = self.L.orc_startup(self.h)
            if n < 0:
                raise RuntimeError("oracle start-up failed")
            self.startup_loops = n

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def tables(self) -> Tables:
        t = Tables()
        self.L.orc_get_tables(self.h, C.byref(t))
        return t

    def set_velocity(self, vx, vy, w):
        self.L.orc_set_velocity(self.h, vx, vy, w)

    def set_imu(self, quat, gyro):
        q = np.ascontiguousarray(quat, dtype=np.float64)
        g = np.ascontiguousarray(gyro, dtype=np.float64)
        self.L.orc_set_imu(self.h, _ptr(q), _ptr(g))

    def set_tip_force(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        self.L.orc_set_tip_force(self.h, _ptr(f))

    def set_joint_effort(self, e):
        e = np.ascontiguousarray(e, dtype=np.float64)
        self.L.orc_set_joint_effort(self.h, _ptr(e))

    def set_pose_input(self, tv, rv):
        tv = np.ascontiguousarray(tv, dtype=np.float64)
        rv = np.ascontiguousarray(rv, dtype=np.float64)
        self.L.orc_set_pose_input(self.h, _ptr(tv), _ptr(rv))

    def cycle(self, n=1):
        for _ in range(n):
            self.L.orc_cycle(self.h)

    def joints(self):
        q = np.zeros(self.dof)
        qd = np.zeros(self.dof)
        self.L.orc_get_joint_state(self.h, _ptr(q), _ptr(qd))
        return q, qd

    def leg_state(self):
        out = {k: np.zeros((self.legs, 3)) for k in ("walker_tip", "poser_tip", "model_tip", "tip_force", "admittance")}
        st = np.zeros(self.legs, dtype=np.int32)
        self.L.orc_get_leg_state(self.h, _ptr(out["walker_tip"]), _ptr(out["poser_tip"]), _ptr(out["model_tip"]),
                                 _ptr(out["tip_force"]), _ptr(out["admittance"]), _ptr(st, _ip))
        out["leg_status"] = st
        return out

    def body_state(self):
        pose = np.zeros(7)
        vel = np.zeros(3)
        ws = np.zeros(1, dtype=np.int32)
        self.L.orc_get_body_state(self.h, _ptr(pose), _ptr(vel), _ptr(ws, _ip))
        return pose, vel, int(ws[0])

    def ik_failures(self):
        return self.L.orc_get_ik_failures(self.h)


class OracleBatch:
    """n robots cloned from one started-up robot; inputs/outputs in the C ABI's instance-major layouts."""

    def __init__(self, params: Params, n: int):
        self.p, self.n = params, n
        self.L = lib()
        self.h = self.L.orc_batch_create(C.byref(params), n)
        if not self.h:
            raise RuntimeError("orc_batch_create failed")
        self.legs, self.dof = params.leg_count, params.dof_total()

    def __del__(self):
        try:
            self.L.orc_batch_destroy(self.h)
        except Exception:
            pass

    def set_velocity(self, lin, ang):
        lin = np.ascontiguousarray(lin, dtype=np.float64)
        ang = np.ascontiguousarray(ang, dtype=np.float64)
        self.L.orc_batch_set_velocity(self.h, _ptr(lin), _ptr(ang))

    def set_imu(self, quat, gyro):
        quat = np.ascontiguousarray(quat, dtype=np.float64)
        gyro = np.ascontiguousarray(gyro, dtype=np.float64)
        self.L.orc_batch_set_imu(self.h, _ptr(quat), _ptr(gyro))

    def set_tip_force(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        self.L.orc_batch_set_tip_force(self.h, _ptr(f))

    def set_joint_effort(self, e):
        e = np.ascontiguousarray(e, dtype=np.float64)
        self.L.orc_batch_set_joint_effort(self.h, _ptr(e))

    def set_pose_input(self, tv, rv):
        tv = np.ascontiguousarray(tv, dtype=np.float64)
        rv = np.ascontiguousarray(rv, dtype=np.float64)
        self.L.orc_batch_set_pose_input(self.h, _ptr(tv), _ptr(rv))

    def set_pose_reset_mode(self, mode):
        mode = np.ascontiguousarray(mode, dtype=np.int32)
        self.L.orc_batch_set_pose_reset_mode(self.h, _ptr(mode, _ip))

    def step(self, n_cycles=1, threads=1) -> float:
        return self.L.orc_batch_step(self.h, n_cycles, threads)

    def joints(self):
        q = np.zeros((self.n, self.dof))
        qd = np.zeros((self.n, self.dof))
        self.L.orc_batch_get_joint_state(self.h, _ptr(q), _ptr(qd))
        return q, qd

    def leg_state(self):
        out = {k: np.zeros((self.n, self.legs, 3)) for k in ("walker_tip", "poser_tip", "model_tip", "tip_force", "admittance")}
        st = np.zeros((self.n, self.legs), dtype=np.int32)
        self.L.orc_batch_get_leg_state(self.h, _ptr(out["walker_tip"]), _ptr(out["poser_tip"]), _ptr(out["model_tip"]),
                                       _ptr(out["tip_force"]), _ptr(out["admittance"]), _ptr(st, _ip))
        out["leg_status"] = st
        return out

    def body_state(self):
        pose = np.zeros((self.n, 7))
        vel = np.zeros((self.n, 3))
        ws = np.zeros(self.n, dtype=np.int32)
        self.L.orc_batch_get_body_state(self.h, _ptr(pose), _ptr(vel), _ptr(ws, _ip))
        return pose, vel, ws

    # ---- ROS message payloads
    def set_joint_states_msg(self, position=None, velocity=None, effort=None):
        a = [None if x is None else np.ascontiguousarray(x, dtype=np.float64).reshape(self.n, -1) for x in (position, velocity, effort)]
        for i in range(self.n):
            self.L.orc_set_joint_states_msg(self.L.orc_batch_robot(self.h, i), *[None if x is None else _ptr(x[i]) for x in a])

    def set_tip_states_msg(self, wrench_force=None, step_plane=None):
        if wrench_force is not None:
            self.set_tip_force(wrench_force)
        if step_plane is not None:
            sp = np.ascontiguousarray(step_plane, dtype=np.float64).reshape(self.n, -1)
            for i in range(self.n):
                self.L.orc_set_step_plane(self.L.orc_batch_robot(self.h, i), _ptr(sp[i]))

    # ---- start-up / shut-down sequences, one call per robot like the engine's batched entry points
    def begin_sequence_startup(self, joint_positions=None, per_instance=False):
        p = self.p
        nl = p.leg_count
        if joint_positions is None:   # (packed leg by leg: the legs of a robot may differ in joint count)
            q = np.array([p.joint[l][j].unpacked for l in range(nl) for j in range(p.leg_dof[l])], dtype=np.float64)
            rows = [q] * self.n
        else:
            a = np.ascontiguousarray(joint_positions, dtype=np.float64)
            rows = list(a.reshape(self.n, -1)) if per_instance else [a.ravel()] * self.n
        for i in range(self.n):
            self.L.orc_sequence_begin(self.L.orc_batch_robot(self.h, i), _ptr(np.ascontiguousarray(rows[i])))

    def execute_sequence(self, sequence):
        """One call per robot; a robot that has completed `sequence` is left alone until the other one is requested (its node
        would have left transitionRobotState) - the engine's batched call does the same."""
        if not hasattr(self, "_seq_done"):
            self._seq_done = np.full(self.n, -1)
        out = np.zeros(self.n, dtype=np.int32)
        for i in range(self.n):
            if self._seq_done[i] == sequence:
                out[i] = 100
                continue
            self._seq_done[i] = -1
            r = self.L.orc_batch_robot(self.h, i)
            self.L.orc_sequence_prologue(r)          # the posing part of the loop the call sits in
            out[i] = self.L.orc_execute_sequence(r, int(sequence))
            assert not self.L.orc_sequence_failed(r)
            if out[i] == 100:
                self._seq_done[i] = sequence
        return out

    def finish_sequence_startup(self):
        for i in range(self.n):
            self.L.orc_sequence_finish_startup(self.L.orc_batch_robot(self.h, i))

    def finish_sequence_shutdown(self):
        for i in range(self.n):
            self.L.orc_sequence_finish_shutdown(self.L.orc_batch_robot(self.h, i))

    # ---- manual leg manipulation
    def toggle_leg_state(self, leg_selection):
        """One StateController loop per robot, with the toggle request pending (-1 = no request: result -3, an ordinary loop)."""
        out = np.full(self.n, -3, dtype=np.int32)
        for i in range(self.n):
            if leg_selection[i] >= 0:
                out[i] = self.L.orc_leg_state_toggle(self.L.orc_batch_robot(self.h, i), int(leg_selection[i]))
            else:
                self.L.orc_cycle(self.L.orc_batch_robot(self.h, i))
        return out

    def set_manual_inputs(self, primary_leg=None, primary_velocity=None, primary_position=None, secondary_leg=None, secondary_velocity=None,
                          secondary_position=None):
        f = lambda a, i: None if a is None else _ptr(np.ascontiguousarray(a[i], dtype=np.float64))
        for i in range(self.n):
            self.L.orc_set_manual_inputs(self.L.orc_batch_robot(self.h, i), -1 if primary_leg is None else int(primary_leg[i]), f(primary_velocity, i),
                                         f(primary_position, i), -1 if secondary_leg is None else int(secondary_leg[i]), f(secondary_velocity, i),
                                         f(secondary_position, i))

    def leg_manipulation_state(self):
        return np.array([[self.L.orc_get_leg_manipulation_state(self.L.orc_batch_robot(self.h, i), l) for l in range(self.legs)] for i in range(self.n)], dtype=np.int32)

    # ---- planner mode
    def set_planner_mode(self, on):
        for i in range(self.n):
            self.L.orc_set_planner_mode(self.L.orc_batch_robot(self.h, i), int(bool(on)))

    def set_target_configuration(self, configuration, first=0):
        a = np.ascontiguousarray(configuration, dtype=np.float64).reshape(-1, self.dof)
        for k in range(a.shape[0]):
            self.L.orc_set_target_configuration(self.L.orc_batch_robot(self.h, first + k), _ptr(a[k]))

    def set_target_body_pose(self, pose, first=0):
        a = np.ascontiguousarray(pose, dtype=np.float64).reshape(-1, 7)
        for k in range(a.shape[0]):
            self.L.orc_set_target_body_pose(self.L.orc_batch_robot(self.h, first + k), _ptr(a[k]))

    def execute_plan(self):
        progress = np.array([self.L.orc_execute_plan(self.L.orc_batch_robot(self.h, i)) for i in range(self.n)], dtype=np.int32)
        step = np.array([self.L.orc_get_plan_step(self.L.orc_batch_robot(self.h, i)) for i in range(self.n)], dtype=np.int32)
        return progress, step

    def pack_legs(self, packed_positions, time_to_pack, unpack=False):
        a = np.ascontiguousarray(packed_positions, dtype=np.float64)
        steps = a.size // self.dof
        f = self.L.orc_unpack_legs if unpack else self.L.orc_pack_legs
        out = [f(self.L.orc_batch_robot(self.h, i), _ptr(a), steps, float(time_to_pack)) for i in range(self.n)]
        assert len(set(out)) == 1
        return out[0]

    def step_to_new_stance(self):
        out = np.zeros(self.n, dtype=np.int32)
        for i in range(self.n):
            r = self.L.orc_batch_robot(self.h, i)
            self.L.orc_sequence_prologue(r)
            out[i] = self.L.orc_step_to_new_stance(r)
        return out

    # ---- external targets / defaults, rows in instance-major (instance, leg) order like the engine's call
    def set_external_target(self, rows, which=0):
        legs = self.legs
        assert len(rows) == self.n * legs
        ignored = 0
        for i in range(self.n):
            for l in range(legs):
                took = self.L.orc_set_external_target(self.L.orc_batch_robot(self.h, i), which, l, C.byref(rows[i * legs + l]))
                ignored += 0 if took else 1  # 1 a LegStepper took it, 2 the planner-mode LegPoser (robot STOPPED), 0 dropped
        return ignored

    def set_external_transform(self, transform, which=0):
        t = np.ascontiguousarray(transform, dtype=np.float64).reshape(self.n, self.legs, 7)
        for i in range(self.n):
            for l in range(self.legs):
                self.L.orc_set_external_transform(self.L.orc_batch_robot(self.h, i), which, l, _ptr(t[i, l]))

    def get_external_target(self, which=0):
        from syropod_highlevel_controller_amd.params import ExternalTarget
        rows = (ExternalTarget * (self.n * self.legs))()
        for i in range(self.n):
            for l in range(self.legs):
                self.L.orc_get_external_target(self.L.orc_batch_robot(self.h, i), which, l, C.byref(rows[i * self.legs + l]))
        return rows

    def joint_commands(self):
        out = [np.zeros((self.n, self.dof)) for _ in range(4)]
        for i in range(self.n):
            self.L.orc_get_joint_commands(self.L.orc_batch_robot(self.h, i), *[_ptr(o[i]) for o in out])
        return out

    # ---- per-leg Leg methods, every (robot, leg) in instance-major order like the engine's shc_leg_* calls
    def _each(self):
        for i in range(self.n):
            r = self.L.orc_batch_robot(self.h, i)
            for l in range(self.legs):
                yield i * self.legs + l, r, l

    def leg_set_desired_tip_pose(self, tip_pose=None, apply_delta=True):
        a = None if tip_pose is None else np.ascontiguousarray(tip_pose, dtype=np.float64).reshape(-1, 7)
        for k, r, l in self._each():
            self.L.orc_leg_set_desired_tip_pose(r, l, None if a is None else _ptr(a[k]), int(apply_delta))

    def leg_solve_ik(self, delta, solve_rotation=False):
        a = np.ascontiguousarray(delta, dtype=np.float64).reshape(-1, 6)
        D = self.dof // self.legs
        out = np.zeros((self.n * self.legs, D))
        for k, r, l in self._each():
            self.L.orc_leg_solve_ik(r, l, _ptr(a[k]), int(solve_rotation), _ptr(out[k]))
        return out

    def leg_update_joint_positions(self, joint_delta, simulation=False):
        D = self.dof // self.legs
        a = np.ascontiguousarray(joint_delta, dtype=np.float64).reshape(-1, D)
        return np.array([self.L.orc_leg_update_joint_positions(r, l, _ptr(a[k]), int(simulation)) for k, r, l in self._each()])

    def leg_apply_ik(self, simulation=False):
        return np.array([self.L.orc_leg_apply_ik(r, l, int(simulation)) for k, r, l in self._each()])

    def leg_apply_fk(self, joint_position=None):
        D = self.dof // self.legs
        a = None if joint_position is None else np.ascontiguousarray(joint_position, dtype=np.float64).reshape(-1, D)
        out = np.zeros((self.n * self.legs, 7))
        for k, r, l in self._each():
            self.L.orc_leg_apply_fk(r, l, None if a is None else _ptr(a[k]), _ptr(out[k]))
        return out

    def leg_step_to_position(self, target_tip_pose, target_pose, lift_height, time_to_step, apply_delta=True):
        a = None if target_tip_pose is None else np.ascontiguousarray(target_tip_pose, dtype=np.float64).reshape(-1, 7)
        b = np.ascontiguousarray(target_pose, dtype=np.float64).reshape(-1, 7)
        out, prog = np.zeros((self.n * self.legs, 7)), np.zeros(self.n * self.legs, dtype=np.int32)
        for k, r, l in self._each():
            prog[k] = self.L.orc_leg_step_to_position(r, l, None if a is None else _ptr(a[k]), _ptr(b[k // self.legs]), lift_height,
                                                      time_to_step, int(apply_delta), _ptr(out[k]))
        return out, prog

    def leg_transition_configuration(self, desired_configuration, transition_time):
        D = self.dof // self.legs
        a = np.ascontiguousarray(desired_configuration, dtype=np.float64).reshape(-1, D)
        return np.array([self.L.orc_leg_transition_configuration(r, l, _ptr(a[k]), transition_time) for k, r, l in self._each()], dtype=np.int32)

    def get_state(self):
        """Full controller state of every robot as a ctypes array of InstanceState (shc_instance_state)."""
        arr = (InstanceState * self.n)()
        self.L.orc_batch_get_state(self.h, arr)
        return arr

    def set_state(self, states):
        assert len(states) == self.n
        self.L.orc_batch_set_state(self.h, states)

    def leg_state_msg(self, instance):
        arr = (LegStateMsg * self.p.leg_count)()
        self.L.orc_get_leg_state_msg(self.L.orc_batch_robot(self.h, int(instance)), arr)
        return list(arr)

    def change_gait(self, new_gait):
        still = int(self.L.orc_batch_change_gait(self.h, C.byref(new_gait)))
        if still == 0:
            self.p = new_gait
        return still

    def tables(self):
        """The step cycle / phase offsets / limit maps robot 0 holds now (every robot of a batch shares them)."""
        t = Tables()
        self.L.orc_get_tables(self.L.orc_batch_robot(self.h, 0), C.byref(t))
        return t

    def adjust_parameter(self, which, value):
        """StateController::adjustParameter, decided for the batch as a whole (as shc_engine_adjust_parameter does): instances still waiting."""
        waiting = int(self.L.orc_batch_adjust_parameter(self.h, int(which), float(value)))
        assert waiting >= 0
        from syropod_highlevel_controller_amd.params import PARAM_FIELD
        setattr(self.p, PARAM_FIELD[int(which)], float(value))
        return waiting

    def odometry(self):
        pose = np.zeros((self.n, 7))
        self.L.orc_batch_get_odometry(self.h, _ptr(pose))
        return pose

    def virtual_stiffness(self):
        k = np.zeros((self.n, self.p.leg_count))
        self.L.orc_batch_get_virtual_stiffness(self.h, _ptr(k))
        return k

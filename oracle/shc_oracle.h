/*
 * shc_oracle.h — TEST INFRASTRUCTURE: CPU oracle for the batched leg-control engine.
 *
 * A scalar, one-robot-at-a-time C restatement of the reference's per-cycle hot path
 * (OpenSHC v0.5.11; file:line citations in shc_oracle.c are relative to /root/reference) and of the
 * host init chain the hot path's tables come from.  It keeps the reference's structure on purpose
 * (per robot -> per leg loops, full 6x6 partial-pivot LU DLS inverse, 30-step RK4 admittance, O(N^2)
 * chain products), so it doubles as the "port" CPU baseline of bench.py.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 *  and cannot be compiled in this image (needs ROS-1, Eigen3, Boost.odeint; none
 * installed, no network), so this oracle is pinned only by (1) hand-derived known answers from the
 * reference formulae (tests/test_oracle_golden.py::test_step_cycle_known_answers), (2) independent numpy/scipy restatements
 * written from the reference sources alone and committed with their fixtures (tests/golden/): the math primitives
 * (make_golden.py), the walking loop over hundreds of cycles incl. rough terrain mode's model-free branches and - for the
 * algorithm paths of BASELINE.json configs 2, 3 and 4 - the whole control cycle with the kinematic model, free-running
 * (make_walk_golden.py: oracle joints within 3e-10 rad), the LegPoser primitives behind sequences / leg manipulation / planner mode
 * (make_sequence_golden.py), the start-up / shut-down choreography (make_startup_golden.py) manual leg manipulation
 * (make_manual_golden.py) and planner mode (make_planner_golden.py), (3) the reference's own runtime invariants (FK(IK(x)) within IK_TOLERANCE, C0/C1
 * continuity of the swing/stance Beziers, sequences and plan steps reaching their goals; tests/test_oracle_invariants.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import, call, link or execute
 * anything under oracle/.  The product (libshc_batch.so) never links this.
 */
#ifndef SHC_ORACLE_H
#define SHC_ORACLE_H

#include <stddef.h>
#include "../include/shc_batch.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_robot orc_robot;

/* StateController ctor + init() + initModel(use_default_joint_positions = true)
 * (state_controller.cpp:13-153, main.cpp:100-101).  Returns NULL on unsupported parameters. */
orc_robot *orc_create(const shc_params *params);
void orc_destroy(orc_robot *r);
orc_robot *orc_clone(const orc_robot *r);
size_t orc_sizeof_robot(void);

/* Run StateController::loop() from UNKNOWN through PACKED -> (direct start-up) -> READY -> RUNNING
 * (state_controller.cpp:197-272).  Returns the number of loop iterations taken, < 0 on failure. */
int orc_startup(orc_robot *r);
void orc_get_tables(const orc_robot *r, shc_tables *out);

/* Inputs (ROS callbacks in the reference). */
void orc_set_velocity(orc_robot *r, double vx, double vy, double omega);       /* state_controller.cpp:1127 */
void orc_set_imu(orc_robot *r, const double quat_wxyz[4], const double gyro[3]); /* :1552, model.h:146 */
void orc_set_tip_force(orc_robot *r, const double *force /* [legs][3] */);      /* :1618 */
void orc_set_joint_effort(orc_robot *r, const double *effort /* [legs][dof] */); /* :1590 */
void orc_set_pose_input(orc_robot *r, const double tv[3], const double rv[3]);  /* :1142 */
void orc_set_pose_reset_mode(orc_robot *r, int mode);

/* One StateController::loop() with robot_state == RUNNING (state_controller.cpp:162-193, 379-447). */
void orc_cycle(orc_robot *r);

/* Outputs. */
void orc_get_joint_state(const orc_robot *r, double *q, double *qd);            /* [legs][dof] */
void orc_get_leg_state(const orc_robot *r, double *walker_tip, double *poser_tip, double *model_tip,
                       double *tip_force, double *admittance, int32_t *leg_status);
void orc_get_body_state(const orc_robot *r, double pose[7], double velocity[3], int32_t *walk_state);
void orc_get_odometry(const orc_robot *r, double pose[7]); /* WalkController::odometry_ideal_: xyz + wxyz */
int orc_get_ik_failures(const orc_robot *r);

/*
 * Batch driver used by tests and by bench.py's cpu_baseline leg: `n` independent robots cloned from
 * one started-up robot, per-instance inputs in the C-ABI's instance-major layouts (NULL = zeros),
 * `n_cycles` cycles each, split over `n_threads` pthreads.  Outputs q/qd [n][legs][dof] (may be NULL).
 * Returns wall seconds spent in the cycle loop (excluding start-up and cloning), < 0 on failure.
 */
typedef struct orc_batch orc_batch;
orc_batch *orc_batch_create(const shc_params *params, int64_t n);
void orc_batch_destroy(orc_batch *b);
orc_robot *orc_batch_robot(orc_batch *b, int64_t i);
void orc_batch_set_velocity(orc_batch *b, const double *lin_xy, const double *ang);
void orc_batch_set_imu(orc_batch *b, const double *quat, const double *gyro);
void orc_batch_set_tip_force(orc_batch *b, const double *force);
void orc_batch_set_joint_effort(orc_batch *b, const double *effort);
void orc_batch_set_pose_input(orc_batch *b, const double *tv, const double *rv);
void orc_batch_set_pose_reset_mode(orc_batch *b, const int32_t *mode);
double orc_batch_step(orc_batch *b, int n_cycles, int n_threads);
void orc_batch_get_joint_state(orc_batch *b, double *q, double *qd);
void orc_batch_get_leg_state(orc_batch *b, double *walker_tip, double *poser_tip, double *model_tip,
                             double *tip_force, double *admittance, int32_t *leg_status);
void orc_batch_get_body_state(orc_batch *b, double *pose, double *velocity, int32_t *walk_state);
void orc_get_leg_state_msg(const orc_robot *r, shc_leg_state_msg *legs /* [leg_count] */);
int orc_change_gait(orc_robot *r, const shc_params *new_gait);
int64_t orc_batch_change_gait(orc_batch *b, const shc_params *new_gait);
/* StateController::adjustParameter (state_controller.cpp:451-509); which = enum ParameterSelection.  1 = set, 0 = step_frequency still waiting, -1 = unknown */
int orc_adjust_parameter(orc_robot *r, int which, double value);
int orc_adjust_parameter_probe(orc_robot *r, int which, double value);
void orc_adjust_parameter_commit(orc_robot *r, int which);
int64_t orc_batch_adjust_parameter(orc_batch *b, int which, double value); /* instances still waiting; 0 = set for all (committed inside the next loop) */
void orc_request_parameter_adjust(orc_robot *r, int which, double value); /* one robot: served inside its loops as runningState does */
int orc_parameter_adjust_pending(const orc_robot *r);
void orc_batch_get_odometry(orc_batch *b, double *pose /* [n][7]: xyz + wxyz */);
void orc_batch_get_virtual_stiffness(orc_batch *b, double *stiffness /* [n][legs] */);

/* Full controller state <-> shc_instance_state (include/shc_batch.h): the teacher-forced parity tests load the oracle's
 * state into the engine before every cycle (orc_get_state) and the checkpoint tests continue the oracle from an engine
 * snapshot (orc_set_state). */
void orc_get_state(const orc_robot *r, shc_instance_state *out);
void orc_set_state(orc_robot *r, const shc_instance_state *in);
void orc_batch_get_state(orc_batch *b, shc_instance_state *states /* [n] */);
void orc_batch_set_state(orc_batch *b, const shc_instance_state *states /* [n] */);

void orc_set_joint_states_msg(orc_robot *r, const double *position, const double *velocity, const double *effort); /* :1566 */
void orc_sequence_begin(orc_robot *r, const double *q /* [legs][dof] */);   /* main.cpp:99-100 */
void orc_sequence_prologue(orc_robot *r);                                    /* state_controller.cpp:165-181 */
int orc_execute_sequence(orc_robot *r, int sequence /* 0 START_UP, 1 SHUT_DOWN */); /* pose_controller.cpp:145 */
int orc_step_to_new_stance(orc_robot *r);                                    /* pose_controller.cpp:521 */
int orc_sequence_failed(const orc_robot *r);
int orc_pack_legs(orc_robot *r, const double *packed_positions /* [steps][legs][dof] */, int number_pack_steps, double time_to_pack);     /* pose_controller.cpp:615 */
int orc_unpack_legs(orc_robot *r, const double *packed_positions, int number_pack_steps, double time_to_unpack);                        /* :662 */
void orc_sequence_finish_startup(orc_robot *r);                              /* state_controller.cpp:305-313 */
void orc_sequence_finish_shutdown(orc_robot *r);
void orc_set_planner_mode(orc_robot *r, int on);                            /* state_controller.cpp:1262-1281 */
void orc_set_target_configuration(orc_robot *r, const double *configuration); /* :1683-1687; [legs][dof], NaN = leg not named */
void orc_set_target_body_pose(orc_robot *r, const double *pose7);           /* :1691-1702 */
int orc_execute_plan(orc_robot *r);                                         /* :653-698 inside one loop() */
int orc_get_plan_step(const orc_robot *r);
int orc_leg_state_toggle(orc_robot *r, int leg);                            /* state_controller.cpp:541-646 */
int orc_get_leg_manipulation_state(const orc_robot *r, int leg);
void orc_set_manual_inputs(orc_robot *r, int primary_leg, const double *primary_tip_velocity, const double *primary_tip_position,
                           int secondary_leg, const double *secondary_tip_velocity, const double *secondary_tip_position); /* :1247-1330 */
int orc_set_external_target(orc_robot *r, int which, int leg, const shc_external_target *t);   /* state_controller.cpp:1706 */
void orc_set_external_transform(orc_robot *r, int which, int leg, const double *transform);   /* :703-773 */
void orc_get_external_target(const orc_robot *r, int which, int leg, shc_external_target *out);
void orc_set_step_plane(orc_robot *r, const double *step_plane /* [legs][3] */);                                   /* :1651 */
void orc_get_joint_commands(const orc_robot *r, double *position, double *velocity, double *effort, double *position_command); /* :777 */

/* Per-leg Leg methods (model.h:448-492) on leg `leg` of a robot: what the engine's shc_leg_* entry points are checked against. */
void orc_leg_set_desired_tip_pose(orc_robot *r, int leg, const double *pose7 /* NULL = Pose::Undefined() */, int apply_delta);
void orc_leg_solve_ik(orc_robot *r, int leg, const double delta[6], int solve_rotation, double *joint_delta);
double orc_leg_update_joint_positions(orc_robot *r, int leg, const double *joint_delta, int simulation);
double orc_leg_apply_ik(orc_robot *r, int leg, int simulation);
void orc_leg_apply_fk(orc_robot *r, int leg, const double *joint_position /* NULL = desired */, double pose7[7]);

int orc_leg_step_to_position(orc_robot *r, int leg, const double *target_tip_pose7 /* NULL = Pose::Undefined() */, const double *target_pose7,
                             double lift_height, double time_to_step, int apply_delta, double tip_pose7[7]);
int orc_leg_transition_configuration(orc_robot *r, int leg, const double *desired_configuration, double transition_time);
/* The direct start-up of orc_startup, one StateController::loop() at a time. */
void orc_startup_begin(orc_robot *r);
int orc_startup_step(orc_robot *r);
void orc_startup_finish(orc_robot *r);

/* ---- unit-level entry points for the KAT / cross-check tests (thin wrappers over the static code) ---- */
void orc_test_generate_step_cycle(const shc_params *p, shc_step_cycle *out);
void orc_test_quat_to_euler(const double q_wxyz[4], int intrinsic, double out[3]);
void orc_test_euler_to_quat(const double e[3], int intrinsic, double out_wxyz[4]);
void orc_test_from_two_vectors(const double a[3], const double b[3], double out_wxyz[4]);
void orc_test_slerp(const double a[4], double t, const double b[4], double out[4]);
void orc_test_quat_from_matrix(const double m[9], double out_wxyz[4]);
int orc_test_lu_inverse(const double *a, int n, double *inv);
void orc_test_dh_matrix(double d, double theta, double r, double alpha, double out[16]);
void orc_test_quartic_bezier(const double nodes[15], double t, double out[3], double out_dot[3]);
/* Leg-level: FK of leg `leg` at joint angles q -> tip pose (robot frame) */
void orc_test_leg_fk(const shc_params *p, int leg, const double *q, double tip_pos[3], double tip_quat[4]);
/* One applyIK(simulation) from joint state (q, qd) towards desired tip position; returns ik result. */
double orc_test_leg_ik_step(const shc_params *p, int leg, const double *q, const double *qd, const double desired[3],
                            int simulation, double *q_out, double *qd_out, double tip_out[3]);
/* Admittance: one updateAdmittance for a single leg state (x, xdot) and force (3) */
void orc_test_admittance(const shc_params *p, double state[2], const double force[3], double delta_out[3]);

#ifdef __cplusplus
}
#endif
#endif

/*
 * shc_oracle.c — TEST INFRASTRUCTURE: CPU oracle (see shc_oracle.h for the contract and the
 * "PARITY UNPINNED" statement).  Every function cites the reference file:line it restates;
 * paths are relative to /root/reference.  Reference = OpenSHC v0.5.11.
 *
 * Scope restated here: model.cpp FK / DLS IK / joint update / tip force / workspace
 * search; walk_controller.cpp step cycle, walkspace, limits, updateWalk, LegStepper; pose_controller.cpp
 * updateCurrentPose (walk-plane, manual, inclination, IMU, auto), updateStance, direct start-up;
 * admittance_controller.cpp; call order of state_controller.cpp loop()/runningState().
 * Restated with rough_terrain_mode: the layered workspace, touchdown detection, the step-surface target shift, default tip
 * updates at every swing / stance start, external targets / default poses (walk_controller.cpp:984-1107, :1160), the tip-align pose
 * (pose_controller.cpp:1024-1088).  Widened with the build: per-leg Leg methods, LegPoser::stepToPosition /
 * transitionConfiguration, directStartup loop by loop, executeSequence (START_UP / SHUT_DOWN), stepToNewStance, packLegs /
 * unpackLegs, legStateToggle + poseForLegManipulation + updateManual, planner mode (executePlan, transitionConfiguration /
 * transitionStance), the message callbacks' unit handling, full-state export / import.
 * Not restated: ROS I/O, tf lookups, parameter server.
 */
#include "shc_oracle.h"
#include "oracle_math.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

/* model.h:17-25 */
#define IK_TOLERANCE 0.005
#define DLS_COEFFICIENT 0.02
#define JOINT_LIMIT_COST_WEIGHT 0.1
#define BEARING_STEP 45
#define MAX_POSITION_DELTA 0.002
#define MAX_WORKSPACE_RADIUS 1.0
#define WORKSPACE_LAYERS 10
/* pose_controller.h:18-25 */
#define JOINT_TOLERANCE 0.01
#define TIP_TOLERANCE 0.01
#define STABILITY_THRESHOLD 100
#define IMU_POSING_DEADBAND 0.0
/* admittance_controller.h:18 */
#define ADMITTANCE_DEADBAND 0.0
#define PROGRESS_COMPLETE 100 /* standard_includes.h:53 */
#define SAFETY_FACTOR 0.15             /* pose_controller.h:20 */
#define HORIZONTAL_TRANSITION_TIME 1.0 /* :21 */
#define VERTICAL_TRANSITION_TIME 3.0   /* :22 */
#define TRANSITION_STEP_THRESHOLD 20   /* :24 */
#define HALF_BODY_DEPTH 0.05           /* model.h:18 */
enum { SEQ_START_UP = 0, SEQ_SHUT_DOWN = 1 }; /* enum SequenceSelection, parameters_and_states.h:183-188 */

enum { WALKING = 0, MANUAL = 1, WALKING_TO_MANUAL = -1, MANUAL_TO_WALKING = -2 }; /* enum LegState, parameters_and_states.h:87-94 */
#define MAX_MANUAL_LEGS 2 /* state_controller.h:26 */
#define LEG_UNDESIGNATED (-1) /* parameters_and_states.h:159 */
enum { STARTING = 0, MOVING = 1, STOPPING = 2, STOPPED = 3 };     /* :99 */
enum { SWING = 0, STANCE = 1, FORCE_STANCE = 2, FORCE_STOP = 3 }; /* :111 */
enum { POSING = 0, STOP_POSING = 1, POSING_COMPLETE = 2 };        /* :123 */
enum { RS_PACKED = 0, RS_READY = 1, RS_RUNNING = 2, RS_UNKNOWN = -1 }; /* :25 */

typedef struct
{ /* class Joint, model.h:558-646 (index j = id_number_ - 1) */
  double min_position, max_position, offset, unpacked_position, max_angular_speed;
  double desired_position, desired_velocity, desired_effort, prev_desired_position;
  double current_position, current_velocity, current_effort;
  double default_position, default_velocity, default_effort;
  orc_m4 current_transform, identity_transform;
} joint_t;

typedef struct { double d, theta, r, alpha; } link_t; /* class Link, model.h:541-553 */

typedef struct
{ /* class LegStepper, walk_controller.h:486-535 */
  int at_correct_phase, completed_first_step;
  int phase, phase_offset;
  double step_progress, swing_progress, stance_progress;
  int step_state;
  orc_v3 swing_1_nodes[5], swing_2_nodes[5], stance_nodes[5];
  orc_v3 walk_plane, walk_plane_normal, stride_vector, swing_clearance;
  double swing_delta_t, stance_delta_t;
  orc_pose identity_tip_pose, default_tip_pose, current_tip_pose, origin_tip_pose, target_tip_pose;
  orc_v3 current_tip_velocity;
  orc_v3 swing_origin_tip_position, swing_origin_tip_velocity, stance_origin_tip_position;
  int touchdown_detection; /* walk_controller.h:495, raised by tipStatesCallback */
  shc_external_target external_target, external_default; /* struct ExternalTarget (walk_controller.h:38-46, :533-534) */
} stepper_t;

#define ORC_MAX_TRANSITION_POSES 32 /* executeSequence gives up beyond TRANSITION_STEP_THRESHOLD = 20 steps (pose_controller.h:24) */
typedef struct
{ /* class LegPoser, pose_controller.h:429-597 (members used on the path) */
  orc_pose auto_pose, current_tip_pose, origin_tip_pose, target_tip_pose;
  int pose_negation_phase_start, pose_negation_phase_end;
  double negation_transition_ratio;
  int negate_auto_pose;
  int first_iteration, master_iteration_count;
  int has_desired_configuration;
  double desired_configuration[SHC_MAX_JOINTS], origin_configuration[SHC_MAX_JOINTS];
  /* start-up / shut-down sequences (pose_controller.h:591-593) */
  orc_pose transition_poses[ORC_MAX_TRANSITION_POSES]; /* std::vector<Pose> transition_poses_ */
  int n_transition_poses;
  int leg_completed_step;
  shc_external_target external_target; /* LegPoser::external_target_ (pose_controller.h:589): planner-mode tip target */
} leg_poser_t;

/* Workspace = std::map<double, Workplane>, Workplane = std::map<int, double> with the nine bearings 0..360 (model.h:27-32).
 * Kept in insertion order; lookups go through the sorted view below (map semantics: insert never overwrites). */
#define ORC_MAX_PLANES 32
typedef struct
{
  int n;
  double height[ORC_MAX_PLANES];
  double radius[ORC_MAX_PLANES][SHC_N_BEARINGS];
} workspace_t;

static int ws_find(const workspace_t *w, double height)
{
  for (int i = 0; i < w->n; ++i)
    if (w->height[i] == height) return i;
  return -1;
}
static void ws_insert(workspace_t *w, double height, const double *plane)
{
  if (ws_find(w, height) >= 0 || w->n >= ORC_MAX_PLANES) return;
  w->height[w->n] = height;
  memcpy(w->radius[w->n], plane, sizeof w->radius[0]);
  w->n++;
}
static double *ws_at(workspace_t *w, double height) { int i = ws_find(w, height); return i >= 0 ? w->radius[i] : NULL; }
/* indices of the planes bounding `height`: upper = first key > height (map::upper_bound), lower = prev(upper); -1 if none */
static void ws_bounds(const workspace_t *w, double height, int *lower, int *upper)
{
  *lower = *upper = -1;
  for (int i = 0; i < w->n; ++i)
  {
    if (w->height[i] > height && (*upper < 0 || w->height[i] < w->height[*upper])) *upper = i;
    if (w->height[i] <= height && (*lower < 0 || w->height[i] > w->height[*lower])) *lower = i;
  }
}
/* Leg::getWorkplane (model.cpp:514-550); returns 0 for the "undefined" (empty) workplane */
static int ws_get_workplane(const workspace_t *w, double height, double *out)
{
  double lo = 0, hi = 0;
  for (int i = 0; i < w->n; ++i)
  {
    if (i == 0 || w->height[i] < lo) lo = w->height[i];
    if (i == 0 || w->height[i] > hi) hi = w->height[i];
  }
  if (w->n == 0 || !(height >= lo && height <= hi)) return 0;
  if (w->n == 1)
  {
    int i = ws_find(w, 0.0);
    if (i < 0) return 0;
    memcpy(out, w->radius[i], sizeof w->radius[0]);
    return 1;
  }
  int lower, upper;
  ws_bounds(w, height, &lower, &upper);
  if (lower < 0 || upper < 0) return 0; /* (the reference dereferences end() here) */
  double upper_h = orc_set_precision(w->height[upper], 3), lower_h = orc_set_precision(w->height[lower], 3);
  double i = (height - lower_h) / (upper_h - lower_h);
  for (int b = 0; b < SHC_N_BEARINGS; ++b) out[b] = w->radius[lower][b] * (1.0 - i) + w->radius[upper][b] * i;
  return 1;
}

typedef struct
{ /* class Leg, model.h:202-536 */
  int id_number, joint_count, leg_state;
  joint_t joint[SHC_MAX_JOINTS];
  link_t link[SHC_MAX_LINKS];
  orc_m4 tip_current_transform, tip_identity_transform; /* class Tip, model.h:652-703 */
  orc_v3 admittance_delta;
  double admittance_state[2];
  double virtual_stiffness;
  orc_pose desired_tip_pose, current_tip_pose;
  orc_v3 desired_tip_velocity, current_tip_velocity;
  orc_v3 tip_force_calculated, tip_force_measured;
  orc_pose step_plane_pose;
  workspace_t workspace;            /* typedef std::map<double, Workplane> Workspace (model.h:31-32): planes keyed by height */
  int workspace_zero;               /* model.cpp:349-353 */
  int ik_failed;                    /* model.cpp:921 this cycle */
  stepper_t stepper;
  leg_poser_t poser;
} leg_t;

typedef struct
{ /* class AutoPoser, pose_controller.h:333-421 */
  int start_phase, end_phase;
  int start_check, end_check_first, end_check_second, allow_posing;
  double x_amplitude, y_amplitude, z_amplitude, gravity_amplitude, roll_amplitude, pitch_amplitude, yaw_amplitude;
} auto_poser_t;

struct orc_robot
{
  shc_params params;
  /* Model (model.h:57-200) */
  int leg_count;
  double time_delta;
  orc_pose current_pose, default_pose;
  orc_quat imu_orientation;
  orc_v3 imu_angular_velocity;
  leg_t leg[SHC_MAX_LEGS];
  /* WalkController (walk_controller.h:240-275) */
  int walk_state, pose_state;
  shc_step_cycle step;
  double walkspace[SHC_N_BEARINGS];
  orc_v3 walk_plane, walk_plane_normal;
  double desired_linear_velocity[2], desired_angular_velocity;
  orc_pose odometry_ideal;
  double max_linear_speed[SHC_N_BEARINGS], max_angular_speed[SHC_N_BEARINGS];
  double max_linear_acceleration[SHC_N_BEARINGS], max_angular_acceleration[SHC_N_BEARINGS];
  int legs_at_correct_phase, legs_completed_first_step, return_to_default_attempted;
  /* PoseController (pose_controller.h:262-322) */
  int pose_reset_mode;
  orc_v3 translation_velocity_input, rotation_velocity_input;
  orc_pose manual_pose, auto_pose, imu_pose, inclination_pose, pc_default_pose, walk_plane_pose, origin_walk_plane_pose;
  orc_pose tip_align_pose, origin_tip_align_pose; /* pose_controller.h:286-287 */
  int executing_transition;
  /* sequence members (pose_controller.h:273-274, :296-304) */
  int legs_completed_step, current_group, transition_step, transition_step_count;
  int set_target, proximity_alert, horizontal_transition_complete, vertical_transition_complete;
  int first_sequence_execution, reset_transition_sequence, sequence_failed;
  int pack_step; /* pose_controller.h:298 */
  /* planner mode: PoseController::target_configuration_ / target_body_pose_ (pose_controller.h:291-292; per leg, "named in the
   * message" flag + positions) and the StateController flags (state_controller.h:337, :360-363) */
  double target_configuration[SHC_MAX_LEGS][SHC_MAX_JOINTS];
  int target_configuration_named[SHC_MAX_LEGS];
  orc_pose target_body_pose;
  int planner_mode, target_configuration_acquired, target_tip_pose_acquired, target_body_pose_acquired, plan_step;
  /* manual leg manipulation: StateController members (state_controller.h:330-360) */
  int manual_leg_count, primary_leg_selection, secondary_leg_selection;
  orc_v3 primary_tip_velocity_input, secondary_tip_velocity_input;
  orc_pose primary_pose_input, secondary_pose_input;
  auto_poser_t auto_poser[SHC_MAX_AUTO_POSERS];
  int n_auto_posers;
  int auto_posing_state, pose_phase;
  double pose_frequency;
  int pose_phase_length, normaliser, auto_pose_reference_leg;
  orc_v3 rotation_absement_error, rotation_position_error, rotation_velocity_error;
  /* StateController (state_controller.h) */
  int robot_state, new_robot_state, transition_state_flag;
  double linear_velocity_input[2], angular_velocity_input;
  /* StateController::parameter_adjust_flag_ / dynamic_parameter_ / new_parameter_value_ (state_controller.h) and, for a batch that has decided
   * together, the parameter whose :491-492 is due in the next loop */
  int parameter_adjust_flag, dynamic_parameter, adjust_commit_due;
  double new_parameter_value;
  int unstable;
  int startup_progress; /* last return value of directStartup (orc_startup_step) */
};

/* ==================================================================================== Model / Leg */

/* Joint::getTransformFromJoint / Tip::getTransformFromJoint (model.h:594-599, 674-679).
 * element e: 1..joint_count = joint id, joint_count + 1 = tip.  target: joint id the product stops at (0 = origin). */
static orc_m4 transform_from_joint(const leg_t *leg, int e, int target)
{
  const orc_m4 *cur = (e == leg->joint_count + 1) ? &leg->tip_current_transform : &leg->joint[e - 1].current_transform;
  int next_joint_id = e - 1; /* reference_link_->actuating_joint_->id_number_ */
  if (target == next_joint_id) return *cur;
  orc_m4 up = transform_from_joint(leg, next_joint_id, target);
  return orc_m4_mul(&up, cur);
}

static orc_m4 m4_inverse_lu(const orc_m4 *t) /* MatrixXd(transform).inverse(), model.h:615-616 */
{
  orc_m4 r;
  orc_lu_inverse(&t->m[0][0], 4, &r.m[0][0]);
  return r;
}

/* Joint::getPoseJointFrame (model.h:613-617) for joint id e */
static orc_pose joint_pose_joint_frame(const leg_t *leg, int e, orc_pose robot_frame_pose)
{
  orc_m4 t = transform_from_joint(leg, e, 0);
  orc_m4 inv = m4_inverse_lu(&t);
  return orc_pose_transform_m4(robot_frame_pose, &inv);
}

/* Leg::applyFK (model.cpp:945-988), set_current = true, use_actual = false */
static orc_pose leg_apply_fk(orc_robot *r, leg_t *leg)
{
  for (int e = 2; e <= leg->joint_count; ++e)
  { /* joint e: reference link e-1, actuated by joint e-1 */
    const link_t *rl = &leg->link[e - 1];
    double joint_angle = leg->joint[e - 2].desired_position;
    leg->joint[e - 1].current_transform = orc_create_dh_matrix(rl->d, rl->theta + joint_angle, rl->r, rl->alpha);
  }
  {
    const link_t *rl = &leg->link[leg->joint_count];
    double joint_angle = leg->joint[leg->joint_count - 1].desired_position;
    leg->tip_current_transform = orc_create_dh_matrix(rl->d, rl->theta + joint_angle, rl->r, rl->alpha);
  }
  orc_m4 t = transform_from_joint(leg, leg->joint_count + 1, 0);
  orc_pose tip_pose = orc_pose_transform_m4(orc_pose_identity(), &t); /* Tip::getPoseRobotFrame model.h:684 */
  if (orc_pose_ne(leg->current_tip_pose, orc_pose_undefined()))
  {
    /* reference: (a - b) / dt, element-wise division */
    leg->current_tip_velocity.x = (tip_pose.p.x - leg->current_tip_pose.p.x) / r->time_delta;
    leg->current_tip_velocity.y = (tip_pose.p.y - leg->current_tip_pose.p.y) / r->time_delta;
    leg->current_tip_velocity.z = (tip_pose.p.z - leg->current_tip_pose.p.z) / r->time_delta;
  }
  leg->current_tip_pose = tip_pose;
  return tip_pose;
}

/* Leg::init (model.cpp:286-305) */
static void leg_init(orc_robot *r, leg_t *leg, int use_default_joint_positions)
{
  for (int j = 0; j < leg->joint_count; ++j)
  {
    joint_t *jt = &leg->joint[j];
    if (use_default_joint_positions)
    {
      jt->current_position = jt->default_position;
      jt->current_velocity = jt->default_velocity;
      jt->current_effort = jt->default_effort;
    }
    jt->desired_position = jt->current_position;
    jt->desired_velocity = jt->current_velocity;
    jt->desired_effort = jt->current_effort;
    jt->prev_desired_position = jt->desired_position;
  }
  leg_apply_fk(r, leg);
  leg->desired_tip_pose = leg->current_tip_pose;
}

/* Leg::updateDefaultConfiguration (model.cpp:593-601) */
static void leg_update_default_configuration(leg_t *leg)
{
  for (int j = 0; j < leg->joint_count; ++j) leg->joint[j].default_position = leg->joint[j].desired_position;
}

/* Leg::setDesiredTipPose (model.cpp:653-663) */
static void leg_set_desired_tip_pose(leg_t *leg, orc_pose tip_pose, int apply_delta)
{
  int use_poser_tip_pose = orc_pose_eq(orc_pose_undefined(), tip_pose);
  leg->desired_tip_pose = use_poser_tip_pose ? leg->poser.current_tip_pose : tip_pose;
  /* "Don't apply delta to manually manipulated legs" (:655-656) */
  apply_delta = apply_delta && !(leg->leg_state == MANUAL || leg->leg_state == WALKING_TO_MANUAL);
  if (apply_delta) leg->desired_tip_pose.p = orc_v3_add(leg->desired_tip_pose.p, leg->admittance_delta);
}

/* Whether a leg's stepper tip rotations are part of the exchanged state: legs of more than 3 joints with gravity-aligned tips / in
 * rough terrain mode (an externally requested target may carry a rotation), and 3-joint legs under joint_control leg manipulation
 * (updateManual hands the stepper the FK tip pose with its rotation, walk_controller.cpp:688-689).  Elsewhere the rotations are
 * write-only (updateTipRotation's else branch, :1230-1233). */
static int tip_rotations_tracked(const orc_robot *r, const leg_t *leg)
{
  if (leg->joint_count > 3) return r->params.gravity_aligned_tips || r->params.rough_terrain_mode;
  return leg->joint_count == 3 && r->params.leg_manipulation_mode == SHC_MANIPULATION_JOINT_CONTROL;
}

/* Leg::setAdmittanceDelta (model.h:365-368) */
static void leg_set_admittance_delta(leg_t *leg, orc_v3 delta)
{
  leg->admittance_delta = orc_get_projection(delta, orc_quat_rotate(leg->current_tip_pose.r, orc_v3_make(1, 0, 0)));
}

/* Geometric Jacobian columns shared by solveIK (model.cpp:731-747) and calculateTipForce (:671-692).
 * lin[i], ang[i] = linear / angular column of joint i (0-based). */
static void leg_jacobian(const leg_t *leg, orc_v3 *lin, orc_v3 *ang)
{
  orc_m4 te = transform_from_joint(leg, leg->joint_count + 1, 1);
  orc_v3 pe = orc_v3_make(te.m[0][3], te.m[1][3], te.m[2][3]);
  orc_v3 z0 = orc_v3_make(0, 0, 1), p0 = orc_v3_make(0, 0, 0);
  lin[0] = orc_v3_cross(z0, orc_v3_sub(pe, p0));
  ang[0] = z0;
  for (int i = 1; i < leg->joint_count; ++i)
  {
    orc_m4 t = transform_from_joint(leg, i + 1, 1);
    orc_v3 zi = orc_v3_make(t.m[0][2], t.m[1][2], t.m[2][2]);
    orc_v3 pi = orc_v3_make(t.m[0][3], t.m[1][3], t.m[2][3]);
    lin[i] = orc_v3_cross(zi, orc_v3_sub(pe, pi));
    ang[i] = zi;
  }
}

/* Leg::calculateTipForce (model.cpp:667-708) */
static void leg_calculate_tip_force(orc_robot *r, leg_t *leg)
{
  int n = leg->joint_count;
  orc_v3 lin[SHC_MAX_JOINTS], ang[SHC_MAX_JOINTS];
  leg_jacobian(leg, lin, ang);
  double jac[6][SHC_MAX_JOINTS];
  for (int i = 0; i < n; ++i)
  {
    jac[0][i] = lin[i].x; jac[1][i] = lin[i].y; jac[2][i] = lin[i].z;
    jac[3][i] = ang[i].x; jac[4][i] = ang[i].y; jac[5][i] = ang[i].z;
  }
  double a[SHC_MAX_JOINTS * SHC_MAX_JOINTS], ainv[SHC_MAX_JOINTS * SHC_MAX_JOINTS];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
    {
      double s = 0.0;
      for (int k = 0; k < 6; ++k) s += jac[k][i] * jac[k][j];
      a[i * n + j] = s + orc_sqr(DLS_COEFFICIENT) * (i == j ? 1.0 : 0.0);
    }
  orc_lu_inverse(a, n, ainv);
  /* transformation = J * inv ; raw = transformation * torques */
  double transformation[6][SHC_MAX_JOINTS];
  for (int k = 0; k < 6; ++k)
    for (int j = 0; j < n; ++j)
    {
      double s = 0.0;
      for (int i = 0; i < n; ++i) s += jac[k][i] * ainv[i * n + j];
      transformation[k][j] = s;
    }
  double raw[3];
  for (int k = 0; k < 3; ++k)
  {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += transformation[k][j] * leg->joint[j].current_effort;
    raw[k] = s;
  }
  orc_quat rotation = joint_pose_joint_frame(leg, 1, orc_pose_identity()).r;
  orc_v3 raw_tip_force = orc_quat_rotate(rotation, orc_v3_make(raw[0], raw[1], raw[2]));
  double s = 0.15;
  double fg = r->params.force_gain;
  leg->tip_force_calculated.x = s * raw_tip_force.x * fg + (1 - s) * leg->tip_force_calculated.x;
  leg->tip_force_calculated.y = s * raw_tip_force.y * fg + (1 - s) * leg->tip_force_calculated.y;
  leg->tip_force_calculated.z = s * raw_tip_force.z * fg + (1 - s) * leg->tip_force_calculated.z;
}

/* Leg::solveIK (model.cpp:726-795).  delta[6]; out dq[n] */
static void leg_solve_ik(const leg_t *leg, const double delta[6], int solve_rotation, double *dq)
{
  int n = leg->joint_count;
  orc_v3 lin[SHC_MAX_JOINTS], ang[SHC_MAX_JOINTS];
  leg_jacobian(leg, lin, ang);
  double j[6][SHC_MAX_JOINTS];
  for (int i = 0; i < n; ++i)
  {
    j[0][i] = lin[i].x; j[1][i] = lin[i].y; j[2][i] = lin[i].z;
    j[3][i] = solve_rotation ? ang[i].x : 0.0;
    j[4][i] = solve_rotation ? ang[i].y : 0.0;
    j[5][i] = solve_rotation ? ang[i].z : 0.0;
  }
  /* jacobian_inverse = J^T * (J J^T + lambda^2 I6)^-1  (:755) */
  double jjt[36], jjt_inv[36];
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b)
    {
      double s = 0.0;
      for (int k = 0; k < n; ++k) s += j[a][k] * j[b][k];
      jjt[a * 6 + b] = s + orc_sqr(DLS_COEFFICIENT) * (a == b ? 1.0 : 0.0);
    }
  orc_lu_inverse(jjt, 6, jjt_inv);
  double jinv[SHC_MAX_JOINTS][6];
  for (int i = 0; i < n; ++i)
    for (int b = 0; b < 6; ++b)
    {
      double s = 0.0;
      for (int a = 0; a < 6; ++a) s += j[a][i] * jjt_inv[a * 6 + b];
      jinv[i][b] = s;
    }
  /* joint limit cost function and gradient (:759-790) */
  double position_limit_cost = 0.0, velocity_limit_cost = 0.0;
  double pg[SHC_MAX_JOINTS], vg[SHC_MAX_JOINTS], cg[SHC_MAX_JOINTS];
  for (int i = 0; i < n; ++i)
  {
    const joint_t *jt = &leg->joint[i];
    pg[i] = 0.0;
    double joint_position_range = jt->max_position - jt->min_position;
    double position_range_centre = jt->min_position + joint_position_range / 2.0;
    if (joint_position_range != 0.0)
    {
      position_limit_cost +=
          orc_sqr(fabs(JOINT_LIMIT_COST_WEIGHT * (jt->desired_position - position_range_centre) / joint_position_range));
      pg[i] = -orc_sqr(JOINT_LIMIT_COST_WEIGHT) * (jt->desired_position - position_range_centre) /
              orc_sqr(joint_position_range);
    }
    double joint_velocity_range = 2 * jt->max_angular_speed;
    double velocity_range_centre = 0.0;
    velocity_limit_cost +=
        orc_sqr(fabs(JOINT_LIMIT_COST_WEIGHT * (jt->desired_velocity - velocity_range_centre) / joint_velocity_range));
    vg[i] = -orc_sqr(JOINT_LIMIT_COST_WEIGHT) * (jt->desired_velocity - velocity_range_centre) /
            orc_sqr(joint_velocity_range);
  }
  double ps = (position_limit_cost == 0.0 ? 0.0 : 1.0 / sqrt(position_limit_cost));
  double vs = (velocity_limit_cost == 0.0 ? 0.0 : 1.0 / sqrt(velocity_limit_cost));
  for (int i = 0; i < n; ++i)
  {
    pg[i] *= ps;
    vg[i] *= vs;
    cg[i] = (1.0 - 0.75) * pg[i] + 0.75 * vg[i]; /* interpolate(pg, vg, 0.75) */
  }
  /* return jinv * delta + (I - jinv * J) * cg  (:793-794) */
  for (int i = 0; i < n; ++i)
  {
    double s = 0.0;
    for (int b = 0; b < 6; ++b) s += jinv[i][b] * delta[b];
    double t = 0.0;
    for (int k = 0; k < n; ++k)
    {
      double jj = 0.0;
      for (int b = 0; b < 6; ++b) jj += jinv[i][b] * j[b][k];
      t += ((i == k ? 1.0 : 0.0) - jj) * cg[k];
    }
    dq[i] = s + t;
  }
}

/* Leg::updateJointPositions (model.cpp:799-857) */
static double leg_update_joint_positions(orc_robot *r, leg_t *leg, const double *delta, int simulation)
{
  double min_limit_proximity = 1.0;
  for (int i = 0; i < leg->joint_count; ++i)
  {
    joint_t *jt = &leg->joint[i];
    jt->desired_velocity = delta[i] / r->time_delta;
    if (r->params.clamp_joint_velocities && !simulation)
    {
      if (fabs(jt->desired_velocity) > jt->max_angular_speed)
      {
        double max_velocity = jt->max_angular_speed;
        jt->desired_velocity = orc_clamped(jt->desired_velocity, -max_velocity, max_velocity);
      }
    }
    jt->prev_desired_position = jt->desired_position;
    jt->desired_position = jt->prev_desired_position + jt->desired_velocity * r->time_delta;
    if (r->params.clamp_joint_positions)
    {
      if (jt->desired_position < jt->min_position) jt->desired_position = jt->min_position;
      else if (jt->desired_position > jt->max_position) jt->desired_position = jt->max_position;
    }
    double min_diff = fabs(jt->min_position - jt->desired_position);
    double max_diff = fabs(jt->max_position - jt->desired_position);
    double half_joint_range = (jt->max_position - jt->min_position) / 2.0;
    double limit_proximity = half_joint_range != 0 ? fmin(min_diff, max_diff) / half_joint_range : 1.0;
    min_limit_proximity = fmin(limit_proximity, min_limit_proximity);
  }
  return min_limit_proximity;
}

/* Leg::applyIK (model.cpp:861-941): one frame of it (the unconstrained retry is a nested frame) */
static double leg_apply_ik_frame(orc_robot *r, leg_t *leg, int simulation)
{
  orc_pose leg_frame_desired_tip_pose = joint_pose_joint_frame(leg, 1, leg->desired_tip_pose);
  orc_pose leg_frame_current_tip_pose = joint_pose_joint_frame(leg, 1, leg->current_tip_pose);
  orc_v3 position_delta = orc_v3_sub(leg_frame_desired_tip_pose.p, leg_frame_current_tip_pose.p);

  double delta[6] = { position_delta.x, position_delta.y, position_delta.z, 0, 0, 0 };
  double joint_position_delta[SHC_MAX_JOINTS];
  leg_solve_ik(leg, delta, 0, joint_position_delta);

  int rotation_constrained = !orc_quat_is_approx(leg->desired_tip_pose.r, ORC_UNDEFINED_ROTATION);
  if (rotation_constrained)
  {
    leg_update_joint_positions(r, leg, joint_position_delta, 1);
    leg_apply_fk(r, leg);
    orc_v3 desired_tip_direction = orc_quat_rotate(leg_frame_desired_tip_pose.r, orc_v3_make(1, 0, 0));
    orc_v3 current_tip_direction = orc_quat_rotate(leg_frame_current_tip_pose.r, orc_v3_make(1, 0, 0));
    orc_quat difference = orc_quat_from_two_vectors(current_tip_direction, desired_tip_direction);
    orc_v3 axis;
    double angle = orc_angle_axis_from_quat(orc_quat_normalized(difference), &axis);
    orc_v3 rotation_delta = orc_v3_scale(axis, angle);
    double delta2[6] = { 0, 0, 0, rotation_delta.x, rotation_delta.y, rotation_delta.z };
    leg_solve_ik(leg, delta2, 1, joint_position_delta);
  }

  double ik_success = leg_update_joint_positions(r, leg, joint_position_delta, simulation);
  leg_apply_fk(r, leg);

  for (int i = 0; i < 3; ++i)
  {
    orc_v3 position_error = orc_v3_sub(leg->current_tip_pose.p, leg->desired_tip_pose.p);
    double pe = i == 0 ? position_error.x : (i == 1 ? position_error.y : position_error.z);
    if (fabs(pe) > IK_TOLERANCE)
    {
      ik_success = 0.0;
      if (!simulation) leg->ik_failed = 1;
    }
  }

  if (rotation_constrained && !ik_success)
  {
    leg->desired_tip_pose.r = ORC_UNDEFINED_ROTATION;
    ik_success = leg_apply_ik_frame(r, leg, simulation);
  }

  leg_calculate_tip_force(r, leg);
  return ik_success;
}

/* ik_failed is this build's record of the reference's "Inverse kinematics deviation" warning (:921-928): raised by any frame of
 * the LAST non-simulated applyIK call of the leg. */
static double leg_apply_ik(orc_robot *r, leg_t *leg, int simulation)
{
  if (!simulation) leg->ik_failed = 0;
  return leg_apply_ik_frame(r, leg, simulation);
}

/* Model::estimateGravity (model.cpp:156-165) */
static orc_v3 model_estimate_gravity(const orc_robot *r)
{
  orc_v3 euler = orc_quat_to_euler(r->imu_orientation, 0); /* raw imu_data_.orientation (may be the zero sentinel) */
  orc_v3 gravity = orc_v3_make(0, 0, ORC_GRAVITY_ACCELERATION);
  gravity = orc_angle_axis_rotate(-euler.y, orc_v3_make(0, 1, 0), gravity);
  gravity = orc_angle_axis_rotate(-euler.x, orc_v3_make(1, 0, 0), gravity);
  return gravity;
}

/* Model::getImuData (model.h:132-140) */
static orc_quat model_imu_orientation(const orc_robot *r)
{
  if (orc_quat_is_approx(r->imu_orientation, ORC_UNDEFINED_ROTATION)) return orc_quat_identity();
  return r->imu_orientation;
}

/* Leg::generateWorkspace (model.cpp:309-510): the simple workspace (one plane at the stance height) or, in rough terrain mode,
 * the layered workspace: lower / upper vertical limit, then WORKSPACE_LAYERS planes searched from the top down. */
static void leg_generate_workspace(orc_robot *r, leg_t *leg)
{
  int simple_workspace = !r->params.rough_terrain_mode;
  double max_workplane[SHC_N_BEARINGS], min_workplane[SHC_N_BEARINGS];
  for (int b = 0; b < SHC_N_BEARINGS; ++b) { max_workplane[b] = MAX_WORKSPACE_RADIUS; min_workplane[b] = 0.0; }
  leg->workspace.n = 0;
  orc_pose current_pose = r->current_pose;
  orc_v3 identity_tip_position = orc_pose_inverse_transform_vector(current_pose, leg->stepper.identity_tip_pose.p);
  leg->workspace_zero = 0;
  if (orc_v3_norm(orc_v3_sub(identity_tip_position, leg->current_tip_pose.p)) > IK_TOLERANCE)
  {
    ws_insert(&leg->workspace, 0.0, min_workplane);
    leg->workspace_zero = 1;
    return;
  }
  if (simple_workspace) ws_insert(&leg->workspace, 0.0, max_workplane);

  int found_lower_limit = simple_workspace ? 1 : 0;
  int found_upper_limit = simple_workspace ? 1 : 0;
  double max_plane_height = simple_workspace ? 0.0 : MAX_WORKSPACE_RADIUS;
  double min_plane_height = simple_workspace ? 0.0 : -MAX_WORKSPACE_RADIUS;
  double search_height_delta = MAX_WORKSPACE_RADIUS / WORKSPACE_LAYERS;
  double search_height = 0.0;
  int search_bearing = 0;
  int within_limits = 1;
  int iteration = 1;
  orc_v3 origin_tip_position = orc_v3_make(0, 0, 0), target_tip_position = orc_v3_make(0, 0, 0);
  double distance_from_origin;
  int number_iterations = 1;
  int workspace_generation_complete = 0;
  (void)max_plane_height;

  while (1)
  {
    current_pose = r->current_pose;
    identity_tip_position = orc_pose_inverse_transform_vector(current_pose, leg->stepper.identity_tip_pose.p);
    identity_tip_position.z += search_height;

    if (iteration == 1)
    {
      within_limits = 1;
      leg_init(r, leg, 1);
      if (!found_lower_limit || !found_upper_limit)
      { /* search for the lower, then the upper vertical limit of the workspace */
        number_iterations = orc_round_to_int(MAX_WORKSPACE_RADIUS / MAX_POSITION_DELTA);
        origin_tip_position = identity_tip_position;
        target_tip_position = identity_tip_position;
        target_tip_position.z += found_lower_limit ? MAX_WORKSPACE_RADIUS : -MAX_WORKSPACE_RADIUS;
      }
      else if (search_bearing == 0)
      {
        number_iterations = orc_round_to_int(search_height_delta / MAX_POSITION_DELTA);
        number_iterations = number_iterations > 1 ? number_iterations : 1;
        origin_tip_position = leg->current_tip_pose.p;
        target_tip_position = identity_tip_position;
      }
      else
      {
        number_iterations = orc_round_to_int(MAX_WORKSPACE_RADIUS / MAX_POSITION_DELTA);
        origin_tip_position = identity_tip_position;
        target_tip_position = origin_tip_position;
        target_tip_position.x += MAX_WORKSPACE_RADIUS * cos(orc_deg2rad(search_bearing));
        target_tip_position.y += MAX_WORKSPACE_RADIUS * sin(orc_deg2rad(search_bearing));
      }
    }

    double i = (double)iteration / number_iterations;
    orc_v3 desired_tip_position = orc_v3_add(orc_v3_scale(origin_tip_position, 1.0 - i), orc_v3_scale(target_tip_position, i));
    leg_set_desired_tip_pose(leg, orc_pose_make(desired_tip_position, ORC_UNDEFINED_ROTATION), 1);
    double ik_result = leg_apply_ik(r, leg, 1);
    distance_from_origin = orc_v3_norm(orc_v3_sub(leg->current_tip_pose.p, identity_tip_position));
    within_limits = within_limits && ik_result != 0.0;

    if (within_limits && iteration < number_iterations)
    {
      iteration++;
    }
    else
    {
      iteration = 1;
      if (!found_lower_limit)
      {
        found_lower_limit = 1;
        min_plane_height = -distance_from_origin;
        ws_insert(&leg->workspace, min_plane_height, min_workplane);
        continue;
      }
      else if (!found_upper_limit)
      {
        found_upper_limit = 1;
        max_plane_height = distance_from_origin;
        search_height_delta = (max_plane_height - min_plane_height) / WORKSPACE_LAYERS;
        int upper_levels = (int)(fabs(max_plane_height) / search_height_delta);
        search_height = upper_levels * search_height_delta;
        ws_insert(&leg->workspace, max_plane_height, min_workplane);
        ws_insert(&leg->workspace, search_height, max_workplane);
        continue;
      }
      else if (search_bearing == 0) leg_update_default_configuration(leg);
      else ws_at(&leg->workspace, search_height)[search_bearing / BEARING_STEP] = distance_from_origin;

      if (search_bearing + BEARING_STEP <= 360)
      {
        search_bearing += BEARING_STEP;
      }
      else
      {
        search_bearing = 0;
        double *plane = ws_at(&leg->workspace, search_height);
        plane[0] = plane[360 / BEARING_STEP];
        search_height -= search_height_delta;
        if (search_height >= min_plane_height) ws_insert(&leg->workspace, search_height, max_workplane);
        else workspace_generation_complete = 1;
      }
    }
    if (workspace_generation_complete) return;
  }
}

/* ==================================================================================== WalkController */

/* WalkController::generateStepCycle (walk_controller.cpp:365-410) */
static shc_step_cycle generate_step_cycle(const shc_params *p)
{
  shc_step_cycle step;
  step.stance_end = (int)(p->stance_phase * 0.5);
  step.swing_start = step.stance_end;
  step.swing_end = step.swing_start + p->swing_phase;
  step.stance_start = step.swing_end;
  int base_step_period = p->stance_phase + p->swing_phase;
  double swing_ratio = (double)p->swing_phase / (double)base_step_period;
  double raw_step_period = ((1.0 / p->step_frequency) / p->time_delta) / swing_ratio;
  step.period = orc_round_to_even_int(raw_step_period / base_step_period) * base_step_period;
  step.frequency = 1.0 / (step.period * p->time_delta);
  int normaliser = step.period / base_step_period;
  step.stance_end *= normaliser;
  step.swing_start *= normaliser;
  step.swing_end *= normaliser;
  step.stance_start *= normaliser;
  step.stance_period = orc_mod(step.stance_end - step.stance_start, step.period);
  step.swing_period = step.swing_end - step.swing_start;
  return step;
}

/* LegStepper::LegStepper (walk_controller.cpp:795-819) */
static void stepper_construct(orc_robot *r, stepper_t *s, orc_pose identity_tip_pose)
{
  memset(s, 0, sizeof *s);
  s->identity_tip_pose = identity_tip_pose;
  s->default_tip_pose = identity_tip_pose;
  s->current_tip_pose = s->default_tip_pose;
  s->origin_tip_pose = s->current_tip_pose;
  s->target_tip_pose = s->default_tip_pose;
  s->walk_plane = orc_v3_make(0, 0, 0);
  s->walk_plane_normal = orc_v3_make(0, 0, 1);
  s->stride_vector = orc_v3_make(0, 0, 0);
  s->current_tip_velocity = orc_v3_make(0, 0, 0);
  s->swing_origin_tip_position = s->default_tip_pose.p;
  s->stance_origin_tip_position = s->default_tip_pose.p;
  s->swing_clearance = orc_v3_make(0.0, 0.0, r->params.swing_height);
  s->at_correct_phase = 0; s->completed_first_step = 0;
  s->phase = 0; s->phase_offset = 0;
  s->step_progress = 0.0; s->swing_progress = -1.0; s->stance_progress = -1.0; /* walk_controller.h:497-499 */
  s->step_state = STANCE;
  s->swing_origin_tip_velocity = orc_v3_make(0, 0, 0); /* uninitialised in the reference until first swing */
}

/* WalkController::init (walk_controller.cpp:22-53) */
static void walker_init(orc_robot *r)
{
  r->walk_state = STOPPED;
  r->pose_state = POSING_COMPLETE;
  r->walk_plane = orc_v3_make(0, 0, 0);
  r->walk_plane_normal = orc_v3_make(0, 0, 1);
  r->odometry_ideal = orc_pose_identity();
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    double x_position = r->params.stance_position[l][0];
    double y_position = r->params.stance_position[l][1];
    orc_quat identity_tip_rotation = ORC_UNDEFINED_ROTATION;
    if (leg->joint_count > 3 && r->params.gravity_aligned_tips)
    {
      identity_tip_rotation = orc_quat_from_two_vectors(orc_v3_make(1, 0, 0), orc_v3_make(-0.0, -0.0, -1));
      identity_tip_rotation = orc_correct_rotation(identity_tip_rotation, orc_quat_identity());
    }
    stepper_construct(r, &leg->stepper, orc_pose_make(orc_v3_make(x_position, y_position, 0.0), identity_tip_rotation));
  }
  r->desired_linear_velocity[0] = r->desired_linear_velocity[1] = 0;
  r->desired_angular_velocity = 0;
  r->legs_at_correct_phase = 0;
  r->legs_completed_first_step = 0;
  r->return_to_default_attempted = 0;
  r->step = generate_step_cycle(&r->params);
}

/* WalkController::generateLimits (walk_controller.cpp:231-361) for the given step cycle; a NULL map is not generated (adjustParameter asks for the two
 * speed maps only, :458-461).  The legs' phase offsets are set from `step` in every case (:277). */
static void walker_generate_limits_for(orc_robot *r, shc_step_cycle step, double *out_linear_speed, double *out_angular_speed,
                                       double *out_linear_acceleration, double *out_angular_acceleration)
{
  const shc_params *p = &r->params;
  int base_step_period = p->stance_phase + p->swing_phase;
  int normaliser = step.period / base_step_period;
  int base_step_offset = (int)(p->phase_offset * normaliser);

  int max_stance_extension = 0;
  for (int l = 0; l < r->leg_count; ++l)
  {
    int multiplier = p->offset_multiplier[l];
    int step_offset = (base_step_offset * multiplier) % step.period;
    r->leg[l].stepper.phase_offset = step_offset;
    if (step_offset > step.swing_start && step_offset < step.swing_end)
    {
      int ext = step.swing_end - step_offset;
      max_stance_extension = max_stance_extension > ext ? max_stance_extension : ext;
    }
  }
  double time_to_max_stride = (max_stance_extension + step.stance_period + step.swing_period) * r->time_delta;

  for (int b = 0; b < SHC_N_BEARINGS; ++b)
  {
    double walkspace_radius = r->walkspace[b];
    double on_ground_ratio = (double)step.stance_period / step.period;
    double max_speed = (walkspace_radius * 2.0) / (on_ground_ratio / step.frequency);
    double max_acceleration = max_speed / time_to_max_stride;

    double stance_overshoot = 0;
    for (int l = 0; l < r->leg_count; ++l)
    {
      double step_offset = r->leg[l].stepper.phase_offset;
      double t = step_offset * r->time_delta;
      double time_to_swing_end = time_to_max_stride - t;
      double v0 = max_acceleration * time_to_swing_end;
      double stride_length = v0 * (on_ground_ratio / step.frequency);
      double d0 = -stride_length / 2.0;
      double d1 = d0 + v0 * t + 0.5 * max_acceleration * orc_sqr(t);
      double d2 = max_speed * (step.stance_period * r->time_delta - t);
      stance_overshoot = fmax(stance_overshoot, d1 + d2 - walkspace_radius);
    }
    double swing_overshoot = 0.5 * max_speed * step.swing_period / (2.0 * step.period * step.frequency);
    double scaled_walkspace_radius =
        (walkspace_radius / (walkspace_radius + stance_overshoot + swing_overshoot)) * walkspace_radius;

    double x_position = r->leg[0].stepper.default_tip_pose.p.x;
    double y_position = r->leg[0].stepper.default_tip_pose.p.y;
    double stance_radius = sqrt(x_position * x_position + y_position * y_position);

    double max_linear_speed = (scaled_walkspace_radius * 2.0) / (on_ground_ratio / step.frequency);
    double max_linear_acceleration = max_linear_speed / time_to_max_stride;
    double max_angular_speed = max_linear_speed / stance_radius;
    double max_angular_acceleration = max_angular_speed / time_to_max_stride;
    if (walkspace_radius == 0.0)
    {
      max_linear_speed = 0.0;
      max_linear_acceleration = ORC_UNASSIGNED_VALUE;
      max_angular_speed = 0.0;
      max_angular_acceleration = ORC_UNASSIGNED_VALUE;
    }
    if (out_linear_speed) out_linear_speed[b] = max_linear_speed;
    if (out_linear_acceleration) out_linear_acceleration[b] = max_linear_acceleration;
    if (out_angular_speed) out_angular_speed[b] = max_angular_speed;
    if (out_angular_acceleration) out_angular_acceleration[b] = max_angular_acceleration;
  }
}
/* ... the set_limits path (no arguments: the walker's own step cycle, all four maps) */
static void walker_generate_limits(orc_robot *r)
{
  walker_generate_limits_for(r, r->step, r->max_linear_speed, r->max_angular_speed, r->max_linear_acceleration, r->max_angular_acceleration);
}

/* WalkController::generateWalkspace (walk_controller.cpp:57-227) */
static void walker_generate_walkspace(orc_robot *r)
{
  int have[SHC_N_BEARINGS];
  for (int b = 0; b < SHC_N_BEARINGS; ++b) have[b] = 0;
  int leg_count = r->leg_count;
  for (int l = 0; l < leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    leg_t *adj1 = &r->leg[orc_mod(l + 1, leg_count)];
    leg_t *adj2 = &r->leg[orc_mod(l - 1, leg_count)];
    orc_v3 dtp = leg->stepper.default_tip_pose.p;
    orc_v3 a1 = adj1->stepper.default_tip_pose.p;
    orc_v3 a2 = adj2->stepper.default_tip_pose.p;
    double distance_to_adjacent_leg_1 = orc_v3_norm(orc_v3_sub(dtp, a1)) / 2.0;
    double distance_to_adjacent_leg_2 = orc_v3_norm(orc_v3_sub(dtp, a2)) / 2.0;
    double bearing_to_adjacent_leg_1 = orc_rad2deg(atan2(a1.y - dtp.y, a1.x - dtp.x));
    double bearing_to_adjacent_leg_2 = orc_rad2deg(atan2(a2.y - dtp.y, a2.x - dtp.x));
    for (int bearing = 0; bearing <= 360; bearing += BEARING_STEP)
    {
      int bearing_diff_1 = abs(orc_mod((int)bearing_to_adjacent_leg_1, 360) - bearing);
      int bearing_diff_2 = abs(orc_mod((int)bearing_to_adjacent_leg_2, 360) - bearing);
      double distance_to_overlap_1 = ORC_UNASSIGNED_VALUE;
      double distance_to_overlap_2 = ORC_UNASSIGNED_VALUE;
      if ((bearing_diff_1 < 90 || bearing_diff_1 > 270) && distance_to_adjacent_leg_1 > 0.0)
        distance_to_overlap_1 = distance_to_adjacent_leg_1 / cos(orc_deg2rad(bearing_diff_1));
      if ((bearing_diff_2 < 90 || bearing_diff_2 > 270) && distance_to_adjacent_leg_2 > 0.0)
        distance_to_overlap_2 = distance_to_adjacent_leg_2 / cos(orc_deg2rad(bearing_diff_2));
      int overlapping = r->params.overlapping_walkspaces;
      double min_distance = overlapping ? MAX_WORKSPACE_RADIUS : fmin(distance_to_overlap_1, distance_to_overlap_2);
      min_distance = fmin(min_distance, MAX_WORKSPACE_RADIUS);
      int bi = bearing / BEARING_STEP;
      if (have[bi] && min_distance < r->walkspace[bi]) r->walkspace[bi] = min_distance;
      else if (!have[bi]) { r->walkspace[bi] = min_distance; have[bi] = 1; }
    }
  }

  for (int l = 0; l < leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    orc_pose current_pose = r->current_pose;
    orc_v3 identity_tip_position = orc_pose_inverse_transform_vector(current_pose, leg->stepper.identity_tip_pose.p);
    orc_v3 default_tip_position = orc_pose_inverse_transform_vector(current_pose, leg->stepper.default_tip_pose.p);
    orc_v3 default_shift = orc_v3_sub(default_tip_position, identity_tip_position);
    double target_workplane_height = default_shift.z;
    double workplane[SHC_N_BEARINGS]; /* Leg::getWorkplane (model.cpp:514-550): interpolated between the bounding planes */
    if (!ws_get_workplane(&leg->workspace, target_workplane_height, workplane)) continue; /* "undefined" workplane -> empty */

    for (int bi = 0; bi < SHC_N_BEARINGS; ++bi)
    {
      int bearing = bi * BEARING_STEP;
      double radius = r->walkspace[bi];
      if (orc_v3_norm(default_shift) == 0.0)
      {
        radius = workplane[bi];
      }
      else
      {
        /* AngleAxisd::_transformVector == toRotationMatrix() * v */
        orc_v3 new_point = orc_angle_axis_rotate(orc_deg2rad(bearing), orc_v3_make(0, 0, 1), orc_v3_make(MAX_WORKSPACE_RADIUS, 0, 0));
        new_point = orc_v3_set_precision(new_point, 3);
        for (int wi = 0; wi < SHC_N_BEARINGS; ++wi)
        {
          int bearing_1 = wi * BEARING_STEP;
          double radius_1 = workplane[wi];
          orc_v3 point_1 = orc_angle_axis_rotate(orc_deg2rad(bearing_1), orc_v3_make(0, 0, 1), orc_v3_make(radius_1, 0, 0));
          point_1 = orc_v3_sub(point_1, default_shift);
          point_1.z = 0.0;
          point_1 = orc_v3_set_precision(point_1, 3);
          if (bearing_1 == 360) { radius = 0.0; break; }
          int bearing_2 = (wi + 1) * BEARING_STEP;
          double radius_2 = workplane[wi + 1];
          orc_v3 point_2 = orc_angle_axis_rotate(orc_deg2rad(bearing_2), orc_v3_make(0, 0, 1), orc_v3_make(radius_2, 0, 0));
          point_2 = orc_v3_sub(point_2, default_shift);
          point_2.z = 0.0;
          point_2 = orc_v3_set_precision(point_2, 3);
          if (orc_v3_norm(orc_v3_cross(point_1, new_point)) == 0.0) { radius = orc_v3_norm(point_1); break; }
          else if (orc_v3_norm(orc_v3_cross(point_2, new_point)) == 0.0) { radius = orc_v3_norm(point_2); break; }
          else if (orc_v3_dot(orc_v3_cross(point_1, new_point), orc_v3_cross(point_1, point_2)) >= 0.0 &&
                   orc_v3_dot(orc_v3_cross(point_2, new_point), orc_v3_cross(point_2, point_1)) >= 0.0)
          {
            double dx = point_2.x - point_1.x;
            double dy = point_2.y - point_1.y;
            orc_v3 normal_1 = orc_v3_normalized(orc_v3_make(dy, -dx, 0.0));
            orc_v3 normal_2 = orc_v3_normalized(orc_v3_make(-dy, dx, 0.0));
            int same_direction_as_new_point = orc_v3_dot(orc_get_projection(new_point, normal_1), normal_1) >= 0.0;
            orc_v3 normal = same_direction_as_new_point ? normal_1 : normal_2;
            orc_v3 new_point_projection = orc_get_projection(new_point, normal);
            orc_v3 point_1_projection = orc_get_projection(point_1, normal);
            double ratio = orc_v3_norm(point_1_projection) / orc_v3_norm(new_point_projection);
            radius = ratio * MAX_WORKSPACE_RADIUS;
            break;
          }
        }
      }
      int opposite_bearing = orc_mod(bearing + 180, 360);
      if (radius < r->walkspace[bi])
      {
        r->walkspace[bi] = radius;
        r->walkspace[opposite_bearing / BEARING_STEP] = radius;
      }
    }
  }
  r->walkspace[360 / BEARING_STEP] = r->walkspace[0];
  walker_generate_limits(r);
}

/* WalkController::getLimit (walk_controller.cpp:414-436) */
static double walker_get_limit(const orc_robot *r, const double lin[2], double ang, const double *limit)
{
  double min_limit = ORC_UNASSIGNED_VALUE;
  for (int l = 0; l < r->leg_count; ++l)
  {
    orc_v3 tip_position = r->leg[l].stepper.current_tip_pose.p;
    double rn0 = -tip_position.y, rn1 = tip_position.x;
    double sv0 = lin[0] + ang * rn0, sv1 = lin[1] + ang * rn1;
    int bearing = orc_mod(orc_round_to_int(orc_rad2deg(atan2(sv1, sv0))), 360);
    int upper_bound = ((bearing + BEARING_STEP - 1) / BEARING_STEP) * BEARING_STEP; /* map::lower_bound(bearing)->first */
    int lower_bound = orc_mod(upper_bound - BEARING_STEP, 360);
    bearing += (bearing < lower_bound) ? 360 : 0;
    upper_bound += (upper_bound < lower_bound) ? 360 : 0;
    double control_input = (bearing - lower_bound) / (upper_bound - lower_bound); /* int / int (quirk 4) */
    double limit_interpolation =
        orc_interpolate(limit[lower_bound / BEARING_STEP], limit[orc_mod(upper_bound, 360) / BEARING_STEP], control_input);
    min_limit = fmin(min_limit, limit_interpolation);
  }
  return min_limit;
}

/* LegStepper::updateStepState (walk_controller.cpp:901-917) */
static void stepper_update_step_state(const orc_robot *r, stepper_t *s)
{
  const shc_step_cycle *step = &r->step;
  if (s->step_state == FORCE_STOP) return;
  else if (s->phase >= step->swing_start && s->phase < step->swing_end && s->step_state != FORCE_STANCE) s->step_state = SWING;
  else if (s->phase < step->stance_end || s->phase >= step->stance_start) s->step_state = STANCE;
}

/* LegStepper::iteratePhase (walk_controller.cpp:871-897) */
static void stepper_iterate_phase(const orc_robot *r, stepper_t *s)
{
  const shc_step_cycle *step = &r->step;
  s->phase = (s->phase + 1) % (step->period);
  stepper_update_step_state(r, s);
  s->step_progress = (double)s->phase / step->period;
  if (s->step_state == SWING)
  {
    s->swing_progress = (double)(s->phase - step->swing_start + 1) / (double)(step->swing_end - step->swing_start);
    s->swing_progress = orc_clamped(s->swing_progress, 0.0, 1.0);
    s->stance_progress = -1.0;
  }
  else if (s->step_state == STANCE)
  {
    s->stance_progress = (double)(orc_mod(s->phase + (step->period - step->stance_start), step->period) + 1) /
                         (double)(orc_mod(step->stance_end - step->stance_start, step->period));
    s->stance_progress = orc_clamped(s->stance_progress, 0.0, 1.0);
    s->swing_progress = -1.0;
  }
  else if (s->step_state == FORCE_STOP)
  {
    s->stance_progress = 0.0;
    s->swing_progress = -1.0;
  }
}

/* LegStepper::updateStride (walk_controller.cpp:921-945) */
static void stepper_update_stride(const orc_robot *r, stepper_t *s)
{
  s->walk_plane = r->walk_plane;
  s->walk_plane_normal = r->walk_plane_normal;
  orc_v3 stride_vector_linear = orc_v3_make(r->desired_linear_velocity[0], r->desired_linear_velocity[1], 0.0);
  orc_v3 radius = orc_get_rejection(s->current_tip_pose.p, orc_v3_make(0, 0, 1));
  orc_v3 angular_velocity = orc_v3_scale(orc_v3_make(0, 0, 1), r->desired_angular_velocity);
  orc_v3 stride_vector_angular = orc_v3_cross(angular_velocity, radius);
  s->stride_vector = orc_v3_add(stride_vector_linear, stride_vector_angular);
  double on_ground_ratio = (double)r->step.stance_period / r->step.period;
  s->stride_vector = orc_v3_scale(s->stride_vector, (on_ground_ratio / r->step.frequency));
  s->swing_clearance = orc_v3_scale(orc_v3_normalized(s->walk_plane_normal), r->params.swing_height);
}

/* LegStepper::calculateStanceSpanChange (walk_controller.cpp:949-980): single-plane workspace, or the layered one of rough terrain mode */
static orc_v3 stepper_calculate_stance_span_change(const orc_robot *r, const leg_t *leg)
{
  const stepper_t *s = &leg->stepper;
  double stance_span_modifier = r->params.stance_span_modifier;
  int positive_y_axis = (s->identity_tip_pose.p.y > 0.0); /* UnitY.dot(identity) > 0 */
  int bearing = (positive_y_axis ^ (stance_span_modifier > 0.0)) ? 270 : 90;
  stance_span_modifier *= (positive_y_axis ? 1.0 : -1.0);
  double radius = 0.0;
  if (leg->workspace.n == 1)
  {
    radius = leg->workspace.radius[0][bearing / BEARING_STEP]; /* workspace.at(0.0).at(bearing) */
  }
  else
  { /* interpolated between the planes bounding the default tip's height shift (:951-967) */
    double target_workplane_height = orc_set_precision(s->default_tip_pose.p.z - s->identity_tip_pose.p.z, 3);
    int lower, upper;
    ws_bounds(&leg->workspace, target_workplane_height, &lower, &upper);
    if (lower >= 0 && upper >= 0)
    {
      double upper_h = orc_set_precision(leg->workspace.height[upper], 3), lower_h = orc_set_precision(leg->workspace.height[lower], 3);
      double i = (target_workplane_height - lower_h) / (upper_h - lower_h);
      radius = leg->workspace.radius[lower][bearing / BEARING_STEP] * (1.0 - i) + leg->workspace.radius[upper][bearing / BEARING_STEP] * i;
    }
  }
  return orc_v3_make(0.0, radius * stance_span_modifier, 0.0);
}

static orc_pose pose_from7(const double *v)
{
  orc_pose o;
  o.p = orc_v3_make(v[0], v[1], v[2]);
  o.r.w = v[3], o.r.x = v[4], o.r.y = v[5], o.r.z = v[6];
  return o;
}

/* LegStepper::updateDefaultTipPosition (walk_controller.cpp:984-1014) */
static void stepper_update_default_tip_position(const orc_robot *r, leg_t *leg)
{
  stepper_t *s = &leg->stepper;
  if (s->external_default.defined)
  { /* new default from the external request, transformed by the robot's movement since the request (:988-990) */
    s->default_tip_pose = orc_pose_remove(pose_from7(s->external_default.pose), pose_from7(s->external_default.transform));
    return;
  }
  orc_v3 identity_tip_position = s->identity_tip_pose.p;
  identity_tip_position = orc_v3_add(identity_tip_position, stepper_calculate_stance_span_change(r, leg));
  identity_tip_position = orc_pose_transform_vector(r->default_pose, identity_tip_position); /* leg_->getDefaultBodyPose() */
  orc_v3 identity_to_stance_origin = orc_v3_sub(s->stance_origin_tip_position, identity_tip_position);
  orc_v3 projection_to_walk_plane = orc_get_projection(identity_to_stance_origin, s->walk_plane_normal);
  s->default_tip_pose = orc_pose_make(orc_v3_add(identity_tip_position, projection_to_walk_plane), ORC_UNDEFINED_ROTATION);
}

/* control node generators (walk_controller.cpp:1238-1329) */
static void stepper_generate_primary_swing_control_nodes(const orc_robot *r, stepper_t *s)
{
  orc_v3 mid_tip_position;
  mid_tip_position.x = (s->swing_origin_tip_position.x + s->target_tip_pose.p.x) / 2.0;
  mid_tip_position.y = (s->swing_origin_tip_position.y + s->target_tip_pose.p.y) / 2.0;
  mid_tip_position.z = fmax(s->swing_origin_tip_position.z, s->target_tip_pose.p.z);
  mid_tip_position = orc_v3_add(mid_tip_position, s->swing_clearance);
  double mid_lateral_shift = r->params.swing_width;
  int positive_y_axis = (s->identity_tip_pose.p.y > 0.0);
  mid_tip_position.y += positive_y_axis ? mid_lateral_shift : -mid_lateral_shift;
  orc_v3 stance_node_seperation = orc_v3_scale(orc_v3_scale(s->swing_origin_tip_velocity, 0.25), (r->time_delta / s->swing_delta_t));
  s->swing_1_nodes[0] = s->swing_origin_tip_position;
  s->swing_1_nodes[1] = orc_v3_add(s->swing_origin_tip_position, stance_node_seperation);
  s->swing_1_nodes[2] = orc_v3_add(s->swing_origin_tip_position, orc_v3_scale(stance_node_seperation, 2.0));
  s->swing_1_nodes[3] = orc_v3_make((mid_tip_position.x + s->swing_1_nodes[2].x) / 2.0,
                                    (mid_tip_position.y + s->swing_1_nodes[2].y) / 2.0,
                                    (mid_tip_position.z + s->swing_1_nodes[2].z) / 2.0);
  s->swing_1_nodes[3].z = mid_tip_position.z;
  s->swing_1_nodes[4] = mid_tip_position;
}

static void stepper_generate_secondary_swing_control_nodes(const orc_robot *r, stepper_t *s, int ground_contact)
{
  orc_v3 final_tip_velocity = orc_v3_scale(orc_v3_neg(s->stride_vector), (s->stance_delta_t / r->time_delta));
  orc_v3 stance_node_seperation = orc_v3_scale(orc_v3_scale(final_tip_velocity, 0.25), (r->time_delta / s->swing_delta_t));
  s->swing_2_nodes[0] = s->swing_1_nodes[4];
  s->swing_2_nodes[1] = orc_v3_sub(s->swing_1_nodes[4], orc_v3_sub(s->swing_1_nodes[3], s->swing_1_nodes[4]));
  s->swing_2_nodes[2] = orc_v3_sub(s->target_tip_pose.p, orc_v3_scale(stance_node_seperation, 2.0));
  s->swing_2_nodes[3] = orc_v3_sub(s->target_tip_pose.p, stance_node_seperation);
  s->swing_2_nodes[4] = s->target_tip_pose.p;
  if (ground_contact)
  {
    for (int k = 0; k < 5; ++k)
      s->swing_2_nodes[k] = orc_v3_add(s->current_tip_pose.p, orc_v3_scale(stance_node_seperation, (double)k));
  }
}

static void stepper_generate_stance_control_nodes(stepper_t *s, double stride_scaler)
{
  orc_v3 stance_node_seperation = orc_v3_scale(orc_v3_scale(orc_v3_neg(s->stride_vector), stride_scaler), 0.25);
  for (int k = 0; k < 5; ++k)
    s->stance_nodes[k] = orc_v3_add(s->stance_origin_tip_position, orc_v3_scale(stance_node_seperation, (double)k));
}

static void stepper_force_normal_touchdown(const orc_robot *r, stepper_t *s)
{
  orc_v3 final_tip_velocity = orc_v3_scale(orc_v3_neg(s->stride_vector), (s->stance_delta_t / r->time_delta));
  orc_v3 stance_node_seperation = orc_v3_scale(orc_v3_scale(final_tip_velocity, 0.25), (r->time_delta / s->swing_delta_t));
  orc_v3 bezier_target = s->target_tip_pose.p;
  orc_v3 bezier_origin = orc_v3_sub(s->target_tip_pose.p, orc_v3_scale(stance_node_seperation, 4.0));
  bezier_origin.z = fmax(s->swing_origin_tip_position.z, s->target_tip_pose.p.z);
  bezier_origin = orc_v3_add(bezier_origin, s->swing_clearance);
  s->swing_1_nodes[4] = bezier_origin;
  s->swing_2_nodes[0] = bezier_origin;
  s->swing_2_nodes[2] = orc_v3_sub(bezier_target, orc_v3_scale(stance_node_seperation, 2.0));
  orc_v3 half = orc_v3_make((s->swing_2_nodes[2].x - bezier_origin.x) / 2.0, (s->swing_2_nodes[2].y - bezier_origin.y) / 2.0,
                     (s->swing_2_nodes[2].z - bezier_origin.z) / 2.0);
  s->swing_1_nodes[3] = orc_v3_sub(s->swing_2_nodes[0], half);
  s->swing_2_nodes[1] = orc_v3_add(s->swing_2_nodes[0], half);
}

/* LegStepper::updateTipPosition (walk_controller.cpp:1018-1189); external targets (:1068-1079) are a planner node's API: not restated */
static void stepper_update_tip_position(const orc_robot *r, leg_t *leg)
{
  stepper_t *s = &leg->stepper;
  double time_delta = r->time_delta;
  const shc_step_cycle *step = &r->step;

  int standard_stance_period = (s->step_state == SWING || s->completed_first_step);
  int modified_stance_start = standard_stance_period ? step->stance_start : s->phase_offset;
  int modified_stance_period = orc_mod(step->stance_end - modified_stance_start, step->period);
  if (step->stance_end == modified_stance_start) modified_stance_period = step->period;

  int swing_iterations = (int)(((double)step->swing_period / step->period) / (step->frequency * time_delta));
  swing_iterations = orc_round_to_even_int(swing_iterations);
  s->swing_delta_t = 1.0 / (swing_iterations / 2.0);

  int stance_iterations = (int)(((double)modified_stance_period / step->period) / (step->frequency * time_delta));
  s->stance_delta_t = 1.0 / stance_iterations;

  s->target_tip_pose.p = orc_v3_add(s->default_tip_pose.p, orc_v3_scale(s->stride_vector, 0.5));

  if (s->step_state == SWING)
  {
    stepper_update_stride(r, s);
    int iteration = s->phase - step->swing_start + 1;
    int first_half = iteration <= swing_iterations / 2;
    int rough_terrain_mode = r->params.rough_terrain_mode;
    if (iteration == 1)
    {
      s->swing_origin_tip_position = s->current_tip_pose.p;
      s->swing_origin_tip_velocity = s->current_tip_velocity;
      if (rough_terrain_mode) stepper_update_default_tip_position(r, leg);
    }
    if (rough_terrain_mode && s->external_target.defined)
    { /* externally requested target, transformed by the movement since the request (:1068-1079) */
      s->target_tip_pose = orc_pose_remove(pose_from7(s->external_target.pose), pose_from7(s->external_target.transform));
      s->swing_clearance = orc_v3_scale(orc_v3_normalized(s->swing_clearance), s->external_target.swing_clearance);
      if (s->external_target.frame_is_odom_ideal)
      { /* lead to compensate for the moving target: WalkController::calculateOdometry(time_to_swing_end).position_ (:783-791) */
        double time_to_swing_end = (swing_iterations - iteration) * time_delta;
        orc_v3 desired_linear_velocity = orc_v3_make(r->desired_linear_velocity[0], r->desired_linear_velocity[1], 0);
        s->target_tip_pose.p = orc_v3_sub(s->target_tip_pose.p, orc_v3_scale(desired_linear_velocity, time_to_swing_end));
      }
    }
    else if (rough_terrain_mode && s->touchdown_detection)
    { /* update default target to meet the step surface proactively or reactively (:1081-1101) */
      orc_pose step_plane_pose = leg->step_plane_pose;
      if (orc_pose_ne(step_plane_pose, orc_pose_undefined()))
      {
        orc_v3 step_plane_position = orc_v3_sub(step_plane_pose.p, leg->current_tip_pose.p);
        orc_v3 target_tip_position = orc_v3_add(s->current_tip_pose.p, step_plane_position);
        orc_v3 difference = orc_v3_sub(target_tip_position, s->target_tip_pose.p);
        s->target_tip_pose.p = orc_v3_add(s->target_tip_pose.p, orc_get_projection(difference, s->walk_plane_normal));
      }
      else
      {
        s->target_tip_pose.p.z -= r->params.step_depth;
      }
    }
    int ground_contact = (orc_pose_ne(leg->step_plane_pose, orc_pose_undefined()) && rough_terrain_mode);
    stepper_generate_primary_swing_control_nodes(r, s);
    stepper_generate_secondary_swing_control_nodes(r, s, !first_half && ground_contact);
    if (r->params.force_normal_touchdown && !ground_contact) stepper_force_normal_touchdown(r, s);

    orc_v3 delta_pos;
    double time_input;
    if (first_half)
    {
      time_input = s->swing_delta_t * iteration;
      delta_pos = orc_v3_scale(orc_quartic_bezier_dot(s->swing_1_nodes, time_input), s->swing_delta_t);
    }
    else
    {
      time_input = s->swing_delta_t * (iteration - swing_iterations / 2);
      delta_pos = orc_v3_scale(orc_quartic_bezier_dot(s->swing_2_nodes, time_input), s->swing_delta_t);
    }
    s->current_tip_pose.p = orc_v3_add(s->current_tip_pose.p, delta_pos);
    s->current_tip_velocity = orc_v3_make(delta_pos.x / time_delta, delta_pos.y / time_delta, delta_pos.z / time_delta);
  }
  else if (s->step_state == STANCE || s->step_state == FORCE_STANCE)
  {
    stepper_update_stride(r, s);
    int iteration = orc_mod(s->phase + (step->period - modified_stance_start), step->period) + 1;
    if (iteration == 1)
    {
      s->stance_origin_tip_position = s->current_tip_pose.p;
      s->external_target.defined = 0; /* reset external target after every swing period (:1159) */
      if (r->params.rough_terrain_mode) stepper_update_default_tip_position(r, leg);
    }
    double stride_scaler = (double)modified_stance_period / (orc_mod(step->stance_end - step->stance_start, step->period));
    stepper_generate_stance_control_nodes(s, stride_scaler);
    double time_input = iteration * s->stance_delta_t;
    orc_v3 delta_pos = orc_v3_scale(orc_quartic_bezier_dot(s->stance_nodes, time_input), s->stance_delta_t);
    s->current_tip_pose.p = orc_v3_add(s->current_tip_pose.p, delta_pos);
    s->current_tip_velocity = orc_v3_make(delta_pos.x / time_delta, delta_pos.y / time_delta, delta_pos.z / time_delta);
  }
}

/* LegStepper::updateTipRotation (walk_controller.cpp:1193-1234) */
static void stepper_update_tip_rotation(const orc_robot *r, leg_t *leg)
{
  stepper_t *s = &leg->stepper;
  if (leg->joint_count > 3 && (s->stance_progress >= 0.0 || s->swing_progress >= 0.5))
  {
    if (r->params.gravity_aligned_tips && orc_quat_is_undefined(s->target_tip_pose.r))
      s->target_tip_pose.r = orc_quat_from_two_vectors(orc_v3_make(1, 0, 0), model_estimate_gravity(r));
    if (orc_quat_is_undefined(s->target_tip_pose.r))
    {
      s->current_tip_pose.r = s->target_tip_pose.r;
    }
    else
    {
      s->current_tip_pose.r = orc_correct_rotation(s->target_tip_pose.r, s->origin_tip_pose.r);
      if (s->swing_progress >= 0.5)
      {
        double c = orc_smooth_step(fmin(1.0, 2.0 * (s->swing_progress - 0.5)));
        orc_v3 origin_tip_direction = orc_quat_rotate(s->origin_tip_pose.r, orc_v3_make(1, 0, 0));
        orc_v3 target_tip_direction = orc_quat_rotate(s->target_tip_pose.r, orc_v3_make(1, 0, 0));
        orc_v3 new_tip_direction = orc_v3_lerp(origin_tip_direction, target_tip_direction, c);
        orc_quat new_tip_rotation = orc_quat_from_two_vectors(orc_v3_make(1, 0, 0), orc_v3_normalized(new_tip_direction));
        s->current_tip_pose.r = orc_correct_rotation(new_tip_rotation, s->current_tip_pose.r);
      }
    }
  }
  else
  {
    s->origin_tip_pose.r = leg->current_tip_pose.r;
    s->current_tip_pose.r = ORC_UNDEFINED_ROTATION;
  }
}

/* WalkController::updateWalkPlane (walk_controller.cpp:748-779) */
static void walker_update_walk_plane(orc_robot *r)
{
  int n = r->leg_count;
  if (n >= 3)
  {
    double ata[9] = { 0 }, atb[3] = { 0 };
    double A[SHC_MAX_LEGS][3], B[SHC_MAX_LEGS];
    for (int l = 0; l < n; ++l)
    {
      A[l][0] = r->leg[l].stepper.default_tip_pose.p.x;
      A[l][1] = r->leg[l].stepper.default_tip_pose.p.y;
      A[l][2] = 1.0;
      B[l] = r->leg[l].stepper.default_tip_pose.p.z;
    }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
      {
        double s = 0.0;
        for (int l = 0; l < n; ++l) s += A[l][i] * A[l][j];
        ata[i * 3 + j] = s;
      }
    double inv[9];
    orc_lu_inverse(ata, 3, inv);
    /* pseudo_inverse_A = (AtA)^-1 * At ; walk_plane = pinv * B */
    double pinv[3][SHC_MAX_LEGS];
    for (int i = 0; i < 3; ++i)
      for (int l = 0; l < n; ++l)
      {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += inv[i * 3 + k] * A[l][k];
        pinv[i][l] = s;
      }
    for (int i = 0; i < 3; ++i)
    {
      double s = 0.0;
      for (int l = 0; l < n; ++l) s += pinv[i][l] * B[l];
      atb[i] = s;
    }
    r->walk_plane = orc_v3_make(atb[0], atb[1], atb[2]);
    r->walk_plane_normal = orc_v3_normalized(orc_v3_make(-r->walk_plane.x, -r->walk_plane.y, 1.0));
  }
  else
  {
    r->walk_plane = orc_v3_make(0, 0, 0);
    r->walk_plane_normal = orc_v3_make(0, 0, 1);
  }
}

/* WalkController::calculateOdometry (walk_controller.cpp:783-791) */
static orc_pose walker_calculate_odometry(const orc_robot *r, double time_period)
{
  orc_v3 position_delta = orc_v3_scale(orc_v3_make(r->desired_linear_velocity[0], r->desired_linear_velocity[1], 0), time_period);
  orc_quat rotation_delta = orc_quat_from_angle_axis(r->desired_angular_velocity * time_period, orc_v3_make(0, 0, 1));
  return orc_pose_make(position_delta, rotation_delta);
}

/* WalkController::updateWalk (walk_controller.cpp:440-648) */
static void walker_update_walk(orc_robot *r, const double lin_in[2], double ang_in)
{
  double new_linear_velocity[2];
  double new_angular_velocity;
  double max_linear_speed = walker_get_limit(r, lin_in, ang_in, r->max_linear_speed);
  double max_angular_speed = walker_get_limit(r, lin_in, ang_in, r->max_angular_speed);
  double max_linear_acceleration = walker_get_limit(r, lin_in, ang_in, r->max_linear_acceleration);
  double max_angular_acceleration = walker_get_limit(r, lin_in, ang_in, r->max_angular_acceleration);
  double lin_norm = sqrt(lin_in[0] * lin_in[0] + lin_in[1] * lin_in[1]);

  if (r->walk_state != STOPPING)
  {
    if (r->params.velocity_input_mode == SHC_VEL_THROTTLE)
    {
      double cl[2] = { lin_in[0], lin_in[1] };
      if (lin_norm > 1.0) { cl[0] = lin_in[0] * (1.0 / lin_norm); cl[1] = lin_in[1] * (1.0 / lin_norm); }
      new_linear_velocity[0] = cl[0] * max_linear_speed;
      new_linear_velocity[1] = cl[1] * max_linear_speed;
      new_angular_velocity = orc_clamped(ang_in, -1.0, 1.0) * max_angular_speed;
      new_linear_velocity[0] *= (1.0 - fabs(ang_in));
      new_linear_velocity[1] *= (1.0 - fabs(ang_in));
    }
    else
    {
      double cl[2] = { lin_in[0], lin_in[1] };
      if (lin_norm > max_linear_speed) { cl[0] = lin_in[0] * (max_linear_speed / lin_norm); cl[1] = lin_in[1] * (max_linear_speed / lin_norm); }
      new_linear_velocity[0] = cl[0];
      new_linear_velocity[1] = cl[1];
      new_angular_velocity = orc_clamped(ang_in, -max_angular_speed, max_angular_speed);
      double sc = (max_angular_speed != 0.0 ? (1.0 - fabs(new_angular_velocity / max_angular_speed)) : 0.0);
      new_linear_velocity[0] *= sc;
      new_linear_velocity[1] *= sc;
    }
  }
  else
  {
    new_linear_velocity[0] = new_linear_velocity[1] = 0.0;
    new_angular_velocity = 0.0;
  }

  int has_velocity_command = (lin_norm != 0.0) || (ang_in != 0.0);

  /* check that all legs are in WALKING state (:492-505) */
  for (int l = 0; l < r->leg_count; ++l)
    if (r->leg[l].leg_state != WALKING) return;

  double la[2] = { new_linear_velocity[0] - r->desired_linear_velocity[0], new_linear_velocity[1] - r->desired_linear_velocity[1] };
  double la_norm = sqrt(la[0] * la[0] + la[1] * la[1]);
  if (la_norm < max_linear_acceleration * r->time_delta)
  {
    r->desired_linear_velocity[0] += la[0];
    r->desired_linear_velocity[1] += la[1];
  }
  else
  {
    double z = la[0] * la[0] + la[1] * la[1];
    double n0 = la[0], n1 = la[1];
    if (z > 0.0) { n0 = la[0] / sqrt(z); n1 = la[1] / sqrt(z); }
    r->desired_linear_velocity[0] += n0 * max_linear_acceleration * r->time_delta;
    r->desired_linear_velocity[1] += n1 * max_linear_acceleration * r->time_delta;
  }
  double angular_acceleration = new_angular_velocity - r->desired_angular_velocity;
  if (fabs(angular_acceleration) < max_angular_acceleration * r->time_delta)
    r->desired_angular_velocity += angular_acceleration;
  else
    r->desired_angular_velocity += orc_sign(angular_acceleration) * max_angular_acceleration * r->time_delta;

  int leg_count = r->leg_count;
  if (r->walk_state == STOPPED && has_velocity_command)
  {
    r->walk_state = STARTING;
    for (int l = 0; l < leg_count; ++l)
    {
      stepper_t *s = &r->leg[l].stepper;
      s->at_correct_phase = 0;
      s->completed_first_step = 0;
      s->step_state = STANCE;
      s->phase = s->phase_offset;
      stepper_update_step_state(r, s);
    }
    return;
  }
  else if (r->walk_state == STARTING && r->legs_at_correct_phase == leg_count && r->legs_completed_first_step == leg_count)
  {
    r->legs_at_correct_phase = 0;
    r->legs_completed_first_step = 0;
    r->walk_state = MOVING;
  }
  else if (r->walk_state == MOVING && !has_velocity_command)
  {
    r->walk_state = STOPPING;
  }
  else if (r->walk_state == STOPPING && r->legs_at_correct_phase == leg_count && r->pose_state == POSING_COMPLETE)
  {
    r->legs_at_correct_phase = 0;
    r->walk_state = STOPPED;
  }

  for (int l = 0; l < leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    stepper_t *s = &leg->stepper;
    if (r->walk_state == STARTING)
    {
      if (r->legs_at_correct_phase == leg_count)
      {
        if (s->phase == r->step.swing_end && !s->completed_first_step)
        {
          s->completed_first_step = 1;
          r->legs_completed_first_step++;
        }
      }
      if (!s->at_correct_phase)
      {
        if (s->phase_offset > r->step.swing_start && s->phase_offset < r->step.swing_end && s->phase != r->step.swing_end)
        {
          s->step_state = FORCE_STANCE;
        }
        else
        {
          r->legs_at_correct_phase++;
          s->at_correct_phase = 1;
        }
      }
    }
    else if (r->walk_state == MOVING)
    {
      s->at_correct_phase = 0;
    }
    else if (r->walk_state == STOPPING)
    {
      int zero_body_velocity = orc_v3_norm(s->stride_vector) == 0;
      orc_v3 error = orc_v3_sub(s->current_tip_pose.p, s->target_tip_pose.p);
      error = orc_get_rejection(error, s->walk_plane_normal);
      int at_target_tip_position = (orc_v3_norm(error) < TIP_TOLERANCE);
      if (zero_body_velocity && !s->at_correct_phase && s->phase == r->step.swing_end)
      {
        if (at_target_tip_position || r->return_to_default_attempted)
        {
          r->return_to_default_attempted = 0;
          stepper_update_default_tip_position(r, leg);
          s->step_state = FORCE_STOP;
          s->at_correct_phase = 1;
          r->legs_at_correct_phase++;
        }
        else
        {
          r->return_to_default_attempted = 1;
        }
      }
    }
    else if (r->walk_state == STOPPED)
    {
      s->step_state = FORCE_STOP;
      s->phase = 0;
    }
    /* leg state WALKING */
    stepper_update_tip_position(r, leg);
    stepper_update_tip_rotation(r, leg);
    stepper_iterate_phase(r, s);
  }
  walker_update_walk_plane(r);
  r->odometry_ideal = orc_pose_add(r->odometry_ideal, walker_calculate_odometry(r, r->time_delta));
}

/* ==================================================================================== PoseController */

/* PoseController::setAutoPoseParams (pose_controller.cpp:44-106) */
static void poser_set_auto_pose_params(orc_robot *r)
{
  const shc_params *p = &r->params;
  double raw_phase_length;
  int base_phase_length;
  r->pose_frequency = p->pose_frequency;
  if (r->pose_frequency == -1.0)
  {
    base_phase_length = p->stance_phase + p->swing_phase;
    double swing_ratio = (double)p->swing_phase / base_phase_length;
    raw_phase_length = ((1.0 / p->step_frequency) / p->time_delta) / swing_ratio;
  }
  else
  {
    base_phase_length = p->pose_phase_length;
    raw_phase_length = ((1.0 / r->pose_frequency) / p->time_delta);
  }
  r->pose_phase_length = orc_round_to_even_int(raw_phase_length / base_phase_length) * base_phase_length;
  r->normaliser = r->pose_phase_length / base_phase_length;
  r->auto_pose_reference_leg = 0;
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_poser_t *lp = &r->leg[l].poser;
    lp->pose_negation_phase_start = p->pose_negation_phase_starts[l];
    lp->pose_negation_phase_end = p->pose_negation_phase_ends[l];
    lp->negation_transition_ratio = p->negation_transition_ratio[l];
    if (p->offset_multiplier[l] == 0) r->auto_pose_reference_leg = l;
  }
  r->n_auto_posers = p->n_auto_posers;
  for (int i = 0; i < r->n_auto_posers; ++i)
  {
    auto_poser_t *ap = &r->auto_poser[i];
    memset(ap, 0, sizeof *ap);
    ap->start_phase = p->pose_phase_starts[i];
    ap->end_phase = p->pose_phase_ends[i];
    ap->x_amplitude = p->x_amplitudes[i];
    ap->y_amplitude = p->y_amplitudes[i];
    ap->z_amplitude = p->z_amplitudes[i];
    ap->gravity_amplitude = p->gravity_amplitudes[i];
    ap->roll_amplitude = p->roll_amplitudes[i];
    ap->pitch_amplitude = p->pitch_amplitudes[i];
    ap->yaw_amplitude = p->yaw_amplitudes[i];
    ap->start_check = 0; ap->end_check_first = 0; ap->end_check_second = 0; ap->allow_posing = 0;
  }
}

/* PoseController ctor + init (pose_controller.cpp:14-41) */
static void poser_init(orc_robot *r)
{
  r->manual_pose = r->auto_pose = r->imu_pose = r->inclination_pose = r->pc_default_pose = orc_pose_identity();
  r->walk_plane_pose = r->origin_walk_plane_pose = orc_pose_identity();
  r->tip_align_pose = r->origin_tip_align_pose = orc_pose_identity();
  r->rotation_absement_error = r->rotation_position_error = r->rotation_velocity_error = orc_v3_make(0, 0, 0);
  r->translation_velocity_input = r->rotation_velocity_input = orc_v3_make(0, 0, 0);
  r->pose_reset_mode = SHC_NO_RESET;
  r->executing_transition = 0;
  r->auto_posing_state = POSING_COMPLETE;
  r->pose_phase = 0;
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_poser_t *lp = &r->leg[l].poser;
    memset(lp, 0, sizeof *lp);
    lp->auto_pose = orc_pose_identity();
    lp->current_tip_pose = orc_pose_undefined();
    lp->target_tip_pose = orc_pose_undefined();
    lp->origin_tip_pose = orc_pose_identity(); /* default-constructed (unset) in the reference */
    lp->negate_auto_pose = 0;
    lp->first_iteration = 1;
    lp->master_iteration_count = 0;
    lp->has_desired_configuration = 0;
  }
  poser_set_auto_pose_params(r);
  r->walk_plane_pose.p = orc_v3_make(0.0, 0.0, r->params.body_clearance);
  r->origin_walk_plane_pose = r->walk_plane_pose;
}

/* PoseController::updateWalkPlanePose (pose_controller.cpp:1092-1130) */
static void poser_update_walk_plane_pose(orc_robot *r)
{
  orc_v3 walk_plane = orc_v3_make(0, 0, 0);
  orc_v3 walk_plane_normal = orc_v3_make(0, 0, 1);
  double c = 0.0;
  for (int l = 0; l < r->leg_count; ++l)
  {
    const stepper_t *s = &r->leg[l].stepper;
    double swing_progress_scaler = fmax(1.0, (double)r->params.swing_phase / r->params.phase_offset);
    double swing_progress = s->swing_progress * swing_progress_scaler;
    if (swing_progress >= 0 && swing_progress <= 1.0)
    {
      c = orc_smooth_step(swing_progress);
      walk_plane = s->walk_plane;
      walk_plane_normal = s->walk_plane_normal;
    }
  }
  orc_pose new_walk_plane_pose;
  new_walk_plane_pose.r = orc_quat_from_two_vectors(orc_v3_make(0, 0, 1), walk_plane_normal);
  new_walk_plane_pose.r = orc_correct_rotation(new_walk_plane_pose.r, orc_quat_identity());
  orc_v3 body_clearance = orc_v3_make(0, 0, r->params.body_clearance);
  new_walk_plane_pose.p = orc_quat_rotate(new_walk_plane_pose.r, body_clearance);
  new_walk_plane_pose.p.z += walk_plane.z;
  r->walk_plane_pose = orc_pose_interpolate(r->origin_walk_plane_pose, c, new_walk_plane_pose);
  if (c == 1.0) r->origin_walk_plane_pose = r->walk_plane_pose;
}

/* PoseController::updateManualPose (pose_controller.cpp:863-1003) */
static void poser_update_manual_pose(orc_robot *r)
{
  const shc_params *p = &r->params;
  double time_delta = p->time_delta;
  double current_position[3] = { r->manual_pose.p.x, r->manual_pose.p.y, r->manual_pose.p.z };
  orc_v3 cr = orc_quat_to_euler(r->manual_pose.r, 1);
  double current_rotation[3] = { cr.x, cr.y, cr.z };
  double default_position[3] = { r->pc_default_pose.p.x, r->pc_default_pose.p.y, r->pc_default_pose.p.z };
  orc_v3 dr = orc_quat_to_euler(r->pc_default_pose.r, 1);
  double default_rotation[3] = { dr.x, dr.y, dr.z };
  const double *max_position = p->max_translation;
  const double *max_rotation = p->max_rotation;
  double translation_limit[3] = { 0, 0, 0 }, rotation_limit[3] = { 0, 0, 0 };
  double translation_velocity[3] = { 0, 0, 0 }, rotation_velocity[3] = { 0, 0, 0 };
  double desired_position[3] = { 0, 0, 0 }, desired_rotation[3] = { 0, 0, 0 };
  double tvi[3] = { r->translation_velocity_input.x, r->translation_velocity_input.y, r->translation_velocity_input.z };
  double rvi[3] = { r->rotation_velocity_input.x, r->rotation_velocity_input.y, r->rotation_velocity_input.z };

  int reset_translation[3] = { 0, 0, 0 }, reset_rotation[3] = { 0, 0, 0 };
  switch (r->pose_reset_mode)
  {
    case SHC_Z_AND_YAW_RESET: reset_translation[2] = 1; reset_rotation[2] = 1; break;
    case SHC_X_AND_Y_RESET: reset_translation[0] = 1; reset_translation[1] = 1; break;
    case SHC_PITCH_AND_ROLL_RESET: reset_rotation[0] = 1; reset_rotation[1] = 1; break;
    case SHC_ALL_RESET:
      reset_translation[0] = reset_translation[1] = reset_translation[2] = 1;
      reset_rotation[0] = reset_rotation[1] = reset_rotation[2] = 1;
      break;
    case SHC_IMMEDIATE_ALL_RESET: r->manual_pose = r->pc_default_pose; return;
    default: break;
  }
  for (int i = 0; i < 3; i++)
  {
    if (reset_translation[i])
    {
      double diff = current_position[i] - default_position[i];
      if (diff < 0) tvi[i] = 1.0; else if (diff > 0) tvi[i] = -1.0;
    }
    if (reset_rotation[i])
    {
      double diff = current_rotation[i] - default_rotation[i];
      if (diff < 0) rvi[i] = 1.0; else if (diff > 0) rvi[i] = -1.0;
    }
    translation_velocity[i] = tvi[i] * p->max_translation_velocity;
    rotation_velocity[i] = rvi[i] * p->max_rotation_velocity;
    desired_position[i] = current_position[i] + translation_velocity[i] * time_delta;
    desired_rotation[i] = current_rotation[i] + rotation_velocity[i] * time_delta;

    translation_limit[i] = orc_sign(translation_velocity[i]) * max_position[i];
    if (reset_translation[i] && default_position[i] < max_position[i] && default_position[i] > -max_position[i])
      translation_limit[i] = default_position[i];
    int positive_translation_velocity = orc_sign(translation_velocity[i]) > 0;
    int exceeds_positive_translation_limit = positive_translation_velocity && desired_position[i] > translation_limit[i];
    int exceeds_negative_translation_limit = !positive_translation_velocity && desired_position[i] < translation_limit[i];
    if (exceeds_positive_translation_limit || exceeds_negative_translation_limit)
      translation_velocity[i] = (translation_limit[i] - current_position[i]) / time_delta;

    rotation_limit[i] = orc_sign(rotation_velocity[i]) * max_rotation[i];
    if (reset_rotation[i] && default_rotation[i] < max_rotation[i] && default_rotation[i] > -max_rotation[i])
      rotation_limit[i] = default_rotation[i];
    int positive_rotation_velocity = orc_sign(rotation_velocity[i]) > 0;
    int exceeds_positive_rotation_limit = positive_rotation_velocity && desired_rotation[i] > rotation_limit[i];
    int exceeds_negative_rotation_limit = !positive_rotation_velocity && desired_rotation[i] < rotation_limit[i];
    if (exceeds_positive_rotation_limit || exceeds_negative_rotation_limit)
      rotation_velocity[i] = (rotation_limit[i] - current_rotation[i]) / time_delta;

    desired_position[i] = current_position[i] + translation_velocity[i] * time_delta;
    desired_rotation[i] = current_rotation[i] + rotation_velocity[i] * time_delta;
  }
  /* the reset modes write back into the member inputs (translation_velocity_input_[i] = ...) */
  r->translation_velocity_input = orc_v3_make(tvi[0], tvi[1], tvi[2]);
  r->rotation_velocity_input = orc_v3_make(rvi[0], rvi[1], rvi[2]);
  r->manual_pose.p = orc_v3_make(desired_position[0], desired_position[1], desired_position[2]);
  r->manual_pose.r = orc_correct_rotation(
      orc_euler_to_quat(orc_v3_make(desired_rotation[0], desired_rotation[1], desired_rotation[2]), 1), orc_quat_identity());
}

/* PoseController::updateInclinationPose (pose_controller.cpp:1240-1259) */
static void poser_update_inclination_pose(orc_robot *r)
{
  orc_quat compensation_combined = orc_quat_normalized(orc_quat_mul(r->manual_pose.r, r->auto_pose.r));
  orc_quat compensation_removed =
      orc_quat_normalized(orc_quat_mul(model_imu_orientation(r), orc_quat_inverse(compensation_combined)));
  orc_v3 euler = orc_quat_to_euler(compensation_removed, 0);
  double body_height = r->params.body_clearance;
  double longitudinal_correction = -body_height * tan(euler.y);
  double lateral_correction = body_height * tan(euler.x);
  double max_translation_x = r->params.max_translation[0];
  double max_translation_y = r->params.max_translation[1];
  longitudinal_correction = orc_clamped(longitudinal_correction, -max_translation_x, max_translation_x);
  lateral_correction = orc_clamped(lateral_correction, -max_translation_y, max_translation_y);
  r->inclination_pose.p.x = longitudinal_correction;
  r->inclination_pose.p.y = lateral_correction;
}

/* PoseController::updateIMUPose (pose_controller.cpp:1191-1236) */
static void poser_update_imu_pose(orc_robot *r)
{
  orc_quat current_rotation = orc_correct_rotation(model_imu_orientation(r), orc_quat_identity());
  orc_quat target_rotation = orc_correct_rotation(r->manual_pose.r, orc_quat_identity());
  orc_quat rotation_error = orc_quat_normalized(orc_quat_mul(current_rotation, orc_quat_inverse(target_rotation)));
  double kp = r->params.rotation_pid_gains[0];
  double ki = r->params.rotation_pid_gains[1];
  double kd = r->params.rotation_pid_gains[2];
  r->rotation_position_error = orc_quat_to_euler(rotation_error, 0);
  r->rotation_position_error.z = 0.0;
  if (orc_v3_norm(r->rotation_position_error) < IMU_POSING_DEADBAND) return;
  r->rotation_absement_error = orc_v3_add(r->rotation_absement_error, orc_v3_scale(r->rotation_position_error, r->params.time_delta));
  double smoothing_factor = 0.15;
  r->rotation_velocity_error = orc_v3_add(orc_v3_scale(orc_v3_neg(r->imu_angular_velocity), smoothing_factor),
                                          orc_v3_scale(r->rotation_velocity_error, (1 - smoothing_factor)));
  orc_v3 rotation_correction = orc_v3_neg(orc_v3_add(orc_v3_add(orc_v3_scale(r->rotation_velocity_error, kd),
                                                                orc_v3_scale(r->rotation_position_error, kp)),
                                                     orc_v3_scale(r->rotation_absement_error, ki)));
  double max_roll = r->params.max_rotation[0];
  double max_pitch = r->params.max_rotation[1];
  rotation_correction.x = orc_clamped(rotation_correction.x, -max_roll, max_roll);
  rotation_correction.y = orc_clamped(rotation_correction.y, -max_pitch, max_pitch);
  rotation_correction.z = orc_quat_to_euler(target_rotation, 0).z;
  if (orc_v3_norm(rotation_correction) > STABILITY_THRESHOLD) r->unstable = 1;
  r->imu_pose.r = orc_euler_to_quat(rotation_correction, 0);
  r->imu_pose.r = orc_correct_rotation(r->imu_pose.r, target_rotation);
}

/* AutoPoser::updatePose (pose_controller.cpp:1338-1439) */
static orc_pose auto_poser_update_pose(orc_robot *r, auto_poser_t *ap, int phase)
{
  orc_pose return_pose = orc_pose_identity();
  int start_phase = ap->start_phase * r->normaliser;
  int end_phase = ap->end_phase * r->normaliser;
  if (start_phase > end_phase)
  {
    end_phase += r->pose_phase_length;
    if (phase < start_phase) phase += r->pose_phase_length;
  }
  int state = r->auto_posing_state;
  int sync_with_step_cycle = (r->pose_frequency == -1.0);
  ap->start_check = !sync_with_step_cycle || (!ap->start_check && state == POSING && phase == start_phase);
  ap->end_check_first = (ap->end_check_first || (state == STOP_POSING && phase == start_phase));
  ap->end_check_second = (ap->end_check_second || (state == STOP_POSING && phase == end_phase && ap->end_check_first));
  if (!ap->allow_posing && ap->start_check)
  {
    ap->allow_posing = 1;
    ap->end_check_first = 0; ap->end_check_second = 0;
  }
  else if (ap->allow_posing && sync_with_step_cycle && ap->end_check_first && ap->end_check_second)
  {
    ap->allow_posing = 0;
    ap->start_check = 0;
  }
  if (phase >= start_phase && phase < end_phase && ap->allow_posing)
  {
    int iteration = phase - start_phase + 1;
    int num_iterations = end_phase - start_phase;
    orc_v3 zero = orc_v3_make(0, 0, 0);
    orc_v3 position_control_nodes[5] = { zero, zero, zero, zero, zero };
    orc_v3 rotation_control_nodes[5] = { zero, zero, zero, zero, zero };
    int first_half = iteration <= num_iterations / 2;
    orc_v3 gravity_direction = orc_v3_normalized(model_estimate_gravity(r));
    orc_v3 rot_amp = orc_v3_make(ap->roll_amplitude, ap->pitch_amplitude, ap->yaw_amplitude);
    orc_v3 pos_amp = (ap->gravity_amplitude != 0.0) ? orc_v3_scale(gravity_direction, ap->gravity_amplitude)
                                                    : orc_v3_make(ap->x_amplitude, ap->y_amplitude, ap->z_amplitude);
    if (first_half)
    {
      rotation_control_nodes[3] = rot_amp; rotation_control_nodes[4] = rot_amp;
      position_control_nodes[3] = pos_amp; position_control_nodes[4] = pos_amp;
    }
    else
    {
      rotation_control_nodes[0] = rot_amp; rotation_control_nodes[1] = rot_amp;
      position_control_nodes[0] = pos_amp; position_control_nodes[1] = pos_amp;
    }
    double delta_t = 1.0 / (num_iterations / 2.0);
    int offset = (int)((first_half ? 0 : num_iterations / 2.0));
    double time_input = (iteration - offset) * delta_t;
    orc_v3 position = orc_quartic_bezier(position_control_nodes, time_input);
    orc_v3 rotation = orc_quartic_bezier(rotation_control_nodes, time_input);
    return_pose = orc_pose_make(position, orc_euler_to_quat(rotation, 0));
  }
  return return_pose;
}

/* LegPoser::updateAutoPose (pose_controller.cpp:1716-1778) */
static void leg_poser_update_auto_pose(orc_robot *r, leg_t *leg, int phase)
{
  leg_poser_t *lp = &leg->poser;
  int start_phase = lp->pose_negation_phase_start * r->normaliser;
  int end_phase = lp->pose_negation_phase_end * r->normaliser;
  int negation_phase = phase;
  if (start_phase == 0) start_phase = r->pose_phase_length;
  if (end_phase == 0) end_phase = r->pose_phase_length;
  if (start_phase > end_phase)
  {
    end_phase += r->pose_phase_length;
    if (negation_phase < start_phase) negation_phase += r->pose_phase_length;
  }
  int step_state = leg->stepper.step_state;
  if (step_state != FORCE_STANCE && step_state != FORCE_STOP && negation_phase == start_phase) lp->negate_auto_pose = 1;
  if (negation_phase < start_phase || negation_phase > end_phase) lp->negate_auto_pose = 0;
  lp->auto_pose = r->auto_pose;
  if (lp->negate_auto_pose)
  {
    int iteration = negation_phase - start_phase + 1;
    int num_iterations = end_phase - start_phase;
    int first_half = iteration <= num_iterations / 2;
    double control_input = 1.0;
    if (lp->negation_transition_ratio > 0.0)
    {
      if (first_half) control_input = fmin(1.0, iteration / (num_iterations * lp->negation_transition_ratio));
      else control_input = fmin(1.0, (num_iterations - iteration) / (num_iterations * lp->negation_transition_ratio));
    }
    control_input = orc_smooth_step(control_input);
    orc_pose negation = orc_pose_interpolate(orc_pose_identity(), control_input, lp->auto_pose);
    lp->auto_pose = orc_pose_remove(lp->auto_pose, negation);
  }
}

/* PoseController::updateAutoPose (pose_controller.cpp:1134-1187) */
static void poser_update_auto_pose(orc_robot *r)
{
  const stepper_t *ls = &r->leg[r->auto_pose_reference_leg].stepper;
  r->auto_pose = orc_pose_identity();
  int zero_body_velocity = orc_v3_norm(ls->stride_vector) == 0;
  if (r->walk_state == STARTING || r->walk_state == MOVING) r->auto_posing_state = POSING;
  else if ((zero_body_velocity && r->walk_state == STOPPING) || r->walk_state == STOPPED) r->auto_posing_state = STOP_POSING;
  int master_phase;
  int sync_with_step_cycle = (r->pose_frequency == -1.0);
  if (sync_with_step_cycle) master_phase = ls->phase;
  else
  {
    master_phase = r->pose_phase;
    r->pose_phase = (r->pose_phase + 1) % r->pose_phase_length;
  }
  int auto_posers_complete = 0;
  for (int i = 0; i < r->n_auto_posers; ++i)
  {
    orc_pose updated_pose = auto_poser_update_pose(r, &r->auto_poser[i], master_phase);
    auto_posers_complete += (int)(!r->auto_poser[i].allow_posing);
    r->auto_pose = orc_pose_add(r->auto_pose, updated_pose);
  }
  if (auto_posers_complete == r->n_auto_posers) r->auto_posing_state = POSING_COMPLETE;
  for (int l = 0; l < r->leg_count; ++l) leg_poser_update_auto_pose(r, &r->leg[l], master_phase);
}

/* PoseController::updateTipAlignPose (pose_controller.cpp:1024-1088) */
static void poser_update_tip_align_pose(orc_robot *r)
{
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    const stepper_t *ls = &leg->stepper;
    double swing_progress = ls->swing_progress;
    if (swing_progress != -1.0)
    {
      orc_v3 walk_plane_normal = ls->walk_plane_normal;
      orc_quat walk_plane_rotation = orc_quat_from_two_vectors(orc_v3_make(0, 0, 1), walk_plane_normal);
      /* vector from the tip to the last joint, robot frame (Tip / Joint::getPoseRobotFrame, model.h:603-617, 684-697) */
      orc_m4 tt = transform_from_joint(leg, leg->joint_count + 1, 0);
      orc_m4 tj = transform_from_joint(leg, leg->joint_count, 0);
      orc_v3 tip_position = orc_v3_make(tt.m[0][3], tt.m[1][3], tt.m[2][3]);
      orc_v3 joint_position = orc_v3_make(tj.m[0][3], tj.m[1][3], tj.m[2][3]);
      orc_v3 tip_to_joint = orc_v3_sub(joint_position, tip_position);
      double link_length = orc_v3_norm(orc_v3_sub(tip_position, joint_position));
      /* body translation that puts the joint in line with the tip along the walk-plane normal */
      orc_v3 a = orc_quat_rotate(walk_plane_rotation, tip_to_joint);
      orc_v3 b = orc_v3_scale(walk_plane_normal, link_length);
      orc_v3 rejection = orc_v3_sub(a, orc_v3_scale(b, orc_v3_dot(a, b) / orc_v3_dot(b, b)));
      orc_v3 translation_to_alignment = orc_v3_neg(rejection);
      a = r->tip_align_pose.p;
      b = walk_plane_normal;
      rejection = orc_v3_sub(a, orc_v3_scale(b, orc_v3_dot(a, b) / orc_v3_dot(b, b)));
      orc_v3 target_translation = orc_v3_add(rejection, translation_to_alignment);
      /* clamped(value, limit): every upper bound is limit[1] (standard_includes.h:134) */
      const double *lim = r->params.max_translation;
      target_translation.x = orc_clamped(target_translation.x, -lim[0], lim[1]);
      target_translation.y = orc_clamped(target_translation.y, -lim[1], lim[1]);
      target_translation.z = orc_clamped(target_translation.z, -lim[2], lim[1]);
      double c = orc_smooth_step(swing_progress);
      if (swing_progress < 0.5)
      {
        c = orc_smooth_step(c * 2.0);
        r->tip_align_pose = orc_pose_interpolate(r->origin_tip_align_pose, c, orc_pose_identity());
      }
      else if (swing_progress >= 0.5)
      {
        c = orc_smooth_step((c - 0.5) * 2.0);
        r->tip_align_pose = orc_pose_interpolate(orc_pose_identity(), c, orc_pose_make(target_translation, orc_quat_identity()));
      }
      if (swing_progress == 1.0) r->origin_tip_align_pose = r->tip_align_pose;
    }
  }
}

/* PoseController::updateCurrentPose (pose_controller.cpp:811-859) */
static void poser_update_current_pose(orc_robot *r, int robot_state)
{
  orc_pose new_pose = orc_pose_identity();
  poser_update_walk_plane_pose(r);
  new_pose = orc_pose_add(new_pose, r->walk_plane_pose);
  r->default_pose = r->walk_plane_pose; /* model_->setDefaultPose */
  if (r->params.manual_posing)
  {
    poser_update_manual_pose(r);
    new_pose = orc_pose_add(new_pose, r->manual_pose);
  }
  if (r->params.inclination_posing)
  {
    poser_update_inclination_pose(r);
    new_pose = orc_pose_add(new_pose, r->inclination_pose);
  }
  if (r->params.imu_posing && robot_state == RS_RUNNING)
  {
    poser_update_imu_pose(r);
    new_pose = orc_pose_add(new_pose, r->imu_pose);
  }
  else if (r->params.auto_posing)
  {
    poser_update_auto_pose(r);
    new_pose = orc_pose_add(new_pose, r->auto_pose);
  }
  /* automatic body posing to align tips orthogonal to the walk plane during the 2nd half of swing ("TODO EXPERIMENTAL", :849-855) */
  if (r->params.gravity_aligned_tips && r->leg[0].joint_count <= 3)
  {
    poser_update_tip_align_pose(r);
    new_pose = orc_pose_add(new_pose, r->tip_align_pose);
  }
  r->current_pose = new_pose;
}

/* PoseController::updateStance (pose_controller.cpp:110-141) */
static void poser_update_stance(orc_robot *r)
{
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    orc_pose current_pose = r->current_pose;
    if (leg->leg_state == MANUAL || leg->leg_state == WALKING_TO_MANUAL)
    { /* do not apply any posing to manually manipulated legs (:135-139) */
      leg->poser.current_tip_pose = leg->stepper.current_tip_pose;
      continue;
    }
    current_pose = orc_pose_remove(current_pose, r->auto_pose);
    current_pose = orc_pose_add(current_pose, leg->poser.auto_pose);
    orc_v3 new_tip_position = orc_pose_inverse_transform_vector(current_pose, leg->stepper.current_tip_pose.p);
    orc_quat new_tip_rotation = orc_quat_mul(orc_quat_inverse(current_pose.r), leg->stepper.current_tip_pose.r);
    leg->poser.current_tip_pose = orc_pose_make(new_tip_position, new_tip_rotation);
  }
}

/* LegPoser::stepToPosition (pose_controller.cpp:1571-1712) */
static int leg_poser_step_to_position(orc_robot *r, leg_t *leg, orc_pose target_tip_pose, orc_pose target_pose,
                                      double lift_height, double time_to_step, int apply_delta)
{
  leg_poser_t *lp = &leg->poser;
  if (lp->first_iteration)
  {
    lp->origin_tip_pose = leg->current_tip_pose;
    lp->master_iteration_count = 0;
    lp->first_iteration = 0;
  }
  orc_pose desired_tip_pose = target_tip_pose;
  if (orc_pose_eq(desired_tip_pose, orc_pose_undefined()))
  {
    desired_tip_pose = lp->origin_tip_pose;
    desired_tip_pose.r = ORC_UNDEFINED_ROTATION;
  }
  orc_v3 position_delta = orc_v3_sub(lp->origin_tip_pose.p, orc_pose_inverse_transform_vector(target_pose, desired_tip_pose.p));
  int transition_position = orc_v3_norm(position_delta) > TIP_TOLERANCE;
  int transition_rotation = 0;
  if (!orc_quat_is_undefined(desired_tip_pose.r))
  {
    orc_v3 origin_tip_direction = orc_quat_rotate(lp->origin_tip_pose.r, orc_v3_make(1, 0, 0));
    orc_v3 desired_tip_direction = orc_quat_rotate(desired_tip_pose.r, orc_v3_make(1, 0, 0));
    orc_v3 ax;
    double ang = orc_angle_axis_from_quat(orc_quat_from_two_vectors(origin_tip_direction, desired_tip_direction), &ax);
    transition_rotation = ang > JOINT_TOLERANCE;
  }
  if (!transition_position && !transition_rotation && lift_height == 0.0)
  {
    lp->first_iteration = 1;
    lp->current_tip_pose = lp->origin_tip_pose;
    return PROGRESS_COMPLETE;
  }
  /* "Apply delta z to target tip position": not to manually manipulated legs (:1610-1614) */
  int manually_manipulated = (leg->leg_state == MANUAL || leg->leg_state == WALKING_TO_MANUAL);
  if (apply_delta && !manually_manipulated) desired_tip_pose.p = orc_v3_add(desired_tip_pose.p, leg->admittance_delta);
  lp->master_iteration_count++;
  int num_iterations = orc_round_to_int(time_to_step / r->params.time_delta);
  num_iterations = num_iterations > 1 ? num_iterations : 1;
  double delta_t = 1.0 / num_iterations;
  double completion_ratio = ((double)(lp->master_iteration_count - 1) / (double)num_iterations);
  orc_pose desired_pose = orc_pose_interpolate(orc_pose_identity(), orc_smooth_step(completion_ratio), target_pose);
  orc_quat new_tip_rotation = ORC_UNDEFINED_ROTATION;
  if (!orc_quat_is_undefined(desired_tip_pose.r))
  {
    orc_v3 origin_tip_direction = orc_quat_rotate(lp->origin_tip_pose.r, orc_v3_make(1, 0, 0));
    orc_v3 desired_tip_direction = orc_quat_rotate(desired_tip_pose.r, orc_v3_make(1, 0, 0));
    orc_v3 new_tip_direction = orc_v3_lerp(origin_tip_direction, desired_tip_direction, orc_smooth_step(completion_ratio));
    new_tip_rotation = orc_quat_from_two_vectors(orc_v3_make(1, 0, 0), orc_v3_normalized(new_tip_direction));
  }
  double time_input;
  orc_v3 new_tip_position = lp->origin_tip_pose.p;
  if (!(desired_tip_pose.p.x == (double)INT_MAX && desired_tip_pose.p.y == (double)INT_MAX && desired_tip_pose.p.z == (double)INT_MAX))
  {
    int half_swing_iteration = num_iterations / 2;
    orc_v3 cp[5], cs[5];
    orc_v3 origin_to_target = orc_v3_sub(lp->origin_tip_pose.p, desired_tip_pose.p);
    cp[0] = lp->origin_tip_pose.p;
    cp[1] = lp->origin_tip_pose.p;
    cp[2] = lp->origin_tip_pose.p;
    cp[3] = orc_v3_add(desired_tip_pose.p, orc_v3_scale(origin_to_target, 0.75));
    cp[4] = orc_v3_add(desired_tip_pose.p, orc_v3_scale(origin_to_target, 0.5));
    cp[2].z += lift_height; cp[3].z += lift_height; cp[4].z += lift_height;
    cs[0] = orc_v3_add(desired_tip_pose.p, orc_v3_scale(origin_to_target, 0.5));
    cs[1] = orc_v3_add(desired_tip_pose.p, orc_v3_scale(origin_to_target, 0.25));
    cs[2] = desired_tip_pose.p;
    cs[3] = desired_tip_pose.p;
    cs[4] = desired_tip_pose.p;
    cs[0].z += lift_height; cs[1].z += lift_height; cs[2].z += lift_height;
    int swing_iteration_count = (lp->master_iteration_count + (num_iterations - 1)) % (num_iterations) + 1;
    if (swing_iteration_count <= half_swing_iteration)
    {
      time_input = swing_iteration_count * delta_t * 2.0;
      new_tip_position = orc_quartic_bezier(cp, time_input);
    }
    else
    {
      time_input = (swing_iteration_count - half_swing_iteration) * delta_t * 2.0;
      new_tip_position = orc_quartic_bezier(cs, time_input);
    }
  }
  if (leg->leg_state != MANUAL) /* a MANUAL leg keeps the tip pose updateStance gave its LegPoser (:1680-1684) */
  {
    lp->current_tip_pose.p = orc_pose_inverse_transform_vector(desired_pose, new_tip_position);
    lp->current_tip_pose.r = new_tip_rotation;
  }
  if (lp->master_iteration_count >= num_iterations)
  {
    lp->first_iteration = 1;
    return PROGRESS_COMPLETE;
  }
  return (int)(completion_ratio * PROGRESS_COMPLETE);
}

/* LegPoser::transitionConfiguration (pose_controller.cpp:1476-1567) */
static int leg_poser_transition_configuration(orc_robot *r, leg_t *leg, double transition_time)
{
  leg_poser_t *lp = &leg->poser;
  if (!lp->has_desired_configuration) return PROGRESS_COMPLETE;
  if (lp->first_iteration)
  {
    for (int i = 0; i < leg->joint_count; ++i) lp->origin_configuration[i] = leg->joint[i].desired_position;
    lp->first_iteration = 0;
    lp->master_iteration_count = 0;
  }
  int num_iterations = orc_round_to_int(transition_time / r->params.time_delta);
  num_iterations = num_iterations > 1 ? num_iterations : 1;
  double delta_t = 1.0 / num_iterations;
  lp->master_iteration_count++;
  for (int i = 0; i < leg->joint_count; ++i)
  {
    joint_t *jt = &leg->joint[i];
    double control_nodes[4] = { lp->origin_configuration[i], lp->origin_configuration[i], lp->desired_configuration[i],
                                lp->desired_configuration[i] };
    jt->prev_desired_position = jt->desired_position;
    jt->desired_position = orc_cubic_bezier_scalar(control_nodes, lp->master_iteration_count * delta_t);
  }
  leg_apply_fk(r, leg);
  int progress = (int)(((double)(lp->master_iteration_count - 1) / (double)num_iterations) * PROGRESS_COMPLETE);
  progress = progress < 1 ? 1 : (progress > PROGRESS_COMPLETE ? PROGRESS_COMPLETE : progress);
  if (lp->master_iteration_count >= num_iterations)
  {
    lp->first_iteration = 1;
    return PROGRESS_COMPLETE;
  }
  return progress;
}

/* PoseController::directStartup (pose_controller.cpp:463-517) */
static int poser_direct_startup(orc_robot *r)
{
  int progress = 0;
  double time_to_start = r->params.time_to_start;
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    if (!r->executing_transition)
    {
      leg_t *test_leg = (leg_t *)malloc(sizeof(leg_t));
      *test_leg = *leg; /* Leg(leg) + generate(leg): copies joints, tip, stepper, poser */
      leg_init(r, test_leg, 1);
      orc_pose default_tip_pose = leg->stepper.default_tip_pose;
      int guard = 0;
      while (progress != PROGRESS_COMPLETE && guard++ < 1000000)
      {
        progress = leg_poser_step_to_position(r, test_leg, default_tip_pose, r->current_pose, 0.0, time_to_start, 1);
        leg_set_desired_tip_pose(test_leg, test_leg->poser.current_tip_pose, 1);
        leg_apply_ik(r, test_leg, 1);
      }
      for (int j = 0; j < leg->joint_count; ++j) leg->poser.desired_configuration[j] = test_leg->joint[j].desired_position;
      leg->poser.has_desired_configuration = 1;
      free(test_leg);
    }
    progress = leg_poser_transition_configuration(r, leg, time_to_start);
  }
  r->executing_transition = (progress != 0 && progress != PROGRESS_COMPLETE);
  return progress;
}

/* Model::legsBearingLoad (model.cpp:78-88) */
static int model_legs_bearing_load(const orc_robot *r)
{
  double body_height_estimate = 0.0;
  for (int l = 0; l < r->leg_count; ++l) body_height_estimate += r->leg[l].current_tip_pose.p.z;
  return -(body_height_estimate / r->leg_count) > HALF_BODY_DEPTH;
}

static void leg_poser_add_transition_pose(leg_poser_t *lp, orc_pose pose)
{
  if (lp->n_transition_poses < ORC_MAX_TRANSITION_POSES) lp->transition_poses[lp->n_transition_poses++] = pose;
}

/* PoseController::executeSequence (pose_controller.cpp:145-459).  sequence: SEQ_START_UP / SEQ_SHUT_DOWN.  Returns the
 * reference's progress (-1 while the first START_UP execution is generating the sequence, 100 = complete). */
static int poser_execute_sequence(orc_robot *r, int sequence)
{
  /* Initialise / reset any saved transition sequence (:149-162) */
  if (r->reset_transition_sequence && sequence == SEQ_START_UP)
  {
    r->reset_transition_sequence = 0;
    r->first_sequence_execution = 1;
    r->transition_step = 0;
    for (int l = 0; l < r->leg_count; ++l)
    {
      leg_t *leg = &r->leg[l];
      leg->poser.n_transition_poses = 0;                               /* resetTransitionSequence */
      leg_poser_add_transition_pose(&leg->poser, leg->current_tip_pose); /* initial transition position */
    }
  }
  int progress = 0;
  int normalised_progress = 0;
  int next_transition_step = 0, transition_step_target = 0, execute_horizontal_transition = 0, execute_vertical_transition = 0, total_progress = 0;
  int count_or_one = r->transition_step_count > 1 ? r->transition_step_count : 1; /* std::max(transition_step_count_, 1) */
  if (sequence == SEQ_START_UP)
  {
    execute_horizontal_transition = !(r->transition_step % 2);
    execute_vertical_transition = r->transition_step % 2;
    next_transition_step = r->transition_step + 1;
    transition_step_target = r->transition_step_count;
    total_progress = r->transition_step * 100 / count_or_one;
  }
  else
  {
    execute_horizontal_transition = r->transition_step % 2;
    execute_vertical_transition = !(r->transition_step % 2);
    next_transition_step = r->transition_step - 1;
    transition_step_target = 0;
    total_progress = 100 - r->transition_step * 100 / count_or_one;
  }
  int final_transition;
  int sequence_complete = 0;
  if (r->first_sequence_execution) final_transition = (r->horizontal_transition_complete || r->vertical_transition_complete);
  else final_transition = (next_transition_step == transition_step_target);

  double safety_factor = (r->first_sequence_execution ? SAFETY_FACTOR / (r->transition_step + 1) : 0.0);
  double step_frequency = r->params.step_frequency;

  if (execute_horizontal_transition)
  {
    if (r->set_target)
    {
      r->set_target = 0;
      for (int l = 0; l < r->leg_count; ++l)
      {
        leg_t *leg = &r->leg[l];
        leg->poser.leg_completed_step = 0;
        orc_v3 target_tip_position;
        if (next_transition_step >= 0 && leg->poser.n_transition_poses > next_transition_step) /* hasTransitionPose (a negative index - SHUT_DOWN before any START_UP - is undefined behaviour in the reference) */
          target_tip_position = leg->poser.transition_poses[next_transition_step].p;
        else
          target_tip_position = orc_pose_inverse_transform_vector(r->current_pose, leg->stepper.default_tip_pose.p);
        target_tip_position.z = leg->current_tip_pose.p.z; /* maintain horizontal position */
        leg->poser.target_tip_pose = orc_pose_make(target_tip_position, leg->stepper.target_tip_pose.r);
      }
    }
    int direct_step = !model_legs_bearing_load(r);
    for (int l = 0; l < r->leg_count; ++l)
    {
      leg_t *leg = &r->leg[l];
      leg_poser_t *lp = &leg->poser;
      if (!lp->leg_completed_step)
      {
        if ((leg->id_number % 2) == r->current_group || direct_step) /* Leg::group_ = id_number % 2 (model.cpp:187) */
        {
          orc_pose target_tip_pose = lp->target_tip_pose;
          int apply_delta = (sequence == SEQ_START_UP && final_transition);
          double step_height = direct_step ? 0.0 : r->params.swing_height;
          double time_to_step = HORIZONTAL_TRANSITION_TIME / step_frequency;
          time_to_step *= (r->first_sequence_execution ? 2.0 : 1.0);
          progress = leg_poser_step_to_position(r, leg, target_tip_pose, orc_pose_identity(), step_height, time_to_step, apply_delta);
          leg_set_desired_tip_pose(leg, lp->current_tip_pose, 1);
          double limit_proximity = leg_apply_ik(r, leg, 0);
          int exceeded_workspace = limit_proximity < safety_factor;
          if (r->first_sequence_execution && exceeded_workspace)
          {
            lp->target_tip_pose = lp->current_tip_pose;
            lp->first_iteration = 1; /* resetStepToPosition */
            progress = PROGRESS_COMPLETE;
            r->proximity_alert = 1;
          }
          if (progress == PROGRESS_COMPLETE)
          {
            lp->leg_completed_step = 1;
            r->legs_completed_step++;
            if (r->first_sequence_execution)
            {
              int reached_target = !exceeded_workspace;
              leg_poser_add_transition_pose(lp, reached_target ? lp->target_tip_pose : lp->current_tip_pose);
            }
          }
        }
        else
        {
          r->legs_completed_step++;
          lp->leg_completed_step = 1;
        }
      }
    }
    count_or_one = r->transition_step_count > 1 ? r->transition_step_count : 1;
    if (direct_step) normalised_progress = progress / count_or_one;
    else normalised_progress = (progress / 2 + (r->current_group == 0 ? 0 : 50)) / count_or_one;
    if (r->legs_completed_step == r->leg_count)
    {
      r->set_target = 1;
      r->legs_completed_step = 0;
      if (r->current_group == 1 || direct_step)
      {
        r->current_group = 0;
        r->transition_step = next_transition_step;
        r->horizontal_transition_complete = !r->proximity_alert;
        sequence_complete = final_transition;
        r->proximity_alert = 0;
      }
      else if (r->current_group == 0)
      {
        r->current_group = 1;
      }
    }
  }

  if (execute_vertical_transition)
  {
    if (r->set_target)
    {
      r->set_target = 0;
      for (int l = 0; l < r->leg_count; ++l)
      {
        leg_t *leg = &r->leg[l];
        orc_v3 target_tip_position;
        if (next_transition_step >= 0 && leg->poser.n_transition_poses > next_transition_step)
          target_tip_position = leg->poser.transition_poses[next_transition_step].p;
        else
          target_tip_position = orc_pose_inverse_transform_vector(r->current_pose, leg->stepper.default_tip_pose.p);
        target_tip_position.x = leg->current_tip_pose.p.x; /* maintain horizontal position */
        target_tip_position.y = leg->current_tip_pose.p.y;
        leg->poser.target_tip_pose = orc_pose_make(target_tip_position, leg->stepper.target_tip_pose.r);
      }
    }
    int all_legs_within_workspace = 1;
    for (int l = 0; l < r->leg_count; ++l)
    {
      leg_t *leg = &r->leg[l];
      leg_poser_t *lp = &leg->poser;
      int apply_delta = (sequence == SEQ_START_UP && final_transition);
      double time_to_step = VERTICAL_TRANSITION_TIME / step_frequency;
      time_to_step *= (r->first_sequence_execution ? 2.0 : 1.0);
      progress = leg_poser_step_to_position(r, leg, lp->target_tip_pose, orc_pose_identity(), 0.0, time_to_step, apply_delta);
      leg_set_desired_tip_pose(leg, lp->current_tip_pose, 0);
      double limit_proximity = leg_apply_ik(r, leg, 0);
      all_legs_within_workspace = all_legs_within_workspace && !(limit_proximity < safety_factor);
    }
    if ((!all_legs_within_workspace && r->first_sequence_execution) || progress == PROGRESS_COMPLETE)
    {
      for (int l = 0; l < r->leg_count; ++l)
      {
        leg_poser_t *lp = &r->leg[l].poser;
        lp->first_iteration = 1; /* resetStepToPosition */
        progress = PROGRESS_COMPLETE;
        if (r->first_sequence_execution)
          leg_poser_add_transition_pose(lp, all_legs_within_workspace ? lp->target_tip_pose : lp->current_tip_pose);
      }
      r->vertical_transition_complete = all_legs_within_workspace;
      r->transition_step = next_transition_step;
      sequence_complete = final_transition;
      r->set_target = 1;
    }
    count_or_one = r->transition_step_count > 1 ? r->transition_step_count : 1;
    normalised_progress = progress / count_or_one;
  }

  if (r->first_sequence_execution)
  {
    r->transition_step_count = r->transition_step;
    transition_step_target = r->transition_step;
  }
  if (r->transition_step > TRANSITION_STEP_THRESHOLD) r->sequence_failed = 1; /* ROS_FATAL + ros::shutdown() (:436-440) */

  if (sequence_complete)
  {
    r->set_target = 1;
    r->vertical_transition_complete = 0;
    r->horizontal_transition_complete = 0;
    r->first_sequence_execution = 0;
    return PROGRESS_COMPLETE;
  }
  total_progress = total_progress + normalised_progress;
  if (total_progress > PROGRESS_COMPLETE - 1) total_progress = PROGRESS_COMPLETE - 1;
  return r->first_sequence_execution ? -1 : total_progress;
}

/* PoseController::stepToNewStance (pose_controller.cpp:521-557), tripod leg coordination */
static int poser_step_to_new_stance(orc_robot *r)
{
  int progress = 0;
  int leg_count = r->leg_count;
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    if ((leg->id_number % 2) == r->current_group)
    {
      double step_height = r->params.swing_height;
      double step_time = 1.0 / r->params.step_frequency;
      orc_pose target_tip_pose = leg->stepper.default_tip_pose;
      progress = leg_poser_step_to_position(r, leg, target_tip_pose, r->current_pose, step_height, step_time, 1);
      leg_set_desired_tip_pose(leg, leg->poser.current_tip_pose, 1);
      leg_apply_ik(r, leg, 0);
      r->legs_completed_step += (progress == PROGRESS_COMPLETE);
    }
  }
  progress = progress / 2 + r->current_group * 50;
  r->current_group = r->legs_completed_step / (leg_count / 2);
  if (r->legs_completed_step == leg_count)
  {
    r->legs_completed_step = 0;
    r->current_group = 0;
  }
  r->reset_transition_sequence = 1;
  return progress;
}

/* PoseController::packLegs / unpackLegs (pose_controller.cpp:615-707), simultaneous leg coordination.  packed_positions =
 * Joint::packed_positions_ ([number_pack_steps][legs][dof] flattened; default.yaml "packed" is a list per joint). */
static int poser_pack_legs(orc_robot *r, const double *packed_positions, int number_pack_steps, double time_to_pack, int unpack)
{
  int progress = 0;
  if (!unpack) r->transition_step = 0; /* reset for the start-up / shut-down sequences (:618) */
  int k = 0, dof_total = 0;
  for (int l = 0; l < r->leg_count; ++l) dof_total += r->leg[l].joint_count;
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    if (!r->executing_transition)
    {
      for (int j = 0; j < leg->joint_count; ++j)
      {
        double target;
        if (!unpack) target = packed_positions[(size_t)r->pack_step * dof_total + k + j];
        else target = (r->pack_step > 0) ? packed_positions[(size_t)(r->pack_step - 1) * dof_total + k + j] : leg->joint[j].unpacked_position;
        leg->poser.desired_configuration[j] = target;
      }
      leg->poser.has_desired_configuration = 1;
    }
    k += leg->joint_count;
    progress = leg_poser_transition_configuration(r, leg, time_to_pack);
  }
  r->executing_transition = (progress != 0 && progress != PROGRESS_COMPLETE);
  if (!unpack)
  {
    if (progress == PROGRESS_COMPLETE && r->pack_step < number_pack_steps - 1)
    {
      r->executing_transition = 0;
      r->pack_step++;
      progress = 0;
    }
  }
  else if (progress == PROGRESS_COMPLETE && r->pack_step != 0)
  {
    r->executing_transition = 0;
    r->pack_step--;
    progress = 0;
  }
  return progress;
}

/* ==================================================================================== planner mode */

/* PoseController::transitionConfiguration (pose_controller.cpp:710-763): every leg the target_configuration message names moves
 * to its joint positions on LegPoser::transitionConfiguration's cubic Bezier; a leg it does not name gets an empty desired
 * configuration and reports completion at once (:1479-1482). */
static int poser_transition_configuration(orc_robot *r, double transition_time)
{
  int min_progress = INT_MAX;
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    if (!r->executing_transition)
    {
      leg->poser.has_desired_configuration = r->target_configuration_named[l];
      for (int j = 0; j < leg->joint_count; ++j) leg->poser.desired_configuration[j] = r->target_configuration[l][j];
    }
    int progress = leg_poser_transition_configuration(r, leg, transition_time);
    min_progress = progress < min_progress ? progress : min_progress;
  }
  r->executing_transition = (min_progress != 0 && min_progress != PROGRESS_COMPLETE);
  return min_progress;
}

/* PoseController::transitionStance (pose_controller.cpp:767-807): every leg steps (no lift unless the request carries a swing
 * clearance) to the planner's tip target - or stays where it is - while the body eases to target_body_pose_. */
static int poser_transition_stance(orc_robot *r, double transition_time)
{
  int min_progress = INT_MAX;
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    shc_external_target *target = &leg->poser.external_target;
    orc_pose target_tip_pose = orc_pose_undefined();
    double swing_clearance = 0.0;
    if (target->defined)
    {
      target_tip_pose = orc_pose_add(pose_from7(target->transform), pose_from7(target->pose));
      swing_clearance = target->swing_clearance;
    }
    if (orc_quat_is_undefined(target_tip_pose.r) && r->params.gravity_aligned_tips)
      target_tip_pose.r = orc_quat_from_two_vectors(orc_v3_make(1, 0, 0), model_estimate_gravity(r));
    int progress = leg_poser_step_to_position(r, leg, target_tip_pose, r->target_body_pose, swing_clearance, transition_time, 1);
    leg_set_desired_tip_pose(leg, leg->poser.current_tip_pose, 1);
    leg_apply_ik(r, leg, 0);
    min_progress = progress < min_progress ? progress : min_progress;
    if (target->defined && progress == PROGRESS_COMPLETE) target->defined = 0;
  }
  return min_progress;
}

static void model_update_model(orc_robot *r);

/* StateController::executePlan (state_controller.cpp:653-698).  Returns the progress of the plan step being executed, -2 while
 * it waits for the planner to publish one (the node then republishes its request for step plan_step_), -1 while the robot is
 * still walking (velocity inputs forced to zero; the loop goes on with its normal cycle). */
static int state_execute_plan(orc_robot *r)
{
  if (r->walk_state != STOPPED)
  {
    r->linear_velocity_input[0] = r->linear_velocity_input[1] = 0.0;
    r->angular_velocity_input = 0.0;
    return -1;
  }
  if (!r->target_configuration_acquired && !r->target_tip_pose_acquired && !r->target_body_pose_acquired)
  {
    model_update_model(r);
    return -2;
  }
  int progress = r->target_configuration_acquired ? poser_transition_configuration(r, 5.0) : poser_transition_stance(r, 5.0);
  if (progress == PROGRESS_COMPLETE)
  {
    r->plan_step++;
    r->target_body_pose = orc_pose_identity();
    r->target_configuration_acquired = r->target_tip_pose_acquired = r->target_body_pose_acquired = 0;
  }
  return progress;
}

/* ==================================================================================== manual leg manipulation */

/* WalkController::updateManual, tip-velocity overload (walk_controller.cpp:652-708).  A MANUAL leg that is neither the primary
 * nor the secondary selection reads an uninitialised Eigen vector in the reference; here its input is zero. */
static void walker_update_manual_velocity(orc_robot *r)
{
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    if (leg->leg_state != MANUAL) continue;
    orc_v3 tip_velocity_input = orc_v3_make(0, 0, 0);
    if (leg->id_number == r->primary_leg_selection) tip_velocity_input = r->primary_tip_velocity_input;
    else if (leg->id_number == r->secondary_leg_selection) tip_velocity_input = r->secondary_tip_velocity_input;
    if (orc_v3_norm(tip_velocity_input) == 0.0) continue;
    if (r->params.leg_manipulation_mode == SHC_MANIPULATION_JOINT_CONTROL && leg->joint_count == 3)
    { /* x / y velocity inputs move the tibia / coxa joints (:677-690) */
      double coxa_joint_velocity = tip_velocity_input.y * r->params.max_rotation_velocity * r->time_delta;
      double tibia_joint_velocity = tip_velocity_input.x * r->params.max_rotation_velocity * r->time_delta;
      leg->joint[0].prev_desired_position = leg->joint[0].desired_position;
      leg->joint[2].prev_desired_position = leg->joint[2].desired_position;
      leg->joint[0].desired_position += coxa_joint_velocity;
      leg->joint[2].desired_position += tibia_joint_velocity;
      /* Leg::applyFK(false) (model.cpp:945-988): the joint transforms follow the new joint positions, current_tip_pose_ /
       * current_tip_velocity_ are left alone */
      orc_pose keep_pose = leg->current_tip_pose;
      orc_v3 keep_velocity = leg->current_tip_velocity;
      leg->stepper.current_tip_pose = leg_apply_fk(r, leg);
      leg->current_tip_pose = keep_pose;
      leg->current_tip_velocity = keep_velocity;
    }
    else if (r->params.leg_manipulation_mode == SHC_MANIPULATION_TIP_CONTROL)
    {
      orc_v3 ik_error = orc_v3_sub(leg->desired_tip_pose.p, leg->current_tip_pose.p);
      orc_v3 tip_position_change = orc_v3_scale(tip_velocity_input, r->params.max_translation_velocity * r->time_delta);
      if (orc_v3_norm(ik_error) >= IK_TOLERANCE)
        tip_position_change = orc_v3_scale(orc_v3_neg(orc_v3_normalized(ik_error)), orc_v3_norm(tip_position_change));
      leg->stepper.current_tip_pose = orc_pose_make(orc_v3_add(leg->stepper.current_tip_pose.p, tip_position_change), ORC_UNDEFINED_ROTATION);
    }
  }
}

/* WalkController::updateManual, tip-pose overload (walk_controller.cpp:712-744) */
static void walker_update_manual_pose(orc_robot *r)
{
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    if (leg->leg_state != MANUAL) continue;
    orc_v3 tip_position_input = orc_v3_make(0, 0, 0);
    if (leg->id_number == r->primary_leg_selection) tip_position_input = r->primary_pose_input.p;
    else if (leg->id_number == r->secondary_leg_selection) tip_position_input = r->secondary_pose_input.p;
    if (orc_v3_norm(tip_position_input) != 0.0 && r->params.leg_manipulation_mode == SHC_MANIPULATION_TIP_CONTROL)
      leg->stepper.current_tip_pose = orc_pose_make(tip_position_input, ORC_UNDEFINED_ROTATION);
  }
}

/* PoseController::poseForLegManipulation (pose_controller.cpp:561-611), simultaneous leg coordination */
static int poser_pose_for_leg_manipulation(orc_robot *r)
{
  int min_progress = INT_MAX; /* UNASSIGNED_VALUE */
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    stepper_t *st = &leg->stepper;
    double step_height = r->params.swing_height;
    double step_time = 1.0 / r->params.step_frequency;
    orc_pose target_pose;
    if (leg->leg_state == WALKING_TO_MANUAL)
    {
      target_pose = orc_pose_identity();
      target_pose.p = orc_v3_add(target_pose.p, r->inclination_pose.p); /* inclination control for the lifted leg */
      target_pose.p.z -= step_height;                                    /* pose the leg at step height */
    }
    else
    {
      target_pose = r->current_pose;
      target_pose.p = orc_v3_sub(target_pose.p, r->manual_pose.p);
      target_pose.p = orc_v3_add(target_pose.p, r->pc_default_pose.p); /* (calculateDefaultPose is never called: identity) */
    }
    orc_pose target_tip_pose = orc_pose_undefined();
    target_tip_pose.p = orc_pose_inverse_transform_vector(target_pose, st->default_tip_pose.p);
    if (leg->leg_state == WALKING_TO_MANUAL)
    {
      st->current_tip_pose = target_tip_pose;
      step_height = 0.0;
    }
    else if (leg->leg_state == MANUAL_TO_WALKING)
    {
      st->current_tip_pose = st->default_tip_pose;
    }
    int progress = leg_poser_step_to_position(r, leg, target_tip_pose, orc_pose_identity(), step_height, step_time, 1);
    if (progress < min_progress) min_progress = progress;
    if (progress != PROGRESS_COMPLETE)
    {
      leg_set_desired_tip_pose(leg, leg->poser.current_tip_pose, 1);
      leg_apply_ik(r, leg, 0);
    }
  }
  return min_progress;
}

/* AdmittanceController::updateStiffness(leg, scale_reference) (admittance_controller.cpp:66-92) */
static void admittance_update_stiffness_leg(orc_robot *r, leg_t *leg, double scale_reference)
{
  leg_t *adj1 = &r->leg[orc_mod(leg->id_number - 1, r->leg_count)];
  leg_t *adj2 = &r->leg[orc_mod(leg->id_number + 1, r->leg_count)];
  double virtual_stiffness = r->params.virtual_stiffness;
  double swing_stiffness = virtual_stiffness * (scale_reference * (r->params.swing_stiffness_scaler - 1) + 1);
  double load_stiffness = virtual_stiffness * (scale_reference * (r->params.load_stiffness_scaler - 1) + 1);
  leg->virtual_stiffness = swing_stiffness;
  if (adj1->leg_state != MANUAL) adj1->virtual_stiffness = load_stiffness;
  if (adj2->leg_state != MANUAL) adj2->virtual_stiffness = load_stiffness;
}

/* StateController::legStateToggle (state_controller.cpp:541-646) for the leg the toggle request designates.
 * Returns 1 when the transition has completed (the request flag is cleared), 2 when the request was refused (MAX_MANUAL_LEGS),
 * 0 while it is in progress, -1 while the robot still has to stop walking (velocity inputs zeroed). */
static int state_leg_state_toggle(orc_robot *r, int leg_id)
{
  if (r->walk_state != STOPPED)
  {
    r->linear_velocity_input[0] = r->linear_velocity_input[1] = 0.0;
    r->angular_velocity_input = 0.0;
    return -1;
  }
  leg_t *leg = &r->leg[leg_id];
  if (leg->leg_state == WALKING)
  {
    if (r->manual_leg_count < MAX_MANUAL_LEGS)
    {
      leg->leg_state = WALKING_TO_MANUAL;
      leg->stepper.swing_progress = -1.0;
      leg->stepper.stance_progress = -1.0;
      return 0;
    }
    return 2;
  }
  if (leg->leg_state == MANUAL)
  {
    leg->leg_state = MANUAL_TO_WALKING;
    return 0;
  }
  int to_manual = (leg->leg_state == WALKING_TO_MANUAL);
  r->pose_reset_mode = SHC_IMMEDIATE_ALL_RESET; /* force the pose to the new default pose */
  int progress = poser_pose_for_leg_manipulation(r);
  if (r->params.dynamic_stiffness)
  {
    double scale_reference = (double)progress / PROGRESS_COMPLETE;
    if (!to_manual) scale_reference = fabs(scale_reference - 1.0);
    admittance_update_stiffness_leg(r, leg, scale_reference);
  }
  if (progress == PROGRESS_COMPLETE)
  {
    leg->leg_state = to_manual ? MANUAL : WALKING;
    r->pose_reset_mode = SHC_NO_RESET;
    r->manual_leg_count += to_manual ? 1 : -1;
    return 1;
  }
  return 0;
}

/* ==================================================================================== AdmittanceController */

/* AdmittanceController::updateAdmittance (admittance_controller.cpp:22-63) for one leg.
 * boost::numeric::odeint::runge_kutta4 + integrate_const(0, step_time, step_time / 30): 30 fixed steps. */
static void admittance_leg(const shc_params *p, double state[2], orc_v3 tip_force_in, double delta_out[3])
{
  double tip_force[3] = { tip_force_in.x * p->force_gain, tip_force_in.y * p->force_gain, tip_force_in.z * p->force_gain };
  for (int i = 0; i < 3; ++i)
  {
    delta_out[i] = 0.0;
    double force_input = fmax(tip_force[i], 0.0);
    double damping = p->virtual_damping_ratio;
    double stiffness = p->virtual_stiffness;
    double mass = p->virtual_mass;
    double step_time = p->integrator_step_time;
    double virtual_damping = damping * 2 * sqrt(mass * stiffness);
    double dt = step_time / 30;
    /* integrate_const: while (less_eq_with_sign(time + dt, end, dt)) { do_step; ++step; time = start + step * dt; } */
    double time = 0.0;
    int step = 0;
    const double eps = 2.220446049250313e-16;
    while ((time + dt) - step_time <= eps)
    {
      double x0 = state[0], x1 = state[1];
      /* k1 */
      double k1_0 = x1;
      double k1_1 = -force_input / mass - virtual_damping / mass * x1 - stiffness / mass * x0;
      /* k2 at x + dt*0.5*k1 */
      double a0 = x0 + (0.5 * dt) * k1_0, a1 = x1 + (0.5 * dt) * k1_1;
      double k2_0 = a1;
      double k2_1 = -force_input / mass - virtual_damping / mass * a1 - stiffness / mass * a0;
      /* k3 at x + dt*(0*k1 + 0.5*k2) */
      double b0 = x0 + (0.0 * dt) * k1_0 + (0.5 * dt) * k2_0, b1 = x1 + (0.0 * dt) * k1_1 + (0.5 * dt) * k2_1;
      double k3_0 = b1;
      double k3_1 = -force_input / mass - virtual_damping / mass * b1 - stiffness / mass * b0;
      /* k4 at x + dt*(0*k1 + 0*k2 + 1*k3) */
      double c0 = x0 + (0.0 * dt) * k1_0 + (0.0 * dt) * k2_0 + (1.0 * dt) * k3_0;
      double c1 = x1 + (0.0 * dt) * k1_1 + (0.0 * dt) * k2_1 + (1.0 * dt) * k3_1;
      double k4_0 = c1;
      double k4_1 = -force_input / mass - virtual_damping / mass * c1 - stiffness / mass * c0;
      state[0] = x0 + (dt / 6.0) * k1_0 + (dt / 3.0) * k2_0 + (dt / 3.0) * k3_0 + (dt / 6.0) * k4_0;
      state[1] = x1 + (dt / 6.0) * k1_1 + (dt / 3.0) * k2_1 + (dt / 3.0) * k3_1 + (dt / 6.0) * k4_1;
      ++step;
      time = 0.0 + (double)step * dt;
    }
    double delta = orc_clamped(-state[0], -0.2, 0.2);
    double delta_direction = delta / fabs(delta);
    if (fabs(delta) > ADMITTANCE_DEADBAND)
      delta_out[i] = delta_direction * (fabs(delta) - ADMITTANCE_DEADBAND) / (1 - ADMITTANCE_DEADBAND);
  }
}

static void admittance_update_admittance(orc_robot *r)
{
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    orc_v3 tip_force = r->params.use_joint_effort ? leg->tip_force_calculated : leg->tip_force_measured;
    double d[3];
    admittance_leg(&r->params, leg->admittance_state, tip_force, d);
    leg_set_admittance_delta(leg, orc_v3_make(d[0], d[1], d[2]));
  }
}

/* AdmittanceController::updateStiffness(walker) (admittance_controller.cpp:96-134) */
static void admittance_update_stiffness(orc_robot *r)
{
  for (int l = 0; l < r->leg_count; ++l) r->leg[l].virtual_stiffness = r->params.virtual_stiffness;
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    if (leg->stepper.step_state == SWING)
    {
      double z_diff = leg->stepper.current_tip_pose.p.z - leg->stepper.default_tip_pose.p.z;
      double step_reference = 0;
      step_reference += fabs(z_diff / r->params.swing_height);
      leg_t *adj1 = &r->leg[orc_mod(l - 1, r->leg_count)];
      leg_t *adj2 = &r->leg[orc_mod(l + 1, r->leg_count)];
      double virtual_stiffness = r->params.virtual_stiffness;
      double swing_stiffness = virtual_stiffness * (step_reference * (r->params.swing_stiffness_scaler - 1) + 1);
      double load_stiffness = virtual_stiffness * (step_reference * (r->params.load_stiffness_scaler - 1));
      double current_stiffness_1 = adj1->virtual_stiffness;
      double current_stiffness_2 = adj2->virtual_stiffness;
      leg->virtual_stiffness = swing_stiffness;
      adj1->virtual_stiffness = current_stiffness_1 + load_stiffness;
      adj2->virtual_stiffness = current_stiffness_2 + load_stiffness;
    }
  }
}

/* ==================================================================================== StateController */

/* Model::updateModel (model.cpp:142-152) */
static void model_update_model(orc_robot *r)
{
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    leg->ik_failed = 0;
    leg_set_desired_tip_pose(leg, orc_pose_undefined(), 1);
    leg_apply_ik(r, leg, 0);
  }
}

/* Model::generateWorkspaces (model.cpp:120-138) */
static void model_generate_workspaces(orc_robot *r)
{
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *search_leg = (leg_t *)malloc(sizeof(leg_t));
    *search_leg = r->leg[l];
    leg_init(r, search_leg, 1); /* search_model->initLegs(true) */
    leg_generate_workspace(r, search_leg);
    r->leg[l].workspace = search_leg->workspace;
    r->leg[l].workspace_zero = search_leg->workspace_zero;
    free(search_leg);
  }
}

/* StateController::runningState (state_controller.cpp:379-447), no transitions / gait change / manual legs */
static void walker_update_manual_velocity(orc_robot *r);
static void walker_update_manual_pose(orc_robot *r);
int orc_adjust_parameter(orc_robot *r, int which, double value);
void orc_adjust_parameter_commit(orc_robot *r, int which);
void orc_request_parameter_adjust(orc_robot *r, int which, double value);
static void state_running_state(orc_robot *r)
{
  /* "Dynamically adjust parameters" (:411-414): after the posing part of this loop, before updateWalk */
  if (r->adjust_commit_due)
  {
    orc_adjust_parameter_commit(r, r->adjust_commit_due);
    r->adjust_commit_due = 0;
  }
  else if (r->parameter_adjust_flag)
  {
    if (orc_adjust_parameter(r, r->dynamic_parameter, r->new_parameter_value) != 0) r->parameter_adjust_flag = 0;
  }
  walker_update_walk(r, r->linear_velocity_input, r->angular_velocity_input);
  walker_update_manual_velocity(r); /* :433-434 */
  walker_update_manual_pose(r);     /* :439-440 */
  poser_update_stance(r);
  model_update_model(r);
}

/* StateController::transitionRobotState (state_controller.cpp:197-375), direct start-up path only */
static void state_transition_robot_state(orc_robot *r)
{
  if (r->robot_state == RS_UNKNOWN)
  {
    r->robot_state = RS_PACKED; /* joints at defaults are neither packed nor "unpacked": "state undefined" branch */
    r->new_robot_state = r->robot_state;
  }
  else if (r->robot_state == RS_PACKED && r->new_robot_state == RS_READY)
  {
    int progress = poser_direct_startup(r);
    r->startup_progress = progress;
    if (progress == PROGRESS_COMPLETE)
    {
      r->robot_state = RS_READY;
      for (int l = 0; l < r->leg_count; ++l) leg_update_default_configuration(&r->leg[l]);
      model_generate_workspaces(r);
      walker_generate_walkspace(r);
    }
  }
  else if (r->robot_state == RS_READY && r->new_robot_state == RS_RUNNING)
  {
    r->robot_state = RS_RUNNING;
  }
  if (r->robot_state == r->new_robot_state) r->transition_state_flag = 0;
}

/* StateController::loop (state_controller.cpp:162-193) */
static void state_loop(orc_robot *r)
{
  if (r->robot_state != RS_UNKNOWN)
  {
    poser_update_current_pose(r, r->robot_state);
    r->pose_state = r->auto_posing_state;
    if (r->params.admittance_control)
    {
      if (r->walk_state != STOPPED && r->params.dynamic_stiffness) admittance_update_stiffness(r);
      admittance_update_admittance(r);
    }
  }
  if (r->transition_state_flag) state_transition_robot_state(r);
  if (r->robot_state == RS_RUNNING) state_running_state(r);
}

/* ==================================================================================== public API */

size_t orc_sizeof_robot(void) { return sizeof(orc_robot); }

orc_robot *orc_create(const shc_params *params)
{
  if (!params || params->leg_count < 1 || params->leg_count > SHC_MAX_LEGS) return NULL;
  for (int l = 0; l < params->leg_count; ++l)
    if (params->leg_dof[l] < 1 || params->leg_dof[l] > SHC_MAX_JOINTS) return NULL;
  orc_robot *r = (orc_robot *)calloc(1, sizeof(orc_robot));
  r->params = *params;
  /* Model::Model (model.cpp:16-27) */
  r->leg_count = params->leg_count;
  r->time_delta = params->time_delta;
  r->current_pose = orc_pose_identity();
  r->default_pose = orc_pose_identity();
  r->imu_orientation = ORC_UNDEFINED_ROTATION;
  r->imu_angular_velocity = orc_v3_make(0, 0, 0);
  r->primary_leg_selection = r->secondary_leg_selection = LEG_UNDESIGNATED;
  r->target_body_pose = orc_pose_identity(); /* (left uninitialised by the reference until the first plan step completes, :685) */
  r->set_target = 1; /* pose_controller.h:299-304 */
  r->first_sequence_execution = 1;
  r->reset_transition_sequence = 1;
  /* Model::generate -> Leg::Leg + Leg::generate (model.cpp:44-62, 169-239) */
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    leg->id_number = l;
    leg->joint_count = params->leg_dof[l];
    leg->leg_state = WALKING;
    leg->admittance_delta = orc_v3_make(0, 0, 0);
    leg->admittance_state[0] = leg->admittance_state[1] = 0.0;
    leg->desired_tip_pose = orc_pose_undefined();
    leg->current_tip_pose = orc_pose_undefined();
    leg->step_plane_pose = orc_pose_undefined();
    for (int k = 0; k <= leg->joint_count; ++k)
    {
      leg->link[k].d = params->link[l][k].d;
      leg->link[k].theta = params->link[l][k].theta;
      leg->link[k].r = params->link[l][k].r;
      leg->link[k].alpha = params->link[l][k].alpha;
    }
    for (int j = 0; j < leg->joint_count; ++j)
    { /* Joint::Joint (model.cpp:1026-1073); reference link = link[j] */
      joint_t *jt = &leg->joint[j];
      jt->min_position = params->joint[l][j].min;
      jt->max_position = params->joint[l][j].max;
      jt->offset = params->joint[l][j].offset;
      jt->unpacked_position = params->joint[l][j].unpacked;
      jt->max_angular_speed = params->joint[l][j].max_vel;
      jt->default_position = orc_clamped(0.0, jt->min_position, jt->max_position);
      jt->current_position = ORC_UNASSIGNED_VALUE;
      jt->identity_transform = orc_create_dh_matrix(leg->link[j].d, leg->link[j].theta, leg->link[j].r, leg->link[j].alpha);
      jt->current_transform = jt->identity_transform;
    }
    { /* Tip::Tip (model.cpp:1119-1127) */
      const link_t *rl = &leg->link[leg->joint_count];
      leg->tip_identity_transform = orc_create_dh_matrix(rl->d, rl->theta, rl->r, rl->alpha);
      leg->tip_current_transform = leg->tip_identity_transform;
    }
  }
  /* StateController::init (state_controller.cpp:127-153) */
  walker_init(r);
  poser_init(r);
  r->robot_state = RS_UNKNOWN;
  r->new_robot_state = RS_UNKNOWN;
  r->transition_state_flag = 0;
  /* initModel(true) (main.cpp:101) */
  for (int l = 0; l < r->leg_count; ++l) leg_init(r, &r->leg[l], 1);
  return r;
}

void orc_destroy(orc_robot *r) { free(r); }

orc_robot *orc_clone(const orc_robot *r)
{
  orc_robot *c = (orc_robot *)malloc(sizeof(orc_robot));
  memcpy(c, r, sizeof(orc_robot));
  return c;
}

int orc_startup(orc_robot *r)
{
  int loops = 0;
  /* first loop: UNKNOWN -> PACKED (transition_state_flag_ is raised by the start button callback) */
  r->transition_state_flag = 1;
  state_loop(r);
  ++loops;
  /* START pressed: PACKED -> READY (direct start-up) */
  r->new_robot_state = RS_READY;
  r->transition_state_flag = 1;
  while (r->robot_state != RS_READY)
  {
    state_loop(r);
    if (++loops > 100000) return -1;
  }
  /* START pressed again: READY -> RUNNING */
  r->new_robot_state = RS_RUNNING;
  r->transition_state_flag = 1;
  state_loop(r);
  ++loops;
  return r->robot_state == RS_RUNNING ? loops : -1;
}

/* StateController::changeGait (state_controller.cpp:513-538) with initGaitParameters (:1941-1969) and, for auto posing,
 * initAutoPoseParameters + setAutoPoseParams (:1973-1999, pose_controller.cpp:44-106): the gait-dependent members of
 * `np` replace the current ones.  Returns 1 when the gait was changed, 0 when the robot is still walking (the reference
 * then zeroes the velocity inputs and retries on the next loop). */
int orc_change_gait(orc_robot *r, const shc_params *np)
{
  if (r->walk_state != STOPPED)
  {
    r->linear_velocity_input[0] = r->linear_velocity_input[1] = 0.0;
    r->angular_velocity_input = 0.0;
    return 0;
  }
  shc_params *p = &r->params;
  p->stance_phase = np->stance_phase;
  p->swing_phase = np->swing_phase;
  p->phase_offset = np->phase_offset;
  memcpy(p->offset_multiplier, np->offset_multiplier, sizeof p->offset_multiplier);
  r->step = generate_step_cycle(p); /* WalkController::generateStepCycle */
  walker_generate_limits(r);
  if (p->auto_posing)
  {
    p->pose_frequency = np->pose_frequency;
    p->pose_phase_length = np->pose_phase_length;
    p->n_auto_posers = np->n_auto_posers;
    memcpy(p->pose_phase_starts, np->pose_phase_starts, sizeof p->pose_phase_starts);
    memcpy(p->pose_phase_ends, np->pose_phase_ends, sizeof p->pose_phase_ends);
    memcpy(p->pose_negation_phase_starts, np->pose_negation_phase_starts, sizeof p->pose_negation_phase_starts);
    memcpy(p->pose_negation_phase_ends, np->pose_negation_phase_ends, sizeof p->pose_negation_phase_ends);
    memcpy(p->negation_transition_ratio, np->negation_transition_ratio, sizeof p->negation_transition_ratio);
    memcpy(p->x_amplitudes, np->x_amplitudes, sizeof p->x_amplitudes);
    memcpy(p->y_amplitudes, np->y_amplitudes, sizeof p->y_amplitudes);
    memcpy(p->z_amplitudes, np->z_amplitudes, sizeof p->z_amplitudes);
    memcpy(p->gravity_amplitudes, np->gravity_amplitudes, sizeof p->gravity_amplitudes);
    memcpy(p->roll_amplitudes, np->roll_amplitudes, sizeof p->roll_amplitudes);
    memcpy(p->pitch_amplitudes, np->pitch_amplitudes, sizeof p->pitch_amplitudes);
    memcpy(p->yaw_amplitudes, np->yaw_amplitudes, sizeof p->yaw_amplitudes);
    poser_set_auto_pose_params(r);
  }
  return 1;
}

/* StateController::adjustParameter (state_controller.cpp:451-509), as runningState calls it while parameter_adjust_flag_ is set (:411-414).
 * `which` = enum ParameterSelection (parameters_and_states.h:165-178); the callbacks (:1419-1548) have clamped `value` to the parameter's range.
 * Eight of the nine parameters are only stored - every reader takes params_.<name>.current_value when it runs.  step_frequency: the value is stored
 * (:454), the step cycle it gives is generated WITHOUT being set (:458), its two speed maps replace the walker's (:459-463; generateLimits also
 * sets every LegStepper's phase offset from the new cycle, walk_controller.cpp:277), and only when the desired body velocity is inside what the
 * velocity input maps to under the new limits (:464-489) are the step cycle and all four maps regenerated (:491-492; a robot that is MOVING maps its
 * legs' phases onto the new period, LegStepper::updatePhase walk_controller.cpp:402-409, :862-867).  Split in two so that a batch can decide
 * together (one engine has one set of tables): orc_adjust_parameter_probe does everything up to the decision and returns it, orc_adjust_parameter_commit is
 * :491-492; orc_adjust_parameter = the reference's function for one robot (state_running_state calls it where runningState does while a request
 * made with orc_request_parameter_adjust is pending).  A batch (orc_batch_adjust_parameter) probes every robot between two loops - nothing the posing
 * part of a loop does enters the decision - and, when all agree, has every robot run :491-492 at adjustParameter's place in its next loop. */
static double *adjustable_field(shc_params *p, int which)
{
  switch (which)
  {
    case 1: return &p->step_frequency;
    case 2: return &p->swing_height;
    case 3: return &p->swing_width;
    case 4: return &p->step_depth;
    case 5: return &p->stance_span_modifier;
    case 6: return &p->virtual_mass;
    case 7: return &p->virtual_stiffness;
    case 8: return &p->virtual_damping_ratio;
    case 9: return &p->force_gain;
    default: return NULL;
  }
}
int orc_adjust_parameter_probe(orc_robot *r, int which, double value)
{
  double *field = adjustable_field(&r->params, which);
  if (!field) return -1;
  *field = value; /* p->current_value = new_parameter_value_ */
  if (which != 1) return 1;
  shc_step_cycle new_step_cycle = generate_step_cycle(&r->params); /* walker_->generateStepCycle(false) */
  double max_linear_speed_map[SHC_N_BEARINGS], max_angular_speed_map[SHC_N_BEARINGS];
  walker_generate_limits_for(r, new_step_cycle, max_linear_speed_map, max_angular_speed_map, NULL, NULL);
  memcpy(r->max_linear_speed, max_linear_speed_map, sizeof max_linear_speed_map);   /* setLinearSpeedLimitMap */
  memcpy(r->max_angular_speed, max_angular_speed_map, sizeof max_angular_speed_map); /* setAngularSpeedLimitMap */
  double max_linear_speed = walker_get_limit(r, r->linear_velocity_input, r->angular_velocity_input, max_linear_speed_map);
  double max_angular_speed = walker_get_limit(r, r->linear_velocity_input, r->angular_velocity_input, max_angular_speed_map);
  double target_linear_velocity[2] = { 0.0, 0.0 }, target_angular_velocity = 0.0;
  const double *in = r->linear_velocity_input;
  double in_norm = sqrt(in[0] * in[0] + in[1] * in[1]);
  if (r->params.velocity_input_mode == SHC_VEL_THROTTLE)
  {
    double k = in_norm > 1.0 ? 1.0 / in_norm : 1.0; /* clamped(linear_velocity_input_, 1.0) */
    target_linear_velocity[0] = in[0] * k * max_linear_speed;
    target_linear_velocity[1] = in[1] * k * max_linear_speed;
    target_angular_velocity = orc_clamped(r->angular_velocity_input, -1.0, 1.0) * max_angular_speed;
    target_linear_velocity[0] *= (1.0 - fabs(r->angular_velocity_input));
    target_linear_velocity[1] *= (1.0 - fabs(r->angular_velocity_input));
  }
  else
  {
    double k = in_norm > max_linear_speed ? max_linear_speed / in_norm : 1.0; /* clamped(linear_velocity_input_, max_linear_speed) */
    target_linear_velocity[0] = in[0] * k;
    target_linear_velocity[1] = in[1] * k;
    target_angular_velocity = orc_clamped(r->angular_velocity_input, -max_angular_speed, max_angular_speed);
  }
  return (r->desired_linear_velocity[0] <= target_linear_velocity[0] && r->desired_linear_velocity[1] <= target_linear_velocity[1] &&
          fabs(r->desired_angular_velocity) <= fabs(target_angular_velocity)) ? 1 : 0;
}
void orc_adjust_parameter_commit(orc_robot *r, int which)
{
  if (which != 1) return;
  r->step = generate_step_cycle(&r->params); /* walker_->generateStepCycle(): set_step_cycle */
  if (r->walk_state == MOVING)
  {
    for (int l = 0; l < r->leg_count; ++l)
    { /* LegStepper::updatePhase */
      stepper_t *s = &r->leg[l].stepper;
      s->phase = (int)(s->step_progress * r->step.period);
      stepper_update_step_state(r, s);
    }
  }
  walker_generate_limits(r); /* walker_->generateLimits() */
}
int orc_adjust_parameter(orc_robot *r, int which, double value)
{
  int set_new_parameter = orc_adjust_parameter_probe(r, which, value);
  if (set_new_parameter == 1) orc_adjust_parameter_commit(r, which);
  return set_new_parameter;
}
void orc_get_tables(const orc_robot *r, shc_tables *out)
{
  memset(out, 0, sizeof *out);
  out->step = r->step;
  for (int l = 0; l < r->leg_count; ++l)
  {
    out->phase_offset[l] = r->leg[l].stepper.phase_offset;
    for (int j = 0; j < r->leg[l].joint_count; ++j) out->default_joint_position[l][j] = r->leg[l].joint[j].default_position;
    { /* the plane the walkspace was generated from: height 0 (default tips at their identity positions) */
      double plane[SHC_N_BEARINGS];
      if (!ws_get_workplane(&r->leg[l].workspace, 0.0, plane)) memset(plane, 0, sizeof plane);
      for (int b = 0; b < SHC_N_BEARINGS; ++b) out->workspace_radius[l][b] = plane[b];
    }
  }
  for (int b = 0; b < SHC_N_BEARINGS; ++b)
  {
    out->walkspace[b] = r->walkspace[b];
    out->max_linear_speed[b] = r->max_linear_speed[b];
    out->max_angular_speed[b] = r->max_angular_speed[b];
    out->max_linear_acceleration[b] = r->max_linear_acceleration[b];
    out->max_angular_acceleration[b] = r->max_angular_acceleration[b];
  }
  out->pose_phase_length = r->pose_phase_length;
  out->pose_normaliser = r->normaliser;
  out->auto_pose_reference_leg = r->auto_pose_reference_leg;
}

void orc_set_velocity(orc_robot *r, double vx, double vy, double omega)
{
  r->linear_velocity_input[0] = vx;
  r->linear_velocity_input[1] = vy;
  r->angular_velocity_input = omega;
}

void orc_set_imu(orc_robot *r, const double q[4], const double gyro[3])
{ /* Model::setImuData (model.h:146-153): orientation.normalized() */
  r->imu_orientation = orc_quat_normalized(orc_quat_make(q[0], q[1], q[2], q[3]));
  r->imu_angular_velocity = orc_v3_make(gyro[0], gyro[1], gyro[2]);
}

/* tipStatesCallback with wrench values (state_controller.cpp:1617-1648): touchdown detection on, force stored,
 * Leg::touchdownDetection (model.cpp:712-722) */
void orc_set_tip_force(orc_robot *r, const double *force)
{
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    leg->stepper.touchdown_detection = 1;
    leg->tip_force_measured = orc_v3_make(force[l * 3 + 0], force[l * 3 + 1], force[l * 3 + 2]);
    if (orc_v3_norm(leg->tip_force_measured) > r->params.touchdown_threshold && orc_pose_eq(leg->step_plane_pose, orc_pose_undefined()))
      leg->step_plane_pose = leg->current_tip_pose;
    else if (orc_v3_norm(leg->tip_force_measured) < r->params.liftoff_threshold)
      leg->step_plane_pose = orc_pose_undefined();
  }
}

void orc_set_joint_effort(orc_robot *r, const double *effort)
{
  int k = 0;
  for (int l = 0; l < r->leg_count; ++l)
    for (int j = 0; j < r->leg[l].joint_count; ++j)
    {
      r->leg[l].joint[j].current_effort = effort[k];
      r->leg[l].joint[j].desired_effort = effort[k++]; /* "HACK" (state_controller.cpp:1590) */
    }
}

/* jointStatesCallback (state_controller.cpp:1566-1594): raw motor positions [legs][dof] (offset removed, :1581), efforts */
void orc_set_joint_states_msg(orc_robot *r, const double *position, const double *velocity, const double *effort)
{
  int k = 0;
  for (int l = 0; l < r->leg_count; ++l)
    for (int j = 0; j < r->leg[l].joint_count; ++j, ++k)
    {
      joint_t *jt = &r->leg[l].joint[j];
      if (position) jt->current_position = position[k] - jt->offset;
      if (velocity) jt->current_velocity = velocity[k];
      if (effort) { jt->current_effort = effort[k]; jt->desired_effort = effort[k]; }
    }
}

/* tipStatesCallback, step_plane values of the tip range sensors (state_controller.cpp:1651-1672): [legs][3] */
/* targetTipPoseCallback (state_controller.cpp:1706-1767): robot RUNNING; a target reaches the LegStepper only while the robot
 * is not STOPPED (else the planner-mode LegPoser takes it, not restated), a default likewise (:1746).  Returns 1 if taken. */
static shc_external_target *external_record(orc_robot *r, int which, int leg)
{
  if (which == SHC_EXTERNAL_PLANNER_TARGET) return &r->leg[leg].poser.external_target;
  stepper_t *s = &r->leg[leg].stepper;
  return which ? &s->external_default : &s->external_target;
}

/* targetTipPoseCallback (state_controller.cpp:1706-1767).  Returns 1 when a LegStepper took the request, 2 when the robot is
 * STOPPED and the LegPoser took the target for planner mode (:1738-1742), 0 when it was dropped (a default for a STOPPED robot). */
int orc_set_external_target(orc_robot *r, int which, int leg, const shc_external_target *t)
{
  if (!t->defined)
  {
    external_record(r, which, leg)->defined = 0;
    return 1;
  }
  if (which == SHC_EXTERNAL_PLANNER_TARGET || (which == SHC_EXTERNAL_TARGET && r->walk_state == STOPPED))
  {
    r->leg[leg].poser.external_target = *t;
    r->target_tip_pose_acquired = 1;
    return 2;
  }
  if (r->walk_state == STOPPED) return 0;
  *external_record(r, which, leg) = *t;
  return 1;
}

/* generateExternalTargetTransforms (state_controller.cpp:703-773): refreshed transform_ of a defined request */
void orc_set_external_transform(orc_robot *r, int which, int leg, const double *transform)
{
  shc_external_target *t = external_record(r, which, leg);
  if (t->defined) memcpy(t->transform, transform, sizeof t->transform);
}

void orc_get_external_target(const orc_robot *r, int which, int leg, shc_external_target *out)
{
  *out = *external_record((orc_robot *)r, which, leg);
}

void orc_set_step_plane(orc_robot *r, const double *step_plane)
{
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    leg->stepper.touchdown_detection = 1;
    const double *sp = step_plane + 3 * l;
    if (sp[2] != ORC_UNASSIGNED_VALUE)
    {
      orc_v3 step_plane_position = orc_v3_make(sp[2], 0.0, 0.0);
      orc_v3 step_plane_normal = orc_v3_make(sp[0], sp[1], -1.0);
      orc_quat step_plane_orientation = orc_quat_from_two_vectors(orc_v3_make(0, 0, 1.0), orc_v3_neg(step_plane_normal));
      orc_m4 t = transform_from_joint(leg, leg->joint_count + 1, 0); /* Tip::getPoseRobotFrame (model.h:684-688) */
      leg->step_plane_pose = orc_pose_transform_m4(orc_pose_make(step_plane_position, step_plane_orientation), &t);
    }
    else
    {
      leg->step_plane_pose = orc_pose_undefined();
    }
  }
}

/* publishDesiredJointState (state_controller.cpp:777-805): [legs][dof] each */
void orc_get_joint_commands(const orc_robot *r, double *position, double *velocity, double *effort, double *position_command)
{
  int k = 0;
  for (int l = 0; l < r->leg_count; ++l)
    for (int j = 0; j < r->leg[l].joint_count; ++j, ++k)
    {
      const joint_t *jt = &r->leg[l].joint[j];
      if (position) position[k] = jt->desired_position;
      if (velocity) velocity[k] = jt->desired_velocity;
      if (effort) effort[k] = jt->desired_effort;
      if (position_command) position_command[k] = jt->desired_position + jt->offset;
    }
}

void orc_set_pose_input(orc_robot *r, const double tv[3], const double rv[3])
{
  r->translation_velocity_input = orc_v3_make(tv[0], tv[1], tv[2]);
  r->rotation_velocity_input = orc_v3_make(rv[0], rv[1], rv[2]);
}

void orc_set_pose_reset_mode(orc_robot *r, int mode) { r->pose_reset_mode = mode; }

void orc_cycle(orc_robot *r) { state_loop(r); }

void orc_get_joint_state(const orc_robot *r, double *q, double *qd)
{
  int k = 0;
  for (int l = 0; l < r->leg_count; ++l)
    for (int j = 0; j < r->leg[l].joint_count; ++j, ++k)
    {
      if (q) q[k] = r->leg[l].joint[j].desired_position;
      if (qd) qd[k] = r->leg[l].joint[j].desired_velocity;
    }
}

static void put3(double *dst, orc_v3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
static void put_pose7(double *dst, orc_pose p);

void orc_get_leg_state(const orc_robot *r, double *walker_tip, double *poser_tip, double *model_tip, double *tip_force,
                       double *admittance, int32_t *leg_status)
{
  for (int l = 0; l < r->leg_count; ++l)
  {
    const leg_t *leg = &r->leg[l];
    if (walker_tip) put3(walker_tip + 3 * l, leg->stepper.current_tip_pose.p);
    if (poser_tip) put3(poser_tip + 3 * l, leg->poser.current_tip_pose.p);
    if (model_tip) put3(model_tip + 3 * l, leg->current_tip_pose.p);
    if (tip_force) put3(tip_force + 3 * l, leg->tip_force_calculated);
    if (admittance) put3(admittance + 3 * l, leg->admittance_delta);
    if (leg_status) leg_status[l] = (leg->stepper.step_state & 3) | (leg->ik_failed ? 4 : 0) | (leg->stepper.phase << 8);
  }
}

void orc_get_body_state(const orc_robot *r, double pose[7], double velocity[3], int32_t *walk_state)
{
  if (pose)
  {
    pose[0] = r->current_pose.p.x; pose[1] = r->current_pose.p.y; pose[2] = r->current_pose.p.z;
    pose[3] = r->current_pose.r.w; pose[4] = r->current_pose.r.x; pose[5] = r->current_pose.r.y; pose[6] = r->current_pose.r.z;
  }
  if (velocity)
  {
    velocity[0] = r->desired_linear_velocity[0];
    velocity[1] = r->desired_linear_velocity[1];
    velocity[2] = r->desired_angular_velocity;
  }
  if (walk_state) *walk_state = r->walk_state;
}

int orc_get_ik_failures(const orc_robot *r)
{
  int n = 0;
  for (int l = 0; l < r->leg_count; ++l) n += r->leg[l].ik_failed;
  return n;
}

/* ------------------------------------------------------------------------------------ batch driver */

struct orc_batch
{
  int64_t n;
  orc_robot *robots; /* contiguous */
  int dof_total;
};

orc_batch *orc_batch_create(const shc_params *params, int64_t n)
{
  orc_robot *proto = orc_create(params);
  if (!proto) return NULL;
  if (orc_startup(proto) < 0) { orc_destroy(proto); return NULL; }
  orc_batch *b = (orc_batch *)calloc(1, sizeof(orc_batch));
  b->n = n;
  b->robots = (orc_robot *)malloc(sizeof(orc_robot) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) memcpy(&b->robots[i], proto, sizeof(orc_robot));
  b->dof_total = 0;
  for (int l = 0; l < proto->leg_count; ++l) b->dof_total += proto->leg[l].joint_count;
  orc_destroy(proto);
  return b;
}

void orc_batch_destroy(orc_batch *b)
{
  if (!b) return;
  free(b->robots);
  free(b);
}

orc_robot *orc_batch_robot(orc_batch *b, int64_t i) { return &b->robots[i]; }

void orc_batch_set_velocity(orc_batch *b, const double *lin_xy, const double *ang)
{
  for (int64_t i = 0; i < b->n; ++i)
    orc_set_velocity(&b->robots[i], lin_xy ? lin_xy[2 * i] : 0.0, lin_xy ? lin_xy[2 * i + 1] : 0.0, ang ? ang[i] : 0.0);
}
void orc_batch_set_imu(orc_batch *b, const double *quat, const double *gyro)
{
  static const double zero3[3] = { 0, 0, 0 };
  for (int64_t i = 0; i < b->n; ++i) orc_set_imu(&b->robots[i], quat + 4 * i, gyro ? gyro + 3 * i : zero3);
}
void orc_batch_set_tip_force(orc_batch *b, const double *force)
{
  for (int64_t i = 0; i < b->n; ++i) orc_set_tip_force(&b->robots[i], force + (size_t)i * b->robots[i].leg_count * 3);
}
void orc_batch_set_joint_effort(orc_batch *b, const double *effort)
{
  for (int64_t i = 0; i < b->n; ++i) orc_set_joint_effort(&b->robots[i], effort + (size_t)i * b->dof_total);
}
void orc_batch_set_pose_input(orc_batch *b, const double *tv, const double *rv)
{
  for (int64_t i = 0; i < b->n; ++i) orc_set_pose_input(&b->robots[i], tv + 3 * i, rv + 3 * i);
}

typedef struct { orc_batch *b; int64_t lo, hi; int n_cycles; } batch_job;

static void *batch_worker(void *arg)
{
  batch_job *j = (batch_job *)arg;
  /* robot-major: each robot runs all its cycles (the reference's one-robot loop), robots are independent */
  for (int64_t i = j->lo; i < j->hi; ++i)
    for (int c = 0; c < j->n_cycles; ++c) state_loop(&j->b->robots[i]);
  return NULL;
}

void orc_batch_set_pose_reset_mode(orc_batch *b, const int32_t *mode)
{
  for (int64_t i = 0; i < b->n; ++i) orc_set_pose_reset_mode(&b->robots[i], mode[i]);
}

double orc_batch_step(orc_batch *b, int n_cycles, int n_threads)
{
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  if ((int64_t)n_threads > b->n) n_threads = (int)b->n;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  if (n_threads == 1)
  {
    batch_job j = { b, 0, b->n, n_cycles };
    batch_worker(&j);
  }
  else
  {
    pthread_t th[256];
    batch_job jobs[256];
    for (int t = 0; t < n_threads; ++t)
    {
      jobs[t].b = b;
      jobs[t].lo = b->n * t / n_threads;
      jobs[t].hi = b->n * (t + 1) / n_threads;
      jobs[t].n_cycles = n_cycles;
      pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

void orc_batch_get_joint_state(orc_batch *b, double *q, double *qd)
{
  for (int64_t i = 0; i < b->n; ++i)
    orc_get_joint_state(&b->robots[i], q ? q + (size_t)i * b->dof_total : NULL, qd ? qd + (size_t)i * b->dof_total : NULL);
}

void orc_batch_get_leg_state(orc_batch *b, double *walker_tip, double *poser_tip, double *model_tip, double *tip_force,
                             double *admittance, int32_t *leg_status)
{
  for (int64_t i = 0; i < b->n; ++i)
  {
    size_t L = (size_t)b->robots[i].leg_count;
    orc_get_leg_state(&b->robots[i], walker_tip ? walker_tip + i * L * 3 : NULL, poser_tip ? poser_tip + i * L * 3 : NULL,
                      model_tip ? model_tip + i * L * 3 : NULL, tip_force ? tip_force + i * L * 3 : NULL,
                      admittance ? admittance + i * L * 3 : NULL, leg_status ? leg_status + i * L : NULL);
  }
}

void orc_batch_get_body_state(orc_batch *b, double *pose, double *velocity, int32_t *walk_state)
{
  for (int64_t i = 0; i < b->n; ++i)
    orc_get_body_state(&b->robots[i], pose ? pose + 7 * i : NULL, velocity ? velocity + 3 * i : NULL,
                       walk_state ? walk_state + i : NULL);
}

/* StateController::publishLegState (state_controller.cpp:809-893): numeric payload per leg */
void orc_get_leg_state_msg(const orc_robot *r, shc_leg_state_msg *legs)
{
  const shc_step_cycle *step = &r->step;
  for (int l = 0; l < r->leg_count; ++l)
  {
    const leg_t *leg = &r->leg[l];
    const stepper_t *ls = &leg->stepper;
    shc_leg_state_msg *m = &legs[l];
    memset(m, 0, sizeof *m);
    put3(m->walker_tip_position, ls->current_tip_pose.p);
    put3(m->target_tip_position, ls->target_tip_pose.p);
    put3(m->poser_tip_position, leg->poser.current_tip_pose.p);
    put3(m->model_tip_position, leg->current_tip_pose.p);
    { /* actual_tip_pose = leg->applyFK(false, true): FK of Joint::current_position_ (:837-839), which Leg::init(true) set to the
       * initial default positions (model.cpp:292-296) and only jointStatesCallback writes afterwards */
      leg_t *copy = (leg_t *)malloc(sizeof(leg_t));
      *copy = *leg;
      for (int j = 0; j < copy->joint_count; ++j) copy->joint[j].desired_position = copy->joint[j].current_position;
      put_pose7(m->actual_tip_pose, leg_apply_fk((orc_robot *)r, copy));
      free(copy);
    }
    for (int j = 0; j < leg->joint_count; ++j)
    {
      m->joint_positions[j] = leg->joint[j].desired_position;
      m->joint_velocities[j] = leg->joint[j].desired_velocity;
      m->joint_efforts[j] = leg->joint[j].desired_effort; /* = the last measured effort (state_controller.cpp:1590) */
    }
    m->swing_progress = ls->swing_progress;
    m->stance_progress = ls->stance_progress;
    double swing_time = ((double)step->swing_period / step->period) / step->frequency;
    double stance_time = ((double)step->stance_period / step->period) / step->frequency;
    double time_to_swing_end;
    if (ls->stance_progress >= 0.0) time_to_swing_end = stance_time * (1.0 - ls->stance_progress) + swing_time;
    else time_to_swing_end = swing_time * (1.0 - ls->swing_progress);
    m->time_to_swing_end = time_to_swing_end;
    orc_pose d = walker_calculate_odometry(r, time_to_swing_end);
    m->pose_delta[0] = d.p.x; m->pose_delta[1] = d.p.y; m->pose_delta[2] = d.p.z;
    m->pose_delta[3] = d.r.w; m->pose_delta[4] = d.r.x; m->pose_delta[5] = d.r.y; m->pose_delta[6] = d.r.z;
    /* model_tip_velocity: publishLegState's own applyFK() on unchanged joints (:840) zeroes Leg::current_tip_velocity_ first */
    m->model_tip_velocity[0] = m->model_tip_velocity[1] = m->model_tip_velocity[2] = 0.0;
    m->auto_pose[0] = leg->poser.auto_pose.p.x; m->auto_pose[1] = leg->poser.auto_pose.p.y; m->auto_pose[2] = leg->poser.auto_pose.p.z;
    m->auto_pose[3] = leg->poser.auto_pose.r.w; m->auto_pose[4] = leg->poser.auto_pose.r.x; m->auto_pose[5] = leg->poser.auto_pose.r.y;
    m->auto_pose[6] = leg->poser.auto_pose.r.z;
    m->tip_force[0] = leg->tip_force_calculated.x * r->params.force_gain;
    m->tip_force[1] = leg->tip_force_calculated.y * r->params.force_gain;
    m->tip_force[2] = leg->tip_force_calculated.z * r->params.force_gain;
    put3(m->admittance_delta, leg->admittance_delta);
    m->virtual_stiffness = leg->virtual_stiffness;
  }
}

/* changeGait on every robot of the batch; returns how many were still walking (0 = the gait was changed everywhere) */
int64_t orc_batch_change_gait(orc_batch *b, const shc_params *np)
{
  int64_t walking = 0;
  for (int64_t i = 0; i < b->n; ++i) walking += b->robots[i].walk_state != STOPPED;
  if (walking) /* a batch shares one gait: nobody changes until everybody has stopped */
  {
    for (int64_t i = 0; i < b->n; ++i) orc_set_velocity(&b->robots[i], 0.0, 0.0, 0.0);
    return walking;
  }
  for (int64_t i = 0; i < b->n; ++i) orc_change_gait(&b->robots[i], np);
  return 0;
}

/* the batch's decision: every robot probes (each takes the side effects of a pending change), all commit only when all may */
int64_t orc_batch_adjust_parameter(orc_batch *b, int which, double value)
{
  if (which != 1)
  { /* nothing to decide: every robot serves the request inside its next loop, where adjustParameter stands (the posing part of that loop - admittance,
     * dynamic stiffness - still reads the old value, updateWalk / updateModel the new one) */
    if (!adjustable_field(&b->robots[0].params, which)) return -1;
    for (int64_t i = 0; i < b->n; ++i) orc_request_parameter_adjust(&b->robots[i], which, value);
    return 0;
  }
  int64_t waiting = 0;
  for (int64_t i = 0; i < b->n; ++i)
  {
    int ok = orc_adjust_parameter_probe(&b->robots[i], which, value);
    if (ok < 0) return -1;
    if (!ok) ++waiting;
  }
  if (!waiting) /* :491-492 run where adjustParameter stands in the loop: after the posing part of the next loop, before its updateWalk (state_running_state) */
    for (int64_t i = 0; i < b->n; ++i) b->robots[i].adjust_commit_due = which;
  return waiting;
}
/* parameterAdjustCallback / dynamicParameterCallback for one robot: the request is served inside its loops (state_running_state) until it is set */
void orc_request_parameter_adjust(orc_robot *r, int which, double value)
{
  r->parameter_adjust_flag = 1;
  r->dynamic_parameter = which;
  r->new_parameter_value = value;
}
int orc_parameter_adjust_pending(const orc_robot *r) { return r->parameter_adjust_flag; }


/* WalkController::getOdometryIdeal (walk_controller.h) per robot: position xyz + rotation wxyz */
void orc_get_odometry(const orc_robot *r, double o[7])
{
  o[0] = r->odometry_ideal.p.x; o[1] = r->odometry_ideal.p.y; o[2] = r->odometry_ideal.p.z;
  o[3] = r->odometry_ideal.r.w; o[4] = r->odometry_ideal.r.x; o[5] = r->odometry_ideal.r.y; o[6] = r->odometry_ideal.r.z;
}
void orc_batch_get_odometry(orc_batch *b, double *pose)
{
  for (int64_t i = 0; i < b->n; ++i) orc_get_odometry(&b->robots[i], pose + 7 * i);
}

/* Leg::getVirtualStiffness (model.h:264) per leg; the reference leaves the member uninitialised until the first
 * updateStiffness call - the restatement (calloc) and the engine both start it at 0 */
void orc_batch_get_virtual_stiffness(orc_batch *b, double *stiffness)
{
  for (int64_t i = 0; i < b->n; ++i)
  {
    const orc_robot *r = &b->robots[i];
    for (int l = 0; l < r->leg_count; ++l) stiffness[i * r->leg_count + l] = r->leg[l].virtual_stiffness;
  }
}

/* ------------------------------------------------------------------------------------ unit-level test hooks */


/* ------------------------------------------------------------------------------------ state snapshots
 * shc_instance_state (include/shc_batch.h) speaks the reference's member names, so the oracle's side is a plain copy.
 * Used by the teacher-forced one-step parity tests: the oracle's state is loaded into the engine before every cycle. */
static void put_pose7(double *dst, orc_pose p)
{
  dst[0] = p.p.x; dst[1] = p.p.y; dst[2] = p.p.z; dst[3] = p.r.w; dst[4] = p.r.x; dst[5] = p.r.y; dst[6] = p.r.z;
}
static orc_pose get_pose7(const double *s) { return orc_pose_make(orc_v3_make(s[0], s[1], s[2]), orc_quat_make(s[3], s[4], s[5], s[6])); }
static void put_quat4(double *dst, orc_quat q) { dst[0] = q.w; dst[1] = q.x; dst[2] = q.y; dst[3] = q.z; }
static orc_v3 get3(const double *s) { return orc_v3_make(s[0], s[1], s[2]); }

void orc_get_state(const orc_robot *r, shc_instance_state *o)
{
  memset(o, 0, sizeof *o);
  o->desired_linear_velocity[0] = r->desired_linear_velocity[0];
  o->desired_linear_velocity[1] = r->desired_linear_velocity[1];
  o->desired_angular_velocity = r->desired_angular_velocity;
  put3(o->walk_plane, r->walk_plane);
  put3(o->walk_plane_normal, r->walk_plane_normal);
  { /* LegStepper::walk_plane_ copies: every leg that stepped last cycle took the same one (updateStride, :924-925) */
    int src = 0;
    for (int l = r->leg_count - 1; l >= 0; --l)
      if (r->leg[l].stepper.step_state != FORCE_STOP) src = l;
    put3(o->stepper_walk_plane, r->leg[src].stepper.walk_plane);
    put3(o->stepper_walk_plane_normal, r->leg[src].stepper.walk_plane_normal);
  }
  put_pose7(o->origin_walk_plane_pose, r->origin_walk_plane_pose);
  put_pose7(o->manual_pose, r->manual_pose);
  put3(o->translation_velocity_input, r->translation_velocity_input);
  put3(o->rotation_velocity_input, r->rotation_velocity_input);
  put3(o->rotation_absement_error, r->rotation_absement_error);
  put3(o->rotation_velocity_error, r->rotation_velocity_error);
  put_quat4(o->auto_pose_rotation, r->auto_pose.r);
  put_pose7(o->current_pose, r->current_pose);
  put_pose7(o->odometry, r->odometry_ideal);
  put_pose7(o->tip_align_pose, r->tip_align_pose);
  put_pose7(o->origin_tip_align_pose, r->origin_tip_align_pose);
  o->walk_state = r->walk_state;
  o->legs_at_correct_phase = r->legs_at_correct_phase;
  o->legs_completed_first_step = r->legs_completed_first_step;
  o->return_to_default_attempted = r->return_to_default_attempted;
  o->auto_posing_state = r->auto_posing_state;
  o->pose_phase = r->pose_phase;
  for (int i = 0; i < r->n_auto_posers && i < SHC_MAX_AUTO_POSERS; ++i)
  {
    const auto_poser_t *ap = &r->auto_poser[i];
    o->auto_poser_flags[i] = (ap->start_check ? 1 : 0) | (ap->end_check_first ? 2 : 0) | (ap->end_check_second ? 4 : 0) | (ap->allow_posing ? 8 : 0);
  }
  for (int l = 0; l < r->leg_count; ++l)
  {
    const leg_t *leg = &r->leg[l];
    const stepper_t *s = &leg->stepper;
    shc_leg_snapshot *g = &o->leg[l];
    for (int j = 0; j < leg->joint_count; ++j)
    {
      g->joint_position[j] = leg->joint[j].desired_position;
      g->joint_velocity[j] = leg->joint[j].desired_velocity;
    }
    put3(g->walker_tip, s->current_tip_pose.p);
    put3(g->walker_tip_velocity, s->current_tip_velocity);
    put3(g->swing_origin_tip, s->swing_origin_tip_position);
    put3(g->swing_origin_tip_velocity, s->swing_origin_tip_velocity);
    put3(g->stance_origin_tip, s->stance_origin_tip_position);
    put3(g->default_tip, s->default_tip_pose.p);
    put3(g->target_tip, s->target_tip_pose.p);
    put3(g->stride_vector, s->stride_vector);
    if (tip_rotations_tracked(r, leg))
    {
      put3(g->walker_tip_direction, orc_quat_rotate(s->current_tip_pose.r, orc_v3_make(1, 0, 0)));
      put3(g->origin_tip_direction, orc_quat_rotate(s->origin_tip_pose.r, orc_v3_make(1, 0, 0)));
      g->tip_rotation_defined = !orc_quat_is_undefined(s->current_tip_pose.r);
      g->target_rotation_defined = !orc_quat_is_undefined(s->target_tip_pose.r);
      if (g->target_rotation_defined) put3(g->target_tip_direction, orc_quat_rotate(s->target_tip_pose.r, orc_v3_make(1, 0, 0)));
    }
    g->admittance_state[0] = leg->admittance_state[0];
    g->admittance_state[1] = leg->admittance_state[1];
    put3(g->admittance_delta, leg->admittance_delta);
    g->virtual_stiffness = leg->virtual_stiffness;
    put3(g->tip_force_calculated, leg->tip_force_calculated);
    g->swing_progress = s->swing_progress;
    g->stance_progress = s->stance_progress;
    g->step_state = s->step_state;
    g->phase = s->phase;
    g->at_correct_phase = s->at_correct_phase;
    g->completed_first_step = s->completed_first_step;
    g->negate_auto_pose = leg->poser.negate_auto_pose;
    g->ik_failed = leg->ik_failed;
    g->step_plane_defined = orc_pose_ne(leg->step_plane_pose, orc_pose_undefined());
    if (g->step_plane_defined) put3(g->step_plane_position, leg->step_plane_pose.p);
  }
  o->touchdown_detection = r->leg[0].stepper.touchdown_detection;
}

/* The reverse direction: a snapshot (e.g. taken from the engine) becomes the oracle's state.  Members the snapshot does
 * not carry are derived the way the reference derives them (joint transforms and the model tip by Leg::applyFK). */
void orc_set_state(orc_robot *r, const shc_instance_state *o)
{
  r->desired_linear_velocity[0] = o->desired_linear_velocity[0];
  r->desired_linear_velocity[1] = o->desired_linear_velocity[1];
  r->desired_angular_velocity = o->desired_angular_velocity;
  r->walk_plane = get3(o->walk_plane);
  r->walk_plane_normal = get3(o->walk_plane_normal);
  r->origin_walk_plane_pose = get_pose7(o->origin_walk_plane_pose);
  r->manual_pose = get_pose7(o->manual_pose);
  r->translation_velocity_input = get3(o->translation_velocity_input);
  r->rotation_velocity_input = get3(o->rotation_velocity_input);
  r->rotation_absement_error = get3(o->rotation_absement_error);
  r->rotation_velocity_error = get3(o->rotation_velocity_error);
  r->auto_pose.r = orc_quat_make(o->auto_pose_rotation[0], o->auto_pose_rotation[1], o->auto_pose_rotation[2], o->auto_pose_rotation[3]);
  r->current_pose = get_pose7(o->current_pose);
  r->odometry_ideal = get_pose7(o->odometry);
  r->tip_align_pose = get_pose7(o->tip_align_pose);
  r->origin_tip_align_pose = get_pose7(o->origin_tip_align_pose);
  r->walk_state = o->walk_state;
  r->legs_at_correct_phase = o->legs_at_correct_phase;
  r->legs_completed_first_step = o->legs_completed_first_step;
  r->return_to_default_attempted = o->return_to_default_attempted;
  r->auto_posing_state = o->auto_posing_state;
  r->pose_state = o->auto_posing_state;
  r->pose_phase = o->pose_phase;
  for (int i = 0; i < r->n_auto_posers && i < SHC_MAX_AUTO_POSERS; ++i)
  {
    auto_poser_t *ap = &r->auto_poser[i];
    ap->start_check = o->auto_poser_flags[i] & 1;
    ap->end_check_first = (o->auto_poser_flags[i] >> 1) & 1;
    ap->end_check_second = (o->auto_poser_flags[i] >> 2) & 1;
    ap->allow_posing = (o->auto_poser_flags[i] >> 3) & 1;
  }
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    stepper_t *s = &leg->stepper;
    const shc_leg_snapshot *g = &o->leg[l];
    for (int j = 0; j < leg->joint_count; ++j)
    {
      leg->joint[j].desired_position = g->joint_position[j];
      leg->joint[j].prev_desired_position = g->joint_position[j];
      leg->joint[j].desired_velocity = g->joint_velocity[j];
    }
    leg_apply_fk(r, leg);
    s->current_tip_pose.p = get3(g->walker_tip);
    s->current_tip_velocity = get3(g->walker_tip_velocity);
    s->swing_origin_tip_position = get3(g->swing_origin_tip);
    s->swing_origin_tip_velocity = get3(g->swing_origin_tip_velocity);
    s->stance_origin_tip_position = get3(g->stance_origin_tip);
    s->default_tip_pose.p = get3(g->default_tip);
    s->target_tip_pose.p = get3(g->target_tip);
    s->stride_vector = get3(g->stride_vector);
    s->walk_plane = get3(o->stepper_walk_plane);
    s->walk_plane_normal = get3(o->stepper_walk_plane_normal);
    if (tip_rotations_tracked(r, leg))
    { /* rotations rebuilt from their x axes the way updateTipRotation builds them (walk_controller.cpp:1224) */
      s->current_tip_pose.r = g->tip_rotation_defined ? orc_quat_from_two_vectors(orc_v3_make(1, 0, 0), get3(g->walker_tip_direction))
                                                      : ORC_UNDEFINED_ROTATION;
      s->origin_tip_pose.r = orc_quat_from_two_vectors(orc_v3_make(1, 0, 0), get3(g->origin_tip_direction));
      s->target_tip_pose.r = g->target_rotation_defined ? orc_quat_from_two_vectors(orc_v3_make(1, 0, 0), get3(g->target_tip_direction))
                                                        : ORC_UNDEFINED_ROTATION;
    }
    leg->admittance_state[0] = g->admittance_state[0];
    leg->admittance_state[1] = g->admittance_state[1];
    leg->admittance_delta = get3(g->admittance_delta);
    leg->virtual_stiffness = g->virtual_stiffness;
    leg->tip_force_calculated = get3(g->tip_force_calculated);
    s->swing_progress = g->swing_progress;
    s->stance_progress = g->stance_progress;
    s->step_state = g->step_state;
    s->phase = g->phase;
    s->step_progress = (double)s->phase / r->step.period; /* (not in the record: what the last iteratePhase left, walk_controller.cpp:878) */
    s->at_correct_phase = g->at_correct_phase;
    s->completed_first_step = g->completed_first_step;
    leg->poser.negate_auto_pose = g->negate_auto_pose;
    leg->ik_failed = g->ik_failed;
    leg->step_plane_pose = orc_pose_undefined();
    if (g->step_plane_defined) leg->step_plane_pose = orc_pose_make(get3(g->step_plane_position), orc_quat_identity());
    s->touchdown_detection = o->touchdown_detection;
  }
}

void orc_batch_get_state(orc_batch *b, shc_instance_state *states)
{
  for (int64_t i = 0; i < b->n; ++i) orc_get_state(&b->robots[i], &states[i]);
}
void orc_batch_set_state(orc_batch *b, const shc_instance_state *states)
{
  for (int64_t i = 0; i < b->n; ++i) orc_set_state(&b->robots[i], &states[i]);
}

/* ------------------------------------------------------------------------------------ per-leg Leg methods on a robot
 * (model.h:448-492) - counterparts of the engine's shc_leg_* entry points */
void orc_leg_set_desired_tip_pose(orc_robot *r, int l, const double *pose7, int apply_delta)
{
  leg_set_desired_tip_pose(&r->leg[l], pose7 ? get_pose7(pose7) : orc_pose_undefined(), apply_delta);
}
void orc_leg_solve_ik(orc_robot *r, int l, const double delta[6], int solve_rotation, double *dq)
{
  leg_solve_ik(&r->leg[l], delta, solve_rotation, dq);
}
double orc_leg_update_joint_positions(orc_robot *r, int l, const double *dq, int simulation)
{
  return leg_update_joint_positions(r, &r->leg[l], dq, simulation);
}
double orc_leg_apply_ik(orc_robot *r, int l, int simulation)
{
  if (!simulation) r->leg[l].ik_failed = 0;
  return leg_apply_ik(r, &r->leg[l], simulation);
}
/* Leg::applyFK(set_current, use_actual) (model.cpp:945-988): joint_position != NULL is use_actual with set_current = false */
void orc_leg_apply_fk(orc_robot *r, int l, const double *joint_position, double pose7[7])
{
  if (!joint_position)
  {
    put_pose7(pose7, leg_apply_fk(r, &r->leg[l]));
    return;
  }
  leg_t *copy = (leg_t *)malloc(sizeof(leg_t));
  *copy = r->leg[l];
  for (int j = 0; j < copy->joint_count; ++j) copy->joint[j].desired_position = joint_position[j];
  put_pose7(pose7, leg_apply_fk(r, copy));
  free(copy);
}

/* LegPoser::stepToPosition / transitionConfiguration on leg l of a robot */
int orc_leg_step_to_position(orc_robot *r, int l, const double *target_tip_pose7, const double *target_pose7, double lift_height,
                             double time_to_step, int apply_delta, double tip_pose7[7])
{
  leg_t *leg = &r->leg[l];
  int progress = leg_poser_step_to_position(r, leg, target_tip_pose7 ? get_pose7(target_tip_pose7) : orc_pose_undefined(), get_pose7(target_pose7),
                                            lift_height, time_to_step, apply_delta);
  put_pose7(tip_pose7, leg->poser.current_tip_pose);
  return progress;
}
int orc_leg_transition_configuration(orc_robot *r, int l, const double *desired_configuration, double transition_time)
{
  leg_t *leg = &r->leg[l];
  for (int j = 0; j < leg->joint_count; ++j) leg->poser.desired_configuration[j] = desired_configuration[j];
  leg->poser.has_desired_configuration = 1;
  return leg_poser_transition_configuration(r, leg, transition_time);
}
/* Direct start-up, loop by loop: orc_startup_begin = the UNKNOWN -> PACKED loop + the START request; every orc_startup_step is
 * one StateController::loop() of the PACKED -> READY transition (returns PoseController::directStartup's progress);
 * orc_startup_finish = the loop that enters RUNNING (and runs the first control cycle). */
void orc_startup_begin(orc_robot *r)
{
  r->transition_state_flag = 1;
  state_loop(r);
  r->new_robot_state = RS_READY;
  r->transition_state_flag = 1;
}
int orc_startup_step(orc_robot *r)
{
  state_loop(r);
  return r->robot_state == RS_READY ? PROGRESS_COMPLETE : r->startup_progress;
}
void orc_startup_finish(orc_robot *r)
{
  r->new_robot_state = RS_RUNNING;
  r->transition_state_flag = 1;
  state_loop(r);
}

/* ---- start-up / shut-down sequences (start_up_sequence: true).  orc_sequence_begin = state.init() + initModel(false) with the
 * joint states `q` ([legs][dof], offsets removed) the motors reported (main.cpp:99-100, model.cpp:286-305) and the UNKNOWN ->
 * READY estimate (state_controller.cpp:236-251).  orc_sequence_prologue = the posing / admittance part of one
 * StateController::loop() (:165-181); orc_execute_sequence / orc_step_to_new_stance = one call of the PoseController method
 * (transitionRobotState calls executeSequence once per loop for START_UP and - through runningState - twice per loop for
 * SHUT_DOWN, :186-192, :384-388).  orc_sequence_finish_startup = what follows a completed START_UP (:305-313) and the
 * runningState() of the same loop. */
void orc_sequence_begin(orc_robot *r, const double *q)
{
  { /* a robot as the constructors leave it (the batch helpers hand out robots that already ran the direct start-up) */
    shc_params params = r->params;
    orc_robot *fresh = orc_create(&params);
    memcpy(r, fresh, sizeof(orc_robot));
    orc_destroy(fresh);
  }
  int k = 0;
  for (int l = 0; l < r->leg_count; ++l)
  {
    leg_t *leg = &r->leg[l];
    for (int j = 0; j < leg->joint_count; ++j) leg->joint[j].current_position = q[k++];
    leg->current_tip_pose = orc_pose_undefined();
    leg_init(r, leg, 0);
  }
  r->robot_state = RS_READY;
  r->new_robot_state = RS_RUNNING;
  r->transition_state_flag = 1;
}
void orc_sequence_prologue(orc_robot *r)
{
  poser_update_current_pose(r, r->robot_state);
  r->pose_state = r->auto_posing_state;
  if (r->params.admittance_control)
  {
    if (r->walk_state != STOPPED && r->params.dynamic_stiffness) admittance_update_stiffness(r);
    admittance_update_admittance(r);
  }
}
int orc_execute_sequence(orc_robot *r, int sequence) { return poser_execute_sequence(r, sequence); }
int orc_step_to_new_stance(orc_robot *r) { return poser_step_to_new_stance(r); }
int orc_sequence_failed(const orc_robot *r) { return r->sequence_failed; }
int orc_pack_legs(orc_robot *r, const double *packed_positions, int number_pack_steps, double time_to_pack)
{
  return poser_pack_legs(r, packed_positions, number_pack_steps, time_to_pack, 0);
}
int orc_unpack_legs(orc_robot *r, const double *packed_positions, int number_pack_steps, double time_to_unpack)
{
  return poser_pack_legs(r, packed_positions, number_pack_steps, time_to_unpack, 1);
}
void orc_sequence_finish_startup(orc_robot *r)
{
  walker_init(r);
  for (int l = 0; l < r->leg_count; ++l) leg_update_default_configuration(&r->leg[l]);
  model_generate_workspaces(r);
  walker_generate_walkspace(r);
  r->robot_state = RS_RUNNING;
  r->transition_state_flag = 0;
  state_running_state(r);
}
void orc_sequence_finish_shutdown(orc_robot *r) { r->robot_state = RS_READY; r->new_robot_state = RS_RUNNING; }

/* ---- planner mode.  plannerModeCallback (state_controller.cpp:1262-1281), targetConfigurationCallback (:1683-1687; rows [legs][dof],
 * a leg whose first entry is NaN is not named in the message), targetBodyPoseCallback (:1691-1702); orc_execute_plan = one
 * StateController::loop() in planner mode: the posing part (:165-181), then runningState -> executePlan (:401-405; a robot that
 * is still walking goes on with its normal cycle, velocity inputs zeroed, :691-697). */
void orc_set_planner_mode(orc_robot *r, int on)
{
  if (r->robot_state != RS_RUNNING || on == r->planner_mode) return;
  r->planner_mode = on;
  if (on) r->plan_step = 0;
}
void orc_set_target_configuration(orc_robot *r, const double *configuration)
{
  int k = 0;
  for (int l = 0; l < r->leg_count; ++l)
  {
    r->target_configuration_named[l] = !isnan(configuration[k]);
    for (int j = 0; j < r->leg[l].joint_count; ++j) r->target_configuration[l][j] = configuration[k++];
  }
  r->target_configuration_acquired = 1;
}
void orc_set_target_body_pose(orc_robot *r, const double *pose7)
{
  r->target_body_pose = pose_from7(pose7);
  r->target_body_pose_acquired = 1;
}
int orc_execute_plan(orc_robot *r)
{
  orc_sequence_prologue(r);
  int result = state_execute_plan(r);
  if (result == -1) state_running_state(r);
  return result;
}
int orc_get_plan_step(const orc_robot *r) { return r->plan_step; }

/* ---- manual leg manipulation.  orc_leg_state_toggle = one StateController::loop() with the toggle request for `leg` pending:
 * the posing part (:165-181), then runningState -> legStateToggle (:396-400; no tip update while the robot is STOPPED); while
 * the robot is still walking the request only zeroes the velocity inputs and the loop runs its normal cycle (:641-645, :421-446). */
int orc_leg_state_toggle(orc_robot *r, int leg)
{
  orc_sequence_prologue(r);
  int result = state_leg_state_toggle(r, leg);
  if (result == -1) state_running_state(r);
  return result;
}
int orc_get_leg_manipulation_state(const orc_robot *r, int leg) { return r->leg[leg].leg_state; }
/* primaryLegSelectionCallback / primaryTipVelocityInputCallback / primaryPoseInput ... (state_controller.cpp:1247-1330) */
void orc_set_manual_inputs(orc_robot *r, int primary_leg, const double *primary_tip_velocity, const double *primary_tip_position,
                           int secondary_leg, const double *secondary_tip_velocity, const double *secondary_tip_position)
{
  r->primary_leg_selection = primary_leg;
  r->secondary_leg_selection = secondary_leg;
  r->primary_tip_velocity_input = primary_tip_velocity ? orc_v3_make(primary_tip_velocity[0], primary_tip_velocity[1], primary_tip_velocity[2]) : orc_v3_make(0, 0, 0);
  r->secondary_tip_velocity_input = secondary_tip_velocity ? orc_v3_make(secondary_tip_velocity[0], secondary_tip_velocity[1], secondary_tip_velocity[2]) : orc_v3_make(0, 0, 0);
  r->primary_pose_input = orc_pose_identity();
  r->secondary_pose_input = orc_pose_identity();
  r->primary_pose_input.p = primary_tip_position ? orc_v3_make(primary_tip_position[0], primary_tip_position[1], primary_tip_position[2]) : orc_v3_make(0, 0, 0);
  r->secondary_pose_input.p = secondary_tip_position ? orc_v3_make(secondary_tip_position[0], secondary_tip_position[1], secondary_tip_position[2]) : orc_v3_make(0, 0, 0);
}

/* ------------------------------------------------------------------------------------ unit-level entry points */
void orc_test_generate_step_cycle(const shc_params *p, shc_step_cycle *out) { *out = generate_step_cycle(p); }
void orc_test_quat_to_euler(const double q[4], int intrinsic, double out[3])
{
  orc_v3 e = orc_quat_to_euler(orc_quat_make(q[0], q[1], q[2], q[3]), intrinsic);
  out[0] = e.x; out[1] = e.y; out[2] = e.z;
}
void orc_test_euler_to_quat(const double e[3], int intrinsic, double out[4])
{
  orc_quat q = orc_euler_to_quat(orc_v3_make(e[0], e[1], e[2]), intrinsic);
  out[0] = q.w; out[1] = q.x; out[2] = q.y; out[3] = q.z;
}
void orc_test_from_two_vectors(const double a[3], const double b[3], double out[4])
{
  orc_quat q = orc_quat_from_two_vectors(orc_v3_make(a[0], a[1], a[2]), orc_v3_make(b[0], b[1], b[2]));
  out[0] = q.w; out[1] = q.x; out[2] = q.y; out[3] = q.z;
}
void orc_test_slerp(const double a[4], double t, const double b[4], double out[4])
{
  orc_quat q = orc_quat_slerp(orc_quat_make(a[0], a[1], a[2], a[3]), t, orc_quat_make(b[0], b[1], b[2], b[3]));
  out[0] = q.w; out[1] = q.x; out[2] = q.y; out[3] = q.z;
}
void orc_test_quat_from_matrix(const double m[9], double out[4])
{
  orc_m3 mm;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) mm.m[i][j] = m[i * 3 + j];
  orc_quat q = orc_quat_from_matrix(&mm);
  out[0] = q.w; out[1] = q.x; out[2] = q.y; out[3] = q.z;
}
int orc_test_lu_inverse(const double *a, int n, double *inv) { return orc_lu_inverse(a, n, inv); }
void orc_test_dh_matrix(double d, double theta, double r, double alpha, double out[16])
{
  orc_m4 m = orc_create_dh_matrix(d, theta, r, alpha);
  memcpy(out, &m.m[0][0], sizeof(double) * 16);
}
void orc_test_quartic_bezier(const double nodes[15], double t, double out[3], double out_dot[3])
{
  orc_v3 p[5];
  for (int k = 0; k < 5; ++k) p[k] = orc_v3_make(nodes[3 * k], nodes[3 * k + 1], nodes[3 * k + 2]);
  orc_v3 a = orc_quartic_bezier(p, t), b = orc_quartic_bezier_dot(p, t);
  out[0] = a.x; out[1] = a.y; out[2] = a.z;
  out_dot[0] = b.x; out_dot[1] = b.y; out_dot[2] = b.z;
}
void orc_test_leg_fk(const shc_params *p, int l, const double *q, double tip_pos[3], double tip_quat[4])
{
  orc_robot *r = orc_create(p);
  leg_t *leg = &r->leg[l];
  for (int j = 0; j < leg->joint_count; ++j) leg->joint[j].desired_position = q[j];
  orc_pose t = leg_apply_fk(r, leg);
  put3(tip_pos, t.p);
  tip_quat[0] = t.r.w; tip_quat[1] = t.r.x; tip_quat[2] = t.r.y; tip_quat[3] = t.r.z;
  orc_destroy(r);
}
double orc_test_leg_ik_step(const shc_params *p, int l, const double *q, const double *qd, const double desired[3],
                            int simulation, double *q_out, double *qd_out, double tip_out[3])
{
  orc_robot *r = orc_create(p);
  leg_t *leg = &r->leg[l];
  for (int j = 0; j < leg->joint_count; ++j)
  {
    leg->joint[j].desired_position = q[j];
    leg->joint[j].desired_velocity = qd ? qd[j] : 0.0;
  }
  leg_apply_fk(r, leg);
  leg_set_desired_tip_pose(leg, orc_pose_make(orc_v3_make(desired[0], desired[1], desired[2]), ORC_UNDEFINED_ROTATION), 0);
  double res = leg_apply_ik(r, leg, simulation);
  for (int j = 0; j < leg->joint_count; ++j)
  {
    q_out[j] = leg->joint[j].desired_position;
    if (qd_out) qd_out[j] = leg->joint[j].desired_velocity;
  }
  if (tip_out) put3(tip_out, leg->current_tip_pose.p);
  orc_destroy(r);
  return res;
}
void orc_test_admittance(const shc_params *p, double state[2], const double force[3], double delta_out[3])
{
  admittance_leg(p, state, orc_v3_make(force[0], force[1], force[2]), delta_out);
}

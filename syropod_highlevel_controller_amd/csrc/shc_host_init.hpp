// shc_host_init.hpp — init chain of the engine (product code; shares shc_leg.hpp with the HIP kernels; the CPU oracle has
// its own, independent restatement).  Every function except the long-double admittance map is host + device: one
// morphology is initialised on the host (shc_generate_tables, ~1 ms), many at once by init_chain kernels
// (shc_generate_tables_batch: one thread per (morphology, leg) for the start-up solve + workspace search).
//
// Produces the per-(morphology, gait) tables the cycle kernel consumes.  Reference chain restated
// (OpenSHC v0.5.11):
//   WalkController::generateStepCycle          src/walk_controller.cpp:365-410
//   PoseController::directStartup              src/pose_controller.cpp:463-517  (LegPoser::stepToPosition :1571-1712)
//   Model::updateDefaultConfiguration          src/model.cpp:108, :593
//   Leg::generateWorkspace (simple workspace)  src/model.cpp:309-510
//   WalkController::generateWalkspace          src/walk_controller.cpp:57-227
//   WalkController::generateLimits             src/walk_controller.cpp:231-361
//   PoseController::setAutoPoseParams          src/pose_controller.cpp:44-106
//   AdmittanceController::updateAdmittance     src/admittance_controller.cpp:22-63 (collapsed to an affine map)
#pragma once

#include "../../include/shc_batch.h"
#include "shc_leg.hpp"

#include <cmath>
#include <cstring>

#define SHC_HDI __host__ __device__ inline

namespace shc {
namespace hostinit {

SHC_HDI int imax(int a, int b) { return a > b ? a : b; }
SHC_HDI int iabs(int a) { return a < 0 ? -a : a; }
SHC_HDI double dmin(double a, double b) { return a < b ? a : b; }
SHC_HDI double dmax(double a, double b) { return a > b ? a : b; }

constexpr int kBearingStep = 45;          // model.h:22
constexpr double kMaxPositionDelta = 0.002; // model.h:23
constexpr double kMaxWorkspaceRadius = 1.0; // model.h:24
constexpr int kWorkspaceLayers = 10;      // model.h:25

template <int NJ>
SHC_HDI void fill_leg_const(const shc_params &p, int l, LegConst<NJ> &lc) {
  __builtin_memset(&lc, 0, sizeof lc);
  const shc_link_params &b = p.link[l][0];
  // createDHMatrix (standard_includes.h:466) of the base link: joint 1's constant transform
  double ct = cos(b.theta), st = sin(b.theta), ca = cos(b.alpha), sa = sin(b.alpha);
  lc.r1[0] = ct; lc.r1[1] = -st * ca; lc.r1[2] = st * sa;
  lc.r1[3] = st; lc.r1[4] = ct * ca;  lc.r1[5] = -ct * sa;
  lc.r1[6] = 0;  lc.r1[7] = sa;       lc.r1[8] = ca;
  lc.p1[0] = b.r * ct; lc.p1[1] = b.r * st; lc.p1[2] = b.d;
  for (int k = 0; k < NJ; ++k) {
    if (k >= p.leg_dof[l]) { // padding behind the tip of a leg with fewer joints than the robot's longest: an identity transform, locked at 0
      lc.link_sa[k] = 0.0, lc.link_ca[k] = 1.0;
      lc.jvmax[k] = 1.0;
      lc.jw_vrange[k] = kJointLimitCostWeight / 2.0;
      continue; // (d, r, theta, limits, centre, jw_range, jactive stay 0)
    }
    lc.jactive[k] = 1.0;
    const shc_link_params &lk = p.link[l][k + 1];
    lc.link_d[k] = lk.d;
    lc.link_r[k] = lk.r;
    lc.link_sa[k] = sin(lk.alpha);
    lc.link_ca[k] = cos(lk.alpha);
    lc.link_th[k] = lk.theta;
    const shc_joint_params &j = p.joint[l][k];
    lc.jmin[k] = j.min;
    lc.jmax[k] = j.max;
    lc.jvmax[k] = j.max_vel;
    double range = j.max - j.min;
    lc.jcentre[k] = j.min + range / 2.0;
    lc.jw_range[k] = range != 0.0 ? kJointLimitCostWeight / range : 0.0;
    lc.jw_vrange[k] = kJointLimitCostWeight / (2 * j.max_vel);
  }
  lc.stance_x = p.stance_position[l][0];
  lc.stance_y = p.stance_position[l][1];
}

// 30 classical RK4 steps (h = integrator_step_time / 30) of  x' = A x + b u,  A = [[0,1],[-k/m,-c/m]], b = [0,-1/m]
// are one affine map x <- M x + g u, because u is constant during the call (admittance_controller.cpp:33-52).
inline void admittance_map(const shc_params &p, double &m00, double &m01, double &m10, double &m11, double &g0, double &g1) {
  typedef long double ld;
  ld mass = p.virtual_mass, k = p.virtual_stiffness;
  ld c = (ld)p.virtual_damping_ratio * 2 * sqrtl(mass * k);
  ld h = (ld)p.integrator_step_time / 30;
  ld A[2][2] = {{0, 1}, {-k / mass, -c / mass}};
  ld H[2][2] = {{h * A[0][0], h * A[0][1]}, {h * A[1][0], h * A[1][1]}};
  auto mul = [](const ld (&x)[2][2], const ld (&y)[2][2], ld (&z)[2][2]) {
    ld t[2][2];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) t[i][j] = x[i][0] * y[0][j] + x[i][1] * y[1][j];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) z[i][j] = t[i][j];
  };
  ld H2[2][2], H3[2][2], H4[2][2];
  mul(H, H, H2);
  mul(H2, H, H3);
  mul(H3, H, H4);
  ld P[2][2], S[2][2]; // P = I + H + H^2/2 + H^3/6 + H^4/24 ; S = I + H/2 + H^2/6 + H^3/24
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      ld I = i == j ? 1 : 0;
      P[i][j] = I + H[i][j] + H2[i][j] / 2 + H3[i][j] / 6 + H4[i][j] / 24;
      S[i][j] = I + H[i][j] / 2 + H2[i][j] / 6 + H3[i][j] / 24;
    }
  ld bvec[2] = {0, -1 / mass};
  ld g1v[2] = {h * (S[0][0] * bvec[0] + S[0][1] * bvec[1]), h * (S[1][0] * bvec[0] + S[1][1] * bvec[1])};
  ld M[2][2] = {{1, 0}, {0, 1}}, G[2] = {0, 0};
  for (int s = 0; s < 30; ++s) { // x_{n+1} = P x_n + g1 u
    ld ng0 = P[0][0] * G[0] + P[0][1] * G[1] + g1v[0], ng1 = P[1][0] * G[0] + P[1][1] * G[1] + g1v[1];
    G[0] = ng0;
    G[1] = ng1;
    mul(P, M, M);
  }
  m00 = (double)M[0][0]; m01 = (double)M[0][1]; m10 = (double)M[1][0]; m11 = (double)M[1][1];
  g0 = (double)G[0]; g1 = (double)G[1];
}

SHC_HDI shc_step_cycle generate_step_cycle(const shc_params &p) {
  shc_step_cycle s;
  s.stance_end = int(p.stance_phase * 0.5);
  s.swing_start = s.stance_end;
  s.swing_end = s.swing_start + p.swing_phase;
  s.stance_start = s.swing_end;
  int base = p.stance_phase + p.swing_phase;
  double swing_ratio = double(p.swing_phase) / double(base);
  double raw = ((1.0 / p.step_frequency) / p.time_delta) / swing_ratio;
  s.period = round_to_even_int(raw / base) * base;
  s.frequency = 1.0 / (s.period * p.time_delta);
  int normaliser = s.period / base;
  s.stance_end *= normaliser;
  s.swing_start *= normaliser;
  s.swing_end *= normaliser;
  s.stance_start *= normaliser;
  s.stance_period = mod_i(s.stance_end - s.stance_start, s.period);
  s.swing_period = s.swing_end - s.swing_start;
  return s;
}

// A leg being driven by single DLS steps on the host (start-up solve and workspace search).
template <int NJ>
struct HostLeg {
  LegConst<NJ> lc;
  double q[NJ], qd[NJ], dflt[NJ];
  Chain<NJ> ch;
  V3 tip; // robot frame
  SHC_HDI void fk() {
    fk_chain<NJ>(lc, q, ch);
    tip = tip_robot_frame(lc, ch.pe);
  }
  SHC_HDI void reset_to_default() { // Leg::init(true) (model.cpp:286-305)
    for (int j = 0; j < NJ; ++j) {
      q[j] = dflt[j];
      qd[j] = 0.0;
    }
    fk();
  }
  // Leg::applyIK(simulation = true) towards `desired` (model.cpp:861-941); returns the ik result.  `desired_dir` (robot
  // frame, unit) is the x axis of the desired tip rotation when has_rotation (rotation-constrained IK, :880-900).
  SHC_HDI double ik(V3 desired, const shc_params &p, bool has_rotation = false, V3 desired_dir = V3{1, 0, 0}) {
    const V3 current_dir = ch.xe; // leg-frame direction of the tip BEFORE this call's updates (the reference reads it first, :866)
    double dq[NJ];
    ik_step<NJ, true>(lc, ch, q, qd, desired, dq); // IEEE division on host and device alike
    if (has_rotation) {
      update_joints<NJ>(lc, dq, p.time_delta, 1.0 / p.time_delta, false, p.clamp_joint_positions != 0, q, qd); // simulation = true (:883)
      fk();
      V3 lin[NJ];
      jacobian_columns<NJ>(ch, lin);
      ik_step_rotation<NJ, true>(lc, ch, lin, q, qd, tip_rotation_delta(current_dir, base_rotate_inv(lc, desired_dir)), dq);
    }
    double prox = update_joints<NJ>(lc, dq, p.time_delta, 1.0 / p.time_delta, false, p.clamp_joint_positions != 0, q, qd);
    fk();
    V3 e = tip - desired;
    double result = prox;
    if (fabs(e.x) > kIkTolerance || fabs(e.y) > kIkTolerance || fabs(e.z) > kIkTolerance) result = 0.0;
    if (has_rotation && result == 0.0) return ik(desired, p); // retry with the rotation unconstrained (:932-936)
    return result;
  }
};

// PoseController::directStartup's simulated solve for one leg: LegPoser::stepToPosition (lift 0, time_to_start)
// towards the default tip pose with the body easing to `body` + one DLS step per iteration.
// identity tip direction of a gravity-aligned leg: x axis of FromTwoVectors(x, -z) (walk_controller.cpp:37-41)
SHC_HDI V3 gravity_aligned_direction() {
  return rotate(correct_rotation(from_two_vectors(V3{1, 0, 0}, V3{-0.0, -0.0, -1.0}), quat_identity()), V3{1, 0, 0});
}
SHC_HDI bool tips_rotation_constrained(const shc_params &p, int nj) { return nj > 3 && p.gravity_aligned_tips != 0; } // the identity tip rotation is defined
// LegStepper::updateTipRotation has something to do: the target tip rotation is defined from the start (gravity-aligned tips) or an
// externally requested target may define it (rough terrain mode, walk_controller.cpp:1068-1071)
SHC_HDI bool tips_rotation_tracked(const shc_params &p, int nj) { return nj > 3 && (p.gravity_aligned_tips != 0 || p.rough_terrain_mode != 0); }

template <int NJ>
SHC_HDI void startup_solve(const shc_params &p, HostLeg<NJ> &leg, V3 default_tip, const Pose &body) {
  leg.reset_to_default();
  V3 origin = leg.tip;
  V3 delta = origin - inverse_transform_vector(body, default_tip);
  int joints_of_this_leg = 0; // (a leg padded up to the robot's longest leg decides by ITS OWN joint count, walk_controller.cpp:37)
  for (int k = 0; k < NJ; ++k) joints_of_this_leg += leg.lc.jactive[k] != 0.0 ? 1 : 0;
  const bool rot = tips_rotation_constrained(p, joints_of_this_leg);
  const V3 origin_dir = base_rotate(leg.lc, leg.ch.xe), target_dir = gravity_aligned_direction();
  bool transition_rotation = false; // pose_controller.cpp:1594-1601
  if (rot) transition_rotation = norm(angle_axis_vector(from_two_vectors(origin_dir, target_dir))) > kJointTolerance;
  if (!(norm(delta) > kTipTolerance) && !transition_rotation) return; // already there (pose_controller.cpp:1603-1608)
  int num = imax(1, round_to_int(p.time_to_start / p.time_delta));
  double dt = 1.0 / num;
  int half = num / 2;
  V3 o2t = origin - default_tip;
  V3 prim[5] = {origin, origin, origin, default_tip + o2t * 0.75, default_tip + o2t * 0.5};
  V3 sec[5] = {default_tip + o2t * 0.5, default_tip + o2t * 0.25, default_tip, default_tip, default_tip};
  for (int it = 1; it <= num; ++it) {
    double ratio = double(it - 1) / double(num);
    Pose dp = interpolate_pose(pose_identity(), smooth_step(ratio), body);
    int sic = (it + (num - 1)) % num + 1;
    V3 np = sic <= half ? quartic_bezier(prim, sic * dt * 2.0) : quartic_bezier(sec, (sic - half) * dt * 2.0);
    // tip direction eased from the current one to the target (:1633-1640); the pose interpolation moves positions only
    V3 dir = rot ? normalized(lerp3(origin_dir, target_dir, smooth_step(ratio))) : V3{1, 0, 0};
    leg.ik(inverse_transform_vector(dp, np), p, rot, dir);
  }
}

// Leg::generateWorkspace, simple (single plane z = 0) workspace.  radius[b], b = bearing / 45.
template <int NJ>
SHC_HDI void generate_workspace(const shc_params &p, HostLeg<NJ> &leg, V3 identity_tip_body, double (&radius)[SHC_N_BEARINGS],
                                int first_bearing = 1, int last_bearing = 8) { // bearing indices (x 45 degrees) searched by this call
  leg.reset_to_default();
  if (norm(identity_tip_body - leg.tip) > kIkTolerance) { // model.cpp:349-353
    for (int b = 0; b < SHC_N_BEARINGS; ++b) radius[b] = 0.0;
    return;
  }
  const bool all = first_bearing == 1 && last_bearing == 8;
  if (all)
    for (int b = 0; b < SHC_N_BEARINGS; ++b) radius[b] = kMaxWorkspaceRadius;
  // track from the default-configuration tip to the identity tip position (model.cpp:397-404), then re-base defaults
  {
    int n = imax(1, round_to_int((kMaxWorkspaceRadius / kWorkspaceLayers) / kMaxPositionDelta));
    V3 o = leg.tip, t = identity_tip_body;
    bool ok = true;
    for (int it = 1; it <= n && ok; ++it) {
      double i = double(it) / n;
      ok = leg.ik(o * (1.0 - i) + t * i, p) != 0.0;
    }
    for (int j = 0; j < NJ; ++j) leg.dflt[j] = leg.q[j]; // updateDefaultConfiguration (model.cpp:465)
  }
  for (int bearing = kBearingStep * first_bearing; bearing <= kBearingStep * last_bearing; bearing += kBearingStep) {
    leg.reset_to_default();
    int n = round_to_int(kMaxWorkspaceRadius / kMaxPositionDelta);
    V3 o = identity_tip_body, t = o;
    t.x += kMaxWorkspaceRadius * cos(deg2rad(bearing));
    t.y += kMaxWorkspaceRadius * sin(deg2rad(bearing));
    for (int it = 1; it <= n; ++it) {
      double i = double(it) / n;
      if (leg.ik(o * (1.0 - i) + t * i, p) == 0.0) break;
    }
    radius[bearing / kBearingStep] = norm(leg.tip - identity_tip_body);
  }
  if (last_bearing == 8) radius[0] = radius[360 / kBearingStep];
}

// Leg::generateWorkspace in rough terrain mode (model.cpp:309-510, simple_workspace == false): the layered workspace.  The tip
// is first driven straight down, then straight up from its identity position until a DLS step fails (lower / upper plane
// heights); the span is cut into WORKSPACE_LAYERS layers and every plane from the top one down is searched like the simple
// workspace, each starting from the configuration reached by tracking to that plane's origin.  What the rest of the init
// chain consumes is Leg::getWorkplane(0) (model.cpp:514-550) - WalkController::generateWalkspace asks for the plane at the
// default tips' height shift, which is zero right after start-up (walk_controller.cpp:117-121) - written to radius[].
// The planes of a leg's layered workspace (Leg::workspace_ in rough terrain mode), kept for LegStepper::calculateStanceSpanChange,
// which interpolates between the planes bounding the default tip's current height (walk_controller.cpp:949-980).
constexpr int kMaxWorkspacePlanes = kWorkspaceLayers + 4;
struct LayeredPlanes {
  int n = 0;
  double height[kMaxWorkspacePlanes];
  double radius[kMaxWorkspacePlanes][SHC_N_BEARINGS];
};
template <int NJ>
SHC_HDI void generate_workspace_layered(const shc_params &p, HostLeg<NJ> &leg, V3 identity_tip_body, double (&radius)[SHC_N_BEARINGS],
                                        LayeredPlanes *keep = nullptr) {
  constexpr int kMaxPlanes = kMaxWorkspacePlanes;
  double height[kMaxPlanes], plane[kMaxPlanes][SHC_N_BEARINGS];
  int planes = 0;
  if (keep) keep->n = 0;
  for (int b = 0; b < SHC_N_BEARINGS; ++b) radius[b] = 0.0;
  leg.reset_to_default();
  if (norm(identity_tip_body - leg.tip) > kIkTolerance) return; // zero workspace (model.cpp:349-353)
  auto add_plane = [&](double h, double fill) { // std::map::insert: an existing key is kept
    for (int k = 0; k < planes; ++k)
      if (height[k] == h) return k;
    if (planes == kMaxPlanes) return planes - 1;
    height[planes] = h;
    for (int b = 0; b < SHC_N_BEARINGS; ++b) plane[planes][b] = fill;
    return planes++;
  };
  const int n_line = round_to_int(kMaxWorkspaceRadius / kMaxPositionDelta);
  auto vertical_limit = [&](double direction) { // distance the tip can be moved along +-z from the identity position
    leg.reset_to_default();
    const V3 o = identity_tip_body, t = o + V3{0, 0, direction * kMaxWorkspaceRadius};
    for (int it = 1; it <= n_line; ++it) {
      const double i = double(it) / n_line;
      if (leg.ik(o * (1.0 - i) + t * i, p) == 0.0) break;
    }
    return norm(leg.tip - identity_tip_body);
  };
  const double min_h = -vertical_limit(-1.0);
  add_plane(min_h, 0.0);
  const double max_h = vertical_limit(1.0);
  const double delta = (max_h - min_h) / kWorkspaceLayers;
  double h = int(fabs(max_h) / delta) * delta;
  add_plane(max_h, 0.0);
  int cur = add_plane(h, kMaxWorkspaceRadius);
  while (true) {
    const V3 origin = identity_tip_body + V3{0, 0, h};
    { // track from the (re-based) default configuration to this plane's origin, then re-base again (model.cpp:397-404, :465)
      leg.reset_to_default();
      const int n = imax(1, round_to_int(delta / kMaxPositionDelta));
      const V3 o = leg.tip;
      for (int it = 1; it <= n; ++it) {
        const double i = double(it) / n;
        if (leg.ik(o * (1.0 - i) + origin * i, p) == 0.0) break;
      }
      for (int j = 0; j < NJ; ++j) leg.dflt[j] = leg.q[j];
    }
    for (int bearing = kBearingStep; bearing <= 360; bearing += kBearingStep) {
      leg.reset_to_default();
      V3 t = origin;
      t.x += kMaxWorkspaceRadius * cos(deg2rad(bearing));
      t.y += kMaxWorkspaceRadius * sin(deg2rad(bearing));
      for (int it = 1; it <= n_line; ++it) {
        const double i = double(it) / n_line;
        if (leg.ik(origin * (1.0 - i) + t * i, p) == 0.0) break;
      }
      plane[cur][bearing / kBearingStep] = norm(leg.tip - origin);
    }
    plane[cur][0] = plane[cur][360 / kBearingStep];
    h -= delta;
    if (!(h >= min_h)) break;
    cur = add_plane(h, kMaxWorkspaceRadius);
  }
  if (keep) {
    keep->n = planes;
    for (int k = 0; k < planes; ++k) {
      keep->height[k] = height[k];
      for (int b = 0; b < SHC_N_BEARINGS; ++b) keep->radius[k][b] = plane[k][b];
    }
  }
  // Leg::getWorkplane(0.0): interpolate between the planes bounding height 0 (heights rounded to 3 decimals, :532-533)
  int lower = -1, upper = -1;
  for (int k = 0; k < planes; ++k) {
    if (height[k] > 0.0 && (upper < 0 || height[k] < height[upper])) upper = k;
    if (height[k] <= 0.0 && (lower < 0 || height[k] > height[lower])) lower = k;
  }
  if (lower < 0 || upper < 0) return;
  const double uh = round_to_int(height[upper] * pow(10, 3)) / pow(10, 3), lh = round_to_int(height[lower] * pow(10, 3)) / pow(10, 3);
  const double i = (0.0 - lh) / (uh - lh);
  for (int b = 0; b < SHC_N_BEARINGS; ++b) radius[b] = plane[lower][b] * (1.0 - i) + plane[upper][b] * i;
}

SHC_HDI V3 rot_z(double ang, V3 v) { // Eigen::AngleAxisd(ang, UnitZ) * v
  double s = sin(ang), c = cos(ang);
  return V3{c * v.x - s * v.y, s * v.x + c * v.y, v.z};
}
SHC_HDI V3 set_precision3(V3 v) { // setPrecision(vector, 3) (standard_includes.h:152)
  return V3{round_to_int(v.x * pow(10, 3)) / pow(10, 3), round_to_int(v.y * pow(10, 3)) / pow(10, 3),
            round_to_int(v.z * pow(10, 3)) / pow(10, 3)};
}

// WalkController::generateWalkspace with default tip == identity tip for every leg (true right after start-up)
SHC_HDI void generate_walkspace(const shc_params &p, const double (*workspace)[SHC_N_BEARINGS], double (&walkspace)[SHC_N_BEARINGS]) {
  const int L = p.leg_count;
  bool have[SHC_N_BEARINGS] = {false};
  for (int l = 0; l < L; ++l) {
    int l1 = mod_i(l + 1, L), l2 = mod_i(l - 1, L);
    V3 d{p.stance_position[l][0], p.stance_position[l][1], 0.0};
    V3 a1{p.stance_position[l1][0], p.stance_position[l1][1], 0.0}, a2{p.stance_position[l2][0], p.stance_position[l2][1], 0.0};
    double dist1 = norm(d - a1) / 2.0, dist2 = norm(d - a2) / 2.0;
    double b1 = rad2deg(atan2(a1.y - d.y, a1.x - d.x)), b2 = rad2deg(atan2(a2.y - d.y, a2.x - d.x));
    for (int bearing = 0; bearing <= 360; bearing += kBearingStep) {
      int diff1 = iabs(mod_i(int(b1), 360) - bearing), diff2 = iabs(mod_i(int(b2), 360) - bearing);
      double o1 = kUnassigned, o2 = kUnassigned;
      if ((diff1 < 90 || diff1 > 270) && dist1 > 0.0) o1 = dist1 / cos(deg2rad(diff1));
      if ((diff2 < 90 || diff2 > 270) && dist2 > 0.0) o2 = dist2 / cos(deg2rad(diff2));
      double md = p.overlapping_walkspaces ? kMaxWorkspaceRadius : dmin(o1, o2);
      md = dmin(md, kMaxWorkspaceRadius);
      int bi = bearing / kBearingStep;
      if (!have[bi]) {
        walkspace[bi] = md;
        have[bi] = true;
      } else if (md < walkspace[bi]) {
        walkspace[bi] = md;
      }
    }
  }
  for (int l = 0; l < L; ++l) {
    for (int bi = 0; bi < SHC_N_BEARINGS; ++bi) {
      double radius = workspace[l][bi]; // default shift is zero: radius = workplane.at(bearing) (walk_controller.cpp:137-140)
      int opposite = mod_i(bi * kBearingStep + 180, 360) / kBearingStep;
      if (radius < walkspace[bi]) {
        walkspace[bi] = radius;
        walkspace[opposite] = radius;
      }
    }
  }
  walkspace[360 / kBearingStep] = walkspace[0];
}

SHC_HDI void generate_limits(const shc_params &p, shc_tables &t) {
  const shc_step_cycle &step = t.step;
  const int L = p.leg_count;
  int base = p.stance_phase + p.swing_phase;
  int normaliser = step.period / base;
  int base_offset = int(p.phase_offset * normaliser);
  int max_ext = 0;
  for (int l = 0; l < L; ++l) {
    int off = (base_offset * p.offset_multiplier[l]) % step.period;
    t.phase_offset[l] = off;
    if (off > step.swing_start && off < step.swing_end) max_ext = imax(max_ext, step.swing_end - off);
  }
  double time_to_max_stride = (max_ext + step.stance_period + step.swing_period) * p.time_delta;
  for (int b = 0; b < SHC_N_BEARINGS; ++b) {
    double wr = t.walkspace[b];
    double ogr = double(step.stance_period) / step.period;
    double max_speed = (wr * 2.0) / (ogr / step.frequency);
    double max_acc = max_speed / time_to_max_stride;
    double overshoot = 0;
    for (int l = 0; l < L; ++l) {
      double off = t.phase_offset[l];
      double tt = off * p.time_delta;
      double tse = time_to_max_stride - tt;
      double v0 = max_acc * tse;
      double stride_length = v0 * (ogr / step.frequency);
      double d0 = -stride_length / 2.0;
      double d1 = d0 + v0 * tt + 0.5 * max_acc * (tt * tt);
      double d2 = max_speed * (step.stance_period * p.time_delta - tt);
      overshoot = dmax(overshoot, d1 + d2 - wr);
    }
    double swing_overshoot = 0.5 * max_speed * step.swing_period / (2.0 * step.period * step.frequency);
    double scaled = (wr / (wr + overshoot + swing_overshoot)) * wr;
    double sx = p.stance_position[0][0], sy = p.stance_position[0][1];
    double stance_radius = sqrt(sx * sx + sy * sy);
    double mls = (scaled * 2.0) / (ogr / step.frequency);
    double mla = mls / time_to_max_stride;
    double mas = mls / stance_radius;
    double maa = mas / time_to_max_stride;
    if (wr == 0.0) {
      mls = 0.0;
      mla = kUnassigned;
      mas = 0.0;
      maa = kUnassigned;
    }
    t.max_linear_speed[b] = mls;
    t.max_linear_acceleration[b] = mla;
    t.max_angular_speed[b] = mas;
    t.max_angular_acceleration[b] = maa;
  }
}

// Step cycle + auto-pose phase tables of one morphology (everything that needs no IK).  False for a degenerate gait.
SHC_HDI bool generate_tables_head(const shc_params &p, shc_tables &t) {
  __builtin_memset(&t, 0, sizeof t);
  t.step = generate_step_cycle(p);
  if (t.step.period <= 0 || t.step.stance_period <= 0 || t.step.swing_period <= 0) return false;
  // auto-pose phase length / normaliser (pose_controller.cpp:44-63) and reference leg (:75-78)
  int base;
  double raw;
  if (p.pose_frequency == -1.0) {
    base = p.stance_phase + p.swing_phase;
    double swing_ratio = double(p.swing_phase) / base;
    raw = ((1.0 / p.step_frequency) / p.time_delta) / swing_ratio;
  } else {
    base = p.pose_phase_length;
    raw = ((1.0 / p.pose_frequency) / p.time_delta);
  }
  if (base <= 0) base = 1;
  t.pose_phase_length = round_to_even_int(raw / base) * base;
  t.pose_normaliser = t.pose_phase_length / base;
  t.auto_pose_reference_leg = 0;
  for (int l = 0; l < p.leg_count; ++l)
    if (p.offset_multiplier[l] == 0) t.auto_pose_reference_leg = l;
  return true;
}

// Model::current_pose_ as PoseController::updateCurrentPose leaves it in loop `call` (0-based) of the direct start-up.
// The loops before RUNNING already pose the body (state_controller.cpp:165-167 runs for every robot state): walk-plane pose
// (0, 0, body_clearance) + identity manual pose (no inputs yet) + - because the IMU branch needs RUNNING, pose_controller.cpp:836 -
// the auto pose whenever auto posing runs on its own clock (pose_frequency != -1: every AutoPoser is allowed from the first
// call on, :1359-1371, and the master phase is the call counter, :1150-1159).  Synchronised with the step cycle the posers
// never start while the walk state is STOPPED, so the auto pose is the identity.  The simulated start-up solve targets the
// pose of loop 0 (directStartup's first call, :483-489); the workspace search runs at the pose of the loop that reaches
// READY (state_controller.cpp:263-272; Leg::generateWorkspace reads model_->getCurrentPose(), model.cpp:338).
SHC_HDI Pose startup_body_pose(const shc_params &p, const shc_tables &t, int call) {
  Pose cp{V3{0, 0, p.body_clearance}, quat_identity()};
  if (!p.auto_posing || p.pose_frequency == -1.0 || t.pose_phase_length <= 0) return cp;
  const int len = t.pose_phase_length, nrm = t.pose_normaliser;
  const int master_phase = call % len;
  Pose auto_pose = pose_identity();
  for (int i = 0; i < p.n_auto_posers && i < SHC_MAX_AUTO_POSERS; ++i) { // AutoPoser::updatePose (:1338-1439)
    int phase = master_phase, sp = p.pose_phase_starts[i] * nrm, ep = p.pose_phase_ends[i] * nrm;
    if (sp > ep) {
      ep += len;
      if (phase < sp) phase += len;
    }
    if (phase >= sp && phase < ep) {
      const int iteration = phase - sp + 1, num = ep - sp;
      const bool first_half = iteration <= num / 2;
      const double delta_t = 1.0 / (num / 2.0);
      const int offset = int(first_half ? 0 : num / 2.0);
      const double tt = (iteration - offset) * delta_t, u = 1.0 - tt;
      // control nodes {0,0,0,A,A} / {A,A,0,0,0}: B(t) = A * weight
      const double wgt = first_half ? (4.0 * tt * tt * tt * u + tt * tt * tt * tt) : (u * u * u * u + 4.0 * tt * u * u * u);
      V3 pos;
      if (p.gravity_amplitudes[i] != 0.0) pos = V3{0, 0, -1.0} * (p.gravity_amplitudes[i] * wgt); // estimateGravity with the IMU still at identity
      else pos = V3{p.x_amplitudes[i] * wgt, p.y_amplitudes[i] * wgt, p.z_amplitudes[i] * wgt};
      const V3 rot{p.roll_amplitudes[i] * wgt, p.pitch_amplitudes[i] * wgt, p.yaw_amplitudes[i] * wgt};
      auto_pose = add_pose(auto_pose, Pose{pos, euler_to_quat(rot, false)});
    }
  }
  return add_pose(cp, auto_pose);
}
SHC_HDI int startup_loops(const shc_params &p) { return imax(1, round_to_int(p.time_to_start / p.time_delta)); } // :1520

// Direct start-up solve + workspace search of ONE leg: the sequential part (thousands of DLS steps), independent per leg.
// (the device batch splits the eight bearing searches of a leg over eight threads: each repeats the start-up solve and
//  the re-basing prefix, ~350 steps, and searches one bearing, <= 500 steps)
template <int NJ>
SHC_HDI void generate_tables_leg(const shc_params &p, int l, shc_tables &t, int first_bearing = 1, int last_bearing = 8,
                                 const double *preset_configuration = nullptr, LayeredPlanes *keep_planes = nullptr, const Pose *workspace_pose = nullptr) {
  // body pose the start-up solve eases to / the workspace search runs at (identical unless auto posing has its own clock); workspace_pose: Model::current_pose_
  // of the loop that completed a start-up SEQUENCE (Leg::generateWorkspace reads it, model.cpp:338)
  const Pose body = startup_body_pose(p, t, 0), body_ws = workspace_pose ? *workspace_pose : startup_body_pose(p, t, startup_loops(p) - 1);
  HostLeg<NJ> leg;
  fill_leg_const<NJ>(p, l, leg.lc);
  for (int j = 0; j < NJ; ++j) leg.dflt[j] = clampd(0.0, leg.lc.jmin[j], leg.lc.jmax[j]); // model.cpp:1038
  V3 default_tip{p.stance_position[l][0], p.stance_position[l][1], 0.0};
  if (preset_configuration) { // the default configuration is given (the joints a start-up SEQUENCE ended on, state_controller.cpp:307-310)
    for (int j = 0; j < NJ; ++j) leg.q[j] = preset_configuration[j], leg.qd[j] = 0.0;
    leg.fk();
  } else {
    startup_solve<NJ>(p, leg, default_tip, body);
  }
  for (int j = 0; j < NJ; ++j) {
    leg.dflt[j] = leg.q[j]; // Model::updateDefaultConfiguration
    if (first_bearing == 1) t.default_joint_position[l][j] = leg.q[j];
  }
  if (p.rough_terrain_mode) { // layered workspace: planes are searched one after the other, no split over bearings
    if (first_bearing == 1) generate_workspace_layered<NJ>(p, leg, inverse_transform_vector(body_ws, default_tip), t.workspace_radius[l], keep_planes);
    return;
  }
  generate_workspace<NJ>(p, leg, inverse_transform_vector(body_ws, default_tip), t.workspace_radius[l], first_bearing, last_bearing);
}

// Walkspace + velocity / acceleration limits from the legs' workspaces.
SHC_HDI void generate_tables_tail(const shc_params &p, shc_tables &t) {
  generate_walkspace(p, t.workspace_radius, t.walkspace);
  generate_limits(p, t);
}

template <int NJ>
SHC_HDI bool generate_tables(const shc_params &p, shc_tables &t, const double *preset_configuration /* [legs][NJ] or nullptr */ = nullptr,
                             const Pose *workspace_pose = nullptr) {
  if (!generate_tables_head(p, t)) return false;
  for (int l = 0; l < p.leg_count; ++l) generate_tables_leg<NJ>(p, l, t, 1, 8, preset_configuration ? preset_configuration + l * NJ : nullptr, nullptr, workspace_pose);
  generate_tables_tail(p, t);
  return true;
}

} // namespace hostinit
} // namespace shc

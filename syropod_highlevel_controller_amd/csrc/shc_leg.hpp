// shc_leg.hpp — per-leg kinematics: DH forward kinematics, geometric Jacobian, one damped-least-squares IK
// step, joint integration / clamping, tip-force estimate.  Host + device (the HIP cycle kernel and the host
// init chain share this code; the CPU oracle does not).
//
// Reference behaviour restated (src/model.cpp of OpenSHC v0.5.11):
//   Leg::applyFK :945-988, Leg::solveIK :726-795, Leg::updateJointPositions :799-857, Leg::applyIK :861-941,
//   Leg::calculateTipForce :667-708; frames per model.h:594-617 (Jacobian and position delta live in the
//   joint-1 frame; joint 1's transform is the constant base-link DH matrix).
//
// MI355X-first formulation (mathematically identical to the reference's, different arithmetic):
//   reference:  dq = J^T (J J^T + l^2 I6)^-1 d  +  (I - J^+ J) g      with a dynamic 6x6 LU inverse
//   here:       dq = (Jp^T Jp + l^2 I_N)^-1 (Jp^T d_p + l^2 g)         one N x N SPD solve in registers
//   (push-through identity; in position-only mode the angular rows of J are zero, so J^T J = Jp^T Jp,
//    and I - J^+ J = l^2 (J^T J + l^2 I)^-1.)  Differences are at rounding level (~1e-13 rad).
#pragma once

#include "shc_math.hpp"

namespace shc {

// Per-leg constant record.  Plain doubles/ints so it can be staged verbatim in LDS.
template <int NJ>
struct LegConst {
  double r1[9];       // rotation of the base transform T1 = DH(base link)  (row-major)
  double p1[3];       // translation of T1
  double link_d[NJ], link_r[NJ], link_sa[NJ], link_ca[NJ], link_th[NJ]; // links 1..NJ (link k actuated by joint k)
  double jmin[NJ], jmax[NJ], jvmax[NJ];
  double jcentre[NJ];  // min + (max - min) / 2                       (model.cpp:771)
  double jw_range[NJ]; // JOINT_LIMIT_COST_WEIGHT / (max - min), 0 if the range is 0   (model.cpp:772-778)
  double jw_vrange[NJ]; // JOINT_LIMIT_COST_WEIGHT / (2 max_vel)      (model.cpp:781-786)
  double jactive[NJ];   // 1: a joint of this leg; 0: padding of a leg with fewer joints than the robot's longest (Parameters::leg_DOF is per
                        // leg): a zero-length link behind the tip with the range [0, 0] - its linear Jacobian column is zero by geometry,
                        // the angular one (tip-force estimate, rotation-constrained solves) is masked with this
  double stance_x, stance_y; // identity tip position (walk_controller.cpp:34-35)
  double span_shift;         // LegStepper::calculateStanceSpanChange().y for the single-plane workspace (walk_controller.cpp:949-980)
  double neg_ratio;          // negation_transition_ratio
  int32_t phase_offset;      // walk_controller.cpp:277
  int32_t neg_start, neg_end; // pose negation phases (already * normaliser, 0 -> phase length; pose_controller.cpp:1718-1731)
  int32_t first_stance_period;     // modified stance period of the first step (walk_controller.cpp:1026-1031)
  int32_t first_stance_iterations; // int((msp / period) / (frequency * dt))   (walk_controller.cpp:1040)
  int32_t pad_;
  double first_stance_dt;          // 1 / first_stance_iterations              (walk_controller.cpp:1041)
  double first_stride_scaler;      // first_stance_period / stance_period      (walk_controller.cpp:1167)
  int32_t starts_in_swing;         // phase_offset strictly inside the swing window (walk_controller.cpp:587-588)
};
// (int32 members are kept in pairs so the record stays a whole number of 8-byte words for the LDS copy)

template <int NJ>
struct Chain {
  V3 z[NJ]; // joint axes in the joint-1 frame (z[0] = (0,0,1))
  V3 p[NJ]; // joint origins in the joint-1 frame (p[0] = 0)
  V3 pe;    // tip position in the joint-1 frame
  V3 xe;    // tip x-axis in the joint-1 frame
};

// sin / cos of the DH joint angles theta_k + q_k: the only transcendental part of Leg::applyFK.
template <int NJ, class LC>
SHC_HD void joint_sincos(const LC &lc, const double (&q)[NJ], double (&sn)[NJ], double (&cs)[NJ]) {
#pragma unroll
  for (int k = 0; k < NJ; ++k) sincos_joint(lc.link_th[k] + q[k], &sn[k], &cs[k]);
}

// Leg::applyFK chain product in the joint-1 frame from the joint sines / cosines.
// The chain and the Jacobian columns are evaluated both in the kernel prologue (state loaded from HBM) and inside the
// cycle loop (state carried in registers); a launch of n cycles must reproduce n single-cycle launches bit for bit, so
// these two functions fix their own rounding: contraction is off and every fused multiply-add is written out.
template <int NJ, class LC>
SHC_HD void chain_from_sincos(const LC &lc, const double (&sn)[NJ], const double (&cs)[NJ], Chain<NJ> &c) {
#pragma clang fp contract(off)
  double X[3] = {1, 0, 0}, Y[3] = {0, 1, 0}, Z[3] = {0, 0, 1}, P[3] = {0, 0, 0};
  c.z[0] = V3{0, 0, 1};
  c.p[0] = V3{0, 0, 0};
#pragma unroll
  for (int k = 0; k < NJ; ++k) {
    const double s = sn[k], co = cs[k];
    const double sa = lc.link_sa[k], ca = lc.link_ca[k], r = lc.link_r[k], d = lc.link_d[k];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double xn = fma(X[a], co, Y[a] * s);   // X cos + Y sin
      const double t = fma(Y[a], co, -(X[a] * s)); // Y cos - X sin
      const double yn = fma(t, ca, Z[a] * sa);
      const double zn = fma(Z[a], ca, -(t * sa));
      P[a] = fma(Z[a], d, fma(xn, r, P[a]));       // P + Xn r + Z d (Z of the previous frame)
      X[a] = xn;
      Y[a] = yn;
      Z[a] = zn;
    }
    if (k + 1 < NJ) {
      c.z[k + 1] = V3{Z[0], Z[1], Z[2]};
      c.p[k + 1] = V3{P[0], P[1], P[2]};
    }
  }
  c.pe = V3{P[0], P[1], P[2]};
  c.xe = V3{X[0], X[1], X[2]};
}

template <int NJ, class LC>
SHC_HD void fk_chain(const LC &lc, const double (&q)[NJ], Chain<NJ> &c) {
  double sn[NJ], cs[NJ];
  joint_sincos<NJ>(lc, q, sn, cs);
  chain_from_sincos<NJ>(lc, sn, cs, c);
}

template <class LC>
SHC_HD V3 base_rotate(const LC &lc, V3 v) { // R1 * v
  return V3{lc.r1[0] * v.x + lc.r1[1] * v.y + lc.r1[2] * v.z, lc.r1[3] * v.x + lc.r1[4] * v.y + lc.r1[5] * v.z,
            lc.r1[6] * v.x + lc.r1[7] * v.y + lc.r1[8] * v.z};
}
template <class LC>
SHC_HD V3 base_rotate_inv(const LC &lc, V3 v) { // R1^T * v
  return V3{lc.r1[0] * v.x + lc.r1[3] * v.y + lc.r1[6] * v.z, lc.r1[1] * v.x + lc.r1[4] * v.y + lc.r1[7] * v.z,
            lc.r1[2] * v.x + lc.r1[5] * v.y + lc.r1[8] * v.z};
}
template <class LC>
SHC_HD V3 tip_robot_frame(const LC &lc, V3 pe) { // T1 * pe
  return base_rotate(lc, pe) + V3{lc.p1[0], lc.p1[1], lc.p1[2]};
}

// One DLS step towards `desired` (robot frame): Leg::applyIK :864-877 + solveIK with solve_rotation = false.
// linear Jacobian columns z_i x (p_e - p_i) in the joint-1 frame (model.cpp:735-747)
template <int NJ>
SHC_HD void jacobian_columns(const Chain<NJ> &c, V3 (&lin)[NJ]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const double dx = c.pe.x - c.p[i].x, dy = c.pe.y - c.p[i].y, dz = c.pe.z - c.p[i].z;
    const V3 z = c.z[i];
    lin[i] = V3{fma(z.y, dz, -(z.z * dy)), fma(z.z, dx, -(z.x * dz)), fma(z.x, dy, -(z.y * dx))};
  }
}

template <int NJ, bool EXACT = false, class LC>
SHC_HD void ik_step_cols(const LC &lc, const V3 (&lin)[NJ], V3 pe, const double (&q)[NJ], const double (&qd)[NJ], V3 desired,
                         double (&dq)[NJ]);

template <int NJ, bool EXACT = false, class LC>
SHC_HD void ik_step(const LC &lc, const Chain<NJ> &c, const double (&q)[NJ], const double (&qd)[NJ], V3 desired,
                    double (&dq)[NJ]) {
  V3 lin[NJ];
  jacobian_columns<NJ>(c, lin);
  ik_step_cols<NJ, EXACT>(lc, lin, c.pe, q, qd, desired, dq);
}

template <int NJ, bool EXACT, class LC>
SHC_HD void ik_step_cols(const LC &lc, const V3 (&lin)[NJ], V3 pe, const double (&q)[NJ], const double (&qd)[NJ], V3 desired,
                         double (&dq)[NJ]) {
  // position delta in the joint-1 frame: T1^-1 desired - T1^-1 current
  V3 delta = base_rotate_inv(lc, desired - V3{lc.p1[0], lc.p1[1], lc.p1[2]}) - pe;
  // joint-limit cost gradient (model.cpp:759-790)
  double pg[NJ], vg[NJ];
  double pcost = 0.0, vcost = 0.0; // (the rotation solve below repeats this block on the intermediate joint state)
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    double e = (q[i] - lc.jcentre[i]) * lc.jw_range[i];
    pcost += e * e;
    pg[i] = -e * lc.jw_range[i];
    double v = qd[i] * lc.jw_vrange[i];
    vcost += v * v;
    vg[i] = -v * lc.jw_vrange[i];
  }
  // evaluated unconditionally and selected afterwards: no exec-mask branch around the sqrt / division
  double ps = fast_rsqrt<EXACT>(pcost), vs = fast_rsqrt<EXACT>(vcost);
  ps = pcost == 0.0 ? 0.0 : ps;
  vs = vcost == 0.0 ? 0.0 : vs;
  const double l2 = kDls * kDls;
  double a[NJ][NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) a[i][j] = dot(lin[i], lin[j]);
    a[i][i] += l2;
    double g = 0.25 * (pg[i] * ps) + 0.75 * (vg[i] * vs);
    dq[i] = dot(lin[i], delta) + l2 * g;
  }
  spd_solve<NJ, EXACT>(a, dq);
}

// The second solve of a rotation-constrained Leg::applyIK (model.cpp:880-900): delta has only its angular rows set
// (axis * angle from the current to the desired tip direction, joint-1 frame) and the Jacobian its angular rows too
// (joint axes).  Same push-through form as ik_step: dq = (J^T J + l^2 I)^-1 (J_w^T d_w + l^2 g), J^T J = lin.lin + z.z.
template <int NJ, bool EXACT = false, class LC>
SHC_HD void ik_step_rotation(const LC &lc, const Chain<NJ> &c, const V3 (&lin)[NJ], const double (&q)[NJ], const double (&qd)[NJ],
                             V3 rot_delta, double (&dq)[NJ]) {
  double pg[NJ], vg[NJ];
  double pcost = 0.0, vcost = 0.0;
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    double e = (q[i] - lc.jcentre[i]) * lc.jw_range[i];
    pcost += e * e;
    pg[i] = -e * lc.jw_range[i];
    double v = qd[i] * lc.jw_vrange[i];
    vcost += v * v;
    vg[i] = -v * lc.jw_vrange[i];
  }
  double ps = fast_rsqrt<EXACT>(pcost), vs = fast_rsqrt<EXACT>(vcost);
  ps = pcost == 0.0 ? 0.0 : ps;
  vs = vcost == 0.0 ? 0.0 : vs;
  const double l2 = kDls * kDls;
  double a[NJ][NJ];
  V3 z[NJ]; // joint axes = the angular Jacobian columns (padding joints have none)
#pragma unroll
  for (int i = 0; i < NJ; ++i) z[i] = c.z[i] * lc.jactive[i];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) a[i][j] = dot(lin[i], lin[j]) + dot(z[i], z[j]);
    a[i][i] += l2;
    double g = 0.25 * (pg[i] * ps) + 0.75 * (vg[i] * vs);
    dq[i] = dot(z[i], rot_delta) + l2 * g;
  }
  spd_solve<NJ, EXACT>(a, dq);
}

// Leg::solveIK for an arbitrary 6-vector delta (model.cpp:726-795), the general form behind ik_step_cols / ik_step_rotation:
// with solve_rotation the Jacobian keeps its angular rows (joint axes), without it they are zero and the angular part of
// delta has no effect.  Push-through form dq = (J^T J + l^2 I)^-1 (J^T delta + l^2 g).
template <int NJ, bool EXACT = false, class LC>
SHC_HD void solve_ik_delta(const LC &lc, const Chain<NJ> &c, const V3 (&lin)[NJ], const double (&q)[NJ], const double (&qd)[NJ], V3 dp, V3 dw,
                           bool solve_rotation, double (&dq)[NJ]) {
  double pg[NJ], vg[NJ];
  double pcost = 0.0, vcost = 0.0;
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    double e = (q[i] - lc.jcentre[i]) * lc.jw_range[i];
    pcost += e * e;
    pg[i] = -e * lc.jw_range[i];
    double v = qd[i] * lc.jw_vrange[i];
    vcost += v * v;
    vg[i] = -v * lc.jw_vrange[i];
  }
  double ps = fast_rsqrt<EXACT>(pcost), vs = fast_rsqrt<EXACT>(vcost);
  ps = pcost == 0.0 ? 0.0 : ps;
  vs = vcost == 0.0 ? 0.0 : vs;
  const double l2 = kDls * kDls;
  double a[NJ][NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) a[i][j] = dot(lin[i], lin[j]) + (solve_rotation ? dot(c.z[i] * lc.jactive[i], c.z[j] * lc.jactive[j]) : 0.0);
    a[i][i] += l2;
    double g = 0.25 * (pg[i] * ps) + 0.75 * (vg[i] * vs);
    dq[i] = dot(lin[i], dp) + (solve_rotation ? dot(c.z[i] * lc.jactive[i], dw) : 0.0) + l2 * g;
  }
  spd_solve<NJ, EXACT>(a, dq);
}

// Tip::getPoseRobotFrame (model.h:684): full tip pose (robot frame) of the chain at joint angles q; the rotation is the
// quaternion of R1 * R_chain as Pose::transform builds it (pose.h:144).
template <int NJ, class LC>
SHC_HD Pose fk_tip_pose(const LC &lc, const double (&q)[NJ]) {
  double X[3] = {1, 0, 0}, Y[3] = {0, 1, 0}, Z[3] = {0, 0, 1}, P[3] = {0, 0, 0};
  for (int k = 0; k < NJ; ++k) {
    double s, co;
    sincos_joint(lc.link_th[k] + q[k], &s, &co);
    const double sa = lc.link_sa[k], ca = lc.link_ca[k], r = lc.link_r[k], d = lc.link_d[k];
    for (int a = 0; a < 3; ++a) {
      const double xn = X[a] * co + Y[a] * s, t = Y[a] * co - X[a] * s;
      const double yn = t * ca + Z[a] * sa, zn = Z[a] * ca - t * sa;
      P[a] = P[a] + xn * r + Z[a] * d;
      X[a] = xn, Y[a] = yn, Z[a] = zn;
    }
  }
  double m[9]; // R1 * [X Y Z]
  for (int i = 0; i < 3; ++i) {
    m[i * 3 + 0] = lc.r1[i * 3] * X[0] + lc.r1[i * 3 + 1] * X[1] + lc.r1[i * 3 + 2] * X[2];
    m[i * 3 + 1] = lc.r1[i * 3] * Y[0] + lc.r1[i * 3 + 1] * Y[1] + lc.r1[i * 3 + 2] * Y[2];
    m[i * 3 + 2] = lc.r1[i * 3] * Z[0] + lc.r1[i * 3 + 1] * Z[1] + lc.r1[i * 3 + 2] * Z[2];
  }
  return Pose{tip_robot_frame(lc, V3{P[0], P[1], P[2]}), normalized(quat_from_matrix(m))};
}

// axis * angle of the shortest rotation taking the current tip direction to the desired one (model.cpp:884-893)
SHC_HD V3 tip_rotation_delta(V3 current_dir, V3 desired_dir) { return angle_axis_vector(normalized(from_two_vectors(current_dir, desired_dir))); }

// Leg::updateJointPositions (model.cpp:799-857).  Returns the minimum limit proximity.
template <int NJ, class LC>
SHC_HD double update_joints(const LC &lc, const double (&dq)[NJ], double dt, double inv_dt, bool clamp_vel, bool clamp_pos,
                            double (&q)[NJ], double (&qd)[NJ]) {
  double prox = 1.0;
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    double v = dq[i] * inv_dt; // delta / time_delta (model.cpp:808) with the reciprocal precomputed on the host
    // clamps as min / max against +-infinity when a clamp is disabled: branch-free, same values (model.cpp:812-841)
    const double vm = clamp_vel ? lc.jvmax[i] : HUGE_VAL;
    v = fmin(fmax(v, -vm), vm);
    double nq = q[i] + v * dt;
    const double lo = clamp_pos ? lc.jmin[i] : -HUGE_VAL, hi = clamp_pos ? lc.jmax[i] : HUGE_VAL;
    nq = fmin(fmax(nq, lo), hi);
    qd[i] = v;
    q[i] = nq;
    double half = (lc.jmax[i] - lc.jmin[i]) / 2.0;
    double lp = half != 0 ? fmin(fabs(lc.jmin[i] - nq), fabs(lc.jmax[i] - nq)) / half : 1.0;
    prox = fmin(lp, prox);
  }
  return prox;
}

// Leg::calculateTipForce (model.cpp:667-708): raw force in the robot frame (before the low-pass filter).
template <int NJ, class LC>
SHC_HD V3 tip_force_cols(const LC &lc, const Chain<NJ> &c, const V3 (&lin)[NJ], const double (&tau)[NJ]);

template <int NJ, class LC>
SHC_HD V3 tip_force_raw(const LC &lc, const Chain<NJ> &c, const double (&tau)[NJ]) {
  V3 lin[NJ];
  jacobian_columns<NJ>(c, lin);
  return tip_force_cols<NJ>(lc, c, lin, tau);
}

template <int NJ, class LC>
SHC_HD V3 tip_force_cols(const LC &lc, const Chain<NJ> &c, const V3 (&lin)[NJ], const double (&tau)[NJ]) {
  double a[NJ][NJ], y[NJ];
  V3 z[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) z[i] = c.z[i] * lc.jactive[i];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) a[i][j] = dot(lin[i], lin[j]) + dot(z[i], z[j]);
    a[i][i] += kDls * kDls;
    y[i] = tau[i];
  }
  spd_solve<NJ>(a, y);
  V3 f{0, 0, 0};
#pragma unroll
  for (int i = 0; i < NJ; ++i) f = f + lin[i] * y[i];
  return base_rotate_inv(lc, f);
}

} // namespace shc

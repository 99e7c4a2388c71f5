// shc_leg_api.hpp — the reference's per-leg Leg methods (model.h:448-492), batched over (instance, leg):
//   Leg::setDesiredTipPose :448 (model.cpp:653)    Leg::solveIK :470 (model.cpp:726)    Leg::updateJointPositions :477 (model.cpp:799)
//   Leg::applyIK :485 (model.cpp:861)              Leg::applyFK :492 (model.cpp:945)
// The reference's cold paths call these on their own (workspace search model.cpp:309-510, sequences pose_controller.cpp:
// 145-805, leg manipulation state_controller.cpp:590-700); inside the fused cycle kernel the same shc_leg.hpp functions run
// in registers.  One thread per (instance, leg); state is read from / written to the engine's SoA planes, so a cycle launched
// afterwards continues from the joints these calls left.
#pragma once

#include "shc_cycle.hpp"

namespace shc {

struct LegSel {
  int64_t first, count; // instances [first, first + count)
  int leg;              // -1: every leg, else this leg only
  int L;
  __device__ __forceinline__ int legs() const { return leg < 0 ? L : 1; }
  // thread t -> (instance, leg); returns false past the end
  __device__ __forceinline__ bool map(int64_t t, int64_t &rob, int &l) const {
    const int nl = legs();
    if (t >= count * nl) return false;
    rob = first + t / nl;
    l = leg < 0 ? int(t % nl) : leg;
    return true;
  }
};

template <int NJ>
struct LegIO {
  const DevState &st;
  int64_t slot;
  __device__ __forceinline__ double get(int f) const { return st.legd[leg_field_index(f, slot, st.n_slots)]; }
  __device__ __forceinline__ void put(int f, double v) const { st.legd[leg_field_index(f, slot, st.n_slots)] = v; }
  __device__ __forceinline__ V3 get3(int f) const { return V3{get(f), get(f + 1), get(f + 2)}; }
  __device__ __forceinline__ void put3(int f, V3 v) const {
    put(f, v.x);
    put(f + 1, v.y);
    put(f + 2, v.z);
  }
  __device__ __forceinline__ void joints(double (&q)[NJ], double (&qd)[NJ]) const {
    for (int j = 0; j < NJ; ++j) {
      q[j] = get(Fields<NJ>::Q + j);
      qd[j] = get(Fields<NJ>::QD + j);
    }
  }
  __device__ __forceinline__ void put_joints(const double (&q)[NJ], const double (&qd)[NJ]) const {
    for (int j = 0; j < NJ; ++j) {
      put(Fields<NJ>::Q + j, q[j]);
      put(Fields<NJ>::QD + j, qd[j]);
    }
  }
};

__device__ __forceinline__ int leg_state_of(const DevState &st, int64_t rob, int leg) { // enum LegState; WALKING until a leg has ever been toggled
  return st.manual ? st.manual[rob].leg_state[leg] : LS_WALKING;
}
// Leg::setDesiredTipPose(tip_pose, apply_delta): tip_pose == NULL is the reference's default argument Pose::Undefined(),
// "use the poser's tip pose" (model.cpp:657-660; POSER_TIP must have been derived, see derive_tips_kernel).
template <int NJ>
__device__ __forceinline__ void set_desired_dev(const DevState &st, const LegIO<NJ> &io, int L, int64_t rob, const double *pose7, int apply_delta,
                                                int have_adm, int gravity_aligned) {
  using FD = Fields<NJ>;
  using R = RobotFields;
  V3 pos, dir{0, 0, 0};
  bool defined = false;
  if (pose7) {
    const double *p = pose7;
    pos = V3{p[0], p[1], p[2]};
    const Quat r{p[3], p[4], p[5], p[6]};
    defined = !(r.w == 0.0 && r.x == 0.0 && r.y == 0.0 && r.z == 0.0); // != UNDEFINED_ROTATION (isApprox with the zero quaternion)
    if (defined) dir = rotate(r, V3{1, 0, 0});
  } else {
    pos = io.get3(FD::POSER_TIP);
    if (gravity_aligned && (st.legi[io.slot] & LW_ROTDEF)) { // pose.rotation^-1 * walker tip rotation (pose_controller.cpp:129-130)
      const int rpw = 64 / L;
      const Quat cr{st.robd[rob_index(rob, R::CPOSE + 3, rpw, R::COUNT)], st.robd[rob_index(rob, R::CPOSE + 4, rpw, R::COUNT)],
                    st.robd[rob_index(rob, R::CPOSE + 5, rpw, R::COUNT)], st.robd[rob_index(rob, R::CPOSE + 6, rpw, R::COUNT)]};
      const int ls = leg_state_of(st, rob, int(io.slot & 63) % L);
      dir = io.get3(FD::CUR_DIR); // (manually manipulated legs: the stepper's tip pose as it is, :135-139)
      if (ls != LS_MANUAL && ls != LS_WALKING_TO_MANUAL) dir = rotate(inverse(cr), dir);
      defined = true;
    }
  }
  const int leg_state = leg_state_of(st, rob, int(io.slot & 63) % L);
  // model.cpp:655-661: the admittance delta is not applied to manually manipulated legs
  if (apply_delta && have_adm && leg_state != LS_MANUAL && leg_state != LS_WALKING_TO_MANUAL) pos = pos + io.get3(FD::ADM_DELTA);
  io.put3(FD::DES_TIP, pos);
  io.put(FD::DES_TIP + 3, defined ? 1.0 : 0.0);
  io.put3(FD::DES_DIR, dir);
}
template <int L_, int NJ>
__global__ void leg_set_desired_kernel(DevState st, const SharedConsts<L_, NJ> *gc, LegSel sel, const double *tip_pose, int apply_delta,
                                       int have_adm, int gravity_aligned) {
  int64_t rob;
  int l;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (!sel.map(t, rob, l)) return;
  const LegIO<NJ> io{st, slot_of(rob, l, sel.L)};
  set_desired_dev<NJ>(st, io, sel.L, rob, tip_pose ? tip_pose + t * 7 : nullptr, apply_delta, have_adm, gravity_aligned);
}

template <int L_, int NJ>
__global__ void leg_solve_ik_kernel(DevState st, const SharedConsts<L_, NJ> *gc, LegSel sel, const double *delta, int solve_rotation, double *out) {
  int64_t rob;
  int l;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (!sel.map(t, rob, l)) return;
  const LegIO<NJ> io{st, slot_of(rob, l, sel.L)};
  const LegConst<NJ> &lc = gc->leg[l];
  double q[NJ], qd[NJ], dq[NJ];
  io.joints(q, qd);
  Chain<NJ> ch;
  fk_chain<NJ>(lc, q, ch); // the joint transforms the last applyFK left (model.cpp:731, :744)
  V3 lin[NJ];
  jacobian_columns<NJ>(ch, lin);
  const double *d = delta + t * 6;
  solve_ik_delta<NJ>(lc, ch, lin, q, qd, V3{d[0], d[1], d[2]}, V3{d[3], d[4], d[5]}, solve_rotation != 0, dq);
  for (int j = 0; j < NJ; ++j) out[t * NJ + j] = dq[j];
}

template <int L_, int NJ>
__global__ void leg_update_joints_kernel(DevState st, const SharedConsts<L_, NJ> *gc, LegSel sel, const double *delta, int simulation, double *prox_out,
                                         double dt, int clamp_vel, int clamp_pos) {
  int64_t rob;
  int l;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (!sel.map(t, rob, l)) return;
  const LegIO<NJ> io{st, slot_of(rob, l, sel.L)};
  const LegConst<NJ> &lc = gc->leg[l];
  double q[NJ], qd[NJ], dq[NJ];
  io.joints(q, qd);
  for (int j = 0; j < NJ; ++j) dq[j] = delta[t * NJ + j];
  const double prox = update_joints<NJ>(lc, dq, dt, 1.0 / dt, clamp_vel && !simulation, clamp_pos != 0, q, qd);
  io.put_joints(q, qd);
  if (prox_out) prox_out[t] = prox;
}

// Leg::applyIK(simulation) towards the stored desired tip pose, including the rotation-constrained pass, the unconstrained
// retry and the closing calculateTipForce (model.cpp:861-941).
template <int NJ>
__device__ __forceinline__ double apply_ik_dev(const DevState &st, const LegIO<NJ> &io, const LegConst<NJ> &lc, int simulation, double dt, int clamp_vel,
                                               int clamp_pos, int tip_force, double force_gain) {
  using FD = Fields<NJ>;
  double q[NJ], qd[NJ], dq[NJ];
  io.joints(q, qd);
  const V3 desired = io.get3(FD::DES_TIP);
  bool constrained = io.get(FD::DES_TIP + 3) != 0.0;
  const V3 desired_dir = io.get3(FD::DES_DIR);
  const bool cv = clamp_vel && !simulation, cp = clamp_pos != 0;
  const double inv_dt = 1.0 / dt;
  Chain<NJ> ch;
  V3 lin[NJ];
  double success = 0.0;
  bool failed = false;
  V3 tf = io.get3(FD::TF);
  for (int pass = 0; pass < 2; ++pass) { // second pass = the unconstrained retry (a nested applyIK, :932-936)
    fk_chain<NJ>(lc, q, ch);
    const V3 current_dir = ch.xe;
    ik_step<NJ>(lc, ch, q, qd, desired, dq);
    if (constrained) {
      update_joints<NJ>(lc, dq, dt, inv_dt, false, cp, q, qd); // simulation = true (:883)
      fk_chain<NJ>(lc, q, ch);
      jacobian_columns<NJ>(ch, lin);
      ik_step_rotation<NJ>(lc, ch, lin, q, qd, tip_rotation_delta(current_dir, base_rotate_inv(lc, desired_dir)), dq);
    }
    success = update_joints<NJ>(lc, dq, dt, inv_dt, cv, cp, q, qd);
    fk_chain<NJ>(lc, q, ch);
    const V3 e = tip_robot_frame(lc, ch.pe) - desired;
    if (fabs(e.x) > kIkTolerance || fabs(e.y) > kIkTolerance || fabs(e.z) > kIkTolerance) {
      success = 0.0;
      failed = true;
    }
    const bool retry = constrained && success == 0.0;
    if (!retry || pass == 1) {
      if (tip_force) { // calculateTipForce closes every applyIK frame: once here, and once more for the outer frame after a retry
        double effort[NJ];
        for (int j = 0; j < NJ; ++j) effort[j] = io.get(FD::EFFORT_IN + j);
        jacobian_columns<NJ>(ch, lin);
        const V3 raw = tip_force_cols<NJ>(lc, ch, lin, effort);
        tf = raw * (0.15 * force_gain) + tf * (1 - 0.15);
        if (pass == 1) tf = raw * (0.15 * force_gain) + tf * (1 - 0.15);
      }
      break;
    }
    constrained = false; // desired_tip_pose_.rotation_ = UNDEFINED_ROTATION
    io.put(FD::DES_TIP + 3, 0.0);
  }
  io.put_joints(q, qd);
  if (tip_force) io.put3(FD::TF, tf);
  if (!simulation) { // the deviation warning is the engine's IK-failure bit (model.cpp:921, not raised in simulation)
    int w = st.legi[io.slot];
    w = failed ? (w | LW_IKFAIL) : (w & ~LW_IKFAIL);
    st.legi[io.slot] = w;
  }
  return success;
}
template <int L_, int NJ>
__global__ void leg_apply_ik_kernel(DevState st, const SharedConsts<L_, NJ> *gc, LegSel sel, int simulation, double *result_out, double dt,
                                    int clamp_vel, int clamp_pos, int tip_force, double force_gain) {
  int64_t rob;
  int l;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (!sel.map(t, rob, l)) return;
  const LegIO<NJ> io{st, slot_of(rob, l, sel.L)};
  const double success = apply_ik_dev<NJ>(st, io, gc->leg[l], simulation, dt, clamp_vel, clamp_pos, tip_force, force_gain);
  if (result_out) result_out[t] = success;
}

// Leg::applyFK: tip pose (robot frame) from the desired joint positions, or from `joint_position` when given (use_actual).
template <int L_, int NJ>
__global__ void leg_apply_fk_kernel(DevState st, const SharedConsts<L_, NJ> *gc, LegSel sel, const double *joint_position, double *tip_pose) {
  int64_t rob;
  int l;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (!sel.map(t, rob, l)) return;
  const LegIO<NJ> io{st, slot_of(rob, l, sel.L)};
  double q[NJ], qd[NJ];
  io.joints(q, qd);
  if (joint_position)
    for (int j = 0; j < NJ; ++j) q[j] = joint_position[t * NJ + j];
  const Pose p = fk_tip_pose<NJ>(gc->leg[l], q);
  double *o = tip_pose + t * 7;
  o[0] = p.p.x, o[1] = p.p.y, o[2] = p.p.z, o[3] = p.r.w, o[4] = p.r.x, o[5] = p.r.y, o[6] = p.r.z;
}

// ------------------------------------------------------------------------------------------------- sequences
// LegPoser::stepToPosition (pose_controller.cpp:1571-1712), one iteration per selected leg: the tip follows two quartic
// Bezier curves from where the leg was when the sequence started (origin_tip_pose_) to the target (with an optional lift)
// while the body pose eases from the identity to target_pose; the result is LegPoser::current_tip_pose_, which the callers
// hand to Leg::setDesiredTipPose + applyIK (stepToNewStance :521, poseForLegManipulation :561, directStartup :463).
constexpr double kUndefinedPosition = 2147483647.0; // UNDEFINED_POSITION = double(INT_MAX) per component (standard_includes.h:57)
template <int NJ>
__device__ __forceinline__ int step_to_position_dev(const DevState &st, const LegIO<NJ> &io, const LegConst<NJ> &lc, const double *target7, const Pose &body,
                                                    double lift_height, double time_to_step, int apply_delta, int have_adm, double dt, Pose &out_pose,
                                                    int leg_state = LS_WALKING) {
  using FD = Fields<NJ>;
  V3 origin = io.get3(FD::SEQ_ORG), origin_dir = io.get3(FD::SEQ_DIR);
  int count = int(io.get(FD::SEQ_ORG + 3));
  if (io.get(FD::SEQ_DIR + 3) == 0.0) { // first_iteration_: origin_tip_pose_ = leg_->getCurrentTipPose() (FK of the current joints)
    double q[NJ], qd[NJ];
    io.joints(q, qd);
    Chain<NJ> ch;
    fk_chain<NJ>(lc, q, ch);
    origin = tip_robot_frame(lc, ch.pe);
    origin_dir = base_rotate(lc, ch.xe);
    count = 0;
  }
  V3 desired = origin, desired_dir{0, 0, 0};
  bool rot = false;
  if (target7) { // NULL = Pose::Undefined(): stay at the origin position, rotation undefined (:1583-1587)
    const double *p = target7;
    desired = V3{p[0], p[1], p[2]};
    const Quat r{p[3], p[4], p[5], p[6]};
    rot = !(r.w == 0.0 && r.x == 0.0 && r.y == 0.0 && r.z == 0.0);
    if (rot) desired_dir = rotate(r, V3{1, 0, 0});
  }
  const bool move = norm(origin - inverse_transform_vector(body, desired)) > kTipTolerance;
  bool turn = false;
  if (rot) turn = norm(angle_axis_vector(from_two_vectors(origin_dir, desired_dir))) > kJointTolerance;
  Pose out{origin, from_two_vectors(V3{1, 0, 0}, origin_dir)};
  int progress = 100;
  double running = 0.0;
  if (move || turn || lift_height != 0.0) {
    // "Apply delta z to target tip position": not to manually manipulated legs (:1610-1614)
    if (apply_delta && have_adm && leg_state != LS_MANUAL && leg_state != LS_WALKING_TO_MANUAL) desired = desired + io.get3(FD::ADM_DELTA);
    ++count;
    int num = round_to_int(time_to_step / dt);
    num = num > 1 ? num : 1;
    const double delta_t = 1.0 / num, ratio = double(count - 1) / double(num);
    const Pose eased = interpolate_pose(pose_identity(), smooth_step(ratio), body);
    out.r = Quat{0, 0, 0, 0};
    if (rot) out.r = from_two_vectors(V3{1, 0, 0}, normalized(lerp3(origin_dir, desired_dir, smooth_step(ratio))));
    const int half = num / 2;
    const V3 o2t = origin - desired;
    V3 prim[5] = {origin, origin, origin, desired + o2t * 0.75, desired + o2t * 0.5};
    V3 sec[5] = {desired + o2t * 0.5, desired + o2t * 0.25, desired, desired, desired};
    prim[2].z += lift_height, prim[3].z += lift_height, prim[4].z += lift_height;
    sec[0].z += lift_height, sec[1].z += lift_height, sec[2].z += lift_height;
    const int sic = (count + (num - 1)) % num + 1;
    V3 np = origin; // a target whose position is UNDEFINED_POSITION (transitionStance with gravity-aligned tips and no tip target): stay (:1637-1638)
    if (desired.x != kUndefinedPosition || desired.y != kUndefinedPosition || desired.z != kUndefinedPosition)
      np = sic <= half ? quartic_bezier(prim, sic * delta_t * 2.0) : quartic_bezier(sec, (sic - half) * delta_t * 2.0);
    out.p = inverse_transform_vector(eased, np);
    if (leg_state == LS_MANUAL) { // a MANUAL leg keeps the tip pose updateStance gave its LegPoser: the stepper's own (:1680-1684, pose_controller.cpp:134-137)
      out.p = io.get3(FD::TIP);
      out.r = Quat{0, 0, 0, 0};
      if (st.legi[io.slot] & LW_ROTDEF) out.r = from_two_vectors(V3{1, 0, 0}, io.get3(FD::CUR_DIR));
    }
    if (count >= num) {
      progress = 100; // first_iteration_ = true
    } else {
      progress = int(ratio * 100);
      running = 1.0;
    }
  }
  io.put3(FD::SEQ_ORG, origin);
  io.put(FD::SEQ_ORG + 3, double(count));
  io.put3(FD::SEQ_DIR, origin_dir);
  io.put(FD::SEQ_DIR + 3, running);
  out_pose = out;
  return progress;
}
template <int L_, int NJ>
__global__ void leg_step_to_position_kernel(DevState st, const SharedConsts<L_, NJ> *gc, LegSel sel, const double *target_tip_pose,
                                            const double *target_pose, double lift_height, double time_to_step, int apply_delta, int have_adm,
                                            double dt, double *tip_pose_out, int32_t *progress_out) {
  int64_t rob;
  int l;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (!sel.map(t, rob, l)) return;
  const LegIO<NJ> io{st, slot_of(rob, l, sel.L)};
  const double *tp = target_pose + (rob - sel.first) * 7;
  const Pose body{V3{tp[0], tp[1], tp[2]}, Quat{tp[3], tp[4], tp[5], tp[6]}};
  Pose out;
  const int progress = step_to_position_dev<NJ>(st, io, gc->leg[l], target_tip_pose ? target_tip_pose + t * 7 : nullptr, body, lift_height, time_to_step,
                                                apply_delta, have_adm, dt, out, leg_state_of(st, rob, l));
  double *o = tip_pose_out + t * 7;
  o[0] = out.p.x, o[1] = out.p.y, o[2] = out.p.z, o[3] = out.r.w, o[4] = out.r.x, o[5] = out.r.y, o[6] = out.r.z;
  if (progress_out) progress_out[t] = progress;
}

// LegPoser::transitionConfiguration (pose_controller.cpp:1476-1567), one iteration for one leg: every joint follows a cubic
// Bezier (nodes origin, origin, target, target) from the configuration the leg had when the transition started.
template <int NJ>
__device__ __forceinline__ int transition_configuration_dev(const LegIO<NJ> &io, const double *d, double transition_time, double dt) {
  using FD = Fields<NJ>;
  double q[NJ], qd[NJ], q0[NJ];
  io.joints(q, qd);
  int count = int(io.get(FD::SEQ_ORG + 3));
  if (io.get(FD::SEQ_DIR + 3) == 0.0) { // first_iteration_: origin_configuration_ = the current desired joint positions
    for (int j = 0; j < NJ; ++j) q0[j] = q[j];
    count = 0;
  } else {
    for (int j = 0; j < NJ; ++j) q0[j] = io.get(FD::SEQ_Q0 + j);
  }
  int num = round_to_int(transition_time / dt);
  num = num > 1 ? num : 1;
  const double delta_t = 1.0 / num;
  ++count;
  const double tt = count * delta_t, s = 1.0 - tt;
  for (int j = 0; j < NJ; ++j) // cubicBezier (standard_includes.h:347)
    q[j] = q0[j] * (s * s * s) + q0[j] * (3.0 * tt * s * s) + d[j] * (3.0 * tt * tt * s) + d[j] * (tt * tt * tt);
  int progress = int((double(count - 1) / double(num)) * 100);
  progress = progress < 1 ? 1 : (progress > 100 ? 100 : progress);
  double running = 1.0;
  if (count >= num) {
    progress = 100;
    running = 0.0;
  }
  io.put_joints(q, qd); // desired_velocity_ is not touched by the reference here
  for (int j = 0; j < NJ; ++j) io.put(FD::SEQ_Q0 + j, q0[j]);
  io.put(FD::SEQ_ORG + 3, double(count));
  io.put(FD::SEQ_DIR + 3, running);
  return progress;
}
template <int L_, int NJ>
__global__ void leg_transition_configuration_kernel(DevState st, const SharedConsts<L_, NJ> *gc, LegSel sel, const double *desired_configuration,
                                                    int per_leg_rows, double transition_time, double dt, int32_t *progress_out) {
  int64_t rob;
  int l;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (!sel.map(t, rob, l)) return;
  const LegIO<NJ> io{st, slot_of(rob, l, sel.L)};
  // per_leg_rows: one target row per selected (instance, leg), else one row per LEG shared by every instance ([legs][dof])
  const int progress = transition_configuration_dev<NJ>(io, desired_configuration + (per_leg_rows ? t : int64_t(l)) * NJ, transition_time, dt);
  if (progress_out) progress_out[t] = progress;
}

} // namespace shc

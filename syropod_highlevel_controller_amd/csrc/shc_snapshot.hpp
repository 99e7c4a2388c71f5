// shc_snapshot.hpp — shc_engine_get_state / shc_engine_set_state: the engine's SoA planes <-> shc_instance_state records
// (include/shc_batch.h).  Checkpoint / restore and state injection; not on the per-cycle path.
//
// The record speaks the reference's member names (LegStepper / LegPoser / WalkController / PoseController members, cited in
// the header); the engine's packed words and direction vectors are converted here:
//   * swing_progress_ / stance_progress_  <->  the 2-bit "progress mode" + the phase (walk_controller.cpp:871-897 writes
//     them from the phase alone, so the pair is a function of (mode, phase));
//   * the stepping legs' copies of the walk plane (LegStepper::walk_plane_) are one per robot in the engine.
#pragma once

#include "shc_cycle.hpp"

namespace shc {

__device__ inline void snap_put3(double *dst, V3 v) {
  dst[0] = v.x;
  dst[1] = v.y;
  dst[2] = v.z;
}

template <int NJ>
__global__ void get_state_kernel(shc_instance_state *out, DevState st, CycleParams P, int L, int64_t first, int64_t count, int touchdown, unsigned long_legs) {
  // long_legs: bit l = leg l has more than 3 joints (its stepper tracks a tip rotation under gravity_aligned_tips / rough terrain mode; a shorter
  // leg of the same robot, padded up to the kernel's joint count, does not - its record fields stay zero, as for a robot of 3-joint legs)
  using FD = Fields<NJ>;
  using R = RobotFields;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int64_t rob = first + t;
  const int rpw = 64 / L;
  shc_instance_state &o = out[t];
  auto rd = [&](int f) { return st.robd[rob_index(rob, f, rpw, R::COUNT)]; };
  auto ri = [&](int f) { return st.robi[rob_index(rob, f, rpw, R::I_COUNT)]; };
  o.desired_linear_velocity[0] = rd(R::VLIN);
  o.desired_linear_velocity[1] = rd(R::VLIN + 1);
  o.desired_angular_velocity = rd(R::VANG);
  for (int k = 0; k < 3; ++k) {
    o.walk_plane[k] = rd(R::PLANE + k);
    o.walk_plane_normal[k] = rd(R::PNORM + k);
    o.stepper_walk_plane[k] = rd(R::PLANE_PREV + k);
    o.stepper_walk_plane_normal[k] = rd(R::PNORM_PREV + k);
    o.translation_velocity_input[k] = rd(R::TVI + k);
    o.rotation_velocity_input[k] = rd(R::RVI + k);
    o.rotation_absement_error[k] = rd(R::ABSE + k);
    o.rotation_velocity_error[k] = rd(R::VERR + k);
  }
  for (int k = 0; k < 7; ++k) {
    o.origin_walk_plane_pose[k] = rd(R::OWPP + k);
    o.manual_pose[k] = rd(R::MPOSE + k);
    o.current_pose[k] = rd(R::CPOSE + k);
    o.odometry[k] = 0.0;
    o.tip_align_pose[k] = rd(R::TALIGN + k);
    o.origin_tip_align_pose[k] = rd(R::OTALIGN + k);
  }
  for (int k = 0; k < 4; ++k) o.auto_pose_rotation[k] = rd(R::APREV + k);
  o.odometry[0] = rd(R::ODOM); // stored as (x, y, qw, qz): pure yaw
  o.odometry[1] = rd(R::ODOM + 1);
  o.odometry[3] = rd(R::ODOM + 2);
  o.odometry[6] = rd(R::ODOM + 3);
  const int rword = ri(R::I_WORD);
  o.walk_state = rword & 3;
  o.legs_at_correct_phase = (rword >> RW_LACP_SHIFT) & 15;
  o.legs_completed_first_step = (rword >> RW_LCFS_SHIFT) & 15;
  o.return_to_default_attempted = (rword & RW_RTDA) ? 1 : 0;
  o.auto_posing_state = (rword >> RW_APS_SHIFT) & 3;
  o.pose_phase = ri(R::I_POSE_PHASE);
  const int ap = ri(R::I_APOSER);
  for (int i = 0; i < SHC_MAX_AUTO_POSERS; ++i) o.auto_poser_flags[i] = (ap >> (4 * i)) & 15;
  o.touchdown_detection = touchdown;
  o.pad_ = 0;
  for (int l = 0; l < SHC_MAX_LEGS; ++l) {
    shc_leg_snapshot &g = o.leg[l];
    __builtin_memset(&g, 0, sizeof g);
    if (l >= L) continue;
    const int64_t slot = slot_of(rob, l, L);
    auto f = [&](int field) { return st.legd[leg_field_index(field, slot, st.n_slots)]; };
    for (int j = 0; j < NJ; ++j) {
      g.joint_position[j] = f(FD::Q + j);
      g.joint_velocity[j] = f(FD::QD + j);
    }
    for (int k = 0; k < 3; ++k) {
      g.walker_tip[k] = f(FD::TIP + k);
      g.walker_tip_velocity[k] = f(FD::TVEL + k);
      g.swing_origin_tip[k] = f(FD::SORG + k);
      g.swing_origin_tip_velocity[k] = f(FD::SVEL + k);
      g.stance_origin_tip[k] = f(FD::TORG + k);
      g.default_tip[k] = f(FD::DFLT + k);
      g.target_tip[k] = f(FD::TARG + k);
      g.stride_vector[k] = f(FD::STRD + k);
      g.admittance_delta[k] = f(FD::ADM_DELTA + k);
      g.tip_force_calculated[k] = f(FD::TF + k);
    }
    g.admittance_state[0] = f(FD::ADM);
    g.admittance_state[1] = f(FD::ADM + 1);
    g.virtual_stiffness = f(FD::ADM_DELTA + 3);
    const int w = st.legi[slot];
    g.step_state = w & 3;
    g.phase = (w >> LW_PHASE_SHIFT) & LW_PHASE_MASK;
    g.at_correct_phase = (w & LW_ACP) ? 1 : 0;
    g.completed_first_step = (w & LW_CFS) ? 1 : 0;
    g.negate_auto_pose = (w & LW_NEG) ? 1 : 0;
    g.ik_failed = (w & LW_IKFAIL) ? 1 : 0;
    g.step_plane_defined = (P.rough_terrain && f(FD::STEP_PLANE + 3) != 0.0) ? 1 : 0;
    if (g.step_plane_defined)
      for (int k = 0; k < 3; ++k) g.step_plane_position[k] = f(FD::STEP_PLANE + k);
    // LegStepper::iteratePhase (walk_controller.cpp:871-897)
    const int pm = (w >> LW_PM_SHIFT) & 3;
    g.swing_progress = g.stance_progress = -1.0; // walk_controller.h:498-499
    if (pm == PM_SWING) {
      g.swing_progress = clampd(double(g.phase - P.swing_start + 1) / double(P.swing_end - P.swing_start), 0.0, 1.0);
    } else if (pm == PM_STANCE) {
      g.stance_progress = clampd(double(mod_i(g.phase + (P.period - P.stance_start), P.period) + 1) /
                                     double(mod_i(P.stance_end - P.stance_start, P.period)),
                                 0.0, 1.0);
    } else if (pm == PM_STOP) {
      g.stance_progress = 0.0;
    }
    if ((P.gravity_aligned && ((long_legs >> l) & 1u)) || P.joint_control == 2) { // tip rotations tracked (joint_control: a MANUAL 3-joint leg holds its FK tip rotation)
      for (int k = 0; k < 3; ++k) {
        g.origin_tip_direction[k] = f(FD::ORG_DIR + k);
        g.walker_tip_direction[k] = f(FD::CUR_DIR + k);
        g.target_tip_direction[k] = (w & LW_TARGROT) ? f(FD::TARG_DIR + k) : 0.0;
      }
      g.tip_rotation_defined = (w & LW_ROTDEF) ? 1 : 0;
      g.target_rotation_defined = (w & LW_TARGROT) ? 1 : 0;
    }
  }
}

template <int NJ>
__global__ void set_state_kernel(const shc_instance_state *in, DevState st, CycleParams P, int L, int64_t first, int64_t count, unsigned long_legs) {
  // long_legs: as in get_state_kernel - a shorter leg of a gravity-aligned robot tracks no tip rotation: its direction planes and word bits stay
  // what the engine holds (the record carries zeros for them), so that get_state -> set_state is the identity on every leg
  using FD = Fields<NJ>;
  using R = RobotFields;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int64_t rob = first + t;
  const int rpw = 64 / L;
  const shc_instance_state &o = in[t];
  auto wr = [&](int f, double v) { st.robd[rob_index(rob, f, rpw, R::COUNT)] = v; };
  auto wi = [&](int f, int v) { st.robi[rob_index(rob, f, rpw, R::I_COUNT)] = v; };
  wr(R::VLIN, o.desired_linear_velocity[0]);
  wr(R::VLIN + 1, o.desired_linear_velocity[1]);
  wr(R::VANG, o.desired_angular_velocity);
  for (int k = 0; k < 3; ++k) {
    wr(R::PLANE + k, o.walk_plane[k]);
    wr(R::PNORM + k, o.walk_plane_normal[k]);
    wr(R::PLANE_PREV + k, o.stepper_walk_plane[k]);
    wr(R::PNORM_PREV + k, o.stepper_walk_plane_normal[k]);
    wr(R::TVI + k, o.translation_velocity_input[k]);
    wr(R::RVI + k, o.rotation_velocity_input[k]);
    wr(R::ABSE + k, o.rotation_absement_error[k]);
    wr(R::VERR + k, o.rotation_velocity_error[k]);
  }
  for (int k = 0; k < 7; ++k) {
    wr(R::OWPP + k, o.origin_walk_plane_pose[k]);
    wr(R::MPOSE + k, o.manual_pose[k]);
    wr(R::CPOSE + k, o.current_pose[k]);
    wr(R::TALIGN + k, o.tip_align_pose[k]);
    wr(R::OTALIGN + k, o.origin_tip_align_pose[k]);
  }
  for (int k = 0; k < 4; ++k) wr(R::APREV + k, o.auto_pose_rotation[k]);
  wr(R::ODOM, o.odometry[0]);
  wr(R::ODOM + 1, o.odometry[1]);
  wr(R::ODOM + 2, o.odometry[3]);
  wr(R::ODOM + 3, o.odometry[6]);
  wi(R::I_WORD, (o.walk_state & 3) | ((o.legs_at_correct_phase & 15) << RW_LACP_SHIFT) | ((o.legs_completed_first_step & 15) << RW_LCFS_SHIFT) |
                    (o.return_to_default_attempted ? RW_RTDA : 0) | ((o.auto_posing_state & 3) << RW_APS_SHIFT));
  wi(R::I_POSE_PHASE, o.pose_phase);
  int ap = 0;
  for (int i = 0; i < SHC_MAX_AUTO_POSERS; ++i) ap |= (o.auto_poser_flags[i] & 15) << (4 * i);
  wi(R::I_APOSER, ap);
  for (int l = 0; l < L; ++l) {
    const shc_leg_snapshot &g = o.leg[l];
    const int64_t slot = slot_of(rob, l, L);
    auto f = [&](int field, double v) { st.legd[leg_field_index(field, slot, st.n_slots)] = v; };
    for (int j = 0; j < NJ; ++j) {
      f(FD::Q + j, g.joint_position[j]);
      f(FD::QD + j, g.joint_velocity[j]);
    }
    for (int k = 0; k < 3; ++k) {
      f(FD::TIP + k, g.walker_tip[k]);
      f(FD::TVEL + k, g.walker_tip_velocity[k]);
      f(FD::SORG + k, g.swing_origin_tip[k]);
      f(FD::SVEL + k, g.swing_origin_tip_velocity[k]);
      f(FD::TORG + k, g.stance_origin_tip[k]);
      f(FD::DFLT + k, g.default_tip[k]);
      f(FD::TARG + k, g.target_tip[k]);
      f(FD::STRD + k, g.stride_vector[k]);
      f(FD::ADM_DELTA + k, g.admittance_delta[k]);
      f(FD::TF + k, g.tip_force_calculated[k]);
    }
    f(FD::ADM, g.admittance_state[0]);
    f(FD::ADM + 1, g.admittance_state[1]);
    f(FD::ADM_DELTA + 3, g.virtual_stiffness);
    for (int k = 0; k < 3; ++k) f(FD::STEP_PLANE + k, g.step_plane_defined ? g.step_plane_position[k] : 0.0);
    f(FD::STEP_PLANE + 3, g.step_plane_defined ? 1.0 : 0.0);
    int pm = PM_NONE;
    if (g.swing_progress >= 0.0) pm = PM_SWING;
    else if (g.stance_progress > 0.0) pm = PM_STANCE;
    else if (g.stance_progress == 0.0) pm = PM_STOP;
    int w = (g.step_state & 3) | (g.at_correct_phase ? LW_ACP : 0) | (g.completed_first_step ? LW_CFS : 0) | (pm << LW_PM_SHIFT) |
            (g.negate_auto_pose ? LW_NEG : 0) | (g.ik_failed ? LW_IKFAIL : 0) | ((g.phase & LW_PHASE_MASK) << LW_PHASE_SHIFT);
    if ((P.gravity_aligned && ((long_legs >> l) & 1u)) || P.joint_control == 2) { // tip rotations tracked (joint_control: a MANUAL 3-joint leg holds its FK tip rotation)
      for (int k = 0; k < 3; ++k) {
        f(FD::ORG_DIR + k, g.origin_tip_direction[k]);
        f(FD::CUR_DIR + k, g.walker_tip_direction[k]);
        f(FD::TARG_DIR + k, g.target_tip_direction[k]);
      }
      if (g.tip_rotation_defined) w |= LW_ROTDEF;
      if (g.target_rotation_defined) w |= LW_TARGROT;
    }
    st.legi[slot] = w;
  }
}

} // namespace shc

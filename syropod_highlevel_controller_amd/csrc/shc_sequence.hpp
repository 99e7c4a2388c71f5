// shc_sequence.hpp — the start-up / shut-down choreography of the reference, batched: one PoseController per instance.
//   PoseController::executeSequence   src/pose_controller.cpp:145-459   (START_UP / SHUT_DOWN, transition poses learnt on the first run)
//   PoseController::stepToNewStance   src/pose_controller.cpp:521-557
// Each call of the kernels below is ONE call of the reference method for every instance: a per-robot state machine
// (PoseController's sequence members, pose_controller.h:273-304, and the LegPosers' transition poses, :591-593) over the per-leg
// primitives of shc_leg_api.hpp (LegPoser::stepToPosition, Leg::setDesiredTipPose, Leg::applyIK).  Cold path (runs once per
// power-up): one thread per robot walks its legs in id order like the reference does; the state lives in a lazily allocated
// array of records, not in the SoA planes of the control cycle.
#pragma once

#include "shc_leg_api.hpp"

namespace shc {

constexpr int kMaxTransitionPoses = 32;    // executeSequence gives up beyond TRANSITION_STEP_THRESHOLD = 20 steps (pose_controller.h:24)
constexpr double kSafetyFactor = 0.15;     // SAFETY_FACTOR (pose_controller.h:20)
constexpr double kHorizontalTransitionTime = 1.0, kVerticalTransitionTime = 3.0; // :21-22
constexpr int kTransitionStepThreshold = 20;
constexpr double kHalfBodyDepth = 0.05;    // HALF_BODY_DEPTH (model.h:18)

struct SeqLegState {                        // class LegPoser
  double transition[kMaxTransitionPoses][3]; // transition_poses_[k].position_ (the only member read back, :224, :320)
  double target[7];                          // target_tip_pose_
  double current[7];                         // current_tip_pose_ as the last stepToPosition left it
  int32_t n_poses, completed;                // transition_poses_.size(), leg_completed_step_
  double plan_configuration[SHC_MAX_JOINTS]; // PoseController::target_configuration_, this leg's joints (planner mode)
  double plan_desired[SHC_MAX_JOINTS];       // LegPoser::desired_configuration_ as latched when the transition began
  int32_t plan_named, plan_latched;          // ... whether the message names the leg at all; LegPoser::desired_configuration_ defined (latched when a transition begins)
};
struct SeqRobotState {                       // class PoseController (pose_controller.h:273-274, :296-304)
  int32_t legs_completed_step, current_group, transition_step, transition_step_count;
  int32_t set_target, proximity_alert, horizontal_transition_complete, vertical_transition_complete;
  int32_t first_sequence_execution, reset_transition_sequence, failed, initialised;
  int32_t completed_sequence, pad_; // 1 + the sequence this robot has completed and not left since (see execute_sequence_kernel)
  // planner mode: StateController::plan_step_ / target_*_acquired_ (state_controller.h:360-363), PoseController::executing_transition_
  // and target_body_pose_ (pose_controller.h:292); poser_tip_from_plan: LegPoser::current_tip_pose_ is what transitionStance left
  // in leg[].current (else it is what the last control cycle's updateStance derives from the walker tips)
  int32_t plan_step, configuration_acquired, tip_pose_acquired, body_pose_acquired, executing_transition, poser_tip_from_plan;
  double target_body_pose[7];
  SeqLegState leg[SHC_MAX_LEGS];
};

struct SeqParams {
  double step_frequency, swing_height, dt, force_gain;
  double target_rotation[4]; // LegStepper::target_tip_pose_.rotation_ while the robot has not walked: the identity tip rotation
                             // (walk_controller.cpp:37-41), UNDEFINED_ROTATION (zeros) without gravity-aligned tips
  int clamp_vel, clamp_pos, tip_force, have_adm, gravity_aligned, inclination_posing;
  int gravity_aligned_tips; // params.gravity_aligned_tips (transitionStance's target rotation)
  int pose_pass;      // the body pose moves while a robot stands (IMU / auto / inclination posing): LOOP_MARK / LOOP_AFTER_POSE below
  int poser_tip_kept; // the cycle kernels store every LegPoser tip (auto posing without IMU posing: the per-leg auto pose is not re-derivable)
  int posed;          // the posing part of this loop (PoseController::updateCurrentPose + the admittance update, state_controller.cpp:165-181) has already
                      // run in the cycle kernel for the robots of this call (RT_POSE_MARKED): auto posing on its own clock keeps posing through a sequence
};
// executeSequence / stepToNewStance under auto posing on its own clock (pose_frequency != -1: the pose phase counter advances in every loop, the posers latch
// on without a step cycle, pose_controller.cpp:1134-1187, :1359-1371): the robots whose sequence is still running are marked, the cycle kernel runs the posing
// part of their loop (pose-only pass), the sequence kernel follows with Model::current_pose_ as that pass left it, and the marks are cleared.
__global__ void sequence_mark_kernel(ManualRobot *manual, const SeqRobotState *seq, int64_t n, int which, int set) {
  const int64_t rob = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (rob >= n) return;
  const bool completed = which < 2 && seq[rob].initialised && seq[rob].completed_sequence == which + 1; // (left alone: its node would have left transitionRobotState)
  manual[rob].skip_cycle = (set && !completed) ? 1 : 0;
}
// One StateController::loop() of the robots that stand while a leg toggle / plan step runs = posing part (:165-181), then legStateToggle /
// executePlan.  With walk-plane + manual posing only, the loop-level kernel does both (LOOP_WHOLE).  With time-dependent posing (IMU / auto /
// inclination) the posing part is the cycle kernel's: the loop-level kernel first only marks its robots (LOOP_MARK), the cycle kernel
// runs PoseController::updateCurrentPose + the admittance update for them (RT_POSE_MARKED), then the loop-level kernel does the rest
// (LOOP_AFTER_POSE).
enum : int { LOOP_WHOLE = 0, LOOP_MARK = 1, LOOP_AFTER_POSE = 2 };

template <int NJ>
__device__ __forceinline__ Pose leg_current_tip_pose(const LegIO<NJ> &io, const LegConst<NJ> &lc) { // Leg::current_tip_pose_ = FK of the joints
  double q[NJ], qd[NJ];
  io.joints(q, qd);
  return fk_tip_pose<NJ>(lc, q);
}
__device__ __forceinline__ void put_pose7(double *o, V3 p, Quat r) { o[0] = p.x, o[1] = p.y, o[2] = p.z, o[3] = r.w, o[4] = r.x, o[5] = r.y, o[6] = r.z; }
__device__ __forceinline__ void put_pose7(double *o, const Pose &p) { put_pose7(o, p.p, p.r); }
template <int L>
__device__ __forceinline__ Pose robot_current_pose(const DevState &st, int64_t rob) { // Model::current_pose_
  using R = RobotFields;
  double v[7];
  for (int k = 0; k < 7; ++k) v[k] = st.robd[rob_index(rob, R::CPOSE + k, 64 / L, R::COUNT)];
  return Pose{V3{v[0], v[1], v[2]}, Quat{v[3], v[4], v[5], v[6]}};
}
__device__ __forceinline__ void seq_defaults(SeqRobotState &s) { // member initialisers (pose_controller.h:296-304)
  if (s.initialised) return;
  s.set_target = 1;
  s.first_sequence_execution = 1;
  s.reset_transition_sequence = 1;
  s.target_body_pose[3] = 1.0; // identity (the reference leaves it uninitialised until the first plan step completes, state_controller.cpp:685)
  s.initialised = 1;
}

// PoseController::updateCurrentPose for a robot none of whose legs swings, with walk-plane + manual posing (what the loop-level
// kernels support): the walk-plane pose is its origin (no swing progress to interpolate with, pose_controller.cpp:1100-1128) and
// the manual pose stays as it is without inputs, so Model::current_pose_ = origin_walk_plane_pose_ (+) manual_pose_.  The start-up
// sequence runs before the first control cycle has ever written the pose.
// With gravity_aligned_tips on a robot whose leg 0 has at most 3 joints the tip-align pose is part of the pose as well: updateTipAlignPose only moves it
// while a leg swings (pose_controller.cpp:1024-1088), so a robot that stands keeps the pose its last step left - added last, as updateCurrentPose does (:849-856).
template <int L>
__device__ __forceinline__ void standing_pose_prologue_dev(const DevState &st, int64_t rob, const bool tip_align = false) {
  using R = RobotFields;
  constexpr int rpw = 64 / L;
  auto rd = [&](int f) -> double & { return st.robd[rob_index(rob, f, rpw, R::COUNT)]; };
  const Pose owpp{V3{rd(R::OWPP), rd(R::OWPP + 1), rd(R::OWPP + 2)}, Quat{rd(R::OWPP + 3), rd(R::OWPP + 4), rd(R::OWPP + 5), rd(R::OWPP + 6)}};
  const Pose manual{V3{rd(R::MPOSE), rd(R::MPOSE + 1), rd(R::MPOSE + 2)}, Quat{rd(R::MPOSE + 3), rd(R::MPOSE + 4), rd(R::MPOSE + 5), rd(R::MPOSE + 6)}};
  Pose cp = add_pose(owpp, manual);
  if (tip_align) cp = add_pose(cp, Pose{V3{rd(R::TALIGN), rd(R::TALIGN + 1), rd(R::TALIGN + 2)}, Quat{rd(R::TALIGN + 3), rd(R::TALIGN + 4), rd(R::TALIGN + 5), rd(R::TALIGN + 6)}});
  rd(R::CPOSE) = cp.p.x, rd(R::CPOSE + 1) = cp.p.y, rd(R::CPOSE + 2) = cp.p.z;
  rd(R::CPOSE + 3) = cp.r.w, rd(R::CPOSE + 4) = cp.r.x, rd(R::CPOSE + 5) = cp.r.y, rd(R::CPOSE + 6) = cp.r.z;
}

// AdmittanceController::updateAdmittance for one leg (admittance_controller.cpp:22-63) as the posing part of a StateController
// loop runs it before transitionRobotState / legStateToggle (state_controller.cpp:172-180); the robot is STOPPED on these paths,
// so the dynamic stiffness update (:175) does not run.  Same arithmetic as the admittance block of the fused cycle.
template <int NJ>
__device__ __forceinline__ void admittance_prologue_dev(const LegIO<NJ> &io, const LegConst<NJ> &lc, const CycleParams &P) {
  using FD = Fields<NJ>;
  if (!P.admittance_control) return;
  const V3 f = (P.use_joint_effort ? io.get3(FD::TF) : io.get3(FD::FORCE_IN)) * P.force_gain;
  double a0 = io.get(FD::ADM), a1 = io.get(FD::ADM + 1);
  const double fi[3] = {f.x, f.y, f.z};
  double d[3];
  for (int i = 0; i < 3; ++i) {
    const double u = fmax(fi[i], 0.0);
    const double x0 = P.adm_m00 * a0 + P.adm_m01 * a1 + P.adm_g0 * u;
    const double x1 = P.adm_m10 * a0 + P.adm_m11 * a1 + P.adm_g1 * u;
    a0 = x0;
    a1 = x1;
    d[i] = clampd(-x0, -0.2, 0.2);
  }
  double q[NJ], qd[NJ];
  io.joints(q, qd);
  Chain<NJ> ch;
  fk_chain<NJ>(lc, q, ch);
  io.put(FD::ADM, a0);
  io.put(FD::ADM + 1, a1);
  io.put3(FD::ADM_DELTA, projection(V3{d[0], d[1], d[2]}, base_rotate(lc, ch.xe))); // Leg::setAdmittanceDelta (model.h:365-368)
}

// LegStepper::target_tip_pose_.rotation_ of ONE leg (pose_controller.cpp:238, :377: leg_stepper->getTargetTipPose().rotation_): the identity tip
// rotation only on legs of more than 3 joints (walk_controller.cpp:37), UNDEFINED (the zero quaternion) on a shorter leg - on a robot whose
// legs differ in DOF that is a per-leg fact of the padded chain (LegConst::jactive), not of the kernel's NJ.
template <int NJ>
__device__ __forceinline__ Quat leg_target_rotation(const LegConst<NJ> &lc, const Quat robot_target_rotation) {
  if constexpr (NJ > 3) {
    if (lc.jactive[3] == 0.0) return Quat{0, 0, 0, 0};
    return robot_target_rotation;
  } else {
    (void)lc;
    return robot_target_rotation; // (SeqParams::target_rotation is only set where the robot's longest leg has more than 3 joints)
  }
}

template <int L, int NJ>
__global__ void execute_sequence_kernel(DevState st, const SharedConsts<L, NJ> *gc, SeqRobotState *seq, int sequence /* 0 START_UP, 1 SHUT_DOWN */,
                                        SeqParams P, int32_t *progress_out) {
  using FD = Fields<NJ>;
  const int64_t rob = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (rob >= st.n_robots) return;
  SeqRobotState &s = seq[rob];
  seq_defaults(s);
  // The robots of a batch may finish after different numbers of calls (each learnt its own sequence): one that has completed
  // `sequence` is left alone - its node would have left transitionRobotState - until the other sequence is requested.
  if (s.completed_sequence == sequence + 1) {
    if (progress_out) progress_out[rob] = 100;
    return;
  }
  s.completed_sequence = 0;
  if (!P.posed) {
    standing_pose_prologue_dev<L>(st, rob, gc->P.tip_align != 0);
    for (int l = 0; l < L; ++l) admittance_prologue_dev<NJ>(LegIO<NJ>{st, slot_of(rob, l, L)}, gc->leg[l], gc->P); // posing part of the loop
  }
  const bool start_up = sequence == 0;
  // Initialise / reset any saved transition sequence (:149-162)
  if (s.reset_transition_sequence && start_up) {
    s.reset_transition_sequence = 0;
    s.first_sequence_execution = 1;
    s.transition_step = 0;
    for (int l = 0; l < L; ++l) {
      const LegIO<NJ> io{st, slot_of(rob, l, L)};
      const Pose tip = leg_current_tip_pose<NJ>(io, gc->leg[l]);
      s.leg[l].n_poses = 1; // resetTransitionSequence + addTransitionPose(leg->getCurrentTipPose())
      s.leg[l].transition[0][0] = tip.p.x, s.leg[l].transition[0][1] = tip.p.y, s.leg[l].transition[0][2] = tip.p.z;
    }
  }
  int progress = 0, normalised_progress = 0;
  int next_transition_step, transition_step_target, total_progress;
  bool execute_horizontal, execute_vertical;
  int count_or_one = s.transition_step_count > 1 ? s.transition_step_count : 1; // std::max(transition_step_count_, 1)
  if (start_up) {
    execute_horizontal = !(s.transition_step % 2);
    execute_vertical = s.transition_step % 2;
    next_transition_step = s.transition_step + 1;
    transition_step_target = s.transition_step_count;
    total_progress = s.transition_step * 100 / count_or_one;
  } else {
    execute_horizontal = s.transition_step % 2;
    execute_vertical = !(s.transition_step % 2);
    next_transition_step = s.transition_step - 1;
    transition_step_target = 0;
    total_progress = 100 - s.transition_step * 100 / count_or_one;
  }
  const bool first = s.first_sequence_execution != 0;
  const bool final_transition = first ? (s.horizontal_transition_complete || s.vertical_transition_complete) : (next_transition_step == transition_step_target);
  bool sequence_complete = false;
  const double safety_factor = first ? kSafetyFactor / (s.transition_step + 1) : 0.0;
  const Pose current_pose = robot_current_pose<L>(st, rob);
  const Quat target_rotation{P.target_rotation[0], P.target_rotation[1], P.target_rotation[2], P.target_rotation[3]};
  const int apply_delta = (start_up && final_transition) ? 1 : 0;
  const Pose identity = pose_identity();

  auto transition_target = [&](int l, const LegIO<NJ> &io) { // transition pose `next_transition_step`, else the default stance tip (:222-233)
    if (next_transition_step >= 0 && s.leg[l].n_poses > next_transition_step && next_transition_step < kMaxTransitionPoses)
      return V3{s.leg[l].transition[next_transition_step][0], s.leg[l].transition[next_transition_step][1], s.leg[l].transition[next_transition_step][2]};
    return inverse_transform_vector(current_pose, io.get3(FD::DFLT));
  };
  auto add_transition = [&](int l, const double *pose7) {
    SeqLegState &g = s.leg[l];
    if (g.n_poses < kMaxTransitionPoses) {
      g.transition[g.n_poses][0] = pose7[0], g.transition[g.n_poses][1] = pose7[1], g.transition[g.n_poses][2] = pose7[2];
      ++g.n_poses;
    }
  };

  if (execute_horizontal) {
    if (s.set_target) {
      s.set_target = 0;
      for (int l = 0; l < L; ++l) {
        const LegIO<NJ> io{st, slot_of(rob, l, L)};
        s.leg[l].completed = 0;
        V3 target = transition_target(l, io);
        target.z = leg_current_tip_pose<NJ>(io, gc->leg[l]).p.z; // maintain horizontal position
        put_pose7(s.leg[l].target, target, leg_target_rotation<NJ>(gc->leg[l], target_rotation));
      }
    }
    double height = 0.0; // Model::legsBearingLoad (model.cpp:78-88)
    for (int l = 0; l < L; ++l) height += leg_current_tip_pose<NJ>(LegIO<NJ>{st, slot_of(rob, l, L)}, gc->leg[l]).p.z;
    const bool direct_step = !(-(height / L) > kHalfBodyDepth);
    for (int l = 0; l < L; ++l) {
      SeqLegState &g = s.leg[l];
      if (g.completed) continue;
      if ((l % 2) == s.current_group || direct_step) { // Leg::group_ = id_number % 2 (model.cpp:187)
        const LegIO<NJ> io{st, slot_of(rob, l, L)};
        const double step_height = direct_step ? 0.0 : P.swing_height;
        double time_to_step = kHorizontalTransitionTime / P.step_frequency;
        time_to_step *= first ? 2.0 : 1.0;
        Pose tip;
        progress = step_to_position_dev<NJ>(st, io, gc->leg[l], g.target, identity, step_height, time_to_step, apply_delta, P.have_adm, P.dt, tip, leg_state_of(st, rob, l));
        put_pose7(g.current, tip);
        set_desired_dev<NJ>(st, io, L, rob, g.current, 1, P.have_adm, P.gravity_aligned);
        const double limit_proximity = apply_ik_dev<NJ>(st, io, gc->leg[l], 0, P.dt, P.clamp_vel, P.clamp_pos, P.tip_force, P.force_gain);
        const bool exceeded_workspace = limit_proximity < safety_factor;
        if (first && exceeded_workspace) { // stop the transition early (:264-270)
          for (int k = 0; k < 7; ++k) g.target[k] = g.current[k];
          io.put(FD::SEQ_DIR + 3, 0.0); // resetStepToPosition: first_iteration_ = true
          progress = 100;
          s.proximity_alert = 1;
        }
        if (progress == 100) {
          g.completed = 1;
          s.legs_completed_step++;
          if (first) add_transition(l, exceeded_workspace ? g.current : g.target);
        }
      } else {
        s.legs_completed_step++;
        g.completed = 1;
      }
    }
    count_or_one = s.transition_step_count > 1 ? s.transition_step_count : 1;
    normalised_progress = direct_step ? progress / count_or_one : (progress / 2 + (s.current_group == 0 ? 0 : 50)) / count_or_one;
    if (s.legs_completed_step == L) {
      s.set_target = 1;
      s.legs_completed_step = 0;
      if (s.current_group == 1 || direct_step) {
        s.current_group = 0;
        s.transition_step = next_transition_step;
        s.horizontal_transition_complete = !s.proximity_alert;
        sequence_complete = final_transition;
        s.proximity_alert = 0;
      } else if (s.current_group == 0) {
        s.current_group = 1;
      }
    }
  }

  if (execute_vertical) {
    if (s.set_target) {
      s.set_target = 0;
      for (int l = 0; l < L; ++l) {
        const LegIO<NJ> io{st, slot_of(rob, l, L)};
        V3 target = transition_target(l, io);
        const V3 tip = leg_current_tip_pose<NJ>(io, gc->leg[l]).p;
        target.x = tip.x, target.y = tip.y; // maintain horizontal position
        put_pose7(s.leg[l].target, target, leg_target_rotation<NJ>(gc->leg[l], target_rotation));
      }
    }
    bool all_legs_within_workspace = true;
    for (int l = 0; l < L; ++l) {
      SeqLegState &g = s.leg[l];
      const LegIO<NJ> io{st, slot_of(rob, l, L)};
      double time_to_step = kVerticalTransitionTime / P.step_frequency;
      time_to_step *= first ? 2.0 : 1.0;
      Pose tip;
      progress = step_to_position_dev<NJ>(st, io, gc->leg[l], g.target, identity, 0.0, time_to_step, apply_delta, P.have_adm, P.dt, tip, leg_state_of(st, rob, l));
      put_pose7(g.current, tip);
      set_desired_dev<NJ>(st, io, L, rob, g.current, 0, P.have_adm, P.gravity_aligned);
      const double limit_proximity = apply_ik_dev<NJ>(st, io, gc->leg[l], 0, P.dt, P.clamp_vel, P.clamp_pos, P.tip_force, P.force_gain);
      all_legs_within_workspace = all_legs_within_workspace && !(limit_proximity < safety_factor);
    }
    if ((!all_legs_within_workspace && first) || progress == 100) {
      for (int l = 0; l < L; ++l) {
        const LegIO<NJ> io{st, slot_of(rob, l, L)};
        io.put(FD::SEQ_DIR + 3, 0.0); // resetStepToPosition
        progress = 100;
        if (first) add_transition(l, all_legs_within_workspace ? s.leg[l].target : s.leg[l].current);
      }
      s.vertical_transition_complete = all_legs_within_workspace;
      s.transition_step = next_transition_step;
      sequence_complete = final_transition;
      s.set_target = 1;
    }
    count_or_one = s.transition_step_count > 1 ? s.transition_step_count : 1;
    normalised_progress = progress / count_or_one;
  }

  if (first) s.transition_step_count = s.transition_step;
  if (s.transition_step > kTransitionStepThreshold) s.failed = 1; // ROS_FATAL + ros::shutdown() in the reference (:436-440)

  int result;
  if (sequence_complete) {
    s.set_target = 1;
    s.vertical_transition_complete = 0;
    s.horizontal_transition_complete = 0;
    s.first_sequence_execution = 0;
    s.completed_sequence = sequence + 1;
    result = 100;
  } else {
    total_progress = total_progress + normalised_progress;
    if (total_progress > 99) total_progress = 99;
    result = s.first_sequence_execution ? -1 : total_progress;
  }
  if (s.failed) result = -2; // more than TRANSITION_STEP_THRESHOLD transitions: the reference shuts the controller down
  if (progress_out) progress_out[rob] = result;
}

// PoseController::stepToNewStance (:521-557): the two leg groups step to the (new) default tip poses one after the other.
template <int L, int NJ>
__global__ void step_to_new_stance_kernel(DevState st, const SharedConsts<L, NJ> *gc, SeqRobotState *seq, SeqParams P, int32_t *progress_out) {
  using FD = Fields<NJ>;
  const int64_t rob = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (rob >= st.n_robots) return;
  SeqRobotState &s = seq[rob];
  seq_defaults(s);
  const Pose current_pose = robot_current_pose<L>(st, rob);
  const Quat target_rotation{P.target_rotation[0], P.target_rotation[1], P.target_rotation[2], P.target_rotation[3]};
  int progress = 0;
  if (!P.posed)
    for (int l = 0; l < L; ++l) admittance_prologue_dev<NJ>(LegIO<NJ>{st, slot_of(rob, l, L)}, gc->leg[l], gc->P); // posing part of the loop
  for (int l = 0; l < L; ++l) {
    if ((l % 2) != s.current_group) continue;
    const LegIO<NJ> io{st, slot_of(rob, l, L)};
    double target[7];
    put_pose7(target, io.get3(FD::DFLT), leg_target_rotation<NJ>(gc->leg[l], target_rotation)); // leg_stepper->getDefaultTipPose()
    Pose tip;
    progress = step_to_position_dev<NJ>(st, io, gc->leg[l], target, current_pose, P.swing_height, 1.0 / P.step_frequency, 1, P.have_adm, P.dt, tip, leg_state_of(st, rob, l));
    put_pose7(s.leg[l].current, tip);
    set_desired_dev<NJ>(st, io, L, rob, s.leg[l].current, 1, P.have_adm, P.gravity_aligned);
    apply_ik_dev<NJ>(st, io, gc->leg[l], 0, P.dt, P.clamp_vel, P.clamp_pos, P.tip_force, P.force_gain);
    s.legs_completed_step += (progress == 100);
  }
  progress = progress / 2 + s.current_group * 50;
  s.current_group = s.legs_completed_step / (L / 2);
  if (s.legs_completed_step == L) {
    s.legs_completed_step = 0;
    s.current_group = 0;
  }
  s.reset_transition_sequence = 1; // a new stance needs a new start-up sequence
  if (progress_out) progress_out[rob] = progress;
}

// ---------------------------------------------------------------------------------------------------- planner mode
// One StateController::loop() in planner mode for every robot that stands (state_controller.cpp:401-405 -> executePlan :653-698):
//   nothing acquired        -> the node republishes its request for plan step plan_step_; Model::updateModel (:666)          result -2
//   configuration acquired  -> PoseController::transitionConfiguration(5.0) (pose_controller.cpp:710-763)                     0 .. 100
//   tip / body pose acquired-> PoseController::transitionStance(5.0) (:767-807)                                               0 .. 100
// A robot that is still walking gets its velocity inputs zeroed (:691-697) and result -1; its loop is the normal control cycle,
// which the host launches next with this kernel's skip marks (the robots handled here are left alone by it).
constexpr double kPlanTransitionTime = 5.0;

template <int L, int NJ>
__global__ void execute_plan_kernel(DevState st, const SharedConsts<L, NJ> *gc, SeqRobotState *seq, SeqParams P, int reset_poser_tips, int32_t *progress_out,
                                    int32_t *plan_step_out, int32_t *walking_out, int phase) {
  using FD = Fields<NJ>;
  using R = RobotFields;
  using X = ExtFields;
  const int64_t rob = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (rob >= st.n_robots) return;
  constexpr int rpw = 64 / L;
  const int walk_state = st.robi[rob_index(rob, R::I_WORD, rpw, R::I_COUNT)] & 3;
  if (phase == LOOP_MARK) { // which robots run their loop here: those that stand (their posing part then runs in the cycle kernel, RT_POSE_MARKED)
    const bool standing = walk_state == WS_STOPPED;
    // LegPoser::current_tip_pose_ is what the last PoseController::updateStance left - with a pose that has moved on since.  A robot
    // whose last loop was a control cycle (this call follows cycles, or the robot was still walking in the previous plan call) gets
    // it re-derived here from Model::current_pose_ of that cycle, BEFORE the pose pass overwrites it; from then on it is state.
    if (standing && !P.poser_tip_kept && (reset_poser_tips || st.manual[rob].skip_cycle == 0)) {
      const Pose pose_of_last_cycle = robot_current_pose<L>(st, rob);
      for (int l = 0; l < L; ++l) {
        const LegIO<NJ> io{st, slot_of(rob, l, L)};
        const int ls = leg_state_of(st, rob, l);
        const V3 walker_tip = io.get3(FD::TIP);
        io.put3(FD::POSER_TIP, (ls == LS_MANUAL || ls == LS_WALKING_TO_MANUAL) ? walker_tip : inverse_transform_vector(pose_of_last_cycle, walker_tip));
      }
    }
    st.manual[rob].skip_cycle = standing ? 1 : 0;
    return;
  }
  SeqRobotState &s = seq[rob];
  seq_defaults(s);
  if (reset_poser_tips) s.poser_tip_from_plan = 0; // control cycles ran since the last plan call: updateStance rewrote the poser tips
  int progress;
  if (walk_state != WS_STOPPED) {
    st.robd[rob_index(rob, R::VIN, rpw, R::COUNT)] = 0.0;
    st.robd[rob_index(rob, R::VIN + 1, rpw, R::COUNT)] = 0.0;
    st.robd[rob_index(rob, R::WIN, rpw, R::COUNT)] = 0.0;
    st.manual[rob].skip_cycle = 0;
    *walking_out = 1;
    progress = -1;
  } else {
    st.manual[rob].skip_cycle = 1;
    if (phase != LOOP_AFTER_POSE) // (... unless the cycle kernel has just run it)
      for (int l = 0; l < L; ++l) admittance_prologue_dev<NJ>(LegIO<NJ>{st, slot_of(rob, l, L)}, gc->leg[l], gc->P); // posing part of the loop
    if (!s.configuration_acquired && !s.tip_pose_acquired && !s.body_pose_acquired) {
      const Pose current_pose = robot_current_pose<L>(st, rob);
      for (int l = 0; l < L; ++l) { // Model::updateModel: setDesiredTipPose() = the poser's tip pose + admittance delta, applyIK
        const LegIO<NJ> io{st, slot_of(rob, l, L)};
        double tip[7];
        const int ls = leg_state_of(st, rob, l);
        if (s.poser_tip_from_plan && ls != LS_MANUAL) { // (stepToPosition leaves a MANUAL leg's LegPoser tip alone, :1680-1684)
          for (int k = 0; k < 7; ++k) tip[k] = s.leg[l].current[k];
        } else {
          const bool manual = ls == LS_MANUAL || ls == LS_WALKING_TO_MANUAL;
          // ... and its rotation with gravity-aligned tips: pose.rotation^-1 * the walker's tip rotation where that is defined (:129-130)
          Quat rotation{0, 0, 0, 0};
          if (P.gravity_aligned && (st.legi[io.slot] & LW_ROTDEF)) {
            const V3 walker_dir = io.get3(FD::CUR_DIR);
            rotation = from_two_vectors(V3{1, 0, 0}, manual ? walker_dir : rotate(inverse(current_pose.r), walker_dir));
          }
          if (P.pose_pass) // the LegPoser's tip of the last updateStance (kept as state: the pose has moved since)
            put_pose7(tip, io.get3(FD::POSER_TIP), rotation);
          else if (manual) // updateStance hands manually manipulated legs the stepper's tip as it is (:134-137)
            put_pose7(tip, io.get3(FD::TIP), rotation);
          else             // updateStance (pose_controller.cpp:122-131)
            put_pose7(tip, inverse_transform_vector(current_pose, io.get3(FD::TIP)), rotation);
        }
        set_desired_dev<NJ>(st, io, L, rob, tip, 1, P.have_adm, 0);
        apply_ik_dev<NJ>(st, io, gc->leg[l], 0, P.dt, P.clamp_vel, P.clamp_pos, P.tip_force, P.force_gain);
      }
      progress = -2;
    } else {
      progress = 0x7fffffff;
      if (s.configuration_acquired) {
        for (int l = 0; l < L; ++l) {
          SeqLegState &sl = s.leg[l];
          const LegIO<NJ> io{st, slot_of(rob, l, L)};
          if (!s.executing_transition) { // setDesiredConfiguration (:750-757)
            sl.plan_latched = sl.plan_named;
            for (int j = 0; j < NJ; ++j) sl.plan_desired[j] = sl.plan_configuration[j];
          }
          int p = 100; // a leg the message does not name has no desired configuration (:1479-1482)
          if (sl.plan_latched) p = transition_configuration_dev<NJ>(io, sl.plan_desired, kPlanTransitionTime, P.dt);
          progress = p < progress ? p : progress;
        }
        s.executing_transition = (progress != 0 && progress != 100);
      } else {
        const double *b = s.target_body_pose;
        const Pose body{V3{b[0], b[1], b[2]}, Quat{b[3], b[4], b[5], b[6]}};
        for (int l = 0; l < L; ++l) {
          const LegIO<NJ> io{st, slot_of(rob, l, L)};
          auto xf = [&](int field) -> double & { return st.ext[leg_field_index(field, io.slot, st.n_slots)]; };
          const bool defined = st.ext != nullptr && (int(xf(X::P_FLAGS)) & 1) != 0;
          double target[7] = {kUndefinedPosition, kUndefinedPosition, kUndefinedPosition, 0.0, 0.0, 0.0, 0.0}, clearance = 0.0; // Pose::Undefined()
          if (defined) { // target.transform_.addPose(target.pose_) (:781)
            const Pose tr{V3{xf(X::P_TRANSFORM), xf(X::P_TRANSFORM + 1), xf(X::P_TRANSFORM + 2)},
                          Quat{xf(X::P_TRANSFORM + 3), xf(X::P_TRANSFORM + 4), xf(X::P_TRANSFORM + 5), xf(X::P_TRANSFORM + 6)}};
            const Quat pr{xf(X::P_POSE + 3), xf(X::P_POSE + 4), xf(X::P_POSE + 5), xf(X::P_POSE + 6)};
            put_pose7(target, transform_vector(tr, V3{xf(X::P_POSE), xf(X::P_POSE + 1), xf(X::P_POSE + 2)}), tr.r * pr);
            clearance = xf(X::P_CLEARANCE);
          }
          // "Update target rotation if gravity alignment is set" (:786-790): FromTwoVectors(UnitX, Model::estimateGravity())
          bool gravity_rotation = false;
          if (P.gravity_aligned_tips && target[3] == 0.0 && target[4] == 0.0 && target[5] == 0.0 && target[6] == 0.0) {
            constexpr int rpw = 64 / L;
            const Quat imu{st.robd[rob_index(rob, R::IMUQ, rpw, R::COUNT)], st.robd[rob_index(rob, R::IMUQ + 1, rpw, R::COUNT)],
                           st.robd[rob_index(rob, R::IMUQ + 2, rpw, R::COUNT)], st.robd[rob_index(rob, R::IMUQ + 3, rpw, R::COUNT)]};
            const V3 e = quat_to_euler(imu, false); // Model::estimateGravity (model.cpp:156-165)
            V3 gv{0, 0, kGravity};
            gv = rotate(angle_axis_y(-e.y), gv);
            gv = rotate(angle_axis_x(-e.x), gv);
            const Quat r = from_two_vectors(V3{1, 0, 0}, gv);
            target[3] = r.w, target[4] = r.x, target[5] = r.y, target[6] = r.z;
            gravity_rotation = true;
          }
          Pose tip;
          const int p = step_to_position_dev<NJ>(st, io, gc->leg[l], (defined || gravity_rotation) ? target : nullptr, body, clearance, kPlanTransitionTime, 1, P.have_adm, P.dt, tip, leg_state_of(st, rob, l));
          put_pose7(s.leg[l].current, tip);
          set_desired_dev<NJ>(st, io, L, rob, s.leg[l].current, 1, P.have_adm, 0);
          apply_ik_dev<NJ>(st, io, gc->leg[l], 0, P.dt, P.clamp_vel, P.clamp_pos, P.tip_force, P.force_gain);
          progress = p < progress ? p : progress;
          if (defined && p == 100) xf(X::P_FLAGS) = double(int(xf(X::P_FLAGS)) & ~1); // target achieved (:801-805)
        }
        s.poser_tip_from_plan = 1;
      }
      if (progress == 100) {
        s.plan_step++;
        s.target_body_pose[0] = s.target_body_pose[1] = s.target_body_pose[2] = 0.0; // poser_->setTargetBodyPose(Pose::Identity())
        s.target_body_pose[3] = 1.0;
        s.target_body_pose[4] = s.target_body_pose[5] = s.target_body_pose[6] = 0.0;
        s.configuration_acquired = s.tip_pose_acquired = s.body_pose_acquired = 0;
      }
    }
  }
  if (progress_out) progress_out[rob] = progress;
  if (plan_step_out) plan_step_out[rob] = s.plan_step;
}

// plannerModeCallback / targetConfigurationCallback / targetBodyPoseCallback (state_controller.cpp:1262-1281, :1683-1702)
__global__ void plan_inputs_kernel(SeqRobotState *seq, int64_t first, int64_t count, int L, int NJ, int reset_plan_step, const double *configuration,
                                   const double *body_pose) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= count) return;
  SeqRobotState &s = seq[first + t];
  seq_defaults(s);
  if (reset_plan_step) s.plan_step = 0;
  if (configuration) {
    for (int l = 0; l < L; ++l) {
      const double *row = configuration + (t * L + l) * NJ;
      s.leg[l].plan_named = !isnan(row[0]);
      for (int j = 0; j < NJ; ++j) s.leg[l].plan_configuration[j] = row[j];
    }
    s.configuration_acquired = 1;
  }
  if (body_pose) {
    for (int k = 0; k < 7; ++k) s.target_body_pose[k] = body_pose[t * 7 + k];
    s.body_pose_acquired = 1;
  }
}

// ---------------------------------------------------------------------------------------------------- manual leg manipulation
// StateController::legStateToggle (state_controller.cpp:541-646) for the leg each instance's request designates (leg_selection[rob],
// -1 = no request), with PoseController::poseForLegManipulation (pose_controller.cpp:561-611) and AdmittanceController::
// updateStiffness(leg, scale) (admittance_controller.cpp:66-92).  result: 1 transition complete (the node clears its toggle
// flag), 0 in progress, 2 refused (MAX_MANUAL_LEGS), -1 the robot is still walking (the node keeps cycling with zero velocity
// inputs, :641-645), -3 no request.  Supported while the body pose is walk-plane pose + manual pose only (the reset to the
// default pose the toggle forces, :597, is then the walk-plane pose itself).
constexpr int kMaxManualLegs = 2; // MAX_MANUAL_LEGS (state_controller.h:26)

template <int L, int NJ>
__global__ void leg_state_toggle_kernel(DevState st, const SharedConsts<L, NJ> *gc, const int32_t *leg_selection, SeqParams P, double virtual_stiffness,
                                        double swing_stiffness_scaler, double load_stiffness_scaler, int dynamic_stiffness, int32_t *result_out,
                                        int32_t *cycle_out, int phase) {
  using FD = Fields<NJ>;
  using R = RobotFields;
  const int64_t rob = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (rob >= st.n_robots) return;
  const int sel = leg_selection[rob];
  int result = -3;
  st.manual[rob].skip_cycle = 0; // no request, or still walking: this robot's loop is the ordinary control cycle (launched next)
  if (phase == LOOP_MARK && !(sel >= 0 && sel < L)) return;
  if (sel >= 0 && sel < L) {
    ManualRobot &m = st.manual[rob];
    constexpr int rpw = 64 / L;
    const int walk_state = st.robi[rob_index(rob, R::I_WORD, rpw, R::I_COUNT)] & 3;
    if (walk_state == WS_STOPPED) { // the posing part of this loop (a robot that is still walking runs its whole loop in the cycle kernel)
      m.skip_cycle = 1;
      if (phase == LOOP_MARK) return;
      if (phase != LOOP_AFTER_POSE) // (... unless the cycle kernel has just run it: RT_POSE_MARKED)
        for (int l = 0; l < L; ++l) admittance_prologue_dev<NJ>(LegIO<NJ>{st, slot_of(rob, l, L)}, gc->leg[l], gc->P);
    }
    if (phase == LOOP_MARK) return;
    if (walk_state != WS_STOPPED) {
      result = -1; // "Stopping Syropod to transition leg state": the velocity inputs are forced to zero (:641-645)
      st.robd[rob_index(rob, R::VIN, rpw, R::COUNT)] = 0.0;
      st.robd[rob_index(rob, R::VIN + 1, rpw, R::COUNT)] = 0.0;
      st.robd[rob_index(rob, R::WIN, rpw, R::COUNT)] = 0.0;
    } else if (m.leg_state[sel] == LS_WALKING) {
      if (m.manual_leg_count < kMaxManualLegs) {
        m.leg_state[sel] = LS_WALKING_TO_MANUAL;
        int &w = st.legi[slot_of(rob, sel, L)]; // leg_stepper->setSwingProgress(-1) / setStanceProgress(-1) (:573-574)
        w = (w & ~(3 << LW_PM_SHIFT)) | (PM_NONE << LW_PM_SHIFT);
        result = 0;
      } else {
        result = 2;
      }
    } else if (m.leg_state[sel] == LS_MANUAL) {
      m.leg_state[sel] = LS_MANUAL_TO_WALKING;
      result = 0;
    } else {
      const bool to_manual = m.leg_state[sel] == LS_WALKING_TO_MANUAL;
      // poser_->setPoseResetMode(IMMEDIATE_ALL_RESET): the next pose update puts manual_pose_ on default_pose_ (identity: calculateDefaultPose
      // is never called), and with walk-plane + manual posing only Model::current_pose_ becomes the walk-plane pose of a standing robot
      // (the reset takes effect in the pose update of the NEXT loop: this call still sees the poses of the last update)
      auto rd = [&](int f) -> double & { return st.robd[rob_index(rob, f, rpw, R::COUNT)]; };
      Pose current_pose = robot_current_pose<L>(st, rob);
      const V3 manual_position{rd(R::MPOSE), rd(R::MPOSE + 1), rd(R::MPOSE + 2)};
      // ---- poseForLegManipulation
      int min_progress = 2147483647;
      for (int l = 0; l < L; ++l) {
        const LegIO<NJ> io{st, slot_of(rob, l, L)};
        const int ls = m.leg_state[l];
        double step_height = P.swing_height;
        const double step_time = 1.0 / P.step_frequency;
        Pose target_pose;
        if (ls == LS_WALKING_TO_MANUAL) {
          target_pose = pose_identity();
          if (P.inclination_posing) { // + inclination_pose_.position_ as this loop's updateInclinationPose left it (the cycle kernel's pose pass)
            target_pose.p.x += rd(R::INCL);
            target_pose.p.y += rd(R::INCL + 1);
          }
          target_pose.p.z -= step_height;
        } else {
          target_pose = current_pose;    // remove the manual pose, add the default pose (identity: calculateDefaultPose is never called)
          target_pose.p = target_pose.p - manual_position;
        }
        double target[7];
        put_pose7(target, inverse_transform_vector(target_pose, io.get3(FD::DFLT)), Quat{0, 0, 0, 0});
        if (ls == LS_WALKING_TO_MANUAL) {
          io.put3(FD::TIP, V3{target[0], target[1], target[2]}); // leg_stepper->setCurrentTipPose(target_tip_pose): rotation undefined
          st.legi[io.slot] &= ~LW_ROTDEF;
          step_height = 0.0;
        } else if (ls == LS_MANUAL_TO_WALKING) {
          io.put3(FD::TIP, io.get3(FD::DFLT));                   // leg_stepper->setCurrentTipPose(default tip pose)
          // ... whose rotation is undefined once updateDefaultTipPosition has re-derived the pose (every stop does, walk_controller.cpp:1003).
          // (A robot that has never walked still holds the gravity-aligned identity rotation there; nothing reads the flag before the
          //  next updateTipRotation recomputes it, so only a snapshot taken in between could tell.)
          st.legi[io.slot] &= ~LW_ROTDEF;
        }
        Pose tip;
        const int progress = step_to_position_dev<NJ>(st, io, gc->leg[l], target, pose_identity(), step_height, step_time, 1, P.have_adm, P.dt, tip, ls);
        min_progress = progress < min_progress ? progress : min_progress;
        if (progress != 100) {
          double cur[7];
          put_pose7(cur, tip);
          set_desired_dev<NJ>(st, io, L, rob, cur, 1, P.have_adm, P.gravity_aligned);
          apply_ik_dev<NJ>(st, io, gc->leg[l], 0, P.dt, P.clamp_vel, P.clamp_pos, P.tip_force, P.force_gain);
        }
      }
      if (dynamic_stiffness) { // admittance_->updateStiffness(leg, scale_reference) (:601-605, :622-626)
        double scale = double(min_progress) / 100.0;
        if (!to_manual) scale = fabs(scale - 1.0);
        const double swing = virtual_stiffness * (scale * (swing_stiffness_scaler - 1) + 1), load = virtual_stiffness * (scale * (load_stiffness_scaler - 1) + 1);
        auto stiff = [&](int l) -> double & { return st.legd[leg_field_index(FD::ADM_DELTA + 3, slot_of(rob, l, L), st.n_slots)]; };
        const int a1 = (sel + L - 1) % L, a2 = (sel + 1) % L;
        stiff(sel) = swing;
        if (m.leg_state[a1] != LS_MANUAL) stiff(a1) = load;
        if (m.leg_state[a2] != LS_MANUAL) stiff(a2) = load;
      }
      if (phase != LOOP_AFTER_POSE) { // (with time-dependent posing the next loop's pose pass applies the reset itself)
        for (int k = 0; k < 7; ++k) rd(R::MPOSE + k) = (k == 3) ? 1.0 : 0.0; // the next pose update: manual_pose_ = default_pose_ ...
        for (int k = 0; k < 7; ++k) rd(R::CPOSE + k) = rd(R::OWPP + k);      // ... and current_pose_ = the standing walk-plane pose
      }
      st.robi[rob_index(rob, R::I_RESET_MODE, rpw, R::I_COUNT)] = 5; // IMMEDIATE_ALL_RESET while the transition runs
      result = 0;
      if (min_progress == 100) {
        m.leg_state[sel] = to_manual ? LS_MANUAL : LS_WALKING;
        m.manual_leg_count += to_manual ? 1 : -1;
        st.robi[rob_index(rob, R::I_RESET_MODE, rpw, R::I_COUNT)] = 0; // NO_RESET
        result = 1;
      }
    }
  }
  if (result == -1 || result == -3) *cycle_out = 1;
  if (result_out) result_out[rob] = result;
}

// primary / secondary leg selection and tip inputs of every instance (primaryLegSelectionCallback ... :1247-1330)
__global__ void set_manual_inputs_kernel(ManualRobot *manual, int64_t n, const int32_t *primary_leg, const double *primary_velocity,
                                         const double *primary_position, const int32_t *secondary_leg, const double *secondary_velocity,
                                         const double *secondary_position) {
  const int64_t rob = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (rob >= n) return;
  ManualRobot &m = manual[rob];
  m.primary_leg = primary_leg ? primary_leg[rob] : -1;
  m.secondary_leg = secondary_leg ? secondary_leg[rob] : -1;
  for (int k = 0; k < 3; ++k) {
    m.primary_velocity[k] = primary_velocity ? primary_velocity[rob * 3 + k] : 0.0;
    m.primary_position[k] = primary_position ? primary_position[rob * 3 + k] : 0.0;
    m.secondary_velocity[k] = secondary_velocity ? secondary_velocity[rob * 3 + k] : 0.0;
    m.secondary_position[k] = secondary_position ? secondary_position[rob * 3 + k] : 0.0;
  }
}
__global__ void get_leg_manipulation_state_kernel(const ManualRobot *manual, int64_t n, int L, int32_t *out) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n * L) return;
  out[t] = manual ? manual[t / L].leg_state[t % L] : LS_WALKING;
}

} // namespace shc

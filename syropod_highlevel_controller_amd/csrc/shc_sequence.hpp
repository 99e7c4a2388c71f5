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
};
struct SeqRobotState {                       // class PoseController (pose_controller.h:273-274, :296-304)
  int32_t legs_completed_step, current_group, transition_step, transition_step_count;
  int32_t set_target, proximity_alert, horizontal_transition_complete, vertical_transition_complete;
  int32_t first_sequence_execution, reset_transition_sequence, failed, initialised;
  int32_t completed_sequence, pad_; // 1 + the sequence this robot has completed and not left since (see execute_sequence_kernel)
  SeqLegState leg[SHC_MAX_LEGS];
};

struct SeqParams {
  double step_frequency, swing_height, dt, force_gain;
  double target_rotation[4]; // LegStepper::target_tip_pose_.rotation_ while the robot has not walked: the identity tip rotation
                             // (walk_controller.cpp:37-41), UNDEFINED_ROTATION (zeros) without gravity-aligned tips
  int clamp_vel, clamp_pos, tip_force, have_adm, gravity_aligned;
};

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
  s.initialised = 1;
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
        put_pose7(s.leg[l].target, target, target_rotation);
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
        progress = step_to_position_dev<NJ>(st, io, gc->leg[l], g.target, identity, step_height, time_to_step, apply_delta, P.have_adm, P.dt, tip);
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
        put_pose7(s.leg[l].target, target, target_rotation);
      }
    }
    bool all_legs_within_workspace = true;
    for (int l = 0; l < L; ++l) {
      SeqLegState &g = s.leg[l];
      const LegIO<NJ> io{st, slot_of(rob, l, L)};
      double time_to_step = kVerticalTransitionTime / P.step_frequency;
      time_to_step *= first ? 2.0 : 1.0;
      Pose tip;
      progress = step_to_position_dev<NJ>(st, io, gc->leg[l], g.target, identity, 0.0, time_to_step, apply_delta, P.have_adm, P.dt, tip);
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
  for (int l = 0; l < L; ++l) {
    if ((l % 2) != s.current_group) continue;
    const LegIO<NJ> io{st, slot_of(rob, l, L)};
    double target[7];
    put_pose7(target, io.get3(FD::DFLT), target_rotation); // leg_stepper->getDefaultTipPose()
    Pose tip;
    progress = step_to_position_dev<NJ>(st, io, gc->leg[l], target, current_pose, P.swing_height, 1.0 / P.step_frequency, 1, P.have_adm, P.dt, tip);
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

} // namespace shc

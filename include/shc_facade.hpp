// shc_facade.hpp — the reference's per-cycle C++ call surface over the batched engine (header only).
//
// The reference node drives ONE robot through these calls every cycle (src/state_controller.cpp:162-193, 379-447):
//     poser_->updateCurrentPose(robot_state_);                 pose_controller.h:216
//     admittance_->updateStiffness(walker_); updateAdmittance();   admittance_controller.h:42,61
//     walker_->updateWalk(linear_velocity_input_, angular_velocity_input_);   walk_controller.h:204
//     poser_->updateStance();                                  pose_controller.h:157
//     model_->updateModel();                                   model.h:163
// and then reads Joint::desired_position_ / desired_velocity_ (state_controller.cpp:777-805) and the LegState fields
// (:809-893); its cold paths call the per-leg methods Leg::setDesiredTipPose / solveIK / updateJointPositions / applyIK /
// applyFK (model.h:448-492) directly.  This façade keeps those class names, method names, argument order and return
// values so the node's code compiles against it unchanged in shape: the five per-cycle calls record their inputs and the
// LAST one of the cycle (Model::updateModel) launches the fused HIP cycle kernel for a batch of one (a larger batch is
// driven through the C ABI arrays; its instance `index` can still be read through these classes).
//
// Types: the reference's signatures use Eigen (Vector2d / Vector3d / VectorXd / MatrixXd / Quaterniond) and its own Pose;
// neither Eigen nor ROS is a dependency of this repository, so vector arguments are templated on "anything indexable"
// (Eigen objects qualify) and the header ships tiny PODs.  See INTEGRATION.md.
//
// Read-back is lazy: a cycle only marks the cached outputs stale; the first getter of a group (joints, leg state, body
// state, odometry, stiffness) fetches that group once.
#pragma once

#include "shc_batch.h"

#include <array>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace shc_facade {

// parameters_and_states.h:25
enum RobotState { PACKED, READY, RUNNING, ROBOT_STATE_COUNT, UNKNOWN = -1, OFF = -2 };

struct Vector3 {
  double v[3];
  double &operator[](int i) { return v[i]; }
  const double &operator[](int i) const { return v[i]; }
};
struct Quaternion {
  double w, x, y, z;
};
struct Pose { // include/syropod_highlevel_controller/pose.h:17
  Vector3 position_;
  Quaternion rotation_;
  // pose.h:206: UNDEFINED_POSITION (UNASSIGNED_VALUE x 3, standard_includes.h:52,55) + UNDEFINED_ROTATION (0, 0, 0, 0)
  static Pose Undefined() { return Pose{{{2147483647.0, 2147483647.0, 2147483647.0}}, {0, 0, 0, 0}}; }
  bool isUndefined() const {
    return position_[0] == 2147483647.0 && position_[1] == 2147483647.0 && position_[2] == 2147483647.0 && rotation_.w == 0 &&
           rotation_.x == 0 && rotation_.y == 0 && rotation_.z == 0;
  }
};

inline void check(int rc, const char *what) {
  if (rc != SHC_OK) throw std::runtime_error(std::string(what) + ": " + shc_last_error());
}

// Owns the engine (batch of `n`, default one robot) and the lazily refreshed copies of what the node reads.
class Engine {
public:
  explicit Engine(const shc_params &params, int64_t n = 1, int device = 0, void *stream = nullptr) : params_(params), n_(n) {
    check(shc_engine_create(&params_, n, device, stream, &e_), "shc_engine_create");
    measured_q_.assign(size_t(n) * params_.leg_count * dof(), 0.0);
  }
  ~Engine() { shc_engine_destroy(e_); }
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;

  shc_engine *handle() { return e_; }
  const shc_params &params() const { return params_; }
  int64_t instances() const { return n_; }
  int legs() const { return params_.leg_count; }
  int dof() const { // joint arrays of the C ABI are [legs][DOF of the robot's longest leg]
    int nj = 0;
    for (int l = 0; l < params_.leg_count; ++l) nj = params_.leg_dof[l] > nj ? params_.leg_dof[l] : nj;
    return nj;
  }
  // One control cycle for the whole batch (asynchronous; the next getter synchronises through its copy).
  void cycle(int n_cycles = 1) {
    check(shc_engine_step(e_, n_cycles), "shc_engine_step");
    invalidate();
  }
  void invalidate() { have_ = 0; }
  // state.init() + state.initModel(false) (main.cpp:99-100): fresh controllers, joints where the motors report them (NULL =
  // the READY / unpacked configuration) - the starting point of PoseController::executeSequence(START_UP)
  void initModel(const double *joint_positions = nullptr) {
    check(shc_engine_begin_sequence_startup(e_, joint_positions, 0), "shc_engine_begin_sequence_startup");
    invalidate();
  }
  // what transitionRobotState does once executeSequence(START_UP) has completed (state_controller.cpp:305-313)
  void finishStartUpSequence() {
    check(shc_engine_finish_sequence_startup(e_), "shc_engine_finish_sequence_startup");
    invalidate();
  }
  // StateController::legStateToggle (state_controller.cpp:541-646) for leg `leg_id`, one call per loop while the node's toggle flag is
  // set: 1 = transition complete, 0 = in progress, 2 = refused (MAX_MANUAL_LEGS), -1 = the robot is still walking (keep cycling
  // with zero velocity inputs)
  int legStateToggle(int leg_id) {
    if (n_ != 1) throw std::runtime_error("shc_facade: legStateToggle takes one leg of one robot: use shc_engine_toggle_leg_state for a batch");
    const int32_t sel = leg_id;
    int32_t result = 0;
    check(shc_engine_toggle_leg_state(e_, &sel, &result), "shc_engine_toggle_leg_state");
    manual_legs_ = true;
    invalidate();
    return result;
  }
  bool manual_legs_ = false;
  // plannerModeCallback (state_controller.cpp:1262-1281) and StateController::executePlan (:653-698), one call per loop while planner
  // mode is on: 0 .. 100 = progress of the plan step being executed (PoseController::transitionConfiguration / transitionStance with
  // the 5 s the node gives them), SHC_PLAN_WAITING = nothing acquired (republish the request for planStep()), SHC_PLAN_WALKING =
  // the robot was still walking: velocity inputs zeroed, this loop was an ordinary control cycle
  void setPlannerMode(bool on) { check(shc_engine_set_planner_mode(e_, on ? 1 : 0), "shc_engine_set_planner_mode"); }
  int executePlan() {
    if (n_ != 1) throw std::runtime_error("shc_facade: executePlan runs one robot's loop: use shc_engine_execute_plan for a batch");
    int32_t progress = 0;
    check(shc_engine_execute_plan(e_, &progress, &plan_step_), "shc_engine_execute_plan");
    manual_legs_ = true; // (the planner shares the per-robot records of the loop-level kernels)
    invalidate();
    return progress;
  }
  int planStep() const { return plan_step_; } // StateController::plan_step_
  int32_t plan_step_ = 0;
  // StateController::changeGait (state_controller.cpp:513): true once the gait has changed, false while the robots are
  // still being stopped (keep cycling and call again, as the reference does while gait_change_flag_ is set)
  bool changeGait(const shc_params &new_gait) {
    int64_t still = 0;
    check(shc_engine_change_gait(e_, &new_gait, &still), "shc_engine_change_gait");
    invalidate();
    return still == 0;
  }

  // StateController::adjustParameter (state_controller.cpp:451-509), as runningState calls it while parameter_adjust_flag_ is set (:411-414): true once the
  // new value is set (the flag drops), false while step_frequency waits for the robots to slow down to the new limits (keep cycling and call again).
  // which = SHC_PARAM_* (enum ParameterSelection); the callbacks' clamping to the parameter's min / max stays with the node.
  bool adjustParameter(int which, double new_parameter_value) {
    int64_t pending = 0;
    check(shc_engine_adjust_parameter(e_, which, new_parameter_value, &pending), "shc_engine_adjust_parameter");
    invalidate();
    return pending == 0;
  }

  // ---- cached outputs, one fetch per group and cycle
  const std::vector<double> &q() { return joints_(), q_; }
  const std::vector<double> &qd() { return joints_(), qd_; }
  const std::vector<double> &walker_tip() { return leg_state_(), walker_tip_; }
  const std::vector<double> &poser_tip() { return leg_state_(), poser_tip_; }
  const std::vector<double> &model_tip() { return leg_state_(), model_tip_; }
  const std::vector<double> &tip_force() { return leg_state_(), tip_force_; }
  const std::vector<double> &admittance() { return leg_state_(), admittance_; }
  const std::vector<int32_t> &leg_status() { return leg_state_(), leg_status_; }
  const std::vector<double> &pose() { return body_(), pose_; }
  const std::vector<double> &velocity() { return body_(), velocity_; }
  const std::vector<int32_t> &walk_state() { return body_(), walk_state_; }
  const std::vector<double> &odometry() {
    if (!(have_ & 8)) {
      odometry_.resize(size_t(n_) * 7);
      check(shc_engine_get_odometry(e_, odometry_.data(), 0), "get_odometry");
      have_ |= 8;
    }
    return odometry_;
  }
  const std::vector<double> &stiffness() {
    if (!(have_ & 16)) {
      stiffness_.assign(size_t(n_) * legs(), 0.0);
      if (params_.admittance_control) check(shc_engine_get_virtual_stiffness(e_, stiffness_.data(), 0), "get_virtual_stiffness");
      have_ |= 16;
    }
    return stiffness_;
  }
  // Joint::current_position_ as jointStatesCallback stores it (state_controller.cpp:1566-1594): only Leg::applyFK(.., use_actual)
  std::vector<double> measured_q_;

private:
  void joints_() {
    if (have_ & 1) return;
    q_.resize(size_t(n_) * legs() * dof());
    qd_.resize(q_.size());
    check(shc_engine_get_joint_state(e_, q_.data(), qd_.data(), 0), "get_joint_state");
    have_ |= 1;
  }
  void leg_state_() {
    if (have_ & 2) return;
    const size_t n3 = size_t(n_) * legs() * 3;
    walker_tip_.resize(n3), poser_tip_.resize(n3), model_tip_.resize(n3), tip_force_.resize(n3), admittance_.resize(n3);
    leg_status_.resize(size_t(n_) * legs());
    check(shc_engine_get_leg_state(e_, walker_tip_.data(), poser_tip_.data(), model_tip_.data(), tip_force_.data(), admittance_.data(),
                                   leg_status_.data(), 0),
          "get_leg_state");
    have_ |= 2;
  }
  void body_() {
    if (have_ & 4) return;
    pose_.resize(size_t(n_) * 7), velocity_.resize(size_t(n_) * 3), walk_state_.resize(size_t(n_));
    check(shc_engine_get_body_state(e_, pose_.data(), velocity_.data(), walk_state_.data(), 0), "get_body_state");
    have_ |= 4;
  }
  shc_params params_;
  int64_t n_;
  shc_engine *e_ = nullptr;
  unsigned have_ = 0;
  std::vector<double> q_, qd_, walker_tip_, poser_tip_, model_tip_, tip_force_, admittance_, pose_, velocity_, odometry_, stiffness_;
  std::vector<int32_t> leg_status_, walk_state_;
};

// The per-robot setters below take ONE robot's values, as the reference's callbacks do; the C ABI copies n instances' worth
// from the pointer it is given, so they are only valid on a batch of one (larger batches set inputs through the C ABI arrays).
inline void require_single(const Engine &e, const char *what) {
  if (e.instances() != 1) throw std::logic_error(std::string(what) + ": per-robot setters need a batch of one; use the C ABI arrays");
}

// class Joint (model.h:558): the outputs the node publishes.
struct Joint {
  double desired_position_ = 0.0; // state_controller.cpp:786
  double desired_velocity_ = 0.0; // :787
  double offset_ = 0.0;           // :798 (added when publishing the per-joint command)
};

class LegStepper;
// class Leg (model.h:202): leg `id` of instance `index`.
class Leg {
public:
  Leg(Engine &eng, int64_t index, int id) : eng_(eng), index_(index), id_(id) {}
  int getIDNumber() const { return id_; }
  int getJointCount() const { return eng_.params().leg_dof[id_]; }
  Joint getJointByIDNumber(int joint_id /* 1-based, model.h:294 */) const {
    const size_t k = (size_t(index_) * eng_.legs() + id_) * eng_.dof() + (joint_id - 1);
    Joint j;
    j.desired_position_ = eng_.q()[k];
    j.desired_velocity_ = eng_.qd()[k];
    j.offset_ = eng_.params().joint[id_][joint_id - 1].offset;
    return j;
  }
  LegStepper getLegStepper() const;                                         // model.h:288
  Vector3 getCurrentTipPosition() const { return v3(eng_.model_tip()); }    // Leg::getCurrentTipPose().position_ (model.h:304)
  Vector3 getDesiredTipPosition() const { return v3(eng_.poser_tip()); }    // LegPoser::getCurrentTipPose() (pose_controller.h:449)
  Vector3 getWalkerTipPosition() const { return v3(eng_.walker_tip()); }    // LegStepper::getCurrentTipPose() (walk_controller.h:389)
  Vector3 getTipForceCalculated() const { return v3(eng_.tip_force()); }    // model.h:244
  Vector3 getAdmittanceDelta() const { return v3(eng_.admittance()); }      // model.h:256
  double getVirtualStiffness() const { return eng_.stiffness()[slot()]; }   // model.h:264
  int getStepState() const { return eng_.leg_status()[slot()] & 3; }        // walk_controller.h:325
  int getPhase() const { return eng_.leg_status()[slot()] >> 8; }           // :317
  bool ikFailed() const { return (eng_.leg_status()[slot()] & 4) != 0; }    // model.cpp:921

  // ---- the per-leg methods the reference's cold paths call (model.h:448-492), same argument order and return values
  // void setDesiredTipPose(const Pose& tip_pose = Pose::Undefined(), bool apply_delta = true)          model.h:448
  void setDesiredTipPose(const Pose &tip_pose = Pose::Undefined(), bool apply_delta = true) {
    const double p[7] = {tip_pose.position_[0], tip_pose.position_[1], tip_pose.position_[2], tip_pose.rotation_.w, tip_pose.rotation_.x,
                         tip_pose.rotation_.y,  tip_pose.rotation_.z};
    check(shc_leg_set_desired_tip_pose(eng_.handle(), index_, 1, id_, tip_pose.isUndefined() ? nullptr : p, apply_delta ? 1 : 0, 0),
          "shc_leg_set_desired_tip_pose");
  }
  // Eigen::VectorXd solveIK(const Eigen::MatrixXd& delta, const bool& solve_rotation)                  model.h:470
  template <class Delta6>
  std::vector<double> solveIK(const Delta6 &delta, const bool &solve_rotation) {
    const double d[6] = {delta[0], delta[1], delta[2], delta[3], delta[4], delta[5]};
    std::vector<double> out(size_t(eng_.dof()));
    check(shc_leg_solve_ik(eng_.handle(), index_, 1, id_, d, solve_rotation ? 1 : 0, out.data(), 0), "shc_leg_solve_ik");
    return out;
  }
  // double updateJointPositions(const Eigen::VectorXd& delta, const bool& simulation)                  model.h:477
  template <class JointDelta>
  double updateJointPositions(const JointDelta &delta, const bool &simulation) {
    std::vector<double> d(size_t(eng_.dof()));
    for (size_t j = 0; j < d.size(); ++j) d[j] = delta[j];
    double prox = 0.0;
    check(shc_leg_update_joint_positions(eng_.handle(), index_, 1, id_, d.data(), simulation ? 1 : 0, &prox, 0), "shc_leg_update_joint_positions");
    eng_.invalidate();
    return prox;
  }
  // double applyIK(const bool& simulation = false)                                                     model.h:485
  double applyIK(const bool &simulation = false) {
    double result = 0.0;
    check(shc_leg_apply_ik(eng_.handle(), index_, 1, id_, simulation ? 1 : 0, &result, 0), "shc_leg_apply_ik");
    eng_.invalidate();
    return result;
  }
  // Pose applyFK(const bool& set_current = true, const bool& use_actual = false)                        model.h:492
  // (set_current only refreshes cached members in the reference; the engine derives the model tip from the joints on demand)
  Pose applyFK(const bool & /*set_current*/ = true, const bool &use_actual = false) {
    double p[7];
    const double *measured = use_actual ? &eng_.measured_q_[(size_t(index_) * eng_.legs() + id_) * eng_.dof()] : nullptr;
    check(shc_leg_apply_fk(eng_.handle(), index_, 1, id_, measured, p, 0), "shc_leg_apply_fk");
    return Pose{{{p[0], p[1], p[2]}}, {p[3], p[4], p[5], p[6]}};
  }

private:
  size_t slot() const { return size_t(index_) * eng_.legs() + id_; }
  Vector3 v3(const std::vector<double> &a) const { return Vector3{{a[slot() * 3], a[slot() * 3 + 1], a[slot() * 3 + 2]}}; }
  Engine &eng_;
  int64_t index_;
  int id_;
};

enum SequenceSelection { START_UP, SHUT_DOWN, SEQUENCE_UNKNOWN }; // parameters_and_states.h:183-188

// struct ExternalTarget (walk_controller.h:38-46); frame_id_ is reduced to what the stepper asks of it (== "odom_ideal").
struct ExternalTarget {
  Pose pose_;
  double swing_clearance_ = 0.0;
  bool frame_is_odom_ideal_ = false;
  Pose transform_ = Pose{{{0.0, 0.0, 0.0}}, {1.0, 0.0, 0.0, 0.0}};
  bool defined_ = false;
};

// class LegStepper (walk_controller.h:304): the external target / default interface of leg `id` of instance `index`
// (rough terrain mode; targetTipPoseCallback and generateExternalTargetTransforms, state_controller.cpp:1706-1767, :703-773).
class LegStepper {
public:
  LegStepper(Engine &eng, int64_t index, int id) : eng_(eng), index_(index), id_(id) {}
  void setExternalTarget(const ExternalTarget &t) { set(SHC_EXTERNAL_TARGET, t); }   // walk_controller.h:437
  void setExternalDefault(const ExternalTarget &t) { set(SHC_EXTERNAL_DEFAULT, t); } // :441
  ExternalTarget getExternalTarget() const { return get(SHC_EXTERNAL_TARGET); }      // :385
  ExternalTarget getExternalDefault() const { return get(SHC_EXTERNAL_DEFAULT); }    // :389
  // LegPoser::setExternalTarget / getExternalTarget (pose_controller.h:489, :457): the planner-mode tip target, which
  // targetTipPoseCallback hands to the LegPoser of a robot that stands (state_controller.cpp:1738-1742)
  void setPoserExternalTarget(const ExternalTarget &t) { set(SHC_EXTERNAL_PLANNER_TARGET, t); }
  ExternalTarget getPoserExternalTarget() const { return get(SHC_EXTERNAL_PLANNER_TARGET); }

private:
  static void pack(const Pose &p, double *o) {
    o[0] = p.position_[0], o[1] = p.position_[1], o[2] = p.position_[2];
    o[3] = p.rotation_.w, o[4] = p.rotation_.x, o[5] = p.rotation_.y, o[6] = p.rotation_.z;
  }
  void set(int which, const ExternalTarget &t) {
    shc_external_target r{};
    pack(t.pose_, r.pose);
    pack(t.transform_, r.transform);
    r.swing_clearance = t.swing_clearance_;
    r.frame_is_odom_ideal = t.frame_is_odom_ideal_ ? 1 : 0;
    r.defined = t.defined_ ? 1 : 0;
    check(shc_engine_set_external_target(eng_.handle(), which, index_, 1, id_, &r, nullptr), "shc_engine_set_external_target");
  }
  ExternalTarget get(int which) const {
    shc_external_target r{};
    check(shc_engine_get_external_target(eng_.handle(), which, index_, 1, id_, &r), "shc_engine_get_external_target");
    ExternalTarget t;
    t.pose_ = Pose{{{r.pose[0], r.pose[1], r.pose[2]}}, {r.pose[3], r.pose[4], r.pose[5], r.pose[6]}};
    t.transform_ = Pose{{{r.transform[0], r.transform[1], r.transform[2]}}, {r.transform[3], r.transform[4], r.transform[5], r.transform[6]}};
    t.swing_clearance_ = r.swing_clearance;
    t.frame_is_odom_ideal_ = r.frame_is_odom_ideal != 0;
    t.defined_ = r.defined != 0;
    return t;
  }
  Engine &eng_;
  int64_t index_;
  int id_;
};

inline LegStepper Leg::getLegStepper() const { return LegStepper(eng_, index_, id_); }

// class Model (model.h:57)
class Model {
public:
  Model(std::shared_ptr<Engine> eng, int64_t index = 0) : eng_(std::move(eng)), index_(index) {}
  int getLegCount() const { return eng_->legs(); }                                // model.h:80
  Leg getLegByIDNumber(int leg_id) { return Leg(*eng_, index_, leg_id); }         // model.h:124
  Pose getCurrentPose() const {                                                   // model.h:84
    const double *p = &eng_->pose()[size_t(index_) * 7];
    return Pose{{{p[0], p[1], p[2]}}, {p[3], p[4], p[5], p[6]}};
  }
  // Model::setImuData (model.h:146): orientation (w,x,y,z), angular velocity
  template <class Q, class V>
  void setImuData(const Q &orientation_wxyz, const V & /*linear_acceleration*/, const V &angular_velocity) {
    double q[4] = {orientation_wxyz[0], orientation_wxyz[1], orientation_wxyz[2], orientation_wxyz[3]};
    double g[3] = {angular_velocity[0], angular_velocity[1], angular_velocity[2]};
    require_single(*eng_, "setImuData");
    check(shc_engine_set_imu(eng_->handle(), q, g, 0), "set_imu");
  }
  // jointStatesCallback (state_controller.cpp:1566-1594): measured positions (Leg::applyFK(.., use_actual)) and efforts
  void setJointState(int leg_id, int joint_id /* 1-based */, double position, double /*velocity*/, double effort) {
    const size_t k = (size_t(index_) * eng_->legs() + leg_id) * eng_->dof() + (joint_id - 1);
    eng_->measured_q_[k] = position;
    if (efforts_.empty()) efforts_.assign(size_t(eng_->legs()) * eng_->dof(), 0.0);
    efforts_[size_t(leg_id) * eng_->dof() + (joint_id - 1)] = effort;
    efforts_dirty_ = true;
  }
  // Model::updateModel (model.h:163, src/model.cpp:142): the last call of the reference's cycle -> launch the fused kernel.
  void updateModel() {
    if (efforts_dirty_) {
      require_single(*eng_, "setJointState");
      check(shc_engine_set_joint_effort(eng_->handle(), efforts_.data(), 0), "set_joint_effort");
      efforts_dirty_ = false;
    }
    eng_->cycle(1);
  }
  Engine &engine() { return *eng_; }

private:
  std::shared_ptr<Engine> eng_;
  int64_t index_;
  std::vector<double> efforts_;
  bool efforts_dirty_ = false;
};

// class WalkController (walk_controller.h:54)
class WalkController {
public:
  explicit WalkController(std::shared_ptr<Engine> eng) : eng_(std::move(eng)) {}
  // void updateWalk(const Eigen::Vector2d& linear_velocity_input, const double& angular_velocity_input)   walk_controller.h:204
  template <class V2>
  void updateWalk(const V2 &linear_velocity_input, const double &angular_velocity_input) {
    double lin[2] = {linear_velocity_input[0], linear_velocity_input[1]};
    double ang = angular_velocity_input;
    require_single(*eng_, "updateWalk");
    check(shc_engine_set_velocity(eng_->handle(), lin, &ang, 0), "set_velocity");
  }
  // void updateManual(primary_leg_selection_ID, primary_tip_velocity_input, secondary_leg_selection_ID, secondary_tip_velocity_input)
  //                                                                                                     walk_controller.h:213
  // The inputs are latched for the next control cycle, whose fused kernel moves the tips of MANUAL legs (walk_controller.cpp:652-708).
  void updateManual(const int &primary_leg_selection_ID, const Vector3 &primary_tip_velocity_input, const int &secondary_leg_selection_ID,
                    const Vector3 &secondary_tip_velocity_input) {
    require_single(*eng_, "updateManual");
    primary_ = primary_leg_selection_ID, secondary_ = secondary_leg_selection_ID;
    pvel_ = primary_tip_velocity_input, svel_ = secondary_tip_velocity_input;
    push_manual();
  }
  // void updateManual(primary_leg_selection_ID, primary_tip_pose_input, secondary_leg_selection_ID, secondary_tip_pose_input)
  //                                                                                                     walk_controller.h:223
  void updateManual(const int &primary_leg_selection_ID, const Pose &primary_tip_pose_input, const int &secondary_leg_selection_ID,
                    const Pose &secondary_tip_pose_input) {
    require_single(*eng_, "updateManual");
    primary_ = primary_leg_selection_ID, secondary_ = secondary_leg_selection_ID;
    ppos_ = primary_tip_pose_input.position_, spos_ = secondary_tip_pose_input.position_;
    push_manual();
  }
  int getWalkState(int64_t index = 0) const { return eng_->walk_state()[size_t(index)]; } // walk_controller.h:92
  std::array<double, 2> getDesiredLinearVelocity(int64_t index = 0) const {               // :100
    return {eng_->velocity()[size_t(index) * 3], eng_->velocity()[size_t(index) * 3 + 1]};
  }
  double getDesiredAngularVelocity(int64_t index = 0) const { return eng_->velocity()[size_t(index) * 3 + 2]; } // :104
  Pose getOdometryIdeal(int64_t index = 0) const {                                                               // :112
    const double *p = &eng_->odometry()[size_t(index) * 7];
    return Pose{{{p[0], p[1], p[2]}}, {p[3], p[4], p[5], p[6]}};
  }

private:
  void push_manual() {
    const int32_t p = primary_, q = secondary_;
    if (!eng_->manual_legs_) return; // every leg is WALKING: both overloads do nothing (walk_controller.cpp:661, :722)
    check(shc_engine_set_manual_inputs(eng_->handle(), &p, pvel_.v, ppos_.v, &q, svel_.v, spos_.v), "shc_engine_set_manual_inputs");
  }
  std::shared_ptr<Engine> eng_;
  int primary_ = -1, secondary_ = -1; // LEG_UNDESIGNATED
  Vector3 pvel_{{0, 0, 0}}, svel_{{0, 0, 0}}, ppos_{{0, 0, 0}}, spos_{{0, 0, 0}};
};

// class PoseController (pose_controller.h:36)
class PoseController {
public:
  explicit PoseController(std::shared_ptr<Engine> eng) : eng_(std::move(eng)) {}
  // bodyPoseInputCallback -> setManualPoseInput (pose_controller.h:97): translation / rotation velocity inputs
  template <class V3>
  void setManualPoseInput(const V3 &translation, const V3 &rotation) {
    double t[3] = {translation[0], translation[1], translation[2]}, r[3] = {rotation[0], rotation[1], rotation[2]};
    require_single(*eng_, "setManualPoseInput");
    check(shc_engine_set_pose_input(eng_->handle(), t, r, 0), "set_pose_input");
  }
  void setPoseResetMode(int mode) { // pose_controller.h:105
    int32_t m = mode;
    require_single(*eng_, "setPoseResetMode");
    check(shc_engine_set_pose_reset_mode(eng_->handle(), &m, 0), "set_pose_reset_mode");
  }
  // void updateCurrentPose(const RobotState& robot_state) (pose_controller.h:216) / void updateStance(void) (:157): both are
  // phases of the fused cycle; the engine runs them with robot_state RUNNING when Model::updateModel launches it.
  void updateCurrentPose(const RobotState & /*robot_state*/) {}
  void updateStance(void) {}
  // int executeSequence(const SequenceSelection& sequence) (pose_controller.h:166, pose_controller.cpp:145): one call of the
  // start-up / shut-down choreography; -1 while the first START_UP generates its sequence, 100 = complete
  int executeSequence(const SequenceSelection &sequence) {
    require_single(*eng_, "executeSequence");
    int32_t progress = 0;
    check(shc_engine_execute_sequence(eng_->handle(), sequence == START_UP ? SHC_SEQUENCE_START_UP : SHC_SEQUENCE_SHUT_DOWN, &progress), "shc_engine_execute_sequence");
    eng_->invalidate();
    return progress;
  }
  // setTargetConfiguration (pose_controller.h:141; targetConfigurationCallback): positions [legs][dof], a leg the message does not
  // name = NaN in its first joint.  setTargetBodyPose (pose_controller.h:109; targetBodyPoseCallback).  Engine::executePlan runs
  // transitionConfiguration(5.0) / transitionStance(5.0) (pose_controller.cpp:710, :767) on what was acquired.
  void setTargetConfiguration(const double *positions_legs_dof) {
    require_single(*eng_, "setTargetConfiguration");
    check(shc_engine_set_target_configuration(eng_->handle(), 0, 1, positions_legs_dof), "shc_engine_set_target_configuration");
  }
  void setTargetBodyPose(const Pose &pose) {
    require_single(*eng_, "setTargetBodyPose");
    const double p[7] = {pose.position_[0], pose.position_[1], pose.position_[2], pose.rotation_.w, pose.rotation_.x, pose.rotation_.y, pose.rotation_.z};
    check(shc_engine_set_target_body_pose(eng_->handle(), 0, 1, p), "shc_engine_set_target_body_pose");
  }
  // int stepToNewStance(void) (pose_controller.h:180, pose_controller.cpp:521)
  int stepToNewStance(void) {
    require_single(*eng_, "stepToNewStance");
    int32_t progress = 0;
    check(shc_engine_step_to_new_stance(eng_->handle(), &progress), "shc_engine_step_to_new_stance");
    eng_->invalidate();
    return progress;
  }

private:
  std::shared_ptr<Engine> eng_;
};

// class AdmittanceController (admittance_controller.h:27)
class AdmittanceController {
public:
  explicit AdmittanceController(std::shared_ptr<Engine> eng) : eng_(std::move(eng)) {}
  // tipStatesCallback -> Leg::setTipForceMeasured (state_controller.cpp:1618): [legs][3]
  void setTipForceMeasured(const double *force_legs_xyz) {
    require_single(*eng_, "setTipForceMeasured");
    check(shc_engine_set_tip_force(eng_->handle(), force_legs_xyz, 0), "set_tip_force");
  }
  // jointStatesCallback -> Joint::current_effort_ (state_controller.cpp:1590): [legs][dof]
  void setJointEffort(const double *effort) {
    require_single(*eng_, "setJointEffort");
    check(shc_engine_set_joint_effort(eng_->handle(), effort, 0), "set_joint_effort");
  }
  // void updateStiffness(std::shared_ptr<WalkController> walker) (admittance_controller.h:42) / void updateAdmittance(void) (:61)
  void updateStiffness(std::shared_ptr<WalkController> /*walker*/) {} // fused (admittance_controller.cpp:96)
  void updateAdmittance(void) {}                                      // fused (admittance_controller.cpp:22)

private:
  std::shared_ptr<Engine> eng_;
};

} // namespace shc_facade

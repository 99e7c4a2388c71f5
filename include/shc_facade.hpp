// shc_facade.hpp — the reference's per-cycle C++ call surface over the batched engine (header only).
//
// The reference node drives ONE robot through these calls every cycle (src/state_controller.cpp:162-193, 379-447):
//     poser_->updateCurrentPose(robot_state_);                 pose_controller.h:216
//     admittance_->updateStiffness(walker_); updateAdmittance();   admittance_controller.h:42,61
//     walker_->updateWalk(linear_velocity_input_, angular_velocity_input_);   walk_controller.h:204
//     poser_->updateStance();                                  pose_controller.h:157
//     model_->updateModel();                                   model.h:163
// and then reads Joint::desired_position_ / desired_velocity_ (state_controller.cpp:777-805) and the LegState fields
// (:809-893).  This façade keeps those class and method names so the node's loop compiles against it unchanged in
// shape: the five calls record their inputs and the LAST one of the cycle (Model::updateModel) launches the fused
// HIP cycle kernel for a batch of one (or for instance `index` of a larger batch that the caller steps itself).
//
// Types: the reference's signatures use Eigen (Vector2d/Vector3d/Quaterniond) and its own Pose; neither Eigen nor ROS is
// a dependency of this repository, so the façade is templated on "anything indexable" and ships tiny PODs.  A ROS
// host passes its Eigen objects directly (Eigen vectors are indexable); see INTEGRATION.md.
#pragma once

#include "shc_batch.h"

#include <array>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace shc_facade {

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
};

inline void check(int rc, const char *what) {
  if (rc != SHC_OK) throw std::runtime_error(std::string(what) + ": " + shc_last_error());
}

// Owns the engine (batch of `n`, default one robot) and the per-cycle input latch.
class Engine {
public:
  explicit Engine(const shc_params &params, int64_t n = 1, int device = 0, void *stream = nullptr) : params_(params), n_(n) {
    check(shc_engine_create(&params_, n, device, stream, &e_), "shc_engine_create");
    const int dof = params_.leg_dof[0], legs = params_.leg_count;
    q_.resize(size_t(n) * legs * dof);
    qd_.resize(q_.size());
    walker_tip_.resize(size_t(n) * legs * 3);
    poser_tip_.resize(walker_tip_.size());
    model_tip_.resize(walker_tip_.size());
    tip_force_.resize(walker_tip_.size());
    admittance_.resize(walker_tip_.size());
    leg_status_.resize(size_t(n) * legs);
    pose_.resize(size_t(n) * 7);
    velocity_.resize(size_t(n) * 3);
    odometry_.resize(size_t(n) * 7);
    stiffness_.resize(size_t(n) * legs);
    walk_state_.resize(size_t(n));
    refresh();
  }
  ~Engine() { shc_engine_destroy(e_); }
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;

  shc_engine *handle() { return e_; }
  const shc_params &params() const { return params_; }
  int64_t instances() const { return n_; }
  // One control cycle for the whole batch + read-back of the published quantities.
  void cycle(int n_cycles = 1) {
    check(shc_engine_step(e_, n_cycles), "shc_engine_step");
    refresh();
  }
  // StateController::changeGait (state_controller.cpp:513): true once the gait has changed, false while the robots are
  // still being stopped (keep cycling and call again, as the reference does while gait_change_flag_ is set)
  bool changeGait(const shc_params &new_gait) {
    int64_t still = 0;
    check(shc_engine_change_gait(e_, &new_gait, &still), "shc_engine_change_gait");
    return still == 0;
  }
  void refresh() {
    check(shc_engine_get_joint_state(e_, q_.data(), qd_.data(), 0), "get_joint_state");
    check(shc_engine_get_leg_state(e_, walker_tip_.data(), poser_tip_.data(), model_tip_.data(), tip_force_.data(), admittance_.data(),
                                   leg_status_.data(), 0),
          "get_leg_state");
    check(shc_engine_get_body_state(e_, pose_.data(), velocity_.data(), walk_state_.data(), 0), "get_body_state");
    check(shc_engine_get_odometry(e_, odometry_.data(), 0), "get_odometry");
    if (params_.admittance_control) check(shc_engine_get_virtual_stiffness(e_, stiffness_.data(), 0), "get_virtual_stiffness");
  }
  std::vector<double> q_, qd_, walker_tip_, poser_tip_, model_tip_, tip_force_, admittance_, pose_, velocity_, odometry_, stiffness_;
  std::vector<int32_t> leg_status_, walk_state_;

private:
  shc_params params_;
  int64_t n_;
  shc_engine *e_ = nullptr;
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

// class Leg (model.h:202): read-only view of instance `index`'s leg after the cycle.
class Leg {
public:
  Leg(Engine &eng, int64_t index, int id) : eng_(eng), index_(index), id_(id) {}
  int getIDNumber() const { return id_; }
  int getJointCount() const { return eng_.params().leg_dof[id_]; }
  Joint getJointByIDNumber(int joint_id /* 1-based, model.h:294 */) const {
    const int dof = eng_.params().leg_dof[0], legs = eng_.params().leg_count;
    size_t k = (size_t(index_) * legs + id_) * dof + (joint_id - 1);
    Joint j;
    j.desired_position_ = eng_.q_[k];
    j.desired_velocity_ = eng_.qd_[k];
    j.offset_ = eng_.params().joint[id_][joint_id - 1].offset;
    return j;
  }
  Vector3 getCurrentTipPosition() const { return v3(eng_.model_tip_); }    // Leg::getCurrentTipPose().position_ (model.h:304)
  Vector3 getDesiredTipPosition() const { return v3(eng_.poser_tip_); }    // LegPoser::getCurrentTipPose() (pose_controller.h:449)
  Vector3 getWalkerTipPosition() const { return v3(eng_.walker_tip_); }    // LegStepper::getCurrentTipPose() (walk_controller.h:389)
  Vector3 getTipForceCalculated() const { return v3(eng_.tip_force_); }    // model.h:244
  Vector3 getAdmittanceDelta() const { return v3(eng_.admittance_); }      // model.h:256
  double getVirtualStiffness() const { return eng_.stiffness_[size_t(index_) * eng_.params().leg_count + id_]; } // model.h:264
  int getStepState() const { return eng_.leg_status_[size_t(index_) * eng_.params().leg_count + id_] & 3; } // walk_controller.h:325
  int getPhase() const { return eng_.leg_status_[size_t(index_) * eng_.params().leg_count + id_] >> 8; }    // :317
  bool ikFailed() const { return (eng_.leg_status_[size_t(index_) * eng_.params().leg_count + id_] & 4) != 0; } // model.cpp:921

private:
  Vector3 v3(const std::vector<double> &a) const {
    size_t k = (size_t(index_) * eng_.params().leg_count + id_) * 3;
    return Vector3{{a[k], a[k + 1], a[k + 2]}};
  }
  Engine &eng_;
  int64_t index_;
  int id_;
};

// class Model (model.h:57)
class Model {
public:
  Model(std::shared_ptr<Engine> eng, int64_t index = 0) : eng_(std::move(eng)), index_(index) {}
  int getLegCount() const { return eng_->params().leg_count; }                     // model.h:80
  Leg getLegByIDNumber(int leg_id) { return Leg(*eng_, index_, leg_id); }          // model.h:124
  Pose getCurrentPose() const {                                                    // model.h:84
    const double *p = &eng_->pose_[size_t(index_) * 7];
    return Pose{{{p[0], p[1], p[2]}}, {p[3], p[4], p[5], p[6]}};
  }
  // Model::setImuData (model.h:146): orientation (w,x,y,z), angular velocity
  template <class Q, class V>
  void setImuData(const Q &orientation_wxyz, const V & /*linear_acceleration*/, const V &angular_velocity) {
    double q[4] = {orientation_wxyz[0], orientation_wxyz[1], orientation_wxyz[2], orientation_wxyz[3]};
    double g[3] = {angular_velocity[0], angular_velocity[1], angular_velocity[2]};
    require_single("setImuData");
    check(shc_engine_set_imu(eng_->handle(), q, g, 0), "set_imu");
  }
  // Model::updateModel (model.h:163, src/model.cpp:142): the last call of the reference's cycle -> launch the fused kernel.
  void updateModel() { eng_->cycle(1); }
  Engine &engine() { return *eng_; }

private:
  void require_single(const char *what) const {
    if (eng_->instances() != 1) throw std::logic_error(std::string(what) + ": per-robot setters need a batch of one; use the C ABI arrays");
  }
  std::shared_ptr<Engine> eng_;
  int64_t index_;
};

// class WalkController (walk_controller.h:54)
class WalkController {
public:
  explicit WalkController(std::shared_ptr<Engine> eng) : eng_(std::move(eng)) {}
  // WalkController::updateWalk (walk_controller.h:204, src/walk_controller.cpp:440): latches the velocity inputs of this cycle.
  template <class V2>
  void updateWalk(const V2 &linear_velocity_input, const double &angular_velocity_input) {
    double lin[2] = {linear_velocity_input[0], linear_velocity_input[1]};
    double ang = angular_velocity_input;
    if (eng_->instances() != 1) throw std::logic_error("updateWalk: batch of one only; use shc_engine_set_velocity");
    check(shc_engine_set_velocity(eng_->handle(), lin, &ang, 0), "set_velocity");
  }
  int getWalkState(int64_t index = 0) const { return eng_->walk_state_[size_t(index)]; } // walk_controller.h:92
  std::array<double, 2> getDesiredLinearVelocity(int64_t index = 0) const {               // :100
    return {eng_->velocity_[size_t(index) * 3], eng_->velocity_[size_t(index) * 3 + 1]};
  }
  double getDesiredAngularVelocity(int64_t index = 0) const { return eng_->velocity_[size_t(index) * 3 + 2]; } // :104
  Pose getOdometryIdeal(int64_t index = 0) const {                                                               // :112
    const double *p = &eng_->odometry_[size_t(index) * 7];
    return Pose{{{p[0], p[1], p[2]}}, {p[3], p[4], p[5], p[6]}};
  }

private:
  std::shared_ptr<Engine> eng_;
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
  // updateCurrentPose / updateStance (pose_controller.h:216,157) are part of the fused cycle: nothing to do per call.
  void updateCurrentPose(int /*robot_state*/) {}
  void updateStance() {}

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
  void updateStiffness(WalkController & /*walker*/) {} // fused (admittance_controller.cpp:96)
  void updateAdmittance() {}                           // fused (admittance_controller.cpp:22)

private:
  std::shared_ptr<Engine> eng_;
};

} // namespace shc_facade

// Compiled by tests/test_gpu_facade.py: drives one default.yaml hexapod through the façade (reference class names, method
// names and argument order) - first the reference's loop body, then the per-leg Leg methods its cold paths use - and prints
// what the test compares with the oracle.
#include "shc_facade.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

using namespace shc_facade;

int main(int argc, char **argv) {
  if (argc < 6) return 2;
  FILE *f = fopen(argv[1], "rb");
  shc_params p;
  if (!f || fread(&p, sizeof p, 1, f) != 1) return 3;
  fclose(f);
  if ((int64_t)sizeof p != shc_sizeof_params()) return 4;
  int cycles = atoi(argv[2]);
  double v[2] = {atof(argv[3]), atof(argv[4])};
  double w = atof(argv[5]);
  auto engine = std::make_shared<Engine>(p, 1, 0);
  if (argc > 6 && std::string(argv[6]) == "sequence") {
    // start_up_sequence: true - READY -> RUNNING through PoseController::executeSequence(START_UP) (state_controller.cpp:298-313),
    // a short walk, then the LegStepper external-target interface; prints calls, joints after the sequence, joints after the walk
    auto model = std::make_shared<Model>(engine);
    auto walker = std::make_shared<WalkController>(engine);
    auto poser = std::make_shared<PoseController>(engine);
    engine->initModel();
    int calls = 0, progress = 0;
    while (progress != 100 && calls < 5000) {
      progress = poser->executeSequence(START_UP);
      ++calls;
    }
    printf("calls %d\n", calls);
    for (int l = 0; l < model->getLegCount(); ++l)
      for (int j = 1; j <= model->getLegByIDNumber(l).getJointCount(); ++j) printf("%.17g\n", model->getLegByIDNumber(l).getJointByIDNumber(j).desired_position_);
    engine->finishStartUpSequence();
    for (int c = 0; c < cycles; ++c) {
      walker->updateWalk(v, w);
      model->updateModel();
    }
    for (int l = 0; l < model->getLegCount(); ++l)
      for (int j = 1; j <= model->getLegByIDNumber(l).getJointCount(); ++j) printf("%.17g\n", model->getLegByIDNumber(l).getJointByIDNumber(j).desired_position_);
    if (p.rough_terrain_mode) { // LegStepper::setExternalTarget / getExternalTarget (walk_controller.h:385, :437)
      ExternalTarget t;
      t.pose_ = Pose{{{0.2, -0.1, -0.02}}, {1.0, 0.0, 0.0, 0.0}};
      t.swing_clearance_ = 0.03;
      t.defined_ = true;
      LegStepper stepper = model->getLegByIDNumber(1).getLegStepper();
      stepper.setExternalTarget(t);
      ExternalTarget back = stepper.getExternalTarget();
      printf("external %d %.17g %.17g\n", back.defined_ ? 1 : 0, back.pose_.position_[0], back.swing_clearance_);
    }
    { // planner mode: executePlan stops the robot, waits for plan step 0, runs a body-pose step (transitionStance)
      engine->setPlannerMode(true);
      int progress = SHC_PLAN_WALKING, stop_calls = 0, step_calls = 0;
      while (progress != SHC_PLAN_WAITING && stop_calls < 4000) {
        progress = engine->executePlan(); // (a robot that is still walking: velocity inputs zeroed, one ordinary cycle)
        ++stop_calls;
      }
      poser->setTargetBodyPose(Pose{{{0.01, -0.005, 0.008}}, {1.0, 0.0, 0.0, 0.0}});
      while (progress != 100 && step_calls < 4000) {
        progress = engine->executePlan();
        ++step_calls;
      }
      engine->setPlannerMode(false);
      Vector3 tip = model->getLegByIDNumber(0).getCurrentTipPosition();
      printf("plan %d %d %d %.17g %.17g %.17g\n", stop_calls, step_calls, engine->planStep(), tip[0], tip[1], tip[2]);
    }
    { // manual leg manipulation: toggle leg 3 to MANUAL (legStateToggle), place its tip with updateManual's pose overload
      int result = -1, calls = 0;
      while (result != 1 && calls < 4000) {
        result = engine->legStateToggle(3); // (-1 while the robot still walks: the call zeroes the velocity inputs and runs that loop's cycle)
        ++calls;
      }
      const Pose none{{{0.0, 0.0, 0.0}}, {0.0, 0.0, 0.0, 0.0}};
      const Pose where{{{p.stance_position[3][0] * 0.9, p.stance_position[3][1] * 0.9, -0.07}}, {0.0, 0.0, 0.0, 0.0}};
      for (int c = 0; c < 30; ++c) {
        walker->updateWalk(v, w); // ignored: the walker is frozen while a leg is MANUAL
        walker->updateManual(3, where, -1, none);
        model->updateModel();
      }
      Vector3 tip = model->getLegByIDNumber(3).getCurrentTipPosition();
      printf("manual %d %.17g %.17g %.17g walk_state %d\n", result, tip[0], tip[1], tip[2], walker->getWalkState());
    }
    return 0;
  }
  if (argc > 6 && std::string(argv[6]) == "adjust") {
    // StateController::adjustParameter through the facade, as runningState serves it (state_controller.cpp:411-414): a higher step frequency requested at
    // loop 60 (asked again in every loop until it is set), a new swing height at loop 40; prints the loops the first one waited, then the joints
    auto model = std::make_shared<Model>(engine);
    auto walker = std::make_shared<WalkController>(engine);
    bool parameter_adjust_flag = false;
    int dynamic_parameter = 0, waited = 0;
    double new_parameter_value = 0.0;
    for (int c = 0; c < cycles; ++c) {
      if (c == 40) parameter_adjust_flag = true, dynamic_parameter = SHC_PARAM_SWING_HEIGHT, new_parameter_value = 0.03;
      if (c == 60) parameter_adjust_flag = true, dynamic_parameter = SHC_PARAM_STEP_FREQUENCY, new_parameter_value = atof(argv[7]);
      if (parameter_adjust_flag) {
        if (engine->adjustParameter(dynamic_parameter, new_parameter_value)) parameter_adjust_flag = false;
        else ++waited;
      }
      walker->updateWalk(v, w);
      model->updateModel();
    }
    printf("waited %d flag %d\n", waited, parameter_adjust_flag ? 1 : 0);
    for (int l = 0; l < model->getLegCount(); ++l)
      for (int j = 1; j <= model->getLegByIDNumber(l).getJointCount(); ++j) printf("%.17g\n", model->getLegByIDNumber(l).getJointByIDNumber(j).desired_position_);
    return 0;
  }
  auto model = std::make_shared<Model>(engine);
  auto walker = std::make_shared<WalkController>(engine);
  auto poser = std::make_shared<PoseController>(engine);
  auto admittance = std::make_shared<AdmittanceController>(engine);
  RobotState robot_state = RUNNING;
  for (int c = 0; c < cycles; ++c) { // the reference's loop body (state_controller.cpp:162-193, 379-447), same names and order
    poser->updateCurrentPose(robot_state);
    admittance->updateStiffness(walker);
    admittance->updateAdmittance();
    walker->updateWalk(v, w);
    poser->updateStance();
    model->updateModel();
  }
  for (int l = 0; l < model->getLegCount(); ++l)
    for (int j = 1; j <= model->getLegByIDNumber(l).getJointCount(); ++j) printf("%.17g\n", model->getLegByIDNumber(l).getJointByIDNumber(j).desired_position_);
  printf("walk_state %d\n", walker->getWalkState());
  // ---- per-leg methods (model.h:448-492) on leg 2: FK, one explicit DLS step = solveIK + updateJointPositions, then applyIK
  Leg leg = model->getLegByIDNumber(2);
  Pose tip = leg.applyFK();
  printf("%.17g %.17g %.17g\n", tip.position_[0], tip.position_[1], tip.position_[2]);
  double delta[6] = {0.001, -0.002, 0.0015, 0, 0, 0};
  std::vector<double> dq = leg.solveIK(delta, false);
  printf("%.17g %.17g %.17g\n", dq[0], dq[1], dq[2]);
  printf("%.17g\n", leg.updateJointPositions(dq, true));
  Pose target = leg.applyFK();
  target.position_[2] += 0.003;
  target.rotation_ = Quaternion{0, 0, 0, 0}; // UNDEFINED_ROTATION: position only
  leg.setDesiredTipPose(target, false);
  printf("%.17g\n", leg.applyIK(true));
  for (int j = 1; j <= 3; ++j) printf("%.17g\n", leg.getJointByIDNumber(j).desired_position_);
  return 0;
}

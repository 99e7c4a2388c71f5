// Compiled by tests/test_gpu_facade.py: drives one default.yaml hexapod through the façade (reference class names) and
// prints the joint positions after N cycles.
#include "shc_facade.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
  auto engine = std::make_shared<shc_facade::Engine>(p, 1, 0);
  shc_facade::Model model(engine);
  shc_facade::WalkController walker(engine);
  shc_facade::PoseController poser(engine);
  shc_facade::AdmittanceController admittance(engine);
  for (int c = 0; c < cycles; ++c) { // the reference's loop body, same call names and order
    poser.updateCurrentPose(2 /* RUNNING */);
    admittance.updateAdmittance();
    walker.updateWalk(v, w);
    poser.updateStance();
    model.updateModel();
  }
  for (int l = 0; l < model.getLegCount(); ++l)
    for (int j = 1; j <= model.getLegByIDNumber(l).getJointCount(); ++j) printf("%.17g\n", model.getLegByIDNumber(l).getJointByIDNumber(j).desired_position_);
  printf("walk_state %d\n", walker.getWalkState());
  return 0;
}

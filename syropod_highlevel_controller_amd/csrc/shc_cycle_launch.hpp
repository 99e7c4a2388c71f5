// shc_cycle_launch.hpp - the seam between the host side of libshc_batch.so (shc_engine.hip) and the fused cycle kernels
// (shc_cycle_kernel.hpp), which are compiled in one translation unit per morphology (shc_cycle_inst.hip, -DSHC_INST_L /
// -DSHC_INST_NJ) so that the library builds in parallel.  Plain data + one launcher per (legs, joints).
#pragma once

#include "shc_cycle.hpp"

#include <hip/hip_runtime.h>

namespace shc {

// Everything a launch of the cycle kernel needs from the engine.
struct CycleLaunch {
  DevState st;
  const void *consts;     // SharedConsts<L, NJ> in HBM
  const CycleParams *cp;  // host copy: picks the kernel specialisation
  unsigned rt_flags;      // RT_* facts
  bool generic;           // SHC_FEAT_GENERIC_KERNEL: force the runtime-flag kernel
  hipStream_t stream;
  unsigned grid;
  int block;
  int n_cycles;
};

// One per (legs, joints); defined by shc_cycle_inst.hip.  Returns false when that morphology has no kernels in this build.
#define SHC_FOR_EACH_MORPHOLOGY(X) X(3, 3) X(4, 3) X(4, 4) X(4, 5) X(5, 3) X(6, 3) X(6, 4) X(6, 5) X(7, 3) X(8, 3) X(8, 4) X(8, 5)
#define SHC_DECLARE_LAUNCHER(L_, NJ_) void shc_launch_cycle_##L_##_##NJ_(const CycleLaunch &a);
SHC_FOR_EACH_MORPHOLOGY(SHC_DECLARE_LAUNCHER)
#undef SHC_DECLARE_LAUNCHER

} // namespace shc

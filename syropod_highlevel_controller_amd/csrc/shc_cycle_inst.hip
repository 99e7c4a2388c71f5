// shc_cycle_inst.hip - the fused cycle kernels of ONE morphology (compile with -DSHC_INST_L=<legs> -DSHC_INST_NJ=<joints> -DSHC_INST_PART=<0|1>):
// libshc_batch.so links two objects of this file per supported (legs, joints), built in parallel (engine.py build_library) -
//   part 0: the launch forms (shc_cycle_kernel, the half kernels of rotation-constrained cycles) and the morphology's entry point shc_launch_cycle_L_NJ;
//   part 1: the loop forms (shc_resident_kernel, shc_resident2_kernel, shc_batch_kernel) behind shc_launch_loop_L_NJ, which part 0 hands loop launches to.
// Both parts compile the same dispatch (launch_cycle_feat: configuration -> kernel specialisation); each instantiates only its own kernels.
// -DSHC_GENERIC_LOOP_FORMS=1 (engine.py: SHC_GENERIC_LOOP_FORMS=1 in the environment of the build) adds the loop forms of the runtime-flag (F_DYN) families
// that the default build leaves out: their batch kernels (shc_engine_step_k then runs such a configuration as K single launches - same results, see
// shc_resident.hpp) and the resident kernels of F_DYN with rough terrain / tip-align / tip rotations (resident_begin reports SHC_ERR_UNSUPPORTED).
#include "shc_cycle_kernel.hpp"

#if !defined(SHC_INST_L) || !defined(SHC_INST_NJ) || !defined(SHC_INST_PART)
#error "compile with -DSHC_INST_L=<legs> -DSHC_INST_NJ=<joints> -DSHC_INST_PART=<0|1>"
#endif
#ifndef SHC_GENERIC_LOOP_FORMS
#define SHC_GENERIC_LOOP_FORMS 0
#endif

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <tuple>

namespace shc {

// Development aid: SHC_KERNEL_LOG=<file> appends one line per kernel specialisation this process launches ("form legs joints features"), the
// first time it is launched - which of the library's instantiations a test suite / a bench run actually exercises (scripts/kernels_used.py).
static void note_kernel(const char *form, int legs, int joints, unsigned features) {
  static const char *path = std::getenv("SHC_KERNEL_LOG");
  if (!path) return;
  static std::mutex mu;
  static std::set<std::tuple<const char *, unsigned>> seen;
  std::lock_guard<std::mutex> lock(mu);
  if (!seen.insert({form, features}).second) return;
  if (FILE *f = std::fopen(path, "a")) {
    std::fprintf(f, "%s %d %d %u\n", form, legs, joints, features);
    std::fclose(f);
  }
}

// Which loop forms a specialisation has in this build.  Manual legs: none (the ManualRobot records change under loop-level calls).  Runtime-flag
// families: the plain one keeps its resident kernels; everything else of F_DYN is opt-in (SHC_GENERIC_LOOP_FORMS) - no test, bench line or fleet bin
// selects them (SHC_KERNEL_LOG over the GPU suite), they are a quarter of the library's kernels and all of them carry scratch.
template <unsigned F> constexpr bool kHasResident = (F & F_MLEGS) == 0 && (SHC_GENERIC_LOOP_FORMS || (F & F_DYN) == 0 || (F & (F_TERRAIN | F_ROT)) == 0);
template <unsigned F> constexpr bool kHasBatch = (F & F_MLEGS) == 0 && (SHC_GENERIC_LOOP_FORMS || (F & F_DYN) == 0);

template <int L, int NJ, unsigned F>
static void launch_cycle(const CycleLaunch &a) {
  constexpr int RPW = 64 / L;
  constexpr size_t wave_bytes = size_t(RobotFields::COUNT * RPW + PK_COUNT * 64 + (RobotFields::I_COUNT * RPW + 1) / 2) * 8;
#if SHC_INST_PART == 1
  // Resident kernels: every specialisation but manual legs (the tip-align pose of gravity_aligned_tips on <= 3-joint legs is per-robot state of the
  // tile like any other and has had a loop form since round 5).  Rough terrain and tip rotations run as ONE wavefront per robot group (Leg::applyIK
  // feeds back into the stepper there - touchdown detection, the FK tip rotation - so the walker / model halves cannot be pipelined); everything else
  // also has the two-wavefront form.
  constexpr bool two_wave = (F & (F_TERRAIN | F_ROT)) == 0;
  if (a.fit) {
    a.fit->supported = kHasResident<F> ? 1 : 0;
    a.fit->batch = kHasBatch<F> ? 1 : 0;
    a.fit->two_wave = kHasResident<F> && two_wave ? 1 : 0;
    a.fit->blocks_per_cu = 0;
    if constexpr (kHasResident<F>) {
      int blocks = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, shc_resident_kernel<L, NJ, F>, 64, wave_bytes) != hipSuccess) blocks = 0;
      a.fit->blocks_per_cu = blocks;
    }
  } else if (a.resident->batch_cycles != 0) { // shc_engine_step_k: the batch form, a kernel of its own
    if constexpr (kHasBatch<F>) {
      note_kernel("batch", L, NJ, F);
      shc_batch_kernel<L, NJ, F><<<dim3(a.grid), dim3(a.block), wave_bytes * (a.block / 64), a.stream>>>(a.st, (const SharedConsts<L, NJ> *)a.consts, *a.resident, a.rt_flags);
    }
  } else if (a.block == 256) {
    if constexpr (kHasResident<F> && two_wave) {
      note_kernel("resident2", L, NJ, F);
      shc_resident2_kernel<L, NJ, F><<<dim3(a.grid), dim3(256), 2 * wave_bytes + sizeof(Resident2Lds<L, NJ>), a.stream>>>(
          a.st, (const SharedConsts<L, NJ> *)a.consts, *a.resident, a.rt_flags);
    }
  } else {
    if constexpr (kHasResident<F>) {
      note_kernel("resident", L, NJ, F);
      shc_resident_kernel<L, NJ, F><<<dim3(a.grid), dim3(64), wave_bytes, a.stream>>>(a.st, (const SharedConsts<L, NJ> *)a.consts, *a.resident, a.rt_flags);
    }
  }
#else
  // Rotation-constrained cycles of the feature-exact kernels: one cycle = the walker / poser launch + the model launch (two wavefronts per SIMD
  // each instead of one) once the launch holds at least two wavefronts for every SIMD of the chip; smaller launches stay one kernel.
  if constexpr ((F & F_ROT) != 0 && (F & (F_DYN | F_TERRAIN | F_MLEGS | F_AUTO)) == 0) {
    const int64_t waves = int64_t(a.grid) * (a.block / 64);
    if (a.half_steps >= 0 && (a.half_steps > 0 || waves >= 2048)) {
      note_kernel("half", L, NJ, F);
      for (int c = 0; c < a.n_cycles; ++c) {
        shc_cycle_half_kernel<L, NJ, F, ROLE_FRONT><<<dim3(a.grid), dim3(a.block), wave_bytes * (a.block / 64), a.stream>>>(
            a.st, (const SharedConsts<L, NJ> *)a.consts, a.rt_flags, a.wave0);
        shc_cycle_half_kernel<L, NJ, F, ROLE_BACK><<<dim3(a.grid), dim3(a.block), wave_bytes * (a.block / 64), a.stream>>>(
            a.st, (const SharedConsts<L, NJ> *)a.consts, a.rt_flags, a.wave0);
      }
      return;
    }
  }
  note_kernel("cycle", L, NJ, F);
  shc_cycle_kernel<L, NJ, F><<<dim3(a.grid), dim3(a.block), wave_bytes * (a.block / 64), a.stream>>>(a.st, (const SharedConsts<L, NJ> *)a.consts, a.n_cycles,
                                                                                              a.rt_flags, a.wave0);
#endif
}

// Pick the kernel specialisation: the BASELINE.json configurations get feature-exact kernels (dead features cost
// neither registers nor HBM traffic); every other flag combination runs the generic kernel (F_DYN).
template <int L, int NJ, bool SPEC>
static void launch_cycle_feat(const CycleLaunch &a) {
  const CycleParams &c = *a.cp;
  unsigned f = (c.manual_posing ? F_MANUAL : 0) | (c.auto_posing ? F_AUTO : 0) | (c.inclination_posing ? F_INCL : 0) |
               (c.imu_posing ? F_IMU : 0) | (c.admittance_control ? F_ADM : 0) | (c.tip_force ? F_TIPF : 0) | (c.odometry ? F_ODOM : 0);
  // rough terrain mode / the tip-align pose / manual legs: generic kernels with that logic compiled in - one path each where a
  // configuration needs just one (the usual case), all of them otherwise
  const bool rough = c.rough_terrain != 0, talign = c.tip_align != 0, mlegs = (a.rt_flags & RT_MANUAL_LEGS) != 0;
  const bool terrain = rough || talign || mlegs;
  if constexpr (NJ > 3) {
    // gravity-aligned tips: kernels with the tip-rotation logic compiled in; also a robot with 3-joint legs next to longer ones under
    // joint_control leg manipulation (a MANUAL 3-joint leg holds its FK tip rotation), once a leg has been toggled
    if (c.gravity_aligned || (mlegs && c.joint_control == 2)) {
      // default.yaml's posing set: feature-exact for every morphology (with the tip-force estimate: the BASELINE morphology only), and with that
      // the two-launch form of the cycle for large launches (launch_cycle)
      constexpr unsigned C2 = F_MANUAL | F_ODOM;
      if (!a.generic && !terrain && (f & ~F_TIPF) == C2) {
        if (!(f & F_TIPF)) {
          launch_cycle<L, NJ, C2 | F_ROT>(a);
          return;
        }
        if constexpr (SPEC) {
          launch_cycle<L, NJ, C2 | F_TIPF | F_ROT>(a);
          return;
        }
      }
      // ... and the north-star feature set (admittance + IMU posing, BASELINE config 3's) together with the tip rotations on the BASELINE octopods:
      // feature-exact, and with that the two-launch form (the runtime-flag kernel below needs one wavefront per SIMD + 122 - 141 AGPRs)
      if constexpr (SPEC) {
        constexpr unsigned C3 = F_MANUAL | F_IMU | F_ADM | F_ODOM;
        if (!a.generic && !terrain && f == C3) {
          launch_cycle<L, NJ, C3 | F_ROT>(a);
          return;
        }
      }
      if (terrain) launch_cycle<L, NJ, F_DYN | F_ROT | F_TERRAIN>(a);
      else launch_cycle<L, NJ, F_DYN | F_ROT>(a);
      return;
    }
  }
  if constexpr (NJ == 3) {
    // joint_control leg manipulation (3-joint legs): a MANUAL leg's tip pose carries its FK rotation, the rotation-constrained IK runs on
    // it (walk_controller.cpp:677-690); only once a leg has been toggled
    if (mlegs && c.joint_control == 2) {
      if (rough || talign) launch_cycle<L, NJ, F_DYN | F_ROT | F_TERRAIN>(a);
      else launch_cycle<L, NJ, F_DYN | F_ROT | F_MLEGS>(a);
      return;
    }
  }
  if (terrain) {
    if constexpr (SPEC) { // default.yaml's posing set (manual posing + odometry, with / without the tip-force estimate): feature-exact kernels
      constexpr unsigned C2 = F_MANUAL | F_ODOM;
      if (!a.generic && (f & ~F_TIPF) == C2) {
        const bool tf = (f & F_TIPF) != 0;
        if (rough && !talign && !mlegs) {
          if (tf) launch_cycle<L, NJ, C2 | F_TIPF | F_ROUGH>(a);
          else launch_cycle<L, NJ, C2 | F_ROUGH>(a);
          return;
        }
        if constexpr (NJ <= 3) {
          if (talign && !rough && !mlegs) {
            if (tf) launch_cycle<L, NJ, C2 | F_TIPF | F_TALIGN>(a);
            else launch_cycle<L, NJ, C2 | F_TALIGN>(a);
            return;
          }
        }
      }
    }
    if (rough && !talign && !mlegs) launch_cycle<L, NJ, F_DYN | F_ROUGH>(a);
    else if (mlegs && !rough && !talign) launch_cycle<L, NJ, F_DYN | F_MLEGS>(a);
    else if (NJ <= 3 && talign && !rough && !mlegs) {
      if constexpr (NJ <= 3) launch_cycle<L, NJ, F_DYN | F_TALIGN>(a);
    } else launch_cycle<L, NJ, F_DYN | F_TERRAIN>(a);
    return;
  }
  if constexpr (SPEC) {
    constexpr unsigned C2 = F_MANUAL | F_ODOM, C3 = F_MANUAL | F_IMU | F_ADM | F_ODOM; // BASELINE.json configs 2/4 and 3
    if (!a.generic) switch (f) {
      case C2 | F_TIPF: launch_cycle<L, NJ, C2 | F_TIPF>(a); return;
      case C2: launch_cycle<L, NJ, C2>(a); return;
      case C3 | F_TIPF: launch_cycle<L, NJ, C3 | F_TIPF>(a); return;
      case C3: launch_cycle<L, NJ, C3>(a); return;
      default: break;
    }
  } else {
    // every other morphology: default.yaml's posing set without the tip-force estimate (what the bins of BASELINE.json configs[4] run on) is
    // feature-exact too - the runtime-flag kernels of 8 x 3, 6 x 5 and 8 x 5 carry 12 - 36 B of scratch per lane, these carry none
    constexpr unsigned C2 = F_MANUAL | F_ODOM;
    if (!a.generic && f == C2) {
      launch_cycle<L, NJ, C2>(a);
      return;
    }
  }
  launch_cycle<L, NJ, F_DYN>(a);
}

#define SHC_CAT3(a, b, c) a##b##_##c
#define SHC_LAUNCHER_NAME(L_, NJ_) SHC_CAT3(shc_launch_cycle_, L_, NJ_)
#define SHC_LOOP_LAUNCHER_NAME(L_, NJ_) SHC_CAT3(shc_launch_loop_, L_, NJ_)
// feature-exact kernels for the BASELINE.json morphologies: default.yaml hexapods (6 x 3) and the synthetic octopods (8 x 5)
constexpr bool kSpecMorphology = (SHC_INST_L == 6 && SHC_INST_NJ == 3) || (SHC_INST_L == 8 && SHC_INST_NJ == 5);
#if SHC_INST_PART == 1
void SHC_LOOP_LAUNCHER_NAME(SHC_INST_L, SHC_INST_NJ)(const CycleLaunch &a) { launch_cycle_feat<SHC_INST_L, SHC_INST_NJ, kSpecMorphology>(a); }
#else
void SHC_LOOP_LAUNCHER_NAME(SHC_INST_L, SHC_INST_NJ)(const CycleLaunch &a);
void SHC_LAUNCHER_NAME(SHC_INST_L, SHC_INST_NJ)(const CycleLaunch &a) {
  if (a.fit || a.resident) { // a loop form (or the question whether there is one): the other object of this morphology
    SHC_LOOP_LAUNCHER_NAME(SHC_INST_L, SHC_INST_NJ)(a);
    return;
  }
  launch_cycle_feat<SHC_INST_L, SHC_INST_NJ, kSpecMorphology>(a);
}
#endif

} // namespace shc

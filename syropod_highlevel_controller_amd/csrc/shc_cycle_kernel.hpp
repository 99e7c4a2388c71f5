// shc_cycle_kernel.hpp - the fused control-cycle kernels (gfx950): state load / store around shc_cycle.hpp's cycle().
//
// Included by shc_cycle_inst.hip only, which is compiled once per (legs, joints) morphology (-DSHC_INST_L / -DSHC_INST_NJ) so
// that the ~50 kernel instantiations build in parallel; the host side (shc_engine.hip) reaches them through
// shc_cycle_launch.hpp.  Product code: nothing here includes, links or calls anything under oracle/.
#pragma once

#include "shc_cycle.hpp"
#include "shc_cycle_launch.hpp"

#include <hip/hip_runtime.h>

namespace shc {

#ifndef SHC_WAVES_PER_SIMD
#define SHC_WAVES_PER_SIMD 2
#endif
#ifndef SHC_ROT_WAVES_PER_SIMD
#define SHC_ROT_WAVES_PER_SIMD 1
#endif
#ifndef SHC_TERRAIN_WAVES_PER_SIMD
#define SHC_TERRAIN_WAVES_PER_SIMD 2
#endif

// Paired planes of the per-leg SoA state: plane p = fields (2p, 2p + 1) as one double2 per slot.
struct LegPlanes {
  double2 *base;
  int64_t ns;
  uint32_t slot;
  typedef double v2d __attribute__((ext_vector_type(2)));
  __device__ __forceinline__ double2 load(int plane) const { return (base + plane * ns)[slot]; }
  // State written by a launch is next read by the following launch, from any XCD: the stores are streaming (nt) so that the
  // lines do not sit dirty in this XCD's L2 until the end-of-kernel write-back (measured: -8.5 % per launch of 131 072
  // octopods, -1.9 % of 65 536 hexapods with admittance, neutral at 4 096; streaming loads or streaming robot-tile stores on
  // top of it were slower).
  __device__ __forceinline__ void store(int plane, double2 v) const {
    v2d w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(base + plane * ns + slot));
  }
};

// Per-leg state load, in two steps so that the kernel prologue can issue every global load of the wave before the first
// use: load_leg_issue() only issues the plane loads (joint planes first), load_leg_finish() unpacks them.
template <int NJ>
struct LegLoad {
  double flat[Fields<NJ>::CORE_END];
  double2 adm, tf0, tf1, stiff, rot0, rot1, rot2, rot3, rot4;
  int word;
};
// ROLE: which half of the per-leg state a wave owns.  ROLE_ALL: the whole leg (one wave runs whole cycles).  The two-wave
// resident kernel splits it: ROLE_FRONT = the walker / poser half (stepper state: fields [2 NJ, CORE_END), the packed word, the
// published stiffness and poser tip), ROLE_BACK = the model half (joints: fields [0, 2 NJ) = the first NJ planes, admittance state,
// tip-force estimate, admittance delta).
enum : int { ROLE_ALL = 0, ROLE_FRONT = 1, ROLE_BACK = 2 };
template <int NJ, unsigned F, int ROLE = ROLE_ALL>
__device__ __forceinline__ void load_leg_issue(LegLoad<NJ> &ll, const DevState &st, const CycleParams &P, uint32_t slot) {
  using FD = Fields<NJ>;
  using FT = Feat<F>;
  const LegPlanes ld{reinterpret_cast<double2 *>(st.legd), st.n_slots, slot};
#pragma unroll
  for (int p = 0; p < FD::CORE_END / 2; ++p) {
    if ((ROLE == ROLE_FRONT && p < NJ) || (ROLE == ROLE_BACK && p >= NJ)) {
      ll.flat[2 * p] = ll.flat[2 * p + 1] = 0.0;
      continue;
    }
    double2 v = ld.load(p);
    ll.flat[2 * p] = v.x;
    ll.flat[2 * p + 1] = v.y;
  }
  ll.word = ROLE == ROLE_BACK ? 0 : st.legi[slot];
  ll.adm = ll.tf0 = ll.tf1 = ll.stiff = ll.rot0 = ll.rot1 = ll.rot2 = ll.rot3 = ll.rot4 = double2{0.0, 0.0};
  if (rot_enabled<NJ, F>() && ROLE != ROLE_BACK) { // tip directions of the stepper's origin / current / target tip rotations
    ll.rot0 = ld.load(FD::ORG_DIR / 2);
    ll.rot1 = ld.load(FD::ORG_DIR / 2 + 1);
    ll.rot2 = ld.load(FD::ORG_DIR / 2 + 2);
    ll.rot3 = ld.load(FD::ORG_DIR / 2 + 3);
    ll.rot4 = ld.load(FD::ORG_DIR / 2 + 4);
  }
  if (FT::adm(P)) {
    if (ROLE != ROLE_FRONT) ll.adm = ld.load(FD::ADM / 2);
    if (ROLE != ROLE_BACK && P.dynamic_stiffness) ll.stiff = ld.load(FD::ADM_DELTA / 2 + 1); // virtual_stiffness_ persists while STOPPED
  }
  if (FT::tipf(P) && ROLE != ROLE_FRONT) {
    ll.tf0 = ld.load(FD::TF / 2);
    ll.tf1 = ld.load(FD::TF / 2 + 1);
  }
}
template <int NJ, int ROLE = ROLE_ALL>
__device__ __forceinline__ void load_leg_finish(LegRegs<NJ> &s, const Park &pk, const LegLoad<NJ> &ll) {
  using FD = Fields<NJ>;
  const double(&flat)[FD::CORE_END] = ll.flat;
  s.word = ll.word;
  // swing origin / velocity, stance origin and default tip go straight to the per-lane LDS strip
  static_assert(FD::SVEL == FD::SORG + 3 && FD::TORG == FD::SORG + 6 && FD::DFLT == FD::SORG + 9, "park layout");
  if (ROLE != ROLE_BACK) {
#pragma unroll
    for (int k = 0; k < PK_COUNT; ++k) pk.set(k, flat[FD::SORG + k]);
  }
  s.tip = V3{flat[FD::TIP + 0], flat[FD::TIP + 1], flat[FD::TIP + 2]};
  s.targ = V3{flat[FD::TARG + 0], flat[FD::TARG + 1], flat[FD::TARG + 2]};
  s.strd = V3{flat[FD::STRD + 0], flat[FD::STRD + 1], flat[FD::STRD + 2]};
  s.tvel = V3{flat[FD::TVEL + 0], flat[FD::TVEL + 1], flat[FD::TVEL + 2]};
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    s.q[i] = flat[FD::Q + i];
    s.qd[i] = flat[FD::QD + i];
  }
  s.adm0 = ll.adm.x;
  s.adm1 = ll.adm.y;
  s.stiff = ll.stiff.y;
  s.tf = V3{ll.tf0.x, ll.tf0.y, ll.tf1.x};
  s.org_dir = V3{ll.rot0.x, ll.rot0.y, ll.rot1.x};
  s.cur_dir = V3{ll.rot1.y, ll.rot2.x, ll.rot2.y};
  s.targ_dir = V3{ll.rot3.x, ll.rot3.y, ll.rot4.x};
}

// SPLIT: the two roles are two LAUNCHES (shc_cycle_half_kernel).  Admittance delta z and the published virtual stiffness share a plane: the walker
// half stores its 8 bytes of it (the stiffness), the model half the other 8 (in the two-wavefront resident kernel the walker hands the
// stiffness over at exit and the model wavefront stores the whole plane).
template <int NJ, unsigned F, int ROLE = ROLE_ALL, bool SPLIT = false>
__device__ __forceinline__ void store_leg(const LegRegs<NJ> &s, const LegOut &out, const Park &pk, const DevState &st, const CycleParams &P,
                                          uint32_t slot, unsigned dirty) {
  using FD = Fields<NJ>;
  using FT = Feat<F>;
  const LegPlanes ld{reinterpret_cast<double2 *>(st.legd), st.n_slots, slot};
  double flat[FD::CORE_END];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    flat[FD::Q + i] = s.q[i];
    flat[FD::QD + i] = s.qd[i];
  }
  flat[FD::TIP] = s.tip.x, flat[FD::TIP + 1] = s.tip.y, flat[FD::TIP + 2] = s.tip.z;
  flat[FD::TVEL] = s.tvel.x, flat[FD::TVEL + 1] = s.tvel.y, flat[FD::TVEL + 2] = s.tvel.z;
#pragma unroll
  for (int k = 0; k < PK_COUNT; ++k) flat[FD::SORG + k] = ROLE == ROLE_BACK ? 0.0 : pk.at(k);
  flat[FD::TARG] = s.targ.x, flat[FD::TARG + 1] = s.targ.y, flat[FD::TARG + 2] = s.targ.z;
  flat[FD::STRD] = s.strd.x, flat[FD::STRD + 1] = s.strd.y, flat[FD::STRD + 2] = s.strd.z;
  // swing origin position / velocity and stance origin / default tip change once per step period: their planes are
  // written back only when some lane of the wave changed them during this launch
  static_assert(FD::SORG % 2 == 0 && FD::TORG % 2 == 0 && FD::TARG % 2 == 0, "park groups must cover whole planes");
#pragma unroll
  for (int p = 0; p < FD::CORE_END / 2; ++p) {
    const bool swing_org = 2 * p >= FD::SORG && 2 * p < FD::TORG, stance_org = 2 * p >= FD::TORG && 2 * p < FD::TARG;
    if (swing_org && !(dirty & DIRTY_SWING_ORG)) continue;
    if (stance_org && !(dirty & DIRTY_STANCE_ORG)) continue;
    if ((ROLE == ROLE_FRONT && p < NJ) || (ROLE == ROLE_BACK && p >= NJ)) continue;
    ld.store(p, double2{flat[2 * p], flat[2 * p + 1]});
  }
  if (FT::adm(P) && ROLE != ROLE_FRONT && !(SPLIT && ROLE == ROLE_BACK)) { // (the two-wave kernel's model half receives s.stiff from the walker half before it stores;
    ld.store(FD::ADM / 2, double2{s.adm0, s.adm1});                       //  the model half of a two-LAUNCH cycle stored these right after updateAdmittance: store_admittance_early)
    ld.store(FD::ADM_DELTA / 2, double2{out.adm_delta.x, out.adm_delta.y});
    ld.store(FD::ADM_DELTA / 2 + 1, double2{out.adm_delta.z, s.stiff});
  }
  if (SPLIT && ROLE == ROLE_FRONT && FT::adm(P) && P.dynamic_stiffness) st.legd[leg_field_index(FD::ADM_DELTA + 3, slot, st.n_slots)] = s.stiff;
  if (FT::tipf(P) && ROLE != ROLE_FRONT) {
    ld.store(FD::TF / 2, double2{s.tf.x, s.tf.y});
    ld.store(FD::TF / 2 + 1, double2{s.tf.z, 0.0});
  }
  // LegState outputs: the model tip is FK(q) and, without per-leg auto poses, the poser tip is the walker tip seen from
  // Model::current_pose_ - both are derived from the stored state when a getter asks (derive_tips_kernel), not written
  // every launch.  Only the auto-pose path's per-leg pose is not recoverable, so it stores its poser tip.
  if (FT::autop(P) && !FT::imu(P) && ROLE != ROLE_BACK) {
    ld.store(FD::POSER_TIP / 2, double2{out.poser_tip.x, out.poser_tip.y});
    ld.store(FD::POSER_TIP / 2 + 1, double2{out.poser_tip.z, 0.0});
  }
  if (rot_enabled<NJ, F>() && ROLE != ROLE_BACK) {
    static_assert(FD::ORG_DIR % 2 == 0 && FD::CUR_DIR == FD::ORG_DIR + 3 && FD::TARG_DIR == FD::ORG_DIR + 6, "tip direction planes");
    ld.store(FD::ORG_DIR / 2, double2{s.org_dir.x, s.org_dir.y});
    ld.store(FD::ORG_DIR / 2 + 1, double2{s.org_dir.z, s.cur_dir.x});
    ld.store(FD::ORG_DIR / 2 + 2, double2{s.cur_dir.y, s.cur_dir.z});
    if (dirty & DIRTY_TARG_DIR) {
      ld.store(FD::ORG_DIR / 2 + 3, double2{s.targ_dir.x, s.targ_dir.y});
      ld.store(FD::ORG_DIR / 2 + 4, double2{s.targ_dir.z, 0.0});
    }
  }
  if (ROLE != ROLE_BACK) st.legi[slot] = s.word;
}

// The model half of a two-launch cycle (shc_cycle_half_kernel<..., ROLE_BACK>): admittance state and admittance delta are final once
// AdmittanceController::updateAdmittance has run, before Model::updateModel - stored there, their ten registers are free during the IK solves (the 8 x 5
// rotation-constrained model half missed its 256 registers by 21: 84 B of scratch per lane = a fifth of the cycle's algorithmic traffic again).
template <int NJ>
__device__ __forceinline__ void store_admittance_early(const LegRegs<NJ> &s, const LegOut &out, const DevState &st, uint32_t slot) {
  using FD = Fields<NJ>;
  const LegPlanes ld{reinterpret_cast<double2 *>(st.legd), st.n_slots, slot};
  ld.store(FD::ADM / 2, double2{s.adm0, s.adm1});
  ld.store(FD::ADM_DELTA / 2, double2{out.adm_delta.x, out.adm_delta.y});
  st.legd[leg_field_index(FD::ADM_DELTA + 2, slot, st.n_slots)] = out.adm_delta.z; // (the other 8 bytes of this element: the walker half's published stiffness)
}

// Robot state lives in HBM as one contiguous tile per wave, [wave][field][RPW] (AoSoA): staging fields [F0, F1) of this
// wave's robots to / from the LDS tile is a straight coalesced copy of (F1 - F0) * RPW doubles.
template <int RPW, int F0, int F1>
__device__ __forceinline__ void load_rob_fields(double (&reg)[((F1 - F0) * RPW + 63) / 64], const double *gtile, int lane) {
  constexpr int total = (F1 - F0) * RPW;
  constexpr int iters = (total + 63) / 64;
#pragma unroll
  for (int it = 0; it < iters; ++it) {
    int idx = it * 64 + lane;
    reg[it] = idx < total ? gtile[F0 * RPW + idx] : 0.0;
  }
}
template <int RPW, int F0, int F1>
__device__ __forceinline__ void put_rob_fields(const double (&reg)[((F1 - F0) * RPW + 63) / 64], double *tile, int lane) {
  constexpr int total = (F1 - F0) * RPW;
  constexpr int iters = (total + 63) / 64;
#pragma unroll
  for (int it = 0; it < iters; ++it) {
    int idx = it * 64 + lane;
    if (idx < total) tile[F0 * RPW + idx] = reg[it];
  }
}
template <int RPW, int F0, int F1>
__device__ __forceinline__ void store_rob_fields(const double *tile, double *gtile, int lane) {
  constexpr int total = (F1 - F0) * RPW;
  constexpr int iters = (total + 63) / 64;
#pragma unroll
  for (int it = 0; it < iters; ++it) {
    int idx = it * 64 + lane;
    if (idx < total) gtile[F0 * RPW + idx] = tile[F0 * RPW + idx];
  }
}

// ================================================================================================= resident mode
// StateController::loop's while-loop (src/main.cpp:106-131: loop(); publish...; spinOnce(); rate.sleep()) kept on the chip.
// Protocol (shc_cycle_launch.hpp): the relay wave mirrors the host's doorbell / stop words into ResidentCtl::gate and the
// workers' progress back to the host; a worker runs cycle c while c < min(doorbell, stop), takes what was posted for cycle c
// (header ring -> fresh input groups in the data rings), writes q / qd of the cycle to the output ring (write-through), and
// publishes "c cycles done" one half cycle later, when those stores have certainly drained.  Every wait is bounded.
typedef unsigned long long u64;
__device__ __forceinline__ u64 ld_agent(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_sys(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double ld_agent_f64(const double *p) { return __longlong_as_double((long long)ld_agent(reinterpret_cast<const u64 *>(p))); }
// a value every lane holds alike, moved to SGPRs so that branches on it are scalar branches
__device__ __forceinline__ u64 uni64(u64 v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
  return (u64(hi) << 32) | lo;
}

// per-leg inputs of a resident cycle: the paired planes that currently hold them (a ring position, or the engine's own planes while
// nothing has been posted), read with agent-scope loads - another agent may have written them while this kernel runs
template <int NJ>
struct LegInRing {
  // this lane's leg: element 2 p + h of its record lies at ptr + p * pair + h - the paired planes of the engine / the input rings
  // (pair = 2 n_slots) and a caller's instance-major array of a bound input set (pair = 2) alike
  const double *force_ptr = nullptr, *effort_ptr = nullptr;
  int64_t force_pair = 0, effort_pair = 0;
  // The joint efforts are consumed at the very end of Model::updateModel (Leg::calculateTipForce) but come from memory another agent
  // may have written (agent-scope loads, ~1 us): a wavefront that runs alone on its SIMD would sit out that latency.  prefetch_effort()
  // issues the loads where the model half starts; they arrive while the IK step runs.
  bool effort_prefetched = false;
  double effort_now[NJ] = {};
  // The batch form (shc_engine_step_k) reads arrays nobody writes while the launch runs: plain loads (L2 / L1 hits between neighbouring lanes and
  // loads), and BOTH per-leg groups are issued at the top of the cycle - a thousand clocks before the admittance update consumes the force
  bool coherent = true; // agent-scope loads (another agent may be writing: the doorbell-driven loop); false: plain loads
  bool force_prefetched = false;
  double force_now[3] = {};
  __device__ __forceinline__ double ld(const double *p) const { return coherent ? ld_agent_f64(p) : *p; }
  __device__ __forceinline__ void prefetch_effort() {
#pragma unroll
    for (int i = 0; i < NJ; ++i) effort_now[i] = ld(effort_ptr + int64_t(i / 2) * effort_pair + (i & 1));
    effort_prefetched = true;
  }
  __device__ __forceinline__ void prefetch_force() {
    force_now[0] = ld(force_ptr), force_now[1] = ld(force_ptr + 1), force_now[2] = ld(force_ptr + force_pair);
    force_prefetched = true;
  }
  __device__ __forceinline__ V3 force() const {
    if (force_prefetched) return V3{force_now[0], force_now[1], force_now[2]};
    return V3{ld(force_ptr), ld(force_ptr + 1), ld(force_ptr + force_pair)};
  }
  __device__ __forceinline__ void effort(double (&e)[NJ]) const {
    if (effort_prefetched) {
#pragma unroll
      for (int i = 0; i < NJ; ++i) e[i] = effort_now[i];
      return;
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) e[i] = ld(effort_ptr + int64_t(i / 2) * effort_pair + (i & 1));
  }
};
// (no dynamic indexing of the kernel-argument struct: that would move it to scratch)
__device__ __forceinline__ const double *bound_array(const ResidentArgs &A, int set, int which) {
  const double *const *row = set == 0 ? A.bound[0] : (set == 1 ? A.bound[1] : (set == 2 ? A.bound[2] : A.bound[3]));
  return which == BND_LIN ? row[BND_LIN] : which == BND_ANG ? row[BND_ANG] : which == BND_IMUQ ? row[BND_IMUQ] : which == BND_IMUW ? row[BND_IMUW]
       : which == BND_FORCE ? row[BND_FORCE] : row[BND_EFFORT];
}
// Where the per-leg inputs in force come from: src >= 0 a ring position, -1 the engine's own planes, <= -2 bound input set -2 - src
// (the caller's arrays [n][legs][3] / [n][legs][dof]).  All wave-uniform selects.
// row: the cycle's row of K-deep bound arrays (batch form: ResidentArgs::kstride doubles per row; 0 in the doorbell-driven loop)
template <int L, int NJ>
__device__ __forceinline__ LegInRing<NJ> leg_inputs_in_force(const ResidentArgs &A, const double *legd, const int src_force, const int src_effort, const int64_t ns,
                                                             const uint32_t slot, const int64_t robot, const int leg, const int64_t row = 0) {
  using FD = Fields<NJ>;
  LegInRing<NJ> in;
  const double *fplanes = src_force < 0 ? legd + int64_t(FD::FORCE_IN / 2) * ns * 2 : A.force + int64_t(src_force) * 2 * ns * 2;
  const double *eplanes = src_effort < 0 ? legd + int64_t(FD::EFFORT_IN / 2) * ns * 2 : A.effort + int64_t(src_effort) * (FD::NJE / 2) * ns * 2;
  in.force_ptr = fplanes + int64_t(slot) * 2, in.force_pair = ns * 2;
  in.effort_ptr = eplanes + int64_t(slot) * 2, in.effort_pair = ns * 2;
  if (src_force <= -2) in.force_ptr = bound_array(A, -2 - src_force, BND_FORCE) + row * A.kstride[BND_FORCE] + (robot * L + leg) * 3, in.force_pair = 2;
  if (src_effort <= -2) in.effort_ptr = bound_array(A, -2 - src_effort, BND_EFFORT) + row * A.kstride[BND_EFFORT] + (robot * L + leg) * NJ, in.effort_pair = 2;
  return in;
}

struct ResidentHeld { // what the worker remembers between the loop and the epilogue (all wave-uniform)
  int src_force = -1, src_effort = -1; // per-leg inputs in force: ring position; -1: the engine's own planes; <= -2: bound input set -2 - src
  unsigned seen = 0;                   // input groups that were posted during this run
  unsigned cycles = 0;
  bool fault = false;
};

template <int RPW, int NF>
__device__ __forceinline__ void ring_to_tile(const double *rec, double *tile_at, int lane) {
  constexpr int total = NF * RPW;
#pragma unroll
  for (int it = 0; it * 64 < total; ++it) {
    const int i = it * 64 + lane;
    if (i < total) tile_at[i] = ld_agent_f64(rec + i);
  }
}

// What was posted for cycle c (header words h0, h1; anything but tag c + 1 = nothing posted, inputs held): fresh robot inputs go
// to the LDS tile (ROBOT), the per-leg inputs are read where they lie - only their ring position is noted (LEG).
enum : int { ROBOT_NONE = 0, ROBOT_VEL = 1, ROBOT_POSE = 2, ROBOT_ALL = 3 }; // which per-robot input groups a wave copies into the tile
template <int RPW, int ROBOT, bool LEG>
__device__ __forceinline__ void resident_take_inputs(const ResidentArgs &A, const unsigned c, const u64 h0, const u64 h1, const int64_t wave, const int lane,
                                                     double *tile, int32_t *tile_i, unsigned &dirty, ResidentHeld &held, const int64_t n_robots, const int64_t row = 0) {
  using R = RobotFields;
  if (h0 != u64(c) + 1) return;
  const unsigned mask = unsigned(h1) & 0x7fffu;
  held.seen |= mask;
  if (unsigned(h1) & kResidentDirect) { // a direct post: the fresh groups come straight from the caller's arrays of a bound input set
    const int set = int((h1 >> 16) & 3);
    if (ROBOT != ROBOT_NONE) {
      if ((ROBOT & ROBOT_VEL) && (mask & (1u << RG_VEL))) { // [n][2], [n]
        static_assert(R::WIN == R::VIN + 2, "velocity inputs are contiguous in the tile");
        const double *lin = bound_array(A, set, BND_LIN) + row * A.kstride[BND_LIN], *ang = bound_array(A, set, BND_ANG) + row * A.kstride[BND_ANG];
        const int field = lane / RPW, r = lane - field * RPW;
        const int64_t rob = wave * RPW + r;
        if (lane < 3 * RPW && rob < n_robots) tile[(R::VIN + field) * RPW + r] = field < 2 ? ld_agent_f64(lin + rob * 2 + field) : ld_agent_f64(ang + rob);
      }
      if ((ROBOT & ROBOT_POSE) && (mask & (1u << RG_IMU))) { // [n][4], [n][3]
        const int64_t rob = wave * RPW + lane;
        if (lane < RPW && rob < n_robots) { // Model::setImuData as shc_engine_set_imu stores it: the orientation normalised
          const double *q = bound_array(A, set, BND_IMUQ) + row * A.kstride[BND_IMUQ] + rob * 4, *w = bound_array(A, set, BND_IMUW) + row * A.kstride[BND_IMUW] + rob * 3;
          const Quat qn = normalized(Quat{ld_agent_f64(q), ld_agent_f64(q + 1), ld_agent_f64(q + 2), ld_agent_f64(q + 3)});
          tile[(R::IMUQ + 0) * RPW + lane] = qn.w, tile[(R::IMUQ + 1) * RPW + lane] = qn.x, tile[(R::IMUQ + 2) * RPW + lane] = qn.y, tile[(R::IMUQ + 3) * RPW + lane] = qn.z;
#pragma unroll
          for (int k = 0; k < 3; ++k) tile[(R::GYRO + k) * RPW + lane] = ld_agent_f64(w + k);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (LEG) { // per-leg inputs are read where they lie, from now on in the set's arrays
      if (mask & (1u << RG_FORCE)) held.src_force = -2 - set;
      if (mask & (1u << RG_EFFORT)) held.src_effort = -2 - set;
    }
    return;
  }
  auto pos = [&](int grp) { return int((h1 >> (16 + 8 * grp)) & 0xff); };
  if (ROBOT != ROBOT_NONE) {
    if ((ROBOT & ROBOT_VEL) && (mask & (1u << RG_VEL)))
      ring_to_tile<RPW, 3>(A.rin + ((int64_t(pos(RG_VEL)) * A.n_waves + wave) * RIN_COUNT + RIN_VEL) * RPW, tile + R::VIN * RPW, lane);
    if ((ROBOT & ROBOT_POSE) && (mask & (1u << RG_IMU))) {
      const double *rec = A.rin + ((int64_t(pos(RG_IMU)) * A.n_waves + wave) * RIN_COUNT + RIN_IMU) * RPW;
      ring_to_tile<RPW, 4>(rec, tile + R::IMUQ * RPW, lane);
      ring_to_tile<RPW, 3>(rec + 4 * RPW, tile + R::GYRO * RPW, lane);
    }
    if ((ROBOT & ROBOT_POSE) && (mask & (1u << RG_POSE))) {
      static_assert(R::RVI == R::TVI + 3, "pose inputs are contiguous in the tile");
      ring_to_tile<RPW, 6>(A.rin + ((int64_t(pos(RG_POSE)) * A.n_waves + wave) * RIN_COUNT + RIN_POSE) * RPW, tile + R::TVI * RPW, lane);
      dirty |= DIRTY_MANUAL;
    }
    if ((ROBOT & ROBOT_POSE) && (mask & (1u << RG_RESET))) {
      if (lane < RPW)
        tile_i[R::I_RESET_MODE * RPW + lane] = int(unsigned(ld_agent(reinterpret_cast<const u64 *>(A.rini) + ((int64_t(pos(RG_RESET)) * A.n_waves + wave) * RPW + lane))));
      dirty |= DIRTY_MANUAL;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier(); // LDS operations of one wave complete in order: the tile now holds the new inputs
  }
  if (LEG) {
    if (mask & (1u << RG_FORCE)) held.src_force = pos(RG_FORCE);
    if (mask & (1u << RG_EFFORT)) held.src_effort = pos(RG_EFFORT);
  }
}

template <int L, int NJ, unsigned F, bool BATCH = false>
__device__ __forceinline__ void resident_loop(const ResidentArgs &A, const DevState &st, LegRegs<NJ> &s, LegOut &out, const SharedConsts<L, NJ> &C,
                                              const RobTile<64 / L> &rb, const Park &pk, const Group<L> g, const int leg, const uint32_t slot,
                                              const int lane, const int64_t wave, const bool live, double *tile, int32_t *tile_i, unsigned &dirty,
                                              const bool manual_live, ResidentHeld &held, const bool touchdown_at_begin, double *ext) {
  using R = RobotFields;
  using FD = Fields<NJ>;
  constexpr int RPW = 64 / L;
  const int64_t ns = st.n_slots;
  const unsigned out_slot_bytes = unsigned(NJ * ns * 16); // q, qd = fields [0, 2 NJ) = NJ paired planes
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, int(unsigned(A.depth) * out_slot_bytes), 0x00020000);
  const u64 emergency_ticks = 4 * A.idle_ticks + 2000 * A.ticks_per_ms; // a worker never waits longer than this for the relay (2 s + 4 idle timeouts)
  unsigned c = 0, oslot = 0; // cycles completed; output ring position of cycle c
  constexpr bool batch = BATCH; // the batch form (shc_engine_step_k) is a kernel of its own: K cycles, released from the start, inputs row c of K-deep arrays -
                                // compiled without the handshake, and the doorbell-driven loop without the batch form's prefetches (scalar registers are scarce in both)
  u64 gate = 0;
  if constexpr (batch) gate = (u64(A.batch_cycles) << 32) | A.batch_cycles;
  else gate = uni64(ld_agent(&A.ctl->gate));
  u64 h0 = 0, h1 = 0;
  bool hdr_valid = false;
  // Batch form: the per-robot inputs of row `r` - velocity command (3 doubles per robot), IMU sample (gyro 3 + orientation 4) - as one load per lane
  // and group (lane = field x robots-per-wave + robot), issued a whole cycle before batch_rob_commit() puts them into the LDS tile.
  static_assert(R::WIN == R::VIN + 2 && R::IMUQ == R::GYRO + 3, "velocity inputs / gyro + orientation are contiguous in the tile");
  constexpr int NI = (7 * RPW + 63) / 64; // loads per lane that cover the 7 IMU fields of the wave's robots
  double rob_vel = 0.0, rob_imu[NI] = {};
  const auto batch_rob_issue = [&](const int64_t r) {
    const int64_t rob0 = wave * RPW;
    if (A.batch_mask & (1u << RG_VEL)) { // [K][n][2], [K][n]
      const int field = lane / RPW, rr = lane - field * RPW;
      const int64_t rob = rob0 + rr;
      const double *lin = bound_array(A, 0, BND_LIN) + r * A.kstride[BND_LIN], *ang = bound_array(A, 0, BND_ANG) + r * A.kstride[BND_ANG];
      if (lane < 3 * RPW && rob < st.n_robots) rob_vel = field < 2 ? lin[rob * 2 + field] : ang[rob];
    }
    if (A.batch_mask & (1u << RG_IMU)) { // [K][n][3] gyro -> tile fields GYRO .. + 2, [K][n][4] orientation -> IMUQ .. + 3 (normalised at commit)
      const double *q = bound_array(A, 0, BND_IMUQ) + r * A.kstride[BND_IMUQ], *w = bound_array(A, 0, BND_IMUW) + r * A.kstride[BND_IMUW];
#pragma unroll
      for (int h = 0; h < NI; ++h) {
        const int idx = lane + 64 * h, field = idx / RPW, rr = idx - field * RPW;
        const int64_t rob = rob0 + rr;
        if (field < 7 && rob < st.n_robots) rob_imu[h] = field < 3 ? w[rob * 3 + field] : q[rob * 4 + (field - 3)];
      }
    }
  };
  const auto batch_rob_commit = [&]() {
    if (A.batch_mask & (1u << RG_VEL)) {
      const int field = lane / RPW, rr = lane - field * RPW;
      if (lane < 3 * RPW && wave * RPW + rr < st.n_robots) tile[(R::VIN + field) * RPW + rr] = rob_vel;
    }
    if (A.batch_mask & (1u << RG_IMU)) {
#pragma unroll
      for (int h = 0; h < NI; ++h) {
        const int idx = lane + 64 * h, field = idx / RPW, rr = idx - field * RPW;
        if (field < 7 && wave * RPW + rr < st.n_robots) tile[(R::GYRO + field) * RPW + rr] = rob_imu[h];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < RPW && wave * RPW + lane < st.n_robots) { // Model::setImuData as shc_engine_set_imu stores it: the orientation normalised
        const Quat qn = normalized(Quat{tile[(R::IMUQ + 0) * RPW + lane], tile[(R::IMUQ + 1) * RPW + lane], tile[(R::IMUQ + 2) * RPW + lane], tile[(R::IMUQ + 3) * RPW + lane]});
        tile[(R::IMUQ + 0) * RPW + lane] = qn.w, tile[(R::IMUQ + 1) * RPW + lane] = qn.x, tile[(R::IMUQ + 2) * RPW + lane] = qn.y, tile[(R::IMUQ + 3) * RPW + lane] = qn.z;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  if constexpr (batch) batch_rob_issue(0);
  for (;;) {
    unsigned db = unsigned(gate), sp = unsigned(gate >> 32);
    if constexpr (batch) {
      if (c >= sp) break;
      h0 = u64(c) + 1, h1 = u64(A.batch_mask) | kResidentDirect; // what a direct post of set 0 would have left in the header ring
      hdr_valid = true;
    }
    if (!batch && !(c < db && c < sp)) {
      if (c >= sp) break;
      // nothing to run yet: the outputs of cycle c - 1 would otherwise be announced half a cycle into cycle c - announce them now
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) st_agent(A.progress + wave, u64(c));
      const u64 t0 = wall_clock64();
      for (;;) {
        __builtin_amdgcn_s_sleep(4);
        gate = uni64(ld_agent(&A.ctl->gate));
        db = unsigned(gate), sp = unsigned(gate >> 32);
        if (c >= sp || c < db) break;
        if (wall_clock64() - t0 > emergency_ticks) {
          held.fault = true;
          break;
        }
      }
      if (held.fault || c >= sp) break;
      hdr_valid = false;
    }
    // what was posted for cycle c
    if (!hdr_valid) {
      const u64 *hp = reinterpret_cast<const u64 *>(A.headers + (c & (kResidentHeaders - 1)));
      h0 = uni64(ld_agent(hp));
      h1 = uni64(ld_agent(hp + 1));
    }
    // prefetch for the next iteration: the gate, and - once the doorbell is known to cover it - the header of cycle c + 1
    u64 gate_next_v = gate;
    if constexpr (!batch) gate_next_v = ld_agent(&A.ctl->gate);
    const bool next_valid = !batch && db > c + 1;
    u64 n0v = 0, n1v = 0;
    if (next_valid) {
      const u64 *hp = reinterpret_cast<const u64 *>(A.headers + ((c + 1) & (kResidentHeaders - 1)));
      n0v = ld_agent(hp);
      n1v = ld_agent(hp + 1);
    }
    const int64_t row = batch ? int64_t(c) : 0;
    if constexpr (batch) { // row c of the K-deep arrays: the robot inputs were loaded one cycle ahead, the per-leg ones are read where they lie
      held.seen |= A.batch_mask;
      if (A.batch_mask & (1u << RG_FORCE)) held.src_force = -2;
      if (A.batch_mask & (1u << RG_EFFORT)) held.src_effort = -2;
      batch_rob_commit();
      if (c + 1 < A.batch_cycles) batch_rob_issue(int64_t(c) + 1); // in flight for the whole of cycle c
    } else {
      resident_take_inputs<64 / L, ROBOT_ALL, true>(A, c, h0, h1, wave, lane, tile, tile_i, dirty, held, st.n_robots, row);
    }
    LegInRing<NJ> in = leg_inputs_in_force<L, NJ>(A, st.legd, held.src_force, held.src_effort, ns, slot, int64_t(slot / 64) * RPW + (slot % 64) / L, leg, row);
    if constexpr (batch) {
      in.coherent = false;
      if (Feat<F>::adm(C.P) || (F & F_ROUGH) != 0) in.prefetch_force();
      if (Feat<F>::tipf(C.P)) in.prefetch_effort();
    }
    // half a cycle after the output stores of cycle c - 1 were issued they have drained: publish "c cycles done"
    const auto publish_previous = [&]() {
      if constexpr (batch) return; // (nobody watches the progress of a batch launch: the stream does)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) st_agent(A.progress + wave, u64(c));
    };
    bool touchdown_detection = false;
    if constexpr ((F & F_ROUGH) != 0) {
      // tipStatesCallback (state_controller.cpp:1618-1648) delivered a force this iteration: Leg::touchdownDetection (model.cpp:712-722) with the
      // tip where the last applyFK left it - what shc_engine_set_tip_force runs as a kernel of its own between two launches
      touchdown_detection = touchdown_at_begin || (held.seen & (1u << RG_FORCE)) != 0; // LegStepper::setTouchdownDetection(true) (:1642)
      const bool fresh_force = h0 == u64(c) + 1 && (unsigned(h1) & (1u << RG_FORCE)) != 0;
      if (fresh_force && uni(C.P.rough_terrain) != 0) {
        double *sp = st.legd + leg_field_index(FD::STEP_PLANE, slot, ns); // fields STEP_PLANE .. + 3 = two paired planes
        double *sp2 = st.legd + leg_field_index(FD::STEP_PLANE + 2, slot, ns);
        const double fn = norm(in.force());
        const double defined = ld_agent_f64(sp2 + 1);
        if (fn > A.touchdown_threshold && defined == 0.0) {
          Chain<NJ> ch;
          fk_chain<NJ>(C.leg[leg], s.q, ch);
          const V3 tipnow = tip_robot_frame(C.leg[leg], ch.pe); // step_plane_pose_ = current_tip_pose_
          if (live) sp[0] = tipnow.x, sp[1] = tipnow.y, sp2[0] = tipnow.z, sp2[1] = 1.0;
        } else if (fn < A.liftoff_threshold) {
          if (live) sp2[1] = 0.0;
        }
      }
    }
    cycle<L, NJ, F>(s, out, C, rb, pk, g, leg, st.legd, ns, slot, dirty, manual_live, touchdown_detection, ext, nullptr, in, publish_previous,
                    (F & F_ROUGH) != 0 ? st.span : nullptr);
    { // desired joint state of this cycle -> output ring (write-through 16-byte stores: visible to any agent once drained)
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      double flat[2 * NJ];
#pragma unroll
      for (int i = 0; i < NJ; ++i) flat[FD::Q + i] = s.q[i], flat[FD::QD + i] = s.qd[i];
      const unsigned soff = oslot * out_slot_bytes;
      if (live) {
#pragma unroll
        for (int p = 0; p < NJ; ++p) {
          const u64 a = u64(__double_as_longlong(flat[2 * p])), b = u64(__double_as_longlong(flat[2 * p + 1]));
          const v4u w = {unsigned(a), unsigned(a >> 32), unsigned(b), unsigned(b >> 32)};
          __builtin_amdgcn_raw_buffer_store_b128(w, out_rsrc, unsigned((int64_t(p) * ns + slot) * 16), soff, 16 /* sc1 */);
        }
      }
    }
    ++c;
    oslot = oslot + 1 == unsigned(A.depth) ? 0 : oslot + 1;
    gate = uni64(gate_next_v);
    hdr_valid = next_valid;
    if (next_valid) h0 = uni64(n0v), h1 = uni64(n1v);
  }
  if constexpr (!batch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) st_agent(A.progress + wave, u64(c));
  }
  held.cycles = c;
}

// The per-leg inputs in force when the loop ends (a ring position or a bound input set) are carried into the engine's own planes.
template <int L, int NJ>
__device__ __forceinline__ void carry_leg_inputs(const ResidentArgs &A, const DevState &st, const ResidentHeld &held, const int64_t ns, const uint32_t slot, const int leg) {
  using FD = Fields<NJ>;
  constexpr int RPW = 64 / L;
  double2 *planes = reinterpret_cast<double2 *>(st.legd);
  const LegInRing<NJ> in = leg_inputs_in_force<L, NJ>(A, st.legd, held.src_force, held.src_effort, ns, slot, int64_t(slot / 64) * RPW + (slot % 64) / L, leg,
                                                      A.batch_cycles ? int64_t(A.batch_cycles) - 1 : 0); // (batch form: the last cycle's row stays in force)
  if (held.src_force != -1) {
    const V3 f = in.force();
    planes[(FD::FORCE_IN / 2) * ns + slot] = double2{f.x, f.y};
    planes[(FD::FORCE_IN / 2 + 1) * ns + slot] = double2{f.z, 0.0};
  }
  if (held.src_effort != -1) {
    double ef[NJ];
    in.effort(ef);
#pragma unroll
    for (int p = 0; p < FD::NJE / 2; ++p) planes[(FD::EFFORT_IN / 2 + p) * ns + slot] = double2{ef[2 * p], 2 * p + 1 < NJ ? ef[2 * p + 1] : 0.0};
  }
}

// After the loop: the inputs the run received become the engine's held inputs (the next ordinary launch reads them from the
// state planes), and the wave reports that it has left.
template <int L, int NJ, unsigned F>
__device__ __forceinline__ void resident_epilogue(const ResidentArgs &A, const DevState &st, const uint32_t slot, const int lane, const bool live,
                                                  const double *tile, const int32_t *tile_i, double *gtile, int32_t *gtile_i, const ResidentHeld &held) {
  using R = RobotFields;
  using FD = Fields<NJ>;
  constexpr int RPW = 64 / L;
  const int64_t ns = st.n_slots;
  store_rob_fields<RPW, R::VIN, R::CORE_END>(tile, gtile, lane);
  if (held.seen & (1u << RG_IMU)) {
    store_rob_fields<RPW, R::GYRO, R::IMU_END>(tile, gtile, lane);
    store_rob_fields<RPW, R::IMUQ, R::IMUQ_END>(tile, gtile, lane);
  }
  if ((held.seen & (1u << RG_RESET)) && lane < RPW) gtile_i[R::I_RESET_MODE * RPW + lane] = tile_i[R::I_RESET_MODE * RPW + lane];
  if (live) carry_leg_inputs<L, NJ>(A, st, held, ns, slot, int(slot % 64) % L);
  if (lane == 0 && A.batch_cycles == 0) {
    if (held.fault) __hip_atomic_fetch_or(&A.ctl->fault, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&A.ctl->exited, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The relay: between the host and the workers.  It alone reads / writes the host-mapped words (a handful of PCIe
// transactions per microsecond instead of one per worker and cycle), it alone writes the gate - so "doorbell" and "stop" change
// together, atomically, and every worker leaves the loop at the same cycle - and it alone decides to stop: on request, at the
// launch's cycle bound, or when the doorbell has not moved for idle_ticks (a host that went away cannot leave the GPU spinning).
// Two directions, PART_GATE (host doorbell / stop -> gate) and PART_DONE (workers' progress -> host, exit report): one wave
// does both (PART_BOTH), or - where the relay workgroup has waves to spare - one wave each, which halves the latency either way.
enum : int { PART_BOTH = 0, PART_GATE = 1, PART_DONE = 2 };
template <int PART>
__device__ __forceinline__ void resident_relay(const ResidentArgs &A) {
  const int lane = threadIdx.x & 63;
  const unsigned max_cycles = A.max_cycles;
  unsigned db = 0, sp = max_cycles, idle_stop = 0xffffffffu;
  u64 reason = RESIDENT_EXIT_MAX, last_reason = 0, last_gate = ~0ull, last_done = 0, iter = 0;
  u64 t_last = wall_clock64(), t_all_done = 0;
  for (;;) {
    const u64 now = wall_clock64();
    // every load of the iteration is issued before the first one is waited for (each is a trip to memory or across PCIe)
    const u64 exited_v = ld_agent(&A.ctl->exited);
    const u64 gate_v = PART == PART_DONE ? ld_agent(&A.ctl->gate) : 0;
    const u64 done_v = PART != PART_DONE ? ld_agent(&A.ctl->pad[1]) : 0; // cycles completed by every wave (the idle clock only runs when nothing is left to run)
    u64 pv[8];
    if (PART != PART_GATE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = j * 64 + lane < A.n_waves ? ld_agent(A.progress + j * 64 + lane) : ~0ull;
    }
    if (PART != PART_DONE) {
      const u64 hd_v = ld_sys(&A.host->doorbell), hs_v = ld_sys(&A.host->stop);
      // the direct-post record of the next unreleased cycle (two words, lanes 0 and 1), read with the doorbell
      const u64 rec_v = lane < 2 ? ld_sys(&A.host->records[size_t(db & (kResidentHeaders - 1)) * 2 + lane]) : 0;
      const u64 hd = uni64(hd_v), hs = uni64(hs_v);
      unsigned want_db = hd > max_cycles ? max_cycles : unsigned(hd);
      if (want_db < db) want_db = db; // the doorbell only moves forward
      if (want_db == db && db < max_cycles) { // nothing released by the doorbell: has cycle db been posted directly?
        const u64 tag = u64(db) + 1;
        const bool word_ok = lane == 0 ? rec_v == tag : (lane == 1 ? (rec_v >> 48) == (tag & 0xffffull) : true);
        if (__all(word_ok)) { // a complete record: it becomes the cycle's header, then the cycle is released
          const u64 w1 = uni64(__shfl(rec_v, 1, 64));
          if (lane == 0) {
            u64 *hp = reinterpret_cast<u64 *>(A.headers + (db & (kResidentHeaders - 1)));
            st_agent(hp + 1, (w1 & 0x7fffull) | u64(kResidentDirect) | (w1 & 0x30000ull));
            st_agent(hp, tag);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the header is out before the gate that lets a worker read it
          want_db = db + 1;
        }
      }
      // idle = the doorbell has not moved AND everything it released has run: a host that published a long burst is not idle while the
      // burst is still running (400 000 cycles released at once take 1.2 s)
      if (want_db != db || uni64(done_v) < u64(db)) t_last = now;
      else if (idle_stop == 0xffffffffu && now - t_last > A.idle_ticks) idle_stop = db;
      unsigned want_sp = max_cycles;
      u64 why = RESIDENT_EXIT_MAX;
      if (hs <= max_cycles) want_sp = unsigned(hs), why = RESIDENT_EXIT_STOP;
      if (idle_stop < want_sp) want_sp = idle_stop, why = RESIDENT_EXIT_IDLE;
      if (want_sp < db) want_sp = db; // cycles already released run
      if (want_db > want_sp) want_db = want_sp;
      db = want_db, sp = want_sp, reason = why;
      if (reason != last_reason) { // (the reason is in place before a gate that can end the loop)
        if (lane == 0) st_agent(&A.ctl->pad[0], reason);
        last_reason = reason;
      }
      const u64 gate = (u64(sp) << 32) | db;
      if (gate != last_gate) {
        if (lane == 0) st_agent(&A.ctl->gate, gate);
        last_gate = gate;
      }
    }
    const u64 exited = uni64(exited_v);
    bool leave = exited == u64(A.n_waves);
    if (PART != PART_GATE) {
      if (PART == PART_DONE) sp = unsigned(uni64(gate_v) >> 32);
      // cycles completed by every worker wave
      u64 m = ~0ull;
#pragma unroll
      for (int j = 0; j < 8; ++j) m = pv[j] < m ? pv[j] : m;
      for (int64_t base = 512; base < A.n_waves; base += 512) { // (more than 512 waves: 8 independent loads per lane in flight at a time)
        u64 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t w = base + j * 64 + lane;
          v[j] = w < A.n_waves ? ld_agent(A.progress + w) : ~0ull;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) m = v[j] < m ? v[j] : m;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_xor(m, off, 64);
        m = o < m ? o : m;
      }
      m = uni64(m);
      if (m != last_done) {
        if (lane == 0) {
          st_sys(&A.host->done, m);
          st_agent(&A.ctl->pad[1], m); // the device's copy: what a stream-ordered read polls (no PCIe traffic next to the doorbell)
        }
        last_done = m;
      }
      bool gave_up = false;
      if (!leave && m >= sp) { // everything that will ever run has run: the workers are on their way out
        if (t_all_done == 0) t_all_done = now;
        else if (now - t_all_done > 5000 * A.ticks_per_ms) leave = gave_up = true; // 5 s: give up on them
      }
      if (leave) {
        const u64 fault = uni64(ld_agent(&A.ctl->fault));
        const u64 why = uni64(ld_agent(&A.ctl->pad[0]));
        if (lane == 0) {
          st_sys(&A.host->fault, fault);
          st_sys(&A.host->done, m);
          st_agent(&A.ctl->pad[1], m);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          st_agent(&A.ctl->pad[2], 1ull);
          st_sys(&A.host->exited, (fault || gave_up) ? u64(RESIDENT_EXIT_FAULT) : why);
        }
      }
      if ((++iter & 1023) == 0 && lane == 0) st_sys(&A.host->heartbeat, iter);
    }
    if (leave) break;
    __builtin_amdgcn_s_sleep(1);
  }
}

// One launch = n_cycles control cycles of every robot; per-leg state stays in registers, per-robot state in LDS.
// The work of one wavefront: load its robots, run cycles, store them.  RES = false: n_cycles cycles with the inputs held (one
// ordinary launch).  RES = true: the resident loop (resident_loop below) - cycles run as the doorbell allows, inputs from the
// rings, outputs to the ring, until the relay says stop.
// HALF (launch mode only): ROLE_ALL = whole cycles.  ROLE_FRONT / ROLE_BACK = ONE cycle as two launches (shc_cycle_half_kernel): the walker /
// poser half (cycle_front; owns the stepper planes, the leg word, the tip directions and the robot tile; reads the joint angles for the FK tip
// rotation Leg::current_tip_pose_ holds) and the model half (cycle_back; owns the joint planes and the tip-force filter, ORs LW_IKFAIL into
// the word).  Nothing is handed over beyond the state itself: the model half redoes PoseController::updateStance's two transforms (poser tip =
// Model::current_pose_^-1 * walker tip, desired tip direction = pose rotation^-1 * walker tip direction, pose_controller.cpp:122-131) from the walker
// tip, the walker tip direction and the current pose the walker half has just stored - 64 bytes read per leg instead of 48 written + 48 read.
template <int L, int NJ, unsigned F, bool RES, int HALF = ROLE_ALL, bool BATCH = false>
__device__ __forceinline__ void cycle_wave(const DevState &st, const SharedConsts<L, NJ> *gc, int n_cycles, unsigned rt_flags, const int64_t wave,
                                           const ResidentArgs *ra) {
  static_assert(HALF == ROLE_ALL || (!RES && (F & (F_DYN | F_TERRAIN | F_MLEGS | F_AUTO)) == 0), "half-step launches: feature-exact kernels without auto posing / terrain paths");
  using R = RobotFields;
  using FT = Feat<F>;
  constexpr int RPW = 64 / L; // robots per wavefront
  __shared__ SharedConsts<L, NJ> C;
  // per-wave LDS (robot tile, its int words, park strip) is dynamic: sized for the workgroup actually launched (1 wave per
  // workgroup for small batches, 4 for large ones), cycle_lds_bytes_per_wave() each
  extern __shared__ double wave_lds[];
  constexpr int kWaveDoubles = R::COUNT * RPW + PK_COUNT * 64 + (R::I_COUNT * RPW + 1) / 2;
  SHC_TICK(0);
  // Launch-uniform run-time facts the host knows (kernel argument = SGPR from the first instruction on, no load to wait for):
  // RT_MANUAL_LIVE - some pose input / reset mode / injected state has ever been given to this engine.  Until then every
  // robot's manual pose is the identity and stays it, so the manual-pose group of the robot tile is neither loaded nor
  // evaluated nor stored.
  const bool manual_live = (rt_flags & RT_MANUAL_LIVE) != 0;
  const bool touchdown_detection = (rt_flags & RT_TOUCHDOWN) != 0;
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int64_t rob0 = wave * RPW;
  const int64_t left = st.n_robots - rob0;
  const int robots_here = left < RPW ? (left < 0 ? 0 : int(left)) : RPW;
  // Lanes without a robot of their own (the 64 % L tail lanes and the groups past the end of the batch) mirror a
  // live lane of the same leg in this wave so that every shuffle stays well defined; they never store.
  int grp = lane / L;
  const int leg = lane - grp * L;
  const bool live = grp < robots_here;
  if (grp >= robots_here) grp = robots_here > 0 ? robots_here - 1 : 0;
  const uint32_t slot = uint32_t(wave * 64 + grp * L + leg);
  double *const my_lds = wave_lds + wib * kWaveDoubles;
  Park pk{my_lds + R::COUNT * RPW, lane};
  LegRegs<NJ> s;
  const CycleParams &GP = gc->P; // feature flags of the generic specialisation: read from HBM before the LDS copy lands
  double *tile = my_lds;
  int32_t *tile_i = reinterpret_cast<int32_t *>(my_lds + R::COUNT * RPW + PK_COUNT * 64);
  double *gtile = st.robd + wave * (R::COUNT * RPW);
  int32_t *gtile_i = st.robi + wave * (R::I_COUNT * RPW);
  // ---- prologue: every global load of this wave is issued before the first wait, so the HBM / L2 latencies overlap:
  //      (1) launch-uniform tables, (2) this wave's robot tile, (3) per-leg state; then the LDS writes; then one barrier.
  LegLoad<NJ> ll;
  double th[NJ]; // DH joint offsets of this lane's leg straight from the table in HBM: the FK of the stored joint state
                 // (sin / cos) then starts as soon as the joint planes arrive, under the latency of the remaining loads
  const bool any_robot = robots_here > 0;
  double2 hand[4] = {double2{0.0, 0.0}, double2{0.0, 0.0}, double2{0.0, 0.0}, double2{0.0, 0.0}}; // (ROLE_BACK) walker tip planes, walker tip direction planes
  double hpose[7] = {0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0};                                            // (ROLE_BACK) Model::current_pose_ of this lane's robot
  double2 qpl[(NJ + 1) / 2];                                                      // (ROLE_FRONT) the joint-angle planes
  int word_back = 0;
  if (any_robot) {
#pragma unroll
    for (int k = 0; k < NJ; ++k) th[k] = gc->leg[leg].link_th[k];
    load_leg_issue<NJ, F, HALF>(ll, st, GP, slot);
    if constexpr (HALF == ROLE_FRONT) {
#pragma unroll
      for (int p = 0; p < (NJ + 1) / 2; ++p) qpl[p] = reinterpret_cast<const double2 *>(st.legd)[p * st.n_slots + slot];
    }
    if constexpr (HALF == ROLE_BACK) {
      word_back = st.legi[slot];
      using FDh = Fields<NJ>;
      static_assert(FDh::TIP % 2 == 0 && FDh::CUR_DIR == FDh::ORG_DIR + 3 && FDh::ORG_DIR % 2 == 0, "plane layout of the walker tip / tip direction");
      const double2 *planes = reinterpret_cast<const double2 *>(st.legd);
      hand[0] = planes[(FDh::TIP / 2) * st.n_slots + slot];          // tip x, y
      hand[1] = planes[(FDh::TIP / 2 + 1) * st.n_slots + slot];      // tip z | -
      hand[2] = planes[(FDh::ORG_DIR / 2 + 1) * st.n_slots + slot];  // - | current direction x
      hand[3] = planes[(FDh::ORG_DIR / 2 + 2) * st.n_slots + slot];  // current direction y, z
#pragma unroll
      for (int k = 0; k < 7; ++k) hpose[k] = gtile[(R::CPOSE + k) * RPW + grp];
    }
  }
  using SC = SharedConsts<L, NJ>;
  static_assert(sizeof(SC) % 16 == 0 && (offsetof(SC, P) + offsetof(CycleParams, ap_start)) % 16 == 0, "const block is copied in 16-byte words");
  constexpr int n16_all = sizeof(SC) / 16;
  constexpr int n16_core = (offsetof(SC, P) + offsetof(CycleParams, ap_start)) / 16; // without the auto-pose tables
  const int n16 = FT::autop(GP) ? n16_all : n16_core;
  constexpr int citers = (n16_all + 63) / 64; // enough for a 64-thread workgroup
  double2 creg[citers];
  {
    const double2 *src = reinterpret_cast<const double2 *>(gc);
    const int nt = blockDim.x;
#pragma unroll
    for (int it = 0; it < citers; ++it) {
      int i = it * nt + threadIdx.x;
      creg[it] = i < n16 ? src[i] : double2{0.0, 0.0};
    }
  }
  double t_core[(R::CORE_END * RPW + 63) / 64], t_man[((R::MANUAL_END - R::MPOSE) * RPW + 63) / 64],
      t_imu[((R::IMU_END - R::ABSE) * RPW + 63) / 64], t_imuq[((R::IMUQ_END - R::IMUQ) * RPW + 63) / 64],
      t_aprev[((R::APREV_END - R::APREV) * RPW + 63) / 64], t_align[((R::COUNT - R::TALIGN) * RPW + 63) / 64], t_odom[((R::ODOM_END - R::ODOM) * RPW + 63) / 64],
      t_cpose[((R::CPOSE_END - R::CPOSE) * RPW + 63) / 64], t_incl[((R::INCL_END - R::INCL) * RPW + 63) / 64];
  // RT_SKIP_MARKED - this launch follows a loop-level kernel (leg toggle, plan execution) that has already run the loop of the robots it
  // marked (ManualRobot::skip_cycle): they are left exactly as they are.  Model::current_pose_ is otherwise output only; here it
  // is loaded too so that the tile write-back is the identity for a skipped robot.
  const bool skip_marked = (F & F_MLEGS) != 0 && (rt_flags & RT_SKIP_MARKED) != 0 && (rt_flags & RT_MANUAL_LEGS) != 0;
  // RT_POSE_MARKED - the launch that PRECEDES such a loop-level kernel when the body pose has time-dependent parts (IMU / auto / inclination
  // posing): the marked robots run the posing part of their loop here (PoseController::updateCurrentPose, the admittance update:
  // state_controller.cpp:165-181) and nothing else; every other robot is left exactly as it is.
  const bool pose_marked = (F & F_MLEGS) != 0 && (rt_flags & RT_POSE_MARKED) != 0 && (rt_flags & RT_MANUAL_LEGS) != 0;
  constexpr int int_iters = (R::I_COUNT * RPW + 63) / 64; // 3-legged robots: 21 per wave x 4 ints = 84 entries > one wave's width
  int32_t t_int[int_iters];
  if (any_robot && HALF == ROLE_BACK) { // the model half reads nothing of the robot tile
#pragma unroll
    for (int k = 0; k < NJ; ++k) sincos_joint(th[k] + ll.flat[Fields<NJ>::Q + k], &s.sn[k], &s.cs[k]);
    load_leg_finish<NJ, HALF>(s, pk, ll);
    s.word = word_back;
  }
  if (any_robot && HALF != ROLE_BACK) {
    load_rob_fields<RPW, 0, R::CORE_END>(t_core, gtile, lane);
    if (FT::manual(GP) && manual_live) load_rob_fields<RPW, R::MPOSE, R::MANUAL_END>(t_man, gtile, lane);
    if (FT::imu(GP)) load_rob_fields<RPW, R::ABSE, R::IMU_END>(t_imu, gtile, lane);
    if (FT::imu(GP) || FT::incl(GP) || FT::autop(GP)) load_rob_fields<RPW, R::IMUQ, R::IMUQ_END>(t_imuq, gtile, lane);
    if (FT::incl(GP) && FT::autop(GP)) load_rob_fields<RPW, R::APREV, R::APREV_END>(t_aprev, gtile, lane);
    if (FT::odom(GP)) load_rob_fields<RPW, R::ODOM, R::ODOM_END>(t_odom, gtile, lane);
    if (skip_marked || pose_marked) load_rob_fields<RPW, R::CPOSE, R::CPOSE_END>(t_cpose, gtile, lane);
    if (pose_marked && FT::incl(GP)) load_rob_fields<RPW, R::INCL, R::INCL_END>(t_incl, gtile, lane);
    if ((F & F_TALIGN) != 0 && GP.tip_align) load_rob_fields<RPW, R::TALIGN, R::COUNT>(t_align, gtile, lane);
#pragma unroll
    for (int it = 0; it < int_iters; ++it) t_int[it] = it * 64 + lane < R::I_COUNT * RPW ? gtile_i[it * 64 + lane] : 0;
    // Leg::applyFK of the previous cycle: sin / cos of the stored joint angles
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
      double qk = ll.flat[Fields<NJ>::Q + k];
      if constexpr (HALF == ROLE_FRONT) qk = (k & 1) ? qpl[k / 2].y : qpl[k / 2].x;
      sincos_joint(th[k] + qk, &s.sn[k], &s.cs[k]);
    }
    load_leg_finish<NJ, HALF>(s, pk, ll);
  }
  {
    double2 *dst = reinterpret_cast<double2 *>(&C);
    const int nt = blockDim.x;
#pragma unroll
    for (int it = 0; it < citers; ++it) {
      int i = it * nt + threadIdx.x;
      if (i < n16) dst[i] = creg[it];
    }
  }
  if (any_robot && HALF != ROLE_BACK) {
    put_rob_fields<RPW, 0, R::CORE_END>(t_core, tile, lane);
    if (FT::manual(GP) && manual_live) put_rob_fields<RPW, R::MPOSE, R::MANUAL_END>(t_man, tile, lane);
    if (FT::imu(GP)) put_rob_fields<RPW, R::ABSE, R::IMU_END>(t_imu, tile, lane);
    if (FT::imu(GP) || FT::incl(GP) || FT::autop(GP)) put_rob_fields<RPW, R::IMUQ, R::IMUQ_END>(t_imuq, tile, lane);
    if (FT::incl(GP) && FT::autop(GP)) put_rob_fields<RPW, R::APREV, R::APREV_END>(t_aprev, tile, lane);
    if (FT::odom(GP)) put_rob_fields<RPW, R::ODOM, R::ODOM_END>(t_odom, tile, lane);
    if (skip_marked || pose_marked) put_rob_fields<RPW, R::CPOSE, R::CPOSE_END>(t_cpose, tile, lane);
    if (pose_marked && FT::incl(GP)) put_rob_fields<RPW, R::INCL, R::INCL_END>(t_incl, tile, lane);
    if ((F & F_TALIGN) != 0 && GP.tip_align) put_rob_fields<RPW, R::TALIGN, R::COUNT>(t_align, tile, lane);
#pragma unroll
    for (int it = 0; it < int_iters; ++it)
      if (it * 64 + lane < R::I_COUNT * RPW) tile_i[it * 64 + lane] = t_int[it];
  }
  SHC_TICK(18);
  __syncthreads();
  SHC_TICK(19);
  if (robots_here == 0) return; // whole wave past the end (wave-uniform)
  const CycleParams &P = C.P;
  Group<L> g{grp * L};
  RobTile<RPW> rb{tile, tile_i, grp};
  s.tipx = V3{1, 0, 0};
  constexpr bool rot_on = rot_enabled<NJ, F>();
  if (FT::adm(P) || LegRegs<NJ>::kKeepJacobian || rot_on) {
    Chain<NJ> ch;
    chain_from_sincos<NJ>(C.leg[leg], s.sn, s.cs, ch);
    if (LegRegs<NJ>::kKeepJacobian) {
      jacobian_columns<NJ>(ch, s.lin);
      s.pe = ch.pe;
    }
    if (FT::adm(P) || rot_on) s.tipx = base_rotate(C.leg[leg], ch.xe);
  }
  LegOut out;
  SHC_TICK(1);
  unsigned dirty = 0;
  double *const ext = ((F & F_ROUGH) != 0 && (rt_flags & RT_EXTERNAL) != 0) ? st.ext : nullptr; // external targets (rough terrain mode)
  const ManualRobot *const mr = ((F & F_MLEGS) != 0 && (rt_flags & RT_MANUAL_LEGS) != 0 && any_robot) ? st.manual + (rob0 + grp) : nullptr;
  const bool marked = mr != nullptr && mr->skip_cycle != 0;              // (uniform over the lanes of a robot)
  const bool skip = (skip_marked && marked) || (pose_marked && !marked);
  const bool pose_only = pose_marked && marked;
  ResidentHeld held;
  if constexpr (RES) {
    resident_loop<L, NJ, F, BATCH>(*ra, st, s, out, C, rb, pk, g, leg, slot, lane, wave, live, tile, tile_i, dirty, manual_live, held, touchdown_detection, ext);
  } else if constexpr (HALF == ROLE_FRONT) {
    FrontToBack fb;
    fb.planes_in_sync = false;
    fb.limit_bracket = -1;
    fb.uf = load_uni_flags(C.P);
    fb.pose_only = false;
    // (with admittance the update itself - AdmittanceController::updateAdmittance: admittance state, tip-force estimate, tip axis of the last FK - is the
    //  model half's, as in the two-wavefront resident kernel; the published dynamic stiffness is this half's)
    cycle_front<L, NJ, F, (F & F_ADM) == 0>(s, out, C, rb, pk, g, leg, st.legd, st.n_slots, slot, dirty, manual_live, touchdown_detection, ext, mr,
                                            LegInPlanes<NJ>{st.legd, st.n_slots, slot}, fb);
  } else if constexpr (HALF == ROLE_BACK) {
    FrontToBack fb;
    fb.uf = load_uni_flags(C.P);
    fb.pose_only = false;
    fb.joint_moved = false;
    fb.my_leg_state = 0;
    fb.rot_def = (s.word & LW_ROTDEF) != 0;
    { // PoseController::updateStance (:122-131) of this cycle, as cycle_front evaluates it (no auto posing in these specialisations: the pose is Model::current_pose_)
      const Pose bp{V3{hpose[0], hpose[1], hpose[2]}, Quat{hpose[3], hpose[4], hpose[5], hpose[6]}};
      out.poser_tip = inverse_transform_vector(bp, V3{hand[0].x, hand[0].y, hand[1].x});
      fb.desired_dir = V3{1, 0, 0};
      if (fb.rot_def) fb.desired_dir = rotate(inverse(bp.r), V3{hand[2].y, hand[3].x, hand[3].y});
    }
    out.adm_delta = V3{0.0, 0.0, 0.0};
    if (FT::adm(P)) {
      cycle_admittance<NJ>(s, out, P, LegInPlanes<NJ>{st.legd, st.n_slots, slot});
      if (live && !skip) store_admittance_early<NJ>(s, out, st, slot);
    }
    cycle_back<L, NJ, F>(s, out, C, leg, st.legd, st.n_slots, slot, mr, LegInPlanes<NJ>{st.legd, st.n_slots, slot}, fb);
  } else if (!skip) {
    for (int c = 0; c < n_cycles; ++c)
      cycle<L, NJ, F>(s, out, C, rb, pk, g, leg, st.legd, st.n_slots, slot, dirty, manual_live, touchdown_detection, ext, mr,
                      LegInPlanes<NJ>{st.legd, st.n_slots, slot}, NoHook(), (F & F_ROUGH) != 0 ? st.span : nullptr, pose_only);
    if ((F & F_MLEGS) != 0 && pose_only && FT::autop(P) && !FT::imu(P)) { // updateStance did not run: the stored per-leg poser tip stays
      const double2 *planes = reinterpret_cast<const double2 *>(st.legd);
      const double2 a = planes[(Fields<NJ>::POSER_TIP / 2) * st.n_slots + slot], b = planes[(Fields<NJ>::POSER_TIP / 2 + 1) * st.n_slots + slot];
      out.poser_tip = V3{a.x, a.y, b.x};
    }
  }
  { // OR over the wave (mirror lanes replay a live lane, so their bits are redundant, never wrong)
    unsigned d = 0;
#pragma unroll
    for (unsigned b = 1; b <= DIRTY_LAST; b <<= 1)
      if (__any((dirty & b) != 0)) d |= b;
    dirty = d;
  }
  if (live && !skip) store_leg<NJ, F, HALF, HALF != ROLE_ALL>(s, out, pk, st, P, slot, dirty);
  if constexpr (HALF == ROLE_BACK) {
    if (live) st.legi[slot] = s.word; // the walker half's word of this cycle | LW_IKFAIL
    return;                           // (the robot tile is the walker half's)
  }
  SHC_TICK(13);
  __builtin_amdgcn_wave_barrier(); // LDS ops of one wave complete in order: the tile now holds the leaders' updates
  // state planes back to this wave's HBM tile (the inputs VIN / WIN / GYRO / IMUQ are not written back)
  store_rob_fields<RPW, 0, R::PLANE>(tile, gtile, lane);
  if (dirty & DIRTY_WALK_PLANE) store_rob_fields<RPW, R::PLANE, R::VIN>(tile, gtile, lane); // walk plane + origin walk-plane pose
  if (FT::manual(P) && manual_live && (dirty & DIRTY_MANUAL)) store_rob_fields<RPW, R::MPOSE, R::MANUAL_END>(tile, gtile, lane);
  if (FT::imu(P)) store_rob_fields<RPW, R::ABSE, R::GYRO>(tile, gtile, lane);
  if (FT::incl(P) && FT::autop(P)) store_rob_fields<RPW, R::APREV, R::APREV_END>(tile, gtile, lane);
  if ((F & F_MLEGS) != 0 && FT::incl(P) && pose_marked) store_rob_fields<RPW, R::INCL, R::INCL_END>(tile, gtile, lane); // for poseForLegManipulation
  store_rob_fields<RPW, R::CPOSE, R::CPOSE_END>(tile, gtile, lane); // (walk_plane_pose_ is recomputed every cycle: LDS only)
  if (FT::odom(P)) store_rob_fields<RPW, R::ODOM, R::ODOM_END>(tile, gtile, lane);
  if ((F & F_TALIGN) != 0 && P.tip_align) store_rob_fields<RPW, R::TALIGN, R::COUNT>(tile, gtile, lane);
  static_assert((R::I_POSE_PHASE + 1) * RPW <= 64, "the written-back int fields (word, poser latches, pose phase) fit one wave-wide store");
  if (lane < (R::I_POSE_PHASE + 1) * RPW) gtile_i[lane] = tile_i[lane];
  if constexpr (RES) resident_epilogue<L, NJ, F>(*ra, st, slot, lane, live, tile, tile_i, gtile, gtile_i, held);
  SHC_TICK(14);
}

template <int L, int NJ, unsigned F>
__global__ void __launch_bounds__(256, (F & F_ROT) ? SHC_ROT_WAVES_PER_SIMD : ((F & F_DYN) && (F & F_TERRAIN)) ? SHC_TERRAIN_WAVES_PER_SIMD : SHC_WAVES_PER_SIMD) shc_cycle_kernel(DevState st, const SharedConsts<L, NJ> *gc, int n_cycles,
                                                                                             unsigned rt_flags, int64_t wave0) {
  cycle_wave<L, NJ, F, false>(st, gc, n_cycles, rt_flags, wave0 + ((int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6), nullptr);
}

// One half of one cycle per launch (see cycle_wave's HALF): each half fits two wavefronts per SIMD where the whole rotation-constrained
// cycle needs one (256 VGPRs + 76 - 92 AGPRs).
template <int L, int NJ, unsigned F, int HALF>
__global__ void __launch_bounds__(256, 2) shc_cycle_half_kernel(DevState st, const SharedConsts<L, NJ> *gc, unsigned rt_flags, int64_t wave0) {
  cycle_wave<L, NJ, F, false, HALF>(st, gc, 1, rt_flags, wave0 + ((int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6), nullptr);
}

// Resident launch: block 0 is the relay (host <-> device handshake), block 1 + w is worker wave w; 64 threads each, every block
// co-resident (the host checks the grid against the occupancy of this kernel before launching).
template <int L, int NJ, unsigned F>
// (rough terrain / tip rotations: one wavefront per SIMD - the resident loop around those cycles does not fit 256 registers without
//  scratch, and a batch that is resident has SIMDs to spare: up to ~990 wavefronts, 9 900 hexapods / 7 900 octopods)
__global__ void __launch_bounds__(64, (F & (F_ROT | F_ROUGH)) ? 1 : SHC_WAVES_PER_SIMD) shc_resident_kernel(DevState st, const SharedConsts<L, NJ> *gc, ResidentArgs ra, unsigned rt_flags) {
  if (blockIdx.x == 0) {
    resident_relay<PART_BOTH>(ra);
    return;
  }
  cycle_wave<L, NJ, F, true>(st, gc, 0, rt_flags, int64_t(blockIdx.x) - 1, &ra);
}

// The batch form of the loop (shc_engine_step_k: K cycles per launch, each with its own inputs, batches of any size): every wavefront a worker,
// no relay, no doorbell; workgroups and wavefronts per SIMD like the cycle kernels.
template <int L, int NJ, unsigned F>
__global__ void __launch_bounds__(256, (F & F_ROT) ? SHC_ROT_WAVES_PER_SIMD : SHC_WAVES_PER_SIMD) shc_batch_kernel(DevState st, const SharedConsts<L, NJ> *gc, ResidentArgs ra, unsigned rt_flags) {
  cycle_wave<L, NJ, F, true, ROLE_ALL, true>(st, gc, 0, rt_flags, ra.batch_wave0 + ((int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6), &ra);
}

// ================================================================================ resident mode, two wavefronts per robot group
// At batch sizes where resident mode applies most SIMDs of the chip have nothing to run, and one wavefront's cycle is one long
// dependent instruction stream (about 2 400 instructions at one 4-clock issue slot each).  This kernel gives every robot group
// TWO wavefronts on two SIMDs of one compute unit and pipelines the cycle across them:
//   walker wave  : cycle_front of cycle c + 1 - pose, velocity limiting, walk state machine, Bezier stepper, updateStance
//   model wave   : cycle_back of cycle c      - admittance, desired tip, DLS IK step, joint integration, FK, tip force, q / qd out
// The model half needs nothing but the poser tip of its leg (24 bytes per lane, through a double-buffered LDS mailbox) and the
// walker half needs nothing of the model half (true for every specialisation without rough terrain / tip rotations: the walker
// integrates its own tip; Leg::applyIK feeds back into the walker only through rough-terrain touchdown and tip-align poses).
// Same arithmetic, same order per leg: results are bit-identical to one wave running whole cycles.  256-thread workgroups = two
// robot groups x (walker, model): the four waves of a workgroup always get the four SIMDs of their CU (scripts/ubench/
// wave_placement.hip), and with at most one such workgroup per CU no SIMD is shared.
// Control is workgroup-uniform: the leader (wave 0) watches the gate and announces, one iteration ahead, what the next iteration
// is (REAL: run a cycle, BUBBLE: nothing released yet, EXIT) together with the header of that cycle; one barrier per iteration.
enum : int { IT_REAL = 1, IT_BUBBLE = 2, IT_EXIT = 3 };
// The walker's wait for Model::current_pose_ / walk_plane_pose_ of its cycle (the model wavefront leaves them in the robot tile and raises
// `flag` to cycle + 1).  Bounded like every other device-side wait: 1 s, then the loop reports a fault.
template <int RPW>
struct PoseWait {
  unsigned *flag_at;
  unsigned want;
  u64 ticks_per_ms;
  bool *fault;
  __device__ __forceinline__ bool give_up(unsigned &spins, u64 &t0) const {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 4095u) == 0) {
      const u64 now = wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > 1000 * ticks_per_ms) {
        *fault = true;
        return true;
      }
    }
    return false;
  }
  __device__ __forceinline__ void operator()() const {
    volatile unsigned *flag = flag_at;
    unsigned spins = 0;
    u64 t0 = 0;
    while (*flag != want)
      if (give_up(spins, t0)) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
};
template <int L, int NJ>
struct Resident2Lds { // dynamic LDS of one workgroup, after the two walker waves' tiles
  double mailbox[2][2][13][64]; // [pair][cycle parity][field][lane]: PoseController::updateStance -> Leg::setDesiredTipPose (xyz); 3..6 the desired body
                                // velocity + whether the walker reached its odometry update (where the model wavefront keeps the odometry); 7..12: the steppers'
                                // walk plane / normal for the NEXT cycle's updateWalkPlanePose (pose on the model wave)
  double pose_c[2][2][64];      // walker -> model: the control input of the next cycle's walk-plane pose (walk_plane_control_input of the leg words updateWalk left)
  unsigned pose_done[2];        // model -> walker: poses completed (current_pose_ / walk-plane pose of that cycle are in the tile)
  unsigned model_dirty[2], model_seen[2], model_fault[2]; // model -> walker at exit: tile groups its pose dirtied, input groups it received, fault
  double stiff[2][64];         // walker -> model at exit (published virtual stiffness shares a plane with the admittance delta)
  int ikfail[2][64];           // model -> walker at exit (IK-deviation flag lives in the leg word)
  unsigned long long ctrl[4][4]; // [iteration & 3]: kind, h0, h1, -
};

template <int L, int NJ, unsigned F>
__global__ void __launch_bounds__(256, 1) shc_resident2_kernel(DevState st, const SharedConsts<L, NJ> *gc, ResidentArgs A, unsigned rt_flags) {
  using R = RobotFields;
  using FD = Fields<NJ>;
  using FT = Feat<F>;
  constexpr int RPW = 64 / L;
  if (blockIdx.x == 0) { // the relay workgroup: one wave per direction
    if (threadIdx.x < 64) resident_relay<PART_GATE>(A);
    else if (threadIdx.x < 128) resident_relay<PART_DONE>(A);
    return;
  }
  __shared__ SharedConsts<L, NJ> C;
  extern __shared__ double wave_lds[];
  constexpr int kWaveDoubles = R::COUNT * RPW + PK_COUNT * 64 + (R::I_COUNT * RPW + 1) / 2;
  Resident2Lds<L, NJ> &X = *reinterpret_cast<Resident2Lds<L, NJ> *>(wave_lds + 2 * kWaveDoubles);
  const bool manual_live = (rt_flags & RT_MANUAL_LIVE) != 0;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wib & 1;
  const bool walker = wib < 2, leader = wib == 2; // the model wavefront of pair 0 decides what the next iteration is: it has the time to spare
  const int64_t wave = (int64_t(blockIdx.x) - 1) * 2 + pair;
  const bool active = wave < A.n_waves;
  const int64_t rob0 = wave * RPW;
  const int64_t left = st.n_robots - rob0;
  const int robots_here = !active ? 0 : (left < RPW ? (left < 0 ? 0 : int(left)) : RPW);
  int grp = lane / L;
  const int leg = lane - grp * L;
  const bool live = grp < robots_here;
  if (grp >= robots_here) grp = robots_here > 0 ? robots_here - 1 : 0;
  const uint32_t slot = uint32_t((active ? wave : 0) * 64 + grp * L + leg);
  double *const my_lds = wave_lds + pair * kWaveDoubles;
  Park pk{my_lds + R::COUNT * RPW, lane};
  double *tile = my_lds;
  int32_t *tile_i = reinterpret_cast<int32_t *>(my_lds + R::COUNT * RPW + PK_COUNT * 64);
  double *gtile = st.robd + (active ? wave : 0) * (R::COUNT * RPW);
  int32_t *gtile_i = st.robi + (active ? wave : 0) * (R::I_COUNT * RPW);
  const CycleParams &GP = gc->P;
  LegRegs<NJ> s;
  // ---- prologue: tables (all four waves), robot tile (walker), the half of the leg state this wave owns
  {
    using SC = SharedConsts<L, NJ>;
    constexpr int n16_all = sizeof(SC) / 16;
    constexpr int n16_core = (offsetof(SC, P) + offsetof(CycleParams, ap_start)) / 16;
    const int n16 = FT::autop(GP) ? n16_all : n16_core;
    const double2 *src = reinterpret_cast<const double2 *>(gc);
    double2 *dst = reinterpret_cast<double2 *>(&C);
    for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
  }
  LegLoad<NJ> ll;
  if (active) {
    if (walker) {
      load_leg_issue<NJ, F, ROLE_FRONT>(ll, st, GP, slot);
      for (int i = lane; i < R::COUNT * RPW; i += 64) tile[i] = gtile[i];
      for (int i = lane; i < R::I_COUNT * RPW; i += 64) tile_i[i] = gtile_i[i];
      load_leg_finish<NJ, ROLE_FRONT>(s, pk, ll);
    } else {
      load_leg_issue<NJ, F, ROLE_BACK>(ll, st, GP, slot);
      load_leg_finish<NJ, ROLE_BACK>(s, pk, ll);
    }
  }
  if (leader && lane == 0) { // what iteration 0 is: nothing has been released yet unless the host was quick
    X.ctrl[0][0] = IT_BUBBLE;
    X.ctrl[0][1] = X.ctrl[0][2] = 0;
  }
  // PoseController::updateCurrentPose runs on the MODEL wavefront where its result does not feed back into updateWalk (no auto posing): it
  // reads what the walker left one iteration earlier (leg words, the steppers' walk plane: mailbox) and the pose inputs, and the walker
  // picks Model::current_pose_ up from the tile when it reaches updateStance - the walker's critical path loses the pose, the model
  // wavefront's idle half fills up.
  constexpr bool POSE_SPLIT = (F & (F_DYN | F_AUTO)) == 0;
  if (POSE_SPLIT && walker && lane == 0) X.pose_done[pair] = 0;
  __syncthreads();
  const CycleParams &P = C.P;
  Group<L> g{grp * L};
  RobTile<RPW> rb{tile, tile_i, grp};
  LegOut out;
  out.poser_tip = out.model_tip = out.adm_delta = V3{0, 0, 0};
  s.tipx = V3{1, 0, 0};
  if (active && !walker) { // Leg::applyFK of the previous cycle, as in cycle_wave
#pragma unroll
    for (int k = 0; k < NJ; ++k) sincos_joint(C.leg[leg].link_th[k] + s.q[k], &s.sn[k], &s.cs[k]);
    if (FT::adm(P) || LegRegs<NJ>::kKeepJacobian) {
      Chain<NJ> ch;
      chain_from_sincos<NJ>(C.leg[leg], s.sn, s.cs, ch);
      if (LegRegs<NJ>::kKeepJacobian) {
        jacobian_columns<NJ>(ch, s.lin);
        s.pe = ch.pe;
      }
      if (FT::adm(P)) s.tipx = base_rotate(C.leg[leg], ch.xe);
    }
    s.word = 0;
  }
  unsigned dirty = 0;
  ResidentHeld held;
  // what PoseController::updateCurrentPose of `cycle` reads of the walker's state: the leg words, and the steppers' walk-plane copy - which
  // changes on rare events only (a default tip moved), so both parities of its mailbox slots are rewritten then and left alone otherwise
  const int swing_c_count_u = uni(P.swing_c_count);
  const auto publish_for_pose = [&](unsigned cycle, bool planes) {
    if (planes) {
      const V3 pp = rb.get3(R::PLANE_PREV), pn = rb.get3(R::PNORM_PREV);
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        double *mb = &X.mailbox[pair][par][0][lane];
        mb[7 * 64] = pp.x, mb[8 * 64] = pp.y, mb[9 * 64] = pp.z, mb[10 * 64] = pn.x, mb[11 * 64] = pn.y, mb[12 * 64] = pn.z;
      }
    }
    X.pose_c[pair][cycle & 1][lane] = walk_plane_control_candidate<L, NJ>(s.word, C, P, swing_c_count_u); // (the model wavefront picks the group's leg)
  };
  if (POSE_SPLIT && walker && active) publish_for_pose(0, true); // (iteration 0 is a bubble: its closing barrier comes before any pose)
  const int64_t ns = st.n_slots;
  const unsigned out_slot_bytes = unsigned(NJ * ns * 16);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, int(unsigned(A.depth) * out_slot_bytes), 0x00020000);
  const u64 emergency_ticks = 4 * A.idle_ticks + 2000 * A.ticks_per_ms;
  unsigned k = 0, c_front = 0, c_back = 0, oslot = 0; // iteration; cycles whose walker half has run; cycles completed; output ring position
  bool prev_real = false;
  u64 prev_h0 = 0, prev_h1 = 0;
  u64 gate_pref = leader ? ld_agent(&A.ctl->gate) : 0;
  u64 bubble_since = 0;
  FrontToBack fb{V3{1, 0, 0}, false, LS_WALKING};
  fb.uf = load_uni_flags(C.P); // (once: the parameter block does not change while the loop runs)
  OdomCache odom_cache;        // walker wavefront: (sin, cos) of the half yaw step while the desired angular velocities stay as they are
  Pose owpp_cache = pose_identity();
  if (POSE_SPLIT && active && !walker) owpp_cache = rb.getpose(R::OWPP); // ... and the origin walk-plane pose (the walker wavefront filled the tile before the barrier)
  // ... and WalkController::odometry_ideal_ itself: nothing inside the loop reads it, so the wavefront that advances it (ODOM_WALKER below) accumulates it in registers and
  // puts it back into the tile when the loop ends (4 LDS reads + 4 writes per cycle less on the longer of the two wavefronts).  Short chains
  // only: the 4- and 5-joint model wavefronts already keep part of their state in AGPRs, four more loop-carried doubles cost them more
  // register moves than the LDS traffic they save (4 000 8 x 5 octopods: 3.47 -> 3.55 us per cycle with it)
  constexpr bool ODOM_REGS = NJ <= 3;
  // ... on whichever wavefront is the shorter one (busy clocks per iteration, profiles/r05_probe_resident_balance.txt): the model wavefront for 3-joint legs
  // without admittance / IMU posing (its IK step, FK and tip force are short: 5 120 against the walker's 5 330 clocks without the odometry), the walker
  // wavefront for everything else (admittance, IMU posing, 4- and 5-joint chains make the model wavefront the longer one by 400 - 700 clocks)
  constexpr bool ODOM_WALKER = NJ > 3 || (F & (F_ADM | F_IMU | F_INCL | F_AUTO | F_DYN)) != 0;
  double odom[4] = {0.0, 0.0, 1.0, 0.0};
  if (ODOM_REGS && FT::odom(P) && active && walker == ODOM_WALKER) {
#pragma unroll
    for (int i = 0; i < 4; ++i) odom[i] = rb.get(R::ODOM + i);
  }
#ifdef SHC_RES2_TIMING
  long long tm_busy = 0, tm_total0 = __builtin_readcyclecounter(), tm_real = 0;
#ifndef SHC_RES2_BUSY_ONLY
  if (threadIdx.x < 64) shc_acc_lds[threadIdx.x] = 0;
#endif
#endif
  // The two roles run SEPARATE loops - one barrier per iteration each, and the same number of iterations: what an iteration is comes from the
  // workgroup's control words.  (One loop with the role branch inside kept every loop-carried value of BOTH roles live across the loop header:
  // phi copies in every iteration and ~100 scalar registers spilled into VGPR lanes, on the walker's and the model's critical path alike.)
#ifdef SHC_RES2_TIMING
#define SHC_R2_ITER_BEGIN() const long long tm0 = __builtin_readcyclecounter(); SHC_TICK(19)
#ifndef SHC_RES2_BUSY_ONLY
#define SHC_R2_PHASES(THREAD, ...)                                                                                                  \
  if (kind == IT_REAL && prev_real && blockIdx.x == 1 && threadIdx.x == (THREAD)) { /* phase clocks of this steady iteration */      \
    const int base = ((THREAD) >> 7) * 32;                                                                                         \
    const int order[] = {__VA_ARGS__};                                                                                             \
    for (int i = 1; i < int(sizeof(order) / sizeof(int)); ++i) shc_acc_lds[base + order[i]] += shc_ticks_lds[base + order[i]] - shc_ticks_lds[base + order[i - 1]]; \
  }
#else
#define SHC_R2_PHASES(THREAD, ...)
#endif
#define SHC_R2_ITER_END(THREAD, ...)                                                         \
  if (kind == IT_REAL && prev_real) tm_busy += __builtin_readcyclecounter() - tm0, ++tm_real; \
  SHC_R2_PHASES(THREAD, __VA_ARGS__)
#else
#define SHC_R2_ITER_BEGIN() do {} while (0)
#define SHC_R2_ITER_END(THREAD, ...) do {} while (0)
#endif
  LegInRing<NJ> in_at; // model wavefront: where this lane's per-leg inputs in force lie (recomputed when a post moves them)
  int in_src_force = -1000, in_src_effort = -1000;
  if (walker) {
    // ---------------------------------------------------------------- walker wavefront: cycle_front of cycle c_front
    for (;;) {
      SHC_R2_ITER_BEGIN();
      const int kind = __builtin_amdgcn_readfirstlane(int(X.ctrl[k & 3][0]));
      const u64 h0 = uni64(X.ctrl[k & 3][1]), h1 = uni64(X.ctrl[k & 3][2]);
      SHC_TICK(20);
      if (kind == IT_REAL && active) {
        resident_take_inputs<RPW, POSE_SPLIT ? ROBOT_VEL : ROBOT_ALL, false>(A, c_front, h0, h1, wave, lane, tile, tile_i, dirty, held, st.n_robots);
        SHC_TICK(21);
        const PoseWait<RPW> pose_wait{&X.pose_done[pair], c_front + 1, A.ticks_per_ms, &held.fault};
        cycle_front<L, NJ, F, false, LegInRing<NJ>, false, !POSE_SPLIT>(s, out, C, rb, pk, g, leg, st.legd, ns, slot, dirty, manual_live, false, nullptr, nullptr,
                                                                        LegInRing<NJ>{}, fb, nullptr, pose_wait);
#ifdef SHC_ABLATE
        if (!(P.debug_skip & 65536))
#endif
        if (POSE_SPLIT) publish_for_pose(c_front + 1, fb.plane_prev_changed);
        SHC_TICK(29);
        double *mb = &X.mailbox[pair][c_front & 1][0][lane];
        mb[0] = out.poser_tip.x, mb[64] = out.poser_tip.y, mb[128] = out.poser_tip.z;
        if constexpr (ODOM_WALKER) {
          if (FT::odom(P) && __any(fb.odom_run)) { // odometry_ideal_ (walk_controller.cpp:643; robots whose updateWalk returned early keep theirs)
            if (fb.odom_run) {
              if constexpr (ODOM_REGS) odometry_advance(odom[0], odom[1], odom[2], odom[3], P, fb.odom_vel.x, fb.odom_vel.y, fb.odom_vel.z, &odom_cache);
              else odometry_step(rb, P, fb.odom_vel.x, fb.odom_vel.y, fb.odom_vel.z, &odom_cache);
            }
          }
        } else if (FT::odom(P)) { // the accumulator is the model wavefront's: the desired body velocity travels with the poser tip
          mb[192] = fb.odom_vel.x, mb[256] = fb.odom_vel.y, mb[320] = fb.odom_vel.z, mb[384] = fb.odom_run ? 1.0 : 0.0;
        }
        SHC_TICK(22);
      }
      SHC_TICK(23);
      if (kind == IT_EXIT) break;
      SHC_R2_ITER_END(0, 19, 20, 21, 2, 3, 4, 5, 6, 7, 16, 8, 17, 15, 29, 22, 23);
      if (kind == IT_REAL) ++c_front;
      prev_real = kind == IT_REAL;
      __syncthreads();
      ++k;
    }
  } else {
    // ---------------------------------------------------------------- model wavefront: the pose of cycle c_front, cycle_back of cycle c_back
    for (;;) {
      SHC_R2_ITER_BEGIN();
      const int kind = __builtin_amdgcn_readfirstlane(int(X.ctrl[k & 3][0]));
      const u64 h0 = uni64(X.ctrl[k & 3][1]), h1 = uni64(X.ctrl[k & 3][2]);
      SHC_TICK(29);
      int nk = IT_EXIT;
      u64 nh0v = 0, nh1v = 0;
      if (leader && kind != IT_EXIT) { // what will iteration k + 1 be?
        u64 gate;
        if (kind == IT_BUBBLE) {
          __builtin_amdgcn_s_sleep(2);
          gate = uni64(ld_agent(&A.ctl->gate)); // nothing else to do: look again
        } else {
          gate = uni64(gate_pref); // read one iteration ago: at worst the loop learns of a release one iteration late
        }
        const unsigned db = unsigned(gate), sp = unsigned(gate >> 32);
        const unsigned cn = c_front + (kind == IT_REAL ? 1u : 0u); // the cycle iteration k + 1 would start
        nk = cn >= sp ? IT_EXIT : (cn < db ? IT_REAL : IT_BUBBLE);
        if (nk == IT_BUBBLE) {
          const u64 now = wall_clock64();
          if (bubble_since == 0) bubble_since = now;
          else if (now - bubble_since > emergency_ticks) nk = IT_EXIT, held.fault = true;
        } else {
          bubble_since = 0;
        }
        if (nk == IT_REAL) {
          const u64 *hp = reinterpret_cast<const u64 *>(A.headers + (cn & (kResidentHeaders - 1)));
          nh0v = ld_agent(hp);
          nh1v = ld_agent(hp + 1);
        }
        // (out of a bubble the gate was read just now: the control words below wait for every load of this iteration, and one more trip to
        //  memory on the way out of a bubble is a microsecond on the latency of every burst)
        if (kind == IT_BUBBLE) gate_pref = gate;
        else gate_pref = ld_agent(&A.ctl->gate);
      }
      if (active) {
        SHC_TICK(24);
        if constexpr (POSE_SPLIT) if (kind == IT_REAL) { // PoseController::updateCurrentPose of the cycle the walker is starting: first thing, the walker waits for it
          resident_take_inputs<RPW, ROBOT_POSE, false>(A, c_front, h0, h1, wave, lane, tile, tile_i, dirty, held, st.n_robots);
          SHC_TICK(30);
          double pose_c = -1.0; // the candidate of the LAST leg (in id order) of this lane's robot that has one (pose_controller.cpp:1100-1108)
#pragma unroll
          for (int j = 0; j < L; ++j) {
            const double cj = X.pose_c[pair][c_front & 1][g.base + j];
            pose_c = cj >= 0.0 ? cj : pose_c;
          }
          int lw[L] = {}; // (not filled in: the walker wavefront has already reduced the leg words to the pose's control input)
          const double *mb = &X.mailbox[pair][c_front & 1][0][lane];
          const V3 plane_prev{mb[7 * 64], mb[8 * 64], mb[9 * 64]}, pnorm_prev{mb[10 * 64], mb[11 * 64], mb[12 * 64]};
          int rword_unused = 0;
          Pose ap = pose_identity(), la = pose_identity();
#ifdef SHC_ABLATE
          if (!(P.debug_skip & 2048))
#endif
          (void)cycle_pose<L, NJ, F, true>(s, C, P, C.leg[leg], rb, g, leg, lw, rword_unused, 0, dirty, manual_live, ap, la, plane_prev, pnorm_prev, fb.uf.swing_c_count, 0,
                                           &owpp_cache, &pose_c);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) *const_cast<volatile unsigned *>(&X.pose_done[pair]) = c_front + 1;
        }
        SHC_TICK(25);
        if (prev_real) { // the model half of the cycle whose walker half ran one iteration ago
          resident_take_inputs<RPW, ROBOT_NONE, true>(A, c_back, prev_h0, prev_h1, wave, lane, tile, tile_i, dirty, held, st.n_robots);
          if (held.src_force != in_src_force || held.src_effort != in_src_effort) { // (wave-uniform; where the per-leg inputs lie changes with a post of that group only)
            in_at = leg_inputs_in_force<L, NJ>(A, st.legd, held.src_force, held.src_effort, ns, slot, wave * RPW + grp, leg);
            in_src_force = held.src_force, in_src_effort = held.src_effort;
          }
          LegInRing<NJ> in = in_at;
          if (FT::tipf(P)) in.prefetch_effort();
          SHC_TICK(31);
          const double *mb = &X.mailbox[pair][c_back & 1][0][lane];
          out.poser_tip = V3{mb[0], mb[64], mb[128]};
          if constexpr (!ODOM_WALKER) if (FT::odom(P)) {
            const V3 ov{mb[192], mb[256], mb[320]};
            if (__any(mb[384] != 0.0)) { // (robots whose updateWalk returned early keep their odometry)
              if (mb[384] != 0.0) {
                if constexpr (ODOM_REGS) odometry_advance(odom[0], odom[1], odom[2], odom[3], P, ov.x, ov.y, ov.z, &odom_cache);
                else odometry_step(rb, P, ov.x, ov.y, ov.z, &odom_cache);
              }
            }
          }
          out.adm_delta = V3{0, 0, 0};
          if (FT::adm(P)) cycle_admittance<NJ>(s, out, P, in);
          s.word = 0;
          SHC_TICK(26);
          cycle_back<L, NJ, F>(s, out, C, leg, st.legd, ns, slot, nullptr, in, fb);
          SHC_TICK(27);
          // the output stores of cycle c_back - 1 were issued one iteration ago: they have drained - announce them, then issue this cycle's
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) st_agent(A.progress + wave, u64(c_back));
          SHC_TICK(13);
          {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            double flat[2 * NJ];
#pragma unroll
            for (int i = 0; i < NJ; ++i) flat[FD::Q + i] = s.q[i], flat[FD::QD + i] = s.qd[i];
            const unsigned soff = oslot * out_slot_bytes;
#ifdef SHC_ABLATE
            if (!(P.debug_skip & 4096))
#endif
            if (live) {
#pragma unroll
              for (int p = 0; p < NJ; ++p) {
                const u64 a = u64(__double_as_longlong(flat[2 * p])), b = u64(__double_as_longlong(flat[2 * p + 1]));
                const v4u w = {unsigned(a), unsigned(a >> 32), unsigned(b), unsigned(b >> 32)};
                __builtin_amdgcn_raw_buffer_store_b128(w, out_rsrc, unsigned((int64_t(p) * ns + slot) * 16), soff, 16 /* sc1 */);
              }
            }
          }
          ++c_back;
          oslot = oslot + 1 == unsigned(A.depth) ? 0 : oslot + 1;
          SHC_TICK(28);
        }
        if (kind != IT_REAL) { // the pipeline runs dry: nothing will follow for a while (or ever) - announce what is done now
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) st_agent(A.progress + wave, u64(c_back));
        }
      }
      if (leader && kind != IT_EXIT && lane == 0) { // (the header loads issued at the top of the iteration have long arrived)
        X.ctrl[(k + 1) & 3][0] = u64(nk);
        X.ctrl[(k + 1) & 3][1] = nh0v;
        X.ctrl[(k + 1) & 3][2] = nh1v;
      }
      SHC_TICK(14);
      if (kind == IT_EXIT) break;
      SHC_R2_ITER_END(128, 19, 29, 24, 30, 25, 31, 26, 9, 10, 11, 12, 27, 13, 28, 14);
      if (kind == IT_REAL) ++c_front;
      prev_real = kind == IT_REAL;
      prev_h0 = h0, prev_h1 = h1;
      __syncthreads();
      ++k;
    }
  }
#undef SHC_R2_ITER_BEGIN
#undef SHC_R2_ITER_END
#undef SHC_R2_PHASES
#ifdef SHC_RES2_TIMING
  if (blockIdx.x == 1 && lane == 0 && pair == 0) { // development: clocks from iteration start to the barrier, REAL iterations in steady state
    unsigned long long *dbg = reinterpret_cast<unsigned long long *>(A.ctl) + 8 + (walker ? 0 : 4);
    dbg[0] = tm_busy, dbg[1] = tm_real, dbg[2] = __builtin_readcyclecounter() - tm_total0, dbg[3] = k;
#ifndef SHC_RES2_BUSY_ONLY
    if (walker) {
      for (int i = 0; i < 32; ++i) dbg[8 + i] = (unsigned long long)shc_acc_lds[i];
    } else { // (dbg points 4 words into ResidentCtl::dbg for the model wavefront: its phase clocks go to dbg[40 .. 72) of the record)
      for (int i = 0; i < 32; ++i) dbg[36 + i] = (unsigned long long)shc_acc_lds[32 + i];
    }
#endif
  }
#endif
  // ---- epilogue: the halves exchange what the other one stores, then each writes its half of the state back
  if (active && ODOM_REGS && FT::odom(P) && walker == ODOM_WALKER) { // the odometry accumulated in registers back into the tile the walker wavefront stores
#pragma unroll
    for (int i = 0; i < 4; ++i) rb.put(R::ODOM + i, odom[i]);
  }
  if (active) {
    if (walker) {
      X.stiff[pair][lane] = s.stiff;
    } else {
      X.ikfail[pair][lane] = s.word & LW_IKFAIL;
      if (POSE_SPLIT) { // what the pose on this wavefront dirtied / received is written back by the walker, which owns the tile stores
        unsigned d = 0;
#pragma unroll
        for (unsigned b = 1; b <= DIRTY_LAST; b <<= 1)
          if (__any((dirty & b) != 0)) d |= b;
        if (lane == 0) X.model_dirty[pair] = d, X.model_seen[pair] = held.seen;
      }
      if (lane == 0) X.model_fault[pair] = held.fault ? 1u : 0u; // (the leader's emergency bound)
    }
  }
  __syncthreads();
  if (!active) return;
  if (walker) {
    if (POSE_SPLIT) dirty |= X.model_dirty[pair], held.seen |= X.model_seen[pair];
    if (X.model_fault[pair]) held.fault = true;
    if (c_front > 0) s.word = (s.word & ~LW_IKFAIL) | X.ikfail[pair][lane];
    {
      unsigned d = 0;
#pragma unroll
      for (unsigned b = 1; b <= DIRTY_LAST; b <<= 1)
        if (__any((dirty & b) != 0)) d |= b;
      dirty = d;
    }
    if (live) store_leg<NJ, F, ROLE_FRONT>(s, out, pk, st, P, slot, dirty);
    __builtin_amdgcn_wave_barrier();
    store_rob_fields<RPW, 0, R::PLANE>(tile, gtile, lane);
    if (dirty & DIRTY_WALK_PLANE) store_rob_fields<RPW, R::PLANE, R::VIN>(tile, gtile, lane);
    if (FT::manual(P) && manual_live && (dirty & DIRTY_MANUAL)) store_rob_fields<RPW, R::MPOSE, R::MANUAL_END>(tile, gtile, lane);
    if (FT::imu(P)) store_rob_fields<RPW, R::ABSE, R::GYRO>(tile, gtile, lane);
    if (FT::incl(P) && FT::autop(P)) store_rob_fields<RPW, R::APREV, R::APREV_END>(tile, gtile, lane);
    store_rob_fields<RPW, R::CPOSE, R::CPOSE_END>(tile, gtile, lane);
    if (FT::odom(P)) store_rob_fields<RPW, R::ODOM, R::ODOM_END>(tile, gtile, lane);
    if (lane < (R::I_POSE_PHASE + 1) * RPW) gtile_i[lane] = tile_i[lane];
    // the inputs the run received become the engine's held inputs
    store_rob_fields<RPW, R::VIN, R::CORE_END>(tile, gtile, lane);
    if (held.seen & (1u << RG_IMU)) {
      store_rob_fields<RPW, R::GYRO, R::IMU_END>(tile, gtile, lane);
      store_rob_fields<RPW, R::IMUQ, R::IMUQ_END>(tile, gtile, lane);
    }
    if ((held.seen & (1u << RG_RESET)) && lane < RPW) gtile_i[R::I_RESET_MODE * RPW + lane] = tile_i[R::I_RESET_MODE * RPW + lane];
    if (lane == 0) {
      if (held.fault) __hip_atomic_fetch_or(&A.ctl->fault, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&A.ctl->exited, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    s.stiff = X.stiff[pair][lane];
    if (live) {
      store_leg<NJ, F, ROLE_BACK>(s, out, pk, st, P, slot, 0);
      carry_leg_inputs<L, NJ>(A, st, held, ns, slot, leg);
    }
  }
}

} // namespace shc

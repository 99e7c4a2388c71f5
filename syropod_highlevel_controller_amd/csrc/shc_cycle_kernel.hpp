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
  double2 adm, tf0, tf1, stiff, rot0, rot1, rot2;
  int word;
};
template <int NJ, unsigned F>
__device__ __forceinline__ void load_leg_issue(LegLoad<NJ> &ll, const DevState &st, const CycleParams &P, uint32_t slot) {
  using FD = Fields<NJ>;
  using FT = Feat<F>;
  const LegPlanes ld{reinterpret_cast<double2 *>(st.legd), st.n_slots, slot};
#pragma unroll
  for (int p = 0; p < FD::CORE_END / 2; ++p) {
    double2 v = ld.load(p);
    ll.flat[2 * p] = v.x;
    ll.flat[2 * p + 1] = v.y;
  }
  ll.word = st.legi[slot];
  ll.adm = ll.tf0 = ll.tf1 = ll.stiff = ll.rot0 = ll.rot1 = ll.rot2 = double2{0.0, 0.0};
  if (NJ > 3 && (F & F_ROT)) { // tip directions of the stepper's origin / current tip rotations
    ll.rot0 = ld.load(FD::ORG_DIR / 2);
    ll.rot1 = ld.load(FD::ORG_DIR / 2 + 1);
    ll.rot2 = ld.load(FD::ORG_DIR / 2 + 2);
  }
  if (FT::adm(P)) {
    ll.adm = ld.load(FD::ADM / 2);
    if (P.dynamic_stiffness) ll.stiff = ld.load(FD::ADM_DELTA / 2 + 1); // virtual_stiffness_ persists while STOPPED
  }
  if (FT::tipf(P)) {
    ll.tf0 = ld.load(FD::TF / 2);
    ll.tf1 = ld.load(FD::TF / 2 + 1);
  }
}
template <int NJ>
__device__ __forceinline__ void load_leg_finish(LegRegs<NJ> &s, const Park &pk, const LegLoad<NJ> &ll) {
  using FD = Fields<NJ>;
  const double(&flat)[FD::CORE_END] = ll.flat;
  s.word = ll.word;
  // swing origin / velocity, stance origin and default tip go straight to the per-lane LDS strip
  static_assert(FD::SVEL == FD::SORG + 3 && FD::TORG == FD::SORG + 6 && FD::DFLT == FD::SORG + 9, "park layout");
#pragma unroll
  for (int k = 0; k < PK_COUNT; ++k) pk.d[k * 64 + pk.lane] = flat[FD::SORG + k];
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
}

template <int NJ, unsigned F>
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
  for (int k = 0; k < PK_COUNT; ++k) flat[FD::SORG + k] = pk.d[k * 64 + pk.lane];
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
    ld.store(p, double2{flat[2 * p], flat[2 * p + 1]});
  }
  if (FT::adm(P)) {
    ld.store(FD::ADM / 2, double2{s.adm0, s.adm1});
    ld.store(FD::ADM_DELTA / 2, double2{out.adm_delta.x, out.adm_delta.y});
    ld.store(FD::ADM_DELTA / 2 + 1, double2{out.adm_delta.z, s.stiff});
  }
  if (FT::tipf(P)) {
    ld.store(FD::TF / 2, double2{s.tf.x, s.tf.y});
    ld.store(FD::TF / 2 + 1, double2{s.tf.z, 0.0});
  }
  // LegState outputs: the model tip is FK(q) and, without per-leg auto poses, the poser tip is the walker tip seen from
  // Model::current_pose_ - both are derived from the stored state when a getter asks (derive_tips_kernel), not written
  // every launch.  Only the auto-pose path's per-leg pose is not recoverable, so it stores its poser tip.
  if (FT::autop(P) && !FT::imu(P)) {
    ld.store(FD::POSER_TIP / 2, double2{out.poser_tip.x, out.poser_tip.y});
    ld.store(FD::POSER_TIP / 2 + 1, double2{out.poser_tip.z, 0.0});
  }
  if (NJ > 3 && (F & F_ROT)) {
    static_assert(FD::ORG_DIR % 2 == 0 && FD::CUR_DIR == FD::ORG_DIR + 3, "tip direction planes");
    ld.store(FD::ORG_DIR / 2, double2{s.org_dir.x, s.org_dir.y});
    ld.store(FD::ORG_DIR / 2 + 1, double2{s.org_dir.z, s.cur_dir.x});
    ld.store(FD::ORG_DIR / 2 + 2, double2{s.cur_dir.y, s.cur_dir.z});
  }
  st.legi[slot] = s.word;
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

// One launch = n_cycles control cycles of every robot; per-leg state stays in registers, per-robot state in LDS.
template <int L, int NJ, unsigned F>
__global__ void __launch_bounds__(256, (F & F_ROT) ? 1 : SHC_WAVES_PER_SIMD) shc_cycle_kernel(DevState st, const SharedConsts<L, NJ> *gc, int n_cycles,
                                                                                             unsigned rt_flags) {
  using R = RobotFields;
  using FT = Feat<F>;
  constexpr int RPW = 64 / L; // robots per wavefront
  __shared__ SharedConsts<L, NJ> C;
  // per-wave LDS (robot tile, its int words, park strip) is dynamic: sized for the workgroup actually launched (1 wave per
  // workgroup for small batches, 4 for large ones), cycle_lds_bytes_per_wave() each
  extern __shared__ double wave_lds[];
  constexpr int kWaveDoubles = R::COUNT * RPW + PK_COUNT * 64 + (R::I_COUNT * RPW + 1) / 2;
#ifdef SHC_TIMING
  const bool shc_tick_on = shc_tick_buf && blockIdx.x == 0 && threadIdx.x == 0;
#endif
  SHC_TICK(0);
  // Launch-uniform run-time facts the host knows (kernel argument = SGPR from the first instruction on, no load to wait for):
  // RT_MANUAL_LIVE - some pose input / reset mode / injected state has ever been given to this engine.  Until then every
  // robot's manual pose is the identity and stays it, so the manual-pose group of the robot tile is neither loaded nor
  // evaluated nor stored.
  const bool manual_live = (rt_flags & RT_MANUAL_LIVE) != 0;
  const bool touchdown_detection = (rt_flags & RT_TOUCHDOWN) != 0;
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int64_t wave = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
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
  if (any_robot) {
#pragma unroll
    for (int k = 0; k < NJ; ++k) th[k] = gc->leg[leg].link_th[k];
    load_leg_issue<NJ, F>(ll, st, GP, slot);
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
      t_cpose[((R::CPOSE_END - R::CPOSE) * RPW + 63) / 64];
  // RT_SKIP_MARKED - this launch follows a loop-level kernel (leg toggle, plan execution) that has already run the loop of the robots it
  // marked (ManualRobot::skip_cycle): they are left exactly as they are.  Model::current_pose_ is otherwise output only; here it
  // is loaded too so that the tile write-back is the identity for a skipped robot.
  const bool skip_marked = (F & F_TERRAIN) != 0 && (rt_flags & RT_SKIP_MARKED) != 0 && (rt_flags & RT_MANUAL_LEGS) != 0;
  constexpr int int_iters = (R::I_COUNT * RPW + 63) / 64; // 3-legged robots: 21 per wave x 4 ints = 84 entries > one wave's width
  int32_t t_int[int_iters];
  if (any_robot) {
    load_rob_fields<RPW, 0, R::CORE_END>(t_core, gtile, lane);
    if (FT::manual(GP) && manual_live) load_rob_fields<RPW, R::MPOSE, R::MANUAL_END>(t_man, gtile, lane);
    if (FT::imu(GP)) load_rob_fields<RPW, R::ABSE, R::IMU_END>(t_imu, gtile, lane);
    if (FT::imu(GP) || FT::incl(GP) || FT::autop(GP)) load_rob_fields<RPW, R::IMUQ, R::IMUQ_END>(t_imuq, gtile, lane);
    if (FT::incl(GP) && FT::autop(GP)) load_rob_fields<RPW, R::APREV, R::APREV_END>(t_aprev, gtile, lane);
    if (FT::odom(GP)) load_rob_fields<RPW, R::ODOM, R::ODOM_END>(t_odom, gtile, lane);
    if (skip_marked) load_rob_fields<RPW, R::CPOSE, R::CPOSE_END>(t_cpose, gtile, lane);
    if ((F & F_TERRAIN) != 0 && NJ <= 3 && GP.tip_align) load_rob_fields<RPW, R::TALIGN, R::COUNT>(t_align, gtile, lane);
#pragma unroll
    for (int it = 0; it < int_iters; ++it) t_int[it] = it * 64 + lane < R::I_COUNT * RPW ? gtile_i[it * 64 + lane] : 0;
    // Leg::applyFK of the previous cycle: sin / cos of the stored joint angles
#pragma unroll
    for (int k = 0; k < NJ; ++k) sincos_joint(th[k] + ll.flat[Fields<NJ>::Q + k], &s.sn[k], &s.cs[k]);
    load_leg_finish<NJ>(s, pk, ll);
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
  if (any_robot) {
    put_rob_fields<RPW, 0, R::CORE_END>(t_core, tile, lane);
    if (FT::manual(GP) && manual_live) put_rob_fields<RPW, R::MPOSE, R::MANUAL_END>(t_man, tile, lane);
    if (FT::imu(GP)) put_rob_fields<RPW, R::ABSE, R::IMU_END>(t_imu, tile, lane);
    if (FT::imu(GP) || FT::incl(GP) || FT::autop(GP)) put_rob_fields<RPW, R::IMUQ, R::IMUQ_END>(t_imuq, tile, lane);
    if (FT::incl(GP) && FT::autop(GP)) put_rob_fields<RPW, R::APREV, R::APREV_END>(t_aprev, tile, lane);
    if (FT::odom(GP)) put_rob_fields<RPW, R::ODOM, R::ODOM_END>(t_odom, tile, lane);
    if (skip_marked) put_rob_fields<RPW, R::CPOSE, R::CPOSE_END>(t_cpose, tile, lane);
    if ((F & F_TERRAIN) != 0 && NJ <= 3 && GP.tip_align) put_rob_fields<RPW, R::TALIGN, R::COUNT>(t_align, tile, lane);
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
  constexpr bool rot_on = NJ > 3 && (F & F_ROT) != 0;
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
  double *const ext = ((F & F_TERRAIN) != 0 && (rt_flags & RT_EXTERNAL) != 0) ? st.ext : nullptr; // external targets (rough terrain mode)
  const ManualRobot *const mr = ((F & F_TERRAIN) != 0 && (rt_flags & RT_MANUAL_LEGS) != 0 && any_robot) ? st.manual + (rob0 + grp) : nullptr;
  const bool skip = skip_marked && mr != nullptr && mr->skip_cycle != 0; // (uniform over the lanes of a robot)
  if (!skip)
    for (int c = 0; c < n_cycles; ++c)
      cycle<L, NJ, F>(s, out, C, rb, pk, g, leg, st.legd, st.n_slots, slot, dirty, manual_live, touchdown_detection, ext, mr);
  { // OR over the wave (mirror lanes replay a live lane, so their bits are redundant, never wrong)
    unsigned d = 0;
#pragma unroll
    for (unsigned b = 1; b <= DIRTY_STANCE_ORG; b <<= 1)
      if (__any((dirty & b) != 0)) d |= b;
    dirty = d;
  }
  if (live && !skip) store_leg<NJ, F>(s, out, pk, st, P, slot, dirty);
  SHC_TICK(13);
  __builtin_amdgcn_wave_barrier(); // LDS ops of one wave complete in order: the tile now holds the leaders' updates
  // state planes back to this wave's HBM tile (the inputs VIN / WIN / GYRO / IMUQ are not written back)
  store_rob_fields<RPW, 0, R::PLANE>(tile, gtile, lane);
  if (dirty & DIRTY_WALK_PLANE) store_rob_fields<RPW, R::PLANE, R::VIN>(tile, gtile, lane); // walk plane + origin walk-plane pose
  if (FT::manual(P) && manual_live && (dirty & DIRTY_MANUAL)) store_rob_fields<RPW, R::MPOSE, R::MANUAL_END>(tile, gtile, lane);
  if (FT::imu(P)) store_rob_fields<RPW, R::ABSE, R::GYRO>(tile, gtile, lane);
  if (FT::incl(P) && FT::autop(P)) store_rob_fields<RPW, R::APREV, R::APREV_END>(tile, gtile, lane);
  store_rob_fields<RPW, R::CPOSE, R::CPOSE_END>(tile, gtile, lane); // (walk_plane_pose_ is recomputed every cycle: LDS only)
  if (FT::odom(P)) store_rob_fields<RPW, R::ODOM, R::ODOM_END>(tile, gtile, lane);
  if ((F & F_TERRAIN) != 0 && NJ <= 3 && P.tip_align) store_rob_fields<RPW, R::TALIGN, R::COUNT>(tile, gtile, lane);
  static_assert((R::I_POSE_PHASE + 1) * RPW <= 64, "the written-back int fields (word, poser latches, pose phase) fit one wave-wide store");
  if (lane < (R::I_POSE_PHASE + 1) * RPW) gtile_i[lane] = tile_i[lane];
  SHC_TICK(14);
}

} // namespace shc

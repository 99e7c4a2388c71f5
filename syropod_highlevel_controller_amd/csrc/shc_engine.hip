// shc_engine.hip — libshc_batch.so: C ABI (include/shc_batch.h), host init chain and HIP kernels (gfx950).
//
// Product code.  Nothing here links, includes or calls anything under oracle/.  There is no CPU fallback for
// the cycle: without a HIP device shc_engine_create fails with SHC_ERR_NO_DEVICE.
#include "../../include/shc_batch.h"
#include "shc_cycle.hpp"
#include "shc_cycle_launch.hpp"
#include "shc_host_init.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace shc;

#ifndef SHC_WAVES_PER_SIMD
#define SHC_WAVES_PER_SIMD 2
#endif

static thread_local std::string g_last_error;
static int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}
#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) return fail(SHC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// as HIP_TRY, running `cleanup` before the early return (error paths must not leak device buffers)
#define HIP_TRY_OR(expr, cleanup)                                                                       \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) {                                                                             \
      cleanup;                                                                                          \
      return fail(SHC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                     \
    }                                                                                                   \
  } while (0)

// ================================================================================================= kernels

// (the fused cycle kernels live in shc_cycle_kernel.hpp / shc_cycle_inst.hip: one translation unit per morphology)

__global__ void shc_plane_copy_kernel(const double2 *__restrict__ src, double2 *__restrict__ dst, int64_t n_pairs) {
  int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n_pairs) dst[i] = src[i];
}

// ---- layout conversion kernels (C ABI instance-major arrays <-> SoA fields)
// element index of field f of robot r in the AoSoA robot tiles ([wave][field][robots-per-wave])
__device__ __forceinline__ int64_t rob_index(int64_t r, int f, int rpw, int nf) { return ((r / rpw) * nf + f) * rpw + (r % rpw); }
__device__ __forceinline__ int64_t slot_of(int64_t rob, int leg, int L) {
  int rpw = 64 / L;
  int64_t w = rob / rpw;
  int gi = int(rob - w * rpw);
  return w * 64 + gi * L + leg;
}

#include "shc_snapshot.hpp" // get_state / set_state kernels (use rob_index / slot_of)
#include "shc_leg_api.hpp"  // per-leg Leg methods, batched
#include "shc_sequence.hpp" // executeSequence / stepToNewStance: per-robot state machines over the per-leg primitives

// AoS [n][L][K] -> leg fields f0..f0+K-1
__global__ void scatter_leg_kernel(const double *src, double *legd, int64_t n_slots, int64_t n, int L, int K, int f0, int64_t r0 = 0) {
  int64_t t = r0 * L + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; // instances [r0, n)
  if (t >= n * L) return;
  int64_t rob = t / L;
  int leg = int(t - rob * L);
  int64_t slot = slot_of(rob, leg, L);
  for (int k = 0; k < K; ++k) legd[leg_field_index(f0 + k, slot, n_slots)] = src[t * K + k];
}
__global__ void gather_leg_kernel(double *dst, const double *legd, int64_t n_slots, int64_t n, int L, int K, int f0) {
  int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n * L) return;
  int64_t rob = t / L;
  int leg = int(t - rob * L);
  int64_t slot = slot_of(rob, leg, L);
  for (int k = 0; k < K; ++k) dst[t * K + k] = legd[leg_field_index(f0 + k, slot, n_slots)];
}
// LegState tips derived from the stored state (see store_leg): model tip = FK(q) in the robot frame (Leg::applyFK,
// model.cpp:975), poser tip = Model::current_pose_^-1 * walker tip (PoseController::updateStance, pose_controller.cpp:122-131).
template <int L, int NJ>
__global__ void derive_tips_kernel(DevState st, const SharedConsts<L, NJ> *gc, int derive_poser, int keep_marked) {
  using FD = Fields<NJ>;
  using R = RobotFields;
  int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= st.n_robots * L) return;
  int64_t rob = t / L;
  int leg = int(t - rob * L);
  int64_t slot = slot_of(rob, leg, L);
  const LegConst<NJ> &lc = gc->leg[leg];
  double q[NJ];
  for (int j = 0; j < NJ; ++j) q[j] = st.legd[leg_field_index(FD::Q + j, slot, st.n_slots)];
  Chain<NJ> ch;
  fk_chain<NJ>(lc, q, ch);
  V3 tip = tip_robot_frame(lc, ch.pe);
  st.legd[leg_field_index(FD::MODEL_TIP, slot, st.n_slots)] = tip.x;
  st.legd[leg_field_index(FD::MODEL_TIP + 1, slot, st.n_slots)] = tip.y;
  st.legd[leg_field_index(FD::MODEL_TIP + 2, slot, st.n_slots)] = tip.z;
  // keep_marked: the last loop of the marked robots was a plan call under time-dependent posing - their LegPoser tips are state (the
  // pose has moved on since the updateStance that produced them, see execute_plan_kernel LOOP_MARK)
  if (derive_poser && !(keep_marked && st.manual != nullptr && st.manual[rob].skip_cycle != 0)) {
    constexpr int rpw = 64 / L;
    double c[7];
    for (int k = 0; k < 7; ++k) c[k] = st.robd[rob_index(rob, R::CPOSE + k, rpw, R::COUNT)];
    V3 w{st.legd[leg_field_index(FD::TIP, slot, st.n_slots)], st.legd[leg_field_index(FD::TIP + 1, slot, st.n_slots)],
         st.legd[leg_field_index(FD::TIP + 2, slot, st.n_slots)]};
    V3 pt = inverse_transform_vector(Pose{V3{c[0], c[1], c[2]}, Quat{c[3], c[4], c[5], c[6]}}, w);
    st.legd[leg_field_index(FD::POSER_TIP, slot, st.n_slots)] = pt.x;
    st.legd[leg_field_index(FD::POSER_TIP + 1, slot, st.n_slots)] = pt.y;
    st.legd[leg_field_index(FD::POSER_TIP + 2, slot, st.n_slots)] = pt.z;
  }
}
// odometry_ideal_ is stored as (x, y, qw, qz): expand to the pose layout of the ABI (x, y, z, qw, qx, qy, qz)
__global__ void gather_odometry_kernel(double *dst, const double *robd, int rpw, int64_t n) {
  constexpr int nf = RobotFields::COUNT;
  int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double *o = dst + r * 7;
  o[0] = robd[rob_index(r, RobotFields::ODOM, rpw, nf)];
  o[1] = robd[rob_index(r, RobotFields::ODOM + 1, rpw, nf)];
  o[2] = 0.0;
  o[3] = robd[rob_index(r, RobotFields::ODOM + 2, rpw, nf)];
  o[4] = 0.0;
  o[5] = 0.0;
  o[6] = robd[rob_index(r, RobotFields::ODOM + 3, rpw, nf)];
}
// everything publishLegState needs of one instance, packed per leg: tip(3) targ(3) poser(3) model(3) q(NJ) qd(NJ) tf(3)
// adm(3) stiff word effort(NJ) measured q(NJ) | then vx vy w, robot word, poser latches, pose phase, IMU orientation (4)
template <int NJ>
__global__ void read_instance_kernel(double *dst, DevState st, int L, int64_t rob, int have_adm) {
  using FD = Fields<NJ>;
  using R = RobotFields;
  const int leg = threadIdx.x;
  constexpr int per_leg = 12 + 4 * NJ + 8;
  if (leg < L) {
    const int64_t slot = slot_of(rob, leg, L);
    auto f = [&](int field) { return st.legd[leg_field_index(field, slot, st.n_slots)]; };
    double *o = dst + leg * per_leg;
    for (int k = 0; k < 3; ++k) {
      o[k] = f(FD::TIP + k);
      o[3 + k] = f(FD::TARG + k);
      o[6 + k] = f(FD::POSER_TIP + k);
      o[9 + k] = f(FD::MODEL_TIP + k);
      o[12 + 2 * NJ + k] = f(FD::TF + k);
      o[15 + 2 * NJ + k] = have_adm ? f(FD::ADM_DELTA + k) : 0.0;
    }
    for (int j = 0; j < NJ; ++j) {
      o[12 + j] = f(FD::Q + j);
      o[12 + NJ + j] = f(FD::QD + j);
      o[20 + 2 * NJ + j] = f(FD::EFFORT_IN + j);
      o[20 + 3 * NJ + j] = f(FD::MEAS_Q + j);
    }
    o[18 + 2 * NJ] = have_adm ? f(FD::ADM_DELTA + 3) : 0.0;
    o[19 + 2 * NJ] = double(st.legi[slot]);
  }
  if (leg == 0) {
    const int rpw = 64 / L;
    double *o = dst + L * per_leg;
    for (int k = 0; k < 3; ++k) o[k] = st.robd[rob_index(rob, R::VLIN + k, rpw, R::COUNT)];
    o[3] = double(st.robi[rob_index(rob, R::I_WORD, rpw, R::I_COUNT)]);
    o[4] = double(st.robi[rob_index(rob, R::I_APOSER, rpw, R::I_COUNT)]);
    o[5] = double(st.robi[rob_index(rob, R::I_POSE_PHASE, rpw, R::I_COUNT)]);
    for (int k = 0; k < 4; ++k) o[6 + k] = st.robd[rob_index(rob, R::IMUQ + k, rpw, R::COUNT)];
  }
}
__global__ void gather_leg_status_kernel(int32_t *dst, const int32_t *legi, int64_t n, int L) {
  int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n * L) return;
  int64_t rob = t / L;
  int leg = int(t - rob * L);
  int w = legi[slot_of(rob, leg, L)];
  int phase = (w >> LW_PHASE_SHIFT) & LW_PHASE_MASK;
  dst[t] = (w & 3) | ((w & LW_IKFAIL) ? 4 : 0) | (phase << 8);
}
// AoS [n][K] -> robot fields

__global__ void scatter_rob_kernel(const double *src, double *robd, int rpw, int64_t n, int K, int f0, int normalize_quat, int64_t r0 = 0) {
  constexpr int nf = RobotFields::COUNT;
  int64_t r = r0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; // instances [r0, n)
  if (r >= n) return;
  if (normalize_quat) { // Model::setImuData normalises the orientation (model.h:150)
    Quat q = normalized(Quat{src[r * 4], src[r * 4 + 1], src[r * 4 + 2], src[r * 4 + 3]});
    robd[rob_index(r, f0 + 0, rpw, nf)] = q.w;
    robd[rob_index(r, f0 + 1, rpw, nf)] = q.x;
    robd[rob_index(r, f0 + 2, rpw, nf)] = q.y;
    robd[rob_index(r, f0 + 3, rpw, nf)] = q.z;
    return;
  }
  for (int k = 0; k < K; ++k) robd[rob_index(r, f0 + k, rpw, nf)] = src[r * K + k];
}
__global__ void gather_rob_kernel(double *dst, const double *robd, int rpw, int64_t n, int K, int f0) {
  constexpr int nf = RobotFields::COUNT;
  int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int k = 0; k < K; ++k) dst[r * K + k] = robd[rob_index(r, f0 + k, rpw, nf)];
}
__global__ void scatter_robi_kernel(const int32_t *src, int32_t *robi, int rpw, int64_t n, int f) {
  int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r < n) robi[rob_index(r, f, rpw, RobotFields::I_COUNT)] = src[r];
}
__global__ void count_walking_kernel(int32_t *count, const int32_t *robi, int rpw, int64_t n) {
  int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  bool walking = r < n && (robi[rob_index(r, RobotFields::I_WORD, rpw, RobotFields::I_COUNT)] & 3) != WS_STOPPED;
  unsigned long long m = __ballot(walking);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}
__global__ void fill_rob_kernel(double *robd, int rpw, int64_t n, int f0, int K, double v) {
  int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int k = 0; k < K; ++k) robd[rob_index(r, f0 + k, rpw, RobotFields::COUNT)] = v;
}
__global__ void fill_robi_kernel(int32_t *robi, int rpw, int64_t n, int f, int32_t v) {
  int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r < n) robi[rob_index(r, f, rpw, RobotFields::I_COUNT)] = v;
}
__global__ void gather_walk_state_kernel(int32_t *dst, const int32_t *robi, int rpw, int64_t n) {
  int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n) return;
  dst[r] = robi[rob_index(r, RobotFields::I_WORD, rpw, RobotFields::I_COUNT)] & 3;
}
// replicate the post-start-up state of one robot into every slot
__global__ void init_state_kernel(DevState st, const double *leg_template /*[L][nf]*/, const int32_t *legw_template /*[L]*/,
                                  const double *rob_template /*[nrf]*/, const int32_t *robi_template /*[nri]*/, int L, int nf,
                                  int nrf, int nri) {
  int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  int64_t n = st.n_robots;
  if (t < n * L) {
    int64_t rob = t / L;
    int leg = int(t - rob * L);
    int64_t slot = slot_of(rob, leg, L);
    for (int f = 0; f < nf; ++f) st.legd[leg_field_index(f, slot, st.n_slots)] = leg_template[leg * nf + f];
    st.legi[slot] = legw_template[leg];
  }
  if (t < n) {
    const int rpw = 64 / L;
    for (int f = 0; f < nrf; ++f) st.robd[rob_index(t, f, rpw, nrf)] = rob_template[f];
    for (int f = 0; f < nri; ++f) st.robi[rob_index(t, f, rpw, nri)] = robi_template[f];
  }
}

// ================================================================================================= engine

constexpr int64_t kPlanePadSlots = 192;

struct shc_engine {
  shc_params params;
  shc_tables tables;
  CycleParams cp;
  int L, NJ;
  int device;
  hipStream_t stream;
  int64_t n, n_waves, n_slots, n_rob_pad;
  int n_leg_fields;
  DevState st;
  void *d_consts;
  double *d_stage;     // staging for host <-> device conversions
  size_t stage_bytes;
  uint32_t features;
  uint32_t rt_flags; // RT_* facts passed to every launch
  int starting_up, startup_calls; // shc_engine_begin_direct_startup .. shc_engine_direct_startup
  SeqRobotState *d_seq;           // start-up / shut-down sequence state (shc_sequence.hpp), allocated by the first sequence call
  int pack_step, executing_transition, transition_calls; // PoseController::pack_step_ / executing_transition_ (pose_controller.h:294, :298)
  bool planner_mode = false;            // StateController::planner_mode_ (state_controller.h:337)
  bool plan_poser_tips_current = false; // no control cycle has run since the last shc_engine_execute_plan (SeqRobotState::poser_tip_from_plan holds)
  double *d_span = nullptr;             // SpanTable (rough terrain mode with a stance span modifier), rebuilt with the tables
  int half_steps = 0;                   // CycleLaunch::half_steps (development switch SHC_ROT_SPLIT = 0 / 1: never / always; unset: by launch size)
  bool span_dirty = true;
  bool step_remap_pending = false;      // an accepted step-frequency change waits for the next cycle (shc_engine_adjust_parameter)
  bool pose_params_held = false;        // the posing part of the next cycle still runs on the parameter values a just-adjusted parameter had (ditto)
  bool fresh_pose_controller = false;   // init_state for shc_engine_begin_sequence_startup: no direct start-up has run, the auto posers have not been called yet
  struct Resident *res = nullptr;       // resident mode (shc_resident.hpp)
  const double *bound_inputs[kBoundSets][BND_COUNT] = {}; // shc_engine_resident_bind_inputs: the caller's device arrays for direct posts
  // Large batches: a step is launched as two halves on two streams with no join between steps (see shc_engine_step).
  hipStream_t half_stream[2] = {nullptr, nullptr}; // the device's pair of split streams (split_streams()), once this engine has used them
  hipEvent_t ev_main = nullptr, ev_half[2] = {nullptr, nullptr};
  hipEvent_t ev_in[2] = {nullptr, nullptr}; // the half streams have read a device-resident input array (the engine's stream is ordered after them)
  double *k_out = nullptr;              // shc_engine_step_k: q / qd of each of the K cycles of the latest launch, [K][dof planes][n_slots] double2
  size_t k_out_bytes = 0;
  int k_out_cycles = 0;
  bool side_busy = false;               // launches are outstanding on the split streams that the engine's stream has not been ordered after
  bool main_dirty = true;               // work other than steps was enqueued on the engine's stream since the last split step
};
struct Resident;
static bool resident_active(const shc_engine *e);
static void resident_shutdown(shc_engine *e); // stop a running resident loop and free its buffers (shc_engine_destroy)
// While the resident kernel owns the engine's stream and state, every other entry point that would touch them is refused.
static int join_side(shc_engine *e);
static int flush_step_remap(shc_engine *e); // an accepted step-frequency change no cycle has consumed yet: map the phases now
#define SHC_BUSY_ONLY(e)                                                                                                       \
  do {                                                                                                                         \
    if ((e) && resident_active(e))                                                                                             \
      return fail(SHC_ERR_BUSY, "the engine is in resident mode: only shc_engine_resident_* calls are valid until shc_engine_resident_end"); \
  } while (0)
// ... and whatever an entry point enqueues on the engine's stream is ordered after the second halves of earlier split steps
#define SHC_BUSY_GUARD(e)                                                                                                      \
  do {                                                                                                                         \
    SHC_BUSY_ONLY(e);                                                                                                          \
    if (e) {                                                                                                                   \
      const int rc_join_ = join_side(e);                                                                                       \
      if (rc_join_ != SHC_OK) return rc_join_;                                                                                 \
    }                                                                                                                          \
  } while (0)

template <int L, int NJ>
static void build_shared_consts(const shc_params &p, const shc_tables &t, const CycleParams &cp, SharedConsts<L, NJ> &c) {
  memset(&c, 0, sizeof c);
  c.P = cp;
  const shc_step_cycle &step = t.step;
  for (int l = 0; l < L; ++l) {
    hostinit::fill_leg_const<NJ>(p, l, c.leg[l]);
    LegConst<NJ> &lc = c.leg[l];
    lc.neg_ratio = p.negation_transition_ratio[l];
    { // calculateStanceSpanChange (walk_controller.cpp:949-980): workspace.size() == 1 -> radius = workspace.at(0.0).at(bearing)
      double m = p.stance_span_modifier;
      const bool positive_y = p.stance_position[l][1] > 0.0;
      const int bearing = (positive_y ^ (m > 0.0)) ? 270 : 90;
      m *= positive_y ? 1.0 : -1.0;
      lc.span_shift = t.workspace_radius[l][bearing / 45] * m;
    }
    lc.phase_offset = t.phase_offset[l];
    int ns = p.pose_negation_phase_starts[l] * t.pose_normaliser, ne = p.pose_negation_phase_ends[l] * t.pose_normaliser;
    if (ns == 0) ns = t.pose_phase_length; // pose_controller.cpp:1723-1730
    if (ne == 0) ne = t.pose_phase_length;
    lc.neg_start = ns;
    lc.neg_end = ne;
    int msp = mod_i(step.stance_end - lc.phase_offset, step.period); // walk_controller.cpp:1026-1031
    if (step.stance_end == lc.phase_offset) msp = step.period;
    lc.first_stance_period = msp;
    lc.first_stance_iterations = int((double(msp) / step.period) / (step.frequency * p.time_delta));
    lc.first_stance_dt = 1.0 / lc.first_stance_iterations;
    lc.first_stride_scaler = double(msp) / double(step.stance_period);
    lc.starts_in_swing = (lc.phase_offset > step.swing_start && lc.phase_offset < step.swing_end) ? 1 : 0;
  }
  for (int it = 0; it < cp.swing_c_count; ++it) { // LegStepper::updatePhase progress (walk_controller.cpp:878-880) -> control input
    double sp = clampd(double(it + 1) / double(step.swing_end - step.swing_start), 0.0, 1.0) * cp.swing_progress_scaler;
    c.swing_c[it] = smooth_step(sp);
  }
  for (int b = 0; b < 9; ++b) {
    c.limit[b][0] = t.max_linear_speed[b];
    c.limit[b][1] = t.max_angular_speed[b];
    c.limit[b][2] = t.max_linear_acceleration[b];
    c.limit[b][3] = t.max_angular_acceleration[b];
  }
}

// Parameters::leg_DOF is per leg (parameters_and_states.h:298): the engine runs a robot on the kernels of its LONGEST leg; a shorter leg
// is padded behind its tip with locked zero-length joints (LegConst::jactive, hostinit::fill_leg_const), which leaves every formula of
// that leg as the reference evaluates it for its own joint count (the padded Jacobian columns are zero, their cost terms too).
static int max_dof(const shc_params &p) {
  int nj = 0;
  for (int l = 0; l < p.leg_count && l < SHC_MAX_LEGS; ++l) nj = p.leg_dof[l] > nj ? p.leg_dof[l] : nj;
  return nj;
}
static bool mixed_dof(const shc_params &p) {
  for (int l = 1; l < p.leg_count; ++l)
    if (p.leg_dof[l] != p.leg_dof[0]) return true;
  return false;
}
// Tip rotations are part of the state: legs of more than 3 joints with gravity-aligned tips / in rough terrain mode, or 3-joint legs that
// joint_control leg manipulation hands their FK tip pose to (walk_controller.cpp:677-690).
static int rotations_tracked(const shc_engine *e) { return (e->cp.gravity_aligned || e->cp.joint_control == 2) ? 1 : 0; }
// ... and the parameter block the engine keeps has neutral entries for the padding (joint arrays of the ABI are [legs][longest leg's DOF]:
// the padded joints read 0 and ignore what is written to them)
static shc_params normalised_params(const shc_params &in) {
  shc_params p = in;
  const int nj = max_dof(p);
  for (int l = 0; l < p.leg_count; ++l)
    for (int j = p.leg_dof[l]; j < nj; ++j) {
      p.joint[l][j] = shc_joint_params{0.0, 0.0, 0.0, 0.0, 1.0};
      p.link[l][j + 1] = shc_link_params{0.0, 0.0, 0.0, 0.0};
    }
  return p;
}
static void build_cycle_params(const shc_params &p, const shc_tables &t, uint32_t features, unsigned rt_flags, CycleParams &c) {
  memset(&c, 0, sizeof c);
  const shc_step_cycle &s = t.step;
  c.dt = p.time_delta;
  c.period = s.period;
  c.swing_period = s.swing_period;
  c.stance_period = s.stance_period;
  c.stance_end = s.stance_end;
  c.swing_start = s.swing_start;
  c.swing_end = s.swing_end;
  c.stance_start = s.stance_start;
  int swing_iterations = int((double(s.swing_period) / s.period) / (s.frequency * p.time_delta)); // walk_controller.cpp:1035
  swing_iterations = round_to_even_int(swing_iterations);
  c.swing_iterations = swing_iterations;
  c.swing_delta_t = 1.0 / (swing_iterations / 2.0);
  c.inv_dt = 1.0 / p.time_delta;
  c.dt_over_swing_dt = p.time_delta / c.swing_delta_t;
  c.stance_iterations = int((double(s.stance_period) / s.period) / (s.frequency * p.time_delta)); // :1040
  c.stance_dt = 1.0 / c.stance_iterations;
  double on_ground_ratio = double(s.stance_period) / s.period;                                    // :940
  c.stride_scale = on_ground_ratio / s.frequency;
  c.swing_height = p.swing_height;
  c.swing_width = p.swing_width;
  c.body_clearance = p.body_clearance;
  c.swing_progress_scaler = fmax(1.0, double(p.swing_phase) / p.phase_offset); // pose_controller.cpp:1103
  c.swing_c_count = (s.swing_end - s.swing_start) <= kSwingTable ? (s.swing_end - s.swing_start) : 0;
  c.swing_c_valid = 0;
  for (int it = 0; it < c.swing_c_count; ++it) { // progress is non-decreasing in the iteration: the valid ones form a prefix
    double sp = clampd(double(it + 1) / double(s.swing_end - s.swing_start), 0.0, 1.0) * c.swing_progress_scaler;
    if (sp >= 0 && sp <= 1.0) c.swing_c_valid = it + 1;
  }
  c.velocity_input_mode = p.velocity_input_mode;
  c.manual_posing = p.manual_posing;
  c.auto_posing = p.auto_posing;
  c.inclination_posing = p.inclination_posing;
  c.imu_posing = p.imu_posing;
  c.admittance_control = p.admittance_control;
  c.dynamic_stiffness = p.dynamic_stiffness;
  c.use_joint_effort = p.use_joint_effort;
  c.clamp_joint_positions = p.clamp_joint_positions;
  c.clamp_joint_velocities = p.clamp_joint_velocities;
  c.force_normal_touchdown = p.force_normal_touchdown;
  // Leg::calculateTipForce (model.cpp:667-708) filters the force of the measured joint torques into tip_force_calculated_.
  // Until a torque has been supplied (RT_EFFORT_LIVE) it filters zeros into a zero state: the estimate is exactly zero and
  // stays it, so the kernels without the estimate run (its planes are neither loaded nor stored).
  c.tip_force = (((features & SHC_FEAT_TIP_FORCE) || p.use_joint_effort) && (rt_flags & RT_EFFORT_LIVE)) ? 1 : 0;
  c.odometry = (features & SHC_FEAT_ODOMETRY) ? 1 : 0;
  c.gravity_aligned = hostinit::tips_rotation_tracked(p, max_dof(p)) ? 1 : 0;
  c.gravity_target = hostinit::tips_rotation_constrained(p, max_dof(p)) ? 1 : 0;
  c.joint_control = 0;
  if (p.leg_manipulation_mode == SHC_MANIPULATION_JOINT_CONTROL) {
    c.joint_control = 1;
    for (int l = 0; l < p.leg_count; ++l)
      if (p.leg_dof[l] == 3) c.joint_control = 2; // (walk_controller.cpp:677: "works only for 3DOF legs")
  }
  c.rough_terrain = p.rough_terrain_mode ? 1 : 0;
  c.tip_align = (p.gravity_aligned_tips && p.leg_dof[0] <= 3) ? 1 : 0; // pose_controller.cpp:849: LEG 0's joint count decides for the whole robot
  c.step_depth = p.step_depth;
  {
    V3 d = hostinit::gravity_aligned_direction();
    c.target_dir[0] = d.x;
    c.target_dir[1] = d.y;
    c.target_dir[2] = d.z;
  }
#ifdef SHC_ABLATE // development builds only (scripts/ablate.py): phase ablation mask
  if (const char *dbg = getenv("SHC_DEBUG_SKIP")) c.debug_skip = atoi(dbg);
#endif
  for (int i = 0; i < 3; ++i) {
    c.max_translation[i] = p.max_translation[i];
    c.max_rotation[i] = p.max_rotation[i];
  }
  c.max_translation_velocity = p.max_translation_velocity;
  c.max_rotation_velocity = p.max_rotation_velocity;
  c.pid_p = p.rotation_pid_gains[0];
  c.pid_i = p.rotation_pid_gains[1];
  c.pid_d = p.rotation_pid_gains[2];
  hostinit::admittance_map(p, c.adm_m00, c.adm_m01, c.adm_m10, c.adm_m11, c.adm_g0, c.adm_g1);
  c.force_gain = c.pose_force_gain = p.force_gain;
  c.pose_swing_height = p.swing_height;
  c.virtual_stiffness = p.virtual_stiffness;
  c.swing_stiffness_scaler = p.swing_stiffness_scaler;
  c.load_stiffness_scaler = p.load_stiffness_scaler;
  c.n_auto_posers = p.n_auto_posers;
  c.pose_phase_length = t.pose_phase_length;
  c.pose_sync = (p.pose_frequency == -1.0) ? 1 : 0;
  c.auto_pose_reference_leg = t.auto_pose_reference_leg;
  for (int i = 0; i < p.n_auto_posers && i < kMaxAutoPosers; ++i) {
    c.ap_start[i] = p.pose_phase_starts[i] * t.pose_normaliser;
    c.ap_end[i] = p.pose_phase_ends[i] * t.pose_normaliser;
    c.ap_amp[i][0] = p.x_amplitudes[i];
    c.ap_amp[i][1] = p.y_amplitudes[i];
    c.ap_amp[i][2] = p.z_amplitudes[i];
    c.ap_amp[i][3] = p.gravity_amplitudes[i];
    c.ap_amp[i][4] = p.roll_amplitudes[i];
    c.ap_amp[i][5] = p.pitch_amplitudes[i];
    c.ap_amp[i][6] = p.yaw_amplitudes[i];
  }
}

static int upload_consts(shc_engine *e);
static int validate_params(const shc_params *p, int *L, int *NJ) {
  if (!p) return fail(SHC_ERR_INVALID_ARG, "params is NULL");
  if (p->leg_count < 3 || p->leg_count > SHC_MAX_LEGS) return fail(SHC_ERR_INVALID_ARG, "leg_count must be 3..8");
  const int nj = max_dof(*p);
  for (int l = 0; l < p->leg_count; ++l)
    if (p->leg_dof[l] < 3 || p->leg_dof[l] > 5) return fail(SHC_ERR_UNSUPPORTED, "supported joints per leg (leg_dof): 3..5 DOF");
  // (gravity_aligned_tips on a robot whose legs differ in DOF: the reference decides per leg - legs of more than 3 joints constrain their tip
  //  rotation, walk_controller.cpp:37, :1195, model.cpp:880 - and by LEG 0's joint count whether the tip-align pose runs, over ALL legs,
  //  pose_controller.cpp:849; both are run-time tests on the padded chains here)
  if (p->rough_terrain_mode && !(p->touchdown_threshold >= p->liftoff_threshold))
    return fail(SHC_ERR_INVALID_ARG, "touchdown_threshold must be >= liftoff_threshold");
  if (p->n_auto_posers < 0 || p->n_auto_posers > kMaxAutoPosers) return fail(SHC_ERR_INVALID_ARG, "n_auto_posers out of range");
  if (!(p->time_delta > 0) || !(p->step_frequency > 0)) return fail(SHC_ERR_INVALID_ARG, "time_delta / step_frequency must be > 0");
  // gait integers feed integer divisions / modulos on the host (generateStepCycle) and on the device (phase arithmetic)
  if (p->stance_phase <= 0 || p->swing_phase <= 0 || p->phase_offset <= 0 || p->stance_phase > 4096 || p->swing_phase > 4096)
    return fail(SHC_ERR_INVALID_ARG, "gait: stance_phase, swing_phase and phase_offset must be positive (and <= 4096)");
  for (int l = 0; l < p->leg_count; ++l)
    if (p->offset_multiplier[l] < 0) return fail(SHC_ERR_INVALID_ARG, "gait: offset_multiplier must be >= 0");
  {
    const double raw = ((1.0 / p->step_frequency) / p->time_delta) / (double(p->swing_phase) / double(p->stance_phase + p->swing_phase));
    const double periods = raw / double(p->stance_phase + p->swing_phase);
    if (!(periods >= 0.0) || !(periods * (p->stance_phase + p->swing_phase) < double(LW_PHASE_MASK)))
      return fail(SHC_ERR_INVALID_ARG, "gait: the step period (1 / (step_frequency * time_delta), rounded) does not fit the phase field");
    if (int(periods) == 0) // roundToEvenInt(raw / base) == 0: period 0
      return fail(SHC_ERR_INVALID_ARG, "gait: step period rounds to 0 iterations (step_frequency * time_delta too large)");
  }
  if (p->auto_posing && p->pose_frequency != -1.0 && (!(p->pose_frequency > 0) || p->pose_phase_length <= 0))
    return fail(SHC_ERR_INVALID_ARG, "auto pose: pose_frequency must be -1 (step-cycle sync) or > 0 with pose_phase_length > 0");
  *L = p->leg_count;
  *NJ = nj;
  return SHC_OK;
}

template <int NJ>
static int generate_tables_nj(const shc_params *p, shc_tables *out) {
  return hostinit::generate_tables<NJ>(*p, *out) ? SHC_OK : fail(SHC_ERR_INVALID_ARG, "init chain failed (unreachable stance?)");
}

extern "C" int shc_abi_version(void) { return 6; } // 6: shc_engine_adjust_parameter; 5: shc_engine_step_k (K cycles per launch, each with its own inputs); 4: shc_cycle_inputs.direct (launch-free posts); 3: shc_leg_snapshot carries the stepper target tip direction; resident mode, join, auxiliary state
extern "C" int64_t shc_sizeof_params(void) { return (int64_t)sizeof(shc_params); }
extern "C" int64_t shc_sizeof_tables(void) { return (int64_t)sizeof(shc_tables); }

extern "C" int shc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char *shc_last_error(void) { return g_last_error.c_str(); }

extern "C" int shc_debug_plane_copy(int device, int64_t n_doubles, int reps) {
  if (n_doubles < 2 || (n_doubles & 1) || reps < 1) return fail(SHC_ERR_INVALID_ARG, "n_doubles must be even and >= 2, reps >= 1");
  HIP_TRY(hipSetDevice(device));
  double *a, *b;
  HIP_TRY(hipMalloc(&a, size_t(n_doubles) * 8));
  HIP_TRY(hipMalloc(&b, size_t(n_doubles) * 8));
  HIP_TRY(hipMemset(a, 0, size_t(n_doubles) * 8));
  const int64_t n_pairs = n_doubles / 2;
  for (int r = 0; r < reps; ++r)
    shc_plane_copy_kernel<<<dim3((unsigned)((n_pairs + 255) / 256)), dim3(256)>>>((const double2 *)a, (double2 *)b, n_pairs);
  HIP_TRY(hipDeviceSynchronize());
  (void)hipFree(a);
  (void)hipFree(b);
  return SHC_OK;
}

extern "C" int shc_stream_create(int device, void **stream) {
  if (!stream) return fail(SHC_ERR_INVALID_ARG, "stream is NULL");
  HIP_TRY(hipSetDevice(device));
  hipStream_t s;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void *)s;
  return SHC_OK;
}
extern "C" int shc_stream_destroy(int device, void *stream) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return SHC_OK;
}

extern "C" int shc_generate_tables(const shc_params *params, shc_tables *out) {
  int L, NJ;
  int rc = validate_params(params, &L, &NJ);
  if (rc != SHC_OK) return rc;
  if (!out) return fail(SHC_ERR_INVALID_ARG, "out is NULL");
  const shc_params np = normalised_params(*params);
  switch (NJ) {
    case 3: return generate_tables_nj<3>(&np, out);
    case 4: return generate_tables_nj<4>(&np, out);
    case 5: return generate_tables_nj<5>(&np, out);
  }
  return fail(SHC_ERR_UNSUPPORTED, "dof");
}

// ---- init chain on the device: the same host + device functions as shc_generate_tables, fanned out over morphologies
template <int NJ>
__global__ void init_chain_legs_kernel(const shc_params *params, shc_tables *tables, const int32_t *status, int64_t count) {
  int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; // one thread per (morphology, leg, bearing)
  int64_t m = t / (SHC_MAX_LEGS * 8);
  int r = int(t - m * (SHC_MAX_LEGS * 8));
  int l = r / 8, b = r % 8 + 1;
  if (m >= count || status[m] != SHC_OK) return;
  const shc_params &p = params[m];
  // (a robot whose legs differ in DOF runs on the chain of its longest leg: the host has padded the shorter legs - normalised_params - exactly as
  //  shc_generate_tables does, so Model::generateWorkspaces' per-leg search, model.cpp:309-510, evaluates each leg's own joints)
  int nj = 0;
  for (int k = 0; k < p.leg_count && k < SHC_MAX_LEGS; ++k) nj = p.leg_dof[k] > nj ? p.leg_dof[k] : nj;
  if (nj != NJ || l >= p.leg_count) return;
  hostinit::generate_tables_leg<NJ>(p, l, tables[m], b, b);
}
__global__ void init_chain_head_kernel(const shc_params *params, shc_tables *tables, int32_t *status, int64_t count) {
  int64_t m = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (m >= count || status[m] != SHC_OK) return;
  if (!hostinit::generate_tables_head(params[m], tables[m])) status[m] = SHC_ERR_INVALID_ARG;
}
__global__ void init_chain_tail_kernel(const shc_params *params, shc_tables *tables, const int32_t *status, int64_t count) {
  int64_t m = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (m >= count || status[m] != SHC_OK) return;
  hostinit::generate_tables_tail(params[m], tables[m]);
}

extern "C" int shc_generate_tables_batch(const shc_params *params, int64_t count, shc_tables *out, int32_t *status, int device) {
  if (!params || !out || count < 1) return fail(SHC_ERR_INVALID_ARG, "params / out NULL or count < 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(SHC_ERR_NO_DEVICE, "no HIP device visible");
  HIP_TRY(hipSetDevice(device));
  std::vector<int32_t> st(count);
  std::vector<shc_params> np(static_cast<size_t>(count));
  for (int64_t i = 0; i < count; ++i) { // parameter screening is host logic (same rules as shc_engine_create)
    int L, NJ;
    st[i] = validate_params(&params[i], &L, &NJ);
    np[size_t(i)] = st[i] == SHC_OK ? normalised_params(params[i]) : params[i]; // (neutral entries for the padded joints of legs shorter than the robot's longest)
  }
  shc_params *d_p = nullptr;
  shc_tables *d_t = nullptr;
  int32_t *d_s = nullptr;
  auto release = [&]() {
    (void)hipFree(d_p);
    (void)hipFree(d_t);
    (void)hipFree(d_s);
  };
  HIP_TRY_OR(hipMalloc(&d_p, size_t(count) * sizeof(shc_params)), release());
  HIP_TRY_OR(hipMalloc(&d_t, size_t(count) * sizeof(shc_tables)), release());
  HIP_TRY_OR(hipMalloc(&d_s, size_t(count) * 4), release());
  HIP_TRY_OR(hipMemcpy(d_p, np.data(), size_t(count) * sizeof(shc_params), hipMemcpyHostToDevice), release());
  HIP_TRY_OR(hipMemcpy(d_s, st.data(), size_t(count) * 4, hipMemcpyHostToDevice), release());
  HIP_TRY_OR(hipMemset(d_t, 0, size_t(count) * sizeof(shc_tables)), release());
  const unsigned gm = (unsigned)((count + 63) / 64), gl = (unsigned)((count * SHC_MAX_LEGS * 8 + 63) / 64);
  init_chain_head_kernel<<<dim3(gm), dim3(64)>>>(d_p, d_t, d_s, count);
  init_chain_legs_kernel<3><<<dim3(gl), dim3(64)>>>(d_p, d_t, d_s, count);
  init_chain_legs_kernel<4><<<dim3(gl), dim3(64)>>>(d_p, d_t, d_s, count);
  init_chain_legs_kernel<5><<<dim3(gl), dim3(64)>>>(d_p, d_t, d_s, count);
  init_chain_tail_kernel<<<dim3(gm), dim3(64)>>>(d_p, d_t, d_s, count);
  HIP_TRY_OR(hipGetLastError(), release());
  HIP_TRY_OR(hipDeviceSynchronize(), release());
  HIP_TRY_OR(hipMemcpy(out, d_t, size_t(count) * sizeof(shc_tables), hipMemcpyDeviceToHost), release());
  HIP_TRY_OR(hipMemcpy(st.data(), d_s, size_t(count) * 4, hipMemcpyDeviceToHost), release());
  release();
  for (int64_t i = 0; i < count; ++i) {
    if (st[i] != SHC_OK) memset(static_cast<void *>(&out[i]), 0, sizeof(shc_tables));
    if (status) status[i] = st[i];
  }
  return SHC_OK;
}

#define SHC_DISPATCH(L_, NJ_, CALL)                                   \
  do {                                                                \
    if (L_ == 3 && NJ_ == 3) { CALL(3, 3); }                          \
    else if (L_ == 4 && NJ_ == 3) { CALL(4, 3); }                     \
    else if (L_ == 4 && NJ_ == 4) { CALL(4, 4); }                     \
    else if (L_ == 4 && NJ_ == 5) { CALL(4, 5); }                     \
    else if (L_ == 5 && NJ_ == 3) { CALL(5, 3); }                     \
    else if (L_ == 6 && NJ_ == 3) { CALL(6, 3); }                     \
    else if (L_ == 6 && NJ_ == 4) { CALL(6, 4); }                     \
    else if (L_ == 6 && NJ_ == 5) { CALL(6, 5); }                     \
    else if (L_ == 7 && NJ_ == 3) { CALL(7, 3); }                     \
    else if (L_ == 8 && NJ_ == 3) { CALL(8, 3); }                     \
    else if (L_ == 8 && NJ_ == 4) { CALL(8, 4); }                     \
    else if (L_ == 8 && NJ_ == 5) { CALL(8, 5); }                     \
    else return fail(SHC_ERR_UNSUPPORTED, "no kernel specialisation for this (legs, dof)"); \
  } while (0)

// rough terrain mode with a stance span modifier: LegStepper::calculateStanceSpanChange interpolates the layered workspace at the
// default tip's height at every swing / stance start - the planes of every leg's workspace, searched again from the default
// configuration the tables hold (the init chain itself only keeps the plane at height 0)
template <int NJ>
static int build_span_table(shc_engine *e) {
  e->span_dirty = false;
  const shc_params &p = e->params;
  if (!(p.rough_terrain_mode && p.stance_span_modifier != 0.0)) {
    e->st.span = nullptr;
    return SHC_OK;
  }
  std::vector<double> tab(size_t(e->L) * SpanTable::kStride, 0.0);
  static_assert(SpanTable::kPlanes >= hostinit::kMaxWorkspacePlanes, "span table holds every plane of the layered workspace");
  for (int l = 0; l < e->L; ++l) {
    hostinit::LayeredPlanes planes;
    shc_tables scratch = e->tables;
    hostinit::generate_tables_leg<NJ>(p, l, scratch, 1, 8, e->tables.default_joint_position[l], &planes);
    double m = p.stance_span_modifier;
    const bool positive_y = p.stance_position[l][1] > 0.0;
    const int bearing = (positive_y ^ (m > 0.0)) ? 270 : 90;
    m *= positive_y ? 1.0 : -1.0;
    double *t = &tab[size_t(l) * SpanTable::kStride];
    t[0] = planes.n;
    t[1] = m;
    for (int k = 0; k < planes.n; ++k) t[2 + 2 * k] = planes.height[k], t[3 + 2 * k] = planes.radius[k][bearing / 45];
  }
  if (!e->d_span) HIP_TRY(hipMalloc(&e->d_span, tab.size() * 8));
  HIP_TRY(hipMemcpyAsync(e->d_span, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->st.span = e->d_span;
  return SHC_OK;
}

static int upload_consts(shc_engine *e) {
  if (e->span_dirty) {
    const int rc = e->NJ == 3 ? build_span_table<3>(e) : (e->NJ == 4 ? build_span_table<4>(e) : build_span_table<5>(e));
    if (rc != SHC_OK) return rc;
  }
#define CALL(L_, NJ_)                                                                                   \
  {                                                                                                     \
    SharedConsts<L_, NJ_> c;                                                                            \
    build_shared_consts<L_, NJ_>(e->params, e->tables, e->cp, c);                                              \
    if (!e->d_consts) HIP_TRY(hipMalloc(&e->d_consts, sizeof c));                                       \
    HIP_TRY(hipMemcpyAsync(e->d_consts, &c, sizeof c, hipMemcpyHostToDevice, e->stream));               \
    HIP_TRY(hipStreamSynchronize(e->stream));                                                           \
  }
  SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
  return SHC_OK;
}

template <int NJ>
static void build_templates(const shc_engine *e, std::vector<double> &legt, std::vector<int32_t> &legw, std::vector<double> &robt,
                            std::vector<int32_t> &robi) {
  using F = Fields<NJ>;
  using R = RobotFields;
  const int nf = F::COUNT;
  legt.assign(size_t(e->L) * nf, 0.0);
  legw.assign(e->L, 0);
  for (int l = 0; l < e->L; ++l) {
    double *t = &legt[size_t(l) * nf];
    for (int j = 0; j < NJ; ++j) {
      t[F::Q + j] = e->tables.default_joint_position[l][j];
      // Joint::current_position_ until a joint state message arrives: Leg::init(true) copied the initial default positions
      // (model.cpp:292-296, :1038) and nothing on this path writes it again
      t[F::MEAS_Q + j] = clampd(0.0, e->params.joint[l][j].min, e->params.joint[l][j].max);
    }
    double sx = e->params.stance_position[l][0], sy = e->params.stance_position[l][1];
    // LegStepper constructor (walk_controller.cpp:795-819): everything at the identity tip pose
    const int at_identity[] = {F::TIP, F::SORG, F::TORG, F::DFLT, F::TARG};
    for (int f : at_identity) {
      t[f] = sx;
      t[f + 1] = sy;
      t[f + 2] = 0.0;
    }
    // model tip of the start-up configuration (for getters before the first cycle)
    LegConst<NJ> lc;
    hostinit::fill_leg_const<NJ>(e->params, l, lc);
    double q[NJ];
    for (int j = 0; j < NJ; ++j) q[j] = t[F::Q + j];
    Chain<NJ> ch;
    fk_chain<NJ>(lc, q, ch);
    V3 tip = tip_robot_frame(lc, ch.pe);
    t[F::MODEL_TIP] = tip.x;
    t[F::MODEL_TIP + 1] = tip.y;
    t[F::MODEL_TIP + 2] = tip.z;
    // step_state STANCE, phase 0, progress "none" (walk_controller.h:493-501)
    legw[l] = SS_STANCE | (PM_NONE << LW_PM_SHIFT);
    if (hostinit::tips_rotation_constrained(e->params, e->params.leg_dof[l])) { // current / origin / target tip poses start at the identity tip pose (:800-803)
      V3 d = hostinit::gravity_aligned_direction();
      t[F::ORG_DIR] = t[F::CUR_DIR] = t[F::TARG_DIR] = d.x;
      t[F::ORG_DIR + 1] = t[F::CUR_DIR + 1] = t[F::TARG_DIR + 1] = d.y;
      t[F::ORG_DIR + 2] = t[F::CUR_DIR + 2] = t[F::TARG_DIR + 2] = d.z;
      legw[l] |= LW_ROTDEF | LW_TARGROT;
    } else if (hostinit::tips_rotation_tracked(e->params, NJ)) { // ... at UNDEFINED_ROTATION, whose rotated x axis is x itself (also a <= 3-joint leg next to longer ones)
      t[F::ORG_DIR] = t[F::CUR_DIR] = 1.0;
    }
  }
  robt.assign(R::COUNT, 0.0);
  robt[R::PNORM + 2] = 1.0;
  robt[R::PNORM_PREV + 2] = 1.0;
  robt[R::OWPP + 2] = e->params.body_clearance; // pose_controller.cpp:39-40
  robt[R::OWPP + 3] = 1.0;
  robt[R::MPOSE + 3] = 1.0;
  robt[R::IMUQ + 0] = 1.0;
  robt[R::APREV + 0] = 1.0;
  robt[R::CPOSE + 2] = e->params.body_clearance;
  robt[R::CPOSE + 3] = 1.0;
  robt[R::WPP + 2] = e->params.body_clearance;
  robt[R::WPP + 3] = 1.0;
  robt[R::ODOM + 2] = 1.0; // identity (walk_controller.cpp:28): x, y, qw, qz
  robt[R::TALIGN + 3] = robt[R::OTALIGN + 3] = 1.0; // identity (pose_controller.h:132-133)
  robi.assign(R::I_COUNT, 0);
  robi[R::I_WORD] = WS_STOPPED | (PS_POSING_COMPLETE << RW_APS_SHIFT);
  // Auto posing on its own clock (pose_frequency != -1): PoseController::updateCurrentPose already runs in every loop of
  // the direct start-up (state_controller.cpp:165-167 with robot_state PACKED; one loop per transitionConfiguration step,
  // pose_controller.cpp:1476-1567), and every call advances the pose phase counter and lets the posers latch on
  // (start_check is unconditional without step-cycle sync, :1359-1371).  Replay those calls for the flags / counter the
  // first RUNNING cycle starts from; the walk state is STOPPED throughout, hence auto_posing_state STOP_POSING.
  if (e->params.auto_posing && e->params.pose_frequency != -1.0 && e->tables.pose_phase_length > 0 && !e->fresh_pose_controller) { // (a start-up SEQUENCE begins with a fresh PoseController:
    int calls = round_to_int(e->params.time_to_start / e->params.time_delta);                                                    //  its loops run the pose themselves, sequence_launch)
    if (calls < 1) calls = 1;
    const int len = e->tables.pose_phase_length, nrm = e->tables.pose_normaliser;
    int phase_counter = 0, flags = 0;
    for (int c = 0; c < calls; ++c) {
      const int master_phase = phase_counter;
      phase_counter = (phase_counter + 1) % len;
      for (int i = 0; i < e->params.n_auto_posers && i < kMaxAutoPosers; ++i) {
        int fl = (flags >> (4 * i)) & 15;
        bool start_check = fl & 1, end1 = fl & 2, end2 = fl & 4, allow = fl & 8;
        int phase = master_phase, sp = e->params.pose_phase_starts[i] * nrm, ep = e->params.pose_phase_ends[i] * nrm;
        if (sp > ep) {
          ep += len;
          if (phase < sp) phase += len;
        }
        start_check = true;
        end1 = end1 || phase == sp;          // auto_posing_state == STOP_POSING
        end2 = end2 || (phase == ep && end1);
        if (!allow && start_check) {
          allow = true;
          end1 = end2 = false;
        }
        flags = (flags & ~(15 << (4 * i))) | ((int(start_check) | int(end1) << 1 | int(end2) << 2 | int(allow) << 3) << (4 * i));
      }
    }
    robi[R::I_POSE_PHASE] = phase_counter;
    robi[R::I_APOSER] = flags;
    // ... and each leg's negation latch (LegPoser::updateAutoPose :1716-1731; step state STANCE throughout the start-up)
    for (int l = 0; l < e->L; ++l) {
      int ns = e->params.pose_negation_phase_starts[l] * nrm, ne = e->params.pose_negation_phase_ends[l] * nrm;
      if (ns == 0) ns = len;
      if (ne == 0) ne = len;
      bool neg = false;
      for (int c = 0; c < calls; ++c) {
        int sp = ns, ep = ne, np = c % len;
        if (sp > ep) {
          ep += len;
          if (np < sp) np += len;
        }
        if (np == sp) neg = true;
        if (np < sp || np > ep) neg = false;
      }
      if (neg) legw[l] |= LW_NEG;
    }
  }
}

static int init_state(shc_engine *e) {
  std::vector<double> legt, robt;
  std::vector<int32_t> legw, robi;
  switch (e->NJ) {
    case 3: build_templates<3>(e, legt, legw, robt, robi); break;
    case 4: build_templates<4>(e, legt, legw, robt, robi); break;
    case 5: build_templates<5>(e, legt, legw, robt, robi); break;
  }
  double *d_legt = nullptr, *d_robt = nullptr;
  int32_t *d_legw = nullptr, *d_robi = nullptr;
  auto release = [&]() {
    (void)hipFree(d_legt);
    (void)hipFree(d_robt);
    (void)hipFree(d_legw);
    (void)hipFree(d_robi);
  };
  HIP_TRY_OR(hipMalloc(&d_legt, legt.size() * 8), release());
  HIP_TRY_OR(hipMalloc(&d_robt, robt.size() * 8), release());
  HIP_TRY_OR(hipMalloc(&d_legw, legw.size() * 4), release());
  HIP_TRY_OR(hipMalloc(&d_robi, robi.size() * 4), release());
  HIP_TRY_OR(hipMemcpyAsync(d_legt, legt.data(), legt.size() * 8, hipMemcpyHostToDevice, e->stream), release());
  HIP_TRY_OR(hipMemcpyAsync(d_robt, robt.data(), robt.size() * 8, hipMemcpyHostToDevice, e->stream), release());
  HIP_TRY_OR(hipMemcpyAsync(d_legw, legw.data(), legw.size() * 4, hipMemcpyHostToDevice, e->stream), release());
  HIP_TRY_OR(hipMemcpyAsync(d_robi, robi.data(), robi.size() * 4, hipMemcpyHostToDevice, e->stream), release());
  HIP_TRY_OR(hipMemsetAsync(e->st.legd, 0, size_t(e->n_leg_fields) * e->n_slots * 8, e->stream), release());
  HIP_TRY_OR(hipMemsetAsync(e->st.legi, 0, size_t(e->n_slots) * 4, e->stream), release());
  int64_t threads = e->n * e->L;
  init_state_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(
      e->st, d_legt, d_legw, d_robt, d_robi, e->L, e->n_leg_fields, RobotFields::COUNT, RobotFields::I_COUNT);
  HIP_TRY_OR(hipGetLastError(), release());
  HIP_TRY_OR(hipStreamSynchronize(e->stream), release());
  release();
  return SHC_OK;
}

static int engine_create(const shc_params *params, const shc_tables *tables, int64_t n_instances, int device, void *stream,
                         shc_engine **out);
extern "C" int shc_engine_create(const shc_params *params, int64_t n_instances, int device, void *stream, shc_engine **out) {
  return engine_create(params, nullptr, n_instances, device, stream, out);
}
extern "C" int shc_engine_create_with_tables(const shc_params *params, const shc_tables *tables, int64_t n_instances, int device,
                                             void *stream, shc_engine **out) {
  if (!tables) return fail(SHC_ERR_INVALID_ARG, "tables is NULL");
  if (tables->step.period <= 0) return fail(SHC_ERR_INVALID_ARG, "tables were not generated (step period 0)");
  return engine_create(params, tables, n_instances, device, stream, out);
}
static int engine_create(const shc_params *params, const shc_tables *tables, int64_t n_instances, int device, void *stream,
                         shc_engine **out) {
  int L, NJ;
  int rc = validate_params(params, &L, &NJ);
  if (rc != SHC_OK) return rc;
  if (!out || n_instances < 1) return fail(SHC_ERR_INVALID_ARG, "n_instances must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(SHC_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(SHC_ERR_INVALID_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  shc_engine *e = new shc_engine();
  memset(static_cast<void *>(e), 0, sizeof *e);
  e->span_dirty = e->main_dirty = true; // (the memset above also clears the members' default initialisers)
  e->params = normalised_params(*params); // (legs shorter than the robot's longest: neutral entries for their padded joints)
  e->L = L;
  e->NJ = NJ;
  e->device = device;
  e->stream = (hipStream_t)stream;
  e->n = n_instances;
  e->features = SHC_FEAT_TIP_FORCE | SHC_FEAT_ODOMETRY;
  if (const char *v = getenv("SHC_ROT_SPLIT")) e->half_steps = atoi(v) > 0 ? 1 : -1;
  if (tables) {
    e->tables = *tables;
    e->span_dirty = true;
  } else {
    rc = shc_generate_tables(params, &e->tables);
    if (rc != SHC_OK) {
      delete e;
      return rc;
    }
  }
  build_cycle_params(e->params, e->tables, e->features, e->rt_flags, e->cp);
  const int rpw = 64 / L;
  e->n_waves = (n_instances + rpw - 1) / rpw;
  // Plane stride: 64 slots per wave + 3 KiB of padding.  A wave touches the same 1 KiB offset of ~30 planes; with a stride that
  // is a multiple of a large power of two (8 192 hexapod waves: 8 MiB, 16 384 octopod waves: 16 MiB) those accesses fall on
  // the same HBM channels.  Measured with 192 slots of padding: 81 920 hexapods 62.3 -> 53.1 us per launch, 131 072 octopods
  // 135.7 -> 129.2, 65 536 hexapods 45.8 -> 44.4, neutral at 4 096 (64 ... 65 600 slots all give the same).
  e->n_slots = e->n_waves * 64 + kPlanePadSlots;
  e->n_rob_pad = e->n_waves * rpw; // robots incl. the padding of the last wave's tile
  e->n_leg_fields = NJ == 3 ? Fields<3>::COUNT : (NJ == 4 ? Fields<4>::COUNT : Fields<5>::COUNT);
  e->st.n_slots = e->n_slots;
  e->st.n_rob_pad = e->n_rob_pad;
  e->st.n_robots = e->n;
  HIP_TRY_OR(hipMalloc(&e->st.legd, size_t(e->n_leg_fields) * e->n_slots * 8), shc_engine_destroy(e));
  HIP_TRY_OR(hipMalloc(&e->st.legi, size_t(e->n_slots) * 4), shc_engine_destroy(e));
  // robot state: one contiguous [field][rpw] tile per wave
  HIP_TRY_OR(hipMalloc(&e->st.robd, size_t(RobotFields::COUNT) * e->n_rob_pad * 8), shc_engine_destroy(e));
  HIP_TRY_OR(hipMalloc(&e->st.robi, size_t(RobotFields::I_COUNT) * e->n_rob_pad * 4), shc_engine_destroy(e));
  HIP_TRY_OR(hipMemsetAsync(e->st.robd, 0, size_t(RobotFields::COUNT) * e->n_rob_pad * 8, e->stream), shc_engine_destroy(e));
  HIP_TRY_OR(hipMemsetAsync(e->st.robi, 0, size_t(RobotFields::I_COUNT) * e->n_rob_pad * 4, e->stream), shc_engine_destroy(e));
  e->stage_bytes = size_t(e->n) * L * (NJ > 3 ? NJ : 3) * 8 + size_t(e->n) * 8 * 8 + size_t(SHC_MAX_LEGS) * SHC_MAX_JOINTS * 8 + 4096;
  HIP_TRY_OR(hipMalloc(&e->d_stage, e->stage_bytes), shc_engine_destroy(e));
  rc = upload_consts(e);
  if (rc == SHC_OK) rc = init_state(e);
  // The reference's loop() that enters RUNNING also executes runningState() once with the (zero) inputs present at
  // that moment (state_controller.cpp:277-281 then :189-192): the engine's initial state includes that cycle.
  if (rc == SHC_OK) rc = shc_engine_step(e, 1);
  if (rc == SHC_OK) rc = shc_engine_synchronize(e);
  if (rc != SHC_OK) {
    shc_engine_destroy(e);
    return rc;
  }
  *out = e;
  return SHC_OK;
}

extern "C" int shc_engine_destroy(shc_engine *e) {
  if (!e) return SHC_OK;
  (void)hipSetDevice(e->device);
  resident_shutdown(e);
  for (hipStream_t hs : e->half_stream) // (the halves of split steps: nothing may still be running on the buffers freed below)
    if (hs) (void)hipStreamSynchronize(hs);
  (void)hipStreamSynchronize(e->stream);
  (void)hipFree(e->st.legd);
  (void)hipFree(e->st.legi);
  (void)hipFree(e->st.robd);
  (void)hipFree(e->st.robi);
  (void)hipFree(e->st.ext);
  (void)hipFree(e->st.manual);
  (void)hipFree(e->d_seq);
  (void)hipFree(e->d_consts);
  (void)hipFree(e->d_stage);
  (void)hipFree(e->d_span);
  (void)hipFree(e->k_out);
  if (e->half_stream[0]) { // (the streams belong to the process-wide pair)
    (void)hipEventDestroy(e->ev_main);
    (void)hipEventDestroy(e->ev_half[0]);
    (void)hipEventDestroy(e->ev_half[1]);
    for (hipEvent_t ev : e->ev_in)
      if (ev) (void)hipEventDestroy(ev);
  }
  delete e;
  return SHC_OK;
}

extern "C" int shc_engine_set_stream(shc_engine *e, void *stream) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  e->stream = (hipStream_t)stream;
  return SHC_OK;
}

extern "C" int shc_engine_set_features(shc_engine *e, uint32_t features) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  e->features = features;
  build_cycle_params(e->params, e->tables, e->features, e->rt_flags, e->cp);
  return upload_consts(e);
}

extern "C" int shc_engine_get_tables(const shc_engine *e, shc_tables *out) {
  if (!e || !out) return fail(SHC_ERR_INVALID_ARG, "NULL argument");
  *out = e->tables;
  return SHC_OK;
}

extern "C" int64_t shc_engine_instances(const shc_engine *e) { return e ? e->n : 0; }

// ---- input / output plumbing
static int to_device(shc_engine *e, const double *src, size_t count, int on_device, const double **dptr) {
  if (on_device) {
    *dptr = src;
    return SHC_OK;
  }
  if (count * 8 > e->stage_bytes) return fail(SHC_ERR_INVALID_ARG, "staging buffer too small");
  HIP_TRY(hipMemcpyAsync(e->d_stage, src, count * 8, hipMemcpyHostToDevice, e->stream));
  *dptr = e->d_stage;
  return SHC_OK;
}

// While split steps are in flight (large batches, shc_engine_step) a device-resident input array is scattered by EACH HALF'S OWN
// stream - ordered after that half's earlier steps and before its later ones by stream order alone, so that new inputs cost no
// join (a join drains both halves and costs the overlap the split exists for).  The caller's array is ready on the engine's
// stream: both half streams first wait for that.
static bool split_inputs(const shc_engine *e, int on_device) { return on_device && e->side_busy; }
static int split_inputs_begin(shc_engine *e) {
  HIP_TRY(hipEventRecord(e->ev_main, e->stream));
  HIP_TRY(hipStreamWaitEvent(e->half_stream[0], e->ev_main, 0));
  HIP_TRY(hipStreamWaitEvent(e->half_stream[1], e->ev_main, 0));
  return SHC_OK;
}
// ... and once both halves have read the caller's array the engine's stream is ordered after those reads (events, no host wait, no
// join of the steps): whatever the caller queues on the engine's stream next - a copy_ into the same tensor, a free that the caching
// allocator turns into a reuse - cannot overtake the scatter kernels, which may sit behind several queued steps of their half.
static int split_inputs_end(shc_engine *e) {
  for (int h = 0; h < 2; ++h) {
    if (!e->ev_in[h]) HIP_TRY(hipEventCreateWithFlags(&e->ev_in[h], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(e->ev_in[h], e->half_stream[h]));
    HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_in[h], 0));
  }
  return SHC_OK;
}
static int64_t split_first_instance_of_second_half(const shc_engine *e) { // as shc_engine_step cuts the batch
  const int64_t waves_per_block = e->n_waves < 1536 ? 1 : 2;
  const int64_t half = ((e->n_waves / 2 + waves_per_block - 1) / waves_per_block) * waves_per_block;
  const int64_t r = half * (64 / e->L);
  return r < e->n ? r : e->n;
}

static int scatter_rob(shc_engine *e, const double *src, int K, int f0, int on_device, int normalize_quat = 0) {
  if (!src) return SHC_OK;
  HIP_TRY(hipSetDevice(e->device));
  if (split_inputs(e, on_device)) {
    const int rc = split_inputs_begin(e);
    if (rc != SHC_OK) return rc;
    const int64_t mid = split_first_instance_of_second_half(e);
    scatter_rob_kernel<<<dim3((unsigned)((mid + 255) / 256)), dim3(256), 0, e->half_stream[0]>>>(src, e->st.robd, 64 / e->L, mid, K, f0, normalize_quat, 0);
    if (e->n > mid)
      scatter_rob_kernel<<<dim3((unsigned)((e->n - mid + 255) / 256)), dim3(256), 0, e->half_stream[1]>>>(src, e->st.robd, 64 / e->L, e->n, K, f0, normalize_quat, mid);
    HIP_TRY(hipGetLastError());
    return split_inputs_end(e);
  }
  const double *d;
  int rc = to_device(e, src, size_t(e->n) * K, on_device, &d);
  if (rc != SHC_OK) return rc;
  scatter_rob_kernel<<<dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, e->stream>>>(d, e->st.robd, 64 / e->L, e->n, K, f0,
                                                                                      normalize_quat);
  HIP_TRY(hipGetLastError());
  if (!on_device) HIP_TRY(hipStreamSynchronize(e->stream)); // the staging buffer is reused by the next call
  return SHC_OK;
}

static int scatter_leg(shc_engine *e, const double *src, int K, int f0, int on_device) {
  if (!src) return SHC_OK;
  HIP_TRY(hipSetDevice(e->device));
  if (split_inputs(e, on_device)) {
    const int rc = split_inputs_begin(e);
    if (rc != SHC_OK) return rc;
    const int64_t mid = split_first_instance_of_second_half(e);
    scatter_leg_kernel<<<dim3((unsigned)((mid * e->L + 255) / 256)), dim3(256), 0, e->half_stream[0]>>>(src, e->st.legd, e->n_slots, mid, e->L, K, f0, 0);
    if (e->n > mid)
      scatter_leg_kernel<<<dim3((unsigned)(((e->n - mid) * e->L + 255) / 256)), dim3(256), 0, e->half_stream[1]>>>(src, e->st.legd, e->n_slots, e->n, e->L, K, f0, mid);
    HIP_TRY(hipGetLastError());
    return split_inputs_end(e);
  }
  const double *d;
  int rc = to_device(e, src, size_t(e->n) * e->L * K, on_device, &d);
  if (rc != SHC_OK) return rc;
  int64_t threads = e->n * e->L;
  scatter_leg_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(d, e->st.legd, e->n_slots, e->n, e->L, K, f0);
  HIP_TRY(hipGetLastError());
  if (!on_device) HIP_TRY(hipStreamSynchronize(e->stream));
  return SHC_OK;
}

static int gather_leg(shc_engine *e, double *dst, int K, int f0, int on_device) {
  if (!dst) return SHC_OK;
  HIP_TRY(hipSetDevice(e->device));
  int64_t threads = e->n * e->L;
  double *d = on_device ? dst : e->d_stage;
  gather_leg_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(d, e->st.legd, e->n_slots, e->n, e->L, K, f0);
  HIP_TRY(hipGetLastError());
  if (!on_device) {
    HIP_TRY(hipMemcpyAsync(dst, d, size_t(threads) * K * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return SHC_OK;
}

static int gather_rob(shc_engine *e, double *dst, int K, int f0, int on_device) {
  if (!dst) return SHC_OK;
  HIP_TRY(hipSetDevice(e->device));
  double *d = on_device ? dst : e->d_stage;
  gather_rob_kernel<<<dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, e->stream>>>(d, e->st.robd, 64 / e->L, e->n, K, f0);
  HIP_TRY(hipGetLastError());
  if (!on_device) {
    HIP_TRY(hipMemcpyAsync(dst, d, size_t(e->n) * K * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return SHC_OK;
}

static int derive_tips(shc_engine *e);
#define LEG_FIELD(e, NAME) ((e)->NJ == 3 ? Fields<3>::NAME : ((e)->NJ == 4 ? Fields<4>::NAME : Fields<5>::NAME))

extern "C" int shc_engine_set_velocity(shc_engine *e, const double *linear_xy, const double *angular, int on_device) {
  SHC_BUSY_ONLY(e);
  if (e && !split_inputs(e, on_device)) {
    const int rc_join_ = join_side(e);
    if (rc_join_ != SHC_OK) return rc_join_;
  }
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  int rc = scatter_rob(e, linear_xy, 2, RobotFields::VIN, on_device);
  if (rc != SHC_OK) return rc;
  return scatter_rob(e, angular, 1, RobotFields::WIN, on_device);
}

extern "C" int shc_engine_set_imu(shc_engine *e, const double *orientation_wxyz, const double *angular_velocity, int on_device) {
  SHC_BUSY_ONLY(e);
  if (e && !split_inputs(e, on_device)) {
    const int rc_join_ = join_side(e);
    if (rc_join_ != SHC_OK) return rc_join_;
  }
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  int rc = scatter_rob(e, orientation_wxyz, 4, RobotFields::IMUQ, on_device, 1);
  if (rc != SHC_OK) return rc;
  return scatter_rob(e, angular_velocity, 3, RobotFields::GYRO, on_device);
}

// Leg::touchdownDetection (model.cpp:712-722), run by tipStatesCallback right after it stored the measured force
// (state_controller.cpp:1644-1646): the step plane is where the tip is when the force first exceeds the touchdown threshold.
template <int L, int NJ>
__global__ void touchdown_detection_kernel(DevState st, const SharedConsts<L, NJ> *gc, double touchdown_threshold, double liftoff_threshold) {
  using FD = Fields<NJ>;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= st.n_robots * L) return;
  const int64_t rob = t / L;
  const int leg = int(t - rob * L);
  const int64_t slot = slot_of(rob, leg, L);
  auto f = [&](int field) -> double & { return st.legd[leg_field_index(field, slot, st.n_slots)]; };
  const double fn = norm(V3{f(FD::FORCE_IN), f(FD::FORCE_IN + 1), f(FD::FORCE_IN + 2)});
  if (fn > touchdown_threshold && f(FD::STEP_PLANE + 3) == 0.0) {
    double q[NJ];
    for (int j = 0; j < NJ; ++j) q[j] = f(FD::Q + j);
    Chain<NJ> ch;
    fk_chain<NJ>(gc->leg[leg], q, ch);
    const V3 tip = tip_robot_frame(gc->leg[leg], ch.pe); // step_plane_pose_ = current_tip_pose_
    f(FD::STEP_PLANE) = tip.x, f(FD::STEP_PLANE + 1) = tip.y, f(FD::STEP_PLANE + 2) = tip.z, f(FD::STEP_PLANE + 3) = 1.0;
  } else if (fn < liftoff_threshold) {
    f(FD::STEP_PLANE + 3) = 0.0;
  }
}

extern "C" int shc_engine_set_tip_force(shc_engine *e, const double *tip_force, int on_device) {
  SHC_BUSY_ONLY(e);
  if (e && !split_inputs(e, on_device && !e->params.rough_terrain_mode)) {
    const int rc_join_ = join_side(e);
    if (rc_join_ != SHC_OK) return rc_join_;
  }
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  int rc = scatter_leg(e, tip_force, 3, LEG_FIELD(e, FORCE_IN), on_device);
  if (rc != SHC_OK || !tip_force) return rc;
  e->rt_flags |= RT_TOUCHDOWN; // LegStepper::setTouchdownDetection(true) (state_controller.cpp:1642)
  if (e->params.rough_terrain_mode) { // the step plane is only read in rough terrain mode (walk_controller.cpp:1065, :1110)
    const int64_t threads = e->n * e->L;
#define CALL(L_, NJ_)                                                                                                           \
  touchdown_detection_kernel<L_, NJ_><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(                      \
      e->st, (const SharedConsts<L_, NJ_> *)e->d_consts, e->params.touchdown_threshold, e->params.liftoff_threshold)
    SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
    HIP_TRY(hipGetLastError());
  }
  return SHC_OK;
}

// First joint effort (or non-zero filter state): switch to the kernels that evaluate the tip-force estimate.
static int effort_live(shc_engine *e) {
  if (e->rt_flags & RT_EFFORT_LIVE) return SHC_OK;
  e->rt_flags |= RT_EFFORT_LIVE;
  build_cycle_params(e->params, e->tables, e->features, e->rt_flags, e->cp);
  return upload_consts(e);
}

extern "C" int shc_engine_set_joint_effort(shc_engine *e, const double *joint_effort, int on_device) {
  SHC_BUSY_ONLY(e);
  if (e && !split_inputs(e, on_device && (e->rt_flags & RT_EFFORT_LIVE))) {
    const int rc_join_ = join_side(e);
    if (rc_join_ != SHC_OK) return rc_join_;
  }
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (joint_effort) { // Leg::calculateTipForce has something to filter from now on
    const int rc = effort_live(e);
    if (rc != SHC_OK) return rc;
  }
  return scatter_leg(e, joint_effort, e->NJ, LEG_FIELD(e, EFFORT_IN), on_device);
}

extern "C" int shc_engine_set_pose_input(shc_engine *e, const double *translation_velocity, const double *rotation_velocity,
                                         int on_device) {
  SHC_BUSY_ONLY(e);
  if (e && !split_inputs(e, on_device)) {
    const int rc_join_ = join_side(e);
    if (rc_join_ != SHC_OK) return rc_join_;
  }
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (translation_velocity || rotation_velocity) e->rt_flags |= RT_MANUAL_LIVE;
  int rc = scatter_rob(e, translation_velocity, 3, RobotFields::TVI, on_device);
  if (rc != SHC_OK) return rc;
  return scatter_rob(e, rotation_velocity, 3, RobotFields::RVI, on_device);
}

extern "C" int shc_engine_set_pose_reset_mode(shc_engine *e, const int32_t *mode, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (!mode) return SHC_OK;
  e->rt_flags |= RT_MANUAL_LIVE;
  HIP_TRY(hipSetDevice(e->device));
  const int32_t *d = mode;
  if (!on_device) {
    HIP_TRY(hipMemcpyAsync(e->d_stage, mode, size_t(e->n) * 4, hipMemcpyHostToDevice, e->stream));
    d = reinterpret_cast<const int32_t *>(e->d_stage);
  }
  scatter_robi_kernel<<<dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, e->stream>>>(d, e->st.robi, 64 / e->L, e->n,
                                                                                       RobotFields::I_RESET_MODE);
  HIP_TRY(hipGetLastError());
  if (!on_device) HIP_TRY(hipStreamSynchronize(e->stream));
  return SHC_OK;
}

// The two streams the halves of split steps run on: ONE pair per device for the whole process, created back to back so that they
// sit on two different hardware queues whatever the caller's own stream is (HIP maps streams to a few hardware queues in
// creation order; two streams on one queue serialise, and a pair created per engine did land on the caller's queue now and then).
// ... "created back to back" is not enough: the runtime hands a new stream the least-used hardware queue of its priority class, and in
// a process that has created and destroyed other streams before (engines that came and went) both streams of the pair can be given the
// same one - the halves then run one after the other and a split step costs what a single launch costs.  The pair is therefore TESTED once: a one-wave kernel on the first
// stream waits (bounded) for a flag that a kernel launched afterwards on the second stream sets - it sees the flag only if the two run
// concurrently, i.e. sit on different queues.  A second stream that fails is kept aside (so that the next one is given another queue)
// and replaced, a few times at most.
__device__ unsigned long long g_queue_probe[2]; // [0] the flag, [1] what the waiting kernel saw
__global__ void queue_probe_reset_kernel() {
  if (threadIdx.x == 0) g_queue_probe[0] = g_queue_probe[1] = 0;
}
__global__ void queue_probe_wait_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  unsigned long long seen = 0;
  while ((seen = __hip_atomic_load(&g_queue_probe[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) g_queue_probe[1] = seen;
}
__global__ void queue_probe_set_kernel() {
  if (threadIdx.x == 0) __hip_atomic_store(&g_queue_probe[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (stream-ordered calls only: no allocation, no null-stream copy - another engine of this process may have a resident loop alive)
static bool streams_run_concurrently(int device, hipStream_t a, hipStream_t b) {
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;
  void *sym = nullptr;
  if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_queue_probe)) != hipSuccess) return true; // (cannot tell: keep the pair)
  unsigned long long h[2] = {0, 1};
  queue_probe_reset_kernel<<<dim3(1), dim3(64), 0, a>>>();
  if (hipStreamSynchronize(a) != hipSuccess) return true;
  queue_probe_wait_kernel<<<dim3(1), dim3(64), 0, a>>>(2ull * (unsigned long long)khz); // 2 ms at most
  queue_probe_set_kernel<<<dim3(1), dim3(64), 0, b>>>();
  bool ok = true;
  if (hipMemcpyAsync(h, sym, 16, hipMemcpyDeviceToHost, a) == hipSuccess && hipStreamSynchronize(a) == hipSuccess) ok = h[1] != 0;
  (void)hipStreamSynchronize(b);
  (void)hipGetLastError();
  return ok;
}
static int split_streams(int device, hipStream_t out[2]) {
  static hipStream_t pool[64][2] = {};
  static std::mutex pool_mutex; // engines of one process may be created and stepped from different host threads
  if (device < 0 || device >= 64) return fail(SHC_ERR_INVALID_ARG, "device index");
  std::lock_guard<std::mutex> lock(pool_mutex);
  if (!pool[device][0]) {
    // The pair is built in locals and published only once both streams exist: a failure on the way destroys what was created (streams set
    // aside included) and leaves the pool empty, so that a later call starts over instead of handing out a null second stream - which
    // would be the legacy default stream.
    hipStream_t a = nullptr, b = nullptr, aside[6] = {};
    int n_aside = 0;
    hipError_t err = hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    while (err == hipSuccess && n_aside < 6 && !streams_run_concurrently(device, a, b)) {
      aside[n_aside++] = b;
      b = nullptr;
      err = hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    }
    for (int k = 0; k < n_aside; ++k) (void)hipStreamDestroy(aside[k]);
    if (err != hipSuccess) {
      if (a) (void)hipStreamDestroy(a);
      if (b) (void)hipStreamDestroy(b);
      return fail(SHC_ERR_HIP, std::string("split streams: ") + hipGetErrorString(err));
    }
    if (getenv("SHC_DEBUG_STREAMS")) fprintf(stderr, "shc: split-stream pair of device %d found after %d replacement(s)\n", device, n_aside);
    pool[device][0] = a, pool[device][1] = b;
  }
  out[0] = pool[device][0], out[1] = pool[device][1];
  return SHC_OK;
}
// Order the engine's stream after everything earlier split steps launched on the split streams (no host wait).
static int join_side(shc_engine *e) {
  e->main_dirty = true;
  if (!e->side_busy) return SHC_OK;
  HIP_TRY(hipSetDevice(e->device));
  for (int h = 0; h < 2; ++h) {
    HIP_TRY(hipEventRecord(e->ev_half[h], e->half_stream[h]));
    HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_half[h], 0));
  }
  e->side_busy = false;
  return SHC_OK;
}
extern "C" int shc_engine_join(shc_engine *e) {
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  SHC_BUSY_ONLY(e);
  return join_side(e);
}

// A launch of W waves on S wave slots runs ceil(W / S) rounds; the last, partly filled round and the next launch's ramp-up leave
// most of the machine idle, and a kernel boundary is a barrier the algorithm does not need: step k + 1 of a robot depends on
// step k of THAT robot only.  From kSplitWaves waves on, a step is therefore launched as two halves of the batch on two streams
// with no join between steps, so that one half's tail overlaps the other half's full rounds (measured on 1 x MI355X: 65 536
// hexapods with admittance + IMU 57.7 -> 48.8 us per step, 131 072 octopods 106 -> 98.7 us).
constexpr int64_t kSplitWaves = 4096;

extern "C" int shc_engine_step(shc_engine *e, int n_cycles) {
  SHC_BUSY_ONLY(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (n_cycles < 1) return SHC_OK;
  HIP_TRY(hipSetDevice(e->device));
  if ((e->step_remap_pending || e->pose_params_held) && !(e->rt_flags & (RT_SKIP_MARKED | RT_POSE_MARKED))) {
    // The loop in which adjustParameter set a new value (state_controller.cpp:411-414, after the posing part of that loop, before updateWalk): its cycle runs
    // alone in its launch on a parameter block of its own - the posing part still on the old force gain / virtual spring / swing height (pose_params_held),
    // the phases of walking robots mapped onto a new step cycle between the posing part and updateWalk (step_remap_pending: cycle_front, on the
    // runtime-flag kernels) - and the plain block of the new values follows it.
    const bool remap = e->step_remap_pending;
    e->step_remap_pending = e->pose_params_held = false;
    const uint32_t keep = e->features;
    if (remap) e->features |= SHC_FEAT_GENERIC_KERNEL;
    int rc = shc_engine_step(e, 1);
    e->features = keep;
    if (rc == SHC_OK) rc = join_side(e);
    build_cycle_params(e->params, e->tables, e->features, e->rt_flags, e->cp);
    if (rc == SHC_OK) rc = upload_consts(e); // (ordered behind the cycle above on the engine's stream, and synchronises it)
    if (rc != SHC_OK || n_cycles == 1) return rc;
    --n_cycles;
  }
  if (!(e->rt_flags & (RT_SKIP_MARKED | RT_POSE_MARKED))) e->plan_poser_tips_current = false; // PoseController::updateStance rewrites every LegPoser's tip pose
  // One wave per workgroup while the batch has about as many waves as the chip has SIMDs (1 024): the dispatcher then spreads
  // them one per SIMD (two-wave groups put pairs on the same SIMDs: 12.9 instead of 9.4 us at 768 waves, 12.2 instead of 10.2 at
  // 1 024; equal at 1 536).  Above that, 128-thread groups
  // (two waves share one LDS copy of the tables): measured against 64 and 256 threads up to 131 072 instances they are the
  // fastest or within noise (-2 ... 3 % on 65 536 hexapods).
  const int block = e->n_waves < 1536 ? 64 : 128;
  const int64_t waves_per_block = block / 64;
  const bool split = e->n_waves >= kSplitWaves && !(e->features & SHC_FEAT_SINGLE_STREAM) && !(e->rt_flags & (RT_SKIP_MARKED | RT_POSE_MARKED));
  if (!split) {
    const int rc = join_side(e);
    if (rc != SHC_OK) return rc;
  }
  const int64_t half = split ? ((e->n_waves / 2 + waves_per_block - 1) / waves_per_block) * waves_per_block : e->n_waves;
  CycleLaunch a{e->st, e->d_consts, &e->cp, e->rt_flags, (e->features & SHC_FEAT_GENERIC_KERNEL) != 0, e->stream,
                (unsigned)((half + waves_per_block - 1) / waves_per_block), block, n_cycles, nullptr, nullptr, 0, e->half_steps};
#define CALL(L_, NJ_) shc_launch_cycle_##L_##_##NJ_(a)
  if (!split) {
    SHC_DISPATCH(e->L, e->NJ, CALL);
    HIP_TRY(hipGetLastError());
    return SHC_OK;
  }
  if (!e->half_stream[0]) {
    const int rc = split_streams(e->device, e->half_stream);
    if (rc != SHC_OK) return rc;
    HIP_TRY(hipEventCreateWithFlags(&e->ev_main, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_half[0], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_half[1], hipEventDisableTiming));
  }
  const bool resync = e->main_dirty; // inputs or state were touched on the engine's stream since the last split step: both halves follow them
  if (resync) {
    HIP_TRY(hipEventRecord(e->ev_main, e->stream));
    HIP_TRY(hipStreamWaitEvent(e->half_stream[0], e->ev_main, 0));
    HIP_TRY(hipStreamWaitEvent(e->half_stream[1], e->ev_main, 0));
    e->main_dirty = false;
  }
  a.stream = e->half_stream[0];
  if (resync) {
    // Two halves that start together end together - their tails and ramp-ups coincide and nothing overlaps (measured: a join
    // every tenth step costs the whole gain).  After a join the halves are therefore STAGGERED: the first half of this one step
    // goes out as two launches and the second half starts when the first of them is through, i.e. a quarter of a step late;
    // both halves take the same time from then on, so the stagger stays until the next join.
    const int64_t quarter = ((half / 2 + waves_per_block - 1) / waves_per_block) * waves_per_block; // (a quarter or three quarters of a half instead: the same step time within 1 %, profiles/r04_probe_build_variants.txt)
    a.grid = (unsigned)(quarter / waves_per_block);
    SHC_DISPATCH(e->L, e->NJ, CALL);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev_half[0], e->half_stream[0]));
    HIP_TRY(hipStreamWaitEvent(e->half_stream[1], e->ev_half[0], 0));
    a.wave0 = quarter;
    a.grid = (unsigned)((half - quarter + waves_per_block - 1) / waves_per_block);
  }
  SHC_DISPATCH(e->L, e->NJ, CALL);
  HIP_TRY(hipGetLastError());
  a.stream = e->half_stream[1];
  a.wave0 = half;
  a.grid = (unsigned)((e->n_waves - half + waves_per_block - 1) / waves_per_block);
  SHC_DISPATCH(e->L, e->NJ, CALL);
  HIP_TRY(hipGetLastError());
  e->side_busy = true;
#undef CALL
  return SHC_OK;
}

extern "C" int shc_engine_synchronize(shc_engine *e) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return SHC_OK;
}

#include "shc_resident.hpp" // resident mode: the control loop kept on the chip (shc_engine_resident_*)
static bool resident_active(const shc_engine *e) { return e->res && e->res->active; }
static void resident_shutdown(shc_engine *e) {
  if (e->res && e->res->active) (void)shc_engine_resident_end(e, nullptr);
  if (e->res && e->res->active) { // the loop did not answer: its buffers must outlive it - wait for the kernel itself (it is bounded: max_cycles,
    (void)hipSetDevice(e->device); //  idle timeout, the workers' emergency bound), however long that takes
    (void)hipStreamSynchronize(e->res->loop_stream);
    e->res->active = false;
  }
  resident_free(e->res);
  e->res = nullptr;
}

extern "C" int shc_engine_get_joint_state(shc_engine *e, double *q, double *qd, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  int rc = gather_leg(e, q, e->NJ, LEG_FIELD(e, Q), on_device);
  if (rc != SHC_OK) return rc;
  return gather_leg(e, qd, e->NJ, LEG_FIELD(e, QD), on_device);
}

extern "C" int shc_engine_joint_buffer(shc_engine *e, double **device_ptr, int64_t *n_doubles) {
  SHC_BUSY_GUARD(e);
  if (!e || !device_ptr || !n_doubles) return fail(SHC_ERR_INVALID_ARG, "NULL argument");
  *device_ptr = e->st.legd; // fields Q (and, for odd DOF, the first QD) occupy the first ceil(NJ / 2) paired planes of the leg state
  *n_doubles = int64_t((e->NJ + 1) / 2) * e->n_slots * 2;
  return SHC_OK;
}

extern "C" int64_t shc_engine_joint_index(const shc_engine *e, int64_t instance, int leg, int joint) {
  if (!e) return -1;
  int rpw = 64 / e->L;
  int64_t w = instance / rpw;
  int gi = int(instance - w * rpw);
  return leg_field_index(joint, w * 64 + gi * e->L + leg, e->n_slots);
}

extern "C" int shc_engine_get_leg_state(shc_engine *e, double *walker_tip, double *poser_tip, double *model_tip, double *tip_force,
                                        double *admittance, int32_t *leg_status, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  int rc;
  if ((rc = gather_leg(e, walker_tip, 3, LEG_FIELD(e, TIP), on_device)) != SHC_OK) return rc;
  if ((poser_tip || model_tip) && (rc = derive_tips(e)) != SHC_OK) return rc; // derived on demand from q / walker tip / body pose
  if ((rc = gather_leg(e, poser_tip, 3, LEG_FIELD(e, POSER_TIP), on_device)) != SHC_OK) return rc;
  if ((rc = gather_leg(e, model_tip, 3, LEG_FIELD(e, MODEL_TIP), on_device)) != SHC_OK) return rc;
  if ((rc = gather_leg(e, tip_force, 3, LEG_FIELD(e, TF), on_device)) != SHC_OK) return rc;
  if ((rc = gather_leg(e, admittance, 3, LEG_FIELD(e, ADM_DELTA), on_device)) != SHC_OK) return rc;
  if (leg_status) {
    HIP_TRY(hipSetDevice(e->device));
    int64_t threads = e->n * e->L;
    int32_t *d = on_device ? leg_status : reinterpret_cast<int32_t *>(e->d_stage);
    gather_leg_status_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(d, e->st.legi, e->n, e->L);
    HIP_TRY(hipGetLastError());
    if (!on_device) {
      HIP_TRY(hipMemcpyAsync(leg_status, d, size_t(threads) * 4, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipStreamSynchronize(e->stream));
    }
  }
  return SHC_OK;
}

extern "C" int shc_engine_change_gait(shc_engine *e, const shc_params *ng, int64_t *still_walking) {
  SHC_BUSY_GUARD(e);
  if (!e || !ng) return fail(SHC_ERR_INVALID_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  const unsigned grid = (unsigned)((e->n + 255) / 256);
  const int rpw = 64 / e->L;
  int32_t *d_count = reinterpret_cast<int32_t *>(e->d_stage);
  int32_t walking = 0;
  HIP_TRY(hipMemsetAsync(d_count, 0, 4, e->stream));
  count_walking_kernel<<<dim3(grid), dim3(256), 0, e->stream>>>(d_count, e->st.robi, rpw, e->n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(&walking, d_count, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (still_walking) *still_walking = walking;
  if (walking) { // state_controller.cpp:531-537
    fill_rob_kernel<<<dim3(grid), dim3(256), 0, e->stream>>>(e->st.robd, rpw, e->n, RobotFields::VIN, 3, 0.0);
    HIP_TRY(hipGetLastError());
    return SHC_OK;
  }
  shc_params p = e->params; // initGaitParameters (state_controller.cpp:1941-1969)
  p.stance_phase = ng->stance_phase;
  p.swing_phase = ng->swing_phase;
  p.phase_offset = ng->phase_offset;
  memcpy(p.offset_multiplier, ng->offset_multiplier, sizeof p.offset_multiplier);
  if (p.auto_posing) { // initAutoPoseParameters (:1973-1999)
    p.pose_frequency = ng->pose_frequency;
    p.pose_phase_length = ng->pose_phase_length;
    p.n_auto_posers = ng->n_auto_posers;
    memcpy(p.pose_phase_starts, ng->pose_phase_starts, sizeof p.pose_phase_starts);
    memcpy(p.pose_phase_ends, ng->pose_phase_ends, sizeof p.pose_phase_ends);
    memcpy(p.pose_negation_phase_starts, ng->pose_negation_phase_starts, sizeof p.pose_negation_phase_starts);
    memcpy(p.pose_negation_phase_ends, ng->pose_negation_phase_ends, sizeof p.pose_negation_phase_ends);
    memcpy(p.negation_transition_ratio, ng->negation_transition_ratio, sizeof p.negation_transition_ratio);
    memcpy(p.x_amplitudes, ng->x_amplitudes, sizeof p.x_amplitudes);
    memcpy(p.y_amplitudes, ng->y_amplitudes, sizeof p.y_amplitudes);
    memcpy(p.z_amplitudes, ng->z_amplitudes, sizeof p.z_amplitudes);
    memcpy(p.gravity_amplitudes, ng->gravity_amplitudes, sizeof p.gravity_amplitudes);
    memcpy(p.roll_amplitudes, ng->roll_amplitudes, sizeof p.roll_amplitudes);
    memcpy(p.pitch_amplitudes, ng->pitch_amplitudes, sizeof p.pitch_amplitudes);
    memcpy(p.yaw_amplitudes, ng->yaw_amplitudes, sizeof p.yaw_amplitudes);
  }
  int L, NJ;
  int rc = validate_params(&p, &L, &NJ);
  if (rc != SHC_OK) return rc;
  shc_tables t;
  if ((rc = shc_generate_tables(&p, &t)) != SHC_OK) return rc; // generateStepCycle + generateLimits (morphology tables come out unchanged)
  e->params = p;
  e->tables = t;
  e->span_dirty = true;
  e->step_remap_pending = e->pose_params_held = false; // (every robot is STOPPED: nothing is left of an adjustParameter in flight)
  build_cycle_params(e->params, e->tables, e->features, e->rt_flags, e->cp);
  if ((rc = upload_consts(e)) != SHC_OK) return rc;
  if (p.auto_posing) { // setAutoPoseParams builds fresh AutoPosers: their start / end checks are reset (pose_controller.cpp:39-61)
    fill_robi_kernel<<<dim3(grid), dim3(256), 0, e->stream>>>(e->st.robi, rpw, e->n, RobotFields::I_APOSER, 0);
    HIP_TRY(hipGetLastError());
  }
  return shc_engine_synchronize(e);
}

// An accepted step-frequency change that no cycle has consumed yet (the caller went on to something other than shc_engine_step): the phases are mapped
// here instead, by a kernel of their own - the same arithmetic as in cycle_front, without the reference's ordering against the posing part of that loop.
__global__ void step_remap_kernel(int32_t *legi, const int32_t *robi, int rpw, int64_t n, int L, int old_period, int period, int swing_start, int swing_end,
                                  int stance_end, int stance_start) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n * L) return;
  const int64_t rob = t / L;
  const int leg = int(t - rob * L);
  if ((robi[rob_index(rob, RobotFields::I_WORD, rpw, RobotFields::I_COUNT)] & 3) != WS_MOVING) return;
  int w = legi[slot_of(rob, leg, L)];
  int ph = (w >> LW_PHASE_SHIFT) & LW_PHASE_MASK, st = w & 3;
  const double step_progress = double(ph) / double(old_period);
  ph = int(step_progress * double(period));
  if (st != SS_FORCE_STOP) {
    if (ph >= swing_start && ph < swing_end && st != SS_FORCE_STANCE) st = SS_SWING;
    else if (ph < stance_end || ph >= stance_start) st = SS_STANCE;
  }
  legi[slot_of(rob, leg, L)] = (w & ~(3 | (LW_PHASE_MASK << LW_PHASE_SHIFT))) | st | (ph << LW_PHASE_SHIFT);
}
static int flush_step_remap(shc_engine *e) {
  if (!e->step_remap_pending) {
    if (!e->pose_params_held) return SHC_OK;
    e->pose_params_held = false; // (the next cycle is not a shc_engine_step: the new values are in force for all of it)
    build_cycle_params(e->params, e->tables, e->features, e->rt_flags, e->cp);
    return upload_consts(e);
  }
  e->step_remap_pending = e->pose_params_held = false;
  HIP_TRY(hipSetDevice(e->device));
  {
    const int rc = join_side(e);
    if (rc != SHC_OK) return rc;
  }
  const int64_t threads = e->n * e->L;
  const shc_step_cycle &s = e->tables.step;
  step_remap_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(e->st.legi, e->st.robi, 64 / e->L, e->n, e->L, e->cp.remap_old_period, s.period,
                                                                                       s.swing_start, s.swing_end, s.stance_end, s.stance_start);
  HIP_TRY(hipGetLastError());
  build_cycle_params(e->params, e->tables, e->features, e->rt_flags, e->cp);
  return upload_consts(e);
}

// WalkController::getLimit (walk_controller.cpp:414-436) on the host, for the acceptance test of a step-frequency change: per leg the bearing of its
// stride velocity (linear + angular x the tip's lever arm), rounded to whole degrees, picks the two neighbouring entries of the 45-degree table; the
// interpolation input is an int / int division in the reference (0 except on a table bearing), the smallest value over the legs is the limit.
static double host_get_limit(const double *tips /* [L][3] walker tip positions */, int L, double lx, double ly, double ang, const double *limit /* [9] */) {
  double lowest = kUnassigned;
  for (int l = 0; l < L; ++l) {
    const double sx = lx + ang * -tips[l * 3 + 1], sy = ly + ang * tips[l * 3 + 0];
    int bearing = mod_i(round_to_int(atan2(sy, sx) * (180.0 / M_PI)), 360);
    int upper = ((bearing + 44) / 45) * 45;
    const int lower = mod_i(upper - 45, 360);
    if (bearing < lower) bearing += 360;
    if (upper < lower) upper += 360;
    const double c = double((bearing - lower) / (upper - lower));
    const double lo = limit[lower / 45], hi = limit[mod_i(upper, 360) / 45];
    lowest = fmin(lowest, lo * (1.0 - c) + hi * c);
  }
  return lowest;
}

static int adjust_step_frequency(shc_engine *e, double value, int64_t *pending) {
  shc_params &p = e->params;
  // What the posing part of the accepting loop reads from the legs' steppers must not depend on the step cycle's constants (it runs on the old cycle in the
  // reference, on the new constants here): the walk-plane blend of rough terrain mode and the tip rotations / tip-align pose of gravity_aligned_tips do.
  // WalkController::generateLimits takes its stance radius from leg 0's CURRENT default tip (walk_controller.cpp:322-326), which a stance span modifier moves.
  if (p.rough_terrain_mode || p.gravity_aligned_tips || p.stance_span_modifier != 0.0)
    return fail(SHC_ERR_UNSUPPORTED, "step_frequency cannot be adjusted at run time in rough_terrain_mode, with gravity_aligned_tips or with a stance span modifier "
                                     "(the other eight parameters can); change it between runs, shc_engine_create");
  if (!(value > 0.0)) return fail(SHC_ERR_INVALID_ARG, "step_frequency must be positive");
  int rc = flush_step_remap(e); // (two changes without a cycle between them)
  if (rc != SHC_OK) return rc;
  shc_params np = p;
  np.step_frequency = value;
  const shc_step_cycle ns = hostinit::generate_step_cycle(np);
  if (ns.period <= 0 || ns.period > LW_PHASE_MASK) return fail(SHC_ERR_INVALID_ARG, "step_frequency gives a degenerate step cycle");
  p.step_frequency = value; // p->current_value = new_parameter_value_ (:454): the sequence / transition timings read it from now on, accepted or not
  shc_tables tn = e->tables;
  tn.step = ns;
  hostinit::generate_limits(p, tn); // the four limit maps + the legs' phase offsets of the new cycle
  // :462-463 and generateLimits' setPhaseOffset (walk_controller.cpp:277): the speed maps and the phase offsets are the new cycle's from here on, whether the
  // change is accepted in this loop or not - the walker slows down to them (updateWalk :456-482), which is what makes a later call succeed
  for (int b = 0; b < SHC_N_BEARINGS; ++b) {
    e->tables.max_linear_speed[b] = tn.max_linear_speed[b];
    e->tables.max_angular_speed[b] = tn.max_angular_speed[b];
  }
  for (int l = 0; l < e->L; ++l) e->tables.phase_offset[l] = tn.phase_offset[l];
  // the test of :464-489, for every instance: desired body velocity inside what its velocity input maps to under the new limits
  std::vector<double> vin(size_t(e->n) * 3), vel(size_t(e->n) * 3), tips(size_t(e->n) * e->L * 3);
  if ((rc = gather_rob(e, vin.data(), 3, RobotFields::VIN, 0)) != SHC_OK) return rc;
  if ((rc = gather_rob(e, vel.data(), 3, RobotFields::VLIN, 0)) != SHC_OK) return rc;
  if ((rc = gather_leg(e, tips.data(), 3, LEG_FIELD(e, TIP), 0)) != SHC_OK) return rc;
  int64_t waiting = 0;
  for (int64_t i = 0; i < e->n; ++i) {
    const double *in = &vin[size_t(i) * 3], *v = &vel[size_t(i) * 3], *tp = &tips[size_t(i) * e->L * 3];
    const double max_lin = host_get_limit(tp, e->L, in[0], in[1], in[2], tn.max_linear_speed);
    const double max_ang = host_get_limit(tp, e->L, in[0], in[1], in[2], tn.max_angular_speed);
    double tx, ty, ta;
    if (p.velocity_input_mode == SHC_VEL_THROTTLE) {
      const double nrm = sqrt(in[0] * in[0] + in[1] * in[1]);
      const double k = nrm > 1.0 ? 1.0 / nrm : 1.0; // clamped(vector, 1.0)
      tx = in[0] * k * max_lin;
      ty = in[1] * k * max_lin;
      ta = clampd(in[2], -1.0, 1.0) * max_ang;
      tx *= 1.0 - fabs(in[2]);
      ty *= 1.0 - fabs(in[2]);
    } else {
      const double nrm = sqrt(in[0] * in[0] + in[1] * in[1]);
      const double k = nrm > max_lin ? max_lin / nrm : 1.0;
      tx = in[0] * k;
      ty = in[1] * k;
      ta = clampd(in[2], -max_ang, max_ang);
    }
    if (!(v[0] <= tx && v[1] <= ty && fabs(v[2]) <= fabs(ta))) ++waiting; // (signed comparisons of the linear components: as the reference has them)
  }
  if (pending) *pending = waiting;
  if (waiting) return upload_consts(e); // not yet: the new speed maps / phase offsets are in force, the step cycle and the acceleration maps are the old ones
  // accepted: walker_->generateStepCycle() + generateLimits() (:491-492).  setAutoPoseParams is NOT called (only init / changeGait do): the auto-pose phase
  // tables keep counting in the old step period, as in the reference.
  const int old_period = e->tables.step.period;
  e->tables.step = ns;
  for (int b = 0; b < SHC_N_BEARINGS; ++b) {
    e->tables.max_linear_acceleration[b] = tn.max_linear_acceleration[b];
    e->tables.max_angular_acceleration[b] = tn.max_angular_acceleration[b];
  }
  {
    const CycleParams before = e->cp;
    build_cycle_params(e->params, e->tables, e->features, e->rt_flags, e->cp);
    if (e->pose_params_held) { // (a direct parameter was adjusted since the last cycle: its old value still belongs to the next cycle's posing part)
      e->cp.adm_m00 = before.adm_m00, e->cp.adm_m01 = before.adm_m01, e->cp.adm_m10 = before.adm_m10, e->cp.adm_m11 = before.adm_m11;
      e->cp.adm_g0 = before.adm_g0, e->cp.adm_g1 = before.adm_g1;
      e->cp.virtual_stiffness = before.virtual_stiffness, e->cp.pose_force_gain = before.pose_force_gain, e->cp.pose_swing_height = before.pose_swing_height;
    }
  }
  e->cp.remap_old_period = old_period; // generateStepCycle's updatePhase for MOVING robots: inside the next cycle (shc_engine_step)
  e->step_remap_pending = true;
  return upload_consts(e);
}

extern "C" int shc_engine_adjust_parameter(shc_engine *e, int which, double value, int64_t *pending) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (pending) *pending = 0;
  if (!(value == value) || fabs(value) > 1e300) return fail(SHC_ERR_INVALID_ARG, "parameter value is not finite");
  HIP_TRY(hipSetDevice(e->device));
  shc_params &p = e->params;
  switch (which) {
    case SHC_PARAM_STEP_FREQUENCY: return adjust_step_frequency(e, value, pending);
    case SHC_PARAM_SWING_HEIGHT: p.swing_height = value; break;          // LegStepper::updateStride's swing clearance, the dynamic-stiffness reference, sequence step heights
    case SHC_PARAM_SWING_WIDTH: p.swing_width = value; break;            // generateSecondarySwingControlNodes' lateral shift (walk_controller.cpp:1243)
    case SHC_PARAM_STEP_DEPTH: p.step_depth = value; break;              // the proactive step-plane target (:1099)
    case SHC_PARAM_STANCE_SPAN_MODIFIER: p.stance_span_modifier = value; e->span_dirty = true; break; // calculateStanceSpanChange (:966), applied at the next stop / swing start
    case SHC_PARAM_VIRTUAL_MASS:
      if (!(value > 0.0)) return fail(SHC_ERR_INVALID_ARG, "virtual_mass must be positive");
      p.virtual_mass = value;
      break;
    case SHC_PARAM_VIRTUAL_STIFFNESS:
      if (!(value > 0.0)) return fail(SHC_ERR_INVALID_ARG, "virtual_stiffness must be positive");
      p.virtual_stiffness = value;
      break;
    case SHC_PARAM_VIRTUAL_DAMPING: p.virtual_damping_ratio = value; break;
    case SHC_PARAM_FORCE_GAIN: p.force_gain = value; break;              // admittance input (admittance_controller.cpp:32), tip-force estimate (model.cpp:705), LegState tip force
    default: return fail(SHC_ERR_INVALID_ARG, "unknown adjustable parameter (SHC_PARAM_*)");
  }
  // The eight parameters the control cycle reads as they are (params_.*.current_value): a new launch-uniform block, in force from the next cycle; no table is
  // regenerated and no state is touched.
  // ... except where the POSING part of the loop reads them (updateStiffness / updateAdmittance run before runningState, state_controller.cpp:170-180): the
  // virtual spring's constants, the force gain as the admittance input scales it and the swing height as the dynamic-stiffness reference divides by it stay
  // what they were for the posing part of the next cycle (the tip-force estimate and the stepper of that same cycle use the new values), then follow.
  const CycleParams before = e->cp;
  build_cycle_params(e->params, e->tables, e->features, e->rt_flags, e->cp);
  e->cp.remap_old_period = before.remap_old_period;
  e->cp.adm_m00 = before.adm_m00, e->cp.adm_m01 = before.adm_m01, e->cp.adm_m10 = before.adm_m10, e->cp.adm_m11 = before.adm_m11;
  e->cp.adm_g0 = before.adm_g0, e->cp.adm_g1 = before.adm_g1;
  e->cp.virtual_stiffness = before.virtual_stiffness;
  e->cp.pose_force_gain = before.pose_force_gain;
  e->cp.pose_swing_height = before.pose_swing_height;
  if (which == SHC_PARAM_SWING_HEIGHT || which == SHC_PARAM_VIRTUAL_MASS || which == SHC_PARAM_VIRTUAL_STIFFNESS || which == SHC_PARAM_VIRTUAL_DAMPING ||
      which == SHC_PARAM_FORCE_GAIN)
    e->pose_params_held = true;
  return upload_consts(e);
}

extern "C" int shc_engine_get_odometry(shc_engine *e, double *pose, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (!e->cp.odometry) return fail(SHC_ERR_UNSUPPORTED, "SHC_FEAT_ODOMETRY is off");
  if (!pose) return SHC_OK;
  HIP_TRY(hipSetDevice(e->device));
  double *d = on_device ? pose : e->d_stage;
  gather_odometry_kernel<<<dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, e->stream>>>(d, e->st.robd, 64 / e->L, e->n);
  HIP_TRY(hipGetLastError());
  if (!on_device) {
    HIP_TRY(hipMemcpyAsync(pose, d, size_t(e->n) * 7 * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return SHC_OK;
}

extern "C" int shc_engine_get_virtual_stiffness(shc_engine *e, double *stiffness, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (!e->params.admittance_control) return fail(SHC_ERR_UNSUPPORTED, "admittance_control is off: updateStiffness never runs");
  return gather_leg(e, stiffness, 1, LEG_FIELD(e, ADM_DELTA) + 3, on_device);
}

static int derive_tips(shc_engine *e) {
  HIP_TRY(hipSetDevice(e->device));
  const int64_t threads = e->n * e->L;
  const int derive_poser = !(e->cp.auto_posing && !e->cp.imu_posing); // the auto-pose path stores its per-leg poser tip
  const int keep_marked = e->plan_poser_tips_current && (e->params.imu_posing || e->params.auto_posing || e->params.inclination_posing);
#define CALL(L_, NJ_)                                                                                             \
  derive_tips_kernel<L_, NJ_><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(                \
      e->st, (const SharedConsts<L_, NJ_> *)e->d_consts, derive_poser, keep_marked)
  SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
  HIP_TRY(hipGetLastError());
  return SHC_OK;
}

// PoseController::auto_pose_ of the cycle whose master phase and (post-update) poser latches are given: the sum the cycle
// kernel forms (AutoPoser::updatePose, pose_controller.cpp:1338-1439), re-derived on the host for LegState.auto_pose.
static Pose host_auto_pose(const shc_params &p, const shc_tables &t, int master_phase, int flags, Quat imu) {
  Pose auto_pose = pose_identity();
  const int len = t.pose_phase_length, nrm = t.pose_normaliser;
  for (int i = 0; i < p.n_auto_posers && i < kMaxAutoPosers; ++i) {
    const bool allow = ((flags >> (4 * i)) & 8) != 0;
    int phase = master_phase, sp = p.pose_phase_starts[i] * nrm, ep = p.pose_phase_ends[i] * nrm;
    if (sp > ep) {
      ep += len;
      if (phase < sp) phase += len;
    }
    if (!(phase >= sp && phase < ep && allow)) continue;
    const int iteration = phase - sp + 1, num = ep - sp;
    const bool first_half = iteration <= num / 2;
    const double delta_t = 1.0 / (num / 2.0);
    const int offset = int(first_half ? 0 : num / 2.0);
    const double tt = (iteration - offset) * delta_t, u = 1.0 - tt;
    const double wgt = first_half ? (4.0 * tt * tt * tt * u + tt * tt * tt * tt) : (u * u * u * u + 4.0 * tt * u * u * u);
    V3 pos;
    if (p.gravity_amplitudes[i] != 0.0) { // Model::estimateGravity (model.cpp:156-165)
      const V3 e = quat_to_euler(imu, false);
      V3 gv{0, 0, kGravity};
      gv = rotate(angle_axis_y(-e.y), gv);
      gv = rotate(angle_axis_x(-e.x), gv);
      pos = normalized(gv) * (p.gravity_amplitudes[i] * wgt);
    } else {
      pos = V3{p.x_amplitudes[i] * wgt, p.y_amplitudes[i] * wgt, p.z_amplitudes[i] * wgt};
    }
    const V3 rot{p.roll_amplitudes[i] * wgt, p.pitch_amplitudes[i] * wgt, p.yaw_amplitudes[i] * wgt};
    auto_pose = add_pose(auto_pose, Pose{pos, euler_to_quat(rot, false)});
  }
  return auto_pose;
}
// LegPoser::updateAutoPose's negation (pose_controller.cpp:1740-1776) for a leg whose negate flag is set
static Pose host_leg_auto_pose(const shc_params &p, const shc_tables &t, int leg, int master_phase, bool negate, const Pose &auto_pose) {
  if (!negate) return auto_pose;
  const int len = t.pose_phase_length, nrm = t.pose_normaliser;
  int sp = p.pose_negation_phase_starts[leg] * nrm, ep = p.pose_negation_phase_ends[leg] * nrm, np = master_phase;
  if (sp == 0) sp = len;
  if (ep == 0) ep = len;
  if (sp > ep) {
    ep += len;
    if (np < sp) np += len;
  }
  const int iteration = np - sp + 1, num = ep - sp;
  const bool first_half = iteration <= num / 2;
  double ci = 1.0;
  const double ratio = p.negation_transition_ratio[leg];
  if (ratio > 0.0) ci = first_half ? fmin(1.0, iteration / (num * ratio)) : fmin(1.0, (num - iteration) / (num * ratio));
  ci = smooth_step(ci);
  return remove_pose(auto_pose, interpolate_pose(pose_identity(), ci, auto_pose));
}

template <int NJ>
static Pose host_fk_tip_pose(const shc_params &p, int leg, const double *q) {
  LegConst<NJ> lc;
  hostinit::fill_leg_const<NJ>(p, leg, lc);
  double qq[NJ];
  for (int j = 0; j < NJ; ++j) qq[j] = q[j];
  return fk_tip_pose<NJ>(lc, qq);
}

extern "C" int shc_engine_read_leg_state_msg(shc_engine *e, int64_t instance, shc_leg_state_msg *legs) {
  SHC_BUSY_GUARD(e);
  if (!e || !legs) return fail(SHC_ERR_INVALID_ARG, "NULL argument");
  if (instance < 0 || instance >= e->n) return fail(SHC_ERR_INVALID_ARG, "instance out of range");
  int rc = derive_tips(e);
  if (rc != SHC_OK) return rc;
  const int L = e->L, NJ = e->NJ, per_leg = 12 + 4 * NJ + 8;
  std::vector<double> h(size_t(L) * per_leg + 10);
  switch (NJ) {
    case 3: read_instance_kernel<3><<<dim3(1), dim3(64), 0, e->stream>>>(e->d_stage, e->st, L, instance, e->params.admittance_control); break;
    case 4: read_instance_kernel<4><<<dim3(1), dim3(64), 0, e->stream>>>(e->d_stage, e->st, L, instance, e->params.admittance_control); break;
    default: read_instance_kernel<5><<<dim3(1), dim3(64), 0, e->stream>>>(e->d_stage, e->st, L, instance, e->params.admittance_control); break;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h.data(), e->d_stage, h.size() * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const shc_step_cycle &step = e->tables.step;
  const double *rb = &h[size_t(L) * per_leg];
  const double vx = rb[0], vy = rb[1], vw = rb[2];
  const double swing_time = (double(step.swing_period) / step.period) / step.frequency;   // state_controller.cpp:863
  const double stance_time = (double(step.stance_period) / step.period) / step.frequency; // :864
  // PoseController::auto_pose_ of the last cycle (auto posing runs only where IMU posing does not, pose_controller.cpp:836-846)
  const bool auto_live = e->params.auto_posing && !e->params.imu_posing && e->tables.pose_phase_length > 0;
  int master_phase = int(rb[5]);
  if (auto_live && e->params.pose_frequency != -1.0) master_phase = mod_i(master_phase - 1, e->tables.pose_phase_length); // the counter has advanced
  const Pose auto_pose = auto_live ? host_auto_pose(e->params, e->tables, master_phase, int(rb[4]), Quat{rb[6], rb[7], rb[8], rb[9]}) : pose_identity();
  for (int l = 0; l < L; ++l) {
    const double *o = &h[size_t(l) * per_leg];
    shc_leg_state_msg &m = legs[l];
    memset(&m, 0, sizeof m);
    for (int k = 0; k < 3; ++k) {
      m.walker_tip_position[k] = o[k];
      m.target_tip_position[k] = o[3 + k];
      m.poser_tip_position[k] = o[6 + k];
      m.model_tip_position[k] = o[9 + k];
      m.tip_force[k] = o[12 + 2 * NJ + k] * e->params.force_gain;
      m.admittance_delta[k] = o[15 + 2 * NJ + k];
    }
    for (int j = 0; j < NJ; ++j) {
      m.joint_positions[j] = o[12 + j];
      m.joint_velocities[j] = o[12 + NJ + j];
      m.joint_efforts[j] = o[20 + 2 * NJ + j]; // desired_effort_ = current_effort_ (state_controller.cpp:1590)
    }
    { // actual_tip_pose: Leg::applyFK(false, true) on the measured joint positions (:839)
      const double *qm = &o[20 + 3 * NJ];
      const Pose tp = NJ == 3 ? host_fk_tip_pose<3>(e->params, l, qm) : (NJ == 4 ? host_fk_tip_pose<4>(e->params, l, qm) : host_fk_tip_pose<5>(e->params, l, qm));
      m.actual_tip_pose[0] = tp.p.x, m.actual_tip_pose[1] = tp.p.y, m.actual_tip_pose[2] = tp.p.z;
      m.actual_tip_pose[3] = tp.r.w, m.actual_tip_pose[4] = tp.r.x, m.actual_tip_pose[5] = tp.r.y, m.actual_tip_pose[6] = tp.r.z;
    }
    m.virtual_stiffness = o[18 + 2 * NJ];
    // LegStepper::swing_progress_ / stance_progress_ as iteratePhase left them (walk_controller.cpp:871-897)
    const int word = int(o[19 + 2 * NJ]);
    const int pm = (word >> LW_PM_SHIFT) & 3, phase = (word >> LW_PHASE_SHIFT) & LW_PHASE_MASK;
    m.swing_progress = m.stance_progress = -1.0; // walk_controller.h:498-499
    if (pm == PM_SWING) {
      m.swing_progress = clampd(double(phase - step.swing_start + 1) / double(step.swing_end - step.swing_start), 0.0, 1.0);
    } else if (pm == PM_STANCE) {
      m.stance_progress = clampd(double(mod_i(phase + (step.period - step.stance_start), step.period) + 1) /
                                     double(mod_i(step.stance_end - step.stance_start, step.period)),
                                 0.0, 1.0);
    } else if (pm == PM_STOP) {
      m.stance_progress = 0.0;
    }
    m.time_to_swing_end = m.stance_progress >= 0.0 ? stance_time * (1.0 - m.stance_progress) + swing_time
                                                   : swing_time * (1.0 - m.swing_progress); // :866-873
    const double t = m.time_to_swing_end; // WalkController::calculateOdometry (walk_controller.cpp:783-791)
    m.pose_delta[0] = vx * t;
    m.pose_delta[1] = vy * t;
    m.pose_delta[2] = 0.0 * t;
    m.pose_delta[3] = cos(0.5 * (vw * t));
    m.pose_delta[6] = sin(0.5 * (vw * t));
    // (model_tip_velocity stays 0: see the header.)  LegPoser::auto_pose_ :877-880
    const Pose la = auto_live ? host_leg_auto_pose(e->params, e->tables, l, master_phase, (word & LW_NEG) != 0, auto_pose) : pose_identity();
    m.auto_pose[0] = la.p.x, m.auto_pose[1] = la.p.y, m.auto_pose[2] = la.p.z;
    m.auto_pose[3] = la.r.w, m.auto_pose[4] = la.r.x, m.auto_pose[5] = la.r.y, m.auto_pose[6] = la.r.z;
  }
  return SHC_OK;
}

// ---- the other messages of the path (include/shc_batch.h)
// raw motor positions - Joint::offset_ -> measured joint positions (jointStatesCallback, state_controller.cpp:1581)
__global__ void store_measured_q_kernel(const double *src, double *legd, int64_t n_slots, int64_t n, int L, int NJ, int f0, const double *offset /*[L][NJ]*/) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n * L) return;
  const int64_t rob = t / L;
  const int leg = int(t - rob * L);
  const int64_t slot = slot_of(rob, leg, L);
  for (int j = 0; j < NJ; ++j) legd[leg_field_index(f0 + j, slot, n_slots)] = src[t * NJ + j] - offset[leg * NJ + j];
}
// desired joint state + per-joint position commands (publishDesiredJointState, state_controller.cpp:777-805)
__global__ void joint_commands_kernel(double *pos, double *vel, double *eff, double *cmd, const double *legd, int64_t n_slots, int64_t n, int L, int NJ,
                                      int fq, int fqd, int feff, const double *offset) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n * L) return;
  const int64_t rob = t / L;
  const int leg = int(t - rob * L);
  const int64_t slot = slot_of(rob, leg, L);
  for (int j = 0; j < NJ; ++j) {
    const double q = legd[leg_field_index(fq + j, slot, n_slots)];
    if (pos) pos[t * NJ + j] = q;
    if (vel) vel[t * NJ + j] = legd[leg_field_index(fqd + j, slot, n_slots)];
    if (eff) eff[t * NJ + j] = legd[leg_field_index(feff + j, slot, n_slots)];
    if (cmd) cmd[t * NJ + j] = q + offset[leg * NJ + j];
  }
}
// range-sensor step plane (tipStatesCallback, state_controller.cpp:1657-1672): the surface lies `z` along the tip's x axis
template <int L, int NJ>
__global__ void step_plane_range_kernel(DevState st, const SharedConsts<L, NJ> *gc, const double *step_plane /*[n][L][3]*/) {
  using FD = Fields<NJ>;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= st.n_robots * L) return;
  const int64_t rob = t / L;
  const int leg = int(t - rob * L);
  const int64_t slot = slot_of(rob, leg, L);
  auto f = [&](int field) -> double & { return st.legd[leg_field_index(field, slot, st.n_slots)]; };
  const double z = step_plane[t * 3 + 2];
  if (z != kUnassigned) {
    double q[NJ];
    for (int j = 0; j < NJ; ++j) q[j] = f(FD::Q + j);
    Chain<NJ> ch;
    fk_chain<NJ>(gc->leg[leg], q, ch);
    const V3 p = tip_robot_frame(gc->leg[leg], ch.pe) + base_rotate(gc->leg[leg], ch.xe) * z; // Tip::getPoseRobotFrame(Pose((z, 0, 0), ..))
    f(FD::STEP_PLANE) = p.x, f(FD::STEP_PLANE + 1) = p.y, f(FD::STEP_PLANE + 2) = p.z, f(FD::STEP_PLANE + 3) = 1.0;
  } else {
    f(FD::STEP_PLANE + 3) = 0.0; // lost contact with the range sensor: Pose::Undefined()
  }
}

static int upload_offsets(shc_engine *e, const double **d_off) {
  std::vector<double> off(size_t(e->L) * e->NJ);
  for (int l = 0; l < e->L; ++l)
    for (int j = 0; j < e->NJ; ++j) off[size_t(l) * e->NJ + j] = e->params.joint[l][j].offset;
  // the tail of the staging buffer (the scatter / gather helpers use its head)
  double *dst = e->d_stage + (e->stage_bytes / 8 - off.size());
  HIP_TRY(hipMemcpyAsync(dst, off.data(), off.size() * 8, hipMemcpyHostToDevice, e->stream));
  *d_off = dst;
  return SHC_OK;
}

extern "C" int shc_engine_set_joint_states_msg(shc_engine *e, const double *position, const double * /*velocity*/, const double *effort, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  HIP_TRY(hipSetDevice(e->device));
  int rc = SHC_OK;
  if (position) {
    const double *d_off, *d;
    if ((rc = upload_offsets(e, &d_off)) != SHC_OK) return rc;
    if ((rc = to_device(e, position, size_t(e->n) * e->L * e->NJ, on_device, &d)) != SHC_OK) return rc;
    const int64_t threads = e->n * e->L;
    store_measured_q_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(d, e->st.legd, e->n_slots, e->n, e->L, e->NJ,
                                                                                              LEG_FIELD(e, MEAS_Q), d_off);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  if (effort) rc = shc_engine_set_joint_effort(e, effort, on_device);
  return rc;
}

extern "C" int shc_engine_set_tip_states_msg(shc_engine *e, const double *wrench_force, const double *step_plane, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  int rc = SHC_OK;
  if (wrench_force && (rc = shc_engine_set_tip_force(e, wrench_force, on_device)) != SHC_OK) return rc;
  if (step_plane) {
    HIP_TRY(hipSetDevice(e->device));
    e->rt_flags |= RT_TOUCHDOWN; // setTouchdownDetection(true) (:1655)
    const double *d;
    if ((rc = to_device(e, step_plane, size_t(e->n) * e->L * 3, on_device, &d)) != SHC_OK) return rc;
    const int64_t threads = e->n * e->L;
#define CALL(L_, NJ_) \
  step_plane_range_kernel<L_, NJ_><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(e->st, (const SharedConsts<L_, NJ_> *)e->d_consts, d)
    SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
    HIP_TRY(hipGetLastError());
    if (!on_device) HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return rc;
}

// ---- externally requested tip targets / default poses (rough terrain mode; struct ExternalTarget, walk_controller.h:38-46)
struct ExtRow { // shc_external_target as doubles (staged on the device)
  double pose[7], transform[7], swing_clearance, flags /* bit 0 defined, bit 1 odom_ideal */;
};
struct ExtAt { int base, flags; };
__host__ __device__ inline ExtAt ext_record(int which) { // 0 LegStepper::external_target_, 1 external_default_, 2 LegPoser::external_target_
  return which == 0 ? ExtAt{ExtFields::T_POSE, ExtFields::T_FLAGS} : which == 1 ? ExtAt{ExtFields::D_POSE, ExtFields::D_FLAGS} : ExtAt{ExtFields::P_POSE, ExtFields::P_FLAGS};
}
__global__ void set_external_kernel(DevState st, int L, int64_t first, int64_t count, int leg_sel, int which, const ExtRow *rows, int transform_only,
                                    unsigned long long *ignored, SeqRobotState *seq, int rough_terrain) {
  const int legs = leg_sel < 0 ? L : 1;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= count * legs) return;
  const int64_t rob = first + t / legs;
  const int leg = leg_sel < 0 ? int(t % legs) : leg_sel;
  const int64_t slot = slot_of(rob, leg, L);
  int base = ext_record(which).base, flags_at = ext_record(which).flags;
  auto f = [&](int field) -> double & { return st.ext[leg_field_index(field, slot, st.n_slots)]; };
  const ExtRow &r = rows[t];
  if (transform_only) { // generateExternalTargetTransforms (state_controller.cpp:703-773): only defined requests are refreshed
    if ((int(f(flags_at)) & 1) != 0)
      for (int k = 0; k < 7; ++k) f(base + 7 + k) = r.transform[k];
    return;
  }
  const int defined = int(r.flags) & 1;
  if (!defined) { // withdrawn: defined_ = false, the rest of the record stays
    f(flags_at) = double(int(f(flags_at)) & ~1);
    return;
  }
  // targetTipPoseCallback (:1734-1757): the LegStepper takes a request only while its robot is not STOPPED; the target of a robot
  // that stands goes to its LegPoser for planner mode (target_tip_pose_acquired_, :1738-1742), a default for it is dropped
  const int walk_state = st.robi[rob_index(rob, RobotFields::I_WORD, 64 / L, RobotFields::I_COUNT)] & 3;
  if (which == 2 || (which == 0 && walk_state == WS_STOPPED)) {
    which = 2;
    base = ext_record(2).base, flags_at = ext_record(2).flags;
    seq[rob].tip_pose_acquired = 1;
  } else if (walk_state == WS_STOPPED || !rough_terrain) { // (without rough_terrain_mode no stepper ever reads its requests)
    atomicAdd(ignored, 1ull);
    return;
  }
  for (int k = 0; k < 7; ++k) f(base + k) = r.pose[k], f(base + 7 + k) = r.transform[k];
  if (which == 0) f(ExtFields::T_CLEARANCE) = r.swing_clearance;
  if (which == 2) f(ExtFields::P_CLEARANCE) = r.swing_clearance;
  f(flags_at) = r.flags;
}
__global__ void get_external_kernel(DevState st, int L, int64_t first, int64_t count, int leg_sel, int which, ExtRow *rows) {
  const int legs = leg_sel < 0 ? L : 1;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= count * legs) return;
  const int64_t slot = slot_of(first + t / legs, leg_sel < 0 ? int(t % legs) : leg_sel, L);
  const int base = ext_record(which).base;
  auto f = [&](int field) { return st.ext[leg_field_index(field, slot, st.n_slots)]; };
  ExtRow &r = rows[t];
  for (int k = 0; k < 7; ++k) r.pose[k] = f(base + k), r.transform[k] = f(base + 7 + k);
  r.swing_clearance = which == 0 ? f(ExtFields::T_CLEARANCE) : which == 2 ? f(ExtFields::P_CLEARANCE) : 0.0;
  r.flags = f(ext_record(which).flags);
}

static int ensure_seq(shc_engine *e);
static int ensure_manual(shc_engine *e, bool planner);
static int ensure_manual_records(shc_engine *e);
static int external_select(shc_engine *e, int which, int64_t first, int64_t count, int leg, int64_t *rows_out) {
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (which != SHC_EXTERNAL_TARGET && which != SHC_EXTERNAL_DEFAULT && which != SHC_EXTERNAL_PLANNER_TARGET)
    return fail(SHC_ERR_INVALID_ARG, "which must be SHC_EXTERNAL_TARGET, SHC_EXTERNAL_DEFAULT or SHC_EXTERNAL_PLANNER_TARGET");
  if (first < 0 || count < 0 || first + count > e->n || leg >= e->L) return fail(SHC_ERR_INVALID_ARG, "instance range / leg out of bounds");
  if (which == SHC_EXTERNAL_DEFAULT && !e->params.rough_terrain_mode)
    return fail(SHC_ERR_UNSUPPORTED, "external default poses are read in rough_terrain_mode only (walk_controller.cpp:988)");
  *rows_out = count * (leg < 0 ? e->L : 1);
  HIP_TRY(hipSetDevice(e->device));
  if (which != SHC_EXTERNAL_DEFAULT) { // a target may end up with the planner-mode LegPoser of a robot that stands
    const int rc = ensure_seq(e);
    if (rc != SHC_OK) return rc;
  }
  if (!e->st.ext) { // first request: allocate the records (all undefined)
    const size_t bytes = size_t(ExtFields::COUNT) * e->n_slots * 8;
    HIP_TRY(hipMalloc(&e->st.ext, bytes));
    HIP_TRY(hipMemsetAsync(e->st.ext, 0, bytes, e->stream));
  }
  return SHC_OK;
}

static int external_write(shc_engine *e, int which, int64_t first, int64_t count, int leg, const std::vector<ExtRow> &host, int transform_only,
                          int64_t *ignored) {
  ExtRow *d_rows = nullptr;
  unsigned long long *d_ignored = nullptr, h_ignored = 0;
  HIP_TRY(hipMalloc(&d_rows, host.size() * sizeof(ExtRow) + 8));
  d_ignored = reinterpret_cast<unsigned long long *>(d_rows + host.size());
  hipError_t err = hipMemcpyAsync(d_rows, host.data(), host.size() * sizeof(ExtRow), hipMemcpyHostToDevice, e->stream);
  if (err == hipSuccess) err = hipMemsetAsync(d_ignored, 0, 8, e->stream);
  if (err == hipSuccess) {
    set_external_kernel<<<dim3((unsigned)((host.size() + 255) / 256)), dim3(256), 0, e->stream>>>(e->st, e->L, first, count, leg, which, d_rows, transform_only, d_ignored, e->d_seq, e->params.rough_terrain_mode);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipMemcpyAsync(&h_ignored, d_ignored, 8, hipMemcpyDeviceToHost, e->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  (void)hipFree(d_rows);
  if (err != hipSuccess) return fail(SHC_ERR_HIP, hipGetErrorString(err));
  if (ignored) *ignored = int64_t(h_ignored);
  return SHC_OK;
}

extern "C" int shc_engine_set_external_target(shc_engine *e, int which, int64_t first, int64_t count, int leg, const shc_external_target *rows,
                                              int64_t *ignored) {
  SHC_BUSY_GUARD(e);
  int64_t n_rows = 0;
  int rc = external_select(e, which, first, count, leg, &n_rows);
  if (rc != SHC_OK) return rc;
  if (!rows) return fail(SHC_ERR_INVALID_ARG, "rows is NULL");
  if (ignored) *ignored = 0;
  if (n_rows == 0) return SHC_OK;
  std::vector<ExtRow> host((size_t)n_rows);
  for (int64_t i = 0; i < n_rows; ++i) {
    const shc_external_target &t = rows[i];
    // (a requested target rotation on > 3-DOF legs becomes LegStepper::target_tip_pose_.rotation_: the cycle kernels with the
    //  tip-rotation logic run for such legs whenever rough terrain mode is on; legs with <= 3 joints never read it)
    for (int k = 0; k < 7; ++k) host[i].pose[k] = t.pose[k], host[i].transform[k] = t.transform[k];
    host[i].swing_clearance = t.swing_clearance;
    host[i].flags = double((t.defined ? 1 : 0) | (t.frame_is_odom_ideal ? 2 : 0));
  }
  e->rt_flags |= RT_EXTERNAL;
  return external_write(e, which, first, count, leg, host, 0, ignored);
}

extern "C" int shc_engine_set_external_transform(shc_engine *e, int which, int64_t first, int64_t count, int leg, const double *transform) {
  SHC_BUSY_GUARD(e);
  int64_t n_rows = 0;
  int rc = external_select(e, which, first, count, leg, &n_rows);
  if (rc != SHC_OK) return rc;
  if (!transform) return fail(SHC_ERR_INVALID_ARG, "transform is NULL");
  if (n_rows == 0) return SHC_OK;
  std::vector<ExtRow> host((size_t)n_rows);
  for (int64_t i = 0; i < n_rows; ++i)
    for (int k = 0; k < 7; ++k) host[i].transform[k] = transform[i * 7 + k];
  return external_write(e, which, first, count, leg, host, 1, nullptr);
}

extern "C" int shc_engine_get_external_target(shc_engine *e, int which, int64_t first, int64_t count, int leg, shc_external_target *rows) {
  SHC_BUSY_GUARD(e);
  int64_t n_rows = 0;
  int rc = external_select(e, which, first, count, leg, &n_rows);
  if (rc != SHC_OK) return rc;
  if (!rows) return fail(SHC_ERR_INVALID_ARG, "rows is NULL");
  if (n_rows == 0) return SHC_OK;
  ExtRow *d_rows = nullptr;
  HIP_TRY(hipMalloc(&d_rows, size_t(n_rows) * sizeof(ExtRow)));
  get_external_kernel<<<dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, e->stream>>>(e->st, e->L, first, count, leg, which, d_rows);
  std::vector<ExtRow> host((size_t)n_rows);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) err = hipMemcpyAsync(host.data(), d_rows, size_t(n_rows) * sizeof(ExtRow), hipMemcpyDeviceToHost, e->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  (void)hipFree(d_rows);
  if (err != hipSuccess) return fail(SHC_ERR_HIP, hipGetErrorString(err));
  for (int64_t i = 0; i < n_rows; ++i) {
    for (int k = 0; k < 7; ++k) rows[i].pose[k] = host[i].pose[k], rows[i].transform[k] = host[i].transform[k];
    rows[i].swing_clearance = host[i].swing_clearance;
    rows[i].defined = int(host[i].flags) & 1;
    rows[i].frame_is_odom_ideal = (int(host[i].flags) >> 1) & 1;
  }
  return SHC_OK;
}

extern "C" int shc_engine_get_joint_commands(shc_engine *e, double *position, double *velocity, double *effort, double *position_command, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  HIP_TRY(hipSetDevice(e->device));
  const double *d_off;
  int rc = upload_offsets(e, &d_off);
  if (rc != SHC_OK) return rc;
  const size_t rows = size_t(e->n) * e->L * e->NJ;
  double *out[4] = {position, velocity, effort, position_command}, *dev[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<void *> temps;
  auto release = [&]() {
    for (void *t : temps) (void)hipFree(t);
  };
  for (int k = 0; k < 4; ++k) {
    if (!out[k]) continue;
    if (on_device) {
      dev[k] = out[k];
      continue;
    }
    void *t = nullptr;
    HIP_TRY_OR(hipMalloc(&t, rows * 8), release());
    temps.push_back(t);
    dev[k] = static_cast<double *>(t);
  }
  const int64_t threads = e->n * e->L;
  joint_commands_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(dev[0], dev[1], dev[2], dev[3], e->st.legd, e->n_slots, e->n,
                                                                                           e->L, e->NJ, LEG_FIELD(e, Q), LEG_FIELD(e, QD),
                                                                                           LEG_FIELD(e, EFFORT_IN), d_off);
  HIP_TRY_OR(hipGetLastError(), release());
  if (!on_device)
    for (int k = 0; k < 4; ++k)
      if (out[k]) HIP_TRY_OR(hipMemcpyAsync(out[k], dev[k], rows * 8, hipMemcpyDeviceToHost, e->stream), release());
  HIP_TRY_OR(hipStreamSynchronize(e->stream), release());
  release();
  return SHC_OK;
}

// ---- per-leg Leg API (include/shc_batch.h "Per-leg methods"): host arrays travel through temporary device buffers
struct LegCall {
  shc_engine *e;
  LegSel sel;
  int64_t rows; // count * selected legs
  std::vector<void *> temps;
  ~LegCall() {
    for (void *p : temps) (void)hipFree(p);
  }
  int init(shc_engine *eng, int64_t first, int64_t count, int leg) {
    e = eng;
    if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
    if (first < 0 || count < 0 || first + count > e->n) return fail(SHC_ERR_INVALID_ARG, "instance range out of bounds");
    if (leg < -1 || leg >= e->L) return fail(SHC_ERR_INVALID_ARG, "leg out of range (-1 = every leg)");
    sel = LegSel{first, count, leg, e->L};
    rows = count * (leg < 0 ? e->L : 1);
    HIP_TRY(hipSetDevice(e->device));
    return SHC_OK;
  }
  // device view of an input array of `width` doubles per row
  int in(const double *src, int width, int on_device, const double **d) {
    *d = src;
    if (!src || on_device || rows == 0) return SHC_OK;
    void *t = nullptr;
    HIP_TRY(hipMalloc(&t, size_t(rows) * width * 8));
    temps.push_back(t);
    HIP_TRY(hipMemcpyAsync(t, src, size_t(rows) * width * 8, hipMemcpyHostToDevice, e->stream));
    *d = static_cast<const double *>(t);
    return SHC_OK;
  }
  int out(double *dst, int width, int on_device, double **d) {
    *d = dst;
    if (!dst || on_device || rows == 0) return SHC_OK;
    void *t = nullptr;
    HIP_TRY(hipMalloc(&t, size_t(rows) * width * 8));
    temps.push_back(t);
    *d = static_cast<double *>(t);
    return SHC_OK;
  }
  int finish(double *dst, const double *d, int width, int on_device) {
    HIP_TRY(hipGetLastError());
    if (dst && !on_device && rows) HIP_TRY(hipMemcpyAsync(dst, d, size_t(rows) * width * 8, hipMemcpyDeviceToHost, e->stream));
    if (!temps.empty()) HIP_TRY(hipStreamSynchronize(e->stream)); // the temporaries are freed on return
    return SHC_OK;
  }
  dim3 grid() const { return dim3((unsigned)((rows + 63) / 64)); }
};
template <class Fn>
static int leg_dispatch(shc_engine *e, Fn &&fn) {
#define CALL(L_, NJ_) fn(std::integral_constant<int, L_>{}, std::integral_constant<int, NJ_>{})
  SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
  return SHC_OK;
}
#define LEG_KERNEL(KERNEL, ...)                                                                                              \
  leg_dispatch(e, [&](auto l_, auto nj_) {                                                                                   \
    constexpr int L_ = decltype(l_)::value, NJ_ = decltype(nj_)::value;                                                      \
    if (c.rows) KERNEL<L_, NJ_><<<c.grid(), dim3(64), 0, e->stream>>>(e->st, (const SharedConsts<L_, NJ_> *)e->d_consts, c.sel, __VA_ARGS__); \
  })

extern "C" int shc_leg_set_desired_tip_pose(shc_engine *e, int64_t first, int64_t count, int leg, const double *tip_pose, int apply_delta,
                                            int on_device) {
  SHC_BUSY_GUARD(e);
  LegCall c;
  int rc = c.init(e, first, count, leg);
  if (rc != SHC_OK) return rc;
  if (!tip_pose && (rc = derive_tips(e)) != SHC_OK) return rc; // Pose::Undefined() = "the poser's tip pose" (model.cpp:657)
  const double *d;
  if ((rc = c.in(tip_pose, 7, on_device, &d)) != SHC_OK) return rc;
  if ((rc = LEG_KERNEL(leg_set_desired_kernel, d, apply_delta, e->params.admittance_control, rotations_tracked(e))) != SHC_OK) return rc;
  return c.finish(nullptr, nullptr, 0, on_device);
}

extern "C" int shc_leg_solve_ik(shc_engine *e, int64_t first, int64_t count, int leg, const double *delta, int solve_rotation,
                                double *joint_delta, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!delta || !joint_delta) return fail(SHC_ERR_INVALID_ARG, "delta / joint_delta is NULL");
  LegCall c;
  int rc = c.init(e, first, count, leg);
  if (rc != SHC_OK) return rc;
  const double *d;
  double *o;
  if ((rc = c.in(delta, 6, on_device, &d)) != SHC_OK || (rc = c.out(joint_delta, e->NJ, on_device, &o)) != SHC_OK) return rc;
  if ((rc = LEG_KERNEL(leg_solve_ik_kernel, d, solve_rotation, o)) != SHC_OK) return rc;
  return c.finish(joint_delta, o, e->NJ, on_device);
}

extern "C" int shc_leg_update_joint_positions(shc_engine *e, int64_t first, int64_t count, int leg, const double *joint_delta, int simulation,
                                              double *limit_proximity, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!joint_delta) return fail(SHC_ERR_INVALID_ARG, "joint_delta is NULL");
  LegCall c;
  int rc = c.init(e, first, count, leg);
  if (rc != SHC_OK) return rc;
  const double *d;
  double *o;
  if ((rc = c.in(joint_delta, e->NJ, on_device, &d)) != SHC_OK || (rc = c.out(limit_proximity, 1, on_device, &o)) != SHC_OK) return rc;
  if ((rc = LEG_KERNEL(leg_update_joints_kernel, d, simulation, o, e->params.time_delta, e->params.clamp_joint_velocities,
                       e->params.clamp_joint_positions)) != SHC_OK)
    return rc;
  return c.finish(limit_proximity, o, 1, on_device);
}

extern "C" int shc_leg_apply_ik(shc_engine *e, int64_t first, int64_t count, int leg, int simulation, double *ik_result, int on_device) {
  SHC_BUSY_GUARD(e);
  LegCall c;
  int rc = c.init(e, first, count, leg);
  if (rc != SHC_OK) return rc;
  double *o;
  if ((rc = c.out(ik_result, 1, on_device, &o)) != SHC_OK) return rc;
  if ((rc = LEG_KERNEL(leg_apply_ik_kernel, simulation, o, e->params.time_delta, e->params.clamp_joint_velocities,
                       e->params.clamp_joint_positions, e->cp.tip_force, e->params.force_gain)) != SHC_OK)
    return rc;
  return c.finish(ik_result, o, 1, on_device);
}

extern "C" int shc_leg_apply_fk(shc_engine *e, int64_t first, int64_t count, int leg, const double *joint_position, double *tip_pose,
                                int on_device) {
  SHC_BUSY_GUARD(e);
  if (!tip_pose) return fail(SHC_ERR_INVALID_ARG, "tip_pose is NULL");
  LegCall c;
  int rc = c.init(e, first, count, leg);
  if (rc != SHC_OK) return rc;
  const double *d;
  double *o;
  if ((rc = c.in(joint_position, e->NJ, on_device, &d)) != SHC_OK || (rc = c.out(tip_pose, 7, on_device, &o)) != SHC_OK) return rc;
  if ((rc = LEG_KERNEL(leg_apply_fk_kernel, d, o)) != SHC_OK) return rc;
  return c.finish(tip_pose, o, 7, on_device);
}

// ---- sequences
extern "C" int shc_leg_step_to_position(shc_engine *e, int64_t first, int64_t count, int leg, const double *target_tip_pose, const double *target_pose,
                                        double lift_height, double time_to_step, int apply_delta, double *tip_pose, int32_t *progress, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!target_pose || !tip_pose) return fail(SHC_ERR_INVALID_ARG, "target_pose / tip_pose is NULL");
  if (!(time_to_step >= 0.0)) return fail(SHC_ERR_INVALID_ARG, "time_to_step must be >= 0");
  LegCall c;
  int rc = c.init(e, first, count, leg);
  if (rc != SHC_OK) return rc;
  const double *dt, *dp;
  double *o, *prog;
  if ((rc = c.in(target_tip_pose, 7, on_device, &dt)) != SHC_OK) return rc;
  { // target_pose has one row per INSTANCE
    const int64_t rows = c.rows;
    c.rows = count;
    rc = c.in(target_pose, 7, on_device, &dp);
    c.rows = rows;
    if (rc != SHC_OK) return rc;
  }
  if ((rc = c.out(tip_pose, 7, on_device, &o)) != SHC_OK) return rc;
  // int32 progress rows ride in a double-width temporary (4 bytes used per row)
  if ((rc = c.out(reinterpret_cast<double *>(progress), 1, on_device, &prog)) != SHC_OK) return rc;
  if ((rc = LEG_KERNEL(leg_step_to_position_kernel, dt, dp, lift_height, time_to_step, apply_delta, e->params.admittance_control,
                       e->params.time_delta, o, reinterpret_cast<int32_t *>(prog))) != SHC_OK)
    return rc;
  HIP_TRY(hipGetLastError());
  if (progress && !on_device && c.rows) HIP_TRY(hipMemcpyAsync(progress, prog, size_t(c.rows) * 4, hipMemcpyDeviceToHost, e->stream));
  return c.finish(tip_pose, o, 7, on_device);
}

extern "C" int shc_leg_transition_configuration(shc_engine *e, int64_t first, int64_t count, int leg, const double *desired_configuration,
                                                double transition_time, int32_t *progress, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!desired_configuration) return fail(SHC_ERR_INVALID_ARG, "desired_configuration is NULL");
  LegCall c;
  int rc = c.init(e, first, count, leg);
  if (rc != SHC_OK) return rc;
  const double *d;
  double *prog;
  if ((rc = c.in(desired_configuration, e->NJ, on_device, &d)) != SHC_OK) return rc;
  if ((rc = c.out(reinterpret_cast<double *>(progress), 1, on_device, &prog)) != SHC_OK) return rc;
  if ((rc = LEG_KERNEL(leg_transition_configuration_kernel, d, 1, transition_time, e->params.time_delta, reinterpret_cast<int32_t *>(prog))) != SHC_OK)
    return rc;
  HIP_TRY(hipGetLastError());
  if (progress && !on_device && c.rows) HIP_TRY(hipMemcpyAsync(progress, prog, size_t(c.rows) * 4, hipMemcpyDeviceToHost, e->stream));
  return c.finish(nullptr, nullptr, 0, on_device);
}

// Joints of every instance to the configuration Leg::init(true) leaves (model.cpp:286-305: Joint::default_position_ =
// clamped(0, min, max), :1038), everything else to the engine's post-start-up template.
template <int NJ>
__global__ void set_initial_joints_kernel(DevState st, const double *q0 /*[L][NJ]*/, int L) {
  using FD = Fields<NJ>;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= st.n_robots * L) return;
  const int64_t rob = t / L;
  const int leg = int(t - rob * L);
  const int64_t slot = slot_of(rob, leg, L);
  for (int j = 0; j < NJ; ++j) {
    st.legd[leg_field_index(FD::Q + j, slot, st.n_slots)] = q0[leg * NJ + j];
    st.legd[leg_field_index(FD::QD + j, slot, st.n_slots)] = 0.0;
  }
}

// Q / QD of every leg slot from a saved copy of the joint planes (the TIP fields sharing the last plane keep their new values)
__global__ void restore_joints_kernel(DevState st, const double2 *saved_planes, int L, int NJ) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= st.n_robots * L) return;
  const int64_t rob = t / L;
  const int64_t slot = slot_of(rob, int(t - rob * L), L);
  const double *saved = reinterpret_cast<const double *>(saved_planes);
  for (int f = 0; f < 2 * NJ; ++f) {
    const int64_t i = leg_field_index(f, slot, st.n_slots);
    st.legd[i] = saved[i];
  }
}

extern "C" int shc_engine_begin_direct_startup(shc_engine *e) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  int rc = init_state(e);
  if (rc != SHC_OK) return rc;
  std::vector<double> q0(size_t(e->L) * e->NJ);
  for (int l = 0; l < e->L; ++l)
    for (int j = 0; j < e->NJ; ++j) q0[size_t(l) * e->NJ + j] = clampd(0.0, e->params.joint[l][j].min, e->params.joint[l][j].max);
  HIP_TRY(hipMemcpyAsync(e->d_stage, q0.data(), q0.size() * 8, hipMemcpyHostToDevice, e->stream));
  const int64_t threads = e->n * e->L;
  const dim3 grid((unsigned)((threads + 255) / 256));
  switch (e->NJ) {
    case 3: set_initial_joints_kernel<3><<<grid, dim3(256), 0, e->stream>>>(e->st, e->d_stage, e->L); break;
    case 4: set_initial_joints_kernel<4><<<grid, dim3(256), 0, e->stream>>>(e->st, e->d_stage, e->L); break;
    default: set_initial_joints_kernel<5><<<grid, dim3(256), 0, e->stream>>>(e->st, e->d_stage, e->L); break;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->startup_calls = 0;
  e->starting_up = 1;
  return SHC_OK;
}

extern "C" int shc_engine_direct_startup(shc_engine *e, int32_t *progress) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (!e->starting_up) return fail(SHC_ERR_INVALID_ARG, "call shc_engine_begin_direct_startup first");
  HIP_TRY(hipSetDevice(e->device));
  // desired_configuration_ of every leg = the joints the simulated start-up solve ended on (pose_controller.cpp:476-510):
  // the init chain's default_joint_position, one row per leg shared by all instances
  std::vector<double> target(size_t(e->L) * e->NJ);
  for (int l = 0; l < e->L; ++l)
    for (int j = 0; j < e->NJ; ++j) target[size_t(l) * e->NJ + j] = e->tables.default_joint_position[l][j];
  HIP_TRY(hipMemcpyAsync(e->d_stage, target.data(), target.size() * 8, hipMemcpyHostToDevice, e->stream));
  LegCall c;
  int rc = c.init(e, 0, e->n, -1);
  if (rc != SHC_OK) return rc;
  const double *d = e->d_stage;
  if ((rc = LEG_KERNEL(leg_transition_configuration_kernel, d, 0, e->params.time_to_start, e->params.time_delta, (int32_t *)nullptr)) != SHC_OK) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream)); // the staging buffer is reused
  // progress is a function of the call count alone (every leg runs the same number of iterations, :1546-1566)
  const int num = hostinit::startup_loops(e->params);
  e->startup_calls++;
  int p = int((double(e->startup_calls - 1) / double(num)) * 100);
  p = p < 1 ? 1 : (p > 100 ? 100 : p);
  if (e->startup_calls >= num) {
    p = 100;
    e->starting_up = 0;
    // READY reached; the loop that enters RUNNING also runs the first control cycle (state_controller.cpp:277-281, :189-192)
    if ((rc = shc_engine_step(e, 1)) != SHC_OK) return rc;
  }
  if (progress) *progress = p;
  return SHC_OK;
}

// ---- start-up / shut-down sequences (start_up_sequence: true; shc_sequence.hpp)
__global__ void set_joint_positions_kernel(DevState st, const double *q, int per_instance, int L, int NJ, int f_q, int f_qd, int f_meas) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= st.n_robots * L) return;
  const int64_t rob = t / L;
  const int leg = int(t - rob * L);
  const int64_t slot = slot_of(rob, leg, L);
  const double *row = q + (per_instance ? t : leg) * NJ;
  for (int j = 0; j < NJ; ++j) { // Leg::init(false) (model.cpp:286-305): desired = current = the reported positions, velocities 0
    st.legd[leg_field_index(f_q + j, slot, st.n_slots)] = row[j];
    st.legd[leg_field_index(f_qd + j, slot, st.n_slots)] = 0.0;
    st.legd[leg_field_index(f_meas + j, slot, st.n_slots)] = row[j];
  }
}

extern "C" int shc_engine_begin_sequence_startup(shc_engine *e, const double *joint_positions, int per_instance) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  // (arguments first: an INVALID_ARG return leaves the batch as it was)
  if (per_instance && !joint_positions) return fail(SHC_ERR_INVALID_ARG, "per_instance needs joint_positions");
  const size_t rows = per_instance ? size_t(e->n) * e->L : size_t(e->L);
  if (rows * e->NJ * 8 > e->stage_bytes) return fail(SHC_ERR_INVALID_ARG, "staging buffer too small");
  std::vector<double> q(rows * e->NJ);
  if (joint_positions) {
    memcpy(q.data(), joint_positions, q.size() * 8);
  } else { // READY: every joint at its `unpacked` position (state_controller.cpp:217, :236)
    for (int l = 0; l < e->L; ++l)
      for (int j = 0; j < e->NJ; ++j) q[size_t(l) * e->NJ + j] = e->params.joint[l][j].unpacked;
  }
  e->fresh_pose_controller = true;
  int rc = init_state(e); // StateController::init(): fresh walker / poser state
  e->fresh_pose_controller = false;
  if (rc != SHC_OK) return rc;
  HIP_TRY(hipMemcpyAsync(e->d_stage, q.data(), q.size() * 8, hipMemcpyHostToDevice, e->stream));
  const int64_t threads = e->n * e->L;
  set_joint_positions_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream>>>(e->st, e->d_stage, per_instance, e->L, e->NJ, LEG_FIELD(e, Q),
                                                                                                 LEG_FIELD(e, QD), LEG_FIELD(e, MEAS_Q));
  HIP_TRY(hipGetLastError());
  if (e->d_seq) HIP_TRY(hipMemsetAsync(e->d_seq, 0, sizeof(SeqRobotState) * size_t(e->n), e->stream)); // fresh PoseController
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->starting_up = 0;
  e->pack_step = e->executing_transition = e->transition_calls = 0;
  return SHC_OK;
}

static int ensure_seq(shc_engine *e) { // the per-robot PoseController records of the sequence / planner kernels, on first use
  HIP_TRY(hipSetDevice(e->device));
  if (!e->d_seq) {
    HIP_TRY(hipMalloc(&e->d_seq, sizeof(SeqRobotState) * size_t(e->n)));
    HIP_TRY(hipMemsetAsync(e->d_seq, 0, sizeof(SeqRobotState) * size_t(e->n), e->stream));
  }
  return SHC_OK;
}
static SeqParams seq_params(const shc_engine *e) {
  SeqParams P{};
  P.step_frequency = e->params.step_frequency;
  P.swing_height = e->params.swing_height;
  P.dt = e->params.time_delta;
  P.force_gain = e->params.force_gain;
  P.clamp_vel = e->params.clamp_joint_velocities;
  P.clamp_pos = e->params.clamp_joint_positions;
  P.tip_force = e->cp.tip_force;
  P.have_adm = e->params.admittance_control;
  P.gravity_aligned = rotations_tracked(e);
  P.inclination_posing = e->params.inclination_posing;
  P.gravity_aligned_tips = e->params.gravity_aligned_tips;
  P.pose_pass = e->params.imu_posing || e->params.auto_posing || e->params.inclination_posing;
  P.poser_tip_kept = e->params.auto_posing && !e->params.imu_posing;
  return P;
}
// The body pose has parts that move while a robot stands (IMU PID, auto-pose latches, inclination): the posing part of a loop-level
// call then runs in the cycle kernel (RT_POSE_MARKED, see LOOP_MARK in shc_sequence.hpp).
static bool posing_needs_pose_pass(const shc_engine *e) { return e->params.imu_posing || e->params.auto_posing || e->params.inclination_posing; }
static int pose_pass(shc_engine *e) { // the marked robots' PoseController::updateCurrentPose + admittance update, everybody else untouched
  e->rt_flags |= RT_POSE_MARKED;
  const int rc = shc_engine_step(e, 1);
  e->rt_flags &= ~RT_POSE_MARKED;
  return rc;
}

static int sequence_launch(shc_engine *e, int which /* 0 / 1: executeSequence(START_UP / SHUT_DOWN), 2: stepToNewStance */, int32_t *progress) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  int rc = ensure_seq(e);
  if (rc != SHC_OK) return rc;
  SeqParams P = seq_params(e);
  if (e->cp.gravity_target) { // identity tip rotation of gravity-aligned tips (walk_controller.cpp:37-41)
    const Quat r = from_two_vectors(V3{1, 0, 0}, V3{e->cp.target_dir[0], e->cp.target_dir[1], e->cp.target_dir[2]});
    P.target_rotation[0] = r.w, P.target_rotation[1] = r.x, P.target_rotation[2] = r.y, P.target_rotation[3] = r.z;
  }
  int32_t *d_progress = reinterpret_cast<int32_t *>(e->d_stage); // [n] ints fit the staging buffer (>= n * 8 * 8 bytes)
  const dim3 grid((unsigned)((e->n + 63) / 64)), block(64);
  // Auto posing on its own clock keeps posing through the sequence (pose_controller.cpp:1134-1187 runs in every loop, state_controller.cpp:165-167): the
  // posing part of this loop runs in the cycle kernel for the robots whose sequence is still running (a pose-only pass, as for leg toggles and plan steps)
  const bool own_clock = e->params.auto_posing && e->params.pose_frequency != -1.0;
  // The pose-only pass runs on the manual-leg kernels (the marks live in the ManualRobot records), but no leg is toggled by a sequence: RT_MANUAL_LEGS /
  // RT_MANUAL_LIVE are raised for the duration of this call only - a later shc_engine_step keeps its kernels, shc_engine_step_k / resident mode stay available.
  const unsigned rt_keep = e->rt_flags;
  if (own_clock) {
    if (e->cp.tip_align)
      return fail(SHC_ERR_UNSUPPORTED, "executeSequence / stepToNewStance with auto posing on its own clock on robots that carry the tip-align pose "
                                       "(gravity_aligned_tips on <= 3-DOF legs): the pose-only pass has no tip-align form");
    if ((rc = ensure_manual_records(e)) != SHC_OK) return rc;
    e->rt_flags |= RT_MANUAL_LEGS | RT_MANUAL_LIVE;
    sequence_mark_kernel<<<grid, block, 0, e->stream>>>(e->st.manual, e->d_seq, e->n, which, 1);
    hipError_t herr = hipGetLastError();
    rc = herr == hipSuccess ? pose_pass(e) : fail(SHC_ERR_HIP, hipGetErrorString(herr));
    if (rc != SHC_OK) { // (the marks do not outlive the call: a marked robot would sit out every later cycle)
      sequence_mark_kernel<<<grid, block, 0, e->stream>>>(e->st.manual, e->d_seq, e->n, which, 0);
      e->rt_flags = rt_keep;
      return rc;
    }
    P.posed = 1;
  }
#define CALL(L_, NJ_)                                                                                                                        \
  if (which == 2) step_to_new_stance_kernel<L_, NJ_><<<grid, block, 0, e->stream>>>(e->st, (const SharedConsts<L_, NJ_> *)e->d_consts, e->d_seq, P, d_progress); \
  else execute_sequence_kernel<L_, NJ_><<<grid, block, 0, e->stream>>>(e->st, (const SharedConsts<L_, NJ_> *)e->d_consts, e->d_seq, which, P, d_progress)
  SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
  HIP_TRY(hipGetLastError());
  if (own_clock) {
    sequence_mark_kernel<<<grid, block, 0, e->stream>>>(e->st.manual, e->d_seq, e->n, which, 0);
    e->rt_flags = rt_keep;
    HIP_TRY(hipGetLastError());
  }
  if (progress) HIP_TRY(hipMemcpyAsync(progress, d_progress, size_t(e->n) * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return SHC_OK;
}

extern "C" int shc_engine_execute_sequence(shc_engine *e, int sequence, int32_t *progress) {
  SHC_BUSY_GUARD(e);
  if (sequence != SHC_SEQUENCE_START_UP && sequence != SHC_SEQUENCE_SHUT_DOWN) return fail(SHC_ERR_INVALID_ARG, "sequence must be SHC_SEQUENCE_START_UP or SHC_SEQUENCE_SHUT_DOWN");
  return sequence_launch(e, sequence, progress);
}

extern "C" int shc_engine_step_to_new_stance(shc_engine *e, int32_t *progress) { return sequence_launch(e, 2, progress); }

// ---- manual leg manipulation (shc_sequence.hpp)
static int ensure_manual_records(shc_engine *e) { // the per-robot ManualRobot records (all legs WALKING, no skip marks), on first use; no flag changes
  HIP_TRY(hipSetDevice(e->device));
  if (!e->st.manual) {
    HIP_TRY(hipMalloc(&e->st.manual, sizeof(ManualRobot) * size_t(e->n_rob_pad)));
    HIP_TRY(hipMemsetAsync(e->st.manual, 0, sizeof(ManualRobot) * size_t(e->n_rob_pad), e->stream)); // all WALKING
    std::vector<int32_t> none(size_t(e->n), -1);
    HIP_TRY(hipMemcpyAsync(e->d_stage, none.data(), none.size() * 4, hipMemcpyHostToDevice, e->stream));
    set_manual_inputs_kernel<<<dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, e->stream>>>(e->st.manual, e->n, reinterpret_cast<const int32_t *>(e->d_stage), nullptr,
                                                                                          nullptr, reinterpret_cast<const int32_t *>(e->d_stage), nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return SHC_OK;
}
static int ensure_manual(shc_engine *e, bool planner) {
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (e->cp.tip_align)
    return fail(SHC_ERR_UNSUPPORTED, "manual leg manipulation / planner mode with the tip-align pose (gravity_aligned_tips on <= 3-DOF legs)");
  const int rc = ensure_manual_records(e);
  if (rc != SHC_OK) return rc;
  e->rt_flags |= RT_MANUAL_LEGS | RT_MANUAL_LIVE; // (the toggle resets the manual pose: its group of the robot tile is live from now on)
  return SHC_OK;
}

extern "C" int shc_engine_toggle_leg_state(shc_engine *e, const int32_t *leg_selection, int32_t *result) {
  SHC_BUSY_GUARD(e);
  int rc = ensure_manual(e, false);
  if (rc != SHC_OK) return rc;
  if (!leg_selection) return fail(SHC_ERR_INVALID_ARG, "leg_selection is NULL");
  int32_t *d_sel = reinterpret_cast<int32_t *>(e->d_stage), *d_res = d_sel + e->n, *d_cycle = d_res + e->n;
  HIP_TRY(hipMemcpyAsync(d_sel, leg_selection, size_t(e->n) * 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemsetAsync(d_cycle, 0, 4, e->stream));
  const SeqParams P = seq_params(e);
  const dim3 grid((unsigned)((e->n + 63) / 64)), block(64);
  int phase = LOOP_WHOLE;
#define CALL(L_, NJ_)                                                                                                                              \
  leg_state_toggle_kernel<L_, NJ_><<<grid, block, 0, e->stream>>>(e->st, (const SharedConsts<L_, NJ_> *)e->d_consts, d_sel, P, e->params.virtual_stiffness, \
                                                                  e->params.swing_stiffness_scaler, e->params.load_stiffness_scaler,                \
                                                                  e->params.admittance_control && e->params.dynamic_stiffness, d_res, d_cycle, phase)
  if (posing_needs_pose_pass(e)) { // mark the robots that stand with a request, run the posing part of their loop in the cycle kernel
    phase = LOOP_MARK;
    SHC_DISPATCH(e->L, e->NJ, CALL);
    HIP_TRY(hipGetLastError());
    if ((rc = pose_pass(e)) != SHC_OK) return rc;
    phase = LOOP_AFTER_POSE;
  }
  SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
  HIP_TRY(hipGetLastError());
  int32_t cycle = 0;
  if (result) HIP_TRY(hipMemcpyAsync(result, d_res, size_t(e->n) * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(&cycle, d_cycle, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (cycle) { // robots without a request, or asked to stop first: their loop is one ordinary control cycle; the others keep out of it
    e->rt_flags |= RT_SKIP_MARKED;
    rc = shc_engine_step(e, 1);
    e->rt_flags &= ~RT_SKIP_MARKED;
    if (rc != SHC_OK) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return SHC_OK;
}

extern "C" int shc_engine_set_manual_inputs(shc_engine *e, const int32_t *primary_leg, const double *primary_tip_velocity, const double *primary_tip_position,
                                            const int32_t *secondary_leg, const double *secondary_tip_velocity, const double *secondary_tip_position) {
  SHC_BUSY_GUARD(e);
  int rc = ensure_manual(e, false);
  if (rc != SHC_OK) return rc;
  // staging layout: [primary leg | secondary leg] ints, then four [n][3] double blocks
  const size_t n = size_t(e->n);
  if (n * 8 + 4 * n * 24 > e->stage_bytes) return fail(SHC_ERR_INVALID_ARG, "staging buffer too small");
  char *base = reinterpret_cast<char *>(e->d_stage);
  int32_t *d_p = reinterpret_cast<int32_t *>(base), *d_s = d_p + n;
  double *d_v[4];
  for (int k = 0; k < 4; ++k) d_v[k] = reinterpret_cast<double *>(base + n * 8) + size_t(k) * n * 3;
  const double *src[4] = {primary_tip_velocity, primary_tip_position, secondary_tip_velocity, secondary_tip_position};
  if (primary_leg) HIP_TRY(hipMemcpyAsync(d_p, primary_leg, n * 4, hipMemcpyHostToDevice, e->stream));
  if (secondary_leg) HIP_TRY(hipMemcpyAsync(d_s, secondary_leg, n * 4, hipMemcpyHostToDevice, e->stream));
  for (int k = 0; k < 4; ++k)
    if (src[k]) HIP_TRY(hipMemcpyAsync(d_v[k], src[k], n * 24, hipMemcpyHostToDevice, e->stream));
  set_manual_inputs_kernel<<<dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, e->stream>>>(e->st.manual, e->n, primary_leg ? d_p : nullptr, src[0] ? d_v[0] : nullptr,
                                                                                        src[1] ? d_v[1] : nullptr, secondary_leg ? d_s : nullptr,
                                                                                        src[2] ? d_v[2] : nullptr, src[3] ? d_v[3] : nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  return SHC_OK;
}

extern "C" int shc_engine_get_leg_manipulation_state(shc_engine *e, int32_t *states) {
  SHC_BUSY_GUARD(e);
  if (!e || !states) return fail(SHC_ERR_INVALID_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  int32_t *d = reinterpret_cast<int32_t *>(e->d_stage);
  const int64_t rows = e->n * e->L;
  get_leg_manipulation_state_kernel<<<dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, e->stream>>>(e->st.manual, e->n, e->L, d);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(states, d, size_t(rows) * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return SHC_OK;
}

// ---- planner mode (shc_sequence.hpp: execute_plan_kernel)
static int ensure_planner(shc_engine *e) {
  int rc = ensure_manual(e, true); // the ManualRobot records carry the skip marks of the loop-level kernels
  if (rc != SHC_OK) return rc;
  return ensure_seq(e);
}
static int plan_inputs(shc_engine *e, int64_t first, int64_t count, int reset_plan_step, const double *configuration, const double *body_pose) {
  int rc = ensure_planner(e);
  if (rc != SHC_OK) return rc;
  if (first < 0 || count < 0 || first + count > e->n) return fail(SHC_ERR_INVALID_ARG, "instance range out of bounds");
  if (count == 0) return SHC_OK;
  const size_t cfg_doubles = configuration ? size_t(count) * e->L * e->NJ : 0, pose_doubles = body_pose ? size_t(count) * 7 : 0;
  double *d = nullptr;
  if (cfg_doubles + pose_doubles) {
    HIP_TRY(hipMalloc(&d, (cfg_doubles + pose_doubles) * 8));
    hipError_t err = hipSuccess;
    if (configuration) err = hipMemcpyAsync(d, configuration, cfg_doubles * 8, hipMemcpyHostToDevice, e->stream);
    if (err == hipSuccess && body_pose) err = hipMemcpyAsync(d + cfg_doubles, body_pose, pose_doubles * 8, hipMemcpyHostToDevice, e->stream);
    if (err != hipSuccess) {
      (void)hipFree(d);
      return fail(SHC_ERR_HIP, hipGetErrorString(err));
    }
  }
  plan_inputs_kernel<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, e->stream>>>(e->d_seq, first, count, e->L, e->NJ, reset_plan_step,
                                                                                 configuration ? d : nullptr, body_pose ? d + cfg_doubles : nullptr);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  (void)hipFree(d);
  if (err != hipSuccess) return fail(SHC_ERR_HIP, hipGetErrorString(err));
  return SHC_OK;
}

extern "C" int shc_engine_set_planner_mode(shc_engine *e, int on) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if ((on != 0) == e->planner_mode) return SHC_OK; // plannerModeCallback acts on a change only (state_controller.cpp:1267)
  e->planner_mode = on != 0;
  return on ? plan_inputs(e, 0, e->n, 1, nullptr, nullptr) : SHC_OK; // plan_step_ = 0 (:1273)
}
extern "C" int shc_engine_set_target_configuration(shc_engine *e, int64_t first, int64_t count, const double *configuration) {
  SHC_BUSY_GUARD(e);
  if (!e || !configuration) return fail(SHC_ERR_INVALID_ARG, "NULL argument");
  return plan_inputs(e, first, count, 0, configuration, nullptr);
}
extern "C" int shc_engine_set_target_body_pose(shc_engine *e, int64_t first, int64_t count, const double *pose) {
  SHC_BUSY_GUARD(e);
  if (!e || !pose) return fail(SHC_ERR_INVALID_ARG, "NULL argument");
  return plan_inputs(e, first, count, 0, nullptr, pose);
}

extern "C" int shc_engine_execute_plan(shc_engine *e, int32_t *progress, int32_t *plan_step) {
  SHC_BUSY_GUARD(e);
  int rc = ensure_planner(e);
  if (rc != SHC_OK) return rc;
  int32_t *d_progress = reinterpret_cast<int32_t *>(e->d_stage), *d_step = d_progress + e->n, *d_walking = d_step + e->n;
  HIP_TRY(hipMemsetAsync(d_walking, 0, 4, e->stream));
  const SeqParams P = seq_params(e);
  const int reset_poser_tips = e->plan_poser_tips_current ? 0 : 1;
  const dim3 grid((unsigned)((e->n + 63) / 64)), block(64);
  int phase = LOOP_WHOLE;
#define CALL(L_, NJ_)                                                                                                                             \
  execute_plan_kernel<L_, NJ_><<<grid, block, 0, e->stream>>>(e->st, (const SharedConsts<L_, NJ_> *)e->d_consts, e->d_seq, P, reset_poser_tips, d_progress, \
                                                              d_step, d_walking, phase)
  if (posing_needs_pose_pass(e)) { // mark the robots that stand, run the posing part of their loop in the cycle kernel
    phase = LOOP_MARK;
    SHC_DISPATCH(e->L, e->NJ, CALL);
    HIP_TRY(hipGetLastError());
    if ((rc = pose_pass(e)) != SHC_OK) return rc;
    phase = LOOP_AFTER_POSE;
  }
  SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
  HIP_TRY(hipGetLastError());
  e->plan_poser_tips_current = true;
  int32_t walking = 0;
  if (progress) HIP_TRY(hipMemcpyAsync(progress, d_progress, size_t(e->n) * 4, hipMemcpyDeviceToHost, e->stream));
  if (plan_step) HIP_TRY(hipMemcpyAsync(plan_step, d_step, size_t(e->n) * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(&walking, d_walking, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (walking) { // the loop of the robots that are still walking is the normal control cycle (inputs zeroed); the others keep out of it
    e->rt_flags |= RT_SKIP_MARKED;
    rc = shc_engine_step(e, 1);
    e->rt_flags &= ~RT_SKIP_MARKED;
    if (rc != SHC_OK) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return SHC_OK;
}

// PoseController::packLegs / unpackLegs (pose_controller.cpp:615-707)
static int pack_transition(shc_engine *e, const double *packed_positions, int n_pack_steps, double transition_time, bool unpack, int32_t *progress) {
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (!packed_positions || n_pack_steps < 1) return fail(SHC_ERR_INVALID_ARG, "packed_positions / n_pack_steps");
  if (!(transition_time > 0)) return fail(SHC_ERR_INVALID_ARG, "transition time must be > 0");
  HIP_TRY(hipSetDevice(e->device));
  if (e->pack_step >= n_pack_steps) e->pack_step = n_pack_steps - 1;
  const size_t per_step = size_t(e->L) * e->NJ;
  std::vector<double> target(per_step);
  for (int l = 0; l < e->L; ++l) // desired_configuration_ of every leg (:628-642, :676-693); the kernel latches it when a transition begins
    for (int j = 0; j < e->NJ; ++j) {
      const size_t k = size_t(l) * e->NJ + j;
      if (!unpack) target[k] = packed_positions[size_t(e->pack_step) * per_step + k];
      else target[k] = e->pack_step > 0 ? packed_positions[size_t(e->pack_step - 1) * per_step + k] : e->params.joint[l][j].unpacked;
    }
  HIP_TRY(hipMemcpyAsync(e->d_stage, target.data(), target.size() * 8, hipMemcpyHostToDevice, e->stream));
  LegCall c;
  int rc = c.init(e, 0, e->n, -1);
  if (rc != SHC_OK) return rc;
  const double *d = e->d_stage;
  if ((rc = LEG_KERNEL(leg_transition_configuration_kernel, d, 0, transition_time, e->params.time_delta, (int32_t *)nullptr)) != SHC_OK) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream)); // the staging buffer is reused
  // LegPoser::transitionConfiguration's progress is a function of its iteration count alone (:1546-1566)
  int num = round_to_int(transition_time / e->params.time_delta);
  num = num > 1 ? num : 1;
  e->transition_calls++;
  int p = int((double(e->transition_calls - 1) / double(num)) * 100);
  p = p < 1 ? 1 : (p > 100 ? 100 : p);
  if (e->transition_calls >= num) {
    p = 100;
    e->transition_calls = 0;
  }
  e->executing_transition = (p != 0 && p != 100);
  if (!unpack) {
    if (p == 100 && e->pack_step < n_pack_steps - 1) {
      e->executing_transition = 0;
      e->pack_step++;
      p = 0;
    }
  } else if (p == 100 && e->pack_step != 0) {
    e->executing_transition = 0;
    e->pack_step--;
    p = 0;
  }
  if (progress) *progress = p;
  return SHC_OK;
}

extern "C" int shc_engine_pack_legs(shc_engine *e, const double *packed_positions, int n_pack_steps, double time_to_pack, int32_t *progress) {
  SHC_BUSY_GUARD(e);
  return pack_transition(e, packed_positions, n_pack_steps, time_to_pack, false, progress);
}

extern "C" int shc_engine_unpack_legs(shc_engine *e, const double *packed_positions, int n_pack_steps, double time_to_unpack, int32_t *progress) {
  SHC_BUSY_GUARD(e);
  return pack_transition(e, packed_positions, n_pack_steps, time_to_unpack, true, progress);
}

__global__ void copy_joint_planes_kernel(double2 *dst, const double2 *src, int64_t count) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t < count) dst[t] = src[t];
}

extern "C" int shc_engine_finish_sequence_startup(shc_engine *e) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  HIP_TRY(hipSetDevice(e->device));
  // Auto posing on its own clock (pose_frequency != -1) poses the body through the sequence calls (sequence_launch), and the PoseController lives on through
  // walker_->init() (:306): its phase counter, the posers' latches and Model::current_pose_ carry over, Leg::generateWorkspace searches at the pose of THAT loop
  // (model.cpp:338) and the runningState() of the same loop reuses it (updateCurrentPose, :165-167, ran once, before the sequence call).  Here: the poser part of
  // every instance's state record is kept across the re-initialisation, the workspace search runs at the kept pose, and the phase counter is set back by one so
  // that the posing part of the full cycle below recomputes the pose of this loop instead of the next one - exact because that part is idempotent in this
  // configuration: with its own clock an AutoPoser's latches are OR-latches that are already set (pose_controller.cpp:1359-1371), LegPoser::updateAutoPose
  // (:1716-1778) is a function of the phase and a latch the same phase sets, the manual pose is checked to be at rest.  IMU posing is the exception
  // (updateIMUPose needs RUNNING, :836: the loop that completes START_UP poses with the auto pose, the cycle below would pose with the PID) and stays refused.
  const bool own_clock = e->params.auto_posing && e->params.pose_frequency != -1.0;
  std::vector<shc_instance_state> kept;
  Pose ws_pose = pose_identity();
  if (own_clock) {
    if (e->params.imu_posing)
      return fail(SHC_ERR_UNSUPPORTED, "finish_sequence_startup with auto posing on its own clock AND IMU posing (the loop that completes START_UP still poses with "
                                       "the auto pose, pose_controller.cpp:836; start such robots with the direct start-up, shc_engine_create)");
    if (e->cp.tip_align) return fail(SHC_ERR_UNSUPPORTED, "finish_sequence_startup with auto posing on its own clock on robots that carry the tip-align pose");
    kept.resize(size_t(e->n));
    int rck = shc_engine_get_state(e, 0, e->n, kept.data());
    if (rck != SHC_OK) return rck;
    const shc_instance_state &k0 = kept[0];
    for (int64_t i = 0; i < e->n; ++i) {
      const shc_instance_state &k = kept[size_t(i)];
      bool same = k.pose_phase == k0.pose_phase && memcmp(k.auto_poser_flags, k0.auto_poser_flags, sizeof(k.auto_poser_flags)) == 0;
      bool rest = k.manual_pose[0] == 0 && k.manual_pose[1] == 0 && k.manual_pose[2] == 0 && k.manual_pose[4] == 0 && k.manual_pose[5] == 0 && k.manual_pose[6] == 0;
      for (int a = 0; a < 3; ++a) rest = rest && k.translation_velocity_input[a] == 0 && k.rotation_velocity_input[a] == 0;
      if (!same)
        return fail(SHC_ERR_UNSUPPORTED, "finish_sequence_startup: the instances completed START_UP in different phases of the auto pose (instance " + std::to_string(i) +
                                             " vs instance 0): the workspace is searched at the body pose of the completing loop and one engine has one set of tables");
      if (!rest)
        return fail(SHC_ERR_UNSUPPORTED, "finish_sequence_startup with auto posing on its own clock while the manual body pose is in motion (instance " + std::to_string(i) + ")");
    }
    ws_pose = Pose{V3{k0.current_pose[0], k0.current_pose[1], k0.current_pose[2]}, Quat{k0.current_pose[3], k0.current_pose[4], k0.current_pose[5], k0.current_pose[6]}};
  }
  // Model::updateDefaultConfiguration + generateWorkspaces + generateWalkspace (state_controller.cpp:307-310): the tables of an
  // engine belong to its morphology, so the configuration of instance 0 stands for the batch.  Robots that were started from
  // different joint positions (shc_engine_begin_sequence_startup per_instance) end their sequences on the same READY stance
  // only up to the tip tolerances of the last steps (a few milliradians): the largest difference to instance 0 is checked here -
  // beyond 0.02 rad (IK_TOLERANCE, 5 mm, at the end of a 0.25 m leg) the batch does not share one configuration and the caller has to
  // start such robots in engines of their own.
  std::vector<double> q(size_t(e->n) * e->L * e->NJ);
  int rc = shc_engine_get_joint_state(e, q.data(), nullptr, 0);
  if (rc != SHC_OK) return rc;
  {
    const size_t row = size_t(e->L) * e->NJ;
    double worst = 0.0;
    for (int64_t i = 1; i < e->n; ++i)
      for (size_t k = 0; k < row; ++k) worst = fmax(worst, fabs(q[size_t(i) * row + k] - q[k]));
    if (!(worst <= 0.02))
      return fail(SHC_ERR_UNSUPPORTED, "finish_sequence_startup: the instances ended their start-up sequences on different configurations (max |dq| to instance 0 = " +
                                           std::to_string(worst) + " rad): one engine has one set of workspace / limit tables");
  }
  shc_tables t;
  bool ok = false;
  switch (e->NJ) {
    case 3: ok = hostinit::generate_tables<3>(e->params, t, q.data(), own_clock ? &ws_pose : nullptr); break;
    case 4: ok = hostinit::generate_tables<4>(e->params, t, q.data(), own_clock ? &ws_pose : nullptr); break;
    default: ok = hostinit::generate_tables<5>(e->params, t, q.data(), own_clock ? &ws_pose : nullptr); break;
  }
  if (!ok) return fail(SHC_ERR_INVALID_ARG, "init chain failed for the configuration the sequence ended on");
  e->tables = t;
  e->span_dirty = true;
  build_cycle_params(e->params, e->tables, e->features, e->rt_flags, e->cp);
  if ((rc = upload_consts(e)) != SHC_OK) return rc;
  // walker_->init() (:306): fresh LegSteppers / walk state; the joints stay where the sequence left them
  const int n_joint_planes = (2 * e->NJ + 1) / 2 + 1; // planes holding Q and QD (Fields: Q = 0, QD = NJ, TIP = 2 NJ)
  const int64_t count = int64_t(n_joint_planes) * e->n_slots;
  double2 *keep = nullptr;
  HIP_TRY(hipMalloc(&keep, size_t(count) * 16));
  const dim3 grid((unsigned)((count + 255) / 256));
  copy_joint_planes_kernel<<<grid, dim3(256), 0, e->stream>>>(keep, reinterpret_cast<const double2 *>(e->st.legd), count);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  if (err == hipSuccess && (rc = init_state(e)) == SHC_OK) {
    restore_joints_kernel<<<dim3((unsigned)((e->n * e->L + 255) / 256)), dim3(256), 0, e->stream>>>(e->st, keep, e->L, e->NJ);
    err = hipGetLastError();
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  }
  (void)hipFree(keep);
  if (err != hipSuccess) return fail(SHC_ERR_HIP, hipGetErrorString(err));
  if (rc != SHC_OK) return rc;
  if (own_clock) { // the PoseController's part of the record carries over, one phase back (see above)
    std::vector<shc_instance_state> fresh(size_t(e->n));
    if ((rc = shc_engine_get_state(e, 0, e->n, fresh.data())) != SHC_OK) return rc;
    const int len = e->tables.pose_phase_length > 0 ? e->tables.pose_phase_length : 1;
    for (int64_t i = 0; i < e->n; ++i) {
      shc_instance_state &f = fresh[size_t(i)];
      const shc_instance_state &k = kept[size_t(i)];
      f.pose_phase = (k.pose_phase + len - 1) % len;
      f.auto_posing_state = k.auto_posing_state;
      memcpy(f.auto_poser_flags, k.auto_poser_flags, sizeof(f.auto_poser_flags));
      memcpy(f.auto_pose_rotation, k.auto_pose_rotation, sizeof(f.auto_pose_rotation));
      memcpy(f.current_pose, k.current_pose, sizeof(f.current_pose));
      for (int l = 0; l < SHC_MAX_LEGS; ++l) f.leg[l].negate_auto_pose = k.leg[l].negate_auto_pose;
    }
    const unsigned rt_keep = e->rt_flags; // (the injected manual pose is the identity it was: its group of the robot tile stays as live as it was)
    rc = shc_engine_set_state(e, 0, e->n, fresh.data());
    e->rt_flags = rt_keep;
    if (rc != SHC_OK) return rc;
  }
  // robot_state_ = RUNNING, and the runningState() of the same loop (:189-192)
  if ((rc = shc_engine_step(e, 1)) != SHC_OK) return rc;
  return shc_engine_synchronize(e);
}

extern "C" int64_t shc_sizeof_instance_state(void) { return (int64_t)sizeof(shc_instance_state); }

// Snapshot records travel through a temporary device buffer (checkpoint / injection are not per-cycle operations).
static int state_transfer(shc_engine *e, int64_t first, int64_t count, shc_instance_state *out, const shc_instance_state *in) {
  { // (a state record shows / replaces the legs' phases as the accepting loop leaves them)
    const int rc_remap = flush_step_remap(e);
    if (rc_remap != SHC_OK) return rc_remap;
  }
  if (!e || (!out && !in)) return fail(SHC_ERR_INVALID_ARG, "NULL argument");
  if (first < 0 || count < 0 || first + count > e->n) return fail(SHC_ERR_INVALID_ARG, "instance range out of bounds");
  if (count == 0) return SHC_OK;
  if (in) e->rt_flags |= RT_MANUAL_LIVE; // an injected state may carry any manual pose
  if (in) // touchdown detection is one flag per engine: on as soon as any injected record has it (tip-state messages arrive for a whole robot)
    for (int64_t i = 0; i < count; ++i)
      if (in[i].touchdown_detection) e->rt_flags |= RT_TOUCHDOWN;
  if (in && !(e->rt_flags & RT_EFFORT_LIVE)) { // a non-zero tip-force filter state decays over the following cycles: evaluate it
    bool any = false;
    for (int64_t i = 0; i < count && !any; ++i)
      for (int l = 0; l < e->L; ++l)
        for (int k = 0; k < 3; ++k) any |= in[i].leg[l].tip_force_calculated[k] != 0.0;
    if (any) {
      const int rc = effort_live(e);
      if (rc != SHC_OK) return rc;
    }
  }
  const int touchdown = (e->rt_flags & RT_TOUCHDOWN) ? 1 : 0;
  unsigned long_legs = 0;
  for (int l = 0; l < e->L; ++l) long_legs |= e->params.leg_dof[l] > 3 ? 1u << l : 0u;
  HIP_TRY(hipSetDevice(e->device));
  shc_instance_state *d = nullptr;
  const size_t bytes = size_t(count) * sizeof(shc_instance_state);
  HIP_TRY(hipMalloc(&d, bytes));
  hipError_t err = hipSuccess;
  const dim3 grid((unsigned)((count + 63) / 64)), block(64);
  if (in) err = hipMemcpyAsync(d, in, bytes, hipMemcpyHostToDevice, e->stream);
  if (err == hipSuccess) {
    switch (e->NJ) {
      case 3: if (in) set_state_kernel<3><<<grid, block, 0, e->stream>>>(d, e->st, e->cp, e->L, first, count, long_legs);
              else get_state_kernel<3><<<grid, block, 0, e->stream>>>(d, e->st, e->cp, e->L, first, count, touchdown, long_legs);
              break;
      case 4: if (in) set_state_kernel<4><<<grid, block, 0, e->stream>>>(d, e->st, e->cp, e->L, first, count, long_legs);
              else get_state_kernel<4><<<grid, block, 0, e->stream>>>(d, e->st, e->cp, e->L, first, count, touchdown, long_legs);
              break;
      default: if (in) set_state_kernel<5><<<grid, block, 0, e->stream>>>(d, e->st, e->cp, e->L, first, count, long_legs);
               else get_state_kernel<5><<<grid, block, 0, e->stream>>>(d, e->st, e->cp, e->L, first, count, touchdown, long_legs);
               break;
    }
    err = hipGetLastError();
  }
  if (err == hipSuccess && out) err = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, e->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  (void)hipFree(d);
  if (err != hipSuccess) return fail(SHC_ERR_HIP, std::string("state transfer: ") + hipGetErrorString(err));
  return SHC_OK;
}
extern "C" int shc_engine_get_state(shc_engine *e, int64_t first, int64_t count, shc_instance_state *states) {
  SHC_BUSY_GUARD(e);
  if (!states) return fail(SHC_ERR_INVALID_ARG, "states is NULL");
  return state_transfer(e, first, count, states, nullptr);
}
extern "C" int shc_engine_set_state(shc_engine *e, int64_t first, int64_t count, const shc_instance_state *states) {
  SHC_BUSY_GUARD(e);
  if (!states) return fail(SHC_ERR_INVALID_ARG, "states is NULL");
  return state_transfer(e, first, count, nullptr, states);
}

// ---- auxiliary state: what only the calls AROUND the control cycle keep (shc_instance_state covers the cycle itself)
struct AuxHeader {
  uint32_t magic;   // 'SHCA'
  uint16_t version; // layout version of this blob
  uint8_t legs, dof;
  uint32_t flags;   // 1: manual-leg record live, 2: external target records live, 4: sequence / planner record live,
                    // 8: the LegPoser tips are state (plan calls under time-dependent posing since the last control cycle)
  int32_t reset_mode; // PoseController::pose_reset_mode_ (RobotFields::I_RESET_MODE: written by the toggle kernel, read by the cycle)
};
constexpr uint32_t kAuxMagic = 0x41434853u;
constexpr uint16_t kAuxVersion = 2; // 2: + the LegPoser tip positions (POSER_TIP) and flag 8
static size_t aux_leg_doubles(int NJ) { // per leg: ExtFields record + leg fields [DES_TIP, COUNT) + the LegPoser tip position
  const int tail = NJ == 3 ? Fields<3>::COUNT - Fields<3>::DES_TIP : (NJ == 4 ? Fields<4>::COUNT - Fields<4>::DES_TIP : Fields<5>::COUNT - Fields<5>::DES_TIP);
  return size_t(ExtFields::COUNT) + size_t(tail) + 3;
}
static size_t aux_bytes(const shc_engine *e) {
  return sizeof(AuxHeader) + sizeof(ManualRobot) + sizeof(SeqRobotState) + size_t(e->L) * aux_leg_doubles(e->NJ) * 8;
}
__global__ void aux_state_kernel(unsigned char *blobs, size_t stride, DevState st, SeqRobotState *seq, int L, int NJ, int des_tip_field, int n_leg_fields, int64_t first,
                                 int64_t count, int to_engine, uint32_t live_flags, int poser_tip_field) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int64_t rob = first + t;
  unsigned char *b = blobs + size_t(t) * stride;
  AuxHeader *h = reinterpret_cast<AuxHeader *>(b);
  ManualRobot *m = reinterpret_cast<ManualRobot *>(b + sizeof(AuxHeader));
  SeqRobotState *q = reinterpret_cast<SeqRobotState *>(b + sizeof(AuxHeader) + sizeof(ManualRobot));
  double *legs = reinterpret_cast<double *>(b + sizeof(AuxHeader) + sizeof(ManualRobot) + sizeof(SeqRobotState));
  const int tail = n_leg_fields - des_tip_field;
  const int per_leg = ExtFields::COUNT + tail + 3;
  int32_t &reset_mode = st.robi[rob_index(rob, RobotFields::I_RESET_MODE, 64 / L, RobotFields::I_COUNT)];
  if (!to_engine) {
    h->magic = kAuxMagic, h->version = kAuxVersion, h->legs = uint8_t(L), h->dof = uint8_t(NJ), h->flags = live_flags, h->reset_mode = reset_mode;
    if (st.manual) *m = st.manual[rob];
    else memset(m, 0, sizeof(ManualRobot));
    if (seq) *q = seq[rob];
    else memset(q, 0, sizeof(SeqRobotState));
  } else {
    reset_mode = h->reset_mode;
    if (st.manual) {
      if (h->flags & 1) st.manual[rob] = *m;
      else memset(&st.manual[rob], 0, sizeof(ManualRobot));
    }
    if (seq) {
      if (h->flags & 4) seq[rob] = *q;
      else memset(&seq[rob], 0, sizeof(SeqRobotState));
    }
  }
  for (int leg = 0; leg < L; ++leg) {
    const int64_t slot = slot_of(rob, leg, L);
    double *row = legs + size_t(leg) * per_leg;
    for (int f = 0; f < ExtFields::COUNT; ++f) {
      if (!to_engine) row[f] = st.ext ? st.ext[leg_field_index(f, slot, st.n_slots)] : 0.0;
      else if (st.ext) st.ext[leg_field_index(f, slot, st.n_slots)] = (h->flags & 2) ? row[f] : 0.0;
    }
    for (int f = 0; f < tail; ++f) {
      double &x = st.legd[leg_field_index(des_tip_field + f, slot, st.n_slots)];
      if (!to_engine) row[ExtFields::COUNT + f] = x;
      else x = row[ExtFields::COUNT + f];
    }
    for (int f = 0; f < 3; ++f) { // LegPoser::current_tip_pose_.position_ (state while flag 8 holds, an output otherwise)
      double &x = st.legd[leg_field_index(poser_tip_field + f, slot, st.n_slots)];
      if (!to_engine) row[ExtFields::COUNT + tail + f] = x;
      else if (h->flags & 8) x = row[ExtFields::COUNT + tail + f];
    }
  }
}
extern "C" int64_t shc_engine_aux_state_bytes(const shc_engine *e) { return e ? int64_t(aux_bytes(e)) : 0; }
static int aux_state(shc_engine *e, int64_t first, int64_t count, void *blobs, int to_engine) {
  if (!e || !blobs) return fail(SHC_ERR_INVALID_ARG, "NULL argument");
  if (first < 0 || count < 0 || first + count > e->n) return fail(SHC_ERR_INVALID_ARG, "instance range out of bounds");
  if (count == 0) return SHC_OK;
  HIP_TRY(hipSetDevice(e->device));
  const size_t stride = aux_bytes(e);
  uint32_t want = 0;
  if (to_engine) { // the engine grows the records the blobs carry
    for (int64_t i = 0; i < count; ++i) {
      const AuxHeader *h = reinterpret_cast<const AuxHeader *>(static_cast<const unsigned char *>(blobs) + size_t(i) * stride);
      if (h->magic != kAuxMagic || h->version != kAuxVersion || h->legs != e->L || h->dof != e->NJ)
        return fail(SHC_ERR_INVALID_ARG, "auxiliary state blob of another library version / morphology");
      want |= h->flags;
    }
    int rc = SHC_OK;
    if ((want & 1) && !e->st.manual) rc = ensure_manual(e, false);
    if (rc == SHC_OK && (want & 4) && !e->d_seq) rc = ensure_seq(e);
    double *ext_new = nullptr;
    if (rc == SHC_OK && (want & 2) && !e->st.ext) { // (allocated and cleared completely before the engine sees it: a failure leaves nothing half-grown)
      const size_t bytes = size_t(ExtFields::COUNT) * e->n_slots * 8;
      if (hipMalloc(&ext_new, bytes) != hipSuccess) rc = fail(SHC_ERR_HIP, "hipMalloc(external target records)");
      else if (hipMemsetAsync(ext_new, 0, bytes, e->stream) != hipSuccess) {
        (void)hipFree(ext_new);
        rc = fail(SHC_ERR_HIP, "hipMemset(external target records)");
      }
    }
    if (rc != SHC_OK) return rc;
    if (ext_new) e->st.ext = ext_new;
    if (want & 1) e->rt_flags |= RT_MANUAL_LEGS | RT_MANUAL_LIVE;
    if (want & 2) e->rt_flags |= RT_EXTERNAL;
    // "The LegPoser tips of the last plan call are still current" is a fact about the whole engine (any control cycle clears it): a restore
    // of the whole batch sets it from the blobs; a partial restore / migration of a few instances can only keep it when both sides agree -
    // it never raises it for the instances it did not touch, and the per-blob flag stays authoritative for the restored ones.
    if (first == 0 && count == e->n) e->plan_poser_tips_current = (want & 8) != 0;
    else e->plan_poser_tips_current = e->plan_poser_tips_current && (want & 8) != 0;
  }
  unsigned char *d = nullptr;
  HIP_TRY(hipMalloc(&d, stride * size_t(count)));
  if (to_engine) HIP_TRY_OR(hipMemcpyAsync(d, blobs, stride * size_t(count), hipMemcpyHostToDevice, e->stream), (void)hipFree(d));
  const uint32_t live = (e->st.manual && (e->rt_flags & RT_MANUAL_LEGS) ? 1u : 0u) | (e->st.ext ? 2u : 0u) | (e->d_seq ? 4u : 0u) |
                        (e->plan_poser_tips_current ? 8u : 0u);
  aux_state_kernel<<<dim3((unsigned)((count + 127) / 128)), dim3(128), 0, e->stream>>>(d, stride, e->st, e->d_seq, e->L, e->NJ, LEG_FIELD(e, DES_TIP), e->n_leg_fields,
                                                                                   first, count, to_engine, live, LEG_FIELD(e, POSER_TIP));
  HIP_TRY_OR(hipGetLastError(), (void)hipFree(d));
  if (!to_engine) HIP_TRY_OR(hipMemcpyAsync(blobs, d, stride * size_t(count), hipMemcpyDeviceToHost, e->stream), (void)hipFree(d));
  HIP_TRY_OR(hipStreamSynchronize(e->stream), (void)hipFree(d));
  (void)hipFree(d);
  return SHC_OK;
}
extern "C" int shc_engine_get_aux_state(shc_engine *e, int64_t first, int64_t count, void *blobs) {
  SHC_BUSY_GUARD(e);
  return aux_state(e, first, count, blobs, 0);
}
extern "C" int shc_engine_set_aux_state(shc_engine *e, int64_t first, int64_t count, const void *blobs) {
  SHC_BUSY_GUARD(e);
  return aux_state(e, first, count, const_cast<void *>(blobs), 1);
}

extern "C" int shc_engine_get_body_state(shc_engine *e, double *pose, double *velocity, int32_t *walk_state, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  int rc;
  if ((rc = gather_rob(e, pose, 7, RobotFields::CPOSE, on_device)) != SHC_OK) return rc;
  if ((rc = gather_rob(e, velocity, 3, RobotFields::VLIN, on_device)) != SHC_OK) return rc;
  if (walk_state) {
    HIP_TRY(hipSetDevice(e->device));
    int32_t *d = on_device ? walk_state : reinterpret_cast<int32_t *>(e->d_stage);
    gather_walk_state_kernel<<<dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, e->stream>>>(d, e->st.robi, 64 / e->L, e->n);
    HIP_TRY(hipGetLastError());
    if (!on_device) {
      HIP_TRY(hipMemcpyAsync(walk_state, d, size_t(e->n) * 4, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipStreamSynchronize(e->stream));
    }
  }
  return SHC_OK;
}

#include "shc_fleet.hpp" // shc_fleet_*: mixed morphologies + multi-device sharding (uses the entry points above)

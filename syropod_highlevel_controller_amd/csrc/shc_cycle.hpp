// shc_cycle.hpp — the fused per-cycle device code of the batched leg-control engine (gfx950).
//
// Mapping: one leg per lane, one robot per L-lane group, floor(64 / L) robots per
// wavefront (10 hexapods or 8 octopods per wave).  All per-leg linear algebra (<= 6x6) stays in that lane's
// registers.  Per-robot state (velocity, walk FSM word, walk plane, body-pose components, PID state) lives in an
// LDS tile owned by the wave — every lane of a group reads the same address (LDS broadcast), only lane 0 of the
// group writes — so the replicated per-robot scalars cost no VGPRs between phases.  The launch-uniform parameter
// block, the per-leg DH / joint-limit records and the 4 x 9 velocity-limit maps are staged in LDS once per
// workgroup.  The only cross-lane traffic is a handful of ds_bpermute shuffles (packed leg words, limit brackets,
// walk-plane sums).  HBM state is structure-of-arrays, one 8-byte slot per (field, lane): every load/store of a
// wave is one fully coalesced 512-byte segment.
//
// Reference call order restated (OpenSHC v0.5.11, StateController::loop, src/state_controller.cpp:162-193,
// runningState :379-447):
//   PoseController::updateCurrentPose   src/pose_controller.cpp:811
//   AdmittanceController::updateStiffness / updateAdmittance   src/admittance_controller.cpp:96 / :22
//   WalkController::updateWalk          src/walk_controller.cpp:440   (+ LegStepper :862-1329)
//   PoseController::updateStance        src/pose_controller.cpp:110
//   Model::updateModel                  src/model.cpp:142  (Leg::setDesiredTipPose :653, Leg::applyIK :861)
#pragma once

#include "../../include/shc_batch.h"
#include "shc_leg.hpp"

namespace shc {

enum : int { WS_STARTING = 0, WS_MOVING = 1, WS_STOPPING = 2, WS_STOPPED = 3 };     // parameters_and_states.h:99
enum : int { SS_SWING = 0, SS_STANCE = 1, SS_FORCE_STANCE = 2, SS_FORCE_STOP = 3 }; // :111
enum : int { PS_POSING = 0, PS_STOP_POSING = 1, PS_POSING_COMPLETE = 2 };           // :123
// how LegStepper::swing_progress_/stance_progress_ currently read (set by iteratePhase, walk_controller.cpp:871-897)
enum : int { PM_NONE = 0, PM_SWING = 1, PM_STANCE = 2, PM_STOP = 3 };

// packed per-leg word
constexpr int LW_ACP = 1 << 2, LW_CFS = 1 << 3, LW_PM_SHIFT = 4, LW_NEG = 1 << 6, LW_IKFAIL = 1 << 7, LW_PHASE_SHIFT = 8,
              LW_PHASE_MASK = 0xFFFFF, LW_ZBV = 1 << 28, LW_ATT = 1 << 29, LW_ROTDEF = 1 << 30, // ROTDEF: walker tip rotation defined
              LW_TARGROT = int(1u << 31); // LegStepper::target_tip_pose_.rotation_ defined (the direction is in Fields::TARG_DIR)
// packed per-robot word: walk state [0:1], legs_at_correct_phase [2:5], legs_completed_first_step [6:9],
// return_to_default_attempted [10], auto_posing_state [11:12]
constexpr int RW_LACP_SHIFT = 2, RW_LCFS_SHIFT = 6, RW_RTDA = 1 << 10, RW_APS_SHIFT = 11;

constexpr int kMaxAutoPosers = 8;
constexpr int kSwingTable = 64;  // entries of the walk-plane-pose control-input table (one per swing iteration)
// "this group of state fields changed during the launch" bits, kept per lane and OR-reduced over the wave before the
// write-back: groups nobody changed are not stored (walk plane / manual pose of the robot tile; the parked stepper
// origins of the per-leg planes, which change once per step period).
enum : unsigned { DIRTY_WALK_PLANE = 1, DIRTY_MANUAL = 2, DIRTY_SWING_ORG = 4, DIRTY_STANCE_ORG = 8,
                  DIRTY_TARG_DIR = 16, // the stepper's target tip direction (assigned once per walk, or by an external target): its planes are written back only then
                  DIRTY_LAST = DIRTY_TARG_DIR };
// launch-uniform run-time facts passed as a kernel argument (see shc_cycle_kernel)
enum : unsigned { RT_MANUAL_LIVE = 1, RT_TOUCHDOWN = 2, RT_EXTERNAL = 4, RT_MANUAL_LEGS = 8, RT_EFFORT_LIVE = 16, RT_SKIP_MARKED = 32, RT_POSE_MARKED = 64 }; // RT_POSE_MARKED: run only the posing part of the loop (updateCurrentPose, admittance), and only for the robots a loop-level kernel marked; // RT_MANUAL_LEGS: a leg has been toggled (ManualRobot records exist); // RT_EXTERNAL: external targets / defaults have been requested; // RT_TOUCHDOWN: tip-state (wrench) messages have arrived (walk_controller.h:495)

// Feature mask of a kernel specialisation.  F_DYN: every feature is compiled in and selected by the runtime flags.
enum : unsigned { F_MANUAL = 1, F_AUTO = 2, F_INCL = 4, F_IMU = 8, F_ADM = 16, F_TIPF = 32, F_ODOM = 64, F_DYN = 1u << 31,
                  F_ROT = 1u << 30, // F_ROT (with F_DYN, > 3 DOF): gravity-aligned tips, rotation-constrained IK
                  // (with F_DYN) the paths around rough terrain, each compiled in only where a configuration needs it - all three in one kernel spill:
                  F_ROUGH = 1u << 29,  // rough_terrain_mode: default tips follow the terrain, step-plane targets, external targets / defaults
                  F_TALIGN = 1u << 28, // gravity_aligned_tips with <= 3 DOF legs: PoseController::updateTipAlignPose
                  F_MLEGS = 1u << 27,  // manual leg manipulation / planner mode: ManualRobot records, updateManual, RT_SKIP_MARKED
                  F_TERRAIN = F_ROUGH | F_TALIGN | F_MLEGS };
// Tip rotations are tracked by a specialisation (stepper rotation planes, rotation-constrained applyIK): legs of more than 3 joints
// (gravity-aligned tips / externally requested targets), or 3-joint legs a joint_control updateManual hands their FK tip pose to
// (walk_controller.cpp:677-690) - the latter only exists in the kernels with the manual-leg logic.
template <int NJ, unsigned F>
constexpr bool rot_enabled() { return (F & F_ROT) != 0 && (NJ > 3 || (F & F_MLEGS) != 0); }

// Launch-uniform parameters (staged in LDS).
struct CycleParams {
  double dt;
  int32_t period, swing_period, stance_period, stance_end, swing_start, swing_end, stance_start;
  int32_t remap_old_period;  // != 0 for the ONE cycle that follows an accepted step-frequency change: the period the legs' phases still count in (see cycle_front)
  int32_t swing_iterations;  // walk_controller.cpp:1035-1036
  int32_t stance_iterations; // :1040 with the standard stance period
  double swing_delta_t;      // :1037
  double inv_dt;             // 1 / time_delta
  double dt_over_swing_dt;   // time_delta / swing_delta_t   (:1247, :1268)
  double stance_dt;          // 1 / stance_iterations (standard stance period, :1041)
  double stride_scale;       // (stance_period / period) / frequency   (:940-941)
  double swing_height, swing_width, body_clearance;
  double swing_progress_scaler; // pose_controller.cpp:1103
  int32_t swing_c_count;        // entries of SharedConsts::swing_c in use; 0 = swing period too long for the table
  int32_t swing_c_valid;        // swing iterations (phase - swing_start) < this have a scaled progress within [0, 1]
  int32_t velocity_input_mode;
  int32_t manual_posing, auto_posing, inclination_posing, imu_posing, admittance_control, dynamic_stiffness, use_joint_effort;
  int32_t clamp_joint_positions, clamp_joint_velocities, force_normal_touchdown;
  int32_t tip_force;  // SHC_FEAT_TIP_FORCE
  int32_t debug_skip; // development ablation mask (SHC_DEBUG_SKIP env): 1 pose, 2 limits, 4 stepper, 8 ik, 16 fk
  int32_t odometry;   // SHC_FEAT_ODOMETRY
  int32_t gravity_aligned;       // tip rotations are tracked (> 3 DOF legs with gravity_aligned_tips, or in rough terrain mode, where an externally
                                 // requested target may carry one): LegStepper::updateTipRotation + the rotation-constrained IK (model.cpp:880-900)
  int32_t rough_terrain;         // rough_terrain_mode (generic kernel): default tips follow the terrain, targets meet the step surface
  int32_t tip_align;             // gravity_aligned_tips with <= 3 DOF legs: PoseController::updateTipAlignPose (generic kernel)
  int32_t gravity_target;        // gravity_aligned_tips (> 3 DOF): an UNDEFINED target rotation is re-assigned from Model::estimateGravity (:1197-1205)
  int32_t joint_control;         // leg_manipulation_mode joint_control: updateManual's velocity inputs move the coxa / tibia joints of 3-joint legs (:677-690);
                                 // 2: the robot has such legs (their tip rotations are tracked), 1: it has none (the inputs do nothing)
  double step_depth;             // walk_controller.h:80
  double target_dir[3];          // x axis of the identity tip rotation FromTwoVectors(x, -z) (walk_controller.cpp:37-41)
  double max_translation[3], max_rotation[3], max_translation_velocity, max_rotation_velocity;
  double pid_p, pid_i, pid_d;
  // admittance: 30 RK4 steps of x'' = -F/m - c/m x' - k/m x collapsed into x <- M x + g F
  double adm_m00, adm_m01, adm_m10, adm_m11, adm_g0, adm_g1;
  double force_gain, virtual_stiffness, swing_stiffness_scaler, load_stiffness_scaler;
  // The posing part of a loop (updateStiffness / updateAdmittance, state_controller.cpp:170-180) runs BEFORE runningState's adjustParameter: in the loop that
  // sets a new force gain or swing height it still reads the old one, the rest of that loop the new one.  These two are what the posing part reads (equal to
  // force_gain / swing_height except in that one cycle; virtual_stiffness and the admittance map above are read by the posing part only).
  double pose_force_gain, pose_swing_height;
  // auto pose (pose_controller.cpp:44-106)
  int32_t n_auto_posers, pose_phase_length, pose_sync, auto_pose_reference_leg;
  // ---- tail staged to LDS only by kernels with auto posing (16-byte aligned start)
  alignas(16) int32_t ap_start[kMaxAutoPosers];
  int32_t ap_end[kMaxAutoPosers]; // both already * normaliser
  double ap_amp[kMaxAutoPosers][7];                         // x y z gravity roll pitch yaw
};

template <int L, int NJ>
struct alignas(16) SharedConsts { // staged in LDS; the parameter block comes last so that its auto-pose tail ends the record
  LegConst<NJ> leg[L];
  alignas(16) double limit[9][4]; // per bearing: max linear speed, max angular speed, max linear / angular acceleration
  // smoothStep(swing_progress * scaler) per swing iteration (pose_controller.cpp:1100-1108): the only per-cycle use of the
  // swing progress is this control input, a function of the integer phase alone
  double swing_c[kSwingTable];
  alignas(16) CycleParams P;
};

// SoA planes in HBM.  Leg fields: legd[f * n_slots + slot]; robot fields: robd[f * n_rob_pad + robot].
// Per-leg state fields.  Two consecutive fields share one 16-byte plane (plane p = fields 2p, 2p + 1, stored as a
// double2 per slot) so that every load/store of a wave moves 16 B per lane = 1 KiB per instruction; every group starts on
// an even index and has an even length (3-vectors are padded to 4).
template <int NJ>
struct Fields {
  static constexpr int NJE = (NJ + 1) & ~1;
  static constexpr int Q = 0, QD = NJ, TIP = 2 * NJ, TVEL = TIP + 3, SORG = TVEL + 3, SVEL = SORG + 3, TORG = SVEL + 3,
                       DFLT = TORG + 3, TARG = DFLT + 3, STRD = TARG + 3,
                       CORE_END = STRD + 3,                            // always loaded / stored (2 NJ + 24: even)
                       ADM = CORE_END, ADM_END = ADM + 2,              // admittance state   (admittance_control)
                       TF = ADM_END, TF_END = TF + 4,                  // tip_force_calculated_ filter state (tip_force)
                       FORCE_IN = TF_END, EFFORT_IN = FORCE_IN + 4,    // inputs
                       POSER_TIP = EFFORT_IN + NJE, MODEL_TIP = POSER_TIP + 4, ADM_DELTA = MODEL_TIP + 4, // outputs
                       // tip directions (x axis of LegStepper::origin_tip_pose_ / current_tip_pose_ rotations), gravity-aligned tips only
                       ORG_DIR = ADM_DELTA + 4, CUR_DIR = ORG_DIR + 3,
                       // ... and of LegStepper::target_tip_pose_ (the identity tip rotation, or what an externally requested target assigned) + pad
                       TARG_DIR = CUR_DIR + 3,
                       // Leg::desired_tip_pose_ as the per-leg API holds it between shc_leg_set_desired_tip_pose and shc_leg_apply_ik:
                       // position (3) + "rotation defined" flag, x axis of the rotation (3) + pad.  The fused cycle never touches these.
                       DES_TIP = TARG_DIR + 4, DES_DIR = DES_TIP + 4,
                       // LegPoser sequence state (pose_controller.h:560-590) for stepToPosition / transitionConfiguration:
                       // origin_tip_pose_ position (3) + master_iteration_count_, x axis of its rotation (3) + "!first_iteration_",
                       // origin_configuration_ (NJ, padded).  Only the sequence entry points touch these.
                       SEQ_ORG = DES_DIR + 4, SEQ_DIR = SEQ_ORG + 4, SEQ_Q0 = SEQ_DIR + 4,
                       // Leg::step_plane_pose_.position_ (3) + "defined" (touchdown detection; read by the cycle in rough terrain mode only)
                       STEP_PLANE = SEQ_Q0 + NJE,
                       // Joint::current_position_ (measured, offset removed) as jointStatesCallback stores it: LegState.actual_tip_pose
                       MEAS_Q = STEP_PLANE + 4, COUNT = MEAS_Q + NJE;
  static_assert(CORE_END % 2 == 0 && SORG % 2 == 0 && COUNT % 2 == 0, "field groups must align to 16-byte planes");
};
// element index of field f of slot `slot` in the plane array (n_slots slots per plane)
SHC_HD int64_t leg_field_index(int f, int64_t slot, int64_t n_slots) { return (int64_t(f >> 1) * n_slots + slot) * 2 + (f & 1); }
struct RobotFields {
  // state + inputs that every specialisation touches
  static constexpr int VLIN = 0, VANG = 2, PLANE = 3, PNORM = 6, PLANE_PREV = 9, PNORM_PREV = 12, OWPP = 15, VIN = 22, WIN = 24,
                       CORE_END = 25;
  static constexpr int MPOSE = 25, TVI = 32, RVI = 35, MANUAL_END = 38; // manual posing
  static constexpr int ABSE = 38, VERR = 41, GYRO = 44, IMU_END = 47;   // IMU posing PID state + gyro input
  static constexpr int IMUQ = 47, IMUQ_END = 51;                        // IMU orientation input
  static constexpr int APREV = 51, APREV_END = 55;                      // previous cycle's auto_pose_.rotation_
  static constexpr int CPOSE = 55, CPOSE_END = 62;                      // output: Model::current_pose_
  static constexpr int WPP = 62, WPP_END = 69;                          // walk_plane_pose_ of the current cycle (LDS tile only)
  static constexpr int ODOM = 69, ODOM_END = 73; // WalkController::odometry_ideal_ (odometry feature): x, y, qw, qz (pure yaw)
  // PoseController::tip_align_pose_ / origin_tip_align_pose_ (gravity_aligned_tips with <= 3 DOF legs, generic kernel only)
  static constexpr int INCL = 73, INCL_END = 75; // output: inclination_pose_.position_ x, y (read by poseForLegManipulation; written by the kernels with manual legs only)
  static constexpr int TALIGN = 75, OTALIGN = 82, COUNT = 89;
  static constexpr int I_WORD = 0, I_APOSER = 1, I_POSE_PHASE = 2, I_RESET_MODE = 3, I_COUNT = 4;
};

// Externally requested tip targets / default poses of rough terrain mode (struct ExternalTarget, walk_controller.h:38-46), one
// record per leg slot in its own lazily allocated plane array (DevState::ext, paired planes like legd): only engines that were
// ever given a request carry it.  flags: bit 0 defined_, bit 1 frame_id_ == "odom_ideal".
struct ExtFields {
  static constexpr int T_POSE = 0, T_TRANSFORM = 7, T_CLEARANCE = 14, T_FLAGS = 15, // external_target_
                       D_POSE = 16, D_TRANSFORM = 23, D_FLAGS = 30,                  // external_default_
                       P_POSE = 32, P_TRANSFORM = 39, P_CLEARANCE = 46, P_FLAGS = 47, COUNT = 48; // LegPoser::external_target_ (planner mode)
};

// Manual leg manipulation (WalkController::updateManual, StateController::legStateToggle): per-robot record in a lazily allocated
// array - only engines that ever toggled a leg carry it, and only the F_TERRAIN kernels read it.
enum : int { LS_WALKING = 0, LS_MANUAL = 1, LS_WALKING_TO_MANUAL = -1, LS_MANUAL_TO_WALKING = -2 }; // enum LegState (parameters_and_states.h:87-94)
struct ManualRobot {
  int32_t leg_state[SHC_MAX_LEGS];
  int32_t manual_leg_count, primary_leg, secondary_leg; // StateController::manual_leg_count_, primary / secondary_leg_selection_
  int32_t skip_cycle; // set by the loop-level kernels (legStateToggle, executePlan) for a robot whose loop they have run: the cycle
                      // launch that follows for the robots still walking (RT_SKIP_MARKED) leaves this robot untouched
  double primary_velocity[3], secondary_velocity[3];         // primary / secondary_tip_velocity_input_
  double primary_position[3], secondary_position[3];         // primary / secondary_pose_input_.position_
};

struct DevState {
  double *legd;
  int32_t *legi;
  double *robd;
  int32_t *robi;
  int64_t n_slots, n_rob_pad, n_robots;
  double *ext; // ExtFields planes, nullptr until the first external request
  ManualRobot *manual; // nullptr until a leg is toggled
  const double *span;  // rough terrain mode with a stance span modifier: the legs' layered-workspace planes (SpanTable), else nullptr
};

// LegStepper::calculateStanceSpanChange on the layered workspace of rough terrain mode (walk_controller.cpp:949-980): per leg the
// plane heights and the plane radii at the one bearing (90 or 270 degrees) the leg's sign of the modifier selects, + the signed modifier.
struct SpanTable {
  static constexpr int kPlanes = 14; // WORKSPACE_LAYERS + 4 (hostinit::kMaxWorkspacePlanes)
  static constexpr int kStride = 2 + 2 * kPlanes; // [0] planes in use, [1] modifier (signed for this leg), then (height, radius) pairs
};
SHC_HD double set_precision3(double v) { return round_to_int(v * 1000.0) / 1000.0; } // setPrecision(value, 3) (standard_includes.h:143; pow(10, 3) is exactly 1000)
SHC_HD double stance_span_change_y(const double *table, int leg, double default_shift_z) {
  const double *t = table + leg * SpanTable::kStride;
  const int n = int(t[0]);
  const double target = set_precision3(default_shift_z);
  int lower = -1, upper = -1; // workspace.upper_bound(target) and its predecessor
  for (int k = 0; k < n; ++k) {
    const double h = t[2 + 2 * k];
    if (h > target && (upper < 0 || h < t[2 + 2 * upper])) upper = k;
    if (h <= target && (lower < 0 || h > t[2 + 2 * lower])) lower = k;
  }
  if (lower < 0 || upper < 0) return 0.0;
  const double uh = set_precision3(t[2 + 2 * upper]), lh = set_precision3(t[2 + 2 * lower]);
  const double i = (target - lh) / (uh - lh);
  const double radius = t[3 + 2 * lower] * (1.0 - i) + t[3 + 2 * upper] * i;
  return radius * t[1];
}

#if defined(__HIPCC__)

// Phase fences (-DSHC_FENCE): scheduling barriers between the phases of the cycle.  They bounded live ranges while the
// kernel was register-starved; with MachineLICM off (see engine.py) the cycle fits without them, and the max-ILP
// scheduler overlaps the LDS / division latencies of neighbouring phases (-6 % per launch), so they are off by default.
// Round 5: they are back for the specialisations where the allocator misses 256 registers by ten without them - the feature-exact kernels of 4- and
// 8-legged robots with 4-joint legs (40 B of scratch per lane otherwise; with the fences 232 / 238 VGPRs, none) and the 8 x 5 kernels with the tip-force estimate.  -DSHC_FENCE: everywhere (development).
template <int L, int NJ, unsigned F>
constexpr bool phase_fences() {
#if defined(SHC_FENCE)
  return true;
#else
  constexpr bool exact = (F & (0x80000000u /* F_DYN */ | 0x40000000u /* F_ROT */)) == 0;
  constexpr bool octo_tipf = L == 8 && NJ == 5 && (F & 32u /* F_TIPF */) != 0 && (F & 16u /* F_ADM */) == 0; // 8 x 5 with the tip-force estimate (12 B otherwise; the
  return exact && ((NJ == 4 && (L == 4 || L == 8)) || octo_tipf);                                             // rotation-constrained model half keeps its 20 B: 40 B with fences)
#endif
}
#define SHC_PHASE_FENCE() do { if constexpr (phase_fences<L, NJ, F>()) __builtin_amdgcn_sched_barrier(0); } while (0)
// Development-only phase timestamps (build with -DSHC_RES2_TIMING) of the two-wavefront resident kernel: the leader of workgroup 1 keeps
// the s_memtime stamps of its latest iteration in LDS; shc_engine_resident_end prints them.
#if defined(SHC_RES2_TIMING) && !defined(SHC_RES2_BUSY_ONLY) // (-DSHC_RES2_BUSY_ONLY: only the two stamps per iteration of each role, no phase ticks)
__shared__ long long shc_ticks_lds[64]; // [0, 32): the walker wavefront of pair 0 (thread 0), [32, 64): its model wavefront (thread 128)
__shared__ long long shc_acc_lds[64];   // per phase: clocks since the previous stamp, summed over the steady REAL iterations
#define SHC_TICK(i) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 1 && (threadIdx.x == 0 || threadIdx.x == 128)) shc_ticks_lds[(threadIdx.x >> 7) * 32 + (i)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SHC_TICK(i) do {} while (0)
#endif
// Development-only phase ablation (build with -DSHC_ABLATE, select with the SHC_DEBUG_SKIP environment variable)
#ifdef SHC_ABLATE
#define SHC_DBG(P) ((P).debug_skip)
#else
#define SHC_DBG(P) 0
#endif
// A launch-uniform parameter read from LDS is a vector value to the compiler: branching on it costs an exec-mask region.
// uni() moves it to an SGPR so the branch is a scalar one.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <unsigned F>
struct Feat {
  __device__ __forceinline__ static bool manual(const CycleParams &P) { return (F & F_DYN) ? P.manual_posing != 0 : (F & F_MANUAL) != 0; }
  __device__ __forceinline__ static bool autop(const CycleParams &P) { return (F & F_DYN) ? P.auto_posing != 0 : (F & F_AUTO) != 0; }
  __device__ __forceinline__ static bool incl(const CycleParams &P) { return (F & F_DYN) ? P.inclination_posing != 0 : (F & F_INCL) != 0; }
  __device__ __forceinline__ static bool imu(const CycleParams &P) { return (F & F_DYN) ? P.imu_posing != 0 : (F & F_IMU) != 0; }
  __device__ __forceinline__ static bool adm(const CycleParams &P) { return (F & F_DYN) ? P.admittance_control != 0 : (F & F_ADM) != 0; }
  __device__ __forceinline__ static bool tipf(const CycleParams &P) { return (F & F_DYN) ? P.tip_force != 0 : (F & F_TIPF) != 0; }
  __device__ __forceinline__ static bool odom(const CycleParams &P) { return (F & F_DYN) ? P.odometry != 0 : (F & F_ODOM) != 0; }
};

template <int L>
struct Group {
  int base; // lane of leg 0
  __device__ __forceinline__ int get(int v, int j) const { return __shfl(v, base + j, 64); }
  __device__ __forceinline__ double get(double v, int j) const { return __shfl(v, base + j, 64); }
  __device__ __forceinline__ double sum(double v) const {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < L; ++j) s += get(v, j);
    return s;
  }
};

// The wave's robot tile in LDS: value of field f for this lane's robot is d[f * RPW + grp].
// Robot-level quantities are computed redundantly by the L lanes of a group from the same tile values and the same
// shuffles, so every lane of a group holds the same bits: put() is an unconditional store (L lanes write one value to
// one address, no exec-mask branch per store; mirror lanes rewrite their live twin's value).
template <int RPW>
struct RobTile {
  double *d;
  int32_t *i;
  int grp;
  __device__ __forceinline__ double get(int f) const { return d[f * RPW + grp]; }
  __device__ __forceinline__ V3 get3(int f) const { return V3{d[f * RPW + grp], d[(f + 1) * RPW + grp], d[(f + 2) * RPW + grp]}; }
  __device__ __forceinline__ Quat getq(int f) const {
    return Quat{d[f * RPW + grp], d[(f + 1) * RPW + grp], d[(f + 2) * RPW + grp], d[(f + 3) * RPW + grp]};
  }
  __device__ __forceinline__ Pose getpose(int f) const { return Pose{get3(f), getq(f + 3)}; }
  __device__ __forceinline__ void put(int f, double v) const { d[f * RPW + grp] = v; }
  __device__ __forceinline__ void put3(int f, V3 v) const {
    d[f * RPW + grp] = v.x;
    d[(f + 1) * RPW + grp] = v.y;
    d[(f + 2) * RPW + grp] = v.z;
  }
  __device__ __forceinline__ void putq(int f, Quat q) const {
    d[f * RPW + grp] = q.w;
    d[(f + 1) * RPW + grp] = q.x;
    d[(f + 2) * RPW + grp] = q.y;
    d[(f + 3) * RPW + grp] = q.z;
  }
  __device__ __forceinline__ void putpose(int f, const Pose &p) const {
    put3(f, p.p);
    putq(f + 3, p.r);
  }
  __device__ __forceinline__ int geti(int f) const { return i[f * RPW + grp]; }
  __device__ __forceinline__ void puti(int f, int v) const { i[f * RPW + grp] = v; }
};

// Per-leg state held in registers across the cycles of one launch.
template <int NJ>
struct LegRegs {
  // What the next cycle's IK step needs of the FK at q (Leg::solveIK reads the joint transforms the previous applyFK left):
  // short chains keep the linear Jacobian columns + tip position themselves (4 vectors for 3 joints), longer ones the
  // sin / cos of the joint angles and rebuild the chain from them (fewer registers, one more chain product per cycle).
  static constexpr bool kKeepJacobian = NJ <= 3;
  double q[NJ], qd[NJ];
  double sn[NJ], cs[NJ];
  V3 lin[NJ], pe;
  V3 tip, tvel, targ, strd;
  double adm0, adm1;
  double stiff; // Leg::virtual_stiffness_ (published only; admittance feature)
  V3 tf, tipx; // tip x axis (robot frame) of the current FK: Leg::setAdmittanceDelta, origin of the tip-rotation blend
  V3 org_dir, cur_dir, targ_dir; // tip directions of LegStepper::origin_tip_pose_ / current_tip_pose_ / target_tip_pose_ (tip rotations tracked)
  int word;
};

// Stepper state that only the swing / stance branch touches is parked in a per-lane LDS strip between uses
// (swing origin position / velocity, stance origin, default tip): 12 doubles = 24 VGPRs off the persistent set.
enum : int { PK_SORG = 0, PK_SVEL = 3, PK_TORG = 6, PK_DFLT = 9, PK_COUNT = 12 };
#ifndef SHC_PARK_REGS
struct Park {
  double *d; // d[f * 64 + lane]
  int lane;
  __device__ __forceinline__ V3 get3(int f) const { return V3{d[f * 64 + lane], d[(f + 1) * 64 + lane], d[(f + 2) * 64 + lane]}; }
  __device__ __forceinline__ void put3(int f, V3 v) const {
    d[f * 64 + lane] = v.x;
    d[(f + 1) * 64 + lane] = v.y;
    d[(f + 2) * 64 + lane] = v.z;
  }
  __device__ __forceinline__ double at(int k) const { return d[k * 64 + lane]; }
  __device__ __forceinline__ void set(int k, double v) const { d[k * 64 + lane] = v; }
};
#else // development variant (scripts/build_variant.py ... -- -DSHC_PARK_REGS): the strip in registers (AGPR copies at one wave per SIMD) instead of LDS
struct Park {
  double *d;
  int lane;
  mutable double r[12];
  __device__ __forceinline__ V3 get3(int f) const { return V3{r[f], r[f + 1], r[f + 2]}; }
  __device__ __forceinline__ void put3(int f, V3 v) const { r[f] = v.x, r[f + 1] = v.y, r[f + 2] = v.z; }
  __device__ __forceinline__ double at(int k) const { return r[k]; }
  __device__ __forceinline__ void set(int k, double v) const { r[k] = v; }
};
#endif

// LegStepper::updateDefaultTipPosition with external_default_.defined_ (walk_controller.cpp:988-990): the requested stance pose,
// moved with the robot since the request (pose_.removePose(transform_), pose.h:178-184), replaces the terrain-following default.
__device__ __forceinline__ V3 external_default(const double *ext, int64_t ns, uint32_t slot, V3 terrain_following) {
  if (ext == nullptr) return terrain_following;
  const double2 *X = reinterpret_cast<const double2 *>(ext);
  if ((int(X[(ExtFields::D_FLAGS / 2) * ns + slot].x) & 1) == 0) return terrain_following;
  double v[14];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const double2 d = X[(ExtFields::D_POSE / 2 + k) * ns + slot];
    v[2 * k] = d.x, v[2 * k + 1] = d.y;
  }
  return V3{v[0], v[1], v[2]} + rotate(Quat{v[3], v[4], v[5], v[6]}, -V3{v[7], v[8], v[9]});
}

// Per-cycle outputs (LegState topic fields); only the last cycle of a launch is written to HBM.
struct LegOut {
  V3 poser_tip, model_tip, adm_delta;
};

// progress value as the reference's LegStepper holds it (walk_controller.cpp:878-896)
__device__ __forceinline__ double swing_progress_of(int word, const CycleParams &P) {
  int pm = (word >> LW_PM_SHIFT) & 3;
  if (pm != PM_SWING) return -1.0;
  int phase = (word >> LW_PHASE_SHIFT) & LW_PHASE_MASK;
  return clampd(double(phase - P.swing_start + 1) / double(P.swing_end - P.swing_start), 0.0, 1.0);
}

// WalkController::getLimit's bracket lookup (walk_controller.cpp:414-436) without the atan2.
// Reference: bearing = mod(roundToInt(degrees(atan2(y, x))), 360); upper = first multiple of 45 >= bearing; the
// interpolation factor is an int / int division (1 iff bearing == upper, else 0), so the limit used is the map entry
// at floor(bearing / 45) * 45.  floor(round(deg) / 45) only changes where deg crosses 45 k - 0.5, i.e. the entry is the
// 45-degree sector of the direction rotated by +0.5 degrees; that sector follows from the signs of the rotated
// components and one magnitude comparison.  Both forms can only disagree for a direction within one rounding error
// of a sector edge (as the device atan2 and glibc's already could).
__device__ __forceinline__ int bearing_bracket(double y, double x) {
  if (x == 0.0 && y == 0.0) return 0; // atan2(0, 0) = 0
  constexpr double kC = 0.99996192306417128874, kS = 0.0087265354983739347; // cos / sin of 0.5 degrees
  double xr = kC * x - kS * y, yr = kS * x + kC * y;
  double ax = fabs(xr), ay = fabs(yr);
  bool upper = yr > 0.0 || (yr == 0.0 && xr > 0.0);
  if (upper) return xr > 0.0 ? (ay < ax ? 0 : 1) : (ay > ax ? 2 : 3);
  return xr < 0.0 ? (ay < ax ? 4 : 5) : (ay > ax ? 6 : 7);
}

// Where a cycle reads its per-leg inputs from (tip_force_measured_, Joint::current_effort_).
// LegInPlanes: the engine's own state planes - one launch = n cycles with the inputs held (L2-resident across the cycles).
template <int NJ>
struct LegInPlanes {
  const double *legd;
  int64_t ns;
  uint32_t slot;
  __device__ __forceinline__ V3 force() const {
    const double2 f01 = reinterpret_cast<const double2 *>(legd)[(Fields<NJ>::FORCE_IN / 2) * ns + slot];
    const double2 f2_ = reinterpret_cast<const double2 *>(legd)[(Fields<NJ>::FORCE_IN / 2 + 1) * ns + slot];
    return V3{f01.x, f01.y, f2_.x};
  }
  __device__ __forceinline__ void effort(double (&e)[NJ]) const {
#pragma unroll
    for (int i = 0; i < NJ; i += 2) {
      const double2 e2 = reinterpret_cast<const double2 *>(legd)[((Fields<NJ>::EFFORT_IN + i) / 2) * ns + slot];
      e[i] = e2.x;
      if (i + 1 < NJ) e[i + 1] = e2.y;
    }
  }
};
// A hook the resident kernel runs between PoseController::updateStance and Model::updateModel (shc_cycle_kernel.hpp); nothing here.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};

// Launch-uniform switches of the parameter block as scalars.  Read from the LDS copy one by one where they are used, each costs a wavefront
// that runs alone on its SIMD an LDS round trip before it can branch; read together once (per cycle() call, or once per launch by the resident
// loop, which keeps its FrontToBack across cycles) they cost one.
// (Round 4 kept only the model half's: on the walker half the grouping cost more scalar registers than it saved.  With the two wavefronts in loops of
//  their own - half the scalar spills - the walker's two switches belong here too: -2.7 % per resident cycle, profiles/r05_probe_walker_variants.txt.)
struct UniFlags {
  int clamp_joint_velocities, clamp_joint_positions, swing_c_count;
  int velocity_input_mode, force_normal_touchdown;
};
__device__ __forceinline__ UniFlags load_uni_flags(const CycleParams &P) {
  const int c = P.clamp_joint_velocities, d = P.clamp_joint_positions, e = P.swing_c_count;
  const int f = P.velocity_input_mode, h = P.force_normal_touchdown;
  return UniFlags{uni(c), uni(d), uni(e), uni(f), uni(h)};
}

// ------------------------------------------------------------------------------------------------- one control cycle
// What the walker / poser half of a cycle hands to the model half (PoseController::updateStance -> Model::updateModel); the
// positions travel in LegOut (poser_tip, adm_delta).
struct FrontToBack {
  V3 desired_dir;   // x axis of the desired tip rotation (body frame) when rot_def
  bool rot_def;
  int my_leg_state; // LegState of this leg (manual leg manipulation)
  bool plane_prev_changed; // out: the steppers' walk-plane copy (PLANE_PREV / PNORM_PREV) was rewritten this cycle (wave-uniform)
  bool pose_only;   // in: stop after the posing part of the loop (state_controller.cpp:165-181) - a robot that stands while a leg toggle / plan step runs
  V3 odom_vel;      // desired linear (x, y) / angular (z) body velocity of this cycle and whether updateWalk reached its odometry
  bool odom_run;    //   update (cycle_front<..., ODOM_HERE = false>: the caller runs odometry_step elsewhere)
  // joint_control updateManual moved this leg's joints (walk_controller.cpp:677-690): Leg::current_tip_pose_ is still the tip the leg had
  // before (applyFK(false)) - its position (joint-1 frame) and x axis (body frame), which the following applyIK starts from
  bool joint_moved;
  V3 held_pe, held_dir;
  // in / out, kept by a caller that runs many cycles with one FrontToBack (the resident loop): the steppers' walk-plane copies of every
  // robot of this wave equal the walker's plane, so the per-cycle comparison (12 LDS reads) is skipped until a default tip moves again
  bool planes_in_sync = false;
  UniFlags uf; // in: load_uni_flags(C.P)
  // in / out, for a caller that keeps fb across cycles: the bearing bracket of this lane's leg in the previous cycle and the four limits
  // WalkController::getLimit derived from the robot's legs then - they are a function of the legs' brackets alone, and a bracket changes
  // when a stride vector crosses a 45-degree sector edge
  int limit_bracket = -1;
  double limit_value[4] = {0.0, 0.0, 0.0, 0.0};
};

// odometry_ideal_ = odometry_ideal_.addPose(calculateOdometry(time_delta_)) (walk_controller.cpp:643, :783-791).  Nothing in the
// cycle reads it back: a pure accumulator over the desired body velocity.
// (sin, cos) of the half yaw step of the last call: the desired angular velocity of a robot that walks steadily does not change from cycle to
// cycle, so a caller that keeps this across cycles (the resident loop) re-evaluates the pair only when some robot of the wave turns differently
struct OdomCache {
  double ha = 0.0, sh = 0.0, ch = 1.0;
  bool valid = false;
};
// (the accumulator itself - x, y, qw, qz - by reference: the robot tile's copy, or registers of a caller that runs many cycles and
//  writes them back once: the two-wavefront resident loop)
__device__ __forceinline__ void odometry_advance(double &ox, double &oy, double &ow, double &oz, const CycleParams &P, double vx, double vy, double vw,
                                                 OdomCache *cache = nullptr) {
  // Both poses are pure yaw (rotation (w, 0, 0, z), z translation 0), so Pose::addPose reduces to its w / z and x / y
  // terms; the dropped terms are exact zeros, the kept ones are evaluated in the general formula's order.
  double sh, ch;
  const double ha = 0.5 * (vw * P.dt); // Quaterniond(AngleAxisd(w dt, z^)): half of one cycle's yaw, a few milliradians
  if (cache != nullptr && cache->valid && __all(ha == cache->ha)) {
    sh = cache->sh, ch = cache->ch; // (the same function of the same argument)
  } else {
    if (__all(fabs(ha) <= 0.5)) sincos_joint<false>(ha, &sh, &ch);
    else sincos_joint(ha, &sh, &ch);
    if (cache != nullptr) cache->ha = ha, cache->sh = sh, cache->ch = ch, cache->valid = true;
  }
  // (this code is compiled into kernels that keep the accumulator in LDS and into one that keeps it in registers, which must agree bit for
  //  bit: contraction is off and the fused multiply-adds are written out)
  {
#pragma clang fp contract(off)
    const double a = vx * P.dt, b = vy * P.dt;
    double ux = -(oz * b), uy = oz * a; // u x v
    ux = ux + ux;
    uy = uy + uy;
    const double nx = ox + fma(-oz, uy, fma(ux, ow, a)), ny = oy + fma(oz, ux, fma(uy, ow, b));
    const double nw = fma(ow, ch, -(oz * sh)), nz = fma(ow, sh, oz * ch);
    ox = nx, oy = ny, ow = nw, oz = nz;
  }
}
template <int RPW>
__device__ __forceinline__ void odometry_step(const RobTile<RPW> &rb, const CycleParams &P, double vx, double vy, double vw, OdomCache *cache = nullptr) {
  using R = RobotFields;
  double ox = rb.get(R::ODOM), oy = rb.get(R::ODOM + 1), ow = rb.get(R::ODOM + 2), oz = rb.get(R::ODOM + 3);
  odometry_advance(ox, oy, ow, oz, P, vx, vy, vw, cache);
  rb.put(R::ODOM, ox);
  rb.put(R::ODOM + 1, oy);
  rb.put(R::ODOM + 2, ow);
  rb.put(R::ODOM + 3, oz);
}

// The control input of PoseController::updateWalkPlanePose (:1100-1108) from each lane's OWN packed leg word: smoothStep of the scaled swing
// progress of the last leg (in id order) of the robot whose scaled progress lies in [0, 1]; -1 when no leg qualifies.  One ballot and one
// shuffle (table path).  The resident pipeline evaluates it on the walker wavefront, which holds the words, for the pose of the next cycle.
template <int L, int NJ>
__device__ __forceinline__ double walk_plane_control_input(const int own_word, const SharedConsts<L, NJ> &C, const CycleParams &P, const Group<L> g,
                                                           const int swing_c_count_u) {
  double c = 0.0;
  bool sel = false;
  if (swing_c_count_u > 0) {
    // clamping the iteration is the clamp of the progress to [0, 1] (walk_controller.cpp:880)
    const int it_own = min(max(((own_word >> LW_PHASE_SHIFT) & LW_PHASE_MASK) - P.swing_start, 0), P.swing_c_count - 1);
    const bool ok_own = ((own_word >> LW_PM_SHIFT) & 3) == PM_SWING && it_own < P.swing_c_valid;
    const unsigned legs_ok = unsigned((__ballot(ok_own) >> g.base) & ((1ull << L) - 1));
    sel = legs_ok != 0;
    const int last = sel ? 31 - __clz(int(legs_ok)) : 0; // the LAST qualifying leg is the one the reference's loop over the legs ends on
    int it_sel = g.get(it_own, last);
    it_sel = sel ? it_sel : 0;
    if (__any(sel)) c = sel ? C.swing_c[it_sel] : 0.0;
  } else { // swing period too long for the table: each lane evaluates its own leg (one division + smoothStep), the group picks the last valid one
    double c_own = -1.0;
    const double sp = swing_progress_of(own_word, P) * P.swing_progress_scaler;
    if (sp >= 0 && sp <= 1.0) c_own = smooth_step(sp);
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const double cj = g.get(c_own, j);
      if (cj >= 0.0) {
        c = cj;
        sel = true;
      }
    }
  }
  return sel ? c : -1.0;
}

// The same control input in two steps for the resident pipeline: each lane's OWN candidate (its leg's table value / smoothStep, -1 when its leg
// does not qualify) - a table read that nothing on the walker wavefront waits for - and the pick of the last qualifying leg of the group, which the
// model wavefront makes when it reads the candidates.  Same value as walk_plane_control_input (the table holds smoothStep results, >= 0).
template <int L, int NJ>
__device__ __forceinline__ double walk_plane_control_candidate(const int own_word, const SharedConsts<L, NJ> &C, const CycleParams &P, const int swing_c_count_u) {
  if (swing_c_count_u > 0) {
    const int it_own = min(max(((own_word >> LW_PHASE_SHIFT) & LW_PHASE_MASK) - P.swing_start, 0), P.swing_c_count - 1);
    const bool ok_own = ((own_word >> LW_PM_SHIFT) & 3) == PM_SWING && it_own < P.swing_c_valid;
    const double c = C.swing_c[it_own];
    return ok_own ? c : -1.0;
  }
  const double sp = swing_progress_of(own_word, P) * P.swing_progress_scaler;
  return (sp >= 0 && sp <= 1.0) ? smooth_step(sp) : -1.0;
}

// Robot-level transcendental work with its independent pieces spread over the lanes of the robot's group (every lane of a group computes the
// same robot-level values from the same bits, so an L-lane group has L - 1 idle copies of every instruction: where a function evaluates the same
// routine on two or three independent arguments, lanes of different legs take one argument each and the results are shuffled back - the same
// instructions on the same operands, hence the same bits as the one-lane-does-all form in shc_math.hpp, at a third / two thirds of the issue slots).
// `leg` is this lane's position in its group (mirror lanes included), L >= 3.
//
// quat_to_euler(q, false) - Eigen's eulerAngles(2, 1, 0) + the reference's flip fix-up (standard_includes.h:248-291): r0 = atan2(m10, m00) and
// r1 = atan2(-m20, +-c2) are independent once the sign of r0 is known, and that is the sign of m10 (atan2(-0, x) is -0 for x >= +0, -pi otherwise);
// the prediction is checked against the r0 that comes back and the sequential form runs if any lane of the wave disagrees (never observed).
template <int L>
__device__ __forceinline__ V3 quat_to_euler_zyx_grouped(Quat q, const Group<L> g, const int leg) {
  static_assert(L >= 3, "three lanes per robot");
  double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  double m00 = 1.0 - (tyy + tzz), m01 = txy - twz, m02 = txz + twy;
  double m10 = txy + twz, m11 = 1.0 - (txx + tzz), m12 = tyz - twx;
  double m20 = txz - twy, m21 = tyz + twx, m22 = 1.0 - (txx + tyy);
  const double c2 = sqrt(m22 * m22 + m21 * m21);
  const bool neg = m10 < 0.0 || (m10 == 0.0 && __builtin_signbit(m10) && __builtin_signbit(m00));
  const bool second = leg == 1;
  const double t = atan2(second ? -m20 : m10, second ? (neg ? -c2 : c2) : m00);
  double r0 = g.get(t, 0), r1 = g.get(t, 1);
  if (__builtin_expect(__any((r0 < 0.0) != neg), 0)) r1 = atan2(-m20, r0 < 0.0 ? -c2 : c2);
  if (r0 < 0.0) r0 += kPi;
  double s1, c1;
  sincos_joint(r0, &s1, &c1);
  double r2 = atan2(s1 * m02 - c1 * m12, c1 * m11 - s1 * m01);
  if (fabs(r1) > kPi / 2 || fabs(r2) > kPi / 2) {
    r0 -= kPi;
    if (r1 > kPi / 2.0) r1 = -r1 + kPi;
    else if (r1 < kPi / 2.0) r1 = -r1 - kPi;
    if (r2 > kPi / 2.0) r2 -= kPi;
    else if (r2 < kPi / 2.0) r2 += kPi;
  }
  return V3{r2, r1, r0};
}
// euler_to_quat(e, false): the three half-angle sin / cos pairs, one per lane of legs 0 / 1 / 2.
template <int L>
__device__ __forceinline__ Quat euler_to_quat_zyx_grouped(V3 e, const Group<L> g, const int leg) {
  static_assert(L >= 3, "three lanes per robot");
  double sn, cs;
  sincos_joint(0.5 * (leg == 0 ? e.x : (leg == 1 ? e.y : e.z)), &sn, &cs);
  const double sx = g.get(sn, 0), cx = g.get(cs, 0), sy = g.get(sn, 1), cy = g.get(cs, 1), sz = g.get(sn, 2), cz = g.get(cs, 2);
  Quat qx{cx, sx, 0, 0}, qy{cy, 0, sy, 0}, qz{cz, 0, 0, sz};
  return (qz * qy) * qx;
}

// PoseController::updateCurrentPose (pose_controller.cpp:811-859): walk-plane pose, manual / inclination / IMU / auto / tip-align pose
// composed into Model::current_pose_ (returned, and left in the robot tile's CPOSE; the walk-plane pose in WPP).  `lw`: the packed
// words of the robot's legs as the previous cycle's updateWalk left them.  The kernels call it inside cycle_front; the two-wavefront
// resident kernel runs it on the model wavefront for the specialisations without auto posing (POSE_HERE = false there).
// OWN_WORD: `lw` is not filled in; the leg of the group that drives the walk-plane pose is found from each lane's own packed word
// (`own_word`) with one ballot and one shuffle instead of L shuffles (specialisations without auto posing / tip-align pose only).
template <int L, int NJ, unsigned F, bool OWN_WORD = false>
__device__ __forceinline__ Pose cycle_pose(LegRegs<NJ> &s, const SharedConsts<L, NJ> &C, const CycleParams &P, const LegConst<NJ> &lc, const RobTile<64 / L> &rb,
                                           const Group<L> g, const int leg, const int (&lw)[L], int &rword, const int walk_state, unsigned &dirty, const bool manual_live,
                                           Pose &auto_pose, Pose &leg_auto, const V3 plane_prev, const V3 pnorm_prev, const int swing_c_count_u,
                                           const int own_word = 0, Pose *owpp_cache = nullptr, const double *c_given = nullptr) {
  // swing_c_count_u: UniFlags::swing_c_count; owpp_cache: the origin walk-plane pose (RobotFields::OWPP) kept in registers by a caller that runs
  // many cycles (it changes at the end of a swing over uneven ground only; the tile copy is kept current)
  static_assert(!OWN_WORD || ((F & (F_DYN | F_AUTO | F_TALIGN)) == 0), "the other legs' words are needed by auto posing and the tip-align pose");
  // (plane_prev / pnorm_prev: the steppers' copy of the walk plane, RobotFields::PLANE_PREV / PNORM_PREV, as the previous updateWalk left it)
  using R = RobotFields;
  using FT = Feat<F>;
  const V3 UZ{0, 0, 1};
  Pose cp; // Model::current_pose_
  {
    // ---- updateWalkPlanePose (:1092-1130)
    Pose wpp;
    {
      // control input of the last leg (in id order) whose scaled swing progress lies in [0, 1]
      double c = 0.0;
      bool sel = false;
      if constexpr (OWN_WORD) { // from this lane's own word, or handed over ready-made (c_given: walk_plane_control_input's result)
        const double ce = c_given != nullptr ? *c_given : walk_plane_control_input<L, NJ>(own_word, C, P, g, swing_c_count_u);
        sel = ce >= 0.0;
        c = sel ? ce : 0.0;
      } else if (swing_c_count_u > 0) {
        // the legs' words are already in every lane: pick the leg with integer tests, then one table read
        int it_sel = 0;
#pragma unroll
        for (int j = 0; j < L; ++j) {
          // clamping the iteration is the clamp of the progress to [0, 1] (walk_controller.cpp:880)
          const int it = min(max(((lw[j] >> LW_PHASE_SHIFT) & LW_PHASE_MASK) - P.swing_start, 0), P.swing_c_count - 1);
          const bool ok = ((lw[j] >> LW_PM_SHIFT) & 3) == PM_SWING && it < P.swing_c_valid;
          it_sel = ok ? it : it_sel;
          sel = sel || ok;
        }
        if (__any(sel)) c = sel ? C.swing_c[it_sel] : 0.0;
      } else { // each lane evaluates its own leg once (one division + smoothStep), the group picks the last valid one
        double c_own = -1.0;
        {
          double sp = swing_progress_of(s.word, P) * P.swing_progress_scaler;
          if (sp >= 0 && sp <= 1.0) c_own = smooth_step(sp);
        }
#pragma unroll
        for (int j = 0; j < L; ++j) {
          double cj = g.get(c_own, j);
          if (cj >= 0.0) {
            c = cj;
            sel = true;
          }
        }
      }
      V3 wplane = sel ? plane_prev : V3{0, 0, 0};
      V3 wnorm = sel ? pnorm_prev : UZ;
      Pose owpp = owpp_cache != nullptr ? *owpp_cache : rb.getpose(R::OWPP);
      // Flat ground (the walk-plane normal is exactly +z and the body is not tilted): FromTwoVectors(z, z) is exactly the
      // identity and slerp(identity, c, identity) takes Eigen's linear branch, so the result below is bit-identical to the
      // general path at a fraction of its cost.  The general path only runs after a default tip moved off the plane.
      bool flat = wnorm.x == 0.0 && wnorm.y == 0.0 && wnorm.z == 1.0 && owpp.r.w == 1.0 && owpp.r.x == 0.0 && owpp.r.y == 0.0 &&
                  owpp.r.z == 0.0;
      if (__builtin_expect(__all(flat), 1)) {
        double s0 = 1.0 - c;
        V3 npp{0.0, 0.0, P.body_clearance + wplane.z};
        wpp.p = npp * c + owpp.p * s0;
        wpp.r = Quat{s0 * 1.0 + c * 1.0, s0 * 0.0 + c * 0.0, s0 * 0.0 + c * 0.0, s0 * 0.0 + c * 0.0};
      } else {
        Pose np;
        np.r = correct_rotation(from_two_vectors(UZ, wnorm), quat_identity());
        np.p = rotate(np.r, V3{0, 0, P.body_clearance});
        np.p.z += wplane.z;
        wpp = interpolate_pose(owpp, c, np);
      }
      if (c == 1.0) {
        const bool same = wpp.p.x == owpp.p.x && wpp.p.y == owpp.p.y && wpp.p.z == owpp.p.z && wpp.r.w == owpp.r.w && wpp.r.x == owpp.r.x &&
                          wpp.r.y == owpp.r.y && wpp.r.z == owpp.r.z;
        if (__any(!same)) {
          rb.putpose(R::OWPP, wpp);
          if (owpp_cache != nullptr) *owpp_cache = wpp;
          dirty |= DIRTY_WALK_PLANE;
        }
      }
      rb.putpose(R::WPP, wpp);
    }
    cp = wpp; // Identity.addPose(walk_plane_pose_)
    // ---- updateManualPose (:863-1003)
    Quat manual_r = quat_identity();
    if (FT::manual(P) && manual_live) {
      V3 tvi_in = rb.get3(R::TVI), rvi_in = rb.get3(R::RVI);
      int reset_mode = rb.geti(R::I_RESET_MODE);
      bool idle = reset_mode == 0 && tvi_in.x == 0.0 && tvi_in.y == 0.0 && tvi_in.z == 0.0 && rvi_in.x == 0.0 && rvi_in.y == 0.0 &&
                  rvi_in.z == 0.0;
      Pose mpose = rb.getpose(R::MPOSE);
      // With zero inputs and NO_RESET the reference re-derives manual_pose_ from its own Euler angles (a no-op up to
      // rounding); that Euler round trip is skipped here.
      if (__any(!idle)) {
        if (!idle) {
          if (reset_mode == 5) { // IMMEDIATE_ALL_RESET (default_pose_ is identity on this path)
            mpose = pose_identity();
          } else {
            double cpos[3] = {mpose.p.x, mpose.p.y, mpose.p.z};
            V3 ce = quat_to_euler(mpose.r, true);
            double crot[3] = {ce.x, ce.y, ce.z};
            double tvi[3] = {tvi_in.x, tvi_in.y, tvi_in.z}, rvi[3] = {rvi_in.x, rvi_in.y, rvi_in.z};
            bool rt[3] = {false, false, false}, rr[3] = {false, false, false};
            switch (reset_mode) {
              case 1: rt[2] = true; rr[2] = true; break;
              case 2: rt[0] = true; rt[1] = true; break;
              case 3: rr[0] = true; rr[1] = true; break;
              case 4: rt[0] = rt[1] = rt[2] = true; rr[0] = rr[1] = rr[2] = true; break;
              default: break;
            }
            double dpos[3], drot[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              // No contraction here: a component that a reset has driven onto its target must land ON it - cur + ((target - cur) / dt) * dt
              // with separately rounded product and sum, as the reference's x86-64 build computes it - because the reset rewrites the
              // velocity input from the SIGN of what is left (:905-925): an exact 0 leaves the input alone, a 1e-17 residue of a fused
              // multiply-add would turn it into +-1, and that input drives the pose once the reset mode is released.
#pragma clang fp contract(off)
              if (rt[i]) {
                double diff = cpos[i] - 0.0;
                if (diff < 0) tvi[i] = 1.0; else if (diff > 0) tvi[i] = -1.0;
              }
              if (rr[i]) {
                double diff = crot[i] - 0.0;
                if (diff < 0) rvi[i] = 1.0; else if (diff > 0) rvi[i] = -1.0;
              }
              double tv = tvi[i] * P.max_translation_velocity;
              double rv = rvi[i] * P.max_rotation_velocity;
              double dp = cpos[i] + tv * P.dt;
              double dr = crot[i] + rv * P.dt;
              double tl = signd(tv) * P.max_translation[i];
              if (rt[i] && 0.0 < P.max_translation[i] && 0.0 > -P.max_translation[i]) tl = 0.0;
              bool ptv = signd(tv) > 0;
              if ((ptv && dp > tl) || (!ptv && dp < tl)) tv = (tl - cpos[i]) / P.dt;
              double rl = signd(rv) * P.max_rotation[i];
              if (rr[i] && 0.0 < P.max_rotation[i] && 0.0 > -P.max_rotation[i]) rl = 0.0;
              bool prv = signd(rv) > 0;
              if ((prv && dr > rl) || (!prv && dr < rl)) rv = (rl - crot[i]) / P.dt;
              dpos[i] = cpos[i] + tv * P.dt;
              drot[i] = crot[i] + rv * P.dt;
            }
            rb.put3(R::TVI, V3{tvi[0], tvi[1], tvi[2]});
            rb.put3(R::RVI, V3{rvi[0], rvi[1], rvi[2]});
            mpose.p = V3{dpos[0], dpos[1], dpos[2]};
            mpose.r = correct_rotation(euler_to_quat(V3{drot[0], drot[1], drot[2]}, true), quat_identity());
          }
          rb.putpose(R::MPOSE, mpose);
        }
        dirty |= DIRTY_MANUAL;
      }
      manual_r = mpose.r;
      // adding the identity pose returns cp unchanged (x + 0, q * 1): skip it while no robot of the wave is posed
      bool unposed = mpose.p.x == 0.0 && mpose.p.y == 0.0 && mpose.p.z == 0.0 && mpose.r.w == 1.0 && mpose.r.x == 0.0 &&
                     mpose.r.y == 0.0 && mpose.r.z == 0.0;
      if (!__all(unposed)) cp = add_pose(cp, mpose);
    }
    if (FT::incl(P)) { // updateInclinationPose (:1240-1259) reads the auto_pose_ left by the previous cycle
      Quat aprev = FT::autop(P) ? rb.getq(R::APREV) : quat_identity();
      Quat comb = normalized(manual_r * aprev);
      Quat removed = normalized(rb.getq(R::IMUQ) * inverse(comb));
      V3 e = quat_to_euler(removed, false);
      double lon = clampd(-P.body_clearance * tan(e.y), -P.max_translation[0], P.max_translation[0]);
      double lat = clampd(P.body_clearance * tan(e.x), -P.max_translation[1], P.max_translation[1]);
      cp = add_pose(cp, Pose{V3{lon, lat, 0.0}, quat_identity()});
      if ((F & F_MLEGS) != 0) {
        rb.put(R::INCL, lon);
        rb.put(R::INCL + 1, lat);
      }
    }
    if (FT::imu(P)) { // updateIMUPose (:1191-1236)
      Quat cur = correct_rotation(rb.getq(R::IMUQ), quat_identity());
      Quat tgt = correct_rotation(manual_r, quat_identity());
#ifdef SHC_POSE_R5
      Quat err = normalized(cur * inverse(tgt));
#else
      // Quaternion::inverse() = conj / |q|^2.  With |q|^2 == 1.0 exactly - the identity an idle manual pose leaves, above all - every x / 1.0 is x: the four
      // divisions are skipped while that holds for every robot of the wave (the same bits either way).
      const double tgt_n2 = dot(tgt, tgt);
      Quat err = normalized(cur * (__all(tgt_n2 == 1.0) ? conj(tgt) : inverse(tgt)));
#endif
#ifdef SHC_POSE_R5 // (development A/B: every lane evaluates all three axes)
      V3 pe = quat_to_euler(err, false);
#else
      V3 pe = quat_to_euler_zyx_grouped<L>(err, g, leg);
#endif
      pe.z = 0.0;
      V3 abse = rb.get3(R::ABSE) + pe * P.dt;
      V3 verr = (-rb.get3(R::GYRO)) * 0.15 + rb.get3(R::VERR) * (1 - 0.15);
      rb.put3(R::ABSE, abse);
      rb.put3(R::VERR, verr);
      V3 corr = -(verr * P.pid_d + pe * P.pid_p + abse * P.pid_i);
      corr.x = clampd(corr.x, -P.max_rotation[0], P.max_rotation[0]);
      corr.y = clampd(corr.y, -P.max_rotation[1], P.max_rotation[1]);
      // yaw of the target rotation; the identity target (idle manual pose) has yaw +0 exactly
      bool tgt_identity = tgt.w == 1.0 && tgt.x == 0.0 && tgt.y == 0.0 && tgt.z == 0.0;
      corr.z = 0.0;
      if (__any(!tgt_identity)) {
        if (!tgt_identity) corr.z = quat_to_euler(tgt, false).z;
      }
#ifdef SHC_POSE_R5
      Quat ir = correct_rotation(euler_to_quat(corr, false), tgt);
#else
      Quat ir = correct_rotation(euler_to_quat_zyx_grouped<L>(corr, g, leg), tgt);
#endif
      cp = add_pose(cp, Pose{V3{0, 0, 0}, ir});
    } else if (FT::autop(P)) { // updateAutoPose (:1134-1187)
      int ref = lw[P.auto_pose_reference_leg];
      bool zbv = (ref & LW_ZBV) != 0;
      int aps = (rword >> RW_APS_SHIFT) & 3;
      if (walk_state == WS_STARTING || walk_state == WS_MOVING) aps = PS_POSING;
      else if ((zbv && walk_state == WS_STOPPING) || walk_state == WS_STOPPED) aps = PS_STOP_POSING;
      int master_phase;
      if (P.pose_sync) {
        master_phase = (ref >> LW_PHASE_SHIFT) & LW_PHASE_MASK;
        rb.puti(R::I_POSE_PHASE, master_phase); // kept for LegState.auto_pose (shc_engine_read_leg_state_msg); unused otherwise
      } else {
        master_phase = rb.geti(R::I_POSE_PHASE);
        rb.puti(R::I_POSE_PHASE, (master_phase + 1) % P.pose_phase_length);
      }
      int aposer = rb.geti(R::I_APOSER);
      int complete = 0;
      for (int i = 0; i < P.n_auto_posers; ++i) { // AutoPoser::updatePose (:1338-1439)
        int fl = (aposer >> (4 * i)) & 15;
        bool start_check = fl & 1, end1 = fl & 2, end2 = fl & 4, allow = fl & 8;
        int phase = master_phase, sp = P.ap_start[i], ep = P.ap_end[i];
        if (sp > ep) {
          ep += P.pose_phase_length;
          if (phase < sp) phase += P.pose_phase_length;
        }
        start_check = !P.pose_sync || (!start_check && aps == PS_POSING && phase == sp);
        end1 = end1 || (aps == PS_STOP_POSING && phase == sp);
        end2 = end2 || (aps == PS_STOP_POSING && phase == ep && end1);
        if (!allow && start_check) {
          allow = true;
          end1 = end2 = false;
        } else if (allow && P.pose_sync && end1 && end2) {
          allow = false;
          start_check = false;
        }
        aposer = (aposer & ~(15 << (4 * i))) | ((int(start_check) | int(end1) << 1 | int(end2) << 2 | int(allow) << 3) << (4 * i));
        complete += allow ? 0 : 1;
        if (phase >= sp && phase < ep && allow) {
          int iteration = phase - sp + 1, num = ep - sp;
          bool first_half = iteration <= num / 2;
          double delta_t = 1.0 / (num / 2.0);
          int offset = int(first_half ? 0 : num / 2.0);
          double t = (iteration - offset) * delta_t;
          double u = 1.0 - t;
          // nodes {0,0,0,A,A} (first half) or {A,A,0,0,0} (second half): B(t) = A * weight
          double wgt = first_half ? (4.0 * t * t * t * u + t * t * t * t) : (u * u * u * u + 4.0 * t * u * u * u);
          V3 pos;
          if (P.ap_amp[i][3] != 0.0) { // gravity amplitude: Model::estimateGravity (model.cpp:156-165)
            V3 e = quat_to_euler(rb.getq(R::IMUQ), false);
            V3 gv{0, 0, kGravity};
            gv = rotate(angle_axis_y(-e.y), gv);
            gv = rotate(angle_axis_x(-e.x), gv);
            pos = normalized(gv) * (P.ap_amp[i][3] * wgt);
          } else {
            pos = V3{P.ap_amp[i][0] * wgt, P.ap_amp[i][1] * wgt, P.ap_amp[i][2] * wgt};
          }
          V3 rot{P.ap_amp[i][4] * wgt, P.ap_amp[i][5] * wgt, P.ap_amp[i][6] * wgt};
          auto_pose = add_pose(auto_pose, Pose{pos, euler_to_quat(rot, false)});
        }
      }
      rb.puti(R::I_APOSER, aposer);
      if (complete == P.n_auto_posers) aps = PS_POSING_COMPLETE;
      rword = (rword & ~(3 << RW_APS_SHIFT)) | (aps << RW_APS_SHIFT);
      // LegPoser::updateAutoPose for this lane's leg (:1716-1778)
      {
        int sp = lc.neg_start, ep = lc.neg_end, np = master_phase;
        if (sp > ep) {
          ep += P.pose_phase_length;
          if (np < sp) np += P.pose_phase_length;
        }
        int st = s.word & 3;
        bool neg = (s.word & LW_NEG) != 0;
        if (st != SS_FORCE_STANCE && st != SS_FORCE_STOP && np == sp) neg = true;
        if (np < sp || np > ep) neg = false;
        s.word = neg ? (s.word | LW_NEG) : (s.word & ~LW_NEG);
        leg_auto = auto_pose;
        if (neg) {
          int iteration = np - sp + 1, num = ep - sp;
          bool first_half = iteration <= num / 2;
          double ci = 1.0;
          if (lc.neg_ratio > 0.0) {
            if (first_half) ci = fmin(1.0, iteration / (num * lc.neg_ratio));
            else ci = fmin(1.0, (num - iteration) / (num * lc.neg_ratio));
          }
          ci = smooth_step(ci);
          Pose negation = interpolate_pose(pose_identity(), ci, auto_pose);
          leg_auto = remove_pose(auto_pose, negation);
        }
      }
      cp = add_pose(cp, auto_pose);
      if (FT::incl(P)) rb.putq(R::APREV, auto_pose.r);
    }
    // ---- updateTipAlignPose (:1024-1088): the legs are visited in id order and each swinging leg overwrites the pose, reading the
    //      translation its predecessor left - re-simulated identically in every lane from L shuffled tip-to-joint vectors
    if ((F & F_TALIGN) != 0 && uni(P.tip_align)) { // (leg 0 has at most 3 joints, :849; the other legs may be longer: every leg takes part)
      Chain<NJ> ch0;
      chain_from_sincos<NJ>(lc, s.sn, s.cs, ch0); // the last applyFK: tip and last joint in the robot frame
      V3 last_joint = ch0.p[NJ - 1]; // tip->reference_link_->actuating_joint_: the last joint this leg really has (a padded leg's locked joints sit at its tip)
      if constexpr (NJ > 3) {
#pragma unroll
        for (int k = NJ - 2; k >= 2; --k) last_joint = lc.jactive[k + 1] != 0.0 ? last_joint : ch0.p[k];
      }
      const V3 t2j_own = base_rotate(lc, last_joint - ch0.pe);
      Pose ta = rb.getpose(R::TALIGN), ota = rb.getpose(R::OTALIGN);
      const V3 n = pnorm_prev; // leg_stepper->getWalkPlaneNormal(): the copy taken by last cycle's updateStride
      const Quat wrot = from_two_vectors(UZ, n);
#pragma unroll
      for (int j = 0; j < L; ++j) {
        const V3 t2j{g.get(t2j_own.x, j), g.get(t2j_own.y, j), g.get(t2j_own.z, j)};
        const double sp = swing_progress_of(lw[j], P);
        if (sp != -1.0) {
          const double link_length = norm(-t2j);
          V3 a = rotate(wrot, t2j), b = n * link_length;
          const V3 to_alignment = -(a - b * (dot(a, b) / dot(b, b)));
          a = ta.p;
          b = n;
          V3 target = (a - b * (dot(a, b) / dot(b, b))) + to_alignment;
          target.x = clampd(target.x, -P.max_translation[0], P.max_translation[1]); // clamped(value, limit): every upper bound is
          target.y = clampd(target.y, -P.max_translation[1], P.max_translation[1]); // limit[1] (standard_includes.h:134)
          target.z = clampd(target.z, -P.max_translation[2], P.max_translation[1]);
          double c = smooth_step(sp);
          if (sp < 0.5) {
            c = smooth_step(c * 2.0);
            ta = interpolate_pose(ota, c, pose_identity());
          } else {
            c = smooth_step((c - 0.5) * 2.0);
            ta = interpolate_pose(pose_identity(), c, Pose{target, quat_identity()});
          }
          if (sp == 1.0) ota = ta;
        }
      }
      rb.putpose(R::TALIGN, ta);
      rb.putpose(R::OTALIGN, ota);
      cp = add_pose(cp, ta);
    }
    rb.putpose(R::CPOSE, cp);
  }
  return cp;
}

// AdmittanceController::updateAdmittance (admittance_controller.cpp:22-61) + Leg::setAdmittanceDelta (model.h:365-368) of one leg:
// touches only the admittance state, the tip-force estimate and the tip axis of the last FK - state of the model half.
template <int NJ, typename IN>
__device__ __forceinline__ void cycle_admittance(LegRegs<NJ> &s, LegOut &out, const CycleParams &P, const IN &in) {
  const V3 force_in = in.force(); // tip_force_measured_
  V3 f = (P.use_joint_effort ? s.tf : force_in) * P.pose_force_gain;
  double fi[3] = {f.x, f.y, f.z}, d[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double u = fmax(fi[i], 0.0);
    double x0 = P.adm_m00 * s.adm0 + P.adm_m01 * s.adm1 + P.adm_g0 * u;
    double x1 = P.adm_m10 * s.adm0 + P.adm_m11 * s.adm1 + P.adm_g1 * u;
    s.adm0 = x0;
    s.adm1 = x1;
    d[i] = clampd(-x0, -0.2, 0.2); // ADMITTANCE_DEADBAND == 0: delta passes through unchanged
  }
  // Leg::setAdmittanceDelta (model.h:365-368): projection on the tip's x axis (robot frame) of the current FK
  out.adm_delta = projection(V3{d[0], d[1], d[2]}, s.tipx);
}

// The walker / poser half of a cycle: updateCurrentPose, updateStiffness, (ADM_HERE: updateAdmittance,) updateWalk with the
// LegSteppers, updateStance.  Leaves out.poser_tip (and fb) for the model half.
template <int L, int NJ, unsigned F, bool ADM_HERE, typename IN, bool ODOM_HERE = true, bool POSE_HERE = true, typename POSEWAIT = NoHook>
__device__ __forceinline__ void cycle_front(LegRegs<NJ> &s, LegOut &out, const SharedConsts<L, NJ> &C, const RobTile<64 / L> &rb, const Park &pk,
                                            const Group<L> g, int leg, const double *__restrict__ legd, int64_t ns, uint32_t slot, unsigned &dirty,
                                            const bool manual_live, const bool touchdown_detection, double *ext, const ManualRobot *mr, const IN &in,
                                            FrontToBack &fb, const double *span = nullptr, const POSEWAIT &pose_wait = POSEWAIT()) {
  using R = RobotFields;
  using FT = Feat<F>;
  // POSE_HERE = false: PoseController::updateCurrentPose of this cycle runs on another wavefront (cycle_pose); pose_wait() returns once
  // Model::current_pose_ (CPOSE) and the walk-plane pose (WPP) of this cycle are in the robot tile.  Only for specialisations whose
  // pose does not feed back into updateWalk (no auto posing: its pose state gates the STOPPING -> STOPPED transition).
  static_assert(POSE_HERE || ((F & F_DYN) == 0 && (F & F_AUTO) == 0 && (F & F_TERRAIN) == 0), "the pose can run elsewhere only without auto posing / terrain paths");
  // The parameter block and the per-leg records are loop-invariant LDS data: addressed directly, the IR-level LICM hoists
  // every one of their ~90 loads out of the n_cycles loop and pins ~180 VGPRs for the whole launch (measured: 556 B of
  // scratch per lane, 2x the launch time).  An opaque zero in the address keeps each load next to its use.
  int zero = 0;
  asm volatile("" : "+v"(zero));
  const CycleParams &P = (&C.P)[zero];
  const LegConst<NJ> &lc = C.leg[leg + zero];
  const V3 UZ{0, 0, 1};
  // gravity-aligned tips: only legs with more than 3 joints constrain the tip rotation (walk_controller.cpp:37, :1197);
  // its own kernel specialisation (F_ROT), launched when the parameter is set
  constexpr bool rot_on = rot_enabled<NJ, F>();
  constexpr bool rot_walk = rot_on && NJ > 3; // LegStepper::updateTipRotation's "more than 3 joints" branch (:1195)
  bool rot_def = (s.word & LW_ROTDEF) != 0;
  bool targ_rot = (s.word & LW_TARGROT) != 0; // LegStepper::target_tip_pose_.rotation_ defined
  fb.odom_run = false;
  fb.plane_prev_changed = false;
  fb.joint_moved = false;
  SHC_TICK(2);

  int rword = rb.geti(R::I_WORD);
  int walk_state = rword & 3;
  // ---- per-leg predicates the walk FSM needs from the previous cycle's stepper state (walk_controller.cpp:607-611).  Their only
  //      readers are the STOPPING branches (walk FSM :599-619, updateAutoPose :1147-1150), and a robot is STOPPING in this cycle
  //      only if it already was or if it is MOVING without a command (:533-536): skipped while no robot of the wave can be.
  const double vin_x = rb.get(R::VIN), vin_y = rb.get(R::VIN + 1), win = rb.get(R::WIN); // the velocity command of this cycle (read once)
  bool may_stop = walk_state == WS_STOPPING;
  if (walk_state == WS_MOVING) { // (a conservative test of "no command": anything that could round to a zero norm counts)
    const double ax = fabs(vin_x), ay = fabs(vin_y), aw = win;
    may_stop = !(aw != 0.0 || ax > 1e-100 || ay > 1e-100);
  }
  s.word &= ~(LW_ZBV | LW_ATT);
  if (!(SHC_DBG(P) & 128) && __any(may_stop)) {
    int w = s.word;
    if (dot(s.strd, s.strd) == 0.0) w |= LW_ZBV;
    const V3 pnp = rb.get3(R::PNORM_PREV);
    V3 err = s.tip - s.targ;
    // rejection from the exact unit normal (0, 0, 1) is (x, y, z - z): skip the projection's division on flat ground
    if (__all(pnp.x == 0.0 && pnp.y == 0.0 && pnp.z == 1.0)) err.z = 0.0;
    else err = rejection(err, pnp);
    if (dot(err, err) < kTipTolerance * kTipTolerance) w |= LW_ATT;
    s.word = w;
  }
  // The packed words of the robot's legs in every lane (L shuffles): read by the pose (here), the dynamic stiffness and the general walk state
  // machine.  Where none of them needs it early the steady state (every robot of the wave MOVING) does without: see the state machine below.
  int lw[L];
  const bool lw_early = POSE_HERE || uni(FT::adm(P) ? 1 : 0) != 0;
  if (lw_early) {
#pragma unroll
    for (int j = 0; j < L; ++j) lw[j] = g.get(s.word, j);
  } else {
#pragma unroll
    for (int j = 0; j < L; ++j) lw[j] = 0;
  }

  SHC_PHASE_FENCE();
  SHC_TICK(3);
  // Manual leg manipulation: while any leg of the robot is not WALKING, updateWalk returns before it touches velocities, walk
  // state or steppers (walk_controller.cpp:492-505)
  int my_leg_state = LS_WALKING;
  bool frozen = false;
  if ((F & F_MLEGS) != 0 && mr != nullptr) {
    my_leg_state = mr->leg_state[leg];
#pragma unroll
    for (int j = 0; j < L; ++j) frozen = frozen || g.get(my_leg_state, j) != LS_WALKING;
  }

  // =============================================================== PoseController::updateCurrentPose (:811-859)
  Pose cp; // Model::current_pose_
  Pose auto_pose = pose_identity();
  Pose leg_auto = pose_identity();
  if (POSE_HERE && !(SHC_DBG(P) & 1)) {
    cp = cycle_pose<L, NJ, F>(s, C, P, lc, rb, g, leg, lw, rword, walk_state, dirty, manual_live, auto_pose, leg_auto, rb.get3(R::PLANE_PREV), rb.get3(R::PNORM_PREV),
                              fb.uf.swing_c_count);
  } else if (POSE_HERE) {
    cp = rb.getpose(R::CPOSE);
  }
  SHC_PHASE_FENCE();
  SHC_TICK(4);
  int pose_state = (rword >> RW_APS_SHIFT) & 3; // walker_->setPoseState (state_controller.cpp:168)

  // =============================================================== AdmittanceController (:22-134)
  out.adm_delta = V3{0, 0, 0};
  if (FT::adm(P)) {
    // updateStiffness (admittance_controller.cpp:96-134) only feeds the published per-leg virtual_stiffness_
    // (state_controller.cpp:889); updateAdmittance reads the global parameters (:35-37), so nothing of it reaches the
    // joint path.  The reference walks the legs in id order: a SWING leg overwrites its own value and adds its load term
    // to both neighbours, so leg j ends with its base value plus the load terms of those SWING neighbours that are
    // processed after leg j's own overwrite (all of them when leg j is not in SWING), added in id order.
    if (uni(P.dynamic_stiffness) && walk_state != WS_STOPPED) { // state_controller.cpp:175 (walk state before updateWalk)
      const bool swing = (s.word & 3) == SS_SWING;
      const double k = P.virtual_stiffness;
      const double ref = fabs((s.tip.z - pk.get3(PK_DFLT).z) / P.pose_swing_height);
      const double load = swing ? k * (ref * (P.load_stiffness_scaler - 1)) : 0.0;
      double v = swing ? k * (ref * (P.swing_stiffness_scaler - 1) + 1) : k;
      const int lo = leg == 0 ? L - 1 : leg - 1, hi = leg == L - 1 ? 0 : leg + 1;
#pragma unroll
      for (int i = 0; i < L; ++i) {
        const double li = g.get(load, i);
        const bool si = (lw[i] & 3) == SS_SWING;
        if (si && (i == lo || i == hi) && (!swing || i > leg)) v += li;
      }
      s.stiff = v;
    }
    if (ADM_HERE) cycle_admittance<NJ>(s, out, P, in);
  }
  if ((F & F_MLEGS) != 0 && fb.pose_only) { // the loop of this robot goes on in a loop-level kernel (legStateToggle / executePlan)
    rb.puti(R::I_WORD, rword);              // walker_->setPoseState(poser_->getAutoPoseState())
    return;
  }

  SHC_PHASE_FENCE();
  // =============================================================== WalkController::updateWalk (:440-648)
  // ---- getLimit x 4 (:414-436): bracket index per leg, min over the robot's legs
  double lim[4] = {0.05, 0.3, 0.02, 0.1};
  if (!(SHC_DBG(P) & 2)) {
    double sx = vin_x + win * (-s.tip.y), sy = vin_y + win * s.tip.x;
    int idx = bearing_bracket(sy, sx);
    if (__all(idx == fb.limit_bracket)) { // every leg of every robot of the wave in the bracket it was in: the same minima
#pragma unroll
      for (int k = 0; k < 4; ++k) lim[k] = fb.limit_value[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) lim[k] = kUnassigned;
#pragma unroll
      for (int j = 0; j < L; ++j) {
        int ij = g.get(idx, j);
#pragma unroll
        for (int k = 0; k < 4; ++k) lim[k] = fmin(lim[k], C.limit[ij][k]);
      }
      fb.limit_bracket = idx;
#pragma unroll
      for (int k = 0; k < 4; ++k) fb.limit_value[k] = lim[k];
    }
  }
  SHC_PHASE_FENCE();
  SHC_TICK(5);
  double vx = rb.get(R::VLIN), vy = rb.get(R::VLIN + 1), vw = rb.get(R::VANG);
  // |linear input|.  Throttle mode only ever asks "> 1" and "!= 0" of it: while no robot of the wave has n2 > 1 (sqrt(n2) <= 1 then, the
  // correctly rounded square root being monotone with sqrt(1) = 1) a stand-in with the same two answers saves the FP64 square root.
  const double lin_n2 = vin_x * vin_x + vin_y * vin_y;
  double lin_norm;
  const int velocity_input_mode = fb.uf.velocity_input_mode;
  if (velocity_input_mode == 0 && __all(lin_n2 <= 1.0)) lin_norm = lin_n2 != 0.0 ? 0.5 : 0.0;
  else lin_norm = sqrt(lin_n2);
  if (!(SHC_DBG(P) & 32)) {
    double nvx, nvy, nw;
    if (velocity_input_mode == 0) { // throttle (:451-466)
      double k = 1.0; // clamped to the unit disc
      if (__any(lin_norm > 1.0)) k = lin_norm > 1.0 ? 1.0 / lin_norm : 1.0;
      const double cx = lin_norm > 1.0 ? vin_x * k : vin_x, cy = lin_norm > 1.0 ? vin_y * k : vin_y;
      nw = clampd(win, -1.0, 1.0) * lim[1];
      const double sc = 1.0 - fabs(win);
      nvx = (cx * lim[0]) * sc;
      nvy = (cy * lim[0]) * sc;
    } else { // real (:467-481)
      const bool over = lin_norm > lim[0];
      const double k = lim[0] / lin_norm;
      const double cx = over ? vin_x * k : vin_x, cy = over ? vin_y * k : vin_y;
      nw = clampd(win, -lim[1], lim[1]);
      const double sc = lim[1] != 0.0 ? (1.0 - fabs(nw / lim[1])) : 0.0;
      nvx = cx * sc;
      nvy = cy * sc;
    }
    if (walk_state == WS_STOPPING) nvx = nvy = nw = 0.0; // :483-487
    if (frozen) nvx = vx, nvy = vy, nw = vw; // (a robot with a manual leg keeps its desired velocities: zero acceleration below)
    // acceleration-limited approach (:508-527)
    const double ax = nvx - vx, ay = nvy - vy;
    const double an2 = ax * ax + ay * ay;
    const double an = __all(an2 == 0.0) ? 0.0 : sqrt(an2); // (every robot already at its target velocity: sqrt(0) = 0)
    const double cap = lim[2] * P.dt;
    if (__all(an < cap)) { // every robot of the wave reaches its target this cycle (the steady state)
      vx += ax;
      vy += ay;
    } else {
      const double inv = an2 > 0.0 ? an : 1.0; // normalized() leaves the zero vector unchanged
      const double sx_ = (ax / inv) * lim[2] * P.dt, sy_ = (ay / inv) * lim[2] * P.dt;
      vx += an < cap ? ax : sx_;
      vy += an < cap ? ay : sy_;
    }
    const double aa = nw - vw;
    vw += fabs(aa) < lim[3] * P.dt ? aa : signd(aa) * lim[3] * P.dt;
    rb.put(R::VLIN, vx);
    rb.put(R::VLIN + 1, vy);
    rb.put(R::VANG, vw);
  }
  const bool has_cmd = (lin_norm != 0.0) || (win != 0.0);

  SHC_PHASE_FENCE();
  SHC_TICK(6);
  // ---- StateController::adjustParameter("step_frequency") accepted in this loop (state_controller.cpp:491-492, before updateWalk and after the posing
  //      part): WalkController::generateStepCycle() has installed the new step cycle - the constants of this launch - and, for a robot that is MOVING,
  //      LegStepper::updatePhase (walk_controller.cpp:862-867) maps every leg's phase onto it: phase_ = int(step_progress_ * period) with
  //      step_progress_ = phase_ / old period as the last iteratePhase left it (:878), then updateStepState (:901-917).  The posing part above has run
  //      on the un-mapped words, as in the reference.  shc_engine_adjust_parameter sets P.remap_old_period for exactly one cycle and that cycle runs
  //      on the runtime-flag kernels (the feature-exact ones carry none of this).
  if constexpr ((F & F_DYN) != 0) {
    const int remap_from = uni(P.remap_old_period);
    if (remap_from > 0) {
      if (walk_state == WS_MOVING) {
        int ph = (s.word >> LW_PHASE_SHIFT) & LW_PHASE_MASK, st = s.word & 3;
        const double step_progress = double(ph) / double(remap_from);
        ph = int(step_progress * double(P.period));
        if (st != SS_FORCE_STOP) {
          if (ph >= P.swing_start && ph < P.swing_end && st != SS_FORCE_STANCE) st = SS_SWING;
          else if (ph < P.stance_end || ph >= P.stance_start) st = SS_STANCE;
        }
        s.word = (s.word & ~(3 | (LW_PHASE_MASK << LW_PHASE_SHIFT))) | st | (ph << LW_PHASE_SHIFT);
      }
      if (lw_early) { // the other legs' words as the walk state machine below reads them
#pragma unroll
        for (int j = 0; j < L; ++j) lw[j] = g.get(s.word, j);
      }
    }
  }
  // ---- walk state machine (:529-564)
  int lacp = (rword >> RW_LACP_SHIFT) & 15, lcfs = (rword >> RW_LCFS_SHIFT) & 15;
  bool rtda = (rword & RW_RTDA) != 0;
  bool early_return = false;
  int my_state = s.word & 3, my_phase = (s.word >> LW_PHASE_SHIFT) & LW_PHASE_MASK;
  bool my_acp = (s.word & LW_ACP) != 0, my_cfs = (s.word & LW_CFS) != 0;
  bool my_update_default = false;
  bool any_stepping = false;
  // The steady state first, with ONE wave-uniform test: every robot of the wave MOVING with a command.  The general machine below then leaves
  // the walk state, the shared counters and the robot word as they are and only clears at_correct_phase (:594-597) - but reaches that through
  // a chain of per-lane conditions that costs a lone wavefront ~700 clocks a cycle.
  const bool steady = __all(!frozen && walk_state == WS_MOVING && has_cmd) && !(SHC_DBG(P) & 16384);
  if (steady) {
    my_acp = false;
    any_stepping = ((__ballot((s.word & 3) != SS_FORCE_STOP) >> g.base) & ((1ull << L) - 1)) != 0;
  } else if (SHC_DBG(P) & 16384) { // (development ablation: the state machine's work in the steady state - valid while every robot is MOVING)
    my_acp = false;
    any_stepping = true;
  } else if (frozen) {
    early_return = true; // updateWalk returned at :503
  } else if (walk_state == WS_STOPPED && has_cmd) {
    walk_state = WS_STARTING;
    my_acp = false;
    my_cfs = false;
    my_phase = lc.phase_offset;
    my_state = SS_STANCE;
    if (my_phase >= P.swing_start && my_phase < P.swing_end) my_state = SS_SWING; // updateStepState (:901-917)
    early_return = true; // "Skips iteration of phase so auto posing can catch up" (:545)
  } else {
    if (walk_state == WS_STARTING && lacp == L && lcfs == L) {
      lacp = 0;
      lcfs = 0;
      walk_state = WS_MOVING;
    } else if (walk_state == WS_MOVING && !has_cmd) {
      walk_state = WS_STOPPING;
    } else if (walk_state == WS_STOPPING && lacp == L && pose_state == PS_POSING_COMPLETE) {
      lacp = 0;
      walk_state = WS_STOPPED;
    }
    // per-leg bookkeeping in leg-id order: counters are shared and read by later legs in the same cycle (:567-632)
    if (__all(walk_state == WS_MOVING)) {
      // every robot of the wave is MOVING: the loop below reduces to "at_correct_phase = false" (:594-597)
      my_acp = false;
      // any leg of this robot not in FORCE_STOP: one ballot over the wave, this group's bits
      any_stepping = ((__ballot((s.word & 3) != SS_FORCE_STOP) >> g.base) & ((1ull << L) - 1)) != 0;
    } else {
    if (!lw_early) { // (wave-uniform: the shuffles run with every lane active)
#pragma unroll
      for (int j = 0; j < L; ++j) lw[j] = g.get(s.word, j);
    }
#pragma unroll
    for (int j = 0; j < L; ++j) {
      int wj = lw[j];
      int st = wj & 3, ph = (wj >> LW_PHASE_SHIFT) & LW_PHASE_MASK;
      bool acp = (wj & LW_ACP) != 0, cfs = (wj & LW_CFS) != 0, upd = false;
      if (walk_state == WS_STARTING) {
        if (lacp == L) {
          if (ph == P.swing_end && !cfs) {
            cfs = true;
            lcfs++;
          }
        }
        if (!acp) {
          if (C.leg[j].starts_in_swing && ph != P.swing_end) {
            st = SS_FORCE_STANCE;
          } else {
            lacp++;
            acp = true;
          }
        }
      } else if (walk_state == WS_MOVING) {
        acp = false;
      } else if (walk_state == WS_STOPPING) {
        if ((wj & LW_ZBV) && !acp && ph == P.swing_end) {
          if ((wj & LW_ATT) || rtda) {
            rtda = false;
            upd = true;
            st = SS_FORCE_STOP;
            acp = true;
            lacp++;
          } else {
            rtda = true;
          }
        }
      } else { // STOPPED
        st = SS_FORCE_STOP;
        ph = 0;
      }
      any_stepping = any_stepping || (st != SS_FORCE_STOP);
      if (j == leg) {
        my_state = st;
        my_phase = ph;
        my_acp = acp;
        my_cfs = cfs;
        my_update_default = upd;
      }
    }
    }
  }
  if (POSE_HERE || !steady) { // (steady: nothing of the robot word changed; with the pose on this wavefront auto posing may have changed its state bits)
    rword = (rword & ~(3 | (15 << RW_LACP_SHIFT) | (15 << RW_LCFS_SHIFT) | RW_RTDA)) | walk_state | (lacp << RW_LACP_SHIFT) |
            (lcfs << RW_LCFS_SHIFT) | (rtda ? RW_RTDA : 0);
    rb.puti(R::I_WORD, rword);
  }
  SHC_TICK(7);
  int my_pm = (s.word >> LW_PM_SHIFT) & 3;
  SHC_PHASE_FENCE();

  if (!early_return) {
    bool default_changed = false;
    // ---- LegStepper::updateDefaultTipPosition (:984-1014) on the STOPPING -> FORCE_STOP edge (rare)
    if (__any(my_update_default)) {
      if (my_update_default) {
        if (!POSE_HERE) pose_wait();
        Pose wpp = rb.getpose(R::WPP); // leg_->getDefaultBodyPose() == walk_plane_pose_
        // identity + stance span change (single-plane workspace: a constant; layered workspace: from the default tip's current height)
        const double span_y = ((F & F_ROUGH) != 0 && span != nullptr) ? stance_span_change_y(span, leg, pk.get3(PK_DFLT).z) : lc.span_shift;
        V3 idp = transform_vector(wpp, V3{lc.stance_x, lc.stance_y + span_y, 0.0});
        V3 proj = projection(pk.get3(PK_TORG) - idp, rb.get3(R::PNORM_PREV));
        pk.put3(PK_DFLT, external_default(ext, ns, slot, idp + proj));
        default_changed = true;
        dirty |= DIRTY_STANCE_ORG; // the default tip shares the stance-origin planes
      }
    }
    // ---- LegStepper::updateTipPosition (:1018-1189)
    bool standard = (my_state == SS_SWING) || my_cfs;
    int msp = standard ? P.stance_period : lc.first_stance_period;
    int stance_iter = standard ? P.stance_iterations : lc.first_stance_iterations;
    int mss = standard ? P.stance_start : lc.phase_offset;
    double stance_dt = standard ? P.stance_dt : lc.first_stance_dt; // 1 / stance_iterations (:1041)
    (void)stance_iter;
    const V3 dflt = pk.get3(PK_DFLT);
    s.targ = dflt + s.strd * 0.5; // uses last cycle's stride (:1044 precedes updateStride)
    bool stepping = my_state != SS_FORCE_STOP;
    const bool rough = (F & F_ROUGH) != 0 && uni(P.rough_terrain) != 0; // rough terrain mode runs on the F_ROUGH kernels
    bool rough_update_default = false;
    V3 model_tip_prev{0, 0, 0};
    if (rough) { // Leg::current_tip_pose_.position_ as the previous cycle's applyFK left it
      if (LegRegs<NJ>::kKeepJacobian) {
        model_tip_prev = tip_robot_frame(lc, s.pe);
      } else {
        Chain<NJ> ch0;
        chain_from_sincos<NJ>(lc, s.sn, s.cs, ch0);
        model_tip_prev = tip_robot_frame(lc, ch0.pe);
      }
    }
    // Two forms of the same arithmetic.  On a wave of de-phased robots some leg is in each of the three cases (first / second half of the
    // swing, stance) in every cycle, so the wave executes all three exec-masked paths anyway - one after the other, each with its own
    // LDS round trip to the parked origins, and the scheduler cannot interleave across the branches.  The STRAIGHT form evaluates the
    // swing and stance curves for every lane in one basic block (same operations per lane, results selected at the end): the same
    // number of instructions, but three independent dependency chains for the scheduler to overlap - what a wavefront that runs alone on
    // its SIMD (resident mode) is short of.  Rough terrain (ground-contact nodes, step-plane targets, external targets) and
    // force_normal_touchdown keep the branching form.
    const int force_normal_touchdown = fb.uf.force_normal_touchdown;
#ifdef SHC_NO_STRAIGHT // (development: the branching form everywhere)
    const bool straight = false;
#else
    const bool straight = (F & F_ROUGH) == 0 && !(SHC_DBG(P) & 4) && force_normal_touchdown == 0;
#endif
    if (straight) {
      const V3 sorg_p = pk.get3(PK_SORG), svel_p = pk.get3(PK_SVEL), torg_p = pk.get3(PK_TORG);
      const bool swing = my_state == SS_SWING;
      const bool step_sw = stepping && swing, step_st = stepping && !swing;
      // updateStride (:921-945)
      const V3 sv{vx - vw * s.tip.y, vy + vw * s.tip.x, 0.0};
      const V3 strd_new = scaled(sv, P.stride_scale);
      s.strd = sel3(stepping, strd_new, s.strd);
      V3 pn = rb.get3(R::PNORM);
      const bool flat_n = pn.x == 0.0 && pn.y == 0.0 && pn.z == 1.0;
      if (!__all(flat_n)) pn = normalized(pn);
      const V3 clearance = scaled(pn, P.swing_height);
      // swing (:1110-1157)
      const int it_sw = my_phase - P.swing_start + 1;
      const bool first_half = it_sw <= P.swing_iterations / 2;
      const bool first_sw = step_sw && it_sw == 1;
      const V3 sorg = sel3(first_sw, s.tip, sorg_p), svel = sel3(first_sw, s.tvel, svel_p);
      pk.put3(PK_SORG, sorg); // (unchanged unless this is the first iteration of a swing: an unconditional LDS store costs less than the branch)
      pk.put3(PK_SVEL, svel);
      dirty |= first_sw ? unsigned(DIRTY_SWING_ORG) : 0u;
      V3 mid{(sorg.x + s.targ.x) / 2.0, (sorg.y + s.targ.y) / 2.0, fmax(sorg.z, s.targ.z)};
      mid = mid + clearance;
      mid.y += (lc.stance_y > 0.0) ? P.swing_width : -P.swing_width;
      const V3 sep1 = scaled(svel * 0.25, P.dt_over_swing_dt);
      const V3 n1_0 = sorg, n1_1 = sorg + sep1, n1_2 = sorg + scaled(sep1, 2.0);
      const V3 n1_3{(mid.x + n1_2.x) / 2.0, (mid.y + n1_2.y) / 2.0, mid.z};
      const V3 n1_4 = mid;
      const V3 fv = scaled(-s.strd, stance_dt * P.inv_dt);
      const V3 sep2 = scaled(fv * 0.25, P.dt_over_swing_dt);
      const V3 n2_0 = n1_4, n2_1 = n1_4 - (n1_3 - n1_4), n2_2 = s.targ - scaled(sep2, 2.0), n2_3 = s.targ - sep2, n2_4 = s.targ;
      const double t_sw = first_half ? P.swing_delta_t * it_sw : P.swing_delta_t * (it_sw - P.swing_iterations / 2);
      const V3 b0 = sel3(first_half, n1_0, n2_0), b1 = sel3(first_half, n1_1, n2_1), b2 = sel3(first_half, n1_2, n2_2), b3 = sel3(first_half, n1_3, n2_3),
               b4 = sel3(first_half, n1_4, n2_4);
      const V3 dpos_sw = scaled(quartic_bezier_dot(b0, b1, b2, b3, b4, t_sw), P.swing_delta_t);
      // stance (:1159-1177)
      int it_st = my_phase + (P.period - mss); // both terms lie in [0, period]: one conditional subtract is the modulo
      if (it_st >= P.period) it_st -= P.period;
      if (it_st < 0 || it_st >= P.period) it_st = mod_i(it_st, P.period); // (... except while a step-frequency change waits: the phase offsets are the NEW cycle's then, see below)
      it_st += 1;
      const bool first_st = step_st && it_st == 1;
      const V3 torg = sel3(first_st, s.tip, torg_p);
      pk.put3(PK_TORG, torg);
      dirty |= first_st ? unsigned(DIRTY_STANCE_ORG) : 0u;
      const double stride_scaler = standard ? 1.0 : lc.first_stride_scaler; // modified / standard stance period (:1167)
      const V3 sep = scaled((-s.strd) * stride_scaler, 0.25);
      const double t_st = it_st * stance_dt;
      const V3 dpos_st = scaled(quartic_bezier_dot(torg, torg + sep, torg + scaled(sep, 2.0), torg + scaled(sep, 3.0), torg + scaled(sep, 4.0), t_st), stance_dt);
      const V3 dpos = sel3(swing, dpos_sw, dpos_st);
      const V3 tip_new = s.tip + dpos, tvel_new = dpos * P.inv_dt; // delta_pos / time_delta (:1135, :1176)
      s.tip = sel3(stepping, tip_new, s.tip);
      s.tvel = sel3(stepping, tvel_new, s.tvel);
    } else if (stepping && !(SHC_DBG(P) & 4)) {
      // updateStride (:921-945)
      V3 sv{vx - vw * s.tip.y, vy + vw * s.tip.x, 0.0}; // v + w z^ x (tip rejected from z^)
      s.strd = scaled(sv, P.stride_scale);
      V3 pn = rb.get3(R::PNORM);
      // normalized() of the exact unit vector (0,0,1) is itself: skip the sqrt + 3 divisions on flat ground (bit-identical)
      bool flat_n = pn.x == 0.0 && pn.y == 0.0 && pn.z == 1.0;
      if (!__all(flat_n)) pn = normalized(pn);
      V3 clearance = scaled(pn, P.swing_height);
      V3 dpos;
      if (my_state == SS_SWING) {
        int iteration = my_phase - P.swing_start + 1;
        bool first_half = iteration <= P.swing_iterations / 2;
        V3 sorg, svel;
        if (iteration == 1) {
          sorg = s.tip;
          svel = s.tvel;
          pk.put3(PK_SORG, sorg);
          pk.put3(PK_SVEL, svel);
          dirty |= DIRTY_SWING_ORG;
          rough_update_default = rough; // walk_controller.cpp:1058-1061
        } else {
          sorg = pk.get3(PK_SORG);
          svel = pk.get3(PK_SVEL);
        }
        bool ground_contact = false;
        if (rough) { // update the default target to meet the step surface, proactively or reactively (:1081-1101)
          const double2 sp01 = reinterpret_cast<const double2 *>(legd)[(Fields<NJ>::STEP_PLANE / 2) * ns + slot];
          const double2 sp23 = reinterpret_cast<const double2 *>(legd)[(Fields<NJ>::STEP_PLANE / 2 + 1) * ns + slot];
          ground_contact = sp23.y != 0.0; // leg_->getStepPlanePose() != Pose::Undefined() (:1110)
          bool external = false;
          if (ext != nullptr) { // externally requested target (:1068-1079): ExtFields record of this leg, see shc_engine.hip
            const double2 *X = reinterpret_cast<const double2 *>(ext);
            const double2 cf = X[(ExtFields::T_CLEARANCE / 2) * ns + slot];
            const int flags = int(cf.y);
            external = (flags & 1) != 0;
            if (external) {
              double v[14];
#pragma unroll
              for (int k = 0; k < 7; ++k) {
                const double2 d = X[(ExtFields::T_POSE / 2 + k) * ns + slot];
                v[2 * k] = d.x, v[2 * k + 1] = d.y;
              }
              // target_tip_pose_ = pose_.removePose(transform_): position = pose_.transformVector(-transform_.position_) (pose.h:178-184)
              s.targ = V3{v[0], v[1], v[2]} + rotate(Quat{v[3], v[4], v[5], v[6]}, -V3{v[7], v[8], v[9]});
              if (rot_walk) { // ... and its rotation: pose_.rotation_ * transform_.rotation_^-1; a product with UNDEFINED_ROTATION (zeros) stays undefined
                const Quat tr = Quat{v[3], v[4], v[5], v[6]} * inverse(Quat{v[10], v[11], v[12], v[13]});
                targ_rot = !(tr.w == 0.0 && tr.x == 0.0 && tr.y == 0.0 && tr.z == 0.0);
                if (targ_rot) s.targ_dir = rotate(tr, V3{1, 0, 0}), dirty |= DIRTY_TARG_DIR;
              }
              clearance = normalized(clearance) * cf.x;
              if (flags & 2) { // "odom_ideal" frame: lead by calculateOdometry(time_to_swing_end).position_ (:1073-1078, :783-791)
                const double time_to_swing_end = (P.swing_iterations - iteration) * P.dt;
                s.targ = s.targ - V3{vx, vy, 0.0} * time_to_swing_end;
              }
            }
          }
          if (external) {
          } else if (touchdown_detection) {
            if (ground_contact) {
              const V3 step_plane_position = V3{sp01.x, sp01.y, sp23.x} - model_tip_prev; // relative to Leg::current_tip_pose_ (last FK)
              const V3 difference = (s.tip + step_plane_position) - s.targ;
              s.targ = s.targ + projection(difference, rb.get3(R::PNORM));
            } else {
              s.targ.z -= P.step_depth;
            }
          }
        }
        // generatePrimarySwingControlNodes (:1238-1261)
        V3 mid{(sorg.x + s.targ.x) / 2.0, (sorg.y + s.targ.y) / 2.0, fmax(sorg.z, s.targ.z)};
        mid = mid + clearance;
        mid.y += (lc.stance_y > 0.0) ? P.swing_width : -P.swing_width;
        V3 sep1 = scaled(svel * 0.25, P.dt_over_swing_dt);
        V3 n1_0 = sorg, n1_1 = sorg + sep1, n1_2 = sorg + scaled(sep1, 2.0);
        V3 n1_3{(mid.x + n1_2.x) / 2.0, (mid.y + n1_2.y) / 2.0, mid.z};
        V3 n1_4 = mid;
        // generateSecondarySwingControlNodes (:1265-1291)
        V3 fv = scaled(-s.strd, stance_dt * P.inv_dt);
        V3 sep2 = scaled(fv * 0.25, P.dt_over_swing_dt);
        V3 n2_0 = n1_4, n2_1 = n1_4 - (n1_3 - n1_4), n2_2 = s.targ - scaled(sep2, 2.0), n2_3 = s.targ - sep2, n2_4 = s.targ;
        if (rough && ground_contact && !first_half) { // ground contact in the second half: stance-like nodes from the current tip (:1286-1290)
          n2_0 = s.tip, n2_1 = s.tip + sep2, n2_2 = s.tip + scaled(sep2, 2.0), n2_3 = s.tip + scaled(sep2, 3.0), n2_4 = s.tip + scaled(sep2, 4.0);
        }
        if (force_normal_touchdown && !(rough && ground_contact)) { // forceNormalTouchdown (:1314-1329), unless in ground contact (:1114)
          V3 bo = s.targ - scaled(sep2, 4.0);
          bo.z = fmax(sorg.z, s.targ.z);
          bo = bo + clearance;
          n1_4 = bo;
          n2_0 = bo;
          n2_2 = s.targ - scaled(sep2, 2.0);
          V3 half = scaled(n2_2 - bo, 0.5);
          n1_3 = n2_0 - half;
          n2_1 = n2_0 + half;
        }
        if (first_half) {
          double t = P.swing_delta_t * iteration;
          dpos = scaled(quartic_bezier_dot(n1_0, n1_1, n1_2, n1_3, n1_4, t), P.swing_delta_t);
        } else {
          double t = P.swing_delta_t * (iteration - P.swing_iterations / 2);
          dpos = scaled(quartic_bezier_dot(n2_0, n2_1, n2_2, n2_3, n2_4, t), P.swing_delta_t);
        }
      } else { // STANCE / FORCE_STANCE
        int iteration = my_phase + (P.period - mss); // both terms lie in [0, period]: one conditional subtract is the modulo
        if (iteration >= P.period) iteration -= P.period;
        if (iteration < 0 || iteration >= P.period) iteration = mod_i(iteration, P.period);
        iteration += 1;
        V3 torg;
        if (iteration == 1) {
          torg = s.tip;
          pk.put3(PK_TORG, torg);
          dirty |= DIRTY_STANCE_ORG;
          rough_update_default = rough; // :1160-1163
          if (ext != nullptr) { // external_target_.defined_ = false (:1159)
            double &flags = ext[((ExtFields::T_CLEARANCE / 2) * ns + slot) * 2 + 1];
            flags = double(int(flags) & ~1);
          }
        } else {
          torg = pk.get3(PK_TORG);
        }
        double stride_scaler = standard ? 1.0 : lc.first_stride_scaler; // modified / standard stance period (:1167)
        (void)msp;
        V3 sep = scaled((-s.strd) * stride_scaler, 0.25);
        double t = iteration * stance_dt;
        // five collinear equispaced nodes (:1295-1310)
        dpos = scaled(quartic_bezier_dot(torg, torg + sep, torg + scaled(sep, 2.0), torg + scaled(sep, 3.0), torg + scaled(sep, 4.0), t), stance_dt);
      }
      s.tip = s.tip + dpos;
      s.tvel = dpos * P.inv_dt; // delta_pos / time_delta (:1135, :1176)
      if (rough && __any(rough_update_default)) { // LegStepper::updateDefaultTipPosition at the start of a swing / stance period
        if (rough_update_default) { // (the stepper's walk-plane copy was refreshed by updateStride just before: the current plane)
          if (!POSE_HERE) pose_wait();
          Pose wpp = rb.getpose(R::WPP);
          const double span_y = span != nullptr ? stance_span_change_y(span, leg, pk.get3(PK_DFLT).z) : 0.0; // calculateStanceSpanChange (:996-997)
          V3 idp = transform_vector(wpp, V3{lc.stance_x, lc.stance_y + span_y, 0.0});
          V3 proj = projection(pk.get3(PK_TORG) - idp, rb.get3(R::PNORM));
          V3 new_default = external_default(ext, ns, slot, idp + proj);
          pk.put3(PK_DFLT, new_default);
          default_changed = true;
          dirty |= DIRTY_STANCE_ORG;
        }
      }
    }
    SHC_TICK(16);
    // ---- updateTipRotation (:1193-1234).  Without gravity-aligned tips every tip rotation stays UNDEFINED.  With them the
    //      target is the constant identity rotation (x axis along -z); only the x axes of the rotations are ever used
    //      downstream (poser, applyIK), so the state is kept as directions + a "defined" bit.
    if (rot_on) {
      const int pm0 = (s.word >> LW_PM_SHIFT) & 3; // swing / stance progress as the previous iteratePhase left them
      const double sp = swing_progress_of(s.word, P);
      bool more_than_3_joints = NJ > 3; // leg_->getJointCount() > 3 (:1195) of THIS leg: a shorter leg padded up to the kernel's NJ has none of it
      if constexpr (NJ > 3) more_than_3_joints = lc.jactive[3] != 0.0;
      if (rot_walk && more_than_3_joints && (pm0 == PM_STANCE || pm0 == PM_STOP || sp >= 0.5)) {
        if (uni(P.gravity_target) && !targ_rot) { // "set target tip rotation to align with gravity if ... currently undefined" (:1197-1205)
          V3 gv{0, 0, kGravity}; // Model::estimateGravity (model.cpp:156-165); the direction of FromTwoVectors(UnitX, gravity) * UnitX
          if (FT::imu(P) || FT::incl(P) || FT::autop(P)) {
            const V3 e = quat_to_euler(rb.getq(R::IMUQ), false);
            gv = rotate(angle_axis_y(-e.y), gv);
            gv = rotate(angle_axis_x(-e.x), gv);
          }
          s.targ_dir = normalized(gv);
          dirty |= DIRTY_TARG_DIR;
          targ_rot = true;
        }
        if (!targ_rot) { // target undefined: so is the current tip rotation (:1208-1211)
          rot_def = false;
        } else {
          s.cur_dir = s.targ_dir; // correctRotation only flips the quaternion's sign
          if (sp >= 0.5) {
            const double c = smooth_step(fmin(1.0, 2.0 * (sp - 0.5)));
            s.cur_dir = normalized(lerp3(s.org_dir, s.targ_dir, c));
          }
          rot_def = true;
        }
      } else {
        s.org_dir = s.tipx; // leg_->getCurrentTipPose().rotation_: the FK tip rotation of the previous cycle
        rot_def = false;
      }
    }
    // ---- iteratePhase (:871-897)
    my_phase = my_phase + 1 == P.period ? 0 : my_phase + 1; // (phase + 1) % period with phase in [0, period)
    // While adjustParameter("step_frequency") waits for the robots to slow down, the legs' phase offsets are already those of the NEW step cycle
    // (generateLimits' setPhaseOffset, walk_controller.cpp:277) and may exceed the period still in force: a robot that starts to walk then takes such
    // an offset as its phase (:542) and the reference's modulo brings it back on the first iteratePhase.  A branch no lane takes otherwise.
    if (my_phase > P.period) my_phase = my_phase % P.period;
    if (my_state != SS_FORCE_STOP) {
      if (my_phase >= P.swing_start && my_phase < P.swing_end && my_state != SS_FORCE_STANCE) my_state = SS_SWING;
      else if (my_phase < P.stance_end || my_phase >= P.stance_start) my_state = SS_STANCE;
    }
    if (my_state == SS_SWING) my_pm = PM_SWING;
    else if (my_state == SS_STANCE) my_pm = PM_STANCE;
    else if (my_state == SS_FORCE_STOP) my_pm = PM_STOP;
    // ---- updateWalkPlane (:748-779): least-squares plane through the default tip positions.  The fit only changes
    //      when a default tip changed, which is a rare event -> recompute under a wave-uniform guard (bit-identical).
    if (!fb.planes_in_sync) { // (wave-uniform; a caller that keeps fb across cycles skips this while nothing has changed)
      const V3 pl = rb.get3(R::PLANE), pn_ = rb.get3(R::PNORM), plp = rb.get3(R::PLANE_PREV), pnp_ = rb.get3(R::PNORM_PREV);
      const bool same = pl.x == plp.x && pl.y == plp.y && pl.z == plp.z && pn_.x == pnp_.x && pn_.y == pnp_.y && pn_.z == pnp_.z;
      if (any_stepping) { // the stepping legs' saved copies (LegStepper::walk_plane_) now hold the pre-update walker plane
        if (__any(!same)) {
          rb.put3(R::PLANE_PREV, pl);
          rb.put3(R::PNORM_PREV, pn_);
          dirty |= DIRTY_WALK_PLANE;
          fb.plane_prev_changed = true;
        }
      }
      fb.planes_in_sync = __all(same || any_stepping); // robots that are not stepping keep their old copies: the comparison stays
    }
    if (__any(default_changed)) {
      fb.planes_in_sync = false;
      dirty |= DIRTY_WALK_PLANE;
      const V3 nd = pk.get3(PK_DFLT);
      double x = nd.x, y = nd.y, z = nd.z;
      double sxx = g.sum(x * x), sxy = g.sum(x * y), sx = g.sum(x), syy = g.sum(y * y), sy = g.sum(y);
      double sxz = g.sum(x * z), syz = g.sum(y * z), sz = g.sum(z);
      // (A^T A)^-1 A^T b with A = [x y 1]
      double a00 = sxx, a01 = sxy, a02 = sx, a11 = syy, a12 = sy, a22 = double(L);
      double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
      double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
      double det = a00 * c00 + a01 * c01 + a02 * c02;
      double pa = (c00 * sxz + c01 * syz + c02 * sz) / det;
      double pb = (c01 * sxz + c11 * syz + c12 * sz) / det;
      double pc = (c02 * sxz + c12 * syz + c22 * sz) / det;
      rb.put3(R::PLANE, V3{pa, pb, pc});
      rb.put3(R::PNORM, normalized(V3{-pa, -pb, 1.0}));
    }
    // ---- odometry_ideal_ = odometry_ideal_.addPose(calculateOdometry(time_delta_)) (:643, :783-791)
    if (FT::odom(P) && !(SHC_DBG(P) & 1024)) {
      if (ODOM_HERE) odometry_step(rb, P, vx, vy, vw);
      fb.odom_vel = V3{vx, vy, vw};
      fb.odom_run = true;
    }
  }
  // =============================================================== WalkController::updateManual x 2 (walk_controller.cpp:652-744)
  if ((F & F_MLEGS) != 0 && mr != nullptr && my_leg_state == LS_MANUAL) {
    const bool primary = leg == mr->primary_leg, secondary = !primary && leg == mr->secondary_leg;
    // (a MANUAL leg that is neither selection reads an uninitialised vector in the reference: its inputs are zero here)
    V3 vin{0, 0, 0}, pin{0, 0, 0};
    if (primary) {
      vin = V3{mr->primary_velocity[0], mr->primary_velocity[1], mr->primary_velocity[2]};
      pin = V3{mr->primary_position[0], mr->primary_position[1], mr->primary_position[2]};
    } else if (secondary) {
      vin = V3{mr->secondary_velocity[0], mr->secondary_velocity[1], mr->secondary_velocity[2]};
      pin = V3{mr->secondary_position[0], mr->secondary_position[1], mr->secondary_position[2]};
    }
    if (norm(vin) != 0.0 && uni(P.joint_control) != 0) {
      // joint_control (:677-690, "works only for 3DOF legs"): the x / y inputs step the tibia / coxa joints, unclamped; the stepper's tip
      // pose becomes the FK tip pose of the moved joints WITH its rotation, so the applyIK that follows is rotation-constrained and
      // starts from the tip pose the leg had before (Leg::applyFK(false) moves the joint transforms only).  Legs of other joint
      // counts ignore the input.
      bool three_joints = NJ == 3; // (a robot whose legs differ in DOF: the padded joints of a shorter leg are inactive)
      if constexpr (NJ > 3) three_joints = lc.jactive[3] == 0.0;
      if constexpr (rot_on) if (three_joints) {
        Chain<NJ> ch;
        chain_from_sincos<NJ>(lc, s.sn, s.cs, ch);
        fb.joint_moved = true;
        fb.held_pe = ch.pe;
        fb.held_dir = s.tipx;
        s.q[0] += vin.y * P.max_rotation_velocity * P.dt;
        s.q[2] += vin.x * P.max_rotation_velocity * P.dt;
        joint_sincos<NJ>(lc, s.q, s.sn, s.cs);
        chain_from_sincos<NJ>(lc, s.sn, s.cs, ch);
        s.tip = tip_robot_frame(lc, ch.pe);
        s.cur_dir = base_rotate(lc, ch.xe);
        rot_def = true;
      }
    } else if (norm(vin) != 0.0) { // tip_control (:690-704)
      // ik_error = desired - current tip of the last updateModel: a leg that could not follow is pushed back towards its tip
      V3 prev_tip;
      if (LegRegs<NJ>::kKeepJacobian) {
        prev_tip = tip_robot_frame(lc, s.pe);
      } else {
        Chain<NJ> ch0;
        chain_from_sincos<NJ>(lc, s.sn, s.cs, ch0);
        prev_tip = tip_robot_frame(lc, ch0.pe);
      }
      const double2 d01 = reinterpret_cast<const double2 *>(legd)[(Fields<NJ>::DES_TIP / 2) * ns + slot];
      const double2 d23 = reinterpret_cast<const double2 *>(legd)[(Fields<NJ>::DES_TIP / 2 + 1) * ns + slot];
      const V3 ik_error = V3{d01.x, d01.y, d23.x} - prev_tip;
      V3 change = vin * (P.max_translation_velocity * P.dt);
      if (norm(ik_error) >= kIkTolerance) change = (-normalized(ik_error)) * norm(change);
      s.tip = s.tip + change;
      rot_def = false; // setCurrentTipPose(Pose(new_tip_position, UNDEFINED_ROTATION)) (:704)
    }
    if (norm(pin) != 0.0 && uni(P.joint_control) == 0) { // tip-pose overload (:712-744; tip_control only): the requested position, rotation undefined
      s.tip = pin;
      rot_def = false;
    }
  }
  s.word = (s.word & ~(3 | LW_ACP | LW_CFS | (3 << LW_PM_SHIFT) | (LW_PHASE_MASK << LW_PHASE_SHIFT) | LW_ZBV | LW_ATT | LW_IKFAIL | LW_ROTDEF | LW_TARGROT)) |
           my_state | (my_acp ? LW_ACP : 0) | (my_cfs ? LW_CFS : 0) | (my_pm << LW_PM_SHIFT) | (my_phase << LW_PHASE_SHIFT) |
           (rot_def ? LW_ROTDEF : 0) | (targ_rot ? LW_TARGROT : 0);

  SHC_PHASE_FENCE();
  SHC_TICK(8);
  // =============================================================== PoseController::updateStance (:110-141)
  V3 desired_dir{1, 0, 0}; // x axis of the desired tip rotation (body frame) when rot_def
  {
    if (!POSE_HERE && !(SHC_DBG(P) & 32768)) {
      pose_wait();
      SHC_TICK(17);
      cp = rb.getpose(R::CPOSE);
    } else if (!POSE_HERE) {
      cp = pose_identity();
    }
    Pose bp = cp;
    if (FT::autop(P) && !FT::imu(P)) {
      bp = remove_pose(bp, auto_pose);
      bp = add_pose(bp, leg_auto);
    }
    out.poser_tip = (SHC_DBG(P) & 256) ? s.tip : inverse_transform_vector(bp, s.tip);
    // no posing for manually manipulated legs (:135-139)
    if ((F & F_MLEGS) != 0 && (my_leg_state == LS_MANUAL || my_leg_state == LS_WALKING_TO_MANUAL)) out.poser_tip = s.tip;
    if (rot_on && rot_def) desired_dir = rotate(inverse(bp.r), s.cur_dir); // pose.rotation^-1 * walker tip rotation (:129-130)
    if (rot_on && rot_def && (F & F_MLEGS) != 0 && (my_leg_state == LS_MANUAL || my_leg_state == LS_WALKING_TO_MANUAL)) desired_dir = s.cur_dir;
  }

  fb.desired_dir = desired_dir;
  fb.rot_def = rot_def;
  fb.my_leg_state = my_leg_state;
  SHC_PHASE_FENCE();
  SHC_TICK(15);
}

// The model half of a cycle: Model::updateModel (model.cpp:142-152) - Leg::setDesiredTipPose, applyIK (one DLS step, joint
// integration and clamps, applyFK, 5 mm check), calculateTipForce.
template <int L, int NJ, unsigned F, typename IN>
__device__ __forceinline__ void cycle_back(LegRegs<NJ> &s, LegOut &out, const SharedConsts<L, NJ> &C, int leg, const double *__restrict__ legd, int64_t ns,
                                           uint32_t slot, const ManualRobot *mr, const IN &in, const FrontToBack &fb) {
  using FT = Feat<F>;
  int zero = 0; // (see cycle_front: keeps the loop-invariant LDS loads next to their uses)
  asm volatile("" : "+v"(zero));
  const CycleParams &P = (&C.P)[zero];
  const LegConst<NJ> &lc = C.leg[leg + zero];
  constexpr bool rot_on = rot_enabled<NJ, F>();
  const V3 desired_dir = fb.desired_dir;
  const bool rot_def = fb.rot_def;
  const int my_leg_state = fb.my_leg_state;
  {
    SHC_TICK(9);
    V3 desired = out.poser_tip + out.adm_delta; // Leg::setDesiredTipPose (:653-663)
    if ((F & F_MLEGS) != 0 && (my_leg_state == LS_MANUAL || my_leg_state == LS_WALKING_TO_MANUAL))
      desired = out.poser_tip; // "Don't apply delta to manually manipulated legs" (:655-656)
    if ((F & F_MLEGS) != 0 && mr != nullptr) { // Leg::desired_tip_pose_.position_ is read back by the next updateManual (:692)
      double *dd = const_cast<double *>(legd);
      reinterpret_cast<double2 *>(dd)[(Fields<NJ>::DES_TIP / 2) * ns + slot] = double2{desired.x, desired.y};
      dd[((Fields<NJ>::DES_TIP / 2 + 1) * ns + slot) * 2] = desired.z;
    }
    Chain<NJ> chain;
    bool retried = false; // the unconstrained retry is a nested applyIK: calculateTipForce then runs twice (:938 in both frames)
    if (rot_on) {
      // Leg::applyIK with a (possibly) defined desired tip rotation (model.cpp:861-941): position solve; if constrained,
      // integrate it without the velocity clamp, FK, solve for the rotation delta between the tip direction the leg had
      // BEFORE this call (:866 is evaluated first) and the desired one; integrate, FK, check; on failure (5 mm deviation or
      // a joint on its limit: proximity 0) retry unconstrained from the state reached.
      const bool cv = fb.uf.clamp_joint_velocities != 0, cp_ = fb.uf.clamp_joint_positions != 0;
      chain_from_sincos<NJ>(lc, s.sn, s.cs, chain);
      V3 current_dir = chain.xe;
      double dq[NJ];
      V3 lin[NJ];
      if ((F & F_MLEGS) != 0 && fb.joint_moved) { // the Jacobian of the moved joints, the tip pose of before the move
        current_dir = base_rotate_inv(lc, fb.held_dir);
        jacobian_columns<NJ>(chain, lin);
        ik_step_cols<NJ>(lc, lin, fb.held_pe, s.q, s.qd, desired, dq);
      } else {
        ik_step<NJ>(lc, chain, s.q, s.qd, desired, dq);
      }
      if (rot_def) {
        update_joints<NJ>(lc, dq, P.dt, P.inv_dt, false, cp_, s.q, s.qd);
        joint_sincos<NJ>(lc, s.q, s.sn, s.cs);
        chain_from_sincos<NJ>(lc, s.sn, s.cs, chain);
        jacobian_columns<NJ>(chain, lin);
        ik_step_rotation<NJ>(lc, chain, lin, s.q, s.qd, tip_rotation_delta(current_dir, base_rotate_inv(lc, desired_dir)), dq);
      }
      double success = update_joints<NJ>(lc, dq, P.dt, P.inv_dt, cv, cp_, s.q, s.qd);
      joint_sincos<NJ>(lc, s.q, s.sn, s.cs);
      chain_from_sincos<NJ>(lc, s.sn, s.cs, chain);
      {
        V3 e = tip_robot_frame(lc, chain.pe) - desired;
        if (fabs(e.x) > kIkTolerance || fabs(e.y) > kIkTolerance || fabs(e.z) > kIkTolerance) {
          success = 0.0;
          s.word |= LW_IKFAIL;
        }
      }
      if (rot_def && success == 0.0) { // desired_tip_pose_.rotation_ = UNDEFINED_ROTATION; applyIK again (:932-936)
        retried = true;
        ik_step<NJ>(lc, chain, s.q, s.qd, desired, dq);
        update_joints<NJ>(lc, dq, P.dt, P.inv_dt, cv, cp_, s.q, s.qd);
        joint_sincos<NJ>(lc, s.q, s.sn, s.cs);
      }
    } else if (!(SHC_DBG(P) & 8)) {
      double dq[NJ];
      if (LegRegs<NJ>::kKeepJacobian) {
        ik_step_cols<NJ>(lc, s.lin, s.pe, s.q, s.qd, desired, dq);
      } else {
        chain_from_sincos<NJ>(lc, s.sn, s.cs, chain); // joint transforms left by the previous applyFK (model.cpp:731,744)
        ik_step<NJ>(lc, chain, s.q, s.qd, desired, dq);
      }
      update_joints<NJ>(lc, dq, P.dt, P.inv_dt, fb.uf.clamp_joint_velocities != 0, fb.uf.clamp_joint_positions != 0, s.q, s.qd);
    }
    SHC_PHASE_FENCE();
    SHC_TICK(10);
    if (!(SHC_DBG(P) & 16) && !rot_on) joint_sincos<NJ>(lc, s.q, s.sn, s.cs); // Leg::applyFK (:904); the rotation path left sn / cs current
    if (!(SHC_DBG(P) & 512)) chain_from_sincos<NJ>(lc, s.sn, s.cs, chain);
    V3 lin[NJ];
    jacobian_columns<NJ>(chain, lin);
    if (LegRegs<NJ>::kKeepJacobian) {
#pragma unroll
      for (int i = 0; i < NJ; ++i) s.lin[i] = lin[i];
      s.pe = chain.pe;
    }
    SHC_PHASE_FENCE();
    SHC_TICK(11);
    out.model_tip = (SHC_DBG(P) & 512) ? desired : tip_robot_frame(lc, chain.pe);
    if (FT::adm(P) || rot_on) s.tipx = base_rotate(lc, chain.xe);
    V3 e = out.model_tip - desired;
    if (fabs(e.x) > kIkTolerance || fabs(e.y) > kIkTolerance || fabs(e.z) > kIkTolerance) s.word |= LW_IKFAIL; // :916-929
    if (FT::tipf(P) && !(SHC_DBG(P) & 8192)) { // Leg::calculateTipForce (:667-708)
      double effort[NJ];
      in.effort(effort); // Joint::current_effort_ input
      V3 raw = tip_force_cols<NJ>(lc, chain, lin, effort);
      if constexpr (rot_on) {
        // (the rounding of the filter step is written out: this code is compiled into one-launch cycles and into the model half of
        //  two-launch cycles, which must agree bit for bit, and the contraction the compiler picks for a * b + c * d depends on its surroundings)
        const double gk = 0.15 * P.force_gain;
        s.tf = V3{fma(raw.x, gk, s.tf.x * (1 - 0.15)), fma(raw.y, gk, s.tf.y * (1 - 0.15)), fma(raw.z, gk, s.tf.z * (1 - 0.15))};
        if (retried) s.tf = V3{fma(raw.x, gk, s.tf.x * (1 - 0.15)), fma(raw.y, gk, s.tf.y * (1 - 0.15)), fma(raw.z, gk, s.tf.z * (1 - 0.15))};
      } else {
        s.tf = raw * (0.15 * P.force_gain) + s.tf * (1 - 0.15);
      }
    }
  }
  SHC_TICK(12);
}

// One control cycle of this lane's leg (and, redundantly per lane group, of its robot).
template <int L, int NJ, unsigned F, typename IN = LegInPlanes<NJ>, typename MID = NoHook>
__device__ __forceinline__ void cycle(LegRegs<NJ> &s, LegOut &out, const SharedConsts<L, NJ> &C, const RobTile<64 / L> &rb, const Park &pk,
                                      const Group<L> g, int leg, const double *__restrict__ legd, int64_t ns, uint32_t slot, unsigned &dirty,
                                      const bool manual_live, const bool touchdown_detection, double *ext, const ManualRobot *mr, const IN &in,
                                      const MID &mid = MID(), const double *span = nullptr, const bool pose_only = false) {
  FrontToBack fb;
  fb.planes_in_sync = false;
  fb.limit_bracket = -1;
  fb.uf = load_uni_flags(C.P);
  fb.pose_only = pose_only;
  cycle_front<L, NJ, F, true>(s, out, C, rb, pk, g, leg, legd, ns, slot, dirty, manual_live, touchdown_detection, ext, mr, in, fb, span);
  if ((F & F_MLEGS) != 0 && pose_only) return; // (RT_POSE_MARKED: updateWalk / updateStance / updateModel do not run for this robot in this loop)
  mid();
  cycle_back<L, NJ, F>(s, out, C, leg, legd, ns, slot, mr, in, fb);
}

#endif // __HIPCC__

} // namespace shc

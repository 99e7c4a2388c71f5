// shc_cycle_launch.hpp - the seam between the host side of libshc_batch.so (shc_engine.hip) and the fused cycle kernels
// (shc_cycle_kernel.hpp), which are compiled in one translation unit per morphology (shc_cycle_inst.hip, -DSHC_INST_L /
// -DSHC_INST_NJ) so that the library builds in parallel.  Plain data + one launcher per (legs, joints).
#pragma once

#include "shc_cycle.hpp"

#include <hip/hip_runtime.h>

namespace shc {

// ---- resident mode: StateController::loop's while-loop (src/main.cpp:106-131) kept on the chip.  The cycle kernel stays resident
//      (one wave per SIMD slot), per-leg state in registers and per-robot state in LDS across cycles; every cycle it waits for the
//      loop tick (a doorbell the host or a producer advances), takes the inputs the "callbacks" of that iteration delivered from
//      device-side rings, and writes the desired joint state of that cycle to an output ring.
enum : int { RG_VEL = 0, RG_IMU = 1, RG_POSE = 2, RG_RESET = 3, RG_FORCE = 4, RG_EFFORT = 5, RG_COUNT = 6 }; // input groups
// robot-input ring record: fields in this order, [position][wave][field][robots-per-wave]
enum : int { RIN_VEL = 0, RIN_IMU = 3, RIN_POSE = 10, RIN_COUNT = 16 }; // v(2) w(1) | quat(4) gyro(3) | tvi(3) rvi(3)

constexpr int kResidentHeaders = 1024; // header ring / direct-record ring entries (cycles that may be posted ahead)
// DIRECT posts (shc_engine_resident_bind_inputs + shc_cycle_inputs.direct): no kernel launch, no copy.  Up to kBoundSets sets of the caller's
// device arrays are bound before the loop starts (their addresses travel as kernel arguments); a direct post is then one 16-byte record in
// pinned, device-mapped memory - the cycle's tag, and { fresh-group mask, set } - which the relay wave, polling the host words anyway,
// turns into the cycle's header before it releases the cycle; the workers read their robots' inputs straight from the bound arrays.  The
// second word carries the low 16 bits of the tag in its top 16 bits: a record read while the host was writing it is seen as incomplete.
constexpr int kBoundSets = 4;
enum : int { BND_LIN = 0, BND_ANG = 1, BND_IMUQ = 2, BND_IMUW = 3, BND_FORCE = 4, BND_EFFORT = 5, BND_COUNT = 6 };
constexpr unsigned kResidentDirect = 1u << 15; // header mask bit: the fresh groups of this cycle are read from bound set (header >> 16) & 3
struct ResidentCtl { // device memory, polled with agent-scope loads; written by the relay wave only (exited: atomic, workers)
  unsigned long long gate;      // (stop << 32) | doorbell: run cycle c (counted from resident_begin) while c < min(doorbell, stop)
  unsigned long long exited;    // worker waves that have left the loop
  unsigned long long fault;     // != 0: a worker gave up waiting (emergency bound) - state may be inconsistent
  unsigned long long pad[5];    // [0] exit reason, [1] cycles completed by every wave (device copy of ResidentHost::done), [2] the relay has left
  unsigned long long dbg[80];   // development builds (-DSHC_RES2_TIMING): [0, 8) busy clocks, [8, 40) phase clocks of the walker, [40, 72) of the model wavefront of workgroup 1
};
struct ResidentHost { // pinned host memory mapped into the device (fine-grained): the host side of the handshake
  unsigned long long doorbell;  // host / producer -> device: cycles published since resident_begin
  unsigned long long stop;      // host -> device: stop after this many cycles (~0ull = keep running)
  unsigned long long done;      // device -> host: cycles completed by every wave, outputs visible
  unsigned long long exited;    // device -> host: 0 running | SHC_RESIDENT_* exit reason
  unsigned long long heartbeat; // device -> host: relay iterations (diagnostic)
  unsigned long long fault;
  unsigned long long late_reads; // device -> host: stream-ordered reads that gave up waiting for their cycle
  unsigned long long pad_[1];
  unsigned long long records[kResidentHeaders * 2]; // host -> device: direct-post records { tag, mask | set << 16 | (tag & 0xffff) << 48 }
};
struct ResidentHeader { // one per cycle (ring of kResidentHeaders), written by shc_engine_resident_post before the doorbell moves
  unsigned long long tag;  // cycle + 1; any other value: nothing was posted for this cycle (inputs held)
  unsigned short mask;     // bit g: group g is fresh this cycle
  unsigned char pos[RG_COUNT]; // ring position of each fresh group's data
};
static_assert(sizeof(ResidentHeader) == 16, "header is read as two 8-byte words");
enum : unsigned long long { RESIDENT_EXIT_STOP = 1, RESIDENT_EXIT_IDLE = 2, RESIDENT_EXIT_MAX = 3, RESIDENT_EXIT_FAULT = 4 };

struct ResidentArgs {
  ResidentCtl *ctl;
  ResidentHost *host;
  unsigned long long *progress; // [n_waves] cycles completed by wave w (its outputs are visible)
  ResidentHeader *headers;       // [kResidentHeaders] (written by the post kernels, and by the relay for direct posts)
  const double *bound[kBoundSets][BND_COUNT]; // the caller's device arrays of each bound input set (nullptr: not bound), instance-major as the C ABI takes them
  const double *rin;    // [depth][n_waves][RIN_COUNT][RPW]
  const int32_t *rini;  // [depth][n_waves][RPW] reset modes
  const double *force;  // [depth][2 planes][n_slots] double2 (as Fields::FORCE_IN)
  const double *effort; // [depth][NJE / 2 planes][n_slots] double2 (as Fields::EFFORT_IN)
  double *out;          // [depth][NJ planes][n_slots] double2: q, qd of every cycle (fields [0, 2 NJ) of the leg state)
  int depth;
  unsigned max_cycles;              // hard bound of this launch
  unsigned long long idle_ticks;    // wall-clock ticks without a new doorbell value before the relay stops the loop
  unsigned long long ticks_per_ms;  // wall_clock64() rate of the device (hipDeviceAttributeWallClockRate): every device-side time bound derives from it
  int64_t n_waves;
  double touchdown_threshold, liftoff_threshold; // Leg::touchdownDetection (model.cpp:712-722) runs inside the loop when a tip force arrives (rough terrain mode)
  // BATCH form of the same loop (shc_engine_step_k: K cycles per launch, each with its own inputs, for batches of any size).  batch_cycles > 0:
  // no relay, no doorbell, no header ring, nothing has to be co-resident - every wavefront runs cycles 0 .. batch_cycles - 1 as soon as it is
  // scheduled and leaves; cycle c takes row c of the K-deep input arrays bound as set 0 (batch_mask: the groups they carry; kstride[which]:
  // doubles per row) the way a direct post delivers them, and writes its q / qd to slot c of the output ring (depth = batch_cycles).
  unsigned batch_cycles, batch_mask;
  int64_t kstride[BND_COUNT];
  int64_t batch_wave0; // first wave of this launch
};

// Everything a launch of the cycle kernel needs from the engine.
struct ResidentFit;
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
  const ResidentArgs *resident; // != nullptr: launch the resident kernel (block = 64: grid = n_waves + 1 relay block of 64 threads, one
                                // wavefront per robot group; block = 256: the two-wavefront pipeline, grid = ceil(n_waves / 2) + 1)
  struct ResidentFit *fit;      // != nullptr: launch nothing, report whether / how densely the resident kernel of this specialisation fits
  int64_t wave0;                // first wave of this launch (a step of a large batch is two launches on two streams)
  int half_steps;               // gravity-aligned tips, legs of more than 3 joints: 0 = a cycle is two launches (walker half, model half) when
                                // the launch has wavefronts for two per SIMD, 1 = always, -1 = never (SHC_ROT_SPLIT, read by the engine)
};
struct ResidentFit {
  int supported;          // this specialisation has a resident kernel
  int blocks_per_cu;      // hipOccupancyMaxActiveBlocksPerMultiprocessor of it (64-thread blocks with its per-wave LDS)
  int two_wave;           // the two-wavefront pipeline exists for it (256-thread blocks, at most one per compute unit)
  int batch;              // the batch form (shc_engine_step_k's kernel) exists for it; without one step_k runs the K cycles as single launches
};

// One per (legs, joints); defined by shc_cycle_inst.hip.  Returns false when that morphology has no kernels in this build.
#define SHC_FOR_EACH_MORPHOLOGY(X) X(3, 3) X(4, 3) X(4, 4) X(4, 5) X(5, 3) X(6, 3) X(6, 4) X(6, 5) X(7, 3) X(8, 3) X(8, 4) X(8, 5)
#define SHC_DECLARE_LAUNCHER(L_, NJ_) void shc_launch_cycle_##L_##_##NJ_(const CycleLaunch &a);
SHC_FOR_EACH_MORPHOLOGY(SHC_DECLARE_LAUNCHER)
#undef SHC_DECLARE_LAUNCHER

} // namespace shc

// shc_resident.hpp - host side of resident mode (include/shc_batch.h: shc_engine_resident_*): the node's control loop
// (src/main.cpp:106-131 around StateController::loop, src/state_controller.cpp:162-193) kept on the chip.  Included by
// shc_engine.hip; the device side (worker loop, relay) is in shc_cycle_kernel.hpp, the protocol in shc_cycle_launch.hpp.
//
// Buffers (all device memory unless noted):
//   ResidentHost (pinned host memory, device-mapped)  doorbell / stop / done / exited - the only words that cross PCIe, read and
//                                                     written on the device by the relay wave alone
//   ResidentCtl, progress[n_waves], headers[1024]     relay <-> workers
//   rin / rini / force / effort [ring_depth][...]     input data rings, one position per POSTED set of a group (not per cycle):
//                                                     a group that is not posted costs nothing and its last data stays where it is
//   out [ring_depth][dof planes][n_slots]             q, qd of the last ring_depth cycles, in the engine's plane layout

struct Resident {
  bool active = false;
  int depth = 0;
  ResidentArgs args{};
  ResidentHost *host = nullptr;     // pinned host memory
  ResidentHost *host_dev = nullptr; // the device's address of it
  ResidentCtl *ctl = nullptr;
  unsigned long long *progress = nullptr;
  ResidentHeader *headers = nullptr;
  double *rin = nullptr, *force = nullptr, *effort = nullptr, *out = nullptr, *stage = nullptr;
  int32_t *rini = nullptr;
  size_t stage_bytes = 0;
  // host arrays handed to shc_engine_resident_post are copied into a ring of pinned, device-mapped staging slots; the post kernel
  // reads them from there over PCIe, so a post neither copies through a pageable-memory bounce buffer nor synchronises
  static constexpr int kPinSlots = 4;
  unsigned *post_blocks_done = nullptr; // device counter: the last block of a post-and-publish kernel rings the doorbell
  char *pin = nullptr, *pin_dev = nullptr;
  hipEvent_t pin_ev[kPinSlots] = {};
  unsigned long long pin_uses = 0;
  hipStream_t in_stream = nullptr;
  // The loop kernel runs on a stream of its own (ordered after the engine's stream at begin, synchronised at end): it never ends by
  // itself, and on the caller's stream - the legacy default stream in particular - it would block whoever else uses that stream.
  hipStream_t loop_stream = nullptr;
  hipEvent_t loop_ev = nullptr;
  unsigned long long published = 0;         // doorbell value
  unsigned long long posted = 0;            // cycles with a header (the next post is for this cycle)
  unsigned long long posts[RG_COUNT] = {};  // sets of each group posted so far (ring write positions)
  std::vector<unsigned long long> post_cycle[RG_COUNT]; // [depth]: the cycle each ring position was posted for
  unsigned max_cycles = 0;
  unsigned long long wall_khz = 100000;     // wall_clock64() rate of this device (hipDeviceAttributeWallClockRate; 100 MHz on MI300 / MI355X)
  bool poisoned = false;                    // the loop did not answer a stop request: the engine stays busy until the kernel is known to have left
  bool stream_doorbell_pending = false;     // a doorbell value travels on in_stream behind the posts it releases
  unsigned groups_posted = 0;
  bool two_wave = false;                    // the two-wavefront (walker / model) pipeline is running
  long long direct_last = -1;               // cycle of the latest direct post (the doorbell only moves once the relay has released it)
};

__global__ void resident_doorbell_kernel(ResidentHost *host, unsigned long long value) {
  __hip_atomic_store(&host->doorbell, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct ResidentPost {
  const double *lin, *ang, *imu_q, *imu_w, *tvi, *rvi, *force, *effort;
  const int32_t *reset;
  int pos[RG_COUNT];
  unsigned mask;
  unsigned long long cycle;
  unsigned long long doorbell; // != 0: the last block to finish rings the doorbell with this value (post + publish in one launch)
  ResidentHost *host;
  unsigned *blocks_done;       // device counter of that election
};
// One posted input set -> the data rings (write-through stores: the resident kernel reads them with agent-scope loads while it
// runs) + the header of its cycle.  The doorbell moves only after this kernel has completed (stream order).
__global__ void resident_post_kernel(ResidentPost pp, ResidentArgs A, ResidentHeader *headers, double *rin, int32_t *rini, double *force, double *effort,
                                     int64_t n, int L, int NJ, int64_t n_slots) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int rpw = 64 / L;
  auto put = [](double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (t < n) {
    const int64_t w = t / rpw;
    const int r = int(t - w * rpw);
    auto rec = [&](int grp, int field) { return rin + ((int64_t(pp.pos[grp]) * A.n_waves + w) * RIN_COUNT + field) * rpw + r; };
    if (pp.mask & (1u << RG_VEL)) {
      put(rec(RG_VEL, RIN_VEL), pp.lin[t * 2]);
      put(rec(RG_VEL, RIN_VEL + 1), pp.lin[t * 2 + 1]);
      put(rec(RG_VEL, RIN_VEL + 2), pp.ang[t]);
    }
    if (pp.mask & (1u << RG_IMU)) { // Model::setImuData input as shc_engine_set_imu stores it: the orientation normalised
      const Quat qn = normalized(Quat{pp.imu_q[t * 4], pp.imu_q[t * 4 + 1], pp.imu_q[t * 4 + 2], pp.imu_q[t * 4 + 3]}); // as scatter_rob_kernel
      put(rec(RG_IMU, RIN_IMU + 0), qn.w);
      put(rec(RG_IMU, RIN_IMU + 1), qn.x);
      put(rec(RG_IMU, RIN_IMU + 2), qn.y);
      put(rec(RG_IMU, RIN_IMU + 3), qn.z);
      for (int k = 0; k < 3; ++k) put(rec(RG_IMU, RIN_IMU + 4 + k), pp.imu_w[t * 3 + k]);
    }
    if (pp.mask & (1u << RG_POSE)) {
      for (int k = 0; k < 3; ++k) put(rec(RG_POSE, RIN_POSE + k), pp.tvi[t * 3 + k]);
      for (int k = 0; k < 3; ++k) put(rec(RG_POSE, RIN_POSE + 3 + k), pp.rvi[t * 3 + k]);
    }
    if (pp.mask & (1u << RG_RESET))
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(rini) + ((int64_t(pp.pos[RG_RESET]) * A.n_waves + w) * rpw + r),
                         (unsigned long long)(unsigned)pp.reset[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (t < n * L && (pp.mask & ((1u << RG_FORCE) | (1u << RG_EFFORT)))) {
    const int64_t rob = t / L;
    const int leg = int(t - rob * L);
    const int64_t slot = slot_of(rob, leg, L);
    if (pp.mask & (1u << RG_FORCE)) {
      double *base = force + int64_t(pp.pos[RG_FORCE]) * 2 * n_slots * 2;
      for (int k = 0; k < 3; ++k) put(base + leg_field_index(k, slot, n_slots), pp.force[t * 3 + k]);
    }
    if (pp.mask & (1u << RG_EFFORT)) {
      const int nje = (NJ + 1) & ~1;
      double *base = effort + int64_t(pp.pos[RG_EFFORT]) * (nje / 2) * n_slots * 2;
      for (int k = 0; k < NJ; ++k) put(base + leg_field_index(k, slot, n_slots), pp.effort[t * NJ + k]);
    }
  }
  if (t == 0) {
    unsigned long long h1 = pp.mask & 0xffffu;
    for (int gi = 0; gi < RG_COUNT; ++gi) h1 |= (unsigned long long)(pp.pos[gi] & 0xff) << (16 + 8 * gi);
    unsigned long long *hp = reinterpret_cast<unsigned long long *>(headers + (pp.cycle & (kResidentHeaders - 1)));
    __hip_atomic_store(hp + 1, h1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(hp, pp.cycle + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (pp.doorbell != 0) { // release the cycle once every block's stores are out: the last block to arrive rings the doorbell.
    // (No LDS here: a resident loop that fills the chip leaves none, and a kernel that asks for any would never be scheduled.)
    __threadfence(); // this thread's write-through stores are visible device-wide before its block is counted
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned arrived = __hip_atomic_fetch_add(pp.blocks_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (arrived == gridDim.x - 1) {
        __hip_atomic_store(pp.blocks_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (the next post on this stream starts from zero)
        __threadfence_system();
        __hip_atomic_store(&pp.host->doorbell, pp.doorbell, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

static void resident_free(Resident *r) {
  if (!r) return;
  if (r->host) (void)hipHostFree(r->host);
  (void)hipFree(r->ctl);
  (void)hipFree(r->progress);
  (void)hipFree(r->headers);
  (void)hipFree(r->rin);
  (void)hipFree(r->rini);
  (void)hipFree(r->force);
  (void)hipFree(r->effort);
  (void)hipFree(r->out);
  (void)hipFree(r->stage);
  (void)hipFree(r->post_blocks_done);
  if (r->pin) (void)hipHostFree(r->pin);
  for (hipEvent_t ev : r->pin_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (r->in_stream) (void)hipStreamDestroy(r->in_stream);
  if (r->loop_stream) (void)hipStreamDestroy(r->loop_stream);
  if (r->loop_ev) (void)hipEventDestroy(r->loop_ev);
  delete r;
}

static inline unsigned long long host_load(const unsigned long long *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline void host_store(unsigned long long *p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
static double now_seconds() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return double(ts.tv_sec) + 1e-9 * double(ts.tv_nsec);
}
// spin until pred() or the timeout; polite after the first 200 us
template <typename Pred>
static bool spin_until(Pred pred, double timeout_s) {
  const double t0 = now_seconds();
  for (unsigned it = 0;; ++it) {
    if (pred()) return true;
    const double dt = now_seconds() - t0;
    if (dt > timeout_s) return false;
    if (dt > 200e-6 && (it & 63) == 0) sched_yield();
  }
}

static int resident_require(shc_engine *e, bool active) {
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (active && !(e->res && e->res->active)) return fail(SHC_ERR_INVALID_ARG, "the engine is not in resident mode (shc_engine_resident_begin)");
  return SHC_OK;
}

extern "C" int shc_engine_resident_bind_inputs(shc_engine *e, int set, const shc_cycle_inputs *arrays) {
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (e->res && e->res->active) return fail(SHC_ERR_BUSY, "resident mode: input sets are bound while the loop is not running");
  if (set < 0 || set >= kBoundSets) return fail(SHC_ERR_INVALID_ARG, "input set 0 .. 3");
  if (arrays && (arrays->linear_xy == nullptr) != (arrays->angular == nullptr)) return fail(SHC_ERR_INVALID_ARG, "linear_xy and angular are bound together");
  if (arrays && (arrays->imu_orientation_wxyz == nullptr) != (arrays->imu_angular_velocity == nullptr)) return fail(SHC_ERR_INVALID_ARG, "the two IMU arrays are bound together");
  if (arrays && (arrays->pose_translation_velocity || arrays->pose_rotation_velocity || arrays->pose_reset_mode))
    return fail(SHC_ERR_INVALID_ARG, "input sets carry velocity, IMU, tip force and joint effort");
  const double *row[BND_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (arrays) {
    row[BND_LIN] = arrays->linear_xy, row[BND_ANG] = arrays->angular, row[BND_IMUQ] = arrays->imu_orientation_wxyz, row[BND_IMUW] = arrays->imu_angular_velocity;
    row[BND_FORCE] = arrays->tip_force, row[BND_EFFORT] = arrays->joint_effort;
  }
  memcpy(e->bound_inputs[set], row, sizeof row);
  return SHC_OK;
}

extern "C" int shc_engine_resident_begin(shc_engine *e, int ring_depth, int64_t max_cycles, int idle_timeout_ms) {
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (e->res && e->res->active) return fail(SHC_ERR_BUSY, "resident mode is already active");
  if (ring_depth < 2 || ring_depth > 255) return fail(SHC_ERR_INVALID_ARG, "ring_depth must be 2..255");
  if (max_cycles < 1 || max_cycles > 0x7ffffffe) return fail(SHC_ERR_INVALID_ARG, "max_cycles must be 1..2^31-2");
  if (idle_timeout_ms < 0 || idle_timeout_ms > 600000) return fail(SHC_ERR_INVALID_ARG, "idle_timeout_ms must be 0..600000");
  if (e->starting_up) return fail(SHC_ERR_UNSUPPORTED, "resident mode starts from a running engine (finish the start-up first)");
  {
    const int rc_remap = flush_step_remap(e);
    if (rc_remap != SHC_OK) return rc_remap;
  }
  HIP_TRY(hipSetDevice(e->device));
  {
    const int rc = join_side(e);
    if (rc != SHC_OK) return rc;
  }
  // does this configuration have a resident kernel, and does the whole batch fit the chip at once (+ the relay block)?
  ResidentFit fit{0, 0, 0, 0};
  {
    CycleLaunch a{e->st, e->d_consts, &e->cp, e->rt_flags, (e->features & SHC_FEAT_GENERIC_KERNEL) != 0, e->stream, 0, 64, 0, nullptr, &fit, 0};
#define CALL(L_, NJ_) shc_launch_cycle_##L_##_##NJ_(a)
    SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
  }
  if (!fit.supported)
    return fail(SHC_ERR_UNSUPPORTED, (e->rt_flags & RT_MANUAL_LEGS)
                                         ? "resident mode: this configuration runs on a manual-leg kernel (a leg has been toggled / planner mode), which has no resident form"
                                         : "resident mode: this configuration runs on a runtime-flag kernel with rough terrain / tip-align / tip-rotation logic, whose resident "
                                           "form is not part of this build (SHC_GENERIC_LOOP_FORMS=1 builds it); use shc_engine_step or shc_engine_step_k");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, e->device));
  // ... less one compute unit's worth per XCD (workgroups are dealt round-robin to the 8 XCDs and placed only inside their own): the loop's
  // wavefronts hold their registers and LDS for as long as it runs, and the kernels that post inputs, ring the doorbell and read the
  // output ring (64-thread workgroups without LDS) need somewhere to run next to it on EVERY XCD - measured: with 8 free wave slots on
  // the chip a post kernel is never scheduled, with ~150 it is
  constexpr int kXcds = 8; // gfx950 (the only target of this library): 8 XCDs of 32 compute units; HIP has no attribute for it
  const int64_t capacity = int64_t(fit.blocks_per_cu) * (prop.multiProcessorCount - kXcds);
  if (e->n_waves + 1 > capacity)
    return fail(SHC_ERR_UNSUPPORTED, "resident mode: the batch needs " + std::to_string(e->n_waves + 1) + " co-resident wavefronts, this device holds " +
                                         std::to_string(capacity) + " of this kernel (" + std::to_string(fit.blocks_per_cu) + " per compute unit); use shc_engine_step");
  const int NJE = (e->NJ + 1) & ~1;
  const size_t out_slot_bytes = size_t(e->NJ) * e->n_slots * 16;
  if (out_slot_bytes * size_t(ring_depth) >= (size_t(1) << 31)) return fail(SHC_ERR_INVALID_ARG, "ring_depth x batch: the output ring must stay below 2 GiB");
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->res && e->res->depth != ring_depth) {
    resident_free(e->res);
    e->res = nullptr;
  }
  if (!e->res) {
    Resident *r = new Resident();
    e->res = r;
    r->depth = ring_depth;
    const int rpw = 64 / e->L;
    auto bail = [&](hipError_t err, const char *what) {
      resident_free(r);
      e->res = nullptr;
      return fail(SHC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(err));
    };
    hipError_t err;
    if ((err = hipHostMalloc(reinterpret_cast<void **>(&r->host), sizeof(ResidentHost), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
      return bail(err, "hipHostMalloc(ResidentHost)");
    if ((err = hipHostGetDevicePointer(reinterpret_cast<void **>(&r->host_dev), r->host, 0)) != hipSuccess) return bail(err, "hipHostGetDevicePointer");
    if ((err = hipStreamCreateWithFlags(&r->in_stream, hipStreamNonBlocking)) != hipSuccess) return bail(err, "hipStreamCreate");
    // The loop's stream sits in a PRIORITY CLASS OF ITS OWN (the lowest).  HIP maps streams onto a few hardware queues per priority class, and
    // a hardware queue is in-order: a stream that shares the loop's queue would wait for the loop kernel to END - a post kernel, a read of the
    // output ring, the caller's collective would starve until the idle timeout (seen once in a long test process, when a stream a test had
    // created landed on the loop's queue).  Streams of normal priority - the caller's, PyTorch's, RCCL's, this engine's input stream - never
    // share a queue with it.  (The loop is not slowed down: once its wavefronts are resident nothing preempts them; measured equal.)
    {
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
      if ((err = hipStreamCreateWithPriority(&r->loop_stream, hipStreamNonBlocking, least)) != hipSuccess) return bail(err, "hipStreamCreate");
    }
    if ((err = hipEventCreateWithFlags(&r->loop_ev, hipEventDisableTiming)) != hipSuccess) return bail(err, "hipEventCreate");
    r->stage_bytes = size_t(e->n) * (16 * 8 + 8) + size_t(e->n) * e->L * (3 + e->NJ) * 8 + 256;
    const size_t D = size_t(ring_depth);
    if ((err = hipMalloc(&r->ctl, sizeof(ResidentCtl))) != hipSuccess) return bail(err, "hipMalloc");
    if ((err = hipMalloc(&r->progress, size_t(e->n_waves) * 8)) != hipSuccess) return bail(err, "hipMalloc");
    if ((err = hipMalloc(&r->headers, sizeof(ResidentHeader) * kResidentHeaders)) != hipSuccess) return bail(err, "hipMalloc");
    if ((err = hipMalloc(&r->rin, D * e->n_waves * RIN_COUNT * rpw * 8)) != hipSuccess) return bail(err, "hipMalloc");
    if ((err = hipMalloc(&r->rini, D * e->n_waves * rpw * 8)) != hipSuccess) return bail(err, "hipMalloc"); // (one 8-byte word per reset mode)
    if ((err = hipMalloc(&r->force, D * 2 * e->n_slots * 16)) != hipSuccess) return bail(err, "hipMalloc");
    if ((err = hipMalloc(&r->effort, D * (NJE / 2) * e->n_slots * 16)) != hipSuccess) return bail(err, "hipMalloc");
    if ((err = hipMalloc(&r->out, D * out_slot_bytes)) != hipSuccess) return bail(err, "hipMalloc");
    if ((err = hipMalloc(&r->stage, r->stage_bytes)) != hipSuccess) return bail(err, "hipMalloc");
    if ((err = hipMalloc(&r->post_blocks_done, 64)) != hipSuccess) return bail(err, "hipMalloc");
    if ((err = hipMemsetAsync(r->post_blocks_done, 0, 64, e->stream)) != hipSuccess) return bail(err, "hipMemset");
    if ((err = hipHostMalloc(reinterpret_cast<void **>(&r->pin), r->stage_bytes * Resident::kPinSlots, hipHostMallocMapped)) != hipSuccess)
      return bail(err, "hipHostMalloc(input staging)");
    if ((err = hipHostGetDevicePointer(reinterpret_cast<void **>(&r->pin_dev), r->pin, 0)) != hipSuccess) return bail(err, "hipHostGetDevicePointer");
    for (hipEvent_t &ev : r->pin_ev)
      if ((err = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail(err, "hipEventCreate");
    for (int gi = 0; gi < RG_COUNT; ++gi) r->post_cycle[gi].assign(D, 0);
  }
  Resident *r = e->res;
  {
    ResidentCtl ctl0{};
    ctl0.gate = (unsigned long long)(unsigned(max_cycles)) << 32; // nothing released yet, stop at the launch's bound (the relay takes over from here)
    HIP_TRY(hipMemcpyAsync(r->ctl, &ctl0, sizeof ctl0, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  HIP_TRY(hipMemsetAsync(r->progress, 0, size_t(e->n_waves) * 8, e->stream));
  HIP_TRY(hipMemsetAsync(r->headers, 0, sizeof(ResidentHeader) * kResidentHeaders, e->stream));
  memset(r->host, 0, sizeof(ResidentHost));
  r->host->stop = ~0ull;
  __sync_synchronize();
  r->published = r->posted = 0;
  r->direct_last = -1;
  r->groups_posted = 0;
  r->stream_doorbell_pending = false;
  for (int gi = 0; gi < RG_COUNT; ++gi) {
    r->posts[gi] = 0;
    std::fill(r->post_cycle[gi].begin(), r->post_cycle[gi].end(), 0ull);
  }
  r->max_cycles = unsigned(max_cycles);
  ResidentArgs &A = r->args;
  A.ctl = r->ctl;
  A.host = r->host_dev;
  A.progress = r->progress;
  A.headers = r->headers;
  memcpy(A.bound, e->bound_inputs, sizeof A.bound);
  A.rin = r->rin;
  A.rini = r->rini;
  A.force = r->force;
  A.effort = r->effort;
  A.out = r->out;
  A.depth = ring_depth;
  A.max_cycles = r->max_cycles;
  int wall_khz = 0; // wall_clock64() rate (100 MHz on MI300 / MI355X)
  if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, e->device) != hipSuccess || wall_khz <= 0) wall_khz = 100000;
  r->wall_khz = (unsigned long long)wall_khz;
  A.idle_ticks = (unsigned long long)(idle_timeout_ms ? idle_timeout_ms : 5000) * r->wall_khz;
  A.ticks_per_ms = r->wall_khz;
  A.n_waves = e->n_waves;
  A.touchdown_threshold = e->params.touchdown_threshold;
  A.liftoff_threshold = e->params.liftoff_threshold;
  e->plan_poser_tips_current = false;
  // Two wavefronts per robot group (walker / model halves of the cycle pipelined over two SIMDs) while every 256-thread workgroup
  // - two robot groups - and the relay get a compute unit of their own; one wavefront per group above that.
  const bool two_wave = fit.two_wave && !(e->features & SHC_FEAT_RESIDENT_ONE_WAVE) && (e->n_waves + 1) / 2 + 1 <= prop.multiProcessorCount;
  r->two_wave = two_wave;
  HIP_TRY(hipEventRecord(r->loop_ev, e->stream)); // everything the engine's stream holds (state, the buffers set up above) comes first
  HIP_TRY(hipStreamWaitEvent(r->loop_stream, r->loop_ev, 0));
  CycleLaunch a{e->st, e->d_consts, &e->cp, e->rt_flags, (e->features & SHC_FEAT_GENERIC_KERNEL) != 0, r->loop_stream,
                two_wave ? unsigned((e->n_waves + 1) / 2 + 1) : unsigned(e->n_waves + 1), two_wave ? 256 : 64, 0, &A, nullptr, 0};
#define CALL(L_, NJ_) shc_launch_cycle_##L_##_##NJ_(a)
  SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
  HIP_TRY(hipGetLastError());
  r->active = true;
  return SHC_OK;
}

// A doorbell value releases every cycle below it in one go, and the relay looks for a direct-post record only while the doorbell has not
// moved past its cycle: before ANY doorbell value that would pass a direct post is written - by shc_engine_resident_publish, by the post
// kernel of an ordinary post with publish = 1, by shc_engine_resident_end - the relay must have installed that post's record (seen here as
// "its cycle has completed").  Bounded; a loop that has stopped by itself ends the wait (the caller reports that).
static int resident_wait_direct_installed(Resident *r) {
  if (r->direct_last < 0) return SHC_OK;
  const unsigned long long need = (unsigned long long)r->direct_last + 1;
  if (!spin_until([&] { return host_load(&r->host->done) >= need || host_load(&r->host->exited) != 0; }, 5.0))
    return fail(SHC_ERR_TIMEOUT, "resident mode: a direct post is still waiting to run");
  r->direct_last = -1;
  return SHC_OK;
}

// the device loop is still running (it has not stopped by itself)
static int resident_alive(Resident *r) {
  if (const unsigned long long why = host_load(&r->host->exited))
    return fail(SHC_ERR_TIMEOUT, std::string("resident mode: the device loop has stopped by itself (") +
                                     (why == RESIDENT_EXIT_IDLE ? "idle timeout" : why == RESIDENT_EXIT_MAX ? "max_cycles" : why == RESIDENT_EXIT_FAULT ? "fault" : "stop") +
                                     " after " + std::to_string(host_load(&r->host->done)) + " cycles): call shc_engine_resident_end");
  return SHC_OK;
}

extern "C" int shc_engine_resident_post(shc_engine *e, const shc_cycle_inputs *in, int64_t *cycle) {
  int rc = resident_require(e, true);
  if (rc != SHC_OK) return rc;
  Resident *r = e->res;
  if ((rc = resident_alive(r)) != SHC_OK) return rc;
  if (!in) return fail(SHC_ERR_INVALID_ARG, "inputs is NULL");
  if ((in->linear_xy == nullptr) != (in->angular == nullptr)) return fail(SHC_ERR_INVALID_ARG, "linear_xy and angular are posted together");
  if ((in->imu_orientation_wxyz == nullptr) != (in->imu_angular_velocity == nullptr)) return fail(SHC_ERR_INVALID_ARG, "the two IMU arrays are posted together");
  if ((in->pose_translation_velocity == nullptr) != (in->pose_rotation_velocity == nullptr))
    return fail(SHC_ERR_INVALID_ARG, "the two pose input arrays are posted together");
  unsigned mask = 0;
  if (in->linear_xy) mask |= 1u << RG_VEL;
  if (in->imu_orientation_wxyz) mask |= 1u << RG_IMU;
  if (in->pose_translation_velocity) mask |= 1u << RG_POSE;
  if (in->pose_reset_mode) mask |= 1u << RG_RESET;
  if (in->tip_force) mask |= 1u << RG_FORCE;
  if (in->joint_effort) mask |= 1u << RG_EFFORT;
  if ((mask & ((1u << RG_POSE) | (1u << RG_RESET))) && !(e->rt_flags & RT_MANUAL_LIVE))
    return fail(SHC_ERR_UNSUPPORTED, "pose inputs: give the engine one (shc_engine_set_pose_input / set_pose_reset_mode) before shc_engine_resident_begin");
  if ((mask & (1u << RG_EFFORT)) && !(e->rt_flags & RT_EFFORT_LIVE))
    return fail(SHC_ERR_UNSUPPORTED, "joint efforts: give the engine one (shc_engine_set_joint_effort) before shc_engine_resident_begin");
  const unsigned long long c = r->posted;
  if (c < r->published) return fail(SHC_ERR_INVALID_ARG, "resident mode: this cycle has already been published without inputs");
  if (c >= r->max_cycles) return fail(SHC_ERR_INVALID_ARG, "resident mode: past max_cycles");
  HIP_TRY(hipSetDevice(e->device));
  // the header slot of cycle c last served cycle c - 1024; a ring position of group g last served post k - depth, which stays in
  // force until the cycle post k - depth + 1 was made for has been reached by every wave
  const double patience = 5.0;
  if (c >= kResidentHeaders && !spin_until([&] { return host_load(&r->host->done) > c - kResidentHeaders; }, patience))
    return fail(SHC_ERR_TIMEOUT, "resident mode: 1024 posted cycles are waiting to run");
  if (in->direct) {
    // Launch-free post: one 16-byte record in the host-mapped ring; the relay turns it into the cycle's header and releases the cycle.
    const int set = in->direct - 1;
    if (set < 0 || set >= kBoundSets) return fail(SHC_ERR_INVALID_ARG, "direct: bound input set 1 .. 4");
    if (mask & ((1u << RG_POSE) | (1u << RG_RESET))) return fail(SHC_ERR_INVALID_ARG, "direct posts carry velocity, IMU, tip force and joint effort; post pose inputs / reset modes the ordinary way");
    if (mask == 0) return fail(SHC_ERR_INVALID_ARG, "direct post without inputs (use shc_engine_resident_publish)");
    const int need[RG_COUNT] = {BND_LIN, BND_IMUQ, -1, -1, BND_FORCE, BND_EFFORT};
    for (int gi = 0; gi < RG_COUNT; ++gi)
      if ((mask & (1u << gi)) && (need[gi] < 0 || !r->args.bound[set][need[gi]]))
        return fail(SHC_ERR_INVALID_ARG, "direct post: that group is not part of the bound input set (shc_engine_resident_bind_inputs before shc_engine_resident_begin)");
    if (r->published < c) { // cycles before this one that are still unpublished are released first, the ordinary way
      const int rc2 = shc_engine_resident_publish(e, int64_t(c - r->published));
      if (rc2 != SHC_OK) return rc2;
    }
    const unsigned long long tag = c + 1;
    volatile unsigned long long *rec = r->host->records + size_t(c & (kResidentHeaders - 1)) * 2;
    rec[1] = (unsigned long long)mask | ((unsigned long long)set << 16) | ((tag & 0xffffull) << 48);
    host_store(const_cast<unsigned long long *>(&rec[0]), tag); // (release: the word above is in place before the tag)
    r->groups_posted |= mask;
    r->posted = r->published = c + 1;
    r->direct_last = (long long)c;
    if (cycle) *cycle = int64_t(c);
    return SHC_OK;
  }
  ResidentPost pp{};
  pp.mask = mask;
  pp.cycle = c;
  for (int gi = 0; gi < RG_COUNT; ++gi) {
    if (!(mask & (1u << gi))) continue;
    const unsigned long long k = r->posts[gi];
    pp.pos[gi] = int(k % r->depth);
    if (k >= (unsigned long long)r->depth) {
      const unsigned long long successor_cycle = r->post_cycle[gi][(k + 1) % r->depth]; // cycle of post k - depth + 1
      if (!spin_until([&] { return host_load(&r->host->done) >= successor_cycle || host_load(&r->host->exited) != 0; }, patience))
        return fail(SHC_ERR_TIMEOUT, "resident mode: ring_depth posted input sets are waiting to be consumed (publish them)");
    }
  }
  // host arrays: into the next pinned staging slot (free once the post kernel that read it kPinSlots posts ago has completed)
  size_t off = 0;
  const int pin_slot = int(r->pin_uses % Resident::kPinSlots);
  if (!in->on_device && r->pin_uses >= (unsigned long long)Resident::kPinSlots) HIP_TRY(hipEventSynchronize(r->pin_ev[pin_slot]));
  auto dev = [&](const void *src, size_t bytes, const void **out) -> int {
    if (in->on_device) {
      *out = src;
      return SHC_OK;
    }
    if (off + bytes > r->stage_bytes) return fail(SHC_ERR_INVALID_ARG, "staging buffer too small");
    const size_t at = size_t(pin_slot) * r->stage_bytes + off;
    memcpy(r->pin + at, src, bytes);
    *out = r->pin_dev + at;
    off += (bytes + 15) & ~size_t(15);
    return SHC_OK;
  };
  const size_t n = size_t(e->n), nl = n * e->L;
#define STAGE(field, member, bytes)                                                              \
  if (in->member) {                                                                              \
    const void *d_;                                                                              \
    if ((rc = dev(in->member, bytes, &d_)) != SHC_OK) return rc;                                 \
    pp.field = reinterpret_cast<decltype(pp.field)>(d_);                                         \
  }
  STAGE(lin, linear_xy, n * 16)
  STAGE(ang, angular, n * 8)
  STAGE(imu_q, imu_orientation_wxyz, n * 32)
  STAGE(imu_w, imu_angular_velocity, n * 24)
  STAGE(tvi, pose_translation_velocity, n * 24)
  STAGE(rvi, pose_rotation_velocity, n * 24)
  STAGE(reset, pose_reset_mode, n * 4)
  STAGE(force, tip_force, nl * 24)
  STAGE(effort, joint_effort, nl * e->NJ * 8)
#undef STAGE
  const int64_t threads = (mask & ((1u << RG_FORCE) | (1u << RG_EFFORT))) ? int64_t(nl) : int64_t(n);
  // publish: release this cycle (and any cycles before it that were left unpublished) as soon as its inputs are in place - the post
  // kernel rings the doorbell itself, one launch instead of two per loop iteration
  const bool publish = in->publish != 0;
  if (publish) {
    if ((rc = resident_wait_direct_installed(r)) != SHC_OK) return rc; // (the post kernel rings the doorbell itself: the same rule as in publish)
    pp.doorbell = c + 1;
    pp.host = r->host_dev;
    pp.blocks_done = r->post_blocks_done;
  }
  resident_post_kernel<<<dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, r->in_stream>>>(pp, r->args, r->headers, r->rin, r->rini, r->force, r->effort,
                                                                                              e->n, e->L, e->NJ, e->n_slots);
  HIP_TRY(hipGetLastError());
  if (!in->on_device) { // (the caller's host arrays were copied: they are free again on return)
    HIP_TRY(hipEventRecord(r->pin_ev[pin_slot], r->in_stream));
    r->pin_uses++;
  }
  for (int gi = 0; gi < RG_COUNT; ++gi)
    if (mask & (1u << gi)) {
      r->post_cycle[gi][r->posts[gi] % r->depth] = c;
      r->posts[gi]++;
    }
  r->groups_posted |= mask;
  r->posted = c + 1;
  if (publish) r->published = c + 1;
  r->stream_doorbell_pending = true; // (a post is in flight on in_stream: the doorbell that releases it must follow it there)
  if (cycle) *cycle = int64_t(c);
  return SHC_OK;
}

extern "C" int shc_engine_resident_publish(shc_engine *e, int64_t n_cycles) {
  int rc = resident_require(e, true);
  if (rc != SHC_OK) return rc;
  Resident *r = e->res;
  if ((rc = resident_alive(r)) != SHC_OK) return rc;
  if (n_cycles < 0) return fail(SHC_ERR_INVALID_ARG, "n_cycles < 0");
  if (n_cycles == 0) return SHC_OK;
  if (r->published + (unsigned long long)n_cycles > r->max_cycles) return fail(SHC_ERR_INVALID_ARG, "resident mode: past max_cycles");
  // (a doorbell value releases everything below it in one go: it may only pass a direct post once the relay has installed that post's record)
  if ((rc = resident_wait_direct_installed(r)) != SHC_OK) return rc;
  r->published += (unsigned long long)n_cycles;
  if (r->posted < r->published) r->posted = r->published; // cycles released without a post run with the inputs held
  if (r->stream_doorbell_pending) {
    // posts may still be in flight on the input stream: once they have all completed the doorbell is a plain store, until
    // then it has to queue behind them
    HIP_TRY(hipSetDevice(e->device));
    if (hipStreamQuery(r->in_stream) == hipSuccess) {
      r->stream_doorbell_pending = false;
    } else {
      resident_doorbell_kernel<<<dim3(1), dim3(1), 0, r->in_stream>>>(r->host_dev, r->published);
      HIP_TRY(hipGetLastError());
      return SHC_OK;
    }
  }
  host_store(&r->host->doorbell, r->published);
  return SHC_OK;
}

extern "C" int shc_engine_resident_wait(shc_engine *e, int64_t cycles, int timeout_ms) {
  int rc = resident_require(e, true);
  if (rc != SHC_OK) return rc;
  Resident *r = e->res;
  if (cycles < 0 || (unsigned long long)cycles > r->published) return fail(SHC_ERR_INVALID_ARG, "resident mode: waiting for cycles that were not published");
  const unsigned long long want = (unsigned long long)cycles;
  if (!spin_until([&] { return host_load(&r->host->done) >= want || host_load(&r->host->exited) != 0; }, timeout_ms > 0 ? timeout_ms * 1e-3 : 10.0))
    return fail(SHC_ERR_TIMEOUT, "resident mode: the published cycles did not complete in time");
  if (host_load(&r->host->done) < want) return fail(SHC_ERR_TIMEOUT, "resident mode: the device loop stopped before these cycles ran");
  return SHC_OK;
}

extern "C" int shc_engine_resident_get_joint_state(shc_engine *e, int64_t cycle, double *q, double *qd, int on_device) {
  int rc = resident_require(e, true);
  if (rc != SHC_OK) return rc;
  Resident *r = e->res;
  const unsigned long long done = host_load(&r->host->done);
  if (cycle < 0 || (unsigned long long)cycle >= done) return fail(SHC_ERR_INVALID_ARG, "resident mode: that cycle has not completed (shc_engine_resident_wait)");
  if (r->published > (unsigned long long)cycle + r->depth) // (cycle + ring_depth shares its ring position and has been released)
    return fail(SHC_ERR_INVALID_ARG, "resident mode: that cycle's outputs may have been overwritten (ring_depth newer cycles were published)");
  HIP_TRY(hipSetDevice(e->device));
  const double *slot = r->out + size_t(cycle % r->depth) * size_t(e->NJ) * e->n_slots * 2;
  const int64_t threads = e->n * e->L;
  for (int which = 0; which < 2; ++which) {
    double *dst = which ? qd : q;
    if (!dst) continue;
    double *d = on_device ? dst : r->stage;
    gather_leg_kernel<<<dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, r->in_stream>>>(d, slot, e->n_slots, e->n, e->L, e->NJ,
                                                                                            which ? LEG_FIELD(e, QD) : LEG_FIELD(e, Q));
    HIP_TRY(hipGetLastError());
    if (!on_device) {
      HIP_TRY(hipMemcpyAsync(dst, d, size_t(threads) * e->NJ * 8, hipMemcpyDeviceToHost, r->in_stream));
      HIP_TRY(hipStreamSynchronize(r->in_stream));
    }
  }
  // Device buffers: the gathers ran on the engine's private input stream, which the caller cannot order anything after - and the ring slot
  // they read is only protected against newer cycles by the check above, made now.  The call therefore returns when q / qd are complete
  // (like the host-buffer form); the stream-ordered form is shc_engine_resident_get_joint_state_async.
  if (on_device) HIP_TRY(hipStreamSynchronize(r->in_stream));
  return SHC_OK;
}

// Device-side wait for a cycle of the resident loop: one thread watches the relay's count of completed cycles (its device copy),
// bounded - a loop that stopped or never gets the cycle must not leave a kernel spinning on the caller's stream.
__global__ void resident_wait_done_kernel(const ResidentCtl *ctl, ResidentHost *host, unsigned long long want, unsigned long long timeout_ticks) {
  const unsigned long long t0 = wall_clock64();
  auto done = [&] { return __hip_atomic_load(&ctl->pad[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); };
  for (;;) {
    if (done() >= want) return;
    if (__hip_atomic_load(&ctl->pad[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0 && done() < want) break; // the loop has ended without it
    if (wall_clock64() - t0 > timeout_ticks) break;
    __builtin_amdgcn_s_sleep(8);
  }
  __hip_atomic_fetch_add(&host->late_reads, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Stream-ordered variant of shc_engine_resident_get_joint_state: nothing here waits on the host.  The engine's stream gets a
// device-side wait for `cycle` followed by the gather into the caller's device buffers, so whatever the caller queues on that
// stream afterwards (the all-gather of the fleet's joint buffer) runs as soon as the cycle's outputs exist - its launch latency
// hides behind the cycles still running.  The cycle may be one that has not been published yet.
extern "C" int shc_engine_resident_get_joint_state_async(shc_engine *e, int64_t cycle, double *q, double *qd, int timeout_ms) {
  int rc = resident_require(e, true);
  if (rc != SHC_OK) return rc;
  Resident *r = e->res;
  if ((rc = resident_alive(r)) != SHC_OK) return rc;
  if (cycle < 0 || (unsigned long long)cycle >= r->max_cycles) return fail(SHC_ERR_INVALID_ARG, "resident mode: no such cycle (max_cycles)");
  if (r->published > (unsigned long long)cycle + r->depth)
    return fail(SHC_ERR_INVALID_ARG, "resident mode: that cycle's outputs may have been overwritten (ring_depth newer cycles were published)");
  HIP_TRY(hipSetDevice(e->device));
  const unsigned long long ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 5000) * r->wall_khz; // wall_clock64 ticks per millisecond
  resident_wait_done_kernel<<<dim3(1), dim3(1), 0, e->stream>>>(r->ctl, r->host_dev, (unsigned long long)cycle + 1, ticks);
  HIP_TRY(hipGetLastError());
  const double *slot = r->out + size_t(cycle % r->depth) * size_t(e->NJ) * e->n_slots * 2;
  const int64_t threads = e->n * e->L;
  for (int which = 0; which < 2; ++which) {
    double *dst = which ? qd : q;
    if (!dst) continue;
    gather_leg_kernel<<<dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, e->stream>>>(dst, slot, e->n_slots, e->n, e->L, e->NJ,
                                                                                         which ? LEG_FIELD(e, QD) : LEG_FIELD(e, Q));
    HIP_TRY(hipGetLastError());
  }
  return SHC_OK;
}

extern "C" int shc_engine_resident_status(shc_engine *e, int64_t *published, int64_t *completed, int32_t *running) {
  int rc = resident_require(e, false);
  if (rc != SHC_OK) return rc;
  Resident *r = e->res;
  const bool active = r && r->active;
  if (published) *published = active ? int64_t(r->published) : 0;
  if (completed) *completed = active ? int64_t(host_load(&r->host->done)) : 0;
  if (running) *running = active && host_load(&r->host->exited) == 0 ? 1 : 0;
  return SHC_OK;
}

extern "C" int shc_engine_resident_end(shc_engine *e, int64_t *cycles_run) {
  int rc = resident_require(e, true);
  if (rc != SHC_OK) return rc;
  Resident *r = e->res;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(r->in_stream)); // every post and stream-ordered doorbell has landed
  // (as in publish: the doorbell passes a direct post only after the relay has installed it.  A direct post that never ran - the loop is wedged -
  //  is reported, and the stop request below still goes out: with the doorbell left where it is the loop stops at the cycles it has released)
  const int direct_rc = resident_wait_direct_installed(r);
  const std::string direct_msg = direct_rc != SHC_OK ? std::string(shc_last_error()) : std::string();
  if (direct_rc == SHC_OK) host_store(&r->host->doorbell, r->published);
  host_store(&r->host->stop, direct_rc == SHC_OK ? r->published : 0ull); // (0: the relay stops the loop at the cycles it has already released)
  // A loop that does not answer keeps the engine: while the kernel may still be running nothing else may touch the state planes, and
  // shc_engine_destroy must not free the rings under it.  The engine stays in resident mode (every other entry point returns
  // SHC_ERR_BUSY); a later shc_engine_resident_end tries again - the loop's own bounds (max_cycles, idle timeout, the workers'
  // emergency bound) end it eventually.
  const bool answered = spin_until([&] { return host_load(&r->host->exited) != 0; }, r->poisoned ? 5.0 : 30.0);
  if (!answered) {
    r->poisoned = true;
    return fail(SHC_ERR_TIMEOUT, "resident mode: the device loop did not answer the stop request in time; the engine stays in resident mode - call "
                                 "shc_engine_resident_end again");
  }
  hipError_t err = hipStreamSynchronize(r->loop_stream); // (the host has waited: whatever follows on the engine's stream is ordered)
  // stream-ordered reads queued on the engine's stream are bounded (their own timeout, or the end of the loop): wait for them too before
  // their verdict (late_reads) is sampled
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  r->active = false;
  r->poisoned = false;
#ifdef SHC_RES2_TIMING
  if (err == hipSuccess) {
    ResidentCtl c;
    (void)hipMemcpy(&c, r->ctl, sizeof c, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w)
      fprintf(stderr, "[res2 timing] %s: %.0f clocks busy per steady REAL iteration (%llu of them), loop %llu clocks, %llu iterations\n", w ? "model " : "walker",
              c.dbg[4 * w + 1] ? double(c.dbg[4 * w]) / double(c.dbg[4 * w + 1]) : 0.0, c.dbg[4 * w + 1], c.dbg[4 * w + 2], c.dbg[4 * w + 3]);
#ifndef SHC_RES2_BUSY_ONLY
    const int order[] = {20, 21, 2, 3, 4, 5, 6, 7, 16, 8, 17, 15, 29, 22, 23};
    const char *wname[] = {"control words (LDS) read", "inputs of the cycle taken", "cycle_front entry", "robot word / command read, stop predicates", "(pose elsewhere)", "getLimit",
                           "velocity shaping", "walk state machine", "stepper (stride, Bezier, tip)", "tip rotation / iteratePhase / planes / leg word", "wait for the pose flag",
                           "pose read + updateStance", "control input for the next pose published", "mailbox written", "to the barrier"};
    const double its = c.dbg[1] ? double(c.dbg[1]) : 1.0;
    fprintf(stderr, "[res2 timing] walker of pair 0, mean clocks per phase:\n");
    for (int i = 0; i < int(sizeof(order) / sizeof(int)); ++i) fprintf(stderr, "[res2 timing]   walker t%-2d %-58s %7.0f\n", order[i], wname[i], double(c.dbg[8 + order[i]]) / its);
    const int morder[] = {29, 24, 30, 25, 31, 26, 9, 10, 11, 12, 27, 13, 28, 14};
    const char *mname[] = {"control words (LDS) read", "leader: gate, what the next iteration is, header prefetch issued", "pose inputs of the cycle taken",
                           "updateCurrentPose of the walker's cycle + flag", "leg inputs in force, joint efforts prefetched", "mailbox read, odometry, admittance",
                           "cycle_back entry (desired tip)", "IK step + joint update", "sin / cos, chain, Jacobian", "model tip, tip force", "cycle_back exit",
                           "wait for the previous stores + progress word", "output-ring stores issued", "control words written (leader: waits for its loads)"};
    const double mits = c.dbg[5] ? double(c.dbg[5]) : 1.0;
    fprintf(stderr, "[res2 timing] model wavefront of pair 0 (the leader), mean clocks per phase:\n");
    for (int i = 0; i < int(sizeof(morder) / sizeof(int)); ++i) fprintf(stderr, "[res2 timing]   model  t%-2d %-58s %7.0f\n", morder[i], mname[i], double(c.dbg[40 + morder[i]]) / mits);
#endif
  }
#endif
  const unsigned long long reason = host_load(&r->host->exited), done = host_load(&r->host->done);
  if (cycles_run) *cycles_run = int64_t(done);
  if (r->groups_posted & (1u << RG_FORCE)) e->rt_flags |= RT_TOUCHDOWN; // as shc_engine_set_tip_force (state_controller.cpp:1642)
  if (err != hipSuccess) return fail(SHC_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(err));
  if (direct_rc != SHC_OK) return fail(direct_rc, direct_msg + " (the loop was stopped after " + std::to_string(done) + " cycles)");
  if (reason == RESIDENT_EXIT_FAULT || host_load(&r->host->fault) != 0)
    return fail(SHC_ERR_HIP, "resident mode: a wavefront gave up waiting for the relay; restore the engine from a snapshot");
  if (host_load(&r->host->late_reads) != 0)
    return fail(SHC_ERR_TIMEOUT, "resident mode: a stream-ordered read (shc_engine_resident_get_joint_state_async) gave up waiting for its cycle");
  // (a loop that reached max_cycles exactly when everything published had run has done what a stop request asks for)
  if (!((reason == RESIDENT_EXIT_STOP || reason == RESIDENT_EXIT_MAX) && done == r->published))
    return fail(SHC_ERR_TIMEOUT, std::string("resident mode: the device loop had stopped by itself (") +
                                     (reason == RESIDENT_EXIT_IDLE ? "idle timeout" : reason == RESIDENT_EXIT_MAX ? "max_cycles" : "unknown") + ") after " +
                                     std::to_string(done) + " of " + std::to_string(r->published) + " published cycles");
  return SHC_OK;
}

// ================================================================================================ K cycles per launch, each with its own inputs
// shc_engine_step_k: what the node's loop does K times - callbacks deliver this iteration's inputs, StateController::loop runs one cycle, the
// desired joint state is published (src/main.cpp:106-131) - as ONE launch for batches of any size: the BATCH form of the resident loop kernel
// (ResidentArgs::batch_cycles).  State is loaded once, stays in registers / LDS for K cycles and is stored once; cycle k reads row k of the
// caller's K-deep input arrays where they lie (as a direct post of a bound set would deliver them) and writes its q / qd to slot k of a K-deep
// output ring.  No relay, no doorbell, nothing has to be co-resident; the launch is ordinary work on the engine's stream.
static int step_k_out_ring(shc_engine *e, int K) {
  const size_t need = size_t(K) * size_t(e->NJ) * size_t(e->n_slots) * 16;
  if (e->k_out && e->k_out_bytes >= need) return SHC_OK;
  if (e->k_out) { // (a launch on the split streams may still be writing the ring that is about to be replaced)
    const int rc = join_side(e);
    if (rc != SHC_OK) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    (void)hipFree(e->k_out);
    e->k_out = nullptr, e->k_out_bytes = 0;
  }
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->k_out), need));
  e->k_out_bytes = need;
  return SHC_OK;
}

// The configurations without a batch kernel - a manual-leg kernel (a leg has been toggled / planner mode: manual inputs and plans are held for the K
// cycles like the pose inputs), the runtime-flag families in a build without SHC_GENERIC_LOOP_FORMS - run the K cycles as the caller's loop would:
// row k of each K-deep array through its setter (which is what a direct post is defined against: test_step_k_is_byte_identical_to_single_launches),
// one shc_engine_step, q / qd into slot k of the output ring.  Same results, K launches instead of one.
__global__ void copy_joint_planes_kernel(double2 *dst, const double2 *src, int64_t count);
static int step_k_serial(shc_engine *e, int K, const shc_cycle_inputs *in) {
  const int64_t n = e->n, slot = int64_t(e->NJ) * e->n_slots; // double2 per ring slot: the planes that hold Q and QD (Fields: Q = 0, QD = NJ)
  for (int k = 0; k < K; ++k) {
    int rc = SHC_OK;
    if (in && in->linear_xy) rc = shc_engine_set_velocity(e, in->linear_xy + k * n * 2, in->angular + k * n, 1);
    if (rc == SHC_OK && in && in->imu_orientation_wxyz) rc = shc_engine_set_imu(e, in->imu_orientation_wxyz + k * n * 4, in->imu_angular_velocity + k * n * 3, 1);
    if (rc == SHC_OK && in && in->tip_force) rc = shc_engine_set_tip_force(e, in->tip_force + k * n * e->L * 3, 1);
    if (rc == SHC_OK && in && in->joint_effort) rc = shc_engine_set_joint_effort(e, in->joint_effort + k * n * e->L * e->NJ, 1);
    if (rc == SHC_OK) rc = shc_engine_step(e, 1);
    if (rc == SHC_OK) rc = join_side(e); // (a large batch steps as two halves on the split streams: the copy below is ordered after both)
    if (rc != SHC_OK) return rc;
    copy_joint_planes_kernel<<<dim3((unsigned)((slot + 255) / 256)), dim3(256), 0, e->stream>>>(reinterpret_cast<double2 *>(e->k_out) + k * slot,
                                                                                              reinterpret_cast<const double2 *>(e->st.legd), slot);
    HIP_TRY(hipGetLastError());
  }
  e->k_out_cycles = K;
  return SHC_OK;
}

extern "C" int shc_engine_step_k(shc_engine *e, int n_cycles, const shc_cycle_inputs *in) {
  SHC_BUSY_ONLY(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (n_cycles < 1 || n_cycles > 4096) return fail(SHC_ERR_INVALID_ARG, "shc_engine_step_k: 1 .. 4096 cycles per launch");
  if (e->starting_up) return fail(SHC_ERR_UNSUPPORTED, "shc_engine_step_k starts from a running engine (finish the start-up first)");
  {
    const int rc_remap = flush_step_remap(e);
    if (rc_remap != SHC_OK) return rc_remap;
  }
  unsigned mask = 0;
  if (in) {
    if (!in->on_device) return fail(SHC_ERR_INVALID_ARG, "shc_engine_step_k: the K-deep input arrays are device arrays (on_device = 1)");
    if ((in->linear_xy == nullptr) != (in->angular == nullptr)) return fail(SHC_ERR_INVALID_ARG, "linear_xy and angular are given together");
    if ((in->imu_orientation_wxyz == nullptr) != (in->imu_angular_velocity == nullptr)) return fail(SHC_ERR_INVALID_ARG, "the two IMU arrays are given together");
    if (in->pose_translation_velocity || in->pose_rotation_velocity || in->pose_reset_mode)
      return fail(SHC_ERR_UNSUPPORTED, "shc_engine_step_k carries velocity, IMU, tip force and joint effort; give pose inputs / reset modes with their setters before the call (held for the K cycles)");
    if (in->linear_xy) mask |= 1u << RG_VEL;
    if (in->imu_orientation_wxyz) mask |= 1u << RG_IMU;
    if (in->tip_force) mask |= 1u << RG_FORCE;
    if (in->joint_effort) mask |= 1u << RG_EFFORT;
  }
  HIP_TRY(hipSetDevice(e->device));
  const bool split = e->n_waves >= kSplitWaves && !(e->features & SHC_FEAT_SINGLE_STREAM);
  if (!split || ((mask & (1u << RG_EFFORT)) && !(e->rt_flags & RT_EFFORT_LIVE))) { // (one launch on the engine's stream / the parameter block is about to be re-uploaded)
    const int rc = join_side(e);
    if (rc != SHC_OK) return rc;
  }
  if (mask & (1u << RG_EFFORT)) { // Leg::calculateTipForce has something to filter from now on (as shc_engine_set_joint_effort)
    const int rc = effort_live(e);
    if (rc != SHC_OK) return rc;
  }
  ResidentFit fit{0, 0, 0, 0};
  {
    CycleLaunch a{e->st, e->d_consts, &e->cp, e->rt_flags, (e->features & SHC_FEAT_GENERIC_KERNEL) != 0, e->stream, 0, 64, 0, nullptr, &fit, 0};
#define CALL(L_, NJ_) shc_launch_cycle_##L_##_##NJ_(a)
    SHC_DISPATCH(e->L, e->NJ, CALL);
#undef CALL
  }
  if (size_t(e->NJ) * e->n_slots * 16 * size_t(n_cycles) >= (size_t(1) << 31))
    return fail(SHC_ERR_INVALID_ARG, "shc_engine_step_k: cycles x batch - the output ring must stay below 2 GiB (fewer cycles per launch)");
  {
    const int rc = step_k_out_ring(e, n_cycles);
    if (rc != SHC_OK) return rc;
  }
  if (!fit.batch || (e->features & SHC_FEAT_STEP_K_SERIAL)) return step_k_serial(e, n_cycles, in);
  e->plan_poser_tips_current = false;
  ResidentArgs A{};
  if (in) {
    A.bound[0][BND_LIN] = in->linear_xy, A.bound[0][BND_ANG] = in->angular;
    A.bound[0][BND_IMUQ] = in->imu_orientation_wxyz, A.bound[0][BND_IMUW] = in->imu_angular_velocity;
    A.bound[0][BND_FORCE] = in->tip_force, A.bound[0][BND_EFFORT] = in->joint_effort;
  }
  A.kstride[BND_LIN] = e->n * 2, A.kstride[BND_ANG] = e->n, A.kstride[BND_IMUQ] = e->n * 4, A.kstride[BND_IMUW] = e->n * 3;
  A.kstride[BND_FORCE] = e->n * e->L * 3, A.kstride[BND_EFFORT] = e->n * e->L * e->NJ;
  A.out = e->k_out;
  A.depth = n_cycles;
  A.max_cycles = unsigned(n_cycles);
  A.batch_cycles = unsigned(n_cycles);
  A.batch_mask = mask;
  A.n_waves = e->n_waves;
  A.idle_ticks = 0, A.ticks_per_ms = 100000;
  A.touchdown_threshold = e->params.touchdown_threshold;
  A.liftoff_threshold = e->params.liftoff_threshold;
  // Workgroups as shc_engine_step picks them; from kSplitWaves waves on the launch goes out as two halves on the two split streams, and - as there -
  // the halves are NOT joined between launches (one half's tail runs under the other half's full rounds, launch after launch): each half is ordered
  // after the engine's stream (where the caller's input rows were written) and after its own previous launch; the engine's stream is ordered after
  // both at the next call that needs it (join_side: every SHC_BUSY_GUARD entry point, shc_engine_get_step_k_joint_state among them).
  const int block = e->n_waves < 1536 ? 64 : 128;
  const int64_t wpb = block / 64;
  CycleLaunch a{e->st, e->d_consts, &e->cp, e->rt_flags, (e->features & SHC_FEAT_GENERIC_KERNEL) != 0, e->stream, unsigned((e->n_waves + wpb - 1) / wpb), block, 0, &A, nullptr, 0};
#define CALL(L_, NJ_) shc_launch_cycle_##L_##_##NJ_(a)
  if (!split) {
    SHC_DISPATCH(e->L, e->NJ, CALL);
    HIP_TRY(hipGetLastError());
  } else {
    if (!e->half_stream[0]) {
      const int rc = split_streams(e->device, e->half_stream);
      if (rc != SHC_OK) return rc;
      HIP_TRY(hipEventCreateWithFlags(&e->ev_main, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->ev_half[0], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->ev_half[1], hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(e->ev_main, e->stream));
    const int64_t half = ((e->n_waves / 2 + wpb - 1) / wpb) * wpb;
    for (int h = 0; h < 2; ++h) {
      HIP_TRY(hipStreamWaitEvent(e->half_stream[h], e->ev_main, 0));
      A.batch_wave0 = h ? half : 0;
      a.stream = e->half_stream[h];
      a.grid = unsigned(((h ? e->n_waves - half : half) + wpb - 1) / wpb);
      SHC_DISPATCH(e->L, e->NJ, CALL);
      HIP_TRY(hipGetLastError());
    }
    e->main_dirty = false;
    e->side_busy = true;
  }
#undef CALL
  e->k_out_cycles = n_cycles;
  if (mask & (1u << RG_FORCE)) e->rt_flags |= RT_TOUCHDOWN; // as shc_engine_set_tip_force (state_controller.cpp:1642)
  return SHC_OK;
}

// q / qd of cycle k (0 .. K - 1) of the latest shc_engine_step_k, from its output ring (stream-ordered after the launch).
extern "C" int shc_engine_get_step_k_joint_state(shc_engine *e, int k, double *q, double *qd, int on_device) {
  SHC_BUSY_GUARD(e);
  if (!e) return fail(SHC_ERR_INVALID_ARG, "engine is NULL");
  if (!e->k_out || k < 0 || k >= e->k_out_cycles) return fail(SHC_ERR_INVALID_ARG, "shc_engine_get_step_k_joint_state: cycle 0 .. K - 1 of the latest shc_engine_step_k");
  HIP_TRY(hipSetDevice(e->device));
  const double *slot = e->k_out + size_t(k) * size_t(e->NJ) * e->n_slots * 2;
  const int64_t threads = e->n * e->L;
  for (int which = 0; which < 2; ++which) {
    double *dst = which ? qd : q;
    if (!dst) continue;
    double *d = on_device ? dst : e->d_stage;
    if (!on_device && size_t(threads) * e->NJ * 8 > e->stage_bytes) return fail(SHC_ERR_INVALID_ARG, "staging buffer too small");
    gather_leg_kernel<<<dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, e->stream>>>(d, slot, e->n_slots, e->n, e->L, e->NJ, which ? LEG_FIELD(e, QD) : LEG_FIELD(e, Q));
    HIP_TRY(hipGetLastError());
    if (!on_device) {
      HIP_TRY(hipMemcpyAsync(dst, d, size_t(threads) * e->NJ * 8, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipStreamSynchronize(e->stream));
    }
  }
  return SHC_OK;
}

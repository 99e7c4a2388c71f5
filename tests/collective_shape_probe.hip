// collective_shape_probe.hip - TEST INFRASTRUCTURE (tests/test_gpu_resident.py::test_collective_shaped_kernel_next_to_a_live_loop).
// A stand-in for the shape of an RCCL collective kernel: a few dozen workgroups of 256-512 threads with tens of KB of LDS each, whose
// blocks spin on each other's flags (every block must be co-resident, as the channels of a ring are).  Launched on the engine's stream
// behind shc_engine_resident_get_joint_state_async while the persistent loop is alive, it tells whether such a kernel is scheduled and
// makes progress next to the loop - a one-GPU box cannot host two RCCL ranks, so the real ring cannot be rehearsed here.  Every wait is
// bounded on the device (wall clock): a starved probe reports failure, it never hangs the GPU.
//   build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libcollective_shape_probe.so collective_shape_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ProbeResult {
  unsigned long long rounds_done;   // min over blocks (written by block 0 at the end)
  unsigned long long timed_out;     // blocks that gave up waiting for a peer
  unsigned long long checksum;      // sum of the payload every block moved through LDS (proves the LDS is really its own)
  unsigned long long ticks;         // wall-clock ticks (100 MHz) block 0 spent inside the kernel
};

__global__ void collective_shape_kernel(unsigned long long *flags, ProbeResult *res, const double *src, double *dst, int64_t n_doubles, int rounds,
                                        unsigned long long timeout_ticks) {
  extern __shared__ double lds[];
  const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  const unsigned long long t0 = wall_clock64();
  unsigned long long local_sum = 0;
  bool gave_up = false;
  for (int r = 1; r <= rounds && !gave_up; ++r) {
    // "send": this block's slice of the payload through LDS into dst (what a ring step does with its channel buffer)
    const int64_t per = (n_doubles + nb - 1) / nb, lo = int64_t(b) * per, hi = lo + per < n_doubles ? lo + per : n_doubles;
    for (int64_t i = lo + t; i < hi; i += nt) {
      lds[t] = src[i];
      dst[i] = lds[t] + double(r);
      local_sum += (unsigned long long)(i & 7);
    }
    __threadfence();
    __syncthreads();
    if (t == 0) __hip_atomic_store(&flags[b], (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // "wait for the peers": every block spins until every other block has reached this round
    if (t < 64) {
      for (int peer = t; peer < nb && !gave_up; peer += 64) {
        while (__hip_atomic_load(&flags[peer], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r) {
          if (wall_clock64() - t0 > timeout_ticks) {
            gave_up = true;
            break;
          }
          __builtin_amdgcn_s_sleep(4);
        }
      }
    }
    gave_up = __syncthreads_or(gave_up ? 1 : 0) != 0;
  }
  if (t == 0) {
    if (gave_up) atomicAdd(&res->timed_out, 1ull);
    atomicAdd(&res->checksum, local_sum);
    if (b == 0) {
      unsigned long long m = ~0ull;
      for (int p = 0; p < nb; ++p) {
        const unsigned long long v = __hip_atomic_load(&flags[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        m = v < m ? v : m;
      }
      res->rounds_done = m;
      res->ticks = wall_clock64() - t0;
    }
  }
}

extern "C" {
// probe_prepare allocates the probe's buffers, probe_launch queues the kernel on `stream`, probe_stream_wait polls the stream (bounded),
// probe_result copies the result out and frees the buffers (call it after the loop has ended: hipFree may synchronise the device).
struct Probe {
  unsigned long long *flags;
  ProbeResult *res;
  double *src, *dst;
  int64_t n;
};
static Probe g{};

// (allocation and clearing happen BEFORE the loop starts: nothing that might synchronise the device runs while it is alive)
int probe_prepare(int blocks, int64_t n_doubles, int lds_bytes) {
  if (g.flags) return -1;
  if (hipMalloc(&g.flags, sizeof(unsigned long long) * blocks) != hipSuccess) return 1;
  if (hipMalloc(&g.res, sizeof(ProbeResult)) != hipSuccess) return 1;
  if (hipMalloc(&g.src, n_doubles * 8) != hipSuccess || hipMalloc(&g.dst, n_doubles * 8) != hipSuccess) return 1;
  g.n = n_doubles;
  if (hipMemset(g.flags, 0, sizeof(unsigned long long) * blocks) != hipSuccess) return 2;
  if (hipMemset(g.res, 0, sizeof(ProbeResult)) != hipSuccess) return 2;
  if (hipMemset(g.src, 0, n_doubles * 8) != hipSuccess) return 2;
  if (hipFuncSetAttribute((const void *)collective_shape_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 3;
  return hipDeviceSynchronize() == hipSuccess ? 0 : 4;
}
int probe_launch(void *stream, int blocks, int threads, int lds_bytes, int rounds, int timeout_ms) {
  if (!g.flags) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_doubles = g.n;
  collective_shape_kernel<<<dim3(blocks), dim3(threads), lds_bytes, s>>>(g.flags, g.res, g.src, g.dst, n_doubles, rounds, (unsigned long long)timeout_ms * 100000ull);
  return hipGetLastError() == hipSuccess ? 0 : 4;
}
// 1: the stream has drained, 0: not within timeout_ms
int probe_stream_wait(void *stream, int timeout_ms) {
  for (int i = 0; i < timeout_ms * 10; ++i) {
    if (hipStreamQuery((hipStream_t)stream) == hipSuccess) return 1;
    timespec ts{0, 100000};
    nanosleep(&ts, nullptr);
  }
  return 0;
}
int probe_result(unsigned long long *out4) {
  if (!g.flags) return -1;
  ProbeResult r{};
  const hipError_t e = hipMemcpy(&r, g.res, sizeof r, hipMemcpyDeviceToHost);
  out4[0] = r.rounds_done, out4[1] = r.timed_out, out4[2] = r.checksum, out4[3] = r.ticks;
  (void)hipFree(g.flags), (void)hipFree(g.res), (void)hipFree(g.src), (void)hipFree(g.dst);
  g = Probe{};
  return e == hipSuccess ? 0 : 1;
}
}

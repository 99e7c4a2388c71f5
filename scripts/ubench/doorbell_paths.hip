// Development micro-benchmark: round trip host -> device -> host of one doorbell tick, for the places the
// doorbell word and the progress word can live.  One persistent wavefront spins on the doorbell and answers into the ack word;
// the host measures store -> answer seen.
//   doorbell: (a) pinned host memory, device-mapped - the device polls it across PCIe (the resident loop's form today)
//             (b) fine-grained device memory written by the host through the BAR (hipExtMallocWithFlags) - the device polls its own memory
//             (c) ordinary hipMalloc memory written by the host through the BAR, if the platform maps it
//   ack:      (1) pinned host memory (the device writes across PCIe, the host polls its own memory: today's form)
//             (2) fine-grained device memory polled by the host across the BAR
// Each candidate runs in a child process: a platform without host access to device memory kills the child, not the probe.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <sys/wait.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
__global__ void echo_kernel(const u64 *doorbell, u64 *ack, u64 rounds, int system_scope) {
  u64 seen = 0;
  for (u64 spins = 0; seen < rounds && spins < (1ull << 31); ++spins) {
    const u64 v = system_scope ? __hip_atomic_load(doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                               : __hip_atomic_load(doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v > seen) {
      seen = v;
      __hip_atomic_store(ack, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static int run(int db_kind, int ack_kind) {
  u64 *db_host = nullptr, *db_dev = nullptr, *ack_host = nullptr, *ack_dev = nullptr;
  auto place = [&](int kind, u64 **host, u64 **dev) -> bool {
    if (kind == 0) {
      if (hipHostMalloc(reinterpret_cast<void **>(host), 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return false;
      return hipHostGetDevicePointer(reinterpret_cast<void **>(dev), *host, 0) == hipSuccess;
    }
    if (kind == 1) {
      if (hipExtMallocWithFlags(reinterpret_cast<void **>(dev), 64, hipDeviceMallocFinegrained) != hipSuccess) return false;
    } else {
      if (hipMalloc(reinterpret_cast<void **>(dev), 64) != hipSuccess) return false;
    }
    if (hipMemset(*dev, 0, 64) != hipSuccess) return false;
    hipDeviceSynchronize();
    *host = *dev; // the host dereferences the device address (large-BAR platforms map it)
    return true;
  };
  if (!place(db_kind, &db_host, &db_dev) || !place(ack_kind, &ack_host, &ack_dev)) {
    printf("RESULT doorbell %d ack %d: allocation refused\n", db_kind, ack_kind);
    return 0;
  }
  if (db_kind == 0) *db_host = 0;
  if (ack_kind == 0) *ack_host = 0;
  const u64 rounds = 2000;
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  echo_kernel<<<1, 1, 0, s>>>(db_dev, ack_dev, rounds, db_kind == 0 ? 1 : (db_kind == 1 ? 1 : 0));
  std::vector<double> us;
  for (u64 r = 1; r <= rounds; ++r) {
    auto t0 = std::chrono::steady_clock::now();
    __atomic_store_n(db_host, r, __ATOMIC_RELEASE);
    while (__atomic_load_n(ack_host, __ATOMIC_ACQUIRE) < r) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
        printf("RESULT doorbell %d ack %d: no answer (round %llu)\n", db_kind, ack_kind, r);
        fflush(stdout);
        _exit(3);
      }
    }
    us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  hipStreamSynchronize(s);
  std::sort(us.begin(), us.end());
  const char *names[] = {"pinned host memory (device polls across PCIe)", "fine-grained device memory (host writes through the BAR)", "hipMalloc device memory (host writes through the BAR)"};
  const char *acks[] = {"pinned host memory", "fine-grained device memory (host polls through the BAR)", "hipMalloc device memory (host polls through the BAR)"};
  printf("RESULT doorbell in %s, answer in %s: round trip median %.2f us  min %.2f  p90 %.2f\n", names[db_kind], acks[ack_kind], us[us.size() / 2], us.front(),
         us[us.size() * 9 / 10]);
  return 0;
}

int main() {
  const int cases[][2] = {{0, 0}, {1, 0}, {2, 0}, {0, 1}, {1, 1}};
  for (auto &c : cases) {
    fflush(stdout);
    const pid_t pid = fork();
    if (pid == 0) {
      const int rc = run(c[0], c[1]);
      fflush(stdout);
      _exit(rc);
    }
    int status = 0;
    waitpid(pid, &status, 0);
    if (WIFSIGNALED(status)) printf("RESULT doorbell %d ack %d: the child died with signal %d (no host access to that memory)\n", c[0], c[1], WTERMSIG(status));
  }
  return 0;
}

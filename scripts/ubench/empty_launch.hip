// Development micro-benchmark: back-to-back launch cost of an empty kernel and of a one-wave load/store kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_kernel() {}
__global__ void touch_kernel(double *p) { p[threadIdx.x] += 1.0; }
int main() {
  double *d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 200; ++i) { if (mode) touch_kernel<<<1, 64>>>(d); else empty_kernel<<<1, 64>>>(); }
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    const int reps = 5000;
    for (int i = 0; i < reps; ++i) { if (mode) touch_kernel<<<1, 64>>>(d); else empty_kernel<<<1, 64>>>(); }
    hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%s: %.2f us per back-to-back launch\n", mode ? "one-wave load+store kernel" : "empty kernel", us);
  }
  return 0;
}

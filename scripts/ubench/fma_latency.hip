// Development micro-benchmark: dependent-issue latency of FP64 VALU ops on one wavefront (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void fma_chain(double *out, long long *ticks, int iters, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], b, a);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int CHAINS>
__global__ void fma_chain_masked(double *out, long long *ticks, int iters, double a, double b, int active) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  if (int(threadIdx.x) < active) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], b, a);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ void div_chain(double *out, long long *ticks, int iters, double a, double b) {
  double x = a + threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) x = b / x + a;
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ void lds_chain(double *out, long long *ticks, int iters) {
  __shared__ int idx[256];
  for (int i = threadIdx.x; i < 256; i += 64) idx[i] = (i * 7 + 1) & 255;
  __syncthreads();
  int k = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters * 16; ++i) k = idx[k];
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = k;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ void shfl_chain(double *out, long long *ticks, int iters) {
  int k = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters * 16; ++i) k = __shfl(k, (k + 1) & 63, 64);
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = k;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
  double *out; long long *t; hipMalloc(&out, 64 * 8); hipMalloc(&t, 8);
  long long h; const int iters = 1000; const double n = iters * 16.0;
#define RUN(K, label, ops) for (int r = 0; r < 2; ++r) { K; hipDeviceSynchronize(); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); } printf("%-28s %7.2f ticks per op\n", label, h / (n * ops));
  RUN((fma_chain<1><<<1, 64>>>(out, t, iters, 0.5, 0.999)), "fma f64, 1 chain", 1)
  RUN((fma_chain<2><<<1, 64>>>(out, t, iters, 0.5, 0.999)), "fma f64, 2 chains", 2)
  RUN((fma_chain<4><<<1, 64>>>(out, t, iters, 0.5, 0.999)), "fma f64, 4 chains", 4)
  RUN((fma_chain_masked<4><<<1, 64>>>(out, t, iters, 0.5, 0.999, 64)), "fma 4 chains, 64 lanes", 4)
  RUN((fma_chain_masked<4><<<1, 64>>>(out, t, iters, 0.5, 0.999, 32)), "fma 4 chains, 32 lanes", 4)
  RUN((fma_chain_masked<4><<<1, 64>>>(out, t, iters, 0.5, 0.999, 16)), "fma 4 chains, 16 lanes", 4)
  RUN((fma_chain_masked<4><<<1, 64>>>(out, t, iters, 0.5, 0.999, 10)), "fma 4 chains, 10 lanes", 4)
  RUN((div_chain<<<1, 64>>>(out, t, iters, 0.5, 1.7)), "div f64 + add, 1 chain", 1)
  RUN((lds_chain<<<1, 64>>>(out, t, iters)), "dependent ds_read_b32", 1)
  RUN((shfl_chain<<<1, 64>>>(out, t, iters)), "dependent ds_bpermute", 1)
  return 0;
}

// Where do the wavefronts of a small grid land?  For every wave: XCC, SE, CU, SIMD (s_getreg HW_ID / XCC_ID), for 64- and
// 128-thread workgroups - the question behind the two-wave (front / back) resident kernel: do the two waves of a 128-thread
// workgroup get two different SIMDs of their CU?
// build: hipcc --offload-arch=gfx950 -O2 -o wave_placement wave_placement.hip ; run: ./wave_placement [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void probe(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // stay resident for a while so that every workgroup of the grid is placed while the others still run
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    out[2 * w] = hw;
    out[2 * w + 1] = xcc;
  }
}

int main(int argc, char **argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 410;
  for (int block : {64, 128}) {
    const int waves = wgs * block / 64;
    unsigned *d;
    hipMalloc(&d, waves * 8);
    hipMemset(d, 0xff, waves * 8);
    probe<<<wgs, block>>>(d, 20000); // 200 us at 100 MHz
    hipDeviceSynchronize();
    std::vector<unsigned> h(waves * 2);
    hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu;
    int same_simd_pairs = 0, same_cu_pairs = 0;
    for (int w = 0; w < waves; ++w) {
      const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const unsigned cu_key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
      per_cu[cu_key]++;
      per_simd[(cu_key << 2) | simd]++;
      if (block == 128 && (w & 1)) {
        const unsigned hw0 = h[2 * (w - 1)], xcc0 = h[2 * (w - 1) + 1] & 0xf;
        if (xcc0 == xcc && ((hw0 ^ hw) & 0xff00) == 0) {
          same_cu_pairs++;
          if (((hw0 >> 4) & 3) == simd) same_simd_pairs++;
        }
      }
    }
    std::map<int, int> hist_simd, hist_cu;
    for (auto &kv : per_simd) hist_simd[kv.second]++;
    for (auto &kv : per_cu) hist_cu[kv.second]++;
    printf("%d workgroups x %d threads = %d waves: %zu CUs, %zu SIMDs in use\n", wgs, block, waves, per_cu.size(), per_simd.size());
    printf("  waves per SIMD histogram:");
    for (auto &kv : hist_simd) printf("  %d waves: %d SIMDs", kv.first, kv.second);
    printf("\n  waves per CU histogram:");
    for (auto &kv : hist_cu) printf("  %d waves: %d CUs", kv.first, kv.second);
    printf("\n");
    if (block == 128) printf("  pairs (wave 0, wave 1 of a workgroup): %d on the same CU, %d of them on the same SIMD\n", same_cu_pairs, same_simd_pairs);
    hipFree(d);
  }
  return 0;
}

// ub_simd_of_wave.hip -- which SIMD does wave w of a 14- / 16-wave workgroup with 160 KB of LDS run on?  (HW_ID: SIMD_ID bits 5:4)
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/ub_simd_of_wave.hip -o scratch/ub_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
  extern __shared__ char smem[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
  if (threadIdx.x == 1023) smem[0] = 1;
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 16 * 4);
  for (int nt : {896, 1024, 384}) {
    hipMemset(d, 0xff, 256 * 16 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(256), dim3(nt), 150 * 1024, 0, d);
    hipDeviceSynchronize();
    std::vector<unsigned> h(256 * 16); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    printf("threads %d\n", nt);
    for (int b : {0, 1, 2, 100, 255}) { printf("  block %3d simd of waves:", b); for (int w = 0; w < nt / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3u); printf("   cu %u se %u\n", (h[b * 16] >> 8) & 15u, (h[b * 16] >> 13) & 7u); }
    int hist[16][4] = {}; for (int b = 0; b < 256; ++b) for (int w = 0; w < nt / 64; ++w) hist[w][(h[b * 16 + w] >> 4) & 3u]++;
    for (int w = 0; w < nt / 64; ++w) printf("  wave %2d: simd0 %3d simd1 %3d simd2 %3d simd3 %3d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  }
  return 0;
}

// Which SIMD does wave w of a 1 024-thread workgroup run on?  (HW_ID of gfx9: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8],
// sh_id [12], se_id [15:13] ...; XCC_ID is a separate register on gfx94x / gfx950.)  One workgroup per CU (100 KB of LDS), 256 of them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void k(unsigned* out) {
  extern __shared__ char smem[];
  unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  smem[threadIdx.x] = 0;
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = id; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 16 * 2 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int nt : {1024, 960}) {
    hipMemset(d, 0, 256 * 16 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(256), dim3(nt), 100 * 1024, 0, d);
    std::vector<unsigned> h(256 * 16 * 2); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    printf("threads %d\n", nt);
    int hist[16][4] = {};
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nt / 64; ++w) hist[w][(h[(b * 16 + w) * 2] >> 4) & 3]++;
    for (int w = 0; w < nt / 64; ++w) printf("  wave %2d: simd 0..3 counts %3d %3d %3d %3d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b : {0, 1, 9, 100}) { printf("  block %3d:", b); for (int w = 0; w < nt / 64; ++w) { unsigned v = h[(b * 16 + w) * 2]; printf(" s%u/w%u", (v >> 4) & 3, v & 15); } printf("  cu %u se %u xcc %u\n", (h[b * 32] >> 8) & 15, (h[b * 32] >> 13) & 7, h[b * 32 + 1] & 15); }
  }
  return 0;
}

// how fast does a dependent VALU chain on ONE wave run while the other waves of its SIMD do heavy VALU work?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int MODE>   // other waves: 0 idle (exit), 1 mad_u64_u32 loop (quarter rate), 2 v_add loop (full rate), 3 LDS traffic
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, int prio, int chain_wave) {
  __shared__ int lds[4096];
  const int w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  if (w == chain_wave) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    int x = threadIdx.x, d = lds[threadIdx.x & 63];
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 20; ++u) { x = max(x - d, 0) + min(d + u, 100 - x); }     // 5 dependent-ish ops per step
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = x; }
  } else {
    if (MODE == 0) return;
    unsigned a = threadIdx.x * 2654435761u, b = blockIdx.x + 12345u; float f = threadIdx.x;
    // run about as long as the chain wave: the host picks iters so that everyone overlaps
#pragma unroll 1
    for (int i = 0; i < iters * 6; ++i) {
      if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < 10; ++u) { unsigned long long p = (unsigned long long)a * 0xD2511F53u; a = (unsigned)(p >> 32) ^ b; b = (unsigned)p + u; }
      } else if (MODE == 2) {
#pragma unroll
        for (int u = 0; u < 20; ++u) { a = a + b; b = b ^ a; }
      } else {
#pragma unroll
        for (int u = 0; u < 10; ++u) { a += lds[(a + u) & 4095]; lds[(b + u * 64 + threadIdx.x) & 4095] = a; }
      }
    }
    if (a == 0x12345 && b == 77 && f == 3.f) out[1000000] = a;
  }
}
template <int MODE> void run(const char* name, int nblocks, int prio) {
  unsigned long long* d; hipMalloc(&d, 1 << 24);
  const int iters = 50;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(nblocks), dim3(256), 0, 0, d, iters, prio, 0); hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(nblocks * 2); hipMemcpy(h.data(), d, nblocks * 16, hipMemcpyDeviceToHost);
  double s = 0; for (int b = 0; b < nblocks; ++b) s += h[b * 2];
  printf("%-28s blocks %5d prio %d: %.1f cycles per chain step (5 VALU), %.1f per instruction\n", name, nblocks, prio, s / nblocks / (iters * 20), s / nblocks / (iters * 20 * 5));
  hipFree(d);
}
int main() {
  for (int nb : {256, 1024}) for (int prio : {0, 1}) {
    run<0>("others idle", nb, prio);
    run<1>("others mad_u64_u32", nb, prio);
    run<2>("others v_add", nb, prio);
    run<3>("others LDS traffic", nb, prio);
  }
  return 0;
}

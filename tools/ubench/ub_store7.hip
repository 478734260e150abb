// from FLAT_T (one 16-byte store per thread, 80 000 short-lived workgroups: 0.83 of 8 TB/s) towards a persistent grid: K stores per
// thread with (a) the block's stores contiguous [block-major], (b) grid-stride.  Where does the write bandwidth drop?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int K, bool STRIDE>
__global__ __launch_bounds__(256) void k(char* buf, size_t bytes) {
  const size_t nthreads = (size_t)gridDim.x * 256, t = (size_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    // STRIDE: iteration j covers [j * nthreads * 16, ...): consecutive threads consecutive pieces; else the block owns K * 4 KB contiguous
    const size_t off = STRIDE ? ((size_t)j * nthreads + t) * 16 : (((size_t)blockIdx.x * K + j) * 256 + threadIdx.x) * 16;
    if (off < bytes) *(float4*)(buf + off) = make_float4(1.f, 2.f, 3.f, (float)j);
  }
}
template <int K, bool STRIDE> void run() {
  const size_t bytes = (size_t)400 * 4096 * 9 * 22;
  std::vector<char*> bufs(2);
  for (auto& b : bufs) hipMalloc(&b, bytes);
  const int blocks = (int)((bytes / 16 + 256 * (size_t)K - 1) / (256 * (size_t)K));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<K, STRIDE>), dim3(blocks), dim3(256), 0, 0, bufs[i % 2], bytes);
  hipEventRecord(e0);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<K, STRIDE>), dim3(blocks), dim3(256), 0, 0, bufs[i % 2], bytes);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("K = %3d stores per thread, %s, %6d blocks: %8.2f us  %.3f of 8 TB/s\n", K, STRIDE ? "grid-stride " : "block-major ", blocks, us, bytes / (us * 1e-6) / 8e12);
  for (auto b : bufs) hipFree(b);
}
int main() {
  run<1, false>(); run<2, false>(); run<4, false>(); run<8, false>(); run<16, false>(); run<64, false>();
  run<2, true>(); run<4, true>(); run<8, true>(); run<16, true>(); run<64, true>();
  return 0;
}

// grid-stride fills with different PIECE sizes: 1152 blocks x 192 store threads (the rollout kernel's shape) write one flat
// buffer of the T=400 trajectory's size; iteration i, block b writes piece (i * nblocks + b) of P bytes.  How small can the
// contiguous piece per block be before the write bandwidth drops?  Also: the same with K independent planes written at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
static __device__ __forceinline__ int xcd_block() {
  const unsigned n = gridDim.x, x = blockIdx.x & 7u, q = n >> 3, rem = n & 7u;
  return (int)(x * q + (x < rem ? x : rem) + (blockIdx.x >> 3));
}
template <int P, int PLANES, bool REMAP>
__global__ __launch_bounds__(256) void k(char* buf, size_t bytes) {
  const int tid = (int)threadIdx.x - 64;                 // 192 store threads
  if (tid < 0) return;
  const size_t nb = gridDim.x, bid = REMAP ? (size_t)xcd_block() : (size_t)blockIdx.x;
  const size_t plane_bytes = bytes / PLANES / 16 * 16;
  constexpr int LPP = P / 16;                             // lanes per piece
  const size_t n_iter = plane_bytes / ((size_t)P * nb);   // pieces per block and plane
  // the 192 threads cover 192 / LPP pieces per pass: consecutive iterations of the block, planes innermost
  const int sub = tid / LPP, ln = tid - sub * LPP, per_pass = 192 / LPP;
  for (size_t it = sub; it < n_iter * PLANES; it += per_pass) {
    const size_t i = it / PLANES, pl = it - i * PLANES;
    char* p = buf + pl * plane_bytes + (i * nb + bid) * P + (size_t)ln * 16;
    *(float4*)p = make_float4(1.f, 2.f, 3.f, (float)it);
  }
}
template <int P, int PLANES, bool REMAP> void run(int nbuf) {
  const size_t bytes = (size_t)400 * 4096 * 9 * 22;
  std::vector<char*> bufs(nbuf);
  for (auto& b : bufs) hipMalloc(&b, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2 * nbuf; ++i) hipLaunchKernelGGL((k<P, PLANES, REMAP>), dim3(1152), dim3(256), 0, 0, bufs[i % nbuf], bytes);
  hipEventRecord(e0);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<P, PLANES, REMAP>), dim3(1152), dim3(256), 0, 0, bufs[i % nbuf], bytes);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("piece %5d B  planes %d  remap %d: %8.2f us  %.2f TB/s = %.3f of 8 TB/s\n", P, PLANES, (int)REMAP, us, bytes / (us * 1e-6) / 1e12, bytes / (us * 1e-6) / 8e12);
  for (auto b : bufs) hipFree(b);
}
int main() {
  run<3072, 1, false>(2); run<1024, 1, false>(2); run<384, 1, false>(2); run<128, 1, false>(2); run<64, 1, false>(2);
  run<128, 1, true>(2); run<384, 1, true>(2);
  run<128, 5, false>(2); run<384, 5, false>(2); run<128, 5, true>(2); run<384, 5, true>(2); run<1024, 5, true>(2);
  return 0;
}

// grid-stride fill (384-byte pieces), launch-shape sweep: how many store waves per CU does the write bandwidth need?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NT, int IDLE, int UNROLL>
__global__ __launch_bounds__(NT) void k(char* buf, size_t bytes) {
  const int tid = (int)threadIdx.x - IDLE;               // NT - IDLE store threads
  if (tid < 0) return;
  constexpr int NW = NT - IDLE, P = 384, LPP = P / 16, PER = NW / LPP;
  const size_t nb = gridDim.x, bid = blockIdx.x;
  const size_t n_iter = bytes / ((size_t)P * nb);
  const int sub = tid / LPP, ln = tid - sub * LPP;
  if (sub >= PER) return;
  size_t it = sub;
  for (; it + (UNROLL - 1) * PER < n_iter; it += UNROLL * PER) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) *(float4*)(buf + ((it + u * PER) * nb + bid) * P + (size_t)ln * 16) = make_float4(1.f, 2.f, 3.f, (float)it);
  }
  for (; it < n_iter; it += PER) *(float4*)(buf + (it * nb + bid) * P + (size_t)ln * 16) = make_float4(1.f, 2.f, 3.f, (float)it);
}
template <int NT, int IDLE, int UNROLL> void run(int blocks) {
  const size_t bytes = (size_t)400 * 4096 * 9 * 22;
  std::vector<char*> bufs(2);
  for (auto& b : bufs) hipMalloc(&b, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<NT, IDLE, UNROLL>), dim3(blocks), dim3(NT), 0, 0, bufs[i % 2], bytes);
  hipEventRecord(e0);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<NT, IDLE, UNROLL>), dim3(blocks), dim3(NT), 0, 0, bufs[i % 2], bytes);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("blocks %5d  threads %4d (%4d storing)  unroll %d: %8.2f us  %.3f of 8 TB/s\n", blocks, NT, NT - IDLE, UNROLL, us, bytes / (us * 1e-6) / 8e12);
  for (auto b : bufs) hipFree(b);
}
int main() {
  run<256, 64, 1>(1152); run<256, 64, 4>(1152); run<256, 0, 1>(1152); run<512, 0, 1>(1152); run<1024, 0, 1>(1152);
  run<256, 64, 1>(2304); run<256, 0, 1>(2304); run<256, 0, 1>(4608); run<256, 0, 1>(9216); run<256, 0, 1>(36864);
  run<256, 0, 4>(4608); run<1024, 0, 4>(1024); run<256, 0, 1>(768); run<320, 64, 1>(768); run<512, 0, 1>(768);
  return 0;
}

// store-pattern microbenchmark: the trajectory stores of the SC64 B=4096 T=100 rollout, no compute
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define B 4096
#define S 9
#define T 100
#define TC 20
static __device__ __forceinline__ int xcd_block() {
  const unsigned n = gridDim.x, x = blockIdx.x & 7u, q = n >> 3, rem = n & 7u;
  return (int)(x * q + (x < rem ? x : rem) + (blockIdx.x >> 3));
}
// MODE 0: as the kernel does (unit = row x 4 pairs: 3 strided obs float4 + reward + action float4 + 2 flag words)
// MODE 1: obs written as contiguous 16-byte pieces across lanes (transposed), rest as MODE 0
// MODE 2: every plane written fill-like: consecutive lanes -> consecutive 16 bytes within the block's row segments
// MODE 3: like 0 but chunks iterate with a delay loop between (spread stores in time)
template <int MODE, int NT, int EPB>
__global__ __launch_bounds__(NT) void k(float* obs, float* rew, float* act, unsigned char* tru, unsigned char* ter, int spin) {
  constexpr int G = EPB * S, G4 = G / 4;
  const int bid = xcd_block();
  const long long total = (long long)B * S, g_base = (long long)bid * G;
  const int tid = threadIdx.x;
  volatile float sink = 0.f;
  for (int c = 0; c < T / TC; ++c) {
    float f = 1.f;
    for (int i = 0; i < spin; ++i) f = f * 1.0001f + 0.5f;
    const long long row0 = (long long)c * TC * total + g_base;
    if (MODE == 0 || MODE == 3 || MODE == 1) {
      for (int u = tid; u < TC * G4; u += NT) {
        const int r = u / G4, gl0 = (u - r * G4) * 4;
        const unsigned eo = (unsigned)r * (unsigned)total + gl0;
        const float4 v = make_float4(f, f, f, (float)u);
        if (MODE != 1) {
          float4* po = (float4*)((char*)(obs + row0 * 3) + (size_t)(eo * 12u));
          po[0] = v; po[1] = v; po[2] = v;
        }
        *(float4*)((char*)(rew + row0) + (size_t)(eo * 4u)) = v;
        *(float4*)((char*)(act + row0) + (size_t)(eo * 4u)) = v;
        *(unsigned*)((char*)(tru + row0) + (size_t)eo) = u;
        *(unsigned*)((char*)(ter + row0) + (size_t)eo) = 0u;
      }
      if (MODE == 1) {
        constexpr int PR = G * 3 / 4;                 // 16-byte pieces per row
        for (int q = tid; q < TC * PR; q += NT) {
          const int r = q / PR, pc = q - r * PR;
          *(float4*)((char*)(obs + row0 * 3) + (size_t)((unsigned)r * (unsigned)total * 12u) + pc * 16) = make_float4(f, f, f, (float)q);
        }
      }
    } else {
      constexpr int PR = G * 3 / 4, PW = G / 4, PF = G / 4;   // pieces per row: obs 16 B, rew/act 16 B, flags 4 B
      for (int q = tid; q < TC * PR; q += NT) { const int r = q / PR, pc = q - r * PR;
        *(float4*)((char*)(obs + row0 * 3) + (size_t)((unsigned)r * (unsigned)total * 12u) + pc * 16) = make_float4(f, f, f, (float)q); }
      for (int q = tid; q < TC * PW; q += NT) { const int r = q / PW, pc = q - r * PW;
        *(float4*)((char*)(rew + row0) + (size_t)((unsigned)r * (unsigned)total * 4u) + pc * 16) = make_float4(f, f, f, (float)q);
        *(float4*)((char*)(act + row0) + (size_t)((unsigned)r * (unsigned)total * 4u) + pc * 16) = make_float4(f, f, f, (float)q); }
      for (int q = tid; q < TC * PF; q += NT) { const int r = q / PF, pc = q - r * PF;
        *(unsigned*)((char*)(tru + row0) + (size_t)((unsigned)r * (unsigned)total) + pc * 4) = q;
        *(unsigned*)((char*)(ter + row0) + (size_t)((unsigned)r * (unsigned)total) + pc * 4) = 0u; }
    }
    sink = f;
  }
}
template <int MODE, int NT, int EPB> void run(const char* name, int spin) {
  const size_t n = (size_t)T * B * S;
  float *obs, *rew, *act; unsigned char *tru, *ter;
  hipMalloc(&obs, n * 12); hipMalloc(&rew, n * 4); hipMalloc(&act, n * 4); hipMalloc(&tru, n); hipMalloc(&ter, n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, NT, EPB>), dim3(B / EPB), dim3(NT), 0, 0, obs, rew, act, tru, ter, spin);
  hipEventRecord(e0);
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k<MODE, NT, EPB>), dim3(B / EPB), dim3(NT), 0, 0, obs, rew, act, tru, ter, spin);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s spin %5d: %7.2f us/launch  %.2f TB/s\n", name, spin, ms * 10.f, (double)n * 22 / (ms * 1e-5) / 1e12 * 1e-3 * 1e3 / 1e3);
  hipFree(obs); hipFree(rew); hipFree(act); hipFree(tru); hipFree(ter);
}
int main() {
  for (int spin : {0, 2000}) {
    run<0, 256, 4>("as kernel (epb 4, 256 thr)", spin);
    run<1, 256, 4>("obs transposed contiguous", spin);
    run<2, 256, 4>("all planes piece-contiguous", spin);
    run<0, 512, 8>("as kernel, epb 8, 512 thr", spin);
    run<2, 512, 8>("piece-contiguous, epb 8", spin);
    run<0, 256, 16>("as kernel, epb 16, 256 thr (256 blocks)", spin);
    run<2, 256, 16>("piece-contiguous, epb 16", spin);
  }
  return 0;
}

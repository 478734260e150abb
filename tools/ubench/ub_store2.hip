// store-pattern microbenchmark, round 3: the trajectory stores of the SC64 B=4096 T=100 rollout (no compute) for
// different trajectory LAYOUTS, rotating over NBUF buffer sets (> 256 MB in total: HBM, not the Infinity Cache).
//   TM    time-major [T][B][S] as phx_sc_rollout_fast_kernel writes it (obs pieces contiguous per tile row, reward /
//         action one float4 and flags one u32 per 4-pair unit)
//   EM    env-major [B][T][S]: every env's chunk is one contiguous run per plane (2160 / 720 / 720 / 180 / 180 bytes)
//   REC   env-major with ONE interleaved 24-byte record per (step, shop): obs 12 | action 4 | reward 4 | flags 4
//   FILL  every block writes one contiguous slice of every plane (the plain-fill bound with this grid)
// hipcc --offload-arch=gfx950 -O3 ub_store2.hip -o ub_store2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define B 4096
#define S 9
#define T 100
#define TC 20
#define EPB 4
static __device__ __forceinline__ int xcd_block() {
  const unsigned n = gridDim.x, x = blockIdx.x & 7u, q = n >> 3, rem = n & 7u;
  return (int)(x * q + (x < rem ? x : rem) + (blockIdx.x >> 3));
}
struct Bufs { float *obs, *rew, *act; unsigned char *tru, *ter; };
enum { TM = 0, EM = 1, REC = 2, FILL = 3 };
template <int MODE, int NT, int NW>       // NW worker threads (the last NW of NT) do the stores, like the kernel's worker waves
__global__ __launch_bounds__(NT) void k(Bufs o, int spin, int remap) {
  constexpr int G = EPB * S, G4 = G / 4;
  const int bid = remap ? xcd_block() : (int)blockIdx.x;
  const long long total = (long long)B * S, g_base = (long long)bid * G;
  const int tid = (int)threadIdx.x - (NT - NW);
  if (tid < 0) return;
  volatile float sink = 0.f;
  for (int c = 0; c < T / TC; ++c) {
    float f = 1.f;
    for (int i = 0; i < spin; ++i) f = f * 1.0001f + 0.5f;
    if (MODE == TM) {
      const long long row0 = (long long)c * TC * total + g_base;
      constexpr int PR = G * 3 / 4;
      for (int q = tid; q < TC * PR; q += NW) { const int r = q / PR, pc = q - r * PR;
        *(float4*)((char*)(o.obs + row0 * 3) + (size_t)((unsigned)r * (unsigned)total * 12u) + pc * 16) = make_float4(f, f, f, (float)q); }
      for (int u = tid; u < TC * G4; u += NW) {
        const int r = u / G4, gl0 = (u - r * G4) * 4;
        const unsigned eo = (unsigned)r * (unsigned)total + gl0;
        const float4 v = make_float4(f, f, f, (float)u);
        *(float4*)((char*)(o.rew + row0) + (size_t)(eo * 4u)) = v;
        *(float4*)((char*)(o.act + row0) + (size_t)(eo * 4u)) = v;
        *(unsigned*)((char*)(o.tru + row0) + (size_t)eo) = u;
        *(unsigned*)((char*)(o.ter + row0) + (size_t)eo) = 0u;
      }
    } else if (MODE == EM) {
      constexpr int PE = TC * S * 3 / 4, WE = TC * S / 4;      // 16-byte pieces per env and chunk: obs 135, reward / action 45 (flags: 45 words)
      for (int q = tid; q < EPB * PE; q += NW) { const int e = q / PE, p = q - e * PE;
        const size_t it = ((size_t)(bid * EPB + e) * T + (size_t)c * TC) * S;
        *(float4*)((char*)(o.obs + it * 3) + p * 16) = make_float4(f, f, f, (float)q); }
      for (int q = tid; q < EPB * WE; q += NW) { const int e = q / WE, p = q - e * WE;
        const size_t it = ((size_t)(bid * EPB + e) * T + (size_t)c * TC) * S;
        const float4 v = make_float4(f, f, f, (float)q);
        *(float4*)((char*)(o.rew + it) + p * 16) = v;
        *(float4*)((char*)(o.act + it) + p * 16) = v;
        *(unsigned*)(o.tru + it + p * 4) = q;
        *(unsigned*)(o.ter + it + p * 4) = 0u; }
    } else if (MODE == REC) {
      constexpr int PE = TC * S * 24 / 16;                     // 270 pieces per env and chunk
      for (int q = tid; q < EPB * PE; q += NW) { const int e = q / PE, p = q - e * PE;
        const size_t it = ((size_t)(bid * EPB + e) * T + (size_t)c * TC) * S;
        *(float4*)((char*)o.obs + it * 24 + p * 16) = make_float4(f, f, f, (float)q); }
    } else {
      // the block's share of every plane, contiguous: chunk c of the block's [T/TC] slices
      const size_t n_it = (size_t)TC * G;                      // items per block and chunk
      const size_t it0 = ((size_t)bid * (T / TC) + c) * n_it;
      for (int q = tid; q < (int)(n_it * 3 / 4); q += NW) *(float4*)((char*)(o.obs + it0 * 3) + q * 16) = make_float4(f, f, f, (float)q);
      for (int q = tid; q < (int)(n_it / 4); q += NW) {
        const float4 v = make_float4(f, f, f, (float)q);
        *(float4*)((char*)(o.rew + it0) + q * 16) = v; *(float4*)((char*)(o.act + it0) + q * 16) = v;
        *(unsigned*)(o.tru + it0 + q * 4) = q; *(unsigned*)(o.ter + it0 + q * 4) = 0u; }
    }
    sink = f;
  }
}
template <int MODE, int NT, int NW> void run(const char* name, int spin, int nbuf, int remap = 1) {
  const size_t n = (size_t)T * B * S;
  std::vector<Bufs> bs(nbuf);
  for (auto& b : bs) {
    if (MODE == REC) { hipMalloc(&b.obs, n * 24); b.rew = b.act = nullptr; b.tru = b.ter = nullptr; }
    else { hipMalloc(&b.obs, n * 12); hipMalloc(&b.rew, n * 4); hipMalloc(&b.act, n * 4); hipMalloc(&b.tru, n); hipMalloc(&b.ter, n); }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2 * nbuf; ++i) hipLaunchKernelGGL((k<MODE, NT, NW>), dim3(B / EPB), dim3(NT), 0, 0, bs[i % nbuf], spin, remap);
  hipEventRecord(e0);
  const int reps = 200;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, NT, NW>), dim3(B / EPB), dim3(NT), 0, 0, bs[i % nbuf], spin, remap);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, bytes = (double)n * (MODE == REC ? 24 : 22);
  printf("%-34s nbuf %d spin %5d remap %d: %7.2f us/launch  %.2f TB/s (%.0f MB)\n", name, nbuf, spin, remap, us, bytes / (us * 1e-6) / 1e12, bytes / 1e6);
  for (auto& b : bs) { hipFree(b.obs); if (b.rew) { hipFree(b.rew); hipFree(b.act); hipFree(b.tru); hipFree(b.ter); } }
}
int main() {
  for (int nbuf : {1, 5}) for (int spin : {0, 1500}) {
    run<TM, 256, 192>("TM  time-major (as the kernel)", spin, nbuf);
    run<EM, 256, 192>("EM  env-major", spin, nbuf);
    run<REC, 256, 192>("REC env-major, 24-byte records", spin, nbuf);
    run<FILL, 256, 192>("FILL block-contiguous planes", spin, nbuf);
    run<EM, 256, 256>("EM  env-major, 256 store threads", spin, nbuf);
    run<FILL, 256, 256>("FILL, 256 store threads", spin, nbuf);
  }
  run<TM, 256, 192>("TM  no xcd remap", 0, 5, 0);
  run<EM, 256, 192>("EM  no xcd remap", 0, 5, 0);
  return 0;
}

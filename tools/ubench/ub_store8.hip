// Short-lived workgroups writing the rollout's TIME-MAJOR record tiles: workgroup (c, p) writes rows [c*R, (c+1)*R) of the G consecutive
// (env, shop) pairs [p*G, (p+1)*G) of every plane (obs [T][N][3] f32, action / reward [T][N] f32, terminated / truncated [T][N] u8),
// 16-byte stores, consecutive lanes consecutive pieces of a row segment.  READ: also load the tile's 4-byte compact records first
// (the expansion pass of a two-phase rollout).  Is the "short-lived workgroups fill faster" effect there for THIS store pattern?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static constexpr int T = 400, N = 4096 * 9;
struct Planes { char *obs, *act, *rew, *ter, *tru; const uint32_t* rec; };
template <int G, int R, bool REMAP, bool READ>
__global__ __launch_bounds__(256) void k(Planes P) {
  constexpr int NP = N / G;
  int id = blockIdx.x;
  if (REMAP) { const int x = id & 7, q = id >> 3, per = gridDim.x >> 3; id = x * per + q; }      // consecutive tiles on one XCD
  const int p = id % NP, c = id / NP;
  float add = 0.f;
  if (READ) {
    uint32_t s = 0;
    for (int i = threadIdx.x; i < G * R; i += 256) s += P.rec[((size_t)(c * R + i / G)) * N + p * G + i % G];
    add = (float)s;
  }
  constexpr int OB = G * 12 / 16, AC = G * 4 / 16, FL = (G + 15) / 16, ROW = OB + 2 * AC + 2 * FL;
  for (int i = threadIdx.x; i < ROW * R; i += 256) {
    const int r = i / ROW, j = i % ROW;
    const size_t t = (size_t)c * R + r;
    const float4 v = make_float4(1.f + add, 2.f, 3.f, (float)j);
    if (j < OB) *(float4*)(P.obs + (t * N + (size_t)p * G) * 12 + j * 16) = v;
    else if (j < OB + AC) *(float4*)(P.act + (t * N + (size_t)p * G) * 4 + (j - OB) * 16) = v;
    else if (j < OB + 2 * AC) *(float4*)(P.rew + (t * N + (size_t)p * G) * 4 + (j - OB - AC) * 16) = v;
    else if (j < OB + 2 * AC + FL) *(float4*)(P.ter + (t * N + (size_t)p * G) + (j - OB - 2 * AC) * 16) = v;
    else *(float4*)(P.tru + (t * N + (size_t)p * G) + (j - OB - 2 * AC - FL) * 16) = v;
  }
}
template <int G, int R, bool REMAP, bool READ> void run() {
  const size_t items = (size_t)T * N, bytes = items * 22;
  std::vector<Planes> bufs(2);
  for (auto& b : bufs) {
    hipMalloc(&b.obs, items * 12); hipMalloc(&b.act, items * 4); hipMalloc(&b.rew, items * 4); hipMalloc(&b.ter, items + 64); hipMalloc(&b.tru, items + 64);
    uint32_t* r; hipMalloc(&r, items * 4); hipMemset(r, 1, items * 4); b.rec = r;
  }
  const int blocks = (N / G) * (T / R);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<G, R, REMAP, READ>), dim3(blocks), dim3(256), 0, 0, bufs[i % 2]);
  hipEventRecord(e0);
  const int reps = 40;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<G, R, REMAP, READ>), dim3(blocks), dim3(256), 0, 0, bufs[i % 2]);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("G = %2d pairs x R = %3d rows%s%s, %6d blocks: %8.2f us  %.3f of 8 TB/s\n", G, R, REMAP ? " xcd-remap" : "", READ ? " +read" : "", blocks, us,
         bytes / (us * 1e-6) / 8e12);
  for (auto& b : bufs) { hipFree(b.obs); hipFree(b.act); hipFree(b.rew); hipFree(b.ter); hipFree(b.tru); hipFree((void*)b.rec); }
}
int main() {
  run<48, 4, false, false>(); run<48, 10, false, false>(); run<48, 20, false, false>(); run<48, 100, false, false>();
  run<64, 4, false, false>(); run<64, 10, false, false>(); run<64, 20, false, false>();
  run<48, 4, true, false>(); run<48, 20, true, false>(); run<64, 4, true, false>(); run<64, 20, true, false>();
  run<48, 4, false, true>(); run<48, 20, false, true>(); run<64, 20, true, true>();
  run<128, 4, false, false>(); run<128, 20, false, false>(); run<256, 4, false, false>(); run<256, 4, false, true>();
  return 0;
}

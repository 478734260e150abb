// Time-major RECORD-INTERLEAVED trajectory (one 24-byte record per (step, env, shop): obs 12 + action 4 + reward 4 + 4 flag bytes)
// against the seven separate planes of ub_store8, same launch shapes: workgroup (c, p) writes rows [c*R, (c+1)*R) of the G consecutive
// pairs [p*G, (p+1)*G).  One row segment of a workgroup is G * 24 contiguous bytes (48 pairs: 1 152 B = nine whole lines) and the
// whole step row is ONE stream.  PERSIST: a grid of N / G workgroups that loop over all chunks (the rollout kernel's shape).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static constexpr int T = 400, N = 4096 * 9;
struct Planes { char *obs, *act, *rew, *ter, *tru; };
template <int G, int R, int AOS, bool PERSIST, int DESYNC = 1>
__global__ __launch_bounds__(256) void k(Planes P, char* rec) {
  constexpr int NP = N / G;
  int id = blockIdx.x;
  { const int x = id & 7, q = id >> 3, per = gridDim.x >> 3; id = x * per + q; }      // consecutive tiles on one XCD
  const int p = PERSIST ? id : id % NP;
  const int c_off = (DESYNC > 1 && PERSIST) ? (id % DESYNC) * ((T / R) / DESYNC) : 0;      // DESYNC: workgroups walk the rows from different starts
  for (int cc = PERSIST ? 0 : id / NP; cc < T / R; cc += PERSIST ? 1 : T) {
    const int c = PERSIST ? (cc + c_off) % (T / R) : cc;
    if (AOS == 1) {
      constexpr int ROW = G * 24 / 16;
      for (int i = threadIdx.x; i < ROW * R; i += 256) {
        const int r = i / ROW, j = i % ROW;
        const size_t t = (size_t)c * R + r;
        *(float4*)(rec + (t * N + (size_t)p * G) * 24 + j * 16) = make_float4(1.f, 2.f, 3.f, (float)j);
      }
    } else {
      constexpr int OB = G * 12 / 16, AC = G * 4 / 16, FL = (G + 15) / 16, ROW = OB + 2 * AC + 2 * FL;
      for (int i = threadIdx.x; i < ROW * R; i += 256) {
        const int r = i / ROW, j = i % ROW;
        const size_t t = (size_t)c * R + r;
        const float4 v = make_float4(1.f, 2.f, 3.f, (float)j);
        if (AOS == 3 || AOS == 4) {
          constexpr int F2 = (AOS == 4) ? 0 : (G * 2 + 15) / 16, ROW3 = OB + 2 * AC + F2;
          if (i >= ROW3 * R) break;
          const int r3 = i / ROW3, j3 = i % ROW3;
          const size_t t3 = (size_t)c * R + r3;
          if (j3 < OB) *(float4*)(P.obs + (t3 * N + (size_t)p * G) * 12 + j3 * 16) = v;
          else if (j3 < OB + AC) *(float4*)(P.act + (t3 * N + (size_t)p * G) * 4 + (j3 - OB) * 16) = v;
          else if (j3 < OB + 2 * AC) *(float4*)(P.rew + (t3 * N + (size_t)p * G) * 4 + (j3 - OB - AC) * 16) = v;
          else *(float4*)(rec + (t3 * N + (size_t)p * G) * 2 + (j3 - OB - 2 * AC) * 16) = v;
        } else
        if (AOS == 2) {                                        // ROW-CONCATENATED planes: step t is ONE 22 N-byte super-row [obs | act | rew | ter | tru]
          char* row = rec + t * (size_t)N * 22;
          if (j < OB) *(float4*)(row + (size_t)p * G * 12 + j * 16) = v;
          else if (j < OB + AC) *(float4*)(row + (size_t)N * 12 + (size_t)p * G * 4 + (j - OB) * 16) = v;
          else if (j < OB + 2 * AC) *(float4*)(row + (size_t)N * 16 + (size_t)p * G * 4 + (j - OB - AC) * 16) = v;
          else if (j < OB + 2 * AC + FL) *(float4*)(row + (size_t)N * 20 + (size_t)p * G + (j - OB - 2 * AC) * 16) = v;
          else *(float4*)(row + (size_t)N * 21 + (size_t)p * G + (j - OB - 2 * AC - FL) * 16) = v;
        } else
        if (j < OB) *(float4*)(P.obs + (t * N + (size_t)p * G) * 12 + j * 16) = v;
        else if (j < OB + AC) *(float4*)(P.act + (t * N + (size_t)p * G) * 4 + (j - OB) * 16) = v;
        else if (j < OB + 2 * AC) *(float4*)(P.rew + (t * N + (size_t)p * G) * 4 + (j - OB - AC) * 16) = v;
        else if (j < OB + 2 * AC + FL) *(float4*)(P.ter + (t * N + (size_t)p * G) + (j - OB - 2 * AC) * 16) = v;
        else *(float4*)(P.tru + (t * N + (size_t)p * G) + (j - OB - 2 * AC - FL) * 16) = v;
      }
    }
    if (PERSIST) __syncthreads();
  }
}
template <int G, int R, int AOS, bool PERSIST, int DESYNC = 1> void run() {
  const size_t items = (size_t)T * N, bytes = items * 22;
  std::vector<Planes> bufs(2); std::vector<char*> recs(2);
  for (int i = 0; i < 2; ++i) {
    auto& b = bufs[i];
    hipMalloc(&b.obs, items * 12); hipMalloc(&b.act, items * 4); hipMalloc(&b.rew, items * 4); hipMalloc(&b.ter, items + 64); hipMalloc(&b.tru, items + 64);
    hipMalloc(&recs[i], items * 24);
  }
  const int blocks = PERSIST ? N / G : (N / G) * (T / R);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<G, R, AOS, PERSIST, DESYNC>), dim3(blocks), dim3(256), 0, 0, bufs[i % 2], recs[i % 2]);
  hipEventRecord(e0);
  const int reps = 40;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<G, R, AOS, PERSIST, DESYNC>), dim3(blocks), dim3(256), 0, 0, bufs[i % 2], recs[i % 2]);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("G = %3d pairs x R = %3d rows %s %s desync %2d, %6d blocks: %8.2f us  %.3f of 8 TB/s (22-byte record; %.3f on the bytes written)\n", G, R,
         AOS == 1 ? "RECORDS (24 B)" : AOS == 2 ? "ROW-CONCAT    " : AOS == 3 ? "flags as u16  " : AOS == 4 ? "no flag planes" : "seven planes  ", PERSIST ? "persistent grid" : "short-lived    ", DESYNC, blocks, us,
         bytes / (us * 1e-6) / 8e12, (AOS == 1 ? items * 24 : bytes) / (us * 1e-6) / 8e12);
  for (int i = 0; i < 2; ++i) { auto& b = bufs[i]; hipFree(b.obs); hipFree(b.act); hipFree(b.rew); hipFree(b.ter); hipFree(b.tru); hipFree(recs[i]); }
}
int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<48, 20, 0, true, 1>(); run<48, 20, 3, true, 1>(); run<48, 20, 4, true, 1>(); run<48, 20, 1, true, 1>();
    run<32, 20, 0, true, 1>(); run<32, 20, 3, true, 1>(); run<32, 20, 4, true, 1>();
  }
  return 0;
}

// Round 4: what can DEDICATED STORE WAVES do?  The rollout kernel's worker waves compute a chunk's outputs and store them themselves: a
// wave stalled on store back-pressure computes nothing.  Here a persistent grid of N / G workgroups (the rollout's shape) walks the
// T = 400 fragment in chunks of TC rows; per chunk the "worker" waves do VW dependent VALU operations per lane (a stand-in for draws +
// outputs) and write a staged tile to LDS, and NSW store waves stream the PREVIOUS chunk's staged tile (obs [TC][3G], reward [TC][G],
// optionally action [TC][G]) out of LDS with 1 KB per store instruction (consecutive 16-byte pieces from consecutive lanes).  One
// barrier per chunk, as in the kernel.  ACT_BY_WORKERS: the action plane is written by the workers as 4-byte stores per lane (what a
// draw phase that never stages the action would do).  FLAGS 1: the store waves also write dense flag rows (G bytes per row and plane) with
// every chunk; 2: the same bytes as whole 128-byte lines with one writer each; 3: whole ROWS of both planes (row t by workgroup t % blocks)
// before the streaming starts.  Round 4's finding: on some boxes FLAGS 1 costs 8-10 us of a 57 us launch, 2 the same, 3 about 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static constexpr int T = 400, N = 4096 * 9;
struct Planes { char *obs, *act, *rew, *ter, *tru; };

template <int G, int TC, int NT, int NSW, int VW, bool ACT_BY_WORKERS, int FLAGS>
__global__ __launch_bounds__(NT) void k(Planes P, int xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PO = 3 * G / 4, PR = G / 4;                      // 16-byte pieces per row: obs, reward (= action)
  constexpr int ROWP = PO + PR + (ACT_BY_WORKERS ? 0 : PR);      // staged pieces per row
  float4* stage = (float4*)smem;                                 // 2 x [TC][ROWP]
  int id = blockIdx.x;
  if (xcd) { const int x = id & 7, q = id >> 3, per = gridDim.x >> 3; id = x * per + q; }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int NW = NT / 64, NWORK = NW - NSW;
  const size_t g0 = (size_t)id * G;
  if (FLAGS == 3 && wave >= NWORK) {   // whole ROWS of the two flag planes, row t by workgroup t % blocks, before the streaming starts
    const int sl0 = (wave - NWORK) * 64 + lane;
    for (int t = id; t < T; t += (int)gridDim.x)
      for (int q = sl0; q < N / 16; q += NSW * 64) {
        *(float4*)(P.tru + (size_t)t * N + (size_t)q * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)(P.ter + (size_t)t * N + (size_t)q * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  for (int c = 0; c <= T / TC; ++c) {
    if (wave < NWORK) {
      if (c < T / TC) {
        // stand-in for compute: VW dependent integer operations per lane, then the chunk's staged tile
        uint32_t v = tid + c;
#pragma unroll 8
        for (int i = 0; i < VW; ++i) v = v * 1664525u + 1013904223u + (v >> 7);
        float4* st = stage + (c & 1) * TC * ROWP;
        for (int i = tid; i < TC * ROWP; i += NWORK * 64) st[i] = make_float4((float)v, 1.f, 2.f, (float)i);
        if (ACT_BY_WORKERS) {
          for (int i = tid; i < TC * G; i += NWORK * 64) {
            const int r = i / G, j = i - r * G;
            *(float*)(P.act + (((size_t)c * TC + r) * N + g0 + j) * 4) = (float)v;
          }
        }
      }
    } else if (c > 0) {
      const int cs = c - 1;
      const float4* st = stage + (cs & 1) * TC * ROWP;
      const int sl = (wave - NWORK) * 64 + lane;
      for (int i = sl; i < TC * ROWP; i += NSW * 64) {
        // plane-major within the chunk: all obs rows, then reward rows, then action rows
        const float4 v = st[i];
        if (i < TC * PO) { const int r = i / PO, j = i - r * PO; *(float4*)(P.obs + (((size_t)cs * TC + r) * N + g0) * 12 + j * 16) = v; }
        else if (i < TC * (PO + PR)) { const int k2 = i - TC * PO, r = k2 / PR, j = k2 - r * PR; *(float4*)(P.rew + (((size_t)cs * TC + r) * N + g0) * 4 + j * 16) = v; }
        else { const int k2 = i - TC * (PO + PR), r = k2 / PR, j = k2 - r * PR; *(float4*)(P.act + (((size_t)cs * TC + r) * N + g0) * 4 + j * 16) = v; }
      }
      if (FLAGS == 3) {
      } else if (FLAGS == 2) {        // whole 128-byte lines, one writer per line: [roundup(g0, 128), roundup(g0 + G, 128))
        const int lo = (int)((g0 + 127) & ~(size_t)127), hi = (int)(((g0 + G + 127) & ~(size_t)127) < (size_t)N ? ((g0 + G + 127) & ~(size_t)127) : (size_t)N);
        const int PFx = (hi - lo) >> 4;
        for (int i = sl; i < 2 * TC * PFx; i += NSW * 64) {
          const int pl = i >= TC * PFx, k2 = i - pl * TC * PFx, r = k2 / PFx, j = k2 - r * PFx;
          *(float4*)((pl ? P.ter : P.tru) + ((size_t)cs * TC + r) * N + lo + j * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else if (FLAGS) {
        constexpr int PF = G / 16;
        for (int i = sl; i < 2 * TC * PF; i += NSW * 64) {
          const int pl = i >= TC * PF, k2 = i - pl * TC * PF, r = k2 / PF, j = k2 - r * PF;
          *(float4*)((pl ? P.ter : P.tru) + ((size_t)cs * TC + r) * N + g0 + j * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

// reference: short-lived workgroups, one 16-byte store per thread, the same bytes
__global__ __launch_bounds__(256) void fill(float4* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

static std::vector<Planes> bufs;
static void alloc() {
  const size_t items = (size_t)T * N;
  bufs.resize(2);
  for (auto& b : bufs) { hipMalloc(&b.obs, items * 12); hipMalloc(&b.act, items * 4); hipMalloc(&b.rew, items * 4); hipMalloc(&b.ter, items + 64); hipMalloc(&b.tru, items + 64); }
}
template <int G, int TC, int NT, int NSW, int VW, bool AW, int FL> void run(int xcd = 1) {
  const size_t items = (size_t)T * N, bytes = items * (FL ? 22 : 20);
  constexpr int ROWP = 3 * G / 4 + G / 4 + (AW ? 0 : G / 4);
  const size_t lds = 2 * (size_t)TC * ROWP * 16;
  hipFuncSetAttribute((const void*)k<G, TC, NT, NSW, VW, AW, FL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = N / G;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<G, TC, NT, NSW, VW, AW, FL>), dim3(blocks), dim3(NT), lds, 0, bufs[i % 2], xcd);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<G, TC, NT, NSW, VW, AW, FL>), dim3(blocks), dim3(NT), lds, 0, bufs[i % 2], xcd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms / reps < best) best = ms / reps;
  }
  const double us = best * 1e3;
  printf("G %3d TC %2d NT %4d store-waves %d VW %5d act-by-workers %d flags %d xcd %d lds %6zu blocks %4d: %8.2f us  %.3f of 8 TB/s on the %d-byte record\n",
         G, TC, NT, NSW, VW, (int)AW, (int)FL, xcd, lds, blocks, us, bytes / (us * 1e-6) / 8e12, FL ? 22 : 20);
  fflush(stdout);
}
static void run_fill() {
  const size_t items = (size_t)T * N, n = items * 20 / 16;
  float4* p; hipMalloc(&p, 2 * n * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p + (i % 2) * n, n);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p + (i % 2) * n, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("plain fill of the same 20-byte bytes, short-lived workgroups: %8.2f us  %.3f of 8 TB/s\n", ms * 1e3 / 20, items * 20 / (ms * 1e-3 / 20) / 8e12);
  hipFree(p);
}
int main() {
  alloc();
  for (int rep = 0; rep < 2; ++rep) {
    run_fill();
    // stores only (VW = 0): the ceiling of the store-wave pattern, by shape
    run<144, 20, 1024, 2, 0, false, 0>(); run<144, 20, 1024, 4, 0, false, 0>(); run<144, 20, 1024, 8, 0, false, 0>(); run<144, 20, 1024, 1, 0, false, 0>();
    run<144, 10, 1024, 2, 0, false, 0>(); run<144, 10, 1024, 4, 0, false, 0>();
    run<144, 20, 1024, 2, 0, true, 0>(); run<144, 20, 1024, 4, 0, true, 0>();
    run<144, 20, 1024, 2, 0, false, 1>(); run<144, 20, 1024, 4, 0, false, 1>(); run<144, 20, 1024, 4, 0, false, 2>();
    run<144, 16, 1024, 4, 0, false, 0>(); run<144, 16, 1024, 4, 0, false, 1>(); run<144, 16, 1024, 4, 0, false, 2>(); run<144, 16, 1024, 4, 0, false, 3>();
    run<144, 20, 1024, 4, 0, false, 0>(0);
    run<48, 20, 384, 1, 0, false, 0>(); run<48, 20, 384, 2, 0, false, 0>(); run<48, 20, 384, 1, 0, true, 0>(); run<48, 20, 384, 1, 0, false, 1>();
    run<96, 20, 512, 2, 0, false, 0>(); run<72, 20, 512, 2, 0, false, 0>();
    // with stand-in compute beside the stores (VW dependent ops per lane and chunk)
    run<144, 20, 1024, 2, 400, false, 0>(); run<144, 20, 1024, 2, 800, false, 0>(); run<144, 20, 1024, 2, 1600, false, 0>(); run<144, 20, 1024, 4, 800, false, 0>();
    run<144, 20, 1024, 2, 800, true, 0>();
    run<48, 20, 384, 1, 800, false, 0>(); run<48, 20, 384, 1, 1600, false, 0>();
  }
  return 0;
}

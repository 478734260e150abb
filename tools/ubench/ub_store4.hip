// pure-store variants, round 3 (SC64 B=4096: 36864 pairs, NSTEP steps, TC=20): which store ORDER does the memory system like?
//   FLAT_T   flat fill of the same bytes, torch-like grid (one float4 per thread)
//   FLAT_G   flat fill, 1152 blocks x 192 threads, each block one contiguous slice
//   TM32     time-major, blocks of 32 pairs, unit order as the kernel (a wave's 64 lanes span 8 tile rows)
//   TM32R    time-major, blocks of 32 pairs, a wave writes ONE tile row of a plane per instruction (lanes beyond the row idle)
//   TM32P    time-major, blocks of 32 pairs, plane after plane (all rows of obs, then reward, ...)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef NSTEP
#define NSTEP 100
#endif
#define B 4096
#define S 9
#ifndef TC
#define TC 20
#endif
#define G 32
static __device__ __forceinline__ int xcd_block() {
  const unsigned n = gridDim.x, x = blockIdx.x & 7u, q = n >> 3, rem = n & 7u;
  return (int)(x * q + (x < rem ? x : rem) + (blockIdx.x >> 3));
}
struct Bufs { float *obs, *rew, *act; unsigned char *tru, *ter; };
enum { FLAT_T, FLAT_G, TM32, TM32R, TM32P, TM32NT, FLAT_TNT, TM32NOFLAG, TM32FLAG128 };
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE, int NW>
__global__ __launch_bounds__(256) void k(Bufs o, char* flat, size_t flat_bytes) {
  const long long total = (long long)B * S;
  if (MODE == FLAT_T || MODE == FLAT_TNT) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i < flat_bytes) { if (MODE == FLAT_TNT) { v4f v = {1.f, 2.f, 3.f, 4.f}; __builtin_nontemporal_store(v, (v4f*)(flat + i)); } else *(float4*)(flat + i) = make_float4(1.f, 2.f, 3.f, 4.f); }
    return;
  }
  const int bid = xcd_block();
  const int tid = (int)threadIdx.x - (256 - NW);
  if (tid < 0) return;
  if (MODE == FLAT_G) {
    const size_t per = flat_bytes / gridDim.x / 16 * 16;
    char* p = flat + (size_t)bid * per;
    for (size_t q = tid; q < per / 16; q += NW) *(float4*)(p + q * 16) = make_float4(1.f, 2.f, 3.f, (float)q);
    return;
  }
  const long long g_base = (long long)bid * G;
  constexpr int G4 = G / 4, PR = G * 3 / 4;
  for (int c = 0; c < NSTEP / TC; ++c) {
    const long long row0 = (long long)c * TC * total + g_base;
    const float f = (float)c;
    if (MODE == TM32NT) {
      for (int q = tid; q < TC * PR; q += NW) { const int r = q / PR, pc = q - r * PR;
        v4f v = {f, f, f, (float)q};
        __builtin_nontemporal_store(v, (v4f*)((char*)(o.obs + row0 * 3) + (size_t)((unsigned)r * (unsigned)total * 12u) + pc * 16)); }
      for (int u = tid; u < TC * G4; u += NW) {
        const int r = u / G4, gl0 = (u - r * G4) * 4;
        const unsigned eo = (unsigned)r * (unsigned)total + gl0;
        v4f v = {f, f, f, (float)u};
        __builtin_nontemporal_store(v, (v4f*)((char*)(o.rew + row0) + (size_t)(eo * 4u)));
        __builtin_nontemporal_store(v, (v4f*)((char*)(o.act + row0) + (size_t)(eo * 4u)));
        __builtin_nontemporal_store((unsigned)u, (unsigned*)((char*)(o.tru + row0) + (size_t)eo));
        __builtin_nontemporal_store(0u, (unsigned*)((char*)(o.ter + row0) + (size_t)eo));
      }
    } else if (MODE == TM32NOFLAG || MODE == TM32FLAG128) {
      for (int q = tid; q < TC * PR; q += NW) { const int r = q / PR, pc = q - r * PR;
        *(float4*)((char*)(o.obs + row0 * 3) + (size_t)((unsigned)r * (unsigned)total * 12u) + pc * 16) = make_float4(f, f, f, (float)q); }
      for (int u = tid; u < TC * G4; u += NW) {
        const int r = u / G4, gl0 = (u - r * G4) * 4;
        const unsigned eo = (unsigned)r * (unsigned)total + gl0;
        const float4 v = make_float4(f, f, f, (float)u);
        *(float4*)((char*)(o.rew + row0) + (size_t)(eo * 4u)) = v;
        *(float4*)((char*)(o.act + row0) + (size_t)(eo * 4u)) = v;
      }
      if (MODE == TM32FLAG128) {
        // the chunk's flag bands ([TC rows][B*S] bytes per plane) as 128-byte pieces dealt round robin to the blocks: block b takes
        // pieces b, b + nblocks, ... of the band (any block can compute any flag: truncated depends on the env's step counter only)
        const size_t band = (size_t)TC * total, npieces = band / 128;
        char* tb = (char*)o.tru + (size_t)c * TC * total; char* eb = (char*)o.ter + (size_t)c * TC * total;
        for (size_t p = (size_t)bid * 8 + (tid >> 3); p < npieces; p += (size_t)gridDim.x * 8) {       // 8 lanes x 16 B = one piece
          if ((tid >> 3) >= 8) break;
          *(float4*)(tb + p * 128 + (tid & 7) * 16) = make_float4(f, f, f, f);
          *(float4*)(eb + p * 128 + (tid & 7) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else if (MODE == TM32) {
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
    } else if (MODE == TM32R) {
      // a wave takes tile rows round robin; within a row: 24 obs pieces + 8 rew + 8 act + 8 + 8 flag words = 56 lanes busy
      const int wave = tid >> 6, lane = tid & 63, nwv = NW / 64;
      for (int r = wave; r < TC; r += nwv) {
        const size_t ro = (size_t)((unsigned)r * (unsigned)total);
        const float4 v = make_float4(f, f, f, (float)r);
        if (lane < 24) *(float4*)((char*)(o.obs + row0 * 3) + ro * 12 + lane * 16) = v;
        else if (lane < 32) *(float4*)((char*)(o.rew + row0) + ro * 4 + (lane - 24) * 16) = v;
        else if (lane < 40) *(float4*)((char*)(o.act + row0) + ro * 4 + (lane - 32) * 16) = v;
        else if (lane < 48) *(unsigned*)((char*)(o.tru + row0) + ro + (lane - 40) * 4) = r;
        else if (lane < 56) *(unsigned*)((char*)(o.ter + row0) + ro + (lane - 48) * 4) = 0u;
      }
    } else {
      for (int q = tid; q < TC * PR; q += NW) { const int r = q / PR, pc = q - r * PR;
        *(float4*)((char*)(o.obs + row0 * 3) + (size_t)((unsigned)r * (unsigned)total * 12u) + pc * 16) = make_float4(f, f, f, (float)q); }
      for (int u = tid; u < TC * G4; u += NW) { const int r = u / G4, gl0 = (u - r * G4) * 4; const unsigned eo = (unsigned)r * (unsigned)total + gl0;
        *(float4*)((char*)(o.rew + row0) + (size_t)(eo * 4u)) = make_float4(f, f, f, (float)u); }
      for (int u = tid; u < TC * G4; u += NW) { const int r = u / G4, gl0 = (u - r * G4) * 4; const unsigned eo = (unsigned)r * (unsigned)total + gl0;
        *(float4*)((char*)(o.act + row0) + (size_t)(eo * 4u)) = make_float4(f, f, f, (float)u); }
      for (int u = tid; u < TC * G4; u += NW) { const int r = u / G4, gl0 = (u - r * G4) * 4; const unsigned eo = (unsigned)r * (unsigned)total + gl0;
        *(unsigned*)((char*)(o.tru + row0) + (size_t)eo) = u; *(unsigned*)((char*)(o.ter + row0) + (size_t)eo) = 0u; }
    }
  }
}
template <int MODE, int NW> void run(const char* name, int nbuf) {
  const size_t n = (size_t)NSTEP * B * S, bytes = n * 22;
  std::vector<Bufs> bs(nbuf); std::vector<char*> flats(nbuf);
  for (int i = 0; i < nbuf; ++i) {
    hipMalloc(&flats[i], bytes);
    char* p = flats[i];
    bs[i].obs = (float*)p; bs[i].rew = (float*)(p + n * 12); bs[i].act = (float*)(p + n * 16); bs[i].tru = (unsigned char*)(p + n * 20); bs[i].ter = (unsigned char*)(p + n * 21);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (MODE == FLAT_T || MODE == FLAT_TNT) ? (int)((bytes / 16 + 255) / 256) : (int)((size_t)B * S / G);
  auto launch = [&](int i) { hipLaunchKernelGGL((k<MODE, NW>), dim3(grid), dim3(256), 0, 0, bs[i % nbuf], flats[i % nbuf], bytes); };
  for (int i = 0; i < 2 * nbuf; ++i) launch(i);
  hipEventRecord(e0);
  const int reps = 100;
  for (int i = 0; i < reps; ++i) launch(i);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("%-46s T=%d nbuf %d: %8.2f us/launch  %.2f TB/s = %.3f of 8 TB/s\n", name, NSTEP, nbuf, us, bytes / (us * 1e-6) / 1e12, bytes / (us * 1e-6) / 8e12);
  for (auto p : flats) hipFree(p);
}
int main(int argc, char** argv) {
  const int nbuf = argc > 1 ? atoi(argv[1]) : 5;
  run<FLAT_T, 256>("FLAT_T flat fill, torch-like grid", nbuf);
  run<FLAT_G, 192>("FLAT_G flat fill, 1152 blocks x 192 thr", nbuf);
  run<FLAT_G, 256>("FLAT_G flat fill, 1152 blocks x 256 thr", nbuf);
  run<TM32, 192>("TM32   kernel order (planes of one flat buffer)", nbuf);
  run<TM32R, 192>("TM32R  one tile row per wave instruction", nbuf);
  run<TM32P, 192>("TM32P  plane after plane", nbuf);
  run<TM32, 256>("TM32   kernel order, 256 store threads", nbuf);
  run<TM32NOFLAG, 192>("TM32   without the two flag planes (91 % of the bytes)", nbuf);
  run<TM32FLAG128, 192>("TM32   flags as 128-byte pieces dealt to the blocks", nbuf);
  run<TM32NT, 192>("TM32NT kernel order, nontemporal stores", nbuf);
  run<FLAT_TNT, 256>("FLAT_T nontemporal", nbuf);
  return 0;
}

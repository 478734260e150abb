// phx_sc_rollout.hip -- the time-parallel supply-chain rollout kernel, round-2 structure.
//
// Same path as phx_sc_rollout_kernel (phx_sc_fused.hip): T consecutive PhantomEnv.step() calls of a
// FACTORY / SHOP / CUSTOMER env (supply_chain.py:36-150, env.py:239-303) per launch, random policy and
// device-RNG orders, auto-reset at episode end (the list-of-envs loop of utils/rllib/rollout.py:361-363),
// bit-identical trajectories.  What changed is who does what.  Measured on the round-1 kernel (PHX_TIMING):
// the stock recurrence -- one lane per (env, shop), sequential in time -- was the critical path of every
// chunk: 210 cycles per step for ~21 instructions, because each of its instructions queues behind the
// quarter-rate multiplies of the three draw waves sharing its SIMD, and the draw and output waves idled at
// the barrier meanwhile (4.4 k of 48 k cycles); setup took another 7 k (six dependent global round trips).
// Here
//   * the recurrence wave does ONLY the dependent chain: its operands (one packed word per step: request R
//     and demand D) are fetched from LDS in one burst before the chain, the episode reset is folded into the
//     operands (D' = 4096, R' = 0 at the episode's last step give stock' = 0 without a select on the chain),
//     and per step it stores one word (the stock BEFORE the step) -- sub, max, add per step on the chain;
//   * everything derived (sales, missed sales, stock after the step, observation, reward) is recomputed
//     from (stock before, D, R) in the output phase, which is parallel over (time, pairs);
//   * all per-block constants (pair table, digit-sum table, observation / penalty tables) are COMPUTED into
//     LDS at setup (1 024 workgroups fetching the same cache lines at launch cost ~10 k cycles);
//   * the recurrence waves do nothing else: the other waves write the outputs of chunk c and draw chunk c + 2
//     while the recurrence of chunk c + 1 runs -- one barrier per chunk;
//   * what bounds the kernel now is how fast the memory system takes the trajectory (stores ablated: 15.5 us;
//     a store-only kernel with this access pattern: 16-20 us): the observation pieces of a work unit are
//     transposed through LDS so that consecutive lanes write consecutive 16-byte pieces, and a short first
//     chunk starts the stores early.
// Fast-path conditions (checked at phx_create / by the launcher; everything else takes
// phx_sc_rollout_kernel): device RNG and random policy (no replay), every shop with the same 1..6 customers
// and the same normaliser, whole envs per block with 16-byte aligned tile rows, num_steps >= the chunk length.
#include "phx_dev.h"
#include "phx_sc_fast.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

struct FastArgs {
  int32_t B, S, epb, G, K, T, num_steps, xcd_remap, first_rows, whole_envs;
  int32_t flags_sparse;             // the flag planes were zero-filled before the launch: only the non-zero words are stored
  uint32_t pK; float inv_pK;        // 5^K and its f32 reciprocal (floor(y * inv) == y / 5^K for y < 5^6: tests/test_host_logic.py)
  uint32_t mG, mG4, mS, mPR;        // ceil(2^32 / d) magics: i / G, i / (G / 4), i / S, i / (3 G / 4)  for i < 2^16
  int32_t norm;                     // the shops' common max_sales_per_step
  uint64_t seed; int64_t env_offset;
  unsigned long long* timing;       // PHX_TIMING builds only
  int32_t *stock, *sales, *missed, *delivered, *env_step, *env_tick, *env_arrive;
  phx_rollout_io io;
};

__device__ __forceinline__ void fast_lds_barrier() {     // orders LDS traffic only: trajectory stores stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// four workgroups per CU must stay co-resident (B = 4096 is ONE round of 1 024 workgroups): NT / 64 waves per SIMD
// The arguments are read from the kernarg segment through a constant-address-space pointer whose provenance is hidden from
// the compiler again at every chunk iteration (FAST_REFRESH): scalar loads where a field is used.  Used by value, the block
// was loaded at entry and kept: 67 spilled SGPRs, v_readlane / v_writelane traffic inside the chunk loop.
typedef const __attribute__((address_space(4))) char* fast_kptr_t;
#define a (*(const FastArgs*)kp)
#define io (a.io)
#define FAST_REFRESH() asm volatile("" : "+s"(kp))
template <int NT>
__global__ __launch_bounds__(NT, (NT > 512 ? NT / 256 : NT / 64)) void phx_sc_rollout_fast_kernel(const FastArgs a_) {
  fast_kptr_t kp = (fast_kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  FAST_REFRESH();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TC = PHX_FAST_TC;
  const int tid = threadIdx.x, nS = a.S, G = a.G;
  const int64_t total = (int64_t)a.B * nS;
  const int bid = xcd_block(a.xcd_remap != 0);
  // the block's G consecutive (env, shop) pairs: whole envs where a multiple of 4 of them fits a block (a.whole_envs),
  // otherwise a range of pairs that starts and ends inside envs -- shops never interact, an env's shops only share
  // its step counter and tick, which every block that holds a part of the env walks identically
  const int64_t g_base = (int64_t)bid * G;
  const int64_t b_first = a.whole_envs ? (int64_t)bid * a.epb : g_base / nS;
  const int r0 = (int)(g_base - b_first * nS);                          // the first pair's shop
  const int n_env = a.whole_envs ? a.epb : (r0 + G - 1) / nS + 1;       // envs the block touches (<= a.epb)
#ifdef PHX_TIMING
  unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define FTICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tm[k] += now_ - tprev; tprev = now_; } while (0)
#else
#define FTICK(k) do {} while (0)
#endif

  // ---- LDS carve (16-byte aligned sections).  The constant tables are COMPUTED here, not loaded: 1 024 workgroups
  //      fetching the same few cache lines at launch cost ~10 k cycles of setup (measured, PHX_TIMING) ------------
  uint32_t* s_pair = (uint32_t*)smem;                                   // [G] shop | env_local << 8
  uint8_t* s_ds = (uint8_t*)(s_pair + ((G + 3) & ~3));                  // [125] base-5 digit sum of k < 5^3
  float* s_tabs = (float*)(s_ds + 128);                                 // [101] stock / 100            encode_observation,
  float* s_tabn = s_tabs + 104;                                         // [5K+1 <= 31] x / norm        supply_chain.py:124-134
  double* s_pen = (double*)(s_tabn + 32);                               // [101] 0.1 * stock in f64     compute_reward :147
  const int items = TC * G;
  int* s_rd0 = (int*)(s_pen + 102);                                     // 3 x [TC][G]  R | D << 8        (chunk c in tile c % 3)
  float* s_act0 = (float*)(s_rd0 + 3 * items);                          // 3 x [TC][G]  action
  int* s_xb0 = (int*)(s_act0 + 3 * items);                              // 2 x [TC][G]  stock before the step (chunk c in c & 1)
  const int G4p = (G + 3) & ~3;
  int* s_ptend0 = s_xb0 + 2 * items;                                    // 2 x [G] chunk row that ends the pair's episode, or -1
  float* s_ostage = (float*)(s_ptend0 + 2 * G4p);                       // [TC][G][3] observation pieces on their way out (wave-private regions)
  int* s_tick0 = (int*)(s_ostage + 3 * items);                          // [epb]
  int* s_flags = s_tick0 + ((a.epb + 3) & ~3);                          // [0] a tick is not a multiple of 4, [1] a stock outside [0, 100]

  // ---- setup: the state loads are in flight while the tables are computed --------------------------------------
  int x = 0, step = 0;                    // lane state of the recurrence (lane tid owns pair g_base + tid)
  {
    int tk = 0;
    const uint32_t pt = (uint32_t)(r0 + tid);
    const uint32_t el = nS == 1 ? pt : __umulhi(pt, a.mS);               // (r0 + tid) / S (the magic of 1 does not fit 32 bits)
    if (tid < G) { x = a.stock[g_base + tid]; step = a.env_step[b_first + el]; }
    if (tid < n_env) tk = a.env_tick[b_first + tid];
    if (tid < G) { s_pair[tid] = (pt - el * (uint32_t)nS) | (el << 8); }
    if (tid < 125) s_ds[tid] = (uint8_t)(tid % 5 + (tid / 5) % 5 + tid / 25);
    if (tid <= PHX_SHOP_MAX_STOCK) {
      // f32 IEEE division == the reference's f64 quotient cast to f32 for |ints| < 2^24 (phx_dev.h: shop_obs_f32)
      s_tabs[tid] = (float)tid / (float)PHX_SHOP_MAX_STOCK;
      s_pen[tid] = __dmul_rn(0.1, (double)tid);
    }
    if (tid <= 5 * a.K) s_tabn[tid] = (float)tid / (float)a.norm;
    if (tid < 2) s_flags[tid] = 0;
    __syncthreads();
    if (tid < n_env) { s_tick0[tid] = tk; if (tk & 3) s_flags[0] = 1; }
    if (tid < G && (unsigned)x > (unsigned)PHX_SHOP_MAX_STOCK) s_flags[1] = 1;
    __syncthreads();
  }
  const int quad_extra = s_flags[0];      // chunk starts are not quad-aligned for every env: one more row quad
  const bool weird = s_flags[1] != 0;     // a stock the caller set outside [0, 100] (any step brings it back into range)
  FTICK(0);

  // Thread roles: the waves that hold a recurrence lane (tid < rec_threads) walk the stock recurrence and nothing
  // else after the first chunk's draws; the other waves ("workers") write the outputs of chunk c and draw chunk
  // c + 2 while the recurrence of chunk c + 1 runs beside them -- one barrier per chunk.
  const int rec_threads = ((G + 63) >> 6) << 6;
  // Chunks: a.first_rows rows first, then TC rows each.  (A short first chunk would start the trajectory stores
  // earlier -- nothing is stored before the first chunk is drawn and walked -- but every extra iteration costs
  // more than that buys: first_rows = 4 / 8 / 12 measured 23.3 / 23.3 / 23.5 us per launch against 21.5 us for
  // first_rows = TC, SC64, B = 4096, T = 100.  PHX_ROLLOUT_FIRST overrides.)
  const int first_rows = a.first_rows;
  const int n_chunks = 1 + (a.T - first_rows + TC - 1) / TC;
  auto start_of = [&](int c) { return c == 0 ? 0 : first_rows + (c - 1) * TC; };
  auto rows_of = [&](int c) { const int left = a.T - start_of(c); return c == 0 ? first_rows : (left < TC ? left : TC); };

  // ---- draws of the chunk starting at step t0 (tc rows) into tile `buf`, by threads [first, NT).
  //      One Philox block serves ticks 4q .. 4q + 3 of a shop: work items are (row quad jr, pair gl).
  auto draws_impl = [&](int t0, int tc, int buf, int first, auto ALIGNED, auto K6) __attribute__((always_inline)) {
    // ALIGNED: every env's chunk starts on a tick quad and tc is a multiple of 4 -> every row of a unit exists
    // K6: every shop has six customers (all six base-5 digits of y count)
    constexpr bool aligned = decltype(ALIGNED)::value, k6 = decltype(K6)::value;
    if (tid < first) return;
    int* s_rd = s_rd0 + buf * items;
    float* s_act = s_act0 + buf * items;
    const int nw = NT - first;
    const int n_work = (((tc + 3) >> 2) + (aligned ? 0 : quad_extra)) * G;
    for (int iw = tid - first; iw < n_work; iw += nw) {
      const int jr = (int)__umulhi((uint32_t)iw, a.mG), gl = iw - (int)__umul24(jr, G);
      const uint32_t pr = s_pair[gl];
      const int s = (int)(pr & 255u), bl = (int)(pr >> 8);
      const int64_t genv = a.env_offset + b_first + bl;
      const uint32_t tick_base = (uint32_t)s_tick0[bl] + (uint32_t)t0;
      const int tla = 4 * jr - (aligned ? 0 : (int)(tick_base & 3u));
      if (!aligned && (tla + 3 < 0 || tla >= tc)) continue;
      const uint32_t tick_a = tick_base + (uint32_t)tla;                 // multiple of 4
      uint32_t w[4];
#ifdef PHX_ABL_NODRAW
      w[0] = tick_a * 2654435761u + (uint32_t)genv; w[1] = w[0] * 40503u + s; w[2] = w[1] ^ 0x9e3779b9u; w[3] = w[2] + w[0];   // dev ablation: no Philox
#else
      rng_block(a.seed, genv, tick_a, s, 0, 0, w);
#endif
      // the four words of the block -> four ticks.  The accept path is branch-free so that the four items' LDS lookups
      // overlap; the redraw of a rejected word (probability 3.3e-6) is ONE cold branch after it
      uint32_t y[4], aj[4];
      bool rej = false;
#pragma unroll
      for (int h = 0; h < 4; ++h) rej |= !rng_split(w[h], y[h], aj[h]);
      if (__builtin_expect(rej, 0)) {
#pragma unroll 1
        for (int h = 0; h < 4; ++h) {
          uint32_t y2, j2;
          if (!rng_split(w[h], y2, j2)) y[h] = rng_group_y(a.seed, genv, tick_a + (uint32_t)h, s, 0, 1, &aj[h]);
        }
      }
      int i = __mul24(tla, G) + gl;
#pragma unroll
      for (int h = 0; h < 4; ++h, i += G) {
        if (!aligned) { const int tl = tla + h; if (tl < 0 || tl >= tc) continue; }
        uint32_t yy = y[h];
        if (!k6) yy -= __umul24((uint32_t)((float)yy * a.inv_pK), a.pK);  // the first K base-5 digits: y mod 5^K
        const uint32_t hi = (uint32_t)((float)yy * 0.008f);              // y / 125, exact through f32
        const int D = (int)s_ds[hi] + (int)s_ds[__mul24((int)hi, -125) + (int)yy];   // digits of y / 125 and y % 125: the customers' order sizes summed, supply_chain.py:61-67
        const float action = rng_j_to_action(aj[h]);                     // random policy, [0, 100)
        s_act[i] = action;
        s_rd[i] = (int)rintf(action) | (D << 8);                         // decode_action: int(round(action)), supply_chain.py:139
      }
    }
  };
  auto draws = [&](int t0, int tc, int buf, int first) __attribute__((always_inline)) {
    if (a.K == 6) {
      if (!quad_extra && (tc & 3) == 0) draws_impl(t0, tc, buf, first, std::true_type{}, std::true_type{});
      else draws_impl(t0, tc, buf, first, std::false_type{}, std::true_type{});
    } else {
      if (!quad_extra && (tc & 3) == 0) draws_impl(t0, tc, buf, first, std::true_type{}, std::false_type{});
      else draws_impl(t0, tc, buf, first, std::false_type{}, std::false_type{});
    }
  };

  // ---- the stock recurrence of chunk c (tc rows), one lane per pair ----------------------------------------------
  //   stock' = max(stock - D, 0) + min(R, 100 - stock)      handle_order_request / handle_stock_response,
  //   supply_chain.py:98-122 with decode_action's clamp :139 (<= 100 whenever 0 <= stock <= 100); at the
  //   episode's last step the caller's env.reset() zeroes the stock (ShopAgent.reset): folded into the operands
  //   (D' = 4096, R' = 0 give stock' = 0), so the chain carries no select.  Per step it stores ONE word, the stock
  //   before the step; sales, missed sales and the stock after it are recomputed by the output phase.
  int fin_xb = 0, fin_rd = 0;             // the launch's last step: stock before it and its packed (R, D)
  auto recurrence = [&](int c, int tc) __attribute__((always_inline)) {
    if (tid >= G) return;
    __builtin_amdgcn_s_setprio(3);        // a dependent chain beside waves of Philox / output work: issue whenever ready
    const int* rdp = s_rd0 + (c % 3) * items + tid;
    int* xb = s_xb0 + (c & 1) * items + tid;
    const int tend = a.num_steps - 1 - step;                            // chunk row that ends the episode (one at most: TC <= num_steps)
    const bool ends = tend >= 0 && tend < tc;
    s_ptend0[(c & 1) * G4p + tid] = ends ? tend : -1;
    // one step of the chain; `MINCAP` keeps the general form for a stock the caller set outside [0, 100]
#define FAST_STEP(h_, rd_, MINCAP)                                                                         \
    {                                                                                                      \
      const bool end_ = ((h_) == tend);                                                                    \
      const int Dm_ = end_ ? 4096 : ((rd_) >> 8), Rm_ = end_ ? 0 : ((rd_) & 255);                          \
      xb[(h_) * G] = x; fin_xb = x;                                                                        \
      const int xn_ = max(x - Dm_, 0) + min(Rm_, PHX_SHOP_MAX_STOCK - x);                                  \
      x = (MINCAP) ? (end_ ? 0 : min(xn_, PHX_SHOP_MAX_STOCK)) : xn_;                                      \
    }
    if (tc == TC && !weird) {               // straight-line code, the chunk's operands fetched in one burst
      // The episode reset costs the chain nothing: the tile word of the episode's last step is PATCHED to D = 255, R = 0
      // before the burst (stock' = max(stock - 255, 0) + min(0, ...) = 0 for stock <= 100) and restored after the chain
      // for the output phase.  Per step that leaves sub, max, sub, min, add (R and D are byte fields of the word) and one
      // LDS store: every instruction of this wave waits ~8 cycles for a quarter-rate multiply of the draw waves to
      // leave the VALU, priority or not, so the recurrence lasts as long as its instruction count.
      int* rdw = s_rd0 + (c % 3) * items + tid;
      int orig = 0;
      if (ends) { orig = rdw[tend * G]; rdw[tend * G] = 0xFF00; }
      int rd[TC];
#pragma unroll
      for (int h = 0; h < TC; ++h) rd[h] = rdw[h * G];
#pragma unroll
      for (int h = 0; h < TC; ++h) {
        xb[h * G] = x; fin_xb = x;
        x = max(x - ((rd[h] >> 8) & 255), 0) + min(rd[h] & 255, PHX_SHOP_MAX_STOCK - x);
      }
      if (ends) rdw[tend * G] = orig;
      fin_rd = (ends && tend == TC - 1) ? orig : rd[TC - 1];
    } else {                                // a ragged last chunk, or an out-of-range stock at launch
      for (int h = 0; h < tc; ++h) { const int rdh = rdp[h * G]; FAST_STEP(h, rdh, true) fin_rd = rdh; }
    }
#undef FAST_STEP
    step += tc;
    if (ends) step -= a.num_steps;
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- outputs of chunk c: observation / reward / flags straight to HBM, by the workers ----------------------------
  // A work unit is 4 consecutive pairs of one tile row: 12 observation floats, 4 rewards, 4 actions and 4 + 4
  // flag bytes = whole 16-byte (4-byte for the flags) segments of the [T][B][S] arrays, computed in registers
  // from (stock before, D, R).
  auto outputs_impl = [&](int c, int t0, int tc, auto GUARD) __attribute__((always_inline)) {
    constexpr bool guard = decltype(GUARD)::value;
    const int first = rec_threads;
    if (tid < first) return;
    int* s_rd = s_rd0 + (c % 3) * items;
    float* s_act = s_act0 + (c % 3) * items;
    int* s_xb = s_xb0 + (c & 1) * items;
    const int* s_ptend = s_ptend0 + (c & 1) * G4p;
    // scalar 64-bit bases of the chunk's first row + 32-bit element offsets per work unit (TC * B * S * 12 < 2^32
    // is checked by the plan): the stores take the SGPR-base + VGPR-offset form, no 64-bit address arithmetic
    const int64_t row0 = (int64_t)t0 * total + g_base;
    char* const p_obs = (char*)(io.obs + row0 * 3);
    char* const p_rew = (char*)(io.reward + row0);
    char* const p_act = (char*)(io.action_out + row0);
    char* const p_tru = (char*)(io.truncated + row0);
    char* const p_ter = (char*)(io.terminated + row0);
    const bool wr_ter = io.terminated != nullptr;           // NULL: the caller does not want the all-zero plane (ShopAgent never terminates)
    const bool flags_sparse = a.flags_sparse != 0;
    const uint32_t utotal = (uint32_t)total;
    const int G4 = G >> 2, nw = NT - first, n_units = tc * G4;
    const uint32_t PR = 3u * (uint32_t)G4;                              // 16-byte observation pieces per tile row
    const uint32_t row_bytes = utotal * 12u;                            // bytes between tile rows of the observation plane
    const int lane = tid & 63;
    for (int ub = (tid - first) - lane; ub < n_units; ub += nw) {       // ub: the wave's first unit (uniform per wave)
      const int u = ub + lane;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vrw = va;
      uint32_t tr = 0u, eo = 0u;
      if (u < n_units) {
        const int r = (int)__umulhi((uint32_t)u, a.mG4);
        const int gl0 = (u - (int)__umul24(r, G4)) << 2, i0 = (int)__umul24(r, G) + gl0;
        const uint4 vr = *(const uint4*)(s_rd + i0), vx = *(const uint4*)(s_xb + i0), ve = *(const uint4*)(s_ptend + gl0);
        va = *(const float4*)(s_act + i0);
        const int rdv[4] = {(int)vr.x, (int)vr.y, (int)vr.z, (int)vr.w}, xbv[4] = {(int)vx.x, (int)vx.y, (int)vx.z, (int)vx.w};
        float o[12], rw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int R = rdv[k] & 255, D = rdv[k] >> 8, x0 = xbv[k];
          const int sales = min(x0, D);                                 // handle_order_request :105-122 (== x0 - max(x0 - D, 0))
          const int missed = D - sales;
          int xa = x0 - sales + min(R, PHX_SHOP_MAX_STOCK - x0);        // handle_stock_response :98-103
          if (guard) xa = min(xa, PHX_SHOP_MAX_STOCK);
          if (!guard || ((unsigned)x0 <= (unsigned)PHX_SHOP_MAX_STOCK)) {
            o[3 * k] = s_tabs[xa]; o[3 * k + 1] = s_tabn[sales]; o[3 * k + 2] = s_tabn[missed];   // encode_observation :124-134
            rw[k] = (float)__dsub_rn((double)sales, s_pen[xa]);                                 // compute_reward :147, rounded once to f32
          } else {                                                      // a stock the caller set outside [0, 100]: the formulas
            float ob[3];
            shop_obs_f32(xa, sales, missed, (float)a.norm, ob);
            o[3 * k] = ob[0]; o[3 * k + 1] = ob[1]; o[3 * k + 2] = ob[2];
            rw[k] = (float)shop_reward(sales, xa);
          }
        }
        vrw = make_float4(rw[0], rw[1], rw[2], rw[3]);
        tr = (uint32_t)(r == (int)ve.x) | ((uint32_t)(r == (int)ve.y) << 8) |
             ((uint32_t)(r == (int)ve.z) << 16) | ((uint32_t)(r == (int)ve.w) << 24);            // truncations["__all__"], env.py:312-318
        eo = (uint32_t)r * utotal + (uint32_t)gl0;                      // element offset from the chunk's first row
        // The unit's three 16-byte observation pieces go through LDS: piece 3 u + j is their position in the chunk's
        // row-major observation block, so that each store instruction below writes CONSECUTIVE pieces from
        // consecutive lanes.  (A 48-byte lane stride costs ~3x the L2 write transactions: the store-only
        // microbenchmark of this trajectory takes 19.9 us per launch with it, 16.2 us with contiguous pieces,
        // 14 us for a plain fill of the same bytes.)
        float4* st = (float4*)(s_ostage + 12 * u);
        st[0] = make_float4(o[0], o[1], o[2], o[3]); st[1] = make_float4(o[4], o[5], o[6], o[7]); st[2] = make_float4(o[8], o[9], o[10], o[11]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the wave's own pieces are in LDS (wave-private region: no barrier)
#ifdef PHX_ABL_NOSTORE
      if (a.norm != -12345) { if (u < n_units && vrw.x == 1.2345e30f) *(float4*)p_rew = vrw; continue; }   // dev ablation: everything but the stores
#endif
      const uint32_t q0 = 3u * (uint32_t)ub, qn = 3u * (uint32_t)n_units;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const uint32_t q = q0 + (uint32_t)lane + 64u * (uint32_t)k;
        if (q < qn) {
          const uint32_t rr = __umulhi(q, a.mPR), pc = q - rr * PR;     // tile row and piece within the row
          *(float4*)(p_obs + (size_t)(rr * row_bytes + pc * 16u)) = *(const float4*)(s_ostage + 4 * q);
        }
      }
      if (u < n_units) {
        *(float4*)(p_rew + (size_t)(eo * 4u)) = vrw;
        *(float4*)(p_act + (size_t)(eo * 4u)) = va;
#ifndef PHX_ABL_NOFLAGS
        if (flags_sparse) { if (tr) *(uint32_t*)(p_tru + (size_t)eo) = tr; }       // (an episode's last row: one row in num_steps)
        else {
          *(uint32_t*)(p_tru + (size_t)eo) = tr;
          if (wr_ter) *(uint32_t*)(p_ter + (size_t)eo) = 0u;
        }
#else
        if (a.norm == -12345) *(uint32_t*)(p_tru + (size_t)eo) = tr;     // dev ablation: neither flag plane is stored
#endif
      }
    }
  };
  auto outputs = [&](int c, int t0, int tc) __attribute__((always_inline)) {
    if (weird && c == 0) outputs_impl(c, t0, tc, std::true_type{});
    else outputs_impl(c, t0, tc, std::false_type{});
  };

  // ---- schedule.  Iteration it = -2: everybody draws chunk 0;  it = -1: recurrence(0) beside draws(1);  it >= 0:
  //        workers: outputs(it), draws(it + 2)    beside    recurrence lanes: recurrence(it + 1)        one barrier
  //      Draw tiles rotate over three buffers (draws(c + 2) overwrites what outputs(c - 1) read before the last
  //      barrier), stock tiles over two.  One call site per phase keeps the code small.
  for (int it = -2; it < n_chunks; ++it) {
    FAST_REFRESH();
    const int co = it, cr = it + 1, cd = it + 2;
    if (tid < rec_threads && it > -2) {
      if (cr < n_chunks) recurrence(cr, rows_of(cr));
      FTICK(3);
    } else {
      if (co >= 0) { outputs(co, start_of(co), rows_of(co)); FTICK(5); }
      if (cd < n_chunks) draws(start_of(cd), rows_of(cd), cd % 3, it == -2 ? 0 : rec_threads);
      FTICK(1);
    }
    fast_lds_barrier(); FTICK(6);
  }
#ifdef PHX_TIMING
  if (a.timing && (tid & 63) == 0) for (int q = 0; q < 8; ++q) a.timing[((int64_t)blockIdx.x * (NT / 64) + (tid >> 6)) * 8 + q] = tm[q];
#endif
  // ---- state after the fragment ---------------------------------------------------------------------------------
  if (tid < G) {
    const int64_t g = g_base + tid;
    const int R = fin_rd & 255, D = fin_rd >> 8;
    const int sales = min(fin_xb, D), missed = D - sales;
    a.stock[g] = x; a.sales[g] = sales; a.missed[g] = missed; a.delivered[g] = min(R, PHX_SHOP_MAX_STOCK - fin_xb);
    if (io.last_obs) {
      float ob[3];
      shop_obs(x, sales, missed, a.norm, ob);
      io.last_obs[g * 3 + 0] = ob[0]; io.last_obs[g * 3 + 1] = ob[1]; io.last_obs[g * 3 + 2] = ob[2];
    }
    const uint32_t pr = s_pair[tid];
    const int bl = (int)(pr >> 8);
    if (a.whole_envs) {
      if ((pr & 255u) == 0u) { a.env_step[b_first + bl] = step; a.env_tick[b_first + bl] = s_tick0[bl] + a.T; }
    } else if ((pr & 255u) == 0u || tid == 0) {
      // Blocks of pair ranges: another block that holds a part of this env may not have read its step counter and
      // tick yet (a block of a later round of the grid), so the block that FINISHES LAST with the env writes them:
      // every block counts itself in after its own reads (they were consumed at setup), the one that completes the
      // count -- the number of blocks whose pair range meets the env -- stores the new words and clears the counter
      // (stream order makes both visible to the next launch).  No second launch, no co-residency assumption.
      const int64_t b = b_first + bl, p0 = b * nS;
      const int n_touch = (int)((p0 + nS - 1) / G - p0 / G) + 1;
      if (n_touch == 1 || atomicAdd(&a.env_arrive[b], 1) + 1 == n_touch) {
        a.env_step[b] = step; a.env_tick[b] = s_tick0[bl] + a.T;
        if (n_touch > 1) a.env_arrive[b] = 0;
      }
    }
  }
}

#undef a
#undef io
#undef FAST_REFRESH

// ---- host: plan, blob, launcher -------------------------------------------------------------------------------------
static uint32_t magic32(int d) { return (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)(d > 0 ? d : 1)); }

bool phx_sc_fast_plan(int B, int S, int K_uniform, bool norm_uniform, int num_steps, int block, bool aligned, ScFastPlan* p) {
  memset(p, 0, sizeof *p);
  const int off = (phx_knobs().rollout_fast == 0);     // development default only:
  if (off) return false;                                                                               // per env: phx_spec.variant_rollout
  if (K_uniform < 1 || K_uniform > 6 || !norm_uniform || S < 1 || S > 255 || num_steps < PHX_FAST_TC) return false;
  const int64_t total = (int64_t)B * S;
  const int g_env = phx_knobs().rollout_g;               // development default of variant_block
  if (block == 0) block = g_env;
  // (the kernel's LDS -- phx_launch_sc_rollout_fast: constants + 11 tile words per item of a 20-row chunk -- has to fit the CU's 160 KB:
  //  ADVICE r4, block = 184 .. 192 planned and then failed at launch)
  auto lds_of = [&](int G) { const size_t epb_ = (size_t)((G + S - 2) / S + 1);
    return (size_t)((G + 3) & ~3) * 12 + 128 + 104 * 4 + 32 * 4 + 102 * 8 + (size_t)PHX_FAST_TC * G * 4 * 11 + ((epb_ + 3) & ~(size_t)3) * 4 + 16; };
  auto pairs_ok = [&](int G) { return G >= 4 && G <= 192 && G % 4 == 0 && total % G == 0 && (G + S - 2) / S + 1 <= 255 && lds_of(G) <= (size_t)160 * 1024; };
  // (1) whole envs per block, a multiple of 4 of them so that every tile row is a whole number of 16-byte segments;
  //     ~32..64 pairs per block (one recurrence wave)
  int epb = 0;
  for (int cand = 4; cand * S <= 96 && cand <= 255; cand += 4) if (cand * S >= 32) { epb = cand; break; }
  if (!epb && 4 * S <= 96) epb = 4;
  const bool whole_ok = epb && B % epb == 0 && total % 4 == 0;
  // (2) blocks of G CONSECUTIVE (env, shop) PAIRS that start and end inside envs.  With G % 16 == 0 every row segment a
  //     block writes is a whole number of 64-byte pieces of the f32 planes (the memory-side write request): the lines at
  //     block boundaries need no merging of partial writes in the L2.  Measured (tools/ubench/ub_store3.hip, the rollout
  //     itself: DESIGN 3.3): on some MI355X boxes the partially written boundary lines of 36-pair blocks cost 1.6x
  //     (T = 400, SC64, B = 4096: 118 vs 75-85 us per launch), on the others whole-env blocks are ~5 % faster; a grid that
  //     is a multiple of the 256 CUs keeps the one-round launch balanced (SC64, B = 4096: 48 pairs -> 768 blocks = 3 per CU).
  int G = 0;
  if (block > 0 && pairs_ok(block)) G = block;
  else if (block != PHX_VB_WHOLE_ENVS && (aligned || !whole_ok)) {
    static const int pref16[] = {32, 48, 64, 16, 80, 96, 112, 128};
    if (aligned) {
      for (int cand : pref16) if (pairs_ok(cand) && (total / cand) % 256 == 0) { G = cand; break; }
      if (!G) for (int cand : pref16) if (pairs_ok(cand)) { G = cand; break; }
    }
    if (!G && !whole_ok) {
      static const int pref[] = {32, 64, 48, 40, 36, 44, 52, 56, 60, 28, 24, 68, 72, 76, 80, 84, 88, 92, 96};
      for (int cand : pref) if (pairs_ok(cand)) { G = cand; break; }
    }
  }
  if (G) { p->G = G; p->epb = (G + S - 2) / S + 1; p->whole_envs = 0; }       // epb: the most envs a block can touch
  else if (whole_ok) { p->epb = epb; p->G = epb * S; p->whole_envs = 1; }
  else return false;
  if ((int64_t)PHX_FAST_TC * B * S * 12 >= ((int64_t)1 << 32)) return false;      // 32-bit store offsets within a chunk
  p->K = K_uniform;
  const int p2w = (p->G + 63) / 64, p1w = ((PHX_FAST_TC / 4) * p->G + 63) / 64;     // draws of a chunk in one pass
  int want = 64 * (p2w + p1w);
  p->nt = want <= 256 ? 256 : (want <= 320 ? 320 : (want <= 384 ? 384 : (want <= 512 ? 512 : (want <= 768 ? 768 : 1024))));
  p->ok = 1;
  return true;
}

// zeros for the flag planes: one 16-byte store per thread, short-lived workgroups, the tail byte by byte.  (A kernel of our own, not
// hipMemsetAsync: as fast, and a memset NODE captured into a hipGraph did not zero the planes on replay with this runtime --
// test_rollout_graph_replays_the_same_fragments_as_rollout_calls.)
__global__ __launch_bounds__(256) void phx_zero_fill_kernel(uint8_t* __restrict__ p, const int64_t n16, const int tail) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) ((float4*)p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < tail) p[n16 * 16 + i] = 0;
}
static hipError_t zero_fill(void* p, int64_t nbytes, hipStream_t st) {
  if (((uintptr_t)p) & 15u) return hipErrorInvalidValue;              // (phx_rollout checks the planes' alignment)
  const int64_t n16 = nbytes >> 4;
  const int64_t threads = n16 > 16 ? n16 : 16;
  hipLaunchKernelGGL(phx_zero_fill_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (uint8_t*)p, n16, (int)(nbytes & 15));
  return hipGetLastError();
}

hipError_t phx_launch_sc_rollout_fast(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st) {
  const ScFastPlan& p = sp.sc_fast;
  FastArgs a;
  a.B = sp.B; a.S = sp.S; a.epb = p.epb; a.G = p.G; a.whole_envs = p.whole_envs; a.K = p.K; a.T = io.T; a.num_steps = sp.num_steps;
  const int remap_env = phx_knobs().rollout_remap;
  a.xcd_remap = remap_env >= 0 ? remap_env : 1;
  uint32_t pk = 1; for (int k = 0; k < p.K; ++k) pk *= 5u;
  static const float inv[7] = {1.0f, 0.2f, 0.04f, 0.008f, 0.0016f, 0.00032f, 0.000064f};
  a.pK = pk; a.inv_pK = inv[p.K];
  a.mG = magic32(p.G); a.mG4 = magic32(p.G / 4); a.mS = magic32(sp.S); a.mPR = magic32(3 * (p.G / 4));
  a.norm = p.norm; a.seed = sp.seed; a.env_offset = sp.env_offset;
  const int first_env = phx_knobs().rollout_first;
  int first = first_env > 0 ? first_env : PHX_FAST_TC;
  if (first > PHX_FAST_TC) first = PHX_FAST_TC;
  if (io.T <= PHX_FAST_TC) first = io.T;
  a.first_rows = first;
  a.stock = (int32_t*)sp.f[F_SHOP_STOCK]; a.sales = (int32_t*)sp.f[F_SHOP_SALES];
  a.missed = (int32_t*)sp.f[F_SHOP_MISSED]; a.delivered = (int32_t*)sp.f[F_SHOP_DELIVERED];
  a.env_step = (int32_t*)sp.f[F_ENV_STEP]; a.env_tick = (int32_t*)sp.f[F_ENV_TICK]; a.env_arrive = (int32_t*)sp.f[F_ENV_ARRIVE];
  a.io = io;
  a.timing = nullptr;
#ifdef PHX_TIMING
  { static unsigned long long* tbuf = nullptr; if (!tbuf) (void)hipMalloc((void**)&tbuf, 8 * 8 * 8192 * sizeof(unsigned long long)); a.timing = tbuf;
    if (getenv("PHX_TIMING_DUMP")) { static int calls = 0; if (++calls == 20) { (void)hipDeviceSynchronize(); std::vector<unsigned long long> h(8 * 8 * 8192); (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
      const int wpb = p.nt / 64, nw = (int)(((int64_t)sp.B * sp.S) / p.G) * wpb; double sum[8] = {0}, w0[8] = {0}; for (int w = 0; w < nw; ++w) for (int q = 0; q < 8; ++q) { sum[q] += h[(size_t)w * 8 + q]; if (w % wpb == 0) w0[q] += h[(size_t)w * 8 + q]; }
      fprintf(stderr, "FAST_TIMING avg cycles per wave:  setup %.0f | draws %.0f | bar %.0f | P2 %.0f | bar %.0f | out %.0f | bar+setup-loads %.0f | P2 preload %.0f\n", sum[0]/nw, sum[1]/nw, sum[2]/nw, sum[3]/nw, sum[4]/nw, sum[5]/nw, sum[6]/nw, sum[7]/nw);
      fprintf(stderr, "FAST_TIMING wave0 of each block:  setup %.0f | draws %.0f | bar %.0f | P2 %.0f | bar %.0f | out %.0f | bar+setup-loads %.0f | P2 preload %.0f\n", w0[0]*wpb/nw, w0[1]*wpb/nw, w0[2]*wpb/nw, w0[3]*wpb/nw, w0[4]*wpb/nw, w0[5]*wpb/nw, w0[6]*wpb/nw, w0[7]*wpb/nw); } } }
#endif
  const int items = PHX_FAST_TC * p.G;
  // constants (pair table, digit sums, observation / penalty tables) + 3 draw tiles x 2 planes + 2 stock tiles +
  // 2 episode-end rows + the observation staging tile (3 floats per item) + ticks + flags
  const size_t lds = (size_t)((p.G + 3) & ~3) * 4 + 128 + 104 * 4 + 32 * 4 + 102 * 8 + (size_t)items * 4 * (6 + 2 + 3) +
                     (size_t)((p.G + 3) & ~3) * 8 + (size_t)((p.epb + 3) & ~3) * 4 + 16;
  // The flag planes are 2 of the record's 22 bytes but a workgroup's share of a row is 32-48 bytes of each -- partial 64-byte
  // write requests unless a neighbour's share reaches the L2 in time to merge; in the store pattern alone they cost a quarter to
  // a third of the launch (tools/ubench/ub_store9.hip: 84-89 us with them, 54-62 without).  They are all zero except the
  // episodes' last rows: a streaming fill (whole lines, ~5 us per T = 400 fragment) writes the zeros, the kernel the exceptions.
  const int sparse_env = phx_knobs().rollout_sparse_flags;      // development
  const int64_t n_flag = (int64_t)io.T * sp.B * sp.S;
  a.flags_sparse = (sparse_env == 2 || (sparse_env && n_flag >= ((int64_t)1 << 23))) ? 1 : 0;      // (PHX_ROLLOUT_SPARSE_FLAGS: 0 never, 1 large fragments, 2 always)
  if (a.flags_sparse) {
    phx_note_kernel("phx_zero_fill_kernel[flag planes]");
    hipError_t me;
    if (io.terminated && io.terminated == io.truncated + n_flag) me = zero_fill(io.truncated, 2 * n_flag, st);
    else {
      me = zero_fill(io.truncated, n_flag, st);
      if (me == hipSuccess && io.terminated) me = zero_fill(io.terminated, n_flag, st);
    }
    if (me != hipSuccess) return me;
  }
  const dim3 grid((unsigned)(((int64_t)sp.B * sp.S) / p.G));
  const int nt_env = phx_knobs().rollout_nt;
  const int rec = ((p.G + 63) / 64) * 64;
  const int nt = (nt_env && nt_env >= rec + 64) ? nt_env : p.nt;
  phx_note_kernel(p.whole_envs ? "phx_sc_rollout_fast_kernel[whole_envs]" : "phx_sc_rollout_fast_kernel[pairs]");
#define PHX_LAUNCH_FAST(NT_) hipLaunchKernelGGL((phx_sc_rollout_fast_kernel<NT_>), grid, dim3(NT_), lds, st, a)
  if (lds > 64 * 1024) {               // more than 64 KB of dynamic LDS needs the attribute (wide workgroups)
    // (per device, result checked: ADVICE r4 -- a second GPU of the process never got the attribute)
    static PhxPerDeviceOnce attr_done; int dev = 0; (void)hipGetDevice(&dev);
    if (!attr_done.done(dev)) {
      hipError_t ae = hipFuncSetAttribute((const void*)phx_sc_rollout_fast_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (ae == hipSuccess) ae = hipFuncSetAttribute((const void*)phx_sc_rollout_fast_kernel<768>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (ae == hipSuccess) ae = hipFuncSetAttribute((const void*)phx_sc_rollout_fast_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (ae != hipSuccess) return ae;
      attr_done.mark(dev);
    }
  }
  if (nt == 1024) PHX_LAUNCH_FAST(1024);
  else if (nt == 768) PHX_LAUNCH_FAST(768);
  else if (nt == 512) PHX_LAUNCH_FAST(512);
  else if (nt == 384) PHX_LAUNCH_FAST(384);
  else if (nt == 320) PHX_LAUNCH_FAST(320);
  else PHX_LAUNCH_FAST(256);
#undef PHX_LAUNCH_FAST
  return hipGetLastError();
}

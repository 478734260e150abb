// phx_sc_rollout_fsm.hip -- time-parallel rollouts of FiniteStateMachineEnv supply chains.
//
// Same division of labour as phx_sc_rollout_fast_kernel (phx_sc_rollout.hip): Philox draws and the trajectory are
// parallel over (time, pairs), one lane per (env, shop) walks only the stock recurrence.  What the FSM adds
// (fsm.py:253-380) is sequential in time only through the stage of each step and three "most recent" lookups:
//   * the stage of a step decides who acts, whose customers order, who observes and who is rewarded.  For an env whose
//     stage follows the handler-less chain from the initial stage (checked per env by phx_fsm_regular_check_kernel
//     before this kernel; any other env sends the whole launch to the lane-per-pair loop of phx_sc_fused.hip) the stage
//     is a function of the step's POSITION in its episode: the host tabulates it (DevSpec::fsm_pos_tab) together with
//   * the reward a shop emits when it observes = the cached reward of the most recent rewarded step of the episode
//     (self._rewards, fsm.py:334-350,378), the observation dumped at the episode's last step = the most recent
//     observation (self._observations, :349,360-375), and the state left behind (reward / observation cache,
//     delivered_stock) = the most recent rewarded / observing / acting step: each a LOOKBACK of at most FSM_LB steps,
//     also tabulated per position.  Everything a step derives is a pure function of (stock before, R, D), so the output
//     phase recomputes the looked-back step from its tile words -- from the previous chunk's last FSM_LB rows (kept by
//     the recurrence lane) when it lies before the chunk, from the state caches when it lies before the fragment.
// Draws mask the operands by the position's flags (no action: R = 0, no orders: D = 0), so the recurrence itself is
// the plain kernel's.  Results are bit-identical to the lane-per-pair loops (tests/test_gpu_round2.py, tests/fuzz_rollouts.py).
#include "phx_dev.h"
#include "phx_sc_fast.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define FSM_LB PHX_FSM_LB

struct FsmFastArgs {
  int32_t B, S, epb, G, K, T, num_steps, xcd_remap, first_rows, whole_envs;
  uint32_t pK; float inv_pK;
  uint32_t mG, mG4, mS, mPR, mNS;   // ceil(2^32 / d) magics: G, G / 4, S, 3 G / 4, num_steps   (i < 2^16)
  int32_t norm;
  uint64_t seed; int64_t env_offset;
  int32_t *stock, *sales, *missed, *delivered, *env_step, *env_tick, *env_stage, *env_prev_stage, *env_arrive;
  double* rew_cache; uint8_t* rew_cache_v; float* obs_cache; uint8_t* obs_cache_v;
  const uint32_t* pos_tab;          // [num_steps] see phx_api.hip: build_fsm_fast
  const int32_t* irregular;         // device word: == gen -> some env is off the tabulated stage chain, this kernel does nothing
  int32_t gen;                      // this launch's number (the check kernel stores it in *irregular: no clearing between launches)
  unsigned long long* timing;       // PHX_TIMING builds only
  phx_rollout_io io;
};

// pos_tab word of episode position p (the step that takes the env from step p to p + 1):
//   bit 0 the shops act, 1 their customers order, 2 the shops observe, 3 they are rewarded
//   bits 4..6   steps back to the most recent rewarded position <= p of the episode (7: none)
//   bits 8..10  steps back to the most recent observing position <= p of the episode (7: none); 0 at the last position (plan)
//   p as the LAST executed position of a fragment (state left behind), lookbacks across the episode boundary:
//   bits 12..14 most recent rewarded position, bit 15 it lies in the same episode as p
//   bits 16..18 most recent observing position, bits 20..22 most recent acting position   (7: none within FSM_LB)
//   bits 24..31 the stage at p
#define FP_ACT(w) ((w) & 1u)
#define FP_ORD(w) (((w) >> 1) & 1u)
#define FP_OBS(w) (((w) >> 2) & 1u)
#define FP_REW(w) (((w) >> 3) & 1u)

__device__ __forceinline__ void fsm_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// per env: is its stage the one the position table holds for its step?
__global__ void phx_fsm_regular_check_kernel(const int32_t* env_step, const int32_t* env_stage, const uint32_t* pos_tab,
                                             int num_steps, int gen, int B, int32_t* irregular) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int st = env_step[b], sg = env_stage[b];
  const bool ok = st >= 0 && st < num_steps && sg == (int)(pos_tab[st] >> 24);
  if (!ok) atomicExch(irregular, gen);
}

struct FsmRow { int sales, missed, xa, del; };
// what a step leaves behind, from the stock before it and its tile word R | D << 8 | act << 16 | ord << 17
// (supply_chain.py:93-122,136-142).  GUARD: a stock the caller set outside [0, 100]
template <bool GUARD>
__device__ __forceinline__ FsmRow fsm_derive(int x0, int rdw) {
  const int R = rdw & 255, D = (rdw >> 8) & 255;
  FsmRow o;
  o.del = min(R, PHX_SHOP_MAX_STOCK - x0);                       // decode_action's clamp
  if (!GUARD) { o.sales = min(x0, D); o.missed = D - o.sales; o.xa = x0 - o.sales + o.del; }
  else {
    const bool act = ((rdw >> 16) & 1) != 0, ord = ((rdw >> 17) & 1) != 0;
    o.sales = ord ? (D < x0 ? D : x0) : 0; o.missed = ord ? D - o.sales : 0;
    int x1 = x0 - o.sales;
    if (act) { const int nx = x1 + o.del; x1 = nx < PHX_SHOP_MAX_STOCK ? nx : PHX_SHOP_MAX_STOCK; }
    o.xa = x1;
  }
  return o;
}

typedef const __attribute__((address_space(4))) char* fsm_kptr_t;
#define a (*(const FsmFastArgs*)kp)
#define io (a.io)
#define FSM_REFRESH() asm volatile("" : "+s"(kp))
template <int NT>
__global__ __launch_bounds__(NT, NT / 64) void phx_sc_rollout_fsmfast_kernel(const FsmFastArgs a_) {
  fsm_kptr_t kp = (fsm_kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  FSM_REFRESH();
  if (*a.irregular == a.gen) return;      // (uniform) the lane-per-pair loop takes this launch
#ifdef PHX_TIMING
  unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define FTICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tm[k] += now_ - tprev; tprev = now_; } while (0)
#else
#define FTICK(k) do {} while (0)
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TC = PHX_FAST_TC, LB = FSM_LB;
  const int tid = threadIdx.x, nS = a.S, G = a.G, ns = a.num_steps;
  const int64_t total = (int64_t)a.B * nS;
  const int bid = xcd_block(a.xcd_remap != 0);
  const int64_t g_base = (int64_t)bid * G;
  const int64_t b_first = a.whole_envs ? (int64_t)bid * a.epb : g_base / nS;
  const int r0 = (int)(g_base - b_first * nS);
  const int n_env = a.whole_envs ? a.epb : (r0 + G - 1) / nS + 1;

  // ---- LDS carve -------------------------------------------------------------------------------------------------------
  const int G4p = (G + 3) & ~3, items = TC * G;
  uint32_t* s_pair = (uint32_t*)smem;                                   // [G] shop | env_local << 8
  uint8_t* s_ds = (uint8_t*)(s_pair + G4p);                             // [125] base-5 digit sums
  float* s_tabs = (float*)(s_ds + 128);                                 // [101] stock / 100
  float* s_tabn = s_tabs + 104;                                         // [5K + 1] x / norm
  double* s_pen = (double*)(s_tabn + 32);                               // [101] 0.1 * stock
  int* s_rd0 = (int*)(s_pen + 102);                                     // 3 x [TC][G]  R | D << 8 | act << 16 | ord << 17
  float* s_act0 = (float*)(s_rd0 + 3 * items);                          // 3 x [TC][G]  action
  int* s_xb0 = (int*)(s_act0 + 3 * items);                              // 2 x [TC][G]  stock before the step
  int* s_ptend0 = s_xb0 + 2 * items;                                    // 2 x [G]  chunk row that ends the pair's episode, or -1
  int* s_pstep0 = s_ptend0 + 2 * G4p;                                   // 2 x [G]  episode position of the chunk's first row
  float* s_ostage = (float*)(s_pstep0 + 2 * G4p);                       // [TC][G][3] observation pieces on their way out
  int* s_carry = (int*)(s_ostage + 3 * items);                          // 3 x [LB][G][2]  the chunk's last LB rows: stock before, word
  float* s_rc0 = (float*)(s_carry + 3 * LB * G4p * 2);                  // [G] self._rewards[aid] at launch, as emitted (f32)
  float* s_oc0 = s_rc0 + G4p;                                           // [3][G] self._observations[aid] at launch
  int* s_cv0 = (int*)(s_oc0 + 3 * G4p);                                 // [G] bit 0 reward cache valid, bit 1 observation cache valid
  uint32_t* s_pos = (uint32_t*)(s_cv0 + G4p);                           // [num_steps] position table
  int* s_tick0 = (int*)(s_pos + ((ns + 3) & ~3));                       // [epb] tick at launch
  int* s_pst0 = s_tick0 + ((a.epb + 3) & ~3);                           // [epb] episode position at launch
  int* s_flags = s_pst0 + ((a.epb + 3) & ~3);                           // [0] a tick is no multiple of 4, [1] a stock outside [0, 100]

  // ---- setup ------------------------------------------------------------------------------------------------------------
  int x = 0, step = 0;                    // lane state of the recurrence: stock, episode position
  {
    int tk = 0, ps = 0;
    const uint32_t pt = (uint32_t)(r0 + tid);
    const uint32_t el = nS == 1 ? pt : __umulhi(pt, a.mS);
    float rc = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f; int cv = 0;
    if (tid < G) {
      const int64_t g = g_base + tid;
      x = a.stock[g]; step = a.env_step[b_first + el];
      rc = (float)a.rew_cache[g]; cv = (a.rew_cache_v[g] ? 1 : 0) | (a.obs_cache_v[g] ? 2 : 0);
      o0 = a.obs_cache[g * 3]; o1 = a.obs_cache[g * 3 + 1]; o2 = a.obs_cache[g * 3 + 2];
    }
    if (tid < n_env) { tk = a.env_tick[b_first + tid]; ps = a.env_step[b_first + tid]; }
    for (int i = tid; i < ns; i += NT) s_pos[i] = a.pos_tab[i];
    if (tid < G) {
      s_pair[tid] = (pt - el * (uint32_t)nS) | (el << 8);
      s_rc0[tid] = rc; s_cv0[tid] = cv; s_oc0[tid] = o0; s_oc0[G4p + tid] = o1; s_oc0[2 * G4p + tid] = o2;
    }
    if (tid < 125) s_ds[tid] = (uint8_t)(tid % 5 + (tid / 5) % 5 + tid / 25);
    if (tid <= PHX_SHOP_MAX_STOCK) {
      s_tabs[tid] = (float)tid / (float)PHX_SHOP_MAX_STOCK;
      s_pen[tid] = __dmul_rn(0.1, (double)tid);
    }
    if (tid <= 5 * a.K) s_tabn[tid] = (float)tid / (float)a.norm;
    if (tid < 2) s_flags[tid] = 0;
    __syncthreads();
    if (tid < n_env) { s_tick0[tid] = tk; s_pst0[tid] = ps; if (tk & 3) s_flags[0] = 1; }
    if (tid < G && (unsigned)x > (unsigned)PHX_SHOP_MAX_STOCK) s_flags[1] = 1;
    __syncthreads();
  }
  const int quad_extra = s_flags[0];
  FTICK(0);
  const bool weird = s_flags[1] != 0;     // (an FSM step without action or orders leaves such a stock as it is: guarded for the whole launch)

  const int rec_threads = ((G + 63) >> 6) << 6;
  const int first_rows = a.first_rows;
  const int n_chunks = 1 + (a.T - first_rows + TC - 1) / TC;
  auto start_of = [&](int c) { return c == 0 ? 0 : first_rows + (c - 1) * TC; };
  auto rows_of = [&](int c) { const int left = a.T - start_of(c); return c == 0 ? first_rows : (left < TC ? left : TC); };

  // ---- draws of the chunk starting at step t0 (tc rows) into tile `buf`, by threads [first, NT) ---------------------------
  auto draws_impl = [&](int t0, int tc, int buf, int first, auto ALIGNED, auto K6) __attribute__((always_inline)) {
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
      const uint32_t tick_a = tick_base + (uint32_t)tla;
      uint32_t w[4];
      rng_block(a.seed, genv, tick_a, s, 0, 0, w);
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
      // episode position of tile row tla: (position at launch + t0 + tla) mod num_steps  (+ num_steps: tla >= -3)
      const uint32_t q = (uint32_t)(s_pst0[bl] + t0 + tla + ns);
      int pos = (int)(q - __umulhi(q, a.mNS) * (uint32_t)ns);
      int i = __mul24(tla, G) + gl;
#pragma unroll
      for (int h = 0; h < 4; ++h, i += G) {
        const uint32_t pw = s_pos[pos];
        pos = pos + 1 == ns ? 0 : pos + 1;
        if (!aligned) { const int tl = tla + h; if (tl < 0 || tl >= tc) continue; }
        uint32_t yy = y[h];
        if (!k6) yy -= __umul24((uint32_t)((float)yy * a.inv_pK), a.pK);
        const uint32_t hi = (uint32_t)((float)yy * 0.008f);
        const int D = (int)s_ds[hi] + (int)s_ds[__mul24((int)hi, -125) + (int)yy];
        const float action = rng_j_to_action(aj[h]);                     // recorded whether or not the shop acts
        s_act[i] = action;
        // the stage's masks folded into the operands: no action -> no request, no orders -> no demand
        s_rd[i] = (FP_ACT(pw) ? (int)rintf(action) : 0) | ((FP_ORD(pw) ? D : 0) << 8) | (int)((pw & 3u) << 16);
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

  // ---- the stock recurrence of chunk c (tc rows), one lane per pair -----------------------------------------------------
  int fin_xb = 0, fin_rd = 0;
  auto recurrence = [&](int c, int tc) __attribute__((always_inline)) {
    if (tid >= G) return;
    __builtin_amdgcn_s_setprio(3);
    int* rdw = s_rd0 + (c % 3) * items + tid;
    int* xb = s_xb0 + (c & 1) * items + tid;
    const int tend = ns - 1 - step;
    const bool ends = tend >= 0 && tend < tc;
    s_ptend0[(c & 1) * G4p + tid] = ends ? tend : -1;
    s_pstep0[(c & 1) * G4p + tid] = step;
    int* cy = s_carry + ((c % 3) * LB * G4p + tid) * 2;                  // [j][pair][2], j = 0 .. LB - 1 <-> rows tc - LB + j
    if (tc == TC && !weird) {
      int orig = 0;
      if (ends) { orig = rdw[tend * G]; rdw[tend * G] = 0xFF00; }       // episode reset folded into the operands (phx_sc_rollout.hip)
      int rd[TC];
#pragma unroll
      for (int h = 0; h < TC; ++h) rd[h] = rdw[h * G];
#pragma unroll
      for (int h = 0; h < TC; ++h) {
        xb[h * G] = x; fin_xb = x;
        if (h >= TC - LB) { cy[(h - (TC - LB)) * G4p * 2] = x; }
        x = max(x - ((rd[h] >> 8) & 255), 0) + min(rd[h] & 255, PHX_SHOP_MAX_STOCK - x);
      }
      if (ends) rdw[tend * G] = orig;
#pragma unroll
      for (int j = 0; j < LB; ++j) cy[j * G4p * 2 + 1] = (ends && tend == TC - LB + j) ? orig : rd[TC - LB + j];
      fin_rd = (ends && tend == TC - 1) ? orig : rd[TC - 1];
    } else {                                // a ragged last chunk, or an out-of-range stock at launch: the general step
      for (int h = 0; h < tc; ++h) {
        const int rdh = rdw[h * G];
        xb[h * G] = x; fin_xb = x; fin_rd = rdh;
        if (h >= tc - LB) { cy[(h - (tc - LB)) * G4p * 2] = x; cy[(h - (tc - LB)) * G4p * 2 + 1] = rdh; }
        x = (h == tend) ? 0 : fsm_derive<true>(x, rdh).xa;              // the caller's env.reset() zeroes the stock
      }
    }
    step += tc;
    if (ends) step -= ns;
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- outputs of chunk c ---------------------------------------------------------------------------------------------------
  auto outputs_impl = [&](int c, int t0, int tc, auto GUARD) __attribute__((always_inline)) {
    constexpr bool guard = decltype(GUARD)::value;
    const int first = rec_threads;
    if (tid < first) return;
    const int* s_rd = s_rd0 + (c % 3) * items;
    const float* s_act = s_act0 + (c % 3) * items;
    const int* s_xb = s_xb0 + (c & 1) * items;
    const int* s_ptend = s_ptend0 + (c & 1) * G4p;
    const int* s_pstep = s_pstep0 + (c & 1) * G4p;
    const int* s_cprev = s_carry + (((c + 2) % 3) * LB * G4p) * 2;      // chunk c - 1's last LB rows
    const int64_t row0 = (int64_t)t0 * total + g_base;
    char* const p_obs = (char*)(io.obs + row0 * 3);
    char* const p_rew = (char*)(io.reward + row0);
    char* const p_act = (char*)(io.action_out + row0);
    char* const p_tru = (char*)(io.truncated + row0);
    char* const p_ter = (char*)(io.terminated + row0);
    char* const p_ov = (char*)(io.obs_valid + row0);
    char* const p_rv = (char*)(io.reward_valid + row0);
    const uint32_t utotal = (uint32_t)total;
    const int G4 = G >> 2, nw = NT - first, n_units = tc * G4;
    const uint32_t PR = 3u * (uint32_t)G4;
    const uint32_t row_bytes = utotal * 12u;
    const int lane = tid & 63;
    auto obs_of = [&](const FsmRow& d, float* o) __attribute__((always_inline)) {
      if (!guard || ((unsigned)d.xa <= (unsigned)PHX_SHOP_MAX_STOCK && (unsigned)d.sales <= (unsigned)(5 * a.K) &&
                     (unsigned)d.missed <= (unsigned)(5 * a.K))) { o[0] = s_tabs[d.xa]; o[1] = s_tabn[d.sales]; o[2] = s_tabn[d.missed]; }
      else shop_obs_f32(d.xa, d.sales, d.missed, (float)a.norm, o);
    };
    auto rew_of = [&](const FsmRow& d) __attribute__((always_inline)) {
      return (!guard || (unsigned)d.xa <= (unsigned)PHX_SHOP_MAX_STOCK) ? __dsub_rn((double)d.sales, s_pen[d.xa]) : shop_reward(d.sales, d.xa);
    };
    for (int ub = (tid - first) - lane; ub < n_units; ub += nw) {
      const int u = ub + lane;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vrw = va;
      uint32_t tr = 0u, eo = 0u, vov = 0u, vrv = 0u;
      if (u < n_units) {
        const int r = (int)__umulhi((uint32_t)u, a.mG4);
        const int gl0 = (u - (int)__umul24(r, G4)) << 2, i0 = (int)__umul24(r, G) + gl0;
        const uint4 vr = *(const uint4*)(s_rd + i0), vx = *(const uint4*)(s_xb + i0), ve = *(const uint4*)(s_ptend + gl0),
                    vp = *(const uint4*)(s_pstep + gl0);
        va = *(const float4*)(s_act + i0);
        const int rdv[4] = {(int)vr.x, (int)vr.y, (int)vr.z, (int)vr.w}, xbv[4] = {(int)vx.x, (int)vx.y, (int)vx.z, (int)vx.w};
        const int tev[4] = {(int)ve.x, (int)ve.y, (int)ve.z, (int)ve.w}, psv[4] = {(int)vp.x, (int)vp.y, (int)vp.z, (int)vp.w};
        float o[12], rw[4];
        const bool last_row = (t0 + r == a.T - 1);
        // One straight path per pair, selects instead of branches (a wave's 64 units mix observing and silent steps and
        // units that straddle envs: with branches every wave walked every side -- 1.5 x the plain kernel's VALU work,
        // 330 instead of 200 us per launch).  Cold branches only for what a fragment meets once: a looked-back step that
        // lies before the fragment (the state caches) and the fragment's last row (last_obs).
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int p = psv[k] + r; if (p >= ns) p -= ns;
          const uint32_t pw = s_pos[p];
          const bool term = (r == tev[k]);                               // the episode's last step: truncations["__all__"], env.py:312-318
          const bool observes = FP_OBS(pw) != 0;                         // (the plan makes every episode's last position observe)
          const FsmRow own = fsm_derive<guard>(xbv[k], rdv[k]);
          float ob[3];
          obs_of(own, ob);                                               // fsm.py:328-332
          // the reward emitted with an observation: self._rewards[aid] = that of the most recent rewarded step of the episode
          // (fsm.py:378, :360-375), recomputed from that step's tile words
          const int lbr = (int)((pw >> 4) & 7u);
          const int rr = r - lbr;                                        // lbr == 7 (none yet this episode: None): any row, unused
          const int* src = (rr >= 0 || lbr == 7) ? (s_xb + (rr >= 0 ? rr : 0) * G + gl0 + k) : (s_cprev + ((LB + rr) * G4p + gl0 + k) * 2);
          const int* srd = (rr >= 0 || lbr == 7) ? (s_rd + (rr >= 0 ? rr : 0) * G + gl0 + k) : (s_cprev + ((LB + rr) * G4p + gl0 + k) * 2 + 1);
          const int xq = lbr == 0 ? xbv[k] : *src, rq = lbr == 0 ? rdv[k] : *srd;
          float rwk = (float)rew_of(fsm_derive<guard>(xq, rq));
          uint32_t rv = lbr == 7 ? 2u : 1u;
          if (__builtin_expect(c == 0 && rr < 0 && lbr != 7, 0)) {       // before the fragment: the launch state's cache
            const int cv = s_cv0[gl0 + k];
            rv = (cv & 1) ? 1u : 2u; rwk = s_rc0[gl0 + k];
          }
          if (rv == 2u) rwk = 0.f;
          o[3 * k] = observes ? ob[0] : 0.f; o[3 * k + 1] = observes ? ob[1] : 0.f; o[3 * k + 2] = observes ? ob[2] : 0.f;
          rw[k] = observes ? rwk : 0.f;
          tr |= (term ? 1u : 0u) << (8 * k); vov |= (observes ? 1u : 0u) << (8 * k); vrv |= (observes ? rv : 0u) << (8 * k);
          if (__builtin_expect(last_row && io.last_obs != nullptr, 0)) {
            // the observation the next fragment starts from: the one just emitted, or after an episode end the reset's
            // (a shop that acts in the initial stage observes its reset state: stock 0, the last step's sales)
            float lo[3] = {o[3 * k], o[3 * k + 1], o[3 * k + 2]};
            if (term) {
              lo[0] = lo[1] = lo[2] = 0.f;
              if (FP_ACT(s_pos[0])) shop_obs_f32(0, own.sales, own.missed, (float)a.norm, lo);
            }
            float* q = io.last_obs + (g_base + gl0 + k) * 3;
            q[0] = lo[0]; q[1] = lo[1]; q[2] = lo[2];
          }
        }
        vrw = make_float4(rw[0], rw[1], rw[2], rw[3]);
        eo = (uint32_t)r * utotal + (uint32_t)gl0;
        float4* st = (float4*)(s_ostage + 12 * u);
        st[0] = make_float4(o[0], o[1], o[2], o[3]); st[1] = make_float4(o[4], o[5], o[6], o[7]); st[2] = make_float4(o[8], o[9], o[10], o[11]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint32_t q0 = 3u * (uint32_t)ub, qn = 3u * (uint32_t)n_units;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const uint32_t q = q0 + (uint32_t)lane + 64u * (uint32_t)k;
        if (q < qn) {
          const uint32_t rr = __umulhi(q, a.mPR), pc = q - rr * PR;
          *(float4*)(p_obs + (size_t)(rr * row_bytes + pc * 16u)) = *(const float4*)(s_ostage + 4 * q);
        }
      }
      if (u < n_units) {
        *(float4*)(p_rew + (size_t)(eo * 4u)) = vrw;
        *(float4*)(p_act + (size_t)(eo * 4u)) = va;
        *(uint32_t*)(p_tru + (size_t)eo) = tr;
        *(uint32_t*)(p_ter + (size_t)eo) = 0u;
        *(uint32_t*)(p_ov + (size_t)eo) = vov;
        *(uint32_t*)(p_rv + (size_t)eo) = vrv;
      }
    }
  };
  auto outputs = [&](int c, int t0, int tc) __attribute__((always_inline)) {
    if (weird) outputs_impl(c, t0, tc, std::true_type{});
    else outputs_impl(c, t0, tc, std::false_type{});
  };

  // ---- schedule (phx_sc_rollout.hip): one barrier per chunk ------------------------------------------------------------------
  for (int it = -2; it < n_chunks; ++it) {
    FSM_REFRESH();
    const int co = it, cr = it + 1, cd = it + 2;
    if (tid < rec_threads && it > -2) {
      if (cr < n_chunks) recurrence(cr, rows_of(cr));
      FTICK(3);
    } else {
      if (co >= 0) { outputs(co, start_of(co), rows_of(co)); FTICK(5); }
      if (cd < n_chunks) draws(start_of(cd), rows_of(cd), cd % 3, it == -2 ? 0 : rec_threads);
      FTICK(1);
    }
    fsm_lds_barrier(); FTICK(6);
  }
#ifdef PHX_TIMING
  if (a.timing && (tid & 63) == 0 && blockIdx.x < 2048) for (int q = 0; q < 8; ++q) a.timing[((int64_t)blockIdx.x * (NT / 64) + (tid >> 6)) * 8 + q] = tm[q];
#endif

  // ---- state after the fragment ---------------------------------------------------------------------------------------------
  if (tid < G) {
    const int64_t g = g_base + tid;
    const FsmRow fin = fsm_derive<true>(fin_xb, fin_rd);
    a.stock[g] = x; a.sales[g] = fin.sales; a.missed[g] = fin.missed;
    int pl = step - 1; if (pl < 0) pl += ns;                            // position of the last executed step
    const uint32_t wl = s_pos[pl];
    const int cl = n_chunks - 1, sl = start_of(cl);
    const int* s_rdl = s_rd0 + (cl % 3) * items;
    const int* s_xbl = s_xb0 + (cl & 1) * items;
    const int* s_cpl = s_carry + (((cl + 2) % 3) * LB * G4p) * 2;
    // (stock before, word) of fragment row `row` (>= 0); false when the tiles no longer hold it
    auto frag_row = [&](int row, int& xq, int& rq) {
      const int rr = row - sl;
      if (rr >= 0) { xq = s_xbl[rr * G + tid]; rq = s_rdl[rr * G + tid]; return true; }
      if (cl > 0 && rr >= -LB) { const int* q = s_cpl + ((LB + rr) * G4p + tid) * 2; xq = q[0]; rq = q[1]; return true; }
      return false;
    };
    int xq, rq;
    const int la = (int)((wl >> 20) & 7u);                              // ShopAgent.delivered_stock: the most recent acting step
    if (la != 7 && a.T - 1 - la >= 0 && frag_row(a.T - 1 - la, xq, rq)) a.delivered[g] = fsm_derive<true>(xq, rq).del;
    const int lr = (int)((wl >> 12) & 7u);                              // self._rewards[aid]
    const bool inep = ((wl >> 15) & 1u) != 0;
    if (lr != 7 && a.T - 1 - lr >= 0 && frag_row(a.T - 1 - lr, xq, rq)) {
      const FsmRow d = fsm_derive<true>(xq, rq);
      a.rew_cache[g] = shop_reward(d.sales, d.xa);
      a.rew_cache_v[g] = inep ? 1 : 0;
    } else if (!inep) a.rew_cache_v[g] = 0;                             // an episode began inside the fragment: nothing cached yet
    const int lo = (int)((wl >> 16) & 7u);                              // self._observations[aid]
    if (lo != 7 && a.T - 1 - lo >= 0 && frag_row(a.T - 1 - lo, xq, rq)) {
      const FsmRow d = fsm_derive<true>(xq, rq);
      float ob[3];
      shop_obs_f32(d.xa, d.sales, d.missed, (float)a.norm, ob);
      a.obs_cache[g * 3] = ob[0]; a.obs_cache[g * 3 + 1] = ob[1]; a.obs_cache[g * 3 + 2] = ob[2];
      a.obs_cache_v[g] = 1;
    }
    const uint32_t pr = s_pair[tid];
    if ((pr & 255u) == 0u || (!a.whole_envs && tid == 0)) {
      // the env's words: by the block itself when it holds whole envs; with blocks of pair ranges by the block that FINISHES LAST
      // with the env (another block that holds a part of it may not have read them yet): every block counts itself in after its
      // own reads, the one that completes the count stores the words and clears the counter (phx_sc_rollout.hip, same rule)
      const int bl = (int)(pr >> 8);
      const int64_t b = b_first + bl, p0 = b * nS;
      const int n_touch = a.whole_envs ? 1 : (int)((p0 + nS - 1) / G - p0 / G) + 1;
      if (n_touch == 1 || atomicAdd(&a.env_arrive[b], 1) + 1 == n_touch) {
        a.env_step[b] = step;
        a.env_tick[b] = s_tick0[bl] + a.T;
        a.env_stage[b] = (int)(s_pos[step] >> 24);                      // step < num_steps: the stage the next step runs in
        a.env_prev_stage[b] = (int)(wl >> 24);
        if (n_touch > 1) a.env_arrive[b] = 0;
      }
    }
  }
}
#undef a
#undef io
#undef FSM_REFRESH

static uint32_t fsm_magic32(int d) { return (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)(d > 0 ? d : 1)); }

// the next launch generation of the env (see below: per env, never 0)
int32_t phx_fsm_next_gen(const DevSpec& sp) {
  std::atomic<int32_t>* gen_host = (std::atomic<int32_t>*)sp.fsm_gen_host;
  int32_t launch_gen = gen_host->fetch_add(1) + 1;
  if (launch_gen <= 0 || launch_gen == 0x7fffffff) { gen_host->store(1); launch_gen = 1; }
  return launch_gen;
}

// true when the launch was issued (the caller then issues the lane-per-pair loop guarded by DevSpec::fsm_irregular)
bool phx_launch_sc_rollout_fsmfast(const DevSpec& sp, const phx_rollout_io& io_, hipStream_t st, hipError_t* err, int32_t* gen_out) {
  *err = hipSuccess;
  const ScFastPlan& p = sp.fsm_fast;
  const int off = (phx_knobs().fsm_fast == 0);
  if (!p.ok || off || io_.actions || io_.exo || !io_.obs_valid || !io_.reward_valid) return false;
  if ((int64_t)io_.T + 2 * (int64_t)sp.num_steps >= 60000) return false;       // magic division of the position
  // Measured against the lane-per-pair loop (tools/roll_time.py --fsm, us per 100-step launch, this kernel / the loop): 9 shops x
  // 4 096 envs 40 / 60, x 16 384 120 / 85, x 65 536 501 / 464; 51 shops x 2 048 envs 104 / 77, x 4 096 191 / 119, x 8 192 330-370 /
  // 204-335.  The output phase costs twice the plain kernel's (own row + looked-back row per pair, silent and observing steps
  // mixed in every wave), so this kernel only wins where the loop's one lane per pair leaves the chip underfilled.
  const int force = phx_knobs().fsm_fast;             // development default
  if (force < 2 && sp.variant_rollout != PHX_VR_TIME_PARALLEL && (int64_t)sp.B * sp.S > 65536) return false;
  FsmFastArgs a;
  memset(&a, 0, sizeof a);
  a.B = sp.B; a.S = sp.S; a.epb = p.epb; a.G = p.G; a.whole_envs = p.whole_envs; a.K = p.K; a.T = io_.T; a.num_steps = sp.num_steps;
  a.xcd_remap = 1;
  uint32_t pk = 1; for (int k = 0; k < p.K; ++k) pk *= 5u;
  static const float inv[7] = {1.0f, 0.2f, 0.04f, 0.008f, 0.0016f, 0.00032f, 0.000064f};
  a.pK = pk; a.inv_pK = inv[p.K];
  a.mG = fsm_magic32(p.G); a.mG4 = fsm_magic32(p.G / 4); a.mS = fsm_magic32(sp.S); a.mPR = fsm_magic32(3 * (p.G / 4));
  a.mNS = fsm_magic32(sp.num_steps);
  a.norm = p.norm; a.seed = sp.seed; a.env_offset = sp.env_offset;
  a.first_rows = io_.T <= PHX_FAST_TC ? io_.T : PHX_FAST_TC;
  a.stock = (int32_t*)sp.f[F_SHOP_STOCK]; a.sales = (int32_t*)sp.f[F_SHOP_SALES];
  a.missed = (int32_t*)sp.f[F_SHOP_MISSED]; a.delivered = (int32_t*)sp.f[F_SHOP_DELIVERED];
  a.env_step = (int32_t*)sp.f[F_ENV_STEP]; a.env_tick = (int32_t*)sp.f[F_ENV_TICK];
  a.env_stage = (int32_t*)sp.f[F_ENV_STAGE]; a.env_prev_stage = (int32_t*)sp.f[F_ENV_PREV_STAGE]; a.env_arrive = (int32_t*)sp.f[F_ENV_ARRIVE];
  a.rew_cache = (double*)sp.f[F_ENV_REW_CACHE]; a.rew_cache_v = (uint8_t*)sp.f[F_ENV_REW_CACHE_VALID];
  a.obs_cache = (float*)sp.f[F_ENV_OBS_CACHE]; a.obs_cache_v = (uint8_t*)sp.f[F_ENV_OBS_CACHE_VALID];
  a.pos_tab = sp.fsm_pos_tab; a.irregular = sp.fsm_irregular;
  // the generation is per env (two envs, or two host threads, never share a counter) and is baked into the kernel
  // arguments: a launch captured into a hipGraph would replay ONE generation for ever, so a capturing stream takes the
  // lane-per-pair loop instead (same results)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
  const int32_t launch_gen = phx_fsm_next_gen(sp);
  a.gen = launch_gen; *gen_out = launch_gen;
  a.io = io_;
  a.timing = nullptr;
#ifdef PHX_TIMING
  { static unsigned long long* tbuf = nullptr; if (!tbuf) (void)hipMalloc((void**)&tbuf, 8 * 8 * 2048 * sizeof(unsigned long long)); a.timing = tbuf;
    if (getenv("PHX_TIMING_DUMP")) { static int calls = 0; if (++calls == 20) { (void)hipDeviceSynchronize(); static unsigned long long h[8 * 8 * 2048]; (void)hipMemcpy(h, tbuf, sizeof h, hipMemcpyDeviceToHost);
      const int wpb = p.nt / 64, nw = 2048 * wpb; double sum[8] = {0}, w0[8] = {0}; for (int w = 0; w < nw; ++w) for (int q = 0; q < 8; ++q) { sum[q] += h[(size_t)w * 8 + q]; if (w % wpb == 0) w0[q] += h[(size_t)w * 8 + q]; }
      fprintf(stderr, "FSMF_TIMING avg cycles per wave:  setup %.0f | draws %.0f | rec %.0f | out %.0f | bar %.0f\n", sum[0]/nw, sum[1]/nw, sum[3]/nw, sum[5]/nw, sum[6]/nw);
      fprintf(stderr, "FSMF_TIMING wave0 of each block:  setup %.0f | draws %.0f | rec %.0f | out %.0f | bar %.0f\n", w0[0]*wpb/nw, w0[1]*wpb/nw, w0[3]*wpb/nw, w0[5]*wpb/nw, w0[6]*wpb/nw); } } }
#endif
  const int G4p = (p.G + 3) & ~3, items = PHX_FAST_TC * p.G, epb4 = (p.epb + 3) & ~3;
  const size_t lds = (size_t)G4p * 4 + 128 + 104 * 4 + 32 * 4 + 102 * 8 + (size_t)items * 4 * (6 + 2 + 3) + (size_t)G4p * 4 * 4 +
                     (size_t)3 * FSM_LB * G4p * 8 + (size_t)G4p * 4 * 5 + (size_t)((sp.num_steps + 3) & ~3) * 4 + (size_t)epb4 * 8 + 32;
  if (lds > 40 * 1024) return false;
  if (!p.whole_envs && !a.env_arrive) return false;           // blocks of pair ranges need the per-env arrival counter
  hipLaunchKernelGGL(phx_fsm_regular_check_kernel, dim3((sp.B + 255) / 256), dim3(256), 0, st, a.env_step, a.env_stage, a.pos_tab,
                     sp.num_steps, a.gen, sp.B, sp.fsm_irregular);
  const dim3 grid((unsigned)(((int64_t)sp.B * sp.S) / p.G));
  const int nt = p.nt;
  phx_note_kernel(p.whole_envs ? "phx_sc_rollout_fsmfast_kernel[whole_envs]" : "phx_sc_rollout_fsmfast_kernel[pairs]");
  if (nt == 512) hipLaunchKernelGGL((phx_sc_rollout_fsmfast_kernel<512>), grid, dim3(512), lds, st, a);
  else if (nt == 384) hipLaunchKernelGGL((phx_sc_rollout_fsmfast_kernel<384>), grid, dim3(384), lds, st, a);
  else if (nt == 320) hipLaunchKernelGGL((phx_sc_rollout_fsmfast_kernel<320>), grid, dim3(320), lds, st, a);
  else hipLaunchKernelGGL((phx_sc_rollout_fsmfast_kernel<256>), grid, dim3(256), lds, st, a);
  *err = hipGetLastError();
  return true;
}

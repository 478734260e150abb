// phx_sc_rollout_sw.hip -- the time-parallel supply-chain rollout kernel, round-4 structure: DEDICATED STORE WAVES.
//
// Same path and the same trajectories as phx_sc_rollout_fast_kernel (phx_sc_rollout.hip): T consecutive PhantomEnv.step()
// calls of a FACTORY / SHOP / CUSTOMER env (supply_chain.py:36-150, env.py:239-303) per launch, random policy and device-RNG
// orders, auto-reset at episode end (the list-of-envs loop of utils/rllib/rollout.py:361-363).  What round 3 left on the
// table (Weak #4): the worker waves of that kernel compute a chunk's outputs and store them themselves, a wave
// stalled on store back-pressure computes nothing, and 44 bytes of LDS per item kept the workgroups narrow (48 pairs), which
// is the slow end of the store pattern.  Measured before this file was written (tools/ubench/ub_store10.hip): a persistent
// grid whose workgroups hand their finished chunk to a few waves that do nothing but `ds_read_b128 -> global_store_dwordx4`
// (1 KB of consecutive 16-byte pieces per instruction) writes the trajectory at 0.82 of 8 TB/s with 144-pair workgroups (one
// per CU, four store waves), dense flag planes included -- against 0.62-0.75 for the old pattern without the flag planes.
// So here
//   * a workgroup owns G consecutive (env, shop) pairs (G % 16 == 0; SC64, B = 4096: 144 pairs = one workgroup per CU) and
//     walks the fragment in chunks of TC rows, one LDS-only barrier per chunk;
//   * WORKER waves draw chunk c + 2 (one Philox block per (pair, tick quad); the order sum comes from ONE table lookup on
//     y < 5^K; the action is staged in LDS beside R | D) and compute the outputs of chunk c from 4 bytes of
//     LDS per item (R | D << 8 and stock-before | stock-after << 8, both u16) into a staged tile laid out like the
//     trajectory rows (observation [TC][3 G] f32, reward [TC][G] f32; the f32 reward comes from a table on 10 sales - stock
//     of the f64 expression, built on the host) -- they never issue a trajectory store;
//   * RECURRENCE waves (one lane per pair) walk the stock chain of chunk c + 1 and store stock before AND after each step;
//   * STORE waves stream the staged tile of chunk c - 1 to HBM (non-temporal, whole 64-byte units); the flag planes of the WHOLE fragment
//     they write before that, in the iterations in which the pipeline fills (dense 16-byte pieces, a closed form of the pairs' step
//     counters at launch: no zero-fill launch before the kernel, no scattered flag stores in the steady state).
// LDS: 10 bytes per item of tiles + 32 of staging (double-buffered) instead of 44 + nothing staged.
// Round 5: fragment lists, several pair groups per workgroup, the REPLAY (MODE 1) and FSM (MODE 2) instantiations, the workers' fused
// phase, one store piece per trip.
// Development macros (never defined in the product build; tools/build_variant.py / scratch variants): PHX_TIMING, PHX_RT_ONLY, PHX_RT_FILL
// (cycle and wall-clock stamps per role), PHX_ABL_NODRAW / PHX_ABL_NOSTORE, SW_ABL_NODTAB / SW_ABL_NORTAB (LDS bank-conflict attribution),
// SWF_ABL_NOTRACK / NOFLAGS / PLAINREC / NOPW (what each part of the FSM instantiation costs), SW_NO_FIXED_DRAWS / SW_NO_FIXED_OUT /
// SW_NO_FUSED_WORK, SW_STORE_DEPTH.
#include "phx_dev.h"
#include "phx_sc_fast.h"

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

struct SwFrag { float* obs; float* action_out; float* reward; uint8_t* terminated; uint8_t* truncated; uint8_t* obs_valid; uint8_t* reward_valid; };
struct SwArgs {
  int32_t B, S, epb, G, K, T, num_steps, xcd_remap, first_rows;
  int32_t n_rec_waves, n_store_waves;   // wave roles: [0, n_rec) recurrence, [n_rec, n_rec + n_store) stores, the rest work
  int32_t dtab_n;                       // 5^K entries of the order-sum table
  uint32_t pK; float inv_pK;            // 5^K and its f32 reciprocal
  uint32_t mG, mG4, mS, mPO, mPF;       // ceil(2^32 / d) magics: i / G, i / (G / 4), i / S, i / (3 G / 4), i / (G / 16)   for i < 2^16
  int32_t norm;
  uint64_t seed; int64_t env_offset;
  int32_t *stock, *sales, *missed, *delivered, *env_step, *env_tick, *env_arrive;
  const float4* tables;                 // the host-built image of the table sections (phx_sc_sw_tables)
  phx_rollout_io io;
  // ---- (past the argument lines the kernel warms at entry) ----
  int32_t n_groups;                     // pair groups of the launch (B S / G): workgroup w walks groups w, w + gridDim.x, .. one after the other
  int32_t frag_T, n_frag;               // rows per trajectory fragment, fragments (frag_T * n_frag == T; one fragment: frag_T == T)
  // replayed inputs (REPLAY instantiations): io.actions [T][B][S] and / or io.exo [T][B][n_exo]
  int32_t n_exo, guard_gen;             // exogenous columns per env; this launch's number for `guard`
  const int32_t* exo_first;             // [S] exogenous column of each shop's first customer (its K customers' columns are consecutive)
  const int32_t* guard;                 // device word: == guard_gen -> a replayed action rounds below zero (the pre-scan of this call found
                                        // one): the stock would leave [0, 100], this kernel does nothing and round 1's kernel serves the call
  SwFrag frag[PHX_MAX_FRAGMENTS];       // row t of the launch is row t - f * frag_T of fragment f = t / frag_T
  // FSM instantiation (FiniteStateMachineEnv supply chains on their handler-less stage chain, fsm.py:253-380)
  const uint16_t* fsm_tab;              // [2][num_steps]: the SWF_* word of every episode position, then the stage it runs in
  int32_t *env_stage, *env_prev_stage;
  double* rew_cache; uint8_t* rew_cache_v; float* obs_cache; uint8_t* obs_cache_v;      // self._rewards / self._observations (fsm.py:334-350)
  unsigned long long* timing;           // PHX_TIMING builds only
  unsigned long long* rt; int32_t launch_idx;   // PHX_TIMING builds only: 100 MHz wall-clock stamps per workgroup and launch
};
// The word of an episode position p (the step that takes the env from step p to p + 1) in the stage the chain runs it in: the stage's
// flags at the bits they have in the tile word, and the masks of the operands it switches off (no action: R = 0, no orders: D = 0) --
//   tile word of the FSM instantiation = ((R | D << 8) | SWF_FLAGS) & position word        (R <= 100, D <= 24)
#define SWF_RMASK 0x007Fu   /* the shops act (StockRequest): R passes                         */
#define SWF_OBS 0x0080u     /* the shops observe: they act in the NEXT stage, fsm.py:320      */
#define SWF_DMASK 0x1F00u   /* their customers order: D passes                                */
#define SWF_REW 0x2000u     /* the shops are rewarded (self._rewards is updated, :335,350)    */
#define SWF_ACT 0x4000u     /* the shops act                                                  */
#define SWF_HASREW 0x8000u  /* a rewarded position <= p exists in the episode (never in a tile word) */
#define SWF_FLAGS (SWF_OBS | SWF_REW | SWF_ACT)
// What a row hands to the workers (s_fo, one u32 per item, written by the recurrence lane): everything the row emits as table indices --
//   stock after | sales << 7 | missed sales << 12 | reward index << 17 | SWF_FO_LAUNCH
// a silent row (no observation) and an observing row with nothing cached emit zeros: index 0 of the observation tables and
// SWF_RW_ZERO of the reward table hold 0.0f.  SWF_FO_LAUNCH: the reward is the cache as the launch found it (s_rc0).
#define SWF_RW_ZERO 100
#define SWF_RW_LAUNCH 0x200  /* in the lane's running index: bit 26 of the s_fo word */
#define SWF_RW_CACHED 0x400  /* in the lane's running index: set by a rewarded step of the episode (bit 27 of the s_fo word, unused there) */
#define SWF_FO_LAUNCH (1u << 26)

__device__ __forceinline__ void sw_lds_barrier() {       // orders LDS traffic only: the trajectory stores stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// LDS bytes of a workgroup (host and device agree through this one function)
// copies of the observation tables (one per LDS bank of a 32-lane group; development: SW_NC48 = 16 halves them for the 48-pair shape, whose
// workgroup then fits a CU twice -- tools/build_variant.py; product build: 32 everywhere)
#ifndef SW_NC48
#define SW_NC48 32
#endif
__host__ __device__ inline int sw_ncopy(int G) { return G == 48 ? SW_NC48 : 32; }
__host__ __device__ inline size_t sw_lds_bytes(int G, int epb, int TC, int dtab_n, int fsm_ns = 0) {
  const size_t G4p = (size_t)((G + 3) & ~3), items = (size_t)TC * G;
  const size_t fsm = fsm_ns > 0 ? items * 2 * 2 + G4p * 4 + (size_t)((G + 15) & ~15) + (size_t)((epb + 3) & ~3) * 4 + (size_t)((2 * fsm_ns + 15) & ~15) : 0;   // (FSM sections, below)
  return fsm + G4p * 4 + (size_t)((epb + 3) & ~3) * 4 + 16            // pair table, ticks, flags
       + (size_t)(101 + 32) * sw_ncopy(G) * 4 + 401 * 8 * 4 + 128  // observation tables (32 copies), reward table (8 copies), digit sums of k < 125
       + 15632                                                   // order sums of y < 5^K (5^6 reserved: the tables sit at fixed offsets)
       + items * 2 * 3 + items * 2 * 2                           // R | D tiles (3), stock tiles (2)
       + (size_t)((G + 15) & ~15) * 3 + 16                       // episode-end rows (3) + pad
       + G4p * 4                                                 // launch stocks (out-of-range stocks only)
       + items * 16 * 2                                          // staged observation + reward, double-buffered
       + items * 4 * 2;                                          // staged action (drawn one iteration before it is stored), double-buffered
}

// the host-built table image: BASE values of the three value tables (replicated per LDS bank inside the kernel), then the digit
// sums and the order sums exactly as they sit in LDS
#define SW_IMG_TABS 0                          /* f32 [104]: stock / 100                      */
#define SW_IMG_TABN 104                        /* f32 [32]:  x / norm                         */
#define SW_IMG_RTAB 136                        /* f32 [404]: f32(n / 10), n = index - 100     */
#define SW_IMG_BASE_BYTES ((104 + 32 + 404) * 4)
#define SW_TABLE_BYTES (SW_IMG_BASE_BYTES + 128 + 15632)
static_assert(SW_IMG_BASE_BYTES % 16 == 0 && SW_TABLE_BYTES % 16 == 0, "table image is copied in 16-byte pieces");

typedef const __attribute__((address_space(4))) char* sw_kptr_t;
// Trajectory stores are NON-TEMPORAL: a store wave writes whole 128-byte lines (64 consecutive 16-byte pieces per instruction), nothing
// reads the fragment during the launch, and lines that bypass the L2 are not left to be written back at the end of the kernel
// (measured: 1-2 us per T = 400 launch; round 2's kernel, whose lanes wrote partial lines, was twice as slow with such stores).
typedef float sw_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sw_store16(char* p, const float4 v) { __builtin_nontemporal_store((sw_f4v){v.x, v.y, v.z, v.w}, (sw_f4v*)p); }
// REPLAY, a hint the caller vouched for that does not hold (phx_rollout_io.hints): an action that rounds below zero under
// PHX_RH_ACTIONS_IN_DOMAIN (its sign lands in `rq_or`, the OR of the quad's rounded requests), an order byte >= 5 under PHX_RH_EXO_IN_DOMAIN
// (`xbad`: per raw dword (w & 0xF8F8F8F8) | ((w + 0x03030303) & 0x08080808) -- a byte >= 8, or one in [5, 8) where no byte carries).  The
// env's rows are unspecified then, but not silently: err[b] = PHX_ERR_HINT (sticky like every soft error; lanes of one env write the same code).
__device__ __forceinline__ uint32_t sw_exo_bad(uint32_t w) { return (w & 0xF8F8F8F8u) | ((w + 0x03030303u) & 0x08080808u); }
__device__ __forceinline__ void sw_hint_violation(int32_t* err, int64_t b, int rq_or, uint32_t xbad) {
  if (__builtin_expect((rq_or < 0) | (xbad != 0u), 0)) { if (err && err[b] == 0) err[b] = PHX_ERR_HINT; }
}
#define a (*(const SwArgs*)kp)
#define io (a.io)
#define SW_REFRESH() asm volatile("" : "+s"(kp))

// GT > 0: the workgroup shape is a compile-time constant (GT pairs, NREC recurrence waves, NSTORE store waves): every
// index computation on G folds, the kernel keeps fewer uniform values alive (the generic instantiation spills ~60 SGPRs).
// REPLAY: the policy's actions and / or the customers' order sizes come from HBM (phx_rollout_io.actions / exo: a recorded policy,
// the reference's own numpy stream -- phx_mt_draw) instead of the Philox block: the draw phase loads them (4 S and S K more bytes
// read per env-step) and the rest of the pipeline is the same.
// MODE 2 (FSM): a FiniteStateMachineEnv supply chain whose envs are all on the handler-less stage chain (checked per launch by
// phx_sw_fsm_check_kernel; otherwise this kernel returns at entry and the lane-per-pair loop serves the launch).  The stage of a step
// is a function of its episode position; its masks fold into the tile word's operands (no action: R = 0, no orders: D = 0), the
// recurrence lane carries what fsm.py's caches hold (self._rewards / self._observations, delivered_stock) and hands the reward a row
// emits to the workers as a table index; obs_valid / reward_valid are closed forms like the truncation plane.
template <int TC, int GT, int NREC, int NSTORE, int NWORK, int MODE>
__global__ __launch_bounds__(1024) void phx_sc_rollout_sw_kernel(const SwArgs a_) {
  constexpr bool REPLAY = MODE == 1, FSM = MODE == 2;
  sw_kptr_t kp = (sw_kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  { uint32_t d0, d1, d2, d3, d4;          // every 64-byte line of the argument block into the scalar cache at once, not one miss per phase (setup 1.6 -> 1.2 us)
    static_assert(offsetof(SwArgs, n_groups) > 0x100 && offsetof(SwArgs, n_groups) <= 0x140, "the offsets below cover the argument block line by line (the fragment table is read once per chunk)");
    asm volatile("s_load_dword %0, %5, 0x0\n s_load_dword %1, %5, 0x40\n s_load_dword %2, %5, 0x80\n s_load_dword %3, %5, 0xc0\n s_load_dword %4, %5, 0x100\n s_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4) : "s"(kp) : "memory"); }
  SW_REFRESH();
  if ((REPLAY || FSM) && a.guard && *a.guard == a.guard_gen) return;    // (uniform) an action outside the kernel's domain: round 1's kernel takes the call;
                                                                        // FSM: an env off the stage chain or a stock outside [0, 100]: the lane-per-pair loop does
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, NT = GT ? 64 * (NREC + NSTORE + NWORK) : (int)blockDim.x, nS = a.S, G = GT ? GT : a.G;
  const int64_t total = (int64_t)a.B * nS;
  // ---- LDS carve (16-byte aligned sections; sw_lds_bytes) ---------------------------------------------------------
  const int G4p = (G + 3) & ~3, G16p = (G + 15) & ~15, items = TC * G;
  // The value tables are REPLICATED so that a lane's lookup lands in the lane's own LDS bank (ds_read_b32: 32 banks, lane
  // groups of 32): entry v of copy c at dword v * 32 + c, lane l reads copy l & 31 -- no bank conflicts whatever the values.
  float* s_tabs = (float*)smem;                                         // [101][32] stock / 100        encode_observation,
  constexpr int NC = (GT == 48) ? SW_NC48 : 32, NCL = (NC == 32) ? 7 : 6;      // table copies; log2 of an entry's bytes
  static_assert(NC == 32 || NC == 16, "SW_NC48: 32 or 16");
  float* s_tabn = s_tabs + 101 * NC;                                    // [32][32]  x / norm           supply_chain.py:124-134
  // compute_reward (:147): f32(f64 sales - 0.1 * stock) depends on n = 10 * sales - stock only and equals the f32 quotient n / 10
  // for every reachable (sales <= 30, stock <= 100) (tests/test_host_logic.py); 8 copies: lanes l, l + 8, .. share one
  float* s_rtab = s_tabn + 32 * NC;                                     // [401][8]  f32(n / 10), n = 10 * sales - stock + 100
  uint8_t* s_ds = (uint8_t*)(s_rtab + 401 * 8);                         // [125] base-5 digit sum of k < 5^3
  uint8_t* s_dtab = s_ds + 128;                                         // [5^K <= 15625] digit sum of y: the customers' order sizes summed
  uint32_t* s_pair = (uint32_t*)(s_dtab + 15632);                       // [G] shop | env_local << 8
  int* s_tick0 = (int*)(s_pair + G4p);                                  // [epb]
  int* s_flags = s_tick0 + ((a.epb + 3) & ~3);                          // [4] per wave 0..3: bit 0 a tick is not a multiple of 4, 1 a stock outside [0, 100], 2 a step counter < 0
  uint16_t* s_rd0 = (uint16_t*)(s_flags + 4);                           // 3 x [TC][G]  R | D << 8                     (chunk c in c % 3)
  uint16_t* s_xx0 = s_rd0 + 3 * items;                                  // 2 x [TC][G]  stock before | stock after << 8 (chunk c in c & 1)
  uint32_t* s_fo0 = (uint32_t*)s_xx0;                                   // FSM: 2 x [TC][G] u32 in place of the stock tiles (chunk c in c & 1)
  uint16_t* s_ftend = (uint16_t*)(s_xx0 + (FSM ? 4 : 2) * items);       // [G] the row of the fragment that ends the pair's current episode (0xFFFF: none): the flag planes
  int* s_x0w = (int*)((uint8_t*)s_ftend + 3 * G16p + 16);                        // [G] stocks at launch (used when one is outside [0, 100])
  float* s_out0 = (float*)(s_x0w + G4p);                                // 2 x { obs [TC][3 G], reward [TC][G] }   (chunk c in c & 1)
  float* s_act0 = s_out0 + 8 * items;                                   // 2 x [TC][G] action, drawn at iteration c - 2, stored at c - 1 (chunk c in c & 1)
  // FSM sections
  float* s_rc0 = s_act0 + 2 * items;                                    // [G] self._rewards[shop] at launch, as emitted (f32)
  uint8_t* s_cv0 = (uint8_t*)(s_rc0 + G4p);                             // [G] is it valid
  int* s_pst0 = (int*)(s_cv0 + G16p);                                   // [epb] episode position of the env at launch
  uint16_t* s_pf = (uint16_t*)(s_pst0 + ((a.epb + 3) & ~3));            // [num_steps] SWF_* word of every episode position

  // i / d for the block's divisors: a literal where the shape is compile-time, the host's magic otherwise (i < 2^16)
  auto div_G = [&](uint32_t i) { return GT ? i / (uint32_t)(GT ? GT : 1) : __umulhi(i, a.mG); };
  auto div_G4 = [&](uint32_t i) { return GT ? i / (uint32_t)(GT ? GT / 4 : 1) : __umulhi(i, a.mG4); };
  auto div_PO = [&](uint32_t i) { return GT ? i / (uint32_t)(GT ? 3 * (GT / 4) : 1) : __umulhi(i, a.mPO); };
  auto div_PF = [&](uint32_t i) { return GT ? i / (uint32_t)(GT ? GT / 16 : 1) : (a.mPF ? __umulhi(i, a.mPF) : i); };
  const int n_store_waves = GT ? NSTORE : a.n_store_waves;
  const int rec_threads = (GT ? NREC : a.n_rec_waves) << 6, store_first = rec_threads, work_first = rec_threads + (n_store_waves << 6);
#ifdef PHX_TIMING
  unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
  unsigned long long rts[8]; rts[6] = rts[7] = 0; rts[0] = __builtin_amdgcn_s_memrealtime();
#define RSTAMP(k) do { rts[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RSTAMP(k) do {} while (0)
#endif
#if defined(PHX_TIMING) && !defined(PHX_RT_ONLY)     /* PHX_RT_ONLY: the wall-clock stamps alone (six per workgroup: no perturbation to speak of) */
#define STICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tm[k] += now_ - tprev; tprev = now_; } while (0)
#else
#define STICK(k) do {} while (0)
#endif

  // ---- the workgroup's pair groups, one after the other (a launch of more groups than resident workgroups: replacing a finished
  //      1 024-thread, 160 KB workgroup by the next costs the CU 3-4 us -- tools/ubench/ub_anyorder.hip -- and the tables would be
  //      staged again; round 4 launched one workgroup per group).  Virtual workgroup vb = blockIdx.x + k gridDim.x keeps the XCD of
  //      blockIdx.x (gridDim.x % 8 == 0 or a single pass) and the XCD-contiguous ranges of xcd_block().
  for (int vb = (int)blockIdx.x, pass = 0; vb < a.n_groups; vb += (int)gridDim.x, ++pass) {
  int bid = vb;
  if (a.xcd_remap) { const unsigned n = (unsigned)a.n_groups, xq = (unsigned)vb & 7u, q = n >> 3, rem = n & 7u; bid = (int)(xq * q + (xq < rem ? xq : rem) + ((unsigned)vb >> 3)); }
  const int64_t g_base = (int64_t)bid * G;
  const int64_t b_first = g_base / nS;
  const int r0 = (int)(g_base - b_first * nS);                          // the first pair's shop
  const int n_env = (r0 + G - 1) / nS + 1;                              // envs the block touches (<= a.epb)

  // ---- setup: the state loads are in flight while the tables are computed ------------------------------------------
  int x = 0, step = 0;                    // lane state of the recurrence (lane tid owns pair g_base + tid)
  {
    int tk = 0;
    const uint32_t pt = (uint32_t)(r0 + tid);
    const uint32_t el = nS == 1 ? pt : __umulhi(pt, a.mS);               // (r0 + tid) / S
    if (tid < G) { x = a.stock[g_base + tid]; step = a.env_step[b_first + el]; }
    if (tid < n_env) tk = a.env_tick[b_first + tid];
    float f_rc = 0.f; int f_cv = 0, f_ps = 0;
    if (FSM) {
      if (tid < G) { f_rc = (float)a.rew_cache[g_base + tid]; f_cv = a.rew_cache_v[g_base + tid] ? 1 : 0; }
      if (tid < n_env) f_ps = a.env_step[b_first + tid];
      if (pass == 0) for (int i = tid; i < a.num_steps; i += NT) s_pf[i] = a.fsm_tab[i];
    }
    // the tables: the host-built image (17.9 KB, L2-resident after the first workgroups), loads issued before anything waits.  Digit
    // and order sums go where they live; the base values of the three value tables go to a scratch area (the staging tile, unused
    // until the first output phase) and are replicated per LDS bank after the barrier.  (The replicated tables as a 45.6 KB image
    // cost every workgroup ~2.4 k more cycles of setup: ~11 bytes per cycle and CU when all 256 workgroups fetch at once.)
    if (pass == 0) {
      constexpr int NP = SW_TABLE_BYTES / 16, NB = SW_IMG_BASE_BYTES / 16;
      float4* const scratch = (float4*)s_out0;
      float4* const sums = (float4*)s_ds;
      auto put = [&](int i, const float4 v) { if (i < NB) scratch[i] = v; else sums[i - NB] = v; };
      // two named pieces per thread (1 024 threads: the whole image), both loads in flight before either is written; narrow
      // workgroups loop over the rest.  (An indexed array of pieces here went through scratch memory, one load at a time.)
      const int i0 = tid, i1 = tid + NT;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (i0 < NP) v0 = a.tables[i0];
      if (i1 < NP) v1 = a.tables[i1];
      if (i0 < NP) put(i0, v0);
      if (i1 < NP) put(i1, v1);
      for (int i = tid + 2 * NT; i < NP; i += NT) put(i, a.tables[i]);
    }
    if (tid < G) {
      s_pair[tid] = (pt - el * (uint32_t)nS) | (el << 8); s_x0w[tid] = x;
      const int et = a.num_steps - 1 - step;
      s_ftend[tid] = (uint16_t)((step < a.num_steps && et < 0xFFFF) ? et : 0xFFFF);      // (a counter >= num_steps never ends an episode: recurrence)
    }
    if (tid < n_env) s_tick0[tid] = tk;
    if (FSM) { if (tid < G) { s_rc0[tid] = f_rc; s_cv0[tid] = (uint8_t)f_cv; } if (tid < n_env) s_pst0[tid] = f_ps; }
    // launch-wide flags without a zeroing pass: the waves that can hold a pair or an env (the first four) publish theirs
    {
      const bool f0 = tid < n_env && (tk & 3) != 0;                                       // a tick that is not a multiple of 4
      const bool f1 = tid < G && (unsigned)x > (unsigned)PHX_SHOP_MAX_STOCK;              // a stock outside [0, 100]
      const bool f2 = tid < G && step < 0;                                                // a step counter below zero
      const int wf = (__ballot(f0) != 0ull ? 1 : 0) | (__ballot(f1) != 0ull ? 2 : 0) | (__ballot(f2) != 0ull ? 4 : 0);
      if ((tid & 63) == 0 && tid < 256) s_flags[tid >> 6] = wf;
    }
    STICK(6);
    __syncthreads();
    STICK(7);
  }
  STICK(0); RSTAMP(1);
  int launch_flags = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) if (k * 64 < NT) launch_flags |= s_flags[k];
  // the value tables, one copy per LDS bank: written during the loop's first iteration by the waves that have nothing else to do
  // there (recurrence and store waves); the outputs first read them two barriers later
  auto replicate_tables = [&](int t, int nthr) __attribute__((always_inline)) {
    // entry v of copy c at dword v * copies + c: the copies of one entry are contiguous -- one 4-byte read, 16-byte writes
    const float* const bv = (const float*)s_out0;
    float4* const w32 = (float4*)s_tabs;                                 // s_tabs [101][32] and s_tabn [32][32] are adjacent: 133 entries x 8 pieces
    float4* const w8 = (float4*)s_rtab;                                  // [401][8]: 2 pieces per entry
    constexpr int N32 = (101 + 32) * (NC / 4), N8 = 401 * 2;
    for (int i = t; i < N32 + N8; i += nthr) {
      const bool wide = i < N32;
      const int e = wide ? i / (NC / 4) : (i - N32) >> 1;
      const float v = bv[wide ? (e < 101 ? SW_IMG_TABS + e : SW_IMG_TABN + (e - 101)) : SW_IMG_RTAB + e];
      (wide ? w32 + i : w8 + (i - N32))[0] = make_float4(v, v, v, v);
    }
  };
  const int quad_extra = launch_flags & 1;      // chunk starts are not quad-aligned for every env: one more row quad
  const bool weird = (launch_flags & 2) != 0;   // a stock the caller set outside [0, 100] (any step brings it back into range)
  const bool neg_steps = (launch_flags & 4) != 0;   // a step counter the caller set below zero: its first episode end is num_steps - 1 - step rows away

  const int first_rows = a.first_rows;
  // (two SHORT chunks at the head to fill the pipeline sooner were measured: every iteration costs ~1.2 us whatever its rows, +1-2 us)
  const int n_chunks = 1 + (a.T - first_rows + TC - 1) / TC;
  auto start_of = [&](int c) { return c == 0 ? 0 : first_rows + (c - 1) * TC; };
  auto rows_of = [&](int c) { const int left = a.T - start_of(c); return c == 0 ? first_rows : (left < TC ? left : TC); };
  const uint32_t utotal = (uint32_t)total;

  // ---- draws of the chunk starting at step t0 (tc rows) into tile `buf`, by the worker waves.
  //      One Philox block serves ticks 4q .. 4q + 3 of a shop: work items are (row quad jr, pair gl).  The action goes
  //      straight to its plane (4 bytes per lane, 256 contiguous bytes per wave and row); R | D << 8 to the tile.
  //      FIXED: the worker threads are a multiple of G, so a lane draws for the SAME pair in every item and every chunk --
  //      shop, env and tick come from registers set once per launch instead of two dependent LDS lookups per item (an LDS
  //      round trip is ~200 cycles in which the wave does nothing else: the draw phase had seven of them per item).
  const int nwk = NT - work_first, wt = tid - work_first;               // worker threads, this thread's rank among them
#ifdef SW_NO_FIXED_DRAWS
  const bool draws_fixed = false;
#else
  const bool draws_fixed = (nwk % G) == 0;
#endif
  int dl_gl = 0, dl_jr0 = 0, dl_s = 0, dl_pos0 = 0; uint32_t dl_tick0 = 0; int64_t dl_genv = 0;
  // FSM: q mod num_steps for q < 2^17 (the f32 quotient is off by one at most)
  const uint32_t ns_u = (uint32_t)a.num_steps; const float inv_ns_f = 1.0f / (float)a.num_steps;
  auto mod_ns = [&](uint32_t q) { uint32_t r = q - (uint32_t)((float)q * inv_ns_f) * ns_u; if ((int)r < 0) r += ns_u; if (r >= ns_u) r -= ns_u; return r; };
  const bool rp_act = REPLAY && io.actions != nullptr, rp_exo = REPLAY && io.exo != nullptr;
  if (wt >= 0) {
    dl_jr0 = (int)div_G((uint32_t)wt); dl_gl = wt - dl_jr0 * G;
    const uint32_t pr = s_pair[dl_gl];
    dl_s = (int)(pr & 255u);
    dl_genv = a.env_offset + b_first + (int)(pr >> 8);
    dl_tick0 = (uint32_t)s_tick0[pr >> 8];
    if (FSM) dl_pos0 = s_pst0[pr >> 8];      // (fixed draws: carried forward chunk by chunk -- the episode position of the next chunk's first row)
  }
  auto draws_impl = [&](int t0, int tc, int buf, int buf2, auto ALIGNED, auto FIXED) __attribute__((always_inline)) {
    constexpr bool aligned = decltype(ALIGNED)::value, fixed = decltype(FIXED)::value;
    uint16_t* s_rd = s_rd0 + buf * items;
    float* const s_act = s_act0 + (buf2) * items;
    const int n_work = (((tc + 3) >> 2) + (aligned ? 0 : quad_extra)) * G;
    const bool k6 = a.K == 6;
    const int djr = fixed ? nwk / G : 0;
    int jr = dl_jr0;
    for (int iw = wt; iw < n_work; iw += nwk, jr += djr) {
      int gl = dl_gl, s = dl_s, pos0 = dl_pos0; int64_t genv = dl_genv; uint32_t tick_base = dl_tick0 + (uint32_t)t0;
      if (!fixed) {
        jr = (int)div_G((uint32_t)iw); gl = iw - (int)__umul24(jr, G);
        const uint32_t pr = s_pair[gl];
        s = (int)(pr & 255u);
        genv = a.env_offset + b_first + (int)(pr >> 8);
        tick_base = (uint32_t)s_tick0[pr >> 8] + (uint32_t)t0;
        if (FSM) pos0 = s_pst0[pr >> 8];
      }
      const int tla = 4 * jr - (aligned ? 0 : (int)(tick_base & 3u));
      if (!aligned && (tla + 3 < 0 || tla >= tc)) continue;
      const uint32_t tick_a = tick_base + (uint32_t)tla;                 // multiple of 4
      // replayed inputs first: their loads are in flight while the Philox block (if one is needed) is computed
      float av[4] = {0.f, 0.f, 0.f, 0.f};
      int Dx[4] = {0, 0, 0, 0};
      uint32_t xbad = 0u; int rq_or = 0;
      if (REPLAY) {
        const int64_t pair = g_base + gl;
        if (rp_act) {
#pragma unroll
          for (int h = 0; h < 4; ++h) { const int tl = tla + h; if (aligned || (tl >= 0 && tl < tc)) av[h] = io.actions[(int64_t)(t0 + tl) * total + pair]; }
        }
        if (rp_exo) {
          const int64_t b = genv - a.env_offset;
          const int K_ = a.K, e0 = a.exo_first[s];
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int tl = tla + h;
            if (aligned || (tl >= 0 && tl < tc)) {
              const uint8_t* row = io.exo + ((int64_t)(t0 + tl) * a.B + b) * a.n_exo + e0;
              // CustomerAgent.generate_messages: the recorded np.random.randint(5) draws of the shop's K customers, supply_chain.py:61-67.
              // K >= 4: two (unaligned) dword loads that stay inside the shop's K bytes -- the first four and the last four -- instead of
              // K byte loads; a byte sum is one multiply (every draw is < 5)
              int d = 0;
              if (K_ >= 4) {
                uint32_t w0, w1;
                __builtin_memcpy(&w0, row, 4); __builtin_memcpy(&w1, row + (K_ - 4), 4);
                xbad |= sw_exo_bad(w0) | sw_exo_bad(w1);
                if (K_ > 4) w0 += (w1 >> (8 * (8 - K_))) ;             // the K - 4 bytes w0 does not hold: byte-wise add (no carries: draws < 5)
                d = (int)((w0 * 0x01010101u) >> 24);
              } else for (int k = 0; k < K_; ++k) { d += (int)row[k]; xbad |= row[k] >= 5 ? 1u : 0u; }
              Dx[h] = d;
            }
          }
        }
      }
      // FSM: the words of the four rows' episode positions ((position at launch + row) mod num_steps; tla >= -3): the lookups are in
      // flight while the Philox block is computed
      uint32_t pw[4] = {0u, 0u, 0u, 0u};
#ifndef SWF_ABL_NOPW
      if (FSM)
#else
      if (false)
#endif
      {
        uint32_t pos;
        if (fixed) { int p = pos0 + tla; if (p < 0) p += (int)ns_u; else if (p >= (int)ns_u) p -= (int)ns_u; pos = (uint32_t)p; }     // (-3 <= tla < tc <= num_steps)
        else pos = mod_ns((uint32_t)(pos0 + t0 + tla) + ns_u);
#pragma unroll
        for (int h = 0; h < 4; ++h) { pw[h] = s_pf[pos]; pos = pos + 1u == ns_u ? 0u : pos + 1u; }
      }
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      uint32_t y[4] = {0u, 0u, 0u, 0u}, aj[4] = {0u, 0u, 0u, 0u};
      if (!(rp_act && rp_exo)) {
#ifdef PHX_ABL_NODRAW
      w[0] = tick_a * 2654435761u + (uint32_t)genv; w[1] = w[0] * 40503u + s; w[2] = w[1] ^ 0x9e3779b9u; w[3] = w[2] + w[0];   // dev ablation: no Philox
#else
      rng_block(a.seed, genv, tick_a, s, 0, 0, w);
#endif
      bool rej = false;
#pragma unroll
      for (int h = 0; h < 4; ++h) rej |= !rng_split(w[h], y[h], aj[h]);
      if (__builtin_expect(rej, 0)) {                                    // probability 3.3e-6 per word: one cold branch
#pragma unroll 1
        for (int h = 0; h < 4; ++h) {
          uint32_t y2, j2;
          if (!rng_split(w[h], y2, j2)) y[h] = rng_group_y(a.seed, genv, tick_a + (uint32_t)h, s, 0, 1, &aj[h]);
        }
      }
      }
      // the four order sums in ONE LDS round trip (the lookups are issued together, then the four items are finished)
      int D[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        uint32_t yy = y[h];
        if (!k6) yy -= __umul24((uint32_t)((float)yy * a.inv_pK), a.pK);  // the first K base-5 digits: y mod 5^K
#ifdef SW_ABL_NODTAB     /* attribution of the LDS bank conflicts: the order sum by arithmetic instead of the byte gather (same value) */
        D[h] = rp_exo ? Dx[h] : rng_digit_sum(yy, a.K, nullptr);
#else
        D[h] = rp_exo ? Dx[h] : (int)s_dtab[yy];                         // the customers' order sizes summed, supply_chain.py:61-67
#endif
      }
      const int i = __mul24(tla, G) + gl;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        if (!aligned) { const int tl = tla + h; if (tl < 0 || tl >= tc) continue; }
        const float action = rp_act ? av[h] : rng_j_to_action(aj[h]);    // the replayed policy, or the random one: [0, 100)
        // decode_action: int(round(action)), supply_chain.py:139.  A replayed action >= -0.5 (the call's pre-scan) rounds to R >= 0 and
        // min(R, 100 - stock) is the same for every R >= 100: 255 stands for all of them in the tile's byte
        const int Rq = rp_act ? (int)fminf(rintf(action), 255.0f) : (int)rintf(action);
#ifdef SWF_ABL_NOPW
        if (false)
#else
        if (FSM)        // the stage's masks folded into the operands (no action: no request; no orders: no demand), its flags beside them
#endif
          s_rd[i + h * G] = (uint16_t)((((uint32_t)Rq | ((uint32_t)D[h] << 8)) | SWF_FLAGS) & pw[h]);
        else
        s_rd[i + h * G] = (uint16_t)(Rq | (D[h] << 8));
        s_act[i + h * G] = action;
        if (REPLAY) rq_or |= Rq;
      }
      if (REPLAY) sw_hint_violation(io.err, genv - a.env_offset, rq_or, xbad);
    }
    if (FSM && fixed) { dl_pos0 += tc; if (dl_pos0 >= (int)ns_u) dl_pos0 -= (int)ns_u; }
  };
  auto draws = [&](int t0, int tc, int c) __attribute__((always_inline)) {
    if (wt < 0) return;
    if (!draws_fixed) draws_impl(t0, tc, c % 3, c & 1, std::false_type{}, std::false_type{});
    else if (!quad_extra && (tc & 3) == 0) draws_impl(t0, tc, c % 3, c & 1, std::true_type{}, std::true_type{});
    else draws_impl(t0, tc, c % 3, c & 1, std::false_type{}, std::true_type{});
  };

  // ---- the stock recurrence of chunk c (tc rows), one lane per pair ------------------------------------------------
  //   stock' = max(stock - D, 0) + min(R, 100 - stock)      handle_order_request / handle_stock_response,
  //   supply_chain.py:98-122 with decode_action's clamp :139; at the episode's last step the caller's env.reset() zeroes
  //   the stock (ShopAgent.reset): the tile word of that step is PATCHED to D = 255, R = 0 before the burst read (stock'
  //   = 0 for stock <= 100, no select on the chain) and restored after it.  Per step one u16 is stored: the stock before
  //   the step and the stock after it; the episode's last row gets its true stock-after (what the last observation
  //   shows) after the chain.
  int fin_xb = 0, fin_rd = 0;             // the launch's last step: stock before it and its packed (R, D)
  auto recurrence = [&](int c, int tc) __attribute__((always_inline)) {
    const int tend = a.num_steps - 1 - step;                            // chunk row that ends the episode (one at most: TC <= num_steps)
    const bool ends = tend >= 0 && tend < tc;
    if (tid >= G) return;
    __builtin_amdgcn_s_setprio(3);        // a dependent chain beside waves of Philox / output work: issue whenever ready
    uint16_t* rdw = s_rd0 + (c % 3) * items + tid;
    uint16_t* xx = s_xx0 + (c & 1) * items + tid;
    if (tc == TC && !(weird && c == 0)) {   // straight-line code, the chunk's operands fetched in one burst
      int orig = 0;
      if (ends) { orig = rdw[tend * G]; rdw[tend * G] = 0xFF00; }
      int rd[TC];
#pragma unroll
      for (int h = 0; h < TC; ++h) rd[h] = rdw[h * G];
#pragma unroll
      for (int h = 0; h < TC; ++h) {
        const int xn = max(x - (rd[h] >> 8), 0) + min(rd[h] & 255, PHX_SHOP_MAX_STOCK - x);
        xx[h * G] = (uint16_t)(x | (xn << 8)); fin_xb = x;
        x = xn;
      }
      fin_rd = rd[TC - 1];
      if (ends) {                           // the true stock after the episode's last step (the chain carried the reset)
        rdw[tend * G] = (uint16_t)orig;
        const int xb = xx[tend * G] & 255;
        const int xa = max(xb - (orig >> 8), 0) + min(orig & 255, PHX_SHOP_MAX_STOCK - xb);
        xx[tend * G] = (uint16_t)(xb | (xa << 8));
        if (tend == TC - 1) fin_rd = orig;
      }
    } else {                                // a ragged last chunk, or an out-of-range stock at launch (general form, capped)
      for (int h = 0; h < tc; ++h) {
        const int rdh = rdw[h * G];
        const int xa = min(max(x - (rdh >> 8), 0) + min(rdh & 255, PHX_SHOP_MAX_STOCK - x), PHX_SHOP_MAX_STOCK);
        xx[h * G] = (uint16_t)((x & 255) | (xa << 8)); fin_xb = x; fin_rd = rdh;
        x = (ends && h == tend) ? 0 : xa;
      }
    }
    step += tc;
    if (ends) step -= a.num_steps;
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- FSM: the same chain with the caches of fsm.py carried along (one lane per pair, the rows in order):
  //   self._rewards[shop] (fsm.py:334-350): set at every rewarded step, cleared by reset (:234) -- what a row emits with its observation
  //   (:378, or the terminal dump :360-375) goes to the workers as the reward table's index (in s_fo);  self._observations[shop] (:349) and
  //   ShopAgent.delivered_stock (supply_chain.py:109-113, set when the shop acts) are only needed as the state the fragment leaves.
  //   The episode's last row: its word's REW bit updates the cache BEFORE the row emits, the reset after it.  No stock outside [0, 100]
  //   and no step counter outside [0, num_steps) reaches this instantiation (phx_sw_fsm_check_kernel).
  int f_rws = 0, f_rew = -1, f_obs = -1, f_del = -1; bool fin_term = false;      // running reward index; the s_fo words of the most recent rewarded / observing row; ..
  if (FSM && tid < G) f_rws = s_cv0[tid] ? SWF_RW_LAUNCH : SWF_RW_ZERO;
  // (The rows cost the chain 19 VALU operations each against the plain one's 7, and every one of them shows in the launch's time --
  //  the ablations in DESIGN.md: flags become lane masks by one v_bfe_i32 and select by one v_bfi_b32; the reset of an episode's last row
  //  and the trackers of the state the fragment leaves are compiled in only for the chunks that need them: a wave none of whose lanes
  //  ends an episode in the chunk, and every chunk but the fragment's last two -- the plan admits stage chains only whose acting /
  //  observing / rewarded positions are at most 16 steps apart, so the most recent of each lies in those chunks or before the fragment.)
  auto recurrence_fsm = [&](int c, int tc) __attribute__((always_inline)) {
    const int tend = a.num_steps - 1 - step;
    const bool ends = tend >= 0 && tend < tc;
    if (tid >= G) return;
    __builtin_amdgcn_s_setprio(3);
    const uint16_t* rdw = s_rd0 + (c % 3) * items + tid;
    uint32_t* fo = s_fo0 + (c & 1) * items + tid;
    auto sel = [](int m, int yes, int no) { return (m & yes) | (~m & no); };                       // v_bfi_b32
    auto one = [&](int h, int w, auto ENDS, auto TRACK) __attribute__((always_inline)) {
      const int R = w & 127, D = (w >> 8) & 31;
      const int sales = min(x, D), del = min(R, PHX_SHOP_MAX_STOCK - x), xn = x - sales + del;
      const int pack = xn | (sales << 7) | ((D - sales) << 12);          // what the row would observe, and the operands of its reward
      fin_xb = x; fin_rd = w;
      const int m_rew = __builtin_amdgcn_sbfe(w, 13, 1), m_obs = __builtin_amdgcn_sbfe(w, 7, 1);  // all ones where the row is rewarded / observes
      f_rws = sel(m_rew, 10 * sales - xn + (100 + SWF_RW_CACHED), f_rws);
      fo[h * G] = (uint32_t)sel(m_obs, pack | (f_rws << 17), SWF_RW_ZERO << 17);
      if (decltype(TRACK)::value) {
        f_rew = sel(m_rew, pack, f_rew); f_obs = sel(m_obs, pack, f_obs);
        f_del = sel(__builtin_amdgcn_sbfe(w, 14, 1), del, f_del);
      }
      x = xn;
      if (decltype(ENDS)::value) { const bool last = h == tend; x = last ? 0 : xn; f_rws = last ? SWF_RW_ZERO : f_rws; }      // the caller's env.reset(): stock 0, nothing cached
    };
    auto chunk = [&](auto ENDS, auto TRACK) __attribute__((always_inline)) {
      if (tc == TC) {
        int rd[TC];
#pragma unroll
        for (int h = 0; h < TC; ++h) rd[h] = rdw[h * G];
#pragma unroll
        for (int h = 0; h < TC; ++h) one(h, rd[h], ENDS, TRACK);
      } else for (int h = 0; h < tc; ++h) one(h, rdw[h * G], ENDS, TRACK);
    };
    const bool track = c >= n_chunks - 2, any_end = __ballot(ends) != 0ull;
    if (track) chunk(std::true_type{}, std::true_type{});
    else if (any_end) chunk(std::true_type{}, std::false_type{});
    else chunk(std::false_type{}, std::false_type{});
    fin_term = ends && tend == tc - 1;
    step += tc;
    if (ends) step -= a.num_steps;
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- outputs of chunk c into the staged tile, by the workers -------------------------------------------------------
  // A work unit is 4 consecutive pairs of one tile row: 12 observation floats and 4 rewards, from the two u16 tiles and
  // the tables; written to LDS in the layout of the trajectory rows (the store waves copy whole rows of pieces).  Where
  // the worker threads are a multiple of G / 4, a lane keeps its column for the whole launch (no division per unit).
  const int G4 = G >> 2;
#ifdef SW_NO_FIXED_OUT
  const bool out_fixed = false;
#else
  const bool out_fixed = (nwk % G4) == 0;
#endif
  const int ol_r0 = wt >= 0 ? (int)div_G4((uint32_t)wt) : 0, ol_gl0 = wt >= 0 ? (wt - ol_r0 * G4) << 2 : 0;
  auto outputs = [&](int c, int tc) __attribute__((always_inline)) {
    if (wt < 0) return;
    // typed views indexed in whole 8- / 16-byte elements from the (16-byte aligned) start of the LDS: the compiler then
    // knows the alignment of every access (b64 reads, b128 writes) although the section offsets are run-time values
    const uint2* const t_rd = (const uint2*)smem + (((int)((const char*)s_rd0 - smem) + (c % 3) * items * 2) >> 3);
    const uint2* const t_xx = (const uint2*)smem + (((int)((const char*)s_xx0 - smem) + (c & 1) * items * 2) >> 3);
    const uint4* const t_fo = (const uint4*)smem + (((int)((const char*)s_fo0 - smem) + (c & 1) * items * 4) >> 4);
    float4* const o_obs4 = (float4*)smem + (((int)((const char*)s_out0 - smem) >> 4) + (c & 1) * items);
    float4* const o_rew4 = o_obs4 + 3 * (items >> 2);
    const int n_units = tc * G4, dr = out_fixed ? nwk / G4 : 0;
    const bool guard = !FSM && weird && c == 0;
    const uint32_t c32 = ((uint32_t)tid & (uint32_t)(NC - 1)) << 2, c8 = ((uint32_t)tid & 7u) << 2;      // the lane's table copies (byte offsets)
    const char* const t_s = (const char*)s_tabs; const char* const t_n = (const char*)s_tabn; const char* const t_r = (const char*)s_rtab;
    int r = ol_r0;
    for (int u = wt; u < n_units; u += nwk, r += dr) {
      int gl0 = ol_gl0;
      if (!out_fixed) { r = (int)div_G4((uint32_t)u); gl0 = (u - (int)__umul24(r, G4)) << 2; }
      const int j4 = (int)__umul24(r, G4) + (gl0 >> 2);                  // the unit's index: its four items start at 4 * j4
      uint2 vx = make_uint2(0u, 0u), vr = vx;
      uint4 vf = make_uint4(0u, 0u, 0u, 0u);
      if (FSM) vf = t_fo[j4]; else { vx = t_xx[j4]; vr = t_rd[j4]; }     // FSM: one word per item holds everything the row emits
      // (measured, dropped: waves whose units are all silent -- every second row of a RESTOCK / SELL chain, with the rows dealt by parity --
      //  writing zeros without the lookups: +32 us per T = 400 launch of config 3; the iteration ends with its slowest wave either way)
      const uint32_t xw[4] = {vx.x & 0xffffu, vx.x >> 16, vx.y & 0xffffu, vx.y >> 16};
      const uint32_t rw4[4] = {vr.x & 0xffffu, vr.x >> 16, vr.y & 0xffffu, vr.y >> 16};
      const uint32_t fo4[4] = {vf.x, vf.y, vf.z, vf.w};
      float o[12], rw[4];
      if (!guard) {
        // sixteen table lookups in ONE LDS round trip: the addresses first, then the loads
        uint32_t as_[4], an_[4], am_[4], ar_[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (FSM) {
            const uint32_t w = fo4[k];
            as_[k] = ((w & 127u) << NCL) | c32; an_[k] = (((w >> 7) & 31u) << NCL) | c32; am_[k] = (((w >> 12) & 31u) << NCL) | c32;
            ar_[k] = (((w >> 17) & 0x1FFu) << 5) | c8;                  // the cached reward the row emits
            continue;
          }
          const int x0 = (int)(xw[k] & 255u), xa = (int)(xw[k] >> 8), D = (int)(rw4[k] >> 8);
          const int sales = min(x0, D), missed = D - sales;             // handle_order_request :105-122 (sales == x0 - max(x0 - D, 0))
          as_[k] = ((uint32_t)xa << NCL) | c32; an_[k] = ((uint32_t)sales << NCL) | c32; am_[k] = ((uint32_t)missed << NCL) | c32;
          ar_[k] = ((uint32_t)(10 * sales - xa + 100) << 5) | c8;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o[3 * k] = *(const float*)(t_s + as_[k]);                     // encode_observation :124-134
          o[3 * k + 1] = *(const float*)(t_n + an_[k]);
          o[3 * k + 2] = *(const float*)(t_n + am_[k]);
#ifdef SW_ABL_NORTAB     /* attribution: the reward as the f32 quotient (the same value, tests/test_host_logic.py) instead of the 8-copy table */
          rw[k] = (float)((int)(ar_[k] >> 5) - 100) / 10.0f;
#else
          rw[k] = *(const float*)(t_r + ar_[k]);                        // compute_reward :147, rounded once to f32
#endif
        }
        if (FSM && __builtin_expect(((fo4[0] | fo4[1] | fo4[2] | fo4[3]) & SWF_FO_LAUNCH) != 0u, 0)) {      // rows before the fragment's first rewarded step
#pragma unroll
          for (int k = 0; k < 4; ++k) if (fo4[k] & SWF_FO_LAUNCH) rw[k] = s_rc0[gl0 + k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int x0 = (int)(xw[k] & 255u);
          const int xa = (int)(xw[k] >> 8), D = (int)(rw4[k] >> 8);
          if (r == 0) x0 = s_x0w[gl0 + k];                               // the launch's first row: the stock as the caller left it
          const int sales = min(x0, D), missed = D - sales;
          if ((unsigned)x0 <= (unsigned)PHX_SHOP_MAX_STOCK) {
            o[3 * k] = s_tabs[xa * NC]; o[3 * k + 1] = s_tabn[sales * NC]; o[3 * k + 2] = s_tabn[missed * NC];
            rw[k] = s_rtab[(10 * sales - xa + 100) << 3];
          } else {                                                      // a stock the caller set outside [0, 100]: the formulas
            float ob[3];
            shop_obs_f32(xa, sales, missed, (float)a.norm, ob);
            o[3 * k] = ob[0]; o[3 * k + 1] = ob[1]; o[3 * k + 2] = ob[2];
            rw[k] = (float)shop_reward(sales, xa);
          }
        }
      }
      float4* so = o_obs4 + 3 * j4;
      so[0] = make_float4(o[0], o[1], o[2], o[3]); so[1] = make_float4(o[4], o[5], o[6], o[7]); so[2] = make_float4(o[8], o[9], o[10], o[11]);
      o_rew4[j4] = make_float4(rw[0], rw[1], rw[2], rw[3]);
    }
  };

  // ---- workers, FUSED form of outputs(co) + draws(cd) for the compile-time shapes in the steady state (every lane: exactly one unit of
  //      four items and one row quad per chunk): the two phases touch different tiles, so their LDS round trips and the Philox block
  //      can overlap -- tile reads of chunk co are in flight while the block of chunk cd is computed, the sixteen table lookups while
  //      its words are split, the four order-sum lookups while the staged tile is written.  (Separately the output phase is ~100
  //      instructions around two exposed LDS round trips, ~20 cycles per instruction.)  The one-in-3e5 rejected word is repaired after the fact.
#ifdef SW_NO_FUSED_WORK
  const bool fused_ok = false;
#else
  // (the FSM instantiation, already at 127 VGPRs, loses with it: config 3, T = 400 894 against 850 us)
  // (replays: of the actions only -- 64.7 against 67.0 us per T = 400; with the order sizes too the fused form loses, 85.9 against 83.7)
  const bool fused_mode = MODE == 0 || (MODE == 1 && !rp_exo);
  const bool fused_ok = fused_mode && GT > 0 && TC == 16 && draws_fixed && out_fixed && !quad_extra && nwk == TC * G4 && nwk == (TC / 4) * G;
#endif
  auto work_fused = [&](int co, int cd, int t0d) __attribute__((always_inline)) {
    if (wt < 0) return;
    // (1) outputs: the unit's tile words
    const uint2* const t_rd = (const uint2*)smem + (((int)((const char*)s_rd0 - smem) + (co % 3) * items * 2) >> 3);
    const uint2* const t_xx = (const uint2*)smem + (((int)((const char*)s_xx0 - smem) + (co & 1) * items * 2) >> 3);
    float4* const o_obs4 = (float4*)smem + (((int)((const char*)s_out0 - smem) >> 4) + (co & 1) * items);
    float4* const o_rew4 = o_obs4 + 3 * (items >> 2);
    const int j4 = (int)__umul24(ol_r0, G4) + (ol_gl0 >> 2);
    const int tla = 4 * dl_jr0;
    // (0) REPLAY: the recorded inputs of the quad first -- the longest latency of the phase
    float av[4] = {0.f, 0.f, 0.f, 0.f}; int Dx[4] = {0, 0, 0, 0};
    uint32_t xbad = 0u; int rq_or = 0;
    if (REPLAY) {
      const int64_t pair = g_base + dl_gl;
      if (rp_act) {
#pragma unroll
        for (int h = 0; h < 4; ++h) av[h] = io.actions[(int64_t)(t0d + tla + h) * total + pair];
      }
      if (rp_exo) {
        const int64_t b = dl_genv - a.env_offset;
        const int K_ = a.K, e0 = a.exo_first[dl_s];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const uint8_t* row = io.exo + ((int64_t)(t0d + tla + h) * a.B + b) * a.n_exo + e0;
          int d = 0;
          if (K_ >= 4) {
            uint32_t w0, w1;
            __builtin_memcpy(&w0, row, 4); __builtin_memcpy(&w1, row + (K_ - 4), 4);
            xbad |= sw_exo_bad(w0) | sw_exo_bad(w1);
            if (K_ > 4) w0 += (w1 >> (8 * (8 - K_)));
            d = (int)((w0 * 0x01010101u) >> 24);
          } else for (int k = 0; k < K_; ++k) { d += (int)row[k]; xbad |= row[k] >= 5 ? 1u : 0u; }
          Dx[h] = d;
        }
      }
    }
    const uint2 vx = t_xx[j4], vr = t_rd[j4];
    // (2) draws: the Philox block of the lane's (pair, row quad) of chunk cd
    uint16_t* const s_rd = s_rd0 + (cd % 3) * items;
    float* const s_act = s_act0 + (cd & 1) * items;
    const uint32_t tick_a = dl_tick0 + (uint32_t)t0d + (uint32_t)tla;
    uint32_t w[4] = {0u, 0u, 0u, 0u}, y[4] = {0u, 0u, 0u, 0u}, aj[4] = {0u, 0u, 0u, 0u};
    bool rej = false;
    if (!(rp_act && rp_exo)) {
      rng_block(a.seed, dl_genv, tick_a, dl_s, 0, 0, w);
#pragma unroll
      for (int h = 0; h < 4; ++h) rej |= !rng_split(w[h], y[h], aj[h]);
    }
    // (3) outputs: sixteen table lookups
    const uint32_t c32 = ((uint32_t)tid & (uint32_t)(NC - 1)) << 2, c8 = ((uint32_t)tid & 7u) << 2;
    const char* const t_s = (const char*)s_tabs; const char* const t_n = (const char*)s_tabn; const char* const t_r = (const char*)s_rtab;
    const uint32_t xw[4] = {vx.x & 0xffffu, vx.x >> 16, vx.y & 0xffffu, vx.y >> 16};
    const uint32_t rw4[4] = {vr.x & 0xffffu, vr.x >> 16, vr.y & 0xffffu, vr.y >> 16};
    uint32_t as_[4], an_[4], am_[4], ar_[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x0 = (int)(xw[k] & 255u), xa = (int)(xw[k] >> 8), D = (int)(rw4[k] >> 8);
      const int sales = min(x0, D), missed = D - sales;
      as_[k] = ((uint32_t)xa << NCL) | c32; an_[k] = ((uint32_t)sales << NCL) | c32; am_[k] = ((uint32_t)missed << NCL) | c32;
      ar_[k] = ((uint32_t)(10 * sales - xa + 100) << 5) | c8;
    }
    float o[12], rw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[3 * k] = *(const float*)(t_s + as_[k]); o[3 * k + 1] = *(const float*)(t_n + an_[k]); o[3 * k + 2] = *(const float*)(t_n + am_[k]);
      rw[k] = *(const float*)(t_r + ar_[k]);
    }
    // (4) draws: the four order sums
    const bool k6 = a.K == 6;
    int D[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      uint32_t yy = y[h];
      if (!k6) yy -= __umul24((uint32_t)((float)yy * a.inv_pK), a.pK);
      D[h] = rp_exo ? Dx[h] : (int)s_dtab[yy];
    }
    // (5) outputs: the staged tile
    float4* so = o_obs4 + 3 * j4;
    so[0] = make_float4(o[0], o[1], o[2], o[3]); so[1] = make_float4(o[4], o[5], o[6], o[7]); so[2] = make_float4(o[8], o[9], o[10], o[11]);
    o_rew4[j4] = make_float4(rw[0], rw[1], rw[2], rw[3]);
    // (6) draws: the tile words and the actions
    const int i = __mul24(tla, G) + dl_gl;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float action = rp_act ? av[h] : rng_j_to_action(aj[h]);
      const int Rq = rp_act ? (int)fminf(rintf(action), 255.0f) : (int)rintf(action);
      s_rd[i + h * G] = (uint16_t)(Rq | (D[h] << 8));
      s_act[i + h * G] = action;
      if (REPLAY) rq_or |= Rq;
    }
    if (REPLAY) sw_hint_violation(io.err, dl_genv - a.env_offset, rq_or, xbad);
    if (__builtin_expect(rej, 0)) {                                      // a rejected word: that row again, from the retry stream
#pragma unroll 1
      for (int h = 0; h < 4; ++h) {
        uint32_t y2, j2;
        if (!rng_split(w[h], y2, j2)) {
          uint32_t jn; uint32_t yy = rng_group_y(a.seed, dl_genv, tick_a + (uint32_t)h, dl_s, 0, 1, &jn);
          if (!k6) yy -= __umul24((uint32_t)((float)yy * a.inv_pK), a.pK);
          const float action = rp_act ? av[h] : rng_j_to_action(jn);
          const int Rq = rp_act ? (int)fminf(rintf(action), 255.0f) : (int)rintf(action);
          s_rd[i + h * G] = (uint16_t)(Rq | ((rp_exo ? Dx[h] : (int)s_dtab[yy]) << 8));
          s_act[i + h * G] = action;
        }
      }
    }
  };

  // ---- the store waves: staged tile of chunk c -> trajectory rows ---------------------------------------------------------------
  // flat piece index q = row * P + piece over a staged tile, advancing by the store lanes per iteration: (piece, byte offset)
  // are carried incrementally -- no multiply in the loop; each instruction writes 64 consecutive 16-byte pieces (1 KB)
  const int nsl = n_store_waves << 6, sl = tid - store_first;
  auto stream = [&](char* dst, const float* src, int tc, uint32_t P, auto divP, uint32_t rb) __attribute__((always_inline)) {
    const uint32_t n = (uint32_t)tc * P, dq = (uint32_t)nsl;
    const uint32_t dr = divP(dq), dp = dq - dr * P;                                  // nsl = dr * P + dp
    uint32_t q = (uint32_t)sl;
    const uint32_t r0_ = divP(q);
    uint32_t pc = q - r0_ * P, off = r0_ * rb + pc * 16u;
    const uint32_t d_off = dr * rb + dp * 16u, wrap = rb - P * 16u;
    const float4* sp = (const float4*)src + q;
    // SW_DEPTH pieces per trip: the LDS reads are issued together (one round trip), then the stores (6 or 8 per trip: no faster).
    // (Round 5, after the workers' fused phase left the store waves as the busiest role -- 89 % at SC64: every trip a full, predicated
    //  batch instead of full trips followed by single pieces: 4 per trip 114.1 against 112.5 us per 8 x 100-step call, 8 per trip 131 --
    //  the stores' pace is the memory system's, not the LDS round trips'; bursts make it worse.)
        // (ONE piece per trip since late round 5: the stores of a lane then leave evenly over the iteration instead of in bursts of four --
    //  the 8-fragment bench shape 111.5-112.3 against 113.6-118.4 us per call on a slow box, 108.6 against 112.1-113.7 on another, the
    //  bench line +4 %; every other shape within noise.  SW_STORE_DEPTH: development.)
#ifndef SW_STORE_DEPTH
#define SW_STORE_DEPTH 1
#endif
    constexpr int SW_DEPTH = SW_STORE_DEPTH;
    for (; q + (uint32_t)(SW_DEPTH - 1) * dq < n; q += (uint32_t)SW_DEPTH * dq, sp += (uint32_t)SW_DEPTH * dq) {
      float4 v[SW_DEPTH];
#pragma unroll
      for (int k = 0; k < SW_DEPTH; ++k) v[k] = sp[(uint32_t)k * dq];
      uint32_t o[SW_DEPTH];
#pragma unroll
      for (int k = 0; k < SW_DEPTH; ++k) { o[k] = off; pc += dp; off += d_off; if (pc >= P) { pc -= P; off += wrap; } }
#ifndef PHX_ABL_NOSTORE
#pragma unroll
      for (int k = 0; k < SW_DEPTH; ++k) sw_store16(dst + (size_t)o[k], v[k]);
#endif
    }
    for (; q < n; q += dq, sp += dq) {
#ifndef PHX_ABL_NOSTORE
      sw_store16(dst + (size_t)off, sp[0]);
#endif
      pc += dp; off += d_off;
      if (pc >= P) { pc -= P; off += wrap; }
    }
  };
  // Rows of the launch -> rows of the caller's fragments (phx_rollout_io.frags: row t is row t - f frag_T of fragment f = t / frag_T; a
  // single fragment is the io's own planes).  The store waves visit the chunks in order: a cursor per plane group, no division; a chunk
  // that straddles two fragments leaves in two runs.
  const int Tf = a.frag_T;
  int fs_f = 0, fs_lo = 0, fa_f = 0, fa_lo = 0;                          // cursors of stores() / store_actions(): fragment and its first row
  auto runs = [&](int t0, int tc, int& f, int& lo, auto body) __attribute__((always_inline)) {
    for (int r = 0; r < tc;) {
      while (t0 + r >= lo + Tf) { lo += Tf; ++f; }
      const int left = lo + Tf - (t0 + r), n = tc - r < left ? tc - r : left;
      // (the fragment index as an SGPR: the store role is a divergent branch to the compiler, which otherwise reads the fragment's plane
      //  pointers with VECTOR loads from the argument block -- and the s_waitcnt vmcnt(0) behind each drains the wave's trajectory stores,
      //  loads and stores share the counter: three drains per chunk)
      body(__builtin_amdgcn_readfirstlane(f), t0 + r - lo, r, n);        // fragment, first row in it, first row of the tile, rows
      r += n;
    }
  };
  auto store_actions = [&](int c, int t0, int tc) __attribute__((always_inline)) {      // chunk c's actions, staged by the draws one iteration ago
    runs(t0, tc, fa_f, fa_lo, [&](int f, int tr, int r, int n) __attribute__((always_inline)) {
      stream((char*)(a.frag[f].action_out + ((int64_t)tr * total + g_base)), s_act0 + (c & 1) * items + r * G, n, (uint32_t)(G >> 2), div_G4, utotal * 4u);
    });
  };
  auto stores = [&](int c, int t0, int tc) __attribute__((always_inline)) {
    const float* const o_obs = s_out0 + (c & 1) * (4 * items);
    const float* const o_rew = o_obs + 3 * items;
    const uint32_t PO = 3u * (uint32_t)(G >> 2), PR = (uint32_t)(G >> 2);
    runs(t0, tc, fs_f, fs_lo, [&](int f, int tr, int r, int n) __attribute__((always_inline)) {
      const int64_t row0 = (int64_t)tr * total + g_base;
      stream((char*)(a.frag[f].obs + row0 * 3), o_obs + r * (3 * G), n, PO, div_PO, utotal * 12u);
      stream((char*)(a.frag[f].reward + row0), o_rew + r * G, n, PR, div_G4, utotal * 4u);
    });
  };

  // ---- the flag planes, by the store waves while they have nothing to stream (iterations -2 .. 0: the pipeline fills, HBM is idle).
  // truncations["__all__"] per shop (env.py:312-318) is a function of the env's step counter at launch and the row alone: row t ends a
  // pair's episode <=> t == e + j num_steps, e = num_steps - 1 - step (s_ftend, set at setup); terminations are all zero
  // (agents.py:292-323).  So the workgroup's G-byte segments of all T rows are written BEFORE the first chunk is streamed instead of
  // with every chunk: on some boxes the two planes cost the steady state 8-10 us per launch (tools/ubench/ub_store10.hip, FLAGS 0 / 1).
  auto flag_segments = [&](int part) __attribute__((always_inline)) {
#ifndef PHX_ABL_NOSTORE
    const uint32_t uT = (uint32_t)a.T, ns = (uint32_t)a.num_steps, PF = (uint32_t)(G >> 4);
    const uint32_t p_lo = part == 0 ? 0u : (part == 1 ? (uT * 2u) / 5u : (uT * 13u) / 20u), p_hi = part == 0 ? (uT * 2u) / 5u : (part == 1 ? (uT * 13u) / 20u : uT);
    const float inv_ns = 1.0f / (float)ns;
#pragma unroll 1
    for (int fr = 0; fr < a.n_frag; ++fr) {                                // the part's rows, fragment by fragment (one fragment: the whole part)
      const uint32_t f_lo = (uint32_t)fr * (uint32_t)Tf, f_hi = f_lo + (uint32_t)Tf;
      const uint32_t r_lo = p_lo > f_lo ? p_lo : f_lo, r_hi = p_hi < f_hi ? p_hi : f_hi;
      if (r_hi <= r_lo) continue;
      const uint32_t n = (r_hi - r_lo) * PF;
      char* const p_tru = (char*)(a.frag[fr].truncated + g_base);
      char* const p_ter = a.frag[fr].terminated ? (char*)(a.frag[fr].terminated + g_base) : nullptr;
      char* const p_ov = FSM ? (char*)(a.frag[fr].obs_valid + g_base) : nullptr;
      char* const p_rv = FSM ? (char*)(a.frag[fr].reward_valid + g_base) : nullptr;
#pragma unroll 1
      for (uint32_t f = (uint32_t)sl; f < n; f += (uint32_t)nsl) {
        const uint32_t rr = div_PF(f), pc = f - rr * PF, t = r_lo + rr;
        uint32_t x = t - (uint32_t)((float)t * inv_ns) * ns;                // t mod num_steps (t < 2^16: the f32 quotient is off by one at most)
        if ((int)x < 0) x += ns;
        if (x >= ns) x -= ns;
        const uint4* const src = (const uint4*)(s_ftend + 16u * pc);
        const uint4 ea = src[0], eb = src[1];
        const uint32_t xx = x | (x << 16);
        // two packed u16 -> two bytes: 1 where the half equals x
        auto eq2 = [&](uint32_t w2) { const uint32_t z = w2 ^ xx; return ((z & 0xFFFFu) == 0u ? 1u : 0u) | ((z >> 16) == 0u ? 0x100u : 0u); };
        uint4 v = make_uint4(eq2(ea.x) | (eq2(ea.y) << 16), eq2(ea.z) | (eq2(ea.w) << 16), eq2(eb.x) | (eq2(eb.y) << 16), eq2(eb.z) | (eq2(eb.w) << 16));
        if (__builtin_expect(neg_steps, 0)) {                               // counters below zero: the general rule, byte by byte
          uint32_t w[4] = {0u, 0u, 0u, 0u};
          for (int b = 0; b < 16; ++b) { const uint32_t et = s_ftend[16u * pc + b]; if (et != 0xFFFFu && t >= et && (t - et) % ns == 0u) w[b >> 2] |= 1u << (8 * (b & 3)); }
          v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        const size_t off = (size_t)(t - f_lo) * (size_t)utotal + (size_t)(pc * 16u);
        // plain stores: a flag row of the block is G bytes, not whole lines -- except for 128-pair workgroups, whose rows ARE whole 128-byte
        // lines; there non-temporal stores win where there are four planes (FSM, config 3: 847 against 858 us per T = 400) and change
        // nothing where there are two (config 4's share); 144-byte rows lose 3 % with them
        constexpr bool flags_nt = FSM && GT == 128;
        auto put = [&](char* q, const uint4 w) __attribute__((always_inline)) {
          if (flags_nt) sw_store16(q, make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)));
          else *(uint4*)q = w;
        };
        put(p_tru + off, v);
        if (p_ter) put(p_ter + off, make_uint4(0u, 0u, 0u, 0u));
        if (FSM) {
          // obs_valid / reward_valid of row t: functions of the pair's episode position p = (step at launch + t) mod num_steps -- with
          // et = num_steps - 1 - step that is x - 1 - et (+ num_steps) -- and, for the rows of the launch's own episode before its first
          // rewarded position, of the cache's validity at launch:   obs_valid = OBS(p),   reward_valid = OBS(p) ? (cached ? 1 : 2) : 0
          // (fsm.py:360-378).  Envs in step (the usual case) share one position per piece: one lookup, splat.
          const uint32_t et0 = ea.x & 0xFFFFu, sp0 = et0 | (et0 << 16);
          const bool same = ((ea.x ^ sp0) | (ea.y ^ sp0) | (ea.z ^ sp0) | (ea.w ^ sp0) | (eb.x ^ sp0) | (eb.y ^ sp0) | (eb.z ^ sp0) | (eb.w ^ sp0)) == 0u;
          auto pos_of = [&](uint32_t et) { int p = (int)x - 1 - (int)et; if (p < 0) p += (int)ns; return p; };
          const uint32_t fl0 = s_pf[pos_of(et0)];
          uint4 vo, vr;
          if (same && (!(fl0 & SWF_OBS) || (fl0 & SWF_HASREW) || t > et0)) {
            const uint32_t ob = (fl0 & SWF_OBS) ? 0x01010101u : 0u, rb = !(fl0 & SWF_OBS) ? 0u : ((fl0 & SWF_HASREW) ? 0x01010101u : 0x02020202u);
            vo = make_uint4(ob, ob, ob, ob); vr = make_uint4(rb, rb, rb, rb);
          } else {
            uint32_t wo[4] = {0u, 0u, 0u, 0u}, wr[4] = {0u, 0u, 0u, 0u};
            for (int b = 0; b < 16; ++b) {
              const uint32_t et = s_ftend[16u * pc + b], fl = s_pf[pos_of(et)];
              if (fl & SWF_OBS) {
                const bool cached = (fl & SWF_HASREW) || (t <= et && s_cv0[16u * pc + b]);
                wo[b >> 2] |= 1u << (8 * (b & 3)); wr[b >> 2] |= (cached ? 1u : 2u) << (8 * (b & 3));
              }
            }
            vo = make_uint4(wo[0], wo[1], wo[2], wo[3]); vr = make_uint4(wr[0], wr[1], wr[2], wr[3]);
          }
#ifndef SWF_ABL_NOFLAGS
          put(p_ov + off, vo); put(p_rv + off, vr);
#endif
        }
      }
    }
#endif
  };

  // ---- state after the fragment: written by the recurrence lanes as soon as the last chunk's chain is done (iteration n_chunks - 1,
  //      beside the last output phase), not after the drain: the arrival counter's round trip would otherwise end the launch
  auto finish = [&]() __attribute__((always_inline)) {
  if (tid < G) {
    const int64_t g = g_base + tid;
    const int R = FSM ? fin_rd & 127 : fin_rd & 255, D = FSM ? (fin_rd >> 8) & 31 : fin_rd >> 8;
    const int xb = fin_xb;
    const int sales = min(xb, D), missed = D - sales;
    a.stock[g] = x; a.sales[g] = sales; a.missed[g] = missed;
    if (!FSM) {
      a.delivered[g] = min(R, PHX_SHOP_MAX_STOCK - xb);
      if (io.last_obs) {
        float ob[3];
        shop_obs(x, sales, missed, a.norm, ob);
        io.last_obs[g * 3 + 0] = ob[0]; io.last_obs[g * 3 + 1] = ob[1]; io.last_obs[g * 3 + 2] = ob[2];
      }
    } else {
      // what fsm.py's caches and the shop hold after the fragment's last step
      if (f_del >= 0) a.delivered[g] = f_del;                            // the most recent acting step's delivery (never reset)
      if (f_rew >= 0) a.rew_cache[g] = shop_reward((f_rew >> 7) & 31, f_rew & 127);        // the value outlives a reset, its validity does not
      if (f_rws != SWF_RW_LAUNCH) a.rew_cache_v[g] = (f_rws & SWF_RW_CACHED) ? 1 : 0;
      if (f_obs >= 0) {
        float ob[3];
        shop_obs_f32(f_obs & 127, (f_obs >> 7) & 31, (f_obs >> 12) & 31, (float)a.norm, ob);
        a.obs_cache[g * 3] = ob[0]; a.obs_cache[g * 3 + 1] = ob[1]; a.obs_cache[g * 3 + 2] = ob[2];
        a.obs_cache_v[g] = 1;
      }
      if (io.last_obs) {
        // the observation the next fragment starts from: the one the last row emitted, or after an episode end the reset's (a shop
        // that acts in the initial stage observes its reset state: stock 0, the last step's sales, fsm.py:237-251)
        float ob[3] = {0.f, 0.f, 0.f};
        if (fin_term) { if (s_pf[0] & SWF_ACT) shop_obs_f32(0, sales, missed, (float)a.norm, ob); }
        else if (fin_rd & (int)SWF_OBS) shop_obs_f32(x, sales, missed, (float)a.norm, ob);
        io.last_obs[g * 3 + 0] = ob[0]; io.last_obs[g * 3 + 1] = ob[1]; io.last_obs[g * 3 + 2] = ob[2];
      }
    }
    const uint32_t pr = s_pair[tid];
    const int bl = (int)(pr >> 8);
    if ((pr & 255u) == 0u || tid == 0) {
      // another block that holds a part of this env may not have read its step counter and tick yet: the block that
      // FINISHES LAST with the env writes them (per-env arrival counter, as in phx_sc_rollout_fast_kernel)
      const int64_t b = b_first + bl, p0 = b * nS;
      const int n_touch = (int)((p0 + nS - 1) / G - p0 / G) + 1;
      if (n_touch == 1 || atomicAdd(&a.env_arrive[b], 1) + 1 == n_touch) {
        a.env_step[b] = step; a.env_tick[b] = s_tick0[bl] + a.T;
        if (FSM) {                                                       // the stage the next step runs in, and the last one's (fsm.py:355)
          a.env_stage[b] = (int32_t)a.fsm_tab[a.num_steps + step];
          a.env_prev_stage[b] = (int32_t)a.fsm_tab[a.num_steps + (step > 0 ? step - 1 : a.num_steps - 1)];
        }
        if (n_touch > 1) a.env_arrive[b] = 0;
      }
    }
  }
  };

  // ---- REPLAY: the store waves TOUCH the replayed rows of the chunk the workers will draw two iterations from now -- one word per 64-byte
  //      line, consumed (xor-ed into a word nobody reads) an iteration later, when it has long landed -- so that the workers' own loads
  //      find the lines in the L2 instead of paying an HBM round trip under full write pressure once per draw pass
  uint32_t pf_acc = 0, pf_a = 0, pf_x = 0;
  auto touch_inputs = [&](int c) __attribute__((always_inline)) {
    pf_acc ^= pf_a ^ pf_x; pf_a = 0; pf_x = 0;
    if (!REPLAY || c < 0 || c >= n_chunks) return;
    const int t0 = start_of(c), tc = rows_of(c);
    if (rp_act) {
      const int nl = (G >> 4) + 1;                                        // 64-byte lines of a row's G floats (+ 1: the segment need not start on one)
      const int r = sl / nl, l = sl - r * nl;
      if (r < tc) { const int w = l * 16 < G ? l * 16 : G - 1; pf_a = __float_as_uint(io.actions[(int64_t)(t0 + r) * total + g_base + w]); }
    }
    if (rp_exo) {
      const int len = n_env * a.n_exo, nl = (len >> 6) + 1;                // bytes of the group's envs in a row of the exo plane
      const int r = sl / nl, l = sl - r * nl;
      if (r < tc) { const int w = l * 64 < len ? l * 64 : len - 1; pf_x = io.exo[((int64_t)(t0 + r) * a.B + b_first) * a.n_exo + w]; }
    }
  };

  // ---- schedule.  it = -2: the workers draw chunk 0;  it = -1: recurrence(0) beside draws(1);  it >= 0:
  //        workers: outputs(it), draws(it + 2) | recurrence lanes: recurrence(it + 1) | store waves: stores(it - 1), actions(it + 1)    one barrier
  for (int it = -2; it <= n_chunks; ++it) {
    SW_REFRESH();
#ifdef PHX_RT_FILL
#define RSTAMP_WORK() do { if (it >= -2 && it <= 2) RSTAMP(it + 4); } while (0)      /* slots 2..6: own work of iterations -2..2 done (before the barrier) */
    if (it == 3) RSTAMP(7);
#else
#define RSTAMP_WORK() do {} while (0)
    if (it == 1) RSTAMP(2);
    if (it == -1) RSTAMP(6);
    if (it == 0) RSTAMP(7);
    if (it == n_chunks) RSTAMP(3);
#endif
    const int co = it, cr = it + 1, cd = it + 2, cs = it - 1;
    if (tid >= work_first) {
      // workers (letting every other worker wave draw first, so that LDS-heavy and VALU-heavy phases overlap on a SIMD, changes nothing)
      if (fused_ok && co >= 0 && cd < n_chunks && rows_of(co) == TC && rows_of(cd) == TC && !(weird && co == 0)) {
        work_fused(co, cd, start_of(cd));
        STICK(2);
      } else {
      if (co >= 0 && co < n_chunks) outputs(co, rows_of(co));
      STICK(2);
      if (cd < n_chunks) draws(start_of(cd), rows_of(cd), cd);
      STICK(1);
      }
    } else if (tid < rec_threads) {
#ifdef SWF_ABL_PLAINREC
      if (cr >= 0 && cr < n_chunks) recurrence(cr, rows_of(cr));
#else
      if (cr >= 0 && cr < n_chunks) { if (FSM) recurrence_fsm(cr, rows_of(cr)); else recurrence(cr, rows_of(cr)); }
#endif
      else if (it == -2 && pass == 0) replicate_tables(tid, work_first);
      else if (cr == n_chunks) finish();
      STICK(3);
    } else {
      if (it == -2 && pass == 0) replicate_tables(tid, work_first);
#ifndef SW_STORE_PRIO
#define SW_STORE_PRIO 2
#endif
      __builtin_amdgcn_s_setprio(SW_STORE_PRIO);      // the store lanes' single-piece trips are paced by their LDS round trips: issue them ahead of the workers (-1 % on the bench shape)
      if (REPLAY) touch_inputs(it + 3);
      if (it <= 0) flag_segments(it + 2);
      if (cs >= 0) stores(cs, start_of(cs), rows_of(cs));
      if (cr >= 0 && cr < n_chunks) store_actions(cr, start_of(cr), rows_of(cr));     // (drawn in the previous iteration)
      STICK(4);
    }
    RSTAMP_WORK();
    sw_lds_barrier(); STICK(5);
  }
  if (REPLAY && pf_acc == 0x5EEDF00Du && a.T < 0) a.env_arrive[0] = (int32_t)pf_acc;     // (never true: keeps the touches alive)
  }   // pair groups of the workgroup
#ifndef PHX_RT_FILL
  RSTAMP(4);
#endif
#ifdef PHX_TIMING
  if (a.timing && (tid & 63) == 0) for (int q = 0; q < 8; ++q) a.timing[((int64_t)blockIdx.x * 16 + (tid >> 6)) * 8 + q] = tm[q];
#endif
#ifdef PHX_TIMING
#ifdef PHX_RT_FILL
  { const int role = tid == 0 ? 0 : (tid == store_first ? 1 : (tid == work_first ? 2 : -1));
    if (a.rt && role >= 0) for (int q = 0; q < 8; ++q) a.rt[((int64_t)(a.launch_idx & 3) * 8192 + blockIdx.x * 3 + role) * 8 + q] = rts[q]; }
#else
  RSTAMP(5);
  if (a.rt && tid == 0) for (int q = 0; q < 8; ++q) a.rt[((int64_t)(a.launch_idx & 3) * 8192 + blockIdx.x) * 8 + q] = rts[q];
#endif
#endif
}

#undef a
#undef io
#undef SW_REFRESH

// ---- host: tables, plan, launcher -----------------------------------------------------------------------------------
// The kernel's tables -- base values of the observation and reward tables, digit sums, order sums -- built ONCE per env on the host with
// the same f32 / f64 operations (phx_create uploads the 17.9 KB image): computing them in every workgroup cost ~5 k cycles of setup.
void phx_sc_sw_tables(int K, int norm, std::vector<uint8_t>* out) {
  out->assign(SW_TABLE_BYTES, 0);
  float* base = (float*)out->data();
  uint8_t* ds = out->data() + SW_IMG_BASE_BYTES; uint8_t* dtab = ds + 128;
  for (int v = 0; v <= PHX_SHOP_MAX_STOCK; ++v) base[SW_IMG_TABS + v] = (float)v / (float)PHX_SHOP_MAX_STOCK;    // IEEE f32 division (shop_obs_f32)
  for (int v = 0; v < 32; ++v) base[SW_IMG_TABN + v] = (float)v / (float)norm;
  for (int i = 0; i < 401; ++i) {
    const int n = i - 100, sl = n >= 0 ? (n + 9) / 10 : 0, st = 10 * sl - n;                    // a (sales, stock) pair with 10 * sales - stock == n
    volatile double pen = 0.1 * (double)st;                                                      // product and difference rounded separately (shop_reward)
    base[SW_IMG_RTAB + i] = (float)((double)sl - pen);
  }
  for (int k = 0; k < 125; ++k) ds[k] = (uint8_t)(k % 5 + (k / 5) % 5 + k / 25);
  int n5 = 1; for (int k = 0; k < K; ++k) n5 *= 5;
  for (int e = 0; e < n5; ++e) { int v = e, sum = 0; while (v) { sum += v % 5; v /= 5; } dtab[e] = (uint8_t)sum; }
}

static uint32_t sw_magic32(int d) { return (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)(d > 0 ? d : 1)); }
static const size_t SW_LDS_MAX = 160 * 1024;

// Decides whether an env shape takes the store-wave kernel and with which workgroup shape.  `block`: phx_spec.variant_block
// (0 = auto; > 0: pairs per workgroup, taken when it is a multiple of 16 that divides B * S).
// `fsm_ns` > 0: the FSM instantiation's plan (its LDS sections sized for num_steps = fsm_ns positions).
bool phx_sc_sw_plan(int B, int S, int K_uniform, bool norm_uniform, int num_steps, int block, ScSwPlan* p, int fsm_ns) {
  memset(p, 0, sizeof *p);
  if (K_uniform < 1 || K_uniform > 6 || !norm_uniform || S < 1 || S > 255 || block < 0 || num_steps >= 0xFFFF) return false;
  const int64_t total = (int64_t)B * S;
  if (total >= ((int64_t)1 << 24)) return false;                                  // 24-bit multiplies on (row, pair) offsets
  int dtab_n = 1; for (int k = 0; k < K_uniform; ++k) dtab_n *= 5;
  const int tc_env = phx_knobs().sw_tc;            // development default
  const int ns_env = phx_knobs().sw_store_waves;
  const int nw_env = phx_knobs().sw_work_waves;
  auto epb_of = [&](int G) { return (G + S - 2) / S + 1; };                        // the most envs a block can touch
  auto pairs_ok = [&](int G) { return G >= 16 && G <= 256 && G % 16 == 0 && total % G == 0 && epb_of(G) <= 255; };
  auto tc_ok = [&](int G, int tc) {
    return tc <= num_steps && (int64_t)tc * total * 12 < ((int64_t)1 << 32) && sw_lds_bytes(G, epb_of(G), tc, dtab_n, fsm_ns) <= SW_LDS_MAX;
  };
  auto tc_for = [&](int G) {
    if (tc_env == 16 || tc_env == 20) return tc_ok(G, tc_env) ? tc_env : 0;
    if ((G == 144 || G == 128 || G == 96 || G == 48) && tc_ok(G, 16)) return 16;              // the shapes with a compile-time instantiation (16-row chunks)
    return tc_ok(G, 20) ? 20 : (tc_ok(G, 16) ? 16 : 0);
  };
  int G = 0;
  if (block > 0) { if (pairs_ok(block) && tc_for(block)) G = block; }
  else {
    // Wide workgroups write long row segments (tools/ubench/ub_store10.hip: 144 pairs 0.82 of 8 TB/s, 72 pairs 0.76, 48 pairs
    // 0.74); a grid that fills whole rounds of the 256 CUs comes first.
    double best = 0.0;
    for (int cand = 192; cand >= 16; cand -= 16) {
      if (!pairs_ok(cand) || !tc_for(cand)) continue;
      const int64_t nblk = total / cand;
      const size_t lds = sw_lds_bytes(cand, epb_of(cand), tc_for(cand), dtab_n, fsm_ns);
      const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(SW_LDS_MAX / lds, 2));
      const int64_t slots = 256 * (int64_t)per_cu, rounds = (nblk + slots - 1) / slots;
      const double eff = (double)nblk / (double)(rounds * slots);
      const double shape = cand >= 128 ? 1.0 : cand >= 96 ? 0.95 : cand >= 64 ? 0.92 : cand >= 48 ? 0.9 : cand >= 32 ? 0.75 : 0.6;
      if (eff * shape > best + 1e-9) { best = eff * shape; G = cand; }
    }
  }
  if (!G) return false;
  p->G = G; p->epb = epb_of(G); p->K = K_uniform; p->tc = tc_for(G); p->dtab_n = dtab_n;
  p->n_rec = (G + 63) / 64;
  p->n_store = ns_env > 0 ? ns_env : (G >= 96 ? 4 : (G >= 48 ? 2 : 1));
  int work = nw_env > 0 ? nw_env : (p->tc * (G / 4) + 63) / 64;
  if (work < 1) work = 1;
  if (p->n_rec + p->n_store + work > 16) work = 16 - p->n_rec - p->n_store;
  p->nt = 64 * (p->n_rec + p->n_store + work);
  p->lds = (int32_t)sw_lds_bytes(G, p->epb, p->tc, dtab_n, fsm_ns);
  // the shapes with a compile-time instantiation (any number of rounds of workgroups: a round of 144-pair workgroups costs the same
  // whether it is the launch's only one or one of four -- B = 8 192 / 16 384: 113 / 221 us per T = 400 against 137 / 242)
  p->specialised = (p->tc == 16 && ((G == 144 && p->n_rec == 3 && work == 9 && (p->n_store == 4 || p->n_store == 2)) ||
                                    (G == 128 && p->n_rec == 2 && work == 8 && p->n_store == 4) ||
                                    (G == 96 && p->n_rec == 2 && work == 6 && (p->n_store == 4 || p->n_store == 2)) ||
                                    (G == 48 && p->n_rec == 1 && work == 3 && p->n_store == 2))) ? 1 : 0;
  p->ok = 1;
  return true;
}

// Do any of the n replayed actions round below zero (or is one NaN)?  Then the stock can leave [0, 100] (StockRequest of a negative
// size, supply_chain.py:98-103,139) and the call goes to round 1's kernel, whose tiles hold 32-bit words: *flag = gen.
__global__ __launch_bounds__(256) void phx_sw_scan_actions_kernel(const float* __restrict__ act, int64_t n, int32_t* flag, int32_t gen) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) bad |= !(act[i] >= -0.5f);
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicExch(flag, gen);
}

// FSM instantiation: is every env on the tabulated stage chain (its stage the one the table holds for its step counter, the counter inside
// the episode), every stock inside [0, 100], and no reward cache invalid although the episode is past a rewarded position (a caller who
// moved the step counter: the closed form of reward_valid assumes the cache of an env that walked there)?  Otherwise *flag = gen: the store-wave launch returns at entry and the lane-per-pair loop
// (phx_sc_fused.hip), launched behind it with the same word, serves the call.
__global__ __launch_bounds__(256) void phx_sw_fsm_check_kernel(const int32_t* __restrict__ env_step, const int32_t* __restrict__ env_stage, const int32_t* __restrict__ stock,
                                                               const uint8_t* __restrict__ rew_cache_v, const uint16_t* __restrict__ tab, int num_steps, int S,
                                                               int64_t total, int32_t* flag, int32_t gen) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / S;
    const int stp = env_step[b];
    const bool in_ep = stp >= 0 && stp < num_steps;
    bad |= (unsigned)stock[i] > (unsigned)PHX_SHOP_MAX_STOCK || !in_ep || env_stage[b] != (int)tab[num_steps + (in_ep ? stp : 0)] ||
           (in_ep && stp > 0 && (tab[stp - 1] & SWF_HASREW) && !rew_cache_v[i]);
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicExch(flag, gen);
}

hipError_t phx_launch_sc_rollout_sw(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st, int32_t guard_gen) {
  const bool fsm = sp.env_type == PHX_ENV_FSM;
  const ScSwPlan& p = fsm ? sp.fsm_sw : sp.sc_sw;
  const bool replay = io.actions != nullptr || io.exo != nullptr;
  SwArgs a; memset(&a, 0, sizeof a);
  a.B = sp.B; a.S = sp.S; a.epb = p.epb; a.G = p.G; a.K = p.K; a.T = io.T; a.num_steps = sp.num_steps;
  const int remap_env = phx_knobs().rollout_remap;
  a.xcd_remap = remap_env >= 0 ? remap_env : 1;
  a.n_rec_waves = p.n_rec; a.n_store_waves = p.n_store; a.dtab_n = p.dtab_n;
  static const float inv[7] = {1.0f, 0.2f, 0.04f, 0.008f, 0.0016f, 0.00032f, 0.000064f};
  a.pK = (uint32_t)p.dtab_n; a.inv_pK = inv[p.K];
  a.mG = sw_magic32(p.G); a.mG4 = sw_magic32(p.G / 4); a.mS = sw_magic32(sp.S); a.mPO = sw_magic32(3 * (p.G / 4)); a.mPF = p.G / 16 > 1 ? sw_magic32(p.G / 16) : 0;     // (the magic of 1 does not fit 32 bits: 0 = no division)
  a.norm = p.norm; a.seed = sp.seed; a.env_offset = sp.env_offset;
  a.first_rows = io.T <= p.tc ? io.T : p.tc;
  a.stock = (int32_t*)sp.f[F_SHOP_STOCK]; a.sales = (int32_t*)sp.f[F_SHOP_SALES];
  a.missed = (int32_t*)sp.f[F_SHOP_MISSED]; a.delivered = (int32_t*)sp.f[F_SHOP_DELIVERED];
  a.env_step = (int32_t*)sp.f[F_ENV_STEP]; a.env_tick = (int32_t*)sp.f[F_ENV_TICK]; a.env_arrive = (int32_t*)sp.f[F_ENV_ARRIVE];
  a.tables = (const float4*)sp.sc_sw_tables;
  a.io = io;
  if (fsm) {
    a.fsm_tab = sp.fsm_sw_tab;
    a.env_stage = (int32_t*)sp.f[F_ENV_STAGE]; a.env_prev_stage = (int32_t*)sp.f[F_ENV_PREV_STAGE];
    a.rew_cache = (double*)sp.f[F_ENV_REW_CACHE]; a.rew_cache_v = (uint8_t*)sp.f[F_ENV_REW_CACHE_VALID];
    a.obs_cache = (float*)sp.f[F_ENV_OBS_CACHE]; a.obs_cache_v = (uint8_t*)sp.f[F_ENV_OBS_CACHE_VALID];
  }
  a.n_exo = sp.n_exo; a.exo_first = sp.sc_sw_exo_first; a.guard = nullptr; a.guard_gen = guard_gen;
  if (io.actions && guard_gen != 0) {     // the pre-scan of this call's actions decides between this kernel and round 1's (same stream: ordered);
                                          // guard_gen == 0: the caller vouched for the actions (PHX_RH_ACTIONS_IN_DOMAIN)
    const int64_t n = (int64_t)io.T * sp.B * sp.S;
    a.guard = sp.sc_sw_guard;
    hipLaunchKernelGGL(phx_sw_scan_actions_kernel, dim3((unsigned)std::min<int64_t>((n + 1023) / 1024, 2048)), dim3(256), 0, st, io.actions, n, sp.sc_sw_guard, guard_gen);
  }
  if (fsm) {                              // (same stream: the check is ordered before the launch it guards)
    const int64_t total = (int64_t)sp.B * sp.S;
    a.guard = sp.fsm_irregular;
    hipLaunchKernelGGL(phx_sw_fsm_check_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, st, a.env_step, a.env_stage, a.stock,
                       a.rew_cache_v, sp.fsm_sw_tab, sp.num_steps, sp.S, total, sp.fsm_irregular, guard_gen);
  }
  // trajectory fragments: the caller's list (phx_rollout_io.frags, validated by phx_rollout) or the io's own planes as the only one
  if (io.n_frag > 1) {
    a.n_frag = io.n_frag; a.frag_T = io.T / io.n_frag;
    for (int f = 0; f < io.n_frag; ++f) a.frag[f] = SwFrag{io.frags[f].obs, io.frags[f].action_out, io.frags[f].reward, io.frags[f].terminated, io.frags[f].truncated,
                                                           io.frags[f].obs_valid, io.frags[f].reward_valid};
  } else {
    a.n_frag = 1; a.frag_T = io.T;
    a.frag[0] = SwFrag{io.obs, io.action_out, io.reward, io.terminated, io.truncated, io.obs_valid, io.reward_valid};
  }
  // the grid: one workgroup per pair group up to what the chip holds at once, the rest of the groups are walked by the same workgroups
  a.n_groups = (int32_t)(((int64_t)sp.B * sp.S) / p.G);
  unsigned n_wg = (unsigned)a.n_groups;
  if (phx_knobs().sw_persist) {
    const int n_cu = phx_device_cu_count();
    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(SW_LDS_MAX / (size_t)p.lds, (size_t)(2048 / p.nt)));
    unsigned resident = (unsigned)n_cu * per_cu;
    if (n_wg > resident) n_wg = resident >= 8 ? resident & ~7u : resident;        // (a multiple of 8: virtual workgroup vb stays on the XCD of vb % 8)
  }
  const dim3 grid(n_wg);
#ifdef PHX_TIMING
  { static unsigned long long* tbuf = nullptr; if (!tbuf) { (void)hipMalloc((void**)&tbuf, 8 * 16 * 8192 * sizeof(unsigned long long)); (void)hipMemset(tbuf, 0, 8 * 16 * 8192 * sizeof(unsigned long long)); } a.timing = grid.x <= 8192 ? tbuf : nullptr;
    { static unsigned long long* rbuf = nullptr; static int li = 0; if (!rbuf) (void)hipMalloc((void**)&rbuf, (4 * 8192 * 8) * sizeof(unsigned long long)); a.rt = grid.x <= 8192 ? rbuf : nullptr; a.launch_idx = li++;
      if (getenv("PHX_TIMING_DUMP") && li == 44) { (void)hipDeviceSynchronize(); std::vector<unsigned long long> h(4 * 8192 * 8); (void)hipMemcpy(h.data(), rbuf, h.size() * 8, hipMemcpyDeviceToHost);
#ifdef PHX_RT_FILL
        { const char* rn[3] = {"rec", "store", "work"}; const char* nm2[8] = {"entry", "setup done", "it -2 work done", "it -1 work done", "it 0 work done", "it 1 work done", "it 2 work done", "it 3 starts"};
          for (int r = 0; r < 3; ++r) for (int q = 1; q < 8; ++q) { double sum = 0, mx = 0; for (unsigned b = 0; b < grid.x; ++b) { const unsigned long long* e = &h[((size_t)1 * 8192 + b * 3 + r) * 8]; const double v = (double)(long long)(e[q] - e[0]) * 0.01; sum += v; mx = std::max(mx, v); }
            fprintf(stderr, "SW_RTF launch 41 %-5s %-18s mean %7.2f max %7.2f us after the workgroup's entry\n", rn[r], nm2[q], sum / grid.x, mx); } }
#endif
        // launches 40..42 (slots 0..2): stamps in 10 ns ticks relative to the earliest entry of launch 40
        unsigned long long base = ~0ull; for (unsigned b = 0; b < grid.x; ++b) base = std::min(base, h[((size_t)0 * 8192 + b) * 8]);
        const char* nm[8] = {"entry", "setup done", "it 1 (first stores)", "it n_chunks (drain)", "loop done", "end", "it -1", "it 0"};
        for (int l = 0; l < 3; ++l) for (int q : {0, 1, 6, 7, 2, 3, 4, 5}) { double mn = 1e30, mx = -1e30, sum = 0; for (unsigned b = 0; b < grid.x; ++b) { const double v = (double)(long long)(h[((size_t)l * 8192 + b) * 8 + q] - base) * 0.01; mn = std::min(mn, v); mx = std::max(mx, v); sum += v; }
          fprintf(stderr, "SW_RT launch %d  %-22s min %8.2f  mean %8.2f  max %8.2f us\n", 40 + l, nm[q], mn, sum / grid.x, mx); }
        // launch 41 by XCD (workgroup b runs on XCD b % 8): entry and end, relative to the launch's earliest entry
        { unsigned long long b1 = ~0ull; for (unsigned b = 0; b < grid.x; ++b) b1 = std::min(b1, h[((size_t)1 * 8192 + b) * 8]);
          for (int xc = 0; xc < 8; ++xc) { double se = 0, sx = 0, mxx = 0, mnx = 1e30; int n = 0; for (unsigned b = xc; b < grid.x; b += 8) { const unsigned long long* e = &h[((size_t)1 * 8192 + b) * 8]; const double en = (double)(long long)(e[0] - b1) * 0.01, ex = (double)(long long)(e[5] - b1) * 0.01; se += en; sx += ex; mxx = std::max(mxx, ex); mnx = std::min(mnx, ex); ++n; }
            fprintf(stderr, "SW_RT launch 41 XCD %d: entry mean %6.2f | end min %6.2f mean %6.2f max %6.2f us\n", xc, se / n, mnx, sx / n, mxx); } } } }
    if (getenv("PHX_TIMING_DUMP")) { static int calls = 0; if (++calls == 20 && a.timing) { (void)hipDeviceSynchronize(); std::vector<unsigned long long> h(8 * 16 * 8192); (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
      const char* role[3] = {"rec  ", "store", "work "}; const int nwv = p.nt / 64;
      for (int r = 0; r < 3; ++r) { double sum[8] = {0}; int n = 0;
        for (unsigned b = 0; b < grid.x; ++b) for (int w = 0; w < nwv; ++w) { const int rr = w < p.n_rec ? 0 : (w < p.n_rec + p.n_store ? 1 : 2); if (rr != r) continue; ++n; for (int q = 0; q < 8; ++q) sum[q] += (double)h[((size_t)b * 16 + w) * 8 + q]; }
        fprintf(stderr, "SW_TIMING %s waves (%d): setup %.0f (before 1st barrier %.0f, in it %.0f) | draws %.0f | outputs %.0f | rec %.0f | stores %.0f | barrier %.0f   cycles per wave and launch\n", role[r], n, sum[0]/n, sum[6]/n, sum[7]/n, sum[1]/n, sum[2]/n, sum[3]/n, sum[4]/n, sum[5]/n); } } } }
#endif
  phx_note_kernel(fsm ? "phx_sc_rollout_sw_kernel[fsm]" : (replay ? "phx_sc_rollout_sw_kernel[replay]" : "phx_sc_rollout_sw_kernel"));
  // more than 64 KB of dynamic LDS needs the attribute (once per instantiation and device)
#define SW_LAUNCH_(TC_, GT_, NREC_, NSTORE_, NWORK_, RP_) do { \
    static PhxPerDeviceOnce attr_done; int dev = 0; (void)hipGetDevice(&dev); \
    if (!attr_done.done(dev)) { hipError_t e = hipFuncSetAttribute((const void*)phx_sc_rollout_sw_kernel<TC_, GT_, NREC_, NSTORE_, NWORK_, RP_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SW_LDS_MAX); if (e != hipSuccess) return e; attr_done.mark(dev); } \
    hipLaunchKernelGGL((phx_sc_rollout_sw_kernel<TC_, GT_, NREC_, NSTORE_, NWORK_, RP_>), grid, dim3(p.nt), (size_t)p.lds, st, a); } while (0)
#define SW_LAUNCH_PLAIN(TC_, GT_, NREC_, NSTORE_, NWORK_) do { if (replay) SW_LAUNCH_(TC_, GT_, NREC_, NSTORE_, NWORK_, 1); else SW_LAUNCH_(TC_, GT_, NREC_, NSTORE_, NWORK_, 0); } while (0)
#define SW_LAUNCH(TC_, GT_, NREC_, NSTORE_, NWORK_) do { if (fsm) SW_LAUNCH_(TC_, GT_, NREC_, NSTORE_, NWORK_, 2); else SW_LAUNCH_PLAIN(TC_, GT_, NREC_, NSTORE_, NWORK_); } while (0)
  const int generic_env = phx_knobs().sw_generic;      // development: the run-time-shape instantiation
  const int work = p.nt / 64 - p.n_rec - p.n_store;
#define SW_SHAPE(G_, NREC_, NSTORE_, NWORK_) (p.tc == 16 && p.G == G_ && p.n_rec == NREC_ && p.n_store == NSTORE_ && work == NWORK_)
  // (144-pair workgroups: no FSM instantiation -- its sections do not fit beside 16-row tiles of 144 pairs)
  if (!generic_env && !fsm && SW_SHAPE(144, 3, 4, 9)) SW_LAUNCH_PLAIN(16, 144, 3, 4, 9);
  else if (!generic_env && !fsm && SW_SHAPE(144, 3, 2, 9)) SW_LAUNCH_PLAIN(16, 144, 3, 2, 9);
  else if (!generic_env && SW_SHAPE(128, 2, 4, 8)) SW_LAUNCH(16, 128, 2, 4, 8);
  else if (!generic_env && SW_SHAPE(96, 2, 4, 6)) SW_LAUNCH(16, 96, 2, 4, 6);
  else if (!generic_env && SW_SHAPE(96, 2, 2, 6)) SW_LAUNCH(16, 96, 2, 2, 6);
  else if (!generic_env && SW_SHAPE(48, 1, 2, 3)) SW_LAUNCH(16, 48, 1, 2, 3);
  else if (p.tc == 20) SW_LAUNCH(20, 0, 0, 0, 0);
  else SW_LAUNCH(16, 0, 0, 0, 0);
#undef SW_SHAPE
#undef SW_LAUNCH
#undef SW_LAUNCH_PLAIN
#undef SW_LAUNCH_
  return hipGetLastError();
}

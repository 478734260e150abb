// phx_sc_fused.hip -- fused static-schedule kernels for supply-chain-shaped envs.
//
// For a topology made only of FactoryAgent / ShopAgent / CustomerAgent (supply_chain.py:36-150)
// the message schedule of a step is static: acting phase = one StockRequest per shop and one
// OrderRequest per customer; round 0 = the factory echoes StockResponses while every shop
// fills its OrderRequests sequentially against the PRE-delivery stock; round 1 = the shop
// books the delivery and customers drop their OrderResponses; round 2 is empty.  The whole
// step is therefore a closed form per shop:
//     D      = sum of the acting customers' order sizes           (the shop's round-0 inbox)
//     sales  = min(D, stock0)  (also for stock0 < 0: the first order takes the negative stock)
//     missed = D - sales ;  stock1 = stock0 - sales
//     stock2 = min(stock1 + request, 100)
// so one lane owns one (env, shop): state is read and written exactly once per step with
// fully coalesced [B][S] accesses, and the only gather -- the K order-size bytes of the
// shop's inbox -- is staged through LDS from 16-byte coalesced loads of the exo rows.
// Results are bit-identical to the generic engine (tests/test_gpu_parity.py).
#include "phx_dev.h"
#include "phx_sc_fast.h"

#ifndef PHX_STEP_REMAP
#define PHX_STEP_REMAP 1     // XCD-aware env mapping: SC256-FSM B=8192 step 12.0 -> 10.9 us, SC64 B=65536 13.3 -> 12.7 us, neutral at B=4096 (graph replay)
#endif
#include <cstdlib>
#include <cstdio>
#include <vector>

#define SC_NT 256
#define SC_STAGE_MAX 32768      // bytes of exo rows staged per block; larger -> direct loads

struct ShopLane {
  int stock, sales, missed, delivered;
};

// one shop, one step; returns observation/reward of the post-step state
__device__ __forceinline__ void sc_shop_step(ShopLane& st, bool has_action, float action, bool any_order,
                                             int D) {
  const int stock0 = st.stock;
  int req = 0;
  if (has_action) {                                              // decode_action supply_chain.py:136-142
    const int r = dev_round_half_even(action);
    const int room = PHX_SHOP_MAX_STOCK - stock0;
    req = r < room ? r : room;
  }
  st.sales = 0; st.missed = 0;                                   // pre_message_resolution :93-96
  int stock1 = stock0;
  if (any_order) {                                               // round 0: handle_order_request :105-122
    const int sales = D < stock0 ? D : stock0;
    st.sales = sales; st.missed = D - sales; stock1 = stock0 - sales;
  }
  if (has_action) {                                              // round 1: handle_stock_response :98-103
    st.delivered = req;
    const int ns = stock1 + req;
    stock1 = ns < PHX_SHOP_MAX_STOCK ? ns : PHX_SHOP_MAX_STOCK;
  }
  st.stock = stock1;
}

template <int NT>
__global__ __launch_bounds__(NT) void phx_sc_step_kernel(const DevSpec sp, const phx_step_io io,
                                                            const int epb, const int stage_exo) {
  // A block owns `epb` WHOLE envs (epb * S <= 256 lanes): the per-env words (step, tick, stage)
  // are read by every lane of the env and rewritten by its shop-0 lane, so readers and writer
  // must sit in one workgroup with a barrier between the two.
  extern __shared__ __attribute__((aligned(16))) unsigned char s_exo[];
  const int nS = sp.S;
  const int64_t b_first = (int64_t)xcd_block(stage_exo >= 0 && PHX_STEP_REMAP) * epb;
  const int64_t b_end = (b_first + epb < sp.B) ? b_first + epb : sp.B;
  const int lanes = (int)(b_end - b_first) * nS;
  const bool active = (int)threadIdx.x < lanes;
  const int64_t g = b_first * nS + threadIdx.x;

  // ---- stage the block's exo rows (the shops' round-0 inboxes) into LDS ----------------------
  int64_t lds_base = 0;
  const bool staged = io.exo != nullptr && stage_exo;
  if (staged) {
    const int64_t lo = b_first * sp.n_exo, hi = b_end * sp.n_exo;
    lds_base = lo & ~(int64_t)15;
    const int64_t hi16 = hi & ~(int64_t)15;
    for (int64_t off = lds_base + (int64_t)threadIdx.x * 16; off < hi16; off += (int64_t)NT * 16)
      *(uint4*)(s_exo + (off - lds_base)) = *(const uint4*)(io.exo + off);
    for (int64_t off = (hi16 > lds_base ? hi16 : lds_base) + threadIdx.x; off < hi; off += NT)
      s_exo[off - lds_base] = io.exo[off];
  }
  const int b = active ? (int)(b_first + threadIdx.x / nS) : (int)b_first;
  const int s = active ? (int)(threadIdx.x % nS) : 0;
  // every load of the step is issued here, BEFORE the barrier (a launch this small lasts as long as its chain of
  // dependent global round trips: the state / action loads do not depend on the env words)
  const int a_shop = sp.shop_agent[s];
  const int cur_stage0 = (sp.env_type == PHX_ENV_FSM) ? fld<int32_t>(sp, F_ENV_STAGE)[b] : 0;
  const int step0 = fld<int32_t>(sp, F_ENV_STEP)[b];
  const uint32_t tick0 = (uint32_t)fld<int32_t>(sp, F_ENV_TICK)[b];
  ShopLane st;
  st.stock = active ? fld<int32_t>(sp, F_SHOP_STOCK)[g] : 0;
  st.delivered = active ? fld<int32_t>(sp, F_SHOP_DELIVERED)[g] : 0;
  const bool av_in = active && (io.action_valid == nullptr || io.action_valid[g] != 0);
  const float action_in = (active && io.actions) ? io.actions[g] : 0.0f;
  const int c_lo = sp.shop_cust_ptr[s], c_hi = sp.shop_cust_ptr[s + 1];
  // who acts / observes / is rewarded in a stage: one flag byte per (list, shop); a plain env has one list
  int fl = sp.env_type == PHX_ENV_FSM ? 0 : sp.sc_shop_flags[s];
  __syncthreads();          // exo rows staged; every lane has read the per-env words
  if (!active) return;

  const int cur_stage = cur_stage0;
  const int list = cur_stage;
  const int t = step0 + 1;                                                   // env.py:252
  const uint32_t tick = tick0;

  if (sp.env_type == PHX_ENV_FSM) fl = sp.sc_shop_flags[(int64_t)list * nS + s];
  // the stage after this step: next_stages[0], or what a tabulated clock / stage handler returns (fsm.py:281-302); the
  // agents acting in THAT stage observe (fsm.py:320) unless rewarded_agents is None (every strategic agent, :315-317)
  int next_stage = 0;
  if (sp.env_type == PHX_ENV_FSM) {
    next_stage = sp.stage_next[cur_stage];
    if (sp.stage_tab) {
      next_stage = sp.stage_tab[(int64_t)cur_stage * (sp.num_steps + 1) + (t <= sp.num_steps ? t : sp.num_steps)];
      const bool obs_next = sp.stage_rew_all[cur_stage] || (sp.sc_shop_flags[(int64_t)next_stage * nS + s] & 1);
      fl = (fl & ~8) | (obs_next ? 8 : 0);
    }
  }
  const bool shop_acts = (fl & 1) != 0;
  const bool has_action = shop_acts && av_in;
  const float action = has_action ? action_in : 0.0f;

  // D = sum over the shop's inbox of OrderRequest sizes (customers that act in this stage)
  const uint8_t* cact = sp.shop_cust_act + (int64_t)list * sp.n_exo;
  int D = 0;
  const bool any_order = (fl & 2) != 0, all_order = (fl & 4) != 0 || c_hi == c_lo;
  if (io.exo) {
    const int64_t row = (int64_t)b * sp.n_exo;
    if (any_order)
      for (int k = c_lo; k < c_hi; ++k)
        if (all_order || cact[k]) {
          const int64_t idx = row + sp.shop_cust_exo[k];
          D += staged ? s_exo[idx - lds_base] : io.exo[idx];
        }
  } else {
    if (all_order) D = rng_shop_order_sum(sp.seed, sp.env_offset + b, tick, s, c_hi - c_lo, nullptr);
    else if (any_order)
      D = rng_shop_orders(sp.seed, sp.env_offset + b, tick, s, c_hi - c_lo, cact + c_lo, -1);
  }
  sc_shop_step(st, has_action, action, any_order, D);

  fld<int32_t>(sp, F_SHOP_STOCK)[g] = st.stock;
  fld<int32_t>(sp, F_SHOP_SALES)[g] = st.sales;
  fld<int32_t>(sp, F_SHOP_MISSED)[g] = st.missed;
  if (has_action) fld<int32_t>(sp, F_SHOP_DELIVERED)[g] = st.delivered;

  // ---- obs / reward / done with the PLAIN or FSM masks (env.py:273-292, fsm.py:309-380) -------
  // ShopAgent never terminates or truncates (agents.py:292-323), so __all__ needs no reduction:
  const bool all_trunc = (t == sp.num_steps);                                // env.py:312-318
  // tutorial 2's typed shop (docs/user/tutorial2.rst:244-307): weighted penalty + 4th observation
  const int OD = sp.D;
  const bool typed = sp.any_typed && sp.shop_type_src[s] != PHX_TYPE_NONE;
  const double tw = typed ? shop_type_value(sp, b, s) : 0.0;
  const float tobs = typed ? (float)(tw / sp.shop_type_prm[2 * s + 1]) : 0.f;
  float ob[4] = {0.f, 0.f, 0.f, 0.f};
  uint8_t ov = 0, rv = 0;
  double rw = 0.0;
  if (sp.env_type == PHX_ENV_PLAIN) {
    shop_obs_f32(st.stock, st.sales, st.missed, (float)sp.param_i[a_shop * PHX_NPI + 1], ob);
    ob[3] = tobs;
    rw = typed ? shop_reward_w(st.sales, st.stock, tw) : shop_reward(st.sales, st.stock);
    ov = 1; rv = 1;
  } else {
    double* rc = fld<double>(sp, F_ENV_REW_CACHE) + g;
    uint8_t* rcv = fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID) + g;
    float* oc = fld<float>(sp, F_ENV_OBS_CACHE) + g * OD;
    uint8_t* ocv = fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID) + g;
    const bool observes = (fl & 8) != 0;
    if (observes) {
      shop_obs_f32(st.stock, st.sales, st.missed, (float)sp.param_i[a_shop * PHX_NPI + 1], ob);
      ob[3] = tobs;
      oc[0] = ob[0]; oc[1] = ob[1]; oc[2] = ob[2]; *ocv = 1;                  // fsm.py:349
      if (OD == 4) oc[3] = ob[3];
    }
    uint8_t cache_valid = *rcv; double cache = *rc;
    if (fl & 16) {                                                            // fsm.py:334-335,350
      cache = typed ? shop_reward_w(st.sales, st.stock, tw) : shop_reward(st.sales, st.stock); cache_valid = 1;
      *rc = cache; *rcv = 1;
    }
    if (all_trunc) {                                                          // fsm.py:360-375
      ov = *ocv;
      ob[0] = ov ? oc[0] : 0.f; ob[1] = ov ? oc[1] : 0.f; ob[2] = ov ? oc[2] : 0.f;
      ob[3] = (ov && OD == 4) ? oc[3] : 0.f;
      rv = cache_valid ? 1 : 2; rw = cache_valid ? cache : 0.0;
    } else if (observes) {                                                    // fsm.py:378
      ov = 1; rv = cache_valid ? 1 : 2; rw = cache_valid ? cache : 0.0;
    } else { ob[0] = ob[1] = ob[2] = ob[3] = 0.f; }
  }
  if (OD == 4) *(float4*)(io.obs + g * 4) = make_float4(ob[0], ob[1], ob[2], ob[3]);
  else { io.obs[g * 3 + 0] = ob[0]; io.obs[g * 3 + 1] = ob[1]; io.obs[g * 3 + 2] = ob[2]; }
  io.reward[g] = rw;
  io.obs_valid[g] = ov; io.reward_valid[g] = rv; io.done_valid[g] = 1;
  io.terminated[g] = 0; io.truncated[g] = 0;
  if (s == 0) {
    fld<int32_t>(sp, F_ENV_STEP)[b] = t;
    fld<int32_t>(sp, F_ENV_TICK)[b] = (int32_t)(tick + 1);
    if (sp.env_type == PHX_ENV_FSM) {                                         // fsm.py:355
      fld<int32_t>(sp, F_ENV_PREV_STAGE)[b] = cur_stage;
      fld<int32_t>(sp, F_ENV_STAGE)[b] = next_stage;
    }
    io.all_terminated[b] = 0; io.all_truncated[b] = all_trunc;
  }
}

// ---- the same step for LARGE batches of plain supply-chain envs: four consecutive (env, shop) pairs per thread ---------------------
// phx_sc_step_kernel is one lane per pair, 4-byte loads and stores and five byte planes: at 2^18 envs of SC64 (2.4 M pairs, 103 MB per
// step) it saturates at 0.36 of the HBM peak (this kernel: 0.52) -- 65 % of its wave cycles wait, 4.6 rounds of waves whose lives are two dependent
// memory round trips.  Here a thread owns pairs 4u .. 4u + 3 of a block of whole envs (epb S pairs, a multiple of 4): state, actions
// and masks arrive as 16-byte loads (all issued before the barrier that separates the env words' readers from their writers), four
// Philox blocks are in flight per thread, and observation (48 B), reward (32 B), state (4 x 16 B) and the five flag planes (4 B each)
// leave as whole 16- / 4-byte stores.  Plain env, every shop acts, every customer orders from the device stream, shops with one
// customer count K <= 6 and one normaliser (DevSpec::sc_wide_K); anything else keeps phx_sc_step_kernel.  Same results bit for bit
// (tests/test_gpu_round4.py); supply_chain.py:98-147, env.py:239-303.
struct StepWideArgs {
  int32_t B, S, K, num_steps, epb, norm;
  uint32_t mS;                       // ceil(2^32 / S): i / S for i < 2^16
  uint64_t seed; int64_t env_offset;
  int32_t *stock, *sales, *missed, *delivered, *env_step, *env_tick;
  const float* sc_tab; int32_t n_quot;      // DevSpec::sc_tab: [0, 101) stock / 100, [101, 101 + n_quot) x / norm (n_quot <= 64 here)
  phx_step_io io;
};

__global__ __launch_bounds__(256) void phx_sc_step_wide_kernel(const StepWideArgs a) {
  __shared__ float s_tab[101 + 64];
  if ((int)threadIdx.x < 101 + a.n_quot) s_tab[threadIdx.x] = a.sc_tab[threadIdx.x];
  const int64_t b_first = (int64_t)blockIdx.x * a.epb;
  const int n_env = (int)((b_first + a.epb <= a.B) ? a.epb : a.B - b_first);
  const int u = (int)threadIdx.x, n_units = (n_env * a.S) >> 2;
  const bool active = u < n_units;
  const int64_t g0 = b_first * a.S + 4 * (int64_t)u;                       // the thread's first pair (a multiple of 4)
  int el[4], sh[4]; uint32_t tick0[4];
  int4 v_stock = make_int4(0, 0, 0, 0), v_deliv = v_stock;
  float4 v_act = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t v_valid = 0x01010101u;
  if (active) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t rel = (uint32_t)(4 * u + k);
      el[k] = a.S == 1 ? (int)rel : (int)__umulhi(rel, a.mS); sh[k] = (int)rel - el[k] * a.S;
      tick0[k] = (uint32_t)a.env_tick[b_first + el[k]];
    }
    v_stock = *(const int4*)(a.stock + g0); v_deliv = *(const int4*)(a.delivered + g0);
    if (a.io.actions) v_act = *(const float4*)(a.io.actions + g0);
    if (a.io.action_valid) v_valid = *(const uint32_t*)(a.io.action_valid + g0);
  }
  // the env words are rewritten by one thread per ENV (contiguous 4-byte stores; written from the lanes that hold an env's first shop they
  // were a million scattered 1- and 4-byte store requests per step at B = 2^18: 28.8 -> 24.9 us)
  int e_step[4], e_tick[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int e = (int)threadIdx.x + 256 * i; if (e < n_env) { e_step[i] = a.env_step[b_first + e]; e_tick[i] = a.env_tick[b_first + e]; } }
  __syncthreads();            // every thread has read its envs' words
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = (int)threadIdx.x + 256 * i;
    if (e < n_env) {                                                                                   // env.py:252,297-298
      const int t = e_step[i] + 1;
      a.env_step[b_first + e] = t; a.env_tick[b_first + e] = e_tick[i] + 1;
      a.io.all_terminated[b_first + e] = 0; a.io.all_truncated[b_first + e] = (uint8_t)(t == a.num_steps);
    }
  }
  if (!active) return;
  const int stock_in[4] = {v_stock.x, v_stock.y, v_stock.z, v_stock.w}, deliv_in[4] = {v_deliv.x, v_deliv.y, v_deliv.z, v_deliv.w};
  const float act_in[4] = {v_act.x, v_act.y, v_act.z, v_act.w};
  int o_stock[4], o_sales[4], o_missed[4], o_deliv[4];
  float ob[12]; double rw[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t b = b_first + el[k];
    const bool has_action = ((v_valid >> (8 * k)) & 0xffu) != 0u;                                     // env.py:330 (no action tensor: actions of 0)
    const int D = rng_shop_order_sum(a.seed, a.env_offset + b, tick0[k], sh[k], a.K, nullptr);      // supply_chain.py:61-67
    ShopLane st; st.stock = stock_in[k]; st.delivered = deliv_in[k]; st.sales = 0; st.missed = 0;
    sc_shop_step(st, has_action, has_action ? act_in[k] : 0.0f, true, D);      // (act_in is 0 without an action tensor)
    o_stock[k] = st.stock; o_sales[k] = st.sales; o_missed[k] = st.missed; o_deliv[k] = st.delivered;
    if ((unsigned)st.stock <= (unsigned)PHX_SHOP_MAX_STOCK && (unsigned)st.sales < (unsigned)a.n_quot && (unsigned)st.missed < (unsigned)a.n_quot) {   // encode_observation :124-134
      ob[3 * k] = s_tab[st.stock]; ob[3 * k + 1] = s_tab[101 + st.sales]; ob[3 * k + 2] = s_tab[101 + st.missed];
    } else shop_obs_f32(st.stock, st.sales, st.missed, (float)a.norm, ob + 3 * k);                   // a stock outside [0, 100] (the caller's, or an action below zero's)
    rw[k] = shop_reward(st.sales, st.stock);                                                         // compute_reward :147
  }
  *(int4*)(a.stock + g0) = make_int4(o_stock[0], o_stock[1], o_stock[2], o_stock[3]);
  *(int4*)(a.sales + g0) = make_int4(o_sales[0], o_sales[1], o_sales[2], o_sales[3]);
  *(int4*)(a.missed + g0) = make_int4(o_missed[0], o_missed[1], o_missed[2], o_missed[3]);
  *(int4*)(a.delivered + g0) = make_int4(o_deliv[0], o_deliv[1], o_deliv[2], o_deliv[3]);
  float4* const po = (float4*)(a.io.obs + g0 * 3);
  po[0] = make_float4(ob[0], ob[1], ob[2], ob[3]); po[1] = make_float4(ob[4], ob[5], ob[6], ob[7]); po[2] = make_float4(ob[8], ob[9], ob[10], ob[11]);
  double2* const pr = (double2*)(a.io.reward + g0);
  pr[0] = make_double2(rw[0], rw[1]); pr[1] = make_double2(rw[2], rw[3]);
  *(uint32_t*)(a.io.obs_valid + g0) = 0x01010101u; *(uint32_t*)(a.io.reward_valid + g0) = 0x01010101u; *(uint32_t*)(a.io.done_valid + g0) = 0x01010101u;
  *(uint32_t*)(a.io.terminated + g0) = 0u; *(uint32_t*)(a.io.truncated + g0) = 0u;
}

// ---- rollout: time-parallel.  The only sequential dependence of an episode is the stock
// recurrence  stock' = min(stock - min(D, stock) + min(R, 100 - stock), 100)  (a dozen integer
// ops); everything expensive -- the Philox draws, the divisions of the observation, the f64
// reward, the trajectory stores -- is independent across time steps.  A block owns G = epb * S
// (env, shop) pairs (whole envs) and walks the fragment in chunks of TC steps; LDS holds the
// chunk's item tiles as three planes [TC][G]: {R | stock, D | missed, sales} plus an action tile:
//   phase 1 (worker waves, item = (row quad, pair)): Philox -> R, D, action    (next chunk)
//   phase 2 (one lane per pair, sequential over t):  the recurrence, in LDS    (this chunk, overlapped)
//   phase 3 (all lanes, item = (row, 4 pairs)):      obs / reward / flags -> HBM, 16-byte stores
// lean argument block of the rollout kernel (passing the whole DevSpec by value costs ~50
// spilled SGPRs per wave)
struct RollArgs {
  int32_t B, S, n_exo, num_steps, T, epb, TC, n_tabn, n_quot;
  int32_t xcd_remap;
  const float* sc_tab;
  unsigned long long* timing;    // PHX_TIMING builds only: [blocks][8] cycle sums per phase
  uint32_t mF;                   // ceil(2^32 / (G / 4)): i / (G / 4) == umulhi(i, mF) for i < 2^16 (G > 4)
  uint64_t seed; int64_t env_offset;
  const int32_t* shop_norm;      // [S] max_sales_per_step of each shop
  const int32_t* shop_cust_ptr;  // [S+1]
  const int32_t* shop_cust_exo;
  int32_t *stock, *sales, *missed, *delivered, *env_step, *env_tick;
  const int32_t* only_if; int32_t gen;   // non-NULL: the kernel runs only if *only_if == gen (the store-wave kernel declined this call's replayed actions)
  phx_rollout_io io;
};

// Barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global
// store (s_waitcnt vmcnt(0)) because it is a workgroup-scope release; the tiles below need no
// global visibility inside the kernel, and draining would expose the HBM write latency at each
// barrier instead of letting the trajectory stores retire under the next chunk's Philox work.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// REPLAY: actions and/or exogenous draws come from HBM (parity / replay); false = pure device RNG.
// WIDE:   every block owns exactly epb envs and every tile row is 16-byte aligned (host-checked),
//         so the output phase writes whole 16-byte segments with magic-number row arithmetic only.
template <int NT, bool REPLAY, bool WIDE>
__global__ __launch_bounds__(NT, NT >= 1024 ? 8 : (NT >= 768 ? 6 : (NT == 384 ? 6 : (NT == 320 ? 5 : 4)))) void phx_sc_rollout_kernel(const RollArgs a) {
  if (a.only_if && *a.only_if != a.gen) return;   // (uniform) the store-wave kernel served this call
  // Software pipeline over chunks of TC steps.  Phase 1 (Philox draws) of chunk c + 1 does not
  // depend on the stock recurrence, so it runs on waves P1W.. while waves 0..P2W-1 walk the
  // recurrence (phase 2) of chunk c; item tiles {R|stock, D, sales} and the action tile are
  // double-buffered in LDS.  Per chunk:   [P2(c) || P1(c+1)]  bar  P3(c)  bar
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PHX_TIMING
  unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tm[k] += now_ - tprev; tprev = now_; } while (0)
#else
#define TICK(k) do {} while (0)
#endif
  const phx_rollout_io& io = a.io;
  const int nS = a.S, tid = threadIdx.x, TC = a.TC;
  const int64_t total = (int64_t)a.B * nS;
  // XCD-aware block -> env mapping: consecutive workgroup ids go round-robin to the 8 XCDs, each with its own
  // L2; giving XCD x the x-th contiguous eighth of the envs lets the partial cache lines at the boundary of two
  // neighbouring workgroups' trajectory segments merge in ONE L2 instead of reaching HBM from two
  const int bid = a.xcd_remap > 1 ? xcd_block_grouped(a.xcd_remap) : xcd_block(a.xcd_remap != 0);
  const int64_t b_first = (int64_t)bid * a.epb;
  const int64_t b_end = (WIDE || b_first + a.epb < a.B) ? b_first + a.epb : a.B;
  const int nb = (int)(b_end - b_first);
  const int Gfull = a.epb * nS, G = WIDE ? Gfull : nb * nS;
  const int64_t g_base = b_first * nS;

  // LDS carve (all offsets multiples of 16 bytes)
  const int items_max = TC * Gfull;
  const int it1 = (items_max + 3) & ~3, it_words = 3 * it1;
  int* s_it0 = (int*)smem;                                 // 2 x 3 planes [TC][G]: R | stock, D | missed, sales
  float* s_act0 = (float*)(s_it0 + 2 * it_words);          // 2 x [TC][G]
  uint32_t* s_pair = (uint32_t*)(s_act0 + 2 * it1);        // [G] shop | env_local << 8 | customers << 16
  int* s_tick0 = (int*)(s_pair + ((Gfull + 3) & ~3));     // [epb]
  int* s_step0 = s_tick0 + ((a.epb + 3) & ~3);             // [epb]
  int* s_tend = s_step0 + ((a.epb + 3) & ~3);              // [epb] chunk-local step index that ends the episode, or -1
  int* s_cptr = s_tend + ((a.epb + 3) & ~3);               // [S+1]
  int* s_norm = s_cptr + ((nS + 1 + 3) & ~3);              // [S]
  uint8_t* s_ds3 = (uint8_t*)(s_norm + ((nS + 3) & ~3));   // [125] base-5 digit sum of k < 5^3
  const int n_tab = 101 + a.n_tabn + 202;
  float* s_tab = (float*)(s_ds3 + 128);                    // [n_tab] host-built lookup tables
  const float* s_tabn = s_tab + 101;
  const double* s_pen = (const double*)(s_tabn + a.n_tabn);   // 0.1 * stock, f64
  for (int k = tid; k < n_tab; k += NT) s_tab[k] = a.sc_tab[k];
  for (int k = tid; k < 125; k += NT) s_ds3[k] = (uint8_t)(k % 5 + (k / 5) % 5 + k / 25);

  for (int gl = tid; gl < G; gl += NT) {
    const int s2 = gl % nS, kk = a.shop_cust_ptr[s2 + 1] - a.shop_cust_ptr[s2];
    s_pair[gl] = (uint32_t)s2 | ((uint32_t)(gl / nS) << 8) | ((uint32_t)(kk < 65535 ? kk : 65535) << 16);
  }
  for (int bl = tid; bl < nb; bl += NT) { s_tick0[bl] = a.env_tick[b_first + bl]; s_step0[bl] = a.env_step[b_first + bl]; }
  for (int k = tid; k <= nS; k += NT) s_cptr[k] = a.shop_cust_ptr[k];
  for (int k = tid; k < nS; k += NT) s_norm[k] = a.shop_norm[k];

  // phase-2 lane state: lane `tid` owns pair g_base + tid
  ShopLane st = {0, 0, 0, 0};
  int step = 0, p2_K = 0;
  if (tid < G) {
    const int64_t g = g_base + tid;
    st.stock = a.stock[g]; st.sales = a.sales[g]; st.missed = a.missed[g]; st.delivered = a.delivered[g];
    step = a.env_step[b_first + tid / nS];
    const int s2 = tid % nS;
    p2_K = a.shop_cust_ptr[s2 + 1] - a.shop_cust_ptr[s2];
  }
  // thread roles in the overlapped phase: the waves that hold a recurrence lane do phase 2, the
  // others phase 1 (when no wave is left over, everybody does phase 1 after phase 2)
  const int p2_threads = ((G + 63) >> 6) << 6;
  const int p1_first = (p2_threads + 64 <= NT) ? p2_threads : 0;
  lds_barrier();
  // row quads per chunk: one more when the chunk starts are not quad-aligned for every env (some
  // tick counter, or the chunk length, is not a multiple of 4)
  int quad_extra = (TC & 3) != 0;
  for (int bl = 0; bl < nb; ++bl) quad_extra |= (s_tick0[bl] & 3) != 0;
  TICK(0);

  // ---- phase 1 of the chunk starting at step t0 (tc rows) into buffer `buf`, by threads
  //      [first, NT).  One Philox block serves the four ticks (4q .. 4q + 3) of a shop, so the flat
  //      work items are (row quad jr, pair gl): rows tla = 4 jr - e .. tla + 3 of the chunk, e =
  //      the env's tick at chunk row 0 modulo 4.  Thread-strided; (jr, gl) advance incrementally.
  auto phase1 = [&](int t0, int tc, int buf, int first) {
    if (tid < first) return;
    int* s_it = s_it0 + buf * it_words;
    float* s_act = s_act0 + buf * it1;
    const int nw = NT - first, wt = tid - first;
    const int npr = ((tc + 3) >> 2) + quad_extra;
    const int n_work = npr * G;
    int jr = wt / G, gl = wt - jr * G;
    const int qG = nw / G, rG = nw - qG * G;
    for (int iw = wt; iw < n_work; iw += nw) {
      const uint32_t pr = s_pair[gl];
      const int s = (int)(pr & 255u), bl = (int)((pr >> 8) & 255u);
      const int b = (int)b_first + bl;
      const int64_t genv = a.env_offset + b;
      const uint32_t tick_base = (uint32_t)s_tick0[bl] + (uint32_t)t0;
      const int e = (int)(tick_base & 3u);
      const int tla = 4 * jr - e;
      if (tla + 3 >= 0 && tla < tc) {
        const uint32_t tick_a = tick_base + (uint32_t)tla;           // multiple of 4
        const int K = (int)(pr >> 16);
        // replayed actions: the quad's four loads are issued before the Philox block hides them
        float av[4] = {0.f, 0.f, 0.f, 0.f};
        if (REPLAY && io.actions) {
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int tl = tla + h;
            if (tl >= 0 && tl < tc) av[h] = io.actions[(int64_t)(t0 + tl) * total + g_base + gl];
          }
        }
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (!(REPLAY && io.exo && io.actions)) rng_block(a.seed, genv, tick_a, s, 0, 0, w);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int tl = tla + h;
          if (tl < 0 || tl >= tc) continue;
          const int t = t0 + tl, i = tl * G + gl;
          int D = 0; uint32_t y = 0, aj = 0;
          if (!(REPLAY && io.exo && io.actions) && !rng_split(w[h], y, aj))
            y = rng_group_y(a.seed, genv, tick_a + (uint32_t)h, s, 0, 1, &aj);                    // 3.3e-6
          if (REPLAY && io.exo) {
            const uint8_t* row = io.exo + ((int64_t)t * a.B + b) * a.n_exo;
            for (int k = s_cptr[s]; k < s_cptr[s + 1]; ++k) D += row[a.shop_cust_exo[k]];
          } else if (K > 0) {
            if (__all(K >= 6)) {
              // digit sum of y < 5^6 from the 125-entry table: y = 125 hi + lo (y / 125 through f32 is exact)
              const uint32_t hi = (uint32_t)((float)y * 0.008f);
              D = (int)s_ds3[hi] + (int)s_ds3[y - 125u * hi];
            } else {
              D = rng_digit_sum(y, K < 6 ? K : 6, nullptr);
            }
            for (int g = 1; 6 * g < K; ++g)
              D += rng_digit_sum(rng_group_y(a.seed, genv, tick_a + (uint32_t)h, s, g, 0), K - 6 * g < 6 ? K - 6 * g : 6, nullptr);
          }
          const float action = (REPLAY && io.actions) ? av[h] : rng_j_to_action(aj);
          s_act[i] = action;
          // the random-policy action lies in [0, 100): no saturation needed before the conversion
          s_it[i] = (REPLAY && io.actions) ? dev_round_half_even(action) : (int)rintf(action);
          s_it[it1 + i] = D;
        }
      }
      jr += qG; gl += rG;
      if (gl >= G) { gl -= G; ++jr; }
    }
  };

  phase1(0, a.T < TC ? a.T : TC, 0, 0);
  TICK(1); lds_barrier(); TICK(2);
  int buf = 0;
  for (int t0 = 0; t0 < a.T; t0 += TC, buf ^= 1) {
    const int tc = (a.T - t0 < TC) ? a.T - t0 : TC;
    const int n_items = tc * G;
    int* s_it = s_it0 + buf * it_words;
    const float* s_act = s_act0 + buf * it1;
    const int t1 = t0 + TC, tc1 = (a.T - t1 < TC) ? a.T - t1 : TC;     // next chunk
    // ---- phase 2: the stock recurrence, one lane per pair ---------------------------------------
    if (tid < G) {
#ifndef PHX_NO_P2_PRIO
      // the recurrence is a dependent chain on ONE wave while the three other waves of its SIMD issue Philox
      // work: at equal priority it gets every fourth issue slot (measured 212 cycles per step); raised, it
      // issues as soon as its operands are ready and the draw waves fill the gaps
      __builtin_amdgcn_s_setprio(3);
#endif
      // A lone wave issues about one instruction every 4-5 cycles, so this phase costs
      // (instructions per step) x T: the loop body is kept to two LDS instructions and seven
      // VALU ops.  stock' = min(max(x - D, 0) + min(R, 100 - x), 100); sales = x - max(x - D, 0).
      // The episode end (at most one per chunk: TC <= num_steps) is a compare on the local index.
      const int tend = a.num_steps - 1 - step;
      if (tid % nS == 0) s_tend[tid / nS] = (tend >= 0 && tend < tc) ? tend : -1;
      int x = st.stock, sales = st.sales, Dl = 0, req = st.delivered;
      int* it = s_it + tid;
      const int istride = G;
      const bool hasK = p2_K > 0;
      // dependent chain per step: sub, max, add, min, and  (the reset is an AND with a mask that
      // does not depend on the stock; shops without customers take the general select)
#define P2_STEP(R_, D_, tl_, it_, HASK)                                                          \
      {                                                                                          \
        Dl = (D_);                                                                               \
        const int a0 = (HASK) ? max(x - Dl, 0) : (hasK ? max(x - Dl, 0) : x);  /* handle_order_request :105-122 */ \
        const int keep = ((tl_) == tend) ? 0 : -1;                /* episode end -> env.reset(): stock = 0 */ \
        req = min((R_), PHX_SHOP_MAX_STOCK - x);                  /* decode_action         :139 */     \
        sales = x - a0;                                                                          \
        const int xn = min(a0 + req, PHX_SHOP_MAX_STOCK);         /* handle_stock_response :98-103 */  \
        (it_)[0] = xn; (it_)[it1] = (HASK || hasK) ? Dl - sales : 0; (it_)[2 * it1] = sales;       \
        x = xn & keep;                                                                           \
      }
      int tl = 0;
      if (__all(hasK)) {
        for (; tl + 8 <= tc; tl += 8, it += 8 * istride) {         // the 8 reads of a group are issued together
          int Rv[8], Dv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { Rv[u] = it[u * istride]; Dv[u] = it[u * istride + it1]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) P2_STEP(Rv[u], Dv[u], tl + u, it + u * istride, true)
        }
      }
      for (; tl < tc; ++tl, it += istride) P2_STEP(it[0], it[it1], tl, it, false)
#undef P2_STEP
      st.stock = x; st.sales = sales; st.missed = hasK ? Dl - sales : 0; st.delivered = req;
      step += tc;
      if (tend >= 0 && tend < tc) step -= a.num_steps;
#ifndef PHX_NO_P2_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    TICK(3);
    // ---- phase 1 of the NEXT chunk, overlapped with the recurrence above -------------------------
    if (t1 < a.T) phase1(t1, tc1, buf ^ 1, p1_first);
    TICK(1); lds_barrier(); TICK(4);
    // ---- phase 3: observations / rewards / flags straight to HBM ------------------------------------
    // WIDE: a work unit is 4 consecutive pairs of one tile row: 12 observation floats, 4 rewards,
    // 4 actions and 4 + 4 flag bytes, i.e. whole 16-byte (4-byte for the flags) segments of the
    // [T][B][S] arrays, computed in registers from the {stock, D, sales} tile and written with
    // dwordx4 stores -- no staging of the outputs in LDS and no second pass.
    const int64_t row0 = (int64_t)t0 * total + g_base;           // element offset of tile row 0
    const int64_t rstride = total;
    auto item_out = [&](int stock, int missed, int sales, int s, float* ob, float& rew) {
      // observation / reward from the host-built tables (= the reference's formulas evaluated on
      // the host) when the operands are in their usual range; otherwise (negative stock from
      // negative requests, ...) the formulas themselves: f32 IEEE division == the reference's f64
      // quotient cast to f32 for |ints| < 2^24
      const bool in100 = (unsigned)stock <= 100u;
      const bool ins = (unsigned)sales < (unsigned)a.n_quot, inm = (unsigned)missed < (unsigned)a.n_quot;
      if (in100 && ins && inm) {
        ob[0] = s_tab[stock]; ob[1] = s_tabn[sales]; ob[2] = s_tabn[missed];
        // reward = sales - 0.1 * stock in f64 (supply_chain.py:147), rounded once to the trajectory's f32
        rew = (float)__dsub_rn((double)sales, s_pen[stock]);
      } else {
        const float norm = (float)s_norm[s];
        ob[0] = in100 ? s_tab[in100 ? stock : 0] : (float)stock / (float)PHX_SHOP_MAX_STOCK;
        ob[1] = ins ? s_tabn[ins ? sales : 0] : (float)sales / norm;
        ob[2] = inm ? s_tabn[inm ? missed : 0] : (float)missed / norm;
        rew = in100 ? (float)__dsub_rn((double)sales, s_pen[in100 ? stock : 0]) : (float)shop_reward(sales, stock);
      }
    };
    if (WIDE) {
      const int G4 = G >> 2;
      for (int u = tid; u < tc * G4; u += NT) {
        const int r = G4 == 1 ? u : (int)__umulhi((uint32_t)u, a.mF);   // u / G4 (the magic of 1 does not fit 32 bits)
        const int c4 = u - r * G4, gl0 = c4 << 2, i0 = r * G + gl0;
        const uint4 vs = *(const uint4*)(s_it + i0), vd = *(const uint4*)(s_it + it1 + i0), vl = *(const uint4*)(s_it + 2 * it1 + i0);
        const uint4 pp = *(const uint4*)(s_pair + gl0);
        const int p0 = (int)(pp.x & 0xffffu), p1 = (int)(pp.y & 0xffffu), p2 = (int)(pp.z & 0xffffu), p3 = (int)(pp.w & 0xffffu);
        float o[12], rw[4];
        item_out((int)vs.x, (int)vd.x, (int)vl.x, p0 & 255, o + 0, rw[0]);
        item_out((int)vs.y, (int)vd.y, (int)vl.y, p1 & 255, o + 3, rw[1]);
        item_out((int)vs.z, (int)vd.z, (int)vl.z, p2 & 255, o + 6, rw[2]);
        item_out((int)vs.w, (int)vd.w, (int)vl.w, p3 & 255, o + 9, rw[3]);
        const uint32_t tr = (uint32_t)(r == s_tend[p0 >> 8]) | ((uint32_t)(r == s_tend[p1 >> 8]) << 8) |
                            ((uint32_t)(r == s_tend[p2 >> 8]) << 16) | ((uint32_t)(r == s_tend[p3 >> 8]) << 24);   // truncations["__all__"], env.py:312-318
        const int64_t e0 = row0 + (int64_t)r * rstride + gl0;
        float4* po = (float4*)(io.obs + e0 * 3);
        po[0] = make_float4(o[0], o[1], o[2], o[3]); po[1] = make_float4(o[4], o[5], o[6], o[7]); po[2] = make_float4(o[8], o[9], o[10], o[11]);
        *(float4*)(io.reward + e0) = make_float4(rw[0], rw[1], rw[2], rw[3]);
        *(float4*)(io.action_out + e0) = *(const float4*)(s_act + i0);
        *(uint32_t*)(io.truncated + e0) = tr;
        *(uint32_t*)(io.terminated + e0) = 0u;
      }
    } else {
      int tl = tid / G, gl = tid - tl * G;
      const int qG = NT / G, rG = NT - qG * G;
      for (int i = tid; i < n_items; i += NT) {
        const int pr = (int)(s_pair[gl] & 0xffffu);
        float o[3], rw;
        item_out(s_it[i], s_it[it1 + i], s_it[2 * it1 + i], pr & 255, o, rw);
        const int64_t e0 = row0 + (int64_t)tl * rstride + gl;
        io.obs[e0 * 3] = o[0]; io.obs[e0 * 3 + 1] = o[1]; io.obs[e0 * 3 + 2] = o[2];
        io.reward[e0] = rw; io.action_out[e0] = s_act[i];
        io.truncated[e0] = (uint8_t)(tl == s_tend[pr >> 8]); io.terminated[e0] = 0;
        tl += qG; gl += rG;
        if (gl >= G) { gl -= G; ++tl; }
      }
    }
    TICK(6); lds_barrier(); TICK(7);
  }
#ifdef PHX_TIMING
  if (a.timing && (tid & 63) == 0) for (int q = 0; q < 8; ++q) a.timing[((int64_t)blockIdx.x * (NT / 64) + (tid >> 6)) * 8 + q] = tm[q];
#endif
  if (tid < G) {
    const int64_t g = g_base + tid;
    const int s = tid % nS, b = (int)b_first + tid / nS;
    a.stock[g] = st.stock; a.sales[g] = st.sales; a.missed[g] = st.missed; a.delivered[g] = st.delivered;
    if (io.last_obs) {
      float ob[3];
      shop_obs(st.stock, st.sales, st.missed, a.shop_norm[s], ob);
      io.last_obs[g * 3 + 0] = ob[0]; io.last_obs[g * 3 + 1] = ob[1]; io.last_obs[g * 3 + 2] = ob[2];
    }
    if (s == 0) {       // env_step / env_tick were read (into LDS / registers) before the first barrier
      a.env_step[b] = step;
      a.env_tick[b] = s_tick0[tid / nS] + a.T;
    }
  }
}

// ---- FSM rollout: T steps per launch for FiniteStateMachineEnv supply chains (BASELINE config 3)
// and for envs with typed shops (tutorial 2; a plain env is the one-stage special case).
// One lane owns one (env, shop) and walks the steps in order: the stage masks (who acts, who
// observes, who is rewarded -- fsm.py:276-345), the reward cache with emit-on-observe and the
// terminal dump of the cached dicts (fsm.py:349-378) are all sequential in time, so this kernel
// keeps the lane-per-pair loop and relies on the batch for parallelism (SC256 x B = 8192 gives
// 6 500 waves).  Per-(stage, shop) mask bits are staged in LDS once per block.
// RULES (round 6): stage handlers declared in rule form (phx_spec.stage_rules: "restock while the shops together hold
// fewer than 60 items", fsm.py:294-307) evaluated HERE, on the state the step's messages left: every rule's value -- one shop's field or
// the sum over the env's shops -- is accumulated by the env's lanes with LDS atomics into a double-buffered row (one workgroup barrier per
// step: a block holds whole envs), the first rule of the stage that holds picks the next stage, and the agents acting in THAT stage are
// the ones that observe (fsm.py:320).  Until round 6 such envs rolled out on the message-passing engine only.
// STATIC (round 6): device-drawn actions and orders, no tabulated handler, no typed shop, every shop's customers acting all or none per
// stage (phx_sc_fsm_static) -- the per-stage words (stage_next, stage_rew_all) are staged in LDS and the step loop holds NO global load.
// The general form reads them -- and the replayed planes, the per-customer masks, the samplers' parameters -- from global memory behind
// run-time conditions; the s_waitcnt vmcnt(0) behind each such load also waits for the previous step's row stores (loads and stores share
// the counter on gfx950), a chain of up to eight L2 round trips per step whether or not the loads are executed.
template <bool RULES, bool STATIC>
__global__ __launch_bounds__(SC_NT) void phx_sc_rollout_fsm_kernel(const DevSpec sp, const phx_rollout_io io,
                                                                  const int epb, const int remap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_fl[];   // [n_lists][S]
  const int nS = sp.S, A = sp.A, nL = sp.n_lists;
  const int n_rules = RULES ? sp.n_rules : 0;
  int* const s_red = (int*)(s_fl + (((size_t)nL * nS + 15) & ~(size_t)15));   // RULES: [3][epb][n_rules] sums, rows rotate step by step
  int* const s_nx = s_red + (RULES ? 3 * epb * n_rules : 0);                  // STATIC: [n_lists] stage_next | stage_rew_all << 16
  // RULES: the rules themselves, [n_rules] DevRule (40 bytes each, 8-byte aligned): the loop that looks for the first rule that holds ends
  // per lane, so its index is a VGPR to the compiler and sp.rules[r] a vector load from global memory
  DevRule* const s_rules = (DevRule*)((char*)s_fl + (((int)((char*)(s_nx + (STATIC ? nL : 0)) - (char*)s_fl) + 7) & ~7));
  if (RULES) for (int idx = threadIdx.x; idx < n_rules * (int)(sizeof(DevRule) / 4); idx += SC_NT) ((int*)s_rules)[idx] = ((const int*)sp.rules)[idx];
  if (RULES) for (int idx = threadIdx.x; idx < 3 * epb * n_rules; idx += SC_NT) s_red[idx] = 0;
  if (STATIC) for (int idx = threadIdx.x; idx < nL; idx += SC_NT) s_nx[idx] = (sp.stage_next[idx] & 0xFFFF) | (sp.stage_rew_all[idx] ? 0x10000 : 0);
  for (int idx = threadIdx.x; idx < nL * nS; idx += SC_NT) {
    const int l = idx / nS, s = idx - l * nS, a_shop = sp.shop_agent[s];
    const uint8_t* cact = sp.shop_cust_act + (int64_t)l * sp.n_exo;
    bool any = false, all = true;
    for (int k = sp.shop_cust_ptr[s]; k < sp.shop_cust_ptr[s + 1]; ++k) { any |= cact[k] != 0; all &= cact[k] != 0; }
    s_fl[idx] = (uint8_t)((sp.act_mask[(int64_t)l * A + a_shop] ? 1 : 0) | (any ? 2 : 0) | ((any && all) ? 4 : 0) |
                          (sp.obs_mask[(int64_t)l * A + a_shop] ? 8 : 0) | (sp.rew_mask[(int64_t)l * A + a_shop] ? 16 : 0));
  }
  const int64_t total = (int64_t)sp.B * nS;
  const int64_t b_first = (int64_t)xcd_block(remap != 0) * epb;
  const int64_t b_end = (b_first + epb < sp.B) ? b_first + epb : sp.B;
  const bool active = (int)threadIdx.x < (int)(b_end - b_first) * nS;
  const int64_t g = b_first * nS + (active ? (int)threadIdx.x : 0);
  const int b = active ? (int)(b_first + threadIdx.x / nS) : (int)b_first;
  const int s = active ? (int)(threadIdx.x % nS) : 0;
  int step = fld<int32_t>(sp, F_ENV_STEP)[b];
  uint32_t tick = (uint32_t)fld<int32_t>(sp, F_ENV_TICK)[b];
  int stage = fld<int32_t>(sp, F_ENV_STAGE)[b];
  int prev_stage = fld<int32_t>(sp, F_ENV_PREV_STAGE)[b];
  __syncthreads();          // mask bits staged; per-env words read before their shop-0 lane rewrites them
  if (!RULES && !active) return;       // (RULES: every lane stays for the per-step barrier; lanes past the block's envs store nothing)
  const int el = active ? (int)(threadIdx.x / nS) : 0;
  int red_p = 0;

  const int a_shop = sp.shop_agent[s];
  const float norm = (float)sp.param_i[a_shop * PHX_NPI + 1];
  const int c_lo = sp.shop_cust_ptr[s], c_hi = sp.shop_cust_ptr[s + 1], K = c_hi - c_lo;
  const int64_t genv = sp.env_offset + b;
  ShopLane st;
  st.stock = fld<int32_t>(sp, F_SHOP_STOCK)[g]; st.sales = fld<int32_t>(sp, F_SHOP_SALES)[g];
  st.missed = fld<int32_t>(sp, F_SHOP_MISSED)[g]; st.delivered = fld<int32_t>(sp, F_SHOP_DELIVERED)[g];
  double rc = fld<double>(sp, F_ENV_REW_CACHE)[g];                           // self._rewards[aid]
  uint8_t rcv = fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[g];
  const int OD = sp.D;            // observation length (D below is the demand)
  float oc[4] = {fld<float>(sp, F_ENV_OBS_CACHE)[g * OD], fld<float>(sp, F_ENV_OBS_CACHE)[g * OD + 1],
                 fld<float>(sp, F_ENV_OBS_CACHE)[g * OD + 2],
                 OD == 4 ? fld<float>(sp, F_ENV_OBS_CACHE)[g * OD + 3] : 0.f};   // self._observations[aid]
  uint8_t ocv = fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID)[g];
  float lo[4] = {0.f, 0.f, 0.f, 0.f};
  // typed shop: weight and 4th observation from the sampler column, redrawn at every auto-reset
  const int tsrc = (!STATIC && sp.any_typed) ? sp.shop_type_src[s] : PHX_TYPE_NONE;
  const bool typed = tsrc != PHX_TYPE_NONE;
  double tw = typed ? shop_type_value(sp, b, s) : 0.0;
  const double tnorm = typed ? sp.shop_type_prm[2 * s + 1] : 1.0;
  float tobs = typed ? (float)(tw / tnorm) : 0.f;
  uint32_t episode = sp.n_samplers > 0 ? (uint32_t)fld<int32_t>(sp, F_ENV_EPISODE)[b] : 0u;
  int n_resets = 0;
  RngQuadCache rc_rng; rc_rng.q = 0xffffffffu;

  for (int t = 0; t < io.T; ++t) {
    const int64_t o = (int64_t)t * total + g;
    int fl = s_fl[stage * nS + s];
    const int nxw = STATIC ? s_nx[stage] : 0;
    int next_stage = STATIC ? (nxw & 0xFFFF) : sp.stage_next[stage];
    const bool rew_all = STATIC ? (nxw >> 16) != 0 : false;
    if (!STATIC && sp.stage_tab) {                             // a tabulated clock / stage handler's choice, fsm.py:294-302,320
      next_stage = sp.stage_tab[(int64_t)stage * (sp.num_steps + 1) + (step + 1 <= sp.num_steps ? step + 1 : sp.num_steps)];
      fl = (fl & ~8) | ((sp.stage_rew_all[stage] || (s_fl[next_stage * nS + s] & 1)) ? 8 : 0);
    }
    const bool has_action = (fl & 1) != 0, any_order = (fl & 2) != 0;
    const uint8_t* cact = sp.shop_cust_act + (int64_t)stage * sp.n_exo;
    int D = 0; uint32_t aj = 0;
    const bool need_orders = (STATIC || !io.exo) && any_order;
    if (STATIC || need_orders || !io.actions) {            // one Philox block per four ticks
      rng_quad_block(rc_rng, sp.seed, genv, tick, s);
      const int Dr = rng_orders_from_block(rc_rng.w, sp.seed, genv, tick, s, need_orders ? K : 0,
                                           (STATIC || (fl & 4)) ? nullptr : cact + c_lo, &aj);
      if (need_orders) D = Dr;
    }
    if (!STATIC && io.exo) {
      const uint8_t* row = io.exo + ((int64_t)t * sp.B + b) * sp.n_exo;
      if (any_order) for (int k = c_lo; k < c_hi; ++k) if (cact[k]) D += row[sp.shop_cust_exo[k]];
    }
    const float action = (!STATIC && io.actions) ? io.actions[o] : rng_j_to_action(aj);
    sc_shop_step(st, has_action, action, any_order, D);
    if (RULES) {
      // the rules' values on the RESOLVED state (what env_handler() reads, fsm.py:294-302): this step's row of sums, one LDS atomic per rule
      // and lane.  Three rows rotate: the one zeroed here for the NEXT step was last read two steps ago -- every wave finished those reads
      // before the previous step's barrier (with two rows a fast wave would zero what a slow one is still reading)
      int* const row = s_red + (red_p * epb + el) * n_rules;
      int* const nxt = s_red + ((red_p == 2 ? 0 : red_p + 1) * epb + el) * n_rules;
      for (int r = 0; r < n_rules; ++r) {
        const DevRule q = s_rules[r];
        if (active && s == 0) nxt[r] = 0;
        if (!active || q.stage != stage || (q.col >= 0 && q.col != s)) continue;
        const int x = q.field_id == F_SHOP_STOCK ? st.stock : q.field_id == F_SHOP_SALES ? st.sales : q.field_id == F_SHOP_MISSED ? st.missed : st.delivered;
        atomicAdd(&row[r], x);
      }
      __syncthreads();
      int chosen = -1;
      for (int r = 0; r < n_rules && chosen < 0; ++r) {
        const DevRule q = s_rules[r];
        if (q.stage != stage) continue;
        const double v = (double)row[r];                        // (i32 fields: the sum is exact whatever the order)
        const bool hit = q.cmp == PHX_CMP_LT ? v < q.threshold : q.cmp == PHX_CMP_LE ? v <= q.threshold : q.cmp == PHX_CMP_GT ? v > q.threshold :
                         q.cmp == PHX_CMP_GE ? v >= q.threshold : q.cmp == PHX_CMP_EQ ? v == q.threshold : v != q.threshold;
        if (hit) chosen = q.next_stage;
      }
      red_p = red_p == 2 ? 0 : red_p + 1;
      if (chosen >= 0) {                                         // the handler chose: the agents acting in THAT stage observe (fsm.py:320)
        next_stage = chosen;
        fl = (fl & ~8) | (((STATIC ? rew_all : sp.stage_rew_all[stage] != 0) || (s_fl[next_stage * nS + s] & 1)) ? 8 : 0);
      }
    }
    ++step; ++tick;
    const bool all_trunc = (step == sp.num_steps);                           // env.py:312-318
    float ob[4] = {0.f, 0.f, 0.f, 0.f};
    uint8_t ov = 0, rv = 0; double rw = 0.0;
    const bool observes = (fl & 8) != 0;
    if (observes) {                                                          // fsm.py:328-332,349
      shop_obs_f32(st.stock, st.sales, st.missed, norm, ob);
      ob[3] = tobs;
      oc[0] = ob[0]; oc[1] = ob[1]; oc[2] = ob[2]; oc[3] = ob[3]; ocv = 1;
    }
    if (fl & 16) { rc = typed ? shop_reward_w(st.sales, st.stock, tw) : shop_reward(st.sales, st.stock); rcv = 1; }   // fsm.py:334-335,350
    if (all_trunc) {                                                         // fsm.py:360-375
      ov = ocv; ob[0] = ocv ? oc[0] : 0.f; ob[1] = ocv ? oc[1] : 0.f; ob[2] = ocv ? oc[2] : 0.f; ob[3] = ocv ? oc[3] : 0.f;
      rv = rcv ? 1 : 2; rw = rcv ? rc : 0.0;
    } else if (observes) {                                                   // fsm.py:378
      ov = 1; rv = rcv ? 1 : 2; rw = rcv ? rc : 0.0;
    }
    if (!RULES || active) {
    if (OD == 4) *(float4*)(io.obs + o * 4) = make_float4(ob[0], ob[1], ob[2], ob[3]);
    else { io.obs[o * 3 + 0] = ob[0]; io.obs[o * 3 + 1] = ob[1]; io.obs[o * 3 + 2] = ob[2]; }
    io.action_out[o] = action;
    io.reward[o] = (float)rw;
    io.terminated[o] = 0; io.truncated[o] = all_trunc;
    if (io.obs_valid) io.obs_valid[o] = ov;
    if (io.reward_valid) io.reward_valid[o] = rv;
    }
    lo[0] = ob[0]; lo[1] = ob[1]; lo[2] = ob[2]; lo[3] = ob[3];
    prev_stage = stage; stage = next_stage;                                  // fsm.py:355
    if (all_trunc) {                                                         // the caller's env.reset(), fsm.py:195-251
      st.stock = 0; step = 0; stage = sp.initial_stage; rcv = 0;
      if (!STATIC && tsrc >= 0) {                                            // env.py:211-212, agents.py:167-168
        tw = rng_uniform(sp.seed, genv, episode, tsrc, sp.sampler_param + 4 * tsrc);
        tobs = (float)(tw / tnorm);
      }
      ++episode; ++n_resets;
      lo[0] = lo[1] = lo[2] = lo[3] = 0.f;
      if (s_fl[sp.initial_stage * nS + s] & 1) { shop_obs_f32(st.stock, st.sales, st.missed, norm, lo); lo[3] = tobs; }
    }
  }
  if (RULES && !active) return;
  fld<int32_t>(sp, F_SHOP_STOCK)[g] = st.stock; fld<int32_t>(sp, F_SHOP_SALES)[g] = st.sales;
  fld<int32_t>(sp, F_SHOP_MISSED)[g] = st.missed; fld<int32_t>(sp, F_SHOP_DELIVERED)[g] = st.delivered;
  fld<double>(sp, F_ENV_REW_CACHE)[g] = rc; fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[g] = rcv;
  for (int d = 0; d < OD; ++d) fld<float>(sp, F_ENV_OBS_CACHE)[g * OD + d] = oc[d];
  fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID)[g] = ocv;
  if (io.last_obs) for (int d = 0; d < OD; ++d) io.last_obs[g * OD + d] = lo[d];
  if (s == 0) {
    fld<int32_t>(sp, F_ENV_STEP)[b] = step; fld<int32_t>(sp, F_ENV_TICK)[b] = (int32_t)tick;
    if (sp.env_type == PHX_ENV_FSM) { fld<int32_t>(sp, F_ENV_STAGE)[b] = stage; fld<int32_t>(sp, F_ENV_PREV_STAGE)[b] = prev_stage; }
    if (sp.n_samplers > 0 && n_resets > 0) {      // every column as drawn at the last auto-reset
      fld<int32_t>(sp, F_ENV_EPISODE)[b] = (int32_t)episode;
      for (int j = 0; j < sp.n_samplers; ++j)
        fld<double>(sp, F_ENV_SAMPLER)[(int64_t)b * sp.n_samplers + j] =
            rng_uniform(sp.seed, genv, episode - 1u, j, sp.sampler_param + 4 * j);
    }
  }
}

// ---- the same loop for the common FSM supply chain, lean.  The kernel above keeps every case in its step loop (exogenous
// and action replay, per-customer acting masks, more than six customers per shop, typed shops): ~180 instructions per
// step and lane through a maze of branches, and with the stores removed the launch still takes 222 of 272 us (SC256-FSM,
// B = 8192, T = 100) -- it is bound by its instruction count.  This variant serves device-RNG random-policy rollouts of
// specs where every shop has the same 1..6 customers and normaliser and a shop's customers act all or none per stage
// (DevSpec::fsm_lean_K): one Philox block per four ticks, the order sum as a digit sum of y mod 5^K, observations from
// LDS tables (computed at setup: exactly the f32 quotients), stores through scalar row bases + 32-bit lane offsets.
__global__ __launch_bounds__(SC_NT) void phx_sc_rollout_fsm_lean_kernel(const DevSpec sp, const phx_rollout_io io,
                                                                       const int epb, const int remap, const uint32_t pK,
                                                                       const float inv_pK, const int wide, const int32_t* only_if, const int32_t gen,
                                                                       const int pairs, const int batch) {
  // `pairs`: a block owns SC_NT CONSECUTIVE (env, shop) PAIRS instead of `epb` whole envs.  Every row segment a block writes
  // is then whole 128-byte lines of the f32 planes and whole 64-byte pieces of the u8 planes, every lane is active (4 envs
  // of 51 shops filled 204 of 256), and the launch no longer depends on WHERE the trajectory buffers lie: the partially
  // written lines at whole-env block boundaries (816-byte segments at SC256) have to be merged in the L2 before they are
  // evicted, which works or fails with the planes' physical placement -- config 3's two modes, 206-232 vs ~335 us per 100
  // steps.  An env's words are then written by the block that finishes LAST with it (env.arrive, as in
  // phx_sc_rollout_fast_kernel): another block of the env may not have read them yet.
  if (only_if && *only_if != gen) return;   // the time-parallel kernel took this launch (phx_sc_rollout_fsm.hip)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int nS = sp.S, nL = sp.n_lists, K = sp.fsm_lean_K;
  float* s_tabs = (float*)s_raw;                           // [101] stock / 100
  float* s_tabn = s_tabs + 104;                            // [5K + 1] x / norm
  unsigned char* s_fl = (unsigned char*)(s_tabn + 32);     // [n_lists][S] sc_shop_flags
  int* s_next = (int*)(s_fl + ((nL * nS + 3) & ~3));       // [n_lists] stage_next
  // `wide` (blocks start on multiples of 4 pairs): a wave's 64 x 3 observation floats of a step leave as 48 consecutive
  // 16-byte pieces through a wave-private LDS tile -- with its stores removed this loop takes 125 of 242 us, so the
  // memory system's cost per store instruction (three 4-byte stores at a 12-byte lane stride) is what it pays for
  float* s_ot = (float*)(s_next + ((nL + 3) & ~3)) + (threadIdx.x >> 6) * 192;
  // `batch` (round 5; blocks of consecutive pairs, B S a multiple of 16): a full wave keeps FOUR steps' outputs in a wave-private LDS
  // tile -- observations [4][64][3] f32, rewards and actions [4][64] f32, the four u8 planes [4][4][64] -- and writes them as 16-byte
  // pieces: 3 + 1 + 1 + 1 store instructions per wave and four steps instead of 4 x (0.75 + 1 + 1 + 4).  The loop is bound by what
  // the memory system charges per store INSTRUCTION (with its stores removed it takes 125 of 213-244 us at config 3), not per byte.
  // (the tile starts on a 16-byte boundary and is addressed in whole 16-byte elements from the start of the LDS where it is read as
  //  pieces, so that the compiler emits ds_read_b128)
  const uint32_t bt_off = ((uint32_t)((const unsigned char*)((float*)(s_next + ((nL + 3) & ~3)) + (SC_NT / 64) * 192) - s_raw) + 15u) & ~15u;
  unsigned char* const s_bt = s_raw + bt_off + (threadIdx.x >> 6) * 6144;
  const float4* const s_bt4 = (const float4*)s_raw + ((bt_off >> 4) + (threadIdx.x >> 6) * 384);
  for (int idx = threadIdx.x; idx < nL * nS; idx += SC_NT) s_fl[idx] = sp.sc_shop_flags[idx];
  for (int idx = threadIdx.x; idx < nL; idx += SC_NT) s_next[idx] = sp.stage_next[idx];
  if (threadIdx.x <= PHX_SHOP_MAX_STOCK) s_tabs[threadIdx.x] = (float)threadIdx.x / (float)PHX_SHOP_MAX_STOCK;
  if ((int)threadIdx.x <= 5 * K) s_tabn[threadIdx.x] = (float)threadIdx.x / (float)sp.fsm_lean_norm;
  const int64_t total = (int64_t)sp.B * nS;
  const int64_t blk = xcd_block(remap != 0);
  const int64_t b_first = pairs ? (blk * SC_NT) / nS : blk * epb;
  const int64_t b_end = (b_first + epb < sp.B) ? b_first + epb : sp.B;
  const int64_t g0 = pairs ? blk * SC_NT : b_first * nS;                     // the block's first pair
  const int64_t g = g0 + threadIdx.x;
  const bool active = pairs ? g < total : (int)threadIdx.x < (int)(b_end - b_first) * nS;
  const int b = active ? (pairs ? (int)(g / nS) : (int)(b_first + threadIdx.x / nS)) : (int)b_first;
  const int s = active ? (pairs ? (int)(g - (int64_t)b * nS) : (int)(threadIdx.x % nS)) : 0;
  int step = fld<int32_t>(sp, F_ENV_STEP)[b];
  uint32_t tick = (uint32_t)fld<int32_t>(sp, F_ENV_TICK)[b];
  int stage = fld<int32_t>(sp, F_ENV_STAGE)[b];
  int prev_stage = fld<int32_t>(sp, F_ENV_PREV_STAGE)[b];
  __syncthreads();          // tables staged; per-env words read before their shop-0 lane rewrites them
  if (!active) return;

  const float norm = (float)sp.fsm_lean_norm;
  const int64_t genv = sp.env_offset + b;
  ShopLane st;
  st.stock = fld<int32_t>(sp, F_SHOP_STOCK)[g]; st.sales = fld<int32_t>(sp, F_SHOP_SALES)[g];
  st.missed = fld<int32_t>(sp, F_SHOP_MISSED)[g]; st.delivered = fld<int32_t>(sp, F_SHOP_DELIVERED)[g];
  double rc = fld<double>(sp, F_ENV_REW_CACHE)[g];                           // self._rewards[aid]
  uint8_t rcv = fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[g];
  float oc[3] = {fld<float>(sp, F_ENV_OBS_CACHE)[g * 3], fld<float>(sp, F_ENV_OBS_CACHE)[g * 3 + 1],
                 fld<float>(sp, F_ENV_OBS_CACHE)[g * 3 + 2]};                // self._observations[aid]
  uint8_t ocv = fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID)[g];
  float lo[3] = {0.f, 0.f, 0.f};
  RngQuadCache rq; rq.q = 0xffffffffu;
  const uint32_t lane_off = (uint32_t)threadIdx.x;                           // g = block base + lane: 32-bit offsets from scalar bases
  const uint32_t wave_off = (uint32_t)threadIdx.x & ~63u;                    // pairs of the block before this wave
  const int lanes_blk = pairs ? (int)((total - g0) < SC_NT ? (total - g0) : SC_NT) : (int)(b_end - b_first) * nS;
  const int n_wave = lanes_blk - (int)wave_off < 64 ? lanes_blk - (int)wave_off : 64;   // active lanes of this wave (a multiple of 4 when wide)
  const int n_pieces = (n_wave * 3) >> 2;
  const int T4 = (batch && n_wave == 64) ? (io.T & ~3) : 0;                  // steps that leave in batches of four (uniform per wave)

  for (int t = 0; t < io.T; ++t) {
    const int64_t row = (int64_t)t * total + g0;                             // uniform over the block
    char* const p_obs = (char*)(io.obs + row * 3);
    char* const p_act = (char*)(io.action_out + row);
    char* const p_rew = (char*)(io.reward + row);
    char* const p_ter = (char*)(io.terminated + row);
    char* const p_tru = (char*)(io.truncated + row);
    const int fl = s_fl[stage * nS + s];
    const bool has_action = (fl & 1) != 0, any_order = (fl & 2) != 0;
    rng_quad_block(rq, sp.seed, genv, tick, s);                              // one Philox block per four ticks
    uint32_t y, aj;
    if (!rng_split(rng_pick(rq.w, tick), y, aj)) y = rng_group_y(sp.seed, genv, tick, s, 0, 1, &aj);   // probability 3.3e-6
    y -= __umul24((uint32_t)((float)y * inv_pK), pK);                        // the first K base-5 digits: the customers' orders
    const int D = any_order ? rng_digit_sum6(y) : 0;
    const float action = rng_j_to_action(aj);
    sc_shop_step(st, has_action, action, any_order, D);
    ++step; ++tick;
    const bool all_trunc = (step == sp.num_steps);                           // env.py:312-318
    float ob[3] = {0.f, 0.f, 0.f};
    uint8_t ov = 0, rv = 0; double rw = 0.0;
    const bool observes = (fl & 8) != 0;
    if (observes) {                                                          // fsm.py:328-332,349
      if ((unsigned)st.stock <= (unsigned)PHX_SHOP_MAX_STOCK && (unsigned)st.sales <= (unsigned)(5 * K) &&
          (unsigned)st.missed <= (unsigned)(5 * K)) { ob[0] = s_tabs[st.stock]; ob[1] = s_tabn[st.sales]; ob[2] = s_tabn[st.missed]; }
      else shop_obs_f32(st.stock, st.sales, st.missed, norm, ob);            // a stock the caller set outside [0, 100]
      oc[0] = ob[0]; oc[1] = ob[1]; oc[2] = ob[2]; ocv = 1;
    }
    if (fl & 16) { rc = shop_reward(st.sales, st.stock); rcv = 1; }          // fsm.py:334-335,350
    if (all_trunc) {                                                         // fsm.py:360-375
      ov = ocv; ob[0] = ocv ? oc[0] : 0.f; ob[1] = ocv ? oc[1] : 0.f; ob[2] = ocv ? oc[2] : 0.f;
      rv = rcv ? 1 : 2; rw = rcv ? rc : 0.0;
    } else if (observes) {                                                   // fsm.py:378
      ov = 1; rv = rcv ? 1 : 2; rw = rcv ? rc : 0.0;
    }
    if (t < T4) {
      const int lane = (int)(threadIdx.x & 63u), r = t & 3;
      float* const to = (float*)(s_bt + r * 768);
      to[lane * 3 + 0] = ob[0]; to[lane * 3 + 1] = ob[1]; to[lane * 3 + 2] = ob[2];
      ((float*)(s_bt + 3072))[r * 64 + lane] = (float)rw;
      ((float*)(s_bt + 4096))[r * 64 + lane] = action;
      unsigned char* const tf = s_bt + 5120 + r * 64 + lane;                 // [plane][step][lane]
      tf[0] = 0; tf[256] = (unsigned char)all_trunc; tf[512] = ov; tf[768] = rv;
      if (r == 3) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // wave-private tile: no barrier
        const int64_t row_b = (int64_t)(t - 3) * total + g0;                 // the batch's first row (uniform over the block)
        const uint32_t utot = (uint32_t)total;                               // (4 total * 12 < 2^32: checked by the launcher)
        char* const b_obs = (char*)(io.obs + row_b * 3) + (size_t)(wave_off * 12u);
#pragma unroll
        for (int k = 0; k < 3; ++k) {                                        // 4 rows x 48 observation pieces
          const uint32_t q = (uint32_t)lane + 64u * (uint32_t)k, rr = q / 48u, pc = q - 48u * rr;
          *(float4*)(b_obs + (size_t)(rr * (utot * 12u) + pc * 16u)) = s_bt4[rr * 48u + pc];
        }
        const uint32_t rr4 = (uint32_t)lane >> 4, pc4 = (uint32_t)lane & 15u;   // 4 rows x 16 pieces of an f32 plane
        *(float4*)((char*)(io.reward + row_b) + (size_t)(wave_off * 4u + rr4 * (utot * 4u) + pc4 * 16u)) = s_bt4[192u + rr4 * 16u + pc4];
        *(float4*)((char*)(io.action_out + row_b) + (size_t)(wave_off * 4u + rr4 * (utot * 4u) + pc4 * 16u)) = s_bt4[256u + rr4 * 16u + pc4];
        const uint32_t pl = (uint32_t)lane >> 4, qf = (uint32_t)lane & 15u, rrf = qf >> 2, pcf = qf & 3u;   // 4 planes x 4 rows x 4 pieces of a u8 plane
        uint8_t* const fb = pl == 0 ? io.terminated : (pl == 1 ? io.truncated : (pl == 2 ? io.obs_valid : io.reward_valid));
        *(float4*)((char*)(fb + row_b) + (size_t)(wave_off + rrf * utot + pcf * 16u)) = s_bt4[320u + pl * 16u + rrf * 4u + pcf];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // read before the next batch overwrites the tile
      }
    } else if (wide) {
      const int lane = (int)(threadIdx.x & 63u);
      s_ot[lane * 3 + 0] = ob[0]; s_ot[lane * 3 + 1] = ob[1]; s_ot[lane * 3 + 2] = ob[2];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // wave-private tile: no barrier
      if (lane < n_pieces)
        *(float4*)(p_obs + (size_t)(wave_off * 12u + (uint32_t)lane * 16u)) = *(const float4*)(s_ot + 4 * lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // read before the next step overwrites the tile
    } else {
      float* po = (float*)(p_obs + (size_t)(lane_off * 12u));
      po[0] = ob[0]; po[1] = ob[1]; po[2] = ob[2];
    }
    if (t >= T4) {
      *(float*)(p_act + (size_t)(lane_off * 4u)) = action;
      *(float*)(p_rew + (size_t)(lane_off * 4u)) = (float)rw;
      *(uint8_t*)(p_ter + (size_t)lane_off) = 0; *(uint8_t*)(p_tru + (size_t)lane_off) = all_trunc;
      if (io.obs_valid) *(uint8_t*)((char*)(io.obs_valid + row) + (size_t)lane_off) = ov;
      if (io.reward_valid) *(uint8_t*)((char*)(io.reward_valid + row) + (size_t)lane_off) = rv;
    }
    lo[0] = ob[0]; lo[1] = ob[1]; lo[2] = ob[2];
    prev_stage = stage; stage = s_next[stage];                               // fsm.py:355
    if (all_trunc) {                                                         // the caller's env.reset(), fsm.py:195-251
      st.stock = 0; step = 0; stage = sp.initial_stage; rcv = 0;
      lo[0] = lo[1] = lo[2] = 0.f;
      if (s_fl[sp.initial_stage * nS + s] & 1) shop_obs_f32(st.stock, st.sales, st.missed, norm, lo);
    }
  }
  fld<int32_t>(sp, F_SHOP_STOCK)[g] = st.stock; fld<int32_t>(sp, F_SHOP_SALES)[g] = st.sales;
  fld<int32_t>(sp, F_SHOP_MISSED)[g] = st.missed; fld<int32_t>(sp, F_SHOP_DELIVERED)[g] = st.delivered;
  fld<double>(sp, F_ENV_REW_CACHE)[g] = rc; fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[g] = rcv;
  for (int d = 0; d < 3; ++d) fld<float>(sp, F_ENV_OBS_CACHE)[g * 3 + d] = oc[d];
  fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID)[g] = ocv;
  if (io.last_obs) for (int d = 0; d < 3; ++d) io.last_obs[g * 3 + d] = lo[d];
  bool writes_env = (s == 0);
  if (pairs) {                               // the env's first lane IN THIS BLOCK counts the block in; the last block to arrive writes
    writes_env = false;
    if (s == 0 || threadIdx.x == 0) {
      const int64_t p0 = (int64_t)b * nS;
      const int n_touch = (int)((p0 + nS - 1) / SC_NT - p0 / SC_NT) + 1;
      int32_t* arrive = fld<int32_t>(sp, F_ENV_ARRIVE) + b;
      if (n_touch == 1 || atomicAdd(arrive, 1) + 1 == n_touch) { writes_env = true; if (n_touch > 1) *arrive = 0; }
    }
  }
  if (writes_env) {
    fld<int32_t>(sp, F_ENV_STEP)[b] = step; fld<int32_t>(sp, F_ENV_TICK)[b] = (int32_t)tick;
    fld<int32_t>(sp, F_ENV_STAGE)[b] = stage; fld<int32_t>(sp, F_ENV_PREV_STAGE)[b] = prev_stage;
  }
}

// ---- launchers ------------------------------------------------------------------------------------
hipError_t phx_launch_sc_step(const DevSpec& sp, const phx_step_io& io, hipStream_t st) {
  // whole envs per block (S <= 256 checked at create).  64-, 128- and 256-thread blocks time the
  // same (the per-launch mode is bound by the host's launch cadence); PHX_STEP_NT overrides.
  // large plain batches: four pairs per thread (AUTO from 2^19 pairs per launch up -- SC64: B = 65 536 9.8 -> 9.6 us, 131 072 19.4 -> 14.3,
  // 262 144 35.9 -> 24.9; PHX_VS_WIDE forces it wherever it applies)
  // (its 16- and 4-byte accesses want the caller's planes aligned like torch / hipMalloc allocations are; anything else: the lane-per-pair kernel)
  const uintptr_t al16 = (uintptr_t)io.obs | (uintptr_t)io.reward | (uintptr_t)io.actions, al4 = (uintptr_t)io.obs_valid | (uintptr_t)io.reward_valid |
                        (uintptr_t)io.done_valid | (uintptr_t)io.terminated | (uintptr_t)io.truncated | (uintptr_t)io.action_valid;
  if (sp.sc_wide_K > 0 && sp.env_type == PHX_ENV_PLAIN && !io.exo && sp.S <= 1024 && sp.sc_tab && (al16 & 15) == 0 && (al4 & 3) == 0 &&
      (sp.variant_step == PHX_VS_WIDE || (sp.variant_step == PHX_VS_AUTO && (int64_t)sp.B * sp.S >= (1 << 19)))) {
    int epb = 1024 / sp.S;
    while (epb > 1 && (epb * sp.S) % 4 != 0) --epb;
    if ((epb * sp.S) % 4 == 0) {
      StepWideArgs a = {};
      a.B = sp.B; a.S = sp.S; a.K = sp.sc_wide_K; a.num_steps = sp.num_steps; a.epb = epb; a.norm = sp.sc_wide_norm;
      a.mS = (uint32_t)((0x100000000ull + (uint64_t)sp.S - 1) / (uint64_t)sp.S);
      a.seed = sp.seed; a.env_offset = sp.env_offset;
      a.stock = (int32_t*)sp.f[F_SHOP_STOCK]; a.sales = (int32_t*)sp.f[F_SHOP_SALES]; a.missed = (int32_t*)sp.f[F_SHOP_MISSED];
      a.delivered = (int32_t*)sp.f[F_SHOP_DELIVERED]; a.env_step = (int32_t*)sp.f[F_ENV_STEP]; a.env_tick = (int32_t*)sp.f[F_ENV_TICK];
      a.sc_tab = sp.sc_tab; a.n_quot = sp.n_quot < 64 ? sp.n_quot : 64;
      a.io = io;
      phx_note_kernel("phx_sc_step_wide_kernel");
      hipLaunchKernelGGL(phx_sc_step_wide_kernel, dim3((unsigned)((sp.B + epb - 1) / epb)), dim3(256), 0, st, a);
      return hipGetLastError();
    }
  }
  int nt = 256;
  const int force_nt = phx_knobs().step_nt;
  if (force_nt == 64 || force_nt == 128 || force_nt == 256) nt = force_nt < sp.S ? 256 : force_nt;
  const int epb = nt / sp.S;
  const int blocks = (sp.B + epb - 1) / epb;
  const int64_t bytes = (int64_t)epb * sp.n_exo + 32;
  const int stage = (io.exo && bytes <= SC_STAGE_MAX) ? 1 : 0;
  const size_t lds = stage ? (size_t)bytes : 0;
  phx_note_kernel("phx_sc_step_kernel");
  if (nt == 64) hipLaunchKernelGGL((phx_sc_step_kernel<64>), dim3(blocks), dim3(64), lds, st, sp, io, epb, stage);
  else if (nt == 128) hipLaunchKernelGGL((phx_sc_step_kernel<128>), dim3(blocks), dim3(128), lds, st, sp, io, epb, stage);
  else hipLaunchKernelGGL((phx_sc_step_kernel<256>), dim3(blocks), dim3(256), lds, st, sp, io, epb, stage);
  return hipGetLastError();
}

// Does the store-wave kernel's FSM instantiation serve this launch?  (Its launch generation is baked into the kernel arguments: a
// capturing stream takes the lane-per-pair loop, like phx_launch_sc_rollout_fsmfast.)
bool phx_fsm_sw_serves(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st) {
  const int vr = sp.variant_rollout;
  if (!sp.fsm_sw.ok || sp.fsm_lean_K <= 0 || sp.env_type != PHX_ENV_FSM || io.actions || io.exo || io.msg_log || io.msg_count || sp.n_samplers != 0 ||
      io.T > 0xFFFF || !sp.f[F_ENV_ARRIVE] || !sp.fsm_irregular) return false;
  // PHX_VR_AUTO: long fragments / fragment lists of batches the time-parallel FSM kernel does not take.  Measured on five boxes, SC256-FSM,
  // B = 8192 (config 3), this kernel / the lane-per-pair loop: T = 400 854-894 us (every box) / 754-985 (bimodal by the buffers' placement);
  // T = 100 268-275 / 226 (ten iterations per pair group of which three fill the pipeline); SC64-FSM, B = 4096: 57 / 37 (time-parallel kernel).
  if (!(vr == PHX_VR_STORE_WAVES || (vr == PHX_VR_AUTO && io.T >= 200 && (int64_t)sp.B * sp.S > 65536))) return false;
  if (!(io.n_frag >= 2) && (!io.obs_valid || !io.reward_valid)) return false;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
  return true;
}

// the STATIC instantiations of phx_sc_rollout_fsm_kernel serve the launch (see the kernel's header)
static bool phx_sc_fsm_static(const DevSpec& sp, const phx_rollout_io& io) {
  return !io.exo && !io.actions && !sp.stage_tab && !sp.any_typed && sp.n_samplers == 0 && sp.sc_all_or_none && sp.n_lists < 65536;
}

hipError_t phx_launch_sc_rollout_fsm(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st, const int32_t* only_if_in, int32_t gen_in) {
  const int epb = SC_NT / sp.S;
  const int remap_env = phx_knobs().rollout_remap;
  const int remap = remap_env >= 0 ? remap_env : 1;
  // store-wave / time-parallel kernel first where a plan applies; the lane-per-pair loop below then runs only if that kernel found
  // an env off the tabulated stage chain (a stage a handler or the caller set) and left the launch alone
  const int32_t* only_if = only_if_in; int32_t gen = gen_in;
  const int vr = sp.variant_rollout;      // phx_spec.variant_rollout: PHX_VR_TIME_PARALLEL / LEAN / GENERAL pick the kernel per env
  if (!only_if && phx_fsm_sw_serves(sp, io, st)) {
    gen = phx_fsm_next_gen(sp);
    const hipError_t fe = phx_launch_sc_rollout_sw(sp, io, st, gen);
    if (fe != hipSuccess) return fe;
    only_if = sp.fsm_irregular;
  }
  if (!only_if && sp.fsm_fast.ok && sp.fsm_lean_K > 0 && (vr == PHX_VR_AUTO || vr == PHX_VR_TIME_PARALLEL)) {
    hipError_t fe = hipSuccess;
    if (phx_launch_sc_rollout_fsmfast(sp, io, st, &fe, &gen)) { if (fe != hipSuccess) return fe; only_if = sp.fsm_irregular; }
  }
  const int lean_env = phx_knobs().fsm_lean;      // development default
  const bool lean = vr == PHX_VR_LEAN || (vr != PHX_VR_GENERAL && lean_env);
  if ((lean || only_if) && sp.fsm_lean_K > 0 && sp.env_type == PHX_ENV_FSM && !io.actions && !io.exo && sp.n_samplers == 0) {
    uint32_t pk = 1; for (int k = 0; k < sp.fsm_lean_K; ++k) pk *= 5u;
    static const float inv[7] = {1.0f, 0.2f, 0.04f, 0.008f, 0.0016f, 0.00032f, 0.000064f};
    // blocks that start on multiples of 4 pairs (16-byte aligned observation rows): whole envs, a multiple of 4 of them
    // unless the shop count is one itself
    int epb_l = epb; int wide = 0;
    const int wide_env = phx_knobs().fsm_wide;
    if (wide_env && ((int64_t)sp.B * sp.S) % 4 == 0) {
      if (sp.S % 4 == 0) wide = (sp.B % epb == 0);
      else if ((SC_NT / sp.S) >= 4) { const int e4 = (SC_NT / sp.S) & ~3; if (sp.B % e4 == 0) { epb_l = e4; wide = 1; } }
    }
    size_t lds = (104 + 32) * 4 + (((size_t)sp.n_lists * sp.S + 3) & ~(size_t)3) + (((size_t)sp.n_lists + 3) & ~(size_t)3) * 4 +
                 (SC_NT / 64) * 192 * 4 + 16;
    phx_note_kernel(only_if ? "phx_sc_rollout_fsm_lean_kernel[if off-chain]" : "phx_sc_rollout_fsm_lean_kernel");
    // blocks of SC_NT consecutive pairs (whole lines per row segment, every lane active) unless the spec asks for whole envs
    // or the pair count is no multiple of 4 (the observation pieces of a wave)
    const int64_t total = (int64_t)sp.B * sp.S;
    const int pairs = (sp.variant_block != PHX_VB_WHOLE_ENVS && total % 4 == 0 && sp.f[F_ENV_ARRIVE]) ? 1 : 0;
    const unsigned grid = pairs ? (unsigned)((total + SC_NT - 1) / SC_NT) : (unsigned)((sp.B + epb_l - 1) / epb_l);
    if (pairs) wide = 1;
    // four steps per store batch (16-byte pieces of every plane): blocks of consecutive pairs whose waves start on 16-pair boundaries,
    // all four u8 planes present, 32-bit offsets within a batch's four rows
    const int batch = (pairs && phx_knobs().fsm_batch && total % 16 == 0 && io.terminated && io.obs_valid && io.reward_valid && io.T >= 4 &&
                       total * 48 < ((int64_t)1 << 32)) ? 1 : 0;
    if (batch) lds += (size_t)(SC_NT / 64) * 6144 + 16;
    hipLaunchKernelGGL(phx_sc_rollout_fsm_lean_kernel, dim3(grid), dim3(SC_NT), lds, st, sp, io, epb_l, remap,
                       pk, inv[sp.fsm_lean_K], wide, only_if, gen, pairs, batch);
    return hipGetLastError();
  }
  phx_note_kernel("phx_sc_rollout_fsm_kernel");
  if (phx_sc_fsm_static(sp, io))
    hipLaunchKernelGGL((phx_sc_rollout_fsm_kernel<false, true>), dim3((sp.B + epb - 1) / epb), dim3(SC_NT),
                       (((size_t)sp.n_lists * sp.S + 15) & ~(size_t)15) + (size_t)sp.n_lists * sizeof(int) + 16, st, sp, io, epb, remap);
  else
    hipLaunchKernelGGL((phx_sc_rollout_fsm_kernel<false, false>), dim3((sp.B + epb - 1) / epb), dim3(SC_NT),
                       (((size_t)sp.n_lists * sp.S + 15) & ~(size_t)15) + 16, st, sp, io, epb, remap);
  return hipGetLastError();
}

// FSM supply chains whose stage handlers are declared as rules (phx_spec.stage_rules): the lane-per-pair loop with the rules evaluated in it
hipError_t phx_launch_sc_rollout_fsm_rules(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st) {
  const int epb = SC_NT / sp.S;
  const int remap_env = phx_knobs().rollout_remap;
  const size_t lds = (((size_t)sp.n_lists * sp.S + 15) & ~(size_t)15) + (size_t)3 * epb * sp.n_rules * sizeof(int) + (size_t)sp.n_lists * sizeof(int) + 8 +
                     (size_t)sp.n_rules * sizeof(DevRule) + 16;
  phx_note_kernel("phx_sc_rollout_fsm_kernel[rules]");
  if (phx_sc_fsm_static(sp, io))
    hipLaunchKernelGGL((phx_sc_rollout_fsm_kernel<true, true>), dim3((sp.B + epb - 1) / epb), dim3(SC_NT), lds, st, sp, io, epb, remap_env >= 0 ? remap_env : 1);
  else
    hipLaunchKernelGGL((phx_sc_rollout_fsm_kernel<true, false>), dim3((sp.B + epb - 1) / epb), dim3(SC_NT), lds, st, sp, io, epb, remap_env >= 0 ? remap_env : 1);
  return hipGetLastError();
}

hipError_t phx_launch_sc_rollout(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st, const int32_t* only_if, int32_t gen) {
  // ~32..64 pairs per block (one wave in the sequential phase), TC steps so that the two item
  // tiles stay around 34 KB
  RollArgs a;
  a.only_if = only_if; a.gen = gen;
  a.B = sp.B; a.S = sp.S; a.n_exo = sp.n_exo; a.num_steps = sp.num_steps; a.T = io.T;
  a.seed = sp.seed; a.env_offset = sp.env_offset;
  a.shop_norm = sp.shop_norm; a.shop_cust_ptr = sp.shop_cust_ptr; a.shop_cust_exo = sp.shop_cust_exo;
  a.sc_tab = sp.sc_tab; a.n_tabn = sp.n_tabn; a.n_quot = sp.n_quot;
  a.timing = nullptr;
#ifdef PHX_TIMING
  { static unsigned long long* tbuf = nullptr; if (!tbuf) (void)hipMalloc((void**)&tbuf, 8 * 8 * 8192 * sizeof(unsigned long long)); a.timing = tbuf;
    if (getenv("PHX_TIMING_DUMP")) { static int calls = 0; if (++calls == 20) { (void)hipDeviceSynchronize(); std::vector<unsigned long long> h(8 * 8 * 8192); (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
      const int nw = ((sp.B + 7) / 8) * 8; double sum[8] = {0}; double w0[8] = {0}; for (int w = 0; w < nw; ++w) for (int q = 0; q < 8; ++q) { sum[q] += h[(size_t)w * 8 + q]; if (w % 8 == 0) w0[q] += h[(size_t)w * 8 + q]; }
      fprintf(stderr, "PHX_TIMING avg cycles per wave: setup %.0f | P1 %.0f | bar %.0f | P2 %.0f | bar %.0f | P3a %.0f | P3b(+bar) %.0f | bar %.0f\n", sum[0]/nw, sum[1]/nw, sum[2]/nw, sum[3]/nw, sum[4]/nw, sum[5]/nw, sum[6]/nw, sum[7]/nw);
      fprintf(stderr, "PHX_TIMING wave0 of each block:      setup %.0f | P1 %.0f | bar %.0f | P2 %.0f | bar %.0f | P3a %.0f | P3b(+bar) %.0f | bar %.0f\n", w0[0]*8/nw, w0[1]*8/nw, w0[2]*8/nw, w0[3]*8/nw, w0[4]*8/nw, w0[5]*8/nw, w0[6]*8/nw, w0[7]*8/nw); } } }
#endif
  a.stock = (int32_t*)sp.f[F_SHOP_STOCK]; a.sales = (int32_t*)sp.f[F_SHOP_SALES];
  a.missed = (int32_t*)sp.f[F_SHOP_MISSED]; a.delivered = (int32_t*)sp.f[F_SHOP_DELIVERED];
  a.env_step = (int32_t*)sp.f[F_ENV_STEP]; a.env_tick = (int32_t*)sp.f[F_ENV_TICK];
  a.io = io;
  // whole envs per block: a multiple of 4 so that G = epb * S makes every tile row a 16-byte
  // multiple; B = 4096, S = 9 -> epb 4, G 36, 1024 blocks of 256 threads
  int epb = 0;
  const int force_epb = phx_knobs().rollout_epb;
  if (force_epb > 0 && force_epb * sp.S <= 256 && force_epb <= sp.B) epb = force_epb;
  // measured best on SC64 and SC256 up to ~100 k pairs: 256-thread blocks owning ~32..64 pairs;
  // beyond that (the chip is full either way) 512-thread blocks with twice the pairs win by ~5 %
  const bool big = (int64_t)sp.B * sp.S >= 131072 && 8 * sp.S <= 256;
  if (!epb && big) epb = 8;
  if (!epb) for (int cand = 4; cand * sp.S <= 256 && cand <= sp.B; cand += 4) if (cand * sp.S >= 32) { epb = cand; break; }
  if (!epb) { epb = 256 / sp.S; if (epb > 4) epb &= ~3; if (epb < 1) epb = 1; if (epb > sp.B) epb = sp.B; }
  const int G = epb * sp.S;
  auto magic = [](int d) { return (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)(d > 0 ? d : 1)); };
  const int ldskb_env = phx_knobs().rollout_ldskb;
  const int ldskb = ldskb_env ? ldskb_env : (big ? 44 : 30);
  int TC = (ldskb * 1024) / (G * 32); if (TC < 1) TC = 1; if (TC > io.T) TC = io.T;   // 32 B of LDS per item (double-buffered tiles)
  while ((int64_t)TC * G * 3 >= 65536 && TC > 1) --TC;          // magic division range
  if (sp.num_steps >= 1 && TC > sp.num_steps) TC = sp.num_steps; // at most one episode end per chunk
  if (TC >= 8) {
    // Both parallel phases have TC * G / 4 work items per chunk (draws: one per row quad and pair,
    // outputs: one per row and 4 pairs).  Among the multiples of 4 that fit, take the chunk length
    // that fills whole waves best (a mostly empty last pass costs as much as a full one).
    int best = TC & ~3; double best_fill = 0.0;
    for (int cand = TC & ~3; cand >= 8 && cand * 2 > TC; cand -= 4) {
      const int work = cand * G / 4;
      const double fill = (double)work / (double)(((work + 63) / 64) * 64);
      if (fill > best_fill + 0.02) { best_fill = fill; best = cand; }
    }
    TC = best;
  }
  a.mF = magic(G / 4);
  const int items = TC * G;
  const size_t lds = (size_t)((items + 3) & ~3) * 4 * 3 * 2 + (size_t)((items + 3) & ~3) * 4 * 2 +
                     (size_t)((G + 3) & ~3) * 4 + (size_t)((epb + 3) & ~3) * 12 + (size_t)((sp.S + 4) & ~3) * 4 +
                     (size_t)((sp.S + 3) & ~3) * 4 + 128 + (size_t)(101 + sp.n_tabn + 202) * 4 + 64;
  a.epb = epb; a.TC = TC;
  // XCD-aware workgroup -> env mapping (xcd_block).  Measured, SC64, T = 100, back-to-back launches: HBM write
  // traffic 90.8 -> 80.8 MB per launch at B = 4096 (the algorithmic 82.3 MB); +1 % at B = 4096, +5 % at 16384,
  // +8 % at 65536.  PHX_ROLLOUT_REMAP = 0 identity, 1 contiguous eighths, n > 1 locality groups of n workgroups.
  const int remap_env = phx_knobs().rollout_remap;
  a.xcd_remap = remap_env >= 0 ? remap_env : 1;
  const dim3 grid((sp.B + epb - 1) / epb);
  const bool replay = io.actions != nullptr || io.exo != nullptr;
  const int64_t total = (int64_t)sp.B * sp.S;
  const bool wide = (sp.B % epb == 0) && (G % 4 == 0) && (total % 4 == 0);
  const int nt_env = phx_knobs().rollout_nt;
  // block size: the waves that hold a recurrence lane + enough waves to take phase 1 (one work item
  // per row quad and pair) in a single pass, when that fits 512 threads
  int nt = 256;
  {
    const int p2w = (G + 63) / 64, p1w = (((TC + 3) >> 2) * G + 63) / 64;      // quad-aligned ticks: one pass
    const int want = 64 * (p2w + p1w);
    if (want <= 256) nt = 256; else if (want <= 320) nt = 320; else if (want <= 384) nt = 384; else if (want <= 512) nt = 512;
    else nt = big ? 512 : 256;
  }
  if (nt_env) nt = nt_env;
  phx_note_kernel(only_if ? "phx_sc_rollout_kernel[if an action rounds below zero]" : "phx_sc_rollout_kernel");
#define PHX_LAUNCH_ROLLOUT(NT_)                                                                              \
  do {                                                                                                        \
    if (wide && !replay) hipLaunchKernelGGL((phx_sc_rollout_kernel<NT_, false, true>), grid, dim3(NT_), lds, st, a);  \
    else if (wide) hipLaunchKernelGGL((phx_sc_rollout_kernel<NT_, true, true>), grid, dim3(NT_), lds, st, a);         \
    else if (!replay) hipLaunchKernelGGL((phx_sc_rollout_kernel<NT_, false, false>), grid, dim3(NT_), lds, st, a);    \
    else hipLaunchKernelGGL((phx_sc_rollout_kernel<NT_, true, false>), grid, dim3(NT_), lds, st, a);                  \
  } while (0)
  if (nt == 1024) PHX_LAUNCH_ROLLOUT(1024); else if (nt == 768) PHX_LAUNCH_ROLLOUT(768); else if (nt == 384) PHX_LAUNCH_ROLLOUT(384);
  else if (nt == 320) PHX_LAUNCH_ROLLOUT(320);
  else if (nt == 512) PHX_LAUNCH_ROLLOUT(512); else PHX_LAUNCH_ROLLOUT(256);
#undef PHX_LAUNCH_ROLLOUT
  return hipGetLastError();
}

// phx_stk_fused.hip -- fused static-schedule step for the Stackelberg market (SELLER / BUYER
// kinds on StackelbergEnv semantics, phantom/stackelberg.py:111-196).
//
// The schedule is static: a step has one non-empty round.  Leaders' step: every acting seller
// posts Price(price) along its CSR row; each buyer stores the price in the slot of that
// neighbour.  Followers' step: every buying buyer sends Order(1) to its cheapest neighbour
// (first minimum in neighbour order); a seller books revenue += price * vol once per order --
// all addends of a round are the same f64, so the sequential sum only needs the ORDER COUNT,
// which the block gets from LDS atomics.  One workgroup per env instance; the 8-byte price
// slots of the buyers are kept in compressed per-seller form (below).  Results are bit-identical to
// the generic engine.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "phx_dev.h"

#define STK_NT 256
#define STKR_SLOTS 3
#ifndef STKR_W384
#define STKR_W384 6
#endif
#define STKR_MINWAVES(NT) ((NT) == 384 ? STKR_W384 : ((NT) / 64 < 4 ? 4 : ((NT) / 64 > 8 ? 8 : (NT) / 64)))
__host__ __device__ inline size_t g_stk_paid_off(int nSell) { return ((size_t)nSell * (8 + 8 + 8 + 4 + 4 + 1) + 15) & ~(size_t)15; }
#ifdef PHX_TIMING
__device__ unsigned long long g_stk_tm[8];
__device__ unsigned long long g_stk_rt[8];      // wall clock (100 MHz) of the last rollout launch: [0] min entry, [1] max exit, [2..4] sums of setup / loop / epilogue, [5] blocks, [6] sum of (entry - min entry)
#define STICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); stm[k] += now_ - sprev; sprev = now_; } while (0)
#else
#define STICK(k) do {} while (0)
#endif

// SellerAgent.handle_order, n times in one round (stackelberg test agents: self.revenue += self.price * vol, one Order(1) per buying
// neighbour): the sequential f64 sum of n EQUAL addends.  From a revenue of +0 (the followers' step resets it before the round) and for an
// addend that is a float32 value (the price is the seller's f32 action) every partial sum k * amount has at most 24 + 11 significant bits:
// each addition is exact, and so is ONE multiplication -- the popular sellers of a 128 x 1024 market book ~100 orders, and the whole
// workgroup waits at the barrier behind that chain (PHX_TIMING: 1.8 k of a step's 8.7 k cycles).  Anything else keeps the loop.
__device__ __forceinline__ double stk_book(double rev, double amount, int n) {
  const bool exact = __double_as_longlong(rev) == 0ll && amount != 0.0 && (double)(float)amount == amount && n <= 2048;
  if (exact) return __dmul_rn((double)n, amount);
  for (int k = 0; k < n; ++k) rev = __dadd_rn(rev, amount);
  return rev;
}

// BuyerAgent.prices in compressed form.  Every Price a seller posts goes to ALL of its neighbours
// in the same round (decode_action returns one message per ctx.neighbour_ids entry) and the
// topology is static, so the slot a buyer keeps for neighbour l always holds "the last price l
// posted since the reset, else 1.0": one f64 per SELLER (seller.posted, 1 KB per env at 128
// sellers) instead of one per (buyer, neighbour) (64 KB per env at 1024 x 8).  The kernel keeps
// the per-seller array in LDS; buyers gather from it through the slot-major neighbour table
// stk_nbr.  phx_sync_fields / the first host-injected message materialise buyer.prices.
template <bool DYN>
__global__ __launch_bounds__(STK_NT) void phx_stk_step_kernel(const DevSpec sp, const phx_step_io io) {
  // Static per-agent record stk_rec[a] = kind | deg << 8 | kind_rank << 16 and per-list flag byte
  // stk_flags[list][a] (1 acts, 2 observes, 4 rewarded) replace the kind / kind_rank / row_ptr /
  // mask lookups: one dependent level between the record and the agent's state instead of four.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int A = sp.A;                                    // every agent of this schedule is strategic: s == a
  const int nSell = sp.kind_count[PHX_KIND_SELLER], nBuy = sp.kind_count[PHX_KIND_BUYER];
  double* s_posted = (double*)smem;                      // [nSell] price every neighbour currently holds
  double* s_price = s_posted + nSell;                    // [nSell] seller.price
  double* s_rev = s_price + nSell;                       // [nSell] seller.revenue
  int* s_tx = (int*)(s_rev + nSell);                     // [nSell] seller.tx
  int* s_count = s_tx + nSell;                           // [nSell] orders received this step
  uint8_t* s_sent = (uint8_t*)(s_count + nSell);         // [nSell] seller broadcast a Price this step
  double* s_paid = (double*)(smem + g_stk_paid_off(nSell));   // [nBuy] buyer.paid as decided this step
  uint8_t* s_bought = (uint8_t*)(s_paid + nBuy);         // [nBuy] buyer.bought this step, 0xFF: the buyer did not act (state in the blob)

  const int t = fld<int32_t>(sp, F_ENV_STEP)[b] + 1;                         // env.py:252
  const uint32_t tick = (uint32_t)fld<int32_t>(sp, F_ENV_TICK)[b];
  const int list = (t & 1) ? 0 : 1;                                          // stackelberg.py:133-137
  const uint8_t* flags = sp.stk_flags + (int64_t)list * A;
  const float* actions_b = io.actions ? io.actions + (int64_t)b * A : nullptr;
  const uint8_t* av_b = io.action_valid ? io.action_valid + (int64_t)b * A : nullptr;
  const int64_t sbase = (int64_t)b * nSell, bbase = (int64_t)b * nBuy;
  double* posted_b = fld<double>(sp, F_SELLER_POSTED) + sbase;
  // StochasticNetwork (network.py:340-453): the env's surviving connections; a slot whose connection is
  // off is not a neighbour this episode (sellers do not post to it, buyers do not consider it)
  const uint8_t* conn_b = nullptr;
  if (DYN) {                                             // the env's connectivity row, staged in LDS
    uint8_t* s_conn = s_bought + ((nBuy + 15) & ~15);
    const uint8_t* src = fld<uint8_t>(sp, F_NET_CONN_ON) + (int64_t)b * sp.n_conn;
    for (int i = tid; i < sp.n_conn; i += STK_NT) s_conn[i] = src[i];
    conn_b = s_conn;
  }

  for (int k = tid; k < nSell; k += STK_NT) {
    s_posted[k] = posted_b[k]; s_price[k] = fld<double>(sp, F_SELLER_PRICE)[sbase + k];
    s_rev[k] = fld<double>(sp, F_SELLER_REVENUE)[sbase + k]; s_tx[k] = fld<int32_t>(sp, F_SELLER_TX)[sbase + k];
    s_count[k] = 0; s_sent[k] = 0;
  }
  for (int k = tid; k < nBuy; k += STK_NT) s_bought[k] = 0xFF;
#ifdef PHX_TIMING
  unsigned long long stm[8] = {0}, sprev = __builtin_readcyclecounter();
#endif
  __syncthreads();
  STICK(0);

  // ---- acting phase (_handle_acting_agents, env.py:320-336): decode_action of every acting agent.
  //      Sellers only record the new price (the Price messages land after the acting phase:
  //      buyers acting in the same step still see the old slots).
  for (int a = tid; a < A; a += STK_NT) {
    if (!(flags[a] & 1)) continue;
    if (!(actions_b && (!av_b || av_b[a]))) continue;                        // aid in actions
    const float action = actions_b[a];
    const uint32_t rec = sp.stk_rec[a];
    const int kr = (int)(rec >> 16), deg = (int)((rec >> 8) & 255u);
    if ((rec & 255u) == PHX_KIND_SELLER) {
      s_price[kr] = (double)action; s_sent[kr] = 1;
    } else {                                                                 // BUYER
      int bought = 0; double paid = 0.0;
      if (action > 0.5f && deg > 0) {
        const uint16_t* nb = sp.stk_nbr + kr;
        int jr = -1; double best = 0.0;                                      // first minimum in (current) neighbour order
        if (!DYN) {
          jr = nb[0]; best = s_posted[jr];
          for (int k = 1; k < deg; ++k) { const int l = nb[(int64_t)k * nBuy]; const double v = s_posted[l]; if (v < best) { best = v; jr = l; } }
        } else {
          for (int k = 0; k < deg; ++k) {
            if (!conn_b[sp.stk_nbr_conn[(int64_t)k * nBuy + kr]]) continue;
            const int l = nb[(int64_t)k * nBuy]; const double v = s_posted[l];
            if (jr < 0 || v < best) { best = v; jr = l; }
          }
        }
        if (jr >= 0) {
          bought = 1; paid = best;
          atomicAdd(&s_count[jr], 1);                                        // Order(1) -> that seller's inbox
        }
      }
      fld<int32_t>(sp, F_BUYER_BOUGHT)[bbase + kr] = bought;
      fld<double>(sp, F_BUYER_PAID)[bbase + kr] = paid;
      s_bought[kr] = (uint8_t)bought; s_paid[kr] = paid;                     // read back by compute_reward below
    }
  }
  STICK(1);
  __syncthreads();
  STICK(2);
  // ---- pre_message_resolution + the single round: sellers book their orders (one add per Order,
  //      in inbox order: all addends are the same f64); this step's Price messages land in every
  //      neighbour's slot, i.e. the seller's posted price changes -----------------------------------
  for (int kr = tid; kr < nSell; kr += STK_NT) {
    double rev = s_rev[kr]; int tx = s_tx[kr];
    if ((t & 1) == 0) { rev = 0.0; tx = 0; }                                 // start of a buying round
    const int n = s_count[kr];
    if (n > 0) {
      const double amount = __dmul_rn(s_price[kr], 1.0);                     // price * vol, vol = 1
      rev = stk_book(rev, amount, n);
      tx += n;
    }
    s_rev[kr] = rev; s_tx[kr] = tx;
    fld<double>(sp, F_SELLER_REVENUE)[sbase + kr] = rev;
    fld<int32_t>(sp, F_SELLER_TX)[sbase + kr] = tx;
    if (s_sent[kr]) { fld<double>(sp, F_SELLER_PRICE)[sbase + kr] = s_price[kr]; s_posted[kr] = s_price[kr]; posted_b[kr] = s_price[kr]; }
  }
  STICK(3);
  __syncthreads();
  STICK(4);
  // ---- obs / reward / done in ONE pass (stackelberg.py:142-196).  Neither kind terminates or
  //      truncates (agents.py:292-323), so "terminal" is the step count alone (env.py:312-318).
  const bool terminal = (t == sp.num_steps);
  double* rew_cache = fld<double>(sp, F_ENV_REW_CACHE) + (int64_t)b * A;
  uint8_t* rew_cache_v = fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID) + (int64_t)b * A;
  for (int a = tid; a < A; a += STK_NT) {
    const int64_t o = (int64_t)b * A + a;
    const int fl = flags[a];
    const uint32_t rec = sp.stk_rec[a];
    const int kr = (int)(rec >> 16), deg = (int)((rec >> 8) & 255u);
    const bool seller = (rec & 255u) == PHX_KIND_SELLER;
    float ob0 = 0.f, ob1 = 0.f; uint8_t ov = 0;
    if (fl & 2) {                                                            // encode_observation
      ov = 1;
      if (seller) {
        int sd = sp.row_ptr[a + 1] - sp.row_ptr[a];                          // len(ctx.neighbour_ids)
        if (DYN) { sd = 0; for (int e = sp.row_ptr[a]; e < sp.row_ptr[a + 1]; ++e) sd += conn_b[sp.col_conn[e]] ? 1 : 0; }
        ob0 = sd ? (float)((double)s_tx[kr] / (double)sd) : 0.f; ob1 = (float)s_price[kr];
      } else {
        const uint16_t* nb = sp.stk_nbr + kr;
        double mn = 1.0;                                                     // min over the price slots (none: 1.0)
        if (!DYN) {
          if (deg > 0) mn = s_posted[nb[0]];
          for (int k = 1; k < deg; ++k) { const double v = s_posted[nb[(int64_t)k * nBuy]]; mn = v < mn ? v : mn; }
        } else {
          bool any = false;
          for (int k = 0; k < deg; ++k) {
            if (!conn_b[sp.stk_nbr_conn[(int64_t)k * nBuy + kr]]) continue;
            const double v = s_posted[nb[(int64_t)k * nBuy]];
            if (!any || v < mn) { mn = v; any = true; }
          }
        }
        ob0 = (float)mn; ob1 = (float)sp.param_f[a * PHX_NPF];
      }
    }
    // self._rewards[aid]: recomputed below for the rewarded group; read only where it is emitted as it stands
    uint8_t cv = 0; double cache = 0.0;
    if (fl & 4) {                                                            // compute_reward -> self._rewards
      if (seller) cache = s_rev[kr];
      else {
        int bought = s_bought[kr]; double paid = s_paid[kr];
        if (bought == 0xFF) { bought = fld<int32_t>(sp, F_BUYER_BOUGHT)[bbase + kr]; paid = fld<double>(sp, F_BUYER_PAID)[bbase + kr]; }
        cache = bought ? __dsub_rn(sp.param_f[a * PHX_NPF], paid) : 0.0;
      }
      cv = 1; rew_cache[a] = cache; rew_cache_v[a] = 1;
    } else if (terminal || ov) { cv = rew_cache_v[a]; cache = rew_cache[a]; }
    uint8_t rv = 0; double rw = 0.0;
    if (terminal) { rv = cv ? 1 : 2; rw = cv ? cache : 0.0; }                // stackelberg.py:180-187
    else if (ov && cv) { rv = 1; rw = cache; }                               // stackelberg.py:190-194
    *(float2*)(io.obs + o * 2) = make_float2(ob0, ob1);
    io.reward[o] = rw;
    io.obs_valid[o] = ov; io.reward_valid[o] = rv; io.done_valid[o] = 1;
    io.terminated[o] = 0; io.truncated[o] = 0;
  }
  STICK(5);
  if (tid == 0) {
    fld<int32_t>(sp, F_ENV_STEP)[b] = t;
    fld<int32_t>(sp, F_ENV_TICK)[b] = (int32_t)(tick + 1);
    io.all_terminated[b] = 0; io.all_truncated[b] = terminal;
  }
#ifdef PHX_TIMING
  if (blockIdx.x < 64 && (tid & 63) == 0 && (tid >> 6) == 1) for (int q = 0; q < 8; ++q) atomicAdd(&g_stk_tm[q], stm[q]);
#endif
}

// ---- the same step, loads batched.  Measured on the kernel above (PHX_TIMING, 128 x 1024, B = 4096): 34 k cycles per
// block, 13 k in the acting pass and 19 k in the output pass -- per agent three to four DEPENDENT table lookups (flags ->
// record -> neighbour slots -> action / cache), 4.5 agents per lane one after the other; removing the output stores
// (half of the kernel's bytes) saved 9 of 57 us.  Here a lane owns (up to) STKR_SLOTS agents, a = tid + k * NT, and the
// step is two load batches: (1) the env's step word, the sellers' state and, per slot, the packed record, the packed
// neighbour list and the agent's value -- none depends on another; (2) once the step parity is known: the action of
// the agents that act and the reward cache of those that emit it without recomputing it.  Barriers order LDS only, so
// batch (2) and the state stores stay in flight across them.  Static graphs with <= 8 neighbours per buyer
// (DevSpec::stk_packed); the kernel above takes the rest.
__device__ __forceinline__ void stk_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int NT>
__global__ __launch_bounds__(NT, STKR_MINWAVES(NT)) void phx_stk_step_fast_kernel(const DevSpec sp, const phx_step_io io) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int A = sp.A;
  const int nSell = sp.kind_count[PHX_KIND_SELLER], nBuy = sp.kind_count[PHX_KIND_BUYER];
  double* s_posted = (double*)smem;                      // [nSell] price every neighbour currently holds
  double* s_price = s_posted + nSell;                    // [nSell] seller.price
  double* s_rev = s_price + nSell;                       // [nSell] seller.revenue
  int* s_tx = (int*)(s_rev + nSell);                     // [nSell] seller.tx
  int* s_count = s_tx + nSell;                           // [nSell] orders received this step
  uint8_t* s_sent = (uint8_t*)(s_count + nSell);         // [nSell] seller broadcast a Price this step
  double* s_paid = (double*)(smem + g_stk_paid_off(nSell));   // [nBuy] buyer.paid as decided this step
  uint8_t* s_bought = (uint8_t*)(s_paid + nBuy);         // [nBuy] buyer.bought this step, 0xFF: the buyer did not act
  const int64_t sbase = (int64_t)b * nSell, bbase = (int64_t)b * nBuy, abase = (int64_t)b * A;

  // ---- batch 1 ---------------------------------------------------------------------------------------------
  const int t = fld<int32_t>(sp, F_ENV_STEP)[b] + 1;                         // env.py:252
  const uint32_t tick = (uint32_t)fld<int32_t>(sp, F_ENV_TICK)[b];
  uint32_t rec[STKR_SLOTS]; uint4 ag[STKR_SLOTS]; double val[STKR_SLOTS];
#pragma unroll
  for (int k = 0; k < STKR_SLOTS; ++k) {
    const int a = tid + k * NT;
    rec[k] = 0; ag[k] = make_uint4(0u, 0u, 0u, 0u); val[k] = 0.0;
    if (a < A) { rec[k] = sp.stk_rec2[a]; ag[k] = *(const uint4*)(sp.stk_agent + 4 * a); val[k] = sp.param_f[a * PHX_NPF]; }
  }
  double* posted_b = fld<double>(sp, F_SELLER_POSTED) + sbase;
  for (int k = tid; k < nSell; k += NT) {
    s_posted[k] = posted_b[k]; s_price[k] = fld<double>(sp, F_SELLER_PRICE)[sbase + k];
    s_rev[k] = fld<double>(sp, F_SELLER_REVENUE)[sbase + k]; s_tx[k] = fld<int32_t>(sp, F_SELLER_TX)[sbase + k];
    s_count[k] = 0; s_sent[k] = 0;
  }
  for (int k = tid; k < nBuy; k += NT) s_bought[k] = 0xFF;
  // ---- batch 2: what depends on the step's parity (stackelberg.py:133-137: leaders on odd steps) ---------------
  const int lsh = (t & 1) ? 1 : 4;
  const bool terminal = (t == sp.num_steps);
  const float* actions_b = io.actions ? io.actions + abase : nullptr;
  const uint8_t* av_b = io.action_valid ? io.action_valid + abase : nullptr;
  const double* rew_cache = fld<double>(sp, F_ENV_REW_CACHE) + abase;
  const uint8_t* rew_cache_v = fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID) + abase;
  float action[STKR_SLOTS]; uint8_t has[STKR_SLOTS], cv[STKR_SLOTS]; double cache[STKR_SLOTS];
#pragma unroll
  for (int k = 0; k < STKR_SLOTS; ++k) {
    const int a = tid + k * NT;
    const uint32_t fl = rec[k] >> lsh;                                       // 1 acts, 2 observes, 4 rewarded
    action[k] = 0.f; has[k] = 0; cv[k] = 0; cache[k] = 0.0;
    if (a < A) {
      if ((fl & 1u) && actions_b) { action[k] = actions_b[a]; has[k] = av_b ? av_b[a] : 1; }   // aid in actions, env.py:330
      // self._rewards[aid]: recomputed for the rewarded group, read only where it is emitted as it stands
      if (!(fl & 4u) && (terminal || (fl & 2u))) { cv[k] = rew_cache_v[a]; cache[k] = rew_cache[a]; }
    }
  }
  stk_lds_barrier();

  auto cheapest = [&](int k, int deg, int& jr) __attribute__((always_inline)) {   // first minimum in neighbour order
    const uint32_t w[4] = {ag[k].x, ag[k].y, ag[k].z, ag[k].w};
    double best = 0.0; jr = -1;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < deg) {
        const int l = (int)((w[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
        const double v = s_posted[l];
        if (j == 0 || v < best) { best = v; jr = l; }
      }
    return best;
  };
  // ---- acting phase (_handle_acting_agents, env.py:320-336) ----------------------------------------------------
#pragma unroll
  for (int k = 0; k < STKR_SLOTS; ++k) {
    const int a = tid + k * NT;
    if (a >= A || !has[k]) continue;
    const int kr = (int)(rec[k] >> 16), deg = (int)((rec[k] >> 8) & 255u);
    if (rec[k] & 1u) { s_price[kr] = (double)action[k]; s_sent[kr] = 1; }
    else {
      int bought = 0; double paid = 0.0;
      if (action[k] > 0.5f && deg > 0) {
        int jr;
        const double best = cheapest(k, deg, jr);
        if (jr >= 0) { bought = 1; paid = best; atomicAdd(&s_count[jr], 1); }     // Order(1) -> that seller's inbox
      }
      fld<int32_t>(sp, F_BUYER_BOUGHT)[bbase + kr] = bought;
      fld<double>(sp, F_BUYER_PAID)[bbase + kr] = paid;
      s_bought[kr] = (uint8_t)bought; s_paid[kr] = paid;
    }
  }
  stk_lds_barrier();
  // ---- pre_message_resolution + the single round ------------------------------------------------------------------
  for (int kr = tid; kr < nSell; kr += NT) {
    double rev = s_rev[kr]; int tx = s_tx[kr];
    if ((t & 1) == 0) { rev = 0.0; tx = 0; }                                 // start of a buying round
    const int n = s_count[kr];
    if (n > 0) {
      const double amount = __dmul_rn(s_price[kr], 1.0);                     // price * vol, vol = 1
      rev = stk_book(rev, amount, n);
      tx += n;
    }
    s_rev[kr] = rev; s_tx[kr] = tx;
    fld<double>(sp, F_SELLER_REVENUE)[sbase + kr] = rev;
    fld<int32_t>(sp, F_SELLER_TX)[sbase + kr] = tx;
    if (s_sent[kr]) { fld<double>(sp, F_SELLER_PRICE)[sbase + kr] = s_price[kr]; s_posted[kr] = s_price[kr]; posted_b[kr] = s_price[kr]; }
  }
  stk_lds_barrier();
  // ---- obs / reward / done in one pass (stackelberg.py:142-196) -----------------------------------------------------
  char* const p_obs = (char*)(io.obs + abase * 2);
  char* const p_rew = (char*)(io.reward + abase);
  char* const p_cache = (char*)(fld<double>(sp, F_ENV_REW_CACHE) + abase);
#pragma unroll
  for (int k = 0; k < STKR_SLOTS; ++k) {
    const int a = tid + k * NT;
    if (a >= A) continue;
    const uint32_t fl = rec[k] >> lsh;
    const int kr = (int)(rec[k] >> 16), deg = (int)((rec[k] >> 8) & 255u);
    const bool seller = (rec[k] & 1u) != 0;
    float ob0 = 0.f, ob1 = 0.f; uint8_t ov = 0;
    if (fl & 2u) {                                                           // encode_observation
      ov = 1;
      if (seller) {
        const int sd = (int)ag[k].x;                                         // len(ctx.neighbour_ids)
        ob0 = sd ? (float)((double)s_tx[kr] / (double)sd) : 0.f; ob1 = (float)s_price[kr];
      } else {
        int jr;
        const double mn = cheapest(k, deg, jr);                              // min over the price slots (none: 1.0)
        ob0 = (float)(jr >= 0 ? mn : 1.0); ob1 = (float)val[k];
      }
    }
    uint8_t cvk = cv[k]; double ck = cache[k];
    if (fl & 4u) {                                                           // compute_reward -> self._rewards
      if (seller) ck = s_rev[kr];
      else {
        int bought = s_bought[kr]; double paid = s_paid[kr];
        if (bought == 0xFF) { bought = fld<int32_t>(sp, F_BUYER_BOUGHT)[bbase + kr]; paid = fld<double>(sp, F_BUYER_PAID)[bbase + kr]; }
        ck = bought ? __dsub_rn(val[k], paid) : 0.0;
      }
      cvk = 1;
      *(double*)(p_cache + (size_t)((uint32_t)a * 8u)) = ck;
      fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[abase + a] = 1;
    }
    uint8_t rv = 0; double rw = 0.0;
    if (terminal) { rv = cvk ? 1 : 2; rw = cvk ? ck : 0.0; }                 // stackelberg.py:180-187
    else if (ov && cvk) { rv = 1; rw = ck; }                                 // stackelberg.py:190-194
    const uint32_t ua = (uint32_t)a;
    *(float2*)(p_obs + (size_t)(ua * 8u)) = make_float2(ob0, ob1);
    *(double*)(p_rew + (size_t)(ua * 8u)) = rw;
    const int64_t o = abase + a;
    io.obs_valid[o] = ov; io.reward_valid[o] = rv; io.done_valid[o] = 1;
    io.terminated[o] = 0; io.truncated[o] = 0;
  }
  if (tid == 0) {
    fld<int32_t>(sp, F_ENV_STEP)[b] = t;
    fld<int32_t>(sp, F_ENV_TICK)[b] = (int32_t)(tick + 1);
    io.all_terminated[b] = 0; io.all_truncated[b] = terminal;
  }
}

// ---- T fused steps of the market (phx_rollout): the env's whole mutable state -- per seller
// posted / price / revenue / tx, per buyer bought / paid, the per-agent reward cache -- lives in
// LDS for the fragment; HBM sees the trajectory rows only.  Random policy (actions == NULL): the
// acting agent's word of the tick (the stream the supply chain uses; its rank j in [0, 274877))
// mapped onto its action space: seller price j / 274877, buyer buys iff j < 137438 (p = 1/2).
// Auto-reset at the end of the terminal step (the caller's env.reset(), stackelberg.py:53-109).
//
// A lane owns the same (up to) STKR_SLOTS agents for the whole fragment, a = tid + k * NT, with NT chosen so that
// the slots are full (1 152 agents: 384 threads x 3): what is static per agent -- its record, both flag bytes, a
// buyer's neighbour list (<= 8 seller ranks, packed u16) and value, a seller's degree -- is loaded into registers
// once, and the agent's Philox block is kept for the four ticks it covers (an agent acts on two of them).  Measured
// before (PHX_TIMING, 512 threads, everything looked up per step): 15.3 k cycles per step = acting 6.3 k (Philox per
// acting agent and step) + booking 1.8 k + outputs 7.0 k (four dependent table loads per agent), a quarter of the
// lanes idle in the third pass over the agents.
//
// FAST (round 6): 1 = static graph, every buyer with at most eight neighbours (DevSpec::stk_packed) and the random policy, 2 = the same with
// replayed actions, 0 = everything else decided at run time.  With FAST == 1 the step loop holds NO global load: the general form's loads
// (io.actions, the neighbour table of a buyer with more than eight neighbours) sit behind run-time conditions that are never true for such a
// spec, but the s_waitcnt vmcnt(0) the compiler puts behind them at the joins is executed by every step -- and loads and stores share the
// counter on gfx950, so every step waited for the previous row's stores (the kernel's only HBM traffic) to be acknowledged.
template <bool DYN, int NT, int FAST>
__global__ __launch_bounds__(NT, STKR_MINWAVES(NT)) void phx_stk_rollout_kernel(const DevSpec* __restrict__ spp_, const phx_rollout_io io_) {
  // The spec (device memory, DevSpec::self_dev) and the rollout arguments (kernarg) are read through the scalar cache where they
  // are used: constant-address-space pointers whose provenance is hidden again at every step (STKR_REFRESH).  By value the two
  // structs cost 126 spilled SGPRs (v_readlane / v_writelane inside the step loop).
  typedef const __attribute__((address_space(4))) char* stk_kptr_t;
  stk_kptr_t spc = (stk_kptr_t)spp_;
  stk_kptr_t kp = (stk_kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
#define sp (*(const DevSpec*)spc)
#define io (*(const phx_rollout_io*)(kp + 8))
#define STKR_REFRESH() asm volatile("" : "+s"(spc), "+s"(kp))
  STKR_REFRESH();
#ifdef PHX_TIMING
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int A = sp.A, B = sp.B;
  const int nSell = sp.kind_count[PHX_KIND_SELLER], nBuy = sp.kind_count[PHX_KIND_BUYER];
  double* s_posted = (double*)smem;
  double* s_price = s_posted + nSell;
  double* s_rev = s_price + nSell;
  double* s_cache = s_rev + nSell;                       // [A] self._rewards
  double* s_paid = s_cache + A;                          // [nBuy]
  int* s_tx = (int*)(s_paid + nBuy);
  int* s_count = s_tx + nSell;
  uint8_t* s_sent = (uint8_t*)(s_count + nSell);
  uint8_t* s_cv = s_sent + nSell;                        // [A] reward cache valid
  uint8_t* s_bought = s_cv + A;                          // [nBuy]
  uint8_t* s_conn = s_bought + nBuy;                     // [n_conn] StochasticNetwork: connection is in this episode's graph
  // the buyers' values in LDS, not in a register pair per slot: with 64 VGPRs (8 waves per SIMD: the step is a latency
  // chain) the 512-thread instantiation spilled 10 VGPRs to scratch -- 0.54 GB of FETCH_SIZE per T = 50 launch at
  // 128 x 1024, B = 4096 (profiles/r03a), and a scratch reload on the step's critical path
  // (an offset from `smem`, not pointer bits rounded up: through uintptr_t the pointer loses its LDS address space, s_val[] became FLAT loads,
  //  and the s_waitcnt vmcnt(0) lgkmcnt(0) behind each -- one per observing buyer and step -- waited for the row's stores in flight: round 6)
  const int val_off = ((int)((char*)(s_conn + (DYN ? sp.n_conn : 0)) - smem) + 7) & ~7;
  double* s_val = (double*)(smem + val_off);                                                       // [nBuy]
  // FAST: a 4-byte ordering KEY per posted price, [nSell + 1] with +inf behind the last seller.  A posted price is 1.0 (a reset) or a seller's
  // action, a float32: the key is the price itself as f32 (exact), a buyer's search for its cheapest neighbour compares 4-byte keys -- four
  // candidates in flight at a time, unused neighbour slots pointing at the +inf entry, no branch -- and decides exactly like the f64 search
  // (same first minimum, NaN never smaller); the price comes back from s_posted.  A state blob whose posted prices are NOT float32 values
  // (written by hand) switches the block to RANK keys -- the number of posted prices below this one, recomputed whenever prices change
  // (rerank): the same order as the f64 values, whatever they are.
  float* s_postf = (float*)(s_val + nBuy);
  auto rerank = [&]() __attribute__((always_inline)) {
    for (int kr = tid; kr < nSell; kr += NT) {
      const double v = s_posted[kr];
      int c = 0;
      for (int l = 0; l < nSell; ++l) c += s_posted[l] < v ? 1 : 0;
      s_postf[kr] = (v != v) ? (float)v : (float)c;            // (a NaN price keeps a NaN key: never smaller, like the f64 compare)
    }
  };
  // (Measured and dropped, round 3: the row's four u8 planes staged in LDS and stored as 16-byte pieces after the step's last
  //  barrier instead of one byte store per lane and plane -- 12 fewer store instructions per lane and step, but one more
  //  dependent stage on a step that is a latency chain: 26.9 -> 29.0 us per step at 128 x 1024, B = 4096.)
  constexpr bool dyn = DYN;
  const int64_t sbase = (int64_t)b * nSell, bbase = (int64_t)b * nBuy, abase = (int64_t)b * A;
  const int64_t genv = sp.env_offset + b;
  uint32_t episode = dyn ? (uint32_t)fld<int32_t>(sp, F_ENV_EPISODE)[b] : 0u;
  int n_resets = 0;
  if (dyn) for (int i = tid; i < sp.n_conn; i += NT) s_conn[i] = fld<uint8_t>(sp, F_NET_CONN_ON)[(int64_t)b * sp.n_conn + i];

  bool my_pf_bad = false;
  for (int k = tid; k < nSell; k += NT) {
    const double pv = fld<double>(sp, F_SELLER_POSTED)[sbase + k];
    s_posted[k] = pv; s_price[k] = fld<double>(sp, F_SELLER_PRICE)[sbase + k];
    s_rev[k] = fld<double>(sp, F_SELLER_REVENUE)[sbase + k]; s_tx[k] = fld<int32_t>(sp, F_SELLER_TX)[sbase + k];
    if (FAST) { const float pf = (float)pv; s_postf[k] = pf; my_pf_bad |= !((double)pf == pv) && (pv == pv); }
  }
  if (FAST && tid == 0) s_postf[nSell] = __builtin_inff();
  for (int k = tid; k < nBuy; k += NT) {
    s_paid[k] = fld<double>(sp, F_BUYER_PAID)[bbase + k]; s_bought[k] = (uint8_t)fld<int32_t>(sp, F_BUYER_BOUGHT)[bbase + k];
  }
  for (int a = tid; a < A; a += NT) {
    s_cache[a] = fld<double>(sp, F_ENV_REW_CACHE)[abase + a]; s_cv[a] = fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[abase + a];
  }
  // the lane's agents: static data in registers.  rec = seller | flags(leaders' step) << 1 | flags(followers' step) << 4 |
  // deg << 8 | kind_rank << 16; nbp: a buyer's neighbours (seller ranks, packed u16), for a seller nbp[0] = its degree;
  // rw2 / rt2: the agent's word of tick rt2, kept from the Philox block of its previous acting tick
  uint32_t rec[STKR_SLOTS], nbp[STKR_SLOTS][4], rw2[STKR_SLOTS], rt2[STKR_SLOTS];
  uint32_t nb8[STKR_SLOTS][2];                                  // FAST: a buyer's eight neighbour ranks as BYTES (at most 254 sellers; unused slots: nSell, the +inf key), a seller's degree in [0]
#pragma unroll
  for (int k = 0; k < STKR_SLOTS; ++k) {
    const int a = tid + k * NT;
    rec[k] = 0; rw2[k] = 0; rt2[k] = 0xffffffffu;
    nbp[k][0] = nbp[k][1] = nbp[k][2] = nbp[k][3] = 0; nb8[k][0] = nb8[k][1] = 0;
    if (FAST || (!dyn && sp.stk_packed)) {
      // the host's packed per-agent tables (the fast step kernel's): three independent loads per slot instead of a chain of up to twelve
      // (record -> flags -> eight neighbour ranks) -- the block's setup was ~45 us of a launch's 180 us fixed cost at 4 096 envs
      if (a < A) {
        const uint32_t r = sp.stk_rec2[a];
        const uint4 ag = *(const uint4*)(sp.stk_agent + 4 * a);
        const double v = sp.param_f[a * PHX_NPF];
        rec[k] = r; nbp[k][0] = ag.x; nbp[k][1] = ag.y; nbp[k][2] = ag.z; nbp[k][3] = ag.w;
        if (!(r & 1u)) s_val[r >> 16] = v;
        if (FAST) {
          if (r & 1u) nb8[k][0] = ag.x;                          // a seller: its degree
          else {                                                 // a buyer: ranks as bytes, neighbour slots j >= deg point at the +inf key of s_postf
            const int deg = (int)((r >> 8) & 255u);
            const uint32_t w4[4] = {ag.x, ag.y, ag.z, ag.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const uint32_t l = j < deg ? ((w4[j >> 1] >> ((j & 1) * 16)) & 0xffffu) : (uint32_t)nSell;
              nb8[k][j >> 2] |= (l & 255u) << ((j & 3) * 8);
            }
          }
        }
      }
    } else
    if (a < A) {
      const uint32_t r = sp.stk_rec[a];
      const bool seller = (r & 255u) == PHX_KIND_SELLER;
      rec[k] = (r & 0xffffff00u) | (seller ? 1u : 0u) | ((uint32_t)(sp.stk_flags[a] & 7) << 1) | ((uint32_t)(sp.stk_flags[(int64_t)A + a] & 7) << 4);
      const int kr = (int)(r >> 16), deg = (int)((r >> 8) & 255u);
      if (seller) nbp[k][0] = (uint32_t)(sp.row_ptr[a + 1] - sp.row_ptr[a]);                       // len(ctx.neighbour_ids)
      else {
        s_val[kr] = sp.param_f[a * PHX_NPF];
        if (!dyn && deg <= 8) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < deg) nbp[k][j >> 1] |= (uint32_t)sp.stk_nbr[(int64_t)j * nBuy + kr] << ((j & 1) * 16);
        }
      }
    }
  }
  for (int k = tid; k < nSell; k += NT) { s_count[k] = 0; s_sent[k] = 0; }    // per-step inbox counters: cleared again by the booking pass
  int step = fld<int32_t>(sp, F_ENV_STEP)[b];
  uint32_t tick = (uint32_t)fld<int32_t>(sp, F_ENV_TICK)[b];
  __syncthreads();
  const bool pf_rank = FAST && __syncthreads_or(my_pf_bad) != 0;      // (uniform; for the whole launch)
  if (pf_rank) { rerank(); __syncthreads(); }
#ifdef PHX_TIMING
  unsigned long long stm[8] = {0}, sprev = __builtin_readcyclecounter();
  const unsigned long long rt_loop = __builtin_amdgcn_s_memrealtime();
#endif

  // a buyer's cheapest current neighbour: first minimum in neighbour order (price slots hold the sellers' posted
  // prices); jr < 0: no neighbour this episode
  auto cheapest = [&](int k, int kr, int deg, int& jr) __attribute__((always_inline)) {
    double best = 0.0; jr = -1;
    if (FAST) {                                                // 4-byte keys, four in flight at a time, no branch: see s_postf
      float bf = 0.f; uint32_t bl = 0u;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const uint32_t w = nb8[k][h >> 1] >> ((h & 1) * 16), l0 = w & 0xffu, l1 = (w >> 8) & 0xffu;
        const float p0 = s_postf[l0], p1 = s_postf[l1];
        if (h == 0) { bf = p0; bl = l0; }
        else { const bool lt = p0 < bf; bf = lt ? p0 : bf; bl = lt ? l0 : bl; }
        { const bool lt = p1 < bf; bf = lt ? p1 : bf; bl = lt ? l1 : bl; }
        if (h & 1) __builtin_amdgcn_sched_barrier(0);                        // (the scheduler would put all eight reads in flight: a spill at 64 VGPRs)
      }
      jr = deg > 0 ? (int)bl : -1;
      return deg > 0 ? s_posted[bl] : 0.0;
    }
    if (!dyn && deg <= 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < deg) {
          const int l = (int)((nbp[k][j >> 1] >> ((j & 1) * 16)) & 0xffffu);
          const double v = s_posted[l];
          if (j == 0 || v < best) { best = v; jr = l; }
        }
    } else {
      const uint16_t* nb = sp.stk_nbr + kr;
      for (int j = 0; j < deg; ++j) {
        if (dyn && !s_conn[sp.stk_nbr_conn[(int64_t)j * nBuy + kr]]) continue;
        const int l = nb[(int64_t)j * nBuy]; const double v = s_posted[l];
        if (jr < 0 || v < best) { best = v; jr = l; }
      }
    }
    return best;
  };

  // The buyers' cheapest neighbour is needed twice per pair of steps -- by the observation after the leaders' step (min over the price
  // slots) and by the order of the followers' step -- from the SAME posted prices: they change only in the booking pass of a step in
  // which a seller acts, at a reset, and (dynamic graphs) with the connectivity a reset resamples.  The lane keeps the seller's rank per
  // slot (one byte each: 0xFE not computed since the last change, 0xFF no neighbour) and reads its price back from LDS: one lookup instead
  // of eight dependent ones (the kernel is bound by its ~150 instructions per agent and step, of which the two searches were ~55).
  const bool cj_ok = nSell <= 253 && STKR_SLOTS <= 4;
  bool my_sa_lead = false, my_sa_foll = false;
#pragma unroll
  for (int k = 0; k < STKR_SLOTS; ++k) if (rec[k] & 1u) { my_sa_lead |= ((rec[k] >> 1) & 1u) != 0; my_sa_foll |= ((rec[k] >> 4) & 1u) != 0; }
  const bool sa_lead = __syncthreads_or(my_sa_lead) != 0, sa_foll = __syncthreads_or(my_sa_foll) != 0;     // does any seller act on a leaders' / followers' step
  uint32_t cj = 0xFEFEFEFEu;
  auto cheapest_c = [&](int k, int kr, int deg, int& jr) __attribute__((always_inline)) {
    const uint32_t c = (cj >> (8 * k)) & 255u;
    if (!cj_ok || c == 0xFEu) {
      const double best = cheapest(k, kr, deg, jr);
      cj = (cj & ~(255u << (8 * k))) | ((jr < 0 ? 255u : (uint32_t)jr) << (8 * k));
      return best;
    }
    jr = c == 255u ? -1 : (int)c;
    return jr >= 0 ? s_posted[jr] : 0.0;
  };

  // (Measured and dropped, round 6: terminated / truncated -- the same byte for a whole row, the rows that end an episode known from the step
  //  counter at entry -- written for ALL T rows of the env as 16-byte pieces before the first step, like the supply-chain kernel's flag planes:
  //  6 of a lane's 21 store instructions per step leave the loop, yet the marginal cost of a step does not move (5.9 us per block-step in a
  //  one-round launch either way; compute alone, without any store, is 3.4), a one-round launch pays the 230 KB burst per block up front
  //  (+5 %) and B = 4 096 gains 0-3 %.  As 8-byte pieces inside the loop they cost the 64-VGPR instantiations a spill.)
  const bool wide_flags = false;
  for (int t = 0; t < io.T; ++t) {
    STKR_REFRESH();
    const int tt = step + 1;                                                 // env.py:252
    const int lsh = (tt & 1) ? 1 : 4;                                        // stackelberg.py:133-137: leaders on odd steps
    const int64_t row = ((int64_t)t * B + b) * A;
    // what derives from the packed words (ranks, LDS addresses, flags) is recomputed per step: hoisted out of the step
    // loop it costs ~90 VGPRs and the occupancy with them
#pragma unroll
    for (int k = 0; k < STKR_SLOTS; ++k)
      if (FAST) asm volatile("" : "+v"(rec[k]), "+v"(nb8[k][0]), "+v"(nb8[k][1]));
      else asm volatile("" : "+v"(rec[k]), "+v"(nbp[k][0]), "+v"(nbp[k][1]), "+v"(nbp[k][2]), "+v"(nbp[k][3]));
    int ltid = tid;
    asm volatile("" : "+v"(ltid));
    // ---- acting phase ---------------------------------------------------------------------------
    float act[STKR_SLOTS];
#pragma unroll
    for (int k = 0; k < STKR_SLOTS; ++k) {
      const int a = ltid + k * NT;
      float action = 0.f;
      if (a < A && ((rec[k] >> lsh) & 1u)) {
        const int kr = (int)(rec[k] >> 16), deg = (int)((rec[k] >> 8) & 255u);
        const bool seller = (rec[k] & 1u) != 0;
        if (FAST == 2 || (FAST == 0 && io.actions)) action = *(const float*)((const char*)(io.actions + row) + (size_t)((uint32_t)a * 4u));
        else {                                                               // the agent's word of this tick: rank j
          uint32_t word = rw2[k];
          if (rt2[k] != tick) {                                              // not kept from the agent's previous acting tick
            uint32_t w[4];
            rng_block(sp.seed, genv, tick, a, 0, 0, w);
            word = rng_pick(w, tick);
            rw2[k] = rng_pick(w, tick + 2u);
            rt2[k] = (tick & 2u) ? 0xffffffffu : tick + 2u;                  // the block covers ticks 4q .. 4q + 3
          }
          uint32_t y, aj;
          if (!rng_split(word, y, aj)) rng_group_y(sp.seed, genv, tick, a, 0, 1, &aj);
          action = seller ? (float)aj * (1.0f / 274877.0f) : (aj < 137438u ? 1.0f : 0.0f);
        }
        if (seller) { s_price[kr] = (double)action; s_sent[kr] = 1; }
        else {
          int bought = 0; double paid = 0.0;
          if (action > 0.5f && deg > 0) {
            int jr;
            const double best = cheapest_c(k, kr, deg, jr);
            if (jr >= 0) { bought = 1; paid = best; atomicAdd(&s_count[jr], 1); }
          }
          s_bought[kr] = (uint8_t)bought; s_paid[kr] = paid;
        }
      }
      act[k] = action;
      __builtin_amdgcn_sched_barrier(0);
    }
    STICK(0);
    stk_lds_barrier();      // orders LDS only: the row's stores stay in flight
    STICK(1);
    // ---- pre_message_resolution + the single round --------------------------------------------------
    for (int kr = tid; kr < nSell; kr += NT) {
      double rev = s_rev[kr]; int tx = s_tx[kr];
      if ((tt & 1) == 0) { rev = 0.0; tx = 0; }
      const int n = s_count[kr];
      if (n > 0) {
        const double amount = __dmul_rn(s_price[kr], 1.0);
        rev = stk_book(rev, amount, n);
        tx += n;
      }
      s_rev[kr] = rev; s_tx[kr] = tx;
      if (s_sent[kr]) { s_posted[kr] = s_price[kr]; if (FAST) s_postf[kr] = (float)s_price[kr]; }
      s_count[kr] = 0; s_sent[kr] = 0;
    }
    stk_lds_barrier();      // orders LDS only: the row's stores stay in flight
    if ((tt & 1) ? sa_lead : sa_foll) {                                      // (uniform) the booking pass may have changed posted prices
      cj = 0xFEFEFEFEu;
      if (FAST && pf_rank) { rerank(); stk_lds_barrier(); }
    }
    STICK(2);
    // ---- obs / reward / flags -> trajectory row (stackelberg.py:142-196) -----------------------------
    const bool terminal = (tt == sp.num_steps);
    const bool last = (t == io.T - 1);
    char* const p_obs = (char*)(io.obs + row * 2);
    char* const p_act = (char*)(io.action_out + row);
    char* const p_rew = (char*)(io.reward + row);
    char* const p_ter = (char*)(io.terminated + row);
    char* const p_tru = (char*)(io.truncated + row);
    char* const p_ov = (char*)(io.obs_valid + row);
    char* const p_rv = (char*)(io.reward_valid + row);
#pragma unroll
    for (int k = 0; k < STKR_SLOTS; ++k) {
      const int a = ltid + k * NT;
      if (a >= A) continue;
      const uint32_t fl = rec[k] >> lsh;                                     // 1 acts, 2 observes, 4 rewarded
      const int kr = (int)(rec[k] >> 16), deg = (int)((rec[k] >> 8) & 255u);
      const bool seller = (rec[k] & 1u) != 0;
      float ob0 = 0.f, ob1 = 0.f; uint8_t ov = 0;
      if (fl & 2u) {
        ov = 1;
        if (seller) {
          int sd = (int)(FAST ? nb8[k][0] : nbp[k][0]);
          if (dyn) { sd = 0; for (int e = sp.row_ptr[a]; e < sp.row_ptr[a + 1]; ++e) sd += s_conn[sp.col_conn[e]] ? 1 : 0; }
          ob0 = sd ? (float)((double)s_tx[kr] / (double)sd) : 0.f; ob1 = (float)s_price[kr];
        } else {
          int jr;
          const double mn = cheapest_c(k, kr, deg, jr);                      // min over the price slots (none: 1.0)
          ob0 = (float)(jr >= 0 ? mn : 1.0); ob1 = (float)s_val[kr];
        }
      }
      uint8_t cv = s_cv[a]; double cache = s_cache[a];
      if (fl & 4u) {
        cache = seller ? s_rev[kr] : (s_bought[kr] ? __dsub_rn(s_val[kr], s_paid[kr]) : 0.0);
        cv = 1; s_cache[a] = cache; s_cv[a] = 1;
      }
      uint8_t rv = 0; double rw = 0.0;
      if (terminal) { rv = cv ? 1 : 2; rw = cv ? cache : 0.0; }
      else if (ov && cv) { rv = 1; rw = cache; }
      // scalar row bases + 32-bit lane offsets: the stores take the SGPR-base + VGPR-offset form (per-slot 64-bit
      // pointers kept across the step loop cost ~50 VGPRs)
      const uint32_t ua = (uint32_t)a;
      // NON-TEMPORAL stores for the three f32 planes (a wave writes whole 128-byte lines of them): the row then leaves for HBM while the next
      // step's acting phase computes, instead of sitting dirty in the L2 until the next row's burst evicts it -- the blocks of a round run in
      // step, so without this HBM idles through every compute phase and every store phase waits for it (round 6: 5.3 -> ? us per block-step)
      typedef float stk_f2v __attribute__((ext_vector_type(2)));
      __builtin_nontemporal_store((stk_f2v){ob0, ob1}, (stk_f2v*)(p_obs + (size_t)(ua * 8u)));
      __builtin_nontemporal_store(act[k], (float*)(p_act + (size_t)(ua * 4u)));
      __builtin_nontemporal_store((float)rw, (float*)(p_rew + (size_t)(ua * 4u)));
      if (!wide_flags) { *(uint8_t*)(p_ter + (size_t)ua) = 0; *(uint8_t*)(p_tru + (size_t)ua) = terminal; }
      *(uint8_t*)(p_ov + (size_t)ua) = ov; *(uint8_t*)(p_rv + (size_t)ua) = rv;
      if (last && io.last_obs) {
        // the observation the next fragment starts from: after a terminal step, the reset's (the
        // leaders observe their reset state, stackelberg.py:95-109: a seller sees tx / deg = 0 and
        // price 0, a buyer among the leaders no prices yet -> 1.0, and its value)
        if (terminal) {
          const bool lead_buyer = (rec[k] & 2u) != 0 && !seller;             // acts on the leaders' step
          ob0 = lead_buyer ? 1.0f : 0.f;
          ob1 = lead_buyer ? (float)s_val[kr] : 0.f;
        }
        *(float2*)(io.last_obs + (abase + a) * 2) = make_float2(ob0, ob1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    step = tt; ++tick;
    STICK(3);
    stk_lds_barrier();      // orders LDS only: the row's stores stay in flight
    STICK(4);
    if (terminal) {                                                          // the caller's env.reset()
      for (int k = tid; k < nSell; k += NT) { s_posted[k] = 1.0; s_price[k] = 0.0; s_rev[k] = 0.0; s_tx[k] = 0; if (FAST) s_postf[k] = 1.0f; }
      for (int k = tid; k < nBuy; k += NT) { s_paid[k] = 0.0; s_bought[k] = 0; }
      for (int a = tid; a < A; a += NT) s_cv[a] = 0;
      cj = 0xFEFEFEFEu;
      if (dyn) {                                                             // resample_connectivity network.py:438-447
        for (int i = tid; i < sp.n_conn; i += NT) s_conn[i] = (uint8_t)rng_connection(sp.seed, genv, episode, i, sp.conn_rate[i]);
        ++episode; ++n_resets;
      }
      step = 0;
      stk_lds_barrier();      // orders LDS only: the row's stores stay in flight
    }
  }
#ifdef PHX_TIMING
  if (blockIdx.x < 64 && (tid & 63) == 0 && (tid >> 6) == 1) for (int q = 0; q < 8; ++q) atomicAdd(&g_stk_tm[q], stm[q]);
  const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
#endif
  if (dyn && n_resets > 0) {
    for (int i = tid; i < sp.n_conn; i += NT) fld<uint8_t>(sp, F_NET_CONN_ON)[(int64_t)b * sp.n_conn + i] = s_conn[i];
    if (tid == 0) fld<int32_t>(sp, F_ENV_EPISODE)[b] = (int32_t)episode;
  }
  for (int k = tid; k < nSell; k += NT) {
    fld<double>(sp, F_SELLER_POSTED)[sbase + k] = s_posted[k]; fld<double>(sp, F_SELLER_PRICE)[sbase + k] = s_price[k];
    fld<double>(sp, F_SELLER_REVENUE)[sbase + k] = s_rev[k]; fld<int32_t>(sp, F_SELLER_TX)[sbase + k] = s_tx[k];
  }
  for (int k = tid; k < nBuy; k += NT) {
    fld<double>(sp, F_BUYER_PAID)[bbase + k] = s_paid[k]; fld<int32_t>(sp, F_BUYER_BOUGHT)[bbase + k] = s_bought[k];
  }
  for (int a = tid; a < A; a += NT) {
    fld<double>(sp, F_ENV_REW_CACHE)[abase + a] = s_cache[a]; fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[abase + a] = s_cv[a];
  }
  if (tid == 0) { fld<int32_t>(sp, F_ENV_STEP)[b] = step; fld<int32_t>(sp, F_ENV_TICK)[b] = (int32_t)tick; }
#ifdef PHX_TIMING
  __syncthreads();
  if (tid == 0) {
    const unsigned long long rt_exit = __builtin_amdgcn_s_memrealtime();
    atomicMin(&g_stk_rt[0], rt_entry); atomicMax(&g_stk_rt[1], rt_exit);
    atomicAdd(&g_stk_rt[2], rt_loop - rt_entry); atomicAdd(&g_stk_rt[3], rt_end - rt_loop); atomicAdd(&g_stk_rt[4], rt_exit - rt_end); atomicAdd(&g_stk_rt[5], 1ull);
  }
#endif
}
#undef sp
#undef io
#undef STKR_REFRESH

size_t phx_stk_rollout_lds(const DevSpec& sp) {
  const size_t nSell = sp.kind_count[PHX_KIND_SELLER], nBuy = sp.kind_count[PHX_KIND_BUYER], A = sp.A;
  return 8 * (3 * nSell + A + nBuy) + 4 * (2 * nSell) + nSell + A + nBuy + (sp.dynamic_graph ? (size_t)sp.n_conn : 0) + 32 + 8 + 8 * nBuy +   // + the buyers' values
         4 * (nSell + 1) + 8;                                                                                                                     // + the posted prices' keys
}

hipError_t phx_launch_stk_rollout(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st) {
#ifdef PHX_TIMING
  { static int calls = 0; if (getenv("PHX_TIMING_DUMP") && ++calls == 10) { (void)hipDeviceSynchronize(); unsigned long long h[8];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stk_tm), sizeof h); fprintf(stderr, "STK_TIMING cycles per step (wave 1 of blocks < 64): act %.0f | bar %.0f | book+bar %.0f | out %.0f | bar %.0f\n",
      h[0] / (9.0 * 64 * io.T), h[1] / (9.0 * 64 * io.T), h[2] / (9.0 * 64 * io.T), h[3] / (9.0 * 64 * io.T), h[4] / (9.0 * 64 * io.T)); }
    if (getenv("PHX_TIMING_DUMP") && calls >= 8 && calls <= 10) {
      if (calls > 8) { (void)hipDeviceSynchronize(); unsigned long long r[8]; (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_stk_rt), sizeof r); const double n = (double)r[5];
        fprintf(stderr, "STK_RT launch %d: wall %.1f us | per block: setup %.2f loop %.2f epilogue %.2f us | blocks %.0f\n", calls - 1, (r[1] - r[0]) * 0.01, r[2] * 0.01 / n, r[3] * 0.01 / n, r[4] * 0.01 / n, n); }
      unsigned long long z[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stk_rt), z, sizeof z); } }
#endif
  // threads per env: a block whose STKR_SLOTS passes cover the agents
  const int nt_env = phx_knobs().stk_rollout_nt;
  // (1 152 agents: 384 threads with three full slots 35.4 us per step, 33.0 with the registers capped for 5 blocks per CU;
  //  512 threads -- 32 waves per CU, a quarter of the lanes idle in the third slot -- 28.1: the step is a latency chain)
  int nt = 1024;
  for (int cand : {128, 256, 512, 1024}) if (STKR_SLOTS * cand >= sp.A) { nt = cand; break; }
  if (nt_env && STKR_SLOTS * nt_env >= sp.A) nt = nt_env;
  if (STKR_SLOTS * nt < sp.A) return hipErrorInvalidConfiguration;
  const size_t lds = phx_stk_rollout_lds(sp);
  phx_note_kernel("phx_stk_rollout_kernel");
#define PHX_LAUNCH_STKR(NT_) do { if (sp.dynamic_graph) hipLaunchKernelGGL((phx_stk_rollout_kernel<true, NT_, 0>), dim3(sp.B), dim3(NT_), lds, st, sp.self_dev, io); \
                                  else if (!sp.stk_packed || sp.kind_count[PHX_KIND_SELLER] > 254) hipLaunchKernelGGL((phx_stk_rollout_kernel<false, NT_, 0>), dim3(sp.B), dim3(NT_), lds, st, sp.self_dev, io); \
                                  else if (io.actions) hipLaunchKernelGGL((phx_stk_rollout_kernel<false, NT_, 2>), dim3(sp.B), dim3(NT_), lds, st, sp.self_dev, io); \
                                  else hipLaunchKernelGGL((phx_stk_rollout_kernel<false, NT_, 1>), dim3(sp.B), dim3(NT_), lds, st, sp.self_dev, io); } while (0)
  switch (nt) {
    case 128: PHX_LAUNCH_STKR(128); break;
    case 256: PHX_LAUNCH_STKR(256); break;
    case 384: PHX_LAUNCH_STKR(384); break;
    case 512: PHX_LAUNCH_STKR(512); break;
    default: PHX_LAUNCH_STKR(1024); break;
  }
#undef PHX_LAUNCH_STKR
  return hipGetLastError();
}

// buyer.prices[b][k][r] = seller.posted[b][neighbour k of buyer r]
__global__ __launch_bounds__(256) void phx_stk_materialise_kernel(const DevSpec sp) {
  const int nSell = sp.kind_count[PHX_KIND_SELLER], nBuy = sp.kind_count[PHX_KIND_BUYER];
  const int64_t n = (int64_t)sp.B * sp.buyer_nnz;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / sp.buyer_nnz; const int slot = (int)(i - b * sp.buyer_nnz);
    const int l = sp.stk_nbr[slot];
    if (l == 0xFFFF) continue;
    // a slot whose connection is off this episode was not written since the reset: 1.0
    const bool on = !sp.dynamic_graph || fld<uint8_t>(sp, F_NET_CONN_ON)[b * sp.n_conn + sp.stk_nbr_conn[slot]];
    fld<double>(sp, F_BUYER_PRICES)[i] = on ? fld<double>(sp, F_SELLER_POSTED)[b * nSell + l] : 1.0;
  }
  (void)nBuy;
}

hipError_t phx_launch_stk_step(const DevSpec& sp, const phx_step_io& io, hipStream_t st) {
#ifdef PHX_TIMING
  { static int calls = 0; if (getenv("PHX_TIMING_DUMP") && ++calls == 101) { (void)hipDeviceSynchronize(); unsigned long long h[8];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stk_tm), sizeof h); const double n = 100.0 * 64;
      fprintf(stderr, "STK_STEP_TIMING cycles (wave 1 of blocks < 64): loads+bar %.0f | act %.0f | bar %.0f | book %.0f | bar %.0f | out %.0f\n",
              h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n); } }
#endif
  const int nSell = sp.kind_count[PHX_KIND_SELLER];
  const int nBuy = sp.kind_count[PHX_KIND_BUYER];
  const size_t lds = g_stk_paid_off(nSell) + (size_t)nBuy * 8 + (((size_t)nBuy + 15) & ~(size_t)15) + 32 + (sp.dynamic_graph ? (size_t)sp.n_conn : 0);
  const int fast_env = phx_knobs().stk_step_fast;
  const int nt_env = phx_knobs().stk_step_nt;
  if (sp.stk_packed && fast_env && sp.A <= STKR_SLOTS * 1024) {
    int nt = 1024;
    for (int cand : {128, 256, 512, 1024}) if (STKR_SLOTS * cand >= sp.A) { nt = cand; break; }
    if (nt_env && STKR_SLOTS * nt_env >= sp.A) nt = nt_env;
    phx_note_kernel("phx_stk_step_fast_kernel");
    switch (nt) {
      case 128: hipLaunchKernelGGL((phx_stk_step_fast_kernel<128>), dim3(sp.B), dim3(128), lds, st, sp, io); break;
      case 256: hipLaunchKernelGGL((phx_stk_step_fast_kernel<256>), dim3(sp.B), dim3(256), lds, st, sp, io); break;
      case 384: hipLaunchKernelGGL((phx_stk_step_fast_kernel<384>), dim3(sp.B), dim3(384), lds, st, sp, io); break;
      case 512: hipLaunchKernelGGL((phx_stk_step_fast_kernel<512>), dim3(sp.B), dim3(512), lds, st, sp, io); break;
      default: hipLaunchKernelGGL((phx_stk_step_fast_kernel<1024>), dim3(sp.B), dim3(1024), lds, st, sp, io); break;
    }
    return hipGetLastError();
  }
  phx_note_kernel("phx_stk_step_kernel");
  if (sp.dynamic_graph) hipLaunchKernelGGL(phx_stk_step_kernel<true>, dim3(sp.B), dim3(STK_NT), lds, st, sp, io);
  else hipLaunchKernelGGL(phx_stk_step_kernel<false>, dim3(sp.B), dim3(STK_NT), lds, st, sp, io);
  return hipGetLastError();
}

hipError_t phx_launch_stk_materialise(const DevSpec& sp, hipStream_t st) {
  const int64_t n = (int64_t)sp.B * sp.buyer_nnz;
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 8192);
  if (n > 0) hipLaunchKernelGGL(phx_stk_materialise_kernel, dim3(blocks), dim3(256), 0, st, sp);
  return hipGetLastError();
}

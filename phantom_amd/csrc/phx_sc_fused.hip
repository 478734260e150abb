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
#include <cstdlib>

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

__global__ __launch_bounds__(SC_NT) void phx_sc_step_kernel(const DevSpec sp, const phx_step_io io,
                                                            const int epb, const int stage_exo) {
  // A block owns `epb` WHOLE envs (epb * S <= 256 lanes): the per-env words (step, tick, stage)
  // are read by every lane of the env and rewritten by its shop-0 lane, so readers and writer
  // must sit in one workgroup with a barrier between the two.
  extern __shared__ __attribute__((aligned(16))) unsigned char s_exo[];
  const int nS = sp.S, A = sp.A;
  const int64_t b_first = (int64_t)blockIdx.x * epb;
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
    for (int64_t off = lds_base + (int64_t)threadIdx.x * 16; off < hi16; off += (int64_t)SC_NT * 16)
      *(uint4*)(s_exo + (off - lds_base)) = *(const uint4*)(io.exo + off);
    for (int64_t off = (hi16 > lds_base ? hi16 : lds_base) + threadIdx.x; off < hi; off += SC_NT)
      s_exo[off - lds_base] = io.exo[off];
  }
  const int b = active ? (int)(b_first + threadIdx.x / nS) : (int)b_first;
  const int s = active ? (int)(threadIdx.x % nS) : 0;
  const int a_shop = sp.shop_agent[s];
  const int cur_stage0 = (sp.env_type == PHX_ENV_FSM) ? fld<int32_t>(sp, F_ENV_STAGE)[b] : 0;
  const int step0 = fld<int32_t>(sp, F_ENV_STEP)[b];
  const uint32_t tick0 = (uint32_t)fld<int32_t>(sp, F_ENV_TICK)[b];
  __syncthreads();          // exo rows staged; every lane has read the per-env words
  if (!active) return;

  const int cur_stage = cur_stage0;
  const int list = cur_stage;
  const int t = step0 + 1;                                                   // env.py:252
  const uint32_t tick = tick0;

  ShopLane st;
  st.stock = fld<int32_t>(sp, F_SHOP_STOCK)[g];
  st.delivered = fld<int32_t>(sp, F_SHOP_DELIVERED)[g];
  const bool shop_acts = sp.act_mask[(int64_t)list * A + a_shop] != 0;
  const bool has_action = shop_acts && (io.action_valid == nullptr || io.action_valid[g] != 0);
  const float action = has_action ? io.actions[g] : 0.0f;

  // D = sum over the shop's inbox of OrderRequest sizes (customers that act in this stage)
  const int c_lo = sp.shop_cust_ptr[s], c_hi = sp.shop_cust_ptr[s + 1];
  const uint8_t* cact = sp.shop_cust_act + (int64_t)list * sp.n_exo;
  int D = 0; bool any_order = false;
  if (io.exo) {
    const int64_t row = (int64_t)b * sp.n_exo;
    for (int k = c_lo; k < c_hi; ++k)
      if (cact[k]) {
        any_order = true;
        const int64_t idx = row + sp.shop_cust_exo[k];
        D += staged ? s_exo[idx - lds_base] : io.exo[idx];
      }
  } else {
    bool all_order = true;
    for (int k = c_lo; k < c_hi; ++k) { any_order |= cact[k] != 0; all_order &= cact[k] != 0; }
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
  float ob[3] = {0.f, 0.f, 0.f};
  uint8_t ov = 0, rv = 0;
  double rw = 0.0;
  if (sp.env_type == PHX_ENV_PLAIN) {
    shop_obs(st.stock, st.sales, st.missed, sp.param_i[a_shop * PHX_NPI + 1], ob);
    rw = shop_reward(st.sales, st.stock);
    ov = 1; rv = 1;
  } else {
    double* rc = fld<double>(sp, F_ENV_REW_CACHE) + g;
    uint8_t* rcv = fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID) + g;
    float* oc = fld<float>(sp, F_ENV_OBS_CACHE) + g * 3;
    uint8_t* ocv = fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID) + g;
    const bool observes = sp.obs_mask[(int64_t)list * A + a_shop] != 0;
    if (observes) {
      shop_obs(st.stock, st.sales, st.missed, sp.param_i[a_shop * PHX_NPI + 1], ob);
      oc[0] = ob[0]; oc[1] = ob[1]; oc[2] = ob[2]; *ocv = 1;                  // fsm.py:349
    }
    uint8_t cache_valid = *rcv; double cache = *rc;
    if (sp.rew_mask[(int64_t)list * A + a_shop]) {                            // fsm.py:334-335,350
      cache = shop_reward(st.sales, st.stock); cache_valid = 1;
      *rc = cache; *rcv = 1;
    }
    if (all_trunc) {                                                          // fsm.py:360-375
      ov = *ocv;
      ob[0] = ov ? oc[0] : 0.f; ob[1] = ov ? oc[1] : 0.f; ob[2] = ov ? oc[2] : 0.f;
      rv = cache_valid ? 1 : 2; rw = cache_valid ? cache : 0.0;
    } else if (observes) {                                                    // fsm.py:378
      ov = 1; rv = cache_valid ? 1 : 2; rw = cache_valid ? cache : 0.0;
    } else { ob[0] = ob[1] = ob[2] = 0.f; }
  }
  io.obs[g * 3 + 0] = ob[0]; io.obs[g * 3 + 1] = ob[1]; io.obs[g * 3 + 2] = ob[2];
  io.reward[g] = rw;
  io.obs_valid[g] = ov; io.reward_valid[g] = rv; io.done_valid[g] = 1;
  io.terminated[g] = 0; io.truncated[g] = 0;
  if (s == 0) {
    fld<int32_t>(sp, F_ENV_STEP)[b] = t;
    fld<int32_t>(sp, F_ENV_TICK)[b] = (int32_t)(tick + 1);
    if (sp.env_type == PHX_ENV_FSM) {                                         // fsm.py:355
      fld<int32_t>(sp, F_ENV_PREV_STAGE)[b] = cur_stage;
      fld<int32_t>(sp, F_ENV_STAGE)[b] = sp.stage_next[cur_stage];
    }
    io.all_terminated[b] = 0; io.all_truncated[b] = all_trunc;
  }
}

// ---- fused rollout: T steps per launch, shop state in registers, only the trajectory
//      streams to HBM.  PLAIN env; auto-reset at episode end (env.py:185-237 folded in). --------
__global__ __launch_bounds__(SC_NT) void phx_sc_rollout_v1_kernel(const DevSpec sp, const phx_rollout_io io,
                                                               const int epb) {
  const int nS = sp.S;
  const int64_t total = (int64_t)sp.B * nS;
  const int64_t b_first = (int64_t)blockIdx.x * epb;
  const int64_t b_end = (b_first + epb < sp.B) ? b_first + epb : sp.B;
  const bool active = (int)threadIdx.x < (int)(b_end - b_first) * nS;
  const int64_t g = b_first * nS + threadIdx.x;
  const int b = active ? (int)(b_first + threadIdx.x / nS) : (int)b_first;
  const int s = active ? (int)(threadIdx.x % nS) : 0;
  int step = fld<int32_t>(sp, F_ENV_STEP)[b];
  uint32_t tick = (uint32_t)fld<int32_t>(sp, F_ENV_TICK)[b];
  __syncthreads();          // per-env words are rewritten by the env's shop-0 lane at the end
  if (!active) return;
  const int a_shop = sp.shop_agent[s];
  const int norm = sp.param_i[a_shop * PHX_NPI + 1];
  const int c_lo = sp.shop_cust_ptr[s], c_hi = sp.shop_cust_ptr[s + 1];
  const int K = c_hi - c_lo;
  const int64_t genv = sp.env_offset + b;

  ShopLane st;
  st.stock = fld<int32_t>(sp, F_SHOP_STOCK)[g];
  st.sales = fld<int32_t>(sp, F_SHOP_SALES)[g];
  st.missed = fld<int32_t>(sp, F_SHOP_MISSED)[g];
  st.delivered = fld<int32_t>(sp, F_SHOP_DELIVERED)[g];
  float ob[3] = {0.f, 0.f, 0.f};

  for (int t = 0; t < io.T; ++t) {
    const int64_t o = (int64_t)t * total + g;
    int D = 0;
    uint32_t w3 = 0;
    if (io.exo) {
      const uint8_t* row = io.exo + ((int64_t)t * sp.B + b) * sp.n_exo;
      for (int k = c_lo; k < c_hi; ++k) D += row[sp.shop_cust_exo[k]];
      if (!io.actions) rng_shop_orders(sp.seed, genv, tick, s, 0, nullptr, -1, &w3);
    } else {
      D = rng_shop_orders(sp.seed, genv, tick, s, K, nullptr, -1, &w3);
    }
    const float action = io.actions ? io.actions[o] : rng_word_to_action(w3);
    sc_shop_step(st, true, action, K > 0, D);
    ++step; ++tick;
    const bool all_trunc = (step == sp.num_steps);
    shop_obs(st.stock, st.sales, st.missed, norm, ob);
    const double rw = shop_reward(st.sales, st.stock);
    io.obs[o * 3 + 0] = ob[0]; io.obs[o * 3 + 1] = ob[1]; io.obs[o * 3 + 2] = ob[2];
    io.action_out[o] = action;
    io.reward[o] = (float)rw;
    io.terminated[o] = 0;
    io.truncated[o] = all_trunc;
    if (all_trunc) {                                             // the caller's env.reset(): stock only
      st.stock = 0; step = 0;                                    // supply_chain.py:149-150
      shop_obs(st.stock, st.sales, st.missed, norm, ob);         // sales/missed stay stale
    }
  }
  fld<int32_t>(sp, F_SHOP_STOCK)[g] = st.stock;
  fld<int32_t>(sp, F_SHOP_SALES)[g] = st.sales;
  fld<int32_t>(sp, F_SHOP_MISSED)[g] = st.missed;
  fld<int32_t>(sp, F_SHOP_DELIVERED)[g] = st.delivered;
  if (io.last_obs) { io.last_obs[g * 3 + 0] = ob[0]; io.last_obs[g * 3 + 1] = ob[1]; io.last_obs[g * 3 + 2] = ob[2]; }
  if (s == 0) {
    fld<int32_t>(sp, F_ENV_STEP)[b] = step;
    fld<int32_t>(sp, F_ENV_TICK)[b] = (int32_t)tick;
  }
}

// ---- rollout v2: time-parallel.  The only sequential dependence of an episode is the stock
// recurrence  stock' = min(stock - min(D, stock) + min(R, 100 - stock), 100)  (a dozen integer
// ops); everything expensive -- the Philox draws, the IEEE divisions of the observation, the
// f64 reward, the trajectory stores -- is independent across time steps.  So a block owns G
// (env, shop) pairs x TC steps, staged in LDS as {R | stock, D, sales}:
//   phase 1 (all lanes, item = (t, pair)): action + order sum  -> LDS, action_out -> HBM
//   phase 2 (one lane per pair, sequential over t): the recurrence, in LDS
//   phase 3 (all lanes, item = (t, pair)): obs / reward / flags -> HBM, rows contiguous in pair
// Waves take whole time rows (lanes = consecutive pairs), so every trajectory store of a wave
// covers one contiguous segment of the [T][B][S] arrays.
// lean argument block of the rollout kernel (passing the whole DevSpec by value costs ~50
// spilled SGPRs per wave)
struct RollArgs {
  int32_t B, S, n_exo, num_steps, T, epb, TC;
  uint64_t seed; int64_t env_offset;
  const int32_t* shop_norm;      // [S] max_sales_per_step of each shop
  const int32_t* shop_cust_ptr;  // [S+1]
  const int32_t* shop_cust_exo;
  int32_t *stock, *sales, *missed, *delivered, *env_step, *env_tick;
  phx_rollout_io io;
};

template <int NT>
__global__ __launch_bounds__(NT) void phx_sc_rollout_kernel(const RollArgs a) {
  extern __shared__ __attribute__((aligned(16))) int s_it[];     // [TC][G][3]
  const phx_rollout_io& io = a.io;
  const int nS = a.S, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int TC = a.TC;
  const int64_t total = (int64_t)a.B * nS;
  const int64_t b_first = (int64_t)blockIdx.x * a.epb;
  const int64_t b_end = (b_first + a.epb < a.B) ? b_first + a.epb : a.B;
  const int G = (int)(b_end - b_first) * nS;
  const int64_t g_base = b_first * nS;

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

  for (int t0 = 0; t0 < a.T; t0 += TC) {
    const int tc = (a.T - t0 < TC) ? a.T - t0 : TC;
    // ---- phase 1: lanes = pairs (64 per chunk), waves = time rows ------------------------------
    for (int c = 0; c * 64 < G; ++c) {
      const int gl = c * 64 + lane;
      if (gl < G) {
        const int bl = gl / nS, s = gl - bl * nS;
        const int b = (int)b_first + bl;
        const int64_t genv = a.env_offset + b;
        const uint32_t tick0 = (uint32_t)a.env_tick[b];
        const int c_lo = a.shop_cust_ptr[s], c_hi = a.shop_cust_ptr[s + 1];
        for (int tl = wave; tl < tc; tl += NT / 64) {
          const int t = t0 + tl;
          const int64_t o = (int64_t)t * total + g_base + gl;
          int D = 0; uint32_t w3 = 0;
          if (io.exo) {
            const uint8_t* row = io.exo + ((int64_t)t * a.B + b) * a.n_exo;
            for (int k = c_lo; k < c_hi; ++k) D += row[a.shop_cust_exo[k]];
            if (!io.actions) rng_shop_order_sum(a.seed, genv, tick0 + t, s, 0, &w3);
          } else {
            D = rng_shop_order_sum(a.seed, genv, tick0 + t, s, c_hi - c_lo, &w3);
          }
          const float action = io.actions ? io.actions[o] : rng_word_to_action(w3);
          io.action_out[o] = action;
          int* it = s_it + ((int64_t)tl * G + gl) * 3;
          it[0] = dev_round_half_even(action); it[1] = D;
        }
      }
    }
    __syncthreads();
    // ---- phase 2: the stock recurrence, one lane per pair ---------------------------------------
    if (tid < G) {
      for (int tl = 0; tl < tc; ++tl) {
        int* it = s_it + ((int64_t)tl * G + tid) * 3;
        const int R = it[0], D = it[1];
        const int stock0 = st.stock;
        const int room = PHX_SHOP_MAX_STOCK - stock0;
        const int req = R < room ? R : room;                      // supply_chain.py:139
        int sales = 0, stock1 = stock0;
        if (p2_K > 0) { sales = D < stock0 ? D : stock0; stock1 = stock0 - sales; }   // :105-122
        const int ns = stock1 + req;                              // :98-103
        stock1 = ns < PHX_SHOP_MAX_STOCK ? ns : PHX_SHOP_MAX_STOCK;
        st.stock = stock1; st.sales = sales; st.missed = (p2_K > 0) ? D - sales : 0; st.delivered = req;
        it[0] = stock1; it[2] = sales;
        if (++step == a.num_steps) { st.stock = 0; step = 0; }    // episode end: env.reset() -> stock = 0
      }
    }
    __syncthreads();
    // ---- phase 3: observations / rewards / flags ------------------------------------------------
    for (int c = 0; c * 64 < G; ++c) {
      const int gl = c * 64 + lane;
      if (gl < G) {
        const int bl = gl / nS, s = gl - bl * nS;
        const int b = (int)b_first + bl;
        const float norm = (float)a.shop_norm[s];
        const int K = a.shop_cust_ptr[s + 1] - a.shop_cust_ptr[s];
        const int step0 = a.env_step[b];
        for (int tl = wave; tl < tc; tl += NT / 64) {
          const int t = t0 + tl;
          const int64_t o = (int64_t)t * total + g_base + gl;
          const int* it = s_it + ((int64_t)tl * G + gl) * 3;
          const int stock = it[0], D = it[1], sales = it[2];
          const int missed = (K > 0) ? D - sales : 0;
          // f32 IEEE division == the reference's f64 quotient cast to f32 for |ints| < 2^24
          float ob[3];
          shop_obs_f32(stock, sales, missed, norm, ob);
          io.obs[o * 3 + 0] = ob[0]; io.obs[o * 3 + 1] = ob[1]; io.obs[o * 3 + 2] = ob[2];
          io.reward[o] = (float)shop_reward(sales, stock);
          // env step counter after this step (an env stepped past num_steps without reset never truncates)
          const bool all_trunc = step0 < a.num_steps && ((step0 + t) % a.num_steps) + 1 == a.num_steps;
          io.terminated[o] = 0;
          io.truncated[o] = all_trunc;
        }
      }
    }
    __syncthreads();
  }
  if (tid < G) {
    const int64_t g = g_base + tid;
    const int s = tid % nS, b = (int)b_first + tid / nS;
    a.stock[g] = st.stock; a.sales[g] = st.sales; a.missed[g] = st.missed; a.delivered[g] = st.delivered;
    if (io.last_obs) {
      float ob[3];
      shop_obs(st.stock, st.sales, st.missed, a.shop_norm[s], ob);
      io.last_obs[g * 3 + 0] = ob[0]; io.last_obs[g * 3 + 1] = ob[1]; io.last_obs[g * 3 + 2] = ob[2];
    }
    if (s == 0) {       // every read of env_step / env_tick above is behind a barrier
      a.env_step[b] = step;
      a.env_tick[b] = a.env_tick[b] + a.T;
    }
  }
}

// ---- launchers ------------------------------------------------------------------------------------
hipError_t phx_launch_sc_step(const DevSpec& sp, const phx_step_io& io, hipStream_t st) {
  const int epb = SC_NT / sp.S;                       // whole envs per block (S <= 256 checked at create)
  const int blocks = (sp.B + epb - 1) / epb;
  const int64_t bytes = (int64_t)epb * sp.n_exo + 32;
  const int stage = (io.exo && bytes <= SC_STAGE_MAX) ? 1 : 0;
  hipLaunchKernelGGL(phx_sc_step_kernel, dim3(blocks), dim3(SC_NT), stage ? (size_t)bytes : 0, st,
                     sp, io, epb, stage);
  return hipGetLastError();
}

hipError_t phx_launch_sc_rollout(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st) {
  static const bool use_v1 = getenv("PHX_ROLLOUT_V1") != nullptr;     // A/B against the lane-per-shop loop
  if (use_v1) {
    const int epb = SC_NT / sp.S;
    hipLaunchKernelGGL(phx_sc_rollout_v1_kernel, dim3((sp.B + epb - 1) / epb), dim3(SC_NT), 0, st, sp, io, epb);
    return hipGetLastError();
  }
  // ~64 pairs per block (one wave in the sequential phase), TC steps so that the item table
  // stays around 36 KB; 1024-thread blocks put 16 time rows in flight per block
  RollArgs a;
  a.B = sp.B; a.S = sp.S; a.n_exo = sp.n_exo; a.num_steps = sp.num_steps; a.T = io.T;
  a.seed = sp.seed; a.env_offset = sp.env_offset;
  a.shop_norm = sp.shop_norm; a.shop_cust_ptr = sp.shop_cust_ptr; a.shop_cust_exo = sp.shop_cust_exo;
  a.stock = (int32_t*)sp.f[F_SHOP_STOCK]; a.sales = (int32_t*)sp.f[F_SHOP_SALES];
  a.missed = (int32_t*)sp.f[F_SHOP_MISSED]; a.delivered = (int32_t*)sp.f[F_SHOP_DELIVERED];
  a.env_step = (int32_t*)sp.f[F_ENV_STEP]; a.env_tick = (int32_t*)sp.f[F_ENV_TICK];
  a.io = io;
  int epb = 64 / sp.S; if (epb < 1) epb = 1; if (epb > sp.B) epb = sp.B;
  const int G = epb * sp.S;
  int TC = (36 * 1024) / (G * 12); if (TC < 1) TC = 1; if (TC > io.T) TC = io.T;
  a.epb = epb; a.TC = TC;
  const size_t lds = (size_t)G * TC * 12;
  static const int nt = getenv("PHX_ROLLOUT_NT") ? atoi(getenv("PHX_ROLLOUT_NT")) : 1024;
  const dim3 grid((sp.B + epb - 1) / epb);
  if (nt == 256) hipLaunchKernelGGL((phx_sc_rollout_kernel<256>), grid, dim3(256), lds, st, a);
  else if (nt == 512) hipLaunchKernelGGL((phx_sc_rollout_kernel<512>), grid, dim3(512), lds, st, a);
  else hipLaunchKernelGGL((phx_sc_rollout_kernel<1024>), grid, dim3(1024), lds, st, a);
  return hipGetLastError();
}

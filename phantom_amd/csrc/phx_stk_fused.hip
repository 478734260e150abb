// phx_stk_fused.hip -- fused static-schedule step for the Stackelberg market (SELLER / BUYER
// kinds on StackelbergEnv semantics, phantom/stackelberg.py:111-196).
//
// The schedule is static: a step has one non-empty round.  Leaders' step: every acting seller
// posts Price(price) along its CSR row; each buyer stores the price in the slot of that
// neighbour.  Followers' step: every buying buyer sends Order(1) to its cheapest neighbour
// (first minimum in neighbour order); a seller books revenue += price * vol once per order --
// all addends of a round are the same f64, so the sequential sum only needs the ORDER COUNT,
// which the block gets from LDS atomics.  One workgroup per env instance; the 8-byte price
// slots (buyer.prices f64[B][sum deg], 64 KB per env at 1024 x 8) dominate the traffic and are
// stored slot-major (ELL) so that the buyers of a wave read/write slot k at consecutive addresses.  Results are bit-identical to the generic engine.
#include "phx_dev.h"
#include "phx_epilogue.h"

#define STK_NT 256

__global__ __launch_bounds__(STK_NT) void phx_stk_step_kernel(const DevSpec sp, const phx_step_io io) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_nterm, s_ntrunc;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int A = sp.A, S = sp.S;
  const Topo tp = topo_global(sp);
  const int nSell = sp.kind_count[PHX_KIND_SELLER];
  double* s_price = (double*)smem;                       // [nSell] price posted this step
  int* s_count = (int*)(s_price + nSell);                // [nSell] orders received this step
  uint8_t* s_sent = (uint8_t*)(s_count + nSell);         // [nSell] seller broadcast a Price this step

  const int t = fld<int32_t>(sp, F_ENV_STEP)[b] + 1;                         // env.py:252
  const uint32_t tick = (uint32_t)fld<int32_t>(sp, F_ENV_TICK)[b];
  const int list = (t & 1) ? 0 : 1;                                          // stackelberg.py:133-137
  const uint8_t* act_mask = sp.act_mask + (int64_t)list * A;
  const float* actions_b = io.actions ? io.actions + (int64_t)b * S : nullptr;
  const uint8_t* av_b = io.action_valid ? io.action_valid + (int64_t)b * S : nullptr;
  double* prices_b = fld<double>(sp, F_BUYER_PRICES) + (int64_t)b * sp.buyer_nnz;

  if (tid == 0) { s_nterm = 0; s_ntrunc = 0; }
  for (int k = tid; k < nSell; k += STK_NT) { s_count[k] = 0; s_sent[k] = 0; }
  __syncthreads();

  // ---- acting phase (_handle_acting_agents, env.py:320-336): decode_action of every acting agent
  for (int a = tid; a < A; a += STK_NT) {
    if (!act_mask[a]) continue;
    const int s = sp.strat_rank[a];
    const bool has = actions_b && (!av_b || av_b[s]);                        // aid in actions
    if (!has) continue;
    const float action = actions_b[s];
    const AgentRef r = agent_ref(sp, tp, b, a);
    if (r.kind == PHX_KIND_SELLER) {
      const double price = (double)action;
      fld<double>(sp, F_SELLER_PRICE)[r.base] = price;
      s_price[r.kr] = price; s_sent[r.kr] = 1;
    } else {                                                                 // BUYER
      const int lo = sp.row_ptr[a], deg = sp.row_ptr[a + 1] - lo;
      if (action > 0.5f && deg > 0) {
        const double* pr = prices_b + sp.buyer_off[a];
        int j = 0; double best = pr[0];
        for (int k = 1; k < deg; ++k) { const double v = pr[(int64_t)k * sp.buyer_stride]; if (v < best) { best = v; j = k; } }
        fld<int32_t>(sp, F_BUYER_BOUGHT)[r.base] = 1;
        fld<double>(sp, F_BUYER_PAID)[r.base] = best;
        atomicAdd(&s_count[sp.kind_rank[sp.col[lo + j]]], 1);               // Order(1) -> that seller's inbox
      } else {
        fld<int32_t>(sp, F_BUYER_BOUGHT)[r.base] = 0;
        fld<double>(sp, F_BUYER_PAID)[r.base] = 0.0;
      }
    }
  }
  __syncthreads();
  // ---- pre_message_resolution + the single round -------------------------------------------------
  for (int a = tid; a < A; a += STK_NT) {
    const AgentRef r = agent_ref(sp, tp, b, a);
    if (r.kind == PHX_KIND_SELLER) {
      double rev = fld<double>(sp, F_SELLER_REVENUE)[r.base];
      int tx = fld<int32_t>(sp, F_SELLER_TX)[r.base];
      if ((t & 1) == 0) { rev = 0.0; tx = 0; }                               // start of a buying round
      const int n = s_count[r.kr];
      if (n > 0) {
        const double amount = __dmul_rn(fld<double>(sp, F_SELLER_PRICE)[r.base], 1.0);   // price * vol, vol = 1
        for (int k = 0; k < n; ++k) rev = __dadd_rn(rev, amount);             // one add per Order, inbox order
        tx += n;
      }
      fld<double>(sp, F_SELLER_REVENUE)[r.base] = rev;
      fld<int32_t>(sp, F_SELLER_TX)[r.base] = tx;
    } else {
      const int lo = sp.row_ptr[a], deg = sp.row_ptr[a + 1] - lo;
      double* pr = prices_b + sp.buyer_off[a];
      for (int k = 0; k < deg; ++k) {                                        // handle Price from each neighbour
        const int kr = sp.kind_rank[sp.col[lo + k]];
        if (s_sent[kr]) pr[(int64_t)k * sp.buyer_stride] = s_price[kr];
      }
    }
  }
  __syncthreads();
  strategic_epilogue<STK_NT>(sp, tp, io, b, t, list, 0, tick, nullptr, &s_nterm, &s_ntrunc);
}

hipError_t phx_launch_stk_step(const DevSpec& sp, const phx_step_io& io, hipStream_t st) {
  const int nSell = sp.kind_count[PHX_KIND_SELLER];
  const size_t lds = (size_t)nSell * (8 + 4 + 1) + 16;
  hipLaunchKernelGGL(phx_stk_step_kernel, dim3(sp.B), dim3(STK_NT), lds, st, sp, io);
  return hipGetLastError();
}

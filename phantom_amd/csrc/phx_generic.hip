// phx_generic.hip -- the generic message-passing engine: one workgroup per env instance.
//
// Replaces, for a whole batch per launch, the interpreter loops of
//   PhantomEnv.step / _handle_acting_agents / resolve_network   phantom/env.py:239-336
//   Network.send / resolve                                      phantom/network.py:233-265
//   BatchResolver.resolve (round loop)                          phantom/resolvers.py:128-163
//   Agent.handle_batch / handle_message                         phantom/agents.py:96-155
//   FiniteStateMachineEnv.step masks + reward cache             phantom/fsm.py:309-380
//   StackelbergEnv.step masks + reward cache                    phantom/stackelberg.py:133-196
//
// Layout of one round (the reference's insertion-ordered dict of lists, resolvers.py:120/142,
// rebuilt without any sequential walk over the queue):
//   * the round's messages sit in a dense queue in SEND order (q[i], i = send sequence number);
//   * LDS atomics give every message its slot in the receiver's inbox and every receiver its
//     first-arrival index; an exclusive block scan over "head" messages turns first-arrival
//     order into inbox offsets, so inbox order[] holds receivers in dict order and, after a
//     short per-receiver sort by sequence number, each batch in send order;
//   * one lane per receiver walks its batch (handlers are sequential in the reference too) and
//     leaves at most one response per message in processing order; a second block scan
//     compacts the responses into the next round's queue = the order network.send was called.
// Queues live in LDS (template LDSQ = true); envs whose queue does not fit use a per-env
// workspace carved from the caller's state blob.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "phx_dev.h"
#include "phx_epilogue.h"

#define ERRKEY_NONE 0x7fffffff


// inclusive scan over the 64 lanes of a wave with DPP row shifts / broadcasts: ~80 cycles against ~400 for the six
// dependent ds_bpermute round trips of a __shfl_up ladder (tools/ubench/ub_dppscan.hip checks and times both)
__device__ __forceinline__ int wave_incl_scan(int v) {
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return x;
}

template <int NT>
__device__ __forceinline__ int block_exscan(int* arr, int n, int* wave_sums) {
  // in-place exclusive scan of arr[0..n); returns the total.  All NT threads must call.
  const int tid = threadIdx.x;
  const int chunk = (n + NT - 1) / NT;
  const int lo = min(tid * chunk, n), hi = min(lo + chunk, n);
  int local = 0;
  for (int i = lo; i < hi; ++i) local += arr[i];
  const int incl = wave_incl_scan(local);
  if ((tid & 63) == 63) wave_sums[tid >> 6] = incl;
  __syncthreads();
  int wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const int s = wave_sums[w];
    if (w < (tid >> 6)) wave_off += s;
    total += s;
  }
  int running = wave_off + incl - local;
  for (int i = lo; i < hi; ++i) {
    const int v = arr[i];
    arr[i] = running;
    running += v;
  }
  __syncthreads();
  return total;
}

__device__ __forceinline__ void set_errkey(int* errkey, int seq, int code) {
  atomicMin(errkey, (seq << 4) | code);
}

// ---- acting phase: decode_action (env.py:330-331) / generate_messages (:332-333) -------------
__device__ __forceinline__ int act_count(const DevSpec& sp, const Topo& tp, int b, int a, bool has_action, float action) {
  switch (tkind(tp, a)) {
    case PHX_KIND_SHOP: return has_action ? 1 : 0;
    case PHX_KIND_CUSTOMER: return 1;
    case PHX_KIND_SELLER: {
      if (!has_action) return 0;
      if (tp.conn_on == nullptr) return tp.row_ptr[a + 1] - tp.row_ptr[a];
      int n = 0;
      for (int k = tp.row_ptr[a]; k < tp.row_ptr[a + 1]; ++k) n += edge_on(tp, k) ? 1 : 0;
      return n;
    }
    case PHX_KIND_BUYER: {
      if (!(has_action && action > 0.5f)) return 0;
      for (int k = tp.row_ptr[a]; k < tp.row_ptr[a + 1]; ++k) if (edge_on(tp, k)) return 1;
      return 0;
    }
    case PHX_KIND_PUBLISHER: return 1;
    case PHX_KIND_ADVERTISER: {                                // a bid is sent iff it is > 0 (digital_ads_market.py:328)
      if (!has_action) return 0;
      const int64_t base = (int64_t)b * sp.kind_count[PHX_KIND_ADVERTISER] + tp.kind_rank[a];
      const TVal left = tv(fld<double>(sp, F_ADV_LEFT)[base], fld<int32_t>(sp, F_ADV_LEFT_TAG)[base]);
      TVal bid = t_mul(tv((double)action, PHX_TAG_F32), adv_budget(sp, tp, b, a));
      if (t_lt(left, bid)) bid = left;
      return bid.v > 0.0 ? 1 : 0;
    }
    default: return 0;
  }
}

template <typename Q>
__device__ __forceinline__ void act_emit(const DevSpec& sp, const Topo& tp, int b, int a, bool has_action, float action,
                                         const uint8_t* exo_b, uint32_t tick, Q* out) {
  const AgentRef r = agent_ref(sp, tp, b, a);
  const int32_t* pi = tp.param_i + a * PHX_NPI;
  DevMsg m;
  m.src = (uint16_t)a; m.pad = 0;
  switch (r.kind) {
    case PHX_KIND_SHOP:
      if (has_action) {                                        // supply_chain.py:136-142
        const int req = dev_round_half_even(action);
        const int room = PHX_SHOP_MAX_STOCK - fld<int32_t>(sp, F_SHOP_STOCK)[r.base];
        m.dst = (uint16_t)pi[0]; m.type = PHX_MSG_STOCK_REQUEST; m.p.i = req < room ? req : room;
        out[0] = m;
      }
      break;
    case PHX_KIND_CUSTOMER: {                                  // supply_chain.py:61-67
      int order;
      if (exo_b) order = exo_b[tp.exo_rank[a]];
      else order = rng_shop_orders(sp.seed, sp.env_offset + b, tick, tp.kind_rank[pi[0]], pi[1] + 1,
                                   nullptr, pi[1]);
      m.dst = (uint16_t)pi[0]; m.type = PHX_MSG_ORDER_REQUEST; m.p.i = order;
      out[0] = m;
      break;
    }
    case PHX_KIND_SELLER:
      if (has_action) {
        const double price = (double)action;
        fld<double>(sp, F_SELLER_PRICE)[r.base] = price;
        m.type = PHX_MSG_PRICE; m.p.f = price;
        const int lo = tp.row_ptr[a], hi = tp.row_ptr[a + 1];
        int n = 0;
        for (int k = lo; k < hi; ++k) if (edge_on(tp, k)) { m.dst = (uint16_t)tp.col[k]; out[n++] = m; }
      }
      break;
    case PHX_KIND_BUYER:
      if (has_action) {
        const int lo = tp.row_ptr[a], deg = tp.row_ptr[a + 1] - lo;
        const double* pr = fld<double>(sp, F_BUYER_PRICES) + (int64_t)b * sp.buyer_nnz + tp.buyer_off[a];
        int j = -1; double best = 0.0;                         // first minimum over the current neighbours
        for (int k = 0; k < deg; ++k)
          if (edge_on(tp, lo + k)) { const double v = pr[(int64_t)k * sp.buyer_stride]; if (j < 0 || v < best) { best = v; j = k; } }
        if (action > 0.5f && j >= 0) {
          fld<int32_t>(sp, F_BUYER_BOUGHT)[r.base] = 1;
          fld<double>(sp, F_BUYER_PAID)[r.base] = best;
          m.dst = (uint16_t)tp.col[tp.row_ptr[a] + j]; m.type = PHX_MSG_ORDER; m.p.i = 1;
          out[0] = m;
        } else {
          fld<int32_t>(sp, F_BUYER_BOUGHT)[r.base] = 0;
          fld<double>(sp, F_BUYER_PAID)[r.base] = 0.0;
        }
      }
      break;
    case PHX_KIND_MOCK_STRAT:
      if (has_action) fld<int32_t>(sp, F_MOCK_DEC)[r.base] += 1;     // tests/__init__.py:53-55
      break;
    case PHX_KIND_PUBLISHER: {                                 // digital_ads_market.py:164-165, :48-53
      const int user = exo_b ? exo_b[tp.exo_rank[a]] : rng_publisher(sp.seed, sp.env_offset + b, tick, a, 0, 0.0);
      m.dst = (uint16_t)pi[0]; m.type = PHX_MSG_IMPRESSION_REQ; m.p.i = user;
      out[0] = m;
      break;
    }
    case PHX_KIND_ADVERTISER:
      if (has_action) {                                        // digital_ads_market.py:318-333
        const TVal left = tv(fld<double>(sp, F_ADV_LEFT)[r.base], fld<int32_t>(sp, F_ADV_LEFT_TAG)[r.base]);
        TVal bid = t_mul(tv((double)action, PHX_TAG_F32), adv_budget(sp, tp, b, a));
        if (t_lt(left, bid)) bid = left;                       // min(action[0] * budget, self.left)
        fld<double>(sp, F_ADV_BID)[r.base] = bid.v; fld<int32_t>(sp, F_ADV_BID_TAG)[r.base] = bid.tag;
        if (bid.v > 0.0) {
          m.dst = (uint16_t)pi[0]; m.type = PHX_MSG_BID; m.p.f = bid.v;
          m.pad = (uint16_t)(pi[1] | (fld<int32_t>(sp, F_ADV_USER)[r.base] << 4) | (bid.tag << 8));
          out[0] = m;
        }
      }
      break;
    default: break;
  }
}

__device__ __forceinline__ void pre_resolution(const DevSpec& sp, const Topo& tp, int b, int a, int step) {
  const AgentRef r = agent_ref(sp, tp, b, a);
  switch (r.kind) {
    case PHX_KIND_SHOP:                                        // supply_chain.py:93-96
      fld<int32_t>(sp, F_SHOP_SALES)[r.base] = 0;
      fld<int32_t>(sp, F_SHOP_MISSED)[r.base] = 0;
      break;
    case PHX_KIND_SELLER:
      if ((step & 1) == 0) {
        fld<double>(sp, F_SELLER_REVENUE)[r.base] = 0.0;
        fld<int32_t>(sp, F_SELLER_TX)[r.base] = 0;
      }
      break;
    case PHX_KIND_ADVERTISER:                                  // digital_ads_market.py:241-247
      fld<int32_t>(sp, F_ADV_CLICKS)[r.base] = 0; fld<int32_t>(sp, F_ADV_WINS)[r.base] = 0;
      break;
    case PHX_KIND_PUBLISHER: fld<int32_t>(sp, F_PUB_ADS_SEEN)[r.base] = 0; break;
    default: break;
  }
}

// The mutable attributes a receiver's handlers touch, held in registers while its lane walks the batch: every
// message of a shop's inbox used to be three or four DEPENDENT read-modify-write round trips to the state blob
// (stock, missed_sales, sales; a load after a store to the same address waits for the store), ~20 k of the
// ~80 k cycles of an SC64 env-step.  Kinds with larger state (advertisers, buyers' price tables) stay in the blob.
struct AgentState { int32_t i0, i1, i2, i3; double f0, f1; };
__device__ __forceinline__ bool kind_state_cached(int kind) {
  return kind == PHX_KIND_SHOP || kind == PHX_KIND_SELLER || kind == PHX_KIND_CASHBOX || kind == PHX_KIND_REQRESP;
}
__device__ __forceinline__ void load_state(const DevSpec& sp, const AgentRef& r, AgentState& st) {
  switch (r.kind) {
    case PHX_KIND_SHOP:
      st.i0 = fld<int32_t>(sp, F_SHOP_STOCK)[r.base]; st.i1 = fld<int32_t>(sp, F_SHOP_SALES)[r.base];
      st.i2 = fld<int32_t>(sp, F_SHOP_MISSED)[r.base]; st.i3 = fld<int32_t>(sp, F_SHOP_DELIVERED)[r.base]; break;
    case PHX_KIND_SELLER:
      st.f0 = fld<double>(sp, F_SELLER_PRICE)[r.base]; st.f1 = fld<double>(sp, F_SELLER_REVENUE)[r.base];
      st.i0 = fld<int32_t>(sp, F_SELLER_TX)[r.base]; break;
    case PHX_KIND_CASHBOX: st.f0 = fld<double>(sp, F_CASHBOX_TOTAL)[r.base]; break;
    case PHX_KIND_REQRESP: st.i0 = fld<int32_t>(sp, F_REQRESP_REQ)[r.base]; st.i1 = fld<int32_t>(sp, F_REQRESP_RES)[r.base]; break;
    default: break;
  }
}
__device__ __forceinline__ void store_state(const DevSpec& sp, const AgentRef& r, const AgentState& st) {
  switch (r.kind) {
    case PHX_KIND_SHOP:
      fld<int32_t>(sp, F_SHOP_STOCK)[r.base] = st.i0; fld<int32_t>(sp, F_SHOP_SALES)[r.base] = st.i1;
      fld<int32_t>(sp, F_SHOP_MISSED)[r.base] = st.i2; fld<int32_t>(sp, F_SHOP_DELIVERED)[r.base] = st.i3; break;
    case PHX_KIND_SELLER:
      fld<double>(sp, F_SELLER_REVENUE)[r.base] = st.f1; fld<int32_t>(sp, F_SELLER_TX)[r.base] = st.i0; break;
    case PHX_KIND_CASHBOX: fld<double>(sp, F_CASHBOX_TOTAL)[r.base] = st.f0; break;
    case PHX_KIND_REQRESP: fld<int32_t>(sp, F_REQRESP_REQ)[r.base] = st.i0; fld<int32_t>(sp, F_REQRESP_RES)[r.base] = st.i1; break;
    default: break;
  }
}
// receivers whose handlers neither read nor write agent state: their messages are handled one lane per MESSAGE
// (a factory's 51 StockRequests in SC256 were one lane's sequential chain, the longest of the step)
__device__ __forceinline__ bool kind_stateless(int kind) {
  return kind == PHX_KIND_FACTORY || kind == PHX_KIND_CUSTOMER || kind == PHX_KIND_HALVER || kind == PHX_KIND_FORWARDER ||
         kind == PHX_KIND_MOCK_STRAT || kind == PHX_KIND_MOCK_AGENT;
}

// Agent.handle_message (agents.py:122-155): returns true and fills `resp` (dst/type/payload)
// when the handler answers; sets `code` to PHX_ERR_UNKNOWN_MSG for an unhandled payload type.
// `st`: the receiver's register-cached state (kind_state_cached kinds); `r`: agent_ref of the receiver.
__device__ __forceinline__ bool handle_message(const DevSpec& sp, const Topo& tp, int b, int a, const AgentRef& r, const DevMsg& m,
                                               int clock, const uint8_t* exo_b, uint32_t tick, DevMsg& resp, int& code,
                                               AgentState& st) {
  const int32_t* pi = tp.param_i + a * PHX_NPI;
  resp.src = (uint16_t)a; resp.dst = m.src; resp.pad = 0; resp.type = 0;
  switch (r.kind) {
    case PHX_KIND_FACTORY:
      if (m.type == PHX_MSG_STOCK_REQUEST) {                   // supply_chain.py:40-45
        resp.type = PHX_MSG_STOCK_RESPONSE; resp.p.i = m.p.i; return true;
      }
      break;
    case PHX_KIND_SHOP: {                                      // st: i0 stock, i1 sales, i2 missed_sales, i3 delivered_stock
      if (m.type == PHX_MSG_STOCK_RESPONSE) {                  // supply_chain.py:98-103
        st.i3 = (int32_t)m.p.i;
        const int ns = st.i0 + (int32_t)m.p.i;
        st.i0 = ns < PHX_SHOP_MAX_STOCK ? ns : PHX_SHOP_MAX_STOCK;
        return false;
      }
      if (m.type == PHX_MSG_ORDER_REQUEST) {                   // supply_chain.py:105-122
        const int req = (int)m.p.i;
        int sell;
        if (req > st.i0) {
          st.i2 += req - st.i0;
          sell = st.i0; st.i0 = 0;
        } else { sell = req; st.i0 -= req; }
        st.i1 += sell;
        resp.type = PHX_MSG_ORDER_RESPONSE; resp.p.i = sell; return true;
      }
      break;
    }
    case PHX_KIND_CUSTOMER:
      if (m.type == PHX_MSG_ORDER_RESPONSE) return false;      // supply_chain.py:55-59
      break;
    case PHX_KIND_SELLER:                                      // st: f0 price, f1 revenue, i0 tx
      if (m.type == PHX_MSG_ORDER) {
        const double amount = __dmul_rn(st.f0, (double)m.p.i);
        st.f1 = __dadd_rn(st.f1, amount);
        st.i0 += (int32_t)m.p.i;
        return false;
      }
      break;
    case PHX_KIND_BUYER:
      if (m.type == PHX_MSG_PRICE) {
        const int slot = dev_nbr_slot(sp, tp, a, m.src);
        if (slot >= 0)
          fld<double>(sp, F_BUYER_PRICES)[(int64_t)b * sp.buyer_nnz + tp.buyer_off[a] + (int64_t)slot * sp.buyer_stride] = m.p.f;
        return false;
      }
      break;
    case PHX_KIND_HALVER:
      if (m.type == PHX_MSG_HALVE) {                           // test_tracking.py:22-27
        if (m.p.i > 1) { resp.type = PHX_MSG_HALVE; resp.p.i = m.p.i / 2; return true; }
        return false;
      }
      break;
    case PHX_KIND_CASHBOX:
      if (m.type == PHX_MSG_CASH) {                            // test_network.py:26-34
        if (m.p.f > 25) {
          st.f0 += m.p.f / 2.0;
          resp.type = PHX_MSG_CASH; resp.p.f = m.p.f / 2.0; return true;
        }
        return false;
      }
      break;
    case PHX_KIND_REQRESP:
      if (m.type == PHX_MSG_REQUEST) {                         // test_resolver.py:31-37
        st.i0 = clock;
        resp.type = PHX_MSG_RESPONSE; resp.p.f = m.p.f / 2.0; return true;
      }
      if (m.type == PHX_MSG_RESPONSE) { st.i1 = clock; return false; }
      break;
    case PHX_KIND_FORWARDER:                                   // test_resolver.py:93-96
      if (pi[0] >= 0) { resp.dst = (uint16_t)pi[0]; resp.type = PHX_MSG_PING; resp.p.i = 1; return true; }
      return false;
    case PHX_KIND_PUBLISHER:
      if (m.type == PHX_MSG_ADS) {                             // digital_ads_market.py:167-196
        const int theme = m.pad & 15, user = (m.pad >> 4) & 15;
        if (user < 1 || user > 2 || theme > 3) { code = PHX_ERR_CONTEXT; return false; }   // dict KeyError :194
        const double p = sp.param_f[a * PHX_NPF + (user - 1) * 4 + theme];
        const int k = ++fld<int32_t>(sp, F_PUB_ADS_SEEN)[r.base];
        int clicked;
        if (exo_b) {
          if (k > pi[1]) { code = PHX_ERR_QUEUE_FULL; return false; }
          clicked = exo_b[tp.exo_rank[a] + k];
        } else clicked = rng_publisher(sp.seed, sp.env_offset + b, tick, a, k, p);
        resp.dst = (uint16_t)m.p.i; resp.type = PHX_MSG_IMPRESSION_RES; resp.p.i = clicked; return true;
      }
      break;
    case PHX_KIND_ADVERTISER:
      if (m.type == PHX_MSG_IMPRESSION_REQ) {                  // :249-271
        fld<int32_t>(sp, F_ADV_USER)[r.base] = (int32_t)m.p.i;
        if (!dev_has_edge(sp, tp, a, pi[0])) { code = PHX_ERR_CONTEXT; return false; }     // ctx[self.exchange_id] :261
        if (m.p.i >= 0 && m.p.i <= 2) fld<int32_t>(sp, F_ADV_TOT_REQUESTS)[r.base * 3 + m.p.i] += 1;
        return false;
      }
      if (m.type == PHX_MSG_AUCTION_RESULT) {                  // :273-282
        const int won = m.p.f != 0.0 ? 1 : 0;
        fld<int32_t>(sp, F_ADV_WINS)[r.base] += won;
        fld<int32_t>(sp, F_ADV_TOT_WINS)[r.base * 3 + fld<int32_t>(sp, F_ADV_USER)[r.base]] += won;
        const TVal left = t_sub(tv(fld<double>(sp, F_ADV_LEFT)[r.base], fld<int32_t>(sp, F_ADV_LEFT_TAG)[r.base]),
                                tv(m.p.f, (m.pad >> 8) & 3));
        fld<double>(sp, F_ADV_LEFT)[r.base] = left.v; fld<int32_t>(sp, F_ADV_LEFT_TAG)[r.base] = left.tag;
        return false;
      }
      if (m.type == PHX_MSG_IMPRESSION_RES) {                  // :284-292
        fld<int32_t>(sp, F_ADV_CLICKS)[r.base] += (int32_t)m.p.i;
        fld<int32_t>(sp, F_ADV_TOT_CLICKS)[r.base * 3 + fld<int32_t>(sp, F_ADV_USER)[r.base]] += (int32_t)m.p.i;
        return false;
      }
      break;
    default: break;
  }
  code = PHX_ERR_UNKNOWN_MSG;                                  // agents.py:140-143
  return false;
}

// ---- AdExchangeAgent.handle_batch (digital_ads_market.py:429-516): an inbox REDUCTION ---------------
// The exchange's lane walks its batch twice.  Pass `emit == false` counts what each position sends
// (an ImpressionRequest fans out to every AdvertiserAgent neighbour, :427; the auction's Ads +
// AuctionResults are booked on the batch's last position, after everything handle_message sent,
// :443-449) and reports errors; after the block scan, pass `emit == true` writes the same messages at
// their queue offsets.  The exchange keeps no state, so both passes see the same batch.
//   ok(k)  : message k of the batch is delivered (receiver live, edge present, send succeeded)
template <typename OkFn>
__device__ __forceinline__ void adexchange_batch(const DevSpec& sp, const Topo& tp, int a, const DevMsg* qc, const int* seg,
                                                 int c, OkFn ok, bool emit, int* counts, const int* offs, DevMsg* qn,
                                                 int* errkey, int seq0) {
  const int32_t* pi = tp.param_i + a * PHX_NPI;
  int nb = 0, w = -1, w2 = -1, last_own = 0;
  for (int k = 0; k < c; ++k) {
    if (!emit) counts[k] = 0;
    if (!ok(k)) continue;
    const DevMsg m = qc[seg[k]];
    if (m.type == PHX_MSG_BID) {
      ++nb;
      if (w < 0 || t_lt(tv(qc[seg[w]].p.f, (qc[seg[w]].pad >> 8) & 3), tv(m.p.f, (m.pad >> 8) & 3))) w = k;
      continue;
    }
    if (m.type != PHX_MSG_IMPRESSION_REQ) {                    // no handler: ValueError agents.py:140-143
      if (!emit) set_errkey(errkey, seq0 + k, PHX_ERR_UNKNOWN_MSG);
      continue;
    }
    int n = 0;
    for (int e = tp.row_ptr[a]; e < tp.row_ptr[a + 1]; ++e) {
      const int dst = tp.col[e];
      if (tkind(tp, dst) != PHX_KIND_ADVERTISER) continue;
      const int sc = dev_send_check(sp, tp, a, dst, PHX_MSG_IMPRESSION_REQ);
      if (sc) { if (!emit) set_errkey(errkey, seq0 + k, sc); continue; }
      if (emit) {
        DevMsg o; o.src = (uint16_t)a; o.dst = (uint16_t)dst; o.type = PHX_MSG_IMPRESSION_REQ; o.pad = 0; o.p.i = m.p.i;
        qn[offs[k] + n] = o;
      }
      ++n;
    }
    if (!emit) counts[k] = n;
    if (k == c - 1) last_own = n;
  }
  if (!nb) return;
  for (int k = 0; k < c; ++k) {                                // sorted_bids[1]: first maximum among the rest
    if (k == w || !ok(k) || qc[seg[k]].type != PHX_MSG_BID) continue;
    if (w2 < 0 || t_lt(tv(qc[seg[w2]].p.f, (qc[seg[w2]].pad >> 8) & 3), tv(qc[seg[k]].p.f, (qc[seg[k]].pad >> 8) & 3))) w2 = k;
  }
  const DevMsg win = qc[seg[w]];
  const DevMsg costm = (pi[1] && w2 >= 0) ? qc[seg[w2]] : win;          // second / first price :498-516
  int n = last_own;                                            // the auction's sends follow the last position's own
  int off = emit ? offs[c - 1] + last_own : 0;
  {                                                            // Ads(advertiser_id, theme, user_id) :472-482
    const int sc = dev_send_check(sp, tp, a, pi[0], PHX_MSG_ADS);
    if (sc) { if (!emit) set_errkey(errkey, seq0 + c - 1, sc); }
    else {
      if (emit) {
        DevMsg o; o.src = (uint16_t)a; o.dst = (uint16_t)pi[0]; o.type = PHX_MSG_ADS; o.pad = win.pad & 0xff; o.p.i = win.src;
        qn[off] = o;
      }
      ++off; ++n;
    }
  }
  for (int k = 0; k < c; ++k) {                                // AuctionResult to every bidder, bid order :484-492
    if (!ok(k) || qc[seg[k]].type != PHX_MSG_BID) continue;
    const int dst = qc[seg[k]].src;
    const int sc = dev_send_check(sp, tp, a, dst, PHX_MSG_AUCTION_RESULT);
    if (sc) { if (!emit) set_errkey(errkey, seq0 + c - 1, sc); continue; }
    if (emit) {
      DevMsg o; o.src = (uint16_t)a; o.dst = (uint16_t)dst; o.type = PHX_MSG_AUCTION_RESULT;
      if (dst == win.src) { o.pad = costm.pad & 0x300; o.p.f = costm.p.f; }
      else { o.pad = PHX_TAG_PYF << 8; o.p.f = 0.0; }
      qn[off] = o;
    }
    ++off; ++n;
  }
  if (!emit) counts[c - 1] = n;
}

// ---- the same reduction, by the whole workgroup ------------------------------------------------------
// The single lane above is the critical path of an ads-market step (a 120-bid auction costs it ~10^6
// cycles); here every thread takes batch positions k = tid, tid + NT, ...  Phase A (before the
// receivers' lanes run; the batch is in send order already): classify every message, count what each
// position sends and reduce the winner; phase B (after the block scan): write the messages.  Sends are
// booked in position order, which is the reference's order unless an ImpressionRequest follows a Bid
// in the batch -- then `mode[a]` = -1 and lane `a` runs the sequential adexchange_batch instead.
//   cls[k] (kept in the round's dead slot[] array): 0 dropped, 1 Bid, 2 ImpressionRequest, 3 no handler,
//   | 16 the first bid of the batch; mode[a] = winner position | (second + 1) << 16.
__device__ __forceinline__ TVal adx_bid(const DevMsg* qc, const int* seg, int k) {
  const DevMsg m = qc[seg[k]];
  return tv(m.p.f, (m.pad >> 8) & 3);
}
// first maximum in position order: of two candidates the earlier one wins unless it is smaller
__device__ __forceinline__ int adx_better(const DevMsg* qc, const int* seg, int x, int y) {
  if (x < 0) return y;
  if (y < 0) return x;
  const int lo = x < y ? x : y, hi = x < y ? y : x;
  return t_lt(adx_bid(qc, seg, lo), adx_bid(qc, seg, hi)) ? hi : lo;
}
template <int NT, typename F>
__device__ __forceinline__ int block_fold(int v, int* red, F combine) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) v = combine(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = combine(r, red[w]);
  return r;
}
__device__ __forceinline__ bool adx_delivered(const DevSpec& sp, const Topo& tp, const uint8_t* live, int a, const DevMsg& m) {
  return live[a] && m.type != 0 && (!(sp.flags & PHX_F_IGNORE_CONN_ERRORS) || dev_has_edge(sp, tp, m.src, m.dst));
}
// every send of the fan-out passes Network.send's checks (no per-entry work, parallel emission)
__device__ __forceinline__ bool adx_fanout_all(const DevSpec& sp) {
  return (sp.flags & PHX_F_IGNORE_CONN_ERRORS) || !sp.dynamic_graph;
}
__device__ __forceinline__ int adx_fanout(const DevSpec& sp, const Topo& tp, int x, int a, const DevMsg& m, DevMsg* out, int* errkey, int key) {
  const int lo = sp.adx_nbr_ptr[x], hi = sp.adx_nbr_ptr[x + 1];
  if (!out && adx_fanout_all(sp)) return hi - lo;              // forward to self.advertiser_ids :427
  int n = 0;
  for (int j = lo; j < hi; ++j) {
    const int e = sp.adx_nbr_e[j];
    if (!(sp.flags & PHX_F_IGNORE_CONN_ERRORS) && !edge_on(tp, e)) { if (!out) set_errkey(errkey, key, PHX_ERR_NETWORK); continue; }
    if (out) { DevMsg o; o.src = (uint16_t)a; o.dst = (uint16_t)tp.col[e]; o.type = PHX_MSG_IMPRESSION_REQ; o.pad = 0; o.p.i = m.p.i; out[n] = o; }
    ++n;
  }
  return n;
}

template <int NT>
__device__ __forceinline__ void adx_coop_count(const DevSpec& sp, const Topo& tp, int x, int a, const uint8_t* live, const DevMsg* qc,
                                               int* seg, int c, int* cls, int* counts, DevMsg* resp, int* red, int* mode,
                                               int* errkey, int seq0) {
  const int tid = threadIdx.x;
  int first_bid = 0x7fffffff, last_req = -1, best = -1;
  for (int k = tid; k < c; k += NT) {
    const DevMsg m = qc[seg[k]];
    int cl = 0, n = 0;
    if (adx_delivered(sp, tp, live, a, m)) {
      if (m.type == PHX_MSG_BID) { cl = 1; first_bid = min(first_bid, k); best = adx_better(qc, seg, best, k); }
      else if (m.type == PHX_MSG_IMPRESSION_REQ) { cl = 2; last_req = k; n = adx_fanout(sp, tp, x, a, m, nullptr, errkey, seq0 + k); }
      else { cl = 3; set_errkey(errkey, seq0 + k, PHX_ERR_UNKNOWN_MSG); }          // agents.py:140-143
    }
    cls[k] = cl; counts[k] = n; resp[k].type = 0;
  }
  first_bid = block_fold<NT>(first_bid, red, [](int x, int y) { return x < y ? x : y; });
  last_req = block_fold<NT>(last_req, red, [](int x, int y) { return x > y ? x : y; });
  if (first_bid == 0x7fffffff) { if (tid == 0) mode[a] = 0; __syncthreads(); return; }
  if (last_req > first_bid) { if (tid == 0) mode[a] = -1; __syncthreads(); return; }   // sequential fallback
  const int w = block_fold<NT>(best, red, [&](int x, int y) { return adx_better(qc, seg, x, y); });
  int best2 = -1;
  for (int k = tid; k < c; k += NT) if (k != w && (cls[k] & 15) == 1) best2 = adx_better(qc, seg, best2, k);
  const int w2 = block_fold<NT>(best2, red, [&](int x, int y) { return adx_better(qc, seg, x, y); });
  const int32_t* pi = tp.param_i + a * PHX_NPI;
  for (int k = tid; k < c; k += NT) {
    if ((cls[k] & 15) != 1) continue;
    int n = 0;
    if (k == first_bid) {                                      // Ads to the publisher :472-482
      cls[k] |= 16;
      const int sc = dev_send_check(sp, tp, a, pi[0], PHX_MSG_ADS);
      if (sc) set_errkey(errkey, seq0 + c - 1, sc); else ++n;
    }
    const int sc = dev_send_check(sp, tp, a, qc[seg[k]].src, PHX_MSG_AUCTION_RESULT);   // :484-492
    if (sc) set_errkey(errkey, seq0 + c - 1, sc); else ++n;
    counts[k] = n;
  }
  if (tid == 0) mode[a] = w | ((w2 + 1) << 16);
  __syncthreads();
}

template <int NT>
__device__ __forceinline__ void adx_coop_emit(const DevSpec& sp, const Topo& tp, int x, int a, const DevMsg* qc, const int* seg, int c,
                                              const int* cls, const int* offs, DevMsg* qn, int md) {
  const int32_t* pi = tp.param_i + a * PHX_NPI;
  const int w = md & 0xffff, w2 = (md >> 16) - 1;
  for (int k = threadIdx.x; k < c; k += NT) {
    const int cl = cls[k] & 15;
    const DevMsg m = qc[seg[k]];
    if (cl == 2) { if (!adx_fanout_all(sp)) adx_fanout(sp, tp, x, a, m, qn + offs[k], nullptr, 0); continue; }
    if (cl != 1) continue;
    const DevMsg win = qc[seg[w]];
    const DevMsg costm = (pi[1] && w2 >= 0) ? qc[seg[w2]] : win;        // second / first price :498-516
    int off = offs[k];
    if ((cls[k] & 16) && dev_send_check(sp, tp, a, pi[0], PHX_MSG_ADS) == 0) {
      DevMsg o; o.src = (uint16_t)a; o.dst = (uint16_t)pi[0]; o.type = PHX_MSG_ADS; o.pad = win.pad & 0xff; o.p.i = win.src;
      qn[off++] = o;
    }
    if (dev_send_check(sp, tp, a, m.src, PHX_MSG_AUCTION_RESULT) == 0) {
      DevMsg o; o.src = (uint16_t)a; o.dst = m.src; o.type = PHX_MSG_AUCTION_RESULT;
      if (m.src == win.src) { o.pad = costm.pad & 0x300; o.p.f = costm.p.f; }
      else { o.pad = PHX_TAG_PYF << 8; o.p.f = 0.0; }
      qn[off] = o;
    }
  }
  if (adx_fanout_all(sp) && (md >> 16) == 0 && (md & 0xffff) == 0) {
    // no auction ran (requests only -- mode 0): the fan-outs are written by all threads, entry-parallel.
    // (mode 0 is also "one bid at position 0", which has no requests after it: the loop finds none)
    const int lo = sp.adx_nbr_ptr[x], nn = sp.adx_nbr_ptr[x + 1] - lo;
    for (int k = 0; k < c; ++k) {
      if ((cls[k] & 15) != 2) continue;
      const DevMsg m = qc[seg[k]];
      for (int j = threadIdx.x; j < nn; j += NT) {
        DevMsg o; o.src = (uint16_t)a; o.dst = (uint16_t)tp.col[sp.adx_nbr_e[lo + j]]; o.type = PHX_MSG_IMPRESSION_REQ; o.pad = 0; o.p.i = m.p.i;
        qn[offs[k] + j] = o;
      }
    }
  } else if (adx_fanout_all(sp)) {                             // requests ahead of an auction: each by its own thread
    for (int k = threadIdx.x; k < c; k += NT)
      if ((cls[k] & 15) == 2) {
        const DevMsg m = qc[seg[k]];
        const int lo = sp.adx_nbr_ptr[x], nn = sp.adx_nbr_ptr[x + 1] - lo;
        for (int j = 0; j < nn; ++j) {
          DevMsg o; o.src = (uint16_t)a; o.dst = (uint16_t)tp.col[sp.adx_nbr_e[lo + j]]; o.type = PHX_MSG_IMPRESSION_REQ; o.pad = 0; o.p.i = m.p.i;
          qn[offs[k] + j] = o;
        }
      }
  }
}

// ---- PhantomEnv.reset of env b by its workgroup (env.py:185-237; fsm.py:195-251; stackelberg.py:53-109):
// the body of phx_reset_kernel, also the tail of a launch-loop rollout step whose episode ended.
// Contains __syncthreads(): call it from uniform control flow.
template <int NT>
__device__ __forceinline__ void reset_env(const DevSpec& sp, const int b, const double* sampler_values, const uint8_t* conn_values,
                                          float* obs, uint8_t* obs_valid, const int kmax = PHX_KIND_COUNT - 1) {
  const int tid = threadIdx.x;
  const int A = sp.A, S = sp.S, D = sp.D;
  Topo tp = topo_env(sp, b);
  tp.kmax = kmax;
  if (sp.n_samplers > 0 || sp.n_conn > 0) {                             // env.py:211-218
    const uint32_t ep = (uint32_t)fld<int32_t>(sp, F_ENV_EPISODE)[b];
    if (sp.n_samplers > 0) {
      double* sv = fld<double>(sp, F_ENV_SAMPLER) + (int64_t)b * sp.n_samplers;
      for (int j = tid; j < sp.n_samplers; j += NT) sv[j] = dev_sample_column(sp, b, j, ep, sampler_values, sv[j]);
    }
    if (sp.n_conn > 0) {                                                // resample_connectivity network.py:438-447
      uint8_t* cv = fld<uint8_t>(sp, F_NET_CONN_ON) + (int64_t)b * sp.n_conn;
      for (int i = tid; i < sp.n_conn; i += NT)
        cv[i] = conn_values ? (conn_values[(int64_t)b * sp.n_conn + i] != 0)
                            : (uint8_t)rng_connection(sp.seed, sp.env_offset + b, ep, i, sp.conn_rate[i]);
    }
    __syncthreads();
    if (tid == 0) fld<int32_t>(sp, F_ENV_EPISODE)[b] = (int32_t)(ep + 1);
  }
  for (int a = tid; a < A; a += NT) dev_agent_reset(sp, tp, b, a);          // network.py:183-184
  for (int s = tid; s < S; s += NT) {
    fld<uint8_t>(sp, F_ENV_TERM)[(int64_t)b * S + s] = 0;               // env.py:223-224
    fld<uint8_t>(sp, F_ENV_TRUNC)[(int64_t)b * S + s] = 0;
    if (sp.env_type != PHX_ENV_PLAIN) fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[(int64_t)b * S + s] = 0;
    if (obs_valid) obs_valid[(int64_t)b * S + s] = 0;
    if (obs) for (int d = 0; d < D; ++d) obs[((int64_t)b * S + s) * D + d] = 0.f;
  }
  if (tid == 0) {
    if (sp.f[F_ENV_ARRIVE]) fld<int32_t>(sp, F_ENV_ARRIVE)[b] = 0;      // (arrival counter of the pair-range rollout kernels: 0 between launches)
    fld<int32_t>(sp, F_ENV_STEP)[b] = 0;                                // env.py:209
    if (sp.env_type == PHX_ENV_FSM) fld<int32_t>(sp, F_ENV_STAGE)[b] = sp.initial_stage;   // fsm.py:217
  }
  __syncthreads();
  if (!obs) return;
  for (int k = tid; k < sp.n_reset_obs; k += NT) {                      // env.py:227-237
    const int a = sp.reset_obs_idx[k], s = tp.strat_rank[a];
    if (s < 0) continue;
    float ob[4] = {0.f, 0.f, 0.f, 0.f};
    const bool v = dev_encode_obs(sp, tp, b, a, 0, ob);              // `if v is not None` env.py:237
    for (int d = 0; d < D; ++d) obs[((int64_t)b * S + s) * D + d] = v ? ob[d] : 0.f;
    if (obs_valid) obs_valid[(int64_t)b * S + s] = v ? 1 : 0;
  }
}

// The spec (device memory, DevSpec::self_dev) and the launch arguments (the kernarg segment) are read through the scalar
// cache WHERE THEY ARE USED: both are constant-address-space pointers whose provenance is hidden from the compiler again
// at every phase boundary (PHX_REFRESH), so no pointer lives in an SGPR across phases.  Passed and used by value, the
// ~130 pointers of the two structs were loaded at entry and spilled: 321 spilled SGPRs, a fifth of the kernel's VALU
// instructions were the v_readlane / v_writelane of that spilling (the engine is VALU-issue bound: 4 waves per SIMD).
typedef const __attribute__((address_space(4))) char* phx_kptr_t;
#define sp (*(const DevSpec*)spc)
#define g (*(const GenArgs*)(kp + PHX_GENARGS_KERNARG_OFF))
#define PHX_REFRESH() asm volatile("" : "+s"(spc), "+s"(kp))
#define PHX_GENARGS_KERNARG_OFF 8
static_assert(alignof(GenArgs) == 8 && sizeof(const DevSpec*) == 8, "kernarg layout: (spec pointer, GenArgs at offset 8)");
// ROLL: the instantiation phx_rollout launches (policy, trajectory row, the caller's reset and the T-step loop compiled in);
// phx_step / phx_resolve run the one without that code
#ifndef PHX_LEAN_WAVES
#define PHX_LEAN_WAVES 6      // waves per SIMD the LEAN instantiations are compiled for (80 VGPRs: twelve two-wave workgroups per CU)
#endif
// LEAN (two-wave supply chains whose acting lists all have a static schedule): order / slot / scanbuf -- the scratch only a DYNAMIC
// step sorts and scans in -- live in the env's workspace in the blob instead of LDS: 11.5 instead of 14.6 KB per SC256 env
// The engine for ONE env instance b, by the calling workgroup of NT threads (whole steps, halves, bare resolves, the T-step loop): the
// body of phx_generic_step_kernel, and -- in the same launch as the compiled schedule -- what phx_sched_step_kernel's tail workgroups run
// for the env instances that kernel flags (phx_generic_sched.hip).  Contains workgroup barriers: uniform calls only.
template <int NT, bool LDSQ, bool TABLDS, int KMAX, bool ROLL, bool LEAN>
__device__ __forceinline__ void phx_generic_env(phx_kptr_t spc, phx_kptr_t kp, char* const smem, const int b) {
  PHX_REFRESH();

  __shared__ int wave_sums[NT / 64];
  __shared__ int s_errkey, s_nterm, s_ntrunc, s_dyn;
#ifdef PHX_TIMING
  __shared__ unsigned long long gtm[17];
  if (threadIdx.x < 17) gtm[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) gtm[16] = wall_clock64();
#define GTICK(k) do { PHX_REFRESH(); if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); gtm[k] += now_ - gtm[16]; gtm[16] = now_; } } while (0)
#else
#define GTICK(k) PHX_REFRESH()
#endif

  const int tid = threadIdx.x;
  const int A = sp.A, S = sp.S, Q = sp.queue_cap;

  char* mem = LDSQ ? smem : ((char*)sp.f[F_WORKSPACE] + (int64_t)b * sp.ws_stride);
  // queues: the round's messages, the responses by inbox position, and -- only where a handler still reads the old
  // queue while the next one is written (the exchanges' cooperative emission) -- a second queue to compact into;
  // otherwise the responses are compacted into the queue they answer (4.3 KB of LDS less per SC256 env)
  const bool two_queues = KMAX >= PHX_KIND_ADEXCHANGE && sp.n_adx > 0;
  DevMsg* q0 = (DevMsg*)mem;
  DevMsg* q1 = two_queues ? q0 + Q : q0;
  DevMsg* resp = q0 + (two_queues ? 2 : 1) * Q;
  int* order = LEAN ? (int*)((char*)sp.f[F_WORKSPACE] + (int64_t)b * sp.ws_stride) : (int*)(resp + Q);
  int* slot = order + Q;
  int* scanbuf = slot + Q;
  int* cnt = LEAN ? (int*)(resp + Q) : scanbuf + sp.scan_cap;
  int* first = cnt + A;
  int* goff = first + A;
  uint8_t* live = (uint8_t*)(goff + A);

  // static topology tables: LDS copies behind the queues (TABLDS) or the global arrays
  Topo tp = topo_env(sp, b);
  tp.kmax = KMAX;
  if (TABLDS) {
    char* tb = smem + g.tab_off;
    const int nnz = sp.nnz;
    int32_t* t_row = (int32_t*)tb;            tb += ((A + 1) * 4 + 15) & ~15;
    int32_t* t_col = (int32_t*)tb;            tb += (nnz * 4 + 15) & ~15;
    int32_t* t_pi = (int32_t*)tb;             tb += (A * PHX_NPI * 4 + 15) & ~15;
    int32_t* t_sr = (int32_t*)tb;             tb += (A * 4 + 15) & ~15;
    int32_t* t_kr = (int32_t*)tb;             tb += (A * 4 + 15) & ~15;
    int32_t* t_xr = (int32_t*)tb;             tb += (A * 4 + 15) & ~15;
    uint8_t* t_kind = (uint8_t*)tb;
    {                                                          // one flat 16-byte copy of the host-packed blob
      const uint4* src = (const uint4*)sp.tab_blob;
      uint4* dst = (uint4*)(smem + g.tab_off);
      for (int k = threadIdx.x; k < sp.tab_bytes / 16; k += NT) dst[k] = src[k];
    }
    tp.row_ptr = t_row; tp.col = t_col; tp.param_i = t_pi; tp.strat_rank = t_sr; tp.kind_rank = t_kr;
    tp.exo_rank = t_xr; tp.kind = t_kind;
    __syncthreads();
  }

  uint8_t* term = fld<uint8_t>(sp, F_ENV_TERM) + (int64_t)b * S;
  uint8_t* trunc = fld<uint8_t>(sp, F_ENV_TRUNC) + (int64_t)b * S;

  // ---- the T-step loop of a rollout launch (GenArgs::roll_T > 0: rollout.py:300-363 for this env instance with the queues,
  //      the staged tables and the workgroup resident; the env's words and agent state stay in the blob, L2-hot) ----------
  const int n_steps = ROLL && g.roll_T > 0 ? g.roll_T : 1;
  // the env's scalar words: read once through the scalar cache (written by an earlier launch), then kept in registers from
  // one step of the loop to the next (a reset inside the loop sets step = 0 and the initial stage, nothing else)
  // INVARIANT: these four words are written by this kernel only through w_step / w_tick / w_clock / w_stage below (the
  // epilogue and reset_env store what those registers hold); a new in-kernel writer of env.step / tick / clock / stage must update the
  // register copies too -- the scalar cache is NOT coherent with this launch's own vector stores.
  typedef const __attribute__((address_space(4))) int32_t* phx_ki32_t;
  int w_step = ((phx_ki32_t)(uintptr_t)fld<int32_t>(sp, F_ENV_STEP))[b];
  int w_tick = ((phx_ki32_t)(uintptr_t)fld<int32_t>(sp, F_ENV_TICK))[b];
  int w_clock = ((phx_ki32_t)(uintptr_t)fld<int32_t>(sp, F_ENV_CLOCK))[b];
  int w_stage = (sp.env_type == PHX_ENV_FSM) ? ((phx_ki32_t)(uintptr_t)fld<int32_t>(sp, F_ENV_STAGE))[b] : 0;
  for (int it = 0; it < n_steps; ++it) {
  PHX_REFRESH();
  const int roll_t = ROLL && g.roll_t >= 0 ? g.roll_t + it : -1;       // trajectory row of this step (-1: a plain phx_step)
  const int64_t step_env = (ROLL && g.roll_T > 0 ? (int64_t)it * sp.B : 0) + b;   // row of the per-step [T][B][..] inputs / logs
  // the env's scalar words (a later step reads what the previous one's epilogue / reset wrote: same workgroup, one L1)
  const bool full = !g.resolve_only;
  const int step_in = w_step;
  const uint32_t tick = (uint32_t)w_tick;
  const int clock0 = w_clock;
  const int cur_stage = w_stage;
  const int t = full ? step_in + 1 : step_in;                  // env.py:252
  int list = 0;                                                // which acting list / mask row
  if (sp.env_type == PHX_ENV_FSM) list = cur_stage;                          // fsm.py:276
  else if (sp.env_type == PHX_ENV_STACKELBERG) list = (t & 1) ? 0 : 1;       // stackelberg.py:133-137

  if (tid == 0) { s_errkey = ERRKEY_NONE; s_nterm = 0; s_ntrunc = 0; s_dyn = 0; }
  // the list's precomputed round schedule (DevSpec::sched), valid for this step only if every agent has a context and every
  // acting supply-chain agent sends its message (s_dyn records a violation); injected sends and bare resolves are dynamic
  const int32_t* sch = nullptr;
  if (sp.sched && full && g.n_inject == 0 && g.phase == 0) { const int so = sp.sched_off[list]; if (so >= 0) sch = sp.sched + so; }
  __syncthreads();
  // _make_ctxs (env.py:338-348): contexts only for agents that are not done
  for (int a = tid; a < A; a += NT) {
    const int s = tp.strat_rank[a];
    live[a] = (g.resolve_only || s < 0) ? 1 : !(term[s] | trunc[s]);
    if (sch) {
      if (!live[a]) s_dyn = 1;
      else if (s >= 0 && tkind(tp, a) == PHX_KIND_SHOP && sp.act_mask[(int64_t)list * A + a]) {     // an acting shop without an action
        const bool has = (ROLL && g.roll_t >= 0) || (g.io.actions && (!g.io.action_valid || g.io.action_valid[(int64_t)b * S + s]));
        if (!has) s_dyn = 1;
      }
    }
  }
  if (ROLL && roll_t >= 0) {
    // the policy of a rollout, fused: every strategic agent's action of this tick (random policy = the
    // agent's word of the tick, rank j mapped onto the kind's action space), recorded in the trajectory whether or not it acts
    for (int s = tid; s < S; s += NT) {
      const int a = sp.strat_idx[s], kind = tkind(tp, a);
      bool acts = true;
      if (sp.env_type == PHX_ENV_STACKELBERG) acts = sp.act_mask[(int64_t)list * A + a] != 0;
      float action = 0.f;
      if (acts) {
        if (g.roll_actions_in) action = g.roll_actions_in[((int64_t)roll_t * sp.B + b) * S + s];
        else {
          uint32_t j;
          rng_group_y(sp.seed, sp.env_offset + b, tick, s, 0, 0, &j);
          action = (kind == PHX_KIND_SELLER || kind == PHX_KIND_ADVERTISER) ? (float)j * (1.0f / 274877.0f)
                 : kind == PHX_KIND_BUYER ? (j < 137438u ? 1.0f : 0.0f) : rng_j_to_action(j);
        }
      }
      g.roll_actions[(int64_t)b * S + s] = action;
      g.roll.action_out[((int64_t)roll_t * sp.B + b) * S + s] = action;
    }
  }
  __syncthreads();
  GTICK(0);

  // ---- round-0 queue: host-injected sends, then the acting agents in list order -------------
  const int n_act = (full && g.phase != 2) ? sp.act_ptr[list + 1] - sp.act_ptr[list] : 0;      // (phx_step_end: acting and resolution were phx_step_begin's)
  const int32_t* act_list = sp.act_idx + sp.act_ptr[list];
  const int n_items = g.n_inject + n_act;
  const float* actions_b = g.io.actions ? g.io.actions + (int64_t)b * S : nullptr;
  const uint8_t* av_b = g.io.action_valid ? g.io.action_valid + (int64_t)b * S : nullptr;
  const uint8_t* exo_b = g.io.exo ? g.io.exo + step_env * sp.n_exo : nullptr;

  const bool use_sched = sch != nullptr && __builtin_amdgcn_readfirstlane(s_dyn) == 0;   // uniform: s_dyn was last written before the barrier above
  int n;
  if (use_sched) {
    // scheduled step: every acting item's queue offset is in the table and every send passed its checks at phx_create
    const int32_t* act_off = sch + 1 + PHX_SCHED_MAX_ROUNDS;
    for (int it = tid; it < n_act; it += NT) {
      const int off = act_off[it];
      const int a = act_list[it], s = tp.strat_rank[a];
      const bool has = s >= 0 && actions_b && (!av_b || av_b[s]);            // aid in actions, env.py:330
      // (an item without a message still decodes its action: the mock agents count decode_action calls)
      // (off < 0: the schedule says this item sends nothing -- mock kinds that only count decode_action calls; q0 + 0 is then never written:
      //  act_emit stores only when the kind emits, and a kind that emits has off >= 0 by construction of the schedule)
      act_emit(sp, tp, b, a, has, has ? actions_b[s] : 0.0f, exo_b, tick, q0 + (off < 0 ? 0 : off));
    }
    n = sch[1];
    __syncthreads();
    GTICK(3);
  } else {
  // per-item message counts -> exclusive scan -> queue offsets (scanbuf holds scan_cap >= n_items)
  for (int it = tid; it < n_items; it += NT) {
    int c = 0;
    if (it < g.n_inject) {
      const DevMsg m = g.inject[it];
      const int code = dev_send_check(sp, tp, m.src, m.dst, m.type);
      if (code) set_errkey(&s_errkey, it, code); else c = 1;
    } else {
      const int a = act_list[it - g.n_inject];
      if (live[a]) {                                           // env.py:324-325
        const int s = tp.strat_rank[a];
        const bool has = s >= 0 && actions_b && (!av_b || av_b[s]);        // aid in actions, env.py:330
        c = act_count(sp, tp, b, a, has, has ? actions_b[s] : 0.0f);
      }
    }
    scanbuf[it] = c;
  }
  __syncthreads();
  GTICK(1);
  n = block_exscan<NT>(scanbuf, n_items, wave_sums);
  if (n > Q) { if (tid == 0) set_errkey(&s_errkey, n_items, PHX_ERR_QUEUE_FULL); n = 0; }
  else {
    for (int it = tid; it < n_items; it += NT) {
      const int off = scanbuf[it];
      if (it < g.n_inject) {
        const DevMsg m = g.inject[it];
        if (dev_send_check(sp, tp, m.src, m.dst, m.type) == 0) q0[off] = m;
      } else {
        const int a = act_list[it - g.n_inject];
        if (live[a]) {
          const int s = tp.strat_rank[a];
          const bool has = s >= 0 && actions_b && (!av_b || av_b[s]);
          // decode_action's state mutation happens exactly once, here; an agent that sends
          // nothing writes nothing
          act_emit(sp, tp, b, a, has, has ? actions_b[s] : 0.0f, exo_b, tick, q0 + off);
        }
      }
    }
  }
  __syncthreads();
  GTICK(2);
  // acting-phase sends are checked like any Network.send (static topologies always pass)
  for (int i = tid; i < n; i += NT) {
    const DevMsg m = q0[i];
    const int code = dev_send_check(sp, tp, m.src, m.dst, m.type);
    if (code) { set_errkey(&s_errkey, g.n_inject + i, code); q0[i].type = 0; }
  }
  __syncthreads();
  GTICK(3);
  }   // dynamic acting phase

  // pre_message_resolution for every live agent (env.py:170-173)
  if (full && g.phase != 2)
    for (int a = tid; a < A; a += NT) if (live[a]) pre_resolution(sp, tp, b, a, t);

  phx_msg_rec* log_b = g.io.msg_log ? g.io.msg_log + step_env * sp.trace_cap : nullptr;
  if (log_b)                                                    // Resolver.push tracking, resolvers.py:41-42
    for (int i = tid; i < n; i += NT)
      if (i < sp.trace_cap) {
        phx_msg_rec r; r.sender = q0[i].src; r.receiver = q0[i].dst; r.type = q0[i].type; r.round = 0;
        r.payload.i = q0[i].p.i; log_b[i] = r;
      }
  int log_n = n;
  int seq_base = n_items;          // error ordering key continues after the acting items
  int clock = clock0;
  int shuf_off = 0;             // messages of the step's earlier rounds (index into the replayed shuffle stream)
  __syncthreads();
  GTICK(4);

  // ---- BatchResolver.resolve round loop (resolvers.py:128-163) ---------------------------------
  DevMsg* qc = q0; DevMsg* qn = q1;
  int round = 0;
  const int sch_R = use_sched ? sch[0] : 0;
  int sch_pos = 1 + PHX_SCHED_MAX_ROUNDS + n_act;            // offset of the current round's tables in the schedule
  while (n > 0 && (sp.round_limit < 0 || round < sp.round_limit) && round < PHX_MAX_ROUNDS) {
    const int* ord;                                          // inbox position -> queue index, batches in send order
    const int32_t* next_off = nullptr;                       // scheduled round: where the reply to each inbox position goes
    // (n is the same in every lane; readfirstlane tells the compiler so: the branches below hold barriers and scalar loads)
    const bool sched_round = use_sched && round < sch_R && __builtin_amdgcn_readfirstlane(n) == sch[1 + round];
    if (sched_round) {
      // static round: inbox sizes, offsets (receivers in first-arrival order) and batch order come from the table
      const int32_t* rt = sch + sch_pos;
      for (int a = tid; a < A; a += NT) { const int c_ = rt[a], o_ = rt[A + a]; cnt[a] = c_; goff[a] = o_; }
      ord = rt + 2 * A;
      next_off = ord + n;
      sch_pos += 2 * A + 2 * n;
      __syncthreads();
      GTICK(9);
    } else {
    for (int a = tid; a < A; a += NT) { cnt[a] = 0; first[a] = 0x7fffffff; }
    __syncthreads();
    GTICK(5);
    for (int i = tid; i < n; i += NT) {
      const int d = qc[i].dst;
      slot[i] = atomicAdd(&cnt[d], 1);
      atomicMin(&first[d], i);
    }
    __syncthreads();
    GTICK(6);
    // receivers in dict (first-arrival) order -> inbox offsets
    for (int i = tid; i < n; i += NT) {
      const int d = qc[i].dst;
      scanbuf[i] = (first[d] == i) ? cnt[d] : 0;
    }
    __syncthreads();
    GTICK(7);
    block_exscan<NT>(scanbuf, n, wave_sums);
    for (int i = tid; i < n; i += NT) {
      const int d = qc[i].dst;
      if (first[d] == i) goff[d] = scanbuf[i];
    }
    __syncthreads();
    GTICK(8);
    for (int i = tid; i < n; i += NT) order[goff[qc[i].dst] + slot[i]] = i;
    __syncthreads();
    GTICK(9);
    // ---- batches in send order.  The slots the atomics handed out are in ARRIVAL order, which is not deterministic:
    //      message i takes the rank of its sequence number among its receiver's batch (c independent LDS reads per
    //      message, whatever the batch length -- the factory of SC256 receives 51 requests, an exchange 120 bids; round 1
    //      sorted short batches with a dependent insertion-sort chain in the receiver's lane and long ones hub by hub)
    for (int i = tid; i < n; i += NT) {
      const int d = qc[i].dst, base = goff[d], c = cnt[d];
      int r = 0;
      for (int j = 0; j < c; ++j) r += order[base + j] < i;
      slot[base + r] = i;
    }
    __syncthreads();
    { int* const t_ = order; order = slot; slot = t_; }          // order: the sorted batches; slot: scratch until the next round
    ord = order;
    }   // dynamic round
    const bool has_adx = KMAX >= PHX_KIND_ADEXCHANGE && sp.n_adx > 0;
    const int n_adx = has_adx ? sp.n_adx : 0;
    for (int x = 0; x < n_adx; ++x) {                          // exchanges: workgroup-wide batch reduction (phase A)
      const int a = sp.adx_idx[x];
      if (cnt[a] > 0)
          adx_coop_count<NT>(sp, tp, x, a, live, qc, order + goff[a], cnt[a], slot + goff[a], scanbuf + goff[a], resp + goff[a],
                             wave_sums, first, &s_errkey, seq_base + goff[a]);
    }
    if (sp.flags & PHX_F_SHUFFLE_BATCHES) {                     // np.random.shuffle(msgs), resolvers.py:150-151
      const uint16_t* sh = g.io.shuffle ? g.io.shuffle + (int64_t)b * 8 * Q + shuf_off : nullptr;
      for (int a = tid; a < A; a += NT) {
        const int c = cnt[a];
        if (c < 2 || !live[a]) continue;                        // `receiver_id not in contexts`: no shuffle call
        int* seg = order + goff[a];
        int* tmp = slot + goff[a];                              // dead entries of this round
        if (sh) {                                               // replayed permutation: batch position k takes message sh[k]
          const bool fits = shuf_off + goff[a] + c <= 8 * Q;
          for (int k = 0; k < c; ++k) { const int j = fits ? sh[goff[a] + k] : k; tmp[k] = seg[j < c ? j : k]; }
          for (int k = 0; k < c; ++k) seg[k] = tmp[k];
        } else {                                                // Fisher-Yates, the loop numpy's legacy shuffle runs
          uint32_t w[4]; int have = -1;
          for (int i = c - 1, d = 0; i >= 1; --i, ++d) {
            if ((d >> 2) != have) { have = d >> 2; rng_shuffle_block(sp.seed, sp.env_offset + b, tick, round, a, (uint32_t)have, w); }
            const int j = (int)__umulhi(w[d & 3], (uint32_t)(i + 1));
            const int vi = seg[i]; seg[i] = seg[j]; seg[j] = vi;
          }
        }
      }
      __syncthreads();
    }
    // dropped: receiver not in contexts (resolvers.py:143-144), edge filter (:146-148), or a send that already
    // failed its checks (type 0).  The receive-side edge filter can only drop something when sends skipped the
    // edge check (ignore_connection_errors): every queued message already passed has_edge.
    // `r`, `live_a`: the receiver's agent_ref and context flag; `src_kind`: kind of the sender (for the payload whitelist of a reply)
    auto deliver = [&](int a, const AgentRef& r, bool live_a, int P, const DevMsg& m, int src_kind, AgentState& st) __attribute__((always_inline)) {
      DevMsg out; out.type = 0;
      if (sched_round) {                                        // scheduled round: every check was made at phx_create
        int code = 0;
        if (!handle_message(sp, tp, b, a, r, m, clock + P, exo_b, tick, out, code, st)) out.type = 0;
        if (code) set_errkey(&s_errkey, seq_base + P, code);
      } else if (live_a && m.type != 0 && (!(sp.flags & PHX_F_IGNORE_CONN_ERRORS) || dev_has_edge(sp, tp, m.src, m.dst))) {
        int code = 0;
        const bool answered = handle_message(sp, tp, b, a, r, m, clock + P, exo_b, tick, out, code, st);
        if (code) set_errkey(&s_errkey, seq_base + P, code);
        if (answered) {                                         // network.send(receiver, sub_receiver, payload) :156-158
          // a reply to the sender travels the edge the delivered message came along (connections are undirected,
          // also per env on a StochasticNetwork): only the payload whitelist is left to check
          const int sc = out.dst == m.src ? dev_payload_check_kinds(sp, r.kind, src_kind, out.type)
                                          : dev_send_check(sp, tp, out.src, out.dst, out.type);
          if (sc) { set_errkey(&s_errkey, seq_base + P, sc); out.type = 0; }
        } else out.type = 0;
      }
      resp[P] = out;
      if (!sched_round) scanbuf[P] = out.type != 0;
    };
    // receivers without state (factory, customer, ...): one lane per MESSAGE -- the handlers of a batch commute
    for (int P = tid; P < n; P += NT) {
      const DevMsg m = qc[ord[P]];
      const int a = m.dst;
      if (!kind_stateless(tkind(tp, a))) continue;
      AgentState none;
      deliver(a, agent_ref(sp, tp, b, a), live[a] != 0, P, m, tkind(tp, m.src), none);
    }
    // every other receiver: one lane walks its batch, handled one message at a time (agents.py:96-120)
    for (int a = tid; a < A; a += NT) {
      const int c = cnt[a];
      if (c == 0) continue;
      const int kind_a = tkind(tp, a);
      if (kind_stateless(kind_a)) continue;
      if (has_adx && kind_a == PHX_KIND_ADEXCHANGE && first[a] != -1) continue;    // done by phase A
      const int* seg = ord + goff[a];
      if (kind_a == PHX_KIND_ADEXCHANGE) {                      // overrides handle_batch: count now, emit after the scan
        for (int k = 0; k < c; ++k) resp[goff[a] + k].type = 0;
        adexchange_batch(sp, tp, a, qc, seg, c,
                         [&](int k) { const DevMsg m = qc[seg[k]];
                                      return live[a] && m.type != 0 && (!(sp.flags & PHX_F_IGNORE_CONN_ERRORS) || dev_has_edge(sp, tp, m.src, m.dst)); },
                         false, scanbuf + goff[a], nullptr, nullptr, &s_errkey, seq_base + goff[a]);
        continue;
      }
      const AgentRef r = agent_ref(sp, tp, b, a);
      const bool cached = kind_state_cached(kind_a);
      const bool live_a = live[a] != 0;
      const int base = goff[a];
      AgentState st;
      if (cached) load_state(sp, r, st);
      // The batch is handled in order, four messages at a time: their queue entries and the senders' kinds are fetched
      // first (independent LDS reads), the handlers then run on registers.  One message at a time cost ~1 000 cycles
      // each -- index, entry, receiver's context flag / kind / rank and the sender's kind re-read behind every response
      // store -- and the longest such chain (a shop's six orders) is what the round waits for.
      for (int k0 = 0; k0 < c; k0 += 4) {
        DevMsg mm[4]; int sk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) mm[j] = qc[seg[k0 + j < c ? k0 + j : c - 1]];
#pragma unroll
        for (int j = 0; j < 4; ++j) sk[j] = tkind(tp, mm[j].src);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k0 + j < c) deliver(a, r, live_a, base + k0 + j, mm[j], sk[j], st);
      }
      if (cached) store_state(sp, r, st);
    }
    __syncthreads();
    GTICK(10);
    int n_next;
    if (sched_round) {                                          // scheduled round: the replies' offsets are static (no scan)
      n_next = round + 1 < sch_R ? sch[2 + round] : 0;
      for (int P = tid; P < n; P += NT) { const int off = next_off[P]; if (off >= 0) qn[off] = resp[P]; }
    } else n_next = block_exscan<NT>(scanbuf, n, wave_sums);
    if (n_next > Q) { if (tid == 0) set_errkey(&s_errkey, seq_base + n, PHX_ERR_QUEUE_FULL); n_next = 0; }
    else {
      if (!sched_round)
        for (int P = tid; P < n; P += NT)
          if (resp[P].type != 0) qn[scanbuf[P]] = resp[P];
      for (int x = 0; x < n_adx; ++x) {
        const int a = sp.adx_idx[x];
        if (cnt[a] > 0 && first[a] != -1)
          adx_coop_emit<NT>(sp, tp, x, a, qc, order + goff[a], cnt[a], slot + goff[a], scanbuf + goff[a], qn, first[a]);
      }
      if (has_adx)
        for (int a = tid; a < A; a += NT)
          if (cnt[a] > 0 && tkind(tp, a) == PHX_KIND_ADEXCHANGE && first[a] == -1)
            adexchange_batch(sp, tp, a, qc, order + goff[a], cnt[a],
                             [&](int k) { const DevMsg m = qc[order[goff[a] + k]];
                                          return live[a] && m.type != 0 && (!(sp.flags & PHX_F_IGNORE_CONN_ERRORS) || dev_has_edge(sp, tp, m.src, m.dst)); },
                             true, nullptr, scanbuf + goff[a], qn, &s_errkey, seq_base + goff[a]);
      if (log_b) {
        __syncthreads();
        for (int i = tid; i < n_next; i += NT)
          if (log_n + i < sp.trace_cap) {
            phx_msg_rec r; r.sender = qn[i].src; r.receiver = qn[i].dst; r.type = qn[i].type;
            r.round = (uint16_t)(round + 1); r.payload.i = qn[i].p.i; log_b[log_n + i] = r;
          }
      }
    }
    __syncthreads();
    GTICK(11);
    log_n += n_next; seq_base += n; clock += n; shuf_off += n;
    n = n_next; ++round;
    DevMsg* tq = qc; qc = qn; qn = tq;
  }
  if (n > 0 && tid == 0) set_errkey(&s_errkey, seq_base + 1, PHX_ERR_ROUND_LIMIT);   // resolvers.py:160-163
  // a stage handler's return value (fsm.py:294-307), decided by the host: must be one of the stage's next_stages
  int next_in = -1;
  bool by_rule = false;
  if (full && g.phase != 1 && sp.env_type == PHX_ENV_FSM && sp.n_rules > 0 && !g.io.next_stage) {
    // a handler declared as rules (phx_spec.stage_rules): evaluated here, on the RESOLVED state -- the handlers' stores of this step
    // are behind the barrier above -- by the env's first wave: a column, or the sum over the kind's agents (lanes stride the
    // columns, wave reduction); first rule of the current stage whose condition holds, else next_stages[0]
    int chosen = -1;
    for (int r = 0; r < sp.n_rules && chosen < 0; ++r) {
      const DevRule q = sp.rules[r];
      if (q.stage != cur_stage) continue;
      double v = 0.0;
      if (tid < 64) {
        const int64_t base = (int64_t)b * q.ncols;
        if (q.col >= 0) { if (tid == 0) v = q.is_f64 ? ((const double*)sp.f[q.field_id])[base + q.col] : (double)((const int32_t*)sp.f[q.field_id])[base + q.col]; }
        else for (int c = tid; c < q.ncols; c += 64) v += q.is_f64 ? ((const double*)sp.f[q.field_id])[base + c] : (double)((const int32_t*)sp.f[q.field_id])[base + c];
        // (i32 fields: exact in f64; f64 fields: the sum is taken in lane order 0..63 over strided partial sums -- the oracle adds in
        //  the same order)
        for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_xor(v, off, 64); v = (tid & off) ? o + v : v + o; }
      }
      v = __shfl(v, 0, 64);
      const bool hit = q.cmp == PHX_CMP_LT ? v < q.threshold : q.cmp == PHX_CMP_LE ? v <= q.threshold : q.cmp == PHX_CMP_GT ? v > q.threshold :
                       q.cmp == PHX_CMP_GE ? v >= q.threshold : q.cmp == PHX_CMP_EQ ? v == q.threshold : v != q.threshold;
      if (hit) chosen = q.next_stage;
    }
    if (NT > 64) {                                             // wider workgroups: the first wave's choice for everybody
      __shared__ int s_rule_next;
      if (tid == 0) s_rule_next = chosen;
      __syncthreads();
      chosen = s_rule_next;
    }
    if (chosen >= 0) { next_in = chosen; by_rule = true; }
  }
  if (!by_rule && full && g.phase != 1 && sp.env_type == PHX_ENV_FSM && (g.io.next_stage || sp.stage_tab)) {
    // the host's handler call for this step, or the tabulated handler's value at (stage, clock) -- validated at phx_create
    next_in = g.io.next_stage ? g.io.next_stage[b] : sp.stage_tab[(int64_t)cur_stage * (sp.num_steps + 1) + (t <= sp.num_steps ? t : sp.num_steps)];
    if (next_in < 0 || next_in >= sp.n_lists || !sp.stage_allowed[(int64_t)cur_stage * sp.n_lists + next_in]) {
      if (tid == 0) set_errkey(&s_errkey, seq_base + 2, PHX_ERR_FSM_TRANSITION);     // FSMRuntimeError, after the resolution
      next_in = -1;
    }
  }
  __syncthreads();
  GTICK(12);

  if (tid == 0) {
    fld<int32_t>(sp, F_ENV_CLOCK)[b] = clock;
    if (g.io.msg_count && g.phase != 2) g.io.msg_count[step_env] = log_n;
    if (g.io.err && s_errkey != ERRKEY_NONE && g.io.err[b] == 0) g.io.err[b] = s_errkey & 15;       // (the sticky word is read only when there is something to report)
  }
  if (g.resolve_only || g.phase == 1) return;      // (phx_step_begin stops here: the env's step counter and tick stay, the epilogue is phx_step_end's)

  GTICK(13);
  int all_flags = 0;
  const bool row_done = strategic_epilogue<NT>(sp, tp, g.io, b, t, list, cur_stage, tick, live, &s_nterm, &s_ntrunc, next_in,
                                               ROLL && roll_t >= 0 ? &g.roll : nullptr, ((int64_t)(roll_t >= 0 ? roll_t : 0) * sp.B + b) * S, &all_flags);
  all_flags = __builtin_amdgcn_readfirstlane(all_flags);
  GTICK(14);
  if (ROLL && roll_t >= 0) {                                   // the step's outputs -> trajectory row roll_t
    const phx_step_io& st = g.io; const phx_rollout_io& io = g.roll;
    const uint8_t at = (uint8_t)(all_flags & 1), au = (uint8_t)((all_flags >> 1) & 1);
    if (!row_done) {                                           // more strategic agents than lanes: copy what the epilogue stored
      __syncthreads();
      for (int s = tid; s < S; s += NT) {
        const int64_t i = (int64_t)b * S + s, o = ((int64_t)roll_t * sp.B + b) * S + s;
        for (int d = 0; d < sp.D; ++d) io.obs[o * sp.D + d] = st.obs[i * sp.D + d];
        io.reward[o] = (float)st.reward[i];
        if (io.terminated) io.terminated[o] = (uint8_t)(st.terminated[i] | at);
        io.truncated[o] = (uint8_t)(st.truncated[i] | au);
        if (io.obs_valid) io.obs_valid[o] = st.obs_valid[i];
        if (io.reward_valid) io.reward_valid[o] = st.reward_valid[i];
      }
    }
    w_step = t; w_tick = (int)(tick + 1); w_clock = clock; w_stage = all_flags >> 8;
    if (at | au) {                                             // the caller's env.reset(), same launch
      __syncthreads();
      reset_env<NT>(sp, b, nullptr, nullptr, st.obs, st.obs_valid, KMAX);
      w_step = 0; w_stage = sp.initial_stage;
    }
    if (g.roll_T > 0 && it == n_steps - 1 && io.last_obs) {    // the observation after the fragment (after a reset: the reset's)
      __syncthreads();
      for (int k = tid; k < S * sp.D; k += NT) io.last_obs[(int64_t)b * S * sp.D + k] = st.obs[(int64_t)b * S * sp.D + k];
    }
  }
  __syncthreads();                                             // the next step reads the words and the state this one wrote
  }   // steps of the launch
#ifdef PHX_TIMING
  if (g.timing && threadIdx.x == 0 && blockIdx.x < 64) for (int q = 0; q < 16; ++q) atomicAdd(&g.timing[q], gtm[q]);
#endif
}

// one env instance per workgroup
template <int NT, bool LDSQ, bool TABLDS, int KMAX, bool ROLL, bool LEAN>
// (the supply-chain rollout instantiation of two-wave workgroups is held to 96 VGPRs: five waves per SIMD = the ten workgroups per CU the queues allow)
__global__ __launch_bounds__(NT, LEAN ? PHX_LEAN_WAVES : ((ROLL && NT == 128 && KMAX <= PHX_KIND_CUSTOMER) ? 5 : 1)) void phx_generic_step_kernel(const DevSpec* __restrict__ spp_, const GenArgs g_) {
  phx_kptr_t spc = (phx_kptr_t)spp_;
  phx_kptr_t kp = (phx_kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  PHX_REFRESH();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  phx_generic_env<NT, LDSQ, TABLDS, KMAX, ROLL, LEAN>(spc, kp, smem, xcd_block(g.xcd_remap != 0));
}

#undef sp
#undef g

size_t phx_generic_queue_bytes(int A, int S, int Q, int scan_cap, int n_adx, bool lean);
#include "phx_generic_sched.hip"      // phx_sched_step_kernel (its tail workgroups call phx_generic_env) and its launcher

// ---- PhantomEnv.reset (env.py:185-237; fsm.py:195-251; stackelberg.py:53-109) -------------------
template <int NT>
__global__ __launch_bounds__(NT) void phx_reset_kernel(const DevSpec sp, const uint8_t* mask,
                                                       const double* sampler_values, const uint8_t* conn_values,
                                                       float* obs, uint8_t* obs_valid) {
  const int b = xcd_block(true);                           // the generic engine's env -> XCD mapping
  if (mask && !mask[b]) return;
  reset_env<NT>(sp, b, sampler_values, conn_values, obs, obs_valid);
}

// ---- launchers (called from phx_api.hip) -----------------------------------------------------
size_t phx_generic_lean_ws_bytes(int Q, int scan_cap) { return ((size_t)2 * Q + scan_cap) * sizeof(int); }
size_t phx_generic_queue_bytes(int A, int S, int Q, int scan_cap, int n_adx, bool lean) {
  return (size_t)Q * ((n_adx > 0 ? 3 : 2) * sizeof(DevMsg)) + (lean ? 0 : phx_generic_lean_ws_bytes(Q, scan_cap)) +
         (size_t)A * 3 * sizeof(int) + (size_t)((A + 15) & ~15);
}

size_t phx_generic_table_bytes(int A, int nnz) {
  return (size_t)(((A + 1) * 4 + 15) & ~15) + (size_t)((nnz * 4 + 15) & ~15) + (size_t)((A * PHX_NPI * 4 + 15) & ~15) +
         3 * (size_t)((A * 4 + 15) & ~15) + (size_t)((A + 15) & ~15);
}

static hipError_t launch_generic_dynamic(const DevSpec& sp, const GenArgs& g_, bool lds, hipStream_t st);

// The engine's entry.  Specs with a compiled schedule (DevSpec::gs_ok: static Network, supply-chain kinds) run whole steps and
// T-step rollout loops on phx_sched_step_kernel (whose tail workgroups step the envs it flags on the dynamic engine, same launch).  Everything else -- injected sends, bare resolves, the two halves of a split step, replayed shuffles, the
// one-launch-per-step loop, every other kind -- is the dynamic kernel's.
hipError_t phx_launch_generic(const DevSpec& sp, const GenArgs& g_, bool lds, hipStream_t st) {
  if (sp.gs_ok && phx_knobs().generic_sched && sp.variant_step != PHX_VS_GENERIC_DYNAMIC && g_.phase == 0 && !g_.resolve_only && g_.n_inject == 0 &&
      !g_.io.shuffle && (g_.roll_t < 0 || g_.roll_T > 0)) {
    const GenArgs& g = g_;
    return phx_launch_sched(sp, g, st);
  }
  return launch_generic_dynamic(sp, g_, lds, st);
}

static hipError_t launch_generic_dynamic(const DevSpec& sp, const GenArgs& g_, bool lds, hipStream_t st) {
  GenArgs g = g_;
  const bool lean = lds && sp.lean_lds != 0;
  size_t bytes = (phx_generic_queue_bytes(sp.A, sp.S, sp.queue_cap, sp.scan_cap, sp.n_adx, lean) + 15) & ~(size_t)15;
  const size_t tab = phx_generic_table_bytes(sp.A, sp.nnz);
  const int tablds_env = phx_knobs().generic_tablds;
  // Staging the topology tables in LDS saves latency per lookup but costs occupancy: every workgroup of the CU holds its
  // own copy.  Worth it only while queues + tables stay small (SC64: 6 KB); at SC256 (15.5 KB of queues + 10.5 KB of
  // tables = 6 workgroups per CU with them, 10 without) leaving them in global memory is 18 % faster (144 -> 118 us).
  const bool tablds = lds && !lean && tablds_env && (tablds_env > 1 || bytes + tab <= 10 * 1024) && tab <= 24 * 1024 && bytes + tab <= 60 * 1024;
  g.tab_off = (int32_t)bytes;
  // one env per workgroup writes ~100-byte output segments: with consecutive envs on one XCD their shared cache
  // lines merge in one L2 (SC64 38.6 -> 37.1 us, SC256-FSM 352 -> 343 us per step)
  const int remap_env = phx_knobs().generic_remap;
  g.xcd_remap = remap_env;
  if (tablds) bytes += tab;
  // threads per env: one wave while the agents fit it (the barriers of a single-wave workgroup are
  // free and a CU holds sixteen of them), two waves for wider envs.  Measured at 64 / 128 / 256:
  // SC64 B=4096 40 / 63 / 94 us per step, SC256-FSM B=8192 819 / 727 / 743 us.  (Keeping the env's
  // agent state in LDS for the step was measured too: no gain, the wave is instruction-bound --
  // about 6 000 instructions and 90 memory operations per env-step at SC64.)
  const int nt_env = phx_knobs().generic_nt;      // development: 64 or 128
  // round 2, after the factory's serial chain went message-parallel: SC256-FSM B=8192 199 / 165 / 149 us per step
  // (with the second queue gone -- 6 instead of 5 workgroups per CU -- 128 threads win again: 144 vs 177 us)
  int nt = nt_env ? nt_env : (sp.A <= 64 ? 64 : 128);
  // a spec whose kinds are all supply-chain kinds (FACTORY / SHOP / CUSTOMER = 1..3) runs the instantiation that compiles
  // only their handlers (fewer registers, no exchange / market code in the round loop)
  int kmax = 0;
  for (int k = 0; k < PHX_KIND_COUNT; ++k) if (sp.kind_count[k] > 0) kmax = k;
  const bool sc_only = kmax <= PHX_KIND_CUSTOMER;
  phx_note_kernel(g.roll_T > 0 ? "phx_generic_step_kernel[T-step loop]" : "phx_generic_step_kernel");
  const bool roll = g.roll_t >= 0;
  const dim3 grid((unsigned)sp.B);
#define PHX_LAUNCH_GENERIC_R(NT_, L_, T_, K_, R_) hipLaunchKernelGGL((phx_generic_step_kernel<NT_, L_, T_, K_, R_, false>), grid, dim3(NT_), bytes, st, sp.self_dev, g)
#define PHX_LAUNCH_GENERIC_K(NT_, L_, T_, K_) do { if (roll) PHX_LAUNCH_GENERIC_R(NT_, L_, T_, K_, true); else PHX_LAUNCH_GENERIC_R(NT_, L_, T_, K_, false); } while (0)
#define PHX_LAUNCH_GENERIC(NT_, L_, T_) do { if (sc_only) PHX_LAUNCH_GENERIC_K(NT_, L_, T_, PHX_KIND_CUSTOMER); else PHX_LAUNCH_GENERIC_K(NT_, L_, T_, PHX_KIND_COUNT - 1); } while (0)
  if (lean) {                                                // (lean_lds_spec: supply-chain kinds only, 64 < A <= 256)
    if (roll) hipLaunchKernelGGL((phx_generic_step_kernel<128, true, false, PHX_KIND_CUSTOMER, true, true>), grid, dim3(128), bytes, st, sp.self_dev, g);
    else hipLaunchKernelGGL((phx_generic_step_kernel<128, true, false, PHX_KIND_CUSTOMER, false, true>), grid, dim3(128), bytes, st, sp.self_dev, g);
  } else if (!lds) { bytes = 0; PHX_LAUNCH_GENERIC(256, false, false); }
  else if (tablds) { if (nt == 64) PHX_LAUNCH_GENERIC(64, true, true); else PHX_LAUNCH_GENERIC(128, true, true); }
  else { if (nt == 64) PHX_LAUNCH_GENERIC(64, true, false); else PHX_LAUNCH_GENERIC(128, true, false); }
#undef PHX_LAUNCH_GENERIC_K
#undef PHX_LAUNCH_GENERIC_R
#undef PHX_LAUNCH_GENERIC
  return hipGetLastError();
}

hipError_t phx_launch_reset(const DevSpec& sp, const uint8_t* mask, const double* sampler_values,
                            const uint8_t* conn_values, float* obs, uint8_t* obs_valid, hipStream_t st) {
  hipLaunchKernelGGL((phx_reset_kernel<64>), dim3(sp.B), dim3(64), 0, st, sp, mask, sampler_values, conn_values, obs, obs_valid);
  return hipGetLastError();
}

// ---- launch-loop rollout for envs without a fused rollout kernel (phx_api.hip: phx_rollout) ------------
// The policy (random: the strategic agent's word of the tick, rank j -- ShopAgent / mock agents j * 100 / 274877,
// Seller price and Advertiser bid fraction j / 274877, Buyer buys iff j < 137438; on a Stackelberg env only the
// side that acts takes and records an action) and the copy of a step's outputs into trajectory row t are fused
// into phx_generic_step_kernel (GenArgs::roll_t >= 0); per step the loop launches the engine and the masked reset.
__global__ void phx_gen_last_obs_kernel(const int64_t n, const float* obs, float* last_obs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) last_obs[i] = obs[i];
}

hipError_t phx_launch_gen_last_obs(const DevSpec& sp, const float* obs, float* last_obs, hipStream_t st) {
  const int64_t n = (int64_t)sp.B * sp.S * sp.D;
  if (n > 0) hipLaunchKernelGGL(phx_gen_last_obs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, obs, last_obs);
  return hipGetLastError();
}

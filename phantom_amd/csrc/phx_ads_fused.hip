// phx_ads_fused.hip -- the digital-ads market with its static schedule fused into one kernel.
//
// examples/environments/digital_ads_market/digital_ads_market.py:525-596 as shipped: one exchange,
// one publisher, N advertisers, every connection present, a two-stage FiniteStateMachineEnv.  The
// messages of a step are then known in advance:
//   publisher stage   PUB -ImpressionRequest-> ADX -> every advertiser            (2 rounds, :164-165, :415-427)
//   advertiser stage  bids -> ADX: auction -> Ads -> PUB: click draw -> ImpressionResult -> winner,
//                     AuctionResult -> every bidder                               (3 rounds, :318-333, :429-516, :167-196)
// so the queues of the generic engine disappear: one thread per advertiser keeps the agent's
// attributes in registers, the exchange's handle_batch reduction is a workgroup arg-max fold over the
// bids ("first maximum in acting order" = the stable sorted(..., reverse=True)[0]), and the FSM
// observation / reward-cache epilogue (fsm.py:309-380) runs on the same registers.  ROLLOUT = T steps
// per launch with the random policy, the trajectory written as it is produced, and the caller's
// env.reset() (sampler redraw included) folded in; otherwise one step in phx_step_io's layout.
// phx_api.hip's derive() decides whether an env has this schedule (`ads_static`); everything else,
// tracking, host-injected messages and dynamic graphs stay on the generic engine, which is also the
// device-side cross-check of this kernel (tests: fused == generic == oracle).
#include <cstdlib>
#include <cstring>

#include "phx_dev.h"

struct AdsArgs {
  phx_step_io sio;
  phx_rollout_io rio;
};

struct AdsBid { double v; int tag; int r; };       // r < 0: none

__device__ __forceinline__ AdsBid ads_better(const AdsBid& x, const AdsBid& y) {
  if (x.r < 0) return y;
  if (y.r < 0) return x;
  const bool x_first = x.r < y.r;
  const AdsBid& lo = x_first ? x : y;
  const AdsBid& hi = x_first ? y : x;
  return t_lt(tv(lo.v, lo.tag), tv(hi.v, hi.tag)) ? hi : lo;
}

// first maximum in advertiser order over the whole workgroup; every thread gets the result
template <int NT>
__device__ __forceinline__ AdsBid ads_fold(AdsBid c, double* red_v, int* red_i) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    AdsBid o;
    o.v = __shfl_xor(c.v, off, 64); o.tag = __shfl_xor(c.tag, off, 64); o.r = __shfl_xor(c.r, off, 64);
    c = ads_better(c, o);
  }
  if (NT == 64) return c;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red_v[w] = c.v; red_i[2 * w] = c.tag; red_i[2 * w + 1] = c.r; }
  __syncthreads();
  AdsBid r; r.v = red_v[0]; r.tag = red_i[0]; r.r = red_i[1];
  for (int w = 1; w < NT / 64; ++w) { AdsBid o; o.v = red_v[w]; o.tag = red_i[2 * w]; o.r = red_i[2 * w + 1]; r = ads_better(r, o); }
  return r;
}

template <int NT>
__device__ __forceinline__ int ads_count(bool flag, int* red_i) {
  int c = __popcll(__ballot(flag));
  if (NT == 64) return c;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red_i[threadIdx.x >> 6] = c;
  __syncthreads();
  int n = 0;
  for (int w = 0; w < NT / 64; ++w) n += red_i[w];
  return n;
}

// REPLAY (rollouts): recorded actions and / or draws come from HBM (phx_rollout_io.actions / exo).  The device-drawn instantiation has NO
// global load on a step's common path -- the acting lists' masks are four bits in a register, the publisher's click probabilities sit in LDS,
// the two stages alternate by the static schedule's premise (phx_api.hip) -- because on gfx950 a load behind the row's stores waits for them
// (s_waitcnt vmcnt counts both), and a load behind a condition that is never true still leaves its wait at the join.
template <int NT, bool ROLLOUT, bool REPLAY>
__global__ __launch_bounds__(NT) void phx_ads_kernel(const DevSpec sp, const AdsArgs args) {
  __shared__ double red_v[NT / 64];
  __shared__ double s_pp[8];                                   // PublisherAgent's click probabilities [user - 1][theme], digital_ads_market.py:167-196
  __shared__ int red_i[2 * (NT / 64)];
  __shared__ int s_win[4];                                     // winner's aux, cost tag; second's r
  __shared__ double s_cost;

  const int b = xcd_block(!ROLLOUT), r = threadIdx.x;   // XCD-aware env mapping for the step kernel only (rollouts: 9.4 vs 9.0 us with it)
  const int N = sp.S;                                          // the advertisers are the strategic agents
  const bool mine = r < N;
  const int a = mine ? sp.strat_idx[r] : 0;
  const int64_t g = (int64_t)b * N + r;
  const int64_t genv = sp.env_offset + b;
  const int pub = sp.ads_pub, adx = sp.ads_adx;
  const int32_t* ppi = sp.param_i + pub * PHX_NPI;
  const int second = sp.param_i[adx * PHX_NPI + 1];
  const int theme = mine ? sp.param_i[a * PHX_NPI + 1] : 0;
  const int strong = mine ? sp.param_i[a * PHX_NPI + 2] : 0;
  const int tsrc = mine ? sp.type_src[a] : PHX_TYPE_CONST;
  const int pub_x = sp.exo_rank[pub];
  // StochasticNetwork with rates < 1 (and ignore_connection_errors, network.py:246-249): a message along a
  // connection that is off this episode is sent, tracked and dropped at the receiver (resolvers.py:146-148).
  // Connections of this thread's advertiser: to the exchange (c_adx) and to the publisher (c_pub).
  const bool dyn = sp.dynamic_graph != 0;
  int c_adx = 0, c_pub = 0, c_ap = 0;
  if (dyn) {
    if (mine) {
      const int e0 = sp.row_ptr[a];
      const bool first_is_adx = sp.col[e0] == adx;
      c_adx = sp.col_conn[first_is_adx ? e0 : e0 + 1]; c_pub = sp.col_conn[first_is_adx ? e0 + 1 : e0];
    }
    for (int e = sp.row_ptr[pub]; e < sp.row_ptr[pub + 1]; ++e) if (sp.col[e] == adx) c_ap = sp.col_conn[e];
  }
  const uint8_t* conn_b = dyn ? fld<uint8_t>(sp, F_NET_CONN_ON) + (int64_t)b * sp.n_conn : nullptr;
  bool e_adx = !dyn || !mine || conn_b[c_adx] != 0, e_pub = !dyn || !mine || conn_b[c_pub] != 0;
  bool e_ap = !dyn || conn_b[c_ap] != 0;

  // ---- state -> registers ------------------------------------------------------------------------
  int step = fld<int32_t>(sp, F_ENV_STEP)[b], stage = fld<int32_t>(sp, F_ENV_STAGE)[b], prev_stage = fld<int32_t>(sp, F_ENV_PREV_STAGE)[b];
  uint32_t tick = (uint32_t)fld<int32_t>(sp, F_ENV_TICK)[b];
  uint32_t episode = (sp.n_samplers > 0 || sp.n_conn > 0) ? (uint32_t)fld<int32_t>(sp, F_ENV_EPISODE)[b] : 0u;
  int n_resets = 0, err = 0, ads_seen = fld<int32_t>(sp, F_PUB_ADS_SEEN)[b];
  double left = 0, bid = 0, budget = 0, rc = 0;
  int left_tag = 0, bid_tag = 0, clicks = 0, wins = 0, user = 0, tq[3] = {0, 0, 0}, tw[3] = {0, 0, 0}, tc[3] = {0, 0, 0};
  uint8_t term = 1, trunc = 0, rcv = 0, ocv = 0;
  float oc[3] = {0.f, 0.f, 0.f}, lo[3] = {0.f, 0.f, 0.f};
  if (mine) {
    left = fld<double>(sp, F_ADV_LEFT)[g]; left_tag = fld<int32_t>(sp, F_ADV_LEFT_TAG)[g];
    bid = fld<double>(sp, F_ADV_BID)[g]; bid_tag = fld<int32_t>(sp, F_ADV_BID_TAG)[g];
    clicks = fld<int32_t>(sp, F_ADV_CLICKS)[g]; wins = fld<int32_t>(sp, F_ADV_WINS)[g]; user = fld<int32_t>(sp, F_ADV_USER)[g];
    for (int u = 0; u < 3; ++u) {
      tq[u] = fld<int32_t>(sp, F_ADV_TOT_REQUESTS)[g * 3 + u]; tw[u] = fld<int32_t>(sp, F_ADV_TOT_WINS)[g * 3 + u];
      tc[u] = fld<int32_t>(sp, F_ADV_TOT_CLICKS)[g * 3 + u];
    }
    term = fld<uint8_t>(sp, F_ENV_TERM)[g]; trunc = fld<uint8_t>(sp, F_ENV_TRUNC)[g];
    rc = fld<double>(sp, F_ENV_REW_CACHE)[g]; rcv = fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[g];
    ocv = fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID)[g];
    for (int d = 0; d < 3; ++d) oc[d] = fld<float>(sp, F_ENV_OBS_CACHE)[g * 3 + d];
    budget = tsrc >= 0 ? fld<double>(sp, F_ENV_SAMPLER)[(int64_t)b * sp.n_samplers + tsrc] : sp.param_f[a * PHX_NPF];
  }
  const int btag = strong ? PHX_TAG_F64 : PHX_TAG_PYF;
  // who observes / is rewarded after a step of list 0 / 1 (fsm.py:328-335): bits 0-1 / 2-3
  const int mbits = mine ? ((sp.obs_mask[a] ? 1 : 0) | (sp.obs_mask[(int64_t)sp.A + a] ? 2 : 0) | (sp.rew_mask[a] ? 4 : 0) | (sp.rew_mask[(int64_t)sp.A + a] ? 8 : 0)) : 0;
  if (r < 8) s_pp[r] = sp.param_f[pub * PHX_NPF + r];
  __syncthreads();
  // what an auto-reset reads -- the budget sampler's parameters, the three connection rates -- in registers too: the step loop holds no load
  double sprm[4] = {0.0, 0.0, 0.0, 0.0};
  if (tsrc >= 0) { const double* q = sp.sampler_param + 4 * tsrc; sprm[0] = q[0]; sprm[1] = q[1]; sprm[2] = q[2]; sprm[3] = q[3]; }
  const double cr_adx = (dyn && mine) ? sp.conn_rate[c_adx] : 0.0, cr_pub = (dyn && mine) ? sp.conn_rate[c_pub] : 0.0, cr_ap = dyn ? sp.conn_rate[c_ap] : 0.0;
  const int T = ROLLOUT ? args.rio.T : 1;
  RngQuadCache rq; rq.q = 0xffffffffu;
  const int64_t total = (int64_t)sp.B * N;

  for (int t = 0; t < T; ++t) {
    const bool live = mine && !(term | trunc);                 // _make_ctxs env.py:338-348
    const int list = stage;
    // ---- the policy's action -----------------------------------------------------------------------
    float action = 0.f; bool has = false;
    if (mine) {
      if (ROLLOUT) {
        if (REPLAY && args.rio.actions) action = args.rio.actions[(int64_t)t * total + g];
        else {                                                 // one Philox block serves four ticks of the agent
          uint32_t j; rng_quad_block(rq, sp.seed, genv, tick, r);
          rng_orders_from_block(rq.w, sp.seed, genv, tick, r, 0, nullptr, &j);
          action = (float)j * (1.0f / 274877.0f);
        }
        has = true;
      } else {
        has = args.sio.actions && (!args.sio.action_valid || args.sio.action_valid[g]);     // aid in actions, env.py:330
        if (has) action = args.sio.actions[g];
      }
    }
    const uint8_t* exo_b = !REPLAY ? nullptr
                         : ROLLOUT ? (args.rio.exo ? args.rio.exo + ((int64_t)t * sp.B + b) * sp.n_exo : nullptr)
                                   : (args.sio.exo ? args.sio.exo + (int64_t)b * sp.n_exo : nullptr);
    ++step;                                                    // env.py:252
    if (stage == sp.ads_pub_stage) {
      // PUB generate_messages :164-165 -> ADX forwards :415-427 -> live advertisers cache the user :249-271
      const int u = exo_b ? exo_b[pub_x] : rng_publisher(sp.seed, genv, tick, pub, 0, 0.0);
      if (live) {
        clicks = 0; wins = 0;                                  // pre_message_resolution :241-247
        if (e_ap && e_adx) { user = u; if (u >= 0 && u <= 2) tq[u] += 1; }
      }
      ads_seen = 0;
    } else {
      // decode_action :318-333
      AdsBid cand; cand.r = -1; cand.v = 0; cand.tag = 0;
      int my_aux = 0;
      if (live && has) {
        TVal bv = t_mul(tv((double)action, PHX_TAG_F32), tv(budget, btag));
        if (t_lt(tv(left, left_tag), bv)) bv = tv(left, left_tag);           // min(action[0] * budget, self.left)
        bid = bv.v; bid_tag = bv.tag;
        if (bv.v > 0.0 && e_adx) { cand.v = bv.v; cand.tag = bv.tag; cand.r = r; my_aux = theme | (user << 4) | (bv.tag << 8); }
      }
      if (live) { clicks = 0; wins = 0; }
      ads_seen = 0;
      const bool bidder = cand.r >= 0;
      // AdExchangeAgent.handle_batch + auction :429-516
      const AdsBid w = ads_fold<NT>(cand, red_v, red_i);
      if (w.r >= 0) {
        AdsBid c2 = cand; if (cand.r == w.r) c2.r = -1;
        AdsBid w2; w2.r = -1; w2.v = 0; w2.tag = 0;
        if (second) w2 = ads_fold<NT>(c2, red_v, red_i);          // sorted_bids[1], only the second-price rule reads it
        __syncthreads();
        if (r == w.r) { s_win[0] = my_aux & 0xffff; }
        const AdsBid& cm = (second && w2.r >= 0) ? w2 : w;       // second / first price :498-516
        if (r == cm.r) { s_cost = cm.v; s_win[1] = cm.tag; }
        __syncthreads();
        const int waux = s_win[0];
        // PublisherAgent.handle_ads :167-196
        const int wth = waux & 15, wuser = (waux >> 4) & 15;
        int clicked = 0; bool answered = false;
        if (!e_ap) { }                                                                     // the Ads message is dropped
        else if (wuser < 1 || wuser > 2 || wth > 3) { if (!err) err = PHX_ERR_CONTEXT; }  // dict KeyError :194
        else {
          ads_seen = 1;
          if (exo_b && ppi[1] < 1) { if (!err) err = PHX_ERR_QUEUE_FULL; }
          else {
            const double p = s_pp[(wuser - 1) * 4 + wth];
            clicked = exo_b ? exo_b[pub_x + 1] : rng_publisher(sp.seed, genv, tick, pub, 1, p);
            answered = true;
          }
        }
        // AuctionResult :273-282 (a loser's cost is the python float 0.0: left - 0.0 keeps value and kind)
        if (bidder && r == w.r) {
          wins += 1; if (user >= 0 && user <= 2) tw[user] += 1;
          const TVal nl = t_sub(tv(left, left_tag), tv(s_cost, s_win[1]));
          left = nl.v; left_tag = nl.tag;
          if (answered && e_pub) { clicks += clicked; if (user >= 0 && user <= 2) tc[user] += clicked; }   // ImpressionResult :284-292
        }
      }
    }
    // ---- FiniteStateMachineEnv epilogue fsm.py:309-380 ---------------------------------------------------
    const int next_stage = stage == 0 ? 1 : 0;                 // the static schedule's premise: stage_next = {1, 0} (phx_api.hip)
    uint8_t ov = 0, rv = 0, dv = 0, tm = 0;
    double rw = 0.0;
    float ob[3] = {0.f, 0.f, 0.f};
    if (live) {
      dv = 1;
      if (((mbits >> (list & 1)) & 1) && user != 0) {                           // :328-331, None when user 0
        ov = 1;
        ob[0] = (float)budget;
        ob[1] = (float)t_div(tv(left, left_tag), tv(budget, btag)).v;
        ob[2] = (float)(user - 1);
        oc[0] = ob[0]; oc[1] = ob[1]; oc[2] = ob[2]; ocv = 1;                             // self._observations.update :349
      }
      if ((mbits >> (2 + (list & 1))) & 1) { rc = (double)clicks; rcv = 1; }         // :334-335, :335-343
      tm = left <= 0.0 ? 1 : 0;                                                           // :345-349
      if (tm) term = 1;
    }
    const int nterm = ads_count<NT>(mine && term, red_i);
    const bool all_term = nterm == N;                                                     // env.py:308-310
    const bool all_trunc = step == sp.num_steps;                                          // env.py:312-318 (no agent truncates)
    const bool terminal = all_term || all_trunc;
    if (terminal) {                                                                       // fsm.py:360-375
      ov = ocv; ob[0] = ocv ? oc[0] : 0.f; ob[1] = ocv ? oc[1] : 0.f; ob[2] = ocv ? oc[2] : 0.f;
      rv = rcv ? 1 : 2; rw = rcv ? rc : 0.0;
    } else if (ov) { rv = rcv ? 1 : 2; rw = rcv ? rc : 0.0; }                             // fsm.py:378
    prev_stage = stage; stage = next_stage; ++tick;                                       // fsm.py:355
    if (mine) {
      if (ROLLOUT) {
        const phx_rollout_io& io = args.rio;
        const int64_t o = (int64_t)t * total + g;
        io.obs[o * 3 + 0] = ob[0]; io.obs[o * 3 + 1] = ob[1]; io.obs[o * 3 + 2] = ob[2];
        io.action_out[o] = action;
        io.reward[o] = (float)rw;
        io.terminated[o] = (uint8_t)(tm | all_term); io.truncated[o] = (uint8_t)all_trunc;
        if (io.obs_valid) io.obs_valid[o] = ov;
        if (io.reward_valid) io.reward_valid[o] = rv;
        lo[0] = ob[0]; lo[1] = ob[1]; lo[2] = ob[2];
      } else {
        const phx_step_io& io = args.sio;
        io.obs[g * 3 + 0] = ob[0]; io.obs[g * 3 + 1] = ob[1]; io.obs[g * 3 + 2] = ob[2];
        io.obs_valid[g] = ov; io.reward_valid[g] = rv; io.done_valid[g] = dv;
        io.terminated[g] = tm; io.truncated[g] = 0; io.reward[g] = rw;
      }
    }
    if (!ROLLOUT && r == 0) { args.sio.all_terminated[b] = all_term; args.sio.all_truncated[b] = all_trunc; }
    if (ROLLOUT && terminal) {
      // the caller's env.reset(): samplers env.py:211-212, agents :353-374, done sets, reward cache fsm.py:195-251
      if (tsrc >= 0) budget = rng_uniform(sp.seed, genv, episode, tsrc, sprm);
      if (dyn) {                                               // resample_connectivity network.py:438-447
        if (mine) { e_adx = rng_connection(sp.seed, genv, episode, c_adx, cr_adx) != 0;
                    e_pub = rng_connection(sp.seed, genv, episode, c_pub, cr_pub) != 0; }
        e_ap = rng_connection(sp.seed, genv, episode, c_ap, cr_ap) != 0;
      }
      ++episode; ++n_resets;
      left = budget; left_tag = btag; bid = 0.0; bid_tag = PHX_TAG_PYF; clicks = wins = user = 0;
      for (int u = 0; u < 3; ++u) tq[u] = tw[u] = tc[u] = 0;
      term = mine ? 0 : 1; trunc = 0; rcv = 0; step = 0; stage = sp.initial_stage;
      lo[0] = lo[1] = lo[2] = 0.f;                             // no advertiser observes at reset: user is 0
    }
  }

  // ---- registers -> state ------------------------------------------------------------------------------
  if (mine) {
    fld<double>(sp, F_ADV_LEFT)[g] = left; fld<int32_t>(sp, F_ADV_LEFT_TAG)[g] = left_tag;
    fld<double>(sp, F_ADV_BID)[g] = bid; fld<int32_t>(sp, F_ADV_BID_TAG)[g] = bid_tag;
    fld<int32_t>(sp, F_ADV_CLICKS)[g] = clicks; fld<int32_t>(sp, F_ADV_WINS)[g] = wins; fld<int32_t>(sp, F_ADV_USER)[g] = user;
    for (int u = 0; u < 3; ++u) {
      fld<int32_t>(sp, F_ADV_TOT_REQUESTS)[g * 3 + u] = tq[u]; fld<int32_t>(sp, F_ADV_TOT_WINS)[g * 3 + u] = tw[u];
      fld<int32_t>(sp, F_ADV_TOT_CLICKS)[g * 3 + u] = tc[u];
    }
    fld<uint8_t>(sp, F_ENV_TERM)[g] = term; fld<uint8_t>(sp, F_ENV_TRUNC)[g] = trunc;
    fld<double>(sp, F_ENV_REW_CACHE)[g] = rc; fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[g] = rcv;
    fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID)[g] = ocv;
    for (int d = 0; d < 3; ++d) fld<float>(sp, F_ENV_OBS_CACHE)[g * 3 + d] = oc[d];
    if (ROLLOUT && args.rio.last_obs) for (int d = 0; d < 3; ++d) args.rio.last_obs[g * 3 + d] = lo[d];
    if (ROLLOUT && dyn && n_resets > 0) {
      uint8_t* cw = fld<uint8_t>(sp, F_NET_CONN_ON) + (int64_t)b * sp.n_conn;
      cw[c_adx] = e_adx; cw[c_pub] = e_pub;
      if (r == 0) cw[c_ap] = e_ap;
    }
  }
  if (r == 0) {
    fld<int32_t>(sp, F_ENV_STEP)[b] = step; fld<int32_t>(sp, F_ENV_TICK)[b] = (int32_t)tick;
    fld<int32_t>(sp, F_ENV_STAGE)[b] = stage; fld<int32_t>(sp, F_ENV_PREV_STAGE)[b] = prev_stage;
    fld<int32_t>(sp, F_PUB_ADS_SEEN)[b] = ads_seen;
    int32_t* errp = ROLLOUT ? args.rio.err : args.sio.err;
    if (errp && errp[b] == 0 && err) errp[b] = err;
    if (ROLLOUT && (sp.n_samplers > 0 || sp.n_conn > 0) && n_resets > 0) fld<int32_t>(sp, F_ENV_EPISODE)[b] = (int32_t)episode;
    if (ROLLOUT && sp.n_samplers > 0 && n_resets > 0) {        // every column as drawn at the last auto-reset
      for (int j = 0; j < sp.n_samplers; ++j)
        fld<double>(sp, F_ENV_SAMPLER)[(int64_t)b * sp.n_samplers + j] =
            rng_uniform(sp.seed, genv, episode - 1u, j, sp.sampler_param + 4 * j);
    }
  }
}

// ---- launchers ------------------------------------------------------------------------------------------
template <bool ROLLOUT, bool REPLAY>
static hipError_t ads_launch(const DevSpec& sp, const AdsArgs& a, hipStream_t st) {
  const int n = sp.S;
  phx_note_kernel(ROLLOUT ? "phx_ads_kernel[rollout]" : "phx_ads_kernel[step]");
#define PHX_ADS(NT_) hipLaunchKernelGGL((phx_ads_kernel<NT_, ROLLOUT, REPLAY>), dim3(sp.B), dim3(NT_), 0, st, sp, a)
  if (n <= 64) PHX_ADS(64); else if (n <= 128) PHX_ADS(128); else if (n <= 256) PHX_ADS(256);
  else if (n <= 512) PHX_ADS(512); else PHX_ADS(1024);
#undef PHX_ADS
  return hipGetLastError();
}
hipError_t phx_launch_ads_step(const DevSpec& sp, const phx_step_io& io, hipStream_t st) {
  AdsArgs a; memset(&a, 0, sizeof a); a.sio = io;
  return ads_launch<false, true>(sp, a, st);
}
hipError_t phx_launch_ads_rollout(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st) {
  AdsArgs a; memset(&a, 0, sizeof a); a.rio = io;
  return (io.actions || io.exo) ? ads_launch<true, true>(sp, a, st) : ads_launch<true, false>(sp, a, st);
}

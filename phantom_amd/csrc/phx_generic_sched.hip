// phx_generic_sched.hip -- the message-passing engine for specs with a STATIC round schedule: several env instances per wave.
//
// Round 6.  phx_generic_step_kernel maps one env instance to one wave whose lanes are MESSAGES in the parallel
// phases and RECEIVERS in the handler phase: at SC64 ten of 64 lanes work while a shop walks its inbox, and every one of the
// ~15 dependent phases of a step costs its uniform control flow (49 % of the instructions were SALU) once per env.  For a spec
// whose message flow cannot drop or add a message (phx_api.hip: build_static_schedule -- static Network, supply-chain kinds,
// every agent live, every acting shop with an action) the whole flow is known at phx_create: which message sits at which
// queue position of which round, whose inbox it is in, at which position, and where the reply goes.  This kernel executes
// that COMPILED schedule:
//   * a lane is a STATEFUL RECEIVER (a shop) of one env instance; an env takes L = 8 / 16 / 32 / 64 lanes (the smallest power
//     of two that holds its shops), so a wave steps 64 / L env instances at once (SC64: 4, SC256: 1) and the schedule's
//     control flow -- rounds, batch lengths, queue offsets, all the same for every env -- is executed once per WAVE;
//   * Network.send / resolve (network.py:233-265) and BatchResolver.resolve (resolvers.py:128-163) are still what happens:
//     every message is materialised as its payload at its queue position in the env's LDS inbox (sender, receiver, type and
//     round of a position are static and live in the schedule table), every stateful receiver handles its batch one message
//     at a time in the reference's order (agents.py:96-155), stateless receivers (the factory) one lane per message, replies go
//     to the position the reference's send order gives them, and Resolver.push tracking (resolvers.py:41-42) writes the same
//     16-byte records in the same order;
//   * the receivers' state (stock, sales, missed_sales, delivered_stock) is in registers from the first load to the epilogue,
//     which writes observations / rewards / flags straight from them; in a rollout launch (the T-step loop) it stays there
//     from one step to the next, together with fsm.py's reward / observation caches;
//   * what the schedule's premise excludes is checked per env at entry (a done agent, an acting shop without an action): such
//     envs are flagged and stepped by the DYNAMIC engine in the same launch -- the grid's last ceil(B / 256) workgroups (the
//     "tail") wait for the flags of their 256 envs (one word per env, published within the first microsecond of the schedule
//     workgroups, all of which have been dispatched before a tail workgroup is) and run phx_generic_env over the flagged ones.
//     Usually none is, and the tail costs a load per env; a second launch (round 6's first form) cost 2.7 us per step.
// Reference: env.py:239-336, network.py:233-265, resolvers.py:128-163, agents.py:96-155, supply_chain.py:36-150, fsm.py:253-380.
// (compiled as part of phx_generic.hip's translation unit: phx_generic_env is defined there)
#include "phx_dev.h"

#define sp (*(const DevSpec*)spc)
#define g (*(const GenArgs*)(kp + 8))
#define GS_REFRESH() asm volatile("" : "+s"(spc), "+s"(kp))

// cross-lane hand-over through LDS inside ONE wave: LDS instructions of a wave execute in issue order, so all that is needed is
// that the compiler keeps the order (no s_barrier: an env never spans waves)
__device__ __forceinline__ void gs_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the customers' draws of one tick: group g's word (retry stream on a rejected word), dev_dev.h: rng_group_y
struct GsQuad { uint32_t w[4]; uint32_t q; };     // group 0's attempt-0 block of the shop's current tick quad (rollouts: 4 ticks per block)

// PURE (round 6, the T-step loop only): device-drawn actions and orders, no message log, no host-chosen or tabulated stage -- nothing in
// the loop reads global memory (the schedule, the queues and the rules are in LDS, per-stage words come through the scalar cache).  The
// general form's loads sit behind run-time conditions, but the s_waitcnt vmcnt(0) the compiler puts behind each at the joins is executed by
// every step, and loads and stores share that counter on gfx950: every step waited for the previous row's stores to be acknowledged.
template <int L, bool ROLL, bool PURE>
__global__ __launch_bounds__(256) void phx_sched_step_kernel(const DevSpec* __restrict__ spp_, const GenArgs g_) {
  phx_kptr_t spc = (phx_kptr_t)spp_;
  phx_kptr_t kp = (phx_kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  GS_REFRESH();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EPB = 256 / L;                                  // env instances per workgroup
  const int tid = threadIdx.x, j = tid & (L - 1), slot = tid / L;
  const int nS = sp.kind_count[PHX_KIND_SHOP], B = sp.B, S = sp.S;
  const int n_sched = (B + EPB - 1) / EPB;                       // schedule workgroups; the grid's rest is the tail
  if ((int)blockIdx.x >= n_sched) {
    // ---- tail: the env instances the schedule workgroups flag, on the dynamic engine ---------------------------------------------
    __shared__ int s_list[256];
    __shared__ int s_n;
    const int bt = ((int)blockIdx.x - n_sched) * 256 + tid;
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (bt < B) {
      // the env's word: 0 = its schedule workgroup has not decided yet, 2 | flagged once it has; consumed (zeroed) here, so that the next
      // launch -- or the next replay of a captured one -- starts from zeros again (no launch numbers: hipGraphs replay their arguments)
      int v = 0, spins = 0;
      do {                                                      // (bounded: a word that never arrives must not hang the GPU)
        v = __hip_atomic_load(sp.gs_dyn_flag + bt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != 0) break;
        __builtin_amdgcn_s_sleep(8);
      } while (++spins < (1 << 22));
      if (v != 0) __hip_atomic_store(sp.gs_dyn_flag + bt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v & 1) s_list[atomicAdd(&s_n, 1)] = bt;
    }
    __syncthreads();
    const int n_flagged = s_n;
    if (n_flagged == 0) return;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                     // the flagged envs' state is what earlier launches left: nothing of this launch precedes it
    for (int i = 0; i < n_flagged; ++i) {
      int bsel = s_list[0];                                      // ascending env order (the atomics handed the slots out in arrival order)
      for (int k = 1; k < n_flagged; ++k) bsel = (s_list[k] < bsel) ? s_list[k] : bsel;
      __syncthreads();
      if (tid == 0) for (int k = 0; k < n_flagged; ++k) if (s_list[k] == bsel) s_list[k] = 0x7fffffff;
      phx_generic_env<256, true, false, PHX_KIND_CUSTOMER, ROLL, false>(spc, kp, smem, bsel);
      __syncthreads();
    }
    return;
  }
  int wg = (int)blockIdx.x;
  { const unsigned n = (unsigned)n_sched, x = blockIdx.x & 7u, q = n >> 3, rem = n & 7u;       // xcd_block() over the schedule workgroups
    wg = (int)(x * q + (x < rem ? x : rem) + (blockIdx.x >> 3)); }
  const int b_raw = wg * EPB + slot;
  const bool in_range = b_raw < B;
  const int b = in_range ? b_raw : B - 1;                       // (lanes past the batch compute on the last env and store nothing)
  const bool shop = j < nS;
  const int jj = shop ? j : 0;
  const int64_t sb = (int64_t)b * nS + jj;                      // the lane's (env, shop): state and per-strategic-agent planes (S == nS, rank == j)

  // ---- the lane's state and the env's words: loads in flight while the schedule tables are staged ----------------------------
  int stock = fld<int32_t>(sp, F_SHOP_STOCK)[sb], sales = fld<int32_t>(sp, F_SHOP_SALES)[sb];
  int missed = fld<int32_t>(sp, F_SHOP_MISSED)[sb], delivered = fld<int32_t>(sp, F_SHOP_DELIVERED)[sb];
  const int done0 = fld<uint8_t>(sp, F_ENV_TERM)[sb] | fld<uint8_t>(sp, F_ENV_TRUNC)[sb];
  int w_step = fld<int32_t>(sp, F_ENV_STEP)[b], w_tick = fld<int32_t>(sp, F_ENV_TICK)[b], w_clock = fld<int32_t>(sp, F_ENV_CLOCK)[b];
  const int env_type = sp.env_type;
  const bool fsm = env_type == PHX_ENV_FSM;
  int w_stage = fsm ? fld<int32_t>(sp, F_ENV_STAGE)[b] : 0;
  // fsm.py's caches (self._rewards / self._observations, :334-350): registers for the launch
  double rc_val = 0.0; int rc_ok = 0; float oc[3] = {0.f, 0.f, 0.f}; int oc_ok = 0;
  if (fsm) {
    rc_val = fld<double>(sp, F_ENV_REW_CACHE)[sb]; rc_ok = fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[sb];
    oc_ok = fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID)[sb];
    const float* ocp = fld<float>(sp, F_ENV_OBS_CACHE) + sb * 3;
    oc[0] = ocp[0]; oc[1] = ocp[1]; oc[2] = ocp[2];
  }
  float act_in = 0.f; int act_has = 0;
  if (!ROLL) {
    const bool given = g.io.actions != nullptr;
    if (given) act_in = g.io.actions[sb];
    act_has = given && (!g.io.action_valid || g.io.action_valid[sb] != 0);             // aid in actions, env.py:330
  }

  // ---- the compiled schedules of every acting list -> LDS (one flat 16-byte copy; the same for every env) --------------------
  int32_t* const tab = (int32_t*)smem;
  const int words = sp.gs_words;
  {
    const uint4* src = (const uint4*)sp.gs_blob;
    uint4* dst = (uint4*)smem;
    for (int k = tid; k < (words + 3) / 4; k += 256) dst[k] = src[k];
  }
  int32_t* const q_env = tab + ((words + 3) & ~3) + slot * (2 * sp.gs_qstride);          // the env's two queues: payloads by queue position
  const int qstride = sp.gs_qstride;
  // the device-evaluated stage rules, [n_rules] DevRule behind the queues: the scan below ends per lane (envs of a wave may differ), so its
  // index is a VGPR to the compiler and sp.rules[rr] would be a vector load from global memory
  DevRule* const s_rules = (DevRule*)(smem + (((((words + 3) & ~3) + EPB * 2 * qstride) * 4 + 7) & ~7));
  for (int k = tid; k < sp.n_rules * (int)(sizeof(DevRule) / 4); k += 256) ((int32_t*)s_rules)[k] = ((const int32_t*)sp.rules)[k];
  int32_t* const s_nx = (int32_t*)(s_rules + sp.n_rules);        // FSM: [n_lists] stage_next | stage_rew_all << 16 (fsm.py:281-307,320)
  if (sp.env_type == PHX_ENV_FSM) for (int k = tid; k < sp.n_lists; k += 256) s_nx[k] = (sp.stage_next[k] & 0xFFFF) | (sp.stage_rew_all[k] ? 0x10000 : 0);
  __syncthreads();
  GS_REFRESH();

  // ---- the schedule's premise, per env (lane masks of the env's L lanes within the wave) --------------------------------------
  const unsigned long long seg = (L == 64) ? ~0ull : (((1ull << (L & 63)) - 1ull) << (((tid & 63) / L) * L));
  auto env_any = [&](bool c) { return (__ballot(c) & seg) != 0ull; };
  const int norm_i = shop ? sp.shop_norm[jj] : 1;
  const float norm_f = (float)norm_i;
  const uint64_t seed = sp.seed;
  const int64_t genv = sp.env_offset + b;
  const int num_steps = sp.num_steps;
  const int n_lists = sp.n_lists;
  const int32_t* const list_off = tab;                          // [n_lists] word offset of each list's program
  // a done strategic agent has no context (env.py:338-348): not this kernel's schedule
  bool dyn = env_any(shop && done0 != 0);
  if (fsm && (w_stage < 0 || w_stage >= n_lists)) dyn = true;
  if (!ROLL) {
    // an acting shop without an action sends nothing (env.py:330-333): not this kernel's schedule either
    const int l0 = fsm ? (dyn ? 0 : w_stage) : 0;
    const int acts0 = shop ? (tab[list_off[l0] + tab[list_off[l0] + 12] + jj] & 4) : 0;
    dyn = dyn || env_any(shop && acts0 && !act_has);
  }
  if (in_range && j == 0) __hip_atomic_store(sp.gs_dyn_flag + b, dyn ? 3 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool on = in_range && !dyn;                                    // this lane's env is stepped here

  GsQuad quad; quad.q = 0xffffffffu; quad.w[0] = quad.w[1] = quad.w[2] = quad.w[3] = 0u;
  float cur_ob[3] = {0.f, 0.f, 0.f};                             // ROLL: what the step-shaped observation buffer would hold (the fragment's last_obs)
  const int n_steps = ROLL ? g.roll_T : 1;
  const int trace_cap = sp.trace_cap;

  for (int it = 0; it < n_steps; ++it) {
    GS_REFRESH();
    const int64_t step_env = (ROLL ? (int64_t)it * B : 0) + b;   // row of the per-step [T][B][..] inputs / logs
    const uint32_t tick = (uint32_t)w_tick;
    const int t = w_step + 1;                                    // env.py:252
    const int cur_stage = w_stage;
    const uint8_t* exo_b = (!PURE && g.io.exo) ? g.io.exo + step_env * sp.n_exo : nullptr;
    phx_msg_rec* log_b = (!PURE && g.io.msg_log) ? g.io.msg_log + step_env * trace_cap : nullptr;

    // ---- the policy of a rollout (rollout.py:300-363): every strategic agent's action of the tick -- replayed, or the random policy
    //      = the shop's word of the tick (group 0's block, which also holds its first six customers' order sizes) ----------------
    float action = act_in; bool has_action = act_has != 0;
    uint32_t y0 = 0u; bool have_y0 = false;
    if (ROLL) {
      has_action = true;
      if (!PURE && g.roll_actions_in) action = g.roll_actions_in[((int64_t)(g.roll_t + it) * B + b) * S + jj];
      else {
        if ((tick >> 2) != quad.q) { rng_block(seed, genv, tick, jj, 0, 0, quad.w); quad.q = tick >> 2; }
        uint32_t jr;
        if (!rng_split(rng_pick(quad.w, tick), y0, jr)) y0 = rng_group_y(seed, genv, tick, jj, 0, 1, &jr);
        have_y0 = true;
        action = rng_j_to_action(jr);
      }
      if (on && shop) g.roll.action_out[((int64_t)(g.roll_t + it) * B + b) * S + jj] = action;
    }

    // ---- envs of one wave may stand in different stages (FSM): the schedule of each stage present, one after the other ----------
    int all_trunc = 0, next_stage = 0, err_code = 0;
    float ob[3] = {0.f, 0.f, 0.f}; double rw = 0.0; int ov = 0, rv = 0;
    unsigned long long todo = __ballot(on);
    while (todo != 0ull) {
      const int lead = __builtin_ctzll(todo);
      const int list = fsm ? __builtin_amdgcn_readlane(cur_stage, lead) : 0;
      const bool m = on && (!fsm || cur_stage == list);          // lanes whose env runs this list now
      todo &= ~__ballot(m);
      const int32_t* const P = tab + list_off[list];
      const int R = P[0], n_total = P[1], n0 = P[2];
      const int32_t* const sflags = P + P[12];
      const int sf = shop ? sflags[jj] : 0;                      // 1 observes, 2 rewarded, 4 acts in this list; 256 observes at reset
      int32_t* qc = q_env; int32_t* qn = q_env + qstride;

      // ---- _handle_acting_agents (env.py:320-336): decode_action / generate_messages of the list's agents into the round-0 queue --
      if (shop && m) {
        const int off = (P + P[4])[jj];
        if (off >= 0 && has_action) {                                          // ShopAgent.decode_action supply_chain.py:136-142 (the stock BEFORE the step)
          const int req = dev_round_half_even(action), room = PHX_SHOP_MAX_STOCK - stock;
          qc[off] = req < room ? req : room;
        }
        // CustomerAgent.generate_messages (:61-67) of the shop's acting customers, ascending customer index: customer k orders digit
        // k % 6 of group k / 6's word (rng_customer_order) or its recorded draw
        const int32_t* const cptr = P + P[5];
        const int c0 = cptr[jj], c1 = cptr[jj + 1];
        const int32_t* const cent = P + P[6];
        const int32_t* const cexo = P + P[7];
        int cur_g = -1, next_i = 0; uint32_t yrem = 0u;
        for (int ci = c0; ci < c1; ++ci) {
          const uint32_t e = (uint32_t)cent[ci];
          const int k = (int)(e & 0xffffu), off_c = (int)(e >> 16);
          int order;
          if (exo_b) order = exo_b[cexo[ci]];
          else {
            const int gq = k / 6, i = k - gq * 6;
            if (gq != cur_g || i < next_i) {
              yrem = (gq == 0 && have_y0) ? y0 : rng_group_y(seed, genv, tick, jj, gq, 0);
              cur_g = gq; next_i = 0;
            }
            for (; next_i < i; ++next_i) yrem = rng_div5(yrem);
            const uint32_t qd = rng_div5(yrem);
            order = (int)(yrem - 5u * qd); yrem = qd; next_i = i + 1;
          }
          qc[off_c] = order;
        }
      }
      // pre_message_resolution of every live agent (env.py:170-173; ShopAgent: supply_chain.py:93-96)
      if (m && shop) { sales = 0; missed = 0; }
      gs_wave_sync();
      const int32_t* const rec = sp.gs_rec + 2 * (int64_t)P[9];  // (sender | receiver << 16, type | round << 16) of every message of the step, log order
      if (log_b) {                                               // Resolver.push tracking, resolvers.py:41-42: round 0
        for (int i = j; i < n0; i += L)
          if (m && i < trace_cap) {
            const uint2 rr = *(const uint2*)(rec + 2 * i);
            uint4 o; o.x = rr.x; o.y = rr.y; const long long pv = (long long)qc[i]; o.z = (uint32_t)pv; o.w = (uint32_t)(pv >> 32);
            *(uint4*)(log_b + i) = o;
          }
      }

      // ---- BatchResolver.resolve (resolvers.py:128-163): the rounds of the compiled schedule ------------------------------------
      const int32_t* D = P + P[8];
      int log_n = n0;
      for (int r = 0; r < R; ++r, D += 8) {
        const int n_echo = D[1], cmax = D[3], n_next = D[7];
        // stateless receivers that answer (FactoryAgent.handle_stock_request, supply_chain.py:40-45): one lane per message
        const int32_t* const echo = P + D[2];
        for (int e = j; e < n_echo; e += L) { const uint32_t w = (uint32_t)echo[e]; qn[w >> 16] = qc[w & 0xffffu]; }
        // stateful receivers: the lane walks its batch in inbox order, one message at a time (agents.py:96-120)
        if (cmax > 0) {
          const uint32_t cw = (shop && m) ? (uint32_t)(P + D[4])[jj] : 0u;      // (lanes of envs that run another list now, or none: an empty batch)
          const int c = (int)(cw & 0xffffu);
          const int32_t* const ent = P + D[5] + (cw >> 16);
          for (int k0 = 0; k0 < cmax; k0 += 4) {
            uint32_t e4[4]; int v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) e4[u] = (k0 + u < c) ? (uint32_t)ent[k0 + u] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] = qc[e4[u] & 0xfffu];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if (k0 + u >= c) continue;
              const uint32_t e = e4[u];
              const int type = (int)(e >> 24), nx = (int)((e >> 12) & 0xfffu), v = v4[u];
              if (type == PHX_MSG_ORDER_REQUEST) {               // ShopAgent.handle_order_request supply_chain.py:105-122
                int sell;
                if (v > stock) { missed += v - stock; sell = stock; stock = 0; }
                else { sell = v; stock -= v; }
                sales += sell;
                if (nx != 0xfff) qn[nx] = sell;                  // OrderResponse(sell) to the customer
              } else {                                           // PHX_MSG_STOCK_RESPONSE: handle_stock_response :98-103
                delivered = v;
                const int ns2 = stock + v;
                stock = ns2 < PHX_SHOP_MAX_STOCK ? ns2 : PHX_SHOP_MAX_STOCK;
              }
            }
          }
        }
        gs_wave_sync();
        if (log_b) {                                             // the replies, in the order network.send was called (:156-158)
          for (int i = j; i < n_next; i += L)
            if (m && log_n + i < trace_cap) {
              const uint2 rr = *(const uint2*)(rec + 2 * (log_n + i));
              uint4 o; o.x = rr.x; o.y = rr.y; const long long pv = (long long)qn[i]; o.z = (uint32_t)pv; o.w = (uint32_t)(pv >> 32);
              *(uint4*)(log_b + log_n + i) = o;
            }
        }
        log_n += n_next;
        int32_t* const tq = qc; qc = qn; qn = tq;
      }
      if (m && j == 0 && g.io.msg_count) g.io.msg_count[step_env] = n_total;

      // ---- the stage the env enters (fsm.py:281-307): a device-evaluated rule, the host's choice, the tabulated handler, or next_stages[0] --
      int next_in = -1, nstage = 0;
      if (fsm) {
        bool by_rule = false;
        if (sp.n_rules > 0 && (PURE || !g.io.next_stage)) {
          for (int rr = 0; rr < sp.n_rules && !by_rule; ++rr) {
            const DevRule q = s_rules[rr];
            if (q.stage != list) continue;
            // (every rule field of a supply-chain spec is a ShopAgent attribute: i32, held in this lane's registers)
            int x = q.field_id == F_SHOP_STOCK ? stock : q.field_id == F_SHOP_SALES ? sales : q.field_id == F_SHOP_MISSED ? missed : delivered;
            long long sum;
            if (q.col >= 0) { sum = __shfl(shop && j == q.col ? x : 0, ((tid & 63) / L) * L + q.col, 64); }
            else {
              int part = shop ? x : 0;
#pragma unroll
              for (int off = 1; off < L; off <<= 1) part += __shfl_xor(part, off, 64);
              sum = part;
            }
            const double v = (double)sum;
            const bool hit = q.cmp == PHX_CMP_LT ? v < q.threshold : q.cmp == PHX_CMP_LE ? v <= q.threshold : q.cmp == PHX_CMP_GT ? v > q.threshold :
                             q.cmp == PHX_CMP_GE ? v >= q.threshold : q.cmp == PHX_CMP_EQ ? v == q.threshold : v != q.threshold;
            // (the first rule of the stage that holds decides; envs of one wave may differ: per-lane result, uniform scan of the rules)
            if (hit && next_in < 0) next_in = q.next_stage;
          }
          by_rule = next_in >= 0;
        }
        if (!PURE && !by_rule && (g.io.next_stage || sp.stage_tab)) {
          next_in = g.io.next_stage ? g.io.next_stage[b] : sp.stage_tab[(int64_t)list * (num_steps + 1) + (t <= num_steps ? t : num_steps)];
          if (next_in < 0 || next_in >= n_lists || !sp.stage_allowed[(int64_t)list * n_lists + next_in]) {
            if (m) err_code = PHX_ERR_FSM_TRANSITION;            // FSMRuntimeError, after the resolution (fsm.py:304-307)
            next_in = -1;
          }
        }
        nstage = next_in >= 0 ? next_in : (s_nx[list] & 0xFFFF);
      }

      // ---- encode_observation / compute_reward / is_done of the strategic agents (env.py:273-301; fsm.py:309-380) -----------------
      if (m) {
        next_stage = nstage;
        all_trunc = (t == num_steps) ? 1 : 0;                     // env.py:312-318 (no ShopAgent terminates or truncates: agents.py:292-323)
        if (shop) {
          // who observes: the list's mask, or after a handler-chosen transition the agents acting in the NEXT stage (fsm.py:320)
          int obs_bit = sf & 1;
          if (fsm && next_in >= 0 && !(s_nx[list] >> 16)) obs_bit = ((tab + list_off[nstage])[(tab + list_off[nstage])[12] + jj] >> 2) & 1;
          ov = obs_bit; rv = 0; rw = 0.0; ob[0] = ob[1] = ob[2] = 0.f;
          if (ov) {
            if ((((unsigned)stock + (1u << 24)) | ((unsigned)sales + (1u << 24)) | ((unsigned)missed + (1u << 24)) | ((unsigned)norm_i + (1u << 24))) < (2u << 24))
              shop_obs_f32(stock, sales, missed, norm_f, ob);
            else shop_obs(stock, sales, missed, norm_i, ob);
          }
          if (env_type == PHX_ENV_PLAIN) { if (ov) { rw = shop_reward(sales, stock); rv = 1; } }      // env.py:283
          else {
            bool cached_now = false;
            if (sf & 2) { rc_val = shop_reward(sales, stock); rc_ok = 1; cached_now = true; }         // self._rewards[aid] = ... fsm.py:334-350
            if (ov) { oc[0] = ob[0]; oc[1] = ob[1]; oc[2] = ob[2]; oc_ok = 1; }                      // self._observations.update :349
            if (ov) { const bool cv = cached_now || rc_ok != 0; rv = cv ? 1 : 2; rw = cv ? rc_val : 0.0; }   // :378
            if (all_trunc) {                                      // the terminal dump of the cached dicts, :360-375
              ov = oc_ok;
              ob[0] = oc_ok ? oc[0] : 0.f; ob[1] = oc_ok ? oc[1] : 0.f; ob[2] = oc_ok ? oc[2] : 0.f;
              rv = rc_ok ? 1 : 2; rw = rc_ok ? rc_val : 0.0;
            }
          }
        }
      }
    }   // lists present in the wave

    // ---- the step's outputs ------------------------------------------------------------------------------------------------------
    w_step = t; w_tick = (int)(tick + 1);
    if (on) {
      // (clock: one tick per delivered message, resolvers.py -- every message of the schedule is delivered)
      const int32_t* const P = tab + list_off[fsm ? cur_stage : 0];
      w_clock += P[1];
    }
    if (!ROLL) {
      if (on && shop) {
        float* o = g.io.obs + sb * 3;
        o[0] = ob[0]; o[1] = ob[1]; o[2] = ob[2];
        g.io.reward[sb] = rw;
        g.io.obs_valid[sb] = (uint8_t)ov; g.io.reward_valid[sb] = (uint8_t)rv; g.io.done_valid[sb] = 1;
        g.io.terminated[sb] = 0; g.io.truncated[sb] = 0;
      }
      if (on && j == 0) {
        g.io.all_terminated[b] = 0; g.io.all_truncated[b] = (uint8_t)all_trunc;
        if (g.io.err && err_code && g.io.err[b] == 0) g.io.err[b] = err_code;
      }
    } else {
      if (on && shop) {                                          // the trajectory row, rollout.py:361-389 (flags with the env's __all__ ORed in)
        const int64_t o = ((int64_t)(g.roll_t + it) * B + b) * S + jj;
        float* po = g.roll.obs + o * 3;
        po[0] = ob[0]; po[1] = ob[1]; po[2] = ob[2];
        g.roll.reward[o] = (float)rw;
        if (g.roll.terminated) g.roll.terminated[o] = 0;
        g.roll.truncated[o] = (uint8_t)all_trunc;
        if (g.roll.obs_valid) g.roll.obs_valid[o] = (uint8_t)ov;
        if (g.roll.reward_valid) g.roll.reward_valid[o] = (uint8_t)rv;
      }
      if (!PURE && on && j == 0 && g.io.err && err_code && g.io.err[b] == 0) g.io.err[b] = err_code;
      cur_ob[0] = ob[0]; cur_ob[1] = ob[1]; cur_ob[2] = ob[2];
    }
    const int prev_stage = cur_stage;
    w_stage = next_stage;
    if (ROLL && all_trunc) {
      // the caller's env.reset() (env.py:185-237; fsm.py:195-251): Agent.reset zeroes the stock (supply_chain.py:149-150), the reward cache
      // is cleared (:234), the agents that act in the initial stage observe their reset state (the last step's sales stay, App. B)
      stock = 0; rc_ok = 0;
      w_step = 0; w_stage = sp.initial_stage;
      const int rs = shop ? (tab + list_off[0])[(tab + list_off[0])[12] + jj] : 0;
      cur_ob[0] = cur_ob[1] = cur_ob[2] = 0.f;
      if (rs & 256) {
        if ((((unsigned)sales + (1u << 24)) | ((unsigned)missed + (1u << 24)) | ((unsigned)norm_i + (1u << 24))) < (2u << 24)) shop_obs_f32(0, sales, missed, norm_f, cur_ob);
        else shop_obs(0, sales, missed, norm_i, cur_ob);
      }
    }
    // ---- state back to the blob: after a phx_step, after the last step of a rollout launch -------------------------------------------
    if (it == n_steps - 1 && on) {
      if (shop) {
        fld<int32_t>(sp, F_SHOP_STOCK)[sb] = stock; fld<int32_t>(sp, F_SHOP_SALES)[sb] = sales;
        fld<int32_t>(sp, F_SHOP_MISSED)[sb] = missed; fld<int32_t>(sp, F_SHOP_DELIVERED)[sb] = delivered;
        if (fsm) {
          fld<double>(sp, F_ENV_REW_CACHE)[sb] = rc_val; fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID)[sb] = (uint8_t)rc_ok;
          fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID)[sb] = (uint8_t)oc_ok;
          float* ocp = fld<float>(sp, F_ENV_OBS_CACHE) + sb * 3;
          ocp[0] = oc[0]; ocp[1] = oc[1]; ocp[2] = oc[2];
        }
        if (ROLL && g.roll.last_obs) { float* lo = g.roll.last_obs + sb * 3; lo[0] = cur_ob[0]; lo[1] = cur_ob[1]; lo[2] = cur_ob[2]; }
      }
      if (j == 0) {
        fld<int32_t>(sp, F_ENV_STEP)[b] = w_step; fld<int32_t>(sp, F_ENV_TICK)[b] = w_tick; fld<int32_t>(sp, F_ENV_CLOCK)[b] = w_clock;
        if (fsm) { fld<int32_t>(sp, F_ENV_STAGE)[b] = w_stage; fld<int32_t>(sp, F_ENV_PREV_STAGE)[b] = prev_stage; }      // fsm.py:355 (a reset leaves it)
      }
    }
  }
}

#undef sp
#undef g
#undef GS_REFRESH

// ---- host: the compiled schedule ------------------------------------------------------------------------------------------------
// Program of one acting list (int32 words; offsets relative to the list's first word):
//   [0] R  [1] messages of the step  [2] n_0  [3] shops  [4] -> req_off[shops] (queue position of the shop's StockRequest, -1: it does not act)
//   [5] -> cust_ptr[shops + 1]  [6] -> cust_ent[]: customer index k | queue position << 16, ascending k per shop  [7] -> cust_exo[]: exogenous column
//   [8] -> R x 8 round words  [9] first record of the list in gs_rec  [10] words  [11] most customers of one shop  [12] -> shop flags[shops]
//   round: [0] n_r  [1] echoes  [2] -> echo[]: source position | reply position << 16  [3] longest stateful batch
//          [4] -> cnt[shops]: batch length | first entry << 16  [5] -> ent[]: source position (12) | reply position (12, 0xFFF none) | type << 24
//          [6] (unused)  [7] n_(r+1)
bool phx_sched_compile(const phx_spec* spec, int A, int n_lists, const int32_t* act_ptr, const int32_t* act_idx, const uint8_t* act_mask,
                       const uint8_t* obs_mask, const uint8_t* rew_mask, const int32_t* kind_rank, const int32_t* exo_rank, const int32_t* strat_rank,
                       const int32_t* reset_obs_idx, int n_reset_obs, std::vector<int32_t>* blob, std::vector<int32_t>* recs, int* L_out, int* qmax_out) {
  blob->clear(); recs->clear();
  if ((spec->flags & (PHX_F_SHUFFLE_BATCHES | PHX_F_IGNORE_CONN_ERRORS)) || spec->n_conn > 0 || spec->n_samplers > 0) return false;
  if (spec->env_type != PHX_ENV_PLAIN && spec->env_type != PHX_ENV_FSM) return false;
  int nS = 0;
  std::vector<int> shop_of_rank;
  for (int a = 0; a < A; ++a) {
    const int k = spec->kind[a];
    if (k != PHX_KIND_FACTORY && k != PHX_KIND_SHOP && k != PHX_KIND_CUSTOMER) return false;
    if (k == PHX_KIND_SHOP) { if (kind_rank[a] != nS || strat_rank[a] != nS) return false; shop_of_rank.push_back(a); ++nS; }
    else if (strat_rank[a] >= 0) return false;
  }
  if (nS < 1 || nS > 64) return false;
  int L = 8; while (L < nS) L <<= 1;
  auto edge = [&](int u, int v) { for (int e = spec->row_ptr[u]; e < spec->row_ptr[u + 1]; ++e) if (spec->col[e] == v) return true; return false; };
  auto payload_ok = [&](int src, int dst, int type) {
    if (spec->flags & PHX_F_NO_PAYLOAD_CHECKS) return true;
    int sk = 0, rk = 0;
    switch (type) {
      case PHX_MSG_ORDER_REQUEST: sk = PHX_KIND_CUSTOMER; rk = PHX_KIND_SHOP; break;
      case PHX_MSG_ORDER_RESPONSE: sk = PHX_KIND_SHOP; rk = PHX_KIND_CUSTOMER; break;
      case PHX_MSG_STOCK_REQUEST: sk = PHX_KIND_SHOP; rk = PHX_KIND_FACTORY; break;
      case PHX_MSG_STOCK_RESPONSE: sk = PHX_KIND_FACTORY; rk = PHX_KIND_SHOP; break;
      default: return false;
    }
    return spec->kind[src] == sk && spec->kind[dst] == rk;
  };
  std::vector<uint8_t> in_reset(A, 0);
  for (int k = 0; k < n_reset_obs; ++k) in_reset[reset_obs_idx[k]] = 1;
  struct M { int src, dst, type; };
  blob->assign((size_t)((n_lists + 3) & ~3), 0);
  int qmax = 1;
  for (int l = 0; l < n_lists; ++l) {
    std::vector<int32_t> P(16, 0);
    std::vector<M> q;
    std::vector<int32_t> req_off(nS, -1);
    struct C { int k, off, exo; };
    std::vector<std::vector<C>> cust(nS);
    for (int kk = act_ptr[l]; kk < act_ptr[l + 1]; ++kk) {
      const int a = act_idx[kk], kind = spec->kind[a], dst = spec->param_i[a * PHX_NPI];
      if (kind == PHX_KIND_SHOP) {
        if (dst < 0 || dst >= A || !edge(a, dst) || !payload_ok(a, dst, PHX_MSG_STOCK_REQUEST)) return false;
        req_off[kind_rank[a]] = (int32_t)q.size(); q.push_back({a, dst, PHX_MSG_STOCK_REQUEST});
      } else if (kind == PHX_KIND_CUSTOMER) {
        if (dst < 0 || dst >= A || spec->kind[dst] != PHX_KIND_SHOP || !edge(a, dst) || !payload_ok(a, dst, PHX_MSG_ORDER_REQUEST)) return false;
        const int k = spec->param_i[a * PHX_NPI + 1];
        if (k < 0 || k > 0xffff || exo_rank[a] < 0) return false;
        cust[kind_rank[dst]].push_back({k, (int)q.size(), exo_rank[a]}); q.push_back({a, dst, PHX_MSG_ORDER_REQUEST});
      }
    }
    if (q.size() > 4094) return false;
    std::vector<int32_t> cptr(1, 0), cent, cexo;
    int kmax = 0;
    for (int s = 0; s < nS; ++s) {
      std::stable_sort(cust[s].begin(), cust[s].end(), [](const C& x, const C& y) { return x.k < y.k; });
      for (size_t i = 1; i < cust[s].size(); ++i) if (cust[s][i].k == cust[s][i - 1].k) return false;      // two customers drawing the same digit
      for (const C& c : cust[s]) { cent.push_back((int32_t)((uint32_t)c.k | ((uint32_t)c.off << 16))); cexo.push_back(c.exo); }
      cptr.push_back((int32_t)cent.size());
      kmax = std::max(kmax, (int)cust[s].size());
    }
    std::vector<int32_t> sfl(nS, 0);
    for (int s = 0; s < nS; ++s) {
      const int a = shop_of_rank[s];
      sfl[s] = (obs_mask[(size_t)l * A + a] ? 1 : 0) | (rew_mask[(size_t)l * A + a] ? 2 : 0) | (act_mask[(size_t)l * A + a] ? 4 : 0) | (in_reset[a] ? 256 : 0);
    }
    const int rec0 = (int)(recs->size() / 2);
    auto log_queue = [&](const std::vector<M>& qq, int round) {
      for (const M& m : qq) { recs->push_back((int32_t)((uint32_t)m.src | ((uint32_t)m.dst << 16))); recs->push_back((int32_t)((uint32_t)m.type | ((uint32_t)round << 16))); }
    };
    log_queue(q, 0);
    std::vector<int32_t> rounds, tails;                          // 8 words per round; the rounds' variable-length tables behind them
    const int n0 = (int)q.size();
    int R = 0, n_total = n0;
    qmax = std::max(qmax, n0);
    std::vector<std::vector<int32_t>> rt;                        // per round: echo, cnt, ent
    struct RD { int n, n_echo, cmax, n_next; std::vector<int32_t> echo, cnt, ent; };
    std::vector<RD> rds;
    while (!q.empty()) {
      if (R == PHX_SCHED_MAX_ROUNDS || (spec->round_limit >= 0 && R >= spec->round_limit)) return false;
      const int n = (int)q.size();
      std::vector<int32_t> cnt(A, 0), first(A, 0x7fffffff), goff(A, 0), order(n, 0), fill(A, 0);
      for (int i = 0; i < n; ++i) { cnt[q[i].dst]++; first[q[i].dst] = std::min(first[q[i].dst], i); }
      int run = 0;
      for (int i = 0; i < n; ++i) if (first[q[i].dst] == i) { goff[q[i].dst] = run; run += cnt[q[i].dst]; }   // receivers in first-arrival (dict) order
      for (int i = 0; i < n; ++i) order[goff[q[i].dst] + fill[q[i].dst]++] = i;                               // batches in send order
      std::vector<M> nq;
      std::vector<int32_t> next_off(n, -1);
      for (int Pq = 0; Pq < n; ++Pq) {                            // replies in handling order (resolvers.py:142-158)
        const M m = q[order[Pq]];
        const int rk = spec->kind[m.dst];
        if (rk == PHX_KIND_FACTORY && m.type == PHX_MSG_STOCK_REQUEST) {
          if (!payload_ok(m.dst, m.src, PHX_MSG_STOCK_RESPONSE)) return false;
          next_off[Pq] = (int32_t)nq.size(); nq.push_back({m.dst, m.src, PHX_MSG_STOCK_RESPONSE});
        } else if (rk == PHX_KIND_SHOP && m.type == PHX_MSG_ORDER_REQUEST) {
          if (!payload_ok(m.dst, m.src, PHX_MSG_ORDER_RESPONSE)) return false;
          next_off[Pq] = (int32_t)nq.size(); nq.push_back({m.dst, m.src, PHX_MSG_ORDER_RESPONSE});
        } else if ((rk == PHX_KIND_SHOP && m.type == PHX_MSG_STOCK_RESPONSE) || (rk == PHX_KIND_CUSTOMER && m.type == PHX_MSG_ORDER_RESPONSE)) {
        } else return false;                                      // no handler: the dynamic engine reports it
      }
      if (nq.size() > 4094) return false;
      RD rd; rd.n = n; rd.n_next = (int)nq.size(); rd.cmax = 0;
      for (int Pq = 0; Pq < n; ++Pq) {
        const M m = q[order[Pq]];
        if (spec->kind[m.dst] == PHX_KIND_FACTORY) rd.echo.push_back((int32_t)((uint32_t)order[Pq] | ((uint32_t)next_off[Pq] << 16)));
      }
      rd.n_echo = (int)rd.echo.size();
      for (int s = 0; s < nS; ++s) {
        const int a = shop_of_rank[s], c = cnt[a];
        if (c > 0xffff || rd.ent.size() > 0xffff) return false;
        rd.cnt.push_back((int32_t)((uint32_t)c | ((uint32_t)rd.ent.size() << 16)));
        for (int k = 0; k < c; ++k) {
          const int Pq = goff[a] + k;
          const M m = q[order[Pq]];
          rd.ent.push_back((int32_t)((uint32_t)order[Pq] | ((uint32_t)(next_off[Pq] < 0 ? 0xfff : next_off[Pq]) << 12) | ((uint32_t)m.type << 24)));
        }
        rd.cmax = std::max(rd.cmax, c);
      }
      for (int pad = 0; pad < 4; ++pad) rd.ent.push_back(0);      // (the batch walk reads entries four at a time)
      rds.push_back(rd);
      log_queue(nq, R + 1);
      n_total += (int)nq.size();
      qmax = std::max(qmax, (int)nq.size());
      ++R; q.swap(nq);
    }
    // assemble
    auto put = [&](const std::vector<int32_t>& v) { const int o = (int)P.size(); P.insert(P.end(), v.begin(), v.end()); return o; };
    P[0] = R; P[1] = n_total; P[2] = n0; P[3] = nS;
    P[4] = put(req_off); P[5] = put(cptr); P[6] = put(cent.empty() ? std::vector<int32_t>(1, 0) : cent); P[7] = put(cexo.empty() ? std::vector<int32_t>(1, 0) : cexo);
    P[12] = put(sfl);
    const int o_rounds = (int)P.size();
    P[8] = o_rounds;
    P.resize(P.size() + 8 * (size_t)std::max(R, 1), 0);
    for (int r = 0; r < R; ++r) {
      const RD& rd = rds[r];
      const int oe = put(rd.echo.empty() ? std::vector<int32_t>(1, 0) : rd.echo), oc = put(rd.cnt), on_ = put(rd.ent);
      int32_t* Dw = P.data() + o_rounds + 8 * r;
      Dw[0] = rd.n; Dw[1] = rd.n_echo; Dw[2] = oe; Dw[3] = rd.cmax; Dw[4] = oc; Dw[5] = on_; Dw[6] = 0; Dw[7] = rd.n_next;
    }
    P[9] = rec0; P[11] = kmax;
    while (P.size() & 3) P.push_back(0);
    P[10] = (int32_t)P.size();
    (*blob)[l] = (int32_t)blob->size();
    blob->insert(blob->end(), P.begin(), P.end());
  }
  *L_out = L; *qmax_out = qmax;
  return true;
}

// (tables + the workgroup's queues + the LDS copies of the rules and of the per-stage words)
size_t phx_sched_lds_bytes(int words, int L, int qstride, int n_rules, int n_lists) {
  return (size_t)((words + 3) & ~3) * 4 + (size_t)(256 / L) * 2 * (size_t)qstride * 4 + 8 + (size_t)n_rules * sizeof(DevRule) + (size_t)n_lists * 4;
}

hipError_t phx_launch_sched(const DevSpec& sp, const GenArgs& g, hipStream_t st) {
  const int L = sp.gs_L, EPB = 256 / L;
  const dim3 grid((unsigned)((sp.B + EPB - 1) / EPB + (sp.B + 255) / 256));      // schedule workgroups + tail
  // LDS: the schedule's tables and queues, or what the dynamic engine needs for one env (the tail workgroups), whichever is larger
  const size_t lds = std::max(phx_sched_lds_bytes(sp.gs_words, L, sp.gs_qstride, sp.n_rules, sp.n_lists),
                              (phx_generic_queue_bytes(sp.A, sp.S, sp.queue_cap, sp.scan_cap, 0, false) + 15) & ~(size_t)15);
  const bool roll = g.roll_t >= 0;
  phx_note_kernel(roll ? "phx_sched_step_kernel[T-step loop]" : "phx_sched_step_kernel");
  const bool pure = roll && !g.roll_actions_in && !g.io.exo && !g.io.msg_log && !g.io.next_stage && !sp.stage_tab;
#define GS_LAUNCH(L_) do { if (pure) hipLaunchKernelGGL((phx_sched_step_kernel<L_, true, true>), grid, dim3(256), lds, st, sp.self_dev, g); \
                           else if (roll) hipLaunchKernelGGL((phx_sched_step_kernel<L_, true, false>), grid, dim3(256), lds, st, sp.self_dev, g); \
                           else hipLaunchKernelGGL((phx_sched_step_kernel<L_, false, false>), grid, dim3(256), lds, st, sp.self_dev, g); } while (0)
  if (L == 8) GS_LAUNCH(8); else if (L == 16) GS_LAUNCH(16); else if (L == 32) GS_LAUNCH(32); else GS_LAUNCH(64);
#undef GS_LAUNCH
  return hipGetLastError();
}

// phx_epilogue.h -- the tail of PhantomEnv.step shared by the generic engine and the fused
// Stackelberg kernel: per strategic agent obs / reward / done with the PLAIN, FSM or Stackelberg
// masks and reward-cache semantics, __all__ flags, clock/stage update.
//   env.py:273-301 ; fsm.py:309-380 ; stackelberg.py:142-196
#pragma once
#include "phx_dev.h"

// One workgroup per env.  `live` = agent has a context this step (NULL: every agent is live).
// `roll` (a rollout's step, S <= NT): the step's outputs also go to the trajectory row at element offset row_o = (t_row * B + b) * S
// straight from the lane's registers (rollout.py:361-389; the flags with the env's __all__ ORed in, as the copy loop of the caller
// does otherwise); returns true when the row has been written.
// s_nterm / s_ntrunc: zero-initialised LDS counters.  Contains a workgroup barrier.
// `encode_obs(a, t, ob)` is the agent's encode_observation; the fused Stackelberg kernel passes one
// that reads the buyers' prices from its LDS copy of seller.posted.
template <int NT, typename ObsFn>
__device__ __forceinline__ bool strategic_epilogue(const DevSpec& sp, const Topo& tp, const phx_step_io& io, int b, int t,
                                                   int list, int cur_stage, uint32_t tick,
                                                   const uint8_t* live, int* s_nterm, int* s_ntrunc, ObsFn encode_obs,
                                                   int next_in = -1, const phx_rollout_io* roll = nullptr, int64_t row_o = 0, int* all_flags = nullptr) {
  const int tid = threadIdx.x;
  const int A = sp.A, S = sp.S, D = sp.D;
  uint8_t* term = fld<uint8_t>(sp, F_ENV_TERM) + (int64_t)b * S;
  uint8_t* trunc = fld<uint8_t>(sp, F_ENV_TRUNC) + (int64_t)b * S;
  // next stage: the handler's choice (next_in >= 0, already validated) or next_stages[0] (fsm.py:281-307); the
  // agents acting in THAT stage observe (fsm.py:320) unless rewarded_agents is None (every strategic agent, :315-317)
  const int next_stage = (sp.env_type == PHX_ENV_FSM) ? (next_in >= 0 ? next_in : sp.stage_next[cur_stage]) : 0;
  const uint8_t* obs_mask = (sp.env_type == PHX_ENV_FSM && next_in >= 0 && !sp.stage_rew_all[cur_stage])
                                ? sp.act_mask + (int64_t)next_stage * A : sp.obs_mask + (int64_t)list * A;
  const uint8_t* rew_mask = sp.rew_mask + (int64_t)list * A;
  float* obs_b = io.obs + (int64_t)b * S * D;
  double* rew_cache = fld<double>(sp, F_ENV_REW_CACHE) + (int64_t)b * S;
  uint8_t* rew_cache_v = fld<uint8_t>(sp, F_ENV_REW_CACHE_VALID) + (int64_t)b * S;
  float* obs_cache = fld<float>(sp, F_ENV_OBS_CACHE) + (int64_t)b * S * D;
  uint8_t* obs_cache_v = fld<uint8_t>(sp, F_ENV_OBS_CACHE_VALID) + (int64_t)b * S;

  // One pass writes the step's outputs as they stand when the step is not the env's last (the reward an FSM / Stackelberg
  // env emits is the cached one, read or just computed by this lane); only a terminal step -- once per episode -- takes
  // the second pass that dumps the cached dicts.  (The second pass used to run every step and re-read from HBM what the
  // first had just written, behind a full fence.)
  const bool row = roll != nullptr && S <= NT;                 // one strategic agent per lane: its outputs stay in registers
  float k_ob[4] = {0.f, 0.f, 0.f, 0.f}; double k_rw = 0.0; uint8_t k_ov = 0, k_rv = 0, k_tm = 0, k_tr = 0;
  for (int s = tid; s < S; s += NT) {                          // env.py:273 / fsm.py:320 / stackelberg.py:150
    const int a = sp.strat_idx[s];
    const int64_t o = (int64_t)b * S + s;
    const uint8_t term_old = term[s], trunc_old = trunc[s];
    uint8_t ov = 0, rv = 0, dv = 0, tm = 0, tr = 0;
    double rw = 0.0;
    float ob[4] = {0.f, 0.f, 0.f, 0.f};
    bool cached_now = false; double rc = 0.0;
    if (!live || live[a]) {                                    // env.py:274-275
      dv = 1;
      if (obs_mask[a]) ov = encode_obs(a, t, ob) ? 1 : 0;      // `if obs is not None` env.py:279-280
      if (sp.env_type == PHX_ENV_PLAIN) { if (ov) { rw = dev_compute_reward(sp, tp, b, a); rv = 1; } }   // env.py:283
      else if (rew_mask[a]) { rc = dev_compute_reward(sp, tp, b, a); cached_now = true; rew_cache[s] = rc; rew_cache_v[s] = 1; }
      tm = dev_is_terminated(sp, tp, b, a, t) ? 1 : 0;         // env.py:285-286
      tr = dev_is_truncated(sp, tp, a, t) ? 1 : 0;
      if (tm) term[s] = 1;                                     // :288-292
      if (tr) trunc[s] = 1;
    }
    if (tm | term_old) atomicAdd(s_nterm, 1);
    if (tr | trunc_old) atomicAdd(s_ntrunc, 1);
    if (sp.env_type == PHX_ENV_FSM && ov) {                    // self._observations.update, fsm.py:349
      for (int d = 0; d < D; ++d) obs_cache[s * D + d] = ob[d];
      obs_cache_v[s] = 1;
    }
    if (sp.env_type != PHX_ENV_PLAIN && ov) {                  // a non-terminal step's reward: the cached one
      const bool cv = cached_now || rew_cache_v[s] != 0;
      if (!cached_now && cv) rc = rew_cache[s];
      if (sp.env_type == PHX_ENV_FSM) { rv = cv ? 1 : 2; rw = cv ? rc : 0.0; }          // fsm.py:378
      else if (cv) { rv = 1; rw = rc; }                                                   // stackelberg.py:190-194
    }
    for (int d = 0; d < D; ++d) obs_b[s * D + d] = ob[d];
    io.obs_valid[o] = ov; io.reward_valid[o] = rv; io.done_valid[o] = dv;
    io.terminated[o] = tm; io.truncated[o] = tr; io.reward[o] = rw;
    if (row) { for (int d = 0; d < 4; ++d) k_ob[d] = ob[d]; k_rw = rw; k_ov = ov; k_rv = rv; k_tm = tm; k_tr = tr; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the two LDS counters: a barrier that leaves the stores in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const bool all_term = *s_nterm == S;                                        // env.py:308-310
  const bool all_trunc = (t == sp.num_steps) || *s_ntrunc == S;              // env.py:312-318
  const bool terminal = all_term || all_trunc;
  if (all_flags) *all_flags = (all_term ? 1 : 0) | (all_trunc ? 2 : 0) | (next_stage << 8);       // (+ the stage the env enters)
  if (sp.env_type != PHX_ENV_PLAIN && terminal) {              // (uniform over the workgroup)
    __syncthreads();                                           // this lane's stores above are re-read below
    for (int s = tid; s < S; s += NT) {
      const int64_t o = (int64_t)b * S + s;
      if (sp.env_type == PHX_ENV_FSM) {                        // fsm.py:360-375: cached dicts of all agents
        const uint8_t v = obs_cache_v[s];
        io.obs_valid[o] = v;
        for (int d = 0; d < D; ++d) obs_b[s * D + d] = v ? obs_cache[s * D + d] : 0.f;
        if (row) { k_ov = v; for (int d = 0; d < D && d < 4; ++d) k_ob[d] = v ? obs_cache[s * D + d] : 0.f; }
      }
      const uint8_t cv = rew_cache_v[s];
      const double rc = cv ? rew_cache[s] : 0.0;
      io.reward_valid[o] = cv ? 1 : 2;                         // fsm.py:360-375 / stackelberg.py:180-187
      io.reward[o] = rc;
      if (row) { k_rv = cv ? 1 : 2; k_rw = rc; }
    }
  }
  if (row && tid < S) {                                        // the trajectory row, rollout.py:361-389
    const int64_t o = row_o + tid;
    for (int d = 0; d < D && d < 4; ++d) roll->obs[o * D + d] = k_ob[d];
    roll->reward[o] = (float)k_rw;
    if (roll->terminated) roll->terminated[o] = (uint8_t)(k_tm | (all_term ? 1 : 0));
    roll->truncated[o] = (uint8_t)(k_tr | (all_trunc ? 1 : 0));
    if (roll->obs_valid) roll->obs_valid[o] = k_ov;
    if (roll->reward_valid) roll->reward_valid[o] = k_rv;
  }
  if (tid == 0) {
    fld<int32_t>(sp, F_ENV_STEP)[b] = t;
    fld<int32_t>(sp, F_ENV_TICK)[b] = (int32_t)(tick + 1);
    if (sp.env_type == PHX_ENV_FSM) {                          // fsm.py:355
      fld<int32_t>(sp, F_ENV_PREV_STAGE)[b] = cur_stage;
      fld<int32_t>(sp, F_ENV_STAGE)[b] = next_stage;
    }
    io.all_terminated[b] = all_term; io.all_truncated[b] = all_trunc;
  }
  return row;
}

template <int NT>
__device__ __forceinline__ bool strategic_epilogue(const DevSpec& sp, const Topo& tp, const phx_step_io& io, int b, int t,
                                                   int list, int cur_stage, uint32_t tick,
                                                   const uint8_t* live, int* s_nterm, int* s_ntrunc, int next_in = -1,
                                                   const phx_rollout_io* roll = nullptr, int64_t row_o = 0, int* all_flags = nullptr) {
  return strategic_epilogue<NT>(sp, tp, io, b, t, list, cur_stage, tick, live, s_nterm, s_ntrunc,
                         [&](int a, int tt, float* ob) { return dev_encode_obs(sp, tp, b, a, tt, ob); }, next_in, roll, row_o, all_flags);
}

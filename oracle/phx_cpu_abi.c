/*
 * phx_cpu_abi.c -- the CPU restatement behind the product's OWN symbols ("exports the identical symbols").
 * TEST INFRASTRUCTURE ONLY, like the rest of oracle/ (see phx_oracle.h).
 *
 * Builds oracle/libphantom_cpu.so = phx_oracle.c + this file: every entry point of include/phantom_amd.h, same names and
 * signatures, HOST pointers instead of device pointers, `stream` ignored, `device` ignored.  The ctypes stub of
 * INTEGRATION.md section 2 (phantom_amd/_abi.py: bind_signatures) drives it unchanged; tests/test_cpu_abi.py does, and replays
 * the reference's goldens through it.  The product never loads this library: phantom_amd/_abi.py binds
 * phantom_amd/_lib/libphantom_amd.so only and fails loudly without it.
 *
 * Differences that follow from keeping the sequential restatement's state inside the handle:
 *   - the caller's state blob is not used (phx_state_nbytes returns a token size, phx_n_fields 0: no zero-copy field views);
 *     state is reached by name through phx_get_state / phx_set_state;
 *   - phx_uses_fused is 0, phx_last_kernel names the restatement, phx_sync_fields is a no-op.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "phx_oracle.h"

struct phx_env { phxo_env* o; int B, A, nnz, n_conn, n_samplers, trace_cap, n_exo, env_type; uint32_t flags; uint32_t* mt; int32_t* mt_pos;
                 /* ABI 7: per acting list the exogenous indices of its drawing agents in acting order; the FSM's transitions */
                 int n_lists, num_steps, initial_stage, has_publisher; int32_t *mt_ptr, *mt_rank, *stage_next, *stage_tab; };

int phx_abi_version(void) { return PHX_ABI_VERSION; }
const char* phx_last_error(void) { return phxo_last_error(); }
const char* phx_last_kernel(void) { return "cpu restatement (oracle/phx_oracle.c)"; }
const char* phx_autotune_note(const phx_env* e) { (void)e; return ""; }

int64_t phx_state_nbytes(const phx_spec* spec) { return (spec && spec->abi_version == PHX_ABI_VERSION) ? 256 : -1; }

static int spec_query(const phx_spec* spec, int (*fn)(const phxo_env*)) {
  phxo_env* o = phxo_create(spec);
  if (!o) return -1;
  const int v = fn(o);
  phxo_destroy(o);
  return v;
}
int phx_obs_dim(const phx_spec* spec) { return spec_query(spec, phxo_obs_dim); }
int phx_n_strategic(const phx_spec* spec) { return spec_query(spec, phxo_n_strategic); }
int phx_n_exo(const phx_spec* spec) { return spec_query(spec, phxo_n_exo); }

int phx_create(const phx_spec* spec, int device, void* state_blob, int64_t state_nbytes, phx_env** out) {
  (void)device; (void)state_blob; (void)state_nbytes;
  if (!spec || !out) return PHX_EINVAL;
  phxo_env* o = phxo_create(spec);
  if (!o) return PHX_EINVAL;
  phx_env* e = (phx_env*)calloc(1, sizeof *e);
  e->o = o; e->B = spec->batch; e->A = spec->n_agents; e->nnz = spec->row_ptr ? spec->row_ptr[spec->n_agents] : 0;
  e->n_conn = spec->n_conn; e->n_samplers = spec->n_samplers; e->trace_cap = spec->trace_cap;
  e->n_exo = phxo_n_exo(o); e->env_type = spec->env_type; e->flags = spec->flags;
  if (spec->flags & PHX_F_MT19937) {
    e->mt = (uint32_t*)calloc((size_t)spec->batch * 624, 4); e->mt_pos = (int32_t*)calloc((size_t)spec->batch, 4);
    const int A = spec->n_agents, fsm = spec->env_type == PHX_ENV_FSM;
    int32_t* rank = (int32_t*)malloc(sizeof(int32_t) * (size_t)A);     /* exogenous index: one per CustomerAgent, agent order */
    int nx = 0;
    for (int a = 0; a < A; ++a) { rank[a] = spec->kind[a] == PHX_KIND_CUSTOMER ? nx++ : -1; if (spec->kind[a] == PHX_KIND_PUBLISHER) e->has_publisher = 1; }
    e->n_lists = fsm ? spec->n_stages : 1; e->num_steps = spec->num_steps; e->initial_stage = spec->initial_stage;
    e->mt_ptr = (int32_t*)calloc((size_t)e->n_lists + 1, 4); e->mt_rank = (int32_t*)calloc((size_t)e->n_lists * (size_t)A + 1, 4);
    int n = 0;
    for (int l = 0; l < e->n_lists; ++l) {
      if (fsm) { for (int k = spec->stage_act_ptr[l]; k < spec->stage_act_ptr[l + 1]; ++k) if (rank[spec->stage_act_idx[k]] >= 0) e->mt_rank[n++] = rank[spec->stage_act_idx[k]]; }
      else for (int a = 0; a < A; ++a) if (rank[a] >= 0) e->mt_rank[n++] = rank[a];       /* env.py:320-336: every agent, insertion order */
      e->mt_ptr[l + 1] = n;
    }
    if (fsm) {
      e->stage_next = (int32_t*)malloc(sizeof(int32_t) * (size_t)e->n_lists); memcpy(e->stage_next, spec->stage_next, sizeof(int32_t) * (size_t)e->n_lists);
      if (spec->stage_tab) { const size_t m = (size_t)e->n_lists * (size_t)(spec->num_steps + 1); e->stage_tab = (int32_t*)malloc(4 * m); memcpy(e->stage_tab, spec->stage_tab, 4 * m); }
    }
    free(rank);
  }
  phxo_reset(o, NULL, NULL, NULL, NULL, NULL);          /* phx_create runs the initial reset (include/phantom_amd.h) */
  *out = e;
  return PHX_OK;
}
void phx_destroy(phx_env* e) { if (e) { phxo_destroy(e->o); free(e->mt); free(e->mt_pos); free(e->mt_ptr); free(e->mt_rank); free(e->stage_next); free(e->stage_tab); free(e); } }
int phx_n_fields(const phx_env* e) { (void)e; return 0; }
int phx_field_info(const phx_env* e, int index, phx_field* out) { (void)e; (void)index; (void)out; return PHX_EINVAL; }
int phx_uses_fused(const phx_env* e) { (void)e; return 0; }
int phx_sync_fields(phx_env* e, void* stream) { (void)e; (void)stream; return PHX_OK; }

int phx_reset(phx_env* e, const uint8_t* reset_mask, const double* sampler_values, const uint8_t* conn_on, float* obs,
              uint8_t* obs_valid, void* stream) {
  (void)stream;
  if (!e) return PHX_EINVAL;
  phxo_reset(e->o, reset_mask, sampler_values, conn_on, obs, obs_valid);
  return PHX_OK;
}
int phx_step(phx_env* e, const phx_step_io* io, void* stream) {
  (void)stream;
  if (!e || !io) return PHX_EINVAL;
  phxo_step(e->o, io);
  return PHX_OK;
}
int phx_step_begin(phx_env* e, const phx_step_io* io, void* stream) {
  (void)stream;
  if (!e || !io) return PHX_EINVAL;
  phxo_step_begin(e->o, io);
  return PHX_OK;
}
int phx_step_end(phx_env* e, const phx_step_io* io, void* stream) {
  (void)stream;
  if (!e || !io) return PHX_EINVAL;
  phxo_step_end(e->o, io);
  return PHX_OK;
}
int phx_inject(phx_env* e, const phx_msg_rec* msgs, int n) { if (!e) return PHX_EINVAL; phxo_inject(e->o, msgs, n); return PHX_OK; }
int phx_resolve(phx_env* e, int32_t* err, phx_msg_rec* msg_log, int32_t* msg_count, void* stream) {
  (void)stream;
  if (!e) return PHX_EINVAL;
  phxo_resolve(e->o, err, msg_log, msg_count);
  return PHX_OK;
}
int phx_rollout(phx_env* e, const phx_rollout_io* io, void* stream) {
  (void)stream;
  if (!e || !io || io->T <= 0) return PHX_EINVAL;
  if ((io->hints & ~(PHX_RH_ACTIONS_IN_DOMAIN | PHX_RH_EXO_IN_DOMAIN)) != 0 || io->reserved_ptr) return PHX_EINVAL;   /* (the hints change nothing here: this path never relies on them, so it reports no PHX_ERR_HINT either) */
  if (io->n_frag >= 2 || io->frags) {                    /* ABI 9, a fragment list: the same steps, the rows handed out fragment by fragment */
    if (io->n_frag < 2 || io->n_frag > PHX_MAX_FRAGMENTS || !io->frags || io->T % io->n_frag) return PHX_EINVAL;
    if (io->obs || io->action_out || io->reward || io->terminated || io->truncated || io->obs_valid || io->reward_valid) return PHX_EINVAL;
    const int Tf = io->T / io->n_frag, S = phxo_n_strategic(e->o);
    for (int f = 0; f < io->n_frag; ++f) {                 /* the validation phx_rollout of the HIP library makes (phx_api.hip): every fragment's required planes */
      const phx_rollout_frag* fr = &io->frags[f];
      if (!fr->obs || !fr->action_out || !fr->reward || !fr->truncated || (e->env_type != PHX_ENV_PLAIN && (!fr->obs_valid || !fr->reward_valid))) return PHX_EINVAL;
      if ((fr->terminated != NULL) != (io->frags[0].terminated != NULL)) return PHX_EINVAL;
    }
    for (int f = 0; f < io->n_frag; ++f) {
      const phx_rollout_frag* fr = &io->frags[f];
      phx_rollout_io sub = *io;
      sub.n_frag = 0; sub.frags = NULL; sub.T = Tf;
      sub.obs = fr->obs; sub.action_out = fr->action_out; sub.reward = fr->reward; sub.terminated = fr->terminated; sub.truncated = fr->truncated;
      sub.obs_valid = fr->obs_valid; sub.reward_valid = fr->reward_valid;
      const int64_t row = (int64_t)f * Tf * e->B;
      if (io->actions) sub.actions = io->actions + row * S;
      if (io->exo) sub.exo = io->exo + row * phxo_n_exo(e->o);
      if (io->msg_log) sub.msg_log = io->msg_log + row * e->trace_cap;
      if (io->msg_count) sub.msg_count = io->msg_count + row;
      phxo_rollout(e->o, &sub);
    }
    return PHX_OK;
  }
  phxo_rollout(e->o, io);
  return PHX_OK;
}

/* copying state access by field name: the restatement knows a field as i32, f64 or u8 */
static size_t scratch_elems(const phx_env* e) {
  size_t per = (size_t)(e->A > e->nnz ? e->A : e->nnz);
  if ((size_t)e->n_conn > per) per = (size_t)e->n_conn;
  if ((size_t)e->n_samplers > per) per = (size_t)e->n_samplers;
  return (size_t)e->B * (per + 1) * 3;
}
int64_t phx_get_state(phx_env* e, const char* field, void* buf, int64_t buf_nbytes, void* stream) {
  (void)stream;
  if (!e || !field || !buf) return PHX_EINVAL;
  const size_t n = scratch_elems(e);
  void* tmp = malloc(n * 8);
  int64_t got, esz = 4;
  got = phxo_get_i32(e->o, field, (int32_t*)tmp);
  if (got < 0) { got = phxo_get_f64(e->o, field, (double*)tmp); esz = 8; }
  if (got < 0) { got = phxo_get_u8(e->o, field, (uint8_t*)tmp); esz = 1; }
  int64_t rc = PHX_EINVAL;
  if (got >= 0 && got * esz <= buf_nbytes) { memcpy(buf, tmp, (size_t)(got * esz)); rc = got * esz; }
  free(tmp);
  return rc;
}
int64_t phx_set_state(phx_env* e, const char* field, const void* buf, int64_t buf_nbytes, void* stream) {
  (void)stream;
  if (!e || !field || !buf) return PHX_EINVAL;
  const size_t n = scratch_elems(e);
  int32_t* tmp = (int32_t*)malloc(n * 4);
  const int64_t have = phxo_get_i32(e->o, field, tmp);      /* the field's element count */
  free(tmp);
  if (have < 0 || have * 4 != buf_nbytes) return PHX_EINVAL;
  return phxo_set_i32(e->o, field, (const int32_t*)buf) < 0 ? PHX_EINVAL : buf_nbytes;
}
int phx_trace(phx_env* e, const phx_msg_rec* msg_log, const int32_t* msg_count, int b, phx_msg_rec* out, int cap, void* stream) {
  (void)stream;
  if (!e || !msg_log || !msg_count || (cap > 0 && !out) || b < 0 || b >= e->B) return PHX_EINVAL;
  int n = msg_count[b];
  if (n > e->trace_cap) n = e->trace_cap;
  const int m = n < cap ? n : cap;
  if (m > 0) memcpy(out, msg_log + (size_t)b * e->trace_cap, (size_t)m * sizeof(phx_msg_rec));
  return n;
}
int phx_pack_flags(const uint8_t* src, uint64_t* dst, int64_t n, void* stream) {
  (void)stream;
  if (!src || !dst || n < 0) return PHX_EINVAL;
  const int64_t words = (n + 63) / 64;
  memset(dst, 0, (size_t)words * 8);
  for (int64_t i = 0; i < n; ++i) if (src[i]) dst[i >> 6] |= (uint64_t)1 << (i & 63);
  return PHX_OK;
}
int phx_unpack_flags(const uint64_t* src, uint8_t* dst, int64_t n, void* stream) {
  (void)stream;
  if (!src || !dst || n < 0) return PHX_EINVAL;
  for (int64_t i = 0; i < n; ++i) dst[i] = (uint8_t)((src[i >> 6] >> (i & 63)) & 1u);
  return PHX_OK;
}

/* ---- ABI 7: per-env legacy-numpy MT19937 streams, the sequential statement (numpy random/src/mt19937/mt19937.c: mt19937_seed,
 * mt19937_gen, mt19937_next; _bounded_integers buffered_bounded_masked_uint32 for np.random.randint(5), supply_chain.py:64) ---- */
static uint32_t mt_next(uint32_t* mt, int32_t* pos) {
  if (*pos >= 624) {
    int k;
    uint32_t y;
    for (k = 0; k < 624 - 397; ++k) { y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu); mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
    for (; k < 623; ++k) { y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu); mt[k] = mt[k - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
    y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu); mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    *pos = 0;
  }
  uint32_t y = mt[(*pos)++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}
int phx_mt_seed(phx_env* e, const uint32_t* seeds, void* stream) {
  (void)stream;
  if (!e || !seeds) return PHX_EINVAL;
  if (!e->mt) return PHX_EUNSUPPORTED;
  for (int b = 0; b < e->B; ++b) {
    uint32_t* mt = e->mt + (size_t)b * 624;
    mt[0] = seeds[b];
    for (uint32_t i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i;
    e->mt_pos[b] = 624;
  }
  return PHX_OK;
}
int phx_mt_draw(phx_env* e, uint8_t* exo, int T, void* stream) {
  (void)stream;
  if (!e || !exo || T < 1) return PHX_EINVAL;
  if (!e->mt || (e->env_type != PHX_ENV_PLAIN && e->env_type != PHX_ENV_FSM) || e->n_exo < 1 || e->has_publisher) return PHX_EUNSUPPORTED;
  const int fsm = e->env_type == PHX_ENV_FSM;
  int32_t* step0 = NULL; int32_t* stage0 = NULL;
  if (fsm) {
    step0 = (int32_t*)malloc(4 * (size_t)e->B); stage0 = (int32_t*)malloc(4 * (size_t)e->B);
    phxo_get_i32(e->o, "env.step", step0); phxo_get_i32(e->o, "env.stage", stage0);
  }
  memset(exo, 0, (size_t)T * (size_t)e->B * (size_t)e->n_exo);            /* customers that do not act draw nothing */
  for (int b = 0; b < e->B; ++b) {
    int step = fsm ? step0[b] : 0, stage = fsm ? stage0[b] : 0;
    for (int t = 0; t < T; ++t) {
      for (int k = e->mt_ptr[stage]; k < e->mt_ptr[stage + 1]; ++k) {    /* the step's CustomerAgents, acting order (supply_chain.py:64) */
        uint32_t v;
        do v = mt_next(e->mt + (size_t)b * 624, e->mt_pos + b) & 7u; while (v > 4u);
        exo[((size_t)t * e->B + b) * e->n_exo + e->mt_rank[k]] = (uint8_t)v;
      }
      if (fsm) {                                                           /* fsm.py:281-307; the rollout's reset at the episode end */
        const int tn = step + 1;
        stage = e->stage_tab ? e->stage_tab[(size_t)stage * (size_t)(e->num_steps + 1) + (size_t)(tn <= e->num_steps ? tn : e->num_steps)] : e->stage_next[stage];
        step = tn;
        if (step >= e->num_steps) { step = 0; stage = e->initial_stage; }
      }
    }
  }
  free(step0); free(stage0);
  return PHX_OK;
}

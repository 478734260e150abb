// phx_api.hip -- the C ABI of include/phantom_amd.h: spec validation, table upload, state-blob
// layout and kernel dispatch.  No torch types cross this boundary; the caller owns the state
// blob and every I/O buffer, the library owns only its copy of the static spec tables.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "phx_dev.h"


size_t phx_generic_queue_bytes(int A, int S, int Q, int scan_cap, int n_adx, bool lean = false);
size_t phx_generic_lean_ws_bytes(int Q, int scan_cap);
size_t phx_generic_table_bytes(int A, int nnz);
hipError_t phx_launch_generic(const DevSpec& sp, const GenArgs& g, bool lds, hipStream_t st);
hipError_t phx_launch_sc_rollout_fsm_rules(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st);
static const int SC_RULES_MAX_S = 256;      // (whole envs per 256-lane workgroup)
const char* phx_sc_policy_unsupported(const DevSpec& sp, const phx_rollout_io& io);
hipError_t phx_launch_sc_rollout_policy(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st);
bool phx_sched_compile(const phx_spec* spec, int A, int n_lists, const int32_t* act_ptr, const int32_t* act_idx, const uint8_t* act_mask,
                       const uint8_t* obs_mask, const uint8_t* rew_mask, const int32_t* kind_rank, const int32_t* exo_rank, const int32_t* strat_rank,
                       const int32_t* reset_obs_idx, int n_reset_obs, std::vector<int32_t>* blob, std::vector<int32_t>* recs, int* L_out, int* qmax_out);
size_t phx_sched_lds_bytes(int words, int L, int qstride, int n_rules, int n_lists);
hipError_t phx_launch_reset(const DevSpec& sp, const uint8_t* mask, const double* sampler_values, const uint8_t* conn_values, float* obs, uint8_t* obs_valid, hipStream_t st);
hipError_t phx_launch_sc_step(const DevSpec& sp, const phx_step_io& io, hipStream_t st);
hipError_t phx_launch_sc_rollout(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st, const int32_t* only_if = nullptr, int32_t gen = 0);
hipError_t phx_launch_stk_step(const DevSpec& sp, const phx_step_io& io, hipStream_t st);
hipError_t phx_launch_stk_materialise(const DevSpec& sp, hipStream_t st);
hipError_t phx_launch_stk_rollout(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st);
hipError_t phx_launch_gen_last_obs(const DevSpec& sp, const float* obs, float* last_obs, hipStream_t st);
size_t phx_stk_rollout_lds(const DevSpec& sp);
#include "phx_sc_fast.h"
hipError_t phx_launch_sc_rollout_fsm(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st, const int32_t* only_if = nullptr, int32_t gen = 0);

hipError_t phx_launch_ads_step(const DevSpec& sp, const phx_step_io& io, hipStream_t st);
hipError_t phx_launch_ads_rollout(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st);

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return fail(PHX_EHIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

static const int GENERIC_LDS_LIMIT = 60 * 1024;

static bool kind_is_strategic(int k) {
  return k == PHX_KIND_SHOP || k == PHX_KIND_SELLER || k == PHX_KIND_BUYER || k == PHX_KIND_MOCK_STRAT ||
         k == PHX_KIND_ADVERTISER;
}
static int kind_obs_dim(int k) {
  switch (k) { case PHX_KIND_SHOP: return 3; case PHX_KIND_SELLER: case PHX_KIND_BUYER: return 2;
               case PHX_KIND_MOCK_STRAT: return 1; case PHX_KIND_ADVERTISER: return 3; default: return 0; }
}

// ---- derived quantities of a spec (host) ---------------------------------------------------------
struct Derived {
  int A = 0, S = 0, D = 1, n_exo = 0, nnz = 0, buyer_nnz = 0, buyer_dmax = 0, n_lists = 1, scan_cap = 0;
  int kind_count[PHX_KIND_COUNT] = {0};
  std::vector<int32_t> strat_rank, strat_idx, kind_rank, exo_rank, buyer_off;
  std::vector<int32_t> act_ptr, act_idx, stage_next, reset_obs_idx;
  std::vector<uint8_t> stage_allowed, stage_rew_all;
  std::vector<int32_t> stage_tab;
  std::vector<uint8_t> stage_has_rules;   // [n_stages] the stage's handler is a rule list (phx_spec.stage_rules)
  std::vector<uint8_t> act_mask, obs_mask, rew_mask;
  // supply-chain schedule
  bool sc_static = false, stk_static = false, ads_static = false, sc_rules_fused = false;
  int ads_pub = -1, ads_adx = -1, ads_pub_stage = 0;
  bool dynamic_graph = false;      // StochasticNetwork with some rate < 1: edges differ per env
  std::vector<int32_t> shop_agent, shop_norm, shop_cust_ptr, shop_cust_exo, shop_cust_agent;
  std::vector<uint8_t> shop_cust_act;
  std::vector<uint8_t> sc_shop_flags;   // [n_lists][nS]: 1 acts, 2 a customer acts, 4 every customer acts, 8 observes, 16 rewarded
  std::vector<float> sc_tab;
  int n_tabn = 0, n_quot = 0, rew_smax = -1;
  int max_cust = 0;
  std::vector<uint16_t> stk_nbr;
  std::vector<int32_t> stk_nbr_conn;
  std::vector<uint32_t> stk_rec;
  std::vector<uint8_t> stk_flags;
  std::vector<uint32_t> stk_rec2, stk_agent;
  bool stk_packed = false;
  // supertypes
  bool any_typed = false, device_sampling = false;
  std::vector<int32_t> type_src, shop_type_src;
  std::vector<double> shop_type_prm;
};

// phx_spec.variant_step == PHX_VS_GENERIC and variant_rollout == PHX_VR_LAUNCH_LOOP select the message-passing engine like PHX_F_FORCE_GENERIC
static uint32_t eff_flags(const phx_spec* sp) {
  return sp->flags | ((sp->variant_step == PHX_VS_GENERIC || sp->variant_step == PHX_VS_GENERIC_DYNAMIC || sp->variant_rollout == PHX_VR_LAUNCH_LOOP) ? PHX_F_FORCE_GENERIC : 0u);
}

static int derive(const phx_spec* sp, Derived& d) {
  if (!sp) return fail(PHX_EINVAL, "null spec");
  if (sp->abi_version != PHX_ABI_VERSION) return fail(PHX_EINVAL, "abi_version %d != %d", sp->abi_version, PHX_ABI_VERSION);
  if (sp->flags & PHX_F_SHUFFLE_BATCHES) {                    // resolvers.py:150-151
    if (sp->flags & PHX_F_IGNORE_CONN_ERRORS) return fail(PHX_EUNSUPPORTED, "shuffle_batches with ignore_connection_errors");
    for (int a = 0; a < sp->n_agents; ++a)
      if (sp->kind && sp->kind[a] == PHX_KIND_ADEXCHANGE) return fail(PHX_EUNSUPPORTED, "shuffle_batches with an AdExchangeAgent (its handle_batch is order sensitive in ties)");
  }
  if (sp->n_agents <= 0 || sp->n_agents > 65535) return fail(PHX_EINVAL, "n_agents out of range");
  if (sp->batch <= 0) return fail(PHX_EINVAL, "batch must be positive");
  if (!sp->kind || !sp->param_i || !sp->param_f || !sp->row_ptr || !sp->col) return fail(PHX_EINVAL, "null table");
  if (sp->queue_cap <= 0) return fail(PHX_EINVAL, "queue_cap must be positive");
  // the device RNG counter carries the global env index in 48 bits (bits 48.. hold the redraw attempt)
  if (sp->env_offset < 0 || sp->env_offset + (int64_t)sp->batch > ((int64_t)1 << 48))
    return fail(PHX_EINVAL, "env_offset + batch must stay below 2^48");
  const int A = sp->n_agents;
  if (sp->n_samplers < 0 || (sp->n_samplers > 0 && (!sp->sampler_kind || !sp->sampler_param)))
    return fail(PHX_EINVAL, "sampler tables missing");
  d.device_sampling = sp->n_samplers > 0;
  for (int j = 0; j < sp->n_samplers; ++j) {
    if (sp->sampler_kind[j] != PHX_SAMPLER_HOST && sp->sampler_kind[j] != PHX_SAMPLER_UNIFORM)
      return fail(PHX_EINVAL, "sampler %d: unknown kind %d", j, sp->sampler_kind[j]);
    if (sp->sampler_kind[j] == PHX_SAMPLER_UNIFORM && !(sp->sampler_param[4 * j + 1] >= sp->sampler_param[4 * j]))
      return fail(PHX_EINVAL, "sampler %d: high < low", j);                       // samplers.py:134
    d.device_sampling = d.device_sampling && sp->sampler_kind[j] == PHX_SAMPLER_UNIFORM;
  }
  if (sp->n_conn < 0 || (sp->n_conn > 0 && (!sp->conn_rate || !sp->col_conn))) return fail(PHX_EINVAL, "StochasticNetwork tables missing");
  bool all_on = true;
  for (int i = 0; i < sp->n_conn; ++i) {
    if (!(sp->conn_rate[i] >= 0.0)) return fail(PHX_EINVAL, "connection %d: bad rate", i);
    all_on = all_on && sp->conn_rate[i] >= 1.0;
  }
  for (int k = 0; k < (sp->n_conn > 0 ? sp->row_ptr[A] : 0); ++k)
    if (sp->col_conn[k] < 0 || sp->col_conn[k] >= sp->n_conn) return fail(PHX_EINVAL, "col_conn out of range");
  d.dynamic_graph = sp->n_conn > 0 && !all_on;
  d.type_src.assign(A, PHX_TYPE_NONE);
  for (int a = 0; a < A && sp->type_src; ++a) {
    const int src = sp->type_src[a];
    if (src == PHX_TYPE_NONE) continue;
    if (src < PHX_TYPE_NONE || src >= sp->n_samplers) return fail(PHX_EINVAL, "agent %d: type_src out of range", a);
    if (sp->kind[a] != PHX_KIND_SHOP && sp->kind[a] != PHX_KIND_ADVERTISER)
      return fail(PHX_EUNSUPPORTED, "agent %d: only ShopAgent / AdvertiserAgent consume a type field on the device", a);
    if (sp->kind[a] == PHX_KIND_SHOP && !(sp->param_f[a * PHX_NPF + 1] != 0.0)) return fail(PHX_EINVAL, "agent %d: type normaliser (pf1) is zero", a);
    d.type_src[a] = src; d.any_typed = d.any_typed || sp->kind[a] == PHX_KIND_SHOP;
  }
  d.A = A; d.nnz = sp->row_ptr[A];
  if (sp->row_ptr[0] != 0) return fail(PHX_EINVAL, "row_ptr[0] != 0");
  for (int a = 0; a < A; ++a) if (sp->row_ptr[a + 1] < sp->row_ptr[a]) return fail(PHX_EINVAL, "row_ptr not monotone");
  for (int k = 0; k < d.nnz; ++k) if (sp->col[k] < 0 || sp->col[k] >= A) return fail(PHX_EINVAL, "col out of range");
  d.strat_rank.assign(A, -1); d.kind_rank.assign(A, 0); d.exo_rank.assign(A, -1); d.buyer_off.assign(A, 0);
  for (int a = 0; a < A; ++a) {
    const int k = sp->kind[a];
    if (k <= 0 || k >= PHX_KIND_COUNT) return fail(PHX_EINVAL, "agent %d: unknown kind %d", a, k);
    d.kind_rank[a] = d.kind_count[k]++;
    if (kind_is_strategic(k)) { d.strat_rank[a] = d.S++; d.strat_idx.push_back(a);
      d.D = std::max(d.D, kind_obs_dim(k) + (k == PHX_KIND_SHOP && d.type_src[a] != PHX_TYPE_NONE ? 1 : 0)); }
    if (k == PHX_KIND_CUSTOMER) d.exo_rank[a] = d.n_exo++;
    if (k == PHX_KIND_PUBLISHER) {                              // user id + pi1 click draws per step
      if (sp->param_i[a * PHX_NPI + 1] < 0 || sp->param_i[a * PHX_NPI + 1] > 4095) return fail(PHX_EINVAL, "agent %d: click draws per step out of range", a);
      d.exo_rank[a] = d.n_exo; d.n_exo += 1 + sp->param_i[a * PHX_NPI + 1];
    }
    if (k == PHX_KIND_BUYER) { d.buyer_off[a] = d.kind_rank[a]; d.buyer_dmax = std::max(d.buyer_dmax, sp->row_ptr[a + 1] - sp->row_ptr[a]); }
    const int32_t* pi = sp->param_i + a * PHX_NPI;
    if ((k == PHX_KIND_SHOP || k == PHX_KIND_CUSTOMER) && (pi[0] < 0 || pi[0] >= A))
      return fail(PHX_EINVAL, "agent %d: target agent index out of range", a);
    if (k == PHX_KIND_SHOP && pi[1] <= 0) return fail(PHX_EINVAL, "agent %d: ShopAgent max_sales_per_step must be > 0", a);
    if (k == PHX_KIND_FORWARDER && pi[0] >= A) return fail(PHX_EINVAL, "agent %d: forward target out of range", a);
    if ((k == PHX_KIND_PUBLISHER || k == PHX_KIND_ADVERTISER || k == PHX_KIND_ADEXCHANGE) && (pi[0] < 0 || pi[0] >= A))
      return fail(PHX_EINVAL, "agent %d: exchange / publisher index out of range", a);
    if (k == PHX_KIND_ADVERTISER && (pi[1] < 0 || pi[1] > 3)) return fail(PHX_EINVAL, "agent %d: theme index must be 0..3", a);
    if (k == PHX_KIND_ADVERTISER && d.type_src[a] == PHX_TYPE_NONE) return fail(PHX_EINVAL, "agent %d: AdvertiserAgent needs a budget (type_src)", a);
    if (k == PHX_KIND_CUSTOMER && sp->kind[pi[0]] != PHX_KIND_SHOP)
      return fail(PHX_EINVAL, "agent %d: CustomerAgent.shop_id is not a ShopAgent", a);
  }
  {  // connections are undirected: every CSR entry u->v has its mirror v->u on the same base connection
    std::unordered_map<uint64_t, int32_t> ent;
    ent.reserve((size_t)d.nnz * 2);
    for (int u = 0; u < A; ++u)
      for (int k = sp->row_ptr[u]; k < sp->row_ptr[u + 1]; ++k)
        if (!ent.emplace(((uint64_t)u << 32) | (uint32_t)sp->col[k], sp->n_conn > 0 ? sp->col_conn[k] : 0).second)
          return fail(PHX_EINVAL, "agent %d: duplicate edge to %d", u, sp->col[k]);
    for (int u = 0; u < A; ++u)
      for (int k = sp->row_ptr[u]; k < sp->row_ptr[u + 1]; ++k) {
        auto it = ent.find(((uint64_t)sp->col[k] << 32) | (uint32_t)u);
        if (it == ent.end() || it->second != (sp->n_conn > 0 ? sp->col_conn[k] : 0))
          return fail(PHX_EINVAL, "edge %d->%d has no mirror edge (connections are undirected, network.py:122-123)", u, sp->col[k]);
      }
  }
  d.buyer_nnz = d.buyer_dmax * d.kind_count[PHX_KIND_BUYER];   // slot-major (ELL) price table per env
  // acting lists + masks
  if (sp->env_type == PHX_ENV_PLAIN) {
    d.n_lists = 1; d.act_ptr = {0, A};
    for (int a = 0; a < A; ++a) d.act_idx.push_back(a);
    d.obs_mask.assign(A, 1); d.rew_mask.assign(A, 1); d.stage_next = {0};
    d.reset_obs_idx = d.strat_idx;                                            // env.py:227
  } else if (sp->env_type == PHX_ENV_FSM) {
    const int ns = sp->n_stages;
    if (ns <= 0 || !sp->stage_act_ptr || !sp->stage_act_idx || !sp->stage_rewarded || !sp->stage_rewarded_all || !sp->stage_next)
      return fail(PHX_EINVAL, "FSM tables missing");
    if (sp->initial_stage < 0 || sp->initial_stage >= ns) return fail(PHX_EINVAL, "initial_stage out of range");
    d.n_lists = ns;
    d.act_ptr.assign(sp->stage_act_ptr, sp->stage_act_ptr + ns + 1);
    d.act_idx.assign(sp->stage_act_idx, sp->stage_act_idx + sp->stage_act_ptr[ns]);
    for (int v : d.act_idx) if (v < 0 || v >= A) return fail(PHX_EINVAL, "acting agent out of range");
    d.stage_next.assign(sp->stage_next, sp->stage_next + ns);
    d.stage_rew_all.assign(sp->stage_rewarded_all, sp->stage_rewarded_all + ns);
    d.stage_allowed.assign((size_t)ns * ns, 0);                               // fsm.py:304: next_stage in next_stages
    for (int st = 0; st < ns; ++st)
      for (int nx = 0; nx < ns; ++nx)
        d.stage_allowed[(size_t)st * ns + nx] = sp->stage_allowed ? (sp->stage_allowed[(size_t)st * ns + nx] != 0) : (nx == sp->stage_next[st]);
    if (sp->stage_tab) {                                                      // tabulated clock / stage handlers (ABI 6)
      d.stage_tab.assign(sp->stage_tab, sp->stage_tab + (size_t)ns * (sp->num_steps + 1));
      for (int st = 0; st < ns; ++st)
        for (int t = 0; t <= sp->num_steps; ++t) {
          const int nx = d.stage_tab[(size_t)st * (sp->num_steps + 1) + t];
          if (nx < 0 || nx >= ns || !d.stage_allowed[(size_t)st * ns + nx])
            return fail(PHX_EINVAL, "stage_tab[%d][%d] = %d is not one of the stage's next_stages (fsm.py:304-307)", st, t, nx);
        }
    }
    d.obs_mask.assign((size_t)ns * A, 0); d.rew_mask.assign((size_t)ns * A, 0);
    for (int st = 0; st < ns; ++st) {
      const int nx = sp->stage_next[st];
      if (nx < 0 || nx >= ns) return fail(PHX_EINVAL, "stage_next out of range");
      if (sp->stage_rewarded_all[st]) {                                       // fsm.py:315-317
        for (int a = 0; a < A; ++a) d.obs_mask[(size_t)st * A + a] = d.rew_mask[(size_t)st * A + a] = 1;
      } else {                                                                // fsm.py:319-320
        for (int a = 0; a < A; ++a) d.rew_mask[(size_t)st * A + a] = sp->stage_rewarded[(size_t)st * A + a];
        for (int k = sp->stage_act_ptr[nx]; k < sp->stage_act_ptr[nx + 1]; ++k)
          d.obs_mask[(size_t)st * A + sp->stage_act_idx[k]] = 1;
      }
    }
    const int i0 = sp->initial_stage;                                         // fsm.py:237-241
    for (int k = sp->stage_act_ptr[i0]; k < sp->stage_act_ptr[i0 + 1]; ++k) d.reset_obs_idx.push_back(sp->stage_act_idx[k]);
  } else if (sp->env_type == PHX_ENV_STACKELBERG) {
    if ((sp->n_leaders && !sp->leaders) || (sp->n_followers && !sp->followers)) return fail(PHX_EINVAL, "leader/follower lists missing");
    d.n_lists = 2; d.act_ptr = {0, sp->n_leaders, sp->n_leaders + sp->n_followers};
    d.obs_mask.assign((size_t)2 * A, 0); d.rew_mask.assign((size_t)2 * A, 0); d.stage_next = {0, 0};
    for (int k = 0; k < sp->n_leaders; ++k) {
      const int a = sp->leaders[k]; if (a < 0 || a >= A) return fail(PHX_EINVAL, "leader out of range");
      d.act_idx.push_back(a); d.rew_mask[a] = 1; d.obs_mask[(size_t)A + a] = 1; d.reset_obs_idx.push_back(a);
    }
    for (int k = 0; k < sp->n_followers; ++k) {
      const int a = sp->followers[k]; if (a < 0 || a >= A) return fail(PHX_EINVAL, "follower out of range");
      d.act_idx.push_back(a); d.obs_mask[a] = 1; d.rew_mask[(size_t)A + a] = 1;
    }
  } else return fail(PHX_EINVAL, "unknown env_type %d", sp->env_type);
  d.act_mask.assign((size_t)d.n_lists * A, 0);
  int longest = 0;
  for (int l = 0; l < d.n_lists; ++l) {
    longest = std::max(longest, d.act_ptr[l + 1] - d.act_ptr[l]);
    for (int k = d.act_ptr[l]; k < d.act_ptr[l + 1]; ++k) d.act_mask[(size_t)l * A + d.act_idx[k]] = 1;
  }
  d.scan_cap = std::max(sp->queue_cap, longest + PHX_MAX_INJECT);

  // ---- device-evaluated state handlers (ABI 9): what can be checked before the state layout exists ----------------------------
  if (sp->n_stage_rules < 0 || (sp->n_stage_rules > 0 && !sp->stage_rules)) return fail(PHX_EINVAL, "stage_rules: bad count / NULL table");
  if (sp->n_stage_rules > 0) {
    if (sp->env_type != PHX_ENV_FSM) return fail(PHX_EINVAL, "stage_rules need a FiniteStateMachineEnv");
    const int ns = sp->n_stages;
    d.stage_has_rules.assign((size_t)ns, 0);
    for (int r = 0; r < sp->n_stage_rules; ++r) {
      const phx_stage_rule& q = sp->stage_rules[r];
      if (q.stage < 0 || q.stage >= ns || q.next_stage < 0 || q.next_stage >= ns) return fail(PHX_EINVAL, "stage_rules[%d]: stage out of range", r);
      if (!d.stage_allowed[(size_t)q.stage * ns + q.next_stage]) return fail(PHX_EINVAL, "stage_rules[%d]: %d is not one of stage %d's next_stages (fsm.py:304-307)", r, q.next_stage, q.stage);
      if (q.cmp < PHX_CMP_LT || q.cmp > PHX_CMP_NE) return fail(PHX_EINVAL, "stage_rules[%d]: unknown comparison", r);
      if (!(q.threshold == q.threshold)) return fail(PHX_EINVAL, "stage_rules[%d]: NaN threshold", r);
      d.stage_has_rules[q.stage] = 1;
    }
    if (sp->stage_tab)
      for (int st = 0; st < ns; ++st)
        if (d.stage_has_rules[st])
          for (int t = 0; t <= sp->num_steps; ++t)
            if (sp->stage_tab[(size_t)st * (sp->num_steps + 1) + t] != sp->stage_next[st]) return fail(PHX_EINVAL, "stage %d has both rules and a tabulated handler", st);
  }
  // ---- static supply-chain schedule? (fused kernels) ------------------------------------------
  bool sc = (sp->env_type == PHX_ENV_PLAIN || sp->env_type == PHX_ENV_FSM) && d.kind_count[PHX_KIND_SHOP] > 0 && sp->n_stage_rules == 0 &&
            d.kind_count[PHX_KIND_SHOP] <= 256 &&
            !(eff_flags(sp) & (PHX_F_FORCE_GENERIC | PHX_F_SHUFFLE_BATCHES)) && sp->trace_cap == 0 &&
            (sp->round_limit < 0 || sp->round_limit >= 2) && !(sp->flags & PHX_F_IGNORE_CONN_ERRORS) && !d.dynamic_graph;
  auto edge = [&](int u, int v) { for (int k = sp->row_ptr[u]; k < sp->row_ptr[u + 1]; ++k) if (sp->col[k] == v) return true; return false; };
  for (int a = 0; a < A && sc; ++a) {
    const int k = sp->kind[a]; const int32_t* pi = sp->param_i + a * PHX_NPI;
    if (k == PHX_KIND_SHOP) sc = sp->kind[pi[0]] == PHX_KIND_FACTORY && edge(a, pi[0]) && edge(pi[0], a);
    else if (k == PHX_KIND_CUSTOMER) sc = edge(a, pi[0]) && edge(pi[0], a);
    else if (k != PHX_KIND_FACTORY) sc = false;
  }
  d.sc_static = sc;
  // the same topology with stage handlers in rule form: phx_rollout has a fused loop that evaluates the rules (phx_sc_rollout_fsm_kernel<true>);
  // phx_step / the engine's other entries stay on the message-passing engine
  {
    bool scr = sp->env_type == PHX_ENV_FSM && d.kind_count[PHX_KIND_SHOP] > 0 && sp->n_stage_rules > 0 && d.kind_count[PHX_KIND_SHOP] <= 256 &&
               !(eff_flags(sp) & (PHX_F_FORCE_GENERIC | PHX_F_SHUFFLE_BATCHES)) && sp->trace_cap == 0 && (sp->round_limit < 0 || sp->round_limit >= 2) &&
               !(sp->flags & PHX_F_IGNORE_CONN_ERRORS) && !d.dynamic_graph && !d.any_typed && sp->n_samplers == 0 && d.D == 3 && d.S == d.kind_count[PHX_KIND_SHOP];
    for (int a = 0; a < A && scr; ++a) {
      const int k = sp->kind[a]; const int32_t* pi = sp->param_i + a * PHX_NPI;
      if (k == PHX_KIND_SHOP) scr = sp->kind[pi[0]] == PHX_KIND_FACTORY && edge(a, pi[0]) && edge(pi[0], a);
      else if (k == PHX_KIND_CUSTOMER) scr = edge(a, pi[0]) && edge(pi[0], a);
      else if (k != PHX_KIND_FACTORY) scr = false;
    }
    d.sc_rules_fused = scr;
  }
  // ---- static Stackelberg-market schedule? (fused kernel) ----------------------------------------
  bool stk = sp->env_type == PHX_ENV_STACKELBERG && d.kind_count[PHX_KIND_SELLER] > 0 &&
             !(eff_flags(sp) & (PHX_F_FORCE_GENERIC | PHX_F_IGNORE_CONN_ERRORS | PHX_F_SHUFFLE_BATCHES)) && sp->trace_cap == 0 &&
             (sp->round_limit < 0 || sp->round_limit >= 1) && (!d.dynamic_graph || sp->n_samplers == 0);
  for (int a = 0; a < A && stk; ++a) {
    const int k = sp->kind[a];
    if (k != PHX_KIND_SELLER && k != PHX_KIND_BUYER) { stk = false; break; }
    for (int e = sp->row_ptr[a]; e < sp->row_ptr[a + 1] && stk; ++e) {
      const int v = sp->col[e];                    // bipartite, symmetric adjacency
      stk = sp->kind[v] == (k == PHX_KIND_SELLER ? PHX_KIND_BUYER : PHX_KIND_SELLER) && edge(v, a);
    }
  }
  d.stk_static = stk && d.kind_count[PHX_KIND_SELLER] < 65535 && d.kind_count[PHX_KIND_BUYER] < 65536 && d.buyer_dmax < 256 && d.S == A && d.D == 2;
  if (d.stk_static) {                       // slot-major neighbour table of the buyers (seller ranks)
    const int nB = d.kind_count[PHX_KIND_BUYER];
    d.stk_nbr.assign((size_t)std::max(d.buyer_dmax, 1) * std::max(nB, 1), 0xFFFF);
    d.stk_nbr_conn.assign(d.stk_nbr.size(), 0);
    for (int a = 0; a < A; ++a)
      if (sp->kind[a] == PHX_KIND_BUYER)
        for (int e = sp->row_ptr[a]; e < sp->row_ptr[a + 1]; ++e) {
          d.stk_nbr[(size_t)(e - sp->row_ptr[a]) * nB + d.kind_rank[a]] = (uint16_t)d.kind_rank[sp->col[e]];
          if (sp->n_conn > 0) d.stk_nbr_conn[(size_t)(e - sp->row_ptr[a]) * nB + d.kind_rank[a]] = sp->col_conn[e];
        }
    d.stk_rec.assign(A, 0); d.stk_flags.assign((size_t)2 * A, 0);
    for (int a = 0; a < A; ++a) {
      // buyers: deg <= buyer_dmax < 256; a seller's degree (its obs divisor) is read from row_ptr
      const int deg = sp->row_ptr[a + 1] - sp->row_ptr[a];
      d.stk_rec[a] = (uint32_t)sp->kind[a] | ((uint32_t)std::min(deg, 255) << 8) | ((uint32_t)d.kind_rank[a] << 16);
      for (int l = 0; l < 2; ++l)
        d.stk_flags[(size_t)l * A + a] = (uint8_t)((d.act_mask[(size_t)l * A + a] ? 1 : 0) | (d.obs_mask[(size_t)l * A + a] ? 2 : 0) |
                                                   (d.rew_mask[(size_t)l * A + a] ? 4 : 0));
    }
    // packed per-agent words of the batched-load kernels (phx_stk_fused.hip)
    d.stk_packed = !d.dynamic_graph && d.buyer_dmax <= 8;
    d.stk_rec2.assign(A, 0); d.stk_agent.assign((size_t)4 * A, 0);
    if (d.stk_packed)
      for (int a = 0; a < A; ++a) {
        const bool seller = sp->kind[a] == PHX_KIND_SELLER;
        const int deg = sp->row_ptr[a + 1] - sp->row_ptr[a];
        d.stk_rec2[a] = (d.stk_rec[a] & 0xffffff00u) | (seller ? 1u : 0u) | ((uint32_t)(d.stk_flags[a] & 7) << 1) |
                        ((uint32_t)(d.stk_flags[(size_t)A + a] & 7) << 4);
        if (seller) d.stk_agent[(size_t)4 * a] = (uint32_t)deg;
        else for (int j = 0; j < deg; ++j)
          d.stk_agent[(size_t)4 * a + (j >> 1)] |= (uint32_t)d.kind_rank[sp->col[sp->row_ptr[a] + j]] << ((j & 1) * 16);
      }
  }
  // ---- static digital-ads schedule? (phx_ads_fused.hip) -------------------------------------------------
  // the shipped env (digital_ads_market.py:525-596): one exchange, one publisher, N advertisers, every
  // connection present, stage P {acting = [publisher]} <-> stage A {acting = the advertisers in agent order}
  {
    const int N = d.kind_count[PHX_KIND_ADVERTISER];
    bool ads = sp->env_type == PHX_ENV_FSM && sp->n_stages == 2 && N >= 1 && N <= 1024 && d.kind_count[PHX_KIND_PUBLISHER] == 1 &&
               d.kind_count[PHX_KIND_ADEXCHANGE] == 1 && A == N + 2 && !(eff_flags(sp) & (PHX_F_FORCE_GENERIC | PHX_F_SHUFFLE_BATCHES)) && sp->trace_cap == 0 &&
               (sp->round_limit < 0 || sp->round_limit >= 3) && (!d.dynamic_graph || (sp->flags & PHX_F_IGNORE_CONN_ERRORS)) && d.D == 3 &&
               sp->stage_next[0] == 1 && sp->stage_next[1] == 0 && !sp->stage_tab;
    if (ads) {
      for (int a = 0; a < A; ++a) { if (sp->kind[a] == PHX_KIND_PUBLISHER) d.ads_pub = a; if (sp->kind[a] == PHX_KIND_ADEXCHANGE) d.ads_adx = a; }
      const int pub = d.ads_pub, adx = d.ads_adx;
      ads = sp->param_i[pub * PHX_NPI] == adx && sp->param_i[adx * PHX_NPI] == pub && sp->param_i[pub * PHX_NPI + 1] >= 1 &&
            edge(pub, adx) && sp->row_ptr[adx + 1] - sp->row_ptr[adx] == N + 1 && sp->row_ptr[pub + 1] - sp->row_ptr[pub] == N + 1;
      for (int a = 0; a < A && ads; ++a)
        if (sp->kind[a] == PHX_KIND_ADVERTISER)
          ads = sp->param_i[a * PHX_NPI] == adx && edge(a, adx) && edge(a, pub) && sp->row_ptr[a + 1] - sp->row_ptr[a] == 2;
      int ps = -1;                                            // which stage is the publisher's
      for (int l = 0; l < 2 && ads; ++l)
        if (d.act_ptr[l + 1] - d.act_ptr[l] == 1 && d.act_idx[d.act_ptr[l]] == pub) ps = l;
      ads = ads && ps >= 0;
      if (ads) {
        const int l = 1 - ps;
        ads = d.act_ptr[l + 1] - d.act_ptr[l] == N;
        for (int k = 0; k < N && ads; ++k) ads = d.act_idx[d.act_ptr[l] + k] == d.strat_idx[k];
        d.ads_pub_stage = ps;
      }
    }
    d.ads_static = ads;
  }
  if (d.kind_count[PHX_KIND_SHOP] > 0) {
    const int nS = d.kind_count[PHX_KIND_SHOP];
    d.shop_agent.assign(nS, 0); d.shop_norm.assign(nS, 1);
    d.shop_type_src.assign(nS, PHX_TYPE_NONE); d.shop_type_prm.assign((size_t)2 * nS, 0.0);
    std::vector<std::vector<int>> cust(nS);
    for (int a = 0; a < A; ++a) {
      if (sp->kind[a] == PHX_KIND_SHOP) { d.shop_agent[d.kind_rank[a]] = a; d.shop_norm[d.kind_rank[a]] = sp->param_i[a * PHX_NPI + 1];
        d.shop_type_src[d.kind_rank[a]] = d.type_src[a];
        d.shop_type_prm[2 * d.kind_rank[a]] = sp->param_f[a * PHX_NPF]; d.shop_type_prm[2 * d.kind_rank[a] + 1] = sp->param_f[a * PHX_NPF + 1]; }
      if (sp->kind[a] == PHX_KIND_CUSTOMER) cust[d.kind_rank[sp->param_i[a * PHX_NPI]]].push_back(a);
    }
    d.shop_cust_ptr.push_back(0);
    for (int s = 0; s < nS; ++s) {
      for (int a : cust[s]) { d.shop_cust_agent.push_back(a); d.shop_cust_exo.push_back(d.exo_rank[a]); }
      d.shop_cust_ptr.push_back((int)d.shop_cust_agent.size());
      d.max_cust = std::max(d.max_cust, (int)cust[s].size());
    }
    // device-RNG counter word = shop | customer group << 20 (six customers per group)
    if (d.max_cust > 6 * 4096 || nS > (1 << 20))
      return fail(PHX_EUNSUPPORTED, "at most 24576 customers per shop and 2^20 shops (device RNG counter layout)");
    // lookup tables of the rollout kernel: the reference's own formulas evaluated on the host
    //   obs   np.float32(x / n)            supply_chain.py:127-134
    //   penalty 0.1*stock (f64)            supply_chain.py:147
    bool uniform = true;
    for (int s2 = 1; s2 < nS; ++s2) uniform = uniform && d.shop_norm[s2] == d.shop_norm[0];
    const int n_quot = uniform ? std::min(4 * d.max_cust + 1, 512) : 0;   // valid x/norm entries
    d.n_tabn = n_quot; d.n_quot = n_quot;
    d.rew_smax = 0;                                             // penalty table present
    for (int x = 0; x <= 100; ++x) d.sc_tab.push_back((float)((double)x / 100.0));
    for (int x = 0; x < d.n_tabn; ++x) d.sc_tab.push_back((float)((double)x / (double)d.shop_norm[0]));
    if ((d.sc_tab.size() & 1) != 0) d.sc_tab.push_back(0.f);    // 8-byte align the f64 part
    d.n_tabn = (int)d.sc_tab.size() - 101;                       // padded length of the x/norm part
    for (int st = 0; st <= 100; ++st) {                         // 0.1 * stock as f64, two floats each
      volatile double pen = 0.1 * (double)st;
      double pv = pen; float two[2]; memcpy(two, &pv, 8);
      d.sc_tab.push_back(two[0]); d.sc_tab.push_back(two[1]);
    }
    d.shop_cust_act.assign((size_t)d.n_lists * std::max(d.n_exo, 1), 0);
    for (int l = 0; l < d.n_lists; ++l)
      for (size_t k = 0; k < d.shop_cust_agent.size(); ++k)
        d.shop_cust_act[(size_t)l * d.n_exo + k] = d.act_mask[(size_t)l * A + d.shop_cust_agent[k]];
    d.sc_shop_flags.assign((size_t)d.n_lists * nS, 0);
    for (int l = 0; l < d.n_lists; ++l)
      for (int s2 = 0; s2 < nS; ++s2) {
        const int a_shop = d.shop_agent[s2];
        bool any = false, all = true;
        for (int k = d.shop_cust_ptr[s2]; k < d.shop_cust_ptr[s2 + 1]; ++k) { const bool on = d.shop_cust_act[(size_t)l * d.n_exo + k] != 0; any |= on; all &= on; }
        d.sc_shop_flags[(size_t)l * nS + s2] = (uint8_t)((d.act_mask[(size_t)l * A + a_shop] ? 1 : 0) | (any ? 2 : 0) | ((any && all) ? 4 : 0) |
                                                         (d.obs_mask[(size_t)l * A + a_shop] ? 8 : 0) | (d.rew_mask[(size_t)l * A + a_shop] ? 16 : 0));
      }
  }
  return PHX_OK;
}

// ---- scratch of the launch-loop rollout: the outputs of one phx_step for the whole batch ----------------
struct GenScratch { int64_t obs, reward, obs_valid, reward_valid, terminated, truncated, done_valid, all_term, all_trunc, done, actions, total; };
static GenScratch gen_scratch(int64_t B, int64_t S, int64_t D) {
  GenScratch g; int64_t off = 0;
  auto take = [&](int64_t bytes) { const int64_t o = off; off += (bytes + 255) & ~(int64_t)255; return o; };
  g.obs = take(B * S * D * 4); g.reward = take(B * S * 8);
  g.obs_valid = take(B * S); g.reward_valid = take(B * S); g.terminated = take(B * S); g.truncated = take(B * S);
  g.done_valid = take(B * S); g.all_term = take(B); g.all_trunc = take(B); g.done = take(B); g.actions = take(B * S * 4);
  g.total = off;
  return g;
}
static int64_t gen_rollout_scratch_bytes(int64_t B, int64_t S, int64_t D) { return gen_scratch(B, S, D).total; }

// ---- state blob layout ------------------------------------------------------------------------------
struct FieldDef { int id; const char* name; int dtype; int kind; int64_t dim0, dim1, dim2; int64_t offset; };

static int64_t esize(int dtype) { return dtype == 1 ? 8 : (dtype == 2 ? 1 : 4); }

static bool lean_lds_spec(const phx_spec* sp, const Derived& d);
static int64_t layout(const phx_spec* sp, const Derived& d, std::vector<FieldDef>& out, int64_t* ws_stride) {
  const int64_t B = sp->batch, S = std::max(d.S, 1);
  auto kc = [&](int k) { return (int64_t)std::max(d.kind_count[k], 0); };
  std::vector<FieldDef> f = {
    {F_ENV_STEP, "env.step", 0, 0, B, 1, 1, 0}, {F_ENV_STAGE, "env.stage", 0, 0, B, 1, 1, 0},
    {F_ENV_PREV_STAGE, "env.prev_stage", 0, 0, B, 1, 1, 0}, {F_ENV_TICK, "env.tick", 0, 0, B, 1, 1, 0},
    {F_ENV_CLOCK, "env.clock", 0, 0, B, 1, 1, 0},
    {F_ENV_TERM, "env.term", 2, 0, B, S, 1, 0}, {F_ENV_TRUNC, "env.trunc", 2, 0, B, S, 1, 0},
    {F_ENV_REW_CACHE, "env.rew_cache", 1, 0, B, S, 1, 0}, {F_ENV_REW_CACHE_VALID, "env.rew_cache_valid", 2, 0, B, S, 1, 0},
    {F_ENV_OBS_CACHE, "env.obs_cache", 3, 0, B, S, d.D, 0}, {F_ENV_OBS_CACHE_VALID, "env.obs_cache_valid", 2, 0, B, S, 1, 0},
    {F_ENV_SAMPLER, "env.sampler", 1, 0, B, sp->n_samplers, 1, 0}, {F_ENV_EPISODE, "env.episode", 0, 0, B, (sp->n_samplers > 0 || sp->n_conn > 0) ? 1 : 0, 1, 0},
    {F_NET_CONN_ON, "net.conn_on", 2, 0, B, sp->n_conn, 1, 0},
    {F_ENV_ARRIVE, "env.arrive", 0, 0, B, d.sc_static ? 1 : 0, 1, 0},
    {F_ENV_MT_STATE, "env.mt_state", 0, 0, B, (sp->flags & PHX_F_MT19937) ? 624 : 0, 1, 0},
    {F_ENV_MT_POS, "env.mt_pos", 0, 0, B, (sp->flags & PHX_F_MT19937) ? 1 : 0, 1, 0},
    {F_SHOP_STOCK, "shop.stock", 0, PHX_KIND_SHOP, B, kc(PHX_KIND_SHOP), 1, 0},
    {F_SHOP_SALES, "shop.sales", 0, PHX_KIND_SHOP, B, kc(PHX_KIND_SHOP), 1, 0},
    {F_SHOP_MISSED, "shop.missed_sales", 0, PHX_KIND_SHOP, B, kc(PHX_KIND_SHOP), 1, 0},
    {F_SHOP_DELIVERED, "shop.delivered_stock", 0, PHX_KIND_SHOP, B, kc(PHX_KIND_SHOP), 1, 0},
    {F_SELLER_PRICE, "seller.price", 1, PHX_KIND_SELLER, B, kc(PHX_KIND_SELLER), 1, 0},
    {F_SELLER_REVENUE, "seller.revenue", 1, PHX_KIND_SELLER, B, kc(PHX_KIND_SELLER), 1, 0},
    {F_SELLER_TX, "seller.tx", 0, PHX_KIND_SELLER, B, kc(PHX_KIND_SELLER), 1, 0},
    {F_SELLER_POSTED, "seller.posted", 1, PHX_KIND_SELLER, B, kc(PHX_KIND_SELLER), 1, 0},
    {F_BUYER_PRICES, "buyer.prices", 1, PHX_KIND_BUYER, B, d.buyer_dmax, kc(PHX_KIND_BUYER), 0},
    {F_BUYER_PAID, "buyer.paid", 1, PHX_KIND_BUYER, B, kc(PHX_KIND_BUYER), 1, 0},
    {F_BUYER_BOUGHT, "buyer.bought", 0, PHX_KIND_BUYER, B, kc(PHX_KIND_BUYER), 1, 0},
    {F_CASHBOX_TOTAL, "cashbox.total_cash", 1, PHX_KIND_CASHBOX, B, kc(PHX_KIND_CASHBOX), 1, 0},
    {F_REQRESP_REQ, "reqresp.req_time", 0, PHX_KIND_REQRESP, B, kc(PHX_KIND_REQRESP), 1, 0},
    {F_REQRESP_RES, "reqresp.res_time", 0, PHX_KIND_REQRESP, B, kc(PHX_KIND_REQRESP), 1, 0},
    {F_MOCK_ENC, "mock.encode_obs_count", 0, PHX_KIND_MOCK_STRAT, B, kc(PHX_KIND_MOCK_STRAT), 1, 0},
    {F_MOCK_DEC, "mock.decode_action_count", 0, PHX_KIND_MOCK_STRAT, B, kc(PHX_KIND_MOCK_STRAT), 1, 0},
    {F_MOCK_REW, "mock.compute_reward_count", 0, PHX_KIND_MOCK_STRAT, B, kc(PHX_KIND_MOCK_STRAT), 1, 0},
    {F_ADV_LEFT, "adv.left", 1, PHX_KIND_ADVERTISER, B, kc(PHX_KIND_ADVERTISER), 1, 0},
    {F_ADV_BID, "adv.bid", 1, PHX_KIND_ADVERTISER, B, kc(PHX_KIND_ADVERTISER), 1, 0},
    {F_ADV_LEFT_TAG, "adv.left_tag", 0, PHX_KIND_ADVERTISER, B, kc(PHX_KIND_ADVERTISER), 1, 0},
    {F_ADV_BID_TAG, "adv.bid_tag", 0, PHX_KIND_ADVERTISER, B, kc(PHX_KIND_ADVERTISER), 1, 0},
    {F_ADV_CLICKS, "adv.step_clicks", 0, PHX_KIND_ADVERTISER, B, kc(PHX_KIND_ADVERTISER), 1, 0},
    {F_ADV_WINS, "adv.step_wins", 0, PHX_KIND_ADVERTISER, B, kc(PHX_KIND_ADVERTISER), 1, 0},
    {F_ADV_USER, "adv.user", 0, PHX_KIND_ADVERTISER, B, kc(PHX_KIND_ADVERTISER), 1, 0},
    {F_ADV_TOT_CLICKS, "adv.total_clicks", 0, PHX_KIND_ADVERTISER, B, kc(PHX_KIND_ADVERTISER), 3, 0},
    {F_ADV_TOT_REQUESTS, "adv.total_requests", 0, PHX_KIND_ADVERTISER, B, kc(PHX_KIND_ADVERTISER), 3, 0},
    {F_ADV_TOT_WINS, "adv.total_wins", 0, PHX_KIND_ADVERTISER, B, kc(PHX_KIND_ADVERTISER), 3, 0},
    {F_PUB_ADS_SEEN, "pub.ads_seen", 0, PHX_KIND_PUBLISHER, B, kc(PHX_KIND_PUBLISHER), 1, 0},
  };
  int64_t off = 0;
  out.clear();
  for (auto& x : f) {
    const int64_t n = x.dim0 * x.dim1 * x.dim2;
    if (n == 0) continue;
    x.offset = off;
    off += (n * esize(x.dtype) + 255) & ~(int64_t)255;
    out.push_back(x);
  }
  // workspace of the generic engine when its queues do not fit LDS
  int n_adx_spec = 0;
  for (int a = 0; a < d.A; ++a) n_adx_spec += sp->kind[a] == PHX_KIND_ADEXCHANGE;
  const size_t qb = phx_generic_queue_bytes(d.A, d.S, sp->queue_cap, d.scan_cap, n_adx_spec);
  *ws_stride = 0;
  if (lean_lds_spec(sp, d)) {
    // scheduled two-wave supply chains: the sort / scan scratch of a DYNAMIC step lives here instead of LDS (phx_generic.hip, LEAN)
    *ws_stride = ((int64_t)phx_generic_lean_ws_bytes(sp->queue_cap, d.scan_cap) + 255) & ~(int64_t)255;
    FieldDef w = {F_WORKSPACE, "workspace", 2, 0, B, *ws_stride, 1, off};
    off += B * *ws_stride;
    out.push_back(w);
  } else if (qb > (size_t)GENERIC_LDS_LIMIT) {
    *ws_stride = ((int64_t)qb + 255) & ~(int64_t)255;
    FieldDef w = {F_WORKSPACE, "workspace", 2, 0, B, *ws_stride, 1, off};
    off += B * *ws_stride;
    out.push_back(w);
  }
  // step-shaped scratch of the launch-loop rollout (envs without a fused rollout kernel)
  // (a market too large for the LDS-resident rollout kernel falls back to the generic engine's loop as well)
  const int64_t stk_nS = d.kind_count[PHX_KIND_SELLER], stk_nB = d.kind_count[PHX_KIND_BUYER];
  const bool stk_big = d.stk_static && (d.A > 3 * 1024 ||
                                        8 * (3 * stk_nS + d.A + stk_nB) + 9 * stk_nS + d.A + stk_nB + sp->n_conn + 32 > 60 * 1024);
  if ((!d.sc_static && !d.stk_static) || stk_big) {       // (the fused ads kernels keep it: injected sends fall back to the loop)
    const int64_t n = gen_rollout_scratch_bytes(B, S, d.D);
    FieldDef r = {F_ROLLOUT_SCRATCH, "rollout.scratch", 2, 0, 1, n, 1, off};
    off += n;
    out.push_back(r);
  }
  return std::max<int64_t>(off, 256);
}

// ---- static round schedule of the generic engine --------------------------------------------
// For specs whose message flow cannot drop or add a message -- static Network, no ignore_connection_errors, no shuffle, only
// the supply-chain kinds (every send and every reply is unconditional: supply_chain.py:40-45,55-67,98-122,136-142) and the mock
// kinds that never send -- the rounds of a step in which every agent has a context and every acting ShopAgent an action are the
// same for every env and every step: which message sits where in which inbox (resolvers.py:128-158 -- receivers in first-arrival
// order, batches in send order, replies in handling order).  Simulated here per acting list; the kernel checks the premise per env
// and step and otherwise runs its atomics / scan / rank sort as before.
struct StaticSched { std::vector<int32_t> blob, off; };
static void build_static_schedule(const phx_spec* sp, const Derived& d, StaticSched& out) {
  const int A = d.A;
  out.blob.clear(); out.off.assign(d.n_lists, -1);
  if ((sp->flags & (PHX_F_SHUFFLE_BATCHES | PHX_F_IGNORE_CONN_ERRORS)) || d.dynamic_graph) return;
  for (int a = 0; a < A; ++a) {
    const int k = sp->kind[a];
    if (k != PHX_KIND_FACTORY && k != PHX_KIND_SHOP && k != PHX_KIND_CUSTOMER && k != PHX_KIND_MOCK_STRAT && k != PHX_KIND_MOCK_AGENT) return;
  }
  auto edge = [&](int u, int v) { for (int e = sp->row_ptr[u]; e < sp->row_ptr[u + 1]; ++e) if (sp->col[e] == v) return true; return false; };
  auto payload_ok = [&](int src, int dst, int type) {
    if (sp->flags & PHX_F_NO_PAYLOAD_CHECKS) return true;
    int sk = 0, rk = 0;
    switch (type) {
      case PHX_MSG_ORDER_REQUEST: sk = PHX_KIND_CUSTOMER; rk = PHX_KIND_SHOP; break;
      case PHX_MSG_ORDER_RESPONSE: sk = PHX_KIND_SHOP; rk = PHX_KIND_CUSTOMER; break;
      case PHX_MSG_STOCK_REQUEST: sk = PHX_KIND_SHOP; rk = PHX_KIND_FACTORY; break;
      case PHX_MSG_STOCK_RESPONSE: sk = PHX_KIND_FACTORY; rk = PHX_KIND_SHOP; break;
      default: return false;
    }
    return sp->kind[src] == sk && sp->kind[dst] == rk;
  };
  struct M { int src, dst, type; };
  for (int l = 0; l < d.n_lists; ++l) {
    std::vector<M> q;
    std::vector<int32_t> act_off;                                // queue offset of each acting item's message, -1: sends nothing
    bool ok = true;
    for (int k = d.act_ptr[l]; k < d.act_ptr[l + 1] && ok; ++k) {
      const int a = d.act_idx[k], kind = sp->kind[a], dst = sp->param_i[a * PHX_NPI];
      if (kind == PHX_KIND_SHOP || kind == PHX_KIND_CUSTOMER) {
        const int type = kind == PHX_KIND_SHOP ? PHX_MSG_STOCK_REQUEST : PHX_MSG_ORDER_REQUEST;
        ok = dst >= 0 && dst < A && edge(a, dst) && payload_ok(a, dst, type);
        act_off.push_back((int32_t)q.size());
        q.push_back({a, dst, type});
      } else act_off.push_back(-1);
    }
    if (!ok || (int)q.size() > sp->queue_cap) continue;
    std::vector<int32_t> rec(1 + PHX_SCHED_MAX_ROUNDS, 0);
    rec.insert(rec.end(), act_off.begin(), act_off.end());
    int R = 0;
    while (!q.empty() && ok) {
      if (R == PHX_SCHED_MAX_ROUNDS || (sp->round_limit >= 0 && R >= sp->round_limit)) { ok = false; break; }
      const int n = (int)q.size();
      std::vector<int32_t> cnt(A, 0), first(A, 0x7fffffff), goff(A, 0), order(n, 0), fill(A, 0);
      for (int i = 0; i < n; ++i) { cnt[q[i].dst]++; first[q[i].dst] = std::min(first[q[i].dst], i); }
      int run = 0;
      for (int i = 0; i < n; ++i) if (first[q[i].dst] == i) { goff[q[i].dst] = run; run += cnt[q[i].dst]; }   // dict order of receivers
      for (int i = 0; i < n; ++i) order[goff[q[i].dst] + fill[q[i].dst]++] = i;                               // batches in send order
      std::vector<M> nq;
      std::vector<int32_t> next_off(n, -1);                       // where the reply to inbox position P goes in the next queue
      for (int P = 0; P < n && ok; ++P) {                       // replies in handling order
        const M m = q[order[P]];
        const int rk = sp->kind[m.dst];
        if (rk == PHX_KIND_FACTORY && m.type == PHX_MSG_STOCK_REQUEST) {
          ok = payload_ok(m.dst, m.src, PHX_MSG_STOCK_RESPONSE); next_off[P] = (int32_t)nq.size(); nq.push_back({m.dst, m.src, PHX_MSG_STOCK_RESPONSE});
        } else if (rk == PHX_KIND_SHOP && m.type == PHX_MSG_ORDER_REQUEST) {
          ok = payload_ok(m.dst, m.src, PHX_MSG_ORDER_RESPONSE); next_off[P] = (int32_t)nq.size(); nq.push_back({m.dst, m.src, PHX_MSG_ORDER_RESPONSE});
        } else if ((rk == PHX_KIND_SHOP && m.type == PHX_MSG_STOCK_RESPONSE) || (rk == PHX_KIND_CUSTOMER && m.type == PHX_MSG_ORDER_RESPONSE)) {
        } else ok = false;                                        // no handler: the dynamic path reports it
      }
      if ((int)nq.size() > sp->queue_cap) ok = false;
      rec[1 + R] = n;
      rec.insert(rec.end(), cnt.begin(), cnt.end()); rec.insert(rec.end(), goff.begin(), goff.end()); rec.insert(rec.end(), order.begin(), order.end());
      rec.insert(rec.end(), next_off.begin(), next_off.end());
      ++R; q.swap(nq);
    }
    if (!ok) continue;
    rec[0] = R;
    out.off[l] = (int32_t)out.blob.size();
    out.blob.insert(out.blob.end(), rec.begin(), rec.end());
  }
}

// LEAN layout of the generic engine (phx_generic.hip): every acting list of a two-wave supply chain has a static schedule ->
// the sort / scan scratch only a dynamic step needs (order, slot, scanbuf) moves from LDS to a per-env workspace in the blob
static bool lean_lds_spec(const phx_spec* sp, const Derived& d) {
  if (d.sc_static) return false;                 // the fused kernels serve this spec: the generic engine is its rare fallback (phx_inject / phx_resolve,
                                                 // the split FSM step) and gets no schedule (DevSpec::sched) -- it keeps the plain layout
  if (d.A <= 64 || d.A > 256) return false;
  for (int a = 0; a < d.A; ++a) { const int k = sp->kind[a]; if (k != PHX_KIND_FACTORY && k != PHX_KIND_SHOP && k != PHX_KIND_CUSTOMER) return false; }
  if (phx_generic_queue_bytes(d.A, d.S, sp->queue_cap, d.scan_cap, 0, true) > (size_t)GENERIC_LDS_LIMIT) return false;
  StaticSched ss;
  build_static_schedule(sp, d, ss);
  if (ss.blob.empty()) return false;
  for (int l = 0; l < d.n_lists; ++l) if (ss.off[l] < 0) return false;
  return true;
}

// ---- handle ----------------------------------------------------------------------------------------------
struct phx_env {
  DevSpec d;
  Derived der;
  std::vector<FieldDef> fields;
  std::vector<void*> dev_allocs;
  int device = 0;
  bool use_fused = false, use_stk = false, use_ads = false, lds_ok = true;
  bool sc_rules_fused = false;      // rule-form FSM supply chain: phx_rollout takes phx_sc_rollout_fsm_kernel<true>
  bool prices_compressed = false;   // buyer.prices is represented by seller.posted (fused Stackelberg kernel)
  std::atomic<int32_t> fsm_gen{0};  // launch generation of the time-parallel FSM rollout (DevSpec::fsm_gen_host)
  int32_t sw_guard_gen = 0;         // number of the last replayed-actions call on the store-wave kernel (DevSpec::sc_sw_guard)
  DevMsg* inject_dev = nullptr;
  DevMsg inject_host[PHX_MAX_INJECT];
  int n_inject = 0;
  // PHX_VR_AUTO by measurement (FSM supply chains): the kernel chosen for a (T, n_frag) shape on THIS box, 0 = the store-wave
  // instantiation, 1 = the lane-per-pair chain; the state blob's copy the probe launches start from and restore
  void* state_blob = nullptr; int64_t state_nbytes = 0;
  void* probe_snapshot = nullptr;
  std::unordered_map<uint64_t, int> fsm_auto;
  std::string fsm_auto_note;        // "T=400: store-wave 861.2 us, loop 772.4 us -> loop" (phx_last_error's sibling: phx_autotune_note)
};

template <typename T>
static int upload(phx_env* e, const T* host, size_t n, const T** out) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  HIPCHK(hipMalloc(&p, bytes));
  e->dev_allocs.push_back(p);
  if (n) HIPCHK(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T*)p;
  return PHX_OK;
}

static thread_local char g_kernels[384] = "";
#ifdef PHX_TIMING
// development builds: phase timers of the generic engine (10 ns ticks of thread 0, first 64 workgroups), dumped every PHX_TIMING_DUMP env-steps
static unsigned long long* phx_gen_timing(int steps) {
  static unsigned long long* tb = nullptr; static long done = 0, next = 0;
  if (!tb) { (void)hipMalloc((void**)&tb, 16 * 8); (void)hipMemset(tb, 0, 16 * 8); next = getenv("PHX_TIMING_DUMP") ? atol(getenv("PHX_TIMING_DUMP")) : 0; }
  if (next > 0 && done >= next) {
    (void)hipDeviceSynchronize(); unsigned long long h[16]; (void)hipMemcpy(h, tb, sizeof h, hipMemcpyDeviceToHost); (void)hipMemset(tb, 0, 16 * 8);
    const double n = (double)done * 64; double tot = 0; fprintf(stderr, "PHX_GTIMING ns/step:");
    for (int q = 0; q < 16; ++q) { fprintf(stderr, " %d:%.0f", q, h[q] * 10.0 / n); tot += h[q] * 10.0 / n; }
    fprintf(stderr, "  total %.0f\n", tot); done = 0;
  }
  done += steps;
  return tb;
}
#endif
static void note_reset() { g_kernels[0] = 0; }
void phx_note_kernel(const char* name) {
  const size_t n = strlen(g_kernels), m = strlen(name);
  if (strstr(g_kernels, name) || n + m + 2 >= sizeof g_kernels) return;
  if (n) { g_kernels[n] = '+'; memcpy(g_kernels + n + 1, name, m + 1); } else memcpy(g_kernels, name, m + 1);
}

extern "C" {

// the development knobs: every PHX_* environment toggle of the library, read here and nowhere else (first use: phx_create)
extern "C++" const DevKnobs& phx_knobs() {
  static const DevKnobs knobs = [] {
    DevKnobs k;
    auto rd = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
    k.fsm_fast = rd("PHX_FSM_FAST", -1);             // -1: unset; 0 off; 2 forces it at any batch size
    k.fsm_lean = rd("PHX_FSM_LEAN", 1);
    k.fsm_wide = rd("PHX_FSM_WIDE", 1);
    k.fsm_batch = rd("PHX_FSM_BATCH", 1);
    k.generic_nt = rd("PHX_GENERIC_NT", 0);
    k.generic_remap = rd("PHX_GENERIC_REMAP", 1);
    k.generic_tablds = rd("PHX_GENERIC_TABLDS", 1);
    k.generic_sched = rd("PHX_GENERIC_SCHED", 1);
    k.autotune = rd("PHX_AUTOTUNE", 1);
    k.rollout_epb = rd("PHX_ROLLOUT_EPB", 0);
    k.rollout_fast = rd("PHX_ROLLOUT_FAST", -1);     // -1: unset; 0 switches the kernel off
    k.rollout_first = rd("PHX_ROLLOUT_FIRST", 0);
    k.rollout_g = rd("PHX_ROLLOUT_G", 0);
    k.rollout_ldskb = rd("PHX_ROLLOUT_LDSKB", 0);
    k.rollout_nt = rd("PHX_ROLLOUT_NT", 0);
    k.rollout_remap = rd("PHX_ROLLOUT_REMAP", -1);
    k.rollout_sparse_flags = rd("PHX_ROLLOUT_SPARSE_FLAGS", 1);
    k.step_nt = rd("PHX_STEP_NT", 0);
    k.stk_rollout_nt = rd("PHX_STK_ROLLOUT_NT", 0);
    k.stk_step_fast = rd("PHX_STK_STEP_FAST", 1);
    k.stk_step_nt = rd("PHX_STK_STEP_NT", 0);
    k.sw_generic = rd("PHX_SW_GENERIC", 0);
    k.sw_persist = rd("PHX_SW_PERSIST", 1);
    k.sw_store_waves = rd("PHX_SW_STORE_WAVES", 0);
    k.sw_tc = rd("PHX_SW_TC", 0);
    k.sw_work_waves = rd("PHX_SW_WORK_WAVES", 0);
    return k;
  }();
  return knobs;
}

int phx_abi_version(void) { return PHX_ABI_VERSION; }
const char* phx_last_error(void) { return g_err; }

const char* phx_last_kernel(void) { return g_kernels; }

int64_t phx_state_nbytes(const phx_spec* spec) {
  Derived d; if (derive(spec, d) != PHX_OK) return -1;
  std::vector<FieldDef> f; int64_t ws;
  return layout(spec, d, f, &ws);
}
int phx_obs_dim(const phx_spec* spec) { Derived d; return derive(spec, d) == PHX_OK ? d.D : -1; }
int phx_n_strategic(const phx_spec* spec) { Derived d; return derive(spec, d) == PHX_OK ? d.S : -1; }
int phx_n_exo(const phx_spec* spec) { Derived d; return derive(spec, d) == PHX_OK ? d.n_exo : -1; }

int phx_create(const phx_spec* spec, int device, void* state_blob, int64_t state_nbytes, phx_env** out) {
  if (!out) return fail(PHX_EINVAL, "null out");
  *out = nullptr;
  (void)phx_knobs();                                    // the development knobs are read here, once per process
  phx_env* e = new phx_env();
  int rc = derive(spec, e->der);
  if (rc != PHX_OK) { delete e; return rc; }
  const Derived& der = e->der;
  int64_t ws_stride = 0;
  const int64_t need = layout(spec, der, e->fields, &ws_stride);
  if (!state_blob || state_nbytes < need) { delete e; return fail(PHX_EINVAL, "state blob too small: %lld < %lld", (long long)state_nbytes, (long long)need); }
  if (((uintptr_t)state_blob & 255) != 0) { delete e; return fail(PHX_EINVAL, "state blob must be 256-byte aligned"); }
  e->device = device;
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) { delete e; return fail(PHX_EHIP, "hipSetDevice(%d): %s", device, hipGetErrorString(he)); }
  DevSpec& d = e->d;
  memset(&d, 0, sizeof d);
  d.A = der.A; d.S = der.S; d.B = spec->batch; d.D = der.D; d.n_exo = der.n_exo; d.nnz = der.nnz;
  d.num_steps = spec->num_steps; d.round_limit = spec->round_limit; d.env_type = spec->env_type;
  d.flags = spec->flags; d.queue_cap = spec->queue_cap; d.trace_cap = spec->trace_cap; d.scan_cap = der.scan_cap;
  d.n_lists = der.n_lists; d.initial_stage = spec->initial_stage; d.buyer_nnz = der.buyer_nnz; d.buyer_stride = der.kind_count[PHX_KIND_BUYER];
  d.seed = spec->seed; d.env_offset = spec->env_offset;
  d.variant_rollout = spec->variant_rollout; d.variant_block = spec->variant_block; d.variant_step = spec->variant_step;
  memcpy(d.kind_count, der.kind_count, sizeof d.kind_count);
  const int A = der.A;
#define UP(dst, ptr, n) do { rc = upload(e, ptr, (size_t)(n), &d.dst); if (rc != PHX_OK) { phx_destroy(e); return rc; } } while (0)
  UP(kind, spec->kind, A); UP(param_i, spec->param_i, A * PHX_NPI); UP(param_f, spec->param_f, A * PHX_NPF);
  UP(row_ptr, spec->row_ptr, A + 1); UP(col, spec->col, der.nnz);
  UP(strat_rank, der.strat_rank.data(), A); UP(strat_idx, der.strat_idx.data(), der.strat_idx.size());
  UP(kind_rank, der.kind_rank.data(), A); UP(exo_rank, der.exo_rank.data(), A); UP(buyer_off, der.buyer_off.data(), A);
  UP(act_ptr, der.act_ptr.data(), der.act_ptr.size()); UP(act_idx, der.act_idx.data(), der.act_idx.size());
  UP(act_mask, der.act_mask.data(), der.act_mask.size());
  UP(obs_mask, der.obs_mask.data(), der.obs_mask.size()); UP(rew_mask, der.rew_mask.data(), der.rew_mask.size());
  UP(stage_next, der.stage_next.data(), der.stage_next.size());
  if (spec->flags & PHX_F_MT19937) {                            // the draw sequence of every acting list (phx_mt_draw)
    std::vector<int32_t> mp = {0}, mr;
    for (int l = 0; l < der.n_lists; ++l) {
      for (int k = der.act_ptr[l]; k < der.act_ptr[l + 1]; ++k) { const int r = der.exo_rank[der.act_idx[k]]; if (r >= 0) mr.push_back(r); }
      mp.push_back((int32_t)mr.size());
    }
    UP(mt_ptr, mp.data(), mp.size()); UP(mt_rank, mr.data(), mr.size());
  }
  UP(stage_allowed, der.stage_allowed.data(), der.stage_allowed.size());
  d.stage_tab = nullptr;
  if (!der.stage_tab.empty()) UP(stage_tab, der.stage_tab.data(), der.stage_tab.size());
  d.n_rules = 0; d.rules = nullptr;
  if (spec->n_stage_rules > 0) {                                // the rules' fields by name: per-agent i32 / f64 state of one kind
    std::vector<DevRule> rl;
    for (int r = 0; r < spec->n_stage_rules; ++r) {
      const phx_stage_rule& q = spec->stage_rules[r];
      char nm[33]; memcpy(nm, q.field, 32); nm[32] = 0;
      const FieldDef* f = nullptr;
      for (const FieldDef& c : e->fields) if (!strcmp(c.name, nm)) f = &c;
      if (!f || f->kind <= 0 || (f->dtype != 0 && f->dtype != 1) || f->dim2 != 1 || f->dim1 < 1) {
        phx_destroy(e); return fail(PHX_EINVAL, "stage_rules[%d]: '%s' is not a per-agent i32 / f64 state field", r, nm);
      }
      if (q.agent < -1 || q.agent >= f->dim1) { phx_destroy(e); return fail(PHX_EINVAL, "stage_rules[%d]: agent column %d outside the %lld agents of '%s'", r, q.agent, (long long)f->dim1, nm); }
      DevRule dr; dr.stage = q.stage; dr.field_id = f->id; dr.col = q.agent; dr.ncols = (int32_t)f->dim1; dr.cmp = q.cmp; dr.next_stage = q.next_stage;
      dr.is_f64 = f->dtype == 1; dr.pad = 0; dr.threshold = q.threshold;
      rl.push_back(dr);
    }
    UP(rules, rl.data(), rl.size());
    d.n_rules = (int32_t)rl.size();
  }
  d.sched = nullptr; d.sched_off = nullptr;
  bool sched_every_list = false;
  if (!der.sc_static && !der.stk_static && !der.ads_static) {      // specs the generic engine serves
    StaticSched ss;
    build_static_schedule(spec, der, ss);
    if (!ss.blob.empty()) {
      UP(sched, ss.blob.data(), ss.blob.size()); UP(sched_off, ss.off.data(), ss.off.size());
      sched_every_list = true;
      for (int l = 0; l < der.n_lists; ++l) sched_every_list = sched_every_list && ss.off[l] >= 0;
    }
  }
  // ... and, where the flow is the supply chain's, the same schedule COMPILED for the several-envs-per-wave kernel (phx_generic_sched.hip)
  d.gs_ok = 0;
  if (d.sched && der.D == 3 && !der.any_typed) {
    const bool all = sched_every_list;
    std::vector<int32_t> gblob, grecs; int gL = 0, gq = 0;
    if (all && phx_sched_compile(spec, A, der.n_lists, der.act_ptr.data(), der.act_idx.data(), der.act_mask.data(), der.obs_mask.data(), der.rew_mask.data(),
                                 der.kind_rank.data(), der.exo_rank.data(), der.strat_rank.data(), der.reset_obs_idx.data(), (int)der.reset_obs_idx.size(),
                                 &gblob, &grecs, &gL, &gq)) {
      int qstride = gq + 1;
      while ((qstride & 31) != 9) ++qstride;                    // (the env instances of a wave start their queues 9 banks apart)
      if (der.n_lists < 65536 && phx_sched_lds_bytes((int)gblob.size(), gL, qstride, d.n_rules, der.n_lists) <= 48 * 1024 &&
          phx_generic_queue_bytes(der.A, der.S, spec->queue_cap, der.scan_cap, 0, false) <= 48 * 1024) {      // (the tail workgroups run the dynamic engine in LDS)
        UP(gs_blob, gblob.data(), gblob.size());
        if (grecs.empty()) grecs.assign(2, 0);
        UP(gs_rec, grecs.data(), grecs.size());
        std::vector<int32_t> zf((size_t)d.B, 0); const int32_t* fl = nullptr;
        rc = upload(e, zf.data(), zf.size(), &fl); if (rc != PHX_OK) { phx_destroy(e); return rc; }
        d.gs_dyn_flag = (int32_t*)fl;
        d.gs_ok = 1; d.gs_L = gL; d.gs_qstride = qstride; d.gs_words = (int32_t)gblob.size();
      }
    }
  }
  UP(stage_rew_all, der.stage_rew_all.data(), der.stage_rew_all.size());
  UP(reset_obs_idx, der.reset_obs_idx.data(), der.reset_obs_idx.size());
  d.n_reset_obs = (int)der.reset_obs_idx.size();
  UP(shop_agent, der.shop_agent.data(), der.shop_agent.size());
  UP(shop_norm, der.shop_norm.data(), der.shop_norm.size());
  UP(shop_cust_ptr, der.shop_cust_ptr.data(), der.shop_cust_ptr.size());
  UP(shop_cust_exo, der.shop_cust_exo.data(), der.shop_cust_exo.size());
  UP(shop_cust_agent, der.shop_cust_agent.data(), der.shop_cust_agent.size());
  UP(shop_cust_act, der.shop_cust_act.data(), der.shop_cust_act.size());
  UP(sc_shop_flags, der.sc_shop_flags.data(), der.sc_shop_flags.size());
  UP(sc_tab, der.sc_tab.data(), der.sc_tab.size());
  UP(conn_rate, spec->conn_rate, spec->n_conn); UP(col_conn, spec->col_conn, spec->n_conn > 0 ? der.nnz : 0);
  d.n_conn = spec->n_conn;
  UP(stk_nbr, der.stk_nbr.data(), der.stk_nbr.size()); UP(stk_nbr_conn, der.stk_nbr_conn.data(), der.stk_nbr_conn.size());
  UP(stk_rec, der.stk_rec.data(), der.stk_rec.size()); UP(stk_flags, der.stk_flags.data(), der.stk_flags.size());
  UP(stk_rec2, der.stk_rec2.data(), der.stk_rec2.size()); UP(stk_agent, der.stk_agent.data(), der.stk_agent.size());
  d.stk_packed = der.stk_packed ? 1 : 0;
  UP(sampler_kind, spec->sampler_kind, spec->n_samplers); UP(sampler_param, spec->sampler_param, 4 * spec->n_samplers);
  UP(type_src, der.type_src.data(), A);
  {                                                            // phx_generic_step_kernel's LDS table layout, packed
    std::vector<char> blob(phx_generic_table_bytes(A, der.nnz), 0);
    char* tb = blob.data();
    auto put = [&](const void* src, size_t bytes) { if (bytes) memcpy(tb, src, bytes); tb += (bytes + 15) & ~(size_t)15; };
    put(spec->row_ptr, (size_t)(A + 1) * 4); put(spec->col, (size_t)der.nnz * 4); put(spec->param_i, (size_t)A * PHX_NPI * 4);
    put(der.strat_rank.data(), (size_t)A * 4); put(der.kind_rank.data(), (size_t)A * 4); put(der.exo_rank.data(), (size_t)A * 4);
    put(spec->kind, (size_t)A);
    d.tab_bytes = (int32_t)blob.size();
    UP(tab_blob, blob.data(), blob.size());
    std::vector<int32_t> adx;
    for (int a = 0; a < A; ++a) if (spec->kind[a] == PHX_KIND_ADEXCHANGE) adx.push_back(a);
    d.n_adx = (int32_t)adx.size();
    UP(adx_idx, adx.data(), adx.size());
    std::vector<int32_t> nptr = {0}, ne;
    for (int a : adx) {
      for (int k = spec->row_ptr[a]; k < spec->row_ptr[a + 1]; ++k) if (spec->kind[spec->col[k]] == PHX_KIND_ADVERTISER) ne.push_back(k);
      nptr.push_back((int32_t)ne.size());
    }
    UP(adx_nbr_ptr, nptr.data(), nptr.size()); UP(adx_nbr_e, ne.data(), ne.size());
    d.dynamic_graph = der.dynamic_graph ? 1 : 0;
  }
  UP(shop_type_src, der.shop_type_src.data(), der.shop_type_src.size());
  UP(shop_type_prm, der.shop_type_prm.data(), der.shop_type_prm.size());
  d.n_samplers = spec->n_samplers; d.any_typed = der.any_typed ? 1 : 0; d.device_sampling = der.device_sampling ? 1 : 0;
  d.n_tabn = der.n_tabn; d.n_quot = der.n_quot; d.rew_smax = der.rew_smax;
#undef UP
  d.max_cust = der.max_cust;
  d.fsm_lean_K = 0; d.fsm_lean_norm = 0;
  d.sc_all_or_none = 1;
  for (size_t i = 0; i < der.sc_shop_flags.size(); ++i) if ((der.sc_shop_flags[i] & 2) && !(der.sc_shop_flags[i] & 4)) d.sc_all_or_none = 0;
  std::vector<uint32_t> fsm_tab;             // position table of the time-parallel FSM rollout
  if (der.sc_static && spec->env_type == PHX_ENV_FSM && !der.any_typed && d.S > 0 && d.D == 3) {
    // lean FSM rollout loop: every shop with the same 1..6 customers and normaliser, a shop's customers act all or none per stage
    int Ku = der.shop_cust_ptr.size() > 1 ? der.shop_cust_ptr[1] - der.shop_cust_ptr[0] : -1;
    bool ok = true;
    for (int s2 = 0; s2 < d.S; ++s2) {
      if (der.shop_cust_ptr[s2 + 1] - der.shop_cust_ptr[s2] != Ku) Ku = -1;
      ok = ok && der.shop_norm[s2] == der.shop_norm[0];
    }
    for (size_t i = 0; i < der.sc_shop_flags.size(); ++i) ok = ok && (!(der.sc_shop_flags[i] & 2) || (der.sc_shop_flags[i] & 4));
    // (tabulated handlers: the general lane-per-pair loop looks every transition up; the lean loop and the time-parallel
    //  kernel are built on the handler-less stage chain)
    if (ok && Ku >= 1 && Ku <= 6 && der.shop_norm[0] > 0 && !spec->stage_tab) { d.fsm_lean_K = Ku; d.fsm_lean_norm = der.shop_norm[0]; }
    // time-parallel FSM rollout (phx_sc_rollout_fsm.hip): additionally the stage's flags are the same for every shop, the
    // env has no samplers, and along the handler-less chain from the initial stage every lookback the kernel serves from
    // its tiles is at most PHX_FSM_LB steps.  The table holds, per episode position, the flags, the lookbacks and the stage.
    memset(&d.fsm_fast, 0, sizeof d.fsm_fast);
    bool fok = d.fsm_lean_K > 0 && spec->n_samplers == 0 && d.num_steps >= PHX_FAST_TC && d.num_steps <= 4096 && d.n_lists <= 255;
    for (int l = 0; l < d.n_lists && fok; ++l)
      for (int s2 = 1; s2 < d.S; ++s2) fok = fok && der.sc_shop_flags[(size_t)l * d.S + s2] == der.sc_shop_flags[(size_t)l * d.S];
    if (fok) {
      const int ns = d.num_steps, LB = PHX_FSM_LB;
      std::vector<int> stage(ns), fl(ns);
      int sg = spec->initial_stage;
      for (int p = 0; p < ns; ++p) {
        if (sg < 0 || sg >= d.n_lists) { fok = false; break; }
        stage[p] = sg; fl[p] = der.sc_shop_flags[(size_t)sg * d.S];
        sg = spec->stage_next[sg];
      }
      fsm_tab.assign(fok ? ns : 0, 0);
      auto back_in_episode = [&](int p, int bit) { for (int k = 0; k <= p; ++k) if (fl[p - k] & bit) return k; return -1; };
      auto back_cyclic = [&](int p, int bit, bool* same_episode) {
        for (int k = 0; k <= LB; ++k) { const int q = p - k; if (fl[((q % ns) + ns) % ns] & bit) { *same_episode = q >= 0; return k; } }
        *same_episode = false; return -1;
      };
      for (int p = 0; p < ns && fok; ++p) {
        const int f = fl[p];
        uint32_t w = (uint32_t)((f & 1) | ((f & 2) ? 2 : 0) | ((f & 8) ? 4 : 0) | ((f & 16) ? 8 : 0));
        const bool emits = (f & 8) || p == ns - 1;               // the shop observes, or the episode ends: a reward is emitted
        const int lr = back_in_episode(p, 16), lo = back_in_episode(p, 8);
        if (emits && lr > LB) fok = false;
        if (p == ns - 1 && lo != 0) fok = false;                   // the episode's last step observes (no dump of an older observation)
        w |= (uint32_t)((lr < 0 || lr > LB) ? 7 : lr) << 4;
        w |= (uint32_t)((lo < 0 || lo > LB) ? 7 : lo) << 8;
        bool se = false, dummy = false;
        const int cr = back_cyclic(p, 16, &se), co = back_cyclic(p, 8, &dummy), ca = back_cyclic(p, 1, &dummy);
        if (cr < 0 || co < 0 || ca < 0) fok = false;               // the state left behind must be in reach from every position
        w |= (uint32_t)(cr < 0 ? 7 : cr) << 12; w |= (uint32_t)(se ? 1 : 0) << 15;
        w |= (uint32_t)(co < 0 ? 7 : co) << 16; w |= (uint32_t)(ca < 0 ? 7 : ca) << 20;
        w |= (uint32_t)stage[p] << 24;
        fsm_tab[p] = w;
      }
      ScFastPlan plan;
      if (fok && phx_sc_fast_plan(d.B, d.S, d.fsm_lean_K, true, d.num_steps, d.variant_block, false, &plan)) { plan.norm = d.fsm_lean_norm; d.fsm_fast = plan; }
    }
  }
  // round 5: the store-wave kernel's FSM instantiation (phx_sc_rollout_sw.hip, MODE 2) for the same envs: per episode position along the
  // handler-less chain the stage's flags (uniform over the shops) and whether a rewarded position lies at or before it; the episode's last
  // position observes (its terminal dump, fsm.py:360-375, is then the row's own observation).  No lookback limit: the recurrence lanes
  // carry fsm.py's caches.
  memset(&d.fsm_sw, 0, sizeof d.fsm_sw); d.fsm_sw_tab = nullptr;
  std::vector<uint16_t> fsm_sw_tab;
  if (d.fsm_lean_K > 0 && spec->n_samplers == 0 && d.num_steps >= 16 && d.num_steps <= 4096 && d.n_lists <= 255 &&
      (d.variant_rollout == PHX_VR_AUTO || d.variant_rollout == PHX_VR_STORE_WAVES)) {
    bool ok = true;
    for (int l = 0; l < d.n_lists && ok; ++l)
      for (int s2 = 1; s2 < d.S; ++s2) ok = ok && der.sc_shop_flags[(size_t)l * d.S + s2] == der.sc_shop_flags[(size_t)l * d.S];
    const int ns = d.num_steps;
    fsm_sw_tab.assign((size_t)2 * ns, 0);
    int sg = spec->initial_stage; bool has_rew = false;
    for (int p = 0; p < ns && ok; ++p) {
      if (sg < 0 || sg >= d.n_lists) { ok = false; break; }
      const int f = der.sc_shop_flags[(size_t)sg * d.S];
      has_rew = has_rew || (f & 16);
      // (the SWF_* word of phx_sc_rollout_sw.hip: operand masks and flags at the tile word's bits)
      fsm_sw_tab[p] = (uint16_t)(((f & 1) ? 0x407F : 0) | ((f & 2) ? 0x1F00 : 0) | ((f & 8) ? 0x0080 : 0) | ((f & 16) ? 0x2000 : 0) | (has_rew ? 0x8000 : 0));
      fsm_sw_tab[ns + p] = (uint16_t)sg;
      sg = spec->stage_next[sg];
    }
    ok = ok && (fsm_sw_tab[ns - 1] & 0x0080);
    // the state a fragment leaves (delivered_stock, self._rewards, self._observations) is tracked over its last two chunks: the acting /
    // rewarded / observing positions of the (cyclic) chain lie at most 16 steps apart, or never occur
    for (uint16_t bit : {(uint16_t)0x4000, (uint16_t)0x2000, (uint16_t)0x0080}) {
      int first = -1, prev = -1, gap = 0;
      for (int p = 0; p < ns && ok; ++p) if (fsm_sw_tab[p] & bit) { if (first < 0) first = p; else gap = std::max(gap, p - prev); prev = p; }
      if (first >= 0) gap = std::max(gap, first + ns - prev);
      ok = ok && gap <= 16;
    }
    ScSwPlan sw;
    if (ok && phx_sc_sw_plan(d.B, d.S, d.fsm_lean_K, true, ns, d.variant_block, &sw, ns) && (sw.specialised || d.variant_rollout == PHX_VR_STORE_WAVES) && sw.G != 144) {
      sw.norm = d.fsm_lean_norm;
      std::vector<uint8_t> img;
      phx_sc_sw_tables(sw.K, sw.norm, &img);
      const uint8_t* dev_img = nullptr;
      rc = upload(e, img.data(), img.size(), &dev_img);
      if (rc == PHX_OK) rc = upload(e, fsm_sw_tab.data(), fsm_sw_tab.size(), &d.fsm_sw_tab);
      if (rc != PHX_OK) { phx_destroy(e); return rc; }
      d.fsm_sw = sw; d.sc_sw_tables = dev_img;
    }
  }
  d.fsm_pos_tab = nullptr; d.fsm_irregular = nullptr;
  d.sc_sw_exo_first = nullptr; d.sc_sw_guard = nullptr;
  if (d.fsm_fast.ok || d.fsm_sw.ok) {
    const int32_t zero = 0; const int32_t* flag = nullptr;
    rc = upload(e, &zero, 1, &flag);
    if (rc != PHX_OK) { phx_destroy(e); return rc; }
    d.fsm_irregular = (int32_t*)flag;
    d.fsm_gen_host = &e->fsm_gen;
  }
  if (d.fsm_fast.ok) {
    rc = upload(e, fsm_tab.data(), fsm_tab.size(), &d.fsm_pos_tab);
    if (rc != PHX_OK) { phx_destroy(e); return rc; }
  }
  if (der.sc_static && spec->env_type == PHX_ENV_PLAIN && !der.any_typed && d.S > 0) {
    // fast rollout kernel (phx_sc_rollout.hip): every shop with the same 1..6 customers and the same normaliser
    int Ku = der.shop_cust_ptr.size() > 1 ? der.shop_cust_ptr[1] - der.shop_cust_ptr[0] : -1;
    bool nu = true;
    for (int s2 = 0; s2 < d.S; ++s2) {
      if (der.shop_cust_ptr[s2 + 1] - der.shop_cust_ptr[s2] != Ku) Ku = -1;
      nu = nu && der.shop_norm[s2] == der.shop_norm[0];
    }
    // four-pairs-per-thread step kernel (large batches): additionally every shop acts and every customer orders in the env's one list
    bool all_act = d.n_lists == 1;
    for (int s2 = 0; s2 < d.S && all_act; ++s2) all_act = (der.sc_shop_flags[s2] & 7) == 7;
    if (all_act && Ku >= 1 && Ku <= 6 && nu && der.shop_norm[0] > 0 && d.D == 3 && ((int64_t)d.B * d.S) % 4 == 0) { d.sc_wide_K = Ku; d.sc_wide_norm = der.shop_norm[0]; }
    ScFastPlan plan;
    if (phx_sc_fast_plan(d.B, d.S, Ku, nu, d.num_steps, d.variant_block, true, &plan)) {
      plan.norm = der.shop_norm[0];
      d.sc_fast = plan;
    }
    // round 4: the store-wave kernel serves the same envs (planes only) where a 16-pair-aligned workgroup shape exists;
    // variant_rollout PHX_VR_TIME_PARALLEL keeps the round-3 kernel, PHX_VR_STORE_WAVES / PHX_VR_AUTO take this one
    ScSwPlan sw;
    if (d.sc_fast.ok && (d.variant_rollout == PHX_VR_AUTO || d.variant_rollout == PHX_VR_STORE_WAVES) &&
        phx_sc_sw_plan(d.B, d.S, Ku, nu, d.num_steps, d.variant_block, &sw) && (sw.specialised || d.variant_rollout == PHX_VR_STORE_WAVES)) {
      sw.norm = der.shop_norm[0];
      std::vector<uint8_t> img;
      phx_sc_sw_tables(sw.K, sw.norm, &img);
      const uint8_t* dev_img = nullptr;
      rc = upload(e, img.data(), img.size(), &dev_img);
      if (rc != PHX_OK) { phx_destroy(e); return rc; }
      d.sc_sw = sw; d.sc_sw_tables = dev_img;
      // replays (REPLAY instantiation): the exogenous column of each shop's first customer where its customers' columns are consecutive
      // (they are for every env the host layer builds: customers are numbered shop by shop), and the pre-scan's device word
      std::vector<int32_t> first((size_t)d.S, 0);
      bool consecutive = true;
      for (int s2 = 0; s2 < d.S; ++s2) {
        const int c0 = der.shop_cust_ptr[s2];
        first[s2] = der.shop_cust_exo[c0];
        for (int k = 0; k < Ku; ++k) consecutive = consecutive && der.shop_cust_exo[c0 + k] == first[s2] + k;
      }
      d.sc_sw_exo_first = nullptr;
      if (consecutive) { rc = upload(e, first.data(), first.size(), &d.sc_sw_exo_first); if (rc != PHX_OK) { phx_destroy(e); return rc; } }
      const int32_t zero = 0; const int32_t* gw = nullptr;
      rc = upload(e, &zero, 1, &gw);
      if (rc != PHX_OK) { phx_destroy(e); return rc; }
      d.sc_sw_guard = (int32_t*)gw;
    }
  }
  e->state_blob = state_blob; e->state_nbytes = need;
  for (auto& f : e->fields) d.f[f.id] = (char*)state_blob + f.offset;
  d.ws_stride = ws_stride;
  d.lean_lds = lean_lds_spec(spec, der) ? 1 : 0;
  e->lds_ok = ws_stride == 0 || d.lean_lds;
  e->use_fused = der.sc_static;
  e->sc_rules_fused = der.sc_rules_fused && d.n_rules > 0 && SC_RULES_MAX_S >= d.S;
  e->use_stk = der.stk_static;
  e->use_ads = der.ads_static;
  d.ads_pub = der.ads_pub; d.ads_adx = der.ads_adx; d.ads_pub_stage = der.ads_pub_stage;
  e->prices_compressed = der.stk_static;
  he = hipMalloc((void**)&e->inject_dev, sizeof(DevMsg) * PHX_MAX_INJECT);
  if (he != hipSuccess) { phx_destroy(e); return fail(PHX_EHIP, "hipMalloc: %s", hipGetErrorString(he)); }
  {                                   // the finished spec in device memory (DevSpec::self_dev)
    void* p = nullptr;
    he = hipMalloc(&p, sizeof(DevSpec));
    if (he != hipSuccess) { phx_destroy(e); return fail(PHX_EHIP, "hipMalloc: %s", hipGetErrorString(he)); }
    e->dev_allocs.push_back(p);
    d.self_dev = (const DevSpec*)p;
    he = hipMemcpy(p, &d, sizeof(DevSpec), hipMemcpyHostToDevice);
    if (he != hipSuccess) { phx_destroy(e); return fail(PHX_EHIP, "hipMemcpy: %s", hipGetErrorString(he)); }
  }
  // constructor state: zero blob, then Agent.reset() for every agent (env.py:122-124)
  he = hipMemset(state_blob, 0, (size_t)need);
  if (he != hipSuccess) { phx_destroy(e); return fail(PHX_EHIP, "hipMemset: %s", hipGetErrorString(he)); }
  if (d.env_type == PHX_ENV_FSM) {
    std::vector<int32_t> st((size_t)d.B, spec->initial_stage), pv((size_t)d.B, -1);
    (void)hipMemcpy(d.f[F_ENV_STAGE], st.data(), st.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d.f[F_ENV_PREV_STAGE], pv.data(), pv.size() * 4, hipMemcpyHostToDevice);
  }
  he = phx_launch_reset(d, nullptr, nullptr, nullptr, nullptr, nullptr, 0);   // also the constructor's first draws, env.py:118-119, network.py:389
  if (he == hipSuccess) he = hipDeviceSynchronize();
  if (he != hipSuccess) { phx_destroy(e); return fail(PHX_EHIP, "initial reset: %s", hipGetErrorString(he)); }
  *out = e;
  return PHX_OK;
}

void phx_destroy(phx_env* e) {
  if (!e) return;
  for (void* p : e->dev_allocs) (void)hipFree(p);
  if (e->inject_dev) (void)hipFree(e->inject_dev);
  if (e->probe_snapshot) (void)hipFree(e->probe_snapshot);
  delete e;
}

int phx_n_fields(const phx_env* e) { return e ? (int)e->fields.size() : 0; }

int phx_field_info(const phx_env* e, int index, phx_field* out) {
  if (!e || !out || index < 0 || index >= (int)e->fields.size()) return fail(PHX_EINVAL, "bad field index");
  const FieldDef& f = e->fields[index];
  memset(out, 0, sizeof *out);
  out->field_id = f.id; out->dtype = f.dtype; out->offset = f.offset;
  out->dim0 = (int32_t)f.dim0; out->dim1 = (int32_t)f.dim1; out->dim2 = (int32_t)f.dim2; out->kind = f.kind;
  strncpy(out->name, f.name, sizeof(out->name) - 1);
  return PHX_OK;
}

const char* phx_autotune_note(const phx_env* e) { return e ? e->fsm_auto_note.c_str() : ""; }
int phx_uses_fused(const phx_env* e) { return e && (e->use_fused || e->use_stk || e->use_ads) ? 1 : 0; }


// the handle's device becomes the calling thread's current device (a no-op when it already is)
static inline hipError_t use_device(const phx_env* e) {
  int cur = -1;
  hipError_t r = hipGetDevice(&cur);
  if (r != hipSuccess) return r;
  return cur == e->device ? hipSuccess : hipSetDevice(e->device);
}

int phx_sync_fields(phx_env* e, void* stream) {
  if (!e) return fail(PHX_EINVAL, "null env");
  if (!e->prices_compressed) return PHX_OK;
  HIPCHK(use_device(e));
  HIPCHK(phx_launch_stk_materialise(e->d, (hipStream_t)stream));
  return PHX_OK;
}

int phx_reset(phx_env* e, const uint8_t* reset_mask, const double* sampler_values, const uint8_t* conn_on, float* obs,
              uint8_t* obs_valid, void* stream) {
  if (!e) return fail(PHX_EINVAL, "null env");
  if (sampler_values && e->d.n_samplers == 0) return fail(PHX_EINVAL, "sampler_values given but the spec has no samplers");
  if (conn_on && e->d.n_conn == 0) return fail(PHX_EINVAL, "conn_on given but the spec is not a StochasticNetwork");
  HIPCHK(use_device(e));
  HIPCHK(phx_launch_reset(e->d, reset_mask, sampler_values, conn_on, obs, obs_valid, (hipStream_t)stream));
  return PHX_OK;
}

static int check_step_io(const phx_env* e, const phx_step_io* io) {
  if (!io) return fail(PHX_EINVAL, "null io");
  if (e->d.S > 0 && !io->actions && !io->action_valid) return fail(PHX_EINVAL, "actions is NULL");
  if (!io->obs || !io->obs_valid || !io->reward || !io->reward_valid || !io->terminated || !io->truncated ||
      !io->done_valid || !io->all_terminated || !io->all_truncated)
    return fail(PHX_EINVAL, "a required output pointer is NULL");
  // obs rows leave as 16-byte pieces, rewards as f64; the actions are read one f32 per strategic agent by every step
  // kernel, so a row slice actions[t] of a [n, B, S] tensor is fine whatever B * S is
  if (((uintptr_t)io->obs & 15u) != 0 || ((uintptr_t)io->reward & 7u) != 0 || ((uintptr_t)io->actions & 3u) != 0)
    return fail(PHX_EINVAL, "phx_step: obs must be 16-byte aligned, reward 8-byte aligned, actions 4-byte aligned");
  if ((io->msg_log || io->msg_count) && e->d.trace_cap <= 0) return fail(PHX_EINVAL, "msg_log given but trace_cap == 0");
  if (io->shuffle && !(e->d.flags & PHX_F_SHUFFLE_BATCHES)) return fail(PHX_EINVAL, "shuffle given but the spec has no PHX_F_SHUFFLE_BATCHES");
  if (io->next_stage && e->d.env_type != PHX_ENV_FSM) return fail(PHX_EINVAL, "next_stage given but the env is not a FiniteStateMachineEnv");
  return PHX_OK;
}

static int upload_inject(phx_env* e, hipStream_t st) {
  if (e->n_inject > 0)
    HIPCHK(hipMemcpyAsync(e->inject_dev, e->inject_host, sizeof(DevMsg) * e->n_inject, hipMemcpyHostToDevice, st));
  return PHX_OK;
}

int phx_step(phx_env* e, const phx_step_io* io, void* stream) {
  note_reset();
  if (!e) return fail(PHX_EINVAL, "null env");
  HIPCHK(use_device(e));
  int rc = check_step_io(e, io);
  if (rc != PHX_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (e->use_fused && e->n_inject == 0 && !io->next_stage) {       // handler-chosen transitions: generic engine
    HIPCHK(phx_launch_sc_step(e->d, *io, st));
    return PHX_OK;
  }
  if (e->use_stk && e->n_inject == 0) {
    HIPCHK(phx_launch_stk_step(e->d, *io, st));
    return PHX_OK;
  }
  if (e->use_ads && e->n_inject == 0 && !io->msg_log && !io->msg_count && !io->next_stage) {
    HIPCHK(phx_launch_ads_step(e->d, *io, st));
    return PHX_OK;
  }
  if (e->prices_compressed) {       // host-injected messages can address a single price slot: materialise
    HIPCHK(phx_launch_stk_materialise(e->d, st));   // the table and stay on the generic engine from here on
    e->prices_compressed = false; e->use_stk = false;
  }
  GenArgs g; memset(&g, 0, sizeof g); g.io = *io; g.inject = e->inject_dev; g.n_inject = e->n_inject; g.resolve_only = 0; g.timing = nullptr; g.roll_t = -1;
#ifdef PHX_TIMING
  g.timing = phx_gen_timing(1);
#endif
  rc = upload_inject(e, st);
  if (rc != PHX_OK) return rc;
  if (e->n_inject) HIPCHK(hipStreamSynchronize(st));   // inject_host is reused right after
  e->n_inject = 0;
  HIPCHK(phx_launch_generic(e->d, g, e->lds_ok, st));
  return PHX_OK;
}

// The two halves of a step around a HOST-side stage handler that reads agent state (fsm.py:275-307): phx_step_begin runs the acting
// phase and resolve_network() (message log included) and leaves the env's clock words alone; the caller's handler then looks at
// the resolved state and picks the next stage; phx_step_end(io->next_stage) makes the transition and computes observations, rewards
// and done flags.  begin + end == phx_step(io->next_stage).  Always the generic engine (the fused kernels do not split).
static int step_half(phx_env* e, const phx_step_io* io, void* stream, int phase) {
  note_reset();
  if (!e) return fail(PHX_EINVAL, "null env");
  HIPCHK(use_device(e));
  int rc = check_step_io(e, io);
  if (rc != PHX_OK) return rc;
  if (phase == 2 && e->n_inject) return fail(PHX_EINVAL, "phx_step_end with injected messages pending (they belong to phx_step_begin)");
  hipStream_t st = (hipStream_t)stream;
  if (e->prices_compressed) {
    HIPCHK(phx_launch_stk_materialise(e->d, st));
    e->prices_compressed = false; e->use_stk = false;
  }
  GenArgs g; memset(&g, 0, sizeof g); g.io = *io; g.inject = e->inject_dev; g.n_inject = phase == 1 ? e->n_inject : 0;
  g.resolve_only = 0; g.phase = phase; g.timing = nullptr; g.roll_t = -1;
  if (phase == 1) {
    rc = upload_inject(e, st);
    if (rc != PHX_OK) return rc;
    if (e->n_inject) HIPCHK(hipStreamSynchronize(st));
    e->n_inject = 0;
  }
  HIPCHK(phx_launch_generic(e->d, g, e->lds_ok, st));
  return PHX_OK;
}
int phx_step_begin(phx_env* e, const phx_step_io* io, void* stream) { return step_half(e, io, stream, 1); }
int phx_step_end(phx_env* e, const phx_step_io* io, void* stream) { return step_half(e, io, stream, 2); }

int phx_inject(phx_env* e, const phx_msg_rec* msgs, int n) {
  if (!e || (n > 0 && !msgs)) return fail(PHX_EINVAL, "null argument");
  if (e->n_inject + n > PHX_MAX_INJECT) return fail(PHX_ECAPACITY, "at most %d injected messages per resolve", PHX_MAX_INJECT);
  for (int k = 0; k < n; ++k) {
    if (msgs[k].sender >= e->d.A || msgs[k].receiver >= e->d.A) return fail(PHX_EINVAL, "agent index out of range");
    DevMsg m; m.src = msgs[k].sender; m.dst = msgs[k].receiver; m.type = msgs[k].type; m.pad = msgs[k].round; m.p.i = msgs[k].payload.i;
    e->inject_host[e->n_inject++] = m;
  }
  return PHX_OK;
}

int phx_resolve(phx_env* e, int32_t* err, phx_msg_rec* msg_log, int32_t* msg_count, void* stream) {
  note_reset();
  if (!e) return fail(PHX_EINVAL, "null env");
  if ((msg_log || msg_count) && e->d.trace_cap <= 0) return fail(PHX_EINVAL, "msg_log given but trace_cap == 0");
  HIPCHK(use_device(e));
  hipStream_t st = (hipStream_t)stream;
  if (e->prices_compressed) {
    HIPCHK(phx_launch_stk_materialise(e->d, st));
    e->prices_compressed = false; e->use_stk = false;
  }
  GenArgs g; memset(&g, 0, sizeof g);
  g.io.err = err; g.io.msg_log = msg_log; g.io.msg_count = msg_count;
  g.inject = e->inject_dev; g.n_inject = e->n_inject; g.resolve_only = 1; g.timing = nullptr; g.roll_t = -1;
  int rc = upload_inject(e, st);
  if (rc != PHX_OK) return rc;
  if (e->n_inject) HIPCHK(hipStreamSynchronize(st));
  e->n_inject = 0;
  HIPCHK(phx_launch_generic(e->d, g, e->lds_ok, st));
  return PHX_OK;
}

static int rollout_impl(phx_env* e, const phx_rollout_io* io, void* stream);

// PHX_VR_AUTO for an FSM supply chain, where the size rule would take the store-wave instantiation (on four boxes of seven
// the lane-per-pair loop was the faster kernel for config 3 and AUTO took the other one): the FIRST call of a (T, n_frag) shape on a handle
// times both -- one warm-up and two timed launches each, from a copy of the state blob that is restored before the call's own launch --
// and the handle keeps the winner for that shape.  Both kernels produce the same bits (tests/test_gpu_fsm_sw.py), the outputs of the
// probe launches are overwritten by the call's own.  Not while a stream is capturing (phx_fsm_sw_serves declines there already).
static bool fsm_auto_applies(phx_env* e, const phx_rollout_io* io, void* stream) {
  if (!phx_knobs().autotune || !e->use_fused || e->d.env_type != PHX_ENV_FSM || e->d.variant_rollout != PHX_VR_AUTO || io->policy) return false;
  if (io->n_frag >= 2 && (!io->frags || !io->frags[0].terminated)) return false;
  if (io->n_frag < 2 && (io->frags || !io->obs || !io->terminated)) return false;
  return phx_fsm_sw_serves(e->d, *io, (hipStream_t)stream);
}
struct FsmAutoScope {             // the handle's spec with one candidate forced, for the duration of a call
  phx_env* e; int32_t vr, ok;
  FsmAutoScope(phx_env* e_, int choice) : e(e_), vr(e_->d.variant_rollout), ok(e_->d.fsm_sw.ok) { if (choice == 0) e->d.variant_rollout = PHX_VR_STORE_WAVES; else e->d.fsm_sw.ok = 0; }
  ~FsmAutoScope() { e->d.variant_rollout = vr; e->d.fsm_sw.ok = ok; }
};
static int fsm_auto_probe(phx_env* e, const phx_rollout_io* io, void* stream, int* choice) {
  hipStream_t st = (hipStream_t)stream;
  if (!e->probe_snapshot) HIPCHK(hipMalloc(&e->probe_snapshot, (size_t)e->state_nbytes));
  HIPCHK(hipMemcpyAsync(e->probe_snapshot, e->state_blob, (size_t)e->state_nbytes, hipMemcpyDeviceToDevice, st));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  float us[2] = {0.f, 0.f};
  int rc = PHX_OK;
  for (int c = 0; c < 2 && rc == PHX_OK; ++c) {
    FsmAutoScope scope(e, c);
    rc = rollout_impl(e, io, stream);                                         // warm-up (first-touch of the planes, the code object)
    if (rc == PHX_OK) { (void)hipEventRecord(e0, st); rc = rollout_impl(e, io, stream); }
    if (rc == PHX_OK) rc = rollout_impl(e, io, stream);
    if (rc == PHX_OK) {
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) rc = fail(PHX_EHIP, "autotune: hipEventSynchronize");
      else { float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1); us[c] = ms * 500.0f; }
    }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  HIPCHK(hipMemcpyAsync(e->state_blob, e->probe_snapshot, (size_t)e->state_nbytes, hipMemcpyDeviceToDevice, st));
  if (rc != PHX_OK) return rc;
  *choice = us[1] < us[0] ? 1 : 0;
  char buf[192];
  snprintf(buf, sizeof buf, "T=%d n_frag=%d: store-wave %.1f us, lane-per-pair chain %.1f us -> %s", io->T, io->n_frag, us[0], us[1], *choice ? "lane-per-pair chain" : "store-wave");
  e->fsm_auto_note = buf;
  return PHX_OK;
}

int phx_rollout(phx_env* e, const phx_rollout_io* io, void* stream) {
  note_reset();
  if (!e || !io) return fail(PHX_EINVAL, "null argument");
  if (io->T > 0 && fsm_auto_applies(e, io, stream)) {
    HIPCHK(use_device(e));
    const uint64_t key = ((uint64_t)(uint32_t)io->T << 8) | (uint64_t)(uint32_t)(io->n_frag & 0xff);
    auto it = e->fsm_auto.find(key);
    int choice = 0;
    if (it != e->fsm_auto.end()) choice = it->second;
    else {
      const int rc = fsm_auto_probe(e, io, stream, &choice);
      if (rc != PHX_OK) return rc;
      e->fsm_auto[key] = choice;
      note_reset();
    }
    FsmAutoScope scope(e, choice);
    return rollout_impl(e, io, stream);
  }
  return rollout_impl(e, io, stream);
}

static int rollout_impl(phx_env* e, const phx_rollout_io* io, void* stream) {
  if (!e || !io) return fail(PHX_EINVAL, "null argument");
  if (e->d.env_type != PHX_ENV_PLAIN && !(io->n_frag >= 2 || io->frags) && (!io->obs_valid || !io->reward_valid))
    return fail(PHX_EINVAL, "FSM / Stackelberg rollouts need obs_valid and reward_valid outputs");
  if ((io->hints & ~(PHX_RH_ACTIONS_IN_DOMAIN | PHX_RH_EXO_IN_DOMAIN)) != 0 || io->reserved_ptr)
    return fail(PHX_EINVAL, "phx_rollout: unknown hint bits / reserved_ptr must be NULL (ABI 9 removed PHX_RH_FLAGS_ZEROED and the record layout)");
  if (io->policy && (io->actions || io->n_frag >= 2 || io->frags || io->msg_log || io->msg_count))
    return fail(PHX_EINVAL, "phx_rollout: `policy` excludes replayed actions, fragment lists and message logs");
  if (io->n_frag >= 2 || io->frags) {   // ABI 9: a fragment list
    if (io->n_frag < 2 || io->n_frag > PHX_MAX_FRAGMENTS || !io->frags) return fail(PHX_EINVAL, "phx_rollout: a fragment list needs 2 .. %d fragments and `frags`", PHX_MAX_FRAGMENTS);
    if (io->T <= 0 || io->T % io->n_frag) return fail(PHX_EINVAL, "phx_rollout: T must be a positive multiple of n_frag");
    if (io->obs || io->action_out || io->reward || io->terminated || io->truncated || io->obs_valid || io->reward_valid)
      return fail(PHX_EINVAL, "phx_rollout: with a fragment list the io's own planes must be NULL");
    const bool need_valid = e->d.env_type != PHX_ENV_PLAIN;
    for (int f = 0; f < io->n_frag; ++f) {
      const phx_rollout_frag& fr = io->frags[f];
      if (!fr.obs || !fr.action_out || !fr.reward || !fr.truncated || (need_valid && (!fr.obs_valid || !fr.reward_valid)))
        return fail(PHX_EINVAL, "phx_rollout: fragment %d lacks a required plane", f);
      if ((fr.terminated != nullptr) != (io->frags[0].terminated != nullptr)) return fail(PHX_EINVAL, "phx_rollout: `terminated` must be given for every fragment or for none");
      const void* bufs[] = {fr.obs, fr.action_out, fr.reward, fr.terminated, fr.truncated, fr.obs_valid, fr.reward_valid};
      for (const void* p : bufs) if (((uintptr_t)p & 15u) != 0) return fail(PHX_EINVAL, "phx_rollout: every buffer must be 16-byte aligned");
    }
    if (((uintptr_t)io->last_obs & 15u) || ((uintptr_t)io->actions & 3u)) return fail(PHX_EINVAL, "phx_rollout: every output buffer must be 16-byte aligned (replayed actions: 4-byte)");
    const int Tf = io->T / io->n_frag;
    // ONE launch where the store-wave supply-chain kernel serves the env (its store waves switch planes at the fragments' first rows) ...
    if (e->use_fused && e->d.env_type == PHX_ENV_PLAIN && !e->d.any_typed && e->d.sc_fast.ok && e->d.sc_sw.ok && !io->msg_log && !io->msg_count &&
        (!io->actions || (io->hints & PHX_RH_ACTIONS_IN_DOMAIN)) && (!io->exo || (e->d.sc_sw_exo_first && (io->hints & PHX_RH_EXO_IN_DOMAIN))) &&   // replays: only vouched-for
        e->d.variant_rollout != PHX_VR_GENERAL && io->T <= 0xFFFF && (e->d.variant_rollout == PHX_VR_STORE_WAVES || io->T >= 40)) {                // ones (round 1's kernel has no fragment lists)
      HIPCHK(use_device(e));
      HIPCHK(phx_launch_sc_rollout_sw(e->d, *io, (hipStream_t)stream));
      return PHX_OK;
    }
    // ... the same for an FSM supply chain on its FSM instantiation; the lane-per-pair loop that takes over when an env is off the stage
    // chain has no fragment lists: one guarded launch per fragment behind it (each returns at entry unless the check found such an env)
    if (e->use_fused && e->d.env_type == PHX_ENV_FSM && io->frags[0].terminated && phx_fsm_sw_serves(e->d, *io, (hipStream_t)stream)) {   // (`terminated`: the loop behind needs it)
      HIPCHK(use_device(e));
      const int32_t gen = phx_fsm_next_gen(e->d);
      HIPCHK(phx_launch_sc_rollout_sw(e->d, *io, (hipStream_t)stream, gen));
      for (int f = 0; f < io->n_frag; ++f) {
        const phx_rollout_frag& fr = io->frags[f];
        phx_rollout_io sub = *io;
        sub.n_frag = 0; sub.frags = nullptr; sub.T = Tf;
        sub.obs = fr.obs; sub.action_out = fr.action_out; sub.reward = fr.reward; sub.terminated = fr.terminated; sub.truncated = fr.truncated;
        sub.obs_valid = fr.obs_valid; sub.reward_valid = fr.reward_valid;
        HIPCHK(phx_launch_sc_rollout_fsm(e->d, sub, (hipStream_t)stream, e->d.fsm_irregular, gen));
      }
      return PHX_OK;
    }
    // ... n_frag consecutive launches everywhere else
    for (int f = 0; f < io->n_frag; ++f) {
      const phx_rollout_frag& fr = io->frags[f];
      phx_rollout_io sub = *io;
      sub.n_frag = 0; sub.frags = nullptr; sub.T = Tf;
      sub.obs = fr.obs; sub.action_out = fr.action_out; sub.reward = fr.reward; sub.terminated = fr.terminated; sub.truncated = fr.truncated;
      sub.obs_valid = fr.obs_valid; sub.reward_valid = fr.reward_valid;
      const int64_t row = (int64_t)f * Tf * e->d.B;
      if (io->actions) sub.actions = io->actions + row * e->d.S;
      if (io->exo) sub.exo = io->exo + row * e->d.n_exo;
      if (io->msg_log) sub.msg_log = io->msg_log + row * e->d.trace_cap;
      if (io->msg_count) sub.msg_count = io->msg_count + row;
      const int rc = rollout_impl(e, &sub, stream);
      if (rc != PHX_OK) return rc;
    }
    return PHX_OK;
  }
  if (io->T <= 0 || !io->obs || !io->action_out || !io->reward || !io->truncated)
    return fail(PHX_EINVAL, "bad rollout io");
  // `terminated` may be NULL where the plane would be all zero AND the kernel that serves the launch can leave it out: the
  // time-parallel supply-chain kernel (ShopAgent never terminates, agents.py:292-323).  4.5 % of the trajectory bytes, ~8 % of the launch.
  if (!io->terminated && !(e->use_fused && e->d.env_type == PHX_ENV_PLAIN && !e->d.any_typed && e->d.sc_fast.ok && !io->actions && !io->exo &&
                           e->d.variant_rollout != PHX_VR_GENERAL))
    return fail(PHX_EINVAL, "phx_rollout: `terminated` is required for this env (only the time-parallel supply-chain rollout can omit the all-zero plane)");
  if ((io->msg_log || io->msg_count) && (e->d.trace_cap <= 0 || !io->msg_log || !io->msg_count))
    return fail(PHX_EINVAL, "rollout message log needs trace_cap > 0 and both msg_log and msg_count");
  {                                   // the rollout kernels write 16-byte pieces
    const void* bufs[] = {io->obs, io->action_out, io->reward, io->terminated, io->truncated, io->obs_valid, io->reward_valid, io->last_obs};
    for (const void* p : bufs)
      if (((uintptr_t)p & 15u) != 0) return fail(PHX_EINVAL, "phx_rollout: every output buffer must be 16-byte aligned");
    // (the replayed inputs are read one float / one byte at a time: a row slice of a longer recording is fine whatever B S is)
    if (((uintptr_t)io->actions & 3u) != 0) return fail(PHX_EINVAL, "phx_rollout: `actions` must be 4-byte aligned");
  }
  if (e->d.n_samplers > 0 && !e->d.device_sampling)
    return fail(PHX_EUNSUPPORTED, "phx_rollout auto-resets on the device: every sampler must be PHX_SAMPLER_UNIFORM");
  HIPCHK(use_device(e));
  if (io->policy) {                   // ABI 10: the policy evaluated on the device, one lane per (env, shop) (phx_sc_policy.hip)
    if (!e->use_fused) return fail(PHX_EUNSUPPORTED, "phx_rollout: `policy` needs a plain supply-chain env on its fused schedule");
    const char* why = phx_sc_policy_unsupported(e->d, *io);
    if (why) return fail(strstr(why, "phx_policy_mlp") ? PHX_EINVAL : PHX_EUNSUPPORTED, "phx_rollout: %s", why);
    HIPCHK(phx_launch_sc_rollout_policy(e->d, *io, (hipStream_t)stream));
    return PHX_OK;
  }
  if (e->use_ads && e->n_inject == 0) {
    if (!io->obs_valid || !io->reward_valid) return fail(PHX_EINVAL, "FSM rollouts need obs_valid and reward_valid outputs");
    HIPCHK(phx_launch_ads_rollout(e->d, *io, (hipStream_t)stream));
    return PHX_OK;
  }
  if (e->use_stk && e->prices_compressed && (phx_stk_rollout_lds(e->d) > 60 * 1024 || e->d.A > 3 * 1024) && e->d.f[F_ROLLOUT_SCRATCH]) {
    // a market too large for the LDS-resident rollout kernel: materialise the price table and roll out on the generic
    // engine (the env stays there, like after a host-injected message)
    HIPCHK(phx_launch_stk_materialise(e->d, (hipStream_t)stream));
    e->prices_compressed = false; e->use_stk = false;
  }
  if (e->sc_rules_fused && e->d.variant_rollout != PHX_VR_LAUNCH_LOOP && e->d.variant_step != PHX_VS_GENERIC_DYNAMIC && !io->msg_log && !io->msg_count && e->n_inject == 0) {
    // an FSM supply chain whose handlers are rules: the fused lane-per-pair loop evaluates them (round 6; until then the engine's T-step loop)
    if (!io->obs_valid || !io->reward_valid || !io->terminated) return fail(PHX_EINVAL, "FSM rollouts need terminated, obs_valid and reward_valid outputs");
    HIPCHK(phx_launch_sc_rollout_fsm_rules(e->d, *io, (hipStream_t)stream));
    return PHX_OK;
  }
  if (!e->use_fused && !(e->use_stk && e->prices_compressed)) {
    // Launch loop for every other env (any topology of the device kinds, tracking off): per step ONE
    // launch of the generic engine (policy, trajectory row and auto-reset fused in), stream ordered,
    // the step-shaped intermediates in the blob's rollout.scratch field.
    if (!e->d.f[F_ROLLOUT_SCRATCH]) return fail(PHX_EUNSUPPORTED, "phx_rollout: this env switched engines after creation");
    if (e->n_inject) return fail(PHX_EINVAL, "phx_rollout with injected messages pending");
    hipStream_t st = (hipStream_t)stream;
    const GenScratch gs = gen_scratch(e->d.B, std::max(e->d.S, 1), e->d.D);
    char* base = (char*)e->d.f[F_ROLLOUT_SCRATCH];
    phx_step_io sio; memset(&sio, 0, sizeof sio);
    sio.actions = (const float*)(base + gs.actions);
    sio.obs = (float*)(base + gs.obs); sio.obs_valid = (uint8_t*)(base + gs.obs_valid);
    sio.reward = (double*)(base + gs.reward); sio.reward_valid = (uint8_t*)(base + gs.reward_valid);
    sio.terminated = (uint8_t*)(base + gs.terminated); sio.truncated = (uint8_t*)(base + gs.truncated);
    sio.done_valid = (uint8_t*)(base + gs.done_valid);
    sio.all_terminated = (uint8_t*)(base + gs.all_term); sio.all_truncated = (uint8_t*)(base + gs.all_trunc);
    sio.err = io->err;
    GenArgs g; memset(&g, 0, sizeof g); g.io = sio; g.inject = e->inject_dev; g.n_inject = 0; g.resolve_only = 0; g.timing = nullptr;
    g.roll = *io; g.roll_actions_in = io->actions; g.roll_actions = (float*)(base + gs.actions);
    if (e->d.variant_rollout != PHX_VR_LAUNCH_LOOP) {
      // ONE launch: the kernel loops over the T steps itself (GenArgs::roll_T) -- queues, staged tables and the env's
      // workgroup stay resident; policy, trajectory row, the caller's reset and the last observation are in the loop
      sio.exo = io->exo; sio.msg_log = io->msg_log; sio.msg_count = io->msg_count;
      g.io = sio; g.roll_t = 0; g.roll_T = io->T;
#ifdef PHX_TIMING
      g.timing = phx_gen_timing(io->T);
#endif
      HIPCHK(phx_launch_generic(e->d, g, e->lds_ok, st));
      return PHX_OK;
    }
    for (int t = 0; t < io->T; ++t) {            // one launch per step: policy + step + trajectory row + the caller's reset
      sio.exo = io->exo ? io->exo + (int64_t)t * e->d.B * e->d.n_exo : nullptr;
      sio.msg_log = io->msg_log ? io->msg_log + (int64_t)t * e->d.B * e->d.trace_cap : nullptr;   // rollout.py:369-373
      sio.msg_count = io->msg_count ? io->msg_count + (int64_t)t * e->d.B : nullptr;
      g.io = sio; g.roll_t = t;
      HIPCHK(phx_launch_generic(e->d, g, e->lds_ok, st));
    }
    if (io->last_obs) HIPCHK(phx_launch_gen_last_obs(e->d, sio.obs, io->last_obs, st));
    return PHX_OK;
  }
  if (e->use_stk) {
    if (io->exo) return fail(PHX_EINVAL, "the market has no exogenous draws");
    if (phx_stk_rollout_lds(e->d) > 60 * 1024 || e->d.A > 3 * 1024) return fail(PHX_EUNSUPPORTED, "market too large for the LDS-resident rollout");
    HIPCHK(phx_launch_stk_rollout(e->d, *io, (hipStream_t)stream)); return PHX_OK;
  }
  // typed shops (obs dim 4, per-env penalty weight) take the lane-per-pair kernel too
  if (e->d.env_type == PHX_ENV_FSM || e->d.any_typed) { HIPCHK(phx_launch_sc_rollout_fsm(e->d, *io, (hipStream_t)stream)); return PHX_OK; }
  if (e->d.sc_fast.ok && !io->actions && !io->exo && e->d.variant_rollout != PHX_VR_GENERAL) {
    // the store-wave kernel where the caller asked for it, or (PHX_VR_AUTO) where a compile-time shape serves the env and the fragment
    // is longer than its pipeline fill (SC64, B = 4 096: T = 32 14.1 us either way, T = 50 14.7 against 16.4, T = 100 21.2 against
    // 23.1, T = 400 57 against 72-76; several rounds of workgroups: B = 16 384 221 against 242; SC256, B = 8 192, T = 100: 195 against 204)
    if (e->d.sc_sw.ok && io->T <= 0xFFFF && (e->d.variant_rollout == PHX_VR_STORE_WAVES || io->T >= 40)) HIPCHK(phx_launch_sc_rollout_sw(e->d, *io, (hipStream_t)stream));
    else HIPCHK(phx_launch_sc_rollout_fast(e->d, *io, (hipStream_t)stream));
    return PHX_OK;
  }
  // Replayed actions and / or order sizes (a recorded policy, the reference's own numpy stream): the store-wave kernel's REPLAY
  // instantiation where its shape serves the env.  Its tiles hold the stock in a byte: a call with an action that rounds below zero
  // (a negative StockRequest takes the stock below zero) is found by a pre-scan of the call's actions and served by round 1's kernel
  // -- both launches are issued, the device word decides which one runs.
  if (e->d.sc_fast.ok && e->d.sc_sw.ok && e->d.variant_rollout != PHX_VR_GENERAL && e->d.variant_rollout != PHX_VR_TIME_PARALLEL && io->T <= 0xFFFF &&
      (e->d.variant_rollout == PHX_VR_STORE_WAVES || io->T >= 40) && e->d.sc_sw_guard &&
      (!io->exo || (e->d.sc_sw_exo_first && (io->hints & PHX_RH_EXO_IN_DOMAIN)))) {      // (order sizes: only those the caller vouches for, see the hint)
    const bool scan = io->actions && !(io->hints & PHX_RH_ACTIONS_IN_DOMAIN);
    const int32_t gen = scan ? ++e->sw_guard_gen : 0;
    HIPCHK(phx_launch_sc_rollout_sw(e->d, *io, (hipStream_t)stream, gen));
    if (scan) HIPCHK(phx_launch_sc_rollout(e->d, *io, (hipStream_t)stream, e->d.sc_sw_guard, gen));
    return PHX_OK;
  }
  HIPCHK(phx_launch_sc_rollout(e->d, *io, (hipStream_t)stream));
  return PHX_OK;
}

// ---- copying state access by field name -------------------------------------------------
static const FieldDef* find_field_by_name(const phx_env* e, const char* name) {
  for (const FieldDef& f : e->fields) if (!strcmp(f.name, name)) return &f;
  return nullptr;
}
static int64_t field_nbytes(const FieldDef& f) {
  static const int esz[4] = {4, 8, 1, 4};
  return (int64_t)f.dim0 * f.dim1 * f.dim2 * esz[f.dtype];
}

int64_t phx_get_state(phx_env* e, const char* field, void* buf, int64_t buf_nbytes, void* stream) {
  if (!e || !field || !buf) return fail(PHX_EINVAL, "null argument");
  const FieldDef* f = find_field_by_name(e, field);
  if (!f) return fail(PHX_EINVAL, "unknown state field '%s'", field);
  const int64_t nb = field_nbytes(*f);
  if (!e->d.f[f->id]) return fail(PHX_EINVAL, "field '%s' is not bound", field);
  if (buf_nbytes < nb) return fail(PHX_EINVAL, "buffer of %lld bytes for field '%s' of %lld bytes", (long long)buf_nbytes, field, (long long)nb);
  HIPCHK(use_device(e));
  if (e->prices_compressed && !strcmp(field, "buyer.prices")) HIPCHK(phx_launch_stk_materialise(e->d, (hipStream_t)stream));
  HIPCHK(hipMemcpyAsync(buf, (const char*)e->d.f[f->id], (size_t)nb, hipMemcpyDefault, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return nb;
}

int64_t phx_set_state(phx_env* e, const char* field, const void* buf, int64_t buf_nbytes, void* stream) {
  if (!e || !field || !buf) return fail(PHX_EINVAL, "null argument");
  const FieldDef* f = find_field_by_name(e, field);
  if (!f) return fail(PHX_EINVAL, "unknown state field '%s'", field);
  const int64_t nb = field_nbytes(*f);
  if (!e->d.f[f->id]) return fail(PHX_EINVAL, "field '%s' is not bound", field);
  if (buf_nbytes != nb) return fail(PHX_EINVAL, "field '%s' holds %lld bytes, got %lld", field, (long long)nb, (long long)buf_nbytes);
  HIPCHK(use_device(e));
  if (e->prices_compressed && !strcmp(field, "buyer.prices"))
    return fail(PHX_EUNSUPPORTED, "buyer.prices is kept compressed by the fused market kernel: set seller.posted instead");
  HIPCHK(hipMemcpyAsync((char*)e->d.f[f->id], buf, (size_t)nb, hipMemcpyDefault, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return nb;
}

int phx_trace(phx_env* e, const phx_msg_rec* msg_log, const int32_t* msg_count, int b, phx_msg_rec* out, int cap, void* stream) {
  if (!e || !msg_log || !msg_count || (cap > 0 && !out)) return fail(PHX_EINVAL, "null argument");
  if (e->d.trace_cap <= 0) return fail(PHX_EINVAL, "tracking is off (trace_cap == 0)");
  if (b < 0 || b >= e->d.B) return fail(PHX_EINVAL, "env index out of range");
  HIPCHK(use_device(e));
  int32_t n = 0;
  HIPCHK(hipMemcpyAsync(&n, msg_count + b, sizeof n, hipMemcpyDefault, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  int m = n < e->d.trace_cap ? n : e->d.trace_cap;
  if (m > cap) m = cap;
  if (m > 0) {
    HIPCHK(hipMemcpyAsync(out, msg_log + (int64_t)b * e->d.trace_cap, sizeof(phx_msg_rec) * m, hipMemcpyDefault, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  }
  return n;
}

// ---- done-flag planes, bit-packed for rollout collection (SURVEY 8e iii) ----------------------------
__global__ __launch_bounds__(256) void phx_pack_flags_kernel(const uint8_t* __restrict__ src, uint64_t* __restrict__ dst, const int64_t n) {
  // one wave packs 64 consecutive bytes into one word with a ballot; consecutive waves take consecutive words
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool on = i < n && src[i] != 0;
  const uint64_t bits = __ballot(on);
  if ((threadIdx.x & 63) == 0 && (i >> 6) < ((n + 63) >> 6)) dst[i >> 6] = bits;
}
__global__ __launch_bounds__(256) void phx_unpack_flags_kernel(const uint64_t* __restrict__ src, uint8_t* __restrict__ dst, const int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (uint8_t)((src[i >> 6] >> (i & 63)) & 1u);
}

int phx_pack_flags(const uint8_t* src, uint64_t* dst, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!src || !dst))) return fail(PHX_EINVAL, "bad argument");
  if (n == 0) return PHX_OK;
  hipLaunchKernelGGL(phx_pack_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  HIPCHK(hipGetLastError());
  return PHX_OK;
}
int phx_unpack_flags(const uint64_t* src, uint8_t* dst, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!src || !dst))) return fail(PHX_EINVAL, "bad argument");
  if (n == 0) return PHX_OK;
  hipLaunchKernelGGL(phx_unpack_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  HIPCHK(hipGetLastError());
  return PHX_OK;
}

// ---- ABI 7: per-env legacy-numpy MT19937 streams (PHX_F_MT19937; supply_chain.py:64's np.random.randint(5)) -----------------
// np.random.seed(seed) of a 32-bit integer = init_genrand (numpy random/src/mt19937/mt19937.c: mt19937_seed): one lane per env
__global__ __launch_bounds__(256) void phx_mt_seed_kernel(const uint32_t* __restrict__ seeds, uint32_t* __restrict__ state,
                                                          int32_t* __restrict__ pos, const int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  uint32_t* mt = state + (int64_t)b * 624;
  uint32_t x = seeds[b];
  mt[0] = x;
  for (uint32_t i = 1; i < 624; ++i) { x = 1812433253u * (x ^ (x >> 30)) + i; mt[i] = x; }
  pos[b] = 624;                                               // the first draw regenerates
}
// One wave per env: the state is staged in LDS, regenerated 64 words at a time (word k of the next state needs the OLD words k and
// k + 1 and, k < 227, the old word k + 397, otherwise the NEW word k - 227: chunks of 64 in order have them all), and 64 words per
// pass are tempered, masked (& 7) and rejected (> 4) with the accepted ones compacted by a ballot -- the words a sequence of
// np.random.randint(5) calls would consume, in order; the pass that reaches a step's last draw consumes up to ITS word only.
// Step by step: the drawing agents of a step are those of the env's acting list (FSM: of its stage, which is walked forward from
// the env's words along stage_next / stage_tab with the reset at the episode's end, as phx_rollout walks it).
struct MtArgs {
  uint32_t* state; int32_t* pos; uint8_t* exo;
  const int32_t *mt_ptr, *mt_rank, *stage_next, *stage_tab, *env_step, *env_stage;
  int32_t B, n_exo, T, fsm, num_steps, initial_stage, zero_rows;
};
__global__ __launch_bounds__(64) void phx_mt_draw_kernel(const MtArgs a) {
  __shared__ uint32_t mt[624];
  const int b = blockIdx.x, lane = threadIdx.x;
  uint32_t* gs = a.state + (int64_t)b * 624;
  for (int i = lane; i < 624; i += 64) mt[i] = gs[i];
  int p = a.pos[b];
  int step = a.fsm ? a.env_step[b] : 0, stage = a.fsm ? a.env_stage[b] : 0;
  __syncthreads();
  for (int t = 0; t < a.T; ++t) {
    const int base = a.mt_ptr[stage], need = a.mt_ptr[stage + 1] - base;
    uint8_t* row = a.exo + ((int64_t)t * a.B + b) * a.n_exo;
    if (a.zero_rows) {                                          // customers that do not act in this step draw nothing: their entries are 0
      for (int j = lane; j < a.n_exo; j += 64) row[j] = 0;       // (one wave: these stores are issued before the draws' stores below)
      __syncthreads();
    }
    int have = 0;
    while (have < need) {
      if (p >= 624) {                                         // genrand's regeneration (mt19937_gen)
        for (int c = 0; c < 624; c += 64) {
          const int k = c + lane;
          uint32_t y = 0, src = 0;
          if (k < 624) {
            y = (mt[k] & 0x80000000u) | (mt[k == 623 ? 0 : k + 1] & 0x7fffffffu);
            src = k < 227 ? mt[k + 397] : mt[k - 227];
          }
          __syncthreads();
          if (k < 624) mt[k] = src ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
          __syncthreads();
        }
        p = 0;
      }
      const int len = 624 - p < 64 ? 624 - p : 64;
      uint32_t y = lane < len ? mt[p + lane] : 0u;
      y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
      const uint32_t v = y & 7u;                              // mask of rng = 4 (buffered_bounded_masked_uint32)
      const bool acc = lane < len && v <= 4u;
      const uint64_t m = __ballot(acc);
      const int before = __popcll(m & ((1ull << lane) - 1ull)), total = __popcll(m);
      const int left = need - have;
      if (acc && before < left) row[a.mt_rank[base + have + before]] = (uint8_t)v;
      if (total <= left) { have += total; p += len; }
      else {                                                  // the step's last draw is inside this pass: stop behind its word
        const uint64_t last = __ballot(acc && before == left - 1);
        p += __ffsll((long long)last);                        // 1-based lane of that word = words consumed
        have = need;
      }
    }
    if (a.fsm) {                                              // fsm.py:281-307 / the auto-reset of a rollout (fsm.py:217)
      const int tn = step + 1;
      stage = a.stage_tab ? a.stage_tab[(int64_t)stage * (a.num_steps + 1) + (tn <= a.num_steps ? tn : a.num_steps)] : a.stage_next[stage];
      step = tn;
      if (step >= a.num_steps) { step = 0; stage = a.initial_stage; }
    }
  }
  __syncthreads();
  for (int i = lane; i < 624; i += 64) gs[i] = mt[i];
  if (lane == 0) a.pos[b] = p;
}

static int mt_check(phx_env* e) {
  if (!e) return fail(PHX_EINVAL, "null env");
  if (!e->d.f[F_ENV_MT_STATE]) return fail(PHX_EUNSUPPORTED, "the spec was compiled without PHX_F_MT19937");
  return PHX_OK;
}
int phx_mt_seed(phx_env* e, const uint32_t* seeds, void* stream) {
  int rc = mt_check(e);
  if (rc != PHX_OK) return rc;
  if (!seeds) return fail(PHX_EINVAL, "seeds is NULL");
  hipStream_t st = (hipStream_t)stream;
  uint32_t* dseeds = nullptr;
  HIPCHK(hipMalloc((void**)&dseeds, sizeof(uint32_t) * e->d.B));
  hipError_t he = hipMemcpyAsync(dseeds, seeds, sizeof(uint32_t) * e->d.B, hipMemcpyHostToDevice, st);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(phx_mt_seed_kernel, dim3((e->d.B + 255) / 256), dim3(256), 0, st, dseeds, (uint32_t*)e->d.f[F_ENV_MT_STATE],
                       (int32_t*)e->d.f[F_ENV_MT_POS], e->d.B);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipStreamSynchronize(st);        // `seeds` is the caller's host memory, the staging buffer is ours
  (void)hipFree(dseeds);
  HIPCHK(he);
  return PHX_OK;
}
int phx_mt_draw(phx_env* e, uint8_t* exo, int T, void* stream) {
  int rc = mt_check(e);
  if (rc != PHX_OK) return rc;
  if (!exo || T < 1) return fail(PHX_EINVAL, "bad argument");
  if (e->d.env_type != PHX_ENV_PLAIN && e->d.env_type != PHX_ENV_FSM) return fail(PHX_EUNSUPPORTED, "phx_mt_draw: PHX_ENV_PLAIN and PHX_ENV_FSM only");
  if (e->d.n_exo < 1) return fail(PHX_EUNSUPPORTED, "the env has no exogenous draws");
  // (a publisher's binomial draws depend on the auction's outcome: they cannot be drawn ahead of the step)
  if (e->der.kind_count[PHX_KIND_PUBLISHER] > 0) return fail(PHX_EUNSUPPORTED, "phx_mt_draw: PublisherAgent draws depend on the auction");
  hipStream_t st = (hipStream_t)stream;
  bool all = e->d.n_lists == 1;                                // every drawing agent draws in every step: no entry is left unwritten
  if (all) all = e->der.act_ptr.size() >= 2 && [&] { int n = 0; for (int k = e->der.act_ptr[0]; k < e->der.act_ptr[1]; ++k) n += e->der.exo_rank[e->der.act_idx[k]] >= 0; return n == e->d.n_exo; }();
  MtArgs a;
  a.state = (uint32_t*)e->d.f[F_ENV_MT_STATE]; a.pos = (int32_t*)e->d.f[F_ENV_MT_POS]; a.exo = exo;
  a.mt_ptr = e->d.mt_ptr; a.mt_rank = e->d.mt_rank; a.stage_next = e->d.stage_next; a.stage_tab = e->d.stage_tab;
  a.env_step = (const int32_t*)e->d.f[F_ENV_STEP]; a.env_stage = (const int32_t*)e->d.f[F_ENV_STAGE];
  a.B = e->d.B; a.n_exo = e->d.n_exo; a.T = T; a.fsm = e->d.env_type == PHX_ENV_FSM ? 1 : 0;
  a.num_steps = e->d.num_steps; a.initial_stage = e->d.initial_stage; a.zero_rows = all ? 0 : 1;
  hipLaunchKernelGGL(phx_mt_draw_kernel, dim3(e->d.B), dim3(64), 0, st, a);
  HIPCHK(hipGetLastError());
  return PHX_OK;
}

}  // extern "C"

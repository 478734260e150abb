// phx_dev.h -- device-side view of a compiled env spec + the per-kind agent behaviour.
//
// The reference executes these bodies as Python methods on agent objects; here they are
// inlined device functions over struct-of-arrays state in HBM.  One field of one kind is a
// dense [B][n_kind] array so that lanes mapped to (env, agent-of-kind) touch consecutive
// addresses.  Reference line numbers refer to /root/reference.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/phantom_amd.h"

#define PHX_SHOP_MAX_STOCK 100   // supply_chain.py:13
#define PHX_MAX_INJECT 16

// message record as it travels through the per-env queues (LDS or workspace)
struct __attribute__((aligned(16))) DevMsg {
  uint16_t src, dst, type, pad;
  union { int64_t i; double f; } p;
};
static_assert(sizeof(DevMsg) == 16, "DevMsg must be 16 bytes");

// state field ids (index into DevSpec::f[])
enum {
  F_ENV_STEP = 0, F_ENV_STAGE, F_ENV_PREV_STAGE, F_ENV_TICK, F_ENV_CLOCK,
  F_ENV_TERM, F_ENV_TRUNC, F_ENV_REW_CACHE, F_ENV_REW_CACHE_VALID, F_ENV_OBS_CACHE,
  F_ENV_OBS_CACHE_VALID, F_ENV_SAMPLER, F_ENV_EPISODE, F_NET_CONN_ON,
  F_SHOP_STOCK, F_SHOP_SALES, F_SHOP_MISSED, F_SHOP_DELIVERED,
  F_SELLER_PRICE, F_SELLER_REVENUE, F_SELLER_TX, F_SELLER_POSTED,
  F_BUYER_PRICES, F_BUYER_PAID, F_BUYER_BOUGHT,
  F_CASHBOX_TOTAL,
  F_REQRESP_REQ, F_REQRESP_RES,
  F_MOCK_ENC, F_MOCK_DEC, F_MOCK_REW,
  F_ADV_LEFT, F_ADV_BID, F_ADV_LEFT_TAG, F_ADV_BID_TAG, F_ADV_CLICKS, F_ADV_WINS, F_ADV_USER,
  F_ADV_TOT_CLICKS, F_ADV_TOT_REQUESTS, F_ADV_TOT_WINS, F_PUB_ADS_SEEN,
  F_WORKSPACE, F_ROLLOUT_SCRATCH,
  F_ENV_ARRIVE,        // i32 [B]: blocks of a time-parallel rollout launch that have finished with the env (0 between launches)
  F_ENV_MT_STATE,      // u32 [B][624]: the env instance's legacy-numpy MT19937 state (PHX_F_MT19937: phx_mt_seed / phx_mt_draw)
  F_ENV_MT_POS,        // i32 [B]: index of the next word of that state (624: regenerate first)
  F_COUNT
};

// block shape and constant-table offsets of the round-2 supply-chain rollout kernel (phx_sc_rollout.hip)
#define PHX_SCHED_MAX_ROUNDS 8   // rounds a precomputed schedule of the generic engine may hold (DevSpec::sched)
#define PHX_FSM_LB 3            // lookback (steps) the time-parallel FSM rollout serves from its tiles
#define PHX_FAST_TC 20          // steps per chunk (one Philox block serves 4 ticks: 5 row quads)
struct ScFastPlan {
  int32_t ok, epb, G, K, nt, norm, whole_envs;
};

// DEVELOPMENT knobs (A/B runs of kernel variants: tools/roll_time.py, tools/gen_time.py).  They are read from the environment in ONE
// place, once per process, at the first phx_create (phx_knobs(), phx_api.hip); the plans and launchers take them from this table --
// no getenv in a launcher.  Per-env, supported selection of kernels is phx_spec.variant_*.
struct DevKnobs {
  int fsm_fast;                // PHX_FSM_FAST (default -1 = unset; 0: off; 2: forced at any batch size)
  int fsm_lean;                // PHX_FSM_LEAN (default 1)
  int fsm_wide;                // PHX_FSM_WIDE (default 1)
  int fsm_batch;               // PHX_FSM_BATCH (default 1): the lean FSM rollout loop stores four steps per batch as 16-byte pieces
  int generic_nt;              // PHX_GENERIC_NT (default 0)
  int generic_remap;           // PHX_GENERIC_REMAP (default 1)
  int generic_tablds;          // PHX_GENERIC_TABLDS (default 1)
  int autotune;                // PHX_AUTOTUNE (default 1): PHX_VR_AUTO times its two FSM candidates on a handle's first call of a shape (0: the size rule alone)
  int generic_sched;           // PHX_GENERIC_SCHED (default 1): specs with a compiled schedule run on phx_sched_step_kernel (0: the dynamic kernel everywhere)
  int rollout_epb;             // PHX_ROLLOUT_EPB (default 0)
  int rollout_fast;            // PHX_ROLLOUT_FAST (default -1 = unset; 0: off)
  int rollout_first;           // PHX_ROLLOUT_FIRST (default 0)
  int rollout_g;               // PHX_ROLLOUT_G (default 0)
  int rollout_ldskb;           // PHX_ROLLOUT_LDSKB (default 0)
  int rollout_nt;              // PHX_ROLLOUT_NT (default 0)
  int rollout_remap;           // PHX_ROLLOUT_REMAP (default -1)
  int rollout_sparse_flags;    // PHX_ROLLOUT_SPARSE_FLAGS (default 1: round 3's kernel zero-fills the flag planes of fragments >= 2^23 agent-steps and stores the non-zero words; 0 never, 2 always)
  int step_nt;                 // PHX_STEP_NT (default 0)
  int stk_rollout_nt;          // PHX_STK_ROLLOUT_NT (default 0)
  int stk_step_fast;           // PHX_STK_STEP_FAST (default 1)
  int stk_step_nt;             // PHX_STK_STEP_NT (default 0)
  int sw_generic;              // PHX_SW_GENERIC (default 0)
  int sw_persist;              // PHX_SW_PERSIST (default 1): the store-wave kernel's workgroups walk several pair groups (0: one workgroup per group)
  int sw_store_waves;          // PHX_SW_STORE_WAVES (default 0)
  int sw_tc;                   // PHX_SW_TC (default 0)
  int sw_work_waves;           // PHX_SW_WORK_WAVES (default 0)
};
const DevKnobs& phx_knobs();

// workgroup shape of the round-4 store-wave rollout kernel (phx_sc_rollout_sw.hip)
struct ScSwPlan {
  int32_t ok, epb, G, K, norm, tc, nt, n_rec, n_store, dtab_n, lds;
  int32_t specialised;           // a compile-time instantiation serves the shape (what PHX_VR_AUTO requires: the run-time-shape kernel is slower than round 3's)
};

// one rule of a device-evaluated state handler (phx_stage_rule with the field resolved)
struct DevRule { int32_t stage, field_id, col, ncols, cmp, next_stage, is_f64, pad; double threshold; };

struct DevSpec {
  int32_t A, S, B, D, n_exo, nnz;
  int32_t num_steps, round_limit, env_type;
  uint32_t flags;
  int32_t queue_cap, trace_cap;
  int32_t scan_cap;              // >= max(queue_cap, longest acting list + PHX_MAX_INJECT)
  int32_t n_lists;               // acting lists: PLAIN 1, FSM n_stages, STACKELBERG 2
  int32_t initial_stage;
  int32_t buyer_nnz;             // price slots per env = max buyer degree x number of buyers
  int32_t buyer_stride;          // buyer.prices is slot-major (ELL): slot k of buyer r at k * stride + r
  uint64_t seed;
  int64_t env_offset;
  int32_t kind_count[PHX_KIND_COUNT];
  // static tables (device memory)
  const uint8_t* kind;           // [A]
  const int32_t* param_i;        // [A][PHX_NPI]
  const double*  param_f;        // [A][PHX_NPF]
  const int32_t* row_ptr;        // [A+1]
  const int32_t* col;            // [nnz]
  const int32_t* strat_rank;     // [A]  -1 for non-strategic
  const int32_t* strat_idx;      // [S]
  const int32_t* kind_rank;      // [A]
  const int32_t* exo_rank;       // [A]  first exo column of a CUSTOMER / PUBLISHER, else -1
  const int32_t* buyer_off;      // [A]  rank of a BUYER among the buyers (its column in buyer.prices)
  const int32_t* act_ptr;        // [n_lists+1] ordered acting lists
  const int32_t* act_idx;
  const uint8_t* act_mask;       // [n_lists][A] 1 <=> agent is in the acting list
  const uint8_t* obs_mask;       // [n_lists][A] who observes after a step in this list/stage
  const uint8_t* rew_mask;       // [n_lists][A] who is rewarded
  const int32_t* stage_next;     // [n_lists]  (FSM)
  const uint8_t* stage_allowed;  // [n_lists][n_lists] FSMStage.next_stages as a matrix (handler-chosen transitions), or NULL
  const int32_t* stage_tab;      // [n_lists][num_steps + 1] tabulated clock / stage handlers (phx_spec.stage_tab), or NULL
  int32_t n_rules;               // device-evaluated state handlers (phx_spec.stage_rules), in the spec's order
  const DevRule* rules;
  const int32_t* mt_ptr;         // PHX_F_MT19937: [n_lists + 1] / the exogenous indices of a list's drawing agents in acting order
  const int32_t* mt_rank;        //   (the order in which the reference's CustomerAgents call np.random.randint in a step of that list)
  // generic engine, drop-out-free supply-chain specs: the round schedule of a step in which every agent is live and every acting
  // strategic agent has an action, simulated once at phx_create (phx_api.hip: build_static_schedule).  sched + sched_off[list]:
  // [R, n_0 .. n_7], act_off[acting items] (queue offset of the item's message or -1), then per round { cnt[A], goff[A], order[n_r],
  // next_off[n_r] } = inbox sizes, inbox offsets in first-arrival (dict) order, the queue index of every inbox position in send
  // order, and where the reply to that position goes in the next queue (-1: none) -- what the acting phase's counts and scan and
  // the per-round LDS atomics, block scans and rank sort compute.
  const int32_t* sched;          // or NULL
  const int32_t* sched_off;      // [n_lists] offset of the list's schedule in `sched`, -1: this list is not static
  const uint8_t* stage_rew_all;  // [n_lists] rewarded_agents is None (every strategic agent observes, fsm.py:315-317)
  const int32_t* reset_obs_ptr;  // agents that observe at reset: CSR with a single row
  const int32_t* reset_obs_idx;
  int32_t n_reset_obs;
  // supply-chain static schedule (fused kernels)
  const int32_t* shop_agent;     // [nS] agent index of each shop (kind-rank order)
  const int32_t* shop_norm;      // [nS] ShopAgent max_sales_per_step (param_i[shop][1])
  const int32_t* shop_cust_ptr;  // [nS+1]
  const int32_t* shop_cust_exo;  // exo rank of each customer of the shop, acting order
  const int32_t* shop_cust_agent;// agent index of each customer
  const uint8_t* shop_cust_act;  // [n_lists][n_exo] customer (by position in shop_cust_*) acts in list
  const uint8_t* sc_shop_flags;  // [n_lists][nS] 1 shop acts, 2 a customer acts, 4 every customer acts, 8 observes, 16 rewarded
  int32_t max_cust;              // max customers of one shop
  int32_t variant_rollout, variant_block, variant_step;   // phx_spec.variant_* (0 = the library's choice)
  ScFastPlan sc_fast;            // fast rollout kernel: plan (ok == 0: not applicable)
  int32_t sc_wide_K, sc_wide_norm;   // phx_sc_step_wide_kernel: the shops' common customer count (0: the kernel does not apply) and normaliser
  ScSwPlan sc_sw;                // store-wave rollout kernel (round 4): plan (ok == 0: not applicable)
  ScSwPlan fsm_sw;               // its FSM instantiation (FSM supply chains on the handler-less stage chain): plan
  const uint16_t* fsm_sw_tab;    // [2][num_steps] SWF_* word of every episode position, then its stage (phx_sc_rollout_sw.hip)
  const void* sc_sw_tables;      // its table image in device memory (phx_sc_sw_tables)
  const int32_t* sc_sw_exo_first;// [S] exogenous column of each shop's first customer, or NULL: the customers' columns are not consecutive (exo replays go to round 1's kernel)
  int32_t* sc_sw_guard;          // device word: the replay pre-scan stores the call's number here when an action rounds below zero
  int32_t fsm_lean_K, fsm_lean_norm;   // lean FSM rollout (phx_sc_fused.hip): every shop's customer count (0: not applicable) / normaliser
  int32_t sc_all_or_none;              // every (acting list, shop): the shop's customers act all or none (sc_shop_flags: bit 2 implies bit 4)
  ScFastPlan fsm_fast;           // time-parallel FSM rollout (phx_sc_rollout_fsm.hip): block shape (ok == 0: not applicable)
  const uint32_t* fsm_pos_tab;   // [num_steps] flags / lookbacks / stage of every episode position (layout: phx_sc_rollout_fsm.hip)
  int32_t* fsm_irregular;        // device word the launch uses to send envs off the tabulated stage chain to the general loop
  void* fsm_gen_host;            // std::atomic<int32_t>* in the env's handle: launch generation written into fsm_irregular (per env, thread safe)
  // host-built lookup tables of the rollout kernel (exactly the values the formulas give):
  //   [0,101) f32 stock/100 ; [101, 101+n_tabn) f32 x/norm, n_quot valid entries (0 unless
  //   every shop has the same norm) ; then 101 f64 penalties 0.1*stock (8-byte aligned)
  const float* sc_tab;
  int32_t n_tabn, n_quot, rew_smax;
  // Stackelberg market static schedule (fused kernel): neighbour table of the buyers, slot-major
  const uint16_t* stk_nbr;       // [buyer_dmax][nBuyers] seller rank of neighbour k, 0xFFFF = none
  const int32_t* stk_nbr_conn;   // same shape: base connection of that slot (StochasticNetwork)
  const uint32_t* stk_rec;       // [A] kind | deg << 8 | kind_rank << 16
  const uint8_t*  stk_flags;     // [2][A] 1 acts, 2 observes, 4 rewarded in list 0 (leaders' step) / 1
  // the same per agent in two words a lane loads without a dependent lookup (static graphs, buyers with <= 8 neighbours):
  const uint32_t* stk_rec2;      // [A] seller | flags(list 0) << 1 | flags(list 1) << 4 | deg << 8 | kind_rank << 16
  const uint32_t* stk_agent;     // [A][4] buyer: its neighbours' seller ranks, packed u16; seller: [0] = its degree
  int32_t stk_packed;            // the two tables are valid
  // StochasticNetwork (network.py:340-453): per-env on/off byte per base connection
  int32_t n_conn;
  const double*  conn_rate;      // [n_conn]
  const int32_t* col_conn;       // [nnz] base connection of each CSR entry
  // Supertypes / Samplers (supertype.py:16-30, utils/samplers.py, env.py:211-216)
  int32_t n_samplers;            // columns of env.sampler
  int32_t any_typed;             // some shop consumes a type field (obs dim 4, weighted penalty)
  int32_t device_sampling;       // every sampler is PHX_SAMPLER_UNIFORM (rollouts can auto-reset)
  const int32_t* sampler_kind;   // [n_samplers]
  const double*  sampler_param;  // [n_samplers][4] low, high, clip_low, clip_high (NaN = None)
  const int32_t* type_src;       // [A] sampler column, PHX_TYPE_CONST or PHX_TYPE_NONE
  const int32_t* shop_type_src;  // [nS] the same per shop (kind-rank order)
  const double*  shop_type_prm;  // [nS][2] constant weight, obs normaliser (param_f of the shop)
  // the generic engine's LDS-staged topology tables, packed in the kernel's LDS layout (one flat copy)
  const char* tab_blob;
  int32_t tab_bytes;
  int32_t n_adx;                 // AdExchangeAgents (their batches are reduced by the whole workgroup)
  const int32_t* adx_idx;        // [n_adx] agent indices
  const int32_t* adx_nbr_ptr;    // [n_adx+1] the CSR entries of each exchange's AdvertiserAgent neighbours
  const int32_t* adx_nbr_e;      //           (= self.advertiser_ids, in order)
  int32_t dynamic_graph;         // StochasticNetwork with some rate < 1
  int32_t ads_pub, ads_adx, ads_pub_stage;   // static digital-ads schedule (phx_ads_fused.hip)
  // state blob field pointers
  void* f[F_COUNT];
  int64_t ws_stride;             // workspace bytes per env
  // the message-passing engine with a COMPILED schedule (phx_generic_sched.hip): several env instances per wave
  int32_t gs_ok, gs_L, gs_qstride, gs_words;   // applies / lanes per env instance / words per queue / words of gs_blob
  const int32_t* gs_blob;        // [n_lists] word offset of each list's program, then the programs (layout: phx_sched_compile)
  const int32_t* gs_rec;         // (sender | receiver << 16, type | round << 16) of every message of every list's step, in log order
  int32_t* gs_dyn_flag;          // [B] 0 between launches; 2 | 1: the env's step is outside the schedule's premise (a done agent, an acting shop
                                 // without an action) -- published by the env's schedule workgroup, awaited and zeroed by the launch's tail workgroups
  int32_t lean_lds;              // generic engine: the dynamic steps' sort / scan scratch is in the workspace, not in LDS (LEAN)
  const DevSpec* self_dev;       // this struct in device memory (kernels that read it through the scalar cache instead of 300 SGPRs)
};

// Per-device "done once" flags of a launcher (function attributes are per device): one bit per device id, updated atomically -- a process
// may drive several GPUs from several host threads; setting an attribute twice is harmless, skipping it on a device is a failed launch.
struct PhxPerDeviceOnce {
  std::atomic<unsigned long long> bits{0ull};
  bool done(int dev) const { return dev >= 0 && dev < 64 && ((bits.load(std::memory_order_acquire) >> dev) & 1ull); }
  void mark(int dev) { if (dev >= 0 && dev < 64) bits.fetch_or(1ull << dev, std::memory_order_release); }
};
// compute units of the calling thread's current device (cached per device id)
inline int phx_device_cu_count() {
  static std::atomic<int> cu[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cu[dev].load(std::memory_order_relaxed);
  if (n <= 0) { hipDeviceProp_t pr; n = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; cu[dev].store(n, std::memory_order_relaxed); }
  return n;
}

// names of the kernels the calling thread's last phx_step / phx_rollout / phx_resolve launched (phx_last_kernel, tests)
void phx_note_kernel(const char* name);

struct GenArgs {               // arguments of the generic engine kernel
  phx_step_io io;
  const DevMsg* inject;         // device copy of host-injected messages (same for every env)
  int32_t n_inject;
  int32_t resolve_only;         // Network.resolve() alone: no clock tick, no acting, no epilogue
  int32_t phase;                // 0: a whole step; 1: phx_step_begin (acting + resolve_network, no epilogue: a host-side stage handler
                                // reads the resolved agent state next, fsm.py:294-302); 2: phx_step_end (transition + epilogue only)
  unsigned long long* timing;   // PHX_TIMING builds only
  int32_t tab_off;              // byte offset of the LDS-staged topology tables (generic engine)
  int32_t xcd_remap;            // XCD-aware workgroup -> env mapping (xcd_block)
  // launch-loop rollout (phx_rollout on the generic engine): the policy, the trajectory row of step roll_t and
  // the caller's reset at an episode end are fused into the step kernel (roll_t < 0: a plain phx_step)
  int32_t roll_t;
  int32_t roll_T;               // > 0: the kernel itself loops over steps roll_t .. roll_t + roll_T - 1 (queues, tables and the env's
                                // workgroup stay resident; io.exo / io.msg_log / io.msg_count are then [T][B][..] bases); 0: one step
  int32_t gs_reserved0, gs_reserved1;
  const float* roll_actions_in; // [T][B][S] replayed policy or NULL -> random policy
  float* roll_actions;          // [B][S] scratch the acting phase reads (= io.actions)
  phx_rollout_io roll;
};

template <typename T>
__device__ __forceinline__ T* fld(const DevSpec& sp, int id) { return (T*)sp.f[id]; }

// XCD-aware workgroup index: consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each); mapping id
// to start(id % 8) + id / 8 gives every XCD a contiguous range of envs, so that the partial cache lines where
// two neighbouring workgroups' output segments meet are merged in one L2 (measured history)
__device__ __forceinline__ int xcd_block(bool on) {
  if (!on) return (int)blockIdx.x;
  const unsigned n = gridDim.x, x = blockIdx.x & 7u, q = n >> 3, rem = n & 7u;    // XCD x runs ids x, x + 8, ...
  return (int)(x * q + (x < rem ? x : rem) + (blockIdx.x >> 3));
}
// the same with locality groups of gs workgroups (gs consecutive envs ranges per XCD, groups round-robin);
// gs must divide gridDim.x / 8
__device__ __forceinline__ int xcd_block_grouped(int gs) {
  if (gs <= 1 || (gridDim.x % (8u * (unsigned)gs)) != 0u) return (int)blockIdx.x;
  const unsigned x = blockIdx.x & 7u, j = blockIdx.x >> 3;
  return (int)((j / (unsigned)gs) * (8u * (unsigned)gs) + x * (unsigned)gs + (j % (unsigned)gs));
}

// The static topology tables the message handlers walk in dependent chains.  The generic engine
// stages them in LDS (every env instance of a launch reads the same few cache lines otherwise);
// other kernels use the global copies through topo_global().
struct Topo {
  const uint8_t* kind;
  const int32_t* param_i;
  const double*  param_f;
  const int32_t* row_ptr;
  const int32_t* col;
  const int32_t* strat_rank;
  const int32_t* kind_rank;
  const int32_t* exo_rank;
  const int32_t* buyer_off;
  const int32_t* col_conn;       // StochasticNetwork: base connection of each CSR entry, else NULL
  const uint8_t* conn_on;        // this env's row of net.conn_on (set by the kernel), else NULL
  int kmax;                      // no agent kind above this one (a compile-time constant in the specialised kernels)
};
// kind of agent a.  With kmax a constant (phx_generic_step_kernel<.., KMAX>) the bound prunes every switch over kinds
// to the cases that can occur: a supply-chain-only spec compiles none of the market / ads handlers.
__device__ __forceinline__ int tkind(const Topo& tp, int a) {
  const int k = tp.kind[a];
  __builtin_assume(k <= tp.kmax);
  return k;
}
// CSR entry k is an edge of this env's graph (always, for a static Network)
__device__ __forceinline__ bool edge_on(const Topo& tp, int k) {
  return tp.conn_on == nullptr || tp.conn_on[tp.col_conn[k]] != 0;
}
__device__ __forceinline__ Topo topo_global(const DevSpec& sp) {
  Topo t;
  t.col_conn = sp.col_conn; t.conn_on = nullptr; t.kmax = PHX_KIND_COUNT - 1;
  t.kind = sp.kind; t.param_i = sp.param_i; t.param_f = sp.param_f; t.row_ptr = sp.row_ptr; t.col = sp.col;
  t.strat_rank = sp.strat_rank; t.kind_rank = sp.kind_rank; t.exo_rank = sp.exo_rank; t.buyer_off = sp.buyer_off;
  return t;
}

// ---- payload whitelists (message.py:20-42): 0 = any -----------------------------------------
__device__ __forceinline__ void dev_payload_types(int type, int& sk, int& rk, int& decorated) {
  decorated = 1; sk = 0; rk = 0;
  switch (type) {
    case PHX_MSG_ORDER_REQUEST:  sk = PHX_KIND_CUSTOMER; rk = PHX_KIND_SHOP; break;
    case PHX_MSG_ORDER_RESPONSE: sk = PHX_KIND_SHOP; rk = PHX_KIND_CUSTOMER; break;
    case PHX_MSG_STOCK_REQUEST:  sk = PHX_KIND_SHOP; rk = PHX_KIND_FACTORY; break;
    case PHX_MSG_STOCK_RESPONSE: sk = PHX_KIND_FACTORY; rk = PHX_KIND_SHOP; break;
    case PHX_MSG_PRICE:          sk = PHX_KIND_SELLER; rk = PHX_KIND_BUYER; break;
    case PHX_MSG_ORDER:          sk = PHX_KIND_BUYER; rk = PHX_KIND_SELLER; break;
    case PHX_MSG_PING:           decorated = 0; break;
    default: break;
  }
}

__device__ __forceinline__ int dev_nbr_slot(const DevSpec& sp, const Topo& tp, int u, int v) {
  const int lo = tp.row_ptr[u], hi = tp.row_ptr[u + 1];
  for (int k = lo; k < hi; ++k)
    if (tp.col[k] == v) return k - lo;
  return -1;
}
__device__ __forceinline__ bool dev_has_edge(const DevSpec& sp, const Topo& tp, int u, int v) {   // network.py:224-231
  // connections are undirected (add_connection adds u->v and v->u with one rate, network.py:122-123,
  // 383-391; phx_create checks the CSR is symmetric): search the shorter of the two adjacency rows
  if (tp.row_ptr[v + 1] - tp.row_ptr[v] < tp.row_ptr[u + 1] - tp.row_ptr[u]) { const int w = u; u = v; v = w; }
  const int lo = tp.row_ptr[u], hi = tp.row_ptr[u + 1];
  for (int k = lo; k < hi; ++k)
    if (tp.col[k] == v && edge_on(tp, k)) return true;
  return false;
}
// topo_global + this env's connectivity row
__device__ __forceinline__ Topo topo_env(const DevSpec& sp, int b) {
  Topo t = topo_global(sp);
  if (sp.n_conn > 0) t.conn_on = fld<uint8_t>(sp, F_NET_CONN_ON) + (int64_t)b * sp.n_conn;
  return t;
}
// the payload whitelist half of Network.send's checks (network.py:297-331)
__device__ __forceinline__ int dev_payload_check(const DevSpec& sp, const Topo& tp, int src, int dst, int type) {
  if (!(sp.flags & PHX_F_NO_PAYLOAD_CHECKS)) {
    int sk, rk, dec;
    dev_payload_types(type, sk, rk, dec);
    if (!dec) return PHX_ERR_PAYLOAD;
    if (sk && tkind(tp, src) != sk) return PHX_ERR_PAYLOAD;
    if (rk && tkind(tp, dst) != rk) return PHX_ERR_PAYLOAD;
  }
  return 0;
}
// the same with both kinds known (a reply: the receiver's kind, and the sender's fetched with the message)
__device__ __forceinline__ int dev_payload_check_kinds(const DevSpec& sp, int src_kind, int dst_kind, int type) {
  if (!(sp.flags & PHX_F_NO_PAYLOAD_CHECKS)) {
    int sk, rk, dec;
    dev_payload_types(type, sk, rk, dec);
    if (!dec) return PHX_ERR_PAYLOAD;
    if (sk && src_kind != sk) return PHX_ERR_PAYLOAD;
    if (rk && dst_kind != rk) return PHX_ERR_PAYLOAD;
  }
  return 0;
}
// Network.send checks (network.py:246-252, 297-331); returns PHX_ERR_* (0 = deliverable)
__device__ __forceinline__ int dev_send_check(const DevSpec& sp, const Topo& tp, int src, int dst, int type) {
  if (!(sp.flags & PHX_F_IGNORE_CONN_ERRORS) && !dev_has_edge(sp, tp, src, dst)) return PHX_ERR_NETWORK;
  return dev_payload_check(sp, tp, src, dst, type);
}

// ---- device RNG (the definition is stated in DESIGN.md; the oracle restates it) ---------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
  // one 32x32->64 product per multiplier and round (v_mad_u64_u32 gives hi and lo together)
#ifndef PHX_PHILOX_ROUNDS
#define PHX_PHILOX_ROUNDS 10
#endif
#pragma unroll
  for (int r = 0; r < PHX_PHILOX_ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    // three-way xor in ONE instruction (gfx950: v_bitop3_b32, truth table 0x96); the compiler emits two v_xor_b32 otherwise
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)      /* PHX_OFFLOAD_ARCH builds for other CDNA parts: no v_bitop3_b32 */
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
#else
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
#endif
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Device RNG (definition restated in oracle/phx_oracle.c and DESIGN.md).  One Philox block
//     ctr = (env_lo, env_hi | attempt << 16, tick >> 2, shop | g << 20), key = seed
// serves FOUR consecutive ticks of customer group g (customers 6g .. 6g+5) of a shop: tick t owns
// word t & 3.  One 32-bit word u yields six exactly uniform order sizes AND (group 0) the shop's
// random-policy action: with m = u * 5^6, the words with low32(m) < 2^32 mod 5^6 = 14171 are
// rejected (Lemire; redrawn at the same position with attempt + 1, probability 3.3e-6); every
// y = m >> 32 in [0, 5^6) is then hit by exactly 274877 consecutive words, so y and the word's
// rank j = (low32(m) - 14171) / 5^6 in [0, 274877) are independent and exactly uniform.  Customer
// i of the group orders base-5 digit i of y; the action is j * (100 / 274877) in [0, 100).
#define PHX_RNG_P6   15625u
#define PHX_RNG_REJ  14171u
#define PHX_RNG_NJ   274877u
__device__ __forceinline__ void rng_block(uint64_t seed, int64_t genv, uint32_t tick, int shop, int g,
                                          uint32_t attempt, uint32_t w[4]) {
  philox4x32_10((uint32_t)genv, (uint32_t)((uint64_t)genv >> 32) | (attempt << 16), tick >> 2,
                (uint32_t)shop | ((uint32_t)g << 20), (uint32_t)seed, (uint32_t)(seed >> 32), w);
}
// BatchResolver(shuffle_batches=True) on the device stream: block `blk` of the Fisher-Yates draws of
// (env, tick, round, receiver); draw d uses word d & 3 of block d >> 2, j = mulhi(word, i + 1)
// (definition restated in oracle/phx_oracle.c: shuffle_draw)
__device__ __forceinline__ void rng_shuffle_block(uint64_t seed, int64_t genv, uint32_t tick, int round, int receiver,
                                                  uint32_t blk, uint32_t w[4]) {
  philox4x32_10((uint32_t)genv, (uint32_t)((uint64_t)genv >> 32) | (blk << 16), tick,
                0x10000000u | (((uint32_t)round & 0xfffu) << 16) | (uint32_t)receiver, (uint32_t)seed, (uint32_t)(seed >> 32), w);
}
// u -> (y, j), false when the word is rejected.  (x / 5^6 == umulhi(x, 2251799814) >> 13 for every
// 32-bit x: ceil(2^45 / 5^6) with error 4918 <= 2^13.)
__device__ __forceinline__ bool rng_split(uint32_t u, uint32_t& y, uint32_t& j) {
  const uint64_t m = (uint64_t)u * PHX_RNG_P6;
  const uint32_t l = (uint32_t)m;
  y = (uint32_t)(m >> 32);
  j = __umulhi(l - PHX_RNG_REJ, 2251799814u) >> 13;
  return l >= PHX_RNG_REJ;
}
__device__ __forceinline__ uint32_t rng_pick(const uint32_t w[4], uint32_t tick) {
  // two levels of selects on the tick's low bits (written as a chain of comparisons the compiler built exec-masked branches around the moves)
  const uint32_t lo = (tick & 1u) ? w[1] : w[0], hi = (tick & 1u) ? w[3] : w[2];
  return (tick & 2u) ? hi : lo;
}
// (y, j) of customer group g, starting at `attempt0` (the cold / generic path: one Philox call per try)
__device__ __forceinline__ uint32_t rng_group_y(uint64_t seed, int64_t genv, uint32_t tick, int shop, int g,
                                                uint32_t attempt0, uint32_t* jout = nullptr) {
  for (uint32_t attempt = attempt0;; ++attempt) {
    uint32_t w[4], y, j;
    rng_block(seed, genv, tick, shop, g, attempt, w);
    if (rng_split(rng_pick(w, tick), y, j)) { if (jout) *jout = j; return y; }
  }
}
__device__ __forceinline__ float rng_j_to_action(uint32_t j) { return (float)j * (100.0f / 274877.0f); }
// x / 5 for x < 2^16 through f32: x * 0.2f carries a relative error < 2^-23, far below the 0.2 gap
// to the next integer boundary, and 0.2f > 0.2 keeps exact multiples on the right side; the same
// holds for 0.04f .. 0.00032f on [0, 5^6) (all checked exhaustively in tests/test_host_logic.py).
__device__ __forceinline__ uint32_t rng_div5(uint32_t x) { return (uint32_t)((float)x * 0.2f); }
// sum of the six base-5 digits of y < 5^6:  y - 4 * (y/5 + y/25 + y/125 + y/625 + y/3125)
__device__ __forceinline__ int rng_digit_sum6(uint32_t y) {
  const float yf = (float)y;
  const uint32_t q = (uint32_t)(yf * 0.2f) + (uint32_t)(yf * 0.04f) + (uint32_t)(yf * 0.008f) +
                     (uint32_t)(yf * 0.0016f) + (uint32_t)(yf * 0.00032f);
  return (int)(y - (q << 2));
}
// sum of digits i in [0, n) selected by mask (NULL = all); n <= 6
__device__ __forceinline__ int rng_digit_sum(uint32_t y, int n, const uint8_t* actmask) {
  int sum = 0;
  for (int i = 0; i < n; ++i) {
    const uint32_t q = rng_div5(y);
    if (actmask == nullptr || actmask[i] != 0) sum += (int)(y - 5u * q);
    y = q;
  }
  return sum;
}
// one customer's draw (generic engine)
__device__ __forceinline__ int rng_customer_order(uint64_t seed, int64_t genv, uint32_t tick, int shop, int k) {
  const int g = k / 6, i = k - g * 6;
  uint32_t y = rng_group_y(seed, genv, tick, shop, g, 0);
  for (int q = 0; q < i; ++q) y = rng_div5(y);
  return (int)(y - 5u * rng_div5(y));
}
// order sum of a shop's K customers (those selected by `actmask`, NULL = all) given group 0's block of
// its tick quad (w); *act_j = the action rank of this tick.  Further groups and the rare redraw
// fetch their own blocks.
__device__ __forceinline__ int rng_orders_from_block(const uint32_t w[4], uint64_t seed, int64_t genv, uint32_t tick,
                                                     int shop, int K, const uint8_t* actmask, uint32_t* act_j) {
  uint32_t y, j;
  if (!rng_split(rng_pick(w, tick), y, j)) y = rng_group_y(seed, genv, tick, shop, 0, 1, &j);   // probability 3.3e-6
  if (act_j) *act_j = j;
  if (K <= 0) return 0;
  int sum = (actmask == nullptr && __all(K >= 6)) ? rng_digit_sum6(y) : rng_digit_sum(y, K < 6 ? K : 6, actmask);
  for (int g = 1; 6 * g < K; ++g)
    sum += rng_digit_sum(rng_group_y(seed, genv, tick, shop, g, 0), K - 6 * g < 6 ? K - 6 * g : 6,
                         actmask ? actmask + 6 * g : nullptr);
  return sum;
}
// Sum over the shop's K customers (those selected by `actmask`, NULL = all), or with kth >= 0 only
// customer kth's draw; *act_j = the shop's action rank of this tick.
__device__ __forceinline__ int rng_shop_orders(uint64_t seed, int64_t genv, uint32_t tick, int shop,
                                               int K, const uint8_t* actmask, int kth,
                                               uint32_t* act_j = nullptr) {
  if (kth >= 0) return rng_customer_order(seed, genv, tick, shop, kth);
  uint32_t w[4];
  rng_block(seed, genv, tick, shop, 0, 0, w);
  return rng_orders_from_block(w, seed, genv, tick, shop, K, actmask, act_j);
}
// all K customers, sum only (same definition)
__device__ __forceinline__ int rng_shop_order_sum(uint64_t seed, int64_t genv, uint32_t tick, int shop,
                                                  int K, uint32_t* act_j) {
  return rng_shop_orders(seed, genv, tick, shop, K, nullptr, -1, act_j);
}
// group 0's block of a shop's tick quad, kept across the four ticks by kernels that walk time in order
struct RngQuadCache { uint32_t w[4]; uint32_t q; };
__device__ __forceinline__ void rng_quad_block(RngQuadCache& c, uint64_t seed, int64_t genv, uint32_t tick, int shop) {
  if ((tick >> 2) != c.q) { rng_block(seed, genv, tick, shop, 0, 0, c.w); c.q = tick >> 2; }
}

// UniformFloatSampler column j at the env's `episode`-th reset (definition restated in
// oracle/phx_oracle.c: phxo_rng_uniform): ctr = (env_lo, env_hi, episode, 0x80000000 | j);
// u = ((w0 >> 5) * 2^26 + (w1 >> 6)) / 2^53; low + (high - low) * u rounded per operation; np.clip.
__device__ __forceinline__ double rng_uniform(uint64_t seed, int64_t genv, uint32_t episode, int j,
                                              const double* prm) {
  uint32_t w[4];
  philox4x32_10((uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), episode, 0x80000000u | (uint32_t)j,
                (uint32_t)seed, (uint32_t)(seed >> 32), w);
  const double u = __dmul_rn(__dadd_rn(__dmul_rn((double)(w[0] >> 5), 67108864.0), (double)(w[1] >> 6)),
                             1.0 / 9007199254740992.0);
  double v = __dadd_rn(prm[0], __dmul_rn(__dsub_rn(prm[1], prm[0]), u));
  if (prm[2] == prm[2] && v < prm[2]) v = prm[2];
  if (prm[3] == prm[3] && v > prm[3]) v = prm[3];
  return v;
}
// `np.random.random() < rate` for base connection i at the env's `episode`-th reset (definition
// restated in oracle/phx_oracle.c: phxo_rng_connection)
__device__ __forceinline__ bool rng_connection(uint64_t seed, int64_t genv, uint32_t episode, int i, double rate) {
  uint32_t w[4];
  philox4x32_10((uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), episode, 0x40000000u | (uint32_t)(i >> 1),
                (uint32_t)seed, (uint32_t)(seed >> 32), w);
  const uint32_t wa = (i & 1) ? w[2] : w[0], wb = (i & 1) ? w[3] : w[1];
  const double u = __dmul_rn(__dadd_rn(__dmul_rn((double)(wa >> 5), 67108864.0), (double)(wb >> 6)),
                             1.0 / 9007199254740992.0);
  return u < rate;
}

// env.reset(): `for sampler in self._samplers: sampler.sample()` (env.py:211-212) for column j
__device__ __forceinline__ double dev_sample_column(const DevSpec& sp, int b, int j, uint32_t episode,
                                                    const double* values, double current) {
  if (values) return values[(int64_t)b * sp.n_samplers + j];
  if (sp.sampler_kind[j] == PHX_SAMPLER_UNIFORM)
    return rng_uniform(sp.seed, sp.env_offset + b, episode, j, sp.sampler_param + 4 * j);
  return current;
}
// agent.type.<field> of a shop: the managed Sampler's current value or the constant
__device__ __forceinline__ double shop_type_value(const DevSpec& sp, int b, int s) {
  const int src = sp.shop_type_src[s];
  return src >= 0 ? fld<double>(sp, F_ENV_SAMPLER)[(int64_t)b * sp.n_samplers + src] : sp.shop_type_prm[2 * s];
}

// int(round(np.float32(a))) -- round-half-to-even, supply_chain.py:139.  rintf of an f32 value is
// the same integer the reference's float64 round() produces (every f32 is exactly representable
// in f64 and the rounded integer is representable in f32); v_rndne_f32 is full rate.
__device__ __forceinline__ int dev_round_half_even(float a) {
  float r = rintf(a);
  r = r > 1073741824.0f ? 1073741824.0f : r;
  r = r < -1073741824.0f ? -1073741824.0f : r;
  return (int)r;
}

// ShopAgent.compute_reward: sales - 0.1 * stock in f64, product and difference rounded
// separately (supply_chain.py:147); the library is compiled with -ffp-contract=off.
__device__ __forceinline__ double shop_reward(int sales, int stock) {
  const double penalty = __dmul_rn(0.1, (double)stock);
  return __dsub_rn((double)sales, penalty);
}
// tutorial 2's shop: sales - type.excess_stock_weight * stock  (docs/user/tutorial2.rst:270-273)
__device__ __forceinline__ double shop_reward_w(int sales, int stock, double w) {
  return __dsub_rn((double)sales, __dmul_rn(w, (double)stock));
}

// ShopAgent.encode_observation (supply_chain.py:124-134): python-float quotient cast to f32
__device__ __forceinline__ void shop_obs(int stock, int sales, int missed, int norm, float* o) {
  const double n = (double)norm;
  o[0] = (float)((double)stock / (double)PHX_SHOP_MAX_STOCK);
  o[1] = (float)((double)sales / n);
  o[2] = (float)((double)missed / n);
}

// Same values from f32 IEEE division: for integers |x| < 2^24 the f32 quotient equals the f64
// quotient rounded to f32 (p_f64 = 53 >= 2 * 24 + 2, so the double rounding is innocuous;
// checked exhaustively over the reachable range in tests/test_host_logic.py).
__device__ __forceinline__ void shop_obs_f32(int stock, int sales, int missed, float norm, float* o) {
  o[0] = (float)stock / (float)PHX_SHOP_MAX_STOCK;
  o[1] = (float)sales / norm;
  o[2] = (float)missed / norm;
}

// x / n for a divisor that does not change over a loop: r = 1.0f / n once (an IEEE division, correctly rounded), then per quotient
// q0 = x r, e = fmaf(-n, q0, x) (exact), q = fmaf(e, r, q0) -- Markstein's correction: three instructions instead of the ~11 of v_div_scale /
// v_rcp / refinement / v_div_fmas / v_div_fixup, and the IEEE quotient bit for bit for every integer 0 <= x < 32768, 1 <= n <= 4096
// (DIV_RECIP_X / DIV_RECIP_N: checked exhaustively with the host's fmaf, oracle phxo_check_recip_div, tests/test_host_logic.py).
#define DIV_RECIP_X 32768
#define DIV_RECIP_N 4096
__device__ __forceinline__ float div_by_recip(float x, float n, float r) {
  const float q0 = x * r;
  return __fmaf_rn(__fmaf_rn(-n, q0, x), r, q0);
}

// ---- numpy scalar promotion of the ads market's floats (NEP 50; see the oracle's restatement) ----
struct TVal { double v; int tag; };
__device__ __forceinline__ TVal tv(double v, int tag) { TVal t; t.v = v; t.tag = tag; return t; }
__device__ __forceinline__ int t_tag(const TVal& a, const TVal& b) { return a.tag > b.tag ? a.tag : b.tag; }
__device__ __forceinline__ TVal t_mul(const TVal& a, const TVal& b) {
  const int tag = t_tag(a, b);
  return tag == PHX_TAG_F32 ? tv((double)__fmul_rn((float)a.v, (float)b.v), tag) : tv(__dmul_rn(a.v, b.v), tag);
}
__device__ __forceinline__ TVal t_sub(const TVal& a, const TVal& b) {
  const int tag = t_tag(a, b);
  return tag == PHX_TAG_F32 ? tv((double)__fsub_rn((float)a.v, (float)b.v), tag) : tv(__dsub_rn(a.v, b.v), tag);
}
__device__ __forceinline__ TVal t_div(const TVal& a, const TVal& b) {
  const int tag = t_tag(a, b);
  return tag == PHX_TAG_F32 ? tv((double)__fdiv_rn((float)a.v, (float)b.v), tag) : tv(__ddiv_rn(a.v, b.v), tag);
}
__device__ __forceinline__ bool t_lt(const TVal& a, const TVal& b) {
  return t_tag(a, b) == PHX_TAG_F32 ? (float)a.v < (float)b.v : a.v < b.v;
}
// PublisherAgent draws when exo == NULL: k = 0 user id in {1, 2}; k >= 1 the k-th click of the step
__device__ __forceinline__ int rng_publisher(uint64_t seed, int64_t genv, uint32_t tick, int agent, int k, double p) {
  uint32_t w[4];
  philox4x32_10((uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), tick, 0x20000000u | ((uint32_t)k << 16) | (uint32_t)agent,
                (uint32_t)seed, (uint32_t)(seed >> 32), w);
  if (k == 0) return 1 + (int)(w[0] & 1u);
  return (double)(w[0] >> 8) * (1.0 / 16777216.0) < p ? 1 : 0;
}

// ---- generic per-kind behaviour used by the generic engine ---------------------------------
struct AgentRef {          // where one agent's state lives for env b
  int a, kind, kr;         // agent index, kind, rank within kind
  int64_t base;            // b * kind_count[kind] + kr
};

__device__ __forceinline__ AgentRef agent_ref(const DevSpec& sp, const Topo& tp, int b, int a) {
  AgentRef r;
  r.a = a; r.kind = tkind(tp, a); r.kr = tp.kind_rank[a];
  r.base = (int64_t)b * sp.kind_count[r.kind] + r.kr;
  return r;
}
// AdvertiserAgent: self.type.budget with its numpy kind (pi2: strong np.float64 / python float)
__device__ __forceinline__ TVal adv_budget(const DevSpec& sp, const Topo& tp, int b, int a) {
  const int src = sp.type_src[a];
  const double v = src >= 0 ? fld<double>(sp, F_ENV_SAMPLER)[(int64_t)b * sp.n_samplers + src] : sp.param_f[a * PHX_NPF];
  return tv(v, tp.param_i[a * PHX_NPI + 2] ? PHX_TAG_F64 : PHX_TAG_PYF);
}

// Agent.reset and subclasses (agents.py:160-175; supply_chain.py:149-150; test_network.py:23-24)
__device__ __forceinline__ void dev_agent_reset(const DevSpec& sp, const Topo& tp, int b, int a) {
  const AgentRef r = agent_ref(sp, tp, b, a);
  switch (r.kind) {
    case PHX_KIND_SHOP: fld<int32_t>(sp, F_SHOP_STOCK)[r.base] = 0; break;
    case PHX_KIND_CASHBOX: fld<double>(sp, F_CASHBOX_TOTAL)[r.base] = 0.0; break;
    case PHX_KIND_SELLER:
      fld<double>(sp, F_SELLER_PRICE)[r.base] = 0.0;
      fld<double>(sp, F_SELLER_REVENUE)[r.base] = 0.0;
      fld<int32_t>(sp, F_SELLER_TX)[r.base] = 0;
      fld<double>(sp, F_SELLER_POSTED)[r.base] = 1.0;      // what every neighbour's price slot holds after reset
      break;
    case PHX_KIND_BUYER: {
      const int deg = tp.row_ptr[a + 1] - tp.row_ptr[a];
      double* pr = fld<double>(sp, F_BUYER_PRICES) + (int64_t)b * sp.buyer_nnz + tp.buyer_off[a];
      for (int k = 0; k < deg; ++k) pr[(int64_t)k * sp.buyer_stride] = 1.0;
      fld<double>(sp, F_BUYER_PAID)[r.base] = 0.0;
      fld<int32_t>(sp, F_BUYER_BOUGHT)[r.base] = 0;
      break;
    }
    case PHX_KIND_ADVERTISER: {                                // digital_ads_market.py:353-374
      const TVal budget = adv_budget(sp, tp, b, a);
      fld<double>(sp, F_ADV_LEFT)[r.base] = budget.v; fld<int32_t>(sp, F_ADV_LEFT_TAG)[r.base] = budget.tag;
      fld<double>(sp, F_ADV_BID)[r.base] = 0.0; fld<int32_t>(sp, F_ADV_BID_TAG)[r.base] = PHX_TAG_PYF;
      fld<int32_t>(sp, F_ADV_CLICKS)[r.base] = 0; fld<int32_t>(sp, F_ADV_WINS)[r.base] = 0;
      fld<int32_t>(sp, F_ADV_USER)[r.base] = 0;
      for (int u = 0; u < 3; ++u) {
        fld<int32_t>(sp, F_ADV_TOT_CLICKS)[r.base * 3 + u] = 0;
        fld<int32_t>(sp, F_ADV_TOT_REQUESTS)[r.base * 3 + u] = 0;
        fld<int32_t>(sp, F_ADV_TOT_WINS)[r.base * 3 + u] = 0;
      }
      break;
    }
    default: break;
  }
}

// encode_observation of a strategic agent; `step` is ctx.env_view.current_step.  false where the
// reference returns None (the agent is then left out of the observations, env.py:279-280)
__device__ __forceinline__ bool dev_encode_obs(const DevSpec& sp, const Topo& tp, int b, int a, int step, float* o) {
  const AgentRef r = agent_ref(sp, tp, b, a);
  switch (r.kind) {
    case PHX_KIND_SHOP: {
      const int st = fld<int32_t>(sp, F_SHOP_STOCK)[r.base], sl = fld<int32_t>(sp, F_SHOP_SALES)[r.base],
                ms = fld<int32_t>(sp, F_SHOP_MISSED)[r.base], nm = tp.param_i[a * PHX_NPI + 1];
      // f32 division == the f64 quotient cast to f32 while the integers are exact in f32 (shop_obs_f32)
      if ((((unsigned)st + (1u << 24)) | ((unsigned)sl + (1u << 24)) | ((unsigned)ms + (1u << 24)) | ((unsigned)nm + (1u << 24))) < (2u << 24))
        shop_obs_f32(st, sl, ms, (float)nm, o);
      else shop_obs(st, sl, ms, nm, o);
      if (sp.any_typed && sp.shop_type_src[r.kr] != PHX_TYPE_NONE)          // tutorial2.rst:283-294
        o[3] = (float)(shop_type_value(sp, b, r.kr) / sp.shop_type_prm[2 * r.kr + 1]);
      break;
    }
    case PHX_KIND_SELLER: {
      int deg = 0;                                             // len(ctx.neighbour_ids)
      for (int k = tp.row_ptr[a]; k < tp.row_ptr[a + 1]; ++k) deg += edge_on(tp, k) ? 1 : 0;
      o[0] = deg ? (float)((double)fld<int32_t>(sp, F_SELLER_TX)[r.base] / (double)deg) : 0.f;
      o[1] = (float)fld<double>(sp, F_SELLER_PRICE)[r.base];
      break;
    }
    case PHX_KIND_BUYER: {
      const int lo = tp.row_ptr[a], deg = tp.row_ptr[a + 1] - lo;
      const double* pr = fld<double>(sp, F_BUYER_PRICES) + (int64_t)b * sp.buyer_nnz + tp.buyer_off[a];
      double mn = 1.0; bool any = false;                       // min(prices.values(), default=1.0)
      for (int k = 0; k < deg; ++k)
        if (edge_on(tp, lo + k)) { const double v = pr[(int64_t)k * sp.buyer_stride]; if (!any || v < mn) { mn = v; any = true; } }
      o[0] = (float)mn;
      o[1] = (float)tp.param_f[a * PHX_NPF];
      break;
    }
    case PHX_KIND_MOCK_STRAT:                                  // tests/__init__.py:49-51
      fld<int32_t>(sp, F_MOCK_ENC)[r.base] += 1;
      o[0] = (float)((double)step / (double)sp.num_steps);
      break;
    case PHX_KIND_ADVERTISER: {                                // digital_ads_market.py:294-316
      const int user = fld<int32_t>(sp, F_ADV_USER)[r.base];
      if (user == 0) return false;
      const TVal budget = adv_budget(sp, tp, b, a);
      o[0] = (float)budget.v;
      o[1] = (float)t_div(tv(fld<double>(sp, F_ADV_LEFT)[r.base], fld<int32_t>(sp, F_ADV_LEFT_TAG)[r.base]), budget).v;
      o[2] = (float)(user - 1);
      break;
    }
    default: break;
  }
  return true;
}

__device__ __forceinline__ double dev_compute_reward(const DevSpec& sp, const Topo& tp, int b, int a) {
  const AgentRef r = agent_ref(sp, tp, b, a);
  switch (r.kind) {
    case PHX_KIND_SHOP:
      if (sp.any_typed && sp.shop_type_src[r.kr] != PHX_TYPE_NONE)
        return shop_reward_w(fld<int32_t>(sp, F_SHOP_SALES)[r.base], fld<int32_t>(sp, F_SHOP_STOCK)[r.base],
                             shop_type_value(sp, b, r.kr));
      return shop_reward(fld<int32_t>(sp, F_SHOP_SALES)[r.base], fld<int32_t>(sp, F_SHOP_STOCK)[r.base]);
    case PHX_KIND_SELLER: return fld<double>(sp, F_SELLER_REVENUE)[r.base];
    case PHX_KIND_BUYER:
      if (fld<int32_t>(sp, F_BUYER_BOUGHT)[r.base])
        return __dsub_rn(tp.param_f[a * PHX_NPF], fld<double>(sp, F_BUYER_PAID)[r.base]);
      return 0.0;
    case PHX_KIND_MOCK_STRAT: fld<int32_t>(sp, F_MOCK_REW)[r.base] += 1; return 0.0;
    case PHX_KIND_ADVERTISER: return (double)fld<int32_t>(sp, F_ADV_CLICKS)[r.base];   // :335-343, risk_aversion 0
    default: return 0.0;
  }
}

__device__ __forceinline__ bool dev_is_terminated(const DevSpec& sp, const Topo& tp, int b, int a, int step) {
  if (tkind(tp, a) == PHX_KIND_MOCK_STRAT) return step == tp.param_i[a * PHX_NPI];          // tests/__init__.py:61-62
  if (tkind(tp, a) == PHX_KIND_ADVERTISER)                                                 // digital_ads_market.py:345-349
    return fld<double>(sp, F_ADV_LEFT)[(int64_t)b * sp.kind_count[PHX_KIND_ADVERTISER] + tp.kind_rank[a]] <= 0.0;
  return false;
}
__device__ __forceinline__ bool dev_is_truncated(const DevSpec& sp, const Topo& tp, int a, int step) {
  return tkind(tp, a) == PHX_KIND_MOCK_STRAT && step == tp.param_i[a * PHX_NPI];            // tests/__init__.py:64-65
}

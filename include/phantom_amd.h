/*
 * phantom_amd.h -- C ABI of the MI355X-native PhantomEnv.step() hot path.
 *
 * The reference (jpmorganchase/Phantom v2.2.0) has NO FFI: the path is a Python class
 * API.  This header is the boundary a maintainer would bind (ctypes stub shown in
 * INTEGRATION.md) to replace the bodies of
 *
 *   PhantomEnv.reset            phantom/env.py:185-237
 *   PhantomEnv.step             phantom/env.py:239-303
 *   FiniteStateMachineEnv.step  phantom/fsm.py:253-380      (reset: fsm.py:195-251)
 *   StackelbergEnv.step         phantom/stackelberg.py:111-196 (reset: :53-109)
 *   Network.send / resolve      phantom/network.py:233-265
 *   BatchResolver.resolve       phantom/resolvers.py:128-163
 *
 * for a whole batch of B independent env instances per call.  All pointers are plain
 * device pointers (hipMalloc'ed / torch tensors' data_ptr()); the library allocates only
 * its own copy of the small static spec tables.  Every call is stream-ordered and
 * asynchronous; functions return 0 or a negative PHX_E* code and never throw.
 *
 * Agent behaviour is a closed set of *kinds* with hand-written device handlers
 * (arbitrary Python handler bodies cannot be compiled); see phx_kind below.
 */
#ifndef PHANTOM_AMD_H
#define PHANTOM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHX_ABI_VERSION 10

/* ---- return codes (host-side failures) ---------------------------------------------- */
#define PHX_OK            0
#define PHX_EINVAL       -1   /* malformed spec / argument                                  */
#define PHX_EUNSUPPORTED -2   /* spec valid but outside what the device engine implements   */
#define PHX_EHIP         -3   /* HIP runtime error (text in phx_last_error)                 */
#define PHX_ECAPACITY    -4   /* LDS / queue capacity exceeded at create time               */

/* ---- per-env soft error codes written to err[B] (first error wins, sticky until reset)
 *      each maps 1:1 to the exception the reference raises inside step().                 */
#define PHX_ERR_NONE        0
#define PHX_ERR_NETWORK     1 /* NetworkError: send along a missing edge  network.py:246-249 */
#define PHX_ERR_PAYLOAD     2 /* NetworkError: payload sender/receiver whitelist :311-331    */
#define PHX_ERR_UNKNOWN_MSG 3 /* ValueError: no handler for payload type  agents.py:140-143  */
#define PHX_ERR_ROUND_LIMIT 4 /* RuntimeError: msgs left after round_limit resolvers.py:160  */
#define PHX_ERR_QUEUE_FULL  5 /* build-specific: per-round message capacity exceeded         */
#define PHX_ERR_CONTEXT     6 /* KeyError: ctx[agent_id] of a non-neighbour  context.py:36-37  */
#define PHX_ERR_FSM_TRANSITION 7 /* FSMRuntimeError: handler returned a stage outside next_stages  fsm.py:304-307 */
#define PHX_ERR_HINT        8 /* build-specific (ABI 10): a replayed input violates a PHX_RH_* hint the caller vouched for
                                 (phx_rollout_io.hints): the env's rows of that call are unspecified                  */
/* BatchResolver(round_limit=None) loops until no message is left (resolvers.py:129-131), i.e. forever
 * on a message cycle; this build stops after PHX_MAX_ROUNDS rounds with PHX_ERR_ROUND_LIMIT.        */
#define PHX_MAX_ROUNDS 4096

/* ---- agent kinds (closed set) -------------------------------------------------------- */
typedef enum phx_kind {
  PHX_KIND_NONE       = 0,
  /* examples/environments/supply_chain/supply_chain.py */
  PHX_KIND_FACTORY    = 1,  /* FactoryAgent  :36-45                                        */
  PHX_KIND_SHOP       = 2,  /* ShopAgent     :70-150   pi0=factory, pi1=max_sales_per_step;
                               with a type (type_src != PHX_TYPE_NONE) it is tutorial 2's shop,
                               docs/user/tutorial2.rst:244-307: reward = sales - w * stock,
                               obs[3] = w / pf1, w = Supertype.excess_stock_weight (pf0 if constant) */
  PHX_KIND_CUSTOMER   = 3,  /* CustomerAgent :48-67    pi0=shop                            */
  /* build-authored Stackelberg market, run on ph.StackelbergEnv     */
  PHX_KIND_SELLER     = 4,  /* leader:   price setter                                      */
  PHX_KIND_BUYER      = 5,  /* follower: best-response buyer   pf0=value                   */
  /* kinds mirroring the agents of the reference's own known-answer tests                 */
  PHX_KIND_HALVER     = 6,  /* tests/network/test_tracking.py:21-28  _TestActor            */
  PHX_KIND_CASHBOX    = 7,  /* tests/network/test_network.py:17-34   MockAgent             */
  PHX_KIND_REQRESP    = 8,  /* tests/network/test_resolver.py:24-46  _TestAgent            */
  PHX_KIND_FORWARDER  = 9,  /* tests/network/test_resolver.py:89-96  _TestAgent2 pi0=target*/
  PHX_KIND_MOCK_STRAT = 10, /* tests/__init__.py:32-65 MockStrategicAgent  pi0=num_steps   */
  PHX_KIND_MOCK_AGENT = 11, /* tests/__init__.py:25-29 MockAgent (no handlers)             */
  /* examples/environments/digital_ads_market/digital_ads_market.py (a custom
   * handle_batch reduction -- the auction -- on a StochasticNetwork FSM env)               */
  PHX_KIND_PUBLISHER  = 12, /* PublisherAgent  :140-196  pi0=exchange, pi1=click-draw slots per step;
                               pf[(user-1)*4 + theme] = P(click | user, theme)   :151-154        */
  PHX_KIND_ADVERTISER = 13, /* AdvertiserAgent :199-374  pi0=exchange, pi1=theme index (0..3),
                               pi2=1 if Supertype.budget is a strong np.float64 (a clipped
                               UniformFloatSampler, samplers.py:144-145) else 0 (python float);
                               budget = the agent's type field (sampler column or pf0)          */
  PHX_KIND_ADEXCHANGE = 14, /* AdExchangeAgent :377-516  pi0=publisher, pi1=0 first / 1 second price;
                               advertiser_ids = its ADVERTISER base neighbours in CSR order     */
  PHX_KIND_COUNT      = 15
} phx_kind;

/* ---- message payload types ------------------------------------------------------------ */
typedef enum phx_msg_type {
  PHX_MSG_NONE           = 0,
  PHX_MSG_STOCK_REQUEST  = 1, /* i32 size   ShopAgent     -> FactoryAgent  supply_chain.py:26-28 */
  PHX_MSG_STOCK_RESPONSE = 2, /* i32 size   FactoryAgent  -> ShopAgent     :31-33                */
  PHX_MSG_ORDER_REQUEST  = 3, /* i32 size   CustomerAgent -> ShopAgent     :16-18                */
  PHX_MSG_ORDER_RESPONSE = 4, /* i32 size   ShopAgent     -> CustomerAgent :21-23                */
  PHX_MSG_PRICE          = 5, /* f64 price  Seller -> Buyer                                      */
  PHX_MSG_ORDER          = 6, /* i32 vol    Buyer  -> Seller                                     */
  PHX_MSG_HALVE          = 7, /* i32 value  any->any   test_tracking.py:16-18                    */
  PHX_MSG_CASH           = 8, /* f64 cash   any->any   test_network.py:12-14                     */
  PHX_MSG_REQUEST        = 9, /* f64 cash   any->any   test_resolver.py:14-16                    */
  PHX_MSG_RESPONSE       = 10,/* f64 cash   any->any   test_resolver.py:19-21                    */
  PHX_MSG_PING           = 11,/* (bool)     undecorated payload, test_resolver.py:94             */
  /* digital_ads_market.py payloads; `aux` = phx_msg_rec.round's sibling field in the queues  */
  PHX_MSG_IMPRESSION_REQ = 12,/* i user_id (timestamp unused)  any->any  :28-53                   */
  PHX_MSG_BID            = 13,/* f bid, aux = theme | user_id<<4 | tag<<8   :56-72                */
  PHX_MSG_AUCTION_RESULT = 14,/* f cost (winning_bid unused by its handler), aux = tag<<8  :75-88 */
  PHX_MSG_ADS            = 15,/* i advertiser index, aux = theme | user_id<<4   :91-107          */
  PHX_MSG_IMPRESSION_RES = 16,/* i clicked   :110-121                                            */
  PHX_MSG_COUNT          = 17
} phx_msg_type;

/* numpy scalar kind of a float that travels through the ads market (NEP 50 promotion, numpy>=2):
 * python float (weak) < np.float32 < np.float64; a binary op yields the larger tag and is
 * computed in f32 iff that tag is F32 (python floats are cast to f32 first).                   */
#define PHX_TAG_PYF 0
#define PHX_TAG_F32 1
#define PHX_TAG_F64 2

#define PHX_NPI 4   /* int32 params per agent  */
#define PHX_NPF 8   /* double params per agent */

/* ---- Supertypes / Samplers (supertype.py:16-30, utils/samplers.py:47-271, env.py:80-124,211-216)
 * Every distinct Sampler object of the env + agent supertypes (env._samplers order) is one
 * column of the per-env state field "env.sampler"; env.reset() resamples every column once
 * (env.py:211-212), agents sharing a Sampler see the same value (supertype.py:23-24).         */
#define PHX_SAMPLER_HOST    0  /* any Sampler: the caller passes the sampled values to phx_reset */
#define PHX_SAMPLER_UNIFORM 1  /* UniformFloatSampler samplers.py:120-147: low + (high-low)*u,
                                  optional clip; drawn on the device when no values are passed  */
#define PHX_TYPE_NONE  -2      /* type_src: the agent has no device-consumed type field         */
#define PHX_TYPE_CONST -1      /* type_src: constant supertype field = param_f[a][0]            */

/* env flavours */
#define PHX_ENV_PLAIN       0  /* PhantomEnv            env.py  */
#define PHX_ENV_FSM         1  /* FiniteStateMachineEnv fsm.py  */
#define PHX_ENV_STACKELBERG 2  /* StackelbergEnv        stackelberg.py */

/* spec flags */
#define PHX_F_IGNORE_CONN_ERRORS 1u  /* Network(ignore_connection_errors=True) network.py:62 */
#define PHX_F_NO_PAYLOAD_CHECKS  2u  /* Network(enforce_msg_payload_checks=False)  :63       */
#define PHX_F_FORCE_GENERIC      4u  /* never use a fused static-schedule kernel             */
#define PHX_F_SHUFFLE_BATCHES    8u  /* BatchResolver(shuffle_batches=True): every delivered batch is permuted before
                                        handle_batch (resolvers.py:150-151).  phx_step_io.shuffle replays recorded
                                        permutations (parity with a reference run); NULL -> the device draws them
                                        (Fisher-Yates, Philox block (env | draw / 4 << 48, tick, 0x10000000 | round << 16 |
                                        receiver), j = mulhi(word, i + 1)).  Always the generic engine.  Not with
                                        PHX_F_IGNORE_CONN_ERRORS or PHX_KIND_ADEXCHANGE (PHX_EUNSUPPORTED).        */
#define PHX_F_MT19937           16u  /* ABI 7: every env instance carries its own legacy-numpy MT19937 stream in the state
                                        blob ("env.mt_state" u32 [B][624], "env.mt_pos" i32 [B]): phx_mt_seed / phx_mt_draw
                                        below.  2.5 KB per env instance.                                            */

/* ---- ABI 9: stage handlers that branch on agent STATE, as a rule table the device evaluates --------------------------
 * The reference's handler of an FSM stage (fsm.py:294-307) is a Python method: it runs after the stage's agents have acted,
 * calls self.resolve_network() and returns the next stage -- typically from a threshold test on a field of the resolved agent
 * state ("RESTOCK while the shops' stock is below 60").  A handler of that form is DECLARED as rules (the host layer checks the
 * declaration against the Python handler on random states): for the env's current stage the rules are scanned in order, the
 * first whose condition holds gives the next stage, none -> phx_spec.stage_next[stage].  Evaluated on the RESOLVED state inside
 * phx_step (between what phx_step_begin and phx_step_end do) and inside every step of phx_rollout: no host round trip.  Each
 * next_stage must be one of the stage's next_stages (stage_allowed).  Served by the message-passing engine.                 */
#define PHX_CMP_LT 0
#define PHX_CMP_LE 1
#define PHX_CMP_GT 2
#define PHX_CMP_GE 3
#define PHX_CMP_EQ 4
#define PHX_CMP_NE 5
typedef struct phx_stage_rule {
  int32_t stage;        /* the rule belongs to the handler of this stage                                              */
  int32_t agent;        /* column of `field` (rank of the agent among the agents of the field's kind), or -1: the SUM over
                           all agents of that kind                                                                      */
  int32_t cmp;          /* PHX_CMP_*: value <cmp> threshold                                                             */
  int32_t next_stage;   /* the stage the handler returns when the condition holds                                       */
  double  threshold;
  char    field[32];    /* phx_field.name of a per-agent i32 or f64 state field ("shop.stock", "seller.revenue", ...)   */
} phx_stage_rule;

/*
 * Flat description of one env class: what the Python host compiles a
 * Network + PhantomEnv construction into.  All arrays are HOST pointers, copied at create.
 */
typedef struct phx_spec {
  int32_t abi_version;          /* PHX_ABI_VERSION                                           */
  int32_t n_agents;             /* A: agents in Network insertion order (env.py:142-144)     */
  int32_t batch;                /* B: env instances owned by this handle (local shard)       */
  int32_t num_steps;            /* PhantomEnv(num_steps) env.py:64                           */
  int32_t round_limit;          /* BatchResolver(round_limit); -1 = None resolvers.py:117    */
  int32_t env_type;             /* PHX_ENV_*                                                 */
  uint32_t flags;               /* PHX_F_*                                                   */
  int32_t queue_cap;            /* max messages alive in one resolver round (per env)        */
  int32_t trace_cap;            /* message-log capacity per env per step (0 = tracing off)   */
  const uint8_t* kind;          /* [A] phx_kind                                              */
  const int32_t* param_i;       /* [A][PHX_NPI]                                              */
  const double*  param_f;       /* [A][PHX_NPF]                                              */
  const int32_t* row_ptr;       /* [A+1] CSR of directed edges u->v, neighbours in           */
  const int32_t* col;           /* [nnz]  nx adjacency (insertion) order network.py:122-123;
                                   connections are undirected: u->v and v->u both present   */
  /* FSM (fsm.py:26-63): ordered acting lists, rewarded masks, handler-less next stage      */
  int32_t n_stages;
  int32_t initial_stage;
  const int32_t* stage_act_ptr; /* [n_stages+1]                                              */
  const int32_t* stage_act_idx; /* agent indices, in FSMStage.acting_agents order            */
  const uint8_t* stage_rewarded;/* [n_stages][A]; ignored where stage_rewarded_all           */
  const uint8_t* stage_rewarded_all; /* [n_stages] 1 <=> rewarded_agents is None fsm.py:315  */
  const int32_t* stage_next;    /* [n_stages] next_stages[0]  fsm.py:292                     */
  /* Stackelberg (stackelberg.py:30-51): ordered leader / follower lists                    */
  int32_t n_leaders, n_followers;
  const int32_t* leaders;
  const int32_t* followers;
  /* device RNG (used when exo == NULL): Philox4x32-10, key = (seed, env_offset + b)        */
  uint64_t seed;
  int64_t  env_offset;          /* global index of local env 0 (multi-GPU sharding)          */
  /* Supertypes / Samplers (ABI 2); n_samplers == 0 and type_src == NULL when unused          */
  int32_t n_samplers;
  const int32_t* sampler_kind;  /* [n_samplers] PHX_SAMPLER_*                                */
  const double*  sampler_param; /* [n_samplers][4] low, high, clip_low, clip_high (NaN = None) */
  const int32_t* type_src;      /* [A] or NULL: sampler column feeding the agent's type field,
                                   PHX_TYPE_CONST or PHX_TYPE_NONE                           */
  /* StochasticNetwork (network.py:340-453): row_ptr/col then describe the BASE connections
   * (network.py:383-394, neighbours in base-connection order); every reset keeps connection i
   * of each env with probability conn_rate[i] (resample_connectivity, network.py:438-452); the
   * surviving subset is the per-env state field "net.conn_on".  n_conn == 0: static Network.  */
  int32_t n_conn;
  const double*  conn_rate;     /* [n_conn]                                                  */
  const int32_t* col_conn;      /* [nnz] base connection of each CSR entry (both directions)  */
  /* ABI 5: FSM stages with handlers: stage_allowed[s][n] != 0 <=> n in FSMStage(s).next_stages (fsm.py:304);
   * NULL: only stage_next[s] is allowed (handler-less stages, fsm.py:281-292)                              */
  const uint8_t* stage_allowed; /* [n_stages][n_stages] or NULL                               */
  /* ABI 6: kernel-variant selection, per env (0 = the library's choice).  Every variant computes the same results
   * bit for bit; the fields exist so that each kernel can be selected -- and tested -- without process-wide
   * environment switches.  A variant whose preconditions the env does not meet is ignored (the library's choice).  */
  int32_t variant_rollout;      /* PHX_VR_*                                                   */
  int32_t variant_block;        /* time-parallel rollout kernels: (env, shop) pairs per workgroup; 0 = auto,
                                   PHX_VB_WHOLE_ENVS = whole envs per workgroup                */
  int32_t variant_step;         /* PHX_VS_*                                                   */
  int32_t variant_reserved;     /* 0 (ABI 7-8: variant_flags, removed in ABI 9 -- how a kernel writes its flag planes is its own business) */
  /* ABI 6: stage handlers that decide from the clock and the current stage alone (fsm.py:294-307), tabulated by the
   * host at spec-compile time: stage_tab[s * (num_steps + 1) + t] = the stage the handler of stage s returns when the
   * clock reads t (1 .. num_steps; the clock is incremented before the handler runs, fsm.py:268); rows of handler-less
   * stages hold stage_next[s].  NULL: no tabulated handler.  With a table the device takes every transition itself --
   * phx_step (phx_step_io.next_stage == NULL) and phx_rollout alike; every entry must be allowed by stage_allowed.   */
  const int32_t* stage_tab;
  /* ABI 9: device-evaluated state handlers (phx_stage_rule above); 0 / NULL: none.  A stage may have rules or a stage_tab row
   * that differs from stage_next, not both (PHX_EINVAL).                                                                    */
  int32_t n_stage_rules;
  int32_t reserved1;
  const phx_stage_rule* stage_rules;
} phx_spec;

/* phx_spec.variant_rollout: which kernel phx_rollout uses for a supply-chain env with a fused schedule */
#define PHX_VR_AUTO          0
#define PHX_VR_TIME_PARALLEL 1  /* plain env: phx_sc_rollout_fast_kernel; FSM env: phx_sc_rollout_fsmfast_kernel at any batch size */
#define PHX_VR_LEAN          2  /* FSM env: phx_sc_rollout_fsm_lean_kernel, one lane per (env, shop) pair                        */
#define PHX_VR_GENERAL       3  /* plain env: phx_sc_rollout_kernel, round 1, also serves replays; FSM env: phx_sc_rollout_fsm_kernel */
#define PHX_VR_LAUNCH_LOOP   4  /* generic engine: one phx_generic_step_kernel launch per step                                    */
#define PHX_VR_STORE_WAVES   5  /* plain env: the store-wave kernel of round 4 (phx_sc_rollout_sw.hip: dedicated store waves, dense flag planes); also what
                                   PHX_VR_AUTO picks where its workgroup shape applies; PHX_VR_TIME_PARALLEL keeps the round-3 kernel.
                                   FSM supply chain (round 5): that kernel's FSM instantiation wherever its plan applies (PHX_VR_AUTO: for
                                   fragments of >= 200 steps on batches above 65 536 (env, shop) pairs) */
#define PHX_VB_WHOLE_ENVS   (-1)
/* phx_spec.variant_step */
#define PHX_VS_AUTO          0
#define PHX_VS_FUSED         1  /* the static-schedule kernel of the env's family (default where one applies)                      */
#define PHX_VS_GENERIC       2  /* the message-passing engine (same as PHX_F_FORCE_GENERIC)                                       */
#define PHX_VS_GENERIC_DYNAMIC 4 /* the message-passing engine WITHOUT its compiled schedule: the dynamic kernel (LDS atomics, block scans, rank sort)
                                   for every step, also where phx_sched_step_kernel would serve the spec (ABI 10; tests compare the two)      */
#define PHX_VS_WIDE          3  /* plain supply chain, device-RNG orders: four (env, shop) pairs per thread, 16-byte accesses (AUTO
                                   takes it from 2^19 pairs per launch up; smaller launches are latency-bound either way)          */

typedef struct phx_env phx_env;   /* opaque */

/* State lives in ONE caller-owned device blob, struct-of-arrays by kind:
 * field f is `count` x B contiguous elements laid out [slot][B... see DESIGN.md] */
typedef struct phx_field {
  int32_t  field_id;
  int32_t  dtype;       /* 0=i32 1=f64 2=u8 3=f32 */
  int64_t  offset;      /* byte offset in the state blob */
  int32_t  dim0, dim1, dim2;  /* logical shape (dim2 = 1 when unused) */
  int32_t  kind;        /* owning phx_kind, 0 = env-level */
  char     name[32];
} phx_field;

/* one message-log record (Resolver.tracked_messages, resolvers.py:35-60) */
typedef struct phx_msg_rec {
  uint16_t sender, receiver, type, round;   /* phx_inject input: `round` = aux bits of the payload */
  union { int64_t i; double f; } payload;
} phx_msg_rec;

/* ---- step I/O: every pointer is a device pointer, NULL where noted --------------------
 * S = number of strategic agents (rank order = agent order), D = obs_dim (phx_obs_dim).   
 * Alignment: obs and actions 16 bytes, reward 8 bytes (PHX_EINVAL otherwise). */
typedef struct phx_step_io {
  const float*   actions;      /* [B][S]    one float per strategic agent                   */
  const uint8_t* action_valid; /* [B][S] or NULL (= every strategic agent has an action);
                                  0 <=> aid not in actions -> generate_messages env.py:330   */
  const uint8_t* exo;          /* [B][n_exo] exogenous draws (np.random.randint(5) of
                                  CustomerAgent, supply_chain.py:64) or NULL -> device RNG   */
  float*    obs;               /* [B][S][D]                                                 */
  uint8_t*  obs_valid;         /* [B][S]   1 <=> aid in step.observations                   */
  double*   reward;            /* [B][S]                                                    */
  uint8_t*  reward_valid;      /* [B][S]   0 absent, 1 value, 2 present-but-None fsm.py:378 */
  uint8_t*  terminated;        /* [B][S]                                                    */
  uint8_t*  truncated;         /* [B][S]                                                    */
  uint8_t*  done_valid;        /* [B][S]   1 <=> aid in step.terminations                   */
  uint8_t*  all_terminated;    /* [B]      terminations["__all__"]  env.py:297              */
  uint8_t*  all_truncated;     /* [B]      truncations["__all__"]   env.py:298              */
  int32_t*  err;               /* [B]      PHX_ERR_*                                        */
  phx_msg_rec* msg_log;        /* [B][trace_cap] or NULL                                    */
  int32_t*  msg_count;         /* [B] or NULL                                               */
  /* ABI 5, PHX_F_SHUFFLE_BATCHES only: recorded np.random.shuffle outcomes, or NULL -> device RNG.
   * Entry (messages of the step's earlier rounds + P), P the inbox position of the round in
   * receiver-major order (receivers in first-arrival order), holds the batch-local index (send
   * order) of the message handled at that position.  shuffle_cap = 8 * queue_cap entries per env. */
  const uint16_t* shuffle;     /* [B][8 * queue_cap] or NULL                                */
  /* FiniteStateMachineEnv stage HANDLERS (fsm.py:294-307): the stage a Python handler returned for each env, or NULL ->
   * the tabulated / device-evaluated handler of the current stage (phx_spec.stage_tab, phx_spec.stage_rule), else its
   * next_stages[0].  A handler that reads agent state is called by the host where the reference calls it: BETWEEN
   * phx_step_begin (acting phase + resolve_network) and phx_step_end, which takes this field; with ONE phx_step the
   * values must have been decided before the launch (handlers that only look at the clock / the stage).  Checked against
   * phx_spec.stage_allowed; an invalid transition sets PHX_ERR_FSM_TRANSITION.  The agents acting in THAT stage are the
   * ones that observe (fsm.py:320).  Runs on the generic engine.                                                */
  const int32_t* next_stage;   /* [B] or NULL                                               */
} phx_step_io;

/* ---- fused on-device rollout: T consecutive steps per launch, auto-reset at episode end.
 * Every OUTPUT buffer must be 16-byte aligned (the kernels write 16-byte pieces); phx_rollout returns PHX_EINVAL otherwise.  The replayed
 * inputs are read one element at a time: `actions` 4-byte aligned, `exo` any address (a row slice of a longer recording is fine). */
/* (ABI 9 removed three experimental pieces of ABI 7-8 that every measurement since had left behind: the PHX_RH_FLAGS_ZEROED hint, the
 *  24-byte record layout phx_rollout_io.records and phx_spec.variant_flags -- DESIGN_HISTORY.md has their numbers.)                   */
/* ABI 9: one trajectory fragment of a launch that writes SEVERAL (phx_rollout_io.frags): the planes of phx_rollout_io, each
 * [frag_T][B][S](..) -- separate allocations, e.g. the buffers a learner takes one at a time.                              */
#define PHX_MAX_FRAGMENTS 8
typedef struct phx_rollout_frag {
  float*    obs;               /* [frag_T][B][S][D]                                         */
  float*    action_out;        /* [frag_T][B][S]                                            */
  float*    reward;            /* [frag_T][B][S]                                            */
  uint8_t*  terminated;        /* [frag_T][B][S] or NULL (all fragments alike), as phx_rollout_io.terminated */
  uint8_t*  truncated;         /* [frag_T][B][S]                                            */
  uint8_t*  obs_valid;         /* [frag_T][B][S] or NULL (FSM / Stackelberg / ads envs)     */
  uint8_t*  reward_valid;      /* [frag_T][B][S] or NULL                                    */
} phx_rollout_frag;

/* phx_rollout_io.hints: what the CALLER vouches for about the replayed inputs, so that a plain supply chain's replay can take the
 * store-wave kernel (whose tiles hold R, D and the stock in bytes) without a scan of the inputs.  A hint that does not hold leaves the
 * rows of the envs it fails for unspecified AND sets their err[b] = PHX_ERR_HINT wherever the kernel that serves the call relies on
 * the hint (ABI 10: the store-wave kernel sees the offending action / order byte where it loads it; until ABI 9 nothing reported it).
 * Kernels that do not rely on a hint (round 1's kernel, short fragments) ignore it: the reference's rows, no error.
 * (Bit 1 was PHX_RH_FLAGS_ZEROED until ABI 8: not reused.)                                                                          */
#define PHX_RH_ACTIONS_IN_DOMAIN 2  /* every replayed action rounds to >= 0 -- e.g. clipped to ShopAgent's action space Box(0, SHOP_MAX_STOCK),
                                       supply_chain.py:87-91, as RLlib's clip_actions does.  Without it the call's actions are pre-scanned on
                                       the device (T B S floats read once more) and a call with an action that rounds below zero is served
                                       by round 1's kernel.                                                                           */
#define PHX_RH_EXO_IN_DOMAIN     4  /* every byte of `exo` is a draw of np.random.randint(CUSTOMER_MAX_ORDER_SIZE = 5), i.e. < 5
                                       (supply_chain.py:64) -- what phx_mt_draw produces.  Without it replayed order sizes are served by
                                       round 1's kernel (32-bit tiles, any byte value).                                               */
/* ABI 10: a POLICY evaluated on the device inside the fused rollout (what the reference's collection loop calls for every agent and
 * step, utils/rllib/rollout.py:300-363): one small MLP shared by the env's strategic agents, fed with the agent's previous observation
 * (the reset observation at an episode's first step; at the fragment's first step what encode_observation gives on the state the env
 * is in).  f32 throughout; the arithmetic is DEFINED here (the oracle restates it bit for bit):
 *     x[0 .. D)            the observation (D = phx_obs_dim)
 *     h0[i] = act(c),  c = b[0][i];  for k = 0 .. D-1 ascending:        c = fmaf(w[0][i * D + k], x[k], c)
 *     h1[i] = act(c),  c = b[1][i];  for k = 0 .. width[0]-1 ascending: c = fmaf(w[1][i * width[0] + k], h0[k], c)      (n_hidden == 2)
 *     y = b[n_hidden][0];            for k = 0 .. width[last]-1 ascending: y = fmaf(w[n_hidden][k], h[k], y)
 *     a = fmaf(out_scale, y, out_bias);   action = (a < out_lo ? out_lo : (a > out_hi ? out_hi : a)) + 0.0f      (an exact zero is +0)
 *     act = PHX_ACT_RELU: c > 0 ? c : +0;   PHX_ACT_HARD_TANH: c < -1 ? -1 : (c > 1 ? 1 : c)
 * (fmaf = the correctly rounded fused multiply-add of C99 / v_fma_f32: one rounding per term, on every machine).  The layouts are
 * torch.nn.Linear's own (weight [out][in] row-major, bias [out]): the parameters of a module are passed as they are, device pointers,
 * read during the launch.  Weights and observations must be finite.  Served for plain supply-chain envs (ShopAgent observations,
 * D = 3) by phx_sc_rollout_policy_kernel, one lane per (env, shop); out_lo >= 0 (ShopAgent's action space is Box(0, SHOP_MAX_STOCK)).  */
#define PHX_ACT_RELU      0
#define PHX_ACT_HARD_TANH 1
#define PHX_POLICY_MAX_WIDTH 64
typedef struct phx_policy_mlp {
  int32_t n_hidden;            /* hidden layers: 1 or 2                                     */
  int32_t width[2];            /* their units, 1 .. PHX_POLICY_MAX_WIDTH                    */
  int32_t activation;          /* PHX_ACT_*                                                 */
  float   out_scale, out_bias; /* a = fmaf(out_scale, y, out_bias)                          */
  float   out_lo, out_hi;      /* action = clip(a, out_lo, out_hi)                          */
  const float* w[3];           /* device: [width0][D], [width1][width0] (or the output row when n_hidden == 1), [1][width_last] */
  const float* b[3];           /* device: [width0], [width1] (or [1]), [1]                  */
} phx_policy_mlp;

typedef struct phx_rollout_io {
  int32_t T;
  int32_t hints;               /* PHX_RH_* below, 0 = none                                  */
  const float*   actions;      /* [T][B][S] replayed policy, or NULL -> random U[0,100)     */
  const uint8_t* exo;          /* [T][B][n_exo] or NULL -> device RNG                       */
  float*    obs;               /* [T][B][S][D]  post-step observation                       */
  float*    action_out;        /* [T][B][S]     action taken                                */
  float*    reward;            /* [T][B][S]     f64 reward rounded to f32                   */
  uint8_t*  terminated;        /* [T][B][S]; may be NULL where the plane is all zero and the serving kernel can omit it
                                  (plain supply-chain env on the time-parallel rollout kernel), PHX_EINVAL otherwise */
  uint8_t*  truncated;         /* [T][B][S]     per-agent flag OR'ed with __all__ truncation */
  uint8_t*  obs_valid;         /* [T][B][S] or NULL: 1 <=> aid in step.observations (FSM envs) */
  uint8_t*  reward_valid;      /* [T][B][S] or NULL: 0 absent, 1 value, 2 None (FSM envs)      */
  float*    last_obs;          /* [B][S][D]     observation the next fragment starts from   */
  int32_t*  err;               /* [B]                                                       */
  /* ABI 4: Resolver.tracked_messages of every step of the fragment (rollout.py:369-373,
   * record_messages=True).  Needs trace_cap > 0 (tracking on, which keeps the env on the
   * generic engine's launch loop); both NULL = not recorded.                               */
  phx_msg_rec* msg_log;        /* [T][B][trace_cap] or NULL                                 */
  int32_t*  msg_count;         /* [T][B] or NULL                                            */
  void*     reserved_ptr;      /* NULL (ABI 7-8: the record layout, removed in ABI 9)       */
  /* ABI 9: a fragment LIST.  n_frag >= 2 (<= PHX_MAX_FRAGMENTS) and frags != NULL (a HOST array, read during the call): the launch
   * advances the envs T steps as always and writes rows [f T / n_frag, (f + 1) T / n_frag) to frags[f] (T % n_frag == 0; obs,
   * action_out, reward, terminated, truncated, obs_valid, reward_valid of the io itself must then be NULL; actions / exo / msg_log /
   * msg_count stay [T][..]; last_obs is the observation after step T).  Why: the fixed cost of a rollout launch on this chip -- a
   * pipeline to fill, a 160 KB workgroup to place on every CU, the kernel boundary -- is ~9 us against 12.5 us of streaming per 100
   * steps of the BASELINE config: a consumer of 100-step fragments asks for k of them per call and gets the rate of one k x 100-step
   * launch (the list-of-envs collection loop, utils/rllib/rollout.py:361-363, fills its buffers episode after episode in the same
   * way).  The store-wave supply-chain kernel writes the list from ONE launch; every other env is served by n_frag consecutive
   * launches inside the call (same results, no gain).  0 / 1 and NULL: the io's own planes.                                    */
  int32_t   n_frag;
  int32_t   reserved0;
  const phx_rollout_frag* frags;
  /* ABI 10: the policy of the rollout evaluated on the device (a HOST struct, read during the call), or NULL -> `actions` / the random
   * policy.  Excludes `actions`, fragment lists and message logs (PHX_EINVAL); PHX_EUNSUPPORTED for envs the policy kernel does not serve. */
  const phx_policy_mlp* policy;
} phx_rollout_io;

/* ---- entry points ---------------------------------------------------------------------- */
int         phx_abi_version(void);
const char* phx_last_error(void);
/* names of the kernels the calling thread's last phx_step / phx_rollout / phx_resolve launched, joined by '+'
 * (kernel-variant tests: phx_spec.variant_*) */
const char* phx_last_kernel(void);
/* ABI 10: what PHX_VR_AUTO measured when it last had two kernels for a rollout shape of this handle -- e.g. "T=400 n_frag=0: store-wave
 * 861.2 us, lane-per-pair chain 772.4 us -> lane-per-pair chain" -- or "" (FSM supply chains: the first phx_rollout of a (T, n_frag) shape
 * times both from a copy of the state blob, restores it, and the handle keeps the winner; nothing is timed while a stream is capturing). */
const char* phx_autotune_note(const phx_env* env);

/* sizes derived from the spec, so the caller (torch) can own every buffer */
int64_t phx_state_nbytes(const phx_spec* spec);
int     phx_obs_dim(const phx_spec* spec);
int     phx_n_strategic(const phx_spec* spec);
int     phx_n_exo(const phx_spec* spec);

/* replaces PhantomEnv.__init__ bookkeeping (env.py:55-124): validates the spec, uploads the
 * tables to `device`, binds the caller's state blob (zero-initialised by this call).       */
int  phx_create(const phx_spec* spec, int device, void* state_blob, int64_t state_nbytes,
                phx_env** out);
void phx_destroy(phx_env* env);

int  phx_n_fields(const phx_env* env);
int  phx_field_info(const phx_env* env, int index, phx_field* out);
/* 1 when step/rollout run a fused static-schedule kernel, 0 for the generic engine        */
int  phx_uses_fused(const phx_env* env);
/* Brings lazily maintained state fields up to date before the caller reads them: the fused
 * Stackelberg kernel keeps BuyerAgent.prices (one f64 per buyer and neighbour) in its
 * compressed form "last price posted by each seller" (seller.posted) and materialises the
 * buyer.prices field only here.  A no-op for every other field / engine.                  */
int  phx_sync_fields(phx_env* env, void* stream);

/* PhantomEnv.reset (env.py:185-237 / fsm.py:195-251 / stackelberg.py:53-109) for every env
 * with reset_mask[b] != 0 (NULL = all).  Writes the initial observations.
 * sampler_values: device f64 [B][n_samplers], the values `sampler.sample()` returned for each
 * env (env.py:211-212), or NULL: UNIFORM samplers are then drawn from the device Philox
 * stream (ctr = (env, episode, 0x80000000 | column)), HOST samplers keep their value.
 * conn_on: device u8 [B][n_conn], the outcome of `np.random.random() < rate` per base connection
 * and env (network.py:444-447), or NULL: drawn on the device (ctr = (env, episode,
 * 0x40000000 | connection / 2), u < rate).                                                  */
int  phx_reset(phx_env* env, const uint8_t* reset_mask, const double* sampler_values,
               const uint8_t* conn_on, float* obs, uint8_t* obs_valid, void* stream);

/* one PhantomEnv.step for all B envs */
int  phx_step(phx_env* env, const phx_step_io* io, void* stream);
/* ABI 8: the two halves of a FiniteStateMachineEnv.step around a host-side stage handler that branches on agent state
 * (fsm.py:275-307: the handler runs AFTER _handle_acting_agents, calls self.resolve_network(), then returns the next stage):
 *   phx_step_begin -- the acting phase and resolve_network() (phx_step_io inputs; msg_log / msg_count / err outputs); the step
 *                     counter, the tick and the stage are NOT advanced, no observation is written;
 *   [the caller's handler reads the resolved state: phx_field_info views or phx_get_state]
 *   phx_step_end   -- io->next_stage [B] (NULL: next_stages[0] / the tabulated handler) -> the transition, observations,
 *                     rewards, done flags (every output of phx_step_io).
 * begin followed by end equals ONE phx_step with the same next_stage.  Served by the message-passing engine. */
int  phx_step_begin(phx_env* env, const phx_step_io* io, void* stream);
int  phx_step_end(phx_env* env, const phx_step_io* io, void* stream);

/* Network.send from outside a step (network.py:233-254; tests/network/ tests): queue n host
 * messages, the same for every env, delivered by the next phx_resolve / phx_step.          */
int  phx_inject(phx_env* env, const phx_msg_rec* host_msgs, int n);
/* Network.resolve(contexts) alone (network.py:256-265): no clock tick, no observations     */
int  phx_resolve(phx_env* env, int32_t* err, phx_msg_rec* msg_log, int32_t* msg_count,
                 void* stream);

/* T consecutive steps with auto-reset at episode end (the list-of-envs loop of
 * utils/rllib/rollout.py:361-363).  Static supply-chain schedules (plain or FSM env) and the static
 * Stackelberg / digital-ads markets run as ONE fused kernel; every other env runs as a stream-ordered
 * launch loop (one generic-engine launch per step with the policy, the trajectory row and the reset at an
 * episode end fused in; intermediates in the state blob's "rollout.scratch" field).  PHX_EUNSUPPORTED for envs with PHX_SAMPLER_HOST samplers: the
 * auto-reset resamples on the device.                                                          */
int  phx_rollout(phx_env* env, const phx_rollout_io* io, void* stream);

/* ---- state / trace access by name.  The state blob is CALLER-owned and phx_field_info
 * gives zero-copy views of it; these two are the copying form of the same access, for bindings that
 * cannot alias device memory: `field` is a phx_field.name ("shop.stock", "env.step", ...), `buf` a
 * device or host pointer (hipMemcpyDefault) of at least the field's byte size; returns the number of
 * bytes copied or a negative PHX_E* code.  Lazily maintained fields are brought up to date first
 * (phx_sync_fields).  Replace reading / poking agent attributes on the reference's Python objects
 * (metrics.py:189-231 reflection; tests that set ShopAgent.stock).                                */
int64_t phx_get_state(phx_env* env, const char* field, void* buf, int64_t buf_nbytes, void* stream);
int64_t phx_set_state(phx_env* env, const char* field, const void* buf, int64_t buf_nbytes, void* stream);
/* Resolver.tracked_messages of the last phx_step / phx_resolve that was given a message log
 * (resolvers.py:35-60): copies min(count[b], trace_cap) records of env `b` from the caller's log
 * buffers to HOST memory `out` (capacity `cap` records) and returns count[b] -- the copying form of
 * reading phx_step_io.msg_log / msg_count directly.  Synchronises `stream`.                       */
int  phx_trace(phx_env* env, const phx_msg_rec* msg_log, const int32_t* msg_count, int b,
               phx_msg_rec* out, int cap, void* stream);

/* ---- ABI 7: the reference's own random stream, per env instance (PHX_F_MT19937) ---------------------------------
 * CustomerAgent.generate_messages draws np.random.randint(CUSTOMER_MAX_ORDER_SIZE) from the process-global legacy
 * MT19937 (examples/environments/supply_chain/supply_chain.py:64); a reference rollout worker runs ONE env on its own
 * stream (utils/rllib/rollout.py:220-258, one process per rollout task).  With PHX_F_MT19937 env instance b of the
 * batch carries that stream:
 *   phx_mt_seed(env, seeds)     == np.random.seed(seeds[b]) in worker b   (init_genrand, numpy random/_mt19937 legacy
 *                                  seeding of a 32-bit integer; `seeds` = B HOST words),
 *   phx_mt_draw(env, exo, T)    == the np.random.randint(5) calls worker b makes in the next T steps, in the reference's call order:
 *                                  step by step, the CustomerAgents of the step's acting list in acting order (a PLAIN env: all of
 *                                  them; a FiniteStateMachineEnv: those of the env's stage, fsm.py:276-279 -- the stages are walked
 *                                  forward from the env's current step / stage words along phx_spec.stage_next / stage_tab with the
 *                                  reset at the episode's end, as phx_rollout walks them); numpy's masked rejection, one 32-bit
 *                                  word per attempt -- v = genrand_uint32() & 7 until v <= 4 (numpy 2.2 legacy RandomState.randint
 *                                  -> _bounded_integers, use_masked; pinned by tests against numpy itself and against the draws the
 *                                  REFERENCE consumed in the seeded golden runs, tests/golden/{sc64,sc_fsm_small,sc256_fsm}.npz `exo`).
 * `exo` is a DEVICE buffer u8 [T][B][n_exo], laid out like phx_rollout_io.exo / T stacked phx_step_io.exo (entries of customers that
 * do not act in a step are 0): pass it on to phx_step / phx_rollout and instance b reproduces reference worker b's seeded run bit for
 * bit, at any batch size, without a host loop.  The stream position persists in the blob between calls (phx_get_state /
 * phx_set_state reach "env.mt_state" / "env.mt_pos").  PHX_EUNSUPPORTED for specs without the flag, for env types other than
 * PHX_ENV_PLAIN / PHX_ENV_FSM and for specs with a PublisherAgent (its binomial draws depend on the auction's outcome).  With stage
 * handlers called by the host (phx_step_io.next_stage) draw one step at a time: the walk follows stage_next / stage_tab.       */
int  phx_mt_seed(phx_env* env, const uint32_t* seeds, void* stream);
int  phx_mt_draw(phx_env* env, uint8_t* exo, int T, void* stream);

/* ---- rollout collection helpers (SURVEY 8e iii): done-flag planes bit-packed for the all-gather.
 * dst word w, bit j = (src[64 w + j] != 0), n = number of source bytes, dst = ceil(n / 64) words.   */
int  phx_pack_flags(const uint8_t* src, uint64_t* dst, int64_t n, void* stream);
int  phx_unpack_flags(const uint64_t* src, uint8_t* dst, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PHANTOM_AMD_H */

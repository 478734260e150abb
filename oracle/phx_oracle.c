/*
 * phx_oracle.c -- CPU ORACLE (test infrastructure, see phx_oracle.h).
 *
 * Sequential restatement of jpmorganchase/Phantom v2.2.0's PhantomEnv.step() path.  Each env
 * instance of the batch is stepped on its own by the same code the reference runs once per
 * Python env object; "file:line" comments point into /root/reference.
 *
 * Deliberately NOT optimised the way the device code is: inboxes are an insertion-ordered
 * map receiver -> list (the reference's DefaultDict), handlers run one message at a time.
 */
#include "phx_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SHOP_MAX_STOCK 100          /* supply_chain.py:13 */

static char g_err[256];
static int g_threads = 1;
const char* phxo_last_error(void) { return g_err; }
void phxo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int phxo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* The device's division by a loop-invariant divisor (phx_dev.h: div_by_recip -- the policy kernel's observations): with r = 1 / n rounded once,
   q0 = x * r, e = fmaf(-n, q0, x), q = fmaf(e, r, q0) IS the IEEE quotient x / n (Markstein's correction step) on the domain the kernel applies
   it to.  Counts the pairs 0 <= x < x_end, 1 <= n <= n_max where it is not (tests/test_host_logic.py: exhaustively, expected 0). */
int64_t phxo_check_recip_div(int n_max, int x_end) {
  int64_t bad = 0;
  for (int n = 1; n <= n_max; ++n) {
    const float fn = (float)n, r = 1.0f / fn;
    for (int x = 0; x < x_end; ++x) {
      const float fx = (float)x, q0 = fx * r, e = fmaf(-fn, q0, fx), q = fmaf(e, r, q0), ref = fx / fn;
      bad += memcmp(&q, &ref, 4) != 0;
    }
  }
  return bad;
}

/* ------------------------------------------------------------------------------------ */
typedef struct {
  int src, dst, type;
  int aux;                /* small extra payload fields (theme | user_id << 4 | tag << 8) */
  union { int64_t i; double f; } p;
} omsg;

typedef struct {          /* python attributes of one agent object */
  int32_t i[16];
  double f[2];
  double* vec;            /* BuyerAgent.prices[deg] */
} ostate;

typedef struct {          /* one BatchResolver.messages DefaultDict  resolvers.py:120 */
  omsg* pool;
  int* next;              /* next message of the same receiver */
  int n;
  int* order;             /* receivers in first-arrival (dict insertion) order */
  int n_recv;
  int* head;              /* [A] first message per receiver, -1 = key absent */
  int* tail;
} oinbox;

typedef struct {
  ostate* ag;             /* [A] */
  double* vecpool;
  int32_t step;           /* PhantomEnv._current_step env.py:63 */
  int32_t stage, prev_stage;   /* fsm.py:126-127 */
  uint32_t tick;          /* device-RNG draw counter (never reset) */
  double* sampler;        /* [n_samplers] Sampler._value of every sampler of env._samplers  samplers.py:60-66 */
  uint8_t* conn_on;       /* [n_conn] StochasticNetwork: base connection is in self.graph  network.py:438-447 */
  int32_t episode;        /* number of env.reset() calls so far (device-RNG counter of the samplers) */
  int32_t clock;          /* handle_message invocation counter (stands in for time.time()) */
  const uint8_t* exo_b;   /* this step's exogenous draws (NULL -> device RNG) */
  int64_t genv;           /* global env index (device-RNG counter) */
  uint8_t* term;          /* [S] PhantomEnv._terminations env.py:71 */
  uint8_t* trunc;         /* [S] PhantomEnv._truncations  env.py:72 */
  double* rew_cache;      /* [S] FSM/Stackelberg _rewards */
  uint8_t* rew_cache_valid;
  float* obs_cache;       /* [S][D] FSM _observations */
  uint8_t* obs_cache_valid;
  oinbox box[2];
  int cur;                /* which box is self.messages */
  int32_t err;
  phx_msg_rec* log;       /* Resolver._tracked_messages resolvers.py:37 */
  int log_cap, log_n;
  int round;
  const uint16_t* shuffle_b; /* this step's replayed shuffle permutations (NULL -> device RNG) */
  int shuf_pos;             /* queued messages of the step's earlier batches */
  int next_in;              /* stage chosen by the stage handler for the coming step (-1: none, -2: out of range) */
  int phase;                /* 0: a whole step; 1: phxo_step_begin (acting + resolve_network, fsm.py:275-280 + the handler's
                             * resolve); 2: phxo_step_end (the handler's stage -> transition, observations, rewards, :304-380) */
} oenv;

struct phxo_env {
  phx_spec s;             /* deep copy */
  int A, S, B, D, n_exo, nnz;
  int* strat_rank;        /* [A] rank among strategic agents or -1 */
  int* strat_idx;         /* [S] */
  int* kind_rank;         /* [A] rank among agents of the same kind */
  int kind_count[PHX_KIND_COUNT];
  int* exo_rank;          /* [A] rank among CUSTOMER agents or -1 */
  int* nbr_slot_base;     /* unused */
  omsg* injected; int n_injected;
  oenv* env;              /* [B] */
};

static int is_strategic_kind(int k) {
  return k == PHX_KIND_SHOP || k == PHX_KIND_SELLER || k == PHX_KIND_BUYER ||
         k == PHX_KIND_MOCK_STRAT || k == PHX_KIND_ADVERTISER;
}
static int obs_dim_of_kind(int k) {
  switch (k) {
    case PHX_KIND_SHOP: return 3;
    case PHX_KIND_SELLER: return 2;
    case PHX_KIND_BUYER: return 2;
    case PHX_KIND_MOCK_STRAT: return 1;
    case PHX_KIND_ADVERTISER: return 3;
    default: return 0;
  }
}

/* @msg_payload(sender_type, receiver_type) whitelists, message.py:20-42; 0 = None (any).  */
static void payload_types(int type, int* sender_kind, int* receiver_kind, int* decorated) {
  *decorated = 1; *sender_kind = 0; *receiver_kind = 0;
  switch (type) {
    case PHX_MSG_ORDER_REQUEST:  *sender_kind = PHX_KIND_CUSTOMER; *receiver_kind = PHX_KIND_SHOP; break;     /* supply_chain.py:16 */
    case PHX_MSG_ORDER_RESPONSE: *sender_kind = PHX_KIND_SHOP; *receiver_kind = PHX_KIND_CUSTOMER; break;     /* :21 */
    case PHX_MSG_STOCK_REQUEST:  *sender_kind = PHX_KIND_SHOP; *receiver_kind = PHX_KIND_FACTORY; break;      /* :26 */
    case PHX_MSG_STOCK_RESPONSE: *sender_kind = PHX_KIND_FACTORY; *receiver_kind = PHX_KIND_SHOP; break;      /* :31 */
    case PHX_MSG_PRICE:          *sender_kind = PHX_KIND_SELLER; *receiver_kind = PHX_KIND_BUYER; break;
    case PHX_MSG_ORDER:          *sender_kind = PHX_KIND_BUYER; *receiver_kind = PHX_KIND_SELLER; break;
    case PHX_MSG_PING:           *decorated = 0; break;       /* bare `True`, test_resolver.py:94 */
    default: break;                                           /* @msg_payload() : any -> any */
  }
}

/* CSR entry k is an edge of this env's graph (always, for a static Network) */
static int edge_on(const phxo_env* E, const oenv* e, int k) {
  return E->s.n_conn == 0 || e->conn_on[E->s.col_conn[k]];
}
static int has_edge(const phxo_env* E, const oenv* e, int u, int v) {   /* network.py:224-231 */
  for (int k = E->s.row_ptr[u]; k < E->s.row_ptr[u + 1]; ++k)
    if (E->s.col[k] == v && edge_on(E, e, k)) return 1;
  return 0;
}
static int nbr_slot(const phxo_env* E, int u, int v) {        /* index of v in ctx.neighbour_ids of u */
  for (int k = E->s.row_ptr[u]; k < E->s.row_ptr[u + 1]; ++k)
    if (E->s.col[k] == v) return k - E->s.row_ptr[u];
  return -1;
}

static void set_err(oenv* e, int code) { if (e->err == 0) e->err = code; }

static void inbox_clear(const phxo_env* E, oinbox* b) {       /* resolvers.py:122-123 */
  b->n = 0; b->n_recv = 0;
  for (int a = 0; a < E->A; ++a) b->head[a] = b->tail[a] = -1;
}

/* Resolver.push + BatchResolver.handle_push  resolvers.py:39-46,125-126 */
static void resolver_push(const phxo_env* E, oenv* e, const omsg* m) {
  if (e->log && e->log_n < e->log_cap) {                      /* enable_tracking */
    phx_msg_rec* r = &e->log[e->log_n];
    r->sender = (uint16_t)m->src; r->receiver = (uint16_t)m->dst;
    r->type = (uint16_t)m->type; r->round = (uint16_t)e->round;
    r->payload.i = m->p.i;
  }
  if (e->log) e->log_n++;
  oinbox* b = &e->box[e->cur];
  if (b->n >= E->s.queue_cap) { set_err(e, PHX_ERR_QUEUE_FULL); return; }
  int id = b->n++;
  b->pool[id] = *m; b->next[id] = -1;
  if (b->head[m->dst] < 0) {                                  /* new dict key */
    b->head[m->dst] = id; b->order[b->n_recv++] = m->dst;
  } else {
    b->next[b->tail[m->dst]] = id;
  }
  b->tail[m->dst] = id;
}

/* Network.send  network.py:233-254 (+ _enforce_payload_checks :297-331) */
static void network_send(const phxo_env* E, oenv* e, int src, int dst, int type, omsg payload) {
  if (!(E->s.flags & PHX_F_IGNORE_CONN_ERRORS) && !has_edge(E, e, src, dst)) {
    set_err(e, PHX_ERR_NETWORK); return;                      /* raise NetworkError :246-249 */
  }
  if (!(E->s.flags & PHX_F_NO_PAYLOAD_CHECKS)) {
    int sk, rk, dec; payload_types(type, &sk, &rk, &dec);
    if (!dec) { set_err(e, PHX_ERR_PAYLOAD); return; }        /* :311-313 */
    if (sk && E->s.kind[src] != sk) { set_err(e, PHX_ERR_PAYLOAD); return; }   /* :317-323 */
    if (rk && E->s.kind[dst] != rk) { set_err(e, PHX_ERR_PAYLOAD); return; }   /* :325-331 */
  }
  payload.src = src; payload.dst = dst; payload.type = type;
  resolver_push(E, e, &payload);
}

static omsg mk_i(int64_t v) { omsg m; memset(&m, 0, sizeof m); m.p.i = v; return m; }
static omsg mk_f(double v)  { omsg m; memset(&m, 0, sizeof m); m.p.f = v; return m; }

/* ---- Supertypes / Samplers ---------------------------------------------------------------
 * agent.type.<field> of a managed supertype is the env-owned Sampler's current value
 * (supertype.py:23-24: `field.value` when `_managed`), i.e. column type_src[a] of e->sampler,
 * or the constant the supertype was built with (param_f[a][0]).                            */
static int type_src_of(const phxo_env* E, int a) { return E->s.type_src ? E->s.type_src[a] : PHX_TYPE_NONE; }
static int agent_is_typed(const phxo_env* E, int a) { return type_src_of(E, a) != PHX_TYPE_NONE; }
static double agent_type_value(const phxo_env* E, const oenv* e, int a) {
  const int src = type_src_of(E, a);
  return src >= 0 ? e->sampler[src] : E->s.param_f[a * PHX_NPF + 0];
}

/* ---- device-RNG definition (build-owned; replaces the global np.random stream) -------- */
void phxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* Device-RNG definition (build-owned; replaces the global numpy stream when exo == NULL).
 * np.random.randint(5) is exactly uniform on {0..4} (numpy draws 3 bits and rejects > 4).  The
 * device stream is exactly uniform too.  One Philox4x32-10 block
 *     ctr = (env_lo, env_hi | attempt << 16, tick >> 2, shop | g << 20),  key = seed
 * serves four consecutive ticks of customer group g (customers 6g .. 6g+5) of a shop; tick t owns
 * word t & 3.  With m = u * 5^6, a word u is rejected iff low32(m) < 2^32 mod 5^6 = 14171 (then
 * redrawn at the same position with attempt + 1); every y = m >> 32 in [0, 5^6) is hit by exactly
 * 274877 consecutive accepted words, so y and the rank j = (low32(m) - 14171) / 5^6 of the word
 * among them are independent and exactly uniform.  Customer i of the group orders base-5 digit i
 * of y; group 0's j is the shop's random-policy action, j * (100 / 274877).                     */
static uint32_t rng_word(uint64_t seed, int64_t genv, uint32_t tick, int shop, int g, uint32_t attempt) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32) | (attempt << 16), tick >> 2,
                     (uint32_t)shop | ((uint32_t)g << 20)};
  uint32_t w[4]; phxo_philox4x32_10(ctr, key, w);
  return w[tick & 3u];
}
static uint32_t rng_group_y(uint64_t seed, int64_t genv, uint32_t tick, int shop, int g, uint32_t* j) {
  for (uint32_t attempt = 0;; ++attempt) {
    uint64_t m = (uint64_t)rng_word(seed, genv, tick, shop, g, attempt) * 15625u;
    uint32_t l = (uint32_t)m;
    if (l >= 14171u) { if (j) *j = (l - 14171u) / 15625u; return (uint32_t)(m >> 32); }
  }
}
void phxo_rng_orders(uint64_t seed, int64_t genv, uint32_t tick, int shop, int K, uint8_t* out) {
  for (int k = 0; k < K; ++k) {
    uint32_t y = rng_group_y(seed, genv, tick, shop, k / 6, NULL);
    for (int i = 0; i < k % 6; ++i) y /= 5u;
    out[k] = (uint8_t)(y % 5u);
  }
}
/* rank j of the agent's word of this tick */
uint32_t phxo_rng_rank(uint64_t seed, int64_t genv, uint32_t tick, int agent) {
  uint32_t j; rng_group_y(seed, genv, tick, agent, 0, &j); return j;
}
/* random policy of the rollout for a shop: U[0,100) */
float phxo_rng_action(uint64_t seed, int64_t genv, uint32_t tick, int shop) {
  return (float)phxo_rng_rank(seed, genv, tick, shop) * (100.0f / 274877.0f);
}

/* Device draw of UniformFloatSampler column j at the env's `episode`-th reset (build-owned
 * definition, replaces np.random.uniform of samplers.py:141): Philox block
 *     ctr = (env_lo, env_hi, episode, 0x80000000 | j), key = seed,
 * u = ((w0 >> 5) * 2^26 + (w1 >> 6)) / 2^53 (numpy's 53-bit double), value = low + (high - low) * u
 * with product and sum rounded separately, then np.clip (samplers.py:143-144).             */
double phxo_rng_uniform(uint64_t seed, int64_t genv, uint32_t episode, int j, const double prm[4]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), episode, 0x80000000u | (uint32_t)j};
  uint32_t w[4]; phxo_philox4x32_10(ctr, key, w);
  const double u = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) / 9007199254740992.0;
  volatile double scaled = (prm[1] - prm[0]) * u;              /* volatile: no fma contraction */
  double v = prm[0] + scaled;
  if (prm[2] == prm[2] && v < prm[2]) v = prm[2];
  if (prm[3] == prm[3] && v > prm[3]) v = prm[3];
  return v;
}

/* Device draw of `np.random.random() < rate` for base connection i (network.py:444-447;
 * build-owned definition): Philox block ctr = (env_lo, env_hi, episode, 0x40000000 | i / 2),
 * u from words (2 (i % 2), 2 (i % 2) + 1) as above.                                          */
int phxo_rng_connection(uint64_t seed, int64_t genv, uint32_t episode, int i, double rate) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), episode, 0x40000000u | (uint32_t)(i >> 1)};
  uint32_t w[4]; phxo_philox4x32_10(ctr, key, w);
  const int h = 2 * (i & 1);
  const double u = ((double)(w[h] >> 5) * 67108864.0 + (double)(w[h + 1] >> 6)) / 9007199254740992.0;
  return u < rate;
}

/* env.reset(): `for sampler in self._samplers: sampler.sample()`  env.py:211-212, then
 * network.reset() -> StochasticNetwork.resample_connectivity()      env.py:218, network.py:449-452 */
static void env_sample(const phxo_env* E, oenv* e, int b, const double* values, const uint8_t* conn) {
  for (int j = 0; j < E->s.n_samplers; ++j) {
    if (values) e->sampler[j] = values[j];
    else if (E->s.sampler_kind[j] == PHX_SAMPLER_UNIFORM)
      e->sampler[j] = phxo_rng_uniform(E->s.seed, E->s.env_offset + b, (uint32_t)e->episode, j,
                                       E->s.sampler_param + 4 * j);
  }
  for (int i = 0; i < E->s.n_conn; ++i)
    e->conn_on[i] = conn ? (conn[i] != 0)
                         : (uint8_t)phxo_rng_connection(E->s.seed, E->s.env_offset + b, (uint32_t)e->episode, i, E->s.conn_rate[i]);
  e->episode += 1;
}

/* ---- numpy scalar promotion of the ads market's floats (NEP 50, numpy >= 2) ----------------
 * digital_ads_market.py mixes np.float32 (action[0]), python floats (a constant or unclipped
 * Supertype.budget, the literal 0.0 cost) and np.float64 (a clipped sampler's budget).  A binary
 * op yields the larger of the two tags PYF < F32 < F64 and is evaluated in float32 iff that tag
 * is F32 -- the weak python float is cast to float32 first.                                   */
typedef struct { double v; int tag; } tval;
static int rule_field(const char* name, int* kind, int* is_f, int* slot);   /* the state-field table below (FIELDS) */
static tval tv(double v, int tag) { tval t; t.v = v; t.tag = tag; return t; }
static int t_tag(tval a, tval b) { return a.tag > b.tag ? a.tag : b.tag; }
static tval t_mul(tval a, tval b) {
  int tag = t_tag(a, b);
  if (tag == PHX_TAG_F32) { volatile float r = (float)a.v * (float)b.v; return tv((double)r, tag); }
  return tv(a.v * b.v, tag);
}
static tval t_sub(tval a, tval b) {
  int tag = t_tag(a, b);
  if (tag == PHX_TAG_F32) { volatile float r = (float)a.v - (float)b.v; return tv((double)r, tag); }
  return tv(a.v - b.v, tag);
}
static tval t_div(tval a, tval b) {
  int tag = t_tag(a, b);
  if (tag == PHX_TAG_F32) { volatile float r = (float)a.v / (float)b.v; return tv((double)r, tag); }
  return tv(a.v / b.v, tag);
}
static int t_lt(tval a, tval b) {
  if (t_tag(a, b) == PHX_TAG_F32) return (float)a.v < (float)b.v;
  return a.v < b.v;
}
/* AdvertiserAgent attribute slots in ostate */
enum { ADV_LEFT_TAG = 0, ADV_BID_TAG = 1, ADV_CLICKS = 2, ADV_WINS = 3, ADV_USER = 4,
       ADV_TOT_CLICKS = 5, ADV_TOT_REQUESTS = 8, ADV_TOT_WINS = 11 };   /* totals: [user 0,1,2] */
static tval adv_budget(const phxo_env* E, const oenv* e, int a) {       /* self.type.budget */
  return tv(agent_type_value(E, e, a), E->s.param_i[a * PHX_NPI + 2] ? PHX_TAG_F64 : PHX_TAG_PYF);
}

/* Device draws of the PublisherAgent when exo == NULL (build-owned; replace np.random.choice([1, 2])
 * digital_ads_market.py:52 and np.random.binomial(1, p) :193-195): Philox block
 *     ctr = (env_lo, env_hi, tick, 0x20000000 | k << 16 | agent), key = seed,
 * k = 0: user_id = 1 + (w0 & 1);  k >= 1 (k-th Ads of the step): clicked = (w0 >> 8) * 2^-24 < p. */
int phxo_rng_publisher(uint64_t seed, int64_t genv, uint32_t tick, int agent, int k, double p) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), tick,
                     0x20000000u | ((uint32_t)k << 16) | (uint32_t)agent};
  uint32_t w[4]; phxo_philox4x32_10(ctr, key, w);
  if (k == 0) return 1 + (int)(w[0] & 1u);
  return (double)(w[0] >> 8) * (1.0 / 16777216.0) < p;
}

/* ---- per-kind agent behaviour --------------------------------------------------------- */

/* Agent.reset / subclasses: agents.py:160-175 */
static void agent_reset(const phxo_env* E, oenv* e, int a) {
  ostate* st = &e->ag[a];
  switch (E->s.kind[a]) {
    case PHX_KIND_SHOP: st->i[0] = 0; break;                  /* self.stock = 0  supply_chain.py:149-150 */
    case PHX_KIND_CASHBOX: st->f[0] = 0.0; break;             /* test_network.py:23-24 */
    case PHX_KIND_SELLER: st->f[0] = 0.0; st->f[1] = 0.0; st->i[0] = 0; break;
    case PHX_KIND_BUYER: {
      int deg = E->s.row_ptr[a + 1] - E->s.row_ptr[a];
      for (int k = 0; k < deg; ++k) st->vec[k] = 1.0;
      st->f[0] = 0.0; st->i[0] = 0; break;
    }
    case PHX_KIND_ADVERTISER: {                               /* digital_ads_market.py:353-374 */
      tval budget = adv_budget(E, e, a);
      st->f[0] = budget.v; st->i[ADV_LEFT_TAG] = budget.tag;  /* self.left = self.type.budget */
      st->f[1] = 0.0; st->i[ADV_BID_TAG] = PHX_TAG_PYF;       /* self.bid = 0.0 */
      for (int k = ADV_CLICKS; k < 14; ++k) st->i[k] = 0;     /* step_*, _current_user_id, total_* */
      break;
    }
    default: break;
  }
}

static int int_round_half_even(float a) {
  /* int(round(np.float32)) : round-half-to-even supply_chain.py:139 */
  double r = rint((double)a);
  if (r > 1073741824.0) r = 1073741824.0;      /* keep inside int32 (documented domain limit) */
  if (r < -1073741824.0) r = -1073741824.0;
  return (int)r;
}

/* decode_action (env.py:330-331) or generate_messages (:332-333) of agent a */
static void agent_act(const phxo_env* E, oenv* e, int b, int a, int has_action, float action,
                      const uint8_t* exo_b) {
  ostate* st = &e->ag[a];
  const int32_t* pi = &E->s.param_i[a * PHX_NPI];
  switch (E->s.kind[a]) {
    case PHX_KIND_SHOP:
      if (has_action) {                                       /* supply_chain.py:136-142 */
        int req = int_round_half_even(action);
        int room = SHOP_MAX_STOCK - st->i[0];
        int stock_to_request = req < room ? req : room;
        network_send(E, e, a, pi[0], PHX_MSG_STOCK_REQUEST, mk_i(stock_to_request));
      }                                                       /* else Agent.generate_messages -> [] agents.py:157 */
      break;
    case PHX_KIND_CUSTOMER: {                                 /* supply_chain.py:61-67 */
      int order_size;
      if (exo_b) order_size = exo_b[E->exo_rank[a]];
      else {
        /* device stream: base-5 digit pi1 % 6 of the word of customer group pi1 / 6 (pi1 = index among
         * the shop's customers) */
        uint32_t y = rng_group_y(E->s.seed, E->s.env_offset + b, e->tick, E->kind_rank[pi[0]], pi[1] / 6, NULL);
        for (int i = 0; i < pi[1] % 6; ++i) y /= 5u;
        order_size = (int)(y % 5u);
      }
      network_send(E, e, a, pi[0], PHX_MSG_ORDER_REQUEST, mk_i(order_size));
      break;
    }
    case PHX_KIND_SELLER:
      if (has_action) {
        st->f[0] = (double)action;                            /* self.price = float(action[0]) */
        for (int k = E->s.row_ptr[a]; k < E->s.row_ptr[a + 1]; ++k)   /* ctx.neighbour_ids order */
          if (edge_on(E, e, k)) network_send(E, e, a, E->s.col[k], PHX_MSG_PRICE, mk_f(st->f[0]));
      }
      break;
    case PHX_KIND_BUYER:
      if (has_action) {
        int deg = E->s.row_ptr[a + 1] - E->s.row_ptr[a];
        int j = -1;                                           /* first minimum over the current neighbours */
        for (int k = 0; k < deg; ++k)
          if (edge_on(E, e, E->s.row_ptr[a] + k) && (j < 0 || st->vec[k] < st->vec[j])) j = k;
        if (action > 0.5f && j >= 0) {
          st->i[0] = 1; st->f[0] = st->vec[j];
          network_send(E, e, a, E->s.col[E->s.row_ptr[a] + j], PHX_MSG_ORDER, mk_i(1));
        } else { st->i[0] = 0; st->f[0] = 0.0; }
      }
      break;
    case PHX_KIND_MOCK_STRAT:
      if (has_action) st->i[1] += 1;                          /* decode_action_count tests/__init__.py:53-55 */
      break;
    case PHX_KIND_PUBLISHER: {                                /* digital_ads_market.py:164-165, :48-53 */
      int user = exo_b ? exo_b[E->exo_rank[a]]
                       : phxo_rng_publisher(E->s.seed, E->s.env_offset + b, e->tick, a, 0, 0.0);
      network_send(E, e, a, pi[0], PHX_MSG_IMPRESSION_REQ, mk_i(user));
      break;
    }
    case PHX_KIND_ADVERTISER:
      if (has_action) {                                       /* digital_ads_market.py:318-333 */
        tval left = tv(st->f[0], st->i[ADV_LEFT_TAG]);
        tval bid = t_mul(tv((double)action, PHX_TAG_F32), adv_budget(E, e, a));
        if (t_lt(left, bid)) bid = left;                      /* min(action[0] * budget, self.left) */
        st->f[1] = bid.v; st->i[ADV_BID_TAG] = bid.tag;
        if (bid.v > 0.0) {
          omsg m = mk_f(bid.v);
          m.aux = pi[1] | (st->i[ADV_USER] << 4) | (bid.tag << 8);
          network_send(E, e, a, pi[0], PHX_MSG_BID, m);
        }
      }
      break;
    default: break;                                           /* generate_messages -> [] */
  }
}

static void agent_pre_resolution(const phxo_env* E, oenv* e, int a) {
  ostate* st = &e->ag[a];
  switch (E->s.kind[a]) {
    case PHX_KIND_SHOP: st->i[1] = 0; st->i[2] = 0; break;    /* sales, missed_sales supply_chain.py:93-96 */
    case PHX_KIND_SELLER:
      if (e->step % 2 == 0) { st->f[1] = 0.0; st->i[0] = 0; } /* revenue, tx of the buying round */
      break;
    case PHX_KIND_ADVERTISER: st->i[ADV_CLICKS] = 0; st->i[ADV_WINS] = 0; break;   /* digital_ads_market.py:241-247 */
    case PHX_KIND_PUBLISHER: st->i[0] = 0; break;             /* build bookkeeping: Ads handled this step */
    default: break;
  }
}

/* Agent.handle_message agents.py:122-155 -- type dispatch to the @msg_handler methods */
static void agent_handle_message(const phxo_env* E, oenv* e, int a, const omsg* m, int clock) {
  ostate* st = &e->ag[a];
  const int32_t* pi = &E->s.param_i[a * PHX_NPI];
  switch (E->s.kind[a]) {
    case PHX_KIND_FACTORY:
      if (m->type == PHX_MSG_STOCK_REQUEST) {                 /* supply_chain.py:40-45 */
        network_send(E, e, a, m->src, PHX_MSG_STOCK_RESPONSE, mk_i(m->p.i));
        return;
      }
      break;
    case PHX_KIND_SHOP:
      if (m->type == PHX_MSG_STOCK_RESPONSE) {                /* supply_chain.py:98-103 */
        st->i[3] = (int32_t)m->p.i;                           /* delivered_stock */
        int ns = st->i[0] + st->i[3];
        st->i[0] = ns < SHOP_MAX_STOCK ? ns : SHOP_MAX_STOCK;
        return;
      }
      if (m->type == PHX_MSG_ORDER_REQUEST) {                 /* supply_chain.py:105-122 */
        int amount_requested = (int)m->p.i, stock_to_sell;
        if (amount_requested > st->i[0]) {
          st->i[2] += amount_requested - st->i[0];
          stock_to_sell = st->i[0];
          st->i[0] = 0;
        } else {
          stock_to_sell = amount_requested;
          st->i[0] -= amount_requested;
        }
        st->i[1] += stock_to_sell;
        network_send(E, e, a, m->src, PHX_MSG_ORDER_RESPONSE, mk_i(stock_to_sell));
        return;
      }
      break;
    case PHX_KIND_CUSTOMER:
      if (m->type == PHX_MSG_ORDER_RESPONSE) return;          /* supply_chain.py:55-59 */
      break;
    case PHX_KIND_SELLER:
      if (m->type == PHX_MSG_ORDER) {
        double vol = (double)m->p.i;
        double amount = st->f[0] * vol;                       /* self.revenue += self.price * vol */
        st->f[1] = st->f[1] + amount;
        st->i[0] += (int)m->p.i;
        return;
      }
      break;
    case PHX_KIND_BUYER:
      if (m->type == PHX_MSG_PRICE) {
        int slot = nbr_slot(E, a, m->src);
        if (slot >= 0) st->vec[slot] = m->p.f;
        return;
      }
      break;
    case PHX_KIND_HALVER:
      if (m->type == PHX_MSG_HALVE) {                         /* test_tracking.py:22-27 */
        if (m->p.i > 1) {
          int64_t v = m->p.i / 2;                             /* value // 2, value > 1 */
          network_send(E, e, a, m->src, PHX_MSG_HALVE, mk_i(v));
        }
        return;
      }
      break;
    case PHX_KIND_CASHBOX:
      if (m->type == PHX_MSG_CASH) {                          /* test_network.py:26-34 */
        if (m->p.f > 25) {
          st->f[0] += m->p.f / 2.0;
          network_send(E, e, a, m->src, PHX_MSG_CASH, mk_f(m->p.f / 2.0));
        }
        return;
      }
      break;
    case PHX_KIND_REQRESP:
      if (m->type == PHX_MSG_REQUEST) {                       /* test_resolver.py:31-37 */
        st->i[0] = clock;                                     /* self.req_time = time.time() */
        network_send(E, e, a, m->src, PHX_MSG_RESPONSE, mk_f(m->p.f / 2.0));
        return;
      }
      if (m->type == PHX_MSG_RESPONSE) { st->i[1] = clock; return; }   /* :39-45 */
      break;
    case PHX_KIND_PUBLISHER:
      if (m->type == PHX_MSG_ADS) {                           /* digital_ads_market.py:167-196 */
        const int theme = m->aux & 15, user = (m->aux >> 4) & 15;
        if (user < 1 || user > 2 || theme > 3) { set_err(e, PHX_ERR_CONTEXT); return; }   /* dict KeyError :194 */
        const double p = E->s.param_f[a * PHX_NPF + (user - 1) * 4 + theme];
        const int k = ++st->i[0];                             /* k-th click draw of this step */
        int clicked;
        if (e->exo_b) {
          if (k > pi[1]) { set_err(e, PHX_ERR_QUEUE_FULL); return; }
          clicked = e->exo_b[E->exo_rank[a] + k];
        } else clicked = phxo_rng_publisher(E->s.seed, e->genv, e->tick, a, k, p);
        network_send(E, e, a, (int)m->p.i, PHX_MSG_IMPRESSION_RES, mk_i(clicked));
        return;
      }
      break;
    case PHX_KIND_ADVERTISER:
      if (m->type == PHX_MSG_IMPRESSION_REQ) {                /* digital_ads_market.py:249-271 */
        st->i[ADV_USER] = (int)m->p.i;
        if (!has_edge(E, e, a, pi[0])) { set_err(e, PHX_ERR_CONTEXT); return; }   /* ctx[self.exchange_id] :261 */
        if (m->p.i >= 0 && m->p.i <= 2) st->i[ADV_TOT_REQUESTS + m->p.i] += 1;
        return;
      }
      if (m->type == PHX_MSG_AUCTION_RESULT) {                /* :273-282 */
        const int won = m->p.f != 0.0;
        st->i[ADV_WINS] += won;
        st->i[ADV_TOT_WINS + st->i[ADV_USER]] += won;
        tval left = t_sub(tv(st->f[0], st->i[ADV_LEFT_TAG]), tv(m->p.f, (m->aux >> 8) & 3));
        st->f[0] = left.v; st->i[ADV_LEFT_TAG] = left.tag;
        return;
      }
      if (m->type == PHX_MSG_IMPRESSION_RES) {                /* :284-292 */
        st->i[ADV_CLICKS] += (int)m->p.i;
        st->i[ADV_TOT_CLICKS + st->i[ADV_USER]] += (int)m->p.i;
        return;
      }
      break;
    case PHX_KIND_ADEXCHANGE:
      if (m->type == PHX_MSG_IMPRESSION_REQ) {                /* :415-427: forward to self.advertiser_ids */
        for (int k = E->s.row_ptr[a]; k < E->s.row_ptr[a + 1]; ++k)
          if (E->s.kind[E->s.col[k]] == PHX_KIND_ADVERTISER)
            network_send(E, e, a, E->s.col[k], PHX_MSG_IMPRESSION_REQ, mk_i(m->p.i));
        return;
      }
      break;
    case PHX_KIND_FORWARDER:                                  /* overrides handle_message test_resolver.py:93-96 */
      if (pi[0] >= 0) network_send(E, e, a, pi[0], PHX_MSG_PING, mk_i(1));
      return;
    default: break;
  }
  set_err(e, PHX_ERR_UNKNOWN_MSG);                            /* raise ValueError agents.py:140-143 */
}

/* returns 0 where encode_observation returns None (the caller then omits the agent, env.py:279-280) */
static int agent_encode_obs(const phxo_env* E, const oenv* e, int a, float* o) {
  const ostate* st = &e->ag[a];
  const int32_t* pi = &E->s.param_i[a * PHX_NPI];
  switch (E->s.kind[a]) {
    case PHX_KIND_SHOP: {                                     /* supply_chain.py:124-134 */
      double max_sales_per_step = (double)pi[1];
      o[0] = (float)((double)st->i[0] / (double)SHOP_MAX_STOCK);
      o[1] = (float)((double)st->i[1] / max_sales_per_step);
      o[2] = (float)((double)st->i[2] / max_sales_per_step);
      if (agent_is_typed(E, a))                               /* docs/user/tutorial2.rst:283-294 */
        o[3] = (float)(agent_type_value(E, e, a) / E->s.param_f[a * PHX_NPF + 1]);
      break;
    }
    case PHX_KIND_SELLER: {
      int deg = 0;                                            /* len(ctx.neighbour_ids) */
      for (int k = E->s.row_ptr[a]; k < E->s.row_ptr[a + 1]; ++k) deg += edge_on(E, e, k);
      o[0] = deg ? (float)((double)st->i[0] / (double)deg) : 0.0f;
      o[1] = (float)st->f[0];
      break;
    }
    case PHX_KIND_BUYER: {
      int deg = E->s.row_ptr[a + 1] - E->s.row_ptr[a], any = 0;
      double mn = 1.0;                                        /* min(prices.values(), default=1.0) */
      for (int k = 0; k < deg; ++k)
        if (edge_on(E, e, E->s.row_ptr[a] + k) && (!any || st->vec[k] < mn)) { mn = st->vec[k]; any = 1; }
      o[0] = (float)mn;
      o[1] = (float)E->s.param_f[a * PHX_NPF + 0];
      break;
    }
    case PHX_KIND_MOCK_STRAT:                                 /* tests/__init__.py:49-51 */
      ((ostate*)st)->i[0] += 1;                               /* encode_obs_count */
      o[0] = (float)((double)e->step / (double)E->s.num_steps);   /* views.py:33-34, env.py:168 */
      break;
    case PHX_KIND_ADVERTISER: {                               /* digital_ads_market.py:294-316 */
      if (st->i[ADV_USER] == 0) return 0;                     /* `if self._current_user_id != 0` else None */
      tval budget = adv_budget(E, e, a);
      o[0] = (float)budget.v;                                 /* "type": np.array([budget], float32) supertype.py:68-69 */
      o[1] = (float)t_div(tv(st->f[0], st->i[ADV_LEFT_TAG]), budget).v;   /* "budget_left" (an f64 array there) */
      o[2] = (float)(st->i[ADV_USER] - 1);                    /* "user_id" */
      break;
    }
    default: break;
  }
  return 1;
}

static double agent_compute_reward(const phxo_env* E, oenv* e, int a) {
  ostate* st = &e->ag[a];
  switch (E->s.kind[a]) {
    case PHX_KIND_SHOP: {                                     /* supply_chain.py:144-147 */
      /* typed shop: self.type.excess_stock_weight * self.stock  docs/user/tutorial2.rst:270-273 */
      const double w = agent_is_typed(E, a) ? agent_type_value(E, e, a) : 0.1;
      volatile double penalty = w * (double)st->i[0];         /* volatile: no fma contraction */
      return (double)st->i[1] - penalty;
    }
    case PHX_KIND_SELLER: return st->f[1];
    case PHX_KIND_BUYER:
      if (st->i[0]) return E->s.param_f[a * PHX_NPF + 0] - st->f[0];
      return 0.0;
    case PHX_KIND_MOCK_STRAT: st->i[2] += 1; return 0.0;      /* tests/__init__.py:57-59 */
    case PHX_KIND_ADVERTISER:                                 /* digital_ads_market.py:335-343, risk_aversion = 0:
                                                                 1.0 * step_clicks + (0.0 * left) / budget       */
      return (double)st->i[ADV_CLICKS];
    default: return 0.0;
  }
}
static int agent_is_terminated(const phxo_env* E, const oenv* e, int a) {
  if (E->s.kind[a] == PHX_KIND_MOCK_STRAT)                    /* tests/__init__.py:61-62 */
    return e->step == E->s.param_i[a * PHX_NPI + 0];
  if (E->s.kind[a] == PHX_KIND_ADVERTISER)                    /* digital_ads_market.py:345-349 */
    return e->ag[a].f[0] <= 0.0;
  return 0;                                                   /* agents.py:292-307 */
}
static int agent_is_truncated(const phxo_env* E, const oenv* e, int a) {
  if (E->s.kind[a] == PHX_KIND_MOCK_STRAT)                    /* tests/__init__.py:64-65 */
    return e->step == E->s.param_i[a * PHX_NPI + 0];
  return 0;                                                   /* agents.py:309-323 */
}

/* AdExchangeAgent.handle_batch + auction  digital_ads_market.py:429-516: every Bid of the batch is
 * held back, the other messages are handled one by one, then ONE auction runs over the bids:
 * sorted(bids, key=bid, reverse=True) is stable, so the winner is the first maximal bid in arrival
 * order and sorted_bids[1] the first maximal one among the rest.                                  */
static void adexchange_handle_batch(const phxo_env* E, oenv* e, int a, const oinbox* proc, const uint8_t* ok,
                                    int clock0) {
  int nb = 0, w = -1, w2 = -1, k = 0;
  for (int id = proc->head[a]; id >= 0; id = proc->next[id], ++k) {
    if (!ok[id]) continue;
    const omsg* m = &proc->pool[id];
    if (m->type == PHX_MSG_BID) { nb++; continue; }
    agent_handle_message(E, e, a, m, clock0 + k);
  }
  if (!nb) return;
#define BIDV(id) tv(proc->pool[id].p.f, (proc->pool[id].aux >> 8) & 3)
  for (int id = proc->head[a]; id >= 0; id = proc->next[id])
    if (ok[id] && proc->pool[id].type == PHX_MSG_BID && (w < 0 || t_lt(BIDV(w), BIDV(id)))) w = id;
  for (int id = proc->head[a]; id >= 0; id = proc->next[id])
    if (ok[id] && proc->pool[id].type == PHX_MSG_BID && id != w && (w2 < 0 || t_lt(BIDV(w2), BIDV(id)))) w2 = id;
  const int second = E->s.param_i[a * PHX_NPI + 1];
  const omsg* win = &proc->pool[w];
  const omsg* costm = (second && w2 >= 0) ? &proc->pool[w2] : win;    /* :498-516 */
  omsg ads = mk_i(win->src); ads.aux = win->aux & 0xff;               /* Ads(advertiser_id, theme, user_id) :472-482 */
  network_send(E, e, a, E->s.param_i[a * PHX_NPI + 0], PHX_MSG_ADS, ads);
  for (int id = proc->head[a]; id >= 0; id = proc->next[id]) {        /* :484-492 */
    if (!ok[id] || proc->pool[id].type != PHX_MSG_BID) continue;
    omsg r;
    if (proc->pool[id].src == win->src) { r = mk_f(costm->p.f); r.aux = costm->aux & 0x300; }
    else { r = mk_f(0.0); r.aux = PHX_TAG_PYF << 8; }
    network_send(E, e, a, proc->pool[id].src, PHX_MSG_AUCTION_RESULT, r);
  }
#undef BIDV
}

/* ---- resolver ---------------------------------------------------------------------------- */

/* shuffle_batches on the device stream (phx_dev.h: rng_shuffle_block): block `blk` of the Fisher-Yates draws of
 * (env, tick, round, receiver); draw d uses word d & 3 of block d >> 2, j = (word * (i + 1)) >> 32 */
static void shuffle_block(const phxo_env* E, const oenv* e, int round, int receiver, uint32_t blk, uint32_t w[4]) {
  uint32_t key[2] = {(uint32_t)E->s.seed, (uint32_t)(E->s.seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)e->genv, (uint32_t)((uint64_t)e->genv >> 32) | (blk << 16), e->tick,
                     0x10000000u | (((uint32_t)round & 0xfffu) << 16) | (uint32_t)receiver};
  phxo_philox4x32_10(ctr, key, w);
}

/* BatchResolver.resolve resolvers.py:128-163; `live` = receiver_id in contexts */
static void batch_resolve(const phxo_env* E, oenv* e, const uint8_t* live) {
  e->round = 0; e->shuf_pos = 0;
  uint8_t* ok = (uint8_t*)alloca(E->s.queue_cap > 0 ? E->s.queue_cap : 1);   /* exchange batches: delivered flags */
  for (int i = 0;; ++i) {
    if (E->s.round_limit >= 0 && i >= E->s.round_limit) break;        /* range(round_limit) :129-131 */
    if (i >= PHX_MAX_ROUNDS) break;                                   /* itertools.count(): build-specific safety cap */
    oinbox* proc = &e->box[e->cur];
    if (proc->n_recv == 0) break;                                     /* :134-135 */
    e->cur ^= 1;                                                      /* self.messages = defaultdict(list) :139-140 */
    inbox_clear(E, &e->box[e->cur]);
    e->round = i + 1;
    for (int r = 0; r < proc->n_recv; ++r) {                          /* dict order :142 */
      int receiver = proc->order[r];
      if (E->s.kind[receiver] == PHX_KIND_ADEXCHANGE) {               /* overrides handle_batch */
        const int clock0 = e->clock;
        for (int id = proc->head[receiver]; id >= 0; id = proc->next[id]) {
          e->clock++;
          ok[id] = live[receiver] && has_edge(E, e, proc->pool[id].src, proc->pool[id].dst);
        }
        if (live[receiver]) adexchange_handle_batch(E, e, receiver, proc, ok, clock0);
        continue;
      }
      if (E->s.flags & PHX_F_SHUFFLE_BATCHES) {                       /* np.random.shuffle(msgs) :150-151 */
        /* every queued message advances the logical clock and the replay stream's position; a receiver
         * without context is skipped before the shuffle (:143-144), the edge filter (:146-148) cannot drop
         * anything (shuffle_batches is rejected together with ignore_connection_errors) */
        int c = 0;
        for (int id = proc->head[receiver]; id >= 0; id = proc->next[id]) ++c;
        const int clock0 = e->clock, pos0 = e->shuf_pos;
        e->clock += c; e->shuf_pos += c;
        if (!live[receiver]) continue;
        int* ids = (int*)malloc(sizeof(int) * (size_t)(c > 0 ? c : 1));
        int* tmp = (int*)malloc(sizeof(int) * (size_t)(c > 0 ? c : 1));
        int k = 0;
        for (int id = proc->head[receiver]; id >= 0; id = proc->next[id]) ids[k++] = id;
        if (c > 1) {
          if (e->shuffle_b) {                                         /* replayed permutation */
            const int fits = pos0 + c <= 8 * E->s.queue_cap;
            for (k = 0; k < c; ++k) { int j = fits ? e->shuffle_b[pos0 + k] : k; tmp[k] = ids[j < c ? j : k]; }
            memcpy(ids, tmp, sizeof(int) * (size_t)c);
          } else {                                                    /* Fisher-Yates on the device stream */
            uint32_t w[4]; int have = -1;
            for (int ii = c - 1, d = 0; ii >= 1; --ii, ++d) {
              if ((d >> 2) != have) { have = d >> 2; shuffle_block(E, e, i, receiver, (uint32_t)have, w); }
              int j = (int)(((uint64_t)w[d & 3] * (uint64_t)(ii + 1)) >> 32);
              int v = ids[ii]; ids[ii] = ids[j]; ids[j] = v;
            }
          }
        }
        for (k = 0; k < c; ++k) agent_handle_message(E, e, receiver, &proc->pool[ids[k]], clock0 + k);
        free(ids); free(tmp);
        continue;
      }
      for (int id = proc->head[receiver]; id >= 0; id = proc->next[id]) {
        /* logical time = position of the message in the processing order, counting every
         * queued message whether it is handled or dropped (stands in for time.time()) */
        int clock = e->clock++;
        if (!live[receiver]) continue;                                /* :143-144 */
        const omsg* m = &proc->pool[id];
        if (!has_edge(E, e, m->src, m->dst)) continue;                /* :146-148 */
        agent_handle_message(E, e, receiver, m, clock);               /* handle_batch agents.py:96-120 */
      }
    }
  }
  if (e->box[e->cur].n_recv > 0) set_err(e, PHX_ERR_ROUND_LIMIT);     /* :160-163 */
  inbox_clear(E, &e->box[e->cur]);                                    /* Network.resolve -> resolver.reset network.py:265 */
}

/* PhantomEnv.is_terminated / is_truncated env.py:308-318 */
static int env_is_terminated(const phxo_env* E, const oenv* e) {
  int n = 0; for (int s = 0; s < E->S; ++s) n += e->term[s];
  return n == E->S;
}
static int env_is_truncated(const phxo_env* E, const oenv* e) {
  int n = 0; for (int s = 0; s < E->S; ++s) n += e->trunc[s];
  return (e->step == E->s.num_steps) || n == E->S;
}

static void env_reset_one(const phxo_env* E, oenv* e, int b, const double* sampler_values,
                          const uint8_t* conn, float* obs, uint8_t* obs_valid) {
  /* PhantomEnv.reset env.py:185-237; fsm.py:195-251; stackelberg.py:53-109 */
  e->step = 0;
  env_sample(E, e, b, sampler_values, conn);                          /* env.py:211-218 */
  if (E->s.env_type == PHX_ENV_FSM) e->stage = E->s.initial_stage;    /* fsm.py:217 */
  inbox_clear(E, &e->box[0]); inbox_clear(E, &e->box[1]); e->cur = 0; /* network.reset -> resolver.reset */
  for (int a = 0; a < E->A; ++a) agent_reset(E, e, a);                /* network.py:183-184 */
  memset(e->term, 0, E->S); memset(e->trunc, 0, E->S);                /* env.py:223-224 */
  e->err = 0;
  if (E->s.env_type != PHX_ENV_PLAIN)                                 /* fsm.py:234 / stackelberg.py:92 */
    memset(e->rew_cache_valid, 0, E->S);
  if (obs_valid) memset(obs_valid, 0, E->S);
  if (obs) memset(obs, 0, sizeof(float) * E->S * E->D);
  if (!obs) return;
  if (E->s.env_type == PHX_ENV_PLAIN) {                               /* env.py:227-237 */
    for (int s = 0; s < E->S; ++s) {
      int v = agent_encode_obs(E, e, E->strat_idx[s], obs + s * E->D);
      if (obs_valid) obs_valid[s] = (uint8_t)v;                       /* `if v is not None` env.py:237 */
    }
  } else if (E->s.env_type == PHX_ENV_FSM) {                          /* fsm.py:237-251 */
    int st = e->stage;
    for (int k = E->s.stage_act_ptr[st]; k < E->s.stage_act_ptr[st + 1]; ++k) {
      int a = E->s.stage_act_idx[k], s = E->strat_rank[a];
      if (s < 0) continue;
      int v = agent_encode_obs(E, e, a, obs + s * E->D);
      if (obs_valid) obs_valid[s] = (uint8_t)v;
    }
  } else {                                                            /* stackelberg.py:95-109 */
    for (int k = 0; k < E->s.n_leaders; ++k) {
      int a = E->s.leaders[k], s = E->strat_rank[a];
      if (s < 0) continue;
      int v = agent_encode_obs(E, e, a, obs + s * E->D);
      if (obs_valid) obs_valid[s] = (uint8_t)v;
    }
  }
}

static int in_list(const int32_t* lst, int n, int a) {
  for (int k = 0; k < n; ++k) if (lst[k] == a) return 1;
  return 0;
}

/* one env, one step.  out pointers address THIS env's rows. */
static void env_step_one(const phxo_env* E, oenv* e, int b, const float* actions,
                         const uint8_t* action_valid, const uint8_t* exo_b, float* obs,
                         uint8_t* obs_valid, double* reward, uint8_t* reward_valid,
                         uint8_t* terminated, uint8_t* truncated, uint8_t* done_valid,
                         uint8_t* all_term, uint8_t* all_trunc) {
  const int A = E->A, S = E->S, D = E->D;
  int next_in = e->next_in; e->next_in = -1;                    /* a stage handler's return value, this step only */
  const int phase = e->phase; e->phase = 0;
  e->step += 1;                                                       /* env.py:252 */
  e->exo_b = exo_b;

  /* _make_ctxs env.py:338-348: contexts exist for agents that are not done */
  uint8_t* live = (uint8_t*)alloca(A);
  for (int a = 0; a < A; ++a) {
    int s = E->strat_rank[a];
    live[a] = (s < 0) ? 1 : !(e->term[s] || e->trunc[s]);
  }
  uint8_t* live0 = (uint8_t*)alloca(S);
  for (int s = 0; s < S; ++s) live0[s] = live[E->strat_idx[s]];

  /* which agents act, in which order */
  const int32_t* act_list = NULL; int n_act = 0;
  const int32_t* next_act_list = NULL; int n_next_act = 0;
  int cur_stage = 0, next_stage = 0;
  if (E->s.env_type == PHX_ENV_FSM) {                                 /* fsm.py:276-277 */
    cur_stage = e->stage;
    act_list = E->s.stage_act_idx + E->s.stage_act_ptr[cur_stage];
    n_act = E->s.stage_act_ptr[cur_stage + 1] - E->s.stage_act_ptr[cur_stage];
  } else if (E->s.env_type == PHX_ENV_STACKELBERG) {                  /* stackelberg.py:133-137 */
    if (e->step % 2 == 1) { act_list = E->s.leaders; n_act = E->s.n_leaders;
                            next_act_list = E->s.followers; n_next_act = E->s.n_followers; }
    else { act_list = E->s.followers; n_act = E->s.n_followers;
           next_act_list = E->s.leaders; n_next_act = E->s.n_leaders; }
  }

  /* injected Network.send calls made before the step are already in the inbox */
  /* _handle_acting_agents env.py:320-336 */
  int n_iter = (E->s.env_type == PHX_ENV_PLAIN) ? A : n_act;
  if (phase == 2) n_iter = 0;                                         /* phxo_step_end: acting and resolution happened in phxo_step_begin */
  for (int k = 0; k < n_iter; ++k) {
    int a = (E->s.env_type == PHX_ENV_PLAIN) ? k : act_list[k];
    if (!live[a]) continue;                                           /* :324-325 */
    int s = E->strat_rank[a];
    int has_action = (s >= 0) && actions && (!action_valid || action_valid[s]);   /* aid in actions :330 */
    agent_act(E, e, b, a, has_action, has_action ? actions[s] : 0.0f, exo_b);
  }

  /* resolve_network env.py:180-183 */
  if (phase != 2) {
    for (int a = 0; a < A; ++a) if (live[a]) agent_pre_resolution(E, e, a);   /* :170-173 */
    batch_resolve(E, e, live);                                                 /* network.py:256-265 */
  }
  /* post_message_resolution: no kind overrides it (agents.py:93-94) */
  if (phase == 1) { e->step -= 1; return; }                           /* the host's stage handler runs now, on the resolved state (fsm.py:294-302) */

  if (E->s.env_type == PHX_ENV_FSM) {
    next_stage = E->s.stage_next[cur_stage];                          /* no handler: next_stages[0]  fsm.py:281-292 */
    /* a handler that branches on the RESOLVED agent state, declared as rules (phx_spec.stage_rules, ABI 9): what env_handler()
     * returns now, fsm.py:294-302 -- the first rule of the stage whose condition holds */
    if (next_in == -1 && E->s.n_stage_rules > 0) {
      for (int r = 0; r < E->s.n_stage_rules && next_in == -1; ++r) {
        const phx_stage_rule* q = &E->s.stage_rules[r];
        if (q->stage != cur_stage) continue;
        char nm[33]; memcpy(nm, q->field, 32); nm[32] = 0;
        int fk, fis, fslot;
        if (!rule_field(nm, &fk, &fis, &fslot)) continue;
        /* the value: one agent's field, or the sum over the kind's agents taken as the device takes it (64 strided partial sums,
         * butterfly reduction: exact for i32 fields, the device's rounding for f64 fields) */
        double part[64]; for (int l = 0; l < 64; ++l) part[l] = 0.0;
        double v = 0.0;
        for (int a = 0; a < A; ++a) {
          if (E->s.kind[a] != fk) continue;
          const int c = E->kind_rank[a];
          const double x = fis ? e->ag[a].f[fslot] : (double)e->ag[a].i[fslot];
          if (q->agent >= 0) { if (c == q->agent) v = x; } else part[c & 63] += x;    /* (columns c, c + 64, .. of a lane: ascending a == ascending c) */
        }
        if (q->agent < 0) {
          for (int off = 1; off < 64; off <<= 1) { double nx[64]; for (int l = 0; l < 64; ++l) nx[l] = (l & off) ? part[l ^ off] + part[l] : part[l] + part[l ^ off]; memcpy(part, nx, sizeof nx); }
          v = part[0];
        }
        const double th = q->threshold;
        const int hit = q->cmp == PHX_CMP_LT ? v < th : q->cmp == PHX_CMP_LE ? v <= th : q->cmp == PHX_CMP_GT ? v > th :
                        q->cmp == PHX_CMP_GE ? v >= th : q->cmp == PHX_CMP_EQ ? v == th : v != th;
        if (hit) next_in = q->next_stage;
      }
    }
    /* a handler that decides from (stage, clock) alone, tabulated by the host: what env_handler() returns now, :294-302 */
    if (next_in == -1 && E->s.stage_tab && e->step >= 0 && e->step <= E->s.num_steps)
      next_in = E->s.stage_tab[(size_t)cur_stage * (E->s.num_steps + 1) + e->step];
    if (next_in >= 0 || next_in == -2) {                              /* env_handler() returned a stage  :294-302 */
      int ok = next_in >= 0 && next_in < E->s.n_stages &&
               (E->s.stage_allowed ? E->s.stage_allowed[(size_t)cur_stage * E->s.n_stages + next_in] != 0
                                   : next_in == E->s.stage_next[cur_stage]);
      if (ok) next_stage = next_in;
      else set_err(e, PHX_ERR_FSM_TRANSITION);                        /* FSMRuntimeError :304-307 */
    }
  }

  memset(obs_valid, 0, S); memset(reward_valid, 0, S); memset(done_valid, 0, S);
  memset(terminated, 0, S); memset(truncated, 0, S);
  memset(obs, 0, sizeof(float) * S * D);
  for (int s = 0; s < S; ++s) reward[s] = 0.0;
  uint8_t* observed = (uint8_t*)alloca(S); memset(observed, 0, S);

  for (int s = 0; s < S; ++s) {                                       /* env.py:273 / fsm.py:320 / stackelberg.py:150 */
    int a = E->strat_idx[s];
    if (!live0[s]) continue;                                          /* env.py:274-275 */
    int do_obs, do_rew;
    if (E->s.env_type == PHX_ENV_PLAIN) { do_obs = 1; do_rew = 1; }
    else if (E->s.env_type == PHX_ENV_FSM) {
      if (E->s.stage_rewarded_all[cur_stage]) { do_obs = 1; do_rew = 1; }     /* fsm.py:315-317 */
      else {
        do_rew = E->s.stage_rewarded[cur_stage * A + a];                      /* :319 */
        do_obs = in_list(E->s.stage_act_idx + E->s.stage_act_ptr[next_stage], /* :320 */
                         E->s.stage_act_ptr[next_stage + 1] - E->s.stage_act_ptr[next_stage], a);
      }
    } else {
      do_obs = in_list(next_act_list, n_next_act, a);                 /* stackelberg.py:156 */
      do_rew = in_list(act_list, n_act, a);                           /* :162 */
    }
    if (do_obs)                                                       /* `if obs is not None` env.py:279-280 */
      observed[s] = (uint8_t)agent_encode_obs(E, e, a, obs + s * D);
    if (E->s.env_type == PHX_ENV_PLAIN) {
      if (observed[s]) {
        reward[s] = agent_compute_reward(E, e, a);                    /* env.py:283 */
        reward_valid[s] = 1;
      }
    } else if (do_rew) {
      e->rew_cache[s] = agent_compute_reward(E, e, a);                /* fsm.py:335,350 / stackelberg.py:163 */
      e->rew_cache_valid[s] = 1;
    }
    terminated[s] = (uint8_t)agent_is_terminated(E, e, a);            /* env.py:285-286 */
    truncated[s] = (uint8_t)agent_is_truncated(E, e, a);
    done_valid[s] = 1;
    if (terminated[s]) e->term[s] = 1;                                /* :288-292 */
    if (truncated[s]) e->trunc[s] = 1;
  }

  if (E->s.env_type == PHX_ENV_FSM) {
    for (int s = 0; s < S; ++s) if (observed[s]) {                    /* self._observations.update fsm.py:349 */
      memcpy(e->obs_cache + s * D, obs + s * D, sizeof(float) * D);
      e->obs_cache_valid[s] = 1;
    }
    e->prev_stage = cur_stage; e->stage = next_stage;                 /* fsm.py:355 */
  }

  *all_term = (uint8_t)env_is_terminated(E, e);                       /* env.py:297 */
  *all_trunc = (uint8_t)env_is_truncated(E, e);                       /* env.py:298 */
  int terminal = *all_term || *all_trunc;

  if (E->s.env_type == PHX_ENV_PLAIN) {
    for (int s = 0; s < S; ++s) obs_valid[s] = observed[s];
  } else if (E->s.env_type == PHX_ENV_FSM) {
    if (terminal) {                                                   /* fsm.py:360-375: cached dicts of ALL agents */
      for (int s = 0; s < S; ++s) {
        obs_valid[s] = e->obs_cache_valid[s];
        if (obs_valid[s]) memcpy(obs + s * D, e->obs_cache + s * D, sizeof(float) * D);
        else memset(obs + s * D, 0, sizeof(float) * D);
        reward_valid[s] = e->rew_cache_valid[s] ? 1 : 2;
        reward[s] = e->rew_cache_valid[s] ? e->rew_cache[s] : 0.0;
      }
    } else {                                                          /* fsm.py:378 */
      for (int s = 0; s < S; ++s) {
        obs_valid[s] = observed[s];
        if (observed[s]) {
          reward_valid[s] = e->rew_cache_valid[s] ? 1 : 2;
          reward[s] = e->rew_cache_valid[s] ? e->rew_cache[s] : 0.0;
        }
      }
    }
  } else {
    for (int s = 0; s < S; ++s) obs_valid[s] = observed[s];
    if (terminal) {                                                   /* stackelberg.py:180-187 */
      for (int s = 0; s < S; ++s) {
        reward_valid[s] = e->rew_cache_valid[s] ? 1 : 2;
        reward[s] = e->rew_cache_valid[s] ? e->rew_cache[s] : 0.0;
      }
    } else {                                                          /* :190-194 */
      for (int s = 0; s < S; ++s)
        if (observed[s] && e->rew_cache_valid[s]) { reward_valid[s] = 1; reward[s] = e->rew_cache[s]; }
    }
  }
  e->tick += 1;
}

/* ---- construction -------------------------------------------------------------------------- */
static void* dup_arr(const void* p, size_t n) {
  if (!p || !n) return NULL;
  void* q = malloc(n); memcpy(q, p, n); return q;
}

static int inbox_alloc(const phxo_env* E, oinbox* b) {
  int cap = E->s.queue_cap > 0 ? E->s.queue_cap : 1;
  b->pool = (omsg*)calloc(cap, sizeof(omsg));
  b->next = (int*)calloc(cap, sizeof(int));
  b->order = (int*)calloc(E->A, sizeof(int));
  b->head = (int*)calloc(E->A, sizeof(int));
  b->tail = (int*)calloc(E->A, sizeof(int));
  return b->pool && b->next && b->order && b->head && b->tail;
}

phxo_env* phxo_create(const phx_spec* sp) {
  if (!sp || sp->abi_version != PHX_ABI_VERSION || sp->n_agents <= 0 || sp->batch <= 0) {
    snprintf(g_err, sizeof g_err, "bad spec"); return NULL;
  }
  if (sp->env_type == PHX_ENV_FSM && sp->n_stage_rules > 0 && sp->stage_rules) {
    /* the same validation phx_create makes (phx_api.hip): an unknown field name, a comparison code outside PHX_CMP_*, an agent column
     * outside the kind or a stage outside the env are errors of the spec, not rules to be skipped */
    for (int r = 0; r < sp->n_stage_rules; ++r) {
      const phx_stage_rule* q = &sp->stage_rules[r];
      char nm[33]; memcpy(nm, q->field, 32); nm[32] = 0;
      int fk = 0, fis, fslot, ncols = 0;
      const int known = rule_field(nm, &fk, &fis, &fslot);
      if (known) for (int a = 0; a < sp->n_agents; ++a) ncols += sp->kind[a] == fk;
      if (!known || fk <= 0 || ncols < 1 || q->agent < -1 || q->agent >= ncols || q->cmp < PHX_CMP_LT || q->cmp > PHX_CMP_NE ||
          q->stage < 0 || q->stage >= sp->n_stages || q->next_stage < 0 || q->next_stage >= sp->n_stages) {
        snprintf(g_err, sizeof g_err, "stage_rules[%d]: '%s' is not a per-agent state field of this env, or a code / column / stage is out of range", r, nm);
        return NULL;
      }
    }
  }
  phxo_env* E = (phxo_env*)calloc(1, sizeof *E);
  E->s = *sp; E->A = sp->n_agents; E->B = sp->batch;
  const int A = E->A;
  E->nnz = sp->row_ptr[A];
  E->s.kind = (const uint8_t*)dup_arr(sp->kind, A);
  E->s.param_i = (const int32_t*)dup_arr(sp->param_i, sizeof(int32_t) * A * PHX_NPI);
  E->s.param_f = (const double*)dup_arr(sp->param_f, sizeof(double) * A * PHX_NPF);
  E->s.row_ptr = (const int32_t*)dup_arr(sp->row_ptr, sizeof(int32_t) * (A + 1));
  E->s.col = (const int32_t*)dup_arr(sp->col, sizeof(int32_t) * (E->nnz ? E->nnz : 1));
  if (sp->env_type == PHX_ENV_FSM) {
    int ns = sp->n_stages;
    E->s.stage_act_ptr = (const int32_t*)dup_arr(sp->stage_act_ptr, sizeof(int32_t) * (ns + 1));
    int na = sp->stage_act_ptr[ns];
    E->s.stage_act_idx = (const int32_t*)dup_arr(sp->stage_act_idx, sizeof(int32_t) * (na ? na : 1));
    E->s.stage_rewarded = (const uint8_t*)dup_arr(sp->stage_rewarded, (size_t)ns * A);
    E->s.stage_rewarded_all = (const uint8_t*)dup_arr(sp->stage_rewarded_all, ns);
    E->s.stage_next = (const int32_t*)dup_arr(sp->stage_next, sizeof(int32_t) * ns);
    E->s.stage_allowed = sp->stage_allowed ? (const uint8_t*)dup_arr(sp->stage_allowed, (size_t)ns * ns) : NULL;
    E->s.stage_tab = sp->stage_tab ? (const int32_t*)dup_arr(sp->stage_tab, sizeof(int32_t) * (size_t)ns * (sp->num_steps + 1)) : NULL;
    E->s.stage_rules = (sp->n_stage_rules > 0 && sp->stage_rules) ? (const phx_stage_rule*)dup_arr(sp->stage_rules, sizeof(phx_stage_rule) * (size_t)sp->n_stage_rules) : NULL;
    if (!E->s.stage_rules) E->s.n_stage_rules = 0;
  } else { E->s.stage_allowed = NULL; E->s.stage_tab = NULL; E->s.stage_rules = NULL; E->s.n_stage_rules = 0; }
  if (sp->env_type == PHX_ENV_STACKELBERG) {
    E->s.leaders = (const int32_t*)dup_arr(sp->leaders, sizeof(int32_t) * (sp->n_leaders ? sp->n_leaders : 1));
    E->s.followers = (const int32_t*)dup_arr(sp->followers, sizeof(int32_t) * (sp->n_followers ? sp->n_followers : 1));
  }
  E->s.sampler_kind = (const int32_t*)dup_arr(sp->sampler_kind, sizeof(int32_t) * sp->n_samplers);
  E->s.sampler_param = (const double*)dup_arr(sp->sampler_param, sizeof(double) * 4 * sp->n_samplers);
  E->s.type_src = (const int32_t*)dup_arr(sp->type_src, sizeof(int32_t) * A);
  E->s.conn_rate = (const double*)dup_arr(sp->conn_rate, sizeof(double) * sp->n_conn);
  E->s.col_conn = (const int32_t*)dup_arr(sp->col_conn, sizeof(int32_t) * (sp->n_conn ? E->nnz : 0));
  E->strat_rank = (int*)calloc(A, sizeof(int));
  E->kind_rank = (int*)calloc(A, sizeof(int));
  E->exo_rank = (int*)calloc(A, sizeof(int));
  E->strat_idx = (int*)calloc(A, sizeof(int));
  int S = 0, D = 0, nx = 0;
  for (int a = 0; a < A; ++a) {
    int k = E->s.kind[a];
    if (k <= 0 || k >= PHX_KIND_COUNT) { snprintf(g_err, sizeof g_err, "bad kind"); return NULL; }
    E->kind_rank[a] = E->kind_count[k]++;
    if (is_strategic_kind(k)) {                                       /* isinstance(a, StrategicAgent) env.py:151-159 */
      E->strat_rank[a] = S; E->strat_idx[S++] = a;
      if (obs_dim_of_kind(k) > D) D = obs_dim_of_kind(k);
      if (k == PHX_KIND_SHOP && sp->type_src && sp->type_src[a] != PHX_TYPE_NONE && D < 4) D = 4;
    } else E->strat_rank[a] = -1;
    /* exogenous draws per step: a customer's order; a publisher's user id + pi1 click draws */
    if (k == PHX_KIND_CUSTOMER) E->exo_rank[a] = nx++;
    else if (k == PHX_KIND_PUBLISHER) { E->exo_rank[a] = nx; nx += 1 + sp->param_i[a * PHX_NPI + 1]; }
    else E->exo_rank[a] = -1;
  }
  E->S = S; E->D = D > 0 ? D : 1; E->n_exo = nx;
  E->env = (oenv*)calloc(E->B, sizeof(oenv));
  for (int b = 0; b < E->B; ++b) {
    oenv* e = &E->env[b];
    e->ag = (ostate*)calloc(A, sizeof(ostate));
    e->vecpool = (double*)calloc(E->nnz ? E->nnz : 1, sizeof(double));
    for (int a = 0; a < A; ++a) e->ag[a].vec = e->vecpool + E->s.row_ptr[a];
    e->term = (uint8_t*)calloc(S ? S : 1, 1); e->trunc = (uint8_t*)calloc(S ? S : 1, 1);
    e->rew_cache = (double*)calloc(S ? S : 1, sizeof(double));
    e->rew_cache_valid = (uint8_t*)calloc(S ? S : 1, 1);
    e->obs_cache = (float*)calloc((size_t)(S ? S : 1) * E->D, sizeof(float));
    e->obs_cache_valid = (uint8_t*)calloc(S ? S : 1, 1);
    if (!inbox_alloc(E, &e->box[0]) || !inbox_alloc(E, &e->box[1])) return NULL;
    inbox_clear(E, &e->box[0]); inbox_clear(E, &e->box[1]);
    e->stage = sp->initial_stage; e->prev_stage = -1; e->genv = sp->env_offset + b;
    e->sampler = (double*)calloc(sp->n_samplers > 0 ? sp->n_samplers : 1, sizeof(double));
    e->conn_on = (uint8_t*)calloc(sp->n_conn > 0 ? sp->n_conn : 1, 1);
    env_sample(E, e, b, NULL, NULL);                                  /* env.py:118-119; add_connection network.py:389-391 */
    for (int a = 0; a < A; ++a) agent_reset(E, e, a);                 /* env.py:122-124 */
  }
  return E;
}

void phxo_destroy(phxo_env* E) {
  if (!E) return;
  for (int b = 0; b < E->B; ++b) {
    oenv* e = &E->env[b];
    free(e->ag); free(e->vecpool); free(e->term); free(e->trunc); free(e->rew_cache);
    free(e->rew_cache_valid); free(e->obs_cache); free(e->obs_cache_valid); free(e->sampler); free(e->conn_on);
    for (int k = 0; k < 2; ++k) {
      free(e->box[k].pool); free(e->box[k].next); free(e->box[k].order);
      free(e->box[k].head); free(e->box[k].tail);
    }
  }
  free(E->env); free(E->strat_rank); free(E->kind_rank); free(E->exo_rank); free(E->strat_idx);
  free(E->injected);
  free((void*)E->s.kind); free((void*)E->s.param_i); free((void*)E->s.param_f);
  free((void*)E->s.row_ptr); free((void*)E->s.col);
  free((void*)E->s.sampler_kind); free((void*)E->s.sampler_param); free((void*)E->s.type_src);
  free((void*)E->s.conn_rate); free((void*)E->s.col_conn);
  if (E->s.env_type == PHX_ENV_FSM) {
    free((void*)E->s.stage_act_ptr); free((void*)E->s.stage_act_idx);
    free((void*)E->s.stage_rewarded); free((void*)E->s.stage_rewarded_all); free((void*)E->s.stage_next); free((void*)E->s.stage_tab); free((void*)E->s.stage_rules);
  }
  if (E->s.env_type == PHX_ENV_STACKELBERG) { free((void*)E->s.leaders); free((void*)E->s.followers); }
  free(E);
}

int phxo_obs_dim(const phxo_env* E) { return E->D; }
int phxo_n_strategic(const phxo_env* E) { return E->S; }
int phxo_n_exo(const phxo_env* E) { return E->n_exo; }

/* ---- batch entry points ---------------------------------------------------------------------- */
void phxo_reset(phxo_env* E, const uint8_t* mask, const double* sampler_values, const uint8_t* conn_on,
                float* obs, uint8_t* obs_valid) {
  const int S = E->S, D = E->D;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int b = 0; b < E->B; ++b) {
    if (mask && !mask[b]) continue;
    env_reset_one(E, &E->env[b], b, sampler_values ? sampler_values + (size_t)b * E->s.n_samplers : NULL,
                  conn_on ? conn_on + (size_t)b * E->s.n_conn : NULL, obs ? obs + (size_t)b * S * D : NULL, obs_valid ? obs_valid + (size_t)b * S : NULL);
  }
}

static void apply_injected(const phxo_env* E, oenv* e) {
  for (int k = 0; k < E->n_injected; ++k) {
    const omsg* m = &E->injected[k];
    network_send(E, e, m->src, m->dst, m->type, *m);                  /* n.send(...) from test code */
  }
}

void phxo_inject(phxo_env* E, const phx_msg_rec* msgs, int n) {
  E->injected = (omsg*)realloc(E->injected, sizeof(omsg) * (E->n_injected + n + 1));
  for (int k = 0; k < n; ++k) {
    omsg m; memset(&m, 0, sizeof m);
    m.src = msgs[k].sender; m.dst = msgs[k].receiver; m.type = msgs[k].type; m.p.i = msgs[k].payload.i;
    m.aux = msgs[k].round;                                            /* on input `round` carries the aux bits */
    E->injected[E->n_injected++] = m;
  }
}

static void phxo_step_phase(phxo_env* E, const phx_step_io* io, int phase);
void phxo_step(phxo_env* E, const phx_step_io* io) { phxo_step_phase(E, io, 0); }
/* the two halves of a step around a host-side stage handler that reads agent state (fsm.py:275-307): begin = the acting phase and
 * resolve_network(); end = the transition to io->next_stage and everything after it.  begin + end == phxo_step.             */
void phxo_step_begin(phxo_env* E, const phx_step_io* io) { phxo_step_phase(E, io, 1); }
void phxo_step_end(phxo_env* E, const phx_step_io* io) { phxo_step_phase(E, io, 2); }

static void phxo_step_phase(phxo_env* E, const phx_step_io* io, int phase) {
  const int S = E->S, D = E->D;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int b = 0; b < E->B; ++b) {
    oenv* e = &E->env[b];
    e->log = (io->msg_log && phase != 2) ? io->msg_log + (size_t)b * E->s.trace_cap : NULL;
    e->log_cap = E->s.trace_cap;
    e->err = io->err ? io->err[b] : 0;
    e->log_n = 0; e->round = 0;
    e->shuffle_b = io->shuffle ? io->shuffle + (size_t)b * 8 * E->s.queue_cap : NULL;
    e->next_in = (io->next_stage && phase != 1) ? (io->next_stage[b] >= 0 ? io->next_stage[b] : -2) : -1;
    e->phase = phase;
    if (phase != 2) apply_injected(E, e);
    uint8_t at = 0, au = 0;
    env_step_one(E, e, b, io->actions ? io->actions + (size_t)b * S : NULL,
                 io->action_valid ? io->action_valid + (size_t)b * S : NULL,
                 io->exo ? io->exo + (size_t)b * E->n_exo : NULL,
                 io->obs + (size_t)b * S * D, io->obs_valid + (size_t)b * S,
                 io->reward + (size_t)b * S, io->reward_valid + (size_t)b * S,
                 io->terminated + (size_t)b * S, io->truncated + (size_t)b * S,
                 io->done_valid + (size_t)b * S, &at, &au);
    if (phase != 1) { io->all_terminated[b] = at; io->all_truncated[b] = au; }
    if (io->err) io->err[b] = e->err;
    if (io->msg_count && phase != 2) io->msg_count[b] = e->log_n;
  }
  if (phase != 2) E->n_injected = 0;
}

void phxo_resolve(phxo_env* E, int32_t* err, phx_msg_rec* msg_log, int32_t* msg_count) {
  uint8_t* live = (uint8_t*)malloc(E->A); memset(live, 1, E->A);     /* contexts for every agent */
  for (int b = 0; b < E->B; ++b) {
    oenv* e = &E->env[b];
    e->log = msg_log ? msg_log + (size_t)b * E->s.trace_cap : NULL;
    e->log_cap = E->s.trace_cap; e->log_n = 0; e->round = 0;
    e->err = err ? err[b] : 0; e->exo_b = NULL; e->shuffle_b = NULL;
    apply_injected(E, e);
    batch_resolve(E, e, live);
    if (err) err[b] = e->err;
    if (msg_count) msg_count[b] = e->log_n;
  }
  free(live);
  E->n_injected = 0;
}

/* random policy of a rollout (build-owned): the strategic agent's word of this tick (see the
 * device-RNG definition: its rank j in [0, 274877)) mapped onto the kind's action
 * space: ShopAgent j * 100 / 274877 (Box(0, SHOP_MAX_STOCK)), Seller price j / 274877, Buyer buys
 * iff j < 137438 (p = 1/2).                                                                          */
static float policy_action(const phxo_env* E, const oenv* e, int b, int s) {
  const int kind = E->s.kind[E->strat_idx[s]];
  const uint32_t j = phxo_rng_rank(E->s.seed, E->s.env_offset + b, e->tick, s);
  if (kind == PHX_KIND_SELLER || kind == PHX_KIND_ADVERTISER) return (float)j * (1.0f / 274877.0f);
  if (kind == PHX_KIND_BUYER) return j < 137438u ? 1.0f : 0.0f;
  return (float)j * (100.0f / 274877.0f);
}

/* phx_policy_mlp (include/phantom_amd.h): the device-evaluated policy of a rollout, restated term by term -- fmaf is C99's correctly
 * rounded fused multiply-add, so the value does not depend on the machine; the host copies of the weights are made by the test harness
 * (the struct's pointers are HOST pointers here).                                                                                      */
static float policy_mlp_action(const phx_policy_mlp* p, const float* x, int D) {
  float h[2][PHX_POLICY_MAX_WIDTH];
  const float* in = x; int n_in = D;
  for (int l = 0; l < p->n_hidden; ++l) {
    for (int i = 0; i < p->width[l]; ++i) {
      float c = p->b[l][i];
      for (int k = 0; k < n_in; ++k) c = fmaf(p->w[l][(size_t)i * n_in + k], in[k], c);
      h[l][i] = p->activation == PHX_ACT_HARD_TANH ? (c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c)) : (c > 0.0f ? c : 0.0f);
    }
    in = h[l]; n_in = p->width[l];
  }
  float y = p->b[p->n_hidden][0];
  for (int k = 0; k < n_in; ++k) y = fmaf(p->w[p->n_hidden][k], in[k], y);
  const float a = fmaf(p->out_scale, y, p->out_bias);
  return (a < p->out_lo ? p->out_lo : (a > p->out_hi ? p->out_hi : a)) + 0.0f;
}

/* rollout = the list-of-envs loop of utils/rllib/rollout.py:361-363, with the caller's
 * reset-after-num_steps folded in (auto-reset at the end of the terminal step).          */
static void rollout_one(phxo_env* E, const phx_rollout_io* io, int b) {
  const int S = E->S, D = E->D, B = E->B, T = io->T;
  oenv* e = &E->env[b];
  float* act = (float*)alloca(sizeof(float) * (S ? S : 1));
  float* o = (float*)alloca(sizeof(float) * (S ? S : 1) * D);
  double* rw = (double*)alloca(sizeof(double) * (S ? S : 1));
  uint8_t* u8 = (uint8_t*)alloca(5 * (S ? S : 1));
  e->log = NULL; e->err = io->err ? io->err[b] : 0; e->shuffle_b = NULL;
  if (io->policy)                                                     /* what the agents observe now: the policy's input at the fragment's first step */
    for (int s = 0; s < S; ++s) { for (int d = 0; d < D; ++d) o[s * D + d] = 0.0f; agent_encode_obs(E, e, E->strat_idx[s], o + s * D); }
  for (int t = 0; t < T; ++t) {
    if (io->msg_log) {   /* rollout.py:369-373: tracked_messages recorded per step, cleared before the next */
      e->log = io->msg_log + ((size_t)t * B + b) * E->s.trace_cap; e->log_cap = E->s.trace_cap;
      e->log_n = 0; e->round = 0;
    }
    for (int s = 0; s < S; ++s) {
      const int a = E->strat_idx[s];
      if (E->s.env_type == PHX_ENV_STACKELBERG) {
        /* only the side that acts this step takes an action (stackelberg.py:133-137); the
         * trajectory records 0 for the other side */
        const int odd = ((e->step + 1) & 1);
        const int acts = odd ? in_list(E->s.leaders, E->s.n_leaders, a) : in_list(E->s.followers, E->s.n_followers, a);
        act[s] = !acts ? 0.0f : io->actions ? io->actions[((size_t)t * B + b) * S + s]
                                            : policy_action(E, e, b, s);
      } else {
        act[s] = io->policy ? policy_mlp_action(io->policy, o + s * D, D)      /* compute_action on the previous observation, rollout.py:300-363 */
               : io->actions ? io->actions[((size_t)t * B + b) * S + s] : policy_action(E, e, b, s);
      }
    }
    uint8_t at = 0, au = 0;
    e->next_in = -1;
    env_step_one(E, e, b, act, NULL, io->exo ? io->exo + ((size_t)t * B + b) * E->n_exo : NULL,
                 o, u8, rw, u8 + S, u8 + 2 * S, u8 + 3 * S, u8 + 4 * S, &at, &au);
    size_t base = ((size_t)t * B + b) * S;
    for (int s = 0; s < S; ++s) {
      for (int d = 0; d < D; ++d) io->obs[(base + s) * D + d] = o[s * D + d];
      io->action_out[base + s] = act[s];
      io->reward[base + s] = (float)rw[s];
      io->terminated[base + s] = (uint8_t)(u8[2 * S + s] | at);
      io->truncated[base + s] = (uint8_t)(u8[3 * S + s] | au);
      if (io->obs_valid) io->obs_valid[base + s] = u8[s];
      if (io->reward_valid) io->reward_valid[base + s] = u8[S + s];
    }
    if (io->msg_count) io->msg_count[(size_t)t * B + b] = e->log_n;
    if (at || au) {                                                   /* caller's env.reset() */
      const int32_t first_err = e->err;                               /* io->err reports the rollout's first error */
      env_reset_one(E, e, b, NULL, NULL, o, u8);
      e->err = first_err;
    }
  }
  if (io->last_obs) memcpy(io->last_obs + (size_t)b * S * D, o, sizeof(float) * S * D);
  if (io->err) io->err[b] = e->err;
}

void phxo_rollout(phxo_env* E, const phx_rollout_io* io) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int b = 0; b < E->B; ++b) rollout_one(E, io, b);
}

/* ---- state read-back -------------------------------------------------------------------------- */
typedef struct { const char* name; int kind; int is_f; int slot; } ofield;
static const ofield FIELDS[] = {
  {"shop.stock", PHX_KIND_SHOP, 0, 0}, {"shop.sales", PHX_KIND_SHOP, 0, 1},
  {"shop.missed_sales", PHX_KIND_SHOP, 0, 2}, {"shop.delivered_stock", PHX_KIND_SHOP, 0, 3},
  {"seller.tx", PHX_KIND_SELLER, 0, 0}, {"seller.price", PHX_KIND_SELLER, 1, 0},
  {"seller.revenue", PHX_KIND_SELLER, 1, 1},
  {"buyer.bought", PHX_KIND_BUYER, 0, 0}, {"buyer.paid", PHX_KIND_BUYER, 1, 0},
  {"cashbox.total_cash", PHX_KIND_CASHBOX, 1, 0},
  {"adv.left", PHX_KIND_ADVERTISER, 1, 0}, {"adv.bid", PHX_KIND_ADVERTISER, 1, 1},
  {"adv.left_tag", PHX_KIND_ADVERTISER, 0, ADV_LEFT_TAG}, {"adv.bid_tag", PHX_KIND_ADVERTISER, 0, ADV_BID_TAG},
  {"adv.step_clicks", PHX_KIND_ADVERTISER, 0, ADV_CLICKS}, {"adv.step_wins", PHX_KIND_ADVERTISER, 0, ADV_WINS},
  {"adv.user", PHX_KIND_ADVERTISER, 0, ADV_USER},
  {"reqresp.req_time", PHX_KIND_REQRESP, 0, 0}, {"reqresp.res_time", PHX_KIND_REQRESP, 0, 1},
  {"mock.encode_obs_count", PHX_KIND_MOCK_STRAT, 0, 0},
  {"mock.decode_action_count", PHX_KIND_MOCK_STRAT, 0, 1},
  {"mock.compute_reward_count", PHX_KIND_MOCK_STRAT, 0, 2},
};
static const ofield* find_field(const char* n) {
  for (size_t k = 0; k < sizeof FIELDS / sizeof FIELDS[0]; ++k)
    if (!strcmp(FIELDS[k].name, n)) return &FIELDS[k];
  return NULL;
}
static int rule_field(const char* name, int* kind, int* is_f, int* slot) {
  const ofield* f = find_field(name);
  if (!f) return 0;
  *kind = f->kind; *is_f = f->is_f; *slot = f->slot;
  return 1;
}

int64_t phxo_get_i32(const phxo_env* E, const char* field, int32_t* out) {
  if (!strcmp(field, "env.step")) { for (int b = 0; b < E->B; ++b) out[b] = E->env[b].step; return E->B; }
  if (!strcmp(field, "env.stage")) { for (int b = 0; b < E->B; ++b) out[b] = E->env[b].stage; return E->B; }
  if (!strcmp(field, "env.prev_stage")) { for (int b = 0; b < E->B; ++b) out[b] = E->env[b].prev_stage; return E->B; }
  if (!strcmp(field, "env.tick")) { for (int b = 0; b < E->B; ++b) out[b] = (int32_t)E->env[b].tick; return E->B; }
  if (!strcmp(field, "env.episode")) { for (int b = 0; b < E->B; ++b) out[b] = E->env[b].episode; return E->B; }
  int tot = !strcmp(field, "adv.total_clicks") ? ADV_TOT_CLICKS : !strcmp(field, "adv.total_requests") ? ADV_TOT_REQUESTS
          : !strcmp(field, "adv.total_wins") ? ADV_TOT_WINS : -1;
  if (tot >= 0) {                                                     /* defaultdict(int) keyed by user id: [B][n][3] */
    int n = E->kind_count[PHX_KIND_ADVERTISER];
    for (int b = 0; b < E->B; ++b)
      for (int a = 0; a < E->A; ++a)
        if (E->s.kind[a] == PHX_KIND_ADVERTISER)
          for (int u = 0; u < 3; ++u) out[((size_t)b * n + E->kind_rank[a]) * 3 + u] = E->env[b].ag[a].i[tot + u];
    return (int64_t)E->B * n * 3;
  }
  const ofield* f = find_field(field);
  if (!f || f->is_f) return -1;
  int n = E->kind_count[f->kind];
  for (int b = 0; b < E->B; ++b)
    for (int a = 0; a < E->A; ++a)
      if (E->s.kind[a] == f->kind) out[(size_t)b * n + E->kind_rank[a]] = E->env[b].ag[a].i[f->slot];
  return (int64_t)E->B * n;
}
int64_t phxo_get_u8(const phxo_env* E, const char* field, uint8_t* out) {
  if (strcmp(field, "net.conn_on")) return -1;
  for (int b = 0; b < E->B; ++b) memcpy(out + (size_t)b * E->s.n_conn, E->env[b].conn_on, E->s.n_conn);
  return (int64_t)E->B * E->s.n_conn;
}
int64_t phxo_set_i32(phxo_env* E, const char* field, const int32_t* in) {
  if (!strcmp(field, "env.tick")) { for (int b = 0; b < E->B; ++b) E->env[b].tick = (uint32_t)in[b]; return E->B; }
  if (!strcmp(field, "env.step")) { for (int b = 0; b < E->B; ++b) E->env[b].step = in[b]; return E->B; }          /* tests: counters a caller moved */
  if (!strcmp(field, "env.episode")) { for (int b = 0; b < E->B; ++b) E->env[b].episode = in[b]; return E->B; }
  if (!strcmp(field, "env.stage")) { for (int b = 0; b < E->B; ++b) E->env[b].stage = in[b]; return E->B; }   /* tests: a stage off the default chain */
  if (!strcmp(field, "env.trunc") || !strcmp(field, "env.term")) {      /* tests: a done flag the caller set, [B][S] (the agent then has no context, env.py:338-348) */
    const int tr = !strcmp(field, "env.trunc");
    for (int b = 0; b < E->B; ++b) for (int s2 = 0; s2 < E->S; ++s2) (tr ? E->env[b].trunc : E->env[b].term)[s2] = (uint8_t)(in[(size_t)b * E->S + s2] != 0);
    return (int64_t)E->B * E->S;
  }
  const ofield* f = find_field(field);
  if (!f || f->is_f) return -1;
  int n = E->kind_count[f->kind];
  for (int b = 0; b < E->B; ++b)
    for (int a = 0; a < E->A; ++a)
      if (E->s.kind[a] == f->kind) E->env[b].ag[a].i[f->slot] = in[(size_t)b * n + E->kind_rank[a]];
  return (int64_t)E->B * n;
}
/* tests: posted prices written by hand, [B][n_sellers] in seller order.  The oracle keeps what the reference keeps -- every buyer's price slots
   (BuyerAgent.seller_prices, stackelberg.py) -- so a seller's posted price is written into the slot of every buyer it is a neighbour of. */
int64_t phxo_set_f64(phxo_env* E, const char* field, const double* in) {
  if (strcmp(field, "seller.posted")) return -1;
  const int nS = E->kind_count[PHX_KIND_SELLER];
  for (int b = 0; b < E->B; ++b)
    for (int a = 0; a < E->A; ++a)
      if (E->s.kind[a] == PHX_KIND_BUYER)
        for (int k = 0; k < E->s.row_ptr[a + 1] - E->s.row_ptr[a]; ++k) {
          const int nb = E->s.col[E->s.row_ptr[a] + k];
          if (E->s.kind[nb] == PHX_KIND_SELLER) E->env[b].ag[a].vec[k] = in[(size_t)b * nS + E->kind_rank[nb]];
        }
  return (int64_t)E->B * nS;
}
int64_t phxo_get_f64(const phxo_env* E, const char* field, double* out) {
  if (!strcmp(field, "buyer.prices")) {
    /* [B][nnz-of-buyers] flattened in agent order */
    size_t w = 0;
    for (int b = 0; b < E->B; ++b)
      for (int a = 0; a < E->A; ++a)
        if (E->s.kind[a] == PHX_KIND_BUYER)
          for (int k = 0; k < E->s.row_ptr[a + 1] - E->s.row_ptr[a]; ++k) out[w++] = E->env[b].ag[a].vec[k];
    return (int64_t)w;
  }
  if (!strcmp(field, "env.sampler")) {
    for (int b = 0; b < E->B; ++b)
      for (int j = 0; j < E->s.n_samplers; ++j) out[(size_t)b * E->s.n_samplers + j] = E->env[b].sampler[j];
    return (int64_t)E->B * E->s.n_samplers;
  }
  const ofield* f = find_field(field);
  if (!f || !f->is_f) return -1;
  int n = E->kind_count[f->kind];
  for (int b = 0; b < E->B; ++b)
    for (int a = 0; a < E->A; ++a)
      if (E->s.kind[a] == f->kind) out[(size_t)b * n + E->kind_rank[a]] = E->env[b].ag[a].f[f->slot];
  return (int64_t)E->B * n;
}

/*
 * phx_oracle.h -- CPU ORACLE for the PhantomEnv.step() hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C, one-env-at-a-time, strictly sequential restatement of the reference's
 * Python algorithm (jpmorganchase/Phantom v2.2.0); every function in phx_oracle.c cites the
 * reference file:line it follows.  It is pinned against golden vectors produced by running
 * the real reference in the build container (tests/golden/gen_goldens.py) and against the
 * reference's own known-answer tests (tests/test_oracle_reference_kats.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library (and libphantom_cpu.so, the same restatement behind the product's phx_* symbols: phx_cpu_abi.c)
 * -- as the checker / the timed CPU baseline, never as the product path.  The
 * product (phantom_amd/csrc, libphantom_amd.so) shares NO algorithm code with this file;
 * the two only share the type definitions in include/phantom_amd.h.
 */
#ifndef PHX_ORACLE_H
#define PHX_ORACLE_H

#include "../include/phantom_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct phxo_env phxo_env;

phxo_env* phxo_create(const phx_spec* spec);          /* NULL on malformed spec */
void      phxo_destroy(phxo_env* e);
const char* phxo_last_error(void);
void      phxo_set_threads(int n);                     /* OpenMP threads over the env batch */
int       phxo_max_threads(void);

int  phxo_obs_dim(const phxo_env* e);
int  phxo_n_strategic(const phxo_env* e);
int  phxo_n_exo(const phxo_env* e);

/* all pointers are HOST pointers; layouts identical to the device ABI */
void phxo_reset(phxo_env* e, const uint8_t* reset_mask, const double* sampler_values,
                const uint8_t* conn_on, float* obs, uint8_t* obs_valid);
void phxo_step(phxo_env* e, const phx_step_io* io);
void phxo_step_begin(phxo_env* e, const phx_step_io* io);   /* acting phase + resolve_network (fsm.py:275-280) */
void phxo_step_end(phxo_env* e, const phx_step_io* io);     /* io->next_stage -> transition, observations, rewards, done flags (fsm.py:304-380) */
void phxo_inject(phxo_env* e, const phx_msg_rec* msgs, int n);
void phxo_resolve(phxo_env* e, int32_t* err, phx_msg_rec* msg_log, int32_t* msg_count);
void phxo_rollout(phxo_env* e, const phx_rollout_io* io);

/* state read-back by field name, shape [B][count-of-that-kind] (or [B] for env fields);
 * returns number of elements written, -1 for an unknown field                           */
int64_t phxo_get_i32(const phxo_env* e, const char* field, int32_t* out);
int64_t phxo_get_f64(const phxo_env* e, const char* field, double* out);
int64_t phxo_set_i32(phxo_env* e, const char* field, const int32_t* in);

/* device-RNG definition (Philox4x32-10), exposed so tests can pin it to known answers */
void phxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* draws the K customer orders of shop-rank `shop` for (seed, global env, tick) */
void phxo_rng_orders(uint64_t seed, int64_t genv, uint32_t tick, int shop, int K, uint8_t* out);
float phxo_rng_action(uint64_t seed, int64_t genv, uint32_t tick, int strat_rank);
uint32_t phxo_rng_rank(uint64_t seed, int64_t genv, uint32_t tick, int agent);
int phxo_rng_publisher(uint64_t seed, int64_t genv, uint32_t tick, int agent, int k, double p);
double phxo_rng_uniform(uint64_t seed, int64_t genv, uint32_t episode, int column, const double prm[4]);
int phxo_rng_connection(uint64_t seed, int64_t genv, uint32_t episode, int conn, double rate);
int64_t phxo_get_u8(const phxo_env* e, const char* field, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif

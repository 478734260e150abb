// phx_sc_fast.h -- host interface of the round-2 supply-chain rollout kernel (phx_sc_rollout.hip)
#pragma once
#include <vector>

#include "phx_dev.h"

// decides whether an env shape takes the fast rollout kernel and with which block shape
// `block`: phx_spec.variant_block (0 auto, PHX_VB_WHOLE_ENVS, or pairs per workgroup); `aligned`: the auto choice prefers
// workgroups of G consecutive (env, shop) pairs whose trajectory row segments are whole 64-byte pieces (G % 16 == 0) and a
// grid that is a multiple of the 256 CUs, over whole envs per workgroup
bool phx_sc_fast_plan(int B, int S, int K_uniform, bool norm_uniform, int num_steps, int block, bool aligned, ScFastPlan* p);
hipError_t phx_launch_sc_rollout_fast(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st);
// round-4 store-wave rollout kernel (phx_sc_rollout_sw.hip): workgroups of G % 16 == 0 consecutive pairs, dense flag planes
bool phx_sc_sw_plan(int B, int S, int K_uniform, bool norm_uniform, int num_steps, int block, ScSwPlan* p, int fsm_ns = 0);
// guard_gen: the call's number for DevSpec::sc_sw_guard (replayed actions: a pre-scan sends calls with an action that rounds below zero to round 1's kernel)
hipError_t phx_launch_sc_rollout_sw(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st, int32_t guard_gen = 0);
void phx_sc_sw_tables(int K, int norm, std::vector<uint8_t>* out);      // the kernel's table image (uploaded once per env)
// time-parallel FSM rollout (phx_sc_rollout_fsm.hip): issues the launch and returns true when the plan applies; the caller then
// issues the lane-per-pair loop guarded by DevSpec::fsm_irregular == *gen (it runs only if some env is off the tabulated stage chain)
bool phx_launch_sc_rollout_fsmfast(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st, hipError_t* err, int32_t* gen);
int32_t phx_fsm_next_gen(const DevSpec& sp);                            // the env's next launch generation for DevSpec::fsm_irregular (never 0)
// the store-wave kernel's FSM instantiation (phx_sc_rollout_sw.hip, MODE 2): does it serve this launch?  (phx_sc_fused.hip)
bool phx_fsm_sw_serves(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st);

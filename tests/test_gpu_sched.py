"""Round 6 on the GPU, part 2: the message-passing engine's compiled-schedule kernel (phx_generic_sched.hip: several env
instances per wave) against the oracle AND against the dynamic kernel it replaces (variants={"step": "generic_dynamic"}) -- per-step
outputs, state, ordered message logs, FSM stages and caches, rule-form handlers, envs of one wave in different stages, envs the
schedule's premise excludes (served by the dynamic kernel behind the same call), the T-step rollout loop."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import phantom_amd as ph
from device_runner import DeviceRunner
from helpers import f32_bits, f64_bits, supply_chain_env
from oracle import OracleEnv

pytestmark = pytest.mark.gpu
NCPU = min(os.cpu_count() or 1, 128)
SCHED, DYN = "phx_sched_step_kernel", "phx_generic_step_kernel"
STATE = ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick")
OUT = ("obs_valid", "reward_valid", "terminated", "truncated", "done_valid", "all_terminated", "all_truncated")


def _cmp_step(d, o, what, fsm=False):
    np.testing.assert_array_equal(d.obs_valid, o.obs_valid, err_msg=f"{what}: obs_valid")
    m = o.obs_valid.astype(bool)
    np.testing.assert_array_equal(f32_bits(d.obs[m]), f32_bits(o.obs[m]), err_msg=f"{what}: obs")
    np.testing.assert_array_equal(d.reward_valid, o.reward_valid, err_msg=f"{what}: reward_valid")
    m = o.reward_valid == 1
    np.testing.assert_array_equal(f64_bits(d.reward[m]), f64_bits(o.reward[m]), err_msg=f"{what}: reward")
    for k in OUT:
        np.testing.assert_array_equal(getattr(d, k), getattr(o, k), err_msg=f"{what}: {k}")
    for f in STATE + (("env.stage",) if fsm else ()):
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{what}: {f}")


@pytest.mark.parametrize("S,ks,B,fsm", [(9, [6] * 9, 70, False), (3, [2, 3, 1], 33, False), (51, [4] * 51, 19, True), (9, [6] * 9, 130, True),
                                        (1, [5], 257, False), (17, [7] * 17, 40, True), (33, [1] * 33, 21, False), (5, [13] * 5, 64, False)])
def test_compiled_schedule_steps_match_the_oracle_and_the_dynamic_kernel(S, ks, B, fsm):
    """phx_step on the compiled schedule (L = 8 / 16 / 32 / 64 lanes per env instance; batches that are not multiples of the envs per
    workgroup; customer counts above one Philox group) == the oracle == the dynamic kernel, over two episodes with device-drawn and
    replayed order sizes."""
    ns = 7
    env = supply_chain_env(S, ks, ns, B, fsm=fsm, force_generic=True, seed=3 + S, env_offset=11)
    envd = supply_chain_env(S, ks, ns, B, fsm=fsm, seed=3 + S, env_offset=11, variants={"step": "generic_dynamic"})
    o, d, dd = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec), DeviceRunner(envd.spec)
    o.reset(); d.reset(); dd.reset()
    rng = np.random.default_rng(S * 7 + B)
    for t in range(2 * ns + 3):
        a = rng.uniform(0, 120, (B, S)).astype(np.float32)
        a[rng.random((B, S)) < 0.1] = 2.5
        x = rng.integers(0, 5, (B, d.n_exo)).astype(np.uint8) if t % 3 == 1 else None
        o.step(a, None, x)
        d.step(a, None, x); assert d.dev.last_kernel().startswith(SCHED), d.dev.last_kernel()       # (the calling thread's LAST call)
        dd.step(a, None, x); assert dd.dev.last_kernel() == DYN, dd.dev.last_kernel()
        _cmp_step(d, o, f"compiled t={t}", fsm); _cmp_step(dd, o, f"dynamic t={t}", fsm)
        done = (o.all_truncated | o.all_terminated).astype(np.uint8)
        if done.any():
            o.reset(done); d.reset(done); dd.reset(done)
    assert (d.err == 0).all() and (dd.err == 0).all()


@pytest.mark.parametrize("S,ks,B,fsm", [(9, [6] * 9, 37, False), (4, [3, 1, 2, 5], 20, True), (51, [4] * 51, 6, False)])
def test_compiled_schedule_writes_the_reference_order_message_log(S, ks, B, fsm):
    """Resolver.push tracking (resolvers.py:41-42) from the compiled schedule: every record (sender, receiver, type, round, payload) of every
    step in the oracle's order, the count, and the same through the T-step loop (rollout.py:369-373)."""
    ns = 6
    env = supply_chain_env(S, ks, ns, B, fsm=fsm, force_generic=True, tracking=True, seed=5)
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(B)
    for t in range(ns + 2):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        o.step(a, None, None); d.step(a, None, None)
        assert d.dev.last_kernel().startswith(SCHED)
        np.testing.assert_array_equal(d.msg_count, o.msg_count)
        for b in (0, B // 2, B - 1):
            lo, ld = o.log(b), d.log(b)
            assert len(lo) == len(ld) and (lo == ld).all(), f"message log of env {b} at step {t}"
        if o.all_truncated.any():
            m = o.all_truncated.astype(np.uint8); o.reset(m); d.reset(m)


@pytest.mark.parametrize("S,ks,B", [(9, [6] * 9, 48), (51, [4] * 51, 10), (3, [2, 2, 2], 100)])
def test_envs_of_one_wave_in_different_stages(S, ks, B):
    """An FSM batch whose env instances are out of step (masked resets): the envs that share a wave stand in different stages, the wave
    runs each stage's schedule over its own lanes -- outputs, caches and stages against the oracle for every env at every step."""
    ns = 9
    env = supply_chain_env(S, ks, ns, B, fsm=True, force_generic=True, seed=21)
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(4)
    mixed = 0
    for t in range(3 * ns):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        mixed += len(np.unique(o.get_i32("env.stage"))) > 1
        o.step(a, None, None); d.step(a, None, None)
        assert d.dev.last_kernel().startswith(SCHED)
        _cmp_step(d, o, f"t={t}", True)
        m = (o.all_truncated | o.all_terminated).astype(np.uint8)
        if t % 2 == 0:
            m |= (rng.random(B) < 0.3).astype(np.uint8)          # resets in mid-episode: those envs return to the initial stage
        if m.any():
            o.reset(m); d.reset(m)
    assert mixed >= ns                                            # (most steps ran with envs of the batch in different stages)
    o.reset((np.arange(B) % 3 == 0).astype(np.uint8)); d.reset((np.arange(B) % 3 == 0).astype(np.uint8))     # out of step again for the rollouts
    for T in (5, 2 * ns + 1):                                     # and the T-step loop from that state
        ro, rd = o.rollout(T), d.rollout(T)
        assert d.dev.last_kernel().startswith(SCHED + "[T-step loop]"), d.dev.last_kernel()
        np.testing.assert_array_equal(rd["obs_valid"], ro["obs_valid"]); np.testing.assert_array_equal(rd["reward_valid"], ro["reward_valid"])
        mo = ro["obs_valid"].astype(bool)
        np.testing.assert_array_equal(f32_bits(rd["obs"][mo]), f32_bits(ro["obs"][mo]))
        mr = ro["reward_valid"] == 1
        np.testing.assert_array_equal(f32_bits(rd["rewards"][mr]), f32_bits(ro["rewards"][mr]))
        np.testing.assert_array_equal(f32_bits(rd["actions"]), f32_bits(ro["actions"]))
        np.testing.assert_array_equal(rd["truncated"], ro["truncated"])
        np.testing.assert_array_equal(f32_bits(rd["last_obs"]), f32_bits(ro["last_obs"]))
        for f in STATE + ("env.stage",):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)


@pytest.mark.parametrize("fsm", [False, True])
def test_envs_outside_the_schedules_premise_are_served_by_the_dynamic_kernel_in_the_same_call(fsm):
    """An acting shop without an action (action_valid) and a strategic agent whose done flag the caller set: the compiled-schedule kernel
    flags exactly those env instances, the dynamic kernel behind it steps them (message logs in the shifted order), every other env
    is stepped by the schedule -- all against the oracle, step after step, also through phx_rollout."""
    S, ks, B, ns = 9, [6] * 9, 150, 8
    env = supply_chain_env(S, ks, ns, B, fsm=fsm, force_generic=True, tracking=True, seed=2)
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(9)
    for t in range(2 * ns):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        valid = np.ones((B, S), np.uint8)
        rows = rng.choice(B, 11, replace=False)
        valid[rows, rng.integers(0, S, 11)] = 0
        if t % 4 == 3:
            valid[rng.choice(B, 3, replace=False)] = 0           # envs in which nobody has an action
        o.step(a, valid, None); d.step(a, valid, None)
        assert d.dev.last_kernel() == SCHED, d.dev.last_kernel()
        _cmp_step(d, o, f"t={t}", fsm)
        np.testing.assert_array_equal(d.msg_count, o.msg_count)
        for b in list(rows[:3]) + [0, B - 1]:
            lo, ld = o.log(b), d.log(b)
            assert len(lo) == len(ld) and (lo == ld).all(), f"message log of env {b} at step {t}"
        if o.all_truncated.any():
            m = o.all_truncated.astype(np.uint8); o.reset(m); d.reset(m)
    # a done flag set by the caller (the agent has no context: env.py:338-348), then a rollout from that state
    import torch
    fl = np.zeros((B, S), np.uint8); fl[5, 2] = 1; fl[77, 0] = 1; fl[B - 1, S - 1] = 1
    o.set_i32("env.trunc", fl.astype(np.int32))
    d.dev.field("env.trunc").copy_(torch.from_numpy(fl).reshape(d.dev.field("env.trunc").shape))
    a = rng.uniform(0, 100, (B, S)).astype(np.float32)
    o.step(a, None, None); d.step(a, None, None)
    assert d.dev.last_kernel() == SCHED
    _cmp_step(d, o, "poked flag", fsm)
    ro, rd = o.rollout(2 * ns + 1), d.rollout(2 * ns + 1)       # the flagged envs run the whole fragment on the dynamic kernel's T-step loop
    if fsm:
        np.testing.assert_array_equal(rd["obs_valid"], ro["obs_valid"]); np.testing.assert_array_equal(rd["reward_valid"], ro["reward_valid"])
    mo = ro["obs_valid"].astype(bool) if fsm else np.ones(ro["truncated"].shape, bool)
    np.testing.assert_array_equal(f32_bits(rd["obs"][mo]), f32_bits(ro["obs"][mo]))
    mr = ro["reward_valid"] == 1 if fsm else mo
    np.testing.assert_array_equal(f32_bits(rd["rewards"][mr]), f32_bits(ro["rewards"][mr]))
    np.testing.assert_array_equal(rd["truncated"], ro["truncated"]); np.testing.assert_array_equal(rd["terminated"], ro["terminated"])
    for f in STATE:
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)
    assert (d.err == 0).all()


def test_rollout_loop_on_the_compiled_schedule_full_size_sample():
    """The generic engine's phx_rollout (T-step loop in the kernel) at the bench's generic shape SC64, B = 4096, T = 50: every row of
    every plane and the state against the oracle; replayed actions + order sizes likewise."""
    S, K, B, T = 9, 6, 4096, 50
    env = supply_chain_env(S, [K] * S, 100, B, force_generic=True, seed=1)
    o, d = OracleEnv(env.spec, threads=NCPU), DeviceRunner(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(1)
    for rep in range(3):
        acts = exo = None
        if rep == 1:
            acts = rng.uniform(0, 110, (T, B, S)).astype(np.float32)
            exo = rng.integers(0, 5, (T, B, d.n_exo)).astype(np.uint8)
        ro, rd = o.rollout(T, acts, exo), d.rollout(T, acts, exo)
        assert d.dev.last_kernel().startswith(SCHED + "[T-step loop]"), d.dev.last_kernel()
        for k in ("obs", "actions", "rewards"):
            np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=f"{k} rep {rep}")
        np.testing.assert_array_equal(rd["truncated"], ro["truncated"]); np.testing.assert_array_equal(rd["terminated"], ro["terminated"])
        np.testing.assert_array_equal(f32_bits(rd["last_obs"]), f32_bits(ro["last_obs"]))
        for f in STATE:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} rep {rep}")
    assert (d.err == 0).all()


def test_config3_shape_on_the_compiled_schedule_sample_against_the_oracle():
    """SC256-FSM, B = 8192 (BASELINE config 3's env on the message-passing engine): ten steps, the first 256 envs against the oracle."""
    S, K, B, Bo = 51, 4, 8192, 256
    env = supply_chain_env(S, [K] * S, 100, B, fsm=True, force_generic=True, seed=42)
    envo = supply_chain_env(S, [K] * S, 100, Bo, fsm=True, force_generic=True, seed=42)
    d, o = DeviceRunner(env.spec), OracleEnv(envo.spec, threads=NCPU)
    d.reset(); o.reset()
    rng = np.random.default_rng(3)
    for t in range(10):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        d.step(a, None, None); o.step(a[:Bo], None, None)
        assert d.dev.last_kernel().startswith(SCHED)
        np.testing.assert_array_equal(d.obs_valid[:Bo], o.obs_valid); np.testing.assert_array_equal(d.reward_valid[:Bo], o.reward_valid)
        m = o.obs_valid.astype(bool)
        np.testing.assert_array_equal(f32_bits(d.obs[:Bo][m]), f32_bits(o.obs[m]))
        m = o.reward_valid == 1
        np.testing.assert_array_equal(f64_bits(d.reward[:Bo][m]), f64_bits(o.reward[m]))
        np.testing.assert_array_equal(d.get_i32("shop.stock")[:Bo], o.get_i32("shop.stock"))
        np.testing.assert_array_equal(d.get_i32("env.stage")[:Bo], o.get_i32("env.stage"))
    assert (d.err == 0).all()

"""Round 4 on the GPU: the two halves of a step around a host-side stage handler (phx_step_begin / phx_step_end, ABI 8)."""
import os
import sys
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import phantom_amd as ph
from device_runner import DeviceRunner
from helpers import env_from_golden, f32_bits, f64_bits, golden, golden_stock_handler, market_env, supply_chain_env
from oracle import OracleEnv
from test_oracle_vs_goldens import replay_supply_chain, stock_handler_stage

pytestmark = pytest.mark.gpu


def test_state_dependent_stage_handler_between_begin_and_end_reproduces_the_reference():
    """golden `sc_fsm_state_handler` (the REFERENCE: a RESTOCK handler that resolves the network, then branches on
    ShopAgent.stock) through the C ABI: phx_step_begin, the handler on the RESOLVED device state, phx_step_end -- the
    handler's decisions, stage sequence, stocks, observation / reward bits and the message log equal the reference's."""
    g = golden("sc_fsm_state_handler")
    replay_supply_chain(g, lambda spec: DeviceRunner(spec), state_handler=stock_handler_stage)


def test_state_dependent_handler_through_the_python_surface_without_a_warning():
    """FiniteStateMachineEnv with a handler that is NOT declared state-independent: step_tensors splits the device step around
    the handler call (fsm.py:275-307 order); no "runs before the step" warning any more."""
    import torch
    g = golden("sc_fsm_state_handler")
    T, B = int(g["T"]), len(g["seeds"])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        env = env_from_golden({k: g[k] for k in g.files if k != "next_stage"}, exogenous="device", restock_handler=golden_stock_handler)
    assert env._has_handlers and env.spec.stage_tab is None
    dev = env._device()
    calls = []
    orig = env._stage_list[0].handler
    env._stage_list[0].handler = lambda e: (calls.append(np.asarray(e.agents["SHOP0"].stock).copy()), orig(e))[1]
    for t in range(T):
        if g["reset_before"][t].any():
            env.reset()
        a = torch.from_numpy(g["actions"][t]).to(dev.device)
        x = torch.from_numpy(g["exo"][t]).to(dev.device)
        env.step_tensors(a, None, x, check_errors=True)
        assert dev.last_kernel().count("phx_generic_step_kernel") >= 1
        np.testing.assert_array_equal(np.atleast_1d(env._h_stage), g["next_stage"][t], err_msg=f"host stage after t={t}")
        np.testing.assert_array_equal(dev.field("env.stage")[:, 0].cpu().numpy(), g["next_stage"][t])
        np.testing.assert_array_equal(dev.obs_valid.cpu().numpy(), g["obs_valid"][t])
        ov = g["obs_valid"][t].astype(bool)
        np.testing.assert_array_equal(f32_bits(dev.obs.cpu().numpy()[ov]), f32_bits(g["obs"][t][ov]))
        rv = g["reward_valid"][t] == 1
        np.testing.assert_array_equal(f64_bits(dev.reward.cpu().numpy()[rv]), f64_bits(g["reward"][t][rv]))
        np.testing.assert_array_equal(dev.field("shop.stock").cpu().numpy(), g["stock"][t])
    # the handler saw the stock AFTER this step's delivery: its first call (t = 0, RESTOCK) sees SHOP0's stock of row 0
    np.testing.assert_array_equal(calls[0], g["stock"][0][:, 0])
    with pytest.raises(NotImplementedError):
        env.rollout(5)


@pytest.mark.parametrize("kind", ["sc_plain", "sc_fsm", "market"])
def test_begin_plus_end_equals_one_step(kind):
    """phx_step_begin followed by phx_step_end == phx_step, for plain, FSM and Stackelberg envs, device against oracle and
    against the device's own single-call step (state, outputs, message counts)."""
    rng = np.random.default_rng(5)
    if kind == "market":
        env = market_env(8, 24, 4, 10, 6, tracking=True)
    else:
        env = supply_chain_env(5, [3, 2, 4, 1, 3], 9, 7, fsm=kind == "sc_fsm", tracking=True, force_generic=True)
    S = env.spec.n_strategic
    one, two, orc = DeviceRunner(env.spec), DeviceRunner(env.spec), OracleEnv(env.spec)
    for r in (one, two, orc):
        r.reset()
    for t in range(14):
        a = rng.uniform(0, 100 if kind != "market" else 1, (env.spec.batch, S)).astype(np.float32)
        av = (rng.random((env.spec.batch, S)) < 0.85).astype(np.uint8)
        x = rng.integers(0, 5, (env.spec.batch, max(one.n_exo, 1))).astype(np.uint8) if one.n_exo else None
        one.step(a, av, x)
        two.step_begin(a, av, x); two.step_end(None)
        orc.step_begin(a, av, x); orc.step_end(None)
        for name in ("obs_valid", "reward_valid", "done_valid", "terminated", "truncated", "all_terminated", "all_truncated", "msg_count"):
            np.testing.assert_array_equal(getattr(one, name), getattr(two, name), err_msg=f"{name} t={t}")
            np.testing.assert_array_equal(getattr(orc, name), getattr(two, name), err_msg=f"oracle {name} t={t}")
        ov = one.obs_valid.astype(bool)
        np.testing.assert_array_equal(f32_bits(one.obs[ov]), f32_bits(two.obs[ov]))
        np.testing.assert_array_equal(f32_bits(orc.obs[ov]), f32_bits(two.obs[ov]))
        rv = one.reward_valid == 1
        np.testing.assert_array_equal(f64_bits(one.reward[rv]), f64_bits(two.reward[rv]))
        np.testing.assert_array_equal(f64_bits(orc.reward[rv]), f64_bits(two.reward[rv]))
        for f in ("env.step", "env.tick") + (("shop.stock", "shop.sales") if kind != "market" else ("seller.tx",)):
            np.testing.assert_array_equal(one.get_i32(f), two.get_i32(f), err_msg=f)
            np.testing.assert_array_equal(orc.get_i32(f), two.get_i32(f), err_msg=f"oracle {f}")
        if (one.all_truncated | one.all_terminated).any():
            for r in (one, two, orc):
                r.reset()


@pytest.mark.gpu
@pytest.mark.parametrize("S,K,B,num_steps", [(9, 6, 64, 7), (3, 2, 48, 5), (51, 4, 16, 9), (4, 4, 64, 3), (1, 3, 16, 4), (12, 1, 8, 6), (7, 5, 4, 5)])
def test_wide_step_kernel_matches_oracle_and_the_lane_per_pair_kernel(S, K, B, num_steps):
    """variants={"step": "wide"}: phx_sc_step_wide_kernel -- four (env, shop) pairs per thread, 16-byte accesses, what PHX_VS_AUTO takes from
    2^19 pairs per launch up -- against the oracle AND against phx_sc_step_kernel on the same inputs: full and partial action dicts
    (env.py:330), episode ends and the steps after them, the per-env words, the state a following rollout starts from."""
    env_w = supply_chain_env(S, [K] * S, num_steps, B, seed=3 + S, env_offset=100, variants={"step": "wide"})
    env_f = supply_chain_env(S, [K] * S, num_steps, B, seed=3 + S, env_offset=100, variants={"step": "fused"})
    o, w, f = OracleEnv(env_w.spec, threads=4), DeviceRunner(env_w.spec), DeviceRunner(env_f.spec)
    o.reset(); w.reset(); f.reset()
    rng = np.random.default_rng(S * 7 + K)
    fields = ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick")
    for t in range(2 * num_steps + 3):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        valid = None if t % 3 else (rng.random((B, S)) < 0.7).astype(np.uint8)
        o.step(a, valid, None)
        w.step(a, valid, None); assert w.dev.last_kernel() == "phx_sc_step_wide_kernel"      # (the calling thread's LAST call)
        f.step(a, valid, None); assert f.dev.last_kernel() == "phx_sc_step_kernel"
        for d in (w, f):
            np.testing.assert_array_equal(f32_bits(d.obs), f32_bits(o.obs), err_msg=f"obs, step {t}")
            np.testing.assert_array_equal(f64_bits(d.reward), f64_bits(o.reward), err_msg=f"reward, step {t}")
            for k in ("obs_valid", "reward_valid", "terminated", "truncated", "done_valid", "all_terminated", "all_truncated"):
                np.testing.assert_array_equal(getattr(d, k), getattr(o, k), err_msg=f"{k}, step {t}")
            for fl in fields:
                np.testing.assert_array_equal(d.get_i32(fl), o.get_i32(fl), err_msg=f"{fl}, step {t}")
        if o.all_truncated.any():                                # the caller's reset of the envs that ended (env.py:195-237)
            m = o.all_truncated.astype(np.uint8)
            o.reset(m); w.reset(m); f.reset(m)
    ro, rw = o.rollout(11), w.rollout(11)
    np.testing.assert_array_equal(f32_bits(rw["obs"]), f32_bits(ro["obs"]))
    assert (w.err == 0).all()

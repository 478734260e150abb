"""GPU (-m gpu): the HIP path, called through the C ABI, against
  (1) the golden vectors produced by the real reference,
  (2) the reference's own known-answer tests (kats.py),
  (3) the CPU oracle on seeded random inputs at moderate sizes,
  (4) size-independent properties at BASELINE.json's full sizes.
Integers, routing, stage ids, flags: bit-exact.  Float obs (f32) and rewards (f64): compared by
bit pattern, i.e. stricter than the 1e-6 relative tolerance north_star allows; the rollout's
f32-rounded reward is compared at rtol 1e-6."""
import numpy as np
import pytest

import phantom_amd as ph

from helpers import (env_from_golden, f32_bits, f64_bits, golden, market_env, market_topology,
                     supply_chain_env)
from kats import ALL_KATS
from oracle import OracleEnv
from test_oracle_vs_goldens import (ADS_CASES, HANDLER_CASES, MARKET_CASES, SC_CASES, SHUFFLE_CASES, replay_ads, replay_market,
                                    replay_supply_chain)

pytestmark = pytest.mark.gpu


def _dev(spec):
    from device_runner import DeviceRunner
    return DeviceRunner(spec)


def test_extension_is_loaded_and_no_fallback():
    import ctypes
    from phantom_amd import _abi
    lib = _abi.load_library()
    assert isinstance(lib, ctypes.CDLL) and lib.phx_abi_version() == _abi.ABI_VERSION


@pytest.mark.parametrize("kat", ALL_KATS, ids=lambda f: f.__name__)
def test_device_kat(kat):
    kat(_dev)


@pytest.mark.parametrize("name", SC_CASES + SHUFFLE_CASES + HANDLER_CASES)
def test_generic_engine_supply_chain_matches_reference(name):
    # tracking on -> the generic engine (with message log) runs
    replay_supply_chain(golden(name), _dev)


@pytest.mark.parametrize("name", SC_CASES)
def test_fused_kernel_supply_chain_matches_reference(name):
    g = dict(golden(name).items())
    g["n_logs"] = np.asarray(0)          # no tracking -> fused static-schedule kernel

    def make(spec):
        r = _dev(spec)
        assert r.dev.uses_fused
        return r
    replay_supply_chain(g, make)


@pytest.mark.parametrize("name", MARKET_CASES)
def test_generic_engine_market_matches_reference(name):
    replay_market(golden(name), _dev)


@pytest.mark.parametrize("name", ADS_CASES)
def test_generic_engine_ads_market_matches_reference(name):
    """digital_ads_market.py on the device: the exchange's handle_batch auction as an inbox reduction,
    the publisher's draws as exogenous inputs, NEP-50 tagged float arithmetic, None observations."""
    replay_ads(golden(name), _dev)


@pytest.mark.parametrize("name", ADS_CASES)
def test_fused_ads_kernel_matches_reference(name):
    """the same goldens through phx_ads_fused.hip (static schedule, no tracking; ads_stochastic: connectivity
    below 1, messages along connections that are off are dropped)."""
    def make(spec):
        r = _dev(spec)
        assert r.dev.uses_fused
        return r
    replay_ads(golden(name), make, tracking=False)


@pytest.mark.parametrize("name", ["stk_small", "stk_full"])
def test_fused_market_kernel_matches_reference(name):
    def make(spec):
        r = _dev(spec)
        assert r.dev.uses_fused
        return r
    replay_market(golden(name), make, tracking=False)


def test_fused_market_random_differential_vs_oracle():
    """Stackelberg market 16 sellers x 96 buyers, B=48, 45 steps over 20-step episodes:
    fused kernel and generic engine against the oracle, ties between sellers included."""
    rng = np.random.RandomState(21)
    L, Fw, d, B, T = 16, 96, 4, 48, 45
    S = L + Fw
    envs = [market_env(L, Fw, d, 20, B, force_generic=fg) for fg in (False, True)]
    o = OracleEnv(envs[0].spec)
    devs = [_dev(e.spec) for e in envs]
    assert devs[0].dev.uses_fused and not devs[1].dev.uses_fused
    o.reset(); [x.reset() for x in devs]
    for t in range(T):
        step = o.get_i32("env.step")[:, 0] + 1
        act = np.zeros((B, S), np.float32); valid = np.zeros((B, S), np.uint8)
        odd = (step % 2 == 1)
        act[:, :L] = rng.randint(1, 9, size=(B, L)) / 8.0
        act[:, L:] = (rng.rand(B, Fw) < 0.7)
        valid[odd, :L] = 1; valid[~odd, L:] = 1
        valid &= (rng.rand(B, S) < 0.95).astype(np.uint8)          # some agents get no action
        o.step(act, valid, None)
        for x in devs:
            x.step(act, valid, None)
            for f in ("obs_valid", "reward_valid", "done_valid", "all_truncated", "all_terminated", "err"):
                np.testing.assert_array_equal(getattr(x, f), getattr(o, f), err_msg=f"{f} t={t}")
            np.testing.assert_array_equal(f32_bits(x.obs), f32_bits(o.obs), err_msg=f"obs t={t}")
            np.testing.assert_array_equal(f64_bits(x.reward), f64_bits(o.reward), err_msg=f"rew t={t}")
            for f in ("seller.tx", "buyer.bought"):
                np.testing.assert_array_equal(x.get_i32(f), o.get_i32(f), err_msg=f)
            for f in ("seller.price", "seller.revenue", "buyer.paid", "buyer.prices"):
                np.testing.assert_array_equal(f64_bits(x.get_f64(f)), f64_bits(o.get_f64(f)), err_msg=f)
        done = o.all_truncated.astype(np.uint8)
        if done.any():
            oo, ov = o.reset(done)
            for x in devs:
                do, dv = x.reset(done)
                np.testing.assert_array_equal(dv[done > 0], ov[done > 0])
                np.testing.assert_array_equal(f32_bits(do[done > 0]), f32_bits(oo[done > 0]))


def test_full_size_market_fused_equals_generic():
    """BASELINE config 5 size (128 leaders / 1024 followers), B=256: fused == generic engine, plus
    conservation: every Order is booked by exactly one seller."""
    rng = np.random.RandomState(4)
    L, Fw, d, B = 128, 1024, 8, 256
    S = L + Fw
    f, g = (_dev(market_env(L, Fw, d, 100, B, force_generic=fg).spec) for fg in (False, True))
    assert f.dev.uses_fused and not g.dev.uses_fused
    f.reset(); g.reset()
    for t in range(6):
        act = np.zeros((B, S), np.float32); valid = np.zeros((B, S), np.uint8)
        if t % 2 == 0:
            act[:, :L] = rng.randint(1, 17, size=(B, L)) / 16.0; valid[:, :L] = 1
        else:
            act[:, L:] = (rng.rand(B, Fw) < 0.6); valid[:, L:] = 1
        f.step(act, valid, None); g.step(act, valid, None)
        np.testing.assert_array_equal(f32_bits(f.obs), f32_bits(g.obs))
        np.testing.assert_array_equal(f64_bits(f.reward), f64_bits(g.reward))
        np.testing.assert_array_equal(f.reward_valid, g.reward_valid)
        np.testing.assert_array_equal(f.get_i32("seller.tx"), g.get_i32("seller.tx"))
        np.testing.assert_array_equal(f64_bits(f.get_f64("seller.revenue")), f64_bits(g.get_f64("seller.revenue")))
        if t % 2 == 1:
            np.testing.assert_array_equal(f.get_i32("seller.tx").sum(1), f.get_i32("buyer.bought").sum(1))
    assert (f.err == 0).all() and (g.err == 0).all()


def _compare_step(o, d, t):
    for f in ("obs_valid", "reward_valid", "done_valid", "terminated", "truncated", "all_terminated",
              "all_truncated", "err"):
        np.testing.assert_array_equal(getattr(d, f), getattr(o, f), err_msg=f"{f} t={t}")
    np.testing.assert_array_equal(f32_bits(d.obs), f32_bits(o.obs), err_msg=f"obs t={t}")
    np.testing.assert_array_equal(f64_bits(d.reward), f64_bits(o.reward), err_msg=f"reward t={t}")
    for f in ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step"):
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} t={t}")


@pytest.mark.parametrize("fsm", [False, True])
@pytest.mark.parametrize("force_generic", [False, True])
@pytest.mark.parametrize("exo_mode", ["replay", "device_rng"])
def test_random_differential_vs_oracle(fsm, force_generic, exo_mode):
    """SC topologies, B=192, 230 steps (two episode ends), partial action masks."""
    rng = np.random.RandomState(5 + fsm)
    B, S, K, T = 192, 9, 6, 230
    env = supply_chain_env(S, [K] * S, 100, B, fsm=fsm, force_generic=force_generic, seed=77,
                           env_offset=1000)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    assert d.dev.uses_fused == (not force_generic)
    o.reset(); d.reset()
    for t in range(T):
        act = rng.uniform(-20, 130, size=(B, S)).astype(np.float32)
        act[rng.rand(B, S) < 0.1] = np.float32(rng.randint(0, 50)) + np.float32(0.5)
        valid = (rng.rand(B, S) < 0.9).astype(np.uint8)
        exo = rng.randint(0, 5, size=(B, S * K)).astype(np.uint8) if exo_mode == "replay" else None
        o.step(act, valid, exo); d.step(act, valid, exo)
        _compare_step(o, d, t)
        done = (o.all_truncated | o.all_terminated).astype(np.uint8)
        if done.any():
            oo, ov = o.reset(done); do, dv = d.reset(done)
            m = done.astype(bool)
            np.testing.assert_array_equal(dv[m], ov[m])
            np.testing.assert_array_equal(f32_bits(do[m]), f32_bits(oo[m]))


def test_ragged_masked_reset_and_large_customer_counts():
    rng = np.random.RandomState(9)
    ks = [1, 17, 3, 64, 2]
    B, T = 50, 40
    env = supply_chain_env(len(ks), ks, 7, B, norm_customers=11)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    for t in range(T):
        act = rng.uniform(0, 100, size=(B, len(ks))).astype(np.float32)
        exo = rng.randint(0, 5, size=(B, sum(ks))).astype(np.uint8) if t % 2 else None
        o.step(act, None, exo); d.step(act, None, exo)
        _compare_step(o, d, t)
        mask = ((o.all_truncated > 0) | (rng.rand(B) < 0.05)).astype(np.uint8)   # also mid-episode resets
        if mask.any():
            oo, ov = o.reset(mask); do, dv = d.reset(mask)
            np.testing.assert_array_equal(f32_bits(do[mask > 0]), f32_bits(oo[mask > 0]))


@pytest.mark.parametrize("mode", ["replay", "device_rng"])
def test_rollout_matches_oracle(mode):
    rng = np.random.RandomState(3)
    B, S, K, T = 96, 9, 6, 250
    env = supply_chain_env(S, [K] * S, 100, B, seed=1234, env_offset=7)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    # a few single steps first so the fragment starts mid-episode
    for t in range(3):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        o.step(a, None, None); d.step(a, None, None)
    if mode == "replay":
        acts = rng.uniform(-5, 120, (T, B, S)).astype(np.float32)
        exo = rng.randint(0, 5, (T, B, S * K)).astype(np.uint8)
    else:
        acts = exo = None
    ro, rd = o.rollout(T, acts, exo), d.rollout(T, acts, exo)
    np.testing.assert_array_equal(f32_bits(rd["obs"]), f32_bits(ro["obs"]))
    np.testing.assert_array_equal(f32_bits(rd["actions"]), f32_bits(ro["actions"]))
    np.testing.assert_allclose(rd["rewards"], ro["rewards"], rtol=1e-6, atol=0)
    np.testing.assert_array_equal(f32_bits(rd["rewards"]), f32_bits(ro["rewards"]))
    np.testing.assert_array_equal(rd["truncated"], ro["truncated"])
    np.testing.assert_array_equal(rd["terminated"], ro["terminated"])
    np.testing.assert_array_equal(f32_bits(rd["last_obs"]), f32_bits(ro["last_obs"]))
    assert ro["truncated"].sum() > 0          # at least one episode boundary inside the fragment
    for f in ("shop.stock", "shop.sales", "shop.missed_sales", "env.step", "env.tick"):
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)
    # stepping continues identically after a rollout
    a = rng.uniform(0, 100, (B, S)).astype(np.float32)
    o.step(a, None, None); d.step(a, None, None)
    _compare_step(o, d, -1)


def test_rng_rejection_branch_on_device():
    """the 3.3e-6 redraw branch of the device RNG: place the batch on a global env index where
    the order word of (tick 0, shop 0) is rejected; fused, generic and rollout vs oracle."""
    from helpers import find_rng_rejection
    genv = find_rng_rejection(seed=1)
    B = 8
    for force_generic in (False, True):
        env = supply_chain_env(2, [6, 6], 20, B, seed=1, env_offset=genv - 3, force_generic=force_generic)
        o, d = OracleEnv(env.spec), _dev(env.spec)
        o.reset(); d.reset()
        a = np.full((B, 2), 50.0, np.float32)
        o.step(a, None, None); d.step(a, None, None)
        _compare_step(o, d, 0)
    env = supply_chain_env(2, [6, 6], 20, B, seed=1, env_offset=genv - 3)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    ro, rd = o.rollout(5), d.rollout(5)
    np.testing.assert_array_equal(f32_bits(rd["obs"]), f32_bits(ro["obs"]))
    np.testing.assert_array_equal(f32_bits(rd["rewards"]), f32_bits(ro["rewards"]))


def test_full_size_rollout_matches_oracle():
    """BASELINE configs[1] at full size: SC64, B=4096, one T=100 fragment, device RNG (3.7 M order
    words, about a dozen of them through the redraw branch) -- bit-equal to the oracle."""
    env = supply_chain_env(9, [6] * 9, 100, 4096, seed=42)
    o, d = OracleEnv(env.spec, threads=8), _dev(env.spec)
    o.reset(); d.reset()
    ro, rd = o.rollout(100), d.rollout(100)
    for k in ("obs", "actions", "rewards"):
        np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=k)
    np.testing.assert_array_equal(rd["truncated"], ro["truncated"])
    np.testing.assert_array_equal(d.get_i32("shop.stock"), o.get_i32("shop.stock"))


def test_rng_fallback_block_is_exercised():
    """more than 6 customers per shop: further order words in blocks 1, 2, ... (ragged last group);
    compare with the oracle."""
    B, ks = 64, [70, 41]
    env = supply_chain_env(2, ks, 20, B, seed=99)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    for t in range(25):
        a = np.full((B, 2), 50.0, np.float32)
        o.step(a, None, None); d.step(a, None, None)
        _compare_step(o, d, t)


# ---- BASELINE.json full sizes: size-independent properties ------------------------------------
@pytest.mark.parametrize("cfg", [dict(S=9, K=6, B=4096, fsm=False), dict(S=51, K=4, B=8192, fsm=True)],
                         ids=["SC64_B4096", "SC256_FSM_B8192"])
def test_full_size_fused_equals_generic_and_invariants(cfg):
    import torch
    S, K, B = cfg["S"], cfg["K"], cfg["B"]
    envf = supply_chain_env(S, [K] * S, 100, B, fsm=cfg["fsm"], seed=5)
    envg = supply_chain_env(S, [K] * S, 100, B, fsm=cfg["fsm"], seed=5, force_generic=True)
    f, g = _dev(envf.spec), _dev(envg.spec)
    assert f.dev.uses_fused and not g.dev.uses_fused
    f.reset(); g.reset()
    gen = torch.Generator(device="cpu").manual_seed(1)
    for t in range(12):
        act = (torch.rand(B, S, generator=gen) * 100).numpy().astype(np.float32)
        exo = torch.randint(0, 5, (B, S * K), generator=gen, dtype=torch.uint8).numpy()
        stock0 = f.get_i32("shop.stock").copy()
        stage = f.get_i32("env.stage")[:, 0].copy()
        f.step(act, None, exo); g.step(act, None, exo)
        _compare_step(g, f, t)
        stock, sales, missed = f.get_i32("shop.stock"), f.get_i32("shop.sales"), f.get_i32("shop.missed_sales")
        D = exo.reshape(B, S, K).sum(-1).astype(np.int64)
        selling = np.ones(B, bool) if not cfg["fsm"] else stage == 1
        restock = np.ones(B, bool) if not cfg["fsm"] else stage == 0
        # every order is either sold or missed; sales never exceed the stock held before delivery
        np.testing.assert_array_equal((sales + missed)[selling], D[selling])
        assert (sales[selling] <= np.maximum(stock0[selling], 0)).all() and (stock <= 100).all()
        req = np.minimum(np.rint(act).astype(np.int64), 100 - stock0)
        exp = np.minimum(stock0 - np.where(selling[:, None], sales, 0) + np.where(restock[:, None], req, 0), 100)
        np.testing.assert_array_equal(stock, exp)
    assert (f.err == 0).all()


def test_python_surface_matches_reference_b1():
    """the drop-in dict API at batch_size=1, exogenous draws from the global numpy stream:
    same np.random.seed -> same trajectory as the reference's SupplyChainEnv (Appendix B)."""
    import phantom_amd as ph
    g = golden("sc7_fixed20")
    env = ph.SupplyChainEnv()
    np.random.seed(0)
    obs, infos = env.reset()
    assert list(obs) == ["SHOP"] and infos == {}
    np.testing.assert_array_equal(obs["SHOP"], np.zeros(3, np.float32))
    for t in range(100):
        step = env.step({"SHOP": np.array([20.0], dtype=np.float32)})
        np.testing.assert_array_equal(f32_bits(step.observations["SHOP"]), f32_bits(g["obs"][t, 0, 0]))
        assert step.rewards["SHOP"] == g["reward"][t, 0, 0]
        assert step.terminations == {"SHOP": False, "__all__": False}
        assert step.truncations == {"SHOP": False, "__all__": t == 99}
        assert step.infos == {"SHOP": {}}
        assert env["SHOP"].stock == g["stock"][t, 0, 0]
        assert env.current_step == t + 1
    obs, _ = env.reset()                   # stale `sales` survives the reset (Appendix B)
    np.testing.assert_array_equal(f32_bits(obs["SHOP"]), f32_bits(g["reset_obs"][100, 0, 0]))


def test_python_surface_tracking_and_errors():
    import phantom_amd as ph
    resolver = ph.BatchResolver(enable_tracking=True)
    env = ph.SupplyChainEnv(resolver=resolver)
    np.random.seed(0)
    env.reset()
    env.step({"SHOP": np.array([20.0], dtype=np.float32)})
    msgs = resolver.tracked_messages
    assert msgs[0] == ph.Message("SHOP", "WAREHOUSE", ph.StockRequest(20))
    assert [m.payload.size for m in msgs[1:6]] == [4, 0, 3, 3, 3]          # Appendix B draws
    assert msgs[6] == ph.Message("WAREHOUSE", "SHOP", ph.StockResponse(20))
    assert all(m.payload == ph.OrderResponse(0) for m in msgs[7:12]) and len(msgs) == 12
    # network-level API parity (tests/network/test_network.py:75-99)
    net = ph.Network([ph.CashboxAgent("mm"), ph.CashboxAgent("inv"), ph.CashboxAgent("inv2")])
    net.add_connection("mm", "inv")
    net.send("mm", "inv", ph.CashMessage(100.0))
    net.resolve()
    assert net.agents["mm"].total_cash == 25.0 and net.agents["inv"].total_cash == 50.0
    with pytest.raises(ph.NetworkError):
        net.send("mm", "inv2", ph.CashMessage(100.0))
    n2 = ph.Network([ph.ReqRespAgent("A"), ph.ReqRespAgent("B")], ph.BatchResolver(round_limit=0))
    n2.add_connection("A", "B")
    n2.send("A", "B", ph.Request(0.0))
    with pytest.raises(RuntimeError):
        n2.resolve()


def test_metrics_and_rllib_adapters():
    """caller-side contract: SimpleAgentMetric reflection on device state,
    RLlibEnvWrapper pass-through, BaseEnv-shaped poll/send_actions with lazy per-env views."""
    import phantom_amd as ph
    from phantom_amd.metrics import SimpleAgentMetric, logging_helper
    from phantom_amd.rllib import BatchedBaseEnv, RLlibEnvWrapper
    g = golden("sc7_fixed20")
    np.random.seed(0)
    wrapped = RLlibEnvWrapper(ph.SupplyChainEnv())
    assert wrapped.get_agent_ids() == {"SHOP"} and wrapped.num_steps == 100      # __getattr__ delegation
    metrics = {"SHOP/stock": SimpleAgentMetric("SHOP", "stock", "mean"),
               "SHOP/sales": SimpleAgentMetric("SHOP", "sales", "sum")}
    values = {}
    np.random.seed(0)
    wrapped.reset()
    for t in range(10):
        step = wrapped.step({"SHOP": np.array([20.0], np.float32)})
        logging_helper(wrapped.env, metrics, values)
        assert step.rewards["SHOP"] == g["reward"][t, 0, 0]
    assert values["SHOP/stock"] == g["stock"][:10, 0, 0].tolist()
    assert metrics["SHOP/stock"].reduce(values["SHOP/stock"], "train") == g["stock"][:10, 0, 0].mean()
    assert metrics["SHOP/sales"].reduce(values["SHOP/sales"], "train") == g["sales"][:10, 0, 0].sum()
    # batched BaseEnv-shaped driver vs the oracle
    B, S, K = 6, 3, 2
    env = supply_chain_env(S, [K] * S, 5, B, seed=3, exogenous="device")
    o = OracleEnv(env.spec)
    base = BatchedBaseEnv(env)
    obs, rew, term, trunc, infos, _ = base.poll()                 # first poll = reset observations
    oo, _ = o.reset()
    assert list(obs[2].keys()) == ["SHOP0", "SHOP1", "SHOP2"]
    np.testing.assert_array_equal(obs[2]["SHOP1"], oo[2, 1])
    rng = np.random.RandomState(0)
    for t in range(5):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        base.send_action_tensor(a)
        obs, rew, term, trunc, infos, _ = base.poll()
        o.step(a, None, None)
        for b in (0, B - 1):
            np.testing.assert_array_equal(obs[b]["SHOP2"], o.obs[b, 2])
            assert rew[b]["SHOP0"] == o.reward[b, 0]
            assert trunc[b]["__all__"] == bool(o.all_truncated[b]) and term[b]["SHOP1"] is False
    assert trunc[0]["__all__"] is True
    sub = base.get_sub_environments()[4]
    assert sub.agents["SHOP1"].stock == int(o.get_i32("shop.stock")[4, 1])
    obs1, _ = base.try_reset(1)
    assert list(obs1) == [1] and env.current_step.tolist() == [5, 0, 5, 5, 5, 5]


@pytest.mark.parametrize("S,K,B,T,num_steps", [
    (51, 4, 18, 130, 100),      # SC256 topology: 204-pair blocks, B not a multiple of the block's env count
    (9, 6, 13, 75, 30),         # tail block + several episode ends per fragment
    (1, 5, 300, 40, 7),         # the shipped SC7 shape, many envs per block, num_steps < chunk length
    (3, 2, 64, 33, 100),        # tiny shops
    (64, 1, 8, 20, 10),         # one block-row wider than a wave
])
def test_rollout_shapes_match_oracle(S, K, B, T, num_steps):
    """every launch geometry of the rollout kernel (wide / narrow copy-out, tail blocks, replay and
    device-RNG variants) against the oracle."""
    rng = np.random.RandomState(S * 1000 + B)
    for mode in ("device_rng", "replay"):
        env = supply_chain_env(S, [K] * S, num_steps, B, seed=5, env_offset=123)
        o, d = OracleEnv(env.spec), _dev(env.spec)
        o.reset(); d.reset()
        acts = exo = None
        if mode == "replay":
            acts = rng.uniform(-10, 130, (T, B, S)).astype(np.float32)
            exo = rng.randint(0, 5, (T, B, S * K)).astype(np.uint8)
        ro, rd = o.rollout(T, acts, exo), d.rollout(T, acts, exo)
        for k in ("obs", "actions", "rewards", "last_obs"):
            np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=f"{k} {mode}")
        np.testing.assert_array_equal(rd["truncated"], ro["truncated"], err_msg=mode)
        np.testing.assert_array_equal(rd["terminated"], ro["terminated"])
        for f in ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick"):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} {mode}")
        ro2, rd2 = o.rollout(5, None, None), d.rollout(5, None, None)       # fragments chain
        np.testing.assert_array_equal(f32_bits(rd2["obs"]), f32_bits(ro2["obs"]))


@pytest.mark.parametrize("S,K,B,T,num_steps", [(2, 3, 5, 40, 6), (9, 6, 40, 130, 100), (51, 4, 12, 60, 25)])
def test_fsm_rollout_matches_oracle(S, K, B, T, num_steps):
    """FiniteStateMachineEnv rollouts (RESTOCK -> SELL -> RESTOCK ...): stage masks, reward cache with
    emit-on-observe, terminal dump of the cached dicts, auto-reset -- fused kernel vs oracle."""
    rng = np.random.RandomState(S + B)
    for mode in ("device_rng", "replay"):
        env = supply_chain_env(S, [K] * S, num_steps, B, fsm=True, seed=8, env_offset=31)
        o, d = OracleEnv(env.spec), _dev(env.spec)
        assert d.dev.uses_fused
        o.reset(); d.reset()
        a0 = rng.uniform(0, 100, (B, S)).astype(np.float32)
        o.step(a0, None, None); d.step(a0, None, None)             # start the fragment in stage SELL
        acts = exo = None
        if mode == "replay":
            acts = rng.uniform(-10, 130, (T, B, S)).astype(np.float32)
            exo = rng.randint(0, 5, (T, B, S * K)).astype(np.uint8)
        ro, rd = o.rollout(T, acts, exo), d.rollout(T, acts, exo)
        for k in ("obs_valid", "reward_valid", "truncated", "terminated"):
            np.testing.assert_array_equal(rd[k], ro[k], err_msg=f"{k} {mode}")
        for k in ("obs", "actions", "rewards", "last_obs"):
            np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=f"{k} {mode}")
        assert (ro["obs_valid"] == 0).any() and (ro["obs_valid"] == 1).any() and ro["truncated"].any()
        for f in ("shop.stock", "shop.sales", "shop.missed_sales", "env.step", "env.tick", "env.stage", "env.prev_stage"):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} {mode}")
        a1 = rng.uniform(0, 100, (B, S)).astype(np.float32)         # per-step launches continue identically
        o.step(a1, None, None); d.step(a1, None, None)
        _compare_step(o, d, -1)


@pytest.mark.parametrize("fsm", [False, True])
def test_pipelined_collector_equals_one_shot_rollout(fsm):
    """distributed.RolloutCollector on one GPU (world 1): a fragment produced in chunks on the
    main stream and collected on the side stream equals the one-shot phx_rollout and the oracle."""
    import torch
    from phantom_amd.distributed import device_env_collector
    S, K, B, T, chunk = 9, 6, 64, 120, 24
    env = supply_chain_env(S, [K] * S, 50, B, fsm=fsm, seed=5, env_offset=7)
    o, d1, d2 = OracleEnv(env.spec), _dev(env.spec), _dev(env.spec)
    o.reset(); d1.reset(); d2.reset()
    ro = o.rollout(T, None, None)
    col = device_env_collector(d2.dev, T, chunk, n_buffers=2)
    out = col.collect()
    torch.cuda.synchronize()
    # gathered payload: obs | actions | rewards | [validity planes] | bit-packed done flags (SURVEY 8e iii)
    names = ["obs", "actions", "rewards"] + (["obs_valid", "reward_valid"] if fsm else [])
    assert out[0].shape == (T // chunk, 1, chunk, B, S, 3) and len(out) == len(names) + 1
    from phantom_amd.distributed import unpack_done_flags
    tr = [unpack_done_flags(d2.dev, out[-1][c, 0], col.flags_per_chunk, col.flag_planes) for c in range(T // chunk)]
    np.testing.assert_array_equal(torch.cat([t.view(chunk, B, S) for t, _ in tr], 0).cpu().numpy(), ro["truncated"])
    np.testing.assert_array_equal(torch.cat([e.view(chunk, B, S) for _, e in tr], 0).cpu().numpy(), ro["terminated"])
    for name, x in zip(names, out):
        got = x[:, 0].reshape((T,) + tuple(x.shape[3:])).cpu().numpy()
        want = ro[name]
        if got.dtype == np.float32:
            got, want = f32_bits(got), f32_bits(want)
        np.testing.assert_array_equal(got, want, err_msg=name)


def test_step_launches_are_graph_capturable():
    """phx_step is stream-ordered with no allocation or sync inside, so a block of steps can be
    captured into a hipGraph on torch's capture stream and replayed; results equal eager launches."""
    import torch
    S, K, B, N = 9, 6, 256, 30
    envs = [supply_chain_env(S, [K] * S, 12, B, seed=3) for _ in range(2)]
    de, dg = _dev(envs[0].spec).dev, _dev(envs[1].spec).dev
    de.reset(); dg.reset()
    acts = torch.rand(N, B, S, device=de.device) * 100.0
    log_e, log_g = (torch.zeros(N, B, S, 3, device=de.device) for _ in range(2))
    rew_e, rew_g = (torch.zeros(N, B, S, dtype=torch.float64, device=de.device) for _ in range(2))
    for i in range(N):
        o = de.step(acts[i]); log_e[i].copy_(o.observations); rew_e[i].copy_(o.rewards)
        if i % 12 == 11:
            de.reset()
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        for i in range(N):
            o = dg.step(acts[i]); log_g[i].copy_(o.observations); rew_g[i].copy_(o.rewards)
            if i % 12 == 11:
                dg.reset()
    log_g.zero_(); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(log_e, log_g) and torch.equal(rew_e, rew_g)
    for f in ("shop.stock", "shop.sales", "env.step", "env.tick"):
        assert torch.equal(de.field(f), dg.field(f)), f


def _typed_env(S, ks, num_steps, B, fsm, force_generic, device_sampling, seed=13, env_offset=40):
    """tutorial-2 shops: shops 0/1 share a sampler, shop 2 has a clipped one, shop 3 a constant,
    the rest stay on the Supertype() default."""
    s0, s1 = ph.UniformFloatSampler(0.0, 0.2), ph.UniformFloatSampler(0.05, 0.15, 0.07, 0.13)
    sup = {"SHOP0": ph.TypedShopAgent.Supertype(s0), "SHOP1": ph.TypedShopAgent.Supertype(s0),
           "SHOP2": ph.TypedShopAgent.Supertype(s1), "SHOP3": ph.TypedShopAgent.Supertype(0.15)}
    return supply_chain_env(S, ks, num_steps, B, fsm=fsm, force_generic=force_generic, typed=True,
                            agent_supertypes=sup, seed=seed, env_offset=env_offset,
                            exogenous="device" if device_sampling else "numpy")


@pytest.mark.parametrize("fsm", [False, True])
@pytest.mark.parametrize("force_generic", [False, True])
@pytest.mark.parametrize("device_sampling", [False, True])
def test_typed_shops_differential_vs_oracle(fsm, force_generic, device_sampling):
    """Supertypes on the device (tutorial 2): per-env sampled excess_stock_weight in the reward and
    as 4th observation; sampler values fed by the host or drawn by the device Philox stream."""
    import phantom_amd as ph_
    rng = np.random.RandomState(17 + fsm)
    S, ks, B, T = 6, [2, 3, 1, 2, 6, 4], 70, 45
    env = _typed_env(S, ks, 10, B, fsm, force_generic, device_sampling)
    spec = env.spec
    assert spec.n_samplers == 2 and spec.obs_dim == 4
    assert (spec.sampler_kind == (ph_._abi.SAMPLER_UNIFORM if device_sampling else ph_._abi.SAMPLER_HOST)).all()
    o, d = OracleEnv(spec), _dev(spec)
    assert d.dev.uses_fused == (not force_generic) and d.D == 4

    def reset(mask=None):
        vals = None if device_sampling else rng.uniform(0.0, 0.2, (B, 2))
        (oo, ov), (do, dv) = o.reset(mask, vals), d.reset(mask, vals)
        m = slice(None) if mask is None else mask.astype(bool)
        np.testing.assert_array_equal(dv[m], ov[m])
        np.testing.assert_array_equal(f32_bits(do[m]), f32_bits(oo[m]))
        np.testing.assert_array_equal(f64_bits(d.get_f64("env.sampler")), f64_bits(o.get_f64("env.sampler")))
        np.testing.assert_array_equal(d.get_i32("env.episode"), o.get_i32("env.episode"))

    np.testing.assert_array_equal(f64_bits(d.get_f64("env.sampler")), f64_bits(o.get_f64("env.sampler")))
    reset()
    for t in range(T):
        act = rng.uniform(-20, 130, size=(B, S)).astype(np.float32)
        exo = rng.randint(0, 5, size=(B, sum(ks))).astype(np.uint8) if t % 3 else None
        o.step(act, None, exo); d.step(act, None, exo)
        _compare_step(o, d, t)
        done = ((o.all_truncated > 0) | (rng.rand(B) < 0.03)).astype(np.uint8)
        if done.any():
            reset(done)
    assert len(np.unique(o.get_f64("env.sampler")[:, 0])) > B // 2     # per-env values differ
    assert (o.obs[:, :, 3][o.obs_valid > 0] > 0).all()
    if device_sampling and not force_generic:
        # fused rollout with auto-reset: the device redraws every sampler at each episode boundary
        for mode in ("device_rng", "replay"):
            acts = exo = None
            if mode == "replay":
                acts = rng.uniform(-10, 130, (33, B, S)).astype(np.float32)
                exo = rng.randint(0, 5, (33, B, sum(ks))).astype(np.uint8)
            ro, rd = o.rollout(33, acts, exo), d.rollout(33, acts, exo)
            for k in ("truncated", "terminated") + (("obs_valid", "reward_valid") if fsm else ()):
                np.testing.assert_array_equal(rd[k], ro[k], err_msg=f"{k} {mode}")
            for k in ("obs", "actions", "rewards", "last_obs"):
                np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=f"{k} {mode}")
            assert ro["truncated"].sum() >= 3 * B * S
            np.testing.assert_array_equal(f64_bits(d.get_f64("env.sampler")), f64_bits(o.get_f64("env.sampler")))
            for f in ("shop.stock", "shop.sales", "env.step", "env.tick", "env.episode"):
                np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} {mode}")
        act = rng.uniform(0, 100, size=(B, S)).astype(np.float32)
        o.step(act, None, None); d.step(act, None, None)
        _compare_step(o, d, -1)
    elif not force_generic:
        with pytest.raises(Exception, match="PHX_SAMPLER_UNIFORM"):
            d.rollout(5)


def test_python_surface_supertypes_match_reference_tests():
    """tests/test_supertypes_env.py:11-44,86-104 through the Python surface (B = 1): samplers are
    drawn once at construction and once per reset; agents sharing nothing see their own values."""
    import phantom_amd as ph_

    class MockSampler(ph_.Sampler):                      # tests/__init__.py:9-15
        def __init__(self, value):
            self._value = value

        def sample(self):
            self._value += 1
            return self._value

    agents = [ph_.MockStrategicAgent("a1"), ph_.MockStrategicAgent("a2")]
    net = ph_.Network(agents)
    s1, s2 = MockSampler(0), MockSampler(10)
    sup = {"a1": ph_.MockStrategicAgent.Supertype(type_value=s1), "a2": {"type_value": s2}}
    env = ph_.PhantomEnv(1, net, agent_supertypes=sup)
    assert env._samplers == [s1, s2]
    assert env.agents["a1"].supertype == sup["a1"]
    assert env.agents["a1"].type == ph_.MockStrategicAgent.Supertype(1)
    assert env.agents["a2"].type == ph_.MockStrategicAgent.Supertype(11)
    env.reset()
    assert env.agents["a1"].type == ph_.MockStrategicAgent.Supertype(2)
    assert env.agents["a2"].type == ph_.MockStrategicAgent.Supertype(12)
    np.testing.assert_array_equal(env._device().field("env.sampler").cpu().numpy(), [[2.0, 12.0]])

    # tutorial 2 through the dict API, numpy stream (B = 1) vs the reference golden
    from helpers import typed_supertypes
    g = golden("sc_typed")
    np.random.seed(int(g["seeds"][0]))
    env = supply_chain_env(int(g["n_shops"]), g["ks"], int(g["num_steps"]), 1, typed=True,
                           norm_customers=int(g["norm_customers"]), agent_supertypes=typed_supertypes(g))
    shops = [f"SHOP{i}" for i in range(int(g["n_shops"]))]
    for t in range(int(g["T"])):
        if g["reset_before"][t, 0]:
            obs, _ = env.reset()
            for i, sid in enumerate(shops):
                np.testing.assert_array_equal(f32_bits(obs[sid]), f32_bits(g["reset_obs"][t, 0, i]))
                assert env[sid].type.excess_stock_weight == g["type_w"][t, 0, i]
        step = env.step({sid: np.array([g["actions"][t, 0, i]], np.float32) for i, sid in enumerate(shops)})
        for i, sid in enumerate(shops):
            assert step.observations[sid].shape == (4,)
            np.testing.assert_array_equal(f32_bits(step.observations[sid]), f32_bits(g["obs"][t, 0, i]))
            assert step.rewards[sid] == g["reward"][t, 0, i]
        assert step.truncations["__all__"] == bool(g["all_truncated"][t, 0])


def test_market_compressed_prices_materialise_on_inject():
    """The fused market kernel keeps BuyerAgent.prices as one value per seller; a host-injected
    Price to a single buyer breaks that form: the table is materialised and the env continues on
    the generic engine -- same results as the oracle throughout."""
    from phantom_amd.message import Message
    rng = np.random.RandomState(8)
    L, Fw, d, B = 8, 32, 4, 6
    S = L + Fw
    env = market_env(L, Fw, d, 50, B)
    o, x = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); x.reset()

    nb0 = market_topology(L, Fw, d)[0][1]                       # second neighbour of buyer B0

    def step(t, mute_seller=None, force_buy=False):
        act = np.zeros((B, S), np.float32); valid = np.zeros((B, S), np.uint8)
        odd = (o.get_i32("env.step")[:, 0] + 1) % 2 == 1
        act[:, :L] = rng.randint(1, 9, size=(B, L)) / 8.0
        act[:, L:] = (rng.rand(B, Fw) < 0.7)
        if force_buy:
            act[:, L] = 1.0
        valid[odd, :L] = 1; valid[~odd, L:] = 1
        if mute_seller is not None:
            valid[:, mute_seller] = 0
        o.step(act, valid, None); x.step(act, valid, None)
        for f in ("obs_valid", "reward_valid", "err"):
            np.testing.assert_array_equal(getattr(x, f), getattr(o, f), err_msg=f"{f} t={t}")
        np.testing.assert_array_equal(f32_bits(x.obs), f32_bits(o.obs), err_msg=f"obs t={t}")
        np.testing.assert_array_equal(f64_bits(x.reward), f64_bits(o.reward), err_msg=f"rew t={t}")

    for t in range(4):
        step(t)
    assert x.dev.uses_fused
    np.testing.assert_array_equal(f64_bits(x.get_f64("buyer.prices")), f64_bits(o.get_f64("buyer.prices")))
    msg = [Message(f"S{nb0}", "B0", ph.Price(0.015625))]
    o.inject(msg); x.dev.inject(msg)
    step(4, mute_seller=nb0)                 # leaders' step: only the injected Price reaches B0's slot
    assert not x.dev.uses_fused
    step(5, force_buy=True)                  # followers' step: B0 buys at the injected price
    assert (o.get_f64("buyer.paid")[:, 0] == 0.015625).all()
    for t in range(6, 10):
        step(t)
    np.testing.assert_array_equal(f64_bits(x.get_f64("buyer.prices")), f64_bits(o.get_f64("buyer.prices")))


@pytest.mark.parametrize("device_draw", [False, True])
@pytest.mark.parametrize("force_generic", [False, True])
def test_stochastic_network_differential_vs_oracle(device_draw, force_generic):
    """StochasticNetwork (network.py:340-453): per-env connectivity resampled at every reset, fed by
    the host or drawn by the device Philox stream; market kinds iterate / check the per-env graph."""
    rng = np.random.RandomState(31)
    L, Fw, d, B, T = 6, 24, 3, 40, 36
    S = L + Fw
    np.random.seed(5)
    env = market_env(L, Fw, d, 9, B, rates=[0.8, 0.3, 1.0, 0.0, 0.55], seed=21, env_offset=100, force_generic=force_generic)
    spec = env.spec
    assert spec.n_conn == Fw * d and len(spec.col_conn) == 2 * spec.n_conn
    o, x = OracleEnv(spec), _dev(spec)
    assert x.dev.uses_fused == (not force_generic)      # the fused market kernels read the per-env connectivity too
    np.testing.assert_array_equal(x.get_u8("net.conn_on"), o.get_u8("net.conn_on"))   # constructor draw

    def reset(mask=None):
        conn = None if device_draw else (rng.rand(B, spec.n_conn) < spec.conn_rate).astype(np.uint8)
        (oo, ov), (do, dv) = o.reset(mask, None, conn), x.reset(mask, None, conn)
        m = slice(None) if mask is None else mask.astype(bool)
        np.testing.assert_array_equal(dv[m], ov[m])
        np.testing.assert_array_equal(f32_bits(do[m]), f32_bits(oo[m]))
        np.testing.assert_array_equal(x.get_u8("net.conn_on"), o.get_u8("net.conn_on"))
        np.testing.assert_array_equal(x.get_i32("env.episode"), o.get_i32("env.episode"))

    reset()
    on = o.get_u8("net.conn_on")
    assert 0.3 < on.mean() < 0.7 and (on[:, 2::5] == 1).all() and (on[:, 3::5] == 0).all()
    assert len({tuple(r) for r in on}) > B // 2                   # the envs have different graphs
    for t in range(T):
        act = np.zeros((B, S), np.float32); valid = np.zeros((B, S), np.uint8)
        odd = (o.get_i32("env.step")[:, 0] + 1) % 2 == 1
        act[:, :L] = rng.randint(1, 9, size=(B, L)) / 8.0
        act[:, L:] = (rng.rand(B, Fw) < 0.7)
        valid[odd, :L] = 1; valid[~odd, L:] = 1
        o.step(act, valid, None); x.step(act, valid, None)
        for f in ("obs_valid", "reward_valid", "done_valid", "all_truncated", "err"):
            np.testing.assert_array_equal(getattr(x, f), getattr(o, f), err_msg=f"{f} t={t}")
        assert (o.err == 0).all()
        np.testing.assert_array_equal(f32_bits(x.obs), f32_bits(o.obs), err_msg=f"obs t={t}")
        np.testing.assert_array_equal(f64_bits(x.reward), f64_bits(o.reward), err_msg=f"rew t={t}")
        for f in ("seller.tx", "buyer.bought"):
            np.testing.assert_array_equal(x.get_i32(f), o.get_i32(f), err_msg=f)
        if t % 7 == 3:      # BuyerAgent.prices: slots of connections that are off this episode keep the reset's 1.0
            np.testing.assert_array_equal(f64_bits(x.get_f64("buyer.prices")), f64_bits(o.get_f64("buyer.prices")), err_msg=f"prices t={t}")
        done = o.all_truncated.astype(np.uint8)
        if done.any():
            reset(done)


def test_stochastic_network_reference_kats():
    """tests/network/test_stochastic_network.py:13-58 through the Python surface: rate 1 keeps the
    edge, rate 0 never creates it, before and after resample_connectivity()."""
    for rate, how in [(1.0, "one"), (0.0, "one"), (0.0, "from"), (0.0, "between")]:
        net = ph.StochasticNetwork([ph.Agent("A"), ph.Agent("B")], ph.BatchResolver(2))
        if how == "one":
            net.add_connection("A", "B", rate)
        elif how == "from":
            net.add_connections_from([("A", "B", rate)])
        else:
            net.add_connections_between(["A"], ["B"], rate=rate)
        for _ in range(2):
            assert net.has_edge("A", "B") == (rate == 1.0) and net.has_edge("B", "A") == (rate == 1.0)
            net.resample_connectivity()
    # a missing edge is a NetworkError inside the step: PHX_ERR_NETWORK on the device
    import phantom_amd as ph_
    env = market_env(2, 4, 2, 4, 3, rates=[0.0], exogenous="device")
    d = env._device()
    env.reset()
    assert (d.field("net.conn_on").cpu().numpy() == 0).all()
    out = env.step(__import__("torch").ones(3, 6, device=d.device))      # tensor in -> StepTensors out
    assert (d.err.cpu().numpy() == 0).all()                      # nobody has a neighbour: nothing is sent
    assert (out.observations.cpu().numpy()[:, 2:, 0] == 1.0).all()   # buyers: min over no prices -> 1.0


@pytest.mark.parametrize("L,Fw,d,B,T,num_steps", [(8, 32, 4, 10, 45, 12), (128, 1024, 8, 6, 14, 9), (5, 7, 2, 3, 30, 7)])
def test_market_rollout_matches_oracle(L, Fw, d, B, T, num_steps):
    """phx_rollout on the Stackelberg market: T fused steps with the whole env state in LDS, random
    policy (seller price U[0,1), buyer buy/skip) or replayed actions, auto-reset -- vs the oracle."""
    rng = np.random.RandomState(L + T)
    S = L + Fw
    for mode in ("device_rng", "replay"):
        env = market_env(L, Fw, d, num_steps, B, seed=17, env_offset=5)
        o, x = OracleEnv(env.spec), _dev(env.spec)
        assert x.dev.uses_fused
        o.reset(); x.reset()
        a0 = np.zeros((B, S), np.float32); a0[:, :L] = rng.randint(1, 9, (B, L)) / 8.0
        v0 = np.zeros((B, S), np.uint8); v0[:, :L] = 1
        o.step(a0, v0, None); x.step(a0, v0, None)               # the fragment starts on a followers' step
        acts = None
        if mode == "replay":
            acts = np.where(rng.rand(T, B, S) < 0.5, rng.randint(1, 9, (T, B, S)) / 8.0, 1.0).astype(np.float32)
        ro, rd = o.rollout(T, acts, None), x.rollout(T, acts, None)
        for k in ("obs_valid", "reward_valid", "truncated", "terminated"):
            np.testing.assert_array_equal(rd[k], ro[k], err_msg=f"{k} {mode}")
        for k in ("obs", "actions", "rewards", "last_obs"):
            np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=f"{k} {mode}")
        assert ro["truncated"].any() and (ro["reward_valid"] == 1).any() and (ro["rewards"] != 0).any()
        for f in ("seller.tx", "buyer.bought", "env.step", "env.tick"):
            np.testing.assert_array_equal(x.get_i32(f), o.get_i32(f), err_msg=f"{f} {mode}")
        for f in ("seller.price", "seller.revenue", "buyer.paid", "buyer.prices"):
            np.testing.assert_array_equal(f64_bits(x.get_f64(f)), f64_bits(o.get_f64(f)), err_msg=f"{f} {mode}")
        # per-step launches continue identically after the fragment
        odd = (o.get_i32("env.step")[:, 0] + 1) % 2 == 1
        a1 = np.zeros((B, S), np.float32); v1 = np.zeros((B, S), np.uint8)
        a1[:, :L] = 0.25; a1[:, L:] = 1.0
        v1[odd, :L] = 1; v1[~odd, L:] = 1
        o.step(a1, v1, None); x.step(a1, v1, None)
        np.testing.assert_array_equal(f32_bits(x.obs), f32_bits(o.obs))
        np.testing.assert_array_equal(f64_bits(x.reward), f64_bits(o.reward))
        np.testing.assert_array_equal(x.reward_valid, o.reward_valid)


def test_rollout_with_per_env_tick_parities():
    """one Philox block serves a tick PAIR; envs of one workgroup whose tick counters differ in parity
    (possible after per-env state surgery) pair their rows differently -- still the oracle's stream."""
    B, S, K, T = 24, 9, 6, 37
    env = supply_chain_env(S, [K] * S, 10, B, seed=6, env_offset=11)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    ticks = (np.arange(B, dtype=np.int32) * 7 + 3) % 23            # mixed parities inside every block
    o.set_i32("env.tick", ticks); d.set_i32("env.tick", ticks.reshape(B, 1))
    ro, rd = o.rollout(T, None, None), d.rollout(T, None, None)
    for k in ("obs", "actions", "rewards", "last_obs"):
        np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=k)
    np.testing.assert_array_equal(rd["truncated"], ro["truncated"])
    np.testing.assert_array_equal(d.get_i32("env.tick")[:, 0], ticks + T)


def test_fuzz_regressions():
    """cases found by the random differential fuzz (scratch/fuzz.py): (1) buyers / sellers without
    any neighbour in the fused market kernels, (2) a rollout whose workgroup owns only 4 pairs."""
    # (1) isolated agents: min over no prices is 1.0, tx / len(neighbours) is 0
    agents = [ph.SellerAgent("S0"), ph.SellerAgent("S1"), ph.BuyerAgent("B0", 0.5), ph.BuyerAgent("B1", 0.75),
              ph.BuyerAgent("B2", 0.25)]
    net = ph.Network(agents)
    net.add_connection("B0", "S0"); net.add_connection("B1", "S0")          # S1 and B2 are isolated
    env = ph.StackelbergEnv(6, net, ["S0", "S1"], ["B0", "B1", "B2"], batch_size=5, seed=3)
    o, x = OracleEnv(env.spec), _dev(env.spec)
    assert x.dev.uses_fused
    o.reset(); x.reset()
    rng = np.random.RandomState(2)
    for t in range(8):
        act = rng.randint(1, 9, (5, 5)).astype(np.float32) / 8.0
        act[:, 2:] = 1.0
        o.step(act, None, None); x.step(act, None, None)
        np.testing.assert_array_equal(f32_bits(x.obs), f32_bits(o.obs), err_msg=f"t={t}")
        np.testing.assert_array_equal(f64_bits(x.reward), f64_bits(o.reward))
        if o.all_truncated.any():
            o.reset(); x.reset()
    ro, rd = o.rollout(9), x.rollout(9)
    np.testing.assert_array_equal(f32_bits(rd["obs"]), f32_bits(ro["obs"]))
    np.testing.assert_array_equal(f32_bits(rd["rewards"]), f32_bits(ro["rewards"]))
    # (2) B = 2 envs x 2 shops: one workgroup of 4 pairs
    env = supply_chain_env(2, [2, 3], 6, 2, seed=4)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    ro, rd = o.rollout(31), d.rollout(31)
    for k in ("obs", "actions", "rewards", "last_obs"):
        np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=k)
    np.testing.assert_array_equal(rd["truncated"], ro["truncated"])


def test_env_supertype_and_dict_vs_tensor_api():
    """tests/test_supertypes_env.py:86-135 (env supertype: sampled in the constructor and at every
    reset, env_type is None before the first reset) and the dict API against the tensor API."""
    import torch
    from dataclasses import dataclass

    class MockSampler(ph.Sampler):
        def __init__(self, value):
            self._value = value

        def sample(self):
            self._value += 1
            return self._value

    class MockEnv(ph.PhantomEnv):
        @dataclass
        class Supertype(ph.Supertype):
            type_value: float = 0.0

        def __init__(self, **kw):
            super().__init__(num_steps=5, network=ph.Network([ph.MockStrategicAgent("a1")]), **kw)

    s1 = MockSampler(0)
    env = MockEnv(env_supertype=MockEnv.Supertype(type_value=s1))
    assert env._samplers == [s1] and env.env_type is None
    env.reset()
    assert env.env_type == MockEnv.Supertype(2)
    env2 = MockEnv(env_supertype={"type_value": MockSampler(0)})
    env2.reset()
    assert env2.env_type == MockEnv.Supertype(2)
    with pytest.raises(Exception):
        MockEnv(env_supertype={"xxx": 0.0})

    # dict API == tensor API (B = 3): same kernel launches, the dicts are views of the tensors
    rng = np.random.RandomState(1)
    ea = ph.SupplyChainEnv(n_shops=3, customers_per_shop=[2, 3, 1], num_steps=4, batch_size=3, seed=5, exogenous="device")
    eb = ph.SupplyChainEnv(n_shops=3, customers_per_shop=[2, 3, 1], num_steps=4, batch_size=3, seed=5, exogenous="device")
    oa, _ = ea.reset(); eb.reset()
    for t in range(9):
        act = rng.uniform(0, 100, (3, 3)).astype(np.float32)
        sa = ea.step({f"SHOP{i}": act[:, i] for i in range(3)})
        sb = eb.step(torch.from_numpy(act).to(eb._device().device))
        obs_b, rew_b = sb.observations.cpu().numpy(), sb.rewards.cpu().numpy()
        for i in range(3):
            np.testing.assert_array_equal(sa.observations[f"SHOP{i}"], obs_b[:, i])
            np.testing.assert_array_equal(sa.rewards[f"SHOP{i}"], rew_b[:, i])
        assert bool(np.all(sa.truncations["__all__"])) == bool(sb.all_truncated.cpu().numpy().all())
        if np.all(sa.truncations["__all__"]):
            ea.reset(); eb.reset()


def test_launch_loop_rollout_for_generic_engine_envs():
    """phx_rollout on envs without a fused rollout kernel (stochastic market; supply chain forced onto
    the generic engine): the stream-ordered launch loop vs the oracle, incl. auto-reset redraws."""
    for force_generic in (True, False):        # launch loop on the generic engine; fused market rollout with in-kernel redraws
        env = market_env(5, 14, 3, 6, 9, rates=[0.8, 0.3, 1.0, 0.0, 0.55], seed=4, env_offset=2, exogenous="device",
                         force_generic=force_generic)
        o, x = OracleEnv(env.spec), _dev(env.spec)
        assert x.dev.uses_fused == (not force_generic)
        o.reset(); x.reset()
        ro, rd = o.rollout(20), x.rollout(20)
        for k in ("obs_valid", "reward_valid", "truncated", "terminated"):
            np.testing.assert_array_equal(rd[k], ro[k], err_msg=k)
        for k in ("obs", "actions", "rewards", "last_obs"):
            np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=k)
        np.testing.assert_array_equal(x.get_u8("net.conn_on"), o.get_u8("net.conn_on"))     # redrawn 3 times
        np.testing.assert_array_equal(x.get_i32("env.episode"), o.get_i32("env.episode"))
        assert ro["truncated"].sum() > 0
    env = supply_chain_env(4, [2, 7, 0, 3], 5, 11, force_generic=True, seed=9, norm_customers=3)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    rng = np.random.RandomState(0)
    acts = rng.uniform(-10, 120, (17, 11, 4)).astype(np.float32)
    exo = rng.randint(0, 5, (17, 11, 12)).astype(np.uint8)
    for a_, x_ in ((None, None), (acts, exo)):
        ro, rd = o.rollout(17, a_, x_), d.rollout(17, a_, x_)
        for k in ("obs", "actions", "rewards", "last_obs"):
            np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=k)
        np.testing.assert_array_equal(rd["truncated"], ro["truncated"])
    for f in ("shop.stock", "shop.sales", "env.step", "env.tick"):
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)


def test_host_clock_mirrors_after_rollout():
    """env.current_step / current_stage / previous_stage follow the device state after env.rollout()."""
    env = ph.SupplyChainFSMEnv(n_shops=2, customers_per_shop=2, num_steps=6, batch_size=3, seed=1, exogenous="device")
    env.reset()
    env.rollout(9)                                  # 6 steps + auto-reset + 3 steps: stage SELL after an odd count
    assert list(env.current_step) == [3, 3, 3]
    assert env.current_stage == ["SELL"] * 3 and env.previous_stage == ["RESTOCK"] * 3
    import torch
    out = env.step(torch.zeros(3, 2, device=env._device().device))
    assert list(env.current_step) == [4, 4, 4] and env.current_stage == ["RESTOCK"] * 3
    assert int(out.obs_valid.sum()) == 6            # shops observe on the SELL -> RESTOCK transition


def test_digital_ads_env_python_surface():
    """ph.DigitalAdsEnv built like digital_ads_market.py:525-596 and driven through the dict API: Dict
    observations, None rewards before the first auction, attribute reflection for the metrics of
    :603-713, and a device-policy rollout against the oracle."""
    st = {f"ADV_{i + 1}": ph.AdvertiserAgent.Supertype(budget=b) for i, b in enumerate([1.5, 2.0, 0.75, 1.0])}
    env = ph.DigitalAdsEnv(num_steps=8, num_agents_theme={"travel": 2, "tech": 1, "sport": 1}, agent_supertypes=st, seed=3)
    assert env.agent_ids[:2] == ["ADX", "PUB"] and env.strategic_agent_ids == [f"ADV_{i}" for i in range(1, 5)]
    obs, _ = env.reset()
    assert obs == {} and env.current_stage == "publisher_step"           # the publisher is not strategic
    step = env.step({})
    assert env.current_stage == "advertiser_step" and set(step.observations) == set(env.strategic_agent_ids)
    o1 = step.observations["ADV_1"]
    assert set(o1) == {"type", "budget_left", "user_id"} and o1["type"]["budget"].dtype == np.float32
    assert o1["budget_left"].dtype == np.float64 and float(o1["budget_left"][0]) == 1.0 and o1["user_id"] in (0, 1)
    assert step.rewards == {aid: None for aid in env.strategic_agent_ids}   # fsm.py:234,378: nothing cached yet
    user = env["ADV_1"]._current_user_id
    assert user in (1, 2) and int(env["ADV_1"].total_requests[user]) == 1
    assert env["ADV_1"]._current_age == {1: 18, 2: 40}[user] and env["ADX"].view("ADV_1").users_info[2]["zipcode"] == 90250
    step = env.step({aid: np.array([0.5], np.float32) for aid in env.strategic_agent_ids})
    lefts = [env[aid].left for aid in env.strategic_agent_ids]
    wins = [env[aid].step_wins for aid in env.strategic_agent_ids]
    assert sum(wins) == 1 and wins[1] == 1                                # ADV_2 bids 0.5 * 2.0, the highest
    assert lefts[1] == float(np.float32(2.0) - np.float32(1.0)) and lefts[0] == 1.5
    assert env["ADV_2"].bid == 1.0 and step.observations == {}           # publishers act next: nobody observes
    # the example's metrics (:603-713) reflect on agent attributes: SimpleAgentMetric(aid, "left", "mean") etc.
    assert ph.metrics.SimpleAgentMetric("ADV_2", "left", "mean").extract(env) == lefts[1]
    assert ph.metrics.SimpleAgentMetric("ADV_2", "step_wins", "mean").extract(env) == 1
    assert int(env["ADV_2"].total_wins[user]) == 1 and int(env["ADV_1"].total_wins[user]) == 0
    # device-policy rollout (launch loop: FSM env on the generic engine) against the oracle
    env2 = ph.DigitalAdsEnv(num_steps=8, num_agents_theme={"travel": 2, "tech": 1, "sport": 1}, agent_supertypes=st,
                            seed=3, batch_size=16, connection_rates=(1.0, 0.8, 0.9))
    o, d = OracleEnv(env2.spec), _dev(env2.spec)
    o.reset(); d.reset()
    ro, rd = o.rollout(30), d.rollout(30)
    for k in ("obs_valid", "reward_valid", "truncated", "terminated"):
        np.testing.assert_array_equal(rd[k], ro[k], err_msg=k)
    for k in ("obs", "actions", "rewards", "last_obs"):
        np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=k)
    assert ro["rewards"].sum() > 0 and (ro["obs_valid"] == 0).any()
    for f in ("adv.total_clicks", "adv.total_wins", "adv.left_tag"):
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)

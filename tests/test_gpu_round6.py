"""Round 6 on the GPU, part 1 (VERDICT r5 "Next round" #1): the parity hole at the timed shapes -- the exact call bench.py times (SC64,
B = 4096, 8 fragments x T = 100), config 4's share as 4 x 100 fragments at B = 8192, vouched replays at B = 4096 / T = 400 -- each against
the oracle at FULL size, and a replay hint that does not hold reported as PHX_ERR_HINT instead of silently unspecified rows."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from phantom_amd import _abi
from device_runner import DeviceRunner
from helpers import f32_bits, supply_chain_env
from oracle import OracleEnv

pytestmark = pytest.mark.gpu
NCPU = min(os.cpu_count() or 1, 128)
STATE = ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick")


def _planes(tr):
    return dict(obs=tr.observations.cpu().numpy(), actions=tr.actions.cpu().numpy(), rewards=tr.rewards.cpu().numpy(),
                terminated=None if tr.terminations is None else tr.terminations.cpu().numpy(), truncated=tr.truncations.cpu().numpy())


def test_the_call_the_bench_times_every_row_of_every_fragment_against_the_oracle():
    """bench.py's timed call: SC64 (9 shops x 6 customers), B = 4096, num_steps = 100, ``rollout_fragments(100, 8 buffers)`` = 800 steps
    in ONE phx_sc_rollout_sw_kernel launch on 256 workgroups (XCD-aware pair-range mapping, base-pointer switches at every fragment
    boundary at full grid): every row of every plane of every fragment, ``last_obs`` and the state the call leaves, bit for bit; twice
    (the second call starts from the first one's state, as in the bench's timed loop, and rotates to a second buffer set)."""
    B, S, K, Tf, k = 4096, 9, 6, 100, 8
    env = supply_chain_env(S, [K] * S, 100, B, seed=42)                 # bench.py: seed 42, the library's own choice of kernel
    o, d = OracleEnv(env.spec, threads=NCPU), DeviceRunner(env.spec)
    o.reset(); d.reset()
    dev = d.dev
    sets = [[dev.alloc_trajectory(Tf) for _ in range(k)] for _ in range(2)]
    for rep in range(2):
        outs = sets[rep]
        for t in outs:
            for x in t[:5]:
                x.fill_(7)                                               # garbage the launch has to overwrite
        got = dev.rollout_fragments(Tf, outs)
        assert dev.last_kernel() == "phx_sc_rollout_sw_kernel", dev.last_kernel()
        assert all(g.last_obs is None for g in got[:-1]) and got[-1].last_obs is outs[-1].last_obs
        ro = o.rollout(k * Tf)
        for i, t in enumerate(outs):
            p = _planes(t)
            lo, hi = i * Tf, (i + 1) * Tf
            for key in ("obs", "actions", "rewards"):
                np.testing.assert_array_equal(f32_bits(p[key]), f32_bits(ro[key][lo:hi]), err_msg=f"rep {rep} fragment {i}: {key}")
            np.testing.assert_array_equal(p["truncated"], ro["truncated"][lo:hi], err_msg=f"rep {rep} fragment {i}: truncated")
            np.testing.assert_array_equal(p["terminated"], ro["terminated"][lo:hi], err_msg=f"rep {rep} fragment {i}: terminated")
        np.testing.assert_array_equal(f32_bits(outs[-1].last_obs.cpu().numpy()), f32_bits(ro["last_obs"]))
        for f in STATE:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after rep {rep}")
    assert (dev.err.cpu().numpy() == 0).all()


def test_config4_share_as_four_fragments_full_size_sample_and_invariants():
    """BASELINE config 4's per-GPU share (SC256: 51 shops x 4 customers, B = 8192) collected as 4 x 100-step fragments from one launch
    (3 264 groups of 128 pairs walked by 256 persistent workgroups): the first 512 envs of every fragment row by row against an oracle
    of those envs (the device RNG is keyed by the global env index), the whole batch through the planes' closed forms (one truncation
    per pair and episode at the episode's last row, no termination, actions in [0, 100), stock/100 observations in [0, 1]) and the
    step counters the call leaves."""
    S, K, B, ns, Tf, k, Bo = 51, 4, 8192, 100, 100, 4, 512
    env = supply_chain_env(S, [K] * S, ns, B, seed=42)
    d = DeviceRunner(env.spec); d.reset()
    rd = d.rollout_fragments(Tf, k)
    assert d.dev.last_kernel() == "phx_sc_rollout_sw_kernel", d.dev.last_kernel()
    envo = supply_chain_env(S, [K] * S, ns, Bo, seed=42)
    o = OracleEnv(envo.spec, threads=NCPU); o.reset()
    ro = o.rollout(k * Tf)
    for key in ("obs", "actions", "rewards"):
        np.testing.assert_array_equal(f32_bits(rd[key][:, :Bo]), f32_bits(ro[key]), err_msg=key)
    for key in ("truncated", "terminated"):
        np.testing.assert_array_equal(rd[key][:, :Bo], ro[key], err_msg=key)
    np.testing.assert_array_equal(f32_bits(rd["last_obs"][:Bo]), f32_bits(ro["last_obs"]))
    for f in STATE:
        np.testing.assert_array_equal(d.get_i32(f)[:Bo], o.get_i32(f), err_msg=f)
    tr = rd["truncated"]
    assert tr.sum() == k * B * S and all((tr[j * ns + ns - 1] == 1).all() for j in range(k)) and not rd["terminated"].any()
    assert (rd["actions"] >= 0).all() and (rd["actions"] < 100).all()
    assert (rd["obs"][..., 0] >= 0).all() and (rd["obs"][..., 0] <= 1).all()
    assert (d.get_i32("env.step") == 0).all() and (d.get_i32("env.tick") == k * Tf).all()
    assert (d.err == 0).all()


@pytest.mark.parametrize("what", ["actions", "both"])
def test_vouched_replays_at_the_bench_batch_full_size(what):
    """Replayed policy (and order sizes) the caller vouches for through the store-wave kernel's REPLAY instantiation at the bench's
    batch: SC64, B = 4096, T = 400 -- every row against the oracle (round 5 ran replays at B <= 128)."""
    B, S, K, T = 4096, 9, 6, 400
    env = supply_chain_env(S, [K] * S, 100, B, seed=42)
    o, d = OracleEnv(env.spec, threads=NCPU), DeviceRunner(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(6)
    acts = rng.uniform(0, 100, (T, B, S)).astype(np.float32)
    acts[rng.random((T, B, S)) < 0.05] = 0.5
    acts[rng.random((T, B, S)) < 0.05] = 117.25
    exo = rng.integers(0, 5, (T, B, d.n_exo)).astype(np.uint8) if what == "both" else None
    rd = d.rollout(T, acts, exo, actions_in_domain=True, exo_in_domain=exo is not None)
    assert d.dev.last_kernel() == "phx_sc_rollout_sw_kernel[replay]", d.dev.last_kernel()
    ro = o.rollout(T, acts, exo)
    for key in ("obs", "actions", "rewards"):
        np.testing.assert_array_equal(f32_bits(rd[key]), f32_bits(ro[key]), err_msg=key)
    np.testing.assert_array_equal(rd["truncated"], ro["truncated"])
    np.testing.assert_array_equal(f32_bits(rd["last_obs"]), f32_bits(ro["last_obs"]))
    for f in STATE:
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)
    assert (d.err == 0).all()


@pytest.mark.parametrize("S,K,B,T", [(9, 6, 64, 48), (51, 4, 128, 64), (3, 2, 48, 41)])
def test_a_replay_hint_that_does_not_hold_is_reported_per_env(S, K, B, T):
    """PHX_RH_ACTIONS_IN_DOMAIN with an action that rounds below zero, PHX_RH_EXO_IN_DOMAIN with an order byte >= 5: the store-wave
    kernel relies on both (byte tiles) -- err[b] == PHX_ERR_HINT for exactly the envs whose inputs break the promise, every other env's
    rows are the oracle's; a following honest call is clean again after the caller's reset."""
    env = supply_chain_env(S, [K] * S, 30, B, seed=8, variants={"rollout": "store_waves"})
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    rng = np.random.default_rng(S + T)
    for what in ("actions", "exo", "both", "honest"):
        o.reset(); d.reset()
        acts = rng.uniform(0, 100, (T, B, S)).astype(np.float32)
        exo = rng.integers(0, 5, (T, B, d.n_exo)).astype(np.uint8)
        bad = np.zeros(B, bool)
        if what in ("actions", "both"):
            for b, t, s, v in ((3, 5, 0, -0.51), (B - 1, T - 1, S - 1, -42.0), (B // 2, 0, S // 2, -1e9)):
                acts[t, b, s] = v; bad[b] = True
        if what in ("exo", "both"):
            for b, t, c, v in ((1, 7, 0, 5), (B - 2, T - 2, d.n_exo - 1, 255), (B // 2 + 1, 1, d.n_exo // 2, 7), (7, 3, 1, 29), (9, 11, 2, 13)):
                exo[t, b, c] = v; bad[b] = True
        rd = d.rollout(T, acts, exo, actions_in_domain=True, exo_in_domain=True)
        assert d.dev.last_kernel() == "phx_sc_rollout_sw_kernel[replay]", d.dev.last_kernel()
        np.testing.assert_array_equal(d.err != 0, bad, err_msg=what)
        assert (d.err[bad] == _abi.ERR_HINT).all()
        ro = o.rollout(T, acts, exo)
        good = ~bad
        for key in ("obs", "actions", "rewards"):
            np.testing.assert_array_equal(f32_bits(rd[key][:, good]), f32_bits(ro[key][:, good]), err_msg=f"{what}: {key}")
        np.testing.assert_array_equal(d.get_i32("shop.stock")[good], o.get_i32("shop.stock")[good])


def test_scale_node_script_dry_run_on_one_gpu(tmp_path):
    """tools/scale_node.sh --dry-run: the first-contact script for an 8-GPU node with every rank on the GPU that exists
    (PHX_BENCH_SHARE_GPU): each stage leaves one JSON line (a bench line or an error line), the 1 / 2 / 8-rank curve stages report a `value`
    over the gloo control plane, the NCCL_ALGO stage records its environment, and the summary tabulates them."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    out = str(tmp_path / "scale")
    p = subprocess.run(["bash", os.path.join(root, "tools", "scale_node.sh"), "--dry-run", out], capture_output=True, text=True, timeout=2400,
                       env={**os.environ, "STAGE_TIMEOUT": "420"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = {}
    for name in ("curve_n1", "curve_n2", "curve_n8", "gather_ring_n8"):
        lines[name] = json.load(open(os.path.join(out, name + ".json")))
    for name, n in (("curve_n1", 1), ("curve_n2", 2), ("curve_n8", 8)):
        d = lines[name]
        assert d.get("n_gpus") == n and d.get("value") and d["value"] > 0, (name, d.get("error"), p.stdout[-1500:])
    assert lines["gather_ring_n8"].get("n_gpus") == 8
    summ = open(os.path.join(out, "summary.txt")).read()
    assert "== scaling" in summ and "N=8" in summ and "NCCL_ALGO=Ring" in summ


def test_golden_state_handler_through_the_fused_rule_rollout():
    """golden `sc_fsm_state_handler` (the REFERENCE running a RESTOCK handler that resolves the network and branches on the shops' total
    stock) through ONE phx_rollout with the handler in rule form: the fused lane-per-pair loop evaluates the rule -- the
    reference's observations, rewards, validity bits, truncations and final stage over 70 steps and three episodes, actions and order sizes
    replayed; the same call on the message-passing engine's compiled schedule (force_generic) agrees."""
    import phantom_amd as ph
    from helpers import env_from_golden, golden, golden_stock_handler
    g = golden("sc_fsm_state_handler")
    T, B = int(g["T"]), len(g["seeds"])
    assert g["reset_before"][[0, 30, 60]].all() and g["reset_before"].sum() == 3 * B      # resets exactly at the episode ends: a rollout's auto-reset
    for generic in (False, True):
        handler = ph.state_rules([ph.StageRule("shop.stock", "<", 60, "RESTOCK")])(lambda env: golden_stock_handler(env))
        env = env_from_golden({k: g[k] for k in g.files if k != "next_stage"}, restock_handler=handler, force_generic=generic)
        d = DeviceRunner(env.spec); d.reset()
        rd = d.rollout(T, g["actions"], g["exo"])
        want = "phx_sched_step_kernel[T-step loop]" if generic else "phx_sc_rollout_fsm_kernel[rules]"
        assert d.dev.last_kernel() == want, d.dev.last_kernel()
        np.testing.assert_array_equal(rd["obs_valid"], g["obs_valid"]); np.testing.assert_array_equal(rd["reward_valid"], g["reward_valid"])
        m = g["obs_valid"].astype(bool)
        np.testing.assert_array_equal(f32_bits(rd["obs"][m]), f32_bits(g["obs"][m]))
        m = g["reward_valid"] == 1
        np.testing.assert_array_equal(f32_bits(rd["rewards"][m]), f32_bits(g["reward"][m].astype(np.float32)))
        np.testing.assert_array_equal(rd["truncated"], g["truncated"] | g["all_truncated"][:, :, None])
        np.testing.assert_array_equal(d.get_i32("shop.stock")[:, :], 0 * g["stock"][-1] if g["all_truncated"][-1].all() else g["stock"][-1])
        assert (d.err == 0).all()


@pytest.mark.parametrize("S,ks,B,ns,rules", [
    (9, [6] * 9, 200, 17, [("shop.stock", "<", 300, "RESTOCK", None)]),
    (51, [4] * 51, 23, 12, [("shop.sales", ">=", 40, "SELL", None), ("shop.stock", "<", 1500, "RESTOCK", None)]),
    (3, [2, 3, 1], 64, 9, [("shop.stock", "<=", 20, "RESTOCK", "SHOP1"), ("shop.missed_sales", ">", 3, "RESTOCK", None)]),
    (4, [3] * 4, 129, 10, [("shop.delivered_stock", "==", 0, "RESTOCK", "SHOP0")]),
])
def test_fused_rule_rollout_matches_the_oracle_and_the_engine(S, ks, B, ns, rules):
    """phx_sc_rollout_fsm_kernel<RULES>: several rules per stage (the first that holds decides), single-agent columns and sums, every
    comparison operator family, batches that do not fill the last workgroup, episode ends and fragments that continue from the state
    the previous one left -- every plane against the oracle, and the engine's compiled schedule (force_generic) on the same spec."""
    import phantom_amd as ph
    def build(**kw):
        h = ph.state_rules([ph.StageRule(f, c, th, nx, agent=ag) for f, c, th, nx, ag in rules])(lambda env: None)
        h._phx_skip_check = True
        env = supply_chain_env(S, ks, ns, B, fsm=True, seed=13, restock_handler=h, **kw)
        env._rules_checked = True
        return env
    env, envg = build(), build(force_generic=True)
    o, d, dg = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec), DeviceRunner(envg.spec)
    o.reset(); d.reset(); dg.reset()
    stages = set()
    for T in (2 * ns + 3, 7, 1):
        ro = o.rollout(T)
        rd = d.rollout(T); assert d.dev.last_kernel() == "phx_sc_rollout_fsm_kernel[rules]", d.dev.last_kernel()
        rg = dg.rollout(T); assert dg.dev.last_kernel() == "phx_sched_step_kernel[T-step loop]", dg.dev.last_kernel()
        for r, what in ((rd, "fused"), (rg, "engine")):
            np.testing.assert_array_equal(r["obs_valid"], ro["obs_valid"], err_msg=what); np.testing.assert_array_equal(r["reward_valid"], ro["reward_valid"], err_msg=what)
            m = ro["obs_valid"].astype(bool)
            np.testing.assert_array_equal(f32_bits(r["obs"][m]), f32_bits(ro["obs"][m]), err_msg=what)
            m = ro["reward_valid"] == 1
            np.testing.assert_array_equal(f32_bits(r["rewards"][m]), f32_bits(ro["rewards"][m]), err_msg=what)
            np.testing.assert_array_equal(f32_bits(r["actions"]), f32_bits(ro["actions"]), err_msg=what)
            np.testing.assert_array_equal(r["truncated"], ro["truncated"], err_msg=what)
            np.testing.assert_array_equal(f32_bits(r["last_obs"]), f32_bits(ro["last_obs"]), err_msg=what)
        for f in STATE + ("env.stage", "env.prev_stage"):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f); np.testing.assert_array_equal(dg.get_i32(f), o.get_i32(f), err_msg=f)
        stages |= set(np.unique(o.get_i32("env.stage")).tolist())
    assert (d.err == 0).all() and (dg.err == 0).all()


@pytest.mark.parametrize("L,Fw,deg,B", [(6, 20, 3, 37), (40, 300, 8, 11), (128, 256, 8, 5)])
def test_market_rollout_from_posted_prices_written_by_hand(L, Fw, deg, B):
    """The fused market rollout searches a buyer's cheapest neighbour on 4-byte keys: the posted price itself where every posted price of the
    env is a float32 value (1.0 after a reset, a seller's float32 action afterwards), RANK keys where a caller wrote other doubles into
    `seller.posted`.  Both against the oracle -- float32 values, full-precision doubles, ties among them, coarse ties -- over
    the sellers' first postings, an episode end and a second fragment."""
    from helpers import f64_bits, market_env
    ns = 8
    env = market_env(L, Fw, deg, ns, B, seed=9, exogenous="device")
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(L + B)
    p = rng.random((B, L))                                    # 53-bit doubles: not float32 values
    p[0] = p[0].astype(np.float32)                            # env 0: float32 values (the key is the price)
    p[1, ::2] = p[1, 1]                                       # ties at full precision
    p[2] = np.round(p[2], 1)                                  # coarse ties of non-float32 values (0.1, 0.2, ...)
    p[3] = 1.0 + p[3] * 2.0 ** -40                            # differences far below float32 resolution
    o.set_f64("seller.posted", p); d.set_f64("seller.posted", p)
    for T in (1, 3, 2 * ns + 1):
        ro, rd = o.rollout(T), d.rollout(T)
        assert d.dev.last_kernel() == "phx_stk_rollout_kernel"
        for k in ("obs", "actions", "rewards", "last_obs"):
            np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=f"T={T} {k}")
        for k in ("truncated", "terminated", "obs_valid", "reward_valid"):
            np.testing.assert_array_equal(rd[k], ro[k], err_msg=f"T={T} {k}")
        for f in ("seller.price", "seller.revenue", "buyer.paid"):
            np.testing.assert_array_equal(f64_bits(d.get_f64(f)), f64_bits(o.get_f64(f)), err_msg=f"T={T} {f}")
        for f in ("seller.tx", "buyer.bought"):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"T={T} {f}")
    assert (d.err == 0).all()


def test_a_fully_read_poll_result_keeps_its_rows_while_it_is_referenced():
    """RLlib's sampler reads EVERY dict of a poll() result.  Until late round 6 the host block a result's observation rows alias was released
    as soon as all six dicts had been built (the only references to the result's token were the fill functions, dropped after use): the next
    step's first read recycled the block and the kept result's observations became the new step's.  The observation MultiEnvDict now holds
    the token; rows and views are valid while the MultiEnvDict they came from is referenced (the rows of a block are made once, with its views)."""
    import torch
    from phantom_amd.rllib import BatchedBaseEnv
    B, S = 64, 3
    env = supply_chain_env(S, [2] * S, 10, B, seed=1)
    be = BatchedBaseEnv(env)
    res = be.poll()
    ids = list(env.strategic_agent_ids)

    def read_all(r):
        for dct in r[:5]:
            for b in range(B):
                row = dct[b]
                for k in row:
                    row[k]
    read_all(res)
    kept = []
    rng = np.random.default_rng(2)
    for i in range(6):
        be.send_action_tensor(torch.from_numpy(rng.uniform(0, 100, (B, S)).astype(np.float32)))
        r = be.poll(); read_all(r)
        kept.append((r, {b: {aid: r[0][b][aid].copy() for aid in ids} for b in (0, 5, B - 1)}, [r[1][b][ids[0]] for b in range(B)]))
    for r, want, rew in kept:                                     # every kept result still shows ITS step
        for b, row in want.items():
            for aid in ids:
                assert np.array_equal(r[0][b][aid], row[aid]), (b, aid)
        assert [r[1][b][ids[0]] for b in range(B)] == rew
    assert len({id(r[0][0]) for r, _, _ in kept}) == len(kept)    # (six live results: six blocks, six sets of rows)
    del kept, r
    import gc; gc.collect()
    n_blocks = len(be._blocks)
    for i in range(4):                                            # nothing is kept any more: the blocks -- and their rows -- are reused
        be.send_action_tensor(torch.zeros(B, S)); r = be.poll(); read_all(r); del r
    assert len(be._blocks) == n_blocks

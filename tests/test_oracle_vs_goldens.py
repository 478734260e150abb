"""CPU: pin the C oracle (oracle/phx_oracle.c) to golden vectors produced by running the real
reference (tests/golden/gen_goldens.py).  Integers/routing/stages bit-exact; the float obs and
rewards are compared by BIT PATTERN as well (stricter than the 1e-6 the task allows)."""
import numpy as np
import pytest

from helpers import (ads_env_from_golden, env_from_golden, f32_bits, f64_bits, golden, log_matrix, market_env)
from oracle import OracleEnv
from phantom_amd import _abi

SC_CASES = ["sc7_fixed20", "sc7_mixed", "sc64", "sc_ragged", "sc256_fsm", "sc_fsm_small",
            "sc_typed", "sc_typed_fsm"]


def stock_handler_stage(run, threshold=60):
    """the RESTOCK handler of tests/golden/gen_goldens_fsm_state.py on a runner's state: called AFTER the step's acting phase
    and resolve_network() (fsm.py:294-302): restock again (stage 0) while the shops together hold fewer than 60 items"""
    total = run.get_i32("shop.stock").sum(axis=1)
    cur = run.get_i32("env.stage")[:, 0]
    return np.where(cur == 0, np.where(total < threshold, 0, 1), 0).astype(np.int32)       # SELL's only next stage is RESTOCK


def rule_form_env(g, **kw):
    """the env of golden `sc_fsm_state_handler` with its RESTOCK handler declared in RULE form (phx_spec.stage_rules, ABI 9): restock
    again while the shops together hold fewer than 60 items, else next_stages[0] = SELL"""
    import phantom_amd as ph
    from helpers import golden_stock_handler
    handler = ph.state_rules([ph.StageRule("shop.stock", "<", 60, "RESTOCK")])(lambda env: golden_stock_handler(env))
    return env_from_golden({k: g[k] for k in g.files if k != "next_stage"}, tracking=int(g["n_logs"]) > 0, restock_handler=handler, **kw)


def replay_supply_chain(g, make_runner, tabulated_handlers=False, state_handler=None, rule_handlers=False):
    """drive a runner (oracle or device adapter) with the golden inputs and compare outputs.  ``tabulated_handlers``: the
    golden's stage handler is declared state-independent and travels as phx_spec.stage_tab instead of a per-step
    next_stage column.  ``rule_handlers``: the handler reads agent state and travels as phx_spec.stage_rules -- ONE step call per
    step, no stage from the host."""
    T, B = int(g["T"]), len(g["seeds"])
    typed = "type_src" in g
    env = rule_form_env(g) if rule_handlers else env_from_golden(g, tracking=int(g["n_logs"]) > 0, tabulated_handlers=tabulated_handlers)
    if rule_handlers:
        assert env.spec.stage_rules and env.spec.stage_tab is None
    if tabulated_handlers:
        assert env.spec.stage_tab is not None and env.spec.stage_tab.shape == (2, int(g["num_steps"]) + 1)
    run = make_runner(env.spec)
    assert run.D == (4 if typed else 3)
    for t in range(T):
        rb = g["reset_before"][t]
        if rb.any():
            # typed goldens: the values the reference's Samplers returned at this reset
            obs, valid = run.reset(rb, g["sampler_values"][t]) if typed else run.reset(rb)
            if typed:
                np.testing.assert_array_equal(f64_bits(run.get_f64("env.sampler")[rb.astype(bool)]),
                                              f64_bits(g["sampler_values"][t][rb.astype(bool)]))
            m = rb.astype(bool)
            np.testing.assert_array_equal(valid[m], g["reset_obs_valid"][t][m])
            sel = g["reset_obs_valid"][t].astype(bool) & m[:, None]
            np.testing.assert_array_equal(f32_bits(obs[sel]), f32_bits(g["reset_obs"][t][sel]))
        exo = g["exo"][t]
        if "shuffle" in g:                 # the reference's np.random.shuffle outcomes of this step, replayed
            cap = 8 * env.spec.queue_cap
            sh = np.zeros((B, cap), np.uint16)
            n = int(g["shuffle_n"][t].max())
            assert n <= cap
            sh[:, :n] = g["shuffle"][t][:, :n]
            run.step(g["actions"][t], None, exo, sh)
        elif state_handler is not None:                      # a handler that reads agent state: the step in two halves around it
            run.step_begin(g["actions"][t], None, exo)
            assert (run.err == 0).all()
            nxt = state_handler(run)
            np.testing.assert_array_equal(nxt, g["next_stage"][t], err_msg=f"the handler's stage at t={t} (it must see the RESOLVED state)")
            run.step_end(nxt)
        elif rule_handlers:                                  # the rules are evaluated inside the step, on the resolved state
            run.step(g["actions"][t], None, exo)
        elif "next_stage" in g and not tabulated_handlers:   # the stage the reference's handler returned (fsm.py:294-302), per env
            run.step(g["actions"][t], None, exo, next_stage=g["next_stage"][t])
        else:
            run.step(g["actions"][t], None, exo)
        assert (run.err == 0).all()
        np.testing.assert_array_equal(run.get_i32("shop.stock"), g["stock"][t], err_msg=f"stock t={t}")
        np.testing.assert_array_equal(run.get_i32("shop.sales"), g["sales"][t])
        np.testing.assert_array_equal(run.get_i32("shop.missed_sales"), g["missed"][t])
        np.testing.assert_array_equal(run.obs_valid, g["obs_valid"][t], err_msg=f"obs_valid t={t}")
        np.testing.assert_array_equal(run.reward_valid, g["reward_valid"][t], err_msg=f"rv t={t}")
        np.testing.assert_array_equal(run.done_valid, g["done_valid"][t])
        ov = g["obs_valid"][t].astype(bool)
        np.testing.assert_array_equal(f32_bits(run.obs[ov]), f32_bits(g["obs"][t][ov]), err_msg=f"obs t={t}")
        rv = g["reward_valid"][t] == 1
        np.testing.assert_array_equal(f64_bits(run.reward[rv]), f64_bits(g["reward"][t][rv]))
        np.testing.assert_array_equal(run.terminated, g["terminated"][t])
        np.testing.assert_array_equal(run.truncated, g["truncated"][t])
        np.testing.assert_array_equal(run.all_terminated, g["all_terminated"][t])
        np.testing.assert_array_equal(run.all_truncated, g["all_truncated"][t])
        if bool(g["fsm"]) and t + 1 < T and not g["reset_before"][t + 1].any():
            np.testing.assert_array_equal(run.get_i32("env.stage")[:, 0], g["stage"][t + 1])
        if t < int(g["n_logs"]):
            np.testing.assert_array_equal(log_matrix(run.log(0)), g[f"log{t}"], err_msg=f"log t={t}")


SHUFFLE_CASES = ["sc_shuffle", "sc_shuffle_fsm"]     # BatchResolver(shuffle_batches=True): generic engine only
HANDLER_CASES = ["sc_fsm_handler"]                    # an FSM stage handler chooses the next stage: generic engine only


def test_shuffle_goldens_are_not_the_identity():
    """the recorded permutations really reorder batches, and the reordering changes the trajectory: replaying
    the golden WITHOUT them (shuffle off) must not reproduce the reference's missed-sales history"""
    g = golden("sc_shuffle")
    n = int(g["shuffle_n"][0, 0])
    assert n > 0 and not np.array_equal(g["shuffle"][0, 0, :n], np.zeros(n))
    g2 = {k: g[k] for k in g.files if k not in ("shuffle", "shuffle_n")}
    with pytest.raises(AssertionError):
        replay_supply_chain(g2, lambda spec: OracleEnv(spec))


@pytest.mark.parametrize("name", SC_CASES + SHUFFLE_CASES + HANDLER_CASES)
def test_oracle_supply_chain_matches_reference(name):
    replay_supply_chain(golden(name), lambda spec: OracleEnv(spec))


def test_oracle_state_dependent_stage_handler_between_the_two_halves_of_a_step():
    """golden `sc_fsm_state_handler`: the REFERENCE running a RESTOCK handler that calls resolve_network() and then branches on
    ShopAgent.stock.  phxo_step_begin (acting + resolution), the handler on the resolved state, phxo_step_end(next_stage):
    stage sequence, stocks, observations and rewards equal the reference's; the same golden replayed with ONE step call per
    step and the recorded stages gives the same outputs (begin + end == step)."""
    g = golden("sc_fsm_state_handler")
    assert (g["next_stage"][:2, 0] == [0, 1]).all() and (g["stage"][:3, 0] == [0, 0, 1]).all()      # RESTOCK twice, then SELL
    replay_supply_chain(g, lambda spec: OracleEnv(spec), state_handler=stock_handler_stage)
    replay_supply_chain(g, lambda spec: OracleEnv(spec))


@pytest.mark.parametrize("name", HANDLER_CASES)
def test_oracle_tabulated_stage_handler_matches_reference(name):
    """the reference's handler decides from the clock: declared state-independent it is tabulated per (stage, clock) at
    spec-compile time (phx_spec.stage_tab) and the oracle takes the transitions itself -- same golden, no next_stage input"""
    replay_supply_chain(golden(name), lambda spec: OracleEnv(spec), tabulated_handlers=True)


def replay_market(g, make_runner, tracking=True):
    L, Fw, d, T = int(g["L"]), int(g["Fw"]), int(g["d"]), int(g["T"])
    stochastic = "conn_rate" in g
    rates = [0.7, 0.35, 1.0, 0.0, 0.5] if stochastic else None            # gen_goldens.py main()
    env = market_env(L, Fw, d, int(g["num_steps"]), 1, tracking=tracking, rates=rates)
    if stochastic:
        np.testing.assert_array_equal(env.spec.conn_rate, g["conn_rate"])
    run = make_runner(env.spec)
    for t in range(T):
        if g["reset_before"][t]:
            # stochastic golden: the connectivity the reference's StochasticNetwork drew at this reset
            obs, valid = run.reset(None, None, g["conn_on"][t][None]) if stochastic else run.reset()
            if stochastic:
                np.testing.assert_array_equal(run.get_u8("net.conn_on")[0], g["conn_on"][t])
            np.testing.assert_array_equal(valid[0], g["reset_obs_valid"][t])
            sel = g["reset_obs_valid"][t].astype(bool)
            np.testing.assert_array_equal(f32_bits(obs[0][sel]), f32_bits(g["reset_obs"][t][sel]))
        run.step(g["actions"][t][None], g["action_valid"][t][None], None)
        assert (run.err == 0).all()
        np.testing.assert_array_equal(run.obs_valid[0], g["obs_valid"][t], err_msg=f"ov t={t}")
        np.testing.assert_array_equal(run.reward_valid[0], g["reward_valid"][t], err_msg=f"rv t={t}")
        np.testing.assert_array_equal(run.done_valid[0], g["done_valid"][t])
        ov = g["obs_valid"][t].astype(bool)
        np.testing.assert_array_equal(f32_bits(run.obs[0][ov]), f32_bits(g["obs"][t][ov]), err_msg=f"obs t={t}")
        rv = g["reward_valid"][t] == 1
        np.testing.assert_array_equal(f64_bits(run.reward[0][rv]), f64_bits(g["reward"][t][rv]))
        np.testing.assert_array_equal(run.get_i32("seller.tx")[0], g["seller_tx"][t])
        np.testing.assert_array_equal(f64_bits(run.get_f64("seller.revenue")[0]), f64_bits(g["seller_revenue"][t]))
        np.testing.assert_array_equal(f64_bits(run.get_f64("seller.price")[0]), f64_bits(g["seller_price"][t]))
        np.testing.assert_array_equal(run.get_i32("buyer.bought")[0], g["buyer_bought"][t])
        np.testing.assert_array_equal(f64_bits(run.get_f64("buyer.paid")[0]), f64_bits(g["buyer_paid"][t]))
        np.testing.assert_array_equal(run.all_truncated[0], g["all_truncated"][t])
        if tracking:
            assert int(run.msg_count[0]) == int(g["n_msgs"][t])
        if tracking and t < 4:
            np.testing.assert_array_equal(log_matrix(run.log(0)), g[f"log{t}"], err_msg=f"log t={t}")


MARKET_CASES = ["stk_small", "stk_full", "stk_stochastic"]


@pytest.mark.parametrize("name", MARKET_CASES)
def test_oracle_market_matches_reference(name):
    replay_market(golden(name), lambda spec: OracleEnv(spec))


def replay_ads(g, make_runner, tracking=True):
    """the digital-ads market (gen_goldens_ads.py): exchange auction, publisher draws, tagged floats."""
    T = int(g["T"])
    env = ads_env_from_golden(g, tracking=tracking)
    spec = env.spec
    np.testing.assert_array_equal(spec.param_f[spec.kind == _abi.KIND_PUBLISHER][0], g["click_table"].ravel())
    run = make_runner(spec)
    assert run.D == 3 and run.n_exo == 2
    stochastic = "conn_rate" in g
    n_s = spec.n_samplers
    for t in range(T):
        if g["reset_before"][t]:
            sv = g["sampler_values"][t][None, :n_s] if n_s else None
            conn = g["conn_on"][t][None] if stochastic else None
            _, valid = run.reset(None, sv, conn)
            np.testing.assert_array_equal(valid[0], g["reset_obs_valid"][t])
        assert int(run.get_i32("env.stage")[0, 0]) == int(g["stage"][t])
        run.step(g["actions"][t][None], g["action_valid"][t][None], g["exo"][t][None])
        assert (run.err == 0).all(), (t, run.err)
        msg = f"t={t}"
        np.testing.assert_array_equal(run.get_i32("adv.user")[0], g["user"][t], err_msg=msg)
        np.testing.assert_array_equal(run.get_i32("adv.left_tag")[0], g["left_tag"][t], err_msg=msg)
        np.testing.assert_array_equal(f64_bits(run.get_f64("adv.left")[0]), f64_bits(g["left"][t]), err_msg=msg)
        np.testing.assert_array_equal(run.get_i32("adv.bid_tag")[0], g["bid_tag"][t], err_msg=msg)
        np.testing.assert_array_equal(f64_bits(run.get_f64("adv.bid")[0]), f64_bits(g["bid"][t]), err_msg=msg)
        np.testing.assert_array_equal(run.get_i32("adv.step_clicks")[0], g["step_clicks"][t], err_msg=msg)
        np.testing.assert_array_equal(run.get_i32("adv.step_wins")[0], g["step_wins"][t], err_msg=msg)
        for f in ("total_clicks", "total_requests", "total_wins"):
            np.testing.assert_array_equal(run.get_i32("adv." + f)[0].reshape(-1, 3), g[f][t], err_msg=f"{f} {msg}")
        np.testing.assert_array_equal(run.obs_valid[0], g["obs_valid"][t], err_msg=msg)
        np.testing.assert_array_equal(run.reward_valid[0], g["reward_valid"][t], err_msg=msg)
        np.testing.assert_array_equal(run.done_valid[0], g["done_valid"][t], err_msg=msg)
        ov = g["obs_valid"][t].astype(bool)
        # the device row is f32: budget is f32 in the reference too, budget_left its f64 rounded to f32
        np.testing.assert_array_equal(f32_bits(run.obs[0][ov]), f32_bits(g["obs"][t][ov].astype(np.float32)), err_msg=msg)
        rv = g["reward_valid"][t] == 1
        np.testing.assert_array_equal(f64_bits(run.reward[0][rv]), f64_bits(g["reward"][t][rv]), err_msg=msg)
        dv = g["done_valid"][t].astype(bool)
        np.testing.assert_array_equal(run.terminated[0][dv], g["terminated"][t][dv], err_msg=msg)
        np.testing.assert_array_equal(run.truncated[0][dv], g["truncated"][t][dv], err_msg=msg)
        np.testing.assert_array_equal(run.all_terminated[0], g["all_terminated"][t])
        np.testing.assert_array_equal(run.all_truncated[0], g["all_truncated"][t])
        if tracking:
            assert int(run.msg_count[0]) == int(g["n_msgs"][t]), msg
        if tracking and t < 6:
            np.testing.assert_array_equal(log_matrix(run.log(0)), g[f"log{t}"], err_msg=f"log {msg}")


ADS_CASES = ["ads_first", "ads_second", "ads_sampled", "ads_stochastic"]


@pytest.mark.parametrize("name", ADS_CASES)
def test_oracle_ads_market_matches_reference(name):
    replay_ads(golden(name), lambda spec: OracleEnv(spec))


def test_state_handler_declared_as_rules_reproduces_the_reference():
    """golden `sc_fsm_state_handler` (the REFERENCE: a RESTOCK handler that resolves the network, then branches on the shops' total
    stock) with the handler in RULE form (phx_spec.stage_rules): the oracle evaluates the rule inside ONE step call per step and
    reproduces the handler's decisions, the stage sequence, stocks, observations and rewards bit for bit."""
    g = golden("sc_fsm_state_handler")
    replay_supply_chain(g, lambda spec: OracleEnv(spec), rule_handlers=True)

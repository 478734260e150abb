"""The bulk exits of a rollout fragment (phantom_amd.rollout) against goldens recorded from the reference's own
``_rollout_task_fn`` loop (tests/golden/gen_goldens_rollout.py; phantom/utils/rllib/rollout.py:300-408): the SampleBatch
columns and the lazily built Rollout / Step / AgentStep containers.  CPU: the fragment comes from the oracle's replay of
the recorded episodes; GPU: from ``PhantomEnv.sample`` (fused device rollouts, one pinned copy)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import phantom_amd as ph
from phantom_amd.rollout import AgentStep, DEFAULT_POLICY_ID, FragmentBatch, Rollout, Step, fragment_from_arrays
from helpers import f32_bits, supply_chain_env
from oracle import OracleEnv

GOLD = os.path.join(HERE, "golden")
STAGES = ["RESTOCK", "SELL"]


def _load(name):
    g = dict(np.load(os.path.join(GOLD, f"rollout_task_{name}.npz")))
    T, B, S, n_exo, fsm = (int(v) for v in g["shape"])
    exo = np.zeros((T, B, n_exo), np.uint8)
    d, k = g["draws"], 0
    for t in range(T):
        for b in range(B):                                  # per step, env instance after env instance (rollout.py:361-363)
            if not fsm or g["stage"][t, b] == 1:            # FSM: the customers act (and draw) in SELL steps only
                exo[t, b] = d[k:k + n_exo]; k += n_exo
    assert k == d.size
    return g, T, B, S, n_exo, bool(fsm), exo


def _env(name, B, T):
    if name == "sc7":
        return supply_chain_env(1, [5], 100, B, norm_customers=5)
    if name == "sc3x2":
        return supply_chain_env(3, [2, 2, 2], 12, B, norm_customers=2)
    return supply_chain_env(2, [3, 3], 10, B, fsm=True, norm_customers=3)


def _check_rollouts(frag, g, ids, T, B, fsm):
    """every field of every Step of every Rollout against the recorded ones"""
    ros = frag.rollouts(rollout_ids=g["rollout_id"].tolist(), repeat_ids=g["repeat_id"].tolist(),
                        rollout_params=[{"j": j} for j in range(B)])
    assert len(ros) == B and all(isinstance(r, Rollout) and len(r.steps) == T for r in ros)
    for b, r in enumerate(ros):
        assert (r.rollout_id, r.repeat_id, r.rollout_params) == (int(g["rollout_id"][b]), int(g["repeat_id"][b]), {"j": b})
        for t in (list(range(T)) if T <= 12 else [0, 1, 2, 50, 98, 99]):
            st = r.steps[t]
            assert isinstance(st, Step) and st.i == int(g["step_i"][t, b]) == t
            assert st.stage == (STAGES[int(g["stage"][t, b])] if fsm else None)
            for s, aid in enumerate(ids):
                assert (aid in st.observations) == bool(g["obs_key"][t, b, s]), (t, b, aid)
                if aid in st.observations:
                    assert (f32_bits(np.asarray(st.observations[aid])) == f32_bits(g["obs"][t, b, s])).all(), (t, b, aid)
                assert (aid in st.actions) == bool(g["action_key"][t, b, s])
                if aid in st.actions:
                    assert st.actions[aid].shape == (1,) and f32_bits(st.actions[aid])[0] == f32_bits(g["actions"][t, b, s:s + 1])[0]
                assert (aid in st.rewards) == (g["reward_key"][t, b, s] != 0), (t, b, aid)
                if aid in st.rewards:
                    if g["reward_key"][t, b, s] == 2:
                        assert st.rewards[aid] is None
                    else:                                   # the trajectory holds the f64 reward rounded once to f32 (1e-6 relative, BASELINE north_star)
                        assert np.float32(st.rewards[aid]) == np.float32(g["rewards"][t, b, s])
                assert (aid in st.terminations) == (aid in st.truncations) == bool(g["done_key"][t, b, s])
                assert (aid in st.infos) == bool(g["info_key"][t, b, s])
                if aid in st.terminations:
                    assert st.terminations[aid] == bool(g["terminations"][t, b, s]) and st.truncations[aid] == bool(g["truncations"][t, b, s])
                if aid in st.infos:
                    assert st.infos[aid] == {}
            assert st.terminations["__all__"] == bool(g["all_terminated"][t, b]) and st.truncations["__all__"] == bool(g["all_truncated"][t, b])
        # the helper methods of the reference's Rollout
        aid = ids[-1]
        assert len(r.observations_for_agent(aid)) == T and len(r.observations_for_agent(aid, drop_nones=True)) == int(g["obs_key"][:, b, -1].sum())
        assert len(r.rewards_for_agent(aid, drop_nones=True)) == int((g["reward_key"][:, b, -1] == 1).sum())
        ags = r.steps_for_agent(aid)
        assert len(ags) == T and isinstance(ags[0], AgentStep) and ags[0].i == 0
        assert ags[-1].done == bool(g["terminations"][-1, b, -1] | g["truncations"][-1, b, -1])      # the agent's OWN flags ("__all__" is the env's)
        if fsm:
            assert len(r.actions_for_agent(aid, stages=["RESTOCK"])) == int((g["stage"][:, b] == 0).sum())
        assert sum(n for _, n in r.count_agent_actions(aid)) == T and r[0].i == 0


def _check_sample_batches(frag, g, ids, T, B, S):
    sb = frag.to_sample_batches()
    assert list(sb) == [DEFAULT_POLICY_ID]
    c = sb[DEFAULT_POLICY_ID]
    n = B * S * T
    assert all(len(v) == n for v in c.values()) and c["obs"].shape == (n, 3) and c["actions"].shape == (n, 1)
    # row (b, s, t) of the batch = step t of env instance b, agent s
    am = lambda a: np.moveaxis(a, 0, 2).reshape((n,) + a.shape[3:])
    assert (f32_bits(c["obs"]) == f32_bits(am(g["obs"]))).all()
    assert (f32_bits(c["actions"][:, 0]) == f32_bits(am(g["actions"]))).all()
    assert (f32_bits(c["rewards"]) == f32_bits(am(g["rewards"]).astype(np.float32))).all()
    # new_obs = what the step returned = the next step's observation; the trajectory planes OR "__all__" into the agent's flags
    nxt = am(g["obs"]).reshape(B, S, T, 3)[:, :, 1:]
    assert (f32_bits(c["new_obs"].reshape(B, S, T, 3)[:, :, :-1]) == f32_bits(nxt)).all()
    assert (c["truncateds"] == (am(g["truncations"]) | np.broadcast_to(g["all_truncated"].T[:, None, :], (B, S, T)).reshape(-1)).astype(bool)).all()
    assert not c["terminateds"].any()
    assert (c["t"].reshape(B, S, T) == np.arange(T)).all() and (c["agent_index"].reshape(B, S, T) == np.arange(S)[None, :, None]).all()
    assert len(np.unique(c["eps_id"])) == B
    by_agent = frag.to_sample_batches(lambda aid: f"p_{aid}")
    assert sorted(by_agent) == sorted(f"p_{a}" for a in ids) and all(len(v["rewards"]) == B * T for v in by_agent.values())
    assert (by_agent[f"p_{ids[-1]}"]["obs"] == c["obs"].reshape(B, S, T, 3)[:, -1].reshape(-1, 3)).all()


@pytest.mark.parametrize("name", ["sc7", "sc3x2", "sc_fsm"])
def test_oracle_replay_through_the_containers_equals_the_reference_rollout_task(name):
    g, T, B, S, n_exo, fsm, exo = _load(name)
    env = _env(name, B, T)
    ids = [env.spec.agent_ids[a] for a in env.spec.strategic_idx]
    o = OracleEnv(env.spec)
    first, first_valid = o.reset()
    acts = g["actions"].copy()
    r = o.rollout(T, actions=acts, exo=exo)
    stage = g["stage"] if fsm else None
    frag = fragment_from_arrays(ids, first, r["obs"], r["actions"], r["rewards"], r["terminated"], r["truncated"], env.num_steps, 0,
                                obs_valid=r["obs_valid"] if fsm else None, reward_valid=r["reward_valid"] if fsm else None,
                                first_obs_valid=first_valid if fsm else None, stage=stage, stage_ids=STAGES if fsm else None)
    assert isinstance(frag, FragmentBatch) and (frag.B, frag.S, frag.T) == (B, S, T)
    _check_rollouts(frag, g, ids, T, B, fsm)
    if not fsm:
        _check_sample_batches(frag, g, ids, T, B, S)
    else:                                                    # rows of absent observations are dropped from the batch
        c = frag.to_sample_batches()[DEFAULT_POLICY_ID]
        assert len(c["rewards"]) == int(g["obs_key"].sum())


def test_fragment_needs_the_reset_observation_where_an_episode_starts_inside_it():
    g, T, B, S, n_exo, fsm, exo = _load("sc3x2")
    env = _env("sc3x2", B, T)
    ids = [env.spec.agent_ids[a] for a in env.spec.strategic_idx]
    o = OracleEnv(env.spec)
    first, _ = o.reset()
    r1 = o.rollout(T, actions=g["actions"].copy(), exo=exo)          # ends at the episode's end: last_obs = the next reset's observation
    r2 = o.rollout(5)
    cat = {k: np.concatenate([r1[k], r2[k]]) for k in ("obs", "actions", "rewards", "terminated", "truncated")}
    with pytest.raises(ValueError):
        fragment_from_arrays(ids, first, cat["obs"], cat["actions"], cat["rewards"], cat["terminated"], cat["truncated"], env.num_steps, 0)
    frag = fragment_from_arrays(ids, first, cat["obs"], cat["actions"], cat["rewards"], cat["terminated"], cat["truncated"], env.num_steps, 0,
                                reset_obs={T: r1["last_obs"]})
    assert frag.episodes() == [(0, T), (T, 5)] and len(frag.rollouts()) == 2 * B
    assert (frag.obs[:, :, T, 0] == 0).all()                 # ShopAgent.reset zeroes the stock; sales / missed sales stay (supply_chain.py:149-150)
    assert (frag.obs[:, :, T, 1:] == frag.new_obs[:, :, T - 1, 1:]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sc7", "sc3x2"])
def test_device_sample_equals_the_reference_rollout_task(name):
    """PhantomEnv.sample: fused device rollouts of the recorded episodes (replayed actions and draws), ONE pinned copy, then the
    same column-by-column and Step-by-Step comparison; a second call continues into the next episode."""
    import torch
    g, T, B, S, n_exo, fsm, exo = _load(name)
    env = _env(name, B, T)
    ids = [env.spec.agent_ids[a] for a in env.spec.strategic_idx]
    env.reset()
    dev = env._device().device
    frag = env.sample(T, torch.from_numpy(g["actions"].copy()).to(dev), torch.from_numpy(exo).to(dev))
    _check_rollouts(frag, g, ids, T, B, False)
    _check_sample_batches(frag, g, ids, T, B, S)
    be = ph.rllib.BatchedBaseEnv(env)
    sb = be.sample(T + 3)                                    # random policy; crosses an episode end: two launches
    c = sb[DEFAULT_POLICY_ID]
    tt = c["t"].reshape(B, S, T + 3)
    assert (tt[:, :, :T] == np.arange(T)).all() and (tt[:, :, T:] == np.arange(3)).all()
    ob = c["obs"].reshape(B, S, T + 3, 3); nb = c["new_obs"].reshape(B, S, T + 3, 3)
    assert (ob[:, :, 0] == frag.new_obs[:, :, -1] * np.array([0, 1, 1], np.float32)).all()   # starts from the reset obs the first call ended with
    assert (ob[:, :, T, 0] == 0).all() and (ob[:, :, T, 1:] == nb[:, :, T - 1, 1:]).all()
    assert c["truncateds"].reshape(B, S, T + 3)[:, :, T - 1].all() and c["truncateds"].sum() == B * S

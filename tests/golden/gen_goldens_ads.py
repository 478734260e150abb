#!/usr/bin/env python
"""Golden vectors of the digital-ads market, produced by RUNNING THE REFERENCE's own example module
(examples/environments/digital_ads_market/digital_ads_market.py, imported from /root/reference) on
the reference's FiniteStateMachineEnv / StochasticNetwork / BatchResolver.  Build-container only;
only DATA is written (inputs: actions, the np.random.choice / np.random.binomial draws the publisher
consumed, sampled budgets, sampled connectivity; outputs: observations, rewards, done flags, agent
attributes with their numpy scalar kind, stage ids, the ordered message log).

    python tests/golden/gen_goldens_ads.py      # rewrites tests/golden/ads_*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import ref_import  # noqa: E402

ph = ref_import.import_phantom()
d = os.path.join(ref_import.REFERENCE_ROOT, "examples/environments/digital_ads_market")
sys.path.insert(0, d)
import digital_ads_market as dam  # noqa: E402  (the reference's example module)

MSG = {"ImpressionRequest": 12, "Bid": 13, "AuctionResult": 14, "Ads": 15, "ImpressionResult": 16}
THEMES = ("sport", "travel", "science", "tech")


class DrawRecorder:
    """captures np.random.choice([1, 2]) (:52) and np.random.binomial(1, p) (:193) in consumption order."""

    def __enter__(self):
        self.users, self.clicks = [], []
        self._c, self._b = np.random.choice, np.random.binomial

        def choice(*a, **k):
            v = self._c(*a, **k)
            self.users.append(int(v))
            return v

        def binomial(*a, **k):
            v = self._b(*a, **k)
            self.clicks.append(int(v))
            return v
        np.random.choice, np.random.binomial = choice, binomial
        return self

    def __exit__(self, *exc):
        np.random.choice, np.random.binomial = self._c, self._b


def tag_of(v):
    if isinstance(v, np.float32):
        return 1
    if isinstance(v, np.float64):
        return 2
    assert isinstance(v, float), type(v)
    return 0


def log_array(msgs, index):
    out = np.zeros((len(msgs), 4), dtype=np.float64)
    for k, m in enumerate(msgs):
        p = m.payload
        name = type(p).__name__
        val = {"ImpressionRequest": lambda: p.user_id, "Bid": lambda: p.bid, "AuctionResult": lambda: p.cost,
               "Ads": lambda: index[p.advertiser_id], "ImpressionResult": lambda: p.clicked}[name]()
        out[k] = (index[m.sender_id], index[m.receiver_id], MSG[name], float(val))
    return out


def run_ads(name, themes, budgets, num_steps, T, seed, strategy="first", rates=None, act_hi=1.0, p_uniform=0.85):
    """``budgets[i]``: a python float, ("uniform", low, high) or ("clipped", low, high, clip_low,
    clip_high) per advertiser; ``rates`` = (adx-pub, adx-adv, pub-adv) connectivity or None (1.0)."""
    np.random.seed(seed)
    supertypes, sampler_cols, samplers = {}, [], []
    for i, b in enumerate(budgets):
        aid = f"ADV_{i + 1}"
        if isinstance(b, tuple):
            sm = (ph.utils.samplers.UniformFloatSampler(b[1], b[2]) if b[0] == "uniform" else
                  ph.utils.samplers.UniformFloatSampler(b[1], b[2], clip_low=b[3], clip_high=b[4]))
            supertypes[aid] = dam.AdvertiserAgent.Supertype(budget=sm)
            sampler_cols.append(len(samplers)); samplers.append((sm, b))
        else:
            supertypes[aid] = dam.AdvertiserAgent.Supertype(budget=b)
            sampler_cols.append(-1)
    counts = {}
    for th in themes:
        counts[th] = counts.get(th, 0) + 1
    assert [t for t in counts for _ in range(counts[t])] == list(themes), "themes must be grouped"
    if rates is None:
        env = dam.DigitalAdsEnv(num_steps=num_steps, num_agents_theme=counts, agent_supertypes=supertypes)
    else:
        # the env of :525-596 with per-connection rates (the shipped constructor uses the default 1.0)
        class RatedEnv(dam.DigitalAdsEnv):
            def __init__(self, **kw):
                orig = ph.StochasticNetwork.add_connections_between

                def rated(net, us, vs, rate=1.0):
                    r = rates[0] if "PUB" in vs and "ADX" in us else rates[1] if "ADX" in us else rates[2]
                    return orig(net, us, vs, r)
                ph.StochasticNetwork.add_connections_between = rated
                try:
                    super().__init__(**kw)
                finally:
                    ph.StochasticNetwork.add_connections_between = orig
        env = RatedEnv(num_steps=num_steps, num_agents_theme=counts, agent_supertypes=supertypes)
    env.agents["ADX"].strategy = strategy
    net = env.network
    net.resolver.enable_tracking = True
    ids = list(net.agent_ids)
    index = {aid: i for i, aid in enumerate(ids)}
    advs = [f"ADV_{i + 1}" for i in range(len(budgets))]
    S = len(advs)
    base = [(u, v) for u, v, _ in net._base_connections] if hasattr(net, "_base_connections") else None
    rng = np.random.RandomState(seed + 1000)
    A = dict(actions=np.zeros((T, S), np.float32), action_valid=np.zeros((T, S), np.uint8),
             exo=np.zeros((T, 2), np.uint8),
             obs=np.zeros((T, S, 3), np.float64), obs_valid=np.zeros((T, S), np.uint8),
             reward=np.zeros((T, S), np.float64), reward_valid=np.zeros((T, S), np.uint8),
             terminated=np.zeros((T, S), np.uint8), truncated=np.zeros((T, S), np.uint8),
             done_valid=np.zeros((T, S), np.uint8), all_truncated=np.zeros(T, np.uint8),
             all_terminated=np.zeros(T, np.uint8), reset_before=np.zeros(T, np.uint8),
             reset_obs_valid=np.zeros((T, S), np.uint8), stage=np.zeros(T, np.int32),
             left=np.zeros((T, S)), left_tag=np.zeros((T, S), np.int32), bid=np.zeros((T, S)),
             bid_tag=np.zeros((T, S), np.int32), step_clicks=np.zeros((T, S), np.int32),
             step_wins=np.zeros((T, S), np.int32), user=np.zeros((T, S), np.int32),
             total_clicks=np.zeros((T, S, 3), np.int32), total_requests=np.zeros((T, S, 3), np.int32),
             total_wins=np.zeros((T, S, 3), np.int32), n_msgs=np.zeros(T, np.int32),
             sampler_values=np.zeros((T, max(len(samplers), 1))), budget=np.zeros((T, S)))
    conn_list = None
    if base is not None:
        A["conn_rate"] = np.asarray([r for _, _, r in net._base_connections])
        A["conn_on"] = np.zeros((T, len(base)), np.uint8)
    logs = {}
    need_reset = True
    stage_idx = {"publisher_step": 0, "advertiser_step": 1}
    for t in range(T):
        if need_reset:
            obs, _ = env.reset()
            A["reset_before"][t] = 1
            for aid in obs:
                A["reset_obs_valid"][t, advs.index(aid)] = 1
            need_reset = False
            if base is not None:
                A["conn_on"][t] = [net.graph.has_edge(u, v) for u, v in base]
            for j, (sm, _) in enumerate(samplers):
                A["sampler_values"][t, j] = sm.value
        A["stage"][t] = stage_idx[env.current_stage]
        acts = {}
        if env.current_stage == "advertiser_step":
            for s, aid in enumerate(advs):
                if aid in env._terminations or aid in env._truncations:
                    continue
                if rng.rand() < 0.1:
                    continue                              # no action -> generate_messages (env.py:332)
                # continuous actions with probability p_uniform, else a coarse grid so that bids tie
                a = np.float32(rng.uniform(0.0, act_hi)) if rng.rand() < p_uniform else np.float32(rng.randint(0, 3) / 2.0)
                acts[aid] = np.array([a], dtype=np.float32)
                A["actions"][t, s], A["action_valid"][t, s] = a, 1
        net.resolver.clear_tracked_messages()
        with DrawRecorder() as rec:
            step = env.step(acts)
        assert len(rec.users) <= 1 and len(rec.clicks) <= 1
        A["exo"][t, 0] = rec.users[0] if rec.users else 0
        A["exo"][t, 1] = rec.clicks[0] if rec.clicks else 0
        A["n_msgs"][t] = len(net.resolver.tracked_messages)
        if t < 6:
            logs[f"log{t}"] = log_array(net.resolver.tracked_messages, index)
        for s, aid in enumerate(advs):
            ag = env.agents[aid]
            if aid in step.observations:
                o = step.observations[aid]
                A["obs"][t, s] = (float(o["type"]["budget"][0]), float(o["budget_left"][0]), float(o["user_id"]))
                assert o["type"]["budget"].dtype == np.float32 and o["budget_left"].dtype == np.float64
                A["obs_valid"][t, s] = 1
            if aid in step.rewards:
                r = step.rewards[aid]
                A["reward_valid"][t, s] = 2 if r is None else 1
                A["reward"][t, s] = 0.0 if r is None else float(r)
            if aid in step.terminations:
                A["done_valid"][t, s] = 1
                A["terminated"][t, s] = step.terminations[aid]
                A["truncated"][t, s] = step.truncations[aid]
            A["left"][t, s], A["left_tag"][t, s] = float(ag.left), tag_of(ag.left)
            A["bid"][t, s], A["bid_tag"][t, s] = float(ag.bid), tag_of(ag.bid)
            A["step_clicks"][t, s], A["step_wins"][t, s] = ag.step_clicks, ag.step_wins
            A["user"][t, s] = int(ag._current_user_id)
            A["budget"][t, s] = float(ag.type.budget)
            for u in range(3):
                A["total_clicks"][t, s, u] = ag.total_clicks.get(u, 0)
                A["total_requests"][t, s, u] = ag.total_requests.get(u, 0)
                A["total_wins"][t, s, u] = ag.total_wins.get(u, 0)
        A["all_terminated"][t] = step.terminations["__all__"]
        A["all_truncated"][t] = step.truncations["__all__"]
        if step.terminations["__all__"] or step.truncations["__all__"]:
            need_reset = True
    click = env.agents["PUB"].user_click_proba
    table = np.asarray([[click[u].get(th, 0.0) for th in THEMES] for u in (1, 2)])
    out = dict(T=np.asarray(T), num_steps=np.asarray(num_steps),
               themes=np.asarray([THEMES.index(t) for t in themes]), second=np.asarray(int(strategy == "second")),
               sampler_cols=np.asarray(sampler_cols), click_table=table,
               sampler_params=np.asarray([[x if x is not None else np.nan for x in
                                           (b[1], b[2]) + (tuple(b[3:5]) if b[0] == "clipped" else (None, None))]
                                          for _, b in samplers], dtype=np.float64).reshape(len(samplers), 4),
               const_budgets=np.asarray([b if not isinstance(b, tuple) else np.nan for b in budgets]),
               **A, **logs)
    if name is None:
        return out                      # live cross-check (tests/test_oracle_vs_live_reference.py)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: S={S} T={T} wins={A['step_wins'].sum()} clicks={A['step_clicks'].sum()} "
          f"terminated={int(A['terminated'].sum())} tags={sorted(set(A['left_tag'].ravel()))} "
          f"none_obs={int(((A['stage'] == 0)[:, None] & (A['obs_valid'] == 0)).sum())} msgs={A['n_msgs'][:6]}")


def main():
    th6 = ["travel", "travel", "tech", "tech", "sport", "sport"]
    # constant python-float budgets: float32 arithmetic after the first win (NEP 50)
    run_ads("ads_first", th6, [1.5, 2.0, 2.5, 1.0, 3.0, 0.75], 20, 70, seed=41, act_hi=1.1)
    run_ads("ads_second", th6, [1.5, 2.0, 2.5, 1.0, 3.0, 0.75], 20, 70, seed=42, strategy="second", act_hi=1.1)
    # the training configuration of :826-866 in small: clipped samplers (np.float64 budgets), plus one
    # unclipped sampler (python float) and a constant
    run_ads("ads_sampled", ["travel", "tech", "tech", "sport", "sport"],
            [("clipped", 0.5, 1.501, 0.5, 1.5), ("clipped", 0.7, 1.701, 0.7, 1.7), ("uniform", 0.5, 2.0), 1.25,
             ("clipped", 1.0, 2.001, 1.0, 2.0)], 12, 60, seed=43, strategy="second")
    # connectivity < 1 on every connection class: dropped requests, None observations, lost ads
    run_ads("ads_stochastic", th6, [1.5, 2.0, 2.5, 1.0, 3.0, 0.75], 10, 80, seed=44, rates=(0.8, 0.7, 0.75))


if __name__ == "__main__":
    main()

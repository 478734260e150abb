#!/usr/bin/env python
"""Build-container only: random differential runs of the CPU oracle against the LIVE reference
(jpmorganchase/Phantom imported from /root/reference through ref_import.py).  Complements the
committed golden vectors: random agent orders, several factories, ragged customers, random FSM
stage tables (acting / rewarded lists), partial action dicts, tutorial-2 typed shops, random
Stackelberg markets on static and stochastic networks.

    python tests/golden/ref_fuzz.py [N] [first]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True
import ref_import  # noqa: E402

rph = ref_import.import_phantom()
rsc = ref_import.import_supply_chain()
import gen_goldens as gg  # noqa: E402  (DrawRecorder, market agents, typed shop class)
import phantom_amd as ph  # noqa: E402
from helpers import f32_bits, f64_bits  # noqa: E402
from oracle import OracleEnv  # noqa: E402


def supply_chain_case(rng, case):
    nF, S = int(rng.randint(1, 4)), int(rng.randint(1, 8))
    ks = [int(rng.randint(0, 7)) for _ in range(S)]
    norm = int(rng.randint(1, 8))
    fsm, typed = rng.rand() < 0.5, rng.rand() < 0.3
    num_steps, T = int(rng.randint(1, 8)), int(rng.randint(4, 30))
    rsc.NUM_CUSTOMERS = norm
    fac = [f"F{i}" for i in range(nF)]
    shop_fac = [fac[rng.randint(nF)] for _ in range(S)]
    names = [("S", i) for i in range(S)] + [("F", i) for i in range(nF)] + [("C", i, j) for i in range(S) for j in range(ks[i])]
    names = [names[i] for i in rng.permutation(len(names))]

    def aid(t):
        return f"SHOP{t[1]}" if t[0] == "S" else (fac[t[1]] if t[0] == "F" else f"C{t[1]}_{t[2]}")
    RShop = gg.make_typed_shop_class() if typed else rsc.ShopAgent
    r_agents, m_agents = [], []
    for t in names:
        if t[0] == "S":
            r_agents.append(RShop(aid(t), factory_id=shop_fac[t[1]]))
            cls = ph.TypedShopAgent if typed else ph.ShopAgent
            m_agents.append(cls(aid(t), factory_id=shop_fac[t[1]], num_customers=norm))
        elif t[0] == "F":
            r_agents.append(rsc.FactoryAgent(aid(t))); m_agents.append(ph.FactoryAgent(aid(t)))
        else:
            r_agents.append(rsc.CustomerAgent(aid(t), shop_id=f"SHOP{t[1]}"))
            m_agents.append(ph.CustomerAgent(aid(t), shop_id=f"SHOP{t[1]}"))
    rnet = rph.Network(r_agents); mnet = ph.Network(m_agents)
    conns = [(f"SHOP{i}", shop_fac[i]) for i in range(S)] + [(f"C{i}_{j}", f"SHOP{i}") for i in range(S) for j in range(ks[i])]
    for k in rng.permutation(len(conns)):
        rnet.add_connection(*conns[k]); mnet.add_connection(*conns[k])
    shops = [f"SHOP{i}" for i in range(S)]
    custs = [f"C{i}_{j}" for i in range(S) for j in range(ks[i])]
    sup_r = sup_m = None
    r_samplers, m_samplers = [], []
    if typed:
        from phantom.utils.samplers import UniformFloatSampler as RUniform
        n_sm = int(rng.randint(0, 3))
        prm = [(float(rng.uniform(0, 0.1)), float(rng.uniform(0.1, 0.2)), None if rng.rand() < 0.5 else 0.04,
                None if rng.rand() < 0.5 else 0.16) for _ in range(n_sm)]
        r_samplers = [RUniform(*q) for q in prm]
        m_samplers = [ph.UniformFloatSampler(*q) for q in prm]
        sup_r, sup_m = {}, {}
        for s in shops:
            r = rng.rand()
            if r < 0.25:
                continue                                     # no supertype passed: Supertype() defaults
            if r < 0.6 and n_sm:
                j = int(rng.randint(n_sm))
                sup_r[s] = RShop.Supertype(excess_stock_weight=r_samplers[j])
                sup_m[s] = ph.TypedShopAgent.Supertype(excess_stock_weight=m_samplers[j])
            else:
                v = float(rng.randint(0, 21)) / 100.0
                sup_r[s] = RShop.Supertype(excess_stock_weight=v)
                sup_m[s] = ph.TypedShopAgent.Supertype(excess_stock_weight=v)
    if fsm:
        n_st = int(rng.randint(2, 5))                          # 2..4 stages in a cycle, random tables
        snames = [chr(ord("A") + i) for i in range(n_st)]
        tabs = []
        for i in range(n_st):
            acting = [c for c in custs if rng.rand() < 0.6] + [s for s in shops if rng.rand() < 0.5]
            acting = [acting[j] for j in rng.permutation(len(acting))]
            rewarded = None if rng.rand() < 0.3 else [s for s in shops if rng.rand() < 0.5]
            tabs.append((acting, rewarded))
        init = snames[int(rng.randint(n_st))]
        renv = rph.FiniteStateMachineEnv(num_steps=num_steps, network=rnet, initial_stage=init, agent_supertypes=sup_r,
                                         stages=[rph.FSMStage(nm, acting_agents=a_, rewarded_agents=r_, next_stages=[snames[(i + 1) % n_st]])
                                                 for i, (nm, (a_, r_)) in enumerate(zip(snames, tabs))])
        menv = ph.FiniteStateMachineEnv(num_steps, mnet, initial_stage=init, agent_supertypes=sup_m,
                                        stages=[ph.FSMStage(nm, acting_agents=a_, rewarded_agents=r_, next_stages=[snames[(i + 1) % n_st]])
                                                for i, (nm, (a_, r_)) in enumerate(zip(snames, tabs))])
        acting_of = {nm: a_ for nm, (a_, r_) in zip(snames, tabs)}
    else:
        renv = rph.PhantomEnv(num_steps=num_steps, network=rnet, agent_supertypes=sup_r)
        menv = ph.PhantomEnv(num_steps, mnet, agent_supertypes=sup_m)
    spec = menv.spec
    assert [str(a) for a in spec.agent_ids] == [str(a) for a in renv.agent_ids]
    o = OracleEnv(spec)
    sidx = {spec.agent_ids[a]: s for s, a in enumerate(spec.strategic_idx)}
    cust_rank = {c: r for r, c in enumerate(a for a in spec.agent_ids if str(a).startswith("C"))}
    Ss, D = spec.n_strategic, spec.obs_dim
    need_reset = True
    for t in range(T):
        if need_reset:
            robs, _ = renv.reset()
            # the values the reference's samplers drew at this reset, in env._samplers order
            vals = np.asarray([[sm.value for sm in renv._samplers]], np.float64) if spec.n_samplers else None
            if spec.n_samplers:
                assert len(renv._samplers) == spec.n_samplers
            oobs, ovalid = o.reset(None, vals)
            assert {k for k in robs} == {spec.strategic_ids[s] for s in range(Ss) if ovalid[0, s]}, (case, t, "reset keys")
            for k, v in robs.items():
                assert np.array_equal(f32_bits(v), f32_bits(oobs[0, sidx[k], :len(v)])), (case, t, "reset obs", k)
            need_reset = False
        acts, act, valid = {}, np.zeros((1, Ss), np.float32), np.zeros((1, Ss), np.uint8)
        for sname, s in sidx.items():
            if rng.rand() < 0.85:
                a = np.float32(rng.choice([rng.uniform(-10, 130), rng.randint(0, 100) + 0.5]))
                acts[sname] = np.array([a], np.float32); act[0, s] = a; valid[0, s] = 1
        with gg.DrawRecorder() as rec:
            st = renv.step(acts)
        # the customers that drew, in consumption order = acting order
        if fsm:
            acting = [c for c in acting_of[renv.previous_stage] if str(c).startswith("C")]
        else:
            acting = [c for c in spec.agent_ids if str(c).startswith("C")]
        exo = np.zeros((1, max(spec.n_exo, 1)), np.uint8)
        assert len(rec.draws) == len(acting), (case, t, "draw count")
        for c, v in zip(acting, rec.draws):
            exo[0, cust_rank[c]] = v
        o.step(act, valid, exo[:, :spec.n_exo] if spec.n_exo else None)
        assert o.err[0] == 0
        for sname, s in sidx.items():
            assert (sname in st.observations) == bool(o.obs_valid[0, s]), (case, t, "obs key", sname)
            if sname in st.observations:
                v = st.observations[sname]
                assert np.array_equal(f32_bits(v), f32_bits(o.obs[0, s, :len(v)])), (case, t, "obs", sname)
            rv = 0 if sname not in st.rewards else (2 if st.rewards[sname] is None else 1)
            assert rv == o.reward_valid[0, s], (case, t, "reward key", sname, rv, o.reward_valid[0, s])
            if rv == 1:
                assert f64_bits(np.float64(st.rewards[sname])) == f64_bits(o.reward[0, s]), (case, t, "reward", sname)
            assert (sname in st.terminations) == bool(o.done_valid[0, s]), (case, t, "done key")
        assert bool(st.truncations["__all__"]) == bool(o.all_truncated[0]) and bool(st.terminations["__all__"]) == bool(o.all_terminated[0])
        for i, s in enumerate(shops):
            ag = renv.agents[s]
            kr = [a for a in spec.agent_ids if str(a).startswith("SHOP")].index(s)
            assert (ag.stock, ag.sales, ag.missed_sales) == (o.get_i32("shop.stock")[0, kr], o.get_i32("shop.sales")[0, kr],
                                                             o.get_i32("shop.missed_sales")[0, kr]), (case, t, "state", s)
        if st.truncations["__all__"] or st.terminations["__all__"]:
            need_reset = True
    return f"sc S={S} ks={ks} nF={nF} fsm={fsm} typed={typed} ns={num_steps}"


def market_case(rng, case):
    L, Fw = int(rng.randint(1, 7)), int(rng.randint(1, 16))
    num_steps, T = int(rng.randint(1, 9)), int(rng.randint(4, 26))
    stochastic = rng.rand() < 0.5
    names = [("S", i) for i in range(L)] + [("B", i) for i in range(Fw)]
    names = [names[i] for i in rng.permutation(len(names))]
    values = [float(rng.randint(1, 9)) / 8.0 for _ in range(Fw)]
    r_agents = [gg.SellerAgent(f"S{i}") if k == "S" else gg.BuyerAgent(f"B{i}", values[i]) for k, i in names]
    m_agents = [ph.SellerAgent(f"S{i}") if k == "S" else ph.BuyerAgent(f"B{i}", values[i]) for k, i in names]
    pairs = [(f"B{f}", f"S{l}") for f in range(Fw) for l in range(L) if rng.rand() < min(1.0, 2.5 / L)]
    pairs = [pairs[i] for i in rng.permutation(len(pairs))]
    seed = int(rng.randint(1 << 30))
    if stochastic:
        rates = [float(rng.choice([0.0, 0.3, 0.7, 1.0])) for _ in pairs]
        np.random.seed(seed)
        rnet = rph.StochasticNetwork(r_agents)
        for (u, v), r in zip(pairs, rates):
            rnet.add_connection(u, v, r)
        np.random.seed(seed)
        mnet = ph.StochasticNetwork(m_agents)
        for (u, v), r in zip(pairs, rates):
            mnet.add_connection(u, v, r)
    else:
        rnet, mnet = rph.Network(r_agents), ph.Network(m_agents)
        for u, v in pairs:
            rnet.add_connection(u, v); mnet.add_connection(u, v)
    if rng.rand() < 0.5:
        leaders = [f"S{i}" for i in rng.permutation(L)]
        followers = [f"B{i}" for i in rng.permutation(Fw)]
    else:                                                  # arbitrary lists: any agent on either side, or on none
        leaders, followers = [], []
        allids = [f"S{i}" for i in range(L)] + [f"B{i}" for i in range(Fw)]
        for aid_ in [allids[i] for i in rng.permutation(len(allids))]:
            if rng.rand() < 0.1:
                continue
            (leaders if rng.rand() < (0.7 if aid_.startswith("S") else 0.2) else followers).append(aid_)
    renv = rph.StackelbergEnv(num_steps, rnet, leaders, followers)
    menv = ph.StackelbergEnv(num_steps, mnet, leaders, followers)
    spec = menv.spec
    o = OracleEnv(spec)
    sidx = {spec.agent_ids[a]: s for s, a in enumerate(spec.strategic_idx)}
    Ss = spec.n_strategic
    need_reset = True
    for t in range(T):
        if need_reset:
            robs, _ = renv.reset()
            conn = None
            if stochastic:
                conn = np.asarray([[rnet.graph.has_edge(u, v) for u, v in pairs]], np.uint8)
            oobs, ovalid = o.reset(None, None, conn)
            assert {k for k in robs} == {spec.strategic_ids[s] for s in range(Ss) if ovalid[0, s]}, (case, t, "reset keys")
            for k, v in robs.items():
                assert np.array_equal(f32_bits(v), f32_bits(oobs[0, sidx[k], :len(v)])), (case, t, "reset obs", k)
            need_reset = False
        odd = (renv.current_step + 1) % 2 == 1
        acts, act, valid = {}, np.zeros((1, Ss), np.float32), np.zeros((1, Ss), np.uint8)
        for aid in (leaders if odd else followers):
            if rng.rand() < 0.9:
                a = np.float32(rng.randint(1, 9) / 8.0) if odd else np.float32(rng.randint(0, 4) > 0)
                acts[aid] = np.array([a], np.float32); act[0, sidx[aid]] = a; valid[0, sidx[aid]] = 1
        st = renv.step(acts)
        o.step(act, valid, None)
        assert o.err[0] == 0
        for sname, s in sidx.items():
            assert (sname in st.observations) == bool(o.obs_valid[0, s]), (case, t, "obs key", sname)
            if sname in st.observations:
                assert np.array_equal(f32_bits(st.observations[sname]), f32_bits(o.obs[0, s, :2])), (case, t, "obs", sname)
            rv = 0 if sname not in st.rewards else (2 if st.rewards[sname] is None else 1)
            assert rv == o.reward_valid[0, s], (case, t, "reward key", sname)
            if rv == 1:
                assert f64_bits(np.float64(st.rewards[sname])) == f64_bits(o.reward[0, s]), (case, t, "reward", sname)
        assert bool(st.truncations["__all__"]) == bool(o.all_truncated[0])
        if st.truncations["__all__"] or st.terminations["__all__"]:
            need_reset = True
    return f"stk L={L} Fw={Fw} edges={len(pairs)} stochastic={stochastic} ns={num_steps}"


def run_case(case):
    rng = np.random.RandomState(case)
    return (market_case if case % 3 == 2 else supply_chain_case)(rng, case)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for case in range(first, first + n):
        try:
            desc = run_case(case)
        except AssertionError as e:
            print("FAIL case", case, e.args, flush=True)
            raise
        if case % 10 == 0:
            print("ok", case, desc, flush=True)
    print("reference fuzz done", n)

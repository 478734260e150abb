"""Random differential cases: the HIP path vs the CPU oracle over random supply-chain and market
topologies (agent order, factories, ragged customers, FSM stage tables, typed shops, stochastic
networks, tracking, masked resets, rollouts).  Used by test_gpu_fuzz.py; `python tests/fuzz_cases.py
N [first]` runs a longer campaign on a GPU box."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import phantom_amd as ph
from oracle import OracleEnv
from device_runner import DeviceRunner
from helpers import f32_bits, f64_bits

def cmp(o, d, tag):
    assert np.array_equal(d.err, o.err), (tag, "err", d.err, o.err)
    ok = o.err == 0                     # after a per-env error that env's outputs are unspecified until reset
    for f in ("obs_valid", "reward_valid", "done_valid", "terminated", "truncated", "all_terminated", "all_truncated"):
        assert np.array_equal(getattr(d, f)[ok], getattr(o, f)[ok]), (tag, f)
    assert np.array_equal(f32_bits(d.obs)[ok], f32_bits(o.obs)[ok]), (tag, "obs")
    assert np.array_equal(f64_bits(d.reward)[ok], f64_bits(o.reward)[ok]), (tag, "reward")

def sc_case(rng, case):
    big = rng.rand() < 0.15
    nF = rng.randint(1, 4); S = rng.randint(1, 60 if big else 12)
    ks = [int(rng.randint(0, 71 if (big and rng.rand() < 0.3) else 9)) for _ in range(S)]
    fsm = rng.rand() < 0.4; typed = rng.rand() < 0.3
    B = int(rng.randint(1, 300 if big else 70)); num_steps = int(rng.randint(1, 30 if big else 9)); T = int(rng.randint(3, 30))
    force_generic = rng.rand() < 0.4; tracking = force_generic and rng.rand() < 0.5
    factories = [ph.FactoryAgent(f"F{i}") for i in range(nF)]
    shop_cls = ph.TypedShopAgent if typed else ph.ShopAgent
    shops = [shop_cls(f"SHOP{i}", factory_id=f"F{rng.randint(nF)}", num_customers=max(ks[i], 1)) for i in range(S)]
    custs = [ph.CustomerAgent(f"C{i}_{j}", shop_id=f"SHOP{i}") for i in range(S) for j in range(ks[i])]
    agents = shops + factories + custs
    order = rng.permutation(len(agents))
    agents = [agents[i] for i in order]                       # arbitrary agent order
    rl = None if rng.rand() < 0.8 else int(rng.randint(0, 4))        # small limits: RuntimeError after the last round
    net = ph.Network(agents, resolver=ph.BatchResolver(enable_tracking=tracking, round_limit=rl),
                     ignore_connection_errors=bool(rng.rand() < 0.15), enforce_msg_payload_checks=bool(rng.rand() < 0.85))
    for s in shops:
        if rng.rand() < 0.97: net.add_connection(s.id, s.factory_id)    # rarely a missing edge: NetworkError
    for c in custs:
        if rng.rand() < 0.99: net.add_connection(c.id, c.shop_id)
    kw = dict(batch_size=B, seed=int(rng.randint(1 << 30)), env_offset=int(rng.randint(1 << 20)), force_generic=force_generic)
    if np.random.RandomState(case + 77_000_001).rand() < 0.5:      # round 4: the four-pairs-per-thread step kernel wherever it applies (plain env, one K)
        kw["variants"] = {"step": "wide"}
    sup = None
    host_fed = False
    if typed:
        sm = ph.UniformFloatSampler(0.0, 0.2)
        sm2 = ph.UniformFloatSampler(0.02, 0.18, 0.05, 0.15)
        sup = {s.id: ph.TypedShopAgent.Supertype(sm if rng.rand() < 0.4 else (sm2 if rng.rand() < 0.3 else float(rng.rand() * 0.2)))
               for s in shops if rng.rand() < 0.8}
        host_fed = rng.rand() < 0.4                      # sampler values passed to phx_reset instead of device-drawn
        kw["exogenous"] = "numpy" if host_fed else "device"
    if fsm:
        n_st = int(rng.randint(2, 5))                          # 2..4 stages in a cycle, random tables
        names = [chr(ord("A") + i) for i in range(n_st)]
        stages = []
        for i, nm in enumerate(names):
            acting = [c.id for c in custs if rng.rand() < 0.6] + [s.id for s in shops if rng.rand() < 0.5]
            acting = [acting[j] for j in rng.permutation(len(acting))]
            rewarded = None if rng.rand() < 0.3 else [s.id for s in shops if rng.rand() < 0.5]
            stages.append(ph.FSMStage(nm, acting_agents=acting, rewarded_agents=rewarded, next_stages=[names[(i + 1) % n_st]]))
        env = ph.FiniteStateMachineEnv(num_steps, net, initial_stage=names[int(rng.randint(n_st))], stages=stages,
                                       agent_supertypes=sup, **kw)
    else:
        env = ph.PhantomEnv(num_steps, net, agent_supertypes=sup, **kw)
    spec = env.spec
    o, d = OracleEnv(spec), DeviceRunner(spec)

    def reset(mask=None):
        vals = rng.uniform(0.0, 0.2, (B, spec.n_samplers)) if (host_fed and spec.n_samplers) else None
        (oo, ov), (do, dv) = o.reset(mask, vals), d.reset(mask, vals)
        m = slice(None) if mask is None else mask.astype(bool)
        assert np.array_equal(dv[m], ov[m]) and np.array_equal(f32_bits(do[m]), f32_bits(oo[m])), (case, "reset")

    reset()
    if rng.rand() < 0.3:                                  # per-env tick surgery: mixed tick phases
        ticks = rng.randint(0, 50, B).astype(np.int32)
        o.set_i32("env.tick", ticks); d.set_i32("env.tick", ticks.reshape(B, 1))
    Ss = spec.n_strategic; nx = spec.n_exo
    for t in range(T):
        act = rng.uniform(-20, 130, (B, Ss)).astype(np.float32)
        valid = (rng.rand(B, Ss) < 0.9).astype(np.uint8) if rng.rand() < 0.5 else None
        exo = rng.randint(0, 5, (B, nx)).astype(np.uint8) if (nx and rng.rand() < 0.5) else None
        o.step(act, valid, exo); d.step(act, valid, exo)
        cmp(o, d, (case, t))
        if tracking and (o.err == 0).all():
            assert np.array_equal(d.msg_count, o.msg_count), (case, t, "msg_count")
            b0 = int(rng.randint(B))
            assert np.array_equal(d.log(b0), o.log(b0)), (case, t, "log")
        if (o.err != 0).any():
            reset((o.err != 0).astype(np.uint8))
            if (o.err != 0).any():                          # static cause (missing edge, round limit): stop here
                return f"S={S} err={int(o.err.max())} rl={rl} flags={spec.flags}"
        done = ((o.all_truncated > 0) | (rng.rand(B) < 0.05)).astype(np.uint8)
        if done.any():
            reset(done)
    if rng.rand() < 0.7 and not (host_fed and spec.n_samplers) and (o.err == 0).all():    # fused kernel or launch loop
        Tr = int(rng.randint(1, 150 if big else 40))
        replay = rng.rand() < 0.4
        acts = rng.uniform(-10, 130, (Tr, B, Ss)).astype(np.float32) if replay else None
        exo = rng.randint(0, 5, (Tr, B, nx)).astype(np.uint8) if (replay and nx) else None
        ro, rd = o.rollout(Tr, acts, exo), d.rollout(Tr, acts, exo)
        for k in ("obs", "actions", "rewards", "last_obs"):
            assert np.array_equal(f32_bits(rd[k]), f32_bits(ro[k])), (case, "rollout", k)
        for k in ("truncated", "terminated") + (("obs_valid", "reward_valid") if fsm else ()):
            assert np.array_equal(rd[k], ro[k]), (case, "rollout", k)
        for f in ("shop.stock", "shop.sales", "shop.missed_sales", "env.step", "env.tick"):
            assert np.array_equal(d.get_i32(f), o.get_i32(f)), (case, "after rollout", f)
    return f"S={S} ks={ks} nF={nF} fsm={fsm} typed={typed} B={B} ns={num_steps} generic={force_generic} fused={d.dev.uses_fused}"

def stk_case(rng, case):
    L = int(rng.randint(1, 10)); Fw = int(rng.randint(1, 40)); B = int(rng.randint(1, 30))
    num_steps = int(rng.randint(1, 11)); T = int(rng.randint(3, 25))
    stochastic = rng.rand() < 0.4; force_generic = rng.rand() < 0.3
    sellers = [ph.SellerAgent(f"S{i}") for i in range(L)]
    buyers = [ph.BuyerAgent(f"B{i}", float(rng.randint(1, 9)) / 8.0) for i in range(Fw)]
    agents = sellers + buyers
    agents = [agents[i] for i in rng.permutation(len(agents))]
    pairs = [(b.id, s.id) for b in buyers for s in sellers if rng.rand() < min(1.0, 3.0 / L)]
    if stochastic:
        net = ph.StochasticNetwork(agents)
        for u, v in pairs: net.add_connection(u, v, float(rng.choice([0.0, 0.3, 0.8, 1.0])))
    else:
        net = ph.Network(agents)
        for u, v in pairs: net.add_connection(u, v)
    if rng.rand() < 0.6:
        leaders = [s.id for s in sellers]; followers = [b.id for b in buyers]
    else:                                                  # arbitrary lists: any agent on either side, or on none
        leaders, followers = [], []
        for ag in [agents[i] for i in rng.permutation(len(agents))]:
            r = rng.rand()
            p_lead = 0.7 if isinstance(ag, ph.SellerAgent) else 0.2
            if r < 0.1: continue
            (leaders if rng.rand() < p_lead else followers).append(ag.id)
    env = ph.StackelbergEnv(num_steps, net, leaders, followers, batch_size=B, seed=int(rng.randint(1 << 30)),
                            env_offset=int(rng.randint(1 << 20)), force_generic=force_generic, exogenous="device")
    spec = env.spec
    o, d = OracleEnv(spec), DeviceRunner(spec)
    o.reset(); d.reset()
    S = spec.n_strategic
    for t in range(T):
        act = np.where(rng.rand(B, S) < 0.5, rng.randint(1, 9, (B, S)) / 8.0, 1.0).astype(np.float32)
        valid = (rng.rand(B, S) < 0.9).astype(np.uint8)
        o.step(act, valid, None); d.step(act, valid, None)
        cmp(o, d, (case, t))
        done = ((o.all_truncated > 0) | (rng.rand(B) < 0.05)).astype(np.uint8)
        if done.any():
            (oo, ov), (do, dv) = o.reset(done), d.reset(done)
            m = done.astype(bool)
            assert np.array_equal(dv[m], ov[m]) and np.array_equal(f32_bits(do[m]), f32_bits(oo[m])), (case, t, "reset")
    if rng.rand() < 0.7:                                  # fused kernel or, for the generic engine, the launch loop
        Tr = int(rng.randint(1, 30))
        ro, rd = o.rollout(Tr, None, None), d.rollout(Tr, None, None)
        for k in ("obs", "actions", "rewards", "last_obs"):
            assert np.array_equal(f32_bits(rd[k]), f32_bits(ro[k])), (case, "rollout", k)
        for k in ("truncated", "obs_valid", "reward_valid"):
            assert np.array_equal(rd[k], ro[k]), (case, "rollout", k)
    for f in ("seller.price", "seller.revenue", "buyer.paid"):
        if spec.n_strategic: assert np.array_equal(f64_bits(d.get_f64(f)), f64_bits(o.get_f64(f))), (case, f)
    return f"L={L} Fw={Fw} B={B} edges={len(pairs)} stochastic={stochastic} generic={force_generic} fused={d.dev.uses_fused}"



def net_case(rng, case):
    """message-passing kinds of the reference's own network tests on random graphs: injected sends,
    Network.resolve alone; routing order, round limit, edge / payload / handler errors, tracking."""
    from phantom_amd.message import Message
    from phantom_amd.spec import compile_spec
    n = int(rng.randint(2, 9))
    ids = [f"N{i}" for i in range(n)]
    agents = []
    for i, aid in enumerate(ids):
        k = rng.randint(5)
        if k == 0: agents.append(ph.HalverAgent(aid))
        elif k == 1: agents.append(ph.CashboxAgent(aid))
        elif k == 2: agents.append(ph.ReqRespAgent(aid))
        elif k == 3: agents.append(ph.ForwarderAgent(aid, target=(ids[rng.randint(n)] if rng.rand() < 0.7 else None)))
        else: agents.append(ph.MockAgent(aid))
    rl = None if rng.rand() < 0.4 else int(rng.randint(0, 7))
    net = ph.Network(agents, ph.BatchResolver(enable_tracking=True, round_limit=rl),
                     ignore_connection_errors=bool(rng.rand() < 0.25), enforce_msg_payload_checks=bool(rng.rand() < 0.8))
    for i in range(n):
        for j in range(i + 1, n):
            if rng.rand() < 0.45: net.add_connection(ids[i], ids[j])
    B = int(rng.randint(1, 4))
    spec = compile_spec(net, num_steps=0, batch_size=B)
    o, d = OracleEnv(spec), DeviceRunner(spec)
    for rep in range(int(rng.randint(1, 4))):
        msgs = []
        for _ in range(int(rng.randint(1, 9))):
            u, v = ids[rng.randint(n)], ids[rng.randint(n)]
            t = rng.randint(4)
            pay = (ph.HalveMessage(int(rng.randint(0, 40))) if t == 0 else ph.CashMessage(float(rng.randint(0, 400))) if t == 1
                   else ph.Request(float(rng.randint(0, 100))) if t == 2 else ph.Response(float(rng.randint(0, 100))))
            msgs.append(Message(u, v, pay))
        o.inject(msgs); d.inject(msgs)
        o.resolve(); d.resolve()
        assert np.array_equal(d.err, o.err), (case, rep, "err", d.err, o.err)
        if (o.err == 0).all():
            assert np.array_equal(d.msg_count, o.msg_count), (case, rep, "msg_count")
            assert np.array_equal(d.log(0), o.log(0)), (case, rep, "log")
            for f in ("cashbox.total_cash",):
                if spec.kind.tolist().count(7): assert np.array_equal(f64_bits(d.get_f64(f)), f64_bits(o.get_f64(f))), (case, rep, f)
            for f in ("reqresp.req_time", "reqresp.res_time"):
                if spec.kind.tolist().count(8): assert np.array_equal(d.get_i32(f), o.get_i32(f)), (case, rep, f)
        else:
            o.reset(); d.reset()
    return f"net n={n} rl={rl} flags={spec.flags} B={B}"


def ads_case(rng, case):
    """digital-ads market on random topologies: several publisher / exchange pairs, advertisers spread
    over the exchanges, random connectivity rates, first / second price, constant and sampled budgets,
    plain env (requests and bids meet in one batch) or the two-stage FSM, exogenous or device draws."""
    shipped = rng.rand() < 0.45        # the example's layout: static schedule -> phx_ads_fused.hip (unless forced generic)
    P = 1 if shipped else int(rng.randint(1, 3)); N = int(rng.randint(1, 150 if (shipped and rng.rand() < 0.2) else 14)); B = int(rng.randint(1, 40))
    num_steps = int(rng.randint(2, 14)); T = int(rng.randint(4, 30))
    themes = [ph.ads_market.THEMES[i] for i in rng.randint(0, 4, N)]
    adx_of = rng.randint(0, P, N)
    pubs = [ph.PublisherAgent(f"PUB{p}", exchange_id=f"ADX{p}", click_draws_per_step=int(rng.randint(1, 3)),
                              user_click_proba={u: {t: float(rng.choice([0.0, 0.25, 0.5, 1.0])) for t in ph.ads_market.THEMES}
                                                for u in (1, 2)}) for p in range(P)]
    advs = [ph.AdvertiserAgent(f"ADV{i}", f"ADX{adx_of[i]}", theme=themes[i]) for i in range(N)]
    adxs = [ph.AdExchangeAgent(f"ADX{p}", publisher_id=f"PUB{p}", advertiser_ids=[a.id for i, a in enumerate(advs) if adx_of[i] == p],
                               strategy=("second" if rng.rand() < 0.5 else "first")) for p in range(P)]
    agents = adxs + pubs + advs
    if not shipped:
        agents = [agents[i] for i in rng.permutation(len(agents))]
    rl = None if rng.rand() < 0.3 else int(rng.randint(2, 7))
    ignore = bool(rng.rand() < 0.7)
    net = ph.StochasticNetwork(agents, ph.BatchResolver(round_limit=rl, enable_tracking=bool(not shipped and rng.rand() < 0.4)),
                               ignore_connection_errors=ignore, enforce_msg_payload_checks=bool(rng.rand() < 0.8))
    shipped_dyn = shipped and ignore and rng.rand() < 0.4      # the fused kernel on a graph that differs per env
    rate = lambda: ((float(rng.choice([1.0, 0.9, 0.6])) if shipped_dyn else 1.0) if shipped else
                    (float(rng.choice([1.0, 1.0, 0.9, 0.6])) if ignore or rng.rand() < 0.5 else 1.0))
    for p in range(P):
        net.add_connection(f"ADX{p}", f"PUB{p}", rate())
        for a in adxs[p].advertiser_ids: net.add_connection(f"ADX{p}", a, rate())
        for a in adxs[p].advertiser_ids:
            if shipped or rng.rand() < 0.9: net.add_connection(f"PUB{p}", a, rate())
    sm = [ph.UniformFloatSampler(0.5, 2.0), ph.UniformFloatSampler(0.4, 1.6, 0.5, 1.5)]
    sup = {a.id: ph.AdvertiserAgent.Supertype(budget=(sm[int(rng.randint(2))] if rng.rand() < 0.5 else float(rng.choice([0.5, 1.0, 1.7, 3.0]))))
           for a in advs}
    kw = dict(batch_size=B, seed=int(rng.randint(1 << 30)), env_offset=int(rng.randint(1 << 20)), exogenous="device",
              agent_supertypes=sup, force_generic=bool(shipped and rng.rand() < 0.25))
    fsm = shipped or rng.rand() < 0.6
    if fsm:
        env = ph.FiniteStateMachineEnv(num_steps, net, initial_stage=("adv" if rng.rand() < 0.15 else "pub"), stages=[
            ph.FSMStage("pub", next_stages=["adv"], acting_agents=[p.id for p in pubs], rewarded_agents=[p.id for p in pubs]),
            ph.FSMStage("adv", next_stages=["pub"], acting_agents=[a.id for a in advs],
                        rewarded_agents=(None if rng.rand() < 0.2 else [a.id for a in advs]))], **kw)
    else:
        env = ph.PhantomEnv(num_steps, net, **kw)
    spec = env.spec
    o, d = OracleEnv(spec), DeviceRunner(spec)
    S, nx = spec.n_strategic, spec.n_exo
    slot = spec.exo_slot()
    user_cols = [int(slot[a]) for a in range(spec.n_agents) if spec.kind[a] == ph._abi.KIND_PUBLISHER]

    def state(tag):
        for f in ("adv.left", "adv.bid"):
            assert np.array_equal(f64_bits(d.get_f64(f)), f64_bits(o.get_f64(f))), (case, tag, f)
        for f in ("adv.left_tag", "adv.bid_tag", "adv.step_clicks", "adv.step_wins", "adv.user", "adv.total_clicks",
                  "adv.total_requests", "adv.total_wins"):
            assert np.array_equal(d.get_i32(f), o.get_i32(f)), (case, tag, f)

    def reset(mask=None):
        (oo, ov), (do, dv) = o.reset(mask), d.reset(mask)
        m = slice(None) if mask is None else mask.astype(bool)
        assert np.array_equal(dv[m], ov[m]) and np.array_equal(f32_bits(do[m]), f32_bits(oo[m])), (case, "reset")
        assert np.array_equal(d.get_u8("net.conn_on"), o.get_u8("net.conn_on")), (case, "conn")

    reset()
    for t in range(T):
        act = np.where(rng.rand(B, S) < 0.85, rng.uniform(0, 1.3, (B, S)), rng.randint(0, 3, (B, S)) / 2.0).astype(np.float32)
        valid = (rng.rand(B, S) < 0.9).astype(np.uint8) if rng.rand() < 0.5 else None
        exo = None
        if rng.rand() < 0.5:
            exo = rng.randint(0, 2, (B, nx)).astype(np.uint8)
            exo[:, user_cols] += 1                                # user ids are 1 or 2
        o.step(act, valid, exo); d.step(act, valid, exo)
        cmp(o, d, (case, t))
        if (o.err == 0).all():
            state(t)
            if spec.trace_cap:
                assert np.array_equal(d.msg_count, o.msg_count), (case, t, "msg_count")
                b0 = int(rng.randint(B))
                assert np.array_equal(d.log(b0), o.log(b0)), (case, t, "log")
        else:
            reset((o.err != 0).astype(np.uint8))
            if (o.err != 0).any():
                return f"ads N={N} P={P} err={int(o.err.max())} rl={rl}"
        done = ((o.all_truncated > 0) | (o.all_terminated > 0) | (rng.rand(B) < 0.05)).astype(np.uint8)
        if done.any():
            reset(done)
    if rng.rand() < 0.6 and (o.err == 0).all():                   # launch-loop rollout, device policy and draws
        Tr = int(rng.randint(1, 40))
        racts = rexo = None
        if rng.rand() < 0.4:                                      # replayed policy and draws
            racts = rng.uniform(0, 1.2, (Tr, B, S)).astype(np.float32)
            if rng.rand() < 0.7:
                rexo = rng.randint(0, 2, (Tr, B, nx)).astype(np.uint8)
                rexo[:, :, user_cols] += 1
        ro, rd = o.rollout(Tr, racts, rexo), d.rollout(Tr, racts, rexo)
        assert np.array_equal(d.err, o.err), (case, "rollout err", d.err, o.err)
        ok = o.err == 0                                           # an env's outputs are unspecified after its first error
        for k in ("obs", "actions", "rewards"):
            assert np.array_equal(f32_bits(rd[k])[:, ok], f32_bits(ro[k])[:, ok]), (case, "rollout", k)
        assert np.array_equal(f32_bits(rd["last_obs"])[ok], f32_bits(ro["last_obs"])[ok]), (case, "rollout last_obs")
        for k in ("truncated", "terminated", "obs_valid", "reward_valid"):
            assert np.array_equal(rd[k][:, ok], ro[k][:, ok]), (case, "rollout", k)
        if ok.all():
            state("rollout")
    return f"ads N={N} P={P} B={B} fsm={fsm} rl={rl} ignore={ignore} flags={spec.flags} fused={d.dev.uses_fused}"


def run_case(case):
    rng = np.random.RandomState(case)
    np.random.seed(case)
    if case % 7 == 6:
        return ads_case(rng, case)
    if case % 5 == 4:
        return net_case(rng, case)
    return (stk_case if case % 3 == 2 else sc_case)(rng, case)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for case in range(first, first + n):
        try:
            desc = run_case(case)
        except AssertionError as e:
            print("FAIL case", case, e.args, flush=True)
            raise
        if case % 20 == 0:
            print("ok", case, desc, flush=True)
    print("fuzz done", n)

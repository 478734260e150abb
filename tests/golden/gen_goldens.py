#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: imports jpmorganchase/Phantom from /root/reference (with inert stand-ins
for the absent gymnasium/ray/tensorboardX/termcolor, see ref_import.py) and executes the
reference's own PhantomEnv / FiniteStateMachineEnv / StackelbergEnv / Network / BatchResolver
on seeded inputs, recording inputs (actions, the np.random.randint draws the agents consumed)
and outputs (agent state, obs as f32, rewards as f64, done flags, stage ids, the ordered
``resolver.tracked_messages`` log).  Only DATA is written; no reference source travels.

    python tests/golden/gen_goldens.py          # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import ref_import  # noqa: E402

ph = ref_import.import_phantom()
sc = ref_import.import_supply_chain()

MSG = {"StockRequest": 1, "StockResponse": 2, "OrderRequest": 3, "OrderResponse": 4,
       "Price": 5, "Order": 6}


class DrawRecorder:
    """wraps np.random.randint to capture the exogenous draws in consumption order."""

    def __enter__(self):
        self.draws = []
        self._orig = np.random.randint

        def rec(*a, **k):
            v = self._orig(*a, **k)
            self.draws.append(int(v))
            return v
        np.random.randint = rec
        return self

    def __exit__(self, *exc):
        np.random.randint = self._orig


class ShuffleRecorder:
    """wraps np.random.shuffle (BatchResolver(shuffle_batches=True), resolvers.py:150-151): shuffles an index
    list with the same draws the reference's call consumes, applies it to the batch, and records the indices
    in call order = receivers in dict order, round after round."""

    def __enter__(self):
        self.perm = []
        self._orig = np.random.shuffle

        def rec(x):
            idx = list(range(len(x)))
            self._orig(idx)
            self.perm.extend(idx)
            x[:] = [x[i] for i in idx]
        np.random.shuffle = rec
        return self

    def __exit__(self, *exc):
        np.random.shuffle = self._orig


def log_array(msgs, index):
    out = np.zeros((len(msgs), 4), dtype=np.float64)
    for k, m in enumerate(msgs):
        name = type(m.payload).__name__
        val = getattr(m.payload, "size", None)
        if val is None:
            val = getattr(m.payload, "price", None)
        if val is None:
            val = getattr(m.payload, "vol", None)
        out[k] = (index[m.sender_id], index[m.receiver_id], MSG[name], val)
    return out


# ------------------------------------------------------------------------------------------
# supply chain
# ------------------------------------------------------------------------------------------
MAX_EXCESS_STOCK_WEIGHT = 0.2                 # docs/user/tutorial2.rst:246


def make_typed_shop_class():
    """Tutorial 2's ShopAgent (docs/user/tutorial2.rst:244-307): a Supertype with one field,
    `excess_stock_weight`, that scales the stock penalty and is appended to the observation.
    User-level subclass of the reference's ShopAgent; the env / samplers executing it are the
    reference's."""
    from dataclasses import dataclass

    class ShopAgent(sc.ShopAgent):     # same class name: the payload whitelists match on it (network.py:311-331)
        @dataclass                     # (the tutorial's frozen=True cannot subclass the non-frozen base)
        class Supertype(ph.Supertype):
            excess_stock_weight: float = 0.1

        def encode_observation(self, ctx):
            max_sales_per_step = sc.NUM_CUSTOMERS * sc.CUSTOMER_MAX_ORDER_SIZE
            return np.array([self.stock / sc.SHOP_MAX_STOCK, self.sales / max_sales_per_step,
                             self.missed_sales / max_sales_per_step,
                             self.type.excess_stock_weight / MAX_EXCESS_STOCK_WEIGHT], dtype=np.float32)

        def compute_reward(self, ctx):
            return self.sales - self.type.excess_stock_weight * self.stock

        def reset(self):
            ph.StrategicAgent.reset(self)      # self.type = supertype.sample()  (agents.py:160-175)
            self.stock = 0

    return ShopAgent


def restock_handler(env):
    """a deterministic FSM stage handler (fsm.py:294-302): resolves the network, then restocks AGAIN instead of
    selling on every third step -- decided from the clock alone"""
    env.resolve_network()
    return "RESTOCK" if env.current_step % 3 == 0 else "SELL"


def build_ref_supply_chain(n_shops, ks, num_steps, norm_customers, tracking=False, fsm=False,
                           typed=None, shuffle=False, handler=False):
    """same ids / agent order / connection order as phantom_amd.supply_chain.build_network,
    built from the reference's own agent classes.  ``typed`` = (samplers, per_shop) with
    samplers = [(low, high, clip_low, clip_high)], per_shop[i] = ("sampler", j) | ("const", v) |
    None (no supertype passed -> Supertype() defaults, agents.py:169-171)."""
    sc.NUM_CUSTOMERS = norm_customers            # read at call time, supply_chain.py:125
    if n_shops == 1:
        shop_ids, cust_ids = ["SHOP"], [[f"CUST{i + 1}" for i in range(ks[0])]]
    else:
        shop_ids = [f"SHOP{i}" for i in range(n_shops)]
        cust_ids = [[f"CUST{i}_{j}" for j in range(ks[i])] for i in range(n_shops)]
    factory_id = "WAREHOUSE"
    ShopCls = make_typed_shop_class() if typed else sc.ShopAgent
    shops = [ShopCls(s, factory_id=factory_id) for s in shop_ids]
    kw = {}
    if typed:
        from phantom.utils.samplers import UniformFloatSampler
        sam = [UniformFloatSampler(lo, hi, clo, chi) for lo, hi, clo, chi in typed[0]]
        kw["agent_supertypes"] = {
            sid: ShopCls.Supertype(excess_stock_weight=(sam[t[1]] if t[0] == "sampler" else t[1]))
            for sid, t in zip(shop_ids, typed[1]) if t is not None}
    customers = [sc.CustomerAgent(c, shop_id=shop_ids[i]) for i in range(n_shops) for c in cust_ids[i]]
    net = ph.Network(shops + [sc.FactoryAgent(factory_id)] + customers,
                     resolver=ph.resolvers.BatchResolver(enable_tracking=tracking, shuffle_batches=shuffle))
    for s in shop_ids:
        net.add_connection(s, factory_id)
    for i, s in enumerate(shop_ids):
        net.add_connections_between([s], cust_ids[i])
    if fsm:
        flat_c = [c for cs in cust_ids for c in cs]
        env = ph.FiniteStateMachineEnv(
            num_steps=num_steps, network=net, initial_stage="RESTOCK",
            stages=[ph.FSMStage("RESTOCK", acting_agents=shop_ids, rewarded_agents=shop_ids,
                                next_stages=["SELL", "RESTOCK"] if handler else ["SELL"],
                                handler=(restock_handler if handler is True else handler) if handler else None),
                    ph.FSMStage("SELL", acting_agents=flat_c, rewarded_agents=[],
                                next_stages=["RESTOCK"])], **kw)
    else:
        env = ph.PhantomEnv(num_steps=num_steps, network=net, **kw)
    if typed:
        env._golden_samplers = sam
    return env, shop_ids, [c for cs in cust_ids for c in cs]


def run_supply_chain(name, n_shops, ks, num_steps, T, seeds, action_fn, norm_customers=None,
                     fsm=False, log_steps=0, use_shipped_env=False, typed=None, shuffle=False, handler=False):
    """B = len(seeds) independent reference envs, each alone on the global numpy stream."""
    B, S = len(seeds), n_shops
    n_exo = sum(ks)
    norm_customers = norm_customers or ks[0]
    A = {k: np.zeros((T, B, S), dt) for k, dt in
         (("actions", np.float32), ("stock", np.int32), ("sales", np.int32), ("missed", np.int32),
          ("reward", np.float64), ("reward_valid", np.uint8), ("obs_valid", np.uint8),
          ("terminated", np.uint8), ("truncated", np.uint8), ("done_valid", np.uint8))}
    D = 4 if typed else 3
    A["obs"] = np.zeros((T, B, S, D), np.float32)
    if typed:
        A["type_w"] = np.zeros((T, B, S), np.float64)                 # agent.type.excess_stock_weight
        A["sampler_values"] = np.zeros((T, B, len(typed[0])), np.float64)   # Sampler.value after reset
    A["exo"] = np.zeros((T, B, n_exo), np.uint8)
    A["exo_valid"] = np.zeros((T, B), np.uint8)
    A["all_terminated"] = np.zeros((T, B), np.uint8)
    A["all_truncated"] = np.zeros((T, B), np.uint8)
    A["reset_before"] = np.zeros((T, B), np.uint8)
    A["reset_obs"] = np.zeros((T, B, S, D), np.float32)
    A["reset_obs_valid"] = np.zeros((T, B, S), np.uint8)
    A["stage"] = np.zeros((T, B), np.int32)
    if handler:                                    # the stage the RESTOCK handler / the table chose in each step
        A["next_stage"] = np.zeros((T, B), np.int32)
    if shuffle:                                    # the np.random.shuffle outcomes of every step, in call order
        A["shuffle"] = np.zeros((T, B, 4 * (n_exo + n_shops)), np.uint16)
        A["shuffle_n"] = np.zeros((T, B), np.int32)
    logs = []
    for b, seed in enumerate(seeds):
        if use_shipped_env:
            sc.NUM_CUSTOMERS = 5
            env = sc.SupplyChainEnv()
            env.network.resolver.enable_tracking = log_steps > 0
            shop_ids, cust_ids = ["SHOP"], [f"CUST{i + 1}" for i in range(5)]
        else:
            np.random.seed(seed)                   # the constructor already samples (env.py:118-119)
            env, shop_ids, cust_ids = build_ref_supply_chain(n_shops, ks, num_steps, norm_customers,
                                                             tracking=log_steps > 0, fsm=fsm, typed=typed, shuffle=shuffle,
                                                             handler=handler)
        index = {aid: i for i, aid in enumerate(env.agent_ids)}
        if not typed:
            np.random.seed(seed)
        need_reset = True
        for t in range(T):
            if need_reset:
                obs, _ = env.reset()
                A["reset_before"][t, b] = 1
                for s, sid in enumerate(shop_ids):
                    if sid in obs:
                        A["reset_obs"][t, b, s] = obs[sid]
                        A["reset_obs_valid"][t, b, s] = 1
                need_reset = False
            if typed:
                A["sampler_values"][t, b] = [sm.value for sm in env._golden_samplers]
                for s, sid in enumerate(shop_ids):
                    A["type_w"][t, b, s] = env.agents[sid].type.excess_stock_weight
            if fsm:
                A["stage"][t, b] = ["RESTOCK", "SELL"].index(env.current_stage)
            acts = {}
            for s, sid in enumerate(shop_ids):
                a = np.float32(action_fn(t, b, s))
                A["actions"][t, b, s] = a
                acts[sid] = np.array([a], dtype=np.float32)
            env.network.resolver.clear_tracked_messages()
            with DrawRecorder() as rec, ShuffleRecorder() as shr:
                step = env.step(acts)
            if handler:
                A["next_stage"][t, b] = ["RESTOCK", "SELL"].index(env.current_stage)
            if shuffle:
                A["shuffle"][t, b, :len(shr.perm)] = shr.perm
                A["shuffle_n"][t, b] = len(shr.perm)
            if rec.draws:
                assert len(rec.draws) == n_exo
                A["exo"][t, b] = rec.draws        # customers draw in agent (= exo rank) order
                A["exo_valid"][t, b] = 1
            if b == 0 and t < log_steps:
                logs.append(log_array(env.network.resolver.tracked_messages, index))
            for s, sid in enumerate(shop_ids):
                ag = env.agents[sid]
                A["stock"][t, b, s], A["sales"][t, b, s], A["missed"][t, b, s] = (
                    ag.stock, ag.sales, ag.missed_sales)
                if sid in step.observations:
                    assert step.observations[sid].dtype == np.float32
                    A["obs"][t, b, s] = step.observations[sid]
                    A["obs_valid"][t, b, s] = 1
                if sid in step.rewards:
                    r = step.rewards[sid]
                    A["reward_valid"][t, b, s] = 2 if r is None else 1
                    A["reward"][t, b, s] = 0.0 if r is None else r
                if sid in step.terminations:
                    A["done_valid"][t, b, s] = 1
                    A["terminated"][t, b, s] = step.terminations[sid]
                    A["truncated"][t, b, s] = step.truncations[sid]
            A["all_terminated"][t, b] = step.terminations["__all__"]
            A["all_truncated"][t, b] = step.truncations["__all__"]
            if step.terminations["__all__"] or step.truncations["__all__"]:
                need_reset = True
    meta = dict(n_shops=n_shops, ks=np.asarray(ks), num_steps=num_steps, T=T,
                seeds=np.asarray(seeds), norm_customers=norm_customers, fsm=int(fsm))
    if typed:
        nan = float("nan")
        meta["sampler_params"] = np.asarray([[lo, hi, nan if clo is None else clo, nan if chi is None else chi]
                                             for lo, hi, clo, chi in typed[0]], np.float64)
        # per shop: source sampler index, -1 constant, -2 default Supertype(); and the constant
        meta["type_src"] = np.asarray([(-2 if t is None else (t[1] if t[0] == "sampler" else -1))
                                       for t in typed[1]], np.int32)
        meta["type_const"] = np.asarray([(0.1 if t is None else (t[1] if t[0] == "const" else 0.1))
                                         for t in typed[1]], np.float64)
        meta["max_weight"] = np.asarray(MAX_EXCESS_STOCK_WEIGHT)
    for k, lg in enumerate(logs):
        A[f"log{k}"] = lg
    A["n_logs"] = np.asarray(len(logs))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **A, **meta)
    print(f"{name}: T={T} B={B} S={S} n_exo={n_exo}")


# ------------------------------------------------------------------------------------------
# Stackelberg market: build-authored agents (user-level code) run by the reference's env
# ------------------------------------------------------------------------------------------
@ph.msg_payload("SellerAgent", "BuyerAgent")
class Price:
    price: float


@ph.msg_payload("BuyerAgent", "SellerAgent")
class Order:
    vol: int


class SellerAgent(ph.StrategicAgent):
    def __init__(self, aid):
        super().__init__(aid)
        self.price, self.revenue, self.tx = 0.0, 0.0, 0

    def reset(self):
        self.price, self.revenue, self.tx = 0.0, 0.0, 0

    def decode_action(self, ctx, action):
        self.price = float(action[0])
        return [(nid, Price(self.price)) for nid in ctx.neighbour_ids]

    def pre_message_resolution(self, ctx):
        if ctx.env_view.current_step % 2 == 0:
            self.revenue, self.tx = 0.0, 0

    @ph.agents.msg_handler(Order)
    def handle_order(self, ctx, message):
        self.revenue += self.price * message.payload.vol
        self.tx += message.payload.vol

    def encode_observation(self, ctx):
        n = len(ctx.neighbour_ids)             # can be 0 on a StochasticNetwork
        return np.array([self.tx / n if n else 0.0, self.price], dtype=np.float32)

    def compute_reward(self, ctx):
        return self.revenue


class BuyerAgent(ph.StrategicAgent):
    def __init__(self, aid, value):
        super().__init__(aid)
        self.value = value
        self.prices, self.bought, self.paid = {}, 0, 0.0

    def reset(self):
        self.prices, self.bought, self.paid = None, 0, 0.0

    def _prices(self, ctx):
        if self.prices is None:
            self.prices = {nid: 1.0 for nid in ctx.neighbour_ids}
        return self.prices

    @ph.agents.msg_handler(Price)
    def handle_price(self, ctx, message):
        self._prices(ctx)[message.sender_id] = message.payload.price

    def decode_action(self, ctx, action):
        prices = self._prices(ctx)
        if action[0] > 0.5 and len(prices) > 0:
            best = None
            for nid in ctx.neighbour_ids:            # first minimum in neighbour order
                if best is None or prices[nid] < prices[best]:
                    best = nid
            self.bought, self.paid = 1, prices[best]
            return [(best, Order(1))]
        self.bought, self.paid = 0, 0.0
        return []

    def encode_observation(self, ctx):
        return np.array([min(self._prices(ctx).values(), default=1.0), self.value], dtype=np.float32)

    def compute_reward(self, ctx):
        return self.value - self.paid if self.bought else 0.0


def market_topology(L, Fw, d):
    """follower f <-> leaders (f*d + j*17) mod L, j = 0..d-1."""
    return [[(f * d + j * 17) % L for j in range(d)] for f in range(Fw)]


def run_market(name, L, Fw, d, num_steps, T, seed, rates=None):
    """``rates``: per-connection connectivity of a ph.StochasticNetwork (network.py:340-453), cycled
    over the base connections; None -> static ph.Network."""
    leaders = [f"S{i}" for i in range(L)]
    followers = [f"B{i}" for i in range(Fw)]
    values = [((f % 7) + 1) / 8.0 for f in range(Fw)]
    agents = [SellerAgent(s) for s in leaders] + [BuyerAgent(b, values[f]) for f, b in enumerate(followers)]
    base = [(followers[f], leaders[l]) for f, nb in enumerate(market_topology(L, Fw, d)) for l in nb]
    if rates is None:
        net = ph.Network(agents, resolver=ph.resolvers.BatchResolver(enable_tracking=True))
        for u, v in base:
            net.add_connection(u, v)
    else:
        np.random.seed(seed)                       # add_connection draws (network.py:389)
        net = ph.StochasticNetwork(agents, resolver=ph.resolvers.BatchResolver(enable_tracking=True))
        conn_rate = [rates[i % len(rates)] for i in range(len(base))]
        for (u, v), r in zip(base, conn_rate):
            net.add_connection(u, v, r)
    env = ph.StackelbergEnv(num_steps, net, leaders, followers)
    ids = leaders + followers
    index = {aid: i for i, aid in enumerate(ids)}
    S = L + Fw
    rng = np.random.RandomState(seed)
    A = dict(actions=np.zeros((T, S), np.float32), action_valid=np.zeros((T, S), np.uint8),
             obs=np.zeros((T, S, 2), np.float32), obs_valid=np.zeros((T, S), np.uint8),
             reward=np.zeros((T, S), np.float64), reward_valid=np.zeros((T, S), np.uint8),
             done_valid=np.zeros((T, S), np.uint8), all_truncated=np.zeros(T, np.uint8),
             all_terminated=np.zeros(T, np.uint8), reset_before=np.zeros(T, np.uint8),
             reset_obs=np.zeros((T, S, 2), np.float32), reset_obs_valid=np.zeros((T, S), np.uint8),
             seller_price=np.zeros((T, L)), seller_revenue=np.zeros((T, L)),
             seller_tx=np.zeros((T, L), np.int32), buyer_bought=np.zeros((T, Fw), np.int32),
             buyer_paid=np.zeros((T, Fw)), n_msgs=np.zeros(T, np.int32))
    if rates is not None:
        A["conn_rate"] = np.asarray(conn_rate)
        A["conn_on"] = np.zeros((T, len(base)), np.uint8)     # base connection is in the graph after this reset
    logs = {}
    need_reset = True
    for t in range(T):
        if need_reset:
            obs, _ = env.reset()
            A["reset_before"][t] = 1
            for aid, o in obs.items():
                A["reset_obs"][t, index[aid]] = o
                A["reset_obs_valid"][t, index[aid]] = 1
            need_reset = False
            if rates is not None:
                A["conn_on"][t] = [net.graph.has_edge(u, v) for u, v in base]
        odd = (env.current_step + 1) % 2 == 1
        acts = {}
        # prices on a coarse grid so that ties between sellers occur (tie -> first neighbour)
        for aid in (leaders if odd else followers):
            a = np.float32(rng.randint(1, 9) / 8.0) if odd else np.float32(rng.randint(0, 4) > 0)
            acts[aid] = np.array([a], dtype=np.float32)
            A["actions"][t, index[aid]] = a
            A["action_valid"][t, index[aid]] = 1
        net.resolver.clear_tracked_messages()
        step = env.step(acts)
        A["n_msgs"][t] = len(net.resolver.tracked_messages)
        if t < 4:
            logs[f"log{t}"] = log_array(net.resolver.tracked_messages, index)
        for aid in ids:
            i = index[aid]
            if aid in step.observations:
                A["obs"][t, i] = step.observations[aid]
                A["obs_valid"][t, i] = 1
            if aid in step.rewards:
                r = step.rewards[aid]
                A["reward_valid"][t, i] = 2 if r is None else 1
                A["reward"][t, i] = 0.0 if r is None else r
            if aid in step.terminations:
                A["done_valid"][t, i] = 1
        for i, aid in enumerate(leaders):
            ag = env.agents[aid]
            A["seller_price"][t, i], A["seller_revenue"][t, i], A["seller_tx"][t, i] = (
                ag.price, ag.revenue, ag.tx)
        for i, aid in enumerate(followers):
            ag = env.agents[aid]
            A["buyer_bought"][t, i], A["buyer_paid"][t, i] = ag.bought, ag.paid
        A["all_terminated"][t] = step.terminations["__all__"]
        A["all_truncated"][t] = step.truncations["__all__"]
        if step.terminations["__all__"] or step.truncations["__all__"]:
            need_reset = True
    np.savez_compressed(os.path.join(HERE, name + ".npz"), L=L, Fw=Fw, d=d, num_steps=num_steps, T=T,
                        values=np.asarray(values), **A, **logs)
    print(f"{name}: L={L} Fw={Fw} d={d} T={T} msgs/step={A['n_msgs'][:4]}")


def main():
    r1 = np.random.RandomState(1)
    u = r1.uniform(0, 100, size=(400, 8, 64)).astype(np.float32)
    special = [0.5, 1.5, 2.5, 3.5, -0.5, -1.5, 99.5, 100.0, 250.0, -7.0, 0.0, 20.0]

    def act_mixed(t, b, s):
        if (t + s) % 11 == 3:
            return special[(t // 11 + b + s) % len(special)]
        return u[t % 400, b % 8, s % 64]

    # config 1: the shipped 7-agent SupplyChainEnv, 3 episodes, seed 0, fixed action 20 (Appendix B)
    run_supply_chain("sc7_fixed20", 1, [5], 100, 300, [0], lambda t, b, s: 20.0, log_steps=3,
                     use_shipped_env=True)
    # same env, random + edge-case actions (half-even rounding, negative, > capacity)
    run_supply_chain("sc7_mixed", 1, [5], 100, 250, [0, 1, 12345], act_mixed, log_steps=2,
                     use_shipped_env=True)
    # config 2 topology: SC64 = 9 shops x 6 customers, 4 seeds, crossing an episode boundary
    run_supply_chain("sc64", 9, [6] * 9, 100, 120, [3, 4, 5, 6], act_mixed, log_steps=2)
    # ragged customers per shop, short episodes
    run_supply_chain("sc_ragged", 3, [1, 4, 7], 10, 35, [7, 8], act_mixed, norm_customers=7,
                     log_steps=2)
    # config 3 topology: SC256 = 51 shops x 4 customers with the 2-stage FSM
    run_supply_chain("sc256_fsm", 51, [4] * 51, 100, 24, [9, 10], act_mixed, fsm=True, log_steps=2)
    # Appendix B FSM case: S=2, K=3, num_steps=6, actions 10+t
    run_supply_chain("sc_fsm_small", 2, [3, 3], 6, 14, [0], lambda t, b, s: 10.0 + (t % 6) + 1,
                     fsm=True, log_steps=2)
    # tutorial 2 (docs/user/tutorial2.rst:244-307): ShopAgent.Supertype.excess_stock_weight fed by
    # shared / clipped UniformFloatSamplers, a constant and the dataclass default; plain and FSM env
    typed = ([(0.0, 0.2, None, None), (0.05, 0.15, 0.07, 0.13)],
             [("sampler", 0), ("sampler", 0), ("sampler", 1), ("const", 0.15), None])
    run_supply_chain("sc_typed", 5, [2, 3, 1, 2, 2], 6, 26, [21, 22, 23], act_mixed, norm_customers=3,
                     typed=typed)
    run_supply_chain("sc_typed_fsm", 5, [2, 3, 1, 2, 2], 6, 26, [31, 32], act_mixed, norm_customers=3,
                     typed=typed, fsm=True)
    # BatchResolver(shuffle_batches=True) (resolvers.py:150-151): every batch permuted by np.random.shuffle; the
    # permutations are recorded in call order and replayed by the oracle / the device (the order of a shop's
    # OrderRequests decides which customers' orders are filled and which are missed)
    run_supply_chain("sc_shuffle", 3, [4, 2, 6], 6, 20, [41, 42, 43], act_mixed, norm_customers=6,
                     log_steps=3, shuffle=True)
    run_supply_chain("sc_shuffle_fsm", 2, [5, 3], 6, 16, [44, 45], act_mixed, norm_customers=5,
                     fsm=True, log_steps=2, shuffle=True)
    # a FiniteStateMachineEnv whose RESTOCK stage has a HANDLER choosing among two next stages (fsm.py:294-307)
    run_supply_chain("sc_fsm_handler", 3, [2, 4, 3], 7, 24, [51, 52], act_mixed, norm_customers=4,
                     fsm=True, log_steps=3, handler=True)
    # config 5: Stackelberg market, small and full size
    run_market("stk_small", 8, 32, 4, 7, 16, seed=11)
    run_market("stk_full", 128, 1024, 8, 100, 6, seed=12)
    # the same market on a StochasticNetwork: connectivity resampled at every reset
    run_market("stk_stochastic", 6, 20, 3, 5, 23, seed=13, rates=[0.7, 0.35, 1.0, 0.0, 0.5])


if __name__ == "__main__":
    main()

"""Shared builders for parity tests: the same topologies the golden generator built from the
reference's classes, expressed through the phantom_amd host API."""
import os

import numpy as np

import phantom_amd as ph
from phantom_amd import _abi

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def supply_chain_env(n_shops, ks, num_steps, batch, fsm=False, norm_customers=None, tracking=False,
                     shuffle=False, **kw):
    resolver = ph.BatchResolver(enable_tracking=tracking, shuffle_batches=shuffle)
    cls = ph.SupplyChainFSMEnv if fsm else ph.SupplyChainEnv
    env = cls(n_shops=n_shops, customers_per_shop=[int(k) for k in ks], num_steps=int(num_steps),
              resolver=resolver, batch_size=batch, **kw)
    if norm_customers is not None:
        for a in env.agents.values():
            if isinstance(a, ph.ShopAgent):
                a.num_customers = int(norm_customers)
    return env


def typed_supertypes(g):
    """agent_supertypes of a typed golden (tests/golden/gen_goldens.py `typed`): shared
    UniformFloatSamplers, constants, and shops left on the Supertype() defaults."""
    nan = lambda v: None if np.isnan(v) else float(v)
    sam = [ph.UniformFloatSampler(float(p[0]), float(p[1]), nan(p[2]), nan(p[3]))
           for p in g["sampler_params"]]
    out = {}
    for i, src in enumerate(g["type_src"].tolist()):
        if src >= 0:
            out[f"SHOP{i}"] = ph.TypedShopAgent.Supertype(excess_stock_weight=sam[src])
        elif src == -1:
            out[f"SHOP{i}"] = ph.TypedShopAgent.Supertype(excess_stock_weight=float(g["type_const"][i]))
    return out


def golden_restock_handler(env):
    """the RESTOCK stage handler of tests/golden/gen_goldens.py (restock_handler), against the phantom_amd surface"""
    env.resolve_network()
    step = np.asarray(env.current_step)
    return np.where(step % 3 == 0, "RESTOCK", "SELL").tolist() if step.ndim else ("RESTOCK" if step % 3 == 0 else "SELL")


def handler_kw(g, tabulated=False):
    """``tabulated``: the same handler DECLARED state-independent (phantom_amd.state_independent): it is evaluated once per
    (stage, clock value) at spec-compile time (phx_spec.stage_tab) and never called at step time"""
    if "next_stage" not in g:
        return {}
    if tabulated:
        return {"restock_handler": ph.state_independent(lambda env: golden_restock_handler(env))}
    return {"restock_handler": golden_restock_handler, "allow_host_handlers": True}


def golden_stock_handler(env, threshold=60):
    """the RESTOCK handler of tests/golden/gen_goldens_fsm_state.py against the phantom_amd surface: agent attributes are [B] arrays"""
    env.resolve_network()
    total = sum(np.asarray(a.stock) for aid, a in env.agents.items() if str(aid).startswith("SHOP"))
    return np.where(total < threshold, "RESTOCK", "SELL").tolist() if np.ndim(total) else ("RESTOCK" if total < threshold else "SELL")


def env_from_golden(g, batch=None, tracking=False, tabulated_handlers=False, **kw):
    if "type_src" in g:
        kw.update(typed=True, agent_supertypes=typed_supertypes(g))
    return supply_chain_env(int(g["n_shops"]), g["ks"], int(g["num_steps"]),
                            batch or len(g["seeds"]), fsm=bool(g["fsm"]),
                            norm_customers=int(g["norm_customers"]), tracking=tracking,
                            shuffle="shuffle" in g, **kw, **handler_kw(g, tabulated_handlers))


def market_topology(L, Fw, d):
    return [[(f * d + j * 17) % L for j in range(d)] for f in range(Fw)]


def market_env(L, Fw, d, num_steps, batch, tracking=False, rates=None, **kw):
    """``rates``: per-connection connectivity of a StochasticNetwork, cycled over the base
    connections (tests/golden/gen_goldens.py run_market); None -> static Network."""
    leaders = [f"S{i}" for i in range(L)]
    followers = [f"B{i}" for i in range(Fw)]
    agents = [ph.SellerAgent(s) for s in leaders] + \
             [ph.BuyerAgent(b, ((f % 7) + 1) / 8.0) for f, b in enumerate(followers)]
    resolver = ph.BatchResolver(enable_tracking=tracking)
    base = [(followers[f], leaders[l]) for f, nb in enumerate(market_topology(L, Fw, d)) for l in nb]
    if rates is None:
        net = ph.Network(agents, resolver=resolver)
        for u, v in base:
            net.add_connection(u, v)
    else:
        net = ph.StochasticNetwork(agents, resolver=resolver)
        for i, (u, v) in enumerate(base):
            net.add_connection(u, v, rates[i % len(rates)])
    return ph.StackelbergEnv(num_steps, net, leaders, followers, batch_size=batch, **kw)


ADS_THEMES = ("sport", "travel", "science", "tech")


def ads_env_from_golden(g, batch=1, tracking=True, **kw):
    """the DigitalAdsEnv of a tests/golden/ads_*.npz case (gen_goldens_ads.py run_ads)."""
    themes = [ADS_THEMES[i] for i in g["themes"]]
    counts = {}
    for th in themes:
        counts[th] = counts.get(th, 0) + 1
    samplers = [ph.UniformFloatSampler(p[0], p[1], None if np.isnan(p[2]) else p[2],
                                       None if np.isnan(p[3]) else p[3]) for p in g["sampler_params"]]
    st = {}
    for i, col in enumerate(g["sampler_cols"]):
        st[f"ADV_{i + 1}"] = ph.AdvertiserAgent.Supertype(
            budget=samplers[col] if col >= 0 else float(g["const_budgets"][i]))
    rates = (1.0, 1.0, 1.0)
    if "conn_rate" in g:
        r, n = g["conn_rate"], len(themes)
        rates = (float(r[0]), float(r[1]), float(r[1 + n]))
    env = ph.DigitalAdsEnv(num_steps=int(g["num_steps"]), num_agents_theme=counts, agent_supertypes=st,
                           strategy="second" if int(g["second"]) else "first", connection_rates=rates,
                           batch_size=batch, **kw)
    env.network.resolver.enable_tracking = tracking
    return env


def f32_bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def f64_bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def log_matrix(recs):
    """(sender, receiver, type, value-as-float) rows from structured log records."""
    out = np.zeros((len(recs), 4), np.float64)
    for k, r in enumerate(recs):
        raw = np.array([r["raw"]], "<i8")
        v = raw.view("<f8")[0] if int(r["type"]) in _abi.FLOAT_PAYLOAD_TYPES else float(raw[0])
        out[k] = (r["sender"], r["receiver"], r["type"], v)
    return out


def philox_np(c0, c1, c2, c3, k0, k1):
    """vectorised Philox4x32-10 (numpy uint64 arithmetic); pinned to oracle.philox in the tests."""
    c0, c1, c2, c3 = (np.asarray(c, np.uint64) & np.uint64(0xffffffff) for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xffffffff)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & MASK, (k1 + np.uint64(0xBB67AE85)) & MASK
    return c0, c1, c2, c3


def find_rng_rejection(seed=1, tick=0, shop=0, limit=4_000_000):
    """first global env index whose order word of (tick, shop), customers 0..5, is rejected
    (low32(u * 5^6) < 14171, probability 3.3e-6): the rare redraw branch of the device RNG."""
    step = 1 << 18
    for lo in range(0, limit, step):
        genv = np.arange(lo, lo + step, dtype=np.uint64)
        w = philox_np(genv, 0, tick >> 2, shop, seed & 0xffffffff, seed >> 32)
        u = w[tick & 3]
        rej = ((u * np.uint64(15625)) & np.uint64(0xffffffff)) < np.uint64(14171)
        if rej.any():
            return int(genv[np.flatnonzero(rej)[0]])
    raise AssertionError("no rejection found")

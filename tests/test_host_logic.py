"""CPU: host-side logic of the drop-in surface (no GPU, no compute calls through the C ABI):
Network/FSM construction + validation parity with the reference's tests, the spec compiler,
the exported symbols of the HIP library, and the pinned definitions (Philox known answers,
exactness of the f32 observation division, numpy stream equivalence)."""
import ctypes
import re
import os

import numpy as np
import pytest

import phantom_amd as ph
from phantom_amd import _abi
from oracle import philox, rng_action, rng_orders

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """include/phantom_amd.h <-> libphantom_amd.so <-> the ctypes table, without touching a GPU."""
    header = open(os.path.join(ROOT, "include", "phantom_amd.h")).read()
    declared = set(re.findall(r"\b(phx_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_abi.EXPORTS)
    lib = _abi.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.phx_abi_version() == _abi.ABI_VERSION
    assert ctypes.sizeof(_abi.PhxMsgRec) == 16 and ctypes.sizeof(_abi.PhxField) == 64


def test_spec_sizes_through_the_abi_without_gpu():
    lib = _abi.load_library()
    env = ph.SupplyChainEnv(n_shops=9, customers_per_shop=6, batch_size=4096)
    cs, keep = env.spec.to_ctypes()
    assert lib.phx_n_strategic(ctypes.byref(cs)) == 9
    assert lib.phx_obs_dim(ctypes.byref(cs)) == 3
    assert lib.phx_n_exo(ctypes.byref(cs)) == 54
    assert lib.phx_state_nbytes(ctypes.byref(cs)) > 0
    cs.abi_version = 99
    assert lib.phx_state_nbytes(ctypes.byref(cs)) == -1
    assert b"abi_version" in lib.phx_last_error()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = ph.SupplyChainEnv()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        env.reset()


def test_supply_chain_spec_matches_reference_layout():
    env = ph.SupplyChainEnv()                        # supply_chain.py:153-175
    assert env.agent_ids == ["SHOP", "WAREHOUSE", "CUST1", "CUST2", "CUST3", "CUST4", "CUST5"]
    assert env.strategic_agent_ids == ["SHOP"] and env.n_agents == 7
    s = env.spec
    assert s.kind.tolist() == [2, 1, 3, 3, 3, 3, 3]
    # nx adjacency order: SHOP -> WAREHOUSE, CUST1..5 ; both directions (network.py:122-123)
    assert s.col[s.row_ptr[0]:s.row_ptr[1]].tolist() == [1, 2, 3, 4, 5, 6]
    assert all(s.col[s.row_ptr[a]:s.row_ptr[a + 1]].tolist() == [0] for a in range(1, 7))
    assert s.param_i[0].tolist()[:2] == [1, 25]      # factory index, NUM_CUSTOMERS * 5
    assert s.param_i[2:, 1].tolist() == [0, 1, 2, 3, 4]
    assert s.round_limit == -1 and s.num_steps == 100


def test_network_construction_parity():
    """tests/network/test_network.py:37-47,128-157 of the reference."""
    net = ph.Network([ph.CashboxAgent("a1"), ph.CashboxAgent("a2")])
    net.add_connection("a1", "a2")
    ph.Network([ph.CashboxAgent("a1"), ph.CashboxAgent("a2")], connections=[("a1", "a2")])
    with pytest.raises(ValueError):
        ph.Network([ph.CashboxAgent("a1"), ph.CashboxAgent("a1")])
    with pytest.raises(ValueError):
        ph.Network([ph.CashboxAgent("a1")], connections=[("a1", "a2")])
    net2 = ph.Network([ph.CashboxAgent("a"), ph.CashboxAgent("b"), ph.CashboxAgent("c")])
    net2.add_connections_with_adjmat(["a", "b"], np.array([[0, 1], [1, 0]]))
    assert net2.has_edge("a", "b") and net2.has_edge("b", "a") and not net2.has_edge("a", "c")
    for bad, msg in ((np.array([[0, 0, 0], [0, 0, 0]]), "Adjacency matrix must be square."),
                     (np.array([[0, 0], [1, 0]]), "Adjacency matrix must be symmetric."),
                     (np.array([[1, 1], [1, 1]]), "Adjacency matrix must be hollow.")):
        with pytest.raises(ValueError) as e:
            net2.add_connections_with_adjmat(["a", "b"], bad)
        assert str(e.value) == msg
    assert net2.get_agents_with_type(ph.Agent) == net2.agents
    assert net2.get_agents_without_type(ph.Agent) == {}


def test_host_send_checks_parity():
    """network.py:246-252,297-331; tests/network/test_payload_checks.py of the reference."""
    net = ph.supply_chain.build_network()
    with pytest.raises(ph.NetworkError):
        net.send("CUST1", "WAREHOUSE", ph.OrderRequest(1))         # no edge
    with pytest.raises(ph.NetworkError):
        net.send("SHOP", "CUST1", ph.OrderRequest(1))              # wrong sender type
    with pytest.raises(ph.NetworkError):
        net.send("CUST1", "SHOP", ph.StockRequest(1))              # wrong sender + receiver
    with pytest.raises(ph.NetworkError):
        net.send("CUST1", "SHOP", True)                            # undecorated payload
    net.send("CUST1", "SHOP", ph.OrderRequest(1))                  # fine: queued for the device


def test_fsm_validation_parity():
    """tests/fsm/test_fsm_validation.py of the reference: same errors at construction."""
    net = ph.Network([ph.MockStrategicAgent("agent")])
    with pytest.raises(ph.FSMValidationError):
        ph.FiniteStateMachineEnv(num_steps=1, network=net, initial_stage="A", stages=[])
    with pytest.raises(ph.FSMValidationError):
        ph.FiniteStateMachineEnv(num_steps=1, network=net, initial_stage="X", stages=[
            ph.FSMStage("A", acting_agents=["agent"], next_stages=["A"])])
    with pytest.raises(ph.FSMValidationError):
        ph.FiniteStateMachineEnv(num_steps=1, network=net, initial_stage="A", stages=[
            ph.FSMStage("A", acting_agents=["agent"], next_stages=["B"])])
    with pytest.raises(ph.FSMValidationError):
        ph.FiniteStateMachineEnv(num_steps=1, network=net, initial_stage="A", stages=[
            ph.FSMStage("A", acting_agents=["agent"], next_stages=[])])
    # a stage with a handler may name several next stages (fsm.py:168-173); the table of allowed transitions
    # (fsm.py:304) goes into the spec, the handler itself runs on the host before each launch
    envh = ph.FiniteStateMachineEnv(num_steps=1, network=net, initial_stage="A", stages=[
        ph.FSMStage("A", acting_agents=["agent"], next_stages=["A", "B"], handler=lambda e: "B"),
        ph.FSMStage("B", acting_agents=["agent"], next_stages=["A"])])
    assert not envh.is_fsm_deterministic() and envh._has_handlers
    assert envh.spec.stage_allowed.tolist() == [[1, 1], [1, 0]] and envh.spec.stage_next.tolist() == [0, 0]
    with pytest.raises(NotImplementedError):
        envh.rollout(4)                           # a fused rollout cannot call Python handlers
    env = ph.FiniteStateMachineEnv(num_steps=3, network=net, initial_stage="A", stages=[
        ph.FSMStage("A", acting_agents=["agent"], next_stages=["B"]),
        ph.FSMStage("B", acting_agents=["agent"], rewarded_agents=[], next_stages=["A"])])
    assert env.is_fsm_deterministic() and env.current_stage == "A"
    s = env.spec
    assert s.stage_next.tolist() == [1, 0] and s.stage_rewarded_all.tolist() == [1, 0]


def test_stackelberg_validation_parity():
    net = ph.Network([ph.MockStrategicAgent("l"), ph.MockStrategicAgent("f")])
    with pytest.raises(AssertionError):
        ph.StackelbergEnv(3, net, ["l"], ["nope"])
    with pytest.raises(AssertionError):
        ph.StackelbergEnv(3, net, ["l"], ["l"])
    env = ph.StackelbergEnv(3, net, ["l"], ["f"])
    assert env.spec.leaders.tolist() == [0] and env.spec.followers.tolist() == [1]


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10 pin the device-RNG primitive."""
    assert [int(x) for x in philox([0, 0, 0, 0], [0, 0])] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(x) for x in philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert [int(x) for x in philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                   [0xa4093822, 0x299f31d0])] == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_device_rng_definition():
    """one Philox block per (shop, customer group g, tick quad): tick t owns word t & 3; m = u * 5^6,
    rejected iff low32(m) < 14171, y = m >> 32 gives six base-5 digits (customers 6g .. 6g+5), the
    word's rank j = (low32(m) - 14171) // 5^6 gives the action j * 100 / 274877."""
    seed, genv, shop = 0x1234567890ABCDEF, 5_000_000_123, 3
    K = 20
    for tick in (76, 77, 78, 79, 80):
        def word(g, attempt=0):
            return int(philox([genv & 0xffffffff, (genv >> 32) | (attempt << 16), tick >> 2, shop | (g << 20)],
                              [seed & 0xffffffff, seed >> 32])[tick & 3])
        got = rng_orders(seed, genv, tick, shop, K)
        for k in range(K):
            g, i = divmod(k, 6)
            m = word(g) * 15625
            assert (m & 0xffffffff) >= 14171
            assert got[k] == ((m >> 32) // 5 ** i) % 5
        m0 = word(0) * 15625
        j = ((m0 & 0xffffffff) - 14171) // 15625
        assert 0 <= j < 274877
        assert rng_action(seed, genv, tick, shop) == np.float32(j) * np.float32(100.0 / 274877.0)
    many = np.concatenate([rng_orders(1, b, t, 0, 6) for b in range(200) for t in range(20)])
    assert many.min() == 0 and many.max() == 4
    assert abs(np.bincount(many, minlength=5) / many.size - 0.2).max() < 0.01
    # the six digits of one word are independent: all 5^2 pairs of (customer 0, customer 5) occur
    pairs = many.reshape(-1, 6)[:, [0, 5]]
    assert len({(int(a), int(b)) for a, b in pairs}) == 25
    acts = np.array([rng_action(1, b, t, 0) for b in range(100) for t in range(20)])
    assert 0.0 <= acts.min() and acts.max() < 100.0 and abs(acts.mean() - 50.0) < 2.0
    # (y, j) is a bijection with the accepted words: exhaustively for the words of three y buckets
    for y in (0, 7777, 15624):
        u0 = -(-(y << 32) // 15625)                                  # first word mapping to y
        us = np.arange(max(u0 - 3, 0), min(u0 + 274877 + 4, 2 ** 32), dtype=np.uint64)
        m = us * np.uint64(15625)
        sel = (m >> np.uint64(32)) == np.uint64(y)
        low = (m & np.uint64(0xffffffff))[sel]
        acc = low >= np.uint64(14171)
        jj = ((low[acc] - np.uint64(14171)) // np.uint64(15625)).astype(np.int64)
        assert acc.sum() == 274877 and np.array_equal(jj, np.arange(274877))
        # the kernels divide by 5^6 with a multiply-high: x // 15625 == (x * 2251799814) >> 45
        x = (low[acc] - np.uint64(14171)).astype(object)
        assert all(int(v) * 2251799814 >> 45 == int(v) // 15625 for v in x[::997])
    xs = np.array([0, 1, 15624, 15625, 15626, 2 ** 32 - 14172, 2 ** 32 - 1], dtype=object)
    assert all(int(v) * 2251799814 >> 45 == int(v) // 15625 for v in xs)


def test_device_rng_rejection_branch():
    """a rejected word is redrawn at the same position with attempt + 1."""
    from helpers import find_rng_rejection, philox_np
    w = philox_np(np.arange(5), 7, 3, 2, 11, 13)                  # the numpy Philox is the oracle's
    for b in range(5):
        assert [int(x[b]) for x in w] == [int(v) for v in philox([b, 7, 3, 2], [11, 13])]
    genv = find_rng_rejection(seed=1)
    u0 = int(philox([genv & 0xffffffff, genv >> 32, 0, 0], [1, 0])[0])
    assert ((u0 * 15625) & 0xffffffff) < 14171
    u1 = int(philox([genv & 0xffffffff, (genv >> 32) | (1 << 16), 0, 0], [1, 0])[0])
    y = (u1 * 15625) >> 32
    assert ((u1 * 15625) & 0xffffffff) >= 14171
    assert list(rng_orders(1, genv, 0, 0, 6)) == [(y // 5 ** i) % 5 for i in range(6)]
    assert rng_action(1, genv, 0, 0) == np.float32((((u1 * 15625) & 0xffffffff) - 14171) // 15625) * np.float32(100.0 / 274877.0)


def test_f32_digit_formulas_are_exact():
    """the kernels take base-5 digits through f32: x // 5 == uint(f32(x) * 0.2f) for x < 2^16 and
    y // 5^i == uint(f32(y) * f32(5^-i)) for y < 5^6 (IEEE multiply, truncating conversion -- the
    same on CPU and GPU), so digit sums are y - 4 * sum_i y // 5^i; Lemire's map is exactly uniform."""
    u = np.arange(65536, dtype=np.uint32)
    q = (u.astype(np.float32) * np.float32(0.2)).astype(np.uint32)
    assert (q == u // 5).all() and ((u - q * 5) == u % 5).all()
    y = np.arange(15625, dtype=np.uint32)
    tot = np.zeros_like(y)
    for i, c in enumerate([0.2, 0.04, 0.008, 0.0016, 0.00032]):
        qi = (y.astype(np.float32) * np.float32(c)).astype(np.uint32)
        assert (qi == y // 5 ** (i + 1)).all()
        tot += qi
    digits = sum((y // 5 ** j) % 5 for j in range(6))
    assert ((y - 4 * tot) == digits).all()
    assert 2 ** 32 % 15625 == 14171
    # every y is hit by exactly floor(2^32 / 5^6) accepted words: check on a coarse residue sample
    us = np.arange(0, 2 ** 32, 9973, dtype=np.uint64)
    m = us * np.uint64(15625)
    acc = (m & np.uint64(0xffffffff)) >= np.uint64(14171)
    assert ((m >> np.uint64(32))[acc] < 15625).all() and abs(acc.mean() - (1 - 14171 / 2 ** 32)) < 1e-4


def test_f32_division_equals_reference_f64_quotient_cast():
    """encode_observation builds python-float quotients and casts to float32
    (supply_chain.py:127-134); the rollout kernel divides in f32.  Identical bit patterns."""
    s = np.arange(-4096, 4097, dtype=np.int64)[:, None]
    n = np.arange(1, 2049, dtype=np.int64)[None, :]
    q64 = (s.astype(np.float64) / n.astype(np.float64)).astype(np.float32)
    q32 = s.astype(np.float32) / n.astype(np.float32)
    assert (q64.view(np.uint32) == q32.view(np.uint32)).all()


def test_division_by_a_kept_reciprocal_is_the_ieee_quotient():
    """phx_dev.h div_by_recip (the policy kernel's observations): q0 = x r, q = fmaf(fmaf(-n, q0, x), r, q0) with r = 1 / n rounded once is
    x / n bit for bit on the whole domain the kernel applies it to (0 <= x < 32768, 1 <= n <= 4096) -- exhaustively, with the host's fmaf."""
    import oracle
    assert oracle.lib().phxo_check_recip_div(4096, 32768) == 0


def test_numpy_stream_vector_draw_equals_scalar_draws():
    """PhantomEnv._draw_exo draws one vector per env; the reference's customers draw scalars."""
    np.random.seed(123)
    a = [np.random.randint(5) for _ in range(3000)]
    np.random.seed(123)
    b = np.concatenate([np.random.randint(5, size=n) for n in (54, 1, 945, 2000)])
    assert a == b.tolist()


def test_auto_chunk_picks_divisors_by_bytes():
    from phantom_amd.distributed import auto_chunk
    assert auto_chunk(100, 811_008) == 100                  # SC64 B=4096: 81 MB fragment -> one chunk
    assert auto_chunk(100, 9_191_424) == 20                 # SC256 B=8192: 9.2 MB/step -> 20-step chunks
    assert auto_chunk(7, 1 << 30) == 1


def test_supertype_and_sampler_semantics():
    """tests/test_supertype.py:12-37 and utils/samplers.py: sample() replaces Samplers by drawn
    values; a managed supertype reads Sampler.value instead of drawing (supertype.py:23-26)."""
    from dataclasses import dataclass

    class MockSampler(ph.Sampler):
        def __init__(self, value):
            self._value = value

        def sample(self):
            self._value += 1
            return self._value

    @dataclass
    class TestSupertype(ph.Supertype):
        a: float
        b: float

    assert TestSupertype(1.0, "string").sample().__dict__ == {"a": 1.0, "b": "string"}
    s2 = TestSupertype(MockSampler(0), "string")
    assert s2.sample().__dict__ == {"a": 1, "b": "string"}
    s2._managed = True
    assert s2.sample().a == 1 and s2.sample().a == 1          # no further draws once managed
    np.random.seed(3)
    u = ph.UniformFloatSampler(0.05, 0.15, 0.07, 0.13)
    np.random.seed(3)
    expect = np.clip(np.random.uniform(0.05, 0.15), 0.07, 0.13)
    np.random.seed(3)
    assert u.sample() == expect and u.value == expect and u <= 0.13 and u >= 0.07
    assert u == u and not (u == ph.UniformFloatSampler())     # identity against other samplers


def test_compile_spec_sampler_tables_and_device_uniform_definition():
    import oracle
    s0, s1 = ph.UniformFloatSampler(0.0, 0.2), ph.UniformFloatSampler(0.05, 0.15, 0.07, 0.13)
    sup = {"SHOP0": ph.TypedShopAgent.Supertype(s0), "SHOP1": ph.TypedShopAgent.Supertype(s0),
           "SHOP2": ph.TypedShopAgent.Supertype(s1), "SHOP3": {"excess_stock_weight": 0.15}}
    env = ph.SupplyChainEnv(n_shops=5, customers_per_shop=2, num_steps=4, batch_size=3, typed=True,
                            agent_supertypes=sup, exogenous="device", seed=9, env_offset=5)
    spec = env.spec
    assert env._samplers == [s0, s1] and env._device_sampling
    assert spec.obs_dim == 4 and spec.n_samplers == 2
    np.testing.assert_array_equal(spec.sampler_kind, [_abi.SAMPLER_UNIFORM] * 2)
    np.testing.assert_array_equal(spec.sampler_param[0, :2], [0.0, 0.2])
    assert np.isnan(spec.sampler_param[0, 2:]).all()
    np.testing.assert_array_equal(spec.sampler_param[1], [0.05, 0.15, 0.07, 0.13])
    np.testing.assert_array_equal(spec.type_src[:5], [0, 0, 1, _abi.TYPE_CONST, _abi.TYPE_CONST])
    np.testing.assert_array_equal(spec.param_f[:5, 0], [0.1, 0.1, 0.1, 0.15, 0.1])
    assert (spec.type_src[5:] == _abi.TYPE_NONE).all()
    # host sampling with the numpy stream when exogenous="numpy" (B = 1 parity mode)
    env1 = ph.SupplyChainEnv(n_shops=2, customers_per_shop=2, typed=True,
                             agent_supertypes={"SHOP0": ph.TypedShopAgent.Supertype(ph.UniformFloatSampler())})
    assert not env1._device_sampling and (env1.spec.sampler_kind == _abi.SAMPLER_HOST).all()
    # the device draw (oracle restatement) against an independent numpy evaluation of its definition
    o = oracle.OracleEnv(spec)
    vals = o.get_f64("env.sampler")
    for b in range(3):
        for j, (lo, hi, clo, chi) in enumerate([(0.0, 0.2, None, None), (0.05, 0.15, 0.07, 0.13)]):
            genv = 5 + b
            w = oracle.philox([genv, 0, 0, 0x80000000 | j], [9, 0])          # episode 0 = constructor
            u = (float(int(w[0]) >> 5) * 67108864.0 + float(int(w[1]) >> 6)) / 9007199254740992.0
            v = lo + (hi - lo) * u
            v = v if clo is None else min(max(v, clo), chi)
            assert vals[b, j] == v
    o.reset()
    assert (o.get_i32("env.episode") == 2).all() and not np.array_equal(o.get_f64("env.sampler"), vals)


def test_stochastic_network_host_semantics_and_spec_tables():
    """tests/network/test_stochastic_network.py:13-58 on the host object, the numpy-stream order of
    the draws, and the base-graph CSR handed to the device."""
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
    with pytest.raises(ValueError):
        ph.StochasticNetwork([ph.Agent("A")]).add_connections_from([("A",)])
    # one np.random.random() per base connection, in order, at add_connection and at each resample
    agents = [ph.SellerAgent("S0"), ph.SellerAgent("S1"), ph.BuyerAgent("B0", 0.5), ph.BuyerAgent("B1", 0.5)]
    rates = [0.5, 0.25, 0.75, 1.0]
    pairs = [("B0", "S0"), ("B0", "S1"), ("B1", "S1"), ("B1", "S0")]
    np.random.seed(4)
    net = ph.StochasticNetwork(agents)
    for (u, v), r in zip(pairs, rates):
        net.add_connection(u, v, r)
    np.random.seed(4)
    expect = [np.random.random() < r for r in rates]
    assert [net.has_edge(u, v) for u, v in pairs] == expect
    state = np.random.get_state()
    expect2 = [np.random.random() < r for r in rates]
    np.random.set_state(state)
    net.resample_connectivity()
    assert [net.has_edge(u, v) for u, v in pairs] == expect2
    env = ph.StackelbergEnv(4, net, ["S0", "S1"], ["B0", "B1"], batch_size=2)
    spec = env.spec
    np.testing.assert_array_equal(spec.conn_rate, rates)
    # base CSR: S0: (B0 c0, B1 c3); S1: (B0 c1, B1 c2); B0: (S0 c0, S1 c1); B1: (S1 c2, S0 c3)
    np.testing.assert_array_equal(spec.row_ptr, [0, 2, 4, 6, 8])
    np.testing.assert_array_equal(spec.col, [2, 3, 2, 3, 0, 1, 1, 0])
    np.testing.assert_array_equal(spec.col_conn, [0, 3, 1, 2, 0, 1, 2, 3])
    # device draw definition (oracle restatement) vs an independent evaluation
    import oracle
    o = oracle.OracleEnv(spec)
    on = o.get_u8("net.conn_on")
    for b in range(2):
        for i, r in enumerate(rates):
            w = oracle.philox([b, 0, 0, 0x40000000 | (i >> 1)], [0, 0])
            h = 2 * (i & 1)
            u = (float(int(w[h]) >> 5) * 67108864.0 + float(int(w[h + 1]) >> 6)) / 9007199254740992.0
            assert on[b, i] == (u < r)


def test_ads_market_spec_compiles_like_the_example():
    """ph.DigitalAdsEnv (digital_ads_market.py:525-596) -> flat spec: agent order, base connections,
    exchange fan-out order, click table, budget kinds (NEP 50 tag), exo columns, queue capacity."""
    from phantom_amd import _abi
    clipped = ph.UniformFloatSampler(5.0, 15.001, clip_low=5.0, clip_high=15.0)
    plain = ph.UniformFloatSampler(1.0, 2.0)
    st = {"ADV_1": ph.AdvertiserAgent.Supertype(budget=clipped), "ADV_2": ph.AdvertiserAgent.Supertype(budget=plain),
          "ADV_3": ph.AdvertiserAgent.Supertype(budget=2.5)}
    env = ph.DigitalAdsEnv(num_steps=20, num_agents_theme={"travel": 1, "tech": 2}, agent_supertypes=st)
    spec = env.spec
    assert spec.agent_ids == ["ADX", "PUB", "ADV_1", "ADV_2", "ADV_3"]
    assert spec.kind.tolist() == [_abi.KIND_ADEXCHANGE, _abi.KIND_PUBLISHER] + [_abi.KIND_ADVERTISER] * 3
    assert spec.round_limit == 5 and spec.flags & _abi.F_IGNORE_CONN_ERRORS and spec.env_type == _abi.ENV_FSM
    assert spec.n_conn == 1 + 3 + 3 and (spec.conn_rate == 1.0).all()
    assert spec.col[spec.row_ptr[0]:spec.row_ptr[1]].tolist() == [1, 2, 3, 4]         # ADX: PUB, then advertiser_ids
    assert spec.param_i[2:, 1].tolist() == [1, 3, 3] and spec.param_i[2:, 2].tolist() == [1, 0, 0]
    assert spec.type_src[2:].tolist() == [0, 1, _abi.TYPE_CONST] and spec.param_f[4, 0] == 2.5
    assert spec.param_f[1].tolist() == [0.0, 1.0, 0.2, 0.5, 1.0, 0.0, 0.7, 0.5]       # :531-534
    assert spec.n_exo == 2 and spec.exo_slot().tolist() == [-1, 0, -1, -1, -1]
    assert spec.queue_cap >= 3 + 2 and spec.n_strategic == 3
    with pytest.raises(ValueError):
        ph.AdExchangeAgent("X", "PUB", strategy="third")
    with pytest.raises(ValueError):          # advertiser_ids must follow the exchange's connection order
        net = ph.StochasticNetwork([ph.AdExchangeAgent("ADX", "PUB", ["A2", "A1"]), ph.PublisherAgent("PUB", "ADX"),
                                    ph.AdvertiserAgent("A1", "ADX", "tech"), ph.AdvertiserAgent("A2", "ADX", "tech")])
        net.add_connections_between(["ADX"], ["PUB", "A1", "A2"])
        ph.compile_spec(net, num_steps=1)
    with pytest.raises(ValueError):
        ph.AdvertiserAgent("A", "ADX", theme="generic").device_params(lambda x: 0)    # not a key of the click table


def test_spec_validation_through_the_abi_without_gpu():
    """derive() rejects what the device path cannot represent: a directed (asymmetric) CSR, duplicate
    edges, more customers per shop than the RNG counter layout holds; message text via phx_last_error."""
    lib = _abi.load_library()
    env = ph.SupplyChainEnv(n_shops=2, customers_per_shop=2, batch_size=4)
    spec = env.spec
    cs, keep = spec.to_ctypes()
    assert lib.phx_state_nbytes(ctypes.byref(cs)) > 0
    col = spec.col.copy()
    # drop the mirror of the first edge: agent 0 (SHOP0) -> factory stays, factory -> SHOP0 is redirected
    a0, f = 0, int(spec.col[spec.row_ptr[0]])
    lo, hi = spec.row_ptr[f], spec.row_ptr[f + 1]
    k = lo + int(np.flatnonzero(spec.col[lo:hi] == a0)[0])
    col[k] = f                                               # a self-loop instead of the mirror edge
    spec2 = type(spec)(**{**spec.__dict__, "col": col})
    cs2, keep2 = spec2.to_ctypes()
    assert lib.phx_state_nbytes(ctypes.byref(cs2)) < 0
    assert b"mirror" in lib.phx_last_error() or b"duplicate" in lib.phx_last_error()
    # an AdvertiserAgent without a budget has no type_src: EINVAL
    net = ph.StochasticNetwork([ph.AdExchangeAgent("ADX", "PUB", ["A1"]), ph.PublisherAgent("PUB", "ADX"),
                                ph.AdvertiserAgent("A1", "ADX", "tech", supertype=ph.AdvertiserAgent.Supertype(budget=1.0))])
    net.add_connections_between(["ADX"], ["PUB", "A1"]); net.add_connection("PUB", "A1")
    s3 = ph.compile_spec(net, num_steps=4)
    cs3, keep3 = s3.to_ctypes()
    assert lib.phx_state_nbytes(ctypes.byref(cs3)) > 0
    s3.type_src[:] = _abi.TYPE_NONE
    cs3, keep3 = s3.to_ctypes()
    assert lib.phx_state_nbytes(ctypes.byref(cs3)) < 0 and b"budget" in lib.phx_last_error()


def test_python_behaviour_on_a_device_kind_is_rejected():
    """A reference-style user class (agents.py:69-79,96-155: @msg_handler methods, overridden
    encode_observation / compute_reward / handle_batch ...) must not silently run as its parent's
    device kind: compile_spec raises and names the method."""
    import phantom_amd as ph

    class MyShop(ph.ShopAgent):
        def compute_reward(self, ctx):
            return 42.0

    class MyAgent(ph.Agent):
        @ph.msg_handler(ph.StockRequest)
        def handle_stock_request(self, ctx, message):
            return []

    class Renamed(ph.ShopAgent):          # no behaviour added: still the SHOP kind
        def helper(self):
            return 1

    class Batchy(ph.StrategicAgent):
        def handle_batch(self, ctx, batch):
            return []

    def net(shop_cls):
        agents = [ph.FactoryAgent("F"), shop_cls("S", "F", 1), ph.CustomerAgent("C", "S")]
        n = ph.Network(agents)
        n.add_connection("F", "S"); n.add_connection("S", "C")
        return n

    ph.compile_spec(net(Renamed), num_steps=3)
    with pytest.raises(ph.UnsupportedAgentBehaviour, match="compute_reward"):
        ph.compile_spec(net(MyShop), num_steps=3)
    with pytest.raises(TypeError, match="msg_handler"):
        ph.compile_spec(ph.Network([MyAgent("A")]), num_steps=3)
    with pytest.raises(NotImplementedError, match="handle_batch"):
        ph.compile_spec(ph.Network([Batchy("B")]), num_steps=3)
    # a grandchild of a user class that overrides is rejected too
    class Deeper(MyShop):
        pass
    with pytest.raises(ph.UnsupportedAgentBehaviour):
        ph.compile_spec(net(Deeper), num_steps=3)


def test_shop_reward_is_a_function_of_10_sales_minus_stock():
    """phx_sc_rollout_sw_kernel tabulates compute_reward (supply_chain.py:144-147) by n = 10 * sales - stock: the f64 expression
    sales - 0.1 * stock rounded once to f32 depends on n only over the reachable range, and equals the f32 quotient n / 10."""
    by_n = {}
    for sales in range(0, 31):
        for stock in range(0, 101):
            r = np.float32(np.float64(sales) - np.float64(0.1) * np.float64(stock))
            n = 10 * sales - stock
            assert by_n.setdefault(n, r.tobytes()) == r.tobytes(), (sales, stock)
            assert r == np.float32(np.float32(n) / np.float32(10.0))
    assert len(by_n) == 401
    # the kernel fills entry n from ONE representative pair: sales = ceil(n / 10) (0 for n < 0), stock = 10 * sales - n
    for n in range(-100, 301):
        sl = (n + 9) // 10 if n >= 0 else 0
        st = 10 * sl - n
        assert 0 <= sl <= 30 and 0 <= st <= 100, n


def test_network_subnet_for_context_for_and_adjacency_matrix():
    """network.py:140-222 on the host surface: add_connections_with_adjmat (validation messages, neighbour order), subnet_for
    (first-order ego network with a reset resolver copy), context_for (neighbour views in graph.neighbors order)."""
    ids = ["A", "B", "C", "D"]
    net = ph.Network([ph.Agent(i) for i in ids])
    m = np.array([[0, 1, 1, 0], [1, 0, 0, 0], [1, 0, 0, 1], [0, 0, 1, 0]])
    net.add_connections_with_adjmat(ids, m)
    assert net.neighbors("A") == ["B", "C"] and net.neighbors("C") == ["A", "D"] and net.neighbors("D") == ["C"]
    for bad, msg in ((np.zeros((3, 3)), "doesn't match"), (np.zeros((4, 3)), "square"), (np.triu(np.ones((4, 4)), 1), "symmetric"),
                     (np.eye(4), "hollow")):
        with pytest.raises(ValueError, match=msg):
            net.add_connections_with_adjmat(ids, bad)
    sub = net.subnet_for("C")
    assert list(sub.agents) == ["A", "C", "D"] and sub.agents["A"] is net.agents["A"]
    assert sub.has_edge("A", "C") and sub.has_edge("D", "C") and not sub.has_edge("A", "B") and "B" not in sub.agents
    assert sub.resolver is not net.resolver and type(sub.resolver) is type(net.resolver)
    ctx = net.context_for("C", ph.EnvView(3, 0.3))
    assert isinstance(ctx, ph.Context) and ctx.agent is net.agents["C"] and ctx.neighbour_ids == ["A", "D"]
    assert ctx["A"] is None and "D" in ctx and "B" not in ctx and ctx.env_view.current_step == 3


def test_kernel_variant_names_resolve_to_the_header_constants():
    """phx_spec.variant_* from the host-side dict (spec.resolve_variants): every name the docs give maps to the constant of
    include/phantom_amd.h, unknown keys and names raise (a typo must not silently select the library's default)."""
    import re
    from phantom_amd import _abi
    from phantom_amd.spec import resolve_variants
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "phantom_amd.h")).read()
    const = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define (PHX_V[RSB]_[A-Z_]+)\s+\(?(-?\d+)\)?", hdr)}
    assert resolve_variants(None) == (0, 0, 0)
    assert resolve_variants({"rollout": "store_waves", "block": 144, "step": "wide"}) == \
        (const["PHX_VR_STORE_WAVES"], 144, const["PHX_VS_WIDE"])
    assert resolve_variants({"rollout": "time_parallel", "block": "whole_envs"})[:2] == (const["PHX_VR_TIME_PARALLEL"], const["PHX_VB_WHOLE_ENVS"])
    assert resolve_variants({"step": "generic"})[2] == const["PHX_VS_GENERIC"] == _abi.VS_GENERIC
    for bad in ({"rolout": "auto"}, {"rollout": "store-waves"}, {"step": "wider"}):
        with pytest.raises(ValueError):
            resolve_variants(bad)


def test_simple_metric_reducers_and_their_argument_errors():
    """metrics.py:141-186: last / mean / sum over the values stacked on axis 0 (the batch axis stays), 'none' keeps every step, values of
    other FSM stages (`not_recorded`) are dropped before reducing, and the constructor refuses unknown actions with the reference's wording."""
    import numpy as np
    import pytest
    from phantom_amd.metrics import SimpleMetric, not_recorded
    vals = [np.array([1.0, 2.0]), np.array([3.0, 6.0]), np.array([5.0, 1.0])]
    assert np.array_equal(SimpleMetric("mean").reduce(vals, "train"), [3.0, 3.0])
    assert np.array_equal(SimpleMetric("sum").reduce(vals, "train"), [9.0, 9.0])
    assert np.array_equal(SimpleMetric("last").reduce(vals, "train"), [5.0, 1.0])
    assert SimpleMetric("last").reduce([], "train") is None
    assert np.array_equal(SimpleMetric("mean", "none").reduce(vals, "evaluate"), np.array(vals))
    staged = SimpleMetric("sum", "last", fsm_stages=["SELL"])
    assert np.array_equal(staged.reduce([vals[0], not_recorded, vals[2]], "train"), [6.0, 3.0])
    assert np.array_equal(staged.reduce([vals[0], not_recorded], "evaluate"), vals[0])
    with pytest.raises(ValueError, match="train_reduce_action field of .* metric must be one of: 'last', 'mean' or 'sum'. Got 'none'"):
        SimpleMetric("none")
    with pytest.raises(ValueError, match="eval_reduce_action field of .* metric class must be one of: 'last', 'mean', 'sum' or 'none'. Got 'max'"):
        SimpleMetric("mean", "max")

"""The reference's own known-answer tests for the path, ported as VALUE fixtures (the expected
values below are the literals asserted in /root/reference/tests; the agents are the device
kinds that mirror those tests' mock agents).  Each function takes a `make_runner(spec)` so the
same KAT pins the CPU oracle (tests/test_oracle_reference_kats.py) and the HIP engine
(tests/test_gpu_parity.py)."""
import numpy as np

import phantom_amd as ph
from phantom_amd import _abi
from phantom_amd.message import Message
from phantom_amd.spec import compile_spec

from helpers import log_matrix


def _net_spec(net, **kw):
    return compile_spec(net, num_steps=kw.pop("num_steps", 0), batch_size=kw.pop("batch", 1), **kw)


def kat_tracking(make_runner):
    """tests/network/test_tracking.py:29-53: exact ordered message log over 3 rounds."""
    net = ph.Network([ph.HalverAgent("A"), ph.HalverAgent("B"), ph.HalverAgent("C")],
                     ph.BatchResolver(enable_tracking=True))
    net.add_connection("A", "B")
    net.add_connection("A", "C")
    run = make_runner(_net_spec(net))
    run.inject([Message("A", "B", ph.HalveMessage(4)), Message("A", "C", ph.HalveMessage(4))])
    run.resolve()
    assert (run.err == 0).all()
    A, B, C, H = 0, 1, 2, _abi.MSG_HALVE
    expected = np.array([[A, B, H, 4], [A, C, H, 4], [B, A, H, 2], [C, A, H, 2],
                         [A, B, H, 1], [A, C, H, 1]], np.float64)
    np.testing.assert_array_equal(log_matrix(run.log(0)), expected)


def _cash_net():
    net = ph.Network([ph.CashboxAgent("mm"), ph.CashboxAgent("inv"), ph.CashboxAgent("inv2")])
    net.add_connection("mm", "inv")
    return net


def kat_call_response(make_runner):
    """tests/network/test_network.py:75-89: halving chain -> total_cash 25/50, 50/100."""
    run = make_runner(_net_spec(_cash_net()))
    run.inject([Message("mm", "inv", ph.CashMessage(100.0))])
    run.resolve()
    np.testing.assert_array_equal(run.get_f64("cashbox.total_cash")[0], [25.0, 50.0, 0.0])
    run = make_runner(_net_spec(_cash_net()))
    run.inject([Message("mm", "inv", ph.CashMessage(100.0))] * 2)
    run.resolve()
    np.testing.assert_array_equal(run.get_f64("cashbox.total_cash")[0], [50.0, 100.0, 0.0])
    run.reset()                                            # test_network.py:110-117
    np.testing.assert_array_equal(run.get_f64("cashbox.total_cash")[0], [0.0, 0.0, 0.0])


def kat_ordering(make_runner):
    """tests/network/test_resolver.py:48-70: receiver processing order within a round."""
    net = ph.Network([ph.ReqRespAgent("A"), ph.ReqRespAgent("B"), ph.ReqRespAgent("C")],
                     ph.BatchResolver())
    net.add_connection("A", "B")
    net.add_connection("A", "C")
    net.add_connection("B", "C")
    run = make_runner(_net_spec(net))
    run.inject([Message("A", "B", ph.Request(100.0)), Message("A", "C", ph.Request(100.0)),
                Message("B", "C", ph.Request(100.0))])
    run.resolve()
    req, res = run.get_i32("reqresp.req_time")[0], run.get_i32("reqresp.res_time")[0]
    assert req[1] <= req[2]              # n["B"].req_time <= n["C"].req_time  (A never gets a Request)
    assert res[2] <= res[0] <= res[1]    # C.res_time <= A.res_time <= B.res_time ... see below
    # exact logical clock: round 0 handles B:req(0), C:req(1), C:req(2); round 1 receivers in
    # first-arrival order A (from B), then ... A gets Response from B and two from C -> A: 3,4 ; B: 5
    assert (req[1], req[2]) == (0, 2)


def kat_round_limit(make_runner):
    """tests/network/test_resolver.py:73-86: round_limit=0 with a queued message raises."""
    net = ph.Network([ph.ReqRespAgent("A"), ph.ReqRespAgent("B")], ph.BatchResolver(round_limit=0))
    net.add_connection("A", "B")
    run = make_runner(_net_spec(net))
    run.inject([Message("A", "B", ph.Request(0.0))])
    run.resolve()
    assert run.err[0] == _abi.ERR_ROUND_LIMIT


def kat_invalid_response_connection(make_runner):
    """tests/network/test_resolver.py:99-113: a response along a missing edge -> NetworkError."""
    net = ph.Network([ph.ForwarderAgent("A"), ph.ForwarderAgent("B", target="C"),
                      ph.ForwarderAgent("C")], ph.BatchResolver())
    net.add_connection("A", "B")
    run = make_runner(_net_spec(net))
    run.inject([Message("A", "B", ph.Request(0.0))])
    run.resolve()
    assert run.err[0] == _abi.ERR_NETWORK


def kat_unknown_message_type(make_runner):
    """tests/test_agent.py:55-68: a payload type without handler -> ValueError."""
    net = ph.Network([ph.HalverAgent("A"), ph.MockAgent("B")], ph.BatchResolver())
    net.add_connection("A", "B")
    run = make_runner(_net_spec(net))
    run.inject([Message("A", "B", ph.HalveMessage(4))])
    run.resolve()
    assert run.err[0] == _abi.ERR_UNKNOWN_MSG


def kat_env_step(make_runner):
    """tests/test_env.py:78-107: key sets, per-agent done removal, __all__ truncation."""
    net = ph.Network([ph.MockStrategicAgent("A", num_steps=1), ph.MockStrategicAgent("B"),
                      ph.MockAgent("C")])
    env = ph.PhantomEnv(num_steps=2, network=net)
    run = make_runner(env.spec)
    obs, valid = run.reset()
    np.testing.assert_array_equal(valid[0], [1, 1])
    run.step(np.zeros((1, 2), np.float32))
    np.testing.assert_array_equal(run.obs_valid[0], [1, 1])
    np.testing.assert_array_equal(run.reward_valid[0], [1, 1])
    np.testing.assert_array_equal(run.done_valid[0], [1, 1])
    np.testing.assert_array_equal(run.terminated[0], [1, 0])
    np.testing.assert_array_equal(run.truncated[0], [1, 0])
    assert (run.all_terminated[0], run.all_truncated[0]) == (0, 0)
    run.step(np.zeros((1, 2), np.float32))
    np.testing.assert_array_equal(run.obs_valid[0], [0, 1])          # keys == ["B"]
    np.testing.assert_array_equal(run.reward_valid[0], [0, 1])
    np.testing.assert_array_equal(run.done_valid[0], [0, 1])
    np.testing.assert_array_equal(run.terminated[0], [0, 0])
    assert (run.all_terminated[0], run.all_truncated[0]) == (0, 1)


def _counts(run):
    return (run.get_i32("mock.compute_reward_count")[0], run.get_i32("mock.encode_obs_count")[0],
            run.get_i32("mock.decode_action_count")[0])


def kat_fsm_odd_even_two_agents(make_runner):
    """tests/fsm/test_odd_even_two_agents.py:42-104."""
    net = ph.Network([ph.MockStrategicAgent("odd_agent"), ph.MockStrategicAgent("even_agent")])
    env = ph.FiniteStateMachineEnv(
        num_steps=3, network=net, initial_stage="ODD",
        stages=[ph.FSMStage("ODD", next_stages=["EVEN"], acting_agents=["odd_agent"],
                            rewarded_agents=["odd_agent"]),
                ph.FSMStage("EVEN", next_stages=["ODD"], acting_agents=["even_agent"],
                            rewarded_agents=["even_agent"])])
    run = make_runner(env.spec)
    obs, valid = run.reset()
    np.testing.assert_array_equal(valid[0], [1, 0])
    assert obs[0, 0, 0] == 0.0
    rew, enc, dec = _counts(run)
    assert (list(rew), list(enc), list(dec)) == ([0, 0], [1, 0], [0, 0])
    run.step(np.array([[1.0, 0.0]], np.float32), np.array([[1, 0]], np.uint8))
    assert run.get_i32("env.stage")[0, 0] == 1
    np.testing.assert_array_equal(run.obs_valid[0], [0, 1])
    np.testing.assert_allclose(run.obs[0, 1, 0], 1.0 / 3.0, rtol=1e-6)
    np.testing.assert_array_equal(run.reward_valid[0], [0, 2])       # {"even_agent": None}
    np.testing.assert_array_equal(run.done_valid[0], [1, 1])
    rew, enc, dec = _counts(run)
    assert (list(rew), list(enc), list(dec)) == ([1, 0], [1, 1], [1, 0])
    run.step(np.array([[0.0, 0.0]], np.float32), np.array([[0, 1]], np.uint8))
    assert run.get_i32("env.stage")[0, 0] == 0
    np.testing.assert_array_equal(run.obs_valid[0], [1, 0])
    np.testing.assert_allclose(run.obs[0, 0, 0], 2.0 / 3.0, rtol=1e-6)
    np.testing.assert_array_equal(run.reward_valid[0], [1, 0])       # {"odd_agent": 0.0}
    assert run.reward[0, 0] == 0.0
    rew, enc, dec = _counts(run)
    assert (list(rew), list(enc), list(dec)) == ([1, 1], [2, 1], [1, 1])


def kat_fsm_odd_even_one_agent(make_runner):
    """tests/fsm/test_odd_even_one_agent.py:40-61 (rewarded_agents=None -> everyone)."""
    net = ph.Network([ph.MockStrategicAgent("agent")])
    env = ph.FiniteStateMachineEnv(
        num_steps=3, network=net, initial_stage="ODD",
        stages=[ph.FSMStage("ODD", acting_agents=["agent"], next_stages=["EVEN"]),
                ph.FSMStage("EVEN", acting_agents=["agent"], next_stages=["ODD"])])
    run = make_runner(env.spec)
    run.reset()
    run.step(np.zeros((1, 1), np.float32))
    assert run.get_i32("env.stage")[0, 0] == 1
    np.testing.assert_allclose(run.obs[0, 0, 0], 1.0 / 3.0, rtol=1e-6)
    assert run.reward_valid[0, 0] == 1 and run.reward[0, 0] == 0.0
    assert (run.terminated[0, 0], run.truncated[0, 0], run.all_terminated[0], run.all_truncated[0]) == (0, 0, 0, 0)
    rew, enc, dec = _counts(run)
    assert (rew[0], enc[0], dec[0]) == (1, 2, 1)


def kat_fsm_one_state(make_runner):
    """tests/fsm/test_one_state.py:113-147 (handler-less single stage, self-loop edge)."""
    net = ph.Network([ph.MockStrategicAgent("agent")])
    net.add_connection("agent", "agent")
    env = ph.FiniteStateMachineEnv(
        num_steps=2, network=net, initial_stage="UNIT",
        stages=[ph.FSMStage("UNIT", acting_agents=["agent"], next_stages=["UNIT"], handler=None)])
    run = make_runner(env.spec)
    obs, valid = run.reset()
    assert valid[0, 0] == 1 and obs[0, 0, 0] == 0.0
    run.step(np.zeros((1, 1), np.float32))
    assert run.obs[0, 0, 0] == 0.5 and run.reward_valid[0, 0] == 1 and run.reward[0, 0] == 0
    assert (run.all_terminated[0], run.all_truncated[0]) == (0, 0)
    rew, enc, dec = _counts(run)
    assert (rew[0], enc[0], dec[0]) == (1, 2, 1)
    run.step(np.zeros((1, 1), np.float32))
    assert run.obs[0, 0, 0] == 1.0 and run.reward_valid[0, 0] == 1
    assert (run.all_terminated[0], run.all_truncated[0]) == (0, 1)
    rew, enc, dec = _counts(run)
    assert (rew[0], enc[0], dec[0]) == (2, 3, 2)


def kat_stackelberg(make_runner):
    """tests/test_stackelberg.py:11-74: 3-step leader/follower trace incl. terminal reward dump."""
    net = ph.Network([ph.MockStrategicAgent("leader"), ph.MockStrategicAgent("follower")])
    env = ph.StackelbergEnv(3, net, ["leader"], ["follower"])
    run = make_runner(env.spec)
    obs, valid = run.reset()
    np.testing.assert_array_equal(valid[0], [1, 0])
    assert obs[0, 0, 0] == 0.0
    run.step(np.zeros((1, 2), np.float32), np.array([[1, 0]], np.uint8))
    np.testing.assert_array_equal(run.obs_valid[0], [0, 1])
    np.testing.assert_allclose(run.obs[0, 1, 0], 1 / 3, rtol=1e-6)
    np.testing.assert_array_equal(run.reward_valid[0], [0, 0])       # rewards == {}
    np.testing.assert_array_equal(run.done_valid[0], [1, 1])
    rew, enc, dec = _counts(run)
    assert (list(rew), list(enc), list(dec)) == ([1, 0], [1, 1], [1, 0])
    run.step(np.zeros((1, 2), np.float32), np.array([[0, 1]], np.uint8))
    np.testing.assert_array_equal(run.obs_valid[0], [1, 0])
    np.testing.assert_allclose(run.obs[0, 0, 0], 2 / 3, rtol=1e-6)
    np.testing.assert_array_equal(run.reward_valid[0], [1, 0])       # {"leader": 0.0}
    rew, enc, dec = _counts(run)
    assert (list(rew), list(enc), list(dec)) == ([1, 1], [2, 1], [1, 1])
    run.step(np.zeros((1, 2), np.float32), np.array([[1, 0]], np.uint8))
    np.testing.assert_array_equal(run.obs_valid[0], [0, 1])
    assert run.obs[0, 1, 0] == 1.0
    np.testing.assert_array_equal(run.reward_valid[0], [1, 1])       # {"leader": 0.0, "follower": 0.0}
    assert (run.all_terminated[0], run.all_truncated[0]) == (0, 1)
    rew, enc, dec = _counts(run)
    assert (list(rew), list(enc), list(dec)) == ([2, 1], [2, 2], [2, 1])


def kat_payload_whitelist(make_runner):
    """tests/network/test_payload_checks.py:14-61 (device side): a decorated payload sent by a
    handler of the wrong agent type is rejected."""
    # a factory answers StockRequest with StockResponse; route it to a non-shop sender
    net = ph.Network([ph.FactoryAgent("F"), ph.HalverAgent("X")], ph.BatchResolver(),
                     enforce_msg_payload_checks=True)
    net.add_connection("F", "X")
    spec = _net_spec(net)
    run = make_runner(spec)
    run.inject([Message("X", "F", ph.StockRequest(3))])    # X is not a ShopAgent
    run.resolve()
    assert run.err[0] == _abi.ERR_PAYLOAD


def kat_ignore_connection_errors(make_runner):
    """Network(ignore_connection_errors=True): a send along a missing edge is queued (and
    tracked) but the receiver's batch filter drops it (network.py:246-249, resolvers.py:146-148)."""
    net = ph.Network([ph.ForwarderAgent("A"), ph.ForwarderAgent("B", target="C"), ph.HalverAgent("C")],
                     ph.BatchResolver(enable_tracking=True), ignore_connection_errors=True,
                     enforce_msg_payload_checks=False)
    net.add_connection("A", "B")
    run = make_runner(_net_spec(net))
    run.inject([Message("A", "B", ph.Request(0.0))])
    run.resolve()
    assert run.err[0] == 0
    log = log_matrix(run.log(0))
    # A->B Request, then B's forward to C along the missing edge: pushed, never handled
    assert log[:, :3].tolist() == [[0, 1, _abi.MSG_REQUEST], [1, 2, _abi.MSG_PING]]


def kat_message_cycle(make_runner):
    """build-specific: BatchResolver(round_limit=None) on a message cycle never returns in the
    reference (resolvers.py:129-131, itertools.count()); this build stops after
    PHX_MAX_ROUNDS rounds and reports ERR_ROUND_LIMIT, on the oracle and on the device alike."""
    net = ph.Network([ph.ForwarderAgent("A", target="B"), ph.ForwarderAgent("B", target="A")],
                     ph.BatchResolver(enable_tracking=True), enforce_msg_payload_checks=False)
    net.add_connection("A", "B")
    run = make_runner(_net_spec(net, batch=2))
    run.inject([Message("A", "B", ph.Request(0.0))])
    run.resolve()
    assert (run.err == _abi.ERR_ROUND_LIMIT).all()
    assert (run.msg_count == _abi.MAX_ROUNDS + 1).all()


ALL_KATS = [kat_message_cycle, kat_ignore_connection_errors, kat_tracking, kat_call_response, kat_ordering, kat_round_limit,
            kat_invalid_response_connection, kat_unknown_message_type, kat_env_step,
            kat_fsm_odd_even_two_agents, kat_fsm_odd_even_one_agent, kat_fsm_one_state,
            kat_stackelberg, kat_payload_whitelist]

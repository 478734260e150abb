"""Round 6 on the GPU, part 3: a policy evaluated on the device inside the fused rollout (phx_rollout_io.policy, ABI 10;
phx_sc_rollout_policy_kernel) -- bit for bit against the oracle's term-by-term restatement of phx_policy_mlp, and against the same
network evaluated by torch on the observations the rollout produced (the reference's collection loop, utils/rllib/rollout.py:300-363)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import phantom_amd as ph
from device_runner import DeviceRunner
from helpers import f32_bits, market_env, supply_chain_env
from oracle import OracleEnv

pytestmark = pytest.mark.gpu
NCPU = min(os.cpu_count() or 1, 128)
STATE = ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick")
KERNEL = "phx_sc_rollout_policy_kernel"


def _policy(widths, act, seed, scale=60.0, bias=45.0):
    rng = np.random.default_rng(seed)
    dims = [3] + list(widths) + [1]
    ws = [rng.normal(0, 1.2 / np.sqrt(dims[l]), (dims[l + 1], dims[l])).astype(np.float32) for l in range(len(dims) - 1)]
    bs = [rng.normal(0, 0.3, (dims[l + 1],)).astype(np.float32) for l in range(len(dims) - 1)]
    return ph.MLPPolicy(ws, bs, activation=act, out_scale=scale, out_bias=bias, out_lo=0.0, out_hi=100.0)


def _cmp(rd, ro, what):
    for k in ("obs", "actions", "rewards"):
        np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=f"{what}: {k}")
    np.testing.assert_array_equal(rd["truncated"], ro["truncated"], err_msg=what); np.testing.assert_array_equal(rd["terminated"], ro["terminated"], err_msg=what)
    np.testing.assert_array_equal(f32_bits(rd["last_obs"]), f32_bits(ro["last_obs"]), err_msg=what)


@pytest.mark.parametrize("widths,act", [((32,), "relu"), ((64, 64), "relu"), ((5,), "hard_tanh"), ((8, 3), "hard_tanh"), ((1,), "relu"), ((17, 33), "relu"), ((16, 12), "relu"), ((24, 8), "hard_tanh")])
@pytest.mark.parametrize("S,ks,B,ns", [(9, [6] * 9, 61, 23), (51, [4] * 51, 9, 30), (3, [2, 7, 1], 100, 11)])
def test_on_policy_rollout_matches_the_oracle(widths, act, S, ks, B, ns):
    """T on-policy steps in one launch == the oracle's (policy on the previous observation, then env.step), bit for bit: the actions the
    MLP produced, every plane, last_obs and the state; over episode ends (the reset observation feeds the first step of the next episode),
    twice in a row (the second fragment starts from the first one's state), then with replayed order sizes."""
    env = supply_chain_env(S, ks, ns, B, seed=31 + S, env_offset=7)
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    o.reset(); d.reset()
    pol = _policy(widths, act, seed=S + len(widths))
    for rep, T in enumerate((2 * ns + 3, 17)):
        ro, rd = o.rollout(T, policy=pol), d.rollout(T, policy=pol)
        assert d.dev.last_kernel() == KERNEL, d.dev.last_kernel()
        _cmp(rd, ro, f"rep {rep}")
        for f in STATE:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} rep {rep}")
        assert widths[0] < 5 or len(np.unique(rd["actions"])) > 10      # (a policy that actually depends on its input)
    exo = np.random.default_rng(1).integers(0, 5, (12, B, d.n_exo)).astype(np.uint8)
    _cmp(d.rollout(12, None, exo, policy=pol), o.rollout(12, None, exo, policy=pol), "replayed order sizes")
    assert (d.err == 0).all()


def test_on_policy_rollout_at_the_bench_shape_and_against_torch():
    """SC64, B = 4096, T = 100 with a 3-32-1 policy: every row against the oracle; and the actions against the SAME network evaluated by
    torch on the observations of the previous row (teacher forcing: the two differ by the order of the additions only)."""
    import torch
    S, K, B, T = 9, 6, 4096, 100
    env = supply_chain_env(S, [K] * S, 100, B, seed=42)
    o, d = OracleEnv(env.spec, threads=NCPU), DeviceRunner(env.spec)
    o.reset(); d.reset()
    pol = _policy((32,), "relu", seed=0)
    first = d.dev.obs.clone()                                      # the reset observation: the policy's input at the first step
    ro, rd = o.rollout(T, policy=pol), d.rollout(T, policy=pol)
    assert d.dev.last_kernel() == KERNEL
    _cmp(rd, ro, "bench shape")
    for f in STATE:
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)
    obs = torch.from_numpy(rd["obs"]).to(d.dev.device)
    prev = torch.cat([first[None], obs[:-1]])                      # (no episode ends inside: T == num_steps, the last row is the terminal one)
    want = pol(prev).cpu().numpy()
    np.testing.assert_allclose(rd["actions"], want, rtol=2e-5, atol=2e-4)
    # the torch module the policy was built from gives the same numbers
    net = torch.nn.Sequential(torch.nn.Linear(3, 32), torch.nn.ReLU(), torch.nn.Linear(32, 1)).to(d.dev.device)
    p2 = ph.MLPPolicy.from_torch(net, out_scale=50.0, out_bias=50.0)
    d.reset()
    first2 = d.dev.obs.clone()
    r2 = d.rollout(10, policy=p2)
    with torch.no_grad():
        y = net(torch.cat([first2[None], torch.from_numpy(r2["obs"]).to(d.dev.device)[:-1]])).squeeze(-1)
    np.testing.assert_allclose(r2["actions"], torch.clamp(y * 50.0 + 50.0, 0.0, 100.0).cpu().numpy(), rtol=2e-5, atol=2e-4)


def test_policy_update_and_argument_errors():
    from phantom_amd.device import DeviceError
    S, B = 9, 32
    env = supply_chain_env(S, [6] * S, 20, B, seed=3)
    o, d = OracleEnv(env.spec, threads=4), DeviceRunner(env.spec)
    o.reset(); d.reset()
    pol = _policy((16,), "relu", seed=5)
    out = d.dev.alloc_trajectory(8)
    d.dev.rollout(8, out=out, policy=pol); o.rollout(8, policy=pol)
    p2 = _policy((16,), "relu", seed=6)
    pol.update(p2.weights, p2.biases)                             # a learner's update: same argument block, new parameters
    d.dev.rollout(8, out=out, policy=pol)
    ro = o.rollout(8, policy=pol)
    np.testing.assert_array_equal(f32_bits(out.actions.cpu().numpy()), f32_bits(ro["actions"]))
    np.testing.assert_array_equal(f32_bits(out.observations.cpu().numpy()), f32_bits(ro["obs"]))
    with pytest.raises(ValueError):
        d.dev.rollout(4, actions=out.actions[:4].contiguous(), policy=pol)
    with pytest.raises(ValueError):
        ph.MLPPolicy([np.zeros((65, 3)), np.zeros((1, 65))], [np.zeros(65), np.zeros(1)])
    with pytest.raises(ValueError):
        ph.MLPPolicy([np.zeros((4, 3)), np.zeros((1, 4))], [np.zeros(4), np.zeros(1)], out_lo=-1.0)
    fsm = DeviceRunner(supply_chain_env(3, [2] * 3, 10, 8, fsm=True).spec); fsm.reset()
    with pytest.raises(DeviceError):
        fsm.dev.rollout(4, policy=_policy((4,), "relu", 1))
    mk = DeviceRunner(market_env(4, 8, 2, 6, 4).spec); mk.reset()
    with pytest.raises((DeviceError, ValueError)):
        mk.dev.rollout(4, policy=_policy((4,), "relu", 1))

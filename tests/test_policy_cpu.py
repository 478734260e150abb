"""phx_policy_mlp without a GPU: the oracle's term-by-term restatement against a float64 evaluation of the same network, the ctypes
mirror of the struct, and the policy through the CPU restatement's own phx_rollout (oracle/libphantom_cpu.so)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import phantom_amd as ph
from phantom_amd import _abi
from helpers import supply_chain_env
from oracle import OracleEnv


def _policy(widths, act, seed):
    rng = np.random.default_rng(seed)
    dims = [3] + list(widths) + [1]
    ws = [rng.normal(0, 1.0 / np.sqrt(dims[l]), (dims[l + 1], dims[l])).astype(np.float32) for l in range(len(dims) - 1)]
    bs = [rng.normal(0, 0.3, (dims[l + 1],)).astype(np.float32) for l in range(len(dims) - 1)]
    return ph.MLPPolicy(ws, bs, activation=act, out_scale=55.0, out_bias=40.0)


def _f64_eval(pol, x):
    h = x.astype(np.float64)
    for l in range(len(pol.weights) - 1):
        h = h @ pol.weights[l].astype(np.float64).T + pol.biases[l]
        h = np.clip(h, -1, 1) if pol.activation == "hard_tanh" else np.maximum(h, 0)
    y = (h @ pol.weights[-1].astype(np.float64).T + pol.biases[-1])[..., 0]
    return np.clip(y * pol.out_scale + pol.out_bias, pol.out_lo, pol.out_hi)


def test_struct_mirror():
    assert C.sizeof(_abi.PhxPolicyMLP) == 16 + 16 + 48
    assert _abi.PhxRolloutIO.policy.offset == _abi.PhxRolloutIO.frags.offset + 8


@pytest.mark.parametrize("widths,act", [((32,), "relu"), ((16, 8), "hard_tanh"), ((64, 64), "relu")])
def test_oracle_policy_rollout_follows_the_network(widths, act):
    S, B, ns, T = 4, 24, 9, 25
    env = supply_chain_env(S, [3] * S, ns, B, seed=5)
    o = OracleEnv(env.spec, threads=2)
    first, _ = o.reset()
    pol = _policy(widths, act, 1)
    ro = o.rollout(T, policy=pol)
    prev = np.concatenate([first[None], ro["obs"][:-1]])
    # after an episode's last row the policy sees the RESET observation (stock 0, the last step's sales), not the row's own
    ends = np.flatnonzero(ro["truncated"][:-1, 0, 0])
    for t in ends:
        prev[t + 1, ..., 0] = 0.0
    np.testing.assert_allclose(ro["actions"], _f64_eval(pol, prev), rtol=1e-5, atol=1e-4)
    assert ro["truncated"].sum() == (T // ns) * B * S and np.unique(ro["actions"]).size > 20
    # deterministic, and the state continues
    o2 = OracleEnv(env.spec, threads=1); o2.reset()
    np.testing.assert_array_equal(o2.rollout(T, policy=pol)["actions"], ro["actions"])

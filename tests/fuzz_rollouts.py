"""Random differential ROLLOUT cases: the fused rollout kernels (supply chain plain / FSM in every variant the plans pick
-- whole-env and pair-range blocks of the time-parallel kernel, its round-1 fallback for ragged shops, the lean and the general
FSM loop -- and the Stackelberg market) against the CPU oracle, bit-exact, on shapes and fragment lengths drawn from a seed.
Used by tests/test_gpu_fuzz.py; `python tests/fuzz_rollouts.py LO HI` runs a sweep by hand."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from helpers import f32_bits, market_env, supply_chain_env
from oracle import OracleEnv


def _cmp(rd, ro, valid):
    for k in ("obs", "actions", "rewards", "last_obs"):
        np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=k)
    for k in ("truncated", "terminated") + (("obs_valid", "reward_valid") if valid else ()):
        np.testing.assert_array_equal(rd[k], ro[k], err_msg=k)


def run_rollout_case(case):
    """returns the number of env-steps compared"""
    from device_runner import DeviceRunner
    rng = np.random.default_rng(case)
    kind = int(rng.integers(0, 10))
    if kind < 8:
        fsm = bool(rng.integers(0, 2))
        S = int(rng.choice([1, 2, 3, 5, 7, 9, 12, 16, 20, 25, 30, 51, 64, 100]))
        K = int(rng.integers(1, 7))
        Ks = [K] * S if rng.random() < 0.8 else [int(rng.integers(1, 9)) for _ in range(S)]
        B = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 20, 32, 48, 64]))
        ns = int(rng.choice([1, 2, 5, 19, 20, 21, 40, 57, 100]))
        # a kernel variant drawn from the seed (phx_spec.variant_*; ignored where its preconditions do not hold)
        vrng = np.random.default_rng(case + 10_000_019)
        variants = {"rollout": str(vrng.choice(["auto", "auto", "time_parallel", "lean", "general"])),
                    "block": [0, 0, "whole_envs", 16, 32, 48, 64, 36][int(vrng.integers(0, 8))],
                    "flags": ["auto", "dense", "sparse", "sparse"][int(vrng.integers(0, 4))]}
        env = supply_chain_env(S, Ks, ns, B, fsm=fsm, seed=int(rng.integers(0, 1000)), env_offset=int(rng.integers(0, 5000)),
                               variants=variants)
        fields = ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick")
        amax, valid = 100.0, fsm
    else:
        L = int(rng.choice([2, 4, 8, 16])); d = min(int(rng.choice([1, 2, 4])), L)
        Fw = int(rng.choice([4, 8, 32, 100])); B = int(rng.choice([1, 3, 8, 16])); ns = int(rng.choice([2, 7, 10, 33]))
        env = market_env(L, Fw, d, ns, B, seed=int(rng.integers(0, 1000)), env_offset=int(rng.integers(0, 5000)))
        fields = ("env.step", "env.tick")
        amax, valid = 1.0, True
    o, dv = OracleEnv(env.spec, threads=4), DeviceRunner(env.spec)
    assert dv.dev.uses_fused
    o.reset(); dv.reset()
    for _ in range(int(rng.integers(0, 4))):                      # fragments that do not start on a tick quad / at a reset
        a = rng.uniform(0, amax, (B, env.spec.n_strategic)).astype(np.float32)
        o.step(a, None, None); dv.step(a, None, None)
    n = 0
    for _ in range(int(rng.integers(1, 4))):
        T = int(rng.choice([1, 2, 3, 7, 19, 20, 21, 39, 41, 64, 100, 130]))
        ro, rd = o.rollout(T), dv.rollout(T)
        _cmp(rd, ro, valid)
        for f in fields:
            np.testing.assert_array_equal(dv.get_i32(f), o.get_i32(f), err_msg=f"case {case}: {f} after T={T}")
        n += T * B
    assert (dv.err == 0).all()
    return n


if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    total = sum(run_rollout_case(c) for c in range(lo, hi))
    print(f"rollout cases {lo}..{hi}: ok, {total} env-steps compared")

"""On-policy rollouts with the policy evaluated on the device (phx_rollout_io.policy): time per step at the bench shape.
    python tools/policy_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phantom_amd as ph
from helpers import supply_chain_env


def ev(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pol(widths, seed=0):
    rng = np.random.default_rng(seed)
    dims = [3] + list(widths) + [1]
    ws = [rng.normal(0, 1 / np.sqrt(dims[l]), (dims[l + 1], dims[l])).astype(np.float32) for l in range(len(dims) - 1)]
    bs = [rng.normal(0, .3, (dims[l + 1],)).astype(np.float32) for l in range(len(dims) - 1)]
    return ph.MLPPolicy(ws, bs, out_scale=60.0, out_bias=45.0)


for name, S, K, B in (("SC64 B=4096", 9, 6, 4096), ("SC64 B=65536", 9, 6, 65536), ("SC256 B=8192", 51, 4, 8192)):
    env = supply_chain_env(S, [K] * S, 100, B, seed=1, exogenous="device")
    d = env._device(); env.reset()
    T = 100
    tr = d.alloc_trajectory(T)
    for widths in ((8,), (16,), (32,), (64,), (8, 8), (32, 32), (64, 64)):
        p = pol(widths)
        us = ev(lambda: d.rollout(T, out=tr, policy=p), 5)
        by = 22 * S * B * T
        print(f"{name:14s} policy 3-{'-'.join(map(str, widths))}-1  {us / T:8.3f} us/step  {by / us / 1e3 / 8000:.3f} of 8 TB/s  {(1 + S + S * K) * B * T / us * 1e6:.3e} agent-steps/s  [{d.last_kernel()}]", flush=True)
    del env, d, tr

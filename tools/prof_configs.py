#!/usr/bin/env python
"""The fused rollout launches of BASELINE.json's configs 3, 4 (one GPU's share) and 5, and nothing else: the command the round-5
kernel trace and the WRITE_SIZE / FETCH_SIZE passes of those kernels are taken on (tools/profile_r05.sh), so that a per-kernel average
in the trace is one launch shape.    python tools/prof_configs.py [c3] [c4] [c5] [gen]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_amd as ph
from helpers import market_env, supply_chain_env

which = set(sys.argv[1:]) or {"c3", "c4", "c5"}


def ev(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def line(name, T, B, bytes_per_env_step, us, dev):
    alg = bytes_per_env_step * B * T
    print(f"{name:44s} T={T:4d} {us:9.1f} us/launch  {alg / 1e6:8.1f} MB algorithmic  {alg / us / 1e3 / 8000:.3f} of 8 TB/s   [{dev.last_kernel()}]", flush=True)
    if dev.autotune_note():
        print(f"{'':44s} PHX_VR_AUTO measured: {dev.autotune_note()}", flush=True)


if "c3" in which:
    env = ph.SupplyChainFSMEnv(n_shops=51, customers_per_shop=4, num_steps=100, batch_size=8192, seed=42, exogenous="device")
    env.reset(); d = env._device()
    for T, n in ((100, 20), (400, 8)):
        trs = [d.alloc_trajectory(T) for _ in range(2)]
        k = [0]
        def f():
            d.rollout(T, out=trs[k[0] & 1]); k[0] += 1
        line("config 3: SC256 FSM B=8192", T, 8192, 24 * 51, ev(f, n), d)
        del trs
    fr = [d.alloc_trajectory(100) for _ in range(4)]
    line("config 3: SC256 FSM B=8192, 4 x 100 frags", 400, 8192, 24 * 51, ev(lambda: d.rollout_fragments(100, fr), 8), d)
    del fr
    del env, d; torch.cuda.empty_cache()
if "c4" in which:
    env = ph.SupplyChainEnv(n_shops=51, customers_per_shop=4, num_steps=100, batch_size=8192, seed=42, exogenous="device")
    env.reset(); d = env._device()
    for T, n in ((100, 20), (400, 8)):
        trs = [d.alloc_trajectory(T) for _ in range(2)]
        k = [0]
        def f():
            d.rollout(T, out=trs[k[0] & 1]); k[0] += 1
        line("config 4 share: SC256 B=8192", T, 8192, 22 * 51, ev(f, n), d)
        del trs
    fr = [d.alloc_trajectory(100) for _ in range(4)]
    line("config 4 share: SC256 B=8192, 4 x 100 frags", 400, 8192, 22 * 51, ev(lambda: d.rollout_fragments(100, fr), 8), d)
    del env, d, fr; torch.cuda.empty_cache()
if "c5" in which:
    env = market_env(128, 1024, 8, 100, 4096, exogenous="device")
    env.reset(); d = env._device()
    tr = d.alloc_trajectory(50)
    line("config 5: Stackelberg 128x1024 B=4096", 50, 4096, 20 * 1152, ev(lambda: d.rollout(50, out=tr), 4), d)
    del tr
    tr = d.alloc_trajectory(100)
    line("config 5: Stackelberg 128x1024 B=4096", 100, 4096, 20 * 1152, ev(lambda: d.rollout(100, out=tr), 3), d)
    del env, d, tr; torch.cuda.empty_cache()
if "gen" in which:       # the generic engine where more waves no longer help (tools/gen_time.py sweep)
    B, S, K = 65536, 9, 6
    env = supply_chain_env(S, [K] * S, 100, B, force_generic=True, seed=1, exogenous="device")
    d = env._device(); env.reset()
    a = torch.rand(B, S, device="cuda") * 100
    us = ev(lambda: d.step(a), 30)
    print(f"{'generic engine SC64 B=65536':44s} phx_step {us:9.1f} us/launch  {B / us:.1f} env-steps/us   [{d.last_kernel()}]", flush=True)

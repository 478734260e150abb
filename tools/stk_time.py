#!/usr/bin/env python
"""Config 5 (Stackelberg 128 x 1024) fused rollouts: launch time against T and B -- the fixed cost of a launch (per-block setup, the last
round's tail) against the marginal cost of a step.    python tools/stk_time.py [B,B,..] [T,T,..]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_amd as ph
from helpers import market_env


def ev(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


BS = tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1024, 2048, 4096, 8192)
TS = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (2, 10, 25, 50, 100)
for B in BS:
    env = market_env(128, 1024, 8, 100, B, exogenous="device")
    env.reset(); d = env._device()
    prev = None
    for T in TS:
        tr = d.alloc_trajectory(T)
        us = ev(lambda: d.rollout(T, out=tr), 4)
        alg = 20 * 1152 * B * T
        marg = "" if prev is None else f"  marginal {(us - prev[1]) / (T - prev[0]):7.2f} us/step = {20 * 1152 * B / ((us - prev[1]) / (T - prev[0])) / 1e3 / 8000:.3f}"
        print(f"B={B:5d} T={T:4d} {us:9.1f} us/launch  {alg / us / 1e3 / 8000:.3f} of 8 TB/s{marg}   [{d.last_kernel()}]", flush=True)
        prev = (T, us)
        del tr
    del env, d; torch.cuda.empty_cache()

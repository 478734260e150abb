#!/usr/bin/env python
"""The lane-per-pair FSM rollout loop (phx_sc_rollout_fsm_kernel) with a rule-form stage handler: time per step.    python tools/rules_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phantom_amd as ph


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


def restock_below(thr):
    def restock(env_):
        env_.resolve_network()
        tot = sum(np.asarray(a_.stock) for aid, a_ in env_.agents.items() if str(aid).startswith("SHOP"))
        return np.where(tot < thr, "RESTOCK", "SELL").tolist()
    return restock


for S, K, B, thr in ((9, 6, 4096, 300), (9, 6, 65536, 300), (51, 4, 8192, 1700)):
    handler = ph.state_rules([ph.StageRule("shop.stock", "<", thr, "RESTOCK")])(restock_below(thr))
    env = ph.SupplyChainFSMEnv(n_shops=S, customers_per_shop=K, num_steps=100, batch_size=B, seed=42, exogenous="device", restock_handler=handler)
    env.reset(); d = env._device()
    for T in (50, 100):
        tr = d.alloc_trajectory(T)
        us = ev(lambda: d.rollout(T, out=tr), 5)
        print(f"S={S:3d} K={K} B={B:6d} rule-form handler  T={T:4d} {us:9.1f} us/launch {us / T:7.2f} us/step  {24 * S * B * T / us / 1e3 / 8000:.3f} of 8 TB/s  [{d.last_kernel()}]", flush=True)
        del tr
    del env, d; torch.cuda.empty_cache()

#!/usr/bin/env python
"""Stress of the compiled schedule's hand-over of flagged envs (phx_generic_sched.hip: schedule workgroups publish a word per env, the launch's
tail workgroups run the dynamic engine over the flagged ones): per-step launches with random partial action masks on a large batch, the
compiled-schedule env against the SAME env on the dynamic engine alone (variants={"step": "generic_dynamic"}), every output and the state
compared on the device after every step.    python tools/engine_selfcheck.py [steps=3000]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import supply_chain_env

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bad = 0
for S, K, B, fsm, ns in ((9, 6, 4096, False, 7), (5, 4, 8192, True, 1), (9, 6, 2048, True, 5), (51, 4, 512, True, 9)):
    a_env = supply_chain_env(S, [K] * S, ns, B, fsm=fsm, force_generic=True, seed=3, exogenous="device")
    b_env = supply_chain_env(S, [K] * S, ns, B, fsm=fsm, seed=3, exogenous="device", variants={"step": "generic_dynamic"})
    a_env.reset(); b_env.reset()
    da, db = a_env._device(), b_env._device()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    names = ("obs", "reward", "obs_valid", "reward_valid", "terminated", "truncated", "all_truncated", "all_terminated")
    fields = ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick") + (("env.stage",) if fsm else ())
    kern = set()
    for t in range(steps):
        acts = torch.rand(B, S, device="cuda", generator=g) * 100
        av = (torch.rand(B, S, device="cuda", generator=g) < 0.93).to(torch.uint8)
        oa = da.step(acts, av); kern.add(da.last_kernel())
        ob = db.step(acts, av)
        for n in names:
            x, y = getattr(oa, n, None), getattr(ob, n, None)
            if x is None or y is None:
                continue
            m = None
            if n == "obs": m = oa.obs_valid.bool().unsqueeze(-1).expand_as(x)
            if n == "reward": m = oa.reward_valid == 1
            same = torch.equal(x[m], y[m]) if m is not None else torch.equal(x, y)
            if not same:
                bad += 1; print(f"MISMATCH S={S} fsm={fsm} step {t}: {n}", flush=True)
        for f in fields:
            if not torch.equal(da.field(f), db.field(f)):
                bad += 1; print(f"MISMATCH S={S} fsm={fsm} step {t}: state {f}", flush=True)
        done = (oa.all_truncated | oa.all_terminated)
        if bool(done.any()):
            da.reset(done); db.reset(done)
        if bad > 5:
            break
    print(f"S={S} K={K} B={B} fsm={fsm} num_steps={ns}: {steps} steps, kernels {sorted(kern)}, mismatches so far {bad}", flush=True)
print("engine self-check:", "clean" if bad == 0 else f"{bad} mismatches")

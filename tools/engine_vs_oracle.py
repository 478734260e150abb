#!/usr/bin/env python
"""Long per-step comparison of the engine (compiled schedule + the dynamic engine for flagged envs) against the ORACLE on the shape of
fuzz case 25 025 203 (lease r06_5's unreproduced mismatch): S=5, Ks=[4,1,1,4,6], B=20, num_steps=1, FSM with a single-agent rule, random
partial action masks, a masked reset after every step.    python tools/engine_vs_oracle.py [steps=4000] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import phantom_amd as ph
from device_runner import DeviceRunner
from helpers import f32_bits, supply_chain_env
from oracle import OracleEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
S, Ks, B = 5, [4, 1, 1, 4, 6], 20
h = ph.state_rules([ph.StageRule("shop.missed_sales", "<", 15.0, "RESTOCK", agent="SHOP3")])(lambda env_: None)
h._phx_skip_check = True
bad = 0
for ns in (1, 2, 5):
    env = supply_chain_env(S, Ks, ns, B, fsm=True, seed=int(rng.integers(0, 1000)), env_offset=int(rng.integers(0, 5000)), force_generic=True, restock_handler=h)
    env._rules_checked = True
    o, d = OracleEnv(env.spec, threads=4), DeviceRunner(env.spec)
    o.reset(); d.reset()
    for t in range(steps):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        av = (rng.random((B, S)) < 0.85).astype(np.uint8) if rng.random() < 0.7 else None
        o.step(a, av, None); d.step(a, av, None)
        m = o.obs_valid.astype(bool)
        ok = (np.array_equal(d.obs_valid, o.obs_valid) and np.array_equal(f32_bits(d.obs[m]), f32_bits(o.obs[m])) and
              np.array_equal(d.reward_valid, o.reward_valid) and np.array_equal(d.reward[o.reward_valid == 1].view(np.uint64), o.reward[o.reward_valid == 1].view(np.uint64)) and
              all(np.array_equal(d.get_i32(f), o.get_i32(f)) for f in ("shop.stock", "shop.sales", "shop.missed_sales", "env.stage", "env.step")))
        if not ok:
            bad += 1
            rows = np.nonzero((f32_bits(d.obs) != f32_bits(o.obs)).reshape(B, -1).any(1))[0]
            print(f"MISMATCH num_steps={ns} step {t}: envs {rows.tolist()} mask rows {None if av is None else av[rows].tolist()} kernel {d.dev.last_kernel()}", flush=True)
            if os.environ.get("PHX_FUZZ_DUMP"):
                np.savez(os.path.join(os.environ["PHX_FUZZ_DUMP"], f"evo_{seed}_{ns}_{t}.npz"), a=a, av=(av if av is not None else np.zeros(0)), dev_obs=d.obs, ora_obs=o.obs,
                         dev_stock=d.get_i32("shop.stock"), ora_stock=o.get_i32("shop.stock"))
            break
        done = (o.all_truncated | o.all_terminated).astype(np.uint8)
        if done.any():
            o.reset(done); d.reset(done)
    print(f"num_steps={ns}: {steps} steps compared, mismatches so far {bad}", flush=True)
print("engine vs oracle:", "clean" if bad == 0 else f"{bad} mismatches")

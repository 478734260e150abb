#!/usr/bin/env python
"""Time phx_rollout launches of one supply-chain config (event pair over N back-to-back launches) and print a
checksum of the fragment, so kernel variants (env knobs, alternative .so via PHX_LIB) can be A/B-compared in one
gpurun call.   python tools/roll_time.py [--shops 9 --cust 6 --batch 4096 --T 100 --n 200 --fsm]"""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import phantom_amd as ph

ap = argparse.ArgumentParser()
ap.add_argument("--shops", type=int, default=9); ap.add_argument("--cust", type=int, default=6)
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--T", type=int, default=100)
ap.add_argument("--n", type=int, default=200); ap.add_argument("--fsm", action="store_true")
ap.add_argument("--tag", default="")
a = ap.parse_args()
cls = ph.SupplyChainFSMEnv if a.fsm else ph.SupplyChainEnv
env = cls(n_shops=a.shops, customers_per_shop=a.cust, num_steps=100, batch_size=a.batch, seed=42, exogenous="device")
env.reset(); dev = env._device()
tr = dev.rollout(a.T)
torch.cuda.synchronize()
h = hashlib.sha1()
for x in (tr.observations, tr.actions, tr.rewards, tr.truncations):
    h.update(x.cpu().numpy().tobytes())
for _ in range(20):
    dev.rollout(a.T, out=tr)
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.n):
        dev.rollout(a.T, out=tr)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / a.n * 1e3)
S = a.shops
alg = a.batch * a.T * 22 * S + a.batch * (S * 32 + 16)
print(f"{a.tag:28s} {best:8.2f} us/launch  {alg / best / 1e3 / 8000:.3f} of 8 TB/s  sha {h.hexdigest()[:12]}", flush=True)

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
ap.add_argument("--bufs", type=int, default=0, help="trajectory buffers rotated over (0: enough for > 320 MB)")
ap.add_argument("--block", default="0", help="variants['block']: pairs per workgroup, whole_envs, or 0 = auto")
ap.add_argument("--rollout", default="auto", help="variants['rollout']")
ap.add_argument("--frags", type=int, default=1, help="fragments of T rows per call (phx_rollout_io.frags)")
ap.add_argument("--vouch", type=int, default=1, help="1: the replayed inputs carry PHX_RH_ACTIONS_IN_DOMAIN / PHX_RH_EXO_IN_DOMAIN")
ap.add_argument("--replay", default="", help="a: replayed actions, x: replayed order sizes, ax: both (phx_rollout_io.actions / exo)")
a = ap.parse_args()
cls = ph.SupplyChainFSMEnv if a.fsm else ph.SupplyChainEnv
blk = a.block if a.block == "whole_envs" else int(a.block)
env = cls(n_shops=a.shops, customers_per_shop=a.cust, num_steps=100, batch_size=a.batch, seed=42, exogenous="device",
          variants={"block": blk, "rollout": a.rollout})
env.reset(); dev = env._device()
S = a.shops
PB = 24 if a.fsm else 22                    # bytes per pair and step: the FSM's obs_valid / reward_valid planes
alg = a.batch * a.T * PB * S + a.batch * (S * 32 + 16)
nb = a.bufs or max(2, -(-(320 << 20) // alg))
trs = [dev.alloc_trajectory(a.T) for _ in range(nb)]
tr = dev.rollout(a.T, out=trs[0])
torch.cuda.synchronize()
h = hashlib.sha1()
for x in (tr.observations, tr.actions, tr.rewards, tr.truncations):
    h.update(x.cpu().numpy().tobytes())
k = a.frags
ra = (torch.rand(a.T, a.batch, S, device=dev.device) * 100.0).contiguous() if "a" in a.replay else None
rx = torch.randint(0, 5, (a.T, a.batch, S * a.cust), dtype=torch.uint8, device=dev.device) if "x" in a.replay else None
if a.replay:
    call = lambda i: dev.rollout(a.T, ra, rx, out=trs[i % nb], actions_in_domain=bool(a.vouch), exo_in_domain=bool(a.vouch))
elif k > 1:                      # k fragments per call, the call's buffers rotated like single fragments
    nb = max(nb, 2 * k) // k * k
    trs = trs + [dev.alloc_trajectory(a.T) for _ in range(nb - len(trs))]
    call = lambda i: dev.rollout_fragments(a.T, trs[(i * k) % nb:(i * k) % nb + k])
else:
    call = lambda i: dev.rollout(a.T, out=trs[i % nb])
for i in range(20):
    call(i)
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.n):
        call(i)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / a.n * 1e3)
tot = a.T * k
alg_call = a.batch * tot * (PB * S + (4 * S if "a" in a.replay else 0) + (S * a.cust if "x" in a.replay else 0)) + a.batch * (S * 32 + 16)   # (+ the replayed inputs' reads)
print(f"{a.tag:28s} T={a.T:5d} x{k} bufs={nb:2d} {best:8.2f} us/call  {best * 100 / tot:7.2f} us/100 steps  {alg_call / best / 1e3 / 8000:.3f} of 8 TB/s  sha {h.hexdigest()[:12]}  {dev.last_kernel()}", flush=True)

#!/usr/bin/env python
"""Does the rollout's speed depend on WHERE a trajectory fragment lies?  Allocates N fragments (SC64, B = 4096, T steps), times
the same kernel into each of them alone (a fragment of T >= 400 steps is larger than the 256 MB Infinity Cache), then rotating
over the fastest / slowest pair and over all of them.    python tools/roll_place.py [--T 400 --n 8 --block 48 --flat]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import phantom_amd as ph

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=400); ap.add_argument("--n", type=int, default=8)
ap.add_argument("--block", default="0"); ap.add_argument("--flat", action="store_true")
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
blk = a.block if a.block == "whole_envs" else int(a.block)
env = ph.SupplyChainEnv(n_shops=9, customers_per_shop=6, num_steps=100, batch_size=4096, seed=42, exogenous="device", variants={"block": blk})
env.reset(); dev = env._device()
bufs = [dev.alloc_trajectory(a.T, flat=a.flat) for _ in range(a.n)]
alg = 4096 * a.T * 22 * 9 + 4096 * (9 * 32 + 16)

def timed(sel, reps=a.reps):
    for i in range(4): dev.rollout(a.T, out=bufs[sel[i % len(sel)]])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): dev.rollout(a.T, out=bufs[sel[i % len(sel)]])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

single = [timed([k]) for k in range(a.n)]
for k, t in enumerate(single):
    b = bufs[k]
    print(f"fragment {k}: {t:7.2f} us alone = {alg / t / 8e6:.3f} of 8 TB/s   obs@{b.observations.data_ptr():#x} act@{b.actions.data_ptr():#x} rew@{b.rewards.data_ptr():#x} "
          f"ter@{b.terminations.data_ptr():#x} tru@{b.truncations.data_ptr():#x}")
order = sorted(range(a.n), key=lambda k: single[k])
print(f"rotating over the 2 fastest {order[:2]}: {timed(order[:2]):7.2f} us;  the 2 slowest {order[-2:]}: {timed(order[-2:]):7.2f} us;  all {a.n}: {timed(list(range(a.n))):7.2f} us")
print(f"{a.kernel if hasattr(a, 'kernel') else dev.last_kernel()}")

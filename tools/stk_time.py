#!/usr/bin/env python
"""Time the Stackelberg-market kernels (BASELINE config 5: 128 sellers x 1024 buyers, B = 4096): one launch per step and
the fused rollout, with a checksum of the outputs so kernel variants (PHX_LIB_PATH) can be A/B-compared in one gpurun call.
   python tools/stk_time.py [--batch 4096 --T 50 --tag name]"""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import market_env

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--T", type=int, default=50)
ap.add_argument("--tag", default=""); ap.add_argument("--which", default="both")
a = ap.parse_args()
B, S = a.batch, 1152
env = market_env(128, 1024, 8, 100, B)
d = env._device(); env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(3)
acts = [torch.rand(B, S, device="cuda", generator=g) for _ in range(4)]
valid = [torch.zeros(B, S, dtype=torch.uint8, device="cuda") for _ in range(2)]
valid[0][:, :128] = 1; valid[1][:, 128:] = 1
h = hashlib.sha1()
if a.which in ("both", "step"):
    for t in range(8):
        st = d.step(acts[t % 4], action_valid=valid[t % 2])
        for x in (st.observations, st.rewards, st.obs_valid, st.reward_valid): h.update(x.cpu().numpy().tobytes())
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(40): d.step(acts[t % 4], action_valid=valid[t % 2])
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
    print(f"{a.tag:24s} step    {best:8.2f} us/step   sha {h.hexdigest()[:12]}", flush=True)
if a.which in ("both", "rollout"):
    h = hashlib.sha1()
    tr = d.rollout(a.T)
    for x in (tr.observations, tr.actions, tr.rewards, tr.truncations, tr.obs_valid, tr.reward_valid): h.update(x.cpu().numpy().tobytes())
    for _ in range(2): d.rollout(a.T, out=tr)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): d.rollout(a.T, out=tr)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (4 * a.T) * 1e3)
    print(f"{a.tag:24s} rollout {best:8.2f} us/step   sha {h.hexdigest()[:12]}", flush=True)

#!/usr/bin/env python
"""Time one-launch-per-step phx_step of a supply-chain config on the fused kernel (events around n launches, and the same
launches replayed from a hipGraph), with a checksum of the outputs for A/B runs (PHX_LIB_PATH).
   python tools/step_time.py [--shops 51 --cust 4 --batch 8192 --fsm --n 200 --tag name]"""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import phantom_amd as ph

ap = argparse.ArgumentParser()
ap.add_argument("--shops", type=int, default=51); ap.add_argument("--cust", type=int, default=4)
ap.add_argument("--batch", type=int, default=8192); ap.add_argument("--n", type=int, default=200)
ap.add_argument("--fsm", action="store_true"); ap.add_argument("--tag", default="")
a = ap.parse_args()
cls = ph.SupplyChainFSMEnv if a.fsm else ph.SupplyChainEnv
env = cls(n_shops=a.shops, customers_per_shop=a.cust, num_steps=100, batch_size=a.batch, seed=42, exogenous="device")
env.reset(); dev = env._device()
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = (torch.rand(50, a.batch, a.shops, device="cuda", generator=g) * 100.0).contiguous()
h = hashlib.sha1()
for t in range(10):
    st = dev.step(acts[t])
    for x in (st.observations, st.rewards, st.obs_valid, st.reward_valid): h.update(x.cpu().numpy().tobytes())
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(a.n): dev.step(acts[t % 50])
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / a.n * 1e3)
sg = dev.step_graph(acts)
sg.replay(); torch.cuda.synchronize()
bg = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): sg.replay()
    e1.record(); torch.cuda.synchronize()
    bg = min(bg, e0.elapsed_time(e1) / 200 * 1e3)
print(f"{a.tag:24s} {best:8.2f} us/step (launch loop)  {bg:8.2f} us/step (hipGraph)  sha {h.hexdigest()[:12]}", flush=True)

#!/usr/bin/env python
"""Time the fused digital-ads kernels at the reference example's size (1 exchange + 1 publisher + 120 advertisers with
clipped-sampler budgets, B = 4096): one launch per step and the fused rollout, with a checksum for A/B runs (PHX_LIB_PATH).
   python tools/ads_time.py [--batch 4096 --T 40 --tag name]"""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import phantom_amd as ph

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--T", type=int, default=40); ap.add_argument("--tag", default="")
a = ap.parse_args()
st = {}
for i in range(120):
    lo = (5.0, 7.0, 10.0)[i // 40]
    st[f"ADV_{i + 1}"] = ph.AdvertiserAgent.Supertype(budget=ph.UniformFloatSampler(lo, lo + 10.001, clip_low=lo, clip_high=lo + 10.0))
env = ph.DigitalAdsEnv(num_steps=20, num_agents_theme={"travel": 40, "tech": 40, "sport": 40}, agent_supertypes=st,
                       batch_size=a.batch, seed=42)
env.reset(); dev = env._device()
g = torch.Generator(device="cuda"); g.manual_seed(2)
acts = torch.rand(a.batch, 120, device="cuda", generator=g)
h = hashlib.sha1()
for t in range(6):
    s_ = dev.step(acts)
    for x in (s_.observations, s_.rewards, s_.obs_valid, s_.reward_valid): h.update(x.cpu().numpy().tobytes())
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(40): dev.step(acts)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
tr = dev.rollout(a.T)
for x in (tr.observations, tr.actions, tr.rewards, tr.truncations, tr.terminations): h.update(x.cpu().numpy().tobytes())
dev.rollout(a.T, out=tr); torch.cuda.synchronize()
br = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): dev.rollout(a.T, out=tr)
    e1.record(); torch.cuda.synchronize()
    br = min(br, e0.elapsed_time(e1) / (4 * a.T) * 1e3)
print(f"{a.tag:20s} step {best:8.2f} us   rollout {br:8.2f} us/step   sha {h.hexdigest()[:12]}", flush=True)

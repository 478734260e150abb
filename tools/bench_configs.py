#!/usr/bin/env python
"""Timings of BASELINE.json's other configs (parity-test cases, not the bench line):
per-launch step time of each config on one GPU, HIP events around N launches.

    python tools/bench_configs.py [--quick]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from helpers import market_env, supply_chain_env


def time_steps(dev, make_actions, n=200, warm=20, valid=None):
    acts = [make_actions(i) for i in range(8)]
    for i in range(warm):
        dev.step(acts[i % 8], valid[i % 2] if valid else None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for i in range(n):
        dev.step(acts[i % 8], valid[i % 2] if valid else None)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, (time.perf_counter() - t0) / n * 1e6


def main():
    quick = "--quick" in sys.argv
    out = []
    dev0 = torch.device("cuda:0")

    def sc(name, S, K, B, fsm, force_generic):
        env = supply_chain_env(S, [K] * S, 100, B, fsm=fsm, force_generic=force_generic, seed=1,
                               exogenous="device")
        d = env._device(); env.reset()
        us, wall = time_steps(d, lambda i: torch.rand(B, S, device=dev0) * 100)
        A = 1 + S + S * K
        out.append(dict(config=name, engine="generic" if force_generic else "fused", agents=A, batch=B,
                        us_per_step_events=us, us_per_step_wall=wall, agent_steps_per_s=A * B / (us * 1e-6)))
        print(json.dumps(out[-1]), flush=True)

    sc("SC64 B=4096 plain", 9, 6, 4096, False, False)
    sc("SC64 B=4096 plain", 9, 6, 4096, False, True)
    sc("SC256 B=8192 FSM", 51, 4, 8192, True, False)
    sc("SC256 B=8192 FSM", 51, 4, 8192, True, True)
    # config 3 as a fused FSM rollout (T = 100 steps per launch)
    S, K, B = 51, 4, 8192
    env = supply_chain_env(S, [K] * S, 100, B, fsm=True, seed=1, exogenous="device")
    d = env._device(); env.reset()
    traj = d.rollout(100)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        d.rollout(100, out=traj)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 1000 * 1e3
    out.append(dict(config="SC256 B=8192 FSM rollout T=100", engine="fused", agents=256, batch=B,
                    us_per_step_events=us, us_per_step_wall=us, agent_steps_per_s=256 * B / (us * 1e-6)))
    print(json.dumps(out[-1]), flush=True)
    del traj, d, env
    B = 512 if quick else 4096
    env = market_env(128, 1024, 8, 100, B)
    d = env._device(); env.reset()
    S = 1152
    valid = [torch.zeros(B, S, dtype=torch.uint8, device=dev0) for _ in range(2)]
    valid[0][:, :128] = 1      # odd steps: leaders act
    valid[1][:, 128:] = 1      # even steps: followers act
    us, wall = time_steps(d, lambda i: torch.rand(B, S, device=dev0), n=40, warm=4, valid=valid)
    out.append(dict(config=f"STK 128x1024 B={B}", engine="fused" if d.uses_fused else "generic", agents=S, batch=B,
                    us_per_step_events=us, us_per_step_wall=wall, agent_steps_per_s=S * B / (us * 1e-6)))
    print(json.dumps(out[-1]), flush=True)

    # config 5 as a fused rollout (T = 50 steps per launch, whole env state in LDS)
    T = 50
    traj = d.rollout(T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        d.rollout(T, out=traj)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (4 * T) * 1e3
    nbytes = sum(x.numel() * x.element_size() for x in traj if hasattr(x, "numel") and x is not traj.last_obs)
    out.append(dict(config=f"STK 128x1024 B={B} rollout T={T}", engine="fused", agents=S, batch=B,
                    us_per_step_events=us, us_per_step_wall=us, agent_steps_per_s=S * B / (us * 1e-6),
                    trajectory_GBps=nbytes / T / (us * 1e-6) / 1e9))
    print(json.dumps(out[-1]), flush=True)

    # the digital-ads market at the example's size (digital_ads_market.py:715-717: 40 + 40 + 40
    # advertisers, clipped-sampler budgets of the training configuration): generic engine, the exchange's
    # auction as an inbox reduction; per-step launches and the launch-loop rollout
    import phantom_amd as ph
    del traj, d, env
    B, T = (256 if quick else 4096), 40
    st = {}
    for i in range(120):
        lo = (5.0, 7.0, 10.0)[i // 40]
        st[f"ADV_{i + 1}"] = ph.AdvertiserAgent.Supertype(
            budget=ph.UniformFloatSampler(lo, lo + 10.001, clip_low=lo, clip_high=lo + 10.0))
    env = ph.DigitalAdsEnv(num_steps=20, num_agents_theme={"travel": 40, "tech": 40, "sport": 40},
                           agent_supertypes=st, batch_size=B, seed=1)
    d = env._device(); env.reset()
    us, wall = time_steps(d, lambda i: torch.rand(B, 120, device=dev0), n=40, warm=4)
    out.append(dict(config=f"ADS 122 agents B={B}", engine="fused" if d.uses_fused else "generic", agents=122, batch=B, us_per_step_events=us,
                    us_per_step_wall=wall, agent_steps_per_s=122 * B / (us * 1e-6)))
    print(json.dumps(out[-1]), flush=True)
    env.reset()
    traj = d.rollout(T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        d.rollout(T, out=traj)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (4 * T) * 1e3
    out.append(dict(config=f"ADS 122 agents B={B} rollout T={T}", engine="fused" if d.uses_fused else "generic launch loop", agents=122, batch=B,
                    us_per_step_events=us, us_per_step_wall=us, agent_steps_per_s=122 * B / (us * 1e-6),
                    clicks_per_env_step=float(traj.rewards.sum().item()) / (T * B)))
    print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()

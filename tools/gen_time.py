"""Time the generic message-passing engine (force_generic) on the two supply-chain shapes VERDICT names: per phx_step launch,
and per step of a phx_rollout (T-step loop in the kernel vs the one-launch-per-step loop).
    python tools/gen_time.py [sc64|sc256|both] [--roll-only]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import supply_chain_env


def ev(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(name, S, K, B, fsm, n=150, T=50, roll_only=False):
    if not roll_only:
        env = supply_chain_env(S, [K] * S, 100, B, fsm=fsm, force_generic=True, seed=1, exogenous="device")
        d = env._device(); env.reset()
        acts = [torch.rand(B, S, device="cuda") * 100 for _ in range(4)]
        k = [0]
        def one():
            d.step(acts[k[0] % 4]); k[0] += 1
        print(f"{name:28s} phx_step                     {ev(one, n):8.2f} us/step   [{d.last_kernel()}]", flush=True)
        # the same launches from a hipGraph (no per-step host work: what the GPU itself needs per step)
        ag = torch.stack([acts[i % 4] for i in range(50)]).contiguous()
        sg = d.step_graph(ag)
        print(f"{name:28s} phx_step, hipGraph of 50     {ev(sg.replay, 6) / 50:8.2f} us/step", flush=True)
        del env, d, sg
    for var in ("auto", "launch_loop"):
        env = supply_chain_env(S, [K] * S, 100, B, fsm=fsm, force_generic=True, seed=1, exogenous="device", variants={"rollout": var})
        d = env._device(); env.reset()
        tr = d.rollout(T)
        us = ev(lambda: d.rollout(T, out=tr), 4)
        print(f"{name:28s} phx_rollout T={T:3d} {var:12s} {us / T:8.2f} us/step   [{d.last_kernel()}]", flush=True)
        del env, d, tr


def sweep():
    """phx_generic_step_kernel by batch size: is the per-step time at B = 4096 one round of latency-bound waves (then the rate keeps
    rising with B) or a throughput limit?"""
    for name, S, K, fsm, Bs in (("SC64", 9, 6, False, (1024, 2048, 4096, 8192, 16384, 32768, 65536)), ("SC256-FSM", 51, 4, True, (2048, 4096, 8192, 16384, 32768))):
        A = 1 + S + S * K
        for B in Bs:
            env = supply_chain_env(S, [K] * S, 100, B, fsm=fsm, force_generic=True, seed=1, exogenous="device")
            d = env._device(); env.reset()
            acts = [torch.rand(B, S, device="cuda") * 100 for _ in range(4)]
            k = [0]
            def one():
                d.step(acts[k[0] % 4]); k[0] += 1
            us = ev(one, 60)
            print(f"{name:10s} generic phx_step B={B:6d} {us:9.2f} us/step  {B / us:8.1f} env-steps/us  {A * B / us * 1e6:.3e} agent-steps/s   [{d.last_kernel()}]", flush=True)
            del env, d, acts


which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which == "sweep":
    sweep(); sys.exit(0)
ro = "--roll-only" in sys.argv
if which in ("sc64", "both"): run("SC64 B=4096 generic", 9, 6, 4096, False, roll_only=ro)
if which in ("sc256", "both"): run("SC256-FSM B=8192 generic", 51, 4, 8192, True, n=90, roll_only=ro)

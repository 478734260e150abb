import os, sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from helpers import supply_chain_env
def run(name, S, K, B, fsm, n=150):
    env = supply_chain_env(S, [K] * S, 100, B, fsm=fsm, force_generic=True, seed=1, exogenous="device")
    d = env._device(); env.reset()
    acts = [torch.rand(B, S, device="cuda") * 100 for _ in range(4)]
    for i in range(20): d.step(acts[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): d.step(acts[i % 4])
    e1.record(); torch.cuda.synchronize()
    print(f"{name:30s} {e0.elapsed_time(e1) / n * 1e3:8.2f} us/step", flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("sc64", "both"): run("SC64 B=4096 generic", 9, 6, 4096, False)
if which in ("sc256", "both"): run("SC256-FSM B=8192 generic", 51, 4, 8192, True, n=90)

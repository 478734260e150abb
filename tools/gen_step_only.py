"""Forty phx_step launches of the generic engine (force_generic) on one supply-chain shape, for counter passes:
    python tools/gen_step_only.py [sc64|sc256]"""
import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from helpers import supply_chain_env
which = sys.argv[1] if len(sys.argv) > 1 else "sc64"
S, K, B, fsm = (9, 6, 4096, False) if which == "sc64" else (51, 4, 8192, True)
env = supply_chain_env(S, [K]*S, 100, B, fsm=fsm, force_generic=True, seed=1, exogenous="device")
d = env._device(); env.reset()
a = torch.rand(B, S, device="cuda") * 100
for _ in range(40): d.step(a)
torch.cuda.synchronize()

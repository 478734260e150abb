import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np
import phantom_amd as ph
from phantom_amd.rllib import BatchedBaseEnv
B = 4096
env = ph.SupplyChainEnv(n_shops=9, customers_per_shop=6, num_steps=100, batch_size=B, seed=42, exogenous="device")
be = BatchedBaseEnv(env)
env.reset(); be._pending = None
obs = be.poll()
ids = list(obs[0][0].keys())
def step():
    be.send_actions({b: {aid: 50.0 for aid in ids} for b in range(B)})
    o = be.poll()
    for b in range(B):
        row, rw = o[0][b], o[1][b]
        for aid in ids:
            row[aid], rw.get(aid)
for _ in range(3): step()
t0 = time.perf_counter()
for _ in range(5): step()
print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

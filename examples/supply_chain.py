#!/usr/bin/env python
"""The supply-chain example of the reference (examples/environments/supply_chain/supply_chain.py)
on the MI355X path, three ways.  Needs a GPU and the built library
(python -c "import __graft_entry__ as g; g.build()").

    python examples/supply_chain.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import phantom_amd as ph

# 1. drop-in: one env instance, the reference's dict API, the global numpy stream for the customers
np.random.seed(0)
env = ph.SupplyChainEnv()                       # 1 factory, 1 shop, 5 customers, 100 steps
obs, _ = env.reset()
total = 0.0
for t in range(env.num_steps):
    step = env.step({"SHOP": np.array([20.0], dtype=np.float32)})
    total += step.rewards["SHOP"]
print(f"B=1 dict API: episode return {total:.2f}, final stock {env['SHOP'].stock}, done {step.truncations['__all__']}")

# 2. batched: 4096 envs of the 64-agent benchmark topology, tensors stay on the GPU
env = ph.SupplyChainEnv(n_shops=9, customers_per_shop=6, batch_size=4096, seed=42, exogenous="device")
env.reset()
actions = torch.full((4096, 9), 30.0, device=env._device().device)
out = env.step(actions)                         # StepTensors: device views, no host sync
print("B=4096 tensor API: mean reward", float(out.rewards.mean()))

# 3. fused rollout: one episode of all 4096 envs per launch, trajectory [T, B, S, ...] on the GPU
traj = env.rollout(100)                         # random policy; pass actions=[T, B, S] to replay a policy
print("rollout:", tuple(traj.observations.shape), "mean reward", float(traj.rewards.mean()),
      "episode ends", int(traj.truncations[:, :, 0].sum()))

# 4. tutorial 2: per-env sampled reward weights (Supertype / Sampler), drawn on the device
weights = ph.UniformFloatSampler(0.0, 0.2)
env = ph.SupplyChainEnv(n_shops=3, customers_per_shop=4, batch_size=1024, typed=True, exogenous="device",
                        agent_supertypes={f"SHOP{i}": ph.TypedShopAgent.Supertype(weights) for i in range(3)})
env.reset()
print("typed shops: obs dim", env.spec.obs_dim, "weights of env 0..3", env["SHOP0"].type.excess_stock_weight[:4])

# 5. (ABI 9) eight one-episode fragments in separate buffers from ONE call: what a learner that takes 100-step fragments asks for
env = ph.SupplyChainEnv(n_shops=9, customers_per_shop=6, batch_size=4096, seed=42, exogenous="device")
env.reset()
dev = env._device()
bufs = [dev.alloc_trajectory(100) for _ in range(8)]
dev.rollout_fragments(100, bufs)
print("fragment list:", len(bufs), "x", tuple(bufs[0].observations.shape), "from", dev.last_kernel(),
      "| episode ends per fragment", [int(b.truncations[:, :, 0].sum()) for b in bufs][:3], "...")

# 6. (ABI 9) an FSM stage handler that branches on agent state, declared in rule form: the Python handler is the definition (checked
#    against the rule on random states when the device env is created), the device evaluates the rule inside step() and rollout()
def restock_again_while_stock_is_low(e):
    e.resolve_network()
    total = sum(np.asarray(a.stock) for aid, a in e.agents.items() if str(aid).startswith("SHOP"))
    return np.where(total < 60, "RESTOCK", "SELL").tolist()

handler = ph.state_rules([ph.StageRule("shop.stock", "<", 60, "RESTOCK")])(restock_again_while_stock_is_low)
env = ph.SupplyChainFSMEnv(n_shops=3, customers_per_shop=2, num_steps=20, batch_size=256, seed=1, exogenous="device", restock_handler=handler)
env.reset()
tr = env.rollout(40)
print("rule-form FSM handler: rollout in one launch,", env._device().last_kernel(), "| stages now:", sorted(set(env.current_stage)))

# 7. a handler-less FSM supply chain (RESTOCK -> SELL -> RESTOCK ...: BASELINE config 3's env) rolled out by the store-wave kernel's FSM
#    instantiation: obs_valid / reward_valid planes beside the five of a plain env; a fragment list is one launch here too
env = ph.SupplyChainFSMEnv(n_shops=9, customers_per_shop=6, num_steps=100, batch_size=2048, seed=3, exogenous="device",
                           variants={"rollout": "store_waves"})
env.reset()
dev = env._device()
bufs = [dev.alloc_trajectory(100) for _ in range(4)]
dev.rollout_fragments(100, bufs)
print("FSM fragment list: 4 x", tuple(bufs[0].observations.shape), "from", dev.last_kernel().split("+")[0],
      "| shops observe on", int(bufs[0].obs_valid[:, 0, 0].sum()), "of 100 steps, rewards emitted on", int((bufs[0].reward_valid[:, 0, 0] == 1).sum()))

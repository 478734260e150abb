#!/usr/bin/env python
"""The digital-ads market of the reference (examples/environments/digital_ads_market/digital_ads_market.py)
on the MI355X path: dict API for one env, a fused rollout of 4096 envs.  Needs a GPU and the built
library (python -c "import __graft_entry__ as g; g.build()").

    python examples/digital_ads.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import phantom_amd as ph

themes = {"travel": 2, "tech": 2, "sport": 2}

# 1. one env, the reference's dict API: publisher step (nobody strategic acts), then the advertisers bid
st = {f"ADV_{i + 1}": ph.AdvertiserAgent.Supertype(budget=b) for i, b in enumerate([5.0, 6.0, 7.0, 8.0, 10.0, 12.0])}
env = ph.DigitalAdsEnv(num_steps=20, num_agents_theme=themes, agent_supertypes=st, seed=1)
env.reset()
clicks = 0.0
for t in range(env.num_steps):
    acting = env.current_stage == "advertiser_step"
    step = env.step({aid: np.array([0.1], np.float32) for aid in env.strategic_agent_ids} if acting else {})
    clicks += sum(r for r in step.rewards.values() if r is not None)
print("B=1 dict API: clicks", clicks, "budget left", {aid: round(float(env[aid].left), 3) for aid in env.strategic_agent_ids})
print("   ADV_1 sees", step.observations.get("ADV_1"), "wins per user id", env["ADV_1"].total_wins)

# 2. the training configuration in small: budgets from clipped samplers, drawn per env on the device,
#    4096 envs, 10 episodes per launch with the random policy; the trajectory stays on the GPU
st = {f"ADV_{i + 1}": ph.AdvertiserAgent.Supertype(budget=ph.UniformFloatSampler(5.0, 15.001, clip_low=5.0, clip_high=15.0))
      for i in range(6)}
env = ph.DigitalAdsEnv(num_steps=20, num_agents_theme=themes, agent_supertypes=st, batch_size=4096, strategy="second", seed=1)
env.reset()
traj = env.rollout(200)
valid = traj.reward_valid == 1
print("rollout:", tuple(traj.observations.shape), "fused kernel:", env._device().uses_fused,
      "mean clicks per rewarded step %.3f" % float(traj.rewards[valid].mean()),
      "episodes ended", int(traj.truncations[:, :, 0].sum() + 0))

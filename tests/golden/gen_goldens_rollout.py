#!/usr/bin/env python
"""Golden vectors of the reference's ROLLOUT COLLECTION loop, by running the reference's own ``_rollout_task_fn``
(phantom/utils/rllib/rollout.py:261-408) in this container: a batch of env instances stepped in lock-step under fixed
policies, yielding one ``Rollout`` of ``Step`` records per instance.  Recorded: every field of every Step (observations
acted on, actions, rewards, terminations, truncations incl. "__all__", info key sets, stage) and the np.random.randint
draws the customers consumed, so that the device env can replay the episodes and its bulk exits
(phantom_amd.rollout.FragmentBatch: SampleBatch columns, Rollout / Step containers) can be held against them.

Build-container only (imports /root/reference through tests/golden/ref_import.py; ray / gymnasium are inert stand-ins:
every agent gets a fixed ``phantom.Policy``, so no RLlib policy is ever loaded).  Only DATA is written.

    python tests/golden/gen_goldens_rollout.py      # rewrites tests/golden/rollout_task_*.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import ref_import  # noqa: E402

ph = ref_import.import_phantom()
sc = ref_import.import_supply_chain()
dbg = ref_import._stub("ray.rllib.utils.debug")
dbg.update_global_seed_if_necessary = lambda *a, **k: None
from phantom.utils.rllib.rollout import _RolloutConfig, _rollout_task_fn  # noqa: E402
import gen_goldens as gg  # noqa: E402  (DrawRecorder, build_ref_supply_chain)


def table_policy(table, batch):
    """a phantom.Policy whose k-th call returns table[k // batch][k % batch]: _rollout_task_fn calls compute_action once per
    env instance (in order) and step"""
    class TablePolicy(ph.Policy):
        def __init__(self, obs_space, act_space):
            super().__init__(obs_space, act_space)
            self.k = 0

        def compute_action(self, observation):
            i, j = divmod(self.k, batch)
            self.k += 1
            return np.array([table[i][j]], dtype=np.float32)
    return TablePolicy


def run(name, make_env, shop_ids, n_cust, num_steps, batch, seed, fsm=False):
    rng = np.random.RandomState(1000 + seed)
    tables = {s: rng.uniform(0, 100, (num_steps, batch)).astype(np.float32) for s in shop_ids}
    configs = [_RolloutConfig(rollout_id=7 + j, repeat_id=j % 2, env_config={}, rollout_params={"j": j}) for j in range(batch)]
    cfg = types.SimpleNamespace(framework_str="torch", policy_mapping_fn=None)
    np.random.seed(seed)
    with gg.DrawRecorder() as rec:
        rollouts = list(_rollout_task_fn(cfg, None, configs, make_env, None, {s: table_policy(tables[s], batch) for s in shop_ids},
                                         batch, False, {}, False))
    S, T, B = len(shop_ids), num_steps, batch
    assert len(rollouts) == B and all(len(r.steps) == T for r in rollouts)
    D = 3
    out = {"obs": np.zeros((T, B, S, D), np.float32), "obs_key": np.zeros((T, B, S), np.uint8),
           "actions": np.zeros((T, B, S), np.float32), "action_key": np.zeros((T, B, S), np.uint8),
           "rewards": np.zeros((T, B, S), np.float64), "reward_key": np.zeros((T, B, S), np.uint8),
           "terminations": np.zeros((T, B, S), np.uint8), "truncations": np.zeros((T, B, S), np.uint8), "done_key": np.zeros((T, B, S), np.uint8),
           "all_terminated": np.zeros((T, B), np.uint8), "all_truncated": np.zeros((T, B), np.uint8),
           "info_key": np.zeros((T, B, S), np.uint8), "step_i": np.zeros((T, B), np.int32), "stage": np.full((T, B), -1, np.int32),
           "rollout_id": np.array([r.rollout_id for r in rollouts]), "repeat_id": np.array([r.repeat_id for r in rollouts])}
    stage_ids = ["RESTOCK", "SELL"]
    for b, r in enumerate(rollouts):
        for t, st in enumerate(r.steps):
            out["step_i"][t, b] = st.i
            if st.stage is not None:
                out["stage"][t, b] = stage_ids.index(st.stage)
            for s, sid in enumerate(shop_ids):
                if sid in st.observations:
                    out["obs"][t, b, s] = st.observations[sid]; out["obs_key"][t, b, s] = 1
                if sid in st.actions:
                    out["actions"][t, b, s] = st.actions[sid][0]; out["action_key"][t, b, s] = 1
                if sid in st.rewards:
                    out["reward_key"][t, b, s] = 2 if st.rewards[sid] is None else 1
                    out["rewards"][t, b, s] = 0.0 if st.rewards[sid] is None else st.rewards[sid]
                if sid in st.terminations:
                    out["terminations"][t, b, s] = st.terminations[sid]; out["truncations"][t, b, s] = st.truncations[sid]
                    out["done_key"][t, b, s] = 1
                if sid in st.infos:
                    out["info_key"][t, b, s] = 1
            out["all_terminated"][t, b] = st.terminations["__all__"]; out["all_truncated"][t, b] = st.truncations["__all__"]
    # the draws: per step, env instance after env instance (vec_steps = [env.step(..) for env in vec_envs]); FSM envs draw
    # only in the steps whose stage lets the customers act
    draws = np.array(rec.draws, dtype=np.uint8)
    out["draws"] = draws
    out["shape"] = np.array([T, B, S, n_cust, int(fsm)])
    out["table_seed"] = np.array([1000 + seed])
    np.savez_compressed(os.path.join(HERE, f"rollout_task_{name}.npz"), **out)
    print(name, "steps", T, "batch", B, "draws", draws.size, "obs[0,0,0]", out["obs"][0, 0, 0], "last reward", out["rewards"][-1, 0])


def main():
    # (1) the shipped example env: 1 shop, 5 customers, 100 steps; three instances in lock-step
    def shipped(**cfg):
        sc.NUM_CUSTOMERS = 5
        return sc.SupplyChainEnv()
    run("sc7", shipped, ["SHOP"], 5, 100, 3, seed=5)
    # (2) three shops x two customers, 12 steps, four instances
    def multi(**cfg):
        return gg.build_ref_supply_chain(3, [2, 2, 2], 12, 2)[0]
    run("sc3x2", multi, ["SHOP0", "SHOP1", "SHOP2"], 6, 12, 4, seed=9)
    # (3) the FSM variant (RESTOCK -> SELL -> RESTOCK ..): Step.stage = previous_stage, observation / reward dicts omit keys
    def fsm(**cfg):
        return gg.build_ref_supply_chain(2, [3, 3], 10, 3, fsm=True)[0]
    run("sc_fsm", fsm, ["SHOP0", "SHOP1"], 6, 10, 2, seed=3, fsm=True)


if __name__ == "__main__":
    main()

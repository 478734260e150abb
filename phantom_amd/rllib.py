"""Caller-side adapters: what RLlib-facing code of the reference touches on an env
(phantom/utils/rllib/wrapper.py:10-57, train.py:185-187,294-297, rollout.py:300-363).

``ray`` is not a dependency: these classes are duck-typed to the interfaces RLlib calls
(`MultiAgentEnv.step/reset`, `BaseEnv.poll/send_actions/try_reset/get_sub_environments`), and
parity with ray[rllib]==2.7.1 itself is unpinned (ray is absent from the build image).
"""
from collections.abc import Mapping
from typing import Any, Dict, Optional, Tuple

import numpy as np

from .message import AgentID


class RLlibEnvWrapper:
    """wrapper.py:10-57: pass-through ``step`` / ``reset`` + attribute delegation.  With
    ``batch_size == 1`` the wrapped env returns exactly the reference's dict shapes."""

    def __init__(self, env) -> None:
        self.env = env
        self.env.reset()
        self._agent_ids = self.env.strategic_agent_ids
        self.action_space = {aid: env.agents[aid].action_space for aid in self._agent_ids}
        self.observation_space = {aid: env.agents[aid].observation_space for aid in self._agent_ids}

    def get_agent_ids(self):
        return set(self._agent_ids)

    def step(self, action_dict):
        return self.env.step(action_dict)

    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None
              ) -> Tuple[Dict[AgentID, Any], Dict[str, Any]]:
        return self.env.reset(seed, options)

    def is_terminated(self):
        return self.env.is_terminated()

    def __getattr__(self, name: str) -> Any:
        return getattr(self.env, name)

    def __getitem__(self, agent_id: AgentID):
        return self.env.__getitem__(agent_id)

    def __str__(self):
        return f"<{type(self).__name__}{self.env}>"


class _EnvRow(Mapping):
    """{agent_id: value} view of one env instance over batched arrays; nothing is materialised
    until a key is read (4096 x S python dict entries per step would erase the GPU win)."""

    def __init__(self, ids, arrays, valid, b, extra=None, scalar=False):
        self._ids, self._arr, self._valid, self._b = ids, arrays, valid, b
        self._extra = extra or {}
        self._scalar = scalar

    def _keys(self):
        v = self._valid
        return [aid for s, aid in enumerate(self._ids) if v is None or v[self._b, s]]

    def __getitem__(self, key):
        if key in self._extra:
            return self._extra[key]
        s = self._ids.index(key)
        if self._valid is not None and not self._valid[self._b, s]:
            raise KeyError(key)
        v = self._arr[self._b, s]
        return v.item() if self._scalar else v

    def __iter__(self):
        return iter(self._keys() + list(self._extra))

    def __len__(self):
        return len(self._keys()) + len(self._extra)


class _InfoRow(_EnvRow):
    """infos of one env instance: `{}` for every agent that has an observation this step
    (env.py:279-283 with the default collect_infos, agents.py:301-306)."""

    def __init__(self, ids, valid, b):
        super().__init__(ids, None, valid, b)

    def __getitem__(self, key):
        s = self._ids.index(key) if key in self._ids else -1
        if s < 0 or (self._valid is not None and not self._valid[self._b, s]):
            raise KeyError(key)
        return {}


class SubEnvView:
    """``base_env.envs[i]``-style shim (train.py:294-297): env instance ``i`` of the batch as
    seen by metric code -- ``.agents[id].<attr>`` returns that instance's scalar."""

    class _AgentRow:
        def __init__(self, agent, b):
            self._agent, self._b = agent, b

        def __getattr__(self, name):
            v = getattr(self._agent, name)
            return v[self._b].item() if isinstance(v, np.ndarray) and v.ndim >= 1 else v

    def __init__(self, env, b: int):
        self._env, self._b = env, b

    @property
    def agents(self):
        return {aid: SubEnvView._AgentRow(a, self._b) for aid, a in self._env.agents.items()}

    @property
    def current_step(self):
        s = self._env.current_step
        return int(s[self._b]) if isinstance(s, np.ndarray) else s

    def __getitem__(self, agent_id):
        return self.agents[agent_id]

    def __getattr__(self, name):
        return getattr(self._env, name)


class BatchedBaseEnv:
    """BaseEnv-shaped poll / send_actions over one batched device env: the list-of-envs loop of
    rollout.py:361-363 becomes one launch; results are lazy per-env mappings."""

    def __init__(self, env) -> None:
        self.env = env
        self._ids = env.strategic_agent_ids
        self._pending = None
        self._first = True

    @property
    def num_envs(self) -> int:
        return self.env.batch_size

    def get_sub_environments(self):
        return [SubEnvView(self.env, b) for b in range(self.env.batch_size)]

    def try_reset(self, env_id: Optional[int] = None):
        mask = None
        if env_id is not None:
            mask = np.zeros(self.env.batch_size, dtype=np.uint8)
            mask[env_id] = 1
        dev = self.env._device()
        sampler_values, conn_on = self.env._host_draws(mask)       # env.py:211-218 (None: device-drawn)
        obs, valid = dev.reset(mask, sampler_values, conn_on)
        self.env._host_reset(mask)
        o, v = obs.cpu().numpy(), valid.cpu().numpy()
        rows = {b: _EnvRow(self._ids, o, v, b) for b in
                (range(self.env.batch_size) if env_id is None else [env_id])}
        return rows, {b: {} for b in rows}

    def send_actions(self, action_tensor) -> None:
        """actions for every env instance as one f32 [B, S] tensor (device or host)."""
        import torch
        dev = self.env._device()
        a = torch.as_tensor(action_tensor, dtype=torch.float32).to(dev.device).contiguous()
        self._pending = self.env.step_tensors(a)

    def poll(self):
        if self._pending is None:
            rows, infos = self.try_reset()
            B = self.env.batch_size
            empty = {b: {} for b in range(B)}
            return rows, empty, {b: {"__all__": False} for b in range(B)}, \
                {b: {"__all__": False} for b in range(B)}, infos, {}
        self._pending = None
        B = self.env.batch_size
        h = self.env._device().pull_step()                         # one device-to-host copy; copies: the
        h = {k: v.copy() for k, v in h.items()}                    # rows outlive the next step
        obs, ov = h["obs"], h["obs_valid"]
        rew, rv = h["reward"], h["reward_valid"] == 1
        term, trunc = h["terminated"].astype(bool), h["truncated"].astype(bool)
        dv = h["done_valid"]
        at, au = h["all_terminated"].astype(bool), h["all_truncated"].astype(bool)
        obs_d = {b: _EnvRow(self._ids, obs, ov, b) for b in range(B)}
        rew_d = {b: _EnvRow(self._ids, rew, rv, b, scalar=True) for b in range(B)}
        term_d = {b: _EnvRow(self._ids, term, dv, b, {"__all__": bool(at[b])}, scalar=True) for b in range(B)}
        trunc_d = {b: _EnvRow(self._ids, trunc, dv, b, {"__all__": bool(au[b])}, scalar=True) for b in range(B)}
        info_d = {b: _InfoRow(self._ids, ov, b) for b in range(B)}       # infos[aid] = {} (agents.py:301-306)
        return obs_d, rew_d, term_d, trunc_d, info_d, {}

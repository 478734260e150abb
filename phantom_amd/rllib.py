"""Caller-side adapters: what RLlib-facing code of the reference touches on an env
(phantom/utils/rllib/wrapper.py:10-57, train.py:185-187,294-297, rollout.py:300-363).

``ray`` is not a dependency.  When ``ray.rllib`` is importable ``RLlibEnvWrapper`` subclasses its ``MultiAgentEnv`` and
``BatchedBaseEnv`` its ``BaseEnv`` (so ``isinstance`` checks in RLlib's env runners pass); otherwise the same
classes stand on ``object``.  Method names, arities and keyword names follow ray[rllib]==2.7.1 (the reference's pin,
pyproject.toml) -- ``tests/rllib_stub.py`` transcribes them and ``tests/test_rllib_conformance.py`` checks both
classes against that transcript and drives them with an RLlib-style sampling loop.  Parity with ray itself stays
unpinned (ray is absent from the build image).
"""
from collections.abc import Mapping
from typing import Any, Dict, Optional, Tuple

import numpy as np

from .message import AgentID

try:                                        # pragma: no cover - ray is absent from the build image
    from ray.rllib import MultiAgentEnv as _MultiAgentEnvBase
    from ray.rllib.env.base_env import BaseEnv as _BaseEnvBase
except Exception:                           # noqa: BLE001
    _MultiAgentEnvBase = object
    _BaseEnvBase = object


def _space_dict(spaces: Dict):
    """gym.spaces.Dict as in wrapper.py:23-35 when gymnasium is importable, else the plain dict."""
    try:                                    # pragma: no cover - gymnasium is absent from the build image
        import gymnasium as gym
        if all(isinstance(v, gym.Space) for v in spaces.values()):
            return gym.spaces.Dict(spaces)
    except Exception:                       # noqa: BLE001
        pass
    return spaces


class RLlibEnvWrapper(_MultiAgentEnvBase):
    """wrapper.py:10-57: pass-through ``step`` / ``reset`` + attribute delegation.  With
    ``batch_size == 1`` the wrapped env returns exactly the reference's dict shapes."""

    def __init__(self, env) -> None:
        self.env = env
        self.env.reset()
        self._agent_ids = self.env.strategic_agent_ids
        self.action_space = _space_dict({aid: env.agents[aid].action_space for aid in self._agent_ids})
        self.observation_space = _space_dict({aid: env.agents[aid].observation_space for aid in self._agent_ids})
        if _MultiAgentEnvBase is not object:                # wrapper.py:37 -- after the spaces, as the reference does
            super().__init__()
        self._agent_ids = self.env.strategic_agent_ids

    def get_agent_ids(self):
        return set(self._agent_ids)

    def step(self, action_dict):
        return self.env.step(action_dict)

    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None
              ) -> Tuple[Dict[AgentID, Any], Dict[str, Any]]:
        return self.env.reset(seed, options)

    def to_base_env(self, make_env=None, num_envs: int = 1, remote_envs: bool = False,
                    remote_env_batch_wait_ms: int = 0, restart_failed_sub_environments: bool = False):
        """what RLlib's env runners call to vectorise an env (convert_to_base_env): instead of RLlib's python list of
        sub-envs, the wrapped env's own batch IS the vector env (``batch_size`` instances, one launch per step)."""
        if num_envs not in (1, self.env.batch_size):
            import warnings
            warnings.warn(f"to_base_env(num_envs={num_envs}): the device env's batch_size ({self.env.batch_size}) is the number of "
                          "env instances; RLlib's num_envs_per_worker is not used", RuntimeWarning, stacklevel=2)
        return BatchedBaseEnv(self.env)

    def is_terminated(self):
        return self.env.is_terminated()

    def __getattr__(self, name: str) -> Any:
        if name == "env":                                   # not yet set (unpickling / failed __init__): no recursion
            raise AttributeError(name)
        return getattr(self.env, name)

    def __getitem__(self, agent_id: AgentID):
        return self.env.__getitem__(agent_id)

    def __str__(self):
        return f"<{type(self).__name__}{self.env}>"


try:                                        # pragma: no cover
    from ray.rllib.algorithms.callbacks import DefaultCallbacks as _CallbacksBase
except Exception:                           # noqa: BLE001
    _CallbacksBase = object


class RLlibMetricLogger(_CallbacksBase):
    """train.py:283-306: the RLlib callback that logs Phantom metrics -- ``episode.user_data[metric_id]`` collects one value
    per step from ``base_env.envs[0]`` (with the batched env: instance 0 of the batch, a SubEnvView), ``on_episode_end``
    reduces them into ``episode.custom_metrics``."""

    def __init__(self, metrics) -> None:
        if _CallbacksBase is not object:
            super().__init__()
        self.metrics = metrics

    def on_episode_start(self, *, episode, **kwargs) -> None:
        for metric_id in self.metrics.keys():
            episode.user_data[metric_id] = []

    def on_episode_step(self, *, base_env, episode, **kwargs) -> None:
        from .metrics import logging_helper
        env = base_env.envs[0]
        logging_helper(env, self.metrics, episode.user_data)

    def on_episode_end(self, *, episode, **kwargs) -> None:
        for metric_id, metric in self.metrics.items():
            episode.custom_metrics[metric_id] = metric.reduce(episode.user_data[metric_id], mode="train")

    def __call__(self) -> "RLlibMetricLogger":
        return self


def register_env(name: str, env_class, registry=None):
    """train.py:185-187: ``ray.tune.registry.register_env(env_class.__name__, lambda config:
    RLlibEnvWrapper(env_class(**config)))``.  ``registry``: anything with ``register_env(name, creator)``
    (default: ray.tune.registry when importable).  Returns the creator."""
    creator = lambda config: RLlibEnvWrapper(env_class(**config))      # noqa: E731
    if registry is None:
        try:                                # pragma: no cover
            import ray.tune.registry as registry
        except Exception:                   # noqa: BLE001
            registry = None
    if registry is not None:
        registry.register_env(name, creator)
    return creator


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

    @property
    def current_stage(self):
        s = getattr(self._env, "current_stage", None)
        return s[self._b] if isinstance(s, (list, tuple, np.ndarray)) else s

    def __getitem__(self, agent_id):
        return self.agents[agent_id]

    def __getattr__(self, name):
        return getattr(self._env, name)


class _Rows(dict):
    """MultiEnvDict {env_id: row} over batched arrays: ONE object per poll() result.  A consumer that reads rows (an RLlib
    sampler reads every one) gets REAL dicts, all B of them built in one vectorised pass on the first access (``make_all``:
    C-level ``tolist`` / ``zip`` / ``dict``); from then on this IS a plain dict (a dict subclass whose ``__missing__`` fills
    it: every later ``rows[b]`` is a C-level lookup -- round 5's Mapping paid a Python ``__getitem__`` per row).  Envs whose
    dicts omit keys (FSM / Stackelberg validity masks) fall back to per-row views (``make_row``)."""
    __slots__ = ("_n", "_make", "_make_all", "_hold")

    def __init__(self, n: int, make_row, make_all=None, hold=None):
        dict.__init__(self)
        self._n, self._make, self._make_all = n, make_row, make_all
        # what the rows alias (a poll() result's token: its host block is not recycled while this MultiEnvDict is referenced -- the fill
        # function, which also refers to it, is dropped once the rows are built)
        self._hold = hold

    def _fill(self):
        if self._make_all is not None:
            rows, self._make_all = self._make_all(), None
            dict.update(self, zip(range(self._n), rows))
        elif self._make is not None and dict.__len__(self) < self._n:
            for b in range(self._n):
                if not dict.__contains__(self, b):
                    dict.__setitem__(self, b, self._make(b))

    def __missing__(self, b):
        if (self._make_all is None and self._make is None) or not isinstance(b, (int, np.integer)) or not 0 <= b < self._n:
            raise KeyError(b)
        if self._make_all is None:                              # per-row views: made on demand, kept
            row = self._make(int(b))
            dict.__setitem__(self, int(b), row)
            return row
        self._fill()
        return dict.__getitem__(self, int(b))

    def __contains__(self, b):
        return isinstance(b, (int, np.integer)) and 0 <= b < self._n

    def __iter__(self):
        return iter(range(self._n))

    def __len__(self):
        return self._n

    def keys(self):
        return range(self._n)

    def values(self):                                        # (one pass over the materialised rows, no per-key lookups)
        self._fill()
        return [dict.__getitem__(self, b) for b in range(self._n)]

    def items(self):
        return zip(range(self._n), self.values())

    def get(self, b, default=None):
        try:
            return self[b]
        except KeyError:
            return default


def _rows_of_views(ids, views, B):
    """[{agent_id: views[b S + s]} for b] from a flat list of per-(env, agent) arrays: ``zip`` stops after S items of the SHARED
    iterator, so consecutive dicts take consecutive runs -- no slicing, no Python-level loop body beyond the comprehension"""
    it = iter(views)
    return [dict(zip(ids, it)) for _ in range(B)]


def _rows_of_arrays(ids, arr):
    """[{agent_id: arr[b, s]} for b]: one numpy row view per (env, agent), made by iterating the flattened array in C."""
    return _rows_of_views(ids, list(arr.reshape((-1,) + arr.shape[2:])), arr.shape[0])


def _rows_of_scalars(ids, arr, extra_key=None, extra=None):
    """[{agent_id: python scalar} for b] (+ {extra_key: extra[b]}): ``tolist`` converts the whole array at once."""
    rows = _rows_of_views(ids, arr.ravel().tolist(), arr.shape[0])
    if extra_key is not None:
        for d, v in zip(rows, extra.tolist()):
            d[extra_key] = v
    return rows


class _HostBlock:
    """Host arrays a poll() result's rows alias, with the per-(env, agent) observation views made ONCE: creating 36 864 numpy views
    was the largest single cost of reading a step's rows at B = 4096.  A block is recycled only when nothing refers
    to the result that used it (``owner`` is a weak reference to that result's token); otherwise a new block is made -- rows handed
    out never change under their reader."""
    __slots__ = ("arrays", "obs_views", "obs_rows", "owner")

    def __init__(self, like):
        self.arrays = {k: v.copy() for k, v in like.items()}
        o = self.arrays["obs"]
        self.obs_views = list(o.reshape((-1,) + o.shape[2:]))
        self.obs_rows = None
        self.owner = None

    def rows(self, ids, B):
        """The B observation rows {agent_id: view} over this block's views, made ONCE per block as well (4 096 dicts of nine entries are
        0.9 ms of every step's first read): the views do not change when the block is refilled, so neither do the dicts.  They travel with
        the block: treat them as read-only, like the views (a result's rows are valid while its MultiEnvDict is referenced)."""
        r = self.obs_rows
        if r is None or r[0] is not ids:
            r = self.obs_rows = (ids, _rows_of_views(ids, self.obs_views, B))
        return r[1]


class _Token:
    __slots__ = ("__weakref__", "block")


class BatchedBaseEnv(_BaseEnvBase):
    """ray.rllib.env.base_env.BaseEnv over ONE batched device env: the list-of-envs loop of rollout.py:361-363 (and of
    RLlib's vector env) becomes one launch.

    * ``poll()`` -> (obs, rewards, terminateds, truncateds, infos, off_policy_actions), each a MultiEnvDict
      ``{env_id: {agent_id: value}}`` (lazy mappings; ``"__all__"`` in the done dicts);
    * ``send_actions(action_dict)`` takes RLlib's MultiEnvDict ``{env_id: {agent_id: action}}`` and converts it ONCE into
      the f32 [B, S] action tensor + the u8 [B, S] "agent has an action" mask (env.py:330);
    * ``send_action_tensor(actions, action_valid=None)`` is the tensor fast path (no python per env);
    * ``try_reset(env_id=None, *, seed=None, options=None)`` -> (obs MultiEnvDict, infos MultiEnvDict)."""

    def __init__(self, env, keep_results: bool = True) -> None:
        """``keep_results`` (default): a poll() result stays readable however many steps / resets follow, whenever it is first read --
        the contract of RLlib's BaseEnv and of every earlier version of this adapter ("rows outlive the next step").  Its arrays travel
        to the host asynchronously with every poll (no host synchronisation; ~25 us of stream time per step at SC64, B = 4096) and a
        result that is still referenced when its pinned buffer is reused is copied out first.
        ``keep_results=False`` is the zero-copy opt-in for loops that stay on tensors (send_action_tensor + the device's StepTensors
        and poll only for form's sake): a result is brought to the host when one of its rows is first read, which then has to happen
        BEFORE the next send_actions() / send_action_tensor() / try_reset() / StepGraph.replay() (DeviceError otherwise)."""
        self.env = env
        self._keep = bool(keep_results)
        self._ids = env.strategic_agent_ids
        self._col = {aid: s for s, aid in enumerate(self._ids)}
        self._pending = None
        self._first = True
        self._always_full = None
        self._blocks = []                                          # host blocks with pre-made observation views (_HostBlock)
        from operator import itemgetter
        self._pick = itemgetter(*self._ids) if len(self._ids) > 1 else None

    # ---- BaseEnv surface -----------------------------------------------------------------------------------------
    @property
    def num_envs(self) -> int:
        return self.env.batch_size

    @property
    def envs(self):
        """``base_env.envs[0]`` of RLlibMetricLogger.on_episode_step (train.py:294-297)."""
        return self.get_sub_environments()

    @property
    def observation_space(self):
        return _space_dict({aid: self.env.agents[aid].observation_space for aid in self._ids})

    @property
    def action_space(self):
        return _space_dict({aid: self.env.agents[aid].action_space for aid in self._ids})

    def get_agent_ids(self):
        return set(self._ids)

    def get_sub_environments(self, as_dict: bool = False):
        B = self.env.batch_size
        if as_dict:
            return {b: SubEnvView(self.env, b) for b in range(B)}
        return [SubEnvView(self.env, b) for b in range(B)]

    def try_render(self, env_id: Optional[int] = None) -> None:
        return None

    def to_base_env(self, make_env=None, num_envs: int = 1, remote_envs: bool = False,
                    remote_env_batch_wait_ms: int = 0, restart_failed_sub_environments: bool = False):
        return self

    def action_space_sample(self, agent_id: Optional[list] = None):
        ids = self._ids if agent_id is None else [a for a in self._ids if a in agent_id]
        return {b: {aid: self.env.agents[aid].action_space.sample() for aid in ids} for b in range(self.env.batch_size)}

    def observation_space_sample(self, agent_id: Optional[list] = None):
        ids = self._ids if agent_id is None else [a for a in self._ids if a in agent_id]
        return {b: {aid: self.env.agents[aid].observation_space.sample() for aid in ids} for b in range(self.env.batch_size)}

    def observation_space_contains(self, x) -> bool:
        return all(self.env.agents[aid].observation_space.contains(v) for row in x.values() for aid, v in row.items())

    def action_space_contains(self, x) -> bool:
        return all(self.env.agents[aid].action_space.contains(np.asarray(v, dtype=np.float32).reshape(
            self.env.agents[aid].action_space.shape)) for row in x.values() for aid, v in row.items())

    def last(self):
        """the most recent poll() result again (BaseEnv.last); None before the first poll"""
        return getattr(self, "_last", None)

    def stop(self) -> None:
        dev = getattr(self.env, "_dev", None)
        if dev is not None:
            dev.close()
            self.env._dev = None                                   # a later call (try_restart / try_reset after stop) rebuilds the device env

    def sample(self, T: int, actions=None, exo=None, policy_mapping_fn=None):
        """The bulk exit: T steps of every env instance in fused device rollouts and the fragment as per-policy ``SampleBatch``
        column dicts (phantom_amd.rollout.FragmentBatch.to_sample_batches) -- no python object per (env, agent, step).
        ``actions`` f32 [T, B, S] replays a policy's actions (None: the device's random policy)."""
        return self.env.sample(T, actions, exo).to_sample_batches(policy_mapping_fn)

    def try_restart(self, env_id: Optional[int] = None) -> None:
        """RLlib calls this after a sub-env fault; the device env has no per-instance process to restart: reset it."""
        self.try_reset(env_id)

    def try_reset(self, env_id: Optional[int] = None, *, seed: Optional[int] = None,
                  options: Optional[Dict[str, Any]] = None):
        mask = None
        if env_id is not None:
            mask = np.zeros(self.env.batch_size, dtype=np.uint8)
            mask[env_id] = 1
        dev = self.env._device()
        sampler_values, conn_on = self.env._host_draws(mask)       # env.py:211-218 (None: device-drawn)
        obs, valid = dev.reset(mask, sampler_values, conn_on)
        self.env._host_reset(mask)
        o, v = obs.cpu().numpy(), valid.cpu().numpy()
        ids = self._ids
        if env_id is None:
            B = self.env.batch_size
            if bool(v.all()):
                return _Rows(B, None, lambda: _rows_of_arrays(ids, o)), _Rows(B, lambda b: {})
            return _Rows(B, lambda b: _EnvRow(ids, o, v, b)), _Rows(B, lambda b: {})
        return {env_id: _EnvRow(ids, o, v, env_id)}, {env_id: {}}

    def send_actions(self, action_dict) -> None:
        """RLlib's signature: ``action_dict`` is a MultiEnvDict {env_id: {agent_id: action}}; an agent missing from an
        env's dict did not act (``aid in actions``, env.py:330).  Converted once into the [B, S] tensors."""
        if not isinstance(action_dict, Mapping):
            raise TypeError("send_actions takes RLlib's MultiEnvDict {env_id: {agent_id: action}}; "
                            "for a [B, S] action tensor use send_action_tensor()")
        import torch
        B, S = self.env.batch_size, len(self._ids)
        ids = self._ids
        act = valid = None
        if len(action_dict) == B:
            # the common case -- every env instance, in env order, every agent, scalar (or 1-element) actions: ONE flat comprehension over
            # agents; a NaN anywhere -- an agent missing from a row, or a policy that really produced one -- and rows with keys
            # outside the env's strategic agents are left to the entry-by-entry path below, which decides by key membership
            # (``aid in actions``, env.py:330) and raises KeyError for an unknown agent id
            try:
                rows = list(map(action_dict.__getitem__, range(B)))
                if self._pick is not None and sum(map(len, rows)) == B * S:
                    # (itemgetter raises KeyError for a missing agent; equal sizes + every id present = exactly the env's agents)
                    try:                                     # python / numpy scalars: one C-level pass
                        import itertools
                        flat = np.fromiter(itertools.chain.from_iterable(map(self._pick, rows)), np.float32, B * S)
                    except (TypeError, ValueError):          # 1-element arrays (Box(1,) actions as RLlib hands them over)
                        flat = np.array(list(map(self._pick, rows)), dtype=np.float32)
                else:
                    nan = float("nan")
                    flat = np.array([row.get(aid, nan) for row in rows for aid in ids], dtype=np.float32)
                if flat.size == B * S and sum(map(len, rows)) == B * S and not np.isnan(flat).any():
                    act = flat.reshape(B, S)
                    valid = np.ones((B, S), dtype=np.uint8)
            except (KeyError, TypeError, ValueError):
                act = valid = None
        if act is None:
            act = np.zeros((B, S), dtype=np.float32)
            valid = np.zeros((B, S), dtype=np.uint8)
            col = self._col
            for b, row in action_dict.items():
                for aid, a in row.items():
                    s = col[aid]
                    act[b, s] = a if np.isscalar(a) else np.asarray(a, dtype=np.float32).reshape(-1)[0]
                    valid[b, s] = 1
        missing = B - len(action_dict)
        if missing and not getattr(self, "_warned_partial", False):
            # RLlib's BaseEnv steps only the env ids it was given; the device env steps its whole batch in lock-step (one launch):
            # instances missing from ``action_dict`` advance one step with no agent acting
            import warnings
            warnings.warn(f"BatchedBaseEnv.send_actions: {missing} of {B} env instances have no entry in action_dict; the batched "
                          "device env steps ALL instances in lock-step (those advance with no agent acting)", RuntimeWarning, stacklevel=2)
            self._warned_partial = True
        dev = self.env._device()
        self._pending = self.env.step_tensors(torch.from_numpy(act).to(dev.device),
                                              torch.from_numpy(valid).to(dev.device))

    def send_action_tensor(self, action_tensor, action_valid=None) -> None:
        """the fast path: actions of every env instance as one f32 [B, S] tensor (device or host)."""
        import torch
        dev = self.env._device()
        a = action_tensor
        if not (isinstance(a, torch.Tensor) and a.dtype == torch.float32 and a.device == dev.device and a.is_contiguous()):
            a = torch.as_tensor(action_tensor, dtype=torch.float32).to(dev.device).contiguous()
        if action_valid is not None:
            action_valid = torch.as_tensor(action_valid, dtype=torch.uint8).to(dev.device).contiguous()
        self._pending = self.env.step_tensors(a, action_valid)

    def poll(self):
        B = self.env.batch_size
        ids = self._ids
        if self._pending is None:
            rows, infos = self.try_reset()
            self._last = (rows, _Rows(B, lambda b: {}), _Rows(B, lambda b: {"__all__": False}),
                          _Rows(B, lambda b: {"__all__": False}), infos, _Rows(B, lambda b: {}))
            return self._last
        self._pending = None
        dev = self.env._device()
        if self._always_full is None:
            # plain envs of kinds that always observe: every strategic agent is in every dict of every step (env.py:279-297) --
            # known without looking at the validity planes, so nothing has to reach the host before a row is read
            from . import _abi
            self._always_full = self.env.spec.env_type == _abi.ENV_PLAIN and not dev._needs_valid_planes()
        if self._always_full:
            # poll() returns six lazy MultiEnvDicts: no copy, no host synchronisation and no python work per env unless a row is read.
            # The first row read of ANY of them brings the step's outputs to the host (one copy) and builds that result's B dicts in
            # one vectorised pass (the tensor path paid 127 us per step for a synchronous copy, numpy copies and three
            # reductions nobody had asked for).
            host = dev.pull_step_async() if self._keep else dev.pull_step_lazy()
            tok = _Token(); tok.block = None
            blocks = self._blocks

            def arrays():                                        # the step's arrays in a host block of this adapter (first read of any of the six)
                if tok.block is None:
                    blk = next((x for x in blocks if x.owner is None or x.owner() is None), None)
                    if blk is None:
                        blk = _HostBlock(host.get())
                        blocks.append(blk)
                    else:
                        host.read_into(blk.arrays)               # (one copy: pinned buffer -> the block)
                    import weakref
                    blk.owner = weakref.ref(tok)
                    tok.block = blk
                return tok.block.arrays
            self._last = (_Rows(B, None, lambda: (arrays(), tok.block.rows(ids, B))[1], hold=tok),
                          _Rows(B, None, lambda: _rows_of_scalars(ids, arrays()["reward"])),
                          _Rows(B, None, lambda: _rows_of_scalars(ids, arrays()["terminated"].astype(bool), "__all__", arrays()["all_terminated"].astype(bool))),
                          _Rows(B, None, lambda: _rows_of_scalars(ids, arrays()["truncated"].astype(bool), "__all__", arrays()["all_truncated"].astype(bool))),
                          _Rows(B, None, lambda: [{aid: {} for aid in ids} for _ in range(B)]),     # infos[aid] = {} (agents.py:301-306)
                          _Rows(B, lambda b: {}))
            return self._last
        h = dev.pull_step()                                        # one device-to-host copy; copies: the
        h = {k: v.copy() for k, v in h.items()}                    # rows outlive the next step
        obs, ov = h["obs"], h["obs_valid"]
        rew, rv = h["reward"], h["reward_valid"] == 1
        term, trunc = h["terminated"].astype(bool), h["truncated"].astype(bool)
        dv = h["done_valid"]
        at, au = h["all_terminated"].astype(bool), h["all_truncated"].astype(bool)
        full = bool(ov.all()) and bool(rv.all()) and bool(dv.all())       # plain envs: every strategic agent in every dict
        if full:
            self._last = (_Rows(B, None, lambda: _rows_of_arrays(ids, obs)),
                          _Rows(B, None, lambda: _rows_of_scalars(ids, rew)),
                          _Rows(B, None, lambda: _rows_of_scalars(ids, term, "__all__", at)),
                          _Rows(B, None, lambda: _rows_of_scalars(ids, trunc, "__all__", au)),
                          _Rows(B, None, lambda: [{aid: {} for aid in ids} for _ in range(B)]),     # infos[aid] = {} (agents.py:301-306)
                          _Rows(B, lambda b: {}))
            return self._last
        self._last = (_Rows(B, lambda b: _EnvRow(ids, obs, ov, b)),
                _Rows(B, lambda b: _EnvRow(ids, rew, rv, b, scalar=True)),
                _Rows(B, lambda b: _EnvRow(ids, term, dv, b, {"__all__": bool(at[b])}, scalar=True)),
                _Rows(B, lambda b: _EnvRow(ids, trunc, dv, b, {"__all__": bool(au[b])}, scalar=True)),
                _Rows(B, lambda b: _InfoRow(ids, ov, b)),          # infos[aid] = {} (agents.py:301-306)
                _Rows(B, lambda b: {}))
        return self._last

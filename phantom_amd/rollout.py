"""Bulk exits of a device rollout fragment (VERDICT r3 Missing #4 / Next #2).

The reference builds, per env instance and per step, a ``Step`` of five dicts and collects them in a ``Rollout``
(phantom/utils/rllib/rollout.py:300-408, containers phantom/utils/rollout.py); RLlib's samplers build per-(env, agent)
trajectories and concatenate them into ``SampleBatch`` columns.  Here the whole fragment ``[T, B, S, ..]`` is already
on the device: ``FragmentBatch`` brings it to the host ONCE (transposed on the device to ``[B, S, T, ..]`` so that an
agent's steps are contiguous, one pinned copy per column) and offers

* ``to_sample_batches()`` -- per-policy column dicts with RLlib's ``SampleBatch`` column names (``obs, new_obs, actions,
  rewards, terminateds, truncateds, eps_id, agent_index, t``): numpy reshapes of the host arrays, no python object per
  (env, agent, step);
* ``rollouts()`` -- the reference's own containers (``Rollout`` / ``Step`` / ``AgentStep`` with the reference's field
  names and helper methods), built LAZILY from the same arrays: a ``Step`` of dicts exists only while somebody looks at it.

Conventions (rollout.py:361-408): step ``i`` of an episode holds the observations the policies acted on (those returned
by ``reset`` or by step ``i - 1``), the actions taken, and the rewards / terminations / truncations / infos that
``env.step`` returned.  The trajectory planes store an agent's done flag OR-ed with ``"__all__"`` (device.Trajectory);
``"__all__"`` is recovered as the AND over the strategic agents (env.py:297-301).
"""
from collections import Counter
from collections.abc import Sequence
from dataclasses import dataclass
from typing import Any, Callable, Dict, Iterable, List, Mapping, Optional, Tuple

import numpy as np

DEFAULT_POLICY_ID = "default_policy"          # ray.rllib.policy.sample_batch.DEFAULT_POLICY_ID


@dataclass(frozen=True)
class AgentStep:
    """One agent's view of one step (phantom/utils/rollout.py:22-32).  The reference's ``Rollout.steps_for_agent`` passes
    terminations and truncations as two positional values into this seven-field record (a TypeError there); here ``done``
    is their OR."""
    i: int
    observation: Optional[Any]
    reward: Optional[float]
    done: bool
    info: Optional[Dict[str, Any]]
    action: Optional[Any]
    stage: Optional[Any] = None


@dataclass(frozen=True)
class Step:
    """One step of one episode (phantom/utils/rollout.py:35-47)."""
    i: int
    observations: Dict[Any, Any]
    rewards: Dict[Any, float]
    terminations: Dict[Any, bool]
    truncations: Dict[Any, bool]
    infos: Dict[Any, Dict[str, Any]]
    actions: Dict[Any, Any]
    messages: Optional[List[Any]] = None
    stage: Optional[Any] = None


class _Steps(Sequence):
    """``Rollout.steps`` over the arrays of a FragmentBatch: episode rows [t0, t0 + n) of env instance b."""

    def __init__(self, frag: "FragmentBatch", b: int, t0: int, n: int):
        self._f, self._b, self._t0, self._n = frag, b, t0, n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._f.step(self._b, self._t0 + i, i)


class Rollout:
    """One episode of one env instance (phantom/utils/rollout.py:50-57): same fields, same helper methods; ``steps`` is a
    lazy sequence (or any list of ``Step``)."""

    def __init__(self, rollout_id: int, repeat_id: int, env_config: Mapping[str, Any], rollout_params: Dict[str, Any],
                 steps: Sequence, metrics: Dict[str, np.ndarray]):
        self.rollout_id, self.repeat_id = rollout_id, repeat_id
        self.env_config, self.rollout_params = env_config, rollout_params
        self.steps, self.metrics = steps, metrics

    def _column(self, field: str, agent_id, drop_nones: bool, stages, none_values: bool = False) -> list:
        out = []
        for step in self.steps:
            if stages is not None and step.stage not in stages:
                continue
            d = getattr(step, field)
            if agent_id in d and not (none_values and drop_nones and d[agent_id] is None):
                out.append(d[agent_id])
            elif not drop_nones:
                out.append(None)
        return out

    def observations_for_agent(self, agent_id, drop_nones: bool = False, stages: Optional[Iterable] = None) -> list:
        return self._column("observations", agent_id, drop_nones, stages)

    def rewards_for_agent(self, agent_id, drop_nones: bool = False, stages: Optional[Iterable] = None) -> list:
        return self._column("rewards", agent_id, drop_nones, stages, none_values=True)

    def terminations_for_agent(self, agent_id, drop_nones: bool = False, stages: Optional[Iterable] = None) -> list:
        return self._column("terminations", agent_id, drop_nones, stages)

    def truncations_for_agent(self, agent_id, drop_nones: bool = False, stages: Optional[Iterable] = None) -> list:
        return self._column("truncations", agent_id, drop_nones, stages)

    def infos_for_agent(self, agent_id, drop_nones: bool = False, stages: Optional[Iterable] = None) -> list:
        return self._column("infos", agent_id, drop_nones, stages)

    def actions_for_agent(self, agent_id, drop_nones: bool = False, stages: Optional[Iterable] = None) -> list:
        return self._column("actions", agent_id, drop_nones, stages)

    def steps_for_agent(self, agent_id, stages: Optional[Iterable] = None) -> List[AgentStep]:
        out = []
        for step in self.steps:
            if stages is not None and step.stage not in stages:
                continue
            term, trunc = step.terminations.get(agent_id), step.truncations.get(agent_id)
            done = None if term is None and trunc is None else bool(term) or bool(trunc)
            out.append(AgentStep(step.i, step.observations.get(agent_id), step.rewards.get(agent_id), done,
                                 step.infos.get(agent_id), step.actions.get(agent_id), step.stage))
        return out

    @staticmethod
    def _hashable(a):
        return a.tobytes() if isinstance(a, np.ndarray) else a

    def count_actions(self, stages: Optional[Iterable] = None) -> List[Tuple[Any, int]]:
        vals = [a for step in self.steps if stages is None or step.stage in stages for a in step.actions.values()]
        return self._count(vals)

    def count_agent_actions(self, agent_id, stages: Optional[Iterable] = None) -> List[Tuple[Any, int]]:
        return self._count([step.actions.get(agent_id) for step in self.steps if stages is None or step.stage in stages])

    def _count(self, vals) -> List[Tuple[Any, int]]:
        # (the reference counts the action objects themselves; numpy arrays are not hashable, so equal arrays are grouped by value)
        first, cnt = {}, Counter()
        for v in vals:
            k = self._hashable(v)
            first.setdefault(k, v)
            cnt[k] += 1
        return [(first[k], n) for k, n in cnt.most_common()]

    def __getitem__(self, index: int):
        return self.steps[index]

    def __repr__(self):
        return f"Rollout(rollout_id={self.rollout_id}, repeat_id={self.repeat_id}, steps={len(self.steps)})"


class FragmentBatch:
    """A rollout fragment on the host, agent-major: every array is ``[B, S, T, ..]`` (env instance, strategic agent, step).

    ``obs``       f32 [B, S, T, D]  what the policy saw before step t        ``new_obs``   f32 [B, S, T, D]  what step t returned
    ``actions``   f32 [B, S, T]                                             ``rewards``   f32 [B, S, T]
    ``terminateds`` / ``truncateds``  bool [B, S, T]  (agent flag OR ``"__all__"``)
    ``obs_valid`` / ``new_obs_valid`` u8 [B, S, T] or None (FSM / Stackelberg envs: key present in the dict),
    ``reward_valid`` u8 [B, S, T] or None (0 absent / 1 value / 2 present-but-None, fsm.py:378)
    ``t``         i32 [B, T] step inside the episode (0-based)              ``eps_id``  i64 [B, T] episode counter * B + b
    ``stage``     i32 [B, T] or None: the stage the step ran in (``previous_stage`` after it, rollout.py:389-391)
    """

    COLUMNS = ("obs", "new_obs", "actions", "rewards", "terminateds", "truncateds")

    def __init__(self, agent_ids, obs, new_obs, actions, rewards, terminateds, truncateds, t, eps_id,
                 obs_valid=None, new_obs_valid=None, reward_valid=None, stage=None, stage_ids=None,
                 action_shape=(1,), never_finishes_alone: bool = True, done_valid=None):
        self.agent_ids = list(agent_ids)
        self.obs, self.new_obs, self.actions, self.rewards = obs, new_obs, actions, rewards
        self.terminateds, self.truncateds, self.t, self.eps_id = terminateds, truncateds, t, eps_id
        self.obs_valid, self.new_obs_valid, self.reward_valid = obs_valid, new_obs_valid, reward_valid
        self.stage, self.stage_ids = stage, stage_ids
        self.done_valid = done_valid                 # u8 [B, S, T] or None (None: every strategic agent has done flags in every step)
        self.action_shape = tuple(action_shape)
        self.never_finishes_alone = never_finishes_alone
        self.B, self.S, self.T = obs.shape[0], obs.shape[1], obs.shape[2]

    # ---- RLlib-shaped exit -------------------------------------------------------------------------------------------
    def to_sample_batches(self, policy_mapping_fn: Optional[Callable[[Any], str]] = None) -> Dict[str, Dict[str, np.ndarray]]:
        """{policy_id: {column: array}} with ``SampleBatch``'s column names.  Rows are ordered (env instance, agent, step):
        an agent's episode is a contiguous run, as in RLlib's per-agent trajectories.  ``policy_mapping_fn(agent_id)``
        (default: everything under ``"default_policy"``).  Agents of one policy that are consecutive in agent order come
        out as reshaped VIEWS of the host arrays (no copy); envs whose dicts omit keys (FSM / Stackelberg) drop the rows of
        absent observations (boolean mask: a copy)."""
        fn = policy_mapping_fn or (lambda aid: DEFAULT_POLICY_ID)
        groups: Dict[str, List[int]] = {}
        for s, aid in enumerate(self.agent_ids):
            groups.setdefault(fn(aid), []).append(s)
        B, T = self.B, self.T
        out = {}
        for pid, idx in groups.items():
            n = len(idx)
            contiguous = idx == list(range(idx[0], idx[0] + n))
            sel = (lambda a: a[:, idx[0]:idx[0] + n]) if contiguous else (lambda a: a[:, idx])
            cols = {
                "obs": sel(self.obs).reshape(B * n * T, -1),
                "new_obs": sel(self.new_obs).reshape(B * n * T, -1),
                "actions": sel(self.actions).reshape((B * n * T,) + self.action_shape),
                "rewards": sel(self.rewards).reshape(-1),
                "terminateds": sel(self.terminateds).reshape(-1),
                "truncateds": sel(self.truncateds).reshape(-1),
                "eps_id": np.broadcast_to(self.eps_id[:, None, :], (B, n, T)).reshape(-1),
                "agent_index": np.broadcast_to(np.asarray(idx, dtype=np.int32)[None, :, None], (B, n, T)).reshape(-1),
                "t": np.broadcast_to(self.t[:, None, :], (B, n, T)).reshape(-1),
                "env_id": np.broadcast_to(np.arange(B, dtype=np.int32)[:, None, None], (B, n, T)).reshape(-1),
            }
            if self.obs_valid is not None:
                keep = sel(self.obs_valid).reshape(-1).astype(bool)
                cols = {k: v[keep] for k, v in cols.items()}
            out[pid] = cols
        return out

    # ---- the reference's containers ------------------------------------------------------------------------------------
    def episodes(self) -> List[Tuple[int, int]]:
        """(first row, length) of every episode piece of the fragment (all env instances run in lock-step)."""
        t = self.t[0]
        starts = [0] + [k for k in range(1, self.T) if t[k] != t[k - 1] + 1]
        return [(a, (starts[i + 1] if i + 1 < len(starts) else self.T) - a) for i, a in enumerate(starts)]

    def step(self, b: int, row: int, i: Optional[int] = None) -> Step:
        """the reference's ``Step`` for env instance b, fragment row ``row`` (dicts built here, on demand)."""
        ids = self.agent_ids
        ov = self.obs_valid[b, :, row] if self.obs_valid is not None else None
        nv = self.new_obs_valid[b, :, row] if self.new_obs_valid is not None else None
        rv = self.reward_valid[b, :, row] if self.reward_valid is not None else None
        term, trunc = self.terminateds[b, :, row], self.truncateds[b, :, row]
        all_t, all_u = bool(term.all()), bool(trunc.all())
        own = (lambda flag, allf: False if (allf and self.never_finishes_alone) else bool(flag))
        obs = {a: self.obs[b, s, row] for s, a in enumerate(ids) if ov is None or ov[s]}
        acts = {a: np.asarray(self.actions[b, s, row], dtype=np.float32).reshape(self.action_shape)
                for s, a in enumerate(ids) if a in obs}
        rew = {a: (None if rv is not None and rv[s] == 2 else float(self.rewards[b, s, row]))
               for s, a in enumerate(ids) if (rv[s] != 0 if rv is not None else True)}
        # done flags: every strategic agent that was live when the step began (env.py:285-292, fsm.py:333-340: the loop runs over the
        # strategic agents, not over the observing ones); infos: the agents that got an observation (env.py:279-283)
        dv = self.done_valid[b, :, row] if self.done_valid is not None else None
        terms = {ids[s]: own(term[s], all_t) for s in range(len(ids)) if dv is None or dv[s]}
        truncs = {ids[s]: own(trunc[s], all_u) for s in range(len(ids)) if dv is None or dv[s]}
        terms["__all__"], truncs["__all__"] = all_t, all_u
        infos = {ids[s]: {} for s in range(len(ids)) if nv is None or nv[s]}
        st = None
        if self.stage is not None:
            k = int(self.stage[b, row])
            st = self.stage_ids[k] if self.stage_ids is not None else k
        return Step(int(self.t[b, row]) if i is None else i, obs, rew, terms, truncs, infos, acts, None, st)

    def rollouts(self, rollout_ids: Optional[Sequence] = None, repeat_ids: Optional[Sequence] = None,
                 env_configs: Optional[Sequence] = None, rollout_params: Optional[Sequence] = None,
                 metrics: Optional[Sequence] = None) -> List[Rollout]:
        """One ``Rollout`` per (episode piece, env instance), episode-major; the per-instance attributes default to the
        instance index / 0 / {}."""
        out = []
        for t0, n in self.episodes():
            for b in range(self.B):
                out.append(Rollout(rollout_ids[b] if rollout_ids is not None else b,
                                   repeat_ids[b] if repeat_ids is not None else 0,
                                   env_configs[b] if env_configs is not None else {},
                                   rollout_params[b] if rollout_params is not None else {},
                                   _Steps(self, b, t0, n),
                                   metrics[b] if metrics is not None else {}))
        return out


def fragment_from_arrays(agent_ids, first_obs, new_obs, actions, rewards, terminated, truncated, num_steps: int, step0,
                         reset_obs: Optional[Dict[int, np.ndarray]] = None, obs_valid=None, reward_valid=None,
                         first_obs_valid=None, stage=None, stage_ids=None, episode0: int = 0,
                         never_finishes_alone: bool = True) -> FragmentBatch:
    """FragmentBatch from TIME-major host arrays (an oracle rollout, a golden): ``new_obs`` [T, B, S, D], the others
    [T, B, S]; ``first_obs`` [B, S, D] = what the policies saw before row 0; ``step0`` = the env's step counter before
    row 0 (int); ``reset_obs[row]`` [B, S, D] = the observation ``env.reset()`` returned before row ``row`` (needed where an
    episode starts inside the fragment)."""
    T, B, S = actions.shape
    tm = lambda a: np.ascontiguousarray(np.moveaxis(a, 0, 2))            # [T, B, S, ..] -> [B, S, T, ..]
    nob = tm(new_obs)
    obs = np.empty_like(nob)
    obs[:, :, 0] = first_obs
    obs[:, :, 1:] = nob[:, :, :-1]
    t = (int(step0) + np.arange(T)) % num_steps
    ep = episode0 + (int(step0) + np.arange(T)) // num_steps
    nv = tm(obs_valid) if obs_valid is not None else None
    ov = None
    if nv is not None:
        ov = np.empty_like(nv)
        ov[:, :, 0] = first_obs_valid if first_obs_valid is not None else 1
        ov[:, :, 1:] = nv[:, :, :-1]
    for row in range(1, T):
        if t[row] == 0:
            if reset_obs is None or row not in reset_obs:
                raise ValueError(f"an episode starts at fragment row {row}: its reset observation is needed (reset_obs[{row}])")
            obs[:, :, row] = reset_obs[row]
            if ov is not None:
                ov[:, :, row] = 1
    return FragmentBatch(agent_ids, obs, nob, tm(actions), tm(rewards).astype(np.float32), tm(terminated).astype(bool),
                         tm(truncated).astype(bool), np.broadcast_to(t.astype(np.int32), (B, T)).copy(),
                         (ep[None, :] * B + np.arange(B)[:, None]).astype(np.int64), ov, nv,
                         tm(reward_valid) if reward_valid is not None else None,
                         np.ascontiguousarray(stage.T) if stage is not None else None, stage_ids,
                         never_finishes_alone=never_finishes_alone)

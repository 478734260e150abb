"""FiniteStateMachineEnv (mirrors phantom/fsm.py:12-380) with table-driven stages.

Stage masking -- which agents act, which observe (the next stage's acting agents) and which
are rewarded -- is compiled into per-stage tables and applied inside the step kernel with the
reward cache / emit-on-observe semantics of fsm.py:309-380.  Handler-less stages (exactly one next
stage, fsm.py:281-292) run entirely on the device.  Stage *handlers* (fsm.py:294-307) are Python
callbacks.  Declared state-independent they are tabulated per (stage, clock) and never run at step time.
Otherwise the device step is split where the reference calls them (fsm.py:275-307): ``phx_step_begin``
runs the acting phase and ``resolve_network()``, the handler is called on the host -- once per stage for
the whole batch, agent attributes are [B] views of the RESOLVED state -- and the stages it returns travel
to ``phx_step_end`` (phx_step_io.next_stage), which makes the transition and computes observations,
rewards and done flags.
"""
from typing import Any, Callable, Dict, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from .env import PhantomEnv
from .message import AgentID
from .network import Network
from .spec import compile_spec
from .views import FSMEnvView

StageID = Any


class FSMValidationError(Exception):
    """fsm.py:12-16"""


class FSMRuntimeError(Exception):
    """fsm.py:19-23"""


def state_independent(handler_fn):
    """Declare a stage handler (fsm.py:294-307) that decides from the CLOCK and the CURRENT STAGE alone -- it may read
    ``env.current_step`` / ``env.current_stage`` / ``env.num_steps`` and call ``env.resolve_network()``, nothing of the
    agents' state.  Such a handler is tabulated once per (stage, clock value) when the env's spec is compiled
    (phx_spec.stage_tab) and the device takes every transition itself: ``step`` needs no host callback and fused
    ``rollout`` launches work.  Use as a decorator under ``@FSMStage(...)`` or on the function passed as ``handler=``."""
    setattr(handler_fn, "_phx_state_independent", True)
    return handler_fn


class StageRule:
    """One rule of a stage handler that branches on agent STATE after ``resolve_network()`` (fsm.py:294-307), in the form the device
    evaluates (phx_stage_rule, ABI 9): ``value <cmp> threshold -> next_stage`` where ``value`` is the state field ``field`` (a
    ``phx_field`` name: "shop.stock", "seller.revenue", ... -- ``DeviceEnv.field_names()``) of ``agent``, or with ``agent=None`` the SUM
    over all agents of the field's kind.  The first rule of the stage that holds wins; none: ``next_stages[0]``."""

    def __init__(self, field: str, cmp: str, threshold: float, next_stage: StageID, agent: Optional[AgentID] = None) -> None:
        self.field, self.cmp, self.threshold, self.next_stage, self.agent = field, cmp, threshold, next_stage, agent

    def __repr__(self):
        who = "sum" if self.agent is None else repr(self.agent)
        return f"StageRule({who} of {self.field} {self.cmp} {self.threshold} -> {self.next_stage!r})"


def state_rules(rules: Sequence["StageRule"]):
    """Declare the rule form of a stage handler: ``@FSMStage(...)`` / ``@state_rules([...])`` on the handler method, or
    ``FSMStage(handler=state_rules([...])(fn))``.  The Python handler stays the definition: when the device env is created it is called
    on random agent states and has to return what the rules give (FSMValidationError otherwise); at step time the device evaluates the
    rules between the two halves of the step -- ``step`` needs no host callback and ``rollout`` runs in one launch."""
    rules = list(rules)

    def mark(handler_fn):
        setattr(handler_fn, "_phx_state_rules", rules)
        return handler_fn
    return mark


class FSMStage:
    """fsm.py:26-63.  ``handler_state_independent``: see ``state_independent`` (same declaration as a flag).  ``rules``: the handler's
    rule form (``state_rules``)."""

    def __init__(self, stage_id: StageID, acting_agents: Sequence[AgentID],
                 rewarded_agents: Optional[Sequence[AgentID]] = None,
                 next_stages: Optional[Sequence[StageID]] = None,
                 handler: Optional[Callable[[], StageID]] = None,
                 handler_state_independent: bool = False, rules: Optional[Sequence["StageRule"]] = None) -> None:
        self.id = stage_id
        self.acting_agents = acting_agents
        self.rewarded_agents = rewarded_agents
        self.next_stages = next_stages or []
        self.handler = handler
        self.handler_state_independent = bool(handler_state_independent)
        self._rules = list(rules) if rules is not None else None

    def rules(self):
        """the handler's rule form (``state_rules`` / ``rules=``), or None"""
        if self._rules is not None:
            return self._rules
        h = self.handler
        return getattr(h, "_phx_state_rules", None) or getattr(getattr(h, "__func__", None), "_phx_state_rules", None)

    def is_tabulated(self) -> bool:
        h = self.handler
        return h is not None and (self.handler_state_independent or getattr(h, "_phx_state_independent", False)
                                  or getattr(getattr(h, "__func__", None), "_phx_state_independent", False))

    def __call__(self, handler_fn):
        setattr(handler_fn, "_decorator", self)
        self.handler = handler_fn
        return handler_fn


class FiniteStateMachineEnv(PhantomEnv):
    _env_type = _abi.ENV_FSM

    def __init__(self, num_steps: int, network: Network, initial_stage: StageID,
                 env_supertype=None, agent_supertypes=None,
                 stages: Optional[Sequence[FSMStage]] = None, allow_host_handlers: bool = False,
                 **device_kwargs) -> None:
        """``allow_host_handlers``: accepted for round-3 callers (it silenced a warning about host-side handlers seeing the state
        before the step; they now run between phx_step_begin and phx_step_end, where the reference runs them)."""
        super().__init__(num_steps, network, env_supertype, agent_supertypes, **device_kwargs)
        self._initial_stage = initial_stage
        self._stages: Dict[StageID, FSMStage] = {}
        self.previous_stage_idx = np.full(self.batch_size, -1, dtype=np.int64)

        for stage in stages or []:                        # fsm.py:130-132
            if stage.id not in self._stages:
                self._stages[stage.id] = stage
        for attr_name in dir(type(self)):                 # fsm.py:135-145 (decorator registration)
            attr = getattr(type(self), attr_name, None)
            if callable(attr) and hasattr(attr, "_decorator"):
                if attr._decorator.id in self._stages:
                    raise FSMValidationError(
                        f"Found multiple stages with ID '{attr._decorator.id}'")
                self._stages[attr._decorator.id] = attr._decorator
        if len(self._stages) == 0:                        # fsm.py:148-151
            raise FSMValidationError(
                "No registered stages. Please use the 'FSMStage' decorator or the "
                "stage_definitions init parameter")
        if self.initial_stage not in self._stages:        # fsm.py:154-157
            raise FSMValidationError(f"Initial stage '{self.initial_stage}' is not a valid stage")
        for stage in self._stages.values():               # fsm.py:160-165
            for next_stage in stage.next_stages:
                if next_stage not in self._stages:
                    raise FSMValidationError(
                        f"Next stage '{next_stage}' given in stage '{stage.id}' is not a valid stage")
        for stage in self._stages.values():               # fsm.py:168-173
            if len(stage.next_stages) != 1 and stage.handler is None:
                raise FSMValidationError(
                    f"Stage '{stage.id}' without handler must have exactly one next stage "
                    f"(got {len(stage.next_stages)})")
        # Stage HANDLERS (fsm.py:294-307) are Python.  Not declared state-independent, they are called on the host between the two
        # halves of the device step -- after the acting phase and resolve_network() (phx_step_begin), before the transition
        # (phx_step_end) -- exactly where fsm.py:275-307 calls them; `self.resolve_network()` inside a handler marks the
        # resolution that phx_step_begin has already done.  One call per stage serves the whole batch: agent attributes are
        # [B] arrays, the handler returns one stage or a sequence of B stages.  Handlers DECLARED state-independent
        # (``state_independent`` / ``handler_state_independent=True``) are tabulated per (stage, clock value) at spec-compile
        # time instead and run nowhere at step time (phx_spec.stage_tab): only those allow fused rollouts.
        # Handlers that branch on agent state in a form the device can evaluate are declared as RULES (``state_rules``): evaluated
        # inside phx_step / phx_rollout on the resolved state (phx_spec.stage_rules), checked against the Python handler on random
        # states when the device env is created.
        self._rule_handlers = [s for s in self._stages.values() if s.handler is not None and s.rules() and not s.is_tabulated()]
        self._host_handlers = [s for s in self._stages.values() if s.handler is not None and not s.is_tabulated() and not s.rules()]
        if self._rule_handlers and self._host_handlers:
            raise NotImplementedError("an env mixes rule-form and host-called stage handlers: give every state-dependent handler a rule form or none")
        self._tab_handlers = [s for s in self._stages.values() if s.is_tabulated()]
        self._rules_checked = False
        self._stage_dirty = False           # rule handlers: the device chose the stages; the host mirror is refreshed on demand
        self._has_handlers = bool(self._host_handlers)
        self._stage_tab = None
        self._in_handler = False
        self._chosen_next = None
        self._stage_list = list(self._stages.values())
        self._stage_index = {s.id: i for i, s in enumerate(self._stage_list)}
        self._h_stage[:] = self._stage_index[initial_stage]

    def _tabulate_handlers(self):
        """stage_tab[s][t] = index of the stage the handler of stage s returns when the clock reads t (t = 1 .. num_steps:
        the clock is incremented before the handler runs, fsm.py:268); handler-less rows hold next_stages[0].  The
        handler is called with the env's host mirrors set to (stage s, clock t) and ``resolve_network()`` as a no-op."""
        if not self._tab_handlers:
            return None
        if self._host_handlers:
            raise NotImplementedError("an env mixes tabulated (state-independent) and host-called stage handlers: declare "
                                      "every handler state-independent or none")
        ns, T = len(self._stage_list), int(self.num_steps)
        tab = np.zeros((ns, T + 1), dtype=np.int32)
        keep = (self._h_step.copy(), self._h_stage.copy())
        try:
            for si, stage in enumerate(self._stage_list):
                if not stage.is_tabulated():
                    tab[si, :] = self._stage_index[stage.next_stages[0]]
                    continue
                for t in range(1, T + 1):
                    self._h_step[:] = t
                    self._h_stage[:] = si
                    self._in_handler = True
                    try:                                                     # bound method vs decorator form, fsm.py:294-302
                        ret = stage.handler() if hasattr(stage.handler, "__self__") else stage.handler(self)
                    finally:
                        self._in_handler = False
                    if not (isinstance(ret, str) or np.isscalar(ret)):
                        vals = list(ret)
                        if any(v != vals[0] for v in vals):
                            raise FSMRuntimeError(f"state-independent handler of '{stage.id}' returned different stages for "
                                                  f"different env instances at clock {t}")
                        ret = vals[0]
                    if ret not in stage.next_stages:                         # fsm.py:304-307
                        raise FSMRuntimeError(
                            f"FiniteStateMachineEnv attempted invalid transition from '{stage.id}' to {ret}")
                    tab[si, t] = self._stage_index[ret]
                tab[si, 0] = tab[si, 1]
        finally:
            self._h_step[:], self._h_stage[:] = keep
        return tab

    def _compile(self):
        self._stage_tab = self._tabulate_handlers()
        rules = [(s.id, r) for s in self._rule_handlers for r in s.rules()]
        for sid, r in rules:
            if r.next_stage not in self._stages[sid].next_stages:            # fsm.py:304-307
                raise FSMValidationError(f"rule of stage '{sid}' returns {r.next_stage!r}, which is not one of its next_stages")
        return compile_spec(self.network, self.num_steps, self.batch_size, _abi.ENV_FSM, stage_tab=self._stage_tab, stage_rules=rules,
                            stages=self._stage_list, initial_stage=self._initial_stage,
                            seed=self._seed, env_offset=self._env_offset,
                            force_generic=self._force_generic, samplers=self._samplers, variants=self._variants,
                            device_sampling=self._device_sampling, mt19937=self.exogenous == "mt19937")

    @property
    def initial_stage(self) -> StageID:
        return self._initial_stage

    def _device(self):
        dev = super()._device()
        if self._rule_handlers and not self._rules_checked:
            self._rules_checked = True
            self._check_rules_against_handlers(dev)
        return dev

    def _check_rules_against_handlers(self, dev, trials: int = 6):
        """The Python handler is the definition, the rules are what the device runs: on random agent states (every field a rule of the
        stage reads, drawn around the rule's threshold; all B env instances differ) the handler of each rule stage has to return the
        stage the rules give.  The state blob is restored afterwards."""
        import torch
        keep_blob = dev.state.clone()
        keep = (self._h_step.copy(), self._h_stage.copy())
        rng = np.random.default_rng(12345)
        B = self.batch_size
        kr = self.spec.kind_rank()
        try:
            for stage in self._rule_handlers:
                si = self._stage_index[stage.id]
                rules = stage.rules()
                for _ in range(trials):
                    for r in rules:                                         # random values around the threshold, per (env, agent)
                        f = dev.field(r.field)
                        n = f.shape[1]
                        per_agent = float(r.threshold) / (n if r.agent is None else 1)
                        hi = max(2.0 * abs(per_agent), 4.0)
                        vals = rng.uniform(min(0.0, -hi if per_agent < 0 else 0.0), hi, size=(B, n))
                        if f.dtype in (torch.int32,):
                            vals = np.rint(vals)
                        f.copy_(torch.as_tensor(vals).to(f.dtype).to(f.device))
                    self._h_stage[:] = si
                    self._h_step[:] = 1
                    self._in_handler = True
                    try:
                        ret = stage.handler() if hasattr(stage.handler, "__self__") else stage.handler(self)
                    finally:
                        self._in_handler = False
                    rets = [ret] * B if (isinstance(ret, str) or np.isscalar(ret)) else list(ret)
                    want = np.full(B, self._stage_index[stage.next_stages[0]], dtype=np.int64)
                    decided = np.zeros(B, dtype=bool)
                    for r in rules:
                        f = dev.field(r.field).cpu().numpy().astype(np.float64)
                        v = f.sum(axis=1) if r.agent is None else f[:, int(kr[self.spec.index_of(r.agent)])]
                        hit = {"<": v < r.threshold, "<=": v <= r.threshold, ">": v > r.threshold, ">=": v >= r.threshold,
                               "==": v == r.threshold, "!=": v != r.threshold}[r.cmp]
                        sel = hit & ~decided
                        want[sel] = self._stage_index[r.next_stage]
                        decided |= hit
                    got = np.asarray([self._stage_index[x] for x in rets], dtype=np.int64)
                    if len(got) != B or (got != want).any():
                        b = int(np.flatnonzero(got != want)[0]) if len(got) == B else 0
                        raise FSMValidationError(f"the rules declared for the handler of stage '{stage.id}' disagree with the handler: env instance {b} "
                                                 f"-> handler {rets[b]!r}, rules {self._stage_list[int(want[b])].id!r} ({rules})")
        finally:
            dev.state.copy_(keep_blob)
            self._h_step[:], self._h_stage[:] = keep

    def _refresh_stage(self):
        if self._stage_dirty:
            dev = self._device()
            self._h_stage = dev.field("env.stage")[:, 0].cpu().numpy().astype(np.int64)
            self.previous_stage_idx = dev.field("env.prev_stage")[:, 0].cpu().numpy().astype(np.int64)
            self._stage_dirty = False

    @property
    def current_stage(self):
        self._refresh_stage()
        ids = [self._stage_list[i].id for i in self._h_stage]
        return ids[0] if self.batch_size == 1 else ids

    @property
    def previous_stage(self):
        self._refresh_stage()
        ids = [None if i < 0 else self._stage_list[i].id for i in self.previous_stage_idx]
        return ids[0] if self.batch_size == 1 else ids

    def is_fsm_deterministic(self) -> bool:               # fsm.py:185-187
        return all(len(s.next_stages) == 1 for s in self._stages.values())

    def view(self, agent_views=None) -> FSMEnvView:       # fsm.py:189-193
        s = int(self._h_step[0])
        return FSMEnvView(s, s / self.num_steps, self.current_stage)

    def _acting_customers(self, b: int):
        self._refresh_stage()
        st = self._stage_list[int(self._h_stage[b])]
        spec = self.spec
        return [spec.index_of(aid) for aid in st.acting_agents
                if spec.kind[spec.index_of(aid)] == _abi.KIND_CUSTOMER]

    def _acting_customer_groups(self):
        self._refresh_stage()
        stages = np.unique(self._h_stage)
        if len(stages) != 1:
            return None
        return [(slice(None), self._acting_customers(0))]

    def _host_reset(self, mask=None):
        sel = slice(None) if mask is None else np.asarray(mask, dtype=bool)
        self._h_step[sel] = 0
        self._h_stage[sel] = self._stage_index[self._initial_stage]      # fsm.py:217

    def _sync_host_state(self):
        super()._sync_host_state()
        self._stage_dirty = False
        dev = self._device()
        self._h_stage = dev.field("env.stage")[:, 0].cpu().numpy().astype(np.int64)
        self.previous_stage_idx = dev.field("env.prev_stage")[:, 0].cpu().numpy().astype(np.int64)

    def resolve_network(self):
        """fsm.py handlers call this (env.py:175-183); on the device it is part of the step itself."""
        if self._in_handler:
            return None
        raise NotImplementedError("message resolution is part of the device step (phx_step); "
                                  "use env.network.resolve() for host-driven resolves outside a step")

    def _launch_step(self, dev, actions, action_valid, exo):
        if not self._has_handlers:
            return super()._launch_step(dev, actions, action_valid, exo)
        dev.step_begin(actions, action_valid, exo)               # acting phase + resolve_network(), fsm.py:275-280
        return dev.step_end(self._call_handlers()["next_stage"])  # the handlers' stages -> transition, observations, rewards

    def _step_extras(self):
        self._chosen_next = None
        return {}

    def _call_handlers(self):
        """call the current stages' handlers (fsm.py:294-302) on the RESOLVED state and return their stages for the device."""
        self._chosen_next = None
        import torch
        B = self.batch_size
        # what the reference's handler would see: the clock already incremented (fsm.py:268), the stage not yet
        self._h_step += 1
        try:
            nxt = np.asarray([self._stage_index[s.next_stages[0]] if s.next_stages else -1
                              for s in self._stage_list], dtype=np.int64)[self._h_stage]
            for si in np.unique(self._h_stage):
                stage = self._stage_list[int(si)]
                if stage.handler is None:
                    continue
                sel = self._h_stage == si
                self._in_handler = True
                try:                                                     # bound method vs decorator form, fsm.py:294-302
                    ret = stage.handler() if hasattr(stage.handler, "__self__") else stage.handler(self)
                finally:
                    self._in_handler = False
                rets = [ret] * B if (isinstance(ret, str) or np.isscalar(ret) or ret is None) else list(ret)
                if len(rets) != B:
                    raise FSMRuntimeError(f"stage handler of '{stage.id}' returned {len(rets)} stages for {B} env instances")
                for b in np.flatnonzero(sel):
                    if rets[b] not in stage.next_stages:                 # fsm.py:304-307
                        raise FSMRuntimeError(
                            f"FiniteStateMachineEnv attempted invalid transition from '{stage.id}' to {rets[b]}")
                    nxt[b] = self._stage_index[rets[b]]
        finally:
            self._h_step -= 1
        self._chosen_next = nxt
        dev = self._device()
        return {"next_stage": torch.as_tensor(nxt.astype(np.int32), device=dev.device)}

    def rollout(self, *args, **kwargs):
        if self._has_handlers:
            raise NotImplementedError("stage handlers are Python callables evaluated per step on the host: a fused "
                                      "on-device rollout cannot call them (use step / step_tensors), unless they are "
                                      "declared state-independent (@phantom_amd.state_independent) and tabulated")
        return super().rollout(*args, **kwargs)

    def _host_advance(self):
        self._h_step += 1
        nxt = np.asarray([self._stage_index[s.next_stages[0]] if s.next_stages else 0 for s in self._stage_list])
        if self._rule_handlers:                                          # the device chose: read back when somebody asks
            self._stage_dirty = True
            self._chosen_next = None
            return
        self.previous_stage_idx = self._h_stage.copy()                   # fsm.py:355
        if self._chosen_next is not None:
            self._h_stage = self._chosen_next.copy()
        elif self._stage_tab is not None:                                # the device looked the transition up (stage, clock)
            self._h_stage = self._stage_tab[self._h_stage, np.minimum(self._h_step, self.num_steps)].astype(np.int64)
        else:
            self._h_stage = nxt[self._h_stage]
        self._chosen_next = None

"""Env-spec compiler: Network + env options -> the flat tables of ``phx_spec``.

Everything the reference re-derives per step from Python objects -- agent order
(env.py:142-144), the strategic-agent scan (env.py:151-159), neighbour order
(network.py:217-220), edge and payload checks (network.py:246-252) -- is static for a compiled
topology and is flattened once here.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _abi

# state-field prefix -> agent kind (phx_api.hip: layout()): what a StageRule's `field` may name
_RULE_FIELD_KIND = {"shop": _abi.KIND_SHOP, "seller": _abi.KIND_SELLER, "buyer": _abi.KIND_BUYER, "cashbox": _abi.KIND_CASHBOX,
                    "reqresp": _abi.KIND_REQRESP, "mock": _abi.KIND_MOCK_STRAT, "adv": _abi.KIND_ADVERTISER, "pub": _abi.KIND_PUBLISHER}


@dataclass
class EnvSpec:
    agent_ids: List
    kind: np.ndarray            # u8 [A]
    param_i: np.ndarray         # i32 [A, NPI]
    param_f: np.ndarray         # f64 [A, NPF]
    row_ptr: np.ndarray         # i32 [A+1]
    col: np.ndarray             # i32 [nnz]
    batch: int
    num_steps: int
    round_limit: int
    env_type: int
    flags: int
    queue_cap: int
    trace_cap: int
    seed: int = 0
    env_offset: int = 0
    # FSM
    stage_ids: List = field(default_factory=list)
    initial_stage: int = 0
    stage_act_ptr: Optional[np.ndarray] = None
    stage_act_idx: Optional[np.ndarray] = None
    stage_rewarded: Optional[np.ndarray] = None
    stage_rewarded_all: Optional[np.ndarray] = None
    stage_next: Optional[np.ndarray] = None
    stage_allowed: Optional[np.ndarray] = None     # u8 [n_stages][n_stages]: FSMStage.next_stages as a matrix
    stage_tab: Optional[np.ndarray] = None         # i32 [n_stages][num_steps + 1]: tabulated clock / stage handlers (or None)
    stage_rules: Optional[list] = None             # [(stage index, field name, agent column or -1, cmp code, threshold, next stage index)]: phx_stage_rule
    # Stackelberg
    leaders: Optional[np.ndarray] = None
    followers: Optional[np.ndarray] = None
    # Supertypes / Samplers: one column per distinct Sampler of env._samplers (env.py:80-119)
    sampler_kind: Optional[np.ndarray] = None     # i32 [n_samplers]
    sampler_param: Optional[np.ndarray] = None    # f64 [n_samplers, 4]
    type_src: Optional[np.ndarray] = None         # i32 [A]
    # StochasticNetwork: base connections (row_ptr/col then describe the base graph)
    conn_rate: Optional[np.ndarray] = None        # f64 [n_conn]
    col_conn: Optional[np.ndarray] = None         # i32 [nnz]
    # kernel variants (phx_spec.variant_*, ABI 6): {"rollout": name | int, "block": pairs | "whole_envs", "step": name | int}
    variants: Dict = field(default_factory=dict)

    # ---- derived ------------------------------------------------------------------------
    @property
    def n_agents(self) -> int:
        return len(self.agent_ids)

    @property
    def strategic_idx(self) -> np.ndarray:
        return np.flatnonzero(np.isin(self.kind, _abi.STRATEGIC_KINDS)).astype(np.int32)

    @property
    def strategic_ids(self) -> List:
        return [self.agent_ids[i] for i in self.strategic_idx]

    @property
    def n_strategic(self) -> int:
        return int(len(self.strategic_idx))

    @property
    def n_samplers(self) -> int:
        return 0 if self.sampler_kind is None else int(len(self.sampler_kind))

    @property
    def n_conn(self) -> int:
        return 0 if self.conn_rate is None else int(len(self.conn_rate))

    def agent_obs_dim(self, a: int) -> int:
        """observation length of strategic agent index ``a`` (a typed shop appends its type)."""
        d = _abi.OBS_DIM[int(self.kind[a])]
        if self.type_src is not None and int(self.kind[a]) == _abi.KIND_SHOP \
                and int(self.type_src[a]) != _abi.TYPE_NONE:
            d += 1
        return d

    @property
    def obs_dim(self) -> int:
        d = [self.agent_obs_dim(a) for a, k in enumerate(self.kind.tolist()) if k in _abi.OBS_DIM]
        return max(d) if d else 1

    @property
    def n_exo(self) -> int:
        """exogenous draws per step: one per CustomerAgent, 1 + pi1 per PublisherAgent."""
        pubs = self.kind == _abi.KIND_PUBLISHER
        return int((self.kind == _abi.KIND_CUSTOMER).sum() + pubs.sum() + self.param_i[pubs, 1].sum())

    def exo_slot(self) -> np.ndarray:
        """first exo column of each agent (-1: none), agent order (phx_api.hip derive())."""
        out, n = np.full(self.n_agents, -1, dtype=np.int64), 0
        for a in range(self.n_agents):
            k = int(self.kind[a])
            if k == _abi.KIND_CUSTOMER:
                out[a] = n; n += 1
            elif k == _abi.KIND_PUBLISHER:
                out[a] = n; n += 1 + int(self.param_i[a, 1])
        return out

    def kind_rank(self) -> np.ndarray:
        """rank of each agent among the agents of its own kind (state column)."""
        r = np.zeros(self.n_agents, dtype=np.int32)
        cnt: Dict[int, int] = {}
        for a, k in enumerate(self.kind.tolist()):
            r[a] = cnt.get(k, 0)
            cnt[k] = r[a] + 1
        return r

    def index_of(self, agent_id) -> int:
        return self.agent_ids.index(agent_id)

    def to_ctypes(self):
        """(PhxSpec, keepalive) -- keepalive holds the numpy buffers the struct points into."""
        keep = []

        def ptr(arr, dtype):
            if arr is None:
                return None
            a = np.ascontiguousarray(arr, dtype=dtype)
            if a.size == 0:
                a = np.zeros(1, dtype=dtype)
            keep.append(a)
            return a.ctypes.data

        s = _abi.PhxSpec()
        s.abi_version = _abi.ABI_VERSION
        s.n_agents, s.batch, s.num_steps = self.n_agents, self.batch, self.num_steps
        s.round_limit, s.env_type, s.flags = self.round_limit, self.env_type, self.flags
        s.queue_cap, s.trace_cap = self.queue_cap, self.trace_cap
        s.kind = ptr(self.kind, np.uint8)
        s.param_i = ptr(self.param_i, np.int32)
        s.param_f = ptr(self.param_f, np.float64)
        s.row_ptr = ptr(self.row_ptr, np.int32)
        s.col = ptr(self.col, np.int32)
        s.n_stages = len(self.stage_ids)
        s.initial_stage = self.initial_stage
        s.stage_act_ptr = ptr(self.stage_act_ptr, np.int32)
        s.stage_act_idx = ptr(self.stage_act_idx, np.int32)
        s.stage_rewarded = ptr(self.stage_rewarded, np.uint8)
        s.stage_rewarded_all = ptr(self.stage_rewarded_all, np.uint8)
        s.stage_next = ptr(self.stage_next, np.int32)
        s.stage_allowed = ptr(self.stage_allowed, np.uint8) if self.stage_allowed is not None else None
        s.n_leaders = 0 if self.leaders is None else len(self.leaders)
        s.n_followers = 0 if self.followers is None else len(self.followers)
        s.leaders = ptr(self.leaders, np.int32)
        s.followers = ptr(self.followers, np.int32)
        s.seed = self.seed & 0xFFFFFFFFFFFFFFFF
        s.env_offset = self.env_offset
        s.n_samplers = self.n_samplers
        s.sampler_kind = ptr(self.sampler_kind, np.int32) if self.n_samplers else None
        s.sampler_param = ptr(self.sampler_param, np.float64) if self.n_samplers else None
        s.type_src = ptr(self.type_src, np.int32)
        s.n_conn = self.n_conn
        s.conn_rate = ptr(self.conn_rate, np.float64) if self.n_conn else None
        s.col_conn = ptr(self.col_conn, np.int32) if self.n_conn else None
        s.variant_rollout, s.variant_block, s.variant_step = resolve_variants(self.variants)
        s.variant_reserved = 0
        s.stage_tab = ptr(self.stage_tab, np.int32) if self.stage_tab is not None else None
        if self.stage_rules:
            arr = (_abi.PhxStageRule * len(self.stage_rules))()
            for k, (st, field, col, cmp, thr, nxt) in enumerate(self.stage_rules):
                arr[k].stage, arr[k].agent, arr[k].cmp, arr[k].next_stage = int(st), int(col), int(cmp), int(nxt)
                arr[k].threshold, arr[k].field = float(thr), str(field).encode()
            keep.append(arr)
            import ctypes as C
            s.n_stage_rules, s.stage_rules = len(self.stage_rules), C.cast(arr, C.c_void_p)
        return s, keep


VARIANT_ROLLOUT = {"auto": _abi.VR_AUTO, "time_parallel": _abi.VR_TIME_PARALLEL, "lean": _abi.VR_LEAN,
                   "general": _abi.VR_GENERAL, "launch_loop": _abi.VR_LAUNCH_LOOP, "store_waves": _abi.VR_STORE_WAVES}
VARIANT_STEP = {"auto": _abi.VS_AUTO, "fused": _abi.VS_FUSED, "generic": _abi.VS_GENERIC, "wide": _abi.VS_WIDE,
                "generic_dynamic": _abi.VS_GENERIC_DYNAMIC}


def resolve_variants(variants) -> tuple:
    """(variant_rollout, variant_block, variant_step) of phx_spec from the host-side dict; unknown names raise."""
    v = dict(variants or {})
    unknown = set(v) - {"rollout", "block", "step"}
    if unknown:
        raise ValueError(f"unknown kernel-variant keys {sorted(unknown)} (rollout, block, step)")

    def pick(table, x, what):
        if isinstance(x, str):
            if x not in table:
                raise ValueError(f"unknown {what} variant {x!r}; one of {sorted(table)}")
            return table[x]
        return int(x or 0)

    blk = v.get("block", 0)
    blk = _abi.VB_WHOLE_ENVS if blk == "whole_envs" else int(blk or 0)
    return (pick(VARIANT_ROLLOUT, v.get("rollout", 0), "rollout"), blk, pick(VARIANT_STEP, v.get("step", 0), "step"))


def _max_emissions(kind: int, deg: int) -> int:
    """upper bound of messages one acting agent of this kind sends in the acting phase."""
    if kind in (_abi.KIND_SHOP, _abi.KIND_CUSTOMER, _abi.KIND_BUYER, _abi.KIND_PUBLISHER,
                _abi.KIND_ADVERTISER):
        return 1
    if kind == _abi.KIND_ADEXCHANGE:      # not acting, but one round of it emits up to deg + 1 messages
        return deg + 1
    if kind == _abi.KIND_SELLER:
        return deg
    return 0


def compile_spec(network, num_steps: int, batch_size: int = 1, env_type: int = _abi.ENV_PLAIN,
                 stages: Optional[Sequence] = None, initial_stage=None,
                 leaders: Optional[Sequence] = None, followers: Optional[Sequence] = None,
                 seed: int = 0, env_offset: int = 0, force_generic: bool = False,
                 extra_queue: int = 16, samplers: Optional[Sequence] = None,
                 device_sampling: bool = False, variants: Optional[Dict] = None,
                 stage_tab: Optional[np.ndarray] = None, mt19937: bool = False, stage_rules=None) -> EnvSpec:
    from .agents import Agent, StrategicAgent, check_device_executable
    agent_ids = list(network.agents.keys())
    A = len(agent_ids)
    if A == 0:
        raise ValueError("network has no agents")
    if A > 65535:
        raise ValueError("at most 65535 agents per env (message records carry u16 ids)")
    index = {aid: i for i, aid in enumerate(agent_ids)}

    def index_of(aid):
        if aid not in index:
            raise ValueError(f"Agent with ID = '{aid}' does not exist.")
        return index[aid]

    kind = np.zeros(A, dtype=np.uint8)
    param_i = np.zeros((A, _abi.NPI), dtype=np.int32)
    param_f = np.zeros((A, _abi.NPF), dtype=np.float64)
    for a, aid in enumerate(agent_ids):
        agent = network.agents[aid]
        if not isinstance(agent, Agent):
            raise TypeError(f"{agent!r} is not a phantom_amd.Agent")
        check_device_executable(agent)
        k = int(agent.device_kind)
        if isinstance(agent, StrategicAgent) != (k in _abi.STRATEGIC_KINDS):
            raise TypeError(f"agent '{aid}': StrategicAgent-ness and device kind disagree")
        kind[a] = k
        pi, pf = agent.device_params(index_of)
        param_i[a, :len(pi)] = pi
        param_f[a, :len(pf)] = pf
        if hasattr(agent, "check_topology"):
            agent.check_topology(network)

    # CustomerAgent.pi1 = index among the customers of its shop, in agent order (keys the
    # device RNG stream so that results do not depend on how lanes are mapped)
    per_shop: Dict[int, int] = {}
    for a in range(A):
        if kind[a] == _abi.KIND_CUSTOMER:
            shop = int(param_i[a, 0])
            param_i[a, 1] = per_shop.get(shop, 0)
            per_shop[shop] = param_i[a, 1] + 1

    row_ptr = np.zeros(A + 1, dtype=np.int32)
    col: List[int] = []
    col_conn: List[int] = []
    stochastic = hasattr(network, "base_neighbors")           # StochasticNetwork: the base graph
    for a, aid in enumerate(agent_ids):
        if stochastic:
            for n, i in network.base_neighbors(aid):
                col.append(index[n]); col_conn.append(i)
        else:
            col.extend(index[n] for n in network.neighbors(aid))
        row_ptr[a + 1] = len(col)
    col_arr = np.asarray(col, dtype=np.int32)

    resolver = network.resolver
    round_limit = -1 if getattr(resolver, "round_limit", None) is None else int(resolver.round_limit)
    flags = 0
    if network.ignore_connection_errors:
        flags |= _abi.F_IGNORE_CONN_ERRORS
    if not network.enforce_msg_payload_checks:
        flags |= _abi.F_NO_PAYLOAD_CHECKS
    if force_generic:
        flags |= _abi.F_FORCE_GENERIC
    if getattr(resolver, "shuffle_batches", False):                       # resolvers.py:150-151
        if network.ignore_connection_errors:
            raise NotImplementedError("shuffle_batches with ignore_connection_errors is not supported on the device")
        flags |= _abi.F_SHUFFLE_BATCHES

    if mt19937:                                                           # per-env legacy-numpy streams in the blob (ABI 7)
        flags |= _abi.F_MT19937

    deg = np.diff(row_ptr)
    queue_cap = int(sum(_max_emissions(int(kind[a]), int(deg[a])) for a in range(A))) + extra_queue
    trace_cap = 0
    if getattr(resolver, "enable_tracking", False):
        trace_cap = getattr(resolver, "trace_capacity", None) or 8 * queue_cap

    spec = EnvSpec(agent_ids=agent_ids, kind=kind, param_i=param_i, param_f=param_f,
                   row_ptr=row_ptr, col=col_arr, batch=int(batch_size), num_steps=int(num_steps),
                   round_limit=round_limit, env_type=env_type, flags=flags, queue_cap=queue_cap,
                   trace_cap=int(trace_cap), seed=int(seed), env_offset=int(env_offset))
    resolve_variants(variants)                              # validates the names
    spec.variants = dict(variants or {})

    if env_type == _abi.ENV_FSM:
        stage_ids = [s.id for s in stages]
        sidx = {sid: i for i, sid in enumerate(stage_ids)}
        spec.stage_ids = stage_ids
        spec.initial_stage = sidx[initial_stage]
        ptr, idx = [0], []
        rewarded = np.zeros((len(stages), A), dtype=np.uint8)
        rewarded_all = np.zeros(len(stages), dtype=np.uint8)
        nxt = np.zeros(len(stages), dtype=np.int32)
        allowed = np.zeros((len(stages), len(stages)), dtype=np.uint8)
        for i, st in enumerate(stages):
            idx.extend(index_of(aid) for aid in st.acting_agents)
            ptr.append(len(idx))
            if st.rewarded_agents is None:
                rewarded_all[i] = 1
            else:
                for aid in st.rewarded_agents:
                    rewarded[i, index_of(aid)] = 1
            nxt[i] = sidx[st.next_stages[0]]
            for ns_id in st.next_stages:
                allowed[i, sidx[ns_id]] = 1
        spec.stage_act_ptr = np.asarray(ptr, dtype=np.int32)
        spec.stage_act_idx = np.asarray(idx, dtype=np.int32)
        spec.stage_rewarded, spec.stage_rewarded_all, spec.stage_next = rewarded, rewarded_all, nxt
        spec.stage_allowed = allowed
        if stage_tab is not None:
            tab = np.ascontiguousarray(stage_tab, dtype=np.int32)
            if tab.shape != (len(stages), int(num_steps) + 1):
                raise ValueError(f"stage_tab must have shape ({len(stages)}, {int(num_steps) + 1})")
            spec.stage_tab = tab
        if stage_rules:
            # [(stage id, StageRule)] -> phx_stage_rule rows: the agent's column among the agents of its kind, or -1 = the kind's sum
            kr = spec.kind_rank()
            rows = []
            for sid, r in stage_rules:
                if r.cmp not in _abi.CMP:
                    raise ValueError(f"StageRule: unknown comparison {r.cmp!r} (one of {sorted(_abi.CMP)})")
                if r.next_stage not in sidx:
                    raise ValueError(f"StageRule: next stage {r.next_stage!r} is not a stage of the env")
                col = -1
                if r.agent is not None:
                    # the column is the agent's rank AMONG THE AGENTS OF THE FIELD'S KIND: an agent of another kind would silently
                    # address some other agent's value (phx_create only checks the column against the kind's size)
                    a = index_of(r.agent)
                    fk = _RULE_FIELD_KIND.get(str(r.field).split(".")[0])
                    if fk is None or int(spec.kind[a]) != fk:
                        from .fsm import FSMValidationError
                        raise FSMValidationError(f"StageRule({r.field!r}, agent={r.agent!r}): the agent is a "
                                                 f"{_abi.KIND_NAMES.get(int(spec.kind[a]), '?')}, the field belongs to another kind")
                    col = int(kr[a])
                rows.append((sidx[sid], r.field, col, _abi.CMP[r.cmp], float(r.threshold), sidx[r.next_stage]))
            spec.stage_rules = rows
    elif env_type == _abi.ENV_STACKELBERG:
        spec.leaders = np.asarray([index_of(a) for a in leaders], dtype=np.int32)
        spec.followers = np.asarray([index_of(a) for a in followers], dtype=np.int32)

    if stochastic:
        spec.conn_rate = np.asarray([r for _, _, r in network._base_connections], dtype=np.float64)
        spec.col_conn = np.asarray(col_conn, dtype=np.int32)

    # ---- Supertypes: the device-consumed type field of each agent (agents.py:160-175) ----------
    from .samplers import Sampler, UniformFloatSampler
    samplers = list(samplers or [])
    if samplers:
        spec.sampler_kind = np.asarray(
            [_abi.SAMPLER_UNIFORM if device_sampling and isinstance(sm, UniformFloatSampler)
             else _abi.SAMPLER_HOST for sm in samplers], dtype=np.int32)
        spec.sampler_param = np.asarray(
            [sm.device_params() if isinstance(sm, UniformFloatSampler) else (np.nan,) * 4
             for sm in samplers], dtype=np.float64).reshape(len(samplers), 4)
    type_src = np.full(A, _abi.TYPE_NONE, dtype=np.int32)
    for a, aid in enumerate(agent_ids):
        agent = network.agents[aid]
        fname = getattr(agent, "device_type_field", None)
        if fname is None:
            continue
        st = agent.supertype if agent.supertype is not None else agent.Supertype()   # agents.py:167-171
        v = getattr(st, fname)
        if isinstance(v, Sampler):
            col = [j for j, sm in enumerate(samplers) if sm is v]
            if not col:
                raise ValueError(f"agent '{aid}': Sampler of '{fname}' is not managed by the env")
            type_src[a] = col[0]
        else:
            type_src[a] = _abi.TYPE_CONST
            spec.param_f[a, 0] = float(v)
        if int(kind[a]) == _abi.KIND_ADVERTISER:
            # numpy kind of type.budget (NEP 50): np.clip makes a clipped UniformFloatSampler return
            # np.float64 (samplers.py:143-145); np.random.uniform alone returns a python float
            if isinstance(v, UniformFloatSampler):
                strong = v.clip_low is not None or v.clip_high is not None
            else:
                strong = isinstance(v.value if isinstance(v, Sampler) else v, np.floating)
            spec.param_i[a, 2] = int(strong)
    if (type_src != _abi.TYPE_NONE).any():
        spec.type_src = type_src
    return spec

"""Network: agents + adjacency (mirrors phantom/network.py:34-338).

The reference keeps a networkx DiGraph and delivers messages by calling Python handlers.
Here the Network is the host-side *description* (agents in insertion order, directed edges
in insertion order) that ``spec.compile_spec`` flattens into CSR tables; ``send`` / ``resolve``
called from outside a step (as the reference's network tests do) are forwarded to the device
through phx_inject / phx_resolve.
"""
import warnings
from itertools import product
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple, Type

import numpy as np

from .agents import Agent
from .message import AgentID, Message, MsgPayload
from .resolvers import BatchResolver, Resolver


class NetworkError(Exception):
    """network.py:30-31"""


class Network:
    def __init__(self, agents: Optional[Iterable[Agent]] = None,
                 resolver: Optional[Resolver] = None,
                 connections: Optional[Iterable[Tuple[AgentID, AgentID]]] = None,
                 ignore_connection_errors: bool = False,
                 enforce_msg_payload_checks: bool = True) -> None:
        self.agents: Dict[AgentID, Agent] = {}
        self._succ: Dict[AgentID, Dict[AgentID, None]] = {}   # nx.DiGraph adjacency order
        self.resolver = resolver or BatchResolver()            # network.py:69
        self.ignore_connection_errors = ignore_connection_errors
        self.enforce_msg_payload_checks = enforce_msg_payload_checks
        self._has_raised_msg_payload_deprecation_warning = False
        self._pending: List[Message] = []
        self._owner = None          # PhantomEnv that owns the device handle, if any
        self._standalone = None     # private device handle for env-less send/resolve
        self._topology_version = 0
        if agents is not None:
            self.add_agents(agents)
        if connections is not None:
            for c in connections:
                self.add_connection(*c)

    # ---- construction (network.py:79-176) -------------------------------------------------
    @property
    def agent_ids(self):
        return self.agents.keys()

    def add_agent(self, agent: Agent) -> None:
        if agent.id in self.agents:
            raise ValueError(f"Agent with ID = '{agent.id}' already exists.")
        self.agents[agent.id] = agent
        self._succ[agent.id] = {}
        self._topology_version += 1

    def add_agents(self, agents: Iterable[Agent]) -> None:
        for a in agents:
            self.add_agent(a)

    def add_connection(self, u: AgentID, v: AgentID) -> None:
        if u not in self.agents:
            raise ValueError(f"Agent with ID = '{u}' does not exist.")
        if v not in self.agents:
            raise ValueError(f"Agent with ID = '{v}' does not exist.")
        self._succ[u].setdefault(v, None)      # graph.add_edge(u, v); graph.add_edge(v, u)
        self._succ[v].setdefault(u, None)
        self._topology_version += 1

    def add_connections_from(self, ebunch: Iterable[Tuple[AgentID, AgentID]]) -> None:
        for u, v in ebunch:
            self.add_connection(u, v)

    def add_connections_between(self, us: Iterable[AgentID], vs: Iterable[AgentID]) -> None:
        self.add_connections_from(product(us, vs))

    def add_connections_with_adjmat(self, agent_ids: Sequence[AgentID],
                                    adjacency_matrix: np.ndarray) -> None:
        """network.py:140-177: connect agent_ids[i] -- agent_ids[j] wherever the (square, symmetric, hollow) matrix is positive.
        The matrix is validated as a whole first (the reference's messages), then every row's neighbours are connected in
        column order -- the insertion order the device's CSR adjacency keeps."""
        m = np.asarray(adjacency_matrix)
        n = len(agent_ids)
        problems = ((m.shape[0] != n, "Number of agent IDs doesn't match adjacency matrix dimensions."),
                    (m.ndim != 2 or m.shape[0] != m.shape[-1], "Adjacency matrix must be square."),
                    (m.ndim == 2 and m.shape[0] == m.shape[-1] and not np.array_equal(m, m.T), "Adjacency matrix must be symmetric."),
                    (m.ndim == 2 and bool((np.abs(np.diagonal(m)) >= 1e-5).any()), "Adjacency matrix must be hollow."))
        for bad, message in problems:
            if bad:
                raise ValueError(message)
        for i, j in zip(*np.nonzero(m > 0)):                   # row-major: row i's neighbours in column order
            self.add_connection(agent_ids[int(i)], agent_ids[int(j)])

    def subnet_for(self, agent_id: AgentID) -> "Network":
        """network.py:186-206: the first-order ego network of ``agent_id`` -- the agent, its successors and predecessors
        (the same set: connections are bidirectional), the connections among them, a reset copy of the resolver.  Host-side
        object; it compiles to its own device env when stepped."""
        import copy
        if agent_id not in self.agents:
            raise KeyError(agent_id)
        keep = [agent_id] + [v for v in self._succ[agent_id] if v != agent_id]
        keep += [u for u in self._succ if agent_id in self._succ[u] and u not in keep]
        inside = set(keep)
        sub = Network.__new__(Network)
        Network.__init__(sub, resolver=copy.deepcopy(self.resolver), ignore_connection_errors=self.ignore_connection_errors,
                         enforce_msg_payload_checks=self.enforce_msg_payload_checks)
        sub.resolver.reset()
        for aid in self.agents:                                # graph.subgraph keeps the parent's node order
            if aid in inside:
                sub.agents[aid] = self.agents[aid]
                sub._succ[aid] = {v: None for v in self._succ[aid] if v in inside}
        return sub

    def context_for(self, agent_id: AgentID, env_view) -> "Context":
        """network.py:208-222: the agent, the view each neighbour publishes to it (graph.neighbors order), the env view."""
        from .views import Context
        views = {nid: self.agents[nid].view(agent_id) for nid in self.neighbors(agent_id)}
        return Context(self.agents[agent_id], views, env_view)

    def neighbors(self, agent_id: AgentID) -> List[AgentID]:
        """graph.neighbors(agent_id) order (network.py:219)."""
        return list(self._succ[agent_id].keys())

    def has_edge(self, sender_id: AgentID, receiver_id: AgentID) -> bool:   # network.py:224-231
        return sender_id in self._succ and receiver_id in self._succ[sender_id]

    # ---- queries (network.py:267-295) -----------------------------------------------------
    def get_agents_where(self, pred: Callable[[Agent], bool]) -> Dict[AgentID, Agent]:
        return {aid: a for aid, a in self.agents.items() if pred(a)}

    def get_agents_with_type(self, agent_type: Type) -> Dict[AgentID, Agent]:
        return self.get_agents_where(lambda a: isinstance(a, agent_type))

    def get_agents_without_type(self, agent_type: Type) -> Dict[AgentID, Agent]:
        return self.get_agents_where(lambda a: not isinstance(a, agent_type))

    def __getitem__(self, agent_id: AgentID) -> Agent:
        return self.agents[agent_id]

    def __len__(self) -> int:
        return len(self.agents)

    # ---- messaging from outside a step ------------------------------------------------------
    def _device(self):
        if self._owner is not None:
            return self._owner._device()
        if self._standalone is None or self._standalone.topology_version != self._topology_version:
            from .device import DeviceEnv
            from .spec import compile_spec
            self._standalone = DeviceEnv(compile_spec(self, num_steps=0, batch_size=1))
            self._standalone.topology_version = self._topology_version
            for a in self.agents.values():
                a._env = self._standalone
        return self._standalone

    def reset(self) -> None:
        """network.py:179-184: clears the queues and resets every agent."""
        self._pending.clear()
        self.resolver.reset()
        self._device().reset_agents()

    def send(self, sender_id: AgentID, receiver_id: AgentID, payload) -> None:
        """network.py:233-254 for sends issued by host code (not by device handlers)."""
        if not self.ignore_connection_errors and not self.has_edge(sender_id, receiver_id):
            raise NetworkError(f"No connection between {sender_id} and {receiver_id}.")
        if self.enforce_msg_payload_checks:
            self._enforce_payload_checks(sender_id, receiver_id, payload)
        self._pending.append(Message(sender_id, receiver_id, payload))

    def resolve(self, contexts=None) -> None:
        """network.py:256-265; ``contexts`` is accepted for signature parity (the env passes the
        live-agent set through the device state instead)."""
        dev = self._device()
        pending, self._pending = self._pending, []
        dev.resolve(pending, self)

    def _enforce_payload_checks(self, sender_id, receiver_id, payload) -> None:   # network.py:297-331
        if not hasattr(payload, "_sender_types") or not hasattr(payload, "_receiver_types"):
            if isinstance(payload, MsgPayload):
                if not self._has_raised_msg_payload_deprecation_warning:
                    warnings.warn("MsgPayload type is deprecated. In future, use the "
                                  "@msg_payload decorator", DeprecationWarning)
                    self._has_raised_msg_payload_deprecation_warning = True
                return
            raise NetworkError("Message payloads sent across the network must use the "
                               f"'msg_payload' decorator (bad payload = '{payload}')")
        sender, receiver = self.agents[sender_id], self.agents[receiver_id]
        if (payload._sender_types is not None
                and sender.__class__.__name__ not in payload._sender_types):
            raise NetworkError(
                f"Message payload of type '{payload.__class__.__name__}' cannot be sent by agent "
                f"with type '{sender.__class__.__name__:}' (expected one of {payload._sender_types})")
        if (payload._receiver_types is not None
                and receiver.__class__.__name__ not in payload._receiver_types):
            raise NetworkError(
                f"Message payload of type '{payload.__class__.__name__}' cannot be received by "
                f"agent with type '{receiver.__class__.__name__:}' (expected one of "
                f"{payload._receiver_types})")


class StochasticNetwork(Network):
    """network.py:340-453: every connection carries a connectivity ``rate``; the graph is resampled
    at each reset (``resample_connectivity``).  The host object keeps the reference's behaviour for
    its own ``graph`` (one numpy draw per base connection, in order, at ``add_connection`` and at
    every ``resample_connectivity``); the device keeps one on/off byte per base connection and env
    instance (state field ``net.conn_on``), fed by the host's draws or drawn by the device RNG."""

    def __init__(self, agents=None, resolver=None, connections=None,
                 ignore_connection_errors: bool = False, enforce_msg_payload_checks: bool = True) -> None:
        self._base_connections: List[Tuple[AgentID, AgentID, float]] = []
        super().__init__(agents, resolver, connections, ignore_connection_errors,
                         enforce_msg_payload_checks)

    def add_connection(self, u: AgentID, v: AgentID, rate: float = 1.0) -> None:
        if u not in self.agents:
            raise ValueError(f"Agent with ID = '{u}' does not exist.")
        if v not in self.agents:
            raise ValueError(f"Agent with ID = '{v}' does not exist.")
        if any({u, v} == {a, b} for a, b, _ in self._base_connections):
            raise NotImplementedError(f"connection ({u}, {v}) is defined twice")
        if np.random.random() < rate:                       # network.py:389-391
            self._succ[u].setdefault(v, None)
            self._succ[v].setdefault(u, None)
        self._base_connections.append((u, v, float(rate)))
        self._topology_version += 1

    def add_connections_from(self, ebunch) -> None:         # network.py:395-423
        for connection in ebunch:
            if len(connection) == 2:
                self.add_connection(connection[0], connection[1])
            elif len(connection) == 3:
                self.add_connection(connection[0], connection[1], connection[2])
            else:
                raise ValueError(f"Ill-formatted connection tuple {connection}.")

    def add_connections_between(self, us, vs, rate: float = 1.0) -> None:   # network.py:425-436
        for u, v in product(us, vs):
            self.add_connection(u, v, rate)

    def base_neighbors(self, agent_id: AgentID) -> List[Tuple[AgentID, int]]:
        """(neighbour, base connection index) in base-connection order: the adjacency order of ANY
        resampled graph is a subsequence of it (network.py:441-447 re-adds edges in that order)."""
        out = []
        for i, (u, v, _) in enumerate(self._base_connections):
            if u == agent_id:
                out.append((v, i))
            elif v == agent_id:
                out.append((u, i))
        return out

    def draw_connectivity(self) -> np.ndarray:
        """one `np.random.random() < rate` per base connection, in order (network.py:444-445)."""
        return np.asarray([np.random.random() < r for _, _, r in self._base_connections], dtype=np.uint8)

    def _apply_connectivity(self, on: np.ndarray) -> None:
        self._succ = {aid: {} for aid in self.agents}
        for (u, v, _), keep in zip(self._base_connections, on):
            if keep:
                self._succ[u].setdefault(v, None)
                self._succ[v].setdefault(u, None)

    def resample_connectivity(self) -> None:                # network.py:438-447
        self._apply_connectivity(self.draw_connectivity())

    def reset(self) -> None:                                # network.py:449-452
        self.resample_connectivity()
        Network.reset(self)

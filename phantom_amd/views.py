"""View dataclasses (mirrors phantom/views.py:7-34, phantom/fsm.py:66-73).

On the device the env view is two scalars per env instance (current_step and
current_step / num_steps); these classes exist for name/shape parity of host code.
"""
from dataclasses import dataclass
from typing import Any, Dict, Hashable, List, Mapping, Optional


@dataclass(frozen=True)
class View:
    """views.py:7-17"""


@dataclass(frozen=True)
class AgentView(View):
    """views.py:20-24"""


@dataclass(frozen=True)
class EnvView(View):
    """views.py:27-34"""
    current_step: int
    proportion_time_elapsed: float


@dataclass(frozen=True)
class FSMEnvView(EnvView):
    """fsm.py:66-73"""
    stage: Hashable


@dataclass(frozen=True)
class Context:
    """context.py:11-40: an agent's local neighbourhood -- the agent, the views its neighbours publish to it (None for
    kinds that publish nothing: every supply-chain agent) and the env view.  On the device no such object exists per step
    (static CSR adjacency + two scalars per env instance); this class serves host code that asks a Network for it
    (``Network.context_for``)."""
    agent: Any
    agent_views: Mapping[Any, Optional[AgentView]]
    env_view: EnvView

    @property
    def neighbour_ids(self) -> List[Any]:
        return list(self.agent_views.keys())

    def __getitem__(self, view_id):
        return self.agent_views[view_id]

    def __contains__(self, view_id) -> bool:
        return view_id in self.agent_views

"""View dataclasses (mirrors phantom/views.py:7-34, phantom/fsm.py:66-73).

On the device the env view is two scalars per env instance (current_step and
current_step / num_steps); these classes exist for name/shape parity of host code.
"""
from dataclasses import dataclass
from typing import Hashable


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

"""StackelbergEnv (mirrors phantom/stackelberg.py:13-196): leaders act on odd steps and
followers on even steps; the acting group is rewarded (cached), the other group observes."""
from typing import Sequence

from . import _abi
from .env import PhantomEnv
from .message import AgentID
from .network import Network
from .spec import compile_spec


class StackelbergEnv(PhantomEnv):
    _env_type = _abi.ENV_STACKELBERG

    def __init__(self, num_steps: int, network: Network, leader_agents: Sequence[AgentID],
                 follower_agents: Sequence[AgentID], env_supertype=None, agent_supertypes=None,
                 **device_kwargs) -> None:
        super().__init__(num_steps, network, env_supertype, agent_supertypes, **device_kwargs)
        for aid in list(leader_agents) + list(follower_agents):          # stackelberg.py:40-41
            assert aid in network.agent_ids, f"Agent '{aid}' not in network"
        for aid in leader_agents:                                        # stackelberg.py:43-44
            assert aid not in follower_agents, f"Agent '{aid}' not in network"
        self.leader_agents = list(leader_agents)
        self.follower_agents = list(follower_agents)

    def _compile(self):
        return compile_spec(self.network, self.num_steps, self.batch_size, _abi.ENV_STACKELBERG,
                            leaders=self.leader_agents, followers=self.follower_agents,
                            seed=self._seed, env_offset=self._env_offset,
                            force_generic=self._force_generic, samplers=self._samplers, variants=self._variants,
                            device_sampling=self._device_sampling, mt19937=self.exogenous == "mt19937")

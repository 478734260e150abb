"""Agent classes (mirrors phantom/agents.py:34-349 for the device-executable closed set).

A reference ``Agent`` is a Python object whose handlers run in the interpreter.  Here an agent
object is a *description*: ``device_kind`` selects hand-written device handlers and
``device_params`` supplies the per-agent constants; the mutable attributes of the reference
classes (``ShopAgent.stock`` ...) live in device memory, one value per env instance, and are
read back lazily through attribute access (so ``env["SHOP"].stock`` and
``SimpleAgentMetric("SHOP", "stock")``-style reflection keep working, metrics.py:230-231).
"""
from dataclasses import dataclass
import numpy as np
from typing import Dict, Optional, Tuple

from . import _abi
from .supertype import Supertype as _Supertype
from .message import AgentID
from .views import AgentView


class Agent:
    """phantom/agents.py:34-179.  Default kind: an agent with no message handlers."""

    device_kind = _abi.KIND_MOCK_AGENT
    #: python attribute name -> device state field
    state_fields: Dict[str, str] = {}
    #: name of the Supertype field the device handlers of this kind consume (None: host-only type)
    device_type_field: Optional[str] = None

    def __init__(self, agent_id: AgentID, supertype=None) -> None:
        self._id = agent_id
        self.supertype = supertype
        self._env = None          # bound by PhantomEnv / Network device binding

    @property
    def type(self):
        """agents.py:160-175: the supertype with every Sampler replaced by the value drawn at the
        last reset (per env instance: a python value for batch_size 1, an array [B] otherwise)."""
        st = self.supertype
        if st is None:
            if not hasattr(self, "Supertype"):
                raise AttributeError("type")
            st = self.Supertype()
        env = self.__dict__.get("_env")
        return st.sample() if env is None else env._resolve_type(st)

    @property
    def id(self) -> AgentID:
        return self._id

    def view(self, neighbour_id: Optional[AgentID] = None) -> Optional[AgentView]:
        return None               # agents.py:86-88

    def device_params(self, index_of) -> Tuple[Tuple[int, ...], Tuple[float, ...]]:
        """(int params, float params) compiled into phx_spec.param_i / param_f."""
        return (), ()

    def reset(self) -> None:      # agents.py:160-175: `type` is resolved lazily, see above
        return None

    def __getattr__(self, name):
        fields = type(self).state_fields
        if name in fields:
            env = self.__dict__.get("_env")
            if env is None:
                raise AttributeError(f"{name}: agent '{self._id}' is not bound to an env")
            return env._read_agent_state(self, fields[name])
        raise AttributeError(name)

    def __repr__(self) -> str:
        return f"[{self.__class__.__name__} {self.id}]"


class StrategicAgent(Agent):
    """phantom/agents.py:182-338.  Encoders/decoders/reward functions are arbitrary Python and
    cannot run on the device; device kinds hard-wire encode_observation / decode_action /
    compute_reward instead (as the supply-chain ShopAgent does by overriding them)."""

    device_kind = _abi.KIND_MOCK_STRAT

    def __init__(self, agent_id: AgentID, observation_encoder=None, action_decoder=None,
                 reward_function=None, supertype=None) -> None:
        if observation_encoder or action_decoder or reward_function:
            raise NotImplementedError("Encoder/Decoder/RewardFunction objects are Python "
                                      "callables; use a device agent kind")
        super().__init__(agent_id, supertype)
        self.observation_encoder = None
        self.action_decoder = None
        self.reward_function = None
        self.action_space = None
        self.observation_space = None


def msg_handler(message_type):
    """agents.py:344-349.  The decorator itself only tags the function, as the reference's does; a
    class that CARRIES such a method has Python behaviour the device cannot run, and
    ``compile_spec`` rejects it (see ``check_device_executable``)."""
    def decorator(fn):
        setattr(fn, "_message_type", message_type)
        return fn
    return decorator


class UnsupportedAgentBehaviour(TypeError, NotImplementedError):
    """A user agent class defines Python behaviour (a handler, an observation encoder, ...) that
    the reference would call (agents.py:69-79,96-155) and the device path cannot."""


#: methods of the reference's Agent / StrategicAgent whose bodies ARE the agent's behaviour on the
#: step path (agents.py:96-155,221-338); device kinds hard-wire them in HIP.
BEHAVIOUR_METHODS = ("handle_batch", "handle_message", "generate_messages", "encode_observation",
                     "decode_action", "compute_reward", "is_terminated", "is_truncated",
                     "pre_message_resolution", "post_message_resolution", "collect_infos")


def _is_builtin_agent_class(cls) -> bool:
    return cls.__module__ == __name__ or cls.__module__.startswith(__name__.rsplit(".", 1)[0] + ".")


def check_device_executable(agent) -> None:
    """Raise ``UnsupportedAgentBehaviour`` when a user-defined subclass adds behaviour on top of the
    device kind it inherits: a ``@msg_handler`` method or an override of one of
    ``BEHAVIOUR_METHODS``.  The reference would run that Python (agents.py:69-79 collects the
    decorated handlers, env.py:273-292 calls the overrides); silently running the parent kind's
    device handlers instead would return wrong observations and rewards without any error."""
    for cls in type(agent).__mro__:
        if cls is object or _is_builtin_agent_class(cls):
            break                      # everything from here up has hand-written device handlers
        for name, attr in vars(cls).items():
            fn = getattr(attr, "__func__", attr)
            if name in BEHAVIOUR_METHODS and callable(fn):
                raise UnsupportedAgentBehaviour(
                    f"agent '{agent.id}': {cls.__name__}.{name}() is Python behaviour; the device path "
                    f"runs only the hand-written handlers of kind "
                    f"'{_abi.KIND_NAMES.get(int(agent.device_kind), agent.device_kind)}' and would ignore it. "
                    "Use one of the device agent kinds unchanged, or add a kind (DESIGN.md \u00a71).")
            if hasattr(fn, "_message_type"):
                raise UnsupportedAgentBehaviour(
                    f"agent '{agent.id}': {cls.__name__}.{name} is a @msg_handler; Python message "
                    "handlers cannot run on the device (the closed set of kinds is listed in "
                    "include/phantom_amd.h, phx_kind).")


class Box:
    """Minimal stand-in for gym.spaces.Box (gymnasium is not a dependency of the hot path)."""

    def __init__(self, low, high, shape, dtype="float32"):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def __eq__(self, o):
        return isinstance(o, Box) and (self.low, self.high, self.shape) == (o.low, o.high, o.shape)

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape})"

    def sample(self):
        """uniform over [low, high) where both are finite, else a standard normal shifted to the finite bound
        (gym.spaces.Box.sample's rule, with numpy's global stream)."""
        lo, hi = float(self.low), float(self.high)
        if np.isfinite(lo) and np.isfinite(hi):
            return np.random.uniform(lo, hi, self.shape).astype(self.dtype)
        base = lo if np.isfinite(lo) else (hi if np.isfinite(hi) else 0.0)
        x = np.abs(np.random.normal(size=self.shape)) if np.isfinite(lo) else np.random.normal(size=self.shape)
        return (base + (x if np.isfinite(lo) or not np.isfinite(hi) else -np.abs(x))).astype(self.dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())


# --------------------------------------------------------------------------------------
# supply chain (examples/environments/supply_chain/supply_chain.py)
# --------------------------------------------------------------------------------------
CUSTOMER_MAX_ORDER_SIZE = 5      # supply_chain.py:12
SHOP_MAX_STOCK = 100             # supply_chain.py:13


class FactoryAgent(Agent):
    """supply_chain.py:36-45"""
    device_kind = _abi.KIND_FACTORY

    def __init__(self, agent_id: str):
        super().__init__(agent_id)


class CustomerAgent(Agent):
    """supply_chain.py:48-67"""
    device_kind = _abi.KIND_CUSTOMER

    def __init__(self, agent_id: AgentID, shop_id: AgentID):
        super().__init__(agent_id)
        self.shop_id = shop_id

    def device_params(self, index_of):
        return (index_of(self.shop_id),), ()   # pi1 (index among the shop's customers): spec.py


class ShopAgent(StrategicAgent):
    """supply_chain.py:70-150.  ``num_customers`` replaces the module global NUM_CUSTOMERS
    read at call time by encode_observation (supply_chain.py:125)."""
    device_kind = _abi.KIND_SHOP
    state_fields = {"stock": "shop.stock", "sales": "shop.sales",
                    "missed_sales": "shop.missed_sales", "delivered_stock": "shop.delivered_stock"}

    def __init__(self, agent_id: str, factory_id: str, num_customers: int = 5):
        super().__init__(agent_id)
        self.factory_id = factory_id
        self.num_customers = num_customers
        self.observation_space = Box(0.0, 1.0, (3,))
        self.action_space = Box(0.0, SHOP_MAX_STOCK, (1,))

    def device_params(self, index_of):
        return (index_of(self.factory_id), self.num_customers * CUSTOMER_MAX_ORDER_SIZE), ()


MAX_EXCESS_STOCK_WEIGHT = 0.2    # docs/user/tutorial2.rst:246


class TypedShopAgent(ShopAgent):
    """Tutorial 2's ShopAgent (docs/user/tutorial2.rst:244-307): a Supertype with one field,
    ``excess_stock_weight``; reward = sales - type.excess_stock_weight * stock and the
    observation gains type.excess_stock_weight / MAX_EXCESS_STOCK_WEIGHT as a 4th entry."""
    device_type_field = "excess_stock_weight"

    @dataclass
    class Supertype(_Supertype):
        excess_stock_weight: float = 0.1

    def __init__(self, agent_id: str, factory_id: str, num_customers: int = 5, supertype=None):
        super().__init__(agent_id, factory_id, num_customers)
        self.supertype = supertype
        self.observation_space = Box(0.0, 1.0, (4,))

    def device_params(self, index_of):
        pi, _ = super().device_params(index_of)
        return pi, (0.1, MAX_EXCESS_STOCK_WEIGHT)     # pf0 (constant weight) is set by compile_spec


# --------------------------------------------------------------------------------------
# Stackelberg market (build-authored agents; golden vectors come from running the same
# behaviour, written against ph.StrategicAgent, on the reference's StackelbergEnv)
# --------------------------------------------------------------------------------------
class SellerAgent(StrategicAgent):
    """Leader: posts a price to every neighbouring buyer; books revenue for orders."""
    device_kind = _abi.KIND_SELLER
    state_fields = {"price": "seller.price", "revenue": "seller.revenue", "tx": "seller.tx"}

    def __init__(self, agent_id: AgentID):
        super().__init__(agent_id)
        self.observation_space = Box(0.0, float("inf"), (2,))
        self.action_space = Box(0.0, 1.0, (1,))


class BuyerAgent(StrategicAgent):
    """Follower: buys one unit from the cheapest neighbouring seller when its action says so."""
    device_kind = _abi.KIND_BUYER
    state_fields = {"bought": "buyer.bought", "paid": "buyer.paid"}

    def __init__(self, agent_id: AgentID, value: float):
        super().__init__(agent_id)
        self.value = float(value)
        self.observation_space = Box(0.0, float("inf"), (2,))
        self.action_space = Box(0.0, 1.0, (1,))

    def device_params(self, index_of):
        return (), (self.value,)


# --------------------------------------------------------------------------------------
# kinds mirroring the agents of the reference's own known-answer tests
# --------------------------------------------------------------------------------------
class HalverAgent(Agent):
    """tests/network/test_tracking.py:21-28 (_TestActor): replies value // 2 while value > 1."""
    device_kind = _abi.KIND_HALVER


class CashboxAgent(Agent):
    """tests/network/test_network.py:17-34 (MockAgent)."""
    device_kind = _abi.KIND_CASHBOX
    state_fields = {"total_cash": "cashbox.total_cash"}


class ReqRespAgent(Agent):
    """tests/network/test_resolver.py:24-46 (_TestAgent); req_time/res_time are positions on a
    per-env logical clock (one tick per handled message) instead of time.time()."""
    device_kind = _abi.KIND_REQRESP
    state_fields = {"req_time": "reqresp.req_time", "res_time": "reqresp.res_time"}


class ForwarderAgent(Agent):
    """tests/network/test_resolver.py:89-96 (_TestAgent2): answers any message by sending an
    undecorated payload to ``target`` (whether or not an edge exists)."""
    device_kind = _abi.KIND_FORWARDER

    def __init__(self, agent_id: AgentID, target: Optional[AgentID] = None):
        super().__init__(agent_id)
        self.target = target

    def device_params(self, index_of):
        return (index_of(self.target) if self.target is not None else -1,), ()


class MockStrategicAgent(StrategicAgent):
    """tests/__init__.py:32-65: obs = proportion_time_elapsed, reward 0, done at num_steps."""
    device_kind = _abi.KIND_MOCK_STRAT
    state_fields = {"encode_obs_count": "mock.encode_obs_count",
                    "decode_action_count": "mock.decode_action_count",
                    "compute_reward_count": "mock.compute_reward_count"}

    @dataclass
    class Supertype(_Supertype):            # tests/__init__.py:34-36
        type_value: float = 0.0

    def __init__(self, agent_id: AgentID, num_steps: Optional[int] = None, supertype=None):
        super().__init__(agent_id, supertype=supertype)
        self.num_steps = num_steps
        self.action_space = Box(0, 1, (1,))
        self.observation_space = Box(0, 1, (1,))

    def device_params(self, index_of):
        return (-1 if self.num_steps is None else int(self.num_steps),), ()


class MockAgent(Agent):
    """tests/__init__.py:25-29"""
    device_kind = _abi.KIND_MOCK_AGENT

    def __init__(self, agent_id: AgentID, num_steps: Optional[int] = None):
        super().__init__(agent_id)
        self.num_steps = num_steps

"""phantom_amd -- MI355X-native PhantomEnv.step() hot path behind Phantom's public surface.

Host side: Python classes with the reference's names and signatures (PhantomEnv, Network,
Agent, ...).  Device side: hand-written HIP kernels for gfx950 behind a C ABI
(include/phantom_amd.h, phantom_amd/csrc).  See DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"

from . import _abi
from .agents import (Agent, BuyerAgent, CashboxAgent, CustomerAgent, FactoryAgent,
                     ForwarderAgent, HalverAgent, MockAgent, MockStrategicAgent, ReqRespAgent,
                     SellerAgent, ShopAgent, StrategicAgent, TypedShopAgent, UnsupportedAgentBehaviour,
                     msg_handler)
from .env import PhantomEnv
from .fsm import (FiniteStateMachineEnv, FSMRuntimeError, FSMStage, FSMValidationError, StageRule, state_independent, state_rules)
from .message import (AgentID, CashMessage, HalveMessage, Message, MsgPayload, Order,
                      OrderRequest, OrderResponse, Price, Request, Response, StockRequest,
                      StockResponse, msg_payload)
from .network import Network, NetworkError, StochasticNetwork
from .resolvers import BatchResolver, Resolver
from .spec import EnvSpec, compile_spec
from .stackelberg import StackelbergEnv
from .supply_chain import SupplyChainEnv, SupplyChainFSMEnv
from .supertype import Supertype
from .ads_market import AdExchangeAgent, AdvertiserAgent, DigitalAdsEnv, PublisherAgent
from .message import Ads, AuctionResult, Bid, ImpressionRequest, ImpressionResult
from . import samplers
from .samplers import (LambdaSampler, NormalArraySampler, NormalSampler, Sampler,
                       UniformArraySampler, UniformFloatSampler, UniformIntSampler)
from .views import AgentView, Context, EnvView, FSMEnvView, View
from . import ads_market, metrics, policy, rllib
from .policy import MLPPolicy
from .distributed import all_gather_trajectory, make_sharded_env, shard_batch

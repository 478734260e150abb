"""The digital-ads market (mirrors examples/environments/digital_ads_market/digital_ads_market.py).

Three agent kinds on a ``StochasticNetwork`` driven by a two-stage ``FiniteStateMachineEnv``:
the publisher announces an impression (a random user), the exchange forwards it to every
advertiser, the advertisers bid, the exchange overrides ``handle_batch`` to run ONE first- or
second-price auction over the round's bids (:429-516) and the publisher draws whether the user
clicked on the winner's ad.  On the device the auction is an inbox reduction of the exchange's
round, the draws come from exogenous inputs or the Philox stream.
"""
from dataclasses import dataclass
from typing import Dict, Iterable, Mapping, Optional

import numpy as np

from . import _abi
from .agents import Agent, Box, StrategicAgent
from .views import AgentView
from .fsm import FiniteStateMachineEnv, FSMStage
from .message import Ads, AuctionResult, Bid, ImpressionRequest, ImpressionResult  # noqa: F401
from .network import StochasticNetwork
from .resolvers import BatchResolver
from .supertype import Supertype as _Supertype

THEMES = ("sport", "travel", "science", "tech")      # keys of _USER_CLICK_PROBABILITIES :151-154


def _theme_index(theme: str) -> int:
    if theme not in THEMES:
        raise ValueError(f"theme '{theme}' is not one of {THEMES} (the click table's keys)")
    return THEMES.index(theme)


USERS_INFO = {1: {"age": 18, "zipcode": 94025}, 2: {"age": 40, "zipcode": 90250}}      # AdExchangeAgent.view :407-410


def _lookup_user(user, field):
    if isinstance(user, np.ndarray):
        return np.array([USERS_INFO[int(u)][field] if int(u) in USERS_INFO else 0.0 for u in user])
    return USERS_INFO[int(user)][field] if int(user) in USERS_INFO else 0.0


@dataclass(frozen=True)
class AdExchangeView(AgentView):
    """digital_ads_market.py:383-393: what the exchange exposes to its advertisers."""
    users_info: dict


class PublisherAgent(Agent):
    """digital_ads_market.py:140-196"""
    device_kind = _abi.KIND_PUBLISHER
    _USER_CLICK_PROBABILITIES = {
        1: {"sport": 0.0, "travel": 1.0, "science": 0.2, "tech": 0.8},
        2: {"sport": 1.0, "travel": 0.0, "science": 0.7, "tech": 0.1},
    }

    def __init__(self, agent_id: str, exchange_id: str, user_click_proba: Optional[dict] = None,
                 click_draws_per_step: int = 1):
        super().__init__(agent_id)
        self.exchange_id = exchange_id
        self.user_click_proba = user_click_proba or self._USER_CLICK_PROBABILITIES
        #: exogenous click draws one step can consume (one per Ads message it handles)
        self.click_draws_per_step = click_draws_per_step

    def device_params(self, index_of):
        table = [float(self.user_click_proba[u].get(t, 0.0)) for u in (1, 2) for t in THEMES]
        return (index_of(self.exchange_id), self.click_draws_per_step), tuple(table)


class AdvertiserAgent(StrategicAgent):
    """digital_ads_market.py:199-374.  Observation (a gym Dict there) is the row
    ``[type.budget, left / type.budget, user_id - 1]``; ``format_observation`` rebuilds the dict."""
    device_kind = _abi.KIND_ADVERTISER
    device_type_field = "budget"
    state_fields = {"left": "adv.left", "bid": "adv.bid", "step_clicks": "adv.step_clicks",
                    "step_wins": "adv.step_wins", "_current_user_id": "adv.user",
                    "total_clicks": "adv.total_clicks", "total_requests": "adv.total_requests",
                    "total_wins": "adv.total_wins"}

    @dataclass
    class Supertype(_Supertype):
        budget: float

    def __init__(self, agent_id: str, exchange_id: str, theme: str = "generic", supertype=None):
        super().__init__(agent_id, supertype=supertype)
        self.exchange_id = exchange_id
        self.theme = theme
        self.action_space = Box(0.0, 1.0, (1,))
        self.observation_space = Box(0.0, float("inf"), (3,))

    def device_params(self, index_of):
        return (index_of(self.exchange_id), _theme_index(self.theme), 0), (0.0,)   # pi2, pf0: compile_spec

    # handle_impression_request also caches the user's age / zipcode from the exchange's view (:261-266);
    # the shipped observation does not use them (commented out there), so they are derived on the host
    @property
    def _current_age(self):
        return _lookup_user(self._current_user_id, "age")

    @property
    def _current_zipcode(self):
        return _lookup_user(self._current_user_id, "zipcode")

    @staticmethod
    def format_observation(row: np.ndarray) -> Dict:
        """the Dict observation of :294-316 from the device row (budget_left was computed in the
        reference's precision and rounded to f32 on the way out)."""
        return {"type": {"budget": np.array([row[0]], dtype=np.float32)},
                "budget_left": np.array([row[1]], dtype=np.float64),
                "user_id": int(row[2])}


class AdExchangeAgent(Agent):
    """digital_ads_market.py:377-516; ``strategy`` "first" | "second" price."""
    device_kind = _abi.KIND_ADEXCHANGE

    def __init__(self, agent_id: str, publisher_id: str, advertiser_ids: Iterable = tuple(),
                 strategy: str = "first"):
        super().__init__(agent_id)
        if strategy not in ("first", "second"):
            raise ValueError(f"Unknown auction strategy: {strategy}")       # :453-454
        self.publisher_id = publisher_id
        self.advertiser_ids = list(advertiser_ids)
        self.strategy = strategy

    def view(self, neighbour_id=None):
        """:399-413: advertisers (ids starting with "ADV") get the users' info, everybody else nothing."""
        if neighbour_id and str(neighbour_id).startswith("ADV"):
            return AdExchangeView(users_info=USERS_INFO)
        return super().view(neighbour_id)

    def device_params(self, index_of):
        for aid in self.advertiser_ids:
            index_of(aid)
        return (index_of(self.publisher_id), int(self.strategy == "second")), ()

    def check_topology(self, network) -> None:
        """the device forwards an ImpressionRequest to the exchange's AdvertiserAgent neighbours in
        base-connection order; that must be exactly ``advertiser_ids`` (:427)."""
        base = [n for n, _ in network.base_neighbors(self.id)] if hasattr(network, "base_neighbors") \
            else list(network.neighbors(self.id))
        advs = [n for n in base if isinstance(network.agents[n], AdvertiserAgent)]
        if advs != self.advertiser_ids:
            raise ValueError(f"AdExchangeAgent '{self.id}': advertiser_ids must equal its AdvertiserAgent "
                             f"neighbours in connection order ({advs})")


class DigitalAdsEnv(FiniteStateMachineEnv):
    """digital_ads_market.py:525-596"""

    def __init__(self, num_steps: int = 20, num_agents_theme: Optional[Mapping[str, int]] = None,
                 strategy: str = "first", user_click_proba: Optional[dict] = None,
                 connection_rates=(1.0, 1.0, 1.0), **kwargs):
        """``strategy`` (AdExchangeAgent's, "first" there) and ``connection_rates`` (exchange-publisher,
        exchange-advertisers, publisher-advertisers; the shipped env uses StochasticNetwork's default
        1.0) are extensions of the shipped constructor."""
        self.exchange_id, self.publisher_id = "ADX", "PUB"
        click = user_click_proba or {
            1: {"sport": 0.0, "travel": 1.0, "science": 0.2, "tech": 0.5},
            2: {"sport": 1.0, "travel": 0.0, "science": 0.7, "tech": 0.5},
        }
        publisher = PublisherAgent(self.publisher_id, exchange_id=self.exchange_id, user_click_proba=click)
        advertisers, i = [], 1
        for theme, n in (num_agents_theme or {}).items():
            for _ in range(n):
                advertisers.append(AdvertiserAgent(f"ADV_{i}", self.exchange_id, theme=theme))
                i += 1
        self.advertiser_ids = [a.id for a in advertisers]
        exchange = AdExchangeAgent(self.exchange_id, publisher_id=self.publisher_id,
                                   advertiser_ids=self.advertiser_ids, strategy=strategy)
        network = StochasticNetwork([exchange, publisher] + advertisers, BatchResolver(round_limit=5),
                                    ignore_connection_errors=True)
        network.add_connections_between([self.exchange_id], [self.publisher_id], connection_rates[0])
        network.add_connections_between([self.exchange_id], self.advertiser_ids, connection_rates[1])
        network.add_connections_between([self.publisher_id], self.advertiser_ids, connection_rates[2])
        kwargs.setdefault("exogenous", "device")     # np.random.binomial(1, p) cannot be pre-drawn on the host
        super().__init__(
            num_steps=num_steps, network=network, initial_stage="publisher_step",
            stages=[
                FSMStage("publisher_step", next_stages=["advertiser_step"],
                         acting_agents=[self.publisher_id], rewarded_agents=[self.publisher_id]),
                FSMStage("advertiser_step", next_stages=["publisher_step"],
                         acting_agents=self.advertiser_ids, rewarded_agents=self.advertiser_ids),
            ], **kwargs)

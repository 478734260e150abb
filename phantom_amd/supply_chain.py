"""The supply-chain workload (mirrors examples/environments/supply_chain/supply_chain.py).

``SupplyChainEnv()`` with no arguments is the shipped 7-agent example (:153-175); the keyword
arguments build the synthetic benchmark topologies of BASELINE.json (SC64 = 1 factory,
9 shops, 6 customers per shop; SC256 = 1 + 51 + 51*4) with the agent order
``[shops..., factory, customers...]`` of supply_chain.py:164.
"""
from typing import List, Optional, Sequence

from .agents import (CUSTOMER_MAX_ORDER_SIZE, SHOP_MAX_STOCK, CustomerAgent, FactoryAgent,
                     ShopAgent, TypedShopAgent)
from .env import PhantomEnv
from .fsm import FiniteStateMachineEnv, FSMStage
from .message import OrderRequest, OrderResponse, StockRequest, StockResponse  # noqa: F401
from .network import Network

NUM_EPISODE_STEPS = 100      # supply_chain.py:9
NUM_CUSTOMERS = 5            # supply_chain.py:11


def build_network(n_shops: int = 1, customers_per_shop=NUM_CUSTOMERS, resolver=None,
                  typed: bool = False) -> Network:
    """Supply-chain topology.  ``customers_per_shop`` may be an int or a per-shop sequence
    (ragged).  With the defaults the ids are those of the shipped example.  ``typed`` builds
    tutorial 2's shops (TypedShopAgent: Supertype.excess_stock_weight)."""
    ks: Sequence[int] = ([customers_per_shop] * n_shops if isinstance(customers_per_shop, int)
                         else list(customers_per_shop))
    assert len(ks) == n_shops
    if n_shops == 1:
        factory_id, shop_ids = "WAREHOUSE", ["SHOP"]
        cust_ids = [[f"CUST{i + 1}" for i in range(ks[0])]]
    else:
        factory_id = "WAREHOUSE"
        shop_ids = [f"SHOP{i}" for i in range(n_shops)]
        cust_ids = [[f"CUST{i}_{j}" for j in range(ks[i])] for i in range(n_shops)]
    shop_cls = TypedShopAgent if typed else ShopAgent
    shops = [shop_cls(sid, factory_id=factory_id, num_customers=ks[i])
             for i, sid in enumerate(shop_ids)]
    customers = [CustomerAgent(cid, shop_id=shop_ids[i]) for i in range(n_shops) for cid in cust_ids[i]]
    network = Network(shops + [FactoryAgent(factory_id)] + customers, resolver=resolver)
    for i, sid in enumerate(shop_ids):
        network.add_connection(sid, factory_id)                 # supply_chain.py:170
    for i, sid in enumerate(shop_ids):
        network.add_connections_between([sid], cust_ids[i])     # supply_chain.py:173
    return network


class SupplyChainEnv(PhantomEnv):
    def __init__(self, n_shops: int = 1, customers_per_shop=NUM_CUSTOMERS,
                 num_steps: int = NUM_EPISODE_STEPS, resolver=None, typed: bool = False,
                 **device_kwargs):
        network = build_network(n_shops, customers_per_shop, resolver, typed)
        super().__init__(num_steps=num_steps, network=network, **device_kwargs)


class SupplyChainFSMEnv(FiniteStateMachineEnv):
    """BASELINE config 3: RESTOCK{acting=shops, rewarded=shops} -> SELL{acting=customers,
    rewarded=[]} -> RESTOCK, handler-less."""

    def __init__(self, n_shops: int = 1, customers_per_shop=NUM_CUSTOMERS,
                 num_steps: int = NUM_EPISODE_STEPS, resolver=None, typed: bool = False,
                 restock_handler=None, **device_kwargs):
        """``restock_handler``: an FSM stage handler for RESTOCK (fsm.py:294-307) choosing between SELL and
        another RESTOCK (called as ``handler(env)``); None -> the handler-less two-stage cycle."""
        network = build_network(n_shops, customers_per_shop, resolver, typed)
        shops = [a.id for a in network.agents.values() if isinstance(a, ShopAgent)]
        customers = [a.id for a in network.agents.values() if isinstance(a, CustomerAgent)]
        stages = [
            FSMStage("RESTOCK", acting_agents=shops, rewarded_agents=shops,
                     next_stages=["SELL", "RESTOCK"] if restock_handler else ["SELL"], handler=restock_handler),
            FSMStage("SELL", acting_agents=customers, rewarded_agents=[], next_stages=["RESTOCK"]),
        ]
        super().__init__(num_steps=num_steps, network=network, initial_stage="RESTOCK",
                         stages=stages, **device_kwargs)

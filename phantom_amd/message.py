"""Message / payload types (mirrors phantom/message.py:10-53).

On the device a payload is ``{type u16, 8-byte value}``.  The closed set of payload classes
below covers the supply-chain example (supply_chain.py:16-33), the Stackelberg market and
the payloads of the reference's network tests.
"""
from dataclasses import dataclass
from typing import Any, Hashable

from . import _abi

AgentID = Hashable


@dataclass(frozen=True)
class MsgPayload:
    """Deprecated payload base (message.py:10-12); kept for name parity."""


def msg_payload(sender_type=None, receiver_type=None):
    """Decorator parity with message.py:20-42.  New payload classes can be declared for host
    bookkeeping, but only classes carrying ``_msg_type`` (below) can travel on the device."""
    def wrap(cls):
        def names(t):
            if t is None:
                return None
            t = t if isinstance(t, list) else [t]
            return [x.__name__ if isinstance(x, type) else x for x in t]
        cls._sender_types = names(sender_type)
        cls._receiver_types = names(receiver_type)
        return dataclass(frozen=True)(cls)
    return wrap


@dataclass(frozen=True)
class Message:
    """message.py:45-53"""
    sender_id: AgentID
    receiver_id: AgentID
    payload: Any


def _device_payload(msg_type, field, sender=None, receiver=None):
    def wrap(cls):
        cls = msg_payload(sender, receiver)(cls)
        cls._msg_type = msg_type
        cls._field = field
        PAYLOAD_BY_TYPE[msg_type] = cls
        return cls
    return wrap


PAYLOAD_BY_TYPE = {}


@_device_payload(_abi.MSG_ORDER_REQUEST, "size", "CustomerAgent", "ShopAgent")
class OrderRequest:            # supply_chain.py:16-18
    size: int


@_device_payload(_abi.MSG_ORDER_RESPONSE, "size", "ShopAgent", "CustomerAgent")
class OrderResponse:           # supply_chain.py:21-23
    size: int


@_device_payload(_abi.MSG_STOCK_REQUEST, "size", "ShopAgent", "FactoryAgent")
class StockRequest:            # supply_chain.py:26-28
    size: int


@_device_payload(_abi.MSG_STOCK_RESPONSE, "size", "FactoryAgent", "ShopAgent")
class StockResponse:           # supply_chain.py:31-33
    size: int


@_device_payload(_abi.MSG_PRICE, "price", "SellerAgent", "BuyerAgent")
class Price:
    price: float


@_device_payload(_abi.MSG_ORDER, "vol", "BuyerAgent", "SellerAgent")
class Order:
    vol: int


@_device_payload(_abi.MSG_HALVE, "value")
class HalveMessage:            # tests/network/test_tracking.py:16-18 _TestMessage
    value: int


@_device_payload(_abi.MSG_CASH, "cash")
class CashMessage:             # tests/network/test_network.py:12-14 MockMessage
    cash: float


@_device_payload(_abi.MSG_REQUEST, "cash")
class Request:                 # tests/network/test_resolver.py:14-16
    cash: float


@_device_payload(_abi.MSG_RESPONSE, "cash")
class Response:                # tests/network/test_resolver.py:19-21
    cash: float


# digital_ads_market.py:28-121.  On the device the small fields ride in the record's aux bits
# (theme | user_id << 4 | tag << 8); only the 8-byte value is logged.
@_device_payload(_abi.MSG_IMPRESSION_REQ, "user_id")
class ImpressionRequest:
    user_id: int
    timestamp: float = 0.0


@_device_payload(_abi.MSG_BID, "bid")
class Bid:
    bid: float
    theme: Any = 0
    user_id: int = 0


@_device_payload(_abi.MSG_AUCTION_RESULT, "cost")
class AuctionResult:
    cost: float
    winning_bid: float = 0.0


@_device_payload(_abi.MSG_ADS, "advertiser_id")
class Ads:
    advertiser_id: Any
    theme: Any = 0
    user_id: int = 0


@_device_payload(_abi.MSG_IMPRESSION_RES, "clicked")
class ImpressionResult:
    clicked: bool


def payload_to_record(payload):
    """(type id, is_float, value) of a payload instance; TypeError if it has no device type."""
    t = getattr(type(payload), "_msg_type", None)
    if t is None:
        if payload is True or payload is False:
            return _abi.MSG_PING, False, int(payload)
        raise TypeError(
            f"payload {payload!r} has no device message type; only the closed set in "
            "phantom_amd.message can be sent on the device network")
    v = getattr(payload, type(payload)._field)
    return t, t in _abi.FLOAT_PAYLOAD_TYPES, v


def record_to_payload(msg_type, raw_i, raw_f):
    if msg_type == _abi.MSG_PING:
        return True
    cls = PAYLOAD_BY_TYPE[msg_type]
    return cls(raw_f if msg_type in _abi.FLOAT_PAYLOAD_TYPES else int(raw_i))

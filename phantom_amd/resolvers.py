"""Resolver configuration objects (mirrors phantom/resolvers.py:17-163).

The round loop itself (BatchResolver.resolve, resolvers.py:128-163) runs inside the HIP
kernels; this class carries its options into the spec and exposes ``tracked_messages``
decoded from the device message log.
"""
from typing import List, Optional

from .message import Message


class Resolver:
    """resolvers.py:17-89"""

    def __init__(self, enable_tracking: bool = False) -> None:
        self.enable_tracking = enable_tracking
        self._tracked_messages: List[Message] = []

    def clear_tracked_messages(self) -> None:      # resolvers.py:48-53
        self._tracked_messages.clear()

    @property
    def tracked_messages(self) -> List[Message]:   # resolvers.py:55-60
        return self._tracked_messages

    def reset(self) -> None:
        return None


class BatchResolver(Resolver):
    """resolvers.py:92-163.  ``shuffle_batches`` (resolvers.py:150-151: ``np.random.shuffle`` of every delivered
    batch) runs on the generic engine: the device draws the permutations from its own Philox stream (one per
    (env, tick, round, receiver)), or replays recorded ones (``DeviceEnv.step(..., shuffle=)``) -- the reference
    draws from the global numpy stream INSIDE the round loop, which a batched launch cannot interleave with."""

    def __init__(self, enable_tracking: bool = False, round_limit: Optional[int] = None,
                 shuffle_batches: bool = False, trace_capacity: Optional[int] = None) -> None:
        super().__init__(enable_tracking)
        self.round_limit = round_limit
        self.shuffle_batches = bool(shuffle_batches)
        self.trace_capacity = trace_capacity

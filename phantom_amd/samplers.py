"""Samplers (mirrors phantom/utils/samplers.py:47-271).

A Sampler is the reset-time source of a Supertype field.  The host classes below draw from the
global legacy numpy stream exactly like the reference (so a seeded B = 1 run consumes the stream
in the reference's order); a ``UniformFloatSampler`` can instead be drawn on the device
(``PHX_SAMPLER_UNIFORM``, one Philox block per env and reset) when the env runs with
``exogenous="device"`` -- that is what lets fused rollouts auto-reset without the host.
"""
from typing import Callable, Generic, Iterable, Optional, Tuple, TypeVar

import numpy as np

T = TypeVar("T")


class Sampler(Generic[T]):
    """samplers.py:47-73: ``sample()`` draws and stores the value, ``value`` reads it back."""

    def __init__(self) -> None:
        self._value: Optional[T] = None

    @property
    def value(self) -> Optional[T]:
        return self._value

    def sample(self) -> T:
        raise NotImplementedError


class ComparableSampler(Sampler[T]):
    """samplers.py:76-117: compares like its current value (identity against other samplers)."""

    def _v(self):
        if self._value is None:
            raise ValueError("`self.value` is None")
        return self._value

    def __lt__(self, other):
        if isinstance(other, ComparableSampler):
            return NotImplemented
        return self._v() < other

    def __eq__(self, other):
        if isinstance(other, ComparableSampler):
            return self is other
        return self._value == other

    def __ne__(self, other):
        return not self.__eq__(other)

    def __le__(self, other):
        return self.__lt__(other) or self.__eq__(other)

    def __gt__(self, other):
        return not self.__le__(other)

    def __ge__(self, other):
        return self.__gt__(other) or self.__eq__(other)

    __hash__ = object.__hash__


def _clip(v, lo, hi):
    if lo is not None or hi is not None:                    # samplers.py:143-144
        v = np.clip(v, lo, hi)
    return v


class UniformFloatSampler(ComparableSampler[float]):
    """samplers.py:120-147 (np.random.uniform).  Device-drawable."""

    def __init__(self, low: float = 0.0, high: float = 1.0, clip_low: Optional[float] = None,
                 clip_high: Optional[float] = None) -> None:
        assert high >= low
        self.low, self.high, self.clip_low, self.clip_high = low, high, clip_low, clip_high
        super().__init__()

    def sample(self) -> float:
        self._value = _clip(np.random.uniform(self.low, self.high), self.clip_low, self.clip_high)
        return self._value

    def device_params(self) -> Tuple[float, float, float, float]:
        nan = float("nan")
        return (float(self.low), float(self.high),
                nan if self.clip_low is None else float(self.clip_low),
                nan if self.clip_high is None else float(self.clip_high))


class UniformIntSampler(ComparableSampler[int]):
    """samplers.py:150-177 (np.random.randint)."""

    def __init__(self, low: int = 0, high: int = 1, clip_low: Optional[int] = None,
                 clip_high: Optional[int] = None) -> None:
        assert high >= low
        self.low, self.high, self.clip_low, self.clip_high = low, high, clip_low, clip_high
        super().__init__()

    def sample(self) -> int:
        self._value = _clip(np.random.randint(self.low, self.high), self.clip_low, self.clip_high)
        return self._value


class UniformArraySampler(ComparableSampler[np.ndarray]):
    """samplers.py:180-209.  Host-only (an array field has no device consumer)."""

    def __init__(self, low: float = 0.0, high: float = 1.0, shape: Iterable[int] = (1,),
                 clip_low: Optional[float] = None, clip_high: Optional[float] = None) -> None:
        assert high >= low
        self.low, self.high, self.shape = low, high, shape
        self.clip_low, self.clip_high = clip_low, clip_high
        super().__init__()

    def sample(self) -> np.ndarray:
        self._value = _clip(np.random.uniform(self.low, self.high, self.shape), self.clip_low,
                            self.clip_high)
        return self._value


class NormalSampler(ComparableSampler[float]):
    """samplers.py:212-237 (np.random.normal)."""

    def __init__(self, mu: float, sigma: float, clip_low: Optional[float] = None,
                 clip_high: Optional[float] = None) -> None:
        self.mu, self.sigma, self.clip_low, self.clip_high = mu, sigma, clip_low, clip_high
        super().__init__()

    def sample(self) -> float:
        self._value = _clip(np.random.normal(self.mu, self.sigma), self.clip_low, self.clip_high)
        return self._value


class NormalArraySampler(ComparableSampler[np.ndarray]):
    """samplers.py:240-269."""

    def __init__(self, mu: float, sigma: float, shape: Tuple[int] = (1,),
                 clip_low: Optional[float] = None, clip_high: Optional[float] = None) -> None:
        self.mu, self.sigma, self.shape = mu, sigma, shape
        self.clip_low, self.clip_high = clip_low, clip_high
        super().__init__()

    def sample(self) -> np.ndarray:
        self._value = _clip(np.random.normal(self.mu, self.sigma, self.shape), self.clip_low,
                            self.clip_high)
        return self._value


class LambdaSampler(Sampler[T]):
    """samplers.py:272-285: arbitrary callable."""

    def __init__(self, func: Callable[..., T], *args, **kwargs):
        self.func, self.args, self.kwargs = func, args, kwargs
        super().__init__()

    def sample(self) -> T:
        self._value = self.func(*self.args, **self.kwargs)
        return self._value

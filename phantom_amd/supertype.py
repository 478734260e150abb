"""Supertype (mirrors phantom/supertype.py:13-107).

Subclass with ``@dataclass``; fields are plain values or Samplers.  ``sample()`` returns the
*type*: the same dataclass with every Sampler replaced by a drawn value -- a fresh draw for a
free-standing supertype, the env-managed Sampler's current value once an env owns it
(supertype.py:23-26).
"""
from dataclasses import dataclass
from typing import Any, Dict

import numpy as np

from .samplers import Sampler


@dataclass
class Supertype:
    def sample(self) -> "Supertype":
        out = {}
        for name in self.__dataclass_fields__:
            v = getattr(self, name)
            if isinstance(v, Sampler):
                out[name] = v.value if hasattr(self, "_managed") else v.sample()
            else:
                out[name] = v
        return self.__class__(**out)

    def to_obs_space_compatible_type(self) -> Dict[str, Any]:
        """supertype.py:32-42"""
        return {name: _compatible(name, getattr(self, name)) for name in self.__dataclass_fields__}

    def to_obs_space(self, low=-np.inf, high=np.inf):
        """supertype.py:44-62, with the Box stand-in of phantom_amd.agents (no gymnasium)."""
        return {name: _space(name, getattr(self, name), low, high) for name in self.__dataclass_fields__}


def _compatible(field: str, obj: Any):
    if isinstance(obj, dict):
        return {k: _compatible(k, v) for k, v in obj.items()}
    if isinstance(obj, (float, int)):
        return np.array([obj], dtype=np.float32)
    if isinstance(obj, list):
        return [_compatible(f"{field}[{i}]", v) for i, v in enumerate(obj)]
    if isinstance(obj, tuple):
        return tuple(_compatible(f"{field}[{i}]", v) for i, v in enumerate(obj))
    if isinstance(obj, np.ndarray):
        return obj
    raise ValueError(f"Can't encode field '{field}' with type '{type(obj)}' into obs space compatible type")


def _space(field: str, obj: Any, low, high):
    from .agents import Box
    if isinstance(obj, dict):
        return {k: _space(k, v, low, high) for k, v in obj.items()}
    if isinstance(obj, (float, int)):
        return Box(low, high, (1,))
    if isinstance(obj, (list, tuple)):
        return tuple(_space(f"{field}[{i}]", v, low, high) for i, v in enumerate(obj))
    if isinstance(obj, np.ndarray):
        return Box(low, high, obj.shape)
    raise ValueError(f"Can't encode field '{field}' with type '{type(obj)}' into gym.Space")

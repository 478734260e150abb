"""Metric helpers over the batched env (the caller-side contract of phantom/metrics.py:38-231).

``SimpleAgentMetric("SHOP", "stock")`` in the reference reflects on a Python agent object
(``_rgetattr(env.agents[id], property)``, metrics.py:230-231).  Agent attributes of the device
kinds are lazy read-backs of device state, so the same reflection works here; with
``batch_size > 1`` each extracted value is an array with one entry per env instance and the
reductions act along the step axis.
"""
from typing import Optional, Sequence

import numpy as np


class NotRecorded:                     # metrics.py:23-33
    def __new__(cls):
        if not hasattr(cls, "instance"):
            cls.instance = super().__new__(cls)
        return cls.instance

    def __repr__(self) -> str:
        return "<NotRecorded>"


not_recorded = NotRecorded()


def _rgetattr(obj, attr):
    for part in attr.split("."):
        obj = getattr(obj, part)
    return obj


class Metric:
    """metrics.py:38-77"""

    def __init__(self, fsm_stages: Optional[Sequence] = None, description: Optional[str] = None):
        self.fsm_stages = fsm_stages
        self.description = description

    def extract(self, env):
        raise NotImplementedError

    def reduce(self, values, mode):
        return values[-1]


#: how an episode's per-step values become one number (metrics.py:170-186): a table of reducers over the values stacked on axis 0
#: (the batch axis stays); "none" keeps every step
_REDUCERS = {
    "last": lambda vals: vals[-1] if len(vals) else None,
    "mean": lambda vals: np.mean(vals, axis=0),
    "sum": lambda vals: np.sum(vals, axis=0),
    "none": np.array,
}
#: (argument name, reducers it accepts, the reference's wording of the complaint)
_ACTION_ARGS = (
    ("train_reduce_action", ("last", "mean", "sum"), "metric must be one of: 'last', 'mean' or 'sum'"),
    ("eval_reduce_action", ("last", "mean", "sum", "none"), "metric class must be one of: 'last', 'mean', 'sum' or 'none'"),
)


class SimpleMetric(Metric):
    """metrics.py:141-186 -- the reduce actions as a dispatch table (``_REDUCERS``)"""

    def __init__(self, train_reduce_action="mean", eval_reduce_action="none", fsm_stages=None,
                 description=None):
        given = {"train_reduce_action": train_reduce_action, "eval_reduce_action": eval_reduce_action}
        for name, accepted, wording in _ACTION_ARGS:
            if given[name] not in accepted:
                raise ValueError(f"{name} field of {self.__class__} {wording}. Got '{given[name]}'.")
            setattr(self, name, given[name])
        super().__init__(fsm_stages, description)

    def reduce(self, values, mode):
        action = self.train_reduce_action if mode == "train" else self.eval_reduce_action
        if action != "none" and self.fsm_stages is not None:   # steps of other stages recorded the `not_recorded` marker
            values = [v for v in values if v is not not_recorded]
        return _REDUCERS[action](values)


class SimpleAgentMetric(SimpleMetric):
    """metrics.py:189-231"""

    def __init__(self, agent_id, agent_property, train_reduce_action="mean",
                 eval_reduce_action="none", fsm_stages=None, description=None):
        self.agent_id = agent_id
        self.agent_property = agent_property
        super().__init__(train_reduce_action, eval_reduce_action, fsm_stages, description)

    def extract(self, env):
        return _rgetattr(env.agents[self.agent_id], self.agent_property)


class SimpleEnvMetric(SimpleMetric):
    """metrics.py:234-270"""

    def __init__(self, env_property, train_reduce_action="mean", eval_reduce_action="none",
                 fsm_stages=None, description=None):
        self.env_property = env_property
        super().__init__(train_reduce_action, eval_reduce_action, fsm_stages, description)

    def extract(self, env):
        return _rgetattr(env, self.env_property)


def logging_helper(env, metrics, metric_values) -> None:
    """metrics.py:355-370: record every metric once per step, honouring FSM-stage filters."""
    for metric_id, metric in metrics.items():
        stage = getattr(env, "current_stage", None)
        if metric.fsm_stages is None or stage in metric.fsm_stages:
            metric_values.setdefault(metric_id, []).append(metric.extract(env))
        else:
            metric_values.setdefault(metric_id, []).append(not_recorded)

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


class SimpleMetric(Metric):
    """metrics.py:141-186"""

    def __init__(self, train_reduce_action="mean", eval_reduce_action="none", fsm_stages=None,
                 description=None):
        if train_reduce_action not in ("last", "mean", "sum"):
            raise ValueError(f"train_reduce_action field of {self.__class__} metric must be one of: "
                             f"'last', 'mean' or 'sum'. Got '{train_reduce_action}'.")
        if eval_reduce_action not in ("last", "mean", "sum", "none"):
            raise ValueError(f"eval_reduce_action field of {self.__class__} metric class must be one "
                             f"of: 'last', 'mean', 'sum' or 'none'. Got '{eval_reduce_action}'.")
        self.train_reduce_action = train_reduce_action
        self.eval_reduce_action = eval_reduce_action
        super().__init__(fsm_stages, description)

    def reduce(self, values, mode):
        action = self.train_reduce_action if mode == "train" else self.eval_reduce_action
        if action == "none":
            return np.array(values)
        if self.fsm_stages is not None:
            values = [v for v in values if v is not not_recorded]
        if action == "last":
            return values[-1] if len(values) > 0 else None
        if action == "mean":
            return np.mean(values, axis=0)
        return np.sum(values, axis=0)


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

"""Interface transcript of the two ray[rllib]==2.7.1 classes the reference's RLlib glue stands on -- method NAMES, parameter
names / kinds and defaults only, no ray code -- plus the call shapes the reference itself makes on them.  ray is absent
from the build image, so this transcript is what tests/test_rllib_conformance.py holds phantom_amd.rllib against.

    ray.rllib.env.multi_agent_env.MultiAgentEnv    (wrapper.py:10 subclasses it)
    ray.rllib.env.base_env.BaseEnv                 (RLlib's vector env: what env runners poll; train.py:294-297 reads
                                                    ``base_env.envs[0]`` inside RLlibMetricLogger.on_episode_step)

MultiEnvDict = {env_id: {agent_id: value}};  MultiAgentDict = {agent_id: value}.
"""
import inspect

P = inspect.Parameter
_E = P.empty


def _sig(*params):
    return inspect.Signature([P("self", P.POSITIONAL_OR_KEYWORD)] + [P(n, k, default=d) for n, k, d in params])


PK, KW = P.POSITIONAL_OR_KEYWORD, P.KEYWORD_ONLY

#: method name -> signature (self first)
BASE_ENV = {
    "poll": _sig(),
    "send_actions": _sig(("action_dict", PK, _E)),
    "try_reset": _sig(("env_id", PK, None), ("seed", KW, None), ("options", KW, None)),
    "try_restart": _sig(("env_id", PK, None)),
    "get_sub_environments": _sig(("as_dict", PK, False)),
    "get_agent_ids": _sig(),
    "try_render": _sig(("env_id", PK, None)),
    "stop": _sig(),
    "action_space_sample": _sig(("agent_id", PK, None)),
    "observation_space_sample": _sig(("agent_id", PK, None)),
    "last": _sig(),
    "observation_space_contains": _sig(("x", PK, _E)),
    "action_space_contains": _sig(("x", PK, _E)),
    "to_base_env": _sig(("make_env", PK, None), ("num_envs", PK, 1), ("remote_envs", PK, False),
                        ("remote_env_batch_wait_ms", PK, 0), ("restart_failed_sub_environments", PK, False)),
}
BASE_ENV_PROPERTIES = ("num_envs", "observation_space", "action_space")

MULTI_AGENT_ENV = {
    "reset": _sig(("seed", KW, None), ("options", KW, None)),
    "step": _sig(("action_dict", PK, _E)),
    "get_agent_ids": _sig(),
    "to_base_env": BASE_ENV["to_base_env"],
}

#: call shapes the REFERENCE makes (file:line) -- (method, args, kwargs); each must bind against our signature
REFERENCE_CALLS_WRAPPER = [
    ("step", ({"SHOP": [1.0]},), {}),                     # wrapper.py:39-40 via RLlib: env.step(action_dict)
    ("reset", (), {"seed": 3, "options": None}),          # RLlib env runner: env.reset(seed=..., options=...)
    ("reset", (3,), {}),                                  # wrapper.py:42-45 keeps seed / options positional too
    ("reset", (), {}),
    ("get_agent_ids", (), {}),
]
REFERENCE_CALLS_BASE_ENV = [
    ("poll", (), {}),                                     # env runner loop
    ("send_actions", ({0: {"SHOP": [1.0]}},), {}),        # MultiEnvDict
    ("try_reset", (), {}), ("try_reset", (2,), {}), ("try_reset", (2,), {"seed": 5, "options": {}}),
    ("try_restart", (1,), {}),
    ("get_sub_environments", (), {}), ("get_sub_environments", (), {"as_dict": True}),
    ("stop", (), {}),
]


def binds(fn, args, kwargs) -> bool:
    try:
        inspect.signature(fn).bind(*args, **kwargs)
        return True
    except TypeError:
        return False


def accepts_like(ours, theirs: inspect.Signature) -> list:
    """every parameter of the transcript exists in ours with the same name and can be passed the same way (ours may
    be more permissive: positional-or-keyword where the transcript is keyword-only, extra parameters with defaults)"""
    problems = []
    sig = inspect.signature(ours)
    mine = sig.parameters
    for name, p in list(theirs.parameters.items())[1:]:
        q = mine.get(name)
        if q is None:
            if not any(x.kind == P.VAR_KEYWORD for x in mine.values()):
                problems.append(f"missing parameter {name}")
            continue
        if p.kind == PK and q.kind not in (PK,):
            problems.append(f"{name}: must be positional-or-keyword")
        if p.kind == KW and q.kind not in (PK, KW):
            problems.append(f"{name}: must accept a keyword")
        if p.default is not _E and q.default is _E:
            problems.append(f"{name}: needs a default")
    for name, q in list(mine.items())[1:]:
        if name not in theirs.parameters and q.default is _E and q.kind in (PK, KW):
            problems.append(f"extra required parameter {name}")
    return problems

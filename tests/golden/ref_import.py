"""Import the reference (jpmorganchase/Phantom at /root/reference) in THIS container.

Build-container-only helper for generating golden vectors (tests/golden/gen_goldens.py)
and for cross-checking the C oracle against the live reference.  The reference needs
gymnasium / ray / tensorboardX / termcolor, which are absent from the image; none of
them take part in the hot-path arithmetic (gymnasium only contributes ``Env.reset``
seeding, which nothing in Phantom consumes, and ``spaces.Box`` objects), so they are
replaced by inert stand-in modules *in sys.modules only* (SURVEY.md Appendix C).
Nothing from /root/reference is copied; this file never travels to the GPU box's
test run because every consumer is skipped when /root/reference is missing.
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PHANTOM_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "phantom"))


class _Anything:
    """Inert stand-in usable as a base class, decorator, callable or container."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()

    def __class_getitem__(cls, item):
        return cls


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_Anything,), {})
        setattr(self, name, cls)
        return cls


def _stub(name: str) -> types.ModuleType:
    mod = _StubModule(name)
    mod.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    mod.__path__ = []
    sys.modules[name] = mod
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], child, mod)
    return mod


def _install_stubs() -> None:
    if "gymnasium" not in sys.modules:
        gym = _stub("gymnasium")

        class Env:
            def reset(self, seed=None, options=None):
                return None

        class Space:
            def __init__(self, *a, **k):
                self.args, self.kwargs = a, k

            def __eq__(self, other):
                return type(self) is type(other) and repr(self.args) == repr(other.args) \
                    and repr(self.kwargs) == repr(other.kwargs)

            __hash__ = None

        gym.Env, gym.Space = Env, Space
        spaces = _stub("gymnasium.spaces")
        spaces.Space = Space
        for n in ("Box", "Discrete", "Tuple", "Dict", "MultiDiscrete", "MultiBinary"):
            setattr(spaces, n, type(n, (Space,), {}))
    if "termcolor" not in sys.modules:
        tc = _stub("termcolor")
        tc.colored = lambda s, *a, **k: s
    if "tensorboardX" not in sys.modules:
        _stub("tensorboardX")
    if "ray" not in sys.modules:
        for n in ("ray", "ray.tune", "ray.tune.result", "ray.tune.registry", "ray.tune.logger",
                  "ray.rllib", "ray.rllib.algorithms", "ray.rllib.algorithms.callbacks",
                  "ray.rllib.evaluation", "ray.rllib.policy", "ray.rllib.policy.sample_batch",
                  "ray.rllib.utils", "ray.rllib.utils.typing", "ray.rllib.utils.spaces",
                  "ray.rllib.utils.spaces.space_utils", "ray.rllib.models",
                  "ray.rllib.models.preprocessors", "ray.util", "ray.util.queue"):
            _stub(n)
        sys.modules["ray.tune.result"].DEFAULT_RESULTS_DIR = "/tmp/ray_results"


_ph = None
_sc = None


def import_phantom():
    """Return the reference ``phantom`` package (imported from /root/reference)."""
    global _ph
    if _ph is None:
        if not reference_available():
            raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
        _install_stubs()
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)
        import phantom  # noqa: the reference package
        assert os.path.realpath(phantom.__file__).startswith(os.path.realpath(REFERENCE_ROOT))
        _ph = phantom
    return _ph


def import_supply_chain():
    """Return the reference example module examples/environments/supply_chain/supply_chain.py."""
    global _sc
    if _sc is None:
        import_phantom()
        import matplotlib
        matplotlib.use("Agg")
        d = os.path.join(REFERENCE_ROOT, "examples/environments/supply_chain")
        if d not in sys.path:
            sys.path.insert(0, d)
        argv = sys.argv
        sys.argv = ["x", "noop"]   # the module dispatches on sys.argv[1] at import time
        try:
            import supply_chain
        finally:
            sys.argv = argv
        _sc = supply_chain
    return _sc

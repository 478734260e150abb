"""CPU: the restatement behind the product's OWN symbols (oracle/libphantom_cpu.so, SURVEY 8b / 8d "exports the identical
symbols").  The ctypes stub a Phantom maintainer would add (phantom_amd/_abi.py: bind_signatures -- INTEGRATION.md section 2)
is applied UNCHANGED to that library; the reference's goldens are replayed through phx_create / phx_reset / phx_step with host
numpy buffers, and phx_rollout / phx_get_state / phx_set_state / phx_trace / phx_pack_flags are exercised through the same
signatures.  Test infrastructure: the product never loads this library."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import env_from_golden, f32_bits, f64_bits, golden, supply_chain_env
from phantom_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, "oracle", "libphantom_cpu.so")


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libphantom_cpu.so"])
    return _abi.bind_signatures(C.CDLL(CPU_LIB))


def _p(a):
    return None if a is None else a.ctypes.data


class CpuAbiRunner:
    """the same calls DeviceEnv makes, with host numpy buffers"""

    def __init__(self, lib, spec):
        self.lib, self.spec = lib, spec
        self.cs, self._keep = spec.to_ctypes()
        cs = C.byref(self.cs)
        self.B, self.S = spec.batch, lib.phx_n_strategic(cs)
        self.D, self.n_exo = lib.phx_obs_dim(cs), lib.phx_n_exo(cs)
        n = lib.phx_state_nbytes(cs)
        assert n > 0
        self.blob = np.zeros(n, np.uint8)
        h = C.c_void_p()
        assert lib.phx_create(cs, 0, _p(self.blob), n, C.byref(h)) == 0, lib.phx_last_error()
        self.h = h
        B, S, D = self.B, max(self.S, 1), self.D
        self.obs = np.zeros((B, S, D), np.float32); self.reward = np.zeros((B, S), np.float64)
        self.obs_valid = np.zeros((B, S), np.uint8); self.reward_valid = np.zeros((B, S), np.uint8)
        self.terminated = np.zeros((B, S), np.uint8); self.truncated = np.zeros((B, S), np.uint8)
        self.done_valid = np.zeros((B, S), np.uint8)
        self.all_terminated = np.zeros(B, np.uint8); self.all_truncated = np.zeros(B, np.uint8)
        self.err = np.zeros(B, np.int32)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.phx_destroy(self.h)

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        assert self.lib.phx_reset(self.h, _p(m), None, None, _p(self.obs), _p(self.obs_valid), None) == 0
        return self.obs.copy(), self.obs_valid.copy()

    def step(self, actions, exo=None):
        io = _abi.PhxStepIO()
        self._a = np.ascontiguousarray(actions, np.float32)
        self._x = None if exo is None else np.ascontiguousarray(exo, np.uint8)
        io.actions, io.exo = _p(self._a), _p(self._x)
        io.obs, io.obs_valid, io.reward, io.reward_valid = _p(self.obs), _p(self.obs_valid), _p(self.reward), _p(self.reward_valid)
        io.terminated, io.truncated, io.done_valid = _p(self.terminated), _p(self.truncated), _p(self.done_valid)
        io.all_terminated, io.all_truncated, io.err = _p(self.all_terminated), _p(self.all_truncated), _p(self.err)
        assert self.lib.phx_step(self.h, C.byref(io), None) == 0

    def get_i32(self, field, shape):
        out = np.zeros(shape, np.int32)
        assert self.lib.phx_get_state(self.h, field.encode(), _p(out), out.nbytes, None) == out.nbytes
        return out


def test_every_declared_symbol_is_exported_with_the_products_signatures(lib):
    for name in _abi.EXPORTS:
        assert hasattr(lib, name), name
    assert lib.phx_abi_version() == _abi.ABI_VERSION and b"restatement" in lib.phx_last_kernel()


@pytest.mark.parametrize("name", ["sc7_fixed20", "sc64", "sc256_fsm"])
def test_reference_goldens_through_the_identical_stub(lib, name):
    g = golden(name)
    T = int(g["T"])
    env = env_from_golden(g)
    run = CpuAbiRunner(lib, env.spec)
    for t in range(T):
        rb = g["reset_before"][t]
        if rb.any():
            obs, valid = run.reset(rb)
            m = rb.astype(bool)
            np.testing.assert_array_equal(valid[m], g["reset_obs_valid"][t][m])
        run.step(g["actions"][t], g["exo"][t])
        assert (run.err == 0).all()
        S = run.S
        np.testing.assert_array_equal(run.get_i32("shop.stock", (run.B, S)), g["stock"][t], err_msg=f"stock t={t}")
        np.testing.assert_array_equal(run.obs_valid, g["obs_valid"][t])
        ov = g["obs_valid"][t].astype(bool)
        np.testing.assert_array_equal(f32_bits(run.obs[ov]), f32_bits(g["obs"][t][ov]), err_msg=f"obs t={t}")
        rv = g["reward_valid"][t] == 1
        np.testing.assert_array_equal(f64_bits(run.reward[rv]), f64_bits(g["reward"][t][rv]))
        np.testing.assert_array_equal(run.all_truncated, g["all_truncated"][t])


def test_rollout_state_trace_and_flag_packing_through_the_abi(lib):
    B, S, T = 6, 3, 25
    env = supply_chain_env(S, [2] * S, 10, B, seed=5, tracking=True, force_generic=True)
    run = CpuAbiRunner(lib, env.spec)
    run.reset()
    io = _abi.PhxRolloutIO()
    io.T = T
    bufs = dict(obs=np.zeros((T, B, S, 3), np.float32), act=np.zeros((T, B, S), np.float32), rew=np.zeros((T, B, S), np.float32),
                ter=np.zeros((T, B, S), np.uint8), tru=np.zeros((T, B, S), np.uint8), last=np.zeros((B, S, 3), np.float32),
                log=np.zeros((T, B, env.spec.trace_cap, 16), np.uint8), cnt=np.zeros((T, B), np.int32))
    io.obs, io.action_out, io.reward, io.terminated, io.truncated = (_p(bufs[k]) for k in ("obs", "act", "rew", "ter", "tru"))
    io.last_obs, io.err, io.msg_log, io.msg_count = _p(bufs["last"]), _p(run.err), _p(bufs["log"]), _p(bufs["cnt"])
    assert lib.phx_rollout(run.h, C.byref(io), None) == 0
    assert bufs["tru"].sum() == 2 * B * S and (bufs["cnt"] > 0).all()
    stock = run.get_i32("shop.stock", (B, S))
    assert ((stock >= 0) & (stock <= 100)).all()
    new = ((np.arange(B * S, dtype=np.int32).reshape(B, S) * 7) % 90).astype(np.int32)
    assert lib.phx_set_state(run.h, b"shop.stock", _p(new), new.nbytes, None) == new.nbytes
    np.testing.assert_array_equal(run.get_i32("shop.stock", (B, S)), new)
    assert lib.phx_get_state(run.h, b"no.such.field", _p(new), new.nbytes, None) < 0
    recs = (_abi.PhxMsgRec * env.spec.trace_cap)()
    n = lib.phx_trace(run.h, _p(bufs["log"][T - 1]), _p(bufs["cnt"][T - 1]), 2, recs, env.spec.trace_cap, None)
    assert n == int(bufs["cnt"][T - 1, 2]) and recs[0].type != 0
    x = (np.random.default_rng(0).random(1003) < 0.3).astype(np.uint8) * 5
    packed = np.zeros((1003 + 63) // 64, np.uint64)
    assert lib.phx_pack_flags(_p(x), _p(packed), x.size, None) == 0
    np.testing.assert_array_equal(packed.view(np.uint8)[:126], np.packbits(x != 0, bitorder="little"))
    back = np.zeros(1003, np.uint8)
    assert lib.phx_unpack_flags(_p(packed), _p(back), x.size, None) == 0
    np.testing.assert_array_equal(back, (x != 0).astype(np.uint8))


# ---- ABI 7: per-env legacy-numpy MT19937 streams -----------------------------------------------------------------
def test_mt19937_streams_equal_numpy_and_the_draws_the_reference_consumed(lib):
    """phx_mt_seed / phx_mt_draw (sequential statement in oracle/phx_cpu_abi.c) against numpy itself -- np.random.seed(s) followed by
    np.random.randint(5) calls, continued over several phx_mt_draw calls and over state regenerations -- and against the
    draws the REFERENCE's CustomerAgents consumed in the seeded golden run sc64 (tests/golden/gen_goldens.py: one env alone on the
    global stream after np.random.seed(seeds[b]))."""
    g = golden("sc64")
    seeds = [int(s) for s in g["seeds"]]
    env = supply_chain_env(9, [6] * 9, 100, len(seeds), exogenous="mt19937")
    r = CpuAbiRunner(lib, env.spec)
    s32 = np.asarray(seeds, np.uint32)
    assert lib.phx_mt_seed(r.h, _p(s32), None) == 0
    T = g["exo"].shape[0]
    got = np.zeros((T, len(seeds), 54), np.uint8)
    for t0, t1 in ((0, 1), (1, 2), (2, 40), (40, T)):           # several calls: the stream position persists
        part = np.zeros((t1 - t0, len(seeds), 54), np.uint8)
        assert lib.phx_mt_draw(r.h, _p(part), t1 - t0, None) == 0
        got[t0:t1] = part
    for b, s in enumerate(seeds):
        want = np.random.RandomState(s).randint(5, size=T * 54).astype(np.uint8).reshape(T, 54)
        np.testing.assert_array_equal(got[:, b], want, err_msg=f"numpy stream of seed {s}")
    np.testing.assert_array_equal(got, g["exo"])                # what the reference consumed
    # a spec without the flag refuses
    r2 = CpuAbiRunner(lib, supply_chain_env(2, [3, 3], 10, 2).spec)
    assert lib.phx_mt_draw(r2.h, _p(np.zeros((1, 2, 6), np.uint8)), 1, None) == -2        # PHX_EUNSUPPORTED


@pytest.mark.parametrize("name", ["sc_fsm_small", "sc256_fsm"])
def test_mt19937_streams_on_an_fsm_env_draw_for_the_stage_s_customers_only(lib, name):
    """FSM supply chains: only the customers of the env's stage call np.random.randint (fsm.py:276-279 -> supply_chain.py:64).
    Draws made step by step beside the steps equal the ones the REFERENCE consumed in the seeded golden run, the run itself
    equals the golden, and ONE phx_mt_draw(T) -- which walks the stages forward itself -- equals the step-by-step draws."""
    g = golden(name)
    seeds = np.asarray([int(s) for s in g["seeds"]], np.uint32)
    B, T, n = len(seeds), int(g["actions"].shape[0]), int(g["exo"].shape[2])
    r = CpuAbiRunner(lib, env_from_golden(g, exogenous="mt19937").spec)
    assert lib.phx_mt_seed(r.h, _p(seeds), None) == 0
    r.reset()
    for t in range(T):
        if t > 0 and g["reset_before"][t].any():
            r.reset(g["reset_before"][t])
        exo = np.zeros((1, B, n), np.uint8)
        assert lib.phx_mt_draw(r.h, _p(exo), 1, None) == 0
        np.testing.assert_array_equal(exo[0], g["exo"][t], err_msg=f"draws t={t}")
        r.step(g["actions"][t], exo[0])
        np.testing.assert_array_equal(r.get_i32("shop.stock", (B, int(g["n_shops"]))), g["stock"][t], err_msg=f"stock t={t}")
    # one call for the whole run from a fresh env: the stage walk (and the reset at the episode's end) is the kernel's own
    if not g["reset_before"][1:].any() or int(g["num_steps"]) > 0:
        r2 = CpuAbiRunner(lib, env_from_golden(g, exogenous="mt19937").spec)
        assert lib.phx_mt_seed(r2.h, _p(seeds), None) == 0
        r2.reset()
        allx = np.zeros((T, B, n), np.uint8)
        assert lib.phx_mt_draw(r2.h, _p(allx), T, None) == 0
        np.testing.assert_array_equal(allx, g["exo"])


def test_fragment_list_through_the_abi_equals_one_long_rollout(lib):
    """ABI 9, phx_rollout_io.frags: k fragments of Tf rows == rows [i Tf, (i + 1) Tf) of ONE k Tf-step rollout from the same state;
    malformed lists are refused."""
    B, S, Tf, k = 5, 3, 7, 3
    env = supply_chain_env(S, [2] * S, 10, B, seed=8)
    one, lst = CpuAbiRunner(lib, env.spec), CpuAbiRunner(lib, env.spec)
    one.reset(); lst.reset()
    T = Tf * k
    mk = lambda n: dict(obs=np.zeros((n, B, S, 3), np.float32), act=np.zeros((n, B, S), np.float32), rew=np.zeros((n, B, S), np.float32),
                        ter=np.full((n, B, S), 9, np.uint8), tru=np.full((n, B, S), 9, np.uint8))
    whole, last_a, last_b = mk(T), np.zeros((B, S, 3), np.float32), np.zeros((B, S, 3), np.float32)
    io = _abi.PhxRolloutIO()
    io.T = T
    io.obs, io.action_out, io.reward, io.terminated, io.truncated = (_p(whole[n]) for n in ("obs", "act", "rew", "ter", "tru"))
    io.last_obs, io.err = _p(last_a), _p(one.err)
    assert lib.phx_rollout(one.h, C.byref(io), None) == 0
    parts = [mk(Tf) for _ in range(k)]
    arr = (_abi.PhxRolloutFrag * k)()
    for i, p in enumerate(parts):
        arr[i].obs, arr[i].action_out, arr[i].reward, arr[i].terminated, arr[i].truncated = (_p(p[n]) for n in ("obs", "act", "rew", "ter", "tru"))
    io2 = _abi.PhxRolloutIO()
    io2.T, io2.n_frag, io2.frags = T, k, C.cast(arr, C.c_void_p)
    io2.last_obs, io2.err = _p(last_b), _p(lst.err)
    assert lib.phx_rollout(lst.h, C.byref(io2), None) == 0
    for i, p in enumerate(parts):
        for n in ("obs", "act", "rew", "ter", "tru"):
            np.testing.assert_array_equal(p[n].view(np.uint8), whole[n][i * Tf:(i + 1) * Tf].view(np.uint8), err_msg=f"fragment {i} {n}")
    np.testing.assert_array_equal(f32_bits(last_a), f32_bits(last_b))
    np.testing.assert_array_equal(one.get_i32("shop.stock", (B, S)), lst.get_i32("shop.stock", (B, S)))
    io2.T = T + 1                                              # not a multiple of n_frag
    assert lib.phx_rollout(lst.h, C.byref(io2), None) < 0
    io2.T, io2.obs = T, _p(whole["obs"])                       # the io's own planes beside a list
    assert lib.phx_rollout(lst.h, C.byref(io2), None) < 0


def test_replay_hints_are_accepted_and_unknown_hint_bits_refused(lib):
    """phx_rollout_io.hints (ABI 9): PHX_RH_ACTIONS_IN_DOMAIN / PHX_RH_EXO_IN_DOMAIN are the caller's word about replayed inputs -- they
    change no result (here: the CPU stub, which has no byte tiles) -- and any other bit (bit 1 was PHX_RH_FLAGS_ZEROED until ABI 8) is an
    argument error, as in the product."""
    B, S, T = 4, 3, 9
    env = supply_chain_env(S, [2] * S, 5, B, seed=2)
    rng = np.random.default_rng(1)
    acts = rng.uniform(0, 100, (T, B, S)).astype(np.float32)
    exo = rng.integers(0, 5, (T, B, env.spec.n_exo)).astype(np.uint8)
    outs = []
    for hints in (0, _abi.RH_ACTIONS_IN_DOMAIN | _abi.RH_EXO_IN_DOMAIN):
        r = CpuAbiRunner(lib, env.spec); r.reset()
        obs, act, rew = np.zeros((T, B, S, 3), np.float32), np.zeros((T, B, S), np.float32), np.zeros((T, B, S), np.float32)
        ter, tru, last = np.zeros((T, B, S), np.uint8), np.zeros((T, B, S), np.uint8), np.zeros((B, S, 3), np.float32)
        io = _abi.PhxRolloutIO()
        io.T, io.hints = T, hints
        io.actions, io.exo = _p(acts), _p(exo)
        io.obs, io.action_out, io.reward, io.terminated, io.truncated, io.last_obs, io.err = _p(obs), _p(act), _p(rew), _p(ter), _p(tru), _p(last), _p(r.err)
        assert lib.phx_rollout(r.h, C.byref(io), None) == 0
        outs.append((obs, rew, tru))
        io.hints = hints | 1
        assert lib.phx_rollout(r.h, C.byref(io), None) < 0
        io.hints = 8
        assert lib.phx_rollout(r.h, C.byref(io), None) < 0
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a.view(np.uint8), b.view(np.uint8))

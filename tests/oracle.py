"""ctypes wrapper of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Takes the same ``EnvSpec`` the product compiles, so a parity test builds ONE env description
and runs it through both the oracle (host numpy buffers) and the HIP path (device tensors).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from phantom_amd import _abi
from phantom_amd.message import payload_to_record

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

_lib = None


def build_oracle(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("phx_oracle.c", "phx_oracle.h")]
    src.append(os.path.join(ROOT, "include", "phantom_amd.h"))
    if force or not os.path.exists(LIB_PATH) or \
            any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "liboracle.so"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.phxo_create.restype = vp
        L.phxo_create.argtypes = [C.POINTER(_abi.PhxSpec)]
        L.phxo_destroy.argtypes = [vp]
        L.phxo_last_error.restype = C.c_char_p
        L.phxo_set_threads.argtypes = [C.c_int]
        L.phxo_max_threads.restype = C.c_int
        L.phxo_set_f64.restype = C.c_int64
        L.phxo_set_f64.argtypes = [vp, C.c_char_p, vp]
        L.phxo_check_recip_div.restype = C.c_int64
        L.phxo_check_recip_div.argtypes = [C.c_int, C.c_int]
        for n in ("phxo_obs_dim", "phxo_n_strategic", "phxo_n_exo"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [vp]
        L.phxo_reset.argtypes = [vp, vp, vp, vp, vp, vp]
        L.phxo_get_u8.restype = C.c_int64
        L.phxo_get_u8.argtypes = [vp, C.c_char_p, vp]
        L.phxo_rng_uniform.restype = C.c_double
        L.phxo_rng_uniform.argtypes = [C.c_uint64, C.c_int64, C.c_uint32, C.c_int, vp]
        for _n in ("phxo_step", "phxo_step_begin", "phxo_step_end"):
            getattr(L, _n).argtypes = [vp, C.POINTER(_abi.PhxStepIO)]
        L.phxo_inject.argtypes = [vp, C.POINTER(_abi.PhxMsgRec), C.c_int]
        L.phxo_resolve.argtypes = [vp, vp, vp, vp]
        L.phxo_rollout.argtypes = [vp, C.POINTER(_abi.PhxRolloutIO)]
        for n in ("phxo_get_i32", "phxo_get_f64", "phxo_set_i32"):
            getattr(L, n).restype = C.c_int64
            getattr(L, n).argtypes = [vp, C.c_char_p, vp]
        L.phxo_philox4x32_10.argtypes = [vp, vp, vp]
        L.phxo_rng_orders.argtypes = [C.c_uint64, C.c_int64, C.c_uint32, C.c_int, C.c_int, vp]
        L.phxo_rng_action.restype = C.c_float
        L.phxo_rng_action.argtypes = [C.c_uint64, C.c_int64, C.c_uint32, C.c_int]
        _lib = L
    return _lib


LOG_DTYPE = np.dtype([("sender", "<u2"), ("receiver", "<u2"), ("type", "<u2"), ("round", "<u2"),
                      ("raw", "<i8")])


def _p(a):
    return None if a is None else a.ctypes.data


class OracleEnv:
    """One batch of env instances stepped by the sequential C restatement."""

    def __init__(self, spec, threads=1):
        self.L = lib()
        self.spec = spec
        cs, self._keep = spec.to_ctypes()
        self.h = self.L.phxo_create(C.byref(cs))
        if not self.h:
            raise RuntimeError("phxo_create: " + self.L.phxo_last_error().decode())
        self.L.phxo_set_threads(threads)
        self.B, self.S = spec.batch, self.L.phxo_n_strategic(self.h)
        self.D, self.n_exo = self.L.phxo_obs_dim(self.h), self.L.phxo_n_exo(self.h)
        B, S, D = self.B, max(self.S, 1), self.D
        self.obs = np.zeros((B, S, D), np.float32)
        self.reward = np.zeros((B, S), np.float64)
        self.obs_valid = np.zeros((B, S), np.uint8)
        self.reward_valid = np.zeros((B, S), np.uint8)
        self.terminated = np.zeros((B, S), np.uint8)
        self.truncated = np.zeros((B, S), np.uint8)
        self.done_valid = np.zeros((B, S), np.uint8)
        self.all_terminated = np.zeros(B, np.uint8)
        self.all_truncated = np.zeros(B, np.uint8)
        self.err = np.zeros(B, np.int32)
        cap = max(spec.trace_cap, 1)
        self.msg_log = np.zeros((B, cap), LOG_DTYPE) if spec.trace_cap else None
        self.msg_count = np.zeros(B, np.int32) if spec.trace_cap else None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.phxo_destroy(self.h)
            self.h = None

    def set_threads(self, n):
        self.L.phxo_set_threads(n)

    def reset(self, mask=None, sampler_values=None, conn_on=None):
        if conn_on is not None:
            conn_on = np.ascontiguousarray(conn_on, np.uint8)
            assert conn_on.shape == (self.B, self.spec.n_conn)
        if sampler_values is not None:
            sampler_values = np.ascontiguousarray(sampler_values, np.float64)
            assert sampler_values.shape == (self.B, self.spec.n_samplers)
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
            self.err[mask.astype(bool)] = 0
        else:
            self.err[:] = 0
        self.L.phxo_reset(self.h, _p(mask), _p(sampler_values), _p(conn_on), _p(self.obs), _p(self.obs_valid))
        return self.obs.copy(), self.obs_valid.copy()

    def step_begin(self, actions, action_valid=None, exo=None, shuffle=None):
        """phxo_step_begin: acting phase + resolve_network() (fsm.py:275-280); the outputs of the step are not touched"""
        return self.step(actions, action_valid, exo, shuffle, None, _fn="phxo_step_begin")

    def step_end(self, next_stage=None):
        """phxo_step_end: the handler's stages -> transition, observations, rewards, done flags (fsm.py:304-380)"""
        return self.step(None, None, None, None, next_stage, _fn="phxo_step_end")

    def step(self, actions, action_valid=None, exo=None, shuffle=None, next_stage=None, _fn="phxo_step"):
        io = _abi.PhxStepIO()
        self._ns = np.ascontiguousarray(next_stage, np.int32) if next_stage is not None else None
        io.next_stage = _p(self._ns)
        self._sh = np.ascontiguousarray(shuffle, np.uint16) if shuffle is not None else None
        if self._sh is not None:
            assert self._sh.shape == (self.B, 8 * self.spec.queue_cap)
        io.shuffle = _p(self._sh)
        self._a = np.ascontiguousarray(actions, np.float32) if actions is not None else None
        self._av = np.ascontiguousarray(action_valid, np.uint8) if action_valid is not None else None
        self._x = np.ascontiguousarray(exo, np.uint8) if exo is not None else None
        io.actions, io.action_valid, io.exo = _p(self._a), _p(self._av), _p(self._x)
        io.obs, io.obs_valid = _p(self.obs), _p(self.obs_valid)
        io.reward, io.reward_valid = _p(self.reward), _p(self.reward_valid)
        io.terminated, io.truncated, io.done_valid = (_p(self.terminated), _p(self.truncated),
                                                      _p(self.done_valid))
        io.all_terminated, io.all_truncated = _p(self.all_terminated), _p(self.all_truncated)
        io.err = _p(self.err)
        io.msg_log, io.msg_count = _p(self.msg_log), _p(self.msg_count)
        getattr(self.L, _fn)(self.h, C.byref(io))
        return self

    def inject(self, messages):
        arr = (_abi.PhxMsgRec * max(len(messages), 1))()
        for k, m in enumerate(messages):
            t, is_f, v = payload_to_record(m.payload)
            arr[k].sender, arr[k].receiver, arr[k].type = (self.spec.index_of(m.sender_id),
                                                           self.spec.index_of(m.receiver_id), t)
            if is_f:
                arr[k].payload.f = float(v)
            else:
                arr[k].payload.i = int(v)
        self.L.phxo_inject(self.h, arr, len(messages))

    def resolve(self):
        self.err[:] = 0
        self.L.phxo_resolve(self.h, _p(self.err), _p(self.msg_log), _p(self.msg_count))

    def rollout(self, T, actions=None, exo=None, record_messages=False, policy=None):
        B, S, D = self.B, self.S, self.D
        out = dict(obs=np.zeros((T, B, S, D), np.float32), actions=np.zeros((T, B, S), np.float32),
                   rewards=np.zeros((T, B, S), np.float32), terminated=np.zeros((T, B, S), np.uint8),
                   truncated=np.zeros((T, B, S), np.uint8), last_obs=np.zeros((B, S, D), np.float32),
                   obs_valid=np.zeros((T, B, S), np.uint8), reward_valid=np.zeros((T, B, S), np.uint8))
        io = _abi.PhxRolloutIO()
        io.T = T
        self._a = np.ascontiguousarray(actions, np.float32) if actions is not None else None
        self._x = np.ascontiguousarray(exo, np.uint8) if exo is not None else None
        io.actions, io.exo = _p(self._a), _p(self._x)
        io.obs, io.action_out, io.reward = _p(out["obs"]), _p(out["actions"]), _p(out["rewards"])
        io.terminated, io.truncated = _p(out["terminated"]), _p(out["truncated"])
        io.obs_valid, io.reward_valid = _p(out["obs_valid"]), _p(out["reward_valid"])
        io.last_obs, io.err = _p(out["last_obs"]), _p(self.err)
        if policy is not None:                                  # phantom_amd.policy.MLPPolicy: the restatement reads the host copies of the weights
            self._pol = policy.host_struct()
            io.policy = C.addressof(self._pol)
        if record_messages:
            out["msg_log"] = np.zeros((T, B, self.spec.trace_cap), LOG_DTYPE)
            out["msg_count"] = np.zeros((T, B), np.int32)
            io.msg_log, io.msg_count = _p(out["msg_log"]), _p(out["msg_count"])
        self.L.phxo_rollout(self.h, C.byref(io))
        return out

    def get_i32(self, field):
        buf = np.zeros(self.B * max(self.spec.n_agents, 1) * 3, np.int32)
        n = self.L.phxo_get_i32(self.h, field.encode(), _p(buf))
        assert n >= 0, field
        return buf[:n].reshape(self.B, -1).copy()

    def set_i32(self, field, arr):
        arr = np.ascontiguousarray(arr, np.int32)
        assert self.L.phxo_set_i32(self.h, field.encode(), _p(arr)) >= 0

    def set_f64(self, field, arr):
        arr = np.ascontiguousarray(arr, np.float64)
        assert self.L.phxo_set_f64(self.h, field.encode(), _p(arr)) >= 0

    def get_u8(self, field):
        buf = np.zeros(self.B * max(self.spec.n_conn, 1), np.uint8)
        n = self.L.phxo_get_u8(self.h, field.encode(), _p(buf))
        assert n >= 0, field
        return buf[:n].reshape(self.B, -1).copy()

    def get_f64(self, field):
        buf = np.zeros(self.B * max(self.spec.n_agents, len(self.spec.col), self.spec.n_samplers, 1), np.float64)
        n = self.L.phxo_get_f64(self.h, field.encode(), _p(buf))
        assert n >= 0, field
        return buf[:n].reshape(self.B, -1).copy()

    def log(self, b=0):
        n = int(self.msg_count[b])
        return self.msg_log[b, :n].copy()


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    out = np.zeros(4, np.uint32)
    lib().phxo_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def rng_orders(seed, genv, tick, shop, K):
    out = np.zeros(K, np.uint8)
    lib().phxo_rng_orders(seed, genv, tick, shop, K, _p(out))
    return out


def rng_action(seed, genv, tick, r):
    return float(lib().phxo_rng_action(seed, genv, tick, r))

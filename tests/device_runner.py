"""Adapter giving the HIP path (phantom_amd.device.DeviceEnv, through the C ABI) the same
numpy-facing interface as tests/oracle.py's OracleEnv, so one test body drives both."""
import numpy as np
import torch

from oracle import LOG_DTYPE
from phantom_amd.device import DeviceEnv


class DeviceRunner:
    def __init__(self, spec):
        self.dev = DeviceEnv(spec)
        self.spec = spec
        self.B, self.S, self.D, self.n_exo = self.dev.B, self.dev.S, self.dev.D, self.dev.n_exo
        self.err = np.zeros(self.B, np.int32)
        self._pending = []

    def _t(self, a, dtype):
        return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype)).to(self.dev.device)

    def _pull(self):
        d = self.dev
        self.obs, self.reward = d.obs.cpu().numpy(), d.reward.cpu().numpy()
        self.obs_valid, self.reward_valid = d.obs_valid.cpu().numpy(), d.reward_valid.cpu().numpy()
        self.terminated, self.truncated = d.terminated.cpu().numpy(), d.truncated.cpu().numpy()
        self.done_valid = d.done_valid.cpu().numpy()
        self.all_terminated, self.all_truncated = d.all_terminated.cpu().numpy(), d.all_truncated.cpu().numpy()
        self.err = d.err.cpu().numpy()
        self.msg_count = d.msg_count.cpu().numpy() if d.msg_count is not None else None

    def reset(self, mask=None, sampler_values=None, conn_on=None):
        obs, valid = self.dev.reset(mask, sampler_values, conn_on)
        self.err = self.dev.err.cpu().numpy()
        return obs.cpu().numpy(), valid.cpu().numpy()

    def step(self, actions, action_valid=None, exo=None, shuffle=None, next_stage=None):
        if actions is None:
            actions = np.zeros((self.B, max(self.S, 1)), np.float32)
        sh = None if shuffle is None else torch.from_numpy(np.ascontiguousarray(shuffle, np.uint16).view(np.int16)).to(self.dev.device)
        self.dev.step(self._t(actions, np.float32), self._t(action_valid, np.uint8),
                      self._t(exo, np.uint8), sh, self._t(next_stage, np.int32))
        self._pull()
        return self

    def step_begin(self, actions, action_valid=None, exo=None, shuffle=None):
        sh = None if shuffle is None else torch.from_numpy(np.ascontiguousarray(shuffle, np.uint16).view(np.int16)).to(self.dev.device)
        self.dev.step_begin(self._t(actions, np.float32), self._t(action_valid, np.uint8), self._t(exo, np.uint8), sh)
        self.err = self.dev.err.cpu().numpy()
        return self

    def step_end(self, next_stage=None):
        self.dev.step_end(self._t(next_stage, np.int32))
        self._pull()
        return self

    def inject(self, messages):
        self._pending.extend(messages)

    def resolve(self):
        d = self.dev
        d.inject(self._pending)
        self._pending = []
        d.err.zero_()
        import ctypes as C
        lp = d.msg_log.data_ptr() if d.msg_log is not None else None
        cp = d.msg_count.data_ptr() if d.msg_count is not None else None
        d._check(d.lib.phx_resolve(d.handle, d.err.data_ptr(), lp, cp, d._stream()), "phx_resolve")
        self.err = d.err.cpu().numpy()
        self.msg_count = d.msg_count.cpu().numpy() if d.msg_count is not None else None

    def rollout(self, T, actions=None, exo=None, **hints):        # (hints: actions_in_domain / exo_in_domain / policy)
        a = None if actions is None else self._t(actions, np.float32)
        x = None if exo is None else self._t(exo, np.uint8)
        tr = self.dev.rollout(T, a, x, **hints)
        if getattr(self, "on_launch", None) is not None:        # (fuzz campaigns: the kernels of the launch are journalled BEFORE the host waits for them)
            self.on_launch(self.dev.last_kernel())
        self.err = self.dev.err.cpu().numpy()
        return dict(obs=tr.observations.cpu().numpy(), actions=tr.actions.cpu().numpy(),
                    rewards=tr.rewards.cpu().numpy(), terminated=tr.terminations.cpu().numpy(),
                    truncated=tr.truncations.cpu().numpy(), last_obs=tr.last_obs.cpu().numpy(),
                    obs_valid=None if tr.obs_valid is None else tr.obs_valid.cpu().numpy(),
                    reward_valid=None if tr.reward_valid is None else tr.reward_valid.cpu().numpy())

    def rollout_fragments(self, Tf, k, actions=None, exo=None, **hints):
        """k fragments of Tf rows from ONE phx_rollout call (phx_rollout_io.frags), concatenated like one k Tf-step rollout"""
        a = None if actions is None else self._t(actions, np.float32)
        x = None if exo is None else self._t(exo, np.uint8)
        outs = [self.dev.alloc_trajectory(Tf) for _ in range(k)]
        self.dev.rollout_fragments(Tf, outs, a, x, **hints)
        if getattr(self, "on_launch", None) is not None:
            self.on_launch(self.dev.last_kernel())
        self.err = self.dev.err.cpu().numpy()
        cat = lambda f: None if getattr(outs[0], f) is None else torch.cat([getattr(o, f) for o in outs]).cpu().numpy()
        return dict(obs=cat("observations"), actions=cat("actions"), rewards=cat("rewards"), terminated=cat("terminations"),
                    truncated=cat("truncations"), last_obs=outs[-1].last_obs.cpu().numpy(), obs_valid=cat("obs_valid"),
                    reward_valid=cat("reward_valid"))

    def get_i32(self, field):
        return self.dev.field(field).cpu().numpy().reshape(self.B, -1)

    def set_i32(self, field, arr):
        self.dev.field(field).copy_(torch.from_numpy(np.ascontiguousarray(arr, np.int32)).reshape(
            self.dev.field(field).shape))

    def set_f64(self, field, arr):
        self.dev.field(field).copy_(torch.from_numpy(np.ascontiguousarray(arr, np.float64)).reshape(self.dev.field(field).shape))

    def get_u8(self, field):
        return self.dev.field(field).cpu().numpy().reshape(self.B, -1)

    def get_f64(self, field):
        v = self.dev.field(field).cpu().numpy()
        if field == "buyer.prices":
            # device layout is slot-major [B, dmax, n_buyers]; the oracle reports buyers in agent
            # order with their deg slots each
            from phantom_amd import _abi
            sp = self.spec
            deg = np.diff(sp.row_ptr)
            buyers = [a for a in range(sp.n_agents) if sp.kind[a] == _abi.KIND_BUYER]
            cols = [v[:, :deg[a], r] for r, a in enumerate(buyers)]
            return np.concatenate(cols, axis=1) if cols else v.reshape(self.B, -1)
        return v.reshape(self.B, -1)

    def log(self, b=0):
        n = int(self.msg_count[b])
        assert n <= self.spec.trace_cap
        raw = self.dev.msg_log[b, :n].cpu().numpy().tobytes()
        return np.frombuffer(raw, dtype=LOG_DTYPE).copy()

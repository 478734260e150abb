"""A policy evaluated on the device inside the fused rollout (``phx_rollout_io.policy``, ABI 10).

The reference's collection loop calls a policy for every agent and step (utils/rllib/rollout.py:300-363).  ``MLPPolicy`` is the
small network the library can evaluate itself, one lane per (env, shop): ``DeviceEnv.rollout(T, policy=pol)`` is ONE launch for T
on-policy steps.  Its arithmetic is defined in include/phantom_amd.h (f32, fused multiply-adds term by term in ascending order);
``__call__`` evaluates the same network with torch ops (the same function up to the order of the additions)."""
import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _abi

ACTIVATIONS = {"relu": _abi.ACT_RELU, "hard_tanh": _abi.ACT_HARD_TANH}


class MLPPolicy:
    """``weights`` / ``biases``: torch.nn.Linear's own layouts -- [H0, D], ([H1, H0],) [1, H_last] and [H0], ([H1],) [1]; one or two
    hidden layers of at most 64 units; ``activation`` "relu" or "hard_tanh" (clip to [-1, 1]).  The scalar output y becomes the action
    ``clip(out_scale * y + out_bias, out_lo, out_hi)`` (ShopAgent's action space is Box(0, 100): out_lo >= 0)."""

    def __init__(self, weights: Sequence, biases: Sequence, activation: str = "relu", out_scale: float = 1.0, out_bias: float = 0.0,
                 out_lo: float = 0.0, out_hi: float = 100.0):
        if activation not in ACTIVATIONS:
            raise ValueError(f"activation {activation!r}: one of {sorted(ACTIVATIONS)}")
        self.weights = [np.ascontiguousarray(_np(w), np.float32) for w in weights]
        self.biases = [np.ascontiguousarray(_np(b), np.float32).reshape(-1) for b in biases]
        n = len(self.weights)
        if n not in (2, 3) or len(self.biases) != n:
            raise ValueError("MLPPolicy: one or two hidden layers (2 or 3 weight matrices and as many biases)")
        for l in range(n):
            w, b = self.weights[l], self.biases[l]
            if w.ndim != 2 or b.shape != (w.shape[0],) or (l > 0 and w.shape[1] != self.weights[l - 1].shape[0]):
                raise ValueError(f"MLPPolicy: layer {l} has weight {w.shape} and bias {b.shape}")
            if l < n - 1 and not 1 <= w.shape[0] <= _abi.POLICY_MAX_WIDTH:
                raise ValueError(f"MLPPolicy: hidden widths 1 .. {_abi.POLICY_MAX_WIDTH}")
            if not (np.isfinite(w).all() and np.isfinite(b).all()):
                raise ValueError("MLPPolicy: weights must be finite")
        if self.weights[-1].shape[0] != 1:
            raise ValueError("MLPPolicy: the output layer has one unit (the agent's scalar action)")
        if not 0.0 <= out_lo <= out_hi:
            raise ValueError("MLPPolicy: 0 <= out_lo <= out_hi")
        self.activation = activation
        self.out_scale, self.out_bias, self.out_lo, self.out_hi = float(out_scale), float(out_bias), float(out_lo), float(out_hi)
        self._dev = {}

    @property
    def obs_dim(self) -> int:
        return int(self.weights[0].shape[1])

    @classmethod
    def from_torch(cls, module, **kw) -> "MLPPolicy":
        """from a torch.nn.Sequential of Linear layers with ReLU / Hardtanh between them"""
        import torch
        lin = [m for m in module.modules() if isinstance(m, torch.nn.Linear)]
        acts = [m for m in module.modules() if isinstance(m, (torch.nn.ReLU, torch.nn.Hardtanh))]
        act = "hard_tanh" if acts and isinstance(acts[0], torch.nn.Hardtanh) else "relu"
        return cls([m.weight.detach() for m in lin], [m.bias.detach() for m in lin], activation=act, **kw)

    def update(self, weights: Sequence, biases: Sequence) -> None:
        """new parameter values of the same shapes (a learner's update): the device copies are refreshed in place, cached argument
        blocks stay valid"""
        import torch
        for l, (w, b) in enumerate(zip(weights, biases)):
            w, b = np.ascontiguousarray(_np(w), np.float32), np.ascontiguousarray(_np(b), np.float32).reshape(-1)
            if w.shape != self.weights[l].shape or b.shape != self.biases[l].shape:
                raise ValueError("MLPPolicy.update: shapes differ from the policy's")
            self.weights[l], self.biases[l] = w, b
            for dev, (ws, bs, _) in self._dev.items():
                ws[l].copy_(torch.from_numpy(w)); bs[l].copy_(torch.from_numpy(b))

    def on(self, device):
        """(device weight tensors, device bias tensors, the phx_policy_mlp argument) for ``device``"""
        import torch
        key = str(device)
        if key not in self._dev:
            ws = [torch.from_numpy(w).to(device).contiguous() for w in self.weights]
            bs = [torch.from_numpy(b).to(device).contiguous() for b in self.biases]
            self._dev[key] = (ws, bs, self._c_struct([w.data_ptr() for w in ws], [b.data_ptr() for b in bs]))
        return self._dev[key]

    def _c_struct(self, wptrs, bptrs) -> "_abi.PhxPolicyMLP":
        p = _abi.PhxPolicyMLP()
        n = len(self.weights)
        p.n_hidden = n - 1
        p.width[0] = self.weights[0].shape[0]
        p.width[1] = self.weights[1].shape[0] if n == 3 else 0
        p.activation = ACTIVATIONS[self.activation]
        p.out_scale, p.out_bias, p.out_lo, p.out_hi = self.out_scale, self.out_bias, self.out_lo, self.out_hi
        for l in range(3):
            p.w[l] = wptrs[l] if l < n else None
            p.b[l] = bptrs[l] if l < n else None
        return p

    def host_struct(self) -> "_abi.PhxPolicyMLP":
        """the same argument over the HOST copies of the weights (the CPU restatement's tests)"""
        return self._c_struct([w.ctypes.data for w in self.weights], [b.ctypes.data for b in self.biases])

    def __call__(self, obs):
        """the network on a torch tensor [..., D] with torch ops (same function; the additions are not in the device's order)"""
        import torch
        ws, bs, _ = self.on(obs.device)
        h = obs
        for l in range(len(ws) - 1):
            h = torch.nn.functional.linear(h, ws[l], bs[l])
            h = torch.clamp(h, -1.0, 1.0) if self.activation == "hard_tanh" else torch.relu(h)
        y = torch.nn.functional.linear(h, ws[-1], bs[-1]).squeeze(-1)
        return torch.clamp(y * self.out_scale + self.out_bias, self.out_lo, self.out_hi)


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)

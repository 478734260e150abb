"""Test infrastructure: a CPU stand-in for DeviceEnv whose env shard is stepped by the oracle, so that the multi-rank
collection code of phantom_amd.distributed (TrajectoryGather, device_env_collector: buffer layouts, bit-packed done flags,
chunked gathers) runs under gloo with world_size > 1 on a box without GPUs.  The fragment layout is DeviceEnv's own
(alloc_trajectory is called unbound on this object); only the launches are replaced."""
import numpy as np
import torch

from phantom_amd.device import DeviceEnv, Trajectory
from oracle import OracleEnv


class OracleShardDev:
    def __init__(self, spec, threads=1):
        self.spec, self.o = spec, OracleEnv(spec, threads=threads)
        self.B, self.S = spec.batch, spec.n_strategic
        self.D = int(self.o.D)
        self.device = torch.device("cpu")
        self._fsm = spec.env_type != 0

    def reset(self):
        self.o.reset()

    def _needs_valid_planes(self):
        return self._fsm

    def never_terminates(self):
        return DeviceEnv.never_terminates(self)

    def alloc_trajectory(self, T, **kw):
        return DeviceEnv.alloc_trajectory(self, T, **kw)

    def rollout(self, T, actions=None, exo=None, out=None):
        r = self.o.rollout(T)
        if out is None:
            out = self.alloc_trajectory(T)
        for name, key in (("observations", "obs"), ("actions", "actions"), ("rewards", "rewards"), ("terminations", "terminated"),
                          ("truncations", "truncated"), ("last_obs", "last_obs"), ("obs_valid", "obs_valid"), ("reward_valid", "reward_valid")):
            dst = getattr(out, name)
            if dst is not None and r.get(key) is not None:
                dst.copy_(torch.from_numpy(np.ascontiguousarray(r[key])).view(dst.shape))
        return out

    def pack_flags(self, plane, dst):
        n = plane.numel()
        bits = np.packbits((plane.reshape(-1).numpy() != 0).astype(np.uint8), bitorder="little")
        words = (n + 63) // 64
        buf = np.zeros(words * 8, np.uint8); buf[:bits.size] = bits
        dst.view(-1)[:words * 8].copy_(torch.from_numpy(buf))

    def pack_done_flags(self, traj):
        return DeviceEnv.pack_done_flags(self, traj)

    def unpack_flags(self, packed, n, out=None):
        bits = np.unpackbits(packed.reshape(-1).numpy()[:((n + 63) // 64) * 8], bitorder="little")[:n]
        t = torch.from_numpy(bits.astype(np.uint8))
        if out is not None:
            out.copy_(t); return out
        return t

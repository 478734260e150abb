"""Multi-GPU layout: one process per GPU, env batch sharded across ranks, no step-time traffic.

Env instances never interact (the reference steps a Python list of independent envs,
utils/rllib/rollout.py:361-363), so rank r simply owns the global envs
[r * B_local, (r + 1) * B_local): identical spec tables are replicated and the device RNG is
keyed by the GLOBAL env index (phx_spec.env_offset), which makes every trajectory independent
of the number of GPUs.  The only exchange is rollout COLLECTION: one all-gather of the
trajectory fragment over RCCL/xGMI when a single learner wants the whole batch
(BASELINE config 4).  ``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" is used by
the CPU tests.
"""
from typing import NamedTuple, Optional, Tuple


class Shard(NamedTuple):
    rank: int
    world_size: int
    local_batch: int
    env_offset: int     # global index of local env 0
    global_batch: int


def shard_batch(global_batch: int, rank: Optional[int] = None,
                world_size: Optional[int] = None) -> Shard:
    """Contiguous, equal split of the env batch (the batch must divide evenly so that every
    rank launches identical grids and the gathered trajectory is a plain [G, T, B/G, ...])."""
    if rank is None or world_size is None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world_size = dist.get_rank(), dist.get_world_size()
        else:
            rank, world_size = 0, 1
    if global_batch % world_size:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world_size}")
    local = global_batch // world_size
    return Shard(rank, world_size, local, rank * local, global_batch)


def make_sharded_env(env_cls, global_batch: int, *args, seed: int = 0, **kwargs):
    """Construct ``env_cls`` for this rank's shard: batch_size = B / world, env_offset = rank * B / world."""
    sh = shard_batch(global_batch)
    return env_cls(*args, batch_size=sh.local_batch, seed=seed, env_offset=sh.env_offset, **kwargs), sh


def all_gather_trajectory(traj, group=None, out=None):
    """Gather a rollout fragment from every rank.

    ``traj`` is a tuple of time-major tensors [T, B_local, ...] (device.Trajectory or any tuple).
    Returns a tuple of tensors [world, T, B_local, ...]: the consumer indexes shards instead of
    paying for a transpose to [T, B_global, ...] (global env = shard * B_local + local env).
    Collectives are issued back to back on the caller's stream; RCCL picks ring vs. direct.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    outs = []
    for k, x in enumerate(traj):
        x = x.contiguous()
        o = out[k] if out is not None else torch.empty((world,) + tuple(x.shape), dtype=x.dtype,
                                                      device=x.device)
        if world == 1:
            o[0].copy_(x)
        else:
            dist.all_gather_into_tensor(o.view(world * x.shape[0], *x.shape[1:]), x, group=group)
        outs.append(o)
    return tuple(outs)


def global_env_index(shard_index: int, local_env: int, local_batch: int) -> int:
    return shard_index * local_batch + local_env

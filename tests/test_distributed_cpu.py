"""CPU, world_size 2, gloo: the sharding + rollout-collection path of phantom_amd.distributed.
The env shards are stepped by the CPU oracle here (the HIP path needs a GPU); what is under test
is that (1) sharding by contiguous env ranges with env_offset keeps every trajectory identical
to the unsharded run (the RNG is keyed by the global env index) and (2) the all-gather layout."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import supply_chain_env
    from oracle import OracleEnv
    from phantom_amd.distributed import all_gather_trajectory, shard_batch
    GB, S, K, T = 8, 3, 2, 30
    sh = shard_batch(GB)
    assert (sh.rank, sh.world_size, sh.local_batch, sh.env_offset) == (rank, world, GB // world, rank * GB // world)
    env = supply_chain_env(S, [K] * S, 10, sh.local_batch, seed=11, env_offset=sh.env_offset)
    o = OracleEnv(env.spec)
    o.reset()
    r = o.rollout(T)                       # device-RNG policy + orders, keyed by global env index
    traj = tuple(torch.from_numpy(r[k]) for k in ("obs", "actions", "rewards", "terminated", "truncated"))
    gathered = all_gather_trajectory(traj)
    assert gathered[0].shape == (world, T, sh.local_batch, S, 3)
    if rank == 0:
        np.savez(os.path.join(tmp, "gathered.npz"), **{f"a{k}": g.numpy() for k, g in enumerate(gathered)})
    # pipelined collection: the same fragment produced and gathered in chunks of 10 steps
    from phantom_amd.distributed import RolloutCollector
    o2 = OracleEnv(env.spec)
    o2.reset()
    names = ("obs", "actions", "rewards", "terminated", "truncated")

    def produce(t0, tc, bufs):
        rr = o2.rollout(tc)                 # continues from the resident state
        for b, k in zip(bufs, names):
            b.copy_(torch.from_numpy(rr[k]))

    col = RolloutCollector(produce, traj, T, 10)
    chunks = col.collect()
    assert chunks[0].shape == (3, world, 10, sh.local_batch, S, 3)
    for g, ch in zip(gathered, chunks):
        # [C, W, tc, ...] -> [W, C*tc, ...]
        glued = torch.cat([ch[c] for c in range(ch.shape[0])], dim=1)
        assert torch.equal(glued, g)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_rollout_equals_unsharded(tmp_path):
    world, port = 2, 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    from helpers import supply_chain_env
    from oracle import OracleEnv
    GB, S, K, T = 8, 3, 2, 30
    env = supply_chain_env(S, [K] * S, 10, GB, seed=11, env_offset=0)
    o = OracleEnv(env.spec)
    o.reset()
    full = o.rollout(T)
    g = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    for k, name in enumerate(("obs", "actions", "rewards", "terminated", "truncated")):
        got = g[f"a{k}"]                                   # [world, T, B/world, ...]
        glued = np.concatenate([got[r] for r in range(world)], axis=1)   # -> [T, B, ...]
        np.testing.assert_array_equal(glued, full[name], err_msg=name)


def test_shard_batch_rejects_uneven_split():
    from phantom_amd.distributed import shard_batch
    with pytest.raises(ValueError):
        shard_batch(10, rank=0, world_size=4)
    assert shard_batch(4096 * 8, rank=3, world_size=8).env_offset == 3 * 4096

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


def test_ads_market_shards_are_independent_of_the_split():
    """every device draw of the ads market (publisher's user / click, budget samplers, connectivity) is
    keyed by the GLOBAL env index: two shards with env_offset reproduce the unsharded rollout."""
    sys.path.insert(0, HERE)
    import phantom_amd as ph
    from oracle import OracleEnv

    def make(batch, offset):
        st = {f"ADV_{i + 1}": ph.AdvertiserAgent.Supertype(budget=ph.UniformFloatSampler(0.5, 1.6, 0.6, 1.5)) for i in range(5)}
        return ph.DigitalAdsEnv(num_steps=6, num_agents_theme={"travel": 2, "tech": 3}, agent_supertypes=st, strategy="second",
                                connection_rates=(1.0, 0.8, 0.9), batch_size=batch, seed=77, env_offset=offset)
    T = 25
    full = OracleEnv(make(6, 0).spec); full.reset()
    rf = full.rollout(T)
    parts = []
    for off in (0, 3):
        o = OracleEnv(make(3, off).spec); o.reset()
        parts.append(o.rollout(T))
    for k in ("obs", "actions", "rewards", "terminated", "truncated", "obs_valid", "reward_valid"):
        np.testing.assert_array_equal(np.concatenate([p[k] for p in parts], axis=1), rf[k], err_msg=k)
    assert rf["rewards"].sum() > 0 and rf["truncated"].sum() > 0

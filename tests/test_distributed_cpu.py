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


def _worker8(rank, world, port, tmp):
    """config 4's shape scaled down: S shops x 4 customers, global batch 2 x world, T = 20, sharded over `world` ranks;
    the HIP launches are replaced by the oracle (tests/cpu_shard_dev.py), the collection code is the product's."""
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_shard_dev import OracleShardDev
    from helpers import supply_chain_env
    from phantom_amd.distributed import TrajectoryGather, device_env_collector, global_env_index, shard_batch
    GB, S, K, T = 2 * world, 5, 4, 20
    sh = shard_batch(GB)
    assert (sh.rank, sh.world_size, sh.local_batch, sh.env_offset) == (rank, world, 2, 2 * rank)
    env = supply_chain_env(S, [K] * S, 7, sh.local_batch, seed=23, env_offset=sh.env_offset)      # episodes of 7 steps: truncations inside the fragment
    dev = OracleShardDev(env.spec); dev.reset()
    tg = TrajectoryGather(dev, T)
    dev.rollout(T, out=tg.traj)
    tg.gather()
    assert tg.traj.packed_flags.numel() == ((T * sh.local_batch * S + 63) // 64) * 8     # truncations as bits; the all-zero terminations plane does not travel
    frs = [tg.unpack(r) for r in range(world)]
    if rank == world - 1:                                   # any rank holds every shard
        np.savez(os.path.join(tmp, "tg.npz"), **{f"{n}{r}": getattr(f, n).numpy() for r, f in enumerate(frs)
                                                   for n in ("observations", "actions", "rewards", "truncations", "terminations")})
    assert global_env_index(rank, 1, sh.local_batch) == sh.env_offset + 1
    # chunked collection of the NEXT fragment (the env continues), chunks of 5 steps, packed flags
    col = device_env_collector(dev, T, chunk=5)
    chunks = col.collect()
    from phantom_amd.distributed import unpack_done_flags
    obs = torch.cat([chunks[0][c] for c in range(col.n_chunks)], dim=1)                 # [W, T, B, S, 3]
    tru = torch.stack([torch.cat([unpack_done_flags(dev, chunks[3][c][r], col.flags_per_chunk, col.flag_planes)[0].view(5, sh.local_batch, S)
                                  for c in range(col.n_chunks)], dim=0) for r in range(world)])
    if rank == 0:
        np.savez(os.path.join(tmp, "col.npz"), obs=obs.numpy(), tru=tru.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_gather_and_chunked_collection_equal_the_unsharded_oracle(tmp_path):
    """VERDICT r3 #4b: world 8 (gloo, CPU): shard_batch + TrajectoryGather + device_env_collector against ONE unsharded oracle."""
    world, port = 8, 31000 + os.getpid() % 2000
    mp.spawn(_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    from helpers import supply_chain_env
    from oracle import OracleEnv
    GB, S, K, T = 2 * world, 5, 4, 20
    o = OracleEnv(supply_chain_env(S, [K] * S, 7, GB, seed=23, env_offset=0).spec)
    o.reset()
    full, nxt = o.rollout(T), o.rollout(T)
    g = np.load(os.path.join(str(tmp_path), "tg.npz"))
    for n, key in (("observations", "obs"), ("actions", "actions"), ("rewards", "rewards"), ("truncations", "truncated"), ("terminations", "terminated")):
        glued = np.concatenate([g[f"{n}{r}"] for r in range(world)], axis=1)          # [T, B_global, ...]
        np.testing.assert_array_equal(glued, full[key], err_msg=n)
    assert full["truncated"].sum() > 0
    c = np.load(os.path.join(str(tmp_path), "col.npz"))
    np.testing.assert_array_equal(np.concatenate([c["obs"][r] for r in range(world)], axis=1), nxt["obs"])
    np.testing.assert_array_equal(np.concatenate([c["tru"][r] for r in range(world)], axis=1), nxt["truncated"])


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

"""One rank of the "two ranks on ONE GPU" parity run (tests/test_gpu_round3.py).

    python two_rank_worker.py RANK WORLD PORT B T OUTDIR

Every rank uses cuda:0 (RCCL refuses several ranks per device, so the process group is gloo and the collectives of
phantom_amd.distributed stage through pinned host memory: `staging="auto"` picks that for a gloo group).  Rank r steps
BASELINE config 4's env (SC256 = 1 factory + 51 shops + 204 customers, plain env, device RNG, random policy) for
its shard of B envs on the HIP path with env_offset = r * B, the fragment is collected twice -- one flat collective
(TrajectoryGather) and the chunked produce + collect pipeline (device_env_collector) -- and rank 0 compares every
rank's gathered, unpacked fragment with the UNSHARDED oracle run of WORLD * B envs, bit for bit.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, B, T = (int(x) for x in sys.argv[1:6])
    outdir = sys.argv[6]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import f32_bits, supply_chain_env
    from phantom_amd.device import DeviceEnv
    from phantom_amd.distributed import TrajectoryGather, device_env_collector, shard_batch, unpack_done_flags
    S, K, NS = 51, 4, 100
    sh = shard_batch(world * B)
    assert (sh.local_batch, sh.env_offset) == (B, rank * B)
    torch.cuda.set_device(0)

    def make():
        env = supply_chain_env(S, [K] * S, NS, B, seed=42, env_offset=sh.env_offset)
        dev = DeviceEnv(env.spec, device="cuda:0")
        assert dev.uses_fused
        dev.reset()
        return dev

    # (1) one flat collective for the whole fragment
    dev = make()
    tg = TrajectoryGather(dev, T)
    dev.rollout(T, out=tg.traj)
    tg.gather()
    torch.cuda.synchronize()
    # (2) the chunked produce + collect pipeline on a second env instance of the same shard
    dev2 = make()
    chunk = T // 5 if T % 5 == 0 else T
    col = device_env_collector(dev2, T, chunk=chunk)
    out = col.collect()
    torch.cuda.synchronize()
    stock = dev.field("shop.stock").cpu().numpy()
    res = {"rank": rank, "staging": "host", "backend": dist.get_backend(), "bytes_per_rank": tg.nbytes, "checked": []}
    if rank == 0:
        from oracle import OracleEnv
        ncpu = max(1, min(os.cpu_count() or 1, 64))
        full = supply_chain_env(S, [K] * S, NS, world * B, seed=42, env_offset=0)
        o = OracleEnv(full.spec, threads=ncpu)
        o.reset()
        ro = o.rollout(T)
        for r in range(world):
            got = tg.unpack(r)
            sl = slice(r * B, (r + 1) * B)
            for name, g in (("obs", got.observations), ("actions", got.actions), ("rewards", got.rewards)):
                np.testing.assert_array_equal(f32_bits(g.cpu().numpy()), f32_bits(ro[name][:, sl]), err_msg=f"flat gather: {name} of rank {r}")
            np.testing.assert_array_equal(got.truncations.cpu().numpy(), ro["truncated"][:, sl], err_msg=f"truncated of rank {r}")
            np.testing.assert_array_equal(got.terminations.cpu().numpy(), ro["terminated"][:, sl], err_msg=f"terminated of rank {r}")
            res["checked"].append(f"flat rank {r}")
            # the pipeline's chunks [n_chunks, world, chunk, B, ...] glued along time
            for k, name in enumerate(("obs", "actions", "rewards")):
                glued = torch.cat([out[k][c, r] for c in range(col.n_chunks)], 0).cpu().numpy()
                np.testing.assert_array_equal(f32_bits(glued), f32_bits(ro[name][:, sl]), err_msg=f"pipeline: {name} of rank {r}")
            tr = []
            for c in range(col.n_chunks):
                t_, e_ = unpack_done_flags(dev2, out[3][c, r], col.flags_per_chunk, col.flag_planes)
                assert int(e_.sum()) == 0
                tr.append(t_.view(chunk, B, S))
            np.testing.assert_array_equal(torch.cat(tr, 0).cpu().numpy(), ro["truncated"][:, sl], err_msg=f"pipeline: truncated of rank {r}")
            res["checked"].append(f"pipeline rank {r}")
        # the shard's resident state after the fragment equals the unsharded run's rows
        np.testing.assert_array_equal(stock, o.get_i32("shop.stock")[:B])
        assert ro["truncated"].sum() == (T // NS) * world * B * S
    dist.barrier()
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

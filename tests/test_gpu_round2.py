"""GPU (-m gpu), round 2: BASELINE.json's full-size configs against the ORACLE (not fused-vs-generic), the
rollout message log, the copying state/trace ABI, the packed rollout-collection payload, captured step graphs,
and bench.py's collective path on one GPU (RCCL world 1).  Everything goes through the C ABI."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import phantom_amd as ph

from helpers import f32_bits, f64_bits, market_env, supply_chain_env
from oracle import LOG_DTYPE, OracleEnv

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCPU = max(1, min(os.cpu_count() or 1, 64))


def _dev(spec):
    from device_runner import DeviceRunner
    return DeviceRunner(spec)


def _cmp_rollout(rd, ro, valid_planes):
    for k in ("obs", "actions", "rewards", "last_obs"):
        np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=k)
    for k in ("truncated", "terminated") + (("obs_valid", "reward_valid") if valid_planes else ()):
        np.testing.assert_array_equal(rd[k], ro[k], err_msg=k)


# ---- full-size parity against the oracle --------------------------------------
def test_config3_fsm_rollout_full_size_matches_oracle():
    """BASELINE configs[2]: SC256 (1 + 51 + 204 agents), 2-stage FSM, B = 8192 -- one T = 100 fragment of
    phx_sc_rollout_fsm_kernel (device RNG, auto-reset) bit-equal to the oracle (fsm.py:253-380)."""
    env = supply_chain_env(51, [4] * 51, 100, 8192, fsm=True, seed=42)
    o, d = OracleEnv(env.spec, threads=NCPU), _dev(env.spec)
    assert d.dev.uses_fused
    o.reset(); d.reset()
    ro, rd = o.rollout(100), d.rollout(100)
    _cmp_rollout(rd, ro, True)
    for f in ("shop.stock", "shop.sales", "shop.missed_sales", "env.stage", "env.step", "env.tick"):
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)
    assert (d.err == 0).all()
    # and a second fragment that starts mid-state (caches carried across the launch boundary)
    ro, rd = o.rollout(37), d.rollout(37)
    _cmp_rollout(rd, ro, True)


def test_config4_share_rollout_full_size_matches_oracle_with_env_offset():
    """BASELINE configs[3], the share of GPU 3 of 8: SC256 plain env, B = 8192, env_offset = 3 * 8192
    (the RNG is keyed by the GLOBAL env index) -- phx_sc_rollout_kernel bit-equal to the oracle."""
    env = supply_chain_env(51, [4] * 51, 100, 8192, seed=42, env_offset=3 * 8192)
    o, d = OracleEnv(env.spec, threads=NCPU), _dev(env.spec)
    assert d.dev.uses_fused
    o.reset(); d.reset()
    ro, rd = o.rollout(100), d.rollout(100)
    _cmp_rollout(rd, ro, False)
    np.testing.assert_array_equal(d.get_i32("shop.stock"), o.get_i32("shop.stock"))
    # the same global envs stepped as part of an unsharded batch give the same rows: shard 3 == rows of offset 0
    env0 = supply_chain_env(51, [4] * 51, 100, 64, seed=42, env_offset=3 * 8192 + 100)
    d0 = _dev(env0.spec); d0.reset()
    r0 = d0.rollout(100)
    np.testing.assert_array_equal(f32_bits(r0["obs"]), f32_bits(rd["obs"][:, 100:164]))


def test_config5_market_full_size_matches_oracle():
    """BASELINE configs[4]: Stackelberg market, 128 leaders / 1024 followers, B = 4096: six phx_step launches of
    the fused market kernel (leaders / followers alternate, stackelberg.py:111-196) and a T = 8 fused rollout,
    against the oracle: obs f32 and rewards f64 by bit pattern, key sets, flags, state."""
    B, L, Fw = 4096, 128, 1024
    env = market_env(L, Fw, 8, 100, B, seed=7)
    o, d = OracleEnv(env.spec, threads=NCPU), _dev(env.spec)
    assert d.dev.uses_fused
    o.reset(); d.reset()
    rng = np.random.default_rng(3)
    S = L + Fw
    for t in range(6):
        a = rng.random((B, S), dtype=np.float32)
        av = np.zeros((B, S), np.uint8)
        if t % 2 == 0:
            av[:, :L] = 1
        else:
            av[:, L:] = 1
        o.step(a, av, None); d.step(a, av, None)
        np.testing.assert_array_equal(d.obs_valid, o.obs_valid, err_msg=f"obs_valid t={t}")
        np.testing.assert_array_equal(d.reward_valid, o.reward_valid, err_msg=f"reward_valid t={t}")
        m = o.obs_valid.astype(bool)
        np.testing.assert_array_equal(f32_bits(d.obs)[m], f32_bits(o.obs)[m], err_msg=f"obs t={t}")
        mr = o.reward_valid == 1
        np.testing.assert_array_equal(f64_bits(d.reward)[mr], f64_bits(o.reward)[mr], err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d.truncated, o.truncated)
        np.testing.assert_array_equal(d.all_truncated, o.all_truncated)
    for f in ("seller.price", "seller.revenue", "buyer.paid"):
        np.testing.assert_array_equal(f64_bits(d.get_f64(f)), f64_bits(o.get_f64(f)), err_msg=f)
    for f in ("seller.tx", "buyer.bought"):
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)
    ro, rd = o.rollout(8), d.rollout(8)
    _cmp_rollout(rd, ro, True)
    assert (d.err == 0).all()


# ---- message log of a rollout (rollout.py:369-373) -----------------------------------------------------
def test_rollout_message_log_matches_oracle():
    """record_messages=True: Resolver.tracked_messages of EVERY step of a fragment, incl. the steps after an
    auto-reset, equal to the oracle's ordered log (sender, receiver, type, round, payload)."""
    import torch
    B, T = 6, 25
    env = supply_chain_env(3, [2, 5, 1], 10, B, tracking=True, seed=9)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    ro = o.rollout(T, record_messages=True)
    dev = d.dev
    tr = dev.alloc_trajectory(T, record_messages=True)
    dev.rollout(T, out=tr)
    cnt = tr.msg_count.cpu().numpy()
    np.testing.assert_array_equal(cnt, ro["msg_count"])
    log = np.frombuffer(tr.msg_log.cpu().numpy().tobytes(), dtype=LOG_DTYPE).reshape(T, B, -1)
    for t in range(T):
        for b in range(B):
            n = int(cnt[t, b])
            assert n > 0
            np.testing.assert_array_equal(log[t, b, :n], ro["msg_log"][t, b, :n], err_msg=f"t={t} b={b}")
    np.testing.assert_array_equal(f32_bits(tr.observations.cpu().numpy()), f32_bits(ro["obs"]))
    # a fused env (tracking off) has no log to give
    env2 = supply_chain_env(3, [2, 5, 1], 10, B, seed=9)
    d2 = _dev(env2.spec).dev
    with pytest.raises(Exception):
        d2.alloc_trajectory(T, record_messages=True)
    with pytest.raises(ValueError):
        d2.rollout(T, out=tr)                      # a log for an env without tracking: rejected on the host


def test_rollout_rejects_malformed_buffers():
    """raw pointers go to the kernel, so dtype / shape / contiguity / device are checked."""
    import torch
    env = supply_chain_env(3, [2, 2, 2], 10, 4, seed=1)
    dev = _dev(env.spec).dev
    dev.reset()
    tr = dev.alloc_trajectory(5)
    with pytest.raises(ValueError):
        dev.rollout(5, out=tr._replace(observations=tr.observations.to(torch.float64)))
    with pytest.raises(ValueError):
        dev.rollout(5, out=tr._replace(rewards=tr.rewards[:, :, :2]))
    with pytest.raises(ValueError):
        dev.rollout(5, out=tr._replace(truncations=tr.truncations.cpu()))
    with pytest.raises(ValueError):
        dev.rollout(5, actions=torch.zeros(4, 4, 3, device=dev.device))
    with pytest.raises(ValueError):
        dev.rollout(6, out=tr)
    # a view that starts off a 16-byte boundary (the kernels write 16-byte pieces): rejected in Python and by the ABI
    big = torch.zeros(tr.truncations.numel() + 16, dtype=torch.uint8, device=dev.device)
    odd = big[3:3 + tr.truncations.numel()].view(tr.truncations.shape)
    with pytest.raises(ValueError):
        dev.rollout(5, out=tr._replace(truncations=odd))
    dev.rollout(5, out=tr)
    # repeated rollouts into internally allocated fragments pin nothing
    for _ in range(8):
        dev.rollout(5)
    assert len(dev._rollout_io_cache) <= 4
    assert all(not any(hasattr(x, "data_ptr") for x in v) for v in dev._rollout_io_cache.values())


# ---- copying state access / trace through the ABI --------------------------------------------
def test_get_set_state_and_trace_through_the_abi():
    import torch
    B = 5
    env = supply_chain_env(2, [3, 3], 10, B, tracking=True, seed=4)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    dev, lib = d.dev, d.dev.lib
    a = np.full((B, 2), 30.0, np.float32)
    exo = (np.arange(B * 6).reshape(B, 6) % 5).astype(np.uint8)
    o.step(a, None, exo); d.step(a, None, exo)
    host = np.zeros((B, 2), np.int32)
    n = lib.phx_get_state(dev.handle, b"shop.stock", host.ctypes.data, host.nbytes, dev._stream())
    assert n == host.nbytes
    np.testing.assert_array_equal(host, o.get_i32("shop.stock"))
    assert lib.phx_get_state(dev.handle, b"no.such.field", host.ctypes.data, host.nbytes, dev._stream()) < 0
    assert lib.phx_get_state(dev.handle, b"shop.stock", host.ctypes.data, 4, dev._stream()) < 0
    new = (np.arange(B * 2, dtype=np.int32).reshape(B, 2) * 7) % 90
    assert lib.phx_set_state(dev.handle, b"shop.stock", new.ctypes.data, new.nbytes, dev._stream()) == new.nbytes
    np.testing.assert_array_equal(dev.field("shop.stock").cpu().numpy(), new)
    # device-pointer destination
    t = torch.zeros(B, 2, dtype=torch.int32, device=dev.device)
    assert lib.phx_get_state(dev.handle, b"shop.stock", t.data_ptr(), t.numel() * 4, dev._stream()) == new.nbytes
    np.testing.assert_array_equal(t.cpu().numpy(), new)
    # phx_trace: the last step's ordered log of env 2
    from phantom_amd import _abi
    recs = (_abi.PhxMsgRec * env.spec.trace_cap)()
    cnt = lib.phx_trace(dev.handle, dev.msg_log.data_ptr(), dev.msg_count.data_ptr(), 2, recs, env.spec.trace_cap,
                        dev._stream())
    want = o.log(2)
    assert cnt == len(want)
    got = np.frombuffer(bytes(recs), dtype=LOG_DTYPE)[:cnt]
    np.testing.assert_array_equal(got, want)


# ---- rollout collection payload (SURVEY 8e iii) -----------------------------------------------------------
def test_pack_flags_roundtrip_and_flat_gather_world1():
    import torch
    env = supply_chain_env(9, [6] * 9, 20, 64, seed=2, env_offset=640)
    dev = _dev(env.spec).dev
    dev.reset()
    x = (torch.rand(100_003, device=dev.device) < 0.3).to(torch.uint8) * 3
    packed = torch.zeros((100_003 + 63) // 64 * 8, dtype=torch.uint8, device=dev.device)
    dev._check(dev.lib.phx_pack_flags(x.data_ptr(), packed.data_ptr(), x.numel(), dev._stream()), "pack")
    want = np.packbits((x.cpu().numpy() != 0), bitorder="little")
    np.testing.assert_array_equal(packed.cpu().numpy()[:len(want)], want)
    back = dev.unpack_flags(packed, x.numel())
    np.testing.assert_array_equal(back.cpu().numpy(), (x.cpu().numpy() != 0).astype(np.uint8))
    # one flat collective for a whole fragment (world 1: the copy path), done planes restored
    from phantom_amd.distributed import TrajectoryGather
    tg = TrajectoryGather(dev, 45)
    assert dev.never_terminates()
    dev.rollout(45, out=tg.traj)
    tg.gather()
    got = tg.unpack(0)
    for a, b in ((got.observations, tg.traj.observations), (got.actions, tg.traj.actions),
                 (got.rewards, tg.traj.rewards), (got.truncations, tg.traj.truncations),
                 (got.terminations, tg.traj.terminations)):
        assert torch.equal(a, b)
    assert tg.traj.truncations.sum().item() == 2 * 64 * 9          # steps 20 and 40 end an episode
    assert tg.nbytes < tg.raw_nbytes * 0.93                        # 22 S -> 20 S + S / 8 bytes per env-step


def test_device_env_collector_hip_path_with_env_offset():
    """the chunked produce + collect pipeline on the HIP path (world 1), env_offset != 0: the glued chunks equal
    the oracle's fragment of the same GLOBAL envs; done flags travel bit-packed."""
    import torch
    from phantom_amd.distributed import device_env_collector, unpack_done_flags
    B, T, off = 32, 40, 5 * 32
    env = supply_chain_env(9, [6] * 9, 25, B, seed=6, env_offset=off)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    ro = o.rollout(T)
    col = device_env_collector(d.dev, T, chunk=10)
    out = col.collect()
    torch.cuda.synchronize()
    obs = torch.cat([out[0][c, 0] for c in range(col.n_chunks)], 0).cpu().numpy()
    act = torch.cat([out[1][c, 0] for c in range(col.n_chunks)], 0).cpu().numpy()
    rew = torch.cat([out[2][c, 0] for c in range(col.n_chunks)], 0).cpu().numpy()
    np.testing.assert_array_equal(f32_bits(obs), f32_bits(ro["obs"]))
    np.testing.assert_array_equal(f32_bits(act), f32_bits(ro["actions"]))
    np.testing.assert_array_equal(f32_bits(rew), f32_bits(ro["rewards"]))
    tr = []
    for c in range(col.n_chunks):
        t_, e_ = unpack_done_flags(d.dev, out[3][c, 0], col.flags_per_chunk, col.flag_planes)
        tr.append(t_.view(10, B, 9)); assert int(e_.sum()) == 0
    np.testing.assert_array_equal(torch.cat(tr, 0).cpu().numpy(), ro["truncated"])
    # the unpacked variant carries the same planes
    d2 = _dev(env.spec); d2.reset()
    col2 = device_env_collector(d2.dev, T, chunk=20, pack_flags=False)
    out2 = col2.collect(); torch.cuda.synchronize()
    np.testing.assert_array_equal(torch.cat([out2[4][c, 0] for c in range(2)], 0).cpu().numpy(), ro["truncated"])
    assert col.nbytes * col.n_chunks < col2.nbytes * col2.n_chunks


# ---- captured step graphs ----------------------------------------------------------------
def test_step_graph_replays_equal_direct_steps():
    import torch
    B, S, n = 256, 9, 12
    env = supply_chain_env(S, [6] * S, 100, B, seed=3)
    o, d = OracleEnv(env.spec, threads=4), _dev(env.spec)
    o.reset(); d.reset()
    dev = d.dev
    acts = torch.rand(n, B, S, device=dev.device) * 100.0
    g = dev.step_graph(acts)
    for rep in range(2):                                          # two replays = 24 consecutive steps
        g.replay()
        for i in range(n):
            o.step(acts[i].cpu().numpy(), None, None)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(f32_bits(dev.obs.cpu().numpy()), f32_bits(o.obs))
        np.testing.assert_array_equal(f64_bits(dev.reward.cpu().numpy()), f64_bits(o.reward))
        np.testing.assert_array_equal(dev.field("shop.stock").cpu().numpy(), o.get_i32("shop.stock"))
    # policy slot: actions computed on the device from the previous step's observation, inside the graph
    d2 = _dev(env.spec); dev2 = d2.dev
    o2 = OracleEnv(env.spec, threads=4)
    obs_prev, _ = o2.reset(); d2.reset()
    pol = lambda st: (st.observations[..., 0] * 37.0 + 11.0).contiguous()
    g2 = dev2.step_graph(n=5, policy=pol)
    g2.replay(); torch.cuda.synchronize()
    for i in range(5):
        a = (obs_prev[..., 0] * np.float32(37.0) + np.float32(11.0)).astype(np.float32)
        o2.step(a, None, None)
        obs_prev = o2.obs
    np.testing.assert_array_equal(dev2.field("shop.stock").cpu().numpy(), o2.get_i32("shop.stock"))
    np.testing.assert_array_equal(f32_bits(dev2.obs.cpu().numpy()), f32_bits(o2.obs))


# ---- bench.py: the collective path on ONE GPU (RCCL world 1) ----------------------------------------------------
def test_bench_collective_path_on_one_gpu():
    """`bench.py` with PHX_BENCH_FORCE_DIST=1 initialises RCCL with one rank and runs the
    all-gather, the pipelined collector and config 4's per-GPU workload on the HIP path."""
    env = dict(os.environ, PHX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29600 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-per-step", "--no-other-configs", "--min-region-ms", "20"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["rccl_ranks_seen"] == 1 and r["n_gpus"] == 1 and r["steps"] == 20
    assert r["repeats"] * 20 == r["timed_steps"] and r["timed_region_ms"] >= 15.0
    assert abs(r["ms_per_step"] - r["timed_region_ms"] / r["timed_steps"]) < 1e-9
    assert r["value"] > 1e10
    ag = r["rollout_allgather"]
    assert ag["bytes_per_rank"] < ag["raw_trajectory_bytes_per_rank"]
    c4 = r["config4_share"]
    assert c4["agent_steps_per_sec_gather_included"] > 0 and c4["agent_steps_per_sec_gather_excluded"] > 0


# ---- the round-2 rollout kernel (phx_sc_rollout.hip): its special cases against the oracle ---------------------
@pytest.mark.parametrize("S,K,B,num_steps", [(9, 6, 64, 100), (9, 6, 64, 23), (3, 2, 48, 40), (12, 5, 16, 31), (1, 1, 64, 20),
                                             (51, 4, 16, 100), (20, 3, 8, 57),
                                             # blocks of pair ranges that start and end inside envs (S too wide for 4 whole envs)
                                             (51, 4, 64, 100), (30, 6, 8, 25), (100, 2, 4, 40), (7, 3, 12, 30)])
def test_fast_rollout_kernel_edge_cases_match_oracle(S, K, B, num_steps):
    """device-RNG rollouts through the fast kernel where its plan applies (uniform 1..6 customers; whole envs per
    block, or blocks of consecutive (env, shop) pairs for wide envs) and through the general kernel otherwise: fragments that start on ticks that are not multiples of 4
    (per-step launches first), fragment lengths that are no multiple of the chunk, several episode ends per
    fragment, stocks poked outside [0, 100] (the reference's arithmetic for a negative stock), state hand-over to
    per-step launches."""
    env = supply_chain_env(S, [K] * S, num_steps, B, seed=11 + S, env_offset=1000)
    o, d = OracleEnv(env.spec, threads=4), _dev(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(S * 100 + K)
    for t in range(3):                                           # ticks 0..2: the next fragment is not quad-aligned
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        o.step(a, None, None); d.step(a, None, None)
    for T in (1, 7, 20, 41, 100, 3):
        ro, rd = o.rollout(T), d.rollout(T)
        _cmp_rollout(rd, ro, False)
        for f in ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick"):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after T={T}")
    # stocks outside [0, 100] (a caller poking ShopAgent.stock): negative sales / a request clamp below zero
    st = rng.integers(-40, 160, (B, S)).astype(np.int32)
    o.set_i32("shop.stock", st); d.set_i32("shop.stock", st)
    for T in (20, 9):
        ro, rd = o.rollout(T), d.rollout(T)
        _cmp_rollout(rd, ro, False)
    a = rng.uniform(0, 100, (B, S)).astype(np.float32)
    o.step(a, None, None); d.step(a, None, None)
    np.testing.assert_array_equal(f32_bits(d.obs), f32_bits(o.obs))
    np.testing.assert_array_equal(f64_bits(d.reward), f64_bits(o.reward))
    assert (d.err == 0).all()


@pytest.mark.parametrize("S,K,B,num_steps", [(9, 6, 64, 100), (51, 4, 16, 100), (3, 2, 48, 7), (5, 1, 40, 33)])
def test_fsm_lean_rollout_loop_edge_cases_match_oracle(S, K, B, num_steps):
    """device-RNG rollouts of FSM supply chains whose shops all have the same 1..6 customers (at these sizes the time-parallel
    kernel of phx_sc_rollout_fsm.hip; PHX_FSM_FAST=0: the lean loop of phx_sc_fused.hip; PHX_FSM_LEAN=0: the general one):
    fragments starting on ticks that are no multiple of 4 and in
    either stage, several episode ends per fragment, caches carried across launches, stocks poked outside [0, 100] (the
    observation tables do not cover them), hand-over to per-step launches."""
    env = supply_chain_env(S, [K] * S, num_steps, B, fsm=True, seed=5 + S, env_offset=77)
    o, d = OracleEnv(env.spec, threads=4), _dev(env.spec)
    assert d.dev.uses_fused
    o.reset(); d.reset()
    rng = np.random.default_rng(S * 10 + K)
    for t in range(3):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        o.step(a, None, None); d.step(a, None, None)
    fields = ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.stage", "env.step", "env.tick")
    for T in (1, 6, 41, 100, 3):
        ro, rd = o.rollout(T), d.rollout(T)
        _cmp_rollout(rd, ro, True)
        for f in fields:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after T={T}")
    st = rng.integers(-40, 160, (B, S)).astype(np.int32)
    o.set_i32("shop.stock", st); d.set_i32("shop.stock", st)
    for T in (12, 5):
        ro, rd = o.rollout(T), d.rollout(T)
        _cmp_rollout(rd, ro, True)
    # envs pushed off the handler-less stage chain (a stage a handler or the caller chose): the time-parallel kernel's
    # position table does not describe them, its check sends the launch to the lane-per-pair loop
    stg = o.get_i32("env.stage").copy()
    stg[::2] = 1 - stg[::2]
    o.set_i32("env.stage", stg); d.set_i32("env.stage", stg)
    for T in (9, 30):
        ro, rd = o.rollout(T), d.rollout(T)
        _cmp_rollout(rd, ro, True)
        for f in fields:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after an off-chain stage, T={T}")
    a = rng.uniform(0, 100, (B, S)).astype(np.float32)
    o.step(a, None, None); d.step(a, None, None)
    np.testing.assert_array_equal(f32_bits(d.obs), f32_bits(o.obs))
    np.testing.assert_array_equal(f64_bits(d.reward), f64_bits(o.reward))
    np.testing.assert_array_equal(d.obs_valid, o.obs_valid)
    assert (d.err == 0).all()


# ---- BatchResolver(shuffle_batches=True) on the device stream (resolvers.py:150-151) ----------------------------
@pytest.mark.parametrize("fsm", [False, True])
def test_shuffle_batches_device_stream_matches_oracle(fsm):
    """the permutations drawn on the device (Fisher-Yates on Philox blocks keyed by env, tick, round, receiver)
    equal the oracle's restatement: per-step launches with tracking (the message log shows the handled order's
    effect), then a launch-loop rollout; batches of up to 23 messages (several draw blocks)."""
    B = 12
    env = supply_chain_env(3, [23, 2, 6], 7, B, fsm=fsm, tracking=True, shuffle=True, seed=77, env_offset=31)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    assert not d.dev.uses_fused
    o.reset(); d.reset()
    rng = np.random.default_rng(5)
    first_logs = None
    for t in range(9):
        a = rng.uniform(0, 100, (B, 3)).astype(np.float32)
        o.step(a, None, None); d.step(a, None, None)
        if t == 0:
            first_logs = [d.log(b) for b in range(B)]
        for f in ("shop.stock", "shop.sales", "shop.missed_sales"):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} t={t}")
        np.testing.assert_array_equal(f32_bits(d.obs), f32_bits(o.obs))
        for b in (0, 5):
            np.testing.assert_array_equal(d.log(b), o.log(b), err_msg=f"log t={t} b={b}")
    ro, rd = o.rollout(10), d.rollout(10)
    _cmp_rollout(rd, ro, fsm)
    assert (d.err == 0).all()
    # and the shuffle matters: the same env without it handles its batches in send order.  A shop's totals have a closed
    # form whatever the order (sales = min(sum D, stock)), so the difference shows in WHO is served: the ordered message log
    # (the OrderResponses of the first step) differs for at least one env
    env2 = supply_chain_env(3, [23, 2, 6], 7, B, fsm=fsm, tracking=True, shuffle=False, seed=77, env_offset=31, force_generic=True)
    d2 = _dev(env2.spec); d2.reset()
    rng = np.random.default_rng(5)
    d2.step(rng.uniform(0, 100, (B, 3)).astype(np.float32), None, None)
    plain_logs = [d2.log(b) for b in range(B)]
    assert all(len(x) == len(y) for x, y in zip(first_logs, plain_logs))
    assert any(not np.array_equal(x, y) for x, y in zip(first_logs, plain_logs))


# ---- FSM stage handlers (fsm.py:294-307): the host calls the Python handler, the device gets its choice ---------
def test_fsm_stage_handler_python_surface_matches_reference():
    """the golden `sc_fsm_handler` is the reference running a FiniteStateMachineEnv whose RESTOCK stage has a handler
    that restocks twice on every third step.  Here the same env is built from phantom_amd classes with a PYTHON
    handler (tests/helpers.py: golden_restock_handler); step_tensors() calls it on the host before each launch and
    the device validates and applies the transition: stage sequence, key sets, observations and rewards equal the
    reference's."""
    import torch
    from helpers import env_from_golden, golden
    g = golden("sc_fsm_handler")
    T, B = int(g["T"]), len(g["seeds"])
    env = env_from_golden(g, exogenous="device")
    dev = env._device()
    assert env._has_handlers and not env.is_fsm_deterministic()
    for t in range(T):
        if g["reset_before"][t].any():
            env.reset()
        np.testing.assert_array_equal(np.atleast_1d(env._h_stage), g["stage"][t], err_msg=f"stage before t={t}")
        a = torch.from_numpy(g["actions"][t]).to(dev.device)
        x = torch.from_numpy(g["exo"][t]).to(dev.device)
        env.step_tensors(a, None, x, check_errors=True)
        np.testing.assert_array_equal(np.atleast_1d(env._h_stage), g["next_stage"][t], err_msg=f"stage after t={t}")
        np.testing.assert_array_equal(dev.field("env.stage")[:, 0].cpu().numpy(), g["next_stage"][t])
        np.testing.assert_array_equal(dev.obs_valid.cpu().numpy(), g["obs_valid"][t], err_msg=f"obs_valid t={t}")
        np.testing.assert_array_equal(dev.reward_valid.cpu().numpy(), g["reward_valid"][t], err_msg=f"reward_valid t={t}")
        ov = g["obs_valid"][t].astype(bool)
        np.testing.assert_array_equal(f32_bits(dev.obs.cpu().numpy()[ov]), f32_bits(g["obs"][t][ov]))
        rv = g["reward_valid"][t] == 1
        np.testing.assert_array_equal(f64_bits(dev.reward.cpu().numpy()[rv]), f64_bits(g["reward"][t][rv]))
        np.testing.assert_array_equal(dev.field("shop.stock").cpu().numpy(), g["stock"][t])


def test_fsm_invalid_transition_is_an_error():
    """fsm.py:304-307: a handler returning a stage outside next_stages -> FSMRuntimeError.  Through the ABI the
    device reports PHX_ERR_FSM_TRANSITION for exactly the envs that asked for it; the oracle agrees."""
    from helpers import env_from_golden, golden
    from phantom_amd import _abi
    g = golden("sc_fsm_handler")
    env = env_from_golden(g, batch=4, exogenous="device")
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    a = np.full((4, 3), 10.0, np.float32)
    nxt = np.asarray([1, 0, 1, 1], np.int32)                 # from RESTOCK both SELL and RESTOCK are allowed
    o.step(a, None, None, next_stage=nxt); d.step(a, None, None, next_stage=nxt)
    assert (d.err == 0).all() and (o.err == 0).all()
    np.testing.assert_array_equal(d.get_i32("env.stage")[:, 0], nxt)
    np.testing.assert_array_equal(o.get_i32("env.stage")[:, 0], nxt)
    np.testing.assert_array_equal(d.obs_valid, o.obs_valid)
    bad = np.asarray([0, 1, 0, 7], np.int32)                 # env 0: SELL -> RESTOCK fine; env 1: RESTOCK -> SELL fine;
    o.step(a, None, None, next_stage=bad); d.step(a, None, None, next_stage=bad)    # env 2: SELL -> RESTOCK; env 3: no stage 7
    np.testing.assert_array_equal(d.err, o.err)
    assert d.err.tolist() == [0, 0, 0, _abi.ERR_FSM_TRANSITION]
    # the Python surface raises the reference's exception before launching
    envp = env_from_golden(g, batch=2, exogenous="device")
    envp._stage_list[0].handler = lambda e: "NOWHERE"
    envp.reset()
    import torch
    with pytest.raises(ph.FSMRuntimeError):
        envp.step_tensors(torch.zeros(2, 3, device=envp._device().device))

"""GPU (-m gpu), round 3: the HIP path with MORE THAN ONE RANK (two ranks sharing cuda:0, host-staged gloo collective)
against the unsharded oracle; bench.py's watchdog / error line; every rollout kernel variant against the oracle at
small sizes; per-step phx_step at config 2's full size against the oracle.  Everything goes through the C ABI."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import phantom_amd as ph

from helpers import f32_bits, f64_bits, supply_chain_env
from oracle import OracleEnv

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NCPU = max(1, min(os.cpu_count() or 1, 64))


def _dev(spec):
    from device_runner import DeviceRunner
    return DeviceRunner(spec)


# ---- world size 2 on one GPU -----------------------------------------------------------------
@pytest.mark.parametrize("B,T", [(1024, 100), (8192, 100)])
def test_two_ranks_on_one_gpu_config4_share_matches_unsharded_oracle(B, T, tmp_path):
    """BASELINE configs[3] with world size 2: each rank steps its shard of SC256 envs on the HIP path
    (env_offset = rank * B), the fragment is collected by ONE flat collective (TrajectoryGather) and by the chunked
    pipeline (device_env_collector) -- gloo group, host-staged because both ranks use cuda:0 -- and rank 0 checks
    every rank's gathered, unpacked fragment against the UNSHARDED oracle run of 2 B envs bit for bit
    (utils/rllib/rollout.py:220-258,361-363: the Ray fan-out + list-of-envs loop this replaces)."""
    world, port = 2, 29100 + (os.getpid() * 7 + B) % 800
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=str(NCPU))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_worker.py"), str(r), str(world), str(port),
                               str(B), str(T), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
    r0 = json.load(open(os.path.join(str(tmp_path), "rank0.json")))
    assert r0["backend"] == "gloo" and len(r0["checked"]) == 2 * world
    assert os.path.exists(os.path.join(str(tmp_path), "rank1.json"))


# ---- bench.py cannot fail silently -------------------------------------------------------------
def _bench(args, extra_env=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}\nstdout: {p.stdout[-1500:]}\nstderr: {p.stderr[-1500:]}"
    return p, json.loads(lines[0])


FAST = ["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-per-step", "--no-other-configs", "--no-frag200",
        "--min-region-ms", "100"]


def test_bench_two_ranks_without_a_second_gpu_prints_an_error_line():
    """`--gpus 2` on a 1-GPU box: rank 1 has no device.  The run must end with ONE JSON line carrying n_gpus,
    rccl_ranks_seen and an `error` field -- never hang, never exit silently."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with one GPU")
    p, r = _bench(["--gpus", "2", "--watchdog-s", "120"] + FAST)
    assert p.returncode != 0
    assert r["n_gpus"] == 2 and r["value"] is None and r["rccl_ranks_seen"] == 0
    assert "error" in r and r["error"]


def test_bench_two_ranks_sharing_one_gpu_reports_value_and_the_rccl_failure():
    """two ranks on cuda:0 (PHX_BENCH_SHARE_GPU=1): the step path has no collective and the control plane is gloo, so
    `value` is measured for both ranks; RCCL refuses the duplicate device, which must surface as `rccl_error` /
    `error` on the line (or, should RCCL accept it, as a working rollout_allgather section)."""
    p, r = _bench(["--gpus", "2", "--watchdog-s", "400"] + FAST, {"PHX_BENCH_SHARE_GPU": "1"})
    assert r["n_gpus"] == 2 and r["control_plane"] == "gloo"
    assert r["value"] is not None and r["value"] > 1e10
    assert r["config"]["global_envs"] == 2 * 4096
    if r["rccl_ranks_seen"] != 2:
        assert "error" in r and ("rccl_error" in r or "failed_stage" in r)
    else:
        assert "ms" in r["rollout_allgather"]


def test_bench_watchdog_fires_on_a_hung_stage():
    """a stage that exceeds its deadline (here: the overall deadline set below the warm-up's duration) produces the
    error line with `failed_stage` instead of a hang."""
    p, r = _bench(["--watchdog-s", "0.01"] + FAST + ["--min-region-ms", "3000"])
    assert p.returncode == 4 and r["value"] is None and "watchdog" in r["error"] and r["failed_stage"]


def test_bench_forced_collective_path_matches_the_plain_run():
    """PHX_BENCH_FORCE_DIST=1 python bench.py --gpus 1 (gloo control plane + RCCL world 1) reports the same workload
    and a `value` within noise of the plain single-process run; rollout buffers rotate over > 256 MB."""
    _, a = _bench(FAST + ["--min-region-ms", "300"])
    _, b = _bench(FAST + ["--min-region-ms", "300"], {"PHX_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1",
                                                       "MASTER_PORT": str(29700 + os.getpid() % 200)})
    assert "error" not in a and "error" not in b, (a.get("error"), b.get("error"))
    assert b["rccl_ranks_seen"] == 1 and a["rccl_ranks_seen"] == 0
    strip = lambda c: {k: v for k, v in c.items() if k != "autotune"}
    assert strip(a["config"]) == strip(b["config"]) and a["roofline"]["buffers_rotated"] >= 2
    assert a["roofline"]["buffers_rotated"] * a["roofline"]["algorithmic_bytes_per_launch"] > 256 << 20
    assert abs(a["value"] - b["value"]) / a["value"] < 0.5           # two processes, two sets of trajectory buffers (66-100 us per launch)
    assert b["rollout_allgather"]["bytes_per_rank"] < b["rollout_allgather"]["raw_trajectory_bytes_per_rank"]
    assert b["config4_share"]["agent_steps_per_sec_gather_included"] > 0


def test_rccl_collective_runs_with_one_rank_on_the_real_config4_buffers():
    """VERDICT r3 #4a: `all_gather_into_tensor` on the RCCL ("nccl") group with the real uint8 device buffers of config 4's
    per-GPU share (919 MB), forced although the world has ONE rank -- raw planes, the packed collection buffer and the
    chunked side-stream collector; bytes equal."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and lines, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    r = json.loads(lines[-1])
    assert r["ok"] and r["mode_raw"] == r["mode_gather"] == r["mode_pipeline"] == "collective:nccl", r
    assert r["raw_bytes"] >= 8192 * 100 * 51 * 22 and r["gather_bytes"] < r["raw_bytes"]


def test_bench_eight_ranks_sharing_one_gpu_print_one_line():
    """VERDICT r3 #4c: `bench.py --gpus 8` with every rank on the one GPU of this box (PHX_BENCH_SHARE_GPU=1) and a small batch:
    eight ranks spawn, the control plane (gloo) carries barrier and max-over-ranks, `value` is reported, ONE JSON line."""
    p, r = _bench(["--gpus", "8", "--batch", "256", "--min-region-ms", "200", "--no-other-configs", "--no-cpu-baseline", "--no-per-step",
                   "--no-frag200", "--steps", "20", "--warmup", "5"], {"PHX_BENCH_SHARE_GPU": "1"}, timeout=900)
    assert r["n_gpus"] == 8 and r["value"] and r["value"] > 0, r
    assert r["config"]["global_envs"] == 8 * 256
    assert sum(1 for l in p.stdout.splitlines() if l.startswith("{")) == 1


# ---- every rollout kernel variant against the oracle at small sizes -----------------------------
def _cmp_rollout(rd, ro, valid_planes):
    for k in ("obs", "actions", "rewards", "last_obs"):
        np.testing.assert_array_equal(f32_bits(rd[k]), f32_bits(ro[k]), err_msg=k)
    for k in ("truncated", "terminated") + (("obs_valid", "reward_valid") if valid_planes else ()):
        np.testing.assert_array_equal(rd[k], ro[k], err_msg=k)


PLAIN_VARIANTS = [
    # the round-3 kernel (rollout="time_parallel") in every workgroup shape
    ({"rollout": "time_parallel", "block": "whole_envs"}, "phx_sc_rollout_fast_kernel[whole_envs]"),
    ({"rollout": "time_parallel", "block": 32}, "phx_sc_rollout_fast_kernel[pairs]"),
    ({"rollout": "time_parallel", "block": 16}, "phx_sc_rollout_fast_kernel[pairs]"),
    ({"rollout": "time_parallel", "block": 48}, "phx_sc_rollout_fast_kernel[pairs]"),
    ({"rollout": "time_parallel"}, "phx_sc_rollout_fast_kernel"),
    # the round-4 store-wave kernel: the library's choice of workgroup, and 16 .. 144 pairs per workgroup
    ({}, "phx_sc_rollout_"),
    ({"rollout": "store_waves"}, "phx_sc_rollout_sw_kernel"),
    ({"rollout": "store_waves", "block": 16}, "phx_sc_rollout_sw_kernel"),
    ({"rollout": "store_waves", "block": 32}, "phx_sc_rollout_sw_kernel"),
    ({"rollout": "store_waves", "block": 48}, "phx_sc_rollout_sw_kernel"),
    ({"rollout": "store_waves", "block": 96}, "phx_sc_rollout_sw_kernel"),
    ({"rollout": "store_waves", "block": 128}, "phx_sc_rollout_sw_kernel"),
    ({"rollout": "store_waves", "block": 144}, "phx_sc_rollout_sw_kernel"),
    ({"rollout": "general"}, "phx_sc_rollout_kernel"),
    ({"rollout": "launch_loop"}, "phx_generic_step_kernel"),
]


@pytest.mark.parametrize("variants,kernel", PLAIN_VARIANTS, ids=[str(v) for v, _ in PLAIN_VARIANTS])
@pytest.mark.parametrize("S,K,B,num_steps", [(9, 6, 64, 100), (9, 6, 32, 23), (3, 2, 48, 40), (51, 4, 16, 100), (7, 3, 96, 30), (4, 4, 64, 50)])
def test_plain_rollout_variants_match_oracle(S, K, B, num_steps, variants, kernel):
    """phx_spec.variant_rollout / variant_block select the kernel PER ENV: the time-parallel kernel with whole-env
    workgroups and with workgroups of 16 / 32 / 48 consecutive (env, shop) pairs (the last block of an env's pair
    range is the one that writes its step counter: the last-arriver count), the round-1 kernel and the generic
    engine's launch loop -- unaligned ticks, ragged fragment lengths, several episode ends per fragment, poked stocks,
    hand-over to per-step launches -- all bit-equal to the oracle."""
    env = supply_chain_env(S, [K] * S, num_steps, B, seed=11 + S, env_offset=1000, variants=variants)
    o, d = OracleEnv(env.spec, threads=4), _dev(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(S * 100 + K)
    for t in range(3):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        o.step(a, None, None); d.step(a, None, None)
    used = set()
    for T in (1, 7, 20, 41, 100, 3, 200):
        ro, rd = o.rollout(T), d.rollout(T)
        used.add(d.dev.last_kernel())
        _cmp_rollout(rd, ro, False)
        for f in ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick"):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after T={T}")
    if "env.arrive" in d.dev.field_names():
        assert int(d.dev.field("env.arrive").abs().sum()) == 0      # the arrival counters are back at zero
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
    # the requested kernel ran wherever its preconditions hold (the plan refuses e.g. 48 pairs when B * S % 48 != 0)
    total = B * S
    blk = variants.get("block")
    applicable = not isinstance(blk, int) or (total % blk == 0 and (blk + S - 2) // S + 1 <= 255)
    if variants.get("rollout") == "store_waves":          # workgroups of a multiple of 16 pairs that divides B * S, chunk <= episode
        applicable = num_steps >= 16 and (any(total % g == 0 for g in range(16, 257, 16)) if blk is None else (blk % 16 == 0 and total % blk == 0))
    if blk == "whole_envs":
        epb = next((c for c in range(4, 256, 4) if c * S <= 96 and c * S >= 32), 4 if 4 * S <= 96 else 0)
        applicable = bool(epb) and B % epb == 0
    if applicable:
        assert any(kernel in u for u in used), (kernel, used)


@pytest.mark.parametrize("variants", [{"rollout": "store_waves"}, {"rollout": "store_waves", "block": 48}, {"rollout": "time_parallel"}],
                         ids=["store_waves", "store_waves48", "time_parallel"])
def test_rollout_with_step_counters_the_caller_moved(variants):
    """The store-wave kernel writes the flag planes of the WHOLE fragment before it streams the first chunk, from the step counters it finds
    at launch (row t ends a pair's episode <=> t == num_steps - 1 - step + j num_steps).  Counters a caller has moved -- ahead, behind,
    below zero (the first end is further away), at or above num_steps (the episode never ends: env.py:298 tests equality) -- give the
    oracle's trajectories all the same, also across several fragments and fragment lengths that are no multiple of anything."""
    S, K, B, ns = 9, 6, 64, 23
    env = supply_chain_env(S, [K] * S, ns, B, seed=5, env_offset=7, variants=variants)
    o, d = OracleEnv(env.spec, threads=4), _dev(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(99)
    st = rng.integers(0, ns, B).astype(np.int32)
    st[3], st[4], st[10], st[11], st[40] = -5, -40, ns, ns + 9, -1
    o.set_i32("env.step", st); d.set_i32("env.step", st)
    for T in (64, 41, 100, 7, 48):
        ro, rd = o.rollout(T), d.rollout(T)
        _cmp_rollout(rd, ro, False)
        for f in ("shop.stock", "shop.sales", "env.step", "env.tick"):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after T={T}")
    if variants.get("block") != 48:
        assert ("phx_sc_rollout_sw_kernel" if variants["rollout"] == "store_waves" else "phx_sc_rollout_fast_kernel") in d.dev.last_kernel()


FSM_VARIANTS = [
    ({"rollout": "time_parallel"}, "phx_sc_rollout_fsmfast_kernel"),
    ({"rollout": "time_parallel", "block": 32}, "phx_sc_rollout_fsmfast_kernel[pairs]"),
    ({"rollout": "lean"}, "phx_sc_rollout_fsm_lean_kernel"),
    ({"rollout": "store_waves"}, "phx_sc_rollout_sw_kernel[fsm]"),          # round 5: the store-wave kernel's FSM instantiation (tests/test_gpu_fsm_sw.py)
    ({"rollout": "general"}, "phx_sc_rollout_fsm_kernel"),
    ({"rollout": "launch_loop"}, "phx_generic_step_kernel"),
]


@pytest.mark.parametrize("variants,kernel", FSM_VARIANTS, ids=[str(v) for v, _ in FSM_VARIANTS])
@pytest.mark.parametrize("S,K,B,num_steps", [(9, 6, 64, 100), (51, 4, 16, 100), (3, 2, 48, 7), (5, 1, 40, 33)])
def test_fsm_rollout_variants_match_oracle(S, K, B, num_steps, variants, kernel):
    """the three FSM rollout kernels (time-parallel, lean lane-per-pair loop, general loop) and the launch loop, selected
    per env, on the edge cases of round 2's lean-loop test: fragments starting in either stage and on unaligned ticks,
    several episode ends, caches carried across launches, poked stocks, envs pushed off the stage chain."""
    env = supply_chain_env(S, [K] * S, num_steps, B, fsm=True, seed=5 + S, env_offset=77, variants=variants)
    o, d = OracleEnv(env.spec, threads=4), _dev(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(S * 10 + K)
    for t in range(3):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        o.step(a, None, None); d.step(a, None, None)
    fields = ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.stage", "env.step", "env.tick")
    used = set()
    for T in (1, 6, 41, 100, 3):
        ro, rd = o.rollout(T), d.rollout(T)
        used.add(d.dev.last_kernel())
        _cmp_rollout(rd, ro, True)
        for f in fields:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after T={T}")
    st = rng.integers(-40, 160, (B, S)).astype(np.int32)
    o.set_i32("shop.stock", st); d.set_i32("shop.stock", st)
    for T in (12, 5):
        ro, rd = o.rollout(T), d.rollout(T)
        _cmp_rollout(rd, ro, True)
    stg = o.get_i32("env.stage").copy()
    stg[::2] = 1 - stg[::2]
    o.set_i32("env.stage", stg); d.set_i32("env.stage", stg)
    for T in (9, 30):
        ro, rd = o.rollout(T), d.rollout(T)
        _cmp_rollout(rd, ro, True)
        for f in fields:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after an off-chain stage, T={T}")
    assert (d.err == 0).all()
    # (the time-parallel FSM kernel's plan needs num_steps >= 20 and <= 40 KB of LDS: 51-shop envs take 48-pair blocks that exceed it)
    # and a position table whose lookbacks fit; the SC64 shape meets every precondition)
    tp_ok = (S, K, num_steps) == (9, 6, 100)
    # (the store-wave instantiation: 16-pair-aligned workgroups, an even num_steps >= 16 -- the chain's last position observes)
    sw_ok = (B * S) % 16 == 0 and num_steps >= 20 and num_steps % 2 == 0
    if (variants["rollout"] != "time_parallel" or tp_ok) and (variants["rollout"] != "store_waves" or sw_ok):
        assert any(kernel in u for u in used), (kernel, used)


def test_config2_full_size_per_step_matches_oracle():
    """BASELINE configs[1] in the drop-in env.step() mode: SC64, B = 4096, twelve phx_step launches of the fused step
    kernel (device-RNG orders, random actions) against the oracle on every host core -- obs f32 / rewards f64 by bit
    pattern, flags, state (the oracle comparison stopped at B <= 192)."""
    B, S = 4096, 9
    env = supply_chain_env(S, [6] * S, 100, B, seed=42)
    o, d = OracleEnv(env.spec, threads=NCPU), _dev(env.spec)
    assert d.dev.uses_fused
    o.reset(); d.reset()
    rng = np.random.default_rng(1234)
    for t in range(12):
        a = (rng.random((B, S), dtype=np.float32) * 100.0).astype(np.float32)
        o.step(a, None, None); d.step(a, None, None)
        assert "phx_sc_step_kernel" in d.dev.last_kernel()
        np.testing.assert_array_equal(f32_bits(d.obs), f32_bits(o.obs), err_msg=f"obs t={t}")
        np.testing.assert_array_equal(f64_bits(d.reward), f64_bits(o.reward), err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d.truncated, o.truncated); np.testing.assert_array_equal(d.terminated, o.terminated)
    for f in ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick"):
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)
    assert (d.err == 0).all()


# ---- table-driven stage handlers in fused rollouts (fsm.py:294-307) --------------------------------
def test_tabulated_stage_handler_fused_step_and_rollout_reproduce_the_reference_golden():
    """golden `sc_fsm_handler` = the REFERENCE running a RESTOCK handler that restocks twice on every third step.  Declared
    state-independent (phantom_amd.state_independent) the handler is tabulated per (stage, clock) at spec-compile time;
    the FUSED kernels then take every transition themselves: (1) per-step launches of phx_sc_step_kernel with no host
    callback, (2) ONE fused phx_rollout launch replaying the golden's actions and draws -- stage sequence, key sets,
    observations (f32 bits), rewards and stock equal the reference's."""
    import torch
    from helpers import env_from_golden, golden
    g = golden("sc_fsm_handler")
    T, B = int(g["T"]), len(g["seeds"])
    env = env_from_golden(g, exogenous="device", tabulated_handlers=True)
    dev = env._device()
    assert not env._has_handlers and env.spec.stage_tab is not None and dev.uses_fused
    for t in range(T):                                             # (1) fused per-step launches
        if g["reset_before"][t].any():
            env.reset()
        a = torch.from_numpy(g["actions"][t]).to(dev.device)
        x = torch.from_numpy(g["exo"][t]).to(dev.device)
        env.step_tensors(a, None, x, check_errors=True)
        assert "phx_sc_step_kernel" in dev.last_kernel()
        np.testing.assert_array_equal(np.atleast_1d(env._h_stage), g["next_stage"][t], err_msg=f"host stage after t={t}")
        np.testing.assert_array_equal(dev.field("env.stage")[:, 0].cpu().numpy(), g["next_stage"][t])
        np.testing.assert_array_equal(dev.obs_valid.cpu().numpy(), g["obs_valid"][t], err_msg=f"obs_valid t={t}")
        np.testing.assert_array_equal(dev.reward_valid.cpu().numpy(), g["reward_valid"][t], err_msg=f"reward_valid t={t}")
        ov = g["obs_valid"][t].astype(bool)
        np.testing.assert_array_equal(f32_bits(dev.obs.cpu().numpy()[ov]), f32_bits(g["obs"][t][ov]))
        rv = g["reward_valid"][t] == 1
        np.testing.assert_array_equal(f64_bits(dev.reward.cpu().numpy()[rv]), f64_bits(g["reward"][t][rv]))
        np.testing.assert_array_equal(dev.field("shop.stock").cpu().numpy(), g["stock"][t])
    # (2) one fused rollout per episode stretch of the golden (the golden resets where its episodes end: auto-reset does too)
    env2 = env_from_golden(g, exogenous="device", tabulated_handlers=True)
    d2 = env2._device()
    env2.reset()
    resets = [t for t in range(T) if g["reset_before"][t].any()]
    assert resets[0] == 0 and all((r % int(g["num_steps"])) == 0 for r in resets), "golden episodes end at num_steps"
    acts = torch.from_numpy(np.ascontiguousarray(g["actions"])).to(d2.device)
    exo = torch.from_numpy(np.ascontiguousarray(g["exo"])).to(d2.device)
    tr = env2.rollout(T, actions=acts, exo=exo)
    assert "phx_sc_rollout_fsm_kernel" in d2.last_kernel()
    ovd, rvd = tr.obs_valid.cpu().numpy(), tr.reward_valid.cpu().numpy()
    np.testing.assert_array_equal(ovd, g["obs_valid"][:T])
    np.testing.assert_array_equal(rvd, g["reward_valid"][:T])
    ov = g["obs_valid"][:T].astype(bool)
    np.testing.assert_array_equal(f32_bits(tr.observations.cpu().numpy()[ov]), f32_bits(g["obs"][:T][ov]))
    rv = g["reward_valid"][:T] == 1
    np.testing.assert_array_equal(f32_bits(tr.rewards.cpu().numpy()[rv]), f32_bits(g["reward"][:T][rv].astype(np.float32)))
    np.testing.assert_array_equal(tr.truncations.cpu().numpy()[..., 0], g["all_truncated"][:T].astype(np.uint8))


@pytest.mark.parametrize("variants", [{}, {"rollout": "launch_loop"}, {"step": "generic"}], ids=str)
def test_tabulated_stage_handlers_random_tables_match_oracle(variants):
    """random (stage, clock) -> next-stage tables on a 3-stage FSM supply chain: fused step kernel, general FSM rollout
    kernel and the generic engine (per step and with the T-step loop in the kernel) against the oracle."""
    rng = np.random.default_rng(5)
    S, K, B, NS = 5, 3, 24, 17
    net_env = supply_chain_env(S, [K] * S, NS, B, fsm=True, seed=3)            # only for the agent ids
    shops = [a for a in net_env.agents if str(a).startswith("SHOP")]
    custs = [a for a in net_env.agents if str(a).startswith("CUST")]
    table = {(s, t): rng.integers(0, 3) for s in range(3) for t in range(NS + 1)}
    names = ["A", "B", "C"]

    def handler_for(si):
        return ph.state_independent(lambda env: names[int(table[(si, int(np.atleast_1d(env.current_step)[0]))])])

    stages = [ph.FSMStage("A", acting_agents=shops, rewarded_agents=shops, next_stages=names, handler=handler_for(0)),
              ph.FSMStage("B", acting_agents=custs, rewarded_agents=[], next_stages=names, handler=handler_for(1)),
              ph.FSMStage("C", acting_agents=shops + custs, rewarded_agents=None, next_stages=names, handler=handler_for(2))]
    from phantom_amd.supply_chain import build_network
    env = ph.FiniteStateMachineEnv(NS, build_network(S, [K] * S, ph.BatchResolver(), False), "A", stages=stages,
                                   batch_size=B, seed=3, exogenous="device", variants=variants)
    assert env.spec.stage_tab is not None
    o, d = OracleEnv(env.spec, threads=4), _dev(env.spec)
    o.reset(); d.reset()
    for t in range(5):
        a = rng.uniform(0, 100, (B, S)).astype(np.float32)
        o.step(a, None, None); d.step(a, None, None)
        np.testing.assert_array_equal(d.obs_valid, o.obs_valid)
        np.testing.assert_array_equal(f32_bits(d.obs), f32_bits(o.obs))
        np.testing.assert_array_equal(d.reward_valid, o.reward_valid)
        np.testing.assert_array_equal(f64_bits(d.reward), f64_bits(o.reward))
        np.testing.assert_array_equal(d.get_i32("env.stage"), o.get_i32("env.stage"))
    for T in (3, 40, 17):
        ro, rd = o.rollout(T), d.rollout(T)
        _cmp_rollout(rd, ro, True)
        for f in ("shop.stock", "env.stage", "env.step", "env.tick"):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after T={T}")
    assert (d.err == 0).all()


# ---- ADVICE r2 ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,S,K", [(1, 3, 2), (3, 1, 4), (5, 51, 1)])
def test_step_accepts_action_rows_at_any_4_byte_offset(B, S, K):
    """a row slice actions[t] of a [n, B, S] tensor starts on a 16-byte boundary only when B * S is a
    multiple of 4; no step kernel vector-loads actions, so phx_step / DeviceEnv.step_graph must take such rows."""
    import torch
    n = 6
    env = supply_chain_env(S, [K] * S, 50, B, seed=8)
    o, d = OracleEnv(env.spec), _dev(env.spec)
    o.reset(); d.reset()
    dev = d.dev
    acts = (torch.rand(n, B, S, device=dev.device) * 100.0).contiguous()
    assert (B * S) % 4 != 0 and acts[1].data_ptr() % 16 != 0
    for t in range(n):
        dev.step(acts[t])                                    # row slices straight into phx_step
        o.step(acts[t].cpu().numpy(), None, None)
    np.testing.assert_array_equal(dev.field("shop.stock").cpu().numpy(), o.get_i32("shop.stock"))
    g = dev.step_graph(acts)                                 # and captured: step i reads actions[i]
    g.replay(); torch.cuda.synchronize()
    for t in range(n):
        o.step(acts[t].cpu().numpy(), None, None)
    np.testing.assert_array_equal(dev.field("shop.stock").cpu().numpy(), o.get_i32("shop.stock"))
    np.testing.assert_array_equal(f32_bits(dev.obs.cpu().numpy()), f32_bits(o.obs))
    env2 = supply_chain_env(S, [K] * S, 50, B, seed=8, exogenous="device")
    env2.reset()
    env2.step_tensors(acts[1])                               # the Python surface with a row slice


def test_market_too_large_for_the_lds_rollout_falls_back_to_the_generic_loop():
    """a market with more than 3072 agents does not fit phx_stk_rollout_kernel's three agent slots per
    lane; phx_rollout materialises the price table and rolls out on the generic engine instead of returning an error."""
    from helpers import market_env
    L, Fw, B = 40, 3100, 2
    env = market_env(L, Fw, 2, 6, B, seed=4)
    o, d = OracleEnv(env.spec, threads=4), _dev(env.spec)
    assert d.dev.uses_fused
    o.reset(); d.reset()
    rng = np.random.default_rng(2)
    S = L + Fw
    a = rng.random((B, S), dtype=np.float32); av = np.zeros((B, S), np.uint8); av[:, :L] = 1
    o.step(a, av, None); d.step(a, av, None)                 # a fused market step first (compressed price slots)
    ro, rd = o.rollout(5), d.rollout(5)
    assert "phx_generic_step_kernel" in d.dev.last_kernel()
    _cmp_rollout(rd, ro, True)
    assert (d.err == 0).all()


def test_never_terminates_agrees_with_the_device_for_every_strategic_kind():
    """DeviceEnv.never_terminates() decides whether the `terminations` plane travels in a rollout
    collection; it must agree with what the device's is_terminated can return.  For every env family: if
    never_terminates() then no rollout ever sets a termination flag."""
    import phantom_amd as ph
    from helpers import market_env
    envs = [supply_chain_env(3, [2] * 3, 5, 4, seed=1), supply_chain_env(3, [2] * 3, 5, 4, fsm=True, seed=1),
            market_env(4, 8, 2, 6, 4, seed=1, exogenous="device"),
            ph.DigitalAdsEnv(num_steps=6, num_agents_theme={"travel": 2, "tech": 2}, batch_size=4, seed=3,
                             agent_supertypes={f"ADV_{i + 1}": ph.AdvertiserAgent.Supertype(budget=0.3) for i in range(4)})]
    seen_terminating = False
    for env in envs:
        d = _dev(env.spec); d.reset()
        r = d.rollout(30)
        if d.dev.never_terminates():
            assert int(r["terminated"].sum()) == 0
        else:
            seen_terminating = seen_terminating or int(r["terminated"].sum()) > 0
    assert seen_terminating                                   # the ads market's advertisers do run out of budget


def test_autotune_rollout_picks_a_candidate_and_keeps_the_trajectories():
    """PhantomEnv.autotune_rollout times the candidate block shapes on this GPU and rebuilds the device env with the
    fastest; whatever it picks, the trajectories are the oracle's."""
    env = supply_chain_env(9, [6] * 9, 100, 256, seed=21, env_offset=64, exogenous="device")
    res = env.autotune_rollout(100, candidates=[{"block": "whole_envs"}, {"block": 48}, {"block": 32}], launches=3)
    assert res["chosen"]["block"] in ("whole_envs", 48, 32) and len(res["us_per_launch"]) == 3
    assert env.spec.variants["block"] == res["chosen"]["block"]
    o = OracleEnv(env.spec, threads=4)
    o.reset(); env.reset()
    tr = env.rollout(100)
    ro = o.rollout(100)
    np.testing.assert_array_equal(f32_bits(tr.observations.cpu().numpy()), f32_bits(ro["obs"]))
    np.testing.assert_array_equal(f32_bits(tr.rewards.cpu().numpy()), f32_bits(ro["rewards"]))
    np.testing.assert_array_equal(tr.truncations.cpu().numpy(), ro["truncated"])
    assert env._device().last_kernel().startswith(("phx_sc_rollout_fast_kernel", "phx_sc_rollout_sw_kernel"))      # (48 pairs: the store-wave kernel)


def test_rollout_without_the_all_zero_terminations_plane():
    """phx_rollout_io.terminated may be NULL on the time-parallel supply-chain kernel (ShopAgent never terminates): every other
    plane equals the oracle's; envs / kernels that cannot leave the plane out refuse loudly."""
    env = supply_chain_env(9, [6] * 9, 100, 64, seed=17)
    o, d = OracleEnv(env.spec, threads=4), _dev(env.spec)
    o.reset(); d.reset()
    dev = d.dev
    tr = dev.alloc_trajectory(130, terminations=False)
    assert tr.terminations is None
    dev.rollout(130, out=tr)
    ro = o.rollout(130)
    np.testing.assert_array_equal(f32_bits(tr.observations.cpu().numpy()), f32_bits(ro["obs"]))
    np.testing.assert_array_equal(f32_bits(tr.rewards.cpu().numpy()), f32_bits(ro["rewards"]))
    np.testing.assert_array_equal(f32_bits(tr.actions.cpu().numpy()), f32_bits(ro["actions"]))
    np.testing.assert_array_equal(tr.truncations.cpu().numpy(), ro["truncated"])
    assert int(ro["terminated"].sum()) == 0
    envf = supply_chain_env(9, [6] * 9, 100, 64, fsm=True, seed=17)
    df = _dev(envf.spec); df.reset()
    trf = df.dev.alloc_trajectory(10, terminations=False)
    from phantom_amd.device import DeviceError
    with pytest.raises(DeviceError):
        df.dev.rollout(10, out=trf)


# ---- the generic engine's LEAN layout (scheduled two-wave supply chains: the dynamic steps' sort / scan scratch in the blob) -----
@pytest.mark.parametrize("fsm", [False, True])
@pytest.mark.parametrize("S,K", [(20, 4), (51, 4), (13, 5)])
def test_generic_engine_lean_layout_scheduled_and_dynamic_steps_match_oracle(S, K, fsm):
    """Supply chains with 64 < A <= 256 agents on the generic engine keep order / slot / scanbuf in a per-env workspace of the state
    blob (phx_generic.hip LEAN).  Steps with every action present follow the static schedule; a missing action or a done agent makes
    the step dynamic (sort and scan through the workspace).  Both, interleaved, against the oracle -- per step and as rollouts."""
    B = 48
    env = supply_chain_env(S, [K] * S, 30, B, fsm=fsm, force_generic=True, seed=5 + S, env_offset=300)
    o, d = OracleEnv(env.spec, threads=4), _dev(env.spec)
    assert "workspace" in d.dev.field_names()
    o.reset(); d.reset()
    rng = np.random.default_rng(S + K + fsm)
    for t in range(70):                                         # two episode ends
        a = rng.uniform(-20, 130, (B, S)).astype(np.float32)
        valid = None if t % 3 == 0 else (rng.random((B, S)) < 0.85).astype(np.uint8)       # t % 3 == 0: scheduled steps
        exo = rng.integers(0, 5, (B, S * K)).astype(np.uint8) if t % 2 else None
        o.step(a, valid, exo); d.step(a, valid, exo)
        np.testing.assert_array_equal(f32_bits(d.obs), f32_bits(o.obs), err_msg=f"obs t={t}")
        np.testing.assert_array_equal(f64_bits(d.reward), f64_bits(o.reward), err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d.truncated, o.truncated); np.testing.assert_array_equal(d.obs_valid, o.obs_valid)
        for f in ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step"):
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} t={t}")
        done = (o.all_truncated | o.all_terminated).astype(np.uint8)
        if done.any():
            o.reset(done); d.reset(done)
    assert d.dev.last_kernel() in ("phx_generic_step_kernel", "phx_sched_step_kernel")      # (round 6: the compiled schedule where the spec has one)
    for T in (1, 17, 45):
        ro, rd = o.rollout(T), d.rollout(T)
        _cmp_rollout(rd, ro, fsm)
    assert (d.err == 0).all()


# ---- ABI 7: per-env legacy-numpy MT19937 streams on the device (VERDICT r2 missing #5) ------------------------------------------
def test_mt19937_streams_reproduce_the_seeded_reference_run():
    """golden sc64 = four REFERENCE envs, each alone on the global numpy stream after np.random.seed(seeds[b]).  With
    exogenous="mt19937" and seed_streams(seeds) the device draws every instance's orders from its own MT19937: the draws equal the
    ones the reference's CustomerAgents consumed (supply_chain.py:64) and the run -- obs, rewards, stock, done flags -- equals the
    golden, with no exogenous input replayed from the host."""
    from helpers import golden
    g = golden("sc64")
    seeds = [int(s) for s in g["seeds"]]
    B, T = len(seeds), int(g["actions"].shape[0])
    env = supply_chain_env(9, [6] * 9, 100, B, exogenous="mt19937")
    d = _dev(env.spec)
    assert {"env.mt_state", "env.mt_pos"} <= set(d.dev.field_names())
    d.reset(); d.dev.mt_seed(seeds)
    for t in range(T):
        if t > 0 and g["reset_before"][t].any():
            d.reset(g["reset_before"][t])
        exo = d.dev.mt_draw(1)[0].cpu().numpy()
        np.testing.assert_array_equal(exo, g["exo"][t], err_msg=f"draws t={t}")
        d.step(g["actions"][t], None, exo)
        np.testing.assert_array_equal(f32_bits(d.obs), f32_bits(g["obs"][t]), err_msg=f"obs t={t}")
        np.testing.assert_array_equal(f64_bits(d.reward), f64_bits(g["reward"][t]), err_msg=f"reward t={t}")
        np.testing.assert_array_equal(d.get_i32("shop.stock"), g["stock"][t], err_msg=f"stock t={t}")
        np.testing.assert_array_equal(d.truncated, g["truncated"][t])
    assert (d.err == 0).all()


def test_mt19937_streams_equal_numpy_at_scale_and_feed_the_fused_rollout():
    """B = 4096 instances, one numpy stream each: phx_mt_draw over several calls (state regenerations inside and between them)
    equals np.random.RandomState(seed_b).randint(5, ...) for EVERY instance, and PhantomEnv.rollout in this mode equals the
    oracle replaying those draws."""
    B, S, K = 4096, 9, 6
    env = supply_chain_env(S, [K] * S, 100, B, exogenous="mt19937", seed=3)
    with pytest.raises(RuntimeError):
        env.rollout(5)                                          # streams not seeded yet
    seeds = (np.arange(B, dtype=np.uint64) * 2654435761 + 12345) % (1 << 32)
    env.reset(); env.seed_streams(seeds)
    dev = env._device()
    n = S * K
    chunks = [dev.mt_draw(T).cpu().numpy() for T in (1, 2, 130, 7)]
    got = np.concatenate(chunks, axis=0)                        # [140, B, n]
    for b in range(B):
        want = np.random.RandomState(int(seeds[b])).randint(5, size=got.shape[0] * n).astype(np.uint8).reshape(-1, n)
        if not np.array_equal(got[:, b], want):
            raise AssertionError(f"instance {b} (seed {int(seeds[b])}) leaves numpy's stream")
    # the same streams, continued, through the env surface: the rollout consumes the next 60 steps' draws
    o = OracleEnv(env.spec, threads=NCPU); o.reset()
    tr = env.rollout(60)
    exo = np.stack([np.random.RandomState(int(seeds[b])).randint(5, size=200 * n).astype(np.uint8).reshape(200, n)[140:200] for b in range(B)], axis=1)
    ro = o.rollout(60, actions=tr.actions.cpu().numpy(), exo=exo)
    np.testing.assert_array_equal(f32_bits(tr.observations.cpu().numpy()), f32_bits(ro["obs"]))
    np.testing.assert_array_equal(f32_bits(tr.rewards.cpu().numpy()), f32_bits(ro["rewards"]))


@pytest.mark.parametrize("name", ["sc_fsm_small", "sc256_fsm"])
def test_mt19937_streams_on_an_fsm_env_reproduce_the_seeded_reference_run(name):
    """FSM supply chains: only the customers of the env's stage call np.random.randint.  Step by step the device draws equal what
    the REFERENCE consumed in the seeded golden run and the run equals the golden; ONE phx_mt_draw(T), which walks the stages (and the
    reset at the episode's end) itself, gives the same draws; and at B = 512 a fused FSM rollout fed by phx_mt_draw(T) equals the
    same env stepped with per-step draws."""
    from helpers import env_from_golden, golden
    g = golden(name)
    seeds = [int(s) for s in g["seeds"]]
    B, T = len(seeds), int(g["actions"].shape[0])
    d = _dev(env_from_golden(g, exogenous="mt19937").spec)
    d.reset(); d.dev.mt_seed(seeds)
    for t in range(T):
        if t > 0 and g["reset_before"][t].any():
            d.reset(g["reset_before"][t])
        exo = d.dev.mt_draw(1)[0].cpu().numpy()
        np.testing.assert_array_equal(exo, g["exo"][t], err_msg=f"draws t={t}")
        d.step(g["actions"][t], None, exo)
        np.testing.assert_array_equal(d.get_i32("shop.stock"), g["stock"][t], err_msg=f"stock t={t}")
        np.testing.assert_array_equal(f32_bits(d.obs), f32_bits(g["obs"][t]), err_msg=f"obs t={t}")
    d2 = _dev(env_from_golden(g, exogenous="mt19937").spec)
    d2.reset(); d2.dev.mt_seed(seeds)
    np.testing.assert_array_equal(d2.dev.mt_draw(T).cpu().numpy(), g["exo"])
    # at scale, through the env surface: rollout(T) draws the T steps' words in one call
    S, K, Bb, Tr = int(g["n_shops"]), [int(k) for k in g["ks"]], 512, 2 * int(g["num_steps"]) + 3
    ea = supply_chain_env(S, K, int(g["num_steps"]), Bb, fsm=True, exogenous="mt19937", seed=9)
    eb = supply_chain_env(S, K, int(g["num_steps"]), Bb, fsm=True, exogenous="mt19937", seed=9)
    for e in (ea, eb):
        e.reset(); e.seed_streams(1000)
    tr = ea.rollout(Tr)
    db = eb._device()
    acts = tr.actions
    for t in range(Tr):
        exo = db.mt_draw(1)[0]
        out = db.step(acts[t].contiguous(), None, exo)
        np.testing.assert_array_equal(f32_bits(out.observations.cpu().numpy()), f32_bits(tr.observations[t].cpu().numpy()), err_msg=f"obs t={t}")
        done = (db.all_truncated | db.all_terminated)
        if bool(done.any()):
            db.reset(done)


def test_rollout_graph_replays_the_same_fragments_as_rollout_calls():
    """DeviceEnv.rollout_graph: consecutive phx_rollout fragments captured once into a hipGraph; two replays give the fragments that
    the same sequence of rollout() calls gives (the time-parallel kernel, and the generic engine's T-step loop)."""
    for kw in ({}, {"force_generic": True}, {"variants": {"rollout": "time_parallel"}}):
        ea = supply_chain_env(9, [6] * 9, 100, 64, seed=21, **kw)
        eb = supply_chain_env(9, [6] * 9, 100, 64, seed=21, **kw)
        for e in (ea, eb):
            e.reset()
        da, db = ea._device(), eb._device()
        bufs = [db.alloc_trajectory(37) for _ in range(3)]
        g = db.rollout_graph(37, bufs)
        for rep in range(2):
            g.replay()
            for i in range(3):
                tr = da.rollout(37)
                np.testing.assert_array_equal(f32_bits(bufs[i].observations.cpu().numpy()), f32_bits(tr.observations.cpu().numpy()), err_msg=f"{kw} replay {rep} fragment {i}")
                np.testing.assert_array_equal(f32_bits(bufs[i].rewards.cpu().numpy()), f32_bits(tr.rewards.cpu().numpy()))
                np.testing.assert_array_equal(bufs[i].truncations.cpu().numpy(), tr.truncations.cpu().numpy())


def test_bench_shape_fragment_sparse_flags_equal_dense_and_oracle_rows():
    """Round 4's bench launch shape (SC64, B = 4096, T = 400).  Round 3's kernel: the flag planes go out as one fill + the non-zero words
    at this size; round 4: the library's own choice for this shape is the
    store-wave kernel (144 pairs per workgroup, one workgroup per CU, dense flags from its store waves) -- every plane of the two
    bit-equal -- and the first episode's rows against the oracle."""
    import torch
    B, S, T = 4096, 9, 400
    ea = supply_chain_env(S, [6] * S, 100, B, seed=42, variants={"rollout": "time_parallel"})      # round-3 kernel: sparse at this size
    ec = supply_chain_env(S, [6] * S, 100, B, seed=42)                                             # the library's choice
    for e in (ea, ec):
        e.reset()
    ta = ea._device().rollout(T)
    assert "phx_zero_fill_kernel[flag planes]" in ea._device().last_kernel()      # (the calling thread's LAST call)
    tc = ec._device().rollout(T)
    assert ec._device().last_kernel() == "phx_sc_rollout_sw_kernel"
    for name, x, z in zip(ta._fields[:6], ta[:6], tc[:6]):
        assert torch.equal(x.contiguous().view(torch.uint8), z.contiguous().view(torch.uint8)), name
    for f in ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick"):
        assert torch.equal(ea._device().field(f), ec._device().field(f)), f
    t1 = ec._device().rollout(20)                                      # a fragment shorter than the pipeline fill goes back to round 3's kernel, same stream of steps
    assert "phx_sc_rollout_fast_kernel" in ec._device().last_kernel()
    t1a = ea._device().rollout(20)
    for name, x, y in zip(t1._fields[:6], t1[:6], t1a[:6]):
        assert torch.equal(x.contiguous().view(torch.uint8), y.contiguous().view(torch.uint8)), name
    t1 = ec._device().rollout(100)
    assert ec._device().last_kernel() == "phx_sc_rollout_sw_kernel"
    t1a = ea._device().rollout(100)
    for name, x, y in zip(t1._fields[:6], t1[:6], t1a[:6]):
        assert torch.equal(x.contiguous().view(torch.uint8), y.contiguous().view(torch.uint8)), name
    assert int(ta.truncations.sum()) == 4 * B * S and int(ta.terminations.sum()) == 0
    o = OracleEnv(ea.spec, threads=NCPU); o.reset()
    ro = o.rollout(100)
    np.testing.assert_array_equal(f32_bits(ta.observations[:100].cpu().numpy()), f32_bits(ro["obs"]))
    np.testing.assert_array_equal(ta.truncations[:100].cpu().numpy(), ro["truncated"])



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
    assert abs(a["value"] - b["value"]) / a["value"] < 0.35          # two processes, two buffer placements
    assert b["rollout_allgather"]["bytes_per_rank"] < b["rollout_allgather"]["raw_trajectory_bytes_per_rank"]
    assert b["config4_share"]["agent_steps_per_sec_gather_included"] > 0

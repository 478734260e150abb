"""CPU: bench.py's "cannot fail silently" contract where no GPU is needed to show it -- a run
that cannot work still ends with exactly ONE JSON line carrying n_gpus, rccl_ranks_seen and an `error` field."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=300):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, (p.stdout[-1000:], p.stderr[-1000:])
    return p, json.loads(lines[0])


@pytest.mark.skipif(torch.cuda.is_available(), reason="the no-GPU failure path")
def test_single_rank_without_gpu_prints_error_line():
    p, r = _run(["--steps", "20", "--warmup", "5"])
    assert p.returncode != 0 and r["value"] is None and r["n_gpus"] == 1 and "no CPU fallback" in r["error"]


@pytest.mark.skipif(torch.cuda.is_available(), reason="the no-GPU failure path")
def test_two_ranks_without_gpu_print_one_error_line():
    """self-spawned torchrun with two ranks that both fail (or one fails and the launcher SIGTERMs the other, which is
    taken by the sigwait thread): still one line, n_gpus = 2."""
    p, r = _run(["--gpus", "2", "--steps", "20", "--warmup", "5", "--watchdog-s", "120"])
    assert p.returncode != 0 and r["value"] is None and r["n_gpus"] == 2 and r["rccl_ranks_seen"] == 0 and r["error"]


def test_error_line_keeps_partial_measurements():
    sys.path.insert(0, ROOT)
    import bench
    line = bench.error_line(8, "boom", stage="rollout_allgather", partial={"value": 1.0e12, "rccl_ranks_seen": 8, "steps": 20})
    assert line["value"] == 1.0e12 and line["n_gpus"] == 8 and line["rccl_ranks_seen"] == 8
    assert line["error"] == "boom" and line["failed_stage"] == "rollout_allgather" and line["metric"] == "agent_steps_per_sec"

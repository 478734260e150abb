"""Round 6 on the GPU, part 1 (VERDICT r5 "Next round" #1): the parity hole at the timed shapes -- the exact call bench.py times (SC64,
B = 4096, 8 fragments x T = 100), config 4's share as 4 x 100 fragments at B = 8192, vouched replays at B = 4096 / T = 400 -- each against
the oracle at FULL size, and a replay hint that does not hold reported as PHX_ERR_HINT instead of silently unspecified rows."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from phantom_amd import _abi
from device_runner import DeviceRunner
from helpers import f32_bits, supply_chain_env
from oracle import OracleEnv

pytestmark = pytest.mark.gpu
NCPU = min(os.cpu_count() or 1, 128)
STATE = ("shop.stock", "shop.sales", "shop.missed_sales", "shop.delivered_stock", "env.step", "env.tick")


def _planes(tr):
    return dict(obs=tr.observations.cpu().numpy(), actions=tr.actions.cpu().numpy(), rewards=tr.rewards.cpu().numpy(),
                terminated=None if tr.terminations is None else tr.terminations.cpu().numpy(), truncated=tr.truncations.cpu().numpy())


def test_the_call_the_bench_times_every_row_of_every_fragment_against_the_oracle():
    """bench.py's timed call: SC64 (9 shops x 6 customers), B = 4096, num_steps = 100, ``rollout_fragments(100, 8 buffers)`` = 800 steps
    in ONE phx_sc_rollout_sw_kernel launch on 256 workgroups (XCD-aware pair-range mapping, base-pointer switches at every fragment
    boundary at full grid): every row of every plane of every fragment, ``last_obs`` and the state the call leaves, bit for bit; twice
    (the second call starts from the first one's state, as in the bench's timed loop, and rotates to a second buffer set)."""
    B, S, K, Tf, k = 4096, 9, 6, 100, 8
    env = supply_chain_env(S, [K] * S, 100, B, seed=42)                 # bench.py: seed 42, the library's own choice of kernel
    o, d = OracleEnv(env.spec, threads=NCPU), DeviceRunner(env.spec)
    o.reset(); d.reset()
    dev = d.dev
    sets = [[dev.alloc_trajectory(Tf) for _ in range(k)] for _ in range(2)]
    for rep in range(2):
        outs = sets[rep]
        for t in outs:
            for x in t[:5]:
                x.fill_(7)                                               # garbage the launch has to overwrite
        got = dev.rollout_fragments(Tf, outs)
        assert dev.last_kernel() == "phx_sc_rollout_sw_kernel", dev.last_kernel()
        assert all(g.last_obs is None for g in got[:-1]) and got[-1].last_obs is outs[-1].last_obs
        ro = o.rollout(k * Tf)
        for i, t in enumerate(outs):
            p = _planes(t)
            lo, hi = i * Tf, (i + 1) * Tf
            for key in ("obs", "actions", "rewards"):
                np.testing.assert_array_equal(f32_bits(p[key]), f32_bits(ro[key][lo:hi]), err_msg=f"rep {rep} fragment {i}: {key}")
            np.testing.assert_array_equal(p["truncated"], ro["truncated"][lo:hi], err_msg=f"rep {rep} fragment {i}: truncated")
            np.testing.assert_array_equal(p["terminated"], ro["terminated"][lo:hi], err_msg=f"rep {rep} fragment {i}: terminated")
        np.testing.assert_array_equal(f32_bits(outs[-1].last_obs.cpu().numpy()), f32_bits(ro["last_obs"]))
        for f in STATE:
            np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f"{f} after rep {rep}")
    assert (dev.err.cpu().numpy() == 0).all()


def test_config4_share_as_four_fragments_full_size_sample_and_invariants():
    """BASELINE config 4's per-GPU share (SC256: 51 shops x 4 customers, B = 8192) collected as 4 x 100-step fragments from one launch
    (3 264 groups of 128 pairs walked by 256 persistent workgroups): the first 512 envs of every fragment row by row against an oracle
    of those envs (the device RNG is keyed by the global env index), the whole batch through the planes' closed forms (one truncation
    per pair and episode at the episode's last row, no termination, actions in [0, 100), stock/100 observations in [0, 1]) and the
    step counters the call leaves."""
    S, K, B, ns, Tf, k, Bo = 51, 4, 8192, 100, 100, 4, 512
    env = supply_chain_env(S, [K] * S, ns, B, seed=42)
    d = DeviceRunner(env.spec); d.reset()
    rd = d.rollout_fragments(Tf, k)
    assert d.dev.last_kernel() == "phx_sc_rollout_sw_kernel", d.dev.last_kernel()
    envo = supply_chain_env(S, [K] * S, ns, Bo, seed=42)
    o = OracleEnv(envo.spec, threads=NCPU); o.reset()
    ro = o.rollout(k * Tf)
    for key in ("obs", "actions", "rewards"):
        np.testing.assert_array_equal(f32_bits(rd[key][:, :Bo]), f32_bits(ro[key]), err_msg=key)
    for key in ("truncated", "terminated"):
        np.testing.assert_array_equal(rd[key][:, :Bo], ro[key], err_msg=key)
    np.testing.assert_array_equal(f32_bits(rd["last_obs"][:Bo]), f32_bits(ro["last_obs"]))
    for f in STATE:
        np.testing.assert_array_equal(d.get_i32(f)[:Bo], o.get_i32(f), err_msg=f)
    tr = rd["truncated"]
    assert tr.sum() == k * B * S and all((tr[j * ns + ns - 1] == 1).all() for j in range(k)) and not rd["terminated"].any()
    assert (rd["actions"] >= 0).all() and (rd["actions"] < 100).all()
    assert (rd["obs"][..., 0] >= 0).all() and (rd["obs"][..., 0] <= 1).all()
    assert (d.get_i32("env.step") == 0).all() and (d.get_i32("env.tick") == k * Tf).all()
    assert (d.err == 0).all()


@pytest.mark.parametrize("what", ["actions", "both"])
def test_vouched_replays_at_the_bench_batch_full_size(what):
    """Replayed policy (and order sizes) the caller vouches for through the store-wave kernel's REPLAY instantiation at the bench's
    batch: SC64, B = 4096, T = 400 -- every row against the oracle (round 5 ran replays at B <= 128)."""
    B, S, K, T = 4096, 9, 6, 400
    env = supply_chain_env(S, [K] * S, 100, B, seed=42)
    o, d = OracleEnv(env.spec, threads=NCPU), DeviceRunner(env.spec)
    o.reset(); d.reset()
    rng = np.random.default_rng(6)
    acts = rng.uniform(0, 100, (T, B, S)).astype(np.float32)
    acts[rng.random((T, B, S)) < 0.05] = 0.5
    acts[rng.random((T, B, S)) < 0.05] = 117.25
    exo = rng.integers(0, 5, (T, B, d.n_exo)).astype(np.uint8) if what == "both" else None
    rd = d.rollout(T, acts, exo, actions_in_domain=True, exo_in_domain=exo is not None)
    assert d.dev.last_kernel() == "phx_sc_rollout_sw_kernel[replay]", d.dev.last_kernel()
    ro = o.rollout(T, acts, exo)
    for key in ("obs", "actions", "rewards"):
        np.testing.assert_array_equal(f32_bits(rd[key]), f32_bits(ro[key]), err_msg=key)
    np.testing.assert_array_equal(rd["truncated"], ro["truncated"])
    np.testing.assert_array_equal(f32_bits(rd["last_obs"]), f32_bits(ro["last_obs"]))
    for f in STATE:
        np.testing.assert_array_equal(d.get_i32(f), o.get_i32(f), err_msg=f)
    assert (d.err == 0).all()


@pytest.mark.parametrize("S,K,B,T", [(9, 6, 64, 48), (51, 4, 128, 64), (3, 2, 48, 41)])
def test_a_replay_hint_that_does_not_hold_is_reported_per_env(S, K, B, T):
    """PHX_RH_ACTIONS_IN_DOMAIN with an action that rounds below zero, PHX_RH_EXO_IN_DOMAIN with an order byte >= 5: the store-wave
    kernel relies on both (byte tiles) -- err[b] == PHX_ERR_HINT for exactly the envs whose inputs break the promise, every other env's
    rows are the oracle's; a following honest call is clean again after the caller's reset."""
    env = supply_chain_env(S, [K] * S, 30, B, seed=8, variants={"rollout": "store_waves"})
    o, d = OracleEnv(env.spec, threads=8), DeviceRunner(env.spec)
    rng = np.random.default_rng(S + T)
    for what in ("actions", "exo", "both", "honest"):
        o.reset(); d.reset()
        acts = rng.uniform(0, 100, (T, B, S)).astype(np.float32)
        exo = rng.integers(0, 5, (T, B, d.n_exo)).astype(np.uint8)
        bad = np.zeros(B, bool)
        if what in ("actions", "both"):
            for b, t, s, v in ((3, 5, 0, -0.51), (B - 1, T - 1, S - 1, -42.0), (B // 2, 0, S // 2, -1e9)):
                acts[t, b, s] = v; bad[b] = True
        if what in ("exo", "both"):
            for b, t, c, v in ((1, 7, 0, 5), (B - 2, T - 2, d.n_exo - 1, 255), (B // 2 + 1, 1, d.n_exo // 2, 7), (7, 3, 1, 29), (9, 11, 2, 13)):
                exo[t, b, c] = v; bad[b] = True
        rd = d.rollout(T, acts, exo, actions_in_domain=True, exo_in_domain=True)
        assert d.dev.last_kernel() == "phx_sc_rollout_sw_kernel[replay]", d.dev.last_kernel()
        np.testing.assert_array_equal(d.err != 0, bad, err_msg=what)
        assert (d.err[bad] == _abi.ERR_HINT).all()
        ro = o.rollout(T, acts, exo)
        good = ~bad
        for key in ("obs", "actions", "rewards"):
            np.testing.assert_array_equal(f32_bits(rd[key][:, good]), f32_bits(ro[key][:, good]), err_msg=f"{what}: {key}")
        np.testing.assert_array_equal(d.get_i32("shop.stock")[good], o.get_i32("shop.stock")[good])


def test_scale_node_script_dry_run_on_one_gpu(tmp_path):
    """tools/scale_node.sh --dry-run: the first-contact script for an 8-GPU node with every rank on the GPU that exists
    (PHX_BENCH_SHARE_GPU): each stage leaves one JSON line (a bench line or an error line), the 1 / 2 / 8-rank curve stages report a `value`
    over the gloo control plane, the NCCL_ALGO stage records its environment, and the summary tabulates them."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    out = str(tmp_path / "scale")
    p = subprocess.run(["bash", os.path.join(root, "tools", "scale_node.sh"), "--dry-run", out], capture_output=True, text=True, timeout=2400,
                       env={**os.environ, "STAGE_TIMEOUT": "420"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = {}
    for name in ("curve_n1", "curve_n2", "curve_n8", "gather_ring_n8"):
        lines[name] = json.load(open(os.path.join(out, name + ".json")))
    for name, n in (("curve_n1", 1), ("curve_n2", 2), ("curve_n8", 8)):
        d = lines[name]
        assert d.get("n_gpus") == n and d.get("value") and d["value"] > 0, (name, d.get("error"), p.stdout[-1500:])
    assert lines["gather_ring_n8"].get("n_gpus") == 8
    summ = open(os.path.join(out, "summary.txt")).read()
    assert "== scaling" in summ and "N=8" in summ and "NCCL_ALGO=Ring" in summ

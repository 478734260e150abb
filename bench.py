#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its configs[1]:
    supply-chain 64 agents (1 factory + 9 shops + 54 customers), batch = 4096 envs per GPU,
    random actions, device-RNG customer orders.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one PhantomEnv.step() of every env instance of the batch.  The timed region runs the
K-step region R times back to back (R chosen so that it lasts >= 1 s; printed as "repeats") as
fused on-device rollouts (phx_rollout, T=400 = four episodes per launch by default; every step's observation,
action, reward and done flags are written to the trajectory buffer in HBM) with inputs/state
already resident in HBM, bracketed by barrier + synchronize, max over ranks.
value = A * B_total * R * K / time  (agent-steps/s, whole job); ms_per_step = time / (R * K).
The launches rotate over >= 4 trajectory buffers (> 256 MB in total, more than the Infinity Cache holds) so that the
written bytes really reach HBM.  `--gpus N` without a torchrun environment re-executes itself under
torch.distributed.run.  The per-launch PhantomEnv.step mode (one kernel per step) is measured right after and
reported under "per_step".

Multi-rank runs cannot fail silently: barrier / max-over-ranks go over a gloo control group (the step path has no
collective, so `value` does not depend on RCCL at all), RCCL is brought up only for the rollout-collection
sections, and a watchdog thread plus SIGTERM / exception handlers make rank 0 print the JSON line with whatever
was measured, `rccl_ranks_seen` and an `error` field when a stage hangs, RCCL fails or the launcher tears the job down.

Extra objects on the JSON line:
  roofline      dominant kernel (phx_sc_rollout_kernel): algorithmic HBM bytes per launch /
                mean launch duration from HIP events on the launch stream, vs 8 TB/s.
  cpu_baseline  the CPU oracle (C restatement of the reference's algorithm, kind "port") timed
                on this box's host cores on a bounded sample of the same workload (rank 0, N=1).
"""
import argparse
import json
import os
import signal
import sys
import time

# SIGTERM is blocked before any library creates a thread (numpy's BLAS pool, torch): rank 0 takes it in a dedicated
# sigwait thread (class Watch) so that the JSON line is printed even while the main thread is blocked in a C call.
if __name__ == "__main__":
    try:
        signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})
    except (AttributeError, ValueError, OSError):
        pass

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_SHOPS, CUST_PER_SHOP, NUM_STEPS, BATCH = 9, 6, 100, 4096
N_AGENTS = 1 + N_SHOPS + N_SHOPS * CUST_PER_SHOP          # 64
# BASELINE.json configs[1] is the bench line; --config sc256 runs configs[3]'s per-GPU workload
CONFIGS = {"sc64": (9, 6, 4096), "sc256": (51, 4, 8192)}
HBM_PEAK_GBS = 8000.0                                      # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes_rollout(B, S, T):
    """trajectory record per env-step: obs f32*3S + action f32*S + reward f32*S +
    terminated/truncated u8*2S = 22*S bytes, written once; plus the shop state read and written
    once per launch (4 x i32 each way per shop) and step/tick per env."""
    return B * T * 22 * S + B * (S * 32 + 16)


def algorithmic_bytes_step(B, S, K, device_rng=True):
    """per-launch mode: S*(43+K)+6 bytes per env-step with replayed draws,
    S*43+6 with the device RNG."""
    return B * (S * (43 + (0 if device_rng else K)) + 6)


def cpu_baseline(budget_s=12.0):
    """time the CPU oracle (oracle/phx_oracle.c, the sequential C restatement of the
    reference's algorithm) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import supply_chain_env
    from oracle import OracleEnv, lib
    cores = os.cpu_count() or 1
    B = 256
    env = supply_chain_env(N_SHOPS, [CUST_PER_SHOP] * N_SHOPS, NUM_STEPS, B, seed=42)
    o = OracleEnv(env.spec, threads=1)
    o.reset()
    t0 = time.perf_counter(); o.rollout(20); dt = time.perf_counter() - t0
    rate1 = B * 20 / dt                                      # env-steps/s on one core (probe)
    # single core leg ~ 1/3 of the budget, all-core leg ~ 2/3
    T1 = max(20, int(rate1 * budget_s / 3 / B))
    t0 = time.perf_counter(); o.rollout(T1); dt1 = time.perf_counter() - t0
    single = N_AGENTS * B * T1 / dt1
    threads = min(cores, lib().phxo_max_threads())
    Bn = 256 * threads
    envn = supply_chain_env(N_SHOPS, [CUST_PER_SHOP] * N_SHOPS, NUM_STEPS, Bn, seed=42)
    on = OracleEnv(envn.spec, threads=threads)
    on.reset()
    t0 = time.perf_counter(); on.rollout(10); dtp = time.perf_counter() - t0     # all-thread probe
    Tn = max(10, int(10 * (budget_s * 2 / 3) / max(dtp, 1e-3)))
    t0 = time.perf_counter(); on.rollout(Tn); dtn = time.perf_counter() - t0
    multi = N_AGENTS * Bn * Tn / dtn
    return {"value": multi, "unit": "agent-steps/s", "cores": threads, "kind": "port",
            "sample": f"SC64, {Bn} envs x {Tn} steps on {threads} threads ({dtn:.1f} s); "
                      f"single core: {B} envs x {T1} steps ({dt1:.1f} s)",
            "single_core_value": single, "host_cpu_count": cores, "nproc": cores, "cpu_model": cpu_model(),
            # the reference's own Python PhantomEnv (SupplyChain agent classes, synthetic SC64) cannot
            # travel to this box; measured once in the build container on 1 core of an 8-vCPU host
            "interpreted_reference": {"env_steps_per_sec": 1060.0, "agent_steps_per_sec": 6.8e4, "cores": 1,
                                      "where": "build container (NOT this box), reference Python PhantomEnv imported from "
                                               "/root/reference, SURVEY section 6 / BASELINE.md section 2"}}


def box_info():
    """What can differ between two leases of "an MI355X" (VERDICT r4 #5): position in the node, serial, partition modes, firmware and
    driver versions, clocks and power at the end of the run -- recorded beside the number so that the two kinds of box the rollout
    kernel sees (53 vs 59-65 us per T = 400 launch at equal clocks and equal plain-fill rate) can be told apart over time."""
    import glob
    import re
    import subprocess
    info = {}

    def sh(cmd):
        try:
            return subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout
        except Exception:
            return ""

    st = sh(["amd-smi", "static", "--gpu", "0"]) or sh(["amd-smi", "static"])
    for key in ("MARKET_NAME", "ASIC_SERIAL", "OAM_ID", "NUM_COMPUTE_UNITS", "PART_NUMBER", "BUILD_DATE", "SOCKET_POWER_LIMIT", "MODEL_NUMBER",
                "PRODUCT_SERIAL", "COMPUTE_PARTITION", "MEMORY_PARTITION", "VRAM_VENDOR", "VRAM_SIZE", "BIT_WIDTH", "MAX_BANDWIDTH"):
        m = re.search(r"^\s*" + key + r":\s*(.+)$", st, re.M)
        if m:
            info[key.lower()] = m.group(1).strip()
    m = re.search(r"DRIVER:\s*\n\s*NAME:\s*(\S+)\s*\n\s*VERSION:\s*(.+)", st)
    if m:
        info["driver"] = (m.group(1) + " " + m.group(2).strip())[:120]
    for name in ("current_memory_partition", "current_compute_partition", "vbios_version"):
        for f in sorted(glob.glob(f"/sys/class/drm/card*/device/{name}"))[:1]:
            try:
                info[name] = open(f).read().strip()
            except OSError:
                pass
    fw = sh(["rocm-smi", "--showfwinfo"])
    info["firmware"] = {m.group(1).strip(): m.group(2).strip() for m in re.finditer(r"GPU\[0\]\s*:\s*(.+?) firmware version:\s*(\S+)", fw)}
    met = sh(["amd-smi", "metric", "--gpu", "0", "--clock", "--power", "--temperature", "--ecc"])
    for key, pat in (("gfx_clk", r"GFX_0:\s*\n\s*CLK:\s*(.+)"), ("mem_clk", r"MEM_0:\s*\n\s*CLK:\s*(.+)"), ("socket_power", r"SOCKET_POWER:\s*(.+)"),
                     ("hotspot_temp", r"HOTSPOT:\s*(.+)"), ("mem_temp", r"\bMEM:\s*(\d.+)"), ("ecc_total_correctable", r"TOTAL_CORRECTABLE_COUNT:\s*(.+)")):
        m = re.search(pat, met)
        if m:
            info[key] = m.group(1).strip()
    try:
        pr = torch.cuda.get_device_properties(0)
        info["torch"] = {"name": pr.name, "cus": pr.multi_processor_count, "total_memory": pr.total_memory, "torch": torch.__version__, "hip": torch.version.hip}
    except Exception:
        pass
    return info


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def other_configs(device):
    """per-step time of BASELINE.json's configs 3, 4 (one GPU's share) and 5 -- not part of `value`."""
    import phantom_amd as ph
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import market_env
    res = []

    def timed(fn, n):
        for _ in range(max(2, n // 4)):                          # (the first ~10 ms after idle run at ramping clocks: 63 against 53 us per SC64 launch)
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3                    # seconds per call

    def add(name, agents, B, steps_per_call, sec, mode, bytes_per_env_step=None):
        r = {"config": name, "mode": mode, "agents": agents, "envs": B,
             "us_per_step": sec / steps_per_call * 1e6,
             "agent_steps_per_sec": agents * B * steps_per_call / sec}
        if bytes_per_env_step:                                   # algorithmic HBM bytes against the 8 TB/s peak
            gbps = bytes_per_env_step * B * steps_per_call / sec / 1e9
            r["hbm"] = {"algorithmic_bytes_per_env_step": bytes_per_env_step, "achieved_GBps": gbps, "frac": gbps / HBM_PEAK_GBS}
        res.append(r)

    # config 3: SC256 (1 + 51 + 204 agents), 2-stage FSM, B = 8192
    env = ph.SupplyChainFSMEnv(n_shops=51, customers_per_shop=4, num_steps=100, batch_size=8192, seed=42,
                               exogenous="device", device=device)
    env.reset(); dev = env._device()
    tr = dev.rollout(100)
    add("SC256 FSM B=8192 (config 3)", 256, 8192, 100, timed(lambda: dev.rollout(100, out=tr), 40), "fused FSM rollout T=100",
        bytes_per_env_step=24 * 51)                              # trajectory: obs 12 + action 4 + reward 4 + 4 flag bytes per shop
    res[-1]["kernels"] = dev.last_kernel()
    tr4 = dev.alloc_trajectory(400)                              # the headline's fragment length: start-up / drain paid once per 400 steps
    add("SC256 FSM B=8192 (config 3)", 256, 8192, 400, timed(lambda: dev.rollout(400, out=tr4), 16), "fused FSM rollout T=400",
        bytes_per_env_step=24 * 51)
    res[-1]["kernels"] = dev.last_kernel()                       # (PHX_VR_AUTO: the store-wave kernel's FSM instantiation from T = 200 on)
    del tr4
    fr3 = [dev.alloc_trajectory(100) for _ in range(4)]
    add("SC256 FSM B=8192 (config 3)", 256, 8192, 400, timed(lambda: dev.rollout_fragments(100, fr3), 16),
        "fused FSM rollout, 4 fragments of T=100 per call (phx_rollout_io.frags)", bytes_per_env_step=24 * 51)
    res[-1]["kernels"] = dev.last_kernel()
    del fr3
    try:                                                         # the lane-per-pair loop on the same launch (variants={"rollout": "lean"}): bimodal by the buffers' placement
        envl = ph.SupplyChainFSMEnv(n_shops=51, customers_per_shop=4, num_steps=100, batch_size=8192, seed=42, exogenous="device", device=device,
                                    variants={"rollout": "lean"})
        envl.reset(); devl = envl._device()
        trl = devl.alloc_trajectory(400)
        add("SC256 FSM B=8192 (config 3)", 256, 8192, 400, timed(lambda: devl.rollout(400, out=trl), 16), "fused FSM rollout T=400, lane-per-pair loop",
            bytes_per_env_step=24 * 51)
        res[-1]["kernels"] = devl.last_kernel()
        del envl, devl, trl
    except Exception as exc:
        res.append({"config": "SC256 FSM B=8192 (config 3)", "mode": "lane-per-pair loop", "error": str(exc)[:200]})
    acts = torch.rand(8192, 51, device=dev.device) * 100.0
    # per env-step: S * (43 + K) + 6 with device-RNG orders (no exo term: S * 43 + 6), + stage and valid planes
    step_bytes = 51 * 43 + 6 + 2 * 51 + 2
    add("SC256 FSM B=8192 (config 3)", 256, 8192, 1, timed(lambda: dev.step(acts), 100), "one launch per step",
        bytes_per_env_step=step_bytes)
    try:                                                         # the same launches replayed from a hipGraph (DeviceEnv.step_graph)
        a50 = (torch.rand(50, 8192, 51, device=dev.device) * 100.0).contiguous()
        sg = dev.step_graph(a50)
        add("SC256 FSM B=8192 (config 3)", 256, 8192, 50, timed(sg.replay, 6), "one launch per step, hipGraph of 50 steps",
            bytes_per_env_step=step_bytes)
        del sg, a50
    except Exception as exc:                                     # graph capture is an optimisation, not the product path
        res.append({"config": "SC256 FSM B=8192 (config 3)", "mode": "hipGraph", "error": str(exc)[:200]})
    del env, dev, tr, acts; torch.cuda.empty_cache()
    # config 4, one GPU's share: SC256 plain, B = 8192 per GPU, rollout T = 100
    env = ph.SupplyChainEnv(n_shops=51, customers_per_shop=4, num_steps=100, batch_size=8192, seed=42,
                            exogenous="device", device=device)
    env.reset(); dev = env._device()
    tr = dev.rollout(100)
    add("SC256 B=8192 per GPU (config 4)", 256, 8192, 100, timed(lambda: dev.rollout(100, out=tr), 40), "fused rollout T=100",
        bytes_per_env_step=22 * 51)
    tr4 = dev.alloc_trajectory(400)
    add("SC256 B=8192 per GPU (config 4)", 256, 8192, 400, timed(lambda: dev.rollout(400, out=tr4), 16), "fused rollout T=400",
        bytes_per_env_step=22 * 51)
    fr = [dev.alloc_trajectory(100) for _ in range(4)]
    add("SC256 B=8192 per GPU (config 4)", 256, 8192, 400, timed(lambda: dev.rollout_fragments(100, fr), 16),
        "fused rollout, 4 fragments of T=100 per call (phx_rollout_io.frags)", bytes_per_env_step=22 * 51)
    del env, dev, tr, tr4, fr; torch.cuda.empty_cache()
    # BASELINE config 2's env with a RECORDED policy and recorded order sizes (phx_rollout_io.actions / exo: a learned policy's actions
    # replayed, the reference's numpy stream): the store-wave kernel's REPLAY instantiation, the caller vouching for the inputs'
    # domain (actions clipped to the action space, order sizes < 5: PHX_RH_*_IN_DOMAIN), and with the device's pre-scan of the actions
    env = ph.SupplyChainEnv(n_shops=9, customers_per_shop=6, num_steps=100, batch_size=4096, seed=42, exogenous="device", device=device)
    env.reset(); dev = env._device()
    Tr = 400
    acts_r = (torch.rand(Tr, 4096, 9, device=dev.device) * 100.0).contiguous()
    exo_r = torch.randint(0, 5, (Tr, 4096, 54), dtype=torch.uint8, device=dev.device)
    trr = [dev.alloc_trajectory(Tr) for _ in range(2)]
    kr = [0]

    def replay(a_, x_, vouch=True):
        dev.rollout(Tr, a_, x_, out=trr[kr[0] & 1], actions_in_domain=vouch, exo_in_domain=x_ is not None); kr[0] += 1
    add("SC64 B=4096 (config 2), replayed actions", 64, 4096, Tr, timed(lambda: replay(acts_r, None), 30),
        f"fused rollout T={Tr}, phx_rollout_io.actions, PHX_RH_ACTIONS_IN_DOMAIN", bytes_per_env_step=26 * 9)   # the 22-byte record + the action read
    res[-1]["kernels"] = dev.last_kernel()
    add("SC64 B=4096 (config 2), replayed actions, pre-scanned on the device", 64, 4096, Tr, timed(lambda: replay(acts_r, None, False), 30),
        f"fused rollout T={Tr}, phx_rollout_io.actions, no hint", bytes_per_env_step=30 * 9)                      # + the scan's read
    res[-1]["kernels"] = dev.last_kernel()
    add("SC64 B=4096 (config 2), replayed actions + order sizes", 64, 4096, Tr, timed(lambda: replay(acts_r, exo_r), 30),
        f"fused rollout T={Tr}, phx_rollout_io.actions + exo, both hints", bytes_per_env_step=26 * 9 + 54)
    res[-1]["kernels"] = dev.last_kernel()
    del env, dev, trr, acts_r, exo_r; torch.cuda.empty_cache()
    # an FSM supply chain whose RESTOCK handler branches on the shops' total stock, declared in rule form (phx_spec.stage_rules, ABI 9):
    # the message-passing engine evaluates the rule inside the step -- no host callback, and rollouts run in one launch
    try:
        def restock(env_):                                       # the Python handler (fsm.py:294-307): the definition the rule is checked against
            env_.resolve_network()
            tot = sum(np.asarray(a_.stock) for aid, a_ in env_.agents.items() if str(aid).startswith("SHOP"))
            return np.where(tot < 300, "RESTOCK", "SELL").tolist()
        handler = ph.state_rules([ph.StageRule("shop.stock", "<", 300, "RESTOCK")])(restock)
        env = ph.SupplyChainFSMEnv(n_shops=9, customers_per_shop=6, num_steps=100, batch_size=4096, seed=42, exogenous="device", device=device,
                                   restock_handler=handler)
        env.reset(); dev = env._device()
        acts = torch.rand(4096, 9, device=dev.device) * 100.0
        add("SC64 FSM B=4096, state-dependent handler in rule form", 64, 4096, 1, timed(lambda: dev.step(acts), 60), "one launch per step (generic engine, rule evaluated on the device)")
        tr = dev.rollout(50)
        add("SC64 FSM B=4096, state-dependent handler in rule form", 64, 4096, 50, timed(lambda: dev.rollout(50, out=tr), 4), "rollout T=50, T-step loop in the kernel")
        del env, dev, tr, acts; torch.cuda.empty_cache()
    except Exception as exc:                                     # report, do not hide
        res.append({"config": "SC64 FSM rule-form handler", "error": str(exc)[:300]})
    # config 5: Stackelberg market 128 leaders / 1024 followers, B = 4096
    env = market_env(128, 1024, 8, 100, 4096, exogenous="device", device=device)
    env.reset(); dev = env._device()
    S = 1152
    valid = [torch.zeros(4096, S, dtype=torch.uint8, device=dev.device) for _ in range(2)]
    valid[0][:, :128] = 1; valid[1][:, 128:] = 1
    acts = torch.rand(4096, S, device=dev.device)
    k = [0]

    def one():
        dev.step(acts, valid[k[0] & 1]); k[0] += 1
    # per env-step, MINIMAL layout: outputs 21 B per agent (obs 8 + reward f64 8 + 5 flag bytes), actions + valid 5 B per
    # acting agent (half on average), seller state 56 B, bought / paid 12 B per buyer every other step.  The kernel also
    # moves the per-agent reward cache (f64 + valid byte, written by the rewarded half and read by the observing half: 9 B
    # per agent-step, derivable from bought / paid / revenue) -- that is traffic, not algorithmic bytes
    add("Stackelberg 128x1024 B=4096 (config 5)", S, 4096, 1, timed(one, 40), "one launch per step",
        bytes_per_env_step=21 * S + 5 * S // 2 + 56 * 128 + 6 * 1024)
    tr = dev.rollout(50)                                        # a 4.7 GB fragment: num_steps = 100, half an episode per launch
    add("Stackelberg 128x1024 B=4096 (config 5)", S, 4096, 50, timed(lambda: dev.rollout(50, out=tr), 4), "fused rollout T=50",
        bytes_per_env_step=20 * S)
    del tr; torch.cuda.empty_cache()
    tr = dev.alloc_trajectory(100)                              # one whole episode per launch (9.4 GB): the launch's ~180 us of block start-up and cold first steps over 100 steps
    dev.rollout(100, out=tr)
    add("Stackelberg 128x1024 B=4096 (config 5)", S, 4096, 100, timed(lambda: dev.rollout(100, out=tr), 3), "fused rollout T=100",
        bytes_per_env_step=20 * S)
    del env, dev, tr
    torch.cuda.empty_cache()
    # not a BASELINE config: the reference's digital-ads example at its own size (the exchange's
    # handle_batch auction), 1 exchange + 1 publisher + 120 advertisers with clipped-sampler budgets, B = 4096
    st = {}
    for i in range(120):
        lo = (5.0, 7.0, 10.0)[i // 40]
        st[f"ADV_{i + 1}"] = ph.AdvertiserAgent.Supertype(
            budget=ph.UniformFloatSampler(lo, lo + 10.001, clip_low=lo, clip_high=lo + 10.0))
    env = ph.DigitalAdsEnv(num_steps=20, num_agents_theme={"travel": 40, "tech": 40, "sport": 40},
                           agent_supertypes=st, batch_size=4096, seed=42, device=device)
    env.reset(); dev = env._device()
    acts = torch.rand(4096, 120, device=dev.device)
    # algorithmic bytes per env-step, N = 120 advertisers (the strategic agents), minimal layout: step outputs 25 B per
    # advertiser (obs f32[3] 12 + reward f64 8 + 5 flag / validity bytes) + action f32 4; advertiser state touched per step: `left` f64
    # read + written 16, `bid` f64 written 8, step clicks / wins / user i32 12, the served user's three total_* counters read +
    # written 24 = 60; publisher + env words 14  =>  N * 89 + 14.  Rollout: the trajectory record, 24 B per advertiser-step
    # (obs 12 + action 4 + reward 4 + terminated / truncated / obs_valid / reward_valid 4).
    add("digital-ads market 122 agents B=4096 (SURVEY 8f-4)", 122, 4096, 1, timed(lambda: dev.step(acts), 40), "one launch per step",
        bytes_per_env_step=120 * 89 + 14)
    tr = dev.rollout(40)
    add("digital-ads market 122 agents B=4096 (SURVEY 8f-4)", 122, 4096, 40, timed(lambda: dev.rollout(40, out=tr), 3),
        "fused rollout T=40", bytes_per_env_step=24 * 120)
    del env, dev, tr, acts
    torch.cuda.empty_cache()
    # the generic message-passing engine (LDS inboxes, static round schedule) on the two supply-chain shapes: what a topology
    # WITHOUT a fused schedule pays (force_generic); VERDICT r2 item 3 quotes these
    for name, S_, K_, B_, cls in (("SC64 B=4096", 9, 6, 4096, ph.SupplyChainEnv), ("SC256 FSM B=8192", 51, 4, 8192, ph.SupplyChainFSMEnv)):
        env = cls(n_shops=S_, customers_per_shop=K_, num_steps=100, batch_size=B_, seed=42, exogenous="device", device=device,
                  force_generic=True)
        env.reset(); dev = env._device()
        acts = torch.rand(B_, S_, device=dev.device) * 100
        A_ = 1 + S_ + S_ * K_
        fsm_ = cls is ph.SupplyChainFSMEnv
        sb = S_ * 43 + 6 + ((2 * S_ + 2) if fsm_ else 0)             # SURVEY 8d, device-RNG orders (+ stage and validity planes)
        add(f"{name}, generic engine (force_generic)", A_, B_, 1, timed(lambda: dev.step(acts), 60), "one launch per step",
            bytes_per_env_step=sb)
        res[-1]["kernel"] = dev.last_kernel()
        tr = dev.rollout(50)
        add(f"{name}, generic engine (force_generic)", A_, B_, 50, timed(lambda: dev.rollout(50, out=tr), 4), "rollout T=50, T-step loop in the kernel",
            bytes_per_env_step=(24 if fsm_ else 22) * S_)
        res[-1]["kernel"] = dev.last_kernel()
        del env, dev, tr, acts
        torch.cuda.empty_cache()
    # ... where more env instances no longer help (the dynamic kernel saturated at 347-353 env-steps per us)
    for name, S_, K_, B_, cls in (("SC64 B=65536", 9, 6, 65536, ph.SupplyChainEnv), ("SC256 FSM B=32768", 51, 4, 32768, ph.SupplyChainFSMEnv)):
        env = cls(n_shops=S_, customers_per_shop=K_, num_steps=100, batch_size=B_, seed=42, exogenous="device", device=device, force_generic=True)
        env.reset(); dev = env._device()
        acts = torch.rand(B_, S_, device=dev.device) * 100
        A_ = 1 + S_ + S_ * K_
        fsm_ = cls is ph.SupplyChainFSMEnv
        sec = timed(lambda: dev.step(acts), 40)
        add(f"{name}, generic engine at saturation", A_, B_, 1, sec, "one launch per step", bytes_per_env_step=S_ * 43 + 6 + ((2 * S_ + 2) if fsm_ else 0))
        res[-1]["kernel"] = dev.last_kernel(); res[-1]["env_steps_per_us"] = B_ / (sec * 1e6)
        del env, dev, acts
        torch.cuda.empty_cache()
    return res


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_spawn(n, argv, watchdog_s):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: re-exec this script under
    torch.distributed.run, one rank per GPU over RCCL, and pass its output (rank 0's JSON line) through.
    If the children die or hang without a JSON line, the parent prints the error line itself."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    err = None
    try:
        out, _ = proc.communicate(timeout=watchdog_s + 120)
    except subprocess.TimeoutExpired:
        import signal
        try:
            os.killpg(proc.pid, signal.SIGKILL)          # exactly the process group started above
        except OSError:
            pass
        out, _ = proc.communicate()
        err = f"launcher watchdog: no exit after {watchdog_s + 120:.0f} s"
    sys.stdout.write(out or "")
    has_line = any(l.startswith("{") and '"metric"' in l for l in (out or "").splitlines())
    if not has_line:
        print(json.dumps(error_line(n, err or f"ranks exited with code {proc.returncode} without a result line",
                                    stage="self_spawn")), flush=True)
    raise SystemExit(proc.returncode if proc.returncode is not None else 1)


def error_line(n_gpus, error, stage=None, partial=None):
    """the JSON line of a run that failed: same keys, value null (or what was measured before the failure)."""
    out = {"metric": "agent_steps_per_sec", "value": None, "unit": "agent-steps/s", "n_gpus": n_gpus,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
           "rccl_ranks_seen": 0}
    out.update(partial or {})
    out["error"] = str(error)[:2000]
    if stage is not None:
        out["failed_stage"] = stage
    return out


class Watch:
    """Rank 0's guarantee of ONE JSON line: `line` accumulates what has been measured; a failure (exception, a stage
    over its deadline, SIGTERM from the launcher because another rank died) prints it with an `error` field."""

    def __init__(self, n_gpus, rank, total_s):
        import threading
        self.n_gpus, self.rank, self.line = n_gpus, rank, {}
        self.stage_name, self.deadline, self.total_deadline = "start", None, time.monotonic() + total_s
        self.lock, self.printed = threading.Lock(), False
        # SIGTERM (torchrun tears the job down when another rank dies) is taken by a dedicated sigwait thread: a Python
        # signal handler would only run once the main thread returns from whatever C call it is blocked in (a gloo
        # rendezvous, a device synchronize) -- i.e. possibly never before the launcher's SIGKILL.  The mask is set
        # before any other thread exists, so every later thread (torch, gloo, RCCL) inherits it.
        try:
            signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})
            threading.Thread(target=self._wait_term, daemon=True).start()
        except (AttributeError, ValueError, OSError):
            pass
        threading.Thread(target=self._run, daemon=True).start()

    def _wait_term(self):
        signal.sigwait({signal.SIGTERM})
        self.emit("SIGTERM from the launcher (another rank failed or the job was torn down)")
        os._exit(143)

    def stage(self, name, seconds):
        self.stage_name, self.deadline = name, time.monotonic() + seconds

    def emit(self, error=None):
        with self.lock:
            if self.printed:
                return
            self.printed = True
        if self.rank == 0:
            line = self.line if error is None else error_line(self.n_gpus, error, self.stage_name, self.line)
            print(json.dumps(line), flush=True)
        elif error is not None:
            print(f"[bench rank {self.rank}] {error} (stage {self.stage_name})", file=sys.stderr, flush=True)

    def _run(self):
        while not self.printed:
            time.sleep(0.5)
            now = time.monotonic()
            if (self.deadline is not None and now > self.deadline) or now > self.total_deadline:
                self.emit(f"watchdog: stage '{self.stage_name}' exceeded its deadline (hang in a collective, RCCL init or a kernel)")
                os._exit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="sc64")
    ap.add_argument("--batch", type=int, default=None, help="envs per GPU (default: the config's)")
    ap.add_argument("--min-region-ms", type=float, default=6000.0,
                    help="the K-step region is repeated back to back until the timed region lasts at least this long")
    ap.add_argument("--buffers", type=int, default=0,
                    help="trajectory buffers the launches rotate over (default: enough for > 320 MB, at least 4)")
    ap.add_argument("--episodes-per-launch", type=int, default=1,
                    help="episodes (of num_steps = 100 steps) per trajectory FRAGMENT (one buffer)")
    ap.add_argument("--fragments-per-call", type=int, default=8,
                    help="fragments per phx_rollout call (phx_rollout_io.frags, ABI 9): the consumer gets one-episode fragments in separate "
                         "buffers, the pipeline fill, the placement of a 160 KB workgroup on every CU and the kernel boundary are paid once "
                         "per call (round 4: ONE four-episode fragment per launch = --episodes-per-launch 4 --fragments-per-call 1)")
    ap.add_argument("--no-autotune", action="store_true", help="keep the library's default block shape (no env.autotune_rollout)")
    ap.add_argument("--watchdog-s", type=float, default=900.0, help="overall deadline; a JSON line with `error` is printed when it passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-per-step", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-frag200", action="store_true",
                    help="skip the informational one-episode-per-launch loop (profiling: keeps per-kernel averages to one launch shape)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        try:
            signal.pthread_sigmask(signal.SIG_UNBLOCK, {signal.SIGTERM})     # the launcher keeps the default disposition
        except (AttributeError, ValueError, OSError):
            pass
        self_spawn(args.gpus, sys.argv[1:], args.watchdog_s)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    watch = Watch(args.gpus, rank, args.watchdog_s)
    try:
        run(args, rank, local_rank, world, watch)
    except SystemExit:
        raise
    except BaseException as exc:                     # noqa: BLE001 -- the line must appear whatever happened
        import traceback
        traceback.print_exc()
        watch.emit(f"{type(exc).__name__}: {exc}")
        os._exit(1)
    watch.emit()
    sys.stdout.flush()
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("PHX_BENCH_FORCE_DIST"):
        os._exit(0)  # process-group / communicator destructors have nothing left to do and can hang when a peer is gone
    # (a single process returns normally: a profiler attached to it -- rocprofv3 -- writes its output at interpreter exit)


def run(args, rank, local_rank, world, watch):
    if world != args.gpus:
        raise RuntimeError(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the PhantomEnv.step() path has no CPU fallback")
    if os.environ.get("PHX_BENCH_SHARE_GPU"):
        # test hook: several ranks on the GPUs that exist (RCCL then refuses the duplicate device -- the failure path
        # of the data plane -- while `value` over the gloo control plane stays valid)
        local_rank %= max(1, torch.cuda.device_count())
    if local_rank >= torch.cuda.device_count():
        raise RuntimeError(f"LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("PHX_BENCH_FORCE_DIST"):
        # one process per GPU.  CONTROL plane (barrier, max over ranks) = gloo on CPU tensors: the step path has no
        # collective, so `value` must not depend on RCCL coming up.  DATA plane (rollout collection) = RCCL (backend
        # "nccl" on ROCm), created below only for the sections that exchange trajectories.  Also taken with a single
        # rank (torchrun with one process, or PHX_BENCH_FORCE_DIST=1) so the path is exercised on a 1-GPU box.
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        watch.stage("init_process_group(gloo)", 180)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=170))

    import phantom_amd as ph
    global N_SHOPS, CUST_PER_SHOP, N_AGENTS
    N_SHOPS, CUST_PER_SHOP, default_batch = CONFIGS[args.config]
    N_AGENTS = 1 + N_SHOPS + N_SHOPS * CUST_PER_SHOP
    B, S, K, W = args.batch or default_batch, N_SHOPS, args.steps, args.warmup
    if K < 1:
        raise RuntimeError("--steps must be >= 1")
    watch.line.update({"metric": "agent_steps_per_sec", "value": None, "unit": "agent-steps/s", "n_gpus": world,
                       "steps": K, "warmup": W, "rccl_ranks_seen": 0})
    cpu_first = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the CPU leg FIRST: everything after it is GPU work, so the chip is busy from here to the end of the
        # run and a utilisation sampler that looks at the process every few seconds sees it
        watch.stage("cpu baseline", 300)
        cpu_first = cpu_baseline()
    watch.stage("create env", 180)
    # one process per GPU owns envs [rank*B, (rank+1)*B); the RNG is keyed by the GLOBAL env
    # index so results do not depend on the number of GPUs.  No collective inside a step.
    env = ph.SupplyChainEnv(n_shops=N_SHOPS, customers_per_shop=CUST_PER_SHOP, num_steps=NUM_STEPS,
                            batch_size=B, seed=42, env_offset=rank * B, exogenous="device",
                            device=f"cuda:{local_rank}")
    NF = max(1, min(args.fragments_per_call, 8))            # fragments per call
    TF = NUM_STEPS * max(1, args.episodes_per_launch)       # steps per fragment
    T = TF * NF                                             # steps per launch
    tune = None
    if not args.no_autotune:
        # block shape of the rollout kernel picked on THIS box (every variant gives the same trajectories): outside
        # the timed region, a few launches per candidate (PhantomEnv.autotune_rollout)
        watch.stage("autotune", 180)
        tune = env.autotune_rollout(min(T, 4 * NUM_STEPS))
    dev = env._device()
    assert dev.uses_fused, "bench expects the fused supply-chain kernels"
    env.reset()
    # The launches rotate over several trajectory buffers whose total exceeds the 256 MB Infinity Cache, so that
    # every fragment's bytes are really written to HBM (one 82 MB buffer rewritten in place could live in the cache)
    frag_bytes = algorithmic_bytes_rollout(B, S, T)          # per CALL (all its fragments)
    n_buf = args.buffers if args.buffers > 0 else max(2, -(-(320 << 20) // frag_bytes))      # sets of NF fragment buffers
    sets = [[dev.alloc_trajectory(TF) for _ in range(NF)] for _ in range(n_buf)]             # allocates the trajectory buffers once
    for st_ in sets:
        dev.rollout_fragments(TF, st_)
    trajs = sets
    traj = sets[0]
    rot = [0]

    # The store-wave kernel (round 4) writes every flag word from its store waves; round 3's kernel, where it serves the launch,
    # stores only the non-zero words after phx_rollout's own in-line fill (phx_spec.variant_flags sparse).
    served_by = dev.last_kernel()                              # what phx_rollout launched for the bench fragment (phx_last_kernel)
    store_waves = "phx_sc_rollout_sw_kernel" in served_by
    sparse = (not store_waves) and T * B * S >= (1 << 23)

    def launches(n, bufs=None):                             # n calls of NF fragments each, back to back
        bufs = bufs or trajs
        k = rot[0]
        for _ in range(n):
            dev.rollout_fragments(TF, bufs[k % len(bufs)]); k += 1
        rot[0] = k

    def sync_barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64)          # CPU tensor: the gloo control group
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # The env steps as a continuous stream (auto-reset at every episode end), issued as fragments of T = 100
    # steps per launch.  A K-step region shorter than ~1 s would time launch + sync latency, not the path
    # (K = 20 is ONE 8 us slice) and the driver's GPU-busy sampler would see an idle chip: the region is therefore
    # repeated R times back to back, R the smallest
    # count that makes R*K a whole number of fragments and the timed region >= --min-region-ms; the line
    # reports steps = K, repeats = R and ms_per_step = elapsed / (R*K).  Warm-up: W steps rounded up to whole
    # fragments (at least 20 fragments, which also calibrates R -- identically on every rank).
    import math
    watch.stage("warm-up + calibration", 240)
    n_warm = max(20, -(-W // T))
    launches(n_warm)
    sync_barrier()
    t0 = time.perf_counter(); launches(200); torch.cuda.synchronize()
    per_launch = max_over_ranks((time.perf_counter() - t0) / 200)
    unit = T // math.gcd(K, T)                              # repeats per whole number of fragments
    R = max(1, math.ceil(1.05 * args.min_region_ms * 1e-3 / per_launch * T / K))
    R = -(-R // unit) * unit
    n_launch = R * K // T
    watch.stage("timed region", 120 + 20 * args.min_region_ms * 1e-3)
    sync_barrier()
    t0 = time.perf_counter()
    launches(n_launch)
    torch.cuda.synchronize()                 # this rank's R*K steps are done ...
    elapsed = time.perf_counter() - t0
    sync_barrier()                           # ... all ranks are; the job's time is the MAX over ranks
    elapsed = max_over_ranks(elapsed)
    value = N_AGENTS * B * world * (R * K) / elapsed
    out = watch.line
    out.update({
        "metric": "agent_steps_per_sec", "value": value, "unit": "agent-steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / (R * K) * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32", "data": "synthetic",
        "config": {"workload": f"supply-chain {args.config.upper()} (1 factory + {N_SHOPS} shops + "
                               f"{N_SHOPS * CUST_PER_SHOP} customers = {N_AGENTS} agents), "
                               f"batch {B} envs per GPU, random actions U[0,100), device-RNG orders, "
                               f"fused on-device rollouts: {NF} fragment(s) of {TF} steps ({TF // NUM_STEPS} episode(s) of {NUM_STEPS} steps) each, "
                               f"in separate buffers, per phx_rollout call (T={T} steps per launch) with the full trajectory written to HBM",
                   "agents": N_AGENTS, "envs_per_gpu": B, "global_envs": B * world,
                   "num_steps": NUM_STEPS, "mode": "phx_rollout", "sharding": f"env-batch x{world}, no step-time collective",
                   "trajectory_buffers": f"{n_buf} sets x {NF} fragments x {frag_bytes / NF / 1e6:.1f} MB, rotated (more than the 256 MB Infinity Cache)",
                   "steps_per_launch": T, "episodes_per_launch": T // NUM_STEPS, "fragments_per_call": NF, "steps_per_fragment": TF,
                   "autotune": tune,
                   "flag_planes": (("zero-filled in line by phx_rollout, non-zero words by the kernel" if sparse else
                                    ("every word stored by the kernel's store waves, whole 16-byte pieces (no fill launch)" if store_waves
                                     else "every word stored by the kernel"))),
                   "served_by": served_by},
        "repeats": R, "timed_steps": R * K, "timed_launches": n_launch, "timed_region_ms": elapsed * 1e3,
        "warmup_launches": n_warm,
        "timing_note": f"the {K}-step region is run {R}x back to back as {n_launch} calls of {NF} x {TF} steps; "
                       "ms_per_step = timed_region_ms / (repeats * steps)",
        "env_steps_per_sec": B * world * (R * K) / elapsed,
        "control_plane": "gloo" if dist is not None else None,
    })

    # ---- kernel-level timing for the roofline (HIP events on the launch stream) -------------
    # ONE event pair around >= 200 full-length launches (independent of K), divided by their number: the
    # average launch duration as the stream sees it, which is what the rocprofv3 kernel trace in profiles/
    # averages too (an event pair per launch would add ~2 us of marker packets to each interval).
    watch.stage("roofline loop", 240)

    def event_ms(n, bufs=None):
        launches(10, bufs)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        launches(n, bufs)
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / n

    n_full = max(60, 400 * NUM_STEPS // T)
    launch_ms = event_ms(n_full)                              # rotating over the buffers: HBM
    same_ms = event_ms(n_full, [traj])                        # one buffer rewritten in place (round 2's loop): may sit in the Infinity Cache
    inline_ms = None
    alg = frag_bytes
    achieved = alg / (launch_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and args.config == "sc64" and B == BATCH:
        try:
            rec = json.load(open(pmc)).get("phx_sc_rollout_sw_kernel" if store_waves else "phx_sc_rollout_fast_kernel", {})
            if int(rec.get("steps_per_launch", NUM_STEPS)) == T:             # the counters were taken on this launch shape
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command: " + \
                    str(rec.get("source", "see file")) + "); not re-measured by this run"
        except Exception:
            traffic = None
    # the same kernel with ONE episode per launch (T = 100, what rounds 1-2 reported): the launch gap, the block ramp and the
    # two pipeline iterations before the first store are paid per 82 MB instead of per launch of T steps; informational
    def single_launch_shape(Tl, n):
        """ONE fragment of Tl steps per launch (rounds 1-4's launch shapes), buffers rotated: what a caller pays who asks for one
        fragment per call"""
        algl = algorithmic_bytes_rollout(B, S, Tl)
        tb = [dev.alloc_trajectory(Tl) for _ in range(max(2, -(-(320 << 20) // algl)))]
        for k in range(10):
            dev.rollout(Tl, out=tb[k % len(tb)])
        kern = dev.last_kernel()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        for k in range(n):
            dev.rollout(Tl, out=tb[k % len(tb)])
        g1.record(); torch.cuda.synchronize()
        ms = g0.elapsed_time(g1) / n
        return {"T": Tl, "launch_ms": ms, "achieved": algl / (ms * 1e-3) / 1e9, "frac": algl / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "buffers": len(tb), "kernel": kern, "agent_steps_per_sec": N_AGENTS * B * Tl / (ms * 1e-3)}
    frag1 = frag4 = None
    try:
        if args.no_frag200:
            raise RuntimeError("skipped")
        frag1 = single_launch_shape(NUM_STEPS, 400)
        frag4 = single_launch_shape(4 * NUM_STEPS, 120)
    except Exception as e:                       # report, do not hide
        frag1 = frag1 or {"error": str(e)}
    # the same launches WITHOUT the `terminations` plane (all zero: ShopAgent never terminates; phx_rollout_io.terminated = NULL):
    # 21 instead of 22 bytes per shop-step.  Informational: `value` and `roofline` above write the full 22-byte record.
    no_term = None
    try:
        tn = [[dev.alloc_trajectory(TF, terminations=False) for _ in range(NF)] for _ in range(n_buf)]
        for k in range(4):
            dev.rollout_fragments(TF, tn[k % n_buf])
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        for k in range(n_full):
            dev.rollout_fragments(TF, tn[k % n_buf])
        g1.record(); torch.cuda.synchronize()
        msn = g0.elapsed_time(g1) / n_full
        algn = alg - B * T * S
        no_term = {"launch_ms": msn, "algorithmic_bytes_per_launch": algn, "achieved": algn / (msn * 1e-3) / 1e9,
                   "frac": algn / (msn * 1e-3) / 1e9 / HBM_PEAK_GBS, "agent_steps_per_sec": N_AGENTS * B * T / (msn * 1e-3)}
        del tn
    except Exception as e:                       # report, do not hide
        no_term = {"error": str(e)[:300]}
    # achievable write bandwidth of this box for buffers of the trajectory's size (a plain fill, same rotation)
    fills = [torch.empty(alg // 4, dtype=torch.float32, device=dev.device) for _ in range(n_buf)]
    for f in fills:
        f.fill_(1.0)
    torch.cuda.synchronize()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record()
    for k in range(40):
        fills[k % n_buf].fill_(1.0)
    f1.record(); torch.cuda.synchronize()
    fill_gbs = alg / (f0.elapsed_time(f1) / 40 * 1e-3) / 1e9
    del fills
    # which kind of box this is (VERDICT r3 Weak #8): the same binary takes 65-69 us per T = 400 fragment on one MI355X and 5-8 % more on
    # another; the plain fill of the same bytes (below) and the tuning pass's per-candidate times tell them apart
    # (round 4 derived the class from the fill rate, which is the same on both kinds; what separates them is how close the rollout's
    #  store pattern gets to that fill.  Until late round 5 the single-launch T = 400 shape's share decided; with the fused worker phase and
    #  one store piece per trip the two shapes no longer rank the boxes alike -- a box with 0.767 of the peak on the bench shape and 0.715
    #  at T = 400 was labelled slow -- so the class is the BENCH shape's own share now, >= 0.88 of the fill (0.75 of the peak and up): fast;
    #  both shares stay on the line)
    fof_t400 = (frag4["achieved"] / fill_gbs) if (frag4 and "achieved" in frag4) else None
    fof = achieved / fill_gbs
    out["roofline"] = {"bound": "hbm", "kernel": served_by, "achieved": achieved,
                       "trace_equivalent": {"what": "kernel(s) of one phx_rollout call as a rocprofv3 kernel trace would sum them "
                                                    "(the event pair brackets back-to-back calls: launch gaps included)",
                                            "kernels": served_by, "us_per_call": launch_ms * 1e3},
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": alg,
                       "launch_ms": launch_ms, "launches_timed": n_full, "launch": f"T={T} steps x B={B} envs",
                       "buffers_rotated": n_buf,
                       "same_buffer": {"launch_ms": same_ms, "frac": alg / (same_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "note": "every launch rewrites ONE buffer in place (a T = 100 fragment fits the 256 MB Infinity Cache): not the HBM figure"},
                       "measured_fill_GBps_same_bytes": fill_gbs, "frac_of_measured_fill": achieved / fill_gbs,
                       "ms_per_100_steps": launch_ms * NUM_STEPS / T,
                       "one_episode_per_launch": frag1, "four_episodes_one_fragment_per_launch": frag4,
                       "shares_of_measured_fill": {"bench_shape": fof, "T400_single_launch": fof_t400,
                                                   "note": "no class label any more (VERDICT r5 weak #9: every box seen was 'slow' under the rule): the ASIC serial "
                                                           "in `box` identifies the chip, these two shares say how close its store pattern gets to its own fill rate"},
                       "without_terminations_plane": no_term}
    try:
        out["box"] = box_info()
    except Exception as e:                       # informational
        out["box"] = {"error": str(e)[:200]}
    if inline_ms is not None:
        out["roofline"]["flag_fill_in_line"] = {"launch_ms": inline_ms, "frac": alg / (inline_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "note": "phx_rollout zeroes the flag planes itself before the kernel (no side stream, no hint): fill + kernel in series"}

    # ---- per-launch PhantomEnv.step mode (one kernel launch per step) ---------------------------
    if not args.no_per_step:
        watch.stage("per-step mode", 300)
        try:
            out["per_step"] = bench_per_step(env, dev, B, S, world, sync_barrier)
        except Exception as e:                   # report, do not hide
            out["per_step"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- rollout collection exchange (BASELINE config 4's RCCL all-gather), outside `value` ------
    if dist is not None:
        watch.stage("RCCL init (data plane)", 240)
        nccl = None
        try:
            import datetime
            nccl = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=600))
            one = torch.ones(1, device=dev.device)
            dist.all_reduce(one, group=nccl)
            torch.cuda.synchronize()
            out["rccl_ranks_seen"] = int(one.item())
            out["rccl_env"] = {k: os.environ[k] for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_P2P_DISABLE", "HSA_ENABLE_IPC_MODE_LEGACY")
                               if k in os.environ}
        except Exception as e:
            out["rccl_error"] = f"{type(e).__name__}: {e}"[:1500]
            out["error"] = "RCCL did not come up: `value` (control plane over gloo) is valid, the rollout-collection sections were skipped"
            nccl = None
        if nccl is not None:
            watch.stage("rollout_allgather", 300)
            try:
                out["rollout_allgather"] = bench_collection(dist, nccl, dev, traj, NUM_STEPS, B, world, rank, sync_barrier,
                                                            max_over_ranks, N_AGENTS)
            except Exception as e:
                out["rollout_allgather"] = {"error": f"{type(e).__name__}: {e}"[:1500]}
                out["error"] = "rollout_allgather failed (see that section); `value` is valid"
            if args.config == "sc64":
                # BASELINE config 4, one GPU's share: SC256 (1 + 51 + 204), B = 8192 per GPU, plain env, T = 100 fused
                # rollouts; throughput with the trajectory all-gather excluded and included
                watch.stage("config4_share", 420)
                del trajs[:]
                traj = None
                torch.cuda.empty_cache()
                try:
                    out["config4_share"] = bench_config4(ph, dist, nccl, world, rank, local_rank, sync_barrier, max_over_ranks)
                except Exception as e:
                    out["config4_share"] = {"error": f"{type(e).__name__}: {e}"[:1500]}
                    out["error"] = "config4_share failed (see that section); `value` is valid"

    # ---- BASELINE.json configs 3-5 on this GPU (parity-test cases; reported for orientation only) ------
    if rank == 0 and world == 1 and args.config == "sc64" and not args.no_other_configs:
        watch.stage("other configs", 420)
        del trajs[:]
        traj = None
        torch.cuda.empty_cache()
        try:
            out["other_configs"] = other_configs(f"cuda:{local_rank}")
        except Exception as e:
            out["other_configs"] = {"error": f"{type(e).__name__}: {e}"[:1500]}

    if cpu_first is not None:
        out["cpu_baseline"] = cpu_first
    watch.stage("done", 60)


def bench_per_step(env, dev, B, S, world, sync_barrier):
    ksteps = 1000
    acts = torch.rand(ksteps, B, S, device=dev.device) * 100.0
    env.reset()
    for i in range(20):
        dev.step(acts[i % ksteps])
    sync_barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for i in range(ksteps):
        dev.step(acts[i])
    e1.record()
    sync_barrier()
    dt = time.perf_counter() - t0
    # kernel-only duration: the tight loop above is host bound, so the kernel time is taken from a
    # second pass with per-launch events
    evs = []
    for i in range(200):
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); dev.step(acts[i % ksteps]); b_.record(); evs.append((a, b_))
    torch.cuda.synchronize()
    kms = float(np.median([a.elapsed_time(b_) for a, b_ in evs]))
    alg_s = algorithmic_bytes_step(B, S, CUST_PER_SHOP, device_rng=True)
    # the same per-step launches captured once into a hipGraph (DeviceEnv.step_graph: 100 steps per replay)
    graph_us = None
    try:
        sg = dev.step_graph(acts[:100])
        for _ in range(3):
            sg.replay()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(20):
            sg.replay()
        torch.cuda.synchronize()
        graph_us = (time.perf_counter() - tg) / 2000 * 1e6
    except Exception as e:                                   # report, do not hide
        graph_us = f"capture failed: {e}"
    res = {"value": N_AGENTS * B * world * ksteps / dt, "unit": "agent-steps/s",
           "steps": ksteps, "ms_per_step_wall": dt / ksteps * 1e3,
           "event_ms_per_launch": kms,
           "hipgraph_us_per_step": graph_us,
           "roofline": {"bound": "hbm", "kernel": "phx_sc_step_kernel",
                        "achieved": alg_s / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": alg_s / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": alg_s,
                        "note": "event interval includes launch gaps; B=4096 is latency-bound"}}
    try:
        res["rllib_adapter"] = bench_rllib_adapter(env, B, S)
    except Exception as e:                                   # report, do not hide
        res["rllib_adapter"] = {"error": f"{type(e).__name__}: {e}"[:500]}
    try:
        res["on_policy"] = bench_on_policy(env, dev, B, S)
        res["on_policy_fused"] = bench_on_policy_fused(env, dev, B, S)
    except Exception as e:                                   # report, do not hide
        res["on_policy"] = {"error": f"{type(e).__name__}: {e}"[:500]}
    if world == 1:
        try:
            res["batch_sweep"] = bench_step_sweep(dev.device)
        except Exception as e:                               # report, do not hide
            res["batch_sweep"] = {"error": f"{type(e).__name__}: {e}"[:500]}
    return res


def bench_on_policy(env, dev, B, S):
    """A LEARNED policy in the loop (utils/rllib/rollout.py:300-363 calls the policy for every agent at every step): a small torch MLP
    maps the step's observations to the next actions between the per-step launches; reset, the T = 100 act / step iterations and the
    copies of every step's observation, action, reward and done flags into a time-major trajectory are captured ONCE in a hipGraph and
    replayed.  What one launch per step plus inference costs next to the random-policy fused rollout."""
    T = NUM_STEPS
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(3, 32), torch.nn.Tanh(), torch.nn.Linear(32, 1)).to(dev.device)
    for p_ in mlp.parameters():
        p_.requires_grad_(False)
    obs = torch.empty(T, B, S, 3, device=dev.device); act = torch.empty(T, B, S, device=dev.device)
    rew = torch.empty(T, B, S, device=dev.device); tru = torch.empty(T, B, S, dtype=torch.uint8, device=dev.device)

    def episode():
        o, _ = dev.reset()
        for i in range(T):
            torch.mul(torch.sigmoid(mlp(o).squeeze(-1)), 100.0, out=act[i])
            st = dev.step(act[i])
            obs[i].copy_(st.observations); rew[i].copy_(st.rewards)
            torch.maximum(st.truncations, st.all_truncated[:, None], out=tru[i])      # per-agent flag OR truncations["__all__"], as phx_rollout records it
            o = st.observations

    dev._ensure_step_io()
    episode(); torch.cuda.synchronize()                       # eager once (also warms the GEMM heuristics), then captured
    eager0 = time.perf_counter(); episode(); torch.cuda.synchronize(); eager = time.perf_counter() - eager0
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(dev.device)
    with torch.cuda.graph(g, stream=side):
        episode()
    torch.cuda.synchronize()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    assert int(tru[T - 1].sum()) == B * S and int(tru[:T - 1].sum()) == 0      # one whole episode per replay
    # policy-only and env-only shares: the same graph without the step launches / without the MLP
    def timed_graph(body):
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg, stream=side):
            body()
        torch.cuda.synchronize(); gg.replay(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            gg.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / n
    o0 = dev.obs

    def pol_only():
        for i in range(T):
            torch.mul(torch.sigmoid(mlp(o0).squeeze(-1)), 100.0, out=act[i])

    def env_only():
        dev.reset()
        for i in range(T):
            dev.step(act[i])
    tp, te = timed_graph(pol_only), timed_graph(env_only)
    return {"policy": "MLP 3-32-1 (tanh, sigmoid x 100) on every (env, shop) observation, torch fp32", "steps_per_replay": T,
            "us_per_step": dt / T * 1e6, "agent_steps_per_sec": N_AGENTS * B * T / dt,
            "us_per_step_policy_alone": tp / T * 1e6, "us_per_step_env_launches_alone": te / T * 1e6,
            "us_per_step_eager": eager / T * 1e6, "trajectory": "obs / action / reward / truncated copied per step into [T, B, S, ..] buffers inside the graph",
            "note": "one phx_step launch per step with the policy's kernels between the launches (hipGraph of one episode incl. reset)"}


def bench_on_policy_fused(env, dev, B, S):
    """The same collection loop with the policy evaluated ON THE DEVICE inside the fused rollout (phx_rollout_io.policy, ABI 10:
    phx_sc_rollout_policy_kernel, one lane per (env, shop)): a 3-32-1 MLP (ReLU, action = clip(60 y + 45, 0, 100)) on the agent's previous
    observation, T = 100 on-policy steps per launch, the full 22-byte trajectory record."""
    import numpy as np
    import phantom_amd as ph
    T = NUM_STEPS
    rng = np.random.default_rng(0)
    res = {}
    for name, widths in (("mlp_3_32_1", (32,)), ("mlp_3_64_64_1", (64, 64))):
        dims = [3] + list(widths) + [1]
        pol = ph.MLPPolicy([rng.normal(0, 1 / np.sqrt(dims[l]), (dims[l + 1], dims[l])).astype(np.float32) for l in range(len(dims) - 1)],
                           [rng.normal(0, 0.3, (dims[l + 1],)).astype(np.float32) for l in range(len(dims) - 1)], out_scale=60.0, out_bias=45.0)
        env.reset()
        trs = [dev.alloc_trajectory(T) for _ in range(2)]
        for k in range(3):
            dev.rollout(T, out=trs[k & 1], policy=pol)
        kern = dev.last_kernel()
        n = 20 if len(widths) == 1 else 4
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(n):
            dev.rollout(T, out=trs[k & 1], policy=pol)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        acts = trs[(n - 1) & 1].actions
        res[name] = {"us_per_step": us / T, "agent_steps_per_sec": N_AGENTS * B * T / (us * 1e-6), "steps_per_launch": T, "kernel": kern,
                     "distinct_actions_in_last_fragment": int(torch.unique(acts).numel()),
                     "hbm_frac_of_peak": algorithmic_bytes_rollout(B, S, T) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        del trs
    res["note"] = ("policy evaluated inside the rollout launch (lane per (env, shop), weights staged in LDS, arithmetic defined in include/phantom_amd.h "
                   "and restated bit for bit by the oracle: tests/test_gpu_policy.py); compare `on_policy` (torch MLP between per-step launches)")
    return res


def bench_step_sweep(device):
    """phx_sc_step_kernel (one PhantomEnv.step per launch) at B = 4096 .. 262 144 env instances, replayed from a hipGraph of 50
    launches so that the host's launch cadence is out of the picture: where the per-launch mode leaves the latency-bound
    regime and what fraction of the 8 TB/s it settles at (VERDICT r3 Weak #7)."""
    import phantom_amd as ph
    rows = []
    for Bs in (4096, 16384, 65536, 262144):
        env = ph.SupplyChainEnv(n_shops=N_SHOPS, customers_per_shop=CUST_PER_SHOP, num_steps=NUM_STEPS, batch_size=Bs, seed=42,
                                exogenous="device", device=str(device))
        env.reset(); d = env._device()
        a = (torch.rand(50, Bs, N_SHOPS, device=d.device) * 100.0).contiguous()
        sg = d.step_graph(a)
        for _ in range(3):
            sg.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            sg.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 500 * 1e3
        alg = algorithmic_bytes_step(Bs, N_SHOPS, CUST_PER_SHOP, device_rng=True)
        rows.append({"envs": Bs, "us_per_step": us, "agent_steps_per_sec": N_AGENTS * Bs / (us * 1e-6),
                     "hbm": {"algorithmic_bytes_per_launch": alg, "achieved_GBps": alg / (us * 1e-6) / 1e9,
                             "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}})
        del sg, a, d, env
        torch.cuda.empty_cache()
    return {"mode": "phx_step per launch, hipGraph of 50 launches, SC64", "rows": rows}


def bench_rllib_adapter(env, B, S):
    """the RLlib-facing vector-env path (phantom_amd/rllib.py): poll() + send_actions() per step at full batch,
    with RLlib's MultiEnvDict signature and with the tensor fast path."""
    from phantom_amd.rllib import BatchedBaseEnv
    if not hasattr(BatchedBaseEnv, "send_action_tensor"):
        return {"error": "adapter without the tensor fast path"}
    be_keep, be_lazy = BatchedBaseEnv(env), BatchedBaseEnv(env, keep_results=False)
    res = {"envs": B, "note": "tensor: the default adapter (every poll() result stays readable: one asynchronous copy per step); tensor_zero_copy: "
                              "keep_results=False (a result is copied when first read, before the next step)"}
    acts = torch.rand(B, S, device=env._device().device) * 100.0
    ids = sorted(be_keep.get_agent_ids())
    def read_rows(o):                          # every env's observation / reward row materialised, as a sampler would
        for b in range(B):
            row, rw = o[0][b], o[1][b]
            for aid in ids:
                row[aid], rw.get(aid)
    for mode, n in (("tensor", 50), ("tensor_zero_copy", 50), ("multi_env_dict", 5), ("multi_env_dict_rows_read", 5)):
        be = be_lazy if mode == "tensor_zero_copy" else be_keep
        env.reset()
        be._pending = None
        obs = be.poll()
        if mode == "multi_env_dict_rows_read":   # one untimed step: the adapter's first host block (36 864 observation views + 4 096 row dicts,
            be.send_action_tensor(acts); obs = be.poll(); read_rows(obs)      # ~6 ms once) is made here, not inside a five-step average
        t0 = time.perf_counter()
        for _ in range(n):
            if mode.startswith("tensor"):        # [B, S] tensor in, lazy MultiEnvDicts out (nothing read)
                be.send_action_tensor(acts)
            else:                                # RLlib's MultiEnvDict in: B x S python entries converted once
                be.send_actions({b: {aid: 50.0 for aid in ids} for b in range(B)})
            obs = be.poll()
            if mode == "multi_env_dict_rows_read":
                read_rows(obs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[mode] = {"ms_per_step": dt * 1e3, "agent_steps_per_sec": N_AGENTS * B / dt, "steps": n}
    # the floor of the rows_read loop itself: the same python (build the B x S action dict, read every obs / reward entry)
    # against plain precomputed dicts -- no env, no adapter
    plain_o = [{aid: None for aid in ids} for _ in range(B)]; plain_r = [{aid: 0.0 for aid in ids} for _ in range(B)]
    t0 = time.perf_counter()
    for _ in range(3):
        _d = {b: {aid: 50.0 for aid in ids} for b in range(B)}
        for b in range(B):
            row, rw = plain_o[b], plain_r[b]
            for aid in ids:
                row[aid], rw.get(aid)
    dt = (time.perf_counter() - t0) / 3
    res["python_floor_of_rows_read_loop"] = {"ms_per_step": dt * 1e3, "agent_steps_per_sec": N_AGENTS * B / dt,
                                             "note": "the measuring loop alone on plain dicts: what no adapter can go below"}
    # the bulk exit: T = 100 steps per call as fused device rollouts, the fragment as per-policy SampleBatch column dicts
    env.reset()
    be = be_keep
    be.sample(NUM_STEPS)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        sb = be.sample(NUM_STEPS)
    dt = (time.perf_counter() - t0) / n
    rows = sum(len(v["rewards"]) for v in sb.values())
    res["bulk_sample_batches"] = {"ms_per_call": dt * 1e3, "steps_per_call": NUM_STEPS, "agent_steps_per_sec": N_AGENTS * B * NUM_STEPS / dt,
                                  "strategic_rows_per_call": rows, "columns": sorted(next(iter(sb.values()))),
                                  "note": "BatchedBaseEnv.sample(100): one episode per launch, transposed on the device, ONE pinned copy per column, "
                                          "numpy views per policy (phantom_amd/rollout.py)"}
    res["note"] = ("poll() + send per step at full batch through phantom_amd.rllib.BatchedBaseEnv: tensor = send_action_tensor; "
                   "multi_env_dict = RLlib's send_actions(MultiEnvDict); rows_read also reads every (env, agent) entry back "
                   "(real dicts, built for all B instances in one vectorised pass on first access)")
    return res


def bench_collection(dist, group, dev, traj, T, B, world, rank, sync_barrier, max_over_ranks, n_agents):
    """rollout collection of one T-step fragment: ONE flat RCCL all-gather (payload without the
    all-zero `terminated` plane, `truncated` bit-packed), and the produce + collect pipeline."""
    from phantom_amd import distributed as phd
    from phantom_amd.distributed import TrajectoryGather, device_env_collector
    if world == 1:
        os.environ["PHX_FORCE_COLLECTIVE"] = "1"           # a world of one rank still runs all_gather_into_tensor on the RCCL group (VERDICT r3 #4a)
    tg = TrajectoryGather(dev, T, group=group)             # one flat buffer; the gathered payload is its prefix
    dev.rollout(T, out=tg.traj)
    tg.gather(); sync_barrier()
    mode = phd.LAST_MODE["mode"]                           # "copy" / "collective:nccl" / "host-staged:gloo": what the numbers below timed
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        tg.gather()
    sync_barrier()
    ag = max_over_ranks((time.perf_counter() - t0) / reps)
    got = tg.unpack(rank)
    assert torch.equal(got.observations, tg.traj.observations) and torch.equal(got.truncations, tg.traj.truncations)
    res = {"ms": ag * 1e3, "bytes_per_rank": tg.nbytes, "raw_trajectory_bytes_per_rank": tg.raw_nbytes,
           "recv_GBps_per_rank": tg.nbytes * (world - 1) / ag / 1e9,
           "per_link_GBps_if_direct": tg.nbytes / ag / 1e9,
           "payload": tg.describe(), "timed": mode,
           "note": "one T=100 fragment, one all_gather_into_tensor (see `timed`: collective:nccl = RCCL on the device buffers, also with one "
                   "rank; copy = a world-1 device copy; host-staged:gloo = through pinned memory); not in `value`"}
    # produce + collect, pipelined: chunk c is gathered on a side stream while chunk c+1 rolls out
    col = device_env_collector(dev, T, group=group)        # chunking by bytes (auto_chunk)
    col.collect(); sync_barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        col.collect()
    sync_barrier()
    pc = max_over_ranks((time.perf_counter() - t0) / reps)
    res["pipelined_rollout_plus_gather_ms"] = pc * 1e3
    res["pipelined_agent_steps_per_sec"] = n_agents * B * world * T / pc
    res["pipeline"] = f"{col.n_chunks} chunk(s) of {col.chunk} steps, 2 staging buffers, gather on a side stream"
    return res


def bench_config4(ph, dist, group, world, rank, local_rank, sync_barrier, max_over_ranks):
    from phantom_amd.distributed import device_env_collector
    nS, nK, Bc, T = 51, 4, 8192, 100
    A = 1 + nS + nS * nK
    env = ph.SupplyChainEnv(n_shops=nS, customers_per_shop=nK, num_steps=T, batch_size=Bc, seed=42,
                            env_offset=rank * Bc, exogenous="device", device=f"cuda:{local_rank}")
    env.reset(); dev = env._device()
    tr = dev.rollout(T)
    for _ in range(3):
        dev.rollout(T, out=tr)
    sync_barrier()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        dev.rollout(T, out=tr)
    torch.cuda.synchronize()
    dt = max_over_ranks((time.perf_counter() - t0) / n)
    sync_barrier()
    res = {"workload": f"SC256 plain env, B={Bc} per GPU x {world} GPUs (global {Bc * world}), T={T} fused rollouts",
           "rollout_ms_per_fragment": dt * 1e3,
           "agent_steps_per_sec_gather_excluded": A * Bc * world * T / dt}
    del tr
    torch.cuda.empty_cache()
    col = device_env_collector(dev, T, chunk=10, group=group)   # 10-step chunks (~92 MB per rank), SURVEY 8e (i)
    col.collect(); sync_barrier()
    t0 = time.perf_counter()
    for _ in range(3):
        col.collect()
    sync_barrier()
    pc = max_over_ranks((time.perf_counter() - t0) / 3)
    res["rollout_plus_allgather_ms_per_fragment"] = pc * 1e3
    res["agent_steps_per_sec_gather_included"] = A * Bc * world * T / pc
    res["gather_bytes_per_rank"] = col.nbytes * col.n_chunks
    from phantom_amd import distributed as phd
    res["timed"] = phd.LAST_MODE["mode"]
    res["pipeline"] = f"{col.n_chunks} chunks of {col.chunk} steps on a side stream, payload: {col.payload}"
    return res


if __name__ == "__main__":
    main()

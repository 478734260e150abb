#!/usr/bin/env python
"""DESIGN 3.2b as a check: no load inside a store loop.  On gfx950 vector loads and stores share `vmcnt`, so a `global_load` / `flat_load` /
`scratch_load` (a spill's reload) in a rollout kernel's step loop makes the wave wait for the trajectory stores it has in flight -- and the wait
is static: a load behind a condition that never holds still leaves its `s_waitcnt vmcnt(0)` at the join.  This tool compiles the kernel files
to ISA (hipcc -S --cuda-device-only, the library's flags), finds each listed instantiation's step loop (the LAST depth-1 loop of the kernel
that contains a global store) and counts what must not be there.
    python tools/isa_lint.py            # table + exit status 1 if an instantiation that has to be clean is not
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phantom_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-I", CSRC]

# (file, regex over the mangled kernel name, what it is)
CLEAN = [
    ("phx_sc_policy.hip", r"phx_sc_rollout_policy_kernelILi\dELb[01]ELi\d+ELb0ELb[01]EE", "policy kernel, device-drawn orders (EXO = false)"),
    ("phx_stk_fused.hip", r"phx_stk_rollout_kernelILb0ELi\d+ELi1EE", "Stackelberg rollout, FAST = 1 (packed static graph, random policy)"),
    ("phx_sc_fused.hip", r"phx_sc_rollout_fsm_kernelILb[01]ELb1EE", "FSM lane-per-pair rollout, STATIC"),
    ("phx_sc_fused.hip", r"phx_sc_rollout_fsm_lean_kernel", "FSM lean rollout loop"),
    ("phx_ads_fused.hip", r"phx_ads_kernelILi\d+ELb1ELb0EE", "ads rollout, device draws (REPLAY = false)"),
    ("phx_generic.hip", r"phx_sched_step_kernelILi\d+ELb1ELb1EE", "engine's T-step loop, PURE"),
]
INFO = [  # reported, not required to be clean: loads are part of what they do
    ("phx_sc_policy.hip", r"phx_sc_rollout_policy_kernelILi\dELb[01]ELi\d+ELb1ELb[01]EE", "policy kernel, replayed orders"),
    ("phx_stk_fused.hip", r"phx_stk_rollout_kernelILb0ELi\d+ELi2EE", "Stackelberg rollout, FAST = 2 (replayed actions)"),
    ("phx_stk_fused.hip", r"phx_stk_rollout_kernelILb[01]ELi\d+ELi0EE", "Stackelberg rollout, general form"),
    ("phx_sc_fused.hip", r"phx_sc_rollout_fsm_kernelILb[01]ELb0EE", "FSM lane-per-pair rollout, general form"),
    ("phx_ads_fused.hip", r"phx_ads_kernelILi\d+ELb1ELb1EE", "ads rollout, replays"),
    ("phx_generic.hip", r"phx_sched_step_kernelILi\d+ELb1ELb0EE", "engine's T-step loop, general form"),
]


def isa_of(src, cache):
    out = os.path.join(cache, src.replace(".hip", ".s"))
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC)):
        subprocess.check_call([HIPCC] + FLAGS + ["-o", out, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(lines):
    cur, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, start = m.group(1), i
        elif cur and "s_endpgm" in l:
            yield cur, lines[start:i + 1]
            cur = None


def step_loop(body):
    heads = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l) and "Loop Header: Depth=1" in l]
    # a depth-1 loop's blocks carry "in Loop: Header=BB<k>_<n> Depth=1" (or are nested in one that does); its extent = first .. last such line
    best = None
    for h in heads:
        lab = re.match(r"^\.L(BB\d+_\d+):", body[h]).group(1)
        tag = f"Header={lab} Depth=1"
        last = max([i for i, l in enumerate(body) if tag in l] + [h])
        while last + 1 < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[last + 1]):      # to the end of the last block
            last += 1
        seg = body[h:last + 1]
        if any(re.search(r"\b(global|buffer)_store", l) for l in seg) and (best is None or len(seg) > len(best)):
            best = seg
    return best


def count(seg):
    c = {"load": 0, "flat": 0, "scratch": 0, "vmwait": 0, "store": 0}
    for l in seg:
        t = l.strip()
        if re.match(r"(global|buffer)_load", t): c["load"] += 1
        elif re.match(r"flat_(load|store|atomic)", t): c["flat"] += 1
        elif re.match(r"scratch_load", t): c["scratch"] += 1
        elif t.startswith("s_waitcnt") and "vmcnt" in t: c["vmwait"] += 1; c.setdefault("waits", []).append(re.search(r"vmcnt\((\d+)\)", t).group(1))
        elif re.match(r"(global|buffer)_store", t): c["store"] += 1
    return c


def run(verbose=True, only_clean=False):
    """-> (number of instantiations that have to be clean and are not, report text)"""
    from concurrent.futures import ThreadPoolExecutor
    cache = os.path.join(tempfile.gettempdir(), "phx_isa_lint")
    os.makedirs(cache, exist_ok=True)
    tables = (("must be clean", CLEAN, True),) + (() if only_clean else (("for information", INFO, False),))
    files = sorted({src for _, t, _ in tables for src, _, _ in t})
    with ThreadPoolExecutor(max_workers=len(files)) as pool:         # (the files compile independently: ~90 s for the longest)
        isa = dict(zip(files, pool.map(lambda f: isa_of(f, cache), files)))
    out = []
    bad = 0
    for title, table, must in tables:
        out.append(f"== {title}")
        for src, rx, what in table:
            n = 0
            for name, body in kernels(isa[src]):
                if not re.search(rx, name):
                    continue
                n += 1
                seg = step_loop(body)
                if seg is None:
                    out.append(f"  {name[:84]:84s} no store loop found"); bad += must; continue
                c = count(seg)
                dirty = c["load"] + c["flat"] + c["scratch"] + c["vmwait"]
                out.append(f"  {name[:84]:84s} loop {len(seg):5d} lines  stores {c['store']:3d}  loads {c['load']:3d}  flat {c['flat']:2d}  spill reloads {c['scratch']:2d}  vmcnt waits {c['vmwait']:3d} {','.join(c.get('waits', []))[:24]:24s}" +
                           ("   <-- " + what if (must and dirty) else ""))
                bad += 1 if (must and dirty) else 0
            if n == 0:
                out.append(f"  (no instantiation matches {rx})"); bad += must
    out.append("clean" if bad == 0 else f"{bad} instantiation(s) that have to be clean are not")
    return bad, "\n".join(out)


def main():
    bad, text = run()
    print(text)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

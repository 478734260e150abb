"""In-tree build of libphantom_amd.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m phantom_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIB_DIR, "libphantom_amd.so")
SOURCES = ["phx_api.hip", "phx_generic.hip", "phx_sc_fused.hip", "phx_sc_rollout.hip", "phx_sc_rollout_sw.hip", "phx_sc_rollout_fsm.hip", "phx_sc_policy.hip", "phx_stk_fused.hip",
           "phx_ads_fused.hip"]
HEADERS = ["phx_dev.h", "phx_epilogue.h", "phx_sc_fast.h", "phx_generic_sched.hip", os.path.join("..", "..", "include", "phantom_amd.h")]
# -ffp-contract=off: rewards are f64 "sales - 0.1*stock" with product and difference rounded
# separately, as the reference's Python floats are (supply_chain.py:147)
# The kernels are written and tuned for gfx950 (MI355X) only.  PHX_OFFLOAD_ARCH="gfx950;gfx942" adds code
# objects for other CDNA parts (untested, untuned); _abi.load_library checks the visible GPU against
# the list the library was built for and fails up front instead of at the first launch.
ARCHS = [a for a in os.environ.get("PHX_OFFLOAD_ARCH", "gfx950").replace(",", ";").split(";") if a]
ARCH_FILE = os.path.join(LIB_DIR, "ARCH")
FLAGS = [f"--offload-arch={a}" for a in ARCHS] + ["-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    try:
        return open(ARCH_FILE).read().strip().split(";") != ARCHS
    except OSError:
        return True


def build(force=False, verbose=True):
    """Compile every translation unit (in parallel: the kernels are independent files) and link the library."""
    if not force and not needs_build():
        return LIB
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    cc = hipcc()
    cflags = [f for f in FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="phx_build_") as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
            cmd = [cc] + cflags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        cmd = [cc] + [f"--offload-arch={a}" for a in ARCHS] + ["-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(ARCH_FILE, "w") as f:
        f.write(";".join(ARCHS) + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

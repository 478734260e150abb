"""DESIGN 3.2b as a build-time check (CPU: hipcc cross-compiles gfx950 here): no global / flat load, no spill reload and no s_waitcnt vmcnt
inside the step loop of the rollout kernels' device-drawn instantiations -- on gfx950 loads and stores share one counter, so any of them
makes the wave wait for the trajectory stores it has in flight (tools/isa_lint.py).  Last file of the suite on purpose: it compiles five
kernel files to ISA (~1.5 minutes)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="hipcc not found")
def test_no_load_inside_the_store_loops_of_the_rollout_kernels():
    import isa_lint
    bad, text = isa_lint.run(only_clean=True)
    assert bad == 0, text

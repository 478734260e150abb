#!/usr/bin/env python
"""Development: build an experimental libphantom_amd variant into scratch/lib/ for A/B runs on the GPU box
(select with PHX_LIB_PATH=scratch/lib/libphantom_<name>.so).  Only the named translation units are recompiled with
the extra flags; the others come from an object cache keyed by their mtime.

    python tools/build_variant.py NAME [-DFOO ...] [--units phx_sc_rollout.hip,...]"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantom_amd import build as B

name, rest = sys.argv[1], sys.argv[2:]
units = ["phx_sc_rollout.hip"]
flags = []
i = 0
while i < len(rest):
    if rest[i] == "--units":
        units = rest[i + 1].split(","); i += 2
    else:
        flags.append(rest[i]); i += 1
cache = os.path.join(ROOT, "scratch", "objcache"); out = os.path.join(ROOT, "scratch", "lib")
os.makedirs(cache, exist_ok=True); os.makedirs(out, exist_ok=True)
cc = B.hipcc()
cflags = [f for f in B.FLAGS if f != "-shared"] + ["-w"]
hdr_t = max(os.path.getmtime(os.path.join(B.CSRC, h)) for h in B.HEADERS)

def obj_for(src):
    special = src in units and flags
    tag = (name if special else "base")
    obj = os.path.join(cache, f"{os.path.splitext(src)[0]}.{tag}.o")
    srcp = os.path.join(B.CSRC, src)
    if special or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(srcp), hdr_t):
        subprocess.check_call([cc] + cflags + (flags if special else []) + ["-c", srcp, "-o", obj])
    return obj

with ThreadPoolExecutor(max_workers=8) as pool:
    objs = list(pool.map(obj_for, B.SOURCES))
lib = os.path.join(out, f"libphantom_{name}.so")
subprocess.check_call([cc] + [f"--offload-arch={a}" for a in B.ARCHS] + ["-shared", "-fPIC"] + objs + ["-o", lib])
print(lib)

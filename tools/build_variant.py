#!/usr/bin/env python3
"""Developer tool: build the tree's engine library with extra compiler flags (tuning macros of the kernel headers) next to the
product's own, for interleaved A/B runs on one GPU box (bench.py picks it up through HWY_ENGINE_LIB).

    python tools/build_variant.py <name> [-DMACRO=value ...]   ->   tools/ablate/_build/libhwy_engine_<name>.so
Several variants at once (parallel compiles): python tools/build_variant.py name1 -DX=1 -- name2 -DX=2 -- ...
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from highwayenv_amd.build import CSRC, HIPCC_FLAGS, build_engine  # noqa: E402

OUT = os.path.join(ROOT, "tools", "ablate", "_build")
os.makedirs(OUT, exist_ok=True)
build_engine()  # hwy_engine.o / hwy_comm.o of the tree (the host side does not depend on the kernels' tuning macros)
groups, cur = [], []
for a in sys.argv[1:]:
    if a == "--":
        groups.append(cur)
        cur = []
    else:
        cur.append(a)
if cur:
    groups.append(cur)
procs = []
for g in groups:
    name, flags = g[0], g[1:]
    obj = os.path.join(OUT, f"hwy_kernels_{name}.o")
    procs.append((name, obj, subprocess.Popen(["hipcc", *HIPCC_FLAGS, *flags, "-c", os.path.join(CSRC, "hwy_kernels.hip"), "-o", obj],
                                              stderr=subprocess.DEVNULL)))
for name, obj, pr in procs:
    if pr.wait() != 0:
        raise SystemExit(f"{name}: compile failed")
    lib = os.path.join(OUT, f"libhwy_engine_{name}.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, os.path.join(CSRC, "hwy_engine.o"),
                    os.path.join(CSRC, "hwy_comm.o"), "-ldl"], check=True)
    os.remove(obj)
    print(lib)

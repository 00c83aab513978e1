#!/usr/bin/env python3
"""Developer tool: build the engine library of another git revision next to the tree's own, for interleaved A/B runs on one
GPU box (bench.py picks it up through HWY_ENGINE_LIB; tools/abn_bench.sh).

    python tools/build_rev.py <rev> [name]   ->   tools/ablate/_build/libhwy_engine_<name or rev>.so
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from highwayenv_amd.build import HIPCC_FLAGS, SOURCES  # noqa: E402

rev = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else rev
out = os.path.join(ROOT, "tools", "ablate", "_build", f"libhwy_engine_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
with tempfile.TemporaryDirectory() as tmp:
    tar = subprocess.run(["git", "-C", ROOT, "archive", rev, "highwayenv_amd/csrc", "include"], check=True, capture_output=True).stdout
    subprocess.run(["tar", "-x", "-C", tmp], input=tar, check=True)
    csrc = os.path.join(tmp, "highwayenv_amd", "csrc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(csrc, src.replace(".hip", ".o"))
        procs.append(subprocess.Popen(["hipcc", *HIPCC_FLAGS, "-c", os.path.join(csrc, src), "-o", obj], stderr=subprocess.DEVNULL))
        objs.append(obj)
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("compile failed")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-ldl"], check=True)
print(out)

"""Build the gfx950 HIP engine in-tree: highwayenv_amd/csrc/libhwy_engine.so.

``hipcc --offload-arch=gfx950`` cross-compiles without a GPU.  ``-ffp-contract=off`` keeps every
``a*b+c`` double-rounded like the reference's numpy scalar arithmetic (see hwy_device.h).
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libhwy_engine.so")
SOURCES = ["hwy_kernels.hip", "hwy_engine.hip", "hwy_comm.hip"]
import glob

# every header the two translation units can include: csrc/*.h (hwy_device.h, hwy_wave.h, hwy_net.h, hwy_ix.h, ...)
# plus the public ABI header -- globbed so that a new kernel header can never be forgotten by is_stale()
HEADERS = sorted(os.path.basename(h) for h in glob.glob(os.path.join(CSRC, "*.h"))) + [
    os.path.join("..", "..", "include", "hwy_engine.h")]
# -disable-machine-licm: MachineLICM hoists the materialisation of every f64 literal (two v_mov_b32 each) and of most LDS
# offsets out of the frame loop, where they sit in ~50 VGPRs for the whole loop (tools/vgpr_pressure.py).  Without it the
# road-network kernel fits 128 VGPRs with NO spills (181 natural / 64 spilled before), the one-wavefront kernel 102 (118),
# the intersection kernel 150 (203); every workload measured 0.3 .. 3 % faster (profiles/r03_history.md).
# -amdgpu-sched-strategy=iterative-ilp: with the registers the first flag frees, the ILP-first scheduler shortens the dependent
# f64 chains the slowest wavefronts of a launch wait on (interleaved A/B on one box: headline 45.38 -> 44.90 us, merge config 5
# 317.5 -> 313.5, highway-v0 135.9 -> 134.4; max-ilp / max-memory-clause / the occupancy bias: within +-0.5 %).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-mllvm", "-disable-machine-licm",
               "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X engine cannot be built (there is no CPU fallback)")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def kernel_source_hash() -> str:
    """sha256 (first 16 hex digits) over every source the DEVICE code is built from (hwy_kernels.hip and the csrc headers
    it includes) + the hipcc flags: the identity of a kernel build.  rocprofv3 counter summaries under profiles/ record it,
    and bench.py only quotes counters whose hash equals the running build's.  (The host side of the library --
    hwy_engine.hip, hwy_comm.hip -- is not part of it: it picks a kernel variant, whose name the summaries record too.)"""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for f in sorted(["hwy_kernels.hip"] + [x for x in HEADERS if not x.startswith("..")]):
        h.update(os.path.basename(f).encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_engine(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, *HIPCC_FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_engine(force=True, verbose=True))

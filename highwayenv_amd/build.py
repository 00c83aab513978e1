"""Build the gfx950 HIP engine in-tree: highwayenv_amd/csrc/libhwy_engine.so.

``hipcc --offload-arch=gfx950`` cross-compiles without a GPU.  ``-ffp-contract=on``: see FP_CONTRACT below.
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
# FP_CONTRACT: "on" = an a*b+c written in ONE source expression is one fused multiply-add (clang forms llvm.fmuladd in the front
# end, gfx950 lowers every f64 fmuladd to v_fma_f64); products and sums of different statements stay separate.  Which operations
# fuse is therefore a property of the SOURCE, the same in every kernel the shared step functions are inlined into -- the K-step
# launch stays bit-identical to K one-step launches (tests/test_rollout.py).  "fast" (fuse whatever the optimiser finds after
# inlining) fuses 4 % more sites and measured 1 ulp apart between hwy_step_kernel and hwy_rollout_kernel on the reward
# (profiles/r04_history.md).  "off" (rounds 1-3) rounded every a*b+c twice like the reference's numpy scalars: +6.6 % on the
# headline launch.  The per-frame comparisons with the reference are at 1e-9 and unaffected; ONE bound moved with this setting:
# the lateral offset of intersection cars between 1 and 2 m/s after 15 free-running frames (steering_control divides by
# not_zero(speed) twice and amplifies the fused roundings: 4.8e-8 measured) is held at 1e-6 where the unfused build held 1e-8
# (tests/test_ix_parity.py, tests/golden_util.py: slow_atol) -- and the exact comparisons (lane indices, flags) now rest on the
# decisions being well conditioned, not on identical arithmetic: DESIGN.md section 4 states both as part of the parity contract.
# tests/emu builds the CPU emulator with the same front end and the same setting.
FP_CONTRACT = "on"
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", f"-ffp-contract={FP_CONTRACT}", "-fPIC", "-mllvm", "-disable-machine-licm",
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


ASAN_LIB_PATH = os.path.join(CSRC, "libhwy_engine_asan.so")


def asan_runtime() -> str:
    """ROCm clang's shared AddressSanitizer runtime (to LD_PRELOAD under an uninstrumented python)."""
    hits = glob.glob(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "lib", "clang", "*", "lib", "linux",
                                  "libclang_rt.asan-x86_64.so"))
    if not hits:
        raise RuntimeError("libclang_rt.asan-x86_64.so not found under $ROCM_PATH/lib/llvm")
    return sorted(hits)[-1]


def build_engine_asan(force: bool = False, verbose: bool = False) -> str:
    """libhwy_engine_asan.so: the HOST side of the library (hwy_engine.hip: argument validation, staging buffers, event
    bookkeeping, state packing; hwy_comm.hip) instrumented with AddressSanitizer; the kernels are the
    product's own object code (ASan for gfx950 device code needs xnack+, which this part does not run with).  Test
    infrastructure: tests/test_asan_host.py runs the ABI tests and a parity test on it with the runtime preloaded."""
    srcs = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    if not force and os.path.exists(ASAN_LIB_PATH) and os.path.getmtime(ASAN_LIB_PATH) >= max(os.path.getmtime(f) for f in srcs):
        return ASAN_LIB_PATH
    hipcc = _hipcc()
    # the kernels' object file is the product's own (hwy_kernels.o): rebuilt if the library is stale OR the object did not travel
    # with it (*.o is git-ignored; a fresh .so without its objects would leave the link below without an input)
    build_engine(force=not os.path.exists(os.path.join(CSRC, "hwy_kernels.o")))
    objs = []
    for src in SOURCES:
        if src == "hwy_kernels.hip":
            objs.append(os.path.join(CSRC, "hwy_kernels.o"))
            continue
        obj = os.path.join(CSRC, src.replace(".hip", "_asan.o"))
        flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address", "-shared-libsan",
                 "-fno-omit-frame-pointer", "-Wno-option-ignored"]
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", "-o", ASAN_LIB_PATH, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return ASAN_LIB_PATH


def kernel_resources(lib_path: str = LIB_PATH) -> dict:
    """What the compiler allocated to every kernel of the gfx950 code object inside the built library, read from the code
    object's own metadata (NT_AMDGPU_METADATA note: the numbers the hardware dispatcher uses), keyed by demangled kernel name
    without the argument list: ``{"hwy::hwy_step_wave_kernel<3, false>": {"vgpr": 102, "vgpr_spill": 0, "sgpr": 106,
    "sgpr_spill": 76, "lds": 8080, "scratch": 36, "workgroup": 64}, ...}``.  Needs no GPU.  Used by tests/test_kernel_resources.py
    (the occupancy DESIGN.md quotes for each step kernel is a property of the build, checked where the build is checked) and
    tools/kernel_table.py --so."""
    import re
    import struct
    import msgpack  # (build / test time only)
    blob = open(lib_path, "rb").read()
    o = blob.index(b"__CLANG_OFFLOAD_BUNDLE__")  # hipcc's fat binary: magic[24], u64 n, n x (u64 offset, u64 size, u64 len, triple)
    n, q, co = struct.unpack_from("<Q", blob, o + 24)[0], o + 32, None
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", blob, q)
        if b"gfx950" in blob[q + 24:q + 24 + tl]:
            co = blob[o + off:o + off + size]
        q += 24 + tl
    if co is None or co[:4] != b"\x7fELF":
        raise RuntimeError(f"{lib_path}: no gfx950 code object in the offload bundle")
    shoff = struct.unpack_from("<Q", co, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", co, 0x3A)
    meta = None
    for i in range(shnum):
        sh = struct.unpack_from("<IIQQQQIIQQ", co, shoff + i * shentsize)
        if sh[1] != 7:  # SHT_NOTE
            continue
        d, k = co[sh[4]:sh[4] + sh[5]], 0
        while k < len(d):
            nsz, dsz, ty = struct.unpack_from("<III", d, k)
            k += 12 + ((nsz + 3) & ~3)
            if ty == 32:  # NT_AMDGPU_METADATA (msgpack)
                meta = msgpack.unpackb(d[k:k + dsz], raw=False)
            k += (dsz + 3) & ~3
    if meta is None:
        raise RuntimeError(f"{lib_path}: the gfx950 code object carries no AMDGPU metadata note")
    kernels = meta["amdhsa.kernels"]
    names = subprocess.run(["c++filt"] + [k[".name"] for k in kernels], capture_output=True, text=True, check=True).stdout.split("\n")
    out = {}
    for k, name in zip(kernels, names):
        out[re.sub(r"\(.*\)$", "", name).replace("void ", "")] = {
            "vgpr": k[".vgpr_count"], "vgpr_spill": k[".vgpr_spill_count"], "sgpr": k[".sgpr_count"],
            "sgpr_spill": k[".sgpr_spill_count"], "lds": k[".group_segment_fixed_size"],
            "scratch": k[".private_segment_fixed_size"], "workgroup": k[".max_flat_workgroup_size"]}
    return out


if __name__ == "__main__":
    print(build_engine(force=True, verbose=True))

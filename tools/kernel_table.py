#!/usr/bin/env python3
"""Developer tool: register / LDS / scratch table of every kernel in the gfx950 code object (from the assembly that
tools/asm_loop_stats.py leaves in /tmp/hwy_asm_base, or a fresh -save-temps compile: `--fresh`).
`--so [path]`: the same table straight from the metadata of the BUILT library (highwayenv_amd.build.kernel_resources: what the
dispatcher reads; no recompilation, no scratch-instruction count)."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd.build import HIPCC_FLAGS  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--so" in sys.argv:
    from highwayenv_amd import build  # noqa: E402
    k = sys.argv.index("--so")
    res = build.kernel_resources(sys.argv[k + 1]) if len(sys.argv) > k + 1 else build.kernel_resources()
    print(f"{'kernel':52s} {'vgpr':>5s} {'spill':>5s} {'sgpr':>5s} {'s-spill':>7s} {'LDS B':>6s} {'priv B':>6s} {'wg':>4s} {'waves/SIMD':>10s} {'wg/CU (LDS)':>11s}")
    for name, r in res.items():
        occ = min(8, 512 // (((r['vgpr'] + 7) // 8) * 8))
        print(f"{name.replace('hwy::', ''):52s} {r['vgpr']:5d} {r['vgpr_spill']:5d} {r['sgpr']:5d} {r['sgpr_spill']:7d} {r['lds']:6d} "
              f"{r['scratch']:6d} {r['workgroup']:4d} {occ:10d} {(160 * 1024 // r['lds']) if r['lds'] else 0:11d}")
    sys.exit(0)
OUT = os.path.join("/tmp", "hwy_asm_base")  # (150 MB of compiler temporaries: kept out of the tree that gpurun ships)
ASM = os.path.join(OUT, "hwy_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
if "--fresh" in sys.argv or not os.path.exists(ASM):
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["hipcc", *HIPCC_FLAGS, "-gline-tables-only",
                    "-c", os.path.join(ROOT, "highwayenv_amd", "csrc", "hwy_kernels.hip"), "-o", os.path.join(OUT, "k.o"),
                    "-save-temps=obj", *os.environ.get("HWY_EXTRA_FLAGS", "").split()], check=True, capture_output=True, cwd=OUT)
s = open(ASM).read()
meta = s[s.index("amdhsa.kernels:"):]
print(f"{'kernel':58s} {'vgpr':>5s} {'spill':>5s} {'sgpr':>5s} {'s-spill':>7s} {'LDS B':>6s} {'priv B':>6s} {'scratch ops':>11s} {'waves/SIMD':>10s}")
for blk in meta.split("  - .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\.%s:\s+(\S+)" % k, blk)
        return m.group(1) if m else "?"
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().replace("hwy::", "")
    name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
    v = int(g("vgpr_count"))
    sym = g("name")
    a = s.index("\n" + sym + ":")
    body = s[a:s.index(".Lfunc_end", a)]
    n_scratch = len(re.findall(r"^\s+(scratch_|buffer_)(load|store)", body, flags=re.M))
    occ = min(8, 512 // max(v, 1)) if v else 8
    print(f"{name:58s} {v:5d} {g('vgpr_spill_count'):>5s} {g('sgpr_count'):>5s} {g('sgpr_spill_count'):>7s} "
          f"{g('group_segment_fixed_size'):>6s} {g('private_segment_fixed_size'):>6s} {n_scratch:11d} {occ:10d}")

#!/usr/bin/env python3
"""Developer tool: static instruction statistics of the frame loop of hwy_step_wave_kernel<3,false>.

    python tools/asm_loop_stats.py            # compiles highwayenv_amd/csrc/hwy_kernels.hip with -save-temps

Prints the opcode mix of the frame-loop body, the SGPR-spill traffic in it (v_writelane / v_readlane with an
immediate lane = spill slot) and what the reloaded values feed (source line), which is how the SGPR pressure
of the loop was tracked down (profiles/r01_history.md).
"""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd.build import HIPCC_FLAGS  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (150 MB of compiler temporaries: kept out of the tree that gpurun ships).  HWY_ASM_OUT / HWY_EXTRA_FLAGS: another directory
# and extra hipcc flags (appended, so they override build.HIPCC_FLAGS) -- how a flag experiment is counted before it is timed:
#   HWY_ASM_OUT=/tmp/hwy_asm_fma HWY_EXTRA_FLAGS=-ffp-contract=fast python tools/asm_loop_stats.py
OUT = os.environ.get("HWY_ASM_OUT", os.path.join("/tmp", "hwy_asm_base"))
KERNEL = "_ZN3hwy20hwy_step_wave_kernelILi3ELb0EEEvNS_10StepParamsE"
if "--kernel" in sys.argv:
    KERNEL = sys.argv.pop(sys.argv.index("--kernel") + 1)
    sys.argv.remove("--kernel")


def main():
    os.makedirs(OUT, exist_ok=True)
    # HWY_ASM_SRC: another translation unit than the product's (tools/mini/*.hip instantiate ONE kernel each: seconds, not minutes)
    src = os.path.abspath(os.environ.get("HWY_ASM_SRC", os.path.join(ROOT, "highwayenv_amd", "csrc", "hwy_kernels.hip")))
    subprocess.run(["hipcc", *HIPCC_FLAGS, "-I", os.path.join(ROOT, "highwayenv_amd", "csrc"),
                    "-gline-tables-only", "-c", src,
                    "-o", os.path.join(OUT, "k.o"), "-save-temps=obj", *os.environ.get("HWY_EXTRA_FLAGS", "").split()],
                   check=True, capture_output=True, cwd=OUT)
    s = open(os.path.join(OUT, os.path.basename(src)[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    files = {}
    for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]+)"(?:\s+"([^"]+)")?', s):
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    start = s.index(KERNEL + ":")
    end = s.index(".Lfunc_end", start)
    ins, loc, labels, cur = [], [], {}, None
    for t in (l.strip() for l in s[start:end].split("\n")):
        if not t or t.startswith((";", "//")):
            continue
        m = re.match(r"^(\.?L?BB\w+|\.L\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith(".loc"):
            p = t.split()
            cur = (files.get(int(p[1]), p[1]), int(p[2]))
            continue
        if t.startswith("."):
            continue
        ins.append(t)
        loc.append(cur)
    # the frame loop = the longest backward unconditional branch
    best = (0, 0, 0)
    for k, t in enumerate(ins):
        m = re.match(r"s_branch\s+(\S+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= k and k - labels[m.group(1)] > best[0]:
            best = (k - labels[m.group(1)], labels[m.group(1)], k)
    _, lo, hi = best
    meta = {k: re.search(r"\.%s:\s+(\d+)" % k, s[s.index(".name:           " + KERNEL) - 400:s.index(".name:           " + KERNEL) + 600])
            for k in ("sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "sgpr_count")}
    print({k: (int(v.group(1)) if v else None) for k, v in meta.items()})
    body = ins[lo:hi + 1]
    ops = collections.Counter(t.split()[0] for t in body)
    print(f"frame loop: instructions {lo}..{hi} ({len(body)}), VALU {sum(c for o, c in ops.items() if o.startswith('v_'))}, "
          f"SALU {sum(c for o, c in ops.items() if o.startswith('s_'))}")
    print("top opcodes:", ops.most_common(int(sys.argv[1]) if len(sys.argv) > 1 else 25))
    spill_vgprs = collections.Counter(re.match(r"v_writelane_b32\s+(v\d+)", t).group(1) for t in ins if t.startswith("v_writelane"))
    print("spill VGPRs (writelane targets, whole kernel):", dict(spill_vgprs))
    n_rd = n_wr = 0
    feeds = collections.Counter()
    for k in range(lo, hi + 1):
        t = ins[k]
        m = re.match(r"v_readlane_b32\s+(s\d+),\s*(v\d+),\s*(\d+)\s*$", t)
        if m and m.group(2) in spill_vgprs:
            n_rd += 1
            sreg = int(m.group(1)[1:])
            for j in range(k + 1, min(k + 40, hi + 1)):
                u = ins[j]
                if u.startswith("v_readlane"):
                    continue
                hit = re.search(r"\bs%d\b" % sreg, u) or any(int(a) <= sreg <= int(b) for a, b in re.findall(r"s\[(\d+):(\d+)\]", u))
                if hit:
                    feeds[loc[j]] += 1
                    break
        if t.startswith("v_writelane"):
            n_wr += 1
    print(f"spill traffic in the loop: {n_rd} v_readlane + {n_wr} v_writelane")
    print("reloads feed (file, line): count ->", feeds.most_common(30))


if __name__ == "__main__":
    main()

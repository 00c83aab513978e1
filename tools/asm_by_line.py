#!/usr/bin/env python3
"""Developer tool: the frame loop of a step kernel, instruction counts by source line (static), and optionally the annotated
listing.  Uses the assembly tools/asm_loop_stats.py leaves in tools/ablate/_build/asm_base.

    python tools/asm_by_line.py [--list] [--kernel MANGLED]
"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.environ.get("HWY_ASM_FILE", os.path.join("/tmp", "hwy_asm_base", "hwy_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"))
KERNEL = "_ZN3hwy20hwy_step_wave_kernelILi3ELb0EEEvNS_10StepParamsE"
if "--kernel" in sys.argv:
    KERNEL = sys.argv[sys.argv.index("--kernel") + 1]
s = open(ASM).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]+)"(?:\s+"([^"]+)")?', s):
    files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = s.index(KERNEL + ":")
end = s.index(".Lfunc_end", start)
ins, loc, labels, lab_at, cur = [], [], {}, collections.defaultdict(list), None
for t in (l.strip() for l in s[start:end].split("\n")):
    if not t or t.startswith((";", "//")):
        continue
    m = re.match(r"^(\.?L?BB\w+|\.L\w+):", t)
    if m:
        labels[m.group(1)] = len(ins)
        lab_at[len(ins)].append(m.group(1))
        continue
    if t.startswith(".loc"):
        p = t.split()
        cur = (files.get(int(p[1]), p[1]), int(p[2]))
        continue
    if t.startswith("."):
        continue
    ins.append(t.split(";")[0].strip())
    loc.append(cur)
best = (0, 0, 0)
for k, t in enumerate(ins):
    m = re.match(r"s_branch\s+(\S+)", t)
    if m and m.group(1) in labels and labels[m.group(1)] <= k and k - labels[m.group(1)] > best[0]:
        best = (k - labels[m.group(1)], labels[m.group(1)], k)
_, lo, hi = best
if "--whole" in sys.argv:
    lo, hi = 0, len(ins) - 1
by = collections.defaultdict(lambda: [0, 0])
for k in range(lo, hi + 1):
    op = ins[k].split()[0]
    by[loc[k] or ("?", 0)][0 if op.startswith("v_") or op.startswith("ds_") or op.startswith("global_") else 1] += 1
print(f"{KERNEL}: instructions {lo}..{hi}")
tot = collections.defaultdict(lambda: [0, 0])
for (f, l), (v, sc) in sorted(by.items(), key=lambda kv: (kv[0][0] or "", kv[0][1] or 0)):
    tot[f][0] += v
    tot[f][1] += sc
    print(f"  {f}:{l:5d}  vector {v:4d}  scalar {sc:4d}")
print({f: tuple(v) for f, v in tot.items()})
if "--list" in sys.argv:
    for k in range(lo, hi + 1):
        for lb in lab_at.get(k, []):
            print(f"{lb}:")
        f, l = loc[k] if loc[k] else ("?", 0)
        print(f"  {k:5d} {f}:{l:<5d} {ins[k]}")

#!/usr/bin/env python3
"""Developer tool: where does a kernel's VGPR pressure peak, and which values are live there?

Post-RA liveness over the assembly tools/asm_loop_stats.py leaves in /tmp/hwy_asm_base (a control-flow graph from the labels and
branches, backward dataflow on the physical VGPRs; a write under a partial EXEC mask is treated as a full definition, which is
what the allocator's own intervals do for a value first written there).  Prints the peak, the source lines of the instructions
around it and, for every register live at the peak, the source line that wrote it last -- i.e. WHAT is being held across WHAT.

    python tools/asm_loop_stats.py                       # (compiles with -save-temps)
    python tools/vgpr_pressure.py --kernel _ZN3hwy19hwy_net_step_kernelILi1ELb0EEEvNS_9NetParamsE [--top 12] [--agpr]
"""
import collections
import os
import re
import sys

ASM = os.path.join("/tmp", "hwy_asm_base", "hwy_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")


def arg(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


KERNEL = arg("--kernel", "_ZN3hwy20hwy_step_wave_kernelILi3ELb0EEEvNS_10StepParamsE")
TOP = int(arg("--top", "10"))
ASM = arg("--asm", ASM)

NO_DEF = ("global_store", "flat_store", "buffer_store", "scratch_store", "ds_write", "ds_add_u", "ds_max_u", "ds_min_u", "ds_or_b",
          "ds_and_b", "ds_add_f", "ds_max_i", "ds_min_i", "global_atomic", "s_", "v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane",
          "ds_gws", "buffer_wbl2", "buffer_inv", "v_nop", "ds_nop", "exp")
RMW = ("v_writelane", "v_fmac", "v_mac", "v_dot", "v_movrel", "v_pk_fmac")


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main():
    s = open(ASM).read()
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
        ins.append(t.split(";")[0].strip())
        loc.append(cur)
    n = len(ins)
    use, dfn, succ = [None] * n, [None] * n, [None] * n
    for i, t in enumerate(ins):
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        d, u = [], []
        if op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load")) and "lds" in rest.split()[-1:]:
            u = [r for o in ops for r in regs(o)]
        elif op.startswith(NO_DEF) or not ops:
            u = [r for o in ops for r in regs(o)]
        else:
            d = regs(ops[0])
            u = [r for o in ops[1:] for r in regs(o)]
            # two-destination forms (v_div_scale, v_add_co, v_mad_u64_u32): second operand is vcc / an SGPR pair, no VGPR
            if op.startswith(RMW) or "dpp" in op or "row_" in rest or "quad_perm" in rest or "sdwa" in op:
                u += d
        use[i], dfn[i] = set(u), set(d)
        nxt = [i + 1] if i + 1 < n else []
        if op == "s_branch":
            nxt = [labels[ops[0]]] if ops[0] in labels else []
        elif op.startswith("s_cbranch"):
            if ops and ops[-1] in labels:
                nxt.append(labels[ops[-1]])
        elif op in ("s_endpgm",):
            nxt = []
        succ[i] = nxt
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for j in succ[i]:
                out |= live_in[j]
            new = use[i] | (out - dfn[i])
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    # pressure right after instruction i = live_out(i) (+ its defs, which are live at least momentarily)
    press = []
    for i in range(n):
        out = set()
        for j in succ[i]:
            out |= live_in[j]
        press.append(len(out | dfn[i]))
    peak = max(press)
    print(f"{KERNEL}: {n} instructions, peak VGPR pressure {peak} (highest register used: v{max(max(d) for d in dfn if d)})")
    # by source line: the maximum pressure reached on that line
    by_line = collections.defaultdict(int)
    for i in range(n):
        if loc[i]:
            by_line[loc[i]] = max(by_line[loc[i]], press[i])
    print("source lines with the highest pressure:")
    for (f, l), p in sorted(by_line.items(), key=lambda kv: -kv[1])[:TOP * 3]:
        print(f"  {f}:{l:5d}  {p}")
    at = press.index(peak)
    print(f"first peak at instruction {at}: {ins[at]}   [{loc[at]}]")
    out = set()
    for j in succ[at]:
        out |= live_in[j]
    # last writer of each live register, walking backwards in layout order (approximation: good inside straight-line regions)
    writers = collections.Counter()
    detail = {}
    for r in sorted(out):
        j = at
        while j >= 0 and r not in dfn[j]:
            j -= 1
        w = loc[j] if j >= 0 else None
        writers[w] += 1
        detail[r] = (j, w)
    print("live registers at the peak, by the source line that wrote them last:")
    for w, c in writers.most_common():
        print(f"  {str(w):40s} {c}")
    if "--regs" in sys.argv:
        for r, (j, w) in detail.items():
            print(f"  v{r}: instr {j} {ins[j] if j >= 0 else ''}  [{w}]")
    if "--profile" in sys.argv:  # pressure along the layout, one line per 50 instructions
        for i in range(0, n, 50):
            seg = press[i:i + 50]
            print(f"  {i:6d} max {max(seg):4d}  {loc[i]}")


if __name__ == "__main__":
    main()

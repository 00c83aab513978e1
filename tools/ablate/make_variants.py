#!/usr/bin/env python3
"""Developer tool: build ablated copies of the step kernel (sections textually removed) to attribute
GPU time to kernel sections.  Results are NOT valid simulations; used only with tools/phase_timing.py.
Output: tools/ablate/_build/libhwy_engine_<variant>.so (git-ignored, shipped to the GPU box by gpurun)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "highwayenv_amd", "csrc")
OUT = os.path.join(HERE, "_build")


def cut(src, start, end, repl=""):
    i = src.index(start)
    j = src.index(end, i)
    return src[:i] + repl + src[j:]


def wave_variants(dev, wav):
    """Variants of the one-wavefront kernel (hwy_wave.h).  Each value is (device_h_text, wave_h_text)."""
    v = {"wbase": (dev, wav)}
    v["winline"] = (dev.replace("__device__ __attribute__((noinline)) inline int pair_collide", "__device__ inline int pair_collide"), wav)
    v["wnocollide"] = (dev, cut(wav, "    const Body mine{me.x, me.y, me.v, me.ch, me.sh};", "  }  // frames", ""))
    v["wnorank"] = (dev, wav.replace("    bool recount = (fr == 0);", "    bool recount = false;"))
    v["wnomobil"] = (dev, wav.replace("    const bool cl = decide && left_ok", "    const bool cl = false && left_ok").replace("    const bool cr = decide && right_ok", "    const bool cr = false && right_ok"))
    v["wnopow"] = (dev.replace("return r > 0.0 ? log_pos(r) : -__builtin_inf();", "return r;").replace("(1 - exp_bounded(delta * log_ratio))", "(1 - delta * log_ratio)"), wav)
    v["wnosincos"] = (dev, wav.replace("      sincos_bounded(me.h, &me.sh, &me.ch);\n", "      me.sh = me.h; me.ch = 1 - me.h;\n"))
    v["wnosteer"] = (dev, wav.replace("      tb = B::steer_tan_beta(p, me.y, me.h, inv_v, me.tgt);", "      tb = inv_v * 1e-9;"))
    v["wnoobs"] = (dev, wav.replace("  if (p.full_step) observe_wave(p, e, me, true);", ""))
    # cycle-stamped variant
    t = wav
    t = t.replace("  Veh me;\n  load_vehicle<1>(p, e, me);\n  const bool controlled",
                  "  long long t_prev = clock64(); long long acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0};\n"
                  "#define TICK(k) { const long long t_now = clock64(); acc[k] += t_now - t_prev; t_prev = t_now; }\n"
                  "  Veh me;\n  load_vehicle<1>(p, e, me);\n  const bool controlled")
    marks = [("    // ---- A. meta-action (abstract.py:294-304", 0),
             ("    // ---- C. rank along the road ---", 1),
             ("    // lane membership (AbstractLane.on_lane, margin 1) -> bits", 2),
             ("    // ---- D. Road.act: lane-change policy (behavior.py:219-263)", 3),
             ("    const double free_self = B::idm_free_from_log(log_ratio, me.delta);", 4),
             ("    const double self_a = free_self - gap_own;", 5),
             ("    // abort rule for ongoing lane changes: ordered chain", 6),
             ("    // ---- E. Road.act: low-level control", 7),
             ("    // ---- F. Road.step: integrate", 8),
             ("    // ---- G. Road.step: collisions", 9),
             ("  }  // frames", 10)]
    for text, k in marks:
        assert text in t, text
        t = t.replace(text, f"    TICK({k})\n" + text)
    tail = "  if (p.full_step) observe_wave(p, e, me, true);\n  store_vehicle<1>(p, e, me, false);\n}"
    assert tail in t
    t = t.replace(tail,
                  "  if (p.full_step) observe_wave(p, e, me, true);\n  TICK(11)\n  store_vehicle<1>(p, e, me, false);\n"
                  "  if (i == 0 && p.obs) for (int k = 0; k < 12; ++k) p.obs[(size_t)e * p.A * p.V * p.F + k] = (float)acc[k];\n}")
    assert "    if (recount) {  // wave-uniform\n" in t
    t = t.replace("    if (recount) {  // wave-uniform\n", "    if (recount) {  n_recount += 1.0f;\n")
    t = t.replace("  long long t_prev = clock64();", "  float n_recount = 0.0f; long long t_prev = clock64();")
    t = t.replace("p.obs[(size_t)e * p.A * p.V * p.F + k] = (float)acc[k];\n}", "p.obs[(size_t)e * p.A * p.V * p.F + k] = (float)acc[k];\n  if (i == 0 && p.obs) p.obs[(size_t)e * p.A * p.V * p.F + 12] = n_recount;\n}")
    v["wticks"] = (dev, t)
    return v


def variants(src):
    v = {"base": src}
    v["norank"] = cut(src, "    int cnt_lt = 0, cnt_le = 0;", "    if (active) sh.perm[rank] = i;",
                      "    int rank = i; const bool tie = false;\n")
    v["nomobil"] = src.replace("    if (decide) {\n      me.timer = 0.0;", "    if (false) {\n      me.timer = 0.0;")
    v["nopow"] = src.replace("pow(fmax(v, 0.0) / fabs(not_zero(v0)), delta)", "(fmax(v, 0.0) / fabs(not_zero(v0)) * delta)")
    v["nocollide"] = cut(src, "    if (all_check) {\n      // full pairwise", "  }  // frames", "    if (false) {}\n")
    v["nosincos"] = src.replace("      sincos(me.h, &me.sh, &me.ch);\n", "      me.sh = me.h; me.ch = 1 - me.h;\n")
    v["noasin"] = src.replace("clipd(asin(a), -HWY_PI / 4, HWY_PI / 4)", "a")
    v["nosteer"] = src.replace("      tb = B::steer_tan_beta(p, me.y, me.h, inv_v, me.tgt);", "      tb = inv_v * 1e-9;")
    v["nolane"] = src.replace("      me.lane = B::closest_lane(p, me.x, me.y, me.h);  // on_state_update\n", "")
    v["nochain"] = cut(src, "    // abort rule for ongoing lane changes: ordered chain", "    // ---- E. Road.act: low-level control", "")
    v["noneigh"] = src.replace("      if (!has_tie) B::neighbours_ranked(sh, me.lane, rank, &f_own, &r_own);\n      else B::neighbours_scan(p, sh, me.lane, i, me.x, &f_own, &r_own);\n", "")
    # cycle-stamped variant: s_memtime deltas per kernel section, returned through the obs buffer
    t = src
    t = t.replace("  Veh me;\n  load_vehicle<NW>(p, e, me);\n  const bool controlled",
                  "  long long t_prev = clock64(); long long acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0};\n"
                  "#define TICK(k) { const long long t_now = clock64(); acc[k] += t_now - t_prev; t_prev = t_now; }\n"
                  "  Veh me;\n  load_vehicle<NW>(p, e, me);\n  const bool controlled")
    marks = [("    // ---- A. action_type.act (abstract.py:294-304)", 0),      # load / loop overhead
             ("    // ---- C. rank along the road + lane membership masks", 1),  # A+B publish
             ("    // ---- D. Road.act: lane-change policy", 2),                 # C rank+masks
             ("    // abort rule for ongoing lane changes: ordered chain", 3),   # D own neigh + free + mobil
             ("    // ---- E. Road.act: low-level control", 4),                  # chain
             ("    // ---- F. Road.step: integrate", 5),                         # E control
             ("    // ---- G. Road.step: collisions", 6),                        # F integrate
             ("  }  // frames", 7),                                              # G collisions
             ]
    for text, k in marks:
        assert text in t, text
        t = t.replace(text, f"    TICK({k})\n" + text)
    t = t.replace("  store_vehicle<NW>(p, e, me);\n}\n\n}  // namespace hwy",
                  "  TICK(8)\n  store_vehicle<NW>(p, e, me);\n  TICK(9)\n"
                  "  __syncthreads();\n  if (i == 0 && p.obs) for (int k = 0; k < 10; ++k) p.obs[(size_t)e * p.A * p.V * p.F + k] = (float)acc[k];\n"
                  "}\n\n}  // namespace hwy")
    v["ticks"] = t
    return v


def build(name, text):
    wave_text = None
    if isinstance(text, tuple):
        text, wave_text = text
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    for f in ("hwy_kernels.hip", "hwy_engine.hip", "hwy_launch.h", "hwy_params.h", "hwy_wave.h", "hwy_math.h"):
        shutil.copy(os.path.join(CSRC, f), d)
    if wave_text is not None:
        open(os.path.join(d, "hwy_wave.h"), "w").write(wave_text)
    # keep relative include of ../../include working
    os.makedirs(os.path.join(OUT, "..", "..", "include_link"), exist_ok=True)
    open(os.path.join(d, "hwy_device.h"), "w").write(text.replace('#include "../../include/hwy_engine.h"', f'#include "{ROOT}/include/hwy_engine.h"'))
    eng = open(os.path.join(d, "hwy_engine.hip")).read().replace('#include "../../include/hwy_engine.h"', f'#include "{ROOT}/include/hwy_engine.h"')
    open(os.path.join(d, "hwy_engine.hip"), "w").write(eng)
    lib = os.path.join(OUT, f"libhwy_engine_{name}.so")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
    subprocess.run(["hipcc", *flags, "-shared", "-o", lib, os.path.join(d, "hwy_kernels.hip"), os.path.join(d, "hwy_engine.hip")],
                   check=True, capture_output=True)
    shutil.rmtree(d)
    return lib


if __name__ == "__main__":
    src = open(os.path.join(CSRC, "hwy_device.h")).read()
    vs = variants(src)
    vs.update(wave_variants(src, open(os.path.join(CSRC, "hwy_wave.h")).read()))
    only = sys.argv[1:]
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(8) as ex:
        for lib in ex.map(lambda kv: build(*kv), [(k, t) for k, t in vs.items() if not only or k in only]):
            print(lib)

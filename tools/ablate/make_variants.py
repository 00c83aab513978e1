#!/usr/bin/env python3
"""Developer tool: build modified copies of the engine (kernel sections textually removed, or s_memtime
stamps inserted) to attribute GPU time / registers to kernel sections.  The ablated variants are NOT
valid simulations; they are only timed (tools/phase_timing.py, bench.py via HWY_ENGINE_LIB) or, for the
`wticks` variant, read out by tools/section_cycles.py.

    python tools/ablate/make_variants.py [variant ...]      # default: all
    -> tools/ablate/_build/libhwy_engine_<variant>.so       (git-ignored; shipped to the GPU box by gpurun)

Every variant is a list of (file, old, new) substitutions on highwayenv_amd/csrc; a variant whose pattern
no longer matches the source is skipped with a message (the kernels evolve).
"""
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from highwayenv_amd.build import HIPCC_FLAGS  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "highwayenv_amd", "csrc")
OUT = os.path.join(HERE, "_build")
FILES = ("hwy_kernels.hip", "hwy_engine.hip", "hwy_comm.hip", "hwy_comm.h", "hwy_launch.h", "hwy_params.h", "hwy_wave.h", "hwy_wave2.h",
         "hwy_device.h", "hwy_math.h", "hwy_net.h", "hwy_ix.h")
W, D, NET, IX = "hwy_wave.h", "hwy_device.h", "hwy_net.h", "hwy_ix.h"
W2 = "hwy_wave2.h"


class Stale(Exception):
    pass


def cut(text, start, end, repl=""):
    """Replace text[start_marker : end_marker) by repl."""
    if start not in text or end not in text[text.index(start):]:
        raise Stale(start[:50])
    i = text.index(start)
    j = text.index(end, i)
    return text[:i] + repl + text[j:]


def sub(old, new):
    def f(text):
        if old not in text:
            raise Stale(old[:50])
        return text.replace(old, new)
    return f


def cutter(start, end, repl=""):
    return lambda text: cut(text, start, end, repl)


def ticks(text):
    """s_memtime stamps around the sections of the one-wavefront kernel; totals returned through the obs buffer."""
    t = sub("  int done_flag = p.autoreset ? (int)p.st.done[e] : 0;\n  double env_time",
            "  float n_recount = 0.0f; long long t_prev = clock64(); long long acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0};\n"
            "#define TICK(k) { const long long t_now = clock64(); acc[k] += t_now - t_prev; t_prev = t_now; }\n"
            "  int done_flag = p.autoreset ? (int)p.st.done[e] : 0;\n  double env_time")(text)
    marks = ["    // ---- A. meta-action (abstract.py:294-304",
             "    // ---- C. rank along the road ---",
             "    // lane membership (AbstractLane.on_lane, margin 1) -> bits",
             "    // ---- D. Road.act: lane-change policy (behavior.py:219-263)",
             "    const double delta = sh.delta[i];",
             "    const double self_a = free_self - gap_own;",
             "    // abort rule for ongoing lane changes (behavior.py:229-244): an ordered chain over Road.vehicles -- a changer c",
             "    // ---- E. Road.act: low-level control",
             "    // ---- F. Road.step: integrate",
             "    // ---- G. Road.step: collisions",
             "  }  // frames"]
    t = sub("    wave_update_rank(me.x, active, N, rank, has_tie);\n    // lane membership",
            "    { const int r_before = rank; wave_update_rank(me.x, active, N, rank, has_tie);\n"
            "      n_recount += __ballot(rank != r_before) ? 1.0f : 0.0f; }\n    // lane membership")(t)
    for k, m in enumerate(marks):
        t = sub(m, f"    TICK({k})\n" + m)(t)
    t = sub("  if (q.full_step) observe_wave<true>(q, e, eo, me, true, rank, env_time);\n  me.rank = rank;",
            "  if (q.full_step) observe_wave<true>(q, e, eo, me, true, rank, env_time);\n  TICK(11)\n  me.rank = rank;")(t)
    t = sub("  store_vehicle<1>(q, e, me, false);\n}",
            "  store_vehicle<1>(q, e, me, false);\n"
            "  if (i == 0 && p.obs) { for (int k = 0; k < 12; ++k) p.obs[(size_t)e * p.A * p.V * p.F + k] = (float)acc[k];\n"
            "    p.obs[(size_t)e * p.A * p.V * p.F + 12] = n_recount; }\n}")(t)
    return t


def wide_ticks(text):
    """s_memtime stamps around the sections of the two-vehicles-per-thread kernel (hwy_wave2.h); totals through the obs buffer
    (tools/wide_section_cycles.py)."""
    t = sub("  int done_flag = p.autoreset ? (int)p.st.done[e] : 0;\n  double env_time",
            "  long long t_prev = clock64(); long long acc[11] = {0,0,0,0,0,0,0,0,0,0,0};\n"
            "#define TICK(k) { const long long t_now = clock64(); acc[k] += t_now - t_prev; t_prev = t_now; }\n"
            "  int done_flag = p.autoreset ? (int)p.st.done[e] : 0;\n  double env_time")(text)
    marks = ["    // ---- A. meta-action (abstract.py:294-304 -> controller.py:295-315)",
             "    // ---- C. rank along the road, lane membership masks, frame-start snapshot",
             "    double log_ratio[K];",
             "    // ---- D. Road.act: lane-change policy (behavior.py:219-263)",
             "    if constexpr (K == 2) {  // EnvBlock::idm_free_from_log for both vehicles at once",
             "      if (n_dec) {  // wave-uniform",
             "    // abort rule for ongoing lane changes (behavior.py:229-244): an ordered chain over Road.vehicles",
             "    // ---- E. Road.act: low-level control, F. Road.step: integrate",
             "    // ---- G. Road.step: collisions (road.py:477-481",
             "  }  // frames\n\n  // ---- H. observe"]
    for k, m in enumerate(marks):
        t = sub(m, f"    TICK({k})\n" + m)(t)
    t = sub("    observe_wide<K, true>(q, sh, e, eo, me, true, rank, env_time);\n  }\n",
            "    observe_wide<K, true>(q, sh, e, eo, me, true, rank, env_time);\n  }\n  TICK(10)\n"
            "  if (l == 0 && q.obs) { for (int k = 0; k < 16; ++k) q.obs[(size_t)eo * q.A * q.V * q.F + k] = (float)acc[k]; }\n")(t)
    # collisions split: [11] = publish, [12] = walk trips, [13] = list passes, (G itself: the verdict reads); [14] walk trips, [15] list passes with >= 1 pair, [18] pairs
    t = sub("long long acc[11] = {0,0,0,0,0,0,0,0,0,0,0};", "long long acc[24] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};")(t)
    t = sub("for (int k = 0; k < 16; ++k) q.obs", "for (int k = 0; k < 24; ++k) q.obs")(t)
    t = sub("      const double reach = reach_from_keys(key_d, key_v, p.dt);", "      TICK(11)\n      const double reach = reach_from_keys(key_d, key_v, p.dt);")(t)
    t = sub("        const int count = n_list < PASS ? n_list : PASS;\n",
            "        const int count = n_list < PASS ? n_list : PASS;\n        TICK(12)\n        acc[15] += count > 0; acc[18] += count;\n")(t)
    t = sub("        head = (head + count) & (RING - 1);\n", "        head = (head + count) & (RING - 1);\n        TICK(13)\n")(t)
    t = sub("          k += 2;\n          if (__ballot(going) == 0 || k > N) walking = false;", "          k += 2;\n          acc[14] += 1;\n          if (__ballot(going) == 0 || k > N) walking = false;")(t)
    # abort chain: [16] changers, [17] frames with a chain, [20] walk trips, [21] fixed-point rounds
    t = sub("      if (chain) {  // wave-uniform\n", "      if (chain) {  // wave-uniform\n        acc[16] += wide_popc<K>(cm); acc[17] += 1;\n")(t)
    t = sub("        while (__ballot(any_left) != 0) {  // wave-uniform\n", "        while (__ballot(any_left) != 0) {  // wave-uniform\n          acc[20] += 1;\n")(t)
    t = sub("            if (same) break;", "            acc[21] += 1;\n            if (same) break;")(t)
    return t


def net_ticks(text):
    """s_memtime stamps around the sections of the road-network kernel (hwy_net.h); totals through the obs buffer."""
    t = sub("  Veh me;\n  load_vehicle<1>(p, e, me);\n  const bool present",
            "  long long t_prev = clock64(); long long acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0};\n"
            "#define TICK(k) { const long long t_now = clock64(); acc[k] += t_now - t_prev; t_prev = t_now; }\n"
            "  Veh me;\n  load_vehicle<1>(p, e, me);\n  const bool present")(text)
    marks = ["    // ---- A. meta-actions of all agents",
             "    // ---- B. rank along x, lane membership masks",
             "    // lane_mask[i] = the ranks lane i is searched with",
             "    // ---- C. Road.act ---",
             "    const double delta = me.delta;\n    const double free_self",
             "    // abort rule for ongoing lane changes on the same road",
             "    // ---- D. low-level control",
             "    // ---- E. Road.step: integrate",
             "    {\n      int cl_new, bits_new;",
             "    // ---- F. Road.step: collisions",
             "  }  // frames\n\n  // ---- G. observe / reward / done"]
    for k, m in enumerate(marks):
        t = sub(m, f"    TICK({k})\n" + m)(t)
    # collisions split: walk steps -> acc[10], list passes (+ the verdict reads) -> acc[12]; walk-step / pass / SAT counters
    t = sub("        const int count = n_list < 64 ? n_list : 64, left = n_list - count;  // left < 128\n",
            "        const int count = n_list < 64 ? n_list : 64, left = n_list - count;  // left < 128\n        TICK(10)\n        n_trip += count > 0 ? 1.0f : 0.0f;\n")(t)
    t = sub("        if (i < left) plist[i] = (unsigned short)carry;\n        if (64 + i < left)", "        TICK(12)\n        if (i < left) plist[i] = (unsigned short)carry;\n        if (64 + i < left)")(t)
    t = sub("  }  // frames\n\n  // ---- G. observe", "  TICK(12)\n  }  // frames\n\n  // ---- G. observe")(t)
    t = sub("    TICK(10)\n  TICK(12)\n  }  // frames", "  TICK(12)\n  }  // frames")(t)
    t = sub("          k += WS;\n          if (__ballot(go_b) == 0 || k >= n_present) walking = false;", "          k += WS;\n          n_walk += (float)WS;\n          if (__ballot(go_b) == 0 || k >= n_present) walking = false;")(t)
    t = sub("            r = net_pair_collide(A, Bb, p.dt, &tx, &ty);", "            n_sat += 1.0f;\n            r = net_pair_collide(A, Bb, p.dt, &tx, &ty);")(t)
    t = sub("  long long t_prev = clock64(); long long acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0};", "  float n_walk = 0, n_trip = 0, n_sat = 0; long long t_prev = clock64(); long long acc[13] = {0,0,0,0,0,0,0,0,0,0,0,0,0};")(t)
    t = sub("    if (p.n_frames > 0) me.rank = rank & 0xff;\n    store_vehicle<1>(p, e, me, false);\n  }\n}",
            "    TICK(11)\n    if (p.n_frames > 0) me.rank = rank & 0xff;\n    store_vehicle<1>(p, e, me, false);\n  }\n"
            "  { float sat_any = __ballot(n_sat > 0) ? 1.0f : 0.0f; n_sat = 0; for (int j = 0; j < 64; ++j) n_sat += __shfl(sat_any, j) * 0 ; n_sat = sat_any;\n"
            "  if (i == 0 && p.obs) { for (int k = 0; k < 13; ++k) p.obs[(size_t)e * p.A * p.V * p.F + k] = (float)acc[k];\n"
            "    p.obs[(size_t)e * p.A * p.V * p.F + 13] = n_walk; p.obs[(size_t)e * p.A * p.V * p.F + 14] = n_trip; p.obs[(size_t)e * p.A * p.V * p.F + 15] = n_sat; } }\n}")(t)
    return t


def ix_ticks(text):
    """s_memtime stamps around the sections of the intersection kernel (hwy_ix.h), accumulated in LDS by thread 0 and
    written over the first observation words of the environment (tools/ix_section_dist.py reads them)."""
    t = sub("  int vw[CAP];", "  int vw[CAP];\n  long long tk[16]; long long tprev;")(text)
    t = sub("// ---- lane geometry from the LDS table, per-thread lane index",
            "#define IXTICK(k) { if (threadIdx.x == 0) { const long long t_ = clock64(); sh.tk[k] += t_ - sh.tprev; sh.tprev = t_; } }\n"
            "// ---- lane geometry from the LDS table, per-thread lane index")(t)
    # inside the table walk: [11] = prologue + straight lanes, [12] = exchange + arcs, the final exchange stays in [5]
    t = sub("  ix_group_min(mp, bd, best);  // the arcs are filtered", "  IXTICK(11)\n  ix_group_min(mp, bd, best);  // the arcs are filtered")(t)
    t = sub("  // the straight walk's membership bits and target-lane coordinate join the arcs' per slot\n", "  IXTICK(12)\n")(t)
    # inside the regulation: [13] = samples + circles, the partner loop stays in [3]
    t = sub("      // is_conflict_possible (regulation.py:88-111) + respect_priorities", "      IXTICK(13)\n      // is_conflict_possible (regulation.py:88-111) + respect_priorities")(t)
    t = sub("    // ---- A. meta-action (abstract.py:294-304 -> MDPVehicle.act", "    IXTICK(fr == 0 ? 0 : 6)\n    // ---- A. meta-action (abstract.py:294-304 -> MDPVehicle.act")(t)
    t = sub("    // ---- C. Road.act (road.py:464-467)", "    IXTICK(1)\n    // ---- C. Road.act (road.py:464-467)")(t)
    t = sub("    // ---- D. RegulatedRoad.step (regulation.py:36-68)", "    IXTICK(2)\n    // ---- D. RegulatedRoad.step (regulation.py:36-68)")(t)
    t = sub("    // ---- E. Vehicle.step (kinematics.py:130-177", "    IXTICK(3)\n    // ---- E. Vehicle.step (kinematics.py:130-177")(t)
    t = sub("    HWY_WAVE_LDS_FENCE();  // the trajectories (if any) are dead", "    IXTICK(4)\n    HWY_WAVE_LDS_FENCE();  // the trajectories (if any) are dead")(t)
    t = sub("    // ---- F. collisions (road.py:477-481", "    IXTICK(5)\n    // ---- F. collisions (road.py:477-481")(t)
    t = sub("      if (present && crash) me.flags |= HWY_F_CRASHED;\n    }\n  }\n}",
            "      if (present && crash) me.flags |= HWY_F_CRASHED;\n    }\n  }\n  IXTICK(6)\n}")(t)
    # the shared step body (ix_policy_block): the first occurrence of each pattern after its head; every pattern must be found
    a = t.index("__device__ __forceinline__ void ix_policy_block(")
    head, k = t[:a], t[a:]

    def once(k, old, new):
        if old not in k:
            raise Stale(old)
        return k.replace(old, new, 1)
    k = once(k, "  if (load_table) ix_load_table(ip, sh);", "  if (load_table) ix_load_table(ip, sh);\n  if (i == 0) { for (int q = 0; q < 16; ++q) sh.tk[q] = 0; sh.tprev = clock64(); }\n  //")
    k = once(k, "  if (finalise) ix_spawn_finalise(ip, sh, seed, next_episode, me);\n",
             "  if (finalise) ix_spawn_finalise(ip, sh, seed, next_episode, me);\n  IXTICK(9)\n")
    k = once(k, "    ix_observe(ip, sh, e, me, role == STEP, eo);\n  }\n", "    ix_observe(ip, sh, e, me, role == STEP, eo);\n  }\n  IXTICK(7)\n")
    k = once(k, "ix_clear_spawn(ip, sh, me, seed, p.st.episode[e], step_no);\n", "ix_clear_spawn(ip, sh, me, seed, p.st.episode[e], step_no);\n  IXTICK(8)\n")
    k = once(k, "  if (i == 0) ip.road_steps[e] = road_steps;\n}",
             "  if (i == 0) ip.road_steps[e] = road_steps;\n  IXTICK(10)\n"
             "  if (i == 0 && p.obs) { float *o = p.obs + (size_t)eo * p.A * (p.obs_type == HWY_OBS_KINEMATICS ? p.V * p.F : p.F * p.gW * p.gH);\n"
             "    for (int q = 0; q < 14; ++q) o[q] = (float)sh.tk[q]; o[14] = (float)role; o[15] = (float)n_run;\n"
             "    const unsigned hw_ = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc_ = __builtin_amdgcn_s_getreg((31 << 11) | 20);\n"
             "    o[16] = (float)((((xcc_ & 0xf) * 8 + ((hw_ >> 13) & 7)) * 16 + ((hw_ >> 8) & 0xf)) * 4 + ((hw_ >> 4) & 3)); o[17] = (float)blockIdx.x; }\n}")
    return head + k


def wave_reload(text):
    """Frame loop of the one-wavefront kernel reads its parameters through a per-iteration view of the kernarg segment
    (short-lived SGPRs instead of values kept -- and spilled -- across the loop)."""
    a = text.index("  for (int fr = 0; fr < p.n_frames; ++fr) {\n    wave_turn(turn);")
    b = text.index("  }  // frames", a)
    body = text[a:b]
    head, rest = body.split("{\n", 1)
    rest = re.sub(r"\bp\.", "pl.", rest)
    rest = re.sub(r"\(p, ", "(pl, ", rest)
    return text[:a] + head + "{\n    HWY_RELOAD_PARAMS(pl, p);\n" + rest + text[b:]


NO_LOGEXP = [(D, sub("return r > 0.0 ? log_pos(r) : -__builtin_inf();", "return r;")),
             (D, sub("(1 - exp_bounded(delta * log_ratio))", "(1 - delta * log_ratio)"))]

VARIANTS = {
    # one-wavefront kernel (hwy_wave.h)
    "wbase": [],
    "wnocollide": [(W, cutter("    const Body mine{me.x, me.y, me.v, me.ch, me.sh};", "  }  // frames"))],
    "wnorecount": [(W, sub("  if (recount) {  // wave-uniform", "  if (false) {"))],
    "wnomobil": [(W, sub("    const bool cl = decide && left_ok", "    const bool cl = false && left_ok")),
                 (W, sub("    const bool cr = decide && right_ok", "    const bool cr = false && right_ok"))],
    "wnologexp": NO_LOGEXP,
    "wnosincos": [(W, sub("      sincos_bounded(me.h, &me.sh, &me.ch);\n", "      me.sh = me.h; me.ch = 1 - me.h;\n"))],
    "wnosteer": [(W, sub("    double tb = B::steer_tan_beta(p, me.y, me.h, inv_v, me.tgt);", "    double tb = inv_v * 1e-9;"))],
    "wnoobs": [(W, sub("    observe_wave<true>(q, e, eo, me, true, rank);\n", ""))],
    "wticks": [(W, ticks)],
    "w2ticks": [(W2, wide_ticks)],
    "wreload": [(W, wave_reload)],
    # road-network kernel (hwy_net.h)
    "nticks": [(NET, net_ticks)],
    # the auto-reset wavefronts of the merge step kernel stamp their lifetime (clock64 ticks) into observation word 0, -1 into word 15
    "nresetticks": [(NET, net_ticks),
                    (NET, sub("  if (p.autoreset && p.st.done[e]) {\n    Veh me;\n    const uint32_t episode = p.st.episode[e] + 1u;\n    net_spawn_env(",
                              "  if (p.autoreset && p.st.done[e]) {\n    const long long rt0 = clock64();\n    Veh me;\n    const uint32_t episode = p.st.episode[e] + 1u;\n    net_spawn_env(")),
                    (NET, sub("      p.terminated[e] = 0;\n      p.truncated[e] = 0;\n    }\n    return;\n  }\n\n  WaveTurn turn;",
                              "      p.terminated[e] = 0;\n      p.truncated[e] = 0;\n    }\n"
                              "    if (i == 0 && p.obs) { float *o = p.obs + (size_t)e * p.A * p.V * p.F; o[0] = (float)(clock64() - rt0); o[15] = -1.0f; }\n"
                              "    return;\n  }\n\n  WaveTurn turn;"))],
    "nnocoll": [(NET, sub("      for (int k = 1; k < n_present; ++k) {\n        const int ra = rank - k, rb = rank + k;", "      for (int k = 1; k < 1; ++k) {\n        const int ra = rank - k, rb = rank + k;"))],
    "nnosat": [(NET, sub("          r = net_pair_collide(A, Bb, p.dt, &tx, &ty);", "          r = 0;"))],
    # (timing only) the in-loop table walk without its closest-lane half / without the whole walk's arithmetic
    "nnoclosest": [(NET, sub("      net_lane_pass<true>(np, sh, sine_mask, present, me.x, me.y, me.h, &bits_new, &cl_new);",
                             "      net_lane_pass<false>(np, sh, sine_mask, present, me.x, me.y, me.h, &bits_new, &cl_new); cl_new = me.lane;"))],
    # wave timeline: every wavefront of hwy_step_wave_kernel writes its start / end s_memrealtime (100 MHz) and HW_ID /
    # XCC_ID over the first four observation words of its environment (tools/wave_timeline.py reads them)
    "wtimeline": [(W, sub("  typedef EnvBlock<1> B;\n  const int i = threadIdx.x;\n  const int N = p.N;",
                          "  typedef EnvBlock<1> B;\n  const int i = threadIdx.x;\n"
                          "  const unsigned long long tl_t0 = wall_clock64();\n"
                          "  const unsigned tl_hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), tl_xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);\n"
                          "  const int N = p.N;")),
                  (W, sub("  store_vehicle<1>(q, e, me, false);\n}",
                          "  store_vehicle<1>(q, e, me, false);\n"
                          "  if (q.obs && i == 0) {\n"
                          "    __builtin_amdgcn_s_waitcnt(0);\n"
                          "    const unsigned long long tl_t1 = wall_clock64();\n"
                          "    unsigned *o = (unsigned *)(q.obs + (size_t)e * q.A * q.V * q.F);\n"
                          "    o[0] = (unsigned)tl_t0; o[1] = (unsigned)tl_t1; o[2] = tl_hw; o[3] = tl_xcc; o[4] = 0u;\n"
                          "  }\n}")),
                  # event counters (LDS atomics): SAT trips, near pairs, chain links (all / with a rival), follower-test trips, recounts
                  (W, sub("struct WaveShared {", "struct WaveShared {\n  int cnt[8];")),
                  (W, sub("  WaveTurn turn;\n  wave_turn_init(turn, p.prio_shift, p.prio_recip);",
                          "  if (i < 8) sh.cnt[i] = 0;\n  WaveTurn turn;\n  wave_turn_init(turn, p.prio_shift, p.prio_recip);")),
                  (W, sub("              r = pair_collide(A, Bb, p.dt, &tx, &ty);", "              atomicAdd(&sh.cnt[0], 1); r = pair_collide(A, Bb, p.dt, &tx, &ty);")),
                  (W, sub("            if (!surely_apart(A, Bb, p.dt)) {", "            atomicAdd(&sh.cnt[1], 1);\n            if (!surely_apart(A, Bb, p.dt)) {")),
                  # (round 5: the rank-space chain -- cnt[2] = frames with a chain, cnt[3] = walk trips of it)
                  (W, sub("        int *const sbits = reinterpret_cast<int *>(sh.nx);  // (the post-integration bodies only live inside section G)",
                          "        if (i == 0) atomicAdd(&sh.cnt[2], 1);\n        int *const sbits = reinterpret_cast<int *>(sh.nx);")),
                  (W, sub("          const bool go = rem != 0;\n          const int rr = go ? ctz64(rem) : 0;", "          if (i == 0) atomicAdd(&sh.cnt[3], 1);\n          const bool go = rem != 0;\n          const int rr = go ? ctz64(rem) : 0;")),
                  (W, sub("        const bool pend = pend_l || pend_r;", "        if (i == 0) atomicAdd(&sh.cnt[4], 1);\n        const bool pend = pend_l || pend_r;")),
                  (W, sub("o[0] = (unsigned)tl_t0; o[1] = (unsigned)tl_t1; o[2] = tl_hw; o[3] = tl_xcc; o[4] = 0u;",
                          "o[0] = (unsigned)tl_t0; o[1] = (unsigned)tl_t1; o[2] = tl_hw; o[3] = tl_xcc; o[4] = 0u; for (int k = 0; k < 5; ++k) o[5 + k] = (unsigned)sh.cnt[k];")),
                  (W, sub("      p.terminated[eo] = 0;\n      p.truncated[eo] = 0;\n    }\n    return;",
                          "      p.terminated[eo] = 0;\n      p.truncated[eo] = 0;\n    }\n"
                          "    if (p.obs && i == 0) {\n"
                          "      __builtin_amdgcn_s_waitcnt(0);\n"
                          "      const unsigned long long tl_t1 = wall_clock64();\n"
                          "      unsigned *o = (unsigned *)(p.obs + (size_t)e * p.A * p.V * p.F);\n"
                          "      o[0] = (unsigned)tl_t0; o[1] = (unsigned)tl_t1; o[2] = tl_hw; o[3] = tl_xcc; o[4] = 1u;\n"
                          "    }\n"
                          "    return;"))],
    # intersection kernel (hwy_ix.h): sections removed (timing only)
    "ixbase": [],
    # (+ the step workgroups honour hwy_set_block_order: tools/ix_placement_probe.py)
    "ixticks": [(IX, ix_ticks),
                (IX, sub("  ix_policy_block<CAP, NT>(ip, sh, (int)blockIdx.x, (int)blockIdx.x);\n}\n\n// hwy_rollout_device on the intersection",
                         "  const int b_ = (int)blockIdx.x, bx_ = (ip.s.block_env && b_ < ip.num_envs) ? (int)ip.s.block_env[b_] : b_;\n"
                         "  ix_policy_block<CAP, NT>(ip, sh, bx_, bx_);\n}\n\n// hwy_rollout_device on the intersection")),
                ("hwy_engine.hip", sub("  if (eng->cfg.scenario != HWY_SCENARIO_HIGHWAY || eng->cfg.num_vehicles > 64 || eng->force_block_kernel || E > 65535)",
                                       "  if (!is_ix(eng) && (eng->cfg.scenario != HWY_SCENARIO_HIGHWAY || eng->cfg.num_vehicles > 64 || eng->force_block_kernel || E > 65535))"))],
    "ixnoreg": [(IX, sub("    if (road_steps % every == 0) {  // wave-uniform", "    if (false) {"))],
    "ixnocoll": [(IX, sub("      for (u64 m = NH > 1 ? ((pm | (pm >> 1)) & 0x5555555555555555ull) : pm; m; m &= m - 1) {  // wave-uniform, ascending", "      for (u64 m = 0; m; m &= m - 1) {"))],
    "ixnoarc": [(IX, sub("    const bool need = mine && present && (fabs(lat) <= sh.wid[L] / 2 + 1.0 || !(fabs(lat) > bd) || L == tgt);", "    const bool need = false;"))],
    "ixnostraight": [(IX, sub("  for (int k = 0; k < ns; k += NH) {  // wave-uniform trip, one lane per half", "  for (int k = 0; k < 0; k += NH) {"))],
    "ixnomask": [(IX, sub("    for (int L = 0; L < ip.n_lanes; ++L) {\n      const u64 b = __ballot(present && ((bits >> L) & 1));", "    for (int L = 0; L < 0; ++L) {\n      const u64 b = __ballot(present && ((bits >> L) & 1));"))],
    "ixnoobs": [(IX, sub("    ix_observe(ip, sh, e, me, true);\n", "    ;\n"))],
    "ixnospawn": [(IX, sub("    if (!ip.host_spawn) ix_clear_spawn(", "    if (false) ix_clear_spawn("))],
    "ixnointeg": [(IX, sub("      sincos_bounded(me.h, &me.sh, &me.ch);\n    }\n    __syncthreads();  // the trajectories", "    }\n    __syncthreads();  // the trajectories"))],
    "ixnoreset": [(IX, sub("  if (p.autoreset && p.st.done[e]) {  // the step after", "  if (false) {  // the step after"))],
    "ixnoact": [(IX, sub("    if (acts) {\n      // follow_road", "    if (false) {\n      // follow_road"))],
    # (A/B, a VALID simulation) the CircularLane coordinate divided by the radius with IEEE divisions again instead of multiplied
    # by the rounded reciprocal (IxSharedT::irad, round 4)
    "ixdiv": [(IX, sub("(sh.ldir[L] * s) * sh.irad[L] + sh.sph[L]", "sh.ldir[L] * s / sh.rad[L] + sh.sph[L]")),
              (IX, sub("((r.d * sa) * sh.irad[r.L] + r.g)", "(r.d * sa / r.c + r.g)"))],
    # generic workgroup kernel (hwy_device.h)
    "base": [],
    "nocollide": [(D, sub("    if (all_check) {\n      // Full pairwise (highway-v0).  A pair can only collide", "    if (false) {\n      // Full pairwise (highway-v0).  A pair can only collide"))],
    "nowalk": [(D, sub("      bool go = active, walking = true, any_impact = false;", "      bool go = false, walking = true, any_impact = false;"))],
    "nomobil": [(D, sub("    if (decide) {\n      me.timer = 0.0;", "    if (false) {\n      me.timer = 0.0;"))],
    "norankcheck": [(D, sub("    if (__syncthreads_or(stale)) {  // block-uniform", "    __syncthreads(); if (false) {"))],
    "nologexp": NO_LOGEXP,
    "nosincos": [(D, sub("      sincos_bounded(me.h, &me.sh, &me.ch);\n", "      me.sh = me.h; me.ch = 1 - me.h;\n"))],
    "nosteer": [(D, sub("      tb = B::steer_tan_beta(p, me.y, me.h, inv_v, me.tgt);", "      tb = inv_v * 1e-9;"))],
    "nochain": [(D, cutter("    // abort rule for ongoing lane changes (behavior.py:229-244): an ordered chain (Gauss-Seidel", "    wave_turn(turn);\n    // ---- E. Road.act: low-level control"))],
    "nomobilb": [(D, sub("      const u64 dm = __ballot(cl || cr);", "      const u64 dm = 0;"))],
    "nosatb": [(D, sub("              r = hwy::pair_collide(A, Bb, p.dt, &tx, &ty);", "              r = 0;"))],
    "nolistb": [(D, sub("          const bool cand = pair >= 0 && !hwy::surely_apart(A, Bb, p.dt);", "          const bool cand = false;"))],
}


# compiler-flag experiments on the unmodified sources
FLAG_VARIANTS = {
    # (the stamped road-network kernel with the SAT's register-allocation constraints crashes this toolchain's register allocator --
    #  hwy_device.h: sat_axis<SETTLE> has the same story for the two-wavefront workgroup kernel; the stamps only time sections)
    "nticks": ["-DHWY_SAT_FENCE()=((void)0)", "-DHWY_SAT_SETTLE(f)=((void)0)"],
    "nresetticks": ["-DHWY_SAT_FENCE()=((void)0)", "-DHWY_SAT_SETTLE(f)=((void)0)"],
    "f_base": [],
    "f_noslp": ["-fno-slp-vectorize"],
    "f_nounroll": ["-fno-unroll-loops"],
    "f_misched": ["-mllvm", "-amdgpu-schedule-metric-bias=0"],
    "f_os": ["-Os"],
    "f_o2": ["-O2"],
    "f_sgprfirst": ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=true"],
    "f_licm": ["-mllvm", "-disable-licm-promotion"],
    "f_mlicm": [],   # WITH MachineLICM (the build flags of highwayenv_amd/build.py minus -disable-machine-licm)
    "f_sink": ["-mllvm", "-sink-insts-to-avoid-spills"],
    "f_maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "f_iterilp": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    "f_memclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "f_bias0": ["-mllvm", "-amdgpu-schedule-metric-bias=0"],
    "f_bias100": ["-mllvm", "-amdgpu-schedule-metric-bias=100"],
    "f_nopostlicm": ["-mllvm", "-disable-postra-machine-licm"],
    "f_fma": ["-ffp-contract=fast"],
    "f_fast": ["-ffp-contract=fast"],   # (the build contracts by source expression, -ffp-contract=on: this fuses whatever the optimiser finds)
    "f_nofma": ["-ffp-contract=off"],   # (since round 4 the build contracts: this is the round-3 arithmetic)
    "f_kcmix": ["-include", os.path.join(os.path.dirname(os.path.abspath(__file__)), "kc_mix.h")],
    "f_fma_kcmix": ["-ffp-contract=fast", "-include", os.path.join(os.path.dirname(os.path.abspath(__file__)), "kc_mix.h")],  # a*b+c fused where the compiler sees it (the build double-rounds like numpy: -ffp-contract=off)
}


def build(name):
    src = {f: open(os.path.join(CSRC, f)).read() for f in FILES}
    try:
        for f, fn in VARIANTS.get(name, []):
            src[f] = fn(src[f])
    except Stale as ex:
        return f"[skip] {name}: pattern not found: {ex}"
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    for f, text in src.items():
        open(os.path.join(d, f), "w").write(text.replace('#include "../../include/hwy_engine.h"',
                                                         f'#include "{ROOT}/include/hwy_engine.h"'))
    lib = os.path.join(OUT, f"libhwy_engine_{name}.so")
    flags = list(HIPCC_FLAGS)
    if name == "f_mlicm":
        k = flags.index("-disable-machine-licm")
        del flags[k - 1:k + 1]  # ("-mllvm", "-disable-machine-licm")
    flags = flags + FLAG_VARIANTS.get(name, [])  # (a later -amdgpu-sched-strategy= overrides the build's)
    r = subprocess.run(["hipcc", *flags, "-shared", "-o", lib, os.path.join(d, "hwy_kernels.hip"),
                        os.path.join(d, "hwy_engine.hip"), os.path.join(d, "hwy_comm.hip"), "-ldl"], capture_output=True, text=True)
    shutil.rmtree(d)
    return lib if r.returncode == 0 else f"[fail] {name}: {r.stderr[-400:]}"


if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS)
    assert all(n in VARIANTS or n in FLAG_VARIANTS for n in names), "unknown variant"
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(6) as ex:
        for res in ex.map(build, names):
            print(res)

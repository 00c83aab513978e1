// hwy_device.h -- device code of the MI355X (gfx950) batched HighwayEnv step engine.
//
// One workgroup == one environment, thread i == vehicle i (index in the
// reference's Road.vehicles list), NW = ceil(N/64) wavefronts per workgroup
// (one 64-wide wavefront for the headline 51-vehicle case).  The whole policy
// step -- T simulation frames of {meta-action, Road.act, Road.step}, then
// KinematicObservation + reward + termination -- is ONE launch: the
// environment's struct-of-arrays is read from HBM once, lives in registers
// and LDS for all frames, and is written back once.
//
// What replaces the reference's O(N) Python scans (road/road.py:483-547,
// 56% of its run time):
//   * each frame every thread computes its RANK by longitudinal coordinate
//     (one LDS-broadcast pass, 3 VALU ops per candidate);
//   * one wave ballot per lane builds a 64-bit membership mask in rank space
//     (bit r == "the r-th vehicle along the road is geometrically on lane L",
//     AbstractLane.on_lane with margin 1, road/lane.py:80-102);
//   * a front/rear neighbour query is then two bit-scans on that mask
//     (s_ff1 / s_flbit) + one LDS gather instead of a scan over N vehicles.
//   Exact-tie semantics of the reference loop (front: last wins, rear: first
//   wins) are preserved by a workgroup-uniform fallback to the literal scan
//   whenever two vehicles share the same x.
// Sequential semantics hidden in the reference's Python loops (SURVEY.md 7,
// "hard parts") are honoured explicitly: the Gauss-Seidel read of neighbours'
// target lanes in the lane-change abort rule (behavior.py:229-244) is an
// ordered chain over the (rare) lane-changing vehicles; "last pair in loop
// order wins" for collision impacts (road.py:477-481, objects.py:103-113) is
// "partner with the highest index wins".
//
// All arithmetic is IEEE f64 like the reference (objects.py:43), compiled
// with -ffp-contract=off (no silent FMA contraction).  The reference pays
// 12 libm calls per vehicle-frame (asin x2, tan x2, atan x2, cos x2, sin x3,
// pow); here the steering -> slip -> bicycle chain is folded with exact
// trigonometric identities (tan(asin w) = w/sqrt(1-w^2), tan(atan u) = u,
// cos/sin(atan t) = (1,t)/sqrt(1+t^2), angle addition against the cached
// cos/sin of the heading) so that only asin, sincos, log and exp remain (hwy_math.h) -- each
// folded expression agrees with the literal one to a few ulp, the same
// order as ocml-vs-numpy libm differences, and is checked against the
// literal C oracle at 1e-9 per frame.
//
// This header has no #include of the HIP runtime on purpose: the product
// translation unit (hwy_kernels.hip) includes <hip/hip_runtime.h> first; the
// CPU emulation harness under tests/emu (test infrastructure, never shipped)
// includes its own shim first.
#pragma once

#include <stdint.h>

#include "../../include/hwy_engine.h"
#include "hwy_math.h"

namespace hwy {

typedef unsigned long long u64;

// ---- class constants of the reference ------------------------------------------------
// vehicle/kinematics.py:21-30
#define HWY_VEH_LENGTH 5.0
#define HWY_VEH_WIDTH 2.0
#define HWY_MAX_SPEED 40.0
#define HWY_MIN_SPEED (-40.0)
// vehicle/controller.py:24-33
#define HWY_KP_A (1.0 / 0.6)
#define HWY_KP_HEADING (1.0 / 0.2)
#define HWY_KP_LATERAL (1.0 / 0.6)
#define HWY_PI 3.141592653589793
#define HWY_MAX_STEER (HWY_PI / 3.0)
// vehicle/behavior.py:21-46
#define HWY_ACC_MAX 6.0
#define HWY_COMFORT_ACC_MAX 3.0
#define HWY_COMFORT_ACC_MIN (-5.0)
#define HWY_DISTANCE_WANTED (5.0 + HWY_VEH_LENGTH)
#define HWY_TIME_WANTED 1.5
#define HWY_LC_MIN_ACC_GAIN 0.2
#define HWY_LC_MAX_BRAKING 2.0
#define HWY_LC_DELAY 1.0

// packed per-vehicle word: lane[0:3] | target_lane[4:7] | speed_index[8:11] | flags[12:15] | rank[16:23] | flags[24:25]
// (flag bits 4-5 -- HWY_F_OBSTACLE, HWY_F_ABSENT, road-network scenarios only -- live in bits 24-25)
// rank = position along the road (a HINT carried from step to step: the one-wavefront kernel verifies it
// every frame and recounts when it is stale; hwy_set_state / spawn write the identity permutation)
__host__ __device__ inline int32_t pack_word(int lane, int tgt, int sidx, int flags, int rank) {
  return (lane & 0xf) | ((tgt & 0xf) << 4) | ((sidx & 0xf) << 8) | ((flags & 0xf) << 12) | ((rank & 0xff) << 16) |
         (((flags >> 4) & 0x3) << 24);
}
__host__ __device__ inline int word_lane(int32_t w) { return w & 0xf; }
__host__ __device__ inline int word_target(int32_t w) { return (w >> 4) & 0xf; }
__host__ __device__ inline int word_speed_index(int32_t w) { return (w >> 8) & 0xf; }
__host__ __device__ inline int word_flags(int32_t w) { return ((w >> 12) & 0xf) | (((w >> 24) & 0x3) << 4); }
__host__ __device__ inline int word_rank(int32_t w) { return (w >> 16) & 0xff; }

struct DevState {
  double *x, *y, *heading, *speed, *timer, *target_speed, *delta, *impact_x, *impact_y;  // [E][pitch]
  int32_t *packed;                                                                       // [E][pitch]
  double *time;       // [E]
  uint8_t *done;      // [E] terminated|truncated of the previous step (auto-reset)
  uint32_t *episode;  // [E] episode counter (RNG stream selector)
};

struct ResetParams {
  double ego_spacing;     // config["ego_spacing"]
  double other_spacing;   // 1 / config["vehicles_density"]
  double lane_factor;     // exp(-5/40 * lanes_count), kinematics.py:92-96 (host libm)
  int32_t initial_lane_id;  // -1: random
  int32_t fast;             // HighwayEnvFast: only controlled vehicles check collisions
  uint64_t base_seed;
};

struct StepParams {
  // compact config (hwy_config subset)
  int32_t N, A, L, T, flags, V, F, n_ts, pitch;
  int32_t action_set;  // hwy_config.action_set: the table the action ids index (HWY_ACTION_TO_ALL)
  int32_t agent_index[HWY_MAX_AGENTS];
  int32_t feat[HWY_MAX_FEATURES];
  double target_speeds[HWY_MAX_TARGET_SPEEDS];
  double dt, policy_dt, duration, lane_width, road_length, speed_limit;
  double collision_reward, right_lane_reward, high_speed_reward, rs0, rs1, perception;
  double rx0, rx1, ry0, ry1, rvx0, rvx1, rvy0, rvy1;
  // reciprocals of the constant denominators of utils.lmap (utils.py:31-33), computed once on the host in f64: the
  // one-wavefront kernel multiplies by them instead of running an IEEE f64 division (~30 instructions) per observed
  // feature and per reward term -- at most 1 ulp (f64) away from the quotient, far below the f32 observation's own
  // rounding and the 1e-9 reward tolerance
  double inv_lane_width;
  double inv_rx, inv_ry, inv_rvx, inv_rvy;  // reciprocals of the observation feature ranges (f32 outputs only)
  // OccupancyGridObservation (obs_type == HWY_OBS_OCCUPANCY_GRID)
  int32_t obs_type, gW, gH, g_nwp;  // grid shape; waypoints per lane of the on-road layer
  int32_t obs_std5;  // Kinematics with features == [presence, x, y, vx, vy] (the default): observe_wave's straight-line path
  double gmin_x, gmin_y, gstep_x, gstep_y, g_spacing;
  int32_t *grid_ws;  // [E][A][2][W*H] cell owner (lowest vehicle index) / on-road flag
  DevState st;
  // per-call
  int32_t k_steps;        // hwy_rollout_device on the one-wavefront kernel: policy steps per launch (outputs / actions of step k in
                          // rows k * num_envs + e); 0 / 1 elsewhere
  int32_t num_envs;
  int32_t n_frames;       // frames to simulate (T for a policy step)
  int32_t full_step;      // 1: advance time, observe, reward, done flags; 0: frames only
  int32_t autoreset;      // 1: envs with done[e] are re-spawned instead of stepped
  int32_t prio_shift;     // > 0: issue-priority rotation among the wavefronts of a SIMD (hwy_wave.h: WaveTurn); 0: off
  uint32_t prio_recip;    // != 0: a turn lasts prio_shift x 64 clock ticks instead of 2^prio_shift; floor(2^32 / prio_shift)
  const int32_t *actions;  // [E][A] or nullptr
  float *obs;              // [E][A][V][F] or nullptr
  double *reward;          // [E][A]
  uint8_t *terminated, *truncated;  // [E]
  double *info_speed;      // [E][A] or nullptr
  uint8_t *info_crashed;   // [E][A] or nullptr
  const uint8_t *reset_mask;  // reset kernel only: [E] or nullptr (= all)
  const uint64_t *reset_seeds;  // reset kernel only: [E] or nullptr (= base_seed + e)
  const uint16_t *block_env;    // one-wavefront step kernel: environment of workgroup b (hwy_set_block_order), or nullptr (= b)
  unsigned long long *counters; // [HWY_CTR_COUNT] event counters of the engine (hwy_get_counters), nullptr = not counted
  ResetParams rp;
};

// A second view `q` of the by-value kernel argument `p` through a pointer the compiler cannot see through: the fields are
// loaded (scalar loads from the kernel-argument segment) where they are used instead of staying live -- as SGPRs spilled to
// VGPR lanes -- across long code.  The kernel must have StepParams as its only argument (offset 0 of the segment).
#ifndef HWY_RELOAD_STEP_PARAMS
#define HWY_RELOAD_STEP_PARAMS(q, p)                                  \
  auto kernarg_##q = __builtin_amdgcn_kernarg_segment_ptr();          \
  asm volatile("" : "+s"(kernarg_##q));                             \
  const StepParams &q = *(const StepParams *)kernarg_##q
#endif

// ---- utils.py ---------------------------------------------------------------------------
// utils.py:50-56: x if |x| > eps else (eps if x >= 0 else -eps).  For every x but NaN that is max(|x|, eps) carrying the sign
// of (x < 0) -- -0.0 counts as >= 0, like in the reference: a max, a compare and a sign flip instead of two compares and two
// 64-bit selects.
__device__ inline double not_zero(double x) {
  const double t = fmax(fabs(x), 1e-2);
  return x < 0 ? -t : t;
}
__device__ inline double abs_not_zero(double x) { return fmax(fabs(x), 1e-2); }  // == fabs(not_zero(x))
// utils.py:59-60: ((x + pi) % (2 pi)) - pi with Python's floor-mod (hwy_math.h: py_mod_pos)
__device__ inline double wrap_to_pi(double x) { return py_mod_pos(x + HWY_PI, 2 * HWY_PI) - HWY_PI; }
__device__ inline double clipd(double a, double lo, double hi) { return fmin(fmax(a, lo), hi); }
// utils.py:31-33
__device__ inline double lmap(double v, double x0, double x1, double y0, double y1) {
  return y0 + (v - x0) * (y1 - y0) / (x1 - x0);
}
// the same with the host-computed reciprocal of (x1 - x0) (StepParams::inv_*)
__device__ inline double lmap_inv(double v, double x0, double inv_dx, double y0, double y1) {
  return y0 + ((v - x0) * (y1 - y0)) * inv_dx;
}

// ---- counter-based RNG for the device-side reset: Philox-4x32-10 --------------------------
// (Salmon et al., SC'11.)  NOT numpy's PCG64 stream: see hwy_reset in hwy_engine.h.
__host__ __device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// two uniforms in [0,1) with 53 random bits each from one Philox block
__host__ __device__ inline void philox_uniform2(uint64_t seed, uint32_t vehicle, uint32_t episode,
                                                uint32_t draw, double *u0, double *u1) {
  uint32_t c[4] = {vehicle, episode, draw, 0x48575931u /* "HWY1" */};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint64_t a = ((uint64_t)c[0] << 32) | c[1], b = ((uint64_t)c[2] << 32) | c[3];
  *u0 = (double)(a >> 11) * (1.0 / 9007199254740992.0);
  *u1 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
}

// ---- small bit helpers ----------------------------------------------------------------------
__device__ inline int ctz64(u64 m) { return __ffsll((long long)m) - 1; }       // m != 0
__device__ inline int msb64(u64 m) { return 63 - __clzll((long long)m); }      // m != 0

// ---- one rectangle (vehicle) as seen by the collision code ---------------------------------------
struct Body {
  double x, y, v, c, s;  // position, speed, cos/sin heading
};
// field-wise select (keeps both bodies in registers; a reference to one of two structs would force them to memory)
__device__ inline Body select_body(bool first, const Body &p, const Body &q) {
  return Body{first ? p.x : q.x, first ? p.y : q.y, first ? p.v : q.v, first ? p.c : q.c, first ? p.s : q.s};
}

// ---- provable non-collision, without running the SAT ---------------------------------------------
// For a unit axis n, the projection of a rectangle is (centre . n) -+ r with
//   r = L/2 |u.n| + W/2 |w.n|   (u, w = the rectangle's own unit axes),
// so on that axis the reference's interval_distance is D - r_a - r_b (static) and at least
// D - r_a - r_b - |n.(disp_a - disp_b)| (swept).  If that lower bound exceeds `margin` on ANY of the
// two body axes of a (which are, up to 1e-16, SAT normals of utils.py:216-217), the reference's
// `distance > 0` fires on that axis in both tests => (intersecting, will_intersect) = (False, False)
// -- the floating-point noise of its corner projections is ~1e-13, far below the 1e-6 margin.
// Cars alongside in the next lane (2 m clear) and cars queued in lane are dismissed here; only
// pairs that really touch (or come within a micrometre of it) run the literal SAT below.
__device__ inline bool surely_apart(const Body &A, const Body &B, double dt) {
  const double ca = A.c, sa = A.s, cb = B.c, sb = B.s;
  const double dx = B.x - A.x, dy = B.y - A.y;
  const double cr = fabs(ca * cb + sa * sb), sr = fabs(sb * ca - cb * sa);  // |cos|, |sin| of (h_b - h_a)
  const double rvx = (A.v * ca - B.v * cb) * dt, rvy = (A.v * sa - B.v * sb) * dt;
  const double margin = 1e-6;
  // a's lateral axis (-sin h_a, cos h_a)
  const double gap_lat = fabs(-sa * dx + ca * dy) - HWY_VEH_WIDTH / 2 - (HWY_VEH_LENGTH / 2 * sr + HWY_VEH_WIDTH / 2 * cr) -
                         fabs(-sa * rvx + ca * rvy);
  // a's longitudinal axis (cos h_a, sin h_a)
  const double gap_lon = fabs(ca * dx + sa * dy) - HWY_VEH_LENGTH / 2 - (HWY_VEH_LENGTH / 2 * cr + HWY_VEH_WIDTH / 2 * sr) -
                         fabs(ca * rvx + sa * rvy);
  return gap_lat > margin || gap_lon > margin;
}

// ---- how far the forward walk of the full pairwise collision check has to look ------------------------------------------------
// The walk visits partners in the order of the FRAME-START x and stops at the first one further than `reach` ahead.  A pair (i, q),
// x0_i <= x0_q, can only pass the reference's pre-check sphere (objects.py:124-127) if x_q - x_i <= 5.5 + max(|v_i|, |v_q|) dt after
// the integration, and x_q - x_i >= (x0_q - x0_i) - 2 D with D = the largest |x - x0| of the environment in this frame (impact
// displacements included).  So reach = 5.5 + S dt + 2 D, S = the largest |v| after the integration, misses nothing -- with the
// ACTUAL maxima of the frame, not the 50 m/s + 3 m constants of rounds 2-5 (21.5 m: three to four partners per vehicle in dense
// traffic where 11 m -- one or two -- suffice) and without their fallback to the all-pairs loop for faster bodies.  The maxima are
// taken over the HIGH WORDS of the doubles (monotone in |x|; wave_max_u32 below) and rounded up: (key + 1, 0) is above
// every double with that high word.  A non-finite operand gives an infinite reach (the literal all-pairs loop).  A filter only:
// which pairs are FOUND does not depend on it (1e-6 of slack against the rounding of the sums), so no result does.
// Maximum of a 32-bit value over the 64 lanes of the wavefront (call it with ALL lanes enabled, 0 from lanes that have nothing to
// say), wave-uniform result.  Six v_max_u32 with DPP operands -- a scan within each row of 16 lanes, then across the rows
// (row_bcast15 / row_bcast31: GFX9) -- and a v_readlane: the obvious alternative, one ds_max_u32 per lane on a shared LDS word,
// serialises its 64 lanes on ONE address in a pipeline the whole CU shares (measured: highway-v0 109 -> 160 us).
// LDS written and read by ONE wavefront of a multi-wavefront workgroup: its LDS instructions execute in order, only the compiler
// must keep them so (the CPU emulation of tests/emu, whose threads are fibers, needs a rendezvous of that wavefront's fibers).
#ifndef HWY_WAVEFRONT_FENCE
#define HWY_WAVEFRONT_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
#endif
#ifndef HWY_WAVE_MAX_U32
__device__ inline unsigned wave_max_u32(unsigned v) {
  int x = (int)v;
#define HWY_DPP_MAX_(ctrl, rows) \
  { const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp(0, x, ctrl, rows, 0xf, false); x = (int)(o_ > (unsigned)x ? o_ : (unsigned)x); }
  HWY_DPP_MAX_(0x111, 0xf)  // row_shr:1
  HWY_DPP_MAX_(0x112, 0xf)  // row_shr:2
  HWY_DPP_MAX_(0x114, 0xf)  // row_shr:4
  HWY_DPP_MAX_(0x118, 0xf)  // row_shr:8  -> lane 15 of every row holds the row's maximum
  HWY_DPP_MAX_(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  HWY_DPP_MAX_(0x143, 0xc)  // row_bcast:31 into rows 2 and 3  -> lane 63 holds the wavefront's
#undef HWY_DPP_MAX_
  return (unsigned)__builtin_amdgcn_readlane(x, 63);
}
#define HWY_WAVE_MAX_U32(v) ::hwy::wave_max_u32(v)
#endif
__device__ inline unsigned reach_key(double x) { return (unsigned)__double2hiint(x) & 0x7fffffffu; }
__device__ inline double reach_from_keys(unsigned key_d, unsigned key_v, double dt) {
  if (key_d >= 0x7fe00000u || key_v >= 0x7fe00000u) return __builtin_inf();
  const double D = __hiloint2double((int)(key_d + 1u), 0), S = __hiloint2double((int)(key_v + 1u), 0);
  return ((5.5 + S * dt) + 2.0 * D) + 1e-6;
}

// ---- rectangle SAT with swept extension (utils.py:196-241, objects.py:122-138,169-181) -----------
// a = lower-index vehicle (the reference's `self`), b = the other.  Returns bit0 intersecting,
// bit1 will_intersect; translation in (*tx,*ty) when will_intersect.
//
// The reference projects the 4+4 corners on the 8 edge normals.  The normals of a rectangle are +-its two body
// axes, in the order -u, +w, +u, -w for the corner order of objects.py:169-181, and a projection interval is
// (centre . n) -+ (L/2 |u.n| + W/2 |w.n|): four axis DIRECTIONS are evaluated, in ~100 flops and a dozen registers
// instead of a 16-double corner table (differences to the literal corner arithmetic: 1e-16).  The two flags are
// symmetric under n -> -n, but interval_distance (utils.py:188-193) is NOT when the swept interval of `a` CONTAINS
// b's: with a = [a0, a1], b = [b0, b1] it returns b0 - a1 if a0 < b0 else a0 - b1, and for -n the test becomes
// a1 > b1, which picks the OTHER candidate exactly in the containment case (two cars nose to tail on one lane centre,
// lateral axis: found by tests/test_collision_steps.py against the reference's traces).  So every direction yields
// both signed distances and the minimum is folded in the reference's own order of the 8 normals (first minimum wins).
// The four axis directions one after the other: the ILP-first scheduler otherwise interleaves them, and the SAT -- a rare path -- then
// holds ~100 VGPRs at once, which is what sets the register count (and, under a 128-register budget, the spills) of every kernel it
// is inlined into.  A scheduling fence between two directions: same instructions, same results, a third of the registers.
#ifndef HWY_SAT_FENCE
#define HWY_SAT_FENCE() __builtin_amdgcn_sched_barrier(0)
#define HWY_SAT_SETTLE(f) asm volatile("" : "+v"(f))
#endif
struct SatAcc {
  int flags;  // bit 0: intersecting, bit 1: will_intersect (both start set; an axis that separates clears its bit)
  double min_distance, axx, axy;
  int order;  // position of the winning normal in the reference's loop over the 8 normals
};
// One axis DIRECTION n of polygon `poly` (0 = a, 1 = b): both of its normals (+n and -n) are folded into the running
// minimum at once.  `first_minus`: the reference meets -n first (the body axis u: order -u .. +u) or +n first (w: +w .. -w);
// k0 / k1 = positions of the two normals in the reference's loop, so "first minimum wins" becomes "smaller |d|, then
// smaller position" and no per-axis result has to stay live until its turn.
// SETTLE = false: without the register-allocation constraints.  An escape hatch, unused at present: midway through round 6 this
// toolchain's greedy register allocator (ROCm 7.2 clang 22, -disable-machine-licm + iterative-ilp together) segfaulted on the
// two-wavefront workgroup kernel -- and on the stamped road-network build of tools/ablate -- with the constraints in place; later
// changes to the surrounding code made it go away, nothing that was understood.
template <bool SETTLE = true>
__device__ inline void sat_axis(SatAcc &acc, double nx, double ny, double ca_, double ra, double cb_, double rb,
                                double vp, double cdx, double cdy, int k_plus, int k_minus) {
  double min_a = ca_ - ra, max_a = ca_ + ra;
  const double min_b = cb_ - rb, max_b = cb_ + rb;
  int clr = (min_a < min_b ? min_b - max_a : min_a - max_b) > 0 ? 1 : 0;
  if (vp < 0) min_a += vp; else max_a += vp;
  const double g1 = min_b - max_a, g2 = min_a - max_b;
  const double dn = min_a < min_b ? g1 : g2;  // normal +n
  const double dm = max_a > max_b ? g2 : g1;  // normal -n: the intervals are the exact negations, [-max, -min]
  clr |= dn > 0 ? 2 : 0;                       // (disjoint intervals: dn == dm)
  acc.flags &= ~clr;
  // (the two verdicts of THIS direction are folded in here and now: left as booleans the optimiser gathers the compares of all four
  //  directions at the end of the routine, and every direction's intervals stay live until then -- see HWY_SAT_FENCE)
  if constexpr (SETTLE) HWY_SAT_SETTLE(acc.flags);
  // the better of the two normals of this direction (ties: the one the reference meets first)
  const double an = fabs(dn), am = fabs(dm);
  const bool take_minus = am < an || (am == an && k_minus < k_plus);
  const double d = take_minus ? am : an;
  const int k = take_minus ? k_minus : k_plus;
  if (d < acc.min_distance || (d == acc.min_distance && k < acc.order)) {
    acc.min_distance = d;
    acc.order = k;
    // translation_axis = normal if d.dot(normal) > 0 else -normal (utils.py:232-236), in units of +n
    const double dot = cdx * nx + cdy * ny;
    const bool pos = take_minus ? !(dot < 0) : (dot > 0);
    acc.axx = pos ? nx : -nx;
    acc.axy = pos ? ny : -ny;
  }
}
template <bool SETTLE = true>
__device__ inline int pair_collide_body(const Body &A, const Body &B, double dt, double *tx, double *ty) {
  const double diagonal = sqrt(HWY_VEH_LENGTH * HWY_VEH_LENGTH + HWY_VEH_WIDTH * HWY_VEH_WIDTH);
  const double dx = B.x - A.x, dy = B.y - A.y;
  *tx = 0;
  *ty = 0;
  // fast spherical pre-check, with the LOWER index's speed (objects.py:124-127)
  if (sqrt(dx * dx + dy * dy) > (diagonal + diagonal) / 2 + A.v * dt) return 0;
  const double hl = HWY_VEH_LENGTH / 2, hw = HWY_VEH_WIDTH / 2;
  // relative displacement over dt, velocity = speed*(cos h, sin h) (objects.py:165-167)
  const double ddx = A.v * A.c * dt - B.v * B.c * dt, ddy = A.v * A.s * dt - B.v * B.s * dt;
  const double cdx = A.x - B.x, cdy = A.y - B.y;  // centre difference (mean of the corners)
  const double cr = fabs(A.c * B.c + A.s * B.s), sr = fabs(B.s * A.c - B.c * A.s);  // |cos|, |sin| of (h_b - h_a)
  SatAcc acc{3, __builtin_inf(), 0.0, 0.0, 8};
  // the reference's order of the 8 normals: -u_a, +w_a, +u_a, -w_a, -u_b, +w_b, +u_b, -w_b  (positions 0..7)
  // u_a = (cos h_a, sin h_a), w_a = (-sin h_a, cos h_a)
  sat_axis<SETTLE>(acc, A.c, A.s, A.x * A.c + A.y * A.s, hl, B.x * A.c + B.y * A.s, hl * cr + hw * sr, A.c * ddx + A.s * ddy, cdx, cdy, 2, 0);
  if constexpr (SETTLE) HWY_SAT_FENCE();
  sat_axis<SETTLE>(acc, -A.s, A.c, A.y * A.c - A.x * A.s, hw, B.y * A.c - B.x * A.s, hl * sr + hw * cr, A.c * ddy - A.s * ddx, cdx, cdy, 1, 3);
  if constexpr (SETTLE) HWY_SAT_FENCE();
  // u_b, w_b
  sat_axis<SETTLE>(acc, B.c, B.s, A.x * B.c + A.y * B.s, hl * cr + hw * sr, B.x * B.c + B.y * B.s, hl, B.c * ddx + B.s * ddy, cdx, cdy, 6, 4);
  if constexpr (SETTLE) HWY_SAT_FENCE();
  sat_axis<SETTLE>(acc, -B.s, B.c, A.y * B.c - A.x * B.s, hl * sr + hw * cr, B.y * B.c - B.x * B.s, hw, B.c * ddy - B.s * ddx, cdx, cdy, 5, 7);
  if (acc.flags & 2) {
    *tx = acc.min_distance * acc.axx;
    *ty = acc.min_distance * acc.axy;
  }
  return acc.flags;
}
// HWY_OUTLINE_SAT: the SAT as a real function call (s_swappc; operands and results in registers: by value, no scratch) instead
// of an inlined body -- a rare path (0.3 trips per headline env-step) whose registers the frame loop's allocation would then stop
// paying for.  Measured in profiles/r06_history.md.
struct SatOut {
  double tx, ty;
  int r;
};
#ifdef HWY_OUTLINE_SAT
__device__ __attribute__((noinline)) SatOut pair_collide_call(double ax, double ay, double av, double ac, double as, double bx,
                                                              double by, double bv, double bc, double bs, double dt) {
  SatOut o;
  o.r = pair_collide_body(Body{ax, ay, av, ac, as}, Body{bx, by, bv, bc, bs}, dt, &o.tx, &o.ty);
  return o;
}
template <bool SETTLE = true>
__device__ inline int pair_collide(const Body &A, const Body &B, double dt, double *tx, double *ty) {
  const SatOut o = pair_collide_call(A.x, A.y, A.v, A.c, A.s, B.x, B.y, B.v, B.c, B.s, dt);
  *tx = o.tx;
  *ty = o.ty;
  return o.r;
}
#else
template <bool SETTLE = true>
__device__ inline int pair_collide(const Body &A, const Body &B, double dt, double *tx, double *ty) {
  return pair_collide_body<SETTLE>(A, B, dt, tx, ty);
}
#endif

// =============================================================================================
template <int NW>
struct EnvBlock {
  static constexpr int NV = NW * 64;

  // LDS image of one environment (frame-start / post-integration snapshots)
  struct Shared {
    double x[NV], y[NV], v[NV], c[NV], s[NV], ts[NV];
    double aux0[NV], aux1[NV];   // collision translation exchange / observation keys
    unsigned gmax[2];  // full pairwise collisions: reach_key maxima of the frame (displacement along x, speed)
    double rpx[NV], rpy[NV], rpv[NV];  // full pairwise collisions: post-integration position / speed in the frame's RANK order
    int lane[NV], tgt[NV], perm[NV];
    u64 mask[HWY_MAX_LANES][NW];  // lane membership in rank space
    u64 amask[HWY_MAX_LANES][NW]; // abort chain: the vehicles heading for lane T from another lane, in rank space (row T)
    int acode[NV];                // abort chain, index space: decided in this frame | changer << 1
    u64 bal0[2 * NW], bal1[2 * NW], bal2[2 * NW];  // block_ballot slot pairs
    u64 chk[NW];                  // vehicles with check_collisions (index space)
    // full pairwise collisions: per-wavefront list of candidate pairs (lower index | higher index << 8) and the per-vehicle
    // verdicts they meet in (highest partner with a pending impact, crashed flag, that pair's translation: aux1 / ipy)
    unsigned short plist[NW][256];
    int jmax[NV], hit[NV];
    double ipy[NV];
  };

  // ---- workgroup-wide ballot: out[w] = ballot of wave w.  Must be called by ALL threads.
  // ONE barrier: every call site owns a slot PAIR and alternates between its halves (`phase`, flipped here), so the next
  // call's write cannot overtake a late reader of this one, and the write two calls later is at least one barrier behind it.
  __device__ static inline void block_ballot(Shared &sh, bool pred, u64 *slot_pair, int &phase, u64 out[NW]) {
    const u64 b = __ballot(pred);
    if (NW == 1) {
      out[0] = b;
    } else {
      const int i = threadIdx.x;
      u64 *slot = slot_pair + phase * NW;
      phase ^= 1;
      if ((i & 63) == 0) slot[i >> 6] = b;
      __syncthreads();
      for (int w = 0; w < NW; ++w) out[w] = slot[w];
    }
  }
  // the same exchange for any wave-uniform word: out[w] = the value wavefront w passed.  Must be called by ALL threads.
  __device__ static inline void block_share(Shared &sh, u64 value, u64 *slot_pair, int &phase, u64 out[NW]) {
    if (NW == 1) {
      out[0] = value;
      __syncthreads();  // (callers order LDS traffic around the exchange: a one-wavefront workgroup's barrier costs next to nothing)
    } else {
      const int i = threadIdx.x;
      u64 *slot = slot_pair + phase * NW;
      phase ^= 1;
      if ((i & 63) == 0) slot[i >> 6] = value;
      __syncthreads();
      for (int w = 0; w < NW; ++w) out[w] = slot[w];
    }
  }
  // two ballots, one barrier (slot pairs A and B advance together)
  __device__ static inline void block_ballot2(Shared &sh, bool pa, bool pb, u64 *pair_a, u64 *pair_b, int &phase_a, int &phase_b,
                                              u64 out_a[NW], u64 out_b[NW]) {
    const u64 a = __ballot(pa), b = __ballot(pb);
    if (NW == 1) {
      out_a[0] = a;
      out_b[0] = b;
    } else {
      const int i = threadIdx.x;
      u64 *sa = pair_a + phase_a * NW, *sb = pair_b + phase_b * NW;
      phase_a ^= 1;
      phase_b ^= 1;
      if ((i & 63) == 0) { sa[i >> 6] = a; sb[i >> 6] = b; }
      __syncthreads();
      for (int w = 0; w < NW; ++w) { out_a[w] = sa[w]; out_b[w] = sb[w]; }
    }
  }
  __device__ static inline bool any_of(const u64 m[NW]) {
    u64 a = 0;
    for (int w = 0; w < NW; ++w) a |= m[w];
    return a != 0;
  }

  // ---- Road.neighbour_vehicles (road/road.py:483-547) ----------------------------------------
  // Fast path: all x distinct.  rank r of the querying vehicle; returns vehicle indices or -1.
  __device__ static inline void neighbours_ranked(const Shared &sh, int Lq, int r, int *front, int *rear) {
    int fr = -1, rr = -1;
    const int rw = r >> 6, rb = r & 63;
    // front: lowest member rank above r
    for (int w = rw; w < NW; ++w) {
      u64 m = sh.mask[Lq][w];
      if (w == rw) m &= ~(((u64)2 << rb) - 1);  // ranks > r  (2<<63 wraps to 0 => all cleared)
      if (m) { fr = w * 64 + ctz64(m); break; }
    }
    // rear: highest member rank below r
    for (int w = rw; w >= 0; --w) {
      u64 m = sh.mask[Lq][w];
      if (w == rw) m &= (((u64)1 << rb) - 1);   // ranks < r
      if (m) { rr = w * 64 + msb64(m); break; }
    }
    *front = fr < 0 ? -1 : sh.perm[fr];
    *rear = rr < 0 ? -1 : sh.perm[rr];
  }
  // Literal restatement of the reference loop (taken only when two vehicles share the same x).
  __device__ static inline void neighbours_scan(const StepParams &p, const Shared &sh, int Lq, int self,
                                                double s, int *front, int *rear) {
    int f = -1, b = -1;
    double s_front = 0, s_rear = 0;
    for (int j = 0; j < p.N; ++j) {
      if (j == self) continue;
      const double s_v = sh.x[j], lat_v = sh.y[j] - Lq * p.lane_width;
      if (!(fabs(lat_v) <= p.lane_width / 2 + 1.0 && -5.0 <= s_v && s_v < p.road_length + 5.0)) continue;
      if (s <= s_v && (f < 0 || s_v <= s_front)) { s_front = s_v; f = j; }
      if (s_v < s && (b < 0 || s_v > s_rear)) { s_rear = s_v; b = j; }
    }
    *front = f;
    *rear = b;
  }

  // ---- IDM (vehicle/behavior.py:150-217) -------------------------------------------------------
  // free-road term COMFORT_ACC_MAX*(1-(v/v0)^delta), with the CALLER's delta (behavior.py:177-183).
  // (v/v0)^delta = exp(delta*log r): r in [0, ~2], delta in [3.5, 4.5] => |delta*log r| < 4, so the result
  // is within ~3 ulp of a correctly rounded pow at a third of the instruction count.  log r depends on
  // the vehicle only, delta on the caller: every vehicle publishes its own log r once per frame and a
  // MOBIL caller evaluating its would-be follower needs one exp, not a second pow.
  __device__ static inline double idm_log_ratio(const StepParams &p, double v, double ts) {
    const double v0 = clipd(ts, 0.0, p.speed_limit);
    const double r = fmax(v, 0.0) * fast_rcp(abs_not_zero(v0));
    return r > 0.0 ? log_pos(r) : -__builtin_inf();  // r == 0 -> exp(-inf) = 0 == pow(0, delta)
  }
  // the same with 1 / |nz(v0)| handed in (v0 = the clipped target speed changes only with the meta-action of frame 0)
  __device__ static inline double idm_inv_v0(const StepParams &p, double ts) {
    return fast_rcp(abs_not_zero(clipd(ts, 0.0, p.speed_limit)));
  }
  __device__ static inline double idm_log_ratio_inv(double v, double inv_v0) {
    const double r = fmax(v, 0.0) * inv_v0;
    return r > 0.0 ? log_pos(r) : -__builtin_inf();
  }
  __device__ static inline double idm_free_from_log(double log_ratio, double delta) {
    return HWY_COMFORT_ACC_MAX * (1 - exp_bounded(delta * log_ratio));
  }
  __device__ static inline double idm_free(const StepParams &p, double v, double ts, double delta) {
    return idm_free_from_log(idm_log_ratio(p, v, ts), delta);
  }
  // desired gap d* (behavior.py:192-217): projected speed difference, velocity = speed*(cos h, sin h)
  __device__ static inline double desired_gap(double ve, double ce, double se, double vf, double cf, double sf) {
    const double dv = (ve * ce - vf * cf) * ce + (ve * se - vf * sf) * se;
    const double inv_2sqrt_ab = 0.12909944487358055;  // 1 / (2*sqrt(-COMFORT_ACC_MAX*COMFORT_ACC_MIN)) = 1/(2 sqrt 15)
    return HWY_DISTANCE_WANTED + ve * HWY_TIME_WANTED + (ve * dv) * inv_2sqrt_ab;
  }
  // interaction term COMFORT_ACC_MAX*(d*/d)^2 of `ego` behind `front`; d = lane_distance_to
  // (objects.py:183-198): on the straight lane the longitudinal coordinate is x
  __device__ static inline double idm_gap(double xe, double ve, double ce, double se, double xf, double vf,
                                          double cf, double sf) {
    const double q = desired_gap(ve, ce, se, vf, cf, sf) * fast_rcp(not_zero(xf - xe));
    return HWY_COMFORT_ACC_MAX * (q * q);
  }

  // ---- ControlledVehicle.steering_control (vehicle/controller.py:145-187) folded with the first line
  //      of Vehicle.step (kinematics.py:141-142), StraightLane heading 0.
  // Reference chain:  a = -KP_LAT*lat / nz(v);  heading_ref = clip(asin(clip(a)), +-pi/4)
  //                   w = clip(L/2 / nz(v) * KP_H*wrap(heading_ref - h));  slip = asin(w)
  //                   steering = clip(atan(2 tan slip), +-pi/3);  beta = atan(1/2 tan steering)
  // Since tan(asin w) = w/sqrt(1-w^2), tan(atan u) = u and tan is increasing on (-pi/2, pi/2):
  //                   tan(steering) = clip(2w/sqrt(1-w^2), +-tan(pi/3))
  // so this returns  t = tan(beta) = 1/2 tan(steering)  without evaluating slip/steering/beta.
  __device__ static inline double steer_tan_beta(const StepParams &p, double y, double h, double inv_v, int tgt) {
    const double lat = y - tgt * p.lane_width;
    const double a = clipd((-HWY_KP_LATERAL * lat) * inv_v, -1.0, 1.0);
    // clip(asin(a), +-pi/4): asin is only evaluated when it is not going to be clipped
    const double s45 = 0.7071067811865476;  // sin(pi/4) rounded up: |a| >= s45 => |asin a| >= pi/4 (clipped)
    const double heading_ref = a >= s45 ? HWY_PI / 4 : (a <= -s45 ? -HWY_PI / 4 : clipd(asin_bounded(a), -HWY_PI / 4, HWY_PI / 4));
    const double heading_rate_command = HWY_KP_HEADING * wrap_to_pi(heading_ref - h);
    const double w = clipd((HWY_VEH_LENGTH / 2 * inv_v) * heading_rate_command, -1.0, 1.0);
    const double tan_max = 1.7320508075688767;  // tan(MAX_STEERING_ANGLE = fl(pi/3)) in f64
    const double w2 = 1 - w * w;
    // |w| -> 1 sends tan(slip) to infinity: clipped to +-tan_max anyway (w2 <= 1e-12 <=> |tan| >= 1e6)
    const double tan_steer = (w2 <= 1e-12) ? copysign(tan_max, w) : clipd((2 * w) * fast_rsqrt(w2), -tan_max, tan_max);
    return 0.5 * tan_steer;
  }

  // ---- AbstractLane.is_reachable_from (road/lane.py:104-118) -------------------------------------
  __device__ static inline bool reachable(const StepParams &p, int lane, double x, double y) {
    return fabs(y - lane * p.lane_width) <= 2 * p.lane_width && 0 <= x && x < p.road_length + 5.0;
  }
  // ---- RoadNetwork.get_closest_lane_index (road/road.py:55-71, lane.py:132-143) --------------------
  __device__ static inline int closest_lane(const StepParams &p, double x, double y, double h) {
    // All lanes share the heading and longitudinal terms, so the argmin is the lane whose centre is
    // nearest in y: k0 = round(y / width), clamped.  Unless y sits within 1e-9 of a lane boundary (where
    // the reference's rounded sums decide, ties -> lowest id) that is provably the argmin; otherwise run
    // the literal loop.
    // (y * (1 / width) instead of the division: the two quotients differ by an ulp at most, which can move rint() only when y
    // is within ~1e-15 of a lane boundary -- and then `off` fails the test below and the literal loop decides)
    const double k0 = rint(y * p.inv_lane_width);
    const double off = fabs(y - k0 * p.lane_width);
    if (off < p.lane_width / 2 - 1e-9) {
      const int k = (int)k0;
      return k < 0 ? 0 : (k > p.L - 1 ? p.L - 1 : k);
    }
    const double angle = fabs(wrap_to_pi(h - 0.0));
    int best = 0;
    double bd = 0;
    for (int k = 0; k < p.L; ++k) {
      // abs(r) + max(s - length, 0) + max(0 - s, 0) + 1.0*angle, summed left to right
      const double d = fabs(y - k * p.lane_width) + fmax(x - p.road_length, 0.0) + fmax(0 - x, 0.0) + 1.0 * angle;
      if (k == 0 || d < bd) { bd = d; best = k; }
    }
    return best;
  }

  // collision helpers on LDS-resident bodies (see the free functions above EnvBlock)
  __device__ static inline Body body_of(const Shared &sh, int k) { return Body{sh.x[k], sh.y[k], sh.v[k], sh.c[k], sh.s[k]}; }
  __device__ static inline bool surely_apart(const Shared &sh, int a, int b, double dt) {
    return hwy::surely_apart(body_of(sh, a), body_of(sh, b), dt);
  }
  __device__ static inline int pair_collide(const Shared &sh, int a, int b, double dt, double *tx, double *ty) {
    return hwy::pair_collide(body_of(sh, a), body_of(sh, b), dt, tx, ty);
  }

  // ---- Vehicle.to_dict feature (vehicle/kinematics.py:237-261) ---------------------------------------
  __device__ static inline double feature(const StepParams &p, int fid, double x, double y, double h, double v,
                                          double c, double s, int lane) {
    switch (fid) {
      case HWY_FEAT_PRESENCE: return 1.0;
      case HWY_FEAT_X: return x;
      case HWY_FEAT_Y: return y;
      case HWY_FEAT_VX: return v * c;
      case HWY_FEAT_VY: return v * s;
      case HWY_FEAT_HEADING: return h;
      case HWY_FEAT_COS_H: return c;
      case HWY_FEAT_SIN_H: return s;
      case HWY_FEAT_LONG_OFF: return x;
      case HWY_FEAT_LAT_OFF: return y - lane * p.lane_width;
      case HWY_FEAT_ANG_OFF: return wrap_to_pi(h - 0.0);
      default: return 0.0;  // cos_d / sin_d: no route => destination == position => zeros
    }
  }
};

// =============================================================================================
// Per-thread vehicle registers
struct Veh {
  double x, y, h, v, timer, ts, delta, impx, impy, ch, sh;
  int lane, tgt, sidx, flags, rank;
};

// ---- device-side spawn: HighwayEnv._create_vehicles (envs/highway_env.py:72-98) with
//      Vehicle.create_random's rule (vehicle/kinematics.py:50-104), IDMVehicle ctor timer
//      (behavior.py:64), randomize_behavior (behavior.py:66-69), MDPVehicle ladder snap
//      (controller.py:287-293).  Thread i == vehicle i.  Needs two LDS scratch arrays (N and 1 doubles).
// Split in two so that a kernel with several vehicles per thread (hwy_wave2.h) runs the same rule: spawn_draw = everything
// vehicle vi can compute alone, spawn_fill = the rest once its x (the running sum of the steps in creation order) is known.
struct SpawnDraw {
  int lane;
  double speed, step, offset, u_delta;
  bool controlled;
};
__device__ inline SpawnDraw spawn_draw(const StepParams &p, int vi, uint64_t seed, uint32_t episode) {
  SpawnDraw d;
  d.controlled = false;
  for (int a = 0; a < p.A; ++a) d.controlled |= (p.agent_index[a] == vi);
  double u_lane, u_speed, u_pos, u_delta;
  philox_uniform2(seed, (uint32_t)vi, episode, 0u, &u_lane, &u_speed);
  philox_uniform2(seed, (uint32_t)vi, episode, 1u, &u_pos, &u_delta);
  int lane = (int)(u_lane * p.L);
  if (lane > p.L - 1) lane = p.L - 1;
  if (d.controlled && p.rp.initial_lane_id >= 0) lane = p.rp.initial_lane_id;
  // speed: ego 25.0; others uniform(0.7*limit, 0.8*limit) == low + (high-low)*u  (numpy's formula)
  const double lo = 0.7 * p.speed_limit, hi = 0.8 * p.speed_limit;
  const double speed = d.controlled ? 25.0 : lo + (hi - lo) * u_speed;
  const double spacing = d.controlled ? p.rp.ego_spacing : p.rp.other_spacing;
  const double default_spacing = 12 + 1.0 * speed;
  const double offset = spacing * default_spacing * p.rp.lane_factor;
  d.lane = lane;
  d.speed = speed;
  d.offset = offset;
  d.step = offset * (0.9 + (1.1 - 0.9) * u_pos);  // offset * uniform(0.9, 1.1)
  d.u_delta = u_delta;
  return d;
}
__device__ inline void spawn_fill(const StepParams &p, const SpawnDraw &d, double x, int vi, Veh &o) {
  const int lane = d.lane;
  const double speed = d.speed;
  o.x = x;
  o.y = lane * p.lane_width;
  o.h = 0.0;
  o.v = speed;
  o.lane = lane;
  o.tgt = lane;
  o.rank = vi & 0xff;  // x increases with the creation index: the identity is the sorted order
  o.impx = o.impy = 0.0;
  o.ch = 1.0;
  o.sh = 0.0;
  if (d.controlled) {
    // MDPVehicle: speed_index = speed_to_index(target_speed=speed); target_speed = ladder[idx]
    const double xs = (speed - p.target_speeds[0]) / (p.target_speeds[p.n_ts - 1] - p.target_speeds[0]);
    o.sidx = (int)clipd(rint(xs * (p.n_ts - 1)), 0.0, (double)(p.n_ts - 1));
    o.ts = p.target_speeds[o.sidx];
    o.timer = 0.0;
    o.delta = 0.0;
    o.flags = HWY_F_CONTROLLED | HWY_F_CHECK_COLLISIONS;
  } else {
    o.sidx = 0;
    o.ts = speed;
    o.timer = py_mod_pos((o.x + o.y) * HWY_PI, HWY_LC_DELAY);
    o.delta = 3.5 + (4.5 - 3.5) * d.u_delta;
    o.flags = p.rp.fast ? 0 : HWY_F_CHECK_COLLISIONS;
  }
}
template <int NW>
__device__ inline void spawn_env(const StepParams &p, double *scratch_step, double *scratch_base, int e, uint64_t seed,
                                 uint32_t episode, Veh &o) {
  const int i = threadIdx.x;
  const bool active = i < p.N;
  const SpawnDraw d = spawn_draw(p, i, seed, episode);
  if (active) scratch_step[i] = d.step;
  if (active && i == 0) scratch_base[0] = 3 * d.offset;  // first vehicle starts from 3*offset
  __syncthreads();
  // x_k = max_x(existing) + step_k == running sum in creation order (steps are positive)
  double x = scratch_base[0];
  for (int k = 0; k <= i && k < p.N; ++k) x += scratch_step[k];
  __syncthreads();
  spawn_fill(p, d, x, i, o);
  (void)e;
}

// ---- OccupancyGridObservation.observe (envs/common/observation.py:354-413) for one observer -----------
// The reference scatters every vehicle's features into a [F][W][H] grid walking the vehicle list in
// REVERSE (df[::-1]) so that the lowest index wins a contested cell, paints the on-road layer from lane
// waypoints (fill_road_layer_by_lanes, :454-484), clips and maps NaN (empty) to 0.  Here: an atomic min
// per vehicle elects each cell's owner, owners write their features, a cell-strided pass writes the
// on-road layer and the zeros.  Cross-thread traffic goes through a small global workspace with
// agent-scope atomics (L2), so the routine is the same for one- and multi-wavefront environments.
// Observer data (position, speed, cos/sin heading) is passed in by the caller.
__device__ inline void grid_ws_store(int32_t *q, int32_t v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int32_t grid_ws_load(int32_t *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void grid_ws_min(int32_t *q, int32_t v) { __hip_atomic_fetch_min(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// OccupancyGridObservation(as_image=True) (observation.py:408-409): ((clip(v, -1, 1) + 1) / 2 * 255).astype(uint8) truncates
__device__ inline double grid_image(double v) { return (double)(int)((clipd(v, -1.0, 1.0) + 1.0) / 2 * 255); }

__device__ inline void grid_cell(const StepParams &p, double px, double py, double ec, double es, int *ci, int *cj) {
  if (p.flags & HWY_C_GRID_ALIGN) {  // pos_to_index: [[c, s], [-s, c]] @ position  (observation.py:431-435)
    const double qx = ec * px + es * py, qy = -es * px + ec * py;
    px = qx;
    py = qy;
  }
  *ci = (int)floor((px - p.gmin_x) / p.gstep_x);
  *cj = (int)floor((py - p.gmin_y) / p.gstep_y);
}

// eo: the row of the output planes this environment writes (== e except in a multi-step launch, hwy_rollout_device)
template <int NW>
__device__ inline void observe_grid(const StepParams &p, int e, int a, const Veh &me, double ex, double ey, double ev,
                                    double ec, double es, int eo = -1) {
  eo = eo < 0 ? e : eo;
  typedef EnvBlock<NW> B;
  const int i = threadIdx.x, NT = NW * 64;
  const bool active = i < p.N;
  const int W = p.gW, H = p.gH, WH = W * H, F = p.F;
  int32_t *own = p.grid_ws + ((size_t)e * p.A + a) * 2 * (size_t)WH, *road = own + WH;
  float *out = p.obs + ((size_t)eo * p.A + a) * (size_t)F * WH;
  for (int t = i; t < WH; t += NT) {
    grid_ws_store(own + t, 0x7fffffff);
    grid_ws_store(road + t, 0);
  }
  __syncthreads();
  // vehicles claim their cell (coordinates relative to the observer; normalised then de-normalised when
  // x / y are in features_range, exactly like observation.py:377-396)
  int my_ci = -1, my_cj = -1;
  if (active) {
    double x = me.x - ex, y = me.y - ey;
    if (p.rx0 > -__builtin_inf()) x = lmap(lmap(x, p.rx0, p.rx1, -1.0, 1.0), -1.0, 1.0, p.rx0, p.rx1);
    if (p.ry0 > -__builtin_inf()) y = lmap(lmap(y, p.ry0, p.ry1, -1.0, 1.0), -1.0, 1.0, p.ry0, p.ry1);
    int ci, cj;
    grid_cell(p, x, y, ec, es, &ci, &cj);
    if (0 <= ci && ci < W && 0 <= cj && cj < H) {
      my_ci = ci;
      my_cj = cj;
      grid_ws_min(own + ci * H + cj, i);
    }
  }
  // on-road layer: waypoints every min(grid_step) within +-100 m of the observer on every lane
  bool has_road = false;
  for (int f = 0; f < F; ++f) has_road |= (p.feat[f] == HWY_FEAT_ON_ROAD);
  if (has_road) {
    const double start = ex - 100.0;  // origin = lane.local_coordinates(observer)[0] = x on the straight road
    for (int t = i; t < p.L * p.g_nwp; t += NT) {
      const int k = t / p.g_nwp, j = t - k * p.g_nwp;
      const double wp = clipd(start + j * p.g_spacing, 0.0, p.road_length);
      int ci, cj;
      grid_cell(p, wp - ex, k * p.lane_width - ey, ec, es, &ci, &cj);  // lane.position(wp, 0) - observer.position
      if (0 <= ci && ci < W && 0 <= cj && cj < H) grid_ws_store(road + ci * H + cj, 1);
    }
  }
  __syncthreads();
  const bool clip = (p.flags & HWY_C_OBS_CLIP) != 0;
  if (my_ci >= 0 && grid_ws_load(own + my_ci * H + my_cj) == i) {  // I own my cell: write the vehicle layers
    for (int f = 0; f < F; ++f) {
      const int fid = p.feat[f];
      if (fid == HWY_FEAT_ON_ROAD) continue;
      double val = B::feature(p, fid, me.x, me.y, me.h, me.v, me.ch, me.sh, me.lane);
      const bool rel = fid == HWY_FEAT_X || fid == HWY_FEAT_Y || fid == HWY_FEAT_VX || fid == HWY_FEAT_VY;
      if (rel) {
        val -= fid == HWY_FEAT_X ? ex : fid == HWY_FEAT_Y ? ey : fid == HWY_FEAT_VX ? ev * ec : ev * es;
        const double r0 = fid == HWY_FEAT_X ? p.rx0 : fid == HWY_FEAT_Y ? p.ry0 : fid == HWY_FEAT_VX ? p.rvx0 : p.rvy0;
        const double r1 = fid == HWY_FEAT_X ? p.rx1 : fid == HWY_FEAT_Y ? p.ry1 : fid == HWY_FEAT_VX ? p.rvx1 : p.rvy1;
        if (r0 > -__builtin_inf()) val = lmap(val, r0, r1, -1.0, 1.0);
      }
      if (clip) val = clipd(val, -1.0, 1.0);
      if (p.flags & HWY_C_GRID_IMAGE) val = grid_image(val);
      out[(f * W + my_ci) * H + my_cj] = (float)val;
    }
  }
  for (int t = i; t < F * WH; t += NT) {  // everything the owners do not write
    const int f = t / WH, c = t - f * WH;
    if (p.feat[f] == HWY_FEAT_ON_ROAD) out[t] = grid_ws_load(road + c) ? ((p.flags & HWY_C_GRID_IMAGE) ? 255.0f : 1.0f) : 0.0f;
    else if (grid_ws_load(own + c) == 0x7fffffff) out[t] = 0.0f;  // NaN (empty) -> 0
  }
  __syncthreads();  // the workspace of this (env, agent) may be reused by the next call
}

// ---- KinematicObservation.observe (envs/common/observation.py:234-276) + Road.close_objects_to
//      (road/road.py:421-450) + reward/termination (envs/highway_env.py:100-151) for every agent.
//      Expects sh.x/y/v/c/s to hold the CURRENT state of all vehicles.  Block-uniform control flow.
template <int NW>
__device__ inline void observe_env(const StepParams &p, typename EnvBlock<NW>::Shared &sh, int e, const Veh &me,
                                   bool write_reward, int eo = -1) {
  eo = eo < 0 ? e : eo;  // row of the output planes (== e except in a multi-step launch, hwy_rollout_device)
  const bool kin = p.obs_type == HWY_OBS_KINEMATICS;
  typedef EnvBlock<NW> B;
  const int i = threadIdx.x;
  const bool active = i < p.N;
  const int V = p.V, F = p.F;
  int ph_elig = 0;
  for (int a = 0; a < p.A; ++a) {
    const int ia = p.agent_index[a];
    const double ex = sh.x[ia], ey = sh.y[ia], ev = sh.v[ia], ec = sh.c[ia], es = sh.s[ia];
    // eligibility (road.py:430-436): within perception distance, not self, not more than 2*LENGTH behind
    const double dxe = me.x - ex, dye = me.y - ey;
    const double d_lane = me.x - ex;  // lane_distance_to on the ego's (straight) lane
    const bool elig = active && i != ia && (sqrt(dxe * dxe + dye * dye) < p.perception) &&
                      ((p.flags & HWY_C_OBS_SEE_BEHIND) || (-2 * HWY_VEH_LENGTH < d_lane));
    const double key = elig ? ((p.flags & HWY_C_OBS_UNSORTED) ? 0.0 : fabs(d_lane)) : __builtin_inf();  // (sort=False: list order)
    __syncthreads();  // previous users of aux0 are done
    sh.aux0[i] = key;
    __syncthreads();
    u64 em[NW];
    B::block_ballot(sh, elig, sh.bal0, ph_elig, em);
    int n_elig = 0;
    for (int w = 0; w < NW; ++w) n_elig += __popcll(em[w]);
    const int m = n_elig < V - 1 ? n_elig : V - 1;  // rows 1..m are filled
    // stable sort position (sorted() keeps list order among equal keys)
    int pos = 0;
    if (elig) {
      for (int k = 0; k < p.N; ++k) {
        const double kk = sh.aux0[k];
        pos += (kk < key) || (kk == key && k < i);
      }
    }
    if (p.obs && !kin) observe_grid<NW>(p, e, a, me, ex, ey, ev, ec, es, eo);
    if (p.obs && kin) {
      float *out = p.obs + ((size_t)eo * p.A + a) * (size_t)(V * F);
      const int row = (i == ia) ? 0 : (elig && pos < V - 1 ? pos + 1 : -1);
      if (active && row >= 0) {
        for (int f = 0; f < F; ++f) {
          const int fid = p.feat[f];
          double val = B::feature(p, fid, me.x, me.y, me.h, me.v, me.ch, me.sh, me.lane);
          const bool rel = fid == HWY_FEAT_X || fid == HWY_FEAT_Y || fid == HWY_FEAT_VX || fid == HWY_FEAT_VY;
          if (row > 0 && rel && !(p.flags & HWY_C_OBS_ABSOLUTE)) {
            const double origin = fid == HWY_FEAT_X ? ex : fid == HWY_FEAT_Y ? ey : fid == HWY_FEAT_VX ? ev * ec : ev * es;
            val -= origin;
          }
          if (rel && (p.flags & HWY_C_OBS_NORMALIZE)) {
            const double r0 = fid == HWY_FEAT_X ? p.rx0 : fid == HWY_FEAT_Y ? p.ry0 : fid == HWY_FEAT_VX ? p.rvx0 : p.rvy0;
            const double r1 = fid == HWY_FEAT_X ? p.rx1 : fid == HWY_FEAT_Y ? p.ry1 : fid == HWY_FEAT_VX ? p.rvx1 : p.rvy1;
            if (r0 > -__builtin_inf()) {  // -inf/+inf == feature absent from features_range
              val = lmap(val, r0, r1, -1.0, 1.0);
              if (p.flags & HWY_C_OBS_CLIP) val = clipd(val, -1.0, 1.0);
            }
          }
          out[row * F + f] = (float)val;
        }
      }
      // zero-fill the missing rows (observation.py:262-267)
      for (int t = i; t < V * F; t += B::NV)
        if (t / F > m) out[t] = 0.0f;
    }
    if (write_reward && i == ia) {
      // HighwayEnv._rewards / _reward (highway_env.py:100-139)
      const bool crashed = (me.flags & HWY_F_CRASHED) != 0;
      const bool on_road = fabs(me.y - me.lane * p.lane_width) <= p.lane_width / 2 + 0.0 && -5.0 <= me.x &&
                           me.x < p.road_length + 5.0;
      const double forward_speed = me.v * me.ch;
      const double scaled_speed = lmap(forward_speed, p.rs0, p.rs1, 0.0, 1.0);
      const int nl = p.L - 1 > 1 ? p.L - 1 : 1;
      double reward = 0.0;
      reward = reward + p.collision_reward * (crashed ? 1.0 : 0.0);
      reward = reward + p.right_lane_reward * ((double)me.tgt / (double)nl);
      reward = reward + p.high_speed_reward * clipd(scaled_speed, 0.0, 1.0);
      reward = reward + 0.0 * (on_road ? 1.0 : 0.0);
      if (p.flags & HWY_C_NORMALIZE_REWARD)
        reward = lmap(reward, p.collision_reward, p.high_speed_reward + p.right_lane_reward, 0.0, 1.0);
      reward *= (on_road ? 1.0 : 0.0);
      p.reward[(size_t)eo * p.A + a] = reward;
      if (p.info_speed) p.info_speed[(size_t)eo * p.A + a] = me.v;
      if (p.info_crashed) p.info_crashed[(size_t)eo * p.A + a] = crashed ? 1 : 0;
      if (a == 0) {
        // _is_terminated looks at controlled_vehicles[0] (highway_env.py:141-147); time += 1/policy_frequency
        // (abstract.py:274); _is_truncated: time >= duration (highway_env.py:149-151)
        const bool term = crashed || ((p.flags & HWY_C_OFFROAD_TERMINAL) && !on_road);
        const double t = p.st.time[e] + p.policy_dt;
        const bool trunc = t >= p.duration;
        p.st.time[e] = t;
        p.terminated[eo] = term ? 1 : 0;
        p.truncated[eo] = trunc ? 1 : 0;
        if (p.autoreset) p.st.done[e] = (term || trunc) ? 1 : 0;
      }
    }
  }
}

// HBM traffic discipline: every dynamic word is read once and written once per policy step; the
// pending-impact pair is only touched for vehicles that have one (flag bit), and per-vehicle constants
// (IDM exponent, an IDM vehicle's target speed) are never written back by the step kernel.
__device__ inline void load_vehicle_at(const StepParams &p, int e, int i, Veh &o) {  // i: vehicle index (thread i of the workgroup kernels)
  o = Veh{};
  if (i < p.N) {
    const size_t k = (size_t)e * p.pitch + i;
    o.x = p.st.x[k]; o.y = p.st.y[k]; o.h = p.st.heading[k]; o.v = p.st.speed[k];
    o.timer = p.st.timer[k]; o.ts = p.st.target_speed[k]; o.delta = p.st.delta[k];
    const int w = p.st.packed[k];
    o.lane = word_lane(w); o.tgt = word_target(w); o.sidx = word_speed_index(w); o.flags = word_flags(w);
    o.rank = word_rank(w);
    if (o.flags & HWY_F_HAS_IMPACT) {
      o.impx = p.st.impact_x[k];
      o.impy = p.st.impact_y[k];
    }
    sincos_bounded(o.h, &o.sh, &o.ch);
  }
}
template <int NW>
__device__ inline void load_vehicle(const StepParams &p, int e, Veh &o) { load_vehicle_at(p, e, threadIdx.x, o); }
// full = true: spawn / reset (every field); false: end of a step (dynamic fields only)
// NaN guard: vehicles whose position / heading / speed left the finite numbers are COUNTED where the state is
// written back (HWY_CTR_NONFINITE_STORES, hwy_get_counters) -- the simulation has no operation that recovers from a NaN (it spreads
// through the neighbour gaps to the whole lane), so a nonzero count says the state handed to hwy_set_state, or a kernel, is broken.
// x - x is 0 for every finite x and NaN for +-inf and NaN: one compare per thread, one ballot, an atomic only on the broken path.
#ifndef HWY_PIN_POINTERS
#define HWY_PIN_POINTERS(a, b) asm volatile("" : "+s"(a), "+s"(b))   // both are in SGPRs here (the emulator defines it away)
#define HWY_GLOBAL_F64 __attribute__((address_space(1))) double        // (a pointer into global memory stays one through the asm: no flat_store)
#endif
__device__ inline void count_nonfinite(unsigned long long *counters, bool bad) {
  const unsigned long long m = __ballot(bad);
  if (m && counters && (threadIdx.x & 63) == 0) atomicAdd(&counters[HWY_CTR_NONFINITE_STORES], (unsigned long long)__popcll(m));
}
__device__ inline void store_vehicle_at(const StepParams &p, int e, int i, const Veh &o, bool full = true) {
  // The stores below are the last thing a wavefront does, and a wavefront waits for its outstanding stores (s_waitcnt vmcnt) before
  // it may overwrite a register one of them still reads: everything that computes -- the NaN guard's predicate, the packed word,
  // the plane pointers -- comes FIRST, so that nothing but the stores and scalar bookkeeping is left between the first store and
  // s_endpgm (the round-4 order -- stores, then the guard's four subtractions in the registers the stores had just been fed from --
  // parked every wavefront for a memory round trip at its very end).
  const bool bad = i < p.N && !((o.x - o.x) + (o.y - o.y) + (o.h - o.h) + (o.v - o.v) == 0.0);
  const unsigned long long bad_m = __ballot(bad);
  if (i < p.N) {
    const size_t k = (size_t)e * p.pitch + i;
    const int32_t word = pack_word(o.lane, o.tgt, o.sidx, o.flags, o.rank);
    // (the two plane pointers as VALUES before the predicated stores: the optimiser merges the two stores into one with a per-lane
    //  choice of the destination, and -- when `p` is the re-read view of the kernel arguments, HWY_RELOAD_PARAMS -- made that choice
    //  by LOADING the pointer from the argument segment with a per-lane address: one more round trip before the wavefront retires)
    HWY_GLOBAL_F64 *timer_plane = (HWY_GLOBAL_F64 *)p.st.timer, *ts_plane = (HWY_GLOBAL_F64 *)p.st.target_speed;
    HWY_PIN_POINTERS(timer_plane, ts_plane);
    p.st.x[k] = o.x; p.st.y[k] = o.y; p.st.heading[k] = o.h; p.st.speed[k] = o.v;
    p.st.packed[k] = word;
    if (full || !(o.flags & HWY_F_CONTROLLED)) timer_plane[k] = o.timer;
    if (full || (o.flags & HWY_F_CONTROLLED)) ts_plane[k] = o.ts;
    if (full) p.st.delta[k] = o.delta;
    if (full || (o.flags & HWY_F_HAS_IMPACT)) {
      p.st.impact_x[k] = o.impx;
      p.st.impact_y[k] = o.impy;
    }
  }
  if (bad_m && p.counters && (threadIdx.x & 63) == 0)
    atomicAdd(&p.counters[HWY_CTR_NONFINITE_STORES], (unsigned long long)__popcll(bad_m));
}
template <int NW>
__device__ inline void store_vehicle(const StepParams &p, int e, const Veh &o, bool full = true) {
  store_vehicle_at(p, e, threadIdx.x, o, full);
}
template <int NW>
__device__ inline void publish(typename EnvBlock<NW>::Shared &sh, const Veh &me, bool active) {
  const int i = threadIdx.x;
  if (active) {
    sh.x[i] = me.x; sh.y[i] = me.y; sh.v[i] = me.v; sh.c[i] = me.ch; sh.s[i] = me.sh;
    sh.lane[i] = me.lane; sh.tgt[i] = me.tgt;
  }
  sh.ts[i] = me.ts;  // (every thread: section G of the step reads its own slot back, see there)
}

// =============================================================================================
// Reset kernel: AbstractEnv.reset for the masked environments + first observation.
template <int NW>
__global__ void __launch_bounds__(NW * 64) hwy_reset_kernel(const StepParams p) {
  typedef EnvBlock<NW> B;
  __shared__ typename B::Shared sh;
  const int e = blockIdx.x, i = threadIdx.x;
  if (p.reset_mask && !p.reset_mask[e]) return;  // block-uniform
  const bool active = i < p.N;
  Veh me = Veh{};
  const uint64_t seed = p.reset_seeds ? p.reset_seeds[e] : p.rp.base_seed + (uint64_t)e;
  spawn_env<NW>(p, sh.aux0, sh.aux1, e, seed, 0u, me);
  publish<NW>(sh, me, active);
  __syncthreads();
  observe_env<NW>(p, sh, e, me, false);
  store_vehicle<NW>(p, e, me);
  if (i == 0) {
    p.st.time[e] = 0.0;
    p.st.done[e] = 0;
    p.st.episode[e] = 0;
  }
}

// Observation-only kernel (hwy_observe).
template <int NW>
__global__ void __launch_bounds__(NW * 64) hwy_observe_kernel(const StepParams p) {
  typedef EnvBlock<NW> B;
  __shared__ typename B::Shared sh;
  const int e = blockIdx.x, i = threadIdx.x;
  Veh me;
  load_vehicle<NW>(p, e, me);
  publish<NW>(sh, me, i < p.N);
  __syncthreads();
  observe_env<NW>(p, sh, e, me, false);
}

// =============================================================================================
// The fused policy-step kernel.
// WPE = minimum waves per SIMD the register allocator must leave room for (occupancy knob).
// One policy step of environment e by its workgroup; eo = row of the action / output planes (hwy_wave.h: observe_wave).
// ---- taking turns at the SIMD's issue port ------------------------------------------------------------------------------
// The hardware arbiter issues from the OLDEST ready wavefront first.  With exactly four environments per SIMD (4096 envs)
// that lets the first wavefront run as if it were alone (latency-bound: ~30 % of the VALU issue slots), the next two fill
// the gaps, the fourth gets the leftovers and then finishes ALONE at the same 30 %: measured end times of the four
// wavefronts of a SIMD 24 / 31 / 37 / 43 us (tools/wave_timeline2.py).  Equal shares would end all four together, sooner.
// So every wavefront sets its own priority (s_setprio, 0..3) to (hardware wave slot + clock >> shift) & 3: at any time the
// four wavefronts of a SIMD hold four different priorities and the top one changes every 2^shift clock ticks.  The clock
// is read with s_memtime one turn ahead (the value requested at the previous checkpoint is used), so no checkpoint waits
// for it.  Scheduling only: no result depends on it.
struct WaveTurn {
  unsigned long long t;
  int slot, shift;
  unsigned recip;  // != 0: turns of shift x 64 ticks (the turn index by a multiply-high with floor(2^32 / shift): scalar instructions)
};
__device__ inline void wave_turn_init(WaveTurn &w, int shift, unsigned recip = 0) {
#ifdef HWY_HAVE_SETPRIO
  w.shift = shift;
  w.recip = recip;
  w.slot = shift > 0 ? (int)(__builtin_amdgcn_s_getreg((4 << 11) | 4 /* HW_REG_HW_ID, WAVE_ID bits 3:0 */) & 3) : 0;
  w.t = shift > 0 ? __builtin_amdgcn_s_memtime() : 0ull;
#else
  (void)w; (void)shift;
#endif
}
// Workgroup kernel (several wavefronts per environment, joined by barriers): all wavefronts of a workgroup must share one
// priority, so the slot is the workgroup's slot on its CU (HW_ID.TG_ID) -- the wavefronts that meet on a SIMD belong to
// different workgroups of the CU.
__device__ inline void wave_turn_init_workgroup(WaveTurn &w, int shift, unsigned recip = 0) {
#ifdef HWY_HAVE_SETPRIO
  w.shift = shift;
  w.recip = recip;
  w.slot = shift > 0 ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4 /* HW_REG_HW_ID, TG_ID bits 19:16 */) & 3) : 0;
  w.t = shift > 0 ? __builtin_amdgcn_s_memtime() : 0ull;
#else
  (void)w; (void)shift;
#endif
}
__device__ inline void wave_turn(WaveTurn &w) {
#ifdef HWY_HAVE_SETPRIO
  if (w.shift > 0) {  // wave-uniform (SGPR)
    const int turn = w.recip ? (int)__umulhi((unsigned)(w.t >> 6), w.recip) : (int)(w.t >> w.shift);
    const int prio = (w.slot + turn) & 3;
    w.t = __builtin_amdgcn_s_memtime();
    if (prio == 0) __builtin_amdgcn_s_setprio(0);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
  }
#else
  (void)w;
#endif
}

template <int NW>
__device__ __forceinline__ void block_policy_step(const StepParams &p, typename EnvBlock<NW>::Shared &sh, const int e, const int eo) {
  typedef EnvBlock<NW> B;
  const int i = threadIdx.x;
  const int N = p.N;
  const bool active = i < N;
  const int lane_id = i & 63, wave = i >> 6;

  // ---- auto-reset (gymnasium vector "next-step" mode): re-spawn instead of stepping ------------
  if (p.autoreset && p.st.done[e]) {  // block-uniform
    Veh me = Veh{};
    const uint32_t episode = p.st.episode[e] + 1u;
    spawn_env<NW>(p, sh.aux0, sh.aux1, e, p.rp.base_seed + (uint64_t)e, episode, me);
    publish<NW>(sh, me, active);
    __syncthreads();
    observe_env<NW>(p, sh, e, me, false, eo);
    store_vehicle<NW>(p, e, me);
    if (active && (me.flags & HWY_F_CONTROLLED)) {
      for (int a = 0; a < p.A; ++a)
        if (p.agent_index[a] == i) {
          p.reward[(size_t)eo * p.A + a] = 0.0;
          if (p.info_speed) p.info_speed[(size_t)eo * p.A + a] = me.v;
          if (p.info_crashed) p.info_crashed[(size_t)eo * p.A + a] = 0;
        }
    }
    if (i == 0) {
      p.st.time[e] = 0.0;
      p.st.done[e] = 0;
      p.st.episode[e] = episode;
      p.terminated[eo] = 0;
      p.truncated[eo] = 0;
    }
    return;
  }

  Veh me;
  load_vehicle<NW>(p, e, me);
  const bool controlled = active && (me.flags & HWY_F_CONTROLLED);
  const bool idm = active && !controlled;
  int agent = 0;
  if (controlled)
    for (int a = 0; a < p.A; ++a)
      if (p.agent_index[a] == i) agent = a;

  WaveTurn turn;
  wave_turn_init_workgroup(turn, p.prio_shift, p.prio_recip);

  // static collision-check membership (index space)
  u64 chk[NW];
  int ph0 = 0, ph1 = 0, ph2 = 0;  // block_ballot phases of the three slot pairs
  B::block_ballot(sh, active && (me.flags & HWY_F_CHECK_COLLISIONS), sh.bal0, ph0, chk);
  // (kept in LDS for the sparse-checker loop of section G: wave-uniform masks the frame loop would otherwise hold in 2 NW VGPRs --
  //  the scalar registers are long gone -- on top of the SAT's peak; published by the barrier of the first frame's snapshot)
  if (i == 0)
    for (int w = 0; w < NW; ++w) sh.chk[w] = chk[w];
  // rank along the road, carried from frame to frame and from step to step (hint in the packed word) and merely VERIFIED
  // (section C); sh.perm = its inverse.  Idle threads keep their own slot so that the table stays a permutation.
  int rank = active ? me.rank : i;
  sh.perm[i] = i;
  int n_chk = 0;
  for (int w = 0; w < NW; ++w) n_chk += __popcll(chk[w]);
  const bool all_check = n_chk == N;  // highway-v0: full pairwise; highway-fast-v0: ego only

  for (int fr = 0; fr < p.n_frames; ++fr) {
    wave_turn(turn);
    // ---- A. action_type.act (abstract.py:294-304) -> MDPVehicle.act(label) (controller.py:295-315):
    //         target updates only; the controllers run below with Road.act (same state => same command)
    if (fr == 0 && p.actions && controlled) {
      const int act = HWY_ACTION_TO_ALL(p.action_set, p.actions[(size_t)eo * p.A + agent]);
      if (act == HWY_FASTER || act == HWY_SLOWER) {
        const double xs = (me.v - p.target_speeds[0]) / (p.target_speeds[p.n_ts - 1] - p.target_speeds[0]);
        int idx = (int)clipd(rint(xs * (p.n_ts - 1)), 0.0, (double)(p.n_ts - 1)) + (act == HWY_FASTER ? 1 : -1);
        idx = idx < 0 ? 0 : (idx > p.n_ts - 1 ? p.n_ts - 1 : idx);
        me.sidx = idx;
        me.ts = p.target_speeds[idx];
      } else if (act == HWY_LANE_LEFT || act == HWY_LANE_RIGHT) {
        int id = me.tgt + (act == HWY_LANE_RIGHT ? 1 : -1);
        id = id < 0 ? 0 : (id > p.L - 1 ? p.L - 1 : id);
        if (B::reachable(p, id, me.x, me.y)) me.tgt = id;
      }
    }

    // ---- B. frame-start snapshot -> LDS -----------------------------------------------------------
    // Only for the first frame: nothing the snapshot holds (position, speed, heading, lane, target lane, target speed)
    // changes between the post-integration publish of section G and the start of the next frame -- collisions only set
    // flags and pending impacts -- and G ends with a barrier.
    if (fr == 0) {  // block-uniform
      __syncthreads();  // (the static ballots above are read)
      publish<NW>(sh, me, active);
      if (active && rank >= 0 && rank < N) sh.perm[rank] = i;  // (a stale hint may collide: verified below)
      __syncthreads();
    }

    // ---- C. rank along the road + lane membership masks ---------------------------------------------
    // The order along the road changes in a few per cent of the frames, so the rank of the previous frame is VERIFIED
    // instead of recounted: my slot of the inverse table still holds me, and the vehicle of the next rank is strictly
    // ahead of me.  If that holds for every vehicle the table is the sorted permutation and all x are distinct.
    bool stale = false;
    if (active) {
      stale = rank < 0 || rank >= N || sh.perm[rank] != i;
      if (!stale && rank + 1 < N) stale = !(me.x < sh.x[sh.perm[rank + 1]]);
    }
    // lane membership masks in rank space (thread r speaks for the vehicle of rank r): written OPTIMISTICALLY from the table as
    // it stands, so that the barrier of the verification publishes them too; a stale table (rare) redoes them below
    auto write_masks = [&]() {
      const int j = active ? sh.perm[i] : 0;
      const double xj = sh.x[j], yj = sh.y[j];
      const bool inr = active && (-5.0 <= xj) && (xj < p.road_length + 5.0);
      for (int L = 0; L < p.L; ++L) {
        const bool m = inr && (fabs(yj - L * p.lane_width) <= p.lane_width / 2 + 1.0);
        const u64 b = __ballot(m);
        if (lane_id == 0) { sh.mask[L][wave] = b; sh.amask[L][wave] = 0; }  // (amask: the abort chain of section D ORs into it)
      }
    };
    write_masks();
    bool has_tie = false;
    if (__syncthreads_or(stale)) {  // block-uniform
      // rank = #{j : x_j < x_i} when all x are distinct; a tie (count_le - count_lt > 1, self included)
      // is resolved by list order, exactly like a stable sort (rare)
      int cnt_lt = 0, cnt_le = 0;
#pragma unroll 4
      for (int j = 0; j < N; ++j) {
        const double xj = sh.x[j];
        cnt_lt += (xj < me.x) ? 1 : 0;
        cnt_le += (xj <= me.x) ? 1 : 0;
      }
      const bool tie = (cnt_le - cnt_lt) > 1;
      rank = active ? cnt_lt : i;
      if (active && tie)
        for (int j = 0; j < i; ++j) rank += (sh.x[j] == me.x) ? 1 : 0;
      sh.perm[rank] = i;
      u64 tm[NW];
      B::block_ballot(sh, active && tie, sh.bal1, ph1, tm);  // (its barrier also publishes the table)
      has_tie = B::any_of(tm);
      write_masks();
      __syncthreads();
    }

    wave_turn(turn);
    // ---- D. Road.act: lane-change policy (behavior.py:219-263) ----------------------------------------
    const bool crashed0 = (me.flags & HWY_F_CRASHED) != 0;
    const bool drives = idm && !crashed0;  // IDMVehicle.act returns early when crashed (behavior.py:102-103)
    const int tgt_old = me.tgt;
    const bool changer = drives && me.lane != me.tgt;
    const bool decide = drives && me.lane == me.tgt && (HWY_LC_DELAY < me.timer);  // utils.do_every
    int f_own = -1, r_own = -1;
    double free_self = 0.0;
    if (drives) {
      if (!has_tie) B::neighbours_ranked(sh, me.lane, rank, &f_own, &r_own);
      else B::neighbours_scan(p, sh, me.lane, i, me.x, &f_own, &r_own);
      free_self = B::idm_free(p, me.v, me.ts, me.delta);
    }
    if (!has_tie) {  // block-uniform
      // MOBIL compacted per wavefront (hwy_wave.h has the one-wavefront version, round 6): a vehicle decides once per second, so in
      // a given frame ~4 of a wavefront's 64 vehicles do -- and rounds 1-5 ran both side lanes one after the other for the whole
      // wavefront whenever ONE of them decided.  Decider number d of the wavefront hands (free-road term, own-lane acceleration,
      // delta, index | lane | side bits | rank) over through the wavefront's own stretch of LDS planes that only section G uses;
      // lane t of the wavefront evaluates side t & 1 of decider t >> 1 from the frame-start snapshot (the very doubles the decider
      // holds in registers); the verdicts come back as a ballot.  Same operations on the same values as the per-thread form below
      // (kept for frames with equal x): bit-identical.
      const bool moving = !(fabs(me.v) < 1);
      const bool cl = decide && me.lane - 1 >= 0 && B::reachable(p, me.lane - 1, me.x, me.y) && moving;
      const bool cr = decide && me.lane + 1 < p.L && B::reachable(p, me.lane + 1, me.x, me.y) && moving;
      if (decide) me.timer = 0.0;
      const u64 dm = __ballot(cl || cr);
      if (dm) {  // wave-uniform
        const int wbase = wave * 64, n_dec = __popcll(dm);
        const int d = __popcll(dm & (((u64)1 << lane_id) - 1));
        if (cl || cr) {
          // self_a: my IDM acceleration behind my current leader (old_preceding)
          const double self_a = free_self - (f_own >= 0 ? B::idm_gap(me.x, me.v, me.ch, me.sh, sh.x[f_own], sh.v[f_own],
                                                                    sh.c[f_own], sh.s[f_own]) : 0.0);
          sh.rpx[wbase + d] = free_self; sh.rpy[wbase + d] = self_a; sh.rpv[wbase + d] = me.delta;
          sh.jmax[wbase + d] = i | (me.lane << 8) | (cl ? 1 << 12 : 0) | (cr ? 1 << 13 : 0) | (rank << 16);
        }
        HWY_WAVEFRONT_FENCE();  // (written and read by this wavefront only)
        int bits = 0;
        for (int base = 0; base < 2 * n_dec; base += 64) {  // wave-uniform
          const int t = base + lane_id;
          const bool tv = t < 2 * n_dec;
          const int dd = tv ? t >> 1 : 0, side = t & 1;
          const int w_ = sh.jmax[wbase + dd];
          const int vi = w_ & 255, ln = (w_ >> 8) & 15, rk = (w_ >> 16) & 255;
          const bool en = tv && ((w_ >> (12 + side)) & 1);
          const double fs = sh.rpx[wbase + dd], sa = sh.rpy[wbase + dd], dl = sh.rpv[wbase + dd];
          const double ex = sh.x[vi], ev = sh.v[vi], ec = sh.c[vi], es = sh.s[vi];
          // mobil(cand)  (behavior.py:265-324; route is None; POLITENESS == 0 so the followers' terms are multiplied by 0.0 --
          // finite by construction -- and only the safety criterion needs new_following): incentive first, it needs no pow()
          // and rejects ~98 % of the candidates
          int nprec, nfoll;
          B::neighbours_ranked(sh, en ? ln + (side ? 1 : -1) : ln, rk, &nprec, &nfoll);
          const int gp = nprec < 0 ? 0 : nprec;
          const double self_pred_a = fs - (nprec >= 0 ? B::idm_gap(ex, ev, ec, es, sh.x[gp], sh.v[gp], sh.c[gp], sh.s[gp]) : 0.0);
          const double jerk = self_pred_a - sa;
          bool ok = en && !(jerk < HWY_LC_MIN_ACC_GAIN);
          const bool pend = ok && nfoll >= 0;
          if (__ballot(pend) != 0) {  // wave-uniform: the new follower's braking
            const int gf = pend ? nfoll : 0;
            const double nf_pred_a = B::idm_free(p, sh.v[gf], sh.ts[gf], dl) - B::idm_gap(sh.x[gf], sh.v[gf], sh.c[gf], sh.s[gf], ex, ev, ec, es);
            if (pend) ok = !(nf_pred_a < -HWY_LC_MAX_BRAKING);
          }
          const u64 okm = __ballot(ok);
          if ((cl || cr) && 2 * d >= base && 2 * d < base + 64) bits = (int)(okm >> (2 * d - base)) & 3;
        }
        // side_lanes order is [left, right] and the loop does not break: right wins if both pass
        if (bits & 1) me.tgt = me.lane - 1;
        if (bits & 2) me.tgt = me.lane + 1;
      }
    } else
      if (decide) {
        me.timer = 0.0;
        // self_a: my IDM acceleration behind my current leader (old_preceding)
        const double self_a = free_self - (f_own >= 0 ? B::idm_gap(me.x, me.v, me.ch, me.sh, sh.x[f_own], sh.v[f_own],
                                                                  sh.c[f_own], sh.s[f_own]) : 0.0);
        for (int side = 0; side < 2; ++side) {  // side_lanes: [id-1], [id+1]  (road.py:200-211)
          const int cand = me.lane + (side == 0 ? -1 : 1);
          if (cand < 0 || cand >= p.L) continue;
          if (!B::reachable(p, cand, me.x, me.y)) continue;
          if (fabs(me.v) < 1) continue;
          // mobil(cand)  (behavior.py:265-324; route is None; POLITENESS == 0 so the followers' terms are
          // multiplied by 0.0 -- finite by construction -- and only the safety criterion needs new_following)
          int nprec, nfoll;
          if (!has_tie) B::neighbours_ranked(sh, cand, rank, &nprec, &nfoll);
          else B::neighbours_scan(p, sh, cand, i, me.x, &nprec, &nfoll);
          // mobil() is a pure predicate: safety (new follower's braking) AND incentive (my gain) -- evaluated
          // incentive first because it needs no pow() and rejects ~98% of the candidates
          const double self_pred_a = free_self - (nprec >= 0 ? B::idm_gap(me.x, me.v, me.ch, me.sh, sh.x[nprec], sh.v[nprec],
                                                                          sh.c[nprec], sh.s[nprec]) : 0.0);
          const double jerk = self_pred_a - self_a;
          if (jerk < HWY_LC_MIN_ACC_GAIN) continue;
          if (nfoll >= 0) {
            const double nf_pred_a = B::idm_free(p, sh.v[nfoll], sh.ts[nfoll], me.delta) -
                                     B::idm_gap(sh.x[nfoll], sh.v[nfoll], sh.c[nfoll], sh.s[nfoll], me.x, me.v, me.ch, me.sh);
            if (nf_pred_a < -HWY_LC_MAX_BRAKING) continue;
          }
          me.tgt = cand;
        }
      }
    // abort rule for ongoing lane changes (behavior.py:229-244): an ordered chain (Gauss-Seidel over Road.vehicles order).
    // A changer c (on its way to lane T since an earlier frame) aborts if ANOTHER vehicle r heading for T from a third lane -- with
    // the target r has when c acts: its current one for r before c in the list, the frame-start one for r after c -- is ahead of it
    // by less than the desired gap d*(c, r).
    {
      // A rival is ANOTHER vehicle on its way to another lane (with the target it had at the start of the frame or the one
      // it has now): with at most one such vehicle in the environment no link can block
      u64 mv[NW], cm[NW];
      B::block_ballot2(sh, active && (me.lane != tgt_old || me.lane != me.tgt), changer, sh.bal0, sh.bal2, ph0, ph2, mv, cm);
      int n_movers = 0;
      for (int w = 0; w < NW; ++w) n_movers += __popcll(mv[w]);
      const bool chain = n_movers > 1 && B::any_of(cm);  // block-uniform
#ifndef HWY_BLOCK_LITERAL_CHAIN
      if (chain && !has_tie) {
        // The chain per THREAD, in rank space (hwy_wave2.h has the argument in full; rounds 1-5 ran one link per changer here, every
        // link a workgroup barrier and a desired gap for all threads: 18 of 222 us at 1024 x 201).
        //  * a rival must be AHEAD and closer than d*, and d* <= 10 + 1.5 v + v (v + 5) / (2 sqrt(ab)) for every possible rival as long
        //    as no vehicle of the environment drives backwards or sideways faster than 5 m/s (checked: otherwise the bound is
        //    infinite): in the rank order of the snapshot a changer walks the members of "heading for T" ahead of it and stops at the
        //    first one beyond that bound -- usually the very first;
        //  * the links only interact through ABORTS, and an abort can only REMOVE a rival (its target becomes its own lane): a
        //    blocking rival that is itself an EARLIER changer counts only while it has not aborted, every other one for good;
        //  * a link depends on earlier links only, so iterating "aborts = blocked by a rival that is not an earlier changer, or by
        //    an earlier changer that does not abort" from "nobody aborts" reaches the literal chain's result after (depth + 1)
        //    rounds: one workgroup ballot per round, two rounds when somebody aborts, one when nobody does.
        // (1) S_T (rank space) = the vehicles heading for lane T from another lane: every such vehicle ORs its rank bit into row T
        // (zeroed with the membership masks of section C); what a walker needs of a rival beyond the snapshot goes by index
        if (active && me.lane != me.tgt)
          __hip_atomic_fetch_or(&sh.amask[me.tgt][rank >> 6], (u64)1 << (rank & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sh.acode[i] = ((me.tgt != tgt_old) ? 1 : 0) | (changer ? 2 : 0);
        const bool sane = !__syncthreads_or(active && !(me.v * me.ch >= 0.0 && fabs(me.v * me.sh) <= 5.0));
        // d*(c, r) = 10 + 1.5 v + v dv / (2 sqrt(ab)) with dv = v (cc^2 + sc^2) - v_r (c_r cc + s_r sc) <= v + 5 + (rounding) when v_r c_r >= 0,
        // |v_r s_r| <= 5 (sane) and cc >= 0, v >= 0; 1e-6 relative + absolute on top of the bound, far above any rounding in d*
        const double bound = (sane && me.v >= 0.0 && me.ch >= 0.0)
                                 ? (HWY_DISTANCE_WANTED + me.v * HWY_TIME_WANTED + me.v * (me.v + 5.0) * 0.12909944487358055) * (1.0 + 1e-6) + 1e-6
                                 : __builtin_inf();
        const u64 *prev = nullptr;  // the previous round's verdicts (index space: the ballot's own LDS slot); none in the first round
        for (;;) {  // block-uniform
          // (2) the walk: nearest member of S_T ahead first; beyond the bound everything farther is beyond it too
          bool fire = false;
          if (changer) {
            int cur = rank;
            for (;;) {
              int rr = -1;
              const int rw = cur >> 6, rb = cur & 63;
              for (int w = rw; w < NW; ++w) {
                u64 m = sh.amask[tgt_old][w];
                if (w == rw) m &= ~(((u64)2 << rb) - 1);  // ranks > cur  (2 << 63 wraps to 0 => all cleared)
                if (m) { rr = w * 64 + ctz64(m); break; }
              }
              if (rr < 0) break;
              const int j = sh.perm[rr];
              const double d = sh.x[j] - me.x;
              if (!(d < bound)) break;
              const int fl = sh.acode[j];
              // the target r shows to c: its current one if it comes before c in the list, else the frame-start one -- and a vehicle
              // that decided in this very frame headed nowhere with that one
              const bool valid = j < i || !(fl & 1);
              const double d_star = B::desired_gap(me.v, me.ch, me.sh, sh.v[j], sh.c[j], sh.s[j]);
              const bool blk = valid && (0 < d) && (d < d_star);
              // an earlier changer may abort itself: it blocks only while it has not
              const bool gone = j < i && (fl & 2) != 0 && prev != nullptr && ((prev[j >> 6] >> (j & 63)) & 1) != 0;
              if (blk && !gone) { fire = true; break; }
              cur = rr;
            }
          }
          // (3) the changers that abort, to the fixed point: the verdicts of a round are a workgroup ballot (its slot pair alternates,
          // so the previous round's words are still there to compare with and to read in the next walk)
          const u64 b = __ballot(fire);
          u64 *slot = sh.bal1 + ph1 * NW;
          ph1 ^= 1;
          if (lane_id == 0) slot[wave] = b;
          __syncthreads();
          bool same = true;
          for (int w = 0; w < NW; ++w) same = same && slot[w] == (prev ? prev[w] : (u64)0);
          prev = slot;
          if (same) break;
        }
#ifndef HWY_BLOCK_MUTANT_NO_ABORT  // (tests/test_wide_kernel.py: a build that never applies the verdict must fail the comparison)
        if ((prev[wave] >> lane_id) & 1) me.tgt = me.lane;  // abort
#endif
      } else
#endif
      if (chain) {  // equal x somewhere in the environment (rare): the literal chain, one link per changer
        for (int w = 0; w < NW; ++w) {
          u64 m = cm[w];
          while (m) {  // block-uniform loop
            const int ci = w * 64 + ctz64(m);
            m &= m - 1;
            const int Tc = sh.tgt[ci];  // the changer's target lane (unchanged so far this frame)
            const double xc = sh.x[ci], vc = sh.v[ci], cc = sh.c[ci], sc = sh.s[ci];
            // what vehicle ci reads from me: my target AFTER my act if I come before it in the list
            const int my_tgt_seen = (i < ci) ? me.tgt : tgt_old;
            bool blk = false;
            if (active && i != ci && me.lane != Tc && my_tgt_seen == Tc) {
              const double d = me.x - xc;
              const double d_star = B::desired_gap(vc, cc, sc, me.v, me.ch, me.sh);
              blk = (0 < d) && (d < d_star);
            }
            u64 bm[NW];
            B::block_ballot(sh, blk, sh.bal1, ph1, bm);
            if (i == ci && B::any_of(bm)) me.tgt = me.lane;  // abort
          }
        }
      }
    }

    wave_turn(turn);
    // ---- E. Road.act: low-level control (controller.py:89-133, behavior.py:104-137) ---------------------
    // tb = tan(beta) = 1/2 tan(steering) (see steer_tan_beta); acceleration command
    double tb = 0.0, accel = 0.0;
    if (controlled || drives) {
      const double inv_v = fast_rcp(not_zero(me.v));
      tb = B::steer_tan_beta(p, me.y, me.h, inv_v, me.tgt);
    }
    if (controlled) {
      accel = HWY_KP_A * (me.ts - me.v);  // speed_control (controller.py:189-198), not clipped
    } else if (drives) {
      accel = free_self - (f_own >= 0 ? B::idm_gap(me.x, me.v, me.ch, me.sh, sh.x[f_own], sh.v[f_own], sh.c[f_own],
                                                   sh.s[f_own]) : 0.0);
      if (me.lane != me.tgt) {
        int f2, r2;
        if (!has_tie) B::neighbours_ranked(sh, me.tgt, rank, &f2, &r2);
        else B::neighbours_scan(p, sh, me.tgt, i, me.x, &f2, &r2);
        const double a2 = free_self - (f2 >= 0 ? B::idm_gap(me.x, me.v, me.ch, me.sh, sh.x[f2], sh.v[f2], sh.c[f2],
                                                            sh.s[f2]) : 0.0);
        accel = (a2 < accel) ? a2 : accel;  // Python min(a, b)
      }
      accel = clipd(accel, -HWY_ACC_MAX, HWY_ACC_MAX);
    }

    // ---- F. Road.step: integrate (behavior.py:139-148, kinematics.py:130-177) --------------------------
    const double x_old = me.x;
    if (active) {
      if (idm) me.timer += p.dt;
      if (crashed0) {  // clip_actions: steering = 0 => tan(beta) = 0
        tb = 0.0;
        accel = -1.0 * me.v;
      }
      if (me.v > HWY_MAX_SPEED) accel = fmin(accel, 1.0 * (HWY_MAX_SPEED - me.v));
      else if (me.v < HWY_MIN_SPEED) accel = fmax(accel, 1.0 * (HWY_MIN_SPEED - me.v));
      // beta = atan(tb):  cos(beta) = 1/sqrt(1+tb^2), sin(beta) = tb*cos(beta);
      // cos(h+beta), sin(h+beta) by angle addition against the cached cos(h), sin(h)
      const double cb = fast_rsqrt(1.0 + tb * tb), sb = tb * cb;
      const double vx = me.v * (me.ch * cb - me.sh * sb), vy = me.v * (me.sh * cb + me.ch * sb);
      me.x += vx * p.dt;
      me.y += vy * p.dt;
      if (me.flags & HWY_F_HAS_IMPACT) {
        me.x += me.impx;
        me.y += me.impy;
        me.flags = (me.flags | HWY_F_CRASHED) & ~HWY_F_HAS_IMPACT;
        me.impx = me.impy = 0.0;
      }
      me.h += me.v * sb * (1.0 / (HWY_VEH_LENGTH / 2)) * p.dt;
      me.v += accel * p.dt;
      me.lane = B::closest_lane(p, me.x, me.y, me.h);  // on_state_update
      sincos_bounded(me.h, &me.sh, &me.ch);
    }

    wave_turn(turn);
    // ---- G. Road.step: collisions (road.py:477-481, objects.py:92-138) ------------------------------------
    if (i < 2) sh.gmax[i] = 0;  // (last read behind the second barrier of the previous frame's section G)
    __syncthreads();  // all reads of the frame-start snapshot are done
    publish<NW>(sh, me, active);
    if (all_check) {  // (block-uniform)
      {  // this frame's largest displacement along x and largest speed (reach_key): per wavefront, then one ds_max per wavefront
        const unsigned kd = HWY_WAVE_MAX_U32(active ? reach_key(me.x - x_old) : 0u), kv = HWY_WAVE_MAX_U32(active ? reach_key(me.v) : 0u);
        if (lane_id == 0) {
          __hip_atomic_fetch_max(&sh.gmax[0], kd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_max(&sh.gmax[1], kv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      // frame-start x, and the post-integration position and speed, once more in the RANK order of this frame's start: a step of
      // the walk below then reads ONE slot (rank + k) of four planes and the index table -- no dependent second round trip
      if (active) { sh.aux0[rank] = x_old; sh.rpx[rank] = me.x; sh.rpy[rank] = me.y; sh.rpv[rank] = me.v; }
      // per-vehicle verdict slots of the pair pass below
      sh.jmax[i] = -1;
      sh.hit[i] = 0;
    }
    __syncthreads();
    if (all_check) {
      // Full pairwise (highway-v0).  A pair can only collide if it is within ~5.5 m + |v| dt, i.e. among neighbours along the
      // road: every vehicle walks FORWARD in the rank order of this frame's start (each unordered pair is met once, from its
      // rear end), bounded by the frame-start distance (collision radius + the most two vehicles can move relative to each
      // other in one frame), and only COLLECTS the partners inside the reference's own pre-check sphere (objects.py:124-127) --
      // a dozen VALU instructions per candidate, one list per wavefront.  The list pass then runs, one PAIR per thread, the
      // provable-separation test and -- if any pair of the wavefront survives it -- the SAT; the verdicts meet per vehicle in
      // LDS, across wavefronts, hence the workgroup barrier between "highest partner with a pending impact" (ds_max: the
      // partner with the highest index is the last writer of `impact` in the reference's (i, j > i) loop) and the winner's
      // write of its translation.  (hwy_wave2.h has the one-wavefront version of the same walk; rounds 1-5 walked outward in
      // both directions through the index table and ran the separation test inside the walk, under divergence.)
      unsigned short *const plist = sh.plist[i >> 6];
      const int lane_id_ = i & 63;
      const double reach = reach_from_keys(sh.gmax[0], sh.gmax[1], p.dt);  // (see reach_key: this frame's actual maxima)
      const u64 below = ((u64)1 << lane_id_) - 1;
      int n_list = 0, k = 1;  // wave-uniform
      bool go = active, walking = true, any_impact = false;
      for (;;) {
        while (walking && n_list < 64) {
          // two walk steps per trip (k and k + 1): every LDS read of the trip is issued before anything depends on one (the
          // slots are clamped by the range alone; `go`, which depends on what is read, is folded in afterwards)
          bool keep[2];
          int q[2], rb[2];
          double x0[2], px[2], py[2], pv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            rb[u] = rank + (k + u);
            const int r = rb[u] < N ? rb[u] : 0;
            x0[u] = sh.aux0[r];  // frame-start x in rank order
            px[u] = sh.rpx[r]; py[u] = sh.rpy[r]; pv[u] = sh.rpv[r];
            q[u] = sh.perm[r];
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const bool near = !(fabs(x0[u] - x_old) > reach);
            go = go & (rb[u] < N) & near;
            const double dx = px[u] - me.x, dy = py[u] - me.y;
            const double lim = 5.5 + fmax(fabs(me.v), fabs(pv[u])) * p.dt;
            keep[u] = go & !(dx * dx + dy * dy > lim * lim);
          }
          k += 2;
          if (__ballot(go) == 0 || k > N) walking = false;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const u64 km = __ballot(keep[u]);
            if (km) {
              if (keep[u]) plist[n_list + __popcll(km & below)] = (unsigned short)(i | (q[u] << 8));
              n_list += __popcll(km);
            }
          }
        }
        // The list pass of this round: this wavefront's pairs (if it has any -- no other wavefront's data is waited for: the
        // bodies were published before the walk), then ONE barrier per round that also carries two bits per wavefront -- "I
        // have more to do" and "one of my pairs holds a pending impact" (a slot per wavefront, B::block_share) -- instead of
        // the three to four barriers per round of rounds 1-5; the winner's write of its translation, which needs every
        // wavefront's ds_max, runs behind that barrier and only in the rounds in which some pair of the WORKGROUP collides.
        const int count = n_list < 64 ? n_list : 64, left = n_list - count;  // left < 128
        int r = 0, a = 0, b = 0;
        double tx = 0.0, ty = 0.0;
        if (count > 0) {  // wave-uniform
          const int pair = lane_id_ < count ? (int)plist[lane_id_] : -1;
          const int u0 = pair < 0 ? 0 : (pair & 255), u1 = pair < 0 ? 0 : (pair >> 8);  // (no pair: slot 0, discarded)
          a = u0 < u1 ? u0 : u1;  // a < b: the reference's `self` and `other`
          b = u0 < u1 ? u1 : u0;
          const Body A = B::body_of(sh, a), Bb = B::body_of(sh, b);
          const bool cand = pair >= 0 && !hwy::surely_apart(A, Bb, p.dt);
          if (__ballot(cand) != 0) {  // wave-uniform
            if (cand) {
              r = hwy::pair_collide(A, Bb, p.dt, &tx, &ty);
              if (r & 1) sh.hit[a] = sh.hit[b] = 1;
              if (r & 2) {  // "last pair in loop order wins" == the partner with the highest index
                __hip_atomic_fetch_max(&sh.jmax[a], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&sh.jmax[b], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
        }
        u64 rb[NW];
        B::block_share(sh, ((walking || left > 0) ? 1u : 0u) | (__ballot((r & 2) != 0) ? 2u : 0u), sh.bal1, ph1, rb);
        u64 round_bits = 0;
        for (int w = 0; w < NW; ++w) round_bits |= rb[w];
        if (r & 2) {  // (behind the barrier: every wavefront's ds_max has landed)
          if (sh.jmax[a] == b) { sh.aux1[a] = tx / 2; sh.ipy[a] = ty / 2; }
          if (sh.jmax[b] == a) { sh.aux1[b] = -tx / 2; sh.ipy[b] = -ty / 2; }
        }
        if (left > 0) {  // (wave-uniform; then count == 64)
          // the entries this pass did not take move to the front: read behind the SAT, not before it -- two registers less across the
          // routine the kernel's pressure peaks in; this wavefront's lanes run in lockstep and its LDS operations in order, so both
          // reads land before either write
          const int c0 = lane_id_ < left ? (int)plist[64 + lane_id_] : 0;
          const int c1 = 64 + lane_id_ < left ? (int)plist[128 + lane_id_] : 0;
          HWY_WAVEFRONT_FENCE();
          if (lane_id_ < left) plist[lane_id_] = (unsigned short)c0;
          if (64 + lane_id_ < left) plist[64 + lane_id_] = (unsigned short)c1;
        }
        n_list = left;
        if (round_bits & 2) any_impact = true;
        if (!(round_bits & 1)) break;  // block-uniform
      }
      if (any_impact) __syncthreads();  // (block-uniform: the translations written behind the last round's barrier)
      // my target speed from the slot this section's publish wrote it to (the same double): the list pass above -- the SAT -- is where
      // the kernel's register pressure peaks, and this way the pair is not held across it
      me.ts = sh.ts[i];
      if (active && sh.jmax[i] >= 0) {
        me.impx = sh.aux1[i];
        me.impy = sh.ipy[i];
        me.flags |= HWY_F_HAS_IMPACT;
      }
      if (active && sh.hit[i]) me.flags |= HWY_F_CRASHED;
    } else {
      // sparse checkers (highway-fast-v0: the ego only): for each checker c, thread q evaluates the pair
      // {c, q}; q applies it to itself (ascending c == loop order), c gathers from all q.
      for (int w = 0; w < NW; ++w) {
        u64 m = sh.chk[w];
        while (m) {  // block-uniform
          const int c = w * 64 + ctz64(m);
          m &= m - 1;
          int r = 0;
          double tx = 0, ty = 0;
          // conservative reject first (never drops a pair the exact pre-check of objects.py:124-127 keeps):
          // the SAT routine (inlined: ~220 instructions) is only reached by vehicles within ~5.5 m + |v| dt
          bool near = false;
          if (active && i != c) {
            const double dx = sh.x[c] - me.x, dy = sh.y[c] - me.y;
            const double lim = 5.5 + fmax(fabs(me.v), fabs(sh.v[c])) * p.dt;
            near = dx * dx + dy * dy <= lim * lim;
          }
          if (near && !B::surely_apart(sh, i < c ? i : c, i < c ? c : i, p.dt)) {
            const int a = i < c ? i : c, b = i < c ? c : i;
            r = B::pair_collide(sh, a, b, p.dt, &tx, &ty);
            const bool i_check = (me.flags & HWY_F_CHECK_COLLISIONS) != 0;
            if (!i_check) {  // my only partners are the checkers
              if (r & 2) {
                me.impx = (i == a) ? tx / 2 : -tx / 2;
                me.impy = (i == a) ? ty / 2 : -ty / 2;
                me.flags |= HWY_F_HAS_IMPACT;
              }
              if (r & 1) me.flags |= HWY_F_CRASHED;
            }
            sh.aux0[i] = tx;
            sh.aux1[i] = ty;
          }
          u64 wm[NW], im[NW];
          const u64 bw = __ballot((r & 2) != 0), bi = __ballot((r & 1) != 0);
          if (lane_id == 0) { sh.bal0[wave] = bw; sh.bal1[wave] = bi; }
          __syncthreads();
          for (int ww = 0; ww < NW; ++ww) { wm[ww] = sh.bal0[ww]; im[ww] = sh.bal1[ww]; }
          if (i == c) {
            if (B::any_of(im)) me.flags |= HWY_F_CRASHED;
            for (int ww = NW - 1; ww >= 0; --ww)
              if (wm[ww]) {
                const int q = ww * 64 + msb64(wm[ww]);  // last partner in loop order
                const double qx = sh.aux0[q], qy = sh.aux1[q];
                me.impx = (c < q) ? qx / 2 : -qx / 2;
                me.impy = (c < q) ? qy / 2 : -qy / 2;
                me.flags |= HWY_F_HAS_IMPACT;
                break;
              }
          }
          __syncthreads();
        }
      }
    }
  }  // frames

  // ---- H. observe / reward / done (abstract.py:277-281) ---------------------------------------------------
  if (p.full_step) {
    // sh.x/y/v/c/s already hold the post-integration state of the last frame (collisions do not move anyone)
    if (p.n_frames == 0) {
      __syncthreads();
      publish<NW>(sh, me, active);
      __syncthreads();
    }
    observe_env<NW>(p, sh, e, me, true, eo);
  }
  me.rank = rank;  // the hint the next step verifies
  store_vehicle<NW>(p, e, me, false);
}

template <int NW, int WPE>
__global__ void __launch_bounds__(NW * 64, WPE) hwy_step_kernel(const StepParams p) {
  __shared__ typename EnvBlock<NW>::Shared sh;
  block_policy_step<NW>(p, sh, blockIdx.x, blockIdx.x);
}

// hwy_rollout_device on the workgroup kernel: p.k_steps policy steps per workgroup in one launch (hwy_wave.h:
// hwy_rollout_wave_kernel has the argument).
template <int NW, int WPE>
__global__ void __launch_bounds__(NW * 64, WPE) hwy_rollout_kernel(const StepParams p) {
  __shared__ typename EnvBlock<NW>::Shared sh;
  const int e = blockIdx.x;
  for (int k = 0; k < p.k_steps; ++k) {  // block-uniform
    HWY_RELOAD_STEP_PARAMS(pk, p);  // a fresh, opaque view of the arguments per step: nothing stays live -- spilled -- across steps
    block_policy_step<NW>(pk, sh, e, k * pk.num_envs + e);
    __syncthreads();  // the next step's loads and LDS writes follow this step's stores and LDS reads
  }
}

// ---- self-test kernel for hwy_math.h (hwy_debug_math) ------------------------------------------------
__device__ inline double math_probe(int op, double x) {
  double s, c;
  switch (op) {
    case 0: return log_pos(x);
    case 1: return exp_bounded(x);
    case 2: sincos_bounded(x, &s, &c); return s;
    case 3: sincos_bounded(x, &s, &c); return c;
    case 4: return asin_bounded(x);
    case 5: return fast_rcp(x);
    case 6: return fast_rsqrt(x);
    case 8: return atan_fd(x);
    case 9: return atan2_bounded(x, 0.75);
    case 10: return atan2_bounded(0.5, x);
    case 11: return atan2_bounded(-0.5, x);
    // the paired forms (hwy_math.h: two evaluations sharing every coefficient) against the scalar ones: ops 20 + 2 k / 21 + 2 k
    // return the first / second result of routine k for the arguments (x, f(x)) / (f(x), x), f = another point of the domain
    case 20: { double a, b; log_pos2(x, 0.37 * x + 0.011, a, b); return a; }
    case 21: { double a, b; log_pos2(0.37 * x + 0.011, x, a, b); return b; }
    case 22: { double a, b; exp_bounded2(x, 0.5 * x - 1.25, a, b); return a; }
    case 23: { double a, b; exp_bounded2(0.5 * x - 1.25, x, a, b); return b; }
    case 24: { double a, b, c2, d; sincos_bounded2(x, 1.0 - x, &a, &b, &c2, &d); return a; }
    case 25: { double a, b, c2, d; sincos_bounded2(x, 1.0 - x, &a, &b, &c2, &d); return b; }
    case 26: { double a, b, c2, d; sincos_bounded2(1.0 - x, x, &a, &b, &c2, &d); return c2; }
    case 27: { double a, b, c2, d; sincos_bounded2(1.0 - x, x, &a, &b, &c2, &d); return d; }
    case 28: { double a, b; asin_bounded2(x, -0.6 * x, a, b); return a; }
    case 29: { double a, b; asin_bounded2(-0.6 * x, x, a, b); return b; }
    case 30: { double a, b; fast_rcp2(x, 3.0 * x, a, b); return a; }
    case 31: { double a, b; fast_rcp2(3.0 * x, x, a, b); return b; }
    case 32: { double a, b; fast_rsqrt2(x, 3.0 * x, a, b); return a; }
    case 33: { double a, b; fast_rsqrt2(3.0 * x, x, a, b); return b; }
    // the collision walk's reach bound: 40 = the wavefront's maximum of reach_key (call with whole wavefronts), 41 = the double the
    // key is rounded up to
    case 40: return (double)HWY_WAVE_MAX_U32(reach_key(x));
    case 41: return __hiloint2double((int)(reach_key(x) + 1u), 0);
    default: return wrap_to_pi(x);
  }
}
}  // namespace hwy

// hwy_wave.h -- the fused policy-step kernel specialised for N <= 64 vehicles: ONE 64-wide
// wavefront per environment (the headline highway-fast-v0 4096 x 51 case).
//
// Same semantics as the generic workgroup kernel in hwy_device.h (the path for N > 128; hwy_wave2.h
// runs 64 < N <= 128 on one wavefront with two vehicles per thread), but built around what a single
// CDNA4 wavefront can do without barriers:
//
//   * cross-vehicle reads with a wave-uniform source index (rank counting, the lane-change abort
//     chain, ego-vs-all collision checks, observation keys) use v_readlane -- the value lands in
//     SGPRs and feeds the f64 compare directly, no LDS round trip and no barrier;
//   * the sort by longitudinal position is carried from frame to frame and merely verified (a
//     ds_permute exchange of x; a readlane counting pass when the order changed); each vehicle ORs
//     its rank bit into the masks of the one or two lanes it is on (ds_or_b64: rank-space
//     membership masks), and fetches the masks of its own / left / right / target lane;
//   * the only LDS-resident data is the frame snapshot stored IN RANK ORDER, so a neighbour found
//     by bit-scan (rank) is fetched with one gather, not rank -> index -> data;
//   * the two MOBIL candidates are evaluated side by side (independent chains) instead of one
//     after the other, and the expensive follower-safety test runs only for candidates that pass
//     the (pow-free) incentive test.
#pragma once

#include "hwy_device.h"

namespace hwy {

// ---- wave-level data movement ------------------------------------------------------------------
// value of lane `src` (wave-uniform index) -> every lane, via v_readlane_b32 (SGPR result)
__device__ inline int wave_bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ inline double wave_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// my value -> lane `dst` (per-lane destination, a permutation), via ds_permute_b32
__device__ inline int wave_send_i(int v, int dst) { return __builtin_amdgcn_ds_permute(dst << 2, v); }

// A second view `q` of the by-value kernel argument `p`, whose fields are loaded after this statement (see the epilogue
// of hwy_step_wave_kernel).  The kernel must have StepParams as its only argument (offset 0 of the segment).
#ifndef HWY_RELOAD_PARAMS
#define HWY_RELOAD_PARAMS(q, p)                                                                      \
  auto kernarg_ = __builtin_amdgcn_kernarg_segment_ptr(); /* constant address space: scalar loads */ \
  asm volatile("" : "+s"(kernarg_));                                                                 \
  const StepParams &q = *(const StepParams *)kernarg_
#endif

// The five values are in registers here: every load that produces one of them has been ISSUED above this point (and is waited for
// here, together), whatever branches follow -- the compiler may neither sink such a load into the block that uses it nor hoist the
// branch's own load above them.
#ifndef HWY_ISSUED_TOGETHER
#define HWY_ISSUED_TOGETHER(a, b, c, d, e_) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e_))
#endif

// The kernel-argument segment (StepParams: ~760 bytes, a dozen cache lines, freshly written by the host for every dispatch) is read
// on demand by scalar loads scattered over the prologue, each followed by its own wait: a chain of four or five cache misses in a
// row before the first vehicle plane is even requested -- and every wavefront of the launch walks it at the same moment.  One dword
// of every line, requested back to back and waited for once, turns the chain into a single miss; the loads that follow hit the
// scalar cache.  Measured (profiles/r05_history.md): headline 41.3 -> 40.7 us with it; NOT for the road-network / intersection
// kernels, whose argument structs are 2-3 x larger (lane table, route table): touching them whole cost +2 us, their first 760 bytes
// nothing either way.
#if defined(HWY_NO_KERNARG_TOUCH) && !defined(HWY_KERNARG_TOUCH)
#define HWY_KERNARG_TOUCH(T) ((void)0)
#endif
#ifndef HWY_KERNARG_TOUCH
#define HWY_KERNARG_TOUCH(T)                                                                              \
  do {                                                                                                    \
    const unsigned *ka_ = (const unsigned *)__builtin_amdgcn_kernarg_segment_ptr();                       \
    unsigned acc_ = 0;                                                                                    \
    _Pragma("unroll") for (unsigned o_ = 0; o_ < sizeof(T); o_ += 64) acc_ ^= ka_[o_ / 4];               \
    asm volatile("" : : "s"(acc_));                                                                       \
  } while (0)
#endif

// One wavefront == one workgroup: LDS instructions of a wavefront execute in order, so a ds_read issued after a ds_write
// of the same wavefront sees it without any wait or s_barrier.  Only the COMPILER must keep the order (and the CPU
// emulation of tests/emu, whose 64 threads are separate fibers, needs a real rendezvous).
#ifndef HWY_WAVE_LDS_FENCE
#define HWY_WAVE_LDS_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
#endif

struct WaveShared {
  // frame snapshot in RANK order (slot r == r-th vehicle along the road)
  double x[64], v[64], c[64], s[64], lr[64];  // lr = log(v/v0), see EnvBlock::idm_log_ratio
  int idx[64];
  u64 lane_mask[HWY_MAX_LANES + 2];  // rank-space membership mask of road lane L in slot L+1; 0 at both ends
  // per-vehicle state touched once per frame (slot i is private to thread i: no barrier involved).
  // Keeping it here instead of in registers trims ~10 VGPRs off the frame loop's loop-carried set.
  double timer[64], ts[64], delta[64], impx[64], impy[64];
  // post-integration bodies by vehicle index (full pairwise collisions only)
  double nx[64], ny[64], nv[64], nc[64], ns[64];
};

// front / rear ranks on a lane from its rank-space membership mask; -1 if none
__device__ inline void mask_neighbours(u64 m, int r, int *front, int *rear) {
  const u64 above = m & ~(((u64)2 << r) - 1);  // ranks > r  (2<<63 wraps to 0 => everything cleared)
  const u64 below = m & (((u64)1 << r) - 1);   // ranks < r
  *front = above ? ctz64(above) : -1;
  *rear = below ? msb64(below) : -1;
}

// Road.neighbour_vehicles literal scan for the equal-x case, reading bodies from registers
__device__ inline void wave_neighbours_scan(const StepParams &p, double myx, double x, double y, int self, int Lq,
                                            int *front, int *rear) {
  int f = -1, b = -1;
  double s_front = 0, s_rear = 0;
  for (int j = 0; j < p.N; ++j) {  // wave-uniform j
    const double s_v = wave_bcast(x, j), lat_v = wave_bcast(y, j) - Lq * p.lane_width;
    if (j == self) continue;
    if (!(fabs(lat_v) <= p.lane_width / 2 + 1.0 && -5.0 <= s_v && s_v < p.road_length + 5.0)) continue;
    if (myx <= s_v && (f < 0 || s_v <= s_front)) { s_front = s_v; f = j; }
    if (s_v < myx && (b < 0 || s_v > s_rear)) { s_rear = s_v; b = j; }
  }
  *front = f;
  *rear = b;
}

// Rank of every vehicle along the road (0 = smallest x; equal x ordered by list index), kept from frame to
// frame and from step to step and merely re-validated.
__device__ inline void wave_update_rank(double x, bool active, int N, int &rank, bool &has_tie) {
  const int i = threadIdx.x;
  // The order along the road changes in ~1 frame out of 5 (measured), and then only by adjacent
  // swaps, so the rank of the previous frame is kept in a register and merely VERIFIED: every vehicle
  // sends its x to lane `rank` and to lane `rank-1` (ds_permute); lane r then holds x of rank r and
  // of rank r+1 and checks x[r] < x[r+1].  Only if some pair is out of order (or equal) does the
  // wave fall back to the exact counting pass.
  bool recount = false;
  {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const double x_r = __hiloint2double(wave_send_i(hi, rank), wave_send_i(lo, rank));
    const double x_r1 = __hiloint2double(wave_send_i(hi, rank - 1), wave_send_i(lo, rank - 1));
    recount = __ballot(i < N - 1 && !(x_r < x_r1)) != 0;
    if (recount) {  // (uniform over the wave)
      // Cheap repair first: if the strict inversions are pairwise disjoint adjacent pairs (the usual case: one
      // overtake somewhere on the road), swapping the two ranks of every inverted pair restores the order.
      // Bit r of `inv` (rank space) says ranks r and r+1 are inverted; the repaired order is verified again and
      // anything else (overlapping inversions, equal x) falls through to the counting pass.
      const u64 inv = __ballot(i < N - 1 && x_r > x_r1);
      if (inv != 0 && (inv & (inv << 1)) == 0) {
        const bool up = (inv >> rank) & 1;
        const bool down = rank > 0 && ((inv >> (rank - 1)) & 1);
        rank += up ? 1 : (down ? -1 : 0);  // idle lanes hold ranks >= N: their bits are never set
        const double y_r = __hiloint2double(wave_send_i(hi, rank), wave_send_i(lo, rank));
        const double y_r1 = __hiloint2double(wave_send_i(hi, rank - 1), wave_send_i(lo, rank - 1));
        recount = __ballot(i < N - 1 && !(y_r < y_r1)) != 0;
      }
    }
    if (!recount) has_tie = false;  // strictly increasing => all x distinct
  }
  if (recount) {  // wave-uniform
    // Counting pass on the HIGH 32 bits of x first: for non-negative doubles the high word orders like
    // an integer and separates any two positions more than ~1 mm apart (2^-20 relative), at a third of
    // the issue cost of f64 compares.  If two high words coincide (or some x is negative) the wave
    // redoes the pass with exact f64 compares.
    const int hi = __double2hiint(x);
    int cnt_lt = 0, cnt_le = 0;
    for (int j = 0; j < N; ++j) {
      const int hj = wave_bcast_i(hi, j);
      cnt_lt += (hj < hi) ? 1 : 0;
      cnt_le += (hj <= hi) ? 1 : 0;
    }
    if (__ballot(active && ((cnt_le - cnt_lt) > 1 || hi < 0)) != 0) {  // ambiguous: exact pass
      cnt_lt = cnt_le = 0;
      for (int j = 0; j < N; ++j) {
        const double xj = wave_bcast(x, j);
        cnt_lt += (xj < x) ? 1 : 0;
        cnt_le += (xj <= x) ? 1 : 0;
      }
    }
    const bool tie = active && (cnt_le - cnt_lt) > 1;
    has_tie = __ballot(tie) != 0;
    rank = active ? cnt_lt : i;  // idle lanes keep their own slot so the permutation stays a bijection
    if (has_tie) {  // equal x: order by list index, like a stable sort (rare)
      for (int j = 0; j < N; ++j) {
        const double xj = wave_bcast(x, j);
        rank += (active && xj == x && j < i) ? 1 : 0;
      }
    }
  }
}

// KinematicObservation + reward + done for every agent, all cross-lane reads through readlane.
// BY_RANK: `rank` is the exact rank along the road of every vehicle (wave_update_rank on the CURRENT positions).
// eo: the row of the output planes (obs, reward, flags, info) this environment writes -- e, or k * num_envs + e for step k of a
// multi-step launch (hwy_rollout_device).
// env_time: the environment's clock at the start of the step (requested with the state by the caller: a load here, at the end of
// the wavefront's life, is a round trip nothing hides); only read when write_reward is set
template <bool BY_RANK>
__device__ inline void observe_wave(const StepParams &p, int e, int eo, const Veh &me, bool write_reward, int rank = 0,
                                    double env_time = 0.0) {
  typedef EnvBlock<1> B;
  const int i = threadIdx.x;
  const bool active = i < p.N;
  const int V = p.V, F = p.F;
  for (int a = 0; a < p.A; ++a) {
    const int ia = p.agent_index[a];
    const double ex = wave_bcast(me.x, ia), ey = wave_bcast(me.y, ia), ev = wave_bcast(me.v, ia);
    const double ec = wave_bcast(me.ch, ia), es = wave_bcast(me.sh, ia);
    const double dxe = me.x - ex, dye = me.y - ey;
    const double d_lane = me.x - ex;
    // norm < distance  <=>  dx^2 + dy^2 < distance^2 when distance^2 is exact (200^2 is): sqrt is monotone and
    // correctly rounded, so the compare can be done on the squares
    const bool elig = active && i != ia && (dxe * dxe + dye * dye < p.perception * p.perception) &&
                      ((p.flags & HWY_C_OBS_SEE_BEHIND) || (-2 * HWY_VEH_LENGTH < d_lane));
    // (sort=False, observation.py:245: every eligible object gets the same key, so the stable order is the list order)
    const double key = elig ? ((p.flags & HWY_C_OBS_UNSORTED) ? 0.0 : fabs(d_lane)) : __builtin_inf();
    const int n_elig = __popcll(__ballot(elig));
    const int m = n_elig < V - 1 ? n_elig : V - 1;
    // stable sort position among the eligible (ties keep list order)
    int pos = 0;
    if (BY_RANK && !(p.flags & (HWY_C_OBS_SEE_BEHIND | HWY_C_OBS_UNSORTED))) {  // wave-uniform
      // Everything eligible BEHIND the observer is closer than 2*LENGTH, so the eligible split into `near`
      // (key < 2*LENGTH: a handful, ordered by explicit compares) and `far` (all in front by at least 2*LENGTH,
      // after every near one, and among themselves ordered like x, i.e. like their rank along the road).
      const bool near = elig && key < 2 * HWY_VEH_LENGTH;
      const bool far = elig && !near;
      const u64 near_m = __ballot(near);
      const u64 far_r = __ballot(wave_send_i(far ? 1 : 0, rank) != 0);  // rank space
      for (u64 em = near_m; em; em &= em - 1) {  // wave-uniform
        const int k = ctz64(em);
        const double kk = wave_bcast(key, k);
        pos += ((kk < key) || (kk == key && k < i)) ? 1 : 0;
      }
      const int lower = __popcll(far_r & (((u64)1 << rank) - 1));
      pos = far ? __popcll(near_m) + lower : pos;
    } else {
      // (only eligible vehicles can precede an eligible one: walk the set bits of the ballot)
      for (u64 em = __ballot(elig); em; em &= em - 1) {  // wave-uniform
        const int k = ctz64(em);
        const double kk = wave_bcast(key, k);
        pos += ((kk < key) || (kk == key && k < i)) ? 1 : 0;
      }
    }
    if (p.obs && p.obs_type != HWY_OBS_KINEMATICS) observe_grid<1>(p, e, a, me, ex, ey, ev, ec, es, eo);
    if (p.obs && p.obs_type == HWY_OBS_KINEMATICS) {
      float *out = p.obs + ((size_t)eo * p.A + a) * (size_t)(V * F);
      const int row = (i == ia) ? 0 : (elig && pos < V - 1 ? pos + 1 : -1);
      if (p.obs_std5) {  // wave-uniform
        // features == [presence, x, y, vx, vy] (KinematicObservation's default): the same arithmetic as the generic loop
        // below (Vehicle.to_dict, origin subtraction, lmap with the host-computed reciprocal, clip), written out per feature
        // instead of five trips through the feature switch
        if (active && row >= 0) {
          double fx = me.x, fy = me.y, fvx = me.v * me.ch, fvy = me.v * me.sh;
          if (row > 0 && !(p.flags & HWY_C_OBS_ABSOLUTE)) { fx -= ex; fy -= ey; fvx -= ev * ec; fvy -= ev * es; }
          if (p.flags & HWY_C_OBS_NORMALIZE) {
            const bool clip = (p.flags & HWY_C_OBS_CLIP) != 0;
            if (p.rx0 > -__builtin_inf()) { fx = lmap_inv(fx, p.rx0, p.inv_rx, -1.0, 1.0); fx = clip ? clipd(fx, -1.0, 1.0) : fx; }
            if (p.ry0 > -__builtin_inf()) { fy = lmap_inv(fy, p.ry0, p.inv_ry, -1.0, 1.0); fy = clip ? clipd(fy, -1.0, 1.0) : fy; }
            if (p.rvx0 > -__builtin_inf()) { fvx = lmap_inv(fvx, p.rvx0, p.inv_rvx, -1.0, 1.0); fvx = clip ? clipd(fvx, -1.0, 1.0) : fvx; }
            if (p.rvy0 > -__builtin_inf()) { fvy = lmap_inv(fvy, p.rvy0, p.inv_rvy, -1.0, 1.0); fvy = clip ? clipd(fvy, -1.0, 1.0) : fvy; }
          }
          float *o5 = out + row * 5;
          o5[0] = 1.0f; o5[1] = (float)fx; o5[2] = (float)fy; o5[3] = (float)fvx; o5[4] = (float)fvy;
        }
      } else if (active && row >= 0) {
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
            const double ir = fid == HWY_FEAT_X ? p.inv_rx : fid == HWY_FEAT_Y ? p.inv_ry : fid == HWY_FEAT_VX ? p.inv_rvx : p.inv_rvy;
            if (r0 > -__builtin_inf()) {
              val = lmap_inv(val, r0, ir, -1.0, 1.0);
              if (p.flags & HWY_C_OBS_CLIP) val = clipd(val, -1.0, 1.0);
            }
          }
          out[row * F + f] = (float)val;
        }
      }
      for (int t = i; t < V * F; t += 64)
        if (t / F > m) out[t] = 0.0f;
    }
    if (write_reward && i == ia) {
      const bool crashed = (me.flags & HWY_F_CRASHED) != 0;
      const bool on_road = fabs(me.y - me.lane * p.lane_width) <= p.lane_width / 2 + 0.0 && -5.0 <= me.x &&
                           me.x < p.road_length + 5.0;
      const double forward_speed = me.v * me.ch;
      // true divisions like the reference, the oracle and the workgroup kernel (one thread per agent and step): the host-computed
      // reciprocals are for the f32 observation features only -- a reward at its maximum must come out as exactly 1.0
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
        const bool term = crashed || ((p.flags & HWY_C_OFFROAD_TERMINAL) && !on_road);
        const double t = env_time + p.policy_dt;
        const bool trunc = t >= p.duration;
        p.st.time[e] = t;
        p.terminated[eo] = term ? 1 : 0;
        p.truncated[eo] = trunc ? 1 : 0;
        if (p.autoreset) p.st.done[e] = (term || trunc) ? 1 : 0;
      }
    }
  }
}

// =============================================================================================
// FULL_SCAN = false builds the kernel without the full-pairwise window scan (every checker is handled by
// the checker loop, which is correct for any set of checkers but O(#checkers)); the engine launches it for
// highway-fast-v0 style configs (HWY_C_EGO_ONLY_COLLISIONS), where it keeps the frame loop at 173 VGPRs.
// One policy step of environment e by its wavefront; eo = row of the action / output planes (see observe_wave).
template <bool FULL_SCAN>
__device__ __forceinline__ void wave_policy_step(const StepParams &p, WaveShared &sh, const int e, const int eo) {
  typedef EnvBlock<1> B;
  const int i = threadIdx.x;
  const int N = p.N;
  const bool active = i < N;

  // Everything the step needs from HBM is requested BEFORE anything is waited for: the environment's done flag and clock, the
  // meta-actions (lane a fetches agent a's) and the vehicle's planes -- one round trip instead of three (flag -> state; the clock
  // at the end of the step, where the wavefront could not retire before it came back).  All wavefronts of a launch start together,
  // so nothing else hides these latencies.  An environment that is re-spawned instead (7 % of the steady state) has fetched its
  // old state for nothing.
  int done_flag = p.autoreset ? (int)p.st.done[e] : 0;
  double env_time = p.st.time[e];
  int act_lane = (p.actions && i < p.A) ? p.actions[(size_t)eo * p.A + i] : HWY_IDLE;
  Veh me;
  load_vehicle<1>(p, e, me);
  HWY_ISSUED_TOGETHER(done_flag, env_time, act_lane, me.x, me.timer);  // (keeps the requests above the branch)
  // ---- auto-reset: re-spawn instead of stepping (rare; shares the generic helpers) -------------
  if (done_flag) {
    me = Veh{};
    const uint32_t episode = p.st.episode[e] + 1u;
    spawn_env<1>(p, sh.x, sh.v, e, p.rp.base_seed + (uint64_t)e, episode, me);
    observe_wave<false>(p, e, eo, me, false);
    store_vehicle<1>(p, e, me);
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

  WaveTurn turn;
  wave_turn_init(turn, p.prio_shift, p.prio_recip);
  const bool controlled = active && (me.flags & HWY_F_CONTROLLED);
  const bool idm = active && !controlled;
  int agent = 0, act0 = HWY_IDLE;
  for (int a = 0; a < p.A; ++a) {  // wave-uniform
    const int act_a = wave_bcast_i(act_lane, a);
    if (controlled && p.agent_index[a] == i) { agent = a; act0 = HWY_ACTION_TO_ALL(p.action_set, act_a); }
  }
  sh.timer[i] = me.timer; sh.ts[i] = me.ts; sh.delta[i] = me.delta; sh.impx[i] = me.impx; sh.impy[i] = me.impy;
  const bool i_check = (me.flags & HWY_F_CHECK_COLLISIONS) != 0;
  const u64 chk = __ballot(active && i_check);
  const bool all_check = FULL_SCAN && __popcll(chk) == N;

  // position along the road, carried from frame to frame AND from step to step (hint in the packed word;
  // idle lanes keep their own slot so that the permutation stays a bijection)
  int rank = active ? me.rank : i;
  bool has_tie = false;  // two vehicles share the same x (=> literal neighbour scans)
  double inv_v0 = 0.0;   // 1 / |nz(clip(target speed))|: a per-STEP invariant of IDM's (v / v0)^delta
  for (int fr = 0; fr < p.n_frames; ++fr) {
    wave_turn(turn);
    // ---- A. meta-action (abstract.py:294-304 -> controller.py:295-315) ------------------------------
    if (fr == 0 && p.actions && controlled) {
      const int act = act0;
      if (act == HWY_FASTER || act == HWY_SLOWER) {
        const double xs = (me.v - p.target_speeds[0]) / (p.target_speeds[p.n_ts - 1] - p.target_speeds[0]);
        int idx = (int)clipd(rint(xs * (p.n_ts - 1)), 0.0, (double)(p.n_ts - 1)) + (act == HWY_FASTER ? 1 : -1);
        idx = idx < 0 ? 0 : (idx > p.n_ts - 1 ? p.n_ts - 1 : idx);
        me.sidx = idx;
        sh.ts[i] = p.target_speeds[idx];
      } else if (act == HWY_LANE_LEFT || act == HWY_LANE_RIGHT) {
        int id = me.tgt + (act == HWY_LANE_RIGHT ? 1 : -1);
        id = id < 0 ? 0 : (id > p.L - 1 ? p.L - 1 : id);
        if (B::reachable(p, id, me.x, me.y)) me.tgt = id;
      }
    }

    // ---- C. rank along the road -----------------------------------------------------------------------
    wave_update_rank(me.x, active, N, rank, has_tie);
    // lane membership (AbstractLane.on_lane, margin 1) -> bits -> sent to lane `rank` -> ballots
    const bool inr = active && (-5.0 <= me.x) && (me.x < p.road_length + 5.0);
    int bits = 0;
    for (int L = 0; L < p.L; ++L)
      bits |= (inr && (fabs(me.y - L * p.lane_width) <= p.lane_width / 2 + 1.0)) ? (1 << L) : 0;
    // the masks live in a small LDS table (slot L+1 = lane L, zero slots on both sides for "no such lane") so that each
    // vehicle fetches own / left / right / target lane with four LDS reads instead of four 64-bit select chains; every vehicle
    // ORs its rank bit into the masks of the lanes it is on (one or two: ds_or_b64) -- rounds 1-3: a ds_permute of the bits to
    // the lane `rank` and one ballot + select per road lane
    HWY_WAVE_LDS_FENCE();  // previous frame's mask reads are complete
    if (i < p.L + 2) sh.lane_mask[i] = 0;
    HWY_WAVE_LDS_FENCE();
    for (int b_ = bits; b_; b_ &= b_ - 1)
      __hip_atomic_fetch_or(&sh.lane_mask[__builtin_ctz(b_) + 1], (u64)1 << rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // frame-start snapshot, stored in rank order (with each vehicle's IDM log speed ratio)
    if (fr == 0) inv_v0 = B::idm_inv_v0(p, sh.ts[i]);  // (after the meta-action of frame 0: the target speed is fixed for the step)
    const double log_ratio = active ? B::idm_log_ratio_inv(me.v, inv_v0) : 0.0;  // egos and wrecks can be followers too
    HWY_WAVE_LDS_FENCE();  // previous frame's gathers are complete
    if (active) {
      sh.x[rank] = me.x; sh.v[rank] = me.v; sh.c[rank] = me.ch; sh.s[rank] = me.sh; sh.lr[rank] = log_ratio;
      sh.idx[rank] = i;
    }
    HWY_WAVE_LDS_FENCE();
    const u64 m_own = sh.lane_mask[me.lane + 1];  // (the side lanes' masks are read by the compacted MOBIL tasks)
    const u64 m_tgt = sh.lane_mask[me.tgt + 1];

    wave_turn(turn);
    // ---- D. Road.act: lane-change policy (behavior.py:219-263) ----------------------------------------
    const bool crashed0 = (me.flags & HWY_F_CRASHED) != 0;
    const bool drives = idm && !crashed0;
    const int tgt_old = me.tgt;
    const bool changer = drives && me.lane != me.tgt;
    const double timer = sh.timer[i];
    const bool decide = drives && me.lane == me.tgt && (HWY_LC_DELAY < timer);
    // IDMVehicle timer: reset by a decision (behavior.py:248), then += dt in step (behavior.py:147)
    sh.timer[i] = idm ? (decide ? 0.0 : timer) + p.dt : timer;
    // neighbours: ranks on own / left / right / target lane (bit scans), or the literal scan on ties
    int fo = -1, ro = -1, fl = -1, rl = -1, frt = -1, rrt = -1, ft = -1, rt_ = -1;
    const bool left_ok = me.lane - 1 >= 0, right_ok = me.lane + 1 < p.L;
    const double delta = sh.delta[i];
    const double free_self = B::idm_free_from_log(log_ratio, delta);
    const bool moving = !(fabs(me.v) < 1);
    const bool cl = decide && left_ok && B::reachable(p, me.lane - 1, me.x, me.y) && moving;
    const bool cr = decide && right_ok && B::reachable(p, me.lane + 1, me.x, me.y) && moving;
    bool ok_l = false, ok_r = false;
    double gap_own = 0.0, gap_new = 0.0;  // gap_new: the IDM gap term towards the leader on the side MOBIL picks in this frame
#ifndef HWY_WAVE_MOBIL_PER_THREAD
    if (!has_tie) {  // wave-uniform
      // Every vehicle needs its leader on its own lane and -- while it changes lanes -- on its target lane in every frame.  MOBIL
      // (behavior.py:265-324: both side lanes' leaders and followers, two more gap terms, the follower's braking) is only
      // evaluated by a vehicle whose timer has run out: once per second, i.e. by a fifth (highway-fast-v0) or a fifteenth
      // (highway-v0) of the traffic in a given frame.  Rounds 1-5 evaluated both candidate lanes of EVERY vehicle side by side in
      // every frame, predicated; now the (vehicle, side) pairs that do decide are COMPACTED: decider number d hands (free-road term,
      // own-lane acceleration, delta, rank | lane | side bits) over through LDS, thread t evaluates side t & 1 of decider t >> 1
      // from the rank-ordered snapshot (the decider's own body is there too) -- one candidate per thread instead of two per
      // thread for everybody -- and the verdicts come back as a ballot, the chosen side's gap term through LDS (it is the target-lane
      // term of the low-level control of this very frame).  Same operations on the same values: bit-identical to the per-thread
      // form (-DHWY_WAVE_MOBIL_PER_THREAD builds it; tests/test_wide_kernel.py::test_compacted_mobil_identity).
      mask_neighbours(m_own, rank, &fo, &ro);
      mask_neighbours(m_tgt, rank, &ft, &rt_);
      const int g_fo = fo < 0 ? 0 : fo;
      gap_own = fo >= 0 ? B::idm_gap(me.x, me.v, me.ch, me.sh, sh.x[g_fo], sh.v[g_fo], sh.c[g_fo], sh.s[g_fo]) : 0.0;
      const double self_a = free_self - gap_own;
      const u64 dm = __ballot(cl || cr);
      if (dm) {  // wave-uniform
        int *const word = reinterpret_cast<int *>(sh.nc);  // (the post-integration bodies only live inside section G)
        const int d = __popcll(dm & (((u64)1 << i) - 1));
        HWY_WAVE_LDS_FENCE();
        if (cl || cr) {
          sh.nx[d] = free_self; sh.ny[d] = self_a; sh.nv[d] = delta;
          word[d] = rank | (me.lane << 8) | (cl ? 1 << 16 : 0) | (cr ? 1 << 17 : 0);
        }
        HWY_WAVE_LDS_FENCE();
        const int n_tasks = 2 * __popcll(dm);
        for (int base = 0; base < n_tasks; base += 64) {  // wave-uniform: one pass unless more than 32 vehicles decide at once
          const int t = base + i;
          const bool tv = t < n_tasks;
          const int dd = tv ? t >> 1 : 0, side = t & 1;
          const int w = word[dd];
          const int rk = w & 255, ln = (w >> 8) & 255;
          const bool en = tv && ((w >> (16 + side)) & 1);
          const double fs = sh.nx[dd], sa = sh.ny[dd], dl = sh.nv[dd];
          const u64 m = sh.lane_mask[ln + (side ? 2 : 0)];
          int f, r;
          mask_neighbours(m, rk, &f, &r);
          const double ex = sh.x[rk], ev = sh.v[rk], ec = sh.c[rk], es = sh.s[rk];
          const int gf = f < 0 ? 0 : f;
          const double gap = f >= 0 ? B::idm_gap(ex, ev, ec, es, sh.x[gf], sh.v[gf], sh.c[gf], sh.s[gf]) : 0.0;
          // incentive (jerk = self_pred_a - self_a, POLITENESS == 0), then safety: the new follower must not have to brake harder
          // than LANE_CHANGE_MAX_BRAKING_IMPOSED -- only for candidates that passed the (pow-free) incentive test
          bool ok = en && !(((fs - gap) - sa) < HWY_LC_MIN_ACC_GAIN);
          const bool pend = ok && r >= 0;
          if (__ballot(pend) != 0) {  // wave-uniform
            const int rf = pend ? r : 0;
            const double lr_f = sh.lr[rf];
            const double g = B::idm_gap(sh.x[rf], sh.v[rf], sh.c[rf], sh.s[rf], ex, ev, ec, es);
            // a_f = 3 (1 - E) - g with E = exp(delta * lr_f) >= 0, so a_f <= 3 - g: g beyond 5 is unsafe whatever E is; and a
            // follower below its target speed (lr_f < 0, delta > 0) has E <= 1 (+ an ulp), so a_f >= -g: g below 2 is safe
            // whatever E is.  Both with a 1e-6 margin, far above any rounding of the three operations involved -- the verdicts
            // are the ones the full expression gives.  The exp runs only if some pending task falls between the two.
            const bool sure_unsafe = g > HWY_COMFORT_ACC_MAX + HWY_LC_MAX_BRAKING + 1e-6;
            const bool sure_safe = lr_f < 0.0 && dl > 0.0 && g <= HWY_LC_MAX_BRAKING - 1e-6;
            bool safe = sure_safe;
            if (__ballot(pend && !sure_unsafe && !sure_safe) != 0) {  // wave-uniform
              const double a_f = B::idm_free_from_log(lr_f, dl) - g;
              safe = !(a_f < -HWY_LC_MAX_BRAKING);
            }
            if (pend) ok = safe;
          }
          const u64 okm = __ballot(ok);
          sh.ns[i] = gap;
          HWY_WAVE_LDS_FENCE();
          if ((cl || cr) && 2 * d >= base && 2 * d < base + 64) {
            const int bits = (int)(okm >> (2 * d - base)) & 3;
            ok_l = (bits & 1) != 0;
            ok_r = (bits & 2) != 0;
            if (bits) gap_new = sh.ns[2 * d - base + (ok_r ? 1 : 0)];  // right wins if both pass
          }
          HWY_WAVE_LDS_FENCE();
        }
      }
    } else
#endif
    {
      if (!has_tie) {
        mask_neighbours(m_own, rank, &fo, &ro);
        mask_neighbours(sh.lane_mask[me.lane], rank, &fl, &rl);
        mask_neighbours(sh.lane_mask[me.lane + 2], rank, &frt, &rrt);
        mask_neighbours(m_tgt, rank, &ft, &rt_);
      } else {
        // literal scans return vehicle INDICES; convert to ranks through the (just written) table
        int a, b;
        wave_neighbours_scan(p, me.x, me.x, me.y, i, me.lane, &a, &b);
        fo = a; ro = b;
        wave_neighbours_scan(p, me.x, me.x, me.y, i, left_ok ? me.lane - 1 : me.lane, &a, &b);
        fl = a; rl = b;
        wave_neighbours_scan(p, me.x, me.x, me.y, i, right_ok ? me.lane + 1 : me.lane, &a, &b);
        frt = a; rrt = b;
        wave_neighbours_scan(p, me.x, me.x, me.y, i, me.tgt, &a, &b);
        ft = a; rt_ = b;
        // index -> rank: rank_of[j] == the rank lane j computed
        const int r_fo = fo, r_fl = fl, r_fr = frt, r_ft = ft, r_rl = rl, r_rr = rrt;
        int rk;
        fo = fl = frt = ft = rl = rrt = -1;
        for (int j = 0; j < N; ++j) {
          rk = wave_bcast_i(rank, j);
          fo = (r_fo == j) ? rk : fo; fl = (r_fl == j) ? rk : fl; frt = (r_fr == j) ? rk : frt;
          ft = (r_ft == j) ? rk : ft; rl = (r_rl == j) ? rk : rl; rrt = (r_rr == j) ? rk : rrt;
        }
      }
      // gather the leaders' bodies (rank-ordered snapshot => one LDS trip); all issued together
      const int g_fo = fo < 0 ? 0 : fo, g_fl = fl < 0 ? 0 : fl, g_fr = frt < 0 ? 0 : frt;
      const double fo_x = sh.x[g_fo], fo_v = sh.v[g_fo], fo_c = sh.c[g_fo], fo_s = sh.s[g_fo];
      const double fl_x = sh.x[g_fl], fl_v = sh.v[g_fl], fl_c = sh.c[g_fl], fl_s = sh.s[g_fl];
      const double fr_x = sh.x[g_fr], fr_v = sh.v[g_fr], fr_c = sh.c[g_fr], fr_s = sh.s[g_fr];
      gap_own = fo >= 0 ? B::idm_gap(me.x, me.v, me.ch, me.sh, fo_x, fo_v, fo_c, fo_s) : 0.0;
      // MOBIL (behavior.py:265-324), both candidates side by side.  jerk = self_pred_a - self_a with
      // self_* = free_self - gap_*  (POLITENESS == 0: the followers' terms are multiplied by 0.0)
      const double self_a = free_self - gap_own;
      const double gap_l = fl >= 0 ? B::idm_gap(me.x, me.v, me.ch, me.sh, fl_x, fl_v, fl_c, fl_s) : 0.0;
      const double gap_r = frt >= 0 ? B::idm_gap(me.x, me.v, me.ch, me.sh, fr_x, fr_v, fr_c, fr_s) : 0.0;
      ok_l = cl && !(((free_self - gap_l) - self_a) < HWY_LC_MIN_ACC_GAIN);
      ok_r = cr && !(((free_self - gap_r) - self_a) < HWY_LC_MIN_ACC_GAIN);
      // safety: the new follower must not have to brake harder than LANE_CHANGE_MAX_BRAKING_IMPOSED.
      // Evaluated only for candidates that passed the (pow-free) incentive test, one side per pass (a
      // vehicle that needs both sides checked -- rare -- takes a second pass); the follower's log speed
      // ratio comes from the snapshot, so the test costs one exp and one gap term.
      {
        bool pend_l = ok_l && rl >= 0, pend_r = ok_r && rrt >= 0;
        while (__ballot(pend_l || pend_r) != 0) {  // wave-uniform
          const bool pend = pend_l || pend_r;
          const bool left = pend_l;
          const int rf = pend ? (left ? rl : rrt) : 0;
          const double lr_f = sh.lr[rf];
          const double g = B::idm_gap(sh.x[rf], sh.v[rf], sh.c[rf], sh.s[rf], me.x, me.v, me.ch, me.sh);
          const bool sure_unsafe = g > HWY_COMFORT_ACC_MAX + HWY_LC_MAX_BRAKING + 1e-6;
          const bool sure_safe = lr_f < 0.0 && delta > 0.0 && g <= HWY_LC_MAX_BRAKING - 1e-6;
          bool safe = sure_safe;
          if (__ballot(pend && !sure_unsafe && !sure_safe) != 0) {  // wave-uniform
            const double a_f = B::idm_free_from_log(lr_f, delta) - g;
            safe = !(a_f < -HWY_LC_MAX_BRAKING);
          }
          if (pend) {
            if (left) { ok_l = safe; pend_l = false; } else { ok_r = safe; pend_r = false; }
          }
        }
      }
      gap_new = ok_r ? gap_r : gap_l;  // (only read when MOBIL picks a side)
    }
    // side_lanes order is [left, right] and the loop does not break: right wins if both pass
    if (ok_l) me.tgt = me.lane - 1;
    if (ok_r) me.tgt = me.lane + 1;
    // abort rule for ongoing lane changes (behavior.py:229-244): an ordered chain over Road.vehicles -- a changer c (on its
    // way to lane T since an earlier frame) aborts if ANOTHER vehicle r heading for T from a third lane, with the target r shows
    // when c acts (its current one for r before c in the list, the frame-start one after c), is ahead of c by less than the
    // desired gap d*(c, r).  Evaluated per THREAD in rank space, without a loop over the changers (derivation and the bound's
    // proof obligations: hwy_wave2.h section D; rounds 1-4 ran one link per changer here, ~25 instructions each and ~60 more
    // with a rival -- 17 links per step on the SIMDs that end the headline launch):
    //  * every mover ORs its rank bit into the mask row of ITS target lane (the membership masks are dead since the neighbour
    //    scans; the next frame's snapshot zeroes them again) and leaves "decided in this frame | changer << 1" in its rank slot;
    //  * a changer walks the members of its target lane's row AHEAD of it, nearest first, and stops at the first one beyond
    //    bound >= d*(c, any sane rival) -- usually the very first;
    //  * blocking rivals that are EARLIER changers only count while they do not abort themselves: ballots to the fixed point
    //    (unique: a link depends on earlier links only).
    {
      const u64 cm = __ballot(changer);
      // with a single vehicle on its way to another lane (the changer itself) no link can block
      if (cm && __popcll(__ballot(active && (me.lane != tgt_old || me.lane != me.tgt))) > 1) {  // wave-uniform
        int *const sbits = reinterpret_cast<int *>(sh.nx);  // (the post-integration bodies only live inside section G)
        HWY_WAVE_LDS_FENCE();  // the masks' readers of this frame and section G of the previous one are done
        if (i < p.L + 2) sh.lane_mask[i] = 0;
        HWY_WAVE_LDS_FENCE();
        sbits[rank] = ((me.tgt != tgt_old) ? 1 : 0) | (changer ? 2 : 0);
        if (active && me.lane != me.tgt)
          __hip_atomic_fetch_or(&sh.lane_mask[me.tgt + 1], (u64)1 << rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool sane = __ballot(active && !(me.v * me.ch >= 0.0 && fabs(me.v * me.sh) <= 5.0)) == 0;
        HWY_WAVE_LDS_FENCE();
        const u64 row = sh.lane_mask[tgt_old + 1];
        u64 rem = changer ? (row & ~(((u64)2 << rank) - 1)) : 0;  // ranks above mine (2 << 63 wraps to 0)
        u64 bc = 0;        // earlier changers (index space) that block me unless they abort
        bool fixed = false;  // blocked for good
        // d*(c, r) = 10 + 1.5 v + v dv / (2 sqrt(ab)), dv <= v + 5 for a sane rival when v >= 0 and cos h >= 0; 1e-6 relative +
        // absolute on top, far above any rounding in d* (tests/test_wide_kernel.py::test_abort_chain_window_bound_...)
        const double bound = (sane && me.v >= 0.0 && me.ch >= 0.0)
                                 ? (HWY_DISTANCE_WANTED + me.v * HWY_TIME_WANTED + me.v * (me.v + 5.0) * 0.12909944487358055) * (1.0 + 1e-6) + 1e-6
                                 : __builtin_inf();
        while (__ballot(rem != 0) != 0) {  // wave-uniform
          const bool go = rem != 0;
          const int rr = go ? ctz64(rem) : 0;
          rem &= rem - 1;
          const double xr = sh.x[rr], vr = sh.v[rr], cr_ = sh.c[rr], sr = sh.s[rr];
          const int ir = sh.idx[rr], fl_r = sbits[rr];
          const double d = xr - me.x;
          const bool inside = go && d < bound;
          // a vehicle later in the list shows its frame-start target -- and one that decided in this very frame headed nowhere
          const bool valid = inside && (ir < i || !(fl_r & 1));
          const double d_star = B::desired_gap(me.v, me.ch, me.sh, vr, cr_, sr);
          const bool blk = valid && (0 < d) && (d < d_star);
          const bool cond = ir < i && (fl_r & 2) != 0;  // an earlier changer: it may abort
          fixed = fixed || (blk && !cond);
          bc |= (blk && cond) ? ((u64)1 << ir) : 0;
          rem = ((go && !inside) || fixed) ? 0 : rem;
        }
        if (__ballot(fixed || bc != 0) != 0) {  // wave-uniform
          u64 ab = 0;
          for (;;) {
            const u64 nab = __ballot(fixed || (bc & ~ab) != 0);
            if (nab == ab) break;
            ab = nab;
          }
#ifndef HWY_WAVE_MUTANT_NO_ABORT  // (tests/test_wide_kernel.py: a build that never applies the verdict must fail the comparison)
          if ((ab >> i) & 1) me.tgt = me.lane;  // abort
#endif
        }
      }
    }

    wave_turn(turn);
    // ---- E. Road.act: low-level control ----------------------------------------------------------------
    const double inv_v = fast_rcp(not_zero(me.v));
    double tb = B::steer_tan_beta(p, me.y, me.h, inv_v, me.tgt);
    double accel = free_self - gap_own;
    if (drives && me.lane != me.tgt) {
      // leader on the target lane.  For a vehicle that was already changing lanes m_tgt is that lane's
      // mask; for one that decided just now the target is the left/right lane evaluated above.
      // (decided just now: the gap term towards the new target lane's leader is the one MOBIL's incentive test evaluated -- 0.0
      //  without a leader, and free_self - 0.0 == free_self)
      double a2 = free_self - gap_new;
      if (me.tgt == tgt_old) {
        a2 = free_self;
        if (ft >= 0) a2 = free_self - B::idm_gap(me.x, me.v, me.ch, me.sh, sh.x[ft], sh.v[ft], sh.c[ft], sh.s[ft]);
      }
      accel = (a2 < accel) ? a2 : accel;  // Python min(a, b)
    }
    accel = clipd(accel, -HWY_ACC_MAX, HWY_ACC_MAX);
    accel = controlled ? HWY_KP_A * (sh.ts[i] - me.v) : accel;  // speed_control (controller.py:189-198), not clipped

    // ---- F. Road.step: integrate -------------------------------------------------------------------------
    const double x_old = me.x;
    {
      // clip_actions (kinematics.py:155-168): a crashed vehicle has steering 0 (tan(beta) = 0), accel = -speed
      tb = crashed0 ? 0.0 : tb;
      accel = crashed0 ? -1.0 * me.v : accel;
      accel = (me.v > HWY_MAX_SPEED) ? fmin(accel, 1.0 * (HWY_MAX_SPEED - me.v))
                                     : ((me.v < HWY_MIN_SPEED) ? fmax(accel, 1.0 * (HWY_MIN_SPEED - me.v)) : accel);
      const double cb = fast_rsqrt(1.0 + tb * tb), sb = tb * cb;
      const double vx = me.v * (me.ch * cb - me.sh * sb), vy = me.v * (me.sh * cb + me.ch * sb);
      me.x += vx * p.dt;
      me.y += vy * p.dt;
      if (me.flags & HWY_F_HAS_IMPACT) {
        me.x += sh.impx[i];
        me.y += sh.impy[i];
        me.flags = (me.flags | HWY_F_CRASHED) & ~HWY_F_HAS_IMPACT;
        sh.impx[i] = sh.impy[i] = 0.0;
      }
      me.h += me.v * sb * (1.0 / (HWY_VEH_LENGTH / 2)) * p.dt;
      me.v += accel * p.dt;
      me.lane = B::closest_lane(p, me.x, me.y, me.h);
      sincos_bounded(me.h, &me.sh, &me.ch);
    }

    wave_turn(turn);
    // ---- G. Road.step: collisions (road.py:477-481, objects.py:92-138) -----------------------------------
    const Body mine{me.x, me.y, me.v, me.ch, me.sh};
    if (all_check) {
      // Full pairwise (highway-v0).  A pair can only collide if it is within ~5.5 m + |v| dt, i.e. among neighbours along
      // the road: every vehicle walks FORWARD in the rank order established at the start of this frame (each unordered pair is
      // met once, from its rear end; rounds 1-3 walked both ways and dropped half of what they met), and stops when the
      // partner's FRAME-START distance exceeds the collision radius plus the most two vehicles can have moved relative to each
      // other within one frame.  Everything within reach is visited, so the result equals the full loop; "last pair in loop
      // order wins" == the partner with the highest index.
      sh.nx[i] = me.x; sh.ny[i] = me.y; sh.nv[i] = me.v; sh.nc[i] = me.ch; sh.ns[i] = me.sh;
      // this frame's largest displacement along x and largest speed (hwy_device.h: reach_key, wave_max_u32)
      const unsigned key_d = HWY_WAVE_MAX_U32(active ? reach_key(me.x - x_old) : 0u), key_v = HWY_WAVE_MAX_U32(active ? reach_key(me.v) : 0u);
      HWY_WAVE_LDS_FENCE();
      // The walk only COLLECTS the partners inside the reference's own pre-check sphere (objects.py:124-127), a dozen VALU
      // instructions per candidate; the list pass then runs, one PAIR per thread, the provable-separation test and -- if any
      // pair of the wavefront survives it -- the SAT, and the verdicts meet per slot in LDS (crashed flags, the highest partner
      // slot with a pending impact, that pair's translation).  The rank-ordered snapshot of this frame (v, lr) is dead here
      // and holds the pair list and the per-slot results.
      int *const jmax = reinterpret_cast<int *>(sh.lr), *const hit = jmax + 64;
      unsigned short *const plist = reinterpret_cast<unsigned short *>(sh.v);  // 256 entries: lower slot | higher slot << 8
      jmax[i] = -1;
      hit[i] = 0;
      // radius + relative motion from the ACTUAL maxima of this frame (hwy_device.h: reach_from_keys; rounds 2-5 assumed 50 m/s
      // and a 3 m impact -- 21.5 m -- and fell back to the all-pairs loop for anything faster)
      const double reach = reach_from_keys(key_d, key_v, p.dt);
      const u64 below = ((u64)1 << i) - 1;
      int n_list = 0, k = 1;  // wave-uniform
      bool go_b = active, walking = true;
      while (walking || n_list) {
        while (walking && n_list < 64) {
          // two walk steps per trip; the slots are clamped by the range alone so that every LDS read of the trip is issued
          // before anything depends on one (a conditional read is a branch with its own round trip)
          bool keep[2];
          int q[2];
          double x0[2], px[2], py[2], pv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int rb = rank + k + u, r = rb < N ? rb : 0;
            q[u] = sh.idx[r];
            x0[u] = sh.x[r];  // sh.x: frame-start x in rank order
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) { px[u] = sh.nx[q[u]]; py[u] = sh.ny[q[u]]; pv[u] = sh.nv[q[u]]; }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            go_b = go_b & (rank + k + u < N) & !(fabs(x0[u] - x_old) > reach);
            const double dx = px[u] - me.x, dy = py[u] - me.y;
            const double lim = 5.5 + fmax(fabs(me.v), fabs(pv[u])) * p.dt;
            keep[u] = go_b & !(dx * dx + dy * dy > lim * lim);
          }
          k += 2;
          if (__ballot(go_b) == 0 || k > N) walking = false;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const u64 km = __ballot(keep[u]);
            if (km) {
              if (keep[u]) plist[n_list + __popcll(km & below)] = (unsigned short)(i < q[u] ? (i | (q[u] << 8)) : (q[u] | (i << 8)));
              n_list += __popcll(km);
            }
          }
        }
        const int count = n_list < 64 ? n_list : 64, left = n_list - count;  // left < 128
        HWY_WAVE_LDS_FENCE();
        const int pair = i < count ? (int)plist[i] : -1;
        const int c0 = i < left ? (int)plist[count + i] : 0, c1 = 64 + i < left ? (int)plist[count + 64 + i] : 0;
        const int a = pair < 0 ? 0 : (pair & 255), b = pair < 0 ? 0 : (pair >> 8);  // a < b: the reference's `self` and `other`
        int r = 0;
        double tx = 0.0, ty = 0.0;
        const Body A{sh.nx[a], sh.ny[a], sh.nv[a], sh.nc[a], sh.ns[a]}, Bb{sh.nx[b], sh.ny[b], sh.nv[b], sh.nc[b], sh.ns[b]};
        const bool cand = pair >= 0 && !surely_apart(A, Bb, p.dt);
        if (__ballot(cand) != 0) {  // wave-uniform
          if (cand) {
            r = pair_collide(A, Bb, p.dt, &tx, &ty);
            if (r & 1) hit[a] = hit[b] = 1;
            if (r & 2) {  // "last pair in loop order wins" == the partner with the highest index
              __hip_atomic_fetch_max(&jmax[a], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_fetch_max(&jmax[b], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
          HWY_WAVE_LDS_FENCE();
          if (r & 2) {
            if (jmax[a] == b) { sh.impx[a] = tx / 2; sh.impy[a] = ty / 2; }
            if (jmax[b] == a) { sh.impx[b] = -tx / 2; sh.impy[b] = -ty / 2; }
          }
        }
        if (i < left) plist[i] = (unsigned short)c0;
        if (64 + i < left) plist[64 + i] = (unsigned short)c1;
        n_list = left;
        HWY_WAVE_LDS_FENCE();
      }
      if (active && jmax[i] >= 0) me.flags |= HWY_F_HAS_IMPACT;
      if (active && hit[i]) me.flags |= HWY_F_CRASHED;
    } else {
      // sparse checkers (highway-fast-v0: the ego only)
      u64 cm = chk;
      while (cm) {  // wave-uniform
        const int c = ctz64(cm);
        cm &= cm - 1;
        // the checker's body to every lane THROUGH LDS (one lane writes, all read the same address): the launch is bound by the VALU
        // port and a v_readlane is a VALU instruction, an LDS access is not -- ten v_readlane per frame were
        // 1.1 % of the headline launch (40.75 -> 40.30 us, profiles/r05_history.md section 8).  (sh.nx is free in this branch.)
        HWY_WAVE_LDS_FENCE();
        if (i == c) { sh.nx[0] = me.x; sh.nx[1] = me.y; sh.nx[2] = me.v; sh.nx[3] = me.ch; sh.nx[4] = me.sh; }
        HWY_WAVE_LDS_FENCE();
        const Body other{sh.nx[0], sh.nx[1], sh.nx[2], sh.nx[3], sh.nx[4]};
        int r = 0;
        double tx = 0, ty = 0;
        if (active && i != c) {
          const double dx = other.x - me.x, dy = other.y - me.y;
          const double lim = 5.5 + fmax(fabs(me.v), fabs(other.v)) * p.dt;
          if (dx * dx + dy * dy <= lim * lim) {
            const bool i_first = i < c;
            const Body A = select_body(i_first, mine, other), Bb = select_body(i_first, other, mine);
            if (!surely_apart(A, Bb, p.dt)) {
              r = pair_collide(A, Bb, p.dt, &tx, &ty);
              if (!i_check) {  // my only partners are the checkers (ascending c == loop order)
                if (r & 2) {
                  sh.impx[i] = i_first ? tx / 2 : -tx / 2;
                  sh.impy[i] = i_first ? ty / 2 : -ty / 2;
                  me.flags |= HWY_F_HAS_IMPACT;
                }
                if (r & 1) me.flags |= HWY_F_CRASHED;
              }
            }
          }
        }
        const u64 wm = __ballot((r & 2) != 0), im = __ballot((r & 1) != 0);
        if (wm | im) {  // wave-uniform: the checker gathers from its partners
          const int q = wm ? msb64(wm) : 0;  // last partner in loop order
          const double qx = wave_bcast(tx, q), qy = wave_bcast(ty, q);
          if (i == c) {
            if (im) me.flags |= HWY_F_CRASHED;
            if (wm) {
              sh.impx[i] = (c < q) ? qx / 2 : -qx / 2;
              sh.impy[i] = (c < q) ? qy / 2 : -qy / 2;
              me.flags |= HWY_F_HAS_IMPACT;
            }
          }
        }
      }
    }
  }  // frames

  // ---- H. observe / reward / done ---------------------------------------------------------------------------
  // The epilogue reads its parameters (output pointers, observation ranges, reward weights: ~70 SGPRs worth)
  // again from the kernel-argument segment, through a pointer the compiler cannot see through, instead of
  // keeping them live -- i.e. spilled to VGPR lanes and re-read with v_readlane -- across the frame loop.
  HWY_RELOAD_PARAMS(q, p);
  if (q.full_step) wave_update_rank(me.x, active, N, rank, has_tie);  // positions moved in the last frame
  // (state stores before or after the observation: 40.8 / 40.6 us, within the noise -- profiles/r05_history.md)
  if (q.full_step) observe_wave<true>(q, e, eo, me, true, rank, env_time);
  me.rank = rank;
  me.timer = sh.timer[i]; me.ts = sh.ts[i]; me.delta = sh.delta[i]; me.impx = sh.impx[i]; me.impy = sh.impy[i];
  store_vehicle<1>(q, e, me, false);
}

template <int WPE, bool FULL_SCAN>
__global__ void __launch_bounds__(64, WPE) hwy_step_wave_kernel(const StepParams p) {
  __shared__ WaveShared sh;
  // which environment a workgroup steps is free (environments are independent): the dispatcher places workgroup b on the same
  // SIMD launch after launch, so a table b -> environment is a PLACEMENT of the environments on the SIMDs (hwy_set_block_order)
  HWY_KERNARG_TOUCH(StepParams);
  const int e = p.block_env ? (int)p.block_env[blockIdx.x] : (int)blockIdx.x;
  wave_policy_step<FULL_SCAN>(p, sh, e, e);
}

// hwy_rollout_device: p.k_steps consecutive policy steps of every environment in ONE launch (actions of step k in row
// k * num_envs + e of the action plane, outputs likewise).  Each step is the code of the one-step kernel, state through the
// same HBM planes (a wavefront re-reads what it has just written), auto-reset in between like in consecutive launches: the results
// are those of k_steps launches, bit for bit.  What the launch saves is everything a launch pays once: the dispatch, the
// wait for the slowest of 1024 SIMDs at the end of EVERY step (the expensive paths hit different SIMDs in different steps, so
// over k steps the loads even out) and the empty issue slots while the first loads and the last stores are in flight.
template <int WPE, bool FULL_SCAN>
__global__ void __launch_bounds__(64, WPE) hwy_rollout_wave_kernel(const StepParams p) {
  __shared__ WaveShared sh;
  HWY_KERNARG_TOUCH(StepParams);
  const int e = blockIdx.x;
  for (int k = 0; k < p.k_steps; ++k) {  // wave-uniform
    // a fresh, opaque view of the kernel arguments per step: nothing of a step's parameter set (or what was derived from it)
    // stays live -- in SGPRs spilled to VGPR lanes -- across the steps
    HWY_RELOAD_PARAMS(pk, p);
    wave_policy_step<FULL_SCAN>(pk, sh, e, k * pk.num_envs + e);
    HWY_WAVE_LDS_FENCE();
    __threadfence_block();  // the next step's loads follow this step's stores (same wavefront, same addresses)
  }
}

// Reset / observe kernels for N <= 64 reuse the generic ones (not on the per-step path).

}  // namespace hwy

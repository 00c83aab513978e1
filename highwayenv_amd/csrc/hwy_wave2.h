// hwy_wave2.h -- the fused policy-step kernel for 64 < N <= 64 K vehicles on ONE wavefront per environment: every thread
// carries K vehicles (K = 2: BASELINE config 3, highway-v0 with 101 vehicles; vehicle v lives in slot h = v >> 6 of thread v & 63).
//
// Why: the workgroup kernel of hwy_device.h steps such an environment with ceil(N / 64) wavefronts joined by LDS and ~8 workgroup
// barriers per frame.  At config 3's 1024 environments per GPU that is two wavefronts per SIMD, each of them waiting half of its
// cycles (s_waitcnt / s_barrier: profiles/r03_history.md) on dependent f64 chains nobody else fills.  Here the environment is ONE
// wavefront again -- every cross-vehicle exchange is a ballot, a v_readlane or an in-order LDS access of the same wavefront, no
// barrier anywhere (hwy_wave.h) -- and the parallelism the second wavefront used to bring comes back as instruction-level
// parallelism: the K vehicles of a thread are independent chains the scheduler interleaves, so the wavefront fills its own
// latencies.  Semantics, arithmetic (the same per-vehicle device functions, the same source expressions) and data layout are those
// of the other two straight-road kernels; the frame loop below follows hwy_wave.h section by section.
//
// Conventions: l = threadIdx.x (0..63); vehicle index vi(h) = 64 h + l; masks over vehicles / ranks are K words of 64 bits
// (bit v of the set = bit v & 63 of word v >> 6), so "ascending vehicle index" = ascending word, then ascending bit.
#pragma once

#include "hwy_wave.h"

namespace hwy {

// ---- paired forms of the per-vehicle routines (hwy_math.h: the two vehicles of a thread share every coefficient and their
//      chains are interleaved by construction); statement by statement the scalar ones, bit-identical results -------------------
__device__ inline void wrap_to_pi2(double x0, double x1, double &o0, double &o1) {
  const double a0 = x0 + HWY_PI, a1 = x1 + HWY_PI;
  double m0 = a0, m1 = a1;
  if (!(a0 >= 0 && a0 < 2 * HWY_PI && a1 >= 0 && a1 < 2 * HWY_PI)) {  // (py_mod_pos returns an already reduced argument as it is)
    m0 = py_mod_pos(a0, 2 * HWY_PI);
    m1 = py_mod_pos(a1, 2 * HWY_PI);
  }
  o0 = m0 - HWY_PI;
  o1 = m1 - HWY_PI;
}
// EnvBlock::steer_tan_beta for two vehicles
__device__ inline void steer_tan_beta2(const StepParams &p, double y0, double h0, double iv0, int tgt0, double y1, double h1, double iv1,
                                       int tgt1, double &o0, double &o1) {
  const double lat0 = y0 - tgt0 * p.lane_width, lat1 = y1 - tgt1 * p.lane_width;
  const double a0 = clipd((-HWY_KP_LATERAL * lat0) * iv0, -1.0, 1.0), a1 = clipd((-HWY_KP_LATERAL * lat1) * iv1, -1.0, 1.0);
  const double s45 = 0.7071067811865476;  // sin(pi/4) rounded up: |a| >= s45 => |asin a| >= pi/4 (clipped)
  double as0, as1;
  asin_bounded2(a0, a1, as0, as1);
  const double hr0 = a0 >= s45 ? HWY_PI / 4 : (a0 <= -s45 ? -HWY_PI / 4 : clipd(as0, -HWY_PI / 4, HWY_PI / 4));
  const double hr1 = a1 >= s45 ? HWY_PI / 4 : (a1 <= -s45 ? -HWY_PI / 4 : clipd(as1, -HWY_PI / 4, HWY_PI / 4));
  double wr0, wr1;
  wrap_to_pi2(hr0 - h0, hr1 - h1, wr0, wr1);
  const double c0 = HWY_KP_HEADING * wr0, c1 = HWY_KP_HEADING * wr1;
  const double w0 = clipd((HWY_VEH_LENGTH / 2 * iv0) * c0, -1.0, 1.0), w1 = clipd((HWY_VEH_LENGTH / 2 * iv1) * c1, -1.0, 1.0);
  const double tan_max = 1.7320508075688767;  // tan(MAX_STEERING_ANGLE = fl(pi/3)) in f64
  const double w20 = 1 - w0 * w0, w21 = 1 - w1 * w1;
  double rs0, rs1;
  fast_rsqrt2(w20 <= 1e-12 ? 1.0 : w20, w21 <= 1e-12 ? 1.0 : w21, rs0, rs1);
  const double ts0 = (w20 <= 1e-12) ? copysign(tan_max, w0) : clipd((2 * w0) * rs0, -tan_max, tan_max);
  const double ts1 = (w21 <= 1e-12) ? copysign(tan_max, w1) : clipd((2 * w1) * rs1, -tan_max, tan_max);
  o0 = 0.5 * ts0;
  o1 = 0.5 * ts1;
}

template <int K>
struct WideShared {
  static constexpr int NV = 64 * K;
  // frame snapshot in RANK order (slot r == r-th vehicle along the road)
  double x[NV], v[NV], c[NV], s[NV], lr[NV];
  int idx[NV];
  int sbits[NV];                          // scratch in rank order: the abort chain's codes / the observation's classes
  u64 lane_mask[HWY_MAX_LANES + 2][K];    // rank-space membership mask of road lane L in row L+1; 0 in rows 0 and L+1
  // collision translations by vehicle index: written by the thread that evaluated the winning pair, read by the owner
  double impx[NV], impy[NV];
  // post-integration bodies by vehicle index (full pairwise collisions), their verdict slots and the pair list (a ring)
  double nx[NV], ny[NV], nv[NV], nc[NV], ns[NV];
  int jmax[NV], hit[NV];
  unsigned short plist[512];
  double scratch_base[1];
};

// value of vehicle j (wave-uniform index) of a per-slot quantity f(h) -> every lane
template <int K, typename F>
__device__ __forceinline__ double wide_bcast(F f, int j) {
  const int ls = j & 63, hs = j >> 6;
  double r = wave_bcast(f(0), ls);
#pragma unroll
  for (int h = 1; h < K; ++h) {
    const double t = wave_bcast(f(h), ls);
    r = (hs == h) ? t : r;
  }
  return r;
}
template <int K, typename F>
__device__ __forceinline__ int wide_bcast_i(F f, int j) {
  const int ls = j & 63, hs = j >> 6;
  int r = wave_bcast_i(f(0), ls);
#pragma unroll
  for (int h = 1; h < K; ++h) {
    const int t = wave_bcast_i(f(h), ls);
    r = (hs == h) ? t : r;
  }
  return r;
}
template <int K>
__device__ __forceinline__ bool wide_any(const u64 (&m)[K]) {
  u64 a = 0;
#pragma unroll
  for (int h = 0; h < K; ++h) a |= m[h];
  return a != 0;
}
template <int K>
__device__ __forceinline__ int wide_popc(const u64 (&m)[K]) {
  int n = 0;
#pragma unroll
  for (int h = 0; h < K; ++h) n += __popcll(m[h]);
  return n;
}
// bit r of the set (r per thread)
template <int K>
__device__ __forceinline__ bool wide_test(const u64 (&m)[K], int r) {
  u64 w = m[0];
#pragma unroll
  for (int h = 1; h < K; ++h) w = ((r >> 6) == h) ? m[h] : w;
  return (w >> (r & 63)) & 1;
}
// number of set bits below position r (r per thread)
template <int K>
__device__ __forceinline__ int wide_popc_below(const u64 (&m)[K], int r) {
  const int rw = r >> 6, rb = r & 63;
  int n = 0;
#pragma unroll
  for (int h = 0; h < K; ++h) {
    const u64 mm = (h < rw) ? m[h] : ((h == rw) ? (m[h] & (((u64)1 << rb) - 1)) : 0);
    n += __popcll(mm);
  }
  return n;
}
// front / rear ranks on a lane from its rank-space membership mask; -1 if none (mask_neighbours of hwy_wave.h on K words)
template <int K>
__device__ __forceinline__ void wide_mask_neighbours(const u64 (&m)[K], int r, int *front, int *rear) {
  const int rw = r >> 6, rb = r & 63;
  int f = -1, b = -1;
#pragma unroll
  for (int h = K - 1; h >= 0; --h) {  // descending: the lowest word with a member above r is taken last
    const u64 mm = (h == rw) ? (m[h] & ~(((u64)2 << rb) - 1)) : ((h < rw) ? 0 : m[h]);
    f = mm ? h * 64 + ctz64(mm) : f;
  }
#pragma unroll
  for (int h = 0; h < K; ++h) {  // ascending: the highest word with a member below r is taken last
    const u64 mm = (h == rw) ? (m[h] & (((u64)1 << rb) - 1)) : ((h > rw) ? 0 : m[h]);
    b = mm ? h * 64 + msb64(mm) : b;
  }
  *front = f;
  *rear = b;
}

// the front rank only (the own lane and the target lane are never asked for a follower)
template <int K>
__device__ __forceinline__ int wide_mask_front(const u64 (&m)[K], int r) {
  const int rw = r >> 6, rb = r & 63;
  int f = -1;
#pragma unroll
  for (int h = K - 1; h >= 0; --h) {
    const u64 mm = (h == rw) ? (m[h] & ~(((u64)2 << rb) - 1)) : ((h < rw) ? 0 : m[h]);
    f = mm ? h * 64 + ctz64(mm) : f;
  }
  return f;
}

// Road.neighbour_vehicles literal scan for the equal-x case (wave_neighbours_scan of hwy_wave.h over K slots per thread)
template <int K>
__device__ inline void wide_neighbours_scan(const StepParams &p, const Veh (&me)[K], double myx, int self, int Lq, int *front,
                                            int *rear) {
  int f = -1, b = -1;
  double s_front = 0, s_rear = 0;
  for (int j = 0; j < p.N; ++j) {  // wave-uniform j
    const double s_v = wide_bcast<K>([&](int h) { return me[h].x; }, j);
    const double lat_v = wide_bcast<K>([&](int h) { return me[h].y; }, j) - Lq * p.lane_width;
    if (j == self) continue;
    if (!(fabs(lat_v) <= p.lane_width / 2 + 1.0 && -5.0 <= s_v && s_v < p.road_length + 5.0)) continue;
    if (myx <= s_v && (f < 0 || s_v <= s_front)) { s_front = s_v; f = j; }
    if (s_v < myx && (b < 0 || s_v > s_rear)) { s_rear = s_v; b = j; }
  }
  *front = f;
  *rear = b;
}

// Rank of every vehicle along the road (0 = smallest x; equal x ordered by list index), carried from frame to frame and merely
// re-validated (wave_update_rank of hwy_wave.h).  The exchange goes through sh.x (LDS memory: a rank is a slot of 64 K, not a
// lane): every vehicle writes its x to slot `rank`, thread l then reads the slots 64 h + l and their successors.  On return
// sh.x holds the x of the FINAL ranks only if nothing was recounted -- callers rewrite the snapshot anyway.
template <int K>
__device__ inline void wide_update_rank(WideShared<K> &sh, const Veh (&me)[K], int N, int (&rank)[K], bool &has_tie) {
  const int l = threadIdx.x;
  bool recount = false;
  {
    auto verify = [&](u64 (&inv)[K]) -> bool {
      HWY_WAVE_LDS_FENCE();  // earlier reads of sh.x are complete
#pragma unroll
      for (int h = 0; h < K; ++h) sh.x[rank[h]] = me[h].x;
      HWY_WAVE_LDS_FENCE();
      bool bad = false;
#pragma unroll
      for (int h = 0; h < K; ++h) {
        const int r = h * 64 + l;
        const double x_r = sh.x[r], x_r1 = sh.x[r + 1 < 64 * K ? r + 1 : r];
        const bool in = r < N - 1;
        bad = bad || (in && !(x_r < x_r1));
        inv[h] = __ballot(in && x_r > x_r1);
      }
      return __ballot(bad) != 0;
    };
    u64 inv[K];
    recount = verify(inv);
    if (recount) {  // (uniform over the wave)
      // Cheap repair: disjoint adjacent inversions (one overtake somewhere on the road) are undone by swapping the two ranks
      u64 shl[K];
#pragma unroll
      for (int h = 0; h < K; ++h) shl[h] = (inv[h] << 1) | (h > 0 ? inv[h - 1] >> 63 : 0);
      bool overlap = false;
#pragma unroll
      for (int h = 0; h < K; ++h) overlap = overlap || (inv[h] & shl[h]) != 0;
      if (wide_any<K>(inv) && !overlap) {
#pragma unroll
        for (int h = 0; h < K; ++h) {
          const bool up = wide_test<K>(inv, rank[h]);
          const bool down = rank[h] > 0 && wide_test<K>(inv, rank[h] - 1);
          rank[h] += up ? 1 : (down ? -1 : 0);  // idle slots hold ranks >= N: their bits are never set
        }
        u64 inv2[K];
        recount = verify(inv2);
      }
    }
    if (!recount) has_tie = false;  // strictly increasing => all x distinct
  }
  if (recount) {  // wave-uniform
    // Counting pass on the HIGH 32 bits of x first (hwy_wave.h), exact f64 compares only if two high words coincide
    int hi[K], cnt_lt[K], cnt_le[K];
#pragma unroll
    for (int h = 0; h < K; ++h) { hi[h] = __double2hiint(me[h].x); cnt_lt[h] = cnt_le[h] = 0; }
#pragma unroll
    for (int hs = 0; hs < K; ++hs) {
      const int n_h = N - 64 * hs < 64 ? N - 64 * hs : 64;
      for (int j = 0; j < n_h; ++j) {
        const int hj = wave_bcast_i(hi[hs], j);
#pragma unroll
        for (int h = 0; h < K; ++h) {
          cnt_lt[h] += (hj < hi[h]) ? 1 : 0;
          cnt_le[h] += (hj <= hi[h]) ? 1 : 0;
        }
      }
    }
    bool amb = false;
#pragma unroll
    for (int h = 0; h < K; ++h) amb = amb || (h * 64 + l < N && ((cnt_le[h] - cnt_lt[h]) > 1 || hi[h] < 0));
    if (__ballot(amb) != 0) {  // ambiguous: exact pass
#pragma unroll
      for (int h = 0; h < K; ++h) cnt_lt[h] = cnt_le[h] = 0;
#pragma unroll
      for (int hs = 0; hs < K; ++hs) {
        const int n_h = N - 64 * hs < 64 ? N - 64 * hs : 64;
        for (int j = 0; j < n_h; ++j) {
          const double xj = wave_bcast(me[hs].x, j);
#pragma unroll
          for (int h = 0; h < K; ++h) {
            cnt_lt[h] += (xj < me[h].x) ? 1 : 0;
            cnt_le[h] += (xj <= me[h].x) ? 1 : 0;
          }
        }
      }
    }
    bool tie[K], any_tie = false;
#pragma unroll
    for (int h = 0; h < K; ++h) {
      const bool active = h * 64 + l < N;
      tie[h] = active && (cnt_le[h] - cnt_lt[h]) > 1;
      any_tie = any_tie || tie[h];
      rank[h] = active ? cnt_lt[h] : h * 64 + l;  // idle slots keep their own: the table stays a bijection
    }
    has_tie = __ballot(any_tie) != 0;
    if (has_tie) {  // equal x: order by list index, like a stable sort (rare)
#pragma unroll
      for (int hs = 0; hs < K; ++hs) {
        const int n_h = N - 64 * hs < 64 ? N - 64 * hs : 64;
        for (int j = 0; j < n_h; ++j) {
          const double xj = wave_bcast(me[hs].x, j);
#pragma unroll
          for (int h = 0; h < K; ++h) rank[h] += (h * 64 + l < N && xj == me[h].x && hs * 64 + j < h * 64 + l) ? 1 : 0;
        }
      }
    }
  }
}

// KinematicObservation + reward + done for every agent (observe_wave of hwy_wave.h with K vehicles per thread; Kinematics only --
// the OccupancyGrid observation of N > 64 stays on the workgroup kernel).
template <int K, bool BY_RANK>
__device__ inline void observe_wide(const StepParams &p, WideShared<K> &sh, int e, int eo, const Veh (&me)[K], bool write_reward,
                                    const int (&rank)[K], double env_time = 0.0) {  // env_time: the clock at the start of the step
  const int l = threadIdx.x;
  const int V = p.V, F = p.F, N = p.N;
  for (int a = 0; a < p.A; ++a) {
    const int ia = p.agent_index[a];
    const double ex = wide_bcast<K>([&](int h) { return me[h].x; }, ia), ey = wide_bcast<K>([&](int h) { return me[h].y; }, ia);
    const double ev = wide_bcast<K>([&](int h) { return me[h].v; }, ia);
    const double ec = wide_bcast<K>([&](int h) { return me[h].ch; }, ia), es = wide_bcast<K>([&](int h) { return me[h].sh; }, ia);
    bool elig[K];
    double key[K];
    u64 elig_m[K];
#pragma unroll
    for (int h = 0; h < K; ++h) {
      const int vi = h * 64 + l;
      const double dxe = me[h].x - ex, dye = me[h].y - ey;
      const double d_lane = me[h].x - ex;
      // norm < distance  <=>  dx^2 + dy^2 < distance^2 (hwy_wave.h)
      elig[h] = vi < N && vi != ia && (dxe * dxe + dye * dye < p.perception * p.perception) &&
                ((p.flags & HWY_C_OBS_SEE_BEHIND) || (-2 * HWY_VEH_LENGTH < d_lane));
      key[h] = elig[h] ? ((p.flags & HWY_C_OBS_UNSORTED) ? 0.0 : fabs(d_lane)) : __builtin_inf();
      elig_m[h] = __ballot(elig[h]);
    }
    const int n_elig = wide_popc<K>(elig_m);
    const int m = n_elig < V - 1 ? n_elig : V - 1;
    // stable sort position among the eligible (ties keep list order)
    int pos[K];
#pragma unroll
    for (int h = 0; h < K; ++h) pos[h] = 0;
    // walk a set of vehicles (wave-uniform) and count, per slot, those that sort before it
    auto count_before = [&](const u64 (&set)[K]) {
#pragma unroll
      for (int hs = 0; hs < K; ++hs) {
        for (u64 em = set[hs]; em; em &= em - 1) {  // wave-uniform
          const int ks = ctz64(em), k = hs * 64 + ks;
          const double kk = wave_bcast(key[hs], ks);
#pragma unroll
          for (int h = 0; h < K; ++h) pos[h] += ((kk < key[h]) || (kk == key[h] && k < h * 64 + l)) ? 1 : 0;
        }
      }
    };
    if (BY_RANK && !(p.flags & (HWY_C_OBS_SEE_BEHIND | HWY_C_OBS_UNSORTED))) {  // wave-uniform
      // near (key < 2 LENGTH: a handful, explicit compares) / far (all in front, ordered like their rank along the road)
      bool near[K], far[K];
      u64 near_m[K], far_r[K];
      HWY_WAVE_LDS_FENCE();
#pragma unroll
      for (int h = 0; h < K; ++h) {
        near[h] = elig[h] && key[h] < 2 * HWY_VEH_LENGTH;
        far[h] = elig[h] && !near[h];
        near_m[h] = __ballot(near[h]);
        sh.sbits[rank[h]] = far[h] ? 1 : 0;
      }
      HWY_WAVE_LDS_FENCE();
#pragma unroll
      for (int h = 0; h < K; ++h) far_r[h] = __ballot(sh.sbits[h * 64 + l] != 0);  // rank space
      count_before(near_m);
      const int n_near = wide_popc<K>(near_m);
#pragma unroll
      for (int h = 0; h < K; ++h) pos[h] = far[h] ? n_near + wide_popc_below<K>(far_r, rank[h]) : pos[h];
    } else {
      count_before(elig_m);  // (only eligible vehicles can precede an eligible one)
    }
    if (p.obs) {
      float *out = p.obs + ((size_t)eo * p.A + a) * (size_t)(V * F);
#pragma unroll
      for (int h = 0; h < K; ++h) {
        const int vi = h * 64 + l;
        const Veh &mv = me[h];
        const int row = (vi == ia) ? 0 : (elig[h] && pos[h] < V - 1 ? pos[h] + 1 : -1);
        if (p.obs_std5) {  // wave-uniform: features == [presence, x, y, vx, vy]
          if (vi < N && row >= 0) {
            double fx = mv.x, fy = mv.y, fvx = mv.v * mv.ch, fvy = mv.v * mv.sh;
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
        } else if (vi < N && row >= 0) {
          for (int f = 0; f < F; ++f) {
            const int fid = p.feat[f];
            double val = EnvBlock<1>::feature(p, fid, mv.x, mv.y, mv.h, mv.v, mv.ch, mv.sh, mv.lane);
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
      }
      for (int t = l; t < V * F; t += 64)
        if (t / F > m) out[t] = 0.0f;
    }
    if (write_reward) {
#pragma unroll
      for (int h = 0; h < K; ++h) {
        if (h * 64 + l != ia) continue;
        const Veh &mv = me[h];
        const bool crashed = (mv.flags & HWY_F_CRASHED) != 0;
        const bool on_road = fabs(mv.y - mv.lane * p.lane_width) <= p.lane_width / 2 + 0.0 && -5.0 <= mv.x &&
                             mv.x < p.road_length + 5.0;
        const double forward_speed = mv.v * mv.ch;
        const double scaled_speed = lmap(forward_speed, p.rs0, p.rs1, 0.0, 1.0);  // true divisions (hwy_wave.h)
        const int nl = p.L - 1 > 1 ? p.L - 1 : 1;
        double reward = 0.0;
        reward = reward + p.collision_reward * (crashed ? 1.0 : 0.0);
        reward = reward + p.right_lane_reward * ((double)mv.tgt / (double)nl);
        reward = reward + p.high_speed_reward * clipd(scaled_speed, 0.0, 1.0);
        reward = reward + 0.0 * (on_road ? 1.0 : 0.0);
        if (p.flags & HWY_C_NORMALIZE_REWARD)
          reward = lmap(reward, p.collision_reward, p.high_speed_reward + p.right_lane_reward, 0.0, 1.0);
        reward *= (on_road ? 1.0 : 0.0);
        p.reward[(size_t)eo * p.A + a] = reward;
        if (p.info_speed) p.info_speed[(size_t)eo * p.A + a] = mv.v;
        if (p.info_crashed) p.info_crashed[(size_t)eo * p.A + a] = crashed ? 1 : 0;
        if (a == 0) {
          const bool term = crashed || ((p.flags & HWY_C_OFFROAD_TERMINAL) && !on_road);
          const double t = env_time + p.policy_dt;  // (the clock was requested with the state: hwy_wave.h)
          const bool trunc = t >= p.duration;
          p.st.time[e] = t;
          p.terminated[eo] = term ? 1 : 0;
          p.truncated[eo] = trunc ? 1 : 0;
          if (p.autoreset) p.st.done[e] = (term || trunc) ? 1 : 0;
        }
      }
    }
  }
}

// =============================================================================================
// One policy step of environment e by its wavefront, K vehicles per thread; eo = row of the action / output planes.
template <int K>
__device__ __forceinline__ void wide_policy_step(const StepParams &p, WideShared<K> &sh, const int e, const int eo) {
  typedef EnvBlock<1> B;
  const int l = threadIdx.x;
  const int N = p.N;
  bool active[K];
  int vi[K];
#pragma unroll
  for (int h = 0; h < K; ++h) { vi[h] = h * 64 + l; active[h] = vi[h] < N; }

  // everything the step needs from HBM is requested before anything is waited for (hwy_wave.h: one round trip instead of three)
  int done_flag = p.autoreset ? (int)p.st.done[e] : 0;
  double env_time = p.st.time[e];
  int act_lane = (p.actions && l < p.A) ? p.actions[(size_t)eo * p.A + l] : HWY_IDLE;
  Veh me[K];
#pragma unroll
  for (int h = 0; h < K; ++h) load_vehicle_at(p, e, vi[h], me[h]);
  HWY_ISSUED_TOGETHER(done_flag, env_time, act_lane, me[0].x, me[K - 1].timer);
  // ---- auto-reset: re-spawn instead of stepping ---------------------------------------------------------------
  if (done_flag) {
    const uint32_t episode = p.st.episode[e] + 1u;
    const uint64_t seed = p.rp.base_seed + (uint64_t)e;
    SpawnDraw d[K];
#pragma unroll
    for (int h = 0; h < K; ++h) {
      me[h] = Veh{};  // (the old state was fetched for nothing)
      d[h] = spawn_draw(p, vi[h], seed, episode);
      if (active[h]) sh.x[vi[h]] = d[h].step;
      if (active[h] && vi[h] == 0) sh.scratch_base[0] = 3 * d[h].offset;  // first vehicle starts from 3*offset
    }
    HWY_WAVE_LDS_FENCE();
    // x_k = running sum of the steps in creation order (spawn_env)
#pragma unroll
    for (int h = 0; h < K; ++h) {
      double x = sh.scratch_base[0];
      for (int k = 0; k <= vi[h] && k < N; ++k) x += sh.x[k];
      spawn_fill(p, d[h], x, vi[h], me[h]);
    }
    HWY_WAVE_LDS_FENCE();
    int rank0[K];
#pragma unroll
    for (int h = 0; h < K; ++h) rank0[h] = vi[h];
    observe_wide<K, false>(p, sh, e, eo, me, false, rank0);
#pragma unroll
    for (int h = 0; h < K; ++h) {
      store_vehicle_at(p, e, vi[h], me[h]);
      if (active[h] && (me[h].flags & HWY_F_CONTROLLED)) {
        for (int a = 0; a < p.A; ++a)
          if (p.agent_index[a] == vi[h]) {
            p.reward[(size_t)eo * p.A + a] = 0.0;
            if (p.info_speed) p.info_speed[(size_t)eo * p.A + a] = me[h].v;
            if (p.info_crashed) p.info_crashed[(size_t)eo * p.A + a] = 0;
          }
      }
    }
    if (l == 0) {
      p.st.time[e] = 0.0;
      p.st.done[e] = 0;
      p.st.episode[e] = episode;
      p.terminated[eo] = 0;
      p.truncated[eo] = 0;
    }
    return;
  }

  bool controlled[K], idm[K], i_check[K];
  int act0[K], rank[K];
  u64 chk[K];
#pragma unroll
  for (int h = 0; h < K; ++h) {
    controlled[h] = active[h] && (me[h].flags & HWY_F_CONTROLLED);
    idm[h] = active[h] && !controlled[h];
    act0[h] = HWY_IDLE;
  }
  for (int a = 0; a < p.A; ++a) {  // wave-uniform
    const int act_a = wave_bcast_i(act_lane, a);
#pragma unroll
    for (int h = 0; h < K; ++h)
      if (controlled[h] && p.agent_index[a] == vi[h]) act0[h] = HWY_ACTION_TO_ALL(p.action_set, act_a);
  }
#pragma unroll
  for (int h = 0; h < K; ++h) {
    i_check[h] = (me[h].flags & HWY_F_CHECK_COLLISIONS) != 0;
    chk[h] = __ballot(active[h] && i_check[h]);
    // idle slots keep their own so that the table stays a bijection (the hint is the engine's own: a permutation of 0 .. N - 1
    // since the last spawn / hwy_set_state; masked to the table's size so that no state word can address outside it)
    rank[h] = active[h] ? (me[h].rank & (64 * K - 1)) : vi[h];
  }
  const bool all_check = wide_popc<K>(chk) == N;
  bool has_tie = false;
  double inv_v0[K];
#pragma unroll
  for (int h = 0; h < K; ++h) inv_v0[h] = 0.0;

  for (int fr = 0; fr < p.n_frames; ++fr) {
    // ---- A. meta-action (abstract.py:294-304 -> controller.py:295-315) ------------------------------
    if (fr == 0 && p.actions) {
#pragma unroll
      for (int h = 0; h < K; ++h) {
        if (!controlled[h]) continue;
        const int act = act0[h];
        if (act == HWY_FASTER || act == HWY_SLOWER) {
          const double xs = (me[h].v - p.target_speeds[0]) / (p.target_speeds[p.n_ts - 1] - p.target_speeds[0]);
          int idx = (int)clipd(rint(xs * (p.n_ts - 1)), 0.0, (double)(p.n_ts - 1)) + (act == HWY_FASTER ? 1 : -1);
          idx = idx < 0 ? 0 : (idx > p.n_ts - 1 ? p.n_ts - 1 : idx);
          me[h].sidx = idx;
          me[h].ts = p.target_speeds[idx];
        } else if (act == HWY_LANE_LEFT || act == HWY_LANE_RIGHT) {
          int id = me[h].tgt + (act == HWY_LANE_RIGHT ? 1 : -1);
          id = id < 0 ? 0 : (id > p.L - 1 ? p.L - 1 : id);
          if (B::reachable(p, id, me[h].x, me[h].y)) me[h].tgt = id;
        }
      }
    }

    // ---- C. rank along the road, lane membership masks, frame-start snapshot ------------------------------
    wide_update_rank<K>(sh, me, N, rank, has_tie);
    double log_ratio[K];
    if (fr == 0) {  // (after the meta-action: the target speed is fixed for the step)
#pragma unroll
      for (int h = 0; h < K; ++h) inv_v0[h] = B::idm_inv_v0(p, me[h].ts);
    }
    if constexpr (K == 2) {  // EnvBlock::idm_log_ratio_inv for both vehicles at once (log_pos2: hwy_math.h)
      const double r0 = fmax(me[0].v, 0.0) * inv_v0[0], r1 = fmax(me[1].v, 0.0) * inv_v0[1];
      double l0, l1;
      log_pos2(r0 > 0.0 ? r0 : 1.0, r1 > 0.0 ? r1 : 1.0, l0, l1);
      log_ratio[0] = active[0] ? (r0 > 0.0 ? l0 : -__builtin_inf()) : 0.0;
      log_ratio[1] = active[1] ? (r1 > 0.0 ? l1 : -__builtin_inf()) : 0.0;
    }
    HWY_WAVE_LDS_FENCE();  // previous readers of the snapshot / the masks are done
    if (l < p.L + 2) {
#pragma unroll
      for (int w = 0; w < K; ++w) sh.lane_mask[l][w] = 0;
    }
    HWY_WAVE_LDS_FENCE();
#pragma unroll
    for (int h = 0; h < K; ++h) {
      const bool inr = active[h] && (-5.0 <= me[h].x) && (me[h].x < p.road_length + 5.0);
      int bits = 0;
      for (int L = 0; L < p.L; ++L)
        bits |= (inr && (fabs(me[h].y - L * p.lane_width) <= p.lane_width / 2 + 1.0)) ? (1 << L) : 0;
      if constexpr (K != 2) log_ratio[h] = active[h] ? B::idm_log_ratio_inv(me[h].v, inv_v0[h]) : 0.0;
      const int r = rank[h];
      // lane membership (AbstractLane.on_lane, margin 1) in rank space: every vehicle ORs its rank bit into the masks of the
      // lanes it is on (one or two: ds_or_b64; row L+1 = lane L, rows 0 and L+1 stay 0 for "no such lane")
      for (int b_ = bits; b_; b_ &= b_ - 1)
        __hip_atomic_fetch_or(&sh.lane_mask[__builtin_ctz(b_) + 1][r >> 6], (u64)1 << (r & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (active[h]) {
        sh.x[r] = me[h].x; sh.v[r] = me[h].v; sh.c[r] = me[h].ch; sh.s[r] = me[h].sh; sh.lr[r] = log_ratio[h];
        sh.idx[r] = vi[h];
      }
    }
    HWY_WAVE_LDS_FENCE();

    // ---- D. Road.act: lane-change policy (behavior.py:219-263) ----------------------------------------
    bool crashed0[K], drives[K], changer[K], decide[K], left_ok[K], right_ok[K], ok_l[K], ok_r[K];
    int tgt_old[K], fo[K], fl[K], frt[K], ft[K], rl[K], rrt[K];
    double free_self[K], gap_own[K], delta[K];
#pragma unroll
    for (int h = 0; h < K; ++h) {
      Veh &mv = me[h];
      crashed0[h] = (mv.flags & HWY_F_CRASHED) != 0;
      drives[h] = idm[h] && !crashed0[h];
      tgt_old[h] = mv.tgt;
      changer[h] = drives[h] && mv.lane != mv.tgt;
      const double timer = mv.timer;
      decide[h] = drives[h] && mv.lane == mv.tgt && (HWY_LC_DELAY < timer);
      // IDMVehicle timer: reset by a decision (behavior.py:248), then += dt in step (behavior.py:147)
      mv.timer = idm[h] ? (decide[h] ? 0.0 : timer) + p.dt : timer;
      left_ok[h] = mv.lane - 1 >= 0;
      right_ok[h] = mv.lane + 1 < p.L;
      if (!has_tie) {
        u64 m_own[K], m_tgt[K];
#pragma unroll
        for (int w = 0; w < K; ++w) { m_own[w] = sh.lane_mask[mv.lane + 1][w]; m_tgt[w] = sh.lane_mask[mv.tgt + 1][w]; }
        fo[h] = wide_mask_front<K>(m_own, rank[h]);
        ft[h] = wide_mask_front<K>(m_tgt, rank[h]);  // (only read by a vehicle on its way to another lane)
        fl[h] = frt[h] = rl[h] = rrt[h] = -1;
#ifdef HWY_WAVE_MOBIL_PER_THREAD
        u64 m_left[K], m_right[K];
#pragma unroll
        for (int w = 0; w < K; ++w) { m_left[w] = sh.lane_mask[mv.lane][w]; m_right[w] = sh.lane_mask[mv.lane + 2][w]; }
        wide_mask_neighbours<K>(m_left, rank[h], &fl[h], &rl[h]);
        wide_mask_neighbours<K>(m_right, rank[h], &frt[h], &rrt[h]);
#endif
      }
    }
    if (has_tie) {  // wave-uniform: literal scans (vehicle INDICES), converted to ranks below
      int q[K][6];
#pragma unroll
      for (int h = 0; h < K; ++h) {
        const Veh &mv = me[h];
        int a, b;
        wide_neighbours_scan<K>(p, me, mv.x, vi[h], mv.lane, &a, &b);
        q[h][0] = a;
        wide_neighbours_scan<K>(p, me, mv.x, vi[h], left_ok[h] ? mv.lane - 1 : mv.lane, &a, &b);
        q[h][1] = a; q[h][4] = b;
        wide_neighbours_scan<K>(p, me, mv.x, vi[h], right_ok[h] ? mv.lane + 1 : mv.lane, &a, &b);
        q[h][2] = a; q[h][5] = b;
        wide_neighbours_scan<K>(p, me, mv.x, vi[h], mv.tgt, &a, &b);
        q[h][3] = a;
        fo[h] = fl[h] = frt[h] = ft[h] = rl[h] = rrt[h] = -1;
      }
      for (int j = 0; j < N; ++j) {  // index -> rank
        const int rk = wide_bcast_i<K>([&](int hh) { return rank[hh]; }, j);
#pragma unroll
        for (int h = 0; h < K; ++h) {
          fo[h] = (q[h][0] == j) ? rk : fo[h]; fl[h] = (q[h][1] == j) ? rk : fl[h]; frt[h] = (q[h][2] == j) ? rk : frt[h];
          ft[h] = (q[h][3] == j) ? rk : ft[h]; rl[h] = (q[h][4] == j) ? rk : rl[h]; rrt[h] = (q[h][5] == j) ? rk : rrt[h];
        }
      }
    }
    // Straight-line evaluation for every slot (hwy_wave.h): the K vehicles of a thread are independent chains
    if constexpr (K == 2) {  // EnvBlock::idm_free_from_log for both vehicles at once (exp_bounded2: hwy_math.h)
      double e0, e1;
      exp_bounded2(me[0].delta * log_ratio[0], me[1].delta * log_ratio[1], e0, e1);
      free_self[0] = HWY_COMFORT_ACC_MAX * (1 - e0);
      free_self[1] = HWY_COMFORT_ACC_MAX * (1 - e1);
    }
    double gap_new[K], gap_sl[K], gap_sr[K];  // (gap_sl / gap_sr: separate arrays -- a [K][2] indexed by the verdict went to scratch memory); gap_new: the IDM gap term towards the leader on the side MOBIL picks in this frame
#ifndef HWY_WAVE_MOBIL_PER_THREAD
    if (!has_tie) {  // wave-uniform
      // MOBIL compacted (hwy_wave.h, round 6): with a decision per vehicle and second, a fifteenth of highway-v0's traffic decides
      // in a given frame -- ~7 of config 3's 101 vehicles -- while rounds 4-5 evaluated both side lanes of BOTH vehicles of every
      // thread in every frame.  Decider number d hands (free-road term, own-lane acceleration, delta, rank | lane | side bits) over
      // through LDS; thread t evaluates side t & 1 of decider t >> 1 from the rank-ordered snapshot; verdicts come back as a
      // ballot, the chosen side's gap term through LDS.  Same operations on the same values as the per-thread form
      // (-DHWY_WAVE_MOBIL_PER_THREAD): bit-identical.
      bool cl[K], cr[K];
      u64 dm[K];
      int n_dec = 0, d[K];
#pragma unroll
      for (int h = 0; h < K; ++h) {
        const Veh &mv = me[h];
        const int g_fo = fo[h] < 0 ? 0 : fo[h];
        delta[h] = mv.delta;
        if constexpr (K != 2) free_self[h] = B::idm_free_from_log(log_ratio[h], delta[h]);
        gap_own[h] = fo[h] >= 0 ? B::idm_gap(mv.x, mv.v, mv.ch, mv.sh, sh.x[g_fo], sh.v[g_fo], sh.c[g_fo], sh.s[g_fo]) : 0.0;
        const bool moving = !(fabs(mv.v) < 1);
        cl[h] = decide[h] && left_ok[h] && B::reachable(p, mv.lane - 1, mv.x, mv.y) && moving;
        cr[h] = decide[h] && right_ok[h] && B::reachable(p, mv.lane + 1, mv.x, mv.y) && moving;
        ok_l[h] = ok_r[h] = false;
        gap_new[h] = 0.0;
        dm[h] = __ballot(cl[h] || cr[h]);
        d[h] = n_dec + __popcll(dm[h] & (((u64)1 << l) - 1));
        n_dec += __popcll(dm[h]);
      }
      if (n_dec) {  // wave-uniform
        int *const word = reinterpret_cast<int *>(sh.nc);  // (the post-integration bodies only live inside section G)
        HWY_WAVE_LDS_FENCE();
#pragma unroll
        for (int h = 0; h < K; ++h)
          if (cl[h] || cr[h]) {
            sh.nx[d[h]] = free_self[h]; sh.ny[d[h]] = free_self[h] - gap_own[h]; sh.nv[d[h]] = delta[h];
            word[d[h]] = rank[h] | (me[h].lane << 8) | (cl[h] ? 1 << 16 : 0) | (cr[h] ? 1 << 17 : 0);
          }
        HWY_WAVE_LDS_FENCE();
        const int n_tasks = 2 * n_dec;
        for (int base = 0; base < n_tasks; base += 64) {  // wave-uniform
          const int t = base + l;
          const bool tv = t < n_tasks;
          const int dd = tv ? t >> 1 : 0, side = t & 1;
          const int w_ = word[dd];
          const int rk = w_ & 255, ln = (w_ >> 8) & 255;
          const bool en = tv && ((w_ >> (16 + side)) & 1);
          const double fs = sh.nx[dd], sa = sh.ny[dd], dl = sh.nv[dd];
          u64 m[K];
#pragma unroll
          for (int w = 0; w < K; ++w) m[w] = sh.lane_mask[ln + (side ? 2 : 0)][w];
          int f, r;
          wide_mask_neighbours<K>(m, rk, &f, &r);
          const double ex = sh.x[rk], ev = sh.v[rk], ec = sh.c[rk], es = sh.s[rk];
          const int gf = f < 0 ? 0 : f;
          const double gap = f >= 0 ? B::idm_gap(ex, ev, ec, es, sh.x[gf], sh.v[gf], sh.c[gf], sh.s[gf]) : 0.0;
          bool ok = en && !(((fs - gap) - sa) < HWY_LC_MIN_ACC_GAIN);
          const bool pend = ok && r >= 0;
          if (__ballot(pend) != 0) {  // wave-uniform: safety of the new follower (hwy_wave.h)
            const int rf = pend ? r : 0;
            const double lr_f = sh.lr[rf];
            const double g = B::idm_gap(sh.x[rf], sh.v[rf], sh.c[rf], sh.s[rf], ex, ev, ec, es);
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
          sh.ns[l] = gap;
          HWY_WAVE_LDS_FENCE();
#pragma unroll
          for (int h = 0; h < K; ++h)
            if ((cl[h] || cr[h]) && 2 * d[h] >= base && 2 * d[h] < base + 64) {
              const int bits = (int)(okm >> (2 * d[h] - base)) & 3;
              ok_l[h] = (bits & 1) != 0;
              ok_r[h] = (bits & 2) != 0;
              if (bits) gap_new[h] = sh.ns[2 * d[h] - base + (ok_r[h] ? 1 : 0)];  // right wins if both pass
            }
          HWY_WAVE_LDS_FENCE();
        }
      }
    } else
#endif
    {
#pragma unroll
    for (int h = 0; h < K; ++h) {
      const Veh &mv = me[h];
      const int g_fo = fo[h] < 0 ? 0 : fo[h], g_fl = fl[h] < 0 ? 0 : fl[h], g_fr = frt[h] < 0 ? 0 : frt[h];
      const double fo_x = sh.x[g_fo], fo_v = sh.v[g_fo], fo_c = sh.c[g_fo], fo_s = sh.s[g_fo];
      const double fl_x = sh.x[g_fl], fl_v = sh.v[g_fl], fl_c = sh.c[g_fl], fl_s = sh.s[g_fl];
      const double fr_x = sh.x[g_fr], fr_v = sh.v[g_fr], fr_c = sh.c[g_fr], fr_s = sh.s[g_fr];
      delta[h] = mv.delta;
      if constexpr (K != 2) free_self[h] = B::idm_free_from_log(log_ratio[h], delta[h]);
      gap_own[h] = fo[h] >= 0 ? B::idm_gap(mv.x, mv.v, mv.ch, mv.sh, fo_x, fo_v, fo_c, fo_s) : 0.0;
      // MOBIL (behavior.py:265-324), both candidates side by side
      const double self_a = free_self[h] - gap_own[h];
      const bool moving = !(fabs(mv.v) < 1);
      const bool cl = decide[h] && left_ok[h] && B::reachable(p, mv.lane - 1, mv.x, mv.y) && moving;
      const bool cr = decide[h] && right_ok[h] && B::reachable(p, mv.lane + 1, mv.x, mv.y) && moving;
      const double gap_l = fl[h] >= 0 ? B::idm_gap(mv.x, mv.v, mv.ch, mv.sh, fl_x, fl_v, fl_c, fl_s) : 0.0;
      const double gap_r = frt[h] >= 0 ? B::idm_gap(mv.x, mv.v, mv.ch, mv.sh, fr_x, fr_v, fr_c, fr_s) : 0.0;
      ok_l[h] = cl && !(((free_self[h] - gap_l) - self_a) < HWY_LC_MIN_ACC_GAIN);
      ok_r[h] = cr && !(((free_self[h] - gap_r) - self_a) < HWY_LC_MIN_ACC_GAIN);
      gap_sl[h] = gap_l; gap_sr[h] = gap_r;
    }
    // safety of the new follower, only for candidates that passed the incentive test, one side per pass (hwy_wave.h)
    {
      bool pend_l[K], pend_r[K], any_pend = false;
#pragma unroll
      for (int h = 0; h < K; ++h) {
        pend_l[h] = ok_l[h] && rl[h] >= 0;
        pend_r[h] = ok_r[h] && rrt[h] >= 0;
        any_pend = any_pend || pend_l[h] || pend_r[h];
      }
      while (__ballot(any_pend) != 0) {  // wave-uniform
        bool need_exp = false, sure_safe[K], pend[K], left[K];
        double g[K], lr_f[K];
#pragma unroll
        for (int h = 0; h < K; ++h) {
          const Veh &mv = me[h];
          pend[h] = pend_l[h] || pend_r[h];
          left[h] = pend_l[h];
          const int rf = pend[h] ? (left[h] ? rl[h] : rrt[h]) : 0;
          lr_f[h] = sh.lr[rf];
          g[h] = B::idm_gap(sh.x[rf], sh.v[rf], sh.c[rf], sh.s[rf], mv.x, mv.v, mv.ch, mv.sh);
          const bool sure_unsafe = g[h] > HWY_COMFORT_ACC_MAX + HWY_LC_MAX_BRAKING + 1e-6;
          sure_safe[h] = lr_f[h] < 0.0 && delta[h] > 0.0 && g[h] <= HWY_LC_MAX_BRAKING - 1e-6;
          need_exp = need_exp || (pend[h] && !sure_unsafe && !sure_safe[h]);
        }
        const bool run_exp = __ballot(need_exp) != 0;  // wave-uniform
        any_pend = false;
#pragma unroll
        for (int h = 0; h < K; ++h) {
          bool safe = sure_safe[h];
          if (run_exp) {
            const double a_f = B::idm_free_from_log(lr_f[h], delta[h]) - g[h];
            safe = !(a_f < -HWY_LC_MAX_BRAKING);
          }
          if (pend[h]) {
            if (left[h]) { ok_l[h] = safe; pend_l[h] = false; } else { ok_r[h] = safe; pend_r[h] = false; }
          }
          any_pend = any_pend || pend_l[h] || pend_r[h];
        }
      }
    }
#pragma unroll
      for (int h = 0; h < K; ++h) gap_new[h] = ok_r[h] ? gap_sr[h] : gap_sl[h];  // (only read when MOBIL picks a side)
    }
    // side_lanes order is [left, right] and the loop does not break: right wins if both pass
#pragma unroll
    for (int h = 0; h < K; ++h) {
      if (ok_l[h]) me[h].tgt = me[h].lane - 1;
      if (ok_r[h]) me[h].tgt = me[h].lane + 1;
    }
    // abort rule for ongoing lane changes (behavior.py:229-244): an ordered chain over Road.vehicles.
    // A changer c (on its way to lane T since an earlier frame) aborts if ANOTHER vehicle r heading for T from a third lane --
    // with the target r has when c acts: its current one for r before c in the list, the frame-start one for r after c -- is
    // ahead of it by less than the desired gap d*(c, r).  Literally that is one link per changer, each reading the targets the
    // earlier links left behind (hwy_wave.h: ~25 instructions per link, ~60 more with a rival; an environment of 101 vehicles
    // holds up to a dozen changers per frame and the slowest wavefronts of a launch spent a quarter of their lifetime here).
    // Three facts make it a per-THREAD computation without any loop over the changers:
    //  * a rival must be AHEAD and closer than d*, and d* <= 10 + 1.5 v + v (v + 5) / (2 sqrt(ab)) for every possible rival as long
    //    as no vehicle of the environment drives backwards or sideways faster than 5 m/s (checked, wave-uniform: otherwise the
    //    bound is infinite): in the rank order of the snapshot a changer walks the members of "heading for T" ahead of it and
    //    stops at the first one beyond that bound -- usually the very first (measured: 30 listed pairs per step, 0.0 inside);
    //  * the links only interact through ABORTS, and an abort can only REMOVE a rival (its target becomes its own lane): a
    //    blocking rival that is itself an EARLIER changer counts only while it has not aborted, every other one for good;
    //  * a link depends on earlier links only, so iterating "aborts = blocked for good, or blocked by an earlier changer that
    //    does not abort" from "nobody aborts" reaches the literal chain's result after (depth + 1) rounds -- ballots only.
    {
      u64 cm[K], mv_m[K];
#pragma unroll
      for (int h = 0; h < K; ++h) {
        cm[h] = __ballot(changer[h]);
        mv_m[h] = __ballot(active[h] && (me[h].lane != tgt_old[h] || me[h].lane != me[h].tgt));
      }
      // with a single vehicle on its way to another lane (the changer itself) no link can block
      const bool chain = wide_any<K>(cm) && wide_popc<K>(mv_m) > 1;
      if (chain) {  // wave-uniform
        // (1) S_T (rank space) = the vehicles heading for lane T from another lane: every such vehicle ORs its rank bit into row T
        // of the mask table (dead since the neighbour scans of this frame; zeroed again by the next frame's snapshot), and leaves
        // "decided in this frame | changer << 1" in its rank slot for the walk; every changer then reads the row of ITS target lane
        bool insane = false;
        HWY_WAVE_LDS_FENCE();  // (earlier readers of sbits and of the masks are done)
        if (l < p.L) {
#pragma unroll
          for (int w = 0; w < K; ++w) sh.lane_mask[l][w] = 0;
        }
        HWY_WAVE_LDS_FENCE();
#pragma unroll
        for (int h = 0; h < K; ++h) {
          const bool mover = active[h] && me[h].lane != me[h].tgt;
          const int r = rank[h];
          sh.sbits[r] = ((me[h].tgt != tgt_old[h]) ? 1 : 0) | ((changer[h] ? 1 : 0) << 1);
          if (mover) __hip_atomic_fetch_or(&sh.lane_mask[me[h].tgt][r >> 6], (u64)1 << (r & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          insane = insane || (active[h] && !(me[h].v * me[h].ch >= 0.0 && fabs(me[h].v * me[h].sh) <= 5.0));
        }
        const bool sane = __ballot(insane) == 0;
        HWY_WAVE_LDS_FENCE();
        u64 Rem[K][K], Bc[K][K];
#pragma unroll
        for (int h = 0; h < K; ++h) {
#pragma unroll
          for (int w = 0; w < K; ++w) {
            const u64 row = sh.lane_mask[tgt_old[h]][w];  // (every slot's target lane is a valid row)
            Rem[h][w] = changer[h] ? row : 0;
            Bc[h][w] = 0;
          }
        }
        bool fixed[K], any_left = false;
        double bound[K];
#pragma unroll
        for (int h = 0; h < K; ++h) {
          const int rw = rank[h] >> 6, rb = rank[h] & 63;
#pragma unroll
          for (int w = 0; w < K; ++w) {
            const u64 ahead = (w > rw) ? ~(u64)0 : ((w == rw) ? ~(((u64)2 << rb) - 1) : 0);  // ranks above mine (2 << 63 wraps to 0)
            Rem[h][w] &= ahead;
            any_left = any_left || Rem[h][w] != 0;
          }
          fixed[h] = false;
          // d*(c, r) = 10 + 1.5 v + v dv / (2 sqrt(ab)) with dv = v (cc^2 + sc^2) - v_r (c_r cc + s_r sc) <= v + 5 + (rounding) when v_r c_r >= 0,
          // |v_r s_r| <= 5 (sane) and cc >= 0, v >= 0; 1e-6 relative + absolute on top of the bound, far above any rounding in d*
          const double v = me[h].v;
          bound[h] = (sane && v >= 0.0 && me[h].ch >= 0.0)
                         ? (HWY_DISTANCE_WANTED + v * HWY_TIME_WANTED + v * (v + 5.0) * 0.12909944487358055) * (1.0 + 1e-6) + 1e-6
                         : __builtin_inf();
        }
        // (2) the walk: nearest remaining member ahead first; beyond the bound everything farther is beyond it too
        while (__ballot(any_left) != 0) {  // wave-uniform
          any_left = false;
#pragma unroll
          for (int h = 0; h < K; ++h) {
            int rr = 0;
            bool go = false;
#pragma unroll
            for (int w = 0; w < K; ++w) {
              const bool take = !go && Rem[h][w] != 0;
              rr = take ? w * 64 + ctz64(Rem[h][w]) : rr;
              Rem[h][w] = take ? (Rem[h][w] & (Rem[h][w] - 1)) : Rem[h][w];
              go = go || take;
            }
            const double xr = sh.x[rr], vr = sh.v[rr], cr = sh.c[rr], sr = sh.s[rr];
            const int ir = sh.idx[rr], fl_r = sh.sbits[rr];  // fl_r: the rival decided in this frame | is a changer << 1
            const double d = xr - me[h].x;
            const bool inside = go && d < bound[h];
            // the target r shows to c: its current one if it comes before c in the list, else the frame-start one -- and a vehicle
            // that decided in this very frame headed nowhere with that one
            const bool valid = inside && (ir < vi[h] || !(fl_r & 1));
            const double d_star = B::desired_gap(me[h].v, me[h].ch, me[h].sh, vr, cr, sr);
            const bool blk = valid && (0 < d) && (d < d_star);
            const bool cond = ir < vi[h] && (fl_r & 2) != 0;  // an earlier changer: it may abort
            fixed[h] = fixed[h] || (blk && !cond);
#pragma unroll
            for (int w = 0; w < K; ++w) {
              Bc[h][w] |= (blk && cond && (ir >> 6) == w) ? ((u64)1 << (ir & 63)) : 0;
              Rem[h][w] = (go && !inside) || fixed[h] ? 0 : Rem[h][w];
              any_left = any_left || Rem[h][w] != 0;
            }
          }
        }
        // (3) the changers that abort (index space: the ballot of slot w is word w), to the fixed point
        bool any_blk = false;
#pragma unroll
        for (int h = 0; h < K; ++h) {
          any_blk = any_blk || fixed[h];
#pragma unroll
          for (int w = 0; w < K; ++w) any_blk = any_blk || Bc[h][w] != 0;
        }
        if (__ballot(any_blk) != 0) {  // wave-uniform
          u64 A[K];
#pragma unroll
          for (int w = 0; w < K; ++w) A[w] = 0;
          for (;;) {
            u64 nA[K];
            bool same = true;
#pragma unroll
            for (int h = 0; h < K; ++h) {
              bool fire = fixed[h];
#pragma unroll
              for (int w = 0; w < K; ++w) fire = fire || (Bc[h][w] & ~A[w]) != 0;
              nA[h] = __ballot(fire);
            }
#pragma unroll
            for (int w = 0; w < K; ++w) {
              same = same && nA[w] == A[w];
              A[w] = nA[w];
            }
            if (same) break;
          }
#ifndef HWY_WIDE_MUTANT_NO_ABORT  // (tests/test_wide_kernel.py: a build that never applies the verdict must fail the comparison)
#pragma unroll
          for (int h = 0; h < K; ++h)
            if ((A[h] >> l) & 1) me[h].tgt = me[h].lane;  // abort
#endif
        }
      }
    }

    // ---- E. Road.act: low-level control, F. Road.step: integrate ---------------------------------------------
    double x_old[K], tb_[K];
    if constexpr (K == 2) {  // the steering chain of both vehicles at once (steer_tan_beta2)
      double iv0, iv1;
      fast_rcp2(not_zero(me[0].v), not_zero(me[1].v), iv0, iv1);
      steer_tan_beta2(p, me[0].y, me[0].h, iv0, me[0].tgt, me[1].y, me[1].h, iv1, me[1].tgt, tb_[0], tb_[1]);
    } else {
#pragma unroll
      for (int h = 0; h < K; ++h) tb_[h] = B::steer_tan_beta(p, me[h].y, me[h].h, fast_rcp(not_zero(me[h].v)), me[h].tgt);
    }
#pragma unroll
    for (int h = 0; h < K; ++h) {
      Veh &mv = me[h];
      double tb = tb_[h];
      double accel = free_self[h] - gap_own[h];
      {
        // leader on the target lane: that lane's mask for an ongoing change, the left / right lane evaluated above otherwise
        // (gathered unconditionally -- a conditional LDS read is a branch with its own round trip -- and selected)
        // (decided just now: the gap term towards the new target lane's leader is the one MOBIL's incentive test evaluated -- 0.0
        //  without a leader, and free_self - 0.0 == free_self)
        const int f2 = ft[h];
        const int g2 = f2 < 0 ? 0 : f2;
        const double g2x = sh.x[g2], g2v = sh.v[g2], g2c = sh.c[g2], g2s = sh.s[g2];
        const double a2_t = f2 >= 0 ? free_self[h] - B::idm_gap(mv.x, mv.v, mv.ch, mv.sh, g2x, g2v, g2c, g2s) : free_self[h];
        const double a2 = (mv.tgt == tgt_old[h]) ? a2_t : free_self[h] - gap_new[h];
        accel = (drives[h] && mv.lane != mv.tgt && a2 < accel) ? a2 : accel;  // Python min(a, b)
      }
      accel = clipd(accel, -HWY_ACC_MAX, HWY_ACC_MAX);
      accel = controlled[h] ? HWY_KP_A * (mv.ts - mv.v) : accel;  // speed_control (controller.py:189-198), not clipped

      x_old[h] = mv.x;
      // clip_actions (kinematics.py:155-168): a crashed vehicle has steering 0 (tan(beta) = 0), accel = -speed
      tb = crashed0[h] ? 0.0 : tb;
      accel = crashed0[h] ? -1.0 * mv.v : accel;
      accel = (mv.v > HWY_MAX_SPEED) ? fmin(accel, 1.0 * (HWY_MAX_SPEED - mv.v))
                                     : ((mv.v < HWY_MIN_SPEED) ? fmax(accel, 1.0 * (HWY_MIN_SPEED - mv.v)) : accel);
      const double cb = fast_rsqrt(1.0 + tb * tb), sb = tb * cb;
      const double vx = mv.v * (mv.ch * cb - mv.sh * sb), vy = mv.v * (mv.sh * cb + mv.ch * sb);
      mv.x += vx * p.dt;
      mv.y += vy * p.dt;
      if (mv.flags & HWY_F_HAS_IMPACT) {
        mv.x += mv.impx;
        mv.y += mv.impy;
        mv.flags = (mv.flags | HWY_F_CRASHED) & ~HWY_F_HAS_IMPACT;
        mv.impx = mv.impy = 0.0;
      }
      mv.h += mv.v * sb * (1.0 / (HWY_VEH_LENGTH / 2)) * p.dt;
      mv.v += accel * p.dt;
      mv.lane = B::closest_lane(p, mv.x, mv.y, mv.h);
      if constexpr (K != 2) sincos_bounded(mv.h, &mv.sh, &mv.ch);
    }
    if constexpr (K == 2) sincos_bounded2(me[0].h, me[1].h, &me[0].sh, &me[0].ch, &me[1].sh, &me[1].ch);

    // ---- G. Road.step: collisions (road.py:477-481, objects.py:92-138) -----------------------------------
    if (all_check) {
      // Full pairwise (highway-v0).  A pair can only collide if it is within ~5.5 m + |v| dt, i.e. among neighbours along the
      // road: every vehicle walks FORWARD in the rank order of this frame's start (each unordered pair is met once, from its
      // rear end), bounded by the frame-start distance (collision radius + the most two vehicles can move relative to each
      // other within one frame), and only COLLECTS the partners inside the reference's own pre-check sphere
      // (objects.py:124-127) -- a dozen VALU instructions per candidate.  The list pass then runs, one PAIR per thread, the
      // provable-separation test and -- if any pair of the wavefront survives it -- the SAT; the verdicts meet per vehicle in LDS
      // ("last pair in loop order wins" == ds_max on the partner index, then the winner's write: hwy_wave.h has the
      // argument).  The list is a ring.
      HWY_WAVE_LDS_FENCE();  // every gather of the frame-start snapshot is done: all of it but x and idx is dead from here on
      unsigned key_d = 0, key_v = 0;
#pragma unroll
      for (int h = 0; h < K; ++h) {
        const int v = vi[h], r = rank[h];
        sh.nx[v] = me[h].x; sh.ny[v] = me[h].y; sh.nv[v] = me[h].v; sh.nc[v] = me[h].ch; sh.ns[v] = me[h].sh;
        // position and speed again in the rank order of this frame's start, over dead planes of the snapshot (lr <- x, c <- y,
        // v): a walk step then needs ONE LDS round trip (frame-start x, index, position, speed of slot rank + k)
        if (active[h]) { sh.lr[r] = me[h].x; sh.c[r] = me[h].y; sh.v[r] = me[h].v; }
        sh.jmax[v] = -1;
        sh.hit[v] = 0;
        // this frame's largest displacement along x and largest speed (hwy_device.h: reach_key)
        const unsigned kd_ = active[h] ? reach_key(me[h].x - x_old[h]) : 0u, kv_ = active[h] ? reach_key(me[h].v) : 0u;
        key_d = kd_ > key_d ? kd_ : key_d;
        key_v = kv_ > key_v ? kv_ : key_v;
      }
      key_d = HWY_WAVE_MAX_U32(key_d);
      key_v = HWY_WAVE_MAX_U32(key_v);
      HWY_WAVE_LDS_FENCE();
      // radius + relative motion from the ACTUAL maxima of this frame (hwy_device.h: reach_from_keys)
      const double reach = reach_from_keys(key_d, key_v, p.dt);
      const u64 below = ((u64)1 << l) - 1;
      constexpr int PASS = 64 * K, RING = 512;  // at most PASS - 1 + 2 steps x 64 K entries are pending at any time
      int n_list = 0, head = 0, k = 1;  // wave-uniform
      bool go[K], walking = true;
#pragma unroll
      for (int h = 0; h < K; ++h) go[h] = active[h];
      while (walking || n_list) {
        while (walking && n_list < PASS) {
          // two walk steps per trip (k and k + 1, K slots): 2 K independent candidates whose LDS reads are in flight together
          // (every LDS read of the trip is issued before anything depends on one: the slots are clamped by the range alone, the
          // `go` flags -- which depend on what is read -- are folded in afterwards; a conditional read would be a branch with
          // its own round trip)
          bool keep[2][K], going = false;
          int q[2][K], rb[2][K];
          double x0[2][K], px[2][K], py[2][K], pv[2][K];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int h = 0; h < K; ++h) {
              rb[u][h] = rank[h] + (k + u);
              const int r = rb[u][h] < N ? rb[u][h] : 0;
              x0[u][h] = sh.x[r];  // frame-start x in rank order
              px[u][h] = sh.lr[r]; py[u][h] = sh.c[r]; pv[u][h] = sh.v[r];
              q[u][h] = sh.idx[r];
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int h = 0; h < K; ++h) {
              const bool near = !(fabs(x0[u][h] - x_old[h]) > reach);
              go[h] = go[h] & (rb[u][h] < N) & near;
              if (u == 1) going = going || go[h];
              const double dx = px[u][h] - me[h].x, dy = py[u][h] - me[h].y;
              const double lim = 5.5 + fmax(fabs(me[h].v), fabs(pv[u][h])) * p.dt;
              keep[u][h] = go[h] & !(dx * dx + dy * dy > lim * lim);
            }
          }
          k += 2;
          if (__ballot(going) == 0 || k > N) walking = false;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int h = 0; h < K; ++h) {
              const u64 km = __ballot(keep[u][h]);
              if (km) {
                if (keep[u][h]) sh.plist[(head + n_list + __popcll(km & below)) & (RING - 1)] = (unsigned short)(vi[h] | (q[u][h] << 8));
                n_list += __popcll(km);
              }
            }
          }
        }
        const int count = n_list < PASS ? n_list : PASS;
        HWY_WAVE_LDS_FENCE();
        int pa[K], pb[K], r[K];
        double tx[K], ty[K];
        bool cand[K], any_cand = false;
        Body A[K], Bb[K];
#pragma unroll
        for (int h = 0; h < K; ++h) {
          r[h] = 0;
          tx[h] = ty[h] = 0.0;
          cand[h] = false;
          pa[h] = pb[h] = 0;
          if (count > 64 * h) {  // wave-uniform: the second slot only when the list holds more than 64 pairs
            const int t = h * 64 + l;
            const int pair = t < count ? (int)sh.plist[(head + t) & (RING - 1)] : -1;
            const int u0 = pair < 0 ? 0 : (pair & 255), u1 = pair < 0 ? 0 : ((pair >> 8) & 255);  // (no pair: slot 0, discarded)
            const int a = u0 < u1 ? u0 : u1, b = u0 < u1 ? u1 : u0;  // a < b: the reference's `self` and `other`
            pa[h] = a;
            pb[h] = b;
            A[h] = Body{sh.nx[a], sh.ny[a], sh.nv[a], sh.nc[a], sh.ns[a]};
            Bb[h] = Body{sh.nx[b], sh.ny[b], sh.nv[b], sh.nc[b], sh.ns[b]};
            cand[h] = pair >= 0 && !surely_apart(A[h], Bb[h], p.dt);
            any_cand = any_cand || cand[h];
          }
        }
        const bool any_sat = __ballot(any_cand) != 0;  // wave-uniform
        if (any_sat) {
#pragma unroll
          for (int h = 0; h < K; ++h) {
            if (count > 64 * h && cand[h]) {
              const int a = pa[h], b = pb[h];
              r[h] = pair_collide(A[h], Bb[h], p.dt, &tx[h], &ty[h]);
              if (r[h] & 1) sh.hit[a] = sh.hit[b] = 1;
              if (r[h] & 2) {  // "last pair in loop order wins" == the partner with the highest index
                __hip_atomic_fetch_max(&sh.jmax[a], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&sh.jmax[b], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
          HWY_WAVE_LDS_FENCE();
#pragma unroll
          for (int h = 0; h < K; ++h) {
            if (r[h] & 2) {
              const int a = pa[h], b = pb[h];
              if (sh.jmax[a] == b) { sh.impx[a] = tx[h] / 2; sh.impy[a] = ty[h] / 2; }
              if (sh.jmax[b] == a) { sh.impx[b] = -tx[h] / 2; sh.impy[b] = -ty[h] / 2; }
            }
          }
        }
        head = (head + count) & (RING - 1);
        n_list -= count;
        HWY_WAVE_LDS_FENCE();
      }
#pragma unroll
      for (int h = 0; h < K; ++h) {
        // (the four reads are issued together; the translation slots only hold something where a pair of THIS frame wrote it)
        const int jm = sh.jmax[vi[h]], ht = sh.hit[vi[h]];
        const double ix = sh.impx[vi[h]], iy = sh.impy[vi[h]];
        const bool pushed = active[h] && jm >= 0;
        me[h].impx = pushed ? ix : me[h].impx;
        me[h].impy = pushed ? iy : me[h].impy;
        if (pushed) me[h].flags |= HWY_F_HAS_IMPACT;
        if (active[h] && ht) me[h].flags |= HWY_F_CRASHED;
      }
    } else {
      // sparse checkers (highway-fast-v0 semantics: the controlled vehicles only), ascending index == loop order
#pragma unroll
      for (int hc = 0; hc < K; ++hc) {
        u64 cm = chk[hc];
        while (cm) {  // wave-uniform
          const int lc = ctz64(cm), c = hc * 64 + lc;
          cm &= cm - 1;
          const Body other{wave_bcast(me[hc].x, lc), wave_bcast(me[hc].y, lc), wave_bcast(me[hc].v, lc), wave_bcast(me[hc].ch, lc),
                           wave_bcast(me[hc].sh, lc)};
          int r[K];
          double tx[K], ty[K];
          u64 wm[K], im[K];
#pragma unroll
          for (int h = 0; h < K; ++h) {
            r[h] = 0;
            tx[h] = ty[h] = 0.0;
            if (active[h] && vi[h] != c) {
              const Body mine{me[h].x, me[h].y, me[h].v, me[h].ch, me[h].sh};
              const double dx = other.x - mine.x, dy = other.y - mine.y;
              const double lim = 5.5 + fmax(fabs(mine.v), fabs(other.v)) * p.dt;
              if (dx * dx + dy * dy <= lim * lim) {
                const bool i_first = vi[h] < c;
                const Body A = select_body(i_first, mine, other), Bb = select_body(i_first, other, mine);
                if (!surely_apart(A, Bb, p.dt)) {
                  r[h] = pair_collide(A, Bb, p.dt, &tx[h], &ty[h]);
                  if (!i_check[h]) {  // my only partners are the checkers (ascending c == loop order)
                    if (r[h] & 2) {
                      me[h].impx = i_first ? tx[h] / 2 : -tx[h] / 2;
                      me[h].impy = i_first ? ty[h] / 2 : -ty[h] / 2;
                      me[h].flags |= HWY_F_HAS_IMPACT;
                    }
                    if (r[h] & 1) me[h].flags |= HWY_F_CRASHED;
                  }
                }
              }
            }
            wm[h] = __ballot((r[h] & 2) != 0);
            im[h] = __ballot((r[h] & 1) != 0);
          }
          if (wide_any<K>(wm) || wide_any<K>(im)) {  // wave-uniform: the checker gathers from its partners
            int q = 0;  // last partner in loop order
#pragma unroll
            for (int h = 0; h < K; ++h) q = wm[h] ? h * 64 + msb64(wm[h]) : q;
            const double qx = wide_bcast<K>([&](int h) { return tx[h]; }, q), qy = wide_bcast<K>([&](int h) { return ty[h]; }, q);
            if (l == lc) {
              if (wide_any<K>(im)) me[hc].flags |= HWY_F_CRASHED;
              if (wide_any<K>(wm)) {
                me[hc].impx = (c < q) ? qx / 2 : -qx / 2;
                me[hc].impy = (c < q) ? qy / 2 : -qy / 2;
                me[hc].flags |= HWY_F_HAS_IMPACT;
              }
            }
          }
        }
      }
    }
  }  // frames

  // ---- H. observe / reward / done ---------------------------------------------------------------------------
  HWY_RELOAD_PARAMS(q, p);
  if (q.full_step) {
    wide_update_rank<K>(sh, me, N, rank, has_tie);  // positions moved in the last frame
    observe_wide<K, true>(q, sh, e, eo, me, true, rank, env_time);
  }
#pragma unroll
  for (int h = 0; h < K; ++h) {
    me[h].rank = rank[h];
    store_vehicle_at(q, e, vi[h], me[h], false);
  }
}

// WPE = resident wavefronts per SIMD the register allocator must leave room for (launch_bounds).
template <int K, int WPE>
__global__ void __launch_bounds__(64, WPE) hwy_step_wide_kernel(const StepParams p) {
  __shared__ WideShared<K> sh;
  HWY_KERNARG_TOUCH(StepParams);
  wide_policy_step<K>(p, sh, blockIdx.x, blockIdx.x);
}

// hwy_rollout_device: p.k_steps consecutive policy steps of every environment in one launch (hwy_wave.h: hwy_rollout_wave_kernel)
template <int K, int WPE>
__global__ void __launch_bounds__(64, WPE) hwy_rollout_wide_kernel(const StepParams p) {
  __shared__ WideShared<K> sh;
  HWY_KERNARG_TOUCH(StepParams);
  const int e = blockIdx.x;
  for (int k = 0; k < p.k_steps; ++k) {  // wave-uniform
    HWY_RELOAD_PARAMS(pk, p);
    wide_policy_step<K>(pk, sh, e, k * pk.num_envs + e);
    HWY_WAVE_LDS_FENCE();
    __threadfence_block();  // the next step's loads follow this step's stores (same wavefront, same addresses)
  }
}

}  // namespace hwy

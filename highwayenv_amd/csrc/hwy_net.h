// hwy_net.h -- the fused policy-step kernel for ROAD-NETWORK scenarios (MergeEnv / MergeGenericEnv,
// highway_env/envs/merge_env.py): ONE 64-wide wavefront per environment, thread i == slot i of the vehicle
// arrays (vehicles in Road.vehicles order, HWY_F_ABSENT holes, then the Obstacle of Road.objects).
//
// Same wave formulation as hwy_wave.h -- rank of every vehicle along x, rank-space membership mask per lane
// (every vehicle ORs its rank bit into the masks that read its lanes), front / rear neighbour == two bit scans + one LDS gather, ordered readlane chains for
// the sequential bits of the reference's Python loops -- generalised from "L parallel lanes of one road" to a
// table of x-aligned lanes (hwy_config.net -> NetParams::lane -> LDS):
//   * every lane has its own start / length (AbstractLane.on_lane, lane.py:80-102) and, for the SineLane of
//     the access ramp, a lateral offset amplitude*sin(pulsation*s + phase) (lane.py:236-283);
//   * because every lane's direction is (1, 0) the longitudinal coordinate on ANY lane is x - x0, so one sort
//     by x still orders every lane's members;
//   * ControlledVehicle.follow_road / RoadNetwork.next_lane switch the target lane at the end of a segment
//     (controller.py:135-143, road.py:73-146); forbidden lanes are never reachable (lane.py:110-111); the
//     lane-change abort rule applies on the same road only (behavior.py:232); IDM clips the target speed to the
//     limit of the ego's current lane (behavior.py:171-175);
//   * the Obstacle (2 m x 2 m, objects.py:25-26,215-222) is a member of lane masks, a collision partner
//     (a vehicle hitting it takes the WHOLE translation, objects.py:106-107) and observable, but never acts;
//   * absent slots (MergeGenericEnv's rejection-sampled spawn, merge_env.py:336-352) take no part in anything.
#pragma once

#include "hwy_device.h"
#include "hwy_wave.h"

namespace hwy {

struct NetParams {
  StepParams s;                    // must stay first (load_vehicle / store_vehicle / observe helpers take it)
  int32_t n_lanes, merge_lane, generic, pad_;  // generic: MergeGenericEnv's spawn rule (device reset)
  double merge_end_x, merging_speed_reward, lane_change_reward;
  hwy_lane lane[HWY_MAX_LANES];
};

// one row of the lane table as the per-frame table walk reads it (three ds_read_b128 with a wave-uniform address)
struct alignas(16) NetLaneRow {
  double x0, y0, length, len5, hw1, pad;  // len5 = length + 5 (on_lane's upper bound), hw1 = width / 2 + 1
};
struct NetShared {
  NetLaneRow row[HWY_MAX_LANES];
  // lane table (struct of arrays: per-thread lane indices read it with one ds_read each)
  double lx0[HWY_MAX_LANES], ly0[HWY_MAX_LANES], llen[HWY_MAX_LANES], lwid[HWY_MAX_LANES], lamp[HWY_MAX_LANES],
      lpuls[HWY_MAX_LANES], lphase[HWY_MAX_LANES], llimit[HWY_MAX_LANES];
  int lroad[HWY_MAX_LANES], lid[HWY_MAX_LANES], lfirst[HWY_MAX_LANES], lcount[HWY_MAX_LANES], lnext[HWY_MAX_LANES],
      lnextn[HWY_MAX_LANES], lforb[HWY_MAX_LANES], lconn[HWY_MAX_LANES], linv[HWY_MAX_LANES];  // linv[L] = {i : lconn[i] has L}
  // frame-start snapshot in RANK order
  double x[64], v[64], c[64], s[64], lr[64], ox[64];  // lr = log(v/v0) (IDM), ox = x0 of the vehicle's own lane
  int idx[64], kind[64];                              // kind: 1 = vehicle, 0 = obstacle
  u64 lane_mask[HWY_MAX_LANES];
  // post-integration bodies by slot index (collisions)
  double nx[64], ny[64], nv[64], nc[64], ns[64];
  double scratch[64 + 8];  // device spawn
};

// ---- lane geometry from the LDS table (per-thread lane index L) ---------------------------------------------
// lateral coordinate on lane L (StraightLane / SineLane.local_coordinates); s = x - x0[L]
__device__ inline double net_lat(const NetShared &sh, int L, double s, double y) {
  double lat = y - sh.ly0[L];
  const double amp = sh.lamp[L];
  if (amp != 0.0) {
    double sn, cs;
    sincos_bounded(sh.lpuls[L] * s + sh.lphase[L], &sn, &cs);
    lat = lat - amp * sn;
  }
  return lat;
}
// atan for the lane slope amplitude*pulsation*cos(.) (|t| <= 0.13 on the merge ramps): fdlibm s_atan.c through
// hwy_math.h's atan_fd (all argument ranges, no library call)
__device__ inline double atan_small(double x) { return atan_fd(x); }
// heading of lane L at longitudinal s (StraightLane.heading_at == 0; SineLane.heading_at, lane.py:259-265)
__device__ inline double net_heading_at(const NetShared &sh, int L, double s) {
  const double amp = sh.lamp[L];
  if (amp == 0.0) return 0.0;
  double sn, cs;
  sincos_bounded(sh.lpuls[L] * s + sh.lphase[L], &sn, &cs);
  return 0.0 + atan_small(amp * sh.lpuls[L] * cs);
}
// AbstractLane.is_reachable_from (lane.py:104-118)
__device__ inline bool net_reachable(const NetShared &sh, int L, double x, double y) {
  if (sh.lforb[L]) return false;
  const double s = x - sh.lx0[L];
  const double lat = net_lat(sh, L, s, y);
  return fabs(lat) <= 2 * sh.lwid[L] && 0 <= s && s < sh.llen[L] + 5.0;
}
// ControlledVehicle.follow_road (controller.py:135-143) + RoadNetwork.next_lane with route=None (road.py:73-146)
__device__ inline int net_follow_road(const NetShared &sh, int tgt, double x, double y) {
  const double s = x - sh.lx0[tgt];
  if (!(s > sh.llen[tgt] - 5.0 / 2)) return tgt;  // AbstractLane.after_end (lane.py:120-125)
  const int nf = sh.lnext[tgt];
  if (nf < 0) return tgt;  // KeyError on graph[_to] -> current index
  if (sh.lcount[tgt] == sh.lnextn[tgt]) return nf + sh.lid[tgt];
  // projected position lane.position(s, 0); closest lane of the next road by AbstractLane.distance (first minimum)
  double py = sh.ly0[tgt];
  if (sh.lamp[tgt] != 0.0) {
    double sn, cs;
    sincos_bounded(sh.lpuls[tgt] * s + sh.lphase[tgt], &sn, &cs);
    py = sh.ly0[tgt] + (0.0 + sh.lamp[tgt] * sn);
  }
  const double px = sh.lx0[tgt] + s;
  int best = 0;
  double bd = 0.0;
  for (int k = 0; k < sh.lnextn[tgt]; ++k) {
    const int L = nf + k;
    const double s2 = px - sh.lx0[L];
    const double r = net_lat(sh, L, s2, py);
    const double d = fabs(r) + fmax(s2 - sh.llen[L], 0.0) + fmax(0 - s2, 0.0);
    if (k == 0 || d < bd) { bd = d; best = k; }
  }
  return nf + best;
}
// RoadNetwork.get_closest_lane_index (road.py:55-71) with distance_with_heading (lane.py:132-147): first minimum
// in table order.  n_lanes is wave-uniform; the straight lanes share the heading term.
__device__ inline int net_closest_lane(const NetShared &sh, int n_lanes, double x, double y, double h) {
  const double angle0 = fabs(wrap_to_pi(h - 0.0));
  int best = 0;
  double bd = 0.0;
  for (int L = 0; L < n_lanes; ++L) {  // wave-uniform L: LDS broadcasts
    const double s = x - sh.lx0[L];
    double r = y - sh.ly0[L], angle = angle0;
    const double amp = sh.lamp[L];
    if (amp != 0.0) {  // wave-uniform branch
      double sn, cs;
      sincos_bounded(sh.lpuls[L] * s + sh.lphase[L], &sn, &cs);
      r = r - amp * sn;
      angle = fabs(wrap_to_pi(h - (0.0 + atan_small(amp * sh.lpuls[L] * cs))));
    }
    const double d = fabs(r) + fmax(s - sh.llen[L], 0.0) + fmax(0 - s, 0.0) + 1.0 * angle;
    if (L == 0 || d < bd) { bd = d; best = L; }
  }
  return best;
}

// ---- IDM pieces with an explicit lane distance d (objects.py:183-198: along the EGO's current lane) -----------
__device__ inline double net_gap_term(double d, double ve, double ce, double se, double vf, double cf, double sf) {
  const double q = EnvBlock<1>::desired_gap(ve, ce, se, vf, cf, sf) * fast_rcp(not_zero(d));
  return HWY_COMFORT_ACC_MAX * (q * q);
}
__device__ inline double net_log_ratio(double v, double ts, double limit) {
  const double v0 = clipd(ts, 0.0, limit);
  const double r = fmax(v, 0.0) * fast_rcp(abs_not_zero(v0));
  return r > 0.0 ? log_pos(r) : -__builtin_inf();
}
// steering_control (controller.py:145-187) folded with the slip / bicycle chain like EnvBlock::steer_tan_beta,
// for a lane whose lateral offset and look-ahead heading are given
__device__ inline double net_steer_tan_beta(double lat, double lane_heading, double h, double inv_v) {
  const double a = clipd((-HWY_KP_LATERAL * lat) * inv_v, -1.0, 1.0);
  const double s45 = 0.7071067811865476;
  const double hc = a >= s45 ? HWY_PI / 4 : (a <= -s45 ? -HWY_PI / 4 : clipd(asin_bounded(a), -HWY_PI / 4, HWY_PI / 4));
  const double heading_ref = lane_heading + hc;
  const double heading_rate_command = HWY_KP_HEADING * wrap_to_pi(heading_ref - h);
  const double w = clipd((HWY_VEH_LENGTH / 2 * inv_v) * heading_rate_command, -1.0, 1.0);
  const double tan_max = 1.7320508075688767;
  const double w2 = 1 - w * w;
  const double tan_steer = (w2 <= 1e-12) ? copysign(tan_max, w) : clipd((2 * w) * fast_rsqrt(w2), -tan_max, tan_max);
  return 0.5 * tan_steer;
}

// ---- rectangles of different sizes (vehicle 5 x 2, obstacle 2 x 2): the SAT of hwy_device.h with per-body
//      half extents; a = the reference's `self` (lower slot), b = `other` -----------------------------------------
struct NetBody {
  double x, y, v, c, s, hl, hw;
};
__device__ inline NetBody select_nbody(bool first, const NetBody &p, const NetBody &q) {
  return NetBody{first ? p.x : q.x, first ? p.y : q.y, first ? p.v : q.v, first ? p.c : q.c, first ? p.s : q.s,
                 first ? p.hl : q.hl, first ? p.hw : q.hw};
}
__device__ inline bool net_surely_apart(const NetBody &A, const NetBody &B, double dt) {
  const double dx = B.x - A.x, dy = B.y - A.y;
  const double cr = fabs(A.c * B.c + A.s * B.s), sr = fabs(B.s * A.c - B.c * A.s);
  const double rvx = (A.v * A.c - B.v * B.c) * dt, rvy = (A.v * A.s - B.v * B.s) * dt;
  const double margin = 1e-6;
  const double gap_lat = fabs(-A.s * dx + A.c * dy) - A.hw - (B.hl * sr + B.hw * cr) - fabs(-A.s * rvx + A.c * rvy);
  const double gap_lon = fabs(A.c * dx + A.s * dy) - A.hl - (B.hl * cr + B.hw * sr) - fabs(A.c * rvx + A.s * rvy);
  return gap_lat > margin || gap_lon > margin;
}
__device__ inline int net_pair_collide(const NetBody &A, const NetBody &B, double dt, double *tx, double *ty) {
  // np.linalg.norm([LENGTH, WIDTH]) of a Vehicle (5 x 2) or an Obstacle (2 x 2): two constants, not two square roots
  const double diag_veh = sqrt(HWY_VEH_LENGTH * HWY_VEH_LENGTH + HWY_VEH_WIDTH * HWY_VEH_WIDTH), diag_obs = sqrt(2.0 * 2.0 + 2.0 * 2.0);
  const double diag_a = A.hl == 1.0 ? diag_obs : diag_veh;
  const double diag_b = B.hl == 1.0 ? diag_obs : diag_veh;
  const double dx = B.x - A.x, dy = B.y - A.y;
  *tx = 0;
  *ty = 0;
  if (sqrt(dx * dx + dy * dy) > (diag_a + diag_b) / 2 + A.v * dt) return 0;  // objects.py:124-127
  const double ddx = A.v * A.c * dt - B.v * B.c * dt, ddy = A.v * A.s * dt - B.v * B.s * dt;
  const double cdx = A.x - B.x, cdy = A.y - B.y;
  const double cr = fabs(A.c * B.c + A.s * B.s), sr = fabs(B.s * A.c - B.c * A.s);
  SatAcc acc{3, __builtin_inf(), 0.0, 0.0, 8};
  // the reference's order of the 8 normals (hwy_device.h: sat_axis): -u_a, +w_a, +u_a, -w_a, -u_b, +w_b, +u_b, -w_b
  sat_axis(acc, A.c, A.s, A.x * A.c + A.y * A.s, A.hl, B.x * A.c + B.y * A.s, B.hl * cr + B.hw * sr, A.c * ddx + A.s * ddy, cdx, cdy, 2, 0);
  HWY_SAT_FENCE();
  sat_axis(acc, -A.s, A.c, A.y * A.c - A.x * A.s, A.hw, B.y * A.c - B.x * A.s, B.hl * sr + B.hw * cr, A.c * ddy - A.s * ddx, cdx, cdy, 1, 3);
  HWY_SAT_FENCE();
  sat_axis(acc, B.c, B.s, A.x * B.c + A.y * B.s, A.hl * cr + A.hw * sr, B.x * B.c + B.y * B.s, B.hl, B.c * ddx + B.s * ddy, cdx, cdy, 6, 4);
  HWY_SAT_FENCE();
  sat_axis(acc, -B.s, B.c, A.y * B.c - A.x * B.s, A.hl * sr + A.hw * cr, B.y * B.c - B.x * B.s, B.hw, B.c * ddy - B.s * ddx, cdx, cdy, 5, 7);
  if (acc.flags & 2) {
    *tx = acc.min_distance * acc.axx;
    *ty = acc.min_distance * acc.axy;
  }
  return acc.flags;
}

// ---- rank along x among the PRESENT slots (0 = smallest x; equal x ordered by slot); absent / idle lanes take
//      the remaining ranks so that the ds_permute sends stay a bijection ------------------------------------------
__device__ inline void net_rank(double x, bool present, u64 pm, int &rank, bool &has_tie) {
  const int i = threadIdx.x;
  int cnt = 0;
  bool tie = false;
  for (u64 m = pm; m; m &= m - 1) {  // wave-uniform
    const int j = ctz64(m);
    const double xj = wave_bcast(x, j);
    cnt += (xj < x || (xj == x && j < i)) ? 1 : 0;
    tie = tie || (xj == x && j != i);
  }
  has_tie = __ballot(present && tie) != 0;
  const u64 below = ((u64)1 << i) - 1;
  rank = present ? cnt : __popcll(pm) + __popcll(~pm & below);
}

// The rank of the previous frame, re-validated like hwy_wave.h's wave_update_rank: every present slot sends its x to
// lanes `rank` and `rank - 1` (ds_permute); lane r then holds x of rank r and of rank r + 1 and checks the order.
// Disjoint adjacent inversions (an overtake somewhere on the road) are repaired by swapping the two ranks;
// anything else (overlapping inversions, equal x) goes back to the counting pass.
// `trusted` = false (wave-uniform): `rank` is not known to be a permutation (a stale hint): count.
__device__ inline void net_update_rank(double x, bool present, u64 pm, int n_present, int &rank, bool &has_tie, bool trusted = true) {
  const int i = threadIdx.x;
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const double x_r = __hiloint2double(wave_send_i(hi, rank), wave_send_i(lo, rank));
  const double x_r1 = __hiloint2double(wave_send_i(hi, rank - 1), wave_send_i(lo, rank - 1));
  bool recount = !trusted || __ballot(i < n_present - 1 && !(x_r < x_r1)) != 0;
  if (recount && trusted) {  // wave-uniform
    const u64 inv = __ballot(i < n_present - 1 && x_r > x_r1);
    if (inv != 0 && (inv & (inv << 1)) == 0) {
      const bool up = present && ((inv >> rank) & 1);
      const bool down = present && rank > 0 && ((inv >> (rank - 1)) & 1);
      rank += up ? 1 : (down ? -1 : 0);
      const double y_r = __hiloint2double(wave_send_i(hi, rank), wave_send_i(lo, rank));
      const double y_r1 = __hiloint2double(wave_send_i(hi, rank - 1), wave_send_i(lo, rank - 1));
      recount = __ballot(i < n_present - 1 && !(y_r < y_r1)) != 0;
    }
  }
  if (!recount) has_tie = false;  // strictly increasing => all x distinct
  if (recount) net_rank(x, present, pm, rank, has_tie);
}

// One pass over the lane table for a body at (x, y) with heading h:
//   * bits:    AbstractLane.on_lane with margin 1 (lane.py:80-102) for every lane -- what Road.neighbour_vehicles
//              tests for each candidate (road.py:503-519);
//   * closest: RoadNetwork.get_closest_lane_index (road.py:55-71) with distance_with_heading (lane.py:132-147),
//              first minimum in table order; the straight lanes share the heading term.
// Straight lanes are walked first (branch-free), the SineLanes (bit set in sine_mask) afterwards.
template <bool CLOSEST>
__device__ inline void net_lane_pass(const NetParams &np, const NetShared &sh, unsigned sine_mask, bool present, double x,
                                     double y, double h, int *bits_out, int *closest_out) {
  const double angle0 = CLOSEST ? fabs(wrap_to_pi(h - 0.0)) : 0.0;
  const int n = np.n_lanes;
  int bits = 0, best = 0;
  double bd = __builtin_inf();
  // straight lanes, four table rows per trip: the rows are fetched from LDS together (wave-uniform addresses:
  // broadcasts), so the walk pays one LDS round trip per four lanes instead of one per lane.  HWY_MAX_LANES is a
  // multiple of 4; rows past n_lanes are read and masked out.
  static_assert(HWY_MAX_LANES % 4 == 0, "the lane table is walked in groups of 4");
  for (int L0 = 0; L0 < n; L0 += 4) {
    NetLaneRow row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) row[k] = sh.row[L0 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int L = L0 + k;
      const NetLaneRow &cur = row[k];
      const bool valid = L < n && !((sine_mask >> L) & 1);  // wave-uniform
      const double s = x - cur.x0, r = y - cur.y0;
      const bool on = valid && fabs(r) <= cur.hw1 && -5.0 <= s && s < cur.len5;
      bits |= on ? (1 << L) : 0;
      if (CLOSEST) {
        const double d = fabs(r) + fmax(s - cur.length, 0.0) + fmax(0 - s, 0.0) + 1.0 * angle0;
        const bool take = valid && d < bd;  // strict: the first minimum in table order
        bd = take ? d : bd;
        best = take ? L : best;
      }
    }
  }
  // SineLanes (lane.py:236-283): one sincos (+ one atan for the heading term) each
  for (unsigned m = sine_mask; m; m &= m - 1) {  // wave-uniform
    const int L = __builtin_ctz(m);
    const hwy_lane &l = np.lane[L];
    const double s = x - l.x0;
    {
      // |lateral| >= |y - y0| - |amplitude| and the distance is at least |lateral| + the longitudinal overshoot: when that
      // bound already exceeds both the membership margin and the best distance so far, the lane can neither hold the
      // body nor be its closest lane -- for most of an episode no body of the wave is anywhere near the ramp
      const double lat_lb = fmax(fabs(y - l.y0) - fabs(l.amplitude), 0.0);
      const double lb = lat_lb + fmax(s - l.length, 0.0) + fmax(0 - s, 0.0);
      const bool need = (!(lat_lb > l.width / 2 + 1.0 + 1e-9) && -5.0 <= s && s < l.length + 5.0) || (CLOSEST && !(lb > bd + 1e-9));
      if (__ballot(present && need) == 0) continue;  // (an empty slot's result is discarded by the caller)
    }
    double sn, cs;
    sincos_bounded(l.pulsation * s + l.phase, &sn, &cs);
    const double r = (y - l.y0) - l.amplitude * sn;
    const bool on = fabs(r) <= l.width / 2 + 1.0 && -5.0 <= s && s < l.length + 5.0;
    bits |= on ? (1 << L) : 0;
    if (CLOSEST) {
      const double angle = fabs(wrap_to_pi(h - (0.0 + atan_small(l.amplitude * l.pulsation * cs))));
      const double d = fabs(r) + fmax(s - l.length, 0.0) + fmax(0 - s, 0.0) + 1.0 * angle;
      const bool take = d < bd || (d == bd && L < best);
      bd = take ? d : bd;
      best = take ? L : best;
    }
  }
  *bits_out = bits;
  *closest_out = best;
}
// lanes of the table with a lateral sine offset (bit L), once per kernel
__device__ inline unsigned net_sine_mask(const NetParams &np) {
  unsigned m = 0;
  for (int L = 0; L < np.n_lanes; ++L) m |= (np.lane[L].amplitude != 0.0) ? (1u << L) : 0u;
  return m;
}

// Road.neighbour_vehicles literal scan on lane Lq (equal-x case only); returns slot indices
__device__ inline void net_neighbours_scan(const NetShared &sh, u64 pm, int bits, double x, int self, int Lq, int *front,
                                           int *rear) {
  int f = -1, b = -1;
  double s_front = 0, s_rear = 0;
  const double x0 = sh.lx0[Lq];
  const double s = x - x0;
  for (u64 m = pm; m; m &= m - 1) {  // wave-uniform
    const int j = ctz64(m);
    const double s_v = wave_bcast(x, j) - x0;
    const int bj = wave_bcast_i(bits, j);
    if (j == self || !(bj & sh.lconn[Lq])) continue;
    if (s <= s_v && (f < 0 || s_v <= s_front)) { s_front = s_v; f = j; }
    if (s_v < s && (b < 0 || s_v > s_rear)) { s_rear = s_v; b = j; }
  }
  *front = f;
  *rear = b;
}

// ---- OccupancyGridObservation (observation.py:354-413) on a road network, for one observer: like observe_grid of
//      hwy_device.h (atomic-min cell ownership: the lowest slot wins like the reference's reverse iteration, owners write
//      their features), but only Road.vehicles are rasterised (not the Obstacle of Road.objects, :366-368) and the on-road
//      layer walks the waypoints of EVERY lane of the table -- lane.position(wp, 0) = (x0 + wp, y0 [+ amplitude * sin(
//      pulsation * wp + phase)]), SineLane included (:454-484). ---------------------------------------------------------
__device__ inline void net_observe_grid(const NetParams &np, const NetShared &sh, int e, int eo, int a, const Veh &me, bool veh,
                                        double ex, double ey, double ev, double ec, double es) {
  const StepParams &p = np.s;
  const int i = threadIdx.x, NT = 64;
  const int W = p.gW, H = p.gH, WH = W * H, F = p.F;
  int32_t *own = p.grid_ws + ((size_t)e * p.A + a) * 2 * (size_t)WH, *road = own + WH;
  float *out = p.obs + ((size_t)eo * p.A + a) * (size_t)F * WH;  // eo: output row (hwy_wave.h: observe_wave)
  for (int t = i; t < WH; t += NT) {
    grid_ws_store(own + t, 0x7fffffff);
    grid_ws_store(road + t, 0);
  }
  __syncthreads();
  int my_ci = -1, my_cj = -1;
  if (veh) {
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
  bool has_road = false;
  for (int f = 0; f < F; ++f) has_road |= (p.feat[f] == HWY_FEAT_ON_ROAD);
  if (has_road) {
    for (int t = i; t < np.n_lanes * p.g_nwp; t += NT) {
      const int k = t / p.g_nwp, j = t - k * p.g_nwp;
      const double origin = ex - sh.lx0[k];  // lane.local_coordinates(observer)[0]: every lane of these networks runs along +x
      const double wp = clipd((origin - 100.0) + j * p.g_spacing, 0.0, sh.llen[k]);
      double py = sh.ly0[k];
      if (sh.lamp[k] != 0.0) {
        double sn, cs;
        sincos_bounded(sh.lpuls[k] * wp + sh.lphase[k], &sn, &cs);
        py = sh.ly0[k] + (0.0 + sh.lamp[k] * sn);
      }
      int ci, cj;
      grid_cell(p, (sh.lx0[k] + wp) - ex, py - ey, ec, es, &ci, &cj);
      if (0 <= ci && ci < W && 0 <= cj && cj < H) grid_ws_store(road + ci * H + cj, 1);
    }
  }
  __syncthreads();
  const bool clip = (p.flags & HWY_C_OBS_CLIP) != 0;
  if (my_ci >= 0 && grid_ws_load(own + my_ci * H + my_cj) == i) {
    for (int f = 0; f < F; ++f) {
      const int fid = p.feat[f];
      if (fid == HWY_FEAT_ON_ROAD) continue;
      double val = EnvBlock<1>::feature(p, fid, me.x, me.y, me.h, me.v, me.ch, me.sh, me.lane);
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
  for (int t = i; t < F * WH; t += NT) {
    const int f = t / WH, c = t - f * WH;
    if (p.feat[f] == HWY_FEAT_ON_ROAD) out[t] = grid_ws_load(road + c) ? ((p.flags & HWY_C_GRID_IMAGE) ? 255.0f : 1.0f) : 0.0f;
    else if (grid_ws_load(own + c) == 0x7fffffff) out[t] = 0.0f;
  }
  __syncthreads();
}

// ---- KinematicObservation (observation.py:234-276) with obstacles (road.py:421-450), MergeEnv reward and
//      termination (merge_env.py:40-82).  All cross-lane reads through readlane.  GRID: the build for the OccupancyGrid
//      observation -- a compile-time switch, so that the Kinematics kernels (the measured configs) carry none of the grid
//      code in their register / scalar allocation. ---------------------------------------------------------------------
template <bool GRID>
__device__ inline void net_observe(const NetParams &np, const NetShared &sh, int e, int eo, const Veh &me, bool write_reward,
                                   int rank = -1) {
  const StepParams &p = np.s;
  const int i = threadIdx.x;
  const bool present = i < p.N && !(me.flags & HWY_F_ABSENT);
  const bool obstacle = present && (me.flags & HWY_F_OBSTACLE);
  const bool veh = present && !obstacle;
  const int V = p.V, F = p.F;
  // altruistic penalty: sum over Road.vehicles on merge_lane, in list order (Python sum() from 0)
  double merging = 0.0;
  if (write_reward) {
    const double term = veh ? (me.ts - me.v) / me.ts : 0.0;
    for (u64 m = __ballot(veh && me.lane == np.merge_lane); m; m &= m - 1) merging = merging + wave_bcast(term, ctz64(m));
  }
  for (int a = 0; a < p.A; ++a) {
    const int ia = p.agent_index[a];
    const double ex = wave_bcast(me.x, ia), ey = wave_bcast(me.y, ia), ev = wave_bcast(me.v, ia);
    const double ec = wave_bcast(me.ch, ia), es = wave_bcast(me.sh, ia);
    const double ox = sh.lx0[wave_bcast_i(me.lane, ia)];
    const double dxe = me.x - ex, dye = me.y - ey;
    const double d_lane = (me.x - ox) - (ex - ox);  // observer.lane_distance_to(me)
    const bool near = dxe * dxe + dye * dye < p.perception * p.perception;
    const bool elig = present && near && (obstacle ? (!(p.flags & HWY_C_OBS_VEHICLES_ONLY) && -2 * HWY_VEH_LENGTH < d_lane)
                                                   : (i != ia && ((p.flags & HWY_C_OBS_SEE_BEHIND) || (-2 * HWY_VEH_LENGTH < d_lane))));
    const double key = elig ? ((p.flags & HWY_C_OBS_UNSORTED) ? 0.0 : fabs(d_lane)) : __builtin_inf();  // (sort=False: list order)
    const int n_elig = __popcll(__ballot(elig));
    const int m = n_elig < V - 1 ? n_elig : V - 1;
    int pos = 0;  // stable sort position; obstacles sit after every vehicle slot, so slot order == list order
    bool by_rank = false;
    if (!GRID && rank >= 0 && !(p.flags & (HWY_C_OBS_SEE_BEHIND | HWY_C_OBS_UNSORTED))) {  // wave-uniform
      // `rank` = the exact rank along x on the CURRENT positions.  Everything eligible BEHIND the observer is closer than 2 LENGTH,
      // so the eligible split into `near` (key < 2 LENGTH: a handful, ordered by explicit compares) and `far` (all in front,
      // after every near one, and among themselves ordered like x, i.e. like their rank) -- hwy_wave.h's observe_wave.  "Like x"
      // needs the key strictly monotone in x: d_lane = (x - ox) - (ex - ox) is, as long as both subtractions are EXACT for the
      // far ones (TwoSum error terms, checked: else two cars a rounding apart could tie, and the reference breaks a tie by
      // list order); one inexact far key sends this observer to the loop over all eligible below.
      const bool nearv = elig && key < 2 * HWY_VEH_LENGTH, farv = elig && !nearv;
      const double d1 = me.x - ox, c1 = ex - ox;
      const double bb1 = d1 - me.x, err1 = (me.x - (d1 - bb1)) + (-ox - bb1);
      const double bb2 = d_lane - d1, err2 = (d1 - (d_lane - bb2)) + (-c1 - bb2);
      if (__ballot(farv && !(err1 == 0.0 && err2 == 0.0)) == 0) {
        by_rank = true;
        const u64 near_m = __ballot(nearv);
        const u64 far_r = __ballot(wave_send_i(farv ? 1 : 0, rank) != 0);  // rank space
        for (u64 em = near_m; em; em &= em - 1) {  // wave-uniform
          const int k = ctz64(em);
          const double kk = wave_bcast(key, k);
          pos += ((kk < key) || (kk == key && k < i)) ? 1 : 0;
        }
        pos = farv ? __popcll(near_m) + __popcll(far_r & (((u64)1 << rank) - 1)) : pos;
      }
    }
    if (!by_rank) {
      for (u64 em = __ballot(elig); em; em &= em - 1) {
        const int k = ctz64(em);
        const double kk = wave_bcast(key, k);
        pos += ((kk < key) || (kk == key && k < i)) ? 1 : 0;
      }
    }
    if constexpr (GRID) {
      if (p.obs) net_observe_grid(np, sh, e, eo, a, me, veh, ex, ey, ev, ec, es);
    }
    if (!GRID && p.obs) {
      float *out = p.obs + ((size_t)eo * p.A + a) * (size_t)(V * F);
      const int row = (i == ia) ? 0 : (elig && pos < V - 1 ? pos + 1 : -1);
      if (present && row >= 0) {
        for (int f = 0; f < F; ++f) {
          const int fid = p.feat[f];
          double val = EnvBlock<1>::feature(p, fid, me.x, me.y, me.h, me.v, me.ch, me.sh, me.lane);
          const bool rel = fid == HWY_FEAT_X || fid == HWY_FEAT_Y || fid == HWY_FEAT_VX || fid == HWY_FEAT_VY;
          if (row > 0 && rel && !(p.flags & HWY_C_OBS_ABSOLUTE)) {
            const double origin = fid == HWY_FEAT_X ? ex : fid == HWY_FEAT_Y ? ey : fid == HWY_FEAT_VX ? ev * ec : ev * es;
            val -= origin;
          }
          if (rel && (p.flags & HWY_C_OBS_NORMALIZE)) {
            const double r0 = fid == HWY_FEAT_X ? p.rx0 : fid == HWY_FEAT_Y ? p.ry0 : fid == HWY_FEAT_VX ? p.rvx0 : p.rvy0;
            // (the host-computed reciprocal of the feature range, like hwy_wave.h: an f64 ulp away from the quotient at most,
            //  far below the rounding of the f32 observation)
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
      const int act = p.actions ? p.actions[(size_t)eo * p.A + a] : HWY_IDLE;
      const double scaled_speed = lmap(me.v, p.rs0, p.rs1, 0.0, 1.0);
      double reward = 0.0;
      reward = reward + p.collision_reward * (crashed ? 1.0 : 0.0);
      reward = reward + p.right_lane_reward * ((double)sh.lid[me.lane] / 1);
      reward = reward + p.high_speed_reward * scaled_speed;
      reward = reward + np.lane_change_reward * ((act == HWY_LANE_LEFT || act == HWY_LANE_RIGHT) ? 1.0 : 0.0);
      reward = reward + np.merging_speed_reward * merging;
      reward = lmap(reward, p.collision_reward + np.merging_speed_reward, p.high_speed_reward + p.right_lane_reward, 0.0, 1.0);
      p.reward[(size_t)eo * p.A + a] = reward;
      if (p.info_speed) p.info_speed[(size_t)eo * p.A + a] = me.v;
      if (p.info_crashed) p.info_crashed[(size_t)eo * p.A + a] = crashed ? 1 : 0;
      if (a == 0) {
        const bool term = crashed || (me.x > np.merge_end_x);  // merge_env.py:77-79 / :365-369
        const double t = p.st.time[e] + p.policy_dt;
        const bool trunc = t >= p.duration;  // duration == +inf: MergeEnv never truncates (merge_env.py:81-82)
        p.st.time[e] = t;
        p.terminated[eo] = term ? 1 : 0;
        p.truncated[eo] = trunc ? 1 : 0;
        if (p.autoreset) p.st.done[e] = (term || trunc) ? 1 : 0;
      }
    }
  }
}

__device__ inline void net_load_table(const NetParams &np, NetShared &sh) {
  const int i = threadIdx.x;
  if (i < np.n_lanes) {
    const hwy_lane &l = np.lane[i];
    sh.lx0[i] = l.x0; sh.ly0[i] = l.y0; sh.llen[i] = l.length; sh.lwid[i] = l.width; sh.lamp[i] = l.amplitude;
    sh.lpuls[i] = l.pulsation; sh.lphase[i] = l.phase; sh.llimit[i] = l.speed_limit;
    sh.row[i] = NetLaneRow{l.x0, l.y0, l.length, l.length + 5.0, l.width / 2 + 1.0, 0.0};
    sh.lroad[i] = l.road; sh.lid[i] = l.id; sh.lfirst[i] = l.road_first; sh.lcount[i] = l.road_lanes;
    sh.lnext[i] = l.next_first; sh.lnextn[i] = l.next_lanes; sh.lforb[i] = l.forbidden;
    // lanes searched together with lane i by Road.neighbour_vehicles (road.py:508-529); just the lane itself by default
    sh.lconn[i] = (np.s.flags & HWY_C_CONNECTED_LANES) ? l.connected : (1 << i);
    // the lanes whose search reads lane i's members (the inverse of lconn): a member of lane i sets its bit in all of them
    int inv = 0;
    for (int K2 = 0; K2 < np.n_lanes; ++K2)
      inv |= ((((np.s.flags & HWY_C_CONNECTED_LANES) ? np.lane[K2].connected : (1 << K2)) >> i) & 1) ? (1 << K2) : 0;
    sh.linv[i] = inv;
  }
  __syncthreads();
}

// ---- device-side spawn: MergeEnv._make_vehicles (merge_env.py:162-187) / MergeGenericEnv._make_vehicles
//      (:320-363) on Philox uniforms (NOT numpy's stream; the stream-identical reset is highwayenv_amd/merge.py).
//      Slot layout as in merge.py: [ego, traffic (A-1 of them controlled), merging vehicle, obstacle]. --------------
__device__ inline void net_spawn_env(const NetParams &np, NetShared &sh, uint64_t seed, uint32_t episode, Veh &o) {
  const StepParams &p = np.s;
  const int i = threadIdx.x;
  const int N = p.N, lanes = p.L;
  const int n_traffic = N - 3;
  const bool generic = np.generic != 0;
  const double w = 4.0;
  // traffic positions: the rejection sampling is sequential over the VEHICLES (a candidate is tested against everything placed
  // before it, merge_env.py:307-331) but not over a vehicle's 10 tries: thread 6 t + part draws try t (its own Philox block,
  // like the serial replay) and tests it against every sixth placed vehicle; the first try that no thread of its group
  // objects to is the one the serial loop would have accepted.  (One thread replaying all of it took longer than a whole
  // 15-frame policy step, and a fifth of the wavefronts of a merge launch are re-spawning: profiles/r02_history.md.)
  double *pos = sh.scratch;  // [n_traffic] longitudinal or -1 (gave up)
  int *lane_of = sh.idx;     // [n_traffic]
  if (generic) {
    const double max_pos = sh.lx0[2 * lanes] + sh.llen[2 * lanes];  // pre + converge + parallel
    const int t = i / 6, part = i - 6 * t;
    for (int k = 0; k < n_traffic; ++k) {  // wave-uniform
      bool conflict = false;
      int L = 0;
      double lon = 0.0;
      if (t < 10) {
        double u_lane, u_pos;
        philox_uniform2(seed, (uint32_t)(k + 1), episode, (uint32_t)t, &u_lane, &u_pos);
        L = (int)(u_lane * lanes);
        L = L > lanes - 1 ? lanes - 1 : L;
        lon = 0.0 + (max_pos - 0.0) * u_pos;
        conflict = part == 0 && (L == lanes - 1) && !(fabs(lon - 30.0) > 15.0);  // the ego sits at 30 m on the last lane
        for (int q = part; q < k; q += 6)
          if (pos[q] >= 0 && lane_of[q] == L && !(fabs(lon - pos[q]) > 15.0)) conflict = true;
      }
      const u64 cm = __ballot(conflict);
      int chosen = -1;
      for (int tt = 9; tt >= 0; --tt)
        if (((cm >> (6 * tt)) & 0x3f) == 0) chosen = tt;  // the FIRST acceptable try
      HWY_WAVE_LDS_FENCE();  // every test of this vehicle has read the table
      if (chosen >= 0 ? i == 6 * chosen : i == 0) {
        pos[k] = chosen >= 0 ? lon : -1.0;
        lane_of[k] = chosen >= 0 ? L : 0;
      }
      HWY_WAVE_LDS_FENCE();
    }
  }
  __syncthreads();
  o = Veh{};
  o.ch = 1.0;
  o.sh = 0.0;
  o.rank = i & 0xff;
  o.flags = HWY_F_ABSENT;
  double x = 0, y = 0, speed = 0, ts = -1;
  bool exists = false, controlled = false;
  if (i == 0) {  // ego on ("a","b",lanes-1) at s = 30, speed 30
    x = 30.0; y = (lanes - 1) * w; speed = 30.0; exists = true; controlled = true;
  } else if (i <= n_traffic) {
    const int k = i - 1;
    double u0, u1;
    philox_uniform2(seed, (uint32_t)i, episode, 100u, &u0, &u1);
    if (generic) {
      if (pos[k] >= 0) {
        x = pos[k]; y = lane_of[k] * w; speed = 30.0 + (-2.0 + (2.0 - -2.0) * u0); exists = true;
      }
    } else {  // (90, 29), (70, 31), (5, 31.5): lane = integers(2), position += uniform(-5, 5), speed += uniform(-1, 1)
      double u2, u3;
      philox_uniform2(seed, (uint32_t)i, episode, 101u, &u2, &u3);
      const double bp = k == 0 ? 90.0 : (k == 1 ? 70.0 : 5.0), bs = k == 0 ? 29.0 : (k == 1 ? 31.0 : 31.5);
      int L = (int)(u2 * 2);
      L = L > 1 ? 1 : L;
      x = bp + (-5.0 + (5.0 - -5.0) * u0); y = L * w; speed = bs + (-1.0 + (1.0 - -1.0) * u1); exists = true;
    }
    controlled = exists && i < p.A;
  } else if (i == N - 2) {  // merging vehicle on ("j","k",0)
    const int jk = np.n_lanes - 2;
    x = generic ? 30.0 + 30 : 110.0; y = sh.ly0[jk]; speed = 20.0; ts = 30.0; exists = true;
  } else if (i == N - 1) {  // Obstacle at the end of the acceleration lane ("b","c",lanes)
    const int acc = 2 * lanes;
    o.x = sh.lx0[acc] + sh.llen[acc]; o.y = sh.ly0[acc];
    o.flags = HWY_F_OBSTACLE | HWY_F_CHECK_COLLISIONS;
  }
  if (exists) {
    o.x = x; o.y = y; o.v = speed;
    if (controlled) {
      const double xs = (speed - p.target_speeds[0]) / (p.target_speeds[p.n_ts - 1] - p.target_speeds[0]);
      o.sidx = (int)clipd(rint(xs * (p.n_ts - 1)), 0.0, (double)(p.n_ts - 1));
      o.ts = p.target_speeds[o.sidx];
      o.flags = HWY_F_CONTROLLED | HWY_F_CHECK_COLLISIONS;
    } else {
      o.ts = ts > 0 ? ts : speed;
      o.timer = py_mod_pos((o.x + o.y) * HWY_PI, HWY_LC_DELAY);
      o.delta = 4.0;  // IDMVehicle.DELTA class default: no randomize_behavior in the merge scenarios
      o.flags = HWY_F_CHECK_COLLISIONS;
    }
  }
  // lane_index = get_closest_lane_index(position, heading 0) for everything that exists (objects.py:46-51)
  const int cl = net_closest_lane(sh, np.n_lanes, o.x, o.y, 0.0);
  if (!(o.flags & HWY_F_ABSENT)) o.lane = o.tgt = cl;
  __syncthreads();
}

// =============================================================================================================
// A second view `q` of the by-value kernel argument `np` through a pointer the compiler cannot see through (hwy_wave.h:
// HWY_RELOAD_PARAMS).  The kernel must have NetParams as its only argument.
#ifndef HWY_RELOAD_NET_PARAMS
#define HWY_RELOAD_NET_PARAMS(q, np)                              \
  auto kernarg_ = __builtin_amdgcn_kernarg_segment_ptr();         \
  asm volatile("" : "+s"(kernarg_));                              \
  const NetParams &q = *(const NetParams *)kernarg_
#endif

#ifndef HWY_NET_WALK_STEPS
#define HWY_NET_WALK_STEPS 2  // partners asked per trip of the collision walk
#endif
// One policy step of environment e by its wavefront (the lane table is in LDS); eo = row of the action / output planes.
template <bool GRID>
__device__ __forceinline__ void net_policy_step(const NetParams &np, NetShared &sh, const int e, const int eo) {
  const StepParams &p = np.s;
  const int i = threadIdx.x;
  const int N = p.N;

  // ---- auto-reset --------------------------------------------------------------------------------------------
  if (p.autoreset && p.st.done[e]) {
    Veh me;
    const uint32_t episode = p.st.episode[e] + 1u;
    net_spawn_env(np, sh, p.rp.base_seed + (uint64_t)e, episode, me);
    net_observe<GRID>(np, sh, e, eo, me, false);
    store_vehicle<1>(p, e, me);
    if (i < p.A) {  // agent a == slot a
      p.reward[(size_t)eo * p.A + i] = 0.0;
      if (p.info_speed) p.info_speed[(size_t)eo * p.A + i] = me.v;
      if (p.info_crashed) p.info_crashed[(size_t)eo * p.A + i] = 0;
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

  WaveTurn turn;  // the wavefronts sharing a SIMD take turns at the top issue priority (hwy_wave.h)
  wave_turn_init(turn, p.prio_shift, p.prio_recip);
  Veh me;
  load_vehicle<1>(p, e, me);
  const bool present = i < N && !(me.flags & HWY_F_ABSENT);
  const bool obstacle = present && (me.flags & HWY_F_OBSTACLE);
  const bool veh = present && !obstacle;
  const bool controlled = veh && (me.flags & HWY_F_CONTROLLED);
  const bool idm = veh && !controlled;
  const u64 pm = __ballot(present);
  const int n_present = __popcll(pm);
  int agent = 0;
  if (controlled)
    for (int a = 0; a < p.A; ++a)
      if (p.agent_index[a] == i) agent = a;
  const bool i_check = present && (me.flags & HWY_F_CHECK_COLLISIONS);
  const u64 chk = __ballot(i_check);
  // lane membership bits of the current position: from the loaded state for the first frame, afterwards from the
  // same table pass that re-indexes the lane after the integration (on_state_update) -- the position does not
  // change between the end of a frame and the start of the next one
  const unsigned sine_mask = net_sine_mask(np);
  int bits;
  {
    int unused;
    net_lane_pass<false>(np, sh, sine_mask, present, me.x, me.y, me.h, &bits, &unused);
    bits = present ? bits : 0;
  }
  int rank = 0;
  bool has_tie = false;

  for (int fr = 0; fr < p.n_frames; ++fr) {
    wave_turn(turn);
    // ---- A. meta-actions of all agents (abstract.py:294-304 -> MDPVehicle.act, controller.py:295-315;
    //         ControlledVehicle.act starts with follow_road, :98) ---------------------------------------------
    if (fr == 0 && p.actions && controlled) {
      const int act = HWY_ACTION_TO_ALL(p.action_set, p.actions[(size_t)eo * p.A + agent]);
      me.tgt = net_follow_road(sh, me.tgt, me.x, me.y);
      if (act == HWY_FASTER || act == HWY_SLOWER) {
        const double xs = (me.v - p.target_speeds[0]) / (p.target_speeds[p.n_ts - 1] - p.target_speeds[0]);
        int idx = (int)clipd(rint(xs * (p.n_ts - 1)), 0.0, (double)(p.n_ts - 1)) + (act == HWY_FASTER ? 1 : -1);
        idx = idx < 0 ? 0 : (idx > p.n_ts - 1 ? p.n_ts - 1 : idx);
        me.sidx = idx;
        me.ts = p.target_speeds[idx];
      } else if (act == HWY_LANE_LEFT || act == HWY_LANE_RIGHT) {
        int id = sh.lid[me.tgt] + (act == HWY_LANE_RIGHT ? 1 : -1);
        id = id < 0 ? 0 : (id > sh.lcount[me.tgt] - 1 ? sh.lcount[me.tgt] - 1 : id);
        const int cand = sh.lfirst[me.tgt] + id;
        if (net_reachable(sh, cand, me.x, me.y)) me.tgt = cand;
      }
    }

    // ---- B. rank along x, lane membership masks, frame-start snapshot ----------------------------------------
    // (counted in the first frame of a step, then carried from frame to frame and merely re-validated)
    {
      // the order at the end of the previous step travels in the packed word (a HINT, like on the highway: hwy_set_state writes
      // the slot index, a device spawn too): taken if it is a permutation that gives the present slots the ranks below
      // n_present -- one ds_permute and two ballots --, then verified against the positions like in every later frame;
      // anything else is counted (rounds 1-4 counted in the first frame of every step: 4.9 % of BASELINE config 5)
      bool trusted = true;  // wave-uniform
      if (fr == 0) {
        rank = present ? me.rank : n_present + __popcll(~pm & (((u64)1 << i) - 1));
        trusted = __ballot(present && rank >= n_present) == 0 && __ballot(wave_send_i(1, rank) != 0) == ~(u64)0;
      }
      net_update_rank(me.x, present, pm, n_present, rank, has_tie, trusted);
    }
    // lane_mask[i] = the ranks lane i is searched with: its own members and, with connected lanes, those of the connected
    // segments too (a vehicle on two of them is one bit; all of these lanes measure s from x, so the order along x is the
    // order of `s_v + offset` of road.py:536-545).  Every vehicle ORs its rank bit into the masks of the lanes that read the
    // lanes it is on (one or two, up to four at a segment joint: ds_or_b64) -- rounds 1-3 ran one ballot + select per LANE of
    // the table, 15 of them per frame on the merge network.
    HWY_WAVE_LDS_FENCE();  // (the previous frame's readers of the masks are done)
    if (i < np.n_lanes) sh.lane_mask[i] = 0;
    int pubs = 0;
    for (int b_ = present ? bits : 0; b_; b_ &= b_ - 1) pubs |= sh.linv[__builtin_ctz(b_)];
    HWY_WAVE_LDS_FENCE();
    for (; pubs; pubs &= pubs - 1)
      __hip_atomic_fetch_or(&sh.lane_mask[__builtin_ctz(pubs)], (u64)1 << rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const double log_ratio = veh ? net_log_ratio(me.v, me.ts, sh.llimit[me.lane]) : 0.0;
    HWY_WAVE_LDS_FENCE();
    if (present) {
      sh.x[rank] = me.x; sh.v[rank] = me.v; sh.c[rank] = me.ch; sh.s[rank] = me.sh; sh.lr[rank] = log_ratio;
      sh.ox[rank] = sh.lx0[me.lane];
      sh.idx[rank] = i;
      sh.kind[rank] = veh ? 1 : 0;
    }
    HWY_WAVE_LDS_FENCE();

    wave_turn(turn);
    // ---- C. Road.act -------------------------------------------------------------------------------------------
    const bool crashed0 = (me.flags & HWY_F_CRASHED) != 0;
    const bool drives = idm && !crashed0;  // IDMVehicle.act returns early when crashed (behavior.py:102-103)
    const int tgt_old = me.tgt;            // what vehicles later in the list read from me (abort rule)
    // follow_road for every acting vehicle (behavior.py:106, controller.py:98)
    if (drives || controlled) me.tgt = net_follow_road(sh, me.tgt, me.x, me.y);
    const int tgt_f = me.tgt;
    const bool changer = drives && me.lane != tgt_f;
    const bool same_road = sh.lroad[me.lane] == sh.lroad[tgt_f];
    const bool decide = drives && me.lane == tgt_f && (HWY_LC_DELAY < me.timer);
    if (idm) me.timer = (decide ? 0.0 : me.timer) + p.dt;  // behavior.py:248, then :147
    // side lanes of my current lane (road.py:200-211): same road, id -+ 1 == table index -+ 1
    const int my_id = sh.lid[me.lane], my_cnt = sh.lcount[me.lane];
    const bool left_ok = my_id - 1 >= 0, right_ok = my_id + 1 < my_cnt;
    const int L_left = left_ok ? me.lane - 1 : me.lane, L_right = right_ok ? me.lane + 1 : me.lane;
    const double ox_me = sh.lx0[me.lane];
    // neighbours (ranks) on own / left / right / target lane
    int fo = -1, ro = -1, fl = -1, rl = -1, frt = -1, rrt = -1, ft = -1, rt_ = -1;
    if (!has_tie) {  // wave-uniform
      mask_neighbours(sh.lane_mask[me.lane], rank, &fo, &ro);
      mask_neighbours(sh.lane_mask[L_left], rank, &fl, &rl);
      mask_neighbours(sh.lane_mask[L_right], rank, &frt, &rrt);
      mask_neighbours(sh.lane_mask[tgt_f], rank, &ft, &rt_);
    } else {
      int a, b, q[8];
      net_neighbours_scan(sh, pm, bits, me.x, i, me.lane, &a, &b); q[0] = a; q[1] = b;
      net_neighbours_scan(sh, pm, bits, me.x, i, L_left, &a, &b); q[2] = a; q[3] = b;
      net_neighbours_scan(sh, pm, bits, me.x, i, L_right, &a, &b); q[4] = a; q[5] = b;
      net_neighbours_scan(sh, pm, bits, me.x, i, tgt_f, &a, &b); q[6] = a; q[7] = b;
      int r8[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
      for (u64 m = pm; m; m &= m - 1) {  // slot index -> rank
        const int j = ctz64(m);
        const int rk = wave_bcast_i(rank, j);
        for (int k = 0; k < 8; ++k) r8[k] = (q[k] == j) ? rk : r8[k];
      }
      fo = r8[0]; ro = r8[1]; fl = r8[2]; rl = r8[3]; frt = r8[4]; rrt = r8[5]; ft = r8[6]; rt_ = r8[7];
    }
    (void)ro; (void)rt_;
    const double delta = me.delta;
    const double free_self = EnvBlock<1>::idm_free_from_log(log_ratio, delta);
    // IDM interaction term behind the leader at rank r (an Obstacle leader has speed 0, heading 0)
#define HWY_NET_GAP(r) \
  ((r) >= 0 ? net_gap_term((sh.x[(r)] - ox_me) - (me.x - ox_me), me.v, me.ch, me.sh, sh.v[(r)], sh.c[(r)], sh.s[(r)]) : 0.0)
    const double gap_own = HWY_NET_GAP(fo);
    const double self_a = free_self - gap_own;
    // MOBIL (behavior.py:265-324), both side lanes; the right one wins if both pass (side_lanes order, no break)
    const bool moving = !(fabs(me.v) < 1);
    const bool cl = decide && left_ok && moving && net_reachable(sh, L_left, me.x, me.y);
    const bool cr = decide && right_ok && moving && net_reachable(sh, L_right, me.x, me.y);
    bool ok_l = cl && !(((free_self - HWY_NET_GAP(fl)) - self_a) < HWY_LC_MIN_ACC_GAIN);
    bool ok_r = cr && !(((free_self - HWY_NET_GAP(frt)) - self_a) < HWY_LC_MIN_ACC_GAIN);
    {
      // safety: the new follower (a Vehicle: an Obstacle "follower" accelerates 0, behavior.py:168-169) must not
      // brake harder than LANE_CHANGE_MAX_BRAKING_IMPOSED; its lane distance is measured on ITS current lane
      bool pend_l = ok_l && rl >= 0, pend_r = ok_r && rrt >= 0;
      while (__ballot(pend_l || pend_r) != 0) {  // wave-uniform
        const bool pend = pend_l || pend_r;
        const bool left = pend_l;
        const int rf = pend ? (left ? rl : rrt) : 0;
        const double oxf = sh.ox[rf], lr_f = sh.lr[rf];
        const bool is_veh = sh.kind[rf] != 0;
        const double g = net_gap_term((me.x - oxf) - (sh.x[rf] - oxf), sh.v[rf], sh.c[rf], sh.s[rf], me.v, me.ch, me.sh);
        // a_f = 3 (1 - E) - g, E = exp(delta * lr_f) >= 0: g beyond 5 is unsafe whatever E is, and a follower below its target
        // speed (lr_f < 0, delta > 0) has E <= 1, so g below 2 is safe whatever E is (1e-6 margins; hwy_wave.h has the argument):
        // the exp runs only if some pending lane falls in between
        const bool sure_unsafe = is_veh && g > HWY_COMFORT_ACC_MAX + HWY_LC_MAX_BRAKING + 1e-6;
        const bool sure_safe = !is_veh || (lr_f < 0.0 && delta > 0.0 && g <= HWY_LC_MAX_BRAKING - 1e-6);
        bool safe = sure_safe;
        if (__ballot(pend && !sure_unsafe && !sure_safe) != 0) {  // wave-uniform
          const double a_f = is_veh ? EnvBlock<1>::idm_free_from_log(lr_f, delta) - g : 0.0;
          safe = !(a_f < -HWY_LC_MAX_BRAKING);
        }
        if (pend) {
          if (left) { ok_l = safe; pend_l = false; } else { ok_r = safe; pend_r = false; }
        }
      }
    }
    if (ok_l) me.tgt = L_left;
    if (ok_r) me.tgt = L_right;
    // abort rule for ongoing lane changes on the same road: ordered chain over Road.vehicles (behavior.py:229-244)
    // (The per-thread rank-space form of hwy_wave.h / hwy_wave2.h was built and measured here in round 5: 232.0 us against this
    // literal chain's 230.5 on BASELINE config 5, interleaved on one box -- a merge frame holds one or two movers, a link without
    // a rival costs three compares and a ballot, and the rank-space form pays its mask exchange in every frame with a chain.
    // profiles/r05_history.md; the literal chain stays.)
    {
      u64 cm = __ballot(changer && same_road);
      // a rival is ANOTHER vehicle on its way to another lane (old or new target): none, no link can block
      if (cm && __popcll(__ballot(veh && (me.lane != tgt_old || me.lane != me.tgt))) <= 1) cm = 0;
      while (cm) {  // wave-uniform
        const int ci = ctz64(cm);
        cm &= cm - 1;
        const int Tc = wave_bcast_i(tgt_f, ci);
        // usually nobody else heads for the changer's lane: the link then costs three compares and a ballot
        if (__ballot(veh && i != ci && me.lane != Tc && ((i < ci) ? me.tgt : tgt_old) == Tc) == 0) continue;
        const double xc = wave_bcast(me.x, ci), vc = wave_bcast(me.v, ci);
        const double cc = wave_bcast(me.ch, ci), sc = wave_bcast(me.sh, ci);
        const double oxc = wave_bcast(ox_me, ci);
        const int my_tgt_seen = (i < ci) ? me.tgt : tgt_old;
        bool blk = false;
        if (veh && i != ci && me.lane != Tc && my_tgt_seen == Tc) {
          const double d = (me.x - oxc) - (xc - oxc);
          const double d_star = EnvBlock<1>::desired_gap(vc, cc, sc, me.v, me.ch, me.sh);
          blk = (0 < d) && (d < d_star);
        }
        // `break` at the first hit does not change the outcome: the target is reset once
        if (__ballot(blk) != 0 && i == ci) me.tgt = me.lane;
      }
    }

    wave_turn(turn);
    // ---- D. low-level control: steering towards the target lane, IDM / speed control --------------------------
    const double inv_v = fast_rcp(not_zero(me.v));
    double tb;
    {
      const double s_t = me.x - sh.lx0[me.tgt];
      const double lat_t = net_lat(sh, me.tgt, s_t, me.y);
      const double head_t = net_heading_at(sh, me.tgt, s_t + me.v * (0.5 * 0.2));  // TAU_PURSUIT = 0.5 * TAU_HEADING
      tb = net_steer_tan_beta(lat_t, head_t, me.h, inv_v);
    }
    double accel = free_self - gap_own;
    if (drives && me.lane != me.tgt) {
      // leader on the target lane: tgt_f's mask for an ongoing change, the side lane's for a decision just taken
      const int f2 = (me.tgt == tgt_f) ? ft : (me.tgt == L_left ? fl : frt);
      const double a2 = free_self - HWY_NET_GAP(f2);
      accel = (a2 < accel) ? a2 : accel;
    }
#undef HWY_NET_GAP
    accel = clipd(accel, -HWY_ACC_MAX, HWY_ACC_MAX);
    accel = controlled ? HWY_KP_A * (me.ts - me.v) : accel;

    // ---- E. Road.step: integrate (kinematics.py:130-177) ------------------------------------------------------
    const double x_old = me.x;
    if (veh) {
      if (!(drives || controlled)) {  // a crashed IDM vehicle keeps its previous action, then clip_actions overrides it
        tb = 0.0;
      }
      tb = crashed0 ? 0.0 : tb;
      accel = crashed0 ? -1.0 * me.v : accel;
      accel = (me.v > HWY_MAX_SPEED) ? fmin(accel, 1.0 * (HWY_MAX_SPEED - me.v))
                                     : ((me.v < HWY_MIN_SPEED) ? fmax(accel, 1.0 * (HWY_MIN_SPEED - me.v)) : accel);
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
      sincos_bounded(me.h, &me.sh, &me.ch);
    }
    {
      int cl_new, bits_new;  // on_state_update (kinematics.py:170-177) + the next frame's membership bits
      net_lane_pass<true>(np, sh, sine_mask, present, me.x, me.y, me.h, &bits_new, &cl_new);
      if (veh) me.lane = cl_new;
      bits = present ? bits_new : 0;
    }

    wave_turn(turn);
    // ---- F. Road.step: collisions (road.py:477-481, objects.py:92-138) ----------------------------------------
    // Outward scan in rank order bounded by the frame-start distance, like hwy_wave.h; "last pair in loop
    // order wins" == the partner with the highest slot (the obstacle, being last, beats every vehicle).
    // bodies after the integration, in the frame-start RANK order (same permutation as the snapshot above)
    if (present) { sh.nx[rank] = me.x; sh.ny[rank] = me.y; sh.nv[rank] = me.v; sh.nc[rank] = me.ch; sh.ns[rank] = me.sh; }
    HWY_WAVE_LDS_FENCE();
    {
      // A wave-uniform walk in rank order (partners at rank + k), bounded by the frame-start distance: lim of the
      // sphere pre-check below + what two bodies can move towards each other in one frame (speed * dt each, + a
      // pending impact each).  Speeds stay below 36 m/s in practice; the bound falls back to 50 m/s if any body is faster.
      // Both tiers are checked on the ACTUAL values of this frame (`moved` = what the integration, pending impact
      // included, did to x; the speed afterwards for the radius term): nothing clamps speeds in the reference
      // (kinematics.py:155-168 only pulls them back) and an Obstacle hands the vehicle the whole translation, so a body
      // outside both tiers turns the walk into the literal all-pairs loop.
      const double moved = fabs(me.x - x_old);
      const bool calm = __ballot(present && !(fabs(me.v) <= 36.0 && moved <= 36.0 * p.dt)) == 0;
      const bool wide = !calm && __ballot(present && !(fabs(me.v) <= 50.0 && moved <= 50.0 * p.dt + 3.0)) != 0;
      const double vb = (calm ? 36.0 : 50.0) * p.dt;
      const double reach = wide ? __builtin_inf() : (5.5 + vb) + 2.0 * (vb + (calm ? 0.0 : 3.0));
      // FORWARD only (round 4): every unordered pair is met once, from its rear end -- half the walk of rounds 1-3, which went
      // both ways and dropped half of what they met.  A walk step only COLLECTS the partner inside the reference's pre-check
      // sphere, as a list entry (lower slot's rank | higher slot's rank << 8); the list pass then filters one PAIR per thread
      // (pair type, checkers, provable separation) and runs the SAT if any pair of the wavefront survives -- one pass for the
      // whole wave instead of one trip per candidate of the busiest thread -- and the verdicts meet per slot in LDS: crashed
      // flags, the highest partner slot with a pending impact ("last pair in loop order wins") and that pair's translation.
      // (the snapshot arrays of this frame are dead here: they hold the per-slot results and the pair list)
      int *const jmax = reinterpret_cast<int *>(sh.lr), *const hit = reinterpret_cast<int *>(sh.ox);
      double *const ipx = sh.v, *const ipy = sh.c;
      unsigned short *const plist = reinterpret_cast<unsigned short *>(sh.scratch);  // 288 entries (at most 63 + 2 x 64 are ever listed)
      jmax[i] = -1;
      hit[i] = 0;
      const u64 below = ((u64)1 << i) - 1;
      constexpr int WS = HWY_NET_WALK_STEPS;
      int n_list = 0, k = 1;  // wave-uniform
      bool go_b = present, walking = true;
      while (walking || n_list) {
        while (walking && n_list < 64) {
          // two walk steps per trip (like the other two sorted kernels); the slots are clamped by the range alone so that every
          // LDS read of the trip is issued before anything depends on one
          bool near[WS];
          int q[WS], r2[WS];
          double x0[WS], px[WS], py[WS], pv[WS];
#pragma unroll
          for (int u = 0; u < WS; ++u) {
            const int rb = rank + k + u;
            r2[u] = rb < n_present ? rb : 0;
            x0[u] = sh.x[r2[u]]; px[u] = sh.nx[r2[u]]; py[u] = sh.ny[r2[u]]; pv[u] = sh.nv[r2[u]];
            q[u] = sh.idx[r2[u]];
          }
#pragma unroll
          for (int u = 0; u < WS; ++u) {
            go_b = go_b & (rank + k + u < n_present) & !(fabs(x0[u] - x_old) > reach);
            const double dx = px[u] - me.x, dy = py[u] - me.y;
            const double lim = 5.5 + fmax(fabs(me.v), fabs(pv[u])) * p.dt;
            near[u] = go_b & !(dx * dx + dy * dy > lim * lim);
          }
          k += WS;
          if (__ballot(go_b) == 0 || k >= n_present) walking = false;
#pragma unroll
          for (int u = 0; u < WS; ++u) {
            const u64 km = __ballot(near[u]);
            if (km) {
              if (near[u]) plist[n_list + __popcll(km & below)] = (unsigned short)(i < q[u] ? (rank | (r2[u] << 8)) : (r2[u] | (rank << 8)));
              n_list += __popcll(km);
            }
          }
        }
        const int count = n_list < 64 ? n_list : 64, left = n_list - count;  // left < 128
        HWY_WAVE_LDS_FENCE();
        const int pair = i < count ? (int)plist[i] : -1;
        const int carry = i < left ? (int)plist[count + i] : 0, carry1 = 64 + i < left ? (int)plist[count + 64 + i] : 0;
        const int ra = pair < 0 ? 0 : (pair & 255), rb = pair < 0 ? 0 : (pair >> 8);
        const int a = sh.idx[ra], b = sh.idx[rb];  // a < b: the reference's `self` and `other`
        const bool a_veh = sh.kind[ra] != 0, b_veh = sh.kind[rb] != 0;
        const NetBody A{sh.nx[ra], sh.ny[ra], sh.nv[ra], sh.nc[ra], sh.ns[ra], a_veh ? HWY_VEH_LENGTH / 2 : 1.0, a_veh ? HWY_VEH_WIDTH / 2 : 1.0};
        const NetBody Bb{sh.nx[rb], sh.ny[rb], sh.nv[rb], sh.nc[rb], sh.ns[rb], b_veh ? HWY_VEH_LENGTH / 2 : 1.0, b_veh ? HWY_VEH_WIDTH / 2 : 1.0};
        // road.py:477-481: vehicle-vehicle and vehicle-object pairs only; objects.py:98: one of the two must check collisions;
        // provable separation on the lower slot's two body axes (any axis of either rectangle is one of the SAT's axes)
        const bool cnd = pair >= 0 && (a_veh || b_veh) && ((((chk >> a) | (chk >> b)) & 1) != 0) && !net_surely_apart(A, Bb, p.dt);
        int r = 0;
        double tx = 0.0, ty = 0.0;
        if (__ballot(cnd) != 0) {  // wave-uniform
          if (cnd) {
            r = net_pair_collide(A, Bb, p.dt, &tx, &ty);
            if (r & 1) hit[a] = hit[b] = 1;
            if (r & 2) {
              if (a_veh) __hip_atomic_fetch_max(&jmax[a], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (b_veh) __hip_atomic_fetch_max(&jmax[b], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
          HWY_WAVE_LDS_FENCE();
          if (r & 2) {  // objects.py:103-113: against an Obstacle the vehicle takes the whole translation
            const double sa = b_veh ? 0.5 : 1.0, sb = a_veh ? 0.5 : 1.0;
            if (a_veh && jmax[a] == b) { ipx[a] = tx * sa; ipy[a] = ty * sa; }
            if (b_veh && jmax[b] == a) { ipx[b] = -tx * sb; ipy[b] = -ty * sb; }
          }
        }
        if (i < left) plist[i] = (unsigned short)carry;
        if (64 + i < left) plist[64 + i] = (unsigned short)carry1;
        n_list = left;
        HWY_WAVE_LDS_FENCE();
      }
      HWY_WAVE_LDS_FENCE();
      if (veh && jmax[i] >= 0) {
        me.impx = ipx[i];
        me.impy = ipy[i];
        me.flags |= HWY_F_HAS_IMPACT;
      }
      if (present && hit[i]) me.flags |= HWY_F_CRASHED;
    }
  }  // frames

  // ---- G. observe / reward / done ------------------------------------------------------------------------------
  if (p.full_step) {
    // the positions moved in the last frame: the rank is re-validated once more, for the observation's sort (ties in x: the
    // loop over all eligible)
    if (!GRID && p.n_frames > 0) net_update_rank(me.x, present, pm, n_present, rank, has_tie);
    net_observe<GRID>(np, sh, e, eo, me, true, (!GRID && p.n_frames > 0 && !has_tie) ? rank : -1);
  }
  {
    // the hint the next step verifies; a call without frames (hwy_step_frames(0), hwy_observe) never formed one and keeps the old one
    if (p.n_frames > 0) me.rank = rank & 0xff;
    store_vehicle<1>(p, e, me, false);
  }
}

template <int WPE, bool GRID = false>
__global__ void __launch_bounds__(64, WPE) hwy_net_step_kernel(const NetParams np) {
  __shared__ NetShared sh;
  net_load_table(np, sh);
  net_policy_step<GRID>(np, sh, blockIdx.x, blockIdx.x);
}

// hwy_rollout_device on the road-network kernel: np.s.k_steps policy steps per wavefront in one launch (hwy_wave.h:
// hwy_rollout_wave_kernel has the argument); the lane table is loaded into LDS once.
template <int WPE, bool GRID = false>
__global__ void __launch_bounds__(64, WPE) hwy_net_rollout_kernel(const NetParams np) {
  __shared__ NetShared sh;
  net_load_table(np, sh);
  const int e = blockIdx.x;
  for (int k = 0; k < np.s.k_steps; ++k) {  // wave-uniform
    HWY_RELOAD_NET_PARAMS(nk, np);  // a fresh, opaque view of the arguments per step: nothing stays live -- spilled -- across steps
    net_policy_step<GRID>(nk, sh, e, k * nk.s.num_envs + e);
    HWY_WAVE_LDS_FENCE();
    __threadfence_block();
  }
}

// Reset kernel: AbstractEnv.reset for the masked environments + first observation.
template <int WPE, bool GRID = false>
__global__ void __launch_bounds__(64, WPE) hwy_net_reset_kernel(const NetParams np) {
  const StepParams &p = np.s;
  __shared__ NetShared sh;
  const int e = blockIdx.x, i = threadIdx.x;
  net_load_table(np, sh);
  if (p.reset_mask && !p.reset_mask[e]) return;  // block-uniform
  Veh me;
  const uint64_t seed = p.reset_seeds ? p.reset_seeds[e] : p.rp.base_seed + (uint64_t)e;
  net_spawn_env(np, sh, seed, 0u, me);
  net_observe<GRID>(np, sh, e, e, me, false);
  store_vehicle<1>(p, e, me);
  if (i == 0) {
    p.st.time[e] = 0.0;
    p.st.done[e] = 0;
    p.st.episode[e] = 0;
  }
}

// Observation-only kernel (hwy_observe).
template <int WPE, bool GRID = false>
__global__ void __launch_bounds__(64, WPE) hwy_net_observe_kernel(const NetParams np) {
  __shared__ NetShared sh;
  net_load_table(np, sh);
  Veh me;
  load_vehicle<1>(np.s, blockIdx.x, me);
  net_observe<GRID>(np, sh, blockIdx.x, blockIdx.x, me, false);
}

}  // namespace hwy

// hwy_ix.h -- the fused policy-step kernel for the INTERSECTION scenario (IntersectionEnv,
// highway_env/envs/intersection_env.py): ONE 64-wide wavefront per environment,
// thread i == slot i of the vehicle arrays (Road.vehicles order; HWY_F_ABSENT slots after the last vehicle, the list
// is re-compacted whenever vehicles are cleared); with at most 32 slots the upper half of the wavefront works as
// helper lanes in the item-parallel loops of a frame (IxSharedT).
//
// What differs from the x-aligned formulations (hwy_wave.h, hwy_net.h): the lanes of this network point in every
// direction and a third of them are circular arcs, so there is no single sort key along "the road".  Instead every
// vehicle projects itself on EVERY lane once per frame (one walk over the lane table after the integration: the
// same projections give the new lane index -- get_closest_lane_index -- the membership bits of the next frame and
// the longitudinal coordinate s on each lane, parked in LDS), every vehicle ORs its slot bit into the masks of the
// lanes it is on, and a front / rear query on lane L is a walk over the few set bits of mask[L] reading s[L][j] from LDS.
//   * planned routes (controller.py:71-87; next_lane, road.py:73-133): every road has one lane, a route is a list of
//     lane-table indices in one packed word;
//   * RegulatedRoad (road/regulation.py): every int(1 / dt / 2) frames every vehicle predicts 11 constant-speed poses
//     along its route ONCE (LDS), then each thread tests itself against every other slot (sphere check, rotated
//     rectangles with the reference's 9 sample points) and decides from lane priorities whether IT yields -- the
//     reference's pair loop is order-independent (a vehicle yields iff some pair names it);
//   * Road.act is order-independent here too (one-lane roads: no lane-change abort chain), collisions keep the
//     "last pair in loop order wins" rule through "highest partner slot wins";
//   * IntersectionEnv.step clears leaving vehicles and spawns at most one per policy step (intersection_env.py:
//     136-140, 292-338): stable compaction by ballot + ds_permute, Philox draws (not numpy's stream).
#pragma once

#include "hwy_device.h"
#include "hwy_wave.h"
#include "hwy_net.h"

namespace hwy {

struct IxParams {
  StepParams s;  // must stay first
  int32_t n_lanes, initial_count, host_spawn, destination;  // host_spawn: the host clears / spawns (reference stream)
  int32_t access_lane[4], exit_of[4];
  double spawn_probability, arrived_reward, d0, tau, a_max, b_min;
  const hwy_glane *lanes;  // device memory [n_lanes]
  long long *route;        // [E][pitch] planned routes (route word, below)
  long long route_table[HWY_MAX_GLANES][4];  // plan_route_to("o" + k) from the END of lane L: hwy_config.gnet_routes
  int32_t *road_steps;     // [E]
  // next-episode pre-warming (auto-reset): a second copy of the vehicle planes and, per environment,
  // {episode the shadow belongs to, warm-up progress, RegulatedRoad.steps, unused}; nullptr = off
  int32_t num_envs;
  int32_t prewarm_frames;  // warm-up frames a pre-warming workgroup advances per launch
  int32_t helpers;         // N <= 32: launch 64 threads per environment, lanes 32..63 help (see IxSharedT); 0 = 32 threads
  DevState shadow;
  long long *shadow_route;
  int32_t *shadow_meta;    // [E][4]
  unsigned long long *counters;  // [HWY_CTR_COUNT] event counters of the engine (hwy_get_counters), nullptr = not counted
};

// packed per-vehicle word of this scenario: lane[0:4] | target_lane[5:9] | speed_index[10:12] | flags[13:19]
__host__ __device__ inline int32_t ix_pack_word(int lane, int tgt, int sidx, int flags) {
  return (lane & 0x1f) | ((tgt & 0x1f) << 5) | ((sidx & 0x7) << 10) | ((flags & 0x7f) << 13);
}
__host__ __device__ inline int ix_word_lane(int32_t w) { return w & 0x1f; }
__host__ __device__ inline int ix_word_target(int32_t w) { return (w >> 5) & 0x1f; }
__host__ __device__ inline int ix_word_speed_index(int32_t w) { return (w >> 10) & 0x7; }
__host__ __device__ inline int ix_word_flags(int32_t w) { return (w >> 13) & 0x7f; }
// route word (64 bits): the remaining roads of ControlledVehicle.route as gnet indices, 5 bits each, first road in the low
// bits (up to HWY_MAX_ROUTE = 11 of them), their number in bits 56..59
typedef long long route_t;
__host__ __device__ inline int route_len(route_t r) { return (int)((r >> 56) & 0xf); }
__host__ __device__ inline int route_at(route_t r, int k) { return (int)((r >> (5 * k)) & 0x1f); }
__host__ __device__ inline route_t route_pop(route_t r) {
  return ((r & (((route_t)1 << 55) - 1)) >> 5) | ((route_t)(route_len(r) - 1) << 56);
}
// [lane] + tail, where `tail` is a route word starting at the road AFTER `lane` (a row of IxParams::route_table)
__host__ __device__ inline route_t route_prepend(int lane, route_t tail) {
  return (route_t)(lane & 0x1f) | ((tail & (((route_t)1 << 50) - 1)) << 5) | ((route_t)(route_len(tail) + 1) << 56);
}

struct IxVeh {
  double x, y, h, v, timer, ts, delta, impx, impy;
  double ch, sh;  // cos / sin of the heading, refreshed whenever the heading changes
  int lane, tgt, sidx, flags;
  route_t route;
  // HBM write-back bookkeeping (ix_store_vehicle): the route word as loaded, and whether this SLOT now holds another
  // vehicle than the one it was loaded with (compaction, spawn, a state taken from the other set of planes)
  route_t route0;
  int dirty;
};

#define HWY_IX_SAMPLES 11  // np.arange(0.25, 3, 0.25) (regulation.py:95)
#define HWY_IX_MAX_CONN 8  // connected lanes per lane (4-way junction: 3 + 1)

// CAP = slots per environment the per-vehicle LDS tables are sized for; NT = threads per workgroup (one wavefront).
//   NT == CAP (32 or 64): thread i == slot i (a 32-thread workgroup still occupies a wavefront, upper half masked off);
//   NT == 2 * CAP (CAP 32): HELPER LANES.  Threads 32..63 are empty slots everywhere except in the three sections of
//   a frame that are loops over independent items -- the walk over the lane table, the partner loop of the collision
//   check, the samples and the partner loop of the regulation -- where thread t works for vehicle t & 31 on the items
//   of parity t >> 5; partial results meet through LDS (ix_xchg) with the same tie rules the serial loops have
//   (closest lane: minimum distance then lowest table index; impact: highest partner slot), so the results are the
//   serial ones bit for bit.
// One row of the lane table as the per-frame table walk reads it, in WALK order (straight lanes first, then arcs): four
// ds_read_b128 at one address instead of an index read followed by eight scattered reads.
//   straight: a, b = start; c, d = direction; g = heading            arc: a, b = centre; c = radius; d = +-1; g = start phase
//   both: e = width / 2 + 1 (on_lane margin 1, lane.py:80-102), f = length, L = index in the lane table
struct alignas(16) IxRow {
  double a, b, c, d, e, f, g;
  int L, pad;
};

template <int CAP, int NT = CAP>
struct IxSharedT {
  static constexpr int kCap = CAP, kNH = NT / CAP;
  // lane table (struct of arrays: per-thread lane indices read it with one ds_read each)
  int kind[HWY_MAX_GLANES], ldir[HWY_MAX_GLANES], prio[HWY_MAX_GLANES], from[HWY_MAX_GLANES], to[HWY_MAX_GLANES],
      exitl[HWY_MAX_GLANES];
  int ord[HWY_MAX_GLANES], n_straight;  // table indices: straight lanes ascending, then circular lanes ascending
  // neighbour_vehicles_connected_lanes (road.py:508-529): the lanes searched after lane L itself -- those leaving its end
  // node, then those arriving at its start node, in table order (every road of this network has one lane)
  signed char conn[HWY_MAX_GLANES][HWY_IX_MAX_CONN];
  int n_next[HWY_MAX_GLANES], n_conn[HWY_MAX_GLANES];
  double sx[HWY_MAX_GLANES], sy[HWY_MAX_GLANES], lhead[HWY_MAX_GLANES], dirx[HWY_MAX_GLANES], diry[HWY_MAX_GLANES],
      cx[HWY_MAX_GLANES], cy[HWY_MAX_GLANES], rad[HWY_MAX_GLANES], sph[HWY_MAX_GLANES], len[HWY_MAX_GLANES],
      wid[HWY_MAX_GLANES], lim[HWY_MAX_GLANES];
  // 1 / radius of a CircularLane, rounded once per launch: CircularLane.position / heading_at (lane.py:341-350) divide the
  // longitudinal coordinate by the radius -- an IEEE f64 division is ~30 dependent instructions on this hardware, and the steering
  // of every frame and the 11 trajectory samples of the regulation each run one; the product with the rounded reciprocal is within
  // 1 ulp of the quotient (1e-16 of an angle: the per-frame parity is held at 1e-9)
  double irad[HWY_MAX_GLANES];
  IxRow row[HWY_MAX_GLANES];  // walk order
  u64 mask[HWY_MAX_GLANES];  // slot-space membership (on_lane, margin 1) of every lane
  // frame snapshot by slot (indexed by thread where every thread writes)
  double x[NT], y[NT], v[NT], c[NT], s[NT];
  double bcx[NT], bcy[NT], brho[NT];  // regulation: a circle around the 11 predicted positions of slot i
  // what the helper threads of a slot (IxMap) need of its vehicle in the regulation: published by the slot's own thread
  double xd[CAP];
  int xb[CAP];
  double hd[CAP];
  int vw[CAP];
  // pair work of a frame (collision partners, regulation conflicts): candidate pairs (lower slot | higher slot << 8) are
  // collected in a list and evaluated one pair per thread (ix_for_pairs); per-slot verdicts meet in jmax / flag / vlane
  unsigned short plist[128];
  int jmax[CAP], flag[CAP], vlane[CAP];
  union {
    double sl[HWY_MAX_GLANES][CAP];          // longitudinal coordinate of slot i on lane L (act phase)
    double traj[HWY_IX_SAMPLES][3][CAP];     // predicted (x, y, heading) of slot i at sample k (regulation)
  };
};

// ---- lane geometry from the LDS table, per-thread lane index ---------------------------------------------------
// StraightLane.local_coordinates (lane.py:209-213); CircularLane.local_coordinates (lane.py:355-362)
template <typename SH>
__device__ inline void ix_local(const SH &sh, int L, double x, double y, double *s, double *lat) {
  if (sh.kind[L] == 0) {
    const double dx = x - sh.sx[L], dy = y - sh.sy[L];
    *s = dx * sh.dirx[L] + dy * sh.diry[L];
    *lat = dx * -sh.diry[L] + dy * sh.dirx[L];
  } else {
    const double dx = x - sh.cx[L], dy = y - sh.cy[L];
    double phi = atan2_bounded(dy, dx);
    phi = sh.sph[L] + wrap_to_pi(phi - sh.sph[L]);
    const double r = sqrt(dx * dx + dy * dy);
    *s = sh.ldir[L] * (phi - sh.sph[L]) * sh.rad[L];
    *lat = sh.ldir[L] * (sh.rad[L] - r);
  }
}
// heading_at (lane.py:203-204, 347-350)
template <typename SH>
__device__ inline double ix_heading_at(const SH &sh, int L, double s) {
  if (sh.kind[L] == 0) return sh.lhead[L];
  const double phi = (sh.ldir[L] * s) * sh.irad[L] + sh.sph[L];
  return phi + HWY_PI / 2 * sh.ldir[L];
}
// position(s, 0) (lane.py:196-201, 341-345)
template <typename SH>
__device__ inline void ix_position(const SH &sh, int L, double s, double *px, double *py) {
  if (sh.kind[L] == 0) {
    *px = sh.sx[L] + s * sh.dirx[L] + 0.0 * -sh.diry[L];
    *py = sh.sy[L] + s * sh.diry[L] + 0.0 * sh.dirx[L];
  } else {
    const double phi = (sh.ldir[L] * s) * sh.irad[L] + sh.sph[L];
    double sn, cs;
    sincos_bounded(phi, &sn, &cs);
    *px = sh.cx[L] + (sh.rad[L] - 0.0 * sh.ldir[L]) * cs;
    *py = sh.cy[L] + (sh.rad[L] - 0.0 * sh.ldir[L]) * sn;
  }
}

template <typename SH>
__device__ inline void ix_load_table(const IxParams &ip, SH &sh) {
  const int i = threadIdx.x;
  if (i < ip.n_lanes) {
    const hwy_glane &l = ip.lanes[i];
    sh.kind[i] = l.kind; sh.ldir[i] = l.direction; sh.prio[i] = l.priority; sh.from[i] = l.from_node;
    sh.to[i] = l.to_node; sh.exitl[i] = l.exit_lane;
    sh.sx[i] = l.sx; sh.sy[i] = l.sy; sh.lhead[i] = l.heading; sh.dirx[i] = l.dirx; sh.diry[i] = l.diry;
    sh.cx[i] = l.cx; sh.cy[i] = l.cy; sh.rad[i] = l.radius; sh.sph[i] = l.start_phase; sh.len[i] = l.length;
    sh.wid[i] = l.width; sh.lim[i] = l.speed_limit;
    sh.irad[i] = l.kind != 0 ? 1.0 / l.radius : 0.0;
  }
  if (i < ip.n_lanes) {
    int n = 0;
    if (ip.s.flags & HWY_C_CONNECTED_LANES) {
      const int to = ip.lanes[i].to_node, from = ip.lanes[i].from_node;
      for (int K = 0; K < ip.n_lanes && n < HWY_IX_MAX_CONN; ++K)
        if (ip.lanes[K].from_node == to) sh.conn[i][n++] = (signed char)K;
      sh.n_next[i] = n;
      for (int K = 0; K < ip.n_lanes && n < HWY_IX_MAX_CONN; ++K)
        if (ip.lanes[K].to_node == from) sh.conn[i][n++] = (signed char)K;
    } else {
      sh.n_next[i] = 0;
    }
    sh.n_conn[i] = n;
  }
  if (i == 0) {
    int n = 0;
    for (int L = 0; L < ip.n_lanes; ++L)
      if (ip.lanes[L].kind == 0) sh.ord[n++] = L;
    sh.n_straight = n;
    for (int L = 0; L < ip.n_lanes; ++L)
      if (ip.lanes[L].kind != 0) sh.ord[n++] = L;
  }
  __syncthreads();
  if (i < ip.n_lanes) {
    const int L = sh.ord[i];
    const hwy_glane &l = ip.lanes[L];
    const bool st = l.kind == 0;
    sh.row[i] = IxRow{st ? l.sx : l.cx, st ? l.sy : l.cy, st ? l.dirx : l.radius, st ? l.diry : (double)l.direction,
                      l.width / 2 + 1.0, l.length, st ? l.heading : l.start_phase, L, 0};
  }
  __syncthreads();
}

// ---- helper GROUPS, sized per frame ---------------------------------------------------------------------------------------
// The item-parallel sections of a frame (the walk over the lane table, the trajectory samples and the partner loop of the
// regulation, the partner loop of the collision check) are loops over independent items of ONE vehicle, and most slots of an
// environment are empty most of the time: BASELINE config 4 holds 8.3 vehicles on average in its 30 slots (16 at most, measured
// over 3 200 env-steps on the emulator), so rounds 1-5's fixed split -- thread t works for slot t & 31 on the items of parity
// t >> 5 -- left three quarters of the wavefront idle in exactly the loops that are 60 % of a step.  Now the split follows the
// traffic: with `top` = the highest present slot + 1 rounded up to a power of two 2^lg, thread t works for slot vi = t & (2^lg - 1)
// as member g = t >> lg of that slot's group of G = width >> lg threads, on the items g, g + G, g + 2 G ...: eight present
// vehicles walk the twelve straight lanes in two trips instead of six, ask their eight arcs in one trip instead of four and meet
// their partners in one trip instead of four.  Per-item arithmetic, tie rules (closest lane: minimum distance then lowest table
// index; impact: highest partner slot) and therefore the results are those of the serial loops, bit for bit
// (tests/test_ix_parity.py::test_helper_groups_identity builds -DHWY_IX_STATIC_HELPERS -- the fixed split through the same code --
// next to it).  Threads at or above 2^lg own empty slots (or none): everybody but the slots' own threads (g == 0) is a helper.
struct IxMap {
  int vi, g, G, lg, top;  // my slot, my place in its group, threads per group (wave-uniform), log2(slots), highest present slot + 1
};
template <typename SH>
__device__ inline IxMap ix_map(u64 pm) {  // pm: ballot of the present slots (wave-uniform)
  const int t = threadIdx.x, W = (int)blockDim.x;
  const int top = pm ? 64 - __clzll((long long)pm) : 0;
#ifdef HWY_IX_STATIC_HELPERS
  const int lg = SH::kCap == 64 ? 6 : 5;
#else
  const int lg = top <= 1 ? 0 : 64 - __clzll((long long)(top - 1));
#endif
  return IxMap{t & ((1 << lg) - 1), t >> lg, W >> lg, lg, top};
}
// (distance, lane) minimum over the G threads of every slot's group -- minimum distance, then lowest table index: the serial
// rule is associative, so a butterfly over the group's lanes (t ^ 2^lg, t ^ 2^(lg+1) ..: ds_bpermute, no LDS storage, no
// fence) leaves the group's minimum in every one of its threads
__device__ inline void ix_group_min(const IxMap &mp, double &bd, int &best) {
  const int t = threadIdx.x, W = (int)blockDim.x;
  for (int st = 1 << mp.lg; st < W; st <<= 1) {  // wave-uniform
    const int src = (t ^ st) << 2;
    const int ohi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(bd)), olo = __builtin_amdgcn_ds_bpermute(src, __double2loint(bd));
    const int ob = __builtin_amdgcn_ds_bpermute(src, best);
    const double obd = __hiloint2double(ohi, olo);
    if (obd < bd || (obd == bd && ob < best)) { bd = obd; best = ob; }
  }
}

// ---- item work of a frame: every vehicle has the items 0 .. n - 1 (the arcs of the table, its possible partner slots); thread
//      (slot vi, member g) asks cand(j) for j = g, g + G, .. -- straight-line code: LDS reads, a dozen f64 operations, no state --
//      and the items that pass are collected in a list (vi | j << 8) and evaluated ONE PER THREAD: proc(pair, more, count) (pair < 0:
//      none; more = another pass follows, count = entries of this pass, in sh.plist[0 .. count): wave-uniform).  The serial formulation ran the expensive part once per item for the
//      whole wave whenever ANY vehicle had that item as a candidate; collected, the candidates of all slots share one pass of it
//      (64 per pass).  The list is filled in trip order, a pass runs whenever it holds a wavefront's worth (a trip adds at most
//      `width` entries to fewer than `width`: the 128 entries of plist suffice); the verdicts of proc meet per slot through LDS
//      atomics (or / min / max), so neither the order of the list nor the size of the groups can change a result.
template <typename SH, typename Cand, typename Proc>
__device__ inline void ix_for_items(SH &sh, int n, const IxMap &mp, Cand cand, Proc proc) {
  const int i = threadIdx.x, width = (int)blockDim.x;  // items per pass: 64, or 32 in the 32-thread build
  const u64 below = ((u64)1 << i) - 1;
  int n_list = 0, j0 = 0;  // wave-uniform
  while (j0 < n || n_list) {
    while (j0 < n && n_list < width) {
      const int j = j0 + mp.g;
      const bool c = cand(j < n ? j : 0) && j < n;
      j0 += mp.G;
      const u64 cm = __ballot(c);
      if (cm) {
        if (c) sh.plist[n_list + __popcll(cm & below)] = (unsigned short)(mp.vi | (j << 8));
        n_list += __popcll(cm);
      }
    }
    const int count = n_list < width ? n_list : width;
    HWY_WAVE_LDS_FENCE();
    const int pair = i < count ? (int)sh.plist[i] : -1;
    const int left = n_list - count;  // < width
    const int carry = i < left ? (int)sh.plist[count + i] : 0;
    proc(pair, j0 < n || left != 0, count);  // (more passes follow; entries of this pass: both wave-uniform)
    HWY_WAVE_LDS_FENCE();
    if (i < left) sh.plist[i] = (unsigned short)carry;
    n_list = left;
  }
}

// One walk over the lane table for my body (lane index wave-uniform): membership bits (on_lane margin 1, lane.py:80-102),
// closest lane (road.py:55-71, lane.py:132-147; minimum distance_with_heading, ties to the lowest table index) and s on
// the lanes that can matter -> sh.sl[L][i].  Straight lanes first; a CircularLane costs an atan2, and it can neither hold
// me (|lateral| > width / 2 + 1) nor be my closest lane (its distance is at least |lateral| = |radius - r|, which already
// exceeds the best distance found) nor be my target lane for most vehicles most of the time: when that is so for the
// whole wave the arc is skipped.  sl[L][i] is only read for members of L, for my own lane and for my target lane; the
// lateral coordinate on the target lane comes back too (the steering of the next frame needs exactly this projection).
template <typename SH>
__device__ inline void ix_lane_pass(const IxParams &ip, SH &sh, bool present, double x, double y, double h, int tgt,
                                    int *bits_out, int *closest_out, double *lat_tgt_out) {
  const int t = threadIdx.x;
  const IxMap mp = ix_map<SH>(__ballot(present));
  const int vi = mp.vi;
  const bool own = present;  // MY slot holds a vehicle (then vi == t); `present` below becomes slot vi's
  // the poses by slot: the threads of a slot's group work on its body, and the arc phase gathers them per pair; the per-slot
  // results of the walk meet in flag (membership bits), bcy / vlane (lateral coordinate on the target lane) and, for the arcs,
  // dmin / jmax
  unsigned long long *const dmin = reinterpret_cast<unsigned long long *>(sh.bcx);
  HWY_WAVE_LDS_FENCE();
  if (t < SH::kCap) {
    sh.x[t] = x; sh.y[t] = y; sh.hd[t] = h; sh.vw[t] = tgt | (present ? 256 : 0);
    dmin[t] = ~0ull; sh.jmax[t] = 0x7fffffff; sh.flag[t] = 0; sh.vlane[t] = 0;
  }
  HWY_WAVE_LDS_FENCE();
  if (mp.G > 1) {  // wave-uniform
    x = sh.x[vi]; y = sh.y[vi]; h = sh.hd[vi];
    const int w = sh.vw[vi];
    tgt = w & 255; present = (w & 256) != 0;
  }
  int bits = 0, best = 0x7fffffff;
  double bd = __builtin_inf(), lat_t = 0.0;  // lat_t: my lateral coordinate on my target lane (the next frame steers by it)
  bool has_lat = false;
  const int ns = sh.n_straight, n = ip.n_lanes;
  // Straight lanes: member g of a slot's group projects the body on the rows g, g + G, ..  (Rounds 4-5 put two or more rows of a
  // thread in one basic block so that their chains interleave: 259.3 / 259.3 / 259.4 us for two / three / six -- and round 6's
  // microbenchmark says why nothing moved: a DEPENDENT v_fma_f64 chain issues every 4.5 cycles like an independent one,
  // profiles/r06_issue_costs.json -- so one row per trip it is.)
  for (int k0 = 0; k0 < ns; k0 += mp.G) {  // wave-uniform trip
    const int k = k0 + mp.g;
    const bool m = k < ns;
    const IxRow r = sh.row[m ? k : k0];
    const double dx = x - r.a, dy = y - r.b;
    const double s_ = dx * r.c + dy * r.d;
    const double lat = dx * -r.d + dy * r.c;
    const bool on = fabs(lat) <= r.e && -5.0 <= s_ && s_ < r.f + 5.0;
    const double ang = fabs(wrap_to_pi(h - r.g));
    const double d = fabs(lat) + fmax(s_ - r.f, 0.0) + fmax(0 - s_, 0.0) + 1.0 * ang;
    if (m) {
      bits |= on ? (1 << r.L) : 0;
      sh.sl[r.L][vi] = s_;
      if (d < bd || (d == bd && r.L < best)) { bd = d; best = r.L; }
      if (r.L == tgt) { lat_t = lat; has_lat = true; }
    }
  }
  ix_group_min(mp, bd, best);  // the arcs are filtered against the best distance over ALL straight lanes
  // Arcs.  A CircularLane costs an atan2, and most (vehicle, arc) pairs cannot matter: the arc can only hold the vehicle
  // if |lateral| = |radius - r| <= width / 2 + 1, can only be its closest lane if |lateral| <= the best distance over the
  // straight lanes (its distance is at least |lateral|), or it is its target lane.  The pairs that pass this filter are
  // collected and projected one pair per thread (ix_for_items); their verdicts meet per vehicle: membership bits (or),
  // the closest arc (minimum of the distance's bit pattern -- distances are >= 0 --, then the lowest table index among the
  // arcs at that minimum: the serial rule) and the lateral coordinate on the target lane.
  const int na = n - ns;
  if (na > 0) {  // wave-uniform
    ix_for_items(
        sh, na, mp,
        [&](int j) {  // (straight-line code)
          const IxRow r = sh.row[ns + j];
          // |radius - rr| <= m with m = max(width / 2 + 1, best straight distance), on the squares (a filter: 1e-9 of slack)
          const double dx = x - r.a, dy = y - r.b, r2 = dx * dx + dy * dy;
          const double m = fmax(r.e, bd) + 1e-9, lo = r.c - m, hi = r.c + m;
          return present && (r.L == tgt || ((lo <= 0.0 || lo * lo <= r2) && r2 <= hi * hi));
        },
        [&](int pair, bool more, int) {
          const int v = pair < 0 ? 0 : (pair & 255);
          unsigned long long key = 0;
          int L = 0;
          if (pair >= 0) {
            const IxRow r = sh.row[ns + (pair >> 8)];
            const double px = sh.x[v], py = sh.y[v], ph = sh.hd[v];
            const double dx = px - r.a, dy = py - r.b;
            const double rr = sqrt(dx * dx + dy * dy);
            const double lat = r.d * (r.c - rr);
            double phi = atan2_bounded(dy, dx);
            phi = r.g + wrap_to_pi(phi - r.g);
            const double sa = r.d * (phi - r.g) * r.c;
            const double lane_h = ((r.d * sa) * sh.irad[r.L] + r.g) + HWY_PI / 2 * r.d;  // (heading_at: see IxSharedT::irad)
            const bool on = fabs(lat) <= r.e && -5.0 <= sa && sa < r.f + 5.0;
            const double angle = fabs(wrap_to_pi(ph - lane_h));
            const double d = fabs(lat) + fmax(sa - r.f, 0.0) + fmax(0 - sa, 0.0) + 1.0 * angle;
            L = r.L;
            sh.sl[L][v] = sa;
            if (on) __hip_atomic_fetch_or(&sh.flag[v], 1 << L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (L == (sh.vw[v] & 255)) { sh.bcy[v] = lat; sh.vlane[v] = 1; }
            key = (unsigned long long)__double_as_longlong(d);
            __hip_atomic_fetch_min(&dmin[v], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          HWY_WAVE_LDS_FENCE();
          if (pair >= 0 && dmin[v] == key)
            __hip_atomic_fetch_min(&sh.jmax[v], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (more) {  // wave-uniform, rare: more candidate pairs than one pass holds
            // every thread of a slot's group folds THIS pass's closest arc into its running minimum and the slots are cleared: the
            // next pass starts over, so "lowest index among the arcs at the minimum" never mixes two passes
            HWY_WAVE_LDS_FENCE();
            const unsigned long long akey = dmin[vi];
            if (present && akey != ~0ull) {
              const double ad = __longlong_as_double((long long)akey);
              const int aL = sh.jmax[vi];
              if (ad < bd || (ad == bd && aL < best)) { bd = ad; best = aL; }
            }
            HWY_WAVE_LDS_FENCE();
            if (t < SH::kCap) { dmin[t] = ~0ull; sh.jmax[t] = 0x7fffffff; }
          }
        });
  }
  // the straight walk's membership bits and target-lane coordinate join the arcs' per slot
  if (bits) __hip_atomic_fetch_or(&sh.flag[vi], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (has_lat) { sh.bcy[vi] = lat_t; sh.vlane[vi] = 1; }
  HWY_WAVE_LDS_FENCE();
  if (present) {
    const unsigned long long akey = dmin[vi];  // (the last pass of the arcs)
    if (akey != ~0ull) {
      const double ad = __longlong_as_double((long long)akey);
      const int aL = sh.jmax[vi];
      if (ad < bd || (ad == bd && aL < best)) { bd = ad; best = aL; }
    }
    bits = sh.flag[vi];
    if (sh.vlane[vi]) lat_t = sh.bcy[vi];
  }
  *bits_out = own ? bits : 0;  // (a helper thread holds its slot's values too: they are its vehicle's, not of the empty slot it owns)
  *closest_out = best;
  *lat_tgt_out = lat_t;
}

// Road.neighbour_vehicles (road.py:483-547): the leader on lane L among the members of mask[L] (slot order == list
// order: `<=` lets a later vehicle at the same s win, like the reference's scan).  With connected lanes (n_conn[L] > 0)
// the members of the lanes leaving L's end (s + L.length) and of those arriving at its start (s - their length) are
// candidates too; a vehicle that is on several of these lanes counts on the first one of the list.  The reference
// walks vehicles in the outer loop and the list in the inner one, so the tie rule runs over SLOT order: the walk below
// keeps, per candidate, its slot, and lets the later slot win an exact tie.
template <typename SH>
__device__ inline int ix_front(const SH &sh, int L, int self) {
  const double s = sh.sl[L][self];
  int f = -1;
  double s_front = 0.0;
  u64 seen = (u64)1 << self;
  for (u64 m = sh.mask[L] & ~seen; m; m &= m - 1) {
    const int j = ctz64(m);
    const double s_v = sh.sl[L][j];
    if (s <= s_v && (f < 0 || s_v <= s_front)) { s_front = s_v; f = j; }
  }
  const int nc = sh.n_conn[L];
  if (nc > 0) {
    seen |= sh.mask[L];
    const int nn = sh.n_next[L];
    for (int k = 0; k < nc; ++k) {
      const int K = sh.conn[L][k];
      const double off = k < nn ? sh.len[L] : -sh.len[K];
      for (u64 m = sh.mask[K] & ~seen; m; m &= m - 1) {
        const int j = ctz64(m);
        const double s_v = sh.sl[K][j] + off;
        if (s <= s_v && (f < 0 || s_v < s_front || (s_v == s_front && j > f))) { s_front = s_v; f = j; }
      }
      seen |= sh.mask[K];
    }
  }
  return f;
}

// RoadNetwork.next_lane (road.py:73-133) for one-lane roads; consumes the head of the route like route.pop(0)
template <typename SH>
__device__ inline int ix_next_lane(const IxParams &ip, const SH &sh, int cur, IxVeh &me) {
  int next = -1;
  if (route_len(me.route) > 0) {
    if (route_at(me.route, 0) == cur) me.route = route_pop(me.route);
    if (route_len(me.route) > 0 && sh.from[route_at(me.route, 0)] == sh.to[cur]) next = route_at(me.route, 0);
  }
  if (next >= 0) return next;
  // no planned successor: the lane leaving `to` that is closest to the projected position (first minimum)
  double s, lat, px, py;
  ix_local(sh, cur, me.x, me.y, &s, &lat);
  ix_position(sh, cur, s, &px, &py);
  double bd = 0.0;
  for (int K = 0; K < ip.n_lanes; ++K) {
    if (sh.from[K] != sh.to[cur]) continue;
    double s2, r2;
    ix_local(sh, K, px, py, &s2, &r2);
    const double d = fabs(r2) + fmax(s2 - sh.len[K], 0.0) + fmax(0 - s2, 0.0);
    if (next < 0 || d < bd) { bd = d; next = K; }
  }
  return next < 0 ? cur : next;  // KeyError -> current index
}

// position_heading_along_route (road.py:323-362) with lateral 0: route = v.route or [v.lane_index]
template <typename SH>
__device__ inline void ix_along_route(const SH &sh, const IxVeh &me, double lon, double *px, double *py, double *hd) {
  const int n = route_len(me.route);
  int pos = 0;
  int li = n > 0 ? route_at(me.route, 0) : me.lane;
  while (n - pos > 1 && lon > sh.len[li]) {
    lon -= sh.len[li];
    ++pos;
    li = route_at(me.route, pos);
  }
  ix_position(sh, li, lon, px, py);
  *hd = ix_heading_at(sh, li, lon);
}

// utils.has_corner_inside (utils.py:160-174): the 9 sample points of rect 1 (corners, centre, edge midpoints) against
// rect 2 with the reference's +angle rotation (utils.py:79-95)
// (c, s) = cos / sin of rect 1's angle, (c2, s2) of rect 2's: the caller tests both directions with one pair of sincos
__device__ inline bool ix_corner_inside(double c1x, double c1y, double c, double s, double c2x, double c2y, double c2, double s2) {
  const double hl = 1.5 * HWY_VEH_LENGTH / 2, hw = 0.9 * HWY_VEH_WIDTH / 2;
  bool any = false;
  for (int k = 0; k < 9; ++k) {
    const double qx = (k == 0 || k == 1 || k == 5) ? -hl : ((k == 2 || k == 3 || k == 6) ? hl : (k == 7 ? -0.0 : 0.0));
    const double qy = (k == 0 || k == 3 || k == 7) ? -hw : ((k == 1 || k == 2 || k == 8) ? hw : (k == 5 ? -0.0 : 0.0));
    const double x = (c * qx + -s * qy) + c1x, y = (s * qx + c * qy) + c1y;
    const double dx = x - c2x, dy = y - c2y;
    const double rux = c2 * dx + -s2 * dy, ruy = s2 * dx + c2 * dy;
    any = any || (-hl <= rux && rux <= hl && -hw <= ruy && ruy <= hw);
  }
  return any;
}

// Vehicle ctor pieces shared by the device spawn paths: lane index, IDM timer, planned route to "o" + dest
template <typename SH>
__device__ inline route_t ix_plan_route(const IxParams &ip, const SH &sh, int lane, int dest) {
  // plan_route_to (controller.py:71-87): [lane_index] + the shortest path from lane_index[1] to "o" + dest
  // (RoadNetwork.shortest_path: breadth-first, neighbours in sorted order, road.py:159-188) -- planned once on the host for
  // every (lane, destination) pair of the table, so any network and any route length up to HWY_MAX_ROUTE works here
  (void)sh;
  return route_prepend(lane, ip.route_table[lane][dest]);
}

__device__ inline void ix_load_vehicle(const IxParams &ip, int e, IxVeh &o, bool from_shadow = false) {
  const StepParams &p = ip.s;
  const DevState &st = from_shadow ? ip.shadow : ip.s.st;  // (wave-uniform choice)
  const route_t *route = from_shadow ? ip.shadow_route : ip.route;
  const int i = threadIdx.x;
  o = IxVeh{};
  o.flags = HWY_F_ABSENT;
  if (i < p.N) {
    const size_t k = (size_t)e * p.pitch + i;
    const int w = st.packed[k];
    o.flags = ix_word_flags(w);
    // an empty slot has nothing but its flag word: its other planes are neither read nor (ix_store_vehicle) written
    if (!(o.flags & HWY_F_ABSENT)) {
      o.x = st.x[k]; o.y = st.y[k]; o.h = st.heading[k]; o.v = st.speed[k];
      o.timer = st.timer[k]; o.ts = st.target_speed[k]; o.delta = st.delta[k];
      o.lane = ix_word_lane(w); o.tgt = ix_word_target(w); o.sidx = ix_word_speed_index(w);
      o.route = o.route0 = route[k];
      if (o.flags & HWY_F_HAS_IMPACT) {
        o.impx = st.impact_x[k];
        o.impy = st.impact_y[k];
      }
    }
  }
  sincos_bounded(o.h, &o.sh, &o.ch);
}
// Write-back of what CHANGED.  Positions, speed, heading, timer, target speed and the packed word of a present slot always
// do; the constant of a vehicle (DELTA) and its route word only when the slot holds another
// vehicle than at load time (`dirty`) or the value moved; the impact pair only where the flag says it is valid; an empty
// slot only its flag word.  `all` (a state loaded from the OTHER set of planes: the re-spawn of a pre-warmed episode)
// writes every plane of every present slot.  The fields of an empty slot are unspecified (hwy_get_state).
__device__ inline void ix_store_vehicle(const IxParams &ip, int e, const IxVeh &o, bool to_shadow = false, bool all = false) {
  const StepParams &p = ip.s;
  const DevState &st = to_shadow ? ip.shadow : ip.s.st;
  route_t *route = to_shadow ? ip.shadow_route : ip.route;
  const int i = threadIdx.x;
  if (i < p.N) {
    const size_t k = (size_t)e * p.pitch + i;
    st.packed[k] = ix_pack_word(o.lane, o.tgt, o.sidx, o.flags);
    if (!(o.flags & HWY_F_ABSENT)) {
      const bool fresh = all || o.dirty != 0;
      st.x[k] = o.x; st.y[k] = o.y; st.heading[k] = o.h; st.speed[k] = o.v;
      st.timer[k] = o.timer;
      st.target_speed[k] = o.ts;  // (RegulatedRoad moves the target speed of yielding traffic too, regulation.py:60-68)
      if (fresh) st.delta[k] = o.delta;
      if (fresh || o.route != o.route0) route[k] = o.route;
      if (o.flags & HWY_F_HAS_IMPACT) {
        st.impact_x[k] = o.impx;
        st.impact_y[k] = o.impy;
      }
    }
  }
  count_nonfinite(to_shadow ? nullptr : ip.counters, i < p.N && !(o.flags & HWY_F_ABSENT) &&
                                                       !((o.x - o.x) + (o.y - o.y) + (o.h - o.h) + (o.v - o.v) == 0.0));
}

// ---- n_frames x { [meta-action]; Road.act(); RegulatedRoad.step(dt) } on the wave's registers + LDS --------------
template <typename SH>
__device__ inline void ix_frames(const IxParams &ip, SH &sh, int e, IxVeh &me, int n_frames, const int32_t *actions,
                                 int &road_steps, int &bits, double &lat_tgt) {
  const StepParams &p = ip.s;
  const int i = threadIdx.x;
  const int every = (int)(1 / p.dt / 2);  // int(1 / dt / REGULATION_FREQUENCY) (regulation.py:38)
  WaveTurn turn;  // the wavefronts sharing a SIMD take turns at the top issue priority (hwy_wave.h)
  wave_turn_init(turn, p.prio_shift, p.prio_recip);
  for (int fr = 0; fr < n_frames; ++fr) {
    wave_turn(turn);
    const bool present = !(me.flags & HWY_F_ABSENT);
    const bool controlled = present && (me.flags & HWY_F_CONTROLLED);
    const u64 pm = __ballot(present);
    const IxMap mp = ix_map<SH>(pm);  // this frame's helper groups (nobody joins or leaves within a frame)
    // ---- A. meta-action (abstract.py:294-304 -> MDPVehicle.act, controller.py:295-315): SLOWER / IDLE / FASTER ------
    // MultiAgentAction.act (action.py:352-355): agent a == the a-th controlled vehicle of the list
    const u64 ctl_m = (fr == 0 && actions) ? __ballot(controlled) : 0;
    if (fr == 0 && actions && controlled) {
      const int agent = __popcll(ctl_m & (((u64)1 << i) - 1));
      const int act = agent < p.A ? actions[(size_t)e * p.A + agent] : 1;
      if (act == 0 || act == 2) {
        const double xs = (me.v - p.target_speeds[0]) / (p.target_speeds[p.n_ts - 1] - p.target_speeds[0]);
        int idx = (int)clipd(rint(xs * (p.n_ts - 1)), 0.0, (double)(p.n_ts - 1)) + (act == 2 ? 1 : -1);
        idx = idx < 0 ? 0 : (idx > p.n_ts - 1 ? p.n_ts - 1 : idx);
        me.sidx = idx;
        me.ts = p.target_speeds[idx];
      }
    }
    // ---- B. membership masks + snapshot ---------------------------------------------------------------------------
    HWY_WAVE_LDS_FENCE();
    // every vehicle ORs its slot bit into the masks of the lanes it is on (one to three: ds_or_b64); rounds 1-3 ran one ballot
    // per lane (pair of lanes with helper lanes) of the 20-lane table
    if (i < ip.n_lanes) sh.mask[i] = 0;
    HWY_WAVE_LDS_FENCE();
    for (int b_ = (i < SH::kCap) ? bits : 0; b_; b_ &= b_ - 1)
      __hip_atomic_fetch_or(&sh.mask[__builtin_ctz(b_)], (u64)1 << i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const double ch = me.ch, shh = me.sh;
    sh.x[i] = me.x; sh.y[i] = me.y; sh.v[i] = me.v; sh.c[i] = ch; sh.s[i] = shh;
    HWY_WAVE_LDS_FENCE();

    wave_turn(turn);
    // ---- C. Road.act (road.py:464-467) ---------------------------------------------------------------------------
    // steering -> slip angle -> bicycle model are folded like in hwy_net.h: tb = tan(beta) with beta = atan(tan(delta) / 2)
    // (controller.py:145-187 + kinematics.py:141-152; exact trigonometric identities, <= 2 ulp from the literal chain)
    double tb = 0.0, accel = 0.0;
    const bool crashed0 = (me.flags & HWY_F_CRASHED) != 0;
    const bool acts = present && (controlled || !crashed0);  // IDMVehicle.act returns early when crashed
    if (acts) {
      // follow_road (controller.py:135-143): AbstractLane.after_end on the target lane (lane.py:120-125)
      if (sh.sl[me.tgt][i] > sh.len[me.tgt] - 5.0 / 2) {
        me.tgt = ix_next_lane(ip, sh, me.tgt, me);
        // my coordinates on the new target lane (the table walk skipped it if I was not near it; only I read this slot
        // unless I am a member of that lane, in which case the walk wrote the same value)
        double s_new, lat_new;
        ix_local(sh, me.tgt, me.x, me.y, &s_new, &lat_new);
        sh.sl[me.tgt][i] = s_new;
        lat_tgt = lat_new;
      }
      if (!controlled && me.lane == me.tgt && HWY_LC_DELAY < me.timer) me.timer = 0.0;  // behavior.py:246-248
      {
        // target_lane.local_coordinates(position): the projection the table walk made after the last integration
        const double s_t = sh.sl[me.tgt][i], lat_t = lat_tgt;
        const double lane_future_heading = ix_heading_at(sh, me.tgt, s_t + me.v * (0.5 * 0.2));
        tb = net_steer_tan_beta(lat_t, lane_future_heading, me.h, fast_rcp(not_zero(me.v)));
      }
      if (controlled) {
        accel = HWY_KP_A * (me.ts - me.v);  // speed_control (controller.py:189-198)
      } else {
        // IDM (behavior.py:150-217) on the current lane, and on the target lane while they differ;
        // (v / v0) ** delta == exp(delta * log(v / v0)) with the bounded-domain routines of hwy_math.h
        const double v0 = clipd(me.ts, 0.0, sh.lim[me.lane]);
        const double ratio = fmax(me.v, 0.0) * fast_rcp(abs_not_zero(v0));
        const double powr = ratio > 0.0 ? exp_bounded(fmin(me.delta * log_pos(ratio), 40.0)) : 0.0;
        const double free_acc = ip.a_max * (1 - powr);
        const double inv_ab2 = fast_rcp(2 * sqrt(-ip.a_max * ip.b_min));
        accel = free_acc;
        for (int q = 0; q < 2; ++q) {
          const int Lq = q == 0 ? me.lane : me.tgt;
          if (q == 1 && me.lane == me.tgt) break;
          double a = free_acc;
          const int f = ix_front(sh, Lq, i);
          if (f >= 0) {
            // lane_distance_to is measured on MY current lane (objects.py:183-198)
            // (the table walk projects a vehicle on an ARC only where the arc can hold it, be its closest lane or its
            //  target: a leader found on my target lane -- or, with connected lanes, beyond my lane's end -- may have none)
            double s_f = sh.sl[me.lane][f];
            if (sh.kind[me.lane] != 0 && !((sh.mask[me.lane] >> f) & 1)) {
              double lat_f;
              ix_local(sh, me.lane, sh.x[f], sh.y[f], &s_f, &lat_f);
            }
            const double d = s_f - sh.sl[me.lane][i];
            const double dv = (me.v * ch - sh.v[f] * sh.c[f]) * ch + (me.v * shh - sh.v[f] * sh.s[f]) * shh;
            const double d_star = ip.d0 + me.v * ip.tau + (me.v * dv) * inv_ab2;
            const double r = d_star * fast_rcp(not_zero(d));
            a -= ip.a_max * (r * r);
          }
          accel = q == 0 ? a : fmin(accel, a);
        }
        accel = clipd(accel, -HWY_ACC_MAX, HWY_ACC_MAX);
      }
    }

    // ---- D. RegulatedRoad.step (regulation.py:36-68) --------------------------------------------------------------
    road_steps += 1;
    if (road_steps % every == 0) {  // wave-uniform
      const bool veh = present;
      if (veh && (me.flags & HWY_F_YIELDING)) {  // YIELD_DURATION == 0: released at the next regulation
        me.ts = sh.lim[me.lane];
        me.flags &= ~HWY_F_YIELDING;
      }
      // (helper groups, IxMap: thread t works for vehicle vi -- the samples and the partners g, g + G, .. of its group)
      const int vi = mp.vi;
      double s_me = veh ? sh.sl[me.lane][i < SH::kCap ? i : 0] : 0.0, v_me = me.v, x_me = me.x, y_me = me.y, c_me = ch, sn_me = shh;
      route_t route_me = me.route;
      int lane_me = me.lane;
      bool veh_v = veh;
      if (mp.G > 1) {  // wave-uniform
        if (i < SH::kCap) { sh.xd[i] = s_me; sh.bcx[i] = __longlong_as_double(me.route); sh.xb[i] = me.lane | (veh ? 256 : 0); }
      }
      HWY_WAVE_LDS_FENCE();  // sl[][] is dead from here on: the trajectories share its storage
      if (mp.G > 1) {
        s_me = sh.xd[vi]; route_me = __double_as_longlong(sh.bcx[vi]);  // (bcx is free until the circles below)
        lane_me = sh.xb[vi] & 255; veh_v = (sh.xb[vi] & 256) != 0;
        v_me = sh.v[vi]; x_me = sh.x[vi]; y_me = sh.y[vi]; c_me = sh.c[vi]; sn_me = sh.s[vi];  // frame snapshot (B)
      }
      if (veh_v) {
        IxVeh r{};
        r.route = route_me; r.lane = lane_me;
        for (int k = mp.g; k < HWY_IX_SAMPLES; k += mp.G) {
          double px, py, hd;
          ix_along_route(sh, r, s_me + v_me * (0.25 + k * 0.25), &px, &py, &hd);
          sh.traj[k][0][vi] = px; sh.traj[k][1][vi] = py; sh.traj[k][2][vi] = hd;
        }
      }
      HWY_WAVE_LDS_FENCE();
      // Two vehicles can only conflict if at some sample their predicted positions are within LENGTH of each other
      // (regulation.py:103): bound every vehicle's 11 positions by a circle (centre = the middle sample) and skip a
      // partner for the whole wave when no pair of circles comes within LENGTH (triangle inequality, 1e-6 of slack)
      double my_cx = 0.0, my_cy = 0.0, my_rho = 0.0;
      if (veh_v) {
        my_cx = sh.traj[HWY_IX_SAMPLES / 2][0][vi];
        my_cy = sh.traj[HWY_IX_SAMPLES / 2][1][vi];
        for (int k = 0; k < HWY_IX_SAMPLES; ++k) {
          const double dx = sh.traj[k][0][vi] - my_cx, dy = sh.traj[k][1][vi] - my_cy;
          my_rho = fmax(my_rho, dx * dx + dy * dy);
        }
        my_rho = sqrt(my_rho);  // == the maximum of the 11 distances (sqrt is monotone)
      }
      if (mp.g == 0) {
        sh.bcx[vi] = my_cx; sh.bcy[vi] = my_cy; sh.brho[vi] = my_rho;
        sh.flag[vi] = 0;
        sh.vlane[vi] = lane_me;
      }
      HWY_WAVE_LDS_FENCE();
      // is_conflict_possible (regulation.py:88-111) + respect_priorities (:70-86) of every candidate pair, one pair per
      // thread; a vehicle yields iff some pair names it (sh.flag)
      ix_for_items(
          sh, mp.top, mp,
          [&](int j) {
            const double bdx = sh.bcx[j] - my_cx, bdy = sh.bcy[j] - my_cy;
            const double reach = my_rho + sh.brho[j] + HWY_VEH_LENGTH + 1e-6;  // (a filter: squares compare as well)
            return veh_v && vi < j && ((pm >> j) & 1) && bdx * bdx + bdy * bdy <= reach * reach;
          },
          [&](int pair, bool, int count) {
            // is_conflict_possible tests the 11 predicted poses of a pair one after the other (regulation.py:95-110; any one
            // that intersects makes the conflict).  Rounds 1-5 ran that loop per pair and thread -- eleven trips of the wavefront,
            // and the rotated-rectangle test (~400 instructions) in every trip in which ANY pair came within LENGTH -- now the
            // (pair, sample) items of the pass are spread over the wavefront, one per thread: a handful of candidate pairs x 11
            // samples are one or two trips.  A hit sets bit 15 of the pair's list entry (vi | j << 8 uses 14 bits).
            const int width = (int)blockDim.x, n_items = count * HWY_IX_SAMPLES;
            unsigned *const words = reinterpret_cast<unsigned *>(sh.plist);
            HWY_WAVE_LDS_FENCE();  // (every thread has read its own entry before anybody marks one)
            for (int it0 = 0; it0 < n_items; it0 += width) {  // wave-uniform
              const int it = it0 + i;
              const bool valid = it < n_items;
              const int q = valid ? (int)(((unsigned)it * 47663u) >> 19) : 0;  // it / 11 (exact below 5 000)
              const int k = valid ? it - HWY_IX_SAMPLES * q : 0;
              const int pw = (int)sh.plist[q] & 0x3fff;
              const int a = pw & 255, b = pw >> 8;  // a < b: v1 = a, v2 = b
              const double ax = sh.traj[k][0][a], ay = sh.traj[k][1][a], bx = sh.traj[k][0][b], by = sh.traj[k][1][b];
              const double dx = bx - ax, dy = by - ay;
              bool conflict = false;
              if (valid && !(sqrt(dx * dx + dy * dy) > HWY_VEH_LENGTH)) {
                const double ah = sh.traj[k][2][a], bh = sh.traj[k][2][b];
                double ca, sa, cb, sb;
                sincos_bounded(ah, &sa, &ca);
                sincos_bounded(bh, &sb, &cb);
                // rotated_rectangles_intersect(rect(lower slot), rect(higher slot))
                conflict = ix_corner_inside(ax, ay, ca, sa, bx, by, cb, sb) || ix_corner_inside(bx, by, cb, sb, ax, ay, ca, sa);
              }
              if (conflict) __hip_atomic_fetch_or(&words[q >> 1], 0x8000u << (16 * (q & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            HWY_WAVE_LDS_FENCE();
            if (pair >= 0 && ((words[i >> 1] >> (16 * (i & 1))) & 0x8000u)) {  // (read through the type the marks were written in)
              const int a = pair & 255, b = pair >> 8;
              const int p1 = sh.prio[sh.vlane[a]], p2 = sh.prio[sh.vlane[b]];
              bool low_yields;
              if (p1 > p2) low_yields = false;
              else if (p1 < p2) low_yields = true;
              else {
                const double f1 = sh.c[a] * (sh.x[b] - sh.x[a]) + sh.s[a] * (sh.y[b] - sh.y[a]);  // a.front_distance_to(b)
                const double f2 = sh.c[b] * (sh.x[a] - sh.x[b]) + sh.s[b] * (sh.y[a] - sh.y[b]);  // b.front_distance_to(a)
                low_yields = f1 > f2;
              }
              sh.flag[low_yields ? a : b] = 1;
            }
          });
      HWY_WAVE_LDS_FENCE();
      const bool yield = i < SH::kCap && sh.flag[i] != 0;
      if (present && yield && !controlled) {  // only a ControlledVehicle that is not the MDPVehicle is stopped
        me.ts = 0.0;
        me.flags |= HWY_F_YIELDING;
      }
    }

    wave_turn(turn);
    // ---- E. Vehicle.step (kinematics.py:130-177, behavior.py:139-148) -------------------------------------------------
    if (present) {
      if (!controlled) me.timer += p.dt;
      if (!acts) tb = 0.0;
      if (crashed0) {  // clip_actions
        tb = 0.0;
        accel = -1.0 * me.v;
      }
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
    HWY_WAVE_LDS_FENCE();  // the trajectories (if any) are dead: sl[][] is written again
    {
      int cl_new, bits_new;  // on_state_update + the next frame's membership bits and s table
      ix_lane_pass(ip, sh, present, me.x, me.y, me.h, me.tgt, &bits_new, &cl_new, &lat_tgt);
      if (present) me.lane = cl_new;
      bits = bits_new;  // (0 for an empty slot; a helper lane keeps its vehicle's bits for the mask ballots)
    }

    // ---- F. collisions (road.py:477-481, objects.py:92-138): every pair; the highest partner slot's impact stays ----
    {
      const int vi = mp.vi;
      const double c2 = me.ch, s2 = me.sh;
      HWY_WAVE_LDS_FENCE();
      sh.x[i] = me.x; sh.y[i] = me.y; sh.v[i] = me.v; sh.c[i] = c2; sh.s[i] = s2;
      HWY_WAVE_LDS_FENCE();
      // (helper groups: thread t looks for the partners of vehicle vi among the slots g, g + G, ..)
      const Body mine = mp.G > 1 ? Body{sh.x[vi], sh.y[vi], sh.v[vi], sh.c[vi], sh.s[vi]} : Body{me.x, me.y, me.v, c2, s2};
      const bool present_v = ((pm >> vi) & 1) != 0;
      if (i < SH::kCap) { sh.jmax[i] = -1; sh.flag[i] = 0; }
      HWY_WAVE_LDS_FENCE();
      ix_for_items(
          sh, mp.top, mp,
          [&](int j) {
            const double ox = sh.x[j], oy = sh.y[j], ov = sh.v[j];
            const double dx = ox - mine.x, dy = oy - mine.y;
            const double lim = 5.5 + fmax(fabs(mine.v), fabs(ov)) * p.dt;
            // objects.py:124-127 (the pre-check sphere: a dozen instructions per trip)
            return present_v && vi < j && ((pm >> j) & 1) && dx * dx + dy * dy <= lim * lim;
          },
          [&](int pair, bool, int) {
            // one PAIR per thread: the provable-separation test (hwy_device.h: provably (False, False) without the SAT) and, if
            // any pair of the wavefront survives it, the SAT.  (Rounds 1-3 ran the separation test inside the partner loop,
            // under divergence, once per trip in which any thread had a partner inside its sphere: nearly every trip in a queue.)
            const int a = pair < 0 ? 0 : (pair & 255), b = pair < 0 ? 0 : (pair >> 8);  // a < b: the reference's `self` and `other`
            const Body A{sh.x[a], sh.y[a], sh.v[a], sh.c[a], sh.s[a]}, Bb{sh.x[b], sh.y[b], sh.v[b], sh.c[b], sh.s[b]};
            const bool cnd = pair >= 0 && !surely_apart(A, Bb, p.dt);
            int r = 0;
            double tx = 0.0, ty = 0.0;
            if (__ballot(cnd) != 0) {  // wave-uniform
              if (cnd) {
                r = pair_collide(A, Bb, p.dt, &tx, &ty);
                if (r & 1) sh.flag[a] = sh.flag[b] = 1;
                if (r & 2) {  // the impact of the HIGHEST partner slot stays (the reference's loop overwrites)
                  __hip_atomic_fetch_max(&sh.jmax[a], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  __hip_atomic_fetch_max(&sh.jmax[b], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
              }
              HWY_WAVE_LDS_FENCE();
              if (r & 2) {
                if (sh.jmax[a] == b) { sh.bcx[a] = tx / 2; sh.bcy[a] = ty / 2; }
                if (sh.jmax[b] == a) { sh.bcx[b] = -tx / 2; sh.bcy[b] = -ty / 2; }
              }
            }
          });
      HWY_WAVE_LDS_FENCE();
      const int j_imp = i < SH::kCap ? sh.jmax[i] : -1;
      const double imp_x = j_imp >= 0 ? sh.bcx[i] : 0.0, imp_y = j_imp >= 0 ? sh.bcy[i] : 0.0;
      const bool crash = i < SH::kCap && sh.flag[i] != 0;
      if (present && j_imp >= 0) {
        me.impx = imp_x;
        me.impy = imp_y;
        me.flags |= HWY_F_HAS_IMPACT;
      }
      if (present && crash) me.flags |= HWY_F_CRASHED;
    }
  }
}

// ---- OccupancyGridObservation.observe (observation.py:354-413) for the controlled vehicle, same scheme as observe_grid in
//      hwy_device.h (atomic-min cell ownership: the lowest slot wins like the reference's reverse iteration; owners write
//      their features; a cell-strided pass writes the on-road layer and the zeros), with the on-road layer painted from
//      the waypoints of EVERY lane of the network (fill_road_layer_by_lanes, :454-484: straight lanes of any direction
//      and circular arcs).  sh.brho doubles as the per-lane origin table. ------------------------------------------------
// Workspace access of the grid observation: LDS (plain accesses, one wavefront per workgroup) or the global workspace
__device__ inline void ix_ws_store(int32_t *q, int32_t v, bool lds) { if (lds) *q = v; else grid_ws_store(q, v); }
__device__ inline int32_t ix_ws_load(int32_t *q, bool lds) { return lds ? *q : grid_ws_load(q); }

template <typename SH>
__device__ inline void ix_observe_grid(const IxParams &ip, SH &sh, int e, int a, const IxVeh &me, bool present, int ia, int eo) {
  const StepParams &p = ip.s;
  const int i = threadIdx.x, NT = (int)blockDim.x;
  const int W = p.gW, H = p.gH, WH = W * H, F = p.F;
  const double ex = wave_bcast(me.x, ia), ey = wave_bcast(me.y, ia), ev = wave_bcast(me.v, ia);
  const double ec = wave_bcast(me.ch, ia), es = wave_bcast(me.sh, ia);
  // The frames are over: the s table / trajectories (8 KB of LDS) are dead and hold the workspace of the observation --
  // cell owners, on-road cells, and the list of the DISTINCT waypoints of the on-road layer -- when it fits (BASELINE
  // config 4: 11 x 11 cells); larger grids go through the global workspace like the other scenarios' (hwy_device.h).
  const int per_lane = p.g_nwp + 1;
  int32_t *const lds_ws = reinterpret_cast<int32_t *>(&sh.traj[0][0][0]);
  const int lds_ints = (int)(sizeof(sh.traj) / sizeof(int32_t));
  const int list_cap = (lds_ints - 2 * WH - 2 * HWY_MAX_GLANES) * 4;  // bytes left for the waypoint list
  const bool lds = 2 * WH + 2 * HWY_MAX_GLANES < lds_ints;              // wave-uniform
  int32_t *own = lds ? lds_ws : p.grid_ws + ((size_t)e * p.A + a) * 2 * (size_t)WH, *road = own + WH;
  int32_t *w_off = lds_ws + 2 * WH, *w_j0 = w_off + HWY_MAX_GLANES;
  unsigned char *w_lane = reinterpret_cast<unsigned char *>(w_j0 + HWY_MAX_GLANES);
  float *out = p.obs + ((size_t)eo * p.A + a) * (size_t)F * WH;  // eo: output row (hwy_wave.h: observe_wave)
  __syncthreads();
  for (int t = i; t < WH; t += NT) {
    ix_ws_store(own + t, 0x7fffffff, lds);
    ix_ws_store(road + t, 0, lds);
  }
  __syncthreads();
  int my_ci = -1, my_cj = -1;
  if (present) {
    double x = me.x - ex, y = me.y - ey;
    if (p.rx0 > -__builtin_inf()) x = lmap(lmap(x, p.rx0, p.rx1, -1.0, 1.0), -1.0, 1.0, p.rx0, p.rx1);
    if (p.ry0 > -__builtin_inf()) y = lmap(lmap(y, p.ry0, p.ry1, -1.0, 1.0), -1.0, 1.0, p.ry0, p.ry1);
    int ci, cj;
    grid_cell(p, x, y, ec, es, &ci, &cj);
    if (0 <= ci && ci < W && 0 <= cj && cj < H) {
      my_ci = ci;
      my_cj = cj;
      // the lowest slot of a cell owns it (the reference scatters the list in REVERSE): ds_min / L2 atomic min
      if (lds) __hip_atomic_fetch_min(own + ci * H + cj, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else grid_ws_min(own + ci * H + cj, i);
    }
  }
  bool has_road = false;
  for (int f = 0; f < F; ++f) has_road |= (p.feat[f] == HWY_FEAT_ON_ROAD);
  if (has_road) {  // wave-uniform
    // origin of lane L = lane.local_coordinates(observer.position)[0], one lane per thread
    // np.arange(origin - 100, origin + 100, spacing) holds ceil(((origin + 100) - (origin - 100)) / spacing) waypoints:
    // g_nwp or, when the rounded difference exceeds 200, one more -- and on these short lanes that last waypoint is
    // clipped to the lane's END, which can be a cell of its own, so the count is evaluated per lane like numpy does
    int cnt = 0, j0 = 0;
    if (i < ip.n_lanes) {
      double s, lat;
      ix_local(sh, i, ex, ey, &s, &lat);
      sh.brho[i] = s;
      // Most of the 200 m window lies before the start or beyond the end of these short lanes, and every waypoint there
      // is clipped to the same end point: waypoint j is (s - 100) + j * spacing, so all j below jl are <= -spacing (clipped
      // to 0 like jl itself) and all j above jh are >= length + spacing (clipped to the end like jh): [jl, jh] visits every
      // DISTINCT clipped waypoint, i.e. paints the same cells
      const int count = (int)ceil(((s + 100.0) - (s - 100.0)) / p.g_spacing);
      const int jl = (int)floor((100.0 - s) / p.g_spacing) - 1, jh = (int)floor((100.0 - s + sh.len[i]) / p.g_spacing) + 2;
      j0 = jl < 0 ? 0 : jl;
      const int j1 = jh > count - 1 ? count - 1 : jh;
      cnt = j1 >= j0 ? j1 - j0 + 1 : 0;
    }
    int off = 0, total = 0;
    for (int k = 0; k < ip.n_lanes; ++k) {  // wave-uniform
      const int c = wave_bcast_i(cnt, k);
      off += k < i ? c : 0;
      total += c;
    }
    const bool listed = lds && total <= list_cap;  // wave-uniform
    if (listed) {
      if (i < ip.n_lanes) {
        w_off[i] = off;
        w_j0[i] = j0;
        for (int q = 0; q < cnt; ++q) w_lane[off + q] = (unsigned char)i;
      }
    }
    __syncthreads();
    const int n_items = listed ? total : ip.n_lanes * per_lane;
    for (int t = i; t < n_items; t += NT) {
      int k, j;
      if (listed) {
        k = w_lane[t];
        j = w_j0[k] + (t - w_off[k]);
      } else {
        k = t / per_lane;
        j = t - k * per_lane;
      }
      const double o = sh.brho[k];
      if (j >= (int)ceil(((o + 100.0) - (o - 100.0)) / p.g_spacing)) continue;
      const double wp = clipd((o - 100.0) + j * p.g_spacing, 0.0, sh.len[k]);
      double px, py;
      ix_position(sh, k, wp, &px, &py);
      int ci, cj;
      grid_cell(p, px - ex, py - ey, ec, es, &ci, &cj);
      if (0 <= ci && ci < W && 0 <= cj && cj < H) ix_ws_store(road + ci * H + cj, 1, lds);
    }
  }
  __syncthreads();
  const bool clip = (p.flags & HWY_C_OBS_CLIP) != 0;
  if (my_ci >= 0 && ix_ws_load(own + my_ci * H + my_cj, lds) == i) {  // I own my cell: write the vehicle layers
    for (int f = 0; f < F; ++f) {
      const int fid = p.feat[f];
      if (fid == HWY_FEAT_ON_ROAD) continue;
      double val = fid == HWY_FEAT_PRESENCE ? 1.0 : fid == HWY_FEAT_X ? me.x : fid == HWY_FEAT_Y ? me.y
                 : fid == HWY_FEAT_VX ? me.v * me.ch : fid == HWY_FEAT_VY ? me.v * me.sh : fid == HWY_FEAT_HEADING ? me.h
                 : fid == HWY_FEAT_COS_H ? me.ch : fid == HWY_FEAT_SIN_H ? me.sh : 0.0;
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
    if (p.feat[f] == HWY_FEAT_ON_ROAD) out[t] = ix_ws_load(road + c, lds) ? ((p.flags & HWY_C_GRID_IMAGE) ? 255.0f : 1.0f) : 0.0f;
    else if (ix_ws_load(own + c, lds) == 0x7fffffff) out[t] = 0.0f;  // NaN (empty) -> 0
  }
  __syncthreads();
}

// ---- KinematicObservation (observation.py:234-276, road.py:421-450) + IntersectionEnv reward / termination -----------
template <typename SH>
__device__ inline void ix_observe(const IxParams &ip, SH &sh, int e, const IxVeh &me, bool write_reward, int eo = -1) {
  eo = eo < 0 ? e : eo;  // row of the output planes (== e except in a multi-step launch, hwy_rollout_device)
  const StepParams &p = ip.s;
  const int i = threadIdx.x;
  const bool present = !(me.flags & HWY_F_ABSENT);
  const bool controlled = present && (me.flags & HWY_F_CONTROLLED);
  const u64 egos = __ballot(controlled);
  if (egos == 0) return;
  const int V = p.V, F = p.F;
  // Vehicle.destination_direction (kinematics.py:211-235) for the cos_d / sin_d features: the unit vector towards the end of
  // the LAST lane of my route (zeros without a route or at the destination itself)
  double dest_x = 0.0, dest_y = 0.0;
  {
    bool wanted = false;  // wave-uniform
    for (int f = 0; f < F; ++f) wanted |= p.feat[f] == HWY_FEAT_COS_D || p.feat[f] == HWY_FEAT_SIN_D;
    if (wanted && present && route_len(me.route) > 0) {
      const int last = route_at(me.route, route_len(me.route) - 1);
      double px, py;
      ix_position(sh, last, sh.len[last], &px, &py);
      const double ex = px - me.x, ey = py - me.y;
      if (ex != 0.0 || ey != 0.0) {
        const double nrm = sqrt(ex * ex + ey * ey);
        dest_x = ex / nrm;
        dest_y = ey / nrm;
      }
    }
  }
  // Vehicle.lane_offset (kinematics.py:228-235) for the long_off / lat_off / ang_off features: my coordinates on my CURRENT lane
  // and the heading relative to it (lane.py:145-147)
  double off_s = 0.0, off_lat = 0.0, off_ang = 0.0;
  {
    bool wanted = false;  // wave-uniform
    for (int f = 0; f < F; ++f)
      wanted |= p.feat[f] == HWY_FEAT_LONG_OFF || p.feat[f] == HWY_FEAT_LAT_OFF || p.feat[f] == HWY_FEAT_ANG_OFF;
    if (wanted && present) {
      ix_local(sh, me.lane, me.x, me.y, &off_s, &off_lat);
      off_ang = wrap_to_pi(me.h - ix_heading_at(sh, me.lane, off_s));
    }
  }
  // MultiAgentObservation.observe (observation.py:733-734): agent a == the a-th controlled vehicle of the list
  int a = 0;
  for (u64 am = egos; am && a < p.A; am &= am - 1, ++a) {
    const int ia = ctz64(am);
    if (p.obs && p.obs_type == HWY_OBS_OCCUPANCY_GRID) ix_observe_grid(ip, sh, e, a, me, present, ia, eo);
    if (!(p.obs && p.obs_type == HWY_OBS_KINEMATICS)) continue;
    const double ex = wave_bcast(me.x, ia), ey = wave_bcast(me.y, ia), ev = wave_bcast(me.v, ia);
    const double ech = wave_bcast(me.ch, ia), esh = wave_bcast(me.sh, ia);  // (cross-lane reads stay in uniform control flow)
    const int elane = wave_bcast_i(me.lane, ia);
    const double dxe = me.x - ex, dye = me.y - ey;
    // observer.lane_distance_to(me): both projected on the observer's lane (wave-uniform lane: no divergence)
    double s_mine, lat_unused;
    ix_local(sh, elane, me.x, me.y, &s_mine, &lat_unused);
    const double d_lane = s_mine - wave_bcast(s_mine, ia);
    const bool elig = present && i != ia && (sqrt(dxe * dxe + dye * dye) < p.perception) &&
                      ((p.flags & HWY_C_OBS_SEE_BEHIND) || (-2 * HWY_VEH_LENGTH < d_lane));
    const double key = elig ? ((p.flags & HWY_C_OBS_UNSORTED) ? 0.0 : fabs(d_lane)) : __builtin_inf();  // (sort=False: list order)
    const int n_elig = __popcll(__ballot(elig));
    const int mrows = n_elig < V - 1 ? n_elig : V - 1;
    int pos = 0;  // stable sort position (ties keep list order)
    for (u64 em = __ballot(elig); em; em &= em - 1) {
      const int k = ctz64(em);
      const double kk = wave_bcast(key, k);
      pos += ((kk < key) || (kk == key && k < i)) ? 1 : 0;
    }
    float *out = p.obs + ((size_t)eo * p.A + a) * (size_t)(V * F);
    const int row = (i == ia) ? 0 : (elig && pos < V - 1 ? pos + 1 : -1);
    if (present && row >= 0) {
      const double ch = me.ch, shh = me.sh;
      for (int f = 0; f < F; ++f) {
        const int fid = p.feat[f];
        double val = fid == HWY_FEAT_PRESENCE ? 1.0 : fid == HWY_FEAT_X ? me.x : fid == HWY_FEAT_Y ? me.y
                   : fid == HWY_FEAT_VX ? me.v * ch : fid == HWY_FEAT_VY ? me.v * shh : fid == HWY_FEAT_HEADING ? me.h
                   : fid == HWY_FEAT_COS_H ? ch : fid == HWY_FEAT_SIN_H ? shh : 0.0;
        // to_dict(origin, observe_intentions) zeroes the destination of the OTHER vehicles unless intentions are observed;
        // the observer's own row is to_dict() with the default True (observation.py:239,253; kinematics.py:255-256)
        if ((fid == HWY_FEAT_COS_D || fid == HWY_FEAT_SIN_D) && (row == 0 || (p.flags & HWY_C_OBS_INTENTIONS)))
          val = fid == HWY_FEAT_COS_D ? dest_x : dest_y;
        if (fid == HWY_FEAT_LONG_OFF || fid == HWY_FEAT_LAT_OFF || fid == HWY_FEAT_ANG_OFF)
          val = fid == HWY_FEAT_LONG_OFF ? off_s : fid == HWY_FEAT_LAT_OFF ? off_lat : off_ang;
        const bool rel = fid == HWY_FEAT_X || fid == HWY_FEAT_Y || fid == HWY_FEAT_VX || fid == HWY_FEAT_VY;
        if (row > 0 && rel && !(p.flags & HWY_C_OBS_ABSOLUTE)) {
          const double origin = fid == HWY_FEAT_X ? ex : fid == HWY_FEAT_Y ? ey : fid == HWY_FEAT_VX ? ev * ech : ev * esh;
          val -= origin;
        }
        if (rel && (p.flags & HWY_C_OBS_NORMALIZE)) {
          const double r0 = fid == HWY_FEAT_X ? p.rx0 : fid == HWY_FEAT_Y ? p.ry0 : fid == HWY_FEAT_VX ? p.rvx0 : p.rvy0;
          const double r1 = fid == HWY_FEAT_X ? p.rx1 : fid == HWY_FEAT_Y ? p.ry1 : fid == HWY_FEAT_VX ? p.rvx1 : p.rvy1;
          if (r0 > -__builtin_inf()) {
            val = lmap(val, r0, r1, -1.0, 1.0);
            if (p.flags & HWY_C_OBS_CLIP) val = clipd(val, -1.0, 1.0);
          }
        }
        out[row * F + f] = (float)val;
      }
    }
    for (int t = i; t < V * F; t += (int)blockDim.x)
      if (t / F > mrows) out[t] = 0.0f;
  }
  if (!write_reward) return;
  // every controlled vehicle rates itself (_agent_reward, intersection_env.py:79-105); the episode ends when ANY of them
  // has crashed, ALL have arrived, or the FIRST one has left the road (:107-112)
  const int agent = __popcll(egos & (((u64)1 << i) - 1));
  const bool rated = controlled && agent < p.A;
  const bool crashed = rated && (me.flags & HWY_F_CRASHED);
  bool arrived = false, on_road = false;
  if (rated) {
    double s, lat;
    ix_local(sh, me.lane, me.x, me.y, &s, &lat);
    arrived = sh.exitl[me.lane] && s >= 25.0;  // has_arrived (intersection_env.py:340-345)
    on_road = fabs(lat) <= sh.wid[me.lane] / 2 + 0.0 && -5.0 <= s && s < sh.len[me.lane] + 5.0;
  }
  const u64 rated_m = __ballot(rated), crashed_m = __ballot(crashed), arrived_m = __ballot(arrived);
  if (rated) {
    const double scaled_speed = lmap(me.v, p.rs0, p.rs1, 0.0, 1.0);
    double reward = 0.0;
    reward = reward + p.collision_reward * (crashed ? 1.0 : 0.0);
    reward = reward + p.high_speed_reward * clipd(scaled_speed, 0.0, 1.0);
    reward = reward + ip.arrived_reward * (arrived ? 1.0 : 0.0);
    reward = reward + 0 * (on_road ? 1.0 : 0.0);
    reward = arrived ? ip.arrived_reward : reward;
    reward *= (on_road ? 1.0 : 0.0);
    if (p.flags & HWY_C_NORMALIZE_REWARD) reward = lmap(reward, p.collision_reward, ip.arrived_reward, 0.0, 1.0);
    const size_t o = (size_t)eo * p.A + agent;
    p.reward[o] = reward;
    if (p.info_speed) p.info_speed[o] = me.v;
    // bit 0: vehicle.crashed; bit 1: has_arrived(vehicle) -- agents_terminated is either (:119-121)
    if (p.info_crashed) p.info_crashed[o] = (crashed ? 1 : 0) | (arrived ? 2 : 0);
    if (agent == 0) {
      const bool term = crashed_m != 0 || arrived_m == rated_m || ((p.flags & HWY_C_OFFROAD_TERMINAL) && !on_road);
      const double t = p.st.time[e] + p.policy_dt;
      const bool trunc = t >= p.duration;
      p.st.time[e] = t;
      p.terminated[eo] = term ? 1 : 0;
      p.truncated[eo] = trunc ? 1 : 0;
      if (p.autoreset) p.st.done[e] = (term || trunc) ? 1 : 0;
    }
  }
}

// ---- device-side traffic management (Philox draws, NOT numpy's stream) ---------------------------------------------------
// standard normal from two uniforms (Box-Muller); u0 in [0,1) -> 1 - u0 in (0,1]
__device__ inline double ix_normal(double u0, double u1) { return sqrt(-2.0 * log(1.0 - u0)) * cos(2 * HWY_PI * u1); }

// stable compaction of the slots that stay (the reference rebuilds the list, intersection_env.py:333-338)
__device__ inline void ix_compact(IxVeh &me, bool keep) {
  const int i = threadIdx.x;
  const u64 km = __ballot(keep);
  const int n_keep = __popcll(km);
  const int dst = keep ? __popcll(km & (((u64)1 << i) - 1)) : n_keep + __popcll(~km & (((u64)1 << i) - 1));
#define MOVE_I(f) f = wave_send_i(f, dst)
#define MOVE_D(f) f = __hiloint2double(wave_send_i(__double2hiint(f), dst), wave_send_i(__double2loint(f), dst))
  int flags = keep ? me.flags : HWY_F_ABSENT;
  MOVE_D(me.x); MOVE_D(me.y); MOVE_D(me.h); MOVE_D(me.v); MOVE_D(me.timer); MOVE_D(me.ts); MOVE_D(me.delta);
  MOVE_D(me.impx); MOVE_D(me.impy); MOVE_D(me.ch); MOVE_D(me.sh);
  MOVE_I(me.lane); MOVE_I(me.tgt); MOVE_I(me.sidx); MOVE_I(flags);
  {
    double rt = __longlong_as_double(me.route);
    MOVE_D(rt);
    me.route = __double_as_longlong(rt);
  }
  me.flags = flags;
  me.dirty = 1;  // the list moved up: this slot may hold another vehicle now (write its constants back)
#undef MOVE_I
#undef MOVE_D
}

// get_closest_lane_index (road.py:55-71) of ONE wave-uniform pose: thread L measures lane L, the minimum (first one in
// table order) is picked from the broadcast distances.  Call in wave-uniform control flow.
template <typename SH>
__device__ inline int ix_closest_lane_uniform(const IxParams &ip, const SH &sh, double px, double py, double ph) {
  const int i = threadIdx.x;
  double d = 0.0;
  if (i < ip.n_lanes) {
    double s, lat;
    ix_local(sh, i, px, py, &s, &lat);
    const double angle = fabs(wrap_to_pi(ph - ix_heading_at(sh, i, s)));
    d = fabs(lat) + fmax(s - sh.len[i], 0.0) + fmax(0 - s, 0.0) + 1.0 * angle;
  }
  int best = 0;
  double bd = wave_bcast(d, 0);
  for (int L = 1; L < ip.n_lanes; ++L) {
    const double dl = wave_bcast(d, L);
    if (dl < bd) { bd = dl; best = L; }
  }
  return best;
}

// IntersectionEnv._spawn_vehicle (intersection_env.py:292-324) on given draws; thread `slot` becomes the new vehicle
template <typename SH>
__device__ inline void ix_spawn(const IxParams &ip, SH &sh, IxVeh &me, double longitudinal, double position_deviation,
                                double speed_deviation, double spawn_probability, bool go_straight, double u_spawn,
                                double u_r0, double u_r1, double u_z0, double u_z1, double u_z2, double u_z3,
                                double u_delta) {
  const StepParams &p = ip.s;
  const int i = threadIdx.x;
  if (u_spawn > spawn_probability) return;  // wave-uniform
  const double z_pos = ix_normal(u_z0, u_z1), z_speed = ix_normal(u_z2, u_z3);  // (only a spawn pays for the normals)
  // route = choice(range(4), size=2, replace=False): r0 uniform over 4, r1 uniform over the other 3
  int r0 = (int)(u_r0 * 4);
  r0 = r0 > 3 ? 3 : r0;
  int r1 = (int)(u_r1 * 3);
  r1 = r1 > 2 ? 2 : r1;
  r1 = r1 >= r0 ? r1 + 1 : r1;
  if (go_straight) r1 = (r0 + 2) % 4;
  const int access = ip.access_lane[r0];
  double nx, ny;
  const double lon = longitudinal + 5.0 + z_pos * position_deviation;
  ix_position(sh, access, lon, &nx, &ny);
  const double nh = ix_heading_at(sh, access, lon);
  const double speed = 8.0 + z_speed * speed_deviation;
  const bool present = !(me.flags & HWY_F_ABSENT);
  const double dx = me.x - nx, dy = me.y - ny;
  if (__ballot(present && sqrt(dx * dx + dy * dy) < 15) != 0) return;  // too close to somebody
  const u64 pm = __ballot(present);
  const int slot = __popcll(pm);  // the list is compact
  // every spawn the reference would perform is counted, and so is every one dropped because all max_vehicles slots are
  // taken (the reference's list is unbounded, intersection_env.py:324-352): hwy_get_counters reports the rate
  if (ip.counters && i == 0) atomicAdd(&ip.counters[slot >= p.N ? HWY_CTR_IX_SPAWNS_DROPPED : HWY_CTR_IX_SPAWNS], 1ull);
  if (slot >= p.N) return;        // capacity reached: no spawn (documented deviation; size num_vehicles accordingly)
  const int best = ix_closest_lane_uniform(ip, sh, nx, ny, nh);  // lane index: get_closest_lane_index(position, heading)
  if (i == slot) {
    me = IxVeh{};
    me.dirty = 1;
    me.x = nx; me.y = ny; me.h = nh; me.v = speed; me.ts = speed;
    me.lane = me.tgt = best;
    me.timer = py_mod_pos((nx + ny) * HWY_PI, HWY_LC_DELAY);
    me.route = ix_plan_route(ip, sh, best, r1);
    me.delta = 3.5 + (4.5 - 3.5) * u_delta;  // randomize_behavior (behavior.py:66-69)
    me.flags = HWY_F_CHECK_COLLISIONS;
    sincos_bounded(me.h, &me.sh, &me.ch);
  }
}

// IntersectionEnv.step's tail (intersection_env.py:136-140): _clear_vehicles + one _spawn_vehicle
template <typename SH>
__device__ inline void ix_clear_spawn(const IxParams &ip, SH &sh, IxVeh &me, uint64_t seed, uint32_t episode,
                                      uint32_t step_no) {
  const bool present = !(me.flags & HWY_F_ABSENT);
  double s = 0.0, lat;
  if (present) ix_local(sh, me.lane, me.x, me.y, &s, &lat);
  const bool leaving = present && sh.exitl[me.lane] && s >= sh.len[me.lane] - 4 * HWY_VEH_LENGTH;
  const bool keep = present && ((me.flags & HWY_F_CONTROLLED) || !leaving);
  if (__ballot(present && !keep) != 0) ix_compact(me, keep);
  double u0, u1, u2, u3, u4, u5, u6, u7;
  philox_uniform2(seed, 1000u + step_no, episode, 0u, &u0, &u1);
  philox_uniform2(seed, 1000u + step_no, episode, 1u, &u2, &u3);
  philox_uniform2(seed, 1000u + step_no, episode, 2u, &u4, &u5);
  philox_uniform2(seed, 1000u + step_no, episode, 3u, &u6, &u7);
  ix_spawn(ip, sh, me, 0.0, 1.0, 1.0, ip.spawn_probability, false, u0, u1, u2, u3, u4, u5, u6, u7);
}

// IntersectionEnv._make_vehicles (intersection_env.py:232-290) on Philox draws, in three parts so that the warm-up can
// be spread over several launches (next-episode pre-warming): the initial random traffic, the simulated seconds, the rest.
template <typename SH>
__device__ inline void ix_spawn_initial(const IxParams &ip, SH &sh, uint64_t seed, uint32_t episode, IxVeh &me) {
  me = IxVeh{};
  me.flags = HWY_F_ABSENT;
  const int n = ip.initial_count;
  for (int t = 0; t < n - 1; ++t) {  // np.linspace(0, 80, n_vehicles)[t]
    double u0, u1, u2, u3, u4, u5, u6, u7;
    philox_uniform2(seed, (uint32_t)t, episode, 0u, &u0, &u1);
    philox_uniform2(seed, (uint32_t)t, episode, 1u, &u2, &u3);
    philox_uniform2(seed, (uint32_t)t, episode, 2u, &u4, &u5);
    philox_uniform2(seed, (uint32_t)t, episode, 3u, &u6, &u7);
    const double lon = n > 1 ? 0.0 + t * ((80.0 - 0.0) / (n - 1)) : 0.0;
    ix_spawn(ip, sh, me, lon, 1.0, 1.0, 0.6, false, u0, u1, u2, u3, u4, u5, u6, u7);
  }
}
// n_frames of the simulated seconds without the ego (the table walk first: bits / s are not part of the stored state)
template <typename SH>
__device__ inline void ix_warm(const IxParams &ip, SH &sh, int e, IxVeh &me, int n_frames, int &road_steps) {
  int bits, unused;
  double lat_tgt;
  __syncthreads();
  ix_lane_pass(ip, sh, !(me.flags & HWY_F_ABSENT), me.x, me.y, me.h, me.tgt, &bits, &unused, &lat_tgt);
  ix_frames(ip, sh, e, me, n_frames, nullptr, road_steps, bits, lat_tgt);
}
template <typename SH>
__device__ inline void ix_spawn_finalise(const IxParams &ip, SH &sh, uint64_t seed, uint32_t episode, IxVeh &me) {
  const StepParams &p = ip.s;
  const int i = threadIdx.x;
  {  // challenger: longitudinal 60, always, straight on, position deviation 0.1, speed deviation 0
    double u0, u1, u2, u3, u4, u5, u6, u7;
    philox_uniform2(seed, 500u, episode, 0u, &u0, &u1);
    philox_uniform2(seed, 500u, episode, 1u, &u2, &u3);
    philox_uniform2(seed, 500u, episode, 2u, &u4, &u5);
    philox_uniform2(seed, 500u, episode, 3u, &u6, &u7);
    ix_spawn(ip, sh, me, 60.0, 0.1, 0.0, 1.0, true, 0.0, u1, u2, u3, u4, u5, u6, u7);
  }
  // the controlled vehicles (:292-318): number k on ("o" + k % 4, "ir" + k % 4, 0) at 60 + 5 * normal(1.0), speed =
  // speed_limit, route to config["destination"] or to "o" + integers(1, 4); after each one the OTHER vehicles within
  // 20 m of it leave ("prevent early collisions", :313-318: controlled vehicles stay)
  for (int k = 0; k < p.A; ++k) {
    double u0, u1;
    philox_uniform2(seed, 501u + (uint32_t)k, episode, 0u, &u0, &u1);
    int destination = ip.destination;
    if (destination < 0) {  // uniform over {1, 2, 3}
      double ud, unused;
      philox_uniform2(seed, 501u + (uint32_t)k, episode, 1u, &ud, &unused);
      const int d = (int)(ud * 3);
      destination = 1 + (d > 2 ? 2 : d);
    }
    const int access = ip.access_lane[k & 3];
    double ex, ey;
    ix_position(sh, access, 60.0 + 5.0 * (1.0 + ix_normal(u0, u1)), &ex, &ey);
    const bool present = !(me.flags & HWY_F_ABSENT);
    const double dx = me.x - ex, dy = me.y - ey;
    const bool keep = present && ((me.flags & HWY_F_CONTROLLED) || !(sqrt(dx * dx + dy * dy) < 20));
    ix_compact(me, keep);
    int slot = __popcll(__ballot(!(me.flags & HWY_F_ABSENT)));
    if (slot >= p.N) {
      // every slot is taken (the reference's list is unbounded, intersection_env.py:302-311): a controlled vehicle is never
      // the one that is dropped -- it replaces the most recently spawned traffic vehicle, counted as a dropped spawn
      // (N >= A is validated at hwy_create, so there is one)
      const unsigned long long traffic = __ballot(!(me.flags & HWY_F_ABSENT) && !(me.flags & HWY_F_CONTROLLED));
      const int victim = 63 - __clzll(traffic);
      ix_compact(me, !(me.flags & HWY_F_ABSENT) && i != victim);
      if (ip.counters && i == 0) atomicAdd(&ip.counters[HWY_CTR_IX_SPAWNS_DROPPED], 1ull);
      slot = p.N - 1;
    }
    const double eh = ix_heading_at(sh, access, 60.0);
    const int best = ix_closest_lane_uniform(ip, sh, ex, ey, eh);
    if (i == slot && slot < p.N) {
      me = IxVeh{};
      me.dirty = 1;
      me.x = ex; me.y = ey; me.h = eh; me.v = sh.lim[access];
      me.lane = me.tgt = best;
      me.route = ix_plan_route(ip, sh, best, destination);
      const double xs = (me.v - p.target_speeds[0]) / (p.target_speeds[p.n_ts - 1] - p.target_speeds[0]);
      me.sidx = (int)clipd(rint(xs * (p.n_ts - 1)), 0.0, (double)(p.n_ts - 1));
      me.ts = p.target_speeds[me.sidx];
      me.flags = HWY_F_CONTROLLED | HWY_F_CHECK_COLLISIONS;
      sincos_bounded(me.h, &me.sh, &me.ch);
    }
  }
}
#ifndef HWY_RELOAD_IX_PARAMS
#define HWY_RELOAD_IX_PARAMS(q, ip)                              \
  auto kernarg_ = __builtin_amdgcn_kernarg_segment_ptr();        \
  asm volatile("" : "+s"(kernarg_));                           \
  const IxParams &q = *(const IxParams *)kernarg_
#endif
template <typename SH>
__device__ inline void ix_spawn_env(const IxParams &ip, SH &sh, int e, uint64_t seed, uint32_t episode, IxVeh &me,
                                    int &road_steps) {
  ix_spawn_initial(ip, sh, seed, episode, me);
  road_steps = 0;
  ix_warm(ip, sh, e, me, 3 * (int)rint(1 / ip.s.dt), road_steps);
  ix_spawn_finalise(ip, sh, seed, episode, me);
}

// =============================================================================================================
// The step kernel.  One workgroup (one wavefront) per environment and, with next-episode pre-warming, a second one
// (blockIdx >= num_envs) that advances the shadow copy of the environment's NEXT episode by one chunk of warm-up
// frames per launch: the shadow is a pure function of (seed, episode + 1), so WHEN it is computed cannot change any
// result; an environment that finishes before its shadow is ready finishes the remaining frames inline.
//
// Every role runs the SAME three stages -- set up the vehicle planes, run n frames, finish -- so that the frame loop, the
// spawn rules and the observation are instantiated ONCE in the kernel: inlined per role they made 250 KB of code, four
// times the instruction cache a pair of CUs shares.
// One block's work of one launch step: block index `bx` < num_envs steps (or re-spawns) environment bx, bx >= num_envs advances
// the shadow of environment bx - num_envs.  eo = row of the action / output planes (hwy_wave.h: observe_wave).
template <int CAP, int NT>
__device__ __forceinline__ void ix_policy_block(const IxParams &ip, IxSharedT<CAP, NT> &sh, const int bx, const int eo_step,
                                                const bool load_table = true) {
  const StepParams &p = ip.s;
  const int i = threadIdx.x;
  const bool shadow_block = bx >= ip.num_envs;  // wave-uniform, like everything that selects a role below
  const int e = shadow_block ? bx - ip.num_envs : bx;
  const int eo = shadow_block ? e : eo_step;
  const int total = 3 * (int)rint(1 / p.dt);  // frames of the simulated seconds of _make_vehicles
  const bool resetting = p.autoreset && p.st.done[e];
  const int32_t *m = ip.shadow_meta ? ip.shadow_meta + 4 * e : nullptr;  // {episode, progress, RegulatedRoad.steps, -}
  const uint32_t next_episode = p.st.episode[e] + 1u;
  const bool shadow_mine = m && (uint32_t)m[0] == next_episode;
  if (shadow_block) {
    // most of these blocks have nothing to do (the environment is being reset right now -- its step block may be
    // consuming the shadow in this launch -- or the shadow is ready): leave before touching LDS
    if (!p.autoreset || resetting) return;
    if (shadow_mine && m[1] > total) return;
  }
  if (load_table) ix_load_table(ip, sh);  // (block-uniform; a multi-step launch loads it once)
  const uint64_t seed = p.rp.base_seed + (uint64_t)e;

  // ---- stage 1: whose planes, how many frames, what comes after them -------------------------------------------------
  enum { STEP, RESPAWN, PREWARM };
  const int role = shadow_block ? PREWARM : (resetting ? RESPAWN : STEP);
  int n_run = 0, road_steps = 0, prog = 0;
  bool from_scratch = false, finalise = false;
  const int32_t *actions = nullptr;
  uint32_t step_no = 0;
  if (role == STEP) {
    n_run = p.n_frames;
    actions = p.actions ? p.actions + (size_t)(eo - e) * p.A : nullptr;  // (ix_frames indexes [e][agent])
    road_steps = ip.road_steps[e];
    step_no = (uint32_t)rint(p.st.time[e] / p.policy_dt);  // read before anybody advances the clock
  } else {
    const bool usable = shadow_mine && m[1] >= 0;  // (RESPAWN: ready, or partly warmed; PREWARM: in progress)
    from_scratch = !usable;
    prog = usable ? m[1] : 0;
    road_steps = usable ? m[2] : 0;
    const int left = total - prog;  // < 0: the shadow is ready (challenger and ego included)
    n_run = left < 0 ? 0 : (role == PREWARM && ip.prewarm_frames < left ? ip.prewarm_frames : left);
    finalise = left >= 0 && n_run == left;
  }
  IxVeh me;
  if (from_scratch) ix_spawn_initial(ip, sh, seed, next_episode, me);
  else ix_load_vehicle(ip, e, me, role != STEP);

  // ---- stage 2: the frames (the table walk first: bits / s are not part of the stored state) ----------------------------
  if (n_run > 0) {
    int bits, unused;
    double lat_tgt;
    __syncthreads();
    ix_lane_pass(ip, sh, !(me.flags & HWY_F_ABSENT), me.x, me.y, me.h, me.tgt, &bits, &unused, &lat_tgt);
    ix_frames(ip, sh, e, me, n_run, actions, road_steps, bits, lat_tgt);
  }
  if (finalise) ix_spawn_finalise(ip, sh, seed, next_episode, me);

  // ---- stage 3 ---------------------------------------------------------------------------------------------------------
  if (role == PREWARM) {
    ix_store_vehicle(ip, e, me, true);
    __threadfence();
    if (i == 0) {
      int32_t *mw = ip.shadow_meta + 4 * e;
      mw[0] = (int32_t)next_episode; mw[1] = finalise ? total + 1 : prog + n_run; mw[2] = road_steps;
    }
    return;
  }
  if (role == RESPAWN || p.full_step) {
    __syncthreads();
    ix_observe(ip, sh, e, me, role == STEP, eo);
  }
  if (role == STEP && p.full_step && !ip.host_spawn) ix_clear_spawn(ip, sh, me, seed, p.st.episode[e], step_no);
  ix_store_vehicle(ip, e, me, false, role == RESPAWN);  // (a re-spawn moves the episode from the shadow planes to these)
  if (role == RESPAWN) {  // the step after terminated | truncated re-spawned the environment
    __threadfence();  // the shadow has been read before `done` is cleared (the pre-warming block starts over once it sees that)
    if (i < p.A) {
      p.reward[(size_t)eo * p.A + i] = 0.0;
      if (p.info_speed) p.info_speed[(size_t)eo * p.A + i] = sh.lim[ip.access_lane[i & 3]];
      if (p.info_crashed) p.info_crashed[(size_t)eo * p.A + i] = 0;
    }
    if (i == 0) {
      p.st.time[e] = 0.0;
      p.st.done[e] = 0;
      p.st.episode[e] = next_episode;
      p.terminated[eo] = 0;
      p.truncated[eo] = 0;
    }
  }
  if (i == 0) ip.road_steps[e] = road_steps;
}

template <int WPE, int CAP, int NT = CAP>
__global__ void __launch_bounds__(NT, WPE) hwy_ix_step_kernel(const IxParams ip) {
  __shared__ IxSharedT<CAP, NT> sh;
  ix_policy_block<CAP, NT>(ip, sh, (int)blockIdx.x, (int)blockIdx.x);
}

// hwy_rollout_device on the intersection kernel: ip.s.k_steps policy steps per wavefront in one launch (hwy_wave.h:
// hwy_rollout_wave_kernel has the argument).  The grid holds the STEP blocks only: no block advances a shadow during the launch, so
// an environment that ends in it warms its next episode up inline (what it does whenever it ends before its shadow is ready) --
// WHEN the warm-up frames are computed cannot change a result (tests/test_ix_device_traffic.py).
template <int WPE, int CAP, int NT = CAP>
__global__ void __launch_bounds__(NT, WPE) hwy_ix_rollout_kernel(const IxParams ip) {
  __shared__ IxSharedT<CAP, NT> sh;
  const int e = blockIdx.x;
  for (int k = 0; k < ip.s.k_steps; ++k) {  // block-uniform
    HWY_RELOAD_IX_PARAMS(ik, ip);  // a fresh, opaque view of the arguments per step: nothing stays live -- spilled -- across steps
    ix_policy_block<CAP, NT>(ik, sh, e, k * ik.num_envs + e, k == 0);
    __syncthreads();
    __threadfence_block();
  }
}

// Reset kernel: AbstractEnv.reset for the masked environments + first observation.
template <int WPE, int CAP, int NT = CAP>
__global__ void __launch_bounds__(NT, WPE) hwy_ix_reset_kernel(const IxParams ip) {
  const StepParams &p = ip.s;
  __shared__ IxSharedT<CAP, NT> sh;
  const int e = blockIdx.x, i = threadIdx.x;
  ix_load_table(ip, sh);
  if (p.reset_mask && !p.reset_mask[e]) return;  // block-uniform
  IxVeh me;
  int road_steps;
  const uint64_t seed = p.reset_seeds ? p.reset_seeds[e] : p.rp.base_seed + (uint64_t)e;
  ix_spawn_env(ip, sh, e, seed, 0u, me, road_steps);
  ix_observe(ip, sh, e, me, false);
  ix_store_vehicle(ip, e, me, false, true);
  if (i == 0) {
    p.st.time[e] = 0.0;
    p.st.done[e] = 0;
    p.st.episode[e] = 0;
    ip.road_steps[e] = road_steps;
    if (ip.shadow_meta) ip.shadow_meta[4 * e] = -1;  // whatever was pre-warmed belonged to another seed / episode
  }
}

// Observation-only kernel (hwy_observe).
template <int WPE, int CAP, int NT = CAP>
__global__ void __launch_bounds__(NT, WPE) hwy_ix_observe_kernel(const IxParams ip) {
  __shared__ IxSharedT<CAP, NT> sh;
  ix_load_table(ip, sh);
  IxVeh me;
  ix_load_vehicle(ip, blockIdx.x, me);
  ix_observe(ip, sh, blockIdx.x, me, false);
}

}  // namespace hwy

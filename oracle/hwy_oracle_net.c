/*
 * hwy_oracle_net.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's Road.act()/Road.step()
 * hot path on an x-aligned ROAD NETWORK (MergeEnv / MergeGenericEnv, highway_env/envs/merge_env.py), in
 * plain sequential C (glibc libm, f64).  Companion of hwy_oracle.c (single straight road); the same
 * rules apply: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, as the
 * checker; the product never touches it.
 *
 * Parity pin: tests/golden/merge_*.npz recorded from the unmodified Python reference by
 * tests/golden/make_golden_merge.py (per simulation frame and per policy step),
 * tests/test_oracle_golden_merge.py.
 *
 * What is new relative to the straight highway:
 *   lanes are entries of hwy_config.net (StraightLane / SineLane, lane.py:150-283), lane indices are
 *   positions in that table; ControlledVehicle.follow_road + RoadNetwork.next_lane switch the target lane
 *   at the end of a segment (controller.py:135-143, road.py:73-146); forbidden lanes (lane.py:110-111);
 *   the lane-change abort rule only applies on the same road (behavior.py:232); per-lane speed limits
 *   (behavior.py:171-175); Road.objects holds an Obstacle that takes part in neighbour search,
 *   collisions and observations (road.py:421-450,469-547, objects.py:92-120) but never acts;
 *   MergeEnv's reward / termination (merge_env.py:40-82).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/highway_env/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/hwy_engine.h"

#define VEH_LENGTH 5.0
#define VEH_WIDTH 2.0
#define OBJ_LENGTH 2.0 /* RoadObject.LENGTH / WIDTH, objects.py:25-26 */
#define OBJ_WIDTH 2.0
#define MAX_SPEED 40.0
#define MIN_SPEED (-40.0)
#define TAU_ACC 0.6
#define TAU_HEADING 0.2
#define TAU_LATERAL 0.6
#define TAU_PURSUIT (0.5 * TAU_HEADING)
#define KP_A (1.0 / TAU_ACC)
#define KP_HEADING (1.0 / TAU_HEADING)
#define KP_LATERAL (1.0 / TAU_LATERAL)
#define MAX_STEERING_ANGLE (M_PI / 3.0)
#define ACC_MAX 6.0
#define COMFORT_ACC_MAX 3.0
#define COMFORT_ACC_MIN (-5.0)
#define DISTANCE_WANTED (5.0 + VEH_LENGTH)
#define TIME_WANTED 1.5
#define POLITENESS 0.0
#define LANE_CHANGE_MIN_ACC_GAIN 0.2
#define LANE_CHANGE_MAX_BRAKING_IMPOSED 2.0
#define LANE_CHANGE_DELAY 1.0
#define LANE_VEHICLE_LENGTH 5.0

typedef struct {
  double x, y, heading, speed;
  double timer, target_speed, delta;
  double impact_x, impact_y;
  double act_steering, act_accel;
  int lane, target_lane, speed_index;
  int crashed, has_impact, check_collisions, controlled, obstacle, present;
  double impact_margin, flag_margin; /* test diagnostics only (hwy_oracle.c: orc_set_margin_buffer, orc_set_flag_margin_buffer) */
} ent_t;

typedef struct {
  const hwy_config *cfg;
  ent_t *v; /* slots: vehicles (Road.vehicles order), then obstacles (Road.objects); absent slots skipped */
  int n;
} net_t;

/* ---- utils.py ---------------------------------------------------------------------- */
static double not_zero(double x) { /* utils.py:50-56 */
  const double eps = 1e-2;
  if (fabs(x) > eps) return x;
  return x >= 0 ? eps : -eps;
}
static double py_mod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0.0) {
    if ((b < 0) != (m < 0)) m += b;
  } else {
    m = copysign(0.0, b);
  }
  return m;
}
static double wrap_to_pi(double x) { return py_mod(x + M_PI, 2 * M_PI) - M_PI; } /* utils.py:59-60 */
static double lmap(double v, double x0, double x1, double y0, double y1) {         /* utils.py:31-33 */
  return y0 + (v - x0) * (y1 - y0) / (x1 - x0);
}
static double clipd(double a, double lo, double hi) { return fmin(fmax(a, lo), hi); }

/* ---- road/lane.py ------------------------------------------------------------------- */
static int is_veh(const ent_t *e) { return e->present && !e->obstacle; }

/* StraightLane.local_coordinates (lane.py:209-213) with direction (1,0), lateral (-0,1);
 * SineLane.local_coordinates (lane.py:267-271) */
static void lane_local(const hwy_lane *l, double x, double y, double *s, double *lat) {
  double dx = x - l->x0, dy = y - l->y0;
  *s = dx * 1.0 + dy * 0.0;
  *lat = dx * -0.0 + dy * 1.0;
  if (l->amplitude != 0) *lat = *lat - l->amplitude * sin(l->pulsation * *s + l->phase);
}
/* StraightLane.heading_at (lane.py:203-204: arctan2(0, length) = 0); SineLane.heading_at (lane.py:259-265) */
static double lane_heading_at(const hwy_lane *l, double s) {
  if (l->amplitude != 0) return 0.0 + atan(l->amplitude * l->pulsation * cos(l->pulsation * s + l->phase));
  return 0.0;
}
/* StraightLane.position (lane.py:196-201); SineLane.position (lane.py:252-257) */
static void lane_position(const hwy_lane *l, double s, double lat, double *px, double *py) {
  if (l->amplitude != 0) lat = lat + l->amplitude * sin(l->pulsation * s + l->phase);
  *px = l->x0 + s * 1.0 + lat * -0.0;
  *py = l->y0 + s * 0.0 + lat * 1.0;
}
/* AbstractLane.on_lane (lane.py:80-102) */
static int lane_on_lane(const hwy_lane *l, double x, double y, double margin) {
  double s, lat;
  lane_local(l, x, y, &s, &lat);
  return fabs(lat) <= l->width / 2 + margin && -LANE_VEHICLE_LENGTH <= s && s < l->length + LANE_VEHICLE_LENGTH;
}
/* AbstractLane.is_reachable_from (lane.py:104-118) */
static int lane_is_reachable_from(const hwy_lane *l, double x, double y) {
  if (l->forbidden) return 0;
  double s, lat;
  lane_local(l, x, y, &s, &lat);
  return fabs(lat) <= 2 * l->width && 0 <= s && s < l->length + LANE_VEHICLE_LENGTH;
}
/* AbstractLane.after_end (lane.py:120-125) */
static int lane_after_end(const hwy_lane *l, double x, double y) {
  double s, lat;
  lane_local(l, x, y, &s, &lat);
  return s > l->length - LANE_VEHICLE_LENGTH / 2;
}
/* AbstractLane.distance (lane.py:127-130) */
static double lane_distance(const hwy_lane *l, double x, double y) {
  double s, r;
  lane_local(l, x, y, &s, &r);
  return fabs(r) + fmax(s - l->length, 0) + fmax(0 - s, 0);
}
/* AbstractLane.distance_with_heading (lane.py:132-143), local_angle (:145-147) */
static double lane_distance_with_heading(const hwy_lane *l, double x, double y, double heading) {
  double s, r;
  lane_local(l, x, y, &s, &r);
  double angle = fabs(wrap_to_pi(heading - lane_heading_at(l, s)));
  return fabs(r) + fmax(s - l->length, 0) + fmax(0 - s, 0) + 1.0 * angle;
}

/* ---- road/road.py: RoadNetwork -------------------------------------------------------- */
/* get_closest_lane_index (road.py:55-71): np.argmin => first minimum in table order */
static int closest_lane_index(const hwy_config *c, double x, double y, double heading) {
  int best = 0;
  double bd = lane_distance_with_heading(&c->net[0], x, y, heading);
  for (int k = 1; k < c->net_lanes; k++) {
    double d = lane_distance_with_heading(&c->net[k], x, y, heading);
    if (d < bd) { bd = d; best = k; }
  }
  return best;
}
/* next_lane (road.py:73-127) with route=None and one successor road per node;
 * next_lane_given_next_road (road.py:129-146) */
static int next_lane(const hwy_config *c, int cur, double x, double y) {
  const hwy_lane *l = &c->net[cur];
  double s, lat, px, py;
  lane_local(l, x, y, &s, &lat);
  lane_position(l, s, 0, &px, &py); /* projected (desired) position */
  if (l->next_first < 0) return cur; /* KeyError on graph[_to] => current_index */
  if (l->road_lanes == l->next_lanes) return l->next_first + l->id;
  int best = 0;
  double bd = lane_distance(&c->net[l->next_first], px, py);
  for (int k = 1; k < l->next_lanes; k++) { /* min(lanes, key=distance): first minimum */
    double d = lane_distance(&c->net[l->next_first + k], px, py);
    if (d < bd) { bd = d; best = k; }
  }
  return l->next_first + best;
}

/* ---- vehicle/objects.py ---------------------------------------------------------------- */
/* lane_distance_to (objects.py:183-198): along self's CURRENT lane */
static double lane_distance_to(const hwy_config *c, const ent_t *self, const ent_t *other) {
  double s_o, s_s, lat;
  lane_local(&c->net[self->lane], other->x, other->y, &s_o, &lat);
  lane_local(&c->net[self->lane], self->x, self->y, &s_s, &lat);
  return s_o - s_s;
}
static double ent_length(const ent_t *e) { return e->obstacle ? OBJ_LENGTH : VEH_LENGTH; }
static double ent_width(const ent_t *e) { return e->obstacle ? OBJ_WIDTH : VEH_WIDTH; }
/* polygon (objects.py:169-181) */
static void polygon(const ent_t *v, double p[5][2]) {
  const double L = ent_length(v), W = ent_width(v);
  const double lx[4] = {-L / 2, -L / 2, +L / 2, +L / 2};
  const double ly[4] = {-W / 2, +W / 2, +W / 2, -W / 2};
  double c = cos(v->heading), s = sin(v->heading);
  for (int k = 0; k < 4; k++) {
    p[k][0] = (c * lx[k] + -s * ly[k]) + v->x;
    p[k][1] = (s * lx[k] + c * ly[k]) + v->y;
  }
  p[4][0] = p[0][0];
  p[4][1] = p[0][1];
}
static void project_polygon(double p[5][2], const double axis[2], double *mn, double *mx) { /* utils.py:177-185 */
  int first = 1;
  for (int k = 0; k < 5; k++) {
    double pr = p[k][0] * axis[0] + p[k][1] * axis[1];
    if (first || pr < *mn) *mn = pr;
    if (first || pr > *mx) *mx = pr;
    first = 0;
  }
}
static double interval_distance(double min_a, double max_a, double min_b, double max_b) { /* utils.py:188-193 */
  return min_a < min_b ? min_b - max_a : min_a - max_b;
}
/* Test diagnostics (as in hwy_oracle.c): |d.normal| of the axis that oriented the last translation (utils.py:232-236);
 * orc_set_margin_buffer (hwy_oracle.c) hands out the per-slot minimum over the impacts assigned during a call. */
static __thread double g_axis_dn = INFINITY;
static __thread double g_flag_dn = INFINITY; /* smallest |interval distance| behind an intersecting / will_intersect decision */
extern __thread double *orc_margin_buf, *orc_flag_margin_buf;
extern __thread double orc_knife_bias; /* 0 except in a knife-edge replay (hwy_oracle.c: orc_set_knife_bias) */
/* utils.py:196-241 */
static void are_polygons_intersecting(double a[5][2], double b[5][2], const double da[2], const double db[2],
                                      int *intersecting, int *will_intersect, double translation[2]) {
  *intersecting = *will_intersect = 1;
  g_flag_dn = INFINITY;
  double min_distance = INFINITY;
  double axis[2] = {0, 0};
  double(*polys[2])[2] = {a, b};
  for (int pi = 0; pi < 2; pi++) {
    double(*poly)[2] = polys[pi];
    for (int k = 0; k < 4; k++) {
      double *p1 = poly[k], *p2 = poly[k + 1];
      double normal[2] = {-p2[1] + p1[1], p2[0] - p1[0]};
      double nn = sqrt(normal[0] * normal[0] + normal[1] * normal[1]);
      normal[0] /= nn;
      normal[1] /= nn;
      double min_a, max_a, min_b, max_b;
      project_polygon(a, normal, &min_a, &max_a);
      project_polygon(b, normal, &min_b, &max_b);
      if (interval_distance(min_a, max_a, min_b, max_b) > orc_knife_bias) *intersecting = 0;
      g_flag_dn = fmin(g_flag_dn, fabs(interval_distance(min_a, max_a, min_b, max_b)));
      double vp = normal[0] * (da[0] - db[0]) + normal[1] * (da[1] - db[1]);
      if (vp < 0) min_a += vp; else max_a += vp;
      double distance = interval_distance(min_a, max_a, min_b, max_b);
      if (distance > orc_knife_bias) *will_intersect = 0;
      g_flag_dn = fmin(g_flag_dn, fabs(distance));
      if (!*intersecting && !*will_intersect) break;
      if (fabs(distance) < min_distance) {
        min_distance = fabs(distance);
        double ca[2] = {0, 0}, cb[2] = {0, 0};
        for (int q = 0; q < 4; q++) { ca[0] += a[q][0]; ca[1] += a[q][1]; cb[0] += b[q][0]; cb[1] += b[q][1]; }
        double d0 = ca[0] / 4 - cb[0] / 4, d1 = ca[1] / 4 - cb[1] / 4;
        if (d0 * normal[0] + d1 * normal[1] > 0) { axis[0] = normal[0]; axis[1] = normal[1]; }
        else { axis[0] = -normal[0]; axis[1] = -normal[1]; }
        g_axis_dn = fabs(d0 * normal[0] + d1 * normal[1]);
      }
    }
  }
  translation[0] = translation[1] = 0;
  if (*will_intersect) {
    translation[0] = min_distance * axis[0];
    translation[1] = min_distance * axis[1];
  }
}
/* _is_colliding (objects.py:122-138): each object has its own diagonal (objects.py:63) */
static void is_colliding(const ent_t *self, const ent_t *other, double dt, int *intersecting, int *will_intersect,
                         double translation[2]) {
  const double diag_s = sqrt(ent_length(self) * ent_length(self) + ent_width(self) * ent_width(self));
  const double diag_o = sqrt(ent_length(other) * ent_length(other) + ent_width(other) * ent_width(other));
  double dx = other->x - self->x, dy = other->y - self->y;
  if (sqrt(dx * dx + dy * dy) > (diag_s + diag_o) / 2 + self->speed * dt) {
    *intersecting = *will_intersect = 0;
    translation[0] = translation[1] = 0;
    return;
  }
  double pa[5][2], pb[5][2];
  polygon(self, pa);
  polygon(other, pb);
  double da[2] = {self->speed * cos(self->heading) * dt, self->speed * sin(self->heading) * dt};
  double db[2] = {other->speed * cos(other->heading) * dt, other->speed * sin(other->heading) * dt};
  are_polygons_intersecting(pa, pb, da, db, intersecting, will_intersect, translation);
}
/* handle_collisions (objects.py:92-120): everything here is collidable and solid */
static void handle_collisions(ent_t *self, ent_t *other, double dt) {
  if (other == self || !(self->check_collisions || other->check_collisions)) return;
  int intersecting, will_intersect;
  double t[2];
  g_flag_dn = INFINITY;
  is_colliding(self, other, dt, &intersecting, &will_intersect, t);
  self->flag_margin = fmin(self->flag_margin, g_flag_dn);
  other->flag_margin = fmin(other->flag_margin, g_flag_dn);
  if (will_intersect) {
    if (other->obstacle) {
      self->impact_x = t[0]; self->impact_y = t[1]; self->has_impact = 1;
    } else if (self->obstacle) {
      other->impact_x = t[0]; other->impact_y = t[1]; other->has_impact = 1;
    } else {
      self->impact_x = t[0] / 2; self->impact_y = t[1] / 2; self->has_impact = 1;
      other->impact_x = -t[0] / 2; other->impact_y = -t[1] / 2; other->has_impact = 1;
    }
    self->impact_margin = fmin(self->impact_margin, g_axis_dn);
    other->impact_margin = fmin(other->impact_margin, g_axis_dn);
  }
  if (intersecting) {
    self->crashed = 1;
    other->crashed = 1;
  }
}

/* ---- road/road.py:483-547: vehicles + objects.  With HWY_C_CONNECTED_LANES the search list is the lane, then lane `id`
 * (else 0) of the road leaving `_to` (offset +lane.length), then lane `id` (else 0) of every road arriving at `_from`
 * (offset -prev.length), in graph-insertion order == table order; a vehicle counts on the FIRST list entry it is on. ----- */
static void neighbour_vehicles(const net_t *r, const ent_t *vehicle, int lane, int *front, int *rear) {
  const hwy_config *c = r->cfg;
  const hwy_lane *l = &c->net[lane];
  double s, lat;
  lane_local(l, vehicle->x, vehicle->y, &s, &lat);
  double s_front = 0, s_rear = 0;
  *front = *rear = -1;
  int search[1 + HWY_MAX_LANES], n_search = 0;
  double offset[1 + HWY_MAX_LANES];
  search[n_search] = lane; offset[n_search++] = 0;
  if (c->flags & HWY_C_CONNECTED_LANES) {
    if (l->next_first >= 0) {   /* at most one road leaves a node of the merge networks */
      search[n_search] = l->next_first + (l->id < l->next_lanes ? l->id : 0);
      offset[n_search++] = l->length;
    }
    for (int q = 0; q < c->net_lanes; q++) {   /* roads in table order, one visit each (their lane 0) */
      const hwy_lane *pl = &c->net[q];
      if (pl->id != 0 || pl->next_first != l->road_first) continue;
      const int k = q + (l->id < pl->road_lanes ? l->id : 0);
      search[n_search] = k; offset[n_search++] = -c->net[k].length;
    }
  }
  for (int j = 0; j < r->n; j++) {
    const ent_t *v = &r->v[j];
    if (!v->present || v == vehicle) continue;
    for (int k = 0; k < n_search; k++) {
      const hwy_lane *sl = &c->net[search[k]];
      double s_v, lat_v;
      lane_local(sl, v->x, v->y, &s_v, &lat_v);
      if (!lane_on_lane(sl, v->x, v->y, 1.0)) continue;
      s_v += offset[k];
      if (s <= s_v && (*front < 0 || s_v <= s_front)) { s_front = s_v; *front = j; }
      if (s_v < s && (*rear < 0 || s_v > s_rear)) { s_rear = s_v; *rear = j; }
      break;
    }
  }
}

/* ---- vehicle/behavior.py ------------------------------------------------------------------- */
static double desired_gap(const ent_t *ego, const ent_t *front) { /* behavior.py:192-217 */
  double d0 = DISTANCE_WANTED, tau = TIME_WANTED, ab = -COMFORT_ACC_MAX * COMFORT_ACC_MIN;
  double ce = cos(ego->heading), se = sin(ego->heading);
  double cf = cos(front->heading), sf = sin(front->heading);
  double dv = (ego->speed * ce - front->speed * cf) * ce + (ego->speed * se - front->speed * sf) * se;
  return d0 + ego->speed * tau + ego->speed * dv / (2 * sqrt(ab));
}
/* behavior.py:150-190; an Obstacle as ego_vehicle is "not isinstance(ego_vehicle, Vehicle)" => 0 */
static double idm_acceleration(const net_t *r, const ent_t *self, const ent_t *ego, const ent_t *front) {
  if (!ego || ego->obstacle) return 0;
  double ego_target_speed = clipd(ego->target_speed, 0, r->cfg->net[ego->lane].speed_limit);
  double acceleration =
      COMFORT_ACC_MAX * (1 - pow(fmax(ego->speed, 0) / fabs(not_zero(ego_target_speed)), self->delta));
  if (front) {
    double d = lane_distance_to(r->cfg, ego, front);
    double q = desired_gap(ego, front) / not_zero(d);
    acceleration -= COMFORT_ACC_MAX * (q * q);
  }
  return acceleration;
}
static const ent_t *vp(const net_t *r, int idx) { return idx < 0 ? NULL : &r->v[idx]; }

static int mobil(const net_t *r, const ent_t *self, int lane) { /* behavior.py:265-324, route None */
  int np_, nf_, op_, of_;
  neighbour_vehicles(r, self, lane, &np_, &nf_);
  const ent_t *new_preceding = vp(r, np_), *new_following = vp(r, nf_);
  double new_following_a = idm_acceleration(r, self, new_following, new_preceding);
  double new_following_pred_a = idm_acceleration(r, self, new_following, self);
  if (new_following_pred_a < -LANE_CHANGE_MAX_BRAKING_IMPOSED) return 0;
  neighbour_vehicles(r, self, self->lane, &op_, &of_);
  const ent_t *old_preceding = vp(r, op_), *old_following = vp(r, of_);
  double self_pred_a = idm_acceleration(r, self, self, new_preceding);
  double self_a = idm_acceleration(r, self, self, old_preceding);
  double old_following_a = idm_acceleration(r, self, old_following, self);
  double old_following_pred_a = idm_acceleration(r, self, old_following, old_preceding);
  double jerk = self_pred_a - self_a +
                POLITENESS * (new_following_pred_a - new_following_a + old_following_pred_a - old_following_a);
  if (jerk < LANE_CHANGE_MIN_ACC_GAIN) return 0;
  return 1;
}
/* controller.py:135-143 */
static void follow_road(const hwy_config *c, ent_t *self) {
  if (lane_after_end(&c->net[self->target_lane], self->x, self->y))
    self->target_lane = next_lane(c, self->target_lane, self->x, self->y);
}
/* behavior.py:219-263 */
static void change_lane_policy(net_t *r, ent_t *self) {
  const hwy_config *c = r->cfg;
  if (self->lane != self->target_lane) {
    if (c->net[self->lane].road == c->net[self->target_lane].road) { /* lane_index[:2] == target_lane_index[:2] */
      for (int j = 0; j < r->n; j++) {
        const ent_t *v = &r->v[j];
        if (!is_veh(v)) continue; /* for v in self.road.vehicles */
        if (v != self && v->lane != self->target_lane && v->target_lane == self->target_lane) {
          double d = lane_distance_to(c, self, v);
          double d_star = desired_gap(self, v);
          if (0 < d && d < d_star) {
            self->target_lane = self->lane;
            break;
          }
        }
      }
    }
    return;
  }
  if (!(LANE_CHANGE_DELAY < self->timer)) return;
  self->timer = 0;
  const hwy_lane *l = &c->net[self->lane];
  for (int side = 0; side < 2; side++) { /* side_lanes (road.py:200-211): id-1 then id+1 on the same road */
    int id = side == 0 ? l->id - 1 : l->id + 1;
    if (id < 0 || id >= l->road_lanes) continue;
    int lane = l->road_first + id;
    if (!lane_is_reachable_from(&c->net[lane], self->x, self->y)) continue;
    if (fabs(self->speed) < 1) continue;
    if (mobil(r, self, lane)) self->target_lane = lane;
  }
}

/* ---- vehicle/controller.py -------------------------------------------------------------------- */
static double steering_control(const hwy_config *c, const ent_t *self, int target_lane) { /* controller.py:145-187 */
  const hwy_lane *l = &c->net[target_lane];
  double s, lat;
  lane_local(l, self->x, self->y, &s, &lat);
  double lane_next_coords = s + self->speed * TAU_PURSUIT;
  double lane_future_heading = lane_heading_at(l, lane_next_coords);
  double lateral_speed_command = -KP_LATERAL * lat;
  double heading_command = asin(clipd(lateral_speed_command / not_zero(self->speed), -1, 1));
  double heading_ref = lane_future_heading + clipd(heading_command, -M_PI / 4, M_PI / 4);
  double heading_rate_command = KP_HEADING * wrap_to_pi(heading_ref - self->heading);
  double slip_angle = asin(clipd(VEH_LENGTH / 2 / not_zero(self->speed) * heading_rate_command, -1, 1));
  double steering_angle = atan(2 * tan(slip_angle));
  return clipd(steering_angle, -MAX_STEERING_ANGLE, MAX_STEERING_ANGLE);
}
static int speed_to_index(const hwy_config *c, double speed) { /* controller.py:326-344 */
  int n = c->num_target_speeds;
  double x = (speed - c->target_speeds[0]) / (c->target_speeds[n - 1] - c->target_speeds[0]);
  return (int)clipd(rint(x * (n - 1)), 0, n - 1);
}
/* controller.py:89-133 */
static void controlled_act(const hwy_config *c, ent_t *self, int action) {
  follow_road(c, self);
  if (action == HWY_LANE_RIGHT || action == HWY_LANE_LEFT) {
    const hwy_lane *t = &c->net[self->target_lane];
    int id = t->id + (action == HWY_LANE_RIGHT ? 1 : -1);
    if (id < 0) id = 0;
    if (id > t->road_lanes - 1) id = t->road_lanes - 1;
    int lane = t->road_first + id;
    if (lane_is_reachable_from(&c->net[lane], self->x, self->y)) self->target_lane = lane;
  }
  double steering = steering_control(c, self, self->target_lane);
  self->act_accel = KP_A * (self->target_speed - self->speed);
  self->act_steering = clipd(steering, -MAX_STEERING_ANGLE, MAX_STEERING_ANGLE);
}
/* controller.py:295-315; action < 0 == None */
static void mdp_act(const hwy_config *c, ent_t *self, int action) {
  if (action == HWY_FASTER || action == HWY_SLOWER) {
    int idx = speed_to_index(c, self->speed) + (action == HWY_FASTER ? 1 : -1);
    if (idx < 0) idx = 0;
    if (idx > c->num_target_speeds - 1) idx = c->num_target_speeds - 1;
    self->speed_index = idx;
    self->target_speed = c->target_speeds[idx];
    controlled_act(c, self, -1);
  } else {
    controlled_act(c, self, action);
  }
}
/* behavior.py:93-137 */
static void idm_act(net_t *r, ent_t *self) {
  if (self->crashed) return;
  follow_road(r->cfg, self);
  change_lane_policy(r, self);
  double steering = steering_control(r->cfg, self, self->target_lane);
  steering = clipd(steering, -MAX_STEERING_ANGLE, MAX_STEERING_ANGLE);
  int f, b;
  neighbour_vehicles(r, self, self->lane, &f, &b);
  double acc = idm_acceleration(r, self, self, vp(r, f));
  if (self->lane != self->target_lane) {
    neighbour_vehicles(r, self, self->target_lane, &f, &b);
    double tacc = idm_acceleration(r, self, self, vp(r, f));
    acc = fmin(acc, tacc);
  }
  acc = clipd(acc, -ACC_MAX, ACC_MAX);
  self->act_steering = steering;
  self->act_accel = acc;
}

/* ---- vehicle/kinematics.py --------------------------------------------------------------------- */
static void clip_actions(ent_t *v) { /* kinematics.py:155-168 */
  if (v->crashed) {
    v->act_steering = 0;
    v->act_accel = -1.0 * v->speed;
  }
  if (v->speed > MAX_SPEED) v->act_accel = fmin(v->act_accel, 1.0 * (MAX_SPEED - v->speed));
  else if (v->speed < MIN_SPEED) v->act_accel = fmax(v->act_accel, 1.0 * (MIN_SPEED - v->speed));
}
static void vehicle_step(const hwy_config *c, ent_t *v, double dt) { /* kinematics.py:130-153, behavior.py:139-148 */
  if (!v->controlled) v->timer += dt;
  clip_actions(v);
  double delta_f = v->act_steering;
  double beta = atan(1.0 / 2 * tan(delta_f));
  double vx = v->speed * cos(v->heading + beta), vy = v->speed * sin(v->heading + beta);
  v->x += vx * dt;
  v->y += vy * dt;
  if (v->has_impact) {
    v->x += v->impact_x;
    v->y += v->impact_y;
    v->crashed = 1;
    v->has_impact = 0;
    v->impact_x = v->impact_y = 0;
  }
  v->heading += v->speed * sin(beta) / (VEH_LENGTH / 2) * dt;
  v->speed += v->act_accel * dt;
  v->lane = closest_lane_index(c, v->x, v->y, v->heading);
}

/* ---- road/road.py:464-481 -------------------------------------------------------------------------- */
static void apply_meta_actions(net_t *r, const int *actions) { /* abstract.py:294-304, action.py:259-260,320-325 */
  for (int a = 0; a < r->cfg->num_agents; a++) /* the id indexes ACTIONS_ALL / _LONGI / _LAT */
    mdp_act(r->cfg, &r->v[r->cfg->agent_index[a]], HWY_ACTION_TO_ALL(r->cfg->action_set, actions[a]));
}
static void road_act(net_t *r) {
  for (int i = 0; i < r->n; i++) {
    ent_t *v = &r->v[i];
    if (!is_veh(v)) continue;
    if (v->controlled) mdp_act(r->cfg, v, -1);
    else idm_act(r, v);
  }
}
static void road_step(net_t *r, double dt) {
  for (int i = 0; i < r->n; i++)
    if (is_veh(&r->v[i])) vehicle_step(r->cfg, &r->v[i], dt);
  for (int i = 0; i < r->n; i++) {
    if (!is_veh(&r->v[i])) continue;
    for (int j = i + 1; j < r->n; j++)
      if (is_veh(&r->v[j])) handle_collisions(&r->v[i], &r->v[j], dt);
    for (int j = 0; j < r->n; j++)
      if (r->v[j].present && r->v[j].obstacle) handle_collisions(&r->v[i], &r->v[j], dt);
  }
}

/* ---- observation -------------------------------------------------------------------------------------- */
/* Vehicle.to_dict (kinematics.py:237-261) / RoadObject.to_dict (objects.py:141-160): only the keys both have */
static double feature_of(const ent_t *v, int fid) {
  switch (fid) {
    case HWY_FEAT_PRESENCE: return 1;
    case HWY_FEAT_X: return v->x;
    case HWY_FEAT_Y: return v->y;
    case HWY_FEAT_VX: return v->obstacle ? 0.0 : v->speed * cos(v->heading);
    case HWY_FEAT_VY: return v->obstacle ? 0.0 : v->speed * sin(v->heading);
    case HWY_FEAT_COS_H: return cos(v->heading);
    case HWY_FEAT_SIN_H: return sin(v->heading);
    case HWY_FEAT_COS_D: return 0; /* no route => destination == position => zeros */
    case HWY_FEAT_SIN_D: return 0;
  }
  return 0;
}
typedef struct { double key; int idx; } close_t;
/* KinematicObservation.observe (observation.py:234-276) + Road.close_objects_to (road.py:421-450), include_obstacles */
static void observe_agent(const net_t *r, int ego_idx, float *obs) {
  const hwy_config *c = r->cfg;
  const ent_t *ego = &r->v[ego_idx];
  int V = c->obs_vehicles, F = c->obs_features;
  int see_behind = (c->flags & HWY_C_OBS_SEE_BEHIND) != 0;
  close_t *close = (close_t *)malloc(sizeof(close_t) * (size_t)r->n);
  int m = 0;
  /* vehicles_only = not include_obstacles (observation.py:246, road.py:444) */
  for (int pass = 0; pass < ((c->flags & HWY_C_OBS_VEHICLES_ONLY) ? 1 : 2); pass++) { /* vehicles, then obstacles */
    for (int j = 0; j < r->n; j++) {
      const ent_t *v = &r->v[j];
      if (!v->present || v->obstacle != pass) continue;
      double dx = v->x - ego->x, dy = v->y - ego->y;
      if (!(sqrt(dx * dx + dy * dy) < c->perception_distance)) continue;
      if (pass == 0) {
        if (v == ego) continue;
        if (!(see_behind || -2 * VEH_LENGTH < lane_distance_to(c, ego, v))) continue;
      } else {
        if (!(-2 * VEH_LENGTH < lane_distance_to(c, ego, v))) continue;
      }
      close[m].key = fabs(lane_distance_to(c, ego, v));
      close[m].idx = j;
      m++;
    }
  }
  for (int a = 1; a < m && !(c->flags & HWY_C_OBS_UNSORTED); a++) { /* sorted() is stable; sort=False: list order */
    close_t t = close[a];
    int b = a - 1;
    while (b >= 0 && close[b].key > t.key) { close[b + 1] = close[b]; b--; }
    close[b + 1] = t;
  }
  if (m > V - 1) m = V - 1;
  for (int row = 0; row < V; row++) {
    for (int f = 0; f < F; f++) {
      int fid = c->obs_feature_ids[f];
      double val = 0;
      if (row <= m) {
        const ent_t *v = row == 0 ? ego : &r->v[close[row - 1].idx];
        val = feature_of(v, fid);
        if (row > 0 && !(c->flags & HWY_C_OBS_ABSOLUTE) &&
            (fid == HWY_FEAT_X || fid == HWY_FEAT_Y || fid == HWY_FEAT_VX || fid == HWY_FEAT_VY))
          val -= feature_of(ego, fid);
        if (c->flags & HWY_C_OBS_NORMALIZE) {
          const double *rg = fid == HWY_FEAT_X ? c->obs_range_x : fid == HWY_FEAT_Y ? c->obs_range_y
                           : fid == HWY_FEAT_VX ? c->obs_range_vx : fid == HWY_FEAT_VY ? c->obs_range_vy : NULL;
          if (rg && isfinite(rg[0])) {
            val = lmap(val, rg[0], rg[1], -1, 1);
            if (c->flags & HWY_C_OBS_CLIP) val = clipd(val, -1, 1);
          }
        }
      }
      obs[row * F + f] = (float)val;
    }
  }
  free(close);
}

/* ---- OccupancyGridObservation.observe (observation.py:354-413) on a road network: only Road.vehicles are rasterised
 *      (the Obstacle of Road.objects is not, :366-368), the on-road layer walks every lane of the network (:454-484) -- */
static void grid_pos_to_index(const hwy_config *c, const ent_t *ego, double px, double py, int relative, int *ci, int *cj) {
  if (!relative) { px -= ego->x; py -= ego->y; } /* observation.py:415-435 */
  if (c->flags & HWY_C_GRID_ALIGN) {
    double cs = cos(ego->heading), sn = sin(ego->heading);
    double qx = cs * px + sn * py, qy = -sn * px + cs * py;
    px = qx; py = qy;
  }
  *ci = (int)floor((px - c->grid_min[0]) / c->grid_step[0]);
  *cj = (int)floor((py - c->grid_min[1]) / c->grid_step[1]);
}
static const double *grid_range(const hwy_config *c, int fid) {
  const double *rg = fid == HWY_FEAT_X ? c->obs_range_x : fid == HWY_FEAT_Y ? c->obs_range_y
                   : fid == HWY_FEAT_VX ? c->obs_range_vx : fid == HWY_FEAT_VY ? c->obs_range_vy : NULL;
  return (rg && isfinite(rg[0])) ? rg : NULL;
}
static void observe_grid_agent(const net_t *r, int ego_idx, float *obs) {
  const hwy_config *c = r->cfg;
  const ent_t *ego = &r->v[ego_idx];
  int F = c->obs_features, W = c->grid_shape[0], H = c->grid_shape[1];
  double *grid = (double *)malloc(sizeof(double) * (size_t)F * W * H);
  for (int k = 0; k < F * W * H; k++) grid[k] = NAN;
  for (int layer = 0; layer < F; layer++) {
    int fid = c->obs_feature_ids[layer];
    if (fid != HWY_FEAT_ON_ROAD) {
      for (int i = r->n - 1; i >= 0; i--) { /* df[::-1].iterrows() over Road.vehicles: the lower index wins a cell */
        const ent_t *v = &r->v[i];
        if (!is_veh(v)) continue;
        double x = v->x - ego->x, y = v->y - ego->y;
        const double *rx = grid_range(c, HWY_FEAT_X), *ry = grid_range(c, HWY_FEAT_Y);
        if (rx) { x = lmap(x, rx[0], rx[1], -1, 1); x = lmap(x, -1, 1, rx[0], rx[1]); }
        if (ry) { y = lmap(y, ry[0], ry[1], -1, 1); y = lmap(y, -1, 1, ry[0], ry[1]); }
        int ci, cj;
        grid_pos_to_index(c, ego, x, y, 1, &ci, &cj);
        if (0 <= ci && ci < W && 0 <= cj && cj < H) {
          double val = feature_of(v, fid);
          if (fid == HWY_FEAT_X || fid == HWY_FEAT_Y || fid == HWY_FEAT_VX || fid == HWY_FEAT_VY) val -= feature_of(ego, fid);
          const double *rg = grid_range(c, fid);
          if (rg) val = lmap(val, rg[0], rg[1], -1, 1);
          grid[((size_t)layer * W + ci) * H + cj] = val;
        }
      }
    } else {
      double spacing = fmin(c->grid_step[0], c->grid_step[1]);
      for (int k = 0; k < c->net_lanes; k++) { /* every lane of the network, StraightLane and SineLane alike */
        const hwy_lane *l = &c->net[k];
        double origin, lat;
        lane_local(l, ego->x, ego->y, &origin, &lat);
        double start = origin - 100.0, stop = origin + 100.0;
        int n = (int)ceil((stop - start) / spacing); /* len(np.arange(start, stop, step)) */
        for (int j = 0; j < n; j++) {
          double wp = clipd(start + j * spacing, 0, l->length);
          double px, py;
          lane_position(l, wp, 0.0, &px, &py);
          int ci, cj;
          grid_pos_to_index(c, ego, px, py, 0, &ci, &cj);
          if (0 <= ci && ci < W && 0 <= cj && cj < H) grid[((size_t)layer * W + ci) * H + cj] = 1;
        }
      }
    }
  }
  for (int k = 0; k < F * W * H; k++) {
    double v = grid[k];
    if (c->flags & HWY_C_OBS_CLIP) v = isnan(v) ? v : clipd(v, -1, 1);
    /* as_image (observation.py:408-409): ((clip(obs, -1, 1) + 1) / 2 * 255).astype(uint8); an empty (NaN) cell casts to 0 */
    if (c->flags & HWY_C_GRID_IMAGE) v = isnan(v) ? 0.0 : (double)(uint8_t)((clipd(v, -1, 1) + 1) / 2 * 255);
    obs[k] = isnan(v) ? 0.0f : (float)v;
  }
  free(grid);
}
static void observe_any(const net_t *r, int ego_idx, float *obs) {
  if (r->cfg->obs_type == HWY_OBS_OCCUPANCY_GRID) observe_grid_agent(r, ego_idx, obs);
  else observe_agent(r, ego_idx, obs);
}

/* ---- MergeEnv._reward / _rewards (merge_env.py:40-75); `action` is the agent's own meta-action ---------- */
static double reward_of(const net_t *r, const ent_t *ego, int action) {
  const hwy_config *c = r->cfg;
  double scaled_speed = lmap(ego->speed, c->reward_speed_range[0], c->reward_speed_range[1], 0, 1);
  double merging = 0; /* sum() over Road.vehicles on ("b","c",lanes) */
  for (int j = 0; j < r->n; j++) {
    const ent_t *v = &r->v[j];
    if (is_veh(v) && v->lane == c->merge_lane) merging = merging + (v->target_speed - v->speed) / v->target_speed;
  }
  double reward = 0;
  reward = reward + c->collision_reward * (double)ego->crashed;
  reward = reward + c->right_lane_reward * ((double)c->net[ego->lane].id / 1);
  reward = reward + c->high_speed_reward * scaled_speed;
  reward = reward + c->lane_change_reward * (double)(action == 0 || action == 2);
  reward = reward + c->merging_speed_reward * merging;
  return lmap(reward, c->collision_reward + c->merging_speed_reward, c->high_speed_reward + c->right_lane_reward, 0, 1);
}

/* ---- SoA <-> AoS -------------------------------------------------------------------------------------------- */
static void load_env(const hwy_config *c, const hwy_state *st, int e, ent_t *v) {
  int N = c->num_vehicles;
  for (int i = 0; i < N; i++) {
    size_t k = (size_t)e * N + i;
    ent_t *o = &v[i];
    memset(o, 0, sizeof(*o));
    o->x = st->x[k]; o->y = st->y[k]; o->heading = st->heading[k]; o->speed = st->speed[k];
    o->timer = st->timer[k]; o->target_speed = st->target_speed[k]; o->delta = st->delta[k];
    o->impact_x = st->impact_x[k]; o->impact_y = st->impact_y[k];
    o->lane = st->lane[k]; o->target_lane = st->target_lane[k]; o->speed_index = st->speed_index[k];
    int f = st->flags[k];
    o->crashed = !!(f & HWY_F_CRASHED); o->has_impact = !!(f & HWY_F_HAS_IMPACT);
    o->check_collisions = !!(f & HWY_F_CHECK_COLLISIONS); o->controlled = !!(f & HWY_F_CONTROLLED);
    o->obstacle = !!(f & HWY_F_OBSTACLE); o->present = !(f & HWY_F_ABSENT);
    o->impact_margin = o->flag_margin = INFINITY;
  }
}
static void store_env(const hwy_config *c, hwy_state *st, int e, const ent_t *v) {
  int N = c->num_vehicles;
  for (int i = 0; i < N; i++) {
    size_t k = (size_t)e * N + i;
    const ent_t *o = &v[i];
    st->x[k] = o->x; st->y[k] = o->y; st->heading[k] = o->heading; st->speed[k] = o->speed;
    st->timer[k] = o->timer; st->target_speed[k] = o->target_speed; st->delta[k] = o->delta;
    st->impact_x[k] = o->impact_x; st->impact_y[k] = o->impact_y;
    st->lane[k] = o->lane; st->target_lane[k] = o->target_lane; st->speed_index[k] = o->speed_index;
    st->flags[k] = (o->crashed ? HWY_F_CRASHED : 0) | (o->has_impact ? HWY_F_HAS_IMPACT : 0) |
                   (o->check_collisions ? HWY_F_CHECK_COLLISIONS : 0) | (o->controlled ? HWY_F_CONTROLLED : 0) |
                   (o->obstacle ? HWY_F_OBSTACLE : 0) | (o->present ? 0 : HWY_F_ABSENT);
    if (orc_margin_buf) orc_margin_buf[k] = o->impact_margin;
    if (orc_flag_margin_buf) orc_flag_margin_buf[k] = o->flag_margin;
  }
}

/* ---- entry points used by hwy_oracle.c's orc_* when cfg->scenario != HWY_SCENARIO_HIGHWAY ------------------ */
int orc_net_frames(const hwy_config *c, hwy_state *st, const int32_t *actions, int32_t n_frames) {
  int N = c->num_vehicles, A = c->num_agents;
  ent_t *v = (ent_t *)malloc(sizeof(ent_t) * (size_t)N);
  int acts[HWY_MAX_AGENTS];
  for (int e = 0; e < c->num_envs; e++) {
    load_env(c, st, e, v);
    net_t r = {c, v, N};
    for (int fr = 0; fr < n_frames; fr++) {
      if (fr == 0 && actions) {
        for (int a = 0; a < A; a++) acts[a] = actions[e * A + a];
        apply_meta_actions(&r, acts);
      }
      road_act(&r);
      road_step(&r, c->dt);
    }
    store_env(c, st, e, v);
  }
  free(v);
  return 0;
}

/* Road.neighbour_vehicles(vehicle = slot `slot` of environment e, lane_index = table index `lane`) -> slot indices of the
 * preceding / following vehicle or -1 (the checker of tests/test_oracle_reference_neighbours.py, which restates the
 * reference's own tests/road/test_neighbour_vehicles.py) */
int orc_net_neighbours(const hwy_config *c, const hwy_state *st, int32_t e, int32_t slot, int32_t lane, int32_t *front,
                       int32_t *rear) {
  int N = c->num_vehicles;
  if (e < 0 || e >= c->num_envs || slot < 0 || slot >= N || lane >= c->net_lanes) return HWY_ERR_INVALID_ARG;
  ent_t *v = (ent_t *)malloc(sizeof(ent_t) * (size_t)N);
  load_env(c, st, e, v);
  net_t r = {c, v, N};
  int f, b;
  /* road.py:499-501: lane_index = lane_index or vehicle.lane_index; if not lane_index: return None, None
   * (lane < 0 == None; a vehicle without a lane index carries lane < 0) */
  if (lane < 0) lane = v[slot].lane;
  if (lane < 0 || lane >= c->net_lanes) {
    *front = *rear = -1;
    free(v);
    return 0;
  }
  neighbour_vehicles(&r, &v[slot], lane, &f, &b);
  *front = f; *rear = b;
  free(v);
  return 0;
}

int orc_net_observe(const hwy_config *c, const hwy_state *st, float *obs) {
  int N = c->num_vehicles, A = c->num_agents;
  size_t VF = c->obs_type == HWY_OBS_OCCUPANCY_GRID ? (size_t)c->obs_features * c->grid_shape[0] * c->grid_shape[1]
                                                    : (size_t)c->obs_vehicles * c->obs_features;
  ent_t *v = (ent_t *)malloc(sizeof(ent_t) * (size_t)N);
  for (int e = 0; e < c->num_envs; e++) {
    load_env(c, st, e, v);
    net_t r = {c, v, N};
    for (int a = 0; a < A; a++) observe_any(&r, c->agent_index[a], obs + ((size_t)e * A + a) * VF);
  }
  free(v);
  return 0;
}

/* AbstractEnv.step (abstract.py:259-285) with MergeEnv's reward / termination */
int orc_net_step(const hwy_config *c, hwy_state *st, const int32_t *actions, float *obs, double *reward,
                 uint8_t *terminated, uint8_t *truncated, double *info_speed, uint8_t *info_crashed) {
  int N = c->num_vehicles, A = c->num_agents;
  size_t VF = c->obs_type == HWY_OBS_OCCUPANCY_GRID ? (size_t)c->obs_features * c->grid_shape[0] * c->grid_shape[1]
                                                    : (size_t)c->obs_vehicles * c->obs_features;
  ent_t *v = (ent_t *)malloc(sizeof(ent_t) * (size_t)N);
  int acts[HWY_MAX_AGENTS];
  for (int e = 0; e < c->num_envs; e++) {
    for (int a = 0; a < A; a++) {
      acts[a] = actions[e * A + a];
      if (acts[a] < 0 || acts[a] >= HWY_NUM_ACTIONS(c->action_set)) { free(v); return HWY_ERR_ACTION; }
    }
    load_env(c, st, e, v);
    net_t r = {c, v, N};
    st->time[e] += c->policy_dt;
    for (int fr = 0; fr < c->frames_per_step; fr++) {
      if (fr == 0) apply_meta_actions(&r, acts);
      road_act(&r);
      road_step(&r, c->dt);
    }
    for (int a = 0; a < A; a++) {
      const ent_t *ego = &v[c->agent_index[a]];
      observe_any(&r, c->agent_index[a], obs + ((size_t)e * A + a) * VF);
      reward[e * A + a] = reward_of(&r, ego, acts[a]);
      if (info_speed) info_speed[e * A + a] = ego->speed;
      if (info_crashed) info_crashed[e * A + a] = (uint8_t)ego->crashed;
    }
    const ent_t *ego = &v[c->agent_index[0]]; /* self.vehicle == controlled_vehicles[0] */
    terminated[e] = (uint8_t)(ego->crashed || ego->x > c->merge_end_x); /* merge_env.py:77-79, :365-369 */
    truncated[e] = (uint8_t)(st->time[e] >= c->duration);               /* never: duration == inf */
    store_env(c, st, e, v);
  }
  free(v);
  return 0;
}

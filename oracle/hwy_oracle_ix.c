/*
 * hwy_oracle_ix.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's Road.act() / RegulatedRoad.step()
 * hot path on the 4-way INTERSECTION network (IntersectionEnv, highway_env/envs/intersection_env.py), in plain
 * sequential C (glibc libm, f64).  Companion of hwy_oracle.c / hwy_oracle_net.c; the same rules apply: only tests/
 * may load it, as the checker for the next hot-path row; the product never touches it.
 *
 * Parity pin: tests/golden/intersection_*.npz recorded from the unmodified Python reference by
 * tests/golden/make_golden_intersection.py (per simulation frame and per policy step),
 * tests/test_oracle_golden_intersection.py.
 *
 * What is new relative to the merge networks:
 *   lanes of any direction and CircularLane (road/lane.py:159-213, 311-369), with priorities; planned routes
 *   (ControlledVehicle.plan_route_to, controller.py:71-87; RoadNetwork.next_lane with a route, road.py:73-146;
 *   position_heading_along_route, road.py:323-362); RegulatedRoad (road/regulation.py): every int(1/dt/2) frames
 *   yielding vehicles are released and every pair of vehicles is tested for a conflict on their constant-speed
 *   predicted trajectories (11 samples, rotated-rectangle test, utils.py:99-174), the lower-priority one is stopped;
 *   a vehicle list that shrinks and grows between policy steps (IntersectionEnv._clear_vehicles / _spawn_vehicle,
 *   intersection_env.py:292-338); IntersectionEnv's IDM parameters (:243-247), 3-action DiscreteMetaAction
 *   (longitudinal only), reward and termination (:60-126).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/highway_env/).
 * The structs below are private to the oracle (the product ABI for this row does not exist yet).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VEH_LENGTH 5.0
#define VEH_WIDTH 2.0
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
#define LANE_CHANGE_DELAY 1.0
#define LANE_VEHICLE_LENGTH 5.0
#define IX_MAX_LANES 32
#define IX_MAX_ROUTE 8 /* the multi-agent default sends its second car round another arm: 6 roads */
#define IX_MAX_AGENTS 4
#define IX_MAX_FEATURES 16

enum { FEAT_PRESENCE = 0, FEAT_X, FEAT_Y, FEAT_VX, FEAT_VY, FEAT_HEADING, FEAT_COS_H, FEAT_SIN_H, FEAT_COS_D, FEAT_SIN_D, FEAT_LONG_OFF, FEAT_LAT_OFF, FEAT_ANG_OFF, FEAT_ON_ROAD = 13 };
enum { ACT_SLOWER = 0, ACT_IDLE = 1, ACT_FASTER = 2 }; /* IntersectionEnv.ACTIONS, intersection_env.py:14 */

typedef struct {
  int32_t kind; /* 0 StraightLane, 1 CircularLane */
  int32_t direction, priority, forbidden, from_node, to_node, id;
  int32_t exit_lane; /* "il" in lane_index[0] and "o" in lane_index[1] (intersection_env.py:328-331, 341-344) */
  double sx, sy, ex, ey, heading, dirx, diry; /* straight */
  double cx, cy, radius, start_phase, end_phase; /* circular */
  double length, width, speed_limit;
} ix_lane;

typedef struct {
  int32_t num_envs, n_slots, n_lanes, n_route, frames_per_step, num_target_speeds, obs_vehicles, obs_features;
  int32_t obs_feature_ids[IX_MAX_FEATURES];
  int32_t obs_absolute, obs_normalize, obs_clip, obs_see_behind, normalize_reward, offroad_terminal,
      connected_lanes /* Road.neighbour_vehicles_connected_lanes (intersection-v2) */,
      obs_unsorted /* order="shuffled": close_objects_to(sort=False) (observation.py:245) */;
  double dt, policy_dt, duration, perception_distance;
  double distance_wanted, time_wanted, comfort_acc_max, comfort_acc_min; /* IDMVehicle class attributes (:243-247) */
  double target_speeds[8];
  double collision_reward, high_speed_reward, arrived_reward, reward_speed_range[2];
  double obs_range_x[2], obs_range_y[2], obs_range_vx[2], obs_range_vy[2];
  double spawn_probability;
  int32_t access_lane[4]; /* table index of ("o" + k, "ir" + k, 0) */
  int32_t outer_node[4];  /* node id of "o" + k */
  int32_t obs_type, grid_align, grid_shape[2]; /* obs_type 1: OccupancyGridObservation (observation.py:279-499) */
  double grid_min[2], grid_step[2];
  int32_t num_agents;      /* controlled_vehicles: MultiAgentIntersectionEnv (intersection_env.py:348-399) */
  int32_t obs_intentions;  /* KinematicObservation.observe_intentions (observation.py:171,253) */
  int32_t grid_image, pad_; /* OccupancyGridObservation.as_image (observation.py:296,408-409) */
  ix_lane lanes[IX_MAX_LANES];
} ix_config;

typedef struct {
  double *x, *y, *heading, *speed, *timer, *target_speed, *delta, *impact_x, *impact_y; /* [E][C] */
  int32_t *present, *lane, *target_lane, *speed_index, *crashed, *has_impact, *controlled, *is_yielding, *yield_timer,
      *route_len;                           /* [E][C] */
  int32_t *route_from, *route_to, *route_id; /* [E][C][R]; id -1 == None */
  int32_t *road_steps;                      /* [E] RegulatedRoad.steps */
  double *time;                             /* [E] */
} ix_state;

typedef struct {
  double x, y, heading, speed, timer, target_speed, delta, impact_x, impact_y, act_steering, act_accel;
  int lane, target_lane, speed_index, crashed, has_impact, controlled, is_yielding, yield_timer, route_len;
  int route_from[IX_MAX_ROUTE], route_to[IX_MAX_ROUTE], route_id[IX_MAX_ROUTE];
  double impact_margin, flag_margin; /* test diagnostics only (hwy_oracle.c: orc_set_margin_buffer, orc_set_flag_margin_buffer) */
} veh_t;

typedef struct {
  const ix_config *cfg;
  veh_t *v; /* Road.vehicles, compact, in list order */
  int n;
  int steps;
} road_t;

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
/* StraightLane.local_coordinates (lane.py:209-213); CircularLane.local_coordinates (lane.py:355-362) */
static void lane_local(const ix_lane *l, double x, double y, double *s, double *lat) {
  if (l->kind == 0) {
    double dx = x - l->sx, dy = y - l->sy;
    *s = dx * l->dirx + dy * l->diry;
    *lat = dx * -l->diry + dy * l->dirx; /* direction_lateral = [-direction[1], direction[0]] */
  } else {
    double dx = x - l->cx, dy = y - l->cy;
    double phi = atan2(dy, dx);
    phi = l->start_phase + wrap_to_pi(phi - l->start_phase);
    double r = sqrt(dx * dx + dy * dy);
    *s = l->direction * (phi - l->start_phase) * l->radius;
    *lat = l->direction * (l->radius - r);
  }
}
/* StraightLane.heading_at (lane.py:203-204); CircularLane.heading_at (lane.py:347-350) */
static double lane_heading_at(const ix_lane *l, double s) {
  if (l->kind == 0) return l->heading;
  double phi = l->direction * s / l->radius + l->start_phase;
  return phi + M_PI / 2 * l->direction;
}
/* StraightLane.position (lane.py:196-201); CircularLane.position (lane.py:341-345) */
static void lane_position(const ix_lane *l, double s, double lat, double *px, double *py) {
  if (l->kind == 0) {
    *px = l->sx + s * l->dirx + lat * -l->diry;
    *py = l->sy + s * l->diry + lat * l->dirx;
  } else {
    double phi = l->direction * s / l->radius + l->start_phase;
    *px = l->cx + (l->radius - lat * l->direction) * cos(phi);
    *py = l->cy + (l->radius - lat * l->direction) * sin(phi);
  }
}
/* AbstractLane.on_lane (lane.py:80-102) */
static int lane_on_lane(const ix_lane *l, double x, double y, double margin) {
  double s, lat;
  lane_local(l, x, y, &s, &lat);
  return fabs(lat) <= l->width / 2 + margin && -LANE_VEHICLE_LENGTH <= s && s < l->length + LANE_VEHICLE_LENGTH;
}
/* AbstractLane.after_end (lane.py:120-125) */
static int lane_after_end(const ix_lane *l, double x, double y) {
  double s, lat;
  lane_local(l, x, y, &s, &lat);
  return s > l->length - LANE_VEHICLE_LENGTH / 2;
}
/* AbstractLane.distance (lane.py:127-130) */
static double lane_distance(const ix_lane *l, double x, double y) {
  double s, r;
  lane_local(l, x, y, &s, &r);
  return fabs(r) + fmax(s - l->length, 0) + fmax(0 - s, 0);
}
/* AbstractLane.distance_with_heading (lane.py:132-143), local_angle (lane.py:145-147) */
static double lane_distance_with_heading(const ix_lane *l, double x, double y, double heading) {
  double s, r;
  lane_local(l, x, y, &s, &r);
  double angle = fabs(wrap_to_pi(heading - lane_heading_at(l, s)));
  return fabs(r) + fmax(s - l->length, 0) + fmax(0 - s, 0) + 1.0 * angle;
}

/* ---- road/road.py: RoadNetwork ------------------------------------------------------------------- */
/* get_closest_lane_index (road.py:55-71): first minimum in graph iteration order == table order */
static int closest_lane_index(const ix_config *c, double x, double y, double heading) {
  int best = 0;
  double bd = 0;
  for (int k = 0; k < c->n_lanes; k++) {
    double d = lane_distance_with_heading(&c->lanes[k], x, y, heading);
    if (k == 0 || d < bd) { bd = d; best = k; }
  }
  return best;
}
/* table index of (from, to, id), -1 if absent; n_road_lanes = len(graph[from][to]) */
static int lane_index_of(const ix_config *c, int from, int to, int id) {
  for (int k = 0; k < c->n_lanes; k++)
    if (c->lanes[k].from_node == from && c->lanes[k].to_node == to && c->lanes[k].id == id) return k;
  return -1;
}
static int road_lanes(const ix_config *c, int from, int to) {
  int n = 0;
  for (int k = 0; k < c->n_lanes; k++)
    if (c->lanes[k].from_node == from && c->lanes[k].to_node == to) n++;
  return n;
}
/* next_lane_given_next_road (road.py:135-154) */
static int next_lane_given_next_road(const ix_config *c, int from, int to, int id, int next_to, int next_id, double px,
                                     double py, double *dist) {
  if (road_lanes(c, from, to) == road_lanes(c, to, next_to)) {
    if (next_id < 0) next_id = id;
  } else {
    int n = road_lanes(c, to, next_to), best = 0;
    double bd = 0;
    for (int l = 0; l < n; l++) {
      double d = lane_distance(&c->lanes[lane_index_of(c, to, next_to, l)], px, py);
      if (l == 0 || d < bd) { bd = d; best = l; }
    }
    next_id = best;
  }
  *dist = lane_distance(&c->lanes[lane_index_of(c, to, next_to, next_id)], px, py);
  return next_id;
}
/* next_lane (road.py:73-133): consumes the head of the route like the reference's route.pop(0) */
static int next_lane(const ix_config *c, int cur, veh_t *v) {
  const ix_lane *cl = &c->lanes[cur];
  int from = cl->from_node, to = cl->to_node, id = cl->id;
  int next_to = -1, next_id = -1;
  if (v->route_len > 0) {
    if (v->route_from[0] == from && v->route_to[0] == to) { /* finished the first step of the route: drop it */
      for (int k = 1; k < v->route_len; k++) {
        v->route_from[k - 1] = v->route_from[k];
        v->route_to[k - 1] = v->route_to[k];
        v->route_id[k - 1] = v->route_id[k];
      }
      v->route_len--;
    }
    if (v->route_len > 0 && v->route_from[0] == to) {
      next_to = v->route_to[0];
      next_id = v->route_id[0];
    }
  }
  double s, lat, px, py;
  lane_local(cl, v->x, v->y, &s, &lat);
  lane_position(cl, s, 0, &px, &py);
  if (next_to < 0) {
    /* closest lane among every road leaving `to`, in graph[_to] order == table order; min() keeps the first */
    int have = 0, best_to = -1, best_id = -1;
    double bd = 0;
    for (int k = 0; k < c->n_lanes; k++) {
      const ix_lane *l = &c->lanes[k];
      if (l->from_node != to || l->id != 0) continue; /* one visit per successor road */
      double d;
      int nid = next_lane_given_next_road(c, from, to, id, l->to_node, -1, px, py, &d);
      if (!have || d < bd) { have = 1; bd = d; best_to = l->to_node; best_id = nid; }
    }
    if (!have) return cur; /* KeyError -> current index */
    next_to = best_to;
    next_id = best_id;
  } else {
    double d;
    next_id = next_lane_given_next_road(c, from, to, id, next_to, next_id, px, py, &d);
  }
  return lane_index_of(c, to, next_to, next_id);
}
/* position_heading_along_route (road.py:323-362) with lateral 0 */
static void position_heading_along_route(const ix_config *c, const veh_t *v, double longitudinal, double *px, double *py,
                                         double *heading) {
  /* route = v.route or [v.lane_index] */
  int rf[IX_MAX_ROUTE + 1], rt[IX_MAX_ROUTE + 1], ri[IX_MAX_ROUTE + 1], n = v->route_len;
  const ix_lane *cur = &c->lanes[v->lane];
  if (n > 0) {
    for (int k = 0; k < n; k++) { rf[k] = v->route_from[k]; rt[k] = v->route_to[k]; ri[k] = v->route_id[k]; }
  } else {
    n = 1;
    rf[0] = cur->from_node; rt[0] = cur->to_node; ri[0] = cur->id;
  }
  int pos = 0;
  /* _get_route_head_with_id: an unknown lane id becomes the current one (0 if out of range) */
#define HEAD_LANE(p) lane_index_of(c, rf[p], rt[p], ri[p] >= 0 ? ri[p] : (cur->id < road_lanes(c, cur->from_node, cur->to_node) ? cur->id : 0))
  int li = HEAD_LANE(pos);
  while (n - pos > 1 && longitudinal > c->lanes[li].length) {
    longitudinal -= c->lanes[li].length;
    pos++;
    li = HEAD_LANE(pos);
  }
#undef HEAD_LANE
  lane_position(&c->lanes[li], longitudinal, 0, px, py);
  *heading = lane_heading_at(&c->lanes[li], longitudinal);
}

/* ---- vehicle/objects.py ----------------------------------------------------------------------- */
/* lane_distance_to (objects.py:183-198): along SELF's current lane */
static double lane_distance_to(const ix_config *c, const veh_t *self, const veh_t *other) {
  const ix_lane *l = &c->lanes[self->lane];
  double so, ss, lat;
  lane_local(l, other->x, other->y, &so, &lat);
  lane_local(l, self->x, self->y, &ss, &lat);
  return so - ss;
}
/* polygon (objects.py:169-181) */
static void polygon(const veh_t *v, double p[5][2]) {
  const double L = VEH_LENGTH, W = VEH_WIDTH;
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
      if (interval_distance(min_a, max_a, min_b, max_b) > 0) *intersecting = 0;
      g_flag_dn = fmin(g_flag_dn, fabs(interval_distance(min_a, max_a, min_b, max_b)));
      double vp = normal[0] * (da[0] - db[0]) + normal[1] * (da[1] - db[1]);
      if (vp < 0) min_a += vp; else max_a += vp;
      double distance = interval_distance(min_a, max_a, min_b, max_b);
      if (distance > 0) *will_intersect = 0;
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
/* handle_collisions + _is_colliding (objects.py:92-138): vehicles only, all collidable and solid */
static void handle_collisions(veh_t *self, veh_t *other, double dt) {
  const double diag = sqrt(VEH_LENGTH * VEH_LENGTH + VEH_WIDTH * VEH_WIDTH);
  double dx = other->x - self->x, dy = other->y - self->y;
  if (sqrt(dx * dx + dy * dy) > (diag + diag) / 2 + self->speed * dt) return;
  double pa[5][2], pb[5][2], t[2];
  int intersecting, will_intersect;
  polygon(self, pa);
  polygon(other, pb);
  double da[2] = {self->speed * cos(self->heading) * dt, self->speed * sin(self->heading) * dt};
  double db[2] = {other->speed * cos(other->heading) * dt, other->speed * sin(other->heading) * dt};
  are_polygons_intersecting(pa, pb, da, db, &intersecting, &will_intersect, t);
  self->flag_margin = fmin(self->flag_margin, g_flag_dn);
  other->flag_margin = fmin(other->flag_margin, g_flag_dn);
  if (will_intersect) {
    self->impact_x = t[0] / 2; self->impact_y = t[1] / 2; self->has_impact = 1;
    other->impact_x = -t[0] / 2; other->impact_y = -t[1] / 2; other->has_impact = 1;
    self->impact_margin = fmin(self->impact_margin, g_axis_dn);
    other->impact_margin = fmin(other->impact_margin, g_axis_dn);
  }
  if (intersecting) {
    self->crashed = 1;
    other->crashed = 1;
  }
}

/* ---- road/road.py:483-547.  With connected_lanes the search list is the lane, then lane `id` (else 0) of every road
 * leaving `_to` (offset +lane.length), then lane `id` (else 0) of every road arriving at `_from` (offset -prev.length),
 * both in graph order == table order; a vehicle counts on the FIRST list entry it is on. ---------------------------------- */
static void neighbour_vehicles(const road_t *r, const veh_t *vehicle, int lane, int *front, int *rear) {
  const ix_config *c = r->cfg;
  const ix_lane *l = &c->lanes[lane];
  double s, lat;
  lane_local(l, vehicle->x, vehicle->y, &s, &lat);
  double s_front = 0, s_rear = 0;
  *front = *rear = -1;
  int search[1 + 2 * IX_MAX_LANES], n_search = 0;
  double offset[1 + 2 * IX_MAX_LANES];
  search[n_search] = lane; offset[n_search++] = 0;
  if (c->connected_lanes) {
    for (int pass = 0; pass < 2; pass++) {
      for (int k = 0; k < c->n_lanes; k++) {   /* roads in table order; k = lane 0 of a road */
        const ix_lane *q = &c->lanes[k];
        if (q->id != 0) continue;
        if (pass == 0 ? q->from_node != l->to_node : q->to_node != l->from_node) continue;
        int n_road = 1;                          /* lanes of the road (from, to) are consecutive, ordered by id */
        while (k + n_road < c->n_lanes && c->lanes[k + n_road].from_node == q->from_node &&
               c->lanes[k + n_road].to_node == q->to_node && c->lanes[k + n_road].id == n_road)
          n_road++;
        const int pick = k + (l->id < n_road ? l->id : 0);
        search[n_search] = pick;
        offset[n_search++] = pass == 0 ? l->length : -c->lanes[pick].length;
      }
    }
  }
  for (int j = 0; j < r->n; j++) {
    const veh_t *v = &r->v[j];
    if (v == vehicle) continue;
    for (int k = 0; k < n_search; k++) {
      const ix_lane *sl = &c->lanes[search[k]];
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

static double desired_gap(const ix_config *c, const veh_t *ego, const veh_t *front) { /* behavior.py:192-217 */
  double d0 = c->distance_wanted, tau = c->time_wanted, ab = -c->comfort_acc_max * c->comfort_acc_min;
  double ce = cos(ego->heading), se = sin(ego->heading);
  double cf = cos(front->heading), sf = sin(front->heading);
  double dv = (ego->speed * ce - front->speed * cf) * ce + (ego->speed * se - front->speed * sf) * se;
  return d0 + ego->speed * tau + ego->speed * dv / (2 * sqrt(ab));
}
/* behavior.py:150-190 */
static double idm_acceleration(const road_t *r, const veh_t *self, const veh_t *ego, const veh_t *front) {
  const ix_config *c = r->cfg;
  if (!ego) return 0;
  double ego_target_speed = clipd(ego->target_speed, 0, c->lanes[ego->lane].speed_limit);
  double acceleration =
      c->comfort_acc_max * (1 - pow(fmax(ego->speed, 0) / fabs(not_zero(ego_target_speed)), self->delta));
  if (front) {
    double d = lane_distance_to(c, ego, front);
    double q = desired_gap(c, ego, front) / not_zero(d);
    acceleration -= c->comfort_acc_max * (q * q);
  }
  return acceleration;
}
static const veh_t *vp(const road_t *r, int idx) { return idx < 0 ? NULL : &r->v[idx]; }

/* controller.py:135-143 */
static void follow_road(const ix_config *c, veh_t *self) {
  if (lane_after_end(&c->lanes[self->target_lane], self->x, self->y)) self->target_lane = next_lane(c, self->target_lane, self);
}
/* behavior.py:219-263: every road of this network has ONE lane, so side_lanes() is empty and MOBIL never runs;
 * what remains is the early return while the target lane is ahead and the timer reset */
static void change_lane_policy(road_t *r, veh_t *self) {
  const ix_config *c = r->cfg;
  if (self->lane != self->target_lane) {
    const ix_lane *a = &c->lanes[self->lane], *b = &c->lanes[self->target_lane];
    if (a->from_node == b->from_node && a->to_node == b->to_node) {
      for (int j = 0; j < r->n; j++) {
        const veh_t *v = &r->v[j];
        if (v != self && v->lane != self->target_lane && v->target_lane == self->target_lane) {
          double d = lane_distance_to(c, self, v);
          double d_star = desired_gap(c, self, v);
          if (0 < d && d < d_star) {
            self->target_lane = self->lane;
            break;
          }
        }
      }
    }
    return;
  }
  if (!(LANE_CHANGE_DELAY < self->timer)) return; /* utils.do_every */
  self->timer = 0;
  /* side_lanes(lane_index) (road.py:200-211): lanes id +- 1 of the same road */
  if (road_lanes(c, c->lanes[self->lane].from_node, c->lanes[self->lane].to_node) != 1) abort();
}

/* ---- vehicle/controller.py -------------------------------------------------------------------- */
static double steering_control(const ix_config *c, const veh_t *self, int target_lane) { /* controller.py:145-187 */
  const ix_lane *l = &c->lanes[target_lane];
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
static int speed_to_index(const ix_config *c, double speed) { /* controller.py:326-344 */
  int n = c->num_target_speeds;
  double x = (speed - c->target_speeds[0]) / (c->target_speeds[n - 1] - c->target_speeds[0]);
  return (int)clipd(rint(x * (n - 1)), 0, n - 1);
}
/* ControlledVehicle.act(None) (controller.py:89-133): lateral actions do not exist here (lateral=False) */
static void controlled_act(const ix_config *c, veh_t *self) {
  follow_road(c, self);
  double steering = steering_control(c, self, self->target_lane);
  self->act_accel = KP_A * (self->target_speed - self->speed);
  self->act_steering = clipd(steering, -MAX_STEERING_ANGLE, MAX_STEERING_ANGLE);
}
/* MDPVehicle.act (controller.py:295-315); action < 0 == None */
static void mdp_act(const ix_config *c, veh_t *self, int action) {
  if (action == ACT_FASTER || action == ACT_SLOWER) {
    int idx = speed_to_index(c, self->speed) + (action == ACT_FASTER ? 1 : -1);
    if (idx < 0) idx = 0;
    if (idx > c->num_target_speeds - 1) idx = c->num_target_speeds - 1;
    self->speed_index = idx;
    self->target_speed = c->target_speeds[idx];
  }
  controlled_act(c, self);
}
/* behavior.py:93-137 */
static void idm_act(road_t *r, veh_t *self) {
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
static void clip_actions(veh_t *v) { /* kinematics.py:155-168 */
  if (v->crashed) {
    v->act_steering = 0;
    v->act_accel = -1.0 * v->speed;
  }
  if (v->speed > MAX_SPEED) v->act_accel = fmin(v->act_accel, 1.0 * (MAX_SPEED - v->speed));
  else if (v->speed < MIN_SPEED) v->act_accel = fmax(v->act_accel, 1.0 * (MIN_SPEED - v->speed));
}
static void vehicle_step(const ix_config *c, veh_t *v, double dt) { /* kinematics.py:130-153, behavior.py:139-148 */
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

/* ---- road/regulation.py -------------------------------------------------------------------------- */
/* utils.point_in_rotated_rectangle (utils.py:79-95): rotates by +angle, like the reference does */
static int point_in_rotated_rectangle(double px, double py, double cx, double cy, double length, double width,
                                      double angle) {
  double c = cos(angle), s = sin(angle);
  double dx = px - cx, dy = py - cy;
  double rux = c * dx + -s * dy, ruy = s * dx + c * dy;
  return -length / 2 <= rux && rux <= length / 2 && -width / 2 <= ruy && ruy <= width / 2;
}
/* utils.has_corner_inside (utils.py:160-174) with rect_corners(include_midpoints, include_center) (:119-157) */
static int has_corner_inside(double c1x, double c1y, double l1, double w1, double a1, double c2x, double c2y,
                             double l2, double w2, double a2) {
  const double hl = l1 / 2, hw = w1 / 2;
  const double pts[9][2] = {{-hl, -hw}, {-hl, +hw}, {+hl, +hw}, {+hl, -hw}, {0, 0}, {-hl, -0.0}, {hl, 0}, {-0.0, -hw}, {0, hw}};
  double c = cos(a1), s = sin(a1);
  int any = 0;
  for (int k = 0; k < 9; k++) {
    double x = (c * pts[k][0] + -s * pts[k][1]) + c1x;
    double y = (s * pts[k][0] + c * pts[k][1]) + c1y;
    if (point_in_rotated_rectangle(x, y, c2x, c2y, l2, w2, a2)) any = 1;
  }
  return any;
}
/* utils.rotated_rectangles_intersect (utils.py:99-116); exported for the reference's own known-answer vectors
 * (tests/test_utils.py:19-28 -> tests/test_oracle_reference_neighbours.py) */
int orc_ix_rotated_rectangles_intersect(double c1x, double c1y, double l1, double w1, double a1, double c2x, double c2y,
                                        double l2, double w2, double a2) {
  return has_corner_inside(c1x, c1y, l1, w1, a1, c2x, c2y, l2, w2, a2) ||
         has_corner_inside(c2x, c2y, l2, w2, a2, c1x, c1y, l1, w1, a1);
}
/* RegulatedRoad.is_conflict_possible (regulation.py:88-111), predict_trajectory_constant_speed (controller.py:236-253) */
static int is_conflict_possible(const ix_config *c, const veh_t *v1, const veh_t *v2) {
  double s1, s2, lat;
  lane_local(&c->lanes[v1->lane], v1->x, v1->y, &s1, &lat);
  lane_local(&c->lanes[v2->lane], v2->x, v2->y, &s2, &lat);
  for (int k = 0; k < 11; k++) { /* np.arange(0.25, 3, 0.25) */
    double t = 0.25 + k * 0.25;
    double p1x, p1y, h1, p2x, p2y, h2;
    position_heading_along_route(c, v1, s1 + v1->speed * t, &p1x, &p1y, &h1);
    position_heading_along_route(c, v2, s2 + v2->speed * t, &p2x, &p2y, &h2);
    double dx = p2x - p1x, dy = p2y - p1y;
    if (sqrt(dx * dx + dy * dy) > VEH_LENGTH) continue;
    if (has_corner_inside(p1x, p1y, 1.5 * VEH_LENGTH, 0.9 * VEH_WIDTH, h1, p2x, p2y, 1.5 * VEH_LENGTH, 0.9 * VEH_WIDTH, h2) ||
        has_corner_inside(p2x, p2y, 1.5 * VEH_LENGTH, 0.9 * VEH_WIDTH, h2, p1x, p1y, 1.5 * VEH_LENGTH, 0.9 * VEH_WIDTH, h1))
      return 1;
  }
  return 0;
}
/* front_distance_to (objects.py:205-206) */
static double front_distance_to(const veh_t *a, const veh_t *b) {
  return cos(a->heading) * (b->x - a->x) + sin(a->heading) * (b->y - a->y);
}
/* RegulatedRoad.respect_priorities (regulation.py:70-86) */
static veh_t *respect_priorities(const ix_config *c, veh_t *v1, veh_t *v2) {
  int p1 = c->lanes[v1->lane].priority, p2 = c->lanes[v2->lane].priority;
  if (p1 > p2) return v2;
  if (p1 < p2) return v1;
  return front_distance_to(v1, v2) > front_distance_to(v2, v1) ? v1 : v2;
}
/* RegulatedRoad.enforce_road_rules (regulation.py:42-68); YIELD_DURATION = 0, REGULATION_FREQUENCY = 2 */
static void enforce_road_rules(road_t *r) {
  const ix_config *c = r->cfg;
  for (int i = 0; i < r->n; i++) {
    veh_t *v = &r->v[i];
    if (v->is_yielding) {
      if (v->yield_timer >= 0.0 * 2) {
        v->target_speed = c->lanes[v->lane].speed_limit;
        v->is_yielding = 0;
      } else {
        v->yield_timer += 1;
      }
    }
  }
  for (int i = 0; i < r->n - 1; i++)
    for (int j = i + 1; j < r->n; j++)
      if (is_conflict_possible(c, &r->v[i], &r->v[j])) {
        veh_t *y = respect_priorities(c, &r->v[i], &r->v[j]);
        if (y && !y->controlled) { /* a ControlledVehicle that is not the MDPVehicle */
          y->target_speed = 0;
          y->is_yielding = 1;
          y->yield_timer = 0;
        }
      }
}

/* ---- road/road.py:464-481, regulation.py:36-40 --------------------------------------------------------- */
static void road_act(road_t *r) {
  for (int i = 0; i < r->n; i++) {
    veh_t *v = &r->v[i];
    if (v->controlled) mdp_act(r->cfg, v, -1);
    else idm_act(r, v);
  }
}
static void road_step(road_t *r, double dt) {
  r->steps += 1;
  if (r->steps % (int)(1 / dt / 2) == 0) enforce_road_rules(r);
  for (int i = 0; i < r->n; i++) vehicle_step(r->cfg, &r->v[i], dt);
  for (int i = 0; i < r->n; i++)
    for (int j = i + 1; j < r->n; j++) handle_collisions(&r->v[i], &r->v[j], dt);
}

/* ---- observation (observation.py:234-276, road.py:421-450, kinematics.py:237-261) ------------------------- */
/* Vehicle.destination / destination_direction (kinematics.py:211-235): the end of the LAST lane of the route (lane id None
 * -> 0), or the position itself without a route; the unit vector towards it, or zeros when they coincide */
static void destination_direction(const ix_config *c, const veh_t *v, double *dx, double *dy) {
  *dx = *dy = 0;
  if (v->route_len <= 0) return;
  int q = v->route_len - 1;
  int L = lane_index_of(c, v->route_from[q], v->route_to[q], v->route_id[q] < 0 ? 0 : v->route_id[q]);
  if (L < 0) return;
  double px, py;
  lane_position(&c->lanes[L], c->lanes[L].length, 0, &px, &py);
  double ex = px - v->x, ey = py - v->y;
  if (ex != 0 || ey != 0) {
    double n = sqrt(ex * ex + ey * ey);
    *dx = ex / n;
    *dy = ey / n;
  }
}
static double feature_of_c(const ix_config *c, const veh_t *v, int fid, int observed_by_another) {
  if (fid == FEAT_COS_D || fid == FEAT_SIN_D) {
    /* to_dict(origin, observe_intentions): zeroed for the OTHER vehicles unless intentions are observed; the observer's
     * own row is to_dict() with the default True (observation.py:239,253; kinematics.py:255-256) */
    if (observed_by_another && !c->obs_intentions) return 0;
    double dx, dy;
    destination_direction(c, v, &dx, &dy);
    return fid == FEAT_COS_D ? dx : dy;
  }
  if (fid == FEAT_LONG_OFF || fid == FEAT_LAT_OFF || fid == FEAT_ANG_OFF) {
    /* Vehicle.lane_offset (kinematics.py:228-235): local coordinates on the CURRENT lane and local_angle (lane.py:145-147) */
    const ix_lane *l = &c->lanes[v->lane];
    double s, lat;
    lane_local(l, v->x, v->y, &s, &lat);
    return fid == FEAT_LONG_OFF ? s : fid == FEAT_LAT_OFF ? lat : wrap_to_pi(v->heading - lane_heading_at(l, s));
  }
  switch (fid) {
    case FEAT_PRESENCE: return 1;
    case FEAT_X: return v->x;
    case FEAT_Y: return v->y;
    case FEAT_VX: return v->speed * cos(v->heading);
    case FEAT_VY: return v->speed * sin(v->heading);
    case FEAT_HEADING: return v->heading;
    case FEAT_COS_H: return cos(v->heading);
    case FEAT_SIN_H: return sin(v->heading);
  }
  return 0;
}
static double feature_of(const veh_t *v, int fid) {
  switch (fid) {
    case FEAT_PRESENCE: return 1;
    case FEAT_X: return v->x;
    case FEAT_Y: return v->y;
    case FEAT_VX: return v->speed * cos(v->heading);
    case FEAT_VY: return v->speed * sin(v->heading);
    case FEAT_HEADING: return v->heading;
    case FEAT_COS_H: return cos(v->heading);
    case FEAT_SIN_H: return sin(v->heading);
  }
  return 0;
}
static void observe_grid_agent(const road_t *r, int ego_idx, float *obs);
typedef struct { double key; int idx; } close_t;
static void observe_agent(const road_t *r, int ego_idx, float *obs) {
  const ix_config *c = r->cfg;
  if (c->obs_type == 1) { observe_grid_agent(r, ego_idx, obs); return; }
  const veh_t *ego = &r->v[ego_idx];
  int V = c->obs_vehicles, F = c->obs_features;
  close_t *close = (close_t *)malloc(sizeof(close_t) * (size_t)(r->n + 1));
  int m = 0;
  for (int j = 0; j < r->n; j++) {
    const veh_t *v = &r->v[j];
    double dx = v->x - ego->x, dy = v->y - ego->y;
    if (!(sqrt(dx * dx + dy * dy) < c->perception_distance)) continue;
    if (v == ego) continue;
    if (!(c->obs_see_behind || -2 * VEH_LENGTH < lane_distance_to(c, ego, v))) continue;
    close[m].key = fabs(lane_distance_to(c, ego, v));
    close[m].idx = j;
    m++;
  }
  for (int a = 1; a < m && !c->obs_unsorted; a++) { /* sorted() is stable; sort=False: list order */
    close_t t = close[a];
    int b = a - 1;
    while (b >= 0 && close[b].key > t.key) { close[b + 1] = close[b]; b--; }
    close[b + 1] = t;
  }
  if (m > V - 1) m = V - 1;
  for (int row = 0; row < V; row++)
    for (int f = 0; f < F; f++) {
      int fid = c->obs_feature_ids[f];
      double val = 0;
      if (row <= m) {
        const veh_t *v = row == 0 ? ego : &r->v[close[row - 1].idx];
        val = feature_of_c(c, v, fid, row > 0);
        if (row > 0 && !c->obs_absolute && (fid == FEAT_X || fid == FEAT_Y || fid == FEAT_VX || fid == FEAT_VY))
          val -= feature_of(ego, fid);
        if (c->obs_normalize) {
          const double *rg = fid == FEAT_X ? c->obs_range_x : fid == FEAT_Y ? c->obs_range_y
                           : fid == FEAT_VX ? c->obs_range_vx : fid == FEAT_VY ? c->obs_range_vy : NULL;
          if (rg && isfinite(rg[0])) {
            val = lmap(val, rg[0], rg[1], -1, 1);
            if (c->obs_clip) val = clipd(val, -1, 1);
          }
        }
      }
      obs[row * F + f] = (float)val;
    }
  free(close);
}

/* ---- OccupancyGridObservation.observe (observation.py:354-413) for the controlled vehicle ------------------------ */
static void grid_pos_to_index(const ix_config *c, const veh_t *ego, double px, double py, int relative, int *ci, int *cj) {
  if (!relative) { px -= ego->x; py -= ego->y; }
  if (c->grid_align) { /* [[c, s], [-s, c]] @ position (observation.py:431-435) */
    double cs = cos(ego->heading), sn = sin(ego->heading);
    double qx = cs * px + sn * py, qy = -sn * px + cs * py;
    px = qx; py = qy;
  }
  *ci = (int)floor((px - c->grid_min[0]) / c->grid_step[0]);
  *cj = (int)floor((py - c->grid_min[1]) / c->grid_step[1]);
}
static const double *grid_range(const ix_config *c, int fid) {
  const double *rg = fid == FEAT_X ? c->obs_range_x : fid == FEAT_Y ? c->obs_range_y
                   : fid == FEAT_VX ? c->obs_range_vx : fid == FEAT_VY ? c->obs_range_vy : NULL;
  return (rg && isfinite(rg[0])) ? rg : NULL;
}
static void observe_grid_agent(const road_t *r, int ego_idx, float *obs) {
  const ix_config *c = r->cfg;
  const veh_t *ego = &r->v[ego_idx];
  int F = c->obs_features, W = c->grid_shape[0], H = c->grid_shape[1];
  double *grid = (double *)malloc(sizeof(double) * (size_t)F * W * H);
  for (int k = 0; k < F * W * H; k++) grid[k] = NAN; /* self.grid.fill(np.nan) */
  for (int layer = 0; layer < F; layer++) {
    int fid = c->obs_feature_ids[layer];
    if (fid != FEAT_ON_ROAD) {
      for (int i = r->n - 1; i >= 0; i--) { /* df[::-1].iterrows(): the lower index wins a contested cell */
        const veh_t *v = &r->v[i];
        double x = v->x - ego->x, y = v->y - ego->y; /* to_dict(observer): relative x, y, vx, vy */
        const double *rx = grid_range(c, FEAT_X), *ry = grid_range(c, FEAT_Y);
        if (rx) { x = lmap(x, rx[0], rx[1], -1, 1); x = lmap(x, -1, 1, rx[0], rx[1]); }
        if (ry) { y = lmap(y, ry[0], ry[1], -1, 1); y = lmap(y, -1, 1, ry[0], ry[1]); }
        int ci, cj;
        grid_pos_to_index(c, ego, x, y, 1, &ci, &cj);
        if (0 <= ci && ci < W && 0 <= cj && cj < H) {
          double val = feature_of(v, fid);
          if (fid == FEAT_X || fid == FEAT_Y || fid == FEAT_VX || fid == FEAT_VY) val -= feature_of(ego, fid);
          const double *rg = grid_range(c, fid);
          if (rg) val = lmap(val, rg[0], rg[1], -1, 1);
          grid[((size_t)layer * W + ci) * H + cj] = val;
        }
      }
    } else { /* fill_road_layer_by_lanes (observation.py:454-484), every lane of the network in graph order */
      double spacing = fmin(c->grid_step[0], c->grid_step[1]);
      for (int k = 0; k < c->n_lanes; k++) {
        const ix_lane *l = &c->lanes[k];
        double origin, lat;
        lane_local(l, ego->x, ego->y, &origin, &lat);
        double start = origin - 100.0, stop = origin + 100.0;
        int n = (int)ceil((stop - start) / spacing); /* len(np.arange(start, stop, step)) */
        for (int j = 0; j < n; j++) {
          double wp = clipd(start + j * spacing, 0, l->length);
          double px, py;
          lane_position(l, wp, 0, &px, &py);
          int ci, cj;
          grid_pos_to_index(c, ego, px, py, 0, &ci, &cj);
          if (0 <= ci && ci < W && 0 <= cj && cj < H) grid[((size_t)layer * W + ci) * H + cj] = 1;
        }
      }
    }
  }
  for (int k = 0; k < F * W * H; k++) {
    double v = grid[k];
    if (c->obs_clip) v = isnan(v) ? v : clipd(v, -1, 1);
    /* as_image (observation.py:408-409): ((clip(obs, -1, 1) + 1) / 2 * 255).astype(uint8); an empty (NaN) cell casts to 0 */
    if (c->grid_image) v = isnan(v) ? 0.0 : (double)(uint8_t)((clipd(v, -1, 1) + 1) / 2 * 255);
    obs[k] = isnan(v) ? 0.0f : (float)v; /* np.nan_to_num */
  }
  free(grid);
}
static size_t ix_obs_len(const ix_config *c) {
  return c->obs_type == 1 ? (size_t)c->obs_features * c->grid_shape[0] * c->grid_shape[1] : (size_t)c->obs_vehicles * c->obs_features;
}

/* ---- IntersectionEnv reward / termination (intersection_env.py:60-126, 340-345) ----------------------------- */
static int is_exit_lane(const ix_config *c, int lane) { return c->lanes[lane].exit_lane; }
static int has_arrived(const ix_config *c, const veh_t *v, double exit_distance) { /* intersection_env.py:340-345 */
  double s, lat;
  lane_local(&c->lanes[v->lane], v->x, v->y, &s, &lat);
  return is_exit_lane(c, v->lane) && s >= exit_distance;
}
static int on_road(const ix_config *c, const veh_t *v) { /* objects.py:200-203: lane.on_lane(position), margin 0 */
  return lane_on_lane(&c->lanes[v->lane], v->x, v->y, 0.0);
}
static double agent_reward(const ix_config *c, const veh_t *v) { /* intersection_env.py:79-105 */
  double scaled_speed = lmap(v->speed, c->reward_speed_range[0], c->reward_speed_range[1], 0, 1);
  int arrived = has_arrived(c, v, 25);
  double reward = 0;
  reward = reward + c->collision_reward * (double)v->crashed;
  reward = reward + c->high_speed_reward * clipd(scaled_speed, 0, 1);
  reward = reward + c->arrived_reward * (double)arrived;
  reward = reward + 0 * (double)on_road(c, v); /* config.get("on_road_reward", 0) */
  reward = arrived ? c->arrived_reward : reward;
  reward *= (double)on_road(c, v);
  if (c->normalize_reward) reward = lmap(reward, c->collision_reward, c->arrived_reward, 0, 1);
  return reward;
}

/* ---- SoA <-> AoS ------------------------------------------------------------------------------------------------ */
static int load_env(const ix_config *c, const ix_state *st, int e, veh_t *v) {
  int C = c->n_slots, R = c->n_route, n = 0;
  for (int i = 0; i < C; i++) {
    size_t k = (size_t)e * C + i;
    if (!st->present[k]) continue;
    veh_t *o = &v[n++];
    memset(o, 0, sizeof(*o));
    o->x = st->x[k]; o->y = st->y[k]; o->heading = st->heading[k]; o->speed = st->speed[k];
    o->timer = st->timer[k]; o->target_speed = st->target_speed[k]; o->delta = st->delta[k];
    o->impact_x = st->impact_x[k]; o->impact_y = st->impact_y[k];
    o->lane = st->lane[k]; o->target_lane = st->target_lane[k]; o->speed_index = st->speed_index[k];
    o->crashed = st->crashed[k]; o->has_impact = st->has_impact[k]; o->controlled = st->controlled[k];
    o->is_yielding = st->is_yielding[k]; o->yield_timer = st->yield_timer[k]; o->route_len = st->route_len[k];
    o->impact_margin = o->flag_margin = INFINITY;
    for (int q = 0; q < R && q < IX_MAX_ROUTE; q++) {
      o->route_from[q] = st->route_from[k * R + q];
      o->route_to[q] = st->route_to[k * R + q];
      o->route_id[q] = st->route_id[k * R + q];
    }
  }
  return n;
}
static void store_env(const ix_config *c, ix_state *st, int e, const veh_t *v, int n) {
  int C = c->n_slots, R = c->n_route;
  for (int i = 0; i < C; i++) {
    size_t k = (size_t)e * C + i;
    st->present[k] = i < n;
    if (orc_margin_buf) orc_margin_buf[k] = i < n ? v[i].impact_margin : INFINITY;
    if (orc_flag_margin_buf) orc_flag_margin_buf[k] = i < n ? v[i].flag_margin : INFINITY;
    if (i >= n) continue;
    const veh_t *o = &v[i];
    st->x[k] = o->x; st->y[k] = o->y; st->heading[k] = o->heading; st->speed[k] = o->speed;
    st->timer[k] = o->timer; st->target_speed[k] = o->target_speed; st->delta[k] = o->delta;
    st->impact_x[k] = o->impact_x; st->impact_y[k] = o->impact_y;
    st->lane[k] = o->lane; st->target_lane[k] = o->target_lane; st->speed_index[k] = o->speed_index;
    st->crashed[k] = o->crashed; st->has_impact[k] = o->has_impact; st->controlled[k] = o->controlled;
    st->is_yielding[k] = o->is_yielding; st->yield_timer[k] = o->yield_timer; st->route_len[k] = o->route_len;
    for (int q = 0; q < R && q < IX_MAX_ROUTE; q++) {
      st->route_from[k * R + q] = q < o->route_len ? o->route_from[q] : -1;
      st->route_to[k * R + q] = q < o->route_len ? o->route_to[q] : -1;
      st->route_id[k * R + q] = q < o->route_len ? o->route_id[q] : -1;
    }
  }
}
/* env.controlled_vehicles[a]: the controlled vehicles keep their relative order in Road.vehicles (they are appended in
 * that order, intersection_env.py:310-311, and every later filter of the list is stable) */
static int agent_index(const road_t *r, int a) {
  for (int i = 0; i < r->n; i++)
    if (r->v[i].controlled && a-- == 0) return i;
  return -1;
}
static int n_agents(const ix_config *c) { return c->num_agents > 0 ? c->num_agents : 1; }

size_t orc_ix_config_size(void) { return sizeof(ix_config); }

/* n_frames x {[meta-action on the first frame]; Road.act(); RegulatedRoad.step(dt)} (abstract.py:287-317) */
int orc_ix_frames(const ix_config *c, ix_state *st, const int32_t *actions, int32_t n_frames) {
  veh_t *buf = (veh_t *)malloc(sizeof(veh_t) * (size_t)c->n_slots);
  for (int e = 0; e < c->num_envs; e++) {
    road_t r = {c, buf, 0, st->road_steps[e]};
    r.n = load_env(c, st, e, buf);
    for (int f = 0; f < n_frames; f++) {
      if (f == 0 && actions) { /* MultiAgentAction.act: one action per controlled vehicle, in order (action.py:352-355) */
        for (int a = 0; a < n_agents(c); a++) {
          int act = actions[e * n_agents(c) + a], ego = agent_index(&r, a);
          if (act < 0 || act > 2) { free(buf); return -6; }
          if (ego >= 0) mdp_act(c, &r.v[ego], act); /* action_type.act(action), action.py:259-260 */
        }
      }
      road_act(&r);
      road_step(&r, c->dt);
    }
    st->road_steps[e] = r.steps;
    store_env(c, st, e, buf, r.n);
  }
  free(buf);
  return 0;
}

/* Road.neighbour_vehicles(vehicle = the slot-th PRESENT vehicle of environment e, lane_index = table index `lane`) ->
 * list positions of the preceding / following vehicle or -1 (tests/test_oracle_reference_neighbours.py) */
int orc_ix_neighbours(const ix_config *c, const ix_state *st, int32_t e, int32_t slot, int32_t lane, int32_t *front,
                      int32_t *rear) {
  veh_t *buf = (veh_t *)malloc(sizeof(veh_t) * (size_t)c->n_slots);
  road_t r = {c, buf, 0, 0};
  r.n = load_env(c, st, e, buf);
  int rc = -1;
  if (slot >= 0 && slot < r.n && lane >= 0 && lane < c->n_lanes) {
    int f, b;
    neighbour_vehicles(&r, &buf[slot], lane, &f, &b);
    *front = f; *rear = b;
    rc = 0;
  }
  free(buf);
  return rc;
}

int orc_ix_observe(const ix_config *c, const ix_state *st, float *obs) {
  veh_t *buf = (veh_t *)malloc(sizeof(veh_t) * (size_t)c->n_slots);
  size_t per = ix_obs_len(c);
  for (int e = 0; e < c->num_envs; e++) {
    road_t r = {c, buf, 0, st->road_steps[e]};
    r.n = load_env(c, st, e, buf);
    for (int a = 0; a < n_agents(c); a++) { /* MultiAgentObservation.observe (observation.py:733-734) */
      int ego = agent_index(&r, a);
      if (ego >= 0) observe_agent(&r, ego, obs + ((size_t)e * n_agents(c) + a) * per);
    }
  }
  free(buf);
  return 0;
}

/* AbstractEnv.step up to (not including) IntersectionEnv.step's clear / spawn (abstract.py:259-285) */
int orc_ix_step(const ix_config *c, ix_state *st, const int32_t *actions, float *obs, double *reward, uint8_t *terminated,
                uint8_t *truncated, double *info_speed, uint8_t *info_crashed) {
  int rc = orc_ix_frames(c, st, actions, c->frames_per_step);
  if (rc) return rc;
  veh_t *buf = (veh_t *)malloc(sizeof(veh_t) * (size_t)c->n_slots);
  size_t per = ix_obs_len(c);
  for (int e = 0; e < c->num_envs; e++) {
    road_t r = {c, buf, 0, st->road_steps[e]};
    r.n = load_env(c, st, e, buf);
    st->time[e] += c->policy_dt;
    /* reward[e][a] = _agent_reward (the env's scalar is their mean, :62-66); info_crashed[e][a] bit 0 = crashed, bit 1 =
     * has_arrived (agents_terminated = either, :119-121); terminated = any crashed or all arrived or the FIRST one off
     * the road (:107-112) */
    int A = n_agents(c), any_crashed = 0, all_arrived = 1;
    for (int a = 0; a < A; a++) {
      int ego = agent_index(&r, a);
      if (ego < 0) { free(buf); return -7; }
      const veh_t *v = &r.v[ego];
      observe_agent(&r, ego, obs + ((size_t)e * A + a) * per);
      reward[e * A + a] = agent_reward(c, v);
      int arrived = has_arrived(c, v, 25);
      any_crashed |= v->crashed;
      all_arrived &= arrived;
      if (info_speed) info_speed[e * A + a] = v->speed;
      if (info_crashed) info_crashed[e * A + a] = (uint8_t)((v->crashed ? 1 : 0) | (arrived ? 2 : 0));
    }
    terminated[e] = any_crashed || all_arrived || (c->offroad_terminal && !on_road(c, &r.v[agent_index(&r, 0)]));
    truncated[e] = st->time[e] >= c->duration;
  }
  free(buf);
  return 0;
}

/* IntersectionEnv._clear_vehicles (intersection_env.py:326-338) then _spawn_vehicle (:292-324) on the given draws:
 * draws[e] = [uniform(), choice(range(4), 2, replace=False) x2, normal(), normal(), uniform(delta)] in call order, as
 * many as the reference consumed (n_draws[e]); returns the number of draws the rule consumed in n_used[e]. */
int orc_ix_clear_spawn(const ix_config *c, ix_state *st, const double *draws, const int32_t *n_draws, int32_t draws_stride,
                       int32_t *n_used) {
  veh_t *buf = (veh_t *)malloc(sizeof(veh_t) * (size_t)(c->n_slots + 1));
  for (int e = 0; e < c->num_envs; e++) {
    road_t r = {c, buf, 0, st->road_steps[e]};
    r.n = load_env(c, st, e, buf);
    /* _clear_vehicles: keep controlled ones; drop vehicles leaving on an exit lane (s >= length - 4 * LENGTH);
     * `vehicle.route is None` never happens here (plan_route_to always leaves a list) */
    int m = 0;
    for (int i = 0; i < r.n; i++) {
      veh_t *v = &r.v[i];
      double s, lat;
      lane_local(&c->lanes[v->lane], v->x, v->y, &s, &lat);
      int leaving = is_exit_lane(c, v->lane) && s >= c->lanes[v->lane].length - 4 * VEH_LENGTH;
      if (v->controlled || !leaving) { if (m != i) r.v[m] = *v; m++; }
    }
    r.n = m;
    /* _spawn_vehicle(longitudinal=0, position_deviation=1, speed_deviation=1, spawn_probability) */
    const double *d = draws + (size_t)e * draws_stride;
    int used = 0, avail = n_draws[e];
    do {
      if (avail < 1) break;
      double u = d[used++];
      if (u > c->spawn_probability) break;
      if (avail < 5) { free(buf); return -8; }
      int r0 = (int)d[used++], r1 = (int)d[used++];
      double n_pos = d[used++], n_speed = d[used++];
      const ix_lane *lane = &c->lanes[c->access_lane[r0]]; /* ("o" + r0, "ir" + r0, 0) */
      veh_t nv;
      memset(&nv, 0, sizeof(nv));
      double lon = 0 + 5.0 + n_pos * 1.0;
      lane_position(lane, lon, 0, &nv.x, &nv.y);
      nv.heading = lane_heading_at(lane, lon);
      nv.speed = 8.0 + n_speed * 1.0;
      int too_close = 0;
      for (int i = 0; i < r.n; i++) {
        double dx = r.v[i].x - nv.x, dy = r.v[i].y - nv.y;
        if (sqrt(dx * dx + dy * dy) < 15) too_close = 1;
      }
      if (too_close) break;
      /* Vehicle ctor (objects.py:42-51, controller.py:42-52, behavior.py:46-64) */
      nv.lane = closest_lane_index(c, nv.x, nv.y, nv.heading);
      nv.target_lane = nv.lane;
      nv.target_speed = nv.speed;
      nv.timer = py_mod((nv.x + nv.y) * M_PI, LANE_CHANGE_DELAY);
      /* plan_route_to("o" + r1) (controller.py:71-87): shortest path lane_index[1] -> destination by BFS; on this
       * network it is always [ir -> il_dest, il_dest -> o_dest] */
      {
        const ix_lane *cur = &c->lanes[nv.lane];
        int dest_node = c->outer_node[r1]; /* "o" + r1 */
        nv.route_len = 1;
        nv.route_from[0] = cur->from_node; nv.route_to[0] = cur->to_node; nv.route_id[0] = cur->id;
        /* BFS over nodes from cur->to_node to dest_node (road.py:156-188: bfs_paths, first found == shortest) */
        int prev[64], queue[64], qh = 0, qt = 0, seen[64];
        for (int k = 0; k < 64; k++) { prev[k] = -1; seen[k] = 0; }
        queue[qt++] = cur->to_node;
        seen[cur->to_node] = 1;
        int found = cur->to_node == dest_node;
        while (qh < qt && !found) {
          int node = queue[qh++];
          for (int k = 0; k < c->n_lanes && !found; k++) {
            if (c->lanes[k].from_node != node || c->lanes[k].id != 0) continue;
            int nx = c->lanes[k].to_node;
            if (seen[nx]) continue;
            seen[nx] = 1;
            prev[nx] = node;
            queue[qt++] = nx;
            if (nx == dest_node) found = 1;
          }
        }
        if (found && cur->to_node != dest_node) {
          int path[IX_MAX_ROUTE + 2], np_ = 0;
          for (int node = dest_node; node >= 0 && np_ < IX_MAX_ROUTE + 2; node = prev[node]) path[np_++] = node;
          for (int k = np_ - 1; k > 0 && nv.route_len < IX_MAX_ROUTE; k--) {
            nv.route_from[nv.route_len] = path[k];
            nv.route_to[nv.route_len] = path[k - 1];
            nv.route_id[nv.route_len] = -1;
            nv.route_len++;
          }
        }
      }
      if (avail < 6) { free(buf); return -8; }
      nv.delta = d[used++]; /* randomize_behavior (behavior.py:66-69) */
      if (r.n >= c->n_slots) { free(buf); return -10; }
      r.v[r.n++] = nv;
    } while (0);
    if (n_used) n_used[e] = used;
    store_env(c, st, e, buf, r.n);
  }
  free(buf);
  return 0;
}

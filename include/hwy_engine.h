/*
 * hwy_engine.h -- C-ABI of the MI355X-native batched HighwayEnv step engine.
 *
 * The reference (Farama-Foundation/HighwayEnv) has no FFI: its hot path is the
 * Python method seam
 *
 *     AbstractEnv.step      highway_env/envs/common/abstract.py:259-285
 *       AbstractEnv._simulate                              :287-317
 *         ActionType.act    envs/common/action.py:259-260  (DiscreteMetaAction)
 *         Road.act          road/road.py:464-467
 *         Road.step         road/road.py:469-481
 *       KinematicObservation.observe   envs/common/observation.py:234-276
 *       HighwayEnv._reward/_is_terminated/_is_truncated  envs/highway_env.py:100-151
 *     AbstractEnv.reset     envs/common/abstract.py:219-249  (HighwayEnv._create_vehicles :72-98)
 *
 * Each entry point below names the reference interface it replaces.  All
 * functions are extern "C", take plain pointers and sizes, return 0 on success
 * or a negative hwy_status; the message of the last failure on an engine is
 * available through hwy_last_error().  One host thread per engine; calls on one
 * engine must be serialised; engines on different GPUs are independent.
 *
 * Data model.  E environments x N vehicles (vehicle index == position in the
 * reference's Road.vehicles list; controlled vehicles are where
 * HighwayEnv._create_vehicles puts them, index 0 for a single agent).  All
 * floating-point state is f64 like the reference (vehicle/objects.py:43);
 * observations are f32 (observation.py:276).  Host-side arrays are row-major
 * [E][N] (state), [E][A][V][F] (Kinematics obs) or [E][A][F][W][H] (OccupancyGrid obs),
 * [E][A] (actions) unless stated otherwise.
 */
#ifndef HWY_ENGINE_H
#define HWY_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HWY_ABI_VERSION 5

#define HWY_MAX_AGENTS 16
#define HWY_MAX_FEATURES 16
#define HWY_MAX_TARGET_SPEEDS 8
#define HWY_MAX_LANES 16
#define HWY_MAX_VEHICLES 256
#define HWY_MAX_GRID_CELLS 65536
#define HWY_MAX_GLANES 24  /* lanes of a general (any direction / circular) road network: HWY_SCENARIO_INTERSECTION */
#define HWY_MAX_ROUTE 11   /* remaining roads of a planned route kept per vehicle (64-bit route word, 5 bits per road) */

typedef enum hwy_status {
  HWY_OK = 0,
  HWY_ERR_INVALID_ARG = -1,
  HWY_ERR_HIP = -2,
  HWY_ERR_UNSUPPORTED = -3,
  HWY_ERR_NO_DEVICE = -4,
  HWY_ERR_ACTION = -5 /* meta-action outside the configured table: the reference raises KeyError (action.py:260) */
} hwy_status;

/* per-vehicle flag bits (hwy_state.flags) */
enum {
  HWY_F_CRASHED = 1,          /* RoadObject.crashed            vehicle/objects.py:64  */
  HWY_F_HAS_IMPACT = 2,       /* Vehicle.impact is not None    vehicle/kinematics.py:47,145-148 */
  HWY_F_CHECK_COLLISIONS = 4, /* RoadObject.check_collisions   vehicle/objects.py:61 (HighwayEnvFast clears it, highway_env.py:177-182) */
  HWY_F_CONTROLLED = 8,       /* MDPVehicle (ego) instead of IDMVehicle */
  /* road-network scenarios (hwy_config.scenario != HWY_SCENARIO_HIGHWAY) only: */
  HWY_F_OBSTACLE = 16,        /* the slot is an Obstacle of Road.objects (vehicle/objects.py:215-222): 2 m x 2 m,
                                 never acts or moves; slots of obstacles come AFTER every vehicle slot, like
                                 `self.vehicles + self.objects` in road.py:531 */
  HWY_F_ABSENT = 32,          /* empty slot: MergeGenericEnv's rejection-sampled spawn (merge_env.py:336-352)
                                 creates a different number of vehicles per episode; IntersectionEnv clears and spawns
                                 vehicles between policy steps (intersection_env.py:136-140) */
  HWY_F_YIELDING = 64         /* HWY_SCENARIO_INTERSECTION: RegulatedRoad made the vehicle yield (is_yielding,
                                 road/regulation.py:60-68); YIELD_DURATION is 0, so yield_timer is never read */
};

/* config flag bits (hwy_config.flags) */
enum {
  HWY_C_NORMALIZE_REWARD = 1, /* config["normalize_reward"]   highway_env.py:108-116 */
  HWY_C_OFFROAD_TERMINAL = 2, /* config["offroad_terminal"]   highway_env.py:141-147 */
  HWY_C_OBS_ABSOLUTE = 4,     /* KinematicObservation.absolute      observation.py:167 */
  HWY_C_OBS_NORMALIZE = 8,    /* KinematicObservation.normalize     observation.py:169 */
  HWY_C_OBS_CLIP = 16,        /* KinematicObservation.clip          observation.py:170 */
  HWY_C_OBS_SEE_BEHIND = 32,  /* KinematicObservation.see_behind    observation.py:171 */
  HWY_C_EGO_ONLY_COLLISIONS = 64, /* HighwayEnvFast: spawned traffic has check_collisions=False (highway_env.py:177-182) */
  HWY_C_GRID_ALIGN = 128,     /* OccupancyGridObservation.align_to_vehicle_axes  observation.py:294,431-435 */
  HWY_C_CONNECTED_LANES = 512, /* Road.neighbour_vehicles_connected_lanes (road.py:483-547; merge-v1, merge-generic-v1): the
                                 leader / follower search on a lane also looks at hwy_lane.connected */
  HWY_C_OBS_UNSORTED = 1024,  /* KinematicObservation(order="shuffled") (observation.py:245,273-274): close_objects_to(sort=False)
                                 keeps the first vehicles_count - 1 eligible objects in LIST order; the shuffle of the rows
                                 itself draws from env.np_random and is done by the host side of the binding */
  HWY_C_OBS_VEHICLES_ONLY = 2048, /* KinematicObservation(include_obstacles=False) (observation.py:172,246): objects of
                                 Road.objects (the merge scenarios' Obstacle) are not observed */
  HWY_C_OBS_INTENTIONS = 4096, /* KinematicObservation(observe_intentions=True) (observation.py:171,253): the cos_d / sin_d features
                                (Vehicle.destination_direction, kinematics.py:211-235) of the OTHER vehicles too, not only the
                                observer's own; only vehicles with a route have a destination (HWY_SCENARIO_INTERSECTION) */
  HWY_C_GRID_IMAGE = 8192,    /* OccupancyGridObservation(as_image=True) (observation.py:296,408-409): every cell holds
                                 uint8(((clip(value, -1, 1) + 1) / 2) * 255) -- written as an integer-valued f32, 0 for an empty
                                 (NaN) cell -- instead of the value */
  HWY_C_HOST_TRAFFIC = 256    /* HWY_SCENARIO_INTERSECTION: the HOST clears / spawns vehicles between policy steps (the
                                 reference-stream mode of highwayenv_amd/intersection.py); otherwise the step kernel does
                                 it on Philox draws */
};

/* observation feature ids (Vehicle.to_dict keys, vehicle/kinematics.py:237-261) */
enum {
  HWY_FEAT_PRESENCE = 0, HWY_FEAT_X, HWY_FEAT_Y, HWY_FEAT_VX, HWY_FEAT_VY, HWY_FEAT_HEADING,
  HWY_FEAT_COS_H, HWY_FEAT_SIN_H, HWY_FEAT_COS_D, HWY_FEAT_SIN_D, HWY_FEAT_LONG_OFF,
  HWY_FEAT_LAT_OFF, HWY_FEAT_ANG_OFF,
  HWY_FEAT_ON_ROAD, /* OccupancyGrid only: the on-road layer (observation.py:454-484) */
  HWY_FEAT_COUNT
};

/* observation types (hwy_config.obs_type) */
enum { HWY_OBS_KINEMATICS = 0, HWY_OBS_OCCUPANCY_GRID = 1 };

/* scenarios (hwy_config.scenario) */
enum {
  HWY_SCENARIO_HIGHWAY = 0, /* HighwayEnv / HighwayEnvFast: one straight road, lanes_count lanes (highway_env.py:59-70) */
  HWY_SCENARIO_MERGE = 1,   /* MergeEnv: x-aligned road network in hwy_config.net (merge_env.py:90-160), 3 + 1 traffic vehicles */
  HWY_SCENARIO_MERGE_GENERIC = 2, /* MergeGenericEnv (merge_env.py:233-363): same kernels, rejection-sampled spawn */
  HWY_SCENARIO_INTERSECTION = 3   /* IntersectionEnv (intersection_env.py): general lane table hwy_config.gnet, planned
                                     routes, RegulatedRoad, vehicles cleared / spawned between policy steps; meta-actions
                                     are {0: SLOWER, 1: IDLE, 2: FASTER} (intersection_env.py:14) */
};

/*
 * One lane of a GENERAL RoadNetwork (HWY_SCENARIO_INTERSECTION): a StraightLane of any direction (road/lane.py:159-213)
 * or a CircularLane (road/lane.py:311-369).  Table order == iteration order of get_closest_lane_index (road.py:55-71).
 * Every road (from, to) of such a network holds ONE lane, so a planned route is a list of table indices.
 */
typedef struct hwy_glane {
  int32_t kind;                       /* 0 StraightLane, 1 CircularLane */
  int32_t direction;                  /* CircularLane.direction: +1 clockwise, -1 otherwise */
  int32_t priority;                   /* lane.priority (RegulatedRoad.respect_priorities, regulation.py:70-86) */
  int32_t forbidden;
  int32_t from_node, to_node;         /* graph node ids; the lanes leaving a node are the candidates of next_lane */
  int32_t exit_lane;                  /* "il" in lane_index[0] and "o" in lane_index[1] (intersection_env.py:328-331,341-344) */
  int32_t reserved;
  double sx, sy;                      /* StraightLane.start */
  double heading, dirx, diry;         /* StraightLane.heading (arctan2), .direction */
  double cx, cy, radius, start_phase; /* CircularLane.center / radius / start_phase */
  double length, width, speed_limit;
} hwy_glane;

/*
 * One lane of an x-aligned RoadNetwork (road/road.py:16-38): StraightLane (road/lane.py:150-233) when
 * amplitude == 0, else SineLane (road/lane.py:236-283).  direction == (1, 0) for every lane, so
 * local_coordinates(p) = (p.x - x0, p.y - y0 [- amplitude*sin(pulsation*(p.x - x0) + phase)]).
 * Table order == iteration order of RoadNetwork.get_closest_lane_index (road.py:55-71: graph[from][to][id]),
 * which is what its argmin tie-break depends on; lanes of one road (from, to) are consecutive, ordered by id.
 */
typedef struct hwy_lane {
  double x0, y0;                      /* lane.start */
  double length;                      /* |end - start| */
  double width;                       /* AbstractLane.DEFAULT_WIDTH = 4 */
  double amplitude, pulsation, phase; /* SineLane; amplitude == 0 => StraightLane */
  double speed_limit;                 /* lane.speed_limit (20 unless given, lane.py:166) */
  int32_t road;                       /* id of the (from, to) pair */
  int32_t id;                         /* lane id on that road: lane_index[2] */
  int32_t road_first;                 /* table index of lane 0 of the same road */
  int32_t road_lanes;                 /* len(graph[from][to]) */
  int32_t next_first;                 /* table index of lane 0 of the road that starts at `to`, -1 if none */
  int32_t next_lanes;                 /* its lane count (RoadNetwork.next_lane, road.py:73-127) */
  int32_t forbidden;                  /* lane.forbidden (is_reachable_from, lane.py:110-111) */
  int32_t connected;                  /* bit K: lane K is searched together with this lane when HWY_C_CONNECTED_LANES is
                                         set -- the lane itself, lane `id` (else 0) of the successor road, lane `id` (else 0) of
                                         every road ending where this one starts (road.py:508-529) */
} hwy_lane;

/* meta-actions: DiscreteMetaAction.ACTIONS_ALL, envs/common/action.py:204 */
enum { HWY_LANE_LEFT = 0, HWY_IDLE = 1, HWY_LANE_RIGHT = 2, HWY_FASTER = 3, HWY_SLOWER = 4 };
/* hwy_config.action_set: ACTIONS_ALL / ACTIONS_LONGI / ACTIONS_LAT (action.py:204-210) */
enum { HWY_ACTIONS_ALL = 0, HWY_ACTIONS_LONGI = 1, HWY_ACTIONS_LAT = 2 };
/* id in the configured table -> id in ACTIONS_ALL (what the kernels and the oracle act on).  An id outside the table maps
 * to HWY_IDLE in all three tables: hwy_step (host pointers) rejects such ids with HWY_ERR_ACTION before any launch -- the
 * reference's KeyError (action.py:260) -- but hwy_step_device cannot look at device memory without a synchronisation, so an
 * on-GPU policy that emits an out-of-table id gets IDLE, never another table's action. */
#define HWY_ACTION_TO_ALL(set, a)                                                                          \
  ((set) == HWY_ACTIONS_LONGI ? ((a) == 0 ? HWY_SLOWER : ((a) == 2 ? HWY_FASTER : HWY_IDLE))                \
                              : ((unsigned)(a) <= ((set) == HWY_ACTIONS_LAT ? 2u : 4u) ? (a) : HWY_IDLE))
#define HWY_NUM_ACTIONS(set) ((set) == HWY_ACTIONS_ALL ? 5 : 3)

/*
 * Flat POD derived from the reference's config dict
 * (HighwayEnv.default_config, highway_env.py:25-53; AbstractEnv.default_config,
 * abstract.py:101-125; KinematicObservation.__init__, observation.py:160-197).
 */
typedef struct hwy_config {
  int32_t abi_version;                 /* = HWY_ABI_VERSION */
  int32_t num_envs;                    /* E */
  int32_t num_vehicles;                /* N = vehicles_count + controlled_vehicles */
  int32_t num_agents;                  /* A = controlled_vehicles (HWY_SCENARIO_INTERSECTION: 1..4, MultiAgentIntersectionEnv) */
  int32_t agent_index[HWY_MAX_AGENTS]; /* index of each controlled vehicle in the vehicle list (HWY_SCENARIO_INTERSECTION: unused --
                                          its list is re-compacted while an episode runs, so agent a is the a-th slot that carries
                                          HWY_F_CONTROLLED) */
  int32_t lanes_count;                 /* L: lane k is centred on y = k*lane_width (road.py:291-321); merge: highway lanes */
  int32_t frames_per_step;             /* T = simulation_frequency // policy_frequency (abstract.py:289-291) */
  int32_t flags;                       /* HWY_C_* */
  int32_t obs_vehicles;                /* V = observation.vehicles_count */
  int32_t obs_features;                /* F */
  int32_t obs_feature_ids[HWY_MAX_FEATURES];
  int32_t num_target_speeds;           /* MDPVehicle.target_speeds (controller.py:259,287-291) */
  int32_t action_set;                  /* DiscreteMetaAction(longitudinal, lateral) (action.py:204-253): which table the action
                                          ids index -- HWY_ACTIONS_ALL (5 ids), HWY_ACTIONS_LONGI {0 SLOWER, 1 IDLE, 2 FASTER},
                                          HWY_ACTIONS_LAT {0 LANE_LEFT, 1 IDLE, 2 LANE_RIGHT}.  HWY_SCENARIO_INTERSECTION always
                                          uses the longitudinal table (intersection_env.py:14) */
  double target_speeds[HWY_MAX_TARGET_SPEEDS];
  double dt;                           /* 1 / simulation_frequency  (abstract.py:307) */
  double policy_dt;                    /* 1 / policy_frequency      (abstract.py:274) */
  double duration;                     /* config["duration"] [s]    (highway_env.py:149-151) */
  double lane_width;                   /* AbstractLane.DEFAULT_WIDTH = 4 (lane.py:16) */
  double road_length;                  /* 10000 (road.py:296) */
  double speed_limit;                  /* 30 (highway_env.py:63) */
  double collision_reward, right_lane_reward, high_speed_reward; /* highway_env.py:38-45 */
  double reward_speed_range[2];        /* highway_env.py:46 */
  double perception_distance;          /* AbstractEnv.PERCEPTION_DISTANCE = 5*MAX_SPEED (abstract.py:58) */
  double obs_range_x[2], obs_range_y[2], obs_range_vx[2], obs_range_vy[2]; /* observation.py:214-226 */
  /* OccupancyGridObservation (observation.py:279-499): obs is f32 [E][A][F][W][H]; features = obs_feature_ids
   * (to_dict keys + HWY_FEAT_ON_ROAD); features_range in obs_range_* (default: vx, vy only, +-inf = absent) */
  int32_t obs_type;                    /* HWY_OBS_* */
  int32_t grid_shape[2];               /* W, H = floor((grid_size[:,1] - grid_size[:,0]) / grid_step) */
  int32_t reserved1;
  double grid_min[2];                  /* grid_size[:,0] */
  double grid_step[2];                 /* grid_step */
  /* road-network scenarios (ABI v3).  scenario == HWY_SCENARIO_HIGHWAY ignores everything below. */
  int32_t scenario;                    /* HWY_SCENARIO_* */
  int32_t net_lanes;                   /* entries used in net[] (<= HWY_MAX_LANES) */
  int32_t merge_lane;                  /* table index of ("b","c",2) -- literally id 2, merge_env.py:72 -- whose vehicles
                                          pay the altruistic penalty of MergeEnv._rewards (:68-75); -1 if absent */
  int32_t reserved2;
  double merge_end_x;                  /* _is_terminated: ego x > 370 (merge_env.py:79) / end_position (:369) */
  double merging_speed_reward;         /* merge_env.py:35 */
  double lane_change_reward;           /* merge_env.py:36 */
  hwy_lane net[HWY_MAX_LANES];
  /* HWY_SCENARIO_INTERSECTION (ABI v4) */
  int32_t gnet_lanes;                  /* entries used in gnet[] */
  int32_t initial_vehicle_count;       /* config["initial_vehicle_count"] (device reset, intersection_env.py:232-290) */
  int32_t destination;                 /* k of config["destination"] == "o" + k (the ego's route, intersection_env.py:262-275);
                                          -1: config["destination"] is None -- "o" + str(np_random.integers(1, 4)) per episode
                                          (:295-297), drawn by the device reset on its own counter-based stream */
  int32_t reserved3;
  int32_t access_lane[4];              /* table index of ("o" + k, "ir" + k, 0): where _spawn_vehicle puts new traffic */
  int32_t exit_of[4];                  /* table index of ("il" + k, "o" + k, 0): the last road of a route to "o" + k */
  double spawn_probability;            /* config["spawn_probability"] */
  double arrived_reward;               /* config["arrived_reward"] */
  double idm_distance_wanted, idm_time_wanted, idm_comfort_acc_max, idm_comfort_acc_min; /* set on the vehicle class by
                                          IntersectionEnv._make_vehicles (intersection_env.py:243-247): 7, 1.5, 6, -3 */
  hwy_glane gnet[HWY_MAX_GLANES];
  int64_t gnet_routes[HWY_MAX_GLANES][4]; /* ControlledVehicle.plan_route_to("o" + k) (controller.py:71-87) for a vehicle on lane L:
                                          the route word (hwy_state.route encoding) of the roads AFTER L on the shortest path from
                                          L's end node to "o" + k -- RoadNetwork.shortest_path (road.py:159-188: breadth-first,
                                          neighbours in sorted name order), planned by the host once per network; length 0 = no path */
  /* Tuning (ABI v5; 0 everywhere = the engine's own choice).  These replace the process-global environment variables
   * earlier builds read with getenv: a knob now belongs to ONE engine and is part of its documented configuration.
   * None of them changes any result (tests/test_engine_parity.py, tests/test_ix_parity.py compare the variants). */
  int32_t tune_block_kernel;           /* 0: the engine's choice -- one wavefront per environment for N <= 128 (hwy_wave.h for N <= 64,
                                          hwy_wave2.h with two vehicles per thread for 64 < N <= 128 and the Kinematics observation), the
                                          generic workgroup kernel (hwy_device.h) beyond; 1: the workgroup kernel everywhere; 2: the
                                          one-wavefront kernels wherever they exist (hwy_wave2.h with three / four vehicles per thread
                                          for N <= 192 / 256: bit-identical, measured slower than the workgroup kernel there) */
  int32_t tune_waves_per_eu;           /* 1..4: register-allocation variant (resident wavefronts per SIMD) of the step kernel;
                                          no effect where the wide kernel runs (64 < N <= 128, Kinematics: one build, hwy_wave2.h) */
  int32_t tune_ix_no_helpers;          /* 1: HWY_SCENARIO_INTERSECTION with N <= 32 runs 32-thread workgroups (no helper lanes) */
  int32_t tune_ix_no_prewarm;          /* 1: HWY_SCENARIO_INTERSECTION auto-resets run their warm-up frames inline */
  int32_t tune_extra_lds;              /* bytes of dynamic LDS per workgroup of the one-wavefront step kernel (<= 65536):
                                          fewer resident wavefronts per SIMD, the rest dispatched as wavefronts retire */
  int32_t tune_prio_shift;             /* one-wavefront step kernels: wavefronts sharing a SIMD take turns at the top issue
                                          priority (s_setprio), a turn lasting 2^shift clock ticks of s_memtime for
                                          1 <= shift <= 30, or shift x 64 ticks for 64 <= shift <= 2^20 (turn lengths
                                          between the powers of two: the optimum is sharp, profiles/r05_history.md);
                                          -1 = off (hardware order: oldest wavefront first), 0 = the engine's default
                                          (a turn of ~7 us for the 5 frames of highway-fast-v0, longer with more frames)
                                          when the whole grid of the step kernel is resident at once (occupancy x
                                          compute units >= num_envs), else off */
  int32_t tune_ix_prewarm_frames;      /* HWY_SCENARIO_INTERSECTION auto-reset: warm-up frames of the NEXT episode advanced per
                                          launch by the pre-warming workgroup of an environment (0 = a third of frames_per_step) */
  int32_t tune_reserved[1];
} hwy_config;

/*
 * Struct-of-arrays view of the full simulation state in HOST memory, each
 * vehicle array [E*N] row-major, `time` [E].  Used by hwy_set_state /
 * hwy_get_state (the reference equivalent is reaching into
 * env.road.vehicles[i].{position,heading,speed,lane_index,target_lane_index,
 * crashed,impact,timer,DELTA,target_speed,speed_index}).  Any pointer may be
 * NULL in hwy_get_state (field skipped); all must be non-NULL in hwy_set_state.
 * HWY_SCENARIO_INTERSECTION: for a slot whose flags hold HWY_F_ABSENT only `flags` is meaningful -- the step kernel
 * neither reads nor writes the other planes of an empty slot.
 */
typedef struct hwy_state {
  double *x, *y, *heading, *speed;      /* RoadObject.position/heading/speed     objects.py:42-45 */
  double *timer;                        /* IDMVehicle.timer                      behavior.py:64   */
  double *target_speed;                 /* ControlledVehicle.target_speed        controller.py:47 */
  double *delta;                        /* IDMVehicle.DELTA (randomize_behavior) behavior.py:66-69 */
  double *impact_x, *impact_y;          /* Vehicle.impact (valid iff HWY_F_HAS_IMPACT) */
  int32_t *lane;                        /* lane_index[2]; road-network scenarios: index into hwy_config.net */
  int32_t *target_lane;                 /* target_lane_index[2]; likewise        */
  int32_t *speed_index;                 /* MDPVehicle.speed_index (controlled vehicles) */
  int32_t *flags;                       /* HWY_F_* */
  double *time;                         /* AbstractEnv.time [E]                  abstract.py:274 */
  /* HWY_SCENARIO_INTERSECTION only (ABI v4; ignored / may be NULL otherwise): */
  int64_t *route;                       /* ControlledVehicle.route [E*N]: r_k << 5k for k < len (r_k = gnet index of the k-th
                                           remaining road, 5 bits each, len <= HWY_MAX_ROUTE) | len << 56
                                           (controller.py:71-87, road.py:98-106) */
  int32_t *road_steps;                  /* RegulatedRoad.steps [E] (regulation.py:36-40) */
} hwy_state;

typedef struct hwy_engine hwy_engine; /* opaque */

/* Library-level queries (usable without a GPU). */
int hwy_abi_version(void);
size_t hwy_config_size(void);            /* sizeof(hwy_config): lets a binding check its struct layout */
int hwy_device_count(void);              /* number of visible HIP devices, 0 if none / no driver */
const char *hwy_status_string(int status);

/*
 * Engine lifetime.  Replaces AbstractEnv.__init__ (abstract.py:60-89) for E
 * batched environments.  `stream` is an existing hipStream_t (e.g. PyTorch's
 * current stream) or NULL for an engine-owned stream.  The engine owns all
 * device memory.  Fails with HWY_ERR_NO_DEVICE when no GPU is present -- there
 * is no CPU fallback.
 */
int hwy_create(const hwy_config *cfg, int device, void *stream, hwy_engine **out);
int hwy_destroy(hwy_engine *eng);
const char *hwy_last_error(const hwy_engine *eng); /* eng may be NULL: last hwy_create failure */

/* Parity injection / inspection: H2D and D2H of the whole SoA (synchronous). */
int hwy_set_state(hwy_engine *eng, const hwy_state *host);
int hwy_get_state(hwy_engine *eng, hwy_state *host);

/*
 * Reset.  Replaces AbstractEnv.reset -> HighwayEnv._reset (abstract.py:219-249,
 * highway_env.py:55-98) for the environments whose mask byte is non-zero (NULL
 * mask = all), spawning traffic ON THE DEVICE with the reference's spawn rule
 * (Vehicle.create_random, kinematics.py:50-104) driven by a counter-based RNG
 * keyed by seeds[e].  The random stream is NOT numpy's PCG64 stream; the
 * stream-identical reset is host-side (highwayenv_amd/spawn.py) + hwy_set_state.
 * `ego_spacing`/`vehicles_density`/`initial_lane_id` (-1 = random) are the
 * config entries of the same names.  Writes the first observation if obs != NULL
 * (host pointer, [E][A][V][F]; rows of unmasked envs untouched).
 */
int hwy_reset(hwy_engine *eng, const uint8_t *mask, const uint64_t *seeds, double ego_spacing,
              double vehicles_density, int32_t initial_lane_id, float *obs);

/*
 * One batched policy step == AbstractEnv.step for every environment:
 * T x { action_type.act (first frame); road.act(); road.step(dt) } + observe +
 * reward + terminated + truncated + info{speed,crashed}.
 *   actions    int32 [E][A]   in   DiscreteMetaAction ids
 *   obs        f32   [E][A][V][F]
 *   reward     f64   [E][A]   (single-agent envs: the reference's scalar reward)
 *   terminated u8    [E]
 *   truncated  u8    [E]
 *   info_speed f64   [E][A], info_crashed u8 [E][A]    (may be NULL)
 *              info_crashed bit 0 = vehicle.crashed; the intersection scenario
 *              also sets bit 1 = has_arrived(vehicle) (intersection_env.py:340-345),
 *              so that agents_terminated = (byte != 0) (:119-121)
 * hwy_step takes HOST pointers (H2D actions, kernels, D2H results, synchronises).
 * hwy_step_device takes DEVICE pointers, only enqueues on the engine's stream
 * and does not synchronise -- the path for on-GPU policies, RCCL gathers and
 * the benchmark's HBM-resident timing.  It does NOT validate the action ids
 * (hwy_step does: HWY_ERR_ACTION): an id outside the configured table acts as
 * IDLE (HWY_ACTION_TO_ALL), where the reference raises KeyError.
 */
int hwy_step(hwy_engine *eng, const int32_t *actions, float *obs, double *reward,
             uint8_t *terminated, uint8_t *truncated, double *info_speed, uint8_t *info_crashed);
/*
 * k_steps consecutive policy steps with pre-staged actions (abstract.py:259-285, k times): open-loop rollouts, action repeat,
 * planners that score action sequences.  DEVICE pointers, block k of every plane belongs to step k:
 *   d_actions int32 [K][E][A], d_obs f32 [K][E][A][V][F], d_reward f64 [K][E][A], d_terminated / d_truncated u8 [K][E],
 *   d_info_speed f64 [K][E][A], d_info_crashed u8 [K][E][A] (the last two may be NULL).
 * Results are those of k_steps calls of hwy_step_device, bit for bit (auto-reset included: an environment that ends in step k
 * is re-spawned in step k + 1 when hwy_set_autoreset is on).  The k steps run in ONE launch of the scenario's multi-step kernel:
 * the dispatch and, above all, the wait for the slowest SIMD at the end of every step are paid once per call.  (Intersection: the
 * launch holds no pre-warming blocks, an environment that ends in it prepares its next episode inline -- same results.)  Enqueues
 * on the engine's stream, does not synchronise, does not validate action ids (see hwy_step_device).
 * HWY_ERR_INVALID_ARG for k_steps > 1 on an intersection engine configured with HWY_C_HOST_TRAFFIC: there the host runs
 * _clear_vehicles / _spawn_vehicle between policy steps (intersection_env.py:199-203), which a K-step launch would skip.
 */
int hwy_rollout_device(hwy_engine *eng, int32_t k_steps, const int32_t *d_actions, float *d_obs, double *d_reward,
                       uint8_t *d_terminated, uint8_t *d_truncated, double *d_info_speed, uint8_t *d_info_crashed);
/* The same through HOST pointers (validates the action ids like hwy_step: HWY_ERR_ACTION; H2D, launch(es), D2H, synchronises). */
int hwy_rollout(hwy_engine *eng, int32_t k_steps, const int32_t *actions, float *obs, double *reward,
                uint8_t *terminated, uint8_t *truncated, double *info_speed, uint8_t *info_crashed);
int hwy_step_device(hwy_engine *eng, const int32_t *d_actions, float *d_obs, double *d_reward,
                    uint8_t *d_terminated, uint8_t *d_truncated, double *d_info_speed,
                    uint8_t *d_info_crashed);

/*
 * Debug / parity: advance `n_frames` simulation frames (Road.act + Road.step)
 * without observing.  If actions != NULL (host, [E][A]) the meta-action is
 * applied on the first of those frames (abstract.py:294-304).  `time` is not
 * advanced.
 */
int hwy_step_frames(hwy_engine *eng, const int32_t *actions, int32_t n_frames);

/* KinematicObservation.observe for the current state (host pointer out). */
int hwy_observe(hwy_engine *eng, float *obs);

/*
 * Auto-reset (gymnasium vector "next-step" mode): when enabled, an environment
 * that returned terminated|truncated is re-spawned on the device at the start
 * of the following hwy_step* call (its action for that step is ignored and the
 * returned obs is the reset obs, reward 0).  seeds advance per episode.
 */
int hwy_set_autoreset(hwy_engine *eng, int32_t enabled, uint64_t base_seed, double ego_spacing,
                      double vehicles_density, int32_t initial_lane_id);

int hwy_sync(hwy_engine *eng); /* hipStreamSynchronize on the engine stream */

/*
 * Event counters of an engine since creation (or since the last call with reset != 0); synchronises the stream.
 * HWY_SCENARIO_INTERSECTION, device traffic mode: IntersectionEnv._spawn_vehicle (intersection_env.py:324-352) appends to
 * an unbounded list; the engine has hwy_config.num_vehicles slots per environment and DROPS a spawn that finds them all
 * taken.  HWY_CTR_IX_SPAWNS counts the spawns performed, HWY_CTR_IX_SPAWNS_DROPPED the ones dropped (initial traffic,
 * the per-step spawn and the pre-warmed next episodes alike), so a caller can size max_vehicles until the second is 0.
 * `out` receives min(n, HWY_CTR_COUNT) values.
 */
/* HWY_CTR_NONFINITE_STORES (every scenario): vehicles written back by a step / frames launch with a non-finite position,
 * heading or speed -- 0 in any healthy run; a NaN handed in through hwy_set_state (or produced by a defect) shows up here. */
enum { HWY_CTR_IX_SPAWNS = 0, HWY_CTR_IX_SPAWNS_DROPPED = 1, HWY_CTR_NONFINITE_STORES = 2, HWY_CTR_COUNT = 8 };
int hwy_get_counters(hwy_engine *eng, uint64_t *out, int32_t n, int32_t reset);

/*
 * Placement of the environments on the GPU's SIMDs (tuning; never changes a result).  Environments are independent
 * (the reference steps them in separate processes: scripts/sb3_highway_ppo.py:16-18), so WHICH workgroup of a launch steps
 * environment e is free: with env_of_block[b] = e (host pointer, a permutation of 0 .. num_envs - 1) workgroup b of every
 * following hwy_step* / hwy_step_device launch steps environment e; all inputs and outputs stay indexed by environment.
 * NULL restores the identity.  One-wavefront step kernel only (HWY_SCENARIO_HIGHWAY, num_vehicles <= 64); synchronises.
 */
int hwy_set_block_order(hwy_engine *eng, const int32_t *env_of_block);

/*
 * Self-test hook: evaluate one of the step kernel's own math routines (csrc/hwy_math.h -- bounded-domain
 * log / exp / sincos / asin, Newton-refined v_rcp_f64 / v_rsq_f64, floor-mod angle wrap) on n doubles on
 * the device.  Host pointers.  op: 0 log_pos, 1 exp_bounded, 2 sin, 3 cos, 4 asin_bounded, 5 fast_rcp, 8 atan_fd,
 * 9 atan2_bounded(x, 0.75), 10 atan2_bounded(0.5, x), 11 atan2_bounded(-0.5, x),
 * 6 fast_rsqrt, 7 wrap_to_pi; 20 .. 33: the paired forms (log_pos2, exp_bounded2, sincos_bounded2, asin_bounded2, fast_rcp2,
 * fast_rsqrt2: first / second result, see math_probe in csrc/hwy_device.h), which must equal the scalar ones bit for bit.
 * 40 / 41: the collision walk's reach bound (hwy_device.h: the wavefront's maximum of reach_key -- call with whole wavefronts, n a
 * multiple of 64 --, and the double a key is rounded up to).
 * Lets the accuracy claims (<= 2 ulp on the stated domains) be checked on the GPU.
 */
int hwy_debug_math(hwy_engine *eng, int32_t op, const double *in, double *out, int64_t n);

/*
 * Kernel timing of the step kernel of hwy_step / hwy_step_device / hwy_step_frames calls with HIP events on the
 * engine's stream: the launch (hipExtLaunchKernelGGL) records the DISPATCH's own begin and end timestamps into
 * the pair, i.e. what rocprofv3 --kernel-trace reports for it (events recorded around the launch with
 * hipEventRecord also measure ~3 us of command processing).  `enabled`: 0 = off, k > 0 = time every k-th
 * launch.  hwy_profile_read synchronises and returns the accumulated kernel time and the number of TIMED
 * launches since the last hwy_profile_enable(eng, k > 0).
 */
int hwy_profile_enable(hwy_engine *eng, int32_t enabled);
int hwy_profile_read(hwy_engine *eng, double *total_ms, int64_t *launches);

/*
 * The issue-priority turn in use (the encoding of hwy_config.tune_prio_shift: 0 = none, 1..30 = 2^k clock ticks, >= 64 = k x 64
 * ticks) and the state of the engine's own selection: with tune_prio_shift == 0 and a scenario whose wavefronts take turns, the
 * engine times its first 85 full-step launches per stage (five turn lengths around the scenario default, interleaved; the dispatch
 * timestamps of hwy_profile_*) and keeps the fastest -- scheduling only, no result depends on it.  state: 0 = no selection (explicit
 * value, turns off, or a kernel without turns), 1 = still sampling, 2 = chosen.  The reference has no counterpart (a CPU
 * single-thread loop: envs/common/abstract.py:287-317); this is the knob a `configure()`d shape other than BASELINE's needs
 * (envs/highway_env.py:25-53).
 */
int hwy_get_prio_turn(hwy_engine *eng, int32_t *turn, int32_t *state);

/*
 * Multi-GPU, one process (and one engine) per GPU.  The reference's only vectorisation is gymnasium's process-level
 * vector env (tests/envs/test_gym.py:158-165); here the environments are block-partitioned over the engines of a node,
 * nothing is exchanged while stepping, and the per-step (obs | reward | done) blocks of every rank are gathered to a root
 * rank with ONE RCCL collective over xGMI (ncclGather, rccl.h:745), enqueued on the engine's stream.
 *   hwy_comm_unique_id  rank 0 only; the caller ships the HWY_COMM_ID_BYTES bytes to every other rank (MPI, a TCP store, ...)
 *   hwy_comm_init       collective over all `world` ranks; librccl is loaded on first use (HWY_ERR_UNSUPPORTED if absent)
 *   hwy_gather          `bytes` bytes from device pointer d_send of every rank -> d_recv[rank * bytes ...] on `root`
 *                       (d_recv is ignored on the other ranks); asynchronous: hwy_sync or stream order makes it visible
 * A caller that keeps its buffers in a framework (PyTorch) may use that framework's collectives instead
 * (highwayenv_amd/dist.py does, bench.py --comm abi uses these entry points).
 */
#define HWY_COMM_ID_BYTES 128
int hwy_comm_unique_id(uint8_t *id);
int hwy_comm_init(hwy_engine *eng, const uint8_t *id, int32_t rank, int32_t world);
int hwy_gather(hwy_engine *eng, const void *d_send, void *d_recv, size_t bytes, int32_t root);
int hwy_comm_destroy(hwy_engine *eng);

#ifdef __cplusplus
}
#endif
#endif /* HWY_ENGINE_H */

// hwy_engine.hip -- C-ABI host side of the MI355X HighwayEnv step engine (include/hwy_engine.h).
//
// Owns the device-resident struct-of-arrays of E environments x N vehicles, one HIP stream,
// pinned staging buffers for the host-pointer entry points, and HIP-event kernel timing.
// There is deliberately NO CPU implementation behind these entry points: without a GPU
// hwy_create fails with HWY_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/hwy_engine.h"
#include "hwy_comm.h"
#include "hwy_launch.h"
#include "hwy_params.h"

using hwy::StepParams;

struct hwy_engine {
  hwy_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int pitch = 0;
  bool force_block_kernel = false;  // hwy_config.tune_block_kernel: use the generic workgroup kernel even for N <= 64
  int waves_per_eu = 3;  // register-allocation variant of the step kernel (hwy_config.tune_waves_per_eu)
  int rollout_waves_per_eu = 3;  // ... and of the multi-step kernel (hwy_rollout_device)
  int prio_shift = 0;    // issue-priority turns of the step kernels (hwy_config.tune_prio_shift; 0 = off)
  // device state
  double *d_f64 = nullptr;   // 9 fields x E x pitch
  int32_t *d_packed = nullptr;
  long long *d_route = nullptr;     // intersection scenario: planned routes [E x pitch] (64-bit route words)
  int32_t *d_road_steps = nullptr;  // intersection scenario: RegulatedRoad.steps [E]
  hwy_glane *d_gnet = nullptr;      // intersection scenario: lane table
  double *d_shadow_f64 = nullptr;   // intersection scenario, pre-warming of next episodes: second copy of the planes
  int32_t *d_shadow_packed = nullptr, *d_shadow_meta = nullptr;
  long long *d_shadow_route = nullptr;
  unsigned long long *d_counters = nullptr;  // [HWY_CTR_COUNT] (hwy_get_counters)
  uint16_t *d_block_env = nullptr;           // [E] hwy_set_block_order (nullptr: workgroup b steps environment b)
  double *d_time = nullptr;
  uint8_t *d_done = nullptr;
  uint32_t *d_episode = nullptr;
  // device I/O buffers for the host-pointer entry points
  int32_t *d_actions = nullptr;
  // step outputs of the host-pointer entry points live in ONE device block (reward | speed | obs | term | trunc |
  // crashed) mirrored by one pinned host block, so that hwy_step needs a single D2H copy
  char *d_out = nullptr;
  char *d_roll = nullptr;  // hwy_rollout (host pointers): K blocks of actions + outputs, grown on demand
  size_t roll_bytes = 0;
  size_t out_bytes = 0, off_reward = 0, off_speed = 0, off_obs = 0, off_term = 0, off_trunc = 0, off_crashed = 0;
  float *d_obs = nullptr;
  double *d_reward = nullptr, *d_info_speed = nullptr;
  uint8_t *d_term = nullptr, *d_trunc = nullptr, *d_info_crashed = nullptr;
  uint8_t *d_mask = nullptr;
  uint64_t *d_seeds = nullptr;
  int32_t *d_grid_ws = nullptr;  // OccupancyGrid workspace [E][A][2][W*H]
  // pinned host staging
  void *h_pinned = nullptr;
  size_t h_pinned_bytes = 0;
  // auto-reset
  int autoreset = 0;
  hwy::ResetParams rp{};
  // profiling
  int profiling = 0;        // 0 = off, k > 0 = HIP events around every k-th step-kernel launch
  int64_t launch_counter = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  double prof_ms = 0.0;
  int64_t prof_launches = 0;
  // self-selection of the issue-priority turn (hwy_config.tune_prio_shift == 0 and turns apply): the first full-step launches are
  // timed with the dispatch's own timestamps, the candidates interleaved (drift of the workload cancels), the best one is kept
  struct TurnTuner {
    enum { IDLE = 0, SAMPLING = 1, DONE = 2 };
    int state = IDLE, stage = 0, launches = 0;
    int cand[5] = {0, 0, 0, 0, 0};
    double ms[5] = {0, 0, 0, 0, 0};
    int n[5] = {0, 0, 0, 0, 0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<int> which;
  } tuner;
  hwy::Comm *comm = nullptr;  // hwy_comm_init
  std::string err;
};

static thread_local std::string g_create_error;

#define HWY_HIP(eng, call)                                                                        \
  do {                                                                                            \
    hipError_t _e = (call);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      (eng)->err = std::string(#call) + ": " + hipGetErrorString(_e);                             \
      return HWY_ERR_HIP;                                                                         \
    }                                                                                             \
  } while (0)

static int fail(hwy_engine *eng, int code, const std::string &msg) {
  if (eng) eng->err = msg; else g_create_error = msg;
  return code;
}

// ---- the engine picks the length of an issue-priority turn itself ---------------------------------------------------------------
// Scheduling only: no result depends on the turn (tests/test_engine_parity.py compares turn variants bit for bit).
// Candidates: 0.75 / 0.875 / 1 / 1.125 / 1.25 x the scenario default, in units of 64 clock ticks (the linear encoding of
// tune_prio_shift), visited round-robin over the first (HWY_TUNE_ROUNDS + 1) x 5 full-step launches, each timed by the dispatch's own
// begin / end timestamps (hipExtLaunchKernelGGL: the events of hwy_profile_*); one synchronisation at the end of a stage.  If the
// best candidate sits at an end of the range, one more stage is centred there (at most HWY_TUNE_STAGES).  While hwy_profile_enable
// is on the tuner pauses.
#define HWY_TUNE_ROUNDS 16
#define HWY_TUNE_STAGES 3
static int turn_in_ticks64(int turn) {  // the linear unit of either encoding (0 = not expressible: a turn below 64 x 64 ticks)
  if (turn >= 64) return turn;
  if (turn >= 12 && turn <= 26) return 1 << (turn - 6);
  return 0;
}
static void tuner_arm(hwy_engine *eng, int centre_turn) {
  auto &t = eng->tuner;
  const int k = turn_in_ticks64(centre_turn);
  if (k < 64) { t.state = hwy_engine::TurnTuner::DONE; return; }
  static const int eighths[5] = {6, 7, 8, 9, 10};
  for (int c = 0; c < 5; ++c) {
    t.cand[c] = std::min(std::max(64, k * eighths[c] / 8), 1 << 20);
    t.ms[c] = 0.0;
    t.n[c] = 0;
  }
  t.launches = 0;
  t.which.clear();
  t.state = hwy_engine::TurnTuner::SAMPLING;
}
static int tuner_finish_stage(hwy_engine *eng) {
  auto &t = eng->tuner;
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  for (size_t k = 0; k < t.which.size(); ++k) {
    float ms = 0;
    HWY_HIP(eng, hipEventElapsedTime(&ms, t.events[k].first, t.events[k].second));
    if (k < 5) continue;  // the first round: every candidate's first launch follows another turn's launch pattern
    t.ms[t.which[k]] += ms;
    t.n[t.which[k]]++;
  }
  int best = 2;
  for (int c = 0; c < 5; ++c)
    if (t.n[c] > 0 && t.n[best] > 0 && t.ms[c] / t.n[c] < t.ms[best] / t.n[best]) best = c;
  eng->prio_shift = t.cand[best];
  const bool at_end = (best == 0 && t.cand[0] > 64) || (best == 4 && t.cand[4] < (1 << 20));
  if (at_end && ++t.stage < HWY_TUNE_STAGES) tuner_arm(eng, t.cand[best]);
  else t.state = hwy_engine::TurnTuner::DONE;
  return HWY_OK;
}

extern "C" int hwy_abi_version(void) { return HWY_ABI_VERSION; }
extern "C" size_t hwy_config_size(void) { return sizeof(hwy_config); }
extern "C" int hwy_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" const char *hwy_status_string(int status) {
  switch (status) {
    case HWY_OK: return "ok";
    case HWY_ERR_INVALID_ARG: return "invalid argument";
    case HWY_ERR_HIP: return "HIP runtime error";
    case HWY_ERR_UNSUPPORTED: return "unsupported configuration";
    case HWY_ERR_NO_DEVICE: return "no MI355X / HIP device available (there is no CPU fallback)";
    case HWY_ERR_ACTION: return "meta-action outside the configured action table";
    default: return "unknown status";
  }
}
extern "C" const char *hwy_last_error(const hwy_engine *eng) {
  return eng ? eng->err.c_str() : g_create_error.c_str();
}

static int validate(const hwy_config *c, std::string &why) {
  char buf[256];
#define BAD(...) do { snprintf(buf, sizeof buf, __VA_ARGS__); why = buf; return HWY_ERR_INVALID_ARG; } while (0)
  if (!c) BAD("config is NULL");
  if (c->abi_version != HWY_ABI_VERSION) BAD("abi_version %d != %d", c->abi_version, HWY_ABI_VERSION);
  if (c->num_envs < 1) BAD("num_envs must be >= 1");
  if (c->num_vehicles < 1 || c->num_vehicles > HWY_MAX_VEHICLES) BAD("num_vehicles must be in [1,%d]", HWY_MAX_VEHICLES);
  if (c->num_agents < 1 || c->num_agents > HWY_MAX_AGENTS || c->num_agents > c->num_vehicles) BAD("num_agents out of range");
  for (int a = 0; a < c->num_agents; ++a)
    if (c->agent_index[a] < 0 || c->agent_index[a] >= c->num_vehicles) BAD("agent_index[%d] out of range", a);
  if (c->lanes_count < 1 || c->lanes_count > HWY_MAX_LANES) BAD("lanes_count must be in [1,%d]", HWY_MAX_LANES);
  if (c->frames_per_step < 0) BAD("frames_per_step must be >= 0");
  if (c->action_set < HWY_ACTIONS_ALL || c->action_set > HWY_ACTIONS_LAT) BAD("action_set must be HWY_ACTIONS_ALL / _LONGI / _LAT");
  if (c->tune_extra_lds < 0 || c->tune_extra_lds > 65536) BAD("tune_extra_lds must be in [0,65536]");
  if (c->tune_ix_prewarm_frames < 0) BAD("tune_ix_prewarm_frames must be >= 0");
  if (c->tune_waves_per_eu < 0 || c->tune_waves_per_eu > 4) BAD("tune_waves_per_eu must be in [0,4]");
  if (c->tune_block_kernel < 0 || c->tune_block_kernel > 2) BAD("tune_block_kernel must be 0 (engine's choice), 1 (workgroup kernel) or 2 (one-wavefront kernels)");
  if (c->tune_prio_shift < -1 || (c->tune_prio_shift > 30 && c->tune_prio_shift < 64) || c->tune_prio_shift > (1 << 20))
    BAD("tune_prio_shift must be -1 (off), 0 (engine's choice), 1..30 (a turn of 2^k clock ticks) or 64..2^20 (a turn of k x 64 ticks)");
  if (c->obs_vehicles < 1 || c->obs_vehicles > c->num_vehicles + 64) BAD("obs_vehicles out of range");
  if (c->obs_type != HWY_OBS_KINEMATICS && c->obs_type != HWY_OBS_OCCUPANCY_GRID) BAD("unknown obs_type");
  if (c->obs_type == HWY_OBS_OCCUPANCY_GRID) {
    if (c->grid_shape[0] < 1 || c->grid_shape[1] < 1 || (int64_t)c->grid_shape[0] * c->grid_shape[1] > HWY_MAX_GRID_CELLS)
      BAD("grid_shape must hold 1..%d cells", HWY_MAX_GRID_CELLS);
    if (!(c->grid_step[0] > 0) || !(c->grid_step[1] > 0)) BAD("grid_step must be positive");
  }
  if (c->obs_features < 1 || c->obs_features > HWY_MAX_FEATURES) BAD("obs_features out of range");
  for (int f = 0; f < c->obs_features; ++f)
    if (c->obs_feature_ids[f] < 0 || c->obs_feature_ids[f] >= HWY_FEAT_COUNT ||
        (c->obs_feature_ids[f] == HWY_FEAT_ON_ROAD && c->obs_type != HWY_OBS_OCCUPANCY_GRID)) BAD("unknown feature id");
  if (c->num_target_speeds < 2 || c->num_target_speeds > HWY_MAX_TARGET_SPEEDS) BAD("num_target_speeds must be in [2,%d]", HWY_MAX_TARGET_SPEEDS);
  if (!(c->dt > 0) || !(c->policy_dt > 0)) BAD("dt and policy_dt must be positive");
  if (!(c->lane_width > 0) || !(c->road_length > 0)) BAD("lane_width and road_length must be positive");
  if (c->scenario == HWY_SCENARIO_INTERSECTION) {
    if (c->num_agents > 4) BAD("the intersection scenario holds 1..4 controlled vehicles (one per access road)");
    if (c->num_vehicles < 4 || c->num_vehicles > 64) BAD("the intersection scenario needs 4..64 slots (one wavefront per environment)");
    if (c->gnet_lanes < 1 || c->gnet_lanes > HWY_MAX_GLANES) BAD("gnet_lanes must be in [1,%d]", HWY_MAX_GLANES);
    if (c->destination < -1 || c->destination > 3) BAD("destination must be the k of \"o\" + k, 0..3, or -1 for a random one");
    if (c->initial_vehicle_count < 1) BAD("initial_vehicle_count must be positive");
    for (int k = 0; k < 4; ++k)
      if (c->access_lane[k] < 0 || c->access_lane[k] >= c->gnet_lanes || c->exit_of[k] < 0 || c->exit_of[k] >= c->gnet_lanes)
        BAD("access_lane / exit_of out of range");
    for (int k = 0; k < c->gnet_lanes; ++k) {
      const hwy_glane &l = c->gnet[k];
      if (!(l.length > 0) || !(l.width > 0)) BAD("gnet[%d]: length and width must be positive", k);
      if (l.kind == 1 && !(l.radius > 0)) BAD("gnet[%d]: radius must be positive", k);
      for (int q = 0; q < k; ++q)
        if (c->gnet[q].from_node == l.from_node && c->gnet[q].to_node == l.to_node) BAD("gnet[%d]: every road holds one lane", k);
    }
  } else if (c->scenario != HWY_SCENARIO_HIGHWAY) {
    if (c->scenario != HWY_SCENARIO_MERGE && c->scenario != HWY_SCENARIO_MERGE_GENERIC) BAD("unknown scenario %d", c->scenario);
    if (c->num_vehicles < 3 || c->num_vehicles > 64) BAD("road-network scenarios need 3..64 slots (one wavefront per environment)");
    if (c->net_lanes < 1 || c->net_lanes > HWY_MAX_LANES) BAD("net_lanes must be in [1,%d]", HWY_MAX_LANES);
    if (c->merge_lane >= c->net_lanes) BAD("merge_lane out of range");
    if (c->net_lanes < 3 * c->lanes_count + 3) BAD("merge networks hold 3*lanes_count + 3 lanes");
    for (int a = 0; a < c->num_agents; ++a)
      if (c->agent_index[a] != a) BAD("road-network scenarios keep agent a in slot a");
    for (int k = 0; k < c->net_lanes; ++k) {
      const hwy_lane &l = c->net[k];
      if (!(l.length > 0) || !(l.width > 0)) BAD("net[%d]: length and width must be positive", k);
      if (l.road_first < 0 || l.road_lanes < 1 || l.road_first + l.road_lanes > c->net_lanes || l.id != k - l.road_first ||
          l.id >= l.road_lanes) BAD("net[%d]: inconsistent road_first / road_lanes / id", k);
      if (l.next_first >= 0 && (l.next_lanes < 1 || l.next_first + l.next_lanes > c->net_lanes)) BAD("net[%d]: successor road out of range", k);
    }
  }
#undef BAD
  return HWY_OK;
}

static void fill_params(const hwy_engine *eng, StepParams &p) {
  hwy::params_from_config(eng->cfg, eng->pitch, p);
  hwy::bind_planes(eng->d_f64, (size_t)eng->cfg.num_envs * eng->pitch, p.st);
  p.st.packed = eng->d_packed; p.st.time = eng->d_time; p.st.done = eng->d_done; p.st.episode = eng->d_episode;
  p.autoreset = eng->autoreset;
  p.rp = eng->rp;
  p.grid_ws = eng->d_grid_ws;
  hwy::set_prio_turn(p, eng->prio_shift);
  p.block_env = eng->d_block_env;
  p.counters = eng->d_counters;
}

static bool is_ix(const hwy_engine *eng) { return eng->cfg.scenario == HWY_SCENARIO_INTERSECTION; }
static bool is_net(const hwy_engine *eng) { return eng->cfg.scenario != HWY_SCENARIO_HIGHWAY && !is_ix(eng); }
static void fill_ix(const hwy_engine *eng, const StepParams &p, hwy::IxParams &ip) {
  hwy::ix_params_from_config(eng->cfg, p, ip);
  ip.lanes = eng->d_gnet;
  ip.route = eng->d_route;
  ip.counters = eng->d_counters;
  ip.road_steps = eng->d_road_steps;
  if (eng->d_shadow_meta) {
    hwy::bind_planes(eng->d_shadow_f64, (size_t)eng->cfg.num_envs * eng->pitch, ip.shadow);
    ip.shadow.packed = eng->d_shadow_packed;
    ip.shadow_route = eng->d_shadow_route;
    ip.shadow_meta = eng->d_shadow_meta;
  }
}
// forget every pre-warmed episode (seeds / episode counters / the state they would follow have changed)
static hipError_t invalidate_shadows(hwy_engine *eng) {
  if (!eng->d_shadow_meta) return hipSuccess;
  return hipMemsetAsync(eng->d_shadow_meta, 0xff, (size_t)eng->cfg.num_envs * 4 * sizeof(int32_t), eng->stream);
}
static hipError_t launch_step_any(const hwy_engine *eng, const StepParams &p) {
  if (is_ix(eng)) {
    hwy::IxParams ip;
    fill_ix(eng, p, ip);
    return hwy::launch_ix_step(ip, eng->cfg.num_envs, eng->stream, eng->waves_per_eu);
  }
  if (is_net(eng)) {
    hwy::NetParams np;
    hwy::net_params_from_config(eng->cfg, p, np);
    return hwy::launch_net_step(np, eng->cfg.num_envs, eng->stream, eng->waves_per_eu);
  }
  return hwy::launch_step(p, eng->cfg.num_envs, eng->stream, eng->waves_per_eu, eng->force_block_kernel,
                          eng->cfg.tune_extra_lds);
}
static hipError_t launch_reset_any(const hwy_engine *eng, const StepParams &p) {
  if (is_ix(eng)) {
    hwy::IxParams ip;
    fill_ix(eng, p, ip);
    return hwy::launch_ix_reset(ip, eng->cfg.num_envs, eng->stream);
  }
  if (is_net(eng)) {
    hwy::NetParams np;
    hwy::net_params_from_config(eng->cfg, p, np);
    return hwy::launch_net_reset(np, eng->cfg.num_envs, eng->stream);
  }
  return hwy::launch_reset(p, eng->cfg.num_envs, eng->stream);
}
static hipError_t launch_observe_any(const hwy_engine *eng, const StepParams &p) {
  if (is_ix(eng)) {
    hwy::IxParams ip;
    fill_ix(eng, p, ip);
    return hwy::launch_ix_observe(ip, eng->cfg.num_envs, eng->stream);
  }
  if (is_net(eng)) {
    hwy::NetParams np;
    hwy::net_params_from_config(eng->cfg, p, np);
    return hwy::launch_net_observe(np, eng->cfg.num_envs, eng->stream);
  }
  return hwy::launch_observe(p, eng->cfg.num_envs, eng->stream);
}

static size_t io_counts(const hwy_config &c, size_t *n_act, size_t *n_obs, size_t *n_ea) {
  *n_act = (size_t)c.num_envs * c.num_agents;
  *n_obs = *n_act * hwy::obs_len(c);
  *n_ea = *n_act;
  return 0;
}

extern "C" int hwy_create(const hwy_config *cfg, int device, void *stream, hwy_engine **out) {
  if (!out) return fail(nullptr, HWY_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  std::string why;
  if (int rc = validate(cfg, why)) return fail(nullptr, rc, why);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(nullptr, HWY_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(nullptr, HWY_ERR_INVALID_ARG, "device index out of range");
  hwy_engine *eng = new (std::nothrow) hwy_engine();
  if (!eng) return fail(nullptr, HWY_ERR_HIP, "out of host memory");
  eng->cfg = *cfg;
  eng->device = device;
  eng->pitch = (cfg->num_vehicles + 7) & ~7;  // 64-byte aligned rows of f64
  // 0 = the engine's choice: the wide kernel (hwy_wave2.h) for 64 < N <= 128, the workgroup kernel beyond -- three / four vehicles
  // per thread are one 338 / 442-VGPR wavefront per SIMD and measured slower there (1024 x 201: 312 us against 240,
  // profiles/r05_history.md); 1 = the workgroup kernel wherever it exists; 2 = the wide kernel wherever it exists (N <= 256)
  eng->force_block_kernel = cfg->tune_block_kernel == 1 || (cfg->tune_block_kernel == 0 && cfg->num_vehicles > 128);
  // road-network kernel: 128 VGPRs, 4 waves/SIMD, no spills.
  // intersection kernel with helper lanes (N <= 32, hwy_ix.h): 150 VGPRs, but 20.2 KB of LDS per one-wavefront workgroup keep it
  // at 2 per SIMD.  Without them (N > 32, or tune_ix_no_helpers): 128 VGPRs / 16.7 KB (2048 x 30: 371.9 us against 285.1)
  if (cfg->scenario == HWY_SCENARIO_INTERSECTION) {
    const bool helpers = cfg->num_vehicles <= 32 && !cfg->tune_ix_no_helpers;
    eng->waves_per_eu = (cfg->num_envs > 2048 && !helpers) ? 3 : 2;
  }
  else if (cfg->scenario != HWY_SCENARIO_HIGHWAY) eng->waves_per_eu = 4;
  else {
    // highway scenario.  One-wavefront kernels (N <= 64; 102 VGPRs ego-only, 128 full-pairwise: every allocation variant is the
    // same code since the build stopped hoisting literals, build.py).  Workgroup kernel (N > 64, ceil(N / 64) wavefronts per
    // environment): 146 .. 154 VGPRs at 3 waves/SIMD, 128 with 18 .. 30 spilled at 4 -- the 4-wave build pays as soon as the grid
    // no longer fits 3 resident wavefronts per SIMD (1024 x 101: 159.2 / 164.4 us; 2048 x 101: 273.2 / 209.6 us)
    hipDeviceProp_t prop;
    const int simds = (hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256) * 4;
    const long long waves = (long long)cfg->num_envs * ((cfg->num_vehicles + 63) / 64);
    eng->waves_per_eu = waves > 3LL * simds ? 4 : 3;
    // (64 < N <= 128 with the Kinematics observation runs the wide kernel, hwy_wave2.h: ONE build, <2, 2> -- 245 VGPRs, two
    //  resident wavefronts per SIMD --, which this variable and hwy_config.tune_waves_per_eu do not select among)
  }
  if (cfg->tune_waves_per_eu >= 1 && cfg->tune_waves_per_eu <= 4) eng->waves_per_eu = cfg->tune_waves_per_eu;
  eng->rollout_waves_per_eu = eng->waves_per_eu;
  auto bail = [&](hipError_t e, const char *what) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(e);
    hwy_destroy(eng);
    return HWY_ERR_HIP;
  };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
  if (stream) {
    eng->stream = (hipStream_t)stream;
  } else {
    if ((e = hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    eng->own_stream = true;
  }
  const size_t E = cfg->num_envs, plane = E * eng->pitch;
  size_t n_act, n_obs, n_ea;
  io_counts(*cfg, &n_act, &n_obs, &n_ea);
#define ALLOC(ptr, bytes) if ((e = hipMalloc((void **)&(ptr), (bytes))) != hipSuccess) return bail(e, "hipMalloc " #ptr)
  ALLOC(eng->d_f64, plane * 9 * sizeof(double));
  ALLOC(eng->d_packed, plane * sizeof(int32_t));
  if (cfg->scenario == HWY_SCENARIO_INTERSECTION) {
    ALLOC(eng->d_route, plane * sizeof(long long));
    ALLOC(eng->d_road_steps, E * sizeof(int32_t));
    ALLOC(eng->d_gnet, sizeof(hwy_glane) * HWY_MAX_GLANES);
    if ((e = hipMemsetAsync(eng->d_route, 0, plane * sizeof(long long), eng->stream)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipMemsetAsync(eng->d_road_steps, 0, E * sizeof(int32_t), eng->stream)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipMemcpy(eng->d_gnet, cfg->gnet, sizeof(hwy_glane) * HWY_MAX_GLANES, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    if (!(cfg->flags & HWY_C_HOST_TRAFFIC) && !cfg->tune_ix_no_prewarm) {  // (tuning: every auto-reset runs its warm-up inline)
      ALLOC(eng->d_shadow_f64, plane * 9 * sizeof(double));
      ALLOC(eng->d_shadow_packed, plane * sizeof(int32_t));
      ALLOC(eng->d_shadow_route, plane * sizeof(long long));
      ALLOC(eng->d_shadow_meta, E * 4 * sizeof(int32_t));
      if ((e = hipMemsetAsync(eng->d_shadow_f64, 0, plane * 9 * sizeof(double), eng->stream)) != hipSuccess) return bail(e, "hipMemset");
      if ((e = hipMemsetAsync(eng->d_shadow_meta, 0xff, E * 4 * sizeof(int32_t), eng->stream)) != hipSuccess) return bail(e, "hipMemset");
    }
  }
  ALLOC(eng->d_counters, HWY_CTR_COUNT * sizeof(unsigned long long));
  if ((e = hipMemsetAsync(eng->d_counters, 0, HWY_CTR_COUNT * sizeof(unsigned long long), eng->stream)) != hipSuccess) return bail(e, "hipMemset");
  ALLOC(eng->d_time, E * sizeof(double));
  ALLOC(eng->d_done, E);
  ALLOC(eng->d_episode, E * sizeof(uint32_t));
  ALLOC(eng->d_actions, n_act * sizeof(int32_t));
  {
    auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
    size_t off = 0;
    eng->off_reward = off;  off = up(off + n_ea * sizeof(double));
    eng->off_speed = off;   off = up(off + n_ea * sizeof(double));
    eng->off_obs = off;     off = up(off + n_obs * sizeof(float));
    eng->off_term = off;    off = up(off + E);
    eng->off_trunc = off;   off = up(off + E);
    eng->off_crashed = off; off = up(off + n_ea);
    eng->out_bytes = off;
  }
  ALLOC(eng->d_out, eng->out_bytes);
  eng->d_reward = (double *)(eng->d_out + eng->off_reward);
  eng->d_info_speed = (double *)(eng->d_out + eng->off_speed);
  eng->d_obs = (float *)(eng->d_out + eng->off_obs);
  eng->d_term = (uint8_t *)(eng->d_out + eng->off_term);
  eng->d_trunc = (uint8_t *)(eng->d_out + eng->off_trunc);
  eng->d_info_crashed = (uint8_t *)(eng->d_out + eng->off_crashed);
  ALLOC(eng->d_mask, E);
  ALLOC(eng->d_seeds, E * sizeof(uint64_t));
  if (cfg->obs_type == HWY_OBS_OCCUPANCY_GRID)
    ALLOC(eng->d_grid_ws, n_act * 2 * (size_t)cfg->grid_shape[0] * cfg->grid_shape[1] * sizeof(int32_t));
#undef ALLOC
  if ((e = hipMemsetAsync(eng->d_f64, 0, plane * 9 * sizeof(double), eng->stream)) != hipSuccess) return bail(e, "hipMemset");
  if ((e = hipMemsetAsync(eng->d_packed, 0, plane * sizeof(int32_t), eng->stream)) != hipSuccess) return bail(e, "hipMemset");
  if ((e = hipMemsetAsync(eng->d_time, 0, E * sizeof(double), eng->stream)) != hipSuccess) return bail(e, "hipMemset");
  if ((e = hipMemsetAsync(eng->d_done, 0, E, eng->stream)) != hipSuccess) return bail(e, "hipMemset");
  if ((e = hipMemsetAsync(eng->d_episode, 0, E * sizeof(uint32_t), eng->stream)) != hipSuccess) return bail(e, "hipMemset");
  // pinned staging: the largest of {state SoA, step I/O}
  const size_t state_bytes = plane * (9 * sizeof(double) + sizeof(int32_t) + sizeof(long long)) + E * (sizeof(double) + sizeof(int32_t));
  const size_t io_bytes = eng->out_bytes + n_act * 4 + E * 9 + 64;
  eng->h_pinned_bytes = state_bytes > io_bytes ? state_bytes : io_bytes;
  if ((e = hipHostMalloc(&eng->h_pinned, eng->h_pinned_bytes, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
  if ((e = hipStreamSynchronize(eng->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
  // default reset parameters: highway-v0 semantics (every vehicle checks collisions)
  eng->rp.ego_spacing = 2.0;
  eng->rp.other_spacing = 1.0;
  eng->rp.lane_factor = std::exp(-5.0 / 40.0 * cfg->lanes_count);
  eng->rp.initial_lane_id = -1;
  eng->rp.fast = (cfg->flags & HWY_C_EGO_ONLY_COLLISIONS) ? 1 : 0;  // HighwayEnvFast (highway_env.py:177-182)
  eng->rp.base_seed = 0;
  // issue-priority turns: on by default only where the whole grid of the step kernel is resident at once
  if (cfg->tune_prio_shift > 0) eng->prio_shift = cfg->tune_prio_shift;
  else if (cfg->tune_prio_shift == 0) {
    StepParams probe;
    hwy::params_from_config(*cfg, eng->pitch, probe);
    int resident = 0;
    if (cfg->scenario == HWY_SCENARIO_HIGHWAY) resident = hwy::step_resident_blocks(probe, eng->waves_per_eu, eng->force_block_kernel, cfg->tune_extra_lds);
    else if (cfg->scenario != HWY_SCENARIO_INTERSECTION) resident = hwy::net_step_resident_blocks(eng->waves_per_eu);
    // the turn that pays is about a sixth of a wavefront's lifetime, i.e. it grows with the frames of a policy step: 2^14 ticks
    // for the 5 frames of highway-fast-v0 (13: 47.0 us, 14: 44.96, 15: 47.7), 2^16 for the 15 frames of highway-v0 (14: 134.3
    // us, 15 / 16: 133.3) and of the merge scenarios (config 5: 14: 313.8, 16 / 17: 296.0, 18: 314.8; merge-v0: 14: 178.2,
    // 15 / 16: 172.7, 17: 182.1) and of the workgroup kernel, whose wavefronts take turns by workgroup (config-3 shard 1024 x 101:
    // off 161.1, 16: 159.4; 2048 x 101: 214.8 / 207.6) -- profiles/r03_history.md
    int shift = HWY_DEFAULT_PRIO_SHIFT;
    for (int f = 7; f <= cfg->frames_per_step && shift < 18; f *= 2) ++shift;  // +1 from 7 frames on, +2 from 14 on, ...
    // Round 5: the optimum is sharp and does not sit on a power of two (turns of k x 64 ticks, tune_prio_shift >= 64; all on the final
    // build, tools/r05_call16.sh .. 18.sh).  One-wavefront highway kernel: 5 frames 224 .. 256 (x 64 ticks: 40.4 us; 192: 40.9,
    // 288: 40.9, 2^15: 42.9); 15 frames 768 (107.5 us; 384: 107.7, 512: 109.6, 2^16: 111.3) -- a turn of 51 x 64 ticks per frame.
    // merge-generic (config 5): 832 (222.5 us; 768: 223.0, 2^16: 226.2); merge-v0 keeps 2^16 (149.3 us; 768 .. 960: 150 .. 152).
    int turn = shift;
    if (cfg->scenario == HWY_SCENARIO_HIGHWAY && cfg->num_vehicles <= 64 && !eng->force_block_kernel)
      turn = std::max(64, 256 * cfg->frames_per_step / 5);
    else if (cfg->scenario == HWY_SCENARIO_MERGE_GENERIC)
      turn = std::max(64, 832 * cfg->frames_per_step / 15);
    if (turn >= 64) turn = std::min(turn, 1 << 20);  // (the validated range of the linear encoding, whatever frames_per_step is)
    eng->prio_shift = (resident > 0 && cfg->num_envs <= resident) ? turn : 0;
    // The optimum is sharp and depends on (envs, vehicles, frames) -- profiles/r05_history.md section 10: up to 7 % between neighbours --
    // and the defaults above were swept on BASELINE's shapes only: the engine refines its own choice on its first launches
    // (tuner_*: below).  An explicit tune_prio_shift (> 0, or -1 = off) is never touched.
    if (eng->prio_shift > 0) tuner_arm(eng, eng->prio_shift);
  }
  *out = eng;
  return HWY_OK;
}

extern "C" int hwy_destroy(hwy_engine *eng) {
  if (!eng) return HWY_OK;
  (void)hipSetDevice(eng->device);
  if (eng->stream) (void)hipStreamSynchronize(eng->stream);
  if (eng->comm) { hwy::comm_destroy(eng->comm); eng->comm = nullptr; }
  for (auto &pr : eng->events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto &pr : eng->tuner.events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  void *ptrs[] = {eng->d_f64, eng->d_packed, eng->d_time, eng->d_done, eng->d_episode, eng->d_actions, eng->d_out, eng->d_roll,
                  eng->d_mask, eng->d_seeds, eng->d_grid_ws, eng->d_route, eng->d_road_steps, eng->d_gnet,
                  eng->d_shadow_f64, eng->d_shadow_packed, eng->d_shadow_route, eng->d_shadow_meta, eng->d_counters, eng->d_block_env};
  for (void *q : ptrs) if (q) (void)hipFree(q);
  if (eng->h_pinned) (void)hipHostFree(eng->h_pinned);
  if (eng->own_stream && eng->stream) (void)hipStreamDestroy(eng->stream);
  delete eng;
  return HWY_OK;
}

// ---- state injection / inspection -----------------------------------------------------------------
static void pack_rows(const hwy_engine *eng, const double *src, double *dst) {  // [E][N] -> [E][pitch]
  const int E = eng->cfg.num_envs, N = eng->cfg.num_vehicles, P = eng->pitch;
  for (int e = 0; e < E; ++e) {
    std::memcpy(dst + (size_t)e * P, src + (size_t)e * N, sizeof(double) * N);
    for (int i = N; i < P; ++i) dst[(size_t)e * P + i] = 0.0;
  }
}
static void unpack_rows(const hwy_engine *eng, const double *src, double *dst) {  // [E][pitch] -> [E][N]
  const int E = eng->cfg.num_envs, N = eng->cfg.num_vehicles, P = eng->pitch;
  for (int e = 0; e < E; ++e) std::memcpy(dst + (size_t)e * N, src + (size_t)e * P, sizeof(double) * N);
}

extern "C" int hwy_set_state(hwy_engine *eng, const hwy_state *h) {
  if (!eng || !h) return HWY_ERR_INVALID_ARG;
  const double *fields[9] = {h->x, h->y, h->heading, h->speed, h->timer, h->target_speed, h->delta, h->impact_x, h->impact_y};
  for (auto f : fields) if (!f) return fail(eng, HWY_ERR_INVALID_ARG, "hwy_set_state: NULL field");
  if (!h->lane || !h->target_lane || !h->speed_index || !h->flags || !h->time)
    return fail(eng, HWY_ERR_INVALID_ARG, "hwy_set_state: NULL field");
  HWY_HIP(eng, hipSetDevice(eng->device));
  const int E = eng->cfg.num_envs, N = eng->cfg.num_vehicles, P = eng->pitch;
  const size_t plane = (size_t)E * P;
  const int n_lane_ids = is_ix(eng) ? eng->cfg.gnet_lanes : is_net(eng) ? eng->cfg.net_lanes : eng->cfg.lanes_count;
  if (is_ix(eng) && (!h->route || !h->road_steps)) return fail(eng, HWY_ERR_INVALID_ARG, "hwy_set_state: route / road_steps are required by the intersection scenario");
  for (size_t k = 0; k < (size_t)E * N; ++k) {
    if (h->lane[k] < 0 || h->lane[k] >= n_lane_ids || h->target_lane[k] < 0 || h->target_lane[k] >= n_lane_ids)
      return fail(eng, HWY_ERR_INVALID_ARG, "hwy_set_state: lane index out of range");
    if (h->speed_index[k] < 0 || h->speed_index[k] >= eng->cfg.num_target_speeds)
      return fail(eng, HWY_ERR_INVALID_ARG, "hwy_set_state: speed_index out of range");
  }
  double *stage = (double *)eng->h_pinned;
  for (int f = 0; f < 9; ++f) pack_rows(eng, fields[f], stage + f * plane);
  int32_t *pk = (int32_t *)(stage + 9 * plane);
  for (int e = 0; e < E; ++e)
    for (int i = 0; i < P; ++i) {
      const size_t k = (size_t)e * N + i;
      if (is_ix(eng)) pk[(size_t)e * P + i] = i < N ? hwy::ix_pack_word(h->lane[k], h->target_lane[k], h->speed_index[k], h->flags[k])
                                                    : hwy::ix_pack_word(0, 0, 0, HWY_F_ABSENT);
      else pk[(size_t)e * P + i] = i < N ? hwy::pack_word(h->lane[k], h->target_lane[k], h->speed_index[k], h->flags[k], i) : 0;
    }
  double *tm = (double *)(pk + plane);
  std::memcpy(tm, h->time, sizeof(double) * E);
  if (is_ix(eng)) {
    long long *rt = (long long *)(tm + E);
    int32_t *rs = (int32_t *)(rt + plane);
    for (int e = 0; e < E; ++e)
      for (int i = 0; i < P; ++i) rt[(size_t)e * P + i] = i < N ? (long long)h->route[(size_t)e * N + i] : 0;
    std::memcpy(rs, h->road_steps, sizeof(int32_t) * E);
    HWY_HIP(eng, hipMemcpyAsync(eng->d_route, rt, plane * sizeof(long long), hipMemcpyHostToDevice, eng->stream));
    HWY_HIP(eng, hipMemcpyAsync(eng->d_road_steps, rs, E * sizeof(int32_t), hipMemcpyHostToDevice, eng->stream));
  }
  HWY_HIP(eng, hipMemcpyAsync(eng->d_f64, stage, plane * 9 * sizeof(double), hipMemcpyHostToDevice, eng->stream));
  HWY_HIP(eng, hipMemcpyAsync(eng->d_packed, pk, plane * sizeof(int32_t), hipMemcpyHostToDevice, eng->stream));
  HWY_HIP(eng, hipMemcpyAsync(eng->d_time, tm, E * sizeof(double), hipMemcpyHostToDevice, eng->stream));
  HWY_HIP(eng, hipMemsetAsync(eng->d_done, 0, E, eng->stream));
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  return HWY_OK;
}

extern "C" int hwy_get_state(hwy_engine *eng, hwy_state *h) {
  if (!eng || !h) return HWY_ERR_INVALID_ARG;
  HWY_HIP(eng, hipSetDevice(eng->device));
  const int E = eng->cfg.num_envs, N = eng->cfg.num_vehicles, P = eng->pitch;
  const size_t plane = (size_t)E * P;
  double *stage = (double *)eng->h_pinned;
  int32_t *pk = (int32_t *)(stage + 9 * plane);
  double *tm = (double *)(pk + plane);
  HWY_HIP(eng, hipMemcpyAsync(stage, eng->d_f64, plane * 9 * sizeof(double), hipMemcpyDeviceToHost, eng->stream));
  HWY_HIP(eng, hipMemcpyAsync(pk, eng->d_packed, plane * sizeof(int32_t), hipMemcpyDeviceToHost, eng->stream));
  HWY_HIP(eng, hipMemcpyAsync(tm, eng->d_time, E * sizeof(double), hipMemcpyDeviceToHost, eng->stream));
  long long *rt = (long long *)(tm + E);
  int32_t *rs = (int32_t *)(rt + plane);
  if (is_ix(eng)) {
    HWY_HIP(eng, hipMemcpyAsync(rt, eng->d_route, plane * sizeof(long long), hipMemcpyDeviceToHost, eng->stream));
    HWY_HIP(eng, hipMemcpyAsync(rs, eng->d_road_steps, E * sizeof(int32_t), hipMemcpyDeviceToHost, eng->stream));
  }
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  if (is_ix(eng)) {
    for (int e = 0; e < E; ++e)
      for (int i = 0; i < N; ++i) {
        const int32_t w = pk[(size_t)e * P + i];
        const size_t k = (size_t)e * N + i;
        if (h->lane) h->lane[k] = hwy::ix_word_lane(w);
        if (h->target_lane) h->target_lane[k] = hwy::ix_word_target(w);
        if (h->speed_index) h->speed_index[k] = hwy::ix_word_speed_index(w);
        if (h->flags) h->flags[k] = hwy::ix_word_flags(w);
        if (h->route) h->route[k] = (int64_t)rt[(size_t)e * P + i];
      }
    if (h->road_steps) std::memcpy(h->road_steps, rs, sizeof(int32_t) * E);
    double *flds[9] = {h->x, h->y, h->heading, h->speed, h->timer, h->target_speed, h->delta, h->impact_x, h->impact_y};
    for (int f = 0; f < 9; ++f) if (flds[f]) unpack_rows(eng, stage + f * plane, flds[f]);
    // the step kernel maintains the impact pair only while its flag is set, and nothing but the flag word of an empty slot
    for (int e = 0; e < E; ++e)
      for (int i = 0; i < N; ++i) {
        const int fl = hwy::ix_word_flags(pk[(size_t)e * P + i]);
        const size_t k = (size_t)e * N + i;
        if (fl & HWY_F_ABSENT) {
          for (int f = 0; f < 9; ++f) if (flds[f]) flds[f][k] = 0.0;
          if (h->lane) h->lane[k] = 0;
          if (h->target_lane) h->target_lane[k] = 0;
          if (h->speed_index) h->speed_index[k] = 0;
          if (h->flags) h->flags[k] = HWY_F_ABSENT;
          if (h->route) h->route[k] = 0;
        } else if (!(fl & HWY_F_HAS_IMPACT)) {
          if (h->impact_x) h->impact_x[k] = 0.0;
          if (h->impact_y) h->impact_y[k] = 0.0;
        }
      }
    if (h->time) std::memcpy(h->time, tm, sizeof(double) * E);
    return HWY_OK;
  }
  double *fields[9] = {h->x, h->y, h->heading, h->speed, h->timer, h->target_speed, h->delta, h->impact_x, h->impact_y};
  for (int f = 0; f < 9; ++f) if (fields[f]) unpack_rows(eng, stage + f * plane, fields[f]);
  for (int e = 0; e < E; ++e)
    for (int i = 0; i < N; ++i) {
      const int32_t w = pk[(size_t)e * P + i];
      const size_t k = (size_t)e * N + i;
      if (h->lane) h->lane[k] = hwy::word_lane(w);
      if (h->target_lane) h->target_lane[k] = hwy::word_target(w);
      if (h->speed_index) h->speed_index[k] = hwy::word_speed_index(w);
      if (h->flags) h->flags[k] = hwy::word_flags(w);
      // the impact pair is only maintained while the flag is set (Vehicle.impact is None otherwise)
      if (!(hwy::word_flags(w) & HWY_F_HAS_IMPACT)) {
        if (h->impact_x) h->impact_x[k] = 0.0;
        if (h->impact_y) h->impact_y[k] = 0.0;
      }
    }
  if (h->time) std::memcpy(h->time, tm, sizeof(double) * E);
  return HWY_OK;
}

// ---- kernel timing ----------------------------------------------------------------------------------
static hipError_t launch_step_any(const hwy_engine *eng, const StepParams &p);
static int timed_launch(hwy_engine *eng, const StepParams &p) {
  auto &tn = eng->tuner;
  if (tn.state == hwy_engine::TurnTuner::SAMPLING && !eng->profiling && p.full_step) {
    const int c = tn.launches % 5;
    if (tn.events.size() <= (size_t)tn.launches) {
      hipEvent_t a, b;
      HWY_HIP(eng, hipEventCreate(&a));
      HWY_HIP(eng, hipEventCreate(&b));
      tn.events.emplace_back(a, b);
    }
    StepParams q = p;
    hwy::set_prio_turn(q, tn.cand[c]);
    auto &pr = tn.events[tn.launches];
    hwy::set_launch_events(pr.first, pr.second);
    const hipError_t err = launch_step_any(eng, q);
    hwy::set_launch_events(nullptr, nullptr);
    HWY_HIP(eng, err);
    tn.which.push_back(c);
    if (++tn.launches == 5 * (HWY_TUNE_ROUNDS + 1)) return tuner_finish_stage(eng);
    return HWY_OK;
  }
  if (!eng->profiling || (eng->launch_counter++ % eng->profiling) != 0) {
    HWY_HIP(eng, launch_step_any(eng, p));
    return HWY_OK;
  }
  if (eng->events_used == eng->events.size()) {
    hipEvent_t a, b;
    HWY_HIP(eng, hipEventCreate(&a));
    HWY_HIP(eng, hipEventCreate(&b));
    eng->events.emplace_back(a, b);
  }
  auto &pr = eng->events[eng->events_used++];
  hwy::set_launch_events(pr.first, pr.second);  // the dispatch's own begin / end timestamps (hwy_kernels.hip)
  const hipError_t err = launch_step_any(eng, p);
  hwy::set_launch_events(nullptr, nullptr);
  HWY_HIP(eng, err);
  return HWY_OK;
}
static int drain_events(hwy_engine *eng) {
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  for (size_t k = 0; k < eng->events_used; ++k) {
    float ms = 0;
    HWY_HIP(eng, hipEventElapsedTime(&ms, eng->events[k].first, eng->events[k].second));
    eng->prof_ms += ms;
    eng->prof_launches++;
  }
  eng->events_used = 0;
  return HWY_OK;
}
extern "C" int hwy_get_prio_turn(hwy_engine *eng, int32_t *turn, int32_t *state) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  if (turn) *turn = eng->prio_shift;
  if (state) *state = eng->tuner.state;
  return HWY_OK;
}
extern "C" int hwy_profile_enable(hwy_engine *eng, int32_t enabled) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  HWY_HIP(eng, hipSetDevice(eng->device));
  if (int rc = drain_events(eng)) return rc;
  eng->profiling = enabled > 0 ? enabled : 0;
  eng->launch_counter = 0;
  if (enabled) { eng->prof_ms = 0.0; eng->prof_launches = 0; }
  return HWY_OK;
}
extern "C" int hwy_profile_read(hwy_engine *eng, double *total_ms, int64_t *launches) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  HWY_HIP(eng, hipSetDevice(eng->device));
  if (int rc = drain_events(eng)) return rc;
  if (total_ms) *total_ms = eng->prof_ms;
  if (launches) *launches = eng->prof_launches;
  return HWY_OK;
}

// ---- stepping ------------------------------------------------------------------------------------------
extern "C" int hwy_step_device(hwy_engine *eng, const int32_t *d_actions, float *d_obs, double *d_reward,
                               uint8_t *d_terminated, uint8_t *d_truncated, double *d_info_speed,
                               uint8_t *d_info_crashed) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  if (!d_actions || !d_obs || !d_reward || !d_terminated || !d_truncated)
    return fail(eng, HWY_ERR_INVALID_ARG, "hwy_step_device: actions/obs/reward/terminated/truncated must be non-NULL");
  HWY_HIP(eng, hipSetDevice(eng->device));
  StepParams p;
  fill_params(eng, p);
  p.n_frames = eng->cfg.frames_per_step;
  p.full_step = 1;
  p.actions = d_actions; p.obs = d_obs; p.reward = d_reward; p.terminated = d_terminated; p.truncated = d_truncated;
  p.info_speed = d_info_speed; p.info_crashed = d_info_crashed;
  if (eng->profiling && eng->events_used >= 65536)
    if (int rc = drain_events(eng)) return rc;
  return timed_launch(eng, p);
}

extern "C" int hwy_rollout_device(hwy_engine *eng, int32_t k_steps, const int32_t *d_actions, float *d_obs, double *d_reward,
                                  uint8_t *d_terminated, uint8_t *d_truncated, double *d_info_speed, uint8_t *d_info_crashed) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  if (k_steps < 1) return fail(eng, HWY_ERR_INVALID_ARG, "hwy_rollout_device: k_steps must be >= 1");
  if (!d_actions || !d_obs || !d_reward || !d_terminated || !d_truncated)
    return fail(eng, HWY_ERR_INVALID_ARG, "hwy_rollout_device: actions/obs/reward/terminated/truncated must be non-NULL");
  if (k_steps > 1 && is_ix(eng) && (eng->cfg.flags & HWY_C_HOST_TRAFFIC))  // the host clears / spawns BETWEEN policy steps
    return fail(eng, HWY_ERR_INVALID_ARG, "hwy_rollout_device: k_steps > 1 needs device traffic (HWY_C_HOST_TRAFFIC is set: "
                                          "_clear_vehicles / _spawn_vehicle run on the host between steps)");
  HWY_HIP(eng, hipSetDevice(eng->device));
  StepParams p;
  fill_params(eng, p);
  p.n_frames = eng->cfg.frames_per_step;
  p.full_step = 1;
  p.actions = d_actions; p.obs = d_obs; p.reward = d_reward; p.terminated = d_terminated; p.truncated = d_truncated;
  p.info_speed = d_info_speed; p.info_crashed = d_info_crashed;
  if (!is_ix(eng) && !is_net(eng)) {  // straight road: K steps in ONE launch
    p.k_steps = k_steps;
    p.num_envs = eng->cfg.num_envs;
    HWY_HIP(eng, hwy::launch_rollout(p, eng->cfg.num_envs, eng->stream, eng->rollout_waves_per_eu, eng->cfg.tune_extra_lds,
                                     eng->force_block_kernel, eng->waves_per_eu));
    return HWY_OK;
  }
  if (is_net(eng)) {  // the merge kernel has the multi-step form too
    p.k_steps = k_steps;
    p.num_envs = eng->cfg.num_envs;
    hwy::NetParams np;
    hwy::net_params_from_config(eng->cfg, p, np);
    HWY_HIP(eng, hwy::launch_net_rollout(np, eng->cfg.num_envs, eng->stream, eng->waves_per_eu));
    return HWY_OK;
  }
  // the intersection kernel: STEP blocks only (no shadow is advanced during the launch; an environment that ends in it warms its
  // next episode up inline -- WHEN the warm-up frames are computed cannot change a result)
  p.k_steps = k_steps;
  p.num_envs = eng->cfg.num_envs;
  hwy::IxParams ip;
  fill_ix(eng, p, ip);
  HWY_HIP(eng, hwy::launch_ix_rollout(ip, eng->cfg.num_envs, eng->stream, eng->waves_per_eu));
  return HWY_OK;
}

extern "C" int hwy_rollout(hwy_engine *eng, int32_t k_steps, const int32_t *actions, float *obs, double *reward,
                           uint8_t *terminated, uint8_t *truncated, double *info_speed, uint8_t *info_crashed) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  if (k_steps < 1) return fail(eng, HWY_ERR_INVALID_ARG, "hwy_rollout: k_steps must be >= 1");
  if (!actions || !obs || !reward || !terminated || !truncated)
    return fail(eng, HWY_ERR_INVALID_ARG, "hwy_rollout: actions/obs/reward/terminated/truncated must be non-NULL");
  if (k_steps > 1 && is_ix(eng) && (eng->cfg.flags & HWY_C_HOST_TRAFFIC))
    return fail(eng, HWY_ERR_INVALID_ARG, "hwy_rollout: k_steps > 1 needs device traffic (HWY_C_HOST_TRAFFIC is set)");
  size_t n_act, n_obs, n_ea;
  io_counts(eng->cfg, &n_act, &n_obs, &n_ea);
  const size_t E = eng->cfg.num_envs, K = (size_t)k_steps;
  const int max_action = is_ix(eng) ? 2 : HWY_NUM_ACTIONS(eng->cfg.action_set) - 1;
  for (size_t k = 0; k < K * n_act; ++k)  // the reference's KeyError, before anything is simulated (action.py:260)
    if (actions[k] < 0 || actions[k] > max_action) return fail(eng, HWY_ERR_ACTION, "meta-action out of range");
  HWY_HIP(eng, hipSetDevice(eng->device));
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_act = 0, o_rew = up(K * n_act * 4), o_spd = o_rew + up(K * n_ea * 8), o_obs = o_spd + up(K * n_ea * 8),
               o_term = o_obs + up(K * n_obs * 4), o_trunc = o_term + up(K * E), o_crash = o_trunc + up(K * E),
               total = o_crash + up(K * n_ea);
  if (total > eng->roll_bytes) {
    HWY_HIP(eng, hipStreamSynchronize(eng->stream));
    if (eng->d_roll) (void)hipFree(eng->d_roll);
    eng->d_roll = nullptr;
    eng->roll_bytes = 0;
    HWY_HIP(eng, hipMalloc((void **)&eng->d_roll, total));
    eng->roll_bytes = total;
  }
  char *d = eng->d_roll;
  HWY_HIP(eng, hipMemcpyAsync(d + o_act, actions, K * n_act * 4, hipMemcpyHostToDevice, eng->stream));
  if (int rc = hwy_rollout_device(eng, k_steps, (const int32_t *)(d + o_act), (float *)(d + o_obs), (double *)(d + o_rew),
                                  (uint8_t *)(d + o_term), (uint8_t *)(d + o_trunc), (double *)(d + o_spd), (uint8_t *)(d + o_crash)))
    return rc;
  HWY_HIP(eng, hipMemcpyAsync(obs, d + o_obs, K * n_obs * 4, hipMemcpyDeviceToHost, eng->stream));
  HWY_HIP(eng, hipMemcpyAsync(reward, d + o_rew, K * n_ea * 8, hipMemcpyDeviceToHost, eng->stream));
  HWY_HIP(eng, hipMemcpyAsync(terminated, d + o_term, K * E, hipMemcpyDeviceToHost, eng->stream));
  HWY_HIP(eng, hipMemcpyAsync(truncated, d + o_trunc, K * E, hipMemcpyDeviceToHost, eng->stream));
  if (info_speed) HWY_HIP(eng, hipMemcpyAsync(info_speed, d + o_spd, K * n_ea * 8, hipMemcpyDeviceToHost, eng->stream));
  if (info_crashed) HWY_HIP(eng, hipMemcpyAsync(info_crashed, d + o_crash, K * n_ea, hipMemcpyDeviceToHost, eng->stream));
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  return HWY_OK;
}

extern "C" int hwy_step(hwy_engine *eng, const int32_t *actions, float *obs, double *reward, uint8_t *terminated,
                        uint8_t *truncated, double *info_speed, uint8_t *info_crashed) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  if (!actions || !obs || !reward || !terminated || !truncated)
    return fail(eng, HWY_ERR_INVALID_ARG, "hwy_step: actions/obs/reward/terminated/truncated must be non-NULL");
  size_t n_act, n_obs, n_ea;
  io_counts(eng->cfg, &n_act, &n_obs, &n_ea);
  const size_t E = eng->cfg.num_envs;
  // the reference raises KeyError for an unknown meta-action before touching the simulation (action.py:260)
  const int max_action = is_ix(eng) ? 2 : HWY_NUM_ACTIONS(eng->cfg.action_set) - 1;  // IntersectionEnv.ACTIONS has 3 entries (intersection_env.py:14)
  for (size_t k = 0; k < n_act; ++k)
    if (actions[k] < 0 || actions[k] > max_action) return fail(eng, HWY_ERR_ACTION, max_action == 2 ? "meta-action outside [0,3)" : "meta-action outside [0,5)");
  HWY_HIP(eng, hipSetDevice(eng->device));
  // pinned layout: [mirror of the device output block][actions]
  char *h_out = (char *)eng->h_pinned;
  int32_t *h_act = (int32_t *)(h_out + eng->out_bytes);
  std::memcpy(h_act, actions, n_act * 4);
  HWY_HIP(eng, hipMemcpyAsync(eng->d_actions, h_act, n_act * 4, hipMemcpyHostToDevice, eng->stream));
  if (int rc = hwy_step_device(eng, eng->d_actions, eng->d_obs, eng->d_reward, eng->d_term, eng->d_trunc,
                               eng->d_info_speed, eng->d_info_crashed))
    return rc;
  HWY_HIP(eng, hipMemcpyAsync(h_out, eng->d_out, eng->out_bytes, hipMemcpyDeviceToHost, eng->stream));
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  std::memcpy(obs, h_out + eng->off_obs, n_obs * 4);
  std::memcpy(reward, h_out + eng->off_reward, n_ea * 8);
  std::memcpy(terminated, h_out + eng->off_term, E);
  std::memcpy(truncated, h_out + eng->off_trunc, E);
  if (info_speed) std::memcpy(info_speed, h_out + eng->off_speed, n_ea * 8);
  if (info_crashed) std::memcpy(info_crashed, h_out + eng->off_crashed, n_ea);
  return HWY_OK;
}

extern "C" int hwy_step_frames(hwy_engine *eng, const int32_t *actions, int32_t n_frames) {
  if (!eng || n_frames < 0) return HWY_ERR_INVALID_ARG;
  HWY_HIP(eng, hipSetDevice(eng->device));
  size_t n_act, n_obs, n_ea;
  io_counts(eng->cfg, &n_act, &n_obs, &n_ea);
  StepParams p;
  fill_params(eng, p);
  p.n_frames = n_frames;
  p.full_step = 0;
  p.autoreset = 0;
  if (actions) {
    for (size_t k = 0; k < n_act; ++k)
      if (actions[k] < 0 || actions[k] > (is_ix(eng) ? 2 : HWY_NUM_ACTIONS(eng->cfg.action_set) - 1)) return fail(eng, HWY_ERR_ACTION, "meta-action out of range");
    std::memcpy(eng->h_pinned, actions, n_act * 4);
    HWY_HIP(eng, hipMemcpyAsync(eng->d_actions, eng->h_pinned, n_act * 4, hipMemcpyHostToDevice, eng->stream));
    p.actions = eng->d_actions;
  }
  p.reward = eng->d_reward; p.terminated = eng->d_term; p.truncated = eng->d_trunc;
  if (int rc = timed_launch(eng, p)) return rc;
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  return HWY_OK;
}

extern "C" int hwy_observe(hwy_engine *eng, float *obs) {
  if (!eng || !obs) return HWY_ERR_INVALID_ARG;
  HWY_HIP(eng, hipSetDevice(eng->device));
  size_t n_act, n_obs, n_ea;
  io_counts(eng->cfg, &n_act, &n_obs, &n_ea);
  StepParams p;
  fill_params(eng, p);
  p.obs = eng->d_obs;
  p.reward = eng->d_reward; p.terminated = eng->d_term; p.truncated = eng->d_trunc;
  HWY_HIP(eng, launch_observe_any(eng, p));
  HWY_HIP(eng, hipMemcpyAsync(eng->h_pinned, eng->d_obs, n_obs * 4, hipMemcpyDeviceToHost, eng->stream));
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  std::memcpy(obs, eng->h_pinned, n_obs * 4);
  return HWY_OK;
}

// ---- reset --------------------------------------------------------------------------------------------------
static int set_reset_params(hwy_engine *eng, double ego_spacing, double vehicles_density, int32_t initial_lane_id) {
  if (!(ego_spacing > 0) || !(vehicles_density > 0)) return fail(eng, HWY_ERR_INVALID_ARG, "spacing/density must be positive");
  if (initial_lane_id >= eng->cfg.lanes_count) return fail(eng, HWY_ERR_INVALID_ARG, "initial_lane_id out of range");
  // (road-network scenarios: the spawn rule of MergeEnv / MergeGenericEnv has no spacing / density / lane parameters)
  eng->rp.ego_spacing = ego_spacing;
  eng->rp.other_spacing = 1 / vehicles_density;  // highway_env.py:94
  eng->rp.initial_lane_id = initial_lane_id < 0 ? -1 : initial_lane_id;
  return HWY_OK;
}

extern "C" int hwy_reset(hwy_engine *eng, const uint8_t *mask, const uint64_t *seeds, double ego_spacing,
                         double vehicles_density, int32_t initial_lane_id, float *obs) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  if (int rc = set_reset_params(eng, ego_spacing, vehicles_density, initial_lane_id)) return rc;
  HWY_HIP(eng, hipSetDevice(eng->device));
  const size_t E = eng->cfg.num_envs;
  size_t n_act, n_obs, n_ea;
  io_counts(eng->cfg, &n_act, &n_obs, &n_ea);
  StepParams p;
  fill_params(eng, p);
  char *base = (char *)eng->h_pinned;
  if (seeds) {
    std::memcpy(base, seeds, E * 8);
    HWY_HIP(eng, hipMemcpyAsync(eng->d_seeds, base, E * 8, hipMemcpyHostToDevice, eng->stream));
    p.reset_seeds = eng->d_seeds;
  }
  if (mask) {
    std::memcpy(base + E * 8, mask, E);
    HWY_HIP(eng, hipMemcpyAsync(eng->d_mask, base + E * 8, E, hipMemcpyHostToDevice, eng->stream));
    p.reset_mask = eng->d_mask;
  }
  p.obs = eng->d_obs;
  p.reward = eng->d_reward; p.terminated = eng->d_term; p.truncated = eng->d_trunc;
  HWY_HIP(eng, launch_reset_any(eng, p));
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  if (obs) {
    HWY_HIP(eng, hipMemcpyAsync(eng->h_pinned, eng->d_obs, n_obs * 4, hipMemcpyDeviceToHost, eng->stream));
    HWY_HIP(eng, hipStreamSynchronize(eng->stream));
    const size_t per_env = n_obs / E;
    const float *src = (const float *)eng->h_pinned;
    for (size_t e = 0; e < E; ++e)
      if (!mask || mask[e]) std::memcpy(obs + e * per_env, src + e * per_env, per_env * 4);
  }
  return HWY_OK;
}

extern "C" int hwy_set_autoreset(hwy_engine *eng, int32_t enabled, uint64_t base_seed, double ego_spacing,
                                 double vehicles_density, int32_t initial_lane_id) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  if (int rc = set_reset_params(eng, ego_spacing, vehicles_density, initial_lane_id)) return rc;
  eng->autoreset = enabled ? 1 : 0;
  if (eng->rp.base_seed != base_seed) {  // pre-warmed next episodes were drawn from the old seeds
    HWY_HIP(eng, hipSetDevice(eng->device));
    HWY_HIP(eng, invalidate_shadows(eng));
  }
  eng->rp.base_seed = base_seed;
  return HWY_OK;
}

extern "C" int hwy_debug_math(hwy_engine *eng, int32_t op, const double *in, double *out, int64_t n) {
  if (!eng || !in || !out || n < 0 || op < 0 || (op > 11 && (op < 20 || op > 33) && op != 40 && op != 41)) return HWY_ERR_INVALID_ARG;
  if (n == 0) return HWY_OK;
  HWY_HIP(eng, hipSetDevice(eng->device));
  double *d_in = nullptr, *d_out = nullptr;
  HWY_HIP(eng, hipMalloc((void **)&d_in, n * sizeof(double)));
  if (hipMalloc((void **)&d_out, n * sizeof(double)) != hipSuccess) { (void)hipFree(d_in); return fail(eng, HWY_ERR_HIP, "hipMalloc"); }
  hipError_t e = hipMemcpyAsync(d_in, in, n * sizeof(double), hipMemcpyHostToDevice, eng->stream);
  if (e == hipSuccess) e = hwy::launch_math_probe(op, d_in, d_out, (long long)n, eng->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, eng->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(eng->stream);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  if (e != hipSuccess) return fail(eng, HWY_ERR_HIP, std::string("hwy_debug_math: ") + hipGetErrorString(e));
  return HWY_OK;
}

extern "C" int hwy_get_counters(hwy_engine *eng, uint64_t *out, int32_t n, int32_t reset) {
  if (!eng || !out || n < 0) return HWY_ERR_INVALID_ARG;
  HWY_HIP(eng, hipSetDevice(eng->device));
  unsigned long long host[HWY_CTR_COUNT];
  HWY_HIP(eng, hipMemcpyAsync(host, eng->d_counters, sizeof host, hipMemcpyDeviceToHost, eng->stream));
  if (reset) HWY_HIP(eng, hipMemsetAsync(eng->d_counters, 0, sizeof host, eng->stream));
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  for (int k = 0; k < n && k < HWY_CTR_COUNT; ++k) out[k] = host[k];
  return HWY_OK;
}

extern "C" int hwy_set_block_order(hwy_engine *eng, const int32_t *env_of_block) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  HWY_HIP(eng, hipSetDevice(eng->device));
  const int E = eng->cfg.num_envs;
  if (!env_of_block) {  // back to the identity
    HWY_HIP(eng, hipStreamSynchronize(eng->stream));
    if (eng->d_block_env) { HWY_HIP(eng, hipFree(eng->d_block_env)); eng->d_block_env = nullptr; }
    return HWY_OK;
  }
  if (eng->cfg.scenario != HWY_SCENARIO_HIGHWAY || eng->cfg.num_vehicles > 64 || eng->force_block_kernel || E > 65535)
    return fail(eng, HWY_ERR_INVALID_ARG, "hwy_set_block_order: only the one-wavefront step kernel (highway scenario, N <= 64, "
                                          "at most 65535 environments) takes a workgroup order");
  std::vector<uint8_t> seen((size_t)E, 0);
  std::vector<uint16_t> tab((size_t)E);
  for (int b = 0; b < E; ++b) {
    const int32_t e = env_of_block[b];
    if (e < 0 || e >= E || seen[(size_t)e]) return fail(eng, HWY_ERR_INVALID_ARG, "hwy_set_block_order: not a permutation of 0 .. num_envs - 1");
    seen[(size_t)e] = 1;
    tab[(size_t)b] = (uint16_t)e;
  }
  if (!eng->d_block_env) HWY_HIP(eng, hipMalloc((void **)&eng->d_block_env, (size_t)E * sizeof(uint16_t)));
  HWY_HIP(eng, hipMemcpyAsync(eng->d_block_env, tab.data(), (size_t)E * sizeof(uint16_t), hipMemcpyHostToDevice, eng->stream));
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));  // (tab is a local)
  return HWY_OK;
}

extern "C" int hwy_comm_unique_id(uint8_t *id) {
  if (!id) return fail(nullptr, HWY_ERR_INVALID_ARG, "id is NULL");
  std::string err;
  const int rc = hwy::comm_unique_id(id, err);
  return rc ? fail(nullptr, rc, err) : HWY_OK;
}
extern "C" int hwy_comm_init(hwy_engine *eng, const uint8_t *id, int32_t rank, int32_t world) {
  if (!eng || !id || world < 1 || rank < 0 || rank >= world) return HWY_ERR_INVALID_ARG;
  if (eng->comm) return fail(eng, HWY_ERR_INVALID_ARG, "hwy_comm_init: this engine already has a communicator");
  HWY_HIP(eng, hipSetDevice(eng->device));
  std::string err;
  const int rc = hwy::comm_init(&eng->comm, id, rank, world, err);
  return rc ? fail(eng, rc, err) : HWY_OK;
}
extern "C" int hwy_gather(hwy_engine *eng, const void *d_send, void *d_recv, size_t bytes, int32_t root) {
  if (!eng || !d_send) return HWY_ERR_INVALID_ARG;
  if (!eng->comm) return fail(eng, HWY_ERR_INVALID_ARG, "hwy_gather: call hwy_comm_init first");
  if (root < 0 || root >= hwy::comm_world(eng->comm)) return fail(eng, HWY_ERR_INVALID_ARG, "hwy_gather: root out of range");
  if (hwy::comm_rank(eng->comm) == root && !d_recv) return fail(eng, HWY_ERR_INVALID_ARG, "hwy_gather: d_recv is NULL on the root");
  HWY_HIP(eng, hipSetDevice(eng->device));
  std::string err;
  const int rc = hwy::comm_gather(eng->comm, d_send, d_recv, bytes, root, eng->stream, err);
  return rc ? fail(eng, rc, err) : HWY_OK;
}
extern "C" int hwy_comm_destroy(hwy_engine *eng) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  if (eng->comm) {
    (void)hipSetDevice(eng->device);
    (void)hipStreamSynchronize(eng->stream);
    hwy::comm_destroy(eng->comm);
    eng->comm = nullptr;
  }
  return HWY_OK;
}

extern "C" int hwy_sync(hwy_engine *eng) {
  if (!eng) return HWY_ERR_INVALID_ARG;
  HWY_HIP(eng, hipSetDevice(eng->device));
  HWY_HIP(eng, hipStreamSynchronize(eng->stream));
  return HWY_OK;
}

// hwy_params.h -- host helper: flatten hwy_config into the kernel-argument block.
#pragma once
#include <cstdlib>
#include <cmath>
#include <cstring>

#include "hwy_device.h"

#ifndef HWY_DEFAULT_PRIO_SHIFT
#define HWY_DEFAULT_PRIO_SHIFT 14  // 16 k clock ticks (~7 us) per turn at 5 frames per policy step (hwy_create scales it); used only when the
                                   // whole grid is resident at once -- otherwise the hardware's oldest-first order lets queued
                                   // workgroups start sooner (measured: highway-v0 at 3 waves/SIMD loses 10 % with turns)
#endif

namespace hwy {

// The length of an issue-priority turn in either encoding of hwy_config.tune_prio_shift (1..30: 2^k clock ticks; >= 64: k x 64 ticks,
// evaluated with a multiply-high by floor(2^32 / k)): the ONLY place the pair (shift, reciprocal) is formed, so the two cannot
// diverge (a shift >= 64 with a zero reciprocal would be a 64-bit shift by >= 64 in wave_turn).
inline void set_prio_turn(StepParams &p, int turn) {
  p.prio_shift = turn > 0 ? turn : 0;
  p.prio_recip = turn >= 64 ? (uint32_t)(0x100000000ull / (unsigned long long)turn) : 0u;
}
inline void params_from_config(const hwy_config &c, int pitch, StepParams &p) {
  std::memset(&p, 0, sizeof p);
  p.N = c.num_vehicles; p.A = c.num_agents; p.L = c.lanes_count; p.T = c.frames_per_step;
  p.flags = c.flags; p.V = c.obs_vehicles; p.F = c.obs_features; p.n_ts = c.num_target_speeds;
  p.pitch = pitch;
  p.action_set = c.action_set;
  for (int a = 0; a < HWY_MAX_AGENTS; ++a) p.agent_index[a] = a < c.num_agents ? c.agent_index[a] : -1;
  for (int f = 0; f < HWY_MAX_FEATURES; ++f) p.feat[f] = c.obs_feature_ids[f];
  for (int k = 0; k < HWY_MAX_TARGET_SPEEDS; ++k) p.target_speeds[k] = c.target_speeds[k];
  p.dt = c.dt; p.policy_dt = c.policy_dt; p.duration = c.duration; p.lane_width = c.lane_width;
  p.road_length = c.road_length; p.speed_limit = c.speed_limit;
  p.collision_reward = c.collision_reward; p.right_lane_reward = c.right_lane_reward;
  p.high_speed_reward = c.high_speed_reward; p.rs0 = c.reward_speed_range[0]; p.rs1 = c.reward_speed_range[1];
  p.perception = c.perception_distance;
  p.rx0 = c.obs_range_x[0]; p.rx1 = c.obs_range_x[1]; p.ry0 = c.obs_range_y[0]; p.ry1 = c.obs_range_y[1];
  p.rvx0 = c.obs_range_vx[0]; p.rvx1 = c.obs_range_vx[1]; p.rvy0 = c.obs_range_vy[0]; p.rvy1 = c.obs_range_vy[1];
  p.inv_lane_width = 1.0 / p.lane_width;
  p.inv_rx = 1.0 / (p.rx1 - p.rx0); p.inv_ry = 1.0 / (p.ry1 - p.ry0);
  p.inv_rvx = 1.0 / (p.rvx1 - p.rvx0); p.inv_rvy = 1.0 / (p.rvy1 - p.rvy0);
  set_prio_turn(p, c.tune_prio_shift > 0 ? c.tune_prio_shift : 0);  // the engine turns the default on where it pays (hwy_create)
  p.obs_type = c.obs_type;
  p.obs_std5 = c.obs_type == HWY_OBS_KINEMATICS && c.obs_features == 5;
  for (int f = 0; f < 5; ++f) p.obs_std5 = p.obs_std5 && c.obs_feature_ids[f] == f;  // presence, x, y, vx, vy
  if (c.obs_type == HWY_OBS_OCCUPANCY_GRID) {
    p.gW = c.grid_shape[0]; p.gH = c.grid_shape[1];
    p.gmin_x = c.grid_min[0]; p.gmin_y = c.grid_min[1]; p.gstep_x = c.grid_step[0]; p.gstep_y = c.grid_step[1];
    p.g_spacing = c.grid_step[0] < c.grid_step[1] ? c.grid_step[0] : c.grid_step[1];  // np.amin(grid_step)
    // len(np.arange(origin - 100, origin + 100, spacing)) == ceil(200 / spacing)
    p.g_nwp = (int32_t)std::ceil(200.0 / p.g_spacing);
  }
}

// road-network scenarios: the lane table and merge reward / termination constants next to the StepParams
template <typename NP>
inline void net_params_from_config(const hwy_config &c, const StepParams &p, NP &np) {
  std::memset(&np, 0, sizeof np);
  np.s = p;
  np.n_lanes = c.net_lanes;
  np.merge_lane = c.merge_lane;
  np.generic = c.scenario == HWY_SCENARIO_MERGE_GENERIC ? 1 : 0;
  np.merge_end_x = c.merge_end_x;
  np.merging_speed_reward = c.merging_speed_reward;
  np.lane_change_reward = c.lane_change_reward;
  for (int k = 0; k < HWY_MAX_LANES; ++k) np.lane[k] = c.net[k];
}

// intersection scenario: constants next to the StepParams; the lane table, route plane and step counters are device
// pointers the caller binds (IP = IxParams)
template <typename IP>
inline void ix_params_from_config(const hwy_config &c, const StepParams &p, IP &ip) {
  std::memset(&ip, 0, sizeof ip);
  ip.s = p;
  ip.n_lanes = c.gnet_lanes;
  ip.num_envs = c.num_envs;
  ip.helpers = c.tune_ix_no_helpers ? 0 : 1;  // 0: 32-thread workgroups (no helper lanes) for N <= 32
  // a third of a policy step per launch: the pre-warming wavefronts start when the first step wavefronts retire, and short
  // ones fill the tail of the launch instead of making a second round of it (profiles/r02_history.md)
  ip.prewarm_frames = c.tune_ix_prewarm_frames > 0 ? c.tune_ix_prewarm_frames : (c.frames_per_step + 2) / 3;
  if (ip.prewarm_frames < 1) ip.prewarm_frames = 1;
  ip.initial_count = c.initial_vehicle_count;
  ip.host_spawn = (c.flags & HWY_C_HOST_TRAFFIC) ? 1 : 0;
  ip.destination = c.destination;
  for (int L = 0; L < HWY_MAX_GLANES; ++L)
    for (int k = 0; k < 4; ++k) ip.route_table[L][k] = (long long)c.gnet_routes[L][k];
  for (int k = 0; k < 4; ++k) { ip.access_lane[k] = c.access_lane[k]; ip.exit_of[k] = c.exit_of[k]; }
  ip.spawn_probability = c.spawn_probability;
  ip.arrived_reward = c.arrived_reward;
  ip.d0 = c.idm_distance_wanted; ip.tau = c.idm_time_wanted; ip.a_max = c.idm_comfort_acc_max; ip.b_min = c.idm_comfort_acc_min;
}

// observation length per agent: V*F (Kinematics) or F*W*H (OccupancyGrid)
inline size_t obs_len(const hwy_config &c) {
  return c.obs_type == HWY_OBS_OCCUPANCY_GRID ? (size_t)c.obs_features * c.grid_shape[0] * c.grid_shape[1]
                                              : (size_t)c.obs_vehicles * c.obs_features;
}

// the 9 f64 planes of the device SoA live in one allocation, [field][E][pitch]
inline void bind_planes(double *f64, size_t plane, DevState &st) {
  st.x = f64 + 0 * plane; st.y = f64 + 1 * plane; st.heading = f64 + 2 * plane; st.speed = f64 + 3 * plane;
  st.timer = f64 + 4 * plane; st.target_speed = f64 + 5 * plane; st.delta = f64 + 6 * plane;
  st.impact_x = f64 + 7 * plane; st.impact_y = f64 + 8 * plane;
}

}  // namespace hwy

// hwy_launch.h -- host-visible launch functions of the kernels in hwy_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "hwy_device.h"
#include "hwy_net.h"
#include "hwy_ix.h"

namespace hwy {
// the events the launches of THIS THREAD record their dispatch begin / end timestamps into (nullptr, nullptr = none)
void set_launch_events(hipEvent_t start, hipEvent_t stop);
hipError_t launch_step(const StepParams &p, int num_envs, hipStream_t stream, int waves_per_eu, bool force_block_kernel,
                       int extra_lds);
// true where launch_step / launch_rollout run the two-vehicles-per-thread one-wavefront kernel of hwy_wave2.h
bool wide_kernel_applies(const StepParams &p, bool force_block_kernel);
// p.k_steps policy steps per launch (hwy_rollout_device): one-wavefront kernel (waves_per_eu, extra_lds) or workgroup kernel
hipError_t launch_rollout(const StepParams &p, int num_envs, hipStream_t stream, int waves_per_eu, int extra_lds,
                          bool force_block_kernel, int block_waves_per_eu);
// workgroups of the step kernel the device holds at once (0 = unknown / not applicable)
int step_resident_blocks(const StepParams &p, int waves_per_eu, bool force_block_kernel, int extra_lds);
int net_step_resident_blocks(int waves_per_eu);
hipError_t launch_reset(const StepParams &p, int num_envs, hipStream_t stream);
hipError_t launch_math_probe(int op, const double *in, double *out, long long n, hipStream_t stream);
hipError_t launch_observe(const StepParams &p, int num_envs, hipStream_t stream);
// road-network scenarios (hwy_net.h): one wavefront per environment
hipError_t launch_net_step(const NetParams &np, int num_envs, hipStream_t stream, int waves_per_eu);
hipError_t launch_net_rollout(const NetParams &np, int num_envs, hipStream_t stream, int waves_per_eu);  // np.s.k_steps steps per launch
hipError_t launch_net_reset(const NetParams &np, int num_envs, hipStream_t stream);
hipError_t launch_net_observe(const NetParams &np, int num_envs, hipStream_t stream);
// intersection scenario (hwy_ix.h): one wavefront per environment
hipError_t launch_ix_step(const IxParams &ip, int num_envs, hipStream_t stream, int waves_per_eu);
hipError_t launch_ix_rollout(const IxParams &ip, int num_envs, hipStream_t stream, int waves_per_eu);  // ip.s.k_steps steps per launch
hipError_t launch_ix_reset(const IxParams &ip, int num_envs, hipStream_t stream);
hipError_t launch_ix_observe(const IxParams &ip, int num_envs, hipStream_t stream);
}  // namespace hwy

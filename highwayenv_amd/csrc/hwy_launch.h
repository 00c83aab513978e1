// hwy_launch.h -- host-visible launch functions of the kernels in hwy_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "hwy_device.h"

namespace hwy {
hipError_t launch_step(const StepParams &p, int num_envs, hipStream_t stream, int waves_per_eu, bool force_block_kernel);
hipError_t launch_reset(const StepParams &p, int num_envs, hipStream_t stream);
hipError_t launch_math_probe(int op, const double *in, double *out, long long n, hipStream_t stream);
hipError_t launch_observe(const StepParams &p, int num_envs, hipStream_t stream);
}  // namespace hwy

// Developer TU: only the workgroup kernel of N = 193 .. 256 (HWY_ASM_SRC=tools/mini/block.hip)
#include <hip/hip_runtime.h>
#define HWY_HAVE_SETPRIO 1
#include "hwy_device.h"
namespace hwy {
template __global__ void hwy_step_kernel<4, 3>(const StepParams);
template __global__ void hwy_step_kernel<4, 4>(const StepParams);
}

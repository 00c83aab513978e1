// Developer TU: only the headline kernel (fast static-count turnaround for tools/asm_loop_stats.py; HWY_ASM_SRC=tools/mini/wave.hip)
#include <hip/hip_runtime.h>
#define HWY_HAVE_SETPRIO 1
#include "hwy_device.h"
#include "hwy_wave.h"
namespace hwy {
template __global__ void hwy_step_wave_kernel<4, false>(const StepParams);
template __global__ void hwy_step_wave_kernel<4, true>(const StepParams);
}

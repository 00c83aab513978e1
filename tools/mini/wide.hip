// Developer TU: only BASELINE config 3's kernel (HWY_ASM_SRC=tools/mini/wide.hip)
#include <hip/hip_runtime.h>
#define HWY_HAVE_SETPRIO 1
#include "hwy_device.h"
#include "hwy_wave.h"
#include "hwy_wave2.h"
namespace hwy {
template __global__ void hwy_step_wide_kernel<2, 2>(const StepParams);
}
namespace hwy {
template __global__ void hwy_step_wide_kernel<2, 1>(const StepParams);
}

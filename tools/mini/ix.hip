// Developer TU: only BASELINE config 4's kernel (HWY_ASM_SRC=tools/mini/ix.hip)
#include <hip/hip_runtime.h>
#define HWY_HAVE_SETPRIO 1
#include "hwy_device.h"
#include "hwy_wave.h"
#include "hwy_net.h"
#include "hwy_ix.h"
namespace hwy {
template __global__ void hwy_ix_step_kernel<2, 32, 64>(const IxParams);
}

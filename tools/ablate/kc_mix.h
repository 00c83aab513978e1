// A/B variant (tools/ablate/make_variants.py f_kcmix, f_fma_kcmix): the SGPR constant of a Horner step materialised by the
// COMPILER (two s_mov_b32 with literal operands, hazards known to it) instead of hwy_math.h's volatile `s_mov_b64 0` + two
// s_or_b32; same constants, same arithmetic.  Frame loop of the headline kernel, static: SALU 952 -> 881, VALU unchanged, no
// spill traffic (possible since -disable-machine-licm keeps materialisations where they are used).  Not timed yet.
#pragma once
#include <hip/hip_runtime.h>
namespace hwy {
template <unsigned long long BITS>
__device__ __forceinline__ double fma_k(double a, double b) {
  double r;
  const double k = __longlong_as_double((long long)BITS);
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
  return r;
}
}
#define HWY_KC(c) (c)
#define HWY_FMA_K(a, b, c) ::hwy::fma_k<__builtin_bit_cast(unsigned long long, (double)(c))>(a, b)

// hwy_math.h -- the five transcendental functions the step kernel needs, written for the bounded
// domains the simulation produces, in explicit-FMA f64.
//
// Why not ocml: the library routines are correct for every double (Payne-Hanek reduction for huge
// angles, subnormals, iterative fmod, ...).  Inlined into the fused step kernel those never-taken
// paths cost ~45 VGPRs and ~250 VALU instructions per vehicle-frame, which is what kept the kernel at
// 2 waves/SIMD.  The algorithms below are the classic fdlibm ones (e_log.c, e_exp.c, k_sin.c, k_cos.c,
// e_asin.c -- Sun Microsystems, freely distributable; coefficients are theirs), restated with fma()
// and with v_rcp_f64 / v_rsq_f64 + Newton instead of IEEE division / sqrt.  Each is within ~2 ulp of
// the correctly rounded result on its stated domain (tests/test_device_math.py checks that on the CPU
// emulation and, through hwy_debug_math, on the GPU), i.e. the same order as the libm-to-libm
// differences that exist between the reference's numpy, glibc and ocml anyway.
//
// Everything here is compiled -ffp-contract=off: only the explicit fma() calls fuse.
#pragma once

namespace hwy {

// A loop-invariant f64 constant that must NOT be hoisted out of the frame loop into a VGPR pair: the
// asm pins it to an SGPR pair materialised (two s_mov_b32) right where it is used.
#ifndef HWY_KC
__device__ inline double sgpr_const(double c) {
  asm volatile("" : "+s"(c));
  return c;
}
#define HWY_KC(c) ::hwy::sgpr_const(c)
#endif

// ---- reciprocal / reciprocal square root: hardware seed (~2^-26) + two Newton steps ----------------
__device__ inline double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}
__device__ inline double fast_rsqrt(double x) {  // x > 0, finite
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  double e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  e = fma(-h * y, y, 0.5);
  return fma(y, e, y);
}

// ---- Python / numpy float `%` for a positive modulus -------------------------------------------------
// a mod b = a - floor(a/b)*b, evaluated with one fma (exact for the |a/b| < 2^30 this code sees) and a
// one-step correction for the case where the rounded quotient lands on the wrong side of an integer.
__device__ inline double py_mod_pos(double a, double b) {
  if (a >= 0 && a < b) return a;  // the common case: already reduced
  const double k = floor(a * fast_rcp(b));
  double r = fma(-k, b, a);
  if (r < 0) r += b;
  else if (r >= b) r -= b;
  return r;
}

// ---- log(x), x positive normal ------------------------------------------------------------------------
__device__ inline double log_pos(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int hx = __double2hiint(x);
  const int lx = __double2loint(x);
  int k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int i = (hx + 0x95f64) & 0x100000;  // normalise the mantissa into [sqrt(1/2), sqrt(2))
  const double m = __hiloint2double(hx | (i ^ 0x3ff00000), lx);
  k += i >> 20;
  const double f = m - 1.0;
  const double s = f * fast_rcp(2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, HWY_KC(Lg6), HWY_KC(Lg4)), HWY_KC(Lg2));
  const double t2 = z * fma(w, fma(w, fma(w, HWY_KC(Lg7), HWY_KC(Lg5)), HWY_KC(Lg3)), HWY_KC(Lg1));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  return fma(dk, HWY_KC(ln2_hi), -((hfsq - fma(s, hfsq + R, dk * HWY_KC(ln2_lo))) - f));
}

// ---- exp(y), y <= 40 (results are finite, no overflow handling); y < -700 flushes to 0 --------------------
__device__ inline double exp_bounded(double y) {
  if (!(y > -700.0)) return 0.0;  // also y == -inf (log of a zero speed ratio)
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  const double k = rint(y * HWY_KC(invln2));
  const double hi = fma(-k, HWY_KC(ln2HI), y), lo = k * HWY_KC(ln2LO);
  const double r = hi - lo;
  const double t = r * r;
  const double c = fma(-t, fma(t, fma(t, fma(t, fma(t, HWY_KC(P5), HWY_KC(P4)), HWY_KC(P3)), HWY_KC(P2)), HWY_KC(P1)), r);
  const double e = 1.0 - ((lo - (r * c) * fast_rcp(2.0 - c)) - hi);
  // scale by 2^k: e in [0.7, 1.5], k in [-1010, 58] => the result is a normal double: add k to the exponent
  return __hiloint2double(__double2hiint(e) + ((int)k << 20), __double2loint(e));
}

// ---- sincos(x), |x| <= 2^20 (headings are O(1)); three-term Cody-Waite reduction by pi/2 ---------------------
__device__ inline void sincos_bounded(double x, double *sn, double *cs) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632679489655800e+00, pio2_2 = 6.12323399573676603587e-17, pio2_3 = -1.49738490485916983294e-33;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double n = rint(x * HWY_KC(invpio2));
  double r = fma(-n, HWY_KC(pio2_1), x);
  r = fma(-n, HWY_KC(pio2_2), r);
  r = fma(-n, HWY_KC(pio2_3), r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, HWY_KC(S6), HWY_KC(S5)), HWY_KC(S4)), HWY_KC(S3)), HWY_KC(S2));
  const double s = fma(z * r, fma(z, ps, HWY_KC(S1)), r);
  const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, HWY_KC(C6), HWY_KC(C5)), HWY_KC(C4)), HWY_KC(C3)), HWY_KC(C2)), HWY_KC(C1));
  const double c = 1.0 - fma(-z, pc, 0.5 * z);
  const int q = (int)n & 3;
  const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}

// ---- asin(x), |x| <= 1 -----------------------------------------------------------------------------------------
__device__ inline double asin_rational(double t) {  // R(t) = t*P(t)/Q(t), asin(x) = x + x*R(x^2) on |x| <= 0.5
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  const double pp = t * fma(t, fma(t, fma(t, fma(t, fma(t, HWY_KC(pS5), HWY_KC(pS4)), HWY_KC(pS3)), HWY_KC(pS2)), HWY_KC(pS1)), HWY_KC(pS0));
  const double qq = fma(t, fma(t, fma(t, fma(t, HWY_KC(qS4), HWY_KC(qS3)), HWY_KC(qS2)), HWY_KC(qS1)), 1.0);
  return pp * fast_rcp(qq);
}
__device__ inline double asin_bounded(double x) {
  const double ax = fabs(x);
  if (ax <= 0.5) return fma(x, asin_rational(x * x), x);
  // asin(x) = pi/2 - 2*asin(sqrt((1-|x|)/2))
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
  const double t = (1.0 - ax) * 0.5;
  const double s = t > 0.0 ? t * fast_rsqrt(t) : 0.0;
  const double r = HWY_KC(pio2_hi) - (2.0 * fma(s, asin_rational(t), s) - HWY_KC(pio2_lo));
  return x < 0 ? -r : r;
}

}  // namespace hwy

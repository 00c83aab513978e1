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

// A polynomial coefficient is the constant-bus operand of its VALU instruction: an aligned SGPR pair that the COMPILER
// materialises where it is used (two s_mov_b32 with literal operands; it knows their hazards).  The build runs without
// MachineLICM (build.py: -disable-machine-licm), so nothing hoists these materialisations out of the frame loop into VGPR
// pairs (that cost +54 VGPRs) or into SGPR pairs that get spilled to VGPR lanes.  (Rounds 1-3 forced the same with a volatile
// `s_mov_b64 0` + two s_or_b32 per constant; handing the plain value to the asm below saves one SALU instruction per
// constant: frame loop of the headline kernel 952 -> 881 SALU, measured -2.4 % on the headline launch, -1.5 .. -4.5 % on the
// other four workloads, profiles/r04_history.md.)
#ifndef HWY_KC
#define HWY_KC(c) (c)
// a*b + K (fused) with the constant K as the constant-bus operand of a VOP3 v_fma_f64.  Written as asm because
// the compiler selects the two-address v_fmac form for a Horner step, whose addend must live in a VGPR pair:
// it would copy K there first (two more VALU ops per coefficient).
template <unsigned long long BITS>
__device__ __forceinline__ double fma_k(double a, double b) {
  double r;
  const double k = __longlong_as_double((long long)BITS);
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
  return r;
}
#define HWY_FMA_K(a, b, c) ::hwy::fma_k<__builtin_bit_cast(unsigned long long, (double)(c))>(a, b)
#endif
// Horner: poly(t; K_n ... K_0) with the two leading coefficients combined as t*K_n + K_(n-1) (two constants
// cannot share one VALU instruction: one constant-bus read per instruction on gfx9)
__device__ __forceinline__ double lead_unfused(double t, double kn, double kn1) {
#pragma clang fp contract(off)
  return t * kn + kn1;
}
#define HWY_LEAD(t, kn, kn1) ::hwy::lead_unfused(t, HWY_KC(kn), HWY_KC(kn1))

// ---- the same, for TWO independent evaluations (hwy_wave2.h: the two vehicles of a thread): one asm statement holds both
// instructions, so the coefficient's SGPR pair is materialised ONCE and the two dependent chains are interleaved by
// construction.  (Two calls of the scalar routine leave both to the compiler: under the SGPR pressure of the step kernels it
// re-materialises every coefficient for the second chain -- SALU instructions a lone wavefront per SIMD pays in full -- and
// emits one chain after the other.)  Same operations on the same values: bit-identical to two scalar calls.
#ifndef HWY_FMA_K2
template <unsigned long long BITS>
__device__ __forceinline__ void fma_k2(double a0, double b0, double a1, double b1, double &r0, double &r1) {
  const double k = __longlong_as_double((long long)BITS);
  asm("v_fma_f64 %0, %2, %3, %6\n\tv_fma_f64 %1, %4, %5, %6" : "=&v"(r0), "=v"(r1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "s"(k));
}
// t * KN + KN1 unfused (lead_unfused), both constants shared
template <unsigned long long KN, unsigned long long KN1>
__device__ __forceinline__ void lead_k2(double t0, double t1, double &r0, double &r1) {
  const double kn = __longlong_as_double((long long)KN), kn1 = __longlong_as_double((long long)KN1);
  double m0, m1;
  asm("v_mul_f64 %0, %2, %4\n\tv_mul_f64 %1, %3, %4" : "=&v"(m0), "=v"(m1) : "v"(t0), "v"(t1), "s"(kn));
  asm("v_add_f64 %0, %2, %4\n\tv_add_f64 %1, %3, %4" : "=&v"(r0), "=v"(r1) : "v"(m0), "v"(m1), "s"(kn1));
}
#define HWY_FMA_K2(r0, r1, a0, b0, a1, b1, c) ::hwy::fma_k2<__builtin_bit_cast(unsigned long long, (double)(c))>(a0, b0, a1, b1, r0, r1)
#define HWY_LEAD2(r0, r1, t0, t1, kn, kn1) \
  ::hwy::lead_k2<__builtin_bit_cast(unsigned long long, (double)(kn)), __builtin_bit_cast(unsigned long long, (double)(kn1))>(t0, t1, r0, r1)
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
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
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
  const double t1 = w * HWY_FMA_K(w, HWY_LEAD(w, Lg6, Lg4), Lg2);
  const double t2 = z * HWY_FMA_K(w, HWY_FMA_K(w, HWY_LEAD(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  return fma(dk, HWY_KC(ln2_hi), -((hfsq - fma(s, hfsq + R, dk * HWY_KC(ln2_lo))) - f));
}

// ---- exp(y), y <= 40 (results are finite, no overflow handling); y < -700 flushes to 0 --------------------
__device__ inline double exp_bounded(double y) {
  if (!(y > -700.0)) return 0.0;  // also y == -inf (log of a zero speed ratio)
  constexpr double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
  constexpr double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  const double k = rint(y * HWY_KC(invln2));
  const double hi = fma(-k, HWY_KC(ln2HI), y), lo = k * HWY_KC(ln2LO);
  const double r = hi - lo;
  const double t = r * r;
  const double c = fma(-t, HWY_FMA_K(t, HWY_FMA_K(t, HWY_FMA_K(t, HWY_LEAD(t, P5, P4), P3), P2), P1), r);
  const double e = 1.0 - ((lo - (r * c) * fast_rcp(2.0 - c)) - hi);
  // scale by 2^k: e in [0.7, 1.5], k in [-1010, 58] => the result is a normal double: add k to the exponent
  return __hiloint2double(__double2hiint(e) + ((int)k << 20), __double2loint(e));
}

// ---- sincos(x), |x| <= 2^20 (headings are O(1)); three-term Cody-Waite reduction by pi/2 ---------------------
__device__ inline void sincos_bounded(double x, double *sn, double *cs) {
  constexpr double invpio2 = 6.36619772367581382433e-01;
  constexpr double pio2_1 = 1.57079632679489655800e+00, pio2_2 = 6.12323399573676603587e-17, pio2_3 = -1.49738490485916983294e-33;
  constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double n = rint(x * HWY_KC(invpio2));
  double r = fma(-n, HWY_KC(pio2_1), x);
  r = fma(-n, HWY_KC(pio2_2), r);
  r = fma(-n, HWY_KC(pio2_3), r);
  const double z = r * r;
  const double ps = HWY_FMA_K(z, HWY_FMA_K(z, HWY_FMA_K(z, HWY_LEAD(z, S6, S5), S4), S3), S2);
  const double s = fma(z * r, HWY_FMA_K(z, ps, S1), r);
  const double pc = z * HWY_FMA_K(z, HWY_FMA_K(z, HWY_FMA_K(z, HWY_FMA_K(z, HWY_LEAD(z, C6, C5), C4), C3), C2), C1);
  const double c = 1.0 - fma(-z, pc, 0.5 * z);
  const int q = (int)n & 3;
  const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}

// ---- asin(x), |x| <= 1 -----------------------------------------------------------------------------------------
__device__ inline double asin_rational(double t) {  // R(t) = t*P(t)/Q(t), asin(x) = x + x*R(x^2) on |x| <= 0.5
  constexpr double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  const double pp = t * HWY_FMA_K(t, HWY_FMA_K(t, HWY_FMA_K(t, HWY_FMA_K(t, HWY_LEAD(t, pS5, pS4), pS3), pS2), pS1), pS0);
  const double qq = fma(t, HWY_FMA_K(t, HWY_FMA_K(t, HWY_LEAD(t, qS4, qS3), qS2), qS1), 1.0);
  return pp * fast_rcp(qq);
}
__device__ inline double asin_bounded(double x) {
  const double ax = fabs(x);
  if (ax <= 0.5) return fma(x, asin_rational(x * x), x);
  // asin(x) = pi/2 - 2*asin(sqrt((1-|x|)/2))
  constexpr double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
  const double t = (1.0 - ax) * 0.5;
  const double s = t > 0.0 ? t * fast_rsqrt(t) : 0.0;
  const double r = HWY_KC(pio2_hi) - (2.0 * fma(s, asin_rational(t), s) - HWY_KC(pio2_lo));
  return x < 0 ? -r : r;
}

// ---- atan(x), any finite x: fdlibm s_atan.c (argument reduction to |t| < 7/16 around 0.5, 1, 1.5, inf; the four
//      branches evaluated as selects so that a wave pays one polynomial, not four) ------------------------------
__device__ inline double atan_fd(double x) {
  const double ax = fabs(x);
  // id: -1 (|x| < 7/16), 0 (< 11/16), 1 (< 19/16), 2 (< 39/16), 3 (>= 39/16)
  const bool r0 = ax < 0.4375, r1 = ax < 0.6875, r2 = ax < 1.1875, r3 = ax < 2.4375;
  const double num = r0 ? ax : r1 ? 2.0 * ax - 1.0 : r2 ? ax - 1.0 : r3 ? ax - 1.5 : -1.0;
  const double den = r0 ? 1.0 : r1 ? 2.0 + ax : r2 ? ax + 1.0 : r3 ? 1.0 + 1.5 * ax : ax;
  const double t = r0 ? ax : num * fast_rcp(den);
  const double hi = r0 ? 0.0 : r1 ? 4.63647609000806093515e-01 : r2 ? 7.85398163397448278999e-01
                  : r3 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00;
  const double lo = r0 ? 0.0 : r1 ? 2.26987774529616870924e-17 : r2 ? 3.06161699786838301793e-17
                  : r3 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17;
  const double z = t * t, w = z * z;
  constexpr double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
               aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
               aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
  const double s1 = z * HWY_FMA_K(w, HWY_FMA_K(w, HWY_FMA_K(w, HWY_FMA_K(w, fma(w, HWY_KC(aT10), HWY_KC(aT8)), aT6), aT4), aT2), aT0);
  const double s2 = w * HWY_FMA_K(w, HWY_FMA_K(w, HWY_FMA_K(w, fma(w, HWY_KC(aT9), HWY_KC(aT7)), aT5), aT3), aT1);
  const double r = r0 ? t - t * (s1 + s2) : hi - ((t * (s1 + s2) - lo) - t);
  return x < 0 ? -r : r;
}
// ---- atan2(y, x) for finite arguments of moderate ratio (positions a few hundred metres apart): fdlibm e_atan2.c
//      without the inf / nan / huge-ratio cases (|y/x| beyond 2^60 cannot happen for this code's inputs unless x == 0,
//      which is handled) --------------------------------------------------------------------------------------------------
__device__ inline double atan2_bounded(double y, double x) {
  const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
  if (x == 0.0) return y == 0.0 ? (y) : (y > 0 ? pi / 2 : -pi / 2);
  const double z = atan_fd(fabs(y * fast_rcp(x)));
  if (x > 0) return y < 0 ? -z : z;
  return y < 0 ? (z - pi_lo) - pi : pi - (z - pi_lo);
}


// =====================================================================================================================
// Paired versions (two arguments, two results) of the four routines every vehicle-frame runs: statement by statement the
// scalar routine above, each statement for both arguments.  tests/test_device_math.py compares them bit for bit.
__device__ inline void fast_rcp2(double x0, double x1, double &y0, double &y1) {
  y0 = __builtin_amdgcn_rcp(x0); y1 = __builtin_amdgcn_rcp(x1);
  double e0 = fma(-x0, y0, 1.0), e1 = fma(-x1, y1, 1.0);
  y0 = fma(y0, e0, y0); y1 = fma(y1, e1, y1);
  e0 = fma(-x0, y0, 1.0); e1 = fma(-x1, y1, 1.0);
  y0 = fma(y0, e0, y0); y1 = fma(y1, e1, y1);
}
__device__ inline void fast_rsqrt2(double x0, double x1, double &y0, double &y1) {
  y0 = __builtin_amdgcn_rsq(x0); y1 = __builtin_amdgcn_rsq(x1);
  const double h0 = 0.5 * x0, h1 = 0.5 * x1;
  double e0 = fma(-h0 * y0, y0, 0.5), e1 = fma(-h1 * y1, y1, 0.5);
  y0 = fma(y0, e0, y0); y1 = fma(y1, e1, y1);
  e0 = fma(-h0 * y0, y0, 0.5); e1 = fma(-h1 * y1, y1, 0.5);
  y0 = fma(y0, e0, y0); y1 = fma(y1, e1, y1);
}
__device__ inline void log_pos2(double x0, double x1, double &o0, double &o1) {
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int hx0 = __double2hiint(x0), hx1 = __double2hiint(x1);
  const int lx0 = __double2loint(x0), lx1 = __double2loint(x1);
  int k0 = (hx0 >> 20) - 1023, k1 = (hx1 >> 20) - 1023;
  hx0 &= 0x000fffff; hx1 &= 0x000fffff;
  const int i0 = (hx0 + 0x95f64) & 0x100000, i1 = (hx1 + 0x95f64) & 0x100000;
  const double m0 = __hiloint2double(hx0 | (i0 ^ 0x3ff00000), lx0), m1 = __hiloint2double(hx1 | (i1 ^ 0x3ff00000), lx1);
  k0 += i0 >> 20; k1 += i1 >> 20;
  const double f0 = m0 - 1.0, f1 = m1 - 1.0;
  double q0, q1;
  fast_rcp2(2.0 + f0, 2.0 + f1, q0, q1);
  const double s0 = f0 * q0, s1 = f1 * q1;
  const double z0 = s0 * s0, z1 = s1 * s1, w0 = z0 * z0, w1 = z1 * z1;
  double a0, a1, b0, b1;
  HWY_LEAD2(a0, a1, w0, w1, Lg6, Lg4);
  HWY_FMA_K2(a0, a1, w0, a0, w1, a1, Lg2);
  const double t10 = w0 * a0, t11 = w1 * a1;
  HWY_LEAD2(b0, b1, w0, w1, Lg7, Lg5);
  HWY_FMA_K2(b0, b1, w0, b0, w1, b1, Lg3);
  HWY_FMA_K2(b0, b1, w0, b0, w1, b1, Lg1);
  const double t20 = z0 * b0, t21 = z1 * b1;
  const double R0 = t20 + t10, R1 = t21 + t11;
  const double hf0 = 0.5 * f0 * f0, hf1 = 0.5 * f1 * f1;
  const double dk0 = (double)k0, dk1 = (double)k1;
  o0 = fma(dk0, HWY_KC(ln2_hi), -((hf0 - fma(s0, hf0 + R0, dk0 * HWY_KC(ln2_lo))) - f0));
  o1 = fma(dk1, HWY_KC(ln2_hi), -((hf1 - fma(s1, hf1 + R1, dk1 * HWY_KC(ln2_lo))) - f1));
}
__device__ inline void exp_bounded2(double y0, double y1, double &o0, double &o1) {
  constexpr double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
  constexpr double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  const bool z0_ = !(y0 > -700.0), z1_ = !(y1 > -700.0);  // (flushed to 0 below; evaluated on 0 so that nothing overflows)
  const double u0 = z0_ ? 0.0 : y0, u1 = z1_ ? 0.0 : y1;
  const double k0 = rint(u0 * HWY_KC(invln2)), k1 = rint(u1 * HWY_KC(invln2));
  const double hi0 = fma(-k0, HWY_KC(ln2HI), u0), hi1 = fma(-k1, HWY_KC(ln2HI), u1);
  const double lo0 = k0 * HWY_KC(ln2LO), lo1 = k1 * HWY_KC(ln2LO);
  const double r0 = hi0 - lo0, r1 = hi1 - lo1;
  const double t0 = r0 * r0, t1 = r1 * r1;
  double a0, a1;
  HWY_LEAD2(a0, a1, t0, t1, P5, P4);
  HWY_FMA_K2(a0, a1, t0, a0, t1, a1, P3);
  HWY_FMA_K2(a0, a1, t0, a0, t1, a1, P2);
  HWY_FMA_K2(a0, a1, t0, a0, t1, a1, P1);
  const double c0 = fma(-t0, a0, r0), c1 = fma(-t1, a1, r1);
  double q0, q1;
  fast_rcp2(2.0 - c0, 2.0 - c1, q0, q1);
  const double e0 = 1.0 - ((lo0 - (r0 * c0) * q0) - hi0), e1 = 1.0 - ((lo1 - (r1 * c1) * q1) - hi1);
  const double v0 = __hiloint2double(__double2hiint(e0) + ((int)k0 << 20), __double2loint(e0));
  const double v1 = __hiloint2double(__double2hiint(e1) + ((int)k1 << 20), __double2loint(e1));
  o0 = z0_ ? 0.0 : v0;
  o1 = z1_ ? 0.0 : v1;
}
__device__ inline void sincos_bounded2(double x0, double x1, double *sn0, double *cs0, double *sn1, double *cs1) {
  constexpr double invpio2 = 6.36619772367581382433e-01;
  constexpr double pio2_1 = 1.57079632679489655800e+00, pio2_2 = 6.12323399573676603587e-17, pio2_3 = -1.49738490485916983294e-33;
  constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double n0 = rint(x0 * HWY_KC(invpio2)), n1 = rint(x1 * HWY_KC(invpio2));
  double r0 = fma(-n0, HWY_KC(pio2_1), x0), r1 = fma(-n1, HWY_KC(pio2_1), x1);
  r0 = fma(-n0, HWY_KC(pio2_2), r0); r1 = fma(-n1, HWY_KC(pio2_2), r1);
  r0 = fma(-n0, HWY_KC(pio2_3), r0); r1 = fma(-n1, HWY_KC(pio2_3), r1);
  const double z0 = r0 * r0, z1 = r1 * r1;
  double p0, p1, c0, c1;
  HWY_LEAD2(p0, p1, z0, z1, S6, S5);
  HWY_FMA_K2(p0, p1, z0, p0, z1, p1, S4);
  HWY_FMA_K2(p0, p1, z0, p0, z1, p1, S3);
  HWY_FMA_K2(p0, p1, z0, p0, z1, p1, S2);
  HWY_FMA_K2(p0, p1, z0, p0, z1, p1, S1);
  const double s0 = fma(z0 * r0, p0, r0), s1 = fma(z1 * r1, p1, r1);
  HWY_LEAD2(c0, c1, z0, z1, C6, C5);
  HWY_FMA_K2(c0, c1, z0, c0, z1, c1, C4);
  HWY_FMA_K2(c0, c1, z0, c0, z1, c1, C3);
  HWY_FMA_K2(c0, c1, z0, c0, z1, c1, C2);
  HWY_FMA_K2(c0, c1, z0, c0, z1, c1, C1);
  const double pc0 = z0 * c0, pc1 = z1 * c1;
  const double co0 = 1.0 - fma(-z0, pc0, 0.5 * z0), co1 = 1.0 - fma(-z1, pc1, 0.5 * z1);
  const int q0 = (int)n0 & 3, q1 = (int)n1 & 3;
  const double ss0 = (q0 & 1) ? co0 : s0, cc0 = (q0 & 1) ? s0 : co0;
  const double ss1 = (q1 & 1) ? co1 : s1, cc1 = (q1 & 1) ? s1 : co1;
  *sn0 = (q0 & 2) ? -ss0 : ss0; *cs0 = ((q0 + 1) & 2) ? -cc0 : cc0;
  *sn1 = (q1 & 2) ? -ss1 : ss1; *cs1 = ((q1 + 1) & 2) ? -cc1 : cc1;
}
__device__ inline void asin_rational2(double t0, double t1, double &o0, double &o1) {
  constexpr double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  double p0, p1, q0, q1;
  HWY_LEAD2(p0, p1, t0, t1, pS5, pS4);
  HWY_FMA_K2(p0, p1, t0, p0, t1, p1, pS3);
  HWY_FMA_K2(p0, p1, t0, p0, t1, p1, pS2);
  HWY_FMA_K2(p0, p1, t0, p0, t1, p1, pS1);
  HWY_FMA_K2(p0, p1, t0, p0, t1, p1, pS0);
  const double pp0 = t0 * p0, pp1 = t1 * p1;
  HWY_LEAD2(q0, q1, t0, t1, qS4, qS3);
  HWY_FMA_K2(q0, q1, t0, q0, t1, q1, qS2);
  HWY_FMA_K2(q0, q1, t0, q0, t1, q1, qS1);
  const double qq0 = fma(t0, q0, 1.0), qq1 = fma(t1, q1, 1.0);
  double i0, i1;
  fast_rcp2(qq0, qq1, i0, i1);
  o0 = pp0 * i0;
  o1 = pp1 * i1;
}
// (both branches of asin_bounded share ONE evaluation of the rational on the branch's own argument; the two epilogues are cheap)
__device__ inline void asin_bounded2(double x0, double x1, double &o0, double &o1) {
  constexpr double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
  const double ax0 = fabs(x0), ax1 = fabs(x1);
  const bool sm0 = ax0 <= 0.5, sm1 = ax1 <= 0.5;
  const double tb0 = (1.0 - ax0) * 0.5, tb1 = (1.0 - ax1) * 0.5;
  const double t0 = sm0 ? x0 * x0 : tb0, t1 = sm1 ? x1 * x1 : tb1;
  double R0, R1;
  asin_rational2(t0, t1, R0, R1);
  double rs0, rs1;
  fast_rsqrt2(tb0 > 0.0 ? tb0 : 1.0, tb1 > 0.0 ? tb1 : 1.0, rs0, rs1);
  const double s0 = tb0 > 0.0 ? tb0 * rs0 : 0.0, s1 = tb1 > 0.0 ? tb1 * rs1 : 0.0;
  const double big0 = HWY_KC(pio2_hi) - (2.0 * fma(s0, R0, s0) - HWY_KC(pio2_lo));
  const double big1 = HWY_KC(pio2_hi) - (2.0 * fma(s1, R1, s1) - HWY_KC(pio2_lo));
  o0 = sm0 ? fma(x0, R0, x0) : (x0 < 0 ? -big0 : big0);
  o1 = sm1 ? fma(x1, R1, x1) : (x1 < 0 ? -big1 : big1);
}

}  // namespace hwy

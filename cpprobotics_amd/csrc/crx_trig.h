// crx_trig.h — single-precision sine/cosine for the crx engine (host + gfx950 device).
//
// The reference evaluates std::cos(float)/std::sin(float) through glibc's libm
// (/root/reference/src/extended_kalman_filter.cpp:30-31,42-45).  OCML's device sinf/cosf
// round differently from glibc's in last-ulp cases, which would break bit parity between
// the HIP path and the CPU oracle.  This header therefore carries ONE implementation of the
// published algorithm glibc >= 2.28 uses for sinf/cosf (Szabolcs Nagy / Wilco Dijkstra,
// "optimized-routines": double-precision minimax polynomials on [-pi/4, pi/4] after a
// one-multiply range reduction; 192-bit 2/pi table for |x| >= 120).
//
// Flavour.  On x86-64 glibc dispatches sinf/cosf at run time (ifunc) to a build with FMA
// contraction on every FMA-capable CPU, and to a plain SSE2 build otherwise; the two differ
// on 34 of the 2^32 inputs.  This header writes the contraction out with explicit fma()
// calls (CRX_TRIG_FMA=1, default; independent of -ffp-contract) and is bit-identical to
// glibc 2.35's FMA variant on ALL 2^32 float inputs; with CRX_TRIG_FMA=0 it is bit-identical
// to the SSE2 variant on all 2^32 inputs (tests/tools/trig_exhaustive.cpp walks the full
// domain; tests/test_trig.py runs a strided subset plus the 34 discriminating inputs).
//
// Nothing in here touches memory besides two small constant tables.
//
// Origin of the algorithm and constants: Arm Optimized Routines, math/sinf.c, cosf.c, sincosf.h, sincosf_data.c —
//   Copyright (c) 2018, Arm Limited.  SPDX-License-Identifier: MIT
// as incorporated into glibc 2.28+ (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h; LGPL-2.1-or-later).
// This header is an independent restatement of that published algorithm for host + gfx950; the notice above is kept as the
// MIT licence asks.
#pragma once
#include <stdint.h>

#ifndef CRX_TRIG_FMA
#define CRX_TRIG_FMA 1
#endif

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define CRX_HD __host__ __device__ __forceinline__
#else
#define CRX_HD static inline
#endif

namespace crx {

// a*b + c with the glibc-variant's rounding: one rounding (fma) or two (mul then add).
CRX_HD double mad_(double a, double b, double c) {
#if CRX_TRIG_FMA
  return __builtin_fma(a, b, c);
#else
  return a * b + c;
#endif
}

CRX_HD uint32_t f32_bits(float x) {
  union { float f; uint32_t u; } v; v.f = x; return v.u;
}
CRX_HD uint32_t abstop12(float x) { return (f32_bits(x) >> 20) & 0x7ffu; }

// Polynomial / reduction constants (doubles written as hex literals so host and device
// see the same bits).
struct SinCosConsts {
  // cosine polynomial c0..c4 in x^2, sine polynomial s1..s3 (x + x^3*s1 + x^5*s2 + x^7*s3)
  static constexpr double c0 = 0x1p0;
  static constexpr double c1 = -0x1.ffffffd0c621cp-2;
  static constexpr double c2 = 0x1.55553e1068f19p-5;
  static constexpr double c3 = -0x1.6c087e89a359dp-10;
  static constexpr double c4 = 0x1.99343027bf8c3p-16;
  static constexpr double s1 = -0x1.555545995a603p-3;
  static constexpr double s2 = 0x1.1107605230bc4p-7;
  static constexpr double s3 = -0x1.994eb3774cf24p-13;
  static constexpr double hpi_inv = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
  static constexpr double hpi = 0x1.921FB54442D18p0;        // pi/2
  static constexpr double pi63 = 0x1.921FB54442D18p-62;     // 2*pi * 2^-64
};

// Evaluate sin (quadrant even) or cos (quadrant odd) polynomial; `neg` selects the negated
// cosine polynomial (quadrants 2,3), exactly as negating every coefficient does.
CRX_HD float sincos_poly(double x, double x2, int n, bool neg) {
  typedef SinCosConsts C;
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double s1 = mad_(x2, C::s3, C::s2);
    double x7 = x3 * x2;
    double s = mad_(x3, C::s1, x);
    return (float)mad_(x7, s1, s);
  } else {
    const double k0 = neg ? -C::c0 : C::c0;
    const double k1 = neg ? -C::c1 : C::c1;
    const double k2 = neg ? -C::c2 : C::c2;
    const double k3 = neg ? -C::c3 : C::c3;
    const double k4 = neg ? -C::c4 : C::c4;
    double x4 = x2 * x2;
    double c2 = mad_(x2, k4, k3);
    double c1 = mad_(x2, k1, k0);
    double x6 = x4 * x2;
    double c = mad_(x4, k2, c1);
    return (float)mad_(x6, c2, c);
  }
}

CRX_HD double reduce_fast(double x, int* np) {
  typedef SinCosConsts C;
  double r = x * C::hpi_inv;
  int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return mad_(-(double)n, C::hpi, x);
}

// 2/pi to 192 bits, 8 new bits per entry (entry i = 32-bit window ending at byte i).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#endif
static const uint32_t kInvPio4[24] = {
  0x000000a2u, 0x0000a2f9u, 0x00a2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u,
  0x6e4e4415u, 0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u,
  0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u,
  0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};

CRX_HD double reduce_large(uint32_t xi, int* np) {
  const uint32_t* arr = &kInvPio4[(xi >> 26) & 15];
  int shift = (xi >> 23) & 7;
  uint64_t n, res0, res1, res2;
  xi = (xi & 0xffffffu) | 0x800000u;
  xi <<= shift;
  res0 = (uint64_t)(uint32_t)(xi * arr[0]);
  res1 = (uint64_t)xi * arr[4];
  res2 = (uint64_t)xi * arr[8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  n = (res0 + (1ULL << 61)) >> 62;
  res0 -= n << 62;
  double x = (double)(int64_t)res0;
  *np = (int)n;
  return x * SinCosConsts::pi63;
}

CRX_HD double quadrant_sign(int q) {  // {1,-1,-1,1}[q&3]
  q &= 3;
  return (q == 1 || q == 2) ? -1.0 : 1.0;
}

CRX_HD float sinf_(float y) {
  double x = (double)y;
  int n;
  if (abstop12(y) < 0x3f4u) {                 // |y| < pi/4 (top12 of 0x1.921FB6p-1f)
    double s = x * x;
    if (abstop12(y) < 0x398u) return y;       // |y| < 2^-12
    return sincos_poly(x, s, 0, false);
  } else if (abstop12(y) < 0x42fu) {          // |y| < 120
    x = reduce_fast(x, &n);
    double s = quadrant_sign(n);
    return sincos_poly(x * s, x * x, n, (n & 2) != 0);
  } else if (abstop12(y) < 0x7f8u) {          // finite
    uint32_t xi = f32_bits(y);
    int sign = (int)(xi >> 31);
    x = reduce_large(xi, &n);
    double s = quadrant_sign(n + sign);
    return sincos_poly(x * s, x * x, n, ((n + sign) & 2) != 0);
  }
  return y - y;                               // inf/nan -> nan
}

CRX_HD float cosf_(float y) {
  double x = (double)y;
  int n;
  if (abstop12(y) < 0x3f4u) {
    double x2 = x * x;
    if (abstop12(y) < 0x398u) return 1.0f;
    return sincos_poly(x, x2, 1, false);
  } else if (abstop12(y) < 0x42fu) {
    x = reduce_fast(x, &n);
    double s = quadrant_sign(n);
    return sincos_poly(x * s, x * x, n ^ 1, (n & 2) != 0);
  } else if (abstop12(y) < 0x7f8u) {
    uint32_t xi = f32_bits(y);
    int sign = (int)(xi >> 31);
    x = reduce_large(xi, &n);
    double s = quadrant_sign(n + sign);
    return sincos_poly(x * s, x * x, n ^ 1, ((n + sign) & 2) != 0);
  }
  return y - y;
}

// Both at once, branch-free on the common path (|y| < 120): one range reduction, each polynomial
// evaluated once, quadrant handled with selects.  Bit-identical to sinf_/cosf_ above: the sine
// polynomial is odd in its argument and the "negated table" of the cosine polynomial is an exact
// sign flip, so applying the quadrant sign after the polynomial commutes with every rounding.
#if defined(__HIP_DEVICE_COMPILE__)
// sincosf_ for a whole wave of ordinary angles (2^-100 <= |y| < 120): one angle of the fused EKF step's sincos_fast2 (csrc/ekf_math.h, where
// the proof is described) — n = rint(y * 2/pi) by one fma against 1.5 * 2^52, Horner polynomials, the quadrant logic as v_bfe / v_bitop3.
// NOT glibc's operations, glibc's RESULTS: tests/tools/trig_fast_exhaustive.cpp walks every float of the domain (run by the CPU tests).
// ~23 VALU instructions against ~45 of the general path below with its compares and double selects.
__device__ __forceinline__ void sincosf_wave_fast_(float y, float* sp, float* cp) {
  typedef SinCosConsts C;
  constexpr double two_over_pi = C::hpi_inv * 0x1p-24, magic = 0x1.8p52;
  const double xd = (double)y;
  const double t = __builtin_fma(xd, two_over_pi, magic);
  const double x = __builtin_fma(-(t - magic), C::hpi, xd);
  const double x2 = x * x, x3 = x * x2;
  const double ps = __builtin_fma(x2, __builtin_fma(x2, C::s3, C::s2), C::s1);
  const double pc = __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, C::c4, C::c3), C::c2), C::c1);
  const uint32_t fs = f32_bits((float)__builtin_fma(x3, ps, x)), fc = f32_bits((float)__builtin_fma(x2, pc, C::c0));
  const uint32_t n = (uint32_t)__double2loint(t);
  const uint32_t odd = (uint32_t)__builtin_amdgcn_sbfe((int)n, 0u, 1u);
  const uint32_t sr = __builtin_amdgcn_bitop3_b32(odd, fc, fs, 0xCA), cr = __builtin_amdgcn_bitop3_b32(odd, fs, fc, 0xCA);
  const uint32_t qs = n << 30;
  *sp = __uint_as_float(__builtin_amdgcn_bitop3_b32(sr, qs, 0x80000000u, 0x78));
  *cp = __uint_as_float(__builtin_amdgcn_bitop3_b32(cr, qs + 0x40000000u, 0x80000000u, 0x78));
}
#endif

CRX_HD void sincosf_(float y, float* sp, float* cp) {
  typedef SinCosConsts C;
  const uint32_t top = abstop12(y);
  double x = (double)y;
  int n, q;
  if (top < 0x42fu) {                     // |y| < 120 (includes |y| < pi/4, where n == 0)
    x = reduce_fast(x, &n);
    q = n;
  } else if (top < 0x7f8u) {              // large finite argument: 192-bit 2/pi reduction
    const uint32_t xi = f32_bits(y);
    x = reduce_large(xi, &n);
    q = n + (int)(xi >> 31);
  } else { *sp = y - y; *cp = y - y; return; }
  const double x2 = x * x;
  // sine polynomial  x + x^3*s1 + x^7*(s2 + x^2*s3)   (operation order of sincos_poly)
  const double x3 = x * x2;
  const double s1 = mad_(x2, C::s3, C::s2);
  const double x7 = x3 * x2;
  const double sa = mad_(x3, C::s1, x);
  const double S = mad_(x7, s1, sa);
  // cosine polynomial (c0 + x^2*c1) + x^4*c2 + x^6*(c3 + x^2*c4)
  const double x4 = x2 * x2;
  const double c2 = mad_(x2, C::c4, C::c3);
  const double c1 = mad_(x2, C::c1, C::c0);
  const double x6 = x4 * x2;
  const double ca = mad_(x4, C::c2, c1);
  const double Cv = mad_(x6, c2, ca);
  const bool neg_s = ((q + 1) & 2) != 0;  // sign {1,-1,-1,1}[q&3] of the sine branch
  const bool neg_c = (q & 2) != 0;        // negated cosine table for quadrants 2,3
  const float fs = (float)(neg_s ? -S : S);
  const float fc = (float)(neg_c ? -Cv : Cv);
  const bool odd = (n & 1) != 0;
  float so = odd ? fc : fs;
  float co = odd ? fs : fc;
  if (top < 0x398u) { so = y; co = 1.0f; }  // |y| < 2^-12: sinf returns y, cosf returns 1
  *sp = so;
  *cp = co;
}

// sincosf_ for kernels whose waves evaluate it on ordinary angles together (the dynamic-window planner's trajectory roll-out: +6.6 %):
// every active lane inside 2^-100 <= |y| < 120 -> the short form above, otherwise every lane takes the general one.  Same bits either
// way.  (Inside sincosf_ itself the test and the second inlined body cost the particle filter 8 %: opt-in per call site.)
CRX_HD void sincosf_wave_(float y, float* sp, float* cp) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float ay = __builtin_fabsf(y);
  if (__builtin_amdgcn_ballot_w64(!(ay >= 0x1p-100f && ay < 120.0f)) == 0) { sincosf_wave_fast_(y, sp, cp); return; }
#endif
  sincosf_(y, sp, cp);
}

// ---- expf -------------------------------------------------------------------------------------------------------------
// glibc >= 2.27's expf (Arm Optimized Routines math/expf.c + exp2f_data.c, Copyright (c) 2017-2018 Arm Limited, MIT; glibc
// sysdeps/ieee754/flt-32/e_expf.c): x*N/ln2 split into an integer k and a remainder r in double, exp(x) = 2^(k/N) * p(r) with a
// 32-entry table of 2^(i/N) and a cubic, one rounding to float at the end.  The particle filter's gauss_likelihood calls
// std::exp on a float (/root/reference/src/particle_filter.cpp:53-57).  FMA flavour (CRX_TRIG_FMA=1): glibc's FMA build
// fuses N/ln2*x into the remainder — r = fma(N/ln2, x, -k) — and the polynomial; bit-identical to glibc 2.35's expf on ALL 2^32
// inputs (tests/tools/expf_exhaustive.cpp).  The SSE2 flavour (two roundings) differs from it on 2 inputs.
struct ExpfConsts {
  static constexpr int N = 32;
  static constexpr double inv_ln2_n = 0x1.71547652b82fep+0 * N, shift = 0x1.8p+52;
  static constexpr double c0 = 0x1.c6af84b912394p-5 / N / N / N, c1 = 0x1.ebfce50fac4f3p-3 / N / N, c2 = 0x1.62e42ff0c52d6p-1 / N;
};
CRX_HD uint64_t expf_tab(unsigned i) {     // bits of 2^(i/32) minus i << 47, i = 0..31 (a kernel may stage the 32 words in LDS)
  constexpr uint64_t T[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
  return T[i];
}
// tab: the 32 words of expf_tab in memory of the caller's choice (nullptr = the constant table)
CRX_HD float expf_(float x, const uint64_t* tab = nullptr) {
  const uint32_t at = abstop12(x);
  if (at >= 0x42bu) {                                        // |x| >= 88 or NaN
    if (f32_bits(x) == 0xff800000u) return 0.0f;             // exp(-inf)
    if (at >= 0x7f8u) return x + x;                          // +inf, NaN
    if (x > 0x1.62e42ep6f) return __builtin_inff();          // overflow
    if (x < -0x1.9fe368p6f) return 0.0f;                     // underflow (the results glibc returns through its error path)
  }
  const double xd = (double)x;
  const double z = ExpfConsts::inv_ln2_n * xd;
  double kd = z + ExpfConsts::shift;
  union { double d; uint64_t u; } kb; kb.d = kd;
  const uint64_t ki = kb.u;
  kd -= ExpfConsts::shift;
#if CRX_TRIG_FMA
  const double r = __builtin_fma(ExpfConsts::inv_ln2_n, xd, -kd);
#else
  const double r = z - kd;
#endif
  // 2^(k/N): table word + (k << 47).  k << 47 only touches the high 32-bit word, and only k's low 17 bits reach it: 32-bit arithmetic
  const uint32_t klo = (uint32_t)ki;
  const uint64_t tw = tab ? tab[klo & 31u] : expf_tab(klo & 31u);
  union { uint64_t u; double d; } sb; sb.u = ((uint64_t)((uint32_t)(tw >> 32) + (klo << 15)) << 32) | (uint32_t)tw;
  const double zz = mad_(ExpfConsts::c0, r, ExpfConsts::c1);
  const double r2 = r * r;
  double y = mad_(ExpfConsts::c2, r, 1.0);
  y = mad_(zz, r2, y);
  y = y * sb.d;
  return (float)y;
}

}  // namespace crx

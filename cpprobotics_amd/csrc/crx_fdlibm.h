// crx_fdlibm.h — atanf / atan2f / tanf for the crx engine (host + gfx950 device), bit-identical to
// glibc 2.35's libm on x86-64.
//
// The reference's tracking code calls std::atan2(float,float) (calc_nearest_index,
// /root/reference/src/lqr_speed_steer_control.cpp:80), std::tan(float) (update, :160;
// src/model_predictive_control.cpp:76) — glibc's atan2f and tanf.  In glibc 2.35 these are the
// single-precision fdlibm routines (Sun Microsystems' freely distributable libm: s_atanf.c,
// e_atan2f.c, s_tanf.c, k_tanf.c, e_rem_pio2f.c), built WITHOUT fma contraction and with no ifunc
// variants on x86-64 (unlike sinf/cosf, see crx_trig.h).  This header restates that published
// algorithm; every operation is a plain fp32 operation in the order fdlibm writes it
// (the engine is compiled with -ffp-contract=off).  OCML's device atan2f/tanf round differently.
//
// atanf_ is written branch-free and atan2f_ with a straight-line common case (same operations, same results: a wavefront
// whose lanes fall into different argument ranges would otherwise run every range's branch in turn).
// Verified against the host libm: atanf and tanf on all 2^32 inputs, atan2f on a 3 x 2^32-point
// structured sweep (tests/tools/fdlibm_exhaustive.cpp; tests/test_fdlibm.py runs a strided subset).
// (The constants were cross-checked against the .rodata of the libm.so.6 in this image.)
//
// Notice carried over from the fdlibm sources this restates (s_atanf.c, e_atan2f.c, k_tanf.c, s_tanf.c, e_rem_pio2f.c):
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
// (float conversions of those files by Ian Lance Taylor, Cygnus Support; glibc's tanf argument reduction is glibc's own,
// LGPL-2.1-or-later — restated here from its published description, not copied.)
#pragma once
#include <stdint.h>
#include "crx_trig.h"

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define CRX_FD __host__ __device__ __forceinline__
#else
#define CRX_FD static inline
#endif

namespace crx {

CRX_FD uint32_t fd_bits(float x) { union { float f; uint32_t u; } v; v.f = x; return v.u; }
CRX_FD float fd_float(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }
CRX_FD float fd_fabsf(float x) { return fd_float(fd_bits(x) & 0x7fffffffu); }

// ---- atanf (s_atanf.c) --------------------------------------------------------------------------
// Branch-free: fdlibm's four argument reductions differ only in the numerator and denominator of ONE division
// (|x| < 7/16: x itself, written as x / 1, which is exact), so both are picked by range and a single division is issued;
// the special cases (|x| >= 2^25, NaN, |x| < 2^-29) are selects on the result.  A wavefront whose lanes fall into
// different ranges — headings of a candidate bundle do — otherwise runs every range's division in turn.
CRX_FD float atanf_(float x) {
  const float aT0 = fd_float(0x3eaaaaabu), aT1 = fd_float(0xbe4ccccdu), aT2 = fd_float(0x3e124925u),
              aT3 = fd_float(0xbde38e38u), aT4 = fd_float(0x3dba2e6eu), aT5 = fd_float(0xbd9d8795u),
              aT6 = fd_float(0x3d886b35u), aT7 = fd_float(0xbd6ef16bu), aT8 = fd_float(0x3d4bda59u),
              aT9 = fd_float(0xbd15a221u), aT10 = fd_float(0x3c8569d7u);
  const int32_t hx = (int32_t)fd_bits(x);
  const int32_t ix = hx & 0x7fffffff;
  const float ax = fd_fabsf(x);
  const bool r0 = ix < 0x3ee00000;                 // |x| < 0.4375: no reduction (id = -1)
  const bool r1 = ix < 0x3f300000;                 // < 11/16: id 0
  const bool r2 = ix < 0x3f980000;                 // < 19/16: id 1
  const bool r3 = ix < 0x401c0000;                 // < 2.4375: id 2, else id 3
  //            id -1      id 0             id 1         id 2               id 3
  const float num = r0 ? x : (r1 ? 2.0f * ax - 1.0f : (r2 ? ax - 1.0f : (r3 ? ax - 1.5f : -1.0f)));
  const float den = r0 ? 1.0f : (r1 ? 2.0f + ax : (r2 ? ax + 1.0f : (r3 ? 1.0f + 1.5f * ax : ax)));
  const float hi = r1 ? fd_float(0x3eed6338u) : (r2 ? fd_float(0x3f490fdau) : (r3 ? fd_float(0x3f7b985eu) : fd_float(0x3fc90fdau)));
  const float lo = r1 ? fd_float(0x31ac3769u) : (r2 ? fd_float(0x33222168u) : (r3 ? fd_float(0x33140fb4u) : fd_float(0x33a22168u)));
  const float t = num / den;
  const float z = t * t;
  const float w = z * z;
  const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  const float small = t - t * (s1 + s2);                                  // id < 0
  const float r = hi - ((t * (s1 + s2) - lo) - t);
  float res = r0 ? small : ((hx < 0) ? -r : r);
  res = (ix < 0x31000000) ? x : res;                                      // |x| < 2^-29
  const float big = fd_float(0x3fc90fdau) + fd_float(0x33a22168u);        // atanhi[3] + atanlo[3]
  res = (ix >= 0x4c000000) ? ((hx > 0) ? big : -fd_float(0x3fc90fdau) - fd_float(0x33a22168u)) : res;   // |x| >= 2^25
  res = (ix > 0x7f800000) ? x + x : res;                                  // NaN
  return res;
}

// ---- atan2f (e_atan2f.c) ------------------------------------------------------------------------
CRX_FD float atan2f_general_(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = fd_float(0x3f490fdbu), pi_o_2 = fd_float(0x3fc90fdbu),
              pi = fd_float(0x40490fdbu), pi_lo = fd_float(0xb3bbbd2eu);
  const int32_t hx = (int32_t)fd_bits(x), hy = (int32_t)fd_bits(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;     // NaN
  if (hx == 0x3f800000) return atanf_(y);                   // x = 1.0
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);        // 2*sign(x) + sign(y)
  if (iy == 0) {                                            // y = 0
    switch (m) {
      case 0: case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;   // x = 0
  if (ix == 0x7f800000) {                                   // x = INF
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    } else {
      switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
      }
    }
  }
  if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;   // y = INF
  const int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                    // |y/x| > 2^60
  else if (hx < 0 && k < -60) z = 0.0f;                     // |y|/x < -2^60
  else z = atanf_(fd_fabsf(y / x));
  switch (m) {
    case 0: return z;
    case 1: return fd_float(fd_bits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

// Common case first — both arguments finite and non-zero, x != 1 — as straight-line code (one test, the ratio, atanf_,
// selects for the |y/x| extremes and the quadrant); everything else goes through the case analysis above.
CRX_FD float atan2f_(float y, float x) {
  const float pi_o_2 = fd_float(0x3fc90fdbu), pi = fd_float(0x40490fdbu), pi_lo = fd_float(0xb3bbbd2eu);
  const int32_t hx = (int32_t)fd_bits(x), hy = (int32_t)fd_bits(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  // ix, iy in [1, 0x7f7fffff] (finite, non-zero) and x != 1.0
  const bool plain = ((uint32_t)(ix - 1) < 0x7f7fffffu) & ((uint32_t)(iy - 1) < 0x7f7fffffu) & (hx != 0x3f800000);
  if (!plain) return atan2f_general_(y, x);
  const int32_t k = (iy - ix) >> 23;
  float z = atanf_(fd_fabsf(y / x));
  z = (hx < 0 && k < -60) ? 0.0f : z;                       // |y|/x < -2^60
  z = (k > 60) ? pi_o_2 + 0.5f * pi_lo : z;                 // |y/x| > 2^60
  const float zl = z - pi_lo;
  const float pos = (hx < 0) ? pi - zl : z;                 // m = 2 : m = 0
  const float neg = (hx < 0) ? zl - pi : fd_float(fd_bits(z) ^ 0x80000000u);   // m = 3 : m = 1
  return (hy < 0) ? neg : pos;
}

// ---- acosf (e_acosf.c) --------------------------------------------------------------------------
// std::acos(float) in the reference's DWA goal cost (/root/reference/src/dynamic_window_approach.cpp:107).
// Branch-free like atanf_: the three ranges evaluate the same rational p(z)/q(z), at z = x*x or z = (1 -+ x)/2, and differ
// in how it is combined; z is selected, p/q and the root are computed once, the combination is selected.
CRX_FD float acosf_(float x) {
  const float pi = fd_float(0x40490fdau), pio2_hi = fd_float(0x3fc90fdau), pio2_lo = fd_float(0x33a22168u);
  const float pS0 = fd_float(0x3e2aaaabu), pS1 = fd_float(0xbea6b090u), pS2 = fd_float(0x3e4e0aa8u),
              pS3 = fd_float(0xbd241146u), pS4 = fd_float(0x3a4f7f04u), pS5 = fd_float(0x3811ef08u),
              qS1 = fd_float(0xc019d139u), qS2 = fd_float(0x4001572du), qS3 = fd_float(0xbf303361u),
              qS4 = fd_float(0x3d9dc62eu);
  const int32_t hx = (int32_t)fd_bits(x);
  const int32_t ix = hx & 0x7fffffff;
  const bool inner = ix < 0x3f000000;                                   // |x| < 0.5
  const float z = inner ? x * x : ((hx < 0) ? (1.0f + x) * 0.5f : (1.0f - x) * 0.5f);
  const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const float r = p / q;
  const float s = __builtin_sqrtf(z);
  const float r_in = pio2_hi - (x - (pio2_lo - x * r));                 // |x| < 0.5
  const float r_neg = pi - 2.0f * (s + (r * s - pio2_lo));              // x < -0.5
  const float df = fd_float(fd_bits(s) & 0xfffff000u);                  // x > 0.5
  const float c = (z - df * df) / (s + df);
  const float r_pos = 2.0f * (df + (r * s + c));
  float res = inner ? r_in : ((hx < 0) ? r_neg : r_pos);
  res = (ix <= 0x23000000) ? pio2_hi + pio2_lo : res;                   // |x| < 2^-57
  res = (ix == 0x3f800000) ? (hx > 0 ? 0.0f : pi + 2.0f * pio2_lo) : res;   // |x| == 1
  res = (ix > 0x3f800000) ? (x - x) / (x - x) : res;                    // |x| > 1 or NaN -> NaN
  return res;
}

// ---- tanf (s_tanf.c, k_tanf.c, the |x| < 2^7*pi/2 part of e_rem_pio2f.c) -------------------------
CRX_FD float kernel_tanf_(float x, float y, int iy) {
  const float pio4 = fd_float(0x3f490fdau), pio4lo = fd_float(0x33222168u);
  const float T0 = fd_float(0x3eaaaaabu), T1 = fd_float(0x3e088889u), T2 = fd_float(0x3d5d0dd1u),
              T3 = fd_float(0x3cb327a4u), T4 = fd_float(0x3c11371fu), T5 = fd_float(0x3b6b6916u),
              T6 = fd_float(0x3abede48u), T7 = fd_float(0x3a1a26c8u), T8 = fd_float(0x398137b9u),
              T9 = fd_float(0x38a3f445u), T10 = fd_float(0x3895c07au), T11 = fd_float(0xb79bae5fu),
              T12 = fd_float(0x37d95384u);
  const int32_t hx = (int32_t)fd_bits(x);
  const int32_t ix = hx & 0x7fffffff;
  if (ix < 0x39000000) {              // |x| < 2^-13
    if ((int)x == 0) {
      if ((ix | (iy + 1)) == 0) return 1.0f / fd_fabsf(x);
      else if (iy == 1) return x;
      else return -1.0f / x;
    }
  }
  if (ix >= 0x3f2ca140) {             // |x| >= 0.6744
    if (hx < 0) { x = -x; y = -y; }
    const float z0 = pio4 - x;
    const float w0 = pio4lo - y;
    x = z0 + w0; y = 0.0f;
    if (fd_fabsf(x) < 0x1p-13f) return (float)(1 - ((hx >> 30) & 2)) * iy * (1.0f - 2 * iy * x);
  }
  float z = x * x;
  float w = z * z;
  float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
  float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
  float s = z * x;
  r = y + z * (s * (r + v) + y);
  r += T0 * s;
  w = x + r;
  if (ix >= 0x3f2ca140) {
    v = (float)iy;
    return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
  }
  if (iy == 1) return w;
  // -1/(x+r), accurately
  z = fd_float(fd_bits(w) & 0xfffff000u);
  v = r - (z - x);
  const float a = -1.0f / w;
  const float t = fd_float(fd_bits(a) & 0xfffff000u);
  s = 1.0f + t * z;
  return t + a * (s + t * v);
}

// tanf (s_tanf.c).  glibc >= 2.28 reduces the argument with the double-precision scheme of its new
// sinf/cosf (e_rem_pio2f.c: reduce_fast below |x| = 120, the 192-bit 2/pi table above) and hands
// y0 = (float)r, y1 = (float)(r - y0) to the fdlibm kernel.  e_rem_pio2f.c has no fma variant on x86-64:
// x - n*(pi/2) is a multiply followed by a subtract (two roundings), unlike the fused form in crx_trig.h.
CRX_FD float tanf_(float x) {
  const uint32_t xi = fd_bits(x);
  const int32_t ix = (int32_t)(xi & 0x7fffffffu);
  if (ix <= 0x3f490fda) return kernel_tanf_(x, 0.0f, 1);   // |x| ~<= pi/4
  if (ix >= 0x7f800000) return x - x;                       // Inf or NaN -> NaN
  double dx = (double)x;
  int n;
  if (abstop12(x) < 0x42fu) {                               // |x| < 120
    const double r = dx * SinCosConsts::hpi_inv;
    n = ((int32_t)r + 0x800000) >> 24;
    const double nh = (double)n * SinCosConsts::hpi;
    dx = dx - nh;
  } else {
    dx = reduce_large(xi, &n);
    if (xi >> 31) dx = -dx;
  }
  const float y0 = (float)dx;
  const float y1 = (float)(dx - (double)y0);
  return kernel_tanf_(y0, y1, 1 - ((n & 1) << 1));
}

}  // namespace crx

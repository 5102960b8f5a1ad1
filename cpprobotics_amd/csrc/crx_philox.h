// crx_philox.h — counter-based standard-normal draws for the synthetic inputs (host + gfx950 device, bit-identical).
//
// The reference draws its simulation noise from a random_device-seeded std::mt19937 through std::normal_distribution
// (/root/reference/src/extended_kalman_filter.cpp:162-164, :174-181) — irreproducible by construction, so the engine's
// benchmark and swarm drivers generate their own draws.  They are keyed by (seed, stream, GLOBAL agent id, step) with
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11): one call yields the
// four N(0,1) draws one pass of the reference's loop consumes (two for ud, two for z), and the value an agent sees does
// not depend on how the swarm is sharded over GPUs.
//
// 32 random bits -> uniform -> Box-Muller, written with IEEE basic operations only (the library and the oracle are both
// built -ffp-contract=off; fp64 division and sqrt are correctly rounded on gfx950 and x86-64): log in double by the
// fdlibm e_log.c scheme (Sun Microsystems' freely distributable libm; notice in crx_fdlibm.h), sin/cos by crx_trig.h.
// Host and device therefore produce the same bytes, which tests/test_philox.py checks.
#pragma once
#include <stdint.h>
#include "crx_trig.h"

namespace crx {

CRX_HD void philox_mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
  const uint64_t p = (uint64_t)a * (uint64_t)b;
  *hi = (uint32_t)(p >> 32);
  *lo = (uint32_t)p;
}

// Philox4x32-10: ctr[4] in/out, key (k0, k1)
CRX_HD void philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    philox_mulhilo(0xD2511F53u, ctr[0], &hi0, &lo0);
    philox_mulhilo(0xCD9E8D57u, ctr[2], &hi1, &lo1);
    const uint32_t c0 = hi1 ^ ctr[1] ^ k0, c1 = lo1, c2 = hi0 ^ ctr[3] ^ k1, c3 = lo0;
    ctr[0] = c0; ctr[1] = c1; ctr[2] = c2; ctr[3] = c3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// log(x) for a normal, positive double: fdlibm's e_log.c (argument reduction to [sqrt(1/2), sqrt(2)), s = f/(2+f),
// even polynomial in s), every operation written out
CRX_HD double philox_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  union { double d; uint64_t u; } v;
  v.d = x;
  int k = (int)((v.u >> 52) & 0x7ff) - 1023;
  uint64_t m = v.u & 0x000fffffffffffffull;
  if (m >= 0x6a09e667f3bcdull) { k += 1; v.u = m | 0x3fe0000000000000ull; }     // mantissa >= sqrt(2): use m/2
  else v.u = m | 0x3ff0000000000000ull;
  const double f = v.d - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// two N(0,1) floats from two 32-bit words (Box-Muller on 24-bit uniforms: u1 in (0,1], u2 in [0,1))
CRX_HD void philox_box_muller(uint32_t a, uint32_t b, float* z0, float* z1) {
  const double u1 = (double)((a >> 8) + 1u) * 0x1p-24;
  const float theta = (float)(b >> 8) * 0x1p-24f * 6.2831855f;
  const double r2 = -2.0 * philox_log(u1);
  const float r = (float)__builtin_sqrt(r2);
  float sn, cs;
#if defined(__HIP_DEVICE_COMPILE__)
  // theta is +0 or in [2^-24 * 2 pi, 2 pi): inside the domain on which the short form of crx_trig.h gives sinf's / cosf's bits (every
  // float walked, tests/tools/trig_fast_exhaustive.cpp); +0 -> (+0, 1) as well.  Half the instructions of the general form.
  sincosf_wave_fast_(theta, &sn, &cs);
#else
  sincosf_(theta, &sn, &cs);
#endif
  *z0 = r * cs;
  *z1 = r * sn;
}

// the four standard-normal draws of (seed, stream, agent, step)
CRX_HD void philox_normal4(uint64_t seed, uint32_t stream, uint64_t agent, uint32_t step, float out[4]) {
  uint32_t c[4] = {(uint32_t)agent, (uint32_t)(agent >> 32), step, stream};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  philox_box_muller(c[0], c[1], &out[0], &out[1]);
  philox_box_muller(c[2], c[3], &out[2], &out[3]);
}

}  // namespace crx

// ekf_math.h — per-vehicle EKF arithmetic of the crx engine (gfx950 device code; also compiles as
// plain host C++ so tests/tools/ekf_packed_host.cpp can check the packed restatement on the CPU).
//
// Replaces the reference's motion_model / jacobF / observation_model / jacobH / ekf_estimation
// (/root/reference/src/extended_kalman_filter.cpp:22-78) for ONE vehicle held in registers.
// Arithmetic contract: see ekf_kernels.hip.h.
#pragma once
#include "crx_trig.h"

namespace crx {

#if defined(__clang__)
typedef float v2f __attribute__((ext_vector_type(2)));
#else
typedef float v2f __attribute__((vector_size(8)));
#endif

struct EkfConsts {
  float Q[16];  // column-major 4x4
  float R[4];   // column-major 2x2
  double dt;
};

struct EkfState {
  float x0, x1, x2, x3;
  float P[16];  // column-major: P[i + 4*j]
};

// motion_model(): x <- F_*x + B_*u   (:22-36)
CRX_HD void motion_model_dev(float& x0, float& x1, float& x2, float& x3,
                                                 float u0, float u1, double dt) {
  float s, c;
  sincosf_(x2, &s, &c);
  const float b0 = (float)(dt * (double)c);  // B_(0,0) = DT*cos(yaw)
  const float b1 = (float)(dt * (double)s);  // B_(1,0) = DT*sin(yaw)
  const float b2 = (float)dt;                // B_(2,1) = DT
  x0 = x0 + b0 * u0;
  x1 = x1 + b1 * u0;
  x2 = x2 + b2 * u1;
  x3 = x3 + u0;  // F_(3,3)=1.0 and B_(3,0)=1.0: the reference's velocity state integrates u0
}

// The four non-trivial entries of jacobF(x,u) (:38-47); the rest of jF is the identity.
struct JacF { float j02, j03, j12, j13; };
CRX_HD JacF jacobF_dev(float yaw, float v, double dt) {
  float s, c;
  sincosf_(yaw, &s, &c);
  JacF j;
  j.j02 = (float)((-dt * (double)v) * (double)s);
  j.j03 = (float)(dt * (double)c);
  j.j12 = (float)((dt * (double)v) * (double)c);
  j.j13 = (float)(dt * (double)s);
  return j;
}

// One ekf_estimation() (:64-78) on register-resident state.
CRX_HD void ekf_step_dev(EkfState& s, float z0, float z1, float u0, float u1,
                                             const EkfConsts& k) {
  // xPred = motion_model(xEst, u)                                           :67
  float xp0 = s.x0, xp1 = s.x1, xp2 = s.x2, xp3 = s.x3;
  motion_model_dev(xp0, xp1, xp2, xp3, u0, u1, k.dt);
  // jF = jacobF(xPred, u)                                                   :68
  const JacF jf = jacobF_dev(xp2, u0, k.dt);
  const float* P = s.P;
  // T1 = jF*PEst ; rows 2,3 of jF are unit rows                              :69
  float T1[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    T1[0 + 4 * j] = (P[0 + 4 * j] + jf.j02 * P[2 + 4 * j]) + jf.j03 * P[3 + 4 * j];
    T1[1 + 4 * j] = (P[1 + 4 * j] + jf.j12 * P[2 + 4 * j]) + jf.j13 * P[3 + 4 * j];
    T1[2 + 4 * j] = P[2 + 4 * j];
    T1[3 + 4 * j] = P[3 + 4 * j];
  }
  // PPred = T1*jF^T + Q
  float PP[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    PP[i + 0] = ((T1[i + 0] + T1[i + 8] * jf.j02) + T1[i + 12] * jf.j03) + k.Q[i + 0];
    PP[i + 4] = ((T1[i + 4] + T1[i + 8] * jf.j12) + T1[i + 12] * jf.j13) + k.Q[i + 4];
    PP[i + 8] = T1[i + 8] + k.Q[i + 8];
    PP[i + 12] = T1[i + 12] + k.Q[i + 12];
  }
  // y = z - H*xPred ; S = H*PPred*H^T + R ; Sinv closed form                 :72-75
  const float y0 = z0 - xp0;
  const float y1 = z1 - xp1;
  const float S00 = PP[0] + k.R[0], S10 = PP[1] + k.R[1];
  const float S01 = PP[4] + k.R[2], S11 = PP[5] + k.R[3];
  const float det = S00 * S11 - S10 * S01;
  const float invdet = 1.0f / det;
  const float Si00 = S11 * invdet, Si10 = -S10 * invdet;
  const float Si01 = -S01 * invdet, Si11 = S00 * invdet;
  // K = (PPred*H^T)*Sinv                                                     :75
  float K0[4], K1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    K0[i] = PP[i] * Si00 + PP[i + 4] * Si10;
    K1[i] = PP[i] * Si01 + PP[i + 4] * Si11;
  }
  // xEst = xPred + K*y                                                       :76
  s.x0 = xp0 + (K0[0] * y0 + K1[0] * y1);
  s.x1 = xp1 + (K0[1] * y0 + K1[1] * y1);
  s.x2 = xp2 + (K0[2] * y0 + K1[2] * y1);
  s.x3 = xp3 + (K0[3] * y0 + K1[3] * y1);
  // PEst = (I - K*H)*PPred                                                   :77
  const float M00 = 1.0f - K0[0], M01 = 0.0f - K1[0];
  const float M10 = 0.0f - K0[1], M11 = 1.0f - K1[1];
  const float M20 = 0.0f - K0[2], M21 = 0.0f - K1[2];
  const float M30 = 0.0f - K0[3], M31 = 0.0f - K1[3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float p0 = PP[0 + 4 * j], p1 = PP[1 + 4 * j], p2 = PP[2 + 4 * j], p3 = PP[3 + 4 * j];
    s.P[0 + 4 * j] = M00 * p0 + M01 * p1;
    s.P[1 + 4 * j] = M10 * p0 + M11 * p1;
    s.P[2 + 4 * j] = (M20 * p0 + M21 * p1) + p2;
    s.P[3 + 4 * j] = (M30 * p0 + M31 * p1) + p3;
  }
}


//
// Fast path ("packed" step): the same arithmetic as ekf_step_dev, written on 2-wide vectors so that
// every fp32 multiply/add is a v_pk_mul_f32 / v_pk_add_f32 (two matrix entries per instruction, the
// rows (0,1) and (2,3) of a column), with the two sincos of the step evaluated on their |yaw| < 120
// path and the 2x2 determinant inverted by the un-scaled Newton sequence.  Both shortcuts are
// bit-identical to the general code on their domain; each lane records whether it stayed inside the
// domain (FastDomain), and a wave in which any lane left it re-runs the whole chunk of D steps through
// ekf_step_dev.  The hot loop is therefore ONE basic block per D steps (no per-lane branches), which
// is what lets the scheduler overlap the fp64 trig chains with the fp32 matrix work.
//
// The two shortcuts and their domains are described at sincos_fast2 / recip_fast below.

struct EkfStateP {
  v2f x01, x23;      // (x, y), (yaw, v)
  v2f Plo[4], Phi[4];  // column j of P: rows (0,1) and rows (2,3)
};

struct EkfConstsP {
  v2f Qlo[4], Qhi[4];
  v2f Rc0, Rc1;
  double dt;
  float dtf;
  float dt_hi, dt_lo;   // dt = dt_hi + dt_lo + (less than 2^-53 dt): the split of dt_mul_split
};

// (float)(dt * (double)t) — B_(0,0), B_(1,0) of motion_model (:30-31) and jF(0,3), jF(1,3) of jacobF (:43,45): a float promoted to
// double, multiplied by the double literal DT and rounded back — WITHOUT leaving fp32: fma(t, dt_hi, t * dt_lo).  On gfx950 the double
// form is two conversions and a multiply; the split form is one v_pk_mul_f32 + one v_pk_fma_f32 per PAIR (6 conversions and 4 fp64
// multiplies fewer per step, 197 -> 190 VALU instructions; measured on one box, alternating builds: 131.7 -> 133.4 G updates/s, +1.3 % —
// less than the instruction count suggests: the fp64 side of the step overlaps the packed fp32 matrix work).  It is the reference's value, bit for bit, for dt = 0.1 (the reference's `#define
// DT 0.1`) and every finite float with |t| >= 2^-120: proved by walking all 2^32 floats (tests/tools/dt_split_exhaustive.cpp;
// below 2^-120 the low product underflows — 10 M mismatches, all there).  The step feeds it sines and cosines of angles in the fast
// domain, 2^-100 <= |yaw| < 120, whose magnitudes are >= 2^-100 (same tool).  For any other dt the kernels are instantiated with the
// double form (dt_split_is_exact; the host decides per launch).
CRX_HD bool dt_split_is_exact(double dt) { return dt == 0.1; }
CRX_HD v2f pk_fma(v2f a, v2f b, v2f c) {
#if defined(__clang__)
  return __builtin_elementwise_fma(a, b, c);
#else
  return v2f{__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1])};
#endif
}
CRX_HD v2f dt_mul_split(v2f t, float dt_hi, float dt_lo) { return pk_fma(t, v2f{dt_hi, dt_hi}, t * v2f{dt_lo, dt_lo}); }

CRX_HD EkfConstsP pack_consts(const EkfConsts& k) {
  EkfConstsP c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c.Qlo[j] = v2f{k.Q[4 * j + 0], k.Q[4 * j + 1]};
    c.Qhi[j] = v2f{k.Q[4 * j + 2], k.Q[4 * j + 3]};
  }
  c.Rc0 = v2f{k.R[0], k.R[1]};
  c.Rc1 = v2f{k.R[2], k.R[3]};
  c.dt = k.dt;
  c.dtf = (float)k.dt;
  c.dt_hi = (float)k.dt;
  c.dt_lo = (float)(k.dt - (double)c.dt_hi);
  return c;
}

CRX_HD void pack_state(EkfStateP& p, const EkfState& s) {
  p.x01 = v2f{s.x0, s.x1};
  p.x23 = v2f{s.x2, s.x3};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p.Plo[j] = v2f{s.P[4 * j + 0], s.P[4 * j + 1]};
    p.Phi[j] = v2f{s.P[4 * j + 2], s.P[4 * j + 3]};
  }
}

CRX_HD void unpack_state(EkfState& s, const EkfStateP& p) {
  s.x0 = p.x01[0]; s.x1 = p.x01[1]; s.x2 = p.x23[0]; s.x3 = p.x23[1];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s.P[4 * j + 0] = p.Plo[j][0]; s.P[4 * j + 1] = p.Plo[j][1];
    s.P[4 * j + 2] = p.Phi[j][0]; s.P[4 * j + 3] = p.Phi[j][1];
  }
}

// ---- fast-domain bookkeeping ------------------------------------------------------------------
// The packed step below takes two shortcuts that are bit-identical to the general code only on a
// domain: 2^-100 <= |yaw| < 120 for both angles of the step (sincos; the lower end keeps their sines above 2^-120, where the fp32
// form of DT * sin is exact — dt_mul_split) and 2^-60 <= |det S| <= 2^60 (reciprocal).
// Instead of a per-step boolean (v_cmp + mask logic), each lane keeps running max/min of the
// quantities involved — one VALU instruction each — and the caller tests them once per chunk.
// NaNs pass through max/min unnoticed; that is harmless: a NaN angle or determinant turns the state
// into NaN on the fast path exactly as it does on the general one.
struct FastDomain {
  float amax, amin;   // max / min |yaw| seen
  float dmax, dmin;   // max / min |det S| seen
};
CRX_HD FastDomain fast_domain_init() { return FastDomain{0.0f, 1.0f, 1.0f, 1.0f}; }
CRX_HD bool fast_domain_ok(const FastDomain& f) {
  return (f.amax < 120.0f) & (f.amin >= 0x1p-100f) & (f.dmax <= 0x1p60f) & (f.dmin >= 0x1p-60f);
}

// Bit helpers of the quadrant logic.  Deliberately NOT inline asm: the compiler's hazard recogniser does
// not look inside asm blocks, and an asm VALU write right behind a 16-byte store of the same register
// corrupted the stored data on gfx950 (seen in the P-history test) — builtins and plain C only.
// 0xffffffff if bit 24 of v is set, else 0 (v_bfe_i32).
CRX_HD uint32_t bit24_mask(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_sbfe((int)v, 24u, 1u);
#else
  return (uint32_t)((int32_t)(v << 7) >> 31);
#endif
}
// (m & a) | (~m & b): ONE v_bitop3_b32 (truth table 0xCA, src0 selects).  Written in C the compiler sees that m is a sign-extended bit
// and emits either v_cmp + two VOP2 v_cndmask reading VCC (19 cycles each back to back, profiles/r01/ubench_issue_patterns.txt row U)
// or, for bit 0, seven v_and/v_or per angle.
CRX_HD uint32_t bitselect(uint32_t m, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA);
#else
  return (m & a) | (~m & b);
#endif
}
// a ^ (b & k)  (v_bitop3_b32, truth table 0x78)
CRX_HD uint32_t xor_masked(uint32_t a, uint32_t b, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, k, 0x78);
#else
  return a ^ (b & k);
#endif
}

CRX_HD uint32_t f2u(float x) { union { float f; uint32_t u; } v; v.f = x; return v.u; }
CRX_HD float u2f(uint32_t x) { union { float f; uint32_t u; } v; v.u = x; return v.f; }

// sincosf_ on its 2^-100 <= |y| < 120 domain, for the two angles of one EKF step at once (the two dependency chains are written
// interleaved so that each fp64 instruction has an independent neighbour).  Outside that domain the outputs are meaningless; `dom`
// records where the angles were.  NOT sincosf_'s operations — 13 fp64 instructions and 3 conversions per angle where its own order
// (rounds 2-4) takes 15 and 5 — but its RESULTS: tests/tools/trig_fast_exhaustive.cpp walks every float of the domain (1,793,064,960
// inputs) and finds the same sine and cosine bits as sincosf_ (itself glibc's sinf / cosf on all 2^32 inputs) on every one.
//   * reduction: n = rint(y * 2/pi) as ONE fma against 1.5 * 2^52 (the integer lands in the low mantissa bits, where the quadrant
//     logic reads it) and a subtraction, instead of glibc's ((int)(y * 2^24 * 2/pi) + 2^23) >> 24 — a conversion to int, an add, a
//     shift and a conversion back.  The two disagree on 5 floats of the domain (y * 2/pi within 2^-24 of a half-integer); there
//     the other quadrant's polynomial rounds to the same floats.
//   * polynomials: Horner in x^2 (sine: 3 fma on x^3; cosine: 4 fma) instead of glibc's split forms (5 and 6 operations).  The doubles
//     differ in the last bit on ~8 % of the inputs, never across a float rounding boundary.
//   * |y| < 2^-12, where sinf_/cosf_ return y and 1: n = 0 and x = y exactly, and the polynomials give (float)(y - y^3/6..) = y and
//     (float)(1 - y^2/2..) = 1 by themselves.  The one input they miss is y = -0.0f (the polynomial yields +0): |y| = 0 is outside
//     the fast domain.
//   * quadrant logic without compares: bit 0 of n says "swap sine and cosine", bit 1 is the sign of the sine output and bit 1 of n + 1
//     the sign of the cosine output; the swap is a bitfield select, the signs are xors of the sign bit.
CRX_HD uint32_t lo32_of(double t) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__double2loint(t);
#else
  uint64_t u; __builtin_memcpy(&u, &t, 8); return (uint32_t)u;
#endif
}
// 0xffffffff if bit 0 of v is set, else 0 (v_bfe_i32)
CRX_HD uint32_t bit0_mask(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_sbfe((int)v, 0u, 1u);
#else
  return (uint32_t)((int32_t)(v << 31) >> 31);
#endif
}
CRX_HD void sincos_fast2(const float y[2], float so[2], float co[2], FastDomain& dom) {
  typedef SinCosConsts C;
  constexpr double two_over_pi = C::hpi_inv * 0x1p-24, magic = 0x1.8p52;
  double x[2], t[2], nd[2], x2[2], x3[2], ps[2], pc[2], S[2], Cv[2];
#define CRX_BOTH for (int i = 0; i < 2; ++i)
  dom.amax = __builtin_fmaxf(__builtin_fmaxf(dom.amax, __builtin_fabsf(y[0])), __builtin_fabsf(y[1]));
  dom.amin = __builtin_fminf(__builtin_fminf(dom.amin, __builtin_fabsf(y[0])), __builtin_fabsf(y[1]));
  _Pragma("unroll") CRX_BOTH x[i] = (double)y[i];
  _Pragma("unroll") CRX_BOTH t[i] = __builtin_fma(x[i], two_over_pi, magic);
  _Pragma("unroll") CRX_BOTH nd[i] = t[i] - magic;
  _Pragma("unroll") CRX_BOTH x[i] = __builtin_fma(-nd[i], C::hpi, x[i]);
  _Pragma("unroll") CRX_BOTH x2[i] = x[i] * x[i];
  _Pragma("unroll") CRX_BOTH ps[i] = __builtin_fma(x2[i], C::s3, C::s2);
  _Pragma("unroll") CRX_BOTH pc[i] = __builtin_fma(x2[i], C::c4, C::c3);
  _Pragma("unroll") CRX_BOTH x3[i] = x[i] * x2[i];
  _Pragma("unroll") CRX_BOTH ps[i] = __builtin_fma(x2[i], ps[i], C::s1);
  _Pragma("unroll") CRX_BOTH pc[i] = __builtin_fma(x2[i], pc[i], C::c2);
  _Pragma("unroll") CRX_BOTH S[i] = __builtin_fma(x3[i], ps[i], x[i]);
  _Pragma("unroll") CRX_BOTH pc[i] = __builtin_fma(x2[i], pc[i], C::c1);
  _Pragma("unroll") CRX_BOTH Cv[i] = __builtin_fma(x2[i], pc[i], C::c0);
  _Pragma("unroll") CRX_BOTH {
    const uint32_t n = lo32_of(t[i]);                                 // n mod 2^32
    const uint32_t fs = f2u((float)S[i]), fc = f2u((float)Cv[i]);     // rounding commutes with the sign flips
    const uint32_t odd = bit0_mask(n);                                // all ones when n is odd
    const uint32_t sr = bitselect(odd, fc, fs);
    const uint32_t cr = bitselect(odd, fs, fc);
    const uint32_t qs = n << 30;                  // bit 31 = bit 1 of n
    const uint32_t qc = qs + 0x40000000u;         // bit 31 = bit 1 of n + 1
    so[i] = u2f(xor_masked(sr, qs, 0x80000000u));
    co[i] = u2f(xor_masked(cr, qc, 0x80000000u));
  }
#undef CRX_BOTH
}

// 1.0f/d, IEEE-rounded, for 2^-60 <= |d| <= 2^60 (`dom` records |d|): v_rcp_f32 and ONE Newton step.
//   LLVM's IEEE fp32 division is v_div_scale x2, v_rcp, 6 fma/mul, v_div_fmas, v_div_fixup; in that range, with numerator 1.0, neither
//   v_div_scale scales, v_div_fmas is a plain fma and v_div_fixup passes the quotient through, so rcp + the same six fma give its bits
//   (rounds 2-4 issued exactly that).  The last four of them never change the result: on gfx950's v_rcp_f32,
//   fma(fma(-d, r, 1), r, r) IS the correctly rounded quotient for every one of the 2,013,265,922 floats of the range
//   (crx_x_recip_sweep_dev, run by tests/test_ekf_gpu.py on the device the tests run on; profiles/r05/recip_exhaustive.txt).
CRX_HD float recip_fast(float d, FastDomain& dom) {
  const float ad = __builtin_fabsf(d);
  dom.dmax = __builtin_fmaxf(dom.dmax, ad);
  dom.dmin = __builtin_fminf(dom.dmin, ad);
#if defined(__HIP_DEVICE_COMPILE__)
  const float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
#else
  return 1.0f / d;                                     // host build (tests): the IEEE quotient itself
#endif
}
CRX_HD v2f bc(float a) { return v2f{a, a}; }

// One ekf_estimation() (:64-78), packed.  Same operation order as ekf_step_dev, entry by entry.
// DTS: the four DT * cos / DT * sin entries through dt_mul_split (requires dt_split_is_exact(k.dt)), otherwise in double as written.
// FMA (round 6, opt-in: crx_ekf_params.arith = CRX_ARITH_CONTRACT): the SAME operations in the SAME order, but wherever the reference
// has a multiply followed by the add that consumes it — every term after the first of a product coefficient, Eigen's
// `res = pmadd(lhs, rhs, res)` — the two are ONE v_pk_fma_f32, i.e. what the reference's own expressions become when they are compiled
// with FMA contraction (-march=haswell / -ffp-contract=fast; its CMakeLists sets neither, which is why the default mode keeps them
// apart).  A k-term sum is k instructions instead of 2k - 1: 71 packed matrix operations per step instead of 103.  Untouched: the
// trig, the double-formed Jacobian entries, the determinant (a difference of two products: fusing it saves nothing), the reciprocal,
// the structural adds (+Q, +R, I - K, + PPred rows).  Results differ from the unfused ones in the last bits only — every fused
// operation is one rounding instead of two; tests/test_ekf_gpu.py and bench.py measure the distance to the oracle under the
// contract's floored metric: the mode exists because north_star's tolerance is 1e-6, not 0.
template <bool DTS = false, bool FMA = false>
CRX_HD void ekf_step_packed(EkfStateP& s, v2f z, v2f u, const EkfConstsP& k, FastDomain& dom) {
  const float u0 = u[0], u1 = u[1];
  // motion_model: both yaw angles of the step are known up front
  const float yaw0 = s.x23[0];
  // xPred(2) = x(2) + DT*u(1), xPred(3) = x(3) + u(0): one packed multiply by (DT, 1) — u(0) * 1.0f is u(0) — and one packed add
  const v2f xp23 = FMA ? pk_fma(v2f{u1, u0}, v2f{k.dtf, 1.0f}, s.x23) : s.x23 + v2f{u1, u0} * v2f{k.dtf, 1.0f};
  const float yaw1 = xp23[0];
  const float yaws[2] = {yaw0, yaw1};
  float sn[2], cs[2];
  sincos_fast2(yaws, sn, cs, dom);
  const float s0 = sn[0], c0 = cs[0], s1 = sn[1], c1 = cs[1];
  v2f b01, jB;
  if constexpr (DTS) {
    b01 = dt_mul_split(v2f{c0, s0}, k.dt_hi, k.dt_lo);
    jB = dt_mul_split(v2f{c1, s1}, k.dt_hi, k.dt_lo);
  } else {
    b01 = v2f{(float)(k.dt * (double)c0), (float)(k.dt * (double)s0)};
    jB = v2f{(float)(k.dt * (double)c1), (float)(k.dt * (double)s1)};
  }
  const v2f xp01 = FMA ? pk_fma(b01, bc(u0), s.x01) : s.x01 + b01 * bc(u0);
  // jacobF(xPred, u): yaw = xPred(2), v = u(0)
  const double dv = k.dt * (double)u0;
  const float j02 = (float)((-dv) * (double)s1);
  const float j12 = (float)(dv * (double)c1);
  const float j03 = jB[0], j13 = jB[1];
  const v2f jA = v2f{j02, j12};
  // The matrix part is written stage-major (the same operation across all columns, then the next
  // operation): consecutive instructions are then independent, and the packed-op result hazard
  // (a dependent instruction right behind a v_pk_* needs a wait state) costs no s_nop.
  // T1 = jF*PEst (rows 2,3 of jF are unit rows):  T1lo[j] = (Plo[j] + jA*P2j) + jB*P3j
  v2f T1lo[4], ta[4], tb[4];
  if constexpr (FMA) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ta[j] = pk_fma(jA, bc(s.Phi[j][0]), s.Plo[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) T1lo[j] = pk_fma(jB, bc(s.Phi[j][1]), ta[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) ta[j] = jA * bc(s.Phi[j][0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) tb[j] = jB * bc(s.Phi[j][1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) ta[j] = s.Plo[j] + ta[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) T1lo[j] = ta[j] + tb[j];
  }
  // PPred = T1*jF^T + Q:  col0 = ((T1c0 + T1c2*j02) + T1c3*j03) + Qc0, col1 likewise with j12, j13
  v2f PPlo[4], PPhi[4];
  if constexpr (FMA) {
    const v2f a0 = pk_fma(T1lo[2], bc(j02), T1lo[0]), a1 = pk_fma(s.Phi[2], bc(j02), s.Phi[0]);
    const v2f a2 = pk_fma(T1lo[2], bc(j12), T1lo[1]), a3 = pk_fma(s.Phi[2], bc(j12), s.Phi[1]);
    PPlo[2] = T1lo[2] + k.Qlo[2];
    PPhi[2] = s.Phi[2] + k.Qhi[2];
    PPlo[3] = T1lo[3] + k.Qlo[3];
    PPhi[3] = s.Phi[3] + k.Qhi[3];
    const v2f c0_ = pk_fma(T1lo[3], bc(j03), a0), c1_ = pk_fma(s.Phi[3], bc(j03), a1);
    const v2f c2_ = pk_fma(T1lo[3], bc(j13), a2), c3_ = pk_fma(s.Phi[3], bc(j13), a3);
    PPlo[0] = c0_ + k.Qlo[0];
    PPhi[0] = c1_ + k.Qhi[0];
    PPlo[1] = c2_ + k.Qlo[1];
    PPhi[1] = c3_ + k.Qhi[1];
  } else {
    const v2f m0 = T1lo[2] * bc(j02), m1 = s.Phi[2] * bc(j02), m2 = T1lo[2] * bc(j12), m3 = s.Phi[2] * bc(j12);
    const v2f n0 = T1lo[3] * bc(j03), n1 = s.Phi[3] * bc(j03), n2 = T1lo[3] * bc(j13), n3 = s.Phi[3] * bc(j13);
    const v2f a0 = T1lo[0] + m0, a1 = s.Phi[0] + m1, a2 = T1lo[1] + m2, a3 = s.Phi[1] + m3;
    PPlo[2] = T1lo[2] + k.Qlo[2];
    PPhi[2] = s.Phi[2] + k.Qhi[2];
    PPlo[3] = T1lo[3] + k.Qlo[3];
    PPhi[3] = s.Phi[3] + k.Qhi[3];
    const v2f c0 = a0 + n0, c1 = a1 + n1, c2 = a2 + n2, c3 = a3 + n3;
    PPlo[0] = c0 + k.Qlo[0];
    PPhi[0] = c1 + k.Qhi[0];
    PPlo[1] = c2 + k.Qlo[1];
    PPhi[1] = c3 + k.Qhi[1];
  }
  // y, S, S^-1
  const v2f y = z - xp01;
  const v2f Sc0 = PPlo[0] + k.Rc0;   // (S00, S10)
  const v2f Sc1 = PPlo[1] + k.Rc1;   // (S01, S11)
  const v2f dd = Sc0 * v2f{Sc1[1], Sc1[0]};   // (S00*S11, S10*S01)
  const float det = dd[0] - dd[1];
  const float inv = recip_fast(det, dom);
  // S^-1 = inv * [S11 -S01; -S10 S00]: the four products as two packed multiplies; (-a) * inv = -(a * inv) exactly, so the two
  // negations ride on the consumers' operands
  const v2f W0 = Sc0 * bc(inv);      // (S00*inv, S10*inv) = ( Si11, -Si10)
  const v2f W1 = Sc1 * bc(inv);      // (S01*inv, S11*inv) = (-Si01,  Si00)
  // K = (PPred*H^T)*Sinv
  v2f K0lo, K0hi, K1lo, K1hi;
  {
    const v2f e0 = PPlo[0] * bc(W1[1]), e1 = PPhi[0] * bc(W1[1]), e2 = PPlo[0] * bc(-W1[0]), e3 = PPhi[0] * bc(-W1[0]);
    if constexpr (FMA) {
      K0lo = pk_fma(PPlo[1], bc(-W0[1]), e0); K0hi = pk_fma(PPhi[1], bc(-W0[1]), e1);
      K1lo = pk_fma(PPlo[1], bc(W0[0]), e2); K1hi = pk_fma(PPhi[1], bc(W0[0]), e3);
    } else {
      const v2f f0 = PPlo[1] * bc(-W0[1]), f1 = PPhi[1] * bc(-W0[1]), f2 = PPlo[1] * bc(W0[0]), f3 = PPhi[1] * bc(W0[0]);
      K0lo = e0 + f0; K0hi = e1 + f1; K1lo = e2 + f2; K1hi = e3 + f3;
    }
  }
  // xEst = xPred + K*y
  {
    const v2f g0 = K0lo * bc(y[0]), g1 = K0hi * bc(y[0]);
    v2f d0, d1;
    if constexpr (FMA) { d0 = pk_fma(K1lo, bc(y[1]), g0); d1 = pk_fma(K1hi, bc(y[1]), g1); }
    else { const v2f h0 = K1lo * bc(y[1]), h1 = K1hi * bc(y[1]); d0 = g0 + h0; d1 = g1 + h1; }
    s.x01 = xp01 + d0;
    s.x23 = xp23 + d1;
  }
  // PEst = (I - K*H)*PPred:  col j = (M0*p0j + M1*p1j) [+ (p2j, p3j) for rows 2,3]
  const v2f M0lo = v2f{1.0f, 0.0f} - K0lo, M0hi = v2f{0.0f, 0.0f} - K0hi;
  const v2f M1lo = v2f{0.0f, 1.0f} - K1lo, M1hi = v2f{0.0f, 0.0f} - K1hi;
  v2f qa[4], qb[4], qc[4], qd[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { qa[j] = M0lo * bc(PPlo[j][0]); qb[j] = M0hi * bc(PPlo[j][0]); }
  if constexpr (FMA) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { s.Plo[j] = pk_fma(M1lo, bc(PPlo[j][1]), qa[j]); qb[j] = pk_fma(M1hi, bc(PPlo[j][1]), qb[j]); }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) { qc[j] = M1lo * bc(PPlo[j][1]); qd[j] = M1hi * bc(PPlo[j][1]); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s.Plo[j] = qa[j] + qc[j]; qb[j] = qb[j] + qd[j]; }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) s.Phi[j] = qb[j] + PPhi[j];
}


}  // namespace crx

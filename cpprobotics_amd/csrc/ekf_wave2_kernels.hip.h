// ekf_wave2_kernels.hip.h — the A/B variant north_star sketches for the fused EKF launch: a vehicle spread over lanes.
//
// TWO LANES PER VEHICLE: the even lane of a pair holds rows (0,1) of every covariance column and (x, y); the odd lane rows
// (2,3) and (yaw, v).  65,536 vehicles are then 2,048 waves = two per SIMD, which is the point of the exercise: one wave
// per SIMD is issue-bound, two waves share the SIMD's issue slots.  Each lane evaluates ONE of the step's two
// sincos, the 4x4 products are split by rows, and what the other half needs crosses the pair as DPP quad_perm moves
// ([0,0,2,2] = "the even lane's value on both", [1,1,3,3] = "the odd lane's"): 21 moves per step.
//
// Same arithmetic as ekf_step_packed (ekf_math.h), operation for operation, on its fast domain (0 < |yaw| < 120,
// 2^-60 <= |det S| <= 2^60); there is no general-path fallback here — the kernel reports through *left_domain if a lane
// left the domain, and the production kernel (ekf_run_kernel) is the one to use.  Results equal the production kernel's
// as IEEE values (an added +0 can turn a -0 into +0 in rows 0,1 of PEst; `==` does not see it).
// Measured outcome: profiles/r02/ekf_wave_ab.txt.
#pragma once
#include <hip/hip_runtime.h>
#include "ekf_math.h"

namespace crx {

// single-angle version of sincos_fast2 (ekf_math.h): same operations, same order
__device__ __forceinline__ void sincos_fast1(float y, float& so, float& co, FastDomain& dom) {
  typedef SinCosConsts C;
  dom.amax = __builtin_fmaxf(dom.amax, __builtin_fabsf(y));
  dom.amin = __builtin_fminf(dom.amin, __builtin_fabsf(y));
  double x = (double)y;
  const uint32_t v = (uint32_t)((int32_t)(x * C::hpi_inv) + 0x800000);
  x = __builtin_fma(-(double)((int32_t)v >> 24), C::hpi, x);
  const double x2 = x * x;
  const double x3 = x * x2;
  const double s1 = __builtin_fma(x2, C::s3, C::s2);
  const double x4 = x2 * x2;
  const double c2 = __builtin_fma(x2, C::c4, C::c3);
  const double c1 = __builtin_fma(x2, C::c1, C::c0);
  const double x7 = x3 * x2;
  const double sa = __builtin_fma(x3, C::s1, x);
  const double x6 = x4 * x2;
  const double ca = __builtin_fma(x4, C::c2, c1);
  const double S = __builtin_fma(x7, s1, sa);
  const double Cv = __builtin_fma(x6, c2, ca);
  const uint32_t fs = f2u((float)S), fc = f2u((float)Cv);
  const uint32_t odd = bit24_mask(v);
  const uint32_t sr = bitselect(odd, fc, fs);
  const uint32_t cr = bitselect(odd, fs, fc);
  const uint32_t qs = v << 6;
  const uint32_t qc = qs + 0x40000000u;
  so = u2f(xor_masked(sr, qs, 0x80000000u));
  co = u2f(xor_masked(cr, qc, 0x80000000u));
}

// DPP within a pair of lanes: the even (lo) lane's value on both lanes / the odd (hi) lane's value on both lanes
__device__ __forceinline__ float from_lo(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xA0, 0xf, 0xf, true));   // quad_perm [0,0,2,2]
}
__device__ __forceinline__ float from_hi(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xF5, 0xf, 0xf, true));   // quad_perm [1,1,3,3]
}
__device__ __forceinline__ v2f from_lo(v2f v) { return v2f{from_lo(v.x), from_lo(v.y)}; }
__device__ __forceinline__ v2f from_hi(v2f v) { return v2f{from_hi(v.x), from_hi(v.y)}; }

struct EkfHalf {       // one lane's half of a vehicle
  v2f x;               // lo: (x, y)   hi: (yaw, v)
  v2f P[4];            // column j: lo rows (0,1), hi rows (2,3)
};
struct EkfHalfConsts {
  v2f Q[4];            // the lane's rows of Q
  v2f Rc0, Rc1;
  v2f e0, e1;          // the lane's rows of the first two columns of I4: lo (1,0),(0,1); hi (0,0),(0,0)
  v2f hi_mask;         // (1,1) on the hi lane, (0,0) on the lo lane
  double dt;
  float dtf;
};

// One ekf_estimation() (:64-78) for the pair; `hi` = this lane holds rows (2,3).
__device__ __forceinline__ void ekf_step_pair(EkfHalf& s, const bool hi, v2f z, v2f u, const EkfHalfConsts& k, FastDomain& dom) {
  const float u0 = u.x, u1 = u.y;
  // the step's two angles: yaw (the hi lane's x component) and yaw + DT*u1; the lo lane takes the first, the hi lane the second
  const float yaw0 = from_hi(s.x.x);
  const float yaw1 = yaw0 + k.dtf * u1;
  const float ang = hi ? yaw1 : yaw0;
  float sn, cs;
  sincos_fast1(ang, sn, cs, dom);
  const float s1 = from_hi(sn), c1 = from_hi(cs);           // of yaw1: both lanes need them for jacobF; (sn, cs) on lo = of yaw0
  // xPred = x + B*u, by rows: lo (x + DT cos(yaw0) u0, y + DT sin(yaw0) u0);  hi (yaw + DT u1, v + u0)
  const float b0 = (float)(k.dt * (double)cs), b1 = (float)(k.dt * (double)sn);
  const v2f bvec = hi ? v2f{k.dtf, 1.0f} : v2f{b0, b1};
  const v2f uvec = hi ? v2f{u1, u0} : v2f{u0, u0};
  const v2f xp = s.x + bvec * uvec;
  // jacobF(xPred, u)
  const double dv = k.dt * (double)u0;
  const float j02 = (float)((-dv) * (double)s1);
  const float j03 = (float)(k.dt * (double)c1);
  const float j12 = (float)(dv * (double)c1);
  const float j13 = (float)(k.dt * (double)s1);
  // T1 = jF*PEst: rows (0,1) pick up rows 2,3 of P (the hi lane's), rows (2,3) are unchanged (jA = jB = 0 there)
  const v2f jA = hi ? v2f{0.0f, 0.0f} : v2f{j02, j12};
  const v2f jB = hi ? v2f{0.0f, 0.0f} : v2f{j03, j13};
  v2f q[4], T1[4], ta[4], tb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = from_hi(s.P[j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) ta[j] = jA * bc(q[j].x);
#pragma unroll
  for (int j = 0; j < 4; ++j) tb[j] = jB * bc(q[j].y);
#pragma unroll
  for (int j = 0; j < 4; ++j) ta[j] = s.P[j] + ta[j];
#pragma unroll
  for (int j = 0; j < 4; ++j) T1[j] = ta[j] + tb[j];
  // PPred = T1*jF^T + Q (the same expression for both row pairs)
  v2f PP[4];
  {
    const v2f m0 = T1[2] * bc(j02), m2 = T1[2] * bc(j12);
    const v2f n0 = T1[3] * bc(j03), n2 = T1[3] * bc(j13);
    const v2f a0 = T1[0] + m0, a2 = T1[1] + m2;
    PP[2] = T1[2] + k.Q[2];
    PP[3] = T1[3] + k.Q[3];
    const v2f c0 = a0 + n0, c2 = a2 + n2;
    PP[0] = c0 + k.Q[0];
    PP[1] = c2 + k.Q[1];
  }
  // rows (0,1) of PPred on both lanes: S, K and the last product need them
  v2f L[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) L[j] = from_lo(PP[j]);
  const v2f y = z - from_lo(xp);
  const v2f Sc0 = L[0] + k.Rc0;
  const v2f Sc1 = L[1] + k.Rc1;
  const v2f dd = Sc0 * v2f{Sc1.y, Sc1.x};
  const float det = dd.x - dd.y;
  const float inv = recip_fast(det, dom);
  const float Si00 = Sc1.y * inv, Si10 = -Sc0.y * inv;
  const float Si01 = -Sc1.x * inv, Si11 = Sc0.x * inv;
  // K = (PPred*H^T)*Sinv, this lane's rows
  const v2f K0 = PP[0] * bc(Si00) + PP[1] * bc(Si10);
  const v2f K1 = PP[0] * bc(Si01) + PP[1] * bc(Si11);
  // xEst = xPred + K*y
  s.x = xp + (K0 * bc(y.x) + K1 * bc(y.y));
  // PEst = (I - K*H)*PPred: rows r of column j = M0_r*PP(0,j) + M1_r*PP(1,j) [+ PP_r(j) for rows 2,3]
  const v2f M0 = k.e0 - K0, M1 = k.e1 - K1;
  v2f qa[4], qc[4], ad[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) qa[j] = M0 * bc(L[j].x);
#pragma unroll
  for (int j = 0; j < 4; ++j) qc[j] = M1 * bc(L[j].y);
#pragma unroll
  for (int j = 0; j < 4; ++j) ad[j] = PP[j] * k.hi_mask;
#pragma unroll
  for (int j = 0; j < 4; ++j) qa[j] = qa[j] + qc[j];
#pragma unroll
  for (int j = 0; j < 4; ++j) s.P[j] = qa[j] + ad[j];
}

// n vehicles, T steps; 32 vehicles per wave.  z, u time-major [T][n][2]; x_hist [T][n][4] (may be null).
template <int D>
__global__ void __launch_bounds__(64)
ekf_run_pair_kernel(int n, int T, float* __restrict__ x, float* __restrict__ P, const float* __restrict__ z, const float* __restrict__ u,
                    float* __restrict__ x_hist, EkfConsts kc, int* __restrict__ left_domain) {
  const size_t gl = (size_t)blockIdx.x * 64 + threadIdx.x;
  const size_t a = gl >> 1;
  const bool hi = (gl & 1) != 0;
  const bool live = a < (size_t)n;
  const size_t ag = live ? a : (size_t)n - 1;
  const unsigned r2 = hi ? 2u : 0u;
  EkfHalfConsts k;
#pragma unroll
  for (int j = 0; j < 4; ++j) k.Q[j] = v2f{kc.Q[4 * j + r2], kc.Q[4 * j + r2 + 1]};
  k.Rc0 = v2f{kc.R[0], kc.R[1]}; k.Rc1 = v2f{kc.R[2], kc.R[3]};
  k.e0 = hi ? v2f{0.0f, 0.0f} : v2f{1.0f, 0.0f};
  k.e1 = hi ? v2f{0.0f, 0.0f} : v2f{0.0f, 1.0f};
  k.hi_mask = hi ? v2f{1.0f, 1.0f} : v2f{0.0f, 0.0f};
  k.dt = kc.dt; k.dtf = (float)kc.dt;
  EkfHalf s;
  s.x = reinterpret_cast<const v2f*>(x)[2 * ag + (hi ? 1 : 0)];
#pragma unroll
  for (int j = 0; j < 4; ++j) s.P[j] = reinterpret_cast<const v2f*>(P)[8 * ag + 2 * j + (hi ? 1 : 0)];
  const v2f* __restrict__ z2 = reinterpret_cast<const v2f*>(z);
  const v2f* __restrict__ u2 = reinterpret_cast<const v2f*>(u);
  v2f* __restrict__ xh = reinterpret_cast<v2f*>(x_hist);
  const size_t ns = (size_t)n;
  FastDomain dom = fast_domain_init();
  v2f zq[D], uq[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const size_t t = d < T ? d : 0;
    zq[d] = __builtin_nontemporal_load(&z2[t * ns + ag]);
    uq[d] = __builtin_nontemporal_load(&u2[t * ns + ag]);
  }
  for (int t0 = 0; t0 < T; t0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int t = t0 + d;
      if (t < T) {                                               // wave-uniform
        const v2f zc = zq[d], uc = uq[d];
        const size_t tn = (size_t)(t + D < T ? t + D : T - 1);   // clamped prefetch (unconditional: counted waits)
        zq[d] = __builtin_nontemporal_load(&z2[tn * ns + ag]);
        uq[d] = __builtin_nontemporal_load(&u2[tn * ns + ag]);
        ekf_step_pair(s, hi, zc, uc, k, dom);
        if (x_hist && live) __builtin_nontemporal_store(s.x, &xh[2 * ((size_t)t * ns + a) + (hi ? 1 : 0)]);
      }
    }
  }
  if (!fast_domain_ok(dom) && live && left_domain) atomicOr(left_domain, 1);
  if (!live) return;
  reinterpret_cast<v2f*>(x)[2 * a + (hi ? 1 : 0)] = s.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) reinterpret_cast<v2f*>(P)[8 * a + 2 * j + (hi ? 1 : 0)] = s.P[j];
}

}  // namespace crx

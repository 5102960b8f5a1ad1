// ekf_kernels.hip.h — batched 4-state EKF predict+update for gfx950 (one vehicle per lane).
//
// Replaces, for n independent vehicles at once, the reference's
//   motion_model / jacobF / observation_model / jacobH / ekf_estimation
//   (/root/reference/src/extended_kalman_filter.cpp:22-78)
// and, in the fused kernel, the body of its simulation loop (:171-188).
//
// Arithmetic contract (bit parity with the Eigen path, see DESIGN.md "EKF arithmetic"):
//   * fp32 mul/add in Eigen's per-coefficient order — first term a bare product, then
//     ascending-k multiply-then-add — with NO fma contraction (-ffp-contract=off);
//   * multiplications by the literal 0/1 entries of F_, jF, jH and I-K*jH are dropped, which
//     is exact for finite operands as long as the surviving terms keep their order;
//   * DT*cos(yaw) and friends are formed in double and rounded once to float, as the
//     reference's `#define DT 0.1` (a double literal) makes them;
//   * sin/cos are crx_trig.h (bit-identical to glibc's sinf/cosf);
//   * the 2x2 inverse is Eigen's closed form with one IEEE division.
//
// Data layout in HBM: x [n][4] (one 16-byte load per lane, fully coalesced); P [n][16]
// column-major per vehicle (the layout of a std::vector<Eigen::Matrix4f>); fused kernel streams
// z,u as time-major [T][n][2] (8-byte loads, coalesced) and writes xEst history [T][n][4]
// (16-byte stores, coalesced).  State and covariance live in VGPRs for all T steps.
#pragma once
#include <hip/hip_runtime.h>
#include "crx_trig.h"

namespace crx {

// native vector types (the nontemporal builtins do not take HIP's float2/float4 wrappers)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ void motion_model_dev(float& x0, float& x1, float& x2, float& x3,
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
__device__ __forceinline__ JacF jacobF_dev(float yaw, float v, double dt) {
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
__device__ __forceinline__ void ekf_step_dev(EkfState& s, float z0, float z1, float u0, float u1,
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

__device__ __forceinline__ void load_state(EkfState& s, const float* __restrict__ x,
                                           const float* __restrict__ P, size_t a) {
  const float4 xv = reinterpret_cast<const float4*>(x)[a];
  s.x0 = xv.x; s.x1 = xv.y; s.x2 = xv.z; s.x3 = xv.w;
  const float4* Pv = reinterpret_cast<const float4*>(P) + 4 * a;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 c = Pv[j];
    s.P[4 * j + 0] = c.x; s.P[4 * j + 1] = c.y; s.P[4 * j + 2] = c.z; s.P[4 * j + 3] = c.w;
  }
}

__device__ __forceinline__ void store_P(const EkfState& s, float* __restrict__ P, size_t a) {
  float4* Pv = reinterpret_cast<float4*>(P) + 4 * a;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    Pv[j] = make_float4(s.P[4 * j + 0], s.P[4 * j + 1], s.P[4 * j + 2], s.P[4 * j + 3]);
}

// ---- single step: n vehicles, one ekf_estimation() each -------------------------------------
__global__ void __launch_bounds__(256)
ekf_step_kernel(int n, float* __restrict__ x, float* __restrict__ P, const float* __restrict__ z,
                const float* __restrict__ u, EkfConsts k) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  EkfState s;
  load_state(s, x, P, a);
  const float2 zv = reinterpret_cast<const float2*>(z)[a];
  const float2 uv = reinterpret_cast<const float2*>(u)[a];
  ekf_step_dev(s, zv.x, zv.y, uv.x, uv.y, k);
  reinterpret_cast<float4*>(x)[a] = make_float4(s.x0, s.x1, s.x2, s.x3);
  store_P(s, P, a);
}

// ---- fused T steps: state/covariance stay in registers, z/u stream in, xEst streams out -----
// D = software prefetch distance in steps (z,u for step t+D are requested while step t runs).
template <int D, bool XHIST, bool PHIST>
__global__ void __launch_bounds__(64)
ekf_run_kernel(int n, int T, float* __restrict__ x, float* __restrict__ P,
               const float* __restrict__ z, const float* __restrict__ u,
               float* __restrict__ x_hist, float* __restrict__ P_hist, EkfConsts k) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  const v2f* __restrict__ z2 = reinterpret_cast<const v2f*>(z);
  const v2f* __restrict__ u2 = reinterpret_cast<const v2f*>(u);
  v4f* __restrict__ xh = reinterpret_cast<v4f*>(x_hist);

  v2f zq[D], uq[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < T) {
      zq[d] = __builtin_nontemporal_load(&z2[(size_t)d * n + a]);
      uq[d] = __builtin_nontemporal_load(&u2[(size_t)d * n + a]);
    }
  }
  EkfState s;
  load_state(s, x, P, a);

  // main part: every chunk of D steps whose prefetches (t + D) stay inside [0, T) — no guards, one
  // basic block per chunk, so the compiler can keep all D loads in flight (counted vmcnt waits)
  int t0 = 0;
  for (; t0 + 2 * D <= T; t0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int t = t0 + d;
      ekf_step_dev(s, zq[d].x, zq[d].y, uq[d].x, uq[d].y, k);
      // refill the slot just consumed (its registers are dead now: no copy at the loop back-edge)
      zq[d] = __builtin_nontemporal_load(&z2[(size_t)(t + D) * n + a]);
      uq[d] = __builtin_nontemporal_load(&u2[(size_t)(t + D) * n + a]);
      if (XHIST)
        __builtin_nontemporal_store(v4f{s.x0, s.x1, s.x2, s.x3}, &xh[(size_t)t * n + a]);
      if (PHIST) store_P(s, P_hist, (size_t)t * n + a);
    }
  }
  // tail: fewer than 2*D steps left
  for (; t0 < T; t0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int t = t0 + d;
      if (t < T) {
        const v2f zc = zq[d], uc = uq[d];
        if (t + D < T) {
          zq[d] = __builtin_nontemporal_load(&z2[(size_t)(t + D) * n + a]);
          uq[d] = __builtin_nontemporal_load(&u2[(size_t)(t + D) * n + a]);
        }
        ekf_step_dev(s, zc.x, zc.y, uc.x, uc.y, k);
        if (XHIST)
          __builtin_nontemporal_store(v4f{s.x0, s.x1, s.x2, s.x3}, &xh[(size_t)t * n + a]);
        if (PHIST) store_P(s, P_hist, (size_t)t * n + a);
      }
    }
  }
  reinterpret_cast<float4*>(x)[a] = make_float4(s.x0, s.x1, s.x2, s.x3);
  store_P(s, P, a);
}

// ---- the small reference functions, batched (drop-in completeness) ---------------------------
__global__ void motion_model_kernel(int n, const float* __restrict__ x, const float* __restrict__ u,
                                    float* __restrict__ x_out, double dt) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  float4 xv = reinterpret_cast<const float4*>(x)[a];
  const float2 uv = reinterpret_cast<const float2*>(u)[a];
  motion_model_dev(xv.x, xv.y, xv.z, xv.w, uv.x, uv.y, dt);
  reinterpret_cast<float4*>(x_out)[a] = xv;
}

__global__ void jacobF_kernel(int n, const float* __restrict__ x, const float* __restrict__ u,
                              float* __restrict__ jF, double dt) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  const float4 xv = reinterpret_cast<const float4*>(x)[a];
  const float2 uv = reinterpret_cast<const float2*>(u)[a];
  const JacF j = jacobF_dev(xv.z, uv.x, dt);
  float4* o = reinterpret_cast<float4*>(jF) + 4 * a;
  o[0] = make_float4(1.f, 0.f, 0.f, 0.f);
  o[1] = make_float4(0.f, 1.f, 0.f, 0.f);
  o[2] = make_float4(j.j02, j.j12, 1.f, 0.f);
  o[3] = make_float4(j.j03, j.j13, 0.f, 1.f);
}

__global__ void observation_model_kernel(int n, const float* __restrict__ x, float* __restrict__ zo) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  const float4 xv = reinterpret_cast<const float4*>(x)[a];
  reinterpret_cast<float2*>(zo)[a] = make_float2(xv.x, xv.y);
}

// Input side of the simulation loop (:174-181): ud, xTrue, xDR, z from caller-supplied normals.
template <bool HIST>
__global__ void __launch_bounds__(64)
ekf_simulate_inputs_kernel(int n, int T, const float* __restrict__ u_true, float* __restrict__ xTrue,
                           float* __restrict__ xDR, const float* __restrict__ w,
                           float* __restrict__ z, float* __restrict__ ud,
                           float* __restrict__ xTrue_hist, float* __restrict__ xDR_hist,
                           float q0, float q1, float r0, float r1, double dt) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  const float2 ut = reinterpret_cast<const float2*>(u_true)[a];
  float4 xt = reinterpret_cast<const float4*>(xTrue)[a];
  float4 xd = reinterpret_cast<const float4*>(xDR)[a];
  const float4* w4 = reinterpret_cast<const float4*>(w);
  for (int t = 0; t < T; ++t) {
    const size_t o = (size_t)t * n + a;
    const float4 wv = w4[o];
    const float ud0 = (float)((double)ut.x + (double)wv.x * (double)q0);
    const float ud1 = (float)((double)ut.y + (double)wv.y * (double)q1);
    motion_model_dev(xt.x, xt.y, xt.z, xt.w, ut.x, ut.y, dt);
    motion_model_dev(xd.x, xd.y, xd.z, xd.w, ud0, ud1, dt);
    const float z0 = (float)((double)xt.x + (double)wv.z * (double)r0);
    const float z1 = (float)((double)xt.y + (double)wv.w * (double)r1);
    reinterpret_cast<float2*>(z)[o] = make_float2(z0, z1);
    reinterpret_cast<float2*>(ud)[o] = make_float2(ud0, ud1);
    if (HIST) {
      if (xTrue_hist) reinterpret_cast<float4*>(xTrue_hist)[o] = xt;
      if (xDR_hist) reinterpret_cast<float4*>(xDR_hist)[o] = xd;
    }
  }
  reinterpret_cast<float4*>(xTrue)[a] = xt;
  reinterpret_cast<float4*>(xDR)[a] = xd;
}

}  // namespace crx

// dare_kernels.hip.h — batched discrete Riccati fixed point + LQR gain for gfx950
// (one agent per lane, X and all temporaries in VGPRs).
//
// Replaces, for n independent agents at once,
//   solve_DARE / dlqr  5x5, 2 inputs: /root/reference/src/lqr_speed_steer_control.cpp:85-106
//   solve_DARE / dlqr  4x4, 1 input : /root/reference/src/lqr_steer_control.cpp:75-96
//
// Two kernel families:
//   dare_dense_kernel<DIM>   — arbitrary per-agent A, B, Q, R (the drop-in for solve_DARE()
//                              called with any matrices).  Products are accumulated in the
//                              order Eigen 3.3.9 uses for that shape (DESIGN.md "Eigen order").
//   dare_from_v_kernel<DIM>  — A, B built from the speed exactly as lqr_steering_control()
//                              builds them (:116-129 / :104-115), Q = I, R = I; the literal
//                              0/1 entries are skipped, which leaves at most two non-zero terms
//                              in every sum, so the result is bit-identical to the dense
//                              evaluation at ~1/5 of the flops.
//
// Semantics kept from the reference: cold start X = Q; stop as soon as max|Xn - X| < eps and
// return Xn; otherwise X = Xn and, after `maxiter` evaluations, return X — which is again the
// last evaluated Xn, so the value handed back is always the most recent iterate; no fma
// contraction; one IEEE division per 2x2 inverse.
#pragma once
#include <hip/hip_runtime.h>

namespace crx {

// ---------- tiny register-matrix helpers (column-major, compile-time sizes) -------------------
// Accumulation orders (see oracle/eigen_order.h for the derivation from Eigen's sources).
enum { ORD_ASC = 0, ORD_TREE = 1, ORD_SSE4 = 2 };

// redux_novec_unroller: sum(start,len) = sum(start,len/2) + sum(start+len/2, len-len/2)
template <int START, int LEN>
struct TreeSum {
  static __device__ __forceinline__ float run(const float* t) {
    return TreeSum<START, LEN / 2>::run(t) + TreeSum<START + LEN / 2, LEN - LEN / 2>::run(t);
  }
};
template <int START>
struct TreeSum<START, 1> {
  static __device__ __forceinline__ float run(const float* t) { return t[START]; }
};

template <int K, int ORD>
__device__ __forceinline__ float sum_terms(const float (&t)[K]) {
  if constexpr (ORD == ORD_SSE4 && K >= 4 && K < 8) {
    // vectorised redux (redux_impl<LinearVectorizedTraversal, CompleteUnrolling>): SSE2 predux of the one product packet,
    // then the K % 4 remaining terms (redux_novec_unroller) are added
    const float v = (t[0] + t[2]) + (t[1] + t[3]);
    if constexpr (K == 4) return v;
    else return v + TreeSum<4, K - 4>::run(t);
  } else if constexpr (ORD == ORD_TREE || ORD == ORD_SSE4) {
    return TreeSum<0, K>::run(t);
  } else {
    float s = t[0];
#pragma unroll
    for (int k = 1; k < K; ++k) s = s + t[k];
    return s;
  }
}

// out(RxC) = A(RxK) * B(KxC);  TA/TB: read A/B through a transposed view of the stored matrix.
template <int R, int K, int C, bool TA, bool TB, int ORD>
__device__ __forceinline__ void mm(const float* __restrict__ A, const float* __restrict__ B,
                                   float* __restrict__ out) {
#pragma unroll
  for (int j = 0; j < C; ++j)
#pragma unroll
    for (int i = 0; i < R; ++i) {
      float t[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float a = TA ? A[k + K * i] : A[i + R * k];
        const float b = TB ? B[j + C * k] : B[k + K * j];
        t[k] = a * b;
      }
      out[i + R * j] = sum_terms<K, ORD>(t);
    }
}

__device__ __forceinline__ void inverse2(const float* m, float* r) {
  const float det = m[0] * m[3] - m[1] * m[2];
  const float invdet = 1.0f / det;
  r[0] = m[3] * invdet;
  r[1] = -m[1] * invdet;
  r[2] = -m[2] * invdet;
  r[3] = m[0] * invdet;
}

// ---------- dense 5x5 ------------------------------------------------------------------------
// Eigen order for 5-row shapes: every product falls on the coefficient path; the redux is vectorised (SSE4: one packet +
// the fifth term) where the left factor is a transposed view (A'*X, B'*X), the unrolled tree (TREE) elsewhere.
__device__ __forceinline__ void dare5_dense_iter(const float* A, const float* B, const float* Q,
                                                 const float* R, const float* X, float* Xn) {
  float AtX[25], P1[25], BtX[10], G[4], Sg[4], Si[4], c1[10], c2[10], c3[25], c4[25], P2[25];
  mm<5, 5, 5, true, false, ORD_SSE4>(A, X, AtX);
  mm<5, 5, 5, false, false, ORD_TREE>(AtX, A, P1);
  mm<2, 5, 5, true, false, ORD_SSE4>(B, X, BtX);
  mm<2, 5, 2, false, false, ORD_TREE>(BtX, B, G);
#pragma unroll
  for (int i = 0; i < 4; ++i) Sg[i] = R[i] + G[i];
  inverse2(Sg, Si);
  mm<5, 5, 2, false, false, ORD_TREE>(AtX, B, c1);
  mm<5, 2, 2, false, false, ORD_TREE>(c1, Si, c2);
  mm<5, 2, 5, false, true, ORD_TREE>(c2, B, c3);
  mm<5, 5, 5, false, false, ORD_TREE>(c3, X, c4);
  mm<5, 5, 5, false, false, ORD_TREE>(c4, A, P2);
#pragma unroll
  for (int i = 0; i < 25; ++i) Xn[i] = (P1[i] - P2[i]) + Q[i];
}

__device__ __forceinline__ void dlqr5_dense_gain(const float* A, const float* B, const float* R,
                                                 const float* X, float* Kout) {
  float BtX[10], G[4], Sg[4], Si[4], BtXA[10];
  mm<2, 5, 5, true, false, ORD_SSE4>(B, X, BtX);
  mm<2, 5, 2, false, false, ORD_TREE>(BtX, B, G);
#pragma unroll
  for (int i = 0; i < 4; ++i) Sg[i] = G[i] + R[i];
  inverse2(Sg, Si);
  mm<2, 5, 5, false, false, ORD_TREE>(BtX, A, BtXA);
  mm<2, 2, 5, false, false, ORD_TREE>(Si, BtXA, Kout);
}

// ---------- dense 4x4 ------------------------------------------------------------------------
// Eigen order for 4-row shapes: column-major left factor -> packet path (ASC); transposed /
// row-vector left factor with inner size 4 -> vectorised redux (SSE4).
__device__ __forceinline__ void dare4_dense_iter(const float* A, const float* B, const float* Q,
                                                 float R, const float* X, float* Xn) {
  float AtX[16], P1[16], BtX[4], g[1], c1[4], c2[4], c3[16], c4[16], P2[16];
  mm<4, 4, 4, true, false, ORD_SSE4>(A, X, AtX);
  mm<4, 4, 4, false, false, ORD_ASC>(AtX, A, P1);
  mm<1, 4, 4, true, false, ORD_SSE4>(B, X, BtX);
  mm<1, 4, 1, false, false, ORD_SSE4>(BtX, B, g);
  const float s = R + g[0];
  mm<4, 4, 1, false, false, ORD_ASC>(AtX, B, c1);
#pragma unroll
  for (int i = 0; i < 4; ++i) c2[i] = c1[i] / s;
  mm<4, 1, 4, false, true, ORD_ASC>(c2, B, c3);
  mm<4, 4, 4, false, false, ORD_ASC>(c3, X, c4);
  mm<4, 4, 4, false, false, ORD_ASC>(c4, A, P2);
#pragma unroll
  for (int i = 0; i < 16; ++i) Xn[i] = (P1[i] - P2[i]) + Q[i];
}

__device__ __forceinline__ void dlqr4_dense_gain(const float* A, const float* B, float R,
                                                 const float* X, float* Kout) {
  float BtX[4], g[1], BtXA[4];
  mm<1, 4, 4, true, false, ORD_SSE4>(B, X, BtX);
  mm<1, 4, 1, false, false, ORD_SSE4>(BtX, B, g);
  const float inv = (float)(1.0 / (double)(g[0] + R));
  mm<1, 4, 4, false, false, ORD_SSE4>(BtX, A, BtXA);
#pragma unroll
  for (int j = 0; j < 4; ++j) Kout[j] = inv * BtXA[j];
}

// (Xn - X).cwiseAbs().maxCoeff(): a strict '>' scan from element 0 in which a NaN element never replaces the
// running maximum, and a NaN FIRST element sticks.  v_max_f32 has exactly that behaviour for every element but the
// first (max(m, NaN) = m), so the scan is one subtract + one max per element, with the first-element NaN patched in.
template <int N>
__device__ __forceinline__ float max_abs_diff(const float* a, const float* b) {
  const float m0 = fabsf(a[0] - b[0]);
  float m = m0;
#pragma unroll
  for (int i = 1; i < N; ++i) m = fmaxf(m, fabsf(a[i] - b[i]));
  return (m0 != m0) ? m0 : m;
}

// The reference's fixed-point loop (solve_DARE :86-99): X = Q; repeat Xn = f(X); stop when max|Xn - X| < eps (return Xn)
// else X = Xn; after maxiter evaluations return the last one.  Either way the value handed back is the most recent
// iterate.  Written as a two-buffer ping-pong (X -> Y -> X ...) so that no 16/25-register copy sits in the loop; lanes
// that have converged are masked off (their buffers are no longer written) while the wave finishes its slowest agent.
// X holds the start value on entry and the result on return; returns the number of evaluations performed.
template <int NN, class IterFn>
__device__ __forceinline__ int riccati_fixed_point(float* X, float eps, int maxiter, bool live, IterFn iter) {
  float Y[NN];
  bool done = !live || maxiter <= 0;
  bool in_y = false;                      // which buffer holds this lane's most recent iterate
  int it = maxiter < 0 ? 0 : maxiter;
  for (int i = 0; i < maxiter; i += 2) {
    if (!done) {
      iter(X, Y);
      in_y = true;
      if (max_abs_diff<NN>(Y, X) < eps) { done = true; it = i + 1; }
    }
    if (!done && i + 1 < maxiter) {
      iter(Y, X);
      in_y = false;
      if (max_abs_diff<NN>(X, Y) < eps) { done = true; it = i + 2; }
    }
    if (__all(done)) break;
  }
  if (in_y) {
#pragma unroll
    for (int j = 0; j < NN; ++j) X[j] = Y[j];
  }
  return it;
}

template <int DIM>
__global__ void __launch_bounds__(64)
dare_dense_kernel(int n, const float* __restrict__ Ag, const float* __restrict__ Bg,
                  const float* __restrict__ Qg, const float* __restrict__ Rg, float eps, int maxiter,
                  float* __restrict__ Xg, float* __restrict__ Kg, int* __restrict__ iters) {
  constexpr int NN = DIM * DIM;
  constexpr int M = (DIM == 5) ? 2 : 1;
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = a < (size_t)n;
  const size_t ai = live ? a : 0;
  float A[NN], B[DIM * M], Q[NN], R[M * M], X[NN];
#pragma unroll
  for (int i = 0; i < NN; ++i) { A[i] = Ag[ai * NN + i]; Q[i] = Qg[ai * NN + i]; X[i] = Q[i]; }
#pragma unroll
  for (int i = 0; i < DIM * M; ++i) B[i] = Bg[ai * DIM * M + i];
#pragma unroll
  for (int i = 0; i < M * M; ++i) R[i] = Rg[ai * M * M + i];

  const int it = riccati_fixed_point<NN>(X, eps, maxiter, live, [&](const float* Xi, float* Xo) {
    if (DIM == 5) dare5_dense_iter(A, B, Q, R, Xi, Xo);
    else dare4_dense_iter(A, B, Q, R[0], Xi, Xo);
  });
  if (!live) return;
  if (Xg) {
#pragma unroll
    for (int j = 0; j < NN; ++j) Xg[a * NN + j] = X[j];
  }
  if (Kg) {
    float K[M * DIM];
    if (DIM == 5) dlqr5_dense_gain(A, B, R, X, K);
    else dlqr4_dense_gain(A, B, R[0], X, K);
#pragma unroll
    for (int j = 0; j < M * DIM; ++j) Kg[a * M * DIM + j] = K[j];
  }
  if (iters) iters[a] = it;
}

// ---------- structured: A, B from v; Q = I; R = I ----------------------------------------------
// 5x5 (:116-129): A00=1 A01=dt A12=v A22=1 A23=dt A44=1 ; B30=v/L B41=dt.
__device__ __forceinline__ void dare5_v_iter(float dt, float v, float bv, float bd, const float* X,
                                             float* Xn) {
  float AtX[25], c4[25];
  float c2[10];
  // A'X : row0 = X0., row1 = dt*X0., row2 = v*X1. + X2., row3 = dt*X2., row4 = X4.
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    AtX[0 + 5 * j] = X[0 + 5 * j];
    AtX[1 + 5 * j] = dt * X[0 + 5 * j];
    AtX[2 + 5 * j] = v * X[1 + 5 * j] + X[2 + 5 * j];
    AtX[3 + 5 * j] = dt * X[2 + 5 * j];
    AtX[4 + 5 * j] = X[4 + 5 * j];
  }
  // G = (B'X)B ; B'X row0 = bv*X3., row1 = bd*X4.
  const float G00 = (bv * X[3 + 5 * 3]) * bv, G10 = (bd * X[4 + 5 * 3]) * bv;
  const float G01 = (bv * X[3 + 5 * 4]) * bd, G11 = (bd * X[4 + 5 * 4]) * bd;
  float Sg[4] = {1.0f + G00, 0.0f + G10, 0.0f + G01, 1.0f + G11}, Si[4];
  inverse2(Sg, Si);
  // c1 = (A'X)B : col0 = AtX.3*bv, col1 = AtX.4*bd ; c2 = c1*Si
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float c10 = AtX[i + 5 * 3] * bv, c11 = AtX[i + 5 * 4] * bd;
    c2[i] = c10 * Si[0] + c11 * Si[1];
    c2[i + 5] = c10 * Si[2] + c11 * Si[3];
  }
  // c3 = c2*B' : only columns 3 (c2.0*bv) and 4 (c2.1*bd) ; c4 = c3*X
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float c33 = c2[i] * bv, c34 = c2[i + 5] * bd;
#pragma unroll
    for (int j = 0; j < 5; ++j) c4[i + 5 * j] = c33 * X[3 + 5 * j] + c34 * X[4 + 5 * j];
  }
  // Xn = ((A'X)A - c4*A) + I ; (M*A) col0 = M.0, col1 = M.0*dt, col2 = M.1*v + M.2, col3 = M.2*dt, col4 = M.4
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float p10 = AtX[i], p11 = AtX[i] * dt, p12 = AtX[i + 5] * v + AtX[i + 10],
                p13 = AtX[i + 10] * dt, p14 = AtX[i + 20];
    const float p20 = c4[i], p21 = c4[i] * dt, p22 = c4[i + 5] * v + c4[i + 10],
                p23 = c4[i + 10] * dt, p24 = c4[i + 20];
    Xn[i + 0] = (p10 - p20) + (i == 0 ? 1.0f : 0.0f);
    Xn[i + 5] = (p11 - p21) + (i == 1 ? 1.0f : 0.0f);
    Xn[i + 10] = (p12 - p22) + (i == 2 ? 1.0f : 0.0f);
    Xn[i + 15] = (p13 - p23) + (i == 3 ? 1.0f : 0.0f);
    Xn[i + 20] = (p14 - p24) + (i == 4 ? 1.0f : 0.0f);
  }
}

__device__ __forceinline__ void dlqr5_v_gain(float dt, float v, float bv, float bd, const float* X,
                                             float* K) {
  float BtX[10];
#pragma unroll
  for (int j = 0; j < 5; ++j) { BtX[0 + 2 * j] = bv * X[3 + 5 * j]; BtX[1 + 2 * j] = bd * X[4 + 5 * j]; }
  float Sg[4] = {BtX[0 + 2 * 3] * bv + 1.0f, BtX[1 + 2 * 3] * bv + 0.0f,
                 BtX[0 + 2 * 4] * bd + 0.0f, BtX[1 + 2 * 4] * bd + 1.0f}, Si[4];
  inverse2(Sg, Si);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float a0 = BtX[i], a1 = BtX[i] * dt, a2 = BtX[i + 2] * v + BtX[i + 4], a3 = BtX[i + 4] * dt,
                a4 = BtX[i + 8];
    BtX[i] = a0; BtX[i + 2] = a1; BtX[i + 4] = a2; BtX[i + 6] = a3; BtX[i + 8] = a4;  // now (B'X)A
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    K[0 + 2 * j] = Si[0] * BtX[0 + 2 * j] + Si[2] * BtX[1 + 2 * j];
    K[1 + 2 * j] = Si[1] * BtX[0 + 2 * j] + Si[3] * BtX[1 + 2 * j];
  }
}

// 4x4 (:104-115): A00=1 A01=dt A12=v A22=1 A23=dt ; B3=v/L ; R=1.
__device__ __forceinline__ void dare4_v_iter(float dt, float v, float bv, const float* X, float* Xn) {
  float AtX[16], c4[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    AtX[0 + 4 * j] = X[0 + 4 * j];
    AtX[1 + 4 * j] = dt * X[0 + 4 * j];
    AtX[2 + 4 * j] = X[2 + 4 * j] + v * X[1 + 4 * j];
    AtX[3 + 4 * j] = dt * X[2 + 4 * j];
  }
  const float g = (bv * X[3 + 4 * 3]) * bv;
  const float s = 1.0f + g;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float c2 = (AtX[i + 12] * bv) / s;
    const float c33 = c2 * bv;
#pragma unroll
    for (int j = 0; j < 4; ++j) c4[i + 4 * j] = c33 * X[3 + 4 * j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float p10 = AtX[i], p11 = AtX[i] * dt, p12 = AtX[i + 4] * v + AtX[i + 8], p13 = AtX[i + 8] * dt;
    const float p20 = c4[i], p21 = c4[i] * dt, p22 = c4[i + 4] * v + c4[i + 8], p23 = c4[i + 8] * dt;
    Xn[i + 0] = (p10 - p20) + (i == 0 ? 1.0f : 0.0f);
    Xn[i + 4] = (p11 - p21) + (i == 1 ? 1.0f : 0.0f);
    Xn[i + 8] = (p12 - p22) + (i == 2 ? 1.0f : 0.0f);
    Xn[i + 12] = (p13 - p23) + (i == 3 ? 1.0f : 0.0f);
  }
}

__device__ __forceinline__ void dlqr4_v_gain(float dt, float v, float bv, const float* X, float* K) {
  const float b0 = bv * X[3 + 0], b1 = bv * X[3 + 4], b2 = bv * X[3 + 8], b3 = bv * X[3 + 12];
  const float g = b3 * bv;
  const float inv = (float)(1.0 / (double)(g + 1.0f));
  K[0] = inv * b0;
  K[1] = inv * (b0 * dt);
  K[2] = inv * (b2 + b1 * v);
  K[3] = inv * (b2 * dt);
}

template <int DIM>
__global__ void __launch_bounds__(64)
dare_from_v_kernel(int n, const float* __restrict__ vg, float dt, double L, float eps, int maxiter,
                   float* __restrict__ Xg, float* __restrict__ Kg, int* __restrict__ iters) {
  constexpr int NN = DIM * DIM;
  constexpr int M = (DIM == 5) ? 2 : 1;
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = a < (size_t)n;
  const float v = live ? vg[a] : 1.0f;
  const float bv = (float)((double)v / L);  // B(3,0) = state.v / L  (float / double literal)
  float X[NN];
#pragma unroll
  for (int i = 0; i < NN; ++i) X[i] = (i % (DIM + 1) == 0) ? 1.0f : 0.0f;
  const int it = riccati_fixed_point<NN>(X, eps, maxiter, live, [&](const float* Xi, float* Xo) {
    if (DIM == 5) dare5_v_iter(dt, v, bv, dt, Xi, Xo);
    else dare4_v_iter(dt, v, bv, Xi, Xo);
  });
  if (!live) return;
  if (Xg) {
#pragma unroll
    for (int j = 0; j < NN; ++j) Xg[a * NN + j] = X[j];
  }
  if (Kg) {
    float K[M * DIM];
    if (DIM == 5) dlqr5_v_gain(dt, v, bv, dt, X, K);
    else dlqr4_v_gain(dt, v, bv, X, K);
#pragma unroll
    for (int j = 0; j < M * DIM; ++j) Kg[a * M * DIM + j] = K[j];
  }
  if (iters) iters[a] = it;
}

}  // namespace crx

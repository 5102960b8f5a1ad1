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
#include <type_traits>

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
// 5x5 (:116-129): A00=1 A01=dt A12=v A22=1 A23=dt A44=1 ; B30=v/L B41=dt.  The iteration itself is dare5_v_iter_pk below.
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

// 4x4 (:104-115): A00=1 A01=dt A12=v A22=1 A23=dt ; B3=v/L ; R=1.  The iteration itself is dare4_v_iter_pk below.
__device__ __forceinline__ void dlqr4_v_gain(float dt, float v, float bv, const float* X, float* K) {
  const float b0 = bv * X[3 + 0], b1 = bv * X[3 + 4], b2 = bv * X[3 + 8], b3 = bv * X[3 + 12];
  const float g = b3 * bv;
  const float inv = (float)(1.0 / (double)(g + 1.0f));
  K[0] = inv * b0;
  K[1] = inv * (b0 * dt);
  K[2] = inv * (b2 + b1 * v);
  K[3] = inv * (b2 * dt);
}

// ---------- structured iterations on packed rows -----------------------------------------------------------------------
// A'X, (A'X)B, its product with the 2x2 inverse, with B', with X and with A — written for the literal 0/1 structure of A and B
// (at most two non-zero terms per sum, so every coefficient equals the dense Eigen-order evaluation bit for bit:
// tests/test_lqr_gpu.py::test_dare_dense_matches_structured_and_oracle) and laid out so that two neighbouring columns of a
// row share one packed fp32 instruction: X is held as rows of column pairs (0,1), (2,3) (+ column 4).  v_pk_mul_f32 /
// v_pk_add_f32 are two independent IEEE operations — the bits are those of the scalar form; the instruction count is not
// (the compiler's own pairing of a scalar formulation spent a fifth of the loop on register moves: -7 % / -9 % VALU).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f bc2(float x) { return (v2f){x, x}; }
__device__ __forceinline__ float max_abs2(float m, v2f d) { return fmaxf(fmaxf(m, fabsf(d.x)), fabsf(d.y)); }

struct Row5 { v2f a, b; float c; };   // columns (0,1), (2,3), 4 of one row
struct Row4 { v2f a, b; };            // columns (0,1), (2,3)

__device__ __forceinline__ void dare5_v_iter_pk(float dt, float v, float bv, float bd, const Row5* X, Row5* Xn) {
  Row5 R[5];                                                   // A'X, row by row
  R[0] = X[0];
  R[1].a = bc2(dt) * X[0].a; R[1].b = bc2(dt) * X[0].b; R[1].c = dt * X[0].c;
  R[2].a = bc2(v) * X[1].a + X[2].a; R[2].b = bc2(v) * X[1].b + X[2].b; R[2].c = v * X[1].c + X[2].c;
  R[3].a = bc2(dt) * X[2].a; R[3].b = bc2(dt) * X[2].b; R[3].c = dt * X[2].c;
  R[4] = X[4];
  const float G00 = (bv * X[3].b.y) * bv, G10 = (bd * X[4].b.y) * bv;
  const float G01 = (bv * X[3].c) * bd, G11 = (bd * X[4].c) * bd;
  float Sg[4] = {1.0f + G00, 0.0f + G10, 0.0f + G01, 1.0f + G11}, Si[4];
  inverse2(Sg, Si);
  const v2f si02 = {Si[0], Si[2]}, si13 = {Si[1], Si[3]}, bvd = {bv, bd};
  const v2f vdt = {v, dt}, one_dt = {1.0f, dt};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float c10 = R[i].b.y * bv, c11 = R[i].c * bd;        // ((A'X)B) row i
    const v2f c2 = bc2(c10) * si02 + bc2(c11) * si13;           // * Si
    const v2f c3 = c2 * bvd;                                   // the two non-zero columns (3, 4) of (..)*B'
    Row5 C;                                                    // row i of (..)*X
    C.a = bc2(c3.x) * X[3].a + bc2(c3.y) * X[4].a;
    C.b = bc2(c3.x) * X[3].b + bc2(c3.y) * X[4].b;
    C.c = c3.x * X[3].c + c3.y * X[4].c;
    // (M*A) columns: 0 = M.0, 1 = M.0*dt, 2 = M.1*v + M.2, 3 = M.2*dt, 4 = M.4.   x*1.0f = x and x + (-0.0f) = x bit for bit
    const v2f p1a = bc2(R[i].a.x) * one_dt, p2a = bc2(C.a.x) * one_dt;
    const v2f p1b = (v2f){R[i].a.y, R[i].b.x} * vdt + (v2f){R[i].b.x, -0.0f};
    const v2f p2b = (v2f){C.a.y, C.b.x} * vdt + (v2f){C.b.x, -0.0f};
    Xn[i].a = (p1a - p2a) + (v2f){i == 0 ? 1.0f : 0.0f, i == 1 ? 1.0f : 0.0f};
    Xn[i].b = (p1b - p2b) + (v2f){i == 2 ? 1.0f : 0.0f, i == 3 ? 1.0f : 0.0f};
    Xn[i].c = (R[i].c - C.c) + (i == 4 ? 1.0f : 0.0f);
  }
}

__device__ __forceinline__ void dare4_v_iter_pk(float dt, float v, float bv, const Row4* X, Row4* Xn) {
  Row4 R[4];
  R[0] = X[0];
  R[1].a = bc2(dt) * X[0].a; R[1].b = bc2(dt) * X[0].b;
  R[2].a = X[2].a + bc2(v) * X[1].a; R[2].b = X[2].b + bc2(v) * X[1].b;
  R[3].a = bc2(dt) * X[2].a; R[3].b = bc2(dt) * X[2].b;
  const float g = (bv * X[3].b.y) * bv;
  const float s = 1.0f + g;
  const v2f vdt = {v, dt}, one_dt = {1.0f, dt};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float c2 = (R[i].b.y * bv) / s;
    const float c33 = c2 * bv;
    Row4 C;
    C.a = bc2(c33) * X[3].a; C.b = bc2(c33) * X[3].b;
    const v2f p1a = bc2(R[i].a.x) * one_dt, p2a = bc2(C.a.x) * one_dt;
    const v2f p1b = (v2f){R[i].a.y, R[i].b.x} * vdt + (v2f){R[i].b.x, -0.0f};
    const v2f p2b = (v2f){C.a.y, C.b.x} * vdt + (v2f){C.b.x, -0.0f};
    Xn[i].a = (p1a - p2a) + (v2f){i == 0 ? 1.0f : 0.0f, i == 1 ? 1.0f : 0.0f};
    Xn[i].b = (p1b - p2b) + (v2f){i == 2 ? 1.0f : 0.0f, i == 3 ? 1.0f : 0.0f};
  }
}

// max |Y - X| with the scan semantics of max_abs_diff: element (0,0) first, NaN there sticks.
__device__ __forceinline__ float max_abs_diff_rows(const Row5* Y, const Row5* X) {
  const v2f d0 = Y[0].a - X[0].a;
  const float m0 = fabsf(d0.x);
  float m = fmaxf(m0, fabsf(d0.y));
  m = max_abs2(m, Y[0].b - X[0].b); m = fmaxf(m, fabsf(Y[0].c - X[0].c));
#pragma unroll
  for (int i = 1; i < 5; ++i) { m = max_abs2(m, Y[i].a - X[i].a); m = max_abs2(m, Y[i].b - X[i].b); m = fmaxf(m, fabsf(Y[i].c - X[i].c)); }
  return (m0 != m0) ? m0 : m;
}
__device__ __forceinline__ float max_abs_diff_rows(const Row4* Y, const Row4* X) {
  const v2f d0 = Y[0].a - X[0].a;
  const float m0 = fabsf(d0.x);
  float m = fmaxf(m0, fabsf(d0.y));
  m = max_abs2(m, Y[0].b - X[0].b);
#pragma unroll
  for (int i = 1; i < 4; ++i) { m = max_abs2(m, Y[i].a - X[i].a); m = max_abs2(m, Y[i].b - X[i].b); }
  return (m0 != m0) ? m0 : m;
}

// The reference's loop (see riccati_fixed_point) for A, B built from v, Q = I, R = I, on packed rows.  Xcm receives the
// result column-major like every other X of this file; returns the number of evaluations.
template <int DIM>
__device__ __forceinline__ int riccati_from_v(float dt, float v, float bv, float bd, float eps, int maxiter, bool live, float* Xcm) {
  using Row = typename std::conditional<DIM == 5, Row5, Row4>::type;
  Row X[DIM], Y[DIM];
#pragma unroll
  for (int i = 0; i < DIM; ++i) {
    X[i].a = (v2f){i == 0 ? 1.0f : 0.0f, i == 1 ? 1.0f : 0.0f};
    X[i].b = (v2f){i == 2 ? 1.0f : 0.0f, i == 3 ? 1.0f : 0.0f};
    if constexpr (DIM == 5) X[i].c = (i == 4) ? 1.0f : 0.0f;
  }
  auto iter = [&](const Row* Xi, Row* Xo) {
    if constexpr (DIM == 5) dare5_v_iter_pk(dt, v, bv, bd, Xi, Xo);
    else dare4_v_iter_pk(dt, v, bv, Xi, Xo);
  };
  bool done = !live || maxiter <= 0;
  bool in_y = false;
  int it = maxiter < 0 ? 0 : maxiter;
  for (int i = 0; i < maxiter; i += 2) {
    if (!done) {
      iter(X, Y);
      in_y = true;
      if (max_abs_diff_rows(Y, X) < eps) { done = true; it = i + 1; }
    }
    if (!done && i + 1 < maxiter) {
      iter(Y, X);
      in_y = false;
      if (max_abs_diff_rows(X, Y) < eps) { done = true; it = i + 2; }
    }
    if (__all(done)) break;
  }
#pragma unroll
  for (int i = 0; i < DIM; ++i) {
    const Row& r = in_y ? Y[i] : X[i];
    Xcm[i + DIM * 0] = r.a.x; Xcm[i + DIM * 1] = r.a.y; Xcm[i + DIM * 2] = r.b.x; Xcm[i + DIM * 3] = r.b.y;
    if constexpr (DIM == 5) Xcm[i + DIM * 4] = r.c;
  }
  return it;
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
  const int it = riccati_from_v<DIM>(dt, v, bv, dt, eps, maxiter, live, X);
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

// dare_dense_math.h — the DENSE Riccati iteration of the crx engine: arbitrary A, B, Q, R (the solve_DARE(A, B, Q, R) signature of
// /root/reference/src/lqr_speed_steer_control.cpp:85-106 and src/lqr_steer_control.cpp:75-96 handed matrices that do not carry
// lqr_steering_control's pattern), every product accumulated in the order Eigen 3.3.9 uses for its shape (oracle/eigen_order.h).
// gfx950 device code that also compiles as plain host C++ (tests/tools/dare_host.cpp checks it against the oracle on the CPU).
// Two layouts: the whole agent in one lane (dare5_dense_iter / dare4_dense_iter), and — round 4 — the agent on the four lanes of a
// quad, one row of X per lane (dare_dense_quad_rows below).
#pragma once
#include "dare_math.h"

#if defined(__HIPCC__)
#define CRX_HDM __host__ __device__ __forceinline__     // (CRX_HD carries `static` in host builds: member functions need their own)
#else
#define CRX_HDM inline
#endif

namespace crx {

// ---------- tiny register-matrix helpers (column-major, compile-time sizes) -------------------
// Accumulation orders (see oracle/eigen_order.h for the derivation from Eigen's sources).
enum { ORD_ASC = 0, ORD_TREE = 1, ORD_SSE4 = 2, ORD_SLICE = 3 };   // ORD_SLICE: per row, see mm

// redux_novec_unroller: sum(start,len) = sum(start,len/2) + sum(start+len/2, len-len/2)
template <int START, int LEN>
struct TreeSum {
  static CRX_HDM float run(const float* t) {
    return TreeSum<START, LEN / 2>::run(t) + TreeSum<START + LEN / 2, LEN - LEN / 2>::run(t);
  }
};
template <int START>
struct TreeSum<START, 1> {
  static CRX_HDM float run(const float* t) { return t[START]; }
};

template <int K, int ORD>
CRX_HD float sum_terms(const float (&t)[K]) {
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
// ORD_SLICE: a column-major left factor with R >= 4, R % 4 != 0 rows (SliceVectorizedTraversal with inner unrolling): the first
// (R/4)*4 rows of every column are packet sums (ascending), the remaining rows coeff() reduxes (the unrolled tree).
template <int R, int K, int C, bool TA, bool TB, int ORD>
CRX_HD void mm(const float* __restrict__ A, const float* __restrict__ B,
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
      if constexpr (ORD == ORD_SLICE) out[i + R * j] = (i < (R / 4) * 4) ? sum_terms<K, ORD_ASC>(t) : sum_terms<K, ORD_TREE>(t);
      else out[i + R * j] = sum_terms<K, ORD>(t);
    }
}

CRX_HD void inverse2(const float* m, float* r) {
  const float det = m[0] * m[3] - m[1] * m[2];
  const float invdet = 1.0f / det;
  r[0] = m[3] * invdet;
  r[1] = -m[1] * invdet;
  r[2] = -m[2] * invdet;
  r[3] = m[0] * invdet;
}

// ---------- dense 5x5 ------------------------------------------------------------------------
// Eigen order for 5-row shapes (oracle/eigen_order.h): a transposed left factor (A'*X, B'*X) puts the product on the coefficient
// path with a vectorised redux (SSE4: one packet + the fifth term); a column-major 5-row left factor is assigned by slices —
// rows 0-3 of each column by packets (ascending), row 4 by the unrolled-tree redux (SLICE); a 2-row left factor stays on the
// coefficient path (TREE).  Inner size 2 has one order.
CRX_HD void dare5_dense_iter(const float* A, const float* B, const float* Q,
                                                 const float* R, const float* X, float* Xn) {
  float AtX[25], P1[25], BtX[10], G[4], Sg[4], Si[4], c1[10], c2[10], c3[25], c4[25], P2[25];
  mm<5, 5, 5, true, false, ORD_SSE4>(A, X, AtX);
  mm<5, 5, 5, false, false, ORD_SLICE>(AtX, A, P1);
  mm<2, 5, 5, true, false, ORD_SSE4>(B, X, BtX);
  mm<2, 5, 2, false, false, ORD_TREE>(BtX, B, G);
#pragma unroll
  for (int i = 0; i < 4; ++i) Sg[i] = R[i] + G[i];
  inverse2(Sg, Si);
  mm<5, 5, 2, false, false, ORD_SLICE>(AtX, B, c1);
  mm<5, 2, 2, false, false, ORD_TREE>(c1, Si, c2);
  mm<5, 2, 5, false, true, ORD_TREE>(c2, B, c3);
  mm<5, 5, 5, false, false, ORD_SLICE>(c3, X, c4);
  mm<5, 5, 5, false, false, ORD_SLICE>(c4, A, P2);
#pragma unroll
  for (int i = 0; i < 25; ++i) Xn[i] = (P1[i] - P2[i]) + Q[i];
}

CRX_HD void dlqr5_dense_gain(const float* A, const float* B, const float* R,
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
CRX_HD void dare4_dense_iter(const float* A, const float* B, const float* Q,
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

CRX_HD void dlqr4_dense_gain(const float* A, const float* B, float R,
                                                 const float* X, float* Kout) {
  float BtX[4], g[1], BtXA[4];
  mm<1, 4, 4, true, false, ORD_SSE4>(B, X, BtX);
  mm<1, 4, 1, false, false, ORD_SSE4>(BtX, B, g);
  const float inv = (float)(1.0 / (double)(g[0] + R));
  mm<1, 4, 4, false, false, ORD_SSE4>(BtX, A, BtXA);
#pragma unroll
  for (int j = 0; j < 4; ++j) Kout[j] = inv * BtXA[j];
}


// ---------- four lanes per agent (round 4) ----------------------------------------------------------------------------------------
// The dense iteration with one row of X per lane of a quad: lane r (0..3) produces row r of Xn and — 5x5 only — every lane also
// produces row 4 (redundantly: a fifth row does not fit a quad, and it is needed by all of them as part of the right operand X).
// Row i of a product L*M is sum_k L(i,k) M(k,:): a lane needs ITS row of the left operand, which it has just produced, and the WHOLE
// right operand — A, B (constants, held by every lane), Si (2x2, computed by every lane) or X, whose rows 0-3 the caller hands in
// as gathered from the four lanes (DPP quad_perm broadcasts in the kernel) — so the chain of :91 runs without any exchange but that
// gather, once per evaluation.  Same products, same accumulation order per coefficient as dare5_dense_iter / dare4_dense_iter
// (rows 0-3 of a 5-row column-major left factor by packets = ascending, row 4 by the unrolled tree; transposed left factors by the
// vectorised redux): the same bits.  A lane executes 2/5 of an agent's evaluation (5x5) or 1/4 (4x4).
//   Acol_r / Acol_4: columns r and 4 of A (= rows r and 4 of A');  Qrow_r / Qrow_4: rows r and 4 of Q;  Xf: the whole X, column-major.
template <int DIM>
CRX_HD void dare_dense_quad_rows(const float* Acol_r, const float* Acol_4, const float* Acm, const float* Bcm, const float* Qrow_r,
                                 const float* Qrow_4, const float* R, const float* Xf, float* xn_r, float* xn_4) {
  if constexpr (DIM == 5) {
    float AtX_r[5], AtX_4[5], P1_r[5], P1_4[5], BtX[10], G[4], Sg[4], Si[4];
    mm<1, 5, 5, false, false, ORD_SSE4>(Acol_r, Xf, AtX_r);
    mm<1, 5, 5, false, false, ORD_SSE4>(Acol_4, Xf, AtX_4);
    mm<1, 5, 5, false, false, ORD_ASC>(AtX_r, Acm, P1_r);
    mm<1, 5, 5, false, false, ORD_TREE>(AtX_4, Acm, P1_4);
    mm<2, 5, 5, true, false, ORD_SSE4>(Bcm, Xf, BtX);
    mm<2, 5, 2, false, false, ORD_TREE>(BtX, Bcm, G);
#pragma unroll
    for (int i = 0; i < 4; ++i) Sg[i] = R[i] + G[i];
    inverse2(Sg, Si);
    float c1_r[2], c1_4[2], c2_r[2], c2_4[2], c3_r[5], c3_4[5], c4_r[5], c4_4[5], P2_r[5], P2_4[5];
    mm<1, 5, 2, false, false, ORD_ASC>(AtX_r, Bcm, c1_r);
    mm<1, 5, 2, false, false, ORD_TREE>(AtX_4, Bcm, c1_4);
    mm<1, 2, 2, false, false, ORD_TREE>(c1_r, Si, c2_r);
    mm<1, 2, 2, false, false, ORD_TREE>(c1_4, Si, c2_4);
    mm<1, 2, 5, false, true, ORD_TREE>(c2_r, Bcm, c3_r);
    mm<1, 2, 5, false, true, ORD_TREE>(c2_4, Bcm, c3_4);
    mm<1, 5, 5, false, false, ORD_ASC>(c3_r, Xf, c4_r);
    mm<1, 5, 5, false, false, ORD_TREE>(c3_4, Xf, c4_4);
    mm<1, 5, 5, false, false, ORD_ASC>(c4_r, Acm, P2_r);
    mm<1, 5, 5, false, false, ORD_TREE>(c4_4, Acm, P2_4);
#pragma unroll
    for (int j = 0; j < 5; ++j) { xn_r[j] = (P1_r[j] - P2_r[j]) + Qrow_r[j]; xn_4[j] = (P1_4[j] - P2_4[j]) + Qrow_4[j]; }
  } else {
    float AtX_r[4], P1_r[4], BtX[4], g[1], c1[1], c3_r[4], c4_r[4], P2_r[4];
    mm<1, 4, 4, false, false, ORD_SSE4>(Acol_r, Xf, AtX_r);
    mm<1, 4, 4, false, false, ORD_ASC>(AtX_r, Acm, P1_r);
    mm<1, 4, 4, true, false, ORD_SSE4>(Bcm, Xf, BtX);
    mm<1, 4, 1, false, false, ORD_SSE4>(BtX, Bcm, g);
    const float s = R[0] + g[0];
    mm<1, 4, 1, false, false, ORD_ASC>(AtX_r, Bcm, c1);
    const float c2 = c1[0] / s;
#pragma unroll
    for (int j = 0; j < 4; ++j) c3_r[j] = c2 * Bcm[j];
    mm<1, 4, 4, false, false, ORD_ASC>(c3_r, Xf, c4_r);
    mm<1, 4, 4, false, false, ORD_ASC>(c4_r, Acm, P2_r);
#pragma unroll
    for (int j = 0; j < 4; ++j) xn_r[j] = (P1_r[j] - P2_r[j]) + Qrow_r[j];
    (void)Acol_4; (void)Qrow_4; (void)xn_4;
  }
}

}  // namespace crx

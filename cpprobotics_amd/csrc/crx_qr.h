// crx_qr.h — x = A.colPivHouseholderQr().solve(b) in float for the crx engine (host + gfx950 device).
//
// The reference solves three small float systems through Eigen 3.3.x's ColPivHouseholderQR: the 3x3 of QuinticPolynomial
// (/root/reference/include/quintic_polynomial.h:49), the 2x2 of QuarticPolynomial (quartic_polynomial.h:45) and the nx x nx
// of Spline (cubic_spline.h:56).  Round 1 solved them exactly in double; this is the algorithm Eigen runs, in float, so that
// the planner's polynomial coefficients carry the same rounding as the reference's: column norms, pivoting on the largest
// (down-dated) norm with LAPACK's re-computation rule (lawn176), Householder reflectors stored below the diagonal
// (MatrixBase::makeHouseholder), Q^T applied reflector by reflector (applyHouseholderOnTheLeft), column-oriented back
// substitution on the leading block, column permutation undone (ColPivHouseholderQR.h: computeInPlace, _solve_impl).
// Reductions: the in-loop ones (Householder tail norms, down-date re-computations, essential^T * bottom) are dynamic-size blocks
// in Eigen even for a fixed-size matrix — ascending below one packet, i.e. for the 2x2 and 3x3 systems.  The INITIAL column norms
// of a fixed-size matrix (FIXED = true: Matrix3f of the quintic, Matrix2f of the quartic) are a completely unrolled redux, a tree
// of halves: t0 + (t1 + t2) for three rows (round 6; oracle/eigen_qr.h (1) has the derivation from Core/Redux.h).  The nx x nx
// spline system is a MatrixXf (FIXED = false): ascending here, address-dependent packets in Eigen from 4 rows on (unpinned).
// (An independent statement of the same algorithm, oracle/eigen_qr.h, is what the oracle and the Eigen stand-in
// of the reference build use; the two must agree bit for bit, tests/test_oracle_frenet.py.)
// The library is built -ffp-contract=off with IEEE fp32 division and sqrt: host and device produce the same bits.
#pragma once
#include <float.h>
#include "crx_trig.h"   // CRX_HD

namespace crx {

// A: n x n column-major (A[i + n*j]), overwritten by the factorisation; b overwritten; x receives the solution.  n <= NMAX.
template <int NMAX, bool FIXED = false>
CRX_HD void colpiv_qr_solve(const int n, float* A, float* b, float* x) {
  static_assert(!FIXED || NMAX <= 3, "the fixed-size reduction tree is written out for the reference's 2x2 and 3x3 systems");
  float hc[NMAX], nu[NMAX], nd[NMAX], tmp[NMAX];
  int perm[NMAX];
#define CRX_AT(i, j) A[(i) + n * (j)]
  auto col_norm = [&](int j, int from) -> float {
    if (from >= n) return 0.0f;
    float s = CRX_AT(from, j) * CRX_AT(from, j);
    for (int i = from + 1; i < n; ++i) s = s + CRX_AT(i, j) * CRX_AT(i, j);
    return __builtin_sqrtf(s);
  };
  auto col_norm_fixed = [&](int j) -> float {      // redux_novec_unroller<0, n>: 1: t0 | 2: t0 + t1 | 3: t0 + (t1 + t2)
    const float t0 = CRX_AT(0, j) * CRX_AT(0, j);
    if (n == 1) return __builtin_sqrtf(t0);
    const float t1 = CRX_AT(1, j) * CRX_AT(1, j);
    if (n == 2) return __builtin_sqrtf(t0 + t1);
    const float t2 = CRX_AT(2, j) * CRX_AT(2, j);
    return __builtin_sqrtf(t0 + (t1 + t2));
  };
  for (int k = 0; k < n; ++k) { nd[k] = FIXED ? col_norm_fixed(k) : col_norm(k, 0); nu[k] = nd[k]; perm[k] = k; }
  float maxnorm = nu[0];
  for (int k = 1; k < n; ++k) if (nu[k] > maxnorm) maxnorm = nu[k];
  const float eps = FLT_EPSILON;
  const float thr0 = (maxnorm * eps) * (maxnorm * eps);
  const float threshold_helper = thr0 / (float)n;
  const float downdate_threshold = __builtin_sqrtf(eps);
  int nonzero = n;
  for (int k = 0; k < n; ++k) {
    int big = k;
    for (int j = k + 1; j < n; ++j) if (nu[j] > nu[big]) big = j;
    const float big_sq = nu[big] * nu[big];
    if (nonzero == n && big_sq < threshold_helper * (float)(n - k)) nonzero = k;
    if (k != big) {
      for (int i = 0; i < n; ++i) { const float t = CRX_AT(i, k); CRX_AT(i, k) = CRX_AT(i, big); CRX_AT(i, big) = t; }
      { const float t = nu[k]; nu[k] = nu[big]; nu[big] = t; }
      { const float t = nd[k]; nd[k] = nd[big]; nd[big] = t; }
    }
    { const int t = perm[k]; perm[k] = perm[big]; perm[big] = t; }      // applyTranspositionOnTheRight(k, big), accumulated
    // makeHouseholderInPlace on column k, rows k..n-1
    float beta, tau;
    {
      float tail_sq = 0.0f;
      for (int i = k + 1; i < n; ++i) tail_sq = (i == k + 1) ? CRX_AT(i, k) * CRX_AT(i, k) : tail_sq + CRX_AT(i, k) * CRX_AT(i, k);
      const float c0 = CRX_AT(k, k);
      if (n - k == 1 || tail_sq <= FLT_MIN) {
        tau = 0.0f; beta = c0;
        for (int i = k + 1; i < n; ++i) CRX_AT(i, k) = 0.0f;
      } else {
        beta = __builtin_sqrtf(c0 * c0 + tail_sq);
        if (c0 >= 0.0f) beta = -beta;
        const float den = c0 - beta;
        for (int i = k + 1; i < n; ++i) CRX_AT(i, k) = CRX_AT(i, k) / den;
        tau = (beta - c0) / beta;
      }
    }
    hc[k] = tau;
    CRX_AT(k, k) = beta;
    // applyHouseholderOnTheLeft on rows k.., columns k+1..
    if (k + 1 < n) {
      if (n - k == 1) {
        for (int j = k + 1; j < n; ++j) CRX_AT(k, j) = CRX_AT(k, j) * (1.0f - tau);
      } else if (tau != 0.0f) {
        for (int j = k + 1; j < n; ++j) {
          float s = CRX_AT(k + 1, k) * CRX_AT(k + 1, j);
          for (int i = k + 2; i < n; ++i) s = s + CRX_AT(i, k) * CRX_AT(i, j);
          tmp[j] = s + CRX_AT(k, j);
        }
        for (int j = k + 1; j < n; ++j) CRX_AT(k, j) = CRX_AT(k, j) - tau * tmp[j];
        for (int j = k + 1; j < n; ++j)
          for (int i = k + 1; i < n; ++i) CRX_AT(i, j) = CRX_AT(i, j) - (tau * CRX_AT(i, k)) * tmp[j];
      }
    }
    // norm down-date
    for (int j = k + 1; j < n; ++j) {
      if (nu[j] != 0.0f) {
        float t = __builtin_fabsf(CRX_AT(k, j)) / nu[j];
        t = (1.0f + t) * (1.0f - t);
        t = t < 0.0f ? 0.0f : t;
        const float q = nu[j] / nd[j];
        const float t2 = t * (q * q);
        if (t2 <= downdate_threshold) { nd[j] = col_norm(j, k + 1); nu[j] = nd[j]; }
        else nu[j] = nu[j] * __builtin_sqrtf(t);
      }
    }
  }
  // solve
  if (nonzero == 0) { for (int j = 0; j < n; ++j) x[j] = 0.0f; return; }
  for (int k = 0; k < nonzero; ++k) {              // c = Q^T b
    const float tau = hc[k];
    if (n - k == 1) { b[k] = b[k] * (1.0f - tau); continue; }
    if (tau == 0.0f) continue;
    float s = CRX_AT(k + 1, k) * b[k + 1];
    for (int i = k + 2; i < n; ++i) s = s + CRX_AT(i, k) * b[i];
    const float t = s + b[k];
    b[k] = b[k] - tau * t;
    for (int i = k + 1; i < n; ++i) b[i] = b[i] - (tau * CRX_AT(i, k)) * t;
  }
  for (int i = nonzero - 1; i >= 0; --i) {          // upper-triangular back substitution, column-oriented
    b[i] = b[i] / CRX_AT(i, i);
    for (int j = 0; j < i; ++j) b[j] = b[j] - b[i] * CRX_AT(j, i);
  }
  for (int i = 0; i < nonzero; ++i) x[perm[i]] = b[i];
  for (int i = nonzero; i < n; ++i) x[perm[i]] = 0.0f;
#undef CRX_AT
}

}  // namespace crx

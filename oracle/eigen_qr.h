// eigen_qr.h — TEST INFRASTRUCTURE ONLY (parity oracle).  A float/double restatement of Eigen 3.3.9's
// ColPivHouseholderQR<MatrixType>::computeInPlace() and _solve_impl() (QR/ColPivHouseholderQR.h) with
// MatrixBase::makeHouseholder / applyHouseholderOnTheLeft (Householder/Householder.h) and the upper-triangular back
// substitution, which the reference calls through `A.colPivHouseholderQr().solve(B)`:
//   /root/reference/include/quintic_polynomial.h:49 (3x3), quartic_polynomial.h:45 (2x2), cubic_spline.h:56 (nx x nx).
// Eigen is an un-vendored dependency absent from this image, so this follows the published algorithm: column norms,
// pivoting on the largest updated norm, LAPACK-style norm downdating (lawn176), Householder vectors stored below the
// diagonal, Q^T applied reflector by reflector, back substitution, column permutation.
//
// WHICH REDUCTION EACH CALL SITE USES (re-derived in round 6 from Core/Redux.h, Householder/Householder.h, Core/GeneralProduct.h and
// Core/ProductEvaluators.h of 3.3.9).  PacketSize is 4 floats (SSE, the reference has no -march):
//   (1) computeInPlace, the initial `m_qr.col(k).norm()`.  For a FIXED-size matrix (Matrix3f, Matrix2f: the quintic and quartic
//       systems) col(k) is a fixed-size block, the redux is completely unrolled (redux_traits: Cost <= UnrollingLimit) and, below one
//       packet, it is redux_novec_unroller<0, Size>: func(unroller<0, Size/2>, unroller<Size/2, Size - Size/2>) — a TREE OF HALVES:
//       3 elements: t0 + (t1 + t2); 2 elements: t0 + t1.  Until round 6 this file summed them ascending, (t0 + t1) + t2 — a different
//       float for ~1 in 4 columns, which moves a pivot choice or a down-date decision only in near-ties (`fixed_size` below; the
//       fixed-size branch with >= 4 rows — whole packets by redux_vec_unroller, SSE2 predux (p0 + p2) + (p1 + p3), then the tail's
//       tree — is written out too, although no call site of the reference reaches it).
//       For a DYNAMIC-size matrix (MatrixXf: the spline system) it is redux_impl<LinearVectorizedTraversal, NoUnrolling>: below one
//       packet ascending; from 4 rows on packets start at the first 16-byte-aligned element of the column — the order depends on
//       the ADDRESS of the data (PARITY UNPINNED there: this file sums ascending).
//   (2) in-loop `m_qr.col(j).tail(rows - k - 1).norm()` (the down-date's re-computation) and makeHouseholder's
//       `tail.squaredNorm()`: dynamic-size blocks even of a fixed-size matrix -> the NoUnrolling path; at most 2 elements for the
//       3x3 / 2x2 systems = ascending (what this file does).
//   (3) applyHouseholderOnTheLeft's `essential.adjoint() * bottom`: for the fixed-size systems product_type_selector<1, Small, Small>
//       = CoeffBasedProduct, no packet path (1 row; the right-hand side is column-major), coefficient = a dynamic-size redux of at
//       most 2 products -> ascending.  In _solve_impl the same expression is an InnerProduct (one right-hand-side column): the
//       same dynamic redux.  For MatrixXf both are GemvProduct (Large): Eigen's GEMV kernel, alignment-dependent (UNPINNED).
//   (4) the outer-product update and the back substitution have no reductions (one multiply, one subtract per coefficient).
// tests/tools/eigen_order_probe.cpp runs the fixed-size and the dynamic systems through <Eigen/Eigen> where a box has it.
#pragma once
#include <cmath>
#include <cstddef>
#include <limits>
#include <utility>
#include <vector>

namespace oracle {

template <class S>
class ColPivQR {
 public:
  // fixed_size: the reference's matrix type is a fixed-size Eigen matrix (Matrix3f / Matrix2f), see (1) above
  ColPivQR(int rows, int cols, bool fixed_size = false)
      : rows_(rows), cols_(cols), fixed_(fixed_size), qr_((size_t)rows * cols), hc_(rows < cols ? rows : cols), perm_(cols) {}
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  int nonzero_pivots() const { return nonzero_; }

  void compute(const S* a) {   // a: rows x cols, column-major
    using std::abs;
    using std::sqrt;
    for (size_t i = 0; i < qr_.size(); ++i) qr_[i] = a[i];
    const int rows = rows_, cols = cols_, size = rows < cols ? rows : cols;
    std::vector<int> transp(cols);
    std::vector<S> norm_upd(cols), norm_dir(cols), temp(cols);
    for (int k = 0; k < cols; ++k) { norm_dir[k] = fixed_ ? col_norm_fixed(k) : col_norm(k, 0); norm_upd[k] = norm_dir[k]; }
    S maxnorm = norm_upd[0];
    for (int k = 1; k < cols; ++k) if (norm_upd[k] > maxnorm) maxnorm = norm_upd[k];
    const S eps = std::numeric_limits<S>::epsilon();
    const S thr0 = (maxnorm * eps) * (maxnorm * eps);           // numext::abs2(max * epsilon)
    const S threshold_helper = thr0 / S(rows);
    const S norm_downdate_threshold = sqrt(eps);
    nonzero_ = size;
    for (int k = 0; k < size; ++k) {
      int big = k;                                               // first maximum of the updated norms of columns k..
      for (int j = k + 1; j < cols; ++j) if (norm_upd[j] > norm_upd[big]) big = j;
      const S big_sq = norm_upd[big] * norm_upd[big];
      if (nonzero_ == size && big_sq < threshold_helper * S(rows - k)) nonzero_ = k;
      transp[k] = big;
      if (k != big) {
        for (int i = 0; i < rows; ++i) std::swap(at(i, k), at(i, big));
        std::swap(norm_upd[k], norm_upd[big]);
        std::swap(norm_dir[k], norm_dir[big]);
      }
      // makeHouseholderInPlace on column k, rows k..rows-1
      S beta, tau;
      {
        S tail_sq = S(0);
        for (int i = k + 1; i < rows; ++i) tail_sq = (i == k + 1) ? at(i, k) * at(i, k) : tail_sq + at(i, k) * at(i, k);
        const S c0 = at(k, k);
        const S tol = (std::numeric_limits<S>::min)();
        if (rows - k == 1 || tail_sq <= tol) {
          tau = S(0); beta = c0;
          for (int i = k + 1; i < rows; ++i) at(i, k) = S(0);
        } else {
          beta = sqrt(c0 * c0 + tail_sq);
          if (c0 >= S(0)) beta = -beta;
          const S den = c0 - beta;
          for (int i = k + 1; i < rows; ++i) at(i, k) = at(i, k) / den;
          tau = (beta - c0) / beta;
        }
      }
      hc_[k] = tau;
      at(k, k) = beta;
      // applyHouseholderOnTheLeft to the bottom-right corner (rows k.., cols k+1..) with essential = column k below the diagonal
      apply_left(k, tau, &qr_[0], rows, k + 1, cols, &temp[0]);
      // norm downdate
      for (int j = k + 1; j < cols; ++j) {
        if (norm_upd[j] != S(0)) {
          S t = abs(at(k, j)) / norm_upd[j];
          t = (S(1) + t) * (S(1) - t);
          t = t < S(0) ? S(0) : t;
          const S q = norm_upd[j] / norm_dir[j];
          const S t2 = t * (q * q);
          if (t2 <= norm_downdate_threshold) {
            norm_dir[j] = col_norm(j, k + 1);
            norm_upd[j] = norm_dir[j];
          } else {
            norm_upd[j] *= sqrt(t);
          }
        }
      }
    }
    // permutation: identity, then applyTranspositionOnTheRight(k, transp[k]) for k = 0..size-1
    for (int j = 0; j < cols; ++j) perm_[j] = j;
    for (int k = 0; k < size; ++k) std::swap(perm_[k], perm_[transp[k]]);
  }

  // x (cols) = solution of A x = b (rows); b and x are plain arrays
  void solve(const S* b, S* x) const {
    const int rows = rows_, cols = cols_, nz = nonzero_;
    if (nz == 0) { for (int j = 0; j < cols; ++j) x[j] = S(0); return; }
    std::vector<S> c(b, b + rows);
    S tmp;
    // c.applyOnTheLeft(householderSequence(qr, hCoeffs).setLength(nz).transpose()): reflectors k = 0..nz-1 in turn
    for (int k = 0; k < nz; ++k) apply_left_vec(k, hc_[k], &c[0], &tmp);
    // upper-triangular solve on the leading nz x nz block (back substitution, column-oriented as Eigen's
    // triangular_solver_selector<..., OnTheLeft, Upper, ColMajor> for a single right-hand side: x_i /= U_ii, then the rows above lose x_i U_ji)
    for (int i = nz - 1; i >= 0; --i) {
      c[i] = c[i] / at(i, i);
      for (int j = 0; j < i; ++j) c[j] = c[j] - c[i] * at(j, i);
    }
    for (int i = 0; i < nz; ++i) x[perm_[i]] = c[i];
    for (int i = nz; i < cols; ++i) x[perm_[i]] = S(0);
  }

 private:
  S& at(int i, int j) { return qr_[i + (size_t)rows_ * j]; }
  const S& at(int i, int j) const { return qr_[i + (size_t)rows_ * j]; }
  S col_norm(int j, int from) const {
    using std::sqrt;
    if (from >= rows_) return S(0);
    S s = at(from, j) * at(from, j);
    for (int i = from + 1; i < rows_; ++i) s = s + at(i, j) * at(i, j);
    return sqrt(s);
  }
  // redux_novec_unroller<Start, Length>: the tree of halves
  S tree(const S* t, int start, int len) const {
    if (len == 1) return t[start];
    const int half = len / 2;
    return tree(t, start, half) + tree(t, start + half, len - half);
  }
  // redux_vec_unroller<Start, Length> over packets of 4: lane-wise tree of halves
  void vec_tree(const S* t, int start, int len, S out[4]) const {
    if (len == 1) { for (int l = 0; l < 4; ++l) out[l] = t[4 * start + l]; return; }
    const int half = len / 2;
    S a[4], b[4];
    vec_tree(t, start, half, a); vec_tree(t, start + half, len - half, b);
    for (int l = 0; l < 4; ++l) out[l] = a[l] + b[l];
  }
  // (1): the norm of a whole column of a FIXED-size matrix — redux_impl<LinearVectorizedTraversal, CompleteUnrolling>
  S col_norm_fixed(int j) const {
    using std::sqrt;
    std::vector<S> t(rows_);
    for (int i = 0; i < rows_; ++i) t[i] = at(i, j) * at(i, j);
    const int vec = (rows_ / 4) * 4;
    if (vec == 0) return sqrt(tree(&t[0], 0, rows_));
    S p[4];
    vec_tree(&t[0], 0, rows_ / 4, p);
    S res = (p[0] + p[2]) + (p[1] + p[3]);                      // SSE2 predux<Packet4f>: movehl add, then the shuffled add
    if (vec != rows_) res = res + tree(&t[0], vec, rows_ - vec);
    return sqrt(res);
  }
  // M = qr rows k.., columns c0..c1-1:  tmp = essential^T * bottom ; tmp += row0 ; row0 -= tau*tmp ; bottom -= tau*essential*tmp
  void apply_left(int k, S tau, S* /*base*/, int rows, int c0, int c1, S* tmp) {
    if (c1 <= c0) return;
    if (rows - k == 1) { for (int j = c0; j < c1; ++j) at(k, j) = at(k, j) * (S(1) - tau); return; }
    if (tau == S(0)) return;
    for (int j = c0; j < c1; ++j) {
      S s = at(k + 1, k) * at(k + 1, j);
      for (int i = k + 2; i < rows; ++i) s = s + at(i, k) * at(i, j);
      tmp[j] = s + at(k, j);
    }
    for (int j = c0; j < c1; ++j) at(k, j) = at(k, j) - tau * tmp[j];
    for (int j = c0; j < c1; ++j)
      for (int i = k + 1; i < rows; ++i) at(i, j) = at(i, j) - (tau * at(i, k)) * tmp[j];
  }
  void apply_left_vec(int k, S tau, S* c, S* tmp) const {
    const int rows = rows_;
    if (rows - k == 1) { c[k] = c[k] * (S(1) - tau); return; }
    if (tau == S(0)) return;
    S s = at(k + 1, k) * c[k + 1];
    for (int i = k + 2; i < rows; ++i) s = s + at(i, k) * c[i];
    *tmp = s + c[k];
    c[k] = c[k] - tau * (*tmp);
    for (int i = k + 1; i < rows; ++i) c[i] = c[i] - (tau * at(i, k)) * (*tmp);
  }

  int rows_, cols_;
  bool fixed_;
  int nonzero_ = 0;
  std::vector<S> qr_, hc_;
  std::vector<int> perm_;
};

}  // namespace oracle

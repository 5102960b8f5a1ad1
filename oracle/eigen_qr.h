// eigen_qr.h — TEST INFRASTRUCTURE ONLY (parity oracle).  A float/double restatement of Eigen 3.3.9's
// ColPivHouseholderQR<MatrixType>::computeInPlace() and _solve_impl() (QR/ColPivHouseholderQR.h) with
// MatrixBase::makeHouseholder / applyHouseholderOnTheLeft (Householder/Householder.h) and the upper-triangular back
// substitution, which the reference calls through `A.colPivHouseholderQr().solve(B)`:
//   /root/reference/include/quintic_polynomial.h:49 (3x3), quartic_polynomial.h:45 (2x2), cubic_spline.h:56 (nx x nx).
// Eigen is an un-vendored dependency absent from this image, so this follows the published algorithm: column norms,
// pivoting on the largest updated norm, LAPACK-style norm downdating (lawn176), Householder vectors stored below the
// diagonal, Q^T applied reflector by reflector, back substitution, column permutation.  Every reduction is a plain
// ascending loop — for the 2x2 / 3x3 systems that is what Eigen's dynamic-size block reductions do as well (fewer than one
// packet), for the nx x nx spline system Eigen's GEMV kernels may associate differently (PARITY UNPINNED there; the
// measured effect is in DESIGN.md §5e).
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
  ColPivQR(int rows, int cols) : rows_(rows), cols_(cols), qr_((size_t)rows * cols), hc_(rows < cols ? rows : cols), perm_(cols) {}
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
    for (int k = 0; k < cols; ++k) { norm_dir[k] = col_norm(k, 0); norm_upd[k] = norm_dir[k]; }
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

  int rows_, cols_, nonzero_ = 0;
  std::vector<S> qr_, hc_;
  std::vector<int> perm_;
};

}  // namespace oracle

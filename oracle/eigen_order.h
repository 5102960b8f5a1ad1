// eigen_order.h — TEST INFRASTRUCTURE ONLY (parity oracle).  Not part of the product; nothing
// under cpprobotics_amd/ may include, link or call this.
//
// A tiny dense fixed-size matrix type plus a product routine that restates, operation for
// operation, how Eigen 3.3.x evaluates the small fixed-size float products the reference
// writes (un-vendored dependency: `find_package(Eigen3 REQUIRED)`, /root/reference/CMakeLists.txt:13;
// Debian-11 libeigen3-dev => Eigen 3.3.9 per .devcontainer/Dockerfile).  Eigen itself is absent
// from this image, so this restatement is PARITY-UNPINNED: it follows the published
// Eigen 3.3.9 sources (Core/ProductEvaluators.h, Core/AssignEvaluator.h, Core/Redux.h,
// arch/SSE/PacketMath.h), not an execution of them.
//
// Eigen 3.3.9, x86-64 baseline (SSE2, no FMA: the reference sets no -march,
// CMakeLists.txt:4-6), float, all dimensions < 8  =>  every `A*B` is a coefficient-based
// lazy product evaluated into a temporary; `A*B*C` is `(A*B)*C`.  How one coefficient
// sum_k A(i,k)*B(k,j) is accumulated depends on how the destination is traversed:
//
//  (P) packet path — destination column-major with rows % 4 == 0 and a column-major
//      (non-transposed) left factor: etor_product_packet_impl walks k upward,
//      res = A(:,0)*B(0,j); res = A(:,k)*B(k,j) + res    (pmul, then pmul+padd)
//      => per coefficient: (((t0 + t1) + t2) + t3) ...            "ASC"
//  (C) coefficient path — anything else: coeff(i,j) =
//      (lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum()
//      (C1) if the left factor is a transposed (row-major) matrix, the right one column-major
//           and the inner size is a multiple of 4, that .sum() is vectorised: one pmul of the
//           two packets, then SSE2 predux: (t0 + t2) + (t1 + t3)     "SSE4"
//           (inner size 8 would add the two product packets first; not needed here)
//      (C2) otherwise the redux is fully unrolled by redux_novec_unroller, which splits the
//           range in halves recursively: sum(s,len) = sum(s,len/2) + sum(s+len/2,len-len/2)
//           e.g. 5 terms: (t0 + t1) + (t2 + (t3 + t4))                "TREE"
//
// For the reference's call sites every sum that falls on path (C) has at most two non-zero
// terms (the factors involved are the 0/1 selection matrix jH, or A/B with one or two
// non-zeros per row/column), so ASC, SSE4 and TREE give identical results there — the tests
// check that claim by running the oracle in all-ASC mode as well.
#pragma once
#include <cstring>

namespace oracle {

enum SumOrder { ORDER_EIGEN = 0, ORDER_ASC = 1 };

template <int R, int C>
struct Mat {
  float d[R * C];  // column-major
  float& operator()(int i, int j) { return d[i + R * j]; }
  float operator()(int i, int j) const { return d[i + R * j]; }
  static Mat zero() { Mat m; std::memset(m.d, 0, sizeof(m.d)); return m; }
  static Mat identity() { Mat m = zero(); for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0f; return m; }
};

template <int R, int C>
Mat<C, R> transpose(const Mat<R, C>& a) {
  Mat<C, R> t;
  for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) t(j, i) = a(i, j);
  return t;
}

inline float tree_sum(const float* t, int start, int len) {
  if (len == 1) return t[start];
  int half = len / 2;
  float a = tree_sum(t, start, half);
  float b = tree_sum(t, start + half, len - half);
  return a + b;
}

// lhs_transposed / rhs_transposed: whether the factor, AS WRITTEN IN THE REFERENCE
// EXPRESSION, is a `.transpose()` view of a stored (column-major) matrix.  A and B here
// are already the logical (R x K) and (K x C) factors.
template <int R, int K, int C>
Mat<R, C> mul(const Mat<R, K>& A, const Mat<K, C>& B, bool lhs_transposed, bool rhs_transposed,
              SumOrder order) {
  Mat<R, C> out;
  // column-major packet path (CanVectorizeLhs), or — both factors transposed views — the
  // row-major packet path (CanVectorizeRhs with EvalToRowMajor); both accumulate k upward.
  const bool packet_path = ((R % 4 == 0) && !lhs_transposed) ||
                           (lhs_transposed && rhs_transposed && (C % 4 == 0) && (C != 1));
  const bool sse_inner = lhs_transposed && !rhs_transposed && (K % 4 == 0);
  for (int j = 0; j < C; ++j)
    for (int i = 0; i < R; ++i) {
      float t[K];
      for (int k = 0; k < K; ++k) t[k] = A(i, k) * B(k, j);
      float s;
      if (order == ORDER_ASC || packet_path) {
        s = t[0];
        for (int k = 1; k < K; ++k) s = s + t[k];
      } else if (sse_inner && K == 4) {
        s = (t[0] + t[2]) + (t[1] + t[3]);
      } else {
        s = tree_sum(t, 0, K);
      }
      out(i, j) = s;
    }
  return out;
}

template <int R, int C>
Mat<R, C> add(const Mat<R, C>& a, const Mat<R, C>& b) {
  Mat<R, C> o; for (int i = 0; i < R * C; ++i) o.d[i] = a.d[i] + b.d[i]; return o;
}
template <int R, int C>
Mat<R, C> sub(const Mat<R, C>& a, const Mat<R, C>& b) {
  Mat<R, C> o; for (int i = 0; i < R * C; ++i) o.d[i] = a.d[i] - b.d[i]; return o;
}

// Eigen's closed-form 2x2 inverse (LU/InverseImpl.h compute_inverse_size2_helper):
// invdet = 1 / (m00*m11 - m10*m01); result = [m11, -m01; -m10, m00] * invdet.
inline Mat<2, 2> inverse2(const Mat<2, 2>& m) {
  float det = m(0, 0) * m(1, 1) - m(1, 0) * m(0, 1);
  float invdet = 1.0f / det;
  Mat<2, 2> r;
  r(0, 0) = m(1, 1) * invdet;
  r(1, 0) = -m(1, 0) * invdet;
  r(0, 1) = -m(0, 1) * invdet;
  r(1, 1) = m(0, 0) * invdet;
  return r;
}

}  // namespace oracle

// eigen_order.h — TEST INFRASTRUCTURE ONLY (parity oracle).  Not part of the product; nothing
// under cpprobotics_amd/ may include, link or call this.
//
// A tiny dense fixed-size matrix type plus a product routine that restates, operation for
// operation, how Eigen 3.3.x evaluates the small fixed-size float products the reference
// writes (un-vendored dependency: `find_package(Eigen3 REQUIRED)`, /root/reference/CMakeLists.txt:13;
// Debian-11 libeigen3-dev => Eigen 3.3.9 per .devcontainer/Dockerfile).  Eigen itself is absent
// from this image, so this restatement is UNPINNED against Eigen's binary (the one thing the reference-line build of
// oracle/ref_build.sh cannot check on a host without Eigen): it follows the published
// Eigen 3.3.9 sources (Core/ProductEvaluators.h, Core/AssignEvaluator.h, Core/Redux.h,
// arch/SSE/PacketMath.h), not an execution of them.
//
// Eigen 3.3.9, x86-64 baseline (SSE2, no FMA: the reference sets no -march,
// CMakeLists.txt:4-6; packet = 4 floats), float, all dimensions < 8  =>  every `A*B` is a
// coefficient-based lazy product evaluated into a plain temporary (column-major, except that a
// 1xN row vector is row-major); `A*B*C` is `(A*B)*C`.  How one coefficient sum_k A(i,k)*B(k,j)
// is accumulated is decided by product_evaluator's enums and the assignment traversal
// (product_order() below states the decision generically; r, c, k = rows, cols, inner size;
// a factor is "row-major" if it is a .transpose() of a column-major object or a plain row vector):
//
//   CanVectorizeLhs = lhs column-major && r != 1            CanVectorizeRhs = rhs row-major && c != 1
//     (ProductEvaluators.h: `(!LhsRowMajor) && (LhsFlags & PacketAccessBit) && (RowsAtCompileTime!=1)` — there is no `% 4`
//      term: with EIGEN_UNALIGNED_VECTORIZE = 1, the SSE default, every float matrix or transposed view has PacketAccessBit;
//      rounds 1-2 of this file wrongly required r % 4 == 0, corrected in round 3)
//   EvalToRowMajor  = (r == 1 && c != 1) ? 1 : (c == 1 && r != 1) ? 0 : (rhs row-major && !CanVectorizeLhs)
//   The assignment into the plain temporary (column-major unless it is a row vector) is chosen by
//   copy_using_evaluator_traits (AssignEvaluator.h).  With `inner` = the temporary's size along its storage order
//   (the whole size for a vector), InnerPacketSize = LinearPacketSize = 4 (find_best_packet<float, n> falls back to Packet4f
//   for every n), and MightVectorize = the product has PacketAccessBit (either CanVectorize*) && storage orders agree:
//     inner % 4 == 0                   -> InnerVectorizedTraversal: every coefficient by packets           (4x4 products)
//     else, a vector                   -> LinearVectorizedTraversal, CompleteUnrolling: the first (inner/4)*4 coefficients
//                                         by packets, the rest by coeff()
//     else, inner >= 4                 -> SliceVectorizedTraversal (MaySliceVectorize: InnerMaxSize >= InnerPacketSize under
//                                         EIGEN_UNALIGNED_VECTORIZE) with InnerUnrolling (inner * (1 + CoeffReadCost) <= 400
//                                         for every size below 8): per column the first (inner/4)*4 rows by packets
//                                         (copy_using_evaluator_innervec_InnerUnrolling), the rest by coeff()
//                                         (copy_using_evaluator_DefaultTraversal_InnerUnrolling)        (5-row products)
//     else                             -> DefaultTraversal: every coefficient by coeff()                 (2- and 3-row products)
//   So for a 5x5 temporary with a column-major left factor, rows 0-3 of every column are packet sums and row 4 is a
//   coeff() redux.  On the reference's own matrices (A, B with at most two non-zeros per row / column) all orders coincide;
//   the distinction matters for dense inputs to crx_dare_batch only.
//
//  (P) packet assignment: etor_product_packet_impl walks k upward,
//      res = A(:,0)*B(0,j); res = A(:,k)*B(k,j) + res    (pmul, then pmul+padd)
//      => per coefficient: (((t0 + t1) + t2) + t3) ...            "ASC"
//  (C) coefficient path — anything else: coeff(i,j) =
//      (lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum()
//      (C1) the redux is vectorised (LinearVectorizedTraversal, complete unrolling) when the row of lhs AND the
//           column of rhs are both contiguous (Block's MaskPacketAccessBit: lhs row-major, rhs column-major) and
//           k >= 4: the k/4 product packets are added pairwise by redux_vec_unroller (halves), the result is reduced
//           by the SSE2 predux (p0 + p2) + (p1 + p3), and the k % 4 remaining terms — summed by
//           redux_novec_unroller — are added last.  k = 4: (t0 + t2) + (t1 + t3); k = 5: ((t0+t2)+(t1+t3)) + t4.   "VEC"
//           (Round 1 applied this only for k = 4 and used (C2) for k = 5; re-derived from
//           redux_impl<Func, Derived, LinearVectorizedTraversal, CompleteUnrolling> in round 2 — matters only for
//           dense 5x5 inputs through A'*X and B'*X, never on the reference's own sparse A, B.)
//      (C2) otherwise the redux is fully unrolled by redux_novec_unroller, which splits the
//           range in halves recursively: sum(s,len) = sum(s,len/2) + sum(s+len/2,len-len/2)
//           e.g. 5 terms: (t0 + t1) + (t2 + (t3 + t4))                "TREE"
//
// For the reference's call sites every sum that falls on path (C) has at most two non-zero
// terms (the factors involved are the 0/1 selection matrix jH, or A/B with one or two
// non-zeros per row/column), so ASC, VEC and TREE give identical results there — the tests
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

enum ProductOrder { PO_ASC = 0, PO_VEC = 1, PO_TREE = 2 };

// The decision described in the header, for fixed sizes: how coefficient (i, j) of an r x c product of inner size k is summed.
inline ProductOrder product_order(int r, int c, int k, bool lhs_rm, bool rhs_rm, int i, int j) {
  const bool can_vec_lhs = !lhs_rm && r != 1;
  const bool can_vec_rhs = rhs_rm && c != 1;
  const bool eval_rm = (r == 1 && c != 1) ? true : (c == 1 && r != 1) ? false : (rhs_rm && !can_vec_lhs);
  const bool dst_rm = (r == 1 && c != 1);
  const bool is_vector = (r == 1 || c == 1);
  const int inner = is_vector ? r * c : (dst_rm ? c : r);
  const int pos = dst_rm ? j : i;                                         // position along the temporary's storage order
  int by_packets = 0;                                                     // leading positions evaluated by packets
  if ((can_vec_lhs || can_vec_rhs) && eval_rm == dst_rm) {
    if (inner % 4 == 0) by_packets = inner;                               // InnerVectorizedTraversal
    else if (is_vector || inner >= 4) by_packets = (inner / 4) * 4;       // Linear- / SliceVectorizedTraversal, unrolled
  }
  if (pos < by_packets) return PO_ASC;
  // Coefficient path.  ORACLE_CPATH_ORDER (a build flag of the test infrastructure only) forces one order for EVERY sum of this
  // path — 1: ascending, 2: the unrolled tree — so that tests can show that on the reference's own call sites the choice
  // made here does not matter (every such sum has at most two non-zero terms): tests/test_oracle_vs_ref.py.
#if defined(ORACLE_CPATH_ORDER) && ORACLE_CPATH_ORDER == 1
  return PO_ASC;
#elif defined(ORACLE_CPATH_ORDER) && ORACLE_CPATH_ORDER == 2
  return PO_TREE;
#endif
  if (lhs_rm && !rhs_rm && k >= 4) return PO_VEC;
  return PO_TREE;
}

template <class S>
inline S tree_sum(const S* t, int start, int len) {
  if (len == 1) return t[start];
  int half = len / 2;
  S a = tree_sum(t, start, half);
  S b = tree_sum(t, start + half, len - half);
  return a + b;
}
template <class S>
inline void packet_tree(const S* t, int start, int npk, S out[4]) {   // redux_vec_unroller
  if (npk == 1) { for (int l = 0; l < 4; ++l) out[l] = t[4 * start + l]; return; }
  const int half = npk / 2;
  S a[4], b[4];
  packet_tree(t, start, half, a);
  packet_tree(t, start + half, npk - half, b);
  for (int l = 0; l < 4; ++l) out[l] = a[l] + b[l];
}
template <class S>
inline S accumulate(const S* t, int k, ProductOrder o) {
  if (k == 0) return S(0);
  if (o == PO_ASC) { S s = t[0]; for (int i = 1; i < k; ++i) s = s + t[i]; return s; }
  if (o == PO_VEC) {
    const int npk = k / 4;
    S p[4];
    packet_tree(t, 0, npk, p);
    S res = (p[0] + p[2]) + (p[1] + p[3]);                       // SSE2 predux
    if (npk * 4 != k) res = res + tree_sum(t, npk * 4, k - npk * 4);
    return res;
  }
  return tree_sum(t, 0, k);
}

// lhs_transposed / rhs_transposed: whether the factor, AS WRITTEN IN THE REFERENCE
// EXPRESSION, is row-major: a `.transpose()` view of a stored (column-major) matrix, or a
// row-vector temporary.  A and B here are already the logical (R x K) and (K x C) factors.
template <int R, int K, int C>
Mat<R, C> mul(const Mat<R, K>& A, const Mat<K, C>& B, bool lhs_transposed, bool rhs_transposed,
              SumOrder order) {
  Mat<R, C> out;
  for (int j = 0; j < C; ++j)
    for (int i = 0; i < R; ++i) {
      float t[K];
      for (int k = 0; k < K; ++k) t[k] = A(i, k) * B(k, j);
      out(i, j) = accumulate(t, K, (order == ORDER_ASC) ? PO_ASC : product_order(R, C, K, lhs_transposed, rhs_transposed, i, j));
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

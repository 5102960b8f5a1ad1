// lqr_ref.cpp — TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline for bench.py).
// Nothing under cpprobotics_amd/ may include, link or call this file.
//
// CPU restatement of the reference's discrete Riccati fixed point and LQR gain:
//   5x5 / 2 inputs: /root/reference/src/lqr_speed_steer_control.cpp:85-100 (solve_DARE),
//                   :102-106 (dlqr), :116-129 (A, B, Q, R as lqr_steering_control builds them)
//   4x4 / 1 input : /root/reference/src/lqr_steer_control.cpp:75-90, :92-96, :104-115
// written as the dense expressions the reference writes, in Eigen 3.3.9's evaluation order
// (oracle/eigen_order.h).  
// PINNED against the reference's own lines (oracle/ref_build.sh compiles them unmodified — against the host's Eigen, or against the
// Eigen stand-in oracle/ref_shim/Eigen/Eigen where there is none — and tests/test_oracle_vs_ref.py demands equal bits); unpinned only
// with respect to Eigen's own binary, absent from every host of this project.
// Also cross-checked against a numpy-float32 twin, a float64 evaluation and
// scipy.linalg.solve_discrete_are (tests/test_oracle_lqr.py).
#include <cmath>
#include <cstring>
#include "eigen_order.h"

using namespace oracle;

namespace {

template <int R, int C> Mat<R, C> load(const float* p) { Mat<R, C> m; std::memcpy(m.d, p, sizeof(m.d)); return m; }
template <int R, int C> void store(float* p, const Mat<R, C>& m) { std::memcpy(p, m.d, sizeof(m.d)); }

template <int N>
float max_abs_diff(const Mat<N, N>& a, const Mat<N, N>& b) {  // (Xn - X).cwiseAbs().maxCoeff()
  float m = std::fabs(a.d[0] - b.d[0]);
  for (int i = 1; i < N * N; ++i) { float e = std::fabs(a.d[i] - b.d[i]); if (e > m) m = e; }
  return m;
}

// ---- 5x5, B 5x2, R 2x2 : src/lqr_speed_steer_control.cpp -------------------------------
Mat<5, 5> dare5_iter(const Mat<5, 5>& A, const Mat<5, 2>& B, const Mat<5, 5>& Q, const Mat<2, 2>& R,
                     const Mat<5, 5>& X, SumOrder o) {
  // :91  Xn = A'*X*A - A'*X*B*(R + B'*X*B).inverse()*B'*X*A + Q
  Mat<5, 5> At = transpose(A);
  Mat<2, 5> Bt = transpose(B);
  Mat<5, 5> AtX = mul(At, X, true, false, o);
  Mat<5, 5> P1 = mul(AtX, A, false, false, o);
  Mat<2, 2> G = mul(mul(Bt, X, true, false, o), B, false, false, o);
  Mat<2, 2> Si = inverse2(add(R, G));
  Mat<5, 2> c1 = mul(AtX, B, false, false, o);          // (A'X)*B      (A'*X is re-evaluated by Eigen; same value)
  Mat<5, 2> c2 = mul(c1, Si, false, false, o);          // *(...)^-1
  Mat<5, 5> c3 = mul(c2, Bt, false, true, o);           // *B'
  Mat<5, 5> c4 = mul(c3, X, false, false, o);           // *X
  Mat<5, 5> P2 = mul(c4, A, false, false, o);           // *A
  return add(sub(P1, P2), Q);
}

int solve_dare5(const Mat<5, 5>& A, const Mat<5, 2>& B, const Mat<5, 5>& Q, const Mat<2, 2>& R,
                float eps, int maxiter, Mat<5, 5>& Xout, SumOrder o) {
  Mat<5, 5> X = Q;                                       // :86
  for (int i = 0; i < maxiter; ++i) {                    // :90
    Mat<5, 5> Xn = dare5_iter(A, B, Q, R, X, o);
    if (max_abs_diff(Xn, X) < eps) { Xout = Xn; return i + 1; }   // :92-95
    X = Xn;                                              // :96
  }
  Xout = X;                                              // :99
  return maxiter;
}

Mat<2, 5> dlqr5_gain(const Mat<5, 5>& A, const Mat<5, 2>& B, const Mat<2, 2>& R, const Mat<5, 5>& X, SumOrder o) {
  // :104  K = (B'*X*B + R).inverse() * (B'*X*A)
  Mat<2, 5> Bt = transpose(B);
  Mat<2, 5> BtX = mul(Bt, X, true, false, o);
  Mat<2, 2> Si = inverse2(add(mul(BtX, B, false, false, o), R));
  Mat<2, 5> BtXA = mul(BtX, A, false, false, o);
  return mul(Si, BtXA, false, false, o);
}

// ---- 4x4, B 4x1, scalar R : src/lqr_steer_control.cpp -------------------------------------
Mat<4, 4> dare4_iter(const Mat<4, 4>& A, const Mat<4, 1>& B, const Mat<4, 4>& Q, float R,
                     const Mat<4, 4>& X, SumOrder o) {
  // :81  Xn = A'*X*A - A'*X*B/(R + B'*X*B) * B'*X*A + Q
  Mat<4, 4> At = transpose(A);
  Mat<1, 4> Bt = transpose(B);
  Mat<4, 4> AtX = mul(At, X, true, false, o);
  Mat<4, 4> P1 = mul(AtX, A, false, false, o);
  Mat<1, 4> BtX = mul(Bt, X, true, false, o);
  float g = mul(BtX, B, true /*row-vector temporary is row-major*/, false, o)(0, 0);
  float s = R + g;
  Mat<4, 1> c1 = mul(AtX, B, false, false, o);
  Mat<4, 1> c2; for (int i = 0; i < 4; ++i) c2.d[i] = c1.d[i] / s;   // true division
  Mat<4, 4> c3 = mul(c2, Bt, false, true, o);
  Mat<4, 4> c4 = mul(c3, X, false, false, o);
  Mat<4, 4> P2 = mul(c4, A, false, false, o);
  return add(sub(P1, P2), Q);
}

int solve_dare4(const Mat<4, 4>& A, const Mat<4, 1>& B, const Mat<4, 4>& Q, float R, float eps,
                int maxiter, Mat<4, 4>& Xout, SumOrder o) {
  Mat<4, 4> X = Q;
  for (int i = 0; i < maxiter; ++i) {
    Mat<4, 4> Xn = dare4_iter(A, B, Q, R, X, o);
    if (max_abs_diff(Xn, X) < eps) { Xout = Xn; return i + 1; }
    X = Xn;
  }
  Xout = X;
  return maxiter;
}

Mat<1, 4> dlqr4_gain(const Mat<4, 4>& A, const Mat<4, 1>& B, float R, const Mat<4, 4>& X, SumOrder o) {
  // :94  K = 1.0/(B'*X*B + R) * (B'*X*A)      (1.0/float -> double, converted to float by Eigen's scalar*matrix)
  Mat<1, 4> Bt = transpose(B);
  Mat<1, 4> BtX = mul(Bt, X, true, false, o);
  float g = mul(BtX, B, true, false, o)(0, 0);
  float inv = (float)(1.0 / (double)(g + R));
  Mat<1, 4> BtXA = mul(BtX, A, true, false, o);
  Mat<1, 4> K; for (int j = 0; j < 4; ++j) K.d[j] = inv * BtXA.d[j];
  return K;
}

}  // namespace

extern "C" {

// dim 5: A n x25, B n x10, Q n x25, R n x4 -> X n x25, K n x10 (2x5 col-major), iters n
// dim 4: A n x16, B n x4,  Q n x16, R n x1 -> X n x16, K n x4,                  iters n
// Agents [a0,a1).  sum_order 0 = Eigen's, 1 = all-ascending.
int oracle_dare(int n, int dim, const float* A, const float* B, const float* Q, const float* R,
                float eps, int maxiter, float* X, float* K, int* iters, int sum_order, int a0, int a1) {
  SumOrder o = (SumOrder)sum_order;
  if (dim == 5) {
    for (int k = a0; k < a1; ++k) {
      Mat<5, 5> Am = load<5, 5>(A + 25 * k), Qm = load<5, 5>(Q + 25 * k), Xm;
      Mat<5, 2> Bm = load<5, 2>(B + 10 * k);
      Mat<2, 2> Rm = load<2, 2>(R + 4 * k);
      int it = solve_dare5(Am, Bm, Qm, Rm, eps, maxiter, Xm, o);
      if (X) store(X + 25 * k, Xm);
      if (K) store(K + 10 * k, dlqr5_gain(Am, Bm, Rm, Xm, o));
      if (iters) iters[k] = it;
    }
    return 0;
  } else if (dim == 4) {
    for (int k = a0; k < a1; ++k) {
      Mat<4, 4> Am = load<4, 4>(A + 16 * k), Qm = load<4, 4>(Q + 16 * k), Xm;
      Mat<4, 1> Bm = load<4, 1>(B + 4 * k);
      int it = solve_dare4(Am, Bm, Qm, R[k], eps, maxiter, Xm, o);
      if (X) store(X + 16 * k, Xm);
      if (K) store(K + 4 * k, dlqr4_gain(Am, Bm, R[k], Xm, o));
      if (iters) iters[k] = it;
    }
    return 0;
  }
  return -1;
}

// A, B, Q, R exactly as lqr_steering_control() fills them from state.v:
// dim 5: src/lqr_speed_steer_control.cpp:116-129; dim 4: src/lqr_steer_control.cpp:104-115.
void oracle_lqr_build(int n, int dim, const float* v, double DT, double L, float* A, float* B, float* Q, float* R) {
  for (int k = 0; k < n; ++k) {
    if (dim == 5) {
      Mat<5, 5> Am = Mat<5, 5>::zero();
      Am(0, 0) = 1.0f; Am(0, 1) = (float)DT; Am(1, 2) = v[k]; Am(2, 2) = 1.0f; Am(2, 3) = (float)DT; Am(4, 4) = 1.0f;
      Mat<5, 2> Bm = Mat<5, 2>::zero();
      Bm(3, 0) = (float)((double)v[k] / L);   // state.v/L : float / double
      Bm(4, 1) = (float)DT;
      store(A + 25 * k, Am); store(B + 10 * k, Bm);
      store(Q + 25 * k, Mat<5, 5>::identity()); store(R + 4 * k, Mat<2, 2>::identity());
    } else {
      Mat<4, 4> Am = Mat<4, 4>::zero();
      Am(0, 0) = 1.0f; Am(0, 1) = (float)DT; Am(1, 2) = v[k]; Am(2, 2) = 1.0f; Am(2, 3) = (float)DT;
      Mat<4, 1> Bm = Mat<4, 1>::zero();
      Bm(3, 0) = (float)((double)v[k] / L);
      store(A + 16 * k, Am); store(B + 4 * k, Bm);
      store(Q + 16 * k, Mat<4, 4>::identity()); R[k] = 1.0f;
    }
  }
}

}  // extern "C"

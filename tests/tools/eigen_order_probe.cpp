// eigen_order_probe.cpp — TEST INFRASTRUCTURE.  Pins oracle/eigen_order.h (the restatement of how Eigen 3.3.x accumulates the small
// fixed-size float products of the reference's hot path) against an EXECUTED Eigen: every product shape and every product chain the hot
// path forms, on random DENSE operands (the reference's own A, B, jH are sparse enough to make the order irrelevant; crx_dare_batch on
// general matrices is where the order decides bits), evaluated by <Eigen/Eigen> and by oracle::mul, compared with memcmp.
//
//   g++ -std=c++17 -O2 -I<dir holding Eigen/Eigen> -I<repo> tests/tools/eigen_order_probe.cpp -o probe && ./probe [seeds]
//
// No -march / -mfma: the reference's build sets none (/root/reference/CMakeLists.txt:4-6), and with FMA Eigen's pmadd would fuse.
// Needs neither /root/reference nor a GPU — only Eigen (find_package(Eigen3 REQUIRED), /root/reference/CMakeLists.txt:13); run by
// tests/test_eigen_probe.py on whichever box has it, and always against the stand-in oracle/ref_shim/Eigen/Eigen (which derives the
// order generically from the operand types: that run checks the per-call-site flags below against the generic rule).
// Call sites: /root/reference/src/extended_kalman_filter.cpp:35,69,74-77; src/lqr_speed_steer_control.cpp:91,104;
// src/lqr_steer_control.cpp:81,94.
// Round 6: the `colPivHouseholderQr().solve()` call sites of the planners — Matrix3f
// (/root/reference/include/quintic_polynomial.h:49), Matrix2f (quartic_polynomial.h:45), MatrixXf n x n, n = 4 .. 12 (cubic_spline.h:56)
// — on random systems and on the reference's own matrices (the quintic / quartic time matrices on the planner's horizon grid, the
// tridiagonal spline system), through <Eigen/Eigen> and through oracle::ColPivQR (oracle/eigen_qr.h), memcmp.  The fixed-size
// systems are the ones eigen_qr.h claims to pin; the dynamic ones are reported separately (their reductions are address-dependent
// in Eigen: eigen_qr.h says UNPINNED, and a mismatch count there is information, not a failure of the claim).
#include <Eigen/Eigen>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#include "oracle/eigen_order.h"
#include "oracle/eigen_qr.h"

namespace {

uint64_t g_state = 1;
float rnd() {   // xorshift64*, mapped to [-2, 2): magnitudes that make every partial sum round
  g_state ^= g_state >> 12; g_state ^= g_state << 25; g_state ^= g_state >> 27;
  return (float)((double)((g_state * 2685821657736338717ULL) >> 40) / (double)(1 << 24) * 4.0 - 2.0);
}
template <int R, int C> void fill(Eigen::Matrix<float, R, C>& e, oracle::Mat<R, C>& o) {
  for (int j = 0; j < C; ++j) for (int i = 0; i < R; ++i) { const float v = rnd(); e(i, j) = v; o(i, j) = v; }
}
int g_fail = 0, g_cases = 0;
template <int R, int C> void check(const char* what, const Eigen::Matrix<float, R, C>& e, const oracle::Mat<R, C>& o) {
  ++g_cases;
  for (int j = 0; j < C; ++j)
    for (int i = 0; i < R; ++i) {
      const float a = e(i, j), b = o(i, j);
      if (std::memcmp(&a, &b, 4) != 0) {
        if (g_fail < 40) std::printf("MISMATCH %-46s (%d,%d): Eigen %a  eigen_order.h %a\n", what, i, j, a, b);
        ++g_fail;
        return;
      }
    }
}
using oracle::mul; using oracle::transpose; using oracle::add; using oracle::sub; using oracle::inverse2;
const oracle::SumOrder E = oracle::ORDER_EIGEN;
template <int R, int C> using EM = Eigen::Matrix<float, R, C>;
template <int R, int C> using OM = oracle::Mat<R, C>;

void one_seed() {
  // ---- single products: every (rows, inner, cols, storage order of the factors) the hot path forms ------------------------------
  { EM<4, 4> a, b; OM<4, 4> oa, ob; fill(a, oa); fill(b, ob);
    check<4, 4>("4x4 * 4x4", a * b, mul(oa, ob, false, false, E));
    check<4, 4>("4x4 * 4x4^T", a * b.transpose(), mul(oa, transpose(ob), false, true, E));
    check<4, 4>("4x4^T * 4x4", a.transpose() * b, mul(transpose(oa), ob, true, false, E)); }
  { EM<2, 4> h; OM<2, 4> oh; EM<4, 4> p; OM<4, 4> op; fill(h, oh); fill(p, op);
    check<2, 4>("2x4 * 4x4", h * p, mul(oh, op, false, false, E));
    check<4, 2>("4x4 * 2x4^T", p * h.transpose(), mul(op, transpose(oh), false, true, E));
    EM<2, 4> hp = h * p; OM<2, 4> ohp = mul(oh, op, false, false, E);
    check<2, 2>("(2x4) * 2x4^T", hp * h.transpose(), mul(ohp, transpose(oh), false, true, E)); }
  { EM<4, 2> k; OM<4, 2> ok; EM<2, 2> s; OM<2, 2> os; EM<2, 1> y; OM<2, 1> oy; EM<2, 4> h; OM<2, 4> oh;
    fill(k, ok); fill(s, os); fill(y, oy); fill(h, oh);
    check<4, 2>("4x2 * 2x2", k * s, mul(ok, os, false, false, E));
    check<4, 1>("4x2 * 2x1", k * y, mul(ok, oy, false, false, E));
    check<4, 4>("4x2 * 2x4", k * h, mul(ok, oh, false, false, E)); }
  { EM<4, 4> f; OM<4, 4> of; EM<4, 1> x; OM<4, 1> ox; EM<4, 2> b; OM<4, 2> ob; EM<2, 1> u; OM<2, 1> ou;
    fill(f, of); fill(x, ox); fill(b, ob); fill(u, ou);
    check<4, 1>("4x4 * 4x1", f * x, mul(of, ox, false, false, E));
    check<4, 1>("4x2 * 2x1 (B*u)", b * u, mul(ob, ou, false, false, E)); }
  { EM<5, 5> a, x; OM<5, 5> oa, ox; EM<5, 2> b; OM<5, 2> ob; EM<2, 2> s; OM<2, 2> os; fill(a, oa); fill(x, ox); fill(b, ob); fill(s, os);
    check<5, 5>("5x5^T * 5x5", a.transpose() * x, mul(transpose(oa), ox, true, false, E));
    check<5, 5>("5x5 * 5x5", a * x, mul(oa, ox, false, false, E));
    check<5, 2>("5x5 * 5x2", a * b, mul(oa, ob, false, false, E));
    check<2, 5>("5x2^T * 5x5", b.transpose() * x, mul(transpose(ob), ox, true, false, E));
    EM<2, 5> btx = b.transpose() * x; OM<2, 5> obtx = mul(transpose(ob), ox, true, false, E);
    check<2, 2>("(2x5) * 5x2", btx * b, mul(obtx, ob, false, false, E));
    check<2, 5>("(2x5) * 5x5", btx * a, mul(obtx, oa, false, false, E));
    check<5, 2>("5x2 * 2x2", b * s, mul(ob, os, false, false, E));
    check<5, 5>("5x2 * 5x2^T", b * b.transpose(), mul(ob, transpose(ob), false, true, E));
    EM<5, 1> v; OM<5, 1> ov; fill(v, ov);
    check<2, 1>("(2x5) * 5x1", btx * v, mul(obtx, ov, false, false, E));
    check<2, 5>("2x2 * (2x5)", s * btx, mul(os, obtx, false, false, E)); }
  { EM<4, 4> a, x; OM<4, 4> oa, ox; EM<4, 1> b; OM<4, 1> ob; fill(a, oa); fill(x, ox); fill(b, ob);
    EM<1, 4> btx = b.transpose() * x; OM<1, 4> obtx = mul(transpose(ob), ox, true, false, E);
    check<1, 4>("4x1^T * 4x4", btx, obtx);
    check<1, 1>("(1x4) * 4x1", EM<1, 1>(btx * b), mul(obtx, ob, true, false, E));
    check<1, 4>("(1x4) * 4x4", btx * a, mul(obtx, oa, true, false, E));
    check<4, 4>("4x1 * 4x1^T", b * b.transpose(), mul(ob, transpose(ob), false, true, E)); }
  { EM<2, 2> s; OM<2, 2> os; fill(s, os); s(0, 0) += 3.0f; os(0, 0) += 3.0f; s(1, 1) += 3.0f; os(1, 1) += 3.0f;
    check<2, 2>("Matrix2f::inverse()", EM<2, 2>(s.inverse()), inverse2(os)); }
  // ---- the chains as the reference writes them --------------------------------------------------------------------------------------
  { // extended_kalman_filter.cpp:69,74-77
    EM<4, 4> jF, P, Q; OM<4, 4> ojF, oP, oQ; EM<2, 4> jH; OM<2, 4> ojH; EM<2, 2> R; OM<2, 2> oR; EM<2, 1> y; OM<2, 1> oy; EM<4, 1> xp; OM<4, 1> oxp;
    fill(jF, ojF); fill(P, oP); fill(Q, oQ); fill(jH, ojH); fill(R, oR); fill(y, oy); fill(xp, oxp);
    R(0, 0) += 6.0f; oR(0, 0) += 6.0f; R(1, 1) += 6.0f; oR(1, 1) += 6.0f;
    EM<4, 4> PP = jF * P * jF.transpose() + Q;
    OM<4, 4> oPP = add(mul(mul(ojF, oP, false, false, E), transpose(ojF), false, true, E), oQ);
    check<4, 4>("jF*P*jF^T + Q  (:69)", PP, oPP);
    EM<2, 2> S = jH * PP * jH.transpose() + R;
    OM<2, 2> oS = add(mul(mul(ojH, oPP, false, false, E), transpose(ojH), false, true, E), oR);
    check<2, 2>("jH*PP*jH^T + R  (:74)", S, oS);
    EM<4, 2> K = PP * jH.transpose() * S.inverse();
    OM<4, 2> oK = mul(mul(oPP, transpose(ojH), false, true, E), inverse2(oS), false, false, E);
    check<4, 2>("PP*jH^T*S.inverse()  (:75)", K, oK);
    check<4, 1>("xPred + K*y  (:76)", EM<4, 1>(xp + K * y), add(oxp, mul(oK, oy, false, false, E)));
    EM<4, 4> Pn = (EM<4, 4>::Identity() - K * jH) * PP;
    check<4, 4>("(I - K*jH)*PP  (:77)", Pn, mul(sub(OM<4, 4>::identity(), mul(oK, ojH, false, false, E)), oPP, false, false, E)); }
  { // lqr_speed_steer_control.cpp:91,104 (5x5, B 5x2, R 2x2)
    EM<5, 5> A, X, Q; OM<5, 5> oA, oX, oQ; EM<5, 2> B; OM<5, 2> oB; EM<2, 2> R; OM<2, 2> oR;
    fill(A, oA); fill(X, oX); fill(Q, oQ); fill(B, oB); fill(R, oR);
    R(0, 0) += 40.0f; oR(0, 0) += 40.0f; R(1, 1) += 40.0f; oR(1, 1) += 40.0f;
    EM<5, 5> Xn = A.transpose() * X * A - A.transpose() * X * B * (R + B.transpose() * X * B).inverse() * B.transpose() * X * A + Q;
    OM<5, 5> AtX = mul(transpose(oA), oX, true, false, E);
    OM<2, 2> Si = inverse2(add(oR, mul(mul(transpose(oB), oX, true, false, E), oB, false, false, E)));
    OM<5, 5> P2 = mul(mul(mul(mul(mul(AtX, oB, false, false, E), Si, false, false, E), transpose(oB), false, true, E), oX, false, false, E), oA, false, false, E);
    check<5, 5>("solve_DARE 5x5 iteration  (:91)", Xn, add(sub(mul(AtX, oA, false, false, E), P2), oQ));
    EM<2, 5> Kg = (B.transpose() * X * B + R).inverse() * (B.transpose() * X * A);
    OM<2, 5> BtX = mul(transpose(oB), oX, true, false, E);
    check<2, 5>("dlqr 5x5  (:104)", Kg, mul(inverse2(add(mul(BtX, oB, false, false, E), oR)), mul(BtX, oA, false, false, E), false, false, E)); }
  { // lqr_steer_control.cpp:81,94 (4x4, B 4x1, R scalar)
    EM<4, 4> A, X, Q; OM<4, 4> oA, oX, oQ; EM<4, 1> B; OM<4, 1> oB; fill(A, oA); fill(X, oX); fill(Q, oQ); fill(B, oB);
    const float R = 37.0f;
    EM<4, 4> Xn = A.transpose() * X * A - A.transpose() * X * B / (R + B.transpose() * X * B) * B.transpose() * X * A + Q;
    OM<4, 4> AtX = mul(transpose(oA), oX, true, false, E);
    OM<1, 4> BtX = mul(transpose(oB), oX, true, false, E);
    const float s = R + mul(BtX, oB, true, false, E)(0, 0);
    OM<4, 1> c2 = mul(AtX, oB, false, false, E);
    for (int i = 0; i < 4; ++i) c2.d[i] = c2.d[i] / s;
    OM<4, 4> P2 = mul(mul(mul(c2, transpose(oB), false, true, E), oX, false, false, E), oA, false, false, E);
    check<4, 4>("solve_DARE 4x4 iteration  (:81)", Xn, add(sub(mul(AtX, oA, false, false, E), P2), oQ));
    EM<1, 4> Kg = 1.0 / (B.transpose() * X * B + R) * (B.transpose() * X * A);
    OM<1, 4> BtXA = mul(BtX, oA, true, false, E);
    const float g = mul(BtX, oB, true, false, E)(0, 0) + R;
    const float f = (float)(1.0 / (double)g);
    OM<1, 4> oK; for (int i = 0; i < 4; ++i) oK.d[i] = f * BtXA.d[i];
    check<1, 4>("dlqr 4x4  (:94)", Kg, oK); }
}

}  // namespace

// ---- colPivHouseholderQr().solve() -------------------------------------------------------------------------------------------
int g_qr_fixed_cases = 0, g_qr_fixed_fail = 0, g_qr_dyn_cases = 0, g_qr_dyn_fail = 0;
template <int N> void qr_fixed(const char* what, const Eigen::Matrix<float, N, N>& A, const Eigen::Matrix<float, N, 1>& b) {
  const Eigen::Matrix<float, N, 1> xe = A.colPivHouseholderQr().solve(b);
  float a[N * N], bb[N], xo[N];
  for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) a[i + N * j] = A(i, j);
  for (int i = 0; i < N; ++i) bb[i] = b(i);
  oracle::ColPivQR<float> qr(N, N, true);
  qr.compute(a); qr.solve(bb, xo);
  ++g_qr_fixed_cases;
  for (int i = 0; i < N; ++i) {
    const float e = xe(i);
    if (std::memcmp(&e, &xo[i], 4) != 0) {
      if (g_qr_fixed_fail < 20) std::printf("QR MISMATCH %-30s x[%d]: Eigen %a  eigen_qr.h %a\n", what, i, e, xo[i]);
      ++g_qr_fixed_fail;
      return;
    }
  }
}
void qr_dynamic(const char* what, const Eigen::MatrixXf& A, const Eigen::VectorXf& b) {
  const int n = (int)A.rows();
  const Eigen::VectorXf xe = A.colPivHouseholderQr().solve(b);
  std::vector<float> a((size_t)n * n), bb(n), xo(n);
  for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) a[i + (size_t)n * j] = A(i, j);
  for (int i = 0; i < n; ++i) bb[i] = b(i);
  oracle::ColPivQR<float> qr(n, n, false);
  qr.compute(a.data()); qr.solve(bb.data(), xo.data());
  ++g_qr_dyn_cases;
  for (int i = 0; i < n; ++i) {
    const float e = xe(i);
    if (std::memcmp(&e, &xo[i], 4) != 0) {
      if (g_qr_dyn_fail < 10) std::printf("QR (dynamic, unpinned) differs %-24s n=%d x[%d]: Eigen %a  eigen_qr.h %a\n", what, n, i, e, xo[i]);
      ++g_qr_dyn_fail;
      return;
    }
  }
}
void qr_one_seed(int seed) {
  { Eigen::Matrix3f A; Eigen::Vector3f b;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) A(i, j) = rnd();
    for (int i = 0; i < 3; ++i) b(i) = rnd();
    qr_fixed<3>("Matrix3f random", A, b); }
  { Eigen::Matrix2f A; Eigen::Vector2f b;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) A(i, j) = rnd();
    for (int i = 0; i < 2; ++i) b(i) = rnd();
    qr_fixed<2>("Matrix2f random", A, b); }
  // the reference's own matrices: the planner's horizons T = MINT .. MAXT in steps of DT (4.0 .. 5.0 by 0.2), a few more around them
  { const float T = 3.0f + 0.05f * (float)(seed % 60);
    Eigen::Matrix3f A; Eigen::Vector3f b;
    A << std::pow(T, 3), std::pow(T, 4), std::pow(T, 5),
         3 * std::pow(T, 2), 4 * std::pow(T, 3), 5 * std::pow(T, 4),
         6 * T, 12 * std::pow(T, 2), 20 * std::pow(T, 3);                  // quintic_polynomial.h:40-42
    for (int i = 0; i < 3; ++i) b(i) = 4.0f * rnd();
    qr_fixed<3>("quintic time matrix", A, b);
    Eigen::Matrix2f A2; Eigen::Vector2f b2;
    A2 << 3 * std::pow(T, 2), 4 * std::pow(T, 3),
          6 * T, 12 * std::pow(T, 2);                                      // quartic_polynomial.h:37-38
    for (int i = 0; i < 2; ++i) b2(i) = 4.0f * rnd();
    qr_fixed<2>("quartic time matrix", A2, b2); }
  for (int n = 4; n <= 12; ++n) {
    Eigen::MatrixXf A(n, n); Eigen::VectorXf b(n);
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) A(i, j) = rnd();
    for (int i = 0; i < n; ++i) b(i) = rnd();
    qr_dynamic("MatrixXf random", A, b);
    // the spline's system, cubic_spline.h:95-116: h = knot spacings
    std::vector<float> h(n - 1);
    for (auto& v : h) v = 1.5f + 0.5f * rnd();
    Eigen::MatrixXf S = Eigen::MatrixXf::Zero(n, n); Eigen::VectorXf B = Eigen::VectorXf::Zero(n);
    S(0, 0) = 1;
    for (int i = 0; i < n - 1; ++i) {
      if (i != n - 2) S(i + 1, i + 1) = 2 * (h[i] + h[i + 1]);
      S(i + 1, i) = h[i];
      S(i, i + 1) = h[i];
    }
    S(0, 1) = 0.0; S(n - 1, n - 2) = 0.0; S(n - 1, n - 1) = 1.0;
    for (int i = 1; i < n - 1; ++i) B(i) = 3.0f * rnd();
    qr_dynamic("spline system", S, B);
  }
}

int main(int argc, char** argv) {
  const int seeds = argc > 1 ? std::atoi(argv[1]) : 200;
  for (int s = 1; s <= seeds; ++s) { g_state = 0x9E3779B97F4A7C15ULL * (uint64_t)s + 1; one_seed(); qr_one_seed(s); }
#ifdef EIGEN_STANDIN_FOR_REFERENCE_TESTS
  const char* what = "the Eigen stand-in (oracle/ref_shim/Eigen/Eigen)";
#else
  char ver[64]; std::snprintf(ver, sizeof ver, "Eigen %d.%d.%d", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION);
  const char* what = ver;
#endif
  std::printf("eigen_order_probe: colPivHouseholderQr().solve(): %d fixed-size systems (Matrix3f / Matrix2f, random and the quintic / quartic "
              "time matrices) against %s: %d mismatching; %d dynamic systems (MatrixXf n = 4 .. 12, random and spline; eigen_qr.h: UNPINNED): %d differing\n",
              g_qr_fixed_cases, what, g_qr_fixed_fail, g_qr_dyn_cases, g_qr_dyn_fail);
  std::printf("eigen_order_probe: %d products / chains x %d seeds against %s: %d mismatching\n", g_cases / (seeds ? seeds : 1), seeds, what, g_fail);
  return (g_fail || g_qr_fixed_fail) ? 1 : 0;
}

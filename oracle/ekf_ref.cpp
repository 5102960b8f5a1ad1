// ekf_ref.cpp — TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline for bench.py).
// Nothing under cpprobotics_amd/ may include, link or call this file.
//
// CPU restatement of the reference's EKF localisation arithmetic,
//   /root/reference/src/extended_kalman_filter.cpp:22-78 (motion_model, jacobF,
//   observation_model, jacobH, ekf_estimation) and the input side of its main loop :171-183,
// written as the dense matrix expressions the reference writes, evaluated in the order
// Eigen 3.3.9 evaluates them (oracle/eigen_order.h).  PINNED against the reference's own lines (oracle/ref_build.sh compiles them unmodified — against the host's Eigen, or against the
// Eigen stand-in oracle/ref_shim/Eigen/Eigen where there is none — and tests/test_oracle_vs_ref.py demands equal bits); unpinned only
// with respect to Eigen's own binary, absent from every host of this project.
// Also validated against an independent numpy-float32 twin (oracle/np_twin.py), a float64 evaluation and the filter's own
// invariants (tests/test_oracle_ekf.py).
//
// Build: oracle/Makefile (g++ -O2 -ffp-contract=off, no -march: SSE2, no FMA — the reference's
// own arithmetic, CMakeLists.txt:4-6).
#include <cmath>
#include <cstdint>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "eigen_order.h"
#define CRX_TRIG_FMA 1
#include "../cpprobotics_amd/csrc/crx_trig.h"  // only for trig_mode != 0 (hosts whose libm is not the FMA flavour)

using namespace oracle;

namespace {

// trig_mode 0: host libm sinf/cosf — what std::cos(float)/std::sin(float) resolve to in the
// reference (:30-31,42-45).  trig_mode 1: the explicit glibc-FMA-flavour restatement.
inline float o_cos(float x, int trig_mode) { return trig_mode == 0 ? cosf(x) : crx::cosf_(x); }
inline float o_sin(float x, int trig_mode) { return trig_mode == 0 ? sinf(x) : crx::sinf_(x); }

// :22-36
Mat<4, 1> motion_model(const Mat<4, 1>& x, const Mat<2, 1>& u, double DT, int tm, SumOrder ord) {
  Mat<4, 4> F_ = Mat<4, 4>::identity();                    // :23-27
  Mat<4, 2> B_;
  B_(0, 0) = (float)(DT * (double)o_cos(x(2, 0), tm)); B_(0, 1) = 0.0f;   // :30
  B_(1, 0) = (float)(DT * (double)o_sin(x(2, 0), tm)); B_(1, 1) = 0.0f;   // :31
  B_(2, 0) = 0.0f;                                      B_(2, 1) = (float)DT;  // :32
  B_(3, 0) = 1.0f;                                      B_(3, 1) = 0.0f;   // :33
  return add(mul(F_, x, false, false, ord), mul(B_, u, false, false, ord));  // :35
}

// :38-47
Mat<4, 4> jacobF(const Mat<4, 1>& x, const Mat<2, 1>& u, double DT, int tm) {
  Mat<4, 4> jF = Mat<4, 4>::identity();
  float yaw = x(2, 0);
  float v = u(0, 0);
  jF(0, 2) = (float)(-DT * (double)v * (double)o_sin(yaw, tm));  // :42  ((-DT)*v)*sin
  jF(0, 3) = (float)(DT * (double)o_cos(yaw, tm));               // :43
  jF(1, 2) = (float)(DT * (double)v * (double)o_cos(yaw, tm));   // :44
  jF(1, 3) = (float)(DT * (double)o_sin(yaw, tm));               // :45
  return jF;
}

Mat<2, 4> jacobH() {  // :57-62
  Mat<2, 4> h = Mat<2, 4>::zero();
  h(0, 0) = 1.0f; h(1, 1) = 1.0f;
  return h;
}

Mat<2, 1> observation_model(const Mat<4, 1>& x, SumOrder ord) {  // :50-55
  return mul(jacobH(), x, false, false, ord);
}

// :64-78
void ekf_estimation(Mat<4, 1>& xEst, Mat<4, 4>& PEst, const Mat<2, 1>& z, const Mat<2, 1>& u,
                    const Mat<4, 4>& Q, const Mat<2, 2>& R, double DT, int tm, SumOrder ord) {
  Mat<4, 1> xPred = motion_model(xEst, u, DT, tm, ord);                                  // :67
  Mat<4, 4> jF = jacobF(xPred, u, DT, tm);                                               // :68
  Mat<4, 4> PPred = add(mul(mul(jF, PEst, false, false, ord), transpose(jF), false, true, ord), Q);  // :69
  Mat<2, 4> jH = jacobH();                                                               // :71
  Mat<2, 1> zPred = observation_model(xPred, ord);                                       // :72
  Mat<2, 1> y = sub(z, zPred);                                                           // :73
  Mat<2, 2> S = add(mul(mul(jH, PPred, false, false, ord), transpose(jH), false, true, ord), R);     // :74
  Mat<4, 2> K = mul(mul(PPred, transpose(jH), false, true, ord), inverse2(S), false, false, ord);    // :75
  xEst = add(xPred, mul(K, y, false, false, ord));                                       // :76
  PEst = mul(sub(Mat<4, 4>::identity(), mul(K, jH, false, false, ord)), PPred, false, false, ord);  // :77
}

template <int R, int C> Mat<R, C> load(const float* p) { Mat<R, C> m; std::memcpy(m.d, p, sizeof(m.d)); return m; }
template <int R, int C> void store(float* p, const Mat<R, C>& m) { std::memcpy(p, m.d, sizeof(m.d)); }

}  // namespace

extern "C" {

void oracle_motion_model(int n, const float* x, const float* u, float* x_out, double dt, int trig_mode) {
  for (int k = 0; k < n; ++k)
    store(x_out + 4 * k, motion_model(load<4, 1>(x + 4 * k), load<2, 1>(u + 2 * k), dt, trig_mode, ORDER_EIGEN));
}

void oracle_jacobF(int n, const float* x, const float* u, float* jF, double dt, int trig_mode) {
  for (int k = 0; k < n; ++k)
    store(jF + 16 * k, jacobF(load<4, 1>(x + 4 * k), load<2, 1>(u + 2 * k), dt, trig_mode));
}

void oracle_observation_model(int n, const float* x, float* z) {
  for (int k = 0; k < n; ++k) store(z + 2 * k, observation_model(load<4, 1>(x + 4 * k), ORDER_EIGEN));
}

void oracle_jacobH(float* jH) { store(jH, jacobH()); }

// One ekf_estimation() per agent.  sum_order: 0 = Eigen's (see eigen_order.h), 1 = all-ascending.
void oracle_ekf_step(int n, float* x, float* P, const float* z, const float* u, const float* Q,
                     const float* R, double dt, int trig_mode, int sum_order) {
  Mat<4, 4> Qm = load<4, 4>(Q);
  Mat<2, 2> Rm = load<2, 2>(R);
  for (int k = 0; k < n; ++k) {
    Mat<4, 1> xe = load<4, 1>(x + 4 * k);
    Mat<4, 4> Pe = load<4, 4>(P + 16 * k);
    ekf_estimation(xe, Pe, load<2, 1>(z + 2 * k), load<2, 1>(u + 2 * k), Qm, Rm, dt, trig_mode, (SumOrder)sum_order);
    store(x + 4 * k, xe);
    store(P + 16 * k, Pe);
  }
}

// T steps per agent; z,u time-major [T][n][2]; x_hist [T][n][4], P_hist [T][n][16] (NULL ok).
// Agents [a0, a1) only (lets bench.py time a bounded sample / split across threads).
void oracle_ekf_run(int n, int T, float* x, float* P, const float* z, const float* u,
                    float* x_hist, float* P_hist, const float* Q, const float* R, double dt,
                    int trig_mode, int sum_order, int a0, int a1) {
  Mat<4, 4> Qm = load<4, 4>(Q);
  Mat<2, 2> Rm = load<2, 2>(R);
  // z,u are time-major: walk time in the outer loop over a block of vehicles so that memory is
  // streamed in the order it is laid out (each vehicle's arithmetic is unaffected by the blocking).
  constexpr int BLK = 64;
  for (int b0 = a0; b0 < a1; b0 += BLK) {
    const int nb = (a1 - b0 < BLK) ? (a1 - b0) : BLK;
    Mat<4, 1> xe[BLK];
    Mat<4, 4> Pe[BLK];
    for (int j = 0; j < nb; ++j) { xe[j] = load<4, 1>(x + 4 * (b0 + j)); Pe[j] = load<4, 4>(P + 16 * (b0 + j)); }
    for (int t = 0; t < T; ++t)
      for (int j = 0; j < nb; ++j) {
        const size_t o = (size_t)t * n + (b0 + j);
        ekf_estimation(xe[j], Pe[j], load<2, 1>(z + 2 * o), load<2, 1>(u + 2 * o), Qm, Rm, dt, trig_mode, (SumOrder)sum_order);
        if (x_hist) store(x_hist + 4 * o, xe[j]);
        if (P_hist) store(P_hist + 16 * o, Pe[j]);
      }
    for (int j = 0; j < nb; ++j) { store(x + 4 * (b0 + j), xe[j]); store(P + 16 * (b0 + j), Pe[j]); }
  }
}

// The same, all agents, in ONE OpenMP region (static partition of the vehicle blocks) — the CPU baseline bench.py times.
// Returns the number of threads the region ran with.
int oracle_ekf_run_omp(int n, int T, float* x, float* P, const float* z, const float* u, float* x_hist, float* P_hist,
                       const float* Q, const float* R, double dt, int trig_mode, int sum_order, int nthreads) {
  int used = 1;
  constexpr int BLK = 64;
  const int nblk = (n + BLK - 1) / BLK;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
  for (int b = 0; b < nblk; ++b) {
#ifdef _OPENMP
    if (b == 0) used = omp_get_num_threads();
#endif
    const int a0 = b * BLK, a1 = (a0 + BLK < n) ? a0 + BLK : n;
    oracle_ekf_run(n, T, x, P, z, u, x_hist, P_hist, Q, R, dt, trig_mode, sum_order, a0, a1);
  }
  return used;
}

// Input side of the reference's loop, /root/reference/src/extended_kalman_filter.cpp:174-181.
// w: [T][n][4] standard-normal draws (float; the reference draws doubles from a
// random_device-seeded mt19937, which cannot be reproduced, so the draws are an input here).
void oracle_ekf_simulate_inputs(int n, int T, const float* u_true, float* xTrue, float* xDR,
                                const float* w, float* z, float* ud, float* xTrue_hist,
                                float* xDR_hist, const float* qsim, const float* rsim, double dt,
                                int trig_mode) {
  for (int k = 0; k < n; ++k) {
    Mat<2, 1> u = load<2, 1>(u_true + 2 * k);
    Mat<4, 1> xt = load<4, 1>(xTrue + 4 * k);
    Mat<4, 1> xd = load<4, 1>(xDR + 4 * k);
    for (int t = 0; t < T; ++t) {
      size_t o = (size_t)t * n + k;
      Mat<2, 1> udv;
      // ud(i) = u(i) + gaussian_d(gen) * Qsim(i,i)   :174-175  (float + double*float -> double -> float)
      udv(0, 0) = (float)((double)u(0, 0) + (double)w[4 * o + 0] * (double)qsim[0]);
      udv(1, 0) = (float)((double)u(1, 0) + (double)w[4 * o + 1] * (double)qsim[1]);
      xt = motion_model(xt, u, dt, trig_mode, ORDER_EIGEN);     // :177
      xd = motion_model(xd, udv, dt, trig_mode, ORDER_EIGEN);   // :178
      // z(i) = xTrue(i) + gaussian_d(gen) * Rsim(i,i)   :180-181
      float z0 = (float)((double)xt(0, 0) + (double)w[4 * o + 2] * (double)rsim[0]);
      float z1 = (float)((double)xt(1, 0) + (double)w[4 * o + 3] * (double)rsim[1]);
      z[2 * o] = z0; z[2 * o + 1] = z1;
      ud[2 * o] = udv(0, 0); ud[2 * o + 1] = udv(1, 0);
      if (xTrue_hist) store(xTrue_hist + 4 * o, xt);
      if (xDR_hist) store(xDR_hist + 4 * o, xd);
    }
    store(xTrue + 4 * k, xt);
    store(xDR + 4 * k, xd);
  }
}

// Which glibc flavour the host libm dispatches to: 1 = FMA variant, 0 = SSE2 variant,
// decided on inputs where the two differ (found by tests/tools/trig_exhaustive.cpp).
int oracle_libm_is_fma_flavour(void) {
  // volatile: keep the compiler from folding cosf(constant) at build time
  volatile uint32_t probes[] = {0xc18a3adbu, 0xc2870e40u};
  int agree = 0;
  for (unsigned i = 0; i < sizeof(probes) / sizeof(probes[0]); ++i) {
    uint32_t b = probes[i]; float x; std::memcpy(&x, &b, 4);
    if (cosf(x) == crx::cosf_(x) && sinf(x) == crx::sinf_(x)) agree++;
  }
  return agree == (int)(sizeof(probes) / sizeof(probes[0]));
}

}  // extern "C"

// ---- the engine's counter-based input noise, evaluated on the host -----------------------------------------------------------
// Not part of the reference (whose generator is random_device-seeded): the SAME header the device code compiles
// (cpprobotics_amd/csrc/crx_philox.h, flowing product -> oracle like crx_trig.h), run on the CPU so that tests can demand
// host == device bytes and shard-independence without a GPU.
#include "../cpprobotics_amd/csrc/crx_philox.h"
extern "C" void oracle_normal_draws(int n, int T, long long agent0, unsigned long long seed, unsigned stream_id, float* w) {
  for (int t = 0; t < T; ++t)
    for (int a = 0; a < n; ++a) crx::philox_normal4(seed, stream_id, (uint64_t)(agent0 + a), (uint32_t)t, w + ((size_t)t * n + a) * 4);
}
extern "C" void oracle_philox4x32_10(unsigned* ctr, unsigned k0, unsigned k1) { crx::philox4x32_10(ctr, k0, k1); }

// ref_ekf.cpp — TEST INFRASTRUCTURE.  Exports the reference's EKF functions, compiled from its own lines
// (/root/reference/src/extended_kalman_filter.cpp:21-78 and the main loop :112-161, :172-188; see oracle/ref_build.sh).
#include <cmath>
#include <cstring>
#include <iostream>
#include <random>
#include <vector>
#include <Eigen/Eigen>
#include "ekf_defs.inc"

namespace ref_ekf {
#include "ekf_fns.inc"

// stands in for `std::normal_distribution<> gaussian_d{0,1}` drawing from `gen` (:162-164): hands out the caller's draws in order
struct NoiseSource {
  const double* w;
  long k;
  template <class G> double operator()(G&) { return w[k++]; }
};
}  // namespace ref_ekf

extern "C" {

int ref_eigen_kind(void) { return REF_EIGEN_KIND; }   // 1 = the host's Eigen, 0 = oracle/ref_shim/Eigen/Eigen

static Eigen::Vector4f v4(const float* p) { Eigen::Vector4f v; std::memcpy(v.data(), p, 16); return v; }
static Eigen::Vector2f v2(const float* p) { Eigen::Vector2f v; std::memcpy(v.data(), p, 8); return v; }
static Eigen::Matrix4f m4(const float* p) { Eigen::Matrix4f m; std::memcpy(m.data(), p, 64); return m; }
static Eigen::Matrix2f m2(const float* p) { Eigen::Matrix2f m; std::memcpy(m.data(), p, 16); return m; }

void ref_motion_model(int n, const float* x, const float* u, float* out) {
  for (int a = 0; a < n; ++a) { Eigen::Vector4f r = ref_ekf::motion_model(v4(x + 4 * a), v2(u + 2 * a)); std::memcpy(out + 4 * a, r.data(), 16); }
}
void ref_jacobF(int n, const float* x, const float* u, float* out) {
  for (int a = 0; a < n; ++a) { Eigen::Matrix4f r = ref_ekf::jacobF(v4(x + 4 * a), v2(u + 2 * a)); std::memcpy(out + 16 * a, r.data(), 64); }
}
void ref_observation_model(int n, const float* x, float* out) {
  for (int a = 0; a < n; ++a) { Eigen::Vector2f r = ref_ekf::observation_model(v4(x + 4 * a)); std::memcpy(out + 2 * a, r.data(), 8); }
}
void ref_jacobH(float* out) { Eigen::Matrix<float, 2, 4> r = ref_ekf::jacobH(); std::memcpy(out, r.data(), 32); }

// T calls of ekf_estimation per agent; z, u time-major [T][n][2]; x_hist [T][n][4], P_hist [T][n][16] may be NULL
void ref_ekf_run(int n, int T, float* x, float* P, const float* z, const float* u, float* x_hist, float* P_hist, const float* Q, const float* R) {
  const Eigen::Matrix4f Qm = m4(Q);
  const Eigen::Matrix2f Rm = m2(R);
  for (int a = 0; a < n; ++a) {
    Eigen::Vector4f xEst = v4(x + 4 * a);
    Eigen::Matrix4f PEst = m4(P + 16 * a);
    for (int t = 0; t < T; ++t) {
      const size_t q = (size_t)t * n + a;
      ref_ekf::ekf_estimation(xEst, PEst, v2(z + 2 * q), v2(u + 2 * q), Qm, Rm);
      if (x_hist) std::memcpy(x_hist + 4 * q, xEst.data(), 16);
      if (P_hist) std::memcpy(P_hist + 16 * q, PEst.data(), 64);
    }
    std::memcpy(x + 4 * a, xEst.data(), 16);
    std::memcpy(P + 16 * a, PEst.data(), 64);
  }
}

// main() :110-186 for `steps` passes of the while loop, the four N(0,1) draws of each pass supplied by the caller (noise[4*steps]).
// Outputs per step: hxTrue, hxDR, hxEst [steps][4], hz, hud [steps][2]; final PEst[16]; the constants Q[16], R[4], Qsim[4], Rsim[4].
void ref_ekf_main(int steps, const double* noise, float* hxTrue_o, float* hxDR_o, float* hxEst_o, float* hz_o, float* hud_o, float* PEst_o,
                  float* Q_o, float* R_o, float* Qsim_o, float* Rsim_o) {
  using namespace ref_ekf;
  float time = 0.0;
#include "ekf_main_setup.inc"
  NoiseSource gaussian_d{noise, 0};
  int gen = 0;
  for (int step = 0; step < steps; ++step) {     // `while(time <= SIM_TIME)` :171, bounded by the caller instead
#include "ekf_main_body.inc"
    std::memcpy(hud_o + 2 * step, ud.data(), 8);
  }
  for (int s = 0; s < steps; ++s) {
    std::memcpy(hxTrue_o + 4 * s, hxTrue[s].data(), 16); std::memcpy(hxDR_o + 4 * s, hxDR[s].data(), 16);
    std::memcpy(hxEst_o + 4 * s, hxEst[s].data(), 16); std::memcpy(hz_o + 2 * s, hz[s].data(), 8);
  }
  std::memcpy(PEst_o, PEst.data(), 64); std::memcpy(Q_o, Q.data(), 64); std::memcpy(R_o, R.data(), 16);
  std::memcpy(Qsim_o, Qsim.data(), 16); std::memcpy(Rsim_o, Rsim.data(), 16);
}

}  // extern "C"

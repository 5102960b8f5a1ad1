// ref_pf.cpp — TEST INFRASTRUCTURE.  Exports the reference's particle-filter functions, compiled from its own lines
// (/root/reference/src/particle_filter.cpp:17-22 the #defines, :25-148 motion_model ... resampling, and of main() the set-up
// :179-235 and the loop body :249-271; see oracle/ref_build.sh).
//
// The reference draws its random numbers inside these functions from `std::mt19937 gen` / `std::normal_distribution<>` /
// `std::uniform_real_distribution<>` objects that it passes BY VALUE (:78, :123-124, :270-271).  To feed it the caller's draws
// the three names are redirected, for the extent of the included lines only, to a source that hands out an array in order —
// copied by value like the originals, so the reference's quirk survives: the callee's copy advances, the caller's does not
// (pf_localization consumes the stream positions main() itself uses in the next pass; resampling sees the same uniforms in
// every pass).
#include <cmath>
#include <cstring>
#include <iostream>
#include <random>
#include <vector>
#include <Eigen/Eigen>
#include "pf_defs.inc"

namespace std {   // test-only names inside std so that the reference's `std::` qualifications resolve (see the #defines below)
struct ref_pf_engine {};
template <class T = double>
struct ref_pf_draws {
  const double* w;
  long k;
  template <class G> double operator()(G&) { return w[k++]; }
};
}  // namespace std

namespace ref_pf {
#define mt19937 ref_pf_engine
#define normal_distribution ref_pf_draws
#define uniform_real_distribution ref_pf_draws
#include "pf_fns.inc"
#undef mt19937
#undef normal_distribution
#undef uniform_real_distribution
}  // namespace ref_pf

extern "C" {

int ref_pf_np(void) { return NP; }

static Eigen::Vector4f pv4(const float* p) { Eigen::Vector4f v; std::memcpy(v.data(), p, 16); return v; }
static Eigen::Vector2f pv2(const float* p) { Eigen::Vector2f v; std::memcpy(v.data(), p, 8); return v; }
typedef Eigen::Matrix<float, 4, NP> PxT;
typedef Eigen::Matrix<float, NP, 1> PwT;

void ref_pf_motion_model(int n, const float* x, const float* u, float* out) {
  for (int a = 0; a < n; ++a) { Eigen::Vector4f r = ref_pf::motion_model(pv4(x + 4 * a), pv2(u + 2 * a)); std::memcpy(out + 4 * a, r.data(), 16); }
}
void ref_pf_gauss_likelihood(int n, const float* x, const float* sigma, float* out) {
  for (int a = 0; a < n; ++a) out[a] = ref_pf::gauss_likelihood(x[a], sigma[a]);
}
void ref_pf_calc_covariance(const float* xEst, const float* px, const float* pw, float* PEst) {
  PxT X; PwT W;
  std::memcpy(X.data(), px, sizeof(float) * 4 * NP); std::memcpy(W.data(), pw, sizeof(float) * NP);
  Eigen::Matrix4f P = ref_pf::calc_covariance(pv4(xEst), X, W);
  std::memcpy(PEst, P.data(), 64);
}
void ref_pf_cumsum(const float* pw, float* out) {
  PwT W; std::memcpy(W.data(), pw, sizeof(float) * NP);
  PwT c = ref_pf::cumsum(W);
  std::memcpy(out, c.data(), sizeof(float) * NP);
}
// one pf_localization (:73-109): z [nz][3] = (range, landmark x, landmark y); nrm: the 2*NP normal draws it consumes, in order
void ref_pf_localization(float* px, float* pw, float* xEst, float* PEst, const float* z, int nz, const float* u, const float* rsim,
                         float Q, const double* nrm) {
  PxT X; PwT W;
  std::memcpy(X.data(), px, sizeof(float) * 4 * NP); std::memcpy(W.data(), pw, sizeof(float) * NP);
  Eigen::Vector4f xe = pv4(xEst);
  Eigen::Matrix4f Pe; std::memcpy(Pe.data(), PEst, 64);
  std::vector<Eigen::RowVector3f> zs;
  for (int i = 0; i < nz; ++i) { Eigen::RowVector3f zi; zi << z[3 * i], z[3 * i + 1], z[3 * i + 2]; zs.push_back(zi); }
  Eigen::Matrix2f Rsim; std::memcpy(Rsim.data(), rsim, 16);
  ref_pf::pf_localization(X, W, xe, Pe, zs, pv2(u), Rsim, Q, std::ref_pf_engine{}, std::ref_pf_draws<>{nrm, 0});
  std::memcpy(px, X.data(), sizeof(float) * 4 * NP); std::memcpy(pw, W.data(), sizeof(float) * NP);
  std::memcpy(xEst, xe.data(), 16); std::memcpy(PEst, Pe.data(), 64);
}
// one resampling (:120-148); uni: the NP uniform draws (the reference's distribution is U[1,2), :242) it consumes if it resamples
void ref_pf_resampling(float* px, float* pw, const double* uni) {
  PxT X; PwT W;
  std::memcpy(X.data(), px, sizeof(float) * 4 * NP); std::memcpy(W.data(), pw, sizeof(float) * NP);
  ref_pf::resampling(X, W, std::ref_pf_engine{}, std::ref_pf_draws<>{uni, 0});
  std::memcpy(px, X.data(), sizeof(float) * 4 * NP); std::memcpy(pw, W.data(), sizeof(float) * NP);
}

// main() :179-235 + `steps` passes of the loop body :249-271 with the caller's streams: w (normal draws, consumed as main and —
// by value — pf_localization consume `gen`), uni (the NP uniforms every resampling call replays).  Per pass: ud [2], xTrue [4],
// xDR [4], z [4][3] (zero padded) + nz, xEst [4], PEst [16], and the particle store after the pass px [NP][4], pw [NP].
// consts_o: Q, Qsim, Rsim(0,0), Rsim(1,1), R(0,0), R(1,1).  Returns the number of normal draws main itself consumed.
long ref_pf_main(int steps, const double* w, const double* uni, float* ud_o, float* xTrue_o, float* xDR_o, float* z_o, int* nz_o,
                 float* xEst_o, float* PEst_o, float* px_o, float* pw_o, float* consts_o) {
  using namespace ref_pf;
#include "pf_main_setup.inc"
  std::ref_pf_engine gen, gen2;
  std::ref_pf_draws<> gaussian_d{w, 0}, uni_d{uni, 0};
  for (int step = 0; step < steps; ++step) {     // `while(time <= SIM_TIME)` :248, bounded by the caller instead
#include "pf_main_body.inc"
    std::memcpy(ud_o + 2 * step, ud.data(), 8); std::memcpy(xTrue_o + 4 * step, xTrue.data(), 16); std::memcpy(xDR_o + 4 * step, xDR.data(), 16);
    std::memset(z_o + 12 * step, 0, 48);
    for (size_t i = 0; i < z.size(); ++i) std::memcpy(z_o + 12 * step + 3 * i, z[i].data(), 12);
    nz_o[step] = (int)z.size();
    std::memcpy(xEst_o + 4 * step, xEst.data(), 16); std::memcpy(PEst_o + 16 * step, PEst.data(), 64);
    std::memcpy(px_o + (size_t)4 * NP * step, px.data(), sizeof(float) * 4 * NP); std::memcpy(pw_o + (size_t)NP * step, pw.data(), sizeof(float) * NP);
  }
  consts_o[0] = Q; consts_o[1] = Qsim; consts_o[2] = Rsim(0, 0); consts_o[3] = Rsim(1, 1); consts_o[4] = R(0, 0); consts_o[5] = R(1, 1);
  return gaussian_d.k;
}

}  // extern "C"

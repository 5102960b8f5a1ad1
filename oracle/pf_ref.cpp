// pf_ref.cpp — TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline).
// Nothing under cpprobotics_amd/ may include, link or call this file.
//
// CPU restatement of the reference's particle-filter localisation, /root/reference/src/particle_filter.cpp:
//   motion_model :26-40 (same as the EKF's), gauss_likelihood :53-57, calc_covariance :59-71,
//   pf_localization :73-109, cumsum :111-118, resampling :120-148, and the observation side of main() :251-268.
// The reference draws its noise from std::mt19937 inside these functions (and passes the generator BY VALUE, so
// every call replays the same stream); here all random numbers are caller-supplied arrays, so that CPU and GPU
// consume identical draws:  nrm[t][a][ip][2] standard normals (motion noise, :87-88),  uni[t][a][j] uniforms in
// [1,2) (uni_d{1.0, 2.0} :242, used as uni/NP :133).
// PINNED against the reference's own lines (oracle/ref_build.sh compiles particle_filter.cpp:25-148 and the loop of main()
// unmodified into oracle/_ref/libref.so, random draws injected; tests/test_oracle_vs_ref.py::test_pf_* demand equal bits) for
// everything but THREE sums: pw.sum() (:104), px * pw (:106) and pw.transpose() * pw (:126) are 100-term reductions that Eigen
// evaluates with its vectorised redux / gemv kernels, whose accumulation order is not restated here (PARITY-UNPINNED, those
// three only).  oracle_pf_step takes them in plain index order; against the reference's lines they are compared at 2e-6
// (relative, floored by the largest entry), and bit for bit wherever the order cannot matter (at most two non-zero weights).
// oracle_pf_step_wave takes them in the engine's order (balanced tree over 64 lanes) and the engine is demanded equal to it bit
// for bit — everything else of a tick (motion model, likelihood, normalisation, covariance terms, Neff test, low-variance
// resampling) is the reference's arithmetic in both.
#include <cmath>
#include <cstring>
#include <vector>
#define CRX_TRIG_FMA 1
#include "../cpprobotics_amd/csrc/crx_trig.h"

namespace {

const double PI_ = 3.141592653;     // `#define PI 3.141592653` :19
int g_trig = 0;
inline float o_cos(float x) { return g_trig == 0 ? std::cos(x) : crx::cosf_(x); }
inline float o_sin(float x) { return g_trig == 0 ? std::sin(x) : crx::sinf_(x); }

// :26-40   x <- F*x + B*u ; B = [[DT cos, 0], [DT sin, 0], [0, DT], [1, 0]]
void motion_model(float* x, const float* u, double DT) {
  const float b0 = (float)(DT * (double)o_cos(x[2])), b1 = (float)(DT * (double)o_sin(x[2])), b2 = (float)DT;
  const float n0 = x[0] + b0 * u[0], n1 = x[1] + b1 * u[0], n2 = x[2] + b2 * u[1], n3 = x[3] + u[0];
  x[0] = n0; x[1] = n1; x[2] = n2; x[3] = n3;
}

// :53-57
float gauss_likelihood(float x, float sigma) {
  float p = 1.0 / std::sqrt(2.0 * PI_ * sigma * sigma) * std::exp(-x * x / (2 * sigma * sigma));
  return p;
}

}  // namespace

extern "C" {

void oracle_pf_set_trig_mode(int m) { g_trig = m; }

void oracle_pf_gauss_likelihood(int n, const float* x, const float* sigma, float* out) {
  for (int i = 0; i < n; ++i) out[i] = gauss_likelihood(x[i], sigma[i]);
}

// calc_covariance :59-71 for one vehicle: PEst_ += pw(i) * dx * dx.transpose(), particle after particle
void oracle_pf_calc_covariance(int NP, const float* xe, const float* X, const float* W, float* Pe) {
  std::memset(Pe, 0, sizeof(float) * 16);
  for (int i = 0; i < NP; ++i) {
    float dx[4]; for (int r = 0; r < 4; ++r) dx[r] = X[4 * i + r] - xe[r];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) Pe[r + 4 * c] += (W[i] * dx[r]) * dx[c];
  }
}

// pf_localization :73-109 for one vehicle.  X [NP][4], W [NP] in/out; Z [nz][3]; nrm [NP][2]
static void pf_localization_one(int NP, float* X, float* W, float* xe, float* Pe, const float* Z, int nz, const float* u,
                                const float* nrm, const float* rsim, float Q, double DT) {
  const float sig = std::sqrt(Q);                                     // std::sqrt(Q) :98
  for (int ip = 0; ip < NP; ++ip) {                                   // :81
    float x[4] = {X[4 * ip], X[4 * ip + 1], X[4 * ip + 2], X[4 * ip + 3]};
    float w = W[ip];
    float ud[2];
    ud[0] = u[0] + (double)nrm[2 * ip] * rsim[0];                     // :87  gaussian_d(gen) is a double
    ud[1] = u[1] + (double)nrm[2 * ip + 1] * rsim[1];
    motion_model(x, ud, DT);                                          // :90
    for (int i = 0; i < nz; ++i) {                                    // :92
      float dx = x[0] - Z[3 * i + 1];
      float dy = x[1] - Z[3 * i + 2];
      float prez = std::sqrt(dx * dx + dy * dy);
      float dz = prez - Z[3 * i];
      w = w * gauss_likelihood(dz, sig);
    }
    X[4 * ip] = x[0]; X[4 * ip + 1] = x[1]; X[4 * ip + 2] = x[2]; X[4 * ip + 3] = x[3];
    W[ip] = w;
  }
  float s = 0.0f;
  for (int i = 0; i < NP; ++i) s += W[i];                             // pw.sum(): index order here (header)
  for (int i = 0; i < NP; ++i) W[i] = W[i] / s;                       // pw = pw / pw.sum() :104
  xe[0] = xe[1] = xe[2] = xe[3] = 0.0f;
  for (int i = 0; i < NP; ++i) for (int r = 0; r < 4; ++r) xe[r] += X[4 * i + r] * W[i];   // xEst = px * pw :106, index order (header)
  oracle_pf_calc_covariance(NP, xe, X, W, Pe);                        // :107
}

// resampling :120-148 for one vehicle.  Returns 1 if it resampled; anc (may be NULL) [NP]: the ancestor of each particle.
static int pf_resampling_one(int NP, float* X, float* W, const float* uni, float nth, int* anc) {
  std::vector<float> wcum(NP), base(NP), rid(NP), out((size_t)NP * 4);
  float ww = 0.0f;
  for (int i = 0; i < NP; ++i) ww += W[i] * W[i];                     // pw.transpose() * pw: index order here (header)
  float Neff = 1.0 / ww;                                              // :126
  if (anc) for (int i = 0; i < NP; ++i) anc[i] = i;
  if (!(Neff < nth)) return 0;                                        // :127
  wcum[0] = W[0];
  for (int i = 1; i < NP; ++i) wcum[i] = wcum[i - 1] + W[i];          // cumsum :111-118
  const float inv = (float)(1.0 / NP);                                // Ones()*1.0/NP
  float c = W[0] * 0.0f + inv;                                        // pw*0.0 + Ones*1.0/NP
  base[0] = c - inv;
  for (int i = 1; i < NP; ++i) { c = c + (W[i] * 0.0f + inv); base[i] = c - inv; }
  for (int j = 0; j < NP; ++j) rid[j] = base[j] + (double)uni[j] / NP;   // :133  uni_d(gen) is a double
  int ind = 0;
  for (int i = 0; i < NP; ++i) {                                      // :138-143
    while (rid[i] > wcum[ind] && ind < NP - 1) ind += 1;
    std::memcpy(&out[4 * (size_t)i], &X[4 * ind], 16);
    if (anc) anc[i] = ind;
  }
  std::memcpy(X, out.data(), sizeof(float) * 4 * NP);
  for (int i = 0; i < NP; ++i) W[i] = inv;                            // :146
  return 1;
}

// One vehicle-tick for agents [a0,a1): pf_localization (:73-109) then resampling (:120-148).
// px: [n][NP][4] (Eigen Matrix<float,4,NP> column-major = particle-major), pw: [n][NP], xEst [n][4], PEst [n][16] col-major,
// obs: [n][L][3] = (dn, landmark x, landmark y), nobs[n] <= L, u [n][2], nrm [n][NP][2], uni [n][NP],
// rsim[2] = (Rsim(0,0), Rsim(1,1)), Q, DT, nth = NP/2.  resampled[n] (may be NULL): 1 if the tick resampled.
// anc (may be NULL): [n][NP] ancestor index chosen for each particle (identity when not resampled).
// parts: 3 = the whole tick, 1 = pf_localization only, 2 = resampling only (px, pw as given).
void oracle_pf_step_parts(int n, int NP, int L, float* px, float* pw, float* xEst, float* PEst, const float* obs, const int* nobs,
                          const float* u, const float* nrm, const float* uni, const float* rsim, float Q, double DT, float nth,
                          int* resampled, int* anc, int a0, int a1, int parts) {
  for (int a = a0; a < a1; ++a) {
    float* X = px + (size_t)a * NP * 4;
    float* W = pw + (size_t)a * NP;
    if (parts & 1)
      pf_localization_one(NP, X, W, xEst + 4 * (size_t)a, PEst + 16 * (size_t)a, obs + (size_t)a * L * 3, nobs[a], u + 2 * (size_t)a,
                          nrm + (size_t)a * NP * 2, rsim, Q, DT);
    int did = 0;
    if (anc) for (int i = 0; i < NP; ++i) anc[(size_t)a * NP + i] = i;
    if (parts & 2) did = pf_resampling_one(NP, X, W, uni + (size_t)a * NP, nth, anc ? anc + (size_t)a * NP : nullptr);
    if (resampled) resampled[a] = did;
  }
}
void oracle_pf_step(int n, int NP, int L, float* px, float* pw, float* xEst, float* PEst, const float* obs, const int* nobs,
                    const float* u, const float* nrm, const float* uni, const float* rsim, float Q, double DT, float nth,
                    int* resampled, int* anc, int a0, int a1) {
  oracle_pf_step_parts(n, NP, L, px, pw, xEst, PEst, obs, nobs, u, nrm, uni, rsim, Q, DT, nth, resampled, anc, a0, a1, 3);
}

// ---- the same tick with the sums taken in the ENGINE's order ------------------------------------------------------------------
// The engine maps one vehicle to one 64-lane wavefront, particle ip on lane ip % 64 (two particles per lane above 64), and takes
// every sum over the particles as a balanced pairwise tree over the lanes (pf_kernels.hip.h: wave_sum), the cumulative sum as a
// lane scan (wave_scan_add), and finds each resampled ancestor by bisection of that cumulative sum followed by a running maximum.
// oracle_pf_step_wave restates exactly that arithmetic on the host — per-particle maths as above, the reductions below — so that
// the kernel can be demanded equal to it BIT FOR BIT; oracle_pf_step (index-order sums, the plain reading of the reference) is what
// both are compared with statistically.  NP <= 128.
namespace {
float tree_sum64(const float* v) {                 // wave_sum: adjacent pairs, level by level
  float t[64];
  std::memcpy(t, v, sizeof(t));
  for (int n = 64; n > 1; n >>= 1) for (int i = 0; i < n / 2; ++i) t[i] = t[2 * i] + t[2 * i + 1];
  return t[0];
}
void scan_add64(float* v) {                        // wave_scan_add: shifts 1,2,4,8 inside each row of 16, then row totals
  for (int s = 1; s <= 8; s <<= 1) {
    float o[64];
    std::memcpy(o, v, sizeof(o));
    for (int i = 0; i < 64; ++i) v[i] = o[i] + ((i & 15) >= s ? o[i - s] : 0.0f);
  }
  { float o[64]; std::memcpy(o, v, sizeof(o));     // rows 1, 3 += lane 15 of the row before
    for (int i = 0; i < 64; ++i) { const int row = i >> 4; v[i] = o[i] + ((row & 1) ? o[16 * row - 1] : 0.0f); } }
  { float o[64]; std::memcpy(o, v, sizeof(o));     // rows 2, 3 += lane 31
    for (int i = 0; i < 64; ++i) v[i] = o[i] + (i >= 32 ? o[31] : 0.0f); }
}
}  // namespace

void oracle_pf_step_wave(int n, int NP, int L, float* px, float* pw, float* xEst, float* PEst, const float* obs, const int* nobs,
                         const float* u, const float* nrm, const float* uni, const float* rsim, float Q, double DT, float nth,
                         int* resampled, int a0, int a1) {
  if (NP < 1 || NP > 128) return;
  const float inv = (float)(1.0 / NP);
  std::vector<float> base(NP), wcum(NP);
  { float c = inv; base[0] = c - inv; for (int i = 1; i < NP; ++i) { c = c + inv; base[i] = c - inv; } }
  for (int a = a0; a < a1; ++a) {
    float* X = px + (size_t)a * NP * 4;
    float* W = pw + (size_t)a * NP;
    const float* Z = obs + (size_t)a * L * 3;
    const float sig = std::sqrt(Q);
    const double lik_c = 1.0 / std::sqrt(2.0 * PI_ * sig * sig);
    const float lik_d = 2 * sig * sig;
    const int nob = nobs[a] < 0 ? 0 : (nobs[a] > L ? L : nobs[a]);
    float x[2][64][4], w[2][64];
    std::memset(x, 0, sizeof(x)); std::memset(w, 0, sizeof(w));
    for (int ip = 0; ip < NP; ++ip) {
      float* xp = x[ip >> 6][ip & 63];
      std::memcpy(xp, X + 4 * ip, 16);
      float wi = W[ip];
      float ud[2];
      ud[0] = u[2 * a] + (double)nrm[((size_t)a * NP + ip) * 2] * rsim[0];
      ud[1] = u[2 * a + 1] + (double)nrm[((size_t)a * NP + ip) * 2 + 1] * rsim[1];
      motion_model(xp, ud, DT);
      for (int i = 0; i < nob; ++i) {
        const float dx = xp[0] - Z[3 * i + 1], dy = xp[1] - Z[3 * i + 2];
        const float prez = std::sqrt(dx * dx + dy * dy);
        const float dz = prez - Z[3 * i];
        const float pl = (float)(lik_c * (double)std::exp(-dz * dz / lik_d));
        wi = wi * pl;
      }
      w[ip >> 6][ip & 63] = wi;
    }
    float t[64];
    for (int l = 0; l < 64; ++l) t[l] = w[0][l] + w[1][l];
    const float s = tree_sum64(t);
    for (int l = 0; l < 64; ++l) { w[0][l] = w[0][l] / s; w[1][l] = w[1][l] / s; }
    float xe[4];
    for (int r = 0; r < 4; ++r) {
      for (int l = 0; l < 64; ++l) t[l] = x[0][l][r] * w[0][l] + x[1][l][r] * w[1][l];
      xe[r] = tree_sum64(t);
    }
    float Pe[16];
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r <= c; ++r) {
        for (int l = 0; l < 64; ++l) {
          const float d0r = x[0][l][r] - xe[r], d0c = x[0][l][c] - xe[c], d1r = x[1][l][r] - xe[r], d1c = x[1][l][c] - xe[c];
          t[l] = (w[0][l] * d0r) * d0c + ((l + 64 < NP) ? (w[1][l] * d1r) * d1c : 0.0f);
        }
        Pe[r + 4 * c] = Pe[c + 4 * r] = tree_sum64(t);
      }
    std::memcpy(xEst + 4 * (size_t)a, xe, sizeof(xe));
    std::memcpy(PEst + 16 * (size_t)a, Pe, sizeof(Pe));
    for (int l = 0; l < 64; ++l) t[l] = w[0][l] * w[0][l] + w[1][l] * w[1][l];
    const float ww = tree_sum64(t);
    const float Neff = (float)(1.0 / (double)ww);
    int did = 0;
    if (Neff < nth) {
      did = 1;
      float c0[64], c1[64];
      std::memcpy(c0, w[0], sizeof(c0)); std::memcpy(c1, w[1], sizeof(c1));
      scan_add64(c0);
      const float tot0 = c0[63];
      scan_add64(c1);
      for (int l = 0; l < 64; ++l) c1[l] = c1[l] + tot0;
      for (int ip = 0; ip < NP; ++ip) wcum[ip] = ip < 64 ? c0[ip] : c1[ip - 64];
      int idx[2][64];
      std::memset(idx, 0, sizeof(idx));
      for (int ip = 0; ip < NP; ++ip) {
        const float rid = (float)((double)base[ip] + (double)uni[(size_t)a * NP + ip] / NP);
        int lo = 0, hi = NP - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rid > wcum[mid]) lo = mid + 1; else hi = mid; }
        idx[ip >> 6][ip & 63] = lo;
      }
      for (int l = 1; l < 64; ++l) if (idx[0][l - 1] > idx[0][l]) idx[0][l] = idx[0][l - 1];      // running maximum, particle order
      const int m0 = idx[0][63];
      for (int l = 1; l < 64; ++l) if (idx[1][l - 1] > idx[1][l]) idx[1][l] = idx[1][l - 1];
      for (int l = 0; l < 64; ++l) if (m0 > idx[1][l]) idx[1][l] = m0;
      for (int ip = 0; ip < NP; ++ip) {
        const int j = idx[ip >> 6][ip & 63];
        std::memcpy(X + 4 * ip, x[j >> 6][j & 63], 16);
        W[ip] = inv;
      }
    } else {
      for (int ip = 0; ip < NP; ++ip) { std::memcpy(X + 4 * ip, x[ip >> 6][ip & 63], 16); W[ip] = w[ip >> 6][ip & 63]; }
    }
    if (resampled) resampled[a] = did;
  }
}

// T ticks: u [T][n][2], obs [T][n][L][3], nobs [T][n], nrm [T][n][NP][2], uni [T][n][NP]; x_hist (may be NULL) [T][n][4].
void oracle_pf_run(int n, int NP, int L, int T, float* px, float* pw, float* xEst, float* PEst, const float* obs,
                   const int* nobs, const float* u, const float* nrm, const float* uni, const float* rsim, float Q,
                   double DT, float nth, float* x_hist, int* n_resampled, int a0, int a1) {
  std::vector<int> did(n);
  for (int t = 0; t < T; ++t) {
    oracle_pf_step(n, NP, L, px, pw, xEst, PEst, obs + (size_t)t * n * L * 3, nobs + (size_t)t * n, u + (size_t)t * n * 2,
                   nrm + (size_t)t * n * NP * 2, uni + (size_t)t * n * NP, rsim, Q, DT, nth, did.data(), nullptr, a0, a1);
    for (int a = a0; a < a1; ++a) {
      if (x_hist) std::memcpy(x_hist + ((size_t)t * n + a) * 4, xEst + 4 * (size_t)a, 16);
      if (n_resampled) n_resampled[a] += did[a];
    }
  }
}

// The observation side of main() (:251-268): ud, xTrue, xDR and the range observations of the landmarks within MAX_RANGE.
// w_u [T][n][2] normals for ud, w_z [T][n][L] normals for the range noise.  Outputs: ud [T][n][2], obs, nobs, xTrue_hist.
void oracle_pf_simulate_inputs(int n, int T, int L, const float* u_true, float* xTrue, float* xDR, const float* rfid,
                               const float* w_u, const float* w_z, const float* rsim, float Qsim, float max_range,
                               double DT, float* ud_out, float* obs, int* nobs, float* xTrue_hist, float* xDR_hist) {
  for (int a = 0; a < n; ++a) {
    float xt[4], xd[4];
    std::memcpy(xt, xTrue + 4 * (size_t)a, 16); std::memcpy(xd, xDR + 4 * (size_t)a, 16);
    const float* u = u_true + 2 * (size_t)a;
    for (int t = 0; t < T; ++t) {
      const size_t o = (size_t)t * n + a;
      float ud[2];
      ud[0] = u[0] + (double)w_u[2 * o] * rsim[0];
      ud[1] = u[1] + (double)w_u[2 * o + 1] * rsim[1];
      motion_model(xt, u, DT);
      motion_model(xd, ud, DT);
      int k = 0;
      for (int i = 0; i < L; ++i) {
        float dx = xt[0] - rfid[2 * i], dy = xt[1] - rfid[2 * i + 1];
        float d = std::sqrt(dx * dx + dy * dy);
        if (d <= max_range) {
          float dn = d + (double)w_z[o * L + i] * Qsim;
          float* z = obs + (o * L + k) * 3;
          z[0] = dn; z[1] = rfid[2 * i]; z[2] = rfid[2 * i + 1];
          ++k;
        }
      }
      for (int i = k; i < L; ++i) { float* z = obs + (o * L + i) * 3; z[0] = z[1] = z[2] = 0.0f; }
      nobs[o] = k;
      ud_out[2 * o] = ud[0]; ud_out[2 * o + 1] = ud[1];
      if (xTrue_hist) std::memcpy(xTrue_hist + 4 * o, xt, 16);
      if (xDR_hist) std::memcpy(xDR_hist + 4 * o, xd, 16);
    }
    std::memcpy(xTrue + 4 * (size_t)a, xt, 16); std::memcpy(xDR + 4 * (size_t)a, xd, 16);
  }
}

// oracle_pf_run with the engine's summation order (oracle_pf_step_wave) in every tick.
void oracle_pf_run_wave(int n, int NP, int L, int T, float* px, float* pw, float* xEst, float* PEst, const float* obs,
                        const int* nobs, const float* u, const float* nrm, const float* uni, const float* rsim, float Q,
                        double DT, float nth, float* x_hist, int* n_resampled, int a0, int a1) {
  std::vector<int> did(n);
  for (int t = 0; t < T; ++t) {
    oracle_pf_step_wave(n, NP, L, px, pw, xEst, PEst, obs + (size_t)t * n * L * 3, nobs + (size_t)t * n, u + (size_t)t * n * 2,
                        nrm + (size_t)t * n * NP * 2, uni + (size_t)t * n * NP, rsim, Q, DT, nth, did.data(), a0, a1);
    for (int a = a0; a < a1; ++a) {
      if (x_hist) std::memcpy(x_hist + ((size_t)t * n + a) * 4, xEst + 4 * (size_t)a, 16);
      if (n_resampled) n_resampled[a] += did[a];
    }
  }
}

}  // extern "C"

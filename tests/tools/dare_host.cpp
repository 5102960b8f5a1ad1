// dare_host.cpp — host build of the engine's structured Riccati iterations (csrc/dare_math.h), so that both register layouts —
// one lane per agent on packed rows, and four lanes per agent with DPP quad_perm exchanges (emulated here by Quad4f, four
// lanes in lockstep) — can be checked bit for bit against the oracle without a GPU.
// Built by tests/test_dare_host.py: g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC.  The loop around the iteration
// mirrors the kernels': cold start X = I, stop at max|Xn - X| < eps, at most maxiter evaluations.
#include <cmath>
#include <cstring>
#include "../../cpprobotics_amd/csrc/dare_math.h"
#include "../../cpprobotics_amd/csrc/dare_dense_math.h"

using namespace crx;

// per-agent free entries (A(0,1) = A(2,3) = dt, A(1,2) = v, B(3,0) = bv, B(4,1) = bd): what crx_dare_batch's structure detection hands to
// the same iteration (DareFromMats, csrc/dare_kernels.hip.h); dts / bvs / bds NULL = built from the speed as dare_*_run does
static int lane_core(int n, int dim, const float* vs, const float* dts, const float* bvs, const float* bds, double dt_d, double L, float eps,
                     int maxiter, float* Xout, int* iters) {
  for (int a = 0; a < n; ++a) {
    const float dt = dts ? dts[a] : (float)dt_d;
    const float v = vs[a], bv = bvs ? bvs[a] : (float)((double)v / L), bd = bds ? bds[a] : dt;
    Row4 X[4], Y[4];
    float x44 = 1.0f, y44 = 1.0f;
    for (int i = 0; i < 4; ++i) {
      X[i].a = d_v2f{i == 0 ? 1.0f : 0.0f, i == 1 ? 1.0f : 0.0f};
      X[i].b = d_v2f{i == 2 ? 1.0f : 0.0f, i == 3 ? 1.0f : 0.0f};
    }
    int it = maxiter < 0 ? 0 : maxiter;
    for (int i = 0; i < maxiter; ++i) {
      float m;
      if (dim == 5) { dare5_v_iter_pk(dt, v, bv, bd, X, x44, Y, y44); m = dare_max_abs_diff(Y, y44, X, x44); }
      else { dare4_v_iter_pk(dt, v, bv, X, Y); m = dare_max_abs_diff(Y, X); }
      std::memcpy(X, Y, sizeof(X)); x44 = y44;
      if (m < eps) { it = i + 1; break; }
    }
    float* Xa = Xout + (size_t)a * dim * dim;
    std::memset(Xa, 0, sizeof(float) * dim * dim);
    for (int i = 0; i < 4; ++i) {
      Xa[i + dim * 0] = X[i].a[0]; Xa[i + dim * 1] = X[i].a[1]; Xa[i + dim * 2] = X[i].b[0]; Xa[i + dim * 3] = X[i].b[1];
    }
    if (dim == 5) Xa[24] = x44;
    iters[a] = it;
  }
  return 0;
}

extern "C" int dare_lane_run(int n, int dim, const float* vs, double dt_d, double L, float eps, int maxiter, float* Xout, int* iters) {
  return lane_core(n, dim, vs, nullptr, nullptr, nullptr, dt_d, L, eps, maxiter, Xout, iters);
}
extern "C" int dare_lane_run_params(int n, int dim, const float* vs, const float* dts, const float* bvs, const float* bds, float eps, int maxiter,
                                    float* Xout, int* iters) {
  return lane_core(n, dim, vs, dts, bvs, bds, 0.0, 1.0, eps, maxiter, Xout, iters);
}

static int quad_core(int n, int dim, const float* vs, const float* dts, const float* bvs, const float* bds, double dt_d, double L, float eps,
                     int maxiter, float* Xout, int* iters) {
  for (int a = 0; a < n; ++a) {
    const float dt = dts ? dts[a] : (float)dt_d;
    const float v = vs[a];
    QuadLane<Quad4f, Quad4u> c;
    c.dt = dt; c.v = v; c.bd = bds ? bds[a] : dt; c.bv = bvs ? bvs[a] : (float)((double)v / L);
    c.a = Quad4f(1.0f, dt, v, dt);
    c.m2 = Quad4u{{0u, 0u, 0xffffffffu, 0u}};
    for (int j = 0; j < 4; ++j) { c.q[j] = Quad4f(0.0f); c.q[j].l[j] = 1.0f; }
    Quad4f X[4], Y[4], x44(1.0f), y44(1.0f);
    for (int j = 0; j < 4; ++j) X[j] = c.q[j];
    int it = maxiter < 0 ? 0 : maxiter;
    for (int i = 0; i < maxiter; ++i) {
      Quad4f m;
      if (dim == 5) m = dare5_quad_iter(c, X, x44, Y, y44);
      else m = dare4_quad_iter(c, X, Y);
      for (int j = 0; j < 4; ++j) X[j] = Y[j];
      x44 = y44;
      for (int l = 1; l < 4; ++l)   // every lane of the quad must see the same maximum (NaN included)
        if (std::memcmp(&m.l[0], &m.l[l], 4) != 0 && !(m.l[0] != m.l[0] && m.l[l] != m.l[l])) return 1;
      if (m.l[0] < eps) { it = i + 1; break; }
    }
    float* Xa = Xout + (size_t)a * dim * dim;
    std::memset(Xa, 0, sizeof(float) * dim * dim);
    for (int r = 0; r < 4; ++r)
      for (int j = 0; j < 4; ++j) Xa[r + dim * j] = X[j].l[r];
    if (dim == 5) Xa[24] = x44.l[3];
    iters[a] = it;
  }
  return 0;
}
extern "C" int dare_quad_run(int n, int dim, const float* vs, double dt_d, double L, float eps, int maxiter, float* Xout, int* iters) {
  return quad_core(n, dim, vs, nullptr, nullptr, nullptr, dt_d, L, eps, maxiter, Xout, iters);
}
extern "C" int dare_quad_run_params(int n, int dim, const float* vs, const float* dts, const float* bvs, const float* bds, float eps, int maxiter,
                                    float* Xout, int* iters) {
  return quad_core(n, dim, vs, dts, bvs, bds, 0.0, 1.0, eps, maxiter, Xout, iters);
}

// The dense iteration in its four-lanes-per-agent form (dare_dense_quad_rows): lane r's code is run for r = 0..3 on the same gathered
// X, the rows are assembled, and the loop of solve_DARE goes round as in the kernel.  Returns 1 if the lanes disagree on row 4.
extern "C" int dare_dense_quad_run(int n, int dim, const float* Ag, const float* Bg, const float* Qg, const float* Rg, float eps, int maxiter,
                                   float* Xout, int* iters) {
  const int NN = dim * dim, M = dim == 5 ? 2 : 1;
  for (int a = 0; a < n; ++a) {
    const float *A = Ag + (size_t)a * NN, *B = Bg + (size_t)a * dim * M, *Q = Qg + (size_t)a * NN, *R = Rg + (size_t)a * M * M;
    float X[25], Xn[25];
    std::memcpy(X, Q, sizeof(float) * NN);
    int it = maxiter < 0 ? 0 : maxiter;
    for (int i = 0; i < maxiter; ++i) {
      float row4_first[5];
      for (int r = 0; r < 4; ++r) {
        float Acol_r[5], Acol_4[5], Qrow_r[5], Qrow_4[5], xr[5], x4[5];
        for (int k = 0; k < dim; ++k) { Acol_r[k] = A[k + dim * r]; Acol_4[k] = dim == 5 ? A[k + dim * 4] : 0.0f; }
        for (int j = 0; j < dim; ++j) { Qrow_r[j] = Q[r + dim * j]; Qrow_4[j] = dim == 5 ? Q[4 + dim * j] : 0.0f; }
        if (dim == 5) dare_dense_quad_rows<5>(Acol_r, Acol_4, A, B, Qrow_r, Qrow_4, R, X, xr, x4);
        else dare_dense_quad_rows<4>(Acol_r, Acol_4, A, B, Qrow_r, Qrow_4, R, X, xr, x4);
        for (int j = 0; j < dim; ++j) Xn[r + dim * j] = xr[j];
        if (dim == 5) {
          if (r == 0) std::memcpy(row4_first, x4, sizeof(x4));
          else if (std::memcmp(row4_first, x4, sizeof(x4)) != 0) return 1;
          for (int j = 0; j < 5; ++j) Xn[4 + 5 * j] = x4[j];
        }
      }
      // (Xn - X).cwiseAbs().maxCoeff() with the reference's NaN rule (a NaN first element sticks, later NaNs are skipped)
      float m = std::fabs(Xn[0] - X[0]);
      const bool first_nan = m != m;
      for (int e = 1; e < NN; ++e) { const float d = std::fabs(Xn[e] - X[e]); if (d > m) m = d; }
      std::memcpy(X, Xn, sizeof(float) * NN);
      if (!first_nan && m < eps) { it = i + 1; break; }
    }
    std::memcpy(Xout + (size_t)a * NN, X, sizeof(float) * NN);
    iters[a] = it;
  }
  return 0;
}

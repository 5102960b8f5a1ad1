// dare_kernels.hip.h — batched discrete Riccati fixed point + LQR gain for gfx950
// (one agent per lane, X and all temporaries in VGPRs).
//
// Replaces, for n independent agents at once,
//   solve_DARE / dlqr  5x5, 2 inputs: /root/reference/src/lqr_speed_steer_control.cpp:85-106
//   solve_DARE / dlqr  4x4, 1 input : /root/reference/src/lqr_steer_control.cpp:75-96
//
// Two kernel families:
//   dare_dense_kernel<DIM>   — arbitrary per-agent A, B, Q, R (the drop-in for solve_DARE()
//                              called with any matrices).  Products are accumulated in the
//                              order Eigen 3.3.9 uses for that shape (DESIGN.md "Eigen order").
//   dare_from_v_kernel<DIM>  — A, B built from the speed exactly as lqr_steering_control()
//                              builds them (:116-129 / :104-115), Q = I, R = I; the literal
//                              0/1 entries are skipped, which leaves at most two non-zero terms
//                              in every sum, so the result is bit-identical to the dense
//                              evaluation at ~1/5 of the flops.
//
// Semantics kept from the reference: cold start X = Q; stop as soon as max|Xn - X| < eps and
// return Xn; otherwise X = Xn and, after `maxiter` evaluations, return X — which is again the
// last evaluated Xn, so the value handed back is always the most recent iterate; no fma
// contraction; one IEEE division per 2x2 inverse.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "dare_math.h"
#include "dare_dense_math.h"

namespace crx {

// (Xn - X).cwiseAbs().maxCoeff(): a strict '>' scan from element 0 in which a NaN element never replaces the
// running maximum, and a NaN FIRST element sticks.  v_max_f32 has exactly that behaviour for every element but the
// first (max(m, NaN) = m), so the scan is one subtract + one max per element, with the first-element NaN patched in.
template <int N>
__device__ __forceinline__ float max_abs_diff(const float* a, const float* b) {
  const float m0 = fabsf(a[0] - b[0]);
  float m = m0;
#pragma unroll
  for (int i = 1; i < N; ++i) m = fmaxf(m, fabsf(a[i] - b[i]));
  return (m0 != m0) ? m0 : m;
}

// The reference's fixed-point loop (solve_DARE :86-99): X = Q; repeat Xn = f(X); stop when max|Xn - X| < eps (return Xn)
// else X = Xn; after maxiter evaluations return the last one.  Either way the value handed back is the most recent
// iterate.  Written as a two-buffer ping-pong (X -> Y -> X ...) so that no 16/25-register copy sits in the loop; lanes
// that have converged are masked off (their buffers are no longer written) while the wave finishes its slowest agent.
// X holds the start value on entry and the result on return; returns the number of evaluations performed.
template <int NN, class IterFn>
__device__ __forceinline__ int riccati_fixed_point(float* X, float eps, int maxiter, bool live, IterFn iter) {
  float Y[NN];
  bool done = !live || maxiter <= 0;
  bool in_y = false;                      // which buffer holds this lane's most recent iterate
  int it = maxiter < 0 ? 0 : maxiter;
  for (int i = 0; i < maxiter; i += 2) {
    if (!done) {
      iter(X, Y);
      in_y = true;
      if (max_abs_diff<NN>(Y, X) < eps) { done = true; it = i + 1; }
    }
    if (!done && i + 1 < maxiter) {
      iter(Y, X);
      in_y = false;
      if (max_abs_diff<NN>(X, Y) < eps) { done = true; it = i + 2; }
    }
    if (__all(done)) break;
  }
  if (in_y) {
#pragma unroll
    for (int j = 0; j < NN; ++j) X[j] = Y[j];
  }
  return it;
}

template <int DIM, int PARTS>
__device__ __forceinline__ bool dare_pattern_part_ok(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Q,
                                                     const float* __restrict__ R, int part);
template <int DIM>
__device__ __forceinline__ void dare_pattern_params(const float* __restrict__ A, const float* __restrict__ B, bool& ok, float& dt, float& v,
                                                    float& bv, float& bd);

// SKIP_STRUCTURED: the agents whose arguments carry lqr_steering_control's pattern have been solved by a structured kernel of
// this launch pair (DareFromMats below, the same predicate); a wave none of whose agents is left returns at once.
template <int DIM, bool SKIP_STRUCTURED>
__global__ void __launch_bounds__(64)
dare_dense_kernel(int n, const float* __restrict__ Ag, const float* __restrict__ Bg,
                  const float* __restrict__ Qg, const float* __restrict__ Rg, float eps, int maxiter,
                  float* __restrict__ Xg, float* __restrict__ Kg, int* __restrict__ iters) {
  constexpr int NN = DIM * DIM;
  constexpr int M = (DIM == 5) ? 2 : 1;
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool live = a < (size_t)n;
  const size_t ai = live ? a : 0;
  if constexpr (SKIP_STRUCTURED) {
    bool st = dare_pattern_part_ok<DIM, 1>(Ag + ai * NN, Bg + ai * DIM * M, Qg + ai * NN, Rg + ai * M * M, 0);
    float p0, p1, p2, p3;
    dare_pattern_params<DIM>(Ag + ai * NN, Bg + ai * DIM * M, st, p0, p1, p2, p3);
    live = live && !st;
    if (!__builtin_amdgcn_ballot_w64(live)) return;
  }
  float A[NN], B[DIM * M], Q[NN], R[M * M], X[NN];
#pragma unroll
  for (int i = 0; i < NN; ++i) { A[i] = Ag[ai * NN + i]; Q[i] = Qg[ai * NN + i]; X[i] = Q[i]; }
#pragma unroll
  for (int i = 0; i < DIM * M; ++i) B[i] = Bg[ai * DIM * M + i];
#pragma unroll
  for (int i = 0; i < M * M; ++i) R[i] = Rg[ai * M * M + i];

  const int it = riccati_fixed_point<NN>(X, eps, maxiter, live, [&](const float* Xi, float* Xo) {
    if (DIM == 5) dare5_dense_iter(A, B, Q, R, Xi, Xo);
    else dare4_dense_iter(A, B, Q, R[0], Xi, Xo);
  });
  if (!live) return;
  if (Xg) {
#pragma unroll
    for (int j = 0; j < NN; ++j) Xg[a * NN + j] = X[j];
  }
  if (Kg) {
    float K[M * DIM];
    if (DIM == 5) dlqr5_dense_gain(A, B, R, X, K);
    else dlqr4_dense_gain(A, B, R[0], X, K);
#pragma unroll
    for (int j = 0; j < M * DIM; ++j) Kg[a * M * DIM + j] = K[j];
  }
  if (iters) iters[a] = it;
}

// ---------- dense, four lanes per agent (round 4) ----------------------------------------------------------------------------
// BASELINE-sized batches of DENSE problems are the same latency chain on a quarter of the chip as the structured ones were (256
// one-lane waves on 1,024 SIMDs, ~1,020 instructions per evaluation issued by one wave).  Here an agent is a quad: lane r holds row
// r of X and (5x5) a copy of row 4, gathers the other rows once per evaluation as DPP quad_perm broadcasts and runs
// dare_dense_quad_rows (dare_dense_math.h) — its own row and row 4 of every product of :91, 2/5 of the agent's evaluation (1/4 for
// 4x4) — so the batch fills every SIMD and an evaluation is ~2x shorter (5x5: 256 VGPRs + 26 AGPRs, one wave per SIMD; capped at
// 256 registers for two waves per SIMD it spills 96 B and gains nothing at 32,768 agents: two waves share one VALU).  Same accumulation order per coefficient, same bits
// (tests/test_dare_host.py runs the lane code on the CPU against the oracle; tests/test_lqr_gpu.py the kernel).
template <int DIM, bool SKIP_STRUCTURED>
__global__ void __launch_bounds__(256)
dare_dense_quad_kernel(int n, const float* __restrict__ Ag, const float* __restrict__ Bg, const float* __restrict__ Qg,
                       const float* __restrict__ Rg, float eps, int maxiter, float* __restrict__ Xg, float* __restrict__ Kg,
                       int* __restrict__ iters) {
  constexpr int NN = DIM * DIM;
  constexpr int M = (DIM == 5) ? 2 : 1;
  constexpr int NB = DIM * M;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t a = t >> 2;
  const int r = (int)(threadIdx.x & 3);
  bool live = a < (size_t)n;
  const size_t ai = live ? a : 0;
  const float* __restrict__ Aa = Ag + ai * NN;
  const float* __restrict__ Ba = Bg + ai * NB;
  const float* __restrict__ Qa = Qg + ai * NN;
  const float* __restrict__ Ra = Rg + ai * M * M;
  if constexpr (SKIP_STRUCTURED) {                            // the agents a structured kernel of this launch pair has served
    bool st = live && dare_pattern_part_ok<DIM, 4>(Aa, Ba, Qa, Ra, r);
    float p0, p1, p2, p3;
    dare_pattern_params<DIM>(Aa, Ba, st, p0, p1, p2, p3);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(st);
    st = ((m >> (threadIdx.x & 60)) & 0xFull) == 0xFull;
    live = live && !st;
    if (!__builtin_amdgcn_ballot_w64(live)) return;
  }
  float A[NN], B[NB], R[M * M], Acol_r[DIM], Acol_4[DIM], Qrow_r[DIM], Qrow_4[DIM], xr[DIM], x4[DIM];
#pragma unroll
  for (int i = 0; i < NN; ++i) A[i] = Aa[i];
#pragma unroll
  for (int i = 0; i < NB; ++i) B[i] = Ba[i];
#pragma unroll
  for (int i = 0; i < M * M; ++i) R[i] = Ra[i];
#pragma unroll
  for (int k = 0; k < DIM; ++k) {                             // this lane's column of A and row of Q, addressed by r in memory
    Acol_r[k] = Aa[k + DIM * r]; Acol_4[k] = A[k + DIM * (DIM - 1)];
    Qrow_r[k] = Qa[r + DIM * k]; Qrow_4[k] = Qa[(DIM - 1) + DIM * k];
    xr[k] = Qrow_r[k]; x4[k] = Qrow_4[k];                     // X = Q
  }
  // the whole X, column-major, from the quad's rows (rows 0-3: one DPP broadcast each; row 4: every lane's own copy)
  auto gather = [&](float* Xf) {
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      Xf[0 + DIM * j] = qperm<QP_0000>(xr[j]);
      Xf[1 + DIM * j] = qperm<0x55>(xr[j]);
      Xf[2 + DIM * j] = qperm<0xAA>(xr[j]);
      Xf[3 + DIM * j] = qperm<QP_3333>(xr[j]);
      if constexpr (DIM == 5) Xf[4 + DIM * j] = x4[j];
    }
  };
  bool done = !live || maxiter <= 0;
  int it = maxiter < 0 ? 0 : maxiter;
  for (int i = 0; i < maxiter; ++i) {
    if (!done) {                                              // (a quad's four lanes take every decision together)
      float Xf[NN], yr[DIM], y4[DIM];
      gather(Xf);
      dare_dense_quad_rows<DIM>(Acol_r, Acol_4, A, B, Qrow_r, Qrow_4, R, Xf, yr, y4);
      // (Xn - X).cwiseAbs().maxCoeff() with the reference's NaN rule: fmaxf drops a NaN operand — every element but the first
      // behaves so in the reference's strict-'>' scan — and element (0,0), lane 0's, makes the result NaN when it is NaN or infinite
      // (dare_quad_first: the maximum is not below eps then either)
      const float d00 = yr[0] - xr[0];
      float m = fabsf(d00);
#pragma unroll
      for (int j = 1; j < DIM; ++j) m = fmaxf(m, fabsf(yr[j] - xr[j]));
      if constexpr (DIM == 5) {
#pragma unroll
        for (int j = 0; j < DIM; ++j) m = fmaxf(m, fabsf(y4[j] - x4[j]));
      }
      m = fmaxf(m, qperm<QP_1032>(m));
      m = fmaxf(m, qperm<QP_2301>(m));
      m += qperm<QP_0000>(d00 - d00);
#pragma unroll
      for (int j = 0; j < DIM; ++j) { xr[j] = yr[j]; x4[j] = y4[j]; }
      if (m < eps) { done = true; it = i + 1; }
    }
    if (__all(done)) break;
  }
  float Xf[NN];
  gather(Xf);                                                 // for the gain (every lane takes part in the broadcasts)
  if (!live) return;
  if (Xg) {
    float* Xa = Xg + a * NN;
#pragma unroll
    for (int j = 0; j < DIM; ++j) Xa[r + DIM * j] = xr[j];
    if constexpr (DIM == 5) {
      if (r == 0) {
#pragma unroll
        for (int j = 0; j < DIM; ++j) Xa[4 + DIM * j] = x4[j];
      }
    }
  }
  if (r == 0) {
    if (Kg) {
      float K[M * DIM];
      if constexpr (DIM == 5) dlqr5_dense_gain(A, B, R, Xf, K);
      else dlqr4_dense_gain(A, B, R[0], Xf, K);
#pragma unroll
      for (int j = 0; j < M * DIM; ++j) Kg[a * M * DIM + j] = K[j];
    }
    if (iters) iters[a] = it;
  }
}

// ---------- structured: A, B from v; Q = I; R = I ----------------------------------------------
// 5x5 (:116-129): A00=1 A01=dt A12=v A22=1 A23=dt A44=1 ; B30=v/L B41=dt.  The iteration itself is dare5_v_iter_pk below.
__device__ __forceinline__ void dlqr5_v_gain(float dt, float v, float bv, float bd, const float* X,
                                             float* K) {
  float BtX[10];
#pragma unroll
  for (int j = 0; j < 5; ++j) { BtX[0 + 2 * j] = bv * X[3 + 5 * j]; BtX[1 + 2 * j] = bd * X[4 + 5 * j]; }
  float Sg[4] = {BtX[0 + 2 * 3] * bv + 1.0f, BtX[1 + 2 * 3] * bv + 0.0f,
                 BtX[0 + 2 * 4] * bd + 0.0f, BtX[1 + 2 * 4] * bd + 1.0f}, Si[4];
  inverse2(Sg, Si);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float a0 = BtX[i], a1 = BtX[i] * dt, a2 = BtX[i + 2] * v + BtX[i + 4], a3 = BtX[i + 4] * dt,
                a4 = BtX[i + 8];
    BtX[i] = a0; BtX[i + 2] = a1; BtX[i + 4] = a2; BtX[i + 6] = a3; BtX[i + 8] = a4;  // now (B'X)A
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    K[0 + 2 * j] = Si[0] * BtX[0 + 2 * j] + Si[2] * BtX[1 + 2 * j];
    K[1 + 2 * j] = Si[1] * BtX[0 + 2 * j] + Si[3] * BtX[1 + 2 * j];
  }
}

// 4x4 (:104-115): A00=1 A01=dt A12=v A22=1 A23=dt ; B3=v/L ; R=1.  The iteration itself is dare4_v_iter_pk below.
__device__ __forceinline__ void dlqr4_v_gain(float dt, float v, float bv, const float* X, float* K) {
  const float b0 = bv * X[3 + 0], b1 = bv * X[3 + 4], b2 = bv * X[3 + 8], b3 = bv * X[3 + 12];
  const float g = b3 * bv;
  const float inv = (float)(1.0 / (double)(g + 1.0f));
  K[0] = inv * b0;
  K[1] = inv * (b0 * dt);
  K[2] = inv * (b2 + b1 * v);
  K[3] = inv * (b2 * dt);
}

// ---------- structured iterations ------------------------------------------------------------------------------------------
// dare_math.h: the iteration for A, B built from v, Q = I, R = I with the literal 0/1 structure of A and B skipped and, for the
// 5x5 problem, its block-diagonal iterate X = diag(X4, x44) (at most two non-zero terms per sum, so every coefficient equals
// the dense Eigen-order evaluation bit for bit: tests/test_lqr_gpu.py::test_dare_dense_matches_structured_and_oracle,
// tests/test_dare_host.py), in two layouts: one lane per agent on packed rows, and four lanes per agent (a DPP quad).
typedef float v2f __attribute__((ext_vector_type(2)));

// The reference's loop (see riccati_fixed_point) for A, B built from v, Q = I, R = I, one agent per lane on packed rows.  Xcm
// receives the result column-major like every other X of this file; returns the number of evaluations.
template <int DIM>
__device__ __forceinline__ int riccati_from_v(float dt, float v, float bv, float bd, float eps, int maxiter, bool live, float* Xcm) {
  Row4 X[4], Y[4];
  float x44 = 1.0f, y44 = 1.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    X[i].a = (v2f){i == 0 ? 1.0f : 0.0f, i == 1 ? 1.0f : 0.0f};
    X[i].b = (v2f){i == 2 ? 1.0f : 0.0f, i == 3 ? 1.0f : 0.0f};
  }
  auto iter = [&](const Row4* Xi, const float& xi44, Row4* Xo, float& xo44) -> float {
    if constexpr (DIM == 5) { dare5_v_iter_pk(dt, v, bv, bd, Xi, xi44, Xo, xo44); return dare_max_abs_diff(Xo, xo44, Xi, xi44); }
    else { dare4_v_iter_pk(dt, v, bv, Xi, Xo); return dare_max_abs_diff(Xo, Xi); }
  };
  bool done = !live || maxiter <= 0;
  bool in_y = false;
  int it = maxiter < 0 ? 0 : maxiter;
  for (int i = 0; i < maxiter; i += 2) {
    if (!done) {
      const float m = iter(X, x44, Y, y44);
      in_y = true;
      if (m < eps) { done = true; it = i + 1; }
    }
    if (!done && i + 1 < maxiter) {
      const float m = iter(Y, y44, X, x44);
      in_y = false;
      if (m < eps) { done = true; it = i + 2; }
    }
    if (__all(done)) break;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const Row4& r = in_y ? Y[i] : X[i];
    Xcm[i + DIM * 0] = r.a.x; Xcm[i + DIM * 1] = r.a.y; Xcm[i + DIM * 2] = r.b.x; Xcm[i + DIM * 3] = r.b.y;
    if constexpr (DIM == 5) { Xcm[i + DIM * 4] = 0.0f; Xcm[4 + DIM * i] = 0.0f; }
  }
  if constexpr (DIM == 5) Xcm[24] = in_y ? y44 : x44;
  return it;
}

// ---------- where the structured kernels take their problem from ------------------------------------------------------------------
// DareFromV: the speed (crx_dare_from_v_batch): A, B as lqr_steering_control builds them, bd = dt.
// DareFromMats: the dense arguments of crx_dare_batch — solve_DARE(A, B, Q, R) as the reference's signature hands them over
// (src/lqr_speed_steer_control.cpp:85, src/lqr_steer_control.cpp:75) — WHEN they carry the pattern lqr_steering_control builds
// (:116-129 / :104-115): every literal 0 a +0.0f, every literal 1 a 1.0f, Q = I, R = I, A(0,1) = A(2,3) bit for bit; the four free
// entries A(0,1) = dt, A(1,2) = v, B(3,0) = bv, B(4,1) = bd may be anything finite inside a generous box (below).  Such an agent
// is solved by the structured iteration — bit-identical to the dense Eigen-order evaluation while every intermediate is finite
// (header; tests/test_lqr_gpu.py) — every other agent by dare_dense_kernel, which applies the same predicate and skips the
// agents this one served.  The box keeps the iterates far from overflow (|v| <= 100, 1e-3 <= |dt|, |bd| <= 1, |bv| <= 1e3: the
// largest entry of any iterate over the box is below 1e11; tests/test_dare_host.py sweeps it, corners included), so "finite" need not be checked afterwards;
// anything outside it — a dense matrix, a NaN, a vehicle at 500 m/s — takes the dense kernel and its exact inf/NaN semantics.
template <int DIM, int PARTS>
__device__ __forceinline__ bool dare_pattern_part_ok(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Q,
                                                     const float* __restrict__ R, int part) {
  constexpr int NN = DIM * DIM, NB = (DIM == 5) ? 10 : 4, NR = (DIM == 5) ? 4 : 1, TOT = NN + NB + NN + NR;
  constexpr uint32_t ONE = 0x3f800000u;
  bool ok = true;
  for (int e = part; e < TOT; e += PARTS) {
    uint32_t got, want = 0u;
    bool free_entry = false;
    if (e < NN) {                                            // A, column-major: e = i + DIM j
      got = __float_as_uint(A[e]);
      const int i = e % DIM, j = e / DIM;
      if (i == j) want = (i == 0 || i == 2 || i == 4) ? ONE : 0u;
      free_entry = (i == 0 && j == 1) || (i == 1 && j == 2) || (i == 2 && j == 3);
    } else if (e < NN + NB) {                                // B: 5x2 (B30, B41 free) or a 4-vector (B3 free)
      const int b = e - NN;
      got = __float_as_uint(B[b]);
      free_entry = (DIM == 5) ? (b == 3 || b == 4 + 5) : (b == 3);
    } else if (e < NN + NB + NN) {                           // Q = I
      const int q = e - NN - NB;
      got = __float_as_uint(Q[q]);
      want = (q % DIM == q / DIM) ? ONE : 0u;
    } else {                                                 // R = I (2x2) or 1
      const int r = e - NN - NB - NN;
      got = __float_as_uint(R[r]);
      want = (DIM == 4 || r == 0 || r == 3) ? ONE : 0u;
    }
    ok = ok && (free_entry || got == want);
  }
  return ok;
}
// the free entries and the box; `ok` only ever gets cleared
template <int DIM>
__device__ __forceinline__ void dare_pattern_params(const float* __restrict__ A, const float* __restrict__ B, bool& ok, float& dt, float& v,
                                                    float& bv, float& bd) {
  dt = A[0 + DIM * 1]; v = A[1 + DIM * 2]; bv = B[3];
  const float dt2 = A[2 + DIM * 3];
  bd = (DIM == 5) ? B[4 + 5] : dt;
  const float adt = fabsf(dt), abd = fabsf(bd);
  ok = ok && __float_as_uint(dt) == __float_as_uint(dt2) && fabsf(v) <= 100.0f && fabsf(bv) <= 1e3f && adt >= 1e-3f && adt <= 1.0f &&
       abd >= 1e-3f && abd <= 1.0f;                         // (a NaN fails every comparison)
}

struct DareFromV {
  const float* v; float dt; double L;
  // PARTS lanes share an agent (1 or 4); -> does this kernel solve agent a?
  template <int DIM, int PARTS>
  __device__ __forceinline__ bool load(size_t a, int part, bool live, float& dt_, float& v_, float& bv, float& bd) const {
    v_ = live ? v[a] : 1.0f;
    dt_ = dt; bd = dt;
    bv = (float)((double)v_ / L);  // B(3,0) = state.v / L  (float / double literal)
    return live;
  }
};
struct DareFromMats {
  const float *A, *B, *Q, *R;
  template <int DIM, int PARTS>
  __device__ __forceinline__ bool load(size_t a, int part, bool live, float& dt_, float& v_, float& bv, float& bd) const {
    constexpr int NN = DIM * DIM, NB = (DIM == 5) ? 10 : 4, NR = (DIM == 5) ? 4 : 1;
    const size_t ai = live ? a : 0;
    bool ok = live && dare_pattern_part_ok<DIM, PARTS>(A + ai * NN, B + ai * NB, Q + ai * NN, R + ai * NR, part);
    dare_pattern_params<DIM>(A + ai * NN, B + ai * NB, ok, dt_, v_, bv, bd);
    if constexpr (PARTS == 4) {                              // the agent's four lanes checked a quarter of the entries each
      const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
      ok = ((m >> (threadIdx.x & 60)) & 0xFull) == 0xFull;
    }
    if (!ok) { dt_ = 0.1f; v_ = 1.0f; bv = 2.0f; bd = 0.1f; }   // benign values for the lanes that idle through the loop
    return ok;
  }
};

// The masked loop (riccati_from_v: 60 VGPRs, eight waves per SIMD) for batches that queue several waves on every SIMD: there the
// converged lanes' exec regions cost nothing that another wave does not hide, and occupancy is what counts (the emit variant
// below needs 89 VGPRs: measured 10 % slower at 1 M agents, 10-28 % faster up to 65,536).
template <int DIM, class Src>
__global__ void __launch_bounds__(64)
dare_from_v_masked_kernel(int n, const Src src, float eps, int maxiter,
                   float* __restrict__ Xg, float* __restrict__ Kg, int* __restrict__ iters) {
  constexpr int NN = DIM * DIM;
  constexpr int M = (DIM == 5) ? 2 : 1;
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float dt, v, bv, bd;
  const bool live = src.template load<DIM, 1>(a, 0, a < (size_t)n, dt, v, bv, bd);
  if (!__builtin_amdgcn_ballot_w64(live)) return;
  float X[NN];
  const int it = riccati_from_v<DIM>(dt, v, bv, bd, eps, maxiter, live, X);
  if (!live) return;
  if (Xg) {
#pragma unroll
    for (int j = 0; j < NN; ++j) Xg[a * NN + j] = X[j];
  }
  if (Kg) {
    float K[M * DIM];
    if (DIM == 5) dlqr5_v_gain(dt, v, bv, bd, X, K);
    else dlqr4_v_gain(dt, v, bv, X, K);
#pragma unroll
    for (int j = 0; j < M * DIM; ++j) Kg[a * M * DIM + j] = K[j];
  }
  if (iters) iters[a] = it;
}

typedef unsigned long long dare_mask_t;

// The same loop with nobody masked off (as in the four-lane kernel below): every lane keeps evaluating, two evaluations per trip behind
// one not-taken branch, and emit(who, W, w44, it) hands the agents of `who` back from the — rare — pass in which their test succeeds
// (W = the iterate after `it` evaluations).  Saves the two exec-mask regions per evaluation of riccati_from_v (~14 scalar
// instructions) and lets an evaluation's test overlap the next one's start; used where the result can go straight to memory.
template <int DIM, class Emit>
__device__ __forceinline__ void riccati_from_v_emit(float dt, float v, float bv, float bd, float eps, int maxiter, dare_mask_t todo, Emit&& emit) {
  Row4 X[4], Y[4];
  float x44 = 1.0f, y44 = 1.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    X[i].a = (v2f){i == 0 ? 1.0f : 0.0f, i == 1 ? 1.0f : 0.0f};
    X[i].b = (v2f){i == 2 ? 1.0f : 0.0f, i == 3 ? 1.0f : 0.0f};
  }
  auto iter = [&](const Row4* Xi, const float& xi44, Row4* Xo, float& xo44) -> float {
    if constexpr (DIM == 5) { dare5_v_iter_pk(dt, v, bv, bd, Xi, xi44, Xo, xo44); return dare_max_abs_diff(Xo, xo44, Xi, xi44); }
    else { dare4_v_iter_pk(dt, v, bv, Xi, Xo); return dare_max_abs_diff(Xo, Xi); }
  };
  if (maxiter <= 0) { emit(todo, X, x44, 0); return; }
  int i = 0;
  if (maxiter & 1) {
    const float m = iter(X, x44, Y, y44);
    const dare_mask_t hit = __builtin_amdgcn_ballot_w64(m < eps) & todo;
    if (hit) { emit(hit, Y, y44, 1); todo &= ~hit; }
#pragma unroll
    for (int j = 0; j < 4; ++j) X[j] = Y[j];
    x44 = y44;
    i = 1;
  }
  for (; i < maxiter && todo; i += 2) {
    const float m1 = iter(X, x44, Y, y44);
    const float m2 = iter(Y, y44, X, x44);
    const dare_mask_t hit1 = __builtin_amdgcn_ballot_w64(m1 < eps) & todo;
    const dare_mask_t hit2 = __builtin_amdgcn_ballot_w64(m2 < eps) & todo & ~hit1;
    if (hit1 | hit2) {
      if (hit1) emit(hit1, Y, y44, i + 1);
      if (hit2) emit(hit2, X, x44, i + 2);
      todo &= ~(hit1 | hit2);
    }
  }
  if (todo) emit(todo, X, x44, maxiter);
}

template <int DIM, class Src>
__global__ void __launch_bounds__(64)
dare_from_v_kernel(int n, const Src src, float eps, int maxiter,
                   float* __restrict__ Xg, float* __restrict__ Kg, int* __restrict__ iters) {
  constexpr int NN = DIM * DIM;
  constexpr int M = (DIM == 5) ? 2 : 1;
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float dt, v, bv, bd;
  const bool live = src.template load<DIM, 1>(a, 0, a < (size_t)n, dt, v, bv, bd);
  if (!__builtin_amdgcn_ballot_w64(live)) return;
  auto emit = [&](dare_mask_t who, const Row4* W, float w44, int it) {
    if (!((who >> (threadIdx.x & 63)) & 1)) return;
    float X[NN];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      X[i + DIM * 0] = W[i].a.x; X[i + DIM * 1] = W[i].a.y; X[i + DIM * 2] = W[i].b.x; X[i + DIM * 3] = W[i].b.y;
      if constexpr (DIM == 5) { X[i + DIM * 4] = 0.0f; X[4 + DIM * i] = 0.0f; }
    }
    if constexpr (DIM == 5) X[24] = w44;
    if (Xg) {
#pragma unroll
      for (int j = 0; j < NN; ++j) Xg[a * NN + j] = X[j];
    }
    if (Kg) {
      float K[M * DIM];
      if (DIM == 5) dlqr5_v_gain(dt, v, bv, bd, X, K);
      else dlqr4_v_gain(dt, v, bv, X, K);
#pragma unroll
      for (int j = 0; j < M * DIM; ++j) Kg[a * M * DIM + j] = K[j];
    }
    if (iters) iters[a] = it;
  };
  riccati_from_v_emit<DIM>(dt, v, bv, bd, eps, maxiter, __builtin_amdgcn_ballot_w64(live), emit);
}

// ---------- one lane per agent, lanes refilled (throughput regime) -------------------------------------------------------------
// With several waves queued on every SIMD what is lost is not latency but lanes: an agent needs 41-150 evaluations (mean 59 at
// BASELINE configs[2]'s speed distribution) and a wave of the masked kernel lasts as long as its slowest lane — 150 evaluations for
// 96 % of the waves, 39 % of the lane-evaluations useful.  Here a wave owns a contiguous range of `chunk` agents; a lane whose agent
// is done holds the result (its evaluations are masked off, as in the masked kernel) until `hold` lanes of the wave hold one — or
// nobody is iterating — then all of them hand their agents back in one pass (gain, X, K, iteration count to memory) and take the
// next agents of the range: index = the wave's scalar cursor + the lane's rank among the lanes asking — no atomics, no workspace.
// (Handing back at once, lane by lane, costs more than it saves: some lane finishes in nine passes out of ten, and the whole wave
// would walk through the ~200 instructions of the gain and the 36 scattered stores for two lanes each time — measured, 1.2-1.4x
// instead of the 1.9x of the batched form.)  The wave ends when its range is exhausted and its last agents are done; only that tail
// and the held lanes idle.  Per agent the same evaluations in the same order as every other kernel of this file (same bits, same
// iteration counts); the cap is the lane's own count.  Agents whose matrices do not carry the structured pattern (Src =
// DareFromMats) are skipped: the dense kernel of the launch pair serves them.
template <int DIM, class Src>
__global__ void __launch_bounds__(64)
dare_from_v_refill_kernel(int n, int chunk, int hold, const Src src, float eps, int maxiter,
                          float* __restrict__ Xg, float* __restrict__ Kg, int* __restrict__ iters) {
  constexpr int NN = DIM * DIM;
  constexpr int M = (DIM == 5) ? 2 : 1;
  const int lo = (int)blockIdx.x * chunk;
  const int hi = (n - lo < chunk) ? n : lo + chunk;                 // this wave's agents: [lo, hi)
  const unsigned lane = threadIdx.x;
  int a = lo + (int)lane;                                          // the lane's current agent
  int next = lo + 64;                                              // wave-uniform cursor: the first agent nobody has taken yet
  float dt, v, bv, bd;
  bool active = src.template load<DIM, 1>((size_t)a, 0, a < hi, dt, v, bv, bd);
  bool holding = false, in_y = false;
  dare_mask_t want = ~__builtin_amdgcn_ballot_w64(active);         // lanes without an agent
  Row4 X[4], Y[4];
  float x44 = 1.0f, y44 = 1.0f;
  auto reset = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      X[i].a = (v2f){i == 0 ? 1.0f : 0.0f, i == 1 ? 1.0f : 0.0f};
      X[i].b = (v2f){i == 2 ? 1.0f : 0.0f, i == 3 ? 1.0f : 0.0f};
    }
    x44 = 1.0f;
  };
  reset();
  int count = 0;                                                    // evaluations of the lane's current agent
  auto iter = [&](const Row4* Xi, const float& xi44, Row4* Xo, float& xo44) -> float {
    if constexpr (DIM == 5) { dare5_v_iter_pk(dt, v, bv, bd, Xi, xi44, Xo, xo44); return dare_max_abs_diff(Xo, xo44, Xi, xi44); }
    else { dare4_v_iter_pk(dt, v, bv, Xi, Xo); return dare_max_abs_diff(Xo, Xi); }
  };
  for (;;) {
    const dare_mask_t todo = __builtin_amdgcn_ballot_w64(active);
    const dare_mask_t held = __builtin_amdgcn_ballot_w64(holding);
    if (held && (!todo || __builtin_popcountll(held) >= hold)) {    // hand the finished agents back, all at once
      if (holding) {
        float Xo[NN];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const Row4& r = in_y ? Y[i] : X[i];
          Xo[i + DIM * 0] = r.a.x; Xo[i + DIM * 1] = r.a.y; Xo[i + DIM * 2] = r.b.x; Xo[i + DIM * 3] = r.b.y;
          if constexpr (DIM == 5) { Xo[i + DIM * 4] = 0.0f; Xo[4 + DIM * i] = 0.0f; }
        }
        if constexpr (DIM == 5) Xo[24] = in_y ? y44 : x44;
        const size_t as = (size_t)a;
        if (Xg) {
#pragma unroll
          for (int j = 0; j < NN; ++j) Xg[as * NN + j] = Xo[j];
        }
        if (Kg) {
          float K[M * DIM];
          if (DIM == 5) dlqr5_v_gain(dt, v, bv, bd, Xo, K);
          else dlqr4_v_gain(dt, v, bv, Xo, K);
#pragma unroll
          for (int j = 0; j < M * DIM; ++j) Kg[as * M * DIM + j] = K[j];
        }
        if (iters) iters[as] = count;
        holding = false;
      }
      want |= held;
    }
    if (want && next < hi) {                                        // the lanes without an agent take the next ones, in lane order
      const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
      const bool wants = (want >> lane) & 1;
      const int mine = next + (int)rank;
      next += __builtin_popcountll(want);
      bool got = false;
      if (wants && mine < hi) {
        a = mine;
        got = src.template load<DIM, 1>((size_t)a, 0, true, dt, v, bv, bd);
        reset();
        count = 0;
        active = got;
      }
      want &= ~__builtin_amdgcn_ballot_w64(got);
      continue;                                                     // (the lane of a skipped agent asks again at once)
    }
    if (!todo) break;                                               // nothing iterating, nothing held, nothing left to take
    if (active) {
      const float m = iter(X, x44, Y, y44);
      in_y = true;
      ++count;
      if (m < eps || count >= maxiter) { active = false; holding = true; }
    }
    if (active) {
      const float m = iter(Y, y44, X, x44);
      in_y = false;
      ++count;
      if (m < eps || count >= maxiter) { active = false; holding = true; }
    }
  }
}

// ---------- four lanes per agent ---------------------------------------------------------------------------------------------
// BASELINE-sized batches (configs[2]: 16,384 agents) are 256 waves of the kernel above on 1,024 SIMDs, and the launch lasts as
// long as one wave needs for the 150 evaluations of an agent at the iteration cap: a latency chain on a quarter of the chip.
// Here an agent is a DPP quad — lane r holds row r of the 4x4 block of X, x44 is replicated — so an evaluation is ~1/3 of the
// instructions (one row per lane; the source row of A'X and row 3 of X arrive as quad_perm operands of the multiplies; the
// convergence test is two DPP max steps) and the same batch fills every SIMD.  Same arithmetic per coefficient, same bits.
// All four lanes of a quad take the same decisions (the maximum is reduced across the quad before the test).
//
// Converged agents are NOT masked off: every quad keeps evaluating (nobody looks at a converged agent's later iterates), and
// an agent's X, K and iteration count go to memory straight from the — rare — pass in which its own test succeeds.  The hot
// loop runs four evaluations per trip behind one not-taken branch (their tests reduced across the quads in one interleaved DPP
// sequence) instead of two exec-mask regions per evaluation (~14 scalar instructions, which cost a lone wave as much as vector
// ones), and carries no live-out registers but the iterates themselves.  The evaluation itself is dare_math.h's
// dare5_quad_iter_dev / dare4_quad_iter_dev: 65 issue slots.

// The quad's loop: iterate from X = Q = I until the test passes or the cap is reached, for the agents of `todo` (a lane mask, whole
// quads).  emit(who, W, w44, it): the agents of `who` return iterate (W = this lane's row of the 4x4 block, w44) after `it`
// evaluations — called from the pass in which their test succeeds (or after the cap), so nothing but the iterate is carried.
template <int DIM, class Emit>
__device__ __forceinline__ void riccati_from_v_quad(const QuadLane<float, uint32_t>& c, float eps, int maxiter, dare_mask_t todo, Emit&& emit) {
  float X[4], Y[4], x44 = 1.0f, y44 = 1.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) X[j] = c.q[j];
  auto iter = [&](const float* Xi, const float& xi44, float* Xo, float& xo44) -> float {
    if constexpr (DIM == 5) return dare5_quad_iter_dev(c, Xi, xi44, Xo, xo44);
    else return dare4_quad_iter_dev(c, Xi, Xo);
  };
  if (maxiter <= 0) { emit(todo, X, x44, 0); return; }
  // Four evaluations per trip (X -> Y -> Z -> W -> X); a cap that is not a multiple of four is made one by single evaluations ahead
  // of the loop.
  int i = 0;
  for (const int peel = maxiter & 3; i < peel; ++i) {
    const float m = iter(X, x44, Y, y44);
    const dare_mask_t hit = __builtin_amdgcn_ballot_w64(m < eps) & todo;
    if (hit) { emit(hit, Y, y44, i + 1); todo &= ~hit; }
#pragma unroll
    for (int j = 0; j < 4; ++j) X[j] = Y[j];
    x44 = y44;
  }
  float Z[4], W[4], z44 = 1.0f, w44 = 1.0f;
  for (; i < maxiter && todo; i += 4) {
    // all four evaluations in one basic block (an evaluation's test — a chain of ten dependent instructions ending in two DPP
    // steps — overlaps the next evaluation's start), one not-taken branch per trip; every iterate of the trip is still intact when
    // its agents are handed back
    auto eval = [&](const float* Xi, const float& xi44, float* Xo, float& xo44) -> DqTest {
      if constexpr (DIM == 5) return dare5_quad_eval_dev(c, Xi, xi44, Xo, xo44);
      else return dare4_quad_eval_dev(c, Xi, Xo);
    };
    const DqTest t1 = eval(X, x44, Y, y44);
    const DqTest t2 = eval(Y, y44, Z, z44);
    const DqTest t3 = eval(Z, z44, W, w44);
    const DqTest t4 = eval(W, w44, X, x44);
    float m1 = t1.m, m2 = t2.m, m3 = t3.m, m4 = t4.m;
    dq_quad_test4(m1, m2, m3, m4, t1.first, t2.first, t3.first, t4.first);     // the four tests' quad reductions, interleaved
    const dare_mask_t hit1 = __builtin_amdgcn_ballot_w64(m1 < eps) & todo;
    const dare_mask_t hit2 = __builtin_amdgcn_ballot_w64(m2 < eps) & todo & ~hit1;
    const dare_mask_t hit3 = __builtin_amdgcn_ballot_w64(m3 < eps) & todo & ~(hit1 | hit2);
    const dare_mask_t hit4 = __builtin_amdgcn_ballot_w64(m4 < eps) & todo & ~(hit1 | hit2 | hit3);
    const dare_mask_t any = hit1 | hit2 | hit3 | hit4;
    if (any) {
      if (hit1) emit(hit1, Y, y44, i + 1);
      if (hit2) emit(hit2, Z, z44, i + 2);
      if (hit3) emit(hit3, W, w44, i + 3);
      if (hit4) emit(hit4, X, x44, i + 4);
      todo &= ~any;
    }
  }
  if (todo) emit(todo, X, x44, maxiter);                    // agents that ran into the cap return the last evaluation
}

// the constants of quad lane r (row r of the 4x4 block) for A(0,1) = A(2,3) = dt, A(1,2) = v, B(3,0) = bv, B(4,1) = bd
__device__ __forceinline__ QuadLane<float, uint32_t> dare_quad_lane(int r, float v, float dt, float bv, float bd) {
  QuadLane<float, uint32_t> c;
  c.dt = dt; c.v = v; c.bd = bd; c.bv = bv;
  c.a = (r == 0) ? 1.0f : ((r == 2) ? v : dt);
  c.m2 = (r == 2) ? 0xffffffffu : 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) c.q[j] = (r == j) ? 1.0f : 0.0f;
  return c;
}
// ... for speed v as lqr_steering_control builds A and B from it
__device__ __forceinline__ QuadLane<float, uint32_t> dare_quad_lane(int r, float v, float dt, double L) {
  return dare_quad_lane(r, v, dt, (float)((double)v / L) /* B(3,0) = state.v / L  (float / double literal) */, dt);
}

// the gain from the lane that holds row 3 of the 4x4 block (rows 3 and 4 of X are all dlqr needs)
template <int DIM>
__device__ __forceinline__ void dlqr_quad_gain_row3(const QuadLane<float, uint32_t>& c, const float* W, float w44, float* K) {
  constexpr int NN = DIM * DIM;
  float Xf[NN];
#pragma unroll
  for (int j = 0; j < NN; ++j) Xf[j] = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) Xf[3 + DIM * j] = W[j];
  if constexpr (DIM == 5) { Xf[24] = w44; dlqr5_v_gain(c.dt, c.v, c.bv, c.bd, Xf, K); }
  else dlqr4_v_gain(c.dt, c.v, c.bv, Xf, K);
}

template <int DIM, class Src>
__global__ void __launch_bounds__(256)
dare_from_v_quad_kernel(int n, const Src src, float eps, int maxiter,
                        float* __restrict__ Xg, float* __restrict__ Kg, int* __restrict__ iters) {
  constexpr int NN = DIM * DIM;
  constexpr int M = (DIM == 5) ? 2 : 1;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t a = t >> 2;
  const int r = (int)(threadIdx.x & 3);
  float dt, v, bv, bd;
  const bool live = src.template load<DIM, 4>(a, r, a < (size_t)n, dt, v, bv, bd);
  if (!__builtin_amdgcn_ballot_w64(live)) return;
  const QuadLane<float, uint32_t> c = dare_quad_lane(r, v, dt, bv, bd);
  // the agents of `who` hand back iterate (W, w44) after `it` evaluations
  auto emit = [&](dare_mask_t who, const float* W, float w44, int it) {
    if (!((who >> (threadIdx.x & 63)) & 1)) return;
    if (Xg) {
      float* Xa = Xg + a * NN;
#pragma unroll
      for (int j = 0; j < 4; ++j) Xa[r + DIM * j] = W[j];
      if constexpr (DIM == 5) {
        Xa[r + DIM * 4] = 0.0f; Xa[4 + DIM * r] = 0.0f;
        if (r == 3) Xa[24] = w44;
      }
    }
    if (Kg && r == 3) {       // the gain needs rows 3 (this lane's) and 4 of X only
      float K[M * DIM];
      dlqr_quad_gain_row3<DIM>(c, W, w44, K);
#pragma unroll
      for (int j = 0; j < M * DIM; ++j) Kg[a * M * DIM + j] = K[j];
    }
    if (iters && r == 0) iters[a] = it;
  };
  riccati_from_v_quad<DIM>(c, eps, maxiter, __builtin_amdgcn_ballot_w64(live), emit);
}

}  // namespace crx

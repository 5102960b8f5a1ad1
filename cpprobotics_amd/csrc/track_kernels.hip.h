// track_kernels.hip.h — the callers on either side of the DARE / MPC solves, batched for gfx950:
// course tracking front-end, bicycle-model update, and the closed LQR loop as one persistent kernel.
//
// Replaces, for n independent agents on one shared course,
//   calc_nearest_index      /root/reference/src/lqr_speed_steer_control.cpp:65-83 (= src/lqr_steer_control.cpp:55-73)
//   lqr_steering_control    src/lqr_speed_steer_control.cpp:108-151 (5-state, speed + steer)
//                           src/lqr_steer_control.cpp:98-133       (4-state, steer only)
//   update                  src/lqr_speed_steer_control.cpp:154-164 ; src/model_predictive_control.cpp:69-81
//   closed_loop_prediction  src/lqr_speed_steer_control.cpp:166-205 ; src/lqr_steer_control.cpp:146-197 (math only)
//   calc_nearest_index      src/model_predictive_control.cpp:107-127 (window of N_IND_SEARCH from pind)
//   calc_ref_trajectory     src/model_predictive_control.cpp:130-170
//
// Arithmetic contract (bit parity with the reference's CPU build): fp32 where the reference computes in
// float, double where a `#define`d double literal (DT, L, WB, M_PI, MAX_STEER ...) promotes the expression,
// one rounding back to float at each assignment to a float variable, no fma contraction; cosf/sinf =
// crx_trig.h, atan2f/tanf = crx_fdlibm.h (both bit-identical to glibc), fmod is exact by definition.
// The feed-forward term atan2(L*k, 1.0) is a DOUBLE atan2 in the reference: crx_datan2.h restates glibc 2.35's (FMA build)
// for x = 1.0 — equal to the host libm on all 2^32 float curvatures (rounds 1-3 used OCML's atan there, which rounds to a
// different float for a handful of curvature values; profiles/r04/datan2.txt has the count).
//
// Layout: agent state [n][4] = (x, y, yaw, v) — the reference's `struct State` (include/motion_model.h:31-42);
// the course is five shared read-only arrays of ncourse floats (cx, cy, cyaw, ck, sp).  One agent per lane.
// The nearest-point search is a linear scan of the whole course per agent per tick, as in the reference; the
// course points are staged once per workgroup in LDS, two per 16-byte word as (cx[2j], cx[2j+1], cy[2j], cy[2j+1]), read
// as wave-wide broadcasts and scored two at a time with packed fp32 arithmetic (same IEEE operations, same order).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "crx_fdlibm.h"
#include "crx_datan2.h"
#include "crx_trig.h"
#include "dare_kernels.hip.h"
#include "mpc_kernels.hip.h"

namespace crx {

struct CourseView {
  const float* cx; const float* cy; const float* cyaw; const float* ck; const float* sp;
  int n;
};

struct VehicleParams {   // update(): LQR files use DT 0.1, L 0.5, no speed clamp; MPC uses DT 0.2, WB 2.5 + clamp
  double dt, wheelbase, max_steer, max_speed, min_speed;
  int clamp_speed;
};

constexpr int kTrackBlock = 64;    // one wave per workgroup: BASELINE-sized batches (8k-16k agents) then cover all 256 CUs
constexpr int kCourseLdsMax = 8192;   // (cx,cy) pairs staged in LDS: 64 KB

// #define YAW_P2P(angle) std::fmod(std::fmod((angle)+M_PI, 2*M_PI)-2*M_PI, 2*M_PI)+M_PI   (include/motion_model.h:18)
// The macro argument is a float expression in every use; M_PI promotes the arithmetic to double.
__device__ __forceinline__ double yaw_p2p(float angle) {
  const double pi = 3.14159265358979323846, two_pi = 2 * 3.14159265358979323846;
  return fmod(fmod((double)angle + pi, two_pi) - two_pi, two_pi) + pi;
}

// calc_nearest_index of the LQR files: full scan, first strict minimum of the SQUARED distance; returns that
// squared distance signed by the side of the course the vehicle is on.  `ind` keeps its incoming value if no
// point compares smaller (NaN position) — as the reference's by-reference parameter does.
template <bool LDS>
__device__ __forceinline__ float calc_nearest_index_dev(float sx, float sy, const CourseView& c,
                                                        const float2* __restrict__ pts, int& ind) {
  float mind = FLT_MAX;
  int best = ind;
  if (LDS) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const float4* __restrict__ pairs = reinterpret_cast<const float4*>(pts);
    const v2f s2x = {sx, sx}, s2y = {sy, sy};
    const int np = (c.n + 1) >> 1;                 // an odd course is padded with a NaN point, which never compares smaller
    for (int j = 0; j < np; ++j) {
      const float4 q = pairs[j];
      const v2f px = {q.x, q.y}, py = {q.z, q.w};
      const v2f idx = px - s2x, idy = py - s2y;
      const v2f d_e = idx * idx + idy * idy;       // v_pk_mul, v_pk_mul, v_pk_add: idx*idx + idy*idy of each point, unfused
      if (d_e.x < mind) { mind = d_e.x; best = 2 * j; }
      if (d_e.y < mind) { mind = d_e.y; best = 2 * j + 1; }
    }
  } else {
    for (int i = 0; i < c.n; ++i) {
      const float idx = c.cx[i] - sx, idy = c.cy[i] - sy;
      const float d_e = idx * idx + idy * idy;
      if (d_e < mind) { mind = d_e; best = i; }
    }
  }
  ind = best;
  const int j = best < 0 ? 0 : (best >= c.n ? c.n - 1 : best);   // memory safety only (stale caller index + NaN position)
  const float dxl = c.cx[j] - sx, dyl = c.cy[j] - sy;
  const float angle = (float)yaw_p2p(c.cyaw[j] - atan2f_(dyl, dxl));
  if (angle < 0) mind = -mind;       // mind * -1
  return mind;
}

// A, B from v; Q = I, R = I; cold-started fixed point (dare_kernels.hip.h).  All lanes of the wave iterate
// until the slowest one has converged (each lane freezes its X when its own test passes).
template <int DIM, bool CHAIN = false>
__device__ __forceinline__ void dlqr_from_v_dev(float v, float dtf, double L, float eps, int maxiter, bool live, float* K) {
  constexpr int NN = DIM * DIM;
  constexpr int M = (DIM == 5) ? 2 : 1;
  const float bv = (float)((double)v / L);   // B(3,0) = state.v / L
  if constexpr (CHAIN) {
    // nobody masked off, two evaluations per branch (dare_kernels.hip.h: riccati_from_v_emit); a lane's gain is taken in the rare pass
    // in which its own test succeeds
#pragma unroll
    for (int i = 0; i < M * DIM; ++i) K[i] = 0.0f;
    auto emit = [&](dare_mask_t who, const Row4* W, float w44, int) {
      if (!((who >> (threadIdx.x & 63)) & 1)) return;
      float X[NN];
#pragma unroll
      for (int i = 0; i < NN; ++i) X[i] = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { X[i + DIM * 0] = W[i].a.x; X[i + DIM * 1] = W[i].a.y; X[i + DIM * 2] = W[i].b.x; X[i + DIM * 3] = W[i].b.y; }
      if constexpr (DIM == 5) { X[24] = w44; dlqr5_v_gain(dtf, v, bv, dtf, X, K); }
      else dlqr4_v_gain(dtf, v, bv, X, K);
    };
    riccati_from_v_emit<DIM>(dtf, v, bv, dtf, eps, maxiter, __builtin_amdgcn_ballot_w64(live), emit);
    return;
  }
  float X[NN];
  riccati_from_v<DIM>(dtf, v, bv, dtf, eps, maxiter, live, X);
  if (DIM == 5) dlqr5_v_gain(dtf, v, bv, dtf, X, K);
  else dlqr4_v_gain(dtf, v, bv, X, K);
}

struct LqrCtl { float ai, delta; };

// lqr_steering_control.  DIM 5: returns {ai, delta} (ind is a local starting at 0, :109).  DIM 4: returns
// delta only (ai = 0 here; the caller adds its PID term) and `ind` is the caller's persistent index (:98).
template <int DIM, bool LDS, bool CHAIN = false>
__device__ __forceinline__ LqrCtl lqr_steering_control_dev(float sx, float sy, float syaw, float sv,
                                                           const CourseView& c, const float2* __restrict__ pts,
                                                           int& ind, float& pe, float& pth_e, double dt, double L,
                                                           float eps, int maxiter, bool live) {
  if (DIM == 5) ind = 0;
  const float e = calc_nearest_index_dev<LDS>(sx, sy, c, pts, ind);
  const int j = ind < 0 ? 0 : (ind >= c.n ? c.n - 1 : ind);
  const float k = c.ck[j];
  const float th_e = (float)yaw_p2p(syaw - c.cyaw[j]);
  float K[(DIM == 5 ? 2 : 1) * DIM];
  dlqr_from_v_dev<DIM, CHAIN>(sv, (float)dt, L, eps, maxiter, live, K);
  const float x0 = e;
  const float x1 = (float)((double)(e - pe) / dt);
  const float x2 = th_e;
  const float x3 = (float)((double)(th_e - pth_e) / dt);
  LqrCtl out;
  float u0;
  if (DIM == 5) {
    const float x4 = sv - c.sp[j];
    // ustar = -K * x : (2x5)*(5x1) lazy product, 5-term unrolled redux (t0+t1)+(t2+(t3+t4))  (oracle/eigen_order.h C2)
    float t[5], s[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { t[j] = (-K[0 + 2 * j]); s[j] = (-K[1 + 2 * j]); }
    t[0] *= x0; t[1] *= x1; t[2] *= x2; t[3] *= x3; t[4] *= x4;
    s[0] *= x0; s[1] *= x1; s[2] *= x2; s[3] *= x3; s[4] *= x4;
    u0 = (t[0] + t[1]) + (t[2] + (t[3] + t[4]));
    out.ai = (s[0] + s[1]) + (s[2] + (s[3] + s[4]));
  } else {
    // (-K * x)(0) : row vector times vector, one SSE packet: (t0+t2)+(t1+t3)  (oracle/eigen_order.h C1)
    const float t0 = (-K[0]) * x0, t1 = (-K[1]) * x1, t2 = (-K[2]) * x2, t3 = (-K[3]) * x3;
    u0 = (t0 + t2) + (t1 + t3);
    out.ai = 0.0f;
  }
  const float ff = (float)datan2_one_(L * (double)k);    // std::atan2((L*k), (double)1.0)
  const float fb = (float)yaw_p2p(u0);
  out.delta = ff + fb;
  pe = e;
  pth_e = th_e;
  return out;
}

// update(State&, a, delta)
__device__ __forceinline__ void update_dev(float& sx, float& sy, float& syaw, float& sv, float a, float delta,
                                           const VehicleParams& p) {
  if ((double)delta >= p.max_steer) delta = (float)p.max_steer;
  if ((double)delta <= -p.max_steer) delta = (float)(-p.max_steer);
  float sn, cs;
  sincosf_(syaw, &sn, &cs);
  const float nx = (float)((double)sx + (double)(sv * cs) * p.dt);
  const float ny = (float)((double)sy + (double)(sv * sn) * p.dt);
  const float nyaw = (float)((double)syaw + (double)sv / p.wheelbase * (double)tanf_(delta) * p.dt);
  float nv = (float)((double)sv + (double)a * p.dt);
  if (p.clamp_speed) {
    if ((double)nv > p.max_speed) nv = (float)p.max_speed;
    if ((double)nv < p.min_speed) nv = (float)p.min_speed;
  }
  sx = nx; sy = ny; syaw = nyaw; sv = nv;
}

__device__ __forceinline__ void stage_course(const CourseView& c, float2* pts) {
  float4* pairs = reinterpret_cast<float4*>(pts);
  const int np = (c.n + 1) >> 1;
  for (int j = threadIdx.x; j < np; j += blockDim.x) {
    const int i0 = 2 * j, i1 = 2 * j + 1;
    const bool has1 = i1 < c.n;
    pairs[j] = make_float4(c.cx[i0], has1 ? c.cx[i1] : __builtin_nanf(""), c.cy[i0], has1 ? c.cy[i1] : __builtin_nanf(""));
  }
  __syncthreads();
}

// ---- one control evaluation per agent --------------------------------------------------------------
template <int DIM, bool LDS>
__global__ void __launch_bounds__(kTrackBlock)
lqr_steering_control_kernel(int n, const float* __restrict__ state, CourseView c, int* __restrict__ ind_io,
                            float* __restrict__ pe_io, float* __restrict__ pth_io, double dt, double L, float eps,
                            int maxiter, float* __restrict__ control) {
  extern __shared__ __attribute__((aligned(16))) float2 pts[];
  if (LDS) stage_course(c, pts);
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = a < (size_t)n;
  const size_t aa = live ? a : 0;
  const float4 s = reinterpret_cast<const float4*>(state)[aa];
  int ind = (DIM == 4 && ind_io) ? ind_io[aa] : 0;
  float pe = pe_io[aa], pth = pth_io[aa];
  const LqrCtl u = lqr_steering_control_dev<DIM, LDS>(s.x, s.y, s.z, s.w, c, pts, ind, pe, pth, dt, L, eps, maxiter, live);
  if (!live) return;
  if (ind_io) ind_io[a] = ind;
  pe_io[a] = pe; pth_io[a] = pth;
  if (DIM == 5) reinterpret_cast<float2*>(control)[a] = make_float2(u.ai, u.delta);
  else control[a] = u.delta;
}

__global__ void __launch_bounds__(kTrackBlock)
update_kernel(int n, float* __restrict__ state, const float* __restrict__ a_in, const float* __restrict__ delta_in,
              VehicleParams p) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  float4 s = reinterpret_cast<float4*>(state)[a];
  update_dev(s.x, s.y, s.z, s.w, a_in[a], delta_in[a], p);
  reinterpret_cast<float4*>(state)[a] = s;
}

// ---- the closed LQR loop: control + update + goal test, max_ticks times, state in registers ------------
// DIM 5: closed_loop_prediction of lqr_speed_steer_control.cpp (:166-205).  DIM 4: of lqr_steer_control.cpp
// (:146-197): ai = KP*(speed_profile[ind] - v), ind += 1 while |v| <= stop_speed.
// The reference's loops never advance `time_` (they run until the goal is reached); max_ticks bounds them here.
// ticks_done[a] = number of control+update ticks executed (the goal tick included); traj_hist (optional,
// [max_ticks][n][4]) receives the state after each tick (rows past ticks_done are left untouched).
template <int DIM, bool LDS, bool CHAIN = false>
__global__ void __launch_bounds__(kTrackBlock)
lqr_closed_loop_kernel(int n, int max_ticks, float* __restrict__ state, CourseView c, float* __restrict__ pe_io,
                       float* __restrict__ pth_io, int* __restrict__ ind_io, double dt, double L, float eps, int maxiter,
                       VehicleParams vp, float goal_x, float goal_y, float goal_dis, double kp, float stop_speed,
                       float* __restrict__ traj_hist, int* __restrict__ ticks_done) {
  extern __shared__ __attribute__((aligned(16))) float2 pts[];
  if (LDS) stage_course(c, pts);
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = a < (size_t)n;
  const size_t aa = live ? a : 0;
  float4 s = reinterpret_cast<const float4*>(state)[aa];
  float pe = pe_io ? pe_io[aa] : 0.0f, pth = pth_io ? pth_io[aa] : 0.0f;
  int ind = ind_io ? ind_io[aa] : 0;
  bool done = !live;
  int ticks = 0;
  for (int t = 0; t < max_ticks; ++t) {
    if (__all(done)) break;
    float4 sn = s;
    float pe_n = pe, pth_n = pth;
    int ind_n = ind;
    const LqrCtl u = lqr_steering_control_dev<DIM, LDS, CHAIN>(sn.x, sn.y, sn.z, sn.w, c, pts, ind_n, pe_n, pth_n, dt, L, eps,
                                                               maxiter, !done);
    float ai = u.ai;
    if (DIM == 4) {                                                   // float ai = KP * (speed_profile[ind]-state.v)
      const int js = ind_n < 0 ? 0 : (ind_n >= c.n ? c.n - 1 : ind_n);
      ai = (float)(kp * (double)(c.sp[js] - sn.w));
    }
    update_dev(sn.x, sn.y, sn.z, sn.w, ai, u.delta, vp);
    if (DIM == 4 && fabsf(sn.w) <= stop_speed) ind_n += 1;
    if (!done) {
      s = sn; pe = pe_n; pth = pth_n; ind = ind_n;
      ticks = t + 1;
      if (traj_hist) reinterpret_cast<float4*>(traj_hist)[(size_t)t * n + a] = s;
      const float dx = s.x - goal_x, dy = s.y - goal_y;
      if (sqrtf(dx * dx + dy * dy) <= goal_dis) done = true;
    }
  }
  if (!live) return;
  reinterpret_cast<float4*>(state)[a] = s;
  if (pe_io) pe_io[a] = pe;
  if (pth_io) pth_io[a] = pth;
  if (ind_io) ind_io[a] = ind;
  if (ticks_done) ticks_done[a] = ticks;
}

// ---- the closed LQR loop, four lanes per agent -----------------------------------------------------------------------------------
// BASELINE-sized batches (configs[2]: 16,384 agents) are 256 waves of the kernel above on 1,024 SIMDs, each a latency chain of
// ~11 k instructions per tick.  Here an agent is a DPP quad (dare_kernels.hip.h): the Riccati iteration runs one row of X per
// lane (~0.6x the instructions per evaluation, and a wave waits for the slowest of 16 agents, not of 64), the course scan is
// split four ways (lane r scores the pairs j = r mod 4; the quad keeps the lexicographic minimum of (distance, index), which
// is the sequential scan's first strict minimum), and everything else of a tick — error state, -K x, update, goal test — is
// evaluated redundantly by the four lanes on identical registers.  The same batch then fills every SIMD.  Same arithmetic per
// coefficient as the one-lane kernel: same bits (tests/test_track_gpu.py runs both layouts against the oracle).
// The gain of the pass in which an agent's test succeeds travels through a per-wave LDS slot (written by the lane that holds row 3).
template <int CTRL>
__device__ __forceinline__ float quad_perm_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int quad_perm_i(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }

// calc_nearest_index for a quad: lane r scans the LDS pairs r, r+4, ...; every lane returns the agent's result.
__device__ __forceinline__ float calc_nearest_index_quad(float sx, float sy, const CourseView& c, const float2* __restrict__ pts, int& ind, int r) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  float mind = FLT_MAX;
  int best = 0x7fffffff;                            // "no point compared smaller" (NaN position)
  const float4* __restrict__ pairs = reinterpret_cast<const float4*>(pts);
  const v2f s2x = {sx, sx}, s2y = {sy, sy};
  const int np = (c.n + 1) >> 1;
  auto score = [&](const float4& q, int j) {
    const v2f px = {q.x, q.y}, py = {q.z, q.w};
    const v2f idx = px - s2x, idy = py - s2y;
    const v2f d_e = idx * idx + idy * idy;
    if (d_e.x < mind) { mind = d_e.x; best = 2 * j; }
    if (d_e.y < mind) { mind = d_e.y; best = 2 * j + 1; }
  };
  int jp = r;
  for (; jp + 12 < np; jp += 16) {                  // four LDS reads in flight: a lone wave would otherwise sit out every read's latency
    const float4 q0 = pairs[jp], q1 = pairs[jp + 4], q2 = pairs[jp + 8], q3 = pairs[jp + 12];
    score(q0, jp); score(q1, jp + 4); score(q2, jp + 8); score(q3, jp + 12);
  }
  for (; jp < np; jp += 4) score(pairs[jp], jp);
  {                                                 // lanes (0,1) <-> (1,0), (2,3) <-> (3,2); then pairs <-> pairs
    const float om = quad_perm_f<0xb1>(mind); const int ob = quad_perm_i<0xb1>(best);      // quad_perm [1,0,3,2]
    if (om < mind || (om == mind && ob < best)) { mind = om; best = ob; }
  }
  {
    const float om = quad_perm_f<0x4e>(mind); const int ob = quad_perm_i<0x4e>(best);      // quad_perm [2,3,0,1]
    if (om < mind || (om == mind && ob < best)) { mind = om; best = ob; }
  }
  if (best != 0x7fffffff) ind = best;
  best = ind;
  const int j = best < 0 ? 0 : (best >= c.n ? c.n - 1 : best);
  const float dxl = c.cx[j] - sx, dyl = c.cy[j] - sy;
  const float angle = (float)yaw_p2p(c.cyaw[j] - atan2f_(dyl, dxl));
  if (angle < 0) mind = -mind;
  return mind;
}

// -K x, feed-forward, feedback: the tail of lqr_steering_control from the error state on (:134-150 / :121-132)
template <int DIM>
__device__ __forceinline__ LqrCtl lqr_control_from_gain(const float* K, float e, float th_e, float pe, float pth_e, float sv, float spj, float k,
                                                        double dt, double L) {
  const float x0 = e;
  const float x1 = (float)((double)(e - pe) / dt);
  const float x2 = th_e;
  const float x3 = (float)((double)(th_e - pth_e) / dt);
  LqrCtl out;
  float u0;
  if (DIM == 5) {
    const float x4 = sv - spj;
    float t[5], s[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { t[j] = (-K[0 + 2 * j]); s[j] = (-K[1 + 2 * j]); }
    t[0] *= x0; t[1] *= x1; t[2] *= x2; t[3] *= x3; t[4] *= x4;
    s[0] *= x0; s[1] *= x1; s[2] *= x2; s[3] *= x3; s[4] *= x4;
    u0 = (t[0] + t[1]) + (t[2] + (t[3] + t[4]));
    out.ai = (s[0] + s[1]) + (s[2] + (s[3] + s[4]));
  } else {
    const float t0 = (-K[0]) * x0, t1 = (-K[1]) * x1, t2 = (-K[2]) * x2, t3 = (-K[3]) * x3;
    u0 = (t0 + t2) + (t1 + t3);
    out.ai = 0.0f;
  }
  const float ff = (float)datan2_one_(L * (double)k);
  const float fb = (float)yaw_p2p(u0);
  out.delta = ff + fb;
  return out;
}

template <int DIM>
__global__ void __launch_bounds__(kTrackBlock)
lqr_closed_loop_quad_kernel(int n, int max_ticks, float* __restrict__ state, CourseView c, float* __restrict__ pe_io,
                            float* __restrict__ pth_io, int* __restrict__ ind_io, double dt, double L, float eps, int maxiter,
                            VehicleParams vp, float goal_x, float goal_y, float goal_dis, double kp, float stop_speed,
                            float* __restrict__ traj_hist, int* __restrict__ ticks_done) {
  constexpr int M = (DIM == 5) ? 2 : 1;
  constexpr int APB = kTrackBlock / 4;               // agents per workgroup (one wave)
  extern __shared__ __attribute__((aligned(16))) float2 pts[];
  __shared__ float s_K[APB][M * DIM + (DIM == 5 ? 1 : 0)];     // odd row stride: the four readers of a quad hit one bank row each
  stage_course(c, pts);
  const size_t a = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const int r = (int)(threadIdx.x & 3), q = (int)(threadIdx.x >> 2);
  const bool live = a < (size_t)n;
  const size_t aa = live ? a : 0;
  float4 s = reinterpret_cast<const float4*>(state)[aa];
  float pe = pe_io ? pe_io[aa] : 0.0f, pth = pth_io ? pth_io[aa] : 0.0f;
  int ind = ind_io ? ind_io[aa] : 0;
  bool done = !live;
  int ticks = 0;
  const float dtf = (float)dt;
  for (int t = 0; t < max_ticks; ++t) {
    if (__all(done)) break;
    float4 sn = s;
    int ind_n = (DIM == 5) ? 0 : ind;
    const float e = calc_nearest_index_quad(sn.x, sn.y, c, pts, ind_n, r);
    const int j = ind_n < 0 ? 0 : (ind_n >= c.n ? c.n - 1 : ind_n);
    const float k = c.ck[j];
    const float th_e = (float)yaw_p2p(sn.z - c.cyaw[j]);
    const QuadLane<float, uint32_t> ql = dare_quad_lane(r, sn.w, dtf, L);
    auto emit = [&](dare_mask_t who, const float* W, float w44, int) {
      if (r != 3 || !((who >> (threadIdx.x & 63)) & 1)) return;
      float K[M * DIM];
      dlqr_quad_gain_row3<DIM>(ql, W, w44, K);
#pragma unroll
      for (int i = 0; i < M * DIM; ++i) s_K[q][i] = K[i];
    };
    riccati_from_v_quad<DIM>(ql, eps, maxiter, __builtin_amdgcn_ballot_w64(!done), emit);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float K[M * DIM];
#pragma unroll
    for (int i = 0; i < M * DIM; ++i) K[i] = s_K[q][i];
    __builtin_amdgcn_wave_barrier();                 // the next tick's writes come after these reads
    const LqrCtl u = lqr_control_from_gain<DIM>(K, e, th_e, pe, pth, sn.w, c.sp[j], k, dt, L);
    float ai = u.ai;
    if (DIM == 4) {
      const int js = ind_n < 0 ? 0 : (ind_n >= c.n ? c.n - 1 : ind_n);
      ai = (float)(kp * (double)(c.sp[js] - sn.w));
    }
    update_dev(sn.x, sn.y, sn.z, sn.w, ai, u.delta, vp);
    if (DIM == 4 && fabsf(sn.w) <= stop_speed) ind_n += 1;
    if (!done) {
      s = sn; pe = e; pth = th_e; ind = ind_n;
      ticks = t + 1;
      if (traj_hist && r == 0) reinterpret_cast<float4*>(traj_hist)[(size_t)t * n + a] = s;
      const float dx = s.x - goal_x, dy = s.y - goal_y;
      if (sqrtf(dx * dx + dy * dy) <= goal_dis) done = true;
    }
  }
  if (!live || r != 0) return;
  reinterpret_cast<float4*>(state)[a] = s;
  if (pe_io) pe_io[a] = pe;
  if (pth_io) pth_io[a] = pth;
  if (ind_io) ind_io[a] = ind;
  if (ticks_done) ticks_done[a] = ticks;
}

// one control evaluation per agent, four lanes per agent (the tick of the loop above as its own launch)
template <int DIM>
__global__ void __launch_bounds__(kTrackBlock)
lqr_steering_control_quad_kernel(int n, const float* __restrict__ state, CourseView c, int* __restrict__ ind_io,
                                 float* __restrict__ pe_io, float* __restrict__ pth_io, double dt, double L, float eps,
                                 int maxiter, float* __restrict__ control) {
  constexpr int M = (DIM == 5) ? 2 : 1;
  constexpr int APB = kTrackBlock / 4;
  extern __shared__ __attribute__((aligned(16))) float2 pts[];
  __shared__ float s_K[APB][M * DIM + (DIM == 5 ? 1 : 0)];
  stage_course(c, pts);
  const size_t a = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const int r = (int)(threadIdx.x & 3), q = (int)(threadIdx.x >> 2);
  const bool live = a < (size_t)n;
  const size_t aa = live ? a : 0;
  const float4 s = reinterpret_cast<const float4*>(state)[aa];
  int ind = (DIM == 4 && ind_io) ? ind_io[aa] : 0;
  const float pe = pe_io[aa], pth = pth_io[aa];
  const float e = calc_nearest_index_quad(s.x, s.y, c, pts, ind, r);
  const int j = ind < 0 ? 0 : (ind >= c.n ? c.n - 1 : ind);
  const float k = c.ck[j];
  const float th_e = (float)yaw_p2p(s.z - c.cyaw[j]);
  const QuadLane<float, uint32_t> ql = dare_quad_lane(r, s.w, (float)dt, L);
  auto emit = [&](dare_mask_t who, const float* W, float w44, int) {
    if (r != 3 || !((who >> (threadIdx.x & 63)) & 1)) return;
    float K[M * DIM];
    dlqr_quad_gain_row3<DIM>(ql, W, w44, K);
#pragma unroll
    for (int i = 0; i < M * DIM; ++i) s_K[q][i] = K[i];
  };
  riccati_from_v_quad<DIM>(ql, eps, maxiter, __builtin_amdgcn_ballot_w64(live), emit);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float K[M * DIM];
#pragma unroll
  for (int i = 0; i < M * DIM; ++i) K[i] = s_K[q][i];
  const LqrCtl u = lqr_control_from_gain<DIM>(K, e, th_e, pe, pth, s.w, c.sp[j], k, dt, L);
  if (!live || r != 0) return;
  if (ind_io) ind_io[a] = ind;
  pe_io[a] = e; pth_io[a] = th_e;
  if (DIM == 5) reinterpret_cast<float2*>(control)[a] = make_float2(u.ai, u.delta);
  else control[a] = u.delta;
}

// ---- MPC front-end --------------------------------------------------------------------------------------
// calc_nearest_index(state, cx, cy, cyaw, pind) of model_predictive_control.cpp: window [pind, pind+N_IND_SEARCH).
// The reference reads cx[i] without a bounds check (:110); the window is clipped to the course here.
__device__ __forceinline__ int calc_nearest_index_window_dev(float sx, float sy, const CourseView& c, int pind, int nsearch) {
  float mind = FLT_MAX;
  float ind = 0;                          // `float ind = 0;` (:108) — the index travels through a float
  const int lo = pind < 0 ? 0 : pind;
  const long long hi_ll = (long long)pind + nsearch;
  const int hi = hi_ll > c.n ? c.n : (int)hi_ll;
  for (int i = lo; i < hi; ++i) {
    const float idx = c.cx[i] - sx, idy = c.cy[i] - sy;
    const float d_e = idx * idx + idy * idy;
    if (d_e < mind) { mind = d_e; ind = (float)i; }
  }
  return (int)ind;
}

// calc_ref_trajectory (:130-170).  xref: column-major 4 x T per agent (Eigen::Matrix<float,NX,T>::data()).
__global__ void __launch_bounds__(kTrackBlock)
calc_ref_trajectory_kernel(int n, int T, const float* __restrict__ state, CourseView c, float dl, double dt, int nsearch,
                           int* __restrict__ target_ind, float* __restrict__ xref, const int* __restrict__ active) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  if (active && !active[a]) return;
  const float4 s = reinterpret_cast<const float4*>(state)[a];
  const int tind = target_ind[a];
  int ind = calc_nearest_index_window_dev(s.x, s.y, c, tind, nsearch);
  if (tind >= ind) ind = tind;
  float* xr = xref + (size_t)a * 4 * T;
  float travel = 0.0f;
  const int last = c.n - 1;
  for (int i = 0; i < T; ++i) {
    travel = (float)((double)travel + (double)fabsf(s.w) * dt);     // travel += std::abs(state.v) * DT
    const int dind = (int)roundf(travel / dl);
    const long long j_ll = (long long)ind + dind;
    const int j = (j_ll < c.n) ? (int)j_ll : last;
    xr[4 * i + 0] = c.cx[j]; xr[4 * i + 1] = c.cy[j]; xr[4 * i + 2] = c.cyaw[j]; xr[4 * i + 3] = c.sp[j];
  }
  target_ind[a] = ind;
}

__global__ void __launch_bounds__(kTrackBlock)
calc_nearest_index_window_kernel(int n, const float* __restrict__ state, CourseView c, const int* __restrict__ pind, int nsearch,
                                 int* __restrict__ ind_out) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  const float4 s = reinterpret_cast<const float4*>(state)[a];
  ind_out[a] = calc_nearest_index_window_dev(s.x, s.y, c, pind[a], nsearch);
}

template <bool LDS>
__global__ void __launch_bounds__(kTrackBlock)
calc_nearest_index_kernel(int n, const float* __restrict__ state, CourseView c, int* __restrict__ ind_io, float* __restrict__ e_out) {
  extern __shared__ __attribute__((aligned(16))) float2 pts[];
  if (LDS) stage_course(c, pts);
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  const float4 s = reinterpret_cast<const float4*>(state)[a];
  int ind = ind_io[a];
  const float e = calc_nearest_index_dev<LDS>(s.x, s.y, c, pts, ind);
  ind_io[a] = ind;
  if (e_out) e_out[a] = e;
}

__global__ void fill_int_kernel(int n, int* __restrict__ p, int v) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a < (size_t)n) p[a] = v;
}

// MPC closed loop (mpc_simulation :371-385), the per-tick tail: update with the first control of the solution,
// goal test, bookkeeping.  sol: n x (4T + 2(T-1)), the reference's layout; delta_start = 4T, a_start = 4T + T-1.
__global__ void __launch_bounds__(kTrackBlock)
mpc_tick_tail_kernel(int n, int T, int tick, float* __restrict__ state, const float* __restrict__ sol, VehicleParams vp,
                     float goal_x, float goal_y, float goal_dis, int* __restrict__ active, int* __restrict__ ticks_done,
                     float* __restrict__ traj_hist) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  if (!active[a]) return;
  const int nv = 4 * T + 2 * (T - 1);
  const float* o = sol + (size_t)a * nv;
  float4 s = reinterpret_cast<float4*>(state)[a];
  update_dev(s.x, s.y, s.z, s.w, o[4 * T + (T - 1)], o[4 * T], vp);
  reinterpret_cast<float4*>(state)[a] = s;
  ticks_done[a] = tick + 1;
  if (traj_hist) reinterpret_cast<float4*>(traj_hist)[(size_t)tick * n + a] = s;
  const float dx = s.x - goal_x, dy = s.y - goal_y;
  if (sqrtf(dx * dx + dy * dy) <= goal_dis) active[a] = 0;
}

// MPC closed loop (mpc_simulation :371-385) for n agents as ONE persistent kernel: per tick calc_ref_trajectory (:130-170, the
// window search included), mpc_solve (mpc_kernels.hip.h: mpc_solve_lane), update with the first control of the solution (:376),
// the goal test (:380-384) — state, target_ind and the reference trajectory of a lane stay in registers / the lane's private
// memory for the whole episode, nothing is enqueued from the host between ticks.  An agent that has reached the goal idles
// (masked) until the last agent of its wave has; a wave leaves the loop when all its agents have.  The course arrays are read
// straight from global memory: T gathers per tick against a solve of ~10^5 instructions.
template <int MAXT>
__global__ void __launch_bounds__(64)
mpc_closed_loop_kernel(int n, int T, int max_ticks, float* __restrict__ state, CourseView c, float dl, int nsearch, MpcP p, VehicleParams vp,
                       float goal_x, float goal_y, float goal_dis, int* __restrict__ target_ind, float* __restrict__ traj_hist,
                       int* __restrict__ ticks_done, int* __restrict__ solve_flags) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = a < (size_t)n;
  const size_t ag = live ? a : 0;
  float4 s = reinterpret_cast<const float4*>(state)[ag];
  int tind = target_ind[ag];
  int ticks = 0, flags = 0;      // flags: bit 0 = some tick's solve did not converge (its first control was applied all the same, as
                                 // the reference applies whatever IPOPT returns, :338-339), bit 1 = some tick's speed bound infeasible
  bool active = live;
  float4 xr[MAXT];
  const int last = c.n - 1;
  for (int tick = 0; tick < max_ticks; ++tick) {
    if (!__any(active)) break;
    if (active) {                                                            // calc_ref_trajectory
      int ind = calc_nearest_index_window_dev(s.x, s.y, c, tind, nsearch);
      if (tind >= ind) ind = tind;
      float travel = 0.0f;
      for (int i = 0; i < T; ++i) {
        travel = (float)((double)travel + (double)fabsf(s.w) * p.dt);
        const int dind = (int)roundf(travel / dl);
        const long long j_ll = (long long)ind + dind;
        const int j = (j_ll < c.n) ? (int)j_ll : last;
        xr[i] = make_float4(c.cx[j], c.cy[j], c.cyaw[j], c.sp[j]);
      }
      tind = ind;
    }
    int st; double J; float a0, d0;
    mpc_solve_lane<MAXT>(active, T, s, xr, p, nullptr, st, J, a0, d0);
    if (active) {
      flags |= ((st & 1) ? 0 : 1) | (st & 2);
      update_dev(s.x, s.y, s.z, s.w, a0, d0, vp);
      ticks = tick + 1;
      if (traj_hist) reinterpret_cast<float4*>(traj_hist)[(size_t)tick * n + a] = s;
      const float dx = s.x - goal_x, dy = s.y - goal_y;
      if (sqrtf(dx * dx + dy * dy) <= goal_dis) active = false;
    }
  }
  if (!live) return;
  reinterpret_cast<float4*>(state)[a] = s;
  target_ind[a] = tind;
  ticks_done[a] = ticks;
  if (solve_flags) solve_flags[a] = flags;
}

}  // namespace crx

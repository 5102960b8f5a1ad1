// frenet_kernels.hip.h — batched Frenet optimal-trajectory planner for gfx950: ONE AGENT PER WAVEFRONT, the candidate
// paths of one planning call (14 lateral offsets x 6 horizons x 2 target speeds = 168 with the reference's constants:
// the float accumulation 4.0, 4.2, ... of the horizon loop stops at 4.9999995 < MAXT) spread over the 64 lanes.
// A planning call runs in three wave-synchronous phases (see frenet_run_kernel): what depends only on (horizon, target
// speed) — the longitudinal quartic, its maxima and jerk sum, the course frame at every sample — is computed once per
// call and parked in an LDS table; each lane then builds its candidate's lateral quintic and walks the table: global
// point, heading/curvature sliding window and obstacle test in registers.  A wave-wide (cost, index) reduction picks the
// reference's winner (the LAST candidate in generation order attaining the minimum cost among the survivors).  All ticks
// of an episode are fused: plan -> hand the winner's second sample over as the new state -> goal test.
//
// Replaces, for n independent agents sharing one course and one obstacle set,
// /root/reference/src/frenet_optimal_trajectory.cpp: calc_frenet_paths :51-100, calc_global_paths :102-136,
// check_collision :138-148, check_paths :150-158, frenet_optimal_planning :160-176, main loop :224-236; with
// /root/reference/include/quintic_polynomial.h:39-69, quartic_polynomial.h:37-64 and the Spline evaluation of
// cubic_spline.h:67-83,:118-127 (the spline coefficients are built once per course on the host, crx_frenet_spline_build).
//
// Arithmetic contract: every expression keeps the reference's C++ types (float members, double macros,
// std::pow(float,int) and std::cos(float + double) in double, atan2/sqrt of floats in float); atan2f is glibc-exact
// (crx_fdlibm.h); the double cos / sin of :111-112 are glibc's sin() / cos() restated (crx_dsincos.h: bit-identical on every
// argument float + M_PI/2 can form, two separate evaluations as in the reference's unoptimised build); std::pow comes from the
// host's libm (FrPowArg below) or is an exact product; the 3x3 / 2x2 coefficient solves are Eigen's ColPivHouseholderQR restated
// in float (crx_qr.h).  No operation is left to a device library: the parity tests demand equal bits (costs, verdicts, winners,
// whole episodes) against the oracle and against the reference's own lines, without exceptions.
// Reference quirks that decide the numbers are kept (the missing factor 5 in the quintic's first derivative,
// quintic_polynomial.h:53; maxima starting at FLT_MIN); where the reference is undefined (a path with < 2 points on the course) the path is dropped, where it would
// throw (s before the course) the path is dropped and status bit 2 is set.
//
// State per agent: (s0, c_speed, c_d, c_d_d, c_d_dd).  History row: (s0, c_speed, c_d, c_d_d, c_d_dd, x, y, cf).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "crx_fdlibm.h"
#include "crx_qr.h"
#include "crx_dsincos.h"

namespace crx {

struct FrenetCfg {   // the #defines :20-38, as the double expressions they expand to
  double max_speed, max_accel, max_curvature, max_road_width, d_road_w, dt, maxt, mint, target_speed, d_t_s;
  int n_s_sample;
  double robot_radius, kj, kt, kd, klat, klon;
};

constexpr int kFrWavesPerBlock = 4;
constexpr int kFrMaxDi = 64, kFrMaxTi = 32, kFrMaxTv = 16, kFrMaxT = 64, kFrMaxKnots = 64, kFrMaxOb = 128;
constexpr int kFrMaxPaths = 4096, kFrMaxCombos = 64, kFrTabLdsBytes = 48 * 1024;

struct FrQuintic { float a0, a1, a2, a3, a4, a5; };
struct FrQuartic { float a0, a1, a2, a3, a4; };

// std::pow(x, k) of a float x (double pow of the promoted arguments, quintic_polynomial.h:41-68, quartic_polynomial.h:39-59) for
// k = 2..5.  x^2 is exact in double (48 significant bits), so pow returns the product; x^3..x^5 are NOT formed here: the
// time grid and the horizons are known on the host before the launch (they depend on the configuration only), and the C ABI
// entry point evaluates libm's pow on them — the reference's own call — and hands the values over as a kernel argument.
struct FrPow { double t2, t3, t4, t5; };
struct FrPowArg { FrPow t[64 /* kFrMaxT */]; FrPow T[32 /* kFrMaxTi */]; };

__device__ __forceinline__ FrQuintic fr_quintic(float xs, float vxs, float axs, float xe, float vxe, float axe, float T, const FrPow& wT) {
  FrQuintic q;
  q.a0 = xs; q.a1 = vxs; q.a2 = (float)((double)axs / 2.0);
  // A and B as the comma initialisers of quintic_polynomial.h:41-47 fill them (double expressions rounded to float entries),
  // then A.colPivHouseholderQr().solve(B) in float (:49; crx_qr.h)
  const double T2 = wT.t2, T3 = wT.t3, T4 = wT.t4, T5 = wT.t5;
  float A[9] = {(float)T3, (float)(3.0 * T2), 6.0f * T,                    // column 0 (column-major)
                (float)T4, (float)(4.0 * T3), (float)(12.0 * T2),          // column 1
                (float)T5, (float)(5.0 * T4), (float)(20.0 * T3)};         // column 2
  float B[3] = {(float)((double)(xe - q.a0 - q.a1 * T) - (double)q.a2 * T2), vxe - q.a1 - 2.0f * q.a2 * T, axe - 2.0f * q.a2};
  float c[3];
  colpiv_qr_solve<3, true>(3, A, B, c);
  q.a3 = c[0]; q.a4 = c[1]; q.a5 = c[2];
  return q;
}
__device__ __forceinline__ FrQuartic fr_quartic(float xs, float vxs, float axs, float vxe, float axe, float T, const FrPow& wT) {
  FrQuartic q;
  q.a0 = xs; q.a1 = vxs; q.a2 = (float)((double)axs / 2.0);
  const double T2 = wT.t2, T3 = wT.t3;                                     // quartic_polynomial.h:38-45
  float A[4] = {(float)(3.0 * T2), 6.0f * T, (float)(4.0 * T3), (float)(12.0 * T2)};
  float B[2] = {vxe - q.a1 - 2.0f * q.a2 * T, axe - 2.0f * q.a2};
  float c[2];
  colpiv_qr_solve<2, true>(2, A, B, c);
  q.a3 = c[0]; q.a4 = c[1];
  return q;
}
// the evaluation expressions of quintic_polynomial.h:44-62 / quartic_polynomial.h:44-60: float until the first pow, double
// after.  The powers t^2..t^5 of the time grid are staged once per block in LDS (frenet_run_kernel).
__device__ __forceinline__ float fr_q5_point(const FrQuintic& q, float t, const FrPow& w) {
  return (float)(((((double)(q.a0 + q.a1 * t) + (double)q.a2 * w.t2) + (double)q.a3 * w.t3) + (double)q.a4 * w.t4) + (double)q.a5 * w.t5);
}
__device__ __forceinline__ float fr_q5_d1(const FrQuintic& q, float t, const FrPow& w) {
  return (float)((((double)(q.a1 + 2.0f * q.a2 * t) + (double)(3.0f * q.a3) * w.t2) + (double)(4.0f * q.a4) * w.t3) + (double)q.a5 * w.t4);
}
__device__ __forceinline__ float fr_q5_d2(const FrQuintic& q, float t, const FrPow& w) {
  return (float)(((double)(2.0f * q.a2 + 6.0f * q.a3 * t) + (double)(12.0f * q.a4) * w.t2) + (double)(20.0f * q.a5) * w.t3);
}
__device__ __forceinline__ float fr_q5_d3(const FrQuintic& q, float t, const FrPow& w) {
  return (float)((double)(6.0f * q.a3 + 24.0f * q.a4 * t) + (double)(60.0f * q.a5) * w.t2);
}
__device__ __forceinline__ float fr_q4_point(const FrQuartic& q, float t, const FrPow& w) {
  return (float)((((double)(q.a0 + q.a1 * t) + (double)q.a2 * w.t2) + (double)q.a3 * w.t3) + (double)q.a4 * w.t4);
}
__device__ __forceinline__ float fr_q4_d1(const FrQuartic& q, float t, const FrPow& w) {
  return (float)(((double)(q.a1 + 2.0f * q.a2 * t) + (double)(3.0f * q.a3) * w.t2) + (double)(4.0f * q.a4) * w.t3);
}
__device__ __forceinline__ float fr_q4_d2(const FrQuartic& q, float t, const FrPow& w) {
  return (float)((double)(2.0f * q.a2 + 6.0f * q.a3 * t) + (double)(12.0f * q.a4) * w.t2);
}
__device__ __forceinline__ float fr_q4_d3(const FrQuartic& q, float t) { return 6.0f * q.a3 + 24.0f * q.a4 * t; }

// Spline::bisect, cubic_spline.h:118-127, iteratively
__device__ __forceinline__ int fr_bisect(const float* __restrict__ x, float t, int start, int end) {
  for (;;) {
    const int mid = (start + end) / 2;
    const float xm = x[mid];
    if (t == xm || end - start <= 1) return mid;
    if (t > xm) start = mid; else end = mid;
  }
}

// What depends only on the (horizon, target speed) pair — 12 "longitudinal combos" with the reference's constants — is
// computed once per planning call instead of once per candidate: phase A, one lane per combo: the quartic, its samples'
// maxima and jerk sum, the hand-over samples and how many points lie on the course; phase B, one lane per (combo, time
// step): the course position and the unit normal's (cos, sin) there (spline lookup, atan2f, double cos and sin), stored in an
// LDS table; phase C, one lane per candidate: the lateral quintic and the walk along the table.
struct FrTab { float px, py; double cs, sn; };   // poi[0], poi[1], cos(iyaw + pi/2), sin(iyaw + pi/2)   :108-112

// Four waves per SIMD (128 VGPRs, a few spilled) beat the compiler's preferred two by 1.5x on the side bench: the kernel is
// bound by per-wave instruction issue, so resident waves are what fills the VALU.
#ifndef CRX_FR_WAVES
#define CRX_FR_WAVES 4
#endif
__attribute__((amdgpu_waves_per_eu(CRX_FR_WAVES, CRX_FR_WAVES)))
__global__ void __launch_bounds__(64 * kFrWavesPerBlock)
frenet_run_kernel(int n, int max_ticks, float* __restrict__ state, const float* __restrict__ coef, int nx, float goal_x, float goal_y,
                  const float* __restrict__ ob, int nob, FrenetCfg g, float* __restrict__ hist, int* __restrict__ ticks_done,
                  int* __restrict__ status, int* __restrict__ best_idx, int* __restrict__ n_valid, float* __restrict__ path_cf,
                  int* __restrict__ path_ok, int path_cap, int tab_stride, const FrPowArg pw) {
  extern __shared__ FrTab s_tab_all[];           // [waves per block][tab_stride], tab_stride >= nTi * ntv * ntt
  __shared__ float s_coef[9 * kFrMaxKnots];      // rows s, ax,bx,cx,dx, ay,by,cy,dy
  __shared__ float s_ob[2 * kFrMaxOb];
  __shared__ float s_di[kFrMaxDi], s_Ti[kFrMaxTi], s_tv[kFrMaxTv], s_t[kFrMaxT];
  __shared__ FrPow s_pw[kFrMaxT], s_pT[kFrMaxTi];
  __shared__ int s_nt[kFrMaxTi], s_cnt[4];
  __shared__ uint64_t s_sc[kDsincosTabLen];      // glibc's sin/cos table (crx_dsincos.h): gathered per lane in phase B
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  for (int i = threadIdx.x; i < 9 * nx; i += blockDim.x) s_coef[i] = coef[i];
  for (int i = threadIdx.x; i < 2 * nob; i += blockDim.x) s_ob[i] = ob[i];
  for (int i = threadIdx.x; i < kDsincosTabLen; i += blockDim.x) s_sc[i] = kDsincosTab[i];
  if (threadIdx.x == 0) {     // the sample grids, by the reference's own float accumulation (:55-56,:58,:66-68); caps checked by the host
    int ndi = 0, nTi = 0, ntv = 0, ntt = 0;
    for (float di = (float)(-1 * g.max_road_width); di < g.max_road_width && ndi < kFrMaxDi; di += g.d_road_w) s_di[ndi++] = di;
    for (float Ti = (float)g.mint; Ti < g.maxt && nTi < kFrMaxTi; Ti += g.dt) s_Ti[nTi++] = Ti;
    for (float tv = (float)(g.target_speed - g.d_t_s * g.n_s_sample); tv < g.target_speed + g.d_t_s * g.n_s_sample && ntv < kFrMaxTv; tv += g.d_t_s) s_tv[ntv++] = tv;
    const float Tmax = nTi ? s_Ti[nTi - 1] : 0.0f;
    for (float t = 0; t < Tmax && ntt < kFrMaxT; t += g.dt) s_t[ntt++] = t;
    for (int k = 0; k < nTi; ++k) { int c = 0; while (c < ntt && s_t[c] < s_Ti[k]) ++c; s_nt[k] = c; }
    s_cnt[0] = ndi; s_cnt[1] = nTi; s_cnt[2] = ntv; s_cnt[3] = ntt;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < s_cnt[3]; i += blockDim.x) s_pw[i] = pw.t[i];
  for (int i = threadIdx.x; i < s_cnt[1]; i += blockDim.x) s_pT[i] = pw.T[i];
  __syncthreads();
  const size_t a = (size_t)blockIdx.x * wpb + wv;
  if (a >= (size_t)n) return;     // whole waves only: no block barrier below
  FrTab* __restrict__ tab = s_tab_all + (size_t)wv * tab_stride;
  const int nTi = s_cnt[1], ntv = s_cnt[2], ntt = s_cnt[3];
  const int nC = nTi * ntv;        // longitudinal combos, <= 64 (host-checked): combo c = iTi * ntv + itv lives in lane c
  const int P = s_cnt[0] * nC;
  const float* sk = s_coef;
  const float *cax = s_coef + nx, *cbx = s_coef + 2 * nx, *ccx = s_coef + 3 * nx, *cdx = s_coef + 4 * nx;
  const float *cay = s_coef + 5 * nx, *cby = s_coef + 6 * nx, *ccy = s_coef + 7 * nx, *cdy = s_coef + 8 * nx;
  const float s_front = sk[0], s_back = sk[nx - 1];
  const double r2 = g.robot_radius * g.robot_radius;
  const double half_pi = 3.14159265358979323846 / 2.0;    // M_PI/2.0
  float s0 = state[5 * a], c_speed = state[5 * a + 1], c_d = state[5 * a + 2], c_d_d = state[5 * a + 3], c_d_dd = state[5 * a + 4];
  int st = 0, ticks = 0, last_best = -1, last_valid = 0;
  for (int tick = 0; tick < max_ticks; ++tick) {
    // ---- phase A: lane c < nC owns combo c -----------------------------------------------------------------------
    FrQuartic lon = FrQuartic{0, 0, 0, 0, 0};
    float c_max_speed = FLT_MIN, c_max_accel = FLT_MIN, c_Js = 0.0f, c_sd_last = 0, c_s1 = 0, c_sd1 = 0;
    int c_npts = 0, c_drop = 0;
    if (lane < nC) {
      const int iTi = lane / ntv, itv = lane - iTi * ntv;
      const int nt = s_nt[iTi];
      lon = fr_quartic(s0, c_speed, 0.0f, s_tv[itv], 0.0f, s_Ti[iTi], s_pT[iTi]);            // :70
      bool walking = true;
      for (int i = 0; i < nt; ++i) {                                              // :74-85
        const float t = s_t[i];
        const FrPow w = s_pw[i];
        const float s_i = fr_q4_point(lon, t, w), sd_i = fr_q4_d1(lon, t, w), sdd_i = fr_q4_d2(lon, t, w), sddd_i = fr_q4_d3(lon, t);
        if (sd_i > c_max_speed) c_max_speed = sd_i;
        if (sdd_i > c_max_accel) c_max_accel = sdd_i;
        c_Js += sddd_i * sddd_i;
        c_sd_last = sd_i;
        if (i == 1) { c_s1 = s_i; c_sd1 = sd_i; }
        if (walking) {                                                            // :105-107
          if (s_i >= s_back) walking = false;
          else if (s_i < s_front) { walking = false; c_drop = 1; st |= 4; }
          else c_npts = i + 1;
        }
      }
    }
    // ---- phase B: the course frame at every (combo, time step) ---------------------------------------------------
    for (int e0 = 0; e0 < nC * ntt; e0 += 64) {
      const int e = e0 + lane;
      const bool live = e < nC * ntt;
      const int c = live ? e / ntt : 0, i = live ? e - c * ntt : 0;
      FrQuartic q;                                   // every lane takes part in the shuffles
      q.a0 = __shfl(lon.a0, c, 64); q.a1 = __shfl(lon.a1, c, 64); q.a2 = __shfl(lon.a2, c, 64);
      q.a3 = __shfl(lon.a3, c, 64); q.a4 = __shfl(lon.a4, c, 64);
      if (!live) continue;
      const float s_i = fr_q4_point(q, s_t[i], s_pw[i]);
      const int seg = fr_bisect(sk, s_i, 0, nx), segd = fr_bisect(sk, s_i, 0, nx - 1);     // Spline::calc / calc_d
      const float dx = s_i - sk[seg], dxd = s_i - sk[segd];
      FrTab r;
      r.px = cax[seg] + cbx[seg] * dx + ccx[seg] * dx * dx + cdx[seg] * dx * dx * dx;
      r.py = cay[seg] + cby[seg] * dx + ccy[seg] * dx * dx + cdy[seg] * dx * dx * dx;
      const float ddx = cbx[segd] + 2.0f * ccx[segd] * dxd + 3.0f * cdx[segd] * dxd * dxd;
      const float ddy = cby[segd] + 2.0f * ccy[segd] * dxd + 3.0f * cdy[segd] * dxd * dxd;
      const float iyaw = atan2f_(ddy, ddx);
      const double nyaw = (double)iyaw + half_pi;
      r.cs = dcos_(nyaw, s_sc);                      // :111
      r.sn = dsin_(nyaw, s_sc);                      // :112
      tab[e] = r;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- phase C: one candidate per lane ---------------------------------------------------------------------------
    float my_cost = FLT_MAX;          // min_cost :167
    int my_idx = -1, my_valid = 0;
    float w_d1 = 0, w_dd1 = 0, w_ddd1 = 0, w_x1 = 0, w_y1 = 0;
    for (int p0 = 0; p0 < P; p0 += 64) {
      const int p = p0 + lane;
      const bool live = p < P;
      const int c = live ? p % nC : 0, idi = live ? p / nC : 0, iTi = c / ntv;
      // the combo's figures (every lane takes part in the shuffles)
      const float max_speed = __shfl(c_max_speed, c, 64), max_accel = __shfl(c_max_accel, c, 64), Js = __shfl(c_Js, c, 64);
      const float sd_last = __shfl(c_sd_last, c, 64);
      const int npts = __shfl(c_npts, c, 64), drop = __shfl(c_drop, c, 64);
      if (!live) continue;
      const float di = s_di[idi], Ti = s_Ti[iTi];
      const int nt = s_nt[iTi];
      const FrQuintic lat = fr_quintic(c_d, c_d_d, c_d_dd, di, 0.0f, 0.0f, Ti, s_pT[iTi]);     // :57
      // what main hands over (:227-231) is sample [1] of the winner; the costs need the last samples (:89-91); nt >= 2
      const float dd1 = fr_q5_d1(lat, s_t[1], s_pw[1]), ddd1 = fr_q5_d2(lat, s_t[1], s_pw[1]);
      const float d_last = fr_q5_point(lat, s_t[nt - 1], s_pw[nt - 1]);
      float max_curv = FLT_MIN, Jp = 0.0f, d1 = 0, x1 = 0, y1 = 0;
      float px = 0, py = 0, pyaw = 0, pds = 0;       // previous global point, previous segment's heading and length
      unsigned dmin = 0x7f800000u;                   // smallest squared obstacle distance so far (float bits), from +inf
      const FrTab* __restrict__ row = tab + c * ntt;
      for (int i = 0; i < nt; ++i) {
        // lateral samples :58-64: one entry of d, d_d, d_dd, d_ddd per time step
        const float dddd_i = fr_q5_d3(lat, s_t[i], s_pw[i]);
        Jp += dddd_i * dddd_i;
        if (i < npts) {                                                            // calc_global_paths :104-116
          const float d_i = fr_q5_point(lat, s_t[i], s_pw[i]);
          const FrTab f = row[i];
          const float x = (float)((double)f.px + (double)d_i * f.cs);
          const float y = (float)((double)f.py + (double)d_i * f.sn);
          if (i == 1) { d1 = d_i; x1 = x; y1 = y; }
          // check_collision :138-148: `dist <= ROBOT_RADIUS^2` for some (point, obstacle) <=> the smallest dist passes.
          // dist is a sum of squares (never -0): the minimum is taken on the bit patterns as unsigned integers — the float
          // order, with every NaN above +inf, i.e. ignored exactly as a failed comparison is.
          for (int k = 0; k < nob; ++k) {
            const double ex = (double)(x - s_ob[2 * k]), ey = (double)(y - s_ob[2 * k + 1]);
            const float dist = (float)(ex * ex + ey * ey);
            const unsigned db = __float_as_uint(dist);
            dmin = db < dmin ? db : dmin;
          }
          if (i >= 1) {                                                            // headings, lengths, curvature :118-135, sliding
            const float gx = x - px, gy = y - py;
            const float yaw = atan2f_(gy, gx), ds = sqrtf(gx * gx + gy * gy);
            if (i >= 2) { const float cc = (yaw - pyaw) / pds; if (cc > max_curv) max_curv = cc; }
            pyaw = yaw; pds = ds;
          }
          px = x; py = y;
        }
      }
      if (npts >= 2) { const float cc = (pyaw - pyaw) / pds; if (cc > max_curv) max_curv = cc; }   // the appended copy of the last heading :123-124
      const bool dropped = drop || npts < 2;
      const bool collide = (double)__uint_as_float(dmin) <= r2;
      const float dsp = (float)(g.target_speed - (double)sd_last);                               // :89
      const float cd = (float)((g.kj * (double)Jp + g.kt * (double)Ti) + g.kd * ((double)d_last * (double)d_last));
      const float cv = (float)((g.kj * (double)Js + g.kt * (double)Ti) + g.kd * (double)dsp);
      const float cf = (float)(g.klat * (double)cd + g.klon * (double)cv);
      const bool ok = !dropped && (double)max_speed < g.max_speed && (double)max_accel < g.max_accel &&
                      (double)max_curv < g.max_curvature && !collide;                            // :153
      if (path_cf && p < path_cap) path_cf[a * path_cap + p] = cf;
      if (path_ok && p < path_cap) path_ok[a * path_cap + p] = ok ? 1 : 0;
      if (ok) {
        ++my_valid;
        if (my_cost >= cf) {                                                                     // :170 (within a lane the candidates come in generation order)
          my_cost = cf; my_idx = p;
          w_d1 = d1; w_dd1 = dd1; w_ddd1 = ddd1; w_x1 = x1; w_y1 = y1;
        }
      }
    }
    // the last candidate in generation order attaining the minimum: (cost, -index) lexicographic minimum across the wave
    float bc = my_cost;
    int bi = my_idx, nv = my_valid;
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
      const float oc = __shfl_xor(bc, msk, 64);
      const int oi = __shfl_xor(bi, msk, 64);
      nv += __shfl_xor(nv, msk, 64);
      st |= __shfl_xor(st, msk, 64);
      if (oi >= 0 && (bi < 0 || oc < bc || (oc == bc && oi > bi))) { bc = oc; bi = oi; }
    }
    last_best = bi; last_valid = nv;
    if (bi < 0) { st |= 1; break; }                  // no surviving candidate: the reference indexes an empty final_path
    const int wl = bi & 63, wc = bi % nC;            // candidate p was evaluated by lane p % 64; its combo lives in lane p % nC
    s0 = __shfl(c_s1, wc, 64); c_speed = __shfl(c_sd1, wc, 64); c_d = __shfl(w_d1, wl, 64);
    c_d_d = __shfl(w_dd1, wl, 64); c_d_dd = __shfl(w_ddd1, wl, 64);
    const float fx = __shfl(w_x1, wl, 64), fy = __shfl(w_y1, wl, 64);
    ticks = tick + 1;
    if (hist && lane == 0) {
      float* h = hist + ((size_t)tick * n + a) * 8;
      h[0] = s0; h[1] = c_speed; h[2] = c_d; h[3] = c_d_d; h[4] = c_d_dd; h[5] = fx; h[6] = fy; h[7] = bc;
    }
    __builtin_amdgcn_wave_barrier();                 // the table is rewritten by the next tick's phase B
    const double ex = (double)(fx - goal_x), ey = (double)(fy - goal_y);
    if (ex * ex + ey * ey <= 1.0) break;             // :232
  }
  if (lane == 0) {
    state[5 * a] = s0; state[5 * a + 1] = c_speed; state[5 * a + 2] = c_d; state[5 * a + 3] = c_d_d; state[5 * a + 4] = c_d_dd;
    if (ticks_done) ticks_done[a] = ticks;
    if (status) status[a] = st;
    if (best_idx) best_idx[a] = last_best;
    if (n_valid) n_valid[a] = last_valid;
  }
}

}  // namespace crx

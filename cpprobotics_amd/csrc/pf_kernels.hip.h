// pf_kernels.hip.h — batched particle-filter localisation for gfx950: ONE VEHICLE PER WAVEFRONT, its NP particles
// spread over the 64 lanes (NP = 100 -> 36 lanes carry two), wave-wide reductions for the weight sum, the weighted
// mean and covariance, a wave scan + LDS binary search for the low-variance resampling.  All T ticks of a launch are
// fused: the particle set lives in registers (LDS only while resampling), inputs stream in, xEst streams out.
//
// Replaces, for n independent vehicles, /root/reference/src/particle_filter.cpp:
//   motion_model :26-40, gauss_likelihood :53-57, calc_covariance :59-71, pf_localization :73-109,
//   cumsum :111-118, resampling :120-148.
// Random numbers are inputs (the reference draws them from std::mt19937 inside these functions): nrm [T][n][NP][2]
// standard normals for the motion noise (:87-88), uni [T][n][NP] uniforms in [1,2) for the resampling (:133, uni_d{1,2}).
//
// Parity.  Per-particle arithmetic follows the reference statement by statement (double promotions included; cosf / sinf /
// expf are the glibc-exact restatements of crx_trig.h).  The sums over the particles are wave (DPP) reductions — a balanced
// pairwise tree over the lanes — not Eigen's vectorised redux / gemv order, which nobody can restate without Eigen's binary
// (statistical parity with the reference).  Against the CPU oracle evaluated in THIS summation order
// (its oracle_pf_step_wave) the kernel is bit-exact, resampling decisions and ancestors included; against the
// oracle's index-order sums it agrees statistically (a resampling threshold can flip on a tie).
//
// Layout: px [n][NP][4] (Eigen::Matrix<float,4,NP> column-major = one float4 per particle), pw [n][NP], xEst [n][4],
// PEst [n][16] column-major, obs [T][n][L][3] = (range, landmark x, landmark y), nobs [T][n], u [T][n][2].
#pragma once
#include <hip/hip_runtime.h>
#include "crx_trig.h"

namespace crx {

struct PfParams {
  float rsim0, rsim1;   // Rsim(0,0), Rsim(1,1)
  float Q;              // observation variance (gauss_likelihood's sigma = sqrt(Q))
  double dt;
  float nth;            // resampling threshold NTh = NP/2
};

// Wave-wide primitives on DPP (data-parallel primitives: the cross-lane operand is read through the VALU's own
// lane-permute network, one instruction per step, no LDS traffic — `__shfl_*` compiles to ds_bpermute, which made the
// first version of this kernel LDS-pipe bound).  gfx9 control words: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E,
// row_half_mirror = 0x141, row_mirror = 0x140, row_shr:n = 0x110 + n, row_bcast15 = 0x142 (rows 1,3 <- lane 15 of the row
// before), row_bcast31 = 0x143 (rows 2,3 <- lane 31).  Lanes a control word does not cover read `old` (the identity).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float v, float identity = 0.0f) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i(int v, int identity) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
// sum over the 64 lanes, returned in every lane (through an SGPR)
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);            // every lane: the sum of its row of 16
  v += dpp_f<0x142, 0xA>(v);       // rows 1,3 += row before
  v += dpp_f<0x143, 0xC>(v);       // rows 2,3 += rows 0+1  -> lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// inclusive scan over the 64 lanes (lane order)
__device__ __forceinline__ float wave_scan_add(float v, int /*lane*/) {
  v += dpp_f<0x111>(v);
  v += dpp_f<0x112>(v);
  v += dpp_f<0x114>(v);
  v += dpp_f<0x118>(v);            // inclusive within the row of 16
  v += dpp_f<0x142, 0xA>(v);
  v += dpp_f<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ int wave_scan_max(int v, int /*lane*/) {
  const int lowest = -2147483647 - 1;
  auto mx = [](int a, int b) { return a > b ? a : b; };
  v = mx(v, dpp_i<0x111>(v, lowest));
  v = mx(v, dpp_i<0x112>(v, lowest));
  v = mx(v, dpp_i<0x114>(v, lowest));
  v = mx(v, dpp_i<0x118>(v, lowest));
  v = mx(v, dpp_i<0x142, 0xA>(v, lowest));
  v = mx(v, dpp_i<0x143, 0xC>(v, lowest));
  return v;
}

constexpr int kPfWavesPerBlock = 4;

template <int NP>
__global__ void __launch_bounds__(64 * kPfWavesPerBlock)
pf_run_kernel(int n, int T, int L, float* __restrict__ px, float* __restrict__ pw, float* __restrict__ xEst,
              float* __restrict__ PEst, const float* __restrict__ obs, const int* __restrict__ nobs,
              const float* __restrict__ u, const float* __restrict__ nrm, const float* __restrict__ uni, PfParams p,
              float* __restrict__ x_hist, int* __restrict__ n_resampled) {
  static_assert(NP >= 1 && NP <= 128, "one or two particles per lane");
  __shared__ float4 s_x[kPfWavesPerBlock][NP];
  __shared__ float s_wc[kPfWavesPerBlock][NP];
  __shared__ float s_base[NP];
  __shared__ uint64_t s_exp[32];                                   // expf_'s table of 2^(i/32): LDS reads instead of constant-memory loads
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t a = (size_t)blockIdx.x * kPfWavesPerBlock + wv;
  const float inv = (float)(1.0 / NP);
  if (threadIdx.x < 32) s_exp[threadIdx.x] = expf_tab(threadIdx.x);
  if (threadIdx.x == 0) {            // base = cumsum(pw*0.0 + Ones*1.0/NP) - Ones*1.0/NP  (:129), input-independent
    float c = inv;
    s_base[0] = c - inv;
    for (int i = 1; i < NP; ++i) { c = c + inv; s_base[i] = c - inv; }
  }
  __syncthreads();
  if (a >= (size_t)n) return;        // whole waves only: no barrier below this point
  const int p0 = lane, p1 = lane + 64;
  const bool v1 = p1 < NP, v0 = p0 < NP;
  float4 x0 = make_float4(0, 0, 0, 0), x1 = x0;
  float w0 = 0.0f, w1 = 0.0f;
  const float4* pxa = reinterpret_cast<const float4*>(px) + a * NP;
  if (v0) { x0 = pxa[p0]; w0 = pw[a * NP + p0]; }
  if (v1) { x1 = pxa[p1]; w1 = pw[a * NP + p1]; }
  const float sig = sqrtf(p.Q);                                              // std::sqrt(Q) :98
  const double lik_c = 1.0 / sqrt(2.0 * 3.141592653 * (double)sig * (double)sig);   // 1.0 / std::sqrt(2.0*PI*sigma*sigma) :54
  const float lik_d = 2 * sig * sig;                                          // (2 * sigma * sigma), float
  float4 xe = make_float4(0, 0, 0, 0);
  float Pe[10];
  int nres = 0;

  auto advance = [&](float4& x, float& w, const float2 nz, const float u0, const float u1, const float* Z, int nob) {
    const float ud0 = (float)((double)u0 + (double)nz.x * (double)p.rsim0);   // :87-88
    const float ud1 = (float)((double)u1 + (double)nz.y * (double)p.rsim1);
    float sn, cs;
    sincosf_(x.z, &sn, &cs);                                                  // motion_model :26-40
    const float b0 = (float)(p.dt * (double)cs), b1 = (float)(p.dt * (double)sn), b2 = (float)p.dt;
    x = make_float4(x.x + b0 * ud0, x.y + b1 * ud0, x.z + b2 * ud1, x.w + ud0);
    for (int i = 0; i < nob; ++i) {                                           // :92-99
      const float dx = x.x - Z[3 * i + 1], dy = x.y - Z[3 * i + 2];
      const float prez = sqrtf(dx * dx + dy * dy);
      const float dz = prez - Z[3 * i];
      const float pl = (float)(lik_c * (double)expf_(-dz * dz / lik_d, s_exp));       // gauss_likelihood :53-57 (glibc-exact expf, crx_trig.h)
      w = w * pl;
    }
  };

  for (int t = 0; t < T; ++t) {
    const size_t o = (size_t)t * n + a;
    const float u0 = u[2 * o], u1 = u[2 * o + 1];
    const int nob = min(max(nobs[o], 0), L);                 // a caller passing nobs > L would read past its obs rows
    const float* Z = obs + o * (size_t)L * 3;
    const float2* nz = reinterpret_cast<const float2*>(nrm) + o * NP;
    if (v0) advance(x0, w0, nz[p0], u0, u1, Z, nob);
    if (v1) advance(x1, w1, nz[p1], u0, u1, Z, nob);
    // pw = pw / pw.sum()  :104
    const float s = wave_sum(w0 + w1);
    w0 = w0 / s; w1 = w1 / s;
    // xEst = px * pw  :106
    xe.x = wave_sum(x0.x * w0 + x1.x * w1);
    xe.y = wave_sum(x0.y * w0 + x1.y * w1);
    xe.z = wave_sum(x0.z * w0 + x1.z * w1);
    xe.w = wave_sum(x0.w * w0 + x1.w * w1);
    // calc_covariance :59-71 (symmetric: 10 sums)
    {
      const float d0[4] = {x0.x - xe.x, x0.y - xe.y, x0.z - xe.z, x0.w - xe.w};
      const float d1[4] = {x1.x - xe.x, x1.y - xe.y, x1.z - xe.z, x1.w - xe.w};
      int k = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r <= c; ++r) Pe[k++] = wave_sum((w0 * d0[r]) * d0[c] + (v1 ? (w1 * d1[r]) * d1[c] : 0.0f));
    }
    if (x_hist && lane == 0) reinterpret_cast<float4*>(x_hist)[o] = xe;
    // resampling :120-148
    const float ww = wave_sum(w0 * w0 + w1 * w1);
    const float Neff = (float)(1.0 / (double)ww);
    if (Neff < p.nth) {                                   // wave-uniform
      ++nres;
      float c0 = wave_scan_add(w0, lane);                 // cumsum :111-118 (particles 0..63, then 64..NP-1)
      const float tot0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c0), 63));
      float c1 = wave_scan_add(w1, lane) + tot0;
      if (v0) { s_wc[wv][p0] = c0; s_x[wv][p0] = x0; }
      if (v1) { s_wc[wv][p1] = c1; s_x[wv][p1] = x1; }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): LDS writes of this wave visible to its own reads
      const float* un = uni + o * NP;
      auto pick = [&](int pidx) -> int {                  // smallest ind with !(resampleid > wcum[ind]), capped at NP-1 (:139-141)
        const float rid = (float)((double)s_base[pidx] + (double)un[pidx] / NP);   // :133
        int lo = 0, hi = NP - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (rid > s_wc[wv][mid]) lo = mid + 1; else hi = mid;
        }
        return lo;
      };
      int i0 = v0 ? pick(p0) : 0, i1 = v1 ? pick(p1) : 0;
      // `ind` never moves back in the reference's loop: running maximum in particle order
      i0 = wave_scan_max(i0, lane);
      const int m0 = __builtin_amdgcn_readlane(i0, 63);
      i1 = wave_scan_max(i1, lane);
      i1 = i1 > m0 ? i1 : m0;
      if (v0) x0 = s_x[wv][i0];
      if (v1) x1 = s_x[wv][i1];
      w0 = v0 ? inv : 0.0f; w1 = v1 ? inv : 0.0f;         // pw = Ones * 1.0/NP  :146
      __builtin_amdgcn_wave_barrier();
    }
  }
  float4* pxo = reinterpret_cast<float4*>(px) + a * NP;
  if (v0) { pxo[p0] = x0; pw[a * NP + p0] = w0; }
  if (v1) { pxo[p1] = x1; pw[a * NP + p1] = w1; }
  if (lane == 0) {
    reinterpret_cast<float4*>(xEst)[a] = xe;
    float* Pa = PEst + a * 16;
    int k = 0;
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r <= c; ++r) { Pa[r + 4 * c] = Pe[k]; Pa[c + 4 * r] = Pe[k]; ++k; }
    if (n_resampled) n_resampled[a] += nres;
  }
}

}  // namespace crx

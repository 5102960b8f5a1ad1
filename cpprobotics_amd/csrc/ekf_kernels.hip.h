// ekf_kernels.hip.h — batched 4-state EKF predict+update for gfx950 (one vehicle per lane).
//
// Replaces, for n independent vehicles at once, the reference's
//   motion_model / jacobF / observation_model / jacobH / ekf_estimation
//   (/root/reference/src/extended_kalman_filter.cpp:22-78)
// and, in the fused kernel, the body of its simulation loop (:171-188).
//
// Arithmetic contract (bit parity with the Eigen path, see DESIGN.md "EKF arithmetic"):
//   * fp32 mul/add in Eigen's per-coefficient order — first term a bare product, then
//     ascending-k multiply-then-add — with NO fma contraction (-ffp-contract=off);
//   * multiplications by the literal 0/1 entries of F_, jF, jH and I-K*jH are dropped, which
//     is exact for finite operands as long as the surviving terms keep their order;
//   * DT*cos(yaw) and friends are formed in double and rounded once to float, as the
//     reference's `#define DT 0.1` (a double literal) makes them;
//   * sin/cos are crx_trig.h (bit-identical to glibc's sinf/cosf);
//   * the 2x2 inverse is Eigen's closed form with one IEEE division.
//
// Data layout in HBM: x [n][4] (one 16-byte load per lane, fully coalesced); P [n][16]
// column-major per vehicle (the layout of a std::vector<Eigen::Matrix4f>); fused kernel streams
// z,u as time-major [T][n][2] (8-byte loads, coalesced) and writes xEst history [T][n][4]
// (16-byte stores, coalesced).  State and covariance live in VGPRs for all T steps.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "crx_trig.h"
#include "ekf_math.h"

namespace crx {

#ifndef CRX_EKF_RUN_BLOCK
#define CRX_EKF_RUN_BLOCK 64   // threads per workgroup of the fused kernel (a multiple of 64)
#endif

#ifndef CRX_EKF_PIN
#define CRX_EKF_PIN 1   // how the per-step prefetch loads are pinned in the chunk's schedule (see the loop)
#endif

// native 4-wide vector type (the nontemporal builtins do not take HIP's float4 wrapper); v2f: ekf_math.h
typedef float v4f __attribute__((ext_vector_type(4)));

// XCD-aware workgroup order.  The hardware hands consecutive workgroups of a launch to the eight XCDs in turn (workgroup b runs on
// XCD b % 8), so with the identity mapping every XCD streams every eighth 512-byte .. 4-KiB piece of each row of the time-major
// arrays.  Long-lived streaming writers do measurably better when every XCD owns ONE contiguous eighth of the row (plain streaming
// kernel, write only, 2,048 persistent workgroups: 4.3 -> 5.9 TB/s; profiles/r04/hbm_xcd_order.jsonl): workgroup b therefore
// works on block  start(b % 8) + b / 8,  start(k) = k * (nb / 8) + min(k, nb % 8)  — a permutation of [0, nb) for any nb.
#ifndef CRX_XCD_ORDER
#define CRX_XCD_ORDER 1
#endif
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nb) {
#if CRX_XCD_ORDER
  const unsigned k = b & 7u, q = nb >> 3, r = nb & 7u;
  return k * q + (k < r ? k : r) + (b >> 3);
#else
  return b;
#endif
}

__device__ __forceinline__ void load_state(EkfState& s, const float* __restrict__ x,
                                           const float* __restrict__ P, size_t a) {
  const float4 xv = reinterpret_cast<const float4*>(x)[a];
  s.x0 = xv.x; s.x1 = xv.y; s.x2 = xv.z; s.x3 = xv.w;
  const float4* Pv = reinterpret_cast<const float4*>(P) + 4 * a;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 c = Pv[j];
    s.P[4 * j + 0] = c.x; s.P[4 * j + 1] = c.y; s.P[4 * j + 2] = c.z; s.P[4 * j + 3] = c.w;
  }
}

__device__ __forceinline__ void store_P(const EkfState& s, float* __restrict__ P, size_t a) {
  float4* Pv = reinterpret_cast<float4*>(P) + 4 * a;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    Pv[j] = make_float4(s.P[4 * j + 0], s.P[4 * j + 1], s.P[4 * j + 2], s.P[4 * j + 3]);
}

// ---- single step: n vehicles, one ekf_estimation() each -------------------------------------
// HBM-bound (176 B per update).  A wave's 64 covariances are one contiguous 4-KiB span; read or written straight from
// per-lane registers, each 16-byte access lands in a different 64-byte block (32 cache lines per instruction).  Full
// waves therefore move the span in four 1-KiB row pieces (16 contiguous bytes per lane) and transpose through a
// wave-private LDS tile (lane stride 80 B: conflict-free b128 accesses); a ragged last wave uses the direct form.
// The step itself is the fused kernel's packed fast step (ekf_math.h) since round 4; a wave with a lane outside its domain
// (yaw = +-0, |yaw| >= 120, extreme determinant) reloads its input — nothing has been stored yet — and takes the general step:
// the same bits either way.
// NT: how the covariance rows pass the caches.  When a filter loop launches step after step on a batch whose state (160 B per
// vehicle) is well beyond the 256-MB Infinity Cache, the dirty lines the previous launch left behind are in the way of the next
// one; nontemporal loads / stores of the 4-KiB covariance rows avoid that — back-to-back launches, 4 M vehicles 0.145 -> 0.118 ms,
// 8 M 0.277 -> 0.217 ms (6.8 TB/s) — while batches the cache holds lose 1-2 % (crossover measured at ~3 M vehicles;
// x, z, u nontemporal as well: slower everywhere; profiles/r04/ekf_step_ab.jsonl).  The host picks by batch size (kEkfStepNtMinN).
#ifndef CRX_EKF_STEP_BLOCK
#define CRX_EKF_STEP_BLOCK 256      // lanes per workgroup (a multiple of 64); the LDS tile is private to a wave
#endif
constexpr int kEkfStepNtMinN = 3 << 20;    // 3 M vehicles: 503 MB of state
template <bool NT, bool DTS>
__global__ void __launch_bounds__(CRX_EKF_STEP_BLOCK)
ekf_step_kernel(int n, float* __restrict__ x, float* __restrict__ P, const float* __restrict__ z,
                const float* __restrict__ u, EkfConsts k) {
  __shared__ v4f s_pt[(CRX_EKF_STEP_BLOCK / 64) * 64 * 5];
  const size_t a = (size_t)xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const size_t wave0 = a - lane;                       // first vehicle of this wave
  const bool full_wave = wave0 + 64 <= (size_t)n;
  if (a >= (size_t)n) return;
  EkfState s;
  v4f* __restrict__ tile = s_pt + wv * (64 * 5);
  v4f* __restrict__ xp = reinterpret_cast<v4f*>(x) + a;
  if (full_wave) {
    const v4f xv = *xp;
    s.x0 = xv.x; s.x1 = xv.y; s.x2 = xv.z; s.x3 = xv.w;
    const v4f* __restrict__ row = reinterpret_cast<const v4f*>(P) + 4 * wave0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      tile[(16u * kk + (lane >> 2)) * 5u + (lane & 3u)] = NT ? __builtin_nontemporal_load(row + 64 * kk + lane) : row[64 * kk + lane];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const v4f c = tile[lane * 5u + j];
      s.P[4 * j + 0] = c.x; s.P[4 * j + 1] = c.y; s.P[4 * j + 2] = c.z; s.P[4 * j + 3] = c.w;
    }
  } else {
    load_state(s, x, P, a);
  }
  const v2f* __restrict__ zp = reinterpret_cast<const v2f*>(z) + a;
  const v2f* __restrict__ up = reinterpret_cast<const v2f*>(u) + a;
  const v2f zv = *zp, uv = *up;
  {
    EkfStateP sp;
    pack_state(sp, s);
    FastDomain dom = fast_domain_init();
    ekf_step_packed<DTS>(sp, zv, uv, pack_consts(k), dom);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast_domain_ok(dom)) != 0, 0)) {
      ekf_step_dev(s, zv.x, zv.y, uv.x, uv.y, k);          // `s` still holds the input: the packed step worked on its copy `sp`
    } else {
      unpack_state(s, sp);
    }
  }
  *xp = v4f{s.x0, s.x1, s.x2, s.x3};
  if (full_wave) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[lane * 5u + j] = v4f{s.P[4 * j + 0], s.P[4 * j + 1], s.P[4 * j + 2], s.P[4 * j + 3]};
    __builtin_amdgcn_wave_barrier();
    v4f* __restrict__ row = reinterpret_cast<v4f*>(P) + 4 * wave0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const v4f po = tile[(16u * kk + (lane >> 2)) * 5u + (lane & 3u)];
      if (NT) __builtin_nontemporal_store(po, row + 64 * kk + lane); else row[64 * kk + lane] = po;
    }
  } else {
    store_P(s, P, a);
  }
}

// ---- fused T steps: state/covariance stay in registers, z/u stream in, xEst streams out -----
// (fast packed step + general step: ekf_math.h)
// D = steps per chunk = software prefetch distance (z,u for step t+D are requested while step t runs).
// BUF = true: the streams are addressed through buffer descriptors — a wave-uniform base (4 SGPRs,
// advanced once per chunk), a per-step scalar offset and one fixed per-lane byte offset — so the
// loads/stores of the hot loop need no vector address arithmetic at all.  The stores keep their step
// offset in (loop-invariant) VGPRs and pass soffset = 0: a 16-byte buffer store WITH a scalar-register
// soffset followed closely by a VALU write of its data registers returned corrupted data on gfx950 (the
// compiler's hazard recogniser only protects the no-soffset form).  Buffer offsets are 32-bit:
// the host picks BUF only while every offset of a chunk stays below 2^31 (n <= kEkfBufMaxN).
constexpr int kEkfBufMaxN = 1 << 22;
#ifndef CRX_EKF_PHIST_LDS
#define CRX_EKF_PHIST_LDS 1
#endif

#ifdef CRX_EKF_TIMING   // debug builds only (scripts/ubench): per-workgroup shader-clock / real-time deltas
__device__ long long g_ekf_timing[4096][2];
#endif

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t stream_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

// FMA: the packed step's multiply-then-add pairs contracted (ekf_math.h: ekf_step_packed<DTS, FMA>; crx_ekf_params.arith =
// CRX_ARITH_CONTRACT).  The rare general steps (a wave outside the fast domain, the last < D steps of a launch) stay unfused.
template <int D, bool XHIST, bool PHIST, bool BUF, bool DTS, bool FMA = false>
__global__ void __launch_bounds__(CRX_EKF_RUN_BLOCK)
ekf_run_kernel(int n, int T, float* __restrict__ x, float* __restrict__ P,
               const float* __restrict__ z, const float* __restrict__ u,
               float* __restrict__ x_hist, float* __restrict__ P_hist, EkfConsts k) {
  __shared__ v4f s_pt[(PHIST && BUF && CRX_EKF_PHIST_LDS) ? CRX_EKF_RUN_BLOCK * 5 : 1];   // P_hist transpose staging (one wave)
  const size_t blk = (size_t)xcd_block(blockIdx.x, gridDim.x) * CRX_EKF_RUN_BLOCK;
  const unsigned lane = threadIdx.x;
  const size_t a = blk + lane;
  if (a >= (size_t)n) return;
#ifdef CRX_EKF_TIMING
  const long long tm0 = clock64(), tr0 = wall_clock64();
#endif
  // per-workgroup views of the time-major streams: element [t*n + lane] belongs to this lane at step t
  const v2f* __restrict__ z2 = reinterpret_cast<const v2f*>(z) + blk;
  const v2f* __restrict__ u2 = reinterpret_cast<const v2f*>(u) + blk;
  v4f* __restrict__ xh = reinterpret_cast<v4f*>(x_hist) + blk;
  v4f* __restrict__ Ph = reinterpret_cast<v4f*>(P_hist) + 4 * blk;
  static_assert(!CRX_EKF_PHIST_LDS || CRX_EKF_RUN_BLOCK == 64, "the P_hist transpose assumes one wave per workgroup");
  const bool full_wave = blk + 64 <= (size_t)n;   // a ragged last wave stores its covariances directly
  const EkfConstsP kp = pack_consts(k);
  const size_t ns = (size_t)n;
  const unsigned un = (unsigned)n;

  v2f zq[D], uq[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < T) {
      zq[d] = __builtin_nontemporal_load(&(z2 + d * ns)[lane]);
      uq[d] = __builtin_nontemporal_load(&(u2 + d * ns)[lane]);
    }
  }
  EkfState s;
  load_state(s, x, P, a);
  EkfStateP sp;
  pack_state(sp, s);
  // Let the initial state arrive before the loop: otherwise the compiler's wait-count bookkeeping,
  // merged over the loop back-edge, puts a full vmcnt(0) drain inside every iteration.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched

  // main part: every chunk of D steps whose prefetches (t + D) stay inside [0, T).  One basic block
  // per chunk; the D loads of the next chunk stay in flight across it (counted vmcnt waits).
  int t0 = 0;
  // LAST: the one chunk behind the main part whose prefetches would partly run past step T - 1 — the same fast steps, the refill of a
  // slot guarded (until round 5 these up to D steps went through the general step of the tail: 8 of the headline's 1000, 4 of the
  // 100 of a configs[4] round, at 2-3x the cost each)
  auto chunk = [&](auto last) {
    constexpr bool LAST = decltype(last)::value;
    const EkfStateP s_in = sp;
    FastDomain dom = fast_domain_init();
    __amdgpu_buffer_rsrc_t rz, ru, rx, rp;
    if (BUF) {
      rz = stream_rsrc(z2 + (size_t)t0 * ns);
      ru = stream_rsrc(u2 + (size_t)t0 * ns);
      if (XHIST) rx = stream_rsrc(xh + (size_t)t0 * ns);
      if (PHIST) rp = stream_rsrc(Ph + 4 * (size_t)t0 * ns);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const size_t t = (size_t)(t0 + d);
      ekf_step_packed<DTS, FMA>(sp, zq[d], uq[d], kp, dom);
      // refill the slot just consumed (its registers are dead now: no copy at the loop back-edge)
      if (!LAST || t0 + d + D < T) {
        if (BUF) {
          zq[d] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rz, lane * 8u, (unsigned)(d + D) * un * 8u, 2));
          uq[d] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ru, lane * 8u, (unsigned)(d + D) * un * 8u, 2));
        } else {
          zq[d] = __builtin_nontemporal_load(&(z2 + (t + D) * ns)[lane]);
          uq[d] = __builtin_nontemporal_load(&(u2 + (t + D) * ns)[lane]);
        }
      }
      if (XHIST) {
        const v4f xo = v4f{sp.x01.x, sp.x01.y, sp.x23.x, sp.x23.y};
        if (BUF) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, xo), rx, lane * 16u + (unsigned)d * un * 16u, 0, 2);
        else __builtin_nontemporal_store(xo, &(xh + t * ns)[lane]);
      }
      if (PHIST) {
        if (BUF && CRX_EKF_PHIST_LDS && full_wave) {
          // The wave's 64 covariances are one contiguous 4-KiB row of P_hist.  Written straight from registers, every
          // store instruction scatters 64 16-byte pieces over 32 cache lines; transposed through LDS (lane stride 80 B:
          // conflict-free b128 writes), store k instead covers bytes [1024 k, 1024 (k+1)) of the row, 16 per lane.
#pragma unroll
          for (int j = 0; j < 4; ++j)
            s_pt[lane * 5u + j] = v4f{sp.Plo[j].x, sp.Plo[j].y, sp.Phi[j].x, sp.Phi[j].y};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const v4f po = s_pt[(16u * kk + (lane >> 2)) * 5u + (lane & 3u)];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, po), rp, lane * 16u + 1024u * kk + (unsigned)d * un * 64u, 0, 2);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const v4f po = v4f{sp.Plo[j].x, sp.Plo[j].y, sp.Phi[j].x, sp.Phi[j].y};
            // plain (not nontemporal) stores: a lane's 64-byte column block goes out as four 16-byte pieces
            // that the L2 has to merge into full lines; streaming stores do not get merged (10x slower)
            if (BUF) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, po), rp, lane * 64u + (unsigned)d * un * 64u + 16u * j, 0, 0);
            else (Ph + 4 * t * ns)[4 * lane + j] = po;
          }
        }
      }
#if CRX_EKF_PIN == 1
      // ALU work may be scheduled across this point (the trig of step d+1 overlaps the covariance
      // update of step d), memory operations may not: keeps each step's prefetch from sinking to the
      // end of the chunk.
      __builtin_amdgcn_sched_barrier(0x7);
#elif CRX_EKF_PIN == 2
      __builtin_amdgcn_sched_barrier(0);   // nothing crosses: one scheduling region per step
#endif
    }
    // rare: some lane of this wave left the fast domain (yaw = 0 or |yaw| >= 120, extreme det):
    // redo the chunk for the whole wave with the general step (bit-identical for the other lanes)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast_domain_ok(dom)) != 0, 0)) {
      unpack_state(s, s_in);
      for (int d = 0; d < D; ++d) {
        const size_t o = (size_t)(t0 + d) * ns + lane;
        const v2f zc = z2[o], uc = u2[o];
        ekf_step_dev(s, zc.x, zc.y, uc.x, uc.y, k);
        if (XHIST) xh[o] = v4f{s.x0, s.x1, s.x2, s.x3};
        if (PHIST) store_P(s, P_hist, (size_t)(t0 + d) * ns + a);
      }
      pack_state(sp, s);
    }
  };
  for (; t0 + 2 * D <= T; t0 += D) chunk(std::false_type{});
  if (t0 + D <= T) { chunk(std::true_type{}); t0 += D; }
  // tail: fewer than D steps left — general step, guarded prefetch
  unpack_state(s, sp);
  for (; t0 < T; t0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int t = t0 + d;
      if (t < T) {
        const v2f zc = zq[d], uc = uq[d];
        if (t + D < T) {
          zq[d] = __builtin_nontemporal_load(&(z2 + (size_t)(t + D) * ns)[lane]);
          uq[d] = __builtin_nontemporal_load(&(u2 + (size_t)(t + D) * ns)[lane]);
        }
        ekf_step_dev(s, zc.x, zc.y, uc.x, uc.y, k);
        if (XHIST)
          __builtin_nontemporal_store(v4f{s.x0, s.x1, s.x2, s.x3}, &(xh + (size_t)t * ns)[lane]);
        if (PHIST) store_P(s, P_hist, (size_t)t * ns + a);
      }
    }
  }
  reinterpret_cast<float4*>(x)[a] = make_float4(s.x0, s.x1, s.x2, s.x3);
  store_P(s, P, a);
#ifdef CRX_EKF_TIMING
  if (lane == 0 && blockIdx.x < 4096) {
    g_ekf_timing[blockIdx.x][0] = clock64() - tm0;
    g_ekf_timing[blockIdx.x][1] = wall_clock64() - tr0;
  }
#endif
}

// ---- the small reference functions, batched (drop-in completeness) ---------------------------
__global__ void motion_model_kernel(int n, const float* __restrict__ x, const float* __restrict__ u,
                                    float* __restrict__ x_out, double dt) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  float4 xv = reinterpret_cast<const float4*>(x)[a];
  const float2 uv = reinterpret_cast<const float2*>(u)[a];
  motion_model_dev(xv.x, xv.y, xv.z, xv.w, uv.x, uv.y, dt);
  reinterpret_cast<float4*>(x_out)[a] = xv;
}

__global__ void jacobF_kernel(int n, const float* __restrict__ x, const float* __restrict__ u,
                              float* __restrict__ jF, double dt) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  const float4 xv = reinterpret_cast<const float4*>(x)[a];
  const float2 uv = reinterpret_cast<const float2*>(u)[a];
  const JacF j = jacobF_dev(xv.z, uv.x, dt);
  float4* o = reinterpret_cast<float4*>(jF) + 4 * a;
  o[0] = make_float4(1.f, 0.f, 0.f, 0.f);
  o[1] = make_float4(0.f, 1.f, 0.f, 0.f);
  o[2] = make_float4(j.j02, j.j12, 1.f, 0.f);
  o[3] = make_float4(j.j03, j.j13, 0.f, 1.f);
}

__global__ void observation_model_kernel(int n, const float* __restrict__ x, float* __restrict__ zo) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  const float4 xv = reinterpret_cast<const float4*>(x)[a];
  reinterpret_cast<float2*>(zo)[a] = make_float2(xv.x, xv.y);
}

// Input side of the simulation loop (:174-181): ud, xTrue, xDR, z from caller-supplied normals.
template <bool HIST>
__global__ void __launch_bounds__(64)
ekf_simulate_inputs_kernel(int n, int T, const float* __restrict__ u_true, float* __restrict__ xTrue,
                           float* __restrict__ xDR, const float* __restrict__ w,
                           float* __restrict__ z, float* __restrict__ ud,
                           float* __restrict__ xTrue_hist, float* __restrict__ xDR_hist,
                           float q0, float q1, float r0, float r1, double dt) {
  const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (size_t)n) return;
  const float2 ut = reinterpret_cast<const float2*>(u_true)[a];
  float4 xt = reinterpret_cast<const float4*>(xTrue)[a];
  float4 xd = reinterpret_cast<const float4*>(xDR)[a];
  const float4* w4 = reinterpret_cast<const float4*>(w);
  for (int t = 0; t < T; ++t) {
    const size_t o = (size_t)t * n + a;
    const float4 wv = w4[o];
    const float ud0 = (float)((double)ut.x + (double)wv.x * (double)q0);
    const float ud1 = (float)((double)ut.y + (double)wv.y * (double)q1);
    motion_model_dev(xt.x, xt.y, xt.z, xt.w, ut.x, ut.y, dt);
    motion_model_dev(xd.x, xd.y, xd.z, xd.w, ud0, ud1, dt);
    const float z0 = (float)((double)xt.x + (double)wv.z * (double)r0);
    const float z1 = (float)((double)xt.y + (double)wv.w * (double)r1);
    reinterpret_cast<float2*>(z)[o] = make_float2(z0, z1);
    reinterpret_cast<float2*>(ud)[o] = make_float2(ud0, ud1);
    if (HIST) {
      if (xTrue_hist) reinterpret_cast<float4*>(xTrue_hist)[o] = xt;
      if (xDR_hist) reinterpret_cast<float4*>(xDR_hist)[o] = xd;
    }
  }
  reinterpret_cast<float4*>(xTrue)[a] = xt;
  reinterpret_cast<float4*>(xDR)[a] = xd;
}

}  // namespace crx

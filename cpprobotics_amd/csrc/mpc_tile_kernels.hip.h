// mpc_tile_kernels.hip.h — the MPC horizon solve with its working set ON THE CHIP (round 6): LDS + accumulator registers.
//
// Same NLP, same solver, same arithmetic as mpc_kernels.hip.h (mpc_solve_lane<.., STORE = 1>; reference:
// /root/reference/src/model_predictive_control.cpp:199-346) — what changes is where a lane keeps its problem.  mpc_kernel keeps
// 3.8 KB per lane in private memory: at one 256-VGPR wave per SIMD that is 1,024 resident waves x 243 KB, the whole Infinity Cache,
// and from ~65 k agents in flight the solve is bound by the HBM traffic of its own scratch (82 KB per solve against 848 algorithmic
// bytes, profiles/mpc_traffic.json of round 5).  A gfx950 CU offers two stores the private segment does not use:
//
//   * LDS, 160 KB per CU = 40 KB for each of four single-wave workgroups = 640 B per lane: both control buffers
//     (U[2][20] x (delta, a) doubles), laid out [buffer][stage][lane] as double2 — every access one conflict-free ds_read/write_b128;
//   * the accumulator half of the register file: a wave alone on its SIMD owns 512 registers per lane, the solver's code can
//     name 256 of them — a40 .. a255 hold the float feedback gains of stages 1 .. 18 (12 per stage; mpc_agpr.inc).  They are
//     the largest array of the solve (31 % of its traffic) and are read back with no latency at all.
//
// What stays in private memory: the knots (two buffers), the feed-forward steps and the gains of stage 19.  One agent per lane, full waves, lockstep sweeps —
// per agent the same operations in the same order as mpc_kernel: BIT-IDENTICAL outputs (tests/test_mpc_gpu.py).
// Horizons of at most kMpcTileStages = 20 stages (T <= 21: BASELINE configs[3] / configs[4] and the reference's own T 6); longer ones
// take mpc_kernel.
//
// Safety of the register block: every statement that touches a40 .. a255 names its registers literally; mpc_agpr_reserve() lists all
// of them as clobbers, which makes the kernel descriptor allocate them.  It does NOT keep the compiler out: its register allocator
// may place a short-lived value of its own in any accumulator register between two of those statements (it takes them in ascending
// order — the first build of this kernel, with the block at a28, had it in a28 .. a32).  So the block starts above what the
// allocator needs (a0 .. a32 today) and the BUILD checks it: scripts/check_isa.py disassembles the tile kernels and demands that
// a40 .. a255 appear only in complete, slot-aligned runs of twelve v_accvgpr_write / v_accvgpr_read — a compiler bump or an edit
// that raises the register pressure into the block fails __graft_entry__.build(), it does not corrupt a solve.
#pragma once
#include "mpc_kernels.hip.h"

namespace crx {

template <int MAXT>
__global__ void __launch_bounds__(64)       // one wave per workgroup: four workgroups (one per SIMD) share a CU's 160 KB of LDS
mpc_tile_kernel(int n, int T, const float* __restrict__ x0g, const float* __restrict__ xrefg, MpcP p,
                float* __restrict__ solg, int* __restrict__ statusg, double* __restrict__ costg) {
  __shared__ mpc_d2_t tile_u[2 * kMpcTileStages * 64];          // 40,960 B
  mpc_agpr_reserve();
  const size_t agent = (size_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = agent < (size_t)n;
  const size_t ag = live ? agent : 0;
  const float4* __restrict__ xr4 = reinterpret_cast<const float4*>(xrefg) + ag * (size_t)T;
  const float4 xi = reinterpret_cast<const float4*>(x0g)[ag];
  const size_t nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
  int status; double J; float a0, d0;
  mpc_solve_lane<MAXT, false, false, true, 1>(live, T, xi, xr4, p, live ? solg + agent * nv : nullptr, status, J, a0, d0, MpcFeed{},
                                              MpcTile{(lds_double2_t*)tile_u});
  if (!live) return;
  if (statusg) statusg[agent] = status;
  if (costg) costg[agent] = J;
}

// The tile layout with the lanes REFILLED (mpc_solve_lane<.., REFILL, .., STORE = 1>): wave w owns the agents [w * chunk, (w + 1) * chunk);
// a lane whose agent has converged holds it until `hold` lanes of the wave hold one, then they write their solutions and take the
// wave's next agents; the line search is scheduled asynchronously across the lanes (mpc_kernels.hip.h).  A lockstep wave lasts as long
// as the slowest of its 64 agents — mean of the wave maximum 11.8 sweeps against a mean of 6.8 — and the launch as long as its unluckiest
// SIMD's queue of waves; here every lane is busy until the wave's range is exhausted and all waves end together.  Rounds 4 and 5
// measured this schedule on the private-memory layout and dropped it (1.14x, then 0.89-1.00x: where waves queue, the memory system was
// the limit, and a long-lived wave never hands its share of the caches on).  With the working set mostly on the chip that limit is gone.
// Per agent the same sweeps in the same order: bit-identical to mpc_kernel.
template <int MAXT>
__global__ void __launch_bounds__(64)
mpc_tile_refill_kernel(int n, int T, int chunk, int hold, const float* __restrict__ x0g, const float* __restrict__ xrefg, MpcP p,
                       float* __restrict__ solg, int* __restrict__ statusg, double* __restrict__ costg) {
  __shared__ mpc_d2_t tile_u[2 * kMpcTileStages * 64];
  mpc_agpr_reserve();
  const int lo = (int)blockIdx.x * chunk;
  const MpcFeed feed{lo, (n - lo < chunk) ? n : lo + chunk, hold, x0g, xrefg, solg, statusg, costg};
  int status; double J; float a0, d0;
  mpc_solve_lane<MAXT, false, true, true, 1>(false, T, float4{0.f, 0.f, 0.f, 0.f}, nullptr, p, nullptr, status, J, a0, d0, feed,
                                             MpcTile{(lds_double2_t*)tile_u});
}
// crx_mpc_solve_batch_dev takes the tile layout from this many agents on (csrc/api_mpc.inl has the numbers)
constexpr int kMpcTileFrom = 131072;
// a geometry for the refilled launch: as many waves as the chip has SIMDs (one persistent wave each), at least 128 agents per wave
inline int mpc_tile_refill_chunk(int n) {
  const int c = (n + 1023) / 1024;
  return c < 128 ? 128 : c;
}
inline hipError_t mpc_tile_refill_launch(int n, int T, const float* x0, const float* xref, const crx_mpc_params& q, float* sol,
                                         int* status, double* cost, hipStream_t stream, int chunk, int hold) {
  const MpcP p = mpc_pack(q);
  const dim3 grid((unsigned)(((size_t)n + chunk - 1) / chunk)), block(64);
  hipLaunchKernelGGL((mpc_tile_refill_kernel<24>), grid, block, 0, stream, n, T, chunk, hold, x0, xref, p, sol, status, cost);
  return hipGetLastError();
}

// T - 1 <= kMpcTileStages
inline hipError_t mpc_tile_launch(int n, int T, const float* x0, const float* xref, const crx_mpc_params& q, float* sol,
                                  int* status, double* cost, hipStream_t stream) {
  const MpcP p = mpc_pack(q);
  const dim3 grid((unsigned)(((size_t)n + 63) / 64)), block(64);
  // one instantiation for every horizon it takes: the <8> build of mpc_solve_lane (fully unrolled short loops) needs more accumulator
  // registers of its own than the block leaves free (scripts/check_isa.py caught it)
  hipLaunchKernelGGL((mpc_tile_kernel<24>), grid, block, 0, stream, n, T, x0, xref, p, sol, status, cost);
  return hipGetLastError();
}

}  // namespace crx

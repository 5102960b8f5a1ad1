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
// Safety of the register block: every statement that touches the block names its registers literally; mpc_agpr_reserve() lists all
// of them as clobbers, which makes the kernel descriptor allocate them.  That alone does NOT keep the compiler out: its register
// allocator is free to place a value of its own in any accumulator register between two of those statements, and it does so
// opportunistically — the first build of the checkpointed layout, with the block at a40, had loads landing in a130 .. a137; moved to
// a52 and a64, the allocator followed.  The fence is the LLVM function attribute "amdgpu-agpr-alloc"="B,B" (B = the block's base):
// the allocator may then use a0 .. a(B-1) only and spills anything beyond to private memory.  clang has no spelling for that
// attribute, so the tile kernels are their own translation unit (mpc_tile_module.hip), compiled to LLVM IR, patched
// (scripts/patch_agpr_alloc.py), compiled to a gfx950 code object and embedded in libcrx.so (section .crx_tile_hsaco; csrc/Makefile);
// the library loads it with hipModuleLoadData on first use.  The BUILD still checks the result: scripts/check_isa.py disassembles
// the tile kernels and demands that the block appear only in complete, slot-aligned runs of twelve v_accvgpr_write / v_accvgpr_read.
#pragma once
#include "mpc_kernels.hip.h"

namespace crx {

// the argument block of every tile kernel (= its kernarg segment: the kernels take it by value, the host hands it to
// hipModuleLaunchKernel as a buffer); chunk / hold: the refilled launch's geometry, unused by the lockstep kernels
struct MpcTileArgs {
  int n, T, chunk, hold;
  const float* x0g; const float* xrefg;
  MpcP p;
  float* solg; int* statusg; double* costg;
  // the phased launch (mpc_kernels.hip.h: MpcPhase): the agents of this phase (list == nullptr: all n, from the zero guess), the sweep
  // index they resume at / suspend at, the state records
  const int* list; const int* count; int resume, cap; double* state;
};

#ifdef CRX_MPC_TILE_MODULE     // ---- device side: compiled by mpc_tile_module.hip only ----------------------------------------------

// STORE 1: knots private, two buffers; 2: every second knot, one buffer (mpc_kernels.hip.h).  One wave per workgroup: four workgroups
// (one per SIMD) share a CU's 160 KB of LDS.
template <int MAXT, int STORE>
__device__ __forceinline__ void mpc_tile_body(const MpcTileArgs& a) {
  __shared__ mpc_d2_t tile_u[2 * kMpcTileStages * 64];          // 40,960 B
  if constexpr (STORE == 3) mpc_agpr_reserve_lite(); else mpc_agpr_reserve();
  const int n = a.n, T = a.T;
  const float* __restrict__ x0g = a.x0g; const float* __restrict__ xrefg = a.xrefg;
  float* __restrict__ solg = a.solg; int* __restrict__ statusg = a.statusg; double* __restrict__ costg = a.costg;
  const size_t agent = (size_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = agent < (size_t)n;
  const size_t ag = live ? agent : 0;
  const float4* __restrict__ xr4 = reinterpret_cast<const float4*>(xrefg) + ag * (size_t)T;
  const float4 xi = reinterpret_cast<const float4*>(x0g)[ag];
  const size_t nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
  int status; double J; float a0, d0;
  mpc_solve_lane<MAXT, false, false, true, STORE>(live, T, xi, xr4, a.p, live ? solg + agent * nv : nullptr, status, J, a0, d0, MpcFeed{},
                                                  MpcTile{(lds_double2_t*)tile_u});
  if (!live) return;
  if (statusg) statusg[agent] = status;
  if (costg) costg[agent] = J;
}

// the tile layout in PHASES (mpc_phase_kernel's twin): entries [64 w, 64 w + 64) of the phase's list on wave w
template <int MAXT>
__device__ __forceinline__ void mpc_tile_phase_body(const MpcTileArgs& a) {
  __shared__ mpc_d2_t tile_u[2 * kMpcTileStages * 64];
  mpc_agpr_reserve();
  const int T = a.T;
  const int cnt = a.list ? *a.count : a.n;
  const int entry = (int)blockIdx.x * 64 + (int)threadIdx.x;
  if ((int)blockIdx.x * 64 >= cnt) return;
  const bool live = entry < cnt;
  const size_t agent = live ? (size_t)(a.list ? a.list[entry] : entry) : (size_t)(a.list ? a.list[0] : 0);
  const float4* __restrict__ xr4 = reinterpret_cast<const float4*>(a.xrefg) + agent * (size_t)T;
  const float4 xi = reinterpret_cast<const float4*>(a.x0g)[agent];
  const size_t nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
  int status; double J; float a0, d0;
  mpc_solve_lane<MAXT, false, false, true, 1, true>(live, T, xi, xr4, a.p, live ? a.solg + agent * nv : nullptr, status, J, a0, d0, MpcFeed{},
                                                    MpcTile{(lds_double2_t*)tile_u},
                                                    MpcPhase{a.cap, a.resume, a.state + agent * (size_t)mpc_phase_record_doubles(T)});
  if (!live) return;
  a.statusg[agent] = status;
  if (status != kMpcSuspended && a.costg) a.costg[agent] = J;
}

// The tile layout with the lanes REFILLED (mpc_solve_lane<.., REFILL, .., STORE>): wave w owns the agents [w * chunk, (w + 1) * chunk);
// a lane whose agent has converged holds it until `hold` lanes of the wave hold one, then they write their solutions and take the
// wave's next agents; the line search is scheduled asynchronously across the lanes (mpc_kernels.hip.h).  A lockstep wave lasts as long
// as the slowest of its 64 agents — mean of the wave maximum 11.8 sweeps against a mean of 6.8 — and the launch as long as its unluckiest
// SIMD's queue of waves; here every lane is busy until the wave's range is exhausted and all waves end together.  Rounds 4 and 5
// measured this schedule on the private-memory layout and dropped it (1.14x, then 0.89-1.00x: where waves queue, the memory system was
// the limit, and a long-lived wave never hands its share of the caches on).  With the working set mostly on the chip that limit is gone.
// Per agent the same sweeps in the same order: bit-identical to mpc_kernel.
template <int MAXT, int STORE>
__device__ __forceinline__ void mpc_tile_refill_body(const MpcTileArgs& a) {
  __shared__ mpc_d2_t tile_u[2 * kMpcTileStages * 64];
  mpc_agpr_reserve();
  const int lo = (int)blockIdx.x * a.chunk;
  const MpcFeed feed{lo, (a.n - lo < a.chunk) ? a.n : lo + a.chunk, a.hold, a.x0g, a.xrefg, a.solg, a.statusg, a.costg};
  int status; double J; float a0, d0;
  mpc_solve_lane<MAXT, false, true, true, STORE>(false, a.T, float4{0.f, 0.f, 0.f, 0.f}, nullptr, a.p, nullptr, status, J, a0, d0, feed,
                                                 MpcTile{(lds_double2_t*)tile_u});
}

#else                          // ---- host side: the library's launchers (crx_api.hip) -----------------------------------------------

// crx_mpc_solve_batch_dev takes the tile layout from this many agents on (csrc/api_mpc.inl has the numbers)
constexpr int kMpcTileFrom = 131072;
// ... and the LITE tile layout (mpc_kernels.hip.h: STORE = 3) from this many; from this many when the caller's launches share the GPU
constexpr int kMpcTileLiteFrom = 49152;
constexpr int kMpcTileLiteFromShared = 16384;
// a geometry for the refilled launch: as many waves as the chip has SIMDs (one persistent wave each), at least 128 agents per wave
inline int mpc_tile_refill_chunk(int n) {
  const int c = (n + 1023) / 1024;
  return c < 128 ? 128 : c;
}

// the embedded code object (csrc/Makefile: mpc_tile_hsaco.inc) and its kernels, loaded once per device
extern "C" __attribute__((section(".crx_tile_hsaco"))) const unsigned char crx_tile_hsaco[];
extern "C" const unsigned int crx_tile_hsaco_len;
enum { kTileLockstep1 = 0, kTileLockstep2, kTileRefill1, kTileRefill2, kTilePhase1, kTileLite, kTileKernels };
struct MpcTileModule { hipModule_t mod = nullptr; hipFunction_t fn[kTileKernels] = {}; hipError_t err = hipSuccess; bool tried = false; };
inline hipError_t mpc_tile_function(int which, hipFunction_t* out) {
  static std::mutex mu;
  static MpcTileModule mods[64];                      // by device ordinal
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> l(mu);
  MpcTileModule& m = mods[dev];
  if (!m.tried) {
    m.tried = true;
    static const char* const names[kTileKernels] = {"crx_mpc_tile_kernel_s1", "crx_mpc_tile_kernel_s2", "crx_mpc_tile_refill_kernel_s1",
                                                    "crx_mpc_tile_refill_kernel_s2", "crx_mpc_tile_phase_kernel_s1", "crx_mpc_tile_lite_kernel"};
    m.err = hipModuleLoadData(&m.mod, crx_tile_hsaco);
    for (int k = 0; k < kTileKernels && m.err == hipSuccess; ++k) m.err = hipModuleGetFunction(&m.fn[k], m.mod, names[k]);
  }
  if (m.err != hipSuccess) return m.err;
  *out = m.fn[which];
  return hipSuccess;
}
inline hipError_t mpc_tile_module_launch(int which, unsigned grid, MpcTileArgs args, hipStream_t stream) {
  hipFunction_t f;
  hipError_t e = mpc_tile_function(which, &f);
  if (e != hipSuccess) return e;
  size_t size = sizeof(args);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(f, grid, 1, 1, 64, 1, 1, 0, stream, nullptr, config);
}

inline hipError_t mpc_tile_refill_launch(int n, int T, const float* x0, const float* xref, const crx_mpc_params& q, float* sol,
                                         int* status, double* cost, hipStream_t stream, int chunk, int hold, int store = 1) {
  const MpcTileArgs args{n, T, chunk, hold, x0, xref, mpc_pack(q), sol, status, cost, nullptr, nullptr, -1, 0, nullptr};
  return mpc_tile_module_launch(store == 2 ? kTileRefill2 : kTileRefill1, (unsigned)(((size_t)n + chunk - 1) / chunk), args, stream);
}

// T - 1 <= kMpcTileStages.  One instantiation (MAXT = 24) for every horizon it takes: the <8> build of mpc_solve_lane (fully unrolled
// short loops) needs more accumulator registers of its own than the block leaves free.
inline hipError_t mpc_tile_launch(int n, int T, const float* x0, const float* xref, const crx_mpc_params& q, float* sol,
                                  int* status, double* cost, hipStream_t stream, int store = 1) {
  const MpcTileArgs args{n, T, 0, 0, x0, xref, mpc_pack(q), sol, status, cost, nullptr, nullptr, -1, 0, nullptr};
  return mpc_tile_module_launch(store == 2 ? kTileLockstep2 : (store == 3 ? kTileLite : kTileLockstep1), (unsigned)(((size_t)n + 63) / 64), args, stream);
}

inline hipError_t mpc_tile_phase_launch(int n, int T, const int* list, const int* count, int resume, int cap, double* state, const float* x0,
                                        const float* xref, const crx_mpc_params& q, float* sol, int* status, double* cost, hipStream_t stream) {
  const MpcTileArgs args{n, T, 0, 0, x0, xref, mpc_pack(q), sol, status, cost, list, count, resume, cap, state};
  return mpc_tile_module_launch(kTilePhase1, (unsigned)(((size_t)n + 63) / 64), args, stream);
}

#endif

}  // namespace crx

// api_mpc.inl — part of the single translation unit crx_api.hip (#included there, in this order: api_internal, api_core, api_ekf,
// api_lqr, api_mpc, api_track, api_planners, api_frenet, api_probes); mpc_solve (src/model_predictive_control.cpp:255-346).
extern "C" {

// ---------------------------------------------------------------------------------------------
// MPC
// ---------------------------------------------------------------------------------------------
// agents_per_wave (1..64) and waves_per_workgroup (1..4): the launch geometry; the product entry point uses full waves in
// single-wave workgroups (every emptier or stacked geometry measured slower: profiles/r02/mpc_tail.txt).
// Every hardware queue a kernel with private (scratch) memory has run on keeps a reservation sized for a full device of its waves, all
// out of one pool: the solver's 3.8-5 KB per lane is ~0.3 MB per wave, 12 queues holding one work, 16 end the PROCESS with
// HSA_STATUS_ERROR_OUT_OF_RESOURCES whatever the batch size (profiles/r05/scratch_queues_probe.jsonl) — an abort no caller of a C
// function expects.  The library therefore counts the distinct (device, stream) pairs the private-memory solver has been launched
// on and refuses the 13th with CRX_ERR_INVALID instead.  Streams are counted, not hardware queues (the runtime
// does not say which queue a stream lands on): conservative when several streams share a queue.  The tile kernels
// (mpc_tile_kernels.hip.h: 1.9 KB of private memory per lane instead of 3.8-5) pass the probe on 32 streams and are not counted.
static const int kMaxPrivateMemoryStreams = 12;
static int scratch_stream_admit(void* stream) {
  static std::mutex mu;
  static std::vector<std::pair<int, void*>> seen;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  std::lock_guard<std::mutex> l(mu);
  for (const auto& s : seen) if (s.first == dev && s.second == stream) return CRX_OK;
  int on_dev = 0;
  for (const auto& s : seen) on_dev += s.first == dev;
  if (on_dev >= kMaxPrivateMemoryStreams)
    return fail(CRX_ERR_INVALID, "mpc_solve: this would be the 13th distinct stream of this device to run the private-memory solver; every hardware "
                                 "queue it has run on keeps a full-device scratch reservation and 16 of them abort the process "
                                 "(HSA_STATUS_ERROR_OUT_OF_RESOURCES) — keep the solver on <= 12 streams (INTEGRATION.md 7)");
  seen.emplace_back(dev, stream);
  return CRX_OK;
}

static int mpc_solve_launch(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm,
                            float* sol, int* status, double* cost, void* stream, int agents_per_wave, int waves_per_workgroup, int trig = -1) {
  if (n < 0 || T < 2 || T > CRX_MPC_MAX_T || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve: bad argument (2 <= T <= 64)");
  if (agents_per_wave < 1 || agents_per_wave > 64 || waves_per_workgroup < 1 || waves_per_workgroup > 4)
    return fail(CRX_ERR_INVALID, "mpc_solve: launch geometry out of range (1..64 agents per wave, 1..4 waves per workgroup)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  if (int rc = scratch_stream_admit(stream)) return rc;
  const hipError_t e = crx::mpc_launch(n, T, x0, xref, p, sol, status, cost, (hipStream_t)stream, agents_per_wave, waves_per_workgroup, trig);
  return e == hipSuccess ? CRX_OK : hip_fail(e, "mpc launch");
}
// The lane-refilling launch (mpc_refill_kernel: a wave owns `agents_per_wave` consecutive agents; bit-identical to mpc_kernel per agent).
static int mpc_solve_refill(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                            double* cost, void* stream, int agents_per_wave, int hold_lanes) {
  if (n < 0 || T < 2 || T > CRX_MPC_MAX_T || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve (refill): bad argument (2 <= T <= 64)");
  if (agents_per_wave < 64 || agents_per_wave > (1 << 20) || hold_lanes < 1 || hold_lanes > 64)
    return fail(CRX_ERR_INVALID, "mpc_solve (refill): agents_per_wave 64 .. 2^20, hold_lanes 1 .. 64");
#if !CRX_EXPERIMENTAL_KERNELS
  return fail(CRX_ERR_INVALID, "mpc_solve (refill): this libcrx.so was built without the experimental kernels (CRX_EXPERIMENTAL_KERNELS=0)");
#else
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  if (p.max_iter < 1) return fail(CRX_ERR_INVALID, "mpc_solve (refill): max_iter must be at least 1");
  const hipError_t e = crx::mpc_refill_launch(n, T, x0, xref, p, sol, status, cost, (hipStream_t)stream, agents_per_wave, hold_lanes);
  return e == hipSuccess ? CRX_OK : hip_fail(e, "mpc refill launch");
#endif
}
// The tile layout (mpc_tile_kernels.hip.h): controls in LDS, feedback gains in accumulator registers; T <= 21.
static int mpc_solve_tile(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                          double* cost, void* stream, int store = 1) {
  if (n < 0 || T < 2 || T - 1 > crx::kMpcTileStages || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve (tile layout): bad argument (2 <= T <= 21)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  const hipError_t e = crx::mpc_tile_launch(n, T, x0, xref, p, sol, status, cost, (hipStream_t)stream, store);
  return e == hipSuccess ? CRX_OK : hip_fail(e, "mpc tile launch");
}
static int mpc_solve_tile_refill(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                 double* cost, void* stream, int agents_per_wave, int hold_lanes, int store = 1) {
  if (n < 0 || T < 2 || T - 1 > crx::kMpcTileStages || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve (tile layout, refilled lanes): bad argument (2 <= T <= 21)");
  if (agents_per_wave < 64 || agents_per_wave > (1 << 20) || hold_lanes < 1 || hold_lanes > 64)
    return fail(CRX_ERR_INVALID, "mpc_solve (tile layout, refilled lanes): agents_per_wave 64 .. 2^20, hold_lanes 1 .. 64");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  if (p.max_iter < 1) return fail(CRX_ERR_INVALID, "mpc_solve (tile layout, refilled lanes): max_iter must be at least 1");
  if (store != 1 && store != 2) return fail(CRX_ERR_INVALID, "mpc_solve (tile layout, refilled lanes): store must be 1 or 2");
  const hipError_t e = crx::mpc_tile_refill_launch(n, T, x0, xref, p, sol, status, cost, (hipStream_t)stream, agents_per_wave, hold_lanes, store);
  return e == hipSuccess ? CRX_OK : hip_fail(e, "mpc tile refill launch");
}
// lanes_per_agent: 1 = mpc_kernel (one agent per lane), 4 = mpc_quad_kernel (a DPP quad per agent, parallel line search; T <= 24;
// measured 0.95x at BASELINE configs[3] and less beyond, never selected), 0 = what the product entry point uses (= 1).
static int mpc_solve_lanes(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                           double* cost, void* stream, int lanes_per_agent) {
  if (lanes_per_agent != 0 && lanes_per_agent != 1 && lanes_per_agent != 4)
    return fail(CRX_ERR_INVALID, "mpc_solve: lanes_per_agent must be 0 (auto), 1 or 4");
  // the product's choice (round 6): one lane per agent, lockstep sweeps, everywhere; from kMpcTileFrom agents on — the throughput regime,
  // where the private-memory kernel is bound by the HBM traffic of its own scratch — the TILE layout (controls in LDS, feedback gains in
  // accumulator registers: 82 -> 43 KB of HBM traffic per solve): 131,072 agents 3.72 -> 3.41 ms, 262,144 6.1-6.3 -> 5.7, 524,288 9.9 ->
  // 8.6, 1 M 17.3-17.7 -> 14.7 = 71 M solves/s (profiles/r06/mpc_store_ab*.jsonl; bit-identical per agent).  Not below: a launch that leaves
  // SIMDs idle is a latency chain, where the tile kernel's register switch costs 15 % (8,192 agents 1.18 -> 1.33 ms).  Not for a caller
  // whose launches share the GPU (crx_mpc_params.shared_gpu — configs[4]): a tile wave owns all 512 registers of its SIMD, the
  // private-memory wave (256 + ~31) leaves room for an EKF wave beside it, and the round is 0.50 ms with it against 0.85
  // (profiles/r06/swarm_store_ab.jsonl).  The tile layout with REFILLED lanes (crx_x_mpc_solve_tile_refill_dev) is 1.57x the
  // private-memory kernel at 1 M agents on a distribution without stragglers and 1.25x / 0.98x / 1.24x at 1 M / 524 k / 262 k on the
  // configs[3] distribution, whose few agents at the 50-sweep cap — each a 2 ms chain started whenever its wave reaches it — set the
  // launch's tail; the PHASED solve on the tile layout (crx_x_mpc_solve_phased_store_dev) 13.7 ms at 1 M against 14.7: both measured,
  // kept as entry points, not selected (the two compacting schedulers within 1 % of each other, 7 % ahead).  (The quad variant lost its A/B at every batch size,
  // profiles/r03/mpc_lanes_ab.txt.)
  // The LITE tile layout (store 3: controls in LDS, the gains of stages 1 .. 7 in a40 .. a123 — a 384-register wave, which leaves a
  // 128-register wave of another kernel room on the SIMD): between the two — 49,152 agents 1.70 -> 1.55 ms, 65,536 2.60 -> 2.45, 98,304
  // 3.06 -> 2.98 against the private-memory kernel, level with the full tile layout at 131,072 - 262,144 and 1-3 % behind it beyond
  // (profiles/r06/mpc_store_ab_lite*.jsonl) — and the one tile form a caller whose launches share the GPU can use: configs[4]'s round
  // 0.475 -> 0.456 ms at depth 6, 0.447 -> 0.439 at depth 7 (profiles/r06/swarm_store_ab_lite_depth*.jsonl).
  if (lanes_per_agent == 0) {
    const bool shared = prm && prm->shared_gpu != 0;
    if (T - 1 <= crx::kMpcTileStages) {
      if (n >= crx::kMpcTileFrom && !shared) return mpc_solve_tile(n, T, x0, xref, prm, sol, status, cost, stream);
      if (n >= (shared ? crx::kMpcTileLiteFromShared : crx::kMpcTileLiteFrom)) return mpc_solve_tile(n, T, x0, xref, prm, sol, status, cost, stream, 3);
    }
    lanes_per_agent = 1;
  }
  if (lanes_per_agent == 1) return mpc_solve_launch(n, T, x0, xref, prm, sol, status, cost, stream, 64, 1);
#if !CRX_EXPERIMENTAL_KERNELS
  return fail(CRX_ERR_INVALID, "mpc_solve (four lanes per agent): this libcrx.so was built without the experimental kernels");
#else
  if (n < 0 || T < 2 || T > 24 || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve (four lanes per agent): bad argument (2 <= T <= 24)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  const hipError_t e = crx::mpc_quad_launch(n, T, x0, xref, p, sol, status, cost, (hipStream_t)stream);
  return e == hipSuccess ? CRX_OK : hip_fail(e, "mpc launch");
#endif
}
int crx_x_mpc_solve_tile_refill_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                    double* cost, void* stream, int agents_per_wave, int hold_lanes) {
  CRX_TRACE();
  return mpc_solve_tile_refill(n, T, x0, xref, prm, sol, status, cost, stream, agents_per_wave, hold_lanes);
}
int crx_x_mpc_solve_store_refill_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                     double* cost, void* stream, int store, int agents_per_wave, int hold_lanes) {
  CRX_TRACE();
  return mpc_solve_tile_refill(n, T, x0, xref, prm, sol, status, cost, stream, agents_per_wave, hold_lanes, store);
}
// The two-phase solve (mpc_kernels.hip.h; measured and not selected, include/crx_experimental.h): phase 1 = the product launch with the sweep cap lowered to first_sweeps, on `stream`; the
// agents that ran into the cap are listed and solved from scratch with the caller's cap on `tail_stream` (behind an event; may be the
// same stream).  Bit for bit crx_mpc_solve_batch_dev's answers.  work: (n + 64) ints of device memory the call owns until both
// streams have passed it.  Events: a small per-process ring (created once; an event is reused only long after its wait was enqueued).
static hipEvent_t two_phase_event() {
  static std::mutex mu;
  static std::vector<hipEvent_t> ring;
  static size_t next = 0;
  std::lock_guard<std::mutex> l(mu);
  if (ring.empty()) {
    ring.resize(256, nullptr);
  }
  hipEvent_t& e = ring[next++ % ring.size()];
  if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return e;
}
int crx_x_mpc_solve_two_phase_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                double* cost, int first_sweeps, int* work, void* stream, void* tail_stream) {
  CRX_TRACE();
  if (n < 0 || T < 2 || T > CRX_MPC_MAX_T || (n && (!x0 || !xref || !sol || !status || !work)))
    return fail(CRX_ERR_INVALID, "mpc_solve (two phases): bad argument (2 <= T <= 64; status and work are required)");
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  if (first_sweeps < 1 || first_sweeps >= p.max_iter)          // nothing to split: the ordinary launch
    return crx_mpc_solve_batch_dev(n, T, x0, xref, &p, sol, status, cost, stream);
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p1 = p;
  p1.max_iter = first_sweeps;
  int* count = work;            // work[0]: the count; work[64 ..]: the list (kept apart from the counter's cache line)
  int* list = work + 64;
  CRX_HIP(hipMemsetAsync(count, 0, sizeof(int), (hipStream_t)stream));
  CRX_TRY(crx_mpc_solve_batch_dev(n, T, x0, xref, &p1, sol, status, cost, stream));
  hipLaunchKernelGGL(crx::mpc_collect_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, n, first_sweeps, status, list, count);
  CRX_HIP(hipGetLastError());
  hipStream_t tail = (hipStream_t)tail_stream;
  if (tail != (hipStream_t)stream) {
    hipEvent_t ev = two_phase_event();
    if (!ev) return fail(CRX_ERR_HIP, "mpc_solve (two phases): no event");
    CRX_HIP(hipEventRecord(ev, (hipStream_t)stream));
    CRX_HIP(hipStreamWaitEvent(tail, ev, 0));
  }
  if (int rc = scratch_stream_admit(tail_stream)) return rc;
  const hipError_t e = crx::mpc_list_launch(n, T, list, count, x0, xref, p, sol, status, cost, tail);
  return e == hipSuccess ? CRX_OK : hip_fail(e, "mpc list launch");
}
// The phased solve (mpc_kernels.hip.h: mpc_phase_kernel): caps[0] < caps[1] < ... sweep indices at which the agents still unconverged
// are suspended, compacted into full waves and resumed.  work: crx_x_mpc_phased_work_bytes(n, T) bytes of device memory (the counter,
// the list, one state record per agent), owned by the call until the stream has passed it.  T <= 24.
size_t crx_x_mpc_phased_work_bytes(int n, int T) {
  if (n < 0 || T < 2) return 0;
  return 256 + (((size_t)n * 4 + 255) / 256) * 256 + (size_t)n * 8 * (size_t)crx::mpc_phase_record_doubles(T);
}
static int mpc_solve_phased(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                            double* cost, const int* caps, int ncaps, void* work, void* stream, int store) {
  if (n < 0 || T < 2 || T > 24 || ncaps < 0 || ncaps > 16 || (ncaps && !caps) || (n && (!x0 || !xref || !sol || !status || !work)))
    return fail(CRX_ERR_INVALID, "mpc_solve (phased): bad argument (2 <= T <= 24; status and work are required; at most 16 caps)");
  for (int k = 0; k < ncaps; ++k)
    if (caps[k] < 1 || (k && caps[k] <= caps[k - 1])) return fail(CRX_ERR_INVALID, "mpc_solve (phased): caps must be positive and increasing");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  if (store != 0 && store != 1) return fail(CRX_ERR_INVALID, "mpc_solve (phased): store must be 0 (private memory) or 1 (tile layout)");
  if (store == 1 && T - 1 > crx::kMpcTileStages) return fail(CRX_ERR_INVALID, "mpc_solve (phased, tile layout): T <= 21");
  if (store == 0) if (int rc = scratch_stream_admit(stream)) return rc;
  hipStream_t s = (hipStream_t)stream;
  int* count = (int*)work;
  int* list = (int*)((char*)work + 256);
  double* state = (double*)((char*)work + 256 + (((size_t)n * 4 + 255) / 256) * 256);
  const bool lean = n >= crx::kMpcLeanFrom || (p.shared_gpu != 0 && n >= crx::kMpcLeanFromShared);
  int resume = -1;
  for (int k = 0; k <= ncaps; ++k) {
    const int cap = (k < ncaps && caps[k] < p.max_iter) ? caps[k] : 0;      // the last phase runs to the solver's own cap
    if (k > 0) {
      CRX_HIP(hipMemsetAsync(count, 0, sizeof(int), s));
      hipLaunchKernelGGL(crx::mpc_collect_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, -1, status, list, count);
      CRX_HIP(hipGetLastError());
    }
    const hipError_t e = store == 1 ? crx::mpc_tile_phase_launch(n, T, k ? list : nullptr, count, resume, cap, state, x0, xref, p, sol, status, cost, s)
                                    : crx::mpc_phase_launch(n, T, k ? list : nullptr, count, resume, cap, state, x0, xref, p, sol, status, cost, s, lean, n);
    if (e != hipSuccess) return hip_fail(e, "mpc phase launch");
    if (cap == 0) break;
    resume = cap;
  }
  return CRX_OK;
}
int crx_x_mpc_solve_phased_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                               double* cost, const int* caps, int ncaps, void* work, void* stream) {
  CRX_TRACE();
  return mpc_solve_phased(n, T, x0, xref, prm, sol, status, cost, caps, ncaps, work, stream, 0);
}
int crx_x_mpc_solve_phased_store_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                     double* cost, const int* caps, int ncaps, void* work, void* stream, int store) {
  CRX_TRACE();
  return mpc_solve_phased(n, T, x0, xref, prm, sol, status, cost, caps, ncaps, work, stream, store);
}
int crx_mpc_solve_batch_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm,
                            float* sol, int* status, double* cost, void* stream) {
  CRX_TRACE();
  return mpc_solve_lanes(n, T, x0, xref, prm, sol, status, cost, stream, 0);
}
// store: 0 = private memory (mpc_kernel), 1 = the tile layout (mpc_tile_kernel).  Bit-identical answers.
int crx_x_mpc_solve_store_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                              double* cost, void* stream, int store) {
  CRX_TRACE();
  if (store >= 1 && store <= 3) return mpc_solve_tile(n, T, x0, xref, prm, sol, status, cost, stream, store);
  if (store != 0) return fail(CRX_ERR_INVALID, "mpc_solve (store): store must be 0 (private memory), 1 (tile layout), 2 (checkpointed tile layout) or 3 (lite tile layout)");
  return mpc_solve_launch(n, T, x0, xref, prm, sol, status, cost, stream, 64, 1);
}
// mpc_solve for n agents with the four-variant portfolio (mpc_kernels.hip.h: mpc_variant): the same NLP, every agent answered by the
// variant of the solver that converges in the fewest sweeps.
int crx_mpc_solve_portfolio_batch_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm,
                                      float* sol, int* status, double* cost, void* stream) {
  CRX_TRACE();
  if (n < 0 || T < 2 || T > CRX_MPC_MAX_T || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve (portfolio): bad argument (2 <= T <= 64)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  const hipError_t e = crx::mpc_portfolio_launch(n, T, x0, xref, p, sol, status, cost, (hipStream_t)stream);
  return e == hipSuccess ? CRX_OK : hip_fail(e, "mpc portfolio launch");
}
int crx_x_mpc_solve_lanes_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                              double* cost, void* stream, int lanes_per_agent) {
  CRX_TRACE();
  return mpc_solve_lanes(n, T, x0, xref, prm, sol, status, cost, stream, lanes_per_agent);
}
int crx_x_mpc_solve_refill_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                               double* cost, void* stream, int agents_per_wave, int hold_lanes) {
  CRX_TRACE();
  return mpc_solve_refill(n, T, x0, xref, prm, sol, status, cost, stream, agents_per_wave, hold_lanes);
}
int crx_x_mpc_solve_trig_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                             double* cost, void* stream, int recompute_trig) {
  CRX_TRACE();
  if (recompute_trig != 0 && recompute_trig != 1) return fail(CRX_ERR_INVALID, "mpc_solve (trig): recompute_trig must be 0 or 1");
  return mpc_solve_launch(n, T, x0, xref, prm, sol, status, cost, stream, 64, 1, recompute_trig);
}
int crx_x_mpc_solve_geometry_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                 double* cost, void* stream, int agents_per_wave, int waves_per_workgroup) {
  CRX_TRACE();
  return mpc_solve_launch(n, T, x0, xref, prm, sol, status, cost, stream, agents_per_wave, waves_per_workgroup);
}

static int mpc_solve_host(bool portfolio, int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol,
                          int* status, double* cost) {
  if (n < 0 || T < 2 || T > CRX_MPC_MAX_T || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve: bad argument (2 <= T <= 64)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
    HostCall hc;
    CRX_TRY(hc.open());
    const int ix = hc.add(x0 + 4 * a0, nullptr, 16 * nl), ir = hc.add(xref + 4 * (size_t)T * a0, nullptr, 16 * (size_t)T * nl);
    const int is = hc.add(nullptr, sol + nv * a0, 4 * nv * nl);
    const int it = hc.add(nullptr, status ? status + a0 : nullptr, 4 * nl), ic = hc.add(nullptr, cost ? cost + a0 : nullptr, 8 * nl);
    CRX_TRY(hc.commit());
    if (portfolio)
      CRX_TRY(crx_mpc_solve_portfolio_batch_dev((int)nl, T, hc.p<float>(ix), hc.p<float>(ir), prm, hc.p<float>(is), hc.p<int>(it),
                                                hc.p<double>(ic), hc.stream()));
    else
      CRX_TRY(crx_mpc_solve_batch_dev((int)nl, T, hc.p<float>(ix), hc.p<float>(ir), prm, hc.p<float>(is), hc.p<int>(it), hc.p<double>(ic),
                                      hc.stream()));
    return hc.finish();
  });
}
int crx_mpc_solve_batch(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol,
                        int* status, double* cost) {
  CRX_TRACE();
  return mpc_solve_host(false, n, T, x0, xref, prm, sol, status, cost);
}
int crx_mpc_solve_portfolio_batch(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol,
                                  int* status, double* cost) {
  CRX_TRACE();
  return mpc_solve_host(true, n, T, x0, xref, prm, sol, status, cost);
}

}  // extern "C"

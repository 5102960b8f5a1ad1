// api_lqr.inl — part of the single translation unit crx_api.hip (#included there, in this order: api_internal, api_core, api_ekf,
// api_lqr, api_mpc, api_track, api_planners, api_frenet, api_probes); solve_DARE / dlqr (src/lqr_speed_steer_control.cpp:85-106, src/lqr_steer_control.cpp:75-96): kernel choice by batch size and
// by the matrices' pattern.
extern "C" {

// ---------------------------------------------------------------------------------------------
// DARE / dlqr
// ---------------------------------------------------------------------------------------------
// structured: 1 = detect the pattern lqr_steering_control builds (per agent) and serve those agents by the structured kernels, the
// rest by a dense kernel (two launches, no workspace, no synchronisation); 0 = a dense kernel for everybody.
// dense_lanes: 1 = dare_dense_kernel (one agent per lane), 4 = dare_dense_quad_kernel (one row of X per lane of a quad), 0 = by batch size.
static int dare_batch_launch(int n, int dim, const float* A, const float* B, const float* Q, const float* R,
                             float eps, int maxiter, float* X, float* K, int* iters, void* stream, int structured, int dense_lanes) {
  if (n < 0 || (dim != 4 && dim != 5) || (n && (!A || !B || !Q || !R)))
    return fail(CRX_ERR_INVALID, "dare: bad argument (dim must be 4 or 5)");
  if (dense_lanes != 0 && dense_lanes != 1 && dense_lanes != 4) return fail(CRX_ERR_INVALID, "dare: lanes_per_agent must be 0 (auto), 1 or 4");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipStream_t s = (hipStream_t)stream;
  const unsigned bs = iter_block();
  const dim3 grid(blocks_for(n, bs)), block(bs);
  const dim3 qgrid(blocks_for(4 * (size_t)n, 256)), qblock(256);
  // a quad per agent while the batch leaves SIMDs without a wave of their own (the structured kernels' crossover)
  if (dense_lanes == 0) dense_lanes = (n <= kDareDenseQuadMaxAgents) ? 4 : 1;
  if (structured) {
    const crx::DareFromMats src{A, B, Q, R};
    if (n <= kDareQuadMaxAgents) {
      if (dim == 5) hipLaunchKernelGGL((crx::dare_from_v_quad_kernel<5, crx::DareFromMats>), qgrid, qblock, 0, s, n, src, eps, maxiter, X, K, iters);
      else hipLaunchKernelGGL((crx::dare_from_v_quad_kernel<4, crx::DareFromMats>), qgrid, qblock, 0, s, n, src, eps, maxiter, X, K, iters);
    } else if (n <= kDareChainMaxAgents) {
      if (dim == 5) hipLaunchKernelGGL((crx::dare_from_v_kernel<5, crx::DareFromMats>), grid, block, 0, s, n, src, eps, maxiter, X, K, iters);
      else hipLaunchKernelGGL((crx::dare_from_v_kernel<4, crx::DareFromMats>), grid, block, 0, s, n, src, eps, maxiter, X, K, iters);
    } else if (maxiter > 0 && dare_refill_chunk(n)) {
      const int chunk = dare_refill_chunk(n);
      const dim3 rgrid(blocks_for(n, chunk));
      if (dim == 5) hipLaunchKernelGGL((crx::dare_from_v_refill_kernel<5, crx::DareFromMats>), rgrid, block, 0, s, n, chunk, kDareRefillHold, src, eps, maxiter, X, K, iters);
      else hipLaunchKernelGGL((crx::dare_from_v_refill_kernel<4, crx::DareFromMats>), rgrid, block, 0, s, n, chunk, kDareRefillHold, src, eps, maxiter, X, K, iters);
    } else {
      if (dim == 5) hipLaunchKernelGGL((crx::dare_from_v_masked_kernel<5, crx::DareFromMats>), grid, block, 0, s, n, src, eps, maxiter, X, K, iters);
      else hipLaunchKernelGGL((crx::dare_from_v_masked_kernel<4, crx::DareFromMats>), grid, block, 0, s, n, src, eps, maxiter, X, K, iters);
    }
    CRX_HIP(hipGetLastError());
  }
#define CRX_LAUNCH_DENSE(DIM, SKIP)                                                                                                     \
  do {                                                                                                                                  \
    if (dense_lanes == 4) hipLaunchKernelGGL((crx::dare_dense_quad_kernel<DIM, SKIP>), qgrid, qblock, 0, s, n, A, B, Q, R, eps, maxiter, X, K, iters); \
    else hipLaunchKernelGGL((crx::dare_dense_kernel<DIM, SKIP>), grid, block, 0, s, n, A, B, Q, R, eps, maxiter, X, K, iters);          \
  } while (0)
  if (structured) { if (dim == 5) CRX_LAUNCH_DENSE(5, true); else CRX_LAUNCH_DENSE(4, true); }
  else { if (dim == 5) CRX_LAUNCH_DENSE(5, false); else CRX_LAUNCH_DENSE(4, false); }
#undef CRX_LAUNCH_DENSE
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_dare_batch_dev(int n, int dim, const float* A, const float* B, const float* Q, const float* R,
                       float eps, int maxiter, float* X, float* K, int* iters, void* stream) {
  CRX_TRACE();
  return dare_batch_launch(n, dim, A, B, Q, R, eps, maxiter, X, K, iters, stream, 1, 0);
}
int crx_x_dare_batch_dense_dev(int n, int dim, const float* A, const float* B, const float* Q, const float* R,
                               float eps, int maxiter, float* X, float* K, int* iters, void* stream, int lanes_per_agent) {
  CRX_TRACE();
  return dare_batch_launch(n, dim, A, B, Q, R, eps, maxiter, X, K, iters, stream, 0, lanes_per_agent);
}

// lanes_per_agent: 1 = dare_from_v_kernel, 4 = dare_from_v_quad_kernel, 0 = chosen by batch size.
// refill_chunk: agents per wave of the lane-refilling kernel; 0 = the product's choice (dare_refill_chunk: above 262,144 agents,
// no refilling below), -1 = never (the masked kernel: rounds 2-3's throughput-regime kernel, kept for the A/B)
static int dare_from_v_launch(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                              int* iters, void* stream, int lanes_per_agent, int refill_chunk = 0, int refill_hold = kDareRefillHold) {
  if (n < 0 || (dim != 4 && dim != 5) || (n && !v))
    return fail(CRX_ERR_INVALID, "dare_from_v: bad argument (dim must be 4 or 5)");
  if (lanes_per_agent != 0 && lanes_per_agent != 1 && lanes_per_agent != 4)
    return fail(CRX_ERR_INVALID, "dare_from_v: lanes_per_agent must be 0 (auto), 1 or 4");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_lqr_params p;
  if (prm) p = *prm; else crx_lqr_default_params(&p);
  // Four lanes per agent shorten the launch while the batch leaves SIMDs without a wave of their own; in the throughput
  // regime the one-lane kernel executes fewer instructions per agent (profiles/r03/dare_lanes_ab.txt).
  if (lanes_per_agent == 0) lanes_per_agent = (n <= kDareQuadMaxAgents) ? 4 : 1;
  const crx::DareFromV src{v, (float)p.dt, p.L};
  if (lanes_per_agent == 4) {
    const dim3 grid(blocks_for(4 * (size_t)n, 256)), block(256);
    if (dim == 5)
      hipLaunchKernelGGL((crx::dare_from_v_quad_kernel<5, crx::DareFromV>), grid, block, 0, (hipStream_t)stream, n, src, p.eps, p.maxiter, X, K, iters);
    else
      hipLaunchKernelGGL((crx::dare_from_v_quad_kernel<4, crx::DareFromV>), grid, block, 0, (hipStream_t)stream, n, src, p.eps, p.maxiter, X, K, iters);
  } else {
    const dim3 grid(blocks_for(n, 64)), block(64);
    // up to ~1.5 waves per SIMD the launch is a latency chain: nobody masked off, two evaluations per branch (89 VGPRs); beyond,
    // the masked loop at eight waves per SIMD (60 VGPRs)
    const bool chain = n <= kDareChainMaxAgents;
    if (refill_chunk == 0) refill_chunk = dare_refill_chunk(n) ? dare_refill_chunk(n) : -1;
    if (refill_chunk > 0 && p.maxiter > 0) {
      const dim3 rgrid(blocks_for(n, refill_chunk));
      if (dim == 5) hipLaunchKernelGGL((crx::dare_from_v_refill_kernel<5, crx::DareFromV>), rgrid, block, 0, (hipStream_t)stream, n, refill_chunk, refill_hold, src, p.eps, p.maxiter, X, K, iters);
      else hipLaunchKernelGGL((crx::dare_from_v_refill_kernel<4, crx::DareFromV>), rgrid, block, 0, (hipStream_t)stream, n, refill_chunk, refill_hold, src, p.eps, p.maxiter, X, K, iters);
      CRX_HIP(hipGetLastError());
      return CRX_OK;
    }
#define CRX_LAUNCH_DV(KERNEL, DIM) \
    hipLaunchKernelGGL((crx::KERNEL<DIM, crx::DareFromV>), grid, block, 0, (hipStream_t)stream, n, src, p.eps, p.maxiter, X, K, iters)
    if (dim == 5) { if (chain) CRX_LAUNCH_DV(dare_from_v_kernel, 5); else CRX_LAUNCH_DV(dare_from_v_masked_kernel, 5); }
    else { if (chain) CRX_LAUNCH_DV(dare_from_v_kernel, 4); else CRX_LAUNCH_DV(dare_from_v_masked_kernel, 4); }
#undef CRX_LAUNCH_DV
  }
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_dare_from_v_batch_dev(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                              int* iters, void* stream) {
  CRX_TRACE();
  return dare_from_v_launch(n, dim, v, prm, X, K, iters, stream, 0);
}

int crx_x_dare_from_v_lanes_dev(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                                int* iters, void* stream, int lanes_per_agent) {
  CRX_TRACE();
  return dare_from_v_launch(n, dim, v, prm, X, K, iters, stream, lanes_per_agent);
}

int crx_x_dare_from_v_refill_dev(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                                 int* iters, void* stream, int agents_per_wave, int hold_lanes) {
  CRX_TRACE();
  if (agents_per_wave != -1 && (agents_per_wave < 64 || agents_per_wave > (1 << 20)))
    return fail(CRX_ERR_INVALID, "dare_from_v_refill: agents_per_wave must be -1 (the masked kernel) or 64 .. 2^20");
  if (hold_lanes < 1 || hold_lanes > 64) return fail(CRX_ERR_INVALID, "dare_from_v_refill: hold_lanes must be 1 .. 64");
  return dare_from_v_launch(n, dim, v, prm, X, K, iters, stream, 1, agents_per_wave, hold_lanes);
}

int crx_dare_batch(int n, int dim, const float* A, const float* B, const float* Q, const float* R, float eps,
                   int maxiter, float* X, float* K, int* iters) {
  CRX_TRACE();
  if (n < 0 || (dim != 4 && dim != 5) || (n && (!A || !B || !Q || !R)))
    return fail(CRX_ERR_INVALID, "dare: bad argument (dim must be 4 or 5)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, m = (dim == 5) ? 2 : 1, d2 = (size_t)dim * dim, db = (size_t)dim * m;
    HostCall hc;
    CRX_TRY(hc.open());
    const int iA = hc.add(A + d2 * a0, nullptr, 4 * d2 * nl), iB = hc.add(B + db * a0, nullptr, 4 * db * nl);
    const int iQ = hc.add(Q + d2 * a0, nullptr, 4 * d2 * nl), iR = hc.add(R + m * m * a0, nullptr, 4 * m * m * nl);
    const int iX = X ? hc.add(nullptr, X + d2 * a0, 4 * d2 * nl) : -1, iK = K ? hc.add(nullptr, K + db * a0, 4 * db * nl) : -1;
    const int iI = iters ? hc.add(nullptr, iters + a0, 4 * nl) : -1;
    CRX_TRY(hc.commit());
    CRX_TRY(crx_dare_batch_dev((int)nl, dim, hc.p<float>(iA), hc.p<float>(iB), hc.p<float>(iQ), hc.p<float>(iR), eps, maxiter,
                               iX >= 0 ? hc.p<float>(iX) : nullptr, iK >= 0 ? hc.p<float>(iK) : nullptr, iI >= 0 ? hc.p<int>(iI) : nullptr,
                               hc.stream()));
    return hc.finish();
  });
}

int crx_dare_from_v_batch(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K, int* iters) {
  CRX_TRACE();
  if (n < 0 || (dim != 4 && dim != 5) || (n && !v))
    return fail(CRX_ERR_INVALID, "dare_from_v: bad argument (dim must be 4 or 5)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, m = (dim == 5) ? 2 : 1, d2 = (size_t)dim * dim, db = (size_t)dim * m;
    HostCall hc;
    CRX_TRY(hc.open());
    const int iv = hc.add(v + a0, nullptr, 4 * nl);
    const int iX = X ? hc.add(nullptr, X + d2 * a0, 4 * d2 * nl) : -1, iK = K ? hc.add(nullptr, K + db * a0, 4 * db * nl) : -1;
    const int iI = iters ? hc.add(nullptr, iters + a0, 4 * nl) : -1;
    CRX_TRY(hc.commit());
    CRX_TRY(crx_dare_from_v_batch_dev((int)nl, dim, hc.p<float>(iv), prm, iX >= 0 ? hc.p<float>(iX) : nullptr,
                                      iK >= 0 ? hc.p<float>(iK) : nullptr, iI >= 0 ? hc.p<int>(iI) : nullptr, hc.stream()));
    return hc.finish();
  });
}
}  // extern "C"

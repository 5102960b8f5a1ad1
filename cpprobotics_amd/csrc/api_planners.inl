// api_planners.inl — part of the single translation unit crx_api.hip (#included there, in this order: api_internal, api_core, api_ekf,
// api_lqr, api_mpc, api_track, api_planners, api_frenet, api_probes); particle filter and dynamic-window planner.
// ---------------------------------------------------------------------------------------------
// particle filter
// ---------------------------------------------------------------------------------------------
extern "C" {

void crx_pf_default_params(crx_pf_params* p) {
  if (!p) return;
  p->rsim0 = 1.0 * 1.0;
  p->rsim1 = (float)(30.0 / 180.0 * 3.141592653 * 30.0 / 180.0 * 3.141592653);
  p->Q = 0.01f;
  p->dt = 0.1;
  p->nth = 0.0f;
}

int crx_pf_run_batch_dev(int n, int np, int T, int L, float* px, float* pw, float* xEst, float* PEst, const float* obs,
                         const int* nobs, const float* u, const float* nrm, const float* uni, const crx_pf_params* prm,
                         float* x_hist, int* n_resampled, void* stream) {
  CRX_TRACE();
  if (n < 0 || T < 0 || L < 0 || (np != 100 && np != 64 && np != 128) ||
      (n && (!px || !pw || !xEst || !PEst)) || (n && T && (!nobs || !u || !nrm || !uni || (L && !obs))))
    return fail(CRX_ERR_INVALID, "pf_run: bad argument (np must be 64, 100 or 128)");
  if (int rc = check_device()) return rc;
  if (n == 0 || T == 0) return CRX_OK;                  // no tick: px, pw, xEst, PEst stay as they are
  crx_pf_params q;
  if (prm) q = *prm; else crx_pf_default_params(&q);
  const crx::PfParams p{q.rsim0, q.rsim1, q.Q, q.dt, q.nth > 0.0f ? q.nth : (float)(np / 2)};
  const dim3 grid(blocks_for(n, crx::kPfWavesPerBlock)), block(64 * crx::kPfWavesPerBlock);
  hipStream_t s = (hipStream_t)stream;
  if (np == 100) hipLaunchKernelGGL((crx::pf_run_kernel<100>), grid, block, 0, s, n, T, L, px, pw, xEst, PEst, obs, nobs, u, nrm, uni, p, x_hist, n_resampled);
  else if (np == 64) hipLaunchKernelGGL((crx::pf_run_kernel<64>), grid, block, 0, s, n, T, L, px, pw, xEst, PEst, obs, nobs, u, nrm, uni, p, x_hist, n_resampled);
  else hipLaunchKernelGGL((crx::pf_run_kernel<128>), grid, block, 0, s, n, T, L, px, pw, xEst, PEst, obs, nobs, u, nrm, uni, p, x_hist, n_resampled);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// dynamic-window planner
// ---------------------------------------------------------------------------------------------
extern "C" {

void crx_dwa_default_config(crx_dwa_config* c) {
  if (!c) return;
  const double PI_ = 3.141592653;   // `#define PI 3.141592653` (src/dynamic_window_approach.cpp:16)
  c->max_speed = 1.0; c->min_speed = -0.5; c->max_yawrate = 40.0 * PI_ / 180.0; c->max_accel = 0.2; c->robot_radius = 1.0;
  c->max_dyawrate = 40.0 * PI_ / 180.0; c->v_reso = 0.01; c->yawrate_reso = 0.1 * PI_ / 180.0; c->dt = 0.1; c->predict_time = 3.0;
  c->to_goal_cost_gain = 1.0; c->speed_cost_gain = 1.0;
}

int crx_dwa_run_batch_dev(int n, int max_ticks, float* state, float* u, const float* goal, const float* ob, int nob,
                          const crx_dwa_config* cfg, float* traj_hist, int* ticks_done, int* status, int* best_idx,
                          int* n_samples, void* stream) {
  CRX_TRACE();
  if (n < 0 || max_ticks < 0 || nob < 0 || nob > crx::kDwaMaxOb || (nob && !ob) || (n && (!state || !u || !goal)))
    return fail(CRX_ERR_INVALID, "dwa_run: bad argument (nob <= 256)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_dwa_config q;
  if (cfg) q = *cfg; else crx_dwa_default_config(&q);
  if (!(q.v_reso > 0.0f) || !(q.yawrate_reso > 0.0f) || !(q.dt > 0.0f)) return fail(CRX_ERR_INVALID, "dwa_run: resolutions and dt must be positive");
  crx::DwaCfg c;
  static_assert(sizeof(c) == sizeof(q), "config layouts must agree");
  std::memcpy(&c, &q, sizeof(c));
  hipLaunchKernelGGL(crx::dwa_run_kernel, dim3(blocks_for(n, crx::kDwaWavesPerBlock)), dim3(64 * crx::kDwaWavesPerBlock), 0,
                     (hipStream_t)stream, n, max_ticks, state, u, goal, ob, nob, c, traj_hist, ticks_done, status, best_idx, n_samples);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

}  // extern "C"

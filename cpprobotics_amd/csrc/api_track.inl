// api_track.inl — part of the single translation unit crx_api.hip (#included there, in this order: api_internal, api_core, api_ekf,
// api_lqr, api_mpc, api_track, api_planners, api_frenet, api_probes); the tracking front-end, the vehicle update and both closed loops.
// ---------------------------------------------------------------------------------------------
// course tracking front-end, vehicle update, closed loops
// ---------------------------------------------------------------------------------------------
namespace {

bool course_ok(const crx_course* c, bool need_ck_sp) {
  return c && c->n > 0 && c->cx && c->cy && c->cyaw && (!need_ck_sp || (c->ck && c->sp));
}
crx::CourseView view(const crx_course* c) { return crx::CourseView{c->cx, c->cy, c->cyaw, c->ck, c->sp, c->n}; }
crx::VehicleParams vparams(const crx_vehicle_params* p, int mpc) {
  crx_vehicle_params d;
  if (p) d = *p; else crx_vehicle_default_params(&d, mpc);
  return crx::VehicleParams{d.dt, d.wheelbase, d.max_steer, d.max_speed, d.min_speed, d.clamp_speed};
}
inline bool use_lds(const crx_course* c) { return c->n <= crx::kCourseLdsMax; }
inline size_t lds_bytes(const crx_course* c) { return use_lds(c) ? sizeof(float4) * (((size_t)c->n + 1) / 2) : 0; }   // two points per word
// the four-lanes-per-agent tracking kernels keep a gain slot per agent in static LDS next to the staged course (64 KB per workgroup in all)
inline bool use_quad(const crx_course* c, int n) { return n <= kDareQuadMaxAgents && use_lds(c) && lds_bytes(c) + 1024 <= 64 * 1024; }

// the course of a host-pointer call: its five arrays travel with the call's other arguments (replicated per shard)
struct CallCourse {
  int idx[5], n;
  void add(HostCall& hc, const crx_course* h) {
    const float* src[5] = {h->cx, h->cy, h->cyaw, h->ck, h->sp};
    n = h->n;
    for (int i = 0; i < 5; ++i) idx[i] = src[i] ? hc.add(src[i], nullptr, sizeof(float) * (size_t)h->n) : -1;
  }
  crx_course dev(HostCall& hc) const {
    const float* d[5];
    for (int i = 0; i < 5; ++i) d[i] = idx[i] >= 0 ? hc.p<float>(idx[i]) : nullptr;
    return crx_course{n, d[0], d[1], d[2], d[3], d[4]};
  }
};

}  // namespace

extern "C" {

void crx_vehicle_default_params(crx_vehicle_params* p, int mpc) {
  if (!p) return;
  p->dt = mpc ? 0.2 : 0.1;
  p->wheelbase = mpc ? 2.5 : 0.5;
  p->max_steer = 45.0 / 180 * 3.14159265358979323846;
  p->clamp_speed = mpc ? 1 : 0;
  p->max_speed = 55.0 / 3.6;
  p->min_speed = -20.0 / 3.6;
}

int crx_calc_nearest_index_batch_dev(int n, const float* state, const crx_course* course, int* ind, float* e, void* stream) {
  CRX_TRACE();
  if (n < 0 || !course_ok(course, false) || (n && (!state || !ind))) return fail(CRX_ERR_INVALID, "calc_nearest_index: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  const dim3 grid(blocks_for(n, crx::kTrackBlock)), block(crx::kTrackBlock);
  if (use_lds(course))
    hipLaunchKernelGGL((crx::calc_nearest_index_kernel<true>), grid, block, lds_bytes(course), (hipStream_t)stream, n, state, view(course), ind, e);
  else
    hipLaunchKernelGGL((crx::calc_nearest_index_kernel<false>), grid, block, 0, (hipStream_t)stream, n, state, view(course), ind, e);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_lqr_steering_control_batch_dev(int n, int dim, const float* state, const crx_course* course, int* ind, float* pe,
                                       float* pth_e, const crx_lqr_params* prm, float* control, void* stream) {
  CRX_TRACE();
  if (n < 0 || (dim != 4 && dim != 5) || !course_ok(course, true) || (n && (!state || !pe || !pth_e || !control)))
    return fail(CRX_ERR_INVALID, "lqr_steering_control: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_lqr_params p;
  if (prm) p = *prm; else crx_lqr_default_params(&p);
  const unsigned bs = iter_block();
  const crx::CourseView cv = view(course);
  hipStream_t s = (hipStream_t)stream;
  if (use_quad(course, n)) {      // a DPP quad per agent while one agent per lane would leave SIMDs idle
    const dim3 qgrid(blocks_for((size_t)n * 4, crx::kTrackBlock)), qblock(crx::kTrackBlock);
    if (dim == 5)
      hipLaunchKernelGGL((crx::lqr_steering_control_quad_kernel<5>), qgrid, qblock, lds_bytes(course), s, n, state, cv, ind, pe, pth_e, p.dt, p.L, p.eps, p.maxiter, control);
    else
      hipLaunchKernelGGL((crx::lqr_steering_control_quad_kernel<4>), qgrid, qblock, lds_bytes(course), s, n, state, cv, ind, pe, pth_e, p.dt, p.L, p.eps, p.maxiter, control);
    CRX_HIP(hipGetLastError());
    return CRX_OK;
  }
  const dim3 grid(blocks_for(n, bs)), block(bs);
#define CRX_LAUNCH_CTL(DIM, LDS) \
  hipLaunchKernelGGL((crx::lqr_steering_control_kernel<DIM, LDS>), grid, block, (LDS) ? lds_bytes(course) : 0, s, n, state, cv, ind, pe, pth_e, p.dt, p.L, p.eps, p.maxiter, control)
  if (dim == 5) { if (use_lds(course)) CRX_LAUNCH_CTL(5, true); else CRX_LAUNCH_CTL(5, false); }
  else { if (use_lds(course)) CRX_LAUNCH_CTL(4, true); else CRX_LAUNCH_CTL(4, false); }
#undef CRX_LAUNCH_CTL
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_update_batch_dev(int n, float* state, const float* a, const float* delta, const crx_vehicle_params* prm, void* stream) {
  CRX_TRACE();
  if (n < 0 || (n && (!state || !a || !delta))) return fail(CRX_ERR_INVALID, "update: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::update_kernel, dim3(blocks_for(n, crx::kTrackBlock)), dim3(crx::kTrackBlock), 0, (hipStream_t)stream,
                     n, state, a, delta, vparams(prm, 0));
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// lanes_per_agent: 0 = by batch size (a DPP quad per agent while the batch would leave SIMDs idle with one agent per lane, and the
// course fits in LDS), 1 / 4 = forced (crx_x_lqr_closed_loop_lanes_dev)
static int lqr_closed_loop_launch(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                                  const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                                  float* traj_hist, int* ticks_done, void* stream, int lanes_per_agent) {
  if (n < 0 || (dim != 4 && dim != 5) || !course_ok(course, true) || !loop || loop->max_ticks < 0 || (n && !state) ||
      (lanes_per_agent != 0 && lanes_per_agent != 1 && lanes_per_agent != 4))
    return fail(CRX_ERR_INVALID, "lqr_closed_loop: bad argument");
  if (lanes_per_agent == 4 && !use_quad(course, 0)) return fail(CRX_ERR_INVALID, "lqr_closed_loop: the four-lane layout needs a course that fits in LDS");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_lqr_params p;
  if (prm) p = *prm; else crx_lqr_default_params(&p);
  const crx::VehicleParams vp = vparams(veh, 0);
  const unsigned bs = iter_block();
  const crx::CourseView cv = view(course);
  hipStream_t s = (hipStream_t)stream;
  const bool quad = lanes_per_agent == 4 || (lanes_per_agent == 0 && use_quad(course, n));
  if (quad) {
    const dim3 grid(blocks_for((size_t)n * 4, crx::kTrackBlock)), block(crx::kTrackBlock);
#define CRX_LAUNCH_LOOPQ(DIM) \
    hipLaunchKernelGGL((crx::lqr_closed_loop_quad_kernel<DIM>), grid, block, lds_bytes(course), s, n, loop->max_ticks, state, cv, \
                       pe, pth_e, ind, p.dt, p.L, p.eps, p.maxiter, vp, loop->goal_x, loop->goal_y, loop->goal_dis, loop->kp,     \
                       loop->stop_speed, traj_hist, ticks_done)
    if (dim == 5) CRX_LAUNCH_LOOPQ(5); else CRX_LAUNCH_LOOPQ(4);
#undef CRX_LAUNCH_LOOPQ
    CRX_HIP(hipGetLastError());
    return CRX_OK;
  }
  const dim3 grid(blocks_for(n, bs)), block(bs);
#define CRX_LAUNCH_LOOP(DIM, LDS, CHAIN) \
  hipLaunchKernelGGL((crx::lqr_closed_loop_kernel<DIM, LDS, CHAIN>), grid, block, (LDS) ? lds_bytes(course) : 0, s, n, loop->max_ticks, state, cv, \
                     pe, pth_e, ind, p.dt, p.L, p.eps, p.maxiter, vp, loop->goal_x, loop->goal_y, loop->goal_dis, loop->kp,        \
                     loop->stop_speed, traj_hist, ticks_done)
  // one agent per lane: the unmasked Riccati loop while a SIMD holds a wave or two (as crx_dare_from_v_batch_dev does), the masked one beyond
  const bool chain = n <= kDareChainMaxAgents && use_lds(course);
  if (dim == 5) { if (chain) CRX_LAUNCH_LOOP(5, true, true); else if (use_lds(course)) CRX_LAUNCH_LOOP(5, true, false); else CRX_LAUNCH_LOOP(5, false, false); }
  else { if (chain) CRX_LAUNCH_LOOP(4, true, true); else if (use_lds(course)) CRX_LAUNCH_LOOP(4, true, false); else CRX_LAUNCH_LOOP(4, false, false); }
#undef CRX_LAUNCH_LOOP
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}
int crx_lqr_closed_loop_batch_dev(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                                  const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                                  float* traj_hist, int* ticks_done, void* stream) {
  CRX_TRACE();
  return lqr_closed_loop_launch(n, dim, state, course, pe, pth_e, ind, prm, veh, loop, traj_hist, ticks_done, stream, 0);
}
int crx_x_lqr_closed_loop_lanes_dev(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                                    const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                                    float* traj_hist, int* ticks_done, void* stream, int lanes_per_agent) {
  CRX_TRACE();
  return lqr_closed_loop_launch(n, dim, state, course, pe, pth_e, ind, prm, veh, loop, traj_hist, ticks_done, stream, lanes_per_agent);
}

int crx_calc_nearest_index_window_batch_dev(int n, const float* state, const crx_course* course, const int* pind, int nsearch,
                                            int* ind_out, void* stream) {
  CRX_TRACE();
  if (n < 0 || nsearch < 0 || !course_ok(course, false) || (n && (!state || !pind || !ind_out)))
    return fail(CRX_ERR_INVALID, "calc_nearest_index(window): bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::calc_nearest_index_window_kernel, dim3(blocks_for(n, crx::kTrackBlock)), dim3(crx::kTrackBlock), 0,
                     (hipStream_t)stream, n, state, view(course), pind, nsearch, ind_out);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_calc_ref_trajectory_batch_dev(int n, int T, const float* state, const crx_course* course, float dl, double dt,
                                      int nsearch, int* target_ind, float* xref, void* stream) {
  CRX_TRACE();
  if (n < 0 || T < 1 || nsearch < 0 || !course_ok(course, true) || (n && (!state || !target_ind || !xref)))
    return fail(CRX_ERR_INVALID, "calc_ref_trajectory: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::calc_ref_trajectory_kernel, dim3(blocks_for(n, crx::kTrackBlock)), dim3(crx::kTrackBlock), 0,
                     (hipStream_t)stream, n, T, state, view(course), dl, dt, nsearch, target_ind, xref, (const int*)nullptr);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// The persistent kernel keeps everything in registers / private memory: no work buffer is needed (kept for source compatibility
// with 0.1: returns 0).
size_t crx_mpc_closed_loop_work_bytes(int n, int T) {
  (void)n; (void)T;
  return 0;
}

int crx_mpc_closed_loop_batch_dev(int n, int T, float* state, const crx_course* course, float dl, int nsearch,
                                  const crx_mpc_params* prm, const crx_loop_params* loop, int* target_ind, float* traj_hist,
                                  int* ticks_done, void* work, void* stream) {
  CRX_TRACE();
  (void)work;    // the 0.2 signature: ignored, never written (0.3.0 had reused this slot for solve_flags)
  return crx_mpc_closed_loop_flags_batch_dev(n, T, state, course, dl, nsearch, prm, loop, target_ind, traj_hist, ticks_done, nullptr, stream);
}

int crx_mpc_closed_loop_flags_batch_dev(int n, int T, float* state, const crx_course* course, float dl, int nsearch,
                                        const crx_mpc_params* prm, const crx_loop_params* loop, int* target_ind, float* traj_hist,
                                        int* ticks_done, int* solve_flags, void* stream) {
  CRX_TRACE();
  if (n < 0 || T < 2 || T > 64 || !course_ok(course, true) || !loop || loop->max_ticks < 0 ||
      (n && (!state || !target_ind || !ticks_done)))
    return fail(CRX_ERR_INVALID, "mpc_closed_loop: bad argument (2 <= T <= 64)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  const crx::VehicleParams vp{p.dt, p.wb, p.max_steer, p.max_speed, p.min_speed, 1};
  const size_t nn = (size_t)n, nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
  (void)nn; (void)nv;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(blocks_for(n, 64)), block(64);
  const crx::CourseView cv = view(course);
  crx::MpcP q;
  q.dt = p.dt; q.wb = p.wb; q.max_steer = p.max_steer; q.max_accel = p.max_accel; q.max_speed = p.max_speed; q.min_speed = p.min_speed;
  q.r_a = p.r_a; q.r_d = p.r_delta; q.rd_a = p.rd_a; q.rd_d = p.rd_delta; q.qx = p.q_x; q.qy = p.q_y; q.qyaw = p.q_yaw; q.qv = p.q_v;
  q.tol = p.tol; q.max_iter = p.max_iter;
  // ONE persistent kernel for the whole episode (round 1 enqueued three kernels per tick from the host)
  if (T <= 8)
    hipLaunchKernelGGL((crx::mpc_closed_loop_kernel<8>), grid, block, 0, s, n, T, loop->max_ticks, state, cv, dl, nsearch, q, vp, loop->goal_x,
                       loop->goal_y, loop->goal_dis, target_ind, traj_hist, ticks_done, solve_flags);
  else if (T <= 24)
    hipLaunchKernelGGL((crx::mpc_closed_loop_kernel<24>), grid, block, 0, s, n, T, loop->max_ticks, state, cv, dl, nsearch, q, vp, loop->goal_x,
                       loop->goal_y, loop->goal_dis, target_ind, traj_hist, ticks_done, solve_flags);
  else
    hipLaunchKernelGGL((crx::mpc_closed_loop_kernel<CRX_MPC_MAX_T>), grid, block, 0, s, n, T, loop->max_ticks, state, cv, dl, nsearch, q, vp,
                       loop->goal_x, loop->goal_y, loop->goal_dis, target_ind, traj_hist, ticks_done, solve_flags);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// ---- host-pointer variants -----------------------------------------------------------------------
int crx_calc_nearest_index_batch(int n, const float* state, const crx_course* course, int* ind, float* e) {
  CRX_TRACE();
  if (n < 0 || !course_ok(course, false) || (n && (!state || !ind))) return fail(CRX_ERR_INVALID, "calc_nearest_index: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, nullptr, 16 * nl), ii = hc.add(ind + a0, ind + a0, 4 * nl), ie = hc.add(nullptr, e ? e + a0 : nullptr, 4 * nl);
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_calc_nearest_index_batch_dev((int)nl, hc.p<float>(is), &dc, hc.p<int>(ii), hc.p<float>(ie), hc.stream()));
    return hc.finish();
  });
}

int crx_lqr_steering_control_batch(int n, int dim, const float* state, const crx_course* course, int* ind, float* pe,
                                   float* pth_e, const crx_lqr_params* prm, float* control) {
  CRX_TRACE();
  if (n < 0 || (dim != 4 && dim != 5) || !course_ok(course, true) || (n && (!state || !pe || !pth_e || !control)))
    return fail(CRX_ERR_INVALID, "lqr_steering_control: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, nc = dim == 5 ? 2 : 1;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, nullptr, 16 * nl);
    const int ii = hc.add(ind ? ind + a0 : nullptr, ind ? ind + a0 : nullptr, 4 * nl, true);
    const int ip = hc.add(pe + a0, pe + a0, 4 * nl), it = hc.add(pth_e + a0, pth_e + a0, 4 * nl);
    const int ic = hc.add(nullptr, control + nc * a0, 4 * nc * nl);
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_lqr_steering_control_batch_dev((int)nl, dim, hc.p<float>(is), &dc, hc.p<int>(ii), hc.p<float>(ip), hc.p<float>(it), prm,
                                               hc.p<float>(ic), hc.stream()));
    return hc.finish();
  });
}

int crx_update_batch(int n, float* state, const float* a, const float* delta, const crx_vehicle_params* prm) {
  CRX_TRACE();
  if (n < 0 || (n && (!state || !a || !delta))) return fail(CRX_ERR_INVALID, "update: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc;
    CRX_TRY(hc.open());
    const int is = hc.add(state + 4 * a0, state + 4 * a0, 16 * nl), ia = hc.add(a + a0, nullptr, 4 * nl), id = hc.add(delta + a0, nullptr, 4 * nl);
    CRX_TRY(hc.commit());
    CRX_TRY(crx_update_batch_dev((int)nl, hc.p<float>(is), hc.p<float>(ia), hc.p<float>(id), prm, hc.stream()));
    return hc.finish();
  });
}

int crx_lqr_closed_loop_batch(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                              const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                              float* traj_hist, int* ticks_done) {
  CRX_TRACE();
  if (n < 0 || (dim != 4 && dim != 5) || !course_ok(course, true) || !loop || loop->max_ticks < 0 || (n && !state))
    return fail(CRX_ERR_INVALID, "lqr_closed_loop: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, nn = (size_t)n, mt = (size_t)loop->max_ticks;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    hc.forbid_zero_copy();
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, state + 4 * a0, 16 * nl);
    const int ip = hc.add(pe ? pe + a0 : nullptr, pe ? pe + a0 : nullptr, 4 * nl, true);
    const int it = hc.add(pth_e ? pth_e + a0 : nullptr, pth_e ? pth_e + a0 : nullptr, 4 * nl, true);
    const int ii = hc.add(ind ? ind + a0 : nullptr, ind ? ind + a0 : nullptr, 4 * nl, true);
    const int ik = hc.add(nullptr, ticks_done ? ticks_done + a0 : nullptr, 4 * nl);
    // the trajectory is time-major [tick][n][4]: a shard's columns, cleared first (agents that reach the goal stop writing)
    const int ih = traj_hist ? hc.add2d(nullptr, traj_hist + 4 * a0, 16 * nl, mt, 16 * nn, true) : -1;
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_lqr_closed_loop_batch_dev((int)nl, dim, hc.p<float>(is), &dc, hc.p<float>(ip), hc.p<float>(it), hc.p<int>(ii), prm, veh,
                                          loop, ih >= 0 ? hc.p<float>(ih) : nullptr, hc.p<int>(ik), hc.stream()));
    return hc.finish();
  });
}

int crx_calc_nearest_index_window_batch(int n, const float* state, const crx_course* course, const int* pind, int nsearch,
                                        int* ind_out) {
  CRX_TRACE();
  if (n < 0 || nsearch < 0 || !course_ok(course, false) || (n && (!state || !pind || !ind_out)))
    return fail(CRX_ERR_INVALID, "calc_nearest_index(window): bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, nullptr, 16 * nl), ip = hc.add(pind + a0, nullptr, 4 * nl), io = hc.add(nullptr, ind_out + a0, 4 * nl);
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_calc_nearest_index_window_batch_dev((int)nl, hc.p<float>(is), &dc, hc.p<int>(ip), nsearch, hc.p<int>(io), hc.stream()));
    return hc.finish();
  });
}

int crx_mpc_closed_loop_batch(int n, int T, float* state, const crx_course* course, float dl, int nsearch, const crx_mpc_params* prm,
                              const crx_loop_params* loop, int* target_ind, float* traj_hist, int* ticks_done, int* solve_flags) {
  CRX_TRACE();
  if (n < 0 || T < 2 || T > 64 || !course_ok(course, true) || !loop || loop->max_ticks < 0 || (n && !state))
    return fail(CRX_ERR_INVALID, "mpc_closed_loop: bad argument (2 <= T <= 64)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, nn = (size_t)n, mt = (size_t)loop->max_ticks;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    hc.forbid_zero_copy();
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, state + 4 * a0, 16 * nl);
    const int ii = hc.add(target_ind ? target_ind + a0 : nullptr, target_ind ? target_ind + a0 : nullptr, 4 * nl, true);
    const int ik = hc.add(nullptr, ticks_done ? ticks_done + a0 : nullptr, 4 * nl);
    const int iflag = hc.add(nullptr, solve_flags ? solve_flags + a0 : nullptr, 4 * nl);
    const int ih = traj_hist ? hc.add2d(nullptr, traj_hist + 4 * a0, 16 * nl, mt, 16 * nn, true) : -1;
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_mpc_closed_loop_flags_batch_dev((int)nl, T, hc.p<float>(is), &dc, dl, nsearch, prm, loop, hc.p<int>(ii),
                                                ih >= 0 ? hc.p<float>(ih) : nullptr, hc.p<int>(ik), hc.p<int>(iflag), hc.stream()));
    return hc.finish();
  });
}

int crx_calc_ref_trajectory_batch(int n, int T, const float* state, const crx_course* course, float dl, double dt, int nsearch,
                                  int* target_ind, float* xref) {
  CRX_TRACE();
  if (n < 0 || T < 1 || nsearch < 0 || !course_ok(course, true) || (n && (!state || !target_ind || !xref)))
    return fail(CRX_ERR_INVALID, "calc_ref_trajectory: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, nullptr, 16 * nl), ii = hc.add(target_ind + a0, target_ind + a0, 4 * nl);
    const int ix = hc.add(nullptr, xref + 4 * (size_t)T * a0, 16 * (size_t)T * nl);
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_calc_ref_trajectory_batch_dev((int)nl, T, hc.p<float>(is), &dc, dl, dt, nsearch, hc.p<int>(ii), hc.p<float>(ix), hc.stream()));
    return hc.finish();
  });
}

}  // extern "C"

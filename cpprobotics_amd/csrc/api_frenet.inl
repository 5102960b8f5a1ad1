// api_frenet.inl — part of the single translation unit crx_api.hip (#included there, in this order: api_internal, api_core, api_ekf,
// api_lqr, api_mpc, api_track, api_planners, api_frenet, api_probes); the Frenet planner and the host helpers that build courses (Spline2D, speed profile, smooth_yaw).
// ---------------------------------------------------------------------------------------------
// Frenet optimal-trajectory planner
// ---------------------------------------------------------------------------------------------
namespace {

// host mirror of cubic_spline.h's Spline: coefficients of one coordinate over the knots s (cubic_spline.h:53-65,:95-116)
void spline1d_build(const float* x, const float* y, int nx, float* a, float* b, float* c, float* d) {
  std::vector<float> h(nx - 1);
  for (int i = 1; i < nx; ++i) h[i - 1] = x[i] - x[i - 1];
  for (int i = 0; i < nx; ++i) a[i] = y[i];
  // calc_A :95-109 and calc_B :110-116 as the reference fills them (float entries), then A.colPivHouseholderQr().solve(B) :56
  std::vector<float> A((size_t)nx * nx, 0.0f), B(nx, 0.0f), sol(nx, 0.0f);
  auto at = [&](int i, int j) -> float& { return A[i + (size_t)nx * j]; };
  at(0, 0) = 1;
  for (int i = 0; i < nx - 1; ++i) {
    if (i != nx - 2) at(i + 1, i + 1) = 2 * (h[i] + h[i + 1]);
    at(i + 1, i) = h[i];
    at(i, i + 1) = h[i];
  }
  at(0, 1) = 0.0;
  at(nx - 1, nx - 2) = 0.0;
  at(nx - 1, nx - 1) = 1.0;
  for (int i = 0; i < nx - 2; ++i) B[i + 1] = (float)(3.0 * (a[i + 2] - a[i + 1]) / h[i + 1] - 3.0 * (a[i + 1] - a[i]) / h[i]);
  crx::colpiv_qr_solve<crx::kFrMaxKnots>(nx, A.data(), B.data(), sol.data());
  for (int i = 0; i < nx; ++i) c[i] = sol[i];
  for (int i = 0; i < nx - 1; ++i) {
    d[i] = (float)((c[i + 1] - c[i]) / (3.0 * h[i]));
    b[i] = (float)((a[i + 1] - a[i]) / h[i] - h[i] * (c[i + 1] + 2 * c[i]) / 3.0);
  }
  b[nx - 1] = 0.0f; d[nx - 1] = 0.0f;
}

int host_bisect(const float* x, float t, int start, int end) {   // cubic_spline.h:118-127
  for (;;) {
    const int mid = (start + end) / 2;
    if (t == x[mid] || end - start <= 1) return mid;
    if (t > x[mid]) start = mid; else end = mid;
  }
}

struct FrenetGrid { int ndi, nTi, ntv, ntt, min_nt; std::vector<float> ts, Tis; };
FrenetGrid frenet_grid(const crx_frenet_config& g) {   // the loop trip counts of :55-56,:58,:66-68
  FrenetGrid r{0, 0, 0, 0, 1 << 30, {}, {}};
  const int cap = 1 << 16;
  for (float di = (float)(-1 * g.max_road_width); di < g.max_road_width && r.ndi < cap; di += g.d_road_w) ++r.ndi;
  float Tmax = 0.0f;
  std::vector<float> Tis;
  for (float Ti = (float)g.mint; Ti < g.maxt && r.nTi < cap; Ti += g.dt) { ++r.nTi; Tmax = Ti; Tis.push_back(Ti); }
  for (float tv = (float)(g.target_speed - g.d_t_s * g.n_s_sample); tv < g.target_speed + g.d_t_s * g.n_s_sample && r.ntv < cap; tv += g.d_t_s) ++r.ntv;
  std::vector<float> ts;
  for (float t = 0; t < Tmax && r.ntt < cap; t += g.dt) { ++r.ntt; ts.push_back(t); }
  for (float Ti : Tis) { int c = 0; while (c < r.ntt && ts[c] < Ti) ++c; if (c < r.min_nt) r.min_nt = c; }
  r.ts = ts; r.Tis = Tis;
  return r;
}
int frenet_check_cfg(const crx_frenet_config& q, FrenetGrid* out) {
  if (!(q.dt > 0.0) || !(q.d_road_w > 0.0) || !(q.d_t_s > 0.0))
    return fail(CRX_ERR_INVALID, "frenet: dt, d_road_w and d_t_s must be positive");
  const FrenetGrid gr = frenet_grid(q);
  if (gr.ndi < 1 || gr.nTi < 1 || gr.ntv < 1) return fail(CRX_ERR_INVALID, "frenet: the configuration generates no candidate path");
  if (gr.ndi > crx::kFrMaxDi || gr.nTi > crx::kFrMaxTi || gr.ntv > crx::kFrMaxTv || gr.ntt > crx::kFrMaxT ||
      gr.nTi * gr.ntv > crx::kFrMaxCombos || gr.ndi * gr.nTi * gr.ntv > crx::kFrMaxPaths ||
      (size_t)gr.nTi * gr.ntv * gr.ntt * sizeof(crx::FrTab) > (size_t)crx::kFrTabLdsBytes)
    return fail(CRX_ERR_INVALID, "frenet: sample grid too large (<= 64 offsets, horizons x speeds <= 64, <= 64 time steps, "
                                 "horizons x speeds x time steps <= 2048)");
  if (gr.min_nt < 2) return fail(CRX_ERR_INVALID, "frenet: every horizon needs at least two time steps (mint > dt)");
  if (out) *out = gr;
  return CRX_OK;
}

}  // namespace

extern "C" {

void crx_frenet_default_config(crx_frenet_config* c) {
  if (!c) return;
  c->max_speed = 50.0 / 3.6; c->max_accel = 2.0; c->max_curvature = 1.0; c->max_road_width = 7.0; c->d_road_w = 1.0;
  c->dt = 0.2; c->maxt = 5.0; c->mint = 4.0; c->target_speed = 30.0 / 3.6; c->d_t_s = 5.0 / 3.6; c->n_s_sample = 1;
  c->robot_radius = 1.5; c->kj = 0.1; c->kt = 0.1; c->kd = 1.0; c->klat = 1.0; c->klon = 1.0;
}

int crx_frenet_num_paths(const crx_frenet_config* cfg) {
  crx_frenet_config q;
  if (cfg) q = *cfg; else crx_frenet_default_config(&q);
  FrenetGrid gr;
  if (int rc = frenet_check_cfg(q, &gr)) return rc;
  return gr.ndi * gr.nTi * gr.ntv;
}

int crx_frenet_spline_build(const float* wx, const float* wy, int nx, float* coef) {
  CRX_TRACE();
  if (!wx || !wy || !coef || nx < 2 || nx > crx::kFrMaxKnots) return fail(CRX_ERR_INVALID, "frenet_spline_build: bad argument (2 <= nx <= 64)");
  float* s = coef;
  s[0] = 0.0f;                                   // Spline2D::calc_s :172-186
  float temp = 0;
  for (int i = 1; i < nx; ++i) {
    const float dx = wx[i] - wx[i - 1], dy = wy[i] - wy[i - 1];
    temp += std::sqrt(dx * dx + dy * dy);
    s[i] = temp;
    if (!(s[i] > s[i - 1])) return fail(CRX_ERR_INVALID, "frenet_spline_build: consecutive way-points must be distinct");
  }
  spline1d_build(s, wx, nx, coef + nx, coef + 2 * nx, coef + 3 * nx, coef + 4 * nx);
  spline1d_build(s, wy, nx, coef + 5 * nx, coef + 6 * nx, coef + 7 * nx, coef + 8 * nx);
  return CRX_OK;
}

int crx_frenet_course_samples(const float* coef, int nx, float* rx, float* ry, int cap) {
  CRX_TRACE();
  if (!coef || nx < 2 || cap < 0 || (cap && (!rx || !ry))) return fail(CRX_ERR_INVALID, "frenet_course_samples: bad argument");
  const float* s = coef;
  int k = 0;
  for (float i = 0; i < s[nx - 1]; i += 0.1) {   // main :205-213
    if (k < cap) {
      const int seg = host_bisect(s, i, 0, nx);
      const float dx = i - s[seg];
      rx[k] = coef[nx + seg] + coef[2 * nx + seg] * dx + coef[3 * nx + seg] * dx * dx + coef[4 * nx + seg] * dx * dx * dx;
      ry[k] = coef[5 * nx + seg] + coef[6 * nx + seg] * dx + coef[7 * nx + seg] * dx * dx + coef[8 * nx + seg] * dx * dx * dx;
    }
    ++k;
  }
  return k;
}

// The course the reference's LQR / MPC mains build from their way-points: Spline2D(wx, wy) sampled every `ds`
// (src/lqr_speed_steer_control.cpp:252-265 with ds = 0.1, src/model_predictive_control.cpp:473-486 with ds = 1.0): position
// (calc_postion), heading (calc_yaw = atan2 of the first derivatives) and curvature (calc_curvature) per sample.  Host, once
// per course.  Returns the number of samples; fills up to cap of each non-null array.
int crx_course_from_waypoints(const float* wx, const float* wy, int nx, double ds, float* cx, float* cy, float* cyaw, float* ck, int cap) {
  CRX_TRACE();
  if (!wx || !wy || nx < 2 || nx > crx::kFrMaxKnots || !(ds > 0.0) || cap < 0) return fail(CRX_ERR_INVALID, "course_from_waypoints: bad argument");
  std::vector<float> coef(9 * (size_t)nx);
  if (int rc = crx_frenet_spline_build(wx, wy, nx, coef.data())) return rc;
  const float* s = coef.data();
  const float *ax = s + nx, *bx = s + 2 * nx, *cxx = s + 3 * nx, *dx_ = s + 4 * nx, *ay = s + 5 * nx, *by = s + 6 * nx, *cyy = s + 7 * nx, *dy_ = s + 8 * nx;
  if (!((float)((double)s[nx - 1] + ds) > s[nx - 1])) return fail(CRX_ERR_INVALID, "course_from_waypoints: ds too small for this course (the float walk would not advance)");
  int k = 0;
  for (float i = 0; i < s[nx - 1]; i += ds) {                      // float i += double literal, as the mains write it
    if (k < cap) {
      const int seg = host_bisect(s, i, 0, nx), segd = host_bisect(s, i, 0, nx - 1);   // calc / calc_dd use bisect(t,0,nx), calc_d bisect(t,0,nx-1)
      const float e = i - s[seg], ed = i - s[segd];
      if (cx) cx[k] = ax[seg] + bx[seg] * e + cxx[seg] * e * e + dx_[seg] * e * e * e;
      if (cy) cy[k] = ay[seg] + by[seg] * e + cyy[seg] * e * e + dy_[seg] * e * e * e;
      const float d1x = bx[segd] + 2 * cxx[segd] * ed + 3 * dx_[segd] * ed * ed;
      const float d1y = by[segd] + 2 * cyy[segd] * ed + 3 * dy_[segd] * ed * ed;
      if (cyaw) cyaw[k] = std::atan2(d1y, d1x);
      if (ck) {
        const float ddx = 2 * cxx[seg] + 6 * dx_[seg] * e, ddy = 2 * cyy[seg] + 6 * dy_[seg] * e;
        ck[k] = (ddy * d1x - ddx * d1y) / (d1x * d1x + d1y * d1y);
      }
    }
    ++k;
  }
  return k;
}

// calc_speed_profile of the two tracking files.  variant 5 (src/lqr_speed_steer_control.cpp:40-62): direction flips where the
// heading jumps by pi/4..pi/2, zero at the switch points, then the last 39 entries ramp down as target/(50-k) with a floor of
// 1/3.6 — the reference's k = 0 pass writes one element PAST the end of the vector (:55-56); that write is not made here.
// variant 0 (src/model_predictive_control.cpp:83-105): sign from the direction of travel against the heading; the reference's
// `speed_profile[-1] = 0.0` (:102) writes BEFORE the vector, so the last entry keeps its value, as here.
int crx_calc_speed_profile(int variant, const float* rx, const float* ry, const float* ryaw, int n, float target_speed, float* sp) {
  CRX_TRACE();
  if ((variant != 0 && variant != 4 && variant != 5) || n < 1 || !ryaw || !sp || (variant == 0 && (!rx || !ry)))
    return fail(CRX_ERR_INVALID, "calc_speed_profile: bad argument (variant 5, 4 or 0)");
  for (int i = 0; i < n; ++i) sp[i] = target_speed;
  float direction = 1.0;
  if (variant == 5 || variant == 4) {
    for (int i = 0; i + 1 < n; ++i) {
      const float dyaw = std::abs(ryaw[i + 1] - ryaw[i]);
      const float switch_point = (M_PI / 4.0 < dyaw) && (dyaw < M_PI / 2.0);
      if (switch_point) direction = direction * -1;
      if (direction != 1.0) sp[i] = target_speed * -1; else sp[i] = target_speed;
      if (switch_point) sp[i] = 0.0;
    }
    if (variant == 5) {
      for (int k = 1; k < 40 && k <= n; ++k) {          // :55-60 (its k = 0 writes past the end and is not made)
        sp[n - k] = target_speed / (50 - k);
        if (sp[n - k] <= 1.0 / 3.6) sp[n - k] = 1.0 / 3.6;
      }
    } else {
      sp[n - 1] = 0.0;                                  // src/lqr_steer_control.cpp:50
    }
  } else {
    for (int i = 0; i + 1 < n; ++i) {
      const float dx = rx[i + 1] - rx[i], dy = ry[i + 1] - ry[i];
      const float move_direction = std::atan2(dy, dx);
      if (dx != 0.0 && dy != 0.0) {
        const double a = (double)(move_direction - ryaw[i]);
        const float dangle = std::abs((float)(std::fmod(std::fmod(a + M_PI, 2 * M_PI) - 2 * M_PI, 2 * M_PI) + M_PI));   // YAW_P2P, motion_model.h:18
        if (dangle >= M_PI / 4.0) direction = -1.0; else direction = 1.0;
      }
      if (direction != 1.0) sp[i] = -1 * target_speed; else sp[i] = target_speed;
    }
  }
  return CRX_OK;
}

int crx_smooth_yaw(float* cyaw, int n) {   // src/model_predictive_control.cpp:172-185
  if (n < 0 || (n && !cyaw)) return fail(CRX_ERR_INVALID, "smooth_yaw: bad argument");
  for (int i = 0; i + 1 < n; ++i) {
    float dyaw = cyaw[i + 1] - cyaw[i];
    if (!std::isfinite(dyaw)) return fail(CRX_ERR_INVALID, "smooth_yaw: non-finite heading");
    while (dyaw > M_PI / 2.0) {
      const float before = cyaw[i + 1];
      cyaw[i + 1] -= M_PI * 2.0;
      if (cyaw[i + 1] == before) return fail(CRX_ERR_INVALID, "smooth_yaw: heading too large to unwind in float");
      dyaw = cyaw[i + 1] - cyaw[i];
    }
    while (dyaw < -M_PI / 2.0) {
      const float before = cyaw[i + 1];
      cyaw[i + 1] += M_PI * 2.0;
      if (cyaw[i + 1] == before) return fail(CRX_ERR_INVALID, "smooth_yaw: heading too large to unwind in float");
      dyaw = cyaw[i + 1] - cyaw[i];
    }
  }
  return CRX_OK;
}


int crx_frenet_run_batch_dev(int n, int max_ticks, float* state, const float* coef, int nx, const float* goal_xy,
                             const float* ob, int nob, const crx_frenet_config* cfg, float* hist, int* ticks_done,
                             int* status, int* best_idx, int* n_valid, float* path_cf, int* path_ok, int path_cap,
                             void* stream) {
  CRX_TRACE();
  if (n < 0 || max_ticks < 0 || nob < 0 || nob > crx::kFrMaxOb || (nob && !ob) || nx < 2 || nx > crx::kFrMaxKnots || !coef ||
      !goal_xy || path_cap < 0 || (n && !state))
    return fail(CRX_ERR_INVALID, "frenet_run: bad argument (2 <= nx <= 64, nob <= 128)");
  crx_frenet_config q;
  if (cfg) q = *cfg; else crx_frenet_default_config(&q);
  FrenetGrid gr;
  if (int rc = frenet_check_cfg(q, &gr)) return rc;
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx::FrenetCfg c;
  static_assert(sizeof(c) == sizeof(q), "config layouts must agree");
  std::memcpy(&c, &q, sizeof(c));
  // the per-wave (combo, time step) table lives in dynamic LDS: as many waves per block as fit beside it
  const int stride = gr.nTi * gr.ntv * gr.ntt;
  int wpb = crx::kFrWavesPerBlock;
  while (wpb > 1 && (size_t)wpb * stride * sizeof(crx::FrTab) > (size_t)crx::kFrTabLdsBytes) wpb >>= 1;
  // std::pow(t, k), k = 2..5, of the time grid and of the horizons: libm's own values, what the reference's polynomial classes
  // call (quintic_polynomial.h:41-68, quartic_polynomial.h:39-59) — float argument and int exponent promoted to double
  crx::FrPowArg pw;
  std::memset(&pw, 0, sizeof(pw));
  auto powers = [](float x) { return crx::FrPow{std::pow((double)x, 2.0), std::pow((double)x, 3.0), std::pow((double)x, 4.0), std::pow((double)x, 5.0)}; };
  for (int i = 0; i < gr.ntt; ++i) pw.t[i] = powers(gr.ts[i]);
  for (int i = 0; i < gr.nTi; ++i) pw.T[i] = powers(gr.Tis[i]);
  hipLaunchKernelGGL(crx::frenet_run_kernel, dim3(blocks_for(n, wpb)), dim3(64 * wpb), (size_t)wpb * stride * sizeof(crx::FrTab),
                     (hipStream_t)stream, n, max_ticks, state, coef, nx, goal_xy[0], goal_xy[1], ob, nob, c, hist, ticks_done,
                     status, best_idx, n_valid, path_cf, path_ok, path_cap, stride, pw);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

}  // extern "C"

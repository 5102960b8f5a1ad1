// frenet_ref.cpp — CPU restatement of the reference's Frenet optimal-trajectory planner.  TEST INFRASTRUCTURE ONLY: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  
// PINNED against the reference's own lines (oracle/ref_build.sh compiles them unmodified — against the host's Eigen, or against the
// Eigen stand-in oracle/ref_shim/Eigen/Eigen where there is none — and tests/test_oracle_vs_ref.py demands equal bits); unpinned only
// with respect to Eigen's own binary, absent from every host of this project.
//
// Follows /root/reference/src/frenet_optimal_trajectory.cpp:
//   calc_frenet_paths :51-100, calc_global_paths :102-136, check_collision :138-148, check_paths :150-158,
//   frenet_optimal_planning :160-176, main loop :224-236 (state hand-over and goal test),
// and /root/reference/include/quintic_polynomial.h:39-69, quartic_polynomial.h:37-64, cubic_spline.h:39-128 (Spline),
// :130-178 (Spline2D).  Every expression keeps the reference's C++ types: float members, the DT/MAX_* macros as double
// literals, std::pow(float,int) and std::cos(float+double) evaluated in double, std::atan2/std::sqrt of floats in float.
// Things the reference does that a reader might take for typos are kept, because they decide the numbers:
//   * QuinticPolynomial::calc_first_derivative ends in a5*t^4, not 5*a5*t^4 (quintic_polynomial.h:53);
//   * max_speed / max_accel / max_curvature start at numeric_limits<float>::min() (the smallest positive normal).
// The 3x3 / 2x2 / nx x nx float systems, which the reference hands to Eigen's colPivHouseholderQr in float
// (quintic_polynomial.h:49, quartic_polynomial.h:45, cubic_spline.h:56), are solved by the float restatement of that
// algorithm in oracle/eigen_qr.h (round 1 solved them exactly in double; the measured effect of the change is in DESIGN.md 5e).
// This file is pinned against the reference's own lines: tests/test_oracle_vs_ref.py runs oracle/_ref/libref.so
// (oracle/ref_shim/ref_frenet.cpp) on the same inputs and demands equal bits.  Where the reference runs into undefined behaviour (a path with fewer than two points on the
// course: vector::back() of an empty vector, size()-1 wrapping) the path is dropped; where it would throw (s before the
// course) the path is dropped and status bit 2 set; no surviving path ends the agent's episode with status bit 0.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#include "eigen_qr.h"

namespace {

struct FrenetCfg {   // the #defines :20-38, as the double expressions they expand to
  double max_speed, max_accel, max_curvature, max_road_width, d_road_w, dt, maxt, mint, target_speed, d_t_s;
  int n_s_sample;
  double robot_radius, kj, kt, kd, klat, klon;
};

// ---- cubic_spline.h ----------------------------------------------------------------------------------------------
// A.colPivHouseholderQr().solve(B) for a column-major float system (oracle/eigen_qr.h)
// fixed_size: the reference's A is a fixed-size Eigen matrix (Matrix3f / Matrix2f) — its initial column norms are reduced by a tree
// of halves, not ascending (oracle/eigen_qr.h (1))
void solve_qr(int n, const std::vector<float>& A, const std::vector<float>& b, std::vector<float>& x, bool fixed_size = false) {
  oracle::ColPivQR<float> qr(n, n, fixed_size);
  qr.compute(A.data());
  x.resize(n);
  qr.solve(b.data(), x.data());
}

struct Spline {   // cubic_spline.h:39-128
  std::vector<float> x, a, b, c, d;
  int nx = 0;
  Spline() {}
  Spline(const std::vector<float>& x_, const std::vector<float>& y_) : x(x_), a(y_), nx((int)x_.size()) {
    std::vector<float> h(nx - 1);
    for (int i = 1; i < nx; ++i) h[i - 1] = x[i] - x[i - 1];
    std::vector<float> A((size_t)nx * nx, 0.0f), B(nx, 0.0f);      // column-major: A(i, j) = A[i + nx * j]
    auto at = [&](int i, int j) -> float& { return A[i + (size_t)nx * j]; };
    at(0, 0) = 1;                                                   // calc_A :95-109
    for (int i = 0; i < nx - 1; ++i) {
      if (i != nx - 2) at(i + 1, i + 1) = 2 * (h[i] + h[i + 1]);
      at(i + 1, i) = h[i];
      at(i, i + 1) = h[i];
    }
    at(0, 1) = 0.0;
    at(nx - 1, nx - 2) = 0.0;
    at(nx - 1, nx - 1) = 1.0;
    for (int i = 0; i < nx - 2; ++i)                                 // calc_B :110-116 (double expression, float entry)
      B[i + 1] = (float)(3.0 * (a[i + 2] - a[i + 1]) / h[i + 1] - 3.0 * (a[i + 1] - a[i]) / h[i]);
    solve_qr(nx, A, B, c);
    for (int i = 0; i < nx - 1; ++i) {                               // :61-64
      d.push_back((float)((c[i + 1] - c[i]) / (3.0 * h[i])));
      b.push_back((float)((a[i + 1] - a[i]) / h[i] - h[i] * (c[i + 1] + 2 * c[i]) / 3.0));
    }
  }
  int bisect(float t, int start, int end) const {                    // :118-127
    const int mid = (start + end) / 2;
    if (t == x[mid] || end - start <= 1) return mid;
    else if (t > x[mid]) return bisect(t, mid, end);
    else return bisect(t, start, mid);
  }
  float calc(float t) const {                                        // :67-74
    const int seg = bisect(t, 0, nx);
    const float dx = t - x[seg];
    return a[seg] + b[seg] * dx + c[seg] * dx * dx + d[seg] * dx * dx * dx;
  }
  float calc_d(float t) const {                                      // :76-83
    const int seg = bisect(t, 0, nx - 1);
    const float dx = t - x[seg];
    return b[seg] + 2 * c[seg] * dx + 3 * d[seg] * dx * dx;
  }
};

struct Spline2D {   // :130-178
  Spline sx, sy;
  std::vector<float> s;
  Spline2D() {}
  Spline2D(const std::vector<float>& x, const std::vector<float>& y) {
    s.push_back(0.0f);                                               // calc_s :164-177
    float temp = 0;
    for (size_t i = 1; i < x.size(); ++i) {
      const float dx = x[i] - x[i - 1], dy = y[i] - y[i - 1];
      temp += std::sqrt(dx * dx + dy * dy);
      s.push_back(temp);
    }
    sx = Spline(s, x);
    sy = Spline(s, y);
  }
  // the flat coefficient table both the oracle entry points and the crx kernel take: rows s, ax,bx,cx,dx, ay,by,cy,dy
  void to_table(float* coef) const {
    const int nx = (int)s.size();
    std::memset(coef, 0, sizeof(float) * 9 * nx);
    for (int i = 0; i < nx; ++i) {
      coef[i] = s[i]; coef[1 * nx + i] = sx.a[i]; coef[3 * nx + i] = sx.c[i]; coef[5 * nx + i] = sy.a[i]; coef[7 * nx + i] = sy.c[i];
      if (i < nx - 1) { coef[2 * nx + i] = sx.b[i]; coef[4 * nx + i] = sx.d[i]; coef[6 * nx + i] = sy.b[i]; coef[8 * nx + i] = sy.d[i]; }
    }
  }
  static Spline2D from_table(const float* coef, int nx) {
    Spline2D sp;
    sp.s.assign(coef, coef + nx);
    for (int k = 0; k < 2; ++k) {
      Spline& q = k ? sp.sy : sp.sx;
      const float* base = coef + (size_t)(1 + 4 * k) * nx;
      q.nx = nx; q.x = sp.s;
      q.a.assign(base, base + nx); q.b.assign(base + nx, base + 2 * nx - 1);
      q.c.assign(base + 2 * nx, base + 3 * nx); q.d.assign(base + 3 * nx, base + 4 * nx - 1);
    }
    return sp;
  }
};

// ---- quintic_polynomial.h / quartic_polynomial.h --------------------------------------------------------------------
// x = A.colPivHouseholderQr().solve(B), in float; A row by row as the comma initialisers of the reference fill it
void solve3(const float A[3][3], const float B[3], float x[3]) {
  std::vector<float> Am(9), Bm(B, B + 3), xv;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Am[i + 3 * j] = A[i][j];
  solve_qr(3, Am, Bm, xv, true);      // Eigen::Matrix3f A  (quintic_polynomial.h:39)
  x[0] = xv[0]; x[1] = xv[1]; x[2] = xv[2];
}
void solve2(const float A[2][2], const float B[2], float x[2]) {
  std::vector<float> Am(4), Bm(B, B + 2), xv;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) Am[i + 2 * j] = A[i][j];
  solve_qr(2, Am, Bm, xv, true);      // Eigen::Matrix2f A  (quartic_polynomial.h:36)
  x[0] = xv[0]; x[1] = xv[1];
}

struct Quintic {   // quintic_polynomial.h:39-69
  float a0, a1, a2, a3, a4, a5;
  Quintic(float xs, float vxs, float axs, float xe, float vxe, float axe, float T) : a0(xs), a1(vxs), a2((float)(axs / 2.0)) {
    const float A[3][3] = {{(float)std::pow(T, 3), (float)std::pow(T, 4), (float)std::pow(T, 5)},
                           {(float)(3 * std::pow(T, 2)), (float)(4 * std::pow(T, 3)), (float)(5 * std::pow(T, 4))},
                           {6 * T, (float)(12 * std::pow(T, 2)), (float)(20 * std::pow(T, 3))}};
    const float B[3] = {(float)(xe - a0 - a1 * T - a2 * std::pow(T, 2)), vxe - a1 - 2 * a2 * T, axe - 2 * a2};
    float c[3];
    solve3(A, B, c);
    a3 = c[0]; a4 = c[1]; a5 = c[2];
  }
  float calc_point(float t) const { return (float)(a0 + a1 * t + a2 * std::pow(t, 2) + a3 * std::pow(t, 3) + a4 * std::pow(t, 4) + a5 * std::pow(t, 5)); }
  float calc_first_derivative(float t) const { return (float)(a1 + 2 * a2 * t + 3 * a3 * std::pow(t, 2) + 4 * a4 * std::pow(t, 3) + a5 * std::pow(t, 4)); }
  float calc_second_derivative(float t) const { return (float)(2 * a2 + 6 * a3 * t + 12 * a4 * std::pow(t, 2) + 20 * a5 * std::pow(t, 3)); }
  float calc_third_derivative(float t) const { return (float)(6 * a3 + 24 * a4 * t + 60 * a5 * std::pow(t, 2)); }
};

struct Quartic {   // quartic_polynomial.h:37-64
  float a0, a1, a2, a3, a4;
  Quartic(float xs, float vxs, float axs, float vxe, float axe, float T) : a0(xs), a1(vxs), a2((float)(axs / 2.0)) {
    const float A[2][2] = {{(float)(3 * std::pow(T, 2)), (float)(4 * std::pow(T, 3))}, {6 * T, (float)(12 * std::pow(T, 2))}};
    const float B[2] = {vxe - a1 - 2 * a2 * T, axe - 2 * a2};
    float c[2];
    solve2(A, B, c);
    a3 = c[0]; a4 = c[1];
  }
  float calc_point(float t) const { return (float)(a0 + a1 * t + a2 * std::pow(t, 2) + a3 * std::pow(t, 3) + a4 * std::pow(t, 4)); }
  float calc_first_derivative(float t) const { return (float)(a1 + 2 * a2 * t + 3 * a3 * std::pow(t, 2) + 4 * a4 * std::pow(t, 3)); }
  float calc_second_derivative(float t) const { return (float)(2 * a2 + 6 * a3 * t + 12 * a4 * std::pow(t, 2)); }
  float calc_third_derivative(float t) const { return 6 * a3 + 24 * a4 * t; }
};

// ---- frenet_optimal_trajectory.cpp ------------------------------------------------------------------------------------
struct FrenetPath {   // frenet_path.h
  float cd = 0, cv = 0, cf = 0;
  std::vector<float> t, d, d_d, d_dd, d_ddd, s, s_d, s_dd, s_ddd, x, y, yaw, ds, c;
  float max_speed = 0, max_accel = 0, max_curvature = 0;
  bool dropped = false;      // see the file header: <2 course points or s before the course
};

float sum_of_power(const std::vector<float>& v) {   // :42-48
  float sum = 0;
  for (float item : v) sum += item * item;
  return sum;
}

std::vector<FrenetPath> calc_frenet_paths(const FrenetCfg& g, float c_speed, float c_d, float c_d_d, float c_d_dd, float s0) {   // :51-100
  std::vector<FrenetPath> fp_list;
  for (float di = (float)(-1 * g.max_road_width); di < g.max_road_width; di += g.d_road_w) {
    for (float Ti = (float)g.mint; Ti < g.maxt; Ti += g.dt) {
      FrenetPath fp;
      Quintic lat_qp(c_d, c_d_d, c_d_dd, di, 0.0f, 0.0f, Ti);
      for (float t = 0; t < Ti; t += g.dt) {
        fp.t.push_back(t);
        fp.d.push_back(lat_qp.calc_point(t));
        fp.d_d.push_back(lat_qp.calc_first_derivative(t));
        fp.d_dd.push_back(lat_qp.calc_second_derivative(t));
        fp.d_ddd.push_back(lat_qp.calc_third_derivative(t));
      }
      for (float tv = (float)(g.target_speed - g.d_t_s * g.n_s_sample); tv < g.target_speed + g.d_t_s * g.n_s_sample; tv += g.d_t_s) {
        FrenetPath fp_bot = fp;
        Quartic lon_qp(s0, c_speed, 0.0f, tv, 0.0f, Ti);
        fp_bot.max_speed = std::numeric_limits<float>::min();
        fp_bot.max_accel = std::numeric_limits<float>::min();
        for (float t_ : fp.t) {
          fp_bot.s.push_back(lon_qp.calc_point(t_));
          fp_bot.s_d.push_back(lon_qp.calc_first_derivative(t_));
          fp_bot.s_dd.push_back(lon_qp.calc_second_derivative(t_));
          fp_bot.s_ddd.push_back(lon_qp.calc_third_derivative(t_));
          if (fp_bot.s_d.back() > fp_bot.max_speed) fp_bot.max_speed = fp_bot.s_d.back();
          if (fp_bot.s_dd.back() > fp_bot.max_accel) fp_bot.max_accel = fp_bot.s_dd.back();
        }
        const float Jp = sum_of_power(fp.d_ddd);
        const float Js = sum_of_power(fp_bot.s_ddd);
        const float ds = (float)(g.target_speed - fp_bot.s_d.back());
        fp_bot.cd = (float)(g.kj * Jp + g.kt * Ti + g.kd * std::pow(fp_bot.d.back(), 2));
        fp_bot.cv = (float)(g.kj * Js + g.kt * Ti + g.kd * ds);
        fp_bot.cf = (float)(g.klat * fp_bot.cd + g.klon * fp_bot.cv);
        fp_list.push_back(fp_bot);
      }
    }
  }
  return fp_list;
}

int calc_global_paths(std::vector<FrenetPath>& path_list, const Spline2D& csp) {   // :102-136
  int st = 0;
  for (FrenetPath& p : path_list) {
    for (size_t i = 0; i < p.s.size(); ++i) {
      if (p.s[i] >= csp.s.back()) break;
      if (p.s[i] < csp.s.front()) { p.dropped = true; st |= 4; break; }   // Spline::calc would throw
      const float px = csp.sx.calc(p.s[i]), py = csp.sy.calc(p.s[i]);
      const float iyaw = std::atan2(csp.sy.calc_d(p.s[i]), csp.sx.calc_d(p.s[i]));
      const float di = p.d[i];
      const float x = (float)(px + di * std::cos(iyaw + M_PI / 2.0));
      const float y = (float)(py + di * std::sin(iyaw + M_PI / 2.0));
      p.x.push_back(x);
      p.y.push_back(y);
    }
    if (p.x.size() < 2) { p.dropped = true; continue; }
    for (size_t i = 0; i + 1 < p.x.size(); ++i) {
      const float dx = p.x[i + 1] - p.x[i], dy = p.y[i + 1] - p.y[i];
      p.yaw.push_back(std::atan2(dy, dx));
      p.ds.push_back(std::sqrt(dx * dx + dy * dy));
    }
    p.yaw.push_back(p.yaw.back());
    p.ds.push_back(p.ds.back());
    p.max_curvature = std::numeric_limits<float>::min();
    for (size_t i = 0; i + 1 < p.x.size(); ++i) {
      p.c.push_back((p.yaw[i + 1] - p.yaw[i]) / p.ds[i]);
      if (p.c.back() > p.max_curvature) p.max_curvature = p.c.back();
    }
  }
  return st;
}

bool check_collision(const FrenetCfg& g, const FrenetPath& path, const float* ob, int nob) {   // :138-148
  for (int k = 0; k < nob; ++k)
    for (size_t i = 0; i < path.x.size(); ++i) {
      const float dist = (float)(std::pow((path.x[i] - ob[2 * k]), 2) + std::pow((path.y[i] - ob[2 * k + 1]), 2));
      if (dist <= g.robot_radius * g.robot_radius) return false;
    }
  return true;
}

struct PlanOut {
  int best = -1, n_paths = 0, n_valid = 0, st = 0;
  float cf = 0, s1 = 0, d1 = 0, d_d1 = 0, d_dd1 = 0, s_d1 = 0, x1 = 0, y1 = 0;
};

// frenet_optimal_planning :160-176 (+ check_paths :150-158); path_cf / path_ok (may be null) receive every path's cost
// and whether it survived the checks, in generation order.
PlanOut plan(const FrenetCfg& g, const Spline2D& csp, const float st5[5], const float* ob, int nob, float* path_cf, int* path_ok) {
  const float s0 = st5[0], c_speed = st5[1], c_d = st5[2], c_d_d = st5[3], c_d_dd = st5[4];
  std::vector<FrenetPath> fp_list = calc_frenet_paths(g, c_speed, c_d, c_d_d, c_d_dd, s0);
  PlanOut o;
  o.st = calc_global_paths(fp_list, csp);
  o.n_paths = (int)fp_list.size();
  float min_cost = std::numeric_limits<float>::max();
  for (int p = 0; p < o.n_paths; ++p) {
    const FrenetPath& path = fp_list[p];
    const bool ok = !path.dropped && path.max_speed < g.max_speed && path.max_accel < g.max_accel &&
                    path.max_curvature < g.max_curvature && check_collision(g, path, ob, nob);
    if (path_cf) path_cf[p] = path.cf;
    if (path_ok) path_ok[p] = ok ? 1 : 0;
    if (!ok) continue;
    ++o.n_valid;
    if (min_cost >= path.cf) { min_cost = path.cf; o.best = p; }
  }
  if (o.best >= 0) {
    const FrenetPath& f = fp_list[o.best];
    o.cf = f.cf; o.s1 = f.s[1]; o.d1 = f.d[1]; o.d_d1 = f.d_d[1]; o.d_dd1 = f.d_dd[1]; o.s_d1 = f.s_d[1]; o.x1 = f.x[1]; o.y1 = f.y[1];
  } else {
    o.st |= 1;
  }
  return o;
}

}  // namespace

extern "C" {

// Spline2D(wx, wy) -> coefficient table coef[9][nx] (rows s, ax,bx,cx,dx, ay,by,cy,dy)
void oracle_frenet_spline_build(const float* wx, const float* wy, int nx, float* coef) {
  Spline2D sp(std::vector<float>(wx, wx + nx), std::vector<float>(wy, wy + nx));
  sp.to_table(coef);
}

// main :205-213: the course sampled every 0.1 (float accumulation); returns the number of samples, the last one is the goal
int oracle_frenet_course_samples(const float* coef, int nx, float* rx, float* ry, int cap) {
  Spline2D sp = Spline2D::from_table(coef, nx);
  int k = 0;
  for (float i = 0; i < sp.s.back(); i += 0.1) {
    if (k < cap) { rx[k] = sp.sx.calc(i); ry[k] = sp.sy.calc(i); }
    ++k;
  }
  return k;
}

// one planning call for n agents; state rows (s0, c_speed, c_d, c_d_d, c_d_dd); out rows (s1, s_d1, d1, d_d1, d_dd1, x1, y1, cf)
void oracle_frenet_plan(int n, const float* state, const float* coef, int nx, const float* ob, int nob, const FrenetCfg* g,
                        float* out, int* best, int* n_valid, int* n_paths, int* status, float* path_cf, int* path_ok, int path_cap) {
  Spline2D sp = Spline2D::from_table(coef, nx);
#pragma omp parallel for schedule(dynamic, 4)
  for (int a = 0; a < n; ++a) {
    std::vector<float> cf(4096, 0.0f);
    std::vector<int> ok(4096, 0);
    const PlanOut o = plan(*g, sp, state + 5 * a, ob, nob, cf.data(), ok.data());
    float* r = out + 8 * (size_t)a;
    r[0] = o.s1; r[1] = o.s_d1; r[2] = o.d1; r[3] = o.d_d1; r[4] = o.d_dd1; r[5] = o.x1; r[6] = o.y1; r[7] = o.cf;
    best[a] = o.best; n_valid[a] = o.n_valid; n_paths[a] = o.n_paths; status[a] = o.st;
    for (int p = 0; p < o.n_paths && p < path_cap; ++p) {
      if (path_cf) path_cf[(size_t)a * path_cap + p] = cf[p];
      if (path_ok) path_ok[(size_t)a * path_cap + p] = ok[p];
    }
  }
}

// main :224-236: plan, hand the second sample of the winner over as the new state, stop within 1 m of the goal
void oracle_frenet_run(int n, int max_ticks, float* state, const float* coef, int nx, const float* goal, const float* ob, int nob,
                       const FrenetCfg* g, float* hist, int* ticks_done, int* status, int* best_idx, int* n_valid) {
  Spline2D sp = Spline2D::from_table(coef, nx);
#pragma omp parallel for schedule(dynamic, 1)
  for (int a = 0; a < n; ++a) {
    float* st5 = state + 5 * (size_t)a;
    int st = 0, ticks = 0, lb = -1, lv = 0;
    for (int tick = 0; tick < max_ticks; ++tick) {
      const PlanOut o = plan(*g, sp, st5, ob, nob, nullptr, nullptr);
      st |= o.st; lb = o.best; lv = o.n_valid;
      if (o.best < 0) break;
      st5[0] = o.s1; st5[1] = o.s_d1; st5[2] = o.d1; st5[3] = o.d_d1; st5[4] = o.d_dd1;
      ticks = tick + 1;
      if (hist) {
        float* h = hist + ((size_t)tick * n + a) * 8;
        h[0] = o.s1; h[1] = o.s_d1; h[2] = o.d1; h[3] = o.d_d1; h[4] = o.d_dd1; h[5] = o.x1; h[6] = o.y1; h[7] = o.cf;
      }
      if (std::pow((o.x1 - goal[0]), 2) + std::pow((o.y1 - goal[1]), 2) <= 1.0) break;
    }
    ticks_done[a] = ticks; status[a] = st; best_idx[a] = lb; n_valid[a] = lv;
  }
}

}  // extern "C"

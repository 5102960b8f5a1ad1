// ref_frenet.cpp — TEST INFRASTRUCTURE.  The reference's Frenet optimal-trajectory planner compiled from its own lines
// (/root/reference/src/frenet_optimal_trajectory.cpp:20-38 defines, :40-176 sum_of_power … frenet_optimal_planning; the headers
// cubic_spline.h, frenet_path.h, quintic_polynomial.h, quartic_polynomial.h are included from the reference as they are).
// colPivHouseholderQr() is Eigen's when the host has Eigen, otherwise the stand-in's float restatement (oracle/eigen_qr.h).
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <vector>
#include <Eigen/Eigen>
#include "cubic_spline.h"
#include "frenet_path.h"
#include "quintic_polynomial.h"
#include "quartic_polynomial.h"
#include "frenet_defs.inc"

namespace ref_frenet {
#include "frenet_fns.inc"
}

extern "C" {

// Spline2D(wx, wy) -> the coefficient table the oracle and the kernel use: rows s, ax,bx,cx,dx, ay,by,cy,dy (b, d: nx-1 entries)
void ref_frenet_spline_build(const float* wx, const float* wy, int nx, float* coef) {
  using namespace cpprobotics;
  Spline2D sp(Vec_f(wx, wx + nx), Vec_f(wy, wy + nx));
  std::memset(coef, 0, sizeof(float) * 9 * nx);
  for (int i = 0; i < nx; ++i) {
    coef[i] = sp.s[i]; coef[1 * nx + i] = sp.sx.a[i]; coef[3 * nx + i] = sp.sx.c[i]; coef[5 * nx + i] = sp.sy.a[i]; coef[7 * nx + i] = sp.sy.c[i];
    if (i < nx - 1) { coef[2 * nx + i] = sp.sx.b[i]; coef[4 * nx + i] = sp.sx.d[i]; coef[6 * nx + i] = sp.sy.b[i]; coef[8 * nx + i] = sp.sy.d[i]; }
  }
}

// the coefficients the polynomial classes solve for: quintic (a3, a4, a5), quartic (a3, a4)
void ref_quintic(int n, const float* args7, float* a345) {
  for (int k = 0; k < n; ++k) {
    const float* q = args7 + 7 * k;
    cpprobotics::QuinticPolynomial p(q[0], q[1], q[2], q[3], q[4], q[5], q[6]);
    a345[3 * k] = p.a3; a345[3 * k + 1] = p.a4; a345[3 * k + 2] = p.a5;
  }
}
void ref_quartic(int n, const float* args6, float* a34) {
  for (int k = 0; k < n; ++k) {
    const float* q = args6 + 6 * k;
    cpprobotics::QuarticPolynomial p(q[0], q[1], q[2], q[3], q[4], q[5]);
    a34[2 * k] = p.a3; a34[2 * k + 1] = p.a4;
  }
}

// main() :224-236 for agents whose state (s0, c_speed, c_d, c_d_d, c_d_dd) the caller supplies, on the course Spline2D(wx, wy)
// with the obstacle list ob: plan, hand sample [1] of the winner over, stop within 1 m of the goal.  Out per tick
// (hist [max_ticks][n][8]): (s0, c_speed, c_d, c_d_d, c_d_dd, x, y, cf).  status bit 0: no candidate survived (the reference
// would index an empty path).  path_cf / path_ok [n][cap] (may be NULL): every candidate of the LAST planning call.
void ref_frenet_run(int n, int max_ticks, float* state, const float* wx, const float* wy, int nx, const float* goal, const float* ob, int nob,
                    float* hist, int* ticks_done, int* status, float* path_cf, int* path_ok, int* n_paths, int cap) {
  using namespace cpprobotics;
  using namespace ref_frenet;
  Spline2D csp_obj(Vec_f(wx, wx + nx), Vec_f(wy, wy + nx));
  Vec_Poi obstcles;
  for (int k = 0; k < nob; ++k) obstcles.push_back({{ob[2 * k], ob[2 * k + 1]}});
  for (int a = 0; a < n; ++a) {
    float s0 = state[5 * a], c_speed = state[5 * a + 1], c_d = state[5 * a + 2], c_d_d = state[5 * a + 3], c_d_dd = state[5 * a + 4];
    int st = 0, ticks = 0;
    for (int i = 0; i < max_ticks; ++i) {
      if (path_cf || path_ok || n_paths) {                      // the same three calls frenet_optimal_planning makes (:163-165)
        Vec_Path fp_list = calc_frenet_paths(c_speed, c_d, c_d_d, c_d_dd, s0);
        calc_global_paths(fp_list, csp_obj);
        if (n_paths) n_paths[a] = (int)fp_list.size();
        for (int p = 0; p < (int)fp_list.size() && p < cap; ++p) {
          if (path_cf) path_cf[(size_t)a * cap + p] = fp_list[p].cf;
          if (path_ok) {
            Vec_Path one(1, fp_list[p]);
            path_ok[(size_t)a * cap + p] = check_paths(one, obstcles).size() == 1 ? 1 : 0;
          }
        }
      }
      FrenetPath final_path = frenet_optimal_planning(csp_obj, s0, c_speed, c_d, c_d_d, c_d_dd, obstcles);
      if (final_path.s.size() < 2) { st |= 1; break; }          // nothing survived check_paths: `final_path.s[1]` would be out of range
      s0 = final_path.s[1];
      c_d = final_path.d[1];
      c_d_d = final_path.d_d[1];
      c_d_dd = final_path.d_dd[1];
      c_speed = final_path.s_d[1];
      ticks = i + 1;
      if (hist) {
        float* h = hist + ((size_t)i * n + a) * 8;
        h[0] = s0; h[1] = c_speed; h[2] = c_d; h[3] = c_d_d; h[4] = c_d_dd; h[5] = final_path.x[1]; h[6] = final_path.y[1]; h[7] = final_path.cf;
      }
      if (std::pow((final_path.x[1] - goal[0]), 2) + std::pow((final_path.y[1] - goal[1]), 2) <= 1.0) break;
    }
    state[5 * a] = s0; state[5 * a + 1] = c_speed; state[5 * a + 2] = c_d; state[5 * a + 3] = c_d_d; state[5 * a + 4] = c_d_dd;
    ticks_done[a] = ticks; status[a] = st;
  }
}

}  // extern "C"

// ref_lqr5.cpp — TEST INFRASTRUCTURE.  The reference's 5-state LQR path compiled from its own lines
// (/root/reference/src/lqr_speed_steer_control.cpp:20-30, :65-164 and the closed loop :167-171, :185-186, :195-205).
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <vector>
#include <Eigen/Eigen>
#include "cubic_spline.h"
#include "motion_model.h"
#include "cpprobotics_types.h"
#include "lqr5_defs.inc"

namespace ref_lqr5 {
#include "lqr5_fns.inc"
}

extern "C" {

// A n x25, B n x10, Q n x25, R n x4 (column-major) -> X n x25, K n x10 (2x5 column-major)
void ref_dare5(int n, const float* A, const float* B, const float* Q, const float* R, float* X, float* K) {
  for (int a = 0; a < n; ++a) {
    Matrix5f Am, Qm; Matrix52f Bm; Eigen::Matrix2f Rm;
    std::memcpy(Am.data(), A + 25 * a, 100); std::memcpy(Qm.data(), Q + 25 * a, 100);
    std::memcpy(Bm.data(), B + 10 * a, 40); std::memcpy(Rm.data(), R + 4 * a, 16);
    if (X) { Matrix5f Xm = ref_lqr5::solve_DARE(Am, Bm, Qm, Rm); std::memcpy(X + 25 * a, Xm.data(), 100); }
    if (K) { Matrix25f Km = ref_lqr5::dlqr(Am, Bm, Qm, Rm); std::memcpy(K + 10 * a, Km.data(), 40); }
  }
}

// one lqr_steering_control per agent on a shared course; control [n][2] = {ai, delta}; pe, pth_e in/out
void ref_lqr5_steering_control(int n, const float* state, int nc, const float* cx, const float* cy, const float* cyaw, const float* ck,
                               const float* sp, float* pe, float* pth_e, float* control) {
  Vec_f vx(cx, cx + nc), vy(cy, cy + nc), vyaw(cyaw, cyaw + nc), vk(ck, ck + nc), vsp(sp, sp + nc);
  for (int a = 0; a < n; ++a) {
    State st(state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]);
    Vec_f c = ref_lqr5::lqr_steering_control(st, vx, vy, vyaw, vk, vsp, pe[a], pth_e[a]);
    control[2 * a] = c[0]; control[2 * a + 1] = c[1];
  }
}

void ref_lqr5_nearest_index(int n, const float* state, int nc, const float* cx, const float* cy, const float* cyaw, int* ind, float* e) {
  Vec_f vx(cx, cx + nc), vy(cy, cy + nc), vyaw(cyaw, cyaw + nc);
  for (int a = 0; a < n; ++a) {
    State st(state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]);
    e[a] = ref_lqr5::calc_nearest_index(st, vx, vy, vyaw, ind[a]);
  }
}

void ref_lqr5_update(int n, float* state, const float* acc, const float* delta) {
  for (int a = 0; a < n; ++a) {
    State st(state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]);
    ref_lqr5::update(st, acc[a], delta[a]);
    state[4 * a] = st.x; state[4 * a + 1] = st.y; state[4 * a + 2] = st.yaw; state[4 * a + 3] = st.v;
  }
}

// closed_loop_prediction :166-205 for agents that start at state0[a] instead of the origin; at most max_ticks passes of the loop
// (the reference's `time_` never advances, so its loop ends at the goal only).  traj [max_ticks][n][4] may be NULL.
void ref_lqr5_closed_loop(int n, int max_ticks, float* state0, int nc, const float* cx_, const float* cy_, const float* cyaw_, const float* ck_,
                          const float* sp_, float goal_x, float goal_y, float* traj, int* ticks_done) {
  using namespace ref_lqr5;
  Vec_f cx(cx_, cx_ + nc), cy(cy_, cy_ + nc), cyaw(cyaw_, cyaw_ + nc), ck(ck_, ck_ + nc), speed_profile(sp_, sp_ + nc);
  Poi_f goal{{goal_x, goal_y}};
  for (int a = 0; a < n; ++a) {
#include "lqr5_loop_setup.inc"
    state = State(state0[4 * a], state0[4 * a + 1], state0[4 * a + 2], state0[4 * a + 3]);
#include "lqr5_loop_e.inc"
    int ticks = 0;
    std::streambuf* keep = std::cout.rdbuf(nullptr);           // the loop prints "Goal"
    for (int tick = 0; tick < max_ticks; ++tick) {
      ticks = tick + 1;
      bool reached = true;
      do {
#include "lqr5_loop_body.inc"
        reached = false;
      } while (0);
      if (traj) { float* h = traj + ((size_t)tick * n + a) * 4; h[0] = state.x; h[1] = state.y; h[2] = state.yaw; h[3] = state.v; }
      if (reached) break;
    }
    std::cout.rdbuf(keep);
    ticks_done[a] = ticks;
    state0[4 * a] = state.x; state0[4 * a + 1] = state.y; state0[4 * a + 2] = state.yaw; state0[4 * a + 3] = state.v;
  }
}

// main() :252-265: the course sampled from the way-points.  Returns the sample count; arrays of capacity cap.
int ref_lqr5_main_course(const float* wx_, const float* wy_, int nx, float* cx, float* cy, float* cyaw, float* ck, int cap) {
  Vec_f wx(wx_, wx_ + nx), wy(wy_, wy_ + nx);
#include "lqr5_main_course.inc"
  const int k = (int)r_x.size();
  for (int i = 0; i < k && i < cap; ++i) { cx[i] = r_x[i]; cy[i] = r_y[i]; cyaw[i] = ryaw[i]; ck[i] = rcurvature[i]; }
  return k;
}

}  // extern "C"

// ref_lqr4.cpp — TEST INFRASTRUCTURE.  The reference's 4-state LQR path compiled from its own lines
// (/root/reference/src/lqr_steer_control.cpp:20-23, :55-146 and the closed loop :149-153, :167-169, :187-198).
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <vector>
#include <Eigen/Eigen>
#include "cubic_spline.h"
#include "motion_model.h"
#include "cpprobotics_types.h"
#include "lqr4_defs.inc"
using namespace cpprobotics;

namespace ref_lqr4 {
#include "lqr4_fns.inc"
}

extern "C" {

// A n x16, B n x4, Q n x16, R n -> X n x16, K n x4
void ref_dare4(int n, const float* A, const float* B, const float* Q, const float* R, float* X, float* K) {
  for (int a = 0; a < n; ++a) {
    Eigen::Matrix4f Am, Qm; Eigen::Vector4f Bm;
    std::memcpy(Am.data(), A + 16 * a, 64); std::memcpy(Qm.data(), Q + 16 * a, 64); std::memcpy(Bm.data(), B + 4 * a, 16);
    if (X) { Eigen::Matrix4f Xm = ref_lqr4::solve_DARE(Am, Bm, Qm, R[a]); std::memcpy(X + 16 * a, Xm.data(), 64); }
    if (K) { Eigen::RowVector4f Km = ref_lqr4::dlqr(Am, Bm, Qm, R[a]); std::memcpy(K + 4 * a, Km.data(), 16); }
  }
}

// delta [n]; ind, pe, pth_e in/out
void ref_lqr4_steering_control(int n, const float* state, int nc, const float* cx, const float* cy, const float* cyaw, const float* ck,
                               int* ind, float* pe, float* pth_e, float* delta) {
  Vec_f vx(cx, cx + nc), vy(cy, cy + nc), vyaw(cyaw, cyaw + nc), vk(ck, ck + nc);
  for (int a = 0; a < n; ++a) {
    State st(state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]);
    delta[a] = ref_lqr4::lqr_steering_control(st, vx, vy, vyaw, vk, ind[a], pe[a], pth_e[a]);
  }
}

void ref_lqr4_closed_loop(int n, int max_ticks, float* state0, int nc, const float* cx_, const float* cy_, const float* cyaw_, const float* ck_,
                          const float* sp_, float goal_x, float goal_y, float* traj, int* ticks_done) {
  using namespace ref_lqr4;
  Vec_f cx(cx_, cx_ + nc), cy(cy_, cy_ + nc), cyaw(cyaw_, cyaw_ + nc), ck(ck_, ck_ + nc), speed_profile(sp_, sp_ + nc);
  Poi_f goal{{goal_x, goal_y}};
  for (int a = 0; a < n; ++a) {
#include "lqr4_loop_setup.inc"
    state = State(state0[4 * a], state0[4 * a + 1], state0[4 * a + 2], state0[4 * a + 3]);
#include "lqr4_loop_e.inc"
    int ticks = 0;
    std::streambuf* keep = std::cout.rdbuf(nullptr);
    for (int tick = 0; tick < max_ticks; ++tick) {
      ticks = tick + 1;
      bool reached = true;
      do {
#include "lqr4_loop_body.inc"
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

}  // extern "C"

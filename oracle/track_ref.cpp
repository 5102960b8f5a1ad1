// track_ref.cpp — TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline).
// Nothing under cpprobotics_amd/ may include, link or call this file.
//
// CPU restatement of the reference's course-tracking front-end and vehicle update, statement by statement:
//   calc_nearest_index      /root/reference/src/lqr_speed_steer_control.cpp:65-83 (same text: src/lqr_steer_control.cpp:55-73)
//   lqr_steering_control    src/lqr_speed_steer_control.cpp:108-151 (dim 5) ; src/lqr_steer_control.cpp:98-133 (dim 4)
//   update                  src/lqr_speed_steer_control.cpp:154-164 ; src/model_predictive_control.cpp:69-81
//   closed_loop_prediction  src/lqr_speed_steer_control.cpp:166-205 ; src/lqr_steer_control.cpp:146-197 (math only)
//   calc_nearest_index      src/model_predictive_control.cpp:107-127 ; calc_ref_trajectory :130-170
//   mpc_simulation          src/model_predictive_control.cpp:348-385 (math only; mpc_solve = oracle/mpc_ref.cpp)
// with the host libm for std::atan2/std::tan/std::cos/std::sin/std::fmod/std::sqrt/std::round — exactly what
// the reference links.  The Riccati solve is oracle_dare() of lqr_ref.cpp (dense, Eigen order).
// PINNED against the reference's own lines (oracle/ref_build.sh compiles them unmodified — against the host's Eigen, or against the
// Eigen stand-in oracle/ref_shim/Eigen/Eigen where there is none — and tests/test_oracle_vs_ref.py demands equal bits); unpinned only
// with respect to Eigen's own binary, absent from every host of this project.
// Memory safety: where the reference indexes the course without a bounds check (:110 of the MPC file, `ind += 1`
// in the 4-state loop) the index is clipped to the course, as the engine does.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>
#define CRX_TRIG_FMA 1
#include "../cpprobotics_amd/csrc/crx_trig.h"  // only for trig mode 1 (hosts whose libm sinf/cosf is not the FMA flavour)

extern "C" {
int oracle_dare(int n, int dim, const float* A, const float* B, const float* Q, const float* R, float eps, int maxiter,
                float* X, float* K, int* iters, int sum_order, int a0, int a1);
void oracle_lqr_build(int n, int dim, const float* v, double DT, double L, float* A, float* B, float* Q, float* R);
int oracle_mpc_solve(int n, int T, const float* x0, const float* xref, const double* prm, int max_iter, float* sol,
                     int* status, double* cost, int a0, int a1);
}

namespace {

int g_trig_mode = 0;   // 0: host libm cosf/sinf (the reference's behaviour); 1: explicit glibc-FMA-flavour restatement
inline float o_cos(float x) { return g_trig_mode == 0 ? std::cos(x) : crx::cosf_(x); }
inline float o_sin(float x) { return g_trig_mode == 0 ? std::sin(x) : crx::sinf_(x); }

struct State { float x, y, yaw, v; };
struct Course { const float *cx, *cy, *cyaw, *ck, *sp; int n; };

// #define YAW_P2P(angle) std::fmod(std::fmod((angle)+M_PI, 2*M_PI)-2*M_PI, 2*M_PI)+M_PI     include/motion_model.h:18
inline double YAW_P2P(float angle) { return std::fmod(std::fmod((angle) + M_PI, 2 * M_PI) - 2 * M_PI, 2 * M_PI) + M_PI; }

inline int clip(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// :65-83
float calc_nearest_index(State state, const Course& c, int& ind) {
  float mind = FLT_MAX;
  for (int i = 0; i < c.n; i++) {
    float idx = c.cx[i] - state.x;
    float idy = c.cy[i] - state.y;
    float d_e = idx * idx + idy * idy;
    if (d_e < mind) { mind = d_e; ind = i; }
  }
  const int j = clip(ind, c.n);
  float dxl = c.cx[j] - state.x;
  float dyl = c.cy[j] - state.y;
  float angle = YAW_P2P(c.cyaw[j] - std::atan2(dyl, dxl));
  if (angle < 0) mind = mind * -1;
  return mind;
}

void dlqr_from_v(int dim, float v, double DT, double L, float eps, int maxiter, float* K) {
  float A[25], B[10], Q[25], R[4], X[25];
  int it;
  oracle_lqr_build(1, dim, &v, DT, L, A, B, Q, R);
  oracle_dare(1, dim, A, B, Q, R, eps, maxiter, X, K, &it, 0, 0, 1);
}

// dim 5: :108-151, returns {ai, delta}.  dim 4: src/lqr_steer_control.cpp:98-133, returns delta in out[1], out[0] = 0.
void lqr_steering_control(int dim, State state, const Course& c, int& ind, float& pe, float& pth_e, double DT, double L,
                          float eps, int maxiter, float out[2]) {
  if (dim == 5) ind = 0;                                             // :109
  float e = calc_nearest_index(state, c, ind);                        // :110
  const int j = clip(ind, c.n);
  float k = c.ck[j];                                                  // :112
  float th_e = YAW_P2P(state.yaw - c.cyaw[j]);                        // :113
  float K[10];
  dlqr_from_v(dim, state.v, DT, L, eps, maxiter, K);                  // :116-132
  float x[5];
  x[0] = e;                                                           // :135
  x[1] = (e - pe) / DT;                                               // :136
  x[2] = th_e;                                                        // :137
  x[3] = (th_e - pth_e) / DT;                                         // :138
  float u0, u1 = 0.0f;
  if (dim == 5) {
    float tv = c.sp[j];                                               // :114
    x[4] = state.v - tv;                                              // :139
    // ustar = -K * x  (:141): 2x5 times 5x1, coefficient-based, 5-term unrolled redux (eigen_order.h C2)
    float t[5], s[5];
    for (int q = 0; q < 5; ++q) { t[q] = (-K[0 + 2 * q]) * x[q]; s[q] = (-K[1 + 2 * q]) * x[q]; }
    u0 = (t[0] + t[1]) + (t[2] + (t[3] + t[4]));
    u1 = (s[0] + s[1]) + (s[2] + (s[3] + s[4]));
  } else {
    // (-K * x)(0) (:122 of the 4-state file): row vector times vector, one SSE packet (eigen_order.h C1)
    float t0 = (-K[0]) * x[0], t1 = (-K[1]) * x[1], t2 = (-K[2]) * x[2], t3 = (-K[3]) * x[3];
    u0 = (t0 + t2) + (t1 + t3);
  }
  float ff = std::atan2((L * k), (double)1.0);                        // :143
  float fb = YAW_P2P(u0);                                             // :144
  float delta = ff + fb;                                              // :145
  pe = e;                                                             // :148
  pth_e = th_e;                                                       // :149
  out[0] = u1; out[1] = delta;
}

// update(): LQR :154-164 (no speed clamp); MPC :69-81
void update(State& state, float a, float delta, double DT, double WB, double MAX_STEER, bool clamp, double MAX_SPEED, double MIN_SPEED) {
  if (delta >= MAX_STEER) delta = MAX_STEER;
  if (delta <= -MAX_STEER) delta = -MAX_STEER;
  float nx = state.x + state.v * o_cos(state.yaw) * DT;
  float ny = state.y + state.v * o_sin(state.yaw) * DT;
  float nyaw = state.yaw + state.v / WB * std::tan(delta) * DT;
  float nv = state.v + a * DT;
  state.x = nx; state.y = ny; state.yaw = nyaw; state.v = nv;
  if (clamp) {
    if (state.v > MAX_SPEED) state.v = MAX_SPEED;
    if (state.v < MIN_SPEED) state.v = MIN_SPEED;
  }
}

// src/model_predictive_control.cpp:107-127
int calc_nearest_index_window(State state, const Course& c, int pind, int nsearch) {
  float mind = FLT_MAX;
  float ind = 0;
  const int lo = pind < 0 ? 0 : pind;
  const long long hi_ll = (long long)pind + nsearch;
  const int hi = hi_ll > c.n ? c.n : (int)hi_ll;
  for (int i = lo; i < hi; i++) {
    float idx = c.cx[i] - state.x;
    float idy = c.cy[i] - state.y;
    float d_e = idx * idx + idy * idy;
    if (d_e < mind) { mind = d_e; ind = i; }
  }
  return ind;
}

// :130-170
void calc_ref_trajectory(State state, const Course& c, float dl, double DT, int nsearch, int T, int& target_ind, float* xref) {
  int ncourse = c.n;
  int ind = calc_nearest_index_window(state, c, target_ind, nsearch);
  if (target_ind >= ind) ind = target_ind;
  float travel = 0.0;
  for (int i = 0; i < T; i++) {
    travel += std::abs(state.v) * DT;
    int dind = (int)std::round(travel / dl);
    long long jj = (long long)ind + dind;
    int j = (jj < ncourse) ? (int)jj : ncourse - 1;
    xref[4 * i + 0] = c.cx[j]; xref[4 * i + 1] = c.cy[j]; xref[4 * i + 2] = c.cyaw[j]; xref[4 * i + 3] = c.sp[j];
  }
  target_ind = ind;
}

}  // namespace

extern "C" {

void oracle_track_set_trig_mode(int m) { g_trig_mode = m; }

void oracle_calc_nearest_index(int n, const float* state, int ncourse, const float* cx, const float* cy, const float* cyaw,
                               int* ind_io, float* e_out) {
  Course c{cx, cy, cyaw, nullptr, nullptr, ncourse};
  for (int a = 0; a < n; ++a) {
    State s{state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]};
    int ind = ind_io[a];
    float e = calc_nearest_index(s, c, ind);
    ind_io[a] = ind;
    if (e_out) e_out[a] = e;
  }
}

void oracle_lqr_steering_control(int n, int dim, const float* state, int ncourse, const float* cx, const float* cy,
                                 const float* cyaw, const float* ck, const float* sp, int* ind_io, float* pe, float* pth_e,
                                 double DT, double L, float eps, int maxiter, float* control, int a0, int a1) {
  Course c{cx, cy, cyaw, ck, sp, ncourse};
  for (int a = a0; a < a1; ++a) {
    State s{state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]};
    int ind = ind_io ? ind_io[a] : 0;
    float out[2];
    lqr_steering_control(dim, s, c, ind, pe[a], pth_e[a], DT, L, eps, maxiter, out);
    if (ind_io) ind_io[a] = ind;
    if (dim == 5) { control[2 * a] = out[0]; control[2 * a + 1] = out[1]; }
    else control[a] = out[1];
  }
}

void oracle_update(int n, float* state, const float* a_in, const float* delta_in, double DT, double WB, double MAX_STEER,
                   int clamp, double MAX_SPEED, double MIN_SPEED) {
  for (int a = 0; a < n; ++a) {
    State s{state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]};
    update(s, a_in[a], delta_in[a], DT, WB, MAX_STEER, clamp != 0, MAX_SPEED, MIN_SPEED);
    state[4 * a] = s.x; state[4 * a + 1] = s.y; state[4 * a + 2] = s.yaw; state[4 * a + 3] = s.v;
  }
}

// closed_loop_prediction: the loop body of :194-205 (dim 5) / :186-196 (dim 4), bounded by max_ticks.
void oracle_lqr_closed_loop(int n, int dim, int max_ticks, float* state, int ncourse, const float* cx, const float* cy,
                            const float* cyaw, const float* ck, const float* sp, float* pe, float* pth_e, int* ind_io,
                            double DT, double L, float eps, int maxiter, double MAX_STEER, float goal_x, float goal_y,
                            float goal_dis, double KP, float stop_speed, float* traj_hist, int* ticks_done, int a0, int a1) {
  Course c{cx, cy, cyaw, ck, sp, ncourse};
  for (int a = a0; a < a1; ++a) {
    State s{state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]};
    float e = pe ? pe[a] : 0.0f, e_th = pth_e ? pth_e[a] : 0.0f;
    int ind = ind_io ? ind_io[a] : 0;
    int ticks = 0;
    for (int t = 0; t < max_ticks; ++t) {
      float out[2];
      lqr_steering_control(dim, s, c, ind, e, e_th, DT, L, eps, maxiter, out);
      float ai = out[0], di = out[1];
      if (dim == 4) ai = KP * (c.sp[clip(ind, c.n)] - s.v);                      // :188
      update(s, ai, di, DT, L, MAX_STEER, false, 0, 0);
      if (dim == 4 && std::abs(s.v) <= stop_speed) ind += 1;                     // :191
      ticks = t + 1;
      if (traj_hist) { float* h = traj_hist + ((size_t)t * n + a) * 4; h[0] = s.x; h[1] = s.y; h[2] = s.yaw; h[3] = s.v; }
      float dx = s.x - goal_x;
      float dy = s.y - goal_y;
      if (std::sqrt(dx * dx + dy * dy) <= goal_dis) break;
    }
    state[4 * a] = s.x; state[4 * a + 1] = s.y; state[4 * a + 2] = s.yaw; state[4 * a + 3] = s.v;
    if (pe) pe[a] = e;
    if (pth_e) pth_e[a] = e_th;
    if (ind_io) ind_io[a] = ind;
    if (ticks_done) ticks_done[a] = ticks;
  }
}

void oracle_calc_nearest_index_window(int n, const float* state, int ncourse, const float* cx, const float* cy,
                                      const int* pind, int nsearch, int* ind_out) {
  Course c{cx, cy, nullptr, nullptr, nullptr, ncourse};
  for (int a = 0; a < n; ++a) {
    State s{state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]};
    ind_out[a] = calc_nearest_index_window(s, c, pind[a], nsearch);
  }
}

void oracle_calc_ref_trajectory(int n, int T, const float* state, int ncourse, const float* cx, const float* cy,
                                const float* cyaw, const float* ck, const float* sp, float dl, double DT, int nsearch,
                                int* target_ind, float* xref) {
  Course c{cx, cy, cyaw, ck, sp, ncourse};
  for (int a = 0; a < n; ++a) {
    State s{state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]};
    calc_ref_trajectory(s, c, dl, DT, nsearch, T, target_ind[a], xref + (size_t)a * 4 * T);
  }
}

// mpc_simulation's loop (:371-385) with the oracle's MPC twin as mpc_solve.
void oracle_mpc_closed_loop(int n, int T, int max_ticks, float* state, int ncourse, const float* cx, const float* cy,
                            const float* cyaw, const float* ck, const float* sp, float dl, int nsearch, const double* mpc_prm,
                            int mpc_max_iter, float goal_x, float goal_y, float goal_dis, int* target_ind, float* traj_hist,
                            int* ticks_done, int a0, int a1) {
  Course c{cx, cy, cyaw, ck, sp, ncourse};
  const double DT = mpc_prm[0], WB = mpc_prm[1], MAX_STEER = mpc_prm[2], MAX_SPEED = mpc_prm[4], MIN_SPEED = mpc_prm[5];
  const int nv = 4 * T + 2 * (T - 1);
  std::vector<float> xref(4 * T), sol(nv);
  for (int a = a0; a < a1; ++a) {
    State s{state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]};
    int tind = target_ind[a];
    int ticks = 0;
    for (int t = 0; t < max_ticks; ++t) {
      calc_ref_trajectory(s, c, dl, DT, nsearch, T, tind, xref.data());
      float x0[4] = {s.x, s.y, s.yaw, s.v};
      int st; double cost;
      oracle_mpc_solve(1, T, x0, xref.data(), mpc_prm, mpc_max_iter, sol.data(), &st, &cost, 0, 1);
      update(s, sol[4 * T + (T - 1)], sol[4 * T], DT, WB, MAX_STEER, true, MAX_SPEED, MIN_SPEED);
      ticks = t + 1;
      if (traj_hist) { float* h = traj_hist + ((size_t)t * n + a) * 4; h[0] = s.x; h[1] = s.y; h[2] = s.yaw; h[3] = s.v; }
      float dx = s.x - goal_x;
      float dy = s.y - goal_y;
      if (std::sqrt(dx * dx + dy * dy) <= goal_dis) break;
    }
    state[4 * a] = s.x; state[4 * a + 1] = s.y; state[4 * a + 2] = s.yaw; state[4 * a + 3] = s.v;
    target_ind[a] = tind;
    if (ticks_done) ticks_done[a] = ticks;
  }
}

}  // extern "C"

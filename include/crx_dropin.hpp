// crx_dropin.hpp — the reference's own C++ call signatures on top of the crx C ABI (include/crx.h).
//
// The reference (onlytailei/CppRobotics) defines its hot-path functions as free functions in the
// same translation unit as main().  To switch a reference executable to the MI355X engine, delete
// those function bodies and include this header instead (INTEGRATION.md shows the exact lines):
//
//   src/extended_kalman_filter.cpp:22-78   motion_model, jacobF, observation_model, jacobH, ekf_estimation
//   src/lqr_speed_steer_control.cpp:85-106 solve_DARE (5x5), dlqr (5x5)
//   src/lqr_steer_control.cpp:75-96        solve_DARE (4x4), dlqr (4x4)
//   src/model_predictive_control.cpp:255-346 mpc_solve
//   and the callers on either side of the solves — calc_nearest_index (both forms), lqr_steering_control, update,
//   calc_ref_trajectory, closed_loop_prediction, mpc_simulation — in one namespace per reference translation unit (the three files reuse the names
//   `update` / `calc_nearest_index` with different bodies): crx_dropin::lqr_speed_steer, ::lqr_steer, ::mpc.
//   `using namespace crx_dropin::lqr_speed_steer;` after deleting src/lqr_speed_steer_control.cpp:65-164 etc.
//
// Every function forwards `.data()` pointers (Eigen fixed-size matrices are column-major and
// contiguous, exactly the ABI's layout) with n = 1 through the host-pointer entry points, which copy
// to the GPU, launch, and copy back — a faithful but slow single-agent path.  The fast path is the
// batched one: crx_dropin::Batch* helpers below, or the C ABI directly.
//
// With <Eigen/Eigen> present the signatures are the reference's verbatim (Eigen::Vector4f ...).
// Without Eigen (this build image has none) the same names bind to crx::Mat<R,C>, a minimal
// column-major stand-in with the members the reference code uses (operator(), data(), <<-free).
#pragma once
#include <array>
#include <cmath>
#include <cstddef>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "crx.h"

#if defined(CRX_DROPIN_USE_EIGEN) || (!defined(CRX_DROPIN_NO_EIGEN) && defined(__has_include))
#if defined(CRX_DROPIN_USE_EIGEN) || __has_include(<Eigen/Eigen>)
#include <Eigen/Eigen>
#define CRX_DROPIN_HAVE_EIGEN 1
#endif
#endif

namespace crx {

#ifdef CRX_DROPIN_HAVE_EIGEN
template <int R, int C> using Mat = Eigen::Matrix<float, R, C>;
#else
// Column-major fixed-size float matrix: M(i,j) = d[i + R*j]  (Eigen's default storage order).
template <int R, int C>
struct Mat {
  float d[R * C];
  Mat() : d{} {}
  float& operator()(int i, int j) { return d[i + R * j]; }
  float operator()(int i, int j) const { return d[i + R * j]; }
  float& operator()(int i) { return d[i]; }          // vectors
  float operator()(int i) const { return d[i]; }
  float* data() { return d; }
  const float* data() const { return d; }
  static Mat Zero() { return Mat(); }
  static Mat Identity() { Mat m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0f; return m; }
};
#endif

inline void dropin_check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + crx_last_error());
}

}  // namespace crx

// Types the reference's sources name.  (Its MPC file also does `#define T 6`; the horizon is a
// template parameter here.)
namespace cpprobotics {
#ifndef _CPPROBOTICS_TYPES_H
using Vec_f = std::vector<float>;
using Poi_f = std::array<float, 2>;
#endif
#ifndef _MOTION_MODEL_H
struct State {          // include/motion_model.h:31-42
  float x, y, yaw, v;
  State(float x_, float y_, float yaw_, float v_) : x(x_), y(y_), yaw(yaw_), v(v_) {}
};
#endif
}  // namespace cpprobotics

#ifndef CRX_DROPIN_NO_GLOBAL_NAMES
// ---- src/extended_kalman_filter.cpp -----------------------------------------------------------------
inline crx::Mat<4, 1> motion_model(crx::Mat<4, 1> x, crx::Mat<2, 1> u) {                        // :22
  crx::Mat<4, 1> out;
  crx::dropin_check(crx_motion_model_batch(1, x.data(), u.data(), out.data(), nullptr), "motion_model");
  return out;
}
inline crx::Mat<4, 4> jacobF(crx::Mat<4, 1> x, crx::Mat<2, 1> u) {                              // :38
  crx::Mat<4, 4> jF;
  crx::dropin_check(crx_jacobF_batch(1, x.data(), u.data(), jF.data(), nullptr), "jacobF");
  return jF;
}
inline crx::Mat<2, 1> observation_model(crx::Mat<4, 1> x) {                                     // :50
  crx::Mat<2, 1> z;
  crx::dropin_check(crx_observation_model_batch(1, x.data(), z.data()), "observation_model");
  return z;
}
inline crx::Mat<2, 4> jacobH() {                                                                // :57
  crx::Mat<2, 4> h;
  crx::dropin_check(crx_jacobH(h.data()), "jacobH");
  return h;
}
inline void ekf_estimation(crx::Mat<4, 1>& xEst, crx::Mat<4, 4>& PEst, crx::Mat<2, 1> z, crx::Mat<2, 1> u,
                           crx::Mat<4, 4> Q, crx::Mat<2, 2> R) {                                // :64-66
  crx::dropin_check(crx_ekf_step_batch(1, xEst.data(), PEst.data(), z.data(), u.data(), Q.data(), R.data(), nullptr),
                    "ekf_estimation");
}

// ---- src/lqr_speed_steer_control.cpp (5x5, two inputs) ----------------------------------------------
inline crx::Mat<5, 5> solve_DARE(crx::Mat<5, 5> A, crx::Mat<5, 2> B, crx::Mat<5, 5> Q, crx::Mat<2, 2> R) {   // :85
  crx::Mat<5, 5> X;
  crx::dropin_check(crx_dare_batch(1, 5, A.data(), B.data(), Q.data(), R.data(), 0.01f, 150, X.data(), nullptr, nullptr),
                    "solve_DARE");
  return X;
}
inline crx::Mat<2, 5> dlqr(crx::Mat<5, 5> A, crx::Mat<5, 2> B, crx::Mat<5, 5> Q, crx::Mat<2, 2> R) {         // :102
  crx::Mat<2, 5> K;
  crx::dropin_check(crx_dare_batch(1, 5, A.data(), B.data(), Q.data(), R.data(), 0.01f, 150, nullptr, K.data(), nullptr),
                    "dlqr");
  return K;
}
// ---- src/lqr_steer_control.cpp (4x4, one input, scalar R) -------------------------------------------
inline crx::Mat<4, 4> solve_DARE(crx::Mat<4, 4> A, crx::Mat<4, 1> B, crx::Mat<4, 4> Q, float R) {            // :75
  crx::Mat<4, 4> X;
  crx::dropin_check(crx_dare_batch(1, 4, A.data(), B.data(), Q.data(), &R, 0.01f, 150, X.data(), nullptr, nullptr),
                    "solve_DARE");
  return X;
}
inline crx::Mat<1, 4> dlqr(crx::Mat<4, 4> A, crx::Mat<4, 1> B, crx::Mat<4, 4> Q, float R) {                  // :92
  crx::Mat<1, 4> K;
  crx::dropin_check(crx_dare_batch(1, 4, A.data(), B.data(), Q.data(), &R, 0.01f, 150, nullptr, K.data(), nullptr), "dlqr");
  return K;
}

// ---- src/model_predictive_control.cpp -----------------------------------------------------------------
// mpc_solve(State x0, M_XREF traj_ref) :255, M_XREF = Matrix<float, NX, T>.  Returns all
// 4*T + 2*(T-1) variables in the reference's layout [x|y|yaw|v|delta|a] (:54-60, :341-345).
template <int T_>
inline cpprobotics::Vec_f mpc_solve(cpprobotics::State x0, crx::Mat<4, T_> traj_ref) {
  const float x0v[4] = {x0.x, x0.y, x0.yaw, x0.v};
  cpprobotics::Vec_f result(4 * T_ + 2 * (T_ - 1));
  crx::dropin_check(crx_mpc_solve_batch(1, T_, x0v, traj_ref.data(), nullptr, result.data(), nullptr, nullptr), "mpc_solve");
  return result;
}
#endif  // CRX_DROPIN_NO_GLOBAL_NAMES

// ---- batched C++ helpers: the way to actually use the engine from C++ ---------------------------------
namespace crx_dropin {

// n vehicles stored as std::vector of fixed-size matrices (contiguous: the ABI's layout).
inline void ekf_estimation_batch(std::vector<crx::Mat<4, 1>>& xEst, std::vector<crx::Mat<4, 4>>& PEst,
                                 const std::vector<crx::Mat<2, 1>>& z, const std::vector<crx::Mat<2, 1>>& u,
                                 const crx::Mat<4, 4>& Q, const crx::Mat<2, 2>& R) {
  static_assert(sizeof(crx::Mat<4, 4>) == 64 && sizeof(crx::Mat<4, 1>) == 16 && sizeof(crx::Mat<2, 1>) == 8,
                "fixed-size matrices must be densely packed");
  const int n = (int)xEst.size();
  if ((int)PEst.size() != n || (int)z.size() != n || (int)u.size() != n) throw std::invalid_argument("batch size mismatch");
  crx::dropin_check(crx_ekf_step_batch(n, xEst[0].data(), PEst[0].data(), z[0].data(), u[0].data(), Q.data(), R.data(), nullptr),
                    "ekf_estimation_batch");
}

// The loop of main() (src/extended_kalman_filter.cpp:171-188) for n vehicles and T steps in ONE call: z and ud time-major ([t][vehicle]),
// hxEst (the reference's `hxEst.push_back(xEst)`, :187) resized to T * n and filled when given.  Any allocator (see pinned_allocator).
template <class A2, class A4>
inline void ekf_estimation_run(std::vector<crx::Mat<4, 1>>& xEst, std::vector<crx::Mat<4, 4>>& PEst, const std::vector<crx::Mat<2, 1>, A2>& z,
                               const std::vector<crx::Mat<2, 1>, A2>& ud, const crx::Mat<4, 4>& Q, const crx::Mat<2, 2>& R,
                               std::vector<crx::Mat<4, 1>, A4>* hxEst = nullptr) {
  const size_t n = xEst.size();
  if (n == 0 || PEst.size() != n || z.size() % n || ud.size() != z.size()) throw std::invalid_argument("batch size mismatch");
  const size_t T = z.size() / n;
  if (hxEst) hxEst->resize(T * n);
  crx::dropin_check(crx_ekf_run_batch((int)n, (int)T, xEst[0].data(), PEst[0].data(), z[0].data(), ud[0].data(),
                                      hxEst ? (*hxEst)[0].data() : nullptr, nullptr, Q.data(), R.data(), nullptr), "ekf_estimation_run");
}

// Fleet-sized host arrays: std::vector<T, crx_dropin::pinned_allocator<T>> lives in pinned memory (crx_host_alloc), which the
// host-pointer entry points DMA in place instead of staging it through copy threads (87 against 66-74 GB/s across PCIe at the
// BASELINE EKF batch).
template <class T>
struct pinned_allocator {
  using value_type = T;
  pinned_allocator() = default;
  template <class U> pinned_allocator(const pinned_allocator<U>&) {}
  T* allocate(std::size_t n) {
    void* p = crx_host_alloc(n * sizeof(T));
    if (!p) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T* p, std::size_t) { crx_host_free(p); }
  template <class U> bool operator==(const pinned_allocator<U>&) const { return true; }
  template <class U> bool operator!=(const pinned_allocator<U>&) const { return false; }
};

// Every host-pointer batch call of this process on all visible GPUs (at least `min_per_gpu` agents each); returns the GPU count.
inline int use_all_devices(int min_per_gpu = 4096) {
  const int g = crx_device_count();
  crx::dropin_check(crx_set_devices(nullptr, g, min_per_gpu), "crx_set_devices");
  return g;
}

}  // namespace crx_dropin

// ---- the callers on either side of the solves, one namespace per reference translation unit -----------
namespace crx_dropin {
using cpprobotics::State;
using cpprobotics::Vec_f;

using cpprobotics::Poi_f;

// What the reference's closed-loop functions draw (x_h / y_h and the final state) — they return void and show an OpenCV window; a
// call site that ignores the result compiles unchanged.  One row per executed tick that did not reach the goal, as the reference
// pushes them (src/lqr_speed_steer_control.cpp:207-208, src/model_predictive_control.cpp:387-388).
struct Trajectory {
  Vec_f x_h, y_h, yaw_h, v_h;
  State final_state{0.0f, 0.0f, 0.0f, 0.0f};
  int ticks = 0;            // passes of the loop executed
  bool goal = false;        // the goal test fired (otherwise max_ticks ran out: the reference's own loops have no other exit)
};
inline crx_course course_of(const Vec_f& cx, const Vec_f& cy, const Vec_f& cyaw, const Vec_f* ck, const Vec_f* sp) {
  if (cy.size() != cx.size() || cyaw.size() != cx.size()) throw std::invalid_argument("course arrays differ in length");
  return crx_course{(int)cx.size(), cx.data(), cy.data(), cyaw.data(), ck ? ck->data() : nullptr, sp ? sp->data() : nullptr};
}

namespace lqr_speed_steer {   // src/lqr_speed_steer_control.cpp
inline float calc_nearest_index(State state, Vec_f cx, Vec_f cy, Vec_f cyaw, int& ind) {                     // :65
  const crx_course c = course_of(cx, cy, cyaw, nullptr, nullptr);
  const float s[4] = {state.x, state.y, state.yaw, state.v};
  float e = 0.0f;
  crx::dropin_check(crx_calc_nearest_index_batch(1, s, &c, &ind, &e), "calc_nearest_index");
  return e;
}
inline Vec_f lqr_steering_control(State state, Vec_f cx, Vec_f cy, Vec_f cyaw, Vec_f ck, Vec_f sp, float& pe, float& pth_e) {   // :108
  const crx_course c = course_of(cx, cy, cyaw, &ck, &sp);
  const float s[4] = {state.x, state.y, state.yaw, state.v};
  float u[2];
  crx::dropin_check(crx_lqr_steering_control_batch(1, 5, s, &c, nullptr, &pe, &pth_e, nullptr, u), "lqr_steering_control");
  return {u[0], u[1]};
}
inline Vec_f calc_speed_profile(Vec_f rx, Vec_f ry, Vec_f ryaw, float target_speed) {                            // :40
  Vec_f sp(ryaw.size());
  crx::dropin_check(crx_calc_speed_profile(5, rx.data(), ry.data(), ryaw.data(), (int)ryaw.size(), target_speed, sp.data()), "calc_speed_profile");
  return sp;
}
inline void update(State& state, float a, float delta) {                                                     // :154
  float s[4] = {state.x, state.y, state.yaw, state.v};
  crx::dropin_check(crx_update_batch(1, s, &a, &delta, nullptr), "update");
  state.x = s[0]; state.y = s[1]; state.yaw = s[2]; state.v = s[3];
}
// closed_loop_prediction(cx, cy, cyaw, ck, speed_profile, goal) :166 — the loop :194-205 for the reference's start State(-0,-0,0,0),
// e = e_th = 0, goal_dis 0.3.  The reference's clock never advances (`time_` :173), its loop ends at the goal only: max_ticks bounds it.
inline Trajectory closed_loop_prediction(Vec_f cx, Vec_f cy, Vec_f cyaw, Vec_f ck, Vec_f speed_profile, Poi_f goal, int max_ticks = 5000) {
  const crx_course c = course_of(cx, cy, cyaw, &ck, &speed_profile);
  float s[4] = {-0.0f, -0.0f, 0.0f, 0.0f};
  crx_loop_params lp{goal[0], goal[1], 0.3f, 1.0, 0.05f, max_ticks};
  std::vector<float> hist((size_t)max_ticks * 4);
  int ticks = 0;
  crx::dropin_check(crx_lqr_closed_loop_batch(1, 5, s, &c, nullptr, nullptr, nullptr, nullptr, nullptr, &lp, hist.data(), &ticks), "closed_loop_prediction");
  Trajectory t;
  t.ticks = ticks; t.final_state = State(s[0], s[1], s[2], s[3]);
  { const float dx = s[0] - goal[0], dy = s[1] - goal[1]; t.goal = std::sqrt(dx * dx + dy * dy) <= lp.goal_dis; }
  const int kept = t.goal ? ticks - 1 : ticks;                  // the tick that reaches the goal breaks before the push_back
  for (int k = 0; k < kept; ++k) { t.x_h.push_back(hist[4 * k]); t.y_h.push_back(hist[4 * k + 1]); t.yaw_h.push_back(hist[4 * k + 2]); t.v_h.push_back(hist[4 * k + 3]); }
  return t;
}
}  // namespace lqr_speed_steer

namespace lqr_steer {         // src/lqr_steer_control.cpp
using lqr_speed_steer::calc_nearest_index;   // :55-73, the same text
using lqr_speed_steer::update;               // :136-146, the same text
inline Vec_f calc_speed_profile(Vec_f rx, Vec_f ry, Vec_f ryaw, float target_speed) {                            // :35
  Vec_f sp(ryaw.size());
  crx::dropin_check(crx_calc_speed_profile(4, rx.data(), ry.data(), ryaw.data(), (int)ryaw.size(), target_speed, sp.data()), "calc_speed_profile");
  return sp;
}
inline float lqr_steering_control(State state, Vec_f cx, Vec_f cy, Vec_f cyaw, Vec_f ck, int& ind, float& pe, float& pth_e) {   // :98
  const Vec_f sp(cx.size(), 0.0f);           // the 4-state controller does not read the speed profile
  const crx_course c = course_of(cx, cy, cyaw, &ck, &sp);
  const float s[4] = {state.x, state.y, state.yaw, state.v};
  float delta = 0.0f;
  crx::dropin_check(crx_lqr_steering_control_batch(1, 4, s, &c, &ind, &pe, &pth_e, nullptr, &delta), "lqr_steering_control");
  return delta;
}
// closed_loop_prediction(cx, cy, cyaw, ck, speed_profile, goal) :148 — the loop :186-197: steering from the 4-state LQR, ai = KP *
// (speed_profile[ind] - v), ind advanced while the vehicle stands still; goal_dis 0.5.
inline Trajectory closed_loop_prediction(Vec_f cx, Vec_f cy, Vec_f cyaw, Vec_f ck, Vec_f speed_profile, Poi_f goal, int max_ticks = 5000) {
  const crx_course c = course_of(cx, cy, cyaw, &ck, &speed_profile);
  float s[4] = {-0.0f, -0.0f, 0.0f, 0.0f};
  crx_loop_params lp{goal[0], goal[1], 0.5f, 1.0, 0.05f, max_ticks};
  std::vector<float> hist((size_t)max_ticks * 4);
  int ticks = 0;
  crx::dropin_check(crx_lqr_closed_loop_batch(1, 4, s, &c, nullptr, nullptr, nullptr, nullptr, nullptr, &lp, hist.data(), &ticks), "closed_loop_prediction");
  Trajectory t;
  t.ticks = ticks; t.final_state = State(s[0], s[1], s[2], s[3]);
  { const float dx = s[0] - goal[0], dy = s[1] - goal[1]; t.goal = std::sqrt(dx * dx + dy * dy) <= lp.goal_dis; }
  const int kept = t.goal ? ticks - 1 : ticks;
  for (int k = 0; k < kept; ++k) { t.x_h.push_back(hist[4 * k]); t.y_h.push_back(hist[4 * k + 1]); t.yaw_h.push_back(hist[4 * k + 2]); t.v_h.push_back(hist[4 * k + 3]); }
  return t;
}
}  // namespace lqr_steer

namespace mpc {               // src/model_predictive_control.cpp
inline Vec_f calc_speed_profile(Vec_f rx, Vec_f ry, Vec_f ryaw, float target_speed) {                            // :83
  Vec_f sp(ryaw.size());
  crx::dropin_check(crx_calc_speed_profile(0, rx.data(), ry.data(), ryaw.data(), (int)ryaw.size(), target_speed, sp.data()), "calc_speed_profile");
  return sp;
}
inline void smooth_yaw(Vec_f& cyaw) {                                                                            // :172
  crx::dropin_check(crx_smooth_yaw(cyaw.data(), (int)cyaw.size()), "smooth_yaw");
}
inline void update(State& state, float a, float delta) {                                                     // :69
  crx_vehicle_params p;
  crx_vehicle_default_params(&p, 1);
  float s[4] = {state.x, state.y, state.yaw, state.v};
  crx::dropin_check(crx_update_batch(1, s, &a, &delta, &p), "update");
  state.x = s[0]; state.y = s[1]; state.yaw = s[2]; state.v = s[3];
}
// calc_ref_trajectory(state, cx, cy, cyaw, ck, sp, dl, target_ind, xref) :130 — xref is the reference's M_XREF.
template <int T_>
inline void calc_ref_trajectory(State state, Vec_f cx, Vec_f cy, Vec_f cyaw, Vec_f ck, Vec_f sp, float dl, int& target_ind,
                                crx::Mat<4, T_>& xref) {
  const crx_course c = course_of(cx, cy, cyaw, &ck, &sp);
  const float s[4] = {state.x, state.y, state.yaw, state.v};
  crx::dropin_check(crx_calc_ref_trajectory_batch(1, T_, s, &c, dl, 0.2, 10, &target_ind, xref.data()), "calc_ref_trajectory");
}
// calc_nearest_index(state, cx, cy, cyaw, pind) :107 — the window of N_IND_SEARCH = 10 points from pind (the reference reads
// cx[pind .. pind+9] unchecked, :110; the engine clips the window at the end of the course)
inline int calc_nearest_index(State state, Vec_f cx, Vec_f cy, Vec_f cyaw, int pind) {
  const crx_course c = course_of(cx, cy, cyaw, nullptr, nullptr);
  const float s[4] = {state.x, state.y, state.yaw, state.v};
  int ind = 0;
  crx::dropin_check(crx_calc_nearest_index_window_batch(1, s, &c, &pind, 10, &ind), "calc_nearest_index");
  return ind;
}
// mpc_simulation(cx, cy, cyaw, ck, speed_profile, goal) :348 — set-up :349-360 (start state from the first course point, the yaw
// wrap, target_ind = 0, smooth_yaw on the by-value copy of cyaw) and the loop :371-385 (calc_ref_trajectory -> mpc_solve -> update
// -> goal test, goal_dis 0.5) as ONE persistent kernel.  The reference's `iter_count` never advances (MAX_TIME is no exit): max_ticks.
template <int T_>
inline Trajectory mpc_simulation(Vec_f cx, Vec_f cy, Vec_f cyaw, Vec_f ck, Vec_f speed_profile, Poi_f goal, int max_ticks = 5000) {
  float s[4] = {cx[0], cy[0], cyaw[0], speed_profile[0]};
  const double pi = 3.14159265358979323846;
  if ((double)(s[2] - cyaw[0]) >= pi) s[2] = (float)((double)s[2] - pi * 2.0);
  else if ((double)(s[2] - cyaw[0]) <= -1.0 * pi) s[2] = (float)((double)s[2] + pi * 2.0);
  smooth_yaw(cyaw);
  const crx_course c = course_of(cx, cy, cyaw, &ck, &speed_profile);
  crx_loop_params lp{goal[0], goal[1], 0.5f, 1.0, 0.05f, max_ticks};
  std::vector<float> hist((size_t)max_ticks * 4);
  int ticks = 0, target_ind = 0;
  crx::dropin_check(crx_mpc_closed_loop_batch(1, T_, s, &c, 1.0f, 10, nullptr, &lp, &target_ind, hist.data(), &ticks, nullptr), "mpc_simulation");
  Trajectory t;
  t.ticks = ticks; t.final_state = State(s[0], s[1], s[2], s[3]);
  { const float dx = s[0] - goal[0], dy = s[1] - goal[1]; t.goal = std::sqrt(dx * dx + dy * dy) <= lp.goal_dis; }
  const int kept = t.goal ? ticks - 1 : ticks;
  for (int k = 0; k < kept; ++k) { t.x_h.push_back(hist[4 * k]); t.y_h.push_back(hist[4 * k + 1]); t.yaw_h.push_back(hist[4 * k + 2]); t.v_h.push_back(hist[4 * k + 3]); }
  return t;
}
}  // namespace mpc

}  // namespace crx_dropin

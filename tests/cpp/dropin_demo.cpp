// dropin_demo.cpp — the reference's call sites, compiled against include/crx_dropin.hpp.
// Mirrors: the EKF loop body (src/extended_kalman_filter.cpp:171-183) with fixed inputs instead of the
// random_device draws; lqr_steering_control()'s dlqr call (src/lqr_speed_steer_control.cpp:116-132,
// src/lqr_steer_control.cpp:104-118); mpc_simulation()'s mpc_solve call (src/model_predictive_control.cpp:374).
// The closed loops under the reference's own signatures — closed_loop_prediction (src/lqr_speed_steer_control.cpp:166,
// src/lqr_steer_control.cpp:148), mpc_simulation (src/model_predictive_control.cpp:348) — and the MPC file's windowed
// calc_nearest_index (:107) run on the arc course below.
// Prints every result as hex floats so the GPU test can compare bit-for-bit with the oracle.
#include <cmath>
#include <cstdio>
#include <string>
#include "crx_dropin.hpp"

using crx::Mat;

static void dump(const char* name, const float* p, int n) {
  std::printf("%s", name);
  for (int i = 0; i < n; ++i) std::printf(" %a", (double)p[i]);
  std::printf("\n");
}

int main() {
  // ---- EKF: 5 steps of the reference loop, deterministic "noise"
  Mat<2, 1> u; u(0) = 1.0f; u(1) = 0.1f;
  Mat<4, 1> xEst, xTrue;
  Mat<4, 4> PEst = Mat<4, 4>::Identity();
  Mat<4, 4> Q = Mat<4, 4>::Identity();
  Q(0, 0) = 0.1 * 0.1; Q(1, 1) = 0.1 * 0.1; Q(2, 2) = (1.0 / 180 * M_PI) * (1.0 / 180 * M_PI); Q(3, 3) = 0.1 * 0.1;
  Mat<2, 2> R = Mat<2, 2>::Identity();
  for (int t = 0; t < 5; ++t) {
    Mat<2, 1> ud; ud(0) = u(0) + 0.05f * (t - 2); ud(1) = u(1) - 0.01f * t;
    xTrue = motion_model(xTrue, u);
    Mat<2, 1> z; z(0) = xTrue(0) + 0.1f * (t % 3 - 1); z(1) = xTrue(1) - 0.07f * (t % 2);
    ekf_estimation(xEst, PEst, z, ud, Q, R);
    dump("ekf_x", xEst.data(), 4);
    dump("ekf_P", PEst.data(), 16);
  }
  {  // the same five steps for a fleet of 3,000 identical vehicles in ONE call, the fleet-sized arrays in pinned memory
    const size_t n = 3000, T = 5;
    std::vector<Mat<2, 1>, crx_dropin::pinned_allocator<Mat<2, 1>>> zf(T * n), uf(T * n);
    std::vector<Mat<4, 1>, crx_dropin::pinned_allocator<Mat<4, 1>>> hx;
    Mat<4, 1> xt;
    for (size_t t = 0; t < T; ++t) {
      Mat<2, 1> ud; ud(0) = u(0) + 0.05f * ((int)t - 2); ud(1) = u(1) - 0.01f * (int)t;
      xt = motion_model(xt, u);
      Mat<2, 1> z; z(0) = xt(0) + 0.1f * ((int)t % 3 - 1); z(1) = xt(1) - 0.07f * ((int)t % 2);
      for (size_t a = 0; a < n; ++a) { zf[t * n + a] = z; uf[t * n + a] = ud; }
    }
    std::vector<Mat<4, 1>> xf(n);
    std::vector<Mat<4, 4>> Pf(n, Mat<4, 4>::Identity());
    crx_dropin::use_all_devices(1024);
    crx_dropin::ekf_estimation_run(xf, Pf, zf, uf, Q, R, &hx);
    for (size_t t = 0; t < T; ++t) dump("fleet_x", hx[t * n + (n - 1)].data(), 4);
    dump("fleet_P", Pf[n / 2].data(), 16);
    crx::dropin_check(crx_set_devices(nullptr, 0, 0), "crx_set_devices");
  }
  dump("jacobF", jacobF(xEst, u).data(), 16);
  dump("obs", observation_model(xEst).data(), 2);
  dump("jacobH", jacobH().data(), 8);

  // ---- LQR 5x5 and 4x4 at v = 2.5
  const float v = 2.5f, DT = 0.1, L = 0.5;
  Mat<5, 5> A5 = Mat<5, 5>::Zero();
  A5(0, 0) = 1.0; A5(0, 1) = DT; A5(1, 2) = v; A5(2, 2) = 1.0; A5(2, 3) = DT; A5(4, 4) = 1.0;
  Mat<5, 2> B5 = Mat<5, 2>::Zero();
  B5(3, 0) = v / 0.5; B5(4, 1) = DT;
  dump("X5", solve_DARE(A5, B5, Mat<5, 5>::Identity(), Mat<2, 2>::Identity()).data(), 25);
  dump("K5", dlqr(A5, B5, Mat<5, 5>::Identity(), Mat<2, 2>::Identity()).data(), 10);
  Mat<4, 4> A4 = Mat<4, 4>::Zero();
  A4(0, 0) = 1.0; A4(0, 1) = DT; A4(1, 2) = v; A4(2, 2) = 1.0; A4(2, 3) = DT;
  Mat<4, 1> B4 = Mat<4, 1>::Zero();
  B4(3) = v / L;
  dump("X4", solve_DARE(A4, B4, Mat<4, 4>::Identity(), 1.0f).data(), 16);
  dump("K4", dlqr(A4, B4, Mat<4, 4>::Identity(), 1.0f).data(), 4);

  // ---- MPC, reference horizon T = 6: straight reference along x at 10 km/h
  cpprobotics::State s0(0.0f, 0.3f, 0.05f, 2.0f);
  Mat<4, 6> xref;
  for (int i = 0; i < 6; ++i) { xref(0, i) = 0.5f * (i + 1); xref(1, i) = 0.0f; xref(2, i) = 0.0f; xref(3, i) = 10.0f / 3.6f; }
  cpprobotics::Vec_f sol = mpc_solve<6>(s0, xref);
  dump("mpc", sol.data(), (int)sol.size());

  // ---- tracking call sites: lqr_steering_control + update (src/lqr_speed_steer_control.cpp:195-197, src/lqr_steer_control.cpp:186-188)
  //      and calc_ref_trajectory + update of the MPC loop (src/model_predictive_control.cpp:372-376), on a small arc course
  cpprobotics::Vec_f cx, cy, cyaw, ck, sp;
  for (int i = 0; i < 60; ++i) {
    const float th = 0.02f * i;
    cx.push_back(10.0f * std::sin(th)); cy.push_back(10.0f * (1.0f - std::cos(th))); cyaw.push_back(th);
    ck.push_back(0.1f); sp.push_back(10.0f / 3.6f);
  }
  {
    using namespace crx_dropin::lqr_speed_steer;
    cpprobotics::State st(0.2f, -0.3f, 0.1f, 1.5f);
    float e = 0.0f, e_th = 0.0f;
    for (int t = 0; t < 3; ++t) {
      cpprobotics::Vec_f control = lqr_steering_control(st, cx, cy, cyaw, ck, sp, e, e_th);
      update(st, control[0], control[1]);
      const float row[8] = {control[0], control[1], e, e_th, st.x, st.y, st.yaw, st.v};
      dump("lqr5_tick", row, 8);
    }
  }
  {
    using namespace crx_dropin::lqr_steer;
    cpprobotics::State st(0.2f, -0.3f, 0.1f, 1.5f);
    float e = 0.0f, e_th = 0.0f;
    int ind = 0;
    for (int t = 0; t < 3; ++t) {
      float di = lqr_steering_control(st, cx, cy, cyaw, ck, ind, e, e_th);
      float ai = 1.0 * (sp[ind] - st.v);
      update(st, ai, di);
      const float row[8] = {ai, di, e, e_th, st.x, st.y, st.yaw, st.v};
      dump("lqr4_tick", row, 8);
    }
  }
  {
    cpprobotics::State st(0.5f, 0.2f, 0.05f, 2.0f);
    int target_ind = 0;
    Mat<4, 6> xr;
    crx_dropin::mpc::calc_ref_trajectory<6>(st, cx, cy, cyaw, ck, sp, 1.0f, target_ind, xr);
    dump("xref", xr.data(), 24);
    const float ti = (float)target_ind;
    dump("target_ind", &ti, 1);
    crx_dropin::mpc::update(st, 0.5f, 0.9f);
    const float row[4] = {st.x, st.y, st.yaw, st.v};
    dump("mpc_update", row, 4);
  }
  // ---- the closed loops under the reference's signatures, and the MPC file's windowed nearest-index search
  {
    cpprobotics::Poi_f goal{{cx.back(), cy.back()}};
    crx_dropin::Trajectory t5 = crx_dropin::lqr_speed_steer::closed_loop_prediction(cx, cy, cyaw, ck, sp, goal, 400);
    crx_dropin::Trajectory t4 = crx_dropin::lqr_steer::closed_loop_prediction(cx, cy, cyaw, ck, sp, goal, 400);
    crx_dropin::Trajectory tm = crx_dropin::mpc::mpc_simulation<6>(cx, cy, cyaw, ck, sp, goal, 60);
    const crx_dropin::Trajectory* ts[3] = {&t5, &t4, &tm};
    const char* names[3] = {"loop5", "loop4", "loopm"};
    for (int k = 0; k < 3; ++k) {
      const crx_dropin::Trajectory& t = *ts[k];
      const float head[6] = {(float)t.ticks, t.goal ? 1.0f : 0.0f, t.final_state.x, t.final_state.y, t.final_state.yaw, t.final_state.v};
      dump((std::string(names[k]) + "_end").c_str(), head, 6);
      dump((std::string(names[k]) + "_x").c_str(), t.x_h.data(), (int)t.x_h.size());
      dump((std::string(names[k]) + "_y").c_str(), t.y_h.data(), (int)t.y_h.size());
    }
    cpprobotics::State st(3.0f, 0.6f, 0.3f, 2.0f);
    float idx[3];
    for (int k = 0; k < 3; ++k) idx[k] = (float)crx_dropin::mpc::calc_nearest_index(st, cx, cy, cyaw, 5 + 20 * k);   // 45 + 10 runs to the course end
    dump("window_ind", idx, 3);
  }
  return 0;
}

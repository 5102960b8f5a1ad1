// tracking_fleet.cpp — the reference's two tracking demos for a whole fleet, in C++ against the C ABI:
//   * LQR speed + steer tracking, src/lqr_speed_steer_control.cpp main() :245-269 + closed_loop_prediction :166-243;
//   * MPC speed + steer tracking, src/model_predictive_control.cpp main() :467-491 + mpc_simulation :348-465.
// The set-up side of each main (Spline2D through the way-points, calc_speed_profile, smooth_yaw) comes from the host helpers
// of the library, bit for bit what the mains compute; every agent's whole episode then runs in ONE kernel launch.
//
//   hipcc -O2 -I include examples/tracking_fleet.cpp -o tracking_fleet -L cpprobotics_amd -lcrx -Wl,-rpath,$PWD/cpprobotics_amd
//   ./tracking_fleet [n=4096]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "crx.h"

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)
#define CRX_OK_(call) do { int rc_ = (call); if (rc_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, crx_last_error()); return 3; } } while (0)

template <class T> static T* upload(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc(&d, sizeof(T) * v.size()) != hipSuccess) return nullptr;
  if (hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

struct HostCourse { std::vector<float> cx, cy, cyaw, ck, sp; };

// Spline2D csp_obj(wx, wy); sampled every ds; calc_speed_profile(...)  — the first lines of both mains
static int build_course(const std::vector<float>& wx, const std::vector<float>& wy, double ds, int variant, float target_speed, HostCourse* c) {
  const int k = crx_course_from_waypoints(wx.data(), wy.data(), (int)wx.size(), ds, nullptr, nullptr, nullptr, nullptr, 0);
  if (k <= 0) return k;
  c->cx.resize(k); c->cy.resize(k); c->cyaw.resize(k); c->ck.resize(k); c->sp.resize(k);
  crx_course_from_waypoints(wx.data(), wy.data(), (int)wx.size(), ds, c->cx.data(), c->cy.data(), c->cyaw.data(), c->ck.data(), k);
  return crx_calc_speed_profile(variant, c->cx.data(), c->cy.data(), c->cyaw.data(), k, target_speed, c->sp.data()) ? -1 : k;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 4096;
  if (crx_device_count() == 0) { std::fprintf(stderr, "no HIP device visible: crx has no CPU fallback\n"); return 1; }
  std::mt19937 gen(11);
  std::normal_distribution<float> off(0.0f, 0.2f);
  const float target_speed = 10.0f / 3.6f;
  int* d_ticks = nullptr;
  HIP_OK(hipMalloc(&d_ticks, 4 * (size_t)n));
  std::vector<int> ticks(n);

  {  // ---- LQR: way-points :248-249, ds = 0.1 :256, closed_loop_prediction(..., {{wx.back(), wy.back()}}) :268
    const std::vector<float> wx{0.0f, 6.0f, 12.5f, 10.0f, 17.5f, 20.0f, 25.0f}, wy{0.0f, -3.0f, -5.0f, 6.5f, 3.0f, 0.0f, 0.0f};
    HostCourse hc;
    const int k = build_course(wx, wy, 0.1, /*variant: lqr_speed_steer*/ 5, target_speed, &hc);
    if (k <= 0) { std::fprintf(stderr, "course: %s\n", crx_last_error()); return 3; }
    crx_course dc{k, upload(hc.cx), upload(hc.cy), upload(hc.cyaw), upload(hc.ck), upload(hc.sp)};
    if (!dc.cx || !dc.cy || !dc.cyaw || !dc.ck || !dc.sp) return 2;
    // agent 0 = the reference's start State(-0.0, -0.0, 0.0, 0.0) :171; the others start beside it
    std::vector<float> st(4 * (size_t)n, 0.0f);
    st[0] = -0.0f; st[1] = -0.0f;
    for (int a = 1; a < n; ++a) { st[4 * (size_t)a] = off(gen); st[4 * (size_t)a + 1] = off(gen); st[4 * (size_t)a + 2] = 0.3f * off(gen); }
    float* d_state = upload(st);
    if (!d_state) return 2;
    crx_lqr_params lp; crx_lqr_default_params(&lp);
    crx_vehicle_params vp; crx_vehicle_default_params(&vp, 0);
    crx_loop_params loop{wx.back(), wy.back(), 0.3f /* goal_dis :168 */, 1.0, 0.05f, 1000};
    const auto t0 = std::chrono::steady_clock::now();
    CRX_OK_(crx_lqr_closed_loop_batch_dev(n, 5, d_state, &dc, nullptr, nullptr, nullptr, &lp, &vp, &loop, nullptr, d_ticks, nullptr));
    HIP_OK(hipDeviceSynchronize());
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    HIP_OK(hipMemcpy(ticks.data(), d_ticks, 4 * (size_t)n, hipMemcpyDeviceToHost));
    long long total = 0; int reached = 0;
    for (int t : ticks) { total += t; reached += t < loop.max_ticks; }
    std::printf("LQR: %d agents on the %d-point course, agent 0 (the reference's vehicle) reached the goal after %d ticks; %d of %d reached it; "
                "%.1f M agent-ticks/s\n", n, k, ticks[0], reached, n, (double)total / sec * 1e-6);
    for (const void* q : {(const void*)d_state, (const void*)dc.cx, (const void*)dc.cy, (const void*)dc.cyaw, (const void*)dc.ck, (const void*)dc.sp}) (void)hipFree(const_cast<void*>(q));
  }

  {  // ---- MPC: way-points :469-471, ds = 1.0 :479, mpc_simulation(r_x, r_y, ryaw, rcurvature, speed_profile, goal) :490
    const std::vector<float> wx{0.0f, 60.0f, 125.0f, 50.0f, 75.0f, 35.0f, -10.0f}, wy{0.0f, 0.0f, 50.0f, 65.0f, 30.0f, 50.0f, -20.0f};
    HostCourse hc;
    const int k = build_course(wx, wy, 1.0, /*variant: MPC*/ 0, target_speed, &hc);
    if (k <= 0) { std::fprintf(stderr, "course: %s\n", crx_last_error()); return 3; }
    // State state(cx[0], cy[0], cyaw[0], speed_profile[0]) :349 takes the heading BEFORE smooth_yaw(cyaw) :360
    const float x0 = hc.cx[0], y0 = hc.cy[0], yaw0 = hc.cyaw[0], v0 = hc.sp[0];
    CRX_OK_(crx_smooth_yaw(hc.cyaw.data(), k));
    crx_course dc{k, upload(hc.cx), upload(hc.cy), upload(hc.cyaw), upload(hc.ck), upload(hc.sp)};
    if (!dc.cx || !dc.cy || !dc.cyaw || !dc.ck || !dc.sp) return 2;
    std::vector<float> st(4 * (size_t)n);
    for (int a = 0; a < n; ++a) {
      float* s = &st[4 * (size_t)a];
      s[0] = x0 + (a ? off(gen) : 0.0f); s[1] = y0 + (a ? 2.0f * off(gen) : 0.0f); s[2] = yaw0 + (a ? 0.3f * off(gen) : 0.0f); s[3] = v0;
    }
    float* d_state = upload(st);
    std::vector<int> tind(n, 0);
    int* d_tind = upload(tind);
    if (!d_state || !d_tind) return 2;
    crx_mpc_params mp; crx_mpc_default_params(&mp);
    crx_loop_params loop{wx.back(), wy.back(), 0.5f /* goal_dis :353 */, 1.0, 0.05f, 120};
    const auto t0 = std::chrono::steady_clock::now();
    CRX_OK_(crx_mpc_closed_loop_batch_dev(n, 6 /* #define T 6 :28 */, d_state, &dc, 1.0f /* dl */, 10 /* N_IND_SEARCH */, &mp, &loop, d_tind,
                                          nullptr, d_ticks, nullptr, nullptr));
    HIP_OK(hipDeviceSynchronize());
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    HIP_OK(hipMemcpy(ticks.data(), d_ticks, 4 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(tind.data(), d_tind, 4 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(st.data(), d_state, 16 * (size_t)n, hipMemcpyDeviceToHost));
    long long total = 0;
    for (int t : ticks) total += t;
    std::printf("MPC: %d agents on the %d-point course, %d ticks of mpc_simulation each (T = 6): agent 0 is at course index %d, "
                "(%.2f, %.2f) at %.2f m/s; %.1f M agent-ticks/s\n", n, k, ticks[0], tind[0], st[0], st[1], st[3], (double)total / sec * 1e-6);
    for (const void* q : {(const void*)d_state, (const void*)d_tind, (const void*)dc.cx, (const void*)dc.cy, (const void*)dc.cyaw, (const void*)dc.ck, (const void*)dc.sp}) (void)hipFree(const_cast<void*>(q));
  }
  (void)hipFree(d_ticks);
  return 0;
}

// planner_fleet.cpp — the reference's two sampling planners for a whole fleet, in C++ against the C ABI:
//   * Frenet optimal-trajectory planner, src/frenet_optimal_trajectory.cpp main() :186-236, on its own course and obstacles;
//   * dynamic-window planner, src/dynamic_window_approach.cpp main() :166-238, on its own obstacle field.
// One agent per wavefront, the whole episode of every agent in ONE kernel launch each.
//
//   hipcc -O2 -I include examples/planner_fleet.cpp -o planner_fleet -L cpprobotics_amd -lcrx -Wl,-rpath,$PWD/cpprobotics_amd
//   ./planner_fleet [n=4096]
#include <hip/hip_runtime.h>
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

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 4096;
  if (crx_device_count() == 0) { std::fprintf(stderr, "no HIP device visible: crx has no CPU fallback\n"); return 1; }
  std::mt19937 gen(7);
  int rc = 0;

  // ---- Frenet: Spline2D csp_obj(wx, wy) :203, the sampled course :205-213 (its last point is the goal), obstacles :188-194
  const std::vector<float> wx{0.0f, 10.0f, 20.5f, 35.0f, 70.5f}, wy{0.0f, -6.0f, 5.0f, 6.5f, 0.0f};
  const std::vector<float> ob{20.0f, 10.0f, 30.0f, 6.0f, 30.0f, 8.0f, 35.0f, 8.0f, 50.0f, 3.0f};
  const int nx = (int)wx.size();
  std::vector<float> coef(9 * nx);
  CRX_OK_(crx_frenet_spline_build(wx.data(), wy.data(), nx, coef.data()));
  const int k = crx_frenet_course_samples(coef.data(), nx, nullptr, nullptr, 0);
  std::vector<float> rx(k), ry(k);
  crx_frenet_course_samples(coef.data(), nx, rx.data(), ry.data(), k);
  const float goal[2] = {rx.back(), ry.back()};
  // agent 0 = the reference's start (:215-219); the others start beside it
  std::uniform_real_distribution<float> lat(-1.5f, 1.5f), spd(2.0f, 4.0f);
  std::vector<float> st0(5 * (size_t)n);
  for (int a = 0; a < n; ++a) { float* s = &st0[5 * (size_t)a]; s[0] = 0.0f; s[1] = a ? spd(gen) : 10.0f / 3.6f; s[2] = a ? 2.0f + lat(gen) : 2.0f; s[3] = s[4] = 0.0f; }
  float* d_coef = upload(coef); float* d_ob = upload(ob);
  int *d_ticks = nullptr, *d_status = nullptr;
  HIP_OK(hipMalloc(&d_ticks, 4 * (size_t)n)); HIP_OK(hipMalloc(&d_status, 4 * (size_t)n));
  if (!d_coef || !d_ob) return 2;
  {
    crx_frenet_config cfg;
    crx_frenet_default_config(&cfg);
    float* d_state = upload(st0);
    if (!d_state) return 2;
    const auto t0 = std::chrono::steady_clock::now();
    CRX_OK_(crx_frenet_run_batch_dev(n, 500 /* SIM_LOOP :20 */, d_state, d_coef, nx, goal, d_ob, (int)ob.size() / 2, &cfg, nullptr,
                                     d_ticks, d_status, nullptr, nullptr, nullptr, nullptr, 0, nullptr));
    HIP_OK(hipDeviceSynchronize());
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<int> ticks(n), status(n);
    HIP_OK(hipMemcpy(ticks.data(), d_ticks, 4 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(status.data(), d_status, 4 * (size_t)n, hipMemcpyDeviceToHost));
    long plans = 0; int reached = 0;
    for (int a = 0; a < n; ++a) { plans += ticks[a]; reached += (status[a] & 1) == 0 && ticks[a] < 500; }
    std::printf("Frenet: %d agents, %ld planning calls in %.2f ms (%.2f M plans/s); agent 0: %d ticks, status %d; "
                "%d agents reached the goal\n", n, plans, sec * 1e3, plans / sec / 1e6, ticks[0], status[0], reached);
    if (status[0] != 0 || ticks[0] >= 500) rc = 4;      // the reference's own start (agent 0) must reach the goal
    HIP_OK(hipFree(d_state));
  }

  // ---- dynamic window: State x :167, goal :168, obstacles :169-180, Config :25-41
  const std::vector<float> dob{-1, -1, 0, 2, 4.0f, 2.0f, 5.0f, 4.0f, 5.0f, 5.0f, 5.0f, 6.0f, 5.0f, 9.0f, 8.0f, 9.0f, 7.0f, 9.0f, 12.0f, 12.0f};
  std::uniform_real_distribution<float> jit(-0.3f, 0.3f);
  std::vector<float> dst(5 * (size_t)n), du(2 * (size_t)n, 0.0f), dgoal(2 * (size_t)n);
  for (int a = 0; a < n; ++a) {
    float* s = &dst[5 * (size_t)a];
    s[0] = a ? jit(gen) : 0.0f; s[1] = a ? jit(gen) : 0.0f; s[2] = 3.141592653f / 8.0f; s[3] = 0.0f; s[4] = 0.0f;
    dgoal[2 * (size_t)a] = 10.0f; dgoal[2 * (size_t)a + 1] = 10.0f;
  }
  float *d_dst = upload(dst), *d_du = upload(du), *d_dgoal = upload(dgoal), *d_dob = upload(dob);
  if (!d_dst || !d_du || !d_dgoal || !d_dob) return 2;
  const auto t0 = std::chrono::steady_clock::now();
  CRX_OK_(crx_dwa_run_batch_dev(n, 1000, d_dst, d_du, d_dgoal, d_dob, (int)dob.size() / 2, nullptr, nullptr, d_ticks, d_status, nullptr,
                                nullptr, nullptr));
  HIP_OK(hipDeviceSynchronize());
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::vector<int> ticks(n);
  HIP_OK(hipMemcpy(ticks.data(), d_ticks, 4 * (size_t)n, hipMemcpyDeviceToHost));
  long steps = 0; int reached = 0;
  for (int a = 0; a < n; ++a) { steps += ticks[a]; reached += ticks[a] < 1000; }
  std::printf("DWA: %d agents, %ld control steps in %.2f ms (%.2f M agent-steps/s); agent 0 reached the goal in %d steps; %d of %d "
              "reached it\n", n, steps, sec * 1e3, steps / sec / 1e6, ticks[0], reached, n);
  if (ticks[0] >= 1000) rc = 4;
  return rc;
}

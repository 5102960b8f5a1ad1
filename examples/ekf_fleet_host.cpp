// ekf_fleet_host.cpp — the reference's EKF demo (src/extended_kalman_filter.cpp main(), :109-223) for a whole fleet, in plain
// C++ (no HIP header, any C++ compiler) through the HOST-pointer side of the C ABI: the fleet's inputs are std::vectors, the
// engine stages them itself — pinned rings, H2D / kernel / D2H overlapped over time chunks — and, with a device set installed,
// splits the vehicles over every visible GPU (one host thread and one set of streams per GPU, no collective).
//
//   g++ -O2 -std=c++17 -I include examples/ekf_fleet_host.cpp -o ekf_fleet_host -L cpprobotics_amd -lcrx -Wl,-rpath,$PWD/cpprobotics_amd
//   ./ekf_fleet_host [n=65536] [T=500] [gpus=all visible] [pinned=0] [split=0]
//   (split = k > 0: name device 0 k times in the device set — k shards on one GPU, the code path of a k-GPU host; same results)
//
// The loop of main() (:171-188) for every vehicle: the input side (ud = u + noise, xTrue = motion_model(xTrue, u), z = position +
// noise, :174-181) is evaluated here on the host with the reference's own statements, the T ekf_estimation() calls (:183) are ONE
// crx_ekf_run_batch call for the whole fleet, estimated trajectory (hxEst, :187) included.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "crx.h"

#define CRX_OK_(call) do { int rc_ = (call); if (rc_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, crx_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 65536;
  const int T = argc > 2 ? std::atoi(argv[2]) : 500;   // SIM_TIME 50.0 / DT 0.1 (:16-17)
  const int have = crx_device_count();
  if (have == 0) { std::fprintf(stderr, "no HIP device visible: crx has no CPU fallback\n"); return 1; }
  const int g = std::max(1, std::min(argc > 3 ? std::atoi(argv[3]) : have, have));
  const bool pinned = argc > 4 && std::atoi(argv[4]) != 0;
  const int split = argc > 5 ? std::atoi(argv[5]) : 0;
  const size_t nn = (size_t)n, tt = (size_t)T;
  const double DT = 0.1;

  float Q[16] = {0}, R[4] = {1, 0, 0, 1};                                      // :142-151
  Q[0] = 0.1f * 0.1f; Q[5] = 0.1f * 0.1f; Q[10] = (float)((1.0 / 180 * M_PI) * (1.0 / 180 * M_PI)); Q[15] = 0.1f * 0.1f;
  const float qsim0 = 1.0f, qsim1 = (float)((30.0 / 180 * M_PI) * (30.0 / 180 * M_PI)), rsim = 0.5f * 0.5f;   // :153-160

  // time-major inputs [T][n][2] and the trajectory [T][n][4]: pageable std::vectors, or pinned memory from the engine
  std::vector<float> vz, vu, vh;
  float *z, *ud, *hist;
  if (pinned) {
    z = static_cast<float*>(crx_host_alloc(8 * nn * tt)); ud = static_cast<float*>(crx_host_alloc(8 * nn * tt));
    hist = static_cast<float*>(crx_host_alloc(16 * nn * tt));
    if (!z || !ud || !hist) { std::fprintf(stderr, "crx_host_alloc: %s\n", crx_last_error()); return 2; }
  } else {
    vz.resize(2 * nn * tt); vu.resize(2 * nn * tt); vh.resize(4 * nn * tt);
    z = vz.data(); ud = vu.data(); hist = vh.data();
  }
  std::vector<float> xTrue(4 * nn, 0.0f), xEst(4 * nn, 0.0f), PEst(16 * nn, 0.0f);
  for (size_t a = 0; a < nn; ++a) for (int i = 0; i < 4; ++i) PEst[16 * a + 5 * i] = 1.0f;      // PEst = I (:139)
  std::mt19937 gen(12345);
  std::normal_distribution<float> gaussian_d(0.0f, 1.0f);
  for (size_t t = 0; t < tt; ++t)
    for (size_t a = 0; a < nn; ++a) {
      const float u0 = 1.0f, u1 = 0.1f;                                        // :113-114
      float* x = &xTrue[4 * a];
      ud[(t * nn + a) * 2 + 0] = u0 + gaussian_d(gen) * qsim0;                 // :174-175
      ud[(t * nn + a) * 2 + 1] = u1 + gaussian_d(gen) * qsim1;
      const float c = std::cos(x[2]), s = std::sin(x[2]);                       // motion_model (:22-36)
      x[0] = x[0] + (float)(DT * c) * u0; x[1] = x[1] + (float)(DT * s) * u0; x[2] = x[2] + (float)DT * u1; x[3] = x[3] + u0;
      z[(t * nn + a) * 2 + 0] = x[0] + gaussian_d(gen) * rsim;                 // :180-181
      z[(t * nn + a) * 2 + 1] = x[1] + gaussian_d(gen) * rsim;
    }

  if (split > 0) {
    std::vector<int> same(split, 0);
    CRX_OK_(crx_set_devices(same.data(), split, 1));                           // `split` shards, all on device 0
  } else {
    CRX_OK_(crx_set_devices(nullptr, g, 4096));                                // devices 0 .. g-1; at least 4,096 vehicles per GPU
  }
  CRX_OK_(crx_ekf_run_batch(std::min(n, 4096), 1, xEst.data(), PEst.data(), z, ud, nullptr, nullptr, Q, R, nullptr));   // warm-up: contexts, code objects
  std::fill(xEst.begin(), xEst.end(), 0.0f);
  std::fill(PEst.begin(), PEst.end(), 0.0f);
  for (size_t a = 0; a < nn; ++a) for (int i = 0; i < 4; ++i) PEst[16 * a + 5 * i] = 1.0f;
  double best = 1e30;
  std::vector<float> x0 = xEst, P0 = PEst;
  for (int rep = 0; rep < 3; ++rep) {                                          // rep 0 grows the workspaces; steady state after
    xEst = x0; PEst = P0;
    const auto t0 = std::chrono::steady_clock::now();
    CRX_OK_(crx_ekf_run_batch(n, T, xEst.data(), PEst.data(), z, ud, hist, nullptr, Q, R, nullptr));
    best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  double e_est = 0;
  for (size_t a = 0; a < nn; ++a) e_est += std::hypot(xEst[4 * a] - xTrue[4 * a], xEst[4 * a + 1] - xTrue[4 * a + 1]);
  const double gb = 32.0 * nn * tt / 1e9;
  std::printf("fleet of %d vehicles x %d steps, host arrays (%s) over %d GPU(s): %.2f ms per call = %.2f G EKF updates/s, %.1f GB/s across the "
              "boundary (z, u in; trajectory out)\n", n, T, pinned ? "pinned" : "pageable", g, best * 1e3, nn * tt / best / 1e9, gb / best);
  unsigned long long sum = 0;                                                  // checksum of every estimate of the trajectory
  for (size_t i = 0; i < 4 * nn * tt; ++i) { unsigned w; std::memcpy(&w, &hist[i], 4); sum = sum * 1099511628211ull + w; }
  std::printf("mean final position error of the estimate: %.3f m; last estimate of vehicle 0 in the trajectory: (%.3f, %.3f); "
              "trajectory checksum %016llx\n", e_est / n, hist[((tt - 1) * nn) * 4 + 0], hist[((tt - 1) * nn) * 4 + 1], sum);
  if (pinned) { crx_host_free(z); crx_host_free(ud); crx_host_free(hist); }
  CRX_OK_(crx_shutdown());
  return (e_est / n < 1.0) ? 0 : 4;
}

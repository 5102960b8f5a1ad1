// ekf_fleet.cpp — the reference's EKF demo (src/extended_kalman_filter.cpp main(), :109-223) for a whole fleet, in C++
// against the C ABI: n vehicles, T steps, everything resident on the GPUs, two kernel launches per GPU in total.
//
//   hipcc -O2 -I include examples/ekf_fleet.cpp -o ekf_fleet -L cpprobotics_amd -lcrx -Wl,-rpath,$PWD/cpprobotics_amd -pthread
//   ./ekf_fleet [n=65536] [T=500] [gpus=all visible]
//
// The fleet is split contiguously over the GPUs (vehicles never read one another, :64-78): one host thread per GPU selects its
// device with crx_set_device and runs the same two launches on its shard — no collective, no change to the per-shard code.
// (examples/ekf_fleet_host.cpp is the same fleet through the host-pointer entry point, sharded by crx_set_devices.)
//
// The reference's loop body — ud = u + noise, xTrue/xDR = motion_model(...), z = observation + noise (:174-181), then
// ekf_estimation(xEst, PEst, z, ud, Q, R) (:183) — becomes crx_ekf_simulate_inputs_dev (all T steps of the input side)
// followed by crx_ekf_run_batch_dev (all T filter updates, estimated trajectory out).  The standard-normal draws the
// reference takes from std::normal_distribution are generated on the host here and uploaded once.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>
#include "crx.h"

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)
#define CRX_OK_(call) do { int rc_ = (call); if (rc_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, crx_last_error()); return 3; } } while (0)

struct ShardResult { double sec = 0, e_est = 0, e_dr = 0; int rc = 0; };

// vehicles [a0, a0 + n) of the fleet on device `dev`
static int run_shard(int dev, int a0, int n, int T, ShardResult* out) {
  CRX_OK_(crx_set_device(dev));
  const size_t nn = (size_t)n, tt = (size_t)T;

  // constants of main(): u = (1.0, 0.1) (:113-114), Q, R (:142-151), Qsim, Rsim (:153-160), PEst = I (:139)
  float Q[16] = {0}, R[4] = {1, 0, 0, 1};
  Q[0] = 0.1f * 0.1f; Q[5] = 0.1f * 0.1f; Q[10] = (float)((1.0 / 180 * M_PI) * (1.0 / 180 * M_PI)); Q[15] = 0.1f * 0.1f;
  const float qsim[2] = {1.0f, (float)((30.0 / 180 * M_PI) * (30.0 / 180 * M_PI))}, rsim[2] = {0.5f * 0.5f, 0.5f * 0.5f};
  std::vector<float> u(2 * nn), x0(4 * nn, 0.0f), P0(16 * nn, 0.0f), w(4 * nn * tt);
  for (size_t a = 0; a < nn; ++a) { u[2 * a] = 1.0f; u[2 * a + 1] = 0.1f; for (int i = 0; i < 4; ++i) P0[16 * a + 5 * i] = 1.0f; }
  std::mt19937 gen(12345 + a0);
  std::normal_distribution<float> gaussian_d(0.0f, 1.0f);
  for (auto& v : w) v = gaussian_d(gen);

  float *d_u, *d_xTrue, *d_xDR, *d_w, *d_z, *d_ud, *d_x, *d_P, *d_hist;
  HIP_OK(hipMalloc(&d_u, 8 * nn)); HIP_OK(hipMalloc(&d_xTrue, 16 * nn)); HIP_OK(hipMalloc(&d_xDR, 16 * nn));
  HIP_OK(hipMalloc(&d_w, 16 * nn * tt)); HIP_OK(hipMalloc(&d_z, 8 * nn * tt)); HIP_OK(hipMalloc(&d_ud, 8 * nn * tt));
  HIP_OK(hipMalloc(&d_x, 16 * nn)); HIP_OK(hipMalloc(&d_P, 64 * nn)); HIP_OK(hipMalloc(&d_hist, 16 * nn * tt));
  HIP_OK(hipMemcpy(d_u, u.data(), 8 * nn, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_xTrue, x0.data(), 16 * nn, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_xDR, x0.data(), 16 * nn, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_x, x0.data(), 16 * nn, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_P, P0.data(), 64 * nn, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_w, w.data(), 16 * nn * tt, hipMemcpyHostToDevice));

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  const auto t0 = std::chrono::steady_clock::now();
  CRX_OK_(crx_ekf_simulate_inputs_dev(n, T, d_u, d_xTrue, d_xDR, d_w, d_z, d_ud, nullptr, nullptr, qsim, rsim, nullptr, stream));
  CRX_OK_(crx_ekf_run_batch_dev(n, T, d_x, d_P, d_z, d_ud, d_hist, nullptr, Q, R, nullptr, stream));
  HIP_OK(hipStreamSynchronize(stream));
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  std::vector<float> xEst(4 * nn), xTrue(4 * nn), xDR(4 * nn);
  HIP_OK(hipMemcpy(xEst.data(), d_x, 16 * nn, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(xTrue.data(), d_xTrue, 16 * nn, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(xDR.data(), d_xDR, 16 * nn, hipMemcpyDeviceToHost));
  double e_est = 0, e_dr = 0;
  for (size_t a = 0; a < nn; ++a) {
    e_est += std::hypot(xEst[4 * a] - xTrue[4 * a], xEst[4 * a + 1] - xTrue[4 * a + 1]);
    e_dr += std::hypot(xDR[4 * a] - xTrue[4 * a], xDR[4 * a + 1] - xTrue[4 * a + 1]);
  }
  out->sec = sec; out->e_est = e_est; out->e_dr = e_dr;
  return 0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 65536;
  const int T = argc > 2 ? std::atoi(argv[2]) : 500;   // SIM_TIME 50.0 / DT 0.1 (:16-17)
  const int have = crx_device_count();
  if (have == 0) { std::fprintf(stderr, "no HIP device visible: crx has no CPU fallback\n"); return 1; }
  const int g = std::max(1, std::min(argc > 3 ? std::atoi(argv[3]) : have, std::min(have, n)));
  std::vector<ShardResult> res(g);
  std::vector<std::thread> th;
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0, lo = 0; r < g; ++r) {
    const int len = n / g + (r < n % g ? 1 : 0);
    th.emplace_back([&res, r, lo, len, T] { res[r].rc = run_shard(r, lo, len, T, &res[r]); });
    lo += len;
  }
  for (auto& t : th) t.join();
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  double sec = 0, e_est = 0, e_dr = 0;
  for (auto& r : res) { if (r.rc) return r.rc; sec = std::max(sec, r.sec); e_est += r.e_est; e_dr += r.e_dr; }
  std::printf("fleet of %d vehicles x %d steps on %d GPU(s): %.3f ms for the two launches of the slowest shard (%.1f G EKF updates/s incl. the "
              "input side; %.2f s wall with set-up and transfers)\n", n, T, g, sec * 1e3, (double)n * T / sec / 1e9, wall);
  std::printf("mean final position error: EKF %.3f m, dead reckoning %.3f m\n", e_est / n, e_dr / n);
  return (e_est < e_dr) ? 0 : 4;
}

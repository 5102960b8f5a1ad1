// ekf_fleet_mgpu.cpp — the reference's EKF demo (src/extended_kalman_filter.cpp main(), :109-223) for a fleet sharded over several
// GPUs, ONE PROCESS PER GPU, with the estimated trajectories' final states concatenated by RCCL's all-gather over xGMI — in plain C++
// against the C ABI (crx_comm_* / crx_allgather_dev, include/crx.h), no Python, no torch.
//
//   hipcc -O2 -I include examples/ekf_fleet_mgpu.cpp -o ekf_fleet_mgpu -L cpprobotics_amd -lcrx -Wl,-rpath,$PWD/cpprobotics_amd
//   for r in 0 1 2 3; do ./ekf_fleet_mgpu $r 4 /tmp/crx_id 262144 500 & done; wait       # rank, world, id file, vehicles per GPU, steps
//   ./ekf_fleet_mgpu 0 1 /tmp/crx_id                                                        # a one-GPU "fleet" (what the tests run)
//
// Rank r drives GPU r (HIP_VISIBLE_DEVICES narrows that as usual) and owns the vehicles [r * n, (r + 1) * n) of the swarm: vehicles
// never read one another (:64-78), so the data path has no collective.  The inputs are keyed by the GLOBAL vehicle id
// (crx_normal_draws_dev: Philox counters), so the swarm computes the same bytes however it is sharded.  The one exchange is the
// concatenation of the per-rank results: every rank ends with the final estimate of every vehicle, in global order.
// The communicator's id travels from rank 0 to the others through a file here; an MPI host would broadcast the 128 bytes instead.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "crx.h"

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)
#define CRX_OK_(call) do { int rc_ = (call); if (rc_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, crx_last_error()); return rc_ == CRX_ERR_NO_DEVICE ? 1 : 3; } } while (0)

static bool read_id(const std::string& path, unsigned char* id) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  const size_t got = std::fread(id, 1, CRX_COMM_ID_BYTES, f);
  std::fclose(f);
  return got == CRX_COMM_ID_BYTES;
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s rank world id_file [vehicles_per_gpu=65536] [steps=500]\n", argv[0]); return 64; }
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const std::string id_file = argv[3];
  const int n = argc > 4 ? std::atoi(argv[4]) : 65536, T = argc > 5 ? std::atoi(argv[5]) : 500;
  const size_t nn = (size_t)n, tt = (size_t)T;
  if (crx_device_count() == 0) { std::fprintf(stderr, "no HIP device available (crx has no CPU fallback)\n"); return 1; }
  CRX_OK_(crx_set_device(rank % crx_device_count()));

  // the communicator: rank 0 makes the id and publishes it (write + rename: the others never see half a file)
  unsigned char id[CRX_COMM_ID_BYTES];
  if (rank == 0) {
    CRX_OK_(crx_comm_unique_id(id));
    const std::string tmp = id_file + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 4; }
    std::fclose(f);
    if (std::rename(tmp.c_str(), id_file.c_str()) != 0) { std::fprintf(stderr, "cannot rename to %s\n", id_file.c_str()); return 4; }
  } else {
    int tries = 0;
    while (!read_id(id_file, id)) {
      if (++tries > 600) { std::fprintf(stderr, "rank %d: no communicator id in %s after 60 s\n", rank, id_file.c_str()); return 4; }
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  crx_comm* comm = nullptr;
  CRX_OK_(crx_comm_init_rank(&comm, id, rank, world));

  // constants of main(): u = (1.0, 0.1) (:113-114), Q, R (:142-151), Qsim, Rsim (:153-160), PEst = I (:139)
  float Q[16] = {0}, R[4] = {1, 0, 0, 1};
  Q[0] = 0.1f * 0.1f; Q[5] = 0.1f * 0.1f; Q[10] = (float)((1.0 / 180 * M_PI) * (1.0 / 180 * M_PI)); Q[15] = 0.1f * 0.1f;
  const float qsim[2] = {1.0f, (float)((30.0 / 180 * M_PI) * (30.0 / 180 * M_PI))}, rsim[2] = {0.5f * 0.5f, 0.5f * 0.5f};
  std::vector<float> u(2 * nn), P0(16 * nn, 0.0f);
  for (size_t a = 0; a < nn; ++a) { u[2 * a] = 1.0f; u[2 * a + 1] = 0.1f; for (int i = 0; i < 4; ++i) P0[16 * a + 5 * i] = 1.0f; }

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  float *d_u, *d_xTrue, *d_xDR, *d_w, *d_z, *d_ud, *d_x, *d_P, *d_hist, *d_all;
  HIP_OK(hipMalloc(&d_u, 8 * nn)); HIP_OK(hipMalloc(&d_xTrue, 16 * nn)); HIP_OK(hipMalloc(&d_xDR, 16 * nn));
  HIP_OK(hipMalloc(&d_w, 16 * nn * tt)); HIP_OK(hipMalloc(&d_z, 8 * nn * tt)); HIP_OK(hipMalloc(&d_ud, 8 * nn * tt));
  HIP_OK(hipMalloc(&d_x, 16 * nn)); HIP_OK(hipMalloc(&d_P, 64 * nn)); HIP_OK(hipMalloc(&d_hist, 16 * nn * tt));
  HIP_OK(hipMalloc(&d_all, 16 * nn * (size_t)world));
  HIP_OK(hipMemcpy(d_u, u.data(), 8 * nn, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(d_xTrue, 0, 16 * nn)); HIP_OK(hipMemset(d_xDR, 0, 16 * nn)); HIP_OK(hipMemset(d_x, 0, 16 * nn));
  HIP_OK(hipMemcpy(d_P, P0.data(), 64 * nn, hipMemcpyHostToDevice));

  const auto t0 = std::chrono::steady_clock::now();
  // the loop of :171-188, all T passes: draws keyed by the global vehicle id, the input side, the filter; then the concat
  CRX_OK_(crx_normal_draws_dev(n, T, (long long)rank * n, 12345ull, 0u, d_w, stream));
  CRX_OK_(crx_ekf_simulate_inputs_dev(n, T, d_u, d_xTrue, d_xDR, d_w, d_z, d_ud, nullptr, nullptr, qsim, rsim, nullptr, stream));
  CRX_OK_(crx_ekf_run_batch_dev(n, T, d_x, d_P, d_z, d_ud, d_hist, nullptr, Q, R, nullptr, stream));
  CRX_OK_(crx_allgather_dev(comm, d_x, d_all, 16 * nn, stream));
  HIP_OK(hipStreamSynchronize(stream));
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // every rank holds the whole swarm's final estimates: its own block must be its own state, and the blocks of the others theirs
  std::vector<float> all(4 * nn * (size_t)world), mine(4 * nn), truth(4 * nn);
  HIP_OK(hipMemcpy(all.data(), d_all, 16 * nn * (size_t)world, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(mine.data(), d_x, 16 * nn, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(truth.data(), d_xTrue, 16 * nn, hipMemcpyDeviceToHost));
  if (std::memcmp(all.data() + 4 * nn * (size_t)rank, mine.data(), 16 * nn) != 0) { std::fprintf(stderr, "rank %d: its block of the gather is not its state\n", rank); return 5; }
  double err = 0, sum = 0;
  for (size_t a = 0; a < nn; ++a) err += std::hypot(mine[4 * a] - truth[4 * a], mine[4 * a + 1] - truth[4 * a + 1]);
  for (float v : all) sum += v;
  std::printf("rank %d of %d: %d vehicles x %d steps in %.3f ms (%.3g EKF updates/s on this GPU, first call included); mean final position error %.3f m; "
              "gathered %zu final estimates (checksum %.6e)\n", rank, world, n, T, sec * 1e3, (double)nn * tt / sec, err / nn, nn * (size_t)world, sum);
  CRX_OK_(crx_comm_destroy(comm));
  if (rank == 0) std::remove(id_file.c_str());
  return 0;
}

// copy_pool_test.cpp — host-only stress test of crxh::CopyPool (csrc/crx_host.h): the thread pool that stages pageable arrays into
// pinned rings for the host-pointer entry points.  Random dense / strided 2-D jobs from several submitter threads at once, tickets
// on the submitters' stacks (the lifetime rule the pool documents), results compared byte for byte with a serial copy.
// Build (no GPU needed; the header pulls in the HIP runtime API only for the context structs):
//   hipcc -O2 -std=c++17 -pthread -o copy_pool_test copy_pool_test.cpp
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>
#include "../../cpprobotics_amd/csrc/crx_host.h"

static int run_submitter(int id, int jobs) {
  std::mt19937 gen(1234 + id);
  int bad = 0;
  crxh::CopyPool& pool = crxh::CopyPool::get();
  for (int j = 0; j < jobs; ++j) {
    const size_t rows = 1 + gen() % 40;
    const size_t row_bytes = (j % 7 == 0) ? (1u << 20) + gen() % (3u << 20) : 1 + gen() % 300000;
    const size_t sp = row_bytes + (gen() % 3 ? 0 : gen() % 4096), dp = row_bytes + (gen() % 3 ? 0 : gen() % 4096);
    std::vector<char> src(rows * sp), dst(rows * dp, 0x5a), ref(rows * dp, 0x5a);
    for (auto& c : src) c = (char)gen();
    for (size_t r = 0; r < rows; ++r) std::memcpy(ref.data() + r * dp, src.data() + r * sp, row_bytes);
    {
      crxh::CopyPool::Ticket t1, t2;                       // two tickets in flight from this thread, joined in either order
      const size_t half = rows / 2;
      if (half) pool.submit(crxh::CopyPool::Job{dst.data(), src.data(), row_bytes, half, dp, sp}, &t1);
      pool.submit(crxh::CopyPool::Job{dst.data() + half * dp, src.data() + half * sp, row_bytes, rows - half, dp, sp}, (j & 1) ? &t2 : nullptr);
      if (j & 2) { pool.wait(&t2); pool.wait(&t1); } else { pool.wait(&t1); pool.wait(&t2); }
    }
    if (dst != ref) ++bad;
  }
  return bad;
}

int main(int argc, char** argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 4, jobs = argc > 2 ? std::atoi(argv[2]) : 150;
  std::vector<int> bad(threads, 0);
  std::vector<std::thread> th;
  for (int i = 0; i < threads; ++i) th.emplace_back([&, i] { bad[i] = run_submitter(i, jobs); });
  for (auto& t : th) t.join();
  int total = 0;
  for (int b : bad) total += b;
  // the contiguous balanced partition of the device set
  {
    crxh::DeviceSet& s = crxh::device_set();
    { std::lock_guard<std::mutex> l(s.m); s.devs = {0, 1, 2}; s.min_agents = 1; }
    const auto sh = crxh::shards_for(10, 0);
    if (sh.size() != 3 || sh[0].a0 != 0 || sh[0].a1 != 4 || sh[1].a1 != 7 || sh[2].a1 != 10 || sh[2].dev != 2) ++total;
    { std::lock_guard<std::mutex> l(s.m); s.min_agents = 4; }
    if (crxh::shards_for(10, 0).size() != 2 || crxh::shards_for(3, 0).size() != 1) ++total;
    { std::lock_guard<std::mutex> l(s.m); s.devs.clear(); }
    const auto one = crxh::shards_for(10, 5);
    if (one.size() != 1 || one[0].dev != 5 || one[0].a1 != 10) ++total;
  }
  std::printf("copy pool: %d worker threads, %d submitters x %d jobs, %d failures\n", crxh::CopyPool::get().threads(), threads, jobs, total);
  return total ? 1 : 0;
}

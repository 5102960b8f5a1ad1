// fdlibm_exhaustive.cpp — crx_fdlibm.h against the host libm (glibc): atanf and tanf on all 2^32 inputs, atan2f on a
// structured 3 x 2^32-point sweep.
// Build & run: g++ -O2 -std=c++17 -ffp-contract=off -pthread fdlibm_exhaustive.cpp -o fde && ./fde [stride]
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../cpprobotics_amd/csrc/crx_fdlibm.h"

static inline uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float fl(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline bool same(float a, float b) { return bits(a) == bits(b) || (std::isnan(a) && std::isnan(b)); }

int main(int argc, char** argv) {
  const uint64_t stride = argc > 1 ? std::strtoull(argv[1], nullptr, 0) : 1;
  const unsigned nt = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 8;
  std::atomic<uint64_t> bad_atan{0}, bad_tan{0}, bad_atan2{0}, n_tan{0}, bad_acos{0};
  std::atomic<uint32_t> first_atan{0}, first_tan{0}, first_a2y{0}, first_a2x{0};
  std::vector<std::thread> th;
  for (unsigned k = 0; k < nt; ++k)
    th.emplace_back([&, k] {
      uint64_t b1 = 0, b2 = 0, b3 = 0, nt2 = 0, b4 = 0;
      for (uint64_t m = k * stride; m < (1ull << 32); m += nt * stride) {
        const float x = fl((uint32_t)m);
        if (!same(crx::atanf_(x), atanf(x))) { if (!b1++) first_atan = (uint32_t)m; }
        if (!same(crx::acosf_(x), acosf(x))) ++b4;
        ++nt2;
        if (!same(crx::tanf_(x), tanf(x))) { if (!b2++) first_tan = (uint32_t)m; }
        // atan2f: pair x with a pseudo-random partner y (covers all quadrants / exponent gaps), plus swapped
        uint32_t h = (uint32_t)m * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float y = fl(h);
        if (!same(crx::atan2f_(y, x), atan2f(y, x))) { if (!b3++) { first_a2y = h; first_a2x = (uint32_t)m; } }
        // near-equal magnitudes and small offsets (typical course geometry)
        const float y2 = x * fl(0x3f800000u + (h & 0x7fffffu));
        if (!same(crx::atan2f_(y2, x), atan2f(y2, x))) { if (!b3++) { first_a2y = bits(y2); first_a2x = (uint32_t)m; } }
        if (!same(crx::atan2f_(x, y2), atan2f(x, y2))) { if (!b3++) { first_a2y = (uint32_t)m; first_a2x = bits(y2); } }
      }
      bad_atan += b1; bad_tan += b2; bad_atan2 += b3; n_tan += nt2; bad_acos += b4;
    });
  for (auto& t : th) t.join();
  std::printf("acosf mismatches %llu; ", (unsigned long long)bad_acos.load());
  std::printf("atanf mismatches %llu (first 0x%08x); tanf mismatches %llu of %llu (first 0x%08x); atan2f mismatches %llu (first y=0x%08x x=0x%08x)\n",
              (unsigned long long)bad_atan.load(), first_atan.load(), (unsigned long long)bad_tan.load(),
              (unsigned long long)n_tan.load(), first_tan.load(), (unsigned long long)bad_atan2.load(), first_a2y.load(), first_a2x.load());
  return (bad_atan.load() || bad_tan.load() || bad_atan2.load() || bad_acos.load()) ? 1 : 0;
}

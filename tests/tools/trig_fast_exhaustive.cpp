// trig_fast_exhaustive.cpp — walks EVERY float with |y| < 120 (and a band above) and checks that the
// fused EKF kernel's fast sincos (csrc/ekf_math.h: sincos_fast2) is bit-identical to crx::sincosf_
// (itself bit-identical to glibc's sinf/cosf on all 2^32 inputs, trig_exhaustive.cpp) wherever it
// reports "inside the fast domain", and that it reports "outside" exactly for |y| >= 120 and |y| < 2^-100.
// Build & run:  g++ -O2 -std=c++17 -ffp-contract=off -pthread trig_fast_exhaustive.cpp -o tfe && ./tfe
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../../cpprobotics_amd/csrc/ekf_math.h"

static inline uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float fl(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const unsigned nt = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 8;
  const uint32_t hi = (argc > 1) ? (uint32_t)std::strtoul(argv[1], nullptr, 0) : 0x43000000u;  // |y| < 128
  std::atomic<uint64_t> bad{0}, checked{0}, inside{0};
  std::vector<std::thread> th;
  for (unsigned k = 0; k < nt; ++k)
    th.emplace_back([&, k] {
      uint64_t b = 0, c = 0, in = 0;
      for (uint64_t m = k; m < hi; m += nt) {
        for (uint32_t sign = 0; sign < 2; ++sign) {
          const float y = fl((uint32_t)m | (sign << 31));
          // pair it with a second, different angle to exercise both slots of the pair
          const float y2 = fl(((uint32_t)m ^ 0x00155555u) | ((sign ^ 1) << 31));
          const float ys[2] = {y, y2};
          float s[2], co[2], rs, rc;
          crx::FastDomain dom = crx::fast_domain_init();
          crx::sincos_fast2(ys, s, co, dom);
          const bool ok = crx::fast_domain_ok(dom);
          const bool expect_ok = (std::fabs(y) < 120.0f) && (std::fabs(y2) < 120.0f) && std::fabs(y) >= 0x1p-100f && std::fabs(y2) >= 0x1p-100f;
          if (ok != expect_ok) ++b;
          if (ok) {
            ++in;
            crx::sincosf_(y, &rs, &rc);
            if (bits(rs) != bits(s[0]) || bits(rc) != bits(co[0])) ++b;
            crx::sincosf_(y2, &rs, &rc);
            if (bits(rs) != bits(s[1]) || bits(rc) != bits(co[1])) ++b;
          }
          ++c;
        }
      }
      bad += b; checked += c; inside += in;
    });
  for (auto& t : th) t.join();
  std::printf("checked %llu angle pairs (%llu inside the fast domain), mismatches: %llu\n",
              (unsigned long long)checked.load(), (unsigned long long)inside.load(), (unsigned long long)bad.load());
  return bad.load() ? 1 : 0;
}

// datan2_exhaustive.cpp — csrc/crx_datan2.h (datan2_one_) against the host libm's atan2(y, 1.0) on every argument the tracking
// controllers can form: y = L * (double)k for EVERY float k (2^32 bit patterns) at wheelbase L — 0.5 is the reference's
// (`#define L 0.5`, src/lqr_speed_steer_control.cpp:21, src/lqr_steer_control.cpp:21) — and, with `random`, on random doubles of
// every branch (2^-60 .. 2^60, both signs) plus the branch boundaries.  Build (no contraction beyond the header's explicit fma):
//   g++ -O2 -std=c++17 -mfma -ffp-contract=off -fno-builtin-atan2 -pthread -o datan2_exhaustive datan2_exhaustive.cpp -lm
// Usage: datan2_exhaustive L threads [stride [random]]
//        datan2_exhaustive L threads sums   -> 4096 lines: the sum mod 2^64 of the bit patterns of the HOST LIBM's atan2(L*(double)k, 1.0)
//                                              over the float bit patterns [j << 20, (j+1) << 20) (every NaN counted as 0x7ff8000000000000):
//                                              the checksums crx_x_datan2_sweep_dev forms on the device (tests/test_datan2.py compares them)
// Result on this image's glibc 2.35 (FMA flavour), stride 1: 0 mismatches on all 4,294,967,296 floats for L = 0.5, 2.5 (the MPC
// file's WB), 1.0, 0.3, 2.9 and 1/3, and on 160 M random doubles; also 0 mismatches of the FLOAT the controllers keep.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../cpprobotics_amd/csrc/crx_datan2.h"

static bool same(double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (a != a && b != b); }

int main(int argc, char** argv) {
  const double L = argc > 1 ? std::atof(argv[1]) : 0.5;
  const int threads = argc > 2 ? std::atoi(argv[2]) : 8;
  if (argc > 3 && !std::strcmp(argv[3], "sums")) {
    std::vector<uint64_t> sums(4096, 0);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
      pool.emplace_back([&, t] {
        for (int j = t; j < 4096; j += threads) {
          uint64_t acc = 0;
          for (uint32_t i = 0; i < (1u << 20); ++i) {
            const uint32_t w = ((uint32_t)j << 20) + i;
            float k; std::memcpy(&k, &w, 4);
            const double a = std::atan2(L * (double)k, (double)1.0);
            uint64_t b; std::memcpy(&b, &a, 8);
            acc += (a != a) ? 0x7ff8000000000000ull : b;
          }
          sums[j] = acc;
        }
      });
    for (auto& th : pool) th.join();
    for (uint64_t v : sums) std::printf("%016llx\n", (unsigned long long)v);
    return 0;
  }
  const uint64_t stride = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1;
  const bool random = argc > 4 && !std::strcmp(argv[4], "random");
  std::atomic<long> bad{0}, n{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      long b_ = 0, n_ = 0;
      for (uint64_t b = (uint64_t)t * stride; b < (1ull << 32); b += (uint64_t)threads * stride) {
        const uint32_t w = (uint32_t)b;
        float k; std::memcpy(&k, &w, 4);
        const double y = L * (double)k;
        const double a = std::atan2(y, (double)1.0), c = crx::datan2_one_(y);
        if (!same(a, c)) { if (b_ < 5) std::printf("mismatch k=%a y=%a libm %a crx %a\n", k, y, a, c); ++b_; }
        ++n_;
      }
      if (random) {
        uint64_t st = 0x9e3779b97f4a7c15ull + 977 * t;
        for (long i = 0; i < 20000000; ++i) {
          st ^= st << 13; st ^= st >> 7; st ^= st << 17;
          const double mag = std::ldexp(1.0, (int)(st % 121) - 60) * (1.0 + (double)((st >> 8) & 0xfffffffffffffull) / 4503599627370496.0);
          const double y = (st >> 63) ? -mag : mag;
          const double a = std::atan2(y, (double)1.0), c = crx::datan2_one_(y);
          if (!same(a, c)) { if (b_ < 5) std::printf("mismatch y=%a libm %a crx %a\n", y, a, c); ++b_; }
          ++n_;
        }
      }
      bad += b_; n += n_;
    });
  for (auto& th : pool) th.join();
  if (random) {                                               // branch boundaries and specials
    const double e[] = {0.0, -0.0, 0.0625, 0x1.fffffffffffffp-5, 0x1.0000000000001p-4, 1.0, 0x1.fffffffffffffp-1, 0x1.0000000000001p+0, 16.0,
                        0x1.fffffffffffffp+3, 0x1.0000000000001p+4, 0x1p-57, 0x1.fffffffffffffp-58, 0x1p-56, 0x1p57, 0x1.fffffffffffffp+56,
                        0x1p56, 4.9e-324, 2.2250738585072014e-308, 1.7976931348623157e308, INFINITY, NAN};
    for (double v : e)
      for (int s = 0; s < 2; ++s) {
        const double y = s ? -v : v;
        const double a = std::atan2(y, (double)1.0), c = crx::datan2_one_(y);
        if (!same(a, c)) { std::printf("mismatch (edge) y=%a libm %a crx %a\n", y, a, c); ++bad; }
        ++n;
      }
    for (int i = 0; i <= 240; ++i)                            // every table row: its centre and the neighbouring doubles, direct and reciprocal
      for (int d = -2; d <= 2; ++d) {
        double x = crx::datan2_dbl_(crx::kDatan2Tab[7 * i]);
        for (int q = 0; q < (d < 0 ? -d : d); ++q) x = std::nextafter(x, d < 0 ? 0.0 : 2.0);
        for (double y : {x, -x, 1.0 / x, -1.0 / x, (i + 16) / 256.0, 256.0 / (i + 16)}) {
          const double a = std::atan2(y, (double)1.0), c = crx::datan2_one_(y);
          if (!same(a, c)) { std::printf("mismatch (row %d) y=%a libm %a crx %a\n", i, y, a, c); ++bad; }
          ++n;
        }
      }
  }
  std::printf("L %.17g stride %llu: %ld inputs, %ld mismatches\n", L, (unsigned long long)stride, n.load(), bad.load());
  return bad ? 1 : 0;
}

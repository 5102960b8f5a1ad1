// dt_split_exhaustive.cpp — the proof behind csrc/ekf_math.h: dt_mul_split.  The EKF step forms four entries as
// (float)(DT * (double)t), t a sine or cosine (/root/reference/src/extended_kalman_filter.cpp:30-31,43,45; DT = 0.1 a double
// literal, :16); the fused kernel forms them as fma(t, dt_hi, t * dt_lo) in fp32.  This walks
//   (1) EVERY float t (stride 1: all 2^32): for finite |t| >= 2^-120 the two forms must give the same bits (below that the low
//       product underflows; the count of mismatches there is printed, not required to be zero);
//   (2) every float angle y of the step's fast domain, 2^-100 <= |y| < 120: |sinf(y)| and |cosf(y)| (crx::sincosf_, itself glibc's
//       on all 2^32 inputs) must be >= 2^-120 — the values the step feeds dt_mul_split lie where (1) holds.
// Build & run:  g++ -O2 -std=c++17 -ffp-contract=off -mfma -pthread dt_split_exhaustive.cpp -o dts && ./dts [stride]
// (-mfma only makes fmaf fast; without it libm's correctly rounded fmaf gives the same answers, slowly.)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../cpprobotics_amd/csrc/ekf_math.h"

static inline float fl(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main(int argc, char** argv) {
  const uint64_t stride = argc > 1 ? std::strtoull(argv[1], nullptr, 0) : 1;
  const unsigned nt = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 8;
  const double dt = 0.1;
  if (!crx::dt_split_is_exact(dt)) { std::printf("dt_split_is_exact(0.1) is false\n"); return 2; }
  crx::EkfConsts k{}; k.dt = dt;
  const crx::EkfConstsP kp = crx::pack_consts(k);
  std::atomic<uint64_t> bad{0}, below{0}, checked{0}, small_trig{0}, angles{0};
  std::vector<std::thread> th;
  for (unsigned w = 0; w < nt; ++w)
    th.emplace_back([&, w] {
      uint64_t b = 0, bl = 0, c = 0, st = 0, an = 0;
      for (uint64_t m = (uint64_t)w * stride; m < (1ull << 32); m += (uint64_t)nt * stride) {
        const float t = fl((uint32_t)m);
        if (!(std::fabs(t) <= 3.4028234663852886e38f)) continue;          // NaN, inf
        const float ref = (float)(dt * (double)t);
        const crx::v2f r = crx::dt_mul_split(crx::v2f{t, t}, kp.dt_hi, kp.dt_lo);
        const bool same = bits(r[0]) == bits(ref) && bits(r[1]) == bits(ref);
        if (std::fabs(t) >= 0x1p-120f) { ++c; if (!same) ++b; }
        else if (!same) ++bl;
        if (std::fabs(t) >= 0x1p-100f && std::fabs(t) < 120.0f) {       // (2): t as an angle of the fast domain
          float s, co;
          crx::sincosf_(t, &s, &co);
          ++an;
          if (!(std::fabs(s) >= 0x1p-120f) || !(std::fabs(co) >= 0x1p-120f)) ++st;
        }
      }
      bad += b; below += bl; checked += c; small_trig += st; angles += an;
    });
  for (auto& t : th) t.join();
  std::printf("dt_mul_split vs (float)(0.1 * (double)t): %llu floats with |t| >= 2^-120 checked (stride %llu), %llu mismatching; "
              "%llu mismatching below 2^-120 (outside the claim); %llu fast-domain angles, %llu with |sin| or |cos| < 2^-120\n",
              (unsigned long long)checked.load(), (unsigned long long)stride, (unsigned long long)bad.load(),
              (unsigned long long)below.load(), (unsigned long long)angles.load(), (unsigned long long)small_trig.load());
  return (bad.load() || small_trig.load()) ? 1 : 0;
}

// trig_scheme_search.cpp — round 5: which cheaper evaluation orders of glibc's sinf / cosf polynomials (FMA flavour) round to the SAME
// float as glibc's own order for every float 2^-100 <= |y| < 120 (the fused EKF kernel's fast domain)?  Walks all 1,793,064,960 inputs;
// prints the mismatch count of every candidate (sine and cosine separately; candidates 4 / 7 and 4 / 6 are sanity copies).  All of them
// came out at 0 (profiles/r05/trig_scheme_search.txt); ekf_math.h: sincos_fast2 uses sine 0 and cosine 0 (Horner), and
// trig_fast_exhaustive.cpp checks the finished function — reduction and quadrant logic included — against crx::sincosf_.
// Build & run:  g++ -O2 -std=c++17 -ffp-contract=off -mfma -pthread trig_scheme_search.cpp -o tss && ./tss
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <thread>
#include <vector>
static inline float fl(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
static const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
static const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
#define F __builtin_fma
constexpr int NS = 8, NC = 10;
int main() {
  const unsigned nt = 8;
  std::atomic<uint64_t> badS[NS], badC[NC], total{0};
  for (auto& b : badS) b = 0; for (auto& b : badC) b = 0;
  const uint32_t lo = bits(0x1p-100f), hi = bits(120.0f);
  std::vector<std::thread> th;
  for (unsigned k = 0; k < nt; ++k) th.emplace_back([&, k] {
    uint64_t bs[NS] = {0}, bc[NC] = {0}, tot = 0;
    for (uint32_t m = lo + k; m < hi; m += nt) for (int sg = 0; sg < 2; ++sg) {
      const float y = fl(m | ((uint32_t)sg << 31));
      double x = (double)y;
      const int n = ((int32_t)(x * hpi_inv) + 0x800000) >> 24;
      x = F(-(double)n, hpi, x);
      const double x2 = x * x;
      // reference (glibc order, fma variant)
      const double x3 = x * x2, q1 = F(x2, s3, s2), x5 = x3 * x2, sa = F(x3, s1, x);
      const float S = (float)F(x5, q1, sa);
      const double x4 = x2 * x2, k2 = F(x2, c4, c3), k1 = F(x2, c1, c0), x6 = x4 * x2, ca = F(x4, c2, k1);
      const float C = (float)F(x6, k2, ca);
      float s[NS], c[NC];
      { double p = F(x2, s3, s2); p = F(x2, p, s1); s[0] = (float)F(x3, p, x); }                       // Horner on x3   (4 ops: x3, 3 fma)
      { double p = F(x2, s3, s2); p = F(x2, p, s1); s[1] = (float)F(x, x2 * p, x); }                   // x*(x2 p) + x
      { double p = F(x2, s3, s2); p = F(x2, p, s1); s[2] = (float)(x * F(x2, p, 1.0)); }               // x * (1 + x2 p)
      { double p = F(x2, s3, s2); p = F(x2, p, s1); s[3] = (float)F(x * p, x2, x); }                   // (x p) x2 + x
      { double p = F(x2, s3, s2); s[4] = (float)F(x3, F(x2, p, s1), x); }                              // same as 0 (sanity)
      { double p = F(x4, s3, F(x2, s2, s1)); s[5] = (float)F(x3, p, x); }                              // s1 + x2 s2 + x4 s3 (needs x4: shared with cos Estrin)
      { double p = F(x2, s2, s1); double q = x4 * s3; s[6] = (float)F(x3, p + q, x); }
      { double p = F(x2, s3, s2); s[7] = (float)F(x5, p, F(x3, s1, x)); }                              // reference again (sanity: 0)
      { double p = F(x2, c4, c3); p = F(x2, p, c2); p = F(x2, p, c1); c[0] = (float)F(x2, p, c0); }    // Horner (4 fma)
      { double a = F(x2, c1, c0), b = F(x2, c3, c2); double d = F(x4, c4, b); c[1] = (float)F(x4, d, a); }   // Estrin (x4 + 4)
      { double p = F(x2, c4, c3); p = F(x2, p, c2); c[2] = (float)F(x4, p, F(x2, c1, c0)); }           // x4*(c2 + x2(c3 + x2 c4)) + (c0 + x2 c1)   (x4 + 4)
      { double p = F(x2, c4, c3); p = F(x2, p, c2); p = F(x2, p, c1); c[3] = (float)(1.0 + x2 * p); }  // mul then add
      { double p = F(x2, c4, c3); p = F(x2, p, c2); c[4] = (float)F(x2, F(x2, p, c1), c0); }           // = 0 (sanity)
      { double p = F(x4, c4, F(x2, c3, c2)); c[5] = (float)F(x4, p, F(x2, c1, c0)); }                  // = Estrin reorder (x4 + 4)
      { double p = F(x2, c4, c3); c[6] = (float)F(x6, p, F(x4, c2, F(x2, c1, c0))); }                  // reference (sanity 0)
      { double p = F(x2, c4, c3); p = F(x4, p, F(x2, c2, c1)); c[7] = (float)F(x2, p, c0); }           // c0 + x2*((c1 + x2 c2) + x4 (c3 + x2 c4))   (x4 + 4)
      { double p = F(x2, c4, c3); double q = F(x2, c2, c1); c[8] = (float)F(x2, F(x4, p, q), 1.0); }   // same as 7
      { double p = F(x2, c3, c2); p = F(x4, c4, p); p = F(x2, p, c1); c[9] = (float)F(x2, p, c0); }    // (x4 + 4)
      for (int i = 0; i < NS; ++i) bs[i] += bits(s[i]) != bits(S);
      for (int i = 0; i < NC; ++i) bc[i] += bits(c[i]) != bits(C);
      ++tot;
    }
    for (int i = 0; i < NS; ++i) badS[i] += bs[i];
    for (int i = 0; i < NC; ++i) badC[i] += bc[i];
    total += tot;
  });
  for (auto& t : th) t.join();
  std::printf("inputs %llu\n", (unsigned long long)total.load());
  for (int i = 0; i < NS; ++i) std::printf("sin scheme %d: %llu mismatches\n", i, (unsigned long long)badS[i].load());
  for (int i = 0; i < NC; ++i) std::printf("cos scheme %d: %llu mismatches\n", i, (unsigned long long)badC[i].load());
}

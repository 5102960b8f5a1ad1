// Exhaustive check of crx::sinf_/cosf_ against the host libm over all 2^32 float bit patterns
// (or a strided subset: argv[1] = stride, default 1).  Test infrastructure only.
// Build: g++ -O2 -ffp-contract=off -fopenmp -I cpprobotics_amd/csrc tests/tools/trig_exhaustive.cpp -o /tmp/trig_ex -lm
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "crx_trig.h"

int main(int argc, char** argv) {
  unsigned long long stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
  unsigned long long bad_s = 0, bad_c = 0, bad_sc = 0, n = 0;
  unsigned first_bad = 0; int have = 0;
#pragma omp parallel for reduction(+:bad_s,bad_c,bad_sc,n) schedule(static)
  for (long long hi = 0; hi < 65536; ++hi) {
    for (unsigned long long lo = 0; lo < 65536; lo += stride) {
      uint32_t bits = ((uint32_t)hi << 16) | (uint32_t)lo;
      float x; memcpy(&x, &bits, 4);
      float a = crx::sinf_(x), b = sinf(x);
      float c = crx::cosf_(x), d = cosf(x);
      float e, f; crx::sincosf_(x, &e, &f);
      bool s_ok = (a == b) || (a != a && b != b);
      bool c_ok = (c == d) || (c != c && d != d);
      bool sc_ok = (memcmp(&e, &a, 4) == 0 && memcmp(&f, &c, 4) == 0) || (a != a);
      if (!s_ok) { bad_s++; if (!have) { have = 1; first_bad = bits; } }
      if (!c_ok) { bad_c++; if (!have) { have = 1; first_bad = bits; } }
      if (!sc_ok) bad_sc++;
      n++;
    }
  }
  printf("{\"checked\": %llu, \"sin_mismatch\": %llu, \"cos_mismatch\": %llu, \"sincos_inconsistent\": %llu, \"first_bad_bits\": \"0x%08x\"}\n",
         n, bad_s, bad_c, bad_sc, first_bad);
  return (bad_s || bad_c || bad_sc) ? 1 : 0;
}

// expf_exhaustive.cpp — crx::expf_ (cpprobotics_amd/csrc/crx_trig.h) against the host libm's expf on every float bit pattern
// (or every stride-th one).  usage: expf_ex [stride]   exit status 0 = no mismatch.  Build with -DCRX_TRIG_FMA=0 for the SSE2 flavour.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "crx_trig.h"

int main(int argc, char** argv) {
  const long stride = argc > 1 ? std::atol(argv[1]) : 1;
  long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (long i = 0; i < (1L << 32); i += stride) {
    const uint32_t b = (uint32_t)i;
    float x; std::memcpy(&x, &b, 4);
    const float a = crx::expf_(x), h = expf(x);
    uint32_t ab, hb; std::memcpy(&ab, &a, 4); std::memcpy(&hb, &h, 4);
    if (ab != hb && !(a != a && h != h)) {
      if (bad < 4) std::printf("x=%a crx=%a libm=%a\n", x, a, h);
      ++bad;
    }
  }
  std::printf("%ld mismatches (stride %ld)\n", bad, stride);
  return bad ? 1 : 0;
}

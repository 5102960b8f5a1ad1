// recip_exhaustive.hip — on the GPU, for EVERY float d with 2^-60 <= |d| <= 2^60 (the fused EKF step's fast domain for det S): is a
// shorter refinement of v_rcp_f32 bit-identical to the correctly rounded 1.0f / d (the compiler's IEEE division, which
// recip_fast's rcp + six fma reproduce)?  Prints the number of mismatching inputs per candidate.
// Build: hipcc --offload-arch=gfx950 -O3 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero tests/tools/recip_exhaustive.hip -o /tmp/recip_ex
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float six(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  const float r1 = __builtin_fmaf(e, r, r);
  const float rem = __builtin_fmaf(-d, r1, 1.0f);
  const float q1 = __builtin_fmaf(rem, r1, r1);
  const float rem2 = __builtin_fmaf(-d, q1, 1.0f);
  return __builtin_fmaf(rem2, r1, q1);
}
__device__ __forceinline__ float two(float d) {      // one Newton step
  const float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float four(float d) {     // two Newton steps
  const float r = __builtin_amdgcn_rcpf(d);
  const float r1 = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  return __builtin_fmaf(__builtin_fmaf(-d, r1, 1.0f), r1, r1);
}
__device__ __forceinline__ float four_b(float d) {   // Newton step, then a residual correction against the first estimate
  const float r = __builtin_amdgcn_rcpf(d);
  const float r1 = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  return __builtin_fmaf(__builtin_fmaf(-d, r1, 1.0f), r, r1);
}
__global__ void k(unsigned long long* bad) {
  const uint32_t lo = 0x21800000u /* 2^-60 */, hi = 0x5d800000u /* 2^60 */;
  unsigned long long b[5] = {0, 0, 0, 0, 0};
  for (uint64_t m = (uint64_t)lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; m <= hi; m += (uint64_t)gridDim.x * blockDim.x)
    for (int sg = 0; sg < 2; ++sg) {
      const float d = __uint_as_float((uint32_t)m | ((uint32_t)sg << 31));
      const float ref = 1.0f / d;
      b[0] += __float_as_uint(six(d)) != __float_as_uint(ref);
      b[1] += __float_as_uint(two(d)) != __float_as_uint(ref);
      b[2] += __float_as_uint(four(d)) != __float_as_uint(ref);
      b[3] += __float_as_uint(four_b(d)) != __float_as_uint(ref);
      b[4] += 1;
    }
  for (int i = 0; i < 5; ++i) atomicAdd(&bad[i], b[i]);
}
int main() {
  unsigned long long* d; (void)hipMalloc(&d, 5 * sizeof(unsigned long long)); (void)hipMemset(d, 0, 5 * sizeof(unsigned long long));
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d);
  unsigned long long h[5]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  std::printf("inputs %llu\nrcp + 6 fma (recip_fast): %llu mismatches vs 1.0f / d\nrcp + 2 fma: %llu\nrcp + 4 fma (two Newton steps): %llu\nrcp + 4 fma (Newton, then residual x first estimate): %llu\n",
              h[4], h[0], h[1], h[2], h[3]);
  return 0;
}

// api_probes.inl — part of the single translation unit crx_api.hip (#included there, in this order: api_internal, api_core, api_ekf,
// api_lqr, api_mpc, api_track, api_planners, api_frenet, api_probes); measurement-only probes of include/crx_experimental.h that are not forced variants of a product kernel: device
// libm restatements (dsincos, datan2) and the HBM streaming kernel.
extern "C" {

// Probe of the device's double sin / cos (crx_dsincos.h, the table staged in LDS as the Frenet kernel does): c[i] = cos(x[i]),
// s[i] = sin(x[i]).  tests/test_dsincos.py compares the bits with the host libm's.
namespace crx {
__global__ void __launch_bounds__(256) dsincos_probe_kernel(int n, const double* __restrict__ x, double* __restrict__ s, double* __restrict__ c) {
  __shared__ uint64_t s_sc[kDsincosTabLen];
  for (int i = threadIdx.x; i < kDsincosTabLen; i += blockDim.x) s_sc[i] = kDsincosTab[i];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  c[i] = dcos_(v, s_sc);
  s[i] = dsin_(v, s_sc);
}
}  // namespace crx
int crx_x_dsincos_dev(int n, const double* x, double* s, double* c, void* stream) {
  CRX_TRACE();
  if (n < 0 || (n > 0 && (!x || !s || !c))) return fail(CRX_ERR_INVALID, "dsincos: bad arguments");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::dsincos_probe_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, n, x, s, c);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// Probes of the device's double atan2(y, 1.0) (crx_datan2.h).  tests/test_datan2.py compares the bits with the host libm's:
// samples through crx_x_datan2_dev, all 2^32 float curvatures through the block checksums of crx_x_datan2_sweep_dev.
namespace crx {
__global__ void __launch_bounds__(256) datan2_probe_kernel(int n, const double* __restrict__ y, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = datan2_one_(y[i]);
}
__global__ void __launch_bounds__(256) datan2_sweep_kernel(double L, unsigned long long* __restrict__ sums, unsigned long long* __restrict__ ocml_diff,
                                                           unsigned* __restrict__ diff_k) {
  const unsigned base = blockIdx.x << 20;
  unsigned long long sum = 0, dd = 0, df = 0;
  for (unsigned i = 0; i < 4096; ++i) {
    const unsigned w = base + i * 256u + threadIdx.x;
    const double y = L * (double)__uint_as_float(w);
    const double a = datan2_one_(y);
    sum += (a != a) ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(a);   // one pattern for every NaN
    const double o = atan(y);                                                                 // OCML's: what rounds 1-3 evaluated here
    if (!(o != o && a != a)) {
      dd += __double_as_longlong(o) != __double_as_longlong(a);
      const bool fd = __float_as_uint((float)o) != __float_as_uint((float)a);
      df += fd;
      if (fd && diff_k) { const unsigned long long slot = atomicAdd(&ocml_diff[2], 1ull); if (slot < 64) diff_k[slot] = w; }
    }
  }
  atomicAdd(&sums[blockIdx.x], sum);
  if (ocml_diff) { if (dd) atomicAdd(&ocml_diff[0], dd); if (df) atomicAdd(&ocml_diff[1], df); }
}
}  // namespace crx
int crx_x_datan2_dev(int n, const double* y, double* out, void* stream) {
  CRX_TRACE();
  if (n < 0 || (n > 0 && (!y || !out))) return fail(CRX_ERR_INVALID, "datan2: bad arguments");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::datan2_probe_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, n, y, out);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}
int crx_x_datan2_sweep_dev(double L, unsigned long long* sums, unsigned long long* ocml_diff, unsigned* diff_k, void* stream) {
  CRX_TRACE();
  if (!sums || (diff_k && !ocml_diff)) return fail(CRX_ERR_INVALID, "datan2_sweep: bad arguments");
  if (int rc = check_device()) return rc;
  CRX_HIP(hipMemsetAsync(sums, 0, 4096 * sizeof(unsigned long long), (hipStream_t)stream));
  if (ocml_diff) CRX_HIP(hipMemsetAsync(ocml_diff, 0, 3 * sizeof(unsigned long long), (hipStream_t)stream));
  if (diff_k) CRX_HIP(hipMemsetAsync(diff_k, 0, 64 * sizeof(unsigned), (hipStream_t)stream));
  hipLaunchKernelGGL(crx::datan2_sweep_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, L, sums, ocml_diff, diff_k);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// HBM calibration (scripts/gpu_hbm_calib.py): what a plain streaming kernel reaches on this box, next to the 8 TB/s the rooflines are
// priced against.  mode 0: dst = src (read + write), 1: read only (a word per workgroup written), 2: write only, 3: dst += 1 in place
// (read and write of the same lines: the single-step EKF's traffic shape).  16 bytes per lane per access, grid-stride.
namespace crx {
__global__ void __launch_bounds__(256) hbm_stream_kernel(int mode, size_t n16, v4f* __restrict__ dst, const v4f* __restrict__ src) {
  // mode + 8: workgroup b works where workgroup (b % 8) * (gridDim / 8) + b / 8 would — consecutive workgroups go to the eight XCDs
  // in turn, so this hands every XCD one contiguous eighth of the buffer instead of every eighth 4-KiB piece
  unsigned b = blockIdx.x;
  const bool plain = mode >= 16;                       // mode + 16: plain (cached) stores / loads instead of nontemporal ones
  if (plain) mode -= 16;
  if (mode >= 8) { mode -= 8; b = (b & 7u) * (gridDim.x >> 3) + (b >> 3); }
  const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)b * 256 + threadIdx.x;
  if (mode == 0) {
    for (size_t i = i0; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
  } else if (mode == 1) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    if (plain) { for (size_t i = i0; i < n16; i += stride) acc += src[i]; }
    else for (size_t i = i0; i < n16; i += stride) acc += __builtin_nontemporal_load(src + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) dst[blockIdx.x] = acc;     // keeps the loads alive; practically never true
  } else if (mode == 2) {
    if (plain) { for (size_t i = i0; i < n16; i += stride) dst[i] = v4f{1.f, 2.f, 3.f, 4.f}; }
    else for (size_t i = i0; i < n16; i += stride) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, dst + i);
  } else {
    for (size_t i = i0; i < n16; i += stride) dst[i] = dst[i] + v4f{1.f, 1.f, 1.f, 1.f};
  }
}
}  // namespace crx
int crx_x_hbm_stream_dev(int mode, void* dst, const void* src, size_t bytes, int workgroups, void* stream) {
  CRX_TRACE();
  if (mode < 0 || (mode & 7) > 3 || mode > 27 || (mode >= 16 && (mode & 7) != 2 && (mode & 7) != 1) || !dst || (((mode & 7) == 0 || (mode & 7) == 1) && !src) || bytes % 16 || workgroups < 1 ||
      ((mode & 8) && workgroups % 8))
    return fail(CRX_ERR_INVALID, "hbm_stream: bad arguments (bytes a multiple of 16; mode + 8 needs a multiple of 8 workgroups)");
  if (int rc = check_device()) return rc;
  const size_t n16 = bytes / 16;
  auto* d = (crx::v4f*)dst; auto* sp = (const crx::v4f*)src;
  const dim3 g((unsigned)workgroups), b(256);
  hipLaunchKernelGGL(crx::hbm_stream_kernel, g, b, 0, (hipStream_t)stream, mode, n16, d, sp);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}
}  // extern "C"

// What FETCH_SIZE / WRITE_SIZE count for NARROW accesses (scripts/experiments/gpu_fetch_size_units.py; VERDICT r5: the x 2 on FETCH_SIZE
// was calibrated on 16-byte-per-lane streaming reads only, the MPC solver's private memory is read 4 and 8 bytes per lane):
//   mode 0: every lane reads 4 bytes, coalesced (256 B per wave instruction), each byte of src once
//   mode 1: 8 bytes per lane
//   mode 2: a PRIVATE array of 512 doubles per lane (4 KiB, like the solver's knots: scratch_store / scratch_load_dwordx2 with a
//           wave-uniform dynamic index), written once and read back once per pass, `passes` passes; dst[tid] gets the sum.
//           Known bytes: passes x 4 KiB written and read per lane.
namespace crx {
__global__ void __launch_bounds__(256) fetch_units_kernel(int mode, size_t n, const unsigned* __restrict__ src, double* __restrict__ dst, int passes) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  if (mode == 0) {
    unsigned acc = 0;
    for (size_t i = tid; i < n; i += stride) acc += src[i];
    if (acc == 0x12345678u) dst[tid] = 1.0;
  } else if (mode == 1) {
    const uint2* s2 = reinterpret_cast<const uint2*>(src);
    unsigned acc = 0;
    for (size_t i = tid; i < n / 2; i += stride) { const uint2 v = s2[i]; acc += v.x ^ v.y; }
    if (acc == 0x12345678u) dst[tid] = 1.0;
  } else {
    double a[512];
    double sum = 0.0;
    for (int p = 0; p < passes; ++p) {
#pragma unroll 1
      for (int k = 0; k < 512; ++k) a[k] = (double)(k + p) + (double)threadIdx.x;
#pragma unroll 1
      for (int k = 0; k < 512; ++k) sum += a[(k * 37 + p) & 511];
    }
    dst[tid] = sum;
  }
}
}  // namespace crx
extern "C" int crx_x_fetch_units_dev(int mode, const void* src, size_t bytes, double* dst, int workgroups, int passes, void* stream) {
  CRX_TRACE();
  if (mode < 0 || mode > 2 || !dst || workgroups < 1 || (mode < 2 && (!src || bytes % 8)) || (mode == 2 && passes < 1))
    return fail(CRX_ERR_INVALID, "fetch_units: bad arguments");
  if (int rc = check_device()) return rc;
  hipLaunchKernelGGL(crx::fetch_units_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, mode, bytes / 4, (const unsigned*)src, dst, passes);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// The reciprocal of the fused EKF step (csrc/ekf_math.h: recip_fast — v_rcp_f32 and one Newton step) against the compiler's IEEE 1.0f / d
// on EVERY float of its domain, 2^-60 <= |d| <= 2^60: counts[0] = inputs walked, counts[1] = inputs where recip_fast's two fma differ from
// 1.0f / d, counts[2] = inputs where the six-fma form of rounds 2-4 does.  The claim rests on this device's v_rcp_f32, so the GPU tests
// run the sweep on the device they run on (~1 ms).
namespace crx {
// the six-fma form of rounds 2-4 (= the compiler's own IEEE division in this range): the sweep's second witness
__device__ __forceinline__ float recip_six_fma(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  const float r1 = __builtin_fmaf(e, r, r);
  const float rem = __builtin_fmaf(-d, r1, 1.0f);
  const float q1 = __builtin_fmaf(rem, r1, r1);
  const float rem2 = __builtin_fmaf(-d, q1, 1.0f);
  return __builtin_fmaf(rem2, r1, q1);
}
__global__ void __launch_bounds__(256) recip_sweep_kernel(unsigned long long* __restrict__ counts) {
  const uint32_t lo = 0x21800000u /* 2^-60 */, hi = 0x5d800000u /* 2^60 */;
  unsigned long long n = 0, b2 = 0, b6 = 0;
  for (uint64_t m = (uint64_t)lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m <= hi; m += (uint64_t)gridDim.x * blockDim.x)
    for (uint32_t sg = 0; sg < 2; ++sg) {
      const float d = __uint_as_float((uint32_t)m | (sg << 31));
      const float ref = 1.0f / d;
      FastDomain dom = fast_domain_init();
      b2 += __float_as_uint(recip_fast(d, dom)) != __float_as_uint(ref);
      b6 += __float_as_uint(recip_six_fma(d)) != __float_as_uint(ref);
      ++n;
    }
  atomicAdd(&counts[0], n);
  if (b2) atomicAdd(&counts[1], b2);
  if (b6) atomicAdd(&counts[2], b6);
}
}  // namespace crx
extern "C" {
int crx_x_recip_sweep_dev(unsigned long long* counts, void* stream) {
  CRX_TRACE();
  if (!counts) return fail(CRX_ERR_INVALID, "recip_sweep: counts is NULL");
  if (int rc = check_device()) return rc;
  CRX_HIP(hipMemsetAsync(counts, 0, 3 * sizeof(unsigned long long), (hipStream_t)stream));
  hipLaunchKernelGGL(crx::recip_sweep_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, counts);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}
}  // extern "C"

// valu_rates.hip — single-wave issue rate / dependent-issue latency of the VALU instructions the EKF
// kernel is made of, on gfx950.  One wave per SIMD (grid = 1024 x 64) like the headline launch, and
// two waves per SIMD for comparison.  Prints shader-clock cycles per instruction per wave.
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// INDEP: 8 independent accumulators round-robin;  DEP: one chain
#define KERNEL(name, decl, body_indep, body_dep)                                        \
  __global__ void __launch_bounds__(64) name##_indep(long long* out, int iters, float seed) { \
    decl;                                                                                \
    long long t0 = clock64();                                                            \
    for (int i = 0; i < iters; ++i) { REP8(body_indep) }                                 \
    long long t1 = clock64();                                                            \
    sink;                                                                                \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                     \
  }                                                                                      \
  __global__ void __launch_bounds__(64) name##_dep(long long* out, int iters, float seed) {   \
    decl;                                                                                \
    long long t0 = clock64();                                                            \
    for (int i = 0; i < iters; ++i) { REP64(body_dep) }                                  \
    long long t1 = clock64();                                                            \
    sink;                                                                                \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                     \
  }

typedef float v2f __attribute__((ext_vector_type(2)));

#define DECL_F32 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5f
#define DECL_F64 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5
#define DECL_V2 v2f a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b = a0 * 0.5f
#define sink asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7))

#define OP1(op, r) asm volatile(op " %0, %0, %1" : "+v"(r) : "v"(b));
#define ALL8(op) OP1(op, a0) OP1(op, a1) OP1(op, a2) OP1(op, a3) OP1(op, a4) OP1(op, a5) OP1(op, a6) OP1(op, a7)

KERNEL(mul_f32, DECL_F32, ALL8("v_mul_f32"), OP1("v_mul_f32", a0))
KERNEL(pk_mul_f32, DECL_V2, ALL8("v_pk_mul_f32"), OP1("v_pk_mul_f32", a0))
KERNEL(pk_add_f32, DECL_V2, ALL8("v_pk_add_f32"), OP1("v_pk_add_f32", a0))
KERNEL(mul_f64, DECL_F64, ALL8("v_mul_f64"), OP1("v_mul_f64", a0))
#define FMA64(r) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r) : "v"(b));
KERNEL(fma_f64, DECL_F64, FMA64(a0) FMA64(a1) FMA64(a2) FMA64(a3) FMA64(a4) FMA64(a5) FMA64(a6) FMA64(a7), FMA64(a0))
#define FMA32(r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(b));
KERNEL(fma_f32, DECL_F32, FMA32(a0) FMA32(a1) FMA32(a2) FMA32(a3) FMA32(a4) FMA32(a5) FMA32(a6) FMA32(a7), FMA32(a0))
#define CNDM(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b));
KERNEL(cndmask, DECL_F32, CNDM(a0) CNDM(a1) CNDM(a2) CNDM(a3) CNDM(a4) CNDM(a5) CNDM(a6) CNDM(a7), CNDM(a0))
#define RCP(r) asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
KERNEL(rcp_f32, DECL_F32, RCP(a0) RCP(a1) RCP(a2) RCP(a3) RCP(a4) RCP(a5) RCP(a6) RCP(a7), RCP(a0))

// conversions: f32 <-> f64 round trips (each pair = 2 instructions)
#define CVTRT(r) { double t; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t) : "v"(r)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(t)); }
KERNEL(cvt_f64_f32_rt, DECL_F32, CVTRT(a0) CVTRT(a1) CVTRT(a2) CVTRT(a3) CVTRT(a4) CVTRT(a5) CVTRT(a6) CVTRT(a7), CVTRT(a0))
#define CVTUP(r, d) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(r));
#define DECL_UP DECL_F32; double d0, d1, d2, d3, d4, d5, d6, d7
#undef sink
#define sink asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7))
KERNEL(cvt_f64_f32, DECL_UP, CVTUP(a0, d0) CVTUP(a1, d1) CVTUP(a2, d2) CVTUP(a3, d3) CVTUP(a4, d4) CVTUP(a5, d5) CVTUP(a6, d6) CVTUP(a7, d7),
       CVTUP(a0, d0) d1 = d2 = d3 = d4 = d5 = d6 = d7 = d0;)
#undef sink
#define DECL_DN DECL_F64; float f0, f1, f2, f3, f4, f5, f6, f7
#define sink asm volatile("" ::"v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7))
#define CVTDN(d, r) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(d));
KERNEL(cvt_f32_f64, DECL_DN, CVTDN(a0, f0) CVTDN(a1, f1) CVTDN(a2, f2) CVTDN(a3, f3) CVTDN(a4, f4) CVTDN(a5, f5) CVTDN(a6, f6) CVTDN(a7, f7),
       CVTDN(a0, f0) f1 = f2 = f3 = f4 = f5 = f6 = f7 = f0;)
#define CVTI(d, r) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(d));
KERNEL(cvt_i32_f64, DECL_DN, CVTI(a0, f0) CVTI(a1, f1) CVTI(a2, f2) CVTI(a3, f3) CVTI(a4, f4) CVTI(a5, f5) CVTI(a6, f6) CVTI(a7, f7),
       CVTI(a0, f0) f1 = f2 = f3 = f4 = f5 = f6 = f7 = f0;)


#undef sink
#define sink asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7))
#define CNDS(r) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[2:3]" : "+v"(r) : "v"(b));
KERNEL(cndmask_sgpr, DECL_F32, CNDS(a0) CNDS(a1) CNDS(a2) CNDS(a3) CNDS(a4) CNDS(a5) CNDS(a6) CNDS(a7), CNDS(a0))
#define CNDZ(r) asm volatile("s_mov_b64 vcc, 0x5555\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b) : "vcc");
KERNEL(cndmask_vccinit, DECL_F32, CNDZ(a0) CNDZ(a1) CNDZ(a2) CNDZ(a3) CNDZ(a4) CNDZ(a5) CNDZ(a6) CNDZ(a7), CNDZ(a0))
#define DECL_I32 unsigned a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = a0 * 3
KERNEL(and_b32, DECL_I32, ALL8("v_and_b32"), OP1("v_and_b32", a0))
KERNEL(xor_b32, DECL_I32, ALL8("v_xor_b32"), OP1("v_xor_b32", a0))
KERNEL(add_u32, DECL_I32, ALL8("v_add_u32"), OP1("v_add_u32", a0))
KERNEL(lshlrev_b32, DECL_I32, ALL8("v_lshlrev_b32"), OP1("v_lshlrev_b32", a0))
#define BFI(r) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(r) : "v"(b));
KERNEL(bfi_b32, DECL_I32, BFI(a0) BFI(a1) BFI(a2) BFI(a3) BFI(a4) BFI(a5) BFI(a6) BFI(a7), BFI(a0))
#define CMPV(r) asm volatile("v_cmp_eq_u32 vcc, %0, %1" :: "v"(r), "v"(b) : "vcc");
KERNEL(cmp_eq_u32_vcc, DECL_I32, CMPV(a0) CMPV(a1) CMPV(a2) CMPV(a3) CMPV(a4) CMPV(a5) CMPV(a6) CMPV(a7), CMPV(a0))
#define CMPCND(r) asm volatile("v_cmp_eq_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b) : "vcc");
KERNEL(cmp_then_cndmask, DECL_I32, CMPCND(a0) CMPCND(a1) CMPCND(a2) CMPCND(a3) CMPCND(a4) CMPCND(a5) CMPCND(a6) CMPCND(a7), CMPCND(a0))
#define MOVV(r) asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(b));
KERNEL(mov_b32, DECL_I32, MOVV(a0) MOVV(a1) MOVV(a2) MOVV(a3) MOVV(a4) MOVV(a5) MOVV(a6) MOVV(a7), MOVV(a0))
#define NEGSEL(r) asm volatile("v_cndmask_b32_e64 %0, -%0, %0, vcc" : "+v"(r));
KERNEL(cndmask_neg, DECL_F32, NEGSEL(a0) NEGSEL(a1) NEGSEL(a2) NEGSEL(a3) NEGSEL(a4) NEGSEL(a5) NEGSEL(a6) NEGSEL(a7), NEGSEL(a0))


#undef sink
#define sink asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7))
#define CMPS(r) asm volatile("v_cmp_eq_u32_e64 s[2:3], %0, %1" :: "v"(r), "v"(b) : "s2", "s3");
KERNEL(cmp_eq_u32_sgpr, DECL_I32, CMPS(a0) CMPS(a1) CMPS(a2) CMPS(a3) CMPS(a4) CMPS(a5) CMPS(a6) CMPS(a7), CMPS(a0))
#define CMPSC(r) asm volatile("v_cmp_eq_u32_e64 s[2:3], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[2:3]" : "+v"(r) : "v"(b) : "s2", "s3");
KERNEL(cmp_sgpr_then_cnd_e64, DECL_I32, CMPSC(a0) CMPSC(a1) CMPSC(a2) CMPSC(a3) CMPSC(a4) CMPSC(a5) CMPSC(a6) CMPSC(a7), CMPSC(a0))
#define CMPVC64(r) asm volatile("v_cmp_eq_u32_e64 vcc, %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(r) : "v"(b) : "vcc");
KERNEL(cmp_vcc64_then_cnd_e64, DECL_I32, CMPVC64(a0) CMPVC64(a1) CMPVC64(a2) CMPVC64(a3) CMPVC64(a4) CMPVC64(a5) CMPVC64(a6) CMPVC64(a7), CMPVC64(a0))
#define MOV64(r) asm volatile("v_mov_b32_e64 %0, %1" : "=v"(r) : "v"(b));
KERNEL(mov_b32_e64, DECL_I32, MOV64(a0) MOV64(a1) MOV64(a2) MOV64(a3) MOV64(a4) MOV64(a5) MOV64(a6) MOV64(a7), MOV64(a0))
#define MOVI(r) asm volatile("v_mov_b32 %0, %0" : "+v"(r));
KERNEL(mov_b32_self, DECL_I32, MOVI(a0) MOVI(a1) MOVI(a2) MOVI(a3) MOVI(a4) MOVI(a5) MOVI(a6) MOVI(a7), MOVI(a0))
#define ANDO(r) asm volatile("v_and_b32 %0, %1, %1" : "=v"(r) : "v"(b));
KERNEL(and_b32_outonly, DECL_I32, ANDO(a0) ANDO(a1) ANDO(a2) ANDO(a3) ANDO(a4) ANDO(a5) ANDO(a6) ANDO(a7), ANDO(a0))
#undef sink
#define sink asm volatile("" ::"v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7))
#define CVTDN64(d, r) asm volatile("v_cvt_f32_f64_e64 %0, %1" : "=v"(r) : "v"(d));
KERNEL(cvt_f32_f64_e64, DECL_DN, CVTDN64(a0, f0) CVTDN64(a1, f1) CVTDN64(a2, f2) CVTDN64(a3, f3) CVTDN64(a4, f4) CVTDN64(a5, f5) CVTDN64(a6, f6) CVTDN64(a7, f7),
       CVTDN64(a0, f0) f1 = f2 = f3 = f4 = f5 = f6 = f7 = f0;)
#define MULDN(d, r) asm volatile("v_mul_f64 %0, %1, %1" : "=v"(d) : "v"(b));
#undef sink
#define sink asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7))
KERNEL(mul_f64_outonly, DECL_F64, MULDN(a0, 0) MULDN(a1, 0) MULDN(a2, 0) MULDN(a3, 0) MULDN(a4, 0) MULDN(a5, 0) MULDN(a6, 0) MULDN(a7, 0), MULDN(a0, 0))

typedef void (*kern_t)(long long*, int, float);
struct K { const char* name; kern_t k; int per_iter; };

int main() {
  const int iters = 2000;
  std::vector<K> ks = {
#define E(n, pi, pd) {#n " indep", n##_indep, pi}, {#n " dep", n##_dep, pd},
      E(mul_f32, 64, 64) E(fma_f32, 64, 64) E(pk_mul_f32, 64, 64) E(pk_add_f32, 64, 64) E(mul_f64, 64, 64) E(fma_f64, 64, 64)
      E(cndmask, 64, 64) E(rcp_f32, 64, 64) E(cvt_f64_f32_rt, 128, 128) E(cvt_f64_f32, 64, 64) E(cvt_f32_f64, 64, 64) E(cvt_i32_f64, 64, 64)
      E(cndmask_sgpr, 64, 64) E(cndmask_vccinit, 64, 64) E(cndmask_neg, 64, 64) E(and_b32, 64, 64) E(xor_b32, 64, 64) E(add_u32, 64, 64) E(lshlrev_b32, 64, 64) E(bfi_b32, 64, 64)
      E(cmp_eq_u32_sgpr, 64, 64) E(cmp_sgpr_then_cnd_e64, 64, 64) E(cmp_vcc64_then_cnd_e64, 64, 64) E(mov_b32_e64, 64, 64) E(mov_b32_self, 64, 64) E(and_b32_outonly, 64, 64) E(cvt_f32_f64_e64, 64, 64) E(mul_f64_outonly, 64, 64)
      E(cmp_eq_u32_vcc, 64, 64) E(cmp_then_cndmask, 64, 64) E(mov_b32, 64, 64)};
  long long* d;
  hipMalloc(&d, sizeof(long long) * 4096);
  std::vector<long long> h(4096);
  std::printf("%-24s %14s %14s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD");
  for (auto& k : ks) {
    double res[2];
    for (int w = 0; w < 2; ++w) {
      const int blocks = 1024 << w;
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, d, iters, 1.25f);
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, d, iters, 1.25f);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
      double s = 0;
      for (int i = 0; i < blocks; ++i) s += (double)h[i];
      res[w] = s / blocks / ((double)iters * k.per_iter);
    }
    std::printf("%-24s %14.2f %14.2f   (clock64 ticks per instruction per wave)\n", k.name, res[0], res[1]);
  }
  // clock64 rate: ticks per microsecond
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(mul_f32_dep, dim3(1024), dim3(64), 0, 0, d, 20000, 1.25f);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), d, sizeof(long long) * 1024, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
  std::printf("clock64: %.1f ticks/us (kernel %.3f ms, mean ticks %.0f)\n", s / 1024 / (ms * 1e3), ms, s / 1024);
  return 0;
}

// vmem_cost.hip — what does ONE vector-memory instruction cost the wave that issues it, next to VALU work?  (Round 5: the fused EKF
// launch is 10 % faster without its one 16-byte x-history store per step.)  Loop body = 160 v_pk_mul_f32 (independent; ~ the EKF step's VALU time) + one memory
// instruction (or none); cycles per iteration by clock64(), one wave per SIMD (1024 workgroups of 64) and four.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/vmem_cost.hip -o scripts/ubench/vmem_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define PK16 "v_pk_mul_f32 v[10:11], v[40:41], v[42:43]\n v_pk_mul_f32 v[12:13], v[40:41], v[42:43]\n v_pk_mul_f32 v[14:15], v[40:41], v[42:43]\n v_pk_mul_f32 v[16:17], v[40:41], v[42:43]\n" \
             "v_pk_mul_f32 v[18:19], v[40:41], v[42:43]\n v_pk_mul_f32 v[20:21], v[40:41], v[42:43]\n v_pk_mul_f32 v[22:23], v[40:41], v[42:43]\n v_pk_mul_f32 v[24:25], v[40:41], v[42:43]\n" \
             "v_pk_mul_f32 v[26:27], v[40:41], v[42:43]\n v_pk_mul_f32 v[28:29], v[40:41], v[42:43]\n v_pk_mul_f32 v[30:31], v[40:41], v[42:43]\n v_pk_mul_f32 v[32:33], v[40:41], v[42:43]\n" \
             "v_pk_mul_f32 v[34:35], v[40:41], v[42:43]\n v_pk_mul_f32 v[36:37], v[40:41], v[42:43]\n v_pk_mul_f32 v[38:39], v[40:41], v[42:43]\n v_pk_mul_f32 v[44:45], v[40:41], v[42:43]\n"
#define PK160 PK16 PK16 PK16 PK16 PK16 PK16 PK16 PK16 PK16 PK16
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v44","v45","v46","v47","v48","v49","v50","v51","memory"
// MODE: 0 none, 1 store x4 nt (advancing), 2 store x4 plain, 3 store x2 nt, 4 load x2 nt, 5 store x4 nt to the same line, 6 two loads x2 + store x4 (the EKF step's three)
template <int MODE>
__global__ void __launch_bounds__(64) k(long long* out, char* buf, size_t stride, int iters) {
  char* p = buf + ((size_t)blockIdx.x * 64 + threadIdx.x) * 16;
  asm volatile("v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0" ::: "v40","v41","v42","v43","v52","v53","v54","v55");
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 1) asm volatile(PK160 "global_store_dwordx4 %0, v[52:55], off nt\n s_waitcnt vmcnt(12)" :: "v"(p) : CLOB);
    else if (MODE == 2) asm volatile(PK160 "global_store_dwordx4 %0, v[52:55], off\n s_waitcnt vmcnt(12)" :: "v"(p) : CLOB);
    else if (MODE == 3) asm volatile(PK160 "global_store_dwordx2 %0, v[52:53], off nt\n s_waitcnt vmcnt(12)" :: "v"(p) : CLOB);
    else if (MODE == 4) asm volatile(PK160 "global_load_dwordx2 v[46:47], %0, off nt\n s_waitcnt vmcnt(12)" :: "v"(p) : CLOB);
    else if (MODE == 5) asm volatile(PK160 "global_store_dwordx4 %0, v[52:55], off nt\n s_waitcnt vmcnt(12)" :: "v"(buf + ((size_t)blockIdx.x * 64 + threadIdx.x) * 16) : CLOB);
    else if (MODE == 6) asm volatile(PK160 "global_load_dwordx2 v[46:47], %0, off nt\n global_load_dwordx2 v[48:49], %0, off offset:8 nt\n global_store_dwordx4 %0, v[52:55], off nt\n s_waitcnt vmcnt(24)" :: "v"(p) : CLOB);
    else asm volatile(PK160 ::: CLOB);
    p += stride;
  }
  asm volatile("s_waitcnt vmcnt(0)");
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
typedef void (*kern_t)(long long*, char*, size_t, int);
int main() {
  const int iters = 2000;
  long long* d; (void)hipMalloc(&d, sizeof(long long) * 65536);
  const char* names[] = {"160 pk_mul", "+ store x4 nt", "+ store x4 plain", "+ store x2 nt", "+ load x2 nt", "+ store x4 nt, same line", "+ 2 loads x2 + store x4 (the EKF step)"};
  kern_t ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
  std::printf("%-42s %16s %16s\n", "loop body (cycles per iteration per wave)", "1 wave/SIMD", "4 waves/SIMD");
  for (int m = 0; m < 7; ++m) {
    double res[2];
    for (int w = 0; w < 2; ++w) {
      const int blocks = w == 0 ? 1024 : 4096;
      const size_t stride = (size_t)blocks * 64 * 16;
      char* buf; (void)hipMalloc(&buf, stride * (size_t)(iters + 1));
      std::vector<long long> h(blocks);
      for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(ks[m], dim3(blocks), dim3(64), 0, 0, d, buf, stride, iters); (void)hipDeviceSynchronize(); }
      (void)hipMemcpy(h.data(), d, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : h) s += (double)v;
      res[w] = s / blocks / iters;
      (void)hipFree(buf);
    }
    std::printf("%-42s %16.1f %16.1f\n", names[m], res[0], res[1]);
  }
  return 0;
}

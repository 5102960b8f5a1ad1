#!/usr/bin/env python
"""Generates issue_patterns.hip: single-wave VALU issue cadence on gfx950 as a function of how an
instruction's destination relates to its sources (explicit VGPR numbers, one asm block per loop body)."""
pat = {}
def body(lines): return "\\n".join(lines)
R = range(8)
pat["A inplace      v_mul_f32 a_i, a_i, k"] = [f"v_mul_f32 v{10+i}, v{10+i}, v40" for i in R]
pat["B 3addr const  v_mul_f32 a_i, k, l"] = [f"v_mul_f32 v{10+i}, v40, v41" for i in R]
pat["C 3addr static v_mul_f32 a_i, s_i, k"] = [f"v_mul_f32 v{10+i}, v{20+i}, v40" for i in R]
pat["D read 4 back  v_mul_f32 a_i, a_(i+4), k"] = [f"v_mul_f32 v{10+i}, v{10+(i+4)%8}, v40" for i in R]
pat["E ping-pong    a<-s then s<-a"] = [f"v_mul_f32 v{10+i}, v{20+i}, v40" for i in R] + [f"v_mul_f32 v{20+i}, v{10+i}, v40" for i in R]
pat["F 3addr pk     v_pk_mul_f32 A_i, K, L"] = [f"v_pk_mul_f32 v[{10+2*i}:{11+2*i}], v[40:41], v[42:43]" for i in R]
pat["G 3addr fma64  v_fma_f64 A_i, K, L, M"] = [f"v_fma_f64 v[{10+2*i}:{11+2*i}], v[40:41], v[42:43], v[44:45]" for i in R]
pat["H inplace pk   v_pk_mul_f32 A_i, A_i, K"] = [f"v_pk_mul_f32 v[{10+2*i}:{11+2*i}], v[{10+2*i}:{11+2*i}], v[40:41]" for i in R]
pat["I 3addr pk st  v_pk_mul_f32 A_i, S_i, K"] = [f"v_pk_mul_f32 v[{10+2*i}:{11+2*i}], v[{46+2*i}:{47+2*i}], v[40:41]" for i in R]
pat["J pk ping-pong A<-S then S<-A"] = [f"v_pk_mul_f32 v[{10+2*i}:{11+2*i}], v[{46+2*i}:{47+2*i}], v[40:41]" for i in R] + [f"v_pk_mul_f32 v[{46+2*i}:{47+2*i}], v[{10+2*i}:{11+2*i}], v[40:41]" for i in R]
pat["K chain of 2   b=a*k; a=b*k (8 pairs)"] = sum([[f"v_mul_f32 v{20+i}, v{10+i}, v40", f"v_mul_f32 v{10+i}, v{20+i}, v40"] for i in R], [])
pat["L 2 chains     interleaved dep chains"] = [f"v_mul_f32 v{10+(i%2)}, v{10+(i%2)}, v40" for i in R]
pat["M 3 chains     interleaved dep chains"] = [f"v_mul_f32 v{10+(i%3)}, v{10+(i%3)}, v40" for i in range(9)]
pat["N dst=src1     v_mul_f32 a_i, k, a_i"] = [f"v_mul_f32 v{10+i}, v40, v{10+i}" for i in R]
pat["O fma dst=src2 v_fma_f32 a_i, k, l, a_i"] = [f"v_fma_f32 v{10+i}, v40, v41, v{10+i}" for i in R]
pat["P cvt inplace  v_cvt_f32_i32 a_i, a_i"] = [f"v_cvt_f32_i32 v{10+i}, v{10+i}" for i in R]
pat["Q cvt 3addr    v_cvt_f32_i32 a_i, s_i"] = [f"v_cvt_f32_i32 v{10+i}, v{20+i}" for i in R]
pat["R pk 3addr reading regs written 5 instrs ago"] = [f"v_pk_mul_f32 v[{10+2*i}:{11+2*i}], v[{10+2*((i+3)%8)}:{11+2*((i+3)%8)}], v[40:41]" for i in R]
pat["S cmp->sgpr x4 then cnd_e64 x4"] = [f"v_cmp_eq_u32_e64 s[{20+2*i}:{21+2*i}], v{10+i}, v40" for i in range(4)] + [f"v_cndmask_b32_e64 v{14+i}, v{14+i}, v40, s[{20+2*i}:{21+2*i}]" for i in range(4)]
pat["T bfe_i32+bfi  (mask select, no cmp)"] = sum([[f"v_bfe_i32 v{20+i}, v{10+i}, 0, 1", f"v_bfi_b32 v{10+i}, v{20+i}, v40, v{10+i}"] for i in range(4)], [])


pat["U cndmask e32 (vcc)          x8"] = [f"v_cndmask_b32 v{10+i}, v{10+i}, v40, vcc" for i in R]
pat["V cndmask e64 vcc            x8"] = [f"v_cndmask_b32_e64 v{10+i}, v{10+i}, v40, vcc" for i in R]
pat["W cndmask e64 sgpr           x8"] = [f"v_cndmask_b32_e64 v{10+i}, v{10+i}, v40, s[20:21]" for i in R]
pat["X cmp e32 ->vcc              x8"] = [f"v_cmp_eq_u32 vcc, v{10+i}, v40" for i in R]
pat["Y cmp e64 ->sgpr             x8"] = [f"v_cmp_eq_u32_e64 s[{20+2*(i%4)}:{21+2*(i%4)}], v{10+i}, v40" for i in R]
pat["Z cmp_f32 e64 |x| ->sgpr     x8"] = [f"v_cmp_lt_f32_e64 s[{20+2*(i%4)}:{21+2*(i%4)}], |v{10+i}|, v40" for i in R]
pat["a cmp e32->vcc, nop1, cnd e32 (compiler's idiom)"] = sum([[f"v_cmp_eq_u32 vcc, v{10+i}, v40", "s_nop 1", f"v_cndmask_b32 v{20+i}, v{20+i}, v41, vcc"] for i in R], [])
pat["b cmp e64->sgpr, nop1, cnd e64"] = sum([[f"v_cmp_eq_u32_e64 s[20:21], v{10+i}, v40", "s_nop 1", f"v_cndmask_b32_e64 v{20+i}, v{20+i}, v41, s[20:21]"] for i in R], [])
pat["c cvt_f64_f32 3addr          x8"] = [f"v_cvt_f64_f32 v[{10+2*i}:{11+2*i}], v{46+i}" for i in R]
pat["d cvt_f32_f64 3addr          x8"] = [f"v_cvt_f32_f64 v{46+i}, v[{10+2*i}:{11+2*i}]" for i in R]
pat["e cvt_i32_f64                x8"] = [f"v_cvt_i32_f64 v{46+i}, v[{10+2*i}:{11+2*i}]" for i in R]
pat["f cvt_f64_i32                x8"] = [f"v_cvt_f64_i32 v[{10+2*i}:{11+2*i}], v{46+i}" for i in R]
pat["g mul_f64 3addr              x8"] = [f"v_mul_f64 v[{10+2*i}:{11+2*i}], v[40:41], v[42:43]" for i in R]
pat["h fmac_f64 same acc twice (a+=k*l; a+=k*l)"] = sum([[f"v_fmac_f64 v[{10+2*i}:{11+2*i}], v[40:41], v[44:45]"]*2 for i in R], [])
pat["i fma_f64 chain-of-2 3addr"] = sum([[f"v_fma_f64 v[{26+2*i}:{27+2*i}], v[{10+2*i}:{11+2*i}], v[40:41], v[44:45]", f"v_fma_f64 v[{10+2*i}:{11+2*i}], v[{26+2*i}:{27+2*i}], v[40:41], v[44:45]"] for i in R], [])
pat["j rcp_f32                    x8"] = [f"v_rcp_f32 v{10+i}, v{20+i}" for i in R]
pat["k mov_b32                    x8"] = [f"v_mov_b32 v{10+i}, v{20+i}" for i in R]
pat["l mov_b32 dpp quad_perm swap x8"] = [f"v_mov_b32_dpp v{10+i}, v{20+i} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" for i in R]
pat["m mul_f32 dpp quad_perm      x8"] = [f"v_mul_f32_dpp v{10+i}, v{20+i}, v40 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" for i in R]
pat["n lshl_add_u64               x8"] = [f"v_lshl_add_u64 v[{10+2*i}:{11+2*i}], v[{26+2*i}:{27+2*i}], 0, s[20:21]" for i in R]
pat["o s_nop 0                    x8"] = ["s_nop 0" for i in R]
pat["p pk_mul + s_mul_i32 alternating (SALU co-issue?)"] = sum([[f"v_pk_mul_f32 v[{10+2*i}:{11+2*i}], v[40:41], v[42:43]", f"s_mul_i32 s22, s23, s24"] for i in R], [])
pat["q pk_mul x2 then xor_b32 x1 (mix)"] = sum([[f"v_pk_mul_f32 v[{10+2*i}:{11+2*i}], v[40:41], v[42:43]", f"v_pk_mul_f32 v[{26+2*i}:{27+2*i}], v[40:41], v[42:43]", f"v_xor_b32 v{46+i}, v{46+i}, v44"] for i in R], [])
pat["r bfe_i32                    x8"] = [f"v_bfe_i32 v{10+i}, v{20+i}, 0, 1" for i in R]
pat["s and_or_b32                 x8"] = [f"v_and_or_b32 v{10+i}, v{20+i}, v40, v41" for i in R]

clob = ",".join(f'"v{i}"' for i in range(10, 62)) + "," + ",".join(f'"s{i}"' for i in range(20, 28)) + ',"vcc"'
out = ['// generated by gen_issue_patterns.py — do not edit', '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>']
names = list(pat)
for k, name in enumerate(names):
    lines = pat[name]
    reps = max(1, 64 // len(lines))
    asm = body(lines * reps)
    out.append(f'''__global__ void __launch_bounds__(64) k{k}(long long* out, int iters) {{
  asm volatile("v_mov_b32 v40, 1.0\\n v_mov_b32 v41, 1.0\\n v_mov_b32 v42, 1.0\\n v_mov_b32 v43, 1.0\\n v_mov_b32 v44, 0\\n v_mov_b32 v45, 0" ::: {clob});
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) asm volatile("{asm}" ::: {clob});
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}}''')
out.append('typedef void (*kern_t)(long long*, int);\nstruct K { const char* name; kern_t k; int n; };\nint main() {\n  std::vector<K> ks = {')
for k, name in enumerate(names):
    lines = pat[name]; reps = max(1, 64 // len(lines))
    out.append(f'    {{"{name}", k{k}, {len(lines) * reps}}},')
out.append('''  };
  long long* d; (void)hipMalloc(&d, sizeof(long long) * 4096);
  std::vector<long long> h(4096);
  const int iters = 2000;
  std::printf("%-62s %12s %12s\\n", "pattern (cycles per instruction per wave)", "1 wave/SIMD", "2 waves/SIMD");
  for (auto& k : ks) {
    double res[2];
    for (int w = 0; w < 2; ++w) {
      const int blocks = 1024 << w;
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, d, iters);
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, d, iters);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), d, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
      res[w] = s / blocks / ((double)iters * k.n);
    }
    std::printf("%-62s %12.2f %12.2f\\n", k.name, res[0], res[1]);
  }
  return 0;
}''')
open("issue_patterns.hip", "w").write("\n".join(out) + "\n")

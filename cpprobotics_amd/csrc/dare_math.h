// dare_math.h — the Riccati iteration of the crx engine for A, B built from the speed (gfx950 device code; also compiles as
// plain host C++ so tests/tools/dare_host.cpp can check both forms below against the oracle on the CPU).
//
// Replaces the body of solve_DARE's loop
//   5x5, 2 inputs: /root/reference/src/lqr_speed_steer_control.cpp:85-100 with A, B of :116-126, Q = I5, R = I2 (:128-129)
//   4x4, 1 input : /root/reference/src/lqr_steer_control.cpp:75-90        with A, B of :104-112, Q = I4, R = 1  (:114-115)
// for ONE agent, in two register layouts:
//   * `*_v_iter_pk`  — the whole agent in one lane, rows of X as packed column pairs (v_pk_mul_f32 / v_pk_add_f32);
//   * `*_quad_iter`  — the agent spread over the four lanes of a DPP quad, lane r holding row r of the 4x4 block; the rows a
//                      lane needs from its neighbours arrive as DPP quad_perm operands of the multiplies themselves.
//
// Structure used (every step below is exact in IEEE arithmetic, so the coefficients equal the reference's dense Eigen
// evaluation bit for bit as long as the iterates are finite; tests: tests/test_dare_host.py, tests/test_lqr_gpu.py):
//   1. literal 0 / 1 entries of A and B are skipped (x*1 = x, x*0 = +-0, s + (+-0) = s for s != 0);
//   2. 5x5 only: A = diag(A4, 1) and B = [b e3, dt e4] never couple state 4 (the speed error) with states 0-3, so with
//      X0 = Q = I every iterate is block diagonal, X = diag(X4, x44), its off-block entries exact zeros: all of them come
//      out of the closing `+ Q` as +0, and every product they enter is a zero that is added to a non-zero or to another
//      zero.  The 2x2 matrix R + B'XB is then diagonal, det = m0*m3 - 0*0 = m0*m3 and its inverse diag(m3/det, m0/det)
//      with the reference's own rounding (one division 1/det, two multiplies).  The two blocks stay coupled through that
//      shared rounding, which is why the 4x4 block cannot simply reuse the one-input iteration.
//   A zero of the reference's evaluation can come out as a zero of the other sign in an intermediate; it can never reach a
//   returned value with its sign (the iterate is normalised by `+ Q`, the gains by their own sums), and the parity bar
//   compares IEEE values.  Once an iterate overflows, skipped 0*inf products make the non-finite patterns differ (as for
//   rule 1 already): an agent whose reference run turns non-finite is non-finite here too, not necessarily entry by entry.
#pragma once
#include <stdint.h>
#include "crx_trig.h"   // CRX_HD

namespace crx {

#if defined(__clang__)
typedef float d_v2f __attribute__((ext_vector_type(2)));
#else
typedef float d_v2f __attribute__((vector_size(8)));
#endif

CRX_HD d_v2f dbc2(float x) { return d_v2f{x, x}; }

struct Row4 { d_v2f a, b; };            // columns (0,1), (2,3) of one row

// ---------- one lane per agent ----------------------------------------------------------------------------------------------
// A'X for A = [1 dt 0 0; 0 0 v 0; 0 0 1 dt; 0 0 0 0] (both files), row by row
CRX_HD void dare_v_AtX(float dt, float v, const Row4* X, Row4* R) {
  R[0] = X[0];
  R[1].a = dbc2(dt) * X[0].a; R[1].b = dbc2(dt) * X[0].b;
  R[2].a = X[2].a + dbc2(v) * X[1].a; R[2].b = X[2].b + dbc2(v) * X[1].b;
  R[3].a = dbc2(dt) * X[2].a; R[3].b = dbc2(dt) * X[2].b;
}

// row i of (A'XA - (c33 * X3)A) + Q for the 4x4 block, given row i of A'X, the row's factor c33 and row 3 of X.
// (M*A) columns: 0 = M.0, 1 = M.0*dt, 2 = M.1*v + M.2, 3 = M.2*dt;  x*1.0f = x and x + (-0.0f) = x bit for bit
template <int I>
CRX_HD void dare_v_row(float dt, float v, const Row4& Ri, float c33, const Row4& X3, Row4& Xn) {
  Row4 C;
  C.a = dbc2(c33) * X3.a; C.b = dbc2(c33) * X3.b;
  const d_v2f vdt = {v, dt}, one_dt = {1.0f, dt};
  const d_v2f p1a = dbc2(Ri.a[0]) * one_dt, p2a = dbc2(C.a[0]) * one_dt;
  const d_v2f p1b = d_v2f{Ri.a[1], Ri.b[0]} * vdt + d_v2f{Ri.b[0], -0.0f};
  const d_v2f p2b = d_v2f{C.a[1], C.b[0]} * vdt + d_v2f{C.b[0], -0.0f};
  Xn.a = (p1a - p2a) + d_v2f{I == 0 ? 1.0f : 0.0f, I == 1 ? 1.0f : 0.0f};
  Xn.b = (p1b - p2b) + d_v2f{I == 2 ? 1.0f : 0.0f, I == 3 ? 1.0f : 0.0f};
}

// 4x4 (:81): Xn = A'XA - ((A'X B / (R + B'XB)) B'X) A + Q, B = bv e3, R = 1
CRX_HD void dare4_v_iter_pk(float dt, float v, float bv, const Row4* X, Row4* Xn) {
  Row4 R[4];
  dare_v_AtX(dt, v, X, R);
  const float g = (bv * X[3].b[1]) * bv;
  const float s = 1.0f + g;
  dare_v_row<0>(dt, v, R[0], ((R[0].b[1] * bv) / s) * bv, X[3], Xn[0]);
  dare_v_row<1>(dt, v, R[1], ((R[1].b[1] * bv) / s) * bv, X[3], Xn[1]);
  dare_v_row<2>(dt, v, R[2], ((R[2].b[1] * bv) / s) * bv, X[3], Xn[2]);
  dare_v_row<3>(dt, v, R[3], ((R[3].b[1] * bv) / s) * bv, X[3], Xn[3]);
}

// the diagonal of inverse2(R + B'XB) for the block-diagonal iterate (header, rule 2): m0 = 1 + (bv X33) bv, m3 = 1 + (bd x44) bd
CRX_HD void dare5_v_sinv(float bv, float bd, float x33, float x44, float& Si0, float& Si3) {
  const float m0 = 1.0f + (bv * x33) * bv, m3 = 1.0f + (bd * x44) * bd;
  const float det = m0 * m3;              // m0*m3 - m1*m2 with m1 = m2 = +0
  const float invdet = 1.0f / det;
  Si0 = m3 * invdet; Si3 = m0 * invdet;
}
// the 1x1 block: x44' = (x44 - ((((x44 bd) Si3) bd) x44)) + 1
CRX_HD float dare5_v_x44(float bd, float Si3, float x44) {
  return (x44 - (((x44 * bd) * Si3) * bd) * x44) + 1.0f;
}

// 5x5 (:91), X = diag(X4, x44)
CRX_HD void dare5_v_iter_pk(float dt, float v, float bv, float bd, const Row4* X, float x44, Row4* Xn, float& x44n) {
  Row4 R[4];
  dare_v_AtX(dt, v, X, R);
  float Si0, Si3;
  dare5_v_sinv(bv, bd, X[3].b[1], x44, Si0, Si3);
  dare_v_row<0>(dt, v, R[0], ((R[0].b[1] * bv) * Si0) * bv, X[3], Xn[0]);
  dare_v_row<1>(dt, v, R[1], ((R[1].b[1] * bv) * Si0) * bv, X[3], Xn[1]);
  dare_v_row<2>(dt, v, R[2], ((R[2].b[1] * bv) * Si0) * bv, X[3], Xn[2]);
  dare_v_row<3>(dt, v, R[3], ((R[3].b[1] * bv) * Si0) * bv, X[3], Xn[3]);
  x44n = dare5_v_x44(bd, Si3, x44);
}

// (Xn - X).cwiseAbs().maxCoeff() (:92): a strict '>' scan from element (0,0) in which a NaN element never replaces the
// running maximum and a NaN FIRST element sticks.  fmaxf drops a NaN operand, which is that behaviour for every element
// but the first; the first element's NaN is patched in.  Exact zeros (the off-block entries) cannot raise a maximum.
CRX_HD float dare_max2(float m, d_v2f d) { return __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(d[0])), __builtin_fabsf(d[1])); }
CRX_HD float dare_max_abs_diff(const Row4* Y, const Row4* X) {
  const d_v2f d0 = Y[0].a - X[0].a;
  const float m0 = __builtin_fabsf(d0[0]);
  float m = __builtin_fmaxf(m0, __builtin_fabsf(d0[1]));
  m = dare_max2(m, Y[0].b - X[0].b);
#pragma unroll
  for (int i = 1; i < 4; ++i) { m = dare_max2(m, Y[i].a - X[i].a); m = dare_max2(m, Y[i].b - X[i].b); }
  return (m0 != m0) ? m0 : m;
}
CRX_HD float dare_max_abs_diff(const Row4* Y, float y44, const Row4* X, float x44) {
  const float m = dare_max_abs_diff(Y, X);
  return (m != m) ? m : __builtin_fmaxf(m, __builtin_fabsf(y44 - x44));
}

// ---------- four lanes per agent ---------------------------------------------------------------------------------------------
// F is `float` in the kernel (one value per lane) and Quad4f, four lanes in lockstep, in the host build.  qperm<CTRL>(x) reads x
// from the lane of the same quad that the DPP quad_perm control names (CTRL = p0 | p1<<2 | p2<<4 | p3<<6: lane i reads lane
// p_i) — written so that the compiler folds the move into the consuming v_mul_f32 / v_max_f32 / v_add_f32 as its DPP operand;
// qand(x, m) is a bitwise AND of x's bits with a per-lane mask.
constexpr int QP_0012 = 0 | (0 << 2) | (1 << 4) | (2 << 6);   // lane r reads row max(r-1, 0): the source row of (A'X) row r
constexpr int QP_3333 = 0xFF;                                 // everyone reads lane 3 (row 3 of X)
constexpr int QP_0000 = 0x00;
constexpr int QP_1032 = 1 | (0 << 2) | (3 << 4) | (2 << 6);
constexpr int QP_2301 = 2 | (3 << 2) | (0 << 4) | (1 << 6);

#if defined(__HIPCC__)
template <int CTRL>
__device__ __forceinline__ float qperm(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float qand(float x, uint32_t m) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & m); }
__device__ __forceinline__ float qfabs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ float qfmax(float a, float b) { return __builtin_fmaxf(a, b); }
#else
struct Quad4f {
  float l[4];
  Quad4f() : l{0, 0, 0, 0} {}
  Quad4f(float s) : l{s, s, s, s} {}
  Quad4f(float a, float b, float c, float d) : l{a, b, c, d} {}
};
struct Quad4u { uint32_t l[4]; };
#define CRX_Q4_OP(op) \
  static inline Quad4f operator op(const Quad4f& a, const Quad4f& b) { Quad4f r; for (int i = 0; i < 4; ++i) r.l[i] = a.l[i] op b.l[i]; return r; }
CRX_Q4_OP(+) CRX_Q4_OP(-) CRX_Q4_OP(*) CRX_Q4_OP(/)
#undef CRX_Q4_OP
template <int CTRL>
static inline Quad4f qperm(const Quad4f& x) { Quad4f r; for (int i = 0; i < 4; ++i) r.l[i] = x.l[(CTRL >> (2 * i)) & 3]; return r; }
static inline Quad4f qand(const Quad4f& x, const Quad4u& m) {
  Quad4f r;
  for (int i = 0; i < 4; ++i) { uint32_t b; __builtin_memcpy(&b, &x.l[i], 4); b &= m.l[i]; __builtin_memcpy(&r.l[i], &b, 4); }
  return r;
}
static inline Quad4f qfabs(const Quad4f& x) { Quad4f r; for (int i = 0; i < 4; ++i) r.l[i] = __builtin_fabsf(x.l[i]); return r; }
static inline Quad4f qfmax(const Quad4f& a, const Quad4f& b) { Quad4f r; for (int i = 0; i < 4; ++i) r.l[i] = __builtin_fmaxf(a.l[i], b.l[i]); return r; }
#endif

template <class F, class M>
struct QuadLane {   // what distinguishes the four lanes of an agent: loop-invariant registers
  F a;              // row r of A'X = a * X[src] (+ X[2] on lane 2):  a = (1, dt, v, dt), src = (0, 0, 1, 2)
  M m2;             // all ones on lane 2, zero elsewhere
  F q[4];           // row r of Q = I
  F dt, v, bv, bd;
};

// row r of A'X on lane r.  Lanes other than 2 add +0.0f, which turns a -0 product into +0 — a zero either way (header).
template <class F, class M>
CRX_HD void dare_quad_AtX(const QuadLane<F, M>& c, const F* x, F* R) {
#pragma unroll
  for (int j = 0; j < 4; ++j) R[j] = c.a * qperm<QP_0012>(x[j]) + qand(x[j], c.m2);
}
// row r of (A'XA - (c33 X3)A) + Q on lane r
template <class F, class M>
CRX_HD void dare_quad_row(const QuadLane<F, M>& c, const F* R, F c33, const F* x, F* xn) {
  F C[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) C[j] = c33 * qperm<QP_3333>(x[j]);
  const F p1[4] = {R[0], R[0] * c.dt, R[1] * c.v + R[2], R[2] * c.dt};
  const F p2[4] = {C[0], C[0] * c.dt, C[1] * c.v + C[2], C[2] * c.dt};
#pragma unroll
  for (int j = 0; j < 4; ++j) xn[j] = (p1[j] - p2[j]) + c.q[j];
}
// max |xn - x| over the agent's four rows (every lane of the quad gets it); NaN elements are dropped (see dare_quad_first)
template <class F>
CRX_HD F dare_quad_maxdiff(const F* xn, const F* x) {
  F m = qfabs(xn[0] - x[0]);
#pragma unroll
  for (int j = 1; j < 4; ++j) m = qfmax(m, qfabs(xn[j] - x[j]));
  m = qfmax(m, qperm<QP_1032>(m));
  return qfmax(m, qperm<QP_2301>(m));
}
// the first-element rule of dare_max_abs_diff as an addend: with d00 the difference of element (0,0) — on lane 0 —
// (d00 - d00) is 0 for a finite d00 and NaN for a NaN or infinite one (where the maximum is not below eps either)
template <class F>
CRX_HD F dare_quad_first(const F* xn, const F* x) {
  const F d0 = xn[0] - x[0];
  return qperm<QP_0000>(d0 - d0);
}

// One evaluation for the 4x4 problem; returns max |xn - x| with the reference's NaN rule.
template <class F, class M>
CRX_HD F dare4_quad_iter(const QuadLane<F, M>& c, const F* x, F* xn) {
  F R[4];
  dare_quad_AtX(c, x, R);
  const F g = (c.bv * qperm<QP_3333>(x[3])) * c.bv;
  const F s = F(1.0f) + g;
  dare_quad_row(c, R, ((R[3] * c.bv) / s) * c.bv, x, xn);
  return dare_quad_maxdiff(xn, x) + dare_quad_first(xn, x);
}

// One evaluation for the 5x5 problem, X = diag(X4, x44) (x44 replicated on the four lanes).
template <class F, class M>
CRX_HD F dare5_quad_iter(const QuadLane<F, M>& c, const F* x, F x44, F* xn, F& x44n) {
  F R[4];
  dare_quad_AtX(c, x, R);
  const F m0 = F(1.0f) + (c.bv * qperm<QP_3333>(x[3])) * c.bv, m3 = F(1.0f) + (c.bd * x44) * c.bd;
  const F det = m0 * m3;
  const F invdet = F(1.0f) / det;
  const F Si0 = m3 * invdet, Si3 = m0 * invdet;
  dare_quad_row(c, R, ((R[3] * c.bv) * Si0) * c.bv, x, xn);
  x44n = (x44 - (((x44 * c.bd) * Si3) * c.bd) * x44) + F(1.0f);
  return qfmax(dare_quad_maxdiff(xn, x), qfabs(x44n - x44)) + dare_quad_first(xn, x);
}

// ---------- the same two evaluations as the kernels issue them ----------------------------------------------------------------------
// Same operations on the same operands, entry by entry, as dare4_quad_iter / dare5_quad_iter above (the host build and the tests keep
// using those); what differs is how they are handed to the machine:
//   * lane 2's extra term of A'X is added under an exec mask of the lanes 2 (mod 4) (lanes 0, 1, 3 keep a * X[src] — the generic code adds +0.0f there,
//     which can only turn a -0 into +0, and every zero's sign is gone after `(p1 - p2) + q`);
//   * the two halves of independent scalar chains — (bv X33 bv, bd x44 bd), (m3, m0) / det, (R3 bv Si0 bv, x44 bd Si3 bd) — and the
//     pairs of a row ride in packed fp32 instructions; products by the literal 1.0f are exact;
//   * the quad maximum takes its partner lane as the DPP operand of v_max_f32 (fmaxf semantics: a NaN operand is dropped).
// 58 instead of 77 VALU instructions per evaluation (65 instead of 87 issue slots with the loop of dare_kernels.hip.h); tests/test_lqr_gpu.py and tests/test_track_gpu.py hold the kernels to the
// oracle's bits (iteration counts included), tests/test_dare_host.py the generic code above.
#if defined(__HIPCC__)
// The inline assembly below is written for wave64 on the gfx9 family's DPP / exec-mask rules (s_and_saveexec_b64 over a 64-bit lane
// mask, quad_perm DPP operands, hand-counted s_nop wait states).  The Makefile's ARCH can be overridden: refuse any target this
// was not written and verified for rather than assemble something that runs wrong without a diagnostic.
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(__gfx950__) && !defined(__gfx942__)
#error "dare_math.h: the hand-issued quad Riccati evaluation targets gfx950 (gfx942 shares its DPP hazard rules); other targets need the generic dare*_quad_iter"
#endif
// (gfx942 / gfx950 execute wave64 only — no wave32 mode exists on the gfx9 family, so the architecture test is the wave-size test)
#endif
typedef float dq_v2f __attribute__((ext_vector_type(2)));

// acc[j] + src[j] on lane 2 of every quad, acc[j] elsewhere: the four adds under an exec mask of the lanes 2 (mod 4)
__device__ __forceinline__ void dq_add_lane2(float& a0, float& a1, float& a2, float& a3, float s0, float s1, float s2, float s3) {
  unsigned long long saved;
  asm("s_and_saveexec_b64 %4, %9\n\t"
      "v_add_f32_e32 %0, %0, %5\n\t"
      "v_add_f32_e32 %1, %1, %6\n\t"
      "v_add_f32_e32 %2, %2, %7\n\t"
      "v_add_f32_e32 %3, %3, %8\n\t"
      "s_mov_b64 exec, %4"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(saved)
      : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "s"(0x4444444444444444ull)
      : "scc");
}
// max(m, m of the lane CTRL names); m was written by the instruction before: two wait states ahead of the DPP read
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ float dq_max_perm(float m) {
  float r;
  if constexpr (P0 == 1) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(m));
  else asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(m));
  return r;
}

struct DqRows { float R0; dq_v2f R12; float R3; };
__device__ __forceinline__ DqRows dq_AtX(const QuadLane<float, uint32_t>& c, const float* x) {
  DqRows o;
  float r0 = c.a * qperm<QP_0012>(x[0]), r1 = c.a * qperm<QP_0012>(x[1]), r2 = c.a * qperm<QP_0012>(x[2]), r3 = c.a * qperm<QP_0012>(x[3]);
  dq_add_lane2(r0, r1, r2, r3, x[0], x[1], x[2], x[3]);
  o.R0 = r0; o.R12[0] = r1; o.R12[1] = r2; o.R3 = r3;
  return o;
}
// row r of (A'XA - (c33 X3)A) + Q, and max |xn - x| over the row
__device__ __forceinline__ float dq_row(const QuadLane<float, uint32_t>& c, const DqRows& R, float c33, const float* x, float* xn) {
  const float C0 = c33 * qperm<QP_3333>(x[0]);
  dq_v2f C12;
  C12[0] = c33 * qperm<QP_3333>(x[1]);
  C12[1] = c33 * qperm<QP_3333>(x[2]);
  const dq_v2f one_dt = {1.0f, c.dt}, v_dt = {c.v, c.dt};
  const dq_v2f p1a = (dq_v2f){R.R0, R.R0} * one_dt;                    // (R0, R0 dt)
  const dq_v2f p2a = (dq_v2f){C0, C0} * one_dt;
  dq_v2f p1b = R.R12 * v_dt, p2b = C12 * v_dt;                         // (R1 v, R2 dt)
  p1b[0] = p1b[0] + R.R12[1];                                          // R1 v + R2
  p2b[0] = p2b[0] + C12[1];
  const dq_v2f xa = (p1a - p2a) + (dq_v2f){c.q[0], c.q[1]};
  const dq_v2f xb = (p1b - p2b) + (dq_v2f){c.q[2], c.q[3]};
  const dq_v2f da = xa - (dq_v2f){x[0], x[1]}, db = xb - (dq_v2f){x[2], x[3]};
  xn[0] = xa[0]; xn[1] = xa[1]; xn[2] = xb[0]; xn[3] = xb[1];
  const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(da[0]), __builtin_fabsf(da[1])), __builtin_fmaxf(__builtin_fabsf(db[0]), __builtin_fabsf(db[1])));
  return m;
}

// lane-local part of an evaluation: the row's maximum |xn - x| (before the quad reduction) and the first-element term
struct DqTest { float m, first; };   // first: (d00 - d00) on every lane from its own row — lane 0's is the one that counts
__device__ __forceinline__ DqTest dare4_quad_eval_dev(const QuadLane<float, uint32_t>& c, const float* x, float* xn) {
  const DqRows R = dq_AtX(c, x);
  const float g = (c.bv * qperm<QP_3333>(x[3])) * c.bv;
  const float s = 1.0f + g;
  DqTest t;
  t.m = dq_row(c, R, ((R.R3 * c.bv) / s) * c.bv, x, xn);
  const float d0 = xn[0] - x[0];
  t.first = d0 - d0;
  return t;
}
// the quad maximum of one evaluation's test; and four tests at once — maxima and first-element terms, the four chains interleaved: no DPP
// read follows its source's write by less than three instructions, so only the first one needs wait states (f1..f4 are written before the block)
__device__ __forceinline__ float dq_quad_max(float m) { return dq_max_perm<2, 3, 0, 1>(dq_max_perm<1, 0, 3, 2>(m)); }
__device__ __forceinline__ void dq_quad_test4(float& m1, float& m2, float& m3, float& m4, float f1, float f2, float f3, float f4) {
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %4, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"       // + lane 0's first-element term
      "v_add_f32_dpp %1, %5, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %6, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %7, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf"
      : "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4) : "v"(f1), "v"(f2), "v"(f3), "v"(f4));
}
__device__ __forceinline__ float dare4_quad_iter_dev(const QuadLane<float, uint32_t>& c, const float* x, float* xn) {
  const DqTest t = dare4_quad_eval_dev(c, x, xn);
  return dq_quad_max(t.m) + qperm<QP_0000>(t.first);
}
__device__ __forceinline__ DqTest dare5_quad_eval_dev(const QuadLane<float, uint32_t>& c, const float* x, float x44, float* xn, float& x44n) {
  const DqRows R = dq_AtX(c, x);
  const dq_v2f bvd = {c.bv, c.bd};
  dq_v2f t;
  t[0] = c.bv * qperm<QP_3333>(x[3]);
  t[1] = c.bd * x44;
  const dq_v2f m03 = (dq_v2f){1.0f, 1.0f} + t * bvd;                   // (m0, m3)
  const float det = m03[0] * m03[1];
  const float invdet = 1.0f / det;
  const dq_v2f Si = (dq_v2f){m03[1], m03[0]} * (dq_v2f){invdet, invdet};  // (Si0, Si3) = (m3, m0) / det
  dq_v2f u;
  u[0] = R.R3 * c.bv;
  u[1] = x44 * c.bd;
  u = (u * Si) * bvd;                                                  // (c33, ((x44 bd) Si3) bd)
  DqTest o;
  o.m = dq_row(c, R, u[0], x, xn);
  x44n = (x44 - u[1] * x44) + 1.0f;
  const float d0 = xn[0] - x[0];
  o.first = d0 - d0;
  o.m = __builtin_fmaxf(o.m, __builtin_fabsf(x44n - x44));
  return o;
}
__device__ __forceinline__ float dare5_quad_iter_dev(const QuadLane<float, uint32_t>& c, const float* x, float x44, float* xn, float& x44n) {
  const DqTest t = dare5_quad_eval_dev(c, x, x44, xn, x44n);
  return dq_quad_max(t.m) + qperm<QP_0000>(t.first);
}
#endif

}  // namespace crx

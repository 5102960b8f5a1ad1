// mpc_kernels.hip.h — batched speed+steer MPC horizon solve for gfx950 (one agent per lane, fp64).
//
// Replaces, for n independent agents at once, mpc_solve() of the reference
//   (/root/reference/src/model_predictive_control.cpp:255-346), i.e. the NLP that FG_EVAL (:199-252)
// defines: variables [x|y|yaw|v|delta|a], bicycle-model equality constraints (:242-245), input
// and input-rate costs (:203-209), tracking cost (:247-250), box bounds on steering, acceleration and speed (:288-301).
// The reference hands that NLP to CppAD+IPOPT (double precision); crx solves the same NLP with a
// solver of its own, built for one-problem-per-lane execution:
//
//   * single shooting — the equality constraints are eliminated by rolling the model forward,
//     leaving the 2(T-1) controls as unknowns with their box bounds; the speed bounds of the knots
//     become a state-dependent box on the acceleration (v+ = v + a*DT), enforced exactly in every
//     rollout and carried through the backward sweep as a feedback row where they are active;
//   * stage-wise Newton (control-limited DDP): a backward Riccati sweep over the T-1 stages on the
//     state augmented by the previous control (the input-rate cost couples consecutive controls),
//     exact second derivatives of the dynamics once Gauss-Newton steps have brought the iterate
//     close, a 2-variable box QP per stage solved in closed form by enumerating its candidate
//     minimisers, and a backtracking forward rollout;
//   * all arithmetic in fp64 (IPOPT's precision), result rounded to float like the reference's
//     `(float)solution.x[i]` (:343).
//
// The structure of the stage matrices is used throughout: with s = (x,y,yaw,v | d_prev,a_prev),
//   F_s = [A 0; 0 0],  A = I + {a02,a03,a12,a13,a23},   F_u = [B; I],  B = {b_delta at (2,0), dt at (3,1)}
// so no dense 6x6 product is ever formed.  oracle/mpc_ref.cpp is the plainly written dense CPU
// twin of this algorithm; tests require agreement to 1e-6 and check optimality against scipy.
//
// Memory: trajectories, feed-forward and feedback gains of the lane's problem live in private
// (scratch) memory, which the hardware interleaves across the 64 lanes, i.e. every access below
// is a coalesced 512-byte wave access served from L1/L2; inputs are read once, the solution is
// written once.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/crx.h"
#include "mpc_agpr.inc"      // the accumulator-register store of the tile layout's feedback gains (generated: scripts/gen_mpc_agpr.py)

#define CRX_MPC_MAX_T 64

// The library is built with -ffp-contract=off because the fp32 EKF / DARE / tracking kernels reproduce the reference's
// unfused mul-then-add arithmetic bit for bit.  The MPC solver is fp64 with tolerance-based parity (the reference's own
// answer is an IPOPT iterate) and wants its multiply-adds fused — but rounds 1-4 got them from `#pragma clang fp contract(fast)`,
// and which product of a sum of products the compiler fuses turned out to depend on the code AROUND the sum: two builds of this
// header that differ only in where the backward sweep's trig comes from disagreed in the last bits of the cost (round 5).
// Since round 5 every fused multiply-add of the solver is WRITTEN (fma(), nested so that a row a*b + c*d + e is two instructions,
// not three) and nothing is left to contraction: every build of mpc_solve_lane computes the same bits.

#ifndef CRX_EXPERIMENTAL_KERNELS
#define CRX_EXPERIMENTAL_KERNELS 0
#endif

namespace crx {

// Selects.  The compiler's idiom for `c ? a : b` on doubles is v_cmp -> VCC and two VOP2 v_cndmask_b32_e32 reading VCC
// implicitly; back-to-back runs of that encoding issue at ~19 cycles per instruction on gfx950, the VOP3 encoding with the
// mask in an SGPR pair at 5.5 (profiles/r01/ubench_issue_patterns.txt, rows U/V/W), and the solver's candidate scoring and
// quadrant logic select several values on one condition.  sel64 emits the VOP3 form (measured: -3 % on the BASELINE solve).
typedef unsigned long long lanemask_t;
__device__ __forceinline__ lanemask_t lanes_where(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ int sel32(lanemask_t m, int a, int b) {   // lane's bit of m set ? a : b
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ double sel64(lanemask_t m, double a, double b) {
  return __hiloint2double(sel32(m, __double2hiint(a), __double2hiint(b)), sel32(m, __double2loint(a), __double2loint(b)));
}
__device__ __forceinline__ double sel64(bool c, double a, double b) { return sel64(lanes_where(c), a, b); }
__device__ __forceinline__ double flip_sign_if(double x, int bit) {   // bit = 0 or 1: x or -x, one shift and one xor
  return __hiloint2double(__double2hiint(x) ^ (bit << 31), __double2loint(x));
}

// sin, cos (and tan = sin/cos) in fp64 for the solver.  OCML's sincos()/tan() carry a full Payne-Hanek reduction and
// cost ~150 instructions each; the solver calls them once per stage per rollout.  For |x| < 2^17 a three-term
// Cody-Waite reduction (fma) and the fdlibm kernel polynomials give <= 1 ulp in ~45 instructions; larger arguments
// (never produced by a sane course) fall back to OCML.  MPC parity is tolerance-based (1e-6 vs the CPU twin, which
// uses the host libm), so last-ulp differences are immaterial.
__device__ __forceinline__ void mpc_sincos(double x, double* sp, double* cp) {
  if (!(fabs(x) < 131072.0)) { sincos(x, sp, cp); return; }
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632679489655800e+00, pio2_mid = 6.12323399573676603587e-17, pio2_lo = -1.49738490485916983693e-33;
  const double fn = rint(x * two_over_pi);
  double r = fma(-fn, pio2_hi, x);
  r = fma(-fn, pio2_mid, r);
  r = fma(-fn, pio2_lo, r);
  const int q = (int)fn;
  const double z = r * r;
  // __kernel_sin / __kernel_cos (fdlibm k_sin.c, k_cos.c), without their extra-precision tails
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                              2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double sr = fma(r * z, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                              -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                               -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const lanemask_t odd = lanes_where((q & 1) != 0);
  const double s0 = sel64(odd, cr, sr);
  const double c0 = sel64(odd, sr, cr);
  *sp = flip_sign_if(s0, (q >> 1) & 1);
  *cp = flip_sign_if(c0, ((q + 1) >> 1) & 1);
}
// n / d for a divisor of ordinary magnitude (cos of a steering angle; the 1x1 / 2x2 pivots of a stage QP): hardware
// reciprocal estimate, two Newton steps, one residual correction of the quotient — <= 1 ulp in 8 instructions, where the
// IEEE division sequence (scaling for subnormals and overflow, which cannot occur here) takes ~35.
__device__ __forceinline__ double fast_div(double n, double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  const double q = n * r;
  return fma(fma(-d, q, n), r, q);
}
__device__ __forceinline__ double mpc_tan(double x) {
  double sn, cs;
  mpc_sincos(x, &sn, &cs);
  return fast_div(sn, cs);
}
// tan of a steering angle known to lie in [-pi/4, pi/4] (MAX_STEER = 45 deg, :288-291): the reduction of mpc_sincos is the
// identity there (quadrant 0, r = x: rint(0.5) = 0), so this is mpc_tan without it — the same bits.
__device__ __forceinline__ double mpc_tan_small(double x) {
  const double z = x * x;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                              2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double sr = fma(x * z, ps, x);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                              -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                               -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  return fast_div(sr, cr);
}

struct MpcP {
  double dt, wb, max_steer, max_accel, max_speed, min_speed;
  double r_a, r_d, rd_a, rd_d, qx, qy, qyaw, qv, tol;
  int max_iter;
};

// lo <= hi at every call site; equal to v < lo ? lo : (v > hi ? hi : v) up to the sign of a zero (v_max_f64 + v_min_f64)
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// Trust box of a Newton step (see the backward sweep) — the same constants as the twin, oracle/mpc_ref.cpp.
constexpr double kMpcTrustSteer = 0.4, kMpcTrustAccel = 0.5;

// The reference bounds every speed knot, v in [MIN_SPEED, MAX_SPEED] (:298-301).  With v+ = v + a*DT (:245) that is exactly
// a state-dependent box on the acceleration at a knot of speed v, intersected with |a| <= MAX_ACCEL (:293-296); the
// acceleration limits win if the two are incompatible (a start speed outside the speed bounds; reported in status bit 1).
struct AccelBox { double lo, hi; bool sp_lo, sp_hi; };
__device__ __forceinline__ AccelBox accel_box(const MpcP& p, double inv_dt, double v) {
  const double a_lo = (p.min_speed - v) * inv_dt, a_hi = (p.max_speed - v) * inv_dt;
  AccelBox b;
  b.sp_lo = a_lo > -p.max_accel; b.sp_hi = a_hi < p.max_accel;
  b.lo = clampd(a_lo, -p.max_accel, p.max_accel);
  b.hi = clampd(a_hi, -p.max_accel, p.max_accel);
  return b;
}

// min 1/2 k'Hk + g'k over the box, H = [h00 hod; hod h11] possibly indefinite.  See file header.
__device__ __forceinline__ void boxqp2(double h00, double hod, double h11, double g0, double g1, double lo0,
                                       double hi0, double lo1, double hi1, double& k0, double& k1, bool& f0,
                                       bool& f1) {
  // Straight-line: every candidate is formed and scored, the winner is picked by selects in the order (and with the
  // strict comparison) of oracle/mpc_ref.cpp — per-lane booleans carried through divergent branches cost more scalar
  // mask bookkeeping than the arithmetic they skip.  Whole candidate sets are skipped only when NO lane of the wave needs them
  // (every lane interior; every lane on the edge rule) — which set a lane's answer comes from is decided by its own problem.
  const double tiny = 1e-12;
  const double det = fma(h00, h11, -(hod * hod));
  const bool pd = h00 > tiny && det > tiny * h00;
  const double idet = fast_div(1.0, pd ? det : 1.0);
  const double ia = -(fma(h11, g0, -(hod * g1)) * idet), ib = -(fma(h00, g1, -(hod * g0)) * idet);
  const bool interior = pd && ia >= lo0 && ia <= hi0 && ib >= lo1 && ib <= hi1;
  k0 = ia; k1 = ib; f0 = true; f1 = true;
  if (__all(interior)) return;
  double best = 1e300, b0 = 0.0, b1 = 0.0;
  int bf = 0;                                  // bit 0: control 0 free, bit 1: control 1 free
  const double hh00 = 0.5 * h00, hh11 = 0.5 * h11;
  auto consider = [&](double a, double b, int flags, bool valid) {
    const double obj = fma(a, fma(hh00, a, fma(hod, b, g0)), b * fma(hh11, b, g1));      // 1/2 k'Hk + g'k
    const lanemask_t take = lanes_where(valid && obj < best);
    best = sel64(take, obj, best); b0 = sel64(take, a, b0); b1 = sel64(take, b, b1); bf = sel32(take, flags, bf);
  };
  const bool c11 = h11 > tiny, c00 = h00 > tiny;
  const double ih11 = fast_div(1.0, c11 ? h11 : 1.0), ih00 = fast_div(1.0, c00 ? h00 : 1.0);   // one reciprocal per edge pair
  // Both diagonal curvatures positive (always so in a Gauss-Newton sweep): the problem along each of the four edges is a convex
  // parabola, whose minimiser over the edge is the stationary point clamped to it — four candidates cover the whole boundary,
  // corners included.  A control counts as free under the rule of the full enumeration below (its unclamped stationary value lies
  // in the closed interval), so the two rules agree on the point and on the flags except for exact ties between two DIFFERENT
  // boundary points of equal objective.  The rule is a property of the LANE's problem (round 4; the CPU twin applies
  // it per problem too): a wave in which some lane needs the enumeration runs both and every lane keeps the result of its own
  // rule, so an agent's answer never depends on which agents share its wave.
  const bool edge_rule = c00 && c11;
  const bool all_edge = __all(edge_rule);      // the common case keeps its mask-free candidate scoring
  auto edge_candidates = [&](auto valid) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const double c0 = b ? hi0 : lo0;
      const double u1 = -(fma(hod, c0, g1) * ih11);
      consider(c0, clampd(u1, lo1, hi1), (u1 >= lo1 && u1 <= hi1) ? 2 : 0, valid());
      const double c1 = b ? hi1 : lo1;
      const double u0 = -(fma(hod, c1, g0) * ih00);
      consider(clampd(u0, lo0, hi0), c1, (u0 >= lo0 && u0 <= hi0) ? 1 : 0, valid());
    }
  };
  if (all_edge) edge_candidates([] { return true; });
  else edge_candidates([&] { return edge_rule; });
  if (!all_edge) {      // rare: an indefinite or semidefinite stage Hessian somewhere in the wave (Newton sweeps only)
    const bool en = !edge_rule;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const double c0 = b ? hi0 : lo0;
      const double t1 = -(fma(hod, c0, g1) * ih11);
      consider(c0, t1, 2, en && c11 && t1 >= lo1 && t1 <= hi1);
      const double c1 = b ? hi1 : lo1;
      const double t0 = -(fma(hod, c1, g0) * ih00);
      consider(t0, c1, 1, en && c00 && t0 >= lo0 && t0 <= hi0);
    }
#pragma unroll
    for (int q0 = 0; q0 < 2; ++q0)
#pragma unroll
      for (int q1 = 0; q1 < 2; ++q1) consider(q0 ? hi0 : lo0, q1 ? hi1 : lo1, 0, en);
  }
  const lanemask_t in = lanes_where(interior);
  k0 = sel64(in, ia, b0); k1 = sel64(in, ib, b1);
  f0 = interior || (bf & 1); f1 = interior || (bf & 2);
}

// The variants of the PORTFOLIO solve (crx_mpc_solve_portfolio_batch_dev; round 4).  The BASELINE batch leaves 7/8 of the SIMDs idle
// and its launch lasts as long as its slowest agent's chain of sweeps; different globalisation settings have different stragglers
// (profiles/r04/mpc_experiments.txt).  In the portfolio an agent is a quad of lanes, lane r solving the SAME problem with variant r
// — the number of leading Gauss-Newton sweeps and the size of a Newton step's trust box — in lockstep; the first variant to converge
// (fewest sweeps; ties: lowest r) is the agent's answer and stops its siblings.  Variant 0 is the engine's solver: an agent never
// needs more sweeps than crx_mpc_solve_batch_dev gives it.  Deterministic, and reproducible by the CPU twin (oracle_mpc_solve_portfolio),
// which runs the four variants one after the other.
struct MpcVariant { int n_gn; double trust_scale; };
__device__ __forceinline__ MpcVariant mpc_variant(int r) {
  return MpcVariant{r == 1 ? 3 : (r == 3 ? 1 : 2), r >= 2 ? 2.0 : 1.0};   // (2, 1) the engine's | (3, 1) | (2, 2) | (1, 2)
}

// One lane's solve.  xi = (x, y, yaw, v) of x0; xr4 = the lane's reference trajectory, T columns (x, y, yaw, v) — global memory
// in mpc_kernel, the lane's private array in the closed loop; so (may be null) receives the solution in the reference's layout;
// a0 / d0 = the first acceleration and steering of the solution rounded to float, what mpc_simulation applies (:376).
// Wave-synchronous: every lane of the wave must call it (finished / padding lanes with live = false).
// PORTFOLIO: the four lanes of a quad solve the same problem with mpc_variant(lane & 3); `so` and the outputs are valid on the
// winning lane only (status_out < 0 on the others).
// REFILL (mpc_refill_kernel, the throughput regime): the wave owns the agents [feed.lo, feed.hi); a lane whose agent is done holds it
// until feed.hold lanes of the wave hold one (or nobody is sweeping), then they write their solutions and take the next agents of the
// range — cursor + rank among the lanes asking, as dare_from_v_refill_kernel does.  The arguments live, xi, xr4, so and the four
// outputs are unused then: problems come from and results go to the feed's arrays.  Per agent the same sweeps in the same order.
struct MpcFeed {
  int lo, hi, hold;
  const float* __restrict__ x0g; const float* __restrict__ xrefg;
  float* __restrict__ solg; int* __restrict__ statusg; double* __restrict__ costg;
};
// LEAN: where the backward sweep's trig comes from.  false: the rollout keeps sin(yaw_i), cos(yaw_i), tan(delta_i) for it (TR: 24 B
// written per rollout stage, 24 B read per backward stage).  true: the backward sweep recomputes them from the knot — the same functions
// of the same doubles, and since round 5 spelled every fused multiply-add of the solver out, THE SAME BITS in every output (solution,
// status, cost: scripts/gpu_mpc_two_builds.py and tests/test_mpc_gpu.py compare them on the device) — ~70 more VALU instructions per
// backward stage for 11 % less memory traffic (96 -> 82 KB per solve).  Recomputing wins where the solver's traffic is HBM traffic
// (lone launches: 131,072 agents 4.36 -> 4.15 ms, 262,144 6.24 -> 5.49, 1 M 21.2 -> 17.6 = 59 M solves/s) and loses 5-6 % where a launch
// is a latency chain (BASELINE configs[3]: 0.88 -> 0.93 ms; the persistent closed loop); 65,536 agents: equal
// (profiles/r05/mpc_two_builds.jsonl).  The launcher picks it from kMpcLeanFrom agents on; the answer does not depend on the choice.
constexpr int kMpcLeanFrom = 98304;
// ... and from this many when the caller says the launch shares the GPU with others (crx_mpc_params.shared_gpu): alone, a 16,384-agent launch is
// 6 % faster with the trig stored (2.66 vs 2.83 ms); six of them in flight next to the EKF launches of a configs[4] round are 3 % faster
// recomputing it (0.481 -> 0.466 ms per round, profiles/r05/swarm_shared_gpu_hint_ab.txt)
constexpr int kMpcLeanFromShared = 16384;
// STORE: where the lane's working set lives.
//   0  private memory, all of it (mpc_kernel, the portfolio, the closed loop; every horizon up to CRX_MPC_MAX_T)
//   1  the TILE layout of mpc_tile_kernels.hip.h (round 6; horizons of at most kMpcTileStages stages): both control buffers in the
//      wave's 40 KB of LDS, the float feedback gains of stages 1 .. 18 in accumulator registers a40 .. a255 (mpc_agpr.inc; stage 0's
//      gains multiply dx = 0 and are never stored; stage 19's stay private), knots and feed-forward steps still private.  The same operations on the same
//      doubles in the same order as STORE = 0: bit-identical outputs (tests/test_mpc_gpu.py compares the two on the device).
//   2  the CHECKPOINTED tile layout (round 6, second step): STORE = 1 with the knots cut to ONE buffer of every SECOND knot.
//        * a candidate rollout no longer reads the accepted trajectory: it re-rolls it from the accepted controls beside the candidate
//          (x_cur(i+1) = step(x_cur(i), U[cur][i]) — the function of the doubles that produced the stored knot, hence the stored bits)
//          and writes its own even knots straight into the one buffer; a line search that fails at every step length re-rolls the
//          accepted controls once to restore it;
//        * the backward sweep takes the stages in pairs: the odd knot 2m+1 is one model step from the even knot 2m, and the trig that
//          step needs is the trig stage 2m needs anyway (LEAN recomputes it) — the odd knots cost four fused multiply-adds each.
//      Private memory left: 10 knots + 20 feed-forward steps = 656 B per lane (3.8 KB in STORE 0, 1.9 KB in STORE 1) + what the fenced
//      register allocator spills (~100 registers).  Again the same operations on the same doubles in the same order: bit-identical
//      outputs.  MEASURED AND NOT SELECTED: 0.70-0.77x of STORE 1 at every batch size (1 M agents 14.7 -> 21.1 ms) — the second
//      sincos / tan per rollout stage and the spills cost more than the traffic saved, the tile kernel not being bandwidth-bound
//      (profiles/r06/mpc_store_ab*.jsonl: `tile2`).  Reachable through crx_x_mpc_solve_store_dev(store = 2).
constexpr int kMpcTileStages = 20;                     // T <= 21: the BASELINE horizon (configs[3], configs[4]) and the reference's own T 6
typedef double mpc_d2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) mpc_d2_t lds_double2_t;
struct MpcTile { lds_double2_t* u; };                  // the wave's [2][kMpcTileStages][64] double2 of LDS: (delta, a) of buffer c, stage i, lane l
// PHASED (round 6; mpc_phase_kernel below): the lockstep solve cut at sweep boundaries.  ph.cap > 0: a lane still sweeping when the
// wave reaches sweep index ph.cap SUSPENDS — its solver state (J, mu, the Gauss-Newton counters, status, the accepted controls as
// doubles) goes to ph.st and status_out is kMpcSuspended; ph.resume >= 0: the lane does not start from the zero guess but RESUMES
// such a record at sweep index ph.resume (the knots, and the stored trig, are re-rolled from the controls: the function of the
// doubles that produced them, hence their bits).  A suspended-and-resumed agent runs exactly the sweeps of an uninterrupted one.
constexpr int kMpcSuspended = -1;
struct MpcPhase { int cap = 0; int resume = -1; double* st = nullptr; };
__host__ __device__ inline int mpc_phase_record_doubles(int T) { return 6 + 2 * (T - 1); }
template <int MAXT, bool PORTFOLIO = false, bool REFILL = false, bool LEAN = false, int STORE = 0, bool PHASED = false>
__device__ __forceinline__ void mpc_solve_lane(const bool live_in, const int T, const float4 xi_in, const float4* __restrict__ xr4_in, const MpcP& p,
                                               float* __restrict__ so_in, int& status_out, double& cost_out, float& a0_out, float& d0_out,
                                               const MpcFeed feed = MpcFeed{}, const MpcTile tile = MpcTile{}, const MpcPhase ph = MpcPhase{}) {
  static_assert(!(PORTFOLIO && REFILL), "the portfolio runs in the latency regime");
  static_assert(!PHASED || (!PORTFOLIO && !REFILL && STORE <= 1), "the phased solve is the lockstep solve (private-memory or tile layout), cut at sweep boundaries");
  constexpr bool TILE = STORE >= 1;
  constexpr bool CKPT = STORE == 2;
  // gain slots in accumulator registers: 18 (a40 .. a255) — or 7 (a40 .. a123) in the LITE tile layout (STORE = 3), whose wave then
  // takes 384 of its SIMD's 512 registers and leaves room for a 128-register wave of another kernel beside it
  constexpr int kSlots = STORE == 3 ? CRX_MPC_AGPR_SLOTS_LITE : CRX_MPC_AGPR_SLOTS;
  static_assert(!TILE || (LEAN && MAXT <= kMpcTileStages + 4), "the tile layout recomputes the trig and holds at most kMpcTileStages stages");
  bool live = live_in;
  float4 xi = xi_in;
  const float4* __restrict__ xr4 = xr4_in;
  float* __restrict__ so = so_in;
  const int N = T - 1;
  const int quad_lane = (int)(threadIdx.x & 3);
  const MpcVariant var = PORTFOLIO ? mpc_variant(quad_lane) : MpcVariant{2, 1.0};
  const double trust_steer = kMpcTrustSteer * var.trust_scale, trust_accel = kMpcTrustAccel * var.trust_scale;

  // per-lane problem storage (private memory)
  double S[CKPT ? 1 : 2][CKPT ? kMpcTileStages / 2 : MAXT][4];   // knots: x, y, yaw, v        (two buffers: accepted / candidate; CKPT: ONE buffer, S[0][j] = knot 2(j+1))
  double U[TILE ? 1 : 2][TILE ? 1 : MAXT][2];   // stages: delta, a   (TILE: in LDS, tile.u)
  double kf[TILE ? kMpcTileStages : MAXT][2];     // feed-forward
  // feedback, K[a + 2*b], b over (x,y,yaw,v,d_prev,a_prev) — kept in FLOAT (round 5): the gains only steer the candidate rollouts
  // (u + alpha k + K dx), the fixed point is decided by the feed-forward k (double) alone, and they are 41 % of the solver's memory
  // traffic as doubles.  The twin rounds them the same way (oracle/mpc_ref.cpp); on 4 x 8,192 problems the sweep counts of 2 agents move by
  // one and no float of any solution by more than one ulp (profiles/r05/mpc_experiments.txt).
  float Kf[TILE ? (kMpcTileStages > kSlots + 1 ? kMpcTileStages - kSlots - 1 : 1) : MAXT][12];   // (TILE: stages 1 .. kSlots in accumulator registers)
  // ---- accessors of the controls and the gains (the only places that know where they live) --------------------------------------
  const int tile_lane = (int)(threadIdx.x & 63);
  auto ldU = [&](int c, int i, double& d, double& a) {
    if constexpr (TILE) { const mpc_d2_t v = tile.u[(c * kMpcTileStages + i) * 64 + tile_lane]; d = v.x; a = v.y; }
    else { d = U[c][i][0]; a = U[c][i][1]; }
  };
  auto stU = [&](int c, int i, double d, double a) {
    if constexpr (TILE) tile.u[(c * kMpcTileStages + i) * 64 + tile_lane] = mpc_d2_t{d, a};
    else { U[c][i][0] = d; U[c][i][1] = a; }
  };
  auto stKf = [&](int i, const float (&g)[12]) {        // i: wave-uniform (the stage loops run in lockstep)
    if constexpr (TILE) {
      if (i > kSlots) {
#pragma unroll
        for (int a = 0; a < 12; ++a) Kf[i - kSlots - 1][a] = g[a];
      } else if (i >= 1) mpc_agpr_store12(i - 1, g);
    } else {
#pragma unroll
      for (int a = 0; a < 12; ++a) Kf[i][a] = g[a];
    }
  };
  auto ldKf = [&](int i, float (&g)[12]) {
    if constexpr (TILE) {
      if (i > kSlots) {
#pragma unroll
        for (int a = 0; a < 12; ++a) g[a] = Kf[i - kSlots - 1][a];
      } else mpc_agpr_load12(i - 1, g);                 // stage 0 (slot -1): zeros — its gains multiply dx = 0
    } else {
#pragma unroll
      for (int a = 0; a < 12; ++a) g[a] = Kf[i][a];
    }
  };
  double TR[LEAN ? 1 : 2][LEAN ? 1 : MAXT][3];  // sin(yaw_i), cos(yaw_i), tan(delta_i) of each rollout: the backward sweep reuses them (LEAN: recomputes them)

  const double dt = p.dt, wb = p.wb;
  const double dt_wb = dt / wb;          // the model uses .../wb*dt once per stage and rollout: one division per solve instead
  const double inv_dt = 1.0 / dt;
  const double lb0 = -p.max_steer, ub0 = p.max_steer;
  const bool small_steer = p.max_steer <= 0.78539816339744830962;   // uniform: every steering angle of a rollout is clamped to it
  const double c2r_d = 2.0 * p.r_d, c2r_a = 2.0 * p.r_a, c2rd_d = 2.0 * p.rd_d, c2rd_a = 2.0 * p.rd_a;    // the doubled weights of the
  const double c2qx = 2.0 * p.qx, c2qy = 2.0 * p.qy, c2qyaw = 2.0 * p.qyaw, c2qv = 2.0 * p.qv;             // cost's second derivatives

  // objective of buffer c (states already rolled out there) is accumulated while rolling; this lambda
  // rolls controls U[c] from x0 and returns fg[0]
  // The arithmetic of a rollout — model step, control and tracking cost, the feedback law — is written with explicit fma(): under
  // `fp contract(fast)` the compiler decides per build which product of a sum of products it fuses, and the candidate's cost J decides
  // accept / reject.  With the fusion spelled out every build of this function (lockstep, refilling, portfolio, the closed loop)
  // rolls the same bits out of the same gains.
  auto track_cost = [&](const float4 r, const double* s) -> double {
    const double e0 = (double)r.x - s[0], e1 = (double)r.y - s[1], e2 = (double)r.z - s[2], e3 = (double)r.w - s[3];
    return fma(p.qx * e0, e0, fma(p.qy * e1, e1, fma(p.qyaw * e2, e2, (p.qv * e3) * e3)));
  };
  auto ctrl_cost = [&](bool inner, double d, double a, double pd, double pa) -> double {
    double v = fma(p.r_d * d, d, (p.r_a * a) * a);
    if (inner) {
      const double dd = d - pd, da = a - pa;
      v = fma(p.rd_d * dd, dd, fma(p.rd_a * da, da, v));
    }
    return v;
  };
  auto track = [&](const double* s, int i) -> double { return track_cost(xr4[i], s); };
  // the trig of a stage (knot yaw, steering d) and the model step given it: step() = the two in a row, for every layout
  auto trig3 = [&](double yaw, double d, double& sn_, double& cs_, double& tn_) {
    mpc_sincos(yaw, &sn_, &cs_);
    tn_ = small_steer ? mpc_tan_small(d) : mpc_tan(d);
  };
  auto step_tr = [&](const double* s, double a, double sn_, double cs_, double tn_, double* sn) {
    sn[0] = fma(s[3] * cs_, dt, s[0]);
    sn[1] = fma(s[3] * sn_, dt, s[1]);
    sn[2] = fma(s[3] * tn_, dt_wb, s[2]);
    sn[3] = fma(a, dt, s[3]);
  };
  auto step = [&](const double* s, double d, double a, double* sn, double* tr) {
    double sn_, cs_, tn_;
    trig3(s[2], d, sn_, cs_, tn_);
    if constexpr (!LEAN) { tr[0] = sn_; tr[1] = cs_; tr[2] = tn_; }
    step_tr(s, a, sn_, cs_, tn_, sn);
  };
  // CKPT: the even knot j (wave-uniform, 0 <= j <= N): knot 0 is the start state, knot 2(m+1) is S[0][m].  The load is unconditional
  // (clamped slot) so that the number of loads in flight does not depend on j.
  auto knot_even = [&](int j, double (&e)[4]) {
    if constexpr (CKPT) {
      const int m = j >= 2 ? (j >> 1) - 1 : 0;
      const bool first = j == 0;
      e[0] = first ? (double)xi.x : S[0][m][0];
      e[1] = first ? (double)xi.y : S[0][m][1];
      e[2] = first ? (double)xi.z : S[0][m][2];
      e[3] = first ? (double)xi.w : S[0][m][3];
    }
  };
  auto put_knot = [&](int j, const double* x) {      // CKPT: knot j of the trajectory being rolled, kept if it is an even one
    if constexpr (CKPT) {
      if (j >= 2 && !(j & 1)) {
        const int m = (j >> 1) - 1;
        S[0][m][0] = x[0]; S[0][m][1] = x[1]; S[0][m][2] = x[2]; S[0][m][3] = x[3];
      }
    }
  };

  struct StageIn { double s0, s1, s2, s3, sn, cs, tn; float4 r; };
  auto load_stage = [&](int c, int i) -> StageIn {
    if constexpr (CKPT) return StageIn{};      // (unused: the checkpointed sweep has its own loads)
    else if constexpr (LEAN) return StageIn{S[c][i][0], S[c][i][1], S[c][i][2], S[c][i][3], 0.0, 0.0, 0.0, xr4[i]};
    else return StageIn{S[c][i][0], S[c][i][1], S[c][i][2], S[c][i][3], TR[c][i][0], TR[c][i][1], TR[c][i][2], xr4[i]};
  };

  struct RollIn { double s[4], u0, u1, k0, k1, K[12]; float4 r; };
  auto load_roll = [&](int c, int i) -> RollIn {
    RollIn q;
    if constexpr (!CKPT) {                     // CKPT: the rollout re-rolls the accepted trajectory itself
#pragma unroll
      for (int a = 0; a < 4; ++a) q.s[a] = S[c][i][a];
    }
    ldU(c, i, q.u0, q.u1); q.k0 = kf[i][0]; q.k1 = kf[i][1];
    float g[12];
    ldKf(i, g);
#pragma unroll
    for (int a = 0; a < 12; ++a) q.K[a] = (double)g[a];
    q.r = xr4[i];
    return q;
  };

  float4 rN = REFILL ? float4{0.f, 0.f, 0.f, 0.f} : xr4[N];   // terminal reference: used by every sweep and every rollout

  int cur = 0;
  double J = 0.0;
  double mu = 0.0;
  const double mu_min = 1e-6, mu_max = 1e10;
  const int n_gn = var.n_gn;
  int gn_left = n_gn, gn_run = n_gn;
  int status = 0, it = 0;
  // the start of a solve from (xi, xr4): zero initial guess (:266-269), rolled out; projected on the acceleration box of each knot
  // (which moves it only if the start speed violates the speed bounds)
  auto start = [&]() {
    if constexpr (CKPT) {
      cur = 0;
      double xs[4] = {(double)xi.x, (double)xi.y, (double)xi.z, (double)xi.w};
      J = 0.0;
      double a_prev = 0.0;
      for (int i = 0; i < N; ++i) {
        const AccelBox ab = accel_box(p, inv_dt, xs[3]);
        const double a0 = clampd(0.0, ab.lo, ab.hi);
        stU(0, i, 0.0, a0);
        J += ctrl_cost(i >= 1, 0.0, a0, 0.0, a_prev);
        a_prev = a0;
        if (i >= 1) J += track(xs, i);
        double xn[4];
        step(xs, 0.0, a0, xn, nullptr);
        put_knot(i + 1, xn);
        xs[0] = xn[0]; xs[1] = xn[1]; xs[2] = xn[2]; xs[3] = xn[3];
      }
      J += track(xs, N);
    } else {
      cur = 0;
      S[0][0][0] = S[1][0][0] = (double)xi.x;
      S[0][0][1] = S[1][0][1] = (double)xi.y;
      S[0][0][2] = S[1][0][2] = (double)xi.z;
      S[0][0][3] = S[1][0][3] = (double)xi.w;
      J = 0.0;
      double a_prev = 0.0;
      for (int i = 0; i < N; ++i) {
        const AccelBox ab = accel_box(p, inv_dt, S[0][i][3]);
        const double a0 = clampd(0.0, ab.lo, ab.hi);
        stU(0, i, 0.0, a0);
        J += ctrl_cost(i >= 1, 0.0, a0, 0.0, a_prev);      // the controls of stages i, i - 1
        a_prev = a0;
        if (i >= 1) J += track(S[0][i], i);
        step(S[0][i], 0.0, a0, S[0][i + 1], LEAN ? TR[0][0] : TR[0][i]);
      }
      J += track(S[0][N], N);
    }
    mu = 0.0; gn_left = n_gn; gn_run = n_gn; status = 0; it = 0;
  };
  // CKPT: the one knot buffer holds a refused candidate — re-roll the accepted controls into it (the same function of the same
  // doubles as when they were accepted: the same bits)
  auto restore_knots = [&]() {
    if constexpr (CKPT) {
      double xc[4] = {(double)xi.x, (double)xi.y, (double)xi.z, (double)xi.w};
      for (int i = 0; i < N; ++i) {
        double ud_, ua_, xn[4];
        ldU(cur, i, ud_, ua_);
        step(xc, ud_, ua_, xn, nullptr);
        put_knot(i + 1, xn);
        xc[0] = xn[0]; xc[1] = xn[1]; xc[2] = xn[2]; xc[3] = xn[3];
      }
    }
  };
  bool resumed = false;
  if constexpr (PHASED) {
    if (ph.resume >= 0) {                      // wave-uniform: every lane of a resumed wave was suspended at the same sweep
      resumed = true;
      cur = 0;
      J = ph.st[0]; mu = ph.st[1]; gn_left = (int)ph.st[2]; gn_run = (int)ph.st[3]; status = (int)ph.st[4]; it = ph.resume;
      S[0][0][0] = S[1][0][0] = (double)xi.x; S[0][0][1] = S[1][0][1] = (double)xi.y;
      S[0][0][2] = S[1][0][2] = (double)xi.z; S[0][0][3] = S[1][0][3] = (double)xi.w;
      for (int i = 0; i < N; ++i) {
        const double ud_ = ph.st[6 + 2 * i], ua_ = ph.st[7 + 2 * i];
        stU(0, i, ud_, ua_);
        step(S[0][i], ud_, ua_, S[0][i + 1], LEAN ? TR[0][0] : TR[0][i]);
      }
    }
  }
  if (!REFILL && !resumed) start();
  bool done = REFILL ? true : !live;
  bool suspended = false;
  // REFILL bookkeeping: the lane's agent (-1: none; a done lane with an agent holds a finished solve), the trip of the loop below at
  // which that agent started, the wave's cursor into its range, the lanes asking for an agent
  int agent_l = -1, next = feed.lo;
  lanemask_t want = ~lanemask_t(0);
  const size_t nv_ = 4 * (size_t)T + 2 * ((size_t)T - 1);
  // the agent's answer in the reference's layout (:341-345) and the speed-bound check of every knot
  auto write_solution = [&](float* __restrict__ so_) {
    if constexpr (CKPT) {                          // the knots of the accepted controls, re-rolled
      double xc[4] = {(double)xi.x, (double)xi.y, (double)xi.z, (double)xi.w};
      for (int i = 0; i < T; ++i) {
        const double v = xc[3];
        if (v > p.max_speed + 1e-9 || v < p.min_speed - 1e-9) status |= 2;
        if (so_) {
          so_[i] = (float)xc[0];
          so_[T + i] = (float)xc[1];
          so_[2 * T + i] = (float)xc[2];
          so_[3 * T + i] = (float)v;
        }
        if (i < N) {
          double ud_, ua_, xn[4];
          ldU(cur, i, ud_, ua_);
          step(xc, ud_, ua_, xn, nullptr);
          xc[0] = xn[0]; xc[1] = xn[1]; xc[2] = xn[2]; xc[3] = xn[3];
        }
      }
    } else {
      for (int i = 0; i < T; ++i) {
        const double v = S[cur][i][3];
        if (v > p.max_speed + 1e-9 || v < p.min_speed - 1e-9) status |= 2;
        if (so_) {
          so_[i] = (float)S[cur][i][0];
          so_[T + i] = (float)S[cur][i][1];
          so_[2 * T + i] = (float)S[cur][i][2];
          so_[3 * T + i] = (float)v;
        }
      }
    }
    if (so_)
      for (int i = 0; i < N; ++i) {
        double ud_, ua_;
        ldU(cur, i, ud_, ua_);
        so_[4 * T + i] = (float)ud_;
        so_[4 * T + N + i] = (float)ua_;
      }
  };
  auto quad_converged = [&]() -> unsigned {       // which lanes of this lane's quad have converged (bit r = variant r)
    const lanemask_t cm = lanes_where(live && (status & 1));
    return (unsigned)(cm >> (threadIdx.x & 60)) & 0xFu;
  };

#ifdef CRX_MPC_TICKS
  long long tk_b = 0, tk_f = 0, tk_nb = 0, tk_nf = 0; const long long tk_0 = clock64();
#endif
  // ---- the two halves of a sweep ----------------------------------------------------------------------------------------------
  double dV1 = 0.0, dV2 = 0.0, gnorm = 0.0;          // results of backward(): expected change of the cost, largest feed-forward step
  // backward(exact): the Riccati sweep over the stages N-1 .. 0 around the accepted iterate (buffer `cur`); writes the gains kf, Kf
  auto backward = [&](const bool exact) {
    // ------------------------------------------------------------------ backward sweep
    double lx[4], lp0, lp1;          // V_s
    double Wxx[4][4], Wxp[4][2], Wpp00, Wpp01, Wpp11;  // V_ss (symmetric)
    auto terminal = [&](const double* s) {     // V_s, V_ss of the terminal cost at knot N
      const float4 r = rN;
      lx[0] = c2qx * (s[0] - (double)r.x);
      lx[1] = c2qy * (s[1] - (double)r.y);
      lx[2] = c2qyaw * (s[2] - (double)r.z);
      lx[3] = c2qv * (s[3] - (double)r.w);
      lp0 = 0.0; lp1 = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) Wxx[a][b] = 0.0;
        Wxp[a][0] = 0.0; Wxp[a][1] = 0.0;
      }
      Wxx[0][0] = c2qx; Wxx[1][1] = c2qy; Wxx[2][2] = c2qyaw; Wxx[3][3] = c2qv;
      Wpp00 = 0.0; Wpp01 = 0.0; Wpp11 = 0.0;
    };
    dV1 = 0.0; dV2 = 0.0; gnorm = 0.0;
    // The sweep's operands live in private memory (L2 / HBM latency once only a few waves are still iterating): stage
    // i - 1's knot, trig and reference and stage i - 2's control are requested at the top of stage i and consumed one
    // iteration later.
    double uc0, uc1, up0, up1;                 // the controls of the stage about to be swept and of the one below it
    ldU(cur, N - 1, uc0, uc1);
    ldU(cur, N >= 2 ? N - 2 : 0, up0, up1);
#if CRX_MPC_TICKS >= 2
    const long long tk_b0 = clock64();
#endif
    // stage i of the sweep, given its knot s, the trig of (s.yaw, ud), the reference row, its control (ud, ua) and the control below (pd, pa)
    auto stage = [&](const int i, const double s_0, const double s_1, const double s_2, const double s_3, const double sn_, const double cs_,
                     const double tn, const float4 r_in, const double ud, const double ua, const double pd, const double pa) {
      const bool inner = i >= 1;
      const double s[4] = {s_0, s_1, s_2, s_3};
      const double v = s[3];
      const double sec2 = fma(tn, tn, 1.0);
      const double vdt = v * dt, vdw = v * dt_wb;
      const double a02 = -(vdt * sn_), a03 = cs_ * dt, a12 = vdt * cs_, a13 = sn_ * dt, a23 = tn * dt_wb;
      const double bd = vdw * sec2;
      // stage cost derivatives
      double l_x[4] = {0.0, 0.0, 0.0, 0.0}, q2[4] = {0.0, 0.0, 0.0, 0.0};
      double l_u0 = c2r_d * ud, l_u1 = c2r_a * ua;
      double l_uu0 = c2r_d, l_uu1 = c2r_a;
      double l_p0 = 0.0, l_p1 = 0.0, l_pp0 = 0.0, l_pp1 = 0.0, l_up0 = 0.0, l_up1 = 0.0;
      if (inner) {
        const float4 r = r_in;
        q2[0] = c2qx; q2[1] = c2qy; q2[2] = c2qyaw; q2[3] = c2qv;
        l_x[0] = c2qx * (s[0] - (double)r.x);
        l_x[1] = c2qy * (s[1] - (double)r.y);
        l_x[2] = c2qyaw * (s[2] - (double)r.z);
        l_x[3] = c2qv * (s[3] - (double)r.w);
        const double dd = ud - pd, da = ua - pa;
        l_u0 = fma(c2rd_d, dd, l_u0); l_u1 = fma(c2rd_a, da, l_u1);
        l_p0 = -(c2rd_d * dd); l_p1 = -(c2rd_a * da);
        l_uu0 += c2rd_d; l_uu1 += c2rd_a;
        l_pp0 = c2rd_d; l_pp1 = c2rd_a;
        l_up0 = -c2rd_d; l_up1 = -c2rd_a;
      }
      // Q_s, Q_u
      double Qx[4];
      Qx[0] = l_x[0] + lx[0];
      Qx[1] = l_x[1] + lx[1];
      Qx[2] = fma(a02, lx[0], fma(a12, lx[1], l_x[2] + lx[2]));
      Qx[3] = fma(a03, lx[0], fma(a13, lx[1], fma(a23, lx[2], l_x[3] + lx[3])));
      const double Qp0 = l_p0, Qp1 = l_p1;
      const double Qu0 = fma(bd, lx[2], l_u0 + lp0);
      const double Qu1 = fma(dt, lx[3], l_u1 + lp1);
      // M = Wxx*A ; Qxx = l_xx + A'*M.  Wxx is symmetric by construction (mirrored upper triangle), so Qxx is symmetric up to
      // rounding: only its upper triangle is formed (a <= b) and used — 7 rows of products instead of 20, and no averaging of
      // the two halves (the CPU twin forms both and averages them; the difference is a rounding of the last bit).
      double M[4][4];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        M[a][0] = Wxx[a][0];
        M[a][1] = Wxx[a][1];
        M[a][2] = fma(Wxx[a][0], a02, fma(Wxx[a][1], a12, Wxx[a][2]));
        M[a][3] = fma(Wxx[a][0], a03, fma(Wxx[a][1], a13, fma(Wxx[a][2], a23, Wxx[a][3])));
      }
      M[3][3] = fma(Wxx[3][0], a03, fma(Wxx[3][1], a13, fma(Wxx[3][2], a23, Wxx[3][3])));
      double Qxx[4][4];     // entries a <= b only
#pragma unroll
      for (int b = 0; b < 4; ++b) Qxx[0][b] = M[0][b];
#pragma unroll
      for (int b = 1; b < 4; ++b) Qxx[1][b] = M[1][b];
#pragma unroll
      for (int b = 2; b < 4; ++b) Qxx[2][b] = fma(a02, M[0][b], fma(a12, M[1][b], M[2][b]));
      Qxx[3][3] = fma(a03, M[0][3], fma(a13, M[1][3], fma(a23, M[2][3], M[3][3])));
#pragma unroll
      for (int a = 0; a < 4; ++a) Qxx[a][a] += q2[a];
      // G = B'*Wxx + Wpx ; Qux = G*A ; Quu = l_uu + G*B + B'*Wxp + Wpp
      double G[2][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        G[0][b] = fma(bd, Wxx[2][b], Wxp[b][0]);
        G[1][b] = fma(dt, Wxx[3][b], Wxp[b][1]);
      }
      double Qux[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        Qux[a][0] = G[a][0];
        Qux[a][1] = G[a][1];
        Qux[a][2] = fma(G[a][0], a02, fma(G[a][1], a12, G[a][2]));
        Qux[a][3] = fma(G[a][0], a03, fma(G[a][1], a13, fma(G[a][2], a23, G[a][3])));
      }
      double Quu00 = fma(G[0][2], bd, fma(bd, Wxp[2][0], l_uu0 + Wpp00));
      // the off-diagonal of Q_uu once: its two halves (B'W B and its transpose) are equal up to rounding, the twin averages them
      const double Quu01 = fma(G[0][3], dt, fma(bd, Wxp[2][1], Wpp01));
      double Quu11 = fma(G[1][3], dt, fma(dt, Wxp[3][1], l_uu1 + Wpp11));
      // the box of the step: steering limits; the acceleration box of this knot's (nominal) speed = acceleration limits and
      // the speed bounds of knot i+1
      AccelBox ab = accel_box(p, inv_dt, v);
      double lo0 = lb0 - ud, hi0 = ub0 - ud, lo1 = ab.lo - ua, hi1 = ab.hi - ua;
      // active set of a Newton step (projected Newton): a control resting on a bound the gradient pushes it against stays
      // there (its box collapses to {0}) and the Hessian is judged on the controls that are left — see oracle/mpc_ref.cpp
      const bool hold0 = exact && ((lo0 >= 0.0 && Qu0 > 0.0) || (hi0 <= 0.0 && Qu0 < 0.0));
      const bool hold1 = exact && ((lo1 >= 0.0 && Qu1 > 0.0) || (hi1 <= 0.0 && Qu1 < 0.0));
      if (exact) {   // V_s . d2F; the steering curvature e00 only where it leaves the control Hessian of the controls not held
                     // positive definite (a saturated steering input otherwise proposes a jump to a box corner)
        const double e00 = ((lx[2] * vdw) * (tn + tn)) * sec2;
        const double g0 = Quu00 + e00 + mu, g3 = Quu11 + mu, go = Quu01;
        Qxx[2][2] = fma(-vdt, fma(lx[0], cs_, lx[1] * sn_), Qxx[2][2]);
        Qxx[2][3] = fma(dt, fma(lx[1], cs_, -(lx[0] * sn_)), Qxx[2][3]);
        Qux[0][3] = fma(lx[2] * sec2, dt_wb, Qux[0][3]);
        if (g0 > 1e-12 && (hold1 || fma(g0, g3, -(go * go)) > 1e-12 * g0)) Quu00 += e00;
      }
      const double hod = Quu01;
      const double h00 = Quu00 + mu, h11 = Quu11 + mu;
      // for a Newton step, a trust box around the current controls — with the exact (possibly indefinite) Hessian an
      // unrestricted stage proposes a jump to the far corner of the box, which the line search rejects at every step
      // length, and the solver falls back to linearly converging Gauss-Newton steps: that is the whole tail of the
      // iteration-count distribution.  Gauss-Newton steps are not restricted.
      if (exact) {
        lo0 = fmax(lo0, -trust_steer); hi0 = fmin(hi0, trust_steer);
        if (lo1 < -trust_accel) { lo1 = -trust_accel; ab.sp_lo = false; }
        if (hi1 > trust_accel) { hi1 = trust_accel; ab.sp_hi = false; }
      }
      if (hold0) { lo0 = 0.0; hi0 = 0.0; }
      if (hold1) { lo1 = 0.0; hi1 = 0.0; }
      double k0, k1; bool f0, f1;
      boxqp2(h00, hod, h11, Qu0, Qu1, lo0, hi0, lo1, hi1, k0, k1, f0, f1);
      // the acceleration rests on a SPEED bound: it is then a function of the state, a = (v_bound - v)/DT — a feedback row
      // -1/DT on v (the next knot's speed stays on the bound whatever v does) — and the steering gains see that row
      const bool sp = !f1 && ((k1 == hi1 && ab.sp_hi) || (k1 == lo1 && ab.sp_lo));
      // feedback K = -H_ff^-1 Q_us,f  over the 6 columns [Qux | l_up on the diagonal]
      double Qus[2][6];
#pragma unroll
      for (int b = 0; b < 4; ++b) { Qus[0][b] = Qux[0][b]; Qus[1][b] = Qux[1][b]; }
      Qus[0][4] = l_up0; Qus[0][5] = 0.0; Qus[1][4] = 0.0; Qus[1][5] = l_up1;
      // K = -Hinv * Qus with Hinv the inverse of the free block of [h00 hod; hod h11]: one reciprocal and selects instead
      // of a divergent four-way branch with up to 24 fp64 divisions (each ~35 instructions on gfx950)
      double K[2][6];
      {
        const bool both = f0 && f1;
        const double den = both ? fma(h00, h11, -(hod * hod)) : (f0 ? h00 : (f1 ? h11 : 1.0));
        const double inv = fast_div(1.0, den);
        const double i00 = both ? h11 * inv : (f0 ? inv : 0.0);
        const double i11 = both ? h00 * inv : (f1 ? inv : 0.0);
        const double i01 = both ? -hod * inv : 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          K[0][b] = -fma(i00, Qus[0][b], i01 * Qus[1][b]);
          K[1][b] = -fma(i01, Qus[0][b], i11 * Qus[1][b]);
        }
        K[0][4] = -(i00 * l_up0); K[1][4] = -(i01 * l_up0);      // Q_us columns 4, 5 = diag(l_up0, l_up1): the zero products are
        K[0][5] = -(i01 * l_up1); K[1][5] = -(i11 * l_up1);      // written out (x*0 and x+0 are not foldable in IEEE arithmetic)
        if (__any(sp)) {          // rare (never on the reference's scenario: 10 km/h against bounds of -20 / +55 km/h)
          if (sp) {
            const double ih = f0 ? fast_div(1.0, h00) : 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
              K[1][b] = (b == 3) ? -inv_dt : 0.0;
              K[0][b] = -fma(hod, K[1][b], Qus[0][b]) * ih;
            }
          }
        }
      }
      kf[i][0] = k0; kf[i][1] = k1;
      {
        float g[12];
#pragma unroll
        for (int b = 0; b < 6; ++b) { g[2 * b] = (float)K[0][b]; g[2 * b + 1] = (float)K[1][b]; }
        stKf(i, g);
      }
      gnorm = fmax(gnorm, fmax(fabs(k0), fabs(k1)));
      // expected change and value function (unregularised, symmetrised Quu)
      const double Quuk0 = fma(Quu00, k0, hod * k1), Quuk1 = fma(hod, k0, Quu11 * k1);
      dV1 = fma(k0, Qu0, fma(k1, Qu1, dV1));
      dV2 = fma(0.5, fma(k0, Quuk0, k1 * Quuk1), dV2);
      const double t0 = Quuk0 + Qu0, t1 = Quuk1 + Qu1;
      double Vs[6];
#pragma unroll
      for (int b = 0; b < 4; ++b) Vs[b] = fma(K[0][b], t0, fma(K[1][b], t1, fma(Qus[0][b], k0, fma(Qus[1][b], k1, Qx[b]))));
      Vs[4] = fma(K[0][4], t0, fma(K[1][4], t1, fma(l_up0, k0, Qp0)));
      Vs[5] = fma(K[0][5], t0, fma(K[1][5], t1, fma(l_up1, k1, Qp1)));
      double Vss[6][6];
      // V_ss = Q_ss + K'Quu K + K'Q_us + Q_us'K.  The gains solve (Quu + mu I)_FF K_F = -Q_us,F on the free controls
      // (rows of clamped controls are zero), so K'Quu K = -K'Q_us - mu K'K and the three products collapse into
      //   V_ss = Q_ss + Q_us'K - mu K'K
      // (the familiar Q_ss - Q_su Quu^-1 Q_us when mu = 0).  Only the upper triangle is formed and mirrored.
#pragma unroll
      for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = a; b < 4; ++b) Vss[a][b] = fma(Qus[0][a], K[0][b], fma(Qus[1][a], K[1][b], Qxx[a][b]));
#pragma unroll
        for (int b = 4; b < 6; ++b) Vss[a][b] = fma(Qus[0][a], K[0][b], Qus[1][a] * K[1][b]);
      }
      Vss[4][4] = fma(l_up0, K[0][4], l_pp0);
      Vss[4][5] = l_up0 * K[0][5];
      Vss[5][5] = fma(l_up1, K[1][5], l_pp1);
      if (mu != 0.0) {      // rare: the regularised iterations
        const double mu_k = sp ? 0.0 : mu;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = a; b < 6; ++b) Vss[a][b] = fma(-mu_k, fma(K[0][a], K[0][b], K[1][a] * K[1][b]), Vss[a][b]);
      }
      if (__any(sp)) {      // a prescribed feedback row: the identity above does not hold, the general form is evaluated
        if (sp) {
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) {
              const double qk0 = fma(Quu00, K[0][b], hod * K[1][b]), qk1 = fma(hod, K[0][b], Quu11 * K[1][b]);
              Vss[a][b] += fma(K[0][a], qk0, K[1][a] * qk1) + fma(K[0][a], Qus[0][b], K[1][a] * Qus[1][b]);
            }
        }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        lx[a] = Vs[a];
#pragma unroll
        for (int b = 0; b < 4; ++b) Wxx[a][b] = (a <= b) ? Vss[a][b] : Vss[b][a];
        Wxp[a][0] = Vss[a][4]; Wxp[a][1] = Vss[a][5];
      }
      lp0 = Vs[4]; lp1 = Vs[5];
      Wpp00 = Vss[4][4]; Wpp01 = Vss[4][5]; Wpp11 = Vss[5][5];
    };
    if constexpr (!CKPT) {
      terminal(S[cur][N]);
      StageIn nx = load_stage(cur, N - 1);
      for (int i = N - 1; i >= 0; --i) {
        const StageIn in = nx;
        const double ud = uc0, ua = uc1;
        const bool inner = i >= 1;
        const double pd = inner ? up0 : 0.0, pa = inner ? up1 : 0.0;
        // unconditional (clamped index) so that the number of loads in flight is the same on every path: with a branch
        // around them the compiler has to drain the memory queue (vmcnt(0)) before the first use of `in`
        nx = load_stage(cur, i >= 1 ? i - 1 : 0);
        uc0 = up0; uc1 = up1;
        ldU(cur, i >= 2 ? i - 2 : 0, up0, up1);
        double sn_ = in.sn, cs_ = in.cs, tn = in.tn;
        if constexpr (LEAN) {            // what step() computed when this knot was rolled out: the same functions of the same doubles
          mpc_sincos(in.s2, &sn_, &cs_);
          tn = small_steer ? mpc_tan_small(ud) : mpc_tan(ud);
        }
        stage(i, in.s0, in.s1, in.s2, in.s3, sn_, cs_, tn, in.r, ud, ua, pd, pa);
      }
    } else {
      // The checkpointed sweep: stages in pairs (odd i, then i - 1).  E = the even knot at or below the stage about to be swept; tE =
      // the trig of (E.yaw, steering of E's stage), computed when the odd stage above E needs it for its model step and used again by
      // E's own stage.  The even knot below is requested at the top of an even stage and consumed by the odd stage after it.
      double E[4], tE0 = 0.0, tE1 = 0.0, tE2 = 0.0, sN[4];
      if (N & 1) {                       // the terminal knot is odd: one step from knot N - 1, whose stage comes first
        knot_even(N - 1, E);
        trig3(E[2], uc0, tE0, tE1, tE2);
        step_tr(E, uc1, tE0, tE1, tE2, sN);
      } else {
        knot_even(N, sN);
        knot_even(N - 2, E);
      }
      terminal(sN);
      float4 rn = xr4[N - 1];
      auto even_stage = [&](const int i) {          // E = knot i, tE valid
        const double ud = uc0, ua = uc1;
        const bool inner = i >= 1;
        const double pd = inner ? up0 : 0.0, pa = inner ? up1 : 0.0;
        const float4 r = rn;
        rn = xr4[i >= 1 ? i - 1 : 0];
        double En[4];
        knot_even(i >= 2 ? i - 2 : 0, En);
        uc0 = up0; uc1 = up1;
        ldU(cur, i >= 2 ? i - 2 : 0, up0, up1);
        stage(i, E[0], E[1], E[2], E[3], tE0, tE1, tE2, r, ud, ua, pd, pa);
        E[0] = En[0]; E[1] = En[1]; E[2] = En[2]; E[3] = En[3];
      };
      auto odd_stage = [&](const int i) {           // E = knot i - 1 (i >= 1)
        const double ud = uc0, ua = uc1, pd = up0, pa = up1;
        const float4 r = rn;
        rn = xr4[i - 1];
        uc0 = up0; uc1 = up1;
        ldU(cur, i >= 2 ? i - 2 : 0, up0, up1);
        trig3(E[2], pd, tE0, tE1, tE2);             // the trig of stage i - 1 ...
        double s[4];
        step_tr(E, pa, tE0, tE1, tE2, s);           // ... whose model step gives knot i: what the rollout stored in the other layouts
        double sn_, cs_, tn;
        trig3(s[2], ud, sn_, cs_, tn);
        stage(i, s[0], s[1], s[2], s[3], sn_, cs_, tn, r, ud, ua, pd, pa);
      };
      int i = N - 1;
      if (N & 1) { even_stage(i); --i; }
      for (; i >= 1; i -= 2) { odd_stage(i); even_stage(i - 1); }
    }
#if CRX_MPC_TICKS >= 2
    tk_b += clock64() - tk_b0; tk_nb++;
#endif
  };
  // rollout(alpha, nxt): the candidate u + alpha k + K dx rolled through the model into buffer `nxt`; returns its cost
  auto rollout = [&](const double alpha, const int nxt) -> double {
#if CRX_MPC_TICKS >= 2
    const long long tk_f0 = clock64();
#endif
    double Jn = 0.0;
    // The candidate rollout carries its state and previous control in registers (they are also written to S[nxt],
    // U[nxt] for the next backward sweep, but never read back here: a store-to-load round trip through private memory
    // per stage would sit on the critical path), and requests stage i + 1's operands while stage i computes.
    double xs[4], xc[4];                                     // the candidate's knot i; CKPT: the accepted trajectory's knot i beside it
    if constexpr (CKPT) {
      xs[0] = xc[0] = (double)xi.x; xs[1] = xc[1] = (double)xi.y; xs[2] = xc[2] = (double)xi.z; xs[3] = xc[3] = (double)xi.w;
    } else {
      xs[0] = S[cur][0][0]; xs[1] = S[cur][0][1]; xs[2] = S[cur][0][2]; xs[3] = S[cur][0][3];
    }
    double pnd = 0.0, pna = 0.0, pcd = 0.0, pca = 0.0;       // previous stage's new / current controls
    // (TILE, round 6: requesting the private-memory part — knot, feed-forward step, reference row — TWO stages ahead and reading the
    // gains from their registers at the point of use was built and measured: 1 M agents 15.0 -> 16.4 ms lockstep, 11.7 -> 12.0 refilled,
    // relative to the private-memory kernel of the same box 0.82 -> 0.92 and 0.64 -> 0.67: slower, like the two-stage queue of round 4.)
    RollIn nx = load_roll(cur, 0);
    for (int i = 0; i < N; ++i) {
      const RollIn in = nx;
      nx = load_roll(cur, i + 1 < N ? i + 1 : N - 1);      // unconditional, clamped: see the backward sweep (two stages in
                                                           // flight were measured too: 4 % slower)
      double d0, d1, d2, d3;
      if constexpr (CKPT) { d0 = xs[0] - xc[0]; d1 = xs[1] - xc[1]; d2 = xs[2] - xc[2]; d3 = xs[3] - xc[3]; }
      else { d0 = xs[0] - in.s[0]; d1 = xs[1] - in.s[1]; d2 = xs[2] - in.s[2]; d3 = xs[3] - in.s[3]; }
      const double d4 = (i >= 1) ? pnd - pcd : 0.0;
      const double d5 = (i >= 1) ? pna - pca : 0.0;
      double du0 = alpha * in.k0;
      du0 = fma(in.K[0], d0, du0); du0 = fma(in.K[2], d1, du0); du0 = fma(in.K[4], d2, du0); du0 = fma(in.K[6], d3, du0); du0 = fma(in.K[8], d4, du0); du0 = fma(in.K[10], d5, du0);
      double du1 = alpha * in.k1;
      du1 = fma(in.K[1], d0, du1); du1 = fma(in.K[3], d1, du1); du1 = fma(in.K[5], d2, du1); du1 = fma(in.K[7], d3, du1); du1 = fma(in.K[9], d4, du1); du1 = fma(in.K[11], d5, du1);
      const AccelBox nb = accel_box(p, inv_dt, xs[3]);            // the box of a_i at the NEW speed of knot i
      const double nd = clampd(in.u0 + du0, lb0, ub0);
      const double na = clampd(in.u1 + du1, nb.lo, nb.hi);
      stU(nxt, i, nd, na);
      Jn += ctrl_cost(i >= 1, nd, na, pnd, pna);                  // ctrl(nxt, i)
      if (i >= 1) Jn += track_cost(in.r, xs);                     // track(xs, i)
      double xn[4];
      step(xs, nd, na, xn, LEAN ? TR[0][0] : TR[nxt][i]);
      if constexpr (CKPT) {
        put_knot(i + 1, xn);
        double xcn[4];                                              // the accepted trajectory's next knot: what S[cur][i + 1] holds in the other layouts
        if (i + 1 < N) {                                            // (the last stage's successor is never compared with)
          step(xc, in.u0, in.u1, xcn, nullptr);
          xc[0] = xcn[0]; xc[1] = xcn[1]; xc[2] = xcn[2]; xc[3] = xcn[3];
        }
      } else {
        S[nxt][i + 1][0] = xn[0]; S[nxt][i + 1][1] = xn[1]; S[nxt][i + 1][2] = xn[2]; S[nxt][i + 1][3] = xn[3];
      }
      xs[0] = xn[0]; xs[1] = xn[1]; xs[2] = xn[2]; xs[3] = xn[3];
      pnd = nd; pna = na; pcd = in.u0; pca = in.u1;
    }
    Jn += track_cost(rN, xs);                                     // track(xs, N), terminal reference kept in registers
#if CRX_MPC_TICKS >= 2
    tk_nf++; tk_f += clock64() - tk_f0;
#endif
    return Jn;
  };
  // what the end of a line search does to the globalisation state
  auto accept_step = [&](const double alpha, const double Jn, const int nxt) {
    J = Jn; cur = nxt;
    if (gn_left > 0) gn_left--;
    if (alpha == 1.0) mu *= 0.1;
    if (mu < mu_min) mu = 0.0;
  };
  auto reject_step = [&](const bool exact) {
    if (exact) {
      gn_run = gn_run * 2 > 16 ? 16 : gn_run * 2;   // every failed Newton attempt doubles the Gauss-Newton run after it
      gn_left = gn_run;
    } else {
      mu = (mu * 10.0 > 1e-3) ? mu * 10.0 : 1e-3;
      if (mu > mu_max) done = true;
    }
  };

  if constexpr (!REFILL) {
    // The latency regime (mpc_kernel, the portfolio, the closed loop): the lanes of a wave sweep in lockstep, a sweep = the backward pass
    // and the whole line search (a Newton step that fails down to alpha = 1/8 is dropped for Gauss-Newton ones: 4 tries, otherwise 10).
    for (int iter = (PHASED && resumed) ? ph.resume : 0; iter < p.max_iter; ++iter) {
      if (PORTFOLIO) { if (quad_converged()) done = true; }    // a sibling variant has the answer: every lane of the quad stops
      if (__all(done)) break;
      if constexpr (PHASED) {
        if (ph.cap > 0 && iter == ph.cap) {                    // the phase ends here: the lanes still sweeping hand their state over
          if (!done) {
            suspended = true;
            ph.st[0] = J; ph.st[1] = mu; ph.st[2] = (double)gn_left; ph.st[3] = (double)gn_run; ph.st[4] = (double)status; ph.st[5] = 0.0;
            for (int i = 0; i < N; ++i) { double ud_, ua_; ldU(cur, i, ud_, ua_); ph.st[6 + 2 * i] = ud_; ph.st[7 + 2 * i] = ua_; }
          }
          break;
        }
      }
      if (done) continue;
      it = iter;
      const bool exact = gn_left <= 0;
      backward(exact);
      if (gnorm < p.tol && mu == 0.0) { status |= 1; done = true; continue; }
      const double aJ = fabs(J);
      const double noise = 1e-12 * (aJ > 1.0 ? aJ : 1.0);
      const bool trust = -(dV1 + dV2) < noise;
      bool accepted = false;
      double alpha = 1.0;
      const int nxt = cur ^ 1;
      const int ls_max = exact ? 4 : 10;
      for (int ls = 0; ls < ls_max; ++ls) {
        const double Jn = rollout(alpha, nxt);
        if (Jn < J || (trust && Jn <= J + noise)) { accept_step(alpha, Jn, nxt); accepted = true; break; }
        alpha *= 0.5;
      }
      if (!accepted) { reject_step(exact); restore_knots(); }
      if (iter == p.max_iter - 1) it = p.max_iter;
    }
  } else {
    // mpc_refill_kernel (A/B build): the line search is scheduled ASYNCHRONOUSLY across the lanes of the wave (round 5).  A trip of the
    // loop is one backward pass for the lanes that are due one and ONE candidate rollout for the lanes with a line search pending; a lane
    // whose candidate is refused halves alpha and rolls again in the NEXT trip, next to the other lanes' first candidates.  A refilled wave
    // always holds some lane in a long line search, so the lockstep loop above makes it run 3-4 rollout passes per sweep, most of them for
    // a handful of lanes.  Measured (profiles/r05/mpc_variants_ab_run2_async_everywhere.jsonl): the refilling kernel at 262,144 agents
    // 7.83 -> 7.29 ms, memory instructions -25 %.  In the latency regime the same scheduling LOSES (8,192 agents: 0.94 -> 1.17 ms): a
    // retrying lane waits through the other lanes' backward pass (three times a rollout) before its next candidate, and it is the
    // retrying lanes that set a launch's critical path — so mpc_kernel keeps the lockstep loop.  Per agent both loops run the same
    // backward passes and the same rollouts with the same step lengths in the same order: bit-identical results (tests/test_mpc_gpu.py).
    int sweep = 0;                    // backward passes the lane's agent has started (= the sweep index reported in `status`)
    bool fwd = false;                 // a line search is pending: the next candidate uses `alpha`
    double alpha = 1.0, noise = 0.0;
    int ls = 0;
    bool trust = false;
    for (;;) {
      {
        const lanemask_t sweeping = lanes_where(!done);
        const lanemask_t held = lanes_where(done && agent_l >= 0);
        if (held && (!sweeping || __builtin_popcountll(held) >= feed.hold)) {        // hand the finished agents back, all at once
          if (done && agent_l >= 0) {
            write_solution(feed.solg ? feed.solg + (size_t)agent_l * nv_ : nullptr);
            if (feed.statusg) feed.statusg[agent_l] = status | (it << 8);
            if (feed.costg) feed.costg[agent_l] = J;
            agent_l = -1;
          }
          want |= held;
        }
        if (want && next < feed.hi) {                                                 // the lanes without an agent take the next ones
          const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
          const int mine = next + (int)rank;
          next += __builtin_popcountll(want);
          const bool got = ((want >> (threadIdx.x & 63)) & 1) && mine < feed.hi;
          if (got) {
            agent_l = mine;
            xi = reinterpret_cast<const float4*>(feed.x0g)[mine];
            xr4 = reinterpret_cast<const float4*>(feed.xrefg) + (size_t)mine * (size_t)T;
              rN = xr4[N];
            start();
            sweep = 0; fwd = false;
            done = false;
          }
          want &= ~lanes_where(got);
        }
        if (!lanes_where(!done)) break;                          // nobody sweeping: every held agent was handed back above
      }
      if (!done && !fwd) {
        it = sweep;
        backward(gn_left <= 0);
        if (gnorm < p.tol && mu == 0.0) { status |= 1; done = true; }
        else {
          const double aJ = fabs(J);
          noise = 1e-12 * (aJ > 1.0 ? aJ : 1.0);
          trust = -(dV1 + dV2) < noise;
          alpha = 1.0; ls = 0; fwd = true;
        }
      }
      if (!done && fwd) {
        const bool exact = gn_left <= 0;
        const int nxt = cur ^ 1;
        const double Jn = rollout(alpha, nxt);
        bool end = false;
        if (Jn < J || (trust && Jn <= J + noise)) { accept_step(alpha, Jn, nxt); end = true; }
        else {
          alpha *= 0.5;
          if (++ls == (exact ? 4 : 10)) { reject_step(exact); restore_knots(); end = true; }   // a Newton step that fails down to alpha = 1/8 is dropped for Gauss-Newton ones
        }
        if (end) {
          fwd = false;
          if (sweep == p.max_iter - 1) { it = p.max_iter; done = true; }
          ++sweep;
        }
      }
    }
  }
  status_out = 0; cost_out = 0.0; a0_out = 0.0f; d0_out = 0.0f;
  if (REFILL) return;                                        // every agent of the range has been written from inside the loop
  if (PORTFOLIO) {
    // the agent's answer: the converged variant of lowest index; if none converged within max_iter sweeps, variant 0's iterate
    const unsigned q = quad_converged();
    const int winner = q ? (__builtin_ctz(q)) : 0;
    status_out = -1;
    if (quad_lane != winner) return;
    if (live) status |= winner << 2;
  }
  if (!live) return;
  if constexpr (PHASED) {
    if (suspended) { status_out = kMpcSuspended; return; }
  }
  if (!(status & 1) && !done) it = p.max_iter;
  write_solution(so);
  status_out = status | (it << 8);
  cost_out = J;
#ifdef CRX_MPC_TICKS
  { const int ln = threadIdx.x & 63; const long long tt = clock64() - tk_0;   // diagnostic build: lanes 0-4 report the wave's
    for (int o = 32; o >= 1; o >>= 1) {                                          // longest lane instead of their cost
      tk_b = max(tk_b, __shfl_xor(tk_b, o)); tk_f = max(tk_f, __shfl_xor(tk_f, o));
      tk_nb = max(tk_nb, __shfl_xor(tk_nb, o)); tk_nf = max(tk_nf, __shfl_xor(tk_nf, o));
    }
    if (ln == 0) cost_out = (double)tt; if (ln == 1) cost_out = (double)tk_b; if (ln == 2) cost_out = (double)tk_f;
    if (ln == 3) cost_out = (double)tk_nb; if (ln == 4) cost_out = (double)tk_nf; }
#endif
  {
    double ud_, ua_;
    ldU(cur, 0, ud_, ua_);
    d0_out = (float)ud_;
    a0_out = (float)ua_;
  }
}

#ifndef CRX_MPC_TILE_MODULE      // (mpc_tile_module.hip takes mpc_solve_lane only: the kernels and launchers below belong to the library's own code object)
// `live_lanes` agents in the low lanes of every wave, `blockDim.x / 64` waves per workgroup.  Production: full waves (64) in
// single-wave workgroups.  The launch lasts as long as its slowest wave, and a wave pays in every sweep for the agent of
// its lanes that needs the most line-search rollouts, so while the batch leaves SIMDs idle (BASELINE configs[3]: 128 full
// waves on 1,024 SIMDs) emptier waves look attractive — measured (profiles/r02/mpc_tail.txt), they are slower: 32 / 16 /
// 8 agents per wave take 1.18 / 1.41 / 1.95 ms against 1.08, and so do two or four waves per workgroup (1.18 / 1.42 ms at
// full waves): every wave streams its lanes' 5.4 KB of private memory through L2 each sweep whether the lanes are used or
// not, and waves that share a CU share its path to it.
template <int MAXT, bool LEAN = false>
__global__ void __launch_bounds__(256)   // 1-4 waves per workgroup, one wave per SIMD: the register budget of a lone wave
mpc_kernel(int n, int T, int live_lanes, const float* __restrict__ x0g, const float* __restrict__ xrefg, MpcP p,
           float* __restrict__ solg, int* __restrict__ statusg, double* __restrict__ costg) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t agent = wave * (size_t)live_lanes + lane;
  const bool live = lane < live_lanes && agent < (size_t)n;
  const size_t ag = live ? agent : 0;
  const float4* __restrict__ xr4 = reinterpret_cast<const float4*>(xrefg) + ag * (size_t)T;
  const float4 xi = reinterpret_cast<const float4*>(x0g)[ag];
  const size_t nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
  int status; double J; float a0, d0;
  mpc_solve_lane<MAXT, false, false, LEAN>(live, T, xi, xr4, p, live ? solg + agent * nv : nullptr, status, J, a0, d0);
  if (!live) return;
  if (statusg) statusg[agent] = status;
  if (costg) costg[agent] = J;
}

// The TWO-PHASE solve (round 6; csrc/api_mpc.inl: crx_mpc_solve_two_phase_dev).  A launch lasts as long as its slowest agent's chain of
// sweeps, and the distribution of sweep counts has a thin, long tail (configs[3]: mean 6.8, 2.8 % above 10, 0.03 % above 16, a few at
// the cap of 50); a pipelined host (configs[4]) can keep only so many launches in flight, so its round time is that latency divided by
// the depth.  Phase 1 is the ordinary launch with the sweep cap lowered to `first_sweeps`; mpc_collect_kernel lists the agents that
// ran into that cap; mpc_list_kernel solves those FROM SCRATCH with the caller's cap — the solver is deterministic, so an agent's
// answer is bit for bit what the single launch gives it (its phase-1 sweeps are thrown away: first_sweeps x a few per cent of the
// agents).  The second launch is a handful of waves and can run on a stream of its own, behind the first one's.
__global__ void __launch_bounds__(256)
mpc_collect_kernel(int n, int cap, const int* __restrict__ statusg, int* __restrict__ list, int* __restrict__ count) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  bool un = false;
  if (i < n) { const int st = statusg[i]; un = cap < 0 ? st == kMpcSuspended : (!(st & 1) && (st >> 8) >= cap); }   // cap < 0: the phased solve's suspended agents
  const lanemask_t m = lanes_where(un);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == __builtin_ctzll(m)) base = atomicAdd(count, __builtin_popcountll(m));
  base = __builtin_amdgcn_readlane(base, __builtin_ctzll(m));
  if (un) list[base + __builtin_popcountll(m & ((lanemask_t(1) << lane) - 1))] = i;
}
// entries [64 w, 64 w + 64) of the list on wave w (waves past the count leave at once: the grid is sized for the worst case)
template <int MAXT>
__global__ void __launch_bounds__(64)
mpc_list_kernel(const int* __restrict__ list, const int* __restrict__ count, int T, const float* __restrict__ x0g, const float* __restrict__ xrefg,
                MpcP p, float* __restrict__ solg, int* __restrict__ statusg, double* __restrict__ costg) {
  const int cnt = *count;
  const int entry = (int)blockIdx.x * 64 + (int)threadIdx.x;
  if ((int)blockIdx.x * 64 >= cnt) return;
  const bool live = entry < cnt;
  const size_t agent = (size_t)list[live ? entry : 0];
  const float4* __restrict__ xr4 = reinterpret_cast<const float4*>(xrefg) + agent * (size_t)T;
  const float4 xi = reinterpret_cast<const float4*>(x0g)[agent];
  const size_t nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
  int status; double J; float a0, d0;
  mpc_solve_lane<MAXT, false, false, false>(live, T, xi, xr4, p, live ? solg + agent * nv : nullptr, status, J, a0, d0);
  if (!live) return;
  if (statusg) statusg[agent] = status;
  if (costg) costg[agent] = J;
}

// The PHASED solve (round 6; csrc/api_mpc.inl: crx_x_mpc_solve_phased_dev).  A lockstep wave runs as many sweeps as the slowest of its 64
// agents — 11.9 on average on the configs[3] distribution against a mean of 6.8 per agent: 43 % of the lane-sweeps are masked off.
// Here the batch is swept in PHASES: phase k runs every agent still unconverged up to sweep index caps[k]; the agents that reach the
// cap suspend (MpcPhase), are compacted into full waves (mpc_collect_kernel on status == kMpcSuspended) and resumed by the next
// phase's launch.  Per agent the same sweeps in the same order on the same doubles: bit-identical outputs.  list == nullptr: phase 0
// (agent = wave * 64 + lane, start from the zero guess).
template <int MAXT, bool LEAN>
__global__ void __launch_bounds__(64)
mpc_phase_kernel(int n, const int* __restrict__ list, const int* __restrict__ count, int T, int resume, int cap, double* __restrict__ state,
                 const float* __restrict__ x0g, const float* __restrict__ xrefg, MpcP p, float* __restrict__ solg, int* __restrict__ statusg,
                 double* __restrict__ costg) {
  const int cnt = list ? *count : n;
  const int entry = (int)blockIdx.x * 64 + (int)threadIdx.x;
  if ((int)blockIdx.x * 64 >= cnt) return;
  const bool live = entry < cnt;
  const size_t agent = live ? (size_t)(list ? list[entry] : entry) : (size_t)(list ? list[0] : 0);
  const float4* __restrict__ xr4 = reinterpret_cast<const float4*>(xrefg) + agent * (size_t)T;
  const float4 xi = reinterpret_cast<const float4*>(x0g)[agent];
  const size_t nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
  int status; double J; float a0, d0;
  mpc_solve_lane<MAXT, false, false, LEAN, 0, true>(live, T, xi, xr4, p, live ? solg + agent * nv : nullptr, status, J, a0, d0, MpcFeed{}, MpcTile{},
                                                    MpcPhase{cap, resume, state + agent * (size_t)mpc_phase_record_doubles(T)});
  if (!live) return;
  statusg[agent] = status;                       // kMpcSuspended for the agents the next phase resumes
  if (status != kMpcSuspended && costg) costg[agent] = J;
}

// The lane-refilling launch: wave w owns the agents [w * chunk, (w + 1) * chunk) and refills its lanes (mpc_solve_lane<.., REFILL>): a wave
// of mpc_kernel lasts as long as the slowest of its 64 agents (mean of the wave maximum ~12 sweeps against a mean of 6.8), here a finished
// lane takes the wave's next agent, and the line search is scheduled asynchronously.  Per agent the same sweeps in the same order:
// bit-identical to mpc_kernel.  MEASURED AND REJECTED, twice: round 4 1.14x at 65,536 agents, 1.16x at 262,144, 0.99x at 1 M
// (profiles/r04/mpc_refill_ab.jsonl); round 5, after the solver's memory traffic fell by a third (float gains, recomputed trig) and with
// the asynchronous line search, 1.00x / 0.96x / 0.89x (profiles/r05/mpc_variants_ab.jsonl): where waves queue the memory system, not the
// idle lanes, is the limit, and mpc_kernel's short-lived waves hand their SIMD and their share of the caches to the next one.  Compiled
// into the A/B build only (CRX_EXPERIMENTAL_KERNELS), reachable through crx_x_mpc_solve_refill_dev.
#if CRX_EXPERIMENTAL_KERNELS
template <int MAXT>
__global__ void __launch_bounds__(64)
mpc_refill_kernel(int n, int T, int chunk, int hold, const float* __restrict__ x0g, const float* __restrict__ xrefg, MpcP p,
                  float* __restrict__ solg, int* __restrict__ statusg, double* __restrict__ costg) {
  const int lo = (int)blockIdx.x * chunk;
  const MpcFeed feed{lo, (n - lo < chunk) ? n : lo + chunk, hold, x0g, xrefg, solg, statusg, costg};
  int status; double J; float a0, d0;
  mpc_solve_lane<MAXT, false, true>(false, T, float4{0.f, 0.f, 0.f, 0.f}, nullptr, p, nullptr, status, J, a0, d0, feed);
}
#endif

// The portfolio launch: agent a on lanes 4a .. 4a+3 (16 agents per wave, single-wave workgroups): 4x the waves of mpc_kernel — at the
// BASELINE batch 512 waves on 1,024 SIMDs, still one per SIMD.
template <int MAXT>
__global__ void __launch_bounds__(64)
mpc_portfolio_kernel(int n, int T, const float* __restrict__ x0g, const float* __restrict__ xrefg, MpcP p,
                     float* __restrict__ solg, int* __restrict__ statusg, double* __restrict__ costg) {
  const size_t agent = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const bool live = agent < (size_t)n;
  const size_t ag = live ? agent : 0;
  const float4* __restrict__ xr4 = reinterpret_cast<const float4*>(xrefg) + ag * (size_t)T;
  const float4 xi = reinterpret_cast<const float4*>(x0g)[ag];
  const size_t nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
  int status; double J; float a0, d0;
  mpc_solve_lane<MAXT, true>(live, T, xi, xr4, p, live ? solg + agent * nv : nullptr, status, J, a0, d0);
  if (!live || status < 0) return;                      // the winning lane of the quad reports
  if (statusg) statusg[agent] = status;
  if (costg) costg[agent] = J;
}

#endif  // CRX_MPC_TILE_MODULE

inline MpcP mpc_pack(const crx_mpc_params& q) {
  MpcP p;
  p.dt = q.dt; p.wb = q.wb; p.max_steer = q.max_steer; p.max_accel = q.max_accel;
  p.max_speed = q.max_speed; p.min_speed = q.min_speed;
  p.r_a = q.r_a; p.r_d = q.r_delta; p.rd_a = q.rd_a; p.rd_d = q.rd_delta;
  p.qx = q.q_x; p.qy = q.q_y; p.qyaw = q.q_yaw; p.qv = q.q_v; p.tol = q.tol; p.max_iter = q.max_iter;
  return p;
}

#ifndef CRX_MPC_TILE_MODULE
inline hipError_t mpc_portfolio_launch(int n, int T, const float* x0, const float* xref, const crx_mpc_params& q, float* sol,
                                       int* status, double* cost, hipStream_t stream) {
  const MpcP p = mpc_pack(q);
  const dim3 grid((unsigned)((4 * (size_t)n + 63) / 64)), block(64);
  if (T <= 8)
    hipLaunchKernelGGL((mpc_portfolio_kernel<8>), grid, block, 0, stream, n, T, x0, xref, p, sol, status, cost);
  else if (T <= 24)
    hipLaunchKernelGGL((mpc_portfolio_kernel<24>), grid, block, 0, stream, n, T, x0, xref, p, sol, status, cost);
  else
    hipLaunchKernelGGL((mpc_portfolio_kernel<CRX_MPC_MAX_T>), grid, block, 0, stream, n, T, x0, xref, p, sol, status, cost);
  return hipGetLastError();
}

inline hipError_t mpc_phase_launch(int n, int T, const int* list, const int* count, int resume, int cap, double* state, const float* x0,
                                   const float* xref, const crx_mpc_params& q, float* sol, int* status, double* cost, hipStream_t stream, bool lean,
                                   int grid_agents) {
  const MpcP p = mpc_pack(q);
  const dim3 grid((unsigned)(((size_t)grid_agents + 63) / 64)), block(64);
  if (T > 24) return hipErrorInvalidValue;
  if (lean) hipLaunchKernelGGL((mpc_phase_kernel<24, true>), grid, block, 0, stream, n, list, count, T, resume, cap, state, x0, xref, p, sol, status, cost);
  else hipLaunchKernelGGL((mpc_phase_kernel<24, false>), grid, block, 0, stream, n, list, count, T, resume, cap, state, x0, xref, p, sol, status, cost);
  return hipGetLastError();
}
// phase 2 of the two-phase solve: n = the size of the batch the list was collected from (the grid's worst case)
inline hipError_t mpc_list_launch(int n, int T, const int* list, const int* count, const float* x0, const float* xref, const crx_mpc_params& q,
                                  float* sol, int* status, double* cost, hipStream_t stream) {
  const MpcP p = mpc_pack(q);
  const dim3 grid((unsigned)(((size_t)n + 63) / 64)), block(64);
  if (T <= 8) hipLaunchKernelGGL((mpc_list_kernel<8>), grid, block, 0, stream, list, count, T, x0, xref, p, sol, status, cost);
  else if (T <= 24) hipLaunchKernelGGL((mpc_list_kernel<24>), grid, block, 0, stream, list, count, T, x0, xref, p, sol, status, cost);
  else hipLaunchKernelGGL((mpc_list_kernel<CRX_MPC_MAX_T>), grid, block, 0, stream, list, count, T, x0, xref, p, sol, status, cost);
  return hipGetLastError();
}

#if CRX_EXPERIMENTAL_KERNELS
// lanes refilled: `chunk` agents per wave, hand-back in batches of `hold` lanes (max_iter >= 1)
inline hipError_t mpc_refill_launch(int n, int T, const float* x0, const float* xref, const crx_mpc_params& q, float* sol,
                                    int* status, double* cost, hipStream_t stream, int chunk, int hold) {
  const MpcP p = mpc_pack(q);
  const dim3 grid((unsigned)(((size_t)n + chunk - 1) / chunk)), block(64);
  if (T <= 8)
    hipLaunchKernelGGL((mpc_refill_kernel<8>), grid, block, 0, stream, n, T, chunk, hold, x0, xref, p, sol, status, cost);
  else if (T <= 24)
    hipLaunchKernelGGL((mpc_refill_kernel<24>), grid, block, 0, stream, n, T, chunk, hold, x0, xref, p, sol, status, cost);
  else
    hipLaunchKernelGGL((mpc_refill_kernel<CRX_MPC_MAX_T>), grid, block, 0, stream, n, T, chunk, hold, x0, xref, p, sol, status, cost);
  return hipGetLastError();
}
#endif

template <bool LEAN>
inline void mpc_launch_T(int n, int T, int live, dim3 grid, dim3 block, hipStream_t stream, const float* x0, const float* xref, const MpcP& p,
                         float* sol, int* status, double* cost) {
  if (T <= 8)
    hipLaunchKernelGGL((mpc_kernel<8, LEAN>), grid, block, 0, stream, n, T, live, x0, xref, p, sol, status, cost);
  else if (T <= 24)
    hipLaunchKernelGGL((mpc_kernel<24, LEAN>), grid, block, 0, stream, n, T, live, x0, xref, p, sol, status, cost);
  else
    hipLaunchKernelGGL((mpc_kernel<CRX_MPC_MAX_T, LEAN>), grid, block, 0, stream, n, T, live, x0, xref, p, sol, status, cost);
}
// trig: 0 = the rollout's trig stored for the backward sweep, 1 = recomputed there, anything else = by batch size (kMpcLeanFrom)
inline hipError_t mpc_launch(int n, int T, const float* x0, const float* xref, const crx_mpc_params& q, float* sol,
                             int* status, double* cost, hipStream_t stream, int live = 64, int wg_waves = 1, int trig = -1) {
  const MpcP p = mpc_pack(q);
  if (live < 1 || live > 64) live = 64;
  if (wg_waves < 1 || wg_waves > 4) wg_waves = 1;
  const size_t waves = ((size_t)n + live - 1) / live;
  const dim3 grid((unsigned)((waves + wg_waves - 1) / wg_waves)), block(64 * wg_waves);
  const bool lean = trig == 1 || (trig != 0 && (n >= kMpcLeanFrom || (q.shared_gpu != 0 && n >= kMpcLeanFromShared)));
  if (lean) mpc_launch_T<true>(n, T, live, grid, block, stream, x0, xref, p, sol, status, cost);
  else mpc_launch_T<false>(n, T, live, grid, block, stream, x0, xref, p, sol, status, cost);
  return hipGetLastError();
}
#endif  // CRX_MPC_TILE_MODULE

}  // namespace crx


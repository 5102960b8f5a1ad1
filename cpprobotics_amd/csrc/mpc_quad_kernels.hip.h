// mpc_quad_kernels.hip.h — the MPC horizon solve with FOUR LANES PER AGENT (a DPP quad), for batches that leave SIMDs idle.
//
// Same NLP, same solver, same decisions as mpc_kernels.hip.h (mpc_solve_lane: which see for the algorithm and the reference's
// lines, /root/reference/src/model_predictive_control.cpp:199-346) — the quad is used where the algorithm has independent work:
// the BACKTRACKING LINE SEARCH.  A sweep's candidate rollouts alpha = 1, 1/2, 1/4, ... are independent of each other; the
// one-lane kernel tries them one after the other and a wave pays, in every sweep, for the agent of its 64 that needs the most
// (BASELINE configs[3]: 31 rollouts in the 16 sweeps of the slowest wave).  Here lane r of the quad rolls out alpha = 2^-(4q+r)
// in round q and the quad takes the first one of the sequence that passes — the sequential search's own decision, so iterates,
// sweep counts and results are unchanged — which makes it one rollout per sweep except where more than four step lengths fail.
// The backward sweep is NOT split over the lanes: its operands are 4x4 / 2x4 fp64 blocks, fp64 has no DPP operand form, every
// double fetched from a neighbour lane costs two v_mov_b32_dpp — the issue slots of the fused multiply-add it would save
// (DESIGN.md 6).  All four lanes run it redundantly; nothing is lost, the SIMDs they occupy had no wave at this batch.
//
// Memory: the agent's accepted trajectory (knots, controls, the rollout's trig) lives in an LDS block shared by its quad;
// so do the gains of a sweep (computed by all four lanes, stored by lane 0); each lane keeps only its candidate in private memory,
// and the owner of the accepted candidate publishes it to the block (184 doubles at T = 21, ~2 % of a sweep).  16 agents per wave,
// one wave per workgroup, 74 KB of LDS at T <= 24 — two workgroups per CU (block stride = 2 mod 32 doubles: the 16 quads'
// broadcast reads fall on distinct banks).  With the gains in private memory as well (first version) the kernel was 0.885x of the
// one-lane kernel at configs[3]: four times the waves stream four times the scratch through L2 (profiles/r03/mpc_lanes_ab.txt).
#pragma once
#include <hip/hip_runtime.h>
#include "mpc_kernels.hip.h"

#pragma clang fp contract(fast)      // as in mpc_kernels.hip.h: tolerance parity, fp64

namespace crx {

template <int MAXT>
__device__ __forceinline__ void mpc_solve_quad(const bool live, const int T, const float4 xi, const float4* __restrict__ xr4, const MpcP& p,
                                               double* __restrict__ cur, const int r, const int qbase,
                                               float* __restrict__ so, int& status_out, double& cost_out, float& a0_out, float& d0_out) {
  const int N = T - 1;
  constexpr int UO = 4 * MAXT, TO = 6 * MAXT, GO = 9 * MAXT;   // offsets of U[i][2], TR[i][3] and the gains (k[2], K[12]) behind S[i][4]

  // The agent's ACCEPTED trajectory lives in its LDS block `cur` (knots S[i][4] at 4i, controls U[i][2] at UO + 2i, the trig of
  // the rollout TR[i][3] at TO + 3i), shared by the four lanes.  Each lane keeps its own CANDIDATE of the line search and its
  // own copy of the gains (all four compute the same backward sweep) in private memory.
  double Sc[MAXT][4];     // candidate knots: x, y, yaw, v
  double Uc[MAXT][2];     // candidate stages: delta, a
  double TRc[MAXT][3];    // sin(yaw_i), cos(yaw_i), tan(delta_i) of the candidate rollout: the next backward sweep reuses them
  // The gains of a sweep (feed-forward k[2], feedback K[a + 2*b], b over (x,y,yaw,v,d_prev,a_prev)) go to the block too, at
  // GO + 14 i: all four lanes compute them, lane 0 stores them, every lane's rollout reads them.

  const double dt = p.dt, wb = p.wb;
  const double dt_wb = dt / wb;          // the model uses .../wb*dt once per stage and rollout: one division per solve instead
  const double inv_dt = 1.0 / dt;
  const double lb0 = -p.max_steer, ub0 = p.max_steer;
  const bool small_steer = p.max_steer <= 0.78539816339744830962;   // uniform: every steering angle of a rollout is clamped to it

  // objective of buffer c (states already rolled out there) is accumulated while rolling; this lambda
  // rolls controls U[c] from x0 and returns fg[0]
  auto track = [&](const double* s, int i) -> double {
    const float4 r = xr4[i];
    const double e0 = (double)r.x - s[0], e1 = (double)r.y - s[1], e2 = (double)r.z - s[2], e3 = (double)r.w - s[3];
    return p.qx * e0 * e0 + p.qy * e1 * e1 + p.qyaw * e2 * e2 + p.qv * e3 * e3;
  };
  auto step = [&](const double* s, double d, double a, double* sn, double* tr) {
    double sn_, cs_;
    mpc_sincos(s[2], &sn_, &cs_);
    const double tn_ = small_steer ? mpc_tan_small(d) : mpc_tan(d);
    tr[0] = sn_; tr[1] = cs_; tr[2] = tn_;
    sn[0] = s[0] + s[3] * cs_ * dt;
    sn[1] = s[1] + s[3] * sn_ * dt;
    sn[2] = s[2] + s[3] * tn_ * dt_wb;
    sn[3] = s[3] + a * dt;
  };

  struct StageIn { double s0, s1, s2, s3, sn, cs, tn; float4 r; };
  auto load_stage = [&](int i) -> StageIn {
    return StageIn{cur[4 * i], cur[4 * i + 1], cur[4 * i + 2], cur[4 * i + 3], cur[TO + 3 * i], cur[TO + 3 * i + 1], cur[TO + 3 * i + 2], xr4[i]};
  };

  struct RollIn { double s[4], u0, u1, k0, k1, K[12]; float4 r; };
  auto load_roll = [&](int i) -> RollIn {
    RollIn q;
#pragma unroll
    for (int a = 0; a < 4; ++a) q.s[a] = cur[4 * i + a];
    q.u0 = cur[UO + 2 * i]; q.u1 = cur[UO + 2 * i + 1]; q.k0 = cur[GO + 14 * i]; q.k1 = cur[GO + 14 * i + 1];
#pragma unroll
    for (int a = 0; a < 12; ++a) q.K[a] = cur[GO + 14 * i + 2 + a];
    q.r = xr4[i];
    return q;
  };

  const float4 rN = xr4[N];   // terminal reference: used by every sweep and every rollout

  const double x0d[4] = {(double)xi.x, (double)xi.y, (double)xi.z, (double)xi.w};
  double J = 0.0;
  {                                      // zero initial guess (:266-269), rolled out; projected on the acceleration box of each
                                         // knot (which moves it only if the start speed violates the speed bounds).  Every lane
                                         // of the quad computes it; lane 0 writes the agent's block.
    double xs[4] = {x0d[0], x0d[1], x0d[2], x0d[3]}, pa = 0.0;
    if (r == 0) { cur[0] = xs[0]; cur[1] = xs[1]; cur[2] = xs[2]; cur[3] = xs[3]; }
    for (int i = 0; i < N; ++i) {
      const AccelBox ab = accel_box(p, inv_dt, xs[3]);
      const double a0 = clampd(0.0, ab.lo, ab.hi);
      double cv = p.r_d * 0.0 * 0.0 + p.r_a * a0 * a0;                 // ctrl(i) with delta = 0
      if (i >= 1) { const double dd = 0.0, da = a0 - pa; cv += p.rd_d * dd * dd + p.rd_a * da * da; }
      J += cv;
      if (i >= 1) J += track(xs, i);
      double xn[4], tr[3];
      step(xs, 0.0, a0, xn, tr);
      if (r == 0) {
        cur[UO + 2 * i] = 0.0; cur[UO + 2 * i + 1] = a0;
        cur[TO + 3 * i] = tr[0]; cur[TO + 3 * i + 1] = tr[1]; cur[TO + 3 * i + 2] = tr[2];
        cur[4 * (i + 1)] = xn[0]; cur[4 * (i + 1) + 1] = xn[1]; cur[4 * (i + 1) + 2] = xn[2]; cur[4 * (i + 1) + 3] = xn[3];
      }
      xs[0] = xn[0]; xs[1] = xn[1]; xs[2] = xn[2]; xs[3] = xn[3]; pa = a0;
    }
    J += track(xs, N);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  double mu = 0.0;
  const double mu_min = 1e-6, mu_max = 1e10;
  const int n_gn = 2;
  int gn_left = n_gn, gn_run = n_gn;
  int status = 0, it = 0;
  bool done = !live;

  for (int iter = 0; iter < p.max_iter; ++iter) {
    if (__all(done)) break;
    if (done) continue;
    it = iter;
    const bool exact = gn_left <= 0;
    // ------------------------------------------------------------------ backward sweep
    double lx[4], lp0, lp1;          // V_s
    double Wxx[4][4], Wxp[4][2], Wpp00, Wpp01, Wpp11;  // V_ss (symmetric)
    {
      const float4 r = rN;
      const double s[4] = {cur[4 * N], cur[4 * N + 1], cur[4 * N + 2], cur[4 * N + 3]};
      lx[0] = -2.0 * p.qx * ((double)r.x - s[0]);
      lx[1] = -2.0 * p.qy * ((double)r.y - s[1]);
      lx[2] = -2.0 * p.qyaw * ((double)r.z - s[2]);
      lx[3] = -2.0 * p.qv * ((double)r.w - s[3]);
      lp0 = 0.0; lp1 = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) Wxx[a][b] = 0.0;
        Wxp[a][0] = 0.0; Wxp[a][1] = 0.0;
      }
      Wxx[0][0] = 2.0 * p.qx; Wxx[1][1] = 2.0 * p.qy; Wxx[2][2] = 2.0 * p.qyaw; Wxx[3][3] = 2.0 * p.qv;
      Wpp00 = 0.0; Wpp01 = 0.0; Wpp11 = 0.0;
    }
    double dV1 = 0.0, dV2 = 0.0, gnorm = 0.0;
    // The sweep's operands live in private memory (L2 / HBM latency once only a few waves are still iterating): stage
    // i - 1's knot, trig and reference and stage i - 2's control are requested at the top of stage i and consumed one
    // iteration later.
    StageIn nx = load_stage(N - 1);
    double uc0 = cur[UO + 2 * (N - 1)], uc1 = cur[UO + 2 * (N - 1) + 1];
    double up0 = cur[UO + 2 * (N >= 2 ? N - 2 : 0)], up1 = cur[UO + 2 * (N >= 2 ? N - 2 : 0) + 1];
    for (int i = N - 1; i >= 0; --i) {
      const StageIn in = nx;
      const double ud = uc0, ua = uc1;
      const bool inner = i >= 1;
      const double pd = inner ? up0 : 0.0, pa = inner ? up1 : 0.0;
      // unconditional (clamped index) so that the number of loads in flight is the same on every path: with a branch
      // around them the compiler has to drain the memory queue (vmcnt(0)) before the first use of `in`
      nx = load_stage(i >= 1 ? i - 1 : 0);
      uc0 = up0; uc1 = up1;
      { const int j = i >= 2 ? i - 2 : 0; up0 = cur[UO + 2 * j]; up1 = cur[UO + 2 * j + 1]; }
      const double s[4] = {in.s0, in.s1, in.s2, in.s3};
      const double sn_ = in.sn, cs_ = in.cs;
      const double v = s[3];
      const double tn = in.tn, sec2 = 1.0 + tn * tn;
      const double a02 = -v * sn_ * dt, a03 = cs_ * dt, a12 = v * cs_ * dt, a13 = sn_ * dt, a23 = tn * dt_wb;
      const double bd = v * sec2 * dt_wb;
      // stage cost derivatives
      double l_x[4] = {0.0, 0.0, 0.0, 0.0}, q2[4] = {0.0, 0.0, 0.0, 0.0};
      double l_u0 = 2.0 * p.r_d * ud, l_u1 = 2.0 * p.r_a * ua;
      double l_uu0 = 2.0 * p.r_d, l_uu1 = 2.0 * p.r_a;
      double l_p0 = 0.0, l_p1 = 0.0, l_pp0 = 0.0, l_pp1 = 0.0, l_up0 = 0.0, l_up1 = 0.0;
      if (inner) {
        const float4 r = in.r;
        q2[0] = 2.0 * p.qx; q2[1] = 2.0 * p.qy; q2[2] = 2.0 * p.qyaw; q2[3] = 2.0 * p.qv;
        l_x[0] = -q2[0] * ((double)r.x - s[0]);
        l_x[1] = -q2[1] * ((double)r.y - s[1]);
        l_x[2] = -q2[2] * ((double)r.z - s[2]);
        l_x[3] = -q2[3] * ((double)r.w - s[3]);
        const double dd = ud - pd, da = ua - pa;
        l_u0 += 2.0 * p.rd_d * dd; l_u1 += 2.0 * p.rd_a * da;
        l_p0 = -2.0 * p.rd_d * dd; l_p1 = -2.0 * p.rd_a * da;
        l_uu0 += 2.0 * p.rd_d; l_uu1 += 2.0 * p.rd_a;
        l_pp0 = 2.0 * p.rd_d; l_pp1 = 2.0 * p.rd_a;
        l_up0 = -2.0 * p.rd_d; l_up1 = -2.0 * p.rd_a;
      }
      // Q_s, Q_u
      double Qx[4];
      Qx[0] = l_x[0] + lx[0];
      Qx[1] = l_x[1] + lx[1];
      Qx[2] = l_x[2] + (a02 * lx[0] + a12 * lx[1] + lx[2]);
      Qx[3] = l_x[3] + (a03 * lx[0] + a13 * lx[1] + a23 * lx[2] + lx[3]);
      const double Qp0 = l_p0, Qp1 = l_p1;
      const double Qu0 = l_u0 + bd * lx[2] + lp0;
      const double Qu1 = l_u1 + dt * lx[3] + lp1;
      // M = Wxx*A ; Qxx = l_xx + A'*M.  Wxx is symmetric by construction (mirrored upper triangle), so Qxx is symmetric up to
      // rounding: only its upper triangle is formed (a <= b) and used — 7 rows of products instead of 20, and no averaging of
      // the two halves (the CPU twin forms both and averages them; the difference is a rounding of the last bit).
      double M[4][4];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        M[a][0] = Wxx[a][0];
        M[a][1] = Wxx[a][1];
        M[a][2] = Wxx[a][0] * a02 + Wxx[a][1] * a12 + Wxx[a][2];
        M[a][3] = Wxx[a][0] * a03 + Wxx[a][1] * a13 + Wxx[a][2] * a23 + Wxx[a][3];
      }
      M[3][3] = Wxx[3][0] * a03 + Wxx[3][1] * a13 + Wxx[3][2] * a23 + Wxx[3][3];
      double Qxx[4][4];     // entries a <= b only
#pragma unroll
      for (int b = 0; b < 4; ++b) Qxx[0][b] = M[0][b];
#pragma unroll
      for (int b = 1; b < 4; ++b) Qxx[1][b] = M[1][b];
#pragma unroll
      for (int b = 2; b < 4; ++b) Qxx[2][b] = a02 * M[0][b] + a12 * M[1][b] + M[2][b];
      Qxx[3][3] = a03 * M[0][3] + a13 * M[1][3] + a23 * M[2][3] + M[3][3];
#pragma unroll
      for (int a = 0; a < 4; ++a) Qxx[a][a] += q2[a];
      // G = B'*Wxx + Wpx ; Qux = G*A ; Quu = l_uu + G*B + B'*Wxp + Wpp
      double G[2][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        G[0][b] = bd * Wxx[2][b] + Wxp[b][0];
        G[1][b] = dt * Wxx[3][b] + Wxp[b][1];
      }
      double Qux[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        Qux[a][0] = G[a][0];
        Qux[a][1] = G[a][1];
        Qux[a][2] = G[a][0] * a02 + G[a][1] * a12 + G[a][2];
        Qux[a][3] = G[a][0] * a03 + G[a][1] * a13 + G[a][2] * a23 + G[a][3];
      }
      double Quu00 = l_uu0 + G[0][2] * bd + bd * Wxp[2][0] + Wpp00;
      // the off-diagonal of Q_uu once: its two halves (B'W B and its transpose) are equal up to rounding, the twin averages them
      const double Quu01 = G[0][3] * dt + bd * Wxp[2][1] + Wpp01;
      double Quu11 = l_uu1 + G[1][3] * dt + dt * Wxp[3][1] + Wpp11;
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
        const double e00 = lx[2] * v * dt_wb * 2.0 * tn * sec2;
        const double g0 = Quu00 + e00 + mu, g3 = Quu11 + mu, go = Quu01;
        Qxx[2][2] += lx[0] * (-v * cs_ * dt) + lx[1] * (-v * sn_ * dt);
        const double cross = lx[0] * (-sn_ * dt) + lx[1] * (cs_ * dt);
        Qxx[2][3] += cross;
        Qux[0][3] += lx[2] * sec2 * dt_wb;
        if (g0 > 1e-12 && (hold1 || g0 * g3 - go * go > 1e-12 * g0)) Quu00 += e00;
      }
      const double hod = Quu01;
      const double h00 = Quu00 + mu, h11 = Quu11 + mu;
      // for a Newton step, a trust box around the current controls — with the exact (possibly indefinite) Hessian an
      // unrestricted stage proposes a jump to the far corner of the box, which the line search rejects at every step
      // length, and the solver falls back to linearly converging Gauss-Newton steps: that is the whole tail of the
      // iteration-count distribution.  Gauss-Newton steps are not restricted.
      if (exact) {
        lo0 = fmax(lo0, -kMpcTrustSteer); hi0 = fmin(hi0, kMpcTrustSteer);
        if (lo1 < -kMpcTrustAccel) { lo1 = -kMpcTrustAccel; ab.sp_lo = false; }
        if (hi1 > kMpcTrustAccel) { hi1 = kMpcTrustAccel; ab.sp_hi = false; }
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
        const double den = both ? (h00 * h11 - hod * hod) : (f0 ? h00 : (f1 ? h11 : 1.0));
        const double inv = fast_div(1.0, den);
        const double i00 = both ? h11 * inv : (f0 ? inv : 0.0);
        const double i11 = both ? h00 * inv : (f1 ? inv : 0.0);
        const double i01 = both ? -hod * inv : 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          K[0][b] = -(i00 * Qus[0][b] + i01 * Qus[1][b]);
          K[1][b] = -(i01 * Qus[0][b] + i11 * Qus[1][b]);
        }
        K[0][4] = -(i00 * l_up0); K[1][4] = -(i01 * l_up0);      // Q_us columns 4, 5 = diag(l_up0, l_up1): the zero products are
        K[0][5] = -(i01 * l_up1); K[1][5] = -(i11 * l_up1);      // written out (x*0 and x+0 are not foldable in IEEE arithmetic)
        if (__any(sp)) {          // rare (never on the reference's scenario: 10 km/h against bounds of -20 / +55 km/h)
          if (sp) {
            const double ih = f0 ? fast_div(1.0, h00) : 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
              K[1][b] = (b == 3) ? -inv_dt : 0.0;
              K[0][b] = -(Qus[0][b] + hod * K[1][b]) * ih;
            }
          }
        }
      }
      if (r == 0) { cur[GO + 14 * i] = k0; cur[GO + 14 * i + 1] = k1; }
#pragma unroll
      for (int b = 0; b < 6; ++b) { if (r == 0) { cur[GO + 14 * i + 2 + 2 * b] = K[0][b]; cur[GO + 14 * i + 3 + 2 * b] = K[1][b]; } }
      gnorm = fmax(gnorm, fmax(fabs(k0), fabs(k1)));
      // expected change and value function (unregularised, symmetrised Quu)
      const double Quuk0 = Quu00 * k0 + hod * k1, Quuk1 = hod * k0 + Quu11 * k1;
      dV1 += k0 * Qu0 + k1 * Qu1;
      dV2 += 0.5 * (k0 * Quuk0 + k1 * Quuk1);
      const double t0 = Quuk0 + Qu0, t1 = Quuk1 + Qu1;
      double Vs[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const double qs = (b < 4) ? Qx[b < 4 ? b : 0] : (b == 4 ? Qp0 : Qp1);
        const double uk = (b < 4) ? Qus[0][b] * k0 + Qus[1][b] * k1 : (b == 4 ? l_up0 * k0 : l_up1 * k1);
        Vs[b] = qs + (K[0][b] * t0 + K[1][b] * t1) + uk;
      }
      double Vss[6][6];
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) {
          const double uK = (a < 4) ? Qus[0][a] * K[0][b] + Qus[1][a] * K[1][b] : (a == 4 ? l_up0 * K[0][b] : l_up1 * K[1][b]);
          // V_ss = Q_ss + K'Quu K + K'Q_us + Q_us'K.  The gains solve (Quu + mu I)_FF K_F = -Q_us,F on the free controls
          // (rows of clamped controls are zero), so K'Quu K = -K'Q_us - mu K'K and the three products collapse into
          //   V_ss = Q_ss + Q_us'K - mu K'K
          // (the familiar Q_ss - Q_su Quu^-1 Q_us when mu = 0).  Only the upper triangle is formed and mirrored.
          if (a < 4 && b < 4) Vss[a][b] = Qxx[a < 4 ? a : 0][b < 4 ? b : 0] + uK;
          else if (a == 4 && b == 4) Vss[a][b] = l_pp0 + uK;
          else if (a == 5 && b == 5) Vss[a][b] = l_pp1 + uK;
          else Vss[a][b] = uK;
        }
      if (mu != 0.0) {      // rare: the regularised iterations
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = a; b < 6; ++b) Vss[a][b] -= sp ? 0.0 : mu * (K[0][a] * K[0][b] + K[1][a] * K[1][b]);
      }
      if (__any(sp)) {      // a prescribed feedback row: the identity above does not hold, the general form is evaluated
        if (sp) {
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) {
              const double qk0 = Quu00 * K[0][b] + hod * K[1][b], qk1 = hod * K[0][b] + Quu11 * K[1][b];
              Vss[a][b] += (K[0][a] * qk0 + K[1][a] * qk1) + (K[0][a] * Qus[0][b] + K[1][a] * Qus[1][b]);
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
    }
    if (gnorm < p.tol && mu == 0.0) { status |= 1; done = true; continue; }
    // ------------------------------------------------------------------ forward rollout + line search
    const double aJ = fabs(J);
    const double noise = 1e-12 * (aJ > 1.0 ? aJ : 1.0);
    const bool trust = -(dV1 + dV2) < noise;
    bool accepted = false;
    double alpha = 1.0;                  // the accepted step length
    const int ls_max = exact ? 4 : 10;   // a Newton step that fails down to alpha = 1/8 is dropped for Gauss-Newton ones
    // The line search, four step lengths at a time: lane r of the quad rolls out alpha = 2^-(4q + r) in round q, and the quad
    // takes the FIRST one in the sequence 1, 1/2, 1/4, ... that passes the test — the decision of the sequential search, so the
    // iterates, sweep counts and results are those of the one-lane kernel and of the CPU twin.
    bool pending = true;
    for (int q = 0; 4 * q < 10; ++q) {
      if (!__any(pending)) break;
      if (!pending) continue;
      const int ls = 4 * q + r;
      const double al = __hiloint2double((1023 - ls) << 20, 0);        // 2^-ls, what `alpha *= 0.5` arrives at
      double Jn = 0.0;
      // The candidate rollout carries its state and previous control in registers (they are also written to the lane's private
      // candidate arrays for a later adoption, but never read back here), and requests stage i + 1's operands while stage i computes.
      double xs[4] = {x0d[0], x0d[1], x0d[2], x0d[3]};
      double pnd = 0.0, pna = 0.0, pcd = 0.0, pca = 0.0;       // previous stage's new / current controls
      RollIn nx = load_roll(0);
      for (int i = 0; i < N; ++i) {
        const RollIn in = nx;
        nx = load_roll(i + 1 < N ? i + 1 : N - 1);           // unconditional, clamped: see the backward sweep
        const double d0 = xs[0] - in.s[0], d1 = xs[1] - in.s[1], d2 = xs[2] - in.s[2], d3 = xs[3] - in.s[3];
        const double d4 = (i >= 1) ? pnd - pcd : 0.0;
        const double d5 = (i >= 1) ? pna - pca : 0.0;
        double du0 = al * in.k0;
        du0 += in.K[0] * d0; du0 += in.K[2] * d1; du0 += in.K[4] * d2; du0 += in.K[6] * d3; du0 += in.K[8] * d4; du0 += in.K[10] * d5;
        double du1 = al * in.k1;
        du1 += in.K[1] * d0; du1 += in.K[3] * d1; du1 += in.K[5] * d2; du1 += in.K[7] * d3; du1 += in.K[9] * d4; du1 += in.K[11] * d5;
        const AccelBox nb = accel_box(p, inv_dt, xs[3]);            // the box of a_i at the NEW speed of knot i
        const double nd = clampd(in.u0 + du0, lb0, ub0);
        const double na = clampd(in.u1 + du1, nb.lo, nb.hi);
        Uc[i][0] = nd; Uc[i][1] = na;
        double cv = p.r_d * nd * nd + p.r_a * na * na;              // ctrl(i) of the candidate
        if (i >= 1) {
          const double dd = nd - pnd, da = na - pna;
          cv += p.rd_d * dd * dd + p.rd_a * da * da;
        }
        Jn += cv;
        if (i >= 1) {                                               // track(xs, i)
          const double e0 = (double)in.r.x - xs[0], e1 = (double)in.r.y - xs[1], e2 = (double)in.r.z - xs[2], e3 = (double)in.r.w - xs[3];
          Jn += p.qx * e0 * e0 + p.qy * e1 * e1 + p.qyaw * e2 * e2 + p.qv * e3 * e3;
        }
        double xn[4];
        step(xs, nd, na, xn, TRc[i]);
        Sc[i + 1][0] = xn[0]; Sc[i + 1][1] = xn[1]; Sc[i + 1][2] = xn[2]; Sc[i + 1][3] = xn[3];
        xs[0] = xn[0]; xs[1] = xn[1]; xs[2] = xn[2]; xs[3] = xn[3];
        pnd = nd; pna = na; pcd = in.u0; pca = in.u1;
      }
      {                                                             // track(xs, N), terminal reference kept in registers
        const double e0 = (double)rN.x - xs[0], e1 = (double)rN.y - xs[1], e2 = (double)rN.z - xs[2], e3 = (double)rN.w - xs[3];
        Jn += p.qx * e0 * e0 + p.qy * e1 * e1 + p.qyaw * e2 * e2 + p.qv * e3 * e3;
      }
      const bool pass = ls < ls_max && (Jn < J || (trust && Jn <= J + noise));
      const unsigned m4 = (unsigned)(__builtin_amdgcn_ballot_w64(pass) >> qbase) & 0xFu;     // the quad's four verdicts
      if (m4) {
        const int w = __builtin_ctz(m4);                            // the first step length of the sequence that passes
        __builtin_amdgcn_wave_barrier();                            // every lane of the quad is done reading `cur`
        if (r == w) {                                               // its owner publishes the candidate as the agent's trajectory
          for (int i = 0; i < N; ++i) {
            cur[4 * (i + 1)] = Sc[i + 1][0]; cur[4 * (i + 1) + 1] = Sc[i + 1][1]; cur[4 * (i + 1) + 2] = Sc[i + 1][2]; cur[4 * (i + 1) + 3] = Sc[i + 1][3];
            cur[UO + 2 * i] = Uc[i][0]; cur[UO + 2 * i + 1] = Uc[i][1];
            cur[TO + 3 * i] = TRc[i][0]; cur[TO + 3 * i + 1] = TRc[i][1]; cur[TO + 3 * i + 2] = TRc[i][2];
          }
        }
        J = __shfl(Jn, qbase + w, 64);
        alpha = __hiloint2double((1023 - (4 * q + w)) << 20, 0);
        accepted = true; pending = false;
      } else if (4 * (q + 1) >= ls_max) {
        pending = false;                                            // the sequence is exhausted: no step length passed
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                // the published trajectory is what the next sweep reads
    if (accepted) {
      if (gn_left > 0) gn_left--;
      if (alpha == 1.0) mu *= 0.1;
      if (mu < mu_min) mu = 0.0;
    } else if (exact) {
      gn_run = gn_run * 2 > 16 ? 16 : gn_run * 2;   // every failed Newton attempt doubles the Gauss-Newton run after it
      gn_left = gn_run;
    } else {
      mu = (mu * 10.0 > 1e-3) ? mu * 10.0 : 1e-3;
      if (mu > mu_max) done = true;
    }
    if (iter == p.max_iter - 1) it = p.max_iter;
  }
  status_out = 0; cost_out = 0.0; a0_out = 0.0f; d0_out = 0.0f;
  if (!live) return;
  if (!(status & 1) && !done) it = p.max_iter;
  for (int i = 0; i < T; ++i) {
    const double v = cur[4 * i + 3];
    if (v > p.max_speed + 1e-9 || v < p.min_speed - 1e-9) status |= 2;
    if (so) {
      so[i] = (float)cur[4 * i];
      so[T + i] = (float)cur[4 * i + 1];
      so[2 * T + i] = (float)cur[4 * i + 2];
      so[3 * T + i] = (float)v;
    }
  }
  if (so)
    for (int i = 0; i < N; ++i) {
      so[4 * T + i] = (float)cur[UO + 2 * i];
      so[4 * T + N + i] = (float)cur[UO + 2 * i + 1];
    }
  status_out = status | (it << 8);
  cost_out = J;
  d0_out = (float)cur[UO];
  a0_out = (float)cur[UO + 1];
}


constexpr int mpc_quad_stride(int maxt) { return ((23 * maxt + 31 - 2) / 32) * 32 + 2; }   // doubles per agent block, = 2 (mod 32)

template <int MAXT>
__global__ void __launch_bounds__(64)
mpc_quad_kernel(int n, int T, const float* __restrict__ x0g, const float* __restrict__ xrefg, MpcP p,
                float* __restrict__ solg, int* __restrict__ statusg, double* __restrict__ costg) {
  constexpr int STRIDE = mpc_quad_stride(MAXT);
  extern __shared__ __attribute__((aligned(16))) double s_traj[];      // 16 * STRIDE doubles (74 KB at MAXT = 24: dynamic)
  const int lane = threadIdx.x & 63, r = lane & 3, qbase = lane & ~3;
  const size_t agent = (size_t)blockIdx.x * 16 + (lane >> 2);
  const bool live = agent < (size_t)n;
  const size_t ag = live ? agent : 0;
  const float4* __restrict__ xr4 = reinterpret_cast<const float4*>(xrefg) + ag * (size_t)T;
  const float4 xi = reinterpret_cast<const float4*>(x0g)[ag];
  const size_t nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
  int status; double J; float a0, d0;
  mpc_solve_quad<MAXT>(live, T, xi, xr4, p, s_traj + (size_t)(lane >> 2) * STRIDE, r, qbase,
                       (live && r == 0) ? solg + agent * nv : nullptr, status, J, a0, d0);
  if (!live || r != 0) return;
  if (statusg) statusg[agent] = status;
  if (costg) costg[agent] = J;
}

inline hipError_t mpc_quad_launch(int n, int T, const float* x0, const float* xref, const crx_mpc_params& q, float* sol,
                                  int* status, double* cost, hipStream_t stream) {
  MpcP p;
  p.dt = q.dt; p.wb = q.wb; p.max_steer = q.max_steer; p.max_accel = q.max_accel;
  p.max_speed = q.max_speed; p.min_speed = q.min_speed;
  p.r_a = q.r_a; p.r_d = q.r_delta; p.rd_a = q.rd_a; p.rd_d = q.rd_delta;
  p.qx = q.q_x; p.qy = q.q_y; p.qyaw = q.q_yaw; p.qv = q.q_v; p.tol = q.tol; p.max_iter = q.max_iter;
  const dim3 grid((unsigned)(((size_t)n + 15) / 16)), block(64);
  if (T <= 8) {
    hipLaunchKernelGGL((mpc_quad_kernel<8>), grid, block, 16 * mpc_quad_stride(8) * sizeof(double), stream, n, T, x0, xref, p, sol, status, cost);
  } else if (T <= 24) {
    constexpr size_t lds = 16 * mpc_quad_stride(24) * sizeof(double);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mpc_quad_kernel<24>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((mpc_quad_kernel<24>), grid, block, lds, stream, n, T, x0, xref, p, sol, status, cost);
  } else
    return hipErrorInvalidValue;       // longer horizons: the one-lane kernel (the blocks would not fit two waves per CU)
  return hipGetLastError();
}

}  // namespace crx

#pragma clang fp contract(off)

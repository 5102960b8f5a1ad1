// mpc_ref.cpp — TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline for bench.py).
// Nothing under cpprobotics_amd/ may include, link or call this file.
//
// The reference solves its MPC problem with CppAD + IPOPT
// (/root/reference/src/model_predictive_control.cpp:188-346; both libraries un-vendored, un-pinned
// — readme.md:9, "Ipopt-3.12.13" only in a commented path CMakeLists.txt:19-20 — and the `mpc`
// target is commented out of the build, CMakeLists.txt:76-77).  IPOPT stops on a CPU-time budget
// (`max_cpu_time 0.05`, :328) and its status is discarded (:338-339), so the reference has no
// reproducible answer beyond "a local optimum of the NLP".  The crx engine replaces the solver.
// This file holds
//   (1) oracle_mpc_cost / oracle_mpc_rollout: the NLP itself — objective :199-252 evaluated on a
//       control sequence by rolling the equality constraints :242-245 forward — used by the tests
//       to check optimality against scipy on the exact NLP, and
//   (2) oracle_mpc_solve: a plainly written (dense 6x6, no structure exploited) CPU twin of the
//       engine's algorithm — control-limited DDP with exact second-order terms on the state
//       augmented by the previous control — against which the HIP kernel must agree to 1e-6.
// The PROBLEM (objective, constraint functions, bounds, initial point, output layout) is pinned against the reference's own
// lines: FG_EVAL and mpc_solve compiled unmodified with AD<double> = double and ipopt::solve replaced by a recorder
// (oracle/ref_shim/ref_mpc.cpp, tests/test_oracle_vs_ref.py).  The SOLVER is ours; it is pinned by optimality: KKT residual and
// agreement with scipy.optimize on the same NLP, speed bounds included (tests/test_oracle_mpc.py).  The reference's own answer
// (an IPOPT iterate under a CPU-time budget) is not reproducible by anything.
#include <cmath>
#include <cstring>
#include <vector>

namespace {

int p_n_gn = 2; double p_up = 10.0, p_down = 0.1;  // tuning knobs (set through oracle_mpc_tune)
int p_trust_mode = 0;  // experiment knob (oracle_mpc_trust_mode): 0 = trust box on every Newton sweep (the engine), 1 = only after the first refused Newton step, 2 = doubled after every accepted full step, back to the base after a refused one
thread_local double t_trust_scale = 1.0; thread_local bool t_trust_on = true;
thread_local int t_variant_n_gn = -1; thread_local double t_variant_trust = 1.0;
thread_local int* t_ls_out = nullptr;   // experiment aid (oracle_mpc_ls_profile): candidate rollouts of every sweep   // the variant of a portfolio solve (oracle_mpc_solve_portfolio)
int p_warm = 0;       // experiment knob (oracle_mpc_warm): 0 = the reference's zero initial guess (:266-274), 1 / 2 = see solve_one

struct MpcParams {  // mirrors crx_mpc_params (include/crx.h); defaults = the reference's #defines
  double dt, wb, max_steer, max_accel, max_speed, min_speed;
  double r_a, r_delta, rd_a, rd_delta, q_x, q_y, q_yaw, q_v, tol;
  int max_iter;
  // false (default): the feedback gains reach the rollouts rounded to float, as the engine stores them.  true: they stay double — the
  // INDEPENDENT form (rounds 1-4), kept so that the float rounding is checked against something that does not share it
  // (oracle_mpc_solve_double_gains, tests/test_oracle_mpc.py::test_float_gains_against_double_gains)
  bool double_gains = false;
};

int p_switch_at = -1, p_switch_n_gn = 2; double p_switch_trust = 1.0;   // experiment knob: oracle_mpc_late_switch
double kTrustSteer = 0.4, kTrustAccel = 0.5;   // trust box of a Newton step, see backward() (variables only for the round-4 portfolio experiment: oracle_mpc_trust)
constexpr int NS = 6;  // x, y, yaw, v, previous delta, previous a
constexpr int NU = 2;  // delta, a

// One step of the equality constraints :242-245, plus "remember the control".
void dyn(const MpcParams& p, const double* s, const double* u, double* sn) {
  sn[0] = s[0] + s[3] * std::cos(s[2]) * p.dt;
  sn[1] = s[1] + s[3] * std::sin(s[2]) * p.dt;
  sn[2] = s[2] + s[3] * std::tan(u[0]) / p.wb * p.dt;
  sn[3] = s[3] + u[1] * p.dt;
  sn[4] = u[0];
  sn[5] = u[1];
}

// tracking cost of knot i (i >= 1) :247-250 ; xr = column i of traj_ref
double track(const MpcParams& p, const double* s, const float* xr) {
  const double e0 = (double)xr[0] - s[0], e1 = (double)xr[1] - s[1], e2 = (double)xr[2] - s[2], e3 = (double)xr[3] - s[3];
  return p.q_x * e0 * e0 + p.q_y * e1 * e1 + p.q_yaw * e2 * e2 + p.q_v * e3 * e3;
}

// control cost :203-204 and rate cost :208-209 of stage i
double ctrl(const MpcParams& p, int i, const double* s, const double* u) {
  double c = p.r_delta * u[0] * u[0] + p.r_a * u[1] * u[1];
  if (i >= 1) {
    const double dd = u[0] - s[4], da = u[1] - s[5];
    c += p.rd_delta * dd * dd + p.rd_a * da * da;
  }
  return c;
}

// Roll controls U (N x 2) from x0, fill S ((N+1) x 6), return the objective fg[0].
double rollout(const MpcParams& p, int T, const float* x0, const float* xref, const double* U, double* S) {
  const int N = T - 1;
  for (int k = 0; k < 4; ++k) S[k] = (double)x0[k];
  S[4] = 0.0; S[5] = 0.0;
  double J = 0.0;
  for (int i = 0; i < N; ++i) {
    const double* s = S + NS * i;
    J += ctrl(p, i, s, U + NU * i);
    if (i >= 1) J += track(p, s, xref + 4 * i);
    dyn(p, s, U + NU * i, S + NS * (i + 1));
  }
  J += track(p, S + NS * N, xref + 4 * N);
  return J;
}

inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// The box of the acceleration at a knot whose speed is v.  The reference bounds every speed knot, v in [MIN_SPEED, MAX_SPEED]
// (:298-301); with v+ = v + a*DT (:245) that is exactly a state-dependent box on a, intersected with |a| <= MAX_ACCEL
// (:293-296).  Should the two be incompatible (a start speed outside the speed bounds) the acceleration limits win and the
// violated speed bound is reported in status bit 1.  sp_lo / sp_hi: the respective end of the box is the speed bound's.
inline void accel_box(const MpcParams& p, double v, double* lo, double* hi, bool* sp_lo, bool* sp_hi) {
  const double a_lo = (p.min_speed - v) / p.dt, a_hi = (p.max_speed - v) / p.dt;
  *sp_lo = a_lo > -p.max_accel; *sp_hi = a_hi < p.max_accel;
  *lo = clampd(a_lo, -p.max_accel, p.max_accel);
  *hi = clampd(a_hi, -p.max_accel, p.max_accel);
}

// 2-variable box QP  min 1/2 k'Hk + g'k,  lo <= k <= hi  (H symmetric, possibly indefinite) by
// enumerating the candidate minimisers: the interior stationary point (only if H is positive
// definite), the stationary point of each edge (only if the curvature along it is positive) and the
// four corners.  The global minimiser over the box is always one of them.
// free_[j] = 1 if variable j is not held at a bound in the chosen candidate.
void boxqp2(const double H[4], const double g[2], const double lo[2], const double hi[2], double k[2], int free_[2]) {
  const double hod = 0.5 * (H[1] + H[2]);
  const double det = H[0] * H[3] - hod * hod;
  double best = 1e300;
  k[0] = 0.0; k[1] = 0.0; free_[0] = free_[1] = 0;
  auto consider = [&](double k0, double k1, int f0, int f1) {
    const double obj = 0.5 * (H[0] * k0 * k0 + 2.0 * hod * k0 * k1 + H[3] * k1 * k1) + g[0] * k0 + g[1] * k1;
    if (obj < best) { best = obj; k[0] = k0; k[1] = k1; free_[0] = f0; free_[1] = f1; }
  };
  const double tiny = 1e-12;
  if (H[0] > tiny && det > tiny * H[0]) {   // positive definite: interior point, if inside the box
    const double k0 = -(H[3] * g[0] - hod * g[1]) / det;
    const double k1 = -(-hod * g[0] + H[0] * g[1]) / det;
    if (k0 >= lo[0] && k0 <= hi[0] && k1 >= lo[1] && k1 <= hi[1]) { consider(k0, k1, 1, 1); return; }
  }
  if (H[0] > tiny && H[3] > tiny) {
    // Both diagonal curvatures positive: along each of the four edges the problem is a convex parabola whose minimiser over the edge
    // is the stationary point clamped to it — four candidates cover the whole boundary, corners included; a control counts as free
    // when its unclamped stationary value lies in the closed interval.  The same point and flags as the enumeration below except for
    // exact ties between two different boundary points of equal objective; the rule is applied PER PROBLEM (the kernel does the
    // same per lane since round 4: ADVICE r3), so which candidates a problem sees never depends on its neighbours in a wave.
    for (int b = 0; b < 2; ++b) {
      const double c0 = b ? hi[0] : lo[0];
      const double u1 = -(g[1] + hod * c0) / H[3];
      consider(c0, clampd(u1, lo[1], hi[1]), 0, (u1 >= lo[1] && u1 <= hi[1]) ? 1 : 0);
      const double c1 = b ? hi[1] : lo[1];
      const double u0 = -(g[0] + hod * c1) / H[0];
      consider(clampd(u0, lo[0], hi[0]), c1, (u0 >= lo[0] && u0 <= hi[0]) ? 1 : 0, 0);
    }
    return;
  }
  for (int b = 0; b < 2; ++b) {
    const double c0 = b ? hi[0] : lo[0];
    if (H[3] > tiny) {
      const double k1 = -(g[1] + hod * c0) / H[3];
      if (k1 >= lo[1] && k1 <= hi[1]) consider(c0, k1, 0, 1);
    }
    const double c1 = b ? hi[1] : lo[1];
    if (H[0] > tiny) {
      const double k0 = -(g[0] + hod * c1) / H[0];
      if (k0 >= lo[0] && k0 <= hi[0]) consider(k0, c1, 1, 0);
    }
  }
  for (int b0 = 0; b0 < 2; ++b0)
    for (int b1 = 0; b1 < 2; ++b1) consider(b0 ? hi[0] : lo[0], b1 ? hi[1] : lo[1], 0, 0);
}

struct Work {
  std::vector<double> S, U, Sn, Un, kff, Kfb;
  explicit Work(int T) : S(NS * T), U(NU * (T - 1)), Sn(NS * T), Un(NU * (T - 1)), kff(NU * (T - 1)), Kfb(NU * NS * (T - 1)) {}
};

// Backward pass.  Returns false if some Quu (+ mu I) is not positive definite.
// exact = include the second derivatives of the dynamics (Newton / DDP); otherwise Gauss-Newton (iLQR),
// whose control Hessian is positive definite by construction.
bool backward(const MpcParams& p, int T, const float* xref, const Work& w, bool exact, double mu, double* kff, double* Kfb,
              double* dV1, double* dV2, double* gnorm, double* deficit) {
  const int N = T - 1;
  double Vs[NS], Vss[NS * NS];
  // terminal: tracking cost of knot N
  {
    const double* s = w.S.data() + NS * N;
    const float* xr = xref + 4 * N;
    const double q[4] = {p.q_x, p.q_y, p.q_yaw, p.q_v};
    std::memset(Vs, 0, sizeof(Vs)); std::memset(Vss, 0, sizeof(Vss));
    for (int k = 0; k < 4; ++k) { Vs[k] = -2.0 * q[k] * ((double)xr[k] - s[k]); Vss[k + NS * k] = 2.0 * q[k]; }
  }
  *dV1 = 0.0; *dV2 = 0.0; *gnorm = 0.0;
  const double lb[2] = {-p.max_steer, -p.max_accel}, ub[2] = {p.max_steer, p.max_accel};
  for (int i = N - 1; i >= 0; --i) {
    const double* s = w.S.data() + NS * i;
    const double* u = w.U.data() + NU * i;
    const double c = std::cos(s[2]), sn = std::sin(s[2]), v = s[3], dt = p.dt;
    const double tn = std::tan(u[0]), sec2 = 1.0 + tn * tn;
    double Fs[NS * NS], Fu[NS * NU];  // column-major: Fs[r + NS*c]
    std::memset(Fs, 0, sizeof(Fs)); std::memset(Fu, 0, sizeof(Fu));
    Fs[0 + NS * 0] = 1; Fs[1 + NS * 1] = 1; Fs[2 + NS * 2] = 1; Fs[3 + NS * 3] = 1;
    Fs[0 + NS * 2] = -v * sn * dt; Fs[0 + NS * 3] = c * dt;
    Fs[1 + NS * 2] = v * c * dt;   Fs[1 + NS * 3] = sn * dt;
    Fs[2 + NS * 3] = tn / p.wb * dt;
    Fu[2 + NS * 0] = v * sec2 / p.wb * dt;
    Fu[3 + NS * 1] = dt;
    Fu[4 + NS * 0] = 1; Fu[5 + NS * 1] = 1;
    // stage cost derivatives
    double ls[NS] = {0}, lu[NU], lss[NS * NS] = {0}, luu[NU * NU] = {0}, lus[NU * NS] = {0};
    lu[0] = 2.0 * p.r_delta * u[0]; lu[1] = 2.0 * p.r_a * u[1];
    luu[0] = 2.0 * p.r_delta; luu[3] = 2.0 * p.r_a;
    if (i >= 1) {
      const float* xr = xref + 4 * i;
      const double q[4] = {p.q_x, p.q_y, p.q_yaw, p.q_v};
      for (int k = 0; k < 4; ++k) { ls[k] = -2.0 * q[k] * ((double)xr[k] - s[k]); lss[k + NS * k] = 2.0 * q[k]; }
      const double dd = u[0] - s[4], da = u[1] - s[5];
      lu[0] += 2.0 * p.rd_delta * dd; lu[1] += 2.0 * p.rd_a * da;
      ls[4] = -2.0 * p.rd_delta * dd; ls[5] = -2.0 * p.rd_a * da;
      luu[0] += 2.0 * p.rd_delta; luu[3] += 2.0 * p.rd_a;
      lss[4 + NS * 4] = 2.0 * p.rd_delta; lss[5 + NS * 5] = 2.0 * p.rd_a;
      lus[0 + NU * 4] = -2.0 * p.rd_delta; lus[1 + NU * 5] = -2.0 * p.rd_a;
    }
    // Q-function
    double Qs[NS], Qu[NU], Qss[NS * NS], Quu[NU * NU], Qus[NU * NS];
    double VF[NS * NS], VFu[NS * NU];  // Vss*Fs, Vss*Fu
    for (int a = 0; a < NS; ++a) {
      for (int b = 0; b < NS; ++b) { double t = 0; for (int k = 0; k < NS; ++k) t += Vss[a + NS * k] * Fs[k + NS * b]; VF[a + NS * b] = t; }
      for (int b = 0; b < NU; ++b) { double t = 0; for (int k = 0; k < NS; ++k) t += Vss[a + NS * k] * Fu[k + NS * b]; VFu[a + NS * b] = t; }
    }
    for (int a = 0; a < NS; ++a) { double t = ls[a]; for (int k = 0; k < NS; ++k) t += Fs[k + NS * a] * Vs[k]; Qs[a] = t; }
    for (int a = 0; a < NU; ++a) { double t = lu[a]; for (int k = 0; k < NS; ++k) t += Fu[k + NS * a] * Vs[k]; Qu[a] = t; }
    for (int a = 0; a < NS; ++a)
      for (int b = 0; b < NS; ++b) { double t = lss[a + NS * b]; for (int k = 0; k < NS; ++k) t += Fs[k + NS * a] * VF[k + NS * b]; Qss[a + NS * b] = t; }
    for (int a = 0; a < NU; ++a)
      for (int b = 0; b < NU; ++b) { double t = luu[a + NU * b]; for (int k = 0; k < NS; ++k) t += Fu[k + NS * a] * VFu[k + NS * b]; Quu[a + NU * b] = t; }
    for (int a = 0; a < NU; ++a)
      for (int b = 0; b < NS; ++b) { double t = lus[a + NU * b]; for (int k = 0; k < NS; ++k) t += Fu[k + NS * a] * VF[k + NS * b]; Qus[a + NU * b] = t; }
    double a_lo, a_hi; bool sp_lo, sp_hi;
    accel_box(p, v, &a_lo, &a_hi, &sp_lo, &sp_hi);           // speed bounds of knot i+1 as a box on a_i at the nominal v_i
    double lo[2] = {lb[0] - u[0], a_lo - u[1]}, hi[2] = {ub[0] - u[0], a_hi - u[1]};
    // Active set of a Newton step (projected Newton): a control that rests on a bound the gradient pushes it against
    // stays there — its box collapses to {0} — and the Hessian is judged on the controls that are left.  Without it an
    // indefinite 2x2 Hessian sends a saturated control to the far corner of the trust box, the line search rejects the
    // step at every length and the solver crawls on Gauss-Newton steps (one agent in ~25,000 never converged).
    const bool hold0 = exact && ((lo[0] >= 0.0 && Qu[0] > 0.0) || (hi[0] <= 0.0 && Qu[0] < 0.0));
    const bool hold1 = exact && ((lo[1] >= 0.0 && Qu[1] > 0.0) || (hi[1] <= 0.0 && Qu[1] < 0.0));
    // second-order dynamics terms  Vs' . d2F
    // The curvature the steering input picks up from the dynamics (e00) is left out of a stage whose control Hessian —
    // over the controls not held — it would make indefinite: typically a steering input saturated along the horizon, whose
    // Newton step is a jump to a box corner that the line search rejects.  Leaving it out of a stage that does not need
    // it costs the quadratic convergence (the slowest agent of the BASELINE batch converged linearly at e00/Quu = 0.25
    // per sweep because its HELD acceleration made the 2x2 test fail).  The state-side terms always go in.
    if (exact) {
      const double e00 = Vs[2] * v / p.wb * dt * 2.0 * tn * sec2;
      const double h0 = Quu[0] + e00 + mu, h3 = Quu[3] + mu, hod = 0.5 * (Quu[1] + Quu[2]);
      Qss[2 + NS * 2] += Vs[0] * (-v * c * dt) + Vs[1] * (-v * sn * dt);
      const double cross = Vs[0] * (-sn * dt) + Vs[1] * (c * dt);
      Qss[2 + NS * 3] += cross; Qss[3 + NS * 2] += cross;
      Qus[0 + NU * 3] += Vs[2] * sec2 / p.wb * dt;
      if (h0 > 1e-12 && (hold1 || h0 * h3 - hod * hod > 1e-12 * h0)) Quu[0] += e00;
    }
    // regularised control Hessian must be positive definite
    const double H[4] = {Quu[0] + mu, Quu[1], Quu[2], Quu[3] + mu};
    // A Newton step is searched inside a trust box around the current controls (|d delta| <= 0.4 rad, |d a| <= 0.5 m/s^2): with
    // the exact (possibly indefinite) Hessian an unconstrained stage proposes a jump to the far box corner, which the line
    // search then rejects at every step length, and the solver falls back to linearly converging Gauss-Newton steps — the
    // whole tail of the iteration-count distribution (on the BASELINE batch: 50-iteration cap hit by 3 agents, 22+ by 8;
    // with the trust box every agent converges in <= 21, with the active set above in <= 16).  Gauss-Newton steps (positive definite) are not restricted.
    if (exact) {
      const double tS = t_trust_on ? kTrustSteer * t_variant_trust * t_trust_scale : 1e9, tA = t_trust_on ? kTrustAccel * t_variant_trust * t_trust_scale : 1e9;
      lo[0] = lo[0] < -tS ? -tS : lo[0]; hi[0] = hi[0] > tS ? tS : hi[0];
      if (lo[1] < -tA) { lo[1] = -tA; sp_lo = false; }     // that end of the box is no longer the speed bound's
      if (hi[1] > tA) { hi[1] = tA; sp_hi = false; }
    }
    if (hold0) { lo[0] = 0.0; hi[0] = 0.0; }
    if (hold1) { lo[1] = 0.0; hi[1] = 0.0; }
    double k[2]; int fr[2];
    boxqp2(H, Qu, lo, hi, k, fr);
    (void)deficit;
    // the acceleration rests on a SPEED bound: it is then a function of the state, a = (v_bound - v)/DT, i.e. a feedback row
    // -1/DT on v (the next knot's speed stays on the bound whatever v does), and the steering gains see that row
    const bool sp = !fr[1] && ((k[1] == hi[1] && sp_hi) || (k[1] == lo[1] && sp_lo));
    double K[NU * NS];
    std::memset(K, 0, sizeof(K));
    if (sp) {
      K[1 + NU * 3] = -1.0 / dt;
      if (fr[0]) {
        const double hod = 0.5 * (H[1] + H[2]);
        for (int b = 0; b < NS; ++b) K[0 + NU * b] = -(Qus[0 + NU * b] + hod * K[1 + NU * b]) / H[0];
      }
    } else if (fr[0] && fr[1]) {
      const double hod = 0.5 * (H[1] + H[2]);
      const double det = H[0] * H[3] - hod * hod;
      for (int b = 0; b < NS; ++b) {
        K[0 + NU * b] = -(H[3] * Qus[0 + NU * b] - hod * Qus[1 + NU * b]) / det;
        K[1 + NU * b] = -(-hod * Qus[0 + NU * b] + H[0] * Qus[1 + NU * b]) / det;
      }
    } else if (fr[0]) {
      for (int b = 0; b < NS; ++b) K[0 + NU * b] = -Qus[0 + NU * b] / H[0];
    } else if (fr[1]) {
      for (int b = 0; b < NS; ++b) K[1 + NU * b] = -Qus[1 + NU * b] / H[3];
    }
    kff[NU * i] = k[0]; kff[NU * i + 1] = k[1];
    std::memcpy(Kfb + NU * NS * i, K, sizeof(K));
    // the feedback gains are handed to the rollouts rounded to FLOAT, as the engine stores them (csrc/mpc_kernels.hip.h: Kf — 41 % of the
    // solver's memory traffic as doubles); the backward sweep itself keeps using K in double.  The gains only steer the candidates, the
    // fixed point is decided by the feed-forward k: on 4 x 8,192 problems 2 sweep counts move by one, no solution float by more than one ulp.
    if (!p.double_gains)
      for (int q = 0; q < NU * NS; ++q) Kfb[NU * NS * i + q] = (double)(float)K[q];
    const double a0 = std::fabs(k[0]), a1 = std::fabs(k[1]);
    if (a0 > *gnorm) *gnorm = a0;
    if (a1 > *gnorm) *gnorm = a1;
    // expected change and value function (unregularised Quu)
    double Quuk[2] = {Quu[0] * k[0] + Quu[2] * k[1], Quu[1] * k[0] + Quu[3] * k[1]};
    *dV1 += k[0] * Qu[0] + k[1] * Qu[1];
    *dV2 += 0.5 * (k[0] * Quuk[0] + k[1] * Quuk[1]);
    for (int a = 0; a < NS; ++a) {
      Vs[a] = Qs[a] + (K[0 + NU * a] * Quuk[0] + K[1 + NU * a] * Quuk[1]) + (K[0 + NU * a] * Qu[0] + K[1 + NU * a] * Qu[1]) +
              (Qus[0 + NU * a] * k[0] + Qus[1 + NU * a] * k[1]);
    }
    // V_ss = Q_ss + K'Quu K + K'Q_us + Q_us'K.  The gains solve (Quu + mu I)_FF K_F = -Q_us,F on the free controls (rows
    // of clamped controls are zero), so K'Quu K = -K'Q_us - mu K'K and the three products collapse into
    //   V_ss = Q_ss + Q_us'K - mu K'K      (= Q_ss - Q_su Quu^-1 Q_us when mu = 0);
    // the upper triangle is formed and mirrored (Q_ss symmetrised), as the kernel does.
    // With a prescribed feedback row (sp) that identity does not hold and the general form is evaluated.
    const double qod = 0.5 * (Quu[1] + Quu[2]);
    for (int a = 0; a < NS; ++a)
      for (int b = a; b < NS; ++b) {
        double v = 0.5 * (Qss[a + NS * b] + Qss[b + NS * a]) + (Qus[0 + NU * a] * K[0 + NU * b] + Qus[1 + NU * a] * K[1 + NU * b]);
        if (sp) {
          const double qk0 = Quu[0] * K[0 + NU * b] + qod * K[1 + NU * b], qk1 = qod * K[0 + NU * b] + Quu[3] * K[1 + NU * b];
          v += (K[0 + NU * a] * qk0 + K[1 + NU * a] * qk1) + (K[0 + NU * a] * Qus[0 + NU * b] + K[1 + NU * a] * Qus[1 + NU * b]);
        } else if (mu != 0.0) {
          v -= mu * (K[0 + NU * a] * K[0 + NU * b] + K[1 + NU * a] * K[1 + NU * b]);
        }
        Vss[a + NS * b] = v; Vss[b + NS * a] = v;
      }
  }
  return true;
}

int solve_one(const MpcParams& p, int T, const float* x0, const float* xref, float* sol, double* cost_out, int* iters_out,
              double* trace = nullptr) {
  const int N = T - 1;
  Work w(T);
  std::fill(w.U.begin(), w.U.end(), 0.0);                    // zero initial guess, :266-269 ...
  {                                                          // ... projected on the bounds (it moves only if the start speed violates them)
    double s[NS] = {(double)x0[0], (double)x0[1], (double)x0[2], (double)x0[3], 0.0, 0.0}, sn[NS];
    for (int i = 0; i < N; ++i) {
      double a_lo, a_hi; bool sp_lo, sp_hi;
      accel_box(p, s[3], &a_lo, &a_hi, &sp_lo, &sp_hi);
      double a_w = 0.0, d_w = 0.0;
      if (p_warm >= 1) a_w = ((double)xref[4 * (i + 1) + 3] - s[3]) / p.dt;                     // reach the reference speed of knot i+1
      if (p_warm >= 2 && std::fabs(s[3]) > 0.5) {                                               // reach the reference heading of knot i+1
        const double tn = ((double)xref[4 * (i + 1) + 2] - s[2]) * p.wb / (s[3] * p.dt);
        d_w = clampd(std::atan(tn), -p.max_steer, p.max_steer);
      }
      if (p_warm == 3 && std::fabs(s[3]) > 0.5) {                                               // heading rate of the REFERENCE (feed-forward)
        const double tn = ((double)xref[4 * (i + 1) + 2] - (double)xref[4 * i + 2]) * p.wb / (s[3] * p.dt);
        d_w = clampd(std::atan(tn), -p.max_steer, p.max_steer);
      }
      w.U[NU * i + 0] = d_w;
      w.U[NU * i + 1] = clampd(a_w, a_lo, a_hi);
      dyn(p, s, w.U.data() + NU * i, sn);
      std::memcpy(s, sn, sizeof(s));
    }
  }
  double J = rollout(p, T, x0, xref, w.U.data(), w.S.data());
  double mu = 0.0;
  const double mu_min = 1e-6, mu_max = 1e10;
  int status = 0, it = 0;
  const int n_gn = t_variant_n_gn >= 0 ? t_variant_n_gn : p_n_gn;
  const double lb[2] = {-p.max_steer, -p.max_accel}, ub[2] = {p.max_steer, p.max_accel};
  t_trust_scale = 1.0; t_trust_on = (p_trust_mode != 1);
  int gn_left = n_gn;   // Gauss-Newton iterations still to do before the next exact (Newton) attempt
  int gn_run = n_gn;    // ... and how many follow a failed one: doubles (up to 16) with every failure
  const double variant_trust_in = t_variant_trust;
  for (it = 0; it < p.max_iter; ++it) {
    if (it == p_switch_at) { gn_left = gn_run = p_switch_n_gn; t_variant_trust = p_switch_trust; }
    const bool exact = gn_left <= 0;
    double dV1, dV2, gnorm, deficit = 0.0;
    bool ok = backward(p, T, xref, w, exact, mu, w.kff.data(), w.Kfb.data(), &dV1, &dV2, &gnorm, &deficit);
    if (trace) { trace[4 * it + 0] = J; trace[4 * it + 1] = ok ? gnorm : -1.0; trace[4 * it + 2] = mu; trace[4 * it + 3] = exact ? -1.0 : 0.0; }
    if (!ok) {
      const double m1 = mu * p_up, m2 = mu + 2.0 * deficit;
      mu = m1 > m2 ? m1 : m2;
      if (mu < mu_min) mu = mu_min;
      if (mu > mu_max) break;
      continue;
    }
    if (gnorm < p.tol && mu == 0.0) { status |= 1; break; }
    const double noise = 1e-12 * (std::fabs(J) > 1.0 ? std::fabs(J) : 1.0);
    const bool trust = -(dV1 + dV2) < noise;
    bool accepted = false;
    double alpha = 1.0;
    const int ls_max = exact ? 4 : 10;   // a Newton step that fails down to alpha = 1/8 is dropped for Gauss-Newton ones
    for (int ls = 0; ls < ls_max; ++ls, alpha *= 0.5) {
      std::memcpy(w.Sn.data(), w.S.data(), sizeof(double) * NS);
      for (int i = 0; i < N; ++i) {
        const double* s = w.S.data() + NS * i;
        double* sn = w.Sn.data() + NS * i;
        double a_lo, a_hi; bool sp_lo, sp_hi;
        accel_box(p, sn[3], &a_lo, &a_hi, &sp_lo, &sp_hi);           // the box of a_i at the NEW speed of knot i
        const double blo[2] = {lb[0], a_lo}, bhi[2] = {ub[0], a_hi};
        for (int a = 0; a < NU; ++a) {
          double du = alpha * w.kff[NU * i + a];
          for (int b = 0; b < NS; ++b) du += w.Kfb[NU * NS * i + a + NU * b] * (sn[b] - s[b]);
          w.Un[NU * i + a] = clampd(w.U[NU * i + a] + du, blo[a], bhi[a]);
        }
        dyn(p, sn, w.Un.data() + NU * i, sn + NS);
      }
      const double Jn = rollout(p, T, x0, xref, w.Un.data(), w.Sn.data());
      if (Jn < J || (trust && Jn <= J + noise)) {
        w.U.swap(w.Un); w.S.swap(w.Sn); J = Jn; accepted = true;
        if (trace) trace[4 * it + 3] = exact ? -alpha : alpha;
        break;
      }
    }
    if (p_trust_mode == 2 && exact) t_trust_scale = (accepted && alpha == 1.0) ? (t_trust_scale * 2.0 > 8.0 ? 8.0 : t_trust_scale * 2.0) : 1.0;
    if (p_trust_mode == 1 && exact && !accepted) t_trust_on = true;
    if (t_ls_out) { int used = 0; double a_ = 1.0; while (a_ > alpha) { a_ *= 0.5; ++used; } t_ls_out[it] = accepted ? used + 1 : ls_max; }
    if (accepted) {
      if (gn_left > 0) gn_left--;
      mu = (alpha == 1.0) ? mu * p_down : mu;
      if (mu < mu_min) mu = 0.0;
    } else if (exact) {
      gn_run = gn_run * 2 > 16 ? 16 : gn_run * 2;
      gn_left = gn_run;       // the Newton model was not trustworthy here: go back to Gauss-Newton for a (growing) while
    } else {
      mu = mu * p_up > 1e-3 ? mu * p_up : 1e-3;
      if (mu > mu_max) break;
    }
  }
  t_variant_trust = variant_trust_in;
  for (int i = 0; i < T; ++i) {
    const double v = w.S[NS * i + 3];
    if (v > p.max_speed + 1e-9 || v < p.min_speed - 1e-9) status |= 2;
    sol[i] = (float)w.S[NS * i + 0];           // x_start   :54
    sol[T + i] = (float)w.S[NS * i + 1];       // y_start
    sol[2 * T + i] = (float)w.S[NS * i + 2];   // yaw_start
    sol[3 * T + i] = (float)w.S[NS * i + 3];   // v_start
  }
  for (int i = 0; i < N; ++i) {
    sol[4 * T + i] = (float)w.U[NU * i + 0];       // delta_start :59
    sol[4 * T + N + i] = (float)w.U[NU * i + 1];   // a_start     :60
  }
  if (cost_out) *cost_out = J;
  if (iters_out) *iters_out = it;
  return status | (it << 8);
}

}  // namespace

extern "C" {

// params: 15 doubles in crx_mpc_params order followed by max_iter.
static MpcParams unpack(const double* pp, int max_iter) {
  MpcParams p;
  p.dt = pp[0]; p.wb = pp[1]; p.max_steer = pp[2]; p.max_accel = pp[3]; p.max_speed = pp[4]; p.min_speed = pp[5];
  p.r_a = pp[6]; p.r_delta = pp[7]; p.rd_a = pp[8]; p.rd_delta = pp[9];
  p.q_x = pp[10]; p.q_y = pp[11]; p.q_yaw = pp[12]; p.q_v = pp[13]; p.tol = pp[14];
  p.max_iter = max_iter;
  return p;
}

// x0: n x 4; xref: n x (4T) column-major 4 x T; sol: n x (4T + 2(T-1)); agents [a0,a1).
void oracle_mpc_solve(int n, int T, const float* x0, const float* xref, const double* params, int max_iter,
                      float* sol, int* status, double* cost, int a0, int a1) {
  const MpcParams p = unpack(params, max_iter);
  const int nv = 4 * T + 2 * (T - 1);
  for (int k = a0; k < a1; ++k) {
    double J; int it;
    const int st = solve_one(p, T, x0 + 4 * k, xref + 4 * (size_t)T * k, sol + (size_t)nv * k, &J, &it);
    if (status) status[k] = st;
    if (cost) cost[k] = J;
  }
}

// The same solve with the feedback gains kept in double (MpcParams::double_gains): what the float-gain twin is bounded against.
void oracle_mpc_solve_double_gains(int n, int T, const float* x0, const float* xref, const double* params, int max_iter,
                                   float* sol, int* status, double* cost, int a0, int a1) {
  MpcParams p = unpack(params, max_iter);
  p.double_gains = true;
  const int nv = 4 * T + 2 * (T - 1);
  for (int k = a0; k < a1; ++k) {
    double J; int it;
    const int st = solve_one(p, T, x0 + 4 * k, xref + 4 * (size_t)T * k, sol + (size_t)nv * k, &J, &it);
    if (status) status[k] = st;
    if (cost) cost[k] = J;
  }
}

// The four-variant portfolio of crx_mpc_solve_portfolio_batch_dev (csrc/mpc_kernels.hip.h: mpc_variant): variant r = (leading
// Gauss-Newton sweeps, trust-box scale) = (2, 1) | (3, 1) | (2, 2) | (1, 2); the agent's answer is the converged variant with the
// fewest sweeps (ties: lowest r), variant 0's iterate if none converges.  The kernel runs the variants in lockstep on a quad of lanes;
// here they run one after the other.  status: as oracle_mpc_solve, plus the winning variant in bits 2-3.
void oracle_mpc_solve_portfolio(int n, int T, const float* x0, const float* xref, const double* params, int max_iter,
                                float* sol, int* status, double* cost, int a0, int a1) {
  const MpcParams p = unpack(params, max_iter);
  const int nv = 4 * T + 2 * (T - 1);
  static const int kGn[4] = {2, 3, 2, 1};
  static const double kTr[4] = {1.0, 1.0, 2.0, 2.0};
  std::vector<float> cand(nv);
  for (int k = a0; k < a1; ++k) {
    int best_st = 0, best_it = 1 << 30; double best_J = 0.0; bool have = false;
    for (int r = 0; r < 4; ++r) {
      t_variant_n_gn = kGn[r]; t_variant_trust = kTr[r];
      double J; int it;
      const int st = solve_one(p, T, x0 + 4 * k, xref + 4 * (size_t)T * k, cand.data(), &J, &it);
      const bool conv = (st & 1) != 0;
      const bool take = conv ? (!have || it < best_it) : (r == 0);          // the first variant stands in until a converged one appears
      if (take && (conv || !have)) {
        std::memcpy(sol + (size_t)nv * k, cand.data(), sizeof(float) * nv);
        best_st = (st & ~0xC) | (r << 2); best_J = J;
        if (conv) { best_it = it; have = true; }
      }
    }
    t_variant_n_gn = -1; t_variant_trust = 1.0;
    if (status) status[k] = best_st;
    if (cost) cost[k] = best_J;
  }
}

void oracle_mpc_tune(int n_gn, double up, double down) { p_n_gn = n_gn; p_up = up; p_down = down; }
void oracle_mpc_warm(int mode) { p_warm = mode; }
// experiment aid: one agent with variant (n_gn, trust scale); ls[i] = candidate rollouts of sweep i (0 for a sweep that only checks
// convergence); returns the status word
int oracle_mpc_ls_profile(int T, const float* x0, const float* xref, const double* params, int max_iter, int n_gn, double trust, int* ls) {
  const MpcParams p = unpack(params, max_iter);
  std::vector<float> sol(4 * T + 2 * (T - 1));
  for (int i = 0; i < max_iter; ++i) ls[i] = 0;
  t_variant_n_gn = n_gn; t_variant_trust = trust; t_ls_out = ls;
  double J; int it;
  const int st = solve_one(p, T, x0, xref, sol.data(), &J, &it);
  t_variant_n_gn = -1; t_variant_trust = 1.0; t_ls_out = nullptr;
  return st;
}
void oracle_mpc_trust_mode(int mode) { p_trust_mode = mode; }
// experiment (scripts/experiments/mpc_twin_late_switch.py): from sweep `at` on the solve continues as variant (n_gn, trust) — what an
// idle lane adopting a straggler's iterate would run; at < 0: off
void oracle_mpc_late_switch(int at, int n_gn, double trust) { p_switch_at = at; p_switch_n_gn = n_gn; p_switch_trust = trust; }
void oracle_mpc_trust(double steer, double accel) { kTrustSteer = steer; kTrustAccel = accel; }

// Debug aid for the tests: per-iteration (J, max|k|, mu, accepted alpha), 4 doubles x max_iter.
int oracle_mpc_trace(int T, const float* x0, const float* xref, const double* params, int max_iter, double* trace) {
  const MpcParams p = unpack(params, max_iter);
  std::vector<float> sol(4 * T + 2 * (T - 1));
  double J; int it;
  return solve_one(p, T, x0, xref, sol.data(), &J, &it, trace);
}

// The NLP objective fg[0] (:199-252) of a control sequence U (N x 2: delta, a), states by rollout;
// S_out ((N+1) x 6) optional.
double oracle_mpc_cost(int T, const float* x0, const float* xref, const double* params, const double* U, double* S_out) {
  const MpcParams p = unpack(params, 0);
  std::vector<double> S(NS * T);
  const double J = rollout(p, T, x0, xref, U, S.data());
  if (S_out) std::memcpy(S_out, S.data(), sizeof(double) * NS * T);
  return J;
}

}  // extern "C"

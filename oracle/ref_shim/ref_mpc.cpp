// ref_mpc.cpp — TEST INFRASTRUCTURE.  The reference's MPC lines compiled as they are
// (/root/reference/src/model_predictive_control.cpp:26-48 defines, :50-60 layout globals, :69-81 update, :107-186
// calc_nearest_index / calc_ref_trajectory / smooth_yaw, :188-346 FG_EVAL + mpc_solve, :349-360 + :372-385 mpc_simulation),
// with CppAD/IPOPT replaced by oracle/ref_shim/cppad_standin.h.  Built twice: the reference's own horizon macro T = 6 (symbols
// ref_mpc6_*) and the BASELINE horizon T = 21 (-DREF_MPC_T=21, symbols ref_mpc21_*).
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <string>
#include <vector>
#include <Eigen/Eigen>
#include "cppad_standin.h"
#include "cubic_spline.h"
#include "motion_model.h"
#include "cpprobotics_types.h"

#ifndef REF_MPC_T
#define REF_MPC_T 6
#endif
#define NX 4                      // :23
#define T REF_MPC_T               // :24 is `#define T 6`
#include "mpc_defs.inc"

#if REF_MPC_T == 6
#define SYM(name) ref_mpc6_##name
namespace ref_mpc6 {
#else
#define SYM(name) ref_mpc21_##name
namespace ref_mpc21 {
#endif

#include "mpc_globals.inc"
#include "mpc_update.inc"
#include "mpc_ref_traj.inc"
#include "mpc_nlp.inc"

static M_XREF load_xref(const float* p) { M_XREF m; std::memcpy(m.data(), p, sizeof(float) * NX * T); return m; }

extern "C" {

int SYM(horizon)(void) { return T; }
void SYM(layout)(int* out) { out[0] = x_start; out[1] = y_start; out[2] = yaw_start; out[3] = v_start; out[4] = delta_start; out[5] = a_start; }

// FG_EVAL::operator() at `vars` (n_vars doubles): fg[0] = cost, fg[1..4T] = constraint functions
void SYM(fg_eval)(const float* xref, const double* vars, double* fg) {
  FG_EVAL f(load_xref(xref));
  const int n_vars = T * 4 + (T - 1) * 2, n_fg = 1 + T * 4;
  std::vector<double> v(vars, vars + n_vars), out(n_fg, 0.0);
  f(out, v);
  std::memcpy(fg, out.data(), sizeof(double) * n_fg);
}

// mpc_solve(): what it hands to the solver (initial point, bounds) and what it returns from the solver's answer.
// solver may be NULL (the answer is then all zeros).  xi, xl, xu: n_vars; gl, gu: 4T; result: n_vars floats.
void SYM(solve)(const float* x0, const float* xref, CppAD::ipopt::SolverFn solver, double* xi, double* xl, double* xu, double* gl, double* gu,
                float* result, char* options, int options_cap) {
  CppAD::ipopt::solver() = solver;
  State s(x0[0], x0[1], x0[2], x0[3]);
  Vec_f r = mpc_solve(s, load_xref(xref));
  const CppAD::ipopt::Capture& c = CppAD::ipopt::capture();
  if (xi) std::memcpy(xi, c.xi.data(), 8 * c.xi.size());
  if (xl) std::memcpy(xl, c.xl.data(), 8 * c.xl.size());
  if (xu) std::memcpy(xu, c.xu.data(), 8 * c.xu.size());
  if (gl) std::memcpy(gl, c.gl.data(), 8 * c.gl.size());
  if (gu) std::memcpy(gu, c.gu.data(), 8 * c.gu.size());
  if (result) std::memcpy(result, r.data(), 4 * r.size());
  if (options && options_cap > 0) { std::strncpy(options, c.options.c_str(), options_cap - 1); options[options_cap - 1] = 0; }
  CppAD::ipopt::solver() = nullptr;
}

// the traj_ref of the FG_EVAL a solver callback is running for (so that the plugged-in solver sees what IPOPT would: x0 through
// the constraint bounds gl[x_start..], the reference trajectory through fg_eval)
void SYM(context_xref)(const void* ctx, float* xref) { std::memcpy(xref, static_cast<const FG_EVAL*>(ctx)->traj_ref.data(), sizeof(float) * NX * T); }

void SYM(update)(int n, float* state, const float* acc, const float* delta) {
  for (int a = 0; a < n; ++a) {
    State st(state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]);
    update(st, acc[a], delta[a]);
    state[4 * a] = st.x; state[4 * a + 1] = st.y; state[4 * a + 2] = st.yaw; state[4 * a + 3] = st.v;
  }
}

void SYM(nearest_index_window)(int n, const float* state, int nc, const float* cx, const float* cy, const float* cyaw, const int* pind, int* ind) {
  Vec_f vx(cx, cx + nc), vy(cy, cy + nc), vyaw(cyaw, cyaw + nc);
  for (int a = 0; a < n; ++a) {
    State st(state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]);
    ind[a] = calc_nearest_index(st, vx, vy, vyaw, pind[a]);
  }
}

// calc_ref_trajectory per agent on a shared course; xref [n][4*T] column-major; target_ind in/out
void SYM(calc_ref_trajectory)(int n, const float* state, int nc, const float* cx, const float* cy, const float* cyaw, const float* ck, const float* sp,
                              float dl, int* target_ind, float* xref) {
  Vec_f vx(cx, cx + nc), vy(cy, cy + nc), vyaw(cyaw, cyaw + nc), vk(ck, ck + nc), vsp(sp, sp + nc);
  for (int a = 0; a < n; ++a) {
    State st(state[4 * a], state[4 * a + 1], state[4 * a + 2], state[4 * a + 3]);
    M_XREF m;
    calc_ref_trajectory(st, vx, vy, vyaw, vk, vsp, dl, target_ind[a], m);
    std::memcpy(xref + (size_t)a * NX * T, m.data(), sizeof(float) * NX * T);
  }
}

void SYM(smooth_yaw)(int nc, float* cyaw) { Vec_f v(cyaw, cyaw + nc); smooth_yaw(v); std::memcpy(cyaw, v.data(), 4 * (size_t)nc); }

// mpc_simulation :348-385 with the plugged-in solver, at most max_ticks passes.  Starts where the reference starts (course point 0).
// cyaw_smoothed (nc, may be NULL) receives the course headings after smooth_yaw.  traj [max_ticks][4].
int SYM(simulation)(int max_ticks, int nc, const float* cx_, const float* cy_, const float* cyaw_, const float* ck_, const float* sp_, float goal_x,
                    float goal_y, CppAD::ipopt::SolverFn solver, float* traj, float* controls, float* cyaw_smoothed) {
  CppAD::ipopt::solver() = solver;
  Vec_f cx(cx_, cx_ + nc), cy(cy_, cy_ + nc), cyaw(cyaw_, cyaw_ + nc), ck(ck_, ck_ + nc), speed_profile(sp_, sp_ + nc);
  Poi_f goal{{goal_x, goal_y}};
#include "mpc_sim_setup.inc"
  if (cyaw_smoothed) std::memcpy(cyaw_smoothed, cyaw.data(), 4 * (size_t)nc);
  M_XREF xref;                                                    // :369
  int ticks = 0;
  std::streambuf* keep = std::cout.rdbuf(nullptr);
  for (int tick = 0; tick < max_ticks; ++tick) {                   // `while (MAX_TIME >= iter_count)` :371 (iter_count never advances)
    ticks = tick + 1;
    bool reached = true;
    do {
#include "mpc_sim_body.inc"
      if (controls) { controls[2 * tick] = output[a_start]; controls[2 * tick + 1] = steer; }
      reached = false;
    } while (0);
    if (traj) { float* h = traj + (size_t)tick * 4; h[0] = state.x; h[1] = state.y; h[2] = state.yaw; h[3] = state.v; }
    if (reached) break;
  }
  std::cout.rdbuf(keep);
  CppAD::ipopt::solver() = nullptr;
  return ticks;
}

// main() :473-486 (ds = 1.0)
int SYM(main_course)(const float* wx_, const float* wy_, int nx, float* cx, float* cy, float* cyaw, float* ck, int cap) {
  Vec_f wx(wx_, wx_ + nx), wy(wy_, wy_ + nx);
#include "mpc_main_course.inc"
  const int k = (int)r_x.size();
  for (int i = 0; i < k && i < cap; ++i) { cx[i] = r_x[i]; cy[i] = r_y[i]; cyaw[i] = ryaw[i]; ck[i] = rcurvature[i]; }
  return k;
}

}  // extern "C"
}  // namespace

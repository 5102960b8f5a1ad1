// cppad_standin.h — TEST INFRASTRUCTURE.  What the reference's MPC lines need from <cppad/cppad.hpp> and
// <cppad/ipopt/solve.hpp> (un-vendored, absent from this image) to COMPILE AND BE EXAMINED:
//   * AD<double> is double, CppAD::pow/sin/cos/tan are libm's — FG_EVAL::operator() (:199-252) then evaluates the cost fg[0]
//     and the constraint functions fg[1..] of the reference's NLP in plain double arithmetic;
//   * CppAD::ipopt::solve(...) does not solve anything: it records the problem mpc_solve (:255-346) hands to IPOPT (initial
//     point, variable bounds, constraint bounds, the options string) in ref_mpc_capture(), then asks a registered callback —
//     the test plugs in the oracle's solver — for the solution (zeros if none is registered).
#pragma once
#include <cmath>
#include <cstddef>
#include <string>
#include <vector>

#define CPPAD_TESTVECTOR(T) std::vector<T>

namespace CppAD {
template <class B> using AD = B;
inline double pow(double x, int e) { return std::pow(x, e); }
inline double pow(double x, double e) { return std::pow(x, e); }
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double tan(double x) { return std::tan(x); }
// CppAD also defines the standard math functions for the base type float (cppad/base_float.hpp,
// CPPAD_STANDARD_MATH_UNARY(float, tan) = `inline float tan(const float& x) { return std::tan(x); }`), which is what
// update() :76 calls with its float `delta`
inline float sin(float x) { return std::sin(x); }
inline float cos(float x) { return std::cos(x); }
inline float tan(float x) { return std::tan(x); }

namespace ipopt {

struct Capture {
  std::string options;
  std::vector<double> xi, xl, xu, gl, gu;
  int calls = 0;
};
inline Capture& capture() { static Capture c; return c; }
// the plugged-in solver: (n_vars, n_constraints, xi, xl, xu, gl, gu, context, x_out)
typedef void (*SolverFn)(int, int, const double*, const double*, const double*, const double*, const double*, const void*, double*);
inline SolverFn& solver() { static SolverFn f = nullptr; return f; }
inline const void*& solver_context() { static const void* p = nullptr; return p; }

template <class Dvector>
struct solve_result {
  enum status_type { not_defined, success, maxiter_exceeded, stop_at_tiny_step, stop_at_acceptable_point, local_infeasibility,
                     user_requested_stop, feasible_point_found, diverging_iterates, restoration_failure,
                     error_in_step_computation, invalid_number_detected, too_few_degrees_of_freedom, internal_error, unknown };
  status_type status = not_defined;
  Dvector x, zl, zu, g, lambda;
  double obj_value = 0;
};

template <class Dvector, class FG_eval>
void solve(const std::string& options, const Dvector& xi, const Dvector& xl, const Dvector& xu, const Dvector& gl, const Dvector& gu,
           FG_eval& fg_eval, solve_result<Dvector>& solution) {
  Capture& c = capture();
  c.options = options;
  c.xi.assign(xi.begin(), xi.end()); c.xl.assign(xl.begin(), xl.end()); c.xu.assign(xu.begin(), xu.end());
  c.gl.assign(gl.begin(), gl.end()); c.gu.assign(gu.begin(), gu.end());
  ++c.calls;
  solution.x = Dvector(xi.size());
  for (std::size_t i = 0; i < xi.size(); ++i) solution.x[i] = 0.0;
  if (solver()) {
    solver_context() = &fg_eval;
    solver()((int)xi.size(), (int)gl.size(), c.xi.data(), c.xl.data(), c.xu.data(), c.gl.data(), c.gu.data(), &fg_eval, &solution.x[0]);
  }
  solution.status = solve_result<Dvector>::success;
}

}  // namespace ipopt
}  // namespace CppAD

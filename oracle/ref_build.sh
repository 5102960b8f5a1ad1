#!/bin/bash
# oracle/ref_build.sh — TEST INFRASTRUCTURE.  Builds oracle/_ref/libref.so from the reference's OWN source lines.
#
#   oracle/ref_build.sh [/root/reference]
#
# The hot-path functions of the reference live in the translation units of its demo executables, next to main() and the OpenCV
# drawing code, and include <Eigen/Eigen> (plus <cppad/...> for the MPC) — none of which exists in this image, and the reference's
# own build system (cmake + find_package(Eigen3/OpenCV)) cannot run here.  So this recipe cuts the cited line ranges out of the
# reference sources WHERE THEY LIE (sed, into oracle/_ref/gen/, git-ignored and deleted again when the libraries are linked: no
# reference source enters this repository or stays on disk),
# and compiles them, unmodified, inside the thin wrappers of oracle/ref_shim/*.cpp, which export them as C symbols (ref_*).
# Headers that need nothing but Eigen (cubic_spline.h, motion_model.h, quintic/quartic_polynomial.h, frenet_path.h,
# cpprobotics_types.h) are included directly from $REF/include.
#   * <Eigen/Eigen>: the host's Eigen if it has one (then this is the literal reference arithmetic), otherwise the stand-in
#     oracle/ref_shim/Eigen/Eigen, which restates Eigen 3.3.9's evaluation order (see its header).  `ref_eigen_kind()` in the
#     library says which one was used; tests report it.
#   * CppAD / IPOPT: oracle/ref_shim/cppad_standin.h — AD<double> = double, ipopt::solve captures the problem it is handed.
# Flags: -std=gnu++11 as the reference's CMakeLists.txt:4 (no -march: SSE2, no FMA); -O1 -ffp-contract=off changes no result
# PROVIDED sin/cos stay the separate libm calls of the reference's unoptimised build: -fdisable-tree-sincos keeps GCC from
# merging `cos(x) ... sin(x)` into one sincos(), which in glibc rounds differently from sin()/cos() (double: ~1 call in 1000).
set -euo pipefail
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
GEN=$OUT/gen
[ -d "$REF/src" ] || { echo "ref_build: $REF/src not found — nothing to do"; exit 0; }
mkdir -p "$GEN"
ext() { sed -n "$2,$3p" "$REF/$1" > "$GEN/$4"; }          # file first last out

# ---- src/extended_kalman_filter.cpp
ext src/extended_kalman_filter.cpp 16 18 ekf_defs.inc            # SIM_TIME, DT, PI
ext src/extended_kalman_filter.cpp 21 78 ekf_fns.inc             # motion_model … ekf_estimation
ext src/extended_kalman_filter.cpp 112 161 ekf_main_setup.inc    # u, xDR, xTrue, xEst, PEst, Q, R, Qsim, Rsim
ext src/extended_kalman_filter.cpp 172 188 ekf_main_body.inc     # one pass of the while loop up to hz.push_back(z)
# ---- src/lqr_speed_steer_control.cpp (5-state)
ext src/lqr_speed_steer_control.cpp 20 30 lqr5_defs.inc          # DT, L, KP, MAX_STEER, matrix aliases
ext src/lqr_speed_steer_control.cpp 65 164 lqr5_fns.inc          # calc_nearest_index, solve_DARE, dlqr, lqr_steering_control, update
ext src/lqr_speed_steer_control.cpp 167 171 lqr5_loop_setup.inc  # T, goal_dis, stop_speed, State state(...)
ext src/lqr_speed_steer_control.cpp 185 186 lqr5_loop_e.inc      # e, e_th
ext src/lqr_speed_steer_control.cpp 195 205 lqr5_loop_body.inc   # control, update, goal test
ext src/lqr_speed_steer_control.cpp 252 265 lqr5_main_course.inc # Spline2D csp_obj(wx, wy) … the sampling loop (ds = 0.1)
ext src/lqr_speed_steer_control.cpp 40 63 lqr5_speed_profile.inc # calc_speed_profile (with the end-of-course slow-down)
# ---- src/lqr_steer_control.cpp (4-state)
ext src/lqr_steer_control.cpp 20 23 lqr4_defs.inc
ext src/lqr_steer_control.cpp 55 146 lqr4_fns.inc               # calc_nearest_index, solve_DARE, dlqr, lqr_steering_control, update
ext src/lqr_steer_control.cpp 149 153 lqr4_loop_setup.inc
ext src/lqr_steer_control.cpp 167 169 lqr4_loop_e.inc            # e, e_th, ind
ext src/lqr_steer_control.cpp 187 198 lqr4_loop_body.inc
ext src/lqr_steer_control.cpp 35 52 lqr4_speed_profile.inc      # calc_speed_profile
# ---- src/model_predictive_control.cpp
ext src/model_predictive_control.cpp 26 48 mpc_defs.inc          # DT … WB (without NX/T, which the wrapper sets)
ext src/model_predictive_control.cpp 50 60 mpc_globals.inc       # using …, M_XREF, x_start … a_start
ext src/model_predictive_control.cpp 69 81 mpc_update.inc
ext src/model_predictive_control.cpp 107 186 mpc_ref_traj.inc    # calc_nearest_index (window), calc_ref_trajectory, smooth_yaw
ext src/model_predictive_control.cpp 188 346 mpc_nlp.inc         # FG_EVAL, mpc_solve
ext src/model_predictive_control.cpp 349 360 mpc_sim_setup.inc   # State state(...), yaw wrap, goal_dis, target_ind, smooth_yaw
ext src/model_predictive_control.cpp 372 385 mpc_sim_body.inc    # calc_ref_trajectory, mpc_solve, update, goal test
ext src/model_predictive_control.cpp 473 486 mpc_main_course.inc # Spline2D csp_obj(wx, wy) … the sampling loop (ds = 1.0)
ext src/model_predictive_control.cpp 83 105 mpc_speed_profile.inc # calc_speed_profile
# ---- src/dynamic_window_approach.cpp (no Eigen in it)
ext src/dynamic_window_approach.cpp 16 41 dwa_types.inc          # PI, the array aliases, class Config
ext src/dynamic_window_approach.cpp 43 155 dwa_fns.inc           # motion … dwa_control
# ---- src/frenet_optimal_trajectory.cpp (+ the headers it includes, taken from $REF/include as they are)
ext src/frenet_optimal_trajectory.cpp 20 38 frenet_defs.inc      # SIM_LOOP … KLON
ext src/frenet_optimal_trajectory.cpp 40 176 frenet_fns.inc      # using namespace, sum_of_power … frenet_optimal_planning
# ---- src/particle_filter.cpp
ext src/particle_filter.cpp 17 22 pf_defs.inc                    # SIM_TIME, DT, PI, MAX_RANGE, NP, NTh
ext src/particle_filter.cpp 25 148 pf_fns.inc                    # motion_model … resampling
ext src/particle_filter.cpp 179 235 pf_main_setup.inc            # time, u, ud, z, RFID, xDR, xTrue, xEst, PEst, Q, R, Qsim, Rsim, px, pw
ext src/particle_filter.cpp 249 271 pf_main_body.inc             # one pass of the while loop up to resampling(...)

if echo '#include <Eigen/Eigen>' | g++ -x c++ -fsyntax-only - 2>/dev/null; then EIGEN_INC=""; KIND=1
elif [ -f /usr/include/eigen3/Eigen/Eigen ]; then EIGEN_INC="-I/usr/include/eigen3"; KIND=1
else EIGEN_INC="-I$HERE/ref_shim"; KIND=0; fi
build_flavour() {      # $1 = library name, $2 = extra flags
  local CXXFLAGS="-std=gnu++11 -O1 -ffp-contract=off -fdisable-tree-sincos -fPIC -w -I$GEN -I$REF/include $EIGEN_INC -DREF_EIGEN_KIND=$KIND $2"
  local OBJS=() tag="$1"
  for f in "$HERE"/ref_shim/ref_*.cpp; do
    o="$OUT/$(basename "${f%.cpp}").$tag.o"
    g++ $CXXFLAGS -c "$f" -o "$o"
    OBJS+=("$o")
  done
  # the MPC unit once more for the BASELINE horizon (the reference's macro T is 6)
  g++ $CXXFLAGS -DREF_MPC_T=21 -c "$HERE/ref_shim/ref_mpc.cpp" -o "$OUT/ref_mpc_T21.$tag.o"
  # the reference headers define non-inline functions (cubic_spline.h, motion_model.h): every unit carries an identical copy
  g++ -shared -Wl,--allow-multiple-definition -o "$OUT/$tag.so" "${OBJS[@]}" "$OUT/ref_mpc_T21.$tag.o" -lm
}
build_flavour libref ""
if [ $KIND = 0 ]; then
  # two more builds of the stand-in flavour in which EVERY coefficient-path sum is forced to one order (eigen_order.h): the tests
  # demand the same bits from all three on the reference's own call sites
  build_flavour libref_cpath_asc "-DORACLE_CPATH_ORDER=1" &
  build_flavour libref_cpath_tree "-DORACLE_CPATH_ORDER=2" &
  wait
fi
rm -f "$OUT"/*.o
rm -rf "$GEN"          # the cut-out line ranges are build intermediates: only the libraries stay (no reference text is kept anywhere)
echo "ref_build: $OUT/libref.so (Eigen: $([ $KIND = 1 ] && echo host || echo stand-in))"

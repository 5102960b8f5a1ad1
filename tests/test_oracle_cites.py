"""Mechanical checks of the citations the oracle and the ABI header rest on.

Round 1 shipped a misreading of the reference (a "doubled fp.d push" that does not exist) through oracle, kernel, golden files
and tests alike.  These tests read /root/reference itself: (1) every "quirk" the oracle claims to reproduce must be visible as a
token pattern in the cited lines; (2) every `file:line` citation in oracle/, include/ and the kernels' headers must point inside
an existing reference file.  Skipped where /root/reference is absent (the GPU box)."""
import glob
import os
import re

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="/root/reference not present")


def lines(rel, a, b=None):
    with open(os.path.join(REF, rel)) as f:
        ls = f.read().split("\n")
    return "\n".join(ls[a - 1:(b or a)])


def squash(s):
    return re.sub(r"\s+", "", s)


def test_frenet_lateral_samples_are_pushed_once_per_step():
    body = lines("src/frenet_optimal_trajectory.cpp", 58, 64)
    for member in ("t", "d", "d_d", "d_dd", "d_ddd"):
        assert len(re.findall(r"\bfp\.%s\.push_back\(" % member, body)) == 1, member
    assert "lat_qp.calc_point(t)" in lines("src/frenet_optimal_trajectory.cpp", 60)
    assert "lat_qp.calc_first_derivative(t)" in lines("src/frenet_optimal_trajectory.cpp", 61)
    # main hands sample [1] of the winner over (:227-231)
    hand = squash(lines("src/frenet_optimal_trajectory.cpp", 227, 231))
    for m in ("s[1]", "d[1]", "d_d[1]", "d_dd[1]", "s_d[1]"):
        assert "final_path." + m in hand, m


def test_frenet_winner_rule_and_float_min_maxima():
    assert squash("if (min_cost >= path.cf)") in squash(lines("src/frenet_optimal_trajectory.cpp", 170))
    assert "numeric_limits<float>::max()" in lines("src/frenet_optimal_trajectory.cpp", 167)
    for ln in (72, 73, 128):
        assert "std::numeric_limits<float>::min()" in lines("src/frenet_optimal_trajectory.cpp", ln)
    assert squash("dist <= ROBOT_RADIUS * ROBOT_RADIUS") in squash(lines("src/frenet_optimal_trajectory.cpp", 138, 148))


def test_quintic_first_derivative_lacks_the_factor_five():
    ln = squash(lines("include/quintic_polynomial.h", 60))
    assert ln.endswith("+a5*std::pow(t,4);") and "5*a5" not in ln
    assert "20*a5*std::pow(t,3)" in squash(lines("include/quintic_polynomial.h", 62, 66))      # the second derivative has its factor
    assert "colPivHouseholderQr().solve(B)" in lines("include/quintic_polynomial.h", 49)
    assert "colPivHouseholderQr().solve(B)" in lines("include/quartic_polynomial.h", 45)
    assert "colPivHouseholderQr().solve(B)" in lines("include/cubic_spline.h", 56)


def test_ekf_model_as_written():
    f = "src/extended_kalman_filter.cpp"
    assert squash(lines(f, 24, 27)) == squash("F_<<1.0, 0, 0, 0,  0, 1.0, 0, 0,  0, 0, 1.0, 0,  0, 0, 0, 1.0;")          # F_(3,3) = 1.0
    assert squash(lines(f, 30, 33)) == squash("B_<< DT * std::cos(x(2,0)), 0, DT * std::sin(x(2,0)), 0, 0.0, DT, 1.0, 0.0;")
    assert squash("jF_(0,2) = -DT * v * std::sin(yaw);") in squash(lines(f, 42))
    assert "float yaw = x(2);" in lines(f, 40) and "float v = u(0);" in lines(f, 41)
    assert squash("jF * PEst * jF.transpose() + Q") in squash(lines(f, 69))
    assert squash("jH * PPred * jH.transpose() + R") in squash(lines(f, 74))
    assert squash("PPred * jH.transpose() * S.inverse()") in squash(lines(f, 75))
    assert squash("(Eigen::Matrix4f::Identity() - K * jH) * PPred") in squash(lines(f, 77))
    assert lines(f, 17).strip() == "#define DT 0.1"
    # the input side: two draws for ud, xTrue with the clean u, xDR with ud, two draws for z
    body = squash(lines(f, 174, 183))
    assert body.index("ud(0)=u(0)+gaussian_d(gen)*Qsim(0,0)") < body.index("xTrue=motion_model(xTrue,u)") < \
        body.index("xDR=motion_model(xDR,ud)") < body.index("z(0)=xTrue(0)+gaussian_d(gen)*Rsim(0,0)") < \
        body.index("ekf_estimation(xEst,PEst,z,ud,Q,R)")


def test_lqr_expressions_as_written():
    f5, f4 = "src/lqr_speed_steer_control.cpp", "src/lqr_steer_control.cpp"
    assert squash("A.transpose()*X*A-A.transpose()*X*B*(R+B.transpose()*X*B).inverse() * B.transpose()*X*A+Q") in squash(lines(f5, 91))
    assert squash("(B.transpose()*X*B + R).inverse() * (B.transpose()*X*A)") in squash(lines(f5, 104))                   # R last here
    assert squash("A.transpose()*X*A-A.transpose()*X*B/(R+B.transpose()*X*B) * B.transpose()*X*A+Q") in squash(lines(f4, 81))  # a division
    assert squash("1.0/(B.transpose()*X*B + R) * (B.transpose()*X*A)") in squash(lines(f4, 94))
    for f, a, b in ((f5, 85, 100), (f4, 75, 90)):
        body = squash(lines(f, a, b))
        assert "intmaxiter=150" in body and "floateps=0.01" in body and "error.cwiseAbs().maxCoeff()<eps" in body
        assert body.index("returnXn") < body.index("X=Xn") < body.rindex("returnX;")                                    # cap exit returns X, not Xn
    assert squash("B(3, 0) = state.v/L;") in squash(lines(f5, 125)) and squash("B(4, 1) = DT;") in squash(lines(f5, 126))
    assert squash("x(1) = (e-pe)/DT;") in squash(lines(f5, 136)) and squash("Eigen::Vector2f ustar = -K * x;") in squash(lines(f5, 141))
    assert squash("std::atan2((L*k), (double)1.0)") in squash(lines(f5, 143))
    assert "if (d_e<mind){" in lines(f5, 72)                                                                            # strict: the first minimum wins
    assert lines("include/motion_model.h", 18).strip() == "#define YAW_P2P(angle) std::fmod(std::fmod((angle)+M_PI, 2*M_PI)-2*M_PI, 2*M_PI)+M_PI"
    # the 4-state loop: proportional speed control on speed_profile[ind], ind bumped when nearly stopped
    assert squash("float ai = KP * (speed_profile[ind]-state.v);") in squash(lines(f4, 188))
    assert squash("if (std::abs(state.v) <= stop_speed) ind += 1;") in squash(lines(f4, 191))


def test_mpc_problem_as_written():
    f = "src/model_predictive_control.cpp"
    assert lines(f, 24).strip() == "#define T 6" and lines(f, 26).strip() == "#define DT 0.2" and lines(f, 36).strip() == "#define WB 2.5"
    assert squash(lines(f, 298, 301)) == squash("for (auto i = v_start; i < v_start+T; i++) { vars_lowerbound[i] = MIN_SPEED; vars_upperbound[i] = MAX_SPEED; }")
    assert squash("0.01 * CppAD::pow(vars[a_start+i], 2)") in squash(lines(f, 203)) and squash("0.01 * CppAD::pow(vars[delta_start+i], 2)") in squash(lines(f, 204))
    assert squash("0.01 * CppAD::pow(vars[a_start+i+1] - vars[a_start+i], 2)") in squash(lines(f, 208))
    assert squash("1 * CppAD::pow(vars[delta_start+i+1] - vars[delta_start+i], 2)") in squash(lines(f, 209))
    assert squash("yaw1 - (yaw0 + v0 * CppAD::tan(delta0) / WB * DT)") in squash(lines(f, 244))
    for ln, w in ((247, ""), (248, ""), (249, "0.5*"), (250, "0.5*")):
        assert squash(lines(f, ln)).startswith("fg[0]+=" + w + "CppAD::pow(traj_ref(")
    assert "max_iter      50" in lines(f, 326) and "max_cpu_time          0.05" in lines(f, 328)
    assert squash("for(unsigned int i=pind; i<pind+N_IND_SEARCH; i++)") in squash(lines(f, 110))                       # unchecked window
    assert squash("if (state.v > MAX_SPEED) state.v = MAX_SPEED;") in squash(lines(f, 78))
    assert squash("update(state, output[a_start], output[delta_start]);") in squash(lines(f, 376))
    assert squash("calc_nearest_index(state, cx, cy, cyaw, target_ind);") == squash(lines(f, 358))                       # result discarded


def test_particle_filter_and_dwa_quirks():
    f = "src/particle_filter.cpp"
    assert lines(f, 22).strip() == "#define NTh NP/2"
    assert "std::mt19937 gen," in lines(f, 123) and "&" not in lines(f, 123)                                             # generator by value
    assert squash("resampleid(j) = base(j) + uni_d(gen)/NP;") in squash(lines(f, 133))
    assert squash("while(resampleid(i) > wcum(ind) && ind<NP-1)") in squash(lines(f, 139))
    g = "src/dynamic_window_approach.cpp"
    assert squash("float min_cost = 10000.0;") in squash(lines(g, 120)) and squash("if (min_cost >= final_cost){") in squash(lines(g, 136))
    assert squash("return u, traj;") in squash(lines(g, 154))


# where every function of the hot path starts and ends in the reference (first line = its signature, last line = its closing brace)
FUNCTIONS = {
    "src/extended_kalman_filter.cpp": dict(motion_model=(22, 36), jacobF=(38, 47), observation_model=(50, 55), jacobH=(57, 62), ekf_estimation=(64, 78)),
    "src/lqr_speed_steer_control.cpp": dict(calc_nearest_index=(65, 83), solve_DARE=(85, 100), dlqr=(102, 106), lqr_steering_control=(108, 151),
                                            update=(154, 164), closed_loop_prediction=(166, 246)),
    "src/lqr_steer_control.cpp": dict(calc_nearest_index=(55, 73), solve_DARE=(75, 90), dlqr=(92, 96), lqr_steering_control=(98, 133), update=(136, 146),
                                      closed_loop_prediction=(148, 214)),
    "src/model_predictive_control.cpp": dict(update=(69, 81), calc_nearest_index=(107, 127), calc_ref_trajectory=(130, 170), smooth_yaw=(172, 185),
                                             mpc_solve=(255, 346), mpc_simulation=(348, 465)),
    "src/particle_filter.cpp": dict(motion_model=(26, 40), gauss_likelihood=(53, 57), calc_covariance=(59, 71), pf_localization=(73, 109), cumsum=(111, 118),
                                    resampling=(120, 148)),
    "src/dynamic_window_approach.cpp": dict(motion=(43, 50), calc_dynamic_window=(52, 60), calc_trajectory=(63, 74), calc_obstacle_cost=(77, 101),
                                            calc_to_goal_cost=(103, 113), calc_final_input=(115, 145), dwa_control=(148, 155)),
    "src/frenet_optimal_trajectory.cpp": dict(sum_of_power=(43, 49), calc_frenet_paths=(51, 100), calc_global_paths=(102, 136), check_collision=(138, 148),
                                              check_paths=(150, 158), frenet_optimal_planning=(160, 176)),
}


def test_function_locations():
    """The ranges the docs and the oracle cite for whole functions: the signature is on the first line, the brace closes on the last."""
    for rel, fns in FUNCTIONS.items():
        for name, (a, b) in fns.items():
            assert re.search(r"\b%s\s*\(" % name, lines(rel, a)), (rel, name, a)
            body = lines(rel, a, b)
            assert body.count("{") == body.count("}") and body.rstrip().endswith(("}", "};")), (rel, name, a, b)
            assert lines(rel, b + 1).strip() == "" or not lines(rel, b + 1).startswith(" "), (rel, name, b)


def test_docs_use_those_locations():
    """`name :a-b` / `name` `:a-b` mentions of a hot-path function in our sources must lie inside that function's real range (in one of the files that define it)."""
    known = {}
    for rel, fns in FUNCTIONS.items():
        for name, ab in fns.items():
            known.setdefault(name, set()).add(ab)
    pat = re.compile(r"`?\b(%s)`?\s*\(?`?:(\d+)-(\d+)" % "|".join(sorted(known)))
    bad = []
    for path in _our_files():
        with open(path, errors="replace") as f:
            for m in pat.finditer(f.read()):
                a, b = int(m.group(2)), int(m.group(3))
                if not any(A <= a <= b <= B for A, B in known[m.group(1)]):       # the whole function, or a part of it
                    bad.append((os.path.relpath(path, ROOT), m.group(0)))
    assert not bad, bad


def _our_files():
    files = glob.glob(os.path.join(ROOT, "oracle", "*.cpp")) + glob.glob(os.path.join(ROOT, "oracle", "*.h")) + \
        glob.glob(os.path.join(ROOT, "oracle", "*.py")) + \
        glob.glob(os.path.join(ROOT, "oracle", "ref_shim", "*.cpp")) + glob.glob(os.path.join(ROOT, "include", "*")) + \
        glob.glob(os.path.join(ROOT, "cpprobotics_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "cpprobotics_amd", "*.py")) + \
        glob.glob(os.path.join(ROOT, "examples", "*.cpp")) + [os.path.join(ROOT, "DESIGN.md"), os.path.join(ROOT, "HISTORY.md"), os.path.join(ROOT, "INTEGRATION.md")]
    return [p for p in files if os.path.isfile(p)]


CITE = re.compile(r"((?:src|include)/[A-Za-z0-9_]+\.(?:cpp|h)):(\d+)(?:-(\d+))?")


def test_every_citation_points_into_the_reference():
    n = 0
    for path in _our_files():
        with open(path, errors="replace") as f:
            text = f.read()
        for m in CITE.finditer(text):
            rel, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            ref_file = os.path.join(REF, rel)
            assert os.path.isfile(ref_file), f"{path}: cites {rel}, which does not exist"
            with open(ref_file) as g:
                nlines = g.read().count("\n") + 1
            assert 1 <= a <= b <= nlines, f"{path}: {m.group(0)} is outside {rel} ({nlines} lines)"
            n += 1
    assert n > 100
    assert "twice" not in open(os.path.join(ROOT, "include", "crx.h")).read()

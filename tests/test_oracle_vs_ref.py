"""The oracle against the reference's own source lines.

oracle/_ref/libref.so (oracle/ref_build.sh, built by __graft_entry__.build() whenever /root/reference is present) is the
reference's hot-path code cut out of its files and compiled unmodified — against the host's Eigen if there is one, else against the
Eigen stand-in of oracle/ref_shim (which restates Eigen's evaluation order; see its header).  Bit equality is demanded wherever
the oracle claims it.  Skipped when the library has not been built (a host with neither /root/reference nor the shipped .so);
tests/test_ref_golden.py then still checks the oracle against the committed outputs of this library.
"""
import math

import numpy as np
import pytest

from common import ekf_QR, ekf_agents, ekf_noise, lqr_course, lqr_speeds, mpc_course_f32, mpc_problem, tracking_agents

from oracle import ref_lib as R

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref.so not built (needs /root/reference)")


def _eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


@pytest.fixture()
def host_trig(oracle_mod, monkeypatch):
    """The reference calls the host libm; make the oracle do the same whatever flavour this host's libm is."""
    monkeypatch.setattr(oracle_mod.oracle_lib, "trig_mode", lambda: 0)
    return oracle_mod


def test_which_eigen():
    print("libref.so built against:", R.eigen_kind())


# ---- EKF ---------------------------------------------------------------------------------------------------------------------
def test_ekf_small_functions(host_trig):
    o = host_trig
    rng = np.random.default_rng(0)
    n = 4000
    x = np.stack([rng.normal(0, 50, n), rng.normal(0, 50, n), rng.uniform(-200, 200, n), rng.normal(0, 5, n)], axis=1).astype(np.float32)
    x[:8, 2] = [0.0, -0.0, 1e-30, 3.1415927, -3.1415927, 1e6, 119.9, 120.1]
    u = np.stack([rng.normal(1, 2, n), rng.normal(0, 1, n)], axis=1).astype(np.float32)
    assert _eq(o.motion_model(x, u, trig=0), R.motion_model(x, u))
    assert _eq(o.jacobF(x, u, trig=0), R.jacobF(x, u))
    assert _eq(o.observation_model(x), R.observation_model(x))
    assert _eq(o.jacobH(), R.jacobH())


@pytest.mark.parametrize("single", [True, False])
def test_ekf_estimation_bit_exact(host_trig, single):
    o = host_trig
    Q, Rm = ekf_QR()
    n, T = (1, 1000) if single else (96, 300)
    u, x0, P0 = ekf_agents(n, 3, single_vehicle=single)
    z, ud, *_ = o.ekf_simulate_inputs(u, x0, x0, ekf_noise(T, n, 4), trig=0)
    for order in (0, 1):                                       # Eigen's order and all-ascending: identical on this path
        xo, Po, xho, pho = o.ekf_run(x0, P0, z, ud, Q, Rm, trig=0, sum_order=order, want_phist=True)
        xr, Pr, xhr, phr = R.ekf_run(x0, P0, z, ud, Q, Rm, want_phist=True)
        assert _eq(xho, xhr) and _eq(pho, phr) and _eq(xo, xr) and _eq(Po, Pr)


def test_ekf_wide_range_inputs(host_trig):
    """Magnitudes over 40 decades, random dense (non-symmetric) P, Q, R: every finite run equal bit for bit."""
    o = host_trig
    rng = np.random.default_rng(5)
    n, T = 400, 40
    mag = lambda lo, hi, shape: (10.0 ** rng.uniform(lo, hi, shape) * rng.choice([-1, 1], shape)).astype(np.float32)
    x0 = mag(-6, 6, (n, 4)); P0 = mag(-8, 4, (n, 16)); z = mag(-6, 6, (T, n, 2)); u = mag(-6, 3, (T, n, 2))
    Q = mag(-6, 0, 16); Rm = np.array([1.0, 0.1, 0.2, 2.0], np.float32)
    xo, Po, xho, pho = o.ekf_run(x0, P0, z, u, Q, Rm, trig=0, want_phist=True)
    xr, Pr, xhr, phr = R.ekf_run(x0, P0, z, u, Q, Rm, want_phist=True)
    fin = np.isfinite(phr).all(axis=(0, 2)) & np.isfinite(xhr).all(axis=(0, 2))
    assert fin.mean() > 0.5
    assert _eq(xho[:, fin], xhr[:, fin]) and _eq(pho[:, fin], phr[:, fin])


def test_ekf_main_loop_and_constants(host_trig):
    """main() :110-188: the constants it builds and the input side (ud, xTrue, xDR, z) of every pass, then the estimate."""
    o = host_trig
    T = 501                                                    # SIM_TIME / DT passes, as the reference runs
    w = ekf_noise(T, 1, 9)[:, 0, :]                            # float draws, promoted exactly to the double the reference multiplies
    m = R.ekf_main(w.astype(np.float64))
    Q, Rm = ekf_QR()
    assert _eq(m["Q"], Q) and _eq(m["R"], Rm)
    qsim = (1.0, (30.0 / 180 * math.pi) * (30.0 / 180 * math.pi))
    assert _eq(m["Qsim"], np.array([qsim[0], 0, 0, qsim[1]], np.float32)) and _eq(m["Rsim"], np.array([0.25, 0, 0, 0.25], np.float32))
    u = np.array([[1.0, 0.1]], np.float32); x0 = np.zeros((1, 4), np.float32)
    z, ud, _, _, xth, xdh = o.ekf_simulate_inputs(u, x0, x0, w[:, None, :], trig=0, want_hist=True)
    assert _eq(ud[:, 0], m["hud"]) and _eq(z[:, 0], m["hz"]) and _eq(xth[:, 0], m["hxTrue"]) and _eq(xdh[:, 0], m["hxDR"])
    P0 = np.eye(4, dtype=np.float32).reshape(1, 16)
    xo, Po, xho, _ = o.ekf_run(x0, P0, z, ud, Q, Rm, trig=0)
    assert _eq(xho[:, 0], m["hxEst"]) and _eq(Po[0], m["PEst"])


# ---- DARE / dlqr ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim", [5, 4])
def test_dare_on_the_references_own_matrices(oracle_mod, dim):
    v = lqr_speeds(600, 11)
    v[:4] = [0.0, 1e-3, -1e-3, 2.7777777]
    A, B, Q, Rm = oracle_mod.lqr_build(v, dim)
    Xo, Ko, it = oracle_mod.dare(A, B, Q, Rm)
    Xr, Kr = R.dare(A, B, Q, Rm)
    assert _eq(Xo, Xr) and _eq(Ko, Kr)
    assert it.min() >= 2 and it.max() == 150                  # includes the iteration-cap exit (returns X, not Xn)


@pytest.mark.parametrize("dim", [5, 4])
def test_dare_dense_random_matrices(oracle_mod, dim):
    """Dense A, B, Q, R — beyond the reference's call sites; here the accumulation order of every product matters."""
    rng = np.random.default_rng(12 + dim)
    n, m = 300, (2 if dim == 5 else 1)
    A = (np.eye(dim)[None] * 0.9 + rng.normal(0, 0.15, (n, dim, dim))).astype(np.float32).reshape(n, -1)
    B = rng.normal(0, 0.5, (n, dim * m)).astype(np.float32)
    Qh = rng.normal(0, 0.4, (n, dim, dim)); Q = (Qh @ Qh.transpose(0, 2, 1) + 0.1 * np.eye(dim)).astype(np.float32).reshape(n, -1)
    if dim == 5:
        Rh = rng.normal(0, 0.4, (n, 2, 2)); Rm = (Rh @ Rh.transpose(0, 2, 1) + 0.5 * np.eye(2)).astype(np.float32).reshape(n, -1)
    else:
        Rm = rng.uniform(0.3, 2.0, (n, 1)).astype(np.float32)
    Xo, Ko, it = oracle_mod.dare(A, B, Q, Rm)
    Xr, Kr = R.dare(A, B, Q, Rm)
    fin = np.isfinite(Xr).all(axis=1)
    assert fin.mean() > 0.9 and _eq(Xo[fin], Xr[fin]) and _eq(Ko[fin], Kr[fin])


# ---- tracking front-end, vehicle update, closed loops -----------------------------------------------------------------------------
def test_nearest_index_and_steering_control(host_trig):
    o = host_trig
    course, _ = lqr_course()
    st = tracking_agents(500, course, 21)
    ind_o, e_o = o.calc_nearest_index(st, course)
    ind_r, e_r = R.calc_nearest_index(st, course)
    assert _eq(ind_o, ind_r) and _eq(e_o, e_r)
    rng = np.random.default_rng(22)
    pe = rng.normal(0, 0.3, len(st)).astype(np.float32); pth = rng.normal(0, 0.2, len(st)).astype(np.float32)
    c_o, _, pe_o, pth_o = o.lqr_steering_control(st, course, pe, pth, dim=5)
    c_r, _, pe_r, pth_r = R.lqr_steering_control(st, course, pe, pth, dim=5)
    assert _eq(c_o, c_r) and _eq(pe_o, pe_r) and _eq(pth_o, pth_r)
    d_o, i_o, pe_o, pth_o = o.lqr_steering_control(st, course, pe, pth, dim=4)
    d_r, i_r, pe_r, pth_r = R.lqr_steering_control(st, course, pe, pth, dim=4)
    assert _eq(d_o, d_r) and _eq(i_o, i_r) and _eq(pe_o, pe_r) and _eq(pth_o, pth_r)


def test_update_both_variants(host_trig):
    o = host_trig
    rng = np.random.default_rng(23)
    n = 3000
    st = np.stack([rng.normal(0, 30, n), rng.normal(0, 30, n), rng.uniform(-10, 10, n), rng.uniform(-8, 18, n)], axis=1).astype(np.float32)
    a = rng.uniform(-2, 2, n).astype(np.float32); d = rng.uniform(-1.5, 1.5, n).astype(np.float32)
    d[:3] = [np.float32(math.pi / 4), -np.float32(math.pi / 4), 0.0]
    assert _eq(o.update(st, a, d), R.lqr_update(st, a, d))                                             # LQR files: DT 0.1, L 0.5
    assert _eq(o.update(st, a, d, dt=0.2, wheelbase=2.5, clamp_speed=True), R.mpc_update(st, a, d))   # MPC file: + speed clamp


@pytest.mark.parametrize("dim", [5, 4])
def test_lqr_closed_loop_every_tick(host_trig, dim):
    o = host_trig
    course, goal = lqr_course()
    st = np.zeros((10, 4), np.float32)                                      # agent 0: the reference's own start (:171 / :153)
    st[1:] = tracking_agents(9, tuple(c[:80] for c in course), 31, spread=0.3)
    so, to, ho, *_ = o.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=700, want_hist=True)
    sr, tr, hr = R.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=700)
    assert _eq(to, tr) and (tr < 700).all()
    for a in range(len(st)):
        assert _eq(ho[: to[a], a], hr[: tr[a], a])
    assert _eq(so, sr)


# ---- MPC: layout, NLP functions, bounds, the callers ---------------------------------------------------------------------------------
@pytest.mark.parametrize("T", [6, 21])
def test_mpc_problem_definition(oracle_mod, T):
    """FG_EVAL (:199-252) and the problem mpc_solve hands to IPOPT (:263-328) against the oracle's statement of the NLP."""
    lay = R.mpc_layout(T)
    assert lay == dict(x=0, y=T, yaw=2 * T, v=3 * T, delta=4 * T, a=4 * T + T - 1)
    x0, xref = mpc_problem(40, T, seed=41)
    rng = np.random.default_rng(42)
    P = oracle_mod.MPC_DEFAULTS
    for a in range(len(x0)):
        U = np.stack([rng.uniform(-0.9, 0.9, T - 1), rng.uniform(-1.2, 1.2, T - 1)], axis=1)      # (delta, a) per stage
        J, S = oracle_mod.mpc_cost(x0[a], xref[a], T, U)                                         # rollout: S[i] = (x, y, yaw, v, ...) at knot i
        vars_ = np.concatenate([S[:, 0], S[:, 1], S[:, 2], S[:, 3], U[:, 0], U[:, 1]])
        fg = R.mpc_fg_eval(xref[a], vars_, T)
        assert abs(fg[0] - J) <= 1e-12 * max(1.0, abs(J))                                       # same objective
        # on a rollout the dynamics constraints vanish and the four initial-state functions return the state itself
        assert np.allclose(fg[1 + np.array([0, T, 2 * T, 3 * T])], x0[a].astype(np.float64), rtol=0, atol=0)
        dyn = np.delete(fg[1:], [0, T, 2 * T, 3 * T])
        assert np.abs(dyn).max() < 1e-12
        # off the rollout the constraint functions are the defects: perturb one knot
        v2 = vars_.copy(); v2[lay["yaw"] + 2] += 0.125
        fg2 = R.mpc_fg_eval(xref[a], v2, T)
        assert abs(fg2[1 + lay["yaw"] + 2] - (fg[1 + lay["yaw"] + 2] + 0.125)) < 1e-12
    cap = R.mpc_solve(x0[0], xref[0], T)
    nv = 4 * T + 2 * (T - 1)
    assert cap["xi"][lay["x"]] == x0[0, 0] and cap["xi"][lay["v"]] == x0[0, 3] and np.count_nonzero(cap["xi"]) <= 4   # zero start :266-274
    lo, hi = cap["xl"], cap["xu"]
    sl = lambda k, m: slice(lay[k], lay[k] + m)
    assert (lo[sl("delta", T - 1)] == -P["max_steer"]).all() and (hi[sl("delta", T - 1)] == P["max_steer"]).all()
    assert (lo[sl("a", T - 1)] == -P["max_accel"]).all() and (hi[sl("a", T - 1)] == P["max_accel"]).all()
    assert (lo[sl("v", T)] == P["min_speed"]).all() and (hi[sl("v", T)] == P["max_speed"]).all()       # every knot, :298-301
    assert (lo[: 3 * T] == -1e7).all() and (hi[: 3 * T] == 1e7).all()
    g = np.zeros(4 * T); g[[0, T, 2 * T, 3 * T]] = x0[0].astype(np.float64)
    assert _eq(cap["gl"], g) and _eq(cap["gu"], g)
    assert "max_iter      50" in cap["options"] and "max_cpu_time          0.05" in cap["options"]
    # what mpc_solve returns is the solver's answer rounded to float, in the same layout (:341-345)
    sol = rng.normal(0, 1, nv)
    assert _eq(R.mpc_solve(x0[0], xref[0], T, solver=lambda x, r: sol)["result"], sol.astype(np.float32))


@pytest.mark.parametrize("T", [6, 21])
def test_calc_ref_trajectory_and_window_search(host_trig, T):
    o = host_trig
    course, _ = mpc_course_f32()
    nc = len(course[0])
    rng = np.random.default_rng(51)
    st = tracking_agents(400, course, 52, spread=1.0)
    st[:, 3] = rng.uniform(-3, 15, len(st)).astype(np.float32)
    pind = rng.integers(0, nc - 10, len(st)).astype(np.int32)              # the reference reads cx[pind .. pind+9] unchecked
    assert _eq(o.calc_nearest_index_window(st, course, pind), R.calc_nearest_index_window(st, course, pind, T=T))
    xo, to = o.calc_ref_trajectory(st, course, pind, T)
    xr, tr = R.calc_ref_trajectory(st, course, pind, T)
    assert _eq(xo, xr) and _eq(to, tr)


def test_mpc_simulation_loop_with_the_oracle_solver(host_trig):
    """mpc_simulation (:348-385) as written, IPOPT replaced by the oracle's solver: the oracle's own closed loop must walk
    the same trajectory, tick for tick (same solver on both sides, so this pins the loop, update and the indices it reads)."""
    o = host_trig
    T = 6
    course, goal = mpc_course_f32()
    cx, cy, cyaw, ck, sp = course
    def solver(x0, xref):
        sol, st, _ = o.mpc_solve(x0.astype(np.float32)[None], xref[None], T)
        return sol[0].astype(np.float64)
    ticks, traj, ctl, cs = R.mpc_simulation(course, goal, T, 60, solver)
    assert _eq(cs, cyaw)                                                     # smooth_yaw leaves an already smooth course alone
    st0 = np.array([[cx[0], cy[0], cyaw[0], sp[0]]], np.float32)
    so, to, ho, _ = o.mpc_closed_loop(st0, course, goal, T, 60, want_hist=True)
    assert ticks == to[0] == 60 and _eq(ho[:60, 0], traj)


# ---- the two sampling planners --------------------------------------------------------------------------------------
def test_dwa_episodes(host_trig):
    o = host_trig
    O = o.oracle_lib
    rng = np.random.default_rng(61)
    n = 10
    st = np.stack([rng.uniform(-1, 9, n), rng.uniform(-1, 9, n), rng.uniform(-3.2, 3.2, n), rng.uniform(-0.5, 1.0, n),
                   rng.uniform(-0.69, 0.69, n)], axis=1).astype(np.float32)
    st[0] = (0.0, 0.0, 3.141592653 / 8.0, 0.0, 0.0)                                   # the reference's start (:167)
    u = st[:, 3:5].copy()
    goal = np.stack([rng.uniform(8, 12, n), rng.uniform(8, 12, n)], axis=1).astype(np.float32)
    goal[0] = (10.0, 10.0)
    assert _eq(R.dwa_config(), O.DWA_CONFIG)                                          # class Config :25-41
    so, uo, to, ho = o.dwa_run(st, u, goal, 80, want_hist=True)
    sr, ur, tr, hr = R.dwa_run(st, u, goal, O.DWA_OBSTACLES, 80)
    assert _eq(to, tr) and _eq(so, sr) and _eq(uo, ur)
    for a in range(n):
        assert _eq(ho[: to[a], a], hr[: tr[a], a])


def test_polynomial_coefficients_and_spline_table(oracle_mod):
    """colPivHouseholderQr().solve() call sites: quintic 3x3 (:49), quartic 2x2 (:45), Spline nx x nx (cubic_spline.h:56)."""
    O = oracle_mod.oracle_lib
    assert _eq(oracle_mod.frenet_spline_build(), R.frenet_spline_build(O.FRENET_WX, O.FRENET_WY))
    rng = np.random.default_rng(62)
    for nx in (2, 3, 7, 20):
        wx = np.cumsum(rng.uniform(2.0, 12.0, nx)).astype(np.float32); wy = rng.uniform(-8, 8, nx).astype(np.float32)
        assert _eq(oracle_mod.frenet_spline_build(wx, wy), R.frenet_spline_build(wx, wy))


def test_frenet_candidates_and_episodes(oracle_mod):
    o = oracle_mod
    O = o.oracle_lib
    coef = o.frenet_spline_build()
    rx, ry = o.frenet_course_samples(coef)
    goal = [rx[-1], ry[-1]]
    rng = np.random.default_rng(63)
    n = 10
    st = np.stack([rng.uniform(0.0, 40.0, n), rng.uniform(1.0, 9.0, n), rng.uniform(-3.0, 3.0, n), rng.uniform(-0.8, 0.8, n),
                   rng.uniform(-0.5, 0.5, n)], axis=1).astype(np.float32)
    st[0] = O.FRENET_STATE0                                                            # the reference's start (:215-219)
    p = o.frenet_plan(st, coef)
    r1 = R.frenet_run(st, O.FRENET_WX, O.FRENET_WY, goal, O.FRENET_OBSTACLES, 1, want_paths=True, cap=p["path_cf"].shape[1])
    assert _eq(p["n_paths"], r1["n_paths"]) and _eq(p["path_ok"], r1["path_ok"]) and np.array_equal(p["path_cf"], r1["path_cf"], equal_nan=True)
    ro = o.frenet_run(st, coef, goal, 25, want_hist=True)
    rr = R.frenet_run(st, O.FRENET_WX, O.FRENET_WY, goal, O.FRENET_OBSTACLES, 25)
    assert _eq(ro["ticks"], rr["ticks"]) and _eq(ro["status"] & 1, rr["status"] & 1)
    for a in range(n):
        assert _eq(ro["hist"][: ro["ticks"][a], a], rr["hist"][: rr["ticks"][a], a])
    # the reference's own scenario, start to goal: 98 planning calls, threads the obstacles, same trajectory
    full_o = o.frenet_run(st[:1], coef, goal, 500, want_hist=True)
    full_r = R.frenet_run(st[:1], O.FRENET_WX, O.FRENET_WY, goal, O.FRENET_OBSTACLES, 500)
    assert full_r["ticks"][0] == full_o["ticks"][0] == 98 and full_r["status"][0] == 0
    assert _eq(full_o["hist"][:98, 0], full_r["hist"][:98, 0])


def test_course_generation_of_the_mains():
    """The product's host course builder (crx_course_from_waypoints: Spline2D by float QR, calc_postion / calc_yaw /
    calc_curvature sampled every ds) against the sampling loops of the reference's own mains."""
    import cpprobotics_amd as crx
    lqr_w = ([0.0, 6.0, 12.5, 10.0, 17.5, 20.0, 25.0], [0.0, -3.0, -5.0, 6.5, 3.0, 0.0, 0.0])            # lqr_speed_steer_control.cpp:249-250
    mpc_w = ([0.0, 60.0, 125.0, 50.0, 75.0, 35.0, -10.0], [0.0, 0.0, 50.0, 65.0, 30.0, 50.0, -20.0])     # model_predictive_control.cpp:469-471
    for (wx, wy), ds, which in ((lqr_w, 0.1, "lqr"), (mpc_w, 1.0, "mpc")):
        cx, cy, cyaw, ck, sp = crx.course_from_waypoints(wx, wy, ds, variant=5 if which == "lqr" else 0)
        rx, ry, ryaw, rk = R.main_course(wx, wy, which)
        assert len(cx) == len(rx) > 100
        assert _eq(cx, rx) and _eq(cy, ry) and _eq(cyaw, ryaw) and _eq(ck, rk)
        assert np.isfinite(sp).all() and abs(abs(sp[0]) - 10.0 / 3.6) < 1e-6


def test_calc_speed_profile_of_the_three_tracking_files():
    """crx_calc_speed_profile (host) against the reference's own three functions (compiled behind a vector type that absorbs their two
    out-of-range writes): the mains' courses, and synthetic headings with direction switches (dyaw between 45 and 90 degrees)."""
    import cpprobotics_amd as crx
    lqr_w = ([0.0, 6.0, 12.5, 10.0, 17.5, 20.0, 25.0], [0.0, -3.0, -5.0, 6.5, 3.0, 0.0, 0.0])
    mpc_w = ([0.0, 60.0, 125.0, 50.0, 75.0, 35.0, -10.0], [0.0, 0.0, 50.0, 65.0, 30.0, 50.0, -20.0])
    rng = np.random.default_rng(3)
    cases = [R.main_course(*lqr_w, "lqr")[:3], R.main_course(*mpc_w, "mpc")[:3]]
    for k in range(6):
        n = int(rng.integers(2, 300))
        yaw = np.cumsum(rng.choice([0.02, -0.03, 1.0, -1.2, 0.5], n, p=[0.45, 0.45, 0.04, 0.03, 0.03])).astype(np.float32)
        x = np.cumsum(np.cos(yaw) * rng.choice([1.0, -1.0, 0.0], n, p=[0.8, 0.1, 0.1])).astype(np.float32)
        y = np.cumsum(np.sin(yaw) * rng.choice([1.0, 0.0], n, p=[0.9, 0.1])).astype(np.float32)
        cases.append((x, y, yaw))
    for rx, ry, ryaw in cases:
        for which in (5, 4, 0):
            for ts in (10.0 / 3.6, 1.0):
                assert _eq(crx.calc_speed_profile(which, rx, ry, ryaw, ts), R.calc_speed_profile(which, rx, ry, ryaw, ts)), (which, len(ryaw))
    assert (crx.calc_speed_profile(5, *cases[0])[-39:] < 10.0 / 3.6).all() and crx.calc_speed_profile(4, *cases[0])[-1] == 0.0


def test_smooth_yaw_of_the_mpc_main():
    """crx_smooth_yaw (host) against the reference's own smooth_yaw (:172-185): the MPC main's course headings (atan2 output, which
    jumps by 2 pi where the course turns through +-pi), synthetic windings of several turns, and the refusals."""
    import cpprobotics_amd as crx
    mpc_w = ([0.0, 60.0, 125.0, 50.0, 75.0, 35.0, -10.0], [0.0, 0.0, 50.0, 65.0, 30.0, 50.0, -20.0])
    *_, ryaw, _ = R.main_course(*mpc_w, "mpc")
    assert np.abs(np.diff(ryaw)).max() > 3.0                       # the raw headings do jump
    out = crx.smooth_yaw(ryaw)
    assert _eq(out, R.smooth_yaw(ryaw)) and np.abs(np.diff(out)).max() < np.pi / 2 + 1e-6 and not _eq(out, ryaw)
    rng = np.random.default_rng(5)
    for k in range(20):
        wind = np.cumsum(rng.normal(0.0, 0.8, 300))                # up to several turns either way
        raw = (np.arctan2(np.sin(wind), np.cos(wind)) + (k % 3 - 1) * 2 * np.pi * (rng.random(300) < 0.05)).astype(np.float32)
        assert _eq(crx.smooth_yaw(raw), R.smooth_yaw(raw))
    assert len(crx.smooth_yaw(np.zeros(0, np.float32))) == 0 and _eq(crx.smooth_yaw(np.float32([1.0])), np.float32([1.0]))
    for bad in (np.float32([0.0, np.inf]), np.float32([0.0, np.nan, 0.0]), np.float32([0.0, 1e30])):
        with pytest.raises(crx.CrxError):
            crx.smooth_yaw(bad)


# ---- how much of this depends on the stand-in's restatement of Eigen's accumulation order? -------------------------------------------
@pytest.mark.skipif(not (R.flavour("cpath_asc").available() and R.flavour("cpath_tree").available()), reason="stand-in flavours not built (host Eigen in use)")
def test_which_reference_call_sites_depend_on_the_coefficient_path_order(oracle_mod):
    """The reference's lines compiled against the stand-in three times: with the order rule of oracle/eigen_order.h, and with every
    coefficient-path sum forced to ascending / to the unrolled tree.  ekf_estimation, solve_DARE and dlqr on the reference's own
    matrices give the same bits in all three — every coefficient-path sum there has at most two non-zero terms — so the one
    decision of the stand-in that cannot be checked without Eigen's binary has no influence on them.  The ONE expression of the
    hot path that is sensitive to it is `-K * x` of lqr_steering_control (dense gain times dense error state,
    lqr_speed_steer_control.cpp:141, lqr_steer_control.cpp:127): 2x5 column-major K -> unrolled tree (the 5-state loop equals the
    forced-tree build), 1x4 row-major K -> vectorised redux (differs from both forced orders by one float ulp of the steering
    command).  Dense random DARE inputs, beyond the reference's call sites, are sensitive too."""
    Q, Rm = ekf_QR()
    n, T = 64, 300
    u, x0, P0 = ekf_agents(n, 3)
    z, ud, *_ = oracle_mod.ekf_simulate_inputs(u, x0, x0, ekf_noise(T, n, 4), trig=0)
    v = lqr_speeds(300, 11)
    course, goal = lqr_course()
    st = tracking_agents(6, tuple(c[:80] for c in course), 31, spread=0.3)
    rng = np.random.default_rng(17)
    Ad = (np.eye(5)[None] * 0.9 + rng.normal(0, 0.15, (50, 5, 5))).astype(np.float32).reshape(50, -1)
    Bd = rng.normal(0, 0.5, (50, 10)).astype(np.float32)
    Qd = np.tile(np.eye(5, dtype=np.float32).reshape(1, 25), (50, 1)); Rd = np.tile(np.eye(2, dtype=np.float32).reshape(1, 4), (50, 1))

    def everything():
        out = dict(ekf=R.ekf_run(x0, P0, z, ud, Q, Rm, want_phist=True)[2:4])
        for dim in (5, 4):
            A, B, Qm, Rr = oracle_mod.lqr_build(v, dim)
            out[f"dare{dim}"] = R.dare(A, B, Qm, Rr)
            out[f"loop{dim}"] = R.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=400)
        out["dense"] = R.dare(Ad, Bd, Qd, Rd)
        return out
    base = everything()
    same = lambda a, b: all(_eq(p, q) for p, q in zip(a, b))
    for name in ("cpath_asc", "cpath_tree"):
        with R.flavour(name):
            other = everything()
        for key in ("ekf", "dare5", "dare4"):
            assert same(base[key], other[key]), (name, key)
        assert not same(base["dense"], other["dense"]), "dense inputs should be sensitive to the order"
        if name == "cpath_tree":
            assert same(base["loop5"], other["loop5"])
        # the 4-state loop: same tick counts, states within a few float ulps (one ulp of the steering command per tick)
        assert _eq(base["loop4"][1], other["loop4"][1]) and np.abs(base["loop4"][0] - other["loop4"][0]).max() < 1e-5


# ---- particle filter ------------------------------------------------------------------------------------------------------------
# Bit equality for every statement of src/particle_filter.cpp:25-148 but three 100-term sums — pw.sum() (:104), px * pw (:106),
# pw.transpose() * pw (:126) — which Eigen evaluates with vectorised redux / gemv kernels whose order is not restated
# (oracle/pf_ref.cpp takes them in index order).  Those three are compared at PF_SUM_TOL, relative to the largest entry, and bit
# for bit where the order cannot matter (two non-zero weights).
PF_SUM_TOL = 2e-6
NPF = 100


def _pf_close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) <= PF_SUM_TOL * max(1.0, np.max(np.abs(b)))


def _pf_case(rng, spread=1.0):
    px = np.stack([rng.normal(3, spread, NPF), rng.normal(4, spread, NPF), rng.uniform(-3.2, 3.2, NPF), rng.normal(1, 0.3, NPF)], axis=1).astype(np.float32)
    pw = rng.uniform(0.2, 1.0, NPF).astype(np.float32); pw /= pw.sum()
    nz = int(rng.integers(0, 5))
    rfid = np.array([[10.0, 0.0], [10.0, 10.0], [0.0, 15.0], [-5.0, 20.0]], np.float32)[:nz]
    z = np.concatenate([(np.hypot(3 - rfid[:, :1], 4 - rfid[:, 1:]) + rng.normal(0, 0.2, (nz, 1))).astype(np.float32), rfid], axis=1) if nz else np.zeros((0, 3), np.float32)
    u = np.array([1.0, 0.1], np.float32)
    nrm = rng.standard_normal((NPF, 2)).astype(np.float32)
    return px, pw.astype(np.float32), z.astype(np.float32), u, nrm


def _pf_oracle_loc(o, px, pw, z, u, nrm, rsim2):
    obs = np.zeros((1, 4, 3), np.float32); obs[0, : len(z)] = z
    p, w, xe, Pe, _, _ = o.pf_step_parts(px[None], pw[None], obs, np.array([len(z)], np.int32), u[None], nrm[None], np.ones((1, NPF), np.float32), 1, rsim=rsim2)
    return p[0], w[0], xe[0], Pe[0]


def test_pf_small_functions(host_trig):
    o = host_trig
    assert R.pf_np() == NPF
    rng = np.random.default_rng(70)
    n = 4000
    x = np.stack([rng.normal(0, 50, n), rng.normal(0, 50, n), rng.uniform(-200, 200, n), rng.normal(0, 5, n)], axis=1).astype(np.float32)
    x[:6, 2] = [0.0, -0.0, 1e-30, 3.1415927, 1e6, 120.1]
    u = np.stack([rng.normal(1, 2, n), rng.normal(0, 1, n)], axis=1).astype(np.float32)
    assert _eq(o.motion_model(x, u, trig=0), R.pf_motion_model(x, u))                      # :26-40, the EKF file's motion model again
    dz = np.concatenate([rng.normal(0, 0.3, n), rng.normal(0, 30, n), [0.0, -0.0, 1e-20, 50.0, np.inf]]).astype(np.float32)
    sg = np.concatenate([np.full(n, np.sqrt(np.float32(0.01)), np.float32), rng.uniform(0.01, 3.0, n).astype(np.float32), np.full(5, 0.1, np.float32)])
    assert np.array_equal(o.pf_gauss_likelihood(dz, sg), R.pf_gauss_likelihood(dz, sg), equal_nan=True)   # :53-57
    for _ in range(20):                                                                   # calc_covariance :59-71: sequential, exact
        px, pw, *_ = _pf_case(rng, spread=rng.uniform(0.1, 5.0))
        xe = rng.normal(3, 1, 4).astype(np.float32)
        assert _eq(o.pf_calc_covariance(xe, px, pw), R.pf_calc_covariance(xe, px, pw))


def test_pf_localization_two_live_particles_bit_exact(host_trig):
    """With at most two non-zero weights every sum of a tick has at most two non-zero terms: the whole pf_localization + resampling
    is order-independent and must equal the reference's lines bit for bit (weights, estimate, covariance, ancestors)."""
    o = host_trig
    rng = np.random.default_rng(71)
    rsim = (1.0, 0.0, 0.0, float(np.float32(o.oracle_lib.PF_RSIM[1])))
    resampled = 0
    for k in range(200):
        px, pw, z, u, nrm = _pf_case(rng)
        live = rng.choice(NPF, 2 if k % 4 else 1, replace=False)
        w2 = np.zeros(NPF, np.float32); w2[live] = rng.uniform(0.1, 1.0, len(live)).astype(np.float32)
        px[live, :2] = (np.array([3.0, 4.0]) + rng.normal(0, 0.05, (len(live), 2))).astype(np.float32)   # near the truth: weights that do not underflow
        nrm[live] *= 0.05
        pr, wr, xr, Pr = R.pf_localization(px, w2, z, u, nrm, rsim=rsim)
        po, wo, xo, Po = _pf_oracle_loc(o, px, w2, z, u, nrm, (rsim[0], rsim[3]))
        assert np.isfinite(wr).all() and (wr > 0).sum() == len(live), k
        assert _eq(po, pr) and _eq(wo, wr) and _eq(xo, xr) and _eq(Po, Pr), k
        uni = rng.uniform(1.0, 2.0, NPF).astype(np.float32)
        p2r, w2r = R.pf_resampling(pr, wr, uni)
        p2o, w2o, _, _, did, _ = o.pf_step_parts(po[None], wo[None], np.zeros((1, 4, 3), np.float32), np.zeros(1, np.int32), u[None], nrm[None], uni[None], 2)
        assert _eq(p2o[0], p2r) and _eq(w2o[0], w2r), k
        resampled += int(did[0])
    assert resampled == 200                                   # Neff <= 2 < NTh: every case resamples


def test_pf_localization_general(host_trig):
    """100 live particles: the motion of every particle bit for bit; the normalised weights, the estimate and the covariance
    through the three unpinned sums at PF_SUM_TOL."""
    o = host_trig
    rng = np.random.default_rng(72)
    rsim = (1.0, 0.0, 0.0, float(np.float32(o.oracle_lib.PF_RSIM[1])))
    for k in range(60):
        px, pw, z, u, nrm = _pf_case(rng, spread=rng.uniform(0.2, 3.0))
        pr, wr, xr, Pr = R.pf_localization(px, pw, z, u, nrm, rsim=rsim)
        po, wo, xo, Po = _pf_oracle_loc(o, px, pw, z, u, nrm, (rsim[0], rsim[3]))
        assert _eq(po, pr), k                                 # px.col(ip) = motion_model(x, ud): exact
        assert _pf_close(wo / wo.max(), wr / wr.max()) and _pf_close(xo, xr) and _pf_close(Po, Pr), k
        # the covariance given the reference's own estimate and weights: exact (sequential sum, :64-68)
        assert _eq(o.pf_calc_covariance(xr, pr, wr), Pr), k


def test_pf_resampling_on_given_weights(host_trig):
    """resampling() on identical weights: cumsum, the comb `base + uni/NP`, the ancestor walk and the reset of the weights — bit for
    bit; the Neff test itself rests on pw'pw (unpinned sum), so the cases keep clear of NTh = 50."""
    o = host_trig
    rng = np.random.default_rng(73)
    did_any = [0, 0]
    for k in range(120):
        px, _, _, u, nrm = _pf_case(rng)
        if k % 3 == 0:
            pw = rng.uniform(0.9, 1.0, NPF)                                   # nearly uniform: Neff ~ 99, no resampling
        else:
            pw = rng.exponential(1.0, NPF) ** rng.uniform(2.0, 6.0)           # peaked: Neff of a few
        pw = (pw / pw.sum()).astype(np.float32)
        neff = 1.0 / float(np.sum(pw.astype(np.float64) ** 2))
        assert abs(neff - 50.0) > 5.0
        uni = rng.uniform(1.0, 2.0, NPF).astype(np.float32)
        assert _eq(np.cumsum(pw, dtype=np.float32), R.pf_cumsum(pw))                        # cumsum :111-118 (numpy's is sequential too)
        pr, wr = R.pf_resampling(px, pw, uni)
        po, wo, _, _, did, anc = o.pf_step_parts(px[None], pw[None], np.zeros((1, 4, 3), np.float32), np.zeros(1, np.int32), u[None], nrm[None], uni[None], 2)
        assert _eq(po[0], pr) and _eq(wo[0], wr), k
        assert (did[0] == 1) == (neff < 50.0) and _eq(po[0], px[anc[0]])
        did_any[int(did[0])] += 1
    assert did_any[0] >= 30 and did_any[1] >= 60


def test_pf_main_loop_input_side_and_first_ticks(host_trig):
    """main() of particle_filter.cpp on an injected stream: the constants (:220-230), the input side of every pass (:251-268: ud,
    xTrue, xDR, the range observations of the landmarks in sight) bit for bit for 300 passes, and the filter itself — called with
    the generator and the distributions BY VALUE (:270-271): its 2*NP draws are the stream positions main uses next, and every
    resampling call sees the same uniforms — through the first passes at PF_SUM_TOL."""
    o = host_trig
    O = o.oracle_lib
    steps = 300
    rng = np.random.default_rng(74)
    w = rng.standard_normal(8 * steps + 2 * NPF + 8).astype(np.float32)
    uni = rng.uniform(1.0, 2.0, NPF).astype(np.float32)
    r = R.pf_main(steps, w, uni)
    c = r["consts"]
    assert c[0] == np.float32(0.01) and c[1] == np.float32(0.04) and c[2] == np.float32(O.PF_RSIM[0]) and c[3] == np.float32(O.PF_RSIM[1])
    # replay: which stream positions main consumed is decided by how many landmarks were in sight
    u = np.array([[1.0, 0.1]], np.float32)
    k = 0
    xT = np.zeros((1, 4), np.float32); xD = np.zeros((1, 4), np.float32)
    px = np.zeros((1, NPF, 4), np.float32); pw = np.full((1, NPF), np.float32(1.0 / NPF), np.float32)
    seen_counts = set()
    for t in range(steps):
        w_u = w[k:k + 2].reshape(1, 1, 2)
        nz = int(r["nz"][t])
        seen_counts.add(nz)
        # the oracle draws one range noise per landmark slot; main draws only for the landmarks in sight, in landmark order
        xt_next = o.motion_model(xT, u, trig=0)
        d = np.hypot(xt_next[0, 0] - O.PF_RFID[:, 0], xt_next[0, 1] - O.PF_RFID[:, 1])
        w_z = np.zeros((1, 1, 4), np.float32)
        w_z[0, 0, np.flatnonzero(d <= 20.0)[:nz]] = w[k + 2:k + 2 + nz]
        ud, obs, nobs, xth, xdh = o.pf_simulate_inputs(u, xT, xD, w_u, w_z)
        assert nobs[0, 0] == nz
        assert _eq(ud[0, 0], r["ud"][t]) and _eq(xth[0, 0], r["xTrue"][t]) and _eq(xdh[0, 0], r["xDR"][t]) and _eq(obs[0, 0], r["z"][t]), t
        xT, xD = xth[0], xdh[0]
        k += 2 + nz
        if t < 12:        # the filter: its copy of the stream starts where main stands now (and main does not advance past it)
            nrm = w[k:k + 2 * NPF].reshape(1, NPF, 2)
            px, pw, xe, Pe, did, _ = o.pf_step_parts(px, pw, obs[0], nobs[0], u, nrm, uni[None], 3)
            assert _pf_close(px[0], r["px"][t]) and _pf_close(pw[0] * NPF, r["pw"][t] * NPF) and _pf_close(xe[0], r["xEst"][t]) and _pf_close(Pe[0], r["PEst"][t]), t
    assert r["draws_used"] == k and len(seen_counts) >= 2

"""The HIP path against the reference's OWN source lines, without the oracle in between.

oracle/_ref/libref.so (oracle/ref_build.sh: the cited line ranges of /root/reference compiled unmodified) is built in the
development container and travels to the GPU box as a prebuilt file; nothing here reads /root/reference.  The other GPU tests
compare the kernels with the oracle and tests/test_oracle_vs_ref.py the oracle with these lines — this file closes the triangle on
the hot path itself: EKF, DARE / dlqr, the tracking closed loops, the dynamic-window and Frenet episodes (all bit for bit), and the
MPC solution judged by the reference's FG_EVAL."""
import numpy as np
import pytest

from common import ekf_QR, ekf_agents, ekf_noise, lqr_course, lqr_speeds, mpc_problem, tracking_agents

from oracle import ref_lib as R

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref.so was not shipped")]


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize("single", [True, False])
def test_ekf_run_equals_the_reference_lines(crx, oracle_mod, single):
    Q, Rm = ekf_QR()
    n, T = (1, 1000) if single else (200, 250)
    u, x0, P0 = ekf_agents(n, 3, single_vehicle=single)
    z, ud, *_ = oracle_mod.ekf_simulate_inputs(u, x0, x0, ekf_noise(T, n, 4))
    xr, Pr, xhr, phr = R.ekf_run(x0, P0, z, ud, Q, Rm, want_phist=True)          # motion_model ... ekf_estimation, :22-78, T passes
    import torch
    xd, Pd = _t(x0), _t(P0)
    xh = torch.empty((T, n, 4), device="cuda"); ph = torch.empty((T, n, 16), device="cuda")
    crx.ekf_run(xd, Pd, _t(z), _t(ud), Q, Rm, x_hist=xh, P_hist=ph)
    assert _eq(xh.cpu().numpy(), xhr) and _eq(ph.cpu().numpy(), phr) and _eq(xd.cpu().numpy(), xr) and _eq(Pd.cpu().numpy(), Pr)


@pytest.mark.parametrize("dim", [5, 4])
def test_dare_and_dlqr_equal_the_reference_lines(crx, oracle_mod, dim):
    v = lqr_speeds(1000, 11)
    v[:4] = [0.0, 1e-3, -1e-3, 2.7777777]
    A, B, Q, Rm = oracle_mod.lqr_build(v, dim)                                    # A, B as lqr_steering_control builds them
    Xr, Kr = R.dare(A, B, Q, Rm)                                                  # solve_DARE + dlqr of the file of that dimension
    K, X, _ = crx.dlqr_from_v(_t(v), dim=dim)                                     # the structured kernel
    assert _eq(X.cpu().numpy(), Xr) and _eq(K.cpu().numpy(), Kr)
    Xd, _ = crx.solve_DARE(_t(A), _t(B), _t(Q), _t(Rm))                           # the dense kernel
    assert _eq(Xd.cpu().numpy(), Xr) and _eq(crx.dlqr(_t(A), _t(B), _t(Q), _t(Rm)).cpu().numpy(), Kr)


@pytest.mark.parametrize("dim", [5, 4])
def test_lqr_closed_loop_equals_the_reference_lines(crx, dim):
    course, goal = lqr_course()
    st = np.zeros((24, 4), np.float32)                                            # agent 0: the reference's own start
    st[1:] = tracking_agents(23, tuple(c[:80] for c in course), 31, spread=0.3)
    sr, tr, hr = R.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=700)      # closed_loop_prediction's loop, every tick
    sd = _t(st)
    ticks, hist = crx.closed_loop_prediction(sd, crx.Course.from_numpy(course), goal, dim=dim, max_ticks=700, want_hist=True)
    ticks, hist = ticks.cpu().numpy(), hist.cpu().numpy()
    assert _eq(ticks, tr) and (tr < 700).all()
    for a in range(len(st)):
        assert _eq(hist[: tr[a], a], hr[: tr[a], a])
    assert _eq(sd.cpu().numpy(), sr)


def test_dwa_episode_equals_the_reference_lines(crx, oracle_mod):
    O = oracle_mod.oracle_lib
    rng = np.random.default_rng(61)
    n = 12
    st = np.stack([rng.uniform(-1, 9, n), rng.uniform(-1, 9, n), rng.uniform(-3.2, 3.2, n), rng.uniform(-0.5, 1.0, n),
                   rng.uniform(-0.69, 0.69, n)], axis=1).astype(np.float32)
    st[0] = (0.0, 0.0, 3.141592653 / 8.0, 0.0, 0.0)                               # the reference's start (:167)
    u = st[:, 3:5].copy()
    goal = np.stack([rng.uniform(8, 12, n), rng.uniform(8, 12, n)], axis=1).astype(np.float32)
    goal[0] = (10.0, 10.0)
    sr, ur, tr, hr = R.dwa_run(st, u, goal, O.DWA_OBSTACLES, 80)                  # dwa_control -> motion -> goal test, :192-221
    sd, ud = _t(st), _t(u)
    ticks, hist, *_ = crx.dwa_run(sd, ud, _t(goal), _t(O.DWA_OBSTACLES), 80, want_hist=True)
    ticks, hist = ticks.cpu().numpy(), hist.cpu().numpy()
    assert _eq(ticks, tr) and _eq(sd.cpu().numpy(), sr) and _eq(ud.cpu().numpy(), ur)
    for a in range(n):
        assert _eq(hist[: tr[a], a], hr[: tr[a], a])


def test_frenet_episode_equals_the_reference_lines(crx, oracle_mod):
    """frenet_optimal_planning + the hand-over and goal test of main (:160-176, :224-236) compiled from the reference's own
    lines (with its Spline2D / QuinticPolynomial / QuarticPolynomial headers), against the kernel: all 168 candidate costs and
    verdicts of a first call, then whole episodes tick by tick — the reference's own scenario (agent 0) from start to goal."""
    O = oracle_mod.oracle_lib
    course = crx.FrenetCourse(O.FRENET_WX, O.FRENET_WY)
    assert _eq(course.coef, R.frenet_spline_build(O.FRENET_WX, O.FRENET_WY))      # the spline table of the C ABI == Spline2D's
    ob, goal = O.FRENET_OBSTACLES, course.goal
    n = 12
    rng = np.random.default_rng(77)
    st = np.stack([rng.uniform(0.0, 40.0, n), rng.uniform(1.0, 9.0, n), rng.uniform(-3.0, 3.0, n), rng.uniform(-0.8, 0.8, n),
                   rng.uniform(-0.5, 0.5, n)], axis=1).astype(np.float32)
    st[0] = O.FRENET_STATE0                                                       # the reference's start (:215-219)
    r1 = R.frenet_run(st, O.FRENET_WX, O.FRENET_WY, goal, ob, 1, want_paths=True, cap=168)
    k1 = crx.frenet_optimal_planning(_t(st), course, _t(ob), want_paths=True)
    assert (r1["n_paths"] == 168).all()
    assert _eq(k1["path_cf"].cpu().numpy(), r1["path_cf"]) and _eq(k1["path_ok"].cpu().numpy(), r1["path_ok"])
    # 25 ticks for everybody (the reference's lines run into undefined behaviour once a path leaves the course: an agent that
    # drives off its end cannot be followed further), the reference's own scenario from start to goal
    for sel, max_ticks in ((slice(0, n), 25), (slice(0, 1), 200)):
        rr = R.frenet_run(st[sel], O.FRENET_WX, O.FRENET_WY, goal, ob, max_ticks)
        sd = _t(st[sel])
        k = crx.frenet_run(sd, course, _t(ob), max_ticks, want_hist=True)
        ticks, hist = k["ticks"].cpu().numpy(), k["hist"].cpu().numpy()
        assert _eq(ticks, rr["ticks"]) and _eq(k["status"].cpu().numpy() & 1, rr["status"] & 1)
        for a in range(len(ticks)):
            assert _eq(hist[: ticks[a], a], rr["hist"][: ticks[a], a]), a
        assert _eq(sd.cpu().numpy(), rr["state"])
    assert rr["status"][0] == 0 and rr["ticks"][0] == 98                          # the reference's scenario reaches its goal in 98 calls


@pytest.mark.parametrize("T", [6, 21])
def test_mpc_solution_judged_by_the_reference_fg_eval(crx, T):
    """The kernel's answer handed to FG_EVAL::operator() (:199-252) as IPOPT would see it: the 4T equality constraints hold (the
    initial-state rows return x0, the dynamics rows vanish to float round-off of the returned vector), fg[0] is the cost the
    kernel reports, and the bounds mpc_solve declares (:283-301) hold."""
    n = 64
    x0, xref = mpc_problem(n, T, seed=90 + T)
    sol, status, cost = crx.mpc_solve(_t(x0), _t(xref), T, return_status=True)
    sol, status, cost = sol.cpu().numpy(), status.cpu().numpy(), cost.cpu().numpy()
    cap = R.mpc_solve(x0[0], xref[0], T)
    lo, hi = cap["xl"], cap["xu"]
    init = 1 + np.array([0, T, 2 * T, 3 * T])
    for a in range(n):
        fg = R.mpc_fg_eval(xref[a], sol[a].astype(np.float64), T)
        assert np.array_equal(fg[init], x0[a].astype(np.float64))
        dyn = np.delete(fg[1:], init - 1)
        assert np.abs(dyn).max() <= 3.0 * 2.0 ** -24 * max(1.0, np.abs(sol[a]).max())    # the solution crosses the ABI as float: a few ulps of a coordinate
        assert abs(fg[0] - cost[a]) <= 2e-5 * max(1.0, abs(cost[a]))
        assert (sol[a] >= lo - 1e-6).all() and (sol[a] <= hi + 1e-6).all()
    assert (status & 1).mean() > 0.95

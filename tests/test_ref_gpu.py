"""The HIP path against the reference's OWN source lines, without the oracle in between.

oracle/_ref/libref.so (oracle/ref_build.sh: the cited line ranges of /root/reference compiled unmodified) is built in the
development container and travels to the GPU box as a prebuilt file; nothing here reads /root/reference.  The other GPU tests
compare the kernels with the oracle and tests/test_oracle_vs_ref.py the oracle with these lines — this file closes the triangle on
the hot path itself: EKF, DARE / dlqr, the tracking closed loops, the dynamic-window and Frenet episodes (all bit for bit), a
particle-filter tick (bit for bit up to its three reordered sums), and the
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


def test_pf_tick_equals_the_reference_lines(crx, oracle_mod):
    """pf_localization + resampling (src/particle_filter.cpp:73-148) compiled from the reference's own lines, draws injected, against
    the kernel.  The kernel takes the three 100-term sums of a tick as a tree over the wavefront, Eigen in its own vectorised order:
    with at most two non-zero weights the order cannot matter and the tick must agree bit for bit (particles, weights, estimate,
    ancestors; of the covariance the upper triangle — the kernel mirrors it, the reference rounds (w dx_r) dx_c and (w dx_c) dx_r
    separately); with 100 live particles the motion of every particle bit for bit and the sums at 2e-6."""
    NP = R.pf_np()
    rng = np.random.default_rng(83)
    rsim4 = (1.0, 0.0, 0.0, float(np.float32(oracle_mod.oracle_lib.PF_RSIM[1])))
    RFID = np.array([[10.0, 0.0], [10.0, 10.0], [0.0, 15.0], [-5.0, 20.0]], np.float32)
    u = np.array([1.0, 0.1], np.float32)
    n = 48
    px = np.stack([rng.normal(3, 1.0, (n, NP)), rng.normal(4, 1.0, (n, NP)), rng.uniform(-3.2, 3.2, (n, NP)), rng.normal(1, 0.3, (n, NP))], axis=2).astype(np.float32)
    pw = np.zeros((n, NP), np.float32)
    nrm = rng.standard_normal((n, NP, 2)).astype(np.float32)
    uni = rng.uniform(1.0, 2.0, (n, NP)).astype(np.float32)
    nobs = rng.integers(0, 5, n).astype(np.int32)
    obs = np.zeros((n, 4, 3), np.float32)
    for a in range(n):
        k = nobs[a]
        obs[a, :k] = np.concatenate([np.hypot(3 - RFID[:k, :1], 4 - RFID[:k, 1:]) + rng.normal(0, 0.2, (k, 1)), RFID[:k]], axis=1)
        if a < n // 2:                           # two (or one) live particles, near the truth so that their weights do not underflow
            live = rng.choice(NP, 2 if a % 4 else 1, replace=False)
            pw[a, live] = rng.uniform(0.1, 1.0, len(live))
            px[a, live, :2] = np.array([3.0, 4.0]) + rng.normal(0, 0.05, (len(live), 2))
            nrm[a, live] *= 0.05
        else:
            w = rng.uniform(0.2, 1.0, NP); pw[a] = w / w.sum()
    ref = []
    for a in range(n):
        pr, wr, xr, Pr = R.pf_localization(px[a], pw[a], obs[a, :nobs[a]], u, nrm[a], rsim=rsim4)
        p2, w2 = R.pf_resampling(pr, wr, uni[a])
        ref.append((pr, wr, xr, Pr, p2, w2))
    pxd, pwd = _t(px), _t(pw)
    xe, Pe, _, nres = crx.pf_run(pxd, pwd, _t(obs[None]), _t(nobs[None]), _t(np.repeat(u[None, None], n, 1)), _t(nrm[None]), _t(uni[None]))
    pxk, pwk, xe, Pe = pxd.cpu().numpy(), pwd.cpu().numpy(), xe.cpu().numpy(), Pe.cpu().numpy().reshape(n, 4, 4)     # [c][r] column-major
    close = lambda g, w: np.max(np.abs(g.astype(np.float64) - w)) <= 2e-6 * max(1.0, np.max(np.abs(w)))
    for a in range(n):
        pr, wr, xr, Pr, p2, w2 = ref[a]
        Prm = Pr.reshape(4, 4)
        if a < n // 2:
            assert _eq(pxk[a], p2) and _eq(pwk[a], w2) and _eq(xe[a], xr), a                     # the whole tick, ancestors included
            for c in range(4):
                for r in range(c + 1):
                    assert Pe[a, c, r] == Prm[c, r], (a, r, c)
            assert close(Pe[a], Prm)
        else:
            resampled = not _eq(w2, wr)
            if not resampled:
                assert _eq(pxk[a], pr)                                                           # motion model of every particle: exact
                assert close(pwk[a] * NP, wr * NP)
            assert close(xe[a], xr) and close(Pe[a], Prm), a


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

"""Pinning the EKF oracle without Eigen: independent numpy twin (bitwise), float64 bound, golden
fixtures, invariants, and the reference's own run shape (C1: one vehicle, u=(1.0,0.1))."""
import os

import numpy as np
import pytest

from common import bit_equal, ekf_QR, ekf_agents, ekf_noise, floored_rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run_twin(tw, x0, P0, z, ud, Q, R, order="eigen"):
    T, n = z.shape[0], x0.shape[0]
    Qm, Rm = Q.reshape(4, 4).T, R.reshape(2, 2).T
    xh = np.zeros((T, n, 4), np.float32); ph = np.zeros((T, n, 16), np.float32)
    for k in range(n):
        xe, Pe = x0[k].reshape(4, 1).copy(), P0[k].reshape(4, 4).T.copy()
        for t in range(T):
            xe, Pe = tw.ekf_estimation(xe, Pe, z[t, k].reshape(2, 1), ud[t, k].reshape(2, 1), Qm, Rm, order=order)
            xh[t, k] = xe.reshape(-1); ph[t, k] = Pe.T.reshape(-1)
    return xh, ph


def test_oracle_matches_numpy_twin_bitwise(oracle_mod):
    import oracle.np_twin as tw
    Q, R = ekf_QR()
    n, T = 3, 60
    u, x0, P0 = ekf_agents(n, 1)
    z, ud, _, _, _, _ = oracle_mod.ekf_simulate_inputs(u, x0, x0, ekf_noise(T, n, 7), trig=0)
    _, _, xh, ph = oracle_mod.ekf_run(x0, P0, z, ud, Q, R, trig=0, want_phist=True)
    xt, pt = _run_twin(tw, x0, P0, z, ud, Q, R)
    assert bit_equal(xh, xt) and bit_equal(ph, pt)


def test_small_functions_match_twin(oracle_mod):
    import oracle.np_twin as tw
    rng = np.random.default_rng(0)
    x = rng.uniform(-20, 20, (50, 4)).astype(np.float32); u = rng.uniform(-2, 2, (50, 2)).astype(np.float32)
    mm, jf = oracle_mod.motion_model(x, u, trig=0), oracle_mod.jacobF(x, u, trig=0)
    for k in range(50):
        assert bit_equal(tw.motion_model(x[k].reshape(4, 1), u[k].reshape(2, 1)).reshape(-1), mm[k])
        assert bit_equal(tw.jacobF(x[k].reshape(4, 1), u[k].reshape(2, 1)).T.reshape(-1), jf[k])
    assert bit_equal(oracle_mod.observation_model(x), x[:, :2])


def test_summation_order_is_irrelevant_on_the_ekf_path(oracle_mod):
    """Every non-packet-path product in ekf_estimation() multiplies by 0/1 selection matrices, so
    Eigen's tree/SSE orders and plain ascending order coincide (oracle/eigen_order.h)."""
    Q, R = ekf_QR()
    n, T = 64, 200
    u, x0, P0 = ekf_agents(n, 3)
    z, ud, _, _, _, _ = oracle_mod.ekf_simulate_inputs(u, x0, x0, ekf_noise(T, n, 4))
    a = oracle_mod.ekf_run(x0, P0, z, ud, Q, R, sum_order=0, want_phist=True)
    b = oracle_mod.ekf_run(x0, P0, z, ud, Q, R, sum_order=1, want_phist=True)
    assert bit_equal(a[2], b[2]) and bit_equal(a[3], b[3])


def test_float64_bound_and_invariants(oracle_mod):
    """float32 path within a few 1e-7 (floored) of a float64 evaluation of the same formulas over 1000
    steps; covariance symmetric to rounding; C1: the reference's own single-vehicle run shape."""
    Q, R = ekf_QR()
    T = 1000
    u, x0, P0 = ekf_agents(4, 0, single_vehicle=True)
    z, ud, xt, _, _, _ = oracle_mod.ekf_simulate_inputs(u, x0, x0, ekf_noise(T, 4, 21))
    x, P, xh, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
    # float64 reference of the textbook equations (:64-78)
    Qd, Rd = Q.reshape(4, 4).T.astype(np.float64), R.reshape(2, 2).T.astype(np.float64)
    H = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], dtype=np.float64)
    worst = 0.0
    for k in range(4):
        xe, Pe = np.zeros(4), np.eye(4)
        for t in range(T):
            u0, u1 = float(ud[t, k, 0]), float(ud[t, k, 1])
            xp = xe + np.array([0.1 * np.cos(xe[2]) * u0, 0.1 * np.sin(xe[2]) * u0, 0.1 * u1, u0])
            jF = np.eye(4)
            jF[0, 2] = -0.1 * u0 * np.sin(xp[2]); jF[0, 3] = 0.1 * np.cos(xp[2])
            jF[1, 2] = 0.1 * u0 * np.cos(xp[2]); jF[1, 3] = 0.1 * np.sin(xp[2])
            PP = jF @ Pe @ jF.T + Qd
            S = H @ PP @ H.T + Rd
            K = PP @ H.T @ np.linalg.inv(S)
            xe = xp + K @ (z[t, k].astype(np.float64) - H @ xp)
            Pe = (np.eye(4) - K @ H) @ PP
            worst = max(worst, floored_rel_err(xh[t, k], xe, 1.0))
        Pm = P[k].reshape(4, 4).T
        assert np.max(np.abs(Pm - Pm.T)) < 1e-6 and np.all(np.diag(Pm) > 0)
        assert floored_rel_err(Pm, Pe, np.max(np.abs(Pe))) < 5e-5
    # single precision over 1000 steps; the velocity state integrates to ~1000 (F_(3,3)=1.0 quirk),
    # so the floored error is relative to that magnitude
    assert worst < 2e-5
    assert 900 < x[0, 3] < 1100          # xEst(3) ~ 1000 after 1000 steps
    assert np.linalg.norm(x[0, :2] - xt[0, :2]) < 1.0   # tracks the true position


def test_golden_fixture(oracle_mod):
    """Committed fixture (tests/golden/make_golden.py): the oracle must keep reproducing it bit for bit,
    and the GPU tests compare the HIP path against the same file."""
    g = np.load(os.path.join(GOLD, "ekf_golden.npz"))
    x, P, xh, ph = oracle_mod.ekf_run(g["x0"], g["P0"], g["z"], g["ud"], g["Q"], g["R"], trig=1, want_phist=True)
    assert bit_equal(xh, g["x_hist"]) and bit_equal(ph[-1], g["P_final"])
    if oracle_mod.libm_is_fma_flavour():
        x2, P2, xh2, _ = oracle_mod.ekf_run(g["x0"], g["P0"], g["z"], g["ud"], g["Q"], g["R"], trig=0)
        assert bit_equal(xh2, g["x_hist"])
    z, ud, _, _, _, _ = oracle_mod.ekf_simulate_inputs(g["u_true"], g["x0"], g["x0"], g["w"], trig=1)
    assert bit_equal(z, g["z"]) and bit_equal(ud, g["ud"])

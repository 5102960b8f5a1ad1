"""The bit-exact kernels under NON-default parameters (the reference's #defines are parameters of the C ABI): the EKF's DT, the
LQR's DT / wheelbase L / eps / iteration cap, in the fused run, the structured DARE and the tracking controller + closed loop."""
import numpy as np
import pytest

from common import bit_equal, ekf_QR, ekf_agents, ekf_noise, lqr_course, lqr_speeds, tracking_agents

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("dt", [0.05, 0.1, 0.25, 1.0])
def test_ekf_run_other_time_steps(crx, oracle_mod, dt):
    import torch
    Q, R = ekf_QR()
    n, T = 300, 120
    u, x0, P0 = ekf_agents(n, 17)
    z, ud, *_ = oracle_mod.ekf_simulate_inputs(u, x0, x0, ekf_noise(T, n, 18))
    xo, Po, xho, pho = oracle_mod.ekf_run(x0, P0, z, ud, Q, R, dt=dt, want_phist=True)
    xd, Pd = _t(x0), _t(P0)
    xh = torch.empty((T, n, 4), device="cuda"); ph = torch.empty((T, n, 16), device="cuda")
    crx.ekf_run(xd, Pd, _t(z), _t(ud), Q, R, dt=dt, x_hist=xh, P_hist=ph)
    assert bit_equal(xh.cpu().numpy(), xho) and bit_equal(ph.cpu().numpy(), pho) and bit_equal(Pd.cpu().numpy(), Po)
    xs, Ps = _t(x0), _t(P0)                                               # the single-step entry point with the same DT
    crx.ekf_estimation(xs, Ps, _t(z[0]), _t(ud[0]), Q, R, dt=dt)
    assert bit_equal(xs.cpu().numpy(), xho[0]) and bit_equal(Ps.cpu().numpy(), pho[0])


@pytest.mark.parametrize("dim", [5, 4])
@pytest.mark.parametrize("dt,L,eps,maxiter", [(0.05, 0.5, 0.01, 150), (0.2, 2.5, 1e-3, 150), (0.1, 0.5, 1e-5, 37), (0.1, 1.0, 0.5, 3)])
def test_dare_from_v_other_parameters(crx, oracle_mod, dim, dt, L, eps, maxiter):
    v = lqr_speeds(1500, 5 + dim)
    A, B, Q, R = oracle_mod.lqr_build(v, dim, dt=dt, L=L)
    Xo, Ko, ito = oracle_mod.dare(A, B, Q, R, eps=eps, maxiter=maxiter)
    K, X, it = crx.dlqr_from_v(_t(v), dim=dim, dt=dt, L_wheelbase=L, eps=eps, maxiter=maxiter)
    assert np.array_equal(it.cpu().numpy(), ito) and bit_equal(X.cpu().numpy(), Xo) and bit_equal(K.cpu().numpy(), Ko)
    Xd, itd = crx.solve_DARE(_t(A), _t(B), _t(Q), _t(R), eps=eps, maxiter=maxiter)
    assert np.array_equal(itd.cpu().numpy(), ito) and bit_equal(Xd.cpu().numpy(), Xo)
    assert ito.max() <= maxiter


@pytest.mark.parametrize("dim", [5, 4])
def test_tracking_other_parameters(crx, oracle_mod, dim):
    course, goal = lqr_course()
    dc = crx.Course.from_numpy(course)
    kw = dict(dt=0.05, eps=1e-3, maxiter=60)
    st = tracking_agents(400, tuple(c[:150] for c in course), 23 + dim, spread=0.4)
    pe = np.zeros(len(st), np.float32); pth = np.zeros(len(st), np.float32)
    co, io, peo, ptho = oracle_mod.lqr_steering_control(st, course, pe, pth, dim=dim, L=0.8, **kw)
    ped, pthd = _t(pe), _t(pth)
    ctl, ind = crx.lqr_steering_control(_t(st), dc, ped, pthd, dim=dim, L_wheelbase=0.8, **kw)
    assert np.array_equal(ind.cpu().numpy(), io) and bit_equal(ctl.cpu().numpy(), co) and bit_equal(ped.cpu().numpy(), peo)
    so, to, ho, *_ = oracle_mod.lqr_closed_loop(st[:64], course, goal, dim=dim, max_ticks=900, L=0.8, want_hist=True, **kw)
    sd = _t(st[:64])
    ticks, hist = crx.closed_loop_prediction(sd, dc, goal, dim=dim, max_ticks=900, L_wheelbase=0.8, want_hist=True, **kw)
    ticks, hist = ticks.cpu().numpy(), hist.cpu().numpy()
    assert np.array_equal(ticks, to) and bit_equal(sd.cpu().numpy(), so)
    for a in range(64):
        assert bit_equal(hist[: to[a], a], ho[: to[a], a])

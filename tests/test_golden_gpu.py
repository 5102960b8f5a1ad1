"""GPU vs the committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from common import bit_equal, floored_rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_ekf_golden(crx):
    import torch
    g = np.load(os.path.join(GOLD, "ekf_golden.npz"))
    T, n = g["z"].shape[:2]
    xt, xd = _t(g["x0"]), _t(g["x0"])
    z, ud = crx.ekf_simulate_inputs(_t(g["u_true"]), xt, xd, _t(g["w"]))
    assert bit_equal(z.cpu().numpy(), g["z"]) and bit_equal(ud.cpu().numpy(), g["ud"])
    x, P = _t(g["x0"]), _t(g["P0"])
    xh = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
    crx.ekf_run(x, P, z, ud, g["Q"], g["R"], x_hist=xh)
    assert bit_equal(xh.cpu().numpy(), g["x_hist"]) and bit_equal(P.cpu().numpy(), g["P_final"])


def test_lqr_golden(crx):
    g = np.load(os.path.join(GOLD, "lqr_golden.npz"))
    for dim in (5, 4):
        K, X, it = crx.dlqr_from_v(_t(g["v"]), dim=dim)
        assert bit_equal(X.cpu().numpy(), g[f"X{dim}"]) and bit_equal(K.cpu().numpy(), g[f"K{dim}"])
        assert np.array_equal(it.cpu().numpy(), g[f"it{dim}"])


def test_mpc_golden(crx):
    g = np.load(os.path.join(GOLD, "mpc_golden.npz"))
    sol, st, cost = crx.mpc_solve(_t(g["x0"]), _t(g["xref"]), int(g["T"]), return_status=True)
    st, cost = st.cpu().numpy(), cost.cpu().numpy()
    assert np.array_equal(st & 3, g["status"] & 3)                       # every agent, converged or not
    ok = (g["status"] & 1) == 1
    assert ok.mean() > 0.95
    assert floored_rel_err(sol.cpu().numpy()[ok], g["sol"][ok], 1.0) <= 1e-6
    assert np.max(np.abs(cost - g["cost"]) / np.maximum(np.abs(g["cost"]), 1.0)) <= 1e-6


def test_track_golden(crx):
    g = np.load(os.path.join(GOLD, "track_golden.npz"))
    course = tuple(g["course"])
    dc = crx.Course.from_numpy(course)
    st = g["state"]
    for dim in (5, 4):
        pe, pth = _t(g["pe"]), _t(g["pth"])
        ctl, ind = crx.lqr_steering_control(_t(st), dc, pe, pth, dim=dim)
        assert bit_equal(ctl.cpu().numpy(), g[f"ctl{dim}"]) and np.array_equal(ind.cpu().numpy(), g[f"ind{dim}"])
        assert bit_equal(pe.cpu().numpy(), g[f"pe{dim}"]) and bit_equal(pth.cpu().numpy(), g[f"pth{dim}"])
        sd = _t(st)
        ticks, _ = crx.closed_loop_prediction(sd, dc, tuple(g["goal"]), dim=dim, max_ticks=600)
        assert np.array_equal(ticks.cpu().numpy(), g[f"loop_ticks{dim}"]) and bit_equal(sd.cpu().numpy(), g[f"loop_state{dim}"])
    for mpc, key in ((False, "update_lqr"), (True, "update_mpc")):
        sd = _t(st)
        crx.update(sd, _t(g["a"]), _t(g["delta"]), crx.vehicle_params(mpc))
        assert bit_equal(sd.cpu().numpy(), g[key])
    mdc = crx.Course.from_numpy(tuple(g["mcourse"]))
    td = _t(g["tind0"])
    xr = crx.calc_ref_trajectory(_t(g["mstate"]), mdc, td, 21)
    assert bit_equal(xr.cpu().numpy(), g["xref21"]) and np.array_equal(td.cpu().numpy(), g["tind"])



def test_planner_golden(crx):
    """DWA and Frenet bit for bit (fixtures = the reference's own lines; the Frenet one is defined on glibc's double pow / sin / cos —
    the kernel takes pow from the host libm and its own sin / cos agree with glibc's on every sample of this fixture)."""
    g = np.load(os.path.join(GOLD, "planner_golden.npz"))
    # dynamic window: one control step, then a 60-tick episode
    from cpprobotics_amd.dwa import dwa_run
    sd, ud = _t(g["dwa_state"]), _t(g["dwa_u"])
    _, _, status, best, ns = dwa_run(sd, ud, _t(g["dwa_goal"]), _dwa_ob(), 1)
    assert np.array_equal(ns.cpu().numpy(), g["dwa_ns"]) and np.array_equal(best.cpu().numpy(), g["dwa_best"])
    assert bit_equal(ud.cpu().numpy(), g["dwa_u1"])
    sd, ud = _t(g["dwa_state"]), _t(g["dwa_u"])
    ticks, *_ = dwa_run(sd, ud, _t(g["dwa_goal"]), _dwa_ob(), 60)
    assert np.array_equal(ticks.cpu().numpy(), g["dwa_ticks60"])
    assert bit_equal(sd.cpu().numpy(), g["dwa_state60"]) and bit_equal(ud.cpu().numpy(), g["dwa_u60"])
    # Frenet: course built by the product's host helper, one planning call, two episodes
    course = crx.FrenetCourse(g["fr_wx"], g["fr_wy"])
    assert bit_equal(course.coef, g["fr_coef"])
    assert len(course.rx) == int(g["fr_nsamples"]) and bit_equal(np.asarray(course.goal, np.float32), g["fr_goal"])
    ob = _t(g["fr_ob"])
    sd = _t(g["fr_state"])
    r = crx.frenet_optimal_planning(sd, course, ob, want_paths=True)
    assert np.array_equal(r["path_cf"].cpu().numpy(), g["fr_path_cf"], equal_nan=True)
    assert np.array_equal(r["path_ok"].cpu().numpy(), g["fr_path_ok"])
    best = r["best_idx"].cpu().numpy()
    assert np.array_equal(best, g["fr_best"]) and np.array_equal(r["n_valid"].cpu().numpy(), g["fr_nvalid"])
    moved = best >= 0
    assert bit_equal(r["hist"].cpu().numpy()[0][moved], g["fr_out"][moved])
    sd = _t(g["fr_state"][:6])
    r = crx.frenet_run(sd, course, ob, 120, crx.frenet_default_config(), want_hist=True)
    assert np.array_equal(r["ticks"].cpu().numpy(), g["fr_run_ticks"]) and np.array_equal(r["status"].cpu().numpy(), g["fr_run_status"])
    assert bit_equal(sd.cpu().numpy(), g["fr_run_state"])
    t0 = int(g["fr_run_ticks"][0])
    assert bit_equal(r["hist"].cpu().numpy()[:t0, 0], g["fr_run_hist0"])


def _dwa_ob():
    # the reference's obstacle list, src/dynamic_window_approach.cpp:169-180
    return _t(np.array([[-1, -1], [0, 2], [4.0, 2.0], [5.0, 4.0], [5.0, 5.0], [5.0, 6.0], [5.0, 9.0], [8.0, 9.0], [7.0, 9.0], [12.0, 12.0]],
                       np.float32))

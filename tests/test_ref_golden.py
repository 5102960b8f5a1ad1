"""The oracle against the committed outputs of the reference's own lines (tests/golden/*.npz) — runs on ANY host.

tests/test_oracle_vs_ref.py pins the oracle to oracle/_ref/libref.so directly but needs /root/reference to build that library.
The fixtures under tests/golden/ are that library's outputs (the generators tests/golden/make_golden*.py assert oracle ==
library before storing anything), so on a host that has neither the reference nor the shipped .so the pin survives here:
the oracle must reproduce the stored arrays bit for bit.  The fixtures are defined on glibc's FMA-flavour libm; the
oracle's deterministic trig mode 1 restates exactly that flavour, so the float functions do not depend on this host's libm.
The Frenet planner calls the host's double pow / sin / cos: equal bits on a glibc >= 2.28 host (every host of this
project), skipped with a message otherwise."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


@pytest.fixture()
def det_trig(oracle_mod, monkeypatch):
    monkeypatch.setattr(oracle_mod.oracle_lib, "trig_mode", lambda: 1)
    return oracle_mod


def test_ekf_fixture(det_trig):
    o = det_trig
    g = np.load(os.path.join(GOLD, "ekf_golden.npz"))
    z, ud, *_ = o.ekf_simulate_inputs(g["u_true"], g["x0"], g["x0"], g["w"], trig=1)
    assert _eq(z, g["z"]) and _eq(ud, g["ud"])
    x, P, xh, ph = o.ekf_run(g["x0"], g["P0"], z, ud, g["Q"], g["R"], trig=1, want_phist=True)
    assert _eq(xh, g["x_hist"]) and _eq(ph[-1], g["P_final"])


def test_lqr_fixture(oracle_mod):
    g = np.load(os.path.join(GOLD, "lqr_golden.npz"))
    for dim in (5, 4):
        A, B, Q, R = oracle_mod.lqr_build(g["v"], dim)
        X, K, it = oracle_mod.dare(A, B, Q, R)
        assert _eq(X, g[f"X{dim}"]) and _eq(K, g[f"K{dim}"]) and _eq(it, g[f"it{dim}"])


def test_mpc_fixture(oracle_mod):
    """The fixture is the oracle solver's own output (the reference's IPOPT result is not reproducible): a regression pin."""
    g = np.load(os.path.join(GOLD, "mpc_golden.npz"))
    sol, st, cost = oracle_mod.mpc_solve(g["x0"], g["xref"], int(g["T"]))
    assert _eq(st & 3, g["status"] & 3)
    ok = (g["status"] & 1) == 1
    assert np.max(np.abs(sol[ok] - g["sol"][ok]) / np.maximum(np.abs(g["sol"][ok]), 1.0)) <= 1e-6
    assert np.max(np.abs(cost - g["cost"]) / np.maximum(np.abs(g["cost"]), 1.0)) <= 1e-9


def test_track_fixture(det_trig):
    o = det_trig
    o.oracle_lib.lib().oracle_track_set_trig_mode(1)
    g = np.load(os.path.join(GOLD, "track_golden.npz"))
    course, st = tuple(g["course"]), g["state"]
    for dim in (5, 4):
        ctl, ind, pe, pth = o.lqr_steering_control(st, course, g["pe"], g["pth"], dim=dim)
        assert _eq(ctl, g[f"ctl{dim}"]) and _eq(ind, g[f"ind{dim}"]) and _eq(pe, g[f"pe{dim}"]) and _eq(pth, g[f"pth{dim}"])
        s1, ticks, *_ = o.lqr_closed_loop(st, course, tuple(g["goal"]), dim=dim, max_ticks=600)
        assert _eq(s1, g[f"loop_state{dim}"]) and _eq(ticks, g[f"loop_ticks{dim}"])
    assert _eq(o.update(st, g["a"], g["delta"]), g["update_lqr"])
    assert _eq(o.update(st, g["a"], g["delta"], dt=0.2, wheelbase=2.5, clamp_speed=True), g["update_mpc"])
    xr, tind = o.calc_ref_trajectory(g["mstate"], tuple(g["mcourse"]), g["tind0"], 21)
    assert _eq(xr, g["xref21"]) and _eq(tind, g["tind"])


def test_planner_fixture(det_trig):
    o = det_trig
    g = np.load(os.path.join(GOLD, "planner_golden.npz"))
    u1, ns, bi = o.dwa_control(g["dwa_state"], g["dwa_u"], g["dwa_goal"])
    assert _eq(u1, g["dwa_u1"]) and _eq(ns, g["dwa_ns"]) and _eq(bi, g["dwa_best"])
    s60, u60, t60, _ = o.dwa_run(g["dwa_state"], g["dwa_u"], g["dwa_goal"], 60)
    assert _eq(s60, g["dwa_state60"]) and _eq(u60, g["dwa_u60"]) and _eq(t60, g["dwa_ticks60"])
    coef = o.frenet_spline_build(g["fr_wx"], g["fr_wy"])
    assert _eq(coef, g["fr_coef"])
    p = o.frenet_plan(g["fr_state"], coef, g["fr_ob"])
    if not np.array_equal(p["path_cf"], g["fr_path_cf"], equal_nan=True):
        pytest.skip("this host's libm (double pow / sin / cos) is not the fixtures' glibc: the Frenet fixture is host-libm defined")
    assert _eq(p["path_ok"], g["fr_path_ok"]) and _eq(p["best"], g["fr_best"]) and _eq(p["n_valid"], g["fr_nvalid"])
    assert _eq(p["out"], g["fr_out"])
    r = o.frenet_run(g["fr_state"][:6], coef, g["fr_goal"], 120, g["fr_ob"], want_hist=True)
    assert _eq(r["ticks"], g["fr_run_ticks"]) and _eq(r["status"], g["fr_run_status"]) and _eq(r["state"], g["fr_run_state"])
    assert _eq(r["hist"][: r["ticks"][0], 0], g["fr_run_hist0"])


def test_pf_fixture(oracle_mod, monkeypatch):
    """pf_golden.npz = particle_filter.cpp's own lines on injected draws (tests/golden/make_golden_pf.py): bit equality for
    everything but the three Eigen reductions (see tests/test_oracle_vs_ref.py), those at 2e-6."""
    monkeypatch.setattr(oracle_mod.oracle_lib, "trig_mode", lambda: 0)      # this fixture is host-libm defined (cosf / sinf / expf)
    o = oracle_mod
    g = np.load(os.path.join(GOLD, "pf_golden.npz"))
    if not _eq(o.motion_model(g["mm_x"], g["mm_u"], trig=0), g["mm_out"]):
        pytest.skip("this host's libm (cosf / sinf) is not the fixtures' glibc")
    assert np.array_equal(o.pf_gauss_likelihood(g["gl_x"], g["gl_sigma"]), g["gl_out"], equal_nan=True)
    NP = g["two_px"].shape[1]
    for k in range(len(g["two_px"])):          # two live particles: the whole tick bit for bit
        nz = int(g["two_nz"][k])
        obs = np.zeros((1, 4, 3), np.float32); obs[0, :nz] = g["two_z"][k, :nz]
        px, pw, xe, Pe, did, _ = o.pf_step_parts(g["two_px"][k][None], g["two_pw"][k][None], obs, np.array([nz], np.int32), g["two_u"][None],
                                                g["two_nrm"][k][None], g["two_uni"][k][None], 3, rsim=tuple(g["rsim2"]))
        assert _eq(px[0], g["two_px_out"][k]) and _eq(pw[0], g["two_pw_out"][k]) and _eq(xe[0], g["two_xEst"][k]) and _eq(Pe[0], g["two_PEst"][k]), k
    for k in range(len(g["gen_px"])):          # 100 live particles: motion exact, sums at 2e-6; resampling on the given weights exact
        nz = int(g["gen_nz"][k])
        obs = np.zeros((1, 4, 3), np.float32); obs[0, :nz] = g["gen_z"][k, :nz]
        px, pw, xe, Pe, _, _ = o.pf_step_parts(g["gen_px"][k][None], g["gen_pw"][k][None], obs, np.array([nz], np.int32), g["two_u"][None],
                                              g["gen_nrm"][k][None], g["gen_uni"][k][None], 1, rsim=tuple(g["rsim2"]))
        assert _eq(px[0], g["gen_px_out"][k]), k
        close = lambda a, b: np.max(np.abs(a.astype(np.float64) - b)) <= 2e-6 * max(1.0, np.max(np.abs(b)))
        assert close(pw[0] / pw[0].max(), g["gen_pw_out"][k] / g["gen_pw_out"][k].max()) and close(xe[0], g["gen_xEst"][k]) and close(Pe[0], g["gen_PEst"][k]), k
        p2, w2, *_ = o.pf_step_parts(g["gen_px_out"][k][None], g["res_pw"][k][None], obs, np.array([nz], np.int32), g["two_u"][None],
                                     g["gen_nrm"][k][None], g["gen_uni"][k][None], 2)
        assert _eq(p2[0], g["res_px_out"][k]) and _eq(w2[0], g["res_pw_out"][k]), k

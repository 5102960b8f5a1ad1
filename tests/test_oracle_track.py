"""Pinning the tracking oracle (oracle/track_ref.cpp): independent numpy restatements and closed-loop behaviour."""
import math

import numpy as np
import pytest

from common import lqr_course, mpc_course_f32, tracking_agents


def _yaw_p2p(a):
    return math.fmod(math.fmod(float(a) + math.pi, 2 * math.pi) - 2 * math.pi, 2 * math.pi) + math.pi


def test_calc_nearest_index_matches_numpy(oracle_mod):
    course, _ = lqr_course()
    cx, cy, cyaw = course[:3]
    st = tracking_agents(200, course, 1)
    ind, e = oracle_mod.calc_nearest_index(st, course)
    for a in range(len(st)):
        d = (cx - st[a, 0]) ** 2 + (cy - st[a, 1]) ** 2          # float32 arithmetic, same order
        j = int(np.argmin(d))                                      # first minimum, like the strict '<' scan
        assert ind[a] == j
        ang = np.float32(_yaw_p2p(np.float32(cyaw[j] - np.float32(math.atan2(np.float32(cy[j] - st[a, 1]), np.float32(cx[j] - st[a, 0]))))))
        assert abs(abs(e[a]) - d[j]) == 0 and (e[a] < 0) == (ang < 0 and d[j] > 0)


def test_update_matches_float64_formula(oracle_mod):
    rng = np.random.default_rng(2)
    n = 500
    st = np.stack([rng.normal(0, 10, n), rng.normal(0, 10, n), rng.uniform(-7, 7, n), rng.uniform(-3, 16, n)], axis=1).astype(np.float32)
    a = rng.uniform(-1.5, 1.5, n).astype(np.float32)
    d = rng.uniform(-1.2, 1.2, n).astype(np.float32)
    for mpc in (False, True):
        dt, wb = (0.2, 2.5) if mpc else (0.1, 0.5)
        out = oracle_mod.update(st, a, d, dt=dt, wheelbase=wb, clamp_speed=mpc)
        dc = np.clip(d.astype(np.float64), -math.pi / 4, math.pi / 4)
        dc = np.where(np.abs(d) >= np.float32(math.pi / 4), np.float32(np.sign(d) * math.pi / 4).astype(np.float64), d.astype(np.float64))
        s = st.astype(np.float64)
        ref = np.stack([s[:, 0] + s[:, 3] * np.cos(s[:, 2]) * dt, s[:, 1] + s[:, 3] * np.sin(s[:, 2]) * dt,
                        s[:, 2] + s[:, 3] / wb * np.tan(dc) * dt, s[:, 3] + a * dt], axis=1)
        if mpc:
            ref[:, 3] = np.clip(ref[:, 3], -20 / 3.6, 55 / 3.6)
        assert np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1.0)) < 3e-7


@pytest.mark.parametrize("dim", [5, 4])
def test_lqr_closed_loop_reaches_the_goal(oracle_mod, dim):
    """The reference's own scenario (start at the origin, lqr_speed_steer_control.cpp:171) plus perturbed starts."""
    course, goal = lqr_course()
    st = np.zeros((6, 4), np.float32)
    st[1:] = tracking_agents(5, tuple(c[:60] for c in course), 3, spread=0.3)
    s, ticks, hist, pe, pth, ind = oracle_mod.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=1000, want_hist=True)
    gd = 0.3 if dim == 5 else 0.5
    assert (ticks < 1000).all() and (np.hypot(s[:, 0] - goal[0], s[:, 1] - goal[1]) <= gd + 1e-6).all()
    # the tracked path stays near the course: lateral error below 0.6 m after the first 3 s
    cx, cy = course[0].astype(np.float64), course[1].astype(np.float64)
    for a in range(len(st)):
        p = hist[30:ticks[a], a, :2].astype(np.float64)
        dmin = np.sqrt(((p[:, None, 0] - cx[None]) ** 2 + (p[:, None, 1] - cy[None]) ** 2).min(axis=1))
        assert dmin.max() < 0.6
    # one control evaluation == the first tick of the loop
    ctl, i0, pe0, pt0 = oracle_mod.lqr_steering_control(st, course, np.zeros(6, np.float32), np.zeros(6, np.float32), dim=dim)
    assert np.isfinite(ctl).all()


def test_calc_ref_trajectory_matches_numpy(oracle_mod):
    course, _ = mpc_course_f32()
    cx, cy, cyaw, ck, sp = course
    st = tracking_agents(100, course, 4, spread=1.0)
    T = 6
    tind0 = np.maximum(0, np.random.default_rng(5).integers(0, len(cx), 100) - 3).astype(np.int32)
    xref, tind = oracle_mod.calc_ref_trajectory(st, course, tind0, T)
    for a in range(100):
        lo, hi = tind0[a], min(tind0[a] + 10, len(cx))
        d = (cx[lo:hi] - st[a, 0]) ** 2 + (cy[lo:hi] - st[a, 1]) ** 2
        ind = max(int(lo + np.argmin(d)) if hi > lo else 0, int(tind0[a]))
        assert tind[a] == ind
        travel = np.float32(0)
        for i in range(T):
            travel = np.float32(np.float64(travel) + np.float64(abs(st[a, 3])) * 0.2)
            j = min(ind + int(math.floor(abs(travel / np.float32(1.0)) + 0.5)), len(cx) - 1)
            assert tuple(xref[a, 4 * i:4 * i + 4]) == (cx[j], cy[j], cyaw[j], sp[j])


def test_mpc_closed_loop_tracks_the_course(oracle_mod):
    course, goal = mpc_course_f32()
    st = np.array([[course[0][0], course[1][0], course[2][0], course[4][0]]], np.float32)
    s, ticks, hist, tind = oracle_mod.mpc_closed_loop(st, course, goal, T=6, max_ticks=60, want_hist=True)
    assert ticks[0] == 60 and tind[0] > 20                      # 12 s at ~10 km/h: > 20 m down the course
    p = hist[:, 0, :2].astype(np.float64)
    d = np.sqrt(((p[:, None, 0] - course[0][None]) ** 2 + (p[:, None, 1] - course[1][None]) ** 2).min(axis=1))
    assert d.max() < 1.0


def test_oracle_matches_committed_golden(oracle_mod):
    """tests/golden/track_golden.npz pins the tracking oracle against silent drift."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "track_golden.npz"))
    course, st = tuple(g["course"]), g["state"]
    for dim in (5, 4):
        ctl, ind, pe, pth = oracle_mod.lqr_steering_control(st, course, g["pe"], g["pth"], dim=dim)
        assert np.array_equal(ctl, g[f"ctl{dim}"]) and np.array_equal(ind, g[f"ind{dim}"])
        s1, ticks, *_ = oracle_mod.lqr_closed_loop(st, course, tuple(g["goal"]), dim=dim, max_ticks=600)
        assert np.array_equal(s1, g[f"loop_state{dim}"]) and np.array_equal(ticks, g[f"loop_ticks{dim}"])
    assert np.array_equal(oracle_mod.update(st, g["a"], g["delta"]), g["update_lqr"])
    xr, tind = oracle_mod.calc_ref_trajectory(g["mstate"], tuple(g["mcourse"]), g["tind0"], 21)
    assert np.array_equal(xr, g["xref21"]) and np.array_equal(tind, g["tind"])

"""The DWA oracle (oracle/dwa_ref.cpp): the reference's own scenario reaches its goal, windows have the expected size."""
import numpy as np


def test_reference_scenario_reaches_goal(oracle_mod):
    st = np.array([[0.0, 0.0, 3.141592653 / 8.0, 0.0, 0.0]], np.float32)          # main() :167
    u = np.zeros((1, 2), np.float32)
    goal = np.array([[10.0, 10.0]], np.float32)
    un, ns, bi = oracle_mod.dwa_control(st, u, goal)
    assert ns[0] == 5 * 81 and 0 <= bi[0] < ns[0]                                  # 5 speeds x 81 yaw rates from standstill
    assert abs(un[0, 0] - 0.02) < 1e-6                                             # full acceleration towards the goal
    s, uu, ticks, hist = oracle_mod.dwa_run(st, u, goal, 1000, want_hist=True)
    assert ticks[0] < 400 and np.hypot(s[0, 0] - 10.0, s[0, 1] - 10.0) <= 1.0
    ob = oracle_mod.oracle_lib.DWA_OBSTACLES
    p = hist[: ticks[0], 0, :2]
    d = np.sqrt(((p[:, None, :] - ob[None]) ** 2).sum(axis=2))
    assert d.min() > 1.0                                                           # never inside an obstacle's radius
    assert np.abs(np.diff(hist[: ticks[0], 0, 3])).max() <= 0.2 * 0.1 + 1e-6       # acceleration limit respected

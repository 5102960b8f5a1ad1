"""GPU parity of the Frenet optimal-trajectory planner (one agent per wavefront) against the CPU oracle: EQUAL BITS.

Every cost, every check_paths verdict, every winner, every tick of every episode must equal the oracle's (itself equal to the
reference's own lines bit for bit, tests/test_oracle_vs_ref.py).  The kernel's std::pow values are libm's (evaluated by the C ABI
entry point on the time grid), atan2f / sqrtf and the double cos / sin of frenet_optimal_trajectory.cpp:111-112 are glibc's
bit for bit (crx_fdlibm.h, crx_dsincos.h; tests/test_fdlibm.py, tests/test_dsincos.py): no operation is left that could differ.
KNOWN_EXCEPTIONS — (test, agent, path) -> reason — stays as the place where a deviation would have to be written down; it is empty."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# {(test name, agent, path or tick): reason}.  An entry exempts that one (agent, path) from the equal-bits demand.
KNOWN_EXCEPTIONS = {}


def _equal(a, b):
    """Equal as IEEE values (+0 == -0), NaN == NaN."""
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _states(n, seed, s_hi=70.0):
    rng = np.random.default_rng(seed)
    st = np.stack([rng.uniform(0.0, s_hi, n), rng.uniform(1.0, 9.0, n), rng.uniform(-3.0, 3.0, n), rng.uniform(-0.8, 0.8, n),
                   rng.uniform(-0.5, 0.5, n)], axis=1).astype(np.float32)
    st[0] = (0.0, 10.0 / 3.6, 2.0, 0.0, 0.0)                      # the reference's start
    return st


def _course(crx, oracle_mod):
    O = oracle_mod.oracle_lib
    return crx.FrenetCourse(O.FRENET_WX, O.FRENET_WY), O.FRENET_OBSTACLES


def _cfg(crx, **kw):
    c = crx.frenet_default_config()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


@pytest.mark.parametrize("n", [1, 5, 203, 2000])
def test_frenet_single_plan(crx, oracle_mod, n):
    course, ob = _course(crx, oracle_mod)
    st = _states(n, 10 + n)
    o = oracle_mod.frenet_plan(st, course.coef, ob)
    sd = _t(st)
    r = crx.frenet_optimal_planning(sd, course, _t(ob), _cfg(crx), want_paths=True)
    P = o["n_paths"][0]
    assert P == 168 and r["path_cf"].shape[1] == P
    assert not KNOWN_EXCEPTIONS
    assert _equal(r["path_cf"].cpu().numpy(), o["path_cf"])                      # all n x 168 costs
    assert _equal(r["path_ok"].cpu().numpy(), o["path_ok"])                      # all n x 168 check_paths verdicts
    assert _equal(r["best_idx"].cpu().numpy(), o["best"])                        # every winner
    assert _equal(r["n_valid"].cpu().numpy(), o["n_valid"]) and _equal(r["status"].cpu().numpy(), o["status"])
    moved = o["best"] >= 0
    h = r["hist"].cpu().numpy()[0]
    assert _equal(h[moved], np.concatenate([o["out"][moved, :5], o["out"][moved, 5:8]], axis=1))
    assert _equal(sd.cpu().numpy()[moved], o["out"][moved, :5])
    assert _equal(sd.cpu().numpy()[~moved], st[~moved]) and (r["status"].cpu().numpy()[~moved] & 1).all()


def test_frenet_reference_episode(crx, oracle_mod):
    course, ob = _course(crx, oracle_mod)
    O = oracle_mod.oracle_lib
    n, max_ticks = 17, 300
    st = np.repeat(O.FRENET_STATE0[None, :], n, axis=0)
    st[1:, 2] += np.linspace(-1.5, 1.5, n - 1).astype(np.float32)          # neighbours of the reference's start
    o = oracle_mod.frenet_run(st, course.coef, course.goal, max_ticks, ob, want_hist=True)
    sd = _t(st)
    r = crx.frenet_run(sd, course, _t(ob), max_ticks, _cfg(crx), want_hist=True)
    ticks = r["ticks"].cpu().numpy()
    h = r["hist"].cpu().numpy()
    # the reference's own start (agent 0) must reach the goal, as the reference does
    assert o["status"][0] == 0 and o["ticks"][0] < max_ticks
    assert np.hypot(h[ticks[0] - 1, 0, 5] - course.goal[0], h[ticks[0] - 1, 0, 6] - course.goal[1]) <= 1.0
    # every agent: same length, same fate, every tick of the trajectory (hand-over state, position, winner's cost) equal
    assert _equal(ticks, o["ticks"]) and _equal(r["status"].cpu().numpy(), o["status"])
    assert _equal(r["best_idx"].cpu().numpy(), o["best_idx"]) and _equal(r["n_valid"].cpu().numpy(), o["n_valid"])
    for a in range(n):
        assert _equal(h[: ticks[a], a], o["hist"][: ticks[a], a]), a
    assert _equal(sd.cpu().numpy(), o["state"])


def test_frenet_edge_cases(crx, oracle_mod):
    import torch
    course, ob = _course(crx, oracle_mod)
    s_end = float(course.coef[0, -1])

    def same(r, o):
        assert _equal(r["path_cf"].cpu().numpy(), o["path_cf"]) and _equal(r["path_ok"].cpu().numpy(), o["path_ok"])
        assert _equal(r["best_idx"].cpu().numpy(), o["best"]) and _equal(r["n_valid"].cpu().numpy(), o["n_valid"])
        assert _equal(r["status"].cpu().numpy(), o["status"])

    # (a) no obstacles; (b) agents about to run off the end of the course (paths truncated at csp.s.back(), some to < 2 points)
    st = _states(64, 3)
    st[:16, 0] = np.linspace(s_end - 6.0, s_end - 0.01, 16).astype(np.float32)
    st[16, 0] = s_end + 1.0                                           # beyond the course: every path has no point -> no survivor
    for obs in (np.zeros((0, 2), np.float32), ob):
        o = oracle_mod.frenet_plan(st, course.coef, obs)
        r = crx.frenet_optimal_planning(_t(st), course, _t(obs) if len(obs) else torch.zeros((0, 2), device="cuda"), want_paths=True)
        same(r, o)
        assert r["best_idx"].cpu().numpy()[16] == -1 and r["status"].cpu().numpy()[16] & 1
    # (c) an obstacle wall: nothing survives anywhere near it
    wall = np.stack([np.full(40, 12.0), np.linspace(-15, 5, 40)], axis=1).astype(np.float32)
    st2 = _states(8, 4); st2[:, 0] = 9.0
    o = oracle_mod.frenet_plan(st2, course.coef, wall)
    r = crx.frenet_optimal_planning(_t(st2), course, _t(wall), want_paths=True)
    same(r, o)
    # (d) a different sample grid (time step 0.25: other powers of the time grid) and course
    rng = np.random.default_rng(8)
    wx = np.cumsum(rng.uniform(6.0, 15.0, 12)).astype(np.float32); wy = rng.uniform(-6, 6, 12).astype(np.float32)
    c2 = crx.FrenetCourse(wx, wy)
    for kw in (dict(max_road_width=4.0, d_road_w=0.5, mint=3.0, maxt=4.1, dt=0.25, n_s_sample=2, d_t_s=1.0, robot_radius=1.0),
               dict(max_road_width=3.0, d_road_w=0.75, mint=2.1, maxt=3.0, dt=0.3, n_s_sample=1, d_t_s=2.0, robot_radius=0.7)):
        st3 = _states(50, 9, s_hi=float(c2.coef[0, -1]) * 0.9)
        ob3 = np.stack([rng.uniform(wx[0], wx[-1], 30), rng.uniform(-8, 8, 30)], axis=1).astype(np.float32)
        o = oracle_mod.frenet_plan(st3, c2.coef, ob3, cfg=oracle_mod.frenet_config(**kw))
        r = crx.frenet_optimal_planning(_t(st3), c2, _t(ob3), _cfg(crx, **kw), want_paths=True)
        assert r["path_cf"].shape[1] == o["n_paths"][0] == crx.frenet_num_paths(_cfg(crx, **kw))
        same(r, o)


def test_frenet_empty_batch(crx, oracle_mod):
    import torch
    course, ob = _course(crx, oracle_mod)
    r = crx.frenet_run(torch.zeros((0, 5), device="cuda"), course, _t(ob), 3)
    assert r["ticks"].numel() == 0

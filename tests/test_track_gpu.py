"""GPU parity of the course-tracking kernels (through the C ABI) against the CPU oracle — bit-exact."""
import ctypes as C
import math

import numpy as np
import pytest

from common import bit_equal, floored_rel_err, lqr_course, mpc_course_f32, tracking_agents

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def lqr_setup(crx):
    course, goal = lqr_course()
    return course, goal, crx.Course.from_numpy(course)


@pytest.mark.parametrize("n", [1, 63, 257, 2000])
def test_calc_nearest_index_bit_exact(crx, oracle_mod, lqr_setup, n):
    course, goal, dc = lqr_setup
    st = tracking_agents(n, course, n, spread=1.5)
    io, eo = oracle_mod.calc_nearest_index(st, course)
    ind, e = crx.calc_nearest_index(_t(st), dc)
    assert np.array_equal(ind.cpu().numpy(), io) and bit_equal(e.cpu().numpy(), eo)


@pytest.mark.parametrize("nc", [1, 2, 3, 64, 425, 8192, 8193, 20001])
def test_calc_nearest_index_course_lengths(crx, oracle_mod, nc):
    """Odd, even, tiny and very long courses: the scan reads two points per LDS word (an odd course is padded with a point that
    never compares smaller); courses beyond 8,192 points are scanned from global memory.  Ties (repeated points) keep the first."""
    rng = np.random.default_rng(nc)
    t = np.linspace(0.0, 6.0, nc)
    cx = (20.0 * np.cos(t) + rng.normal(0, 0.05, nc)).astype(np.float32)
    cy = (15.0 * np.sin(1.3 * t) + rng.normal(0, 0.05, nc)).astype(np.float32)
    if nc >= 64:
        cx[nc // 2] = cx[nc // 2 - 7]; cy[nc // 2] = cy[nc // 2 - 7]          # an exact duplicate: the earlier index must win
    cyaw = rng.uniform(-3, 3, nc).astype(np.float32)
    course = (cx, cy, cyaw, np.zeros(nc, np.float32), np.full(nc, 2.0, np.float32))
    n = 333
    st = np.stack([rng.uniform(-25, 25, n), rng.uniform(-20, 20, n), rng.uniform(-3, 3, n), rng.uniform(0, 3, n)], axis=1).astype(np.float32)
    if nc >= 64:
        st[:5, 0] = cx[nc // 2]; st[:5, 1] = cy[nc // 2]                       # agents sitting exactly on the duplicated point
    io, eo = oracle_mod.calc_nearest_index(st, course)
    ind, e = crx.calc_nearest_index(_t(st), crx.Course.from_numpy(course))
    assert np.array_equal(ind.cpu().numpy(), io) and bit_equal(e.cpu().numpy(), eo)
    if nc >= 64:
        assert (io[:5] == nc // 2 - 7).all()


@pytest.mark.parametrize("nc", [1, 2, 3, 5, 64, 425, 8063, 8064, 8065, 8192, 8193])
def test_tracking_course_lengths_both_layouts(crx, oracle_mod, nc):
    """lqr_steering_control and a short closed loop on odd, even, tiny and long courses, ties included: the four-lane layout splits the scan
    over the lanes of a quad (pairs j = r mod 4, lexicographic minimum of (distance, index)); courses whose staging would not leave room
    for its LDS gain slots (> 8,064 points) and courses beyond 8,192 points run one agent per lane."""
    from cpprobotics_amd.experimental import closed_loop_prediction_lanes
    rng = np.random.default_rng(nc + 1)
    t = np.linspace(0.0, 6.0, nc)
    cx = (20.0 * np.cos(t) + rng.normal(0, 0.05, nc)).astype(np.float32)
    cy = (15.0 * np.sin(1.3 * t) + rng.normal(0, 0.05, nc)).astype(np.float32)
    if nc >= 64:
        for k in (nc // 2, nc // 3):
            cx[k] = cx[k - 7]; cy[k] = cy[k - 7]                                 # exact duplicates: the earlier index must win
    cyaw = rng.uniform(-3, 3, nc).astype(np.float32)
    course = (cx, cy, cyaw, rng.uniform(-0.2, 0.2, nc).astype(np.float32), np.full(nc, 2.0, np.float32))
    dc = crx.Course.from_numpy(course)
    n = 133
    st = np.stack([rng.uniform(-25, 25, n), rng.uniform(-20, 20, n), rng.uniform(-3, 3, n), rng.uniform(0.2, 3, n)], axis=1).astype(np.float32)
    if nc >= 64:
        st[:5, 0] = cx[nc // 2]; st[:5, 1] = cy[nc // 2]
        st[5:9, 0] = cx[nc // 3]; st[5:9, 1] = cy[nc // 3]
    pe = rng.normal(0, 0.3, n).astype(np.float32); pth = rng.normal(0, 0.2, n).astype(np.float32)
    ind0 = rng.integers(0, nc, n).astype(np.int32)
    goal = (1e6, 1e6)
    for dim in (5, 4):
        co, io, peo, ptho = oracle_mod.lqr_steering_control(st, course, pe, pth, dim=dim, ind=ind0)
        ped, pthd, indd = _t(pe), _t(pth), _t(ind0)
        ctl, ind = crx.lqr_steering_control(_t(st), dc, ped, pthd, dim=dim, ind=indd)
        assert np.array_equal(ind.cpu().numpy(), io) and bit_equal(ctl.cpu().numpy(), co)
        assert bit_equal(ped.cpu().numpy(), peo) and bit_equal(pthd.cpu().numpy(), ptho)
        if nc >= 64:
            assert (io[:5] == nc // 2 - 7).all() and (io[5:9] == nc // 3 - 7).all()
        so, tio, histo, peo, ptho, indo = oracle_mod.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=6, want_hist=True, pe=pe, pth_e=pth, ind=ind0)
        for lanes in ((1, 4) if nc <= 8064 else (1,)):
            sd, ped, pthd, indd = _t(st), _t(pe), _t(pth), _t(ind0)
            ticks, hist = closed_loop_prediction_lanes(sd, dc, goal, lanes, dim=dim, max_ticks=6, want_hist=True, pe=ped, pth_e=pthd, ind=indd)
            assert np.array_equal(ticks.cpu().numpy(), tio) and bit_equal(sd.cpu().numpy(), so) and bit_equal(hist.cpu().numpy(), histo)
            assert np.array_equal(indd.cpu().numpy(), indo)
        if nc > 8064:
            with pytest.raises(crx.CrxError):
                closed_loop_prediction_lanes(_t(st), dc, goal, 4, dim=dim, max_ticks=1)


@pytest.mark.parametrize("dim", [5, 4])
@pytest.mark.parametrize("n", [1, 100, 1025, 33000])      # up to 32,768 agents: a DPP quad per agent; above: one agent per lane
def test_lqr_steering_control_bit_exact(crx, oracle_mod, lqr_setup, dim, n):
    course, goal, dc = lqr_setup
    rng = np.random.default_rng(n + dim)
    st = tracking_agents(n, course, n + 10 * dim, spread=0.8)
    st[: max(1, n // 10), 3] = rng.uniform(-0.08, 0.08, max(1, n // 10))      # near-zero speeds: the DARE iteration cap
    pe = rng.normal(0, 0.3, n).astype(np.float32)
    pth = rng.normal(0, 0.2, n).astype(np.float32)
    ind0 = rng.integers(0, len(course[0]), n).astype(np.int32)
    co, io, peo, ptho = oracle_mod.lqr_steering_control(st, course, pe, pth, dim=dim, ind=ind0)
    ped, pthd, indd = _t(pe), _t(pth), _t(ind0)
    ctl, ind = crx.lqr_steering_control(_t(st), dc, ped, pthd, dim=dim, ind=indd)
    assert np.array_equal(ind.cpu().numpy(), io)
    assert bit_equal(ped.cpu().numpy(), peo) and bit_equal(pthd.cpu().numpy(), ptho)
    assert bit_equal(ctl.cpu().numpy(), co)


@pytest.mark.parametrize("dim", [5, 4])
def test_lqr_steering_control_adversarial_curvatures(crx, oracle_mod, dim):
    """The feed-forward term `(float)atan2(L*k, 1.0)` on the curvatures where a merely accurate double atan is not enough: the
    values k on which OCML's device atan rounds to a different float than glibc's atan2 (found by the exhaustive device sweep,
    crx_x_datan2_sweep_dev), curvatures at every branch boundary of glibc's algorithm (|L k| = 1/16, 1, 16, 2^-57, 2^57), the
    table's own abscissae, huge / tiny / denormal / zero curvatures of both signs.  Equal bits with the oracle (host libm)."""
    from cpprobotics_amd.experimental import datan2_sweep
    _, diff, ks = datan2_sweep(0.5)
    rng = np.random.default_rng(17 + dim)
    edge = np.float32([0.0, -0.0, 0.125, 2.0, 32.0, 2.0 ** -56, 2.0 ** 58, 1e-45, 1e-38, 3e38, 0.12500001, 0.124999993, 1.9999999, 2.0000002,
                       31.999998, 32.000004, 0.2539, 0.1269, 1.998, 0.5, 1.0, 3.0, 7.0, 15.9, 100.0, 1e6])
    ck = np.concatenate([ks.cpu().numpy(), edge, -edge, rng.uniform(-4, 4, 400).astype(np.float32),
                         (rng.uniform(-1, 1, 400) * np.exp2(rng.integers(-40, 40, 400))).astype(np.float32)]).astype(np.float32)
    nc = ck.size
    t = np.linspace(0.0, 6.0, nc)
    cx = (20.0 * np.cos(t)).astype(np.float32); cy = (15.0 * np.sin(1.3 * t)).astype(np.float32)
    course = (cx, cy, rng.uniform(-3, 3, nc).astype(np.float32), ck, np.full(nc, 2.0, np.float32))
    st = np.stack([cx, cy, rng.uniform(-3, 3, nc), rng.uniform(0.2, 3, nc)], axis=1).astype(np.float32)    # agent j sits on course point j
    pe = rng.normal(0, 0.3, nc).astype(np.float32); pth = rng.normal(0, 0.2, nc).astype(np.float32)
    ind0 = np.zeros(nc, np.int32)
    co, io, peo, ptho = oracle_mod.lqr_steering_control(st, course, pe, pth, dim=dim, ind=ind0)
    assert len(set(io.tolist())) > nc * 0.9                      # the agents really read (nearly) every curvature of the list
    ped, pthd, indd = _t(pe), _t(pth), _t(ind0)
    ctl, ind = crx.lqr_steering_control(_t(st), crx.Course.from_numpy(course), ped, pthd, dim=dim, ind=indd)
    assert np.array_equal(ind.cpu().numpy(), io) and bit_equal(ctl.cpu().numpy(), co)
    print(f"adversarial list: {ks.numel()} curvatures on which OCML's atan rounds differently ({int(diff[1])} of 2^32 in all)")


@pytest.mark.parametrize("mpc", [False, True])
def test_update_bit_exact(crx, oracle_mod, mpc):
    rng = np.random.default_rng(7)
    n = 5000
    st = np.stack([rng.normal(0, 30, n), rng.normal(0, 30, n), rng.uniform(-40, 40, n), rng.uniform(-8, 18, n)], axis=1).astype(np.float32)
    a = rng.uniform(-2, 2, n).astype(np.float32)
    d = rng.uniform(-1.2, 1.2, n).astype(np.float32)
    d[:8] = np.float32([0.0, -0.0, math.pi / 4, -math.pi / 4, 0.78539816, 0.7853982, 1e-6, -3e-5])
    dt, wb = (0.2, 2.5) if mpc else (0.1, 0.5)
    ref = oracle_mod.update(st, a, d, dt=dt, wheelbase=wb, clamp_speed=mpc)
    sd = _t(st)
    crx.update(sd, _t(a), _t(d), crx.vehicle_params(mpc))
    assert bit_equal(sd.cpu().numpy(), ref)


@pytest.mark.parametrize("dim", [5, 4])
def test_lqr_closed_loop_bit_exact(crx, oracle_mod, lqr_setup, dim):
    """closed_loop_prediction for a batch: every tick of every agent equals the oracle's, including the tick count."""
    course, goal, dc = lqr_setup
    n = 300
    st = tracking_agents(n, tuple(c[:200] for c in course), 31 + dim, spread=0.4)
    st[0] = 0.0                                                  # the reference's own start state (:171)
    st[0, :2] = -0.0
    max_ticks = 700
    so, tio, histo, peo, ptho, indo = oracle_mod.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=max_ticks, want_hist=True)
    sd = _t(st)
    ticks, hist = crx.closed_loop_prediction(sd, dc, goal, dim=dim, max_ticks=max_ticks, want_hist=True)
    assert np.array_equal(ticks.cpu().numpy(), tio)
    assert (tio < max_ticks).mean() > 0.9                        # nearly all reach the goal
    assert bit_equal(sd.cpu().numpy(), so)
    h = hist.cpu().numpy()
    for a in range(0, n, 7):
        assert bit_equal(h[: tio[a], a], histo[: tio[a], a])


@pytest.mark.parametrize("lanes", [1, 4])
@pytest.mark.parametrize("dim", [5, 4])
def test_lqr_closed_loop_both_layouts(crx, oracle_mod, lqr_setup, dim, lanes):
    """One agent per lane and one agent per DPP quad, forced through the experimental entry point: every tick equals the oracle's.
    Batch sizes that leave part of the last wave / the last quad-wave empty; carried pe / pth_e / ind; near-zero speeds (DARE cap)."""
    from cpprobotics_amd.experimental import closed_loop_prediction_lanes
    course, goal, dc = lqr_setup
    for n, seed in ((1, 3), (17, 5), (203, 9)):
        rng = np.random.default_rng(seed + dim)
        st = tracking_agents(n, tuple(c[:200] for c in course), seed + dim, spread=0.4)
        st[: max(1, n // 8), 3] = rng.uniform(-0.08, 0.08, max(1, n // 8))
        pe = rng.normal(0, 0.3, n).astype(np.float32); pth = rng.normal(0, 0.2, n).astype(np.float32)
        ind0 = rng.integers(0, len(course[0]), n).astype(np.int32)
        max_ticks = 120
        so, tio, histo, peo, ptho, indo = oracle_mod.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=max_ticks, want_hist=True, pe=pe, pth_e=pth, ind=ind0)
        sd, ped, pthd, indd = _t(st), _t(pe), _t(pth), _t(ind0)
        ticks, hist = closed_loop_prediction_lanes(sd, dc, goal, lanes, dim=dim, max_ticks=max_ticks, want_hist=True, pe=ped, pth_e=pthd, ind=indd)
        assert np.array_equal(ticks.cpu().numpy(), tio)
        assert bit_equal(sd.cpu().numpy(), so)
        assert bit_equal(ped.cpu().numpy(), peo) and bit_equal(pthd.cpu().numpy(), ptho) and np.array_equal(indd.cpu().numpy(), indo)
        h = hist.cpu().numpy()
        for a in range(n):
            assert bit_equal(h[: tio[a], a], histo[: tio[a], a])


@pytest.mark.parametrize("dim", [5, 4])
def test_lqr_closed_loop_large_batch_variant(crx, oracle_mod, lqr_setup, dim):
    """Beyond 98,304 agents the one-lane loop runs the masked Riccati iteration (eight waves per SIMD); up to there the unmasked one (forced
    one-lane runs of the tests above), up to 32,768 a DPP quad per agent.  Two ticks of 100,001 agents through the product entry point."""
    course, goal, dc = lqr_setup
    n = 100001
    st = tracking_agents(n, tuple(c[:200] for c in course), 91 + dim, spread=0.4)
    so, tio, histo, peo, ptho, indo = oracle_mod.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=2, want_hist=True)
    sd = _t(st)
    ticks, hist = crx.closed_loop_prediction(sd, dc, goal, dim=dim, max_ticks=2, want_hist=True)
    assert np.array_equal(ticks.cpu().numpy(), tio) and bit_equal(sd.cpu().numpy(), so) and bit_equal(hist.cpu().numpy(), histo)


def test_lqr_closed_loop_nan_position_both_layouts(crx, oracle_mod, lqr_setup):
    """A NaN position never compares smaller in the course scan: the index keeps its incoming value (0 in the 5-state file, the caller's in the
    4-state file) in both layouts."""
    from cpprobotics_amd.experimental import closed_loop_prediction_lanes
    course, goal, dc = lqr_setup
    st = tracking_agents(9, tuple(c[:200] for c in course), 77, spread=0.4)
    st[2, 0] = np.nan; st[5, 1] = np.nan
    ind0 = np.arange(9, dtype=np.int32) * 11
    for dim in (5, 4):
        so, tio, histo, peo, ptho, indo = oracle_mod.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=5, want_hist=True, ind=ind0)
        for lanes in (1, 4):
            sd, indd = _t(st), _t(ind0)
            ticks, hist = closed_loop_prediction_lanes(sd, dc, goal, lanes, dim=dim, max_ticks=5, want_hist=True, ind=indd)
            assert np.array_equal(ticks.cpu().numpy(), tio) and np.array_equal(indd.cpu().numpy(), indo)
            assert np.array_equal(sd.cpu().numpy(), so, equal_nan=True) and np.array_equal(hist.cpu().numpy(), histo, equal_nan=True)
            assert np.isnan(so[2]).any() and np.isnan(so[5]).any() and not np.isnan(np.delete(so, [2, 5], axis=0)).any()


def test_lqr_closed_loop_host_pointer_abi(crx, oracle_mod, lqr_setup):
    from cpprobotics_amd import _lib as L
    course, goal, _ = lqr_setup
    n, max_ticks = 70, 400
    st = tracking_agents(n, tuple(c[:100] for c in course), 77, spread=0.3)
    so, tio, *_ = oracle_mod.lqr_closed_loop(st, course, goal, dim=5, max_ticks=max_ticks)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    hc = L.Course(len(course[0]), *(a.ctypes.data for a in course))
    lp = L.LoopParams(goal[0], goal[1], 0.3, 1.0, 0.05, max_ticks)
    s = st.copy(); ticks = np.zeros(n, np.int32)
    L.check(crx.lib().crx_lqr_closed_loop_batch(n, 5, vp(s), C.byref(hc), None, None, None, None, None, C.byref(lp), None, vp(ticks)), "loop")
    assert np.array_equal(ticks, tio) and bit_equal(s, so)


def test_calc_ref_trajectory_bit_exact(crx, oracle_mod):
    course, goal = mpc_course_f32()
    dc = crx.Course.from_numpy(course)
    n = 3000
    st = tracking_agents(n, course, 5, spread=1.0)
    st[:, 3] = np.random.default_rng(6).uniform(-3, 15, n).astype(np.float32)
    tind0 = np.maximum(0, np.random.default_rng(8).integers(0, len(course[0]) + 5, n) - 4).astype(np.int32)
    for T in (6, 21):
        xo, to = oracle_mod.calc_ref_trajectory(st, course, tind0, T)
        td = _t(tind0)
        xref = crx.calc_ref_trajectory(_t(st), dc, td, T)
        assert np.array_equal(td.cpu().numpy(), to) and bit_equal(xref.cpu().numpy(), xo)
    wo = oracle_mod.calc_nearest_index_window(st, course, tind0)
    assert np.array_equal(crx.calc_nearest_index_window(_t(st), dc, _t(tind0)).cpu().numpy(), wo)


@pytest.mark.parametrize("T,n", [(6, 48), (6, 130), (21, 70)])
def test_mpc_closed_loop_matches_oracle(crx, oracle_mod, T, n):
    """mpc_simulation's loop as ONE persistent kernel: the HIP solver and the oracle's twin agree to ~1e-15 per solve and the
    applied control is that solution rounded to float, so the loops stay together — north_star's 1e-6 (floor 1.0) is asserted, and
    measured (scripts/gpu_mpc_loop_err.py) every agent's trajectory equals the oracle's bit for bit; agents that start near the
    end of the course reach the goal and stop (same tick as the oracle's)."""
    course, goal = mpc_course_f32()
    dc = crx.Course.from_numpy(course)
    max_ticks = 40
    st = tracking_agents(n, tuple(c[:150] for c in course), 9, spread=0.5)
    st[:, 3] = np.random.default_rng(10).uniform(0.5, 4.0, n).astype(np.float32)
    st[0] = (course[0][0], course[1][0], course[2][0], course[4][0])          # the reference's start (:349)
    nc = len(course[0])
    for k in range(1, 9):                                                      # a few agents a short drive from the goal
        j = nc - 3 - 2 * k
        st[k] = (course[0][j], course[1][j], course[2][j], 2.5)
    tind0 = oracle_mod.calc_nearest_index(st, course)[0].astype(np.int32)
    so, tio, histo, tindo = oracle_mod.mpc_closed_loop(st, course, goal, T=T, max_ticks=max_ticks, target_ind=tind0, want_hist=True)
    sd, td = _t(st), _t(tind0)
    ticks, hist = crx.mpc_simulation(sd, dc, goal, T, max_ticks, target_ind=td, want_hist=True)
    assert np.array_equal(ticks.cpu().numpy(), tio) and (tio[1:9] < max_ticks).any()
    for a in range(n):                                                         # rows past an agent's last tick are not written
        assert floored_rel_err(hist.cpu().numpy()[: tio[a], a], histo[: tio[a], a], 1.0) <= 1e-6
    assert floored_rel_err(sd.cpu().numpy(), so, 1.0) <= 1e-6
    assert np.array_equal(td.cpu().numpy(), tindo)


def test_mpc_closed_loop_full_episode(crx, oracle_mod):
    """The reference's own run (src/model_predictive_control.cpp:349-385: start on the first course point, T = 6) driven to the
    goal, with neighbours: same tick counts, every tick of every trajectory within 1e-6 of the oracle's loop."""
    course, goal = mpc_course_f32()
    dc = crx.Course.from_numpy(course)
    n, T, max_ticks = 16, 6, 700
    st = tracking_agents(n, tuple(c[:60] for c in course), 29, spread=0.4)
    st[:, 3] = np.random.default_rng(30).uniform(0.5, 4.0, n).astype(np.float32)
    st[0] = (course[0][0], course[1][0], course[2][0], course[4][0])          # the reference's start (:349)
    tind0 = oracle_mod.calc_nearest_index(st, course)[0].astype(np.int32)
    tind0[0] = 0                                                               # :357
    so, tio, histo, tindo = oracle_mod.mpc_closed_loop(st, course, goal, T=T, max_ticks=max_ticks, target_ind=tind0, want_hist=True)
    sd, td = _t(st), _t(tind0)
    ticks, hist, flags = crx.mpc_simulation(sd, dc, goal, T, max_ticks, target_ind=td, want_hist=True, want_flags=True)
    assert np.array_equal(ticks.cpu().numpy(), tio) and tio[0] < max_ticks and (tio < max_ticks).mean() > 0.9
    h = hist.cpu().numpy()
    for a in range(n):
        assert floored_rel_err(h[: tio[a], a], histo[: tio[a], a], 1.0) <= 1e-6, a
    assert floored_rel_err(sd.cpu().numpy(), so, 1.0) <= 1e-6 and np.array_equal(td.cpu().numpy(), tindo)
    assert not (flags.cpu().numpy()[0] & 2)


def test_mpc_closed_loop_long_horizon_and_flags(crx, oracle_mod):
    """The MAXT = 64 instantiation of the persistent kernel (T = 30) against the oracle's loop, and the per-agent solve flags:
    clear on the course; bit 1 set for an agent that starts faster than MAX_SPEED (crx_mpc_solve's status bit 1)."""
    course, goal = mpc_course_f32()
    dc = crx.Course.from_numpy(course)
    n, T, max_ticks = 40, 30, 6
    st = tracking_agents(n, tuple(c[:150] for c in course), 19, spread=0.4)
    st[:, 3] = np.random.default_rng(20).uniform(0.5, 4.0, n).astype(np.float32)
    st[3, 3] = 25.0                                                            # above MAX_SPEED = 55 km/h
    tind0 = oracle_mod.calc_nearest_index(st, course)[0].astype(np.int32)
    so, tio, histo, tindo = oracle_mod.mpc_closed_loop(st, course, goal, T=T, max_ticks=max_ticks, target_ind=tind0, want_hist=True)
    sd, td = _t(st), _t(tind0)
    ticks, hist, flags = crx.mpc_simulation(sd, dc, goal, T, max_ticks, target_ind=td, want_hist=True, want_flags=True)
    flags = flags.cpu().numpy()
    assert np.array_equal(ticks.cpu().numpy(), tio) and np.array_equal(td.cpu().numpy(), tindo)
    assert floored_rel_err(sd.cpu().numpy(), so, 1.0) <= 1e-5
    assert flags[3] & 2 and not (np.delete(flags, 3) & 2).any()
    assert (flags & 1).mean() < 0.1


def test_tracking_fuzz_wide_ranges(crx, oracle_mod, lqr_setup):
    """Control evaluation and update with states spread over many decades (vehicles far off the course, huge/tiny speeds,
    NaN positions): wherever the oracle's result is finite the engine's is bit-identical; NaN positions keep the incoming
    index, as the reference's reference parameter does."""
    course, goal, dc = lqr_setup
    rng = np.random.default_rng(99)
    n = 3000
    def wide(shape, lo, hi):
        return (rng.choice([-1.0, 1.0], shape) * np.exp(rng.uniform(lo, hi, shape) * np.log(10.0))).astype(np.float32)
    st = np.stack([wide(n, -6, 6), wide(n, -6, 6), wide(n, -8, 4), wide(n, -12, 6)], axis=1)
    st[::97, 0] = np.nan
    pe, pth = wide(n, -10, 3), wide(n, -10, 3)
    ind0 = rng.integers(0, len(course[0]), n).astype(np.int32)
    for dim in (5, 4):
        with np.errstate(all="ignore"):
            co, io, peo, ptho = oracle_mod.lqr_steering_control(st, course, pe, pth, dim=dim, ind=ind0)
        ped, pthd, indd = _t(pe), _t(pth), _t(ind0)
        ctl, ind = crx.lqr_steering_control(_t(st), dc, ped, pthd, dim=dim, ind=indd)
        assert np.array_equal(ind.cpu().numpy(), io)
        c = ctl.cpu().numpy().reshape(n, -1); cr = co.reshape(n, -1)
        fin = np.isfinite(cr).all(axis=1) & np.isfinite(peo) & np.isfinite(ptho)
        assert fin.mean() > 0.5
        assert np.array_equal(c[fin], cr[fin]) and np.array_equal(ped.cpu().numpy()[fin], peo[fin]) and np.array_equal(pthd.cpu().numpy()[fin], ptho[fin])
    a, d = wide(n, -8, 3), wide(n, -8, 2)
    for mpc in (False, True):
        dt, wb = (0.2, 2.5) if mpc else (0.1, 0.5)
        with np.errstate(all="ignore"):
            ref = oracle_mod.update(st, a, d, dt=dt, wheelbase=wb, clamp_speed=mpc)
        sd = _t(st)
        crx.update(sd, _t(a), _t(d), crx.vehicle_params(mpc))
        got = sd.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])

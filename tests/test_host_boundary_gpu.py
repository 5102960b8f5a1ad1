"""The host-pointer side of the C ABI on the GPU (csrc/crx_host.h): zero-copy small calls, workspace + staged copies, the
three-stream EKF pipeline over time chunks, pinned caller memory, and the device set that shards a batch — forced here onto ONE
GPU named several times, which exercises exactly the code an 8-GPU host runs.  Every result must equal the `_dev` path's (one
launch on device-resident tensors) bit for bit: the boundary moves bytes, it does not compute."""
import numpy as np
import pytest

from common import bit_equal, ekf_QR, ekf_agents, ekf_noise, lqr_course, lqr_speeds, mpc_course_f32, mpc_problem, tracking_agents

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ekf_inputs(oracle_mod, n, T, seed):
    u, x0, P0 = ekf_agents(n, seed)
    w = ekf_noise(T, n, seed + 1)
    z, ud, *_ = oracle_mod.ekf_simulate_inputs(u, x0, x0, w)
    return x0, P0, z, ud


def _ekf_dev(crx, x0, P0, z, ud, want_P=False):
    import torch
    Q, R = ekf_QR()
    T, n = z.shape[0], z.shape[1]
    xd, Pd = _t(x0), _t(P0)
    xh = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
    Ph = torch.empty((T, n, 16), dtype=torch.float32, device="cuda") if want_P else None
    crx.ekf_run(xd, Pd, _t(z), _t(ud), Q, R, x_hist=xh, P_hist=Ph)
    return xd.cpu().numpy(), Pd.cpu().numpy(), xh.cpu().numpy(), (Ph.cpu().numpy() if want_P else None)


@pytest.mark.parametrize("n,T,want_P", [(1, 1, False), (1, 30, True), (7, 5, True), (300, 40, True),      # zero-copy (<= 256 KB) ...
                                        (5000, 40, True), (20000, 64, False),                             # ... workspace, one chunk / a few
                                        (65536, 50, False), (65537, 37, True), (300000, 3, False)])        # ... many chunks, ragged tails
def test_ekf_run_host_equals_device_path(crx, oracle_mod, n, T, want_P):
    Q, R = ekf_QR()
    x0, P0, z, ud = _ekf_inputs(oracle_mod, n, T, seed=n + T)
    xr, Pr, hr, Phr = _ekf_dev(crx, x0, P0, z, ud, want_P)
    x, P, h, Ph = crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R, want_hist=True, want_P_hist=want_P)
    assert bit_equal(x, xr) and bit_equal(P, Pr) and bit_equal(h, hr)
    if want_P:
        assert bit_equal(Ph, Phr)
    if n * T <= 20000:                                   # and the oracle, where it is quick
        xo, Po, ho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
        assert bit_equal(x, xo) and bit_equal(P, Po) and bit_equal(h, ho)
    # no history asked for: the final state alone (T = 1 takes the single-step kernel)
    x2, P2, _, _ = crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R)
    assert bit_equal(x2, xr) and bit_equal(P2, Pr)


def test_ekf_run_host_random_shapes(crx, oracle_mod):
    """Thirty random (n, T, histories, device-set) combinations around the boundary's thresholds — the 256 KB zero-copy limit, one
    chunk / several chunks, ragged last chunk, odd n (unaligned rows inside the rings), shards of a forced split: always the device
    path's bits."""
    rng = np.random.default_rng(2024)
    Q, R = ekf_QR()
    for case in range(30):
        kind = case % 3
        if kind == 0:      # around the zero-copy threshold: n * T * (32 or 96) + 80 n ~ 256 KB
            T = int(rng.integers(1, 40)); want_P = bool(rng.integers(0, 2))
            n = max(1, int(262144 / (T * (96 if want_P else 32) + 80)) + int(rng.integers(-3, 4)))
        elif kind == 1:    # a few chunks with a ragged tail: chunk = 16 MB / (n * 16 or 80) steps
            n = int(rng.integers(20000, 90000)) | 1; want_P = bool(rng.integers(0, 4) == 0)
            tc = max(1, (16 << 20) // (n * (80 if want_P else 16)))
            T = int(tc * rng.integers(1, 4) + rng.integers(0, tc + 1)); T = max(1, min(T, 60))
        else:              # small and odd
            n, T, want_P = int(rng.integers(1, 3000)), int(rng.integers(1, 25)), bool(rng.integers(0, 2))
        split = int(rng.integers(0, 4))
        x0, P0, z, ud = _ekf_inputs(oracle_mod, n, T, seed=1000 + case)
        xr, Pr, hr, Phr = _ekf_dev(crx, x0, P0, z, ud, want_P)
        crx.host.set_devices([0] * split if split else None, min_agents_per_device=1)
        try:
            x, P, h, Ph = crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R, want_hist=True, want_P_hist=want_P)
        finally:
            crx.host.set_devices(None)
        assert bit_equal(x, xr) and bit_equal(P, Pr) and bit_equal(h, hr) and (not want_P or bit_equal(Ph, Phr)), (case, n, T, want_P, split)


def test_ekf_run_host_pinned_arrays(crx, oracle_mod):
    """Caller memory from crx_host_alloc: DMA'd in place (strided 2-D copies straight between the caller's arrays and the rings)."""
    Q, R = ekf_QR()
    n, T = 40000, 48
    x0, P0, z, ud = _ekf_inputs(oracle_mod, n, T, seed=3)
    xr, Pr, hr, _ = _ekf_dev(crx, x0, P0, z, ud)
    pz, pu, ph = crx.host.PinnedArray((T, n, 2)), crx.host.PinnedArray((T, n, 2)), crx.host.PinnedArray((T, n, 4))
    pz.array[...] = z; pu.array[...] = ud; ph.array[...] = -1.0
    x, P, h, _ = crx.host.ekf_run(x0.copy(), P0.copy(), pz.array, pu.array, Q, R, x_hist=ph.array)
    assert bit_equal(x, xr) and bit_equal(P, Pr) and bit_equal(ph.array, hr)
    # mixed: pinned inputs, pageable history
    x, P, h, _ = crx.host.ekf_run(x0.copy(), P0.copy(), pz.array, pu.array, Q, R, want_hist=True)
    assert bit_equal(h, hr)
    for a in (pz, pu, ph):
        a.free()


def test_workspace_is_reused_and_released(crx, oracle_mod):
    Q, R = ekf_QR()
    x0, P0, z, ud = _ekf_inputs(oracle_mod, 30000, 20, seed=9)
    a = crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R, want_hist=True)
    b = crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R, want_hist=True)            # second call: no allocation
    crx.host.release_workspace()
    crx.host.reserve_workspace(64 << 20, 64 << 20)                                      # ahead of time this time
    c = crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R, want_hist=True)
    for r in (b, c):
        assert bit_equal(r[0], a[0]) and bit_equal(r[1], a[1]) and bit_equal(r[2], a[2])
    crx.host.release_workspace()


@pytest.fixture()
def split3(crx):
    """Three shards on device 0, whatever the batch size: the sharding code of crx_set_devices on one GPU."""
    crx.host.set_devices([0, 0, 0], min_agents_per_device=1)
    assert crx.host.get_devices() == [0, 0, 0]
    yield
    crx.host.set_devices(None)
    assert crx.host.get_devices() == []


def test_sharded_ekf_equals_unsharded(crx, oracle_mod, split3):
    Q, R = ekf_QR()
    for n, T, want_P in ((2, 4, True), (1000, 33, True), (70001, 40, False)):            # ragged shards; zero-copy and pipeline shards
        x0, P0, z, ud = _ekf_inputs(oracle_mod, n, T, seed=n)
        crx.host.set_devices(None)
        xr, Pr, hr, Phr = crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R, want_hist=True, want_P_hist=want_P)
        crx.host.set_devices([0, 0, 0], min_agents_per_device=1)
        x, P, h, Ph = crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R, want_hist=True, want_P_hist=want_P)
        assert bit_equal(x, xr) and bit_equal(P, Pr) and bit_equal(h, hr) and (not want_P or bit_equal(Ph, Phr))
    xd, Pd, hd, _ = _ekf_dev(crx, x0, P0, z, ud)
    assert bit_equal(x, xd) and bit_equal(h, hd)


def test_sharded_solves_equal_unsharded(crx, oracle_mod, split3):
    n = 5003
    v = lqr_speeds(n, seed=21)
    A, B, Q, R = oracle_mod.lqr_build(v, 5)
    x0, xref = mpc_problem(700, 6, seed=2)
    sharded = (crx.host.dare_from_v(v, 5), crx.host.dare_from_v(v, 4), crx.host.dare(A, B, Q, R), crx.host.mpc_solve(x0, xref, 6),
               crx.host.mpc_solve(x0, xref, 6, portfolio=True))
    crx.host.set_devices(None)
    single = (crx.host.dare_from_v(v, 5), crx.host.dare_from_v(v, 4), crx.host.dare(A, B, Q, R), crx.host.mpc_solve(x0, xref, 6),
              crx.host.mpc_solve(x0, xref, 6, portfolio=True))
    for a, b in zip(sharded, single):
        for s, t in zip(a, b):
            assert np.array_equal(s.view(np.uint8), t.view(np.uint8))
    Xo, Ko, ito = oracle_mod.dare(A, B, Q, R)
    assert bit_equal(sharded[0][0], Xo) and bit_equal(sharded[2][1], Ko) and np.array_equal(sharded[2][2], ito)


def test_sharded_closed_loops_equal_unsharded(crx, oracle_mod, split3):
    course, goal = lqr_course()
    st = tracking_agents(301, tuple(c[:150] for c in course), 5, spread=0.3)
    mcourse, mgoal = mpc_course_f32()
    mst = tracking_agents(50, tuple(c[:150] for c in mcourse), 8, spread=0.2)
    sharded = (crx.host.lqr_closed_loop(st, course, goal, dim=5, max_ticks=300, want_hist=True),
               crx.host.mpc_closed_loop(mst, mcourse, mgoal, T=6, max_ticks=40, want_hist=True))
    crx.host.set_devices(None)
    single = (crx.host.lqr_closed_loop(st, course, goal, dim=5, max_ticks=300, want_hist=True),
              crx.host.mpc_closed_loop(mst, mcourse, mgoal, T=6, max_ticks=40, want_hist=True))
    for a, b in zip(sharded, single):
        for s, t in zip(a, b):
            assert np.array_equal(s.view(np.uint8), t.view(np.uint8))
    so, tio, histo, *_ = oracle_mod.lqr_closed_loop(st, course, goal, dim=5, max_ticks=300, want_hist=True)
    assert bit_equal(sharded[0][0], so) and np.array_equal(sharded[0][1], tio) and bit_equal(sharded[0][2], histo)


def test_concurrent_host_calls(crx, oracle_mod):
    """Host threads calling host-pointer entry points at once (ctypes drops the GIL): the per-device context is locked for the
    duration of a call, the shared workspaces are never used by two calls at a time, every result is the serial one."""
    from concurrent.futures import ThreadPoolExecutor
    Q, R = ekf_QR()
    jobs = []
    for k, (n, T) in enumerate([(30000, 24), (7, 9), (50000, 16), (1, 1), (20000, 40), (300, 30)]):
        x0, P0, z, ud = _ekf_inputs(oracle_mod, n, T, seed=50 + k)
        jobs.append((x0, P0, z, ud, crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R, want_hist=True)))
    v = lqr_speeds(9000, seed=4)
    dref = crx.host.dare_from_v(v, 5)

    def ekf(j):
        x0, P0, z, ud, ref = jobs[j % len(jobs)]
        got = crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R, want_hist=True)
        return all(bit_equal(a, b) for a, b in zip(got[:3], ref[:3]))

    def dare(_):
        got = crx.host.dare_from_v(v, 5)
        return all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(got, dref))
    with ThreadPoolExecutor(6) as ex:
        res = list(ex.map(lambda j: ekf(j) if j % 3 else dare(j), range(36)))
    assert all(res)


def test_device_selection(crx):
    import torch
    assert crx.host.get_device() == torch.cuda.current_device()
    crx.host.set_device(0)
    l = crx.lib()
    assert l.crx_set_device(torch.cuda.device_count()) == -1 and l.crx_set_device(-1) == -1
    bad = (__import__("ctypes").c_int * 2)(0, torch.cuda.device_count())
    assert l.crx_set_devices(bad, 2, 1) == -1 and crx.host.get_devices() == []


def test_single_call_latency_report(crx, oracle_mod):
    """The literal drop-in calls (n = 1) through the host boundary: bits first, then the latency, printed for the record."""
    import time
    Q, R = ekf_QR()
    x0, P0, z, ud = _ekf_inputs(oracle_mod, 1, 1, seed=1)
    xo, Po, _, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
    x, P, _, _ = crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R)
    assert bit_equal(x, xo) and bit_equal(P, Po)
    t0 = time.perf_counter()
    for _ in range(200):
        crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R)
    print(f"ekf_estimation (n = 1) through crx_ekf_run_batch: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call (Python overhead included)")


def test_several_large_pageable_inputs_in_one_call(crx, oracle_mod):
    """a call with more than one pageable input of >= 1 MB stages them through the same two pinned slots; the fence on
    slot reuse used to restart with every argument, so the second input could overwrite a slot whose H2D was still in flight.  Large
    pageable numpy arrays through the host-pointer entry points must give the `_dev` path's bits: mpc_solve (x0 1.6 MB, xref 9.6 MB at
    T = 6), solve_DARE (A 7 MB, B 2.8 MB, Q 7 MB, R 1.1 MB), and the same three times over (a race is a matter of timing)."""
    import torch
    n = 100000
    x0, xref = mpc_problem(n, 6, seed=31)
    sd, std, cd = crx.mpc_solve(torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda(), 6, return_status=True)
    sd, std, cd = sd.cpu().numpy(), std.cpu().numpy(), cd.cpu().numpy()
    for _ in range(3):
        s, st, c = crx.host.mpc_solve(x0, xref, 6)
        assert np.array_equal(st, std) and bit_equal(s, sd) and np.array_equal(c.view(np.int64), cd.view(np.int64))
    m = 70000
    v = lqr_speeds(m, seed=33)
    A, B, Q, R = oracle_mod.lqr_build(v, 5)
    Xd, itd = crx.solve_DARE(*(torch.from_numpy(a).cuda() for a in (A, B, Q, R)))
    Kd = crx.dlqr(*(torch.from_numpy(a).cuda() for a in (A, B, Q, R)))
    for _ in range(3):
        X, K, it = crx.host.dare(A, B, Q, R)
        assert bit_equal(X, Xd.cpu().numpy()) and np.array_equal(it, itd.cpu().numpy()) and bit_equal(K, Kd.cpu().numpy())

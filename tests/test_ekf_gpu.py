"""GPU parity: the HIP EKF path (through the C ABI) against the CPU oracle — bit-exact."""
import numpy as np
import pytest

from common import bit_equal, ekf_QR, ekf_agents, ekf_noise, floored_rel_err

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _inputs(oracle, n, T, seed, single=False):
    u, x0, P0 = ekf_agents(n, seed, single_vehicle=single)
    w = ekf_noise(T, n, seed + 1000)
    z, ud, _, _, _, _ = oracle.ekf_simulate_inputs(u, np.zeros_like(x0) if single else x0, x0, w)
    return u, x0, P0, w, z, ud


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000])
def test_small_functions_bit_exact(crx, oracle_mod, n):
    rng = np.random.default_rng(n)
    x = rng.uniform(-30, 30, (n, 4)).astype(np.float32)
    u = rng.uniform(-2, 2, (n, 2)).astype(np.float32)
    assert bit_equal(crx.motion_model(_t(x), _t(u)).cpu().numpy(), oracle_mod.motion_model(x, u))
    assert bit_equal(crx.jacobF(_t(x), _t(u)).cpu().numpy(), oracle_mod.jacobF(x, u))
    assert bit_equal(crx.observation_model(_t(x)).cpu().numpy(), oracle_mod.observation_model(x))
    assert bit_equal(crx.jacobH(), oracle_mod.jacobH())


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 257, 4096])
def test_ekf_single_step_bit_exact(crx, oracle_mod, n):
    Q, R = ekf_QR()
    u, x0, P0, w, z, ud = _inputs(oracle_mod, n, 3, seed=n)
    x, P = x0.copy(), P0.copy()
    xd, Pd = _t(x0), _t(P0)
    for t in range(3):
        x, P = oracle_mod.ekf_step(x, P, z[t], ud[t], Q, R)
        crx.ekf_estimation(xd, Pd, _t(z[t]), _t(ud[t]), Q, R)
        assert bit_equal(xd.cpu().numpy(), x), f"x differs at step {t}"
        assert bit_equal(Pd.cpu().numpy(), P), f"P differs at step {t}"


def test_ekf_single_step_outside_the_fast_domain(crx, oracle_mod):
    """The single-step kernel runs the packed fast step (round 4) and sends a wave to the general step when one of its lanes has
    yaw = +-0, |yaw| >= 120, a non-finite state or an extreme determinant: wide-range fuzz in which waves of every mix occur, a
    ragged last wave included.  Vehicles whose reference step is finite must agree bit for bit."""
    rng = np.random.default_rng(77)
    n = 64 * 11 + 5
    def wide(shape, lo, hi):
        return (rng.choice([-1.0, 1.0], shape) * np.exp(rng.uniform(lo, hi, shape) * np.log(10.0))).astype(np.float32)
    x0 = wide((n, 4), -20, 8)
    x0[:, 2] = wide(n, -45, 3)                       # yaw from denormal to 1000 rad
    x0[: 64 * 4, 2] = rng.uniform(-3, 3, 64 * 4).astype(np.float32)          # four waves wholly inside the domain ...
    x0[64 * 2 + 5, 2] = 0.0; x0[64 * 3 + 9, 2] = 121.0                       # ... two of them with a single lane outside
    A = rng.standard_normal((n, 4, 4)).astype(np.float32)
    P0 = (np.einsum("nij,nkj->nik", A, A) * wide((n, 1, 1), -30, 30)).astype(np.float32).reshape(n, 16)
    P0[: 64 * 4] = (np.einsum("nij,nkj->nik", A[:256], A[:256]) + np.eye(4, dtype=np.float32)).reshape(256, 16)
    z, ud = wide((n, 2), -10, 6), wide((n, 2), -10, 2)
    z[: 64 * 4], ud[: 64 * 4] = rng.uniform(-5, 5, (256, 2)), rng.uniform(-1, 1, (256, 2))
    x0[300, 2] = np.inf; x0[370, 0] = np.nan; P0[430, 5] = np.inf; z[500, 0] = np.nan; ud[600, 1] = np.inf
    for qs, rs in ((1.0, 1.0), (1e-10, 1e-8), (1e18, 1e20)):
        Q0, R0 = ekf_QR()
        Q, R = (Q0 * np.float32(qs)).astype(np.float32), (R0 * np.float32(rs)).astype(np.float32)
        with np.errstate(all="ignore"):
            xo, Po = oracle_mod.ekf_step(x0, P0, z, ud, Q, R)
        xd, Pd = _t(x0), _t(P0)
        crx.ekf_estimation(xd, Pd, _t(z), _t(ud), Q, R)
        ok = np.isfinite(xo).all(axis=1) & np.isfinite(Po).all(axis=1)
        assert ok.sum() > 300 and ok[:256].sum() >= 250
        # IEEE equality (skipping the products by literal 0 / 1 entries can flip the sign of an exact zero, nothing else)
        assert bit_equal(xd.cpu().numpy()[ok], xo[ok]) and bit_equal(Pd.cpu().numpy()[ok], Po[ok])
        assert not (np.isfinite(xd.cpu().numpy()[~ok]).all(axis=1) & np.isfinite(Pd.cpu().numpy()[~ok]).all(axis=1)).any()


def test_ekf_single_step_streaming_variant_equals_the_plain_one(crx):
    """From 3 M vehicles on the covariance rows go past the caches (nontemporal loads / stores, ekf_step_kernel<true>): the same
    arithmetic — the first vehicles of a 3 M + 70 batch end with the bits a small batch (plain accesses) gives them, three steps."""
    import torch
    Q, R = ekf_QR()
    n, m = (3 << 20) + 70, 64 * 40 + 70
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = torch.zeros((n, 4), dtype=torch.float32, device="cuda"); x[:, 2] = torch.rand(n, device="cuda", generator=g) * 6 - 3
    A = torch.randn((n, 4, 4), device="cuda", generator=g)
    P = (A @ A.transpose(1, 2) + torch.eye(4, device="cuda")).reshape(n, 16).contiguous()
    del A
    z = torch.rand((3, n, 2), device="cuda", generator=g) * 4; u = torch.rand((3, n, 2), device="cuda", generator=g)
    tail = slice(n - m, n)                                   # the ragged last wave is in here
    xs, Ps = x[tail].clone(), P[tail].clone()
    xh, Ph = x[:m].clone(), P[:m].clone()
    for t in range(3):
        crx.ekf_estimation(x, P, z[t], u[t], Q, R)
        crx.ekf_estimation(xs, Ps, z[t, tail].contiguous(), u[t, tail].contiguous(), Q, R)
        crx.ekf_estimation(xh, Ph, z[t, :m].contiguous(), u[t, :m].contiguous(), Q, R)
    assert torch.equal(x[:m].view(torch.int32), xh.view(torch.int32)) and torch.equal(P[:m].view(torch.int32), Ph.view(torch.int32))
    # the tail batch starts at a vehicle index that is not a multiple of 64, so wave membership differs there: the fast path and its
    # general-step fallback agree bit for bit, hence so do the two runs
    assert torch.equal(x[tail].view(torch.int32), xs.view(torch.int32)) and torch.equal(P[tail].view(torch.int32), Ps.view(torch.int32))
    assert torch.isfinite(x).all() and torch.isfinite(P).all()


@pytest.mark.parametrize("n,T", [(1, 1000), (64, 1), (65, 7), (100, 8), (257, 9), (1024, 200), (300, 17)])
def test_ekf_fused_run_bit_exact(crx, oracle_mod, n, T):
    import torch
    Q, R = ekf_QR()
    u, x0, P0, w, z, ud = _inputs(oracle_mod, n, T, seed=n + T, single=(n == 1))
    xo, Po, xho, pho = oracle_mod.ekf_run(x0, P0, z, ud, Q, R, want_phist=True)
    xd, Pd = _t(x0), _t(P0)
    xh = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
    ph = torch.empty((T, n, 16), dtype=torch.float32, device="cuda")
    crx.ekf_run(xd, Pd, _t(z), _t(ud), Q, R, x_hist=xh, P_hist=ph)
    assert bit_equal(xh.cpu().numpy(), xho)
    assert bit_equal(ph.cpu().numpy(), pho)
    assert bit_equal(xd.cpu().numpy(), xo) and bit_equal(Pd.cpu().numpy(), Po)
    # the other three template instantiations (no history / x only / P only) give the same state
    for want_x, want_p in [(False, False), (True, False), (False, True)]:
        xd2, Pd2 = _t(x0), _t(P0)
        crx.ekf_run(xd2, Pd2, _t(z), _t(ud), Q, R,
                    x_hist=torch.empty_like(xh) if want_x else None, P_hist=torch.empty_like(ph) if want_p else None)
        assert bit_equal(xd2.cpu().numpy(), xo) and bit_equal(Pd2.cpu().numpy(), Po)


@pytest.mark.parametrize("T", [256, 257, 263, 264, 265, 271, 272])
def test_ekf_fused_run_long_chunks_every_remainder(crx, oracle_mod, T):
    """Launches of >= 256 steps without the covariance history run eight steps per chunk (csrc/api_ekf.inl); the last full chunk keeps
    the fast step with a guarded refill and fewer than eight steps are left to the general tail (round 5): every remainder T mod 8,
    histories and final state bit for bit against the oracle, with and without the x history."""
    import torch
    n = 130
    Q, R = ekf_QR()
    u, x0, P0, w, z, ud = _inputs(oracle_mod, n, T, seed=9000 + T)
    xo, Po, xho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
    for want_x in (True, False):
        xd, Pd = _t(x0), _t(P0)
        xh = torch.empty((T, n, 4), dtype=torch.float32, device="cuda") if want_x else None
        crx.ekf_run(xd, Pd, _t(z), _t(ud), Q, R, x_hist=xh)
        if want_x:
            assert bit_equal(xh.cpu().numpy(), xho)
        assert bit_equal(xd.cpu().numpy(), xo) and bit_equal(Pd.cpu().numpy(), Po)


def test_ekf_simulate_inputs_bit_exact(crx, oracle_mod):
    import torch
    n, T = 333, 50
    u, x0, P0 = ekf_agents(n, 5)
    w = ekf_noise(T, n, 6)
    zo, udo, xto, xdo, xth, xdh = oracle_mod.ekf_simulate_inputs(u, x0, x0, w, want_hist=True)
    xt, xd = _t(x0), _t(x0)
    h1 = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
    h2 = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
    z, ud = crx.ekf_simulate_inputs(_t(u), xt, xd, _t(w), xTrue_hist=h1, xDR_hist=h2)
    assert bit_equal(z.cpu().numpy(), zo) and bit_equal(ud.cpu().numpy(), udo)
    assert bit_equal(xt.cpu().numpy(), xto) and bit_equal(xd.cpu().numpy(), xdo)
    assert bit_equal(h1.cpu().numpy(), xth) and bit_equal(h2.cpu().numpy(), xdh)


def test_ekf_empty_and_errors(crx):
    import torch
    Q, R = ekf_QR()
    e4 = torch.empty((0, 4), dtype=torch.float32, device="cuda")
    e16 = torch.empty((0, 16), dtype=torch.float32, device="cuda")
    e2 = torch.empty((0, 2), dtype=torch.float32, device="cuda")
    crx.ekf_estimation(e4, e16, e2, e2, Q, R)                                  # n = 0 is a no-op
    crx.ekf_run(e4, e16, torch.empty((5, 0, 2), device="cuda"), torch.empty((5, 0, 2), device="cuda"), Q, R)
    x = torch.zeros((4, 4), device="cuda")
    P = torch.zeros((4, 16), device="cuda")
    crx.ekf_run(x, P, torch.empty((0, 4, 2), device="cuda"), torch.empty((0, 4, 2), device="cuda"), Q, R)  # T = 0
    assert float(x.abs().sum()) == 0.0
    with pytest.raises(crx.CrxError):
        crx.ekf_estimation(x.cpu(), P, e2, e2, Q, R)                          # host tensor -> loud failure


def test_ekf_host_pointer_abi(crx, oracle_mod):
    """The non-_dev entry points (host pointers in, host pointers out)."""
    import ctypes as C
    from cpprobotics_amd import _lib as L
    Q, R = ekf_QR()
    n, T = 130, 12
    u, x0, P0, w, z, ud = _inputs(oracle_mod, n, T, seed=77)
    xo, Po, xho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
    x, P = x0.copy(), P0.copy()
    xh = np.empty((T, n, 4), dtype=np.float32)
    p = L.EkfParams(); p.dt = 0.1
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    L.check(crx.lib().crx_ekf_run_batch(n, T, vp(x), vp(P), vp(z), vp(ud), vp(xh), None, vp(Q), vp(R), C.byref(p)), "run")
    assert bit_equal(x, xo) and bit_equal(P, Po) and bit_equal(xh, xho)
    x1, P1 = x0.copy(), P0.copy()
    L.check(crx.lib().crx_ekf_step_batch(n, vp(x1), vp(P1), vp(z[0]), vp(ud[0]), vp(Q), vp(R), C.byref(p)), "step")
    xs, Ps = oracle_mod.ekf_step(x0, P0, z[0], ud[0], Q, R)
    assert bit_equal(x1, xs) and bit_equal(P1, Ps)


def test_ekf_full_size_properties(crx, oracle_mod):
    """BASELINE config 2 (65,536 vehicles x 1000 steps): size-independent properties plus a
    bit-exact comparison of a strided sample of vehicles against the oracle."""
    import torch
    n, T = 65536, 1000
    Q, R = ekf_QR()
    u, x0, P0 = ekf_agents(n, 2024)
    g = torch.Generator(device="cuda"); g.manual_seed(99)
    w = torch.randn((T, n, 4), generator=g, device="cuda", dtype=torch.float32)
    xt, xdr = _t(x0), _t(x0)
    z, ud = crx.ekf_simulate_inputs(_t(u), xt, xdr, w)
    del w
    xd, Pd = _t(x0), _t(P0)
    xh = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
    crx.ekf_run(xd, Pd, z, ud, Q, R, x_hist=xh)
    # (1) composition: one 1000-step launch == 400-step launch followed by a 600-step launch
    xa, Pa = _t(x0), _t(P0)
    crx.ekf_run(xa, Pa, z[:400].contiguous(), ud[:400].contiguous(), Q, R)
    assert torch.equal(xa, xh[399])
    crx.ekf_run(xa, Pa, z[400:].contiguous(), ud[400:].contiguous(), Q, R)
    assert torch.equal(xa, xd) and torch.equal(Pa, Pd)
    # (2) fused == T single-step launches (first 25 steps)
    xs, Ps = _t(x0), _t(P0)
    for t in range(25):
        crx.ekf_estimation(xs, Ps, z[t], ud[t], Q, R)
        assert torch.equal(xs, xh[t])
    # (3) covariance stays symmetric to rounding and positive on the diagonal; everything finite
    Pm = Pd.view(n, 4, 4)
    assert torch.isfinite(xh).all() and torch.isfinite(Pd).all()
    assert float((Pm - Pm.transpose(1, 2)).abs().max()) < 1e-5 * float(Pm.abs().max())
    assert float(torch.diagonal(Pm, dim1=1, dim2=2).min()) > 0
    # (4) the filter tracks: position estimate closer to truth than dead reckoning, on average
    err_est = (xd[:, :2] - xt[:, :2]).norm(dim=1).mean()
    err_dr = (xdr[:, :2] - xt[:, :2]).norm(dim=1).mean()
    assert float(err_est) < float(err_dr)
    # (5) bit-exact against the oracle on every 512th vehicle (128 vehicles x 1000 steps)
    idx = np.arange(0, n, 512)
    zs = z[:, idx].contiguous().cpu().numpy(); us = ud[:, idx].contiguous().cpu().numpy()
    xo, Po, xho, _ = oracle_mod.ekf_run(x0[idx], P0[idx], zs, us, Q, R)
    assert bit_equal(xh[:, idx].cpu().numpy(), xho)
    assert bit_equal(Pd[idx].cpu().numpy(), Po)
    assert floored_rel_err(xd[idx].cpu().numpy(), xo, 1.0) <= 1e-6   # the stated tolerance, trivially


# ---- the fused kernel's fast-domain exits (packed step -> general step for the whole wave) ----------
def _run_both(crx, oracle_mod, x0, P0, z, ud, Q, R):
    import torch
    T, n = z.shape[0], x0.shape[0]
    xo, Po, xho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
    xd, Pd = _t(x0), _t(P0)
    xh = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
    crx.ekf_run(xd, Pd, _t(z), _t(ud), Q, R, x_hist=xh)
    return (xd.cpu().numpy(), Pd.cpu().numpy(), xh.cpu().numpy()), (xo, Po, xho)


@pytest.mark.parametrize("yaw", [119.99999, 120.0, 121.0, -500.0, 1.0e6, -3.0e9, 1.0e30])
def test_ekf_fused_large_yaw_bit_exact(crx, oracle_mod, yaw):
    """|yaw| >= 120 leaves the fast sincos path (192-bit 2/pi reduction needed): some waves of the launch
    redo their chunk with the general step, others stay on the packed path — all must match the oracle."""
    Q, R = ekf_QR()
    n, T = 64 * 5 + 7, 37
    u, x0, P0 = ekf_agents(n, 21)
    x0[64:128:3, 2] = np.float32(yaw)      # one wave partly outside the domain
    x0[200, 2] = np.float32(-yaw)          # one lane of another wave
    w = ekf_noise(T, n, 22)
    z, ud, _, _, _, _ = oracle_mod.ekf_simulate_inputs(u, x0, x0, w)
    (x, P, xh), (xo, Po, xho) = _run_both(crx, oracle_mod, x0, P0, z, ud, Q, R)
    assert bit_equal(xh, xho) and bit_equal(x, xo) and bit_equal(P, Po)


@pytest.mark.parametrize("scale", [1e-12, 1e12])
def test_ekf_fused_extreme_determinant_bit_exact(crx, oracle_mod, scale):
    """det(S) outside [2^-60, 2^60]: the un-scaled Newton reciprocal is not used (v_div_scale would scale)."""
    Q, R = ekf_QR()
    Qs, Rs = (Q * np.float32(scale)).astype(np.float32), (R * np.float32(scale)).astype(np.float32)
    n, T = 130, 23
    u, x0, P0 = ekf_agents(n, 23)
    P0 = (P0 * np.float32(scale)).astype(np.float32)
    w = ekf_noise(T, n, 24)
    z, ud, _, _, _, _ = oracle_mod.ekf_simulate_inputs(u, x0, x0, w)
    (x, P, xh), (xo, Po, xho) = _run_both(crx, oracle_mod, x0, P0, z, ud, Qs, Rs)
    assert np.isfinite(xho).all()
    assert bit_equal(xh, xho) and bit_equal(x, xo) and bit_equal(P, Po)


def test_ekf_fused_moderate_determinants_bit_exact(crx, oracle_mod):
    """Determinants spread over 2^-55 .. 2^55 stay on the fast reciprocal: it must round like IEEE division."""
    Q, R = ekf_QR()
    n, T = 4096, 9
    u, x0, P0 = ekf_agents(n, 25)
    rng = np.random.default_rng(26)
    sc = np.exp2(rng.uniform(-27, 27, n)).astype(np.float32)          # det ~ sc^2
    P0 = (P0 * sc[:, None]).astype(np.float32)
    w = ekf_noise(T, n, 27)
    z, ud, _, _, _, _ = oracle_mod.ekf_simulate_inputs(u, x0, x0, w)
    Rz = np.zeros(4, dtype=np.float32)                                 # S = H*PPred*H^T alone: det follows P0's scale
    (x, P, xh), (xo, Po, xho) = _run_both(crx, oracle_mod, x0, P0, z, ud, Q, Rz)
    assert bit_equal(xh, xho) and bit_equal(x, xo) and bit_equal(P, Po)


def test_ekf_fused_nonfinite_state(crx, oracle_mod):
    Q, R = ekf_QR()
    n, T = 70, 11
    u, x0, P0 = ekf_agents(n, 28)
    x0[1, 2] = np.inf
    x0[65, 2] = np.nan
    w = ekf_noise(T, n, 29)
    z, ud, _, _, _, _ = oracle_mod.ekf_simulate_inputs(u, np.zeros_like(x0), np.zeros_like(x0), w)
    (x, P, xh), (xo, Po, xho) = _run_both(crx, oracle_mod, x0, P0, z, ud, Q, R)
    ok = np.ones(n, dtype=bool); ok[[1, 65]] = False
    assert bit_equal(xh[:, ok], xho[:, ok]) and bit_equal(P[ok], Po[ok])
    assert np.isnan(xh[:, ~ok, :3]).all() and np.isnan(xho[:, ~ok, :3]).all()


def test_ekf_fused_64bit_address_kernels(crx, oracle_mod):
    """Batches above 4 M vehicles use the fused kernel's 64-bit-address instantiations; forced here on a small input through the
    experimental entry point."""
    import torch
    from cpprobotics_amd.experimental import ekf_run_addr64
    Q, R = ekf_QR()
    n, T = 333, 41
    u, x0, P0, w, z, ud = _inputs(oracle_mod, n, T, seed=91)
    x0[5, 2] = np.float32(300.0)               # one vehicle through the general-step redo as well
    z, ud, _, _, _, _ = oracle_mod.ekf_simulate_inputs(u, x0, x0, w)
    xo, Po, xho, pho = oracle_mod.ekf_run(x0, P0, z, ud, Q, R, want_phist=True)
    xd, Pd = _t(x0), _t(P0)
    xh = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
    ph = torch.empty((T, n, 16), dtype=torch.float32, device="cuda")
    ekf_run_addr64(xd, Pd, _t(z), _t(ud), Q, R, x_hist=xh, P_hist=ph)
    assert bit_equal(xh.cpu().numpy(), xho) and bit_equal(ph.cpu().numpy(), pho)
    assert bit_equal(xd.cpu().numpy(), xo) and bit_equal(Pd.cpu().numpy(), Po)


@pytest.mark.parametrize("n,T", [(1 << 20, 24), ((1 << 22) + 64 * 3 + 5, 9)])
def test_ekf_fused_large_batches(crx, oracle_mod, n, T):
    """Million-vehicle launches: 2^20 uses the buffer-addressed kernel at its intended scale, 2^22 + 197 the 64-bit-address
    one (beyond kEkfBufMaxN) with a ragged last workgroup.  A strided sample of vehicles is compared bit-for-bit."""
    import torch
    Q, R = ekf_QR()
    g = torch.Generator(device="cuda"); g.manual_seed(n % 1000)
    x0 = torch.zeros((n, 4), device="cuda")
    x0[:, 2] = (torch.rand(n, generator=g, device="cuda") - 0.5) * 6.0
    P0 = torch.eye(4, device="cuda").reshape(1, 16).repeat(n, 1).contiguous()
    z = torch.randn((T, n, 2), generator=g, device="cuda") * 0.5
    ud = torch.randn((T, n, 2), generator=g, device="cuda") * 0.3 + 1.0
    xh = torch.empty((T, n, 4), device="cuda")
    xd, Pd = x0.clone(), P0.clone()
    crx.ekf_run(xd, Pd, z, ud, Q, R, x_hist=xh)
    idx = torch.cat([torch.arange(0, n, 4099, device="cuda"), torch.arange(n - 70, n, device="cuda")])
    ii = idx.cpu().numpy()
    xo, Po, xho, _ = oracle_mod.ekf_run(x0[idx].cpu().numpy(), P0[idx].cpu().numpy(), z[:, idx].cpu().numpy(), ud[:, idx].cpu().numpy(), Q, R)
    assert bit_equal(xh[:, idx].cpu().numpy(), xho)
    assert bit_equal(xd[idx].cpu().numpy(), xo) and bit_equal(Pd[idx].cpu().numpy(), Po)
    assert len(ii) > 300


def test_ekf_fused_fuzz_wide_ranges(crx, oracle_mod):
    """Fuzz: states, covariances, measurements and Q/R spread over 60 decades (denormals, huge values, a few inf/nan),
    so that lanes leave and re-enter the fast domain in every combination."""
    import torch
    rng = np.random.default_rng(2025)
    n, T = 64 * 9, 13
    def wide(shape, lo=-30, hi=30):
        return (rng.choice([-1.0, 1.0], shape) * np.exp(rng.uniform(lo, hi, shape) * np.log(10.0))).astype(np.float32)
    x0 = wide((n, 4), -20, 8)
    x0[:, 2] = wide(n, -45, 3)                       # yaw from denormal to 1000 rad
    A = rng.standard_normal((n, 4, 4)).astype(np.float32)
    P0 = (np.einsum("nij,nkj->nik", A, A) * wide((n, 1, 1), -30, 30)).astype(np.float32).reshape(n, 16)
    z = wide((T, n, 2), -10, 6)
    ud = wide((T, n, 2), -10, 2)
    x0[5, 2] = np.inf; x0[70, 0] = np.nan; P0[130, 5] = np.inf; z[3, 200, 0] = np.nan; ud[5, 300, 1] = np.inf
    for qs, rs in ((1.0, 1.0), (1e-10, 1e-8), (1e18, 1e20)):
        Q0, R0 = ekf_QR()
        Q, R = (Q0 * np.float32(qs)).astype(np.float32), (R0 * np.float32(rs)).astype(np.float32)
        with np.errstate(all="ignore"):
            xo, Po, xho, pho = oracle_mod.ekf_run(x0, P0, z, ud, Q, R, want_phist=True)
        xd, Pd = _t(x0), _t(P0)
        xh = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
        ph = torch.empty((T, n, 16), dtype=torch.float32, device="cuda")
        crx.ekf_run(xd, Pd, _t(z), _t(ud), Q, R, x_hist=xh, P_hist=ph)
        # Vehicles whose reference run stays finite must match bit for bit.  Once an operand is inf/nan the two differ by
        # design: the engine skips the multiplications by the literal 0/1 entries of F_, jF, jH (exact for finite operands
        # only: 0*inf = nan in the dense Eigen expression), so a non-finite vehicle is only required to end non-finite.
        ok = np.isfinite(xho).all(axis=(0, 2)) & np.isfinite(pho).all(axis=(0, 2))
        assert ok.sum() > 100
        assert np.array_equal(xh.cpu().numpy()[:, ok], xho[:, ok]) and np.array_equal(ph.cpu().numpy()[:, ok], pho[:, ok])
        assert np.array_equal(Pd.cpu().numpy()[ok], Po[ok]) and np.array_equal(xd.cpu().numpy()[ok], xo[ok])
        bad = ~np.isfinite(xho[-1]).all(axis=1)
        assert not np.isfinite(xh.cpu().numpy()[-1][bad]).all(axis=1).any()


def test_ekf_step_is_graph_capturable_and_bit_exact(crx, oracle_mod):
    """The per-tick _dev entry point only enqueues (no allocation, no synchronisation): a HIP graph captured over K ticks
    replays to the same bits as K plain launches and as the oracle."""
    import torch
    n, K = 1000, 12
    Q, R = ekf_QR()
    u, x0, P0, w, z, ud = _inputs(oracle_mod, n, K, seed=99)
    x, P = x0.copy(), P0.copy()
    for t in range(K):
        x, P = oracle_mod.ekf_step(x, P, z[t], ud[t], Q, R)
    zd, udd = _t(z), _t(ud)
    xd, Pd = _t(x0), _t(P0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                    # warm-up outside the capture
        crx.ekf_estimation(xd.clone(), Pd.clone(), zd[0], udd[0], Q, R)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(K):
            crx.ekf_estimation(xd, Pd, zd[t], udd[t], Q, R)
    xd.copy_(_t(x0)); Pd.copy_(_t(P0))                               # capture does not execute: start from the initial state
    g.replay()
    torch.cuda.synchronize()
    assert bit_equal(xd.cpu().numpy(), x) and bit_equal(Pd.cpu().numpy(), P)


def test_two_lanes_per_vehicle_variant_equals_the_production_kernel(crx, oracle_mod):
    """The A/B variant (ekf_wave2_kernels.hip.h: a vehicle on a pair of lanes, DPP moves across the pair) reproduces the oracle
    as IEEE values on benign inputs — the numbers of profiles/r02/ekf_wave_ab.txt compare like with like."""
    import torch
    from cpprobotics_amd.experimental import ekf_run_pair
    Q, R = ekf_QR()
    for n, T in ((1, 9), (33, 64), (1000, 130)):
        u, x0, P0 = ekf_agents(n, 5 + n)
        z, ud, *_ = oracle_mod.ekf_simulate_inputs(u, x0, x0, ekf_noise(T, n, 6))
        xo, Po, xho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
        xd, Pd = _t(x0), _t(P0)
        xh = torch.empty((T, n, 4), device="cuda")
        assert ekf_run_pair(xd, Pd, _t(z), _t(ud), Q, R, x_hist=xh)          # every vehicle stayed on the fast domain
        assert np.array_equal(xh.cpu().numpy(), xho) and np.array_equal(xd.cpu().numpy(), xo) and np.array_equal(Pd.cpu().numpy(), Po)


def test_chunked_launches_equal_one_launch(crx):
    """The chunked trajectory gather cuts the T-step launch into launches of T/chunks steps with the filter state carried over in
    place (cpprobotics_amd/swarm.py: ChunkedTrajectoryGather): every byte of the history and of the final state must be the same."""
    import torch
    Q, R = ekf_QR()
    n, T = 777, 240
    u, x0, P0 = ekf_agents(n, 31)
    w = crx.normal_draws(n, T, agent0=5000, seed=11)
    xt, xd = _t(x0), _t(x0)
    z, ud = crx.ekf_simulate_inputs(_t(u), xt, xd, w)
    x1, P1 = _t(x0), _t(P0)
    h1 = torch.empty((T, n, 4), device="cuda")
    crx.ekf_run(x1, P1, z, ud, Q, R, x_hist=h1)
    for chunks in (2, 10, 60):
        x2, P2 = _t(x0), _t(P0)
        h2 = torch.empty((T, n, 4), device="cuda")
        Tc = T // chunks
        for c in range(chunks):
            crx.ekf_run(x2, P2, z[c * Tc:(c + 1) * Tc], ud[c * Tc:(c + 1) * Tc], Q, R, x_hist=h2[c * Tc:(c + 1) * Tc])
        assert torch.equal(h1.view(torch.int32), h2.view(torch.int32)) and torch.equal(x1, x2) and torch.equal(P1, P2)


@pytest.mark.gpu
def test_reciprocal_of_the_fused_step_is_the_ieee_quotient_on_this_device(crx):
    """recip_fast (csrc/ekf_math.h) is v_rcp_f32 and ONE Newton step; that this is the correctly rounded 1.0f / d on every float
    2^-60 <= |d| <= 2^60 is a property of the device's v_rcp_f32 — checked here, exhaustively, on the device the tests run on."""
    from cpprobotics_amd import experimental as X
    n, bad_two, bad_six = X.recip_sweep()
    print(f"recip sweep: {n} inputs, rcp + 2 fma: {bad_two} mismatches, rcp + 6 fma: {bad_six}")
    assert n == 2 * (0x5d800000 - 0x21800000 + 1) and bad_two == 0 and bad_six == 0


def _contract_err(crx, oracle_mod, x0, P0, z, ud, Q, R, dt=0.1):
    """-> (floored error of the whole x history, covariance error under the max|P_ref| floor, fraction of history floats that differ)"""
    import torch
    from cpprobotics_amd.experimental import ekf_run_contracted
    T, n = z.shape[0], x0.shape[0]
    xo, Po, xho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R, dt=dt)
    xd, Pd = _t(x0), _t(P0)
    xh = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
    ekf_run_contracted(xd, Pd, _t(z), _t(ud), Q, R, dt=dt, x_hist=xh)
    xh = xh.cpu().numpy()
    assert np.isfinite(xh).all() and bit_equal(xh[-1], xd.cpu().numpy())
    return floored_rel_err(xh, xho, 1.0), float(np.max(np.abs(Pd.cpu().numpy() - Po)) / np.max(np.abs(Po))), float(np.mean(xh != xho))


def test_contracted_arithmetic_experiment_single_vehicle_within_1e6(crx, oracle_mod):
    """crx_x_ekf_run_contracted_dev (round 6): the packed step with its multiply-then-add pairs fused — an EXPERIMENT, not a product
    mode.  On configs[0] — the reference's own single vehicle, 1000 steps — it is within north_star's 1e-6 (floored metric)."""
    Q, R = ekf_QR()
    u, x0, P0, w, z, ud = _inputs(oracle_mod, 1, 1000, seed=0, single=True)
    ex, eP, _ = _contract_err(crx, oracle_mod, x0, P0, z, ud, Q, R)
    assert ex <= 1e-6 and eP <= 1e-6, (ex, eP)


@pytest.mark.parametrize("n,T,seed", [(4096, 1000, 11), (65536, 200, 12), (1000, 999, 13)])
def test_contracted_arithmetic_experiment_is_not_within_1e6_for_every_vehicle(crx, oracle_mod, n, T, seed):
    """... and over a batch it is NOT: the worst vehicle of a few thousand leaves 1e-6 within a thousand steps (the filter's velocity
    state integrates the noisy input — an undamped random walk — so a last-bit perturbation is carried, not contracted).  VERDICT r5
    made "1e-6 for every vehicle" the condition for shipping the mode: this test RECORDS why it is not in crx_ekf_params.  What it
    asserts: the fused kernel really computes something else than the exact one (the flag is honoured), stays close (a sanity bound,
    NOT the contract: 5e-5), and — for the long runs — that the 1e-6 claim would indeed be false."""
    Q, R = ekf_QR()
    u, x0, P0, w, z, ud = _inputs(oracle_mod, n, T, seed=seed)
    ex, eP, differ = _contract_err(crx, oracle_mod, x0, P0, z, ud, Q, R)
    assert differ > 0.0
    assert ex <= 5e-5 and eP <= 5e-5, (ex, eP)
    if T >= 999:
        assert ex > 1e-6, f"the batch stayed within 1e-6 ({ex:.2e}): re-measure on the full workload (bench.py extra.ekf_contracted) before promoting the mode"


def test_contracted_arithmetic_experiment_falls_back_to_the_exact_arithmetic_for_other_dt(crx, oracle_mod):
    """dt other than the reference's 0.1: the entry point computes in the exact arithmetic — the reference's bits."""
    Q, R = ekf_QR()
    u, x0, P0, w, z, ud = _inputs(oracle_mod, 300, 40, seed=5)
    ex, eP, differ = _contract_err(crx, oracle_mod, x0, P0, z, ud, Q, R, dt=0.05)
    assert ex == 0.0 and eP == 0.0 and differ == 0.0

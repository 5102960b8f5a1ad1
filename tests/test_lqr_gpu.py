"""GPU parity: HIP DARE / dlqr kernels (through the C ABI) against the CPU oracle — bit-exact."""
import numpy as np
import pytest

from common import bit_equal, lqr_speeds

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("dim", [5, 4])
@pytest.mark.parametrize("n", [1, 64, 65, 1000])
def test_dare_from_v_bit_exact(crx, oracle_mod, dim, n):
    v = lqr_speeds(n, seed=n + dim)
    v[0] = 0.0  # iteration-cap case
    A, B, Q, R = oracle_mod.lqr_build(v, dim)
    Xo, Ko, ito = oracle_mod.dare(A, B, Q, R)
    K, X, it = crx.dlqr_from_v(_t(v), dim=dim)
    assert np.array_equal(it.cpu().numpy(), ito)
    assert bit_equal(X.cpu().numpy(), Xo)
    assert bit_equal(K.cpu().numpy(), Ko)
    assert ito[0] == 150


@pytest.mark.parametrize("lanes", [1, 4])
@pytest.mark.parametrize("dim", [5, 4])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000])
def test_dare_from_v_both_register_layouts_bit_exact(crx, oracle_mod, dim, n, lanes):
    """One agent per lane and one agent per DPP quad (forced through the experimental entry point; the product entry point
    picks by batch size): the same bits, iteration counts included; ragged last quad / wave, other dt / L / eps."""
    from cpprobotics_amd.experimental import dlqr_from_v_lanes
    v = lqr_speeds(n, seed=7 * n + dim)
    v[0] = 0.0
    for dt, L, eps, maxiter in ((0.1, 0.5, 0.01, 150), (0.05, 2.9, 1e-3, 40), (0.1, 0.5, 0.01, 1), (0.1, 0.5, 0.01, 0), (0.1, 0.5, 0.01, 2),
                                (0.1, 0.5, 0.01, 3), (0.1, 0.5, 1e9, 9), (0.2, 0.5, 1e-4, 77)):      # caps 0 / odd / even, a test that passes at once
        A, B, Q, R = oracle_mod.lqr_build(v, dim, dt=dt, L=L)
        Xo, Ko, ito = oracle_mod.dare(A, B, Q, R, eps=eps, maxiter=maxiter)
        K, X, it = dlqr_from_v_lanes(_t(v), dim, lanes, dt=dt, L_wheelbase=L, eps=eps, maxiter=maxiter)
        assert np.array_equal(it.cpu().numpy(), ito)
        assert bit_equal(X.cpu().numpy(), Xo)
        assert bit_equal(K.cpu().numpy(), Ko)


@pytest.mark.parametrize("lanes", [1, 4])
def test_dare_from_v_non_finite_speed(crx, oracle_mod, lanes):
    """NaN / huge speeds: the reference's loop runs to the cap (a NaN first element never compares below eps) and returns
    non-finite matrices; so does every layout here, for that agent only."""
    from cpprobotics_amd.experimental import dlqr_from_v_lanes
    v = lqr_speeds(64, seed=1)
    v[5] = np.nan; v[9] = 1e30; v[13] = -np.inf
    A, B, Q, R = oracle_mod.lqr_build(v, 5)
    Xo, Ko, ito = oracle_mod.dare(A, B, Q, R)
    K, X, it = dlqr_from_v_lanes(_t(v), 5, lanes)
    X, K, it = X.cpu().numpy(), K.cpu().numpy(), it.cpu().numpy()
    ok = np.isfinite(Xo).all(axis=1)
    assert (~ok).sum() == 3 and not ok[[5, 9, 13]].any()
    assert np.array_equal(it, ito)
    assert bit_equal(X[ok], Xo[ok]) and bit_equal(K[ok], Ko[ok])
    assert not np.isfinite(X[~ok]).all(axis=1).any()


@pytest.mark.parametrize("dim", [5, 4])
def test_dare_dense_matches_structured_and_oracle(crx, oracle_mod, dim):
    n = 777
    v = lqr_speeds(n, seed=11)
    A, B, Q, R = oracle_mod.lqr_build(v, dim)
    Xo, Ko, ito = oracle_mod.dare(A, B, Q, R)
    X, it = crx.solve_DARE(_t(A), _t(B), _t(Q), _t(R))
    K = crx.dlqr(_t(A), _t(B), _t(Q), _t(R))
    assert np.array_equal(it.cpu().numpy(), ito)
    assert bit_equal(X.cpu().numpy(), Xo) and bit_equal(K.cpu().numpy(), Ko)
    K2, X2, it2 = crx.dlqr_from_v(_t(v), dim=dim)
    assert bit_equal(X2.cpu().numpy(), X.cpu().numpy()) and bit_equal(K2.cpu().numpy(), K.cpu().numpy())
    # since round 4 solve_DARE() recognises these matrices and serves them by the structured kernels: the dense kernel itself on the
    # same matrices, forced
    from cpprobotics_amd.experimental import dare_dense
    X3, K3, it3 = dare_dense(_t(A), _t(B), _t(Q), _t(R))
    assert np.array_equal(it3.cpu().numpy(), ito) and bit_equal(X3.cpu().numpy(), Xo) and bit_equal(K3.cpu().numpy(), Ko)


def _pattern_mats(dim, dt, v, bv, bd):
    """A, B, Q, R with lqr_steering_control's zero pattern and per-agent free entries (column-major rows of length dim*dim ...)."""
    n = len(v)
    m = 2 if dim == 5 else 1
    A = np.zeros((n, dim * dim), np.float32); B = np.zeros((n, dim * m), np.float32)
    A[:, 0] = 1; A[:, 0 + dim * 1] = dt; A[:, 1 + dim * 2] = v; A[:, 2 + dim * 2] = 1; A[:, 2 + dim * 3] = dt
    B[:, 3] = bv
    if dim == 5:
        A[:, 24] = 1; B[:, 4 + 5] = bd
    Q = np.tile(np.eye(dim, dtype=np.float32).reshape(-1), (n, 1)); R = np.tile(np.eye(m, dtype=np.float32).reshape(-1), (n, 1))
    return A, B, Q, R


@pytest.mark.parametrize("dim", [5, 4])
@pytest.mark.parametrize("n", [1, 1000, 16384, 40000, 100001])
def test_dare_batch_recognises_the_reference_pattern(crx, oracle_mod, dim, n):
    """crx_dare_batch (the solve_DARE(A,B,Q,R) signature) on matrices that carry lqr_steering_control's pattern, the four free
    entries different for every agent and anywhere in the accepted box: served by the structured kernels (a DPP quad per agent up to
    32,768 agents, one lane per agent above: all three variants here) with the bits and iteration counts of the dense Eigen-order
    evaluation (oracle), and of the dense kernel forced on the same matrices."""
    from cpprobotics_amd.experimental import dare_dense
    rng = np.random.default_rng(n + dim)
    lu = lambda lo, hi: (np.exp(rng.uniform(np.log(lo), np.log(hi), n)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    v, dt, bv, bd = lu(1e-4, 100.0), lu(1e-3, 1.0), lu(1e-4, 1e3), lu(1e-3, 1.0)
    k = min(n, 64)
    v[:k] = lqr_speeds(k, seed=5); dt[:k] = np.float32(0.1); bd[:k] = np.float32(0.1); bv[:k] = (v[:k].astype(np.float64) / 0.5).astype(np.float32)
    v[0] = 0.0; bv[0] = 0.0                                       # the reference's stand-still case: the iteration cap
    A, B, Q, R = _pattern_mats(dim, dt, v, bv, bd)
    maxiter = 150 if n <= 16384 else 24
    Xo, Ko, ito = oracle_mod.dare(A, B, Q, R, maxiter=maxiter)
    assert np.isfinite(Xo).all()
    X, it = crx.solve_DARE(_t(A), _t(B), _t(Q), _t(R), maxiter=maxiter)
    K = crx.dlqr(_t(A), _t(B), _t(Q), _t(R), maxiter=maxiter)
    assert np.array_equal(it.cpu().numpy(), ito)
    assert bit_equal(X.cpu().numpy(), Xo) and bit_equal(K.cpu().numpy(), Ko)
    if n <= 16384:
        X3, K3, it3 = dare_dense(_t(A), _t(B), _t(Q), _t(R), maxiter=maxiter)
        assert np.array_equal(it3.cpu().numpy(), ito) and bit_equal(X3.cpu().numpy(), Xo) and bit_equal(K3.cpu().numpy(), Ko)


@pytest.mark.parametrize("dim", [5, 4])
def test_dare_batch_mixed_structured_and_dense_agents(crx, oracle_mod, dim):
    """One batch with every kind of agent: the reference's pattern, the pattern spoilt in one entry (a -0.0f, a 1e-30, A(0,1) !=
    A(2,3), Q or R off the identity by an ulp, a NaN), the pattern outside the accepted box (a vehicle at 500 m/s, dt = 1e-5), and
    dense random matrices — interleaved within waves and quads.  The entry point's result is the dense kernel's (forced) on every
    agent, bit for bit, iteration counts included, and the oracle's wherever that stays finite."""
    from cpprobotics_amd.experimental import dare_dense
    rng = np.random.default_rng(40 + dim)
    n, m = 3000, (2 if dim == 5 else 1)
    v = lqr_speeds(n, seed=9)
    A, B, Q, R = oracle_mod.lqr_build(v, dim)
    kind = rng.integers(0, 12, n)
    kind[:8] = np.arange(8)
    for a in range(n):
        k = kind[a]
        if k == 1: A[a, 1] = -0.0
        elif k == 2: A[a, 3 + dim * 0] = 1e-30
        elif k == 3: A[a, 2 + dim * 3] = np.nextafter(A[a, 0 + dim * 1], np.float32(1))
        elif k == 4: Q[a, 1 + dim * 1] = np.nextafter(np.float32(1), np.float32(2))
        elif k == 5: R[a, 0] = np.nextafter(np.float32(1), np.float32(0))
        elif k == 6: A[a, 1 + dim * 2] = 500.0
        elif k == 7: A[a, 0 + dim * 1] = A[a, 2 + dim * 3] = 1e-5
        elif k == 8: B[a, 3] = np.nan
        elif k == 9:
            A[a] = (np.eye(dim) * 0.9 + 0.15 * rng.standard_normal((dim, dim))).astype(np.float32).T.reshape(-1)
            B[a] = rng.standard_normal(dim * m).astype(np.float32)
        elif k == 10: Q[a, 2] = 0.25; Q[a, 2 * dim] = 0.25
    Xo, Ko, ito = oracle_mod.dare(A, B, Q, R)
    X, it = crx.solve_DARE(_t(A), _t(B), _t(Q), _t(R))
    K = crx.dlqr(_t(A), _t(B), _t(Q), _t(R))
    X3, K3, it3 = dare_dense(_t(A), _t(B), _t(Q), _t(R))
    X, K, it, X3, K3, it3 = (t.cpu().numpy() for t in (X, K, it, X3, K3, it3))
    assert np.array_equal(it, it3) and np.array_equal(it, ito)
    ok = np.isfinite(Xo).all(axis=1) & np.isfinite(Ko).all(axis=1)
    assert ok.sum() > n // 2 and (~ok).sum() > 0
    assert bit_equal(X[ok], Xo[ok]) and bit_equal(K[ok], Ko[ok]) and bit_equal(X3[ok], Xo[ok])
    # non-finite agents took the dense kernel in both calls: identical bit patterns (NaN payloads included)
    assert np.array_equal(X[~ok].view(np.uint32), X3[~ok].view(np.uint32)) and np.array_equal(K[~ok].view(np.uint32), K3[~ok].view(np.uint32))


@pytest.mark.parametrize("lanes", [1, 4])
@pytest.mark.parametrize("dim", [5, 4])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 2500])
def test_dare_dense_both_register_layouts(crx, oracle_mod, dim, n, lanes):
    """The two dense kernels forced (one agent per lane / one row of X per lane of a quad) on dense random matrices — where every
    product's accumulation order matters — and on the reference's matrices: the oracle's bits, iteration counts included; ragged last
    quad / wave; caps 0 / 1 / odd / even; agents whose iterates turn non-finite stay non-finite (same NaN / inf positions)."""
    from cpprobotics_amd.experimental import dare_dense
    rng = np.random.default_rng(100 * dim + n)
    m = 2 if dim == 5 else 1
    A = (np.eye(dim)[None] * 0.9 + 0.15 * rng.standard_normal((n, dim, dim))).astype(np.float32)
    B = rng.standard_normal((n, dim, m)).astype(np.float32)
    Qh = rng.standard_normal((n, dim, dim)).astype(np.float32)
    Q = (np.einsum("nij,nkj->nik", Qh, Qh) * 0.2 + np.eye(dim)[None]).astype(np.float32)
    Rh = rng.standard_normal((n, m, m)).astype(np.float32)
    R = (np.einsum("nij,nkj->nik", Rh, Rh) + np.eye(m)[None]).astype(np.float32)
    cm = lambda M_: np.ascontiguousarray(np.transpose(M_, (0, 2, 1))).reshape(n, -1)
    for eps, maxiter in ((1e-3, 60), (0.01, 7), (0.01, 1), (0.01, 0), (1e9, 4)):
        Xo, Ko, ito = oracle_mod.dare(cm(A), cm(B), cm(Q), cm(R), eps=eps, maxiter=maxiter)
        X, K, it = dare_dense(_t(cm(A)), _t(cm(B)), _t(cm(Q)), _t(cm(R)), eps=eps, maxiter=maxiter, lanes_per_agent=lanes)
        X, K, it = X.cpu().numpy(), K.cpu().numpy(), it.cpu().numpy()
        assert np.array_equal(it, ito)
        fin = np.isfinite(Xo).all(axis=1) & np.isfinite(Ko).all(axis=1)
        assert bit_equal(X[fin], Xo[fin]) and bit_equal(K[fin], Ko[fin])
        assert np.array_equal(np.isfinite(X), np.isfinite(Xo))
    v = lqr_speeds(n, seed=n)
    v[0] = 0.0
    Ar, Br, Qr, Rr = oracle_mod.lqr_build(v, dim)
    Xo, Ko, ito = oracle_mod.dare(Ar, Br, Qr, Rr)
    X, K, it = dare_dense(_t(Ar), _t(Br), _t(Qr), _t(Rr), lanes_per_agent=lanes)
    assert np.array_equal(it.cpu().numpy(), ito) and bit_equal(X.cpu().numpy(), Xo) and bit_equal(K.cpu().numpy(), Ko)


@pytest.mark.parametrize("dim", [5, 4])
def test_dare_dense_general_matrices(crx, oracle_mod, dim):
    """Arbitrary (dense) A, B, Q, R: the order of accumulation matters here."""
    rng = np.random.default_rng(dim)
    n, m = 500, (2 if dim == 5 else 1)
    A = (np.eye(dim)[None] * 0.9 + 0.15 * rng.standard_normal((n, dim, dim))).astype(np.float32)
    B = rng.standard_normal((n, dim, m)).astype(np.float32)
    Qh = rng.standard_normal((n, dim, dim)).astype(np.float32)
    Q = (np.einsum("nij,nkj->nik", Qh, Qh) * 0.2 + np.eye(dim)[None]).astype(np.float32)
    Rh = rng.standard_normal((n, m, m)).astype(np.float32)
    R = (np.einsum("nij,nkj->nik", Rh, Rh) + np.eye(m)[None]).astype(np.float32)
    cm = lambda M: np.ascontiguousarray(np.transpose(M, (0, 2, 1))).reshape(n, -1)  # column-major blocks
    Xo, Ko, ito = oracle_mod.dare(cm(A), cm(B), cm(Q), cm(R), eps=1e-3, maxiter=60)
    X, it = crx.solve_DARE(_t(cm(A)), _t(cm(B)), _t(cm(Q)), _t(cm(R)), eps=1e-3, maxiter=60)
    K = crx.dlqr(_t(cm(A)), _t(cm(B)), _t(cm(Q)), _t(cm(R)), eps=1e-3, maxiter=60)
    assert np.array_equal(it.cpu().numpy(), ito)
    ok = np.isfinite(Xo).all(axis=1)
    assert ok.sum() > n // 2
    assert bit_equal(X.cpu().numpy()[ok], Xo[ok]) and bit_equal(K.cpu().numpy()[ok], Ko[ok])


def test_dare_full_size(crx, oracle_mod):
    """BASELINE config 3: 16,384 agents, 5x5, to convergence; plus the 4x4 variant."""
    n = 16384
    v = lqr_speeds(n, seed=3)
    for dim in (5, 4):
        A, B, Q, R = oracle_mod.lqr_build(v, dim)
        Xo, Ko, ito = oracle_mod.dare(A, B, Q, R)
        K, X, it = crx.dlqr_from_v(_t(v), dim=dim)
        assert np.array_equal(it.cpu().numpy(), ito)
        assert bit_equal(X.cpu().numpy(), Xo) and bit_equal(K.cpu().numpy(), Ko)
        assert (ito == 150).sum() > 0 and ito.min() >= 40


@pytest.mark.parametrize("n", [40000, 100001])
def test_dare_from_v_every_kernel_variant(crx, oracle_mod, n):
    """The product entry point picks its kernel by batch size: a DPP quad per agent up to 32,768 agents (test_dare_full_size), one agent per
    lane with nobody masked off up to 98,304, the masked loop at eight waves per SIMD beyond.  Same bits from all of them, iteration counts and
    odd / even / tiny caps included."""
    v = lqr_speeds(n, seed=n)
    for dim, maxiter in ((5, 150), (4, 150), (5, 7), (4, 2)):
        A, B, Q, R = oracle_mod.lqr_build(v, dim)
        Xo, Ko, ito = oracle_mod.dare(A, B, Q, R, maxiter=maxiter)
        K, X, it = crx.dlqr_from_v(_t(v), dim=dim, maxiter=maxiter)
        assert np.array_equal(it.cpu().numpy(), ito)
        assert bit_equal(X.cpu().numpy(), Xo)
        if maxiter == 150:
            assert bit_equal(K.cpu().numpy(), Ko)


@pytest.mark.parametrize("dim", [5, 4])
def test_dare_lane_refilling_kernel(crx, oracle_mod, dim):
    """dare_from_v_refill_kernel (one agent per lane; a lane whose agent is done takes the next agent of its wave's range — what the
    product uses above 262,144 agents): forced on small batches with odd range lengths and hand-back thresholds, ragged sizes, every
    kind of cap — the oracle's bits and iteration counts on every agent."""
    from cpprobotics_amd.experimental import dlqr_from_v_refill
    for n, chunk, hold, maxiter, eps in ((5000, 64, 1, 150, 0.01), (5001, 100, 7, 150, 0.01), (20000, 777, 16, 150, 0.01), (3000, 4096, 64, 150, 0.01),
                                         (4000, 128, 16, 7, 0.01), (4000, 320, 3, 1, 0.01), (4000, 320, 16, 2, 0.01), (4000, 256, 16, 150, 1e9),
                                         (63, 64, 16, 150, 0.01), (1, 64, 1, 150, 0.01)):
        v = lqr_speeds(n, seed=n + chunk)
        v[0] = 0.0
        A, B, Q, R = oracle_mod.lqr_build(v, dim)
        Xo, Ko, ito = oracle_mod.dare(A, B, Q, R, eps=eps, maxiter=maxiter)
        K, X, it = dlqr_from_v_refill(_t(v), dim, chunk, hold, eps=eps, maxiter=maxiter)
        assert np.array_equal(it.cpu().numpy(), ito), (n, chunk, hold, maxiter)
        assert bit_equal(X.cpu().numpy(), Xo) and bit_equal(K.cpu().numpy(), Ko), (n, chunk, hold, maxiter)


def test_dare_throughput_regime_entry_points(crx, oracle_mod):
    """Above 262,144 agents both entry points — crx_dare_from_v_batch and the solve_DARE(A, B, Q, R) signature on matrices that carry
    the pattern, with other agents interleaved — run the lane-refilling kernel: equal to the masked kernel (rounds 2-3) on every agent,
    and to the oracle on a sample."""
    import torch
    from cpprobotics_amd.experimental import dlqr_from_v_refill
    n = 300007
    v = lqr_speeds(n, seed=12)
    for dim in (5, 4):
        K, X, it = crx.dlqr_from_v(_t(v), dim=dim)
        Km, Xm, itm = dlqr_from_v_refill(_t(v), dim, -1)
        assert torch.equal(it, itm) and torch.equal(X.view(torch.int32), Xm.view(torch.int32)) and torch.equal(K.view(torch.int32), Km.view(torch.int32))
        idx = np.r_[0:3000, n - 3000:n]
        A, B, Q, R = oracle_mod.lqr_build(v[idx], dim)
        Xo, Ko, ito = oracle_mod.dare(A, B, Q, R)
        assert np.array_equal(it.cpu().numpy()[idx], ito) and bit_equal(X.cpu().numpy()[idx], Xo) and bit_equal(K.cpu().numpy()[idx], Ko)
        # the dense signature: every 7th agent's matrices spoilt (a -0.0f where the pattern has +0.0f): those go to the dense kernel
        A, B, Q, R = oracle_mod.lqr_build(v, dim)
        A[::7, 1] = -0.0
        X2, it2 = crx.solve_DARE(_t(A), _t(B), _t(Q), _t(R))
        assert torch.equal(it2, it) and torch.equal(X2, X)          # (-0.0 where the reference multiplies by it changes no value)
        del A, B, Q, R


def test_dare_edge_cases(crx, oracle_mod):
    import torch
    v = np.array([1.0, 2.0], dtype=np.float32)
    # maxiter = 1: one evaluation, returned whether or not it converged
    A, B, Q, R = oracle_mod.lqr_build(v, 5)
    Xo, Ko, ito = oracle_mod.dare(A, B, Q, R, maxiter=1)
    X, it = crx.solve_DARE_from_v(_t(v), dim=5, maxiter=1)
    assert bit_equal(X.cpu().numpy(), Xo) and np.array_equal(it.cpu().numpy(), ito)
    # n = 0
    X, it = crx.solve_DARE_from_v(torch.empty((0,), device="cuda"), dim=4)
    assert X.shape == (0, 16)
    with pytest.raises(crx.CrxError):
        crx.solve_DARE(torch.zeros((2, 9), device="cuda"), torch.zeros((2, 3), device="cuda"),
                       torch.zeros((2, 9), device="cuda"), torch.zeros((2, 1), device="cuda"))

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

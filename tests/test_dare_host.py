"""Host logic of the structured Riccati kernels (csrc/dare_math.h): the one-lane-per-agent iteration on packed rows and the
four-lanes-per-agent iteration (DPP quad exchanges emulated by a 4-lane value type) built for the CPU and compared bit for bit
with the oracle's dense Eigen-order evaluation.  No GPU needed; the same cases run through the HIP kernels in
tests/test_lqr_gpu.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from common import bit_equal, lqr_speeds

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "tools", "dare_host.cpp")


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("dare") / "dare_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, SRC])
    lib = C.CDLL(so)

    def run(kind, v, dim, dt=0.1, L=0.5, eps=0.01, maxiter=150):
        n = len(v)
        X = np.empty((n, dim * dim), dtype=np.float32)
        it = np.empty(n, dtype=np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        fn = getattr(lib, f"dare_{kind}_run")
        rc = fn(C.c_int(n), C.c_int(dim), vp(np.ascontiguousarray(v, dtype=np.float32)), C.c_double(dt), C.c_double(L),
                C.c_float(eps), C.c_int(maxiter), vp(X), vp(it))
        assert rc == 0
        return X, it

    def run_params(kind, dim, dt, v, bv, bd, eps=0.01, maxiter=150):
        n = len(v)
        X = np.empty((n, dim * dim), dtype=np.float32)
        it = np.empty(n, dtype=np.int32)
        vp = lambda a: np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.c_void_p)
        keep = [np.ascontiguousarray(a, dtype=np.float32) for a in (v, dt, bv, bd)]
        rc = getattr(lib, f"dare_{kind}_run_params")(C.c_int(n), C.c_int(dim), *[k.ctypes.data_as(C.c_void_p) for k in keep], C.c_float(eps),
                                                     C.c_int(maxiter), X.ctypes.data_as(C.c_void_p), it.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return X, it
    def run_dense_quad(dim, A, B, Q, R, eps=0.01, maxiter=150):
        n = len(A)
        X = np.empty((n, dim * dim), dtype=np.float32)
        it = np.empty(n, dtype=np.int32)
        keep = [np.ascontiguousarray(a, dtype=np.float32) for a in (A, B, Q, R)]
        rc = lib.dare_dense_quad_run(C.c_int(n), C.c_int(dim), *[k.ctypes.data_as(C.c_void_p) for k in keep], C.c_float(eps), C.c_int(maxiter),
                                     X.ctypes.data_as(C.c_void_p), it.ctypes.data_as(C.c_void_p))
        assert rc == 0, "the lanes of a quad disagree on row 4"
        return X, it
    run.params = run_params
    run.dense_quad = run_dense_quad
    return run


@pytest.mark.parametrize("kind", ["lane", "quad"])
@pytest.mark.parametrize("dim", [5, 4])
def test_structured_iteration_equals_the_dense_oracle(host, oracle_mod, kind, dim):
    v = lqr_speeds(3000, seed=17 + dim)
    v[:4] = [0.0, -0.0, 1e-3, -7.5]
    A, B, Q, R = oracle_mod.lqr_build(v, dim)
    Xo, _, ito = oracle_mod.dare(A, B, Q, R)
    X, it = host(kind, v, dim)
    assert np.array_equal(it, ito)
    assert bit_equal(X, Xo)
    assert (ito == 150).sum() > 0


@pytest.mark.parametrize("kind", ["lane", "quad"])
@pytest.mark.parametrize("dim", [5, 4])
def test_structured_iteration_other_parameters(host, oracle_mod, kind, dim):
    """Other dt / L / eps / iteration caps (the block structure does not depend on them), wide speed range."""
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.uniform(-30, 30, 500), 10.0 ** rng.uniform(-6, 2.5, 300) * rng.choice([-1, 1], 300)]).astype(np.float32)
    for dt, L, eps, maxiter in ((0.1, 0.5, 1e-3, 40), (0.05, 2.9, 0.01, 150), (0.2, 0.5, 1e-4, 7), (0.1, 0.5, 0.01, 1), (0.1, 0.5, 0.01, 0), (0.1, 0.5, 1e9, 9)):
        A, B, Q, R = oracle_mod.lqr_build(v, dim, dt=dt, L=L)
        Xo, _, ito = oracle_mod.dare(A, B, Q, R, eps=eps, maxiter=maxiter)
        X, it = host(kind, v, dim, dt=dt, L=L, eps=eps, maxiter=maxiter)
        ok = np.isfinite(Xo).all(axis=1)
        assert ok.mean() > 0.9
        assert np.array_equal(it[ok], ito[ok])
        assert bit_equal(X[ok], Xo[ok])
        assert not np.isfinite(X[~ok]).all(axis=1).any()      # non-finite in the reference -> non-finite here


@pytest.mark.parametrize("kind", ["lane", "quad"])
@pytest.mark.parametrize("dim", [5, 4])
def test_structured_iteration_over_the_accepted_box(host, oracle_mod, kind, dim):
    """crx_dare_batch serves every agent whose dense arguments carry lqr_steering_control's pattern by the structured iteration
    (DareFromMats, csrc/dare_kernels.hip.h) when its four free entries lie in the box |v| <= 100, 1e-3 <= |dt|, |bd| <= 1,
    |bv| <= 1e3, each chosen per agent: over that box (log-uniform samples, both signs, its corners) the structured iteration is the
    dense Eigen-order evaluation bit for bit, iteration counts included, and every iterate stays finite (largest entry < 1e11)."""
    import itertools
    rng = np.random.default_rng(100 + dim)
    n = 6000
    lu = lambda lo, hi, k=n: (np.exp(rng.uniform(np.log(lo), np.log(hi), k)) * rng.choice([-1.0, 1.0], k)).astype(np.float32)
    v, dt, bv, bd = lu(1e-6, 100.0), lu(1e-3, 1.0), lu(1e-6, 1e3), lu(1e-3, 1.0)
    corners = np.array(list(itertools.product([100.0, -100.0, 1e-6, 0.0], [1e-3, 1.0, -1.0], [1e3, -1e3, 1e-6, 0.0], [1e-3, 1.0])), dtype=np.float32)
    v = np.concatenate([v, corners[:, 0]]); dt = np.concatenate([dt, corners[:, 1]])
    bv = np.concatenate([bv, corners[:, 2]]); bd = np.concatenate([bd, corners[:, 3]])
    n = len(v)
    m = 2 if dim == 5 else 1
    A = np.zeros((n, dim * dim), np.float32); B = np.zeros((n, dim * m), np.float32)
    A[:, 0] = 1; A[:, 0 + dim * 1] = dt; A[:, 1 + dim * 2] = v; A[:, 2 + dim * 2] = 1; A[:, 2 + dim * 3] = dt
    B[:, 3] = bv
    if dim == 5:
        A[:, 24] = 1; B[:, 4 + 5] = bd
    Q = np.tile(np.eye(dim, dtype=np.float32).reshape(-1), (n, 1)); R = np.tile(np.eye(m, dtype=np.float32).reshape(-1), (n, 1))
    for maxiter in (150, 1, 2, 3, 5, 17):                  # every iterate, not only the last, must agree and be finite
        Xo, _, ito = oracle_mod.dare(A, B, Q, R, maxiter=maxiter)
        assert np.isfinite(Xo).all() and np.abs(Xo).max() < 1e11
        X, it = host.params(kind, dim, dt, v, bv, bd, maxiter=maxiter)
        assert np.array_equal(it, ito)
        assert bit_equal(X, Xo)


@pytest.mark.parametrize("dim", [5, 4])
def test_dense_quad_rows_equal_the_dense_oracle(host, oracle_mod, dim):
    """dare_dense_quad_rows (csrc/dare_dense_math.h) — the dense iteration with one row of X per lane of a quad, row 4 on every lane
    — run lane by lane on the CPU: the oracle's dense Eigen-order evaluation bit for bit, iteration counts included, on dense random
    matrices (where the accumulation order of every product matters), on the reference's own matrices, and on matrices whose iterates
    turn non-finite."""
    rng = np.random.default_rng(dim)
    n, m = 1500, (2 if dim == 5 else 1)
    A = (np.eye(dim)[None] * 0.9 + 0.15 * rng.standard_normal((n, dim, dim))).astype(np.float32)
    B = rng.standard_normal((n, dim, m)).astype(np.float32)
    Qh = rng.standard_normal((n, dim, dim)).astype(np.float32)
    Q = (np.einsum("nij,nkj->nik", Qh, Qh) * 0.2 + np.eye(dim)[None]).astype(np.float32)
    Rh = rng.standard_normal((n, m, m)).astype(np.float32)
    R = (np.einsum("nij,nkj->nik", Rh, Rh) + np.eye(m)[None]).astype(np.float32)
    cm = lambda M: np.ascontiguousarray(np.transpose(M, (0, 2, 1))).reshape(n, -1)
    for eps, maxiter in ((1e-3, 60), (0.01, 7), (1e9, 3), (0.01, 0)):
        Xo, _, ito = oracle_mod.dare(cm(A), cm(B), cm(Q), cm(R), eps=eps, maxiter=maxiter)
        X, it = host.dense_quad(dim, cm(A), cm(B), cm(Q), cm(R), eps=eps, maxiter=maxiter)
        assert np.array_equal(it, ito)
        assert np.array_equal(X.view(np.uint32), Xo.view(np.uint32))             # NaN / inf patterns included: the same operations
    v = lqr_speeds(1000, seed=2)
    v[:3] = [0.0, 1e30, np.nan]
    Ar, Br, Qr, Rr = oracle_mod.lqr_build(v, dim)
    Xo, _, ito = oracle_mod.dare(Ar, Br, Qr, Rr)
    X, it = host.dense_quad(dim, Ar, Br, Qr, Rr)
    nan = np.isnan(Xo)
    assert np.array_equal(it, ito) and np.array_equal(np.isnan(X), nan)          # (a NaN's sign / payload is not part of the contract)
    assert np.array_equal(X[~nan].view(np.uint32), Xo[~nan].view(np.uint32))


def test_block_structure_of_the_reference_iterates(oracle_mod):
    """What rule 2 of dare_math.h rests on: the dense evaluation itself returns exact zeros off the 4 + 1 blocks."""
    v = lqr_speeds(2000, seed=3)
    A, B, Q, R = oracle_mod.lqr_build(v, 5)
    for maxiter in (1, 2, 3, 10, 150):
        Xo, _, _ = oracle_mod.dare(A, B, Q, R, maxiter=maxiter)
        X = Xo.reshape(-1, 5, 5)
        assert (X[:, 4, :4] == 0).all() and (X[:, :4, 4] == 0).all()
        assert not np.signbit(X[:, 4, :4]).any() and not np.signbit(X[:, :4, 4]).any()

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


def test_block_structure_of_the_reference_iterates(oracle_mod):
    """What rule 2 of dare_math.h rests on: the dense evaluation itself returns exact zeros off the 4 + 1 blocks."""
    v = lqr_speeds(2000, seed=3)
    A, B, Q, R = oracle_mod.lqr_build(v, 5)
    for maxiter in (1, 2, 3, 10, 150):
        Xo, _, _ = oracle_mod.dare(A, B, Q, R, maxiter=maxiter)
        X = Xo.reshape(-1, 5, 5)
        assert (X[:, 4, :4] == 0).all() and (X[:, :4, 4] == 0).all()
        assert not np.signbit(X[:, 4, :4]).any() and not np.signbit(X[:, :4, 4]).any()

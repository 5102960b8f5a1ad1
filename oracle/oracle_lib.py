"""ctypes front-end of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

_P = C.c_void_p
_I = C.c_int
_D = C.c_double
_F = C.c_float


def build(force=False):
    """g++ -O2 -ffp-contract=off (oracle/Makefile).  Idempotent."""
    srcs = [os.path.join(_HERE, f) for f in ("ekf_ref.cpp", "lqr_ref.cpp", "mpc_ref.cpp", "eigen_order.h", "Makefile")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_libm_is_fma_flavour.restype = _I
        _lib.oracle_dare.restype = _I
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def trig_mode():
    """0 = host libm (the reference's behaviour) when libm is glibc's FMA flavour — the one the
    HIP kernels reproduce; otherwise 1 = the explicit FMA-flavour restatement."""
    return 0 if lib().oracle_libm_is_fma_flavour() else 1


def libm_is_fma_flavour():
    return bool(lib().oracle_libm_is_fma_flavour())


# ---- EKF ---------------------------------------------------------------------------------------
def motion_model(x, u, dt=0.1, trig=None):
    x, u = _f32(x), _f32(u)
    out = np.empty_like(x)
    lib().oracle_motion_model(_I(x.shape[0]), _p(x), _p(u), _p(out), _D(dt), _I(trig_mode() if trig is None else trig))
    return out


def jacobF(x, u, dt=0.1, trig=None):
    x, u = _f32(x), _f32(u)
    out = np.empty((x.shape[0], 16), dtype=np.float32)
    lib().oracle_jacobF(_I(x.shape[0]), _p(x), _p(u), _p(out), _D(dt), _I(trig_mode() if trig is None else trig))
    return out


def observation_model(x):
    x = _f32(x)
    out = np.empty((x.shape[0], 2), dtype=np.float32)
    lib().oracle_observation_model(_I(x.shape[0]), _p(x), _p(out))
    return out


def jacobH():
    out = np.empty(8, dtype=np.float32)
    lib().oracle_jacobH(_p(out))
    return out


def ekf_step(x, P, z, u, Q, R, dt=0.1, trig=None, sum_order=0):
    """Returns new (x, P); inputs untouched."""
    x, P, z, u, Q, R = _f32(x).copy(), _f32(P).copy(), _f32(z), _f32(u), _f32(Q), _f32(R)
    lib().oracle_ekf_step(_I(x.shape[0]), _p(x), _p(P), _p(z), _p(u), _p(Q), _p(R), _D(dt),
                          _I(trig_mode() if trig is None else trig), _I(sum_order))
    return x, P


def ekf_run(x, P, z, u, Q, R, dt=0.1, trig=None, sum_order=0, want_xhist=True, want_phist=False, agents=None):
    """z,u: [T,n,2].  Returns (x, P, x_hist or None, P_hist or None)."""
    x, P, z, u, Q, R = _f32(x).copy(), _f32(P).copy(), _f32(z), _f32(u), _f32(Q), _f32(R)
    T, n = z.shape[0], x.shape[0]
    xh = np.zeros((T, n, 4), dtype=np.float32) if want_xhist else None
    ph = np.zeros((T, n, 16), dtype=np.float32) if want_phist else None
    a0, a1 = (0, n) if agents is None else agents
    lib().oracle_ekf_run(_I(n), _I(T), _p(x), _p(P), _p(z), _p(u), _p(xh), _p(ph), _p(Q), _p(R), _D(dt),
                         _I(trig_mode() if trig is None else trig), _I(sum_order), _I(a0), _I(a1))
    return x, P, xh, ph


def ekf_simulate_inputs(u_true, xTrue, xDR, w, dt=0.1, qsim=None, rsim=None, trig=None, want_hist=False):
    import math
    if qsim is None:
        qsim = (1.0, (30.0 / 180 * math.pi) * (30.0 / 180 * math.pi))
    if rsim is None:
        rsim = (0.25, 0.25)
    u_true, xTrue, xDR, w = _f32(u_true), _f32(xTrue).copy(), _f32(xDR).copy(), _f32(w)
    T, n = w.shape[0], w.shape[1]
    z = np.empty((T, n, 2), dtype=np.float32)
    ud = np.empty((T, n, 2), dtype=np.float32)
    xth = np.empty((T, n, 4), dtype=np.float32) if want_hist else None
    xdh = np.empty((T, n, 4), dtype=np.float32) if want_hist else None
    q, r = _f32(qsim), _f32(rsim)
    lib().oracle_ekf_simulate_inputs(_I(n), _I(T), _p(u_true), _p(xTrue), _p(xDR), _p(w), _p(z), _p(ud), _p(xth), _p(xdh),
                                     _p(q), _p(r), _D(dt), _I(trig_mode() if trig is None else trig))
    return z, ud, xTrue, xDR, xth, xdh


# ---- LQR ---------------------------------------------------------------------------------------
def lqr_build(v, dim=5, dt=0.1, L=0.5):
    v = _f32(v)
    n = v.shape[0]
    m = 2 if dim == 5 else 1
    A = np.empty((n, dim * dim), dtype=np.float32)
    B = np.empty((n, dim * m), dtype=np.float32)
    Q = np.empty((n, dim * dim), dtype=np.float32)
    R = np.empty((n, m * m), dtype=np.float32)
    lib().oracle_lqr_build(_I(n), _I(dim), _p(v), _D(dt), _D(L), _p(A), _p(B), _p(Q), _p(R))
    return A, B, Q, R


def dare(A, B, Q, R, eps=0.01, maxiter=150, sum_order=0, agents=None):
    """Returns X [n,dim*dim], K [n,m*dim], iters [n]."""
    A, B, Q, R = _f32(A), _f32(B), _f32(Q), _f32(R)
    n = A.shape[0]
    dim = 5 if A.shape[1] == 25 else 4
    m = 2 if dim == 5 else 1
    X = np.zeros((n, dim * dim), dtype=np.float32)
    K = np.zeros((n, m * dim), dtype=np.float32)
    it = np.zeros((n,), dtype=np.int32)
    a0, a1 = (0, n) if agents is None else agents
    rc = lib().oracle_dare(_I(n), _I(dim), _p(A), _p(B), _p(Q), _p(R), _F(eps), _I(maxiter), _p(X), _p(K), _p(it),
                           _I(sum_order), _I(a0), _I(a1))
    assert rc == 0
    return X, K, it


# ---- MPC ---------------------------------------------------------------------------------------
import math as _math

MPC_DEFAULTS = dict(dt=0.2, wb=2.5, max_steer=45.0 / 180 * _math.pi, max_accel=1.0, max_speed=55.0 / 3.6,
                    min_speed=-20.0 / 3.6, r_a=0.01, r_delta=0.01, rd_a=0.01, rd_delta=1.0, q_x=1.0, q_y=1.0,
                    q_yaw=0.5, q_v=0.5, tol=1e-9)
_MPC_ORDER = ("dt", "wb", "max_steer", "max_accel", "max_speed", "min_speed", "r_a", "r_delta", "rd_a", "rd_delta",
              "q_x", "q_y", "q_yaw", "q_v", "tol")


def _mpc_params(overrides):
    d = dict(MPC_DEFAULTS)
    d.update(overrides or {})
    return np.array([d[k] for k in _MPC_ORDER], dtype=np.float64)


def mpc_solve(x0, xref, T, params=None, max_iter=50, agents=None):
    """CPU twin of the engine's MPC solver.  Returns sol [n, 4T+2(T-1)], status [n], cost [n]."""
    x0, xref = _f32(x0), _f32(xref)
    n = x0.shape[0]
    nv = 4 * T + 2 * (T - 1)
    sol = np.zeros((n, nv), dtype=np.float32)
    status = np.zeros((n,), dtype=np.int32)
    cost = np.zeros((n,), dtype=np.float64)
    pp = _mpc_params(params)
    a0, a1 = (0, n) if agents is None else agents
    lib().oracle_mpc_solve(_I(n), _I(T), _p(x0), _p(xref), _p(pp), _I(max_iter), _p(sol), _p(status), _p(cost), _I(a0), _I(a1))
    return sol, status, cost


def mpc_cost(x0, xref, T, U, params=None):
    """NLP objective of ONE agent for controls U [(T-1),2] (delta, a); returns (J, S [(T),6])."""
    x0, xref = _f32(x0), _f32(xref)
    U = np.ascontiguousarray(U, dtype=np.float64)
    S = np.zeros((T, 6), dtype=np.float64)
    pp = _mpc_params(params)
    f = lib().oracle_mpc_cost
    f.restype = _D
    J = f(_I(T), _p(x0), _p(xref), _p(pp), _p(U), _p(S))
    return float(J), S

"""Host-side mirror of the reference's solve_DARE()/dlqr(), batched over n agents.

5x5/2-input: /root/reference/src/lqr_speed_steer_control.cpp:85-106;
4x4/1-input: /root/reference/src/lqr_steer_control.cpp:75-96.
Matrices are float32 CUDA tensors [n, rows*cols], column-major per agent.
"""
import ctypes as C

from . import _lib as L


def _dims(A):
    nn = A.shape[1]
    if nn == 25:
        return 5, 2
    if nn == 16:
        return 4, 1
    raise L.CrxError("solve_DARE: A must be [n,25] (5x5) or [n,16] (4x4)")


def _dare(A, B, Q, R, eps, maxiter, want_X, want_K):
    import torch
    L.require_cuda(A, B, Q, R)
    n = A.shape[0]
    dim, m = _dims(A)
    L.expect("A", A, "f", n, dim * dim); L.expect("B", B, "f", n, dim * m); L.expect("Q", Q, "f", n, dim * dim); L.expect("R", R, "f", n, m * m)
    X = torch.empty((n, dim * dim), dtype=torch.float32, device=A.device) if want_X else None
    K = torch.empty((n, m * dim), dtype=torch.float32, device=A.device) if want_K else None
    iters = torch.empty((n,), dtype=torch.int32, device=A.device)
    L.check(L.lib().crx_dare_batch_dev(n, dim, L.ptr(A), L.ptr(B), L.ptr(Q), L.ptr(R), float(eps), int(maxiter),
                                       L.ptr(X), L.ptr(K), L.ptr(iters), L.stream_ptr()), "crx_dare_batch_dev")
    return X, K, iters


def solve_DARE(A, B, Q, R, eps=0.01, maxiter=150):
    """solve_DARE(A,B,Q,R) -> X (and the per-agent iteration count)."""
    X, _, iters = _dare(A, B, Q, R, eps, maxiter, True, False)
    return X, iters


def dlqr(A, B, Q, R, eps=0.01, maxiter=150):
    """dlqr(A,B,Q,R) -> K ([n,10] = 2x5 col-major, or [n,4])."""
    _, K, _ = _dare(A, B, Q, R, eps, maxiter, False, True)
    return K


def _params(dt, Lw, eps, maxiter):
    p = L.LqrParams()
    p.dt, p.L, p.eps, p.maxiter = float(dt), float(Lw), float(eps), int(maxiter)
    return p


def _from_v(v, dim, dt, Lw, eps, maxiter, want_X, want_K):
    import torch
    L.require_cuda(v)
    n = v.shape[0]
    L.expect("v", v, "f", n)
    if dim not in (4, 5):
        raise L.CrxError("dim must be 4 or 5")
    m = 2 if dim == 5 else 1
    X = torch.empty((n, dim * dim), dtype=torch.float32, device=v.device) if want_X else None
    K = torch.empty((n, m * dim), dtype=torch.float32, device=v.device) if want_K else None
    iters = torch.empty((n,), dtype=torch.int32, device=v.device)
    p = _params(dt, Lw, eps, maxiter)
    L.check(L.lib().crx_dare_from_v_batch_dev(n, dim, L.ptr(v), C.byref(p), L.ptr(X), L.ptr(K), L.ptr(iters),
                                              L.stream_ptr()), "crx_dare_from_v_batch_dev")
    return X, K, iters


def solve_DARE_from_v(v, dim=5, dt=0.1, L_wheelbase=0.5, eps=0.01, maxiter=150):
    """solve_DARE with A,B,Q,R built from the speed as lqr_steering_control() does."""
    X, _, iters = _from_v(v, dim, dt, L_wheelbase, eps, maxiter, True, False)
    return X, iters


def dlqr_from_v(v, dim=5, dt=0.1, L_wheelbase=0.5, eps=0.01, maxiter=150):
    X, K, iters = _from_v(v, dim, dt, L_wheelbase, eps, maxiter, True, True)
    return K, X, iters

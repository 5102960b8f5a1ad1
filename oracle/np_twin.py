"""Independent numpy-float32 restatement of the reference arithmetic — TEST INFRASTRUCTURE ONLY.

A second, separately written CPU statement of
  /root/reference/src/extended_kalman_filter.cpp:22-78 and
  /root/reference/src/lqr_speed_steer_control.cpp:85-106 / src/lqr_steer_control.cpp:75-96
using scalar np.float32 operations (every + and * rounds to float32, nothing is fused), dense
loops, and Eigen 3.3.9's accumulation orders re-derived here (see oracle/eigen_order.h for the
rules).  tests/ require it to agree BITWISE with the C++ oracle.  sinf/cosf come from the host
libm through ctypes, i.e. the very functions the reference's std::cos(float) resolves to.
"""
import ctypes
import ctypes.util

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.sinf.restype = ctypes.c_float
_libm.sinf.argtypes = [ctypes.c_float]
_libm.cosf.restype = ctypes.c_float
_libm.cosf.argtypes = [ctypes.c_float]


def sinf(x):
    return f32(_libm.sinf(float(x)))


def cosf(x):
    return f32(_libm.cosf(float(x)))


def _tree(t):
    if len(t) == 1:
        return t[0]
    h = len(t) // 2
    return f32(_tree(t[:h]) + _tree(t[h:]))


def _padd(ps):
    if len(ps) == 1:
        return ps[0]
    m = len(ps) // 2
    a, b = _padd(ps[:m]), _padd(ps[m:])
    return [f32(a[l] + b[l]) for l in range(4)]


def mul(A, B, lhs_t=False, rhs_t=False, order="eigen"):
    """A (R x K) times B (K x C), float32, one rounding per operation, Eigen's summation order."""
    R, K = A.shape
    C = B.shape[1]
    can_l, can_r = (not lhs_t) and R != 1, rhs_t and C != 1       # CanVectorizeLhs / Rhs (no `% 4` term: eigen_order.h)
    eval_rm = True if (R == 1 and C != 1) else False if (C == 1 and R != 1) else (rhs_t and not can_l)
    dst_rm = R == 1 and C != 1
    is_vec = R == 1 or C == 1
    inner = R * C if is_vec else (C if dst_rm else R)
    by_packets = 0        # leading positions (along the temporary's storage order) that the assignment evaluates by packets
    if (can_l or can_r) and eval_rm == dst_rm:
        if inner % 4 == 0:
            by_packets = inner                     # InnerVectorizedTraversal
        elif is_vec or inner >= 4:
            by_packets = (inner // 4) * 4          # Linear- / SliceVectorizedTraversal, unrolled: the rest by coeff()
    sse = lhs_t and (not rhs_t) and K >= 4
    out = np.zeros((R, C), dtype=f32)
    for j in range(C):
        for i in range(R):
            t = [f32(A[i, k] * B[k, j]) for k in range(K)]
            if order == "asc" or (j if dst_rm else i) < by_packets:
                s = t[0]
                for k in range(1, K):
                    s = f32(s + t[k])
            elif sse:                       # vectorised redux: packets pairwise (halves), SSE2 predux, then the k % 4 tail
                p = _padd([t[4 * q:4 * q + 4] for q in range(K // 4)])
                s = f32(f32(p[0] + p[2]) + f32(p[1] + p[3]))
                if K % 4:
                    s = f32(s + _tree(t[4 * (K // 4):]))
            else:
                s = _tree(t)
            out[i, j] = s
    return out


def inverse2(m):
    det = f32(f32(m[0, 0] * m[1, 1]) - f32(m[1, 0] * m[0, 1]))
    invdet = f32(f32(1.0) / det)
    r = np.zeros((2, 2), dtype=f32)
    r[0, 0] = f32(m[1, 1] * invdet)
    r[1, 0] = f32(f32(-m[1, 0]) * invdet)
    r[0, 1] = f32(f32(-m[0, 1]) * invdet)
    r[1, 1] = f32(m[0, 0] * invdet)
    return r


# ---- EKF -----------------------------------------------------------------------------------------
def motion_model(x, u, DT=0.1, order="eigen"):
    F = np.eye(4, dtype=f32)
    B = np.zeros((4, 2), dtype=f32)
    B[0, 0] = f32(DT * float(cosf(x[2, 0])))
    B[1, 0] = f32(DT * float(sinf(x[2, 0])))
    B[2, 1] = f32(DT)
    B[3, 0] = f32(1.0)
    return (mul(F, x, order=order) + mul(B, u, order=order)).astype(f32)


def jacobF(x, u, DT=0.1):
    jF = np.eye(4, dtype=f32)
    yaw, v = x[2, 0], u[0, 0]
    jF[0, 2] = f32((-DT * float(v)) * float(sinf(yaw)))
    jF[0, 3] = f32(DT * float(cosf(yaw)))
    jF[1, 2] = f32((DT * float(v)) * float(cosf(yaw)))
    jF[1, 3] = f32(DT * float(sinf(yaw)))
    return jF


def jacobH():
    h = np.zeros((2, 4), dtype=f32)
    h[0, 0] = 1
    h[1, 1] = 1
    return h


def ekf_estimation(xEst, PEst, z, u, Q, R, DT=0.1, order="eigen"):
    """xEst (4,1), PEst (4,4), z (2,1), u (2,1), Q (4,4), R (2,2) float32 -> new (xEst, PEst)."""
    xPred = motion_model(xEst, u, DT, order)
    jF = jacobF(xPred, u, DT)
    PPred = (mul(mul(jF, PEst, order=order), jF.T, rhs_t=True, order=order) + Q).astype(f32)
    jH = jacobH()
    zPred = mul(jH, xPred, order=order)
    y = (z - zPred).astype(f32)
    S = (mul(mul(jH, PPred, order=order), jH.T, rhs_t=True, order=order) + R).astype(f32)
    K = mul(mul(PPred, jH.T, rhs_t=True, order=order), inverse2(S), order=order)
    xNew = (xPred + mul(K, y, order=order)).astype(f32)
    PNew = mul((np.eye(4, dtype=f32) - mul(K, jH, order=order)).astype(f32), PPred, order=order)
    return xNew, PNew


# ---- DARE / dlqr ---------------------------------------------------------------------------------
def lqr_build(v, dim=5, DT=0.1, L=0.5):
    v = f32(v)
    A = np.zeros((dim, dim), dtype=f32)
    A[0, 0] = 1
    A[0, 1] = f32(DT)
    A[1, 2] = v
    A[2, 2] = 1
    A[2, 3] = f32(DT)
    if dim == 5:
        A[4, 4] = 1
        B = np.zeros((5, 2), dtype=f32)
        B[3, 0] = f32(float(v) / L)
        B[4, 1] = f32(DT)
        return A, B, np.eye(5, dtype=f32), np.eye(2, dtype=f32)
    B = np.zeros((4, 1), dtype=f32)
    B[3, 0] = f32(float(v) / L)
    return A, B, np.eye(4, dtype=f32), np.eye(1, dtype=f32)


def _dare_iter(A, B, Q, R, X, order):
    dim = A.shape[0]
    AtX = mul(A.T, X, lhs_t=True, order=order)
    P1 = mul(AtX, A, order=order)
    BtX = mul(B.T, X, lhs_t=True, order=order)
    if dim == 5:
        G = mul(BtX, B, order=order)
        Si = inverse2((R + G).astype(f32))
        c2 = mul(mul(AtX, B, order=order), Si, order=order)
    else:
        g = mul(BtX, B, lhs_t=True, order=order)[0, 0]   # row-vector temporary is row-major
        s = f32(R[0, 0] + g)
        c2 = (mul(AtX, B, order=order) / s).astype(f32)
    c3 = mul(c2, B.T, rhs_t=True, order=order)
    P2 = mul(mul(c3, X, order=order), A, order=order)
    return ((P1 - P2).astype(f32) + Q).astype(f32)


def solve_DARE(A, B, Q, R, eps=0.01, maxiter=150, order="eigen"):
    X = Q.copy()
    eps = f32(eps)
    for i in range(maxiter):
        Xn = _dare_iter(A, B, Q, R, X, order)
        if np.max(np.abs((Xn - X).astype(f32))) < eps:
            return Xn, i + 1
        X = Xn
    return X, maxiter


def dlqr(A, B, Q, R, eps=0.01, maxiter=150, order="eigen"):
    X, it = solve_DARE(A, B, Q, R, eps, maxiter, order)
    dim = A.shape[0]
    BtX = mul(B.T, X, lhs_t=True, order=order)
    if dim == 5:
        Si = inverse2((mul(BtX, B, order=order) + R).astype(f32))
        K = mul(Si, mul(BtX, A, order=order), order=order)
    else:
        g = mul(BtX, B, lhs_t=True, order=order)[0, 0]
        inv = f32(1.0 / float(f32(g + R[0, 0])))
        K = (inv * mul(BtX, A, lhs_t=True, order=order)).astype(f32)
    return K, X, it

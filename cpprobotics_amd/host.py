"""Host-pointer side of the C ABI from Python: numpy arrays in, numpy arrays out (no torch tensors) — the entry points a C++
fleet host binds (include/crx.h), device selection, the device set that shards a batch over several GPUs, pinned buffers.

    crx.host.set_devices([0, 1, 2, 3])            # every host-pointer batch call now splits its agents over four GPUs
    x, P, hist = crx.host.ekf_run(x, P, z, u, Q, R, want_hist=True)

The arithmetic is the `_dev` kernels'; these wrappers only hand pointers over."""
import ctypes as C

import numpy as np

from . import _lib as L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, *shape):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape and a.shape != tuple(shape):
        raise L.CrxError(f"expected shape {shape}, got {a.shape}")
    return a


def set_device(device):
    L.check(L.lib().crx_set_device(int(device)), "crx_set_device")


def get_device():
    return L.lib().crx_get_device()


def set_devices(devices=None, min_agents_per_device=4096):
    """Install (or, with None / [], remove) the device set of the host-pointer batch entry points."""
    devices = list(devices or [])
    arr = (C.c_int * max(1, len(devices)))(*devices)
    L.check(L.lib().crx_set_devices(arr if devices else None, len(devices), int(min_agents_per_device)), "crx_set_devices")


def get_devices():
    n = L.lib().crx_get_devices(None, 0)
    arr = (C.c_int * max(1, n))()
    L.lib().crx_get_devices(arr, n)
    return list(arr[:n])


def reserve_workspace(device_bytes, pinned_bytes):
    """Grow the current device's workspaces ahead of the first large call."""
    L.check(L.lib().crx_reserve_workspace(int(device_bytes), int(pinned_bytes)), "crx_reserve_workspace")


def release_workspace():
    L.check(L.lib().crx_release_workspace(), "crx_release_workspace")


class PinnedArray:
    """A numpy view of pinned host memory from crx_host_alloc (DMA'd in place by the host-pointer entry points); free() or the
    garbage collector gives it back."""

    def __init__(self, shape, dtype=np.float32):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._ptr = L.lib().crx_host_alloc(self.nbytes)
        if not self._ptr:
            raise L.CrxError("crx_host_alloc failed: " + L.lib().crx_last_error().decode("utf-8", "replace"))
        buf = (C.c_char * self.nbytes).from_address(self._ptr)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self._ptr:
            self.array = None
            L.lib().crx_host_free(C.c_void_p(self._ptr))
            self._ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def ekf_run(x, P, z, u, Q, R, dt=0.1, want_hist=False, want_P_hist=False, x_hist=None, P_hist=None):
    """crx_ekf_run_batch: T fused ekf_estimation() steps for n vehicles, host arrays.  z, u: [T, n, 2].  x [n, 4] and P [n, 16]
    are updated IN PLACE when they are contiguous float32 arrays (copies are returned otherwise).  -> x, P, x_hist, P_hist."""
    z, u = np.asarray(z), np.asarray(u)
    T, n = z.shape[0], z.shape[1]
    x = x if (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.flags.c_contiguous) else _f32(x).copy()
    P = P if (isinstance(P, np.ndarray) and P.dtype == np.float32 and P.flags.c_contiguous) else _f32(P).copy()
    if x.shape != (n, 4) or P.shape != (n, 16) or z.shape != (T, n, 2) or u.shape != (T, n, 2):
        raise L.CrxError("ekf_run: x [n,4], P [n,16], z and u [T,n,2]")
    if z.dtype != np.float32 or u.dtype != np.float32 or not z.flags.c_contiguous or not u.flags.c_contiguous:
        z, u = _f32(z), _f32(u)
    if x_hist is None and want_hist:
        x_hist = np.empty((T, n, 4), np.float32)
    if P_hist is None and want_P_hist:
        P_hist = np.empty((T, n, 16), np.float32)
    from .ekf import _qr
    q, r = _qr(Q, R)                                       # 16 / 4 floats, column-major as the caller holds them
    prm = L.EkfParams(); prm.dt = float(dt)
    L.check(L.lib().crx_ekf_run_batch(n, T, _p(x), _p(P), _p(z), _p(u), _p(x_hist), _p(P_hist), _p(q), _p(r), C.byref(prm)), "crx_ekf_run_batch")
    return x, P, x_hist, P_hist


def dare(A, B, Q, R, eps=0.01, maxiter=150):
    """crx_dare_batch (solve_DARE + dlqr for n agents, host arrays) -> X, K, iters."""
    A, B, Q, R = _f32(A), _f32(B), _f32(Q), _f32(R)
    n = A.shape[0]
    dim = 5 if A.shape[1] == 25 else 4
    m = 2 if dim == 5 else 1
    X = np.empty((n, dim * dim), np.float32); K = np.empty((n, m * dim), np.float32); it = np.empty((n,), np.int32)
    L.check(L.lib().crx_dare_batch(n, dim, _p(A), _p(B), _p(Q), _p(R), float(eps), int(maxiter), _p(X), _p(K), _p(it)), "crx_dare_batch")
    return X, K, it


def dare_from_v(v, dim=5, dt=0.1, L_wheelbase=0.5, eps=0.01, maxiter=150):
    """crx_dare_from_v_batch -> X, K, iters."""
    from .lqr import _params
    v = _f32(v)
    n = v.shape[0]
    m = 2 if dim == 5 else 1
    X = np.empty((n, dim * dim), np.float32); K = np.empty((n, m * dim), np.float32); it = np.empty((n,), np.int32)
    p = _params(dt, L_wheelbase, eps, maxiter)
    L.check(L.lib().crx_dare_from_v_batch(n, dim, _p(v), C.byref(p), _p(X), _p(K), _p(it)), "crx_dare_from_v_batch")
    return X, K, it


def mpc_solve(x0, xref, T, params=None, portfolio=False):
    """crx_mpc_solve_batch (portfolio=True: crx_mpc_solve_portfolio_batch) -> sol [n, 4T + 2(T-1)], status, cost."""
    from .mpc import default_params, mpc_n_vars
    x0, xref = _f32(x0), _f32(xref)
    n = x0.shape[0]
    p = params if params is not None else default_params()
    sol = np.empty((n, mpc_n_vars(T)), np.float32); st = np.empty((n,), np.int32); cost = np.empty((n,), np.float64)
    fn = L.lib().crx_mpc_solve_portfolio_batch if portfolio else L.lib().crx_mpc_solve_batch
    L.check(fn(n, int(T), _p(x0), _p(xref), C.byref(p), _p(sol), _p(st), _p(cost)), "crx_mpc_solve_batch")
    return sol, st, cost


def _course(course):
    arrs = [None if a is None else _f32(a) for a in course]
    c = L.Course(len(arrs[0]), *[_p(a) for a in arrs])
    return c, arrs


def lqr_closed_loop(state, course, goal, dim=5, max_ticks=500, goal_dis=None, want_hist=False, dt=0.1, L_wheelbase=0.5, eps=0.01, maxiter=150,
                    kp=1.0, stop_speed=0.05):
    """crx_lqr_closed_loop_batch (closed_loop_prediction for n agents, host arrays) -> state, ticks, traj_hist."""
    from .track import _lqr_params, loop_params, vehicle_params
    state = _f32(state).copy()
    n = state.shape[0]
    c, keep = _course(course)
    if goal_dis is None:
        goal_dis = 0.3 if dim == 5 else 0.5
    hist = np.empty((max_ticks, n, 4), np.float32) if want_hist else None
    ticks = np.empty((n,), np.int32)
    p = _lqr_params(dt, L_wheelbase, eps, maxiter); vp = vehicle_params(False, dt=float(dt), wheelbase=float(L_wheelbase))
    lp = loop_params(goal, goal_dis, max_ticks, kp, stop_speed)
    L.check(L.lib().crx_lqr_closed_loop_batch(n, dim, _p(state), C.byref(c), None, None, None, C.byref(p), C.byref(vp), C.byref(lp), _p(hist),
                                              _p(ticks)), "crx_lqr_closed_loop_batch")
    return state, ticks, hist


def mpc_closed_loop(state, course, goal, T=6, max_ticks=120, goal_dis=0.5, dl=1.0, nsearch=10, want_hist=False, params=None):
    """crx_mpc_closed_loop_batch (mpc_simulation for n agents, host arrays) -> state, ticks, traj_hist, target_ind, solve_flags."""
    from .mpc import default_params
    from .track import loop_params
    state = _f32(state).copy()
    n = state.shape[0]
    c, keep = _course(course)
    hist = np.empty((max_ticks, n, 4), np.float32) if want_hist else None
    ticks = np.empty((n,), np.int32); tind = np.zeros((n,), np.int32); flags = np.empty((n,), np.int32)
    p = params if params is not None else default_params()
    lp = loop_params(goal, goal_dis, max_ticks)
    L.check(L.lib().crx_mpc_closed_loop_batch(n, int(T), _p(state), C.byref(c), float(dl), int(nsearch), C.byref(p), C.byref(lp), _p(tind), _p(hist),
                                              _p(ticks), _p(flags)), "crx_mpc_closed_loop_batch")
    return state, ticks, hist, tind, flags

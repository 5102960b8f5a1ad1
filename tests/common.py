"""Shared synthetic workloads and the error metric used by the parity tests."""
import math

import numpy as np


def ekf_QR():
    Q = np.zeros((4, 4), dtype=np.float32)
    Q[0, 0] = np.float32(0.1 * 0.1)
    Q[1, 1] = np.float32(0.1 * 0.1)
    Q[2, 2] = np.float32((1.0 / 180 * math.pi) * (1.0 / 180 * math.pi))
    Q[3, 3] = np.float32(0.1 * 0.1)
    R = np.eye(2, dtype=np.float32)
    return Q.T.copy().reshape(-1), R.T.copy().reshape(-1)


def ekf_agents(n, seed, single_vehicle=False):
    """Per-vehicle true input and initial state.  C1: the reference's own u=(1.0,0.1), x=0.
    C2: yaw ~ U(-pi,pi), u0 ~ U(0.5,2), u1 ~ U(-0.3,0.3)."""
    rng = np.random.default_rng(seed)
    if single_vehicle:
        u = np.tile(np.array([[1.0, 0.1]], dtype=np.float32), (n, 1))
        x0 = np.zeros((n, 4), dtype=np.float32)
    else:
        u = np.stack([rng.uniform(0.5, 2.0, n), rng.uniform(-0.3, 0.3, n)], axis=1).astype(np.float32)
        x0 = np.zeros((n, 4), dtype=np.float32)
        x0[:, 2] = rng.uniform(-math.pi, math.pi, n).astype(np.float32)
    P0 = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (n, 1))
    return u, x0, P0


def ekf_noise(T, n, seed):
    rng = np.random.Generator(np.random.Philox(key=seed))
    return rng.standard_normal((T, n, 4), dtype=np.float32)


def lqr_speeds(n, seed):
    """C3: v ~ U(-3, 6) with 5 % of the agents at |v| < 0.1 (iteration-cap case)."""
    rng = np.random.default_rng(seed)
    v = rng.uniform(-3.0, 6.0, n)
    slow = rng.random(n) < 0.05
    v[slow] = rng.uniform(-0.1, 0.1, slow.sum())
    return v.astype(np.float32)


def floored_rel_err(a, b, floor):
    """max |a-b| / max(|b|, floor)  (SURVEY.md §8d error definition)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def bit_equal(a, b):
    """Equality as IEEE values (+0 == -0), NaN never equal — the parity bar for fp32 paths."""
    return bool(np.array_equal(np.asarray(a), np.asarray(b)))


# ---- MPC workload (C4) -----------------------------------------------------------------------------
def natural_cubic_spline(s, y):
    """Coefficients of the natural cubic spline the reference's Spline class builds
    (/root/reference/include/cubic_spline.h:53-65, 95-116), in float64."""
    s = np.asarray(s, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    nx = len(s); h = np.diff(s)
    A = np.zeros((nx, nx)); B = np.zeros(nx)
    A[0, 0] = 1.0
    for i in range(nx - 1):
        if i != nx - 2:
            A[i + 1, i + 1] = 2.0 * (h[i] + h[i + 1])
        A[i + 1, i] = h[i]; A[i, i + 1] = h[i]
    A[0, 1] = 0.0; A[nx - 1, nx - 2] = 0.0; A[nx - 1, nx - 1] = 1.0
    for i in range(nx - 2):
        B[i + 1] = 3.0 * (y[i + 2] - y[i + 1]) / h[i + 1] - 3.0 * (y[i + 1] - y[i]) / h[i]
    c = np.linalg.solve(A, B)
    d = (c[1:] - c[:-1]) / (3.0 * h)
    b = (y[1:] - y[:-1]) / h - h * (c[1:] + 2.0 * c[:-1]) / 3.0
    return s, y, b, c, d


def _spline_eval(sp, t):
    s, a, b, c, d = sp
    i = np.clip(np.searchsorted(s, t, side="right") - 1, 0, len(s) - 2)
    dx = t - s[i]
    val = a[i] + b[i] * dx + c[i] * dx ** 2 + d[i] * dx ** 3
    d1 = b[i] + 2 * c[i] * dx + 3 * d[i] * dx ** 2
    return val, d1


def mpc_course(ds=1.0):
    """The reference's MPC course: Spline2D through its waypoints, sampled every ds
    (/root/reference/src/model_predictive_control.cpp:469-486), speed profile 10 km/h (:488)."""
    wx = [0.0, 60.0, 125.0, 50.0, 75.0, 35.0, -10.0]
    wy = [0.0, 0.0, 50.0, 65.0, 30.0, 50.0, -20.0]
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(wx), np.diff(wy)))])
    sx, sy = natural_cubic_spline(s, wx), natural_cubic_spline(s, wy)
    t = np.arange(0.0, s[-1], ds)
    cx, dx = _spline_eval(sx, t); cy, dy = _spline_eval(sy, t)
    cyaw = np.unwrap(np.arctan2(dy, dx))                      # smooth_yaw (:172-185)
    sp = np.full_like(cx, 10.0 / 3.6)
    return cx, cy, cyaw, sp


def mpc_problem(n, T, seed, dt=0.2, dl=1.0):
    """n independent MPC problems: x0 = a course point + perturbation (lateral ~N(0,0.3 m), yaw ~N(0,5 deg),
    v ~ U(0,4) m/s); xref by the calc_ref_trajectory rule (:130-170): xref(:,i) = course[ind + round(travel_i/dl)],
    travel_i = (i+1)*|v|*DT, clamped to the last course point.  Returns x0 [n,4], xref [n,4*T] (column-major 4xT)."""
    cx, cy, cyaw, sp = mpc_course(dl)
    rng = np.random.default_rng(seed)
    nc = len(cx)
    ind = rng.integers(0, nc - 5, n)
    lat = rng.normal(0.0, 0.3, n); dyaw = rng.normal(0.0, math.radians(5.0), n); v = rng.uniform(0.0, 4.0, n)
    x0 = np.stack([cx[ind] - lat * np.sin(cyaw[ind]), cy[ind] + lat * np.cos(cyaw[ind]), cyaw[ind] + dyaw, v], axis=1)
    x0 = x0.astype(np.float32)
    xref = np.zeros((n, T, 4), dtype=np.float32)
    for i in range(T):
        travel = (i + 1) * np.abs(x0[:, 3].astype(np.float64)) * dt
        j = np.minimum(ind + np.round(travel / dl).astype(np.int64), nc - 1)
        xref[:, i, 0] = cx[j]; xref[:, i, 1] = cy[j]; xref[:, i, 2] = cyaw[j]; xref[:, i, 3] = sp[j]
    return x0, xref.reshape(n, 4 * T)


# ---- LQR course (lqr_speed_steer_control.cpp main(): waypoints :248-249, ds = 0.1, target speed 10 km/h) -------
def _spline_dd(sp, t):
    s, a, b, c, d = sp
    i = np.clip(np.searchsorted(s, t, side="right") - 1, 0, len(s) - 2)
    return 2 * c[i] + 6 * d[i] * (t - s[i])


def lqr_course(ds=0.1):
    """Course arrays (cx, cy, cyaw, ck, sp) and the goal, built as the reference's main() builds them (Spline2D
    through its waypoints, calc_speed_profile's end-of-course slow-down :55-60).  Input data for the tracking
    kernels; float64 maths rounded to float32 (the course is an input, not part of the parity contract)."""
    wx = [0.0, 6.0, 12.5, 10.0, 17.5, 20.0, 25.0]
    wy = [0.0, -3.0, -5.0, 6.5, 3.0, 0.0, 0.0]
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(wx), np.diff(wy)))])
    sx, sy = natural_cubic_spline(s, wx), natural_cubic_spline(s, wy)
    t = np.arange(0.0, s[-1], ds)
    cx, dx = _spline_eval(sx, t)
    cy, dy = _spline_eval(sy, t)
    ddx, ddy = _spline_dd(sx, t), _spline_dd(sy, t)
    cyaw = np.arctan2(dy, dx)
    ck = (ddy * dx - ddx * dy) / (dx * dx + dy * dy)
    sp = np.full_like(cx, 10.0 / 3.6)
    for k in range(1, 40):
        sp[-k] = max((10.0 / 3.6) / (50 - k), 1.0 / 3.6)
    course = tuple(np.ascontiguousarray(a, dtype=np.float32) for a in (cx, cy, cyaw, ck, sp))
    return course, (float(wx[-1]), float(wy[-1]))


def mpc_course_f32(ds=1.0):
    cx, cy, cyaw, sp = mpc_course(ds)
    ck = np.zeros_like(cx)
    sp = sp.copy(); sp[-1] = 0.0
    course = tuple(np.ascontiguousarray(a, dtype=np.float32) for a in (cx, cy, cyaw, ck, sp))
    return course, (-10.0, -20.0)


def tracking_agents(n, course, seed, spread=0.5):
    """n vehicle states scattered around random course points (lateral/longitudinal ~N(0,spread), yaw ~N(0,0.2),
    v ~ U(-1, 4)) — the batch analogue of the reference's single start state."""
    rng = np.random.default_rng(seed)
    cx, cy, cyaw = course[0], course[1], course[2]
    i = rng.integers(0, len(cx), n)
    st = np.stack([cx[i] + rng.normal(0, spread, n), cy[i] + rng.normal(0, spread, n),
                   cyaw[i] + rng.normal(0, 0.2, n), rng.uniform(-1.0, 4.0, n)], axis=1)
    return st.astype(np.float32)


def speed_bound_problems(n, T, seed, fast=True):
    """Problems whose optimum rides a speed bound: the reference trajectory asks for 60 km/h (MAX_SPEED is 55) or for
    -25 km/h in reverse (MIN_SPEED is -20), from a start a little inside the bound."""
    x0, xref = mpc_problem(n, T, seed=seed)
    rng = np.random.default_rng(seed + 1)
    xr = xref.reshape(n, T, 4).copy()
    if fast:
        x0[:, 3] = rng.uniform(14.8, 15.25, n).astype(np.float32)
        xr[:, :, 3] = np.float32(60.0 / 3.6)
    else:
        x0[:, 3] = rng.uniform(-5.5, -5.1, n).astype(np.float32)
        xr[:, :, 3] = np.float32(-25.0 / 3.6)
    # a reference the vehicle can follow at that speed: straight ahead of the start pose
    step = xr[:, :, 3] * 0.2
    for i in range(T):
        xr[:, i, 0] = x0[:, 0] + np.cos(x0[:, 2]) * step[:, :i + 1].sum(axis=1)
        xr[:, i, 1] = x0[:, 1] + np.sin(x0[:, 2]) * step[:, :i + 1].sum(axis=1)
        xr[:, i, 2] = x0[:, 2]
    return x0, xr.reshape(n, 4 * T)


def mpc_solve_threads(oracle_mod, x0, xref, T, **kw):
    """The CPU twin on every core (the twin's entry point takes an agent range; ctypes releases the GIL)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = len(x0)
    nt = max(1, min(os.cpu_count() or 1, 64, n // 16 or 1))
    cuts = [k * n // nt for k in range(nt + 1)]
    with ThreadPoolExecutor(nt) as ex:
        parts = list(ex.map(lambda k: oracle_mod.mpc_solve(x0[cuts[k]:cuts[k + 1]], xref[cuts[k]:cuts[k + 1]], T, **kw), range(nt)))
    return tuple(np.concatenate([p[j] for p in parts]) for j in range(3))

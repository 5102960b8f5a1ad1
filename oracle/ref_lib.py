"""ctypes front-end of oracle/_ref/libref.so — TEST INFRASTRUCTURE ONLY.

libref.so is the reference's own source lines (cut out of /root/reference at build time by oracle/ref_build.sh) compiled
against the host's Eigen when there is one, else against the stand-in oracle/ref_shim/Eigen/Eigen.  It is what the oracle is
pinned against: tests/test_oracle_vs_ref.py runs both on the same inputs, tests/golden/make_ref_golden.py stores its outputs
as fixtures for hosts that have neither /root/reference nor the built library.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libref.so")
_lib = None
_P, _I, _F = C.c_void_p, C.c_int, C.c_float

SOLVER_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double))


def available():
    return os.path.exists(_SO)


def build(reference_root="/root/reference"):
    """Runs oracle/ref_build.sh when the reference sources are present; returns whether libref.so exists afterwards."""
    if os.path.isdir(os.path.join(reference_root, "src")):
        subprocess.check_call(["bash", os.path.join(_HERE, "ref_build.sh"), reference_root], stdout=subprocess.DEVNULL)
    return available()


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
        _lib.ref_eigen_kind.restype = _I
    return _lib


class flavour:
    """`with flavour("cpath_asc"): ...` — run the functions of this module against libref_cpath_asc.so / libref_cpath_tree.so, the
    stand-in builds in which every coefficient-path sum of a product is forced to one order (oracle/eigen_order.h)."""

    def __init__(self, name):
        self.path = os.path.join(_HERE, "_ref", f"libref_{name}.so")

    def available(self):
        return os.path.exists(self.path)

    def __enter__(self):
        global _lib
        self.saved = _lib
        _lib = C.CDLL(self.path)
        _lib.ref_eigen_kind.restype = _I
        return self

    def __exit__(self, *a):
        global _lib
        _lib = self.saved


def eigen_kind():
    return "host Eigen" if lib().ref_eigen_kind() == 1 else "stand-in (oracle/ref_shim/Eigen/Eigen)"


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


# ---- EKF (src/extended_kalman_filter.cpp) -------------------------------------------------------------------------------
def motion_model(x, u):
    x, u = _f32(x), _f32(u)
    out = np.empty_like(x)
    lib().ref_motion_model(_I(len(x)), _p(x), _p(u), _p(out))
    return out


def jacobF(x, u):
    x, u = _f32(x), _f32(u)
    out = np.empty((len(x), 16), np.float32)
    lib().ref_jacobF(_I(len(x)), _p(x), _p(u), _p(out))
    return out


def observation_model(x):
    x = _f32(x)
    out = np.empty((len(x), 2), np.float32)
    lib().ref_observation_model(_I(len(x)), _p(x), _p(out))
    return out


def jacobH():
    out = np.empty(8, np.float32)
    lib().ref_jacobH(_p(out))
    return out


def ekf_run(x, P, z, u, Q, R, want_phist=False):
    x, P, z, u, Q, R = _f32(x).copy(), _f32(P).copy(), _f32(z), _f32(u), _f32(Q), _f32(R)
    T, n = z.shape[0], x.shape[0]
    xh = np.zeros((T, n, 4), np.float32)
    ph = np.zeros((T, n, 16), np.float32) if want_phist else None
    lib().ref_ekf_run(_I(n), _I(T), _p(x), _p(P), _p(z), _p(u), _p(xh), _p(ph), _p(Q), _p(R))
    return x, P, xh, ph


def ekf_main(noise):
    """main() of the reference for len(noise) passes of its loop; noise [steps,4] = the four N(0,1) draws of each pass."""
    w = np.ascontiguousarray(noise, dtype=np.float64)
    s = w.shape[0]
    o = dict(hxTrue=np.zeros((s, 4), np.float32), hxDR=np.zeros((s, 4), np.float32), hxEst=np.zeros((s, 4), np.float32),
             hz=np.zeros((s, 2), np.float32), hud=np.zeros((s, 2), np.float32), PEst=np.zeros(16, np.float32),
             Q=np.zeros(16, np.float32), R=np.zeros(4, np.float32), Qsim=np.zeros(4, np.float32), Rsim=np.zeros(4, np.float32))
    lib().ref_ekf_main(_I(s), _p(w), *[_p(o[k]) for k in ("hxTrue", "hxDR", "hxEst", "hz", "hud", "PEst", "Q", "R", "Qsim", "Rsim")])
    return o


# ---- LQR (src/lqr_speed_steer_control.cpp, src/lqr_steer_control.cpp) ------------------------------------------------------
def dare(A, B, Q, R):
    """solve_DARE and dlqr of the file matching the dimension; returns (X, K)."""
    A, B, Q, R = _f32(A), _f32(B), _f32(Q), _f32(R)
    n = A.shape[0]
    dim = 5 if A.shape[1] == 25 else 4
    X = np.zeros((n, dim * dim), np.float32)
    K = np.zeros((n, (2 if dim == 5 else 1) * dim), np.float32)
    (lib().ref_dare5 if dim == 5 else lib().ref_dare4)(_I(n), _p(A), _p(B), _p(Q), _p(R), _p(X), _p(K))
    return X, K


def _course(course):
    return tuple(_f32(a) for a in course)


def lqr_steering_control(state, course, pe, pth_e, dim=5, ind=None):
    state = _f32(state)
    n = len(state)
    cx, cy, cyaw, ck, sp = _course(course)
    pe, pth_e = _f32(pe).copy(), _f32(pth_e).copy()
    if dim == 5:
        control = np.zeros((n, 2), np.float32)
        lib().ref_lqr5_steering_control(_I(n), _p(state), _I(len(cx)), _p(cx), _p(cy), _p(cyaw), _p(ck), _p(sp), _p(pe), _p(pth_e), _p(control))
        return control, None, pe, pth_e
    ind = np.zeros(n, np.int32) if ind is None else np.ascontiguousarray(ind, np.int32).copy()
    delta = np.zeros(n, np.float32)
    lib().ref_lqr4_steering_control(_I(n), _p(state), _I(len(cx)), _p(cx), _p(cy), _p(cyaw), _p(ck), _p(ind), _p(pe), _p(pth_e), _p(delta))
    return delta, ind, pe, pth_e


def calc_nearest_index(state, course):
    state = _f32(state)
    n = len(state)
    cx, cy, cyaw, ck, sp = _course(course)
    ind = np.zeros(n, np.int32); e = np.zeros(n, np.float32)
    lib().ref_lqr5_nearest_index(_I(n), _p(state), _I(len(cx)), _p(cx), _p(cy), _p(cyaw), _p(ind), _p(e))
    return ind, e


def lqr_update(state, a, delta):
    state, a, delta = _f32(state).copy(), _f32(a), _f32(delta)
    lib().ref_lqr5_update(_I(len(state)), _p(state), _p(a), _p(delta))
    return state


def lqr_closed_loop(state, course, goal, dim=5, max_ticks=500):
    state = _f32(state).copy()
    n = len(state)
    cx, cy, cyaw, ck, sp = _course(course)
    traj = np.zeros((max_ticks, n, 4), np.float32)
    ticks = np.zeros(n, np.int32)
    f = lib().ref_lqr5_closed_loop if dim == 5 else lib().ref_lqr4_closed_loop
    f(_I(n), _I(max_ticks), _p(state), _I(len(cx)), _p(cx), _p(cy), _p(cyaw), _p(ck), _p(sp), _F(goal[0]), _F(goal[1]), _p(traj), _p(ticks))
    return state, ticks, traj


# ---- MPC (src/model_predictive_control.cpp) ------------------------------------------------------------------------------
def _mpc(T, name):
    assert T in (6, 21), "libref is built for the reference's horizon macro T = 6 and for T = 21"
    return getattr(lib(), f"ref_mpc{T}_{name}")


def mpc_layout(T):
    out = np.zeros(6, np.int32)
    _mpc(T, "layout")(_p(out))
    return dict(zip(("x", "y", "yaw", "v", "delta", "a"), out.tolist()))


def mpc_fg_eval(xref, vars_, T):
    """FG_EVAL::operator(): fg[0] = cost, fg[1:] = the 4T constraint functions, at the point vars_ (reference layout)."""
    xref = _f32(xref); v = np.ascontiguousarray(vars_, np.float64)
    fg = np.zeros(1 + 4 * T, np.float64)
    _mpc(T, "fg_eval")(_p(xref), _p(v), _p(fg))
    return fg


def mpc_solve(x0, xref, T, solver=None):
    """mpc_solve(): returns what it hands to IPOPT and what it returns.  solver: python callable
    (x0[4] float64, xref[4T] float32) -> sol[n_vars] float64, standing in for IPOPT; None -> zeros."""
    x0, xref = _f32(x0), _f32(xref)
    nv, ng = 4 * T + 2 * (T - 1), 4 * T
    o = dict(xi=np.zeros(nv), xl=np.zeros(nv), xu=np.zeros(nv), gl=np.zeros(ng), gu=np.zeros(ng), result=np.zeros(nv, np.float32))
    opts = C.create_string_buffer(512)
    cb = _wrap_solver(T, solver)
    _mpc(T, "solve")(_p(x0), _p(xref), cb, _p(o["xi"]), _p(o["xl"]), _p(o["xu"]), _p(o["gl"]), _p(o["gu"]), _p(o["result"]), opts, _I(512))
    o["options"] = opts.value.decode()
    return o


def _wrap_solver(T, solver):
    if solver is None:
        return C.cast(None, SOLVER_FN)
    lay = mpc_layout(T)

    def cb(n_vars, n_con, xi, xl, xu, gl, gu, ctx, x_out):
        xref = np.zeros(4 * T, np.float32)
        _mpc(T, "context_xref")(C.c_void_p(ctx), _p(xref))
        x0 = np.array([gl[lay["x"]], gl[lay["y"]], gl[lay["yaw"]], gl[lay["v"]]], np.float64)
        sol = np.asarray(solver(x0, xref), np.float64)
        for i in range(n_vars):
            x_out[i] = sol[i]
    return SOLVER_FN(cb)


def mpc_update(state, a, delta, T=6):
    state, a, delta = _f32(state).copy(), _f32(a), _f32(delta)
    _mpc(T, "update")(_I(len(state)), _p(state), _p(a), _p(delta))
    return state


def calc_nearest_index_window(state, course, pind, T=6):
    state = _f32(state)
    cx, cy, cyaw, ck, sp = _course(course)
    pind = np.ascontiguousarray(pind, np.int32); out = np.zeros(len(state), np.int32)
    _mpc(T, "nearest_index_window")(_I(len(state)), _p(state), _I(len(cx)), _p(cx), _p(cy), _p(cyaw), _p(pind), _p(out))
    return out


def calc_ref_trajectory(state, course, target_ind, T, dl=1.0):
    state = _f32(state)
    cx, cy, cyaw, ck, sp = _course(course)
    tind = np.ascontiguousarray(target_ind, np.int32).copy()
    xref = np.zeros((len(state), 4 * T), np.float32)
    _mpc(T, "calc_ref_trajectory")(_I(len(state)), _p(state), _I(len(cx)), _p(cx), _p(cy), _p(cyaw), _p(ck), _p(sp), _F(dl), _p(tind), _p(xref))
    return xref, tind


def calc_speed_profile(which, rx, ry, ryaw, target_speed):
    """which: 5 = lqr_speed_steer_control.cpp:40-63, 4 = lqr_steer_control.cpp:35-52, 0 = model_predictive_control.cpp:83-105."""
    rx, ry, ryaw = _f32(rx), _f32(ry), _f32(ryaw)
    out = np.zeros(len(ryaw), np.float32)
    lib().ref_calc_speed_profile(_I(int(which)), _I(len(ryaw)), _p(rx), _p(ry), _p(ryaw), _F(target_speed), _p(out))
    return out


def smooth_yaw(cyaw, T=6):
    c = _f32(cyaw).copy()
    _mpc(T, "smooth_yaw")(_I(len(c)), _p(c))
    return c


def mpc_simulation(course, goal, T, max_ticks, solver):
    cx, cy, cyaw, ck, sp = _course(course)
    traj = np.zeros((max_ticks, 4), np.float32); ctl = np.zeros((max_ticks, 2), np.float32); cs = np.zeros(len(cx), np.float32)
    f = _mpc(T, "simulation"); f.restype = _I
    cb = _wrap_solver(T, solver)
    ticks = f(_I(max_ticks), _I(len(cx)), _p(cx), _p(cy), _p(cyaw), _p(ck), _p(sp), _F(goal[0]), _F(goal[1]), cb, _p(traj), _p(ctl), _p(cs))
    return ticks, traj[:ticks], ctl[:ticks], cs


# ---- dynamic window (src/dynamic_window_approach.cpp) and Frenet planner (src/frenet_optimal_trajectory.cpp) -------------------
def dwa_config():
    out = np.zeros(12, np.float32)
    f = lib().ref_dwa_config_floats
    f.restype = _I
    assert f(_p(out)) == 12
    return out


def dwa_run(state, u, goal, ob, max_ticks, cfg=None):
    state, u, goal, ob = _f32(state).copy(), _f32(u).copy(), _f32(goal), _f32(ob)
    cfg = dwa_config() if cfg is None else _f32(cfg)
    n = len(state)
    traj = np.zeros((max_ticks, n, 5), np.float32)
    ticks = np.zeros(n, np.int32)
    lib().ref_dwa_run(_I(n), _I(max_ticks), _p(state), _p(u), _p(goal), _p(ob), _I(len(ob)), _p(cfg), _p(traj), _p(ticks))
    return state, u, ticks, traj


def frenet_spline_build(wx, wy):
    wx, wy = _f32(wx), _f32(wy)
    coef = np.zeros((9, len(wx)), np.float32)
    lib().ref_frenet_spline_build(_p(wx), _p(wy), _I(len(wx)), _p(coef))
    return coef


def quintic(args7):
    a = _f32(args7)
    out = np.zeros((len(a), 3), np.float32)
    lib().ref_quintic(_I(len(a)), _p(a), _p(out))
    return out


def quartic(args6):
    a = _f32(args6)
    out = np.zeros((len(a), 2), np.float32)
    lib().ref_quartic(_I(len(a)), _p(a), _p(out))
    return out


def frenet_run(state, wx, wy, goal, ob, max_ticks, want_paths=False, cap=4096):
    state, wx, wy, goal, ob = _f32(state).copy(), _f32(wx), _f32(wy), _f32(goal), _f32(ob)
    n = len(state)
    hist = np.zeros((max_ticks, n, 8), np.float32)
    ticks = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
    pcf = np.zeros((n, cap), np.float32) if want_paths else None
    pok = np.zeros((n, cap), np.int32) if want_paths else None
    npth = np.zeros(n, np.int32) if want_paths else None
    lib().ref_frenet_run(_I(n), _I(max_ticks), _p(state), _p(wx), _p(wy), _I(len(wx)), _p(goal), _p(ob), _I(len(ob)), _p(hist), _p(ticks),
                         _p(status), _p(pcf), _p(pok), _p(npth), _I(cap))
    return dict(state=state, hist=hist, ticks=ticks, status=status, path_cf=pcf, path_ok=pok, n_paths=npth)


def main_course(wx, wy, which="lqr"):
    """The sampling loops of the reference's mains: lqr_speed_steer_control.cpp:252-265 (ds 0.1) / model_predictive_control.cpp:473-486 (ds 1.0)."""
    wx, wy = _f32(wx), _f32(wy)
    cap = 100000
    out = [np.zeros(cap, np.float32) for _ in range(4)]
    f = lib().ref_lqr5_main_course if which == "lqr" else lib().ref_mpc6_main_course
    f.restype = _I
    k = f(_p(wx), _p(wy), _I(len(wx)), *[_p(a) for a in out], _I(cap))
    return tuple(a[:k] for a in out)


# ---- particle filter (src/particle_filter.cpp:25-148 and the loop of main) --------------------------------------------------------
def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def pf_np():
    lib().ref_pf_np.restype = _I
    return lib().ref_pf_np()


def pf_motion_model(x, u):
    x, u = _f32(x), _f32(u)
    out = np.zeros_like(x)
    lib().ref_pf_motion_model(_I(len(x)), _p(x), _p(u), _p(out))
    return out


def pf_gauss_likelihood(x, sigma):
    x, sigma = _f32(x), _f32(sigma)
    out = np.zeros_like(x)
    lib().ref_pf_gauss_likelihood(_I(len(x)), _p(x), _p(sigma), _p(out))
    return out


def pf_calc_covariance(xEst, px, pw):
    out = np.zeros(16, np.float32)
    lib().ref_pf_calc_covariance(_p(_f32(xEst)), _p(_f32(px)), _p(_f32(pw)), _p(out))
    return out


def pf_cumsum(pw):
    out = np.zeros(pf_np(), np.float32)
    lib().ref_pf_cumsum(_p(_f32(pw)), _p(out))
    return out


def pf_localization(px, pw, z, u, nrm, rsim=(1.0, 0.0, 0.0, 1.0), Q=0.01):
    """One pf_localization call for one vehicle.  px [NP,4], pw [NP], z [nz,3], u [2], nrm [NP,2] (the normal draws, in the order
    the function consumes them), rsim = Rsim column-major.  -> px, pw, xEst, PEst."""
    px, pw, z = _f32(px).copy(), _f32(pw).copy(), _f32(z).reshape(-1, 3)
    xEst, PEst = np.zeros(4, np.float32), np.zeros(16, np.float32)
    lib().ref_pf_localization(_p(px), _p(pw), _p(xEst), _p(PEst), _p(z), _I(len(z)), _p(_f32(u)), _p(_f32(rsim)), _F(Q), _p(_f64(nrm)))
    return px, pw, xEst, PEst


def pf_resampling(px, pw, uni):
    px, pw = _f32(px).copy(), _f32(pw).copy()
    lib().ref_pf_resampling(_p(px), _p(pw), _p(_f64(uni)))
    return px, pw


def pf_main(steps, w, uni):
    """main() :179-235 and `steps` passes of its loop :249-271 on the caller's streams (see oracle/ref_shim/ref_pf.cpp)."""
    NP = pf_np()
    o = dict(ud=np.zeros((steps, 2), np.float32), xTrue=np.zeros((steps, 4), np.float32), xDR=np.zeros((steps, 4), np.float32),
             z=np.zeros((steps, 4, 3), np.float32), nz=np.zeros(steps, np.int32), xEst=np.zeros((steps, 4), np.float32),
             PEst=np.zeros((steps, 16), np.float32), px=np.zeros((steps, NP, 4), np.float32), pw=np.zeros((steps, NP), np.float32),
             consts=np.zeros(6, np.float32))
    f = lib().ref_pf_main
    f.restype = C.c_long
    o["draws_used"] = f(_I(steps), _p(_f64(w)), _p(_f64(uni)), *[_p(o[k]) for k in ("ud", "xTrue", "xDR", "z", "nz", "xEst", "PEst", "px", "pw", "consts")])
    return o

"""Host-side mirror of the reference's course-tracking functions, batched over n agents on one shared course.

  calc_nearest_index / lqr_steering_control / update / closed_loop_prediction
      /root/reference/src/lqr_speed_steer_control.cpp:65-83, 108-151, 154-164, 166-205   (dim 5)
      /root/reference/src/lqr_steer_control.cpp:55-73, 98-133, 136-146, 146-197          (dim 4)
  calc_nearest_index (window) / calc_ref_trajectory / update / mpc_simulation
      /root/reference/src/model_predictive_control.cpp:107-127, 130-170, 69-81, 348-385

state: float32 CUDA tensor [n,4] = (x, y, yaw, v).  course: `Course(cx, cy, cyaw, ck, sp)` of float32 CUDA tensors.
"""
import ctypes as C
import math

from . import _lib as L


class Course:
    """The reference's (cx, cy, cyaw, ck, speed_profile) vectors, resident on the GPU."""

    def __init__(self, cx, cy, cyaw, ck, sp):
        L.require_cuda(cx, cy, cyaw, ck, sp)
        n = cx.shape[0]
        for name, t in (("cx", cx), ("cy", cy), ("cyaw", cyaw), ("ck", ck), ("sp", sp)):
            L.expect(name, t, "f", n)
        self.tensors = (cx, cy, cyaw, ck, sp)
        self.n = n
        self.c = L.Course(n, cx.data_ptr(), cy.data_ptr(), cyaw.data_ptr(), ck.data_ptr(), sp.data_ptr())

    @classmethod
    def from_numpy(cls, arrays, device="cuda"):
        import numpy as np
        import torch
        return cls(*(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device) for a in arrays))

    def ref(self):
        return C.byref(self.c)


def course_from_waypoints(wx, wy, ds, target_speed=10.0 / 3.6, variant=5):
    """The course arrays (cx, cy, cyaw, ck, speed_profile) the reference's mains build: Spline2D(wx, wy) sampled every ds
    (lqr files ds = 0.1, MPC ds = 1.0) and calc_speed_profile (variant 5 = lqr_speed_steer_control.cpp, 4 = lqr_steer_control.cpp,
    0 = MPC).  numpy, host."""
    import numpy as np
    wx = np.ascontiguousarray(wx, dtype=np.float32); wy = np.ascontiguousarray(wy, dtype=np.float32)
    l = L.lib()
    k = l.crx_course_from_waypoints(wx.ctypes.data, wy.ctypes.data, len(wx), float(ds), None, None, None, None, 0)
    if k < 0:
        L.check(k, "crx_course_from_waypoints")
    cx, cy, cyaw, ck, sp = (np.zeros(k, np.float32) for _ in range(5))
    l.crx_course_from_waypoints(wx.ctypes.data, wy.ctypes.data, len(wx), float(ds), cx.ctypes.data, cy.ctypes.data, cyaw.ctypes.data, ck.ctypes.data, k)
    L.check(l.crx_calc_speed_profile(int(variant), cx.ctypes.data, cy.ctypes.data, cyaw.ctypes.data, k, float(target_speed), sp.ctypes.data),
            "crx_calc_speed_profile")
    return cx, cy, cyaw, ck, sp


def calc_speed_profile(variant, cx, cy, cyaw, target_speed=10.0 / 3.6):
    """calc_speed_profile of the reference's tracking files: variant 5 = lqr_speed_steer_control.cpp:40-62, 4 = lqr_steer_control.cpp:35-52,
    0 = model_predictive_control.cpp:83-105.  numpy, host."""
    import numpy as np
    cx, cy, cyaw = (np.ascontiguousarray(a, dtype=np.float32) for a in (cx, cy, cyaw))
    sp = np.zeros(len(cyaw), np.float32)
    L.check(L.lib().crx_calc_speed_profile(int(variant), cx.ctypes.data, cy.ctypes.data, cyaw.ctypes.data, len(cyaw), float(target_speed),
                                           sp.ctypes.data), "crx_calc_speed_profile")
    return sp


def smooth_yaw(cyaw):
    """smooth_yaw of the reference's MPC file (:172-185): the course headings made continuous, as mpc_simulation does before
    its loop (:360).  numpy, host; returns a new array."""
    import numpy as np
    c = np.array(cyaw, dtype=np.float32, copy=True, order="C")
    L.check(L.lib().crx_smooth_yaw(c.ctypes.data, len(c)), "crx_smooth_yaw")
    return c


def _lqr_params(dt, Lw, eps, maxiter):
    p = L.LqrParams()
    p.dt, p.L, p.eps, p.maxiter = float(dt), float(Lw), float(eps), int(maxiter)
    return p


def vehicle_params(mpc=False, **over):
    p = L.VehicleParams()
    L.lib().crx_vehicle_default_params(C.byref(p), 1 if mpc else 0)
    for k, v in over.items():
        setattr(p, k, v)
    return p


def calc_nearest_index(state, course, ind=None, e=None):
    """-> (ind int32 [n], e float32 [n]); ind (and e) are written in place when given."""
    import torch
    L.require_cuda(state, ind, e)
    n = state.shape[0]
    L.expect("state", state, "f", n, 4); L.expect("ind", ind, "i", n, optional=True); L.expect("e", e, "f", n, optional=True)
    if ind is None:
        ind = torch.zeros((n,), dtype=torch.int32, device=state.device)
    if e is None:
        e = torch.empty((n,), dtype=torch.float32, device=state.device)
    L.check(L.lib().crx_calc_nearest_index_batch_dev(n, L.ptr(state), course.ref(), L.ptr(ind), L.ptr(e), L.stream_ptr()),
            "crx_calc_nearest_index_batch_dev")
    return ind, e


def lqr_steering_control(state, course, pe, pth_e, dim=5, ind=None, dt=0.1, L_wheelbase=0.5, eps=0.01, maxiter=150):
    """pe, pth_e (and ind for dim 4) are updated in place, like the reference's reference parameters.
    dim 5 -> control [n,2] = {ai, delta}; dim 4 -> delta [n]."""
    import torch
    L.require_cuda(state, pe, pth_e, ind)
    n = state.shape[0]
    L.expect("state", state, "f", n, 4); L.expect("pe", pe, "f", n); L.expect("pth_e", pth_e, "f", n); L.expect("ind", ind, "i", n, optional=True)
    if ind is None:
        ind = torch.zeros((n,), dtype=torch.int32, device=state.device)
    control = torch.empty((n, 2) if dim == 5 else (n,), dtype=torch.float32, device=state.device)
    p = _lqr_params(dt, L_wheelbase, eps, maxiter)
    L.check(L.lib().crx_lqr_steering_control_batch_dev(n, dim, L.ptr(state), course.ref(), L.ptr(ind), L.ptr(pe), L.ptr(pth_e),
                                                       C.byref(p), L.ptr(control), L.stream_ptr()),
            "crx_lqr_steering_control_batch_dev")
    return control, ind


def update(state, a, delta, params=None):
    """update(state, a, delta), in place."""
    L.require_cuda(state, a, delta)
    n = state.shape[0]
    L.expect("state", state, "f", n, 4); L.expect("a", a, "f", n); L.expect("delta", delta, "f", n)
    p = params if params is not None else vehicle_params()
    L.check(L.lib().crx_update_batch_dev(state.shape[0], L.ptr(state), L.ptr(a), L.ptr(delta), C.byref(p), L.stream_ptr()),
            "crx_update_batch_dev")
    return state


def loop_params(goal, goal_dis, max_ticks, kp=1.0, stop_speed=0.05):
    p = L.LoopParams()
    p.goal_x, p.goal_y, p.goal_dis, p.kp, p.stop_speed, p.max_ticks = float(goal[0]), float(goal[1]), float(goal_dis), float(kp), \
        float(stop_speed), int(max_ticks)
    return p


def closed_loop_prediction(state, course, goal, dim=5, max_ticks=500, goal_dis=None, dt=0.1, L_wheelbase=0.5, eps=0.01,
                           maxiter=150, kp=1.0, stop_speed=0.05, want_hist=False, pe=None, pth_e=None, ind=None):
    """The reference's closed_loop_prediction (maths only) for n agents in one kernel; state is updated in place.
    -> (ticks_done int32 [n], traj_hist [max_ticks,n,4] or None)."""
    import torch
    L.require_cuda(state, pe, pth_e, ind)
    n = state.shape[0]
    L.expect("state", state, "f", n, 4); L.expect("pe", pe, "f", n, optional=True); L.expect("pth_e", pth_e, "f", n, optional=True)
    L.expect("ind", ind, "i", n, optional=True)
    if goal_dis is None:
        goal_dis = 0.3 if dim == 5 else 0.5
    ticks = torch.zeros((n,), dtype=torch.int32, device=state.device)
    hist = torch.zeros((max_ticks, n, 4), dtype=torch.float32, device=state.device) if want_hist else None
    p = _lqr_params(dt, L_wheelbase, eps, maxiter)
    vp = vehicle_params(False, dt=float(dt), wheelbase=float(L_wheelbase))
    lp = loop_params(goal, goal_dis, max_ticks, kp, stop_speed)
    L.check(L.lib().crx_lqr_closed_loop_batch_dev(n, dim, L.ptr(state), course.ref(), L.ptr(pe), L.ptr(pth_e), L.ptr(ind),
                                                  C.byref(p), C.byref(vp), C.byref(lp), L.ptr(hist), L.ptr(ticks), L.stream_ptr()),
            "crx_lqr_closed_loop_batch_dev")
    return ticks, hist


def calc_nearest_index_window(state, course, pind, nsearch=10):
    import torch
    L.require_cuda(state, pind)
    n = state.shape[0]
    L.expect("state", state, "f", n, 4); L.expect("pind", pind, "i", n)
    out = torch.empty((n,), dtype=torch.int32, device=state.device)
    L.check(L.lib().crx_calc_nearest_index_window_batch_dev(n, L.ptr(state), course.ref(), L.ptr(pind), int(nsearch), L.ptr(out),
                                                            L.stream_ptr()), "crx_calc_nearest_index_window_batch_dev")
    return out


def calc_ref_trajectory(state, course, target_ind, T, dl=1.0, dt=0.2, nsearch=10, out=None):
    """-> xref [n,4T] (column-major 4xT per agent; `out` when given); target_ind (int32 [n]) is updated in place."""
    import torch
    L.require_cuda(state, target_ind, out)
    n = state.shape[0]
    L.expect("state", state, "f", n, 4); L.expect("target_ind", target_ind, "i", n); L.expect("out", out, "f", n, 4 * T, optional=True)
    xref = torch.empty((n, 4 * T), dtype=torch.float32, device=state.device) if out is None else out
    L.check(L.lib().crx_calc_ref_trajectory_batch_dev(n, T, L.ptr(state), course.ref(), float(dl), float(dt), int(nsearch),
                                                      L.ptr(target_ind), L.ptr(xref), L.stream_ptr()),
            "crx_calc_ref_trajectory_batch_dev")
    return xref


def mpc_simulation(state, course, goal, T, max_ticks, target_ind=None, dl=1.0, nsearch=10, goal_dis=0.5, params=None,
                   want_hist=False, want_flags=False):
    """mpc_simulation's loop (:371-385, maths only) for n vehicles at once: state and target_ind are updated in place.
    -> (ticks_done int32 [n], traj_hist or None[, solve_flags int32 [n]: bit 0 = a tick's solve did not converge, bit 1 = a tick
    started outside the speed bounds]).
    The set-up of the reference (:349-360) is the caller's, from the pieces of this module: the course arrays from
    course_from_waypoints(wx, wy, 1.0, variant=0), the start state (cx[0], cy[0], cyaw[0], sp[0]) taken BEFORE the headings are
    passed through smooth_yaw (:360), target_ind = 0."""
    import torch
    from .mpc import default_params
    L.require_cuda(state, target_ind)
    n = state.shape[0]
    L.expect("state", state, "f", n, 4); L.expect("target_ind", target_ind, "i", n, optional=True)
    if target_ind is None:
        target_ind = torch.zeros((n,), dtype=torch.int32, device=state.device)
    p = params if params is not None else default_params()
    ticks = torch.zeros((n,), dtype=torch.int32, device=state.device)
    hist = torch.zeros((max_ticks, n, 4), dtype=torch.float32, device=state.device) if want_hist else None
    lp = loop_params(goal, goal_dis, max_ticks)
    flags = torch.zeros((n,), dtype=torch.int32, device=state.device) if want_flags else None
    L.check(L.lib().crx_mpc_closed_loop_flags_batch_dev(n, T, L.ptr(state), course.ref(), float(dl), int(nsearch), C.byref(p),
                                                        C.byref(lp), L.ptr(target_ind), L.ptr(hist), L.ptr(ticks), L.ptr(flags),
                                                        L.stream_ptr()), "crx_mpc_closed_loop_flags_batch_dev")
    return (ticks, hist, flags) if want_flags else (ticks, hist)

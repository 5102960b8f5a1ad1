"""Host-side mirror of the reference's Frenet optimal-trajectory planner, batched over n agents (one agent per wavefront).

/root/reference/src/frenet_optimal_trajectory.cpp: frenet_optimal_planning :160-176 and the main loop :224-236;
/root/reference/include/cubic_spline.h: Spline2D :130-187 (built once per course on the host).
state [n,5] = (s0, c_speed, c_d, c_d_d, c_d_dd), ob [nob,2] — float32 CUDA tensors; the course is a `FrenetCourse`.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def frenet_default_config():
    c = L.FrenetConfig()
    L.lib().crx_frenet_default_config(C.byref(c))
    return c


def frenet_num_paths(config=None):
    c = config if config is not None else frenet_default_config()
    k = L.lib().crx_frenet_num_paths(C.byref(c))
    if k < 0:
        L.check(k, "crx_frenet_num_paths")
    return k


class FrenetCourse:
    """Spline2D csp_obj(wx, wy) (:203) plus the sampled course r_x, r_y (:205-213) whose last point is the goal."""

    def __init__(self, wx, wy, device=None):
        wx = np.ascontiguousarray(wx, dtype=np.float32)
        wy = np.ascontiguousarray(wy, dtype=np.float32)
        self.nx = int(wx.shape[0])
        self.coef = np.zeros((9, self.nx), np.float32)
        l = L.lib()
        L.check(l.crx_frenet_spline_build(wx.ctypes.data, wy.ctypes.data, self.nx, self.coef.ctypes.data), "crx_frenet_spline_build")
        k = l.crx_frenet_course_samples(self.coef.ctypes.data, self.nx, None, None, 0)
        if k < 0:
            L.check(k, "crx_frenet_course_samples")
        self.rx, self.ry = np.zeros(k, np.float32), np.zeros(k, np.float32)
        l.crx_frenet_course_samples(self.coef.ctypes.data, self.nx, self.rx.ctypes.data, self.ry.ctypes.data, k)
        self.goal = np.array([self.rx[-1], self.ry[-1]], np.float32)
        self._dev = None
        if device is not None:
            self.to(device)

    def to(self, device):
        import torch
        self._dev = torch.from_numpy(self.coef).to(device)
        return self

    def dev(self, device):
        if self._dev is None or self._dev.device != device:
            self.to(device)
        return self._dev


def frenet_run(state, course, ob, max_ticks, config=None, want_hist=False, want_paths=False):
    """state is updated in place.  -> dict(ticks, status, best_idx, n_valid [int32 n], hist [max_ticks,n,8] or None,
    path_cf / path_ok [n,P] or None)."""
    import torch
    L.require_cuda(state, ob)
    n = state.shape[0]
    L.expect("state", state, "f", n, 5); L.expect("ob", ob, "f", None, 2)
    c = config if config is not None else frenet_default_config()
    i32 = lambda: torch.zeros((n,), dtype=torch.int32, device=state.device)
    ticks, status, best, nv = i32(), i32(), i32(), i32()
    hist = torch.zeros((max_ticks, n, 8), dtype=torch.float32, device=state.device) if want_hist else None
    P = frenet_num_paths(c) if want_paths else 0
    pcf = torch.zeros((n, P), dtype=torch.float32, device=state.device) if want_paths else None
    pok = torch.zeros((n, P), dtype=torch.int32, device=state.device) if want_paths else None
    coef = course.dev(state.device)
    goal = np.ascontiguousarray(course.goal, dtype=np.float32)
    L.check(L.lib().crx_frenet_run_batch_dev(n, int(max_ticks), L.ptr(state), L.ptr(coef), course.nx, goal.ctypes.data, L.ptr(ob),
                                             ob.shape[0], C.byref(c), L.ptr(hist), L.ptr(ticks), L.ptr(status), L.ptr(best), L.ptr(nv),
                                             L.ptr(pcf), L.ptr(pok), P, L.stream_ptr()), "crx_frenet_run_batch_dev")
    return dict(ticks=ticks, status=status, best_idx=best, n_valid=nv, hist=hist, path_cf=pcf, path_ok=pok)


def frenet_optimal_planning(state, course, ob, config=None, want_paths=False):
    """One planning call + state hand-over (the body of the reference's loop)."""
    return frenet_run(state, course, ob, 1, config, want_hist=True, want_paths=want_paths)

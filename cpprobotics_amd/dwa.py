"""Host-side mirror of the reference's dynamic-window planner, batched over n agents (one agent per wavefront).

/root/reference/src/dynamic_window_approach.cpp: dwa_control :148-155 and the main loop :192-194 / goal test :225.
state [n,5] = (x, y, yaw, v, yawrate), u [n,2], goal [n,2], ob [nob,2] — float32 CUDA tensors.
"""
import ctypes as C

from . import _lib as L


def dwa_default_config():
    c = L.DwaConfig()
    L.lib().crx_dwa_default_config(C.byref(c))
    return c


def dwa_run(state, u, goal, ob, max_ticks, config=None, want_hist=False):
    """state and u are updated in place.  -> (ticks_done, traj_hist or None, status, best_idx, n_samples) (int32 [n])."""
    import torch
    L.require_cuda(state, u, goal, ob)
    n = state.shape[0]
    L.expect("state", state, "f", n, 5); L.expect("u", u, "f", n, 2); L.expect("goal", goal, "f", n, 2); L.expect("ob", ob, "f", None, 2)
    i32 = lambda: torch.zeros((n,), dtype=torch.int32, device=state.device)
    ticks, status, best, ns = i32(), i32(), i32(), i32()
    hist = torch.zeros((max_ticks, n, 5), dtype=torch.float32, device=state.device) if want_hist else None
    c = config if config is not None else dwa_default_config()
    L.check(L.lib().crx_dwa_run_batch_dev(n, int(max_ticks), L.ptr(state), L.ptr(u), L.ptr(goal), L.ptr(ob), ob.shape[0], C.byref(c),
                                          L.ptr(hist), L.ptr(ticks), L.ptr(status), L.ptr(best), L.ptr(ns), L.stream_ptr()),
            "crx_dwa_run_batch_dev")
    return ticks, hist, status, best, ns


def dwa_control(state, u, goal, ob, config=None):
    """One dwa_control + motion step (the body of the reference's loop)."""
    return dwa_run(state, u, goal, ob, 1, config)

"""Host-side mirror of the reference's EKF free functions, batched over n vehicles.

Names and argument meaning follow /root/reference/src/extended_kalman_filter.cpp:22-78; every
argument that is an Eigen fixed-size object there is a torch float32 CUDA tensor with a leading
batch dimension here (column-major trailing block, i.e. ``P[k]`` holds ``Matrix4f::data()``).
Q and R are shared by all vehicles and may be numpy arrays, lists or CPU tensors.
"""
import ctypes as C
import math

import numpy as np

from . import _lib as L


def _params(dt):
    p = L.EkfParams()
    p.dt = float(dt)
    return p


def _qr(Q, R):
    def to_np(m, n):
        if hasattr(m, "detach"):
            m = m.detach().cpu().numpy()
        return L.host_floats(np.asarray(m, dtype=np.float32).reshape(-1), n)
    return to_np(Q, 16), to_np(R, 4)


def ekf_default_QR():
    """Q and R exactly as main() builds them (:142-151), as column-major float32 arrays."""
    Q = np.zeros((4, 4), dtype=np.float32)
    Q[0, 0] = np.float32(0.1 * 0.1)
    Q[1, 1] = np.float32(0.1 * 0.1)
    Q[2, 2] = np.float32((1.0 / 180 * math.pi) * (1.0 / 180 * math.pi))
    Q[3, 3] = np.float32(0.1 * 0.1)
    R = np.eye(2, dtype=np.float32)
    return Q.T.copy().reshape(-1), R.T.copy().reshape(-1)


def motion_model(x, u, dt=0.1, out=None):
    """motion_model(x, u) :22-36.  x: [n,4], u: [n,2] -> [n,4]."""
    import torch
    L.require_cuda(x, u, out)
    n = x.shape[0]
    L.expect("x", x, "f", n, 4); L.expect("u", u, "f", n, 2); L.expect("out", out, "f", n, 4, optional=True)
    out = torch.empty_like(x) if out is None else out
    p = _params(dt)
    L.check(L.lib().crx_motion_model_batch_dev(n, L.ptr(x), L.ptr(u), L.ptr(out), C.byref(p), L.stream_ptr()),
            "crx_motion_model_batch_dev")
    return out


def jacobF(x, u, dt=0.1):
    """jacobF(x, u) :38-47.  Returns [n,16] (column-major 4x4 per vehicle)."""
    import torch
    L.require_cuda(x, u)
    n = x.shape[0]
    L.expect("x", x, "f", n, 4); L.expect("u", u, "f", n, 2)
    out = torch.empty((n, 16), dtype=torch.float32, device=x.device)
    p = _params(dt)
    L.check(L.lib().crx_jacobF_batch_dev(n, L.ptr(x), L.ptr(u), L.ptr(out), C.byref(p), L.stream_ptr()),
            "crx_jacobF_batch_dev")
    return out


def observation_model(x):
    """observation_model(x) :50-55.  [n,4] -> [n,2]."""
    import torch
    L.require_cuda(x)
    n = x.shape[0]
    L.expect("x", x, "f", n, 4)
    out = torch.empty((n, 2), dtype=torch.float32, device=x.device)
    L.check(L.lib().crx_observation_model_batch_dev(n, L.ptr(x), L.ptr(out), L.stream_ptr()),
            "crx_observation_model_batch_dev")
    return out


def jacobH():
    """jacobH() :57-62 as a column-major float32 array of 8."""
    out = np.empty(8, dtype=np.float32)
    L.check(L.lib().crx_jacobH(out.ctypes.data_as(C.c_void_p)), "crx_jacobH")
    return out


def ekf_estimation(xEst, PEst, z, u, Q, R, dt=0.1):
    """ekf_estimation(xEst, PEst, z, u, Q, R) :64-78 — xEst [n,4] and PEst [n,16] updated IN PLACE."""
    L.require_cuda(xEst, PEst, z, u)
    n = xEst.shape[0]
    L.expect("xEst", xEst, "f", n, 4); L.expect("PEst", PEst, "f", n, 16); L.expect("z", z, "f", n, 2); L.expect("u", u, "f", n, 2)
    q, r = _qr(Q, R)
    p = _params(dt)
    L.check(L.lib().crx_ekf_step_batch_dev(n, L.ptr(xEst), L.ptr(PEst), L.ptr(z), L.ptr(u),
                                           q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                           C.byref(p), L.stream_ptr()), "crx_ekf_step_batch_dev")
    return xEst, PEst


def ekf_run(xEst, PEst, z, u, Q, R, dt=0.1, x_hist=None, P_hist=None):
    """T fused ekf_estimation() steps (the loop body :171-188).  z,u: [T,n,2]; x_hist [T,n,4] and
    P_hist [T,n,16] are optional outputs.  xEst/PEst updated in place."""
    L.require_cuda(xEst, PEst, z, u, x_hist, P_hist)
    T, n = z.shape[0], xEst.shape[0]
    L.expect("xEst", xEst, "f", n, 4); L.expect("PEst", PEst, "f", n, 16); L.expect("z", z, "f", T, n, 2); L.expect("u", u, "f", T, n, 2)
    L.expect("x_hist", x_hist, "f", T, n, 4, optional=True); L.expect("P_hist", P_hist, "f", T, n, 16, optional=True)
    q, r = _qr(Q, R)
    p = _params(dt)
    L.check(L.lib().crx_ekf_run_batch_dev(n, T, L.ptr(xEst), L.ptr(PEst), L.ptr(z), L.ptr(u), L.ptr(x_hist),
                                          L.ptr(P_hist), q.ctypes.data_as(C.c_void_p),
                                          r.ctypes.data_as(C.c_void_p), C.byref(p), L.stream_ptr()),
            "crx_ekf_run_batch_dev")
    return xEst, PEst


def normal_draws(n, T, agent0=0, seed=0xC0FFEE, stream_id=0, device=None, out=None):
    """[T,n,4] standard-normal draws keyed by (seed, stream_id, global agent id agent0 + a, step): the four draws each pass
    of the reference's loop consumes (:174-181).  Counter-based (Philox4x32-10), so a shard of a swarm gets exactly the
    draws the whole swarm would have had for its agents."""
    import torch
    if out is None:
        out = torch.empty((T, n, 4), dtype=torch.float32, device=device or "cuda")
    L.require_cuda(out)
    L.expect("out", out, "f", T, n, 4)
    L.check(L.lib().crx_normal_draws_dev(n, T, int(agent0), int(seed), int(stream_id), L.ptr(out), L.stream_ptr()),
            "crx_normal_draws_dev")
    return out


QSIM = (1.0, (30.0 / 180 * math.pi) * (30.0 / 180 * math.pi))   # Qsim diag (:154-156)
RSIM = (0.5 * 0.5, 0.5 * 0.5)                                    # Rsim diag (:159-161)


def ekf_simulate_inputs(u_true, xTrue, xDR, w, dt=0.1, qsim=QSIM, rsim=RSIM, xTrue_hist=None, xDR_hist=None):
    """Input side of the simulation loop (:174-181).  w: [T,n,4] standard normals.
    Returns z, ud ([T,n,2] each); xTrue/xDR ([n,4]) advance in place."""
    import torch
    L.require_cuda(u_true, xTrue, xDR, w, xTrue_hist, xDR_hist)
    T, n = w.shape[0], w.shape[1]
    L.expect("w", w, "f", T, n, 4); L.expect("u_true", u_true, "f", n, 2); L.expect("xTrue", xTrue, "f", n, 4); L.expect("xDR", xDR, "f", n, 4)
    L.expect("xTrue_hist", xTrue_hist, "f", T, n, 4, optional=True); L.expect("xDR_hist", xDR_hist, "f", T, n, 4, optional=True)
    z = torch.empty((T, n, 2), dtype=torch.float32, device=w.device)
    ud = torch.empty((T, n, 2), dtype=torch.float32, device=w.device)
    q = L.host_floats(qsim, 2)
    r = L.host_floats(rsim, 2)
    p = _params(dt)
    L.check(L.lib().crx_ekf_simulate_inputs_dev(n, T, L.ptr(u_true), L.ptr(xTrue), L.ptr(xDR), L.ptr(w), L.ptr(z),
                                                L.ptr(ud), L.ptr(xTrue_hist), L.ptr(xDR_hist),
                                                q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                                C.byref(p), L.stream_ptr()), "crx_ekf_simulate_inputs_dev")
    return z, ud

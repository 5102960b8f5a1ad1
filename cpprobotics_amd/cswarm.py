"""The C side of the swarm round and of the multi-GPU gather, bound for the tests and the bench: crx_swarm_* and crx_comm_* /
crx_allgather_dev of include/crx.h (csrc/api_swarm.inl).  A C++ host calls those directly (examples/ekf_fleet_mgpu.cpp,
INTEGRATION.md 8); cpprobotics_amd/swarm.py is the Python twin of the same round, and tests/test_swarm_gpu.py demands the same bytes
from both."""
import ctypes as C

from . import _lib as L


class CSwarm:
    """crx_swarm: one rank's shard of BASELINE.json configs[4], the round issued by ONE C call (crx_swarm_round_dev).
    x0 [n,4], P0 [n,16] device tensors; course: cpprobotics_amd.Course; Q, R as for ekf_run."""

    def __init__(self, x0, P0, course, Q, R, T, Tm=21, plan_every=8, depth=6, v_cmd=2.5, allow_shared_queues=False, params=None, streams=None):
        from .ekf import _qr
        L.require_cuda(x0, P0)
        n = x0.shape[0]
        L.expect("x0", x0, "f", n, 4); L.expect("P0", P0, "f", n, 16)
        cfg = L.SwarmConfig()
        L.lib().crx_swarm_default_config(C.byref(cfg))
        cfg.n, cfg.T, cfg.Tm, cfg.plan_every, cfg.depth, cfg.v_cmd = n, int(T), int(Tm), int(plan_every), int(depth), float(v_cmd)
        cfg.allow_shared_queues = 1 if allow_shared_queues else 0
        if params is not None:
            cfg.mpc = params
        if streams is not None:                   # torch streams of the caller's (e.g. swarm.planner_streams): not owned by the C object
            assert len(streams) >= int(depth)
            self._streams = streams
            arr = (C.c_void_p * int(depth))(*[s.cuda_stream for s in streams[:int(depth)]])
            cfg.planner_streams = C.cast(arr, C.POINTER(C.c_void_p))
        q, r = _qr(Q, R)
        self.course = course                       # the course arrays must outlive the object
        self.n, self.T, self.Tm, self.depth, self.device = n, int(T), int(Tm), int(depth), x0.device
        self.n_plan = (n + int(plan_every) - 1) // int(plan_every)
        self._h = C.c_void_p()
        L.check(L.lib().crx_swarm_create(C.byref(self._h), C.byref(cfg), course.ref(), L.ptr(x0), L.ptr(P0), q.ctypes.data_as(C.c_void_p),
                                         r.ctypes.data_as(C.c_void_p)), "crx_swarm_create")

    def round(self, z, u, x_hist=None):
        """Issue one round on the current stream; -> its index."""
        L.require_cuda(z, u, x_hist)
        L.expect("z", z, "f", self.T, self.n, 2); L.expect("u", u, "f", self.T, self.n, 2)
        L.expect("x_hist", x_hist, "f", self.T, self.n, 4, optional=True)
        rnd = C.c_longlong()
        L.check(L.lib().crx_swarm_round_dev(self._h, L.ptr(z), L.ptr(u), L.ptr(x_hist), L.stream_ptr(), C.byref(rnd)), "crx_swarm_round_dev")
        return rnd.value

    def wait(self):
        """The current stream waits for every planner in flight."""
        L.check(L.lib().crx_swarm_wait(self._h, L.stream_ptr()), "crx_swarm_wait")

    def plans(self, rnd):
        """-> dict of tensors COPIED out of round `rnd`'s slot (sol, status, cost, xref, est); call wait() + synchronize first."""
        import torch
        from .mpc import mpc_n_vars
        n_plan = C.c_int()
        ps = [C.c_void_p() for _ in range(5)]
        L.check(L.lib().crx_swarm_plans(self._h, int(rnd), C.byref(n_plan), *[C.byref(p) for p in ps]), "crx_swarm_plans")
        npl = n_plan.value
        shapes = (("sol", torch.float32, (npl, mpc_n_vars(self.Tm))), ("status", torch.int32, (npl,)), ("cost", torch.float64, (npl,)),
                  ("xref", torch.float32, (npl, 4 * self.Tm)), ("est", torch.float32, (npl, 4)))
        out = {}
        for (name, dt, shape), p in zip(shapes, ps):
            t = torch.empty(shape, dtype=dt, device=self.device)
            _d2d(t, p.value)
            out[name] = t
        return out

    def state(self):
        """[n,4]: a copy of the filter state after the most recent round's EKF launch."""
        import torch
        t = torch.empty((self.n, 4), dtype=torch.float32, device=self.device)
        _d2d(t, L.lib().crx_swarm_state(self._h))
        return t

    def close(self):
        if self._h:
            L.check(L.lib().crx_swarm_destroy(self._h), "crx_swarm_destroy")
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _d2d(dst, src_ptr):
    """Device-to-device copy from a raw pointer into a tensor, on the current stream (hipMemcpyAsync through the HIP runtime torch mapped)."""
    import torch
    hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipMemcpyAsync.restype = C.c_int
    rc = hip.hipMemcpyAsync(C.c_void_p(dst.data_ptr()), C.c_void_p(src_ptr), dst.numel() * dst.element_size(), 3, L.stream_ptr())
    if rc != 0:
        raise L.CrxError(f"hipMemcpyAsync failed: {rc}")


def hw_queues():
    return L.lib().crx_hw_queues()


class Comm:
    """crx_comm: an RCCL communicator behind the C ABI (one per process / GPU)."""

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        L.check(L.lib().crx_comm_unique_id(buf), "crx_comm_unique_id")
        return bytes(buf)

    def __init__(self, uid, rank, world):
        self._h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(uid)
        L.check(L.lib().crx_comm_init_rank(C.byref(self._h), buf, int(rank), int(world)), "crx_comm_init_rank")
        self.rank, self.world = L.lib().crx_comm_rank(self._h), L.lib().crx_comm_world(self._h)

    def allgather(self, send, recv=None):
        """recv[r] = rank r's `send` (any dtype, contiguous, on the device); on the current stream."""
        import torch
        assert send.is_cuda and send.is_contiguous()
        if recv is None:
            recv = torch.empty((self.world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        assert recv.is_contiguous() and recv.numel() == self.world * send.numel() and recv.dtype == send.dtype
        L.check(L.lib().crx_allgather_dev(self._h, L.ptr(send), L.ptr(recv), send.numel() * send.element_size(), L.stream_ptr()), "crx_allgather_dev")
        return recv

    def close(self):
        if self._h:
            L.check(L.lib().crx_comm_destroy(self._h), "crx_comm_destroy")
            self._h = C.c_void_p()

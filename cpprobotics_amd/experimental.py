"""Bindings of the measurement-only entry points (include/crx_experimental.h, prefix crx_x_): forced kernel variants and
launch geometries for A/B scripts and bench.py.  Not part of the drop-in surface; the product functions live in the other
modules of this package and choose their variant by themselves."""
import ctypes as C

from . import _lib as L

_P, _I = C.c_void_p, C.c_int
_X_SIGNATURES = {
    "crx_x_dare_from_v_lanes_dev": (_I, [_I, _I, _P, C.POINTER(L.LqrParams), _P, _P, _P, _P, _I]),
    "crx_x_mpc_solve_geometry_dev": (_I, [_I, _I, _P, _P, C.POINTER(L.MpcParams), _P, _P, _P, _P, _I, _I]),
    "crx_x_mpc_solve_trig_dev": (_I, [_I, _I, _P, _P, C.POINTER(L.MpcParams), _P, _P, _P, _P, _I]),
    "crx_x_mpc_solve_store_dev": (_I, [_I, _I, _P, _P, C.POINTER(L.MpcParams), _P, _P, _P, _P, _I]),
    "crx_x_mpc_solve_tile_refill_dev": (_I, [_I, _I, _P, _P, C.POINTER(L.MpcParams), _P, _P, _P, _P, _I, _I]),
    "crx_x_mpc_phased_work_bytes": (C.c_size_t, [_I, _I]),
    "crx_x_mpc_solve_phased_dev": (_I, [_I, _I, _P, _P, C.POINTER(L.MpcParams), _P, _P, _P, _P, _I, _P, _P]),
    "crx_x_mpc_solve_phased_store_dev": (_I, [_I, _I, _P, _P, C.POINTER(L.MpcParams), _P, _P, _P, _P, _I, _P, _P, _I]),
    "crx_x_mpc_solve_two_phase_dev": (_I, [_I, _I, _P, _P, C.POINTER(L.MpcParams), _P, _P, _P, _I, _P, _P, _P]),
    "crx_x_mpc_solve_store_refill_dev": (_I, [_I, _I, _P, _P, C.POINTER(L.MpcParams), _P, _P, _P, _P, _I, _I, _I]),
    "crx_x_mpc_solve_refill_dev": (_I, [_I, _I, _P, _P, C.POINTER(L.MpcParams), _P, _P, _P, _P, _I, _I]),
    "crx_x_mpc_solve_lanes_dev": (_I, [_I, _I, _P, _P, C.POINTER(L.MpcParams), _P, _P, _P, _P, _I]),
    "crx_x_lqr_closed_loop_lanes_dev": (_I, [_I, _I, _P, L._CP, _P, _P, _P, C.POINTER(L.LqrParams), C.POINTER(L.VehicleParams),
                                             C.POINTER(L.LoopParams), _P, _P, _P, _I]),
    "crx_x_dare_batch_dense_dev": (_I, [_I, _I, _P, _P, _P, _P, C.c_float, _I, _P, _P, _P, _P, _I]),
    "crx_x_dsincos_dev": (_I, [_I, _P, _P, _P, _P]),
    "crx_x_datan2_dev": (_I, [_I, _P, _P, _P]),
    "crx_x_datan2_sweep_dev": (_I, [C.c_double, _P, _P, _P, _P]),
    "crx_x_dare_from_v_refill_dev": (_I, [_I, _I, _P, C.POINTER(L.LqrParams), _P, _P, _P, _P, _I, _I]),
    "crx_x_hbm_stream_dev": (_I, [_I, _P, _P, C.c_size_t, _I, _P]),
    "crx_x_fetch_units_dev": (_I, [_I, _P, C.c_size_t, _P, _I, _I, _P]),
    "crx_x_recip_sweep_dev": (_I, [_P, _P]),
    "crx_x_ekf_run_addr64_dev": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(L.EkfParams), _P]),
    "crx_x_ekf_run_contracted_dev": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, C.POINTER(L.EkfParams), _P]),
    "crx_x_ekf_run_pair_batch_dev": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, C.POINTER(L.EkfParams), _P, _P]),
}
EXPERIMENTAL_SYMBOLS = tuple(sorted(_X_SIGNATURES))
_bound = False
_ab = None


def _bind(l, what):
    for name, (res, args) in _X_SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise L.CrxError(f"{what} does not export {name}") from e
        fn.restype, fn.argtypes = res, args


def xlib():
    """The product library with the crx_x_ entry points bound (forced variants of PRODUCT kernels)."""
    global _bound
    l = L.lib()
    if not _bound:
        _bind(l, "libcrx.so")
        _bound = True
    return l


def ab_lib_path():
    import os
    return os.environ.get("CRX_AB_LIB_PATH") or os.path.join(os.path.dirname(L.lib_path()), "libcrx_x.so")


def ablib():
    """The A/B build (libcrx_x.so = the same sources with CRX_EXPERIMENTAL_KERNELS=1): the only library that contains the two
    measured-and-rejected kernel variants (two-lane EKF, four-lane MPC).  The product libcrx.so does not carry them."""
    global _ab
    if _ab is None:
        import os
        path = ab_lib_path()
        if not os.path.exists(path):
            raise L.CrxError(f"{path} is missing: `make -C cpprobotics_amd/csrc all` builds the A/B library beside libcrx.so")
        L.lib()                       # torch's HIP runtime first, as for the product library
        _ab = C.CDLL(path, mode=C.RTLD_LOCAL)
        _bind(_ab, "libcrx_x.so")
        _ab.crx_last_error.restype = C.c_char_p
    return _ab


def _check_ab(rc, what):
    if rc != 0:
        raise L.CrxError(f"{what} failed with status {rc}: {ablib().crx_last_error().decode('utf-8', 'replace')}")


def dlqr_from_v_lanes(v, dim=5, lanes_per_agent=0, dt=0.1, L_wheelbase=0.5, eps=0.01, maxiter=150):
    """dlqr_from_v with the register layout forced: 1 = one agent per lane, 4 = one agent per DPP quad, 0 = automatic."""
    import torch
    from .lqr import _params
    L.require_cuda(v)
    n = v.shape[0]
    L.expect("v", v, "f", n)
    m = 2 if dim == 5 else 1
    X = torch.empty((n, dim * dim), dtype=torch.float32, device=v.device)
    K = torch.empty((n, m * dim), dtype=torch.float32, device=v.device)
    iters = torch.empty((n,), dtype=torch.int32, device=v.device)
    p = _params(dt, L_wheelbase, eps, maxiter)
    L.check(xlib().crx_x_dare_from_v_lanes_dev(n, dim, L.ptr(v), C.byref(p), L.ptr(X), L.ptr(K), L.ptr(iters),
                                               L.stream_ptr(), int(lanes_per_agent)), "crx_x_dare_from_v_lanes_dev")
    return K, X, iters


def dlqr_from_v_refill(v, dim=5, agents_per_wave=512, hold_lanes=16, dt=0.1, L_wheelbase=0.5, eps=0.01, maxiter=150, poison=True):
    """dlqr_from_v, one agent per lane, through the lane-refilling kernel with `agents_per_wave` agents per wave, handed back
    `hold_lanes` at a time — or, with agents_per_wave = -1, through the masked kernel (no refilling).  -> K, X, iters."""
    import torch
    from .lqr import _params
    L.require_cuda(v)
    n = v.shape[0]
    L.expect("v", v, "f", n)
    m = 2 if dim == 5 else 1
    # poisoned outputs: an agent the kernel's range bookkeeping skipped would show (this entry point exists for tests and A/B scripts)
    # (poison = False for timing loops: three fill kernels less per call)
    X = torch.empty((n, dim * dim), dtype=torch.float32, device=v.device)
    K = torch.empty((n, m * dim), dtype=torch.float32, device=v.device)
    iters = torch.empty((n,), dtype=torch.int32, device=v.device)
    if poison:
        X.fill_(float("nan")); K.fill_(float("nan")); iters.fill_(-1)
    p = _params(dt, L_wheelbase, eps, maxiter)
    L.check(xlib().crx_x_dare_from_v_refill_dev(n, dim, L.ptr(v), C.byref(p), L.ptr(X), L.ptr(K), L.ptr(iters),
                                                L.stream_ptr(), int(agents_per_wave), int(hold_lanes)), "crx_x_dare_from_v_refill_dev")
    return K, X, iters


def dare_dense(A, B, Q, R, eps=0.01, maxiter=150, lanes_per_agent=0):
    """solve_DARE + dlqr through a DENSE kernel for every agent, whatever its matrices look like (the product entry point serves
    agents that carry lqr_steering_control's pattern by the structured kernels); lanes_per_agent = 1 / 4 forces the dense kernel's
    register layout (0: what the product picks for this batch size).  -> X, K, iters."""
    import torch
    from .lqr import _dims
    L.require_cuda(A, B, Q, R)
    n = A.shape[0]
    dim, m = _dims(A)
    L.expect("A", A, "f", n, dim * dim); L.expect("B", B, "f", n, dim * m); L.expect("Q", Q, "f", n, dim * dim); L.expect("R", R, "f", n, m * m)
    X = torch.empty((n, dim * dim), dtype=torch.float32, device=A.device)
    K = torch.empty((n, m * dim), dtype=torch.float32, device=A.device)
    iters = torch.empty((n,), dtype=torch.int32, device=A.device)
    L.check(xlib().crx_x_dare_batch_dense_dev(n, dim, L.ptr(A), L.ptr(B), L.ptr(Q), L.ptr(R), float(eps), int(maxiter),
                                              L.ptr(X), L.ptr(K), L.ptr(iters), L.stream_ptr(), int(lanes_per_agent)), "crx_x_dare_batch_dense_dev")
    return X, K, iters


def mpc_solve_lanes(x0, xref, T, lanes_per_agent=0, params=None):
    """mpc_solve with the register layout forced: 1 = one agent per lane, 4 = one agent per DPP quad (parallel line search),
    0 = automatic.  -> sol, status, cost."""
    import torch
    from .mpc import default_params, mpc_n_vars
    L.require_cuda(x0, xref)
    n = x0.shape[0]
    L.expect("x0", x0, "f", n, 4); L.expect("xref", xref, "f", n, 4 * T)
    p = params if params is not None else default_params()
    sol = torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=x0.device)
    status = torch.empty((n,), dtype=torch.int32, device=x0.device)
    cost = torch.empty((n,), dtype=torch.float64, device=x0.device)
    if int(lanes_per_agent) == 4:      # the rejected four-lane kernel lives in the A/B build only
        _check_ab(ablib().crx_x_mpc_solve_lanes_dev(n, T, L.ptr(x0), L.ptr(xref), C.byref(p), L.ptr(sol), L.ptr(status), L.ptr(cost),
                                                    L.stream_ptr(), 4), "crx_x_mpc_solve_lanes_dev (libcrx_x.so)")
        return sol, status, cost
    L.check(xlib().crx_x_mpc_solve_lanes_dev(n, T, L.ptr(x0), L.ptr(xref), C.byref(p), L.ptr(sol), L.ptr(status), L.ptr(cost),
                                             L.stream_ptr(), int(lanes_per_agent)), "crx_x_mpc_solve_lanes_dev")
    return sol, status, cost


def mpc_solve_refill(x0, xref, T, agents_per_wave=1024, hold_lanes=16, params=None, poison=True, out=None):
    """mpc_solve through the lane-refilling kernel (a wave owns `agents_per_wave` consecutive agents, finished lanes hand their
    agents back `hold_lanes` at a time and take the next ones; asynchronous line search): measured and rejected, in the A/B build
    libcrx_x.so only.  -> sol, status, cost."""
    import torch
    from .mpc import default_params, mpc_n_vars
    L.require_cuda(x0, xref)
    n = x0.shape[0]
    L.expect("x0", x0, "f", n, 4); L.expect("xref", xref, "f", n, 4 * T)
    p = params if params is not None else default_params()
    if out is not None:
        sol, status, cost = out
    else:
        sol = torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=x0.device)
        status = torch.empty((n,), dtype=torch.int32, device=x0.device)
        cost = torch.empty((n,), dtype=torch.float64, device=x0.device)
        if poison:                         # an agent the range bookkeeping skipped would show
            sol.fill_(float("nan")); status.fill_(-1); cost.fill_(float("nan"))
    _check_ab(ablib().crx_x_mpc_solve_refill_dev(n, T, L.ptr(x0), L.ptr(xref), C.byref(p), L.ptr(sol), L.ptr(status), L.ptr(cost),
                                                 L.stream_ptr(), int(agents_per_wave), int(hold_lanes)), "crx_x_mpc_solve_refill_dev (libcrx_x.so)")
    return sol, status, cost


def mpc_solve_geometry(x0, xref, T, agents_per_wave=64, waves_per_workgroup=1, params=None):
    """mpc_solve with the launch geometry forced (the product uses full waves in single-wave workgroups).  -> sol, status, cost."""
    import torch
    from .mpc import default_params, mpc_n_vars
    L.require_cuda(x0, xref)
    n = x0.shape[0]
    L.expect("x0", x0, "f", n, 4); L.expect("xref", xref, "f", n, 4 * T)
    p = params if params is not None else default_params()
    sol = torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=x0.device)
    status = torch.empty((n,), dtype=torch.int32, device=x0.device)
    cost = torch.empty((n,), dtype=torch.float64, device=x0.device)
    L.check(xlib().crx_x_mpc_solve_geometry_dev(n, T, L.ptr(x0), L.ptr(xref), C.byref(p), L.ptr(sol), L.ptr(status), L.ptr(cost),
                                                L.stream_ptr(), int(agents_per_wave), int(waves_per_workgroup)), "crx_x_mpc_solve_geometry_dev")
    return sol, status, cost


def mpc_solve_trig(x0, xref, T, recompute_trig, params=None):
    """mpc_solve with the source of the backward sweep's trig forced (0: stored by the rollout, 1: recomputed in the sweep; the product
    picks by batch size).  The two give the same bits.  -> sol, status, cost."""
    import torch
    from .mpc import default_params, mpc_n_vars
    L.require_cuda(x0, xref)
    n = x0.shape[0]
    L.expect("x0", x0, "f", n, 4); L.expect("xref", xref, "f", n, 4 * T)
    p = params if params is not None else default_params()
    sol = torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=x0.device)
    status = torch.empty((n,), dtype=torch.int32, device=x0.device)
    cost = torch.empty((n,), dtype=torch.float64, device=x0.device)
    L.check(xlib().crx_x_mpc_solve_trig_dev(n, T, L.ptr(x0), L.ptr(xref), C.byref(p), L.ptr(sol), L.ptr(status), L.ptr(cost),
                                            L.stream_ptr(), int(recompute_trig)), "crx_x_mpc_solve_trig_dev")
    return sol, status, cost


def mpc_solve_store(x0, xref, T, store, params=None, out=None):
    """mpc_solve with the layout of the lane's working set forced (0: private memory, crx::mpc_kernel; 1: the tile layout of round 6,
    crx::mpc_tile_kernel — controls in LDS, feedback gains in accumulator registers, T <= 21).  The two give the same bits.
    -> sol, status, cost."""
    import torch
    from .mpc import default_params, mpc_n_vars
    L.require_cuda(x0, xref)
    n = x0.shape[0]
    L.expect("x0", x0, "f", n, 4); L.expect("xref", xref, "f", n, 4 * T)
    p = params if params is not None else default_params()
    if out is not None:
        sol, status, cost = out
    else:
        sol = torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=x0.device)
        status = torch.empty((n,), dtype=torch.int32, device=x0.device)
        cost = torch.empty((n,), dtype=torch.float64, device=x0.device)
    L.check(xlib().crx_x_mpc_solve_store_dev(n, T, L.ptr(x0), L.ptr(xref), C.byref(p), L.ptr(sol), L.ptr(status), L.ptr(cost),
                                             L.stream_ptr(), int(store)), "crx_x_mpc_solve_store_dev")
    return sol, status, cost


def mpc_solve_tile_refill(x0, xref, T, agents_per_wave=1024, hold_lanes=16, params=None, out=None, store=1):
    """mpc_solve through crx::mpc_tile_refill_kernel (tile layout, lanes refilled; store = 1, or 2: the checkpointed tile layout).
    -> sol, status, cost."""
    import torch
    from .mpc import default_params, mpc_n_vars
    L.require_cuda(x0, xref)
    n = x0.shape[0]
    L.expect("x0", x0, "f", n, 4); L.expect("xref", xref, "f", n, 4 * T)
    p = params if params is not None else default_params()
    if out is not None:
        sol, status, cost = out
    else:
        sol = torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=x0.device)
        status = torch.empty((n,), dtype=torch.int32, device=x0.device)
        cost = torch.empty((n,), dtype=torch.float64, device=x0.device)
    L.check(xlib().crx_x_mpc_solve_store_refill_dev(n, T, L.ptr(x0), L.ptr(xref), C.byref(p), L.ptr(sol), L.ptr(status), L.ptr(cost),
                                                    L.stream_ptr(), int(store), int(agents_per_wave), int(hold_lanes)), "crx_x_mpc_solve_store_refill_dev")
    return sol, status, cost


def mpc_solve_two_phase(x0, xref, T, first_sweeps, params=None, out=None, work=None, tail_stream=None):
    """crx_x_mpc_solve_two_phase_dev (measured and not selected, include/crx_experimental.h): the launch with its sweep cap lowered to `first_sweeps` on the current stream, then the agents that
    ran into the cap solved from scratch on `tail_stream` (a torch stream; None = the current one).  Bit for bit mpc_solve's answers,
    complete once both streams have passed the call.  -> (sol, status, cost, work); work: int32 [n + 64], reusable by the next call on
    the same pair of streams."""
    import torch
    from .mpc import default_params, mpc_n_vars
    L.require_cuda(x0, xref)
    n = x0.shape[0]
    L.expect("x0", x0, "f", n, 4); L.expect("xref", xref, "f", n, 4 * T)
    p = params if params is not None else default_params()
    if out is not None:
        sol, status, cost = out
    else:
        sol = torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=x0.device)
        status = torch.empty((n,), dtype=torch.int32, device=x0.device)
        cost = torch.empty((n,), dtype=torch.float64, device=x0.device)
    if work is None:
        work = torch.empty((n + 64,), dtype=torch.int32, device=x0.device)
    L.expect("work", work, "i", n + 64)
    tail = C.c_void_p(tail_stream.cuda_stream) if tail_stream is not None else L.stream_ptr()
    L.check(xlib().crx_x_mpc_solve_two_phase_dev(n, T, L.ptr(x0), L.ptr(xref), C.byref(p), L.ptr(sol), L.ptr(status), L.ptr(cost),
                                                int(first_sweeps), L.ptr(work), L.stream_ptr(), tail), "crx_x_mpc_solve_two_phase_dev")
    return sol, status, cost, work


def mpc_solve_phased(x0, xref, T, caps, params=None, out=None, work=None, store=0):
    """crx_x_mpc_solve_phased_dev: the lockstep solve cut at the sweep indices `caps`, unconverged agents compacted into full waves between
    the phases.  Bit for bit mpc_solve's answers.  -> (sol, status, cost, work); work: uint8 [crx_x_mpc_phased_work_bytes], reusable."""
    import torch
    from .mpc import default_params, mpc_n_vars
    L.require_cuda(x0, xref)
    n = x0.shape[0]
    L.expect("x0", x0, "f", n, 4); L.expect("xref", xref, "f", n, 4 * T)
    p = params if params is not None else default_params()
    if out is not None:
        sol, status, cost = out
    else:
        sol = torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=x0.device)
        status = torch.empty((n,), dtype=torch.int32, device=x0.device)
        cost = torch.empty((n,), dtype=torch.float64, device=x0.device)
    nbytes = xlib().crx_x_mpc_phased_work_bytes(n, T)
    if work is None:
        work = torch.empty((nbytes,), dtype=torch.uint8, device=x0.device)
    assert work.numel() >= nbytes and work.is_contiguous()
    arr = (C.c_int * len(caps))(*[int(c) for c in caps])
    L.check(xlib().crx_x_mpc_solve_phased_store_dev(n, T, L.ptr(x0), L.ptr(xref), C.byref(p), L.ptr(sol), L.ptr(status), L.ptr(cost), arr, len(caps),
                                                    C.c_void_p(work.data_ptr()), L.stream_ptr(), int(store)), "crx_x_mpc_solve_phased_store_dev")
    return sol, status, cost, work


def ekf_run_contracted(xEst, PEst, z, u, Q, R, dt=0.1, x_hist=None):
    """The fused EKF run with the packed step's multiply-then-add pairs fused (crx_x_ekf_run_contracted_dev): an experiment, NOT the
    reference's bits and not within 1e-6 for every vehicle (include/crx_experimental.h)."""
    n, T, q, r, p = _ekf_args(xEst, PEst, z, u, Q, R, dt, x_hist)
    L.check(xlib().crx_x_ekf_run_contracted_dev(n, T, L.ptr(xEst), L.ptr(PEst), L.ptr(z), L.ptr(u), L.ptr(x_hist), q.ctypes.data_as(C.c_void_p),
                                                r.ctypes.data_as(C.c_void_p), C.byref(p), L.stream_ptr()), "crx_x_ekf_run_contracted_dev")
    return xEst, PEst


def _ekf_args(xEst, PEst, z, u, Q, R, dt, x_hist, P_hist=None):
    from .ekf import _params, _qr
    L.require_cuda(xEst, PEst, z, u, x_hist, P_hist)
    T, n = z.shape[0], xEst.shape[0]
    L.expect("xEst", xEst, "f", n, 4); L.expect("PEst", PEst, "f", n, 16); L.expect("z", z, "f", T, n, 2); L.expect("u", u, "f", T, n, 2)
    L.expect("x_hist", x_hist, "f", T, n, 4, optional=True); L.expect("P_hist", P_hist, "f", T, n, 16, optional=True)
    q, r = _qr(Q, R)
    return n, T, q, r, _params(dt)


def ekf_run_addr64(xEst, PEst, z, u, Q, R, dt=0.1, x_hist=None, P_hist=None):
    """ekf_run through the 64-bit-address instantiations of the fused kernel (what batches above 4 M vehicles get)."""
    n, T, q, r, p = _ekf_args(xEst, PEst, z, u, Q, R, dt, x_hist, P_hist)
    L.check(xlib().crx_x_ekf_run_addr64_dev(n, T, L.ptr(xEst), L.ptr(PEst), L.ptr(z), L.ptr(u), L.ptr(x_hist), L.ptr(P_hist),
                                            q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), C.byref(p), L.stream_ptr()),
            "crx_x_ekf_run_addr64_dev")


def ekf_run_pair(xEst, PEst, z, u, Q, R, dt=0.1, x_hist=None):
    """Round 2's A/B variant: ekf_run with two lanes per vehicle.  Returns True if every vehicle stayed on the kernel's fast
    domain (the results then equal ekf_run's as IEEE values); False means xEst / PEst / x_hist of this call are NOT valid — the
    variant has no general-step fallback."""
    import torch
    n, T, q, r, p = _ekf_args(xEst, PEst, z, u, Q, R, dt, x_hist)
    flag = torch.zeros((4,), dtype=torch.int32, device=xEst.device)
    _check_ab(ablib().crx_x_ekf_run_pair_batch_dev(n, T, L.ptr(xEst), L.ptr(PEst), L.ptr(z), L.ptr(u), L.ptr(x_hist),
                                                   q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), C.byref(p), L.ptr(flag),
                                                   L.stream_ptr()), "crx_x_ekf_run_pair_batch_dev (libcrx_x.so)")
    return bool(flag[0].item() == 0)


def dsincos(x):
    """(sin(x), cos(x)) of a float64 tensor as the Frenet kernel evaluates them on the device (csrc/crx_dsincos.h)."""
    import torch
    L.require_cuda(x, dtypes=(torch.float64,))
    n = x.numel()
    L.expect("x", x, "d", n)
    s, c = torch.empty_like(x), torch.empty_like(x)
    L.check(xlib().crx_x_dsincos_dev(n, L.ptr(x), L.ptr(s), L.ptr(c), L.stream_ptr()), "crx_x_dsincos_dev")
    return s, c


def datan2_one(y):
    """atan2(y, 1.0) of a float64 tensor as the tracking kernels evaluate it on the device (csrc/crx_datan2.h)."""
    import torch
    L.require_cuda(y, dtypes=(torch.float64,))
    n = y.numel()
    L.expect("y", y, "d", n)
    out = torch.empty_like(y)
    L.check(xlib().crx_x_datan2_dev(n, L.ptr(y), L.ptr(out), L.stream_ptr()), "crx_x_datan2_dev")
    return out


def datan2_sweep(wheelbase=0.5, device="cuda"):
    """All 2^32 float curvatures k, y = wheelbase * (double)k: (sums, ocml_diff, diff_k) — 4096 block checksums of the device's
    atan2(y, 1.0) bit patterns (int64 tensor, wrap-around sums), the number of curvatures on which OCML's atan differs from it
    (in double, after rounding to float) and up to 64 float32 curvatures of the second kind."""
    import torch
    sums = torch.zeros((4096,), dtype=torch.int64, device=device)
    diff = torch.zeros((3,), dtype=torch.int64, device=device)
    ks = torch.zeros((64,), dtype=torch.int32, device=device)
    L.check(xlib().crx_x_datan2_sweep_dev(float(wheelbase), L.ptr(sums), L.ptr(diff), L.ptr(ks), L.stream_ptr()), "crx_x_datan2_sweep_dev")
    m = int(min(64, diff[1].item()))
    return sums, diff[:2], ks[:m].view(torch.float32)


def hbm_stream(mode, dst, src=None, workgroups=8192):
    """The calibration kernel over dst's bytes: mode 0 dst = src, 1 read src, 2 write dst, 3 dst += 1 in place."""
    L.require_cuda(dst, src)
    nbytes = dst.numel() * dst.element_size()
    L.check(xlib().crx_x_hbm_stream_dev(int(mode), L.ptr(dst), L.ptr(src) if src is not None else None, nbytes, int(workgroups), L.stream_ptr()),
            "crx_x_hbm_stream_dev")


def fetch_units(mode, src, dst, workgroups, passes=1):
    """crx_x_fetch_units_dev: 4-byte (mode 0) / 8-byte (mode 1) per-lane reads of src, or private-memory round trips (mode 2)."""
    L.check(xlib().crx_x_fetch_units_dev(int(mode), L.ptr(src) if src is not None else None, 0 if src is None else src.numel() * src.element_size(),
                                         C.c_void_p(dst.data_ptr()), int(workgroups), int(passes), L.stream_ptr()), "crx_x_fetch_units_dev")


def closed_loop_prediction_lanes(state, course, goal, lanes_per_agent, dim=5, max_ticks=500, goal_dis=None, dt=0.1, L_wheelbase=0.5, eps=0.01,
                                 maxiter=150, kp=1.0, stop_speed=0.05, want_hist=False, pe=None, pth_e=None, ind=None):
    """track.closed_loop_prediction with the register layout forced: 1 = one agent per lane, 4 = one agent per DPP quad."""
    import torch
    from .track import _lqr_params, vehicle_params, loop_params
    L.require_cuda(state, pe, pth_e, ind)
    n = state.shape[0]
    L.expect("state", state, "f", n, 4)
    if goal_dis is None:
        goal_dis = 0.3 if dim == 5 else 0.5
    ticks = torch.zeros((n,), dtype=torch.int32, device=state.device)
    hist = torch.zeros((max_ticks, n, 4), dtype=torch.float32, device=state.device) if want_hist else None
    p = _lqr_params(dt, L_wheelbase, eps, maxiter)
    vp = vehicle_params(False, dt=float(dt), wheelbase=float(L_wheelbase))
    lp = loop_params(goal, goal_dis, max_ticks, kp, stop_speed)
    L.check(xlib().crx_x_lqr_closed_loop_lanes_dev(n, dim, L.ptr(state), course.ref(), L.ptr(pe), L.ptr(pth_e), L.ptr(ind), C.byref(p), C.byref(vp),
                                                   C.byref(lp), L.ptr(hist), L.ptr(ticks), L.stream_ptr(), int(lanes_per_agent)),
            "crx_x_lqr_closed_loop_lanes_dev")
    return ticks, hist


def recip_sweep(device="cuda"):
    """(inputs, mismatches of the EKF step's v_rcp_f32 + one-Newton-step reciprocal, mismatches of rounds 2-4's six-fma form) against
    the IEEE 1.0f / d over every float 2^-60 <= |d| <= 2^60, on this device (csrc/api_probes.inl: crx_x_recip_sweep_dev)."""
    import torch
    counts = torch.zeros(3, dtype=torch.int64, device=device)
    L.check(xlib().crx_x_recip_sweep_dev(L.ptr(counts), L.stream_ptr()), "crx_x_recip_sweep_dev")
    torch.cuda.synchronize()
    return tuple(int(v) for v in counts.cpu())

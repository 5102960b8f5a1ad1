"""Bindings of the measurement-only entry points (include/crx_experimental.h, prefix crx_x_): forced kernel variants and
launch geometries for A/B scripts and bench.py.  Not part of the drop-in surface; the product functions live in the other
modules of this package and choose their variant by themselves."""
import ctypes as C

from . import _lib as L

_P, _I = C.c_void_p, C.c_int
_X_SIGNATURES = {
    "crx_x_dare_from_v_lanes_dev": (_I, [_I, _I, _P, C.POINTER(L.LqrParams), _P, _P, _P, _P, _I]),
}
EXPERIMENTAL_SYMBOLS = tuple(sorted(_X_SIGNATURES))
_bound = False


def xlib():
    global _bound
    l = L.lib()
    if not _bound:
        for name, (res, args) in _X_SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError as e:
                raise L.CrxError(f"libcrx.so does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        _bound = True
    return l


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

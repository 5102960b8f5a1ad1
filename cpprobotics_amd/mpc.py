"""Host-side mirror of the reference's mpc_solve(), batched over n agents.

/root/reference/src/model_predictive_control.cpp:255-346.  x0: [n,4] (x,y,yaw,v) float32 CUDA;
xref: [n,4*T] (column-major 4xT per agent).  Returns the solution in the reference's variable
layout [x(T)|y(T)|yaw(T)|v(T)|delta(T-1)|a(T-1)] as float32 [n, 4T+2(T-1)].
"""
import ctypes as C

from . import _lib as L


def mpc_n_vars(T):
    return 4 * T + 2 * (T - 1)


def default_params():
    p = L.MpcParams()
    L.lib().crx_mpc_default_params(C.byref(p))
    return p


def mpc_solve(x0, xref, T, params=None, return_status=False, portfolio=False, out=None):
    """mpc_solve(State, M_XREF) for n agents (device tensors).  portfolio=True: the four-variant portfolio solve
    (crx_mpc_solve_portfolio_batch_dev): same NLP, every agent answered by the solver variant that converges in the fewest sweeps;
    status bits 2-3 then carry the winning variant.  out = (sol, status, cost): write into the caller's tensors (a pipelined caller
    keeps one set per launch in flight)."""
    import torch
    L.require_cuda(x0, xref)
    n = x0.shape[0]
    L.expect("x0", x0, "f", n, 4); L.expect("xref", xref, "f", n, 4 * T)
    p = params if params is not None else default_params()
    if out is not None:
        sol, status, cost = out
        L.require_cuda(sol, status); L.require_cuda(cost, dtypes=(torch.float64,))
        L.expect("sol", sol, "f", n, mpc_n_vars(T)); L.expect("status", status, "i", n)
        if tuple(cost.shape) != (n,) or cost.dtype != torch.float64 or not cost.is_contiguous():
            raise ValueError("cost must be a contiguous float64 tensor of shape (n,)")
    else:
        sol = torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=x0.device)
        status = torch.empty((n,), dtype=torch.int32, device=x0.device)
        cost = torch.empty((n,), dtype=torch.float64, device=x0.device)
    fn = L.lib().crx_mpc_solve_portfolio_batch_dev if portfolio else L.lib().crx_mpc_solve_batch_dev
    L.check(fn(n, T, L.ptr(x0), L.ptr(xref), C.byref(p), L.ptr(sol), L.ptr(status), L.ptr(cost), L.stream_ptr()),
            "crx_mpc_solve_portfolio_batch_dev" if portfolio else "crx_mpc_solve_batch_dev")
    if return_status:
        return sol, status, cost
    return sol


"""Host-side mirror of the reference's particle-filter localisation, batched over n vehicles.

/root/reference/src/particle_filter.cpp: pf_localization :73-109 + resampling :120-148, T fused ticks per launch,
one vehicle per wavefront.  All tensors float32 CUDA (nobs int32):
  px [n,NP,4], pw [n,NP] (updated in place), obs [T,n,L,3], nobs [T,n], u [T,n,2], nrm [T,n,NP,2], uni [T,n,NP].
"""
import ctypes as C

from . import _lib as L


def pf_default_params():
    p = L.PfParams()
    L.lib().crx_pf_default_params(C.byref(p))
    return p


def pf_run(px, pw, obs, nobs, u, nrm, uni, params=None, want_hist=True):
    """-> (xEst [n,4], PEst [n,16], x_hist [T,n,4] or None, n_resampled int32 [n]); px, pw are updated in place."""
    import torch
    L.require_cuda(px, pw, obs, nobs, u, nrm, uni)
    n, NP = px.shape[0], px.shape[1]
    T, Lm = u.shape[0], obs.shape[2]
    L.expect("px", px, "f", n, NP, 4); L.expect("pw", pw, "f", n, NP); L.expect("obs", obs, "f", T, n, Lm, 3); L.expect("nobs", nobs, "i", T, n)
    L.expect("u", u, "f", T, n, 2); L.expect("nrm", nrm, "f", T, n, NP, 2); L.expect("uni", uni, "f", T, n, NP)
    xEst = torch.empty((n, 4), dtype=torch.float32, device=px.device)
    PEst = torch.empty((n, 16), dtype=torch.float32, device=px.device)
    hist = torch.empty((T, n, 4), dtype=torch.float32, device=px.device) if want_hist else None
    nres = torch.zeros((n,), dtype=torch.int32, device=px.device)
    p = params if params is not None else pf_default_params()
    L.check(L.lib().crx_pf_run_batch_dev(n, NP, T, Lm, L.ptr(px), L.ptr(pw), L.ptr(xEst), L.ptr(PEst), L.ptr(obs), L.ptr(nobs),
                                         L.ptr(u), L.ptr(nrm), L.ptr(uni), C.byref(p), L.ptr(hist), L.ptr(nres), L.stream_ptr()),
            "crx_pf_run_batch_dev")
    return xEst, PEst, hist, nres

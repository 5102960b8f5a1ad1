"""GPU vs the committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from common import bit_equal, floored_rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_ekf_golden(crx):
    import torch
    g = np.load(os.path.join(GOLD, "ekf_golden.npz"))
    T, n = g["z"].shape[:2]
    xt, xd = _t(g["x0"]), _t(g["x0"])
    z, ud = crx.ekf_simulate_inputs(_t(g["u_true"]), xt, xd, _t(g["w"]))
    assert bit_equal(z.cpu().numpy(), g["z"]) and bit_equal(ud.cpu().numpy(), g["ud"])
    x, P = _t(g["x0"]), _t(g["P0"])
    xh = torch.empty((T, n, 4), dtype=torch.float32, device="cuda")
    crx.ekf_run(x, P, z, ud, g["Q"], g["R"], x_hist=xh)
    assert bit_equal(xh.cpu().numpy(), g["x_hist"]) and bit_equal(P.cpu().numpy(), g["P_final"])


def test_lqr_golden(crx):
    g = np.load(os.path.join(GOLD, "lqr_golden.npz"))
    for dim in (5, 4):
        K, X, it = crx.dlqr_from_v(_t(g["v"]), dim=dim)
        assert bit_equal(X.cpu().numpy(), g[f"X{dim}"]) and bit_equal(K.cpu().numpy(), g[f"K{dim}"])
        assert np.array_equal(it.cpu().numpy(), g[f"it{dim}"])


def test_mpc_golden(crx):
    g = np.load(os.path.join(GOLD, "mpc_golden.npz"))
    sol, st, cost = crx.mpc_solve(_t(g["x0"]), _t(g["xref"]), int(g["T"]), return_status=True)
    ok = ((st.cpu().numpy() & 1) == 1) & ((g["status"] & 1) == 1)
    assert ok.mean() > 0.95
    assert floored_rel_err(sol.cpu().numpy()[ok], g["sol"][ok], 1.0) <= 1e-6

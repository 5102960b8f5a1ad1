"""GPU parity: HIP MPC solver (through the C ABI) against its CPU twin — 1e-6 floored relative
(fp64 arithmetic on both sides; libm vs device trig and summation order differ, so not bit-exact)."""
import numpy as np
import pytest

from common import floored_rel_err, mpc_problem

pytestmark = pytest.mark.gpu
TOL = 1e-6   # BASELINE.json: "within 1e-6 relative float tolerance"; floor 1.0


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _compare(crx, oracle_mod, n, T, seed, min_conv):
    x0, xref = mpc_problem(n, T, seed)
    so, sto, co = oracle_mod.mpc_solve(x0, xref, T)
    sd, std, cd = crx.mpc_solve(_t(x0), _t(xref), T, return_status=True)
    sd, std, cd = sd.cpu().numpy(), std.cpu().numpy(), cd.cpu().numpy()
    both = ((sto & 1) == 1) & ((std & 1) == 1)
    assert both.mean() >= min_conv, f"only {both.mean():.3f} of the problems converged on both sides"
    assert floored_rel_err(sd[both], so[both], 1.0) <= TOL
    assert np.max(np.abs(cd[both] - co[both]) / np.maximum(np.abs(co[both]), 1.0)) <= 1e-9
    # bounds hold for every agent, converged or not
    N = T - 1
    assert np.all(np.abs(sd[:, 4 * T:4 * T + N]) <= np.float32(np.pi / 4) + 1e-6)
    assert np.all(np.abs(sd[:, 4 * T + N:]) <= 1.0 + 1e-6)
    # initial state pinned (:309-317)
    assert np.array_equal(sd[:, [0, T, 2 * T, 3 * T]], x0)
    return sd, std


@pytest.mark.parametrize("T", [6, 21])
@pytest.mark.parametrize("n", [1, 64, 65, 500])
def test_mpc_matches_cpu_twin(crx, oracle_mod, n, T):
    _compare(crx, oracle_mod, n, T, seed=n + T, min_conv=0.95 if n > 1 else 1.0)


@pytest.mark.parametrize("T", [2, 3, 9, 30])
def test_mpc_other_horizons(crx, oracle_mod, T):
    _compare(crx, oracle_mod, 200, T, seed=T, min_conv=0.95)


def test_mpc_rollout_is_consistent(crx, oracle_mod):
    """Returned states satisfy the equality constraints (:242-245) for the returned controls."""
    T = 21
    x0, xref = mpc_problem(300, T, 9)
    sd = crx.mpc_solve(_t(x0), _t(xref), T).cpu().numpy().astype(np.float64)
    N = T - 1
    x, y, yaw, v = sd[:, :T], sd[:, T:2 * T], sd[:, 2 * T:3 * T], sd[:, 3 * T:4 * T]
    d, a = sd[:, 4 * T:4 * T + N], sd[:, 4 * T + N:]
    assert np.max(np.abs(x[:, 1:] - (x[:, :-1] + v[:, :-1] * np.cos(yaw[:, :-1]) * 0.2))) < 2e-4
    assert np.max(np.abs(y[:, 1:] - (y[:, :-1] + v[:, :-1] * np.sin(yaw[:, :-1]) * 0.2))) < 2e-4
    assert np.max(np.abs(yaw[:, 1:] - (yaw[:, :-1] + v[:, :-1] * np.tan(d) / 2.5 * 0.2))) < 1e-5
    assert np.max(np.abs(v[:, 1:] - (v[:, :-1] + a * 0.2))) < 1e-5


def test_mpc_full_size(crx, oracle_mod):
    """BASELINE config 4: 8,192 agents, 20 control intervals (T = 21)."""
    n, T = 8192, 21
    x0, xref = mpc_problem(n, T, 4)
    sd, std, cd = crx.mpc_solve(_t(x0), _t(xref), T, return_status=True)
    sd, std, cd = sd.cpu().numpy(), std.cpu().numpy(), cd.cpu().numpy()
    assert (std & 1).mean() >= 0.97
    assert np.isfinite(sd).all() and np.isfinite(cd).all()
    # a strided sample against the CPU twin, and optimality: the cost never exceeds the zero-control cost
    idx = np.arange(0, n, 16)
    so, sto, co = oracle_mod.mpc_solve(x0[idx], xref[idx], T)
    both = ((sto & 1) == 1) & ((std[idx] & 1) == 1)
    assert both.mean() >= 0.95
    assert floored_rel_err(sd[idx][both], so[both], 1.0) <= TOL
    j0 = np.array([oracle_mod.mpc_cost(x0[k], xref[k], T, np.zeros((T - 1, 2)))[0] for k in idx])
    assert np.all(cd[idx] <= j0 + 1e-9)


def test_mpc_edge_cases(crx):
    import torch
    sol = crx.mpc_solve(torch.empty((0, 4), device="cuda"), torch.empty((0, 24), device="cuda"), 6)
    assert sol.shape == (0, 34)
    with pytest.raises(crx.CrxError):
        crx.mpc_solve(torch.zeros((2, 4), device="cuda"), torch.zeros((2, 4), device="cuda"), 1)   # T < 2
    with pytest.raises(crx.CrxError):
        crx.mpc_solve(torch.zeros((2, 4), device="cuda"), torch.zeros((2, 4 * 65), device="cuda"), 65)  # T > 64

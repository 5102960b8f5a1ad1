"""The particle-filter oracle (oracle/pf_ref.cpp): it localises, it resamples, and a float64 numpy restatement of one
tick agrees with it."""
import numpy as np


def _scenario(oracle, n, T, NP, seed):
    rng = np.random.default_rng(seed)
    u = np.tile(np.array([[1.0, 0.1]], np.float32), (n, 1))
    w_u = rng.standard_normal((T, n, 2)).astype(np.float32)
    w_z = rng.standard_normal((T, n, 4)).astype(np.float32)
    ud, obs, nobs, xth, xdh = oracle.pf_simulate_inputs(u, np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32), w_u, w_z)
    nrm = rng.standard_normal((T, n, NP, 2)).astype(np.float32)
    uni = rng.uniform(1.0, 2.0, (T, n, NP)).astype(np.float32)
    return np.tile(u[None], (T, 1, 1)), obs, nobs, nrm, uni, xth, xdh


def test_pf_localises_and_resamples(oracle_mod):
    n, T, NP = 6, 500, 100
    ut, obs, nobs, nrm, uni, xth, xdh = _scenario(oracle_mod, n, T, NP, 1)
    px, pw = np.zeros((n, NP, 4), np.float32), np.full((n, NP), 1.0 / NP, np.float32)
    px2, pw2, xe, Pe, xh, nres = oracle_mod.pf_run(px, pw, obs, nobs, ut, nrm, uni)
    err = np.hypot(xh[..., 0] - xth[..., 0], xh[..., 1] - xth[..., 1])
    err_dr = np.hypot(xdh[..., 0] - xth[..., 0], xdh[..., 1] - xth[..., 1])
    assert err.mean() < 0.1 and err_dr.mean() > 1.0                   # the reference's behaviour: PF tracks, dead reckoning drifts
    assert (nres > 100).all() and (nres < T).all()
    assert np.allclose(pw2.sum(axis=1), 1.0, atol=1e-5)
    P = Pe.reshape(n, 4, 4)
    assert np.allclose(P, P.transpose(0, 2, 1), atol=1e-6) and (np.linalg.eigvalsh(P.astype(np.float64)) > -1e-6).all()


def test_pf_step_matches_float64_numpy(oracle_mod):
    n, NP = 16, 100
    rng = np.random.default_rng(2)
    ut, obs, nobs, nrm, uni, xth, _ = _scenario(oracle_mod, n, 30, NP, 3)
    pw = rng.uniform(0.5, 1.5, (n, NP)).astype(np.float32); pw /= pw.sum(axis=1, keepdims=True)
    t = 20
    px = (xth[t - 1][:, None, :] + rng.normal(0, 0.05, (n, NP, 4))).astype(np.float32)       # a particle cloud around the truth
    px2, pw2, xe, Pe, res, anc = oracle_mod.pf_step(px, pw, obs[t], nobs[t], ut[t], nrm[t], uni[t])
    rs = np.array(oracle_mod.oracle_lib.PF_RSIM)
    for a in range(n):
        x = px[a].astype(np.float64)
        ud = ut[t, a].astype(np.float64) + nrm[t, a].astype(np.float64) * rs
        xn = np.stack([x[:, 0] + 0.1 * np.cos(x[:, 2]) * ud[:, 0], x[:, 1] + 0.1 * np.sin(x[:, 2]) * ud[:, 0],
                       x[:, 2] + 0.1 * ud[:, 1], x[:, 3] + ud[:, 0]], axis=1)
        w = pw[a].astype(np.float64)
        for i in range(nobs[t, a]):
            dz = np.hypot(xn[:, 0] - obs[t, a, i, 1], xn[:, 1] - obs[t, a, i, 2]) - obs[t, a, i, 0]
            w = w * (1.0 / np.sqrt(2 * 3.141592653 * 0.01) * np.exp(-dz * dz / (2 * 0.01)))
        w /= w.sum()
        xe64 = xn.T @ w
        dx = xn - xe64
        P64 = (dx * w[:, None]).T @ dx
        assert np.allclose(xe[a], xe64, rtol=2e-5, atol=2e-5) and np.allclose(Pe[a].reshape(4, 4).T, P64, rtol=1e-3, atol=2e-6)
        neff = 1.0 / (w * w).sum()
        if abs(neff - NP / 2) > 0.5:
            assert res[a] == int(neff < NP / 2)
        if res[a]:
            wc = np.cumsum(w)
            rid = np.arange(NP) / NP + uni[t, a].astype(np.float64) / NP
            ind = np.minimum(np.searchsorted(wc, rid, side="left"), NP - 1)
            assert (np.maximum.accumulate(ind) == anc[a]).mean() > 0.97
            assert np.allclose(px2[a], xn[anc[a]].astype(np.float32), rtol=1e-5, atol=1e-5)
        else:
            assert np.allclose(px2[a], xn, rtol=1e-5, atol=1e-5) and np.allclose(pw2[a], w, rtol=1e-4, atol=1e-8)


def test_wave_order_sums_agree_statistically_with_index_order(oracle_mod):
    """The oracle in the engine's summation order (oracle_pf_step_wave: what the kernel is demanded equal to bit for bit) against
    its index-order statement: same resampling rhythm, same estimation error, single ticks equal to float round-off."""
    n, T, NP = 40, 120, 100
    ut, obs, nobs, nrm, uni, xth, _ = _scenario(oracle_mod, n, T, NP, 5)
    px, pw = np.zeros((n, NP, 4), np.float32), np.full((n, NP), 1.0 / NP, np.float32)
    a = oracle_mod.pf_run(px, pw, obs, nobs, ut, nrm, uni)
    b = oracle_mod.pf_run(px, pw, obs, nobs, ut, nrm, uni, wave_order=True)
    err = lambda h: np.hypot(h[..., 0] - xth[..., 0], h[..., 1] - xth[..., 1]).mean()
    assert abs(err(a[4]) - err(b[4])) < 5e-3 and err(b[4]) < 0.1
    assert np.abs(a[5].astype(int) - b[5]).max() <= 0.05 * T
    s1 = oracle_mod.pf_step(px, pw, obs[0], nobs[0], ut[0], nrm[0], uni[0])
    s2 = oracle_mod.pf_step(px, pw, obs[0], nobs[0], ut[0], nrm[0], uni[0], wave_order=True)
    assert np.allclose(s1[2], s2[2], rtol=1e-5, atol=1e-5) and np.allclose(s1[3], s2[3], rtol=1e-3, atol=1e-6)

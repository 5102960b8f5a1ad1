"""GPU parity of the particle filter (one vehicle per wavefront) against the CPU oracle.

Bit for bit against the oracle evaluated in the engine's summation order (balanced tree over the lanes: oracle_pf_step_wave) —
whole episodes, resampling included.  Statistically against the oracle's index-order sums (the plain reading of the reference,
whose own order is Eigen's vectorised redux / gemv): single ticks agree to float round-off; over long runs a resampling tie can
flip, so episodes are compared through their estimation error and resampling counts."""
import numpy as np
import pytest

from test_oracle_pf import _scenario

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("NP", [100, 64, 128])
def test_pf_episode_bit_exact_in_the_engines_summation_order(crx, oracle_mod, NP):
    """150 ticks, 150 vehicles: every estimate of every tick, the final particles, weights, covariance and the number of resampling
    ticks equal to the wave-order oracle bit for bit (ticks in which nobody sees a landmark included)."""
    n, T = 150, 150
    ut, obs, nobs, nrm, uni, xth, _ = _scenario(oracle_mod, n, T, NP, 31 + NP)
    nobs = nobs.copy(); nobs[7] = 0; nobs[40, ::3] = 0
    px, pw = np.zeros((n, NP, 4), np.float32), np.full((n, NP), 1.0 / NP, np.float32)
    pxo, pwo, xeo, Peo, xho, nro = oracle_mod.pf_run(px, pw, obs, nobs, ut, nrm, uni, wave_order=True)
    pxd, pwd = _t(px), _t(pw)
    xe, Pe, hist, nres = crx.pf_run(pxd, pwd, _t(obs), _t(nobs), _t(ut), _t(nrm), _t(uni))
    assert np.array_equal(hist.cpu().numpy(), xho) and np.array_equal(xe.cpu().numpy(), xeo) and np.array_equal(Pe.cpu().numpy(), Peo)
    assert np.array_equal(pxd.cpu().numpy(), pxo) and np.array_equal(pwd.cpu().numpy(), pwo)
    assert np.array_equal(nres.cpu().numpy(), nro) and nro.min() > 10          # the resampling branch is exercised
    err = np.hypot(xho[..., 0] - xth[..., 0], xho[..., 1] - xth[..., 1])
    assert err.mean() < 0.1


@pytest.mark.parametrize("n", [1, 3, 5, 130])
def test_pf_single_tick_bit_exact_in_the_engines_summation_order(crx, oracle_mod, n):
    NP = 100
    rng = np.random.default_rng(n)
    ut, obs, nobs, nrm, uni, xth, _ = _scenario(oracle_mod, n, 30, NP, 10 + n)
    pw = rng.uniform(0.5, 1.5, (n, NP)).astype(np.float32); pw /= pw.sum(axis=1, keepdims=True)
    for t in (1, 7, 20):
        px = (xth[t - 1][:, None, :] + rng.normal(0, 0.05, (n, NP, 4))).astype(np.float32)
        pxo, pwo, xeo, Peo, reso, _ = oracle_mod.pf_step(px, pw, obs[t], nobs[t], ut[t], nrm[t], uni[t], wave_order=True)
        pxd, pwd = _t(px), _t(pw)
        xe, Pe, hist, nres = crx.pf_run(pxd, pwd, _t(obs[t:t + 1]), _t(nobs[t:t + 1]), _t(ut[t:t + 1]), _t(nrm[t:t + 1]), _t(uni[t:t + 1]))
        assert np.array_equal(xe.cpu().numpy(), xeo) and np.array_equal(Pe.cpu().numpy(), Peo) and np.array_equal(nres.cpu().numpy(), reso)
        assert np.array_equal(pxd.cpu().numpy(), pxo) and np.array_equal(pwd.cpu().numpy(), pwo)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 130])
def test_pf_single_tick_matches_oracle(crx, oracle_mod, n):
    NP = 100
    rng = np.random.default_rng(n)
    ut, obs, nobs, nrm, uni, xth, _ = _scenario(oracle_mod, n, 30, NP, 10 + n)
    pw = rng.uniform(0.5, 1.5, (n, NP)).astype(np.float32); pw /= pw.sum(axis=1, keepdims=True)
    for t in (1, 7, 20):
        px = (xth[t - 1][:, None, :] + rng.normal(0, 0.05, (n, NP, 4))).astype(np.float32)   # a particle cloud around the truth
        pxo, pwo, xeo, Peo, reso, anco = oracle_mod.pf_step(px, pw, obs[t], nobs[t], ut[t], nrm[t], uni[t])
        pxd, pwd = _t(px), _t(pw)
        xe, Pe, hist, nres = crx.pf_run(pxd, pwd, _t(obs[t:t + 1]), _t(nobs[t:t + 1]), _t(ut[t:t + 1]), _t(nrm[t:t + 1]), _t(uni[t:t + 1]))
        assert np.allclose(xe.cpu().numpy(), xeo, rtol=1e-5, atol=1e-5)
        assert np.allclose(Pe.cpu().numpy(), Peo, rtol=1e-3, atol=1e-6)
        assert np.allclose(hist.cpu().numpy()[0], xeo, rtol=1e-5, atol=1e-5)
        same = nres.cpu().numpy() == reso
        assert same.mean() >= 0.95                                   # Neff within round-off of NP/2 may flip
        g = pxd.cpu().numpy()
        close = np.isclose(g, pxo, rtol=1e-5, atol=1e-5).all(axis=2)
        assert close[same].mean() > 0.98                             # a particle on a cumulative-weight boundary may pick its neighbour
        assert np.allclose(pwd.cpu().numpy()[same], pwo[same], rtol=1e-4, atol=1e-8)


def test_pf_episode_tracks_like_the_oracle(crx, oracle_mod):
    n, T, NP = 96, 300, 100
    ut, obs, nobs, nrm, uni, xth, xdh = _scenario(oracle_mod, n, T, NP, 5)
    px, pw = np.zeros((n, NP, 4), np.float32), np.full((n, NP), 1.0 / NP, np.float32)
    _, _, xeo, Peo, xho, nreso = oracle_mod.pf_run(px, pw, obs, nobs, ut, nrm, uni)
    pxd, pwd = _t(px), _t(pw)
    xe, Pe, hist, nres = crx.pf_run(pxd, pwd, _t(obs), _t(nobs), _t(ut), _t(nrm), _t(uni))
    h = hist.cpu().numpy()
    err_g = np.hypot(h[..., 0] - xth[..., 0], h[..., 1] - xth[..., 1])
    err_o = np.hypot(xho[..., 0] - xth[..., 0], xho[..., 1] - xth[..., 1])
    assert err_g.mean() < 0.1 and abs(err_g.mean() - err_o.mean()) < 0.01
    # vehicles whose runs never diverged (no flipped resampling decision) agree to round-off for the whole episode
    same = np.abs(h - xho).max(axis=(0, 2)) < 1e-3
    assert same.mean() > 0.5
    assert np.abs(nres.cpu().numpy().astype(int) - nreso).max() <= 0.05 * T
    assert np.allclose(pwd.cpu().numpy().sum(axis=1), 1.0, atol=1e-4)


@pytest.mark.parametrize("NP", [64, 128])
def test_pf_other_particle_counts(crx, oracle_mod, NP):
    """The other two instantiations (one full wave of particles; two particles per lane), single ticks and a short episode,
    including ticks in which a vehicle sees no landmark at all."""
    n, T = 37, 40
    rng = np.random.default_rng(NP)
    ut, obs, nobs, nrm, uni, xth, _ = _scenario(oracle_mod, n, T, NP, 77 + NP)
    nobs = nobs.copy()
    nobs[3] = 0                                                      # tick 3: nobody observes anything
    pw = np.full((n, NP), 1.0 / NP, np.float32)
    for t in (1, 3, 9):
        px = (xth[t - 1][:, None, :] + rng.normal(0, 0.05, (n, NP, 4))).astype(np.float32)
        pxo, pwo, xeo, Peo, reso, _ = oracle_mod.pf_step(px, pw, obs[t], nobs[t], ut[t], nrm[t], uni[t])
        pxd, pwd = _t(px), _t(pw)
        xe, Pe, hist, nres = crx.pf_run(pxd, pwd, _t(obs[t:t + 1]), _t(nobs[t:t + 1]), _t(ut[t:t + 1]), _t(nrm[t:t + 1]), _t(uni[t:t + 1]))
        assert np.allclose(xe.cpu().numpy(), xeo, rtol=1e-5, atol=1e-5)
        assert np.allclose(Pe.cpu().numpy(), Peo, rtol=1e-3, atol=1e-6)
        assert (nres.cpu().numpy() == reso).mean() >= 0.9
    px, pw = np.zeros((n, NP, 4), np.float32), np.full((n, NP), 1.0 / NP, np.float32)
    _, _, xeo, _, xho, nreso = oracle_mod.pf_run(px, pw, obs, nobs, ut, nrm, uni)
    pxd, pwd = _t(px), _t(pw)
    xe, Pe, hist, nres = crx.pf_run(pxd, pwd, _t(obs), _t(nobs), _t(ut), _t(nrm), _t(uni))
    h = hist.cpu().numpy()
    err_g = np.hypot(h[..., 0] - xth[..., 0], h[..., 1] - xth[..., 1])
    err_o = np.hypot(xho[..., 0] - xth[..., 0], xho[..., 1] - xth[..., 1])
    assert abs(err_g.mean() - err_o.mean()) < 0.02 and err_g.mean() < 0.2
    assert np.allclose(pwd.cpu().numpy().sum(axis=1), 1.0, atol=1e-4)

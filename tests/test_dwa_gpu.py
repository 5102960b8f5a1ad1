"""GPU parity of the dynamic-window planner (one agent per wavefront) against the CPU oracle — bit-exact, including
which of the ~405 sampled trajectories wins."""
import numpy as np
import pytest

from common import bit_equal

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _agents(n, seed):
    rng = np.random.default_rng(seed)
    st = np.stack([rng.uniform(-1, 9, n), rng.uniform(-1, 9, n), rng.uniform(-3.2, 3.2, n), rng.uniform(-0.5, 1.0, n),
                   rng.uniform(-0.69, 0.69, n)], axis=1).astype(np.float32)
    st[0] = (0.0, 0.0, 3.141592653 / 8.0, 0.0, 0.0)                 # the reference's start
    u = np.stack([st[:, 3], st[:, 4]], axis=1).copy()
    goal = np.stack([rng.uniform(8, 12, n), rng.uniform(8, 12, n)], axis=1).astype(np.float32)
    goal[0] = (10.0, 10.0)
    return st, u, goal


@pytest.mark.parametrize("n", [1, 4, 5, 203])
def test_dwa_control_single_tick_bit_exact(crx, oracle_mod, n):
    ob = oracle_mod.oracle_lib.DWA_OBSTACLES
    st, u, goal = _agents(n, n)
    so, uo, to, _ = oracle_mod.dwa_run(st, u, goal, 1)
    _, nso, bio = oracle_mod.dwa_control(st, u, goal)
    sd, ud = _t(st), _t(u)
    ticks, _, status, best, ns = crx.dwa_run(sd, ud, _t(goal), _t(ob), 1)
    assert (status.cpu().numpy() == 0).all()
    assert np.array_equal(ns.cpu().numpy(), nso) and np.array_equal(best.cpu().numpy(), bio)
    assert bit_equal(ud.cpu().numpy(), uo) and bit_equal(sd.cpu().numpy(), so)


def test_dwa_episode_bit_exact(crx, oracle_mod):
    ob = oracle_mod.oracle_lib.DWA_OBSTACLES
    n, max_ticks = 40, 350
    st, u, goal = _agents(n, 77)
    so, uo, to, ho = oracle_mod.dwa_run(st, u, goal, max_ticks, want_hist=True)
    sd, ud = _t(st), _t(u)
    ticks, hist, status, _, _ = crx.dwa_run(sd, ud, _t(goal), _t(ob), max_ticks, want_hist=True)
    assert np.array_equal(ticks.cpu().numpy(), to)
    assert bit_equal(sd.cpu().numpy(), so) and bit_equal(ud.cpu().numpy(), uo)
    h = hist.cpu().numpy()
    for a in range(n):
        assert bit_equal(h[: to[a], a], ho[: to[a], a])
    assert (to < max_ticks).mean() > 0.2          # DWA is a local planner: random starts behind obstacles get stuck, as in the reference

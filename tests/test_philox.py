"""The engine's counter-based input noise (cpprobotics_amd/csrc/crx_philox.h): known answers of Philox4x32-10, the
distribution of the draws, shard-independence, and host == device bytes."""
import numpy as np
import pytest


def test_philox4x32_10_known_answers(oracle_mod):
    """Random123's kat_vectors for philox4x32, 10 rounds."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        assert tuple(int(v) for v in oracle_mod.philox4x32_10(ctr, key)) == out


def test_draws_are_standard_normal_and_keyed_by_global_agent(oracle_mod):
    w = oracle_mod.normal_draws(4096, 50, agent0=0, seed=7)
    f = w.astype(np.float64).reshape(-1)
    assert np.isfinite(f).all() and abs(f.mean()) < 5e-3 and abs(f.std() - 1.0) < 5e-3
    assert abs((f ** 3).mean()) < 2e-2 and abs((f ** 4).mean() - 3.0) < 5e-2 and np.abs(f).max() < 6.0
    # the four draws of a pass and consecutive passes are uncorrelated
    c = np.corrcoef(w.reshape(-1, 4).T.astype(np.float64))
    assert np.abs(c - np.eye(4)).max() < 1e-2
    # shard-independence: agents [1000, 1300) drawn on their own are the same bytes
    part = oracle_mod.normal_draws(300, 50, agent0=1000, seed=7)
    assert np.array_equal(part, w[:, 1000:1300])
    # seed and stream select different sequences
    assert not np.array_equal(oracle_mod.normal_draws(8, 4, seed=8), oracle_mod.normal_draws(8, 4, seed=7))
    assert not np.array_equal(oracle_mod.normal_draws(8, 4, seed=7, stream_id=1), oracle_mod.normal_draws(8, 4, seed=7))
    # agent ids beyond 2^32 use the high counter word
    hi = oracle_mod.normal_draws(4, 3, agent0=(1 << 32), seed=7)
    assert not np.array_equal(hi, w[:3, :4])


@pytest.mark.gpu
def test_device_draws_equal_host_draws(crx, oracle_mod):
    for n, T, a0 in ((1, 1, 0), (257, 33, 0), (1000, 64, 123456789), (64, 5, (1 << 32) - 10)):
        d = crx.normal_draws(n, T, agent0=a0, seed=0xC0FFEE, stream_id=3).cpu().numpy()
        h = oracle_mod.normal_draws(n, T, agent0=a0, seed=0xC0FFEE, stream_id=3)
        assert np.array_equal(d.view(np.uint32), h.view(np.uint32))
    assert crx.normal_draws(0, 5).shape == (5, 0, 4)

"""Shared synthetic workloads and the error metric used by the parity tests."""
import math

import numpy as np


def ekf_QR():
    Q = np.zeros((4, 4), dtype=np.float32)
    Q[0, 0] = np.float32(0.1 * 0.1)
    Q[1, 1] = np.float32(0.1 * 0.1)
    Q[2, 2] = np.float32((1.0 / 180 * math.pi) * (1.0 / 180 * math.pi))
    Q[3, 3] = np.float32(0.1 * 0.1)
    R = np.eye(2, dtype=np.float32)
    return Q.T.copy().reshape(-1), R.T.copy().reshape(-1)


def ekf_agents(n, seed, single_vehicle=False):
    """Per-vehicle true input and initial state.  C1: the reference's own u=(1.0,0.1), x=0.
    C2: yaw ~ U(-pi,pi), u0 ~ U(0.5,2), u1 ~ U(-0.3,0.3)."""
    rng = np.random.default_rng(seed)
    if single_vehicle:
        u = np.tile(np.array([[1.0, 0.1]], dtype=np.float32), (n, 1))
        x0 = np.zeros((n, 4), dtype=np.float32)
    else:
        u = np.stack([rng.uniform(0.5, 2.0, n), rng.uniform(-0.3, 0.3, n)], axis=1).astype(np.float32)
        x0 = np.zeros((n, 4), dtype=np.float32)
        x0[:, 2] = rng.uniform(-math.pi, math.pi, n).astype(np.float32)
    P0 = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (n, 1))
    return u, x0, P0


def ekf_noise(T, n, seed):
    rng = np.random.Generator(np.random.Philox(key=seed))
    return rng.standard_normal((T, n, 4), dtype=np.float32)


def lqr_speeds(n, seed):
    """C3: v ~ U(-3, 6) with 5 % of the agents at |v| < 0.1 (iteration-cap case)."""
    rng = np.random.default_rng(seed)
    v = rng.uniform(-3.0, 6.0, n)
    slow = rng.random(n) < 0.05
    v[slow] = rng.uniform(-0.1, 0.1, slow.sum())
    return v.astype(np.float32)


def floored_rel_err(a, b, floor):
    """max |a-b| / max(|b|, floor)  (SURVEY.md §8d error definition)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def bit_equal(a, b):
    """Equality as IEEE values (+0 == -0), NaN never equal — the parity bar for fp32 paths."""
    return bool(np.array_equal(np.asarray(a), np.asarray(b)))

"""Generates the committed golden fixtures (tests/golden/*.npz).

The reference cannot run in this image (no Eigen / OpenCV / CppAD / IPOPT)
and holds no test vectors of its own, so these fixtures are produced by the CPU oracle — the
Eigen-order restatement validated in tests/test_oracle_*.py — with its deterministic trig mode
(trig=1: glibc's FMA-variant sinf/cosf restated; identical to libm on FMA hosts).  They pin the
oracle against silent drift and give the GPU tests a fixed target.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from common import ekf_QR, ekf_agents, ekf_noise, lqr_speeds, mpc_problem  # noqa: E402

Q, R = ekf_QR()
n, T = 32, 250
u, x0, P0 = ekf_agents(n, 42)
u[0] = (1.0, 0.1); x0[0] = 0      # vehicle 0 = the reference's own scenario
w = ekf_noise(T, n, 43)
z, ud, _, _, _, _ = oracle.ekf_simulate_inputs(u, x0, x0, w, trig=1)
x, P, xh, ph = oracle.ekf_run(x0, P0, z, ud, Q, R, trig=1, want_phist=True)
np.savez_compressed(os.path.join(HERE, "ekf_golden.npz"), u_true=u, x0=x0, P0=P0, w=w, z=z, ud=ud, Q=Q, R=R,
                    x_hist=xh, P_final=ph[-1])

v = lqr_speeds(256, 44)
v[:4] = (0.0, 0.05, 2.78, -1.0)
out = dict(v=v)
for dim in (5, 4):
    A, B, Qm, Rm = oracle.lqr_build(v, dim)
    X, K, it = oracle.dare(A, B, Qm, Rm)
    out[f"X{dim}"], out[f"K{dim}"], out[f"it{dim}"] = X, K, it
np.savez_compressed(os.path.join(HERE, "lqr_golden.npz"), **out)

Tm = 21
mx0, mxref = mpc_problem(64, Tm, 45)
sol, st, cost = oracle.mpc_solve(mx0, mxref, Tm)
np.savez_compressed(os.path.join(HERE, "mpc_golden.npz"), x0=mx0, xref=mxref, T=Tm, sol=sol, status=st, cost=cost)
print("wrote", os.listdir(HERE))

#!/usr/bin/env python3
"""CPU-twin experiment for DESIGN.md 5 (round 4): would it shorten the slowest agent if, from sweep K on, idle lanes of its wave
adopted its iterate and continued it as the other variants of the portfolio — (n_gn, trust) = (3, 1), (2, 2), (1, 2) beside the
engine's own (2, 1)?  For each seed: the engine's sweep counts; then for K in 4 .. 12 the sweep count of every agent under each
variant switched in at sweep K; an agent's count is the minimum over the four continuations (the first lane to converge answers).
Prints the slowest agent and the mean per (seed, K)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle.oracle_lib as O  # noqa: E402
from common import mpc_problem  # noqa: E402

lib = O.lib()
lib.oracle_mpc_late_switch.argtypes = [C.c_int, C.c_int, C.c_double]
T, n = 21, 8192
for seed in (4, 1, 2, 3, 5, 6):
    x0, xref = mpc_problem(n, T, seed)
    lib.oracle_mpc_late_switch(-1, 2, 1.0)
    _, st, _ = O.mpc_solve(x0, xref, T)
    base = (st >> 8).astype(np.int64)
    _, stp, _ = O.mpc_solve_portfolio(x0, xref, T)
    port = (stp >> 8).astype(np.int64)
    print(f"seed {seed}: engine max {base.max()} mean {base.mean():.2f} | portfolio from the start: max {port.max()} mean {port.mean():.2f}")
    for K in (4, 6, 8, 10, 12):
        best = base.copy()
        for n_gn, trust in ((3, 1.0), (2, 2.0), (1, 2.0)):
            lib.oracle_mpc_late_switch(K, n_gn, trust)
            _, s2, _ = O.mpc_solve(x0, xref, T)
            it2 = (s2 >> 8).astype(np.int64)
            conv = (s2 & 1) == 1
            best = np.where(conv & (it2 < best), it2, best)
        lib.oracle_mpc_late_switch(-1, 2, 1.0)
        print(f"   switch at sweep {K:2d}: slowest agent {best.max():2d}  mean {best.mean():.2f}  agents helped {(best < base).sum()}")

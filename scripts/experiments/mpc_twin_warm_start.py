#!/usr/bin/env python3
"""Round-4 experiment on the CPU twin of the MPC solver (oracle/mpc_ref.cpp; no GPU): does a warm start from the reference
trajectory shorten the sweep-count distribution that sets the kernel's launch time?

  mode 0  the reference's zero initial guess (src/model_predictive_control.cpp:266-274) — what the engine uses
  mode 1  a_i = (v_ref[i+1] - v_i) / DT (reach the reference speed of the next knot), steering 0
  mode 2  mode 1 + steering that reaches the reference heading of the next knot, atan(dyaw * WB / (v DT)), clamped
  mode 3  mode 1 + steering from the heading rate of the reference itself (feed-forward)

A launch lasts as long as its slowest wave, i.e. sum over sweeps of the per-sweep maximum over the wave's 64 lanes; the columns
are the mean / maximum sweep count, the mean over waves of the wave maximum, and how many agents end at a different cost.
Usage: python scripts/experiments/mpc_twin_warm_start.py [seeds...]      (writes nothing; profiles/r04/mpc_experiments.txt keeps a run)"""
import ctypes
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from common import mpc_problem  # noqa: E402
from oracle import oracle_lib  # noqa: E402

oracle.build()
lib = oracle_lib.lib()
THREADS = len(os.sched_getaffinity(0))


def run(x0, xref, T, mode, tol=None):
    lib.oracle_mpc_warm(mode)
    n = len(x0)
    cuts = [n * k // THREADS for k in range(THREADS + 1)]
    res = [None] * THREADS
    prm = {"tol": tol} if tol else None

    def f(k):
        res[k] = oracle.mpc_solve(x0[cuts[k]:cuts[k + 1]], xref[cuts[k]:cuts[k + 1]], T, params=prm)
    with ThreadPoolExecutor(THREADS) as ex:
        list(ex.map(f, range(THREADS)))
    lib.oracle_mpc_warm(0)
    return tuple(np.concatenate([r[j] for r in res]) for j in range(3))


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [4, 5, 6, 7]
    print("seed mode | sweeps mean  max | mean of wave max | converged | agents ending lower / higher in cost (> 1e-6 rel) than mode 0")
    for seed in seeds:
        x0, xref = mpc_problem(8192, 21, seed)
        base = None
        for mode in (0, 1, 2, 3):
            _, st, c = run(x0, xref, 21, mode)
            it = st >> 8
            wm = it.reshape(-1, 64).max(axis=1)
            base = c if mode == 0 else base
            d = c - base
            s = np.maximum(1.0, np.abs(base))
            print(f"{seed:4d} {mode:4d} | {it.mean():11.2f} {it.max():4d} | {wm.mean():16.2f} | {((st & 1) == 1).mean():9.4f} | "
                  f"{int((d < -1e-6 * s).sum()):6d} / {int((d > 1e-6 * s).sum())}")
    print("\ntolerance on the Newton step (mode 0): what one backward sweep less would cost in accuracy")
    x0, xref = mpc_problem(8192, 21, 4)
    ref = run(x0, xref, 21, 0)
    for tol in (1e-9, 1e-7, 1e-6, 1e-5):
        sol, st, c = run(x0, xref, 21, 0, tol=tol)
        it = st >> 8
        err = np.max(np.abs(sol - ref[0]) / np.maximum(1.0, np.abs(ref[0])))
        print(f"tol {tol:7.0e}: sweeps mean {it.mean():.2f} max {it.max()} mean of wave max {it.reshape(-1, 64).max(axis=1).mean():.2f}; "
              f"max floored rel. difference of the float solution from the tol 1e-9 one: {err:.2e}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B of the MPC solve in its throughput regime: mpc_kernel (a wave lasts as long as its slowest agent) against mpc_refill_kernel
(lanes refilled) by agents per wave and hand-back threshold, T = 21, 65,536 - 1 M agents; solutions, status words and costs compared
bit for bit.  JSON lines (profiles/r04/mpc_refill_ab.jsonl)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpprobotics_amd as crx  # noqa: E402
from cpprobotics_amd.experimental import mpc_solve_refill  # noqa: E402
from common import mpc_problem  # noqa: E402


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


T = 21
sizes = [int(a) for a in sys.argv[1:]] or [65536, 262144, 1 << 20]
for n in sizes:
    x0, xref = mpc_problem(n, T, 4)
    x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
    sol0, st0, c0 = crx.mpc_solve(x0, xref, T, return_status=True)
    sweeps = (st0 >> 8).cpu().numpy().astype(np.int64)
    row = {"agents": n, "T": T, "sweeps_mean": float(sweeps.mean()), "sweeps_max": int(sweeps.max()),
           "mean_of_wave_max": float(sweeps[: n // 64 * 64].reshape(-1, 64).max(axis=1).mean()),
           "ms_mpc_kernel": timeit(lambda: crx.mpc_solve(x0, xref, T, return_status=True))}
    ok = True
    for chunk, hold in ((256, 16), (512, 16), (1024, 8), (1024, 16), (1024, 32), (2048, 16), (4096, 16)):
        if n // chunk < 256:
            continue
        row[f"ms_refill_{chunk}_{hold}"] = timeit(lambda: mpc_solve_refill(x0, xref, T, chunk, hold, poison=False))
        sol, st, c = mpc_solve_refill(x0, xref, T, chunk, hold)
        ok = ok and torch.equal(sol.view(torch.int32), sol0.view(torch.int32)) and torch.equal(st, st0) and torch.equal(c.view(torch.int64), c0.view(torch.int64))
    best = min((row[k], k) for k in row if k.startswith("ms_refill_"))
    row.update(bit_identical=bool(ok), best=best[1], speedup=row["ms_mpc_kernel"] / best[0], solves_per_s_best=n / best[0] * 1e3)
    print(json.dumps(row), flush=True)

#!/usr/bin/env python
"""A/B of the MPC horizon solve at BASELINE configs[3] size (8,192 agents, T = 21) on several problem draws: the engine's solver
(crx_mpc_solve_batch_dev, one agent per lane, 128 waves) against the four-variant portfolio (crx_mpc_solve_portfolio_batch_dev, an agent
on a quad of lanes, 512 waves).  Per seed: kernel time of both, slowest agent and mean sweeps of both, agents whose cost differs.
JSON lines (profiles/r04/mpc_portfolio_ab.jsonl)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cpprobotics_amd as crx  # noqa: E402
from common import mpc_problem  # noqa: E402


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


for n, T in ((8192, 21), (8192, 6), (2048, 21), (16384, 21)):
    for seed in (4, 5, 6, 7, 8, 9):
        if (n, T) != (8192, 21) and seed > 5:
            continue
        x0, xref = mpc_problem(n, T, seed)
        x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
        ms1 = timeit(lambda: crx.mpc_solve(x0, xref, T), 7)
        ms4 = timeit(lambda: crx.mpc_solve(x0, xref, T, portfolio=True), 7)
        _, s1, c1 = crx.mpc_solve(x0, xref, T, return_status=True)
        _, s4, c4 = crx.mpc_solve(x0, xref, T, return_status=True, portfolio=True)
        s1, s4, c1, c4 = s1.cpu().numpy(), s4.cpu().numpy(), c1.cpu().numpy(), c4.cpu().numpy()
        i1, i4 = s1 >> 8, s4 >> 8
        d = (c4 - c1) / np.maximum(1.0, np.abs(c1))
        print(json.dumps({"agents": n, "T": T, "seed": seed, "ms_single": ms1, "ms_portfolio": ms4, "portfolio_speedup": ms1 / ms4,
                          "sweeps_single": {"mean": float(i1.mean()), "max": int(i1.max()), "converged": float((s1 & 1).mean())},
                          "sweeps_portfolio": {"mean": float(i4.mean()), "max": int(i4.max()), "converged": float((s4 & 1).mean())},
                          "cost_lower": int((d < -1e-9).sum()), "cost_higher": int((d > 1e-9).sum()),
                          "winners": np.bincount((s4 >> 2) & 3, minlength=4).tolist()}), flush=True)

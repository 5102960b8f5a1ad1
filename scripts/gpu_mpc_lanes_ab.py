#!/usr/bin/env python
"""A/B of the MPC horizon solve: one agent per lane (mpc_kernel) against four lanes per agent (mpc_quad_kernel: the line search's
step lengths rolled out side by side by a DPP quad), at BASELINE configs[3] (8,192 agents, T = 21) and in the throughput regime.
One JSON line per (T, batch); profiles/r03/mpc_lanes_ab.txt is this script's output."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cpprobotics_amd.experimental import mpc_solve_lanes  # noqa: E402
from common import mpc_problem  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2], ms[0]


def run(T, n, reps, seed=4):
    x0, xref = mpc_problem(n, T, seed)
    x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
    out = {"T": T, "agents": n}
    res = {}
    for lanes in (1, 4):
        med, mn = timed(lambda: mpc_solve_lanes(x0, xref, T, lanes), reps)
        sol, st, cost = mpc_solve_lanes(x0, xref, T, lanes)
        res[lanes] = (sol.cpu().numpy(), st.cpu().numpy(), cost.cpu().numpy())
        out[f"lanes{lanes}"] = {"ms_median": round(med, 4), "ms_min": round(mn, 4), "solves_per_s": round(n / (med * 1e-3)), "waves": (n * lanes + 63) // 64}
    it = res[1][1] >> 8
    per_wave = lambda k: float(np.mean([it[i:i + k].max() for i in range(0, n, k)]))
    out["sweeps"] = {"mean": float(it.mean()), "max": int(it.max()), "mean_of_wave_max_64_agents": per_wave(64), "mean_of_wave_max_16_agents": per_wave(16)}
    a, b = res[1], res[4]
    out["status_and_sweeps_equal"] = bool(np.array_equal(a[1], b[1]))
    out["max_rel_diff_of_solutions"] = float(np.max(np.abs(a[0] - b[0]) / np.maximum(np.abs(a[0]), 1.0)))
    out["max_rel_diff_of_costs"] = float(np.max(np.abs(a[2] - b[2]) / np.maximum(np.abs(a[2]), 1.0)))
    out["speed_of_quad_variant"] = round(out["lanes1"]["ms_median"] / out["lanes4"]["ms_median"], 3)
    print(json.dumps(out), flush=True)


for T, n, reps in ((21, 8192, 30), (21, 16384, 20), (21, 32768, 15), (21, 65536, 10), (6, 8192, 30), (6, 65536, 15)):
    run(T, n, reps)

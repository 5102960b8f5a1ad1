#!/usr/bin/env python
"""A/B of the structured Riccati solve (A, B from v): one agent per lane (dare_from_v_kernel) against four lanes per agent
(dare_from_v_quad_kernel: a DPP quad holds the 4x4 block of X row by row), at BASELINE configs[2] (16,384 agents) and in the
throughput regime.  One JSON line per (dim, batch); profiles/r03/dare_lanes_ab.txt is this script's output."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cpprobotics_amd.experimental import dlqr_from_v_lanes  # noqa: E402
from common import lqr_speeds  # noqa: E402


def timed(fn, reps):
    for _ in range(max(5, reps // 4)):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2], ms[0]


def run(dim, n, reps):
    v = torch.from_numpy(lqr_speeds(n, seed=3)).cuda()
    out = {"dim": dim, "agents": n}
    res = {}
    for lanes in (1, 4):
        med, mn = timed(lambda: dlqr_from_v_lanes(v, dim, lanes), reps)
        K, X, it = dlqr_from_v_lanes(v, dim, lanes)
        res[lanes] = (K.cpu().numpy(), X.cpu().numpy(), it.cpu().numpy())
        out[f"lanes{lanes}"] = {"ms_median": round(med, 5), "ms_min": round(mn, 5), "solves_per_s": round(n / (med * 1e-3)),
                                "waves": (n * lanes + 63) // 64}
    it = res[1][2]
    per_wave = lambda k: float(np.mean([it[i:i + k].max() for i in range(0, n, k)]))
    out["iterations"] = {"mean": float(it.mean()), "max": int(it.max()), "mean_of_wave_max_64_agents": per_wave(64),
                         "mean_of_wave_max_16_agents": per_wave(16)}
    out["results_equal_as_ieee_values"] = bool(all(np.array_equal(a, b) for a, b in zip(res[1], res[4])))
    out["speed_of_quad_variant"] = round(out["lanes1"]["ms_median"] / out["lanes4"]["ms_median"], 3)
    print(json.dumps(out), flush=True)


for dim in (5, 4):
    for n, reps in ((16384, 200), (32768, 100), (65536, 100), (131072, 60), (262144, 40), (1048576, 20)):
        run(dim, n, reps)

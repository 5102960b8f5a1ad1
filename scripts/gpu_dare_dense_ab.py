#!/usr/bin/env python
"""A/B of the DENSE Riccati kernels (crx_x_dare_batch_dense_dev): one agent per lane (dare_dense_kernel) against one row of X per lane
of a quad (dare_dense_quad_kernel), 5x5 and 4x4, on general dense matrices (eps 1e-3, cap 60) and on the reference's own matrices
(eps 0.01, cap 150: the agents at the cap set the launch time).  JSON lines (profiles/r04/dare_dense_lanes_ab.jsonl)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cpprobotics_amd as crx  # noqa: E402
from cpprobotics_amd.experimental import dare_dense  # noqa: E402
from common import lqr_speeds  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)


def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


for dim in (5, 4):
    m = 2 if dim == 5 else 1
    for n in (4096, 16384, 32768, 65536):
        rng = np.random.default_rng(5)
        A = torch.from_numpy((np.eye(dim)[None] * 0.9 + 0.15 * rng.standard_normal((n, dim, dim))).astype(np.float32).reshape(n, -1)).cuda()
        B = torch.from_numpy(rng.standard_normal((n, dim * m)).astype(np.float32)).cuda()
        Q = torch.eye(dim, device="cuda").reshape(1, -1).repeat(n, 1).contiguous()
        R = torch.eye(m, device="cuda").reshape(1, -1).repeat(n, 1).contiguous()
        row = {"dim": dim, "agents": n, "matrices": "dense random (eps 1e-3, cap 60)"}
        for lanes in (1, 4):
            row[f"ms_lanes{lanes}"] = timeit(lambda: dare_dense(A, B, Q, R, eps=1e-3, maxiter=60, lanes_per_agent=lanes))
        row["quad_speedup"] = row["ms_lanes1"] / row["ms_lanes4"]
        x1, x4 = dare_dense(A, B, Q, R, eps=1e-3, maxiter=60, lanes_per_agent=1), dare_dense(A, B, Q, R, eps=1e-3, maxiter=60, lanes_per_agent=4)
        row["bit_identical"] = bool(all(torch.equal(a_.view(torch.int32), b_.view(torch.int32)) for a_, b_ in zip(x1, x4)))
        row["solves_per_s_lanes4"] = n / (row["ms_lanes4"] * 1e-3)
        print(json.dumps(row), flush=True)
        if dim == 5:
            Ar, Br, Qr, Rr = (torch.from_numpy(a_).cuda() for a_ in bench.lqr_pattern_mats(lqr_speeds(n, 3)))
            row = {"dim": dim, "agents": n, "matrices": "the reference's (eps 0.01, cap 150), dense kernels forced"}
            for lanes in (1, 4):
                row[f"ms_lanes{lanes}"] = timeit(lambda: dare_dense(Ar, Br, Qr, Rr, lanes_per_agent=lanes), 5)
            row["quad_speedup"] = row["ms_lanes1"] / row["ms_lanes4"]
            row["solves_per_s_lanes4"] = n / (row["ms_lanes4"] * 1e-3)
            print(json.dumps(row), flush=True)

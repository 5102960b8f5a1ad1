#!/usr/bin/env python3
"""A/B of the throughput-regime Riccati kernels (one agent per lane): the masked kernel (lanes idle behind their wave's slowest
agent) against the lane-refilling kernel (dare_from_v_refill_kernel) by agents per wave, 5x5 and 4x4, 131,072 - 4 M agents of
BASELINE configs[2]'s speed distribution; results compared bit for bit (X, K, iteration counts).  JSON lines
(profiles/r04/dare_refill_ab.jsonl)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpprobotics_amd as crx  # noqa: E402,F401
from cpprobotics_amd.experimental import dlqr_from_v_refill  # noqa: E402
from common import lqr_speeds  # noqa: E402


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for dim in (5, 4):
    for n in (131072, 262144, 1 << 20, 1 << 22):
        v = torch.from_numpy(lqr_speeds(n, 3)).cuda()
        Km, Xm, itm = dlqr_from_v_refill(v, dim, -1)
        row = {"dim": dim, "agents": n, "mean_iters": float(itm.float().mean()), "mean_of_wave_max": float(itm.view(-1, 64).max(dim=1).values.float().mean()),
               "ms_masked": timeit(lambda: dlqr_from_v_refill(v, dim, -1, poison=False))}
        ok = True
        for chunk, hold in ((256, 16), (512, 1), (512, 4), (512, 8), (512, 16), (512, 32), (1024, 8), (1024, 16), (1024, 32), (2048, 16)):
            row[f"ms_refill_{chunk}_{hold}"] = timeit(lambda: dlqr_from_v_refill(v, dim, chunk, hold, poison=False))
            K, X, it = dlqr_from_v_refill(v, dim, chunk, hold)
            ok = ok and torch.equal(X.view(torch.int32), Xm.view(torch.int32)) and torch.equal(K.view(torch.int32), Km.view(torch.int32)) and torch.equal(it, itm)
        best = min((row[k], k) for k in row if k.startswith("ms_refill_"))
        row.update(bit_identical=bool(ok), best=best[1], speedup=row["ms_masked"] / best[0], solves_per_s_best=n / best[0] * 1e3)
        print(json.dumps(row), flush=True)
    # ragged sizes and other caps: equality only
    for n, maxiter, eps in ((98304 + 77, 150, 0.01), (200001, 7, 0.01), (150000, 1, 0.01), (131072, 150, 1e9), (131072, 40, 1e-4)):
        v = torch.from_numpy(lqr_speeds(n, 5)).cuda()
        Km, Xm, itm = dlqr_from_v_refill(v, dim, -1, eps=eps, maxiter=maxiter)
        res = {}
        for chunk in (64, 100, 512, 777, 4096):
            K, X, it = dlqr_from_v_refill(v, dim, chunk, 1 + chunk % 23, eps=eps, maxiter=maxiter)
            res[chunk] = bool(torch.equal(X.view(torch.int32), Xm.view(torch.int32)) and torch.equal(K.view(torch.int32), Km.view(torch.int32)) and torch.equal(it, itm))
        print(json.dumps({"dim": dim, "agents": n, "maxiter": maxiter, "eps": eps, "equal_by_chunk": res}), flush=True)

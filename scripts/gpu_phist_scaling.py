#!/usr/bin/env python3
"""The fused EKF launch with the covariance history written out (96 B per update) by batch size: 65,536 vehicles are one
wave per SIMD — a wave that waits for its stores to drain has nobody to hand the SIMD to; with 2, 4, 8 waves queued per SIMD the same
kernel shows what the memory system takes from it.  JSON lines -> profiles/rNN/ekf_phist_scaling.jsonl."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpprobotics_amd as crx  # noqa: E402
from common import ekf_QR  # noqa: E402

dev = torch.device("cuda:0")
Q, R = ekf_QR()


def timeit(fn, reps=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best, tot = 1e30, 0.0
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1); best = min(best, ms); tot += ms
    return best, tot / reps


for n, T in ((65536, 250), (131072, 250), (262144, 125), (524288, 64), (1048576, 32)):
    x = torch.zeros((n, 4), dtype=torch.float32, device=dev); x[:, 2] = 0.3
    P = torch.eye(4, dtype=torch.float32, device=dev).reshape(1, 16).repeat(n, 1).contiguous()
    z = torch.rand((T, n, 2), dtype=torch.float32, device=dev); u = torch.rand((T, n, 2), dtype=torch.float32, device=dev) * 0.2
    xh = torch.empty((T, n, 4), dtype=torch.float32, device=dev); Ph = torch.empty((T, n, 16), dtype=torch.float32, device=dev)
    row = {"vehicles": n, "steps": T, "waves_per_simd": n / 64 / 1024.0}
    for label, kw, bpu in (("with_P_history", dict(x_hist=xh, P_hist=Ph), 96.0), ("x_history_only", dict(x_hist=xh), 32.0)):
        xs, Ps = x.clone(), P.clone()
        best, mean = timeit(lambda: crx.ekf_run(xs, Ps, z, u, Q, R, **kw))
        row[label] = {"ms_best": best, "ms_mean": mean, "TB_per_s_best": (bpu * n * T + 160 * n) / best / 1e9,
                      "frac_of_8TBps_best": (bpu * n * T + 160 * n) / best / 1e9 / 8.0, "G_updates_per_s_best": n * T / best / 1e6}
    print(json.dumps(row), flush=True)
    del x, P, z, u, xh, Ph

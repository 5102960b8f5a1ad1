#!/usr/bin/env python3
"""HBM calibration of the box: what plain streaming kernels reach (crx_x_hbm_stream_dev: copy / read / write / in-place update, by
workgroup count and buffer size), torch's own device copy, and beside them the HBM-bound EKF launches priced the same way — the
single-step update of 4 M vehicles (176 B per update) and the fused run with the covariance history (96 B per update).  JSON lines
(profiles/r04/hbm_calibration.jsonl)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cpprobotics_amd as crx  # noqa: E402
from cpprobotics_amd.experimental import hbm_stream  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best, tot = 1e30, 0.0
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1); best = min(best, ms); tot += ms
    return best, tot / reps


dev = torch.device("cuda:0")
out = []
for mib in (64, 704, 2048):
    nbytes = mib << 20
    src = torch.ones(nbytes // 4, dtype=torch.float32, device=dev)
    dst = torch.zeros_like(src)
    for mode, name, moved in ((0, "copy", 2), (1, "read", 1), (2, "write", 1), (3, "update_in_place", 2)):
        for wgs in (2048, 8192, 32768, nbytes // 4096):
            best, mean = timeit(lambda: hbm_stream(mode, dst, src, workgroups=wgs))
            out.append({"kernel": "hbm_stream", "mode": name, "MiB": mib, "workgroups": wgs, "ms_best": best, "ms_mean": mean,
                        "TB_per_s_best": moved * nbytes / best / 1e9, "TB_per_s_mean": moved * nbytes / mean / 1e9})
            print(json.dumps(out[-1]), flush=True)
    best, mean = timeit(lambda: dst.copy_(src))
    out.append({"kernel": "torch copy_", "MiB": mib, "ms_best": best, "TB_per_s_best": 2 * nbytes / best / 1e9, "TB_per_s_mean": 2 * nbytes / mean / 1e9})
    print(json.dumps(out[-1]), flush=True)
    del src, dst

# the HBM-bound EKF launches, same clock
rng = np.random.default_rng(0)
n = 4 << 20
x = torch.zeros((n, 4), dtype=torch.float32, device=dev); x[:, 2] = torch.rand(n, device=dev) * 6 - 3
P = torch.eye(4, dtype=torch.float32, device=dev).reshape(1, 16).repeat(n, 1).contiguous()
z = torch.rand((n, 2), dtype=torch.float32, device=dev); u = torch.rand((n, 2), dtype=torch.float32, device=dev)
Q = np.diag([0.1, 0.1, np.deg2rad(1.0), 1.0]).astype(np.float32) ** 2       # extended_kalman_filter.cpp:142-151
R = np.eye(2, dtype=np.float32)
step = lambda: crx.ekf_estimation(x, P, z, u, Q, R)
best, mean = timeit(step)
row = {"kernel": "ekf_step_kernel", "vehicles": n, "bytes_per_update": 176, "ms_best": best, "ms_mean": mean,
       "TB_per_s_best": 176 * n / best / 1e9, "TB_per_s_mean": 176 * n / mean / 1e9}
print(json.dumps(row), flush=True)
del x, P, z, u
n, T = 65536, 250
x = torch.zeros((n, 4), dtype=torch.float32, device=dev); P = torch.eye(4, dtype=torch.float32, device=dev).reshape(1, 16).repeat(n, 1).contiguous()
z = torch.rand((T, n, 2), dtype=torch.float32, device=dev); u = torch.rand((T, n, 2), dtype=torch.float32, device=dev) * 0.2
xh = torch.empty((T, n, 4), dtype=torch.float32, device=dev); Ph = torch.empty((T, n, 16), dtype=torch.float32, device=dev)
for label, kw, bpu in (("ekf_run_kernel + P history", dict(x_hist=xh, P_hist=Ph), 96.0), ("ekf_run_kernel", dict(x_hist=xh), 32.0)):
    xs, Ps = x.clone(), P.clone()
    best, mean = timeit(lambda: crx.ekf_run(xs, Ps, z, u, Q, R, **kw), 10)
    row = {"kernel": label, "vehicles": n, "steps": T, "bytes_per_update": bpu, "ms_best": best, "ms_mean": mean,
           "TB_per_s_best": (bpu * n * T + 160 * n) / best / 1e9, "TB_per_s_mean": (bpu * n * T + 160 * n) / mean / 1e9}
    print(json.dumps(row), flush=True)

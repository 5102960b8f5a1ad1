#!/usr/bin/env python3
"""The fused EKF launch with the covariance history (96 B per update, HBM-bound) and without it (32 B, VALU-bound) by batch size:
does the HBM rate of the history variant rise with more waves per SIMD?  JSON lines."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cpprobotics_amd as crx  # noqa: E402

dev = torch.device("cuda:0")
Q = np.diag([0.1, 0.1, np.deg2rad(1.0), 1.0]).astype(np.float32) ** 2
R = np.eye(2, dtype=np.float32)
for n, T in ((65536, 250), (131072, 125), (262144, 64), (1 << 20, 16), (65536, 1000)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.zeros((n, 4), dtype=torch.float32, device=dev); x[:, 2] = torch.rand(n, device=dev, generator=g) * 6 - 3
    P = torch.eye(4, dtype=torch.float32, device=dev).reshape(1, 16).repeat(n, 1).contiguous()
    z = torch.rand((T, n, 2), dtype=torch.float32, device=dev, generator=g); u = torch.rand((T, n, 2), dtype=torch.float32, device=dev, generator=g) * 0.2
    xh = torch.empty((T, n, 4), dtype=torch.float32, device=dev)
    Ph = torch.empty((T, n, 16), dtype=torch.float32, device=dev) if T <= 250 else None
    for label, kw, bpu in (("P history", dict(x_hist=xh, P_hist=Ph), 96.0), ("x history only", dict(x_hist=xh), 32.0)):
        if kw.get("P_hist", 1) is None:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for i in range(12):
            xs, Ps = x.clone(), P.clone()
            e0.record(); crx.ekf_run(xs, Ps, z, u, Q, R, **kw); e1.record(); e1.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        print(json.dumps({"vehicles": n, "steps": T, "out": label, "ms_median": ms, "ms_min": float(min(ts)),
                          "TB_per_s": (bpu * n * T + 160 * n) / ms / 1e9, "G_updates_per_s": n * T / ms / 1e6}), flush=True)
    del x, P, z, u, xh, Ph

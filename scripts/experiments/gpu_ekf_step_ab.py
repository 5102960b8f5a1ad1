#!/usr/bin/env python3
"""Times crx_ekf_step_batch_dev (the single-step EKF update, 176 B per update, HBM-bound) for one build of the library — the one
CRX_LIB_PATH names — at several batch sizes, and checks the result against the first build's checksum.  Driven by
gpu_ekf_step_ab.sh, which loops over the builds in scripts/_diag/ (CRX_EKF_STEP_* macros of csrc/ekf_kernels.hip.h)."""
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
row = {"lib": os.path.basename(os.environ.get("CRX_LIB_PATH", "libcrx.so"))}
for n in (1 << 23, 5 << 20, 1 << 22, (1 << 22) - 37, 7 << 19, 13 << 18, 3 << 20, 1 << 21, 1 << 20, 1 << 18):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.zeros((n, 4), dtype=torch.float32, device=dev); x[:, 2] = torch.rand(n, device=dev, generator=g) * 6 - 3
    P = torch.eye(4, dtype=torch.float32, device=dev).reshape(1, 16).repeat(n, 1).contiguous()
    z = torch.rand((n, 2), dtype=torch.float32, device=dev, generator=g); u = torch.rand((n, 2), dtype=torch.float32, device=dev, generator=g)
    x0, P0 = x.clone(), P.clone()
    crx.ekf_estimation(x, P, z, u, Q, R)
    chk = int(x.view(torch.int32).to(torch.int64).sum().item()) ^ int(P.view(torch.int32).to(torch.int64).sum().item())
    for _ in range(5):
        crx.ekf_estimation(x, P, z, u, Q, R)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(40):
        x.copy_(x0); P.copy_(P0)          # the same data every time (and the caches hold what a fresh launch would find)
        e0.record(); crx.ekf_estimation(x, P, z, u, Q, R); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = np.array(ts)
    tb = []
    for _ in range(40):                  # back to back, the state evolving: what a filter loop sees
        e0.record(); crx.ekf_estimation(x, P, z, u, Q, R); e1.record(); e1.synchronize()
        tb.append(e0.elapsed_time(e1))
    row[str(n)] = {"back_to_back_ms_median": float(np.median(tb)), "back_to_back_TB_per_s": 176.0 * n / float(np.median(tb)) / 1e9,"ms_min": float(ts.min()), "ms_median": float(np.median(ts)), "TB_per_s_median": 176.0 * n / float(np.median(ts)) / 1e9, "checksum": chk}
print(json.dumps(row), flush=True)

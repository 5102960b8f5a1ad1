#!/usr/bin/env python
"""MPC solve time against the batch size (same problem generator): does the kernel time follow the slowest agent
(flat while waves <= SIMDs) or the number of waves?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cpprobotics_amd as crx  # noqa: E402
from common import mpc_problem  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 21
x0a, xra = mpc_problem(65536, T, 4)
for n in (64, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 262144, 1048576):
    reps = (n + 65535) // 65536                      # beyond 65,536 agents the generated batch is tiled
    x0 = torch.from_numpy(x0a[:min(n, 65536)]).cuda().repeat(reps, 1).contiguous(); xr = torch.from_numpy(xra[:min(n, 65536)]).cuda().repeat(reps, 1).contiguous()
    sol, st, _ = crx.mpc_solve(x0, xr, T, return_status=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        crx.mpc_solve(x0, xr, T)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    it = st.cpu().numpy() >> 8
    print(f"n={n:6d} waves={n // 64:5d}  {ms:8.3f} ms  {n / ms / 1e3:7.3f} M solves/s  max_it={it.max()} mean_it={it.mean():.2f} at_cap={(it >= 50).sum()}")

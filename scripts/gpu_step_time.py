#!/usr/bin/env python
"""Single-step EKF kernel (crx_ekf_step_batch_dev) over batch sizes: updates/s and HBM rate at 176 B per update."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import cpprobotics_amd as crx  # noqa: E402
from common import ekf_QR  # noqa: E402

Q, R = ekf_QR()
for n in (65536, 1 << 20, 1 << 22, (1 << 22) + 37):
    x = torch.randn((n, 4), device="cuda")
    A = torch.randn((n, 4, 4), device="cuda")
    P = (A @ A.transpose(1, 2)).reshape(n, 16).contiguous()
    z, u = torch.randn((n, 2), device="cuda"), torch.rand((n, 2), device="cuda")
    for _ in range(5):
        crx.ekf_estimation(x, P, z, u, Q, R)
    torch.cuda.synchronize()
    reps = 50
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        crx.ekf_estimation(x, P, z, u, Q, R)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"n={n:8d}  {ms * 1e3:8.1f} us/launch  {n / ms / 1e6:7.2f} G updates/s  {176.0 * n / ms / 1e9:6.2f} TB/s ({176.0 * n / ms / 8e9 * 100:.1f} % of 8 TB/s)")

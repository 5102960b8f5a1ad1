#!/usr/bin/env python
"""Single-step EKF API (one crx_ekf_step_batch_dev per tick) at the BASELINE batch: plain launches vs a captured HIP graph
of K consecutive ticks (the per-tick entry points only enqueue, so they can be captured)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import cpprobotics_amd as crx  # noqa: E402
from common import ekf_QR  # noqa: E402

Q, R = ekf_QR()
n, K = 65536, 50
x = torch.zeros((n, 4), device="cuda")
P = torch.eye(4, device="cuda").reshape(1, 16).repeat(n, 1).contiguous()
z = torch.randn((K, n, 2), device="cuda")
u = torch.rand((K, n, 2), device="cuda")


def ticks():
    for t in range(K):
        crx.ekf_estimation(x, P, z[t], u[t], Q, R)


ticks(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ticks()
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / (20 * K)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    ticks()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    ticks()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / (20 * K)
print(f"n={n}: plain launches {plain * 1e6:.2f} us/tick = {n / plain / 1e9:.2f} G updates/s; graph of {K} ticks {graph * 1e6:.2f} us/tick = {n / graph / 1e9:.2f} G updates/s")

#!/usr/bin/env python
"""The MPC solve in its throughput regime under rocprofv3 (scripts/gpu_mpc_traffic.sh): mpc_kernel (private memory), mpc_tile_kernel
(LDS + accumulator registers, round 6) and — where the A/B library is present — the lane-refilling kernel; N agents (default 262,144),
T = 21, `reps` launches each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from cpprobotics_amd import experimental as X
from common import mpc_problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
x0, xref = mpc_problem(n, 21, 4)
x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["private", "tile", "refill"]
for v in variants:
    for _ in range(reps):
        if v == "tile_refill":
            X.mpc_solve_tile_refill(x0, xref, 21, max(128, (n + 1023) // 1024), 32 if (n + 1023) // 1024 <= 512 else 16)   # the product's geometry
        elif v == "refill":
            X.mpc_solve_refill(x0, xref, 21, 512, 16, poison=False)
        elif v == "tile":
            X.mpc_solve_store(x0, xref, 21, 1)
        elif v == "tile_lite":
            X.mpc_solve_store(x0, xref, 21, 3)
        elif v == "tile2":
            X.mpc_solve_store(x0, xref, 21, 2)
        elif v == "tile2_refill":
            X.mpc_solve_tile_refill(x0, xref, 21, max(128, (n + 1023) // 1024), 32 if (n + 1023) // 1024 <= 512 else 16, store=2)
        else:
            X.mpc_solve_store(x0, xref, 21, 0)
    torch.cuda.synchronize()

#!/usr/bin/env python
"""Kernel-only timing of the structured Riccati variants at the BASELINE batch (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from cpprobotics_amd.experimental import dlqr_from_v_lanes
from common import lqr_speeds
for n in (16384, 32768):
    v = torch.from_numpy(lqr_speeds(n, seed=3)).cuda()
    for dim in (5, 4):
        for lanes in (1, 4):
            for _ in range(50):
                dlqr_from_v_lanes(v, dim, lanes)
            torch.cuda.synchronize()

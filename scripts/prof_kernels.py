#!/usr/bin/env python
"""The DARE and MPC launches of BASELINE configs[2] / configs[3], a few times each, for rocprofv3 (--kernel-trace --stats or --pmc).
Prints the iteration statistics the counter post-processing needs (scripts/gpu_prof2.sh)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cpprobotics_amd as crx  # noqa: E402
from common import lqr_speeds, mpc_problem  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
v = torch.from_numpy(lqr_speeds(16384, 3)).cuda()
for _ in range(reps):
    K, X, it = crx.dlqr_from_v(v, dim=5)
it5 = it.cpu().numpy().astype(np.int64)
x0, xref = mpc_problem(8192, 21, 4)
x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
for _ in range(reps):
    sol, st, cost = crx.mpc_solve(x0, xref, 21, return_status=True)
torch.cuda.synchronize()
mit = (st.cpu().numpy() >> 8).astype(np.int64)
print(json.dumps({"dare5": {"agents": 16384, "iters_sum": int(it5.sum()), "wave_max_iters_sum": int(it5.reshape(-1, 64).max(axis=1).sum())},
                  "mpc_T21": {"agents": 8192, "iters_sum": int(mit.sum()), "iters_max": int(mit.max()),
                              "wave_max_iters_sum": int(mit.reshape(-1, 64).max(axis=1).sum())}}))
